"""
TEST INFRASTRUCTURE ONLY -- runs many CPU-oracle searches (oracle/mcts_oracle.py + net_oracle.py) in a pool of
worker processes, one torch thread each, so that the at-size GPU parity tests can compare hundreds of trees against
the oracle in seconds.  Workers are spawned (never forked: the parent holds a GPU context) and build the oracle
network once.  Only tests/ imports this.
"""
import multiprocessing
import os
import sys

import numpy

_STATE = {}


def _init(paths, cfg, sd, dtype_name):
    for p in paths:
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch

    torch.set_num_threads(1)
    from oracle import net_oracle

    _STATE["cfg"] = cfg
    _STATE["net"] = net_oracle.make_oracle_network(cfg, sd, dtype=getattr(torch, dtype_name))


def _one(job):
    from oracle import mcts_oracle, net_oracle

    obs, legal, to_play, seed = job
    cfg = _STATE["cfg"]
    ev = net_oracle.NetworkEvaluator(_STATE["net"], cfg.support_size)
    tree = mcts_oracle.run_search(cfg, ev, obs, legal, to_play, True, numpy.random.RandomState(seed))
    return dict(trace=[(p, a) for p, a, _ in tree.trace], margins=list(tree.margins), value_margins=list(tree.value_margins),
                root_visit_counts=tree.root_visit_counts(cfg.action_space), root_value=tree.node_value(0),
                max_depth=tree.max_depth)


def run_searches(cfg, sd, jobs, processes=None, dtype_name="float32"):
    """jobs: list of (observation, legal_actions, to_play, seed); returns one summary dict per job, in order."""
    processes = processes or max(1, min(len(jobs), (os.cpu_count() or 2) - 2, 48))
    if processes <= 1 or len(jobs) < 4:
        _init(list(sys.path), cfg, sd, dtype_name)
        return [_one(j) for j in jobs]
    ctx = multiprocessing.get_context("spawn")
    with ctx.Pool(processes, initializer=_init, initargs=(list(sys.path), cfg, sd, dtype_name)) as pool:
        return pool.map(_one, jobs, chunksize=max(1, len(jobs) // (4 * processes)))
