"""
N>1 path on CPU: world_size-2 gloo.  The hot path shards by games (no data-path
collective); the ONE exchange step is the flat-weight broadcast
(mzx.shared_storage.broadcast_weights; RCCL on the GPU box, gloo here).  The
model in these processes is bound to the serial test double of the ABI
(tests/hostcheck), which exercises the same host code: flat buffer layout,
refresh of derived BatchNorm terms, shard seeding.
"""
import os
import socket
import sys

import numpy
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "muzero-general_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hostcheck
    from mzx import configs, models, self_play, shared_storage, synthetic

    be = hostcheck.backend()
    cfg = configs.tictactoe()  # resnet: has BatchNorm-derived terms that must be refreshed
    net = models.MuZeroNetwork(cfg, _backend=be)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 40 + rank))  # ranks start different
    before = net.flat_weights().clone()
    shared_storage.broadcast_weights(net, src=0)
    after = net.flat_weights().clone()
    # every rank searches the SAME roots with the SAME streams: results must now coincide
    B = 4
    obs = synthetic.observations(B, cfg.observation_shape, seed=77)
    engine = self_play.BatchedMCTS(cfg, net, B)
    res = engine.run(list(obs), [list(cfg.action_space)] * B, [0] * B, True,
                     [numpy.random.RandomState(5 + i) for i in range(B)])
    seeds = shared_storage.shard_seeds(cfg.seed, 3)
    torch.save(dict(before=before, after=after, visits=res.visit_counts, root_values=res.root_values, seeds=seeds),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert not torch.equal(r0["before"], r1["before"])
    assert torch.equal(r0["after"], r0["before"])            # the source keeps its weights
    assert torch.equal(r1["after"], r0["after"])             # the other rank received them
    assert numpy.array_equal(r0["visits"], r1["visits"])     # ... and its derived BN terms were refreshed
    assert numpy.array_equal(r0["root_values"], r1["root_values"])
    assert r0["seeds"] == [0, 1, 2] and r1["seeds"] == [3, 4, 5]  # disjoint game shards (muzero.py:185)
