"""
GPU parity of the streamed MFMA engine AT THE SIZES bench.py TIMES (-m gpu, through the C ABI).

tests/test_gpu_streamed.py pins the engine on small batches, which launch rb_gemm_kernel<1,1> / <2,1> only.  The
workloads behind the reported roofline fractions -- games/gomoku.py at 512 / 1024 trees, games/connect4.py forced
onto this engine at 4608 / 9216 trees, games/atari.py at 256 / 512 / 1024 trees -- launch OTHER code: the trunks as
rb_tower_kernel<8,1> / <5,2> (a whole conv + residual-block run per launch, round 4) and rb_gemm_kernel<8,1>, <9,1>,
<9,2>, <6,2>, <5,1>, <4,1> (position fragments refilled in place, two to four channel phases staged across barriers).  This file runs exactly those launches (tests/streamed_coverage.py is the
table; tests/test_streamed_coverage.py, CPU, keeps bench.py inside it):

  (a) every operator of both programs against the one-element-kernel-per-operator engine (mode 0), at size;
  (b) all heads of sampled rows against the CPU oracle network (oracle/net_oracle.py), 1e-4 (north_star); rows on
      which the two device engines disagree (near-flat planes in front of the per-plane min-max scaling) are held to
      the binary64 yardstick instead of being waived;
  (c) whole searches at size against the CPU oracle, simulation by simulation, on sampled trees (64 of connect4 at
      9216 trees, 16 of gomoku at 1024, 8 + 4 of atari), with the UCB-margin gate at a first divergence -- and, next to
      the device's divergence rate, the rate at which the ORACLE'S OWN fp32 search diverges from its binary64
      evaluation on the same trees;
  (d) the two half-shards on two HIP streams, at the DEFAULT threshold, against the undivided run: every statistic
      of every tree bit for bit.
Each test asserts which (MT, NT, phases, K loop) it launched.
"""
import math
import os

import numpy
import pytest
import torch

import at_size
import streamed_coverage as sc
import test_gpu_parity as parity
from mzx import _lib, configs, models, self_play, synthetic
from oracle import net_oracle, parallel

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star


@pytest.fixture(scope="module")
def backend():
    return _lib.default_backend()


@pytest.fixture(autouse=True)
def _launch_by_launch(backend):
    """This file tests the streamed engine's LAUNCHES at the sizes the bench times; a search the library would run as one
    launch of rt_search_kernel stays on the per-simulation launches here (its own at-size cases:
    tests/test_gpu_tower_search.py).  conftest restores the default afterwards."""
    backend.lib.tuning_set("rt_search", 0)


# The instantiations every case must launch (MT, NT, channel phases, K loop): hard-wired, so that a planner change that
# moves a workload onto other code paths fails HERE and has to be acknowledged (together with streamed_coverage)
MUST_LAUNCH = {
    # trunks as towers (rb_tower_kernel): one sample (gomoku) / two to six boards (connect4) / two samples (atari) per
    # workgroup, with the scaling and the small 1x1 head convolutions in their tails; head MLPs: one rb_gemm_multi_kernel<1,1>
    # launch per level (the same layers' shapes as slices)
    "gomoku-512": {(8, 1, 1, "tower in-place"), (1, 1, 1, "ring grouped")},
    "gomoku-1024": {(8, 1, 1, "tower in-place"), (1, 1, 1, "ring grouped")},
    "connect4-512": {(3, 1, 1, "tower two-sets"), (1, 1, 1, "ring grouped")},
    "connect4-1024": {(6, 1, 1, "tower in-place"), (1, 1, 1, "ring grouped")},
    "connect4-4608": {(8, 1, 1, "tower in-place"), (1, 1, 1, "ring grouped")},
    "connect4-9216": {(8, 1, 1, "tower in-place"), (1, 1, 1, "ring grouped")},
    # 256 trees: one 6 x 6 sample per workgroup (36 of 48 rows) so that the chip is filled -- towers all the same (round 6:
    # whether a trunk runs as a tower is a property of the network, not of the batch; round 5 launched these layer by layer)
    "atari-256": {(3, 2, 1, "tower two-sets"), (5, 1, 1, "two-sets"), (9, 1, 2, "in-place"), (9, 2, 2, "in-place"),
                  (6, 2, 4, "in-place"), (6, 2, 2, "in-place"), (4, 1, 3, "ring"), (1, 1, 4, "ring")},
    "atari-512": {(5, 2, 1, "tower in-place"), (9, 1, 1, "in-place"), (9, 2, 4, "in-place"), (6, 2, 4, "in-place"),
                  (6, 2, 2, "in-place"), (4, 1, 3, "ring"), (1, 1, 4, "ring")},
    "atari-1024": {(5, 2, 1, "tower in-place"), (9, 2, 1, "in-place"), (9, 2, 4, "in-place"), (9, 1, 2, "in-place"),
                   (6, 2, 4, "in-place"), (6, 2, 2, "in-place"), (4, 1, 3, "ring"), (1, 2, 1, "ring")},
    # layer by layer (modes 4 / 5)
    "connect4-4608-layers": {(8, 1, 2, "in-place"), (8, 1, 1, "in-place"), (1, 1, 1, "ring")},
    "gomoku-512-layers": {(8, 1, 2, "in-place"), (8, 1, 1, "in-place"), (1, 1, 1, "ring")},
}


def _network(game, mode, seed, weights="synthetic"):
    """weights = "synthetic": mzx.synthetic.fill_state_dict(seed) -- large random weights, searches that dig single lines
    a hundred plies deep, near-flat planes in front of the min-max scaling: the stress case.  "reference": what SURVEY.md
    section 8(d) prescribes and bench.py times -- torch.manual_seed(0); models.MuZeroNetwork(config) of the UNMODIFIED
    reference, from the oracle/_ref bytecode (skipped where that did not travel)."""
    cfg = configs.BY_NAME[game]()
    net = models.MuZeroNetwork(cfg)
    if weights == "reference":
        from oracle import build_ref
        if not build_ref.available():
            pytest.skip("oracle/_ref not built (no /root/reference at build time)")
        ref_models, _ = build_ref.load()
        torch.manual_seed(0)
        sd = {k: v.clone() for k, v in ref_models.MuZeroNetwork(cfg).get_weights().items()}
    else:
        sd = synthetic.fill_state_dict(net.state_dict(), seed)
    net.set_weights(sd)
    if mode is not None:
        net.set_mode(mode)
    return cfg, net, sd


# How far a row's logits may be from the oracle evaluated in binary64, in units of the oracle's OWN fp32 error on that
# row, when it misses 1e-4 against the fp32 oracle (ill-conditioned rows: a near-flat plane in front of the min-max
# scaling).  Synthetic weights: 32 x (observed up to 18 x: gomoku's 1 152-term fp32 chains against ATen's blocked sums);
# the reference constructor's weights -- the case the north star's 1e-4 is stated for -- 4 x.
OWN_ERROR_FACTOR = {"synthetic": 32.0, "reference": 4.0}


def _device_rand(shape, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.rand(shape, device="cuda", generator=g)


def _rows_against_oracle(cfg, sd, rows, obs, hid, act, outs, label, loose=(), factor=32.0):
    """
    Device heads of `rows` (initial_inference of obs, recurrent_inference of (hid, act)) against the oracle network.
    Bound: 1e-4 against the oracle's fp32 arithmetic -- or, where that fails (ill-conditioned rows: a near-flat plane
    in front of the min-max scaling, models.py:541-549, amplifies fp32 round-off of ANY implementation), within
    max(1e-4, 32 x the oracle-fp32's own error) of the oracle evaluated in binary64.
    """
    o32 = net_oracle.make_oracle_network(cfg, sd)
    o64 = net_oracle.make_oracle_network(cfg, sd, dtype=torch.float64)
    idx = torch.as_tensor(numpy.asarray(rows), device=obs.device)
    x, h = obs[idx].cpu(), hid[idx].cpu()
    a = act[idx].cpu().long().reshape(-1, 1)
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) - 2)))
    try:
        with torch.no_grad():
            w32 = o32.initial_inference(x) + o32.recurrent_inference(h, a)
            w64 = o64.initial_inference(x.double()) + o64.recurrent_inference(h.double(), a)
    finally:
        torch.set_num_threads(threads)
    names = ("init value", "init reward", "init policy", "init hidden", "rec value", "rec reward", "rec policy", "rec hidden")
    worst = 0.0
    for k, name in enumerate(names):
        got = outs[k][idx].cpu().numpy().reshape(len(rows), -1).astype(numpy.float64)
        ref32 = w32[k].numpy().reshape(len(rows), -1).astype(numpy.float64)
        ref64 = w64[k].numpy().reshape(len(rows), -1)
        if k == 1:
            assert numpy.array_equal(got, ref32)     # log(one-hot) pattern of the root reward (models.py:606-617)
            continue
        err32 = numpy.abs(got - ref32).max(axis=1)
        err64 = numpy.abs(got - ref64).max(axis=1)
        own = numpy.abs(ref32 - ref64).max(axis=1)
        ok = (err32 < TOL) | (err64 <= numpy.maximum(TOL, factor * own))
        worst = max(worst, float(err32[[r not in loose for r in rows]].max(initial=0.0)))
        assert ok.all(), (label, name, [(rows[i], err32[i], err64[i], own[i]) for i in numpy.nonzero(~ok)[0]])
        if k in (3, 7) or not len(loose):
            continue
        # rows the engines disagreed on are the ill-conditioned ones: report how the oracle's own fp32 does there
        li = [i for i, r in enumerate(rows) if r in loose]
        if li:
            print(f"{label}: {name}: {len(li)} loose rows: device-vs-f64 {err64[li].max():.2e}, oracle-fp32-vs-f64 {own[li].max():.2e}")
    print(f"{label}: {len(rows)} rows against the oracle network: worst error of a well-conditioned row {worst:.2e}")


# (the reference constructor's weights at one size per configuration: the launch shapes are those of the synthetic run)
AT_SIZE_WEIGHTS = [(c, "synthetic") for c in sorted(sc.AT_SIZE)] + [(c, "reference") for c in ("gomoku-1024", "connect4-1024", "atari-512")]


@pytest.mark.parametrize("case,weights", AT_SIZE_WEIGHTS)
def test_at_size_operators_and_heads(backend, case, weights):
    game, mode, B = sc.AT_SIZE[case]
    cfg, net, sd = _network(game, mode, seed=41, weights=weights)
    case = case if weights == "synthetic" else case + " (reference weights)"
    if mode in (None, 5):
        assert net.fused_supported() == 0 and net.streamed_supported() == 3
    launched = set()
    for recurrent in (0, 1):
        launches = net.streamed_launches(recurrent, B)
        assert [models.launch_key(l) for l in launches] == [models.launch_key(l) for l in
                                                            sc.inference_launches(backend.lib, game, B, recurrent, mode=mode)]
        launched |= {models.instantiation_key(l) for l in launches}
        print(f"{case} {'recurrent' if recurrent else 'initial'}: {models.summarize_launches(launches)}")
    assert MUST_LAUNCH[case.split(" ")[0]] <= launched, (case, sorted(MUST_LAUNCH[case.split(" ")[0]] - launched))

    obs = _device_rand((B,) + tuple(net.input_shape), 5)
    hid = _device_rand((B,) + tuple(net.hidden_shape), 6)
    act = torch.randint(0, len(cfg.action_space), (B,), generator=torch.Generator().manual_seed(7)).to(torch.int32).cuda()

    # (a) operator by operator, both programs.  Operators in front of the first min-max scaling must agree on EVERY
    # sample; behind it, samples with a near-flat plane may differ between any two fp32 summation orders -- they are
    # collected (at most 3 % of the batch) and up to eight of them held to the binary64 yardstick in (b)
    loose_rows = set()
    for recurrent, x, a in ((0, obs, None), (1, hid, act)):
        n = net.num_operators(recurrent)
        picks = list(range(1, n + 1)) if n <= 32 else sorted(set(range(1, n + 1, 4)) | set(range(n - 12, n + 1)))
        first_scale = _first_scale_operator(net, recurrent)
        worst = 0.0
        for n_ops in picks:
            got = net.debug_prefix(recurrent, 1, n_ops, x, a)
            want = net.debug_prefix(recurrent, 0, n_ops, x, a)
            assert got.shape == want.shape
            scale = 1.0 + float(want.abs().max())
            err = (got - want).abs().amax(dim=1)
            loose = err >= 2e-5 * scale
            bad = torch.nonzero(loose).reshape(-1).cpu().tolist()
            if bad:      # only at / behind a scaling operator, and few (measured: 20 of 9216 connect4 samples -- 64 planes
                # each -- and 19 of 1024 gomoku samples -- 128 planes each -- have such a plane under these random weights)
                assert n_ops >= first_scale, (case, "recurrent" if recurrent else "initial", n_ops, float(err.max()), bad[:8])
                assert len(bad) <= max(2, B // 32), (case, recurrent, n_ops, len(bad), float(err.max()))
                loose_rows.update(bad)
            if len(bad) < B:
                worst = max(worst, float(err[~loose].max()) / scale)
        print(f"{case} {'recurrent' if recurrent else 'initial'}: {len(picks)} of {n} operators, worst relative error "
              f"of a tight operator {worst:.2e}, loose samples so far {len(loose_rows)}")

    # (b) heads against the oracle network: eight rows across the batch (first / last workgroup included) + loose rows
    o = net.initial_inference(obs)
    r = net.recurrent_inference(hid, act)
    rows = sorted(set([0, 1, B // 3, B // 2 - 1, B // 2, (2 * B) // 3, B - 2, B - 1]) | set(sorted(loose_rows)[:8]))
    _rows_against_oracle(cfg, sd, rows, obs, hid, act, list(o) + list(r), case, loose=loose_rows, factor=OWN_ERROR_FACTOR[weights])


def _first_scale_operator(net, recurrent):
    """1-based index of the first per-plane scaling operator of the program (streamed plan kind 1)."""
    for op in range(net.num_operators(recurrent)):
        if net.streamed_plan(recurrent, op)["kind"] == 1:
            return op + 1
    return net.num_operators(recurrent)


def _traces(t, rows, legal):
    """(parent, action) of every simulation of the given trees, from exported canonical-order trees."""
    out = []
    for i in rows:
        tr = []
        for n in range(1, int(t["n_nodes"][i])):
            par = int(t["parent"][i, n])
            slot = int(numpy.nonzero(t["child"][i, par] == n)[0][0])
            tr.append((par, legal[i][slot] if par == 0 else slot))
        out.append(tr)
    return out


TREE_KEYS = ("visit", "value_sum", "reward", "prior", "child", "parent", "to_play", "minmax", "n_nodes")


AT_SIZE_SEARCH_WEIGHTS = ([(c, "synthetic") for c in sorted(sc.AT_SIZE_SEARCHES)] +
                          [(c, "reference") for c in ("gomoku-1024", "connect4-1024", "atari-256")])


@pytest.mark.parametrize("case,weights", AT_SIZE_SEARCH_WEIGHTS)
def test_at_size_search_two_streams_and_oracle(backend, case, weights, monkeypatch):
    game, mode, B, sims, n_sample = sc.AT_SIZE_SEARCHES[case]
    cfg, net, sd = _network(game, mode, seed=9, weights=weights)
    case = case if weights == "synthetic" else case + " (reference weights)"
    if sims is not None:
        cfg.num_simulations = sims
    S, A = cfg.num_simulations, len(cfg.action_space)
    launches, parts = sc.search_launches(backend.lib, game, B, mode=mode)
    print(f"{case}: half-shards {parts}; launches {sc.summarize(launches)}")
    assert MUST_LAUNCH[f"{game}-{parts[0]}"] & {models.instantiation_key(l) for l in launches if l["program"] == "recurrent"}

    obs = _device_rand((B,) + tuple(net.input_shape), 4)
    rs = numpy.random.RandomState(2)
    legal = [sorted(rs.choice(A, size=rs.randint(2, A + 1), replace=False).tolist()) for _ in range(B)]
    to_play = [int(i % len(cfg.players)) for i in range(B)]
    seeds = [3000 + i for i in range(B)]
    engine = self_play.BatchedMCTS(cfg, net, B)
    res = engine.run(obs, legal, to_play, True, [numpy.random.RandomState(s) for s in seeds])
    kernel = engine.kernel_name(B)
    assert "rb_gemm_kernel" in kernel and "row_select_kernel" in kernel
    assert ("two half-shards" in kernel) == (parts[1] > 0), kernel
    parity._tree_invariants(cfg, res, S)
    for i in range(0, B, max(1, B // 256)):
        assert set(numpy.nonzero(res.visit_counts[i])[0]).issubset(set(legal[i]))
    trees = engine.export_trees(B)

    # (d) the undivided shard: every statistic of every tree bit for bit
    if parts[1] > 0:
        backend.lib.tuning_set("row_split_min", 0)
        res1 = engine.run(obs, legal, to_play, True, [numpy.random.RandomState(s) for s in seeds])
        assert "two half-shards" not in engine.kernel_name(B)
        backend.lib.tuning_set("row_split_min", 1024)
        trees1 = engine.export_trees(B)
        assert numpy.array_equal(res.visit_counts, res1.visit_counts)
        assert numpy.array_equal(res.root_values.view(numpy.int64), res1.root_values.view(numpy.int64))
        assert numpy.array_equal(res.max_tree_depth, res1.max_tree_depth) and numpy.array_equal(res.tape_used, res1.tape_used)
        for key in TREE_KEYS:
            a, b = trees[key], trees1[key]
            if a.dtype == numpy.float64:
                a, b = a.view(numpy.int64), b.view(numpy.int64)
            assert numpy.array_equal(a, b), (case, key)
        del trees1

    # (c) sampled trees against the CPU oracle, simulation by simulation; the same trees on the oracle in binary64
    sample = sorted({min(B - 1, (k * B) // n_sample + k % 5) for k in range(n_sample)})   # (not all at one offset of a workgroup)
    got = _traces(trees, sample, legal)
    jobs = [(obs[i].cpu().numpy(), legal[i], to_play[i], seeds[i]) for i in sample]
    procs = max(1, min(len(jobs), (os.cpu_count() or 2) - 2, 32))
    s32 = parallel.run_searches(cfg, sd, jobs, processes=procs)
    s64 = parallel.run_searches(cfg, sd, jobs, processes=procs, dtype_name="float64")
    identical = own_identical = 0
    failures = []
    first_diff = lambda a, b: next((k for k in range(max(len(a), len(b))) if k >= len(a) or k >= len(b) or a[k] != b[k]), None)
    for i, g, t32, t64 in zip(sample, got, s32, s64):
        want = t32["trace"]
        k64 = first_diff(t64["trace"], want)          # where the oracle's own fp32 search leaves its binary64 evaluation
        own_identical += int(k64 is None)
        k = first_diff(g, want)
        if k is not None:
            gap, depth = t32["margins"][k]
            print(f"{case}: tree {i} diverges at simulation {k} of {len(want)} (oracle {want[k]}, device "
                  f"{g[k] if k < len(g) else None}); oracle UCB top-2 margin on that walk {gap:.3e} at depth {depth}; the "
                  f"oracle's own fp32 search leaves its binary64 evaluation at simulation {k64}")
            # A divergence must be a near-tie of the oracle's UCB scores (MARGIN_GATE) -- unless this is a tree on which
            # the oracle's OWN fp32 arithmetic cannot hold the exact line either and the device leaves the fp32 oracle
            # in the same stretch of the search (not before half of the oracle's own distance).  Deep single-line
            # searches compound the round-off of every recurrent_inference on the path: the 400 simulations of
            # games/gomoku.py dig 90- to 120-ply lines with these weights, NONE of 16 sampled trees keeps the oracle's
            # fp32 and binary64 searches together (they part at simulations 54 ... 262, the device parts from the fp32
            # oracle at 77 ... 251, often at the very same simulation, profiles/r04_pytest_gpu_*.log), and the margin at
            # such a node (4e-3 at depth 94) measures the compounded error of BOTH fp32 evaluations, not a defect
            # With the reference constructor's weights (the weights bench.py times) there is NO such waiver: every divergence
            # must be a near-tie.  (Their searches go even deeper -- 110 to 400 plies for games/gomoku.py -- and the oracle's
            # fp32 and binary64 searches part on every tree as well, but at margins of 1e-6 ... 1.2e-4: CPU probe, round 5.)
            waived = weights == "synthetic" and k64 is not None and 2 * k >= k64
            if not parity.near_tie(gap, t32["value_margins"][k]) and not waived:
                failures.append((i, k, gap, depth, k64))
            continue
        identical += 1
        assert t32["root_visit_counts"] == list(res.visit_counts[i]), (case, i)
        assert res.max_tree_depth[i] == t32["max_depth"], (case, i)
        # root value of a tree that agrees in every simulation.  The yardstick is the tree itself: when the oracle's fp32 and
        # binary64 searches agree on it too, the device must be within 8 x the oracle-fp32's own error of the binary64
        # value (floor 1e-3: the decode cancels ~3 digits); otherwise within 3e-3 of the fp32 oracle
        rv32 = t32["root_value"]
        tol, ref = 30 * TOL, rv32
        if k64 is None:
            ref = t64["root_value"]
            tol = max(10 * TOL, 8 * abs(rv32 - ref))
        if not abs(res.root_values[i] - ref) < tol * max(1.0, abs(ref)):
            failures.append((i, "root value", res.root_values[i], rv32, ref))
    assert not failures, (case, failures)
    # the visit statistics the replay buffer consumes (root child_visits, root value) on ALL sampled trees, diverged ones
    # included, against absolute per-case bounds (tests/at_size.py; round 5's bound was relative to the oracle's own
    # fp32-vs-binary64 divergence count, which is total for games/gomoku.py and could not fail)
    at_size.gate(case, at_size.statistics(S, [res.visit_counts[i] for i in sample], [res.root_values[i] for i in sample],
                                          s32, s64, identical))
