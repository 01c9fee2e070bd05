"""
GPU parity tests of rt_search_kernel (csrc/mzx_tower_search.inc): every simulation of MCTS.run
(/root/reference/self_play.py:319-355) in ONE launch for wide residual networks, the trunks as towers inside.

The kernel's claim is that it builds, bit for bit, the trees of the per-simulation launches of the streamed engine
(row_select_kernel -> rb_tower_kernel x 2 -> grouped head levels -> row_expand_backprop_kernel, csrc/mzx_row_search.h),
which tests/test_gpu_streamed.py ties to the one-thread-per-tree generic operators and tests/test_gpu_streamed_at_size.py /
tests/test_gpu_parity.py to the CPU oracle.  Here: every statistic of the finished trees against that route (small
ragged shards with forced trees per workgroup -> every <MT, 1, AW> instantiation; BASELINE config C4 at its 1024-tree
shard with 200 simulations), the routing (host-side planner through the C ABI), and C4 at size against the oracle.
"""
import math
import os

import numpy
import pytest

from mzx import _lib, configs, models, self_play, synthetic
from oracle import parallel

import at_size
import test_gpu_parity as parity

pytestmark = pytest.mark.gpu

TREE_KEYS = ("visit", "value_sum", "reward", "prior", "child", "parent", "to_play", "minmax", "n_nodes")


@pytest.fixture(scope="module")
def backend():
    return _lib.default_backend()


CASES = {
    # games/connect4.py as shipped: 6 x 7, 7 actions (16-lane child records), 64 channels x 3 blocks per trunk
    "connect4": lambda: configs.connect4(),
    # a 4 x 4 board: ONE row tile per tree (<1, 1>), 4 actions (4-lane records), one player, 48 channels (three column tiles:
    # a wave grid with an idle wave), an 81-bin support (wide decode is NOT taken: 4 actions, but F > 32 -> AW = 0)
    "board4x4": lambda: configs.connect4(observation_shape=(3, 4, 4), action_space=list(range(4)), players=list(range(1)),
                                         channels=48, blocks=1, support_size=40, discount=0.997),
    # 4 actions and a narrow support: AW = 4
    "narrow4": lambda: configs.connect4(observation_shape=(3, 4, 4), action_space=list(range(4)), channels=64, blocks=2),
    # 32 actions on 4 x 8: several child slots per lane (AW = 0), head inputs that are no multiple of 16, head chains of
    # one and three Linear layers
    # games/gomoku.py as shipped (128 channels x 6 blocks, 11 x 11, 121 actions): one board per 512-thread workgroup, <8,1>,
    # eight column tiles; the library does not route it here by itself (launch by launch is faster), "rt_search" = 1 does
    "gomoku": lambda: configs.gomoku(),
    "wide32": lambda: configs.connect4(observation_shape=(3, 4, 8), action_space=list(range(32)), channels=64, blocks=1,
                                       reduced_channels_reward=3, reduced_channels_value=5, reduced_channels_policy=7,
                                       resnet_fc_reward_layers=[24], resnet_fc_value_layers=[40, 16], resnet_fc_policy_layers=[]),
}


def _route(backend, engine, B):
    out = (backend.lib.mzx_search_route.argtypes[1]._type_)()
    backend.lib.check(backend.lib.mzx_search_route(engine.handle(B), out))
    return list(out)


def _inputs(cfg, net, B, seed):
    A = len(cfg.action_space)
    obs = synthetic.observations(B, net.input_shape, seed=seed)
    rs = numpy.random.RandomState(seed + 1)
    legal = [sorted(rs.choice(A, size=rs.randint(1, A + 1), replace=False).tolist()) for _ in range(B)]
    to_play = [int(i % len(cfg.players)) for i in range(B)]
    return obs, legal, to_play


def _assert_same(a, b, label):
    (r0, t0), (r1, t1) = a, b
    assert numpy.array_equal(r0.visit_counts, r1.visit_counts), label
    assert numpy.array_equal(r0.root_values.view(numpy.int64), r1.root_values.view(numpy.int64)), label
    assert numpy.array_equal(r0.root_predicted_values.view(numpy.int64), r1.root_predicted_values.view(numpy.int64)), label
    assert numpy.array_equal(r0.max_tree_depth, r1.max_tree_depth) and numpy.array_equal(r0.tape_used, r1.tape_used), label
    for key in TREE_KEYS:
        x, y = t0[key], t1[key]
        if x.dtype == numpy.float64:
            x, y = x.view(numpy.int64), y.view(numpy.int64)
        assert numpy.array_equal(x, y), (label, key)


@pytest.mark.parametrize("name,B,S,trees_per_wg,waves", [
    ("connect4", 19, 40, 0, 0), ("connect4", 50, 30, 1, 8), ("connect4", 51, 30, 2, 8), ("connect4", 131, 24, 3, 8),
    ("connect4", 37, 30, 1, 4), ("connect4", 21, 30, 2, 4), ("connect4", 133, 24, 4, 8), ("connect4", 50, 24, 5, 8),
    ("connect4", 67, 20, 6, 8), ("connect4", 40, 20, 3, 4),
    ("board4x4", 37, 30, 0, 0), ("board4x4", 9, 30, 5, 8), ("board4x4", 50, 24, 12, 8), ("narrow4", 23, 30, 0, 0),
    ("narrow4", 10, 30, 3, 8), ("narrow4", 29, 20, 7, 4), ("wide32", 21, 36, 0, 0), ("wide32", 13, 20, 1, 8), ("wide32", 19, 20, 3, 4),
    ("gomoku", 6, 12, 0, 0)])
def test_tower_search_kernel_bit_identical_to_per_simulation_launches(backend, name, B, S, trees_per_wg, waves):
    """
    rt_search_kernel against the launch-by-launch route of the SAME engine (tuning "rt_search" = 0) on ragged shards (a
    last workgroup with missing trees, partial wavefronts of rows, ragged legal sets, both players): every statistic of
    every finished tree bit for bit, whatever the trees per workgroup and the workgroup size (512 / 256 threads) -- row tiles
    per wave 1 .. 8, i.e. the 128-register instantiations with four waves per SIMD and the deep ones with two -- and a second
    run on the same handle too.
    """
    cfg = CASES[name]()
    cfg.num_simulations = S
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 12))
    obs, legal, to_play = _inputs(cfg, net, B, 6)
    outs = []
    for rt in (0, 1, 1):
        backend.lib.tuning_set("rt_search", rt)
        backend.lib.tuning_set("rt_trees", trees_per_wg if rt else 0)
        backend.lib.tuning_set("rt_waves", waves if rt else 0)
        backend.lib.tuning_set("row_split_min", 0 if len(outs) == 0 else 32)
        engine = self_play.BatchedMCTS(cfg, net, B, mode=1) if len(outs) < 2 else engine
        route = _route(backend, engine, B)
        assert route[0] == (3 if rt else 2), (name, route)
        if rt and trees_per_wg:
            assert route[1] == trees_per_wg, route
        res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(500 + i) for i in range(B)])
        kernel = engine.kernel_name(B)
        assert (kernel == "mzx::rt_search_kernel") == bool(rt), kernel
        if not rt:
            assert "rb_tower_kernel" in kernel and "row_select_kernel" in kernel
        outs.append((res, engine.export_trees(B)))
        if rt:
            if waves:
                assert route[6] == waves * 64, route
            print(f"{name}: {B} trees, rt_search_kernel with {route[1]} trees per workgroup of {route[6]} threads, <{route[2]},1>, "
                  f"{route[3]} workgroups, {route[4]} per CU, {route[5]} bytes of LDS")
    assert (outs[0][0].visit_counts.sum(1) == S).all()
    _assert_same(outs[0], outs[1], (name, "rt vs launches"))
    _assert_same(outs[1], outs[2], (name, "second run"))


@pytest.mark.parametrize("name,B,S,trees_per_wg,waves", [("connect4", 133, 24, 4, 8), ("connect4", 50, 30, 1, 8), ("narrow4", 29, 20, 7, 8)])
def test_short_row_group_same_trees(backend, name, B, S, trees_per_wg, waves):
    """
    A workgroup whose row tiles do not divide evenly among its row groups (four connect4 boards: eleven tiles dealt 6 + 5;
    one board: three tiles dealt 2 + 1): the waves of the short group run a K loop over MT - 1 tiles (the <MT, 1, AW, true>
    instantiations, tuning "rt_short" = 1, the default) instead of multiplying a tile that does not exist.  Same trees, bit
    for bit, as with "rt_short" = 0.
    """
    cfg = CASES[name]()
    cfg.num_simulations = S
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 13))
    obs, legal, to_play = _inputs(cfg, net, B, 8)
    outs = []
    for short in (1, 0):
        with backend.lib.tuning(rt_search=1, rt_trees=trees_per_wg, rt_waves=waves, rt_short=short):
            engine = self_play.BatchedMCTS(cfg, net, B, mode=1)
            route = _route(backend, engine, B)
            assert route[0] == 3 and route[1] == trees_per_wg, route
            res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(700 + i) for i in range(B)])
            assert engine.kernel_name(B) == "mzx::rt_search_kernel"
            outs.append((res, engine.export_trees(B)))
    assert (outs[0][0].visit_counts.sum(1) == S).all()
    _assert_same(outs[0], outs[1], (name, "short row group vs every wave multiplying MT tiles"))


def test_tower_search_routing(backend):
    """Which searches the library sends to rt_search_kernel (mzx_search_route, host-side): connect4 at every shard size --
    the same arithmetic whatever the shard, csrc/mzx_row_search.h wide_search_route --, never a narrow network, a network
    whose towers are too wide for the kernel's tilings, or a fully connected one."""
    for B, want in ((8, (1, 256)), (512, (1, 256)), (768, (3, 512)), (1024, (4, 512)), (1536, (6, 512)), (2048, (4, 512))):
        cfg = configs.connect4()
        net = models.MuZeroNetwork(cfg)
        net.set_weights(synthetic.fill_state_dict(net.state_dict(), 1))
        engine = self_play.BatchedMCTS(cfg, net, B)
        route = _route(backend, engine, B)
        assert route[0] == 3, (B, route)
        assert route[3] == -(-B // route[1]) and route[5] <= 160 * 1024 and route[2] <= 8
        # whole rounds of few, fat workgroups: four boards = 168 rows in eleven row tiles <6,1> at 1024 trees; at 1536 six
        # boards per 512-thread workgroup and three per 256-thread one (two per CU) cost the same, both <8,1>
        assert (route[1], route[6]) == want or (B == 1536 and (route[1], route[6]) == (3, 256)), (B, route)
        with backend.lib.tuning(rt_search=0):
            r2 = _route(backend, engine, B)
            assert r2[0] == 2 and r2[6] + r2[7] == B and (r2[7] > 0) == (B >= 1024), (B, r2)
        with backend.lib.tuning(rt_trees=2, rt_waves=8):
            r3 = _route(backend, engine, B)
            assert r3[:3] == [3, 2, 3] and r3[6] == 512, (B, r3)
        with backend.lib.tuning(wide_towers=0):
            assert _route(backend, engine, B)[0] == 1
    for name, want in (("tictactoe", 1), ("breakout", 1), ("gomoku", 2), ("cartpole", 4)):
        cfg = configs.BY_NAME[name]()
        net = models.MuZeroNetwork(cfg)
        net.set_weights(synthetic.fill_state_dict(net.state_dict(), 1))
        engine = self_play.BatchedMCTS(cfg, net, 64)
        assert _route(backend, engine, 64)[0] == want, name


def test_tower_search_at_size_same_trees_as_launches(backend):
    """BASELINE config C4 at its shard (1024 trees x 200 simulations, games/connect4.py as shipped): rt_search_kernel,
    the two half-shards on two streams and the undivided per-simulation launches build the same 1024 trees, bit for
    bit -- the instantiation and launch shape bench.py times (`c4`)."""
    cfg = configs.connect4()
    B, S = 1024, cfg.num_simulations
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 9))
    obs, legal, to_play = _inputs(cfg, net, B, 4)
    seeds = [3000 + i for i in range(B)]
    outs = {}
    for label, tuning in (("rt", {}), ("two streams", {"rt_search": 0}), ("undivided", {"rt_search": 0, "row_split_min": 0})):
        with backend.lib.tuning(**tuning):
            engine = self_play.BatchedMCTS(cfg, net, B)
            res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(s) for s in seeds])
            kernel = engine.kernel_name(B)
            assert (kernel == "mzx::rt_search_kernel") == (label == "rt"), (label, kernel)
            assert ("two half-shards" in kernel) == (label == "two streams"), (label, kernel)
            outs[label] = (res, engine.export_trees(B))
    parity._tree_invariants(cfg, outs["rt"][0], S)
    _assert_same(outs["rt"], outs["two streams"], "rt vs two half-shards")
    _assert_same(outs["rt"], outs["undivided"], "rt vs undivided launches")


@pytest.mark.parametrize("B,trees_per_wg,n_sample", [(512, 1, 32), (1536, 6, 48), (9216, 6, 48)])
def test_tower_search_other_shards_against_oracle(backend, B, trees_per_wg, n_sample):
    """The other tilings the planner picks at full size (one tree per 256-thread workgroup = <3,1> at 512 trees, three = <8,1>
    at 1536; four per 512-thread workgroup = <6,1> at 1024 trees: test_full_size_residual_configs[connect4]) against the CPU
    oracle, simulation by simulation, with the oracle's own fp32-vs-binary64 divergence as the yardstick."""
    cfg = configs.connect4()
    S = cfg.num_simulations
    net = models.MuZeroNetwork(cfg)
    sd = synthetic.fill_state_dict(net.state_dict(), 9)
    net.set_weights(sd)
    obs, legal, to_play = _inputs(cfg, net, B, 4)
    legal = [l if len(l) > 1 else sorted(set(l + [(l[0] + 1) % 7])) for l in legal]
    seeds = [3000 + i for i in range(B)]
    engine = self_play.BatchedMCTS(cfg, net, B)
    route = _route(backend, engine, B)
    assert route[0] == 3 and (route[1] == trees_per_wg or (trees_per_wg == 6 and route[1] == 3)), route
    res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(s) for s in seeds])
    assert engine.kernel_name(B) == "mzx::rt_search_kernel"
    parity._tree_invariants(cfg, res, S)
    trees = engine.export_trees(B)
    sample = sorted({min(B - 1, (k * B) // n_sample + k % 5) for k in range(n_sample)})
    got = []
    for i in sample:
        tr = []
        for n in range(1, int(trees["n_nodes"][i])):
            par = int(trees["parent"][i, n])
            slot = int(numpy.nonzero(trees["child"][i, par] == n)[0][0])
            tr.append((par, legal[i][slot] if par == 0 else slot))
        got.append(tr)
    jobs = [(obs[i], legal[i], to_play[i], seeds[i]) for i in sample]
    procs = max(1, min(len(jobs), (os.cpu_count() or 2) - 2, 32))
    s32 = parallel.run_searches(cfg, sd, jobs, processes=procs)
    s64 = parallel.run_searches(cfg, sd, jobs, processes=procs, dtype_name="float64")
    first_diff = lambda a, b: next((k for k in range(max(len(a), len(b))) if k >= len(a) or k >= len(b) or a[k] != b[k]), None)
    identical = own = 0
    for i, g, t32, t64 in zip(sample, got, s32, s64):
        own += int(first_diff(t64["trace"], t32["trace"]) is None)
        k = first_diff(g, t32["trace"])
        if k is None:
            identical += 1
            assert t32["root_visit_counts"] == list(res.visit_counts[i]), i
            continue
        gap, depth = t32["margins"][k]
        print(f"connect4 x {B}: tree {i} diverges at simulation {k}; oracle UCB top-2 margin {gap:.3e} at depth {depth}")
        assert parity.near_tie(gap, t32["value_margins"][k]), (i, k, gap)
    # the visit statistics the replay buffer consumes, on all sampled trees, against absolute bounds (tests/at_size.py)
    at_size.gate(f"connect4 x {B} on rt_search_kernel", at_size.statistics(
        S, [res.visit_counts[i] for i in sample], [res.root_values[i] for i in sample], s32, s64, identical))
