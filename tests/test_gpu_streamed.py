"""
GPU parity of the STREAMED MFMA engine (csrc/mzx_batched.hip, -m gpu), through the C ABI:
  * the reference's two large residual configurations AS SHIPPED (games/gomoku.py, games/atari.py) against
    fixtures the unmodified reference produced (oracle/make_golden.py --large-residual): heads <= 1e-4, whole
    searches with identical visit counts;
  * every operator of both programs against the one-element-kernel-per-operator engine (mode 0), on shapes that
    exercise whole-sample tiles, patch tiles with halos, channel phases, strides, ragged channel counts;
  * the reference's SMALL configurations forced onto this engine (mode 3) against their reference fixtures, so
    that the same goldens pin both MFMA engines.
"""
import json
import os

import numpy
import pytest
import torch

import lockstep
import test_hostcheck_search as common
from conftest import GOLDEN
from mzx import _lib, configs, models, self_play, synthetic
from oracle import mcts_oracle, net_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star


@pytest.fixture(scope="module")
def backend():
    return _lib.default_backend()


@pytest.fixture(autouse=True)
def _launch_by_launch(backend):
    """This file tests the streamed engine's LAUNCHES (towers, layers, grouped head levels between the row-per-tree
    kernels); searches the library would run as one launch of rt_search_kernel (tests/test_gpu_tower_search.py) stay
    on the per-simulation launches here.  conftest restores the default afterwards."""
    backend.lib.tuning_set("rt_search", 0)


def _fixture_obs(z):
    if "obs" in z.files:
        return z["obs"]
    return synthetic.observations(int(z["obs_shape"][0]), tuple(int(v) for v in z["obs_shape"][1:]), seed=int(z["obs_seed"]))


@pytest.mark.parametrize("name,mode", [("resnet_gomoku", 1), ("resnet_atari", 1), ("resnet_tictactoe", 3),
                                       ("resnet_connect4", 3), ("resnet_breakout", 3), ("resnet_breakout_cnn", 3),
                                       ("resnet_cnn_small", 3)])
def test_streamed_heads_within_tolerance(backend, name, mode):
    z = numpy.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    meta = json.loads(str(z["meta"]))
    cfg = configs.BY_NAME[meta["game"]](**meta["overrides"])
    net = models.MuZeroNetwork(cfg)
    if mode == 1:
        assert net.fused_supported() == 0 and net.streamed_supported() == 3   # as shipped: too large for the LDS engine
    net.set_mode(mode)
    assert [k for k, _, _ in meta["keys"]] == list(net.state_dict().keys())
    sd = synthetic.fill_state_dict(net.state_dict(), meta["weight_seed"])
    net.set_weights(sd)
    obs = _fixture_obs(z)
    o = net.initial_inference(torch.tensor(obs))
    r1 = net.recurrent_inference(o[3], torch.tensor(z["act1"]))
    r2 = net.recurrent_inference(r1[3], torch.tensor(z["act2"]))
    o1 = net.initial_inference(torch.tensor(obs[:1]))
    for tag, res in (("init", o), ("rec1", r1), ("rec2", r2), ("init_b1", o1)):
        for key, t in zip(("value", "reward", "policy", "hidden"), res):
            ref, got = z[f"{tag}_{key}"], t.cpu().numpy()
            assert got.shape == ref.shape
            if key == "reward" and tag.startswith("init"):
                assert numpy.array_equal(got, ref)
            else:
                err = numpy.abs(got - ref).max()
                print(f"{name} {tag} {key}: max abs error {err:.2e}")
                assert err < TOL, (name, tag, key, err)
        vs = models.support_to_scalar(res[0], cfg.support_size).cpu().numpy()
        assert numpy.allclose(vs, z[f"{tag}_value_scalar"], atol=3 * TOL, rtol=3 * TOL)


STREAMED_CASES = {
    # as shipped (whole-sample tiles, one phase; NCHW gather of the hidden state, action plane)
    "gomoku": (lambda: configs.gomoku(), 5),
    # 64 channels on 19 x 19: patch tiles with halos crossing tile borders, ragged last tiles
    "go19": (lambda: configs.connect4(observation_shape=(3, 19, 19), action_space=list(range(361))), 3),
    # channel counts that are no multiple of 16 / 4, stacked observations, non-square board, wide heads
    "odd": (lambda: configs.gomoku(channels=70, observation_shape=(5, 13, 9), action_space=list(range(117)),
                                   stacked_observations=2, reduced_channels_value=33, reduced_channels_policy=48,
                                   resnet_fc_value_layers=[200, 77], support_size=40, blocks=2), 7),
    # the atari architecture at reduced width: stride-2 stem convolutions, pooling, channel phases at 6 x 6
    "atari_narrow": (lambda: configs.atari(channels=96, blocks=2, stacked_observations=3, reduced_channels_reward=40,
                                           reduced_channels_value=40, reduced_channels_policy=24,
                                           resnet_fc_reward_layers=[72], resnet_fc_value_layers=[72, 40],
                                           resnet_fc_policy_layers=[56], support_size=50), 3),
    # small configurations that normally run on the LDS engine
    "tictactoe": (lambda: configs.tictactoe(), 37),
    "connect4": (lambda: configs.connect4(), 37),
    "breakout": (lambda: configs.breakout(), 3),
    # DownsampleCNN stems (models.py:278-297): K x K stride-4 and 5 x 5 convolutions with bias, max / adaptive pooling
    "breakout_cnn": (lambda: configs.breakout(downsample="CNN"), 3),
    "cnn_small": (lambda: configs.breakout(downsample="CNN", observation_shape=(2, 40, 56), stacked_observations=1,
                                           channels=8, blocks=1), 5),
    # nine column tiles: two per wave (NT = 2) in the layer kernel and in the tower kernel
    "wide144": (lambda: configs.connect4(channels=144, blocks=2), 5),
}


@pytest.mark.parametrize("name", sorted(STREAMED_CASES))
def test_streamed_operator_by_operator(backend, name):
    """Bisection harness: the output tensor of EVERY operator of both programs, streamed engine vs element kernels."""
    make, B = STREAMED_CASES[name]
    cfg = make()
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 31))
    net.set_mode(3)
    rs = numpy.random.RandomState(5)
    obs = torch.tensor(rs.rand(B, *net.input_shape).astype(numpy.float32))
    hid = torch.tensor(rs.rand(B, *net.hidden_shape).astype(numpy.float32))
    act = torch.tensor(rs.randint(0, len(cfg.action_space), size=B).astype(numpy.int32))
    for recurrent, x, a in ((0, obs, None), (1, hid, act)):
        worst = 0.0
        for n_ops in range(1, net.num_operators(recurrent) + 1):
            got = net.debug_prefix(recurrent, 1, n_ops, x, a).cpu().numpy()
            want = net.debug_prefix(recurrent, 0, n_ops, x, a).cpu().numpy()
            err = numpy.abs(got - want).max()
            scale = 1.0 + numpy.abs(want).max()
            worst = max(worst, err / scale)
            assert err < 2e-5 * scale, (name, "recurrent" if recurrent else "initial", n_ops, err)
        print(f"{name} {'recurrent' if recurrent else 'initial'}: worst relative operator error {worst:.2e}")


@pytest.mark.parametrize("name,B,T", [("connect4", 50, 1), ("connect4", 50, 2), ("connect4", 50, 3), ("connect4", 47, 4),
                                      ("connect4", 50, 6), ("tictactoe", 37, 1), ("tictactoe", 200, 29), ("tictactoe", 300, 64),
                                      ("gomoku", 3, 1), ("odd", 5, 1), ("atari_narrow", 7, 1), ("atari_narrow", 7, 2),
                                      ("atari_narrow", 9, 4), ("breakout", 9, 3), ("wide144", 5, 1), ("wide144", 7, 2)])
def test_tower_kernel_layer_by_layer(backend, name, B, T, monkeypatch):
    """
    rb_tower_kernel (a whole trunk -- conv + residual blocks -- in one launch, activations in LDS in place, the block
    input kept in registers as the residual) against the element kernels, LAYER BY LAYER: a prefix that ends inside a
    tower runs the tower's first layers only, so every layer's output is compared.  Forced samples per workgroup T
    (tuning "rb_tower_t") walk the instantiations <MT, NT> -- one to nine row tiles per wave, one and two column tiles, the
    in-place and the two-set K loops -- with batches that do not fill the last workgroup.
    """
    make, _ = STREAMED_CASES[name]
    cfg = make()
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 33))
    net.set_mode(3)
    backend.lib.tuning_set("rb_tower_t", T)
    rs = numpy.random.RandomState(7)
    obs = torch.tensor(rs.rand(B, *net.input_shape).astype(numpy.float32))
    hid = torch.tensor(rs.rand(B, *net.hidden_shape).astype(numpy.float32))
    act = torch.tensor(rs.randint(0, len(cfg.action_space), size=B).astype(numpy.int32))
    seen = set()
    for recurrent, x, a in ((0, obs, None), (1, hid, act)):
        towers = [l for l in net.streamed_launches(recurrent, B) if l["k_loop"].startswith("tower")]
        assert towers, (name, recurrent)
        in_tower = set()
        for t in towers:
            assert t["T"] <= T
            seen.add((t["MT"], t["NT"], t["T"]))
            # (a tower entry: in_layout = -layers, nsplit - 1 = operators of its tail, which run inside the launch too)
            in_tower.update(range(t["op"], t["op"] - t["in_layout"] + t["nsplit"] - 1))
        worst = 0.0
        for n_ops in range(1, net.num_operators(recurrent) + 1):
            if n_ops - 1 not in in_tower and n_ops % 3:
                continue          # operators outside towers: test_streamed_operator_by_operator covers them; sample a third
            got = net.debug_prefix(recurrent, 1, n_ops, x, a).cpu().numpy()
            want = net.debug_prefix(recurrent, 0, n_ops, x, a).cpu().numpy()
            err = numpy.abs(got - want).max()
            scale = 1.0 + numpy.abs(want).max()
            worst = max(worst, err / scale)
            assert err < 2e-5 * scale, (name, T, "recurrent" if recurrent else "initial", n_ops, err)
        print(f"{name} T={T} {'recurrent' if recurrent else 'initial'}: towers {[(t['op'], -t['in_layout'], t['MT'], t['NT'], t['T']) for t in towers]}, "
              f"worst relative error {worst:.2e}")
    # the tail (per-plane scaling, small 1x1 head convolutions on the LDS-resident output) against the same operators
    # launched on their own: the scaled hidden state bit for bit (same arithmetic on the same values), everything else tightly
    o1, r1 = net.initial_inference(obs), net.recurrent_inference(hid, act)
    backend.lib.tuning_set("rb_tail", 0)
    assert all(l["nsplit"] == 1 for l in net.streamed_launches(1, B) if l["k_loop"].startswith("tower"))
    o2, r2 = net.initial_inference(obs), net.recurrent_inference(hid, act)
    backend.lib.tuning_set("rb_tail", 1)
    assert any(l["nsplit"] > 1 for l in net.streamed_launches(1, B) if l["k_loop"].startswith("tower")), "no tower has a tail"
    assert torch.equal(o1[3], o2[3]) and torch.equal(r1[3], r2[3])
    for got, want in zip(o1 + r1, o2 + r2):
        g, w = got.cpu().numpy(), want.cpu().numpy()
        fin = numpy.isfinite(w)
        assert numpy.array_equal(numpy.isfinite(g), fin) and numpy.abs(numpy.where(fin, g - w, 0.0)).max() < 1e-5 * (1.0 + numpy.abs(w[fin]).max())
    # grouped head launches (tuning "rb_heads" = 2, the default: the k-th Linear layers of all chains as blockIdx.z slices
    # of ONE rb_gemm_multi_kernel launch, inputs and inner outputs in the private region) against one launch per layer
    # ("rb_heads" = 0) -- the same kernel body on the same shapes: the same bits
    assert backend.lib.tuning_get("rb_heads") == 2
    if name in ("connect4", "gomoku", "breakout"):
        assert any("grouped" in l["k_loop"] for l in net.streamed_launches(1, B))
    backend.lib.tuning_set("rb_heads", 0)
    assert not any("grouped" in l["k_loop"] for l in net.streamed_launches(1, B))
    o5, r5 = net.initial_inference(obs), net.recurrent_inference(hid, act)
    backend.lib.tuning_set("rb_heads", 2)
    for got, want in zip(o1 + r1, o5 + r5):
        assert torch.equal(got, want), (name, T, "grouped head launches")
    # and the whole inferences (heads behind the towers) against the layer-by-layer streamed path
    net.set_mode(4)
    o0, r0 = net.initial_inference(obs), net.recurrent_inference(hid, act)
    for got, want in zip(o1 + r1, o0 + r0):
        g, w = got.cpu().numpy(), want.cpu().numpy()
        fin = numpy.isfinite(w)
        assert numpy.array_equal(numpy.isfinite(g), fin)
        # (a near-flat plane in front of the min-max scaling amplifies the two summation orders: a few samples may differ)
        err = numpy.abs(numpy.where(fin, g - w, 0.0)).reshape(B, -1).max(axis=1)
        assert (err < 5e-5 * (1.0 + numpy.abs(w[fin]).max())).sum() >= B - max(1, B // 16), (name, T, err.max())


def test_atari_as_shipped_operator_by_operator(backend):
    """games/atari.py:61-69 unchanged (256 channels x 16 blocks, 73.5 M parameters): every 5th operator + the heads."""
    cfg = configs.atari()
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 29))
    assert net.streamed_supported() == 3
    B = 2
    rs = numpy.random.RandomState(6)
    obs = torch.tensor(rs.rand(B, *net.input_shape).astype(numpy.float32))
    hid = torch.tensor(rs.rand(B, *net.hidden_shape).astype(numpy.float32))
    act = torch.tensor(rs.randint(0, 4, size=B).astype(numpy.int32))
    for recurrent, x, a in ((0, obs, None), (1, hid, act)):
        n = net.num_operators(recurrent)
        picks = sorted(set(list(range(1, n + 1, 5)) + list(range(n - 12, n + 1))))
        for n_ops in picks:
            got = net.debug_prefix(recurrent, 1, n_ops, x, a).cpu().numpy()
            want = net.debug_prefix(recurrent, 0, n_ops, x, a).cpu().numpy()
            err = numpy.abs(got - want).max()
            assert err < 2e-5 * (1.0 + numpy.abs(want).max()), ("atari", recurrent, n_ops, err)


@pytest.mark.parametrize("name", ["gomoku", "atari"])
def test_streamed_search_matches_reference(backend, name):
    """
    MCTS.run of the unmodified reference on the shipped architecture (tree_<name>.npz: eight gomoku trees x 48
    simulations, four atari trees x the 50 simulations of games/atari.py:42), simulation by simulation: the (parent,
    action) of every expansion against the reference's own trace, identical visit counts / depth / root value for trees
    that agree throughout.  A tree that leaves the reference's trace must do so at a near-tie of the UCB scores (the
    margin comes from the CPU oracle, which reproduces the reference's trace on these fixtures bit for bit:
    tests/test_oracle_golden.py) -- and at most a quarter of the trees may.
    """
    z, meta, cfg = lockstep.load_fixture(name)
    cfg.num_simulations = meta["num_simulations"]
    net = common.build_model(backend, cfg, meta["weight_seed"])
    sd = synthetic.fill_state_dict(net.state_dict(), meta["weight_seed"] or 0)
    assert net.streamed_supported() == 3
    cases = meta["cases"]
    B = len(cases)
    assert B >= (8 if name == "gomoku" else 4) and cfg.num_simulations >= 48
    c_in = cfg.observation_shape[0] * (cfg.stacked_observations + 1) + cfg.stacked_observations
    all_obs = synthetic.observations(B, (c_in,) + tuple(cfg.observation_shape[1:]), seed=123)
    engine = self_play.BatchedMCTS(cfg, net, B)
    rngs = [numpy.random.RandomState(c["rng_seed"]) for c in cases]
    res = engine.run([all_obs[c] for c in range(B)], [c["legal"] for c in cases], [c["to_play"] for c in cases], True, rngs)
    assert "rb_gemm_kernel" in engine.kernel_name(B)
    trees = engine.export_trees(B)
    A = len(cfg.action_space)
    diverged = 0
    for c, case in enumerate(cases):
        g = lambda k: z[f"c{c}_{k}"]
        want_trace = [(int(g("parent")[n]), int(g("parent_action")[n])) for n in range(1, len(g("parent")))]
        got_trace = []
        for n in range(1, int(trees["n_nodes"][c])):
            par = int(trees["parent"][c, n])
            slot = int(numpy.nonzero(trees["child"][c, par] == n)[0][0])
            got_trace.append((par, case["legal"][slot] if par == 0 else slot))
        k = next((k for k in range(len(want_trace)) if k >= len(got_trace) or got_trace[k] != want_trace[k]), None)
        if k is not None:
            tree = mcts_oracle.run_search(cfg, net_oracle.NetworkEvaluator(net_oracle.make_oracle_network(cfg, sd), cfg.support_size),
                                          all_obs[c], case["legal"], case["to_play"], True, numpy.random.RandomState(case["rng_seed"]))
            assert [(p, a) for p, a, _ in tree.trace] == want_trace        # the oracle IS the reference on this tree
            gap, depth = tree.margins[k]
            print(f"{name}: tree {c} leaves the reference's trace at simulation {k} (reference {want_trace[k]}, device "
                  f"{got_trace[k] if k < len(got_trace) else None}); UCB top-2 margin there {gap:.3e} at depth {depth}")
            assert gap < 5e-4, (name, c, k, gap)
            diverged += 1
            continue
        want = numpy.zeros(A, numpy.int32)
        for s, a in enumerate(case["legal"]):
            ch = g("child")[0, s]
            want[a] = g("visit")[ch] if ch >= 0 else 0
        assert numpy.array_equal(res.visit_counts[c], want), (name, c)
        # decoded scalars: the inverse value transform cancels ~3 digits in fp32; through 25
        # 128-channel layers two evaluations of the value logits that agree to 1e-5 decode to values 1.5e-4 apart
        want_rv = g("value_sum")[0] / g("visit")[0]
        assert abs(res.root_values[c] - want_rv) < 3e-4 * max(1.0, abs(want_rv))
        assert res.max_tree_depth[c] == int(g("max_tree_depth"))
    print(f"{name}: {B - diverged}/{B} trees follow the reference's trace in every simulation")
    assert diverged <= B // 4, (name, diverged, B)


class _Evaluator(net_oracle.NetworkEvaluator):
    """NetworkEvaluator that feeds the observation in the network's own precision (binary64 yardstick runs)."""

    def initial(self, observation, actions):
        with torch.no_grad():
            dtype = getattr(self.net, "dtype", torch.float32)
            obs = torch.tensor(observation).to(dtype).unsqueeze(0)
            return self._finish(*self.net.initial_inference(obs), actions)


def test_gomoku_search_against_oracle_at_size(backend):
    """
    64 gomoku trees x 60 simulations on the shipped network: invariants, repeatability, and sampled trees against the
    CPU oracle simulation by simulation (a diverging tree must diverge at a near-tie of the oracle's UCB scores).
    """
    import test_gpu_parity as parity
    cfg = configs.gomoku(num_simulations=60)
    net = models.MuZeroNetwork(cfg)
    sd = synthetic.fill_state_dict(net.state_dict(), 9)
    net.set_weights(sd)
    B = 64
    obs = synthetic.observations(B, net.input_shape, seed=4)
    rs = numpy.random.RandomState(2)
    legal = [sorted(rs.choice(121, size=rs.randint(2, 122), replace=False).tolist()) for _ in range(B)]
    to_play = [int(i % 2) for i in range(B)]
    seeds = [3000 + i for i in range(B)]
    engine = self_play.BatchedMCTS(cfg, net, B)
    res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(s) for s in seeds])
    assert "rb_gemm_kernel" in engine.kernel_name(B)
    parity._tree_invariants(cfg, res, 60)
    for i in range(B):
        assert set(numpy.nonzero(res.visit_counts[i])[0]).issubset(set(legal[i]))
    res2 = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(s) for s in seeds])
    assert numpy.array_equal(res.visit_counts, res2.visit_counts)
    assert numpy.array_equal(res.root_values.view(numpy.int64), res2.root_values.view(numpy.int64))
    sample = list(range(0, B, 4))
    traces = parity._device_trace(lambda n: self_play.BatchedMCTS(cfg, net, n), cfg, [obs[i] for i in sample],
                                  [legal[i] for i in sample], [to_play[i] for i in sample], [seeds[i] for i in sample])
    o32 = net_oracle.make_oracle_network(cfg, sd)
    o64 = net_oracle.make_oracle_network(cfg, sd, dtype=torch.float64)
    identical = 0
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for i, got in zip(sample, traces):
            run = lambda onet: mcts_oracle.run_search(cfg, _Evaluator(onet, cfg.support_size), obs[i], legal[i],
                                                      to_play[i], True, numpy.random.RandomState(seeds[i]))
            tree = run(o32)
            want = [(p, a) for p, a, _ in tree.trace]
            k = next((k for k in range(len(want)) if k >= len(got) or got[k] != want[k]), None)
            if k is not None:   # a diverging tree must diverge at a near-tie of the oracle's UCB scores
                gap, depth = tree.margins[k]
                print(f"gomoku: tree {i} diverges at simulation {k} (oracle {want[k]}, device {got[k] if k < len(got) else None}); "
                      f"oracle UCB top-2 margin {gap:.3e} at depth {depth}")
                assert gap < 5e-4, (i, k, gap)
                continue
            identical += 1
            assert tree.root_visit_counts(cfg.action_space) == list(res.visit_counts[i])
            assert res.max_tree_depth[i] == tree.max_depth
            # Root value of a tree that agrees in every simulation.  Yardstick: the SAME search evaluated in binary64.
            # Random 128-channel weights produce near-flat planes in front of the per-plane min-max scaling
            # (models.py:541-549), which divides fp32 round-off by the plane's range: on such trees torch's own
            # fp32 result moves by several 1e-3 with the thread count.  Bound: 3e-3, or 8x the error torch-fp32 has
            # against binary64 on this very tree.
            rv32 = tree.node_value(0)
            t64 = run(o64)
            tol = 30 * TOL
            if [(p, a) for p, a, _ in t64.trace] == want:
                tol = max(tol, 8 * abs(rv32 - t64.node_value(0)))
                ref = t64.node_value(0)
            else:
                ref = rv32
            err = abs(res.root_values[i] - ref)
            print(f"gomoku: tree {i} identical in every simulation; root value error {err:.2e} (torch fp32 vs f64 {abs(rv32 - ref):.2e})")
            assert err < tol * max(1.0, abs(ref)), (i, res.root_values[i], rv32, ref)
    finally:
        torch.set_num_threads(threads)
    print(f"gomoku: {identical}/{len(sample)} sampled trees identical to the oracle in every simulation")
    assert identical >= (len(sample) * 3) // 4


@pytest.mark.parametrize("name,B,S", [("gomoku", 9, 30), ("connect4", 50, 40), ("tictactoe", 7, 25), ("atari_narrow", 5, 12)])
def test_row_kernels_bit_identical_to_one_thread_per_tree(backend, name, B, S):
    """
    The streamed path's tree kernels (a 16-lane row per tree, csrc/mzx_row_search.h) against the generic operators
    (one thread per tree) around the SAME network engine: every statistic of the finished trees bit for bit.
    B is no multiple of four (a partial wavefront of rows), ragged legal sets, both players.
    """
    make = STREAMED_CASES[name][0]
    cfg = make()
    cfg.num_simulations = S
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 12))
    net.set_mode(3)
    A = len(cfg.action_space)
    obs = synthetic.observations(B, net.input_shape, seed=6)
    rs = numpy.random.RandomState(4)
    legal = [sorted(rs.choice(A, size=rs.randint(1, A + 1), replace=False).tolist()) for _ in range(B)]
    to_play = [int(i % len(cfg.players)) for i in range(B)]
    out = {}
    for mode in (0, 1):
        engine = self_play.BatchedMCTS(cfg, net, B, mode=mode)
        res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(500 + i) for i in range(B)])
        kernel = engine.kernel_name(B)
        assert ("row_select_kernel" in kernel) == (mode == 1), kernel
        out[mode] = (res, engine.export_trees(B))
    (r0, t0), (r1, t1) = out[0], out[1]
    assert numpy.array_equal(r0.visit_counts, r1.visit_counts)
    assert numpy.array_equal(r0.root_values.view(numpy.int64), r1.root_values.view(numpy.int64))
    assert numpy.array_equal(r0.max_tree_depth, r1.max_tree_depth) and numpy.array_equal(r0.tape_used, r1.tape_used)
    for key in ("visit", "value_sum", "reward", "prior", "child", "parent", "to_play", "minmax", "n_nodes"):
        a, b = t0[key], t1[key]
        if a.dtype == numpy.float64:
            a, b = a.view(numpy.int64), b.view(numpy.int64)
        assert numpy.array_equal(a, b), (name, key)


WAVE_SELECT_CASES = {
    "gomoku": (lambda: STREAMED_CASES["gomoku"][0](), 9, 30, False),      # 121 actions: two 64-slot chunks per lane
    "wide200": (lambda: configs.connect4(observation_shape=(3, 4, 8), action_space=list(range(200)), channels=32, blocks=1), 11, 40, False),   # four chunks
    "wide32_deep": (lambda: configs.connect4(observation_shape=(3, 4, 8), action_space=list(range(32)), channels=64, blocks=1), 10, 100, True),
    "atari_narrow": (lambda: STREAMED_CASES["atari_narrow"][0](), 5, 12, False),
}


@pytest.mark.parametrize("name", sorted(WAVE_SELECT_CASES))
def test_wave_per_tree_selection_walks_the_same_trees(backend, name):
    """
    wave_select_kernel (csrc/mzx_row_search.h, round 6: a WAVEFRONT per tree scores the child slots of a wide action space, lane
    l the slots l, l + 64, ...) against row_select_kernel<0> (a 16-lane row per tree, slots l, l + 16, ...; tuning "wave_select"
    = 0) and against the generic operators (one thread per tree): every statistic of the finished trees bit for bit -- ragged
    legal sets (root slots), forced ties at every first level (tape draws in slot order), both players, walks beyond sixteen
    plies (the path record), a shard that is no multiple of four.
    """
    make, B, S, flat = WAVE_SELECT_CASES[name]
    cfg = make()
    cfg.num_simulations = S
    net = models.MuZeroNetwork(cfg)
    sd = synthetic.fill_state_dict(net.state_dict(), 12)
    if flat:
        last = [k for k in sd if "fc_policy" in k and k.endswith(".weight")][-1]
        sd[last] = sd[last] * 0
    net.set_weights(sd)
    net.set_mode(3)
    A = len(cfg.action_space)
    obs = synthetic.observations(B, net.input_shape, seed=6)
    rs = numpy.random.RandomState(4)
    legal = [list(range(A))] * B if flat else [sorted(rs.choice(A, size=rs.randint(1, A + 1), replace=False).tolist()) for _ in range(B)]
    to_play = [int(i % len(cfg.players)) for i in range(B)]
    out = []
    for mode, wave in ((0, 1), (1, 0), (1, 1)):
        backend.lib.tuning_set("wave_select", wave)
        engine = self_play.BatchedMCTS(cfg, net, B, mode=mode)
        res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(500 + i) for i in range(B)])
        kernel = engine.kernel_name(B)
        assert ("row_select_kernel" in kernel) == (mode == 1), kernel
        out.append((res, engine.export_trees(B)))
    if flat:
        assert numpy.asarray(out[0][0].max_tree_depth).max() >= 17
    for k, label in ((1, "a row per tree"), (2, "a wavefront per tree")):
        (r0, t0), (r1, t1) = out[0], out[k]
        assert numpy.array_equal(r0.visit_counts, r1.visit_counts), (name, label)
        assert numpy.array_equal(r0.root_values.view(numpy.int64), r1.root_values.view(numpy.int64)), (name, label)
        assert numpy.array_equal(r0.max_tree_depth, r1.max_tree_depth) and numpy.array_equal(r0.tape_used, r1.tape_used), (name, label)
        for key in ("visit", "value_sum", "reward", "prior", "child", "parent", "to_play", "minmax", "n_nodes"):
            a, b = t0[key], t1[key]
            if a.dtype == numpy.float64:
                a, b = a.view(numpy.int64), b.view(numpy.int64)
            assert numpy.array_equal(a, b), (name, label, key)


DEEP_CASES = {
    # 32 actions (several child slots per lane: row_select_wide), two players: walks of up to ~33 plies in 100 simulations
    "wide32": (lambda: configs.connect4(observation_shape=(3, 4, 8), action_space=list(range(32)), channels=64, blocks=1), 100, 17),
    # 16 actions (one slot per lane: row_select<16>), two players: up to ~48 plies in 160 simulations
    "narrow16": (lambda: configs.connect4(observation_shape=(3, 4, 4), action_space=list(range(16)), channels=32, blocks=1), 160, 32),
    # one player (the other back-propagation variant, self_play.py:412-419), 32 actions
    "wide32_one_player": (lambda: configs.connect4(observation_shape=(3, 4, 8), action_space=list(range(32)), channels=64, blocks=1,
                                                   players=[0]), 100, 16),
}


@pytest.mark.parametrize("name", sorted(DEEP_CASES))
def test_row_kernels_deep_walks_bit_identical_to_one_thread_per_tree(backend, name):
    """
    Walks DEEPER than a row's sixteen lanes (the reference constructor's gomoku weights dig 107-ply lines): the selection
    walk writes the whole path down, back-propagation takes it sixteen nodes at a time, the leaf's chunk first
    (csrc/mzx_fused_fc.h row_backprop, round 6; before: a serial walk up the parent links on lane 0).  Against the generic
    operators (one thread per tree, the serial tree_backprop of csrc/mzx_tree.h) around the same network engine: every
    statistic of the finished trees bit for bit.  A flat policy head (last layer zeroed) makes the searches dig lines;
    rows of one wavefront end at different depths (1 .. 4 chunks), full legal sets, a partial wavefront of rows.
    """
    make, S, want_depth = DEEP_CASES[name]
    cfg = make()
    cfg.num_simulations = S
    net = models.MuZeroNetwork(cfg)
    sd = synthetic.fill_state_dict(net.state_dict(), 3 if name == "wide32_one_player" else 12)
    last = [k for k in sd if "fc_policy" in k and k.endswith(".weight")][-1]
    sd[last] = sd[last] * 0
    net.set_weights(sd)
    net.set_mode(3)
    B, A = 10, len(cfg.action_space)
    obs = synthetic.observations(B, net.input_shape, seed=6)
    legal = [list(range(A))] * B
    to_play = [int(i % len(cfg.players)) for i in range(B)]
    out = {}
    for mode in (0, 1):
        engine = self_play.BatchedMCTS(cfg, net, B, mode=mode)
        res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(500 + i) for i in range(B)])
        kernel = engine.kernel_name(B)
        assert ("row_select_kernel" in kernel) == (mode == 1), kernel
        out[mode] = (res, engine.export_trees(B))
    (r0, t0), (r1, t1) = out[0], out[1]
    depths = numpy.asarray(r0.max_tree_depth)
    print(f"{name}: deepest walk per tree {depths.tolist()}")
    assert depths.max() >= want_depth and (depths >= 16).sum() >= 1, depths       # (the case exercises what it is for)
    assert numpy.array_equal(r0.visit_counts, r1.visit_counts)
    assert numpy.array_equal(r0.root_values.view(numpy.int64), r1.root_values.view(numpy.int64))
    assert numpy.array_equal(r0.max_tree_depth, r1.max_tree_depth) and numpy.array_equal(r0.tape_used, r1.tape_used)
    for key in ("visit", "value_sum", "reward", "prior", "child", "parent", "to_play", "minmax", "n_nodes"):
        a, b = t0[key], t1[key]
        if a.dtype == numpy.float64:
            a, b = a.view(numpy.int64), b.view(numpy.int64)
        assert numpy.array_equal(a, b), (name, key)


@pytest.mark.parametrize("name,B,S,split", [("gomoku", 77, 24, False), ("gomoku", 64, 16, True), ("connect4", 130, 30, True)])
def test_two_half_shards_on_two_streams_build_the_same_trees(backend, name, B, S, split, monkeypatch):
    """
    search_run_rows splits large shards into two halves on two HIP streams (default: from 1024 trees).  With the
    threshold lowered (tuning "row_split_min") the halves -- 16-tree aligned, the second one ragged -- must
    build bit for bit the trees of the undivided run, and a second run on the same handle (stream and events reused) too.
    Halves whose layers would run with other channel groups than the undivided launch (another summation order:
    gomoku at 48 + 29 trees) are not split.
    """
    make = STREAMED_CASES[name][0]
    cfg = make()
    cfg.num_simulations = S
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 12))
    net.set_mode(3)
    A = len(cfg.action_space)
    obs = synthetic.observations(B, net.input_shape, seed=6)
    rs = numpy.random.RandomState(4)
    legal = [sorted(rs.choice(A, size=rs.randint(1, A + 1), replace=False).tolist()) for _ in range(B)]
    to_play = [int(i % len(cfg.players)) for i in range(B)]
    out = []
    for split_min in (0, 32, 32):
        backend.lib.tuning_set("row_split_min", split_min)
        engine = self_play.BatchedMCTS(cfg, net, B, mode=1) if split_min == 0 or len(out) == 1 else engine
        res = engine.run(list(obs), legal, to_play, True, [numpy.random.RandomState(500 + i) for i in range(B)])
        kernel = engine.kernel_name(B)
        assert "row_select_kernel" in kernel
        if split_min == 0:
            assert "two half-shards" not in kernel
        elif split:      # halves whose layers keep the channel groups of the undivided launch
            assert "two half-shards" in kernel
        out.append((res, engine.export_trees(B)))
    (r0, t0) = out[0]
    for r1, t1 in out[1:]:
        assert numpy.array_equal(r0.visit_counts, r1.visit_counts)
        assert numpy.array_equal(r0.root_values.view(numpy.int64), r1.root_values.view(numpy.int64))
        assert numpy.array_equal(r0.max_tree_depth, r1.max_tree_depth) and numpy.array_equal(r0.tape_used, r1.tape_used)
        for key in ("visit", "value_sum", "reward", "prior", "child", "parent", "to_play", "minmax", "n_nodes"):
            a, b = t0[key], t1[key]
            if a.dtype == numpy.float64:
                a, b = a.view(numpy.int64), b.view(numpy.int64)
            assert numpy.array_equal(a, b), (name, key)
