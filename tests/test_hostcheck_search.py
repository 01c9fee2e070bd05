"""
CPU check of the generic search path end to end (network programs + tree
operators + host driver) through tests/hostcheck, against the reference traces:
fp32 heads within 1e-4, visit counts / root values expected identical
(tolerance mode of SURVEY.md section 8c').  GPU twins: test_gpu_parity.py.
"""
import json
import os

import numpy
import pytest
import torch

import hostcheck
import lockstep
from conftest import GOLDEN
from mzx import configs, models, self_play, synthetic

NET_FOR_GAME = {"cartpole": "net_fc_cartpole.npz", "tictactoe": "net_resnet_tictactoe.npz",
                "connect4": "net_resnet_connect4.npz", "lunarlander": "net_fc_lunarlander_pretrained.npz"}


@pytest.fixture(scope="module")
def backend():
    return hostcheck.backend()


def build_model(backend, cfg, weight_seed, zero_keys=(), fixture=None):
    net = models.MuZeroNetwork(cfg, _backend=backend)
    sd = fixture is not None and lockstep.fixture_weights(fixture, net.state_dict())
    sd = sd or synthetic.fill_state_dict(net.state_dict(), weight_seed or 0)
    for k in zero_keys:
        sd[k] = torch.zeros_like(sd[k])
    net.set_weights(sd)
    return net


@pytest.mark.parametrize("name", ["cartpole", "tictactoe", "connect4", "cartpole_ties", "tictactoe_custom",
                                  "cartpole_custom", "lunarlander_pretrained"])
def test_search_matches_reference(backend, name):
    z, meta, cfg = lockstep.load_fixture(name)
    cfg.num_simulations = meta["num_simulations"]
    net = build_model(backend, cfg, meta["weight_seed"], meta.get("zero_keys", ()), fixture=z)
    cases = meta["cases"]
    B = len(cases)
    engine = self_play.BatchedMCTS(cfg, net, B)
    rngs = [numpy.random.RandomState(c["rng_seed"]) for c in cases]
    obs = [z[f"c{c}_obs"] for c in range(B)]
    res = engine.run(obs, [c["legal"] for c in cases], [c["to_play"] for c in cases], True, rngs)
    A = len(cfg.action_space)
    for c, case in enumerate(cases):
        g = lambda k: z[f"c{c}_{k}"]
        want = numpy.zeros(A, numpy.int32)
        for s, a in enumerate(case["legal"]):
            ch = g("child")[0, s]
            want[a] = g("visit")[ch] if ch >= 0 else 0
        assert numpy.array_equal(res.visit_counts[c], want), (name, c)
        # decoded scalars: the inverse value transform (sqrt(1+eps*..)-1) cancels ~4 digits, so two
        # fp32 evaluations of logits that agree to 1e-6 differ by ~1e-5 RELATIVE after decoding
        want_rv = g("value_sum")[0] / g("visit")[0]
        assert abs(res.root_values[c] - want_rv) < 1e-4 * max(1.0, abs(want_rv))
        want_pv = float(g("root_predicted_value"))
        assert abs(res.root_predicted_values[c] - want_pv) < 1e-4 * max(1.0, abs(want_pv))
        assert res.max_tree_depth[c] == int(g("max_tree_depth"))
        # the host stream must end where the reference's global stream ended
        ref = numpy.random.RandomState(case["rng_seed"])
        ref.dirichlet([cfg.root_dirichlet_alpha] * len(case["legal"]))
        if res.tape_used[c]:
            ref.randint(0, 2 ** 32, size=int(res.tape_used[c]), dtype=numpy.uint32)
        assert ref.get_state()[2] == rngs[c].get_state()[2]
        assert numpy.array_equal(ref.get_state()[1], rngs[c].get_state()[1])


@pytest.mark.parametrize("name", ["fc_cartpole_stacked", "fc_lunarlander_pretrained", "resnet_tictactoe",
                                  "resnet_breakout", "resnet_breakout_cnn", "resnet_cnn_small"])
def test_network_heads_within_tolerance(backend, name):
    """The per-operator network programs (serial build) against models.py outputs, 1e-4."""
    z = numpy.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    meta = json.loads(str(z["meta"]))
    cfg = configs.BY_NAME[meta["game"]](**meta["overrides"])
    net = build_model(backend, cfg, meta["weight_seed"], fixture=z)
    assert [k for k, _, _ in meta["keys"]] == list(net.state_dict().keys())
    o = net.initial_inference(torch.tensor(z["obs"]))
    r1 = net.recurrent_inference(o[3], torch.tensor(z["act1"]))
    for tag, res in (("init", o), ("rec1", r1)):
        for key, t in zip(("value", "reward", "policy", "hidden"), res):
            ref, got = z[f"{tag}_{key}"], t.cpu().numpy()
            assert got.shape == ref.shape
            if key == "reward" and tag == "init":
                assert numpy.array_equal(got, ref)
            else:
                assert numpy.abs(got - ref).max() < 1e-4, (name, tag, key)


def load_game(name):
    z = numpy.load(os.path.join(GOLDEN, f"game_{name}.npz"))
    return z, json.loads(str(z["meta"]))


@pytest.mark.parametrize("name", ["tictactoe", "connect4", "cartpole_synth", "tictactoe_stacked",
                                  "cartpole_synth_stacked"])
def test_whole_game_matches_reference(backend, name):
    """SelfPlay.play_game: same GameHistory as the reference actor with the same seed."""
    z, meta = load_game(name)
    cfg = configs.BY_NAME[meta["game"]](**meta["overrides"])
    if meta["synthetic_game"]:
        Game = synthetic.make_synthetic_game(cfg.observation_shape, len(cfg.action_space), len(cfg.players))
    else:
        Game = pytest.importorskip("games_fixture").GAMES[meta["game"]]
    template = models.MuZeroNetwork(cfg, _backend=backend).state_dict()
    weights = synthetic.fill_state_dict(template, meta["weight_seed"])
    sp = self_play.SelfPlay({"weights": weights}, Game, cfg, meta["seed"], _backend=backend)
    gh = sp.play_game(meta["temperature"], cfg.temperature_threshold, False, "self", 0)
    assert [int(a) for a in gh.action_history] == z["action_history"].tolist()
    assert [float(r) for r in gh.reward_history] == z["reward_history"].tolist()
    assert [int(p) for p in gh.to_play_history] == z["to_play_history"].tolist()
    assert numpy.array_equal(numpy.array(gh.child_visits, numpy.float64), z["child_visits"])
    assert numpy.allclose(numpy.array(gh.root_values, numpy.float64), z["root_values"], atol=1e-4, rtol=1e-4)
    for a, b in zip(gh.observation_history, z["observation_history"]):
        assert numpy.array_equal(numpy.array(a, dtype=numpy.float64), b)


def _zero_weight_search(backend, rngs_kind, tape_words):
    """
    A network whose priors are all equal ties at EVERY selection level: the tape must grow, not
    fail.  (Not compared with the oracle: with all-zero logits torch's softmax leaves a ~1e-8 residue in the
    decoded value that min-max normalisation blows up to O(1) -- an ill-conditioned case by construction.)
    """
    cfg = configs.cartpole()
    net = models.MuZeroNetwork(cfg, _backend=backend)
    net.set_weights({k: torch.zeros_like(v) for k, v in net.state_dict().items()})
    B = 3
    obs = synthetic.observations(B, cfg.observation_shape, seed=3)
    legal = [list(cfg.action_space)] * B
    old = self_play.TAPE_WORDS
    self_play.TAPE_WORDS = tape_words
    try:
        engine = self_play.BatchedMCTS(cfg, net, B)
        if rngs_kind == "bank":
            from mzx import _rng
            bank = _rng.StreamBank(backend.lib, [900 + i for i in range(B)])
            res = engine.run(list(obs), legal, [0] * B, True, (bank, numpy.arange(B)))
            final = [bank.get_state(i) for i in range(B)]
        else:
            rngs = [numpy.random.RandomState(900 + i) for i in range(B)]
            res = engine.run(list(obs), legal, [0] * B, True, rngs)
            final = [r.get_state() for r in rngs]
    finally:
        self_play.TAPE_WORDS = old
    return res, final


@pytest.mark.parametrize("rngs_kind", ["bank", "randomstate"])
def test_tie_tape_overflow_reruns_flagged_trees(backend, rngs_kind):
    res, final = _zero_weight_search(backend, rngs_kind, 16)          # overflows: re-run with 128 words
    want, want_final = _zero_weight_search(backend, rngs_kind, 4096)  # never overflows
    assert (res.tape_used > 16).any() and (res.tape_used <= 16).sum() < len(res.tape_used)
    assert numpy.array_equal(res.visit_counts, want.visit_counts)
    assert numpy.array_equal(res.root_values, want.root_values)
    assert numpy.array_equal(res.tape_used, want.tape_used) and (res.flags == 0).all()
    cfg = configs.cartpole()
    for i in range(len(final)):
        assert final[i][2] == want_final[i][2] and numpy.array_equal(final[i][1], want_final[i][1])
        # ... which is where the reference's global stream would be: Dirichlet draw, then one word per tie draw
        ref = numpy.random.RandomState(900 + i)
        ref.dirichlet([cfg.root_dirichlet_alpha] * len(cfg.action_space))
        ref.randint(0, 2 ** 32, size=int(res.tape_used[i]), dtype=numpy.uint32)
        assert ref.get_state()[2] == final[i][2] and numpy.array_equal(ref.get_state()[1], final[i][1])


def test_arena_contents_need_not_persist(backend):
    """the pb_c / sqrt tables must survive a cleared (or re-allocated) arena."""
    cfg = configs.cartpole()
    net = build_model(backend, cfg, 5)
    B = 4
    obs = synthetic.observations(B, cfg.observation_shape, seed=8)
    engine = self_play.BatchedMCTS(cfg, net, B)
    run = lambda: engine.run(list(obs), [list(cfg.action_space)] * B, [0] * B, True,
                             [numpy.random.RandomState(40 + i) for i in range(B)])
    first = run()
    engine.arena(B).zero_()
    second = run()
    assert numpy.array_equal(first.visit_counts, second.visit_counts)
    assert numpy.array_equal(first.root_values, second.root_values)
