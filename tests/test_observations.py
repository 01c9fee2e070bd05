"""
Observation pipeline + Reanalyse on the CPU test double of the C ABI:
the oracle's ``stacked_observations`` is pinned against arrays produced by the unmodified reference
``GameHistory.get_stacked_observations`` (tests/golden/obs_stack.npz, oracle/make_golden.py), and the
device operator ``mzx_obs_stack`` / the ``FrameStore`` / ``Reanalyse`` mirror are checked against both.
Integer/byte work: bit-exact.  Decoded values: 1e-4 relative (the inverse value transform cancels ~4
digits, see test_hostcheck_search).  GPU twins: tests/test_gpu_parity.py.
"""
import json
import os

import numpy
import pytest
import torch

import hostcheck
from conftest import GOLDEN
from mzx import _lib, configs, models, observations, replay, self_play, synthetic
from oracle import mcts_oracle, net_oracle

TOL = 1e-4


@pytest.fixture(scope="module")
def backend():
    return hostcheck.backend()


def _cfg(shape, k, A):
    return configs.HotPathConfig(observation_shape=tuple(shape), stacked_observations=k, action_space=list(range(A)))


def _oracle_stack(history, actions, index, k, A):
    """the reference's array after ``torch.tensor(obs).float()`` (self_play.py:280-285)."""
    st = mcts_oracle.stacked_observations(list(history), [int(a) for a in actions], index, k, A)
    return torch.tensor(numpy.array(st)).float().numpy()


def load_obs_cases():
    z = numpy.load(os.path.join(GOLDEN, "obs_stack.npz"))
    return z, json.loads(str(z["meta"]))


def test_oracle_stacking_matches_reference_arrays():
    z, cases = load_obs_cases()
    for c, case in enumerate(cases):
        hist, acts = z[f"c{c}_history"], z[f"c{c}_actions"]
        assert hist.dtype == numpy.dtype(case["dtype"])
        for j, i in enumerate(z[f"c{c}_index"]):
            got = _oracle_stack(hist, acts, int(i), case["k"], case["A"])
            assert got.dtype == numpy.float32 and numpy.array_equal(got, z[f"c{c}_stacked"][j]), (c, i)


def test_stack_history_bit_exact(backend):
    z, cases = load_obs_cases()
    for c, case in enumerate(cases):
        hist, acts = z[f"c{c}_history"], z[f"c{c}_actions"]
        cfg = _cfg(case["shape"], case["k"], case["A"])
        T1 = hist.shape[0]
        out = observations.stack_history(backend, cfg, list(hist), list(acts)).cpu().numpy()
        assert out.shape == (T1,) + z[f"c{c}_stacked"].shape[1:]
        assert numpy.array_equal(out, z[f"c{c}_stacked"][:T1]), c
        # a prefix only (Reanalyse asks for len(root_values) = T positions of T + 1 observations)
        part = observations.stack_history(backend, cfg, list(hist), list(acts), count=T1 - 1).cpu().numpy()
        assert numpy.array_equal(part, z[f"c{c}_stacked"][: T1 - 1])
        assert observations.stack_history(backend, cfg, list(hist), list(acts), count=0).shape[0] == 0


@pytest.mark.parametrize("shape,k,A,G,moves", [((3, 3, 3), 2, 9, 5, 7), ((1, 1, 4), 3, 2, 3, 9), ((2, 4, 6), 4, 5, 4, 3),
                                               ((3, 4, 4), 0, 3, 2, 4)])
def test_frame_store_ring_matches_oracle(backend, shape, k, A, G, moves):
    """A shard's frame store (ring of k + 1 slots, wraps) against per-game histories through the oracle."""
    check_frame_store(backend, shape, k, A, G, moves)


def check_frame_store(backend, shape, k, A, G, moves, probe=None):
    rs = numpy.random.RandomState(5)
    cfg = _cfg(shape, k, A)
    store = observations.FrameStore(cfg, G, backend)
    assert store.ring == k + 1 and store.sample_shape == (shape[0] * (k + 1) + k,) + tuple(shape[1:])
    hist = [[] for _ in range(G)]
    acts = [[] for _ in range(G)]
    for t in range(moves + 1):
        frame = rs.standard_normal((G,) + tuple(shape)).astype(numpy.float32)
        a = numpy.zeros(G, numpy.int64) if t == 0 else rs.randint(0, A, size=G)
        store.push(frame, None if t == 0 else a)
        for g in range(G):
            hist[g].append(frame[g])
            acts[g].append(int(a[g]))
        games = None if t % 2 == 0 else numpy.sort(rs.choice(G, size=max(1, G // 2), replace=False))
        got = store.stacked(games).cpu().numpy()
        ids = range(G) if games is None else games
        assert got.shape[0] == len(ids)
        for row, g in enumerate(ids):
            if probe is not None and g not in probe:
                continue
            want = _oracle_stack(hist[g], acts[g], -1, k, A)
            assert numpy.array_equal(got[row], want), (t, g)


def test_obs_stack_rejects_bad_arguments(backend):
    lib = backend.lib
    cfg = _cfg((3, 3, 3), 2, 9)
    with pytest.raises(ValueError):
        observations.FrameStore(cfg, 4, backend, ring=2)          # cannot hold k + 1 frames
    store = observations.FrameStore(cfg, 4, backend)
    with pytest.raises(_lib.MzxError):
        store.stacked()                                           # nothing pushed yet
    with pytest.raises(ValueError):
        store.push(numpy.zeros((4, 3, 3, 4), numpy.float32))      # wrong frame shape
    store.push(numpy.zeros((4, 3, 3, 3), numpy.float32))
    with pytest.raises(ValueError):
        store.stacked([0, 4])                                     # game index out of range
    import ctypes
    bad = observations._layout((3, 3, 3), 2, 0, 1, 3)             # action_space_size 0
    assert lib.mzx_obs_stacked_floats(ctypes.byref(bad)) == 0
    assert lib.mzx_obs_stack(ctypes.byref(bad), None, None, None, None, 0, 1, None, None) != 0
    assert b"layout" in lib.mzx_last_error()


def test_support_to_scalar_matches_oracle(backend):
    rs = numpy.random.RandomState(3)
    for support in (10, 1, 300):
        logits = torch.tensor(rs.standard_normal((37, 2 * support + 1)).astype(numpy.float32) * 3)
        got = models.support_to_scalar(logits, support, _backend=backend).cpu().numpy()
        want = net_oracle.support_to_scalar(logits, support).numpy()
        assert got.shape == want.shape == (37, 1)
        assert numpy.allclose(got, want, atol=3 * TOL, rtol=3 * TOL)
    with pytest.raises(ValueError):
        models.support_to_scalar(torch.zeros(4, 20), 10, _backend=backend)


def load_game(name):
    z = numpy.load(os.path.join(GOLDEN, f"game_{name}.npz"))
    return z, json.loads(str(z["meta"]))


def _history_from_fixture(z):
    gh = self_play.GameHistory()
    gh.observation_history = [o for o in z["observation_history"]]
    gh.action_history = [int(a) for a in z["action_history"]]
    gh.reward_history = [float(r) for r in z["reward_history"]]
    gh.to_play_history = [int(p) for p in z["to_play_history"]]
    gh.root_values = [float(v) for v in z["root_values"]]
    gh.child_visits = z["child_visits"].tolist()
    return gh


@pytest.mark.parametrize("name", ["tictactoe_stacked", "cartpole_synth_stacked"])
def test_oracle_reanalyse_matches_reference(name):
    """Pins the checker: oracle stacking + oracle network + oracle decode == the reference's Reanalyse worker."""
    z, meta = load_game(name)
    cfg = configs.BY_NAME[meta["game"]](**meta["overrides"])
    gh = _history_from_fixture(z)
    template = models_template(cfg)
    weights = synthetic.fill_state_dict(template, meta["weight_seed"])
    net = net_oracle.make_oracle_network(cfg, weights)
    A = len(cfg.action_space)
    obs = numpy.array([mcts_oracle.stacked_observations(gh.observation_history, gh.action_history, i,
                                                        cfg.stacked_observations, A) for i in range(len(gh.root_values))])
    with torch.no_grad():
        values = net_oracle.support_to_scalar(net.initial_inference(torch.tensor(obs).float())[0], cfg.support_size)
    want = z["reanalysed_predicted_root_values"]
    assert numpy.allclose(torch.squeeze(values).numpy(), want, atol=1e-5, rtol=1e-5)


def models_template(cfg):
    """reference-format state_dict template (keys + shapes) without touching the reference."""
    return models.MuZeroNetwork(cfg, _backend=hostcheck.backend()).state_dict()


@pytest.mark.parametrize("name", ["tictactoe_stacked", "cartpole_synth_stacked"])
def test_reanalyse_matches_reference(backend, name):
    check_reanalyse(backend, name)


def check_reanalyse(backend, name):
    """mzx.replay.Reanalyse (worker loop included) against the values the reference's worker stored."""
    z, meta = load_game(name)
    cfg = configs.BY_NAME[meta["game"]](**meta["overrides"])
    gh = _history_from_fixture(z)
    template = models.MuZeroNetwork(cfg, _backend=backend).state_dict()
    weights = synthetic.fill_state_dict(template, meta["weight_seed"])

    class Storage:
        def __init__(self):
            self.passes, self.info = 0, {"num_played_games": 1, "terminate": False, "weights": weights}

        def get_info(self, key):
            if key == "training_step":
                self.passes += 1
                return 0 if self.passes == 1 else cfg.training_steps
            return self.info[key]

        def set_info(self, key, value=None):
            self.info[key] = value

    class Buffer:
        updated = None

        def sample_game(self, force_uniform=False):
            assert force_uniform
            return 7, gh, 1.0

        def update_game_history(self, game_id, game_history):
            assert game_id == 7
            self.updated = game_history

    storage, buffer = Storage(), Buffer()
    # start from different weights: the loop must pull the fresh ones from the storage (replay_buffer.py:339)
    worker = replay.Reanalyse({"weights": synthetic.fill_state_dict(template, 999), "num_reanalysed_games": 3}, cfg,
                              _backend=backend)
    worker.reanalyse(buffer, storage)
    assert buffer.updated is gh and worker.num_reanalysed_games == 4 and storage.info["num_reanalysed_games"] == 4
    got, want = gh.reanalysed_predicted_root_values, z["reanalysed_predicted_root_values"]
    assert got.dtype == numpy.float32 and got.shape == want.shape
    assert numpy.allclose(got, want, atol=3 * TOL, rtol=3 * TOL), numpy.abs(got - want).max()
    # downstream consumer: the n-step targets switch to the refreshed values (replay_buffer.py:236-240)
    cfg.td_steps, cfg.PER, cfg.PER_alpha = 3, True, 0.5
    gh.priorities = None
    assert replay.fill_initial_priorities(gh, cfg)
