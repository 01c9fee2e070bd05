"""
mzx.replay.ReplayBuffer: the reference's own ReplayBuffer (replay_buffer.py:11-303;
storage, eviction, sampling stay its code -- composition, needs the reference on the path: build container only)
with save_game's priority step and get_batch / make_target vectorised.  With the same numpy seed and the same games
the batches must be IDENTICAL to the reference's -- sampled (game, position) pairs, n-step value targets
(binary64 bit patterns), rewards, policies, actions incl. the random actions of absorbing steps, PER weights,
gradient scales, stacked observations.  Checked against the unmodified reference when /root/reference is
present and against a fixture the reference produced (tests/golden/replay_batch.npz) everywhere.
"""
import copy
import json
import os
import types

import numpy
import pytest

from conftest import GOLDEN
from mzx import replay, self_play
from oracle import ref_shim


def make_games(seed, n_games, players, A=4, obs_shape=(2, 3, 3), reanalysed_every=3, reanalysed_dtype=numpy.float64):
    rs = numpy.random.RandomState(seed)
    games = []
    for g in range(n_games):
        T = int(rs.randint(1, 40))
        gh = self_play.GameHistory()
        gh.action_history = [0] + [int(a) for a in rs.randint(0, A, size=T)]
        gh.reward_history = [0] + [float(r) for r in rs.standard_normal(T)]
        gh.to_play_history = [int(i % players) for i in range(T + 1)]
        gh.root_values = [float(v) for v in rs.standard_normal(T)]
        visits = rs.randint(0, 20, size=(T, A)) + 1
        gh.child_visits = [[int(v) / int(row.sum()) for v in row] for row in visits]
        gh.observation_history = [rs.rand(*obs_shape).astype(numpy.float32) for _ in range(T + 1)]
        if g % reanalysed_every == 1:
            # (binary64 here: with float32 arrays -- what Reanalyse writes -- the reference's scalar arithmetic depends
            # on the numpy version, see test_float32_reanalysed_values_follow_the_pinned_numpy)
            gh.reanalysed_predicted_root_values = rs.standard_normal(T).astype(numpy.float32).astype(reanalysed_dtype)
        games.append(gh)
    return games


def config_for(per, players, stacked):
    return types.SimpleNamespace(PER=per, PER_alpha=0.5, seed=7, replay_buffer_size=10 ** 6, batch_size=24,
                                 num_unroll_steps=6, td_steps=5, discount=0.97, stacked_observations=stacked,
                                 action_space=list(range(4)), players=list(range(players)))


def as_arrays(batch):
    index_batch, (obs, actions, values, rewards, policies, weights, scales) = batch
    return dict(index=numpy.array(index_batch, numpy.int64), obs=numpy.array(obs), actions=numpy.array(actions, numpy.int64),
                values=numpy.array(values, numpy.float64), rewards=numpy.array(rewards, numpy.float64),
                policies=numpy.array(policies, numpy.float64),
                weights=None if weights is None else numpy.array(weights, numpy.float32),
                scales=numpy.array(scales, numpy.int64))


def assert_same(a, b, tag):
    for k in a:
        if a[k] is None or b[k] is None:
            assert a[k] is None and b[k] is None, (tag, k)
        elif a[k].dtype.kind == "f":
            assert a[k].shape == b[k].shape and numpy.array_equal(a[k].view(numpy.uint8), b[k].view(numpy.uint8)), (tag, k)
        else:
            assert numpy.array_equal(a[k], b[k]), (tag, k)


CASES = [(True, 1, 0), (True, 2, 2), (False, 2, 0), (False, 1, 1)]


@pytest.mark.reference
@pytest.mark.parametrize("per,players,stacked", CASES)
def test_batches_identical_to_the_reference(per, players, stacked):
    ref_shim.load()
    import replay_buffer as ref_rb
    config = config_for(per, players, stacked)
    games = make_games(11 + players, 9, players)
    theirs = ref_rb.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    ours = replay.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    for gh in games:
        theirs.save_game(copy.deepcopy(gh))
        ours.save_game(copy.deepcopy(gh))
    assert ours.total_samples == theirs.total_samples and ours.num_played_steps == theirs.num_played_steps
    for rounds in range(4):
        numpy.random.seed(100 + rounds)
        want = as_arrays(theirs.get_batch())
        numpy.random.seed(100 + rounds)
        got = as_arrays(ours.get_batch())
        assert_same(got, want, (per, players, stacked, rounds))
        if per:   # the trainer's feedback (trainer.py:97-98) changes the next round's sampling on both sides alike
            pr = numpy.abs(numpy.random.RandomState(rounds).standard_normal((config.batch_size, config.num_unroll_steps + 1))).astype("float32")
            theirs.update_priorities(pr, want["index"].tolist())
            ours.update_priorities(pr, got["index"].tolist())
        if rounds == 1:   # a reanalysed game (replay_buffer.py:204-211) invalidates the cached n-step values
            gid, gh, _ = theirs.sample_game(force_uniform=True)
            fresh = numpy.random.RandomState(5).standard_normal(len(gh.root_values)).astype(numpy.float32).astype(numpy.float64)
            for rb in (theirs, ours):
                g2 = copy.deepcopy(rb.buffer[gid])
                g2.reanalysed_predicted_root_values = fresh
                rb.update_game_history(gid, g2)


@pytest.mark.reference
def test_float32_reanalysed_values_follow_the_pinned_numpy():
    """
    Reanalyse stores float32 arrays (replay_buffer.py:361-367).  compute_target_value multiplies such an element by
    the Python float ``discount ** td_steps``: under the reference's pinned numpy 1.21.4 (requirements.lock:112)
    value-based casting makes that binary64 arithmetic on the widened value -- what n_step_values does -- while
    numpy >= 2 (NEP 50, this container) keeps the whole accumulation in float32.  So against the reference
    executed HERE the targets of reanalysed games agree to float32 round-off only, everything else exactly.
    """
    ref_shim.load()
    import replay_buffer as ref_rb
    config = config_for(True, 1, 0)
    games = make_games(12, 9, 1, reanalysed_dtype=numpy.float32)
    theirs = ref_rb.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    ours = replay.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    for gh in games:
        theirs.save_game(copy.deepcopy(gh))
        ours.save_game(copy.deepcopy(gh))
    numpy.random.seed(100)
    want = as_arrays(theirs.get_batch())
    numpy.random.seed(100)
    got = as_arrays(ours.get_batch())
    values_w, values_g = want.pop("values"), got.pop("values")
    assert_same(got, want, "float32 reanalysed")
    assert numpy.allclose(values_g, values_w, rtol=2e-6, atol=2e-6)
    if int(numpy.__version__.split(".")[0]) < 2:
        assert numpy.array_equal(values_g, values_w)


@pytest.mark.reference
def test_batches_equal_the_reference_fixture():
    ref_shim.load()
    z = numpy.load(os.path.join(GOLDEN, "replay_batch.npz"))
    meta = json.loads(str(z["meta"]))
    for c, case in enumerate(meta["cases"]):
        config = config_for(case["per"], case["players"], case["stacked"])
        ours = replay.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
        for gh in make_games(case["games_seed"], case["n_games"], case["players"]):
            ours.save_game(gh)
        for r in range(case["rounds"]):
            numpy.random.seed(case["seed0"] + r)
            got = as_arrays(ours.get_batch())
            want = {k: (z[f"c{c}_r{r}_{k}"] if f"c{c}_r{r}_{k}" in z.files else None) for k in got}
            assert_same(got, want, (c, r))


@pytest.mark.reference
def test_single_position_api_matches_the_scalar_form():
    ref_shim.load()
    config = config_for(True, 2, 0)
    rb = replay.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    gh = make_games(3, 1, 2)[0]
    rb.save_game(gh)
    T = len(gh.root_values)
    assert not rb._arrays            # nothing is cached for a game looked at outside get_batch (no id()-keyed entries)
    for pos in (0, T // 2, T - 1):
        numpy.random.seed(1)
        values, rewards, policies, actions = rb.make_target(gh, pos)
        for k, idx in enumerate(range(pos, pos + config.num_unroll_steps + 1)):
            if idx < T:
                assert values[k] == rb.compute_target_value(gh, idx)
                assert rewards[k] == gh.reward_history[idx] and list(policies[k]) == gh.child_visits[idx]
            else:
                assert values[k] == 0 and (rewards[k] == (gh.reward_history[idx] if idx == T else 0))
                assert list(policies[k]) == [0.25] * 4


@pytest.mark.reference
def test_eviction_drops_cached_arrays_and_storage_is_the_stock_code():
    ref_shim.load()
    import replay_buffer as ref_rb
    config = config_for(True, 1, 0)
    config.replay_buffer_size = 4
    rb = replay.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    assert type(rb._stock).__name__ == "ReplayBuffer" and type(rb._stock).__module__ == ref_rb.__name__
    for gh in make_games(21, 9, 1):
        rb.save_game(gh)
        numpy.random.seed(3)
        rb.get_batch()
        assert set(rb._arrays) <= set(rb.buffer) and len(rb.buffer) <= 4
    assert rb.num_played_games == 9 and min(rb.buffer) == 5
