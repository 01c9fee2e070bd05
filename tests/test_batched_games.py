"""
mzx.games: the batched-protocol tic-tac-toe and connect4 (one object steps a whole self-play shard) against
  * the reference's own games/tictactoe.py / games/connect4.py Game classes, observation for observation
    (values and dtype), legal lists, side to move, rewards, termination, on random play incl. games that end at
    different moves (live test, build container only);
  * the per-object plugin path of mzx.self_play.SelfPlay: a shard behind the batched protocol must produce the
    GameHistory records the same shard of per-object games produces (CPU, through tests/hostcheck).
"""
import numpy
import pytest

import games_fixture
import hostcheck
from mzx import configs, games, models, self_play, synthetic


def _random_playout(Batched, make_single, B, seed):
    """Steps B games with random legal actions through both implementations; asserts equality at every move."""
    rs = numpy.random.RandomState(seed)
    shard = Batched(list(range(B)))
    singles = [make_single(i) for i in range(B)]
    obs = shard.reset()
    firsts = [g.reset() for g in singles]
    alive = numpy.ones(B, bool)
    lengths = numpy.zeros(B, int)
    for i in range(B):
        assert numpy.asarray(firsts[i]).dtype == obs[i].dtype and numpy.array_equal(firsts[i], obs[i])
    while alive.any():
        legal = shard.legal_actions()
        tp = shard.to_play()
        actions = numpy.zeros(B, numpy.int64)
        for i in range(B):
            if not alive[i]:
                continue
            want = singles[i].legal_actions()
            assert [int(a) for a in legal[i] if a >= 0] == list(want), i
            assert (legal[i][len(want):] == -1).all()
            assert int(tp[i]) == singles[i].to_play()
            actions[i] = int(rs.choice(want))
        obs, reward, done = shard.step(actions, alive.copy())
        for i in range(B):
            if not alive[i]:
                continue
            o, r, d = singles[i].step(int(actions[i]))
            assert numpy.asarray(o).dtype == obs[i].dtype, (numpy.asarray(o).dtype, obs[i].dtype)
            assert numpy.array_equal(o, obs[i]), i
            assert int(reward[i]) == r and bool(done[i]) == bool(d), i
            lengths[i] += 1
        alive &= ~numpy.asarray(done, bool)
    return lengths


@pytest.mark.parametrize("name", ["tictactoe", "connect4", "gomoku"])
def test_batched_games_equal_the_test_doubles(name):
    lengths = _random_playout(games.BATCHED[name], lambda i: games.PER_OBJECT[name](i), 40 if name != "gomoku" else 12, 5)
    assert len(set(lengths.tolist())) > 2      # games of the shard end at different moves


@pytest.mark.reference
@pytest.mark.parametrize("name", ["tictactoe", "connect4", "gomoku"])
def test_batched_games_equal_the_reference_game_files(name):
    from oracle import ref_shim
    Ref = ref_shim.game_module(name).Game
    for seed in range(3 if name != "gomoku" else 1):
        lengths = _random_playout(games.BATCHED[name], lambda i: Ref(i), 48 if name != "gomoku" else 10, 100 + seed)
        assert len(set(lengths.tolist())) > 2


@pytest.mark.reference
def test_per_object_gomoku_equals_the_reference_game_file():
    from oracle import ref_shim
    Ref = ref_shim.game_module("gomoku").Game
    rs = numpy.random.RandomState(3)
    for episode in range(4):
        a, b = Ref(episode), games.Gomoku(episode)
        oa, ob = a.reset(), b.reset()
        done = False
        while not done:
            assert numpy.asarray(oa).dtype == numpy.asarray(ob).dtype and numpy.array_equal(oa, ob)
            assert a.legal_actions() == b.legal_actions() and a.to_play() == b.to_play()
            act = int(rs.choice(a.legal_actions()))
            (oa, ra, done), (ob, rb, db) = a.step(act), b.step(act)
            assert ra == rb and done == db
        assert numpy.array_equal(oa, ob)


@pytest.mark.parametrize("name,temperature,threshold", [("tictactoe", 1.0, None), ("tictactoe", 0.5, 3), ("connect4", 1.0, None)])
def test_batched_board_games_give_the_per_object_game_histories(name, temperature, threshold):
    backend = hostcheck.backend()
    small = dict(channels=8, blocks=1) if name == "connect4" else {}     # the game logic is under test, not the network
    cfg = configs.BY_NAME[name](num_simulations=8, temperature_threshold=threshold, **small)
    template = models.MuZeroNetwork(cfg, _backend=backend).state_dict()
    weights = synthetic.fill_state_dict(template, 17)
    B, seed = 5, 30
    a = self_play.SelfPlay({"weights": weights}, games_fixture.GAMES[name], cfg, seed, num_games=B, _backend=backend)
    b = self_play.SelfPlay({"weights": weights}, games.BATCHED[name], cfg, seed, num_games=B, _backend=backend)
    ha = a.play_games(temperature, cfg.temperature_threshold, False, "self", 0)
    hb = b.play_games(temperature, cfg.temperature_threshold, False, "self", 0)
    lengths = set()
    for want, got in zip(ha, hb):
        assert [int(x) for x in got.action_history] == [int(x) for x in want.action_history]
        assert got.reward_history == want.reward_history and got.to_play_history == want.to_play_history
        assert got.child_visits == want.child_visits
        assert numpy.array_equal(numpy.array(got.root_values).view(numpy.int64), numpy.array(want.root_values).view(numpy.int64))
        assert len(got.observation_history) == len(want.observation_history)
        for x, y in zip(got.observation_history, want.observation_history):
            assert numpy.asarray(x).dtype == numpy.asarray(y).dtype and numpy.array_equal(x, y)
        lengths.add(len(got.action_history))
    assert len(lengths) > 1
