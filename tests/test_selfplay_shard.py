"""
A shard of B games played in lock-step (SelfPlay(num_games=B), native stream bank, one batched search
per move) must produce, game by game, exactly the GameHistory a single-game actor seeded seed + i
produces (the reference's i-th actor, muzero.py:185): same Dirichlet noise, same tie draws, same
sampled actions -- for every temperature regime of SelfPlay.select_action (self_play.py:222-245).
CPU test through tests/hostcheck (the host logic is what is under test).
"""
import numpy
import pytest

import hostcheck
from mzx import configs, models, self_play, synthetic


@pytest.fixture(scope="module")
def backend():
    return hostcheck.backend()


@pytest.mark.parametrize("temperature,threshold,players,actions", [
    (1.0, None, 1, 2), (0.5, None, 1, 3), (0.25, 4, 2, 5), (0, None, 1, 2), (float("inf"), None, 1, 4)])
def test_shard_equals_independent_actors(backend, temperature, threshold, players, actions):
    cfg = configs.cartpole(num_simulations=12, max_moves=7, action_space=list(range(actions)),
                           players=list(range(players)), temperature_threshold=threshold)
    Game = synthetic.make_synthetic_game(cfg.observation_shape, actions, players)
    template = models.MuZeroNetwork(cfg, _backend=backend).state_dict()
    weights = synthetic.fill_state_dict(template, 3)
    B, seed = 6, 40
    shard = self_play.SelfPlay({"weights": weights}, Game, cfg, seed, num_games=B, _backend=backend)
    histories = shard.play_games(temperature, cfg.temperature_threshold, False, "self", 0)
    assert len(histories) == B
    for i in range(B):
        single = self_play.SelfPlay({"weights": weights}, Game, cfg, seed + i, _backend=backend)
        want = single.play_game(temperature, cfg.temperature_threshold, False, "self", 0)
        got = histories[i]
        assert [int(a) for a in got.action_history] == [int(a) for a in want.action_history], i
        assert got.reward_history == want.reward_history and got.to_play_history == want.to_play_history
        assert got.child_visits == want.child_visits
        assert numpy.array_equal(numpy.array(got.root_values).view(numpy.int64), numpy.array(want.root_values).view(numpy.int64))
        for a, b in zip(got.observation_history, want.observation_history):
            assert numpy.array_equal(a, b)


@pytest.mark.parametrize("temperature,threshold,players,actions,stacked", [
    (1.0, None, 1, 2, 0), (0.5, 3, 2, 5, 2), (0, None, 1, 3, 1)])
def test_batched_game_protocol_equals_per_object_games(backend, temperature, threshold, players, actions, stacked):
    """The optional batched plugin protocol (one object steps the shard) against the per-object Game path."""
    cfg = configs.cartpole(num_simulations=9, max_moves=6, action_space=list(range(actions)), observation_shape=(2, 1, 3),
                           players=list(range(players)), temperature_threshold=threshold, stacked_observations=stacked)
    Game = synthetic.make_synthetic_game(cfg.observation_shape, actions, players)
    Batched = synthetic.make_synthetic_batched_game(cfg.observation_shape, actions, players)
    template = models.MuZeroNetwork(cfg, _backend=backend).state_dict()
    weights = synthetic.fill_state_dict(template, 5)
    B, seed = 7, 11
    a = self_play.SelfPlay({"weights": weights}, Game, cfg, seed, num_games=B, _backend=backend)
    b = self_play.SelfPlay({"weights": weights}, Batched, cfg, seed, num_games=B, _backend=backend)
    ha = a.play_games(temperature, cfg.temperature_threshold, False, "self", 0)
    hb = b.play_games(temperature, cfg.temperature_threshold, False, "self", 0)
    for want, got in zip(ha, hb):
        assert [int(x) for x in got.action_history] == [int(x) for x in want.action_history]
        assert got.reward_history == want.reward_history and got.to_play_history == want.to_play_history
        assert got.child_visits == want.child_visits
        assert numpy.array_equal(numpy.array(got.root_values).view(numpy.int64), numpy.array(want.root_values).view(numpy.int64))
        assert len(got.observation_history) == len(want.observation_history)
        for x, y in zip(got.observation_history, want.observation_history):
            assert numpy.array_equal(x, y)
        for i in range(len(want.root_values) + 1):
            assert numpy.array_equal(got.get_stacked_observations(i, stacked, actions), want.get_stacked_observations(i, stacked, actions))


def test_lazy_shard_histories_behave_like_game_history(backend):
    """ShardGameHistory (batched protocol): fields appear as plain lists on first touch; pickling, the
    replay hand-off and mutation work as on an eager GameHistory."""
    import pickle

    from mzx import replay

    cfg = configs.cartpole(num_simulations=8, max_moves=5, PER=True, PER_alpha=0.5, td_steps=3)
    Batched = synthetic.make_synthetic_batched_game(cfg.observation_shape, 2, 1)
    Game = synthetic.make_synthetic_game(cfg.observation_shape, 2, 1)
    template = models.MuZeroNetwork(cfg, _backend=backend).state_dict()
    weights = synthetic.fill_state_dict(template, 9)
    lazy = self_play.SelfPlay({"weights": weights}, Batched, cfg, 3, num_games=4, _backend=backend).play_games(1.0, None, False, "self", 0)
    eager = self_play.SelfPlay({"weights": weights}, Game, cfg, 3, num_games=4, _backend=backend).play_games(1.0, None, False, "self", 0)
    assert all(isinstance(h, self_play.GameHistory) for h in lazy)
    assert "child_visits" not in lazy[0].__dict__            # nothing materialised yet
    for a, b in zip(lazy, eager):
        assert replay.fill_initial_priorities(a, cfg) and replay.fill_initial_priorities(b, cfg)
        assert numpy.array_equal(a.priorities, b.priorities) and a.game_priority == b.game_priority
        assert "observation_history" not in a.__dict__        # the hand-off only touched what it needs
        c = pickle.loads(pickle.dumps(a))
        for name in self_play.ShardGameHistory._LAZY[1:]:
            assert getattr(c, name) == getattr(b, name), name
        assert all(numpy.array_equal(x, y) for x, y in zip(c.observation_history, b.observation_history))
        a.root_values.append(1.5)                              # a real list from now on
        assert a.root_values[-1] == 1.5 and len(a.root_values) == len(b.root_values) + 1


@pytest.mark.parametrize("temperature,threshold", [(1.0, None), (0.5, 3)])
def test_shard_of_board_games_with_ragged_legal_sets(backend, temperature, threshold):
    """
    Tic-tac-toe through the plugin surface: legal-action lists shrink and differ between the games of a shard,
    games end at different moves -- the shard path must still equal independent single-game actors.
    """
    import games_fixture

    cfg = configs.tictactoe(num_simulations=10, temperature_threshold=threshold)
    Game = games_fixture.GAMES["tictactoe"]
    template = models.MuZeroNetwork(cfg, _backend=backend).state_dict()
    weights = synthetic.fill_state_dict(template, 21)
    B, seed = 6, 70
    shard = self_play.SelfPlay({"weights": weights}, Game, cfg, seed, num_games=B, _backend=backend)
    histories = shard.play_games(temperature, cfg.temperature_threshold, False, "self", 0)
    lengths = set()
    for i in range(B):
        single = self_play.SelfPlay({"weights": weights}, Game, cfg, seed + i, _backend=backend)
        want = single.play_game(temperature, cfg.temperature_threshold, False, "self", 0)
        got = histories[i]
        assert [int(a) for a in got.action_history] == [int(a) for a in want.action_history], i
        assert got.reward_history == want.reward_history and got.to_play_history == want.to_play_history
        assert got.child_visits == want.child_visits
        assert numpy.array_equal(numpy.array(got.root_values).view(numpy.int64), numpy.array(want.root_values).view(numpy.int64))
        for a, b in zip(got.observation_history, want.observation_history):
            assert numpy.array_equal(numpy.array(a), numpy.array(b))
        lengths.add(len(got.action_history))
    assert len(lengths) > 1, "the games of the shard should end at different moves for this test to mean anything"


def test_action_draws_with_mixed_temperatures_follow_each_game_s_own_stream(backend):
    """
    _select_actions_bank with DIFFERENT non-zero temperatures in one call: every game is drawn exactly
    once, under its own temperature, from its own stream -- the same action and the same stream position as the
    reference's select_action (self_play.py:222-245) on a RandomState seeded like that game.
    """
    import types

    cfg = configs.cartpole(num_simulations=4, action_space=list(range(5)))
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 2)
    k, seed = 6, 90
    sp = self_play.SelfPlay({"weights": weights}, synthetic.make_synthetic_game(cfg.observation_shape, 5, 1), cfg, seed,
                            num_games=k, _backend=backend)
    rs = numpy.random.RandomState(0)
    visits = rs.randint(0, 30, size=(k, 5)).astype(numpy.int32)
    legal = [sorted(rs.choice(5, size=rs.randint(2, 6), replace=False).tolist()) for _ in range(k)]
    for r in range(k):
        visits[r, [a for a in range(5) if a not in legal[r]]] = 0
    result = types.SimpleNamespace(visit_counts=visits, legal_actions=legal, shared_legal=None)
    temps = [1.0, 0.5, 0.25, 0.5, 0, float("inf")]
    got = sp._select_actions_bank(result, list(range(k)), temps)
    for r in range(k):
        ref = numpy.random.RandomState(seed + r)
        node = types.SimpleNamespace(children={a: types.SimpleNamespace(visit_count=int(visits[r, a])) for a in legal[r]})
        want = self_play.SelfPlay._select_action(node, temps[r], ref)
        assert int(got[r]) == int(want), (r, temps[r])
        state = sp.bank.get_state(r)
        assert state[2] == ref.get_state()[2] and numpy.array_equal(state[1], ref.get_state()[1]), r


@pytest.mark.parametrize("shape,actions,players", [((1, 1, 4), 2, 1), ((3, 5, 5), 6, 2), ((2, 1, 3), 3, 2)])
def test_synthetic_batched_game_steps_like_the_per_object_game(shape, actions, players):
    """
    The batched synthetic environment (uint32 arrays that wrap) against the per-object one (Python ints masked to 32
    bits): observations (values and dtype), rewards, side to move over 300 moves with restarts, seeds up to 2**32 - 1.
    """
    Single = synthetic.make_synthetic_game(shape, actions, players)
    Batched = synthetic.make_synthetic_batched_game(shape, actions, players)
    seeds = [0, 1, 7, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 123456789, 4000000000]
    singles, shard = [Single(s) for s in seeds], Batched(seeds)
    obs, want = shard.reset(), [g.reset() for g in singles]
    rs = numpy.random.RandomState(0)
    for move in range(300):
        for i in range(len(seeds)):
            assert obs[i].dtype == want[i].dtype and numpy.array_equal(obs[i], want[i]), (move, i)
        acts = rs.randint(0, actions, size=len(seeds))
        obs, reward, done = shard.step(acts)
        stepped = [g.step(int(a)) for g, a in zip(singles, acts)]
        want = [s[0] for s in stepped]
        assert [int(r) for r in reward] == [s[1] for s in stepped] and not numpy.asarray(done).any()
        assert [int(p) for p in shard.to_play()] == [g.to_play() for g in singles]
        if move % 50 == 49:                    # the refill hook restarts some games
            idx = [1, 4]
            obs = obs.copy()
            obs[idx] = shard.reset_games(idx)
            for i in idx:
                want[i] = singles[i].reset()
