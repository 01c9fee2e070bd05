"""
SelfPlay.play_rounds: finished games hand their slot to the next game at once, so every search runs at the full shard
width -- the reference actor's own loop (self_play.py:31-52: play_game, save, play the next game on the SAME numpy
stream and the SAME Game object), per slot.  The contract: the k-th game slot s hands out equals the k-th game of a
lone reference-style actor seeded ``seed + s`` (muzero.py:185 seeds actor s so) that plays its games one after the
other -- for the per-object plugin surface and for the batched protocol, with ragged game lengths, temperature
thresholds and stacked observations.  CPU test through tests/hostcheck (the host logic is what is under test).
"""
import numpy
import pytest

import games_fixture
import hostcheck
from mzx import configs, games, models, self_play, synthetic


@pytest.fixture(scope="module")
def backend():
    return hostcheck.backend()


def _same(got, want, where):
    assert [int(a) for a in got.action_history] == [int(a) for a in want.action_history], where
    assert [float(x) for x in got.reward_history] == [float(x) for x in want.reward_history], where
    assert [int(x) for x in got.to_play_history] == [int(x) for x in want.to_play_history], where
    assert got.child_visits == want.child_visits, where
    assert numpy.array_equal(numpy.array(got.root_values, numpy.float64).view(numpy.int64),
                             numpy.array(want.root_values, numpy.float64).view(numpy.int64)), where
    assert len(got.observation_history) == len(want.observation_history), where
    for a, b in zip(got.observation_history, want.observation_history):
        assert numpy.array_equal(numpy.array(a), numpy.array(b)), where


def _lone_actor_games(weights, Game, cfg, seed, temperature, count, backend):
    actor = self_play.SelfPlay({"weights": weights}, Game, cfg, seed, _backend=backend)
    return [actor.play_game(temperature, cfg.temperature_threshold, False, "self", 0) for _ in range(count)]


CASES = {
    # name: (config, per-object Game, batched Game (or None), temperature)
    "tictactoe": (lambda: configs.tictactoe(num_simulations=8), lambda: games_fixture.GAMES["tictactoe"],
                  lambda: games.TicTacToeBatched, 1.0),
    "tictactoe-threshold": (lambda: configs.tictactoe(num_simulations=8, temperature_threshold=3),
                            lambda: games.TicTacToe, lambda: games.TicTacToeBatched, 0.5),
    "connect4": (lambda: configs.connect4(num_simulations=6, blocks=1, channels=8), lambda: games.Connect4,
                 lambda: games.Connect4Batched, 1.0),
}


@pytest.mark.parametrize("protocol", ["per-object", "per-object-pipelined", "batched", "batched-pipelined"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_slot_games_equal_a_lone_actor_s_sequence(backend, name, protocol):
    make_cfg, make_game, make_batched, temperature = CASES[name]
    cfg = make_cfg()
    if protocol.endswith("pipelined"):      # two slot groups take turns: one is searched while the other is stepped
        cfg.self_play_pipeline = True
    elif protocol == "batched":
        cfg.self_play_pipeline = False
    Game = make_game()
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 21)
    B, seed, per_slot = 5, 70, 3
    shard = self_play.SelfPlay({"weights": weights}, make_batched() if protocol.startswith("batched") else Game, cfg, seed,
                               num_games=B, _backend=backend)
    by_slot = {s: [] for s in range(B)}
    rounds_played = 0
    searches0 = shard.stats["searches"]
    while min(len(v) for v in by_slot.values()) < per_slot:
        before = shard.stats["searches"]
        out = shard.play_rounds(temperature, cfg.temperature_threshold, min_games=2)
        assert len(out) >= 2
        rounds_played += (shard.stats["searches"] - before) // B
        assert len(shard.finished_slots) == len(out)
        for gh, slot in zip(out, shard.finished_slots):
            by_slot[slot].append(gh)
    assert shard.stats["searches"] - searches0 == rounds_played * B      # every round searched ALL slots
    lengths = set()
    for s in range(B):
        want = _lone_actor_games(weights, Game, cfg, seed + s, temperature, per_slot, backend)
        for k in range(per_slot):
            _same(by_slot[s][k], want[k], (name, protocol, s, k))
            lengths.add(len(want[k].action_history))
    assert len(lengths) > 1, "games should end at different moves for this test to mean anything"
    if protocol.endswith("pipelined"):
        assert len(shard._live["groups"]) == 2 and [len(g["slots"]) for g in shard._live["groups"]] == [3, 2]
    if protocol == "per-object-pipelined":
        assert all(g["pending"] is None or g["pending"].done() for g in shard._live["groups"])   # nothing runs between calls
    if protocol == "batched-pipelined":      # the batched protocol never leaves a search queued across calls
        assert all(g["pending"] is None for g in shard._live["groups"])
    if protocol == "batched":
        assert len(shard._live["groups"]) == 1
    shard.close_game()


@pytest.mark.parametrize("protocol", ["per-object", "per-object-pipelined", "batched", "batched-pipelined"])
def test_refill_with_stacked_observations_and_fixed_length_games(backend, protocol):
    """Synthetic game (never ends on its own: max_moves does), stacked observations through the device frame store:
    a restarted slot must see zeros in front of its first frame, not the previous game's frames."""
    cfg = configs.cartpole(num_simulations=7, max_moves=4, action_space=list(range(3)), observation_shape=(2, 1, 3),
                           players=list(range(2)), stacked_observations=2)
    Game = synthetic.make_synthetic_game(cfg.observation_shape, 3, 2)
    Batched = synthetic.make_synthetic_batched_game(cfg.observation_shape, 3, 2)
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 5)
    B, seed = 4, 11
    cfg.self_play_pipeline = protocol.endswith("pipelined")    # (two groups of two slots, a frame store each)
    shard = self_play.SelfPlay({"weights": weights}, Batched if protocol.startswith("batched") else Game, cfg, seed, num_games=B,
                               _backend=backend)
    first = shard.play_rounds(0.5, None)              # one shard's worth: all four games end with round 4
    first_slots = list(shard.finished_slots)
    assert len(first) == B and shard.stats["searches"] == 4 * B
    second = shard.play_rounds(0.5, None, max_rounds=3)
    assert second == []                               # three rounds into the next games: nothing finished
    second = shard.play_rounds(0.5, None)
    assert len(second) == B and shard.stats["searches"] == 8 * B
    for s in range(B):
        want = _lone_actor_games(weights, Game, cfg, seed + s, 0.5, 2, backend)
        _same(first[first_slots.index(s)], want[0], (protocol, s, 0))
        _same(second[shard.finished_slots.index(s)], want[1], (protocol, s, 1))


def test_play_games_after_play_rounds_starts_over(backend):
    """play_games (whole shards in lock-step) resets every game: it must not continue play_rounds' games in progress."""
    cfg = configs.tictactoe(num_simulations=6)
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 2)
    shard = self_play.SelfPlay({"weights": weights}, games.TicTacToeBatched, cfg, 5, num_games=3, _backend=backend)
    shard.play_rounds(1.0, None, max_rounds=2)
    assert shard._live is not None
    out = shard.play_games(1.0, None, False, "self", 0)
    assert len(out) == 3 and shard._live is None
    assert all(len(h.action_history) >= 6 for h in out)       # whole games (a tic-tac-toe game lasts >= 5 moves)


def test_continuous_self_play_refills_by_default(backend):
    """continuous_self_play hands games to the replay buffer as they finish; all slots are searched in every round."""
    from mzx import shared_storage

    cfg = configs.tictactoe(num_simulations=5, training_steps=10, ratio=None, self_play_delay=0)
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 4)
    storage = shared_storage.LocalStorage(weights=weights, training_step=0, terminate=False, num_played_games=0,
                                          num_played_steps=0)

    class Buffer:
        def __init__(self):
            self.games = []

        def save_game(self, game_history, shared_storage=None):
            self.games.append(game_history)
            if len(self.games) >= 9:
                storage.set_info("terminate", True)

    buf = Buffer()
    B = 4
    actor = self_play.SelfPlay({"weights": weights}, games.TicTacToeBatched, cfg, 1, num_games=B, _backend=backend)
    actor.continuous_self_play(storage, buf)
    assert len(buf.games) >= 9
    assert actor.stats["searches"] % B == 0
    moves = sum(len(g.action_history) - 1 for g in buf.games)
    assert actor.stats["searches"] >= moves                   # (games in progress at the stop are not handed out)
    assert all(g.priorities is not None or not getattr(cfg, "PER", False) for g in buf.games)


def test_two_slot_groups_play_what_one_group_plays(backend):
    """A whole shard, several calls (so that a call ends with a search queued for the next round), stacked observations:
    two slot groups taking turns hand out, slot by slot, the games one group hands out."""
    def run(pipeline, B=96):
        cfg = configs.cartpole(num_simulations=6, max_moves=9, stacked_observations=2)
        cfg.self_play_pipeline = pipeline
        Game = synthetic.make_synthetic_game(cfg.observation_shape, len(cfg.action_space), len(cfg.players))
        weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 3)
        shard = self_play.SelfPlay({"weights": weights}, Game, cfg, 7, num_games=B, _backend=backend)
        by_slot = {}
        for min_games in (1, 40, 40, 150):       # calls that end in the middle of the games of most slots
            for gh, slot in zip(shard.play_rounds(0.7, None, min_games=min_games), shard.finished_slots):
                by_slot.setdefault(slot, []).append(gh)
        groups = len(shard._live["groups"])
        shard.close_game()
        return by_slot, groups, shard.stats["searches"]

    one, g1, s1 = run(False)
    two, g2, s2 = run(True)
    assert (g1, g2) == (1, 2) and s1 == s2 and one.keys() == two.keys()
    for slot in one:
        assert len(one[slot]) == len(two[slot])
        for a, b in zip(one[slot], two[slot]):
            _same(a, b, slot)


def test_dropped_queued_search_gives_its_draws_back(backend):
    """A pipelined play_rounds call that ends with a search queued for the next round, followed by play_games instead of
    another play_rounds: the queued search is dropped and the root noise / tie words it drew go back to the slots' streams
     -- the streams are then exactly those of a run that never queued it."""
    def streams(pipeline):
        cfg = configs.cartpole(num_simulations=5, max_moves=7)
        cfg.self_play_pipeline = pipeline
        Game = synthetic.make_synthetic_game(cfg.observation_shape, len(cfg.action_space), len(cfg.players))
        weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 3)
        shard = self_play.SelfPlay({"weights": weights}, Game, cfg, 11, num_games=8, _backend=backend)
        # (fixed-length games: all eight end in one round; the first group's four leave the call short of five games, so its
        # next search is queued before the second group's four end the call)
        shard.play_rounds(1.0, None, min_games=5)
        queued = any(g.get("pending") is not None for g in shard._live["groups"])
        shard._drain_searches()
        states = [shard.bank.get_state(s) for s in range(8)]
        shard.close_game()
        return states, queued

    one, q1 = streams(False)
    two, q2 = streams(True)
    assert not q1 and q2
    for a, b in zip(one, two):
        assert a[2] == b[2] and (a[1] == b[1]).all() and a[3] == b[3]


def test_reward_dtype_change_across_a_ring_growth_keeps_earlier_rows(backend):
    """a batched plugin that returns float rewards for its first rounds and integer arrays afterwards (numpy.asarray
    of an all-integer list).  The move log is a ring that grows while a game outlasts it; the grown ring must take the wider of
    the old and the new dtype, or the float rewards already logged are truncated by the copy (they then flow into the PER
    priorities).  max_moves 40 > the ring's first 16 rows, so the growth happens in an integer round."""
    cfg = configs.cartpole(num_simulations=4, max_moves=40)
    Base = synthetic.make_synthetic_batched_game(cfg.observation_shape, len(cfg.action_space), len(cfg.players))

    class HalfStepRewards(Base):
        def step(self, actions, active=None):
            obs, reward, done = super().step(actions, active)
            early = self.t <= 10               # (per game: refilled slots restart at 0)
            if early.all():
                return obs, reward * 0.5 + 0.25, done
            assert not early.any()
            return obs, numpy.asarray([int(x) for x in reward]), done

    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 9)
    shard = self_play.SelfPlay({"weights": weights}, HalfStepRewards, cfg, 3, num_games=3, _backend=backend)
    games_out = shard.play_rounds(1.0, None)
    assert len(games_out) == 3
    for gh in games_out:
        rewards = [float(x) for x in gh.reward_history]
        assert len(rewards) == 41 and rewards[0] == 0.0
        assert all(x in (0.25, 0.75) for x in rewards[1:11]), rewards[:12]
        assert all(x in (0.0, 1.0) for x in rewards[11:])
