"""
SelfPlay.play_rounds behind ONE library call (mzx_selfplay_rounds, csrc/mzx_actor.h; mzx/native_rounds.py) against the
Python round loop (SelfPlay._rounds_batched) on the SAME natively stepped game object, and against the Python game classes:
the same games, field for field (observations with their dtype, actions, rewards, to_play, child_visits, root values bit
for bit), in the same order, slot for slot, and the streams of the bank in the same state afterwards -- one group and two
groups taking turns, ragged game lengths, temperature thresholds, several calls with games in progress in between,
tie-break tape overflows (the retry callback).  CPU test on the serial build of the ABI: the host logic is under test.
"""
import copy

import numpy
import pytest

import hostcheck
from mzx import configs, games, models, self_play, synthetic
from test_selfplay_refill import _same


@pytest.fixture(scope="module")
def backend():
    b = hostcheck.backend()
    games.NativeBatchedGame.backend = b
    return b


def _run(backend, Game, cfg, weights, B, seed, calls, native, pipeline=None):
    c = copy.copy(cfg)
    c.native_rounds = native
    c.self_play_pipeline = pipeline
    shard = self_play.SelfPlay({"weights": weights}, Game, c, seed, num_games=B, _backend=backend)
    out, slots = [], []
    for temperature, kwargs in calls:
        got = shard.play_rounds(temperature, c.temperature_threshold, **kwargs)
        out += got
        slots += shard.finished_slots
        assert len(got) == len(shard.finished_slots)
    states = [shard.bank.get_state(s) for s in range(B)]
    groups = len(shard._live["groups"]) if shard._live else 0
    used_native = bool(shard._live and shard._live.get("native"))
    stats = dict(shard.stats)
    shard.close_game()
    return out, slots, states, groups, used_native, stats


def _assert_equal_runs(a, b):
    (ga, sa, sta, _, _, stats_a), (gb, sb, stb, _, _, stats_b) = a, b
    assert sa == sb and len(ga) == len(gb) and len(ga) > 0
    for k, (x, y) in enumerate(zip(ga, gb)):
        _same(x, y, (k, sa[k]))
        for u, v in zip(x.observation_history, y.observation_history):
            assert numpy.asarray(u).dtype == numpy.asarray(v).dtype
    for x, y in zip(sta, stb):
        assert x[2] == y[2] and (x[1] == y[1]).all() and x[3] == y[3]
    assert stats_a["searches"] == stats_b["searches"] and stats_a["simulations"] == stats_b["simulations"]


CASES = {
    "tictactoe": (lambda: configs.tictactoe(num_simulations=8), "tictactoe"),
    "tictactoe-threshold": (lambda: configs.tictactoe(num_simulations=8, temperature_threshold=3), "tictactoe"),
    "connect4": (lambda: configs.connect4(num_simulations=6, blocks=1, channels=8), "connect4"),
}


@pytest.mark.parametrize("pipeline", [False, True])
@pytest.mark.parametrize("name", sorted(CASES))
def test_native_rounds_equal_the_python_loop_on_board_games(backend, name, pipeline):
    make_cfg, game = CASES[name]
    cfg = make_cfg()
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 21)
    B = 10
    # several calls: one that ends with most games in progress, temperatures that change between calls (a game keeps the
    # one it started under), a bounded number of rounds, then a long one
    calls = [(1.0, dict(min_games=3)), (0.5, dict(max_rounds=4, min_games=1 << 60)), (0.25, dict(min_games=25)), (0.0, dict(min_games=12))]
    native = _run(backend, games.NATIVE[game], cfg, weights, B, 7, calls, True, pipeline)
    python = _run(backend, games.NATIVE[game], cfg, weights, B, 7, calls, False, pipeline)
    plain = _run(backend, games.BATCHED[game], cfg, weights, B, 7, calls, False, pipeline)
    assert native[4] and not python[4] and native[3] == python[3] == (2 if pipeline else 1)
    _assert_equal_runs(native, python)
    _assert_equal_runs(native, plain)          # ... and the Python game class played by the Python loop


@pytest.mark.parametrize("pipeline", [False, True])
def test_native_rounds_on_the_synthetic_game(backend, pipeline):
    """The fixed-shape environment of the metric: fixed-length games (every slot of a group ends in the same round), two
    players, three actions."""
    cfg = configs.cartpole(num_simulations=7, max_moves=5, action_space=list(range(3)), observation_shape=(2, 1, 3),
                           players=list(range(2)))
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 5)
    Native = games.make_native_synthetic_game(cfg.observation_shape, 3, 2)
    Python = synthetic.make_synthetic_batched_game(cfg.observation_shape, 3, 2)
    calls = [(1.0, {}), (1.0, dict(max_rounds=3, min_games=1 << 60)), (0.5, {}), (1.0, dict(min_games=40))]
    native = _run(backend, Native, cfg, weights, 12, 11, calls, True, pipeline)
    python = _run(backend, Python, cfg, weights, 12, 11, calls, False, pipeline)
    assert native[4] and not python[4]
    _assert_equal_runs(native, python)
    assert all(len(g.action_history) == 6 for g in native[0])


def test_three_slot_groups_play_what_the_python_loop_plays(backend):
    """config.self_play_groups: more than two groups take turns (a natively played C2 shard runs as four); the order in which
    games finish is (round, group, slot) in both loops."""
    cfg = configs.tictactoe(num_simulations=8)
    cfg.self_play_groups = 3
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 21)
    calls = [(1.0, dict(min_games=5)), (0.5, dict(max_rounds=3, min_games=1 << 60)), (1.0, dict(min_games=30))]
    native = _run(backend, games.TicTacToeNative, cfg, weights, 11, 7, calls, True, True)
    python = _run(backend, games.TicTacToeBatched, cfg, weights, 11, 7, calls, False, True)
    assert native[3] == python[3] == 3 and native[4]
    _assert_equal_runs(native, python)


def test_native_rounds_retry_searches_that_exhaust_their_tape(backend):
    """All-zero weights: equal priors and values everywhere, a tie at every level of every walk -- the 16-word tape of a
    search is exhausted at once and the library calls back for a longer one.  Same games as the Python loop's retries."""
    import torch

    cfg = configs.cartpole(num_simulations=25, max_moves=4)
    weights = {k: torch.zeros_like(v) if v.dtype.is_floating_point else v
               for k, v in models.MuZeroNetwork(cfg, _backend=backend).state_dict().items()}
    Native = games.make_native_synthetic_game(cfg.observation_shape, len(cfg.action_space), 1)
    calls = [(1.0, {}), (1.0, {})]
    native = _run(backend, Native, cfg, weights, 6, 3, calls, True)
    python = _run(backend, Native, cfg, weights, 6, 3, calls, False)
    _assert_equal_runs(native, python)


def test_play_games_after_native_rounds_starts_over(backend):
    cfg = configs.tictactoe(num_simulations=6)
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 2)
    shard = self_play.SelfPlay({"weights": weights}, games.TicTacToeNative, cfg, 5, num_games=3, _backend=backend)
    shard.play_rounds(1.0, None, max_rounds=2, min_games=1 << 60)
    assert shard._live is not None and shard._live.get("native") is not None
    out = shard.play_games(1.0, None, False, "self", 0)
    assert len(out) == 3 and shard._live is None
    assert all(len(h.action_history) >= 6 for h in out)
    shard.close_game()


@pytest.mark.parametrize("game", ["synthetic", "connect4"])
def test_hand_off_of_native_games_fills_the_per_game_priorities(backend, game):
    """The list play_rounds returns for a natively played shard carries its records (ShardGames.records);
    fill_initial_priorities_many takes them as they are (no per-game discovery) and computes on the device functor -- the
    priorities of every game must be the per-game host function's (= the reference's, tests/test_replay_handoff.py)."""
    import pickle

    from mzx import replay

    if game == "synthetic":
        cfg = configs.cartpole(num_simulations=6, max_moves=7, players=list(range(2)))
        Game = games.make_native_synthetic_game(cfg.observation_shape, len(cfg.action_space), 2)
    else:
        cfg = configs.connect4(num_simulations=6, blocks=1, channels=8)
        Game = games.Connect4Native
    cfg.PER, cfg.PER_alpha, cfg.td_steps = True, 0.5, 5
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 3)
    shard = self_play.SelfPlay({"weights": weights}, Game, cfg, 9, num_games=9, _backend=backend)
    out = shard.play_rounds(1.0, None, min_games=30)
    assert isinstance(out, self_play.ShardGames) and out.records and sum(len(m) for _, _, m in out.records) == len(out)
    twins = pickle.loads(pickle.dumps(list(out)))            # materialised copies: the per-game path
    assert replay.fill_initial_priorities_many(out, cfg, backend=backend) == len(out)
    for a, b in zip(out, twins):
        assert replay.fill_initial_priorities(b, cfg)
        assert a.priorities.dtype == numpy.float32
        assert numpy.array_equal(a.priorities.view(numpy.int32), b.priorities.view(numpy.int32))
        assert a.game_priority == b.game_priority and type(a.game_priority) is type(b.game_priority)
    shard.close_game()


@pytest.mark.parametrize("overlap", [True, False])
def test_continuous_self_play_hands_native_games_off_while_the_next_call_plays(backend, overlap):
    """continuous_self_play on a natively played shard: the hand-off of call k's games (device priorities + save_game) runs
    on the main thread while a worker is inside call k + 1.  The SAME games reach the buffer in the same order with the same
    priorities as with the hand-off strictly in turn; only when they arrive differs (one call later), so the overlapped
    run plays one call more before it sees the stop flag -- and still hands every finished game over."""
    from mzx import shared_storage

    def run(flag):
        cfg = configs.tictactoe(num_simulations=5, training_steps=10, ratio=None, self_play_delay=0)
        cfg.PER, cfg.PER_alpha, cfg.td_steps = True, 0.5, 4
        cfg.self_play_overlap_handoff = flag
        weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 4)
        storage = shared_storage.LocalStorage(weights=weights, training_step=0, terminate=False, num_played_games=0,
                                              num_played_steps=0)

        class Buffer:
            def __init__(self):
                self.games = []

            def save_game(self, game_history, shared_storage=None):
                assert game_history.priorities is not None
                self.games.append(game_history)
                if len(self.games) >= 20:
                    storage.set_info("terminate", True)

        buf = Buffer()
        actor = self_play.SelfPlay({"weights": weights}, games.TicTacToeNative, cfg, 1, num_games=6, _backend=backend)
        actor.continuous_self_play(storage, buf)
        actor.close_game()
        return buf.games, actor.stats["searches"]

    games_a, searches = run(overlap)
    assert len(games_a) >= 20
    if not overlap:
        return
    games_b, _ = run(False)
    n = min(len(games_a), len(games_b))
    assert n >= 20 and len(games_a) >= len(games_b)
    for a, b in zip(games_a[:n], games_b[:n]):
        _same(a, b, "overlapped hand-off")
        assert numpy.array_equal(a.priorities.view(numpy.int32), b.priorities.view(numpy.int32))
