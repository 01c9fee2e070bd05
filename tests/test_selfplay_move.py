"""
A self-play move of a shard behind two library calls (include/mzx.h ``mzx_selfplay_search`` / ``mzx_selfplay_select``)
against the separate calls they replace (StreamBank.root_draws, BatchedMCTS._launch, StreamBank.advance, the numpy
statements of SelfPlay._select_actions_bank) and against the reference's own statements on numpy RandomState streams
(self_play.py:222-245 select_action, :473 dirichlet).  CPU, through tests/hostcheck: host logic is under test.
"""
import copy
import types

import numpy
import pytest

import hostcheck
from mzx import configs, games, models, self_play, synthetic


@pytest.fixture(scope="module")
def backend():
    return hostcheck.backend()


def _shard(backend, actions, k, seed, sims=6, players=1):
    cfg = configs.cartpole(num_simulations=sims, action_space=list(range(actions)), players=list(range(players)))
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 2)
    Game = synthetic.make_synthetic_game(cfg.observation_shape, actions, players)
    return cfg, self_play.SelfPlay({"weights": weights}, Game, cfg, seed, num_games=k, _backend=backend)


@pytest.mark.parametrize("with_words", [False, True])
def test_native_select_is_the_reference_select_action(backend, with_words):
    """Mixed temperatures (0, inf, 1, 0.5, 0.25, 0.35) and ragged legal lists in one call; stream positions included."""
    A, k, seed = 7, 12, 300
    cfg, sp = _shard(backend, A, k, seed)
    rs = numpy.random.RandomState(1)
    for trial in range(6):
        visits = rs.randint(0, 40, size=(k, A)).astype(numpy.int32)
        legal_lists = [sorted(rs.choice(A, size=rs.randint(1, A + 1), replace=False).tolist()) for _ in range(k)]
        if trial % 2:
            legal_lists = [list(reversed(l)) for l in legal_lists]          # the game's own order, not sorted
        legal = numpy.full((k, A), -1, numpy.int32)
        for r, l in enumerate(legal_lists):
            legal[r, : len(l)] = l
            visits[r, [a for a in range(A) if a not in l]] = 0
        temps = rs.choice([0.0, float("inf"), 1.0, 0.5, 0.25, 0.35], size=k)
        words = rs.randint(0, 9, size=k).astype(numpy.int32) if with_words else None
        refs = [sp.bank.as_random_state(r) for r in range(k)]
        result = types.SimpleNamespace(visit_counts=visits, legal_actions=legal_lists, shared_legal=None, legal_array=legal,
                                       n_legal=(legal >= 0).sum(1).astype(numpy.int32), streams=numpy.arange(k, dtype=numpy.int32),
                                       pending_words=words)
        got = sp._select_actions_bank(result, list(range(k)), temps)
        assert isinstance(got, list) and result.pending_words is None
        for r in range(k):
            if with_words and words[r]:
                refs[r].randint(0, 2 ** 32, size=int(words[r]), dtype=numpy.uint32)
            node = types.SimpleNamespace(children={a: types.SimpleNamespace(visit_count=int(visits[r, a])) for a in legal_lists[r]})
            want = self_play.SelfPlay._select_action(node, float(temps[r]), refs[r])
            assert int(got[r]) == int(want), (trial, r, temps[r])
            state, ref_state = sp.bank.get_state(r), refs[r].get_state()
            assert state[2] == ref_state[2] and numpy.array_equal(state[1], ref_state[1]), (trial, r)


@pytest.mark.parametrize("game,noise", [("synthetic", True), ("synthetic", False), ("tictactoe", True), ("connect4", True)])
def test_fused_move_equals_the_separate_calls(backend, game, noise):
    """engine.run + _select_actions_bank, fused against unfused: outputs, actions and every stream's state, over moves."""
    B, seed = 9, 70
    if game == "synthetic":
        cfg = configs.cartpole(num_simulations=10, action_space=list(range(3)))
        Batched = synthetic.make_synthetic_batched_game(cfg.observation_shape, 3, 1)
    else:
        small = dict(channels=8, blocks=1) if game == "connect4" else {}
        cfg = configs.BY_NAME[game](num_simulations=7, **small)
        Batched = games.BATCHED[game]
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 9)
    a = self_play.SelfPlay({"weights": weights}, Batched, cfg, seed, num_games=B, _backend=backend)
    b = self_play.SelfPlay({"weights": weights}, Batched, cfg, seed, num_games=B, _backend=backend)
    b.engine.fused_move = False
    obs_a, obs_b = a.batched_game.reset(), b.batched_game.reset()
    everyone = numpy.arange(B)
    alive = numpy.ones(B, bool)
    for move in range(5):
        temps = [1.0, numpy.array([1.0, 0.5, 0.0] * 3), 0.25, float("inf"), 0.0][move]
        ra = a.engine.run(obs_a, a.batched_game.legal_actions(), a.batched_game.to_play(), noise, (a.bank, everyone),
                          _defer_advance=True)
        rb = b.engine.run(obs_b, b.batched_game.legal_actions(), b.batched_game.to_play(), noise, (b.bank, everyone))
        assert ra.pending_words is not None and rb.pending_words is None
        assert numpy.array_equal(ra.visit_counts, rb.visit_counts)
        assert numpy.array_equal(ra.root_values.view(numpy.int64), rb.root_values.view(numpy.int64))
        assert numpy.array_equal(ra.root_predicted_values.view(numpy.int64), rb.root_predicted_values.view(numpy.int64))
        assert numpy.array_equal(ra.max_tree_depth, rb.max_tree_depth) and numpy.array_equal(ra.tape_used, rb.tape_used)
        act_a = numpy.asarray(a._select_actions_bank(ra, everyone, temps))
        act_b = numpy.asarray(b._select_actions_bank(rb, everyone, temps))
        assert numpy.array_equal(act_a, act_b), move
        for i in range(B):
            sa, sb = a.bank.get_state(i), b.bank.get_state(i)
            assert sa[2:] == sb[2:] and numpy.array_equal(sa[1], sb[1]), (move, i)
        obs_a, _, done_a = a.batched_game.step(act_a, alive.copy())
        obs_b, _, _ = b.batched_game.step(act_b, alive.copy())
        if numpy.asarray(done_a).any():
            break


def test_fused_move_reports_illegal_lists_like_the_reference(backend):
    cfg, sp = _shard(backend, 4, 3, 5)
    obs = numpy.zeros((3,) + tuple(cfg.observation_shape), numpy.float32)
    everyone = numpy.arange(3)
    bad = numpy.array([[0, 1, -1, -1], [-1, -1, -1, -1], [0, 1, 2, 3]], numpy.int32)
    with pytest.raises(AssertionError, match="should not be an empty array"):
        sp.engine.run(obs, bad, [0, 0, 0], True, (sp.bank, everyone))
    for rows in ([[0, 7, -1, -1]] * 3, [[0, -1, 2, -1]] * 3):
        with pytest.raises(AssertionError, match="subset of the action space"):
            sp.engine.run(obs, numpy.array(rows, numpy.int32), [0, 0, 0], True, (sp.bank, everyone))


def test_batched_shard_two_groups_against_one_and_against_the_separate_calls(backend):
    """The CPU twin of tests/test_gpu_parity.py::test_batched_shard_of_4096_cartpole_games_...: 48 games, 3 rounds of games."""
    import copy

    cfg = configs.cartpole(num_simulations=9, max_moves=5)
    Batched = synthetic.make_synthetic_batched_game(cfg.observation_shape, len(cfg.action_space), len(cfg.players))
    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 4)
    B, seed = 48, 1000

    def play(pipeline, fused):
        c = copy.copy(cfg)
        c.self_play_pipeline = pipeline
        sp = self_play.SelfPlay({"weights": weights}, Batched, c, seed, num_games=B, _backend=backend)
        sp.engine.fused_move = fused
        out, slots = [], []
        for _ in range(3):
            out += sp.play_rounds(1.0, None)
            slots += sp.finished_slots
        groups = len(sp._live["groups"])
        sp.close_game()
        return out, slots, groups

    a, slots_a, groups_a = play(True, True)
    b, slots_b, groups_b = play(None, True)        # (48 games: below the automatic threshold)
    c, slots_c, _ = play(False, False)
    assert groups_a == 2 and groups_b == 1 and len(a) == len(b) == len(c) == 3 * B
    assert slots_a == slots_b == slots_c
    for k in range(3 * B):
        for other in (b, c):
            assert a[k].action_history == other[k].action_history, k
            assert a[k].child_visits == other[k].child_visits and a[k].reward_history == other[k].reward_history, k
            assert numpy.array_equal(numpy.array(a[k].root_values).view(numpy.int64), numpy.array(other[k].root_values).view(numpy.int64)), k
            assert all(numpy.array_equal(x, y) for x, y in zip(a[k].observation_history, other[k].observation_history)), k


def test_move_entry_points_reject_bad_arguments(backend):
    """Error behaviour of the two entry points through the C ABI: codes + messages, nothing drawn on a rejected call."""
    import ctypes

    from mzx import _lib

    cfg, sp = _shard(backend, 4, 3, 5)
    lib, bank, engine = backend.lib, sp.bank, sp.engine
    B, A = 3, 4
    obs = numpy.zeros((B, 4), numpy.float32)
    legal = numpy.tile(numpy.arange(A, dtype=numpy.int32), (B, 1))
    to_play = numpy.zeros(B, numpy.int32)
    streams = numpy.arange(B, dtype=numpy.int32)
    before = [bank.get_state(i) for i in range(B)]
    outputs = engine._move_search(B, obs, legal, to_play, bank, streams, True)      # a valid move: fills the cached struct
    outputs()
    for i in range(B):
        bank.set_state(i, before[i])
    st = engine._staging(B, self_play.TAPE_WORDS, 4, True)
    good = st["move"]
    n_legal = numpy.empty(B, numpy.int32)
    arena = engine.arena(B)

    def call(mv, n=n_legal, bank_handle=bank.handle):
        return lib.mzx_selfplay_search(engine.handle(B, self_play.TAPE_WORDS), bank_handle, ctypes.byref(mv),
                                       None if n is None else n.ctypes.data, backend.ptr(arena), arena.numel(), backend.stream())

    def variant(**fields):
        mv = _lib.Move()
        ctypes.memmove(ctypes.byref(mv), ctypes.byref(good), ctypes.sizeof(_lib.Move))
        for k, v in fields.items():
            setattr(mv, k, v)
        return mv

    assert call(good) == 0
    for i in range(B):
        bank.set_state(i, before[i])
    assert call(variant(num_games=0)) != 0 and b"sizes" in lib.mzx_last_error()
    assert call(variant(legal_actions=None)) != 0 and b"legal_actions" in lib.mzx_last_error()
    assert call(good, n=None) != 0 and b"missing buffer" in lib.mzx_last_error()
    assert call(variant(h_in=None)) != 0 and b"missing buffer" in lib.mzx_last_error()
    bad_streams = numpy.array([0, 1, 99], numpy.int32)
    assert call(variant(streams=bad_streams.ctypes.data)) != 0 and b"out of range" in lib.mzx_last_error()
    outside = variant()
    outside.io.d_tape = 12345                      # a field of the staged block that does not lie inside it
    assert call(outside) != 0 and b"inside the staged input block" in lib.mzx_last_error()
    no_noise = variant(add_exploration_noise=0)    # io.d_noise still set: contradiction
    assert call(no_noise) != 0 and b"inside the staged input block" in lib.mzx_last_error()
    for i in range(B):                             # nothing was drawn by any rejected call
        state = bank.get_state(i)
        assert state[2:] == before[i][2:] and numpy.array_equal(state[1], before[i][1]), i
    # select
    visits = numpy.array([[3, 1, 0, 2]] * B, numpy.int32)
    temps = numpy.ones(B)
    table, table_t = numpy.arange(8, dtype="int32") ** 1.0, numpy.array([1.0])
    act = numpy.empty(B, numpy.int64)
    nl = numpy.full(B, A, numpy.int32)

    def select(mv, n=nl, vis=visits, t=temps, tab=table, stride=8, tt=table_t, nt=1):
        return lib.mzx_selfplay_select(bank.handle, ctypes.byref(mv), n.ctypes.data, None, vis.ctypes.data, t.ctypes.data,
                                       None if tab is None else tab.ctypes.data, stride, None if tt is None else tt.ctypes.data, nt,
                                       act.ctypes.data)

    assert select(good) == 0 and set(act.tolist()) <= set(range(A))
    assert select(good, t=numpy.full(B, 0.5)) != 0 and b"no power table" in lib.mzx_last_error()
    assert select(good, vis=numpy.array([[9, 1, 0, 2]] * B, numpy.int32)) != 0 and b"outside the power table" in lib.mzx_last_error()
    assert select(good, n=numpy.array([4, 0, 4], numpy.int32)) != 0 and b"n_legal" in lib.mzx_last_error()
    assert select(good, tab=None) != 0 and b"missing argument" in lib.mzx_last_error()
    assert select(good, t=numpy.zeros(B), tab=None, tt=None, nt=0) == 0 and act.tolist() == [0, 0, 0]      # arg-max needs no table


def test_ring_log_keeps_rewards_a_plugin_returns_as_ints_first_and_floats_later(backend):
    """The log's reward rows take the wider dtype when a later round brings one (numpy.stack of round 4 promoted too)."""
    cfg = configs.cartpole(num_simulations=5, max_moves=6)
    Base = synthetic.make_synthetic_batched_game(cfg.observation_shape, len(cfg.action_space), 1)

    class Mixed(Base):
        def step(self, actions, active=None):
            obs, reward, done = super().step(actions, active)
            if int(self.t[0]) >= 3:
                reward = reward * 0.5            # floats from the third move on
            return obs, reward, done

    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 4)
    sp = self_play.SelfPlay({"weights": weights}, Mixed, cfg, 9, num_games=4, _backend=backend)
    games_played = sp.play_rounds(1.0, None)
    assert len(games_played) == 4
    halves = 0
    for gh in games_played:
        assert len(gh.reward_history) == 7 and gh.reward_history[0] == 0
        assert all(float(x) in (0.0, 1.0) for x in gh.reward_history[1:3])
        assert all(float(x) in (0.0, 0.5) for x in gh.reward_history[3:])
        halves += sum(1 for x in gh.reward_history[3:] if float(x) == 0.5)
    assert halves > 0


@pytest.mark.parametrize("seed", range(8))
def test_fused_move_equals_the_separate_calls_on_random_shards(backend, seed):
    """
    Random shard shapes (1 - 40 games, 2 - 9 actions, one / two players), RAGGED legal lists that change every round
    (a batched game whose legal_actions() offers a random subset in a random order), random per-game temperatures
    (0, inf, 1, 0.5, 0.25, 0.7) that change every round, noise on / off: play_rounds through the two library calls
    against play_rounds through the separate calls -- the same games, slot by slot.
    """
    rs = numpy.random.RandomState(1000 + seed)
    A, players, B = int(rs.randint(2, 10)), int(rs.randint(1, 3)), int(rs.randint(1, 41))
    cfg = configs.cartpole(num_simulations=int(rs.randint(3, 12)), max_moves=int(rs.randint(3, 9)), action_space=list(range(A)),
                           players=list(range(players)), temperature_threshold=[None, 2, 4][seed % 3])
    Base = synthetic.make_synthetic_batched_game(cfg.observation_shape, A, players)

    class Ragged(Base):
        def legal_actions(self):
            out = numpy.full((self.num_games, A), -1, numpy.int32)
            for i in range(self.num_games):           # a function of (game seed, move): the same for both actors
                r = numpy.random.RandomState(int(self.seeds[i]) * 131 + int(self.t[i]))
                acts = r.permutation(A)[: r.randint(1, A + 1)]
                out[i, : acts.size] = acts
            return out

    weights = synthetic.fill_state_dict(models.MuZeroNetwork(cfg, _backend=backend).state_dict(), 40 + seed)
    temperature = [1.0, 0.5, 0.25, 0.7, 0.0, float("inf")][seed % 6]
    played = []
    for fused, pipeline in ((True, True), (False, False)):
        c = copy.copy(cfg)
        c.self_play_pipeline = pipeline and B > 1
        sp = self_play.SelfPlay({"weights": weights}, Ragged, c, 500 + seed, num_games=B, _backend=backend)
        sp.engine.fused_move = fused
        by_slot = {}
        for _ in range(3):
            for gh, slot in zip(sp.play_rounds(temperature, c.temperature_threshold), sp.finished_slots):
                by_slot.setdefault(slot, []).append(gh)
        played.append(by_slot)
        sp.close_game()
    a, b = played
    assert a.keys() == b.keys() and len(a) == B
    for slot in a:
        assert len(a[slot]) == len(b[slot]) >= 3
        for x, y in zip(a[slot], b[slot]):
            assert x.action_history == y.action_history and x.to_play_history == y.to_play_history, (seed, slot)
            assert x.child_visits == y.child_visits and x.reward_history == y.reward_history, (seed, slot)
            assert numpy.array_equal(numpy.array(x.root_values, numpy.float64).view(numpy.int64),
                                     numpy.array(y.root_values, numpy.float64).view(numpy.int64)), (seed, slot)
