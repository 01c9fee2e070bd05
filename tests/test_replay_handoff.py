"""
Replay-buffer hand-off (first "next" row): mzx.replay.fill_initial_priorities must give
bit-identical float32 priorities to the reference's save_game / compute_target_value
(replay_buffer.py:39-51, :230-262) -- against the oracle restatement everywhere, against the unmodified
reference when /root/reference is present.
"""
import copy
import types

import numpy
import pytest

from mzx import replay, self_play
from oracle import ref_shim, replay_oracle


def _game(rs, length, players, float_rewards):
    gh = self_play.GameHistory()
    gh.action_history = [0] + [int(a) for a in rs.randint(0, 4, size=length)]
    gh.reward_history = [0] + [float(r) if float_rewards else int(r) for r in
                               (rs.standard_normal(length) if float_rewards else rs.randint(0, 2, size=length))]
    gh.to_play_history = [int(i % players) for i in range(length + 1)]
    gh.root_values = [float(v) for v in rs.standard_normal(length)]
    gh.child_visits = [[0.25] * 4 for _ in range(length)]
    gh.observation_history = [numpy.zeros((1, 1, 1))] * (length + 1)
    return gh


CONFIGS = [dict(td_steps=50, discount=0.997, PER_alpha=0.5), dict(td_steps=9, discount=1, PER_alpha=0.5),
           dict(td_steps=3, discount=0.9, PER_alpha=1), dict(td_steps=200, discount=0.997, PER_alpha=0.7)]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_against_oracle(cfg):
    config = types.SimpleNamespace(PER=True, **cfg)
    rs = numpy.random.RandomState(cfg["td_steps"])
    for length, players, fl in [(1, 1, False), (2, 2, True), (9, 2, False), (57, 1, True), (500, 1, False), (131, 2, True)]:
        gh = _game(rs, length, players, fl)
        if length == 57:
            gh.reanalysed_predicted_root_values = [float(v) for v in rs.standard_normal(length)]
        want, want_max = replay_oracle.initial_priorities(gh, config)
        assert replay.fill_initial_priorities(gh, config)
        assert gh.priorities.dtype == numpy.float32
        assert numpy.array_equal(gh.priorities.view(numpy.int32), want.view(numpy.int32)), (cfg, length)
        assert gh.game_priority == want_max
        assert not replay.fill_initial_priorities(gh, config)      # already present: untouched


def test_skips_when_not_applicable():
    config = types.SimpleNamespace(PER=False, td_steps=5, discount=0.9, PER_alpha=0.5)
    gh = _game(numpy.random.RandomState(0), 6, 1, False)
    assert not replay.fill_initial_priorities(gh, config) and gh.priorities is None
    config.PER = True
    gh.root_values[2] = None                                        # opponent move (self_play.py:509-511)
    assert not replay.fill_initial_priorities(gh, config) and gh.priorities is None


@pytest.mark.reference
def test_against_reference():
    """The unmodified ReplayBuffer.save_game (ray stubbed to a plain class) on copies of the same games."""
    ref_shim.load()
    import replay_buffer as ref_rb
    for cfg in CONFIGS:
        config = types.SimpleNamespace(PER=True, seed=0, replay_buffer_size=10 ** 6, **cfg)
        rb = ref_rb.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
        rs = numpy.random.RandomState(3)
        for length, players, fl in [(1, 1, False), (40, 2, True), (300, 1, False)]:
            gh = _game(rs, length, players, fl)
            theirs = copy.deepcopy(gh)
            rb.save_game(theirs)
            assert replay.fill_initial_priorities(gh, config)
            assert numpy.array_equal(gh.priorities.view(numpy.int32), theirs.priorities.view(numpy.int32))
            assert gh.game_priority == theirs.game_priority
            # and the stock buffer accepts a pre-filled game through its "already present" branch
            rb.save_game(copy.deepcopy(gh))


def _record_views(rs, k, T, players, float_rewards, time_major, holes):
    """k ShardGameHistory views of one record of T-move games (random statistics; ``holes``: unvisited roots + illegal actions)."""
    A = 4
    vis = rs.randint(0, 20, size=(k, T, A)).astype(numpy.int32)
    mask = None
    if holes:
        vis[rs.rand(k, T) < 0.15] = 0                       # roots that were never visited report value 0
        mask = rs.rand(k, T, A) < 0.8
        vis = numpy.where(mask, vis, 0).astype(numpy.int32)
    totals = vis.sum(2)
    ratios = vis / numpy.maximum(totals, 1)[:, :, None]
    vals = rs.standard_normal((k, T))
    rews = numpy.zeros((k, T + 1), numpy.float64 if float_rewards else numpy.int64)
    rews[:, 1:] = rs.standard_normal((k, T)) if float_rewards else rs.randint(0, 2, size=(k, T))
    tps = numpy.tile(numpy.arange(T + 1) % players, (k, 1)).astype(numpy.int64)
    if players == 2:
        tps[rs.rand(k) < 0.5] ^= 1                          # half of the games start with the other player
    acts = rs.randint(0, A, size=(k, T + 1)).astype(numpy.int64)
    obs = rs.rand(k, T + 1, 1, 1, 2).astype(numpy.float32)
    plain = totals > 0 if mask is None else (totals > 0) & mask.all(2)
    arrays = [obs, acts, rews, tps, vis, vals, totals, ratios]
    if time_major:
        arrays = [numpy.ascontiguousarray(numpy.swapaxes(a, 0, 1)) for a in arrays]
        mask = None if mask is None else numpy.ascontiguousarray(numpy.swapaxes(mask, 0, 1))
    record = self_play._ShardRecord(A, *arrays[:4], arrays[4], arrays[5], arrays[6], arrays[7], plain.all(1), mask, time_major=time_major)
    return [self_play.ShardGameHistory(record, j, T) for j in range(k)]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_a_shard_s_games_in_one_pass_equal_the_per_game_function(cfg):
    """fill_initial_priorities_many on the views of shard records == fill_initial_priorities game by game, bit for bit."""
    config = types.SimpleNamespace(PER=True, **cfg)
    rs = numpy.random.RandomState(7 + cfg["td_steps"])
    for k, T, players, fl, tm, holes in [(5, 1, 1, False, True, False), (17, 9, 2, False, True, False), (6, 57, 1, True, False, False),
                                         (9, 42, 2, True, True, True), (3, 300, 1, False, True, False), (4, 12, 2, False, False, True)]:
        state = rs.get_state()
        bulk = _record_views(rs, k, T, players, fl, tm, holes)
        rs.set_state(state)
        single = _record_views(rs, k, T, players, fl, tm, holes)
        extra = _game(rs, 11, 2, True)                       # an ordinary GameHistory rides along
        extra_twin = copy.deepcopy(extra)
        touched = bulk[0]
        touched.root_values                                  # a view whose field was already materialised: per-game path
        assert replay.fill_initial_priorities_many(bulk + [extra], config) == k + 1
        assert replay.fill_initial_priorities(extra_twin, config)
        assert numpy.array_equal(extra.priorities, extra_twin.priorities)
        for a, b in zip(bulk, single):
            assert replay.fill_initial_priorities(b, config)
            assert a.priorities.dtype == numpy.float32 and a.priorities.shape == (T,)
            assert numpy.array_equal(a.priorities.view(numpy.int32), b.priorities.view(numpy.int32)), (cfg, k, T)
            assert type(a.game_priority) is type(b.game_priority) and a.game_priority == b.game_priority
            assert a.root_values == b.root_values and a.reward_history == b.reward_history
        assert replay.fill_initial_priorities_many(bulk, config) == 0      # already present: untouched
    config.PER = False
    assert replay.fill_initial_priorities_many(_record_views(rs, 3, 5, 1, False, True, False), config) == 0


@pytest.mark.parametrize("cfg", CONFIGS)
def test_plain_game_histories_of_equal_length_in_one_pass(cfg):
    """The per-object plugin surface hands out ordinary GameHistory objects: equal lengths are stacked and computed together."""
    config = types.SimpleNamespace(PER=True, **cfg)
    rs = numpy.random.RandomState(3 + cfg["td_steps"])
    games = []
    for length, players, fl, count in [(9, 2, False, 7), (32, 1, False, 12), (32, 1, True, 5), (57, 2, True, 4), (5, 1, False, 2)]:
        games += [_game(rs, length, players, fl) for _ in range(count)]
    games[3].root_values[2] = None                       # an opponent's move: the per-game function skips this game
    games[8].root_values[0] = 0                          # an unvisited root reports the integer 0
    games[20].reanalysed_predicted_root_values = [float(v) for v in rs.standard_normal(32)]
    games[21].priorities = numpy.ones(32, numpy.float32)  # already present: untouched
    rs.shuffle(games)
    twins = copy.deepcopy(games)
    want = sum(bool(replay.fill_initial_priorities(t, config)) for t in twins)
    assert replay.fill_initial_priorities_many(games, config) == want == len(games) - 2
    for a, b in zip(games, twins):
        if b.priorities is None:
            assert a.priorities is None
            continue
        assert a.priorities.dtype == numpy.float32
        assert numpy.array_equal(a.priorities.view(numpy.int32), b.priorities.view(numpy.int32))
        assert a.game_priority == b.game_priority and type(a.game_priority) is type(b.game_priority)


@pytest.mark.parametrize("cfg", CONFIGS)
def test_device_kernel_path_equals_the_per_game_function(cfg):
    """``fill_initial_priorities_many(..., backend=...)``: the groups go through mzx_replay_priorities (csrc/mzx_replay.h; here
    the serial build of the same functor, on the device in tests/test_gpu_parity.py) -- target values bit for bit the
    reference's (binary64), priorities the same float32."""
    import hostcheck

    check_device_priorities(hostcheck.backend(), cfg, exact_pow=True)


def check_device_priorities(backend, cfg, exact_pow):
    config = types.SimpleNamespace(PER=True, **cfg)
    rs = numpy.random.RandomState(11 + cfg["td_steps"])
    for k, T, players, fl, tm, holes in [(5, 1, 1, False, True, False), (17, 9, 2, False, True, False), (6, 57, 1, True, False, False),
                                         (9, 42, 2, True, True, True), (3, 300, 1, False, True, False), (130, 32, 1, False, False, False)]:
        state = rs.get_state()
        bulk = _record_views(rs, k, T, players, fl, tm, holes)
        rs.set_state(state)
        single = _record_views(rs, k, T, players, fl, tm, holes)
        plain = [_game(rs, 21, 2, True) for _ in range(6)]          # ordinary GameHistory objects of one length ride along
        twins = copy.deepcopy(plain)
        assert replay.fill_initial_priorities_many(bulk + plain, config, backend=backend) == k + 6
        for a, b in zip(bulk + plain, single + twins):
            assert replay.fill_initial_priorities(b, config)
            assert a.priorities.dtype == numpy.float32 and a.priorities.shape == b.priorities.shape
            if exact_pow or cfg["PER_alpha"] in (0.5, 1):
                assert numpy.array_equal(a.priorities.view(numpy.int32), b.priorities.view(numpy.int32)), (cfg, k, T)
                assert a.game_priority == b.game_priority
            else:       # the device's pow against libm's: one float32 ulp (csrc/mzx_replay.h)
                assert (numpy.abs(a.priorities.view(numpy.int32).astype(numpy.int64) - b.priorities.view(numpy.int32)) <= 1).all()
            assert type(a.game_priority) is type(b.game_priority)
        # compute_target_value itself, binary64 bit patterns, against the reference restatement
        record, T0 = single[0].__dict__["_view"][0], T
        rv = numpy.where(record.totals[:, :T0] > 0, record.vals[:, :T0], 0.0)
        _, _, targets = replay.device_priorities(backend, rv, record.tps[:, : T0 + 1], record.rews[:, : T0 + 1], config, want_targets=True)
        for j in (0, k - 1):
            want = [replay_oracle.compute_target_value(single[j], i, config) for i in range(T0)]
            assert numpy.array_equal(targets[j].view(numpy.int64), numpy.array(want, numpy.float64).view(numpy.int64)), (cfg, k, T, j)
