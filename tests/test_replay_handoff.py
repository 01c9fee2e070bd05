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
