"""
Hand-off of finished games to the replay buffer (first "next" row).

The stock ``ReplayBuffer.save_game`` (/root/reference/replay_buffer.py:33-65) computes the initial
prioritised-replay priorities of a game with a Python double loop -- every position x ``td_steps``
rewards (``compute_target_value``, :230-262): ~10 ms per 500-move game, i.e. tens of seconds for a shard
of thousands of games, far more than playing them.  ``fill_initial_priorities`` computes the SAME numbers
(same binary64 operations in the same order per position; the loop over the reward horizon is a vector
operation over all positions) and stores them on the ``GameHistory``; ``save_game`` then takes its
"priorities already present" branch (:35-37).  The stock buffer, trainer and reanalyse stay untouched.
"""
import numpy


def n_step_values(game_history, config):
    """
    compute_target_value (replay_buffer.py:230-262) for every position of a game: float64 array [T].
    Per position the accumulation order is the reference's: bootstrap term first, then the rewards i = 0,
    1, ... each as (+-reward) * discount ** i.
    """
    T = len(game_history.root_values)
    td = int(config.td_steps)
    discount = config.discount
    root_values = (game_history.root_values if game_history.reanalysed_predicted_root_values is None
                   else game_history.reanalysed_predicted_root_values)
    rv = numpy.array([float(v) for v in root_values], dtype=numpy.float64)
    tp = numpy.asarray(game_history.to_play_history)
    rewards = numpy.array([float(r) for r in game_history.reward_history], dtype=numpy.float64)
    n_r = rewards.size
    index = numpy.arange(T)
    # bootstrap: +-root_values[index + td] * discount ** td, or the integer 0
    value = numpy.zeros(T, numpy.float64)
    b = index + td
    has = b < T
    if has.any():
        bi = b[has]
        last = numpy.where(tp[bi] == tp[index[has]], rv[bi], -rv[bi])
        value[has] = last * (discount ** td)
    # rewards: enumerate(reward_history[index + 1 : index + td + 1]) -- truncated at the end of the game
    for i in range(td):
        pos = index + 1 + i
        ok = pos < n_r
        if not ok.any():
            break
        k = index[ok]
        r = rewards[pos[ok]]
        signed = numpy.where(tp[k] == tp[k + i], r, -r)
        value[ok] = value[ok] + signed * (discount ** i)
    return value


def fill_initial_priorities(game_history, config):
    """
    replay_buffer.py:39-51: priorities[i] = |root_value_i - target_value_i| ** PER_alpha (float32) and
    game_priority = max.  No-op (returns False) when PER is off, priorities exist, the game is empty or a
    root value is missing (moves played by an opponent carry None, self_play.py:509-511).
    """
    if not getattr(config, "PER", False) or game_history.priorities is not None:
        return False
    roots = game_history.root_values
    if len(roots) == 0 or any(v is None for v in roots):
        return False
    values = n_step_values(game_history, config)
    alpha = config.PER_alpha
    priorities = [numpy.abs(roots[i] - float(values[i])) ** alpha for i in range(len(roots))]
    game_history.priorities = numpy.array(priorities, dtype="float32")
    game_history.game_priority = numpy.max(game_history.priorities)
    return True
