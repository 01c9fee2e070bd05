"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's initial PER priorities
(/root/reference/replay_buffer.py:39-51 ``save_game`` and :230-262 ``compute_target_value``), plain
Python loops in the reference's own order.  Checker of ``mzx.replay.fill_initial_priorities``; pinned
against the unmodified reference by tests/test_replay_handoff.py::test_against_reference (build container).
"""
import numpy


def compute_target_value(game_history, index, config):
    bootstrap_index = index + config.td_steps
    if bootstrap_index < len(game_history.root_values):
        root_values = (game_history.root_values if game_history.reanalysed_predicted_root_values is None
                       else game_history.reanalysed_predicted_root_values)
        same = game_history.to_play_history[bootstrap_index] == game_history.to_play_history[index]
        last_step_value = root_values[bootstrap_index] if same else -root_values[bootstrap_index]
        value = last_step_value * config.discount ** config.td_steps
    else:
        value = 0
    for i, reward in enumerate(game_history.reward_history[index + 1: bootstrap_index + 1]):
        same = game_history.to_play_history[index] == game_history.to_play_history[index + i]
        value += (reward if same else -reward) * config.discount ** i
    return value


def initial_priorities(game_history, config):
    priorities = []
    for i, root_value in enumerate(game_history.root_values):
        priorities.append(numpy.abs(root_value - compute_target_value(game_history, i, config)) ** config.PER_alpha)
    priorities = numpy.array(priorities, dtype="float32")
    return priorities, numpy.max(priorities)
