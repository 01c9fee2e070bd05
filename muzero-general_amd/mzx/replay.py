"""
Hand-off of finished games to the replay buffer (first "next" row).

The stock ``ReplayBuffer.save_game`` (/root/reference/replay_buffer.py:33-65) computes the initial
prioritised-replay priorities of a game with a Python double loop -- every position x ``td_steps``
rewards (``compute_target_value``, :230-262): ~10 ms per 500-move game, i.e. tens of seconds for a shard
of thousands of games, far more than playing them.  ``fill_initial_priorities`` computes the SAME numbers
(same binary64 operations in the same order per position; the loop over the reward horizon is a vector
operation over all positions) and stores them on the ``GameHistory``; ``save_game`` then takes its
"priorities already present" branch (:35-37).  The stock buffer and trainer stay untouched.

``Reanalyse`` (section 8f, second row) mirrors the reference worker of the same name
(replay_buffer.py:306-373): same constructor and ``reanalyse(replay_buffer, shared_storage)`` loop, with the
per-game work -- stacked observations of every position, one batched ``initial_inference``, value decode --
on the device (``mzx.observations.stack_history`` + the network kernels + ``mzx_support_to_scalar``).
"""
import time

import numpy
import torch

from . import models, observations


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


def _remote(method, *args, **kwargs):
    """Call an actor method (``.remote`` + ``ray.get``) or a plain method alike."""
    if hasattr(method, "remote"):
        import ray
        return ray.get(method.remote(*args, **kwargs))
    return method(*args, **kwargs)


class Reanalyse:
    """
    replay_buffer.py:306-373 -- refreshes ``reanalysed_predicted_root_values`` of stored games with the
    latest network.  ``reanalyse_game`` is the per-game step (new, reusable); ``reanalyse`` the worker loop.
    """

    def __init__(self, initial_checkpoint, config, _backend=None):
        self.config = config
        # Fix random generator seed (replay_buffer.py:318-319)
        numpy.random.seed(self.config.seed)
        torch.manual_seed(self.config.seed)
        # the network lives on the GPU whatever config.reanalyse_on_gpu says: the engine has no CPU path
        self.model = models.MuZeroNetwork(self.config, _backend=_backend)
        self.model.set_weights(initial_checkpoint["weights"])
        self.model.eval()
        self.num_reanalysed_games = initial_checkpoint["num_reanalysed_games"]

    def reanalyse_game(self, game_history):
        """replay_buffer.py:343-367: float32 array [len(root_values)] of decoded root values under the current weights."""
        n = len(game_history.root_values)
        backend = self.model.backend
        stacked = observations.stack_history(backend, self.config, game_history.observation_history,
                                             game_history.action_history, count=n)
        if n == 0:
            return numpy.zeros((0,), numpy.float32)
        value_logits = self.model.initial_inference(stacked)[0]
        values = models.support_to_scalar(value_logits, self.config.support_size, _backend=backend)
        return torch.squeeze(values).detach().cpu().numpy()

    def reanalyse(self, replay_buffer, shared_storage):
        get = lambda key: _remote(shared_storage.get_info, key)
        while get("num_played_games") < 1:
            time.sleep(0.1)
        while get("training_step") < self.config.training_steps and not get("terminate"):
            self.model.set_weights(get("weights"))
            game_id, game_history, _ = _remote(replay_buffer.sample_game, force_uniform=True)
            # Use the last model to provide a fresher, stable n-step value (See paper appendix Reanalyze)
            if self.config.use_last_model_value:
                game_history.reanalysed_predicted_root_values = self.reanalyse_game(game_history)
            _remote(replay_buffer.update_game_history, game_id, game_history)
            self.num_reanalysed_games += 1
            _remote(shared_storage.set_info, "num_reanalysed_games", self.num_reanalysed_games)
