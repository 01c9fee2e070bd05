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
from .history import gc_paused


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


def device_priorities(backend, root_values, to_play, rewards, config, want_targets=False):
    """
    ``mzx_replay_priorities`` (include/mzx.h, csrc/mzx_replay.h): the initial priorities of G games of T positions each
    ON THE DEVICE -- root_values [G][T] binary64 (0 for an unvisited root), to_play / rewards [G][T + 1] as the histories
    hold them.  Returns (priorities float32 [G][T], game_priority float32 [G][, targets binary64 [G][T]]) as host arrays.
    The target values are the reference's bit for bit (binary64 multiply / add in its order, ``discount ** i`` from a
    table filled HERE with Python's own float pow -- the reference's expression, replay_buffer.py:247, :260); the final
    ``** PER_alpha`` is a binary64 sqrt for 0.5 (every shipped configuration), the identity for 1.
    """
    lib, dev = backend.lib, backend.device
    rv = numpy.ascontiguousarray(root_values, dtype=numpy.float64)
    G, T = rv.shape
    td = int(config.td_steps)
    up = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)
    d_rv = up(rv)
    d_rw = up(numpy.ascontiguousarray(rewards, dtype=numpy.float64))
    d_tp = up(numpy.ascontiguousarray(to_play, dtype=numpy.int32))
    d_pw = up(numpy.array([config.discount ** i for i in range(td + 1)], dtype=numpy.float64))
    assert d_rw.shape == (G, T + 1) and d_tp.shape == (G, T + 1)
    d_pri, d_top = backend.empty((G, T), torch.float32), backend.empty((G,), torch.float32)
    d_tg = backend.empty((G, T), torch.float64) if want_targets else None
    lib.check(lib.mzx_replay_priorities(backend.ptr(d_rv), backend.ptr(d_rw), backend.ptr(d_tp), G, T, td, backend.ptr(d_pw),
                                        float(config.PER_alpha), backend.ptr(d_tg), backend.ptr(d_pri), backend.ptr(d_top),
                                        backend.stream()))
    out = (d_pri.cpu().numpy(), d_top.cpu().numpy())
    return out + (d_tg.cpu().numpy(),) if want_targets else out


def fill_initial_priorities_many(histories, config, backend=None):
    """
    ``fill_initial_priorities`` for the games a self-play shard hands out together, in ONE pass per group of games of the
    same length: the loop over the reward horizon (``td_steps`` iterations of a dozen numpy statements) runs once per
    group instead of once per game -- 4096 cartpole games cost one game's worth of interpreter time (per game it was
    0.4 ms, twenty times what playing the game's 32 moves costs).  Groups: games that are still VIEWS of one shard record
    (``mzx.self_play.ShardGameHistory``, the batched protocol: the record's arrays are used as they are), and ordinary
    ``GameHistory`` objects of equal length (the per-object plugin surface: their lists are stacked).  Same binary64
    operations in the same order per position, the power through the same scalar ``pow``, the same float32 rounding:
    bit-identical to the per-game function (tests/test_replay_handoff.py).  Games the per-game function would skip
    (priorities present, a missing root value) or treat differently (reanalysed values) go through it one by one.
    Returns the number of games that got priorities.

    ``backend`` (a ``mzx._lib.Backend``: what ``SelfPlay.continuous_self_play`` passes): the groups are computed ON THE
    DEVICE (``device_priorities``: one upload of a group's root values / rewards / to_play, one kernel, one download) --
    the horizon loop and the power leave the interpreter altogether; same float32 priorities
    (tests/test_replay_handoff.py on the serial build, tests/test_gpu_parity.py on the device against the reference's
    own save_game outputs).
    """
    if not getattr(config, "PER", False):
        return 0
    filled, views, plain = 0, {}, {}
    jobs = []
    grouped = getattr(histories, "records", None)
    if (grouped and all(record.priorities is not None for record, _, _ in grouped)
            and sum(len(members) for _, _, members in grouped) == len(histories)):
        return 0        # (every record came with its priorities: the views were created with their rows)
    if grouped and backend is not None and all(len(h.__dict__) == 2 for _, _, members in grouped for h in members) and all(
            record.priorities is None for record, _, _ in grouped):
        # the shard's own grouping (mzx.self_play.ShardGames): fresh views of whole records, in record order (nothing of them
        # materialised or assigned yet) -- the record's arrays as they lie, the result stored ON the record: a view
        # resolves its row of it on first access (ShardGameHistory._PER); nothing is done per game
        for record, T, members in grouped:
            if T > 0:
                k = len(members)
                rv = numpy.where(record.totals[:k, :T] > 0, record.vals[:k, :T], 0.0)        # root.value() or 0
                record.priorities, record.game_priority = device_priorities(backend, rv, record.tps[:k, : T + 1],
                                                                            record.rews[:k, : T + 1], config)
                filled += k
        if sum(len(m) for _, _, m in grouped) == len(histories):
            histories = ()
        else:
            done = {id(h) for _, _, members in grouped for h in members}
            histories = [h for h in histories if id(h) not in done]
    for h in histories:
        if h.priorities is not None:
            continue
        view = h.__dict__.get("_view")
        if h.reanalysed_predicted_root_values is not None:
            filled += bool(fill_initial_priorities(h, config))
        elif view is not None and view[2] > 0 and not any(name in h.__dict__ for name in ("root_values", "reward_history",
                                                                                           "to_play_history")):
            views.setdefault((id(view[0]), view[2]), (view[0], view[2], []))[2].append((h, view[1]))
        else:
            roots = h.root_values
            if len(roots) == 0 or len(h.reward_history) != len(roots) + 1 or len(h.to_play_history) != len(roots) + 1:
                filled += bool(fill_initial_priorities(h, config))      # (empty, or not a finished game's shape)
            else:
                plain.setdefault(len(roots), []).append(h)
    for record, T, members in views.values():
        rows = numpy.array([i for _, i in members])
        rv = numpy.where(record.totals[rows, :T] > 0, record.vals[rows, :T], 0.0).astype(numpy.float64)   # root.value() or 0
        jobs.append((rv, numpy.asarray(record.tps[rows, : T + 1]), numpy.asarray(record.rews[rows, : T + 1]).astype(numpy.float64),
                     [h for h, _ in members]))
    for T, members in plain.items():
        if len(members) < 4:       # (nothing to share)
            for h in members:
                filled += bool(fill_initial_priorities(h, config))
            continue
        with gc_paused():
            roots = [h.root_values for h in members]
            if any(v is None for row in roots for v in row):      # opponent moves carry None (self_play.py:509-511)
                keep = [h for h, row in zip(members, roots) if not any(v is None for v in row)]
                roots = [h.root_values for h in keep]
                members = keep
            if not members:
                continue
            jobs.append((numpy.array(roots, dtype=numpy.float64), numpy.array([h.to_play_history for h in members]),
                         numpy.array([h.reward_history for h in members], dtype=numpy.float64), members))
    td, discount, alpha = int(config.td_steps), config.discount, config.PER_alpha
    for rv, tp, rewards, members in jobs:
        k, T = rv.shape
        if backend is not None:
            priorities, top = device_priorities(backend, rv, tp, rewards, config)
            with gc_paused():
                for h, p, t in zip(members, priorities, top):      # (rows of the downloaded array: nothing else refers to it)
                    d = h.__dict__
                    d["priorities"] = p
                    d["game_priority"] = t
            filled += k
            continue
        value = numpy.zeros((k, T), numpy.float64)
        m = T - td
        if m > 0:          # bootstrap: +-root_values[index + td] * discount ** td
            value[:, :m] = numpy.where(tp[:, td:td + m] == tp[:, :m], rv[:, td:td + m], -rv[:, td:td + m]) * (discount ** td)
        same, term = numpy.empty((k, T), bool), numpy.empty((k, T), numpy.float64)
        for i in range(td):      # rewards index + 1 + i, truncated at the end of the game
            m = T - i
            if m <= 0:
                break
            # value[:m] + where(to_play[k] == to_play[k + i], r, -r) * discount ** i, in two reused buffers (a dozen
            # half-megabyte temporaries per iteration cost more in page faults than the arithmetic)
            r, sg = rewards[:, 1 + i: 1 + i + m], term[:, :m]
            numpy.equal(tp[:, :m], tp[:, i:i + m], out=same[:, :m])
            numpy.negative(r, out=sg)
            numpy.copyto(sg, r, where=same[:, :m])
            sg *= (discount ** i)
            value[:, :m] += sg
        with gc_paused():
            gaps = numpy.abs(rv - value).ravel().tolist()
            priorities = numpy.array([g ** alpha for g in gaps], dtype="float32").reshape(k, T)   # the scalar pow, as :44
            top = priorities.max(axis=1)
            for j, h in enumerate(members):
                h.priorities = priorities[j].copy()
                h.game_priority = top[j]
        filled += k
    return filled


def _stock_replay_buffer_class():
    """
    The user's own ``ReplayBuffer`` (the reference's replay_buffer.py:11-303, importable wherever its trainer runs):
    storage, eviction and sampling stay ITS code.  Under Ray the module attribute is an ActorClass; the plain class
    behind it is what gets instantiated here (wrap the accelerated buffer with ``ray.remote`` like the reference).
    """
    import importlib
    try:
        module = importlib.import_module("replay_buffer")
    except ModuleNotFoundError as e:
        raise ModuleNotFoundError(
            "mzx.replay.ReplayBuffer wraps the reference's own buffer: module 'replay_buffer' (replay_buffer.py of "
            "muzero-general) must be importable -- put the reference checkout on sys.path, or pass the class / an "
            "instance as ReplayBuffer(..., stock=...)") from e
    return _plain_class(module.ReplayBuffer)


def _plain_class(cls):
    """The class behind a ``ray.remote`` wrapper (``ActorClass.__ray_metadata__.modified_class``), else ``cls`` itself."""
    meta = getattr(cls, "__ray_metadata__", None)
    return getattr(meta, "modified_class", cls)


class ReplayBuffer:
    """
    The stock replay buffer with its two per-element Python loops of the hand-off replaced -- by COMPOSITION: an
    instance of the reference's own ``ReplayBuffer`` (replay_buffer.py:11-303; pass the class or an instance as
    ``stock``, default: ``import replay_buffer``) keeps storage, eviction, priority feedback and every sampling draw;
    this class adds

      * ``save_game``: initial PER priorities through ``fill_initial_priorities`` before the stock ``save_game``,
        which then takes its "priorities already present" branch (:35-37);
      * ``get_batch`` / ``make_target`` (:70-138, :264-303): the reference evaluates, per sample and unroll step,
        ``compute_target_value`` with its own ``td_steps`` loop -- batch x (unroll + 1) x td_steps interpreter
        iterations per training step.  Here the n-step values of every position of a sampled game are one
        vectorised pass (``n_step_values``, same binary64 operations in the same order per position, cached per
        game until its root values are reanalysed or the game leaves the buffer), and a batch is gathers from
        per-game arrays.

    The numpy draws happen in the reference's order -- the stock ``sample_n_games`` (:166-184), then per sample its
    position (:193-202) followed by the random actions of its absorbing steps (:301) -- so with the same seed and
    buffer the batches are IDENTICAL to the reference's, element for element (tests/test_replay_batch.py runs the
    two side by side).  The tensors come back as numpy arrays instead of nested lists (``trainer.py:55-75`` feeds
    them to ``torch.tensor`` either way).  Attributes (``buffer``, ``num_played_games``, ``total_samples``, ...)
    read through to the stock object.
    """

    def __init__(self, initial_checkpoint, initial_buffer, config, stock=None):
        factory = _plain_class(stock) if stock is not None else _stock_replay_buffer_class()
        # a class (also the one behind a ray.remote ActorClass) is instantiated; anything that already has the buffer's
        # methods is taken as the instance to wrap
        built = factory(initial_checkpoint, initial_buffer, config) if isinstance(factory, type) else factory
        object.__setattr__(self, "_stock", built)
        object.__setattr__(self, "_arrays", {})   # game_id -> (game_history, per-game numpy views); dropped when the game changes or leaves

    def __getattr__(self, name):
        if name in ("_stock", "_arrays"):
            raise AttributeError(name)
        return getattr(self._stock, name)

    def __setattr__(self, name, value):
        # the buffer's state lives in the stock object: writes to its public attributes (buffer, num_played_games, ...)
        # land there, like the reads above
        if name.startswith("_"):
            object.__setattr__(self, name, value)
        else:
            setattr(self._stock, name, value)

    # ---- storage / sampling: the stock buffer's own code (explicit so that ray.remote sees the methods)
    def get_buffer(self):
        return self._stock.get_buffer()

    def update_priorities(self, priorities, index_info):
        return self._stock.update_priorities(priorities, index_info)

    def sample_game(self, force_uniform=False):
        return self._stock.sample_game(force_uniform)

    def sample_n_games(self, n_games, force_uniform=False):
        return self._stock.sample_n_games(n_games, force_uniform)

    def sample_position(self, game_history, force_uniform=False):
        return self._stock.sample_position(game_history, force_uniform)

    def compute_target_value(self, game_history, index):
        return self._stock.compute_target_value(game_history, index)

    def save_game(self, game_history, shared_storage=None):
        fill_initial_priorities(game_history, self._stock.config)    # no-op for games it does not cover
        out = self._stock.save_game(game_history, shared_storage)
        if self._arrays and self._stock.buffer:                      # evicted games take their cached arrays along
            oldest = next(iter(self._stock.buffer))                  # game ids only grow (replay_buffer.py:53-62)
            for game_id in [g for g in self._arrays if g < oldest]:
                del self._arrays[game_id]
        return out

    def update_game_history(self, game_id, game_history):
        self._arrays.pop(game_id, None)          # reanalysed root values change the n-step targets
        return self._stock.update_game_history(game_id, game_history)

    # ---- position draw
    def _sample_position_fast(self, game_history):
        """
        sample_position (:186-202) without the interpreter loop of ``sum(priorities)`` and without
        ``numpy.random.choice``'s argument checks -- same numbers, same single draw: Python's ``sum`` over a float32
        array is the sequential float32 accumulation ``cumsum`` performs; ``choice(n, p=p)`` widens p to binary64,
        takes ``cdf = p.cumsum(); cdf /= cdf[-1]`` and bisects one ``random_sample()`` from the right.
        """
        if not self._stock.config.PER:
            return numpy.random.randint(0, len(game_history.root_values)), None      # choice(n) == randint(0, n)
        pr = game_history.priorities
        if pr.dtype != numpy.float32 or pr.ndim != 1 or pr.size == 0:
            return self._stock.sample_position(game_history)
        position_probs = pr / numpy.cumsum(pr, dtype=numpy.float32)[-1]
        cdf = position_probs.astype(numpy.float64).cumsum()
        cdf /= cdf[-1]
        position_index = int(cdf.searchsorted(numpy.random.random_sample(), side="right"))
        return position_index, position_probs[position_index]

    # ---- targets (replay_buffer.py:264-303 as gathers)
    def _build_arrays(self, game_history):
        """Per-game arrays a batch gathers from: n-step values [T], rewards / actions [T + 1], child visits [T][A]."""
        T = len(game_history.root_values)
        return dict(
            T=T,
            values=n_step_values(game_history, self._stock.config) if T else numpy.zeros(0),
            rewards=numpy.array([float(r) for r in game_history.reward_history], dtype=numpy.float64),
            actions=numpy.array([int(a) for a in game_history.action_history], dtype=numpy.int64),
            visits=numpy.array(game_history.child_visits, dtype=numpy.float64).reshape(T, -1),
        )

    def _game_arrays(self, game_id, game_history):
        hit = self._arrays.get(game_id)
        if hit is not None and hit[0] is game_history:
            return hit[1]
        arrays = self._build_arrays(game_history)
        self._arrays[game_id] = (game_history, arrays)
        return arrays

    def make_target(self, game_history, state_index, _arrays=None):
        """replay_buffer.py:264-303 for one position: (values, rewards, policies, actions) as arrays.  Without the
        arrays of a buffered game (``get_batch`` passes them) they are built on the fly, nothing is cached."""
        cfg = self._stock.config
        g = _arrays if _arrays is not None else self._build_arrays(game_history)
        U, T = cfg.num_unroll_steps, g["T"]
        A = g["visits"].shape[1] if T else len(cfg.action_space)
        idx = state_index + numpy.arange(U + 1)
        inside, at_end = idx < T, idx == T
        safe = numpy.minimum(idx, max(T - 1, 0))
        values = numpy.where(inside, g["values"][safe] if T else 0.0, 0.0)
        rewards = numpy.where(inside | at_end, g["rewards"][numpy.minimum(idx, T)], 0.0)
        policies = numpy.where(inside[:, None], g["visits"][safe] if T else 0.0, 1 / A)
        actions = g["actions"][numpy.minimum(idx, T)].copy()
        if idx[-1] > T:                          # States past the end of games are treated as absorbing states
            space = cfg.action_space
            for k in range(max(0, T + 1 - state_index), U + 1):
                actions[k] = space[numpy.random.randint(0, len(space))]    # == numpy.random.choice(space), same draw
        return values, rewards, policies, actions

    def get_batch(self):
        cfg = self._stock.config
        U, A = cfg.num_unroll_steps, len(cfg.action_space)
        n = cfg.batch_size
        total_samples = self._stock.total_samples
        index_batch, observation_batch = [], []
        action_batch = numpy.empty((n, U + 1), numpy.int64)
        value_batch = numpy.empty((n, U + 1), numpy.float64)
        reward_batch = numpy.empty((n, U + 1), numpy.float64)
        policy_batch = numpy.empty((n, U + 1, A), numpy.float64)
        gradient_scale_batch = numpy.empty((n, U + 1), numpy.int64)
        weight_batch = [] if cfg.PER else None
        for i, (game_id, game_history, game_prob) in enumerate(self._stock.sample_n_games(n)):
            game_pos, pos_prob = self._sample_position_fast(game_history)
            values, rewards, policies, actions = self.make_target(game_history, game_pos,
                                                                  self._game_arrays(game_id, game_history))
            index_batch.append([game_id, game_pos])
            observation_batch.append(game_history.get_stacked_observations(game_pos, cfg.stacked_observations, A))
            action_batch[i], value_batch[i], reward_batch[i], policy_batch[i] = actions, values, rewards, policies
            gradient_scale_batch[i] = min(U, len(game_history.action_history) - game_pos)
            if cfg.PER:
                weight_batch.append(1 / (total_samples * game_prob * pos_prob))
        if cfg.PER:
            weight_batch = numpy.array(weight_batch, dtype="float32") / max(weight_batch)
        # observation_batch: batch, channels, height, width; action / value / reward / gradient_scale: batch,
        # num_unroll_steps + 1; policy_batch: batch, num_unroll_steps + 1, len(action_space); weight_batch: batch
        return (index_batch, (observation_batch, action_batch, value_batch, reward_batch, policy_batch, weight_batch,
                              gradient_scale_batch))


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
