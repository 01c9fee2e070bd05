"""
``SelfPlay.play_rounds`` for a shard of NATIVELY stepped games: the whole round loop behind one library call.

Reference: the actor loop and ``play_game`` of /root/reference/self_play.py:31-52, :110-183 -- per move of one game:
MCTS.run, select_action, Game.step, GameHistory appends; the next game starts the moment one ends.  ``mzx.self_play``
plays B such actors as one shard; through a Python plugin game (batched protocol) a round costs ~75 interpreter
statements per slot group around two library calls, which holds BASELINE config C2 (a 0.2 ms search of 4096 trees) at a
third of its search rate.  When the game object steps inside the library (``mzx.games.NativeBatchedGame``:
``native_handle``) the rounds run in ``mzx_selfplay_rounds`` (include/mzx.h, csrc/mzx_actor.h) -- searches, action draws,
steps, history rows, finished games copied out, slots refilled, two slot groups taking turns on the GPU -- and this module
is what remains on the Python side: building the actors once, one call per ``play_rounds``, and wrapping the finished
games it hands back (game-major arrays) into the same ``ShardGameHistory`` views the Python loop produces.

Same games, field for field, in the same order as ``SelfPlay._rounds_batched`` on the same game object
(``config.native_rounds = False`` keeps that loop: the A/B and the parity reference, tests/test_native_rounds.py).
Stacked observations (the device frame store) and Python plugin games stay on the Python loop.
"""
import ctypes

import numpy

from . import _lib
from .history import ShardGameHistory, ShardGames, _ShardRecord, gc_paused
from .search import TAPE_WORDS, BatchedMCTS


def usable(actor):
    """Whether ``actor``'s shard can play its rounds natively."""
    game = actor.batched_game
    return (game is not None and getattr(game, "native_handle", None) and actor.bank is not None
            and actor.config.stacked_observations == 0 and getattr(actor.config, "native_rounds", True)
            and actor.engine.fused_move and len(actor.config.action_space) <= 4096)


class _Group:
    def __init__(self, actor, game, first, n, engine, temperature):
        cfg, lib = actor.config, actor.model.backend.lib
        self.game, self.engine, self.first, self.n = game, engine, first, n
        A = len(cfg.action_space)
        obs_floats = int(numpy.prod(cfg.observation_shape))
        assert tuple(game.shape) == tuple(cfg.observation_shape), \
            f"Observation should match the observation_shape defined in MuZeroConfig. Expected {tuple(cfg.observation_shape)} but got {tuple(game.shape)}."
        assert game.A == A, f"the game has {game.A} actions, the configuration {A}"
        st = self.staging = engine._staging(n, TAPE_WORDS, obs_floats, True)
        pin, pout = st["pin"], st["pout"]
        c_vp = ctypes.c_void_p
        self.streams = numpy.arange(first, first + n, dtype=numpy.int32)
        self.arena = engine.arena(n)
        conf = _lib.ActorConfig()
        conf.game, conf.search, conf.bank = game.native_handle, engine.handle(n, TAPE_WORDS), actor.bank.handle
        conf.streams, conf.first_slot, conf.max_moves = self.streams.ctypes.data, first, int(cfg.max_moves)
        conf.temperature = float(temperature)
        conf.d_arena, conf.arena_bytes = self.arena.data_ptr(), self.arena.numel()
        mv = conf.move
        mv.num_games, mv.action_space_size, mv.tape_words, mv.num_threads = n, A, TAPE_WORDS, actor.bank.threads
        mv.dirichlet_alpha, mv.add_exploration_noise = float(cfg.root_dirichlet_alpha), 1
        mv.h_in, mv.d_in, mv.in_bytes = st["h_in"].data_ptr(), st["d_in"].data_ptr(), st["h_in"].numel()
        mv.h_out, mv.d_out, mv.out_bytes = st["h_out"].data_ptr(), st["d_out"].data_ptr(), st["h_out"].numel()
        mv.io = _lib.SearchIO(c_vp(pin["obs"]), c_vp(pin["legal"]), c_vp(pin["to_play"]), c_vp(pin["noise"]), c_vp(pin["tape"]),
                              c_vp(pout["visits"]), c_vp(pout["root_value"]), c_vp(pout["predicted"]), c_vp(pout["info"]))
        self.handle = c_vp()
        lib.check(lib.mzx_actor_create(ctypes.byref(conf), ctypes.byref(self.handle)))
        self.lib = lib

    def close(self):
        handle, self.handle = self.handle, None
        if handle:
            self.lib.mzx_actor_destroy(handle)

    # the dict surface SelfPlay's bookkeeping reads from a slot group
    def get(self, key, default=None):
        return {"game": self.game, "pending": None, "n": self.n}.get(key, default)

    def __getitem__(self, key):
        return {"game": self.game, "engine": self.engine, "n": self.n, "slots": list(range(self.first, self.first + self.n)),
                "pending": None}[key]


class NativeShard:
    """The state of ``play_rounds`` for a natively stepped shard: one ``mzx_actor`` per slot group."""

    def __init__(self, actor, temperature):
        cfg, B = actor.config, actor.num_games
        spans = actor._batched_spans(B)
        self.groups = []
        for first, last in spans:
            single = len(spans) == 1
            game = actor.batched_game if single else type(actor.batched_game)(actor._game_seeds[first:last],
                                                                              _backend=actor.model.backend)
            engine = actor.engine if single else BatchedMCTS(cfg, actor.model, last - first)
            self.groups.append(_Group(actor, game, first, last - first, engine, temperature))
        self.handles = (ctypes.c_void_p * len(self.groups))(*[g.handle for g in self.groups])
        self.temperatures = set()
        self.sequence = 0
        self.error = None
        self._actor = actor
        self._retry = _lib.RETRY_FN(self._retry_flagged)      # (kept alive: the library calls it during a rounds call)

    def close(self):
        for g in self.groups:
            g.close()

    # ---- the rare path: a search exhausted its tie-break tape (equal priors at every level of a walk) ------------------
    def _retry_flagged(self, ctx, group, count, games):
        """Searches the flagged games of slot group ``group`` again on a longer tape -- same roots, same noise, the stream
        peeked further (BatchedMCTS._complete_run's loop) -- and writes the results into the group's output block."""
        try:
            actor, g = self._actor, self.groups[group]
            cfg, A, n = actor.config, len(actor.config.action_space), g.n
            redo = numpy.array([games[k] for k in range(count)], dtype=numpy.int64)
            vin, vout = g.staging["vin"], g.staging["vout"]
            legal, to_play = vin["legal"].reshape(n, A), vin["to_play"]
            noise, obs = vin["noise"].reshape(n, A), vin["obs"].reshape(n, -1)
            visits, info = vout["visits"].reshape(n, A), vout["info"].reshape(n, 4)
            n_legal = (legal >= 0).sum(1).astype(numpy.int32)
            words = TAPE_WORDS
            while redo.size:
                words *= 8
                if words > (1 << 22):
                    raise _lib.MzxError("tie-break tape: a search consumed more than 4M random words")
                _, long_tape = actor.bank.root_draws(g.streams[redo], cfg.root_dirichlet_alpha, n_legal[redo], A, words,
                                                     with_noise=False)
                v2, r2, p2, i2 = g.engine._launch(len(redo), obs[redo].copy(), legal[redo].copy(), to_play[redo].copy(),
                                                  noise[redo].copy(), long_tape, words, None)
                visits[redo], vout["root_value"][redo], vout["predicted"][redo], info[redo] = v2, r2, p2, i2
                redo = redo[(i2[:, 1] & 1) != 0]
            return 0
        except Exception as e:      # noqa: BLE001  (must not propagate through the C frame; re-raised by play())
            self.error = e
            return 1

    # ---- one play_rounds call -------------------------------------------------------------------------------------------
    def play(self, temperature, temperature_threshold, min_games, max_rounds):
        import time

        before = self.rounds(temperature, temperature_threshold, min_games, max_rounds)
        t0 = time.perf_counter()
        out = self.collect(before)
        self._actor.stats["native_phase_seconds"][6] += time.perf_counter() - t0      # the finished games wrapped into GameHistory views (Python)
        return out

    def rounds(self, temperature, temperature_threshold, min_games, max_rounds):
        """The library call alone (it holds no interpreter lock: ``continuous_self_play`` runs it on a worker thread while
        the main thread hands the previous call's games over).  Returns the sequence number behind the last game this call
        finished: ``collect(before)`` hands out exactly the games finished so far."""
        actor = self._actor
        lib, engine = actor.model.backend.lib, actor.engine
        t = float(temperature)
        if t != 0 and not numpy.isinf(t):
            self.temperatures.add(t)
        key = tuple(sorted(self.temperatures))
        stride = engine.num_simulations + 2
        cache = actor.__dict__.setdefault("_pow_tables", {})
        entry = cache.get((key, stride))
        if entry is None:       # visit_count ** (1 / T) over 0 .. num_simulations + 1: numpy's own pow (self_play.py:236-243)
            table = (numpy.ascontiguousarray(numpy.stack([numpy.arange(stride, dtype="int32") ** (1 / x) for x in key]))
                     if key else None)
            entry = cache[(key, stride)] = (table, numpy.array(key, numpy.float64))
        table, distinct = entry
        io = _lib.Rounds()
        io.temperature, io.temperature_threshold = t, int(temperature_threshold or 0)
        io.table_stride = stride
        io.pow_table = None if table is None else table.ctypes.data
        io.table_temperatures = distinct.ctypes.data if distinct.size else None
        io.num_temperatures = int(distinct.size)
        io.min_games, io.max_rounds = int(min(min_games, 1 << 62)), -1 if max_rounds is None else int(max_rounds)
        io.sequence = self.sequence
        io.retry, io.retry_ctx = self._retry, None
        self.error = None
        rc = lib.mzx_selfplay_rounds(self.handles, len(self.groups), ctypes.byref(io), actor.model.backend.stream())
        if self.error is not None:
            raise self.error
        lib.check(rc)
        self.sequence = int(io.sequence)
        actor.stats["searches"] += int(io.searches)
        actor.stats["simulations"] += int(io.searches) * engine.num_simulations
        actor.stats["search_seconds"] += float(io.search_seconds)
        phases = actor.stats.setdefault("native_phase_seconds", [0.0] * 7)      # (diagnostics: where the host time of the calls went)
        for k in range(6):
            phases[k] += float(io.phase_seconds[k])
        return self.sequence

    def collect(self, before=-1, priorities_for=None):
        """The finished games of every group (numbered below ``before``; -1: all) as ``ShardGameHistory`` views, in the order
        they finished; their slots.  ``priorities_for`` (a configuration with PER on; ``continuous_self_play``'s hand-off):
        the initial PER priorities of every record are computed on the device right here and every view is created with its
        row -- the buffer's save_game reads them for every game."""
        actor = self._actor
        lib, A = actor.model.backend.lib, len(actor.config.action_space)
        shape = tuple(actor.config.observation_shape)
        E = int(numpy.prod(shape))
        views_all, seq_all, slot_all, records = [], [], [], []
        for g in self.groups:
            counts = (ctypes.c_int64 * 3)()
            lib.check(lib.mzx_actor_finished(g.handle, int(before), ctypes.byref(counts)))
            G, rows = int(counts[0]), int(counts[1])
            if G == 0:
                continue
            slot, length, seq = numpy.empty(G, numpy.int32), numpy.empty(G, numpy.int32), numpy.empty(G, numpy.int64)
            obs = numpy.empty((rows + G, E), numpy.float32)
            acts, tps = numpy.empty(rows + G, numpy.int64), numpy.empty(rows + G, numpy.int64)
            rews = numpy.empty(rows + G, numpy.float64)
            vis, vals = numpy.empty((rows, A), numpy.int32), numpy.empty(rows, numpy.float64)
            mask = numpy.empty((rows, A), numpy.uint8) if counts[2] else None
            illegal = ctypes.c_int32()
            lib.check(lib.mzx_actor_take(g.handle, int(before), slot.ctypes.data, length.ctypes.data, seq.ctypes.data, obs.ctypes.data,
                                         acts.ctypes.data, rews.ctypes.data, tps.ctypes.data, vis.ctypes.data, vals.ctypes.data,
                                         None if mask is None else mask.ctypes.data, ctypes.byref(illegal)))
            game = g.game
            if game.obs_dtype != numpy.float32:       # the dtype the reference game's arrays have (planes of -1 / 0 / 1: exact)
                obs = obs.astype(game.obs_dtype)
            if game.reward_dtype is numpy.int64:
                rews = rews.astype(numpy.int64)
            off1 = numpy.cumsum(length + 1) - (length + 1)
            off0 = numpy.cumsum(length) - length
            for n in numpy.unique(length).tolist():
                rows_n = numpy.nonzero(length == n)[0]
                k = int(rows_n.size)
                if k == G:          # one length (fixed-length games, or a single game): the arrays as they lie
                    take1 = lambda a: a.reshape((k, n + 1) + a.shape[1:])
                    take0 = lambda a: a.reshape((k, n) + a.shape[1:])
                else:
                    i1 = (off1[rows_n][:, None] + numpy.arange(n + 1)[None, :])
                    i0 = (off0[rows_n][:, None] + numpy.arange(n)[None, :])
                    take1, take0 = (lambda a: a[i1]), (lambda a: a[i0])
                v, vl = take0(vis), take0(vals)
                if A <= 8:          # (numpy's reduction over a short last axis is slow: a few strided adds instead)
                    totals = v[:, :, 0].astype(numpy.int64)
                    for x in range(1, A):
                        totals += v[:, :, x]
                else:
                    totals = v.sum(2)
                ratios = v / numpy.maximum(totals, 1)[:, :, None]       # true division of small integers == Python's int / int
                plain = totals > 0
                legal_mask = None
                if mask is not None and illegal.value:
                    legal_mask = take0(mask).astype(bool)
                    plain = plain & legal_mask.all(2)
                record = _ShardRecord(A, take1(obs).reshape((k, n + 1) + shape), take1(acts), take1(rews), take1(tps), v, vl,
                                      totals, ratios, plain.all(1), legal_mask)
                if priorities_for is not None and getattr(priorities_for, "PER", False) and n > 0:
                    from . import replay
                    record.priorities, record.game_priority = replay.device_priorities(
                        actor.model.backend, numpy.where(totals > 0, vl, 0.0), record.tps, record.rews, priorities_for)
                with gc_paused():
                    views = ShardGameHistory.make_many(record, k, n)
                views_all += views
                seq_all.append(seq if k == G else seq[rows_n])
                slot_all.append(slot if k == G else slot[rows_n])
                records.append((record, n, views))
        if not views_all:
            return ShardGames(), []
        seq, slot = numpy.concatenate(seq_all), numpy.concatenate(slot_all)
        if (seq[1:] > seq[:-1]).all():          # one group, one length: already in finishing order
            out, slots = ShardGames(views_all), slot.tolist()
        else:
            order = numpy.argsort(seq, kind="stable")
            out, slots = ShardGames([views_all[i] for i in order.tolist()]), slot[order].tolist()
        out.records = records
        return out, slots
