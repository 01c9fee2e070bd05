"""
Host-side mirror of the reference's self-play surface, driving the batched HIP search.

Same names / arguments / error behaviour as /root/reference/self_play.py:
  SelfPlay(initial_checkpoint, Game, config, seed)      :11-29
  SelfPlay.continuous_self_play(shared_storage, replay_buffer, test_mode)  :31-108
  SelfPlay.play_game(temperature, temperature_threshold, render, opponent, muzero_player)  :110-183
  SelfPlay.select_action(node, temperature)              :222-245
  MCTS(config).run(model, observation, legal_actions, to_play, add_exploration_noise)  :260-361
  Node / GameHistory / MinMaxStats                        :433-570
plus the batched forms the reference lacks ("Batch MCTS" is an open TODO in its
README): ``BatchedMCTS`` searches B roots per call and ``SelfPlay(..., num_games=B)``
plays B games in lock-step on one GPU (the game shard of this process).

Division of labour: Python owns the plugin ``Game`` objects, the per-game numpy
``RandomState`` streams and the ``GameHistory`` records; everything between "stacked
observation" and "root visit counts" runs on the device behind include/mzx.h.

Random streams.  The reference draws from the process-global numpy stream in
this order per move: Dirichlet noise (:473), one ``choice`` per tied argmax
(:371), the action sample (:229-243).  A batched engine needs one stream per
game: game i of a shard uses ``RandomState(seed + i)`` -- the stream the
reference's i-th actor has (muzero.py:185) -- and with ``num_games == 1`` the
process-global stream itself, so a single game reproduces the reference draw for
draw.  Tie draws happen on the device from a tape of the stream's next raw
words (numpy's masked rejection, exactly); the host then advances the stream by
the number of words the device consumed.
"""
import ctypes
import math
import time
import warnings

import numpy
import torch

from . import _lib, _rng, models, native_rounds, replay
from . import observations as observations_mod
# round 6: the records / tree facades and the batched search live in modules of their own; every name stays importable
# from here (the reference has them all in self_play.py)
from .history import GameHistory, MinMaxStats, Node, ShardGameHistory, ShardGames, _ShardRecord, gc_paused  # noqa: F401
from .search import TAPE_WORDS, BatchedMCTS, MCTS, PendingSearch, SearchResult, _validate  # noqa: F401


def _remote(method, *args):
    """Call a storage/replay method that may be a Ray actor method (``.remote``) or a plain one."""
    if hasattr(method, "remote"):
        import ray
        return ray.get(method.remote(*args))
    return method(*args)


class _BatchedGroup(dict):
    """
    A group of slots of a shard behind the batched plugin protocol (``SelfPlay._rounds_batched``): its game object,
    engine, streams, frame store and the LOG of the games in progress -- ring arrays [row][slot], row = round % capacity,
    a move is one contiguous row per field (a game lasts at most ``max_moves`` rounds, so its rows never collide; the
    ring starts small and grows to that bound when a game outlasts it); a finished game's rows are copied out as they
    lie (move-major) and viewed game by game -- no transposition on the actor's path.  A dict for the few fields the
    actor's other methods read (``pending``, ``n``, ``slots``).
    """

    def __init__(self, actor, game, first, n, engine, temperature):
        cfg = actor.config
        obs = numpy.asarray(game.reset())
        assert obs.shape == (n,) + tuple(cfg.observation_shape), \
            f"Observation should match the observation_shape defined in MuZeroConfig. Expected {(n,) + tuple(cfg.observation_shape)} but got {obs.shape}."
        super().__init__(game=game, first=first, n=n, engine=engine, slots=list(range(first, first + n)),
                         streams=numpy.arange(first, first + n, dtype=numpy.int32), everyone=numpy.arange(n), obs=obs,
                         tp=numpy.asarray(game.to_play()).astype(numpy.int64), start=numpy.zeros(n, numpy.int64),
                         temps=numpy.full(n, float(temperature)), round=0, store=None, pending=None, legal=None)
        self.A, self.max_moves = len(cfg.action_space), int(cfg.max_moves)
        self.ring, self.cap = None, 0
        if cfg.stacked_observations > 0:   # frames stay in HBM; the stacked inputs are assembled there (csrc/mzx_obs.h)
            whole = n == actor.num_games
            self["store"] = actor._frame_store(n) if whole else observations_mod.FrameStore(cfg, n, actor.model.backend)
            self["store"].push(obs, None)

    def _row(self, r, obs, reward):
        """Ring row of round ``r``; allocates / grows the ring so that the rounds of every running game keep their own."""
        need = r - int(self["start"].min()) + 1
        if need > self.cap:
            row_bytes = max(1, obs.nbytes)      # (large observations: start with a ring of at most ~256 MB of frames)
            first = max(2, min(16, (256 << 20) // row_bytes))
            cap = min(max(2 * self.cap, need, first), self.max_moves + 1)
            n, A = self["n"], self.A
            obs_dtype, rew_dtype = obs.dtype, reward.dtype
            if self.ring is not None:        # rows already logged keep their precision (a plugin may return floats in
                # one round and ints in the next: the copy below must not truncate the earlier rows -- ADVICE r5)
                obs_dtype = numpy.result_type(self.ring["obs"].dtype, obs_dtype)
                rew_dtype = numpy.result_type(self.ring["rew"].dtype, rew_dtype)
            ring = dict(obs=numpy.empty((cap, n) + obs.shape[1:], obs_dtype), tp=numpy.empty((cap, n), numpy.int64),
                        act=numpy.empty((cap, n), numpy.int64), rew=numpy.empty((cap, n), rew_dtype),
                        vis=numpy.empty((cap, n, A), numpy.int32), val=numpy.empty((cap, n), numpy.float64), mask=None)
            if self.ring is not None:
                if self.ring["mask"] is not None:
                    ring["mask"] = numpy.ones((cap, n, A), bool)
                for q in range(int(self["start"].min()), r):
                    for name, arr in ring.items():
                        if arr is not None:
                            arr[q % cap] = self.ring[name][q % self.cap]
            self.ring, self.cap = ring, cap
        for name, value in (("obs", obs), ("rew", reward)):      # a plugin may return ints in one round and floats in another
            have = self.ring[name].dtype
            if value.dtype != have and numpy.result_type(value.dtype, have) != have:
                self.ring[name] = self.ring[name].astype(numpy.result_type(value.dtype, have))
        return r % self.cap

    def play(self, actor, result, temperature, temperature_threshold):
        """One move of every slot after its search: action draw, game step, log, finished games out, slots refilled."""
        cfg, g, n, A = actor.config, self["game"], self["n"], self.A
        r, start, store, everyone = self["round"], self["start"], self["store"], self["everyone"]
        legal = self["legal"]
        moves_before = r - start                      # len(action_history) - 1 of every slot's game
        temps = self["temps"]
        if temperature_threshold:                     # self_play.py:151-157
            temps = numpy.where(moves_before + 1 < temperature_threshold, temps, 0.0)
        actions = numpy.asarray(actor._select_actions_bank(result, self["streams"], temps), numpy.int64)
        obs2, reward, done = g.step(actions, None)
        obs2, reward = numpy.asarray(obs2), numpy.asarray(reward)
        c = self._row(r, self["obs"], reward)
        ring = self.ring
        ring["obs"][c], ring["tp"][c], ring["act"][c], ring["rew"][c] = self["obs"], self["tp"], actions, reward
        ring["vis"][c], ring["val"][c] = result.visit_counts, result.root_values
        if isinstance(legal, numpy.ndarray) and (legal >= 0).all():
            if ring["mask"] is not None:              # every action legal for every game
                ring["mask"][c] = True
        else:
            if ring["mask"] is None:
                ring["mask"] = numpy.ones((self.cap, n, A), bool)
            mask = numpy.zeros((n, A), bool)
            if isinstance(legal, numpy.ndarray):
                mask[numpy.repeat(everyone, (legal >= 0).sum(1)), legal[legal >= 0]] = True
            else:
                for i, acts in enumerate(legal):
                    mask[i, acts] = True
            ring["mask"][c] = mask
        tp2 = numpy.asarray(g.to_play()).astype(numpy.int64)
        over = numpy.asarray(done, bool) | (moves_before + 2 > cfg.max_moves)     # len(action_history) <= max_moves, :129
        pushed = actions
        finished, slots = [], []
        if over.any():
            idx = numpy.nonzero(over)[0]
            finished = self._harvest(idx, r, obs2[idx], tp2[idx])
            slots = (idx + self["first"]).tolist()
            if not obs2.flags.writeable or obs2 is self["obs"]:
                obs2 = obs2.copy()
            obs2[idx] = numpy.asarray(g.reset_games(idx))       # the slots' next games begin (self_play.py:31-52)
            tp2 = numpy.asarray(g.to_play()).astype(numpy.int64)
            start[idx] = r + 1
            self["temps"][idx] = float(temperature)
            pushed = actions.copy()
            pushed[idx] = 0
            if store is not None:
                store.clear_history(idx)
        if store is not None:
            store.push(obs2, pushed)
        self["obs"], self["tp"] = obs2, tp2
        self["round"] = r + 1
        return finished, slots

    def _harvest(self, idx, r, final_obs, final_tp):
        """GameHistory views of the games of slots ``idx`` that ended with round ``r`` (self_play.py:479-511)."""
        A, ring, cap = self.A, self.ring, self.cap
        out = [None] * len(idx)
        starts = self["start"][idx]
        for st in numpy.unique(starts):
            where = numpy.nonzero(starts == st)[0]
            slots = idx[where]
            n = r - int(st) + 1
            rows = numpy.arange(int(st), r + 1) % cap
            everyone = slots.size == self["n"]
            contiguous = rows[-1] - rows[0] == n - 1               # the game's rounds did not wrap around the ring
            if everyone and contiguous:
                take = lambda a: a[rows[0]: rows[0] + n]           # (a view: copied into the record's arrays below)
            elif everyone:
                take = lambda a: a[rows]
            else:
                take = lambda a: a[rows[:, None], slots[None, :]]
            k = slots.size
            obs_all = numpy.empty((n + 1, k) + ring["obs"].shape[2:], ring["obs"].dtype)
            obs_all[:n], obs_all[n] = take(ring["obs"]), final_obs[where]
            acts = numpy.zeros((n + 1, k), numpy.int64)
            acts[1:] = take(ring["act"])
            rews = numpy.zeros((n + 1, k), ring["rew"].dtype)
            rews[1:] = take(ring["rew"])
            tps = numpy.empty((n + 1, k), numpy.int64)
            tps[:n], tps[n] = take(ring["tp"]), final_tp[where]
            vis, vals = numpy.array(take(ring["vis"])), numpy.array(take(ring["val"]))      # [n][games][A], [n][games] (copies)
            totals = vis.sum(2)
            ratios = vis / numpy.maximum(totals, 1)[:, :, None]       # true division of small integers == Python's int / int
            plain = totals > 0
            legal_mask = None
            if ring["mask"] is not None:
                legal_mask = numpy.array(take(ring["mask"]))
                plain = plain & legal_mask.all(2)
            record = _ShardRecord(A, obs_all, acts, rews, tps, vis, vals, totals, ratios, plain.all(0), legal_mask, time_major=True)
            with gc_paused():
                views = [ShardGameHistory(record, j, n) for j in range(k)]
            if k == len(idx):
                return views
            for j, w in enumerate(where):
                out[w] = views[j]
        return out


class SelfPlay:
    """
    self_play.py:11-245.  ``num_games`` (new, default 1) is the number of games
    this process plays in lock-step on its GPU; game i is seeded ``seed + i``.
    """

    def __init__(self, initial_checkpoint, Game, config, seed, num_games=1, _backend=None):
        _validate(config)
        self.config = config
        self.num_games = int(num_games)
        self.batched_game = None
        if getattr(Game, "batched", False):
            # optional batched plugin protocol (mzx.synthetic.make_synthetic_batched_game documents it): ONE
            # object steps the whole shard, the per-move host work is a handful of numpy calls
            self._game_seeds = [seed + i for i in range(self.num_games)]
            # (a natively stepped game -- mzx.games.NativeBatchedGame -- lives in the library this actor's model runs on)
            self.batched_game = (Game(self._game_seeds, _backend=_backend) if getattr(Game, "native", False)
                                 else Game(self._game_seeds))
            self.games = []
            self.game = self.batched_game
        else:
            self.games = [Game(seed + i) for i in range(self.num_games)]
            self.game = self.games[0]

        # Fix random generator seed (self_play.py:21-23)
        numpy.random.seed(seed)
        torch.manual_seed(seed)
        # game 0 of a single-game actor draws from the process-global stream like the reference; a shard of
        # games uses the native stream bank, stream i = RandomState(seed + i) (muzero.py:185 seeds actor i so)
        self.rngs = [numpy.random.mtrand._rand]
        self.bank = None

        # Initialize the network (self_play.py:25-29)
        self.model = models.MuZeroNetwork(self.config, _backend=_backend)
        self.model.set_weights(initial_checkpoint["weights"])
        self.model.eval()
        self.engine = BatchedMCTS(self.config, self.model, self.num_games)
        if self.num_games > 1 or self.batched_game is not None:
            self.bank = _rng.StreamBank(self.model.backend.lib, [(seed + i) & 0xFFFFFFFF for i in range(self.num_games)])
        self.stats = {"searches": 0, "simulations": 0, "search_seconds": 0.0}
        self._live = None       # state of the games in progress under play_rounds

    # ------------------------------------------------------------------ loops
    def continuous_self_play(self, shared_storage, replay_buffer, test_mode=False):
        """
        self_play.py:31-108.  With a ``mzx.shared_storage.ShardedStorage`` (one self-play process per GPU under
        torch.distributed) the per-game weight pull (:37) becomes the storage's ``refresh``: a NON-BLOCKING step of
        an asynchronous exchange (one six-word control all-reduce in flight at a time; when the trainer published
        new weights ONE RCCL broadcast of the flat buffer) -- a rank never waits for another one inside this loop,
        it plays on with the control values and weights it has; the stop condition is evaluated on the same
        exchange by every rank, ``finish`` drains the sequence.  This rank's games are seeded ``seed + i`` with ``seed`` =
        ``shard_seeds(config.seed, num_games)[0]`` (muzero.py:185).
        """
        sharded = hasattr(shared_storage, "refresh")
        get = (shared_storage.get_info if sharded else (lambda key: _remote(shared_storage.get_info, key)))
        if sharded:
            # what every rank must evaluate alike / how often the trainer publishes (trainer.py: checkpoint_interval)
            if getattr(shared_storage, "training_steps", 0) is None:
                shared_storage.training_steps = self.config.training_steps
            if getattr(shared_storage, "checkpoint_interval", 0) is None:
                shared_storage.checkpoint_interval = getattr(self.config, "checkpoint_interval", 10)
            shared_storage.refresh(self.model, block=True)     # the trainer's weights before the first game
        save_game = replay_buffer.save_game
        save_is_remote = hasattr(save_game, "remote")

        def hand_off(histories):
            # initial PER priorities, on the device (replay_buffer.py:39-51 would loop in Python) -- the games a shard
            # hands out together in ONE pass over their record; save_game then takes its "priorities already present" branch
            replay.fill_initial_priorities_many(histories, self.config, backend=self.model.backend)
            if save_is_remote:
                for game_history in histories:
                    _remote(save_game, game_history, shared_storage)
            else:
                for game_history in histories:
                    save_game(game_history, shared_storage)

        # A natively played shard (mzx/native_rounds.py) spends its rounds inside ONE library call that needs no
        # interpreter: the hand-off of the games call k returned -- priorities, save_game of every game: Python per game, as
        # much wall time as playing them -- runs on this thread WHILE a worker thread is inside call k + 1.  Weights are
        # only set between calls, with no call in flight; a game reaches the buffer one call later than it would otherwise
        # (``config.self_play_overlap_handoff = False``: strictly in turn).
        overlap = (not test_mode and native_rounds.usable(self) and getattr(self.config, "refill_finished_games", True)
                   and getattr(self.config, "self_play_overlap_handoff", True))
        waiting, executor = None, None
        if overlap:
            import concurrent.futures
            executor = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            on_gpu = self.model.backend.device.type == "cuda"
            stream = torch.cuda.current_stream(self.model.backend.device) if on_gpu else None

            def rounds_on_worker(temperature, threshold):
                # ONLY the library call runs here (no interpreter lock held inside it); wrapping the finished games into
                # GameHistory views is Python and stays on the main thread, inside the hand-off the call overlaps with
                shard = self._native_shard(temperature)
                if stream is None:
                    return shard.rounds(temperature, threshold, self.num_games, None)
                torch.cuda.set_device(self.model.backend.device)      # (device and stream are thread-local in torch)
                with torch.cuda.stream(stream):
                    return shard.rounds(temperature, threshold, self.num_games, None)

        while get("training_step") < self.config.training_steps and not get("terminate"):
            if not sharded:
                self.model.set_weights(get("weights"))
            if overlap:
                future = executor.submit(rounds_on_worker, self.config.visit_softmax_temperature_fn(trained_steps=get("training_step")),
                                         self.config.temperature_threshold)
                try:
                    if waiting is not None:          # the games the previous call finished (numbered below `waiting`)
                        t0 = time.perf_counter()
                        histories, slots = self._live["native"].collect(waiting, priorities_for=self.config)
                        self.finished_slots = slots
                        t1 = time.perf_counter()
                        hand_off(histories)
                        t2 = time.perf_counter()
                        spent = self.stats.setdefault("handoff_seconds", [0.0, 0.0, 0.0])     # collect (+ priorities), save_game, idle
                        spent[0] += t1 - t0
                        spent[1] += t2 - t1
                finally:
                    t3 = time.perf_counter()
                    waiting = future.result()
                    self.stats.setdefault("handoff_seconds", [0.0, 0.0, 0.0])[2] += time.perf_counter() - t3
            elif not test_mode:
                # every slot of the shard is one reference actor: its next game starts the moment one ends (:31-52), so
                # every search runs at full width (``refill_finished_games = False``: whole shards in lock-step)
                play = self.play_rounds if getattr(self.config, "refill_finished_games", True) else (
                    lambda t, th: self.play_games(t, th, False, "self", 0))
                hand_off(play(self.config.visit_softmax_temperature_fn(trained_steps=get("training_step")),
                              self.config.temperature_threshold))
            else:
                game_history = self.play_game(
                    0, self.config.temperature_threshold, False,
                    "self" if len(self.config.players) == 1 else self.config.opponent, self.config.muzero_player,
                )
                _remote(shared_storage.set_info, {
                    "episode_length": len(game_history.action_history) - 1,
                    "total_reward": sum(game_history.reward_history),
                    "mean_value": numpy.mean([value for value in game_history.root_values if value]),
                })
                if 1 < len(self.config.players):
                    _remote(shared_storage.set_info, {
                        "muzero_reward": sum(
                            reward for i, reward in enumerate(game_history.reward_history)
                            if game_history.to_play_history[i - 1] == self.config.muzero_player),
                        "opponent_reward": sum(
                            reward for i, reward in enumerate(game_history.reward_history)
                            if game_history.to_play_history[i - 1] != self.config.muzero_player),
                    })
            if not test_mode and self.config.self_play_delay:
                time.sleep(self.config.self_play_delay)
            if sharded:
                shared_storage.refresh(self.model)     # played counts out, control + (new) weights in
            if not test_mode and self.config.ratio:
                while (get("training_step") / max(1, get("num_played_steps")) < self.config.ratio
                       and get("training_step") < self.config.training_steps and not get("terminate")):
                    time.sleep(0.5)
                    if sharded:
                        shared_storage.refresh(self.model)
        if executor is not None:
            executor.shutdown(wait=True)
            if waiting is not None:          # the games of the last call
                histories, slots = self._live["native"].collect(waiting, priorities_for=self.config)
                self.finished_slots = slots
                hand_off(histories)
        if sharded and hasattr(shared_storage, "finish"):
            shared_storage.finish(self.model)
        self.close_game()

    def play_game(self, temperature, temperature_threshold, render, opponent, muzero_player):
        """self_play.py:110-183: one game (game slot 0, process-global numpy stream)."""
        return self._play([0], temperature, temperature_threshold, render, opponent, muzero_player)[0]

    def play_games(self, temperature, temperature_threshold, render, opponent, muzero_player):
        """All ``num_games`` games of this shard in lock-step; returns their GameHistory list."""
        self._end_live()        # (games in progress under play_rounds end here: every game object is reset)
        if self.batched_game is not None:
            if opponent != "self" or render:
                raise NotImplementedError("the batched game protocol covers self-play without rendering")
            return self._play_batched(temperature, temperature_threshold)
        if self.bank is not None and opponent == "self" and not render:
            return self._play_shard(temperature, temperature_threshold)
        return self._play(list(range(self.num_games)), temperature, temperature_threshold, render, opponent,
                          muzero_player)

    def _end_live(self):
        """Ends the games in progress under ``play_rounds``: queued searches drained, the slot groups' own game objects
        (and native actors) released."""
        self._drain_searches()
        live, self._live = self._live, None
        for group in (live or {}).get("groups", ()):      # (slot groups of the batched protocol own their game objects)
            if group.get("game") is not None and group["game"] is not self.batched_game:
                group["game"].close()
        if live and live.get("native") is not None:
            live["native"].close()

    def _check_observation(self, observation):
        # self_play.py:132-137 (same messages); ndarray observations of the right shape take the fast exit
        shape = self.config.observation_shape
        if getattr(observation, "shape", None) == shape:
            return
        got = numpy.array(observation).shape
        assert len(got) == 3, \
            f"Observation should be 3 dimensionnal instead of {len(got)} dimensionnal. Got observation of shape: {got}"
        assert got == shape, \
            f"Observation should match the observation_shape defined in MuZeroConfig. Expected {shape} but got {got}."

    def _play_shard(self, temperature, temperature_threshold):
        """
        play_game (self_play.py:110-183) for all games of the shard through the REFERENCE plugin surface (B
        unmodified ``Game`` objects): self-play, no rendering, native stream bank.  Everything that is not a call
        into a plugin object or an append to one of its GameHistory lists is done once per move for the whole
        shard: observations land in one persistent float32 batch as they are produced, legal-action lists are
        validated once per distinct list, actions / child_visits rows / root values come from array operations.
        Game by game the result equals the single-game actor's (tests/test_selfplay_shard.py).
        """
        cfg, games = self.config, self.games
        A, G = len(cfg.action_space), len(self.games)
        shape = tuple(cfg.observation_shape)
        cfg_shape_is_tuple = cfg.observation_shape if isinstance(cfg.observation_shape, tuple) else shape
        batch = getattr(self, "_obs_batch", None)
        if batch is None or batch.shape != (G,) + shape:
            batch = self._obs_batch = numpy.empty((G,) + shape, numpy.float32)
        histories = []
        for s, game in enumerate(games):
            gh = GameHistory()
            observation = game.reset()
            gh.action_history.append(0)
            gh.observation_history.append(observation)
            gh.reward_history.append(0)
            gh.to_play_history.append(game.to_play())
            self._check_observation(observation)
            batch[s] = observation
            histories.append(gh)
        store = None
        if cfg.stacked_observations > 0:
            store = self._frame_store(G)
            store.push(batch, None)
        max_moves = cfg.max_moves
        active = list(range(G))
        moved = numpy.zeros(G, numpy.int32)
        while active:
            everyone = len(active) == G
            legal = [games[s].legal_actions() for s in active]
            to_play = [games[s].to_play() for s in active]
            if store is not None:
                stacked = store.stacked(None if everyone else active)
            else:
                stacked = batch if everyone else batch[active]
            t0 = time.perf_counter()
            result = self.engine.run(stacked, legal, to_play, True, (self.bank, active))
            self.stats["search_seconds"] += time.perf_counter() - t0
            self.stats["searches"] += len(active)
            self.stats["simulations"] += len(active) * self.engine.num_simulations
            if temperature_threshold:
                temps = [temperature if len(histories[s].action_history) < temperature_threshold else 0 for s in active]
            else:
                temps = temperature
            actions = self._select_actions_bank(result, active, temps)
            vis = result.visit_counts
            totals = vis.sum(1)
            rows = (vis / numpy.maximum(totals, 1)[:, None]).tolist()     # int / int true division, as Python's
            values = result.root_values.tolist()
            plain = bool((totals > 0).all()) and all(len(l) == A for l in result.legal_actions)
            still = []
            for j, s in enumerate(active):
                game, gh, action = games[s], histories[s], actions[j]
                observation, reward, done = game.step(action)
                if plain:
                    gh.child_visits.append(rows[j])
                    gh.root_values.append(values[j])
                else:      # illegal actions get 0, an unvisited root reports value 0 (self_play.py:496-511, :446-449)
                    total, legal_set = int(totals[j]), set(result.legal_actions[j])
                    gh.child_visits.append([int(vis[j][a]) / total if a in legal_set else 0 for a in cfg.action_space])
                    gh.root_values.append(values[j] if total else 0)
                gh.action_history.append(action)
                gh.observation_history.append(observation)
                gh.reward_history.append(reward)
                gh.to_play_history.append(game.to_play())
                if getattr(observation, "shape", None) != cfg_shape_is_tuple:
                    self._check_observation(observation)
                batch[s] = observation
                if not done and len(gh.action_history) <= max_moves:
                    still.append(s)
            if store is not None and still:
                for s in active:
                    moved[s] = histories[s].action_history[-1]
                store.push(batch, moved)
            active = still
        return histories

    def _play(self, slots, temperature, temperature_threshold, render, opponent, muzero_player):
        cfg = self.config
        A = len(cfg.action_space)
        histories = {}
        observations = {}
        for s in slots:
            gh = GameHistory()
            observation = self.games[s].reset()
            gh.action_history.append(0)
            gh.observation_history.append(observation)
            gh.reward_history.append(0)
            gh.to_play_history.append(self.games[s].to_play())
            histories[s] = gh
            observations[s] = observation
            if render:
                self.games[s].render()
        # stacked observations of a self-play shard are assembled on the device from a frame store
        # (one upload of the new frames per move) instead of per-game numpy concatenations
        store = None
        if cfg.stacked_observations > 0 and opponent == "self":
            store = self._frame_store(len(self.games))
            frame = numpy.zeros((len(self.games),) + tuple(cfg.observation_shape), numpy.float32)
            for s in slots:
                frame[s] = numpy.asarray(observations[s])
            store.push(frame, None)
        active = list(slots)
        while active:
            searching, stacked, stacked_by_slot = [], [], {}
            for s in active:
                observation, gh = observations[s], histories[s]
                # self_play.py:132-137
                assert len(numpy.array(observation).shape) == 3, \
                    f"Observation should be 3 dimensionnal instead of {len(numpy.array(observation).shape)} dimensionnal. Got observation of shape: {numpy.array(observation).shape}"
                assert numpy.array(observation).shape == cfg.observation_shape, \
                    f"Observation should match the observation_shape defined in MuZeroConfig. Expected {cfg.observation_shape} but got {numpy.array(observation).shape}."
                st = None if store is not None else gh.get_stacked_observations(-1, cfg.stacked_observations, A)
                if opponent == "self" or muzero_player == self.games[s].to_play():
                    searching.append(s)
                    stacked.append(st)
                else:
                    stacked_by_slot[s] = st      # what THIS game's opponent sees (self_play.py:161-165)
            result = None
            if searching:
                t0 = time.perf_counter()
                result = self.engine.run(
                    stacked if store is None else store.stacked(searching),
                    [self.games[s].legal_actions() for s in searching],
                    [self.games[s].to_play() for s in searching], True,
                    (self.bank, searching) if self.bank is not None else [self.rngs[s] for s in searching],
                )
                self.stats["search_seconds"] += time.perf_counter() - t0
                self.stats["searches"] += len(searching)
                self.stats["simulations"] += len(searching) * self.engine.num_simulations
            still = []
            position = {s: k for k, s in enumerate(searching)}
            batch_actions = None
            if self.bank is not None and searching:
                temps = [temperature if not temperature_threshold or len(histories[s].action_history) < temperature_threshold
                         else 0 for s in searching]
                batch_actions = self._select_actions_bank(result, searching, temps)
            # shard bookkeeping in bulk: with the stream bank the per-game Node objects are only needed for
            # rendering; the child_visits rows (visit / total, 0 for illegal actions, self_play.py:496-511)
            # and root values come from two array operations per move
            fast_rows = None
            if batch_actions is not None and not render:
                vis = result.visit_counts
                totals = vis.sum(1)
                ratios = (vis / numpy.maximum(totals, 1)[:, None]).tolist()   # int / int true division, as Python's
                fast_rows = (vis, totals, ratios)
            for s in active:
                gh, game = histories[s], self.games[s]
                root = None
                if s in position and fast_rows is not None:
                    k = position[s]
                    action = batch_actions[k]
                elif s in position:
                    root = result.root(position[s])
                    t = temperature if not temperature_threshold or len(gh.action_history) < temperature_threshold else 0
                    action = batch_actions[position[s]] if batch_actions is not None else self._select_action(root, t, self.rngs[s])
                    if render:
                        print(f'Tree depth: {result.max_tree_depth[position[s]]}')
                        print(f"Root value for player {game.to_play()}: {root.value():.2f}")
                else:
                    action, root = self.select_opponent_action(opponent, stacked_by_slot[s], game)
                observation, reward, done = game.step(action)
                if render:
                    print(f"Played action: {game.action_to_string(action)}")
                    game.render()
                if s in position and fast_rows is not None:
                    vis, totals, ratios = fast_rows
                    k, legal_k = position[s], result.legal_actions[position[s]]
                    total = int(totals[k])
                    if len(legal_k) == A and total:
                        gh.child_visits.append(ratios[k])
                    else:
                        legal_set = set(legal_k)
                        gh.child_visits.append([int(vis[k][a]) / total if a in legal_set else 0 for a in cfg.action_space])
                    gh.root_values.append(float(result.root_values[k]) if total else 0)
                else:
                    gh.store_search_statistics(root, cfg.action_space)
                gh.action_history.append(action)
                gh.observation_history.append(observation)
                gh.reward_history.append(reward)
                gh.to_play_history.append(game.to_play())
                observations[s] = observation
                if not done and len(gh.action_history) <= cfg.max_moves:
                    still.append(s)
            if store is not None and still:
                moved = numpy.zeros(len(self.games), numpy.int32)
                for s in active:
                    frame[s] = numpy.asarray(observations[s])
                    moved[s] = histories[s].action_history[-1]
                store.push(frame, moved)
            active = still
        return [histories[s] for s in slots]

    def _play_batched(self, temperature, temperature_threshold):
        """
        play_game (self_play.py:110-183) for a shard behind the batched plugin protocol: per move ONE game
        call, ONE search, ONE action draw for all running games; the per-game GameHistory objects
        (field-identical to the per-object path, tests/test_selfplay_shard.py) are materialised at the end.
        While every game of the shard is still running (the common case) no index gathers / scatters happen.
        """
        cfg, g, B = self.config, self.batched_game, self.num_games
        A, k = len(cfg.action_space), int(cfg.stacked_observations)
        obs = numpy.asarray(g.reset())
        assert obs.shape == (B,) + tuple(cfg.observation_shape), \
            f"Observation should match the observation_shape defined in MuZeroConfig. Expected {(B,) + tuple(cfg.observation_shape)} but got {obs.shape}."
        obs_hist, act_hist, rew_hist = [obs], [numpy.zeros(B, numpy.int64)], [numpy.zeros(B, numpy.int64)]
        tp_hist = [numpy.asarray(g.to_play()).astype(numpy.int64)]
        visits_hist, value_hist, legal_hist = [], [], []
        alive = numpy.ones(B, bool)
        n_alive = B
        length = numpy.zeros(B, numpy.int64)
        everyone_idx = numpy.arange(B)
        move = 0
        store = None
        if k > 0:   # frames stay in HBM; the stacked inputs are assembled there (csrc/mzx_obs.h)
            store = self._frame_store(B)
            store.push(obs, None)
        while n_alive and move + 1 <= cfg.max_moves:      # len(action_history) <= max_moves, :129
            everyone = n_alive == B
            idx = everyone_idx if everyone else numpy.nonzero(alive)[0]
            if store is not None:
                stacked = store.stacked(None if everyone else idx)
            else:
                stacked = obs_hist[-1] if everyone else obs_hist[-1][idx]
            legal = g.legal_actions()
            if not everyone:
                legal = legal[idx] if isinstance(legal, numpy.ndarray) else [legal[i] for i in idx]
            to_play = tp_hist[-1] if everyone else tp_hist[-1][idx]
            t0 = time.perf_counter()
            result = self.engine.run(stacked, legal, to_play, True, (self.bank, idx))
            self.stats["search_seconds"] += time.perf_counter() - t0
            self.stats["searches"] += len(idx)
            self.stats["simulations"] += len(idx) * self.engine.num_simulations
            t = temperature if not temperature_threshold or move + 1 < temperature_threshold else 0
            chosen = self._select_actions_bank(result, idx, t)
            if everyone:
                actions = numpy.asarray(chosen, numpy.int64)
                visits, values = result.visit_counts, result.root_values
            else:
                actions = numpy.zeros(B, numpy.int64)
                actions[idx] = chosen
                visits = numpy.zeros((B, A), numpy.int32)
                visits[idx] = result.visit_counts
                values = numpy.zeros(B, numpy.float64)
                values[idx] = result.root_values
            obs, reward, done = g.step(actions, alive if everyone else alive.copy())
            if isinstance(legal, numpy.ndarray) and (legal >= 0).all():
                mask = None                                  # every action legal for every searched game
            else:
                mask = numpy.zeros((B, A), bool)
                if isinstance(legal, numpy.ndarray):
                    rows = numpy.repeat(idx, (legal >= 0).sum(1))
                    mask[rows, legal[legal >= 0]] = True
                else:
                    for r, acts in zip(idx, legal):
                        mask[r, acts] = True
            visits_hist.append(visits); value_hist.append(values); legal_hist.append(mask)
            obs_hist.append(numpy.asarray(obs)); act_hist.append(actions)
            if store is not None:
                store.push(obs_hist[-1], actions)
            rew_hist.append(numpy.asarray(reward)); tp_hist.append(numpy.asarray(g.to_play()).astype(numpy.int64))
            if everyone:
                length += 1
            else:
                length[idx] += 1
            done = numpy.asarray(done, bool)
            if done.any():
                alive = alive & ~done
                n_alive = int(alive.sum())
            move += 1
        # ---- per-game records (self_play.py:479-511): transpose once to game-major, slice per game
        n_moves = len(visits_hist)
        by_game = lambda seq: numpy.ascontiguousarray(numpy.swapaxes(numpy.stack(seq), 0, 1))
        acts, rews, tps, obs_all = by_game(act_hist), by_game(rew_hist), by_game(tp_hist), by_game(obs_hist)
        record = _ShardRecord(A, obs_all, acts, rews, tps)
        if n_moves:
            vis = by_game(visits_hist)                    # [B][n_moves][A]
            vals = by_game(value_hist)
            totals = vis.sum(2)
            # visit_count / total: true division of small integers == Python's int / int
            ratios = vis / numpy.maximum(totals, 1)[:, :, None]
            played = numpy.arange(n_moves)[None, :] < length[:, None]
            plain = (totals > 0) | ~played
            legal_mask = None
            if any(m is not None for m in legal_hist):
                legal_mask = by_game([numpy.ones((B, A), bool) if m is None else m for m in legal_hist])
                plain &= legal_mask.all(2)
            record = _ShardRecord(A, obs_all, acts, rews, tps, vis, vals, totals, ratios, plain.all(1), legal_mask)
        return [ShardGameHistory(record, i, n) for i, n in enumerate(length.tolist())]

    # ------------------------------------------------------------------ continuous play: finished slots are refilled
    def play_rounds(self, temperature, temperature_threshold, min_games=None, max_rounds=None):
        """
        Self-play of the shard WITHOUT lock-step game boundaries: the reference actor starts its next game the moment
        one ends (self_play.py:31-52) -- here every slot s of the shard is such an actor (its ``Game`` object and its
        numpy stream ``RandomState(seed + s)`` live on from game to game, exactly the sequence a lone reference-style
        actor seeded ``seed + s`` produces, tests/test_selfplay_refill.py), and ALL slots are searched together in every
        round: when the game of slot s ends its GameHistory is handed out and the slot restarts at once, so every
        search launch runs at the full shard width (``play_games`` searches a thinning batch until the longest game of
        the shard ends: connect4 games average 21 moves and last up to 42).

        Plays rounds (one move of every slot) until at least ``min_games`` games have finished (default: one shard's
        worth) or ``max_rounds`` rounds were played; returns the games that finished, in the order they did.  Games in
        progress stay in the shard and continue with the next call.  ``temperature`` applies to the games that START
        during this call (the reference reads it once per game, self_play.py:39-41); weights may be refreshed between
        calls, i.e. between two searches -- a game can straddle a refresh (deliberate: the alternative is lock-step).
        Batched games need the optional ``reset_games(idx)`` hook; without it, and for a single-game actor, this falls
        back to ``play_games``.
        """
        if self.bank is None or (self.batched_game is not None and not hasattr(self.batched_game, "reset_games")):
            self.finished_slots = list(range(self.num_games))
            return self.play_games(temperature, temperature_threshold, False, "self", 0)
        min_games = self.num_games if min_games is None else int(min_games)
        self.finished_slots = []      # slot of every returned game, in the same order (slot s = the actor seeded seed + s)
        if self.batched_game is not None:
            return self._rounds_batched(temperature, temperature_threshold, min_games, max_rounds)
        return self._rounds_shard(temperature, temperature_threshold, min_games, max_rounds)

    def _slot_groups(self, G):
        """
        The slots of a shard of B ``Game`` objects as ONE group, or as TWO that take turns on the GPU: while the search of
        one group runs (a worker thread inside the C call / the stream synchronisation, both of which release the GIL),
        the host steps the other group's ``Game`` objects -- the Python work of the reference plugin surface (a third of
        the wall for connect4 at 1024 games) hides behind the other half's search.  Slots are independent actors (own
        ``Game``, own numpy stream) and the library searches a network on ONE arithmetic whatever the shard size (wide
        residual networks: csrc/mzx_row_search.h wide_search_route), so which slots share a launch does not change what any of
        them plays -- bit for bit, on the device too (tests/test_gpu_parity.py: 1024 connect4 games, two groups against one).
        ``config.self_play_pipeline``: True / False; unset = from 1024 games on -- each half must still fill the chip: the
        residual whole-search kernels run a workgroup per tree or pair of trees, so a search of fewer than ~512 trees takes as
        long as one of 512 and two half searches would cost more than they hide.
        """
        want = getattr(self.config, "self_play_pipeline", None)
        if want is None:
            want = G >= 1024
        if not want or G < 2:
            return [(0, G)]
        half = (G + 1) // 2
        return [(0, half), (half, G)]

    def _search_job(self, group, stacked, legal, to_play, stream):
        """One search of a slot group (runs on the worker thread when the shard is pipelined)."""
        device = self.model.backend.device
        t0 = time.perf_counter()
        if stream is not None:       # the submitting thread's device and stream (both are thread-local in torch)
            torch.cuda.set_device(device)
            with torch.cuda.stream(stream):
                result = group["engine"].run(stacked, legal, to_play, True, (self.bank, group["slots"]))
        else:
            result = group["engine"].run(stacked, legal, to_play, True, (self.bank, group["slots"]))
        return result, time.perf_counter() - t0

    def _rounds_shard(self, temperature, temperature_threshold, min_games, max_rounds):
        """``play_rounds`` through the reference plugin surface (B ``Game`` objects); bookkeeping as ``_play_shard``."""
        cfg, games = self.config, self.games
        A, G = len(cfg.action_space), len(self.games)
        shape = tuple(cfg.observation_shape)
        cfg_shape_is_tuple = cfg.observation_shape if isinstance(cfg.observation_shape, tuple) else shape
        live = self._live

        def start(s, group):
            gh = GameHistory()
            observation = games[s].reset()
            gh.action_history.append(0)
            gh.observation_history.append(observation)
            gh.reward_history.append(0)
            gh.to_play_history.append(games[s].to_play())
            self._check_observation(observation)
            group["batch"][s - group["first"]] = observation
            live["histories"][s] = gh
            live["temps"][s] = temperature

        if live is None:
            live = self._live = dict(histories=[None] * G, temps=[temperature] * G, groups=[])
            spans = self._slot_groups(G)
            for first, last in spans:
                n = last - first
                group = dict(first=first, slots=list(range(first, last)), batch=numpy.empty((n,) + shape, numpy.float32),
                             moved=numpy.zeros(n, numpy.int32), store=None, pending=None,
                             engine=self.engine if len(spans) == 1 else BatchedMCTS(cfg, self.model, n))
                live["groups"].append(group)
                for s in group["slots"]:
                    start(s, group)
                if cfg.stacked_observations > 0:
                    group["store"] = (self._frame_store(G) if len(spans) == 1 else
                                      observations_mod.FrameStore(cfg, n, self.model.backend))
                    group["store"].push(group["batch"], None)
        histories, groups = live["histories"], live["groups"]
        pipelined = len(groups) > 1
        if pipelined and getattr(self, "_search_worker", None) is None:
            import concurrent.futures
            self._search_worker = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="mzx-search")
        max_moves = cfg.max_moves
        finished, rounds = [], 0

        def submit(group):
            slots = group["slots"]
            legal = [games[s].legal_actions() for s in slots]
            to_play = [games[s].to_play() for s in slots]
            stacked = group["store"].stacked(None) if group["store"] is not None else group["batch"]
            if pipelined:
                device = self.model.backend.device
                stream = torch.cuda.current_stream(device) if device.type == "cuda" else None
                # A search queued across the end of this call is consumed by the NEXT play_rounds call; should play_games /
                # close_game come instead, its result is dropped -- and the root noise / tie words it drew are given back
                # (_drain_searches), so that the slots' streams stay those of lone actors.  The states are only kept when
                # this call can actually end before the result is consumed.
                may_be_last = (len(finished) + sum(len(g["slots"]) for g in groups) >= min_games
                               or (max_rounds is not None and rounds + 2 >= max_rounds))
                group["streams_before"] = [self.bank.get_state(s) for s in slots] if may_be_last else None
                group["pending"] = self._search_worker.submit(self._search_job, group, stacked, legal, to_play, stream)
            else:
                group["pending"] = self._search_job(group, stacked, legal, to_play, None)

        def consume(group):
            pending, group["pending"] = group["pending"], None
            result, seconds = pending.result() if pipelined else pending
            slots, first, batch, moved, store = group["slots"], group["first"], group["batch"], group["moved"], group["store"]
            n = len(slots)
            self.stats["search_seconds"] += seconds
            self.stats["searches"] += n
            self.stats["simulations"] += n * group["engine"].num_simulations
            slot_temps = live["temps"][first:first + n]
            if temperature_threshold:
                temps = [slot_temps[s - first] if len(histories[s].action_history) < temperature_threshold else 0 for s in slots]
            elif slot_temps.count(slot_temps[0]) == n:
                temps = slot_temps[0]
            else:
                temps = list(slot_temps)
            actions = self._select_actions_bank(result, slots, temps)
            vis = result.visit_counts
            totals = vis.sum(1)
            rows = (vis / numpy.maximum(totals, 1)[:, None]).tolist()     # int / int true division, as Python's
            values = result.root_values.tolist()
            plain = bool((totals > 0).all()) and all(len(l) == A for l in result.legal_actions)
            restarted = []
            for k, s in enumerate(slots):
                game, gh, action = games[s], histories[s], actions[k]
                observation, reward, done = game.step(action)
                if plain:
                    gh.child_visits.append(rows[k])
                    gh.root_values.append(values[k])
                else:      # illegal actions get 0, an unvisited root reports value 0 (self_play.py:496-511, :446-449)
                    total, legal_set = int(totals[k]), set(result.legal_actions[k])
                    gh.child_visits.append([int(vis[k][a]) / total if a in legal_set else 0 for a in cfg.action_space])
                    gh.root_values.append(values[k] if total else 0)
                gh.action_history.append(action)
                gh.observation_history.append(observation)
                gh.reward_history.append(reward)
                gh.to_play_history.append(game.to_play())
                if getattr(observation, "shape", None) != cfg_shape_is_tuple:
                    self._check_observation(observation)
                batch[k] = observation
                moved[k] = action
                if done or len(gh.action_history) > max_moves:     # self_play.py:129: the game is over
                    finished.append(gh)
                    self.finished_slots.append(s)
                    start(s, group)                                # ... and the slot's next game begins (:31-52)
                    moved[k] = 0
                    restarted.append(k)
            if store is not None:
                store.clear_history(restarted)
                store.push(batch, moved)

        while len(finished) < min_games and (max_rounds is None or rounds < max_rounds):
            for group in groups:
                if group["pending"] is None:     # (a pipelined shard may hold a search from the end of the last call)
                    submit(group)
            for k, group in enumerate(groups):
                consume(group)
                # pipelined: this group's NEXT search is queued now, so that it runs while the host steps the groups
                # after it -- unless this is known to be the last round of the call
                if (pipelined and k + 1 < len(groups) and len(finished) < min_games
                        and (max_rounds is None or rounds + 1 < max_rounds)):
                    submit(group)
            rounds += 1
        if pipelined:
            # nothing runs on the worker between calls (the caller may load new weights): a search queued for the next
            # round finishes here, its result waits in the group (a round is one move of EVERY slot, so the call ends on
            # a round boundary; the slots of that group have already drawn their root noise for the next move -- in the
            # order a lone actor draws it)
            for group in groups:
                if group["pending"] is not None:
                    group["pending"].result()
        return finished

    def _rounds_batched(self, temperature, temperature_threshold, min_games, max_rounds):
        """
        ``play_rounds`` behind the batched plugin protocol: per round and slot group ONE game call, ONE library call for
        the host side of the search (``mzx_selfplay_search``: draws, staging, upload, launch, download), ONE for the
        streams' advance and the action draw (``mzx_selfplay_select``).  A move is logged as one column of a handful
        of game-major arrays (``_BatchedGroup``); a finished game is a gather of its slot's columns, handed out as a
        ``ShardGameHistory`` view (games that started and ended together share one record).

        Slot groups (``_batched_spans``): a large shard of a cheaply searched network runs as two groups with their own
        game object, engine and staging that take turns on the GPU -- the search of one group is queued, the host
        draws / steps / logs the other group, then waits for the first (an event behind the asynchronous library call;
        no worker thread).  Slots are independent actors with their own stream: which of them share a launch changes
        nothing any of them plays (tests/test_selfplay_refill.py, tests/test_selfplay_move.py); games finish in the
        order (round, slot) either way.  A search is only queued ahead when the call cannot end before it is consumed,
        so nothing is in flight between calls (weights may change there).
        """
        cfg, B = self.config, self.num_games
        if native_rounds.usable(self):
            # the game steps inside the library: the whole loop below runs there (mzx_selfplay_rounds), same schedule, same
            # draws, same games (mzx/native_rounds.py; config.native_rounds = False keeps this loop on such a game)
            finished, slots = self._native_shard(temperature).play(temperature, temperature_threshold, min_games, max_rounds)
            self.finished_slots += slots
            return finished
        live = self._live
        if live is None:
            spans = self._batched_spans(B)
            groups = []
            for first, last in spans:
                game = self.batched_game if len(spans) == 1 else type(self.batched_game)(self._game_seeds[first:last])
                engine = self.engine if len(spans) == 1 else BatchedMCTS(cfg, self.model, last - first)
                groups.append(_BatchedGroup(self, game, first, last - first, engine, temperature))
            live = self._live = dict(groups=groups)
        groups = live["groups"]
        pipelined = len(groups) > 1
        finished, rounds = [], 0

        def begin(group):
            t0 = time.perf_counter()
            legal = group["game"].legal_actions()
            stacked = group["store"].stacked(None) if group["store"] is not None else group["obs"]
            group["legal"] = legal
            group["pending"] = group["engine"].run(stacked, legal, group["tp"], True, (self.bank, group["streams"]),
                                                   _defer_advance=True, _asynchronous=True)
            self.stats["search_seconds"] += time.perf_counter() - t0

        def consume(group):
            t0 = time.perf_counter()
            pending, group["pending"] = group["pending"], None
            result = pending.result()
            self.stats["search_seconds"] += time.perf_counter() - t0
            n = group["n"]
            self.stats["searches"] += n
            self.stats["simulations"] += n * group["engine"].num_simulations
            done_games, slots = group.play(self, result, temperature, temperature_threshold)
            finished.extend(done_games)
            self.finished_slots += slots

        while len(finished) < min_games and (max_rounds is None or rounds < max_rounds):
            for group in groups:
                if group["pending"] is None:
                    begin(group)
            for k, group in enumerate(groups):
                consume(group)
                later = sum(g["n"] for g in groups[k + 1:])
                if (pipelined and k + 1 < len(groups) and len(finished) + later < min_games
                        and (max_rounds is None or rounds + 1 < max_rounds)):
                    begin(group)      # certain to be consumed by this call: runs while the host plays the groups after it
            rounds += 1
        return finished

    def _native_shard(self, temperature):
        """The actors of a natively played shard (created with the first round; ``temperature``: of the games that start then)."""
        if self._live is None:
            shard = native_rounds.NativeShard(self, temperature)
            self._live = dict(groups=shard.groups, native=shard)
        return self._live["native"]

    def _batched_spans(self, B):
        """
        Slot groups of a shard behind the batched protocol.  ``config.self_play_pipeline`` True / False; unset: two groups
        when the host side of a move weighs as much as its search -- a fully connected network (one whole-search launch
        of a fraction of a millisecond) from 2048 games on.  Residual networks are search-bound (connect4 at 1024 games:
        98 % of the wall is the search, and two searches of 512 trees cost 14 % more than one of 1024): one group.
        """
        want = getattr(self.config, "self_play_pipeline", None)
        if want is None:
            want = self.config.network == "fullyconnected" and B >= 2048
        if not want or B < 2 or not hasattr(self, "_game_seeds"):
            return [(0, B)]
        # ``config.self_play_groups``: how many groups take turns (default two).  A round of k groups lasts max(search + host
        # work of ONE group, k x host work of a group), so a latency-bound search (C2: 0.2 ms for 1024 trees as for 4096)
        # would want more, smaller groups -- measured on the natively played C2 shard (profiles/r06_native_rounds.txt): four
        # groups of 1024 on four streams 5.1 M steps/s against 10.3 M with two (queueing a search took 97 us instead of 40)
        groups = max(2, min(int(getattr(self.config, "self_play_groups", None) or 2), B))
        edges = [(B * g + groups - 1) // groups for g in range(groups + 1)]
        return [(edges[g], edges[g + 1]) for g in range(groups) if edges[g + 1] > edges[g]]

    @staticmethod
    def _stacked_batch(obs_hist, act_hist, k, A):
        """GameHistory.get_stacked_observations(-1, k, A) (self_play.py:513-550) for every game of the shard."""
        current = obs_hist[-1]
        if k == 0:
            return current
        index = len(obs_hist) - 1
        B = current.shape[0]
        plane_shape = (B, 1) + current.shape[2:]
        pieces = [current]
        for past in range(index - 1, index - 1 - k, -1):
            if past >= 0:
                pieces.append(obs_hist[past])
                a = act_hist[past + 1].reshape((B,) + (1,) * (current.ndim - 1))
                pieces.append(numpy.ones(plane_shape, current.dtype) * a / A)
            else:
                pieces.append(numpy.zeros_like(current))
                pieces.append(numpy.zeros(plane_shape, current.dtype))
        return numpy.concatenate(pieces, axis=1)

    def _frame_store(self, num_games):
        """The shard's device frame store: allocated once per actor, rewound for every game."""
        store = getattr(self, "_store", None)
        if store is None or store.G != num_games:
            store = self._store = observations_mod.FrameStore(self.config, num_games, self.model.backend)
        else:
            store.reset()
        return store

    def _drain_searches(self):
        """
        A search a pipelined ``play_rounds`` call queued for its next round and nobody will consume (``play_games`` or
        ``close_game`` follows): waited for, dropped, and the draws it took from its slots' streams are undone.  A failure
        of the worker thread surfaces here instead of disappearing.
        """
        for group in (self._live or {}).get("groups", ()):
            pending = group.get("pending")
            if pending is None or not hasattr(pending, "result"):
                continue
            group["pending"] = None
            try:
                pending.result()
            except Exception as e:       # (the search is being discarded anyway; the failure is not)
                warnings.warn(f"a queued self-play search failed on the worker thread: {e!r}")
            before = group.get("streams_before")
            if before is not None and self.bank is not None:
                for s, state in zip(group["slots"], before):
                    self.bank.set_state(s, state)
            group["streams_before"] = None

    def close_game(self):
        self._drain_searches()
        if self._live and self._live.get("native") is not None:
            self._live["native"].close()
        worker = getattr(self, "_search_worker", None)
        if worker is not None:
            worker.shutdown(wait=True)
            self._search_worker = None
        if self.batched_game is not None:
            for group in (self._live or {}).get("groups", ()):
                if group.get("game") is not None and group["game"] is not self.batched_game:
                    group["game"].close()
            self.batched_game.close()
        for g in self.games:
            g.close()

    def select_opponent_action(self, opponent, stacked_observations, game=None):
        """self_play.py:188-220"""
        game = self.game if game is None else game
        if opponent == "human":
            root, mcts_info = MCTS(self.config).run(self.model, stacked_observations, game.legal_actions(),
                                                    game.to_play(), True)
            print(f'Tree depth: {mcts_info["max_tree_depth"]}')
            print(f"Root value for player {game.to_play()}: {root.value():.2f}")
            print(f"Player {game.to_play()} turn. MuZero suggests {game.action_to_string(self.select_action(root, 0))}")
            return game.human_to_action(), root
        elif opponent == "expert":
            return game.expert_agent(), None
        elif opponent == "random":
            assert game.legal_actions(), f"Legal actions should not be an empty array. Got {game.legal_actions()}."
            assert set(game.legal_actions()).issubset(set(self.config.action_space)), \
                "Legal actions should be a subset of the action space."
            return numpy.random.choice(game.legal_actions()), None
        else:
            raise NotImplementedError(
                'Wrong argument: "opponent" argument should be "self", "human", "expert" or "random"'
            )

    def _select_actions_bank(self, result, searching, temps):
        """
        SelfPlay.select_action (self_play.py:222-245) for all searched games of a move, the draws taken from
        the native stream bank.  Same arithmetic as the reference per game: int32 counts ** (1 / T), the
        sequential Python ``sum``, numpy.random.choice's cumulative sum / renormalisation / right-bisection.
        """
        k = len(searching)
        A = self.engine.A
        if self.engine.fused_move and getattr(result, "legal_array", None) is not None and A <= 4096:
            return self._select_actions_native(result, temps)
        counts = numpy.zeros((k, A), numpy.int32)     # in CHILD order (= legal-action order), zero padded
        n = numpy.empty(k, numpy.int32)
        if isinstance(result.legal_actions, numpy.ndarray):
            legal_arr = result.legal_actions
            n[:] = (legal_arr >= 0).sum(1)
            counts = numpy.where(legal_arr >= 0, numpy.take_along_axis(result.visit_counts, numpy.maximum(legal_arr, 0), 1), 0).astype(numpy.int32)
        elif result.shared_legal is not None:          # one list for every game: gather its columns once
            shared = numpy.asarray(result.shared_legal, numpy.int64)
            n[:] = shared.size
            counts[:, : shared.size] = result.visit_counts[:, shared]
        else:
            for r, legal in enumerate(result.legal_actions):
                n[r] = len(legal)
                counts[r, : n[r]] = result.visit_counts[r][legal]
        actions = numpy.zeros(k, numpy.int64)
        temps = numpy.full(k, float(temps)) if numpy.isscalar(temps) else numpy.asarray(temps, dtype=numpy.float64)
        greedy = numpy.nonzero(temps == 0)[0]
        if greedy.size:
            masked = numpy.where(numpy.arange(A)[None, :] < n[greedy, None], counts[greedy], -1)
            actions[greedy] = numpy.argmax(masked, axis=1)           # first maximum, like numpy.argmax on the child list
        uniform = numpy.nonzero(numpy.isinf(temps))[0]
        if uniform.size:
            actions[uniform] = self.bank.randint([searching[r] for r in uniform], n[uniform])
        rest = numpy.nonzero((temps != 0) & ~numpy.isinf(temps))[0]
        distinct = numpy.unique(temps[rest]) if rest.size else ()
        for t in distinct:
            # the whole-batch shortcut (no index gather) only when ONE temperature covers every game: with several,
            # each game must be drawn exactly once, under its own temperature
            rows = rest if (rest.size == k and len(distinct) == 1) else rest[temps[rest] == t]
            dist = counts[rows] ** (1 / float(t))                    # int32 ** float -> float64 pow, elementwise (numpy's own)
            # sum / normalise / cumulative sum / bisection + the one random_sample() per game: native, one call
            stream = searching if rows.size == k else [searching[r] for r in rows]
            actions[rows] = self.bank.choice_weighted(stream, dist, n[rows])
        if isinstance(result.legal_actions, numpy.ndarray):
            return result.legal_actions[numpy.arange(k), actions].astype(numpy.int64)
        if result.shared_legal is not None:
            return [result.shared_legal[a] for a in actions.tolist()]
        return [result.legal_actions[r][int(actions[r])] for r in range(k)]

    def _select_actions_native(self, result, temps):
        """
        ``mzx_selfplay_select``: the streams consume the search's tie-break words (when ``run`` deferred that), then
        every game's action is drawn -- one pass over the bank on the library's host threads.  The power
        ``visit_count ** (1 / T)`` stays numpy's own: a table over 0 .. num_simulations per temperature, computed
        with the reference's expression (int32 array ** float) and indexed by the library.
        """
        lib, bank = self.model.backend.lib, self.bank
        B, A = result.legal_array.shape
        if numpy.isscalar(temps):
            single, temps = float(temps), numpy.full(B, float(temps))
        else:
            temps = numpy.ascontiguousarray(temps, dtype=numpy.float64)
            first = temps[0]
            single = float(first) if (temps == first).all() else None       # the common case: one temperature for the shard
        if single is not None:
            key = () if single == 0 or math.isinf(single) else (single,)
        else:
            finite = temps[(temps != 0) & ~numpy.isinf(temps)]
            key = tuple(numpy.unique(finite).tolist())
        # power tables (visit_count ** (1 / T) over 0 .. num_simulations + 1, numpy's own pow) per set of temperatures, kept
        stride = self.engine.num_simulations + 2
        peak = int(result.visit_counts.max()) if key else 0
        if peak >= stride:
            stride = peak + 1
        cache = self.__dict__.setdefault("_pow_tables", {})
        entry = cache.get((key, stride))
        if entry is None:
            distinct = numpy.array(key, numpy.float64)
            table = (numpy.ascontiguousarray(numpy.stack([numpy.arange(stride, dtype="int32") ** (1 / t) for t in key]))
                     if key else None)
            entry = cache[(key, stride)] = (table, distinct)
        table, distinct = entry
        mv = _lib.Move()
        mv.num_games, mv.action_space_size, mv.num_threads = B, A, bank.threads
        mv.streams, mv.legal_actions = result.streams.ctypes.data, result.legal_array.ctypes.data
        visits = numpy.ascontiguousarray(result.visit_counts, dtype=numpy.int32)
        actions = numpy.empty(B, numpy.int64)
        words = result.pending_words
        result.pending_words = None
        lib.check(lib.mzx_selfplay_select(bank.handle, ctypes.byref(mv), result.n_legal.ctypes.data,
                                          None if words is None else words.ctypes.data, visits.ctypes.data,
                                          temps.ctypes.data, None if table is None else table.ctypes.data, stride,
                                          distinct.ctypes.data if distinct.size else None, int(distinct.size),
                                          actions.ctypes.data))
        if isinstance(result.legal_actions, numpy.ndarray):
            return actions
        return actions.tolist()

    @staticmethod
    def _select_action(node, temperature, rng):
        visit_counts = numpy.array([child.visit_count for child in node.children.values()], dtype="int32")
        actions = [action for action in node.children.keys()]
        if temperature == 0:
            action = actions[numpy.argmax(visit_counts)]
        elif temperature == float("inf"):
            action = rng.choice(actions)
        else:
            # See paper appendix Data Generation
            dist = visit_counts ** (1 / temperature)
            dist = dist / sum(dist)
            action = rng.choice(actions, p=dist)
        return action

    @staticmethod
    def select_action(node, temperature):
        """self_play.py:222-245 (process-global numpy stream, like the reference)."""
        return SelfPlay._select_action(node, temperature, numpy.random.mtrand._rand)
