"""
Weight hand-off between the trainer and the self-play shards.

Reference: /root/reference/shared_storage.py:7-40 is a Ray actor holding the
checkpoint dict; every SelfPlay actor pulls the full ``state_dict`` through the
object store before each game (self_play.py:37).  Here each GPU runs one
self-play process; the only exchange step of the hot path is this weight
refresh, done as ONE RCCL broadcast (over xGMI) of the network's flat fp32
buffer (parameters + BatchNorm running statistics; 6 KB for CartPole ... 2.9 MB
for Connect4), followed by re-deriving the folded BatchNorm terms on each rank.
No other collective exists on the path.

``LocalStorage`` offers the ``get_info`` / ``set_info`` duck type for the
single-process / test case; with Ray installed the reference's own actor can be
passed to ``SelfPlay.continuous_self_play`` instead.  ``ShardedStorage`` is what
``continuous_self_play`` talks to when the job runs one self-play process per GPU
(``torch.distributed``): the rank next to the trainer holds the real storage; control
values travel in one tiny ASYNCHRONOUS all-reduce at a time (no rank ever waits for
another inside the actor loop) and -- only when the trainer published new weights --
the flat-buffer broadcast.
"""
import torch
import torch.distributed as dist


def broadcast_weights(model, src=0, group=None, with_one_rank=False):
    """
    Broadcast ``model``'s flat weight buffer from rank ``src`` to every rank of
    the process group (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)
    and rebind derived terms.  A no-op group-wise when torch.distributed is not
    initialised (single GPU).  ``with_one_rank``: a group of ONE rank issues the
    collective too (the 1-GPU box's check that the flat buffer is something RCCL
    takes, tests/test_gpu_rccl.py); by default a lone rank skips it.
    """
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or with_one_rank):
        dist.broadcast(model.flat_weights(), src=src, group=group)
    model.refresh_derived()


def broadcast_flat(flat, src=0, group=None):
    """The collective alone, on any fp32 tensor (used by the gloo tests)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def shard_seeds(base_seed, games_per_rank, rank=None):
    """
    Seeds of this rank's games: rank r owns games [r*G, (r+1)*G) seeded
    ``base_seed + r*G + i`` -- the reference seeds worker k with ``config.seed + k``
    (muzero.py:185).
    """
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return [base_seed + rank * games_per_rank + i for i in range(games_per_rank)]


class LocalStorage:
    """
    In-process stand-in for the storage actor ``continuous_self_play`` talks to
    (duck type of shared_storage.py:7-40: ``get_info`` / ``set_info``); entries
    are plain attributes of one dict, no Ray, no deep copies.
    """

    def __init__(self, **entries):
        self._entries = dict(entries)

    def get_info(self, keys):
        if isinstance(keys, (list, tuple)):
            return {k: self._entries[k] for k in keys}
        if not isinstance(keys, str):
            raise TypeError(f"keys must be str or list, got {type(keys).__name__}")
        return self._entries[keys]

    def set_info(self, keys, values=None):
        if isinstance(keys, dict):
            self._entries.update(keys)
        elif isinstance(keys, str) and values is not None:
            self._entries[keys] = values
        else:
            raise TypeError("set_info(key, value) or set_info({key: value, ...})")


class ShardedStorage:
    """
    The storage one rank of a sharded self-play job sees (duck type of shared_storage.py:7-40).

    The reference's N self-play actors each call the storage actor (self_play.py:33-37, :93-107) and the
    replay-buffer actor; here rank ``src`` (the process next to the trainer) holds the real ``storage`` and the
    other ranks hold ``None``.  The ranks exchange through a SEQUENCE of collectives that every rank issues in the
    same order but AT ITS OWN PACE -- ``refresh(model)``, called once per loop iteration of
    ``SelfPlay.continuous_self_play``, never waits for another rank:

      * control exchange #n: one ASYNCHRONOUS all-reduce(SUM) of six binary64 words -- ``src`` contributes
        ``training_step`` / ``terminate`` / the weights version, every rank its own shard's ``num_played_games`` /
        ``num_played_steps`` (what its replay buffer reported through ``set_info``, replay_buffer.py:63-65).  A rank
        issues #n + 1 only after it has seen #n complete, so at most one exchange is in flight and the sequence
        numbers agree across ranks; a rank whose neighbours have not issued #n yet simply keeps playing games on
        the control values and weights it has (shards with uneven game lengths do not lock-step);
      * when exchange #n carries a new weights version, every rank issues -- again when IT gets there -- one
        asynchronous broadcast of the flat fp32 buffer into a staging tensor (RCCL over xGMI; gloo in the CPU
        tests; ``src`` loads ``storage.get_info("weights")`` first); the model takes the staged weights over at the
        next loop boundary, never under a running search;
      * every rank evaluates the stop condition (``training_step >= training_steps`` or ``terminate``) on the SAME
        exchange, so all ranks issue exactly the same collectives and ``finish()`` (a blocking drain) returns.

    ``get_info`` serves the control keys from the last completed exchange.  The weights version is the trainer's
    publication count ``training_step // checkpoint_interval`` (trainer.py publishes every ``checkpoint_interval``
    steps; the step itself moves all the time), plus -- for storages living in this process -- the identity of the
    weights object, so that a checkpoint loaded without a step change is seen too.
    """

    CONTROL = ("training_step", "terminate", "num_played_games", "num_played_steps")

    def __init__(self, storage=None, src=0, group=None, checkpoint_interval=None, training_steps=None,
                 collectives_with_one_rank=False):
        self.storage, self.src, self.group = storage, src, group
        on = dist.is_available() and dist.is_initialized()
        # a world of ONE rank has nobody to exchange with and skips the collectives; with this flag it issues them
        # all the same and waits for each on the spot (same flow otherwise): the 1-GPU box's check of the RCCL side
        # of this class -- device control words, the staged broadcast (tests/test_gpu_rccl.py)
        self._lone_collectives = bool(collectives_with_one_rank) and on
        self.lone_collectives_issued = 0
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        if (self.rank == src) != (storage is not None):
            raise ValueError("exactly the source rank holds the real storage")
        self.checkpoint_interval = checkpoint_interval      # None: continuous_self_play fills it from its config
        self.training_steps = training_steps                # idem: the stop condition every rank evaluates alike
        self.control = {"training_step": 0, "terminate": False, "num_played_games": 0, "num_played_steps": 0}
        self.local = {"num_played_games": 0, "num_played_steps": 0}
        self.version = None          # weights version the model currently holds
        self.refreshes = self.weight_broadcasts = self.polls_without_progress = 0
        self._pending = None         # ("control", work, tensor) | ("weights", work, staging tensor, version)
        self._stopped = False        # the stop condition was seen: no further collectives are issued
        self._weights_ident = None
        self._publications = 0

    # ---- the real storage (source rank only)
    def _src_get(self, key):
        get = self.storage.get_info
        if hasattr(get, "remote"):
            import ray
            return ray.get(get.remote(key))
        return get(key)

    def _src_set(self, *args):
        put = self.storage.set_info
        return put.remote(*args) if hasattr(put, "remote") else put(*args)

    def _src_version(self, training_step):
        """Publication counter of the trainer's weights (source rank)."""
        version = int(training_step) // max(1, int(self.checkpoint_interval or 1))
        if not hasattr(self.storage.get_info, "remote"):     # same process: notice a replaced weights object, no copy
            ident = id(self._src_get("weights"))
            if ident != self._weights_ident:
                self._weights_ident = ident
                self._publications += 1
            version = version * 1000003 + self._publications
        return version

    def _finished(self):
        if self.control["terminate"]:
            return True
        return self.training_steps is not None and self.control["training_step"] >= self.training_steps

    # ---- the collective sequence
    def _device(self, model):
        device = model.flat_weights().device if model is not None else torch.device("cpu")
        if device.type == "cpu" and self._collective() and dist.get_backend(self.group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        return device

    def _collective(self):
        return self.world > 1 or self._lone_collectives

    def _lone_wait(self, work):
        """A lone rank's collective (``collectives_with_one_rank``) is waited for where it is issued."""
        if work is not None and self.world == 1:
            work.wait()
            self.lone_collectives_issued += 1
            return None
        return work

    def _issue_control(self, model):
        word = torch.zeros(6, dtype=torch.float64)
        if self.rank == self.src:
            step = self._src_get("training_step")
            word[0] = float(step)
            word[1] = 1.0 if self._src_get("terminate") else 0.0
            word[2] = float(self._src_version(step))
        word[3] = float(self.local["num_played_games"])
        word[4] = float(self.local["num_played_steps"])
        word[5] = 1.0
        word = word.to(self._device(model))
        work = dist.all_reduce(word, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self._collective() else None
        self._pending = ("control", self._lone_wait(work), word)

    def _issue_weights(self, model, version):
        if self.rank == self.src:
            model.set_weights(self._src_get("weights"))      # between searches: refresh() runs at loop boundaries
            staging = model.flat_weights().clone()
        else:
            staging = torch.empty_like(model.flat_weights())
        work = dist.broadcast(staging, src=self.src, group=self.group, async_op=True) if self._collective() else None
        self._pending = ("weights", self._lone_wait(work), staging, version)

    def _complete(self, model):
        """Consumes the completed pending collective; returns True when another one was issued right away."""
        kind = self._pending[0]
        if kind == "control":
            w = self._pending[2].cpu().tolist()
            self._pending = None
            assert int(w[5]) == self.world
            self.control = {"training_step": int(w[0]), "terminate": bool(w[1]), "num_played_games": int(w[3]),
                            "num_played_steps": int(w[4])}
            self.refreshes += 1
            if self.rank == self.src and self.world > 1:
                self._src_set({"num_played_games": int(w[3]), "num_played_steps": int(w[4])})
            if model is not None and self.version != int(w[2]):
                self._issue_weights(model, int(w[2]))
                return True
            if self._finished():
                self._stopped = True
            return False
        _, _, staging, version = self._pending
        self._pending = None
        if self.rank != self.src:
            model.flat_weights().copy_(staging)
        model.refresh_derived()
        self.version = version
        self.weight_broadcasts += 1
        if self._finished():
            self._stopped = True
        return False

    def refresh(self, model=None, block=False):
        """
        One step of the exchange at a loop boundary of the actor: consume the pending collective if it has completed
        (``block``: wait for it -- the first refresh of a run, which must deliver the trainer's weights), issue the
        next one.  Returns immediately otherwise; the caller keeps playing on what it has.
        """
        if self._stopped:
            return
        if self._pending is None:
            self._issue_control(model)
        while self._pending is not None:
            work = self._pending[1]
            if work is not None:
                if block:
                    work.wait()
                elif not work.is_completed():
                    self.polls_without_progress += 1
                    return
            if self._complete(model):
                continue                       # a weights broadcast follows its control exchange immediately
            break
        if block and not self._stopped and self._pending is None:
            return
        if not self._stopped and self._pending is None and not block and self.world > 1:
            # in flight while the next games are played.  A single process has nobody to wait for: its exchange is
            # issued AND consumed inside one refresh(), so control values and weights are never one refresh stale
            self._issue_control(model)

    def finish(self, model=None):
        """Blocking drain at the end of the actor loop: every rank has issued the same collectives, so this returns."""
        while self._pending is not None:
            work = self._pending[1]
            if work is not None:
                work.wait()
            if not self._complete(model):
                break
        while not self._stopped and self.world > 1:      # left the loop for a local reason: run the sequence out
            self.refresh(model, block=True)
            if self._pending is None and not self._stopped:
                self._issue_control(model)

    def get_info(self, keys):
        if isinstance(keys, (list, tuple)):
            return {k: self.get_info(k) for k in keys}
        if keys in self.control:
            return self.control[keys]
        if self.rank == self.src:
            return self._src_get(keys)
        raise KeyError(f"{keys!r} is only known to the source rank (weights arrive through refresh())")

    def set_info(self, keys, values=None):
        if isinstance(keys, str) and values is not None:
            keys = {keys: values}
        elif not isinstance(keys, dict):
            raise TypeError("set_info(key, value) or set_info({key: value, ...})")
        played = {k: v for k, v in keys.items() if k in self.local}
        self.local.update(played)            # this shard's counts; summed over ranks by the next exchange
        rest = {k: v for k, v in keys.items() if k not in self.local}
        if self.world == 1:
            rest = dict(keys)
        if rest and self.rank == self.src:
            self._src_set(rest)
