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
(``torch.distributed``): the rank next to the trainer holds the real storage, every
refresh is one tiny control all-reduce plus -- only when the trainer published new
weights -- the flat-buffer broadcast.
"""
import torch
import torch.distributed as dist


def broadcast_weights(model, src=0, group=None):
    """
    Broadcast ``model``'s flat weight buffer from rank ``src`` to every rank of
    the process group (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests)
    and rebind derived terms.  A no-op group-wise when torch.distributed is not
    initialised (single GPU).
    """
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
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
    other ranks hold ``None``.  ``refresh(model)`` is the COLLECTIVE every rank calls once per loop iteration
    of ``SelfPlay.continuous_self_play``:

      1. one all-reduce(SUM) of six binary64 words: ``src`` contributes ``training_step`` / ``terminate`` /
         a weights version, every rank its own shard's ``num_played_games`` / ``num_played_steps`` (what its
         replay buffer reported through ``set_info``, replay_buffer.py:63-65) -- afterwards every rank holds the
         same control values (so all ranks leave the loop in the same iteration) and ``src`` publishes the
         job-wide played counts to the real storage (the trainer's ``ratio`` throttle reads them, trainer.py);
      2. when the version moved: ``src`` loads ``storage.get_info("weights")`` into its model and the flat
         fp32 buffer is broadcast (RCCL over xGMI; gloo in the CPU tests), derived terms re-folded.

    ``get_info`` then serves the control keys from the cached copy, identically on every rank.
    """

    CONTROL = ("training_step", "terminate", "num_played_games", "num_played_steps")

    def __init__(self, storage=None, src=0, group=None):
        self.storage, self.src, self.group = storage, src, group
        on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        if (self.rank == src) != (storage is not None):
            raise ValueError("exactly the source rank holds the real storage")
        self.control = {"training_step": 0, "terminate": False, "num_played_games": 0, "num_played_steps": 0}
        self.local = {"num_played_games": 0, "num_played_steps": 0}
        self.version = None          # weights version the model currently holds
        self.refreshes = self.weight_broadcasts = 0

    def _src_get(self, key):
        get = self.storage.get_info
        if hasattr(get, "remote"):
            import ray
            return ray.get(get.remote(key))
        return get(key)

    def _src_set(self, *args):
        put = self.storage.set_info
        return put.remote(*args) if hasattr(put, "remote") else put(*args)

    def refresh(self, model=None):
        device = model.flat_weights().device if model is not None else torch.device("cpu")
        if device.type == "cpu" and self.world > 1 and dist.get_backend(self.group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        word = torch.zeros(6, dtype=torch.float64)
        if self.rank == self.src:
            word[0] = float(self._src_get("training_step"))
            word[1] = 1.0 if self._src_get("terminate") else 0.0
            word[2] = word[0]                       # weights version = the step they were published at
        word[3] = float(self.local["num_played_games"])
        word[4] = float(self.local["num_played_steps"])
        word[5] = 1.0
        word = word.to(device)
        if self.world > 1:
            dist.all_reduce(word, op=dist.ReduceOp.SUM, group=self.group)
        w = word.cpu().tolist()
        assert int(w[5]) == self.world
        self.control = {"training_step": int(w[0]), "terminate": bool(w[1]), "num_played_games": int(w[3]),
                        "num_played_steps": int(w[4])}
        self.refreshes += 1
        if model is not None and self.version != int(w[2]):
            if self.rank == self.src:
                model.set_weights(self._src_get("weights"))
            broadcast_weights(model, src=self.src, group=self.group)
            self.version = int(w[2])
            self.weight_broadcasts += 1
        if self.rank == self.src and self.world > 1:
            self._src_set({"num_played_games": int(w[3]), "num_played_steps": int(w[4])})

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
        self.local.update(played)            # this shard's counts; summed over ranks by the next refresh
        rest = {k: v for k, v in keys.items() if k not in self.local}
        if self.world == 1:
            rest = dict(keys)
        if rest and self.rank == self.src:
            self._src_set(rest)
