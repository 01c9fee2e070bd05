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
passed to ``SelfPlay.continuous_self_play`` instead.
"""
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
