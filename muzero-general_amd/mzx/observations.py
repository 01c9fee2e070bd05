"""
Device-resident observation history of a self-play shard.

``GameHistory.get_stacked_observations`` (/root/reference/self_play.py:513-550) rebuilds the network
input of a position from Python lists on every move: the current observation plus, for each of the
``config.stacked_observations`` previous positions, that observation and a constant action plane.
``FrameStore`` keeps the last ``stacked_observations + 1`` frames and actions of all games of the
shard in HBM -- one upload of the NEW frame per move -- and ``mzx_obs_stack`` (csrc/mzx_obs.h)
assembles the stacked inputs on the device, bit-identical to the reference's array after its
``torch.tensor(...).float()`` (self_play.py:280-285).  ``stack_history`` does the same for every
position of one finished game (what Reanalyse feeds to ``initial_inference``,
replay_buffer.py:343-360).
"""
import ctypes

import numpy
import torch

from . import _lib


def _layout(observation_shape, stacked_observations, action_space_size, num_games, ring):
    L = _lib.ObsLayout()
    L.channels, L.height, L.width = (int(v) for v in observation_shape)
    L.stacked_observations = int(stacked_observations)
    L.action_space_size = int(action_space_size)
    L.num_games = int(num_games)
    L.ring = int(ring)
    return L


def _frames_to_device(backend, array):
    """numpy observations -> fp32 device tensor (the reference's torch.tensor(obs).float())."""
    t = torch.as_tensor(numpy.ascontiguousarray(array))
    return t.to(torch.float32).contiguous().to(backend.device, non_blocking=True)


_STAGE = {}


def _history_to_device(backend, observation_history, shape):
    """
    A game's observation list -> [T + 1, C, H, W] fp32 on the device.  The frames of a long game are hundreds
    of megabytes (breakout: 2 501 x 110 KB): they are converted straight into ONE persistent pinned block -- by a
    few threads, numpy releases the GIL for these copies -- and leave in one asynchronous transfer, instead
    of being concatenated into a pageable array first (two extra passes over the data and a staged copy).
    """
    T1 = len(observation_history)
    if backend.device.type != "cuda" or T1 * int(numpy.prod(shape)) < (1 << 18):
        return _frames_to_device(backend, numpy.array([numpy.asarray(o) for o in observation_history]).reshape((T1,) + shape))
    key = (backend.device, shape)
    block = _STAGE.get(key)
    if block is None or block.shape[0] < T1:
        block = _STAGE[key] = torch.empty((max(T1, 64),) + shape, dtype=torch.float32, pin_memory=True)
    torch.cuda.current_stream(backend.device).synchronize()      # the previous transfer out of this block is done
    view = block.numpy()

    def fill(lo, hi):
        for i in range(lo, hi):
            view[i] = observation_history[i]                     # dtype conversion as torch.tensor(obs).float()

    workers = 8
    if T1 >= 4 * workers:
        from concurrent.futures import ThreadPoolExecutor
        step = (T1 + workers - 1) // workers
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(lambda k: fill(k * step, min(T1, (k + 1) * step)), range(workers)))
    else:
        fill(0, T1)
    return block[:T1].to(backend.device, non_blocking=True)


class FrameStore:
    """frames[ring][num_games][C][H][W] fp32 + actions[ring][num_games] int32 on the backend's device."""

    def __init__(self, config, num_games, backend, ring=None):
        self.backend = backend
        self.shape = tuple(int(v) for v in config.observation_shape)
        self.k = int(config.stacked_observations)
        self.A = len(config.action_space)
        self.G = int(num_games)
        self.ring = int(ring) if ring is not None else self.k + 1
        if self.ring < self.k + 1:
            raise ValueError("ring must hold stacked_observations + 1 frames")
        self.layout = _layout(self.shape, self.k, self.A, self.G, self.ring)
        self.sample_floats = int(backend.lib.mzx_obs_stacked_floats(ctypes.byref(self.layout)))
        if self.sample_floats <= 0:
            raise _lib.MzxError("invalid observation layout")
        self.sample_shape = (self.shape[0] * (self.k + 1) + self.k,) + self.shape[1:]
        self.frames = backend.zeros((self.ring, self.G) + self.shape, torch.float32)
        self.actions = backend.zeros((self.ring, self.G), torch.int32)
        self.time = -1   # history index of the newest frame
        self._out = None

    def reset(self):
        self.time = -1

    def push(self, observations, actions=None):
        """
        Append history index ``time + 1`` for ALL games: observations [G, C, H, W] (any numpy dtype, rows of
        games that no longer play are ignored later) and the actions that led to them (None = the
        leading 0 of action_history, self_play.py:118).
        """
        obs = numpy.asarray(observations)
        if obs.shape != (self.G,) + self.shape:
            raise ValueError(f"expected observations of shape {(self.G,) + self.shape}, got {obs.shape}")
        self.time += 1
        slot = self.time % self.ring
        # one conversion + upload straight into the slot (the reference's torch.tensor(obs).float())
        self.frames[slot].copy_(torch.as_tensor(numpy.ascontiguousarray(obs)))
        if actions is None:
            self.actions[slot].zero_()
        else:
            act = numpy.ascontiguousarray(numpy.asarray(actions).reshape(self.G), dtype=numpy.int32)
            self.actions[slot].copy_(torch.as_tensor(act).to(self.backend.device, non_blocking=True))

    def clear_history(self, games):
        """
        The given games start over (a finished game's slot is refilled, SelfPlay.play_rounds): every frame and action the
        ring holds for them becomes zero -- exactly what get_stacked_observations emits for positions before the start
        of a game (zero frames, zero action planes: self_play.py:541-548) -- while the shard's common history index
        keeps counting.  The next ``push`` writes their first frame (with action 0, the leading entry of action_history).
        """
        idx = numpy.ascontiguousarray(games, dtype=numpy.int64)
        if idx.size == 0:
            return
        if idx.min() < 0 or idx.max() >= self.G:
            raise ValueError("game index out of range")
        d = torch.as_tensor(idx).to(self.backend.device)
        self.frames.index_fill_(1, d, 0.0)
        self.actions.index_fill_(1, d, 0)

    def stacked(self, games=None):
        """get_stacked_observations(-1, k, A) of the given games (default all): device tensor [n, C', H, W]."""
        if self.time < 0:
            raise _lib.MzxError("FrameStore.stacked before the first push")
        be, lib = self.backend, self.backend.lib
        if games is None:
            n, d_game = self.G, None
        else:
            idx = numpy.ascontiguousarray(games, dtype=numpy.int32)
            if idx.size and (idx.min() < 0 or idx.max() >= self.G):
                raise ValueError("game index out of range")
            n = int(idx.size)
            d_game = torch.as_tensor(idx).to(be.device, non_blocking=True)
        if self._out is None:
            self._out = be.empty((self.G,) + self.sample_shape, torch.float32)
        out = self._out[:n]
        # no time array: sample n < num_games reads index time0 + n // num_games = the current one
        lib.check(lib.mzx_obs_stack(ctypes.byref(self.layout), be.ptr(self.frames), be.ptr(self.actions),
                                    be.ptr(d_game), None, self.time, n, be.ptr(out), be.stream()))
        return out


def stack_history(backend, config, observation_history, action_history, count=None):
    """
    get_stacked_observations(i, k, A) for i = 0 .. count-1 of ONE game (default: every position):
    device tensor [count, C', H, W].  The frames are uploaded once ([T+1, C, H, W]).
    """
    shape = tuple(int(v) for v in config.observation_shape)
    T1 = len(observation_history)
    count = T1 if count is None else int(count)
    if count > T1:
        raise ValueError("count exceeds the history length")
    k, A = int(config.stacked_observations), len(config.action_space)
    layout = _layout(shape, k, A, 1, max(T1, 1))
    out = backend.empty((count, shape[0] * (k + 1) + k) + shape[1:], torch.float32)
    if count == 0:
        return out
    frames = _history_to_device(backend, observation_history, shape)
    actions = torch.as_tensor(numpy.asarray([int(a) for a in action_history], dtype=numpy.int32)).to(backend.device)
    backend.lib.check(backend.lib.mzx_obs_stack(ctypes.byref(layout), backend.ptr(frames), backend.ptr(actions), None,
                                                None, 0, count, backend.ptr(out), backend.stream()))
    return out
