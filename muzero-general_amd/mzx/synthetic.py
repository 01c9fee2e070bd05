"""
Synthetic, fixed-shape inputs for benchmarking and parity tests.

Nothing here mirrors a reference file: the reference ships no benchmark.  These
helpers produce (a) deterministic non-trivial weights for any reference-format
``state_dict`` layout, and (b) the synthetic observation batches the metric is
quoted on.  Deterministic given the seed and independent of torch's RNG so the
same tensors can be rebuilt on the GPU box, in the build container and inside
``oracle/make_golden.py``.
"""
import collections

import numpy
import torch


def fill_state_dict(template, seed):
    """
    Return an OrderedDict with the same keys / shapes / dtypes as ``template``
    (a reference-format state_dict or a {name: tensor} spec) filled from
    ``numpy.random.RandomState(seed)`` in key order:
      * ``*.running_var``  U(0.5, 1.5)      * ``*.running_mean``  0.1 N(0,1)
      * BatchNorm weight   U(0.5, 1.5)      * biases              0.1 N(0,1)
      * conv / linear weights  N(0,1) / sqrt(fan_in)
      * integer tensors (``num_batches_tracked``) are zero.
    """
    rs = numpy.random.RandomState(seed)
    out = collections.OrderedDict()
    for name, ref in template.items():
        shape = tuple(ref.shape)
        if not ref.dtype.is_floating_point:
            out[name] = torch.zeros(shape, dtype=ref.dtype)
            continue
        n = int(numpy.prod(shape)) if shape else 1
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            v = rs.uniform(0.5, 1.5, n)
        elif leaf == "running_mean":
            v = 0.1 * rs.standard_normal(n)
        elif leaf == "bias":
            v = 0.1 * rs.standard_normal(n)
        elif len(shape) == 1:  # BatchNorm gamma
            v = rs.uniform(0.5, 1.5, n)
        else:
            fan_in = int(numpy.prod(shape[1:]))
            v = rs.standard_normal(n) / numpy.sqrt(fan_in)
        out[name] = torch.from_numpy(v.astype(numpy.float32).reshape(shape))
    return out


def observations(batch, observation_shape, seed=123):
    """RandomState(seed).rand(B, *observation_shape) as float32."""
    return numpy.random.RandomState(seed).rand(batch, *observation_shape).astype(numpy.float32)


def _hash_u32(x):
    """32-bit integer mix (Wang/Jenkins style) on numpy uint64 lanes, result < 2**32."""
    x = numpy.asarray(x, dtype=numpy.uint64) & numpy.uint64(0xFFFFFFFF)
    x = ((x ^ numpy.uint64(61)) ^ (x >> numpy.uint64(16))) & numpy.uint64(0xFFFFFFFF)
    x = (x * numpy.uint64(9)) & numpy.uint64(0xFFFFFFFF)
    x = x ^ (x >> numpy.uint64(4))
    x = (x * numpy.uint64(0x27D4EB2D)) & numpy.uint64(0xFFFFFFFF)
    x = x ^ (x >> numpy.uint64(15))
    return x


def _hash_u32_wrapping(x):
    """``_hash_u32`` on a uint32 array, IN PLACE: unsigned 32-bit arithmetic wraps by itself, so the masks go and the
    temporaries are half as wide -- the same values (the batched game's step is on the per-round path of the bench)."""
    assert x.dtype == numpy.uint32
    t = x >> numpy.uint32(16)
    x ^= numpy.uint32(61)
    x ^= t
    x *= numpy.uint32(9)
    numpy.right_shift(x, numpy.uint32(4), out=t)
    x ^= t
    x *= numpy.uint32(0x27D4EB2D)
    numpy.right_shift(x, numpy.uint32(15), out=t)
    x ^= t
    return x


def _hash_u32_int(x):
    """``_hash_u32`` on one Python int (same 32-bit mix, no numpy): the per-object game's step is on the
    per-move path of the plugin-surface benchmark, where a numpy call on a 4-element array costs 20x the work."""
    x &= 0xFFFFFFFF
    x = ((x ^ 61) ^ (x >> 16)) & 0xFFFFFFFF
    x = (x * 9) & 0xFFFFFFFF
    x = x ^ (x >> 4)
    x = (x * 0x27D4EB2D) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    return x


def make_synthetic_game(observation_shape, num_actions, num_players=1):
    """
    Build a ``Game`` class with the reference plugin surface
    (games/abstract_game.py:9-105) for a fixed-shape synthetic environment:
    the next observation is a counter hash of (seed, step, action), the reward
    is one hash bit in {0, 1}, every action is always legal and the episode
    never terminates on its own (``config.max_moves`` ends it), so all games of
    a shard stay in lock-step.
    """
    shape = tuple(observation_shape)
    size = int(numpy.prod(shape))
    lane = numpy.arange(size, dtype=numpy.uint64)
    legal = list(range(num_actions))

    class SyntheticGame:
        def __init__(self, seed=None):
            self.seed = 0 if seed is None else int(seed)
            self.t = 0
            self.player = 0
            self.key = self.seed & 0xFFFFFFFF

        def _observation(self):
            if size <= 16:   # small vectors: plain integer arithmetic (identical values; x / 2**32 is exact)
                key = self.key
                return numpy.array([_hash_u32_int(k * 2654435761 + key) / 4294967296.0 for k in range(size)],
                                   dtype=numpy.float32).reshape(shape)
            h = _hash_u32(lane * numpy.uint64(2654435761) + numpy.uint64(self.key))
            return (h.astype(numpy.float64) / 4294967296.0).astype(numpy.float32).reshape(shape)

        def reset(self):
            self.t = 0
            self.player = 0
            self.key = _hash_u32_int(self.seed * 7919 + 17)
            return self._observation()

        def step(self, action):
            self.t += 1
            self.key = _hash_u32_int(self.key * 31 + int(action) * 131 + self.t)
            self.player = (self.player + 1) % num_players
            reward = int(self.key & 1)
            return self._observation(), reward, False

        def to_play(self):
            return self.player

        def legal_actions(self):
            return legal           # every action is always legal: one shared list (callers must not mutate it)

        def render(self):
            print(f"SyntheticGame t={self.t} key={self.key:08x}")

        def close(self):
            pass

        def action_to_string(self, action_number):
            return str(action_number)

    return SyntheticGame


def make_synthetic_batched_game(observation_shape, num_actions, num_players=1):
    """
    The same environment as ``make_synthetic_game`` for a whole shard at once, through the optional BATCHED
    plugin protocol of ``mzx.self_play.SelfPlay`` (class attribute ``batched = True``; ``Game(seeds)``;
    ``reset() -> obs [B, *shape]``; ``step(actions [B], active=None) -> (obs, rewards [B], done [B])``;
    ``legal_actions() -> int32 [B][A]`` padded with -1; ``to_play() -> [B]``).  Game i behaves exactly like
    ``make_synthetic_game(...)(seeds[i])`` -- tests compare the two game by game.
    """
    shape = tuple(observation_shape)
    size = int(numpy.prod(shape))
    lane = (numpy.arange(size, dtype=numpy.uint64) * numpy.uint64(2654435761)).astype(numpy.uint32)     # (mod 2**32)
    u32 = numpy.uint32

    class SyntheticBatchedGame:
        batched = True

        def __init__(self, seeds):
            self.seeds = numpy.asarray([0 if s is None else int(s) for s in seeds], dtype=numpy.uint64)
            self.num_games = int(self.seeds.size)
            self._legal = numpy.tile(numpy.arange(num_actions, dtype=numpy.int32), (self.num_games, 1))
            self._never_done = numpy.zeros(self.num_games, bool)
            self.t = numpy.zeros(self.num_games, numpy.uint32)       # per game: refilled slots restart at 0
            self.player = numpy.zeros(self.num_games, numpy.int64)
            self.key = self.seeds.astype(numpy.uint32)

        # all arithmetic is the per-object game's (Python ints masked to 32 bits) modulo 2**32: uint32 arrays wrap

        def _observation(self, key=None):
            key = self.key if key is None else key
            h = _hash_u32_wrapping(lane[None, :] + key[:, None])
            return (h.astype(numpy.float64) / 4294967296.0).astype(numpy.float32).reshape((key.size,) + shape)

        def _first_key(self, seeds):
            return _hash_u32_wrapping((seeds * numpy.uint64(7919) + numpy.uint64(17)).astype(numpy.uint32))

        def reset(self):
            self.t[:] = 0
            self.player[:] = 0
            self.key = self._first_key(self.seeds)
            return self._observation()

        def reset_games(self, games):
            """Restart only the given games (refill hook of the batched protocol); their first observations."""
            g = numpy.asarray(games, numpy.int64)
            self.t[g] = 0
            self.player[g] = 0
            self.key[g] = self._first_key(self.seeds[g])
            return self._observation(self.key[g])

        def step(self, actions, active=None):
            self.t += u32(1)
            key = self.key * u32(31)
            key += numpy.asarray(actions).astype(numpy.uint32) * u32(131)
            key += self.t
            self.key = _hash_u32_wrapping(key)
            self.player = (self.player + 1) % num_players if num_players > 1 else self.player
            reward = (self.key & u32(1)).astype(numpy.int64)
            return self._observation(), reward, self._never_done

        def to_play(self):
            return self.player

        def legal_actions(self):
            return self._legal

        def close(self):
            pass

    return SyntheticBatchedGame
