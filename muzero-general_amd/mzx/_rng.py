"""
Bank of numpy-compatible random streams for a shard of games (native, include/mzx.h ``mzx_rng_*``).

Stream i behaves bit for bit like ``numpy.random.RandomState(seed_i)`` for the calls one self-play
actor of the reference makes (self_play.py:22 seed, :473 dirichlet, :371 choice(ties) -- drawn on
the device from the tape -- and :229-243 choice(actions[, p])), but all games of a move are served by
ONE call into C++ instead of B Python ``RandomState`` objects.
"""
import ctypes
import os

import numpy

from . import _lib


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class StreamBank:
    def __init__(self, lib, seeds):
        seeds = numpy.ascontiguousarray(seeds, dtype=numpy.uint32)
        self.lib = lib
        self.n = int(seeds.size)
        self.handle = ctypes.c_void_p()
        lib.check(lib.mzx_rng_create(self.n, ctypes.byref(self.handle)))
        lib.check(lib.mzx_rng_seed(self.handle, 0, self.n, _ptr(seeds)))
        self.threads = max(1, min(16, (os.cpu_count() or 1) // 2))     # persistent native workers (csrc/mzx_rng.h)

    def __del__(self):
        try:
            if self.handle:
                self.lib.mzx_rng_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @staticmethod
    def _idx(idx):
        return numpy.ascontiguousarray(idx, dtype=numpy.int32)

    def root_draws(self, idx, alpha, n_legal, action_space_size, tape_words, with_noise=True):
        """(noise [k][A] float64 or None, tape [k][tape_words] uint32) for the games idx."""
        idx = self._idx(idx)
        k = int(idx.size)
        n_legal = numpy.ascontiguousarray(n_legal, dtype=numpy.int32)
        noise = numpy.empty((k, action_space_size), numpy.float64) if with_noise else None
        tape = numpy.empty((k, tape_words), numpy.uint32)
        self.lib.check(self.lib.mzx_rng_root_draws(self.handle, _ptr(idx), k, float(alpha), _ptr(n_legal),
                                                   int(action_space_size), _ptr(noise), int(tape_words), _ptr(tape),
                                                   self.threads))
        return noise, tape

    def advance(self, idx, words):
        idx = self._idx(idx)
        words = numpy.ascontiguousarray(words, dtype=numpy.int32)
        self.lib.check(self.lib.mzx_rng_advance(self.handle, _ptr(idx), int(idx.size), _ptr(words)))

    def random_sample(self, idx):
        idx = self._idx(idx)
        out = numpy.empty(idx.size, numpy.float64)
        self.lib.check(self.lib.mzx_rng_random_sample(self.handle, _ptr(idx), int(idx.size), _ptr(out)))
        return out

    def randint(self, idx, n):
        idx = self._idx(idx)
        n = numpy.ascontiguousarray(n, dtype=numpy.int32)
        out = numpy.empty(idx.size, numpy.int32)
        self.lib.check(self.lib.mzx_rng_randint(self.handle, _ptr(idx), int(idx.size), _ptr(n), _ptr(out)))
        return out

    def choice_weighted(self, idx, weights, n):
        """numpy.random.choice(range(n[k]), p=weights[k] / sum(weights[k])) per stream: positions, int32 [k]."""
        idx = self._idx(idx)
        w = numpy.ascontiguousarray(weights, dtype=numpy.float64)
        n = numpy.ascontiguousarray(n, dtype=numpy.int32)
        out = numpy.empty(idx.size, numpy.int32)
        self.lib.check(self.lib.mzx_rng_choice_weighted(self.handle, _ptr(idx), int(idx.size), _ptr(w), int(w.shape[1]),
                                                        _ptr(n), _ptr(out)))
        return out

    def get_state(self, i):
        """The tuple ``RandomState.get_state()`` returns."""
        key = numpy.empty(624, numpy.uint32)
        pos, has, g = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        self.lib.check(self.lib.mzx_rng_get_state(self.handle, int(i), _ptr(key), ctypes.byref(pos), ctypes.byref(has),
                                                  ctypes.byref(g)))
        return ("MT19937", key, int(pos.value), int(has.value), float(g.value))

    def set_state(self, i, state):
        key = numpy.ascontiguousarray(state[1], dtype=numpy.uint32)
        self.lib.check(self.lib.mzx_rng_set_state(self.handle, int(i), _ptr(key), int(state[2]), int(state[3]),
                                                  float(state[4])))

    def as_random_state(self, i):
        """A numpy RandomState continuing stream i (diagnostics / tests)."""
        rs = numpy.random.RandomState(0)
        rs.set_state(self.get_state(i))
        return rs
