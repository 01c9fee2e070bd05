"""
The natively stepped games (include/mzx.h mzx_game_*, csrc/mzx_games.h) against the Python classes they restate -- which
tests/test_batched_games.py in turn holds against the reference's game files: game i of a native shard must emit,
observation for observation (values AND dtype), reward for reward, legal list for legal list, what game i of the Python
batched class emits for the same actions, including restarts of single slots (the refill hook).  CPU test on the serial
build of the ABI (host code: the same source the device library compiles).
"""
import numpy
import pytest

import hostcheck
from mzx import games, synthetic


@pytest.fixture(scope="module")
def backend():
    return hostcheck.backend()


def _play_side_by_side(native, python, rounds, rs, restart_every=0):
    B = python.num_games
    o1, o2 = native.reset(), python.reset()
    assert o1.dtype == o2.dtype and o1.shape == o2.shape and numpy.array_equal(o1, o2)
    finished = 0
    for r in range(rounds):
        l1, l2 = native.legal_actions(), python.legal_actions()
        assert l1.dtype == numpy.int32 and numpy.array_equal(l1, l2), r
        assert numpy.array_equal(numpy.asarray(native.to_play()), numpy.asarray(python.to_play())), r
        n = (l2 >= 0).sum(1)
        assert (n > 0).all()
        actions = l2[numpy.arange(B), (rs.randint(0, 1 << 30, size=B) % n)]
        o1, r1, d1 = native.step(actions)
        o2, r2, d2 = python.step(actions)
        assert o1.dtype == o2.dtype and numpy.array_equal(o1, o2), r
        assert numpy.array_equal(numpy.asarray(r1, numpy.float64), numpy.asarray(r2, numpy.float64)), r
        assert numpy.asarray(r1).dtype == numpy.asarray(r2).dtype
        assert numpy.array_equal(numpy.asarray(d1, bool), numpy.asarray(d2, bool)), r
        over = numpy.nonzero(numpy.asarray(d2, bool))[0]
        if restart_every and r % restart_every == restart_every - 1:
            over = numpy.union1d(over, rs.choice(B, size=max(1, B // 7), replace=False))
        if over.size:
            finished += int(over.size)
            f1, f2 = native.reset_games(over), python.reset_games(over)
            assert f1.dtype == f2.dtype and numpy.array_equal(f1, f2), r
    return finished


@pytest.mark.parametrize("name", ["tictactoe", "connect4", "gomoku"])
def test_native_board_games_equal_the_python_classes(backend, name):
    B = 37
    games.NativeBatchedGame.backend = backend
    native, python = games.NATIVE[name](list(range(B))), games.BATCHED[name](list(range(B)))
    finished = _play_side_by_side(native, python, {"tictactoe": 60, "connect4": 200, "gomoku": 400}[name],
                                  numpy.random.RandomState(5))
    assert finished > B       # every slot ended at least once on average: wins, full boards, restarts
    native.close()


@pytest.mark.parametrize("shape,A,players", [((1, 1, 4), 2, 1), ((2, 1, 3), 3, 2), ((3, 5, 7), 6, 1)])
def test_native_synthetic_game_equals_the_python_class(backend, shape, A, players):
    games.NativeBatchedGame.backend = backend
    seeds = [0, 1, 2, 7919, 2 ** 32 - 1, 12345678, 4000000000] + list(range(100, 130))
    native = games.make_native_synthetic_game(shape, A, players)(seeds)
    python = synthetic.make_synthetic_batched_game(shape, A, players)(seeds)
    _play_side_by_side(native, python, 70, numpy.random.RandomState(9), restart_every=11)
    native.close()


def test_native_game_rejects_bad_arguments(backend):
    import ctypes

    lib = backend.lib
    h = ctypes.c_void_p()
    assert lib.mzx_game_create(b"chess", 4, None, None, 0, 0, ctypes.byref(h)) != 0
    assert b"unknown game" in lib.mzx_last_error()
    assert lib.mzx_game_create(b"synthetic", 4, None, None, 2, 1, ctypes.byref(h)) != 0
    games.NativeBatchedGame.backend = backend
    g = games.TicTacToeNative([0, 1])
    with pytest.raises(Exception):
        g.step([0, 9])           # outside the action space
    with pytest.raises(Exception):
        g.reset_games([2])
