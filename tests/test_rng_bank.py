"""
The native stream bank (include/mzx.h mzx_rng_*, csrc/mzx_rng.h) against numpy.random.RandomState:
every draw the self-play actor makes must be bit-identical to numpy's legacy stream.  Host-only
entry points: runs without a GPU, through the PRODUCT library (libmzx.so loads on any host).
"""
import numpy
import pytest

from mzx import _lib, _rng


@pytest.fixture(scope="module")
def lib():
    return _lib.Library(_lib.LIB_PATH)


SEEDS = [0, 1, 2, 7, 1234, 99999, 2 ** 31, 2 ** 32 - 1]


def test_seeding_and_state_round_trip(lib):
    bank = _rng.StreamBank(lib, SEEDS)
    for i, s in enumerate(SEEDS):
        ref = numpy.random.RandomState(s).get_state()
        got = bank.get_state(i)
        assert numpy.array_equal(got[1], ref[1]) and got[2:] == ref[2:]
    rs = numpy.random.RandomState(5)
    rs.standard_normal(3)     # leaves a cached gaussian
    bank.set_state(2, rs.get_state())
    assert numpy.array_equal(bank.as_random_state(2).random_sample(5), rs.random_sample(5))


@pytest.mark.parametrize("alpha", [0.03, 0.1, 0.25, 0.3, 1.0, 1.7, 10.0])
def test_root_draws_match_numpy(lib, alpha):
    A, W, moves = 9, 16, 40
    bank = _rng.StreamBank(lib, SEEDS)
    refs = [numpy.random.RandomState(s) for s in SEEDS]
    rs = numpy.random.RandomState(0)
    idx = numpy.arange(len(SEEDS))
    for _ in range(moves):
        n_legal = rs.randint(1, A + 1, size=len(SEEDS))
        noise, tape = bank.root_draws(idx, alpha, n_legal, A, W)
        used = rs.randint(0, 5, size=len(SEEDS))
        for i, ref in enumerate(refs):
            want = ref.dirichlet([alpha] * int(n_legal[i]))
            assert numpy.array_equal(noise[i, : n_legal[i]].view(numpy.int64), want.view(numpy.int64)), (alpha, i)
            assert (noise[i, n_legal[i]:] == 0).all()
            state = ref.get_state()
            assert numpy.array_equal(tape[i], ref.randint(0, 2 ** 32, size=W, dtype=numpy.uint32))
            ref.set_state(state)
            if used[i]:
                ref.randint(0, 2 ** 32, size=int(used[i]), dtype=numpy.uint32)
        bank.advance(idx, used)
    for i, ref in enumerate(refs):
        got, want = bank.get_state(i), ref.get_state()
        assert numpy.array_equal(got[1], want[1]) and got[2] == want[2]


def test_action_draws_match_numpy_choice(lib):
    bank = _rng.StreamBank(lib, SEEDS)
    refs = [numpy.random.RandomState(s) for s in SEEDS]
    idx = numpy.arange(len(SEEDS))
    rs = numpy.random.RandomState(3)
    for _ in range(200):
        n = rs.randint(1, 20, size=len(SEEDS))
        got = bank.randint(idx, n)
        for i, ref in enumerate(refs):
            assert got[i] == ref.choice(list(range(100, 100 + n[i]))) - 100
        u = bank.random_sample(idx)
        for i, ref in enumerate(refs):
            assert u[i] == ref.random_sample()
    # subset of streams, arbitrary order
    sub = numpy.array([5, 1, 6])
    u = bank.random_sample(sub)
    for k, i in enumerate(sub):
        assert u[k] == refs[i].random_sample()


def test_no_noise_draws_nothing(lib):
    bank = _rng.StreamBank(lib, [11])
    noise, tape = bank.root_draws([0], 0.25, [3], 4, 8, with_noise=False)
    assert noise is None
    ref = numpy.random.RandomState(11)
    assert numpy.array_equal(tape[0], ref.randint(0, 2 ** 32, size=8, dtype=numpy.uint32))


@pytest.mark.timeout(120)
def test_two_threads_call_into_one_bank(lib):
    """
    The pipelined per-object shard calls into the bank from TWO threads at once (the queued search's root draws on the
    worker, the other group's action draw on the main thread), both large enough for the bank's thread pool: the pool
    serves one caller, the other does its range itself -- no deadlock (round 5 found one on the device), and every
    stream produces what it produces alone.
    """
    import ctypes
    import threading

    from mzx import _lib as lib_mod

    B, A, rounds = 1024, 7, 200
    half = B // 2
    banks = [_rng.StreamBank(lib, list(range(B))) for _ in range(2)]
    for b in banks:
        b.threads = 4
    idx_a, idx_b = numpy.arange(half, dtype=numpy.int32), numpy.arange(half, B, dtype=numpy.int32)
    n = numpy.full(half, A, numpy.int32)
    legal = numpy.ascontiguousarray(numpy.tile(numpy.arange(A, dtype=numpy.int32), (half, 1)))
    visits = numpy.random.RandomState(0).randint(0, 30, size=(half, A)).astype(numpy.int32)
    temps, table, table_t = numpy.ones(half), numpy.arange(64, dtype="int32") ** 1.0, numpy.array([1.0])

    def draws(bank, out):
        for _ in range(rounds):
            out.append(bank.root_draws(idx_a, 0.3, n, A, 16, True))

    def selects(bank, out):
        mv = lib_mod.Move()
        mv.num_games, mv.action_space_size, mv.num_threads = half, A, 4
        mv.streams, mv.legal_actions = idx_b.ctypes.data, legal.ctypes.data
        for _ in range(rounds):
            act = numpy.empty(half, numpy.int64)
            lib.check(lib.mzx_selfplay_select(bank.handle, ctypes.byref(mv), n.ctypes.data, None, visits.ctypes.data,
                                              temps.ctypes.data, table.ctypes.data, 64, table_t.ctypes.data, 1, act.ctypes.data))
            out.append(act)

    together, alone = ([], []), ([], [])
    threads = [threading.Thread(target=draws, args=(banks[0], together[0])),
               threading.Thread(target=selects, args=(banks[0], together[1]))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    draws(banks[1], alone[0])
    selects(banks[1], alone[1])
    for (na, ta), (nb, tb) in zip(together[0], alone[0]):
        assert numpy.array_equal(na.view(numpy.int64), nb.view(numpy.int64)) and numpy.array_equal(ta, tb)
    for x, y in zip(together[1], alone[1]):
        assert numpy.array_equal(x, y)
