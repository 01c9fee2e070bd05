"""
CPU check of the kernel LOGIC: the element functors the HIP kernels are made of
(muzero-general_amd/csrc/mzx_tree.h, mzx_ops.h), compiled serially by
tests/hostcheck, driven through the same C ABI, against traces of the reference.
The GPU twins of these tests live in test_gpu_parity.py.
"""
import pytest

import hostcheck
import lockstep


@pytest.fixture(scope="module")
def backend():
    return hostcheck.backend()


@pytest.mark.parametrize("name", ["cartpole", "tictactoe", "connect4", "cartpole_ties", "tictactoe_custom",
                                  "cartpole_custom", "lunarlander_pretrained"])
def test_lockstep_tree_bit_exact(backend, name):
    got = lockstep.run_fixture(backend, name)
    if name == "cartpole_ties":
        # repeated argmax ties were resolved from the tape: more words consumed than the 1 of sim 0
        assert (got["info"][:, 2] > 3).all()
