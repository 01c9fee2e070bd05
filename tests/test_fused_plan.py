"""
Host-side planning of the fused residual engine (csrc/mzx_resnet_fused.h: rz_plan) through the C ABI,
no GPU: which configurations are fused, that the derived buffer grows by the packed weight images, and
the packing functors (run serially by tests/hostcheck at set_weights) against a numpy restatement of the
fragment order documented in mzx_resnet_plan.h.
"""
import ctypes

import numpy
import pytest
import torch

import hostcheck
from mzx import _lib, configs, models, synthetic


@pytest.fixture(scope="module")
def lib():
    return _lib.Library(_lib.LIB_PATH)


def _create(lib, cfg):
    c = models.net_config_from(cfg)
    h = ctypes.c_void_p()
    lib.check(lib.mzx_net_create(ctypes.byref(c), ctypes.byref(h)))
    return h


def test_which_networks_are_fused(lib):
    for name, want in (("cartpole", 0), ("tictactoe", 3), ("connect4", 3), ("breakout", 3)):
        h = _create(lib, configs.BY_NAME[name]())
        assert lib.mzx_net_fused_supported(h) == want, name
        lib.mzx_net_destroy(h)
    # a board too large for one workgroup's LDS image falls back to the per-operator engine
    h = _create(lib, configs.connect4(observation_shape=(3, 19, 19), action_space=list(range(361))))
    assert lib.mzx_net_fused_supported(h) == 0
    unfused = lib.mzx_net_derived_floats(h)          # folded BatchNorm terms + the streamed engine's fragment images
    lib.mzx_net_destroy(h)
    h = _create(lib, configs.connect4())
    assert lib.mzx_net_derived_floats(h) > unfused + 700000   # + the fused engine's images of both programs (26 conv layers x 36 864 floats; the 19 x 19 net's wider head MLPs take ~170 k of the difference back)
    lib.mzx_net_destroy(h)


def test_schedule_puts_independent_operators_into_shared_slots(lib):
    """
    rz_schedule: the reward head runs beside the prediction trunk and the value / policy towers beside each
    other wherever the layer GEMMs are small enough to share a workgroup (tic-tac-toe, breakout); connect4's wide
    trunk convolutions keep the whole workgroup, only its heads pair up.  Every operator runs after everything it
    reads: the slot index never decreases along a chain of the program.
    """
    want = {"tictactoe": ((12, 9), (15, 9)), "connect4": ((20, 17), (23, 20)), "breakout": ((35, 12), (19, 13))}
    for name, per_program in want.items():
        h = _create(lib, configs.BY_NAME[name]())
        for recurrent, (n_ops, n_slots) in enumerate(per_program):
            assert lib.mzx_net_num_operators(h, recurrent) == n_ops
            slot = (ctypes.c_int32 * n_ops)()
            assert lib.mzx_net_fused_schedule(h, recurrent, slot, n_ops) == n_slots, (name, recurrent)
            fused = [s for s in slot if s >= 0]
            assert fused[0] == 0 and max(fused) == n_slots - 1 and sorted(set(fused)) == list(range(n_slots))
            first = list(slot).index(0)
            assert all(s == -1 for s in slot[:first])            # the down-sampling stem precedes the fused part
            assert all(slot[first + k] <= k for k in range(len(fused)))   # operators only move EARLIER than program order
            assert max(fused.count(s) for s in set(fused)) <= 3
        assert lib.mzx_net_fused_schedule(h, 0, slot, 1) == 0      # capacity too small
        lib.mzx_net_destroy(h)
    h = _create(lib, configs.cartpole())
    assert lib.mzx_net_fused_schedule(h, 1, (ctypes.c_int32 * 16)(), 16) == 0   # fully connected: no fused programs
    lib.mzx_net_destroy(h)


def test_flops_match_survey(lib):
    # 2 x MAC per simulation / per initial inference
    want = {"cartpole": (1312, 2752), "tictactoe": (187968, 231504), "connect4": (37372160, 40396160),
            "breakout": (34192160, 1532480)}
    for name, (fi, fr) in want.items():
        h = _create(lib, configs.BY_NAME[name]())
        assert (lib.mzx_net_flops(h, 0), lib.mzx_net_flops(h, 1)) == (fi, fr)
        lib.mzx_net_destroy(h)


def test_packed_fragments_follow_the_documented_order():
    """
    tests/hostcheck runs RzPackOp serially at set_weights: find the first 3x3 convolution of tic-tac-toe's
    recurrent program in the derived buffer and compare with the lane order of mzx_resnet_plan.h:
    [column tile][chunk = (tap, 16-channel chunk)][lane = 16 g + n][j] = W[n][16 cc + 4 g + j][tap].
    """
    be = hostcheck.backend()
    cfg = configs.tictactoe()
    net = models.MuZeroNetwork(cfg, _backend=be)
    sd = synthetic.fill_state_dict(net.state_dict(), 9)
    net.set_weights(sd)
    derived = net._derived.numpy()
    W = sd["dynamics_network.module.resblocks.0.conv1.weight"].numpy()   # [16][16][3][3]
    cout, cin = W.shape[:2]
    nchunks = 9 * ((cin + 15) // 16)
    want = numpy.zeros((nchunks + (nchunks & 1), 64, 4), numpy.float32)
    for c in range(nchunks):
        tap, cc = c // ((cin + 15) // 16), c % ((cin + 15) // 16)
        for lane in range(64):
            g, n = lane >> 4, lane & 15
            for j in range(4):
                ci = 16 * cc + 4 * g + j
                if ci < cin and n < cout:
                    want[c, lane, j] = W[n, ci, tap // 3, tap % 3]
    flat = want.reshape(-1)
    # locate the image (unique, non-trivial content) inside the derived buffer
    hits = [o for o in range(0, derived.size - flat.size + 1, 4) if derived[o] == flat[0] and
            numpy.array_equal(derived[o:o + flat.size], flat)]
    assert len(hits) >= 1
