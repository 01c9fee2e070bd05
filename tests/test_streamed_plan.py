"""
Host-side planning of the streamed MFMA engine (csrc/mzx_resnet_batched.h: rb_plan) through the C ABI, no GPU:
which of the reference's configurations go there (games/gomoku.py, games/atari.py as shipped), that every planned
workgroup tile fits the LDS budget and the 144-row / 9-tile limits of rb_gemm_kernel, and that the tiling covers
every output position of every layer.
"""
import ctypes

import pytest

from mzx import _lib, configs, models

FIELDS = models.HipNetwork.STREAMED_PLAN_FIELDS


@pytest.fixture(scope="module")
def lib():
    return _lib.Library(_lib.LIB_PATH)


def _create(lib, cfg):
    c = models.net_config_from(cfg)
    h = ctypes.c_void_p()
    lib.check(lib.mzx_net_create(ctypes.byref(c), ctypes.byref(h)))
    return h


def _plan(lib, h, recurrent, op):
    out = (ctypes.c_int32 * 24)()
    lib.check(lib.mzx_net_streamed_plan(h, recurrent, op, ctypes.byref(out)))
    return dict(zip(FIELDS, list(out)))


def test_which_networks_are_streamed(lib):
    for name, fused, streamed in (("cartpole", 0, 0), ("tictactoe", 3, 0), ("connect4", 3, 0), ("breakout", 3, 0),
                                  ("gomoku", 0, 3), ("atari", 0, 3)):
        h = _create(lib, configs.BY_NAME[name]())
        assert lib.mzx_net_fused_supported(h) == fused, name
        assert lib.mzx_net_streamed_supported(h) == streamed, name
        lib.mzx_net_destroy(h)
    # a 64-channel network on a 19 x 19 board: too large for the LDS engine, streamed
    h = _create(lib, configs.connect4(observation_shape=(3, 19, 19), action_space=list(range(361))))
    assert lib.mzx_net_fused_supported(h) == 0 and lib.mzx_net_streamed_supported(h) == 3
    lib.mzx_net_destroy(h)


@pytest.mark.parametrize("name", ["gomoku", "atari", "go19", "wide_heads", "odd_channels", "breakout_cnn"])
def test_streamed_tiles_fit_the_kernel(lib, name):
    cfg = {
        "gomoku": configs.gomoku, "atari": configs.atari,
        "go19": lambda: configs.connect4(observation_shape=(3, 19, 19), action_space=list(range(361))),
        "wide_heads": lambda: configs.gomoku(reduced_channels_value=48, reduced_channels_policy=33, support_size=300,
                                             resnet_fc_value_layers=[200, 77]),
        "odd_channels": lambda: configs.gomoku(channels=70, observation_shape=(5, 13, 9), action_space=list(range(117)),
                                               stacked_observations=2),
        # DownsampleCNN (models.py:278-297): 12 x 12 stride-4 and 5 x 5 convolutions, max / adaptive pooling
        "breakout_cnn": lambda: configs.breakout(downsample="CNN"),
    }[name]()
    h = _create(lib, cfg)
    assert lib.mzx_net_streamed_supported(h) == (0 if name == "breakout_cnn" else 3)   # (the small trunk fits the LDS engine)
    derived = lib.mzx_net_derived_floats(h)
    gemm_weights = 0
    for recurrent in (0, 1):
        n = lib.mzx_net_num_operators(h, recurrent)
        kinds = []
        for op in range(n):
            p = _plan(lib, h, recurrent, op)
            kinds.append(p["kind"])
            if p["kind"] != 0:
                continue
            ksize = int(round(p["taps"] ** 0.5))     # square kernels: 1, 3 (trunk), 5 / 2 ceil(H / 16) (DownsampleCNN)
            assert ksize * ksize == p["taps"]
            assert p["lds_bytes"] <= 78 * 1024
            # nine row tiles per wave; the waves the column tiles leave over (8 / WN, at most 4 deep) split the rows
            ntiles = (p["cout"] + 15) // 16
            wn = min(8, -(-min(ntiles, 16) // (2 if ntiles > 8 else 1)))
            deep = min(4, max(1, 8 // wn)) if (p["taps"] > 1 and ntiles >= 2) else 1      # trunk convolutions only
            assert p["rows"] == p["T"] * p["th"] * p["tw"] <= 144 * deep
            assert p["mtiles"] == (p["rows"] + 15) // 16 <= 9 * deep
            assert p["PH"] == (p["th"] - 1) * p["stride"] + ksize and p["PW"] == (p["tw"] - 1) * p["stride"] + ksize
            assert p["tiles_x"] * p["tw"] >= p["wout"] and p["tiles_y"] * p["th"] >= p["hout"]
            assert (p["tiles_x"] - 1) * p["tw"] < p["wout"] and (p["tiles_y"] - 1) * p["th"] < p["hout"]
            if p["T"] > 1:
                assert p["tiles_x"] == p["tiles_y"] == 1
            cchunks = (p["cin"] + 15) // 16
            assert p["cpg"] * p["phases"] >= cchunks and p["cpg"] * (p["phases"] - 1) < cchunks
            gemm_weights += p["taps"] * cchunks * 16 * ((p["cout"] + 15) // 16) * 16
        assert kinds.count(0) >= 10 and kinds.count(1) == 1      # GEMM layers, one scaling operator
        assert 3 not in kinds                                     # nothing is left on the per-operator element kernels
    assert derived * 2 >= gemm_weights // 2       # the packed images exist (prediction weights shared by both programs)
    lib.mzx_net_destroy(h)


def test_gomoku_and_atari_tiles_as_designed(lib):
    """The hot layers, two workgroups per CU (78 KB of LDS each): gomoku 128 -> 128 on 11 x 11 = one sample per
    workgroup in two 64-channel phases; atari 256 -> 256 on 6 x 6 = four samples (nine full row tiles), four phases."""
    h = _create(lib, configs.gomoku())
    p = _plan(lib, h, 1, 1)
    assert (p["taps"], p["cin"], p["cout"], p["T"], p["rows"], p["mtiles"], p["phases"]) == (9, 128, 128, 1, 121, 8, 2)
    assert p["in_layout"] == 0 and p["out_layout"] == 0
    p0 = _plan(lib, h, 1, 0)
    assert p0["in_layout"] == 1 and p0["cin"] == 128      # dynamics input: NCHW hidden state, action plane folded away
    lib.mzx_net_destroy(h)
    h = _create(lib, configs.atari())
    p = _plan(lib, h, 1, 1)
    assert (p["taps"], p["cin"], p["cout"], p["T"], p["rows"], p["mtiles"], p["phases"]) == (9, 256, 256, 4, 144, 9, 4)
    s = _plan(lib, h, 0, 0)                                # stem: 131 -> 128 channels, stride 2, 96 x 96 -> 48 x 48
    assert (s["stride"], s["cin"], s["cout"], s["hout"], s["wout"]) == (2, 131, 128, 48, 48) and s["T"] == 1
    lib.mzx_net_destroy(h)


SHAPE = ("T", "rows", "mtiles", "lds", "ntiles_wg", "nsplit", "NT", "WN", "WM", "MT", "groups", "cpg", "phases", "Cs",
         "ntiles", "cchunks")


@pytest.mark.parametrize("name", ["gomoku", "atari"])
def test_launch_shapes_cover_the_batch_and_fit_the_kernel(lib, name):
    """rb_choose_shape at batches from 1 to 4096: every sample / column tile / channel chunk is covered, the tiling
    fits the kernel's limits, small batches spread over more workgroups, large ones keep the planned tile."""
    h = _create(lib, configs.BY_NAME[name]())
    for recurrent in (0, 1):
        for op in range(lib.mzx_net_num_operators(h, recurrent)):
            p = _plan(lib, h, recurrent, op)
            if p["kind"] != 0:
                continue
            grids = {}
            for batch in (1, 3, 16, 64, 200, 512, 4096):
                out = (ctypes.c_int32 * 16)()
                lib.check(lib.mzx_net_streamed_shape(h, recurrent, op, batch, ctypes.byref(out)))
                s = dict(zip(SHAPE, list(out)))
                spatial = p["tiles_x"] * p["tiles_y"]
                assert 1 <= s["T"] <= p["T"] and s["rows"] == s["T"] * p["th"] * p["tw"] and s["mtiles"] == (s["rows"] + 15) // 16
                assert s["groups"] == -(-batch // s["T"]) * spatial
                assert s["nsplit"] * s["ntiles_wg"] >= s["ntiles"] > (s["nsplit"] - 1) * s["ntiles_wg"]
                assert s["WN"] * s["WM"] <= 8 and s["WN"] * s["NT"] >= s["ntiles_wg"] and s["WM"] * s["MT"] >= s["mtiles"]
                assert 1 <= s["MT"] <= 9 and s["NT"] in (1, 2)
                assert s["cpg"] * s["phases"] >= s["cchunks"] and s["Cs"] == 16 * s["cpg"] + 8 and s["lds"] <= 156 * 1024
                assert s["phases"] <= p["phases"]
                grids[batch] = s["groups"] * s["nsplit"]
                if batch >= 4096 and p["taps"] == 9 and -(-batch // p["T"]) * spatial >= 512:
                    assert s["T"] == p["T"]            # trunk layers at large batches: the planned tile
            assert grids[1] >= 1 and grids[4096] >= grids[64] >= grids[1]
    lib.mzx_net_destroy(h)


@pytest.mark.parametrize("name,B", [("gomoku", 1024), ("connect4", 9216), ("atari", 1024), ("gomoku", 4096), ("connect4", 1024)])
def test_half_shards_keep_the_channel_groups_of_the_whole_shard(lib, name, B):
    """
    search_run_rows (csrc/mzx_row_search.h) runs shards of >= 1024 trees as two halves on two streams -- but only when every
    layer of both halves keeps the channel groups (phases x chunks per group: the one launch-shape property that changes a
    summation order) of the undivided launch.  For the shipped configurations that holds from 512 trees per half on, so
    the split engages for the bench workloads (gomoku 1024, c4-large 9216, atari 1024).
    """
    h = _create(lib, configs.BY_NAME[name]())
    first = ((B // 2 + 15) // 16) * 16
    gemms = 0
    for op in range(lib.mzx_net_num_operators(h, 1)):
        if _plan(lib, h, 1, op)["kind"] != 0:
            continue
        gemms += 1
        shapes = []
        for batch in (B, first, B - first):
            out = (ctypes.c_int32 * 16)()
            lib.check(lib.mzx_net_streamed_shape(h, 1, op, batch, ctypes.byref(out)))
            shapes.append(dict(zip(SHAPE, list(out))))
        assert {(s["phases"], s["cpg"]) for s in shapes} == {(shapes[0]["phases"], shapes[0]["cpg"])}, (name, op, shapes)
    assert gemms >= 10
    lib.mzx_net_destroy(h)


# ---- round 4: towers (rb_tower_kernel), their tails, the head chains -- host-side planning through the C ABI

TOWER = ("first", "count", "C", "H", "W", "T", "MT", "NT", "WM", "WN", "lds", "groups", "n_tail")


def _towers(lib, h, recurrent, batch):
    out, index = [], 0
    buf = (ctypes.c_int32 * 16)()
    while lib.mzx_net_streamed_tower(h, recurrent, index, batch, ctypes.byref(buf)) == 0:
        out.append(dict(zip(TOWER, list(buf))))
        index += 1
    return out


def test_towers_of_the_shipped_networks(lib):
    """
    Which runs of operators become ONE rb_tower_kernel launch: the representation / dynamics trunk (conv + residual
    blocks) and the prediction trunk of every residual configuration; the per-plane scaling and the small 1x1 head
    convolutions ride in the tail; boards too large for a whole-sample tile have none.
    """
    want = {   # name: (recurrent towers as (first, layers, tail operators), initial towers)
        "connect4": ([(0, 7, 2), (11, 6, 2)], [(0, 7, 1), (8, 6, 2)]),
        "gomoku": ([(0, 13, 2), (17, 12, 2)], [(0, 13, 1), (14, 12, 2)]),
        "atari": ([(0, 33, 1), (38, 32, 0)], [(20, 32, 1), (53, 32, 0)]),          # 256-channel heads: GEMM launches
        "tictactoe": ([(0, 3, 1), (7, 2, 0)], [(0, 3, 1), (4, 2, 0)]),            # 16 reduced channels: heads stay GEMMs
    }
    for name, (rec, init) in want.items():
        h = _create(lib, configs.BY_NAME[name]())
        for recurrent, expect in ((1, rec), (0, init)):
            got = [(t["first"], t["count"], t["n_tail"]) for t in _towers(lib, h, recurrent, 1024)]
            assert got == expect, (name, recurrent, got)
        lib.mzx_net_destroy(h)
    # 19 x 19: a sample's positions do not fit a workgroup's row tiles -> no towers, every layer launches on its own
    h = _create(lib, configs.connect4(observation_shape=(3, 19, 19), action_space=list(range(361))))
    assert _towers(lib, h, 1, 64) == []
    lib.mzx_net_destroy(h)
    # network modes 4 / 5: no towers
    h = _create(lib, configs.gomoku())
    lib.check(lib.mzx_net_set_mode(h, 5))
    assert _towers(lib, h, 1, 512) == []
    lib.mzx_net_destroy(h)


def test_tower_shapes_follow_the_calibrated_cost_model(lib):
    """
    Samples per workgroup by batch (profiles/r04_tower_experiments.txt section 4: whole rounds of workgroups decide), the
    LDS fit, and the refusal of shapes that waste more than a fifth of the MFMA rows.
    """
    h = _create(lib, configs.connect4())
    lib.check(lib.mzx_net_set_mode(h, 3))
    for batch, T, MT in ((512, 2, 3), (768, 3, 4), (1024, 4, 6), (3072, 6, 8), (4608, 6, 8)):
        t = _towers(lib, h, 1, batch)[0]
        assert (t["T"], t["MT"], t["NT"]) == (T, MT, 1), (batch, t)
        assert t["groups"] == -(-batch // T) and t["lds"] <= 156 * 1024
        assert t["WM"] * t["WN"] <= 8 and t["WM"] * t["MT"] * 16 >= t["T"] * 42
    lib.mzx_net_destroy(h)
    h = _create(lib, configs.atari())
    # whether a tower runs as one launch is a property of the NETWORK (towers and layer launches sum in different
    # orders, so the choice must not move with the batch): two 6 x 6 samples fill 72 of 80 rows -> towers at every batch;
    # 256 trees take one sample per workgroup (36 of 48 rows) to fill the chip
    small = _towers(lib, h, 1, 256)
    assert [(t["T"], t["MT"], t["NT"], t["groups"]) for t in small] == [(1, 3, 2, 256), (1, 3, 2, 256)]
    for batch in (1, 2, 7, 64, 100, 256, 300, 512, 1024, 4096):
        assert all(t["groups"] > 0 for t in _towers(lib, h, 1, batch)), batch
    big = _towers(lib, h, 1, 512)
    assert [(t["T"], t["MT"], t["NT"]) for t in big] == [(2, 5, 2), (2, 5, 2)] and all(t["groups"] == 256 for t in big)
    lib.mzx_net_destroy(h)
    h = _create(lib, configs.gomoku())
    t = _towers(lib, h, 1, 512)[0]
    assert (t["T"], t["MT"], t["NT"], t["WN"], t["groups"]) == (1, 8, 1, 8, 512)    # one 11 x 11 board per workgroup
    lib.mzx_net_destroy(h)


def test_route_and_tower_use_do_not_depend_on_the_shard_size(lib):
    """
    ADVICE r5: mzx_net_search_route for 64-channel networks on several boards, swept over the shard size.  The ARITHMETIC a
    search runs on (route 1: the LDS-resident engine of rz_search_kernel; routes 2 / 3: the towers, launch by launch or
    inside rt_search_kernel -- bit-identical to each other, tests/test_gpu_tower_search.py) must be the same at every shard
    size: in round 5 a 5 x 5 board took the LDS-resident engine up to 512 trees and the towers from 1024 on, so a 1024-game
    shard pipelined as two groups of 512 played other games than the undivided shard.  Also: the towers of every shipped
    streamed configuration are in use at every batch or at none.
    """
    import streamed_coverage as sc

    boards = {
        "connect4 6x7": dict(),
        "5x5": dict(observation_shape=(3, 5, 5), action_space=list(range(25))),
        "5x7": dict(observation_shape=(3, 5, 7), action_space=list(range(7))),
        "4x4": dict(observation_shape=(3, 4, 4), action_space=list(range(16))),
        "8x8": dict(observation_shape=(3, 8, 8), action_space=list(range(64))),
        "3x3 wide": dict(observation_shape=(3, 3, 3), action_space=list(range(9))),
        "9x9 48ch": dict(observation_shape=(3, 9, 9), action_space=list(range(81)), channels=48),
    }
    sizes = (1, 2, 3, 16, 64, 100, 255, 256, 512, 640, 768, 1024, 1536, 2048, 4096, 9216)
    for name, overrides in boards.items():
        kinds = {}
        for B in sizes:
            route = sc.net_route(lib, "connect4", B, overrides, simulations=8)[0]
            kinds[B] = "towers" if route in (2, 3) else {1: "lds-resident", 0: "generic", 4: "fc"}[route]
        assert len(set(kinds.values())) == 1, (name, kinds)
    for game in ("connect4", "gomoku", "atari"):
        h = _create(lib, configs.BY_NAME[game]())
        if game == "connect4":
            lib.check(lib.mzx_net_set_mode(h, 3))
        for recurrent in (0, 1):
            used = {B: tuple(t["groups"] > 0 for t in _towers(lib, h, recurrent, B)) for B in sizes}
            assert len(set(used.values())) == 1, (game, recurrent, used)
        lib.mzx_net_destroy(h)


def test_head_chains_run_level_by_level_in_grouped_launches(lib):
    """The head MLPs behind a tower's tail: by default ONE rb_gemm_multi_kernel launch per level (tuning "rb_heads" = 2);
    0 = one launch per layer."""
    from mzx import models

    h = _create(lib, configs.connect4())
    lib.check(lib.mzx_net_set_mode(h, 3))
    out = (ctypes.c_int32 * 16)()
    lib.check(lib.mzx_net_streamed_heads(h, 1, 512, ctypes.byref(out)))
    assert out[0] == 6 and out[1] == 3 and out[15] == 2          # reward, value, policy: two Linear layers each
    assert [(out[14] >> (2 * k)) & 3 for k in range(6)] == [0, 1, 0, 1, 0, 1]
    launches = models.streamed_launches(lib, h, 1, 512)
    heads = [l for l in launches if l["op"] in set(out[2:8])]
    assert len(heads) == 6 and all(l["k_loop"] == "ring grouped" and (l["MT"], l["NT"]) == (1, 1) for l in heads)
    lib.tuning_set("rb_heads", 0)
    lib.check(lib.mzx_net_streamed_heads(h, 1, 512, ctypes.byref(out)))
    assert list(out)[:2] == [0, 0] and out[15] == 0
    assert all(l["k_loop"] == "ring" for l in models.streamed_launches(lib, h, 1, 512) if l["taps"] == 1)
    lib.tuning_set("rb_heads", 2)
    lib.check(lib.mzx_net_streamed_heads(h, 0, 512, ctypes.byref(out)))
    assert out[0] == 4 and out[1] == 2                         # initial_inference: value and policy
    # atari: 256-channel head convolutions are GEMM launches, not tails -> no chains; tictactoe: 16 reduced channels, same
    for name in ("atari", "tictactoe"):
        h3 = _create(lib, configs.BY_NAME[name]())
        lib.check(lib.mzx_net_streamed_heads(h3, 1, 512, ctypes.byref(out)))
        assert out[0] == 0, name
        lib.mzx_net_destroy(h3)
    lib.tuning_set("rb_tail", 0)                               # no tails -> nobody writes the chains' private inputs
    lib.check(lib.mzx_net_streamed_heads(h, 1, 512, ctypes.byref(out)))
    assert out[0] == 0
    lib.mzx_net_destroy(h)
    # the workspace carries the private head-input region behind the temporaries
    h2 = _create(lib, configs.connect4())
    assert lib.mzx_net_workspace_floats(h2, 7) == 7 * lib.mzx_net_workspace_floats(h2, 1)
    lib.mzx_net_destroy(h2)
