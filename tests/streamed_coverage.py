"""
Which launches of the streamed MFMA engine (csrc/mzx_batched.hip: rb_gemm_kernel<MT, NT>, channel phases, K loop) the
AT-SIZE GPU parity tests make, and which ones bench.py's workloads make -- both derived host-side through the C ABI
(mzx_net_streamed_shape / mzx_net_streamed_split; no GPU), from the same tables the tests and the bench run on:

  AT_SIZE           the (configuration, network mode, batch) cases tests/test_gpu_streamed_at_size.py runs operator by
                    operator and against the oracle network -- both programs at `batch`;
  AT_SIZE_SEARCHES  the whole searches the same file runs against the CPU oracle (initial_inference at the shard size,
                    recurrent_inference at the half-shard sizes AND at the undivided shard: the test runs both ways);
  bench_launches()  what every streamed workload of bench.py launches at its default shard.

tests/test_streamed_coverage.py (CPU) fails when a bench launch is not among the GPU-tested ones, so the roofline
fractions the bench reports always have device parity evidence behind the very instantiation that was timed.
"""
import ctypes

from mzx import configs, models

# name -> (configs factory, network mode for set_mode (None: the default routing), batch)
AT_SIZE = {
    "gomoku-512": ("gomoku", None, 512),
    "gomoku-1024": ("gomoku", None, 1024),
    "connect4-512": ("connect4", 3, 512),
    "connect4-1024": ("connect4", 3, 1024),
    "connect4-4608": ("connect4", 3, 4608),
    "connect4-9216": ("connect4", 3, 9216),
    "atari-256": ("atari", None, 256),
    "atari-512": ("atari", None, 512),
    "atari-1024": ("atari", None, 1024),
    # the same trunks LAYER BY LAYER (network modes 4 / 5: no tower launches) -- the path of boards too large for a tower
    # and the A/B of rb_tower_kernel; keeps rb_gemm_kernel<8,1> with two channel phases covered at size
    "connect4-4608-layers": ("connect4", 4, 4608),
    "gomoku-512-layers": ("gomoku", 5, 512),
}

# name -> (configs factory, network mode, trees, simulations (None: as shipped), trees compared with the oracle)
AT_SIZE_SEARCHES = {
    "gomoku-1024": ("gomoku", None, 1024, None, 16),
    "connect4-9216": ("connect4", 3, 9216, None, 64),
    # BASELINE config C4 at its 1024-tree shard: the library's own routing (no network mode set) sends it to this engine
    "connect4-1024": ("connect4", None, 1024, None, 64),
    "atari-1024": ("atari", None, 1024, None, 8),
    "atari-256": ("atari", None, 256, None, 4),
}


def _create(lib, cfg):
    c = models.net_config_from(cfg)
    h = ctypes.c_void_p()
    lib.check(lib.mzx_net_create(ctypes.byref(c), ctypes.byref(h)))
    return h


def inference_launches(lib, game, batch, recurrent, overrides=None, mode=None):
    h = _create(lib, configs.BY_NAME[game](**(overrides or {})))
    try:
        if mode is not None:
            lib.check(lib.mzx_net_set_mode(h, int(mode)))
        return models.streamed_launches(lib, h, recurrent, batch)
    finally:
        lib.mzx_net_destroy(h)


def search_launches(lib, game, trees, overrides=None, both_ways=False, mode=None):
    """
    The GEMM launches of one search over `trees` roots on the row-per-tree path: initial_inference at the shard size,
    recurrent_inference at the sizes of the two half-shards (or the shard, when it runs undivided); `both_ways` adds the
    undivided shard (the parity test runs the search split AND unsplit).  Returns (launch dicts, (first, second)).
    """
    h = _create(lib, configs.BY_NAME[game](**(overrides or {})))
    try:
        if mode is not None:
            lib.check(lib.mzx_net_set_mode(h, int(mode)))
        parts = models.streamed_split(lib, h, trees)
        out = [dict(l, program="initial", batch=trees) for l in models.streamed_launches(lib, h, 0, trees)]
        sizes = {p for p in parts if p > 0}
        if both_ways:
            sizes.add(trees)
        for b in sorted(sizes):
            out += [dict(l, program="recurrent", batch=b) for l in models.streamed_launches(lib, h, 1, b)]
        return out, parts
    finally:
        lib.mzx_net_destroy(h)


def gpu_tested_launches(lib):
    """Every launch the at-size GPU parity tests make."""
    out = []
    for game, mode, batch in AT_SIZE.values():
        for recurrent in (0, 1):
            out += inference_launches(lib, game, batch, recurrent, mode=mode)
    for game, mode, trees, _, _ in AT_SIZE_SEARCHES.values():
        out += search_launches(lib, game, trees, both_ways=True, mode=mode)[0]
    return out


def net_route(lib, game, trees, overrides=None, simulations=None):
    """mzx_net_search_route for a search of `trees` roots of configuration `game` (host-side planner, no GPU): the list
    [route, trees per workgroup, row tiles per wave, workgroups, workgroups per CU, LDS bytes, threads | first half, second half]."""
    cfg = configs.BY_NAME[game](**(overrides or {}))
    h = _create(lib, cfg)
    try:
        out = (ctypes.c_int32 * 8)()
        lib.check(lib.mzx_net_search_route(h, int(trees), int(simulations or cfg.num_simulations), ctypes.byref(out)))
        return list(out)
    finally:
        lib.mzx_net_destroy(h)


def bench_streamed_workloads(lib, bench):
    """bench.py's default workloads whose SIMULATIONS run launch by launch on the streamed engine (route 2):
    {name: (game, overrides, trees, network mode)}."""
    names = [bench.DEFAULT_WORKLOAD] + [w for w in bench.DEFAULT_ALSO.split(",") if w]
    out = {}
    for w in names:
        game, overrides, trees, _ = bench.WORKLOADS[w]
        if configs.BY_NAME[game](**overrides).network != "resnet" or bench.WORKLOAD_TUNING.get(w, {}).get("wide_towers") == 0:
            continue
        forced = bench.WORKLOAD_NET_MODE.get(w) == "streamed"
        if net_route(lib, game, trees, overrides)[0] == 2 or bench.WORKLOAD_TUNING.get(w, {}).get("rt_search") == 0:
            out[w] = (game, overrides, trees, 3 if forced else None)
    return out


# rt_search_kernel (csrc/mzx_tower_search.inc: every simulation in one launch) AT SIZE: (game, trees) -> the -m gpu test that
# runs the planner's shape for that shard against the CPU oracle / the per-simulation launches
RT_AT_SIZE = {
    ("connect4", 512): "test_gpu_tower_search.py::test_tower_search_other_shards_against_oracle[512-1-32]",
    ("connect4", 1024): "test_gpu_parity.py::test_full_size_residual_configs[connect4] + test_gpu_tower_search.py::test_tower_search_at_size_same_trees_as_launches",
    ("connect4", 1536): "test_gpu_tower_search.py::test_tower_search_other_shards_against_oracle[1536-6-48]",
    ("connect4", 9216): "test_gpu_tower_search.py::test_tower_search_other_shards_against_oracle[9216-6-48]",
}


def bench_rt_workloads(lib, bench):
    """bench.py's default workloads the library runs on rt_search_kernel (route 3): {name: (game, overrides, trees, route list)}."""
    names = [bench.DEFAULT_WORKLOAD] + [w for w in bench.DEFAULT_ALSO.split(",") if w]
    out = {}
    for w in names:
        game, overrides, trees, _ = bench.WORKLOADS[w]
        if configs.BY_NAME[game](**overrides).network != "resnet" or bench.WORKLOAD_TUNING.get(w):
            continue
        r = net_route(lib, game, trees, overrides)
        if r[0] == 3:
            out[w] = (game, overrides, trees, r)
    return out


summarize = models.summarize_launches
