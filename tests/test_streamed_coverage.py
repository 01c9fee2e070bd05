"""
CPU guard of the streamed engine's parity coverage (no GPU: launch shapes come from the C ABI's host-side planner).

bench.py's roofline fractions for games/gomoku.py, games/atari.py and connect4 at a large shard are measured on
particular instantiations of rb_gemm_kernel<MT, NT> (channel phases, K loop, wave grid, samples per workgroup).  Every
one of those launches must also be made by a -m gpu parity test (tests/test_gpu_streamed_at_size.py, table in
tests/streamed_coverage.py) -- a performance number without parity evidence for the very code path that was timed is
not a result.  This test fails when a change to bench.py's defaults or to the launch planner moves a bench workload
outside the GPU-tested set.
"""
import importlib.util
import os

import pytest

import streamed_coverage as sc
from conftest import ROOT
from mzx import _lib, models


@pytest.fixture(scope="module")
def lib():
    return _lib.Library(_lib.LIB_PATH)


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_streamed_workloads_are_the_expected_ones(lib, bench):
    # (if a workload joins or leaves the default line, the at-size tables have to be looked at again)
    assert sorted(sc.bench_streamed_workloads(lib, bench)) == ["atari", "gomoku"]
    assert sorted(sc.bench_rt_workloads(lib, bench)) == ["c4", "c4-large"]
    assert bench.WORKLOAD_NET_MODE == {}


def test_bench_whole_search_workloads_run_tested_shapes(lib, bench):
    """The workloads the library runs as ONE launch of rt_search_kernel: the (game, shard) is one a -m gpu test runs on the
    planner's own shape, and the root -- initial_inference at the shard size on the streamed engine -- launches only
    GPU-tested shapes."""
    tested = sc.gpu_tested_launches(lib)
    tested_full = {models.launch_key(l) for l in tested}
    for name, (game, overrides, trees, route) in sc.bench_rt_workloads(lib, bench).items():
        assert (game, trees) in sc.RT_AT_SIZE, f"bench workload {name}: no at-size test of rt_search_kernel at {trees} trees"
        assert route[2] <= 8 and route[6] in (256, 512) and route[5] <= 160 * 1024, (name, route)
        initial = sc.inference_launches(lib, game, trees, 0, overrides, mode=3)
        missing = [l for l in initial if models.launch_key(l) not in tested_full]
        assert not missing, f"bench workload {name}: root launch shapes outside the GPU-tested set: {missing[:3]}"


def test_every_bench_launch_is_gpu_parity_tested(lib, bench):
    tested = sc.gpu_tested_launches(lib)
    tested_full = {models.launch_key(l) for l in tested}
    tested_inst = {models.instantiation_key(l) for l in tested}
    for name, (game, overrides, trees, mode) in sc.bench_streamed_workloads(lib, bench).items():
        launches, parts = sc.search_launches(lib, game, trees, overrides, mode=mode)
        assert launches
        missing = sorted({models.instantiation_key(l) for l in launches} - tested_inst)
        assert not missing, f"bench workload {name} ({trees} trees, half-shards {parts}) launches instantiations no -m gpu parity test runs: {missing}"
        missing = [l for l in launches if models.launch_key(l) not in tested_full]
        assert not missing, f"bench workload {name}: launch shapes outside the GPU-tested set: {missing[:3]}"


def test_bench_workloads_run_at_exactly_tested_sizes(lib, bench):
    """Stronger than shape equality: the (configuration, program, batch) triples themselves are in the at-size tables."""
    tested = set()
    towers_on = lambda mode: mode not in (4, 5)
    for game, mode, batch in sc.AT_SIZE.values():
        tested |= {(game, towers_on(mode), "initial", batch), (game, towers_on(mode), "recurrent", batch)}
    for game, mode, trees, _, _ in sc.AT_SIZE_SEARCHES.values():
        tested |= {(game, towers_on(mode), l["program"], l["batch"])
                   for l in sc.search_launches(lib, game, trees, both_ways=True, mode=mode)[0]}
    for name, (game, overrides, trees, mode) in sc.bench_streamed_workloads(lib, bench).items():
        for l in sc.search_launches(lib, game, trees, overrides, mode=mode)[0]:
            assert (game, towers_on(mode), l["program"], l["batch"]) in tested, (name, l["program"], l["batch"])


def test_split_matches_the_row_search(lib):
    """mzx_net_streamed_split: 16-tree aligned halves from 1024 trees, undivided below."""
    from mzx import configs
    import ctypes
    h = sc._create(lib, configs.gomoku())
    assert models.streamed_split(lib, h, 1024) == (512, 512)
    assert models.streamed_split(lib, h, 1000) == (1000, 0)
    assert models.streamed_split(lib, h, 1030) == (528, 502)
    assert models.streamed_split(lib, h, 1) == (1, 0)
    n = lib.mzx_net_num_operators(h, 1)
    assert lib.mzx_net_operator_out_floats(h, 1, 0) == 128 * 11 * 11 and lib.mzx_net_operator_out_floats(h, 1, n) == 0
    lib.mzx_net_destroy(h)
    h = sc._create(lib, configs.cartpole())          # fully connected: no streamed engine, never split
    assert models.streamed_split(lib, h, 4096) == (4096, 0)
    lib.mzx_net_destroy(h)
