"""
bench.py plumbing on the CPU (no GPU): `python bench.py --gpus 2` must launch its own two ranks, shard the
trees, broadcast the flat weight buffer (gloo here, RCCL on the GPU box) and print ONE JSON line from rank 0
(the N > 1 bench was not launchable as `python bench.py --gpus N`).  --dry-run binds the serial
test double of the ABI; its numbers are not measurements.  Also: the cpu_baseline workers (the unmodified
reference from oracle/_ref bytecode, and the oracle port) count simulations the same way.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(argv, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    line = _run(["--gpus", "2", "--dry-run", "--trees", "8", "--steps", "2", "--warmup", "1", "--selfplay-moves", "0",
                 "--also", "c3", "--also-steps", "1"])
    # every workload of a multi-GPU run reports every rank's own time per step (the first real 8-GPU run yields the scaling
    # of the CartPole AND the residual workloads in one go)
    assert len(line["per_rank"]["ms_per_step"]) == 2 and all(t > 0 for t in line["per_rank"]["ms_per_step"])
    assert [w["w"] for w in line["workloads"]] == ["c3"] and len(line["workloads"][0]["per_rank_ms"]) == 2
    assert line["dry_run"] is True and "NOT a measurement" in line["data"]
    assert line["n_gpus"] == 2 and line["collective_world_size"] == 2 and line["scaling"] == "weak"
    assert line["steps"] == 2 and line["warmup"] == 1
    assert len(line["per_rank"]["sims_per_sec"]) == 2
    # whole-job aggregate: both shards' simulations over the slower rank's time
    assert line["value"] <= sum(line["per_rank"]["sims_per_sec"]) * 1.001
    assert line["value"] >= 2 * line["per_rank"]["min"] * 0.999
    assert "weak_scaling_efficiency" in line["single_gpu_reference"]
    assert line["weight_broadcast_ms"] > 0
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "config", "roofline"):
        assert key in line


def test_bench_single_rank_dry_run_with_second_workload():
    line = _run(["--dry-run", "--trees", "4", "--steps", "1", "--warmup", "0", "--selfplay-moves", "2",
                 "--also", "c3", "--also-steps", "1"])
    assert line["n_gpus"] == 1 and "per_rank" not in line
    assert [w["w"] for w in line["workloads"]] == ["c3"]      # compact entries, short keys (bench.py docstring)
    assert set(line["workloads"][0]) == {"w", "v", "ms", "B", "S", "L", "k", "min", "max", "n", "roofline"}
    assert set(line["workloads"][0]["roofline"]) == {"bound", "achieved", "frac", "traffic", "launch_ms"}
    assert line["workloads"][0]["roofline"]["bound"] == "mfma"
    assert "reference constructor" in line["config"]["weights"] or "synthetic" in line["config"]["weights"]
    assert line["selfplay_end_to_end"]["steps_per_sec"] > 0
    assert line["selfplay_end_to_end_batched_game"]["steps_per_sec"] > 0
    loop = line["selfplay_actor_loop"]         # continuous_self_play with the replay hand-off: 3 shards of 4 games, all with priorities
    assert "error" not in loop and loop["games_saved"] >= 12 and loop["with_priorities"] == loop["games_saved"]
    assert loop["steps_saved"] == 2 * loop["games_saved"] and loop["steps_per_sec"] > 0


def test_cpu_baseline_workers_reference_and_port():
    sys.path.insert(0, ROOT)
    import torch

    import bench
    from oracle import build_ref

    threads = torch.get_num_threads()     # the workers pin torch to one thread (one process per core in the bench):
    try:                                  # restore it, ATen's convolutions are not bit-reproducible across thread counts
        sims, searches, wall = bench._cpu_worker_port(("c2", 0, 0.5, "reference"))
        assert searches >= 1 and sims == 50 * searches and wall > 0
        if not build_ref.available():
            pytest.skip("oracle/_ref not built (no /root/reference at build time)")
        sims, searches, wall = bench._cpu_worker_reference(("c2", 0, 0.5, "reference"))
        assert searches >= 1 and sims == 50 * searches and wall > 0
    finally:
        torch.set_num_threads(threads)


def test_bench_large_shard_workload_labels_what_ran():
    """A `--trees` override shows in the line's label; the dry run (serial test double) names its path honestly."""
    line = _run(["--dry-run", "--workload", "c4-large", "--trees", "4", "--steps", "1", "--warmup", "0",
                 "--selfplay-moves", "0", "--also", "none"])
    cfg = line["config"]
    assert cfg["trees_per_gpu"] == 4 and "4 trees x 200 sims" in cfg["workload"] and "9216" not in cfg["workload"]
    assert cfg["search_kernel"] == "one kernel per operator"      # (the test double has no tuned kernels)
    assert line["roofline"]["bound"] == "mfma" and line["roofline"]["traffic"] is None     # PMC entry is for the default shard
