"""
Throughput of the device observation stacker (csrc/mzx_obs.h, `mzx_obs_stack`) at the games/atari.py
geometry: 3x96x96 frames, 32 stacked observations -> 131 planes per sample.

    python muzero-general_amd/tools/obs_stack_bench.py [--games 64] [--iters 50]

Prints one JSON line: algorithmic bytes per launch (frame planes read once + every output plane written),
mean launch time from HIP events on the launch stream, GB/s and the fraction of the 8 TB/s HBM peak, next
to (a) a plain device-to-device copy of the same number of bytes and (b) the host path it replaces
(numpy concatenation of the stacked batch + upload).
"""
import argparse
import json
import os
import sys
import time

import numpy
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mzx import _lib, configs, observations, self_play  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=64)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--stacked", type=int, default=32)
    args = ap.parse_args()
    be = _lib.default_backend()
    cfg = configs.HotPathConfig(observation_shape=(3, 96, 96), stacked_observations=args.stacked,
                                action_space=list(range(4)))
    G, k = args.games, args.stacked
    C, H, W = cfg.observation_shape
    store = observations.FrameStore(cfg, G, be)
    rs = numpy.random.RandomState(0)
    hist, acts = [], []
    for t in range(k + 3):   # past the start of the game: every stacked slot reads a real frame
        frame = rs.rand(G, C, H, W).astype(numpy.float32)
        a = rs.randint(0, 4, size=G)
        store.push(frame, None if t == 0 else a)
        hist.append(frame); acts.append(numpy.zeros(G, numpy.int64) if t == 0 else a)
    out = store.stacked()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.iters):
        store.stacked()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / args.iters
    planes_out = C * (k + 1) + k
    planes_in = C * (k + 1)
    bytes_alg = G * (planes_in + planes_out) * H * W * 4
    # plain copy moving the same bytes (read + write)
    src = torch.empty(bytes_alg // 8, dtype=torch.float32, device=be.device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(args.iters):
        dst.copy_(src)
    ev[1].record()
    torch.cuda.synchronize()
    ms_copy = ev[0].elapsed_time(ev[1]) / args.iters
    # the host path: concatenate on the CPU, upload the stacked batch
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        host = self_play.SelfPlay._stacked_batch(hist, acts, k, 4)
        dev = torch.as_tensor(host).to(torch.float32).to(be.device)
    torch.cuda.synchronize()
    ms_host = (time.perf_counter() - t0) * 1e3 / reps
    assert torch.equal(dev, out)
    print(json.dumps({
        "kernel": "mzx_obs_stack", "games": G, "stacked_observations": k, "planes_per_sample": planes_out,
        "algorithmic_bytes_per_launch": bytes_alg, "ms_per_launch": round(ms, 4),
        "achieved_GBps": round(bytes_alg / ms / 1e6, 1), "peak_GBps": 8000, "frac": round(bytes_alg / ms / 1e6 / 8000, 4),
        "d2d_copy_same_bytes_ms": round(ms_copy, 4), "d2d_copy_GBps": round(bytes_alg / ms_copy / 1e6, 1),
        "host_concat_plus_upload_ms": round(ms_host, 2), "speedup_vs_host_path": round(ms_host / ms, 1),
    }))


if __name__ == "__main__":
    main()
