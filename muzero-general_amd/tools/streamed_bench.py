"""
Times the streamed MFMA engine (csrc/mzx_batched.hip) on the reference's large residual configurations:
recurrent_inference / initial_inference at a given batch, HIP events on the library's stream, FLOPs from
mzx_net_flops (2 x MAC of every convolution / Linear layer) against the 157.3 TFLOP/s FP32 matrix peak.

    python muzero-general_amd/tools/streamed_bench.py gomoku 512 [--mode 0|1|3] [--iters 20]
"""
import argparse
import os
import sys

import numpy
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mzx import configs, models, synthetic  # noqa: E402

PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("game")
    ap.add_argument("batch", type=int)
    ap.add_argument("--mode", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--initial", action="store_true")
    a = ap.parse_args()
    cfg = configs.BY_NAME[a.game]()
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 3))
    net.set_mode(a.mode)
    lib = net.backend.lib
    B = a.batch
    rs = numpy.random.RandomState(0)
    hid = torch.tensor(rs.rand(B, *net.hidden_shape).astype(numpy.float32)).to(net.backend.device)
    act = torch.tensor(rs.randint(0, len(cfg.action_space), size=B).astype(numpy.int32)).to(net.backend.device)
    legs = [("recurrent_inference", lambda: net.recurrent_inference(hid, act), lib.mzx_net_flops(net.handle, 1))]
    if a.initial:
        obs = torch.tensor(rs.rand(B, *net.input_shape).astype(numpy.float32)).to(net.backend.device)
        legs.append(("initial_inference", lambda: net.initial_inference(obs), lib.mzx_net_flops(net.handle, 0)))
    stream = torch.cuda.current_stream()
    for name, fn, flops in legs:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        for _ in range(a.iters):
            fn()
        t1.record(stream)
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / a.iters
        tf = flops * B / (ms * 1e-3) / 1e12
        print(f"{a.game} {name} batch {B} mode {a.mode}: {ms:.3f} ms, {flops * B / 1e9:.1f} GFLOP, "
              f"{tf:.1f} TFLOP/s = {tf * 1e12 / PEAK:.3f} of the FP32 MFMA peak "
              f"(streamed {net.streamed_supported()}, fused {net.fused_supported()})")


if __name__ == "__main__":
    main()
