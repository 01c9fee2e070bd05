"""
Probe: does splitting a shard into two halves on two HIP streams fill the launch ramps / tails of the layer-by-layer
network?  recurrent_inference of `game` at batch B on one stream against two networks at B / 2 on two streams.

    python muzero-general_amd/tools/two_stream_probe.py gomoku 1024 [--mode 1|3]
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
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    cfg = configs.BY_NAME[a.game]()
    nets = []
    for _ in range(3):
        net = models.MuZeroNetwork(cfg)
        net.set_weights(synthetic.fill_state_dict(net.state_dict(), 3))
        net.set_mode(a.mode)
        nets.append(net)
    dev = nets[0].backend.device
    rs = numpy.random.RandomState(0)
    B, H = a.batch, a.batch // 2
    hid = torch.tensor(rs.rand(B, *nets[0].hidden_shape).astype(numpy.float32)).to(dev)
    act = torch.tensor(rs.randint(0, len(cfg.action_space), size=B).astype(numpy.int32)).to(dev)
    flops = nets[0].backend.lib.mzx_net_flops(nets[0].handle, 1) * B
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def one():
        nets[0].recurrent_inference(hid, act)

    def two():
        with torch.cuda.stream(s1):
            nets[1].recurrent_inference(hid[:H], act[:H])
        with torch.cuda.stream(s2):
            nets[2].recurrent_inference(hid[H:], act[H:])

    for name, fn in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        for _ in range(a.iters):
            fn()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / a.iters
        print(f"{a.game} batch {B} {name}: {ms:.3f} ms = {flops / (ms * 1e-3) / PEAK:.3f} of the FP32 MFMA peak")


if __name__ == "__main__":
    main()
