"""
Cycle breakdown of the fused whole-search kernel (mode flag 8): mean shader cycles per
tree spent in each phase, read back from the arena's workspace region.

    python muzero-general_amd/tools/fused_phase_profile.py [--trees 4096] [--lds]
"""
import argparse
import os
import sys

import numpy
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mzx import configs, models, self_play, synthetic  # noqa: E402

PHASES = ["setup/stage", "initial+root", "select", "recurrent net", "priors+expand", "attach+backprop", "finalize",
          "path prefetch (v2)", "value chain (v2)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=4096)
    ap.add_argument("--lds", action="store_true", help="force the LDS-weight engine")
    ap.add_argument("--v1", action="store_true", help="first-generation fully connected kernel (mode flag 16)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "lunarlander"],
                    help="lunarlander: games/lunarlander.py shape (encoding 10, 64-wide hidden layers, 4 actions) on the LDS-weight engine")
    args = ap.parse_args()
    if args.workload not in ("c2", "lunarlander"):
        return residual(args)
    cfg = configs.cartpole() if args.workload == "c2" else configs.lunarlander()
    B = args.trees
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 0))
    engine = self_play.BatchedMCTS(cfg, net, B, mode=1 | 8 | (4 if args.lds else 0) | (16 if args.v1 else 0))
    obs = synthetic.observations(B, cfg.observation_shape, seed=123)
    res = engine.run(list(obs), [list(cfg.action_space)] * B, [0] * B, True,
                     [numpy.random.RandomState(1000 + i) for i in range(B)])
    torch.cuda.synchronize()
    off = engine.arena_offsets(B)
    raw = engine.arena(B)[off["workspace"]: off["workspace"] + B * 16 * 4].view(torch.int32).cpu().numpy().reshape(B, 16)
    S = cfg.num_simulations
    total = raw[:, : len(PHASES)].sum(1).mean()
    print(f"trees {B}  sims {S}  mean leaf depth {res.sum_depth.mean() / S:.2f}  engine {'LdsNet' if args.lds else 'auto'}{' (v1 kernel)' if args.v1 else ''}")
    for k, name in enumerate(PHASES):
        c = raw[:, k].mean()
        per = c / S if k in (2, 3, 4, 5, 7, 8) else c
        print(f"  {name:18s} {c:12.0f} cycles/tree  ({100 * c / total:5.1f}%)" + (f"   {per:8.0f} per simulation" if k in (2, 3, 4, 5, 7, 8) else ""))
    print(f"  total              {total:12.0f} cycles/tree  = {total / 2.4e3:.1f} us at 2.4 GHz")


def residual(args):
    """Phase cycles of the residual whole-search kernel (thread 0 of every workgroup), summed over the simulations."""
    name, B = {"c3": ("tictactoe", 1024), "c4": ("connect4", 1024), "c5": ("breakout", 64)}[args.workload]
    cfg = configs.BY_NAME[name](**({"num_simulations": 50} if name == "breakout" else {}))
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 0))
    engine = self_play.BatchedMCTS(cfg, net, B, mode=1 | 8)
    obs = synthetic.observations(B, net.input_shape, seed=123)
    for _ in range(2):
        res = engine.run(list(obs), [list(cfg.action_space)] * B, [0] * B, True,
                         [numpy.random.RandomState(1000 + i) for i in range(B)])
    torch.cuda.synchronize()
    off = engine.arena_offsets(B)
    raw = engine.arena(B)[off["workspace"]: off["workspace"] + 256 * 8 * 4].view(torch.int32).cpu().numpy().reshape(-1, 8)
    raw = raw[((raw[:, [0, 2, 3, 4]] > 0) & (raw < 2 ** 29).all(1, keepdims=True)).all(1)]   # workgroups that wrote counters (the region is shared with the stem's workspace)
    if (raw[:, 6] > 0).any():   # wave-per-tree kernel (mzx_resnet_wave.h): the network by operator class, no barriers
        S = cfg.num_simulations
        mean = raw.mean(0)
        names8 = ["select + fetch of the path", "9-chunk GEMMs, weights in LDS", "gather parent state -> LDS", "descriptor fetches + fence",
                  "decode + expand + backpropagate", "9-chunk GEMMs, weights from L2", "short GEMMs (1 chunk)", "min-max scaling + state store"]
        print(f"{args.workload}: {B} trees x {S} sims, {len(raw)} workgroups sampled (wave 0), mean leaf depth {res.sum_depth.mean() / S:.2f}, wave-per-tree kernel")
        for k, nm in enumerate(names8):
            print(f"  {nm:38s} {mean[k] / S:10.0f} cycles per simulation  ({100 * mean[k] / mean.sum():5.1f}%)")
        print(f"  total                                  {mean.sum() / S:10.0f} cycles per simulation = {mean.sum() / S / 2.4e3:.2f} us at 2.4 GHz")
        return
    S = cfg.num_simulations
    names = ["select (lane-parallel)              ", "barrier after select", "gather parent states -> LDS", "network layers",
             "decode + expand + backpropagate", "barrier"]
    mean = raw[:, :6].mean(0)
    print(f"{args.workload}: {B} trees x {S} sims, {len(raw)} workgroups sampled, mean leaf depth {res.sum_depth.mean() / S:.2f}")
    for k, nm in enumerate(names):
        print(f"  {nm:38s} {mean[k] / S:10.0f} cycles per simulation  ({100 * mean[k] / mean.sum():5.1f}%)")
    print(f"  total                                  {mean.sum() / S:10.0f} cycles per simulation = {mean.sum() / S / 2.4e3:.2f} us at 2.4 GHz")
    ticks = raw[:, 7].astype(numpy.float64)
    if (ticks > 0).all():   # s_memrealtime (100 MHz) beside s_memtime: the clock the kernel actually ran at
        mhz = raw[:, :6].sum(1) / ticks * 100.0
        print(f"  effective shader clock over the launch: {mhz.mean():.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f}); "
              f"launch {ticks.mean() / 100.0:.0f} us per workgroup")


if __name__ == "__main__":
    main()
