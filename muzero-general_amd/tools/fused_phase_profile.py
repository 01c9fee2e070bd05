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

PHASES = ["setup/stage", "initial+root", "select", "recurrent net", "priors+expand", "attach+backprop", "finalize"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=4096)
    ap.add_argument("--lds", action="store_true", help="force the LDS-weight engine")
    args = ap.parse_args()
    cfg = configs.cartpole()
    B = args.trees
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 0))
    engine = self_play.BatchedMCTS(cfg, net, B, mode=1 | 8 | (4 if args.lds else 0))
    obs = synthetic.observations(B, cfg.observation_shape, seed=123)
    res = engine.run(list(obs), [list(cfg.action_space)] * B, [0] * B, True,
                     [numpy.random.RandomState(1000 + i) for i in range(B)])
    torch.cuda.synchronize()
    off = engine.arena_offsets(B)
    raw = engine.arena(B)[off["workspace"]: off["workspace"] + B * 16 * 4].view(torch.int32).cpu().numpy().reshape(B, 16)
    S = cfg.num_simulations
    total = raw[:, : len(PHASES)].sum(1).mean()
    print(f"trees {B}  sims {S}  mean leaf depth {res.sum_depth.mean() / S:.2f}  engine {'LdsNet' if args.lds else 'auto'}")
    for k, name in enumerate(PHASES):
        c = raw[:, k].mean()
        per = c / S if k in (2, 3, 4, 5) else c
        print(f"  {name:18s} {c:12.0f} cycles/tree  ({100 * c / total:5.1f}%)" + (f"   {per:8.0f} per simulation" if k in (2, 3, 4, 5) else ""))
    print(f"  total              {total:12.0f} cycles/tree  = {total / 2.4e3:.1f} us at 2.4 GHz")


if __name__ == "__main__":
    main()
