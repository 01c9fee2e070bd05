"""
Whole-search kernel vs the per-simulation launches of the generic path on a gomoku-shaped configuration
(games/gomoku.py:22-23: 11 x 11 board, 121 actions; network sized for the fused engine).

    python muzero-general_amd/tools/wide_action_bench.py [--trees 256] [--sims 100]
"""
import argparse
import os
import sys
import time

import numpy
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mzx import configs, models, self_play, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=256)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--channels", type=int, default=16)
    ap.add_argument("--blocks", type=int, default=2)
    args = ap.parse_args()
    cfg = configs.connect4(observation_shape=(3, 11, 11), action_space=list(range(121)), channels=args.channels,
                           blocks=args.blocks, reduced_channels_reward=2, reduced_channels_value=2, reduced_channels_policy=4,
                           resnet_fc_reward_layers=[32], resnet_fc_value_layers=[32], resnet_fc_policy_layers=[32],
                           num_simulations=args.sims, root_dirichlet_alpha=0.3)
    B, S = args.trees, args.sims
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 0))
    obs = synthetic.observations(B, net.input_shape, seed=1)
    legal = [list(cfg.action_space)] * B
    out = {}
    for name, mode in (("generic (3 launches per simulation)", 0), ("whole-search kernel", 1)):
        engine = self_play.BatchedMCTS(cfg, net, B, mode=mode)
        kind = engine.backend.lib.mzx_search_fused_supported(engine.handle(B))
        run = lambda: engine.run(list(obs), legal, [0] * B, True, [numpy.random.RandomState(7 + i) for i in range(B)])
        res = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            res = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name] = (dt, res)
        print(f"{name:38s} fused_supported={kind}  {dt * 1e3:9.2f} ms per move  {B * S / dt / 1e6:8.3f} M sims/s")
    (t0, r0), (t1, r1) = out.values()
    same = (r0.visit_counts == r1.visit_counts).all()
    print(f"speed-up {t0 / t1:.2f}x   visit counts identical: {bool(same)}  (trees {B}, sims {S}, 121 actions, "
          f"{args.channels} channels x {args.blocks} blocks; wall incl. host-side noise / tape generation)")


if __name__ == "__main__":
    main()
