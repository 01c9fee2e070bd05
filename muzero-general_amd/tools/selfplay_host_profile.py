"""
cProfile of the host envelope of batched self-play on the GPU (C2, 4096 games through the batched game protocol):
where a move's wall time goes around the search kernel.

    python muzero-general_amd/tools/selfplay_host_profile.py [--moves 32] [--plugin]
"""
import argparse
import cProfile
import copy
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mzx import configs, models, self_play, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--moves", type=int, default=32)
    ap.add_argument("--trees", type=int, default=4096)
    ap.add_argument("--plugin", action="store_true", help="reference plugin surface (B Game objects) instead of the batched protocol")
    args = ap.parse_args()
    cfg = copy.copy(configs.cartpole())
    cfg.max_moves = args.moves
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 0))
    make = synthetic.make_synthetic_game if args.plugin else synthetic.make_synthetic_batched_game
    Game = make(cfg.observation_shape, len(cfg.action_space), len(cfg.players))
    sp = self_play.SelfPlay({"weights": net.get_weights()}, Game, cfg, 0, num_games=args.trees)
    run = lambda: sp.play_rounds(1.0, None, min_games=1 << 60, max_rounds=args.moves)    # (what bench.py's self-play legs run)
    run()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.trees} games x {args.moves} rounds: {dt / args.moves * 1e3:.3f} ms per round = {args.trees * args.moves / dt / 1e6:.3f} M steps/s")
    pr = cProfile.Profile()
    pr.enable()
    run()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
