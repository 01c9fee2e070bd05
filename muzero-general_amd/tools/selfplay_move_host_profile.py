"""
The host envelope of a self-play move of the batched game protocol WITHOUT the search (no GPU needed): an instrumented
CPU build of the library (g++ -DMZX_HOSTCHECK -DMZX_EXPERIMENT, built here into /tmp) whose mzx_selfplay_search skips
mzx_search_run when MZX_MOVE_NO_SEARCH=1 -- root draws, staging, select, the game's step and the round's bookkeeping
remain.  `--unfused` times the separate calls of rounds 1-4 (root_draws, _launch, advance, numpy select) the same way.

    python muzero-general_amd/tools/selfplay_move_host_profile.py [--games 4096] [--rounds 32] [--unfused] [--profile]
"""
import argparse
import copy
import cProfile
import os
import pstats
import subprocess
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "muzero-general_amd"))
from mzx import _lib, configs, models, self_play, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=32)
    ap.add_argument("--unfused", action="store_true")
    ap.add_argument("--groups", type=int, default=0, help="1 / 2 slot groups (0: the engine's default)")
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    lib_path = "/tmp/libmzx_hostexp.so"
    src = os.path.join(ROOT, "muzero-general_amd", "csrc", "mzx_lib.cpp")
    if not os.path.isfile(lib_path) or os.path.getmtime(lib_path) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DMZX_HOSTCHECK",
                        "-DMZX_EXPERIMENT", "-x", "c++", src, "-o", lib_path], check=True)
    os.environ["MZX_MOVE_NO_SEARCH"] = "1"
    be = _lib.Backend(_lib.Library(lib_path), "cpu")
    B, A = args.games, 2
    cfg = copy.copy(configs.cartpole())
    cfg.max_moves = args.rounds
    cfg.self_play_pipeline = {0: None, 1: False, 2: True}[args.groups]
    net = models.MuZeroNetwork(cfg, _backend=be)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 0))
    Game = synthetic.make_synthetic_batched_game(cfg.observation_shape, A, len(cfg.players))
    sp = self_play.SelfPlay({"weights": net.get_weights()}, Game, cfg, 0, num_games=B, _backend=be)
    rs = numpy.random.RandomState(0)
    fake_vis = rs.randint(1, 30, size=(B, A)).astype(numpy.int32)
    fake_vis[:, 0] += (cfg.num_simulations - fake_vis.sum(1)).astype(numpy.int32)
    if args.unfused:
        sp.engine.fused_move = False

        def fake_launch(self, B_, obs, legal, to_play, noise, tape, tape_words, override):     # the staging writes of _launch
            vin = self._staging(B_, tape_words, int(obs.size // B_), noise is not None)["vin"]
            vin["obs"][:] = obs.reshape(-1)
            if noise is not None:
                vin["noise"][:] = noise.reshape(-1)
            vin["legal"][:] = legal.reshape(-1)
            vin["to_play"][:] = to_play
            vin["tape"][:] = tape.reshape(-1).view(numpy.uint32)
            return fake_vis.copy(), rs.rand(B_), rs.rand(B_), numpy.zeros((B_, 4), numpy.int32)
        self_play.BatchedMCTS._launch = fake_launch
    else:
        real = self_play.BatchedMCTS._move_search

        def with_fake_counts(self, B_, *a):      # (the skipped search leaves the output block as it was allocated)
            outputs = real(self, B_, *a)

            def clean():
                out = outputs()
                return (fake_vis[:B_].copy(), rs.rand(B_), rs.rand(B_), numpy.zeros_like(out[3])) + out[4:]
            return clean
        self_play.BatchedMCTS._move_search = with_fake_counts
    run = lambda: sp.play_rounds(1.0, None, min_games=1 << 60, max_rounds=args.rounds)
    run()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        run()
        best = min(best, time.perf_counter() - t0)
    print(f"{'separate calls' if args.unfused else 'fused move'}: {B} games x {args.rounds} rounds, host envelope without the search: "
          f"{best / args.rounds * 1e3:.3f} ms per round = {B * args.rounds / best / 1e6:.2f} M steps/s (best of 5, "
          f"{os.cpu_count()} cores, {sp.bank.threads} draw threads)")
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        run()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
