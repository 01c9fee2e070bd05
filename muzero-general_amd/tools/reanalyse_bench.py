"""
Throughput of the Reanalyse step (mzx.replay.Reanalyse.reanalyse_game: upload of a game's frames, device
stacking of every position, ONE batched initial_inference, value decode, download) per BASELINE network.

    python muzero-general_amd/tools/reanalyse_bench.py [--repeat 5]

One JSON line per configuration: positions/s and the initial_inference FLOP rate it corresponds to (wall clock of
the whole step incl. the upload of the frames), and the FLOP rate of the batched initial_inference alone on
observations already resident in HBM (HIP events).
"""
import argparse
import json
import os
import sys
import time

import numpy
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mzx import configs, models, replay, self_play, synthetic  # noqa: E402

CASES = [("cartpole", dict(), 500), ("tictactoe", dict(), 9), ("connect4", dict(), 42),
         ("connect4", dict(stacked_observations=4), 42), ("breakout", dict(), 2500),
         ("breakout", dict(stacked_observations=8), 1000)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=5)
    args = ap.parse_args()
    for name, kw, T in CASES:
        cfg = configs.BY_NAME[name](**kw)
        rs = numpy.random.RandomState(0)
        gh = self_play.GameHistory()
        gh.observation_history = [rs.rand(*cfg.observation_shape).astype(numpy.float32) for _ in range(T + 1)]
        gh.action_history = [0] + [int(a) for a in rs.randint(0, len(cfg.action_space), size=T)]
        gh.root_values = [0.0] * T
        template = models.MuZeroNetwork(cfg).state_dict()
        worker = replay.Reanalyse({"weights": synthetic.fill_state_dict(template, 0), "num_reanalysed_games": 0}, cfg)
        worker.reanalyse_game(gh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.repeat):
            values = worker.reanalyse_game(gh)
        dt = (time.perf_counter() - t0) / args.repeat
        flops = worker.model.backend.lib.mzx_net_flops(worker.model.handle, 0)
        # the batched initial_inference alone, observations already stacked in HBM (no upload / download): HIP events
        obs = torch.rand((T,) + tuple(worker.model.input_shape), device=worker.model.backend.device)
        worker.model.initial_inference(obs)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(args.repeat):
            worker.model.initial_inference(obs)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / args.repeat
        print(json.dumps({"config": name, "overrides": kw, "positions": T, "ms_per_game": round(dt * 1e3, 3),
                          "positions_per_s": round(T / dt, 1), "initial_inference_TFLOPs": round(T * flops / dt / 1e12, 3),
                          "device_resident_ms": round(ms, 3),
                          "device_resident_initial_inference_TFLOPs": round(T * flops / (ms * 1e-3) / 1e12, 3),
                          "finite": bool(numpy.isfinite(values).all())}))


if __name__ == "__main__":
    main()
