"""
ReplayBuffer.get_batch: the vectorised mzx.replay.ReplayBuffer against the reference's (replay_buffer.py:70-138),
host only.  With /root/reference present (build container) both are timed on the same seeded buffer; on the GPU
box only ours.  CartPole-like training shape: batch 128, 10 unroll steps, td_steps 50, 500-move games.

    python muzero-general_amd/tools/replay_batch_bench.py
"""
import os
import sys
import time
import types

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "muzero-general_amd"))
sys.path.insert(0, ROOT)
from mzx import replay, self_play  # noqa: E402
from oracle import ref_shim  # noqa: E402


def games(n, T, A=2):
    rs = numpy.random.RandomState(0)
    out = []
    for _ in range(n):
        gh = self_play.GameHistory()
        gh.action_history = [0] + [int(a) for a in rs.randint(0, A, size=T)]
        gh.reward_history = [0] + [1.0] * T
        gh.to_play_history = [0] * (T + 1)
        gh.root_values = [float(v) for v in rs.rand(T) * 20]
        gh.child_visits = [[0.5, 0.5] for _ in range(T)]
        gh.observation_history = [rs.rand(1, 1, 4).astype(numpy.float32) for _ in range(T + 1)]
        out.append(gh)
    return out


def main():
    config = types.SimpleNamespace(PER=True, PER_alpha=0.5, seed=0, replay_buffer_size=10 ** 6, batch_size=128,
                                   num_unroll_steps=10, td_steps=50, discount=0.997, stacked_observations=0,
                                   action_space=[0, 1], players=[0])
    data = games(64, 500)
    impls = [("mzx.replay.ReplayBuffer", replay.ReplayBuffer)]
    if ref_shim.available():
        ref_shim.load()
        import replay_buffer as ref_rb
        impls.append(("reference ReplayBuffer", ref_rb.ReplayBuffer))
    for name, cls in impls:
        rb = cls({"num_played_games": 0, "num_played_steps": 0}, {}, config)
        t0 = time.perf_counter()
        for gh in data:
            import copy
            rb.save_game(copy.deepcopy(gh))
        t_save = time.perf_counter() - t0
        rb.get_batch()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            rb.get_batch()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:28s} save_game {t_save / len(data) * 1e3:8.2f} ms per 500-move game   get_batch {dt * 1e3:8.2f} ms "
              f"({config.batch_size / dt:9.0f} samples/s)")


if __name__ == "__main__":
    main()
