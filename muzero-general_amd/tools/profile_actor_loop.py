"""cProfile of the MAIN thread of bench.py's selfplay_actor_loop leg (C2, natively played shard): where the Python time of
the hand-off goes while the worker thread is inside mzx_selfplay_rounds.  python muzero-general_amd/tools/profile_actor_loop.py"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

from mzx import configs, models  # noqa: E402

if len(sys.argv) > 1:      # e.g. "atari gomoku": run these search workloads first, as the default bench line does
    after = sys.argv[1:]
    sys.argv = ["bench.py"]
    args = bench.parse()
    env = bench.Env(args)
    for w in after:
        bench.run_search_workload(env, args, w, 1, 1, compact=True)
cfg = configs.cartpole()
net = models.MuZeroNetwork(cfg)
net.set_weights(bench.bench_weights(cfg, net, "c2", "reference")[0])
bench.actor_loop_leg(cfg, net, 4096, 32, shards=2)
pr = cProfile.Profile()
pr.enable()
out = bench.actor_loop_leg(cfg, net, 4096, 32, shards=6)
pr.disable()
print(out)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
import gc
print("gc", gc.get_count(), gc.get_stats())
