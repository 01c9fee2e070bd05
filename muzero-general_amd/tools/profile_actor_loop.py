"""cProfile of the MAIN thread of bench.py's selfplay_actor_loop leg (C2, natively played shard): where the Python time of
the hand-off goes while the worker thread is inside mzx_selfplay_rounds.  python muzero-general_amd/tools/profile_actor_loop.py"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import bench  # noqa: E402

from mzx import configs, models  # noqa: E402

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
