"""
Per-operator shader-clock profile of the fused residual engine (workgroup 0), on the GPU box:

    python muzero-general_amd/tools/resnet_phase_profile.py c3 1024

The intra-operator stamps need the instrumented build
(MZX_CXXFLAGS=-DMZX_RZ_EXPERIMENT python muzero-general_amd/build.py --force); the per-operator totals
work with the product build.
"""
import os
import sys

import numpy
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "muzero-general_amd"))

from mzx import configs, models, synthetic  # noqa: E402

NAMES = {"c3": "tictactoe", "c4": "connect4", "c5": "breakout"}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    cfg = configs.BY_NAME[NAMES[wl]]()
    net = models.MuZeroNetwork(cfg)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 0))
    rs = numpy.random.RandomState(0)
    hid = torch.tensor(rs.rand(batch, *net.hidden_shape).astype(numpy.float32))
    obs = torch.tensor(rs.rand(batch, *net.input_shape).astype(numpy.float32))
    act = torch.tensor(rs.randint(0, len(cfg.action_space), size=batch).astype(numpy.int32))
    for recurrent, x, a in ((1, hid, act), (0, obs, None)):
        n = net.num_operators(recurrent)
        for _ in range(3):
            out = net.debug_prefix(recurrent, 2, n, x, a)
        torch.cuda.synchronize()
        raw = out.cpu().numpy().reshape(-1).view(numpy.uint64)
        fine = raw[52: 52 + 8 * 48].reshape(48, 8).astype(numpy.int64)
        stamps = raw[: n + 4]
        stamps = stamps[stamps > 0]
        d = numpy.diff(stamps.astype(numpy.int64))
        print(f"{wl} {'recurrent' if recurrent else 'initial'} batch {batch}: {len(d)} intervals, total {d.sum()} cycles")
        print("  staging %d; input load %d; ops: %s" % (d[0], d[1], " ".join(str(int(v)) for v in d[2:])))
        # operators sharing a slot (no data dependence, disjoint wave teams) carry one stamp set: that of the
        # slot's first operator, which wave 0 runs
        import ctypes
        sched = (ctypes.c_int32 * n)()
        n_slots = net.backend.lib.mzx_net_fused_schedule(net.handle, recurrent, sched, n)
        members = {}
        for k, sl in enumerate(sched):
            if sl >= 0:
                members.setdefault(sl, []).append(k)
        print(f"  {n_slots} slots: " + " ".join("{" + ",".join(str(k) for k in members[sl]) + "}" for sl in sorted(members)))
        print("  intra-slot (wave 0, first operator): descriptor fetch | dispatch | row bases | K loop | epilogue | other operators of the slot | barrier")
        for o in range(48):
            f = fine[o]
            if f[1] and f[4] and f[6] > f[0]:
                print("   table %2d: %6d %6d %6d %6d %6d %6d %6d   = %6d" % (o, f[7] - f[0], f[1] - f[7], f[2] - f[1], f[3] - f[2], f[4] - f[3], f[5] - f[4], f[6] - f[5], f[6] - f[0]))


if __name__ == "__main__":
    main()
