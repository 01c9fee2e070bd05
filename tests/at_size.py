"""
What an at-size GPU parity case reports and gates, beyond "which simulation took which branch": the quantities the replay
buffer CONSUMES from a search (GameHistory.store_search_statistics, self_play.py:496-511) -- the root's child visit
distribution ``child_visits[a] = visit_count[a] / sum(visit_count)`` and the root value -- device against the CPU oracle
(fp32, the reference's arithmetic) on the sampled trees of the run:

  identical      trees whose every simulation expands the oracle's (parent, action)
  roots_equal    trees whose root visit COUNTS equal the oracle's (a search may leave the oracle's line deep in the tree and
                 still hand the same counts to the buffer)
  max_dvisits    max over sampled trees and actions of |child_visits_device - child_visits_oracle|  (= |count diff| / S)
  mean_l1        mean over sampled trees of sum_a |child_visits_device - child_visits_oracle|
  max_dvalue     max over sampled trees of |root value device - oracle| / max(1, |oracle|)
and the same five numbers for the ORACLE'S OWN fp32 search against its binary64 evaluation of the same trees (`own_*`): how
far the reference's arithmetic is from exact arithmetic on these inputs -- printed beside, never part of a gate.

Gates are ABSOLUTE, per case (the round-5 gate was relative to the oracle's own instability and could not
fail where that was total).  The table below was calibrated on an MI355X box (profiles/r06_pytest_gpu_*.log): every bound
leaves roughly a factor two (counts: a few trees) over what was measured; device arithmetic is deterministic and the
oracle runs on the same image's torch CPU kernels, so the measured figures repeat.  A case without an entry fails: a new
at-size case has to be measured and entered.
"""
import os

import numpy

# label -> (min share of trees identical in every simulation, min share with equal root visit counts,
#           max |delta child_visits|, max mean L1 of child_visits, max relative |delta root value|)
GATES = {
    # measured (profiles/r06_pytest_gpu_calibrate.log), n sampled trees:                identical  roots equal  max dv   L1        d value
    "C2 mode 0": (0.92, 0.98, 0.04, 0.002, 1e-3),                                     # 246/256    256/256      0        0         4.4e-5
    "C2 mode 1": (0.93, 0.99, 0.04, 0.002, 2e-2),                                     # 990/1024   1024/1024    0        0         5.9e-3
    "C2 mode 1 (reference weights)": (0.93, 0.99, 0.30, 0.002, 5e-2),                 # 993/1024   1021/1024    0.14     5.5e-4    2.0e-2
    "C2 mode 1 (checkpoint weights)": (0.90, 0.99, 0.06, 0.001, 1.5e-2),              # 960/1024   1023/1024    0.02     4e-5      4.4e-3
    "gomoku-shaped": (0.80, 0.90, 0.05, 0.005, 2e-3),                                 # 24/24      24/24        0        0         1.8e-4
    "tictactoe x 1024": (0.95, 0.97, 0.10, 0.002, 2e-2),                              # 255/256    255/256      0.04     3.1e-4    7.1e-3
    "tictactoe x 1024 (reference weights)": (0.93, 0.97, 0.10, 0.002, 2e-2),          # 250/256    255/256      0.04     3.1e-4    3.2e-3
    "connect4 x 1024": (0.70, 0.95, 0.05, 0.002, 2e-2),                               # 213/256    253/256      0.01     2.0e-4    6.2e-3
    "connect4-ws x 1024": (0.75, 0.92, 0.05, 0.003, 2e-2),                            # 58/64      63/64        0.01     3.1e-4    5.7e-3
    "connect4 x 1024 (reference weights)": (0.40, 0.90, 0.05, 0.003, 1e-2),           # 35/64      63/64        0.005    1.6e-4    1.7e-3
    "breakout x 64": (0.90, 0.93, 0.05, 0.003, 1e-3),                                 # 32/32      32/32        0        0         2.3e-5
    "breakout x 64 (reference weights)": (0.80, 0.93, 0.05, 0.003, 1e-3),             # 30/32      32/32        0        0         3.9e-5
    "breakout x 512 (reference weights)": (0.80, 0.93, 0.10, 0.005, 5e-3),            # 59/64      63/64        0.04     1.25e-3   8.9e-4
    # streamed engine at the bench's sizes (tests/test_gpu_streamed_at_size.py)
    "atari-1024": (0.75, 0.75, 0.10, 0.02, 1e-3),                                     # 8/8        8/8          0        0         1.1e-5
    "atari-256": (0.75, 0.75, 0.10, 0.02, 1e-3),                                      # 4/4        4/4          0        0         1.0e-5
    "atari-256 (reference weights)": (0.75, 0.75, 0.10, 0.02, 1e-3),                  # 4/4        4/4          0        0         5.4e-5
    "connect4-1024": (0.65, 0.92, 0.05, 0.003, 5e-3),                                 # 53/64      64/64        0        0         1.2e-4
    "connect4-1024 (reference weights)": (0.35, 0.88, 0.05, 0.004, 5e-3),             # 32/64      61/64        0.015    1.1e-3    9.4e-4
    "connect4-9216": (0.60, 0.88, 0.12, 0.006, 5e-2),                                 # 48/64      61/64        0.045    1.7e-3    1.5e-2
    # games/gomoku.py: 400 simulations dig 100- to 400-ply single lines; the oracle's OWN fp32 and binary64 searches share no
    # tree and end at distributions 0.27 .. 0.40 apart (L1 0.10 .. 0.23) -- the device is as far from the fp32 oracle as exact
    # arithmetic is.  The bounds are what a search on garbage would break (max dv -> 1, L1 -> 2), no more; the margin gate at
    # every first divergence is the sharp check for these trees
    "gomoku-1024": (0.0, 0.0, 0.70, 0.40, 3.0),                                       # 0/16       2/16         0.415    0.158     1.05
    "gomoku-1024 (reference weights)": (0.0, 0.35, 0.65, 0.35, 1.0),                  # 0/16       10/16        0.39     0.157     0.37
    # rt_search_kernel at the other shard sizes (tests/test_gpu_tower_search.py)
    "connect4 x 512 on rt_search_kernel": (0.65, 0.90, 0.05, 0.004, 5e-3),            # 27/32      32/32        0        0         1.3e-4
    "connect4 x 1536 on rt_search_kernel": (0.50, 0.90, 0.05, 0.004, 1e-2),           # 31/48      48/48        0        0         7.7e-4
    "connect4 x 9216 on rt_search_kernel": (0.55, 0.90, 0.08, 0.005, 0.5),            # 34/48      47/48        0.03     1.25e-3   0.19
}


def statistics(S, counts, values, s32, s64, identical):
    """counts [n][A] / values [n]: the device's root visit counts and root values of the sampled trees; s32 / s64: the
    oracle summaries (oracle/parallel.py) of the same trees in fp32 and binary64; identical: trees the caller found
    identical in every simulation."""
    n = len(s32)
    counts = numpy.asarray(counts, numpy.int64).reshape(n, -1)
    want = numpy.asarray([t["root_visit_counts"] for t in s32], numpy.int64)
    exact = numpy.asarray([t["root_visit_counts"] for t in s64], numpy.int64)
    v32 = numpy.asarray([t["root_value"] for t in s32], numpy.float64)
    v64 = numpy.asarray([t["root_value"] for t in s64], numpy.float64)
    values = numpy.asarray(values, numpy.float64).reshape(n)

    def dist(a, b, va, vb):
        d = numpy.abs(a - b) / float(S)
        return dict(roots_equal=int((a == b).all(1).sum()), max_dvisits=float(d.max(initial=0.0)),
                    mean_l1=float(d.sum(1).mean()) if n else 0.0,
                    max_dvalue=float((numpy.abs(va - vb) / numpy.maximum(1.0, numpy.abs(vb))).max(initial=0.0)))

    dev, own = dist(counts, want, values, v32), dist(want, exact, v32, v64)
    own_identical = sum(int(a["trace"] == b["trace"]) for a, b in zip(s32, s64))
    return dict(n=n, S=int(S), identical=int(identical), own_identical=own_identical, **dev,
                **{"own_" + k: v for k, v in own.items()})


def report(label, st):
    n = st["n"]
    print(f"{label}: visit statistics of {n} sampled trees x {st['S']} simulations, device vs the fp32 oracle: identical in every "
          f"simulation {st['identical']}/{n}; root visit counts equal {st['roots_equal']}/{n}; max |d child_visits| "
          f"{st['max_dvisits']:.4f}; mean L1(child_visits) {st['mean_l1']:.5f}; max |d root value| {st['max_dvalue']:.2e}   "
          f"[the oracle's own fp32 vs its binary64 evaluation: {st['own_identical']}/{n}; {st['own_roots_equal']}/{n}; "
          f"{st['own_max_dvisits']:.4f}; {st['own_mean_l1']:.5f}; {st['own_max_dvalue']:.2e}]")


def gate(label, st):
    """Absolute bounds (see GATES).  Returns nothing; raises AssertionError with the figures."""
    report(label, st)
    if os.environ.get("MZX_AT_SIZE_CALIBRATE"):      # measuring run: print the figures of every case, gate nothing
        return
    assert label in GATES, f"at-size case {label!r} has no calibrated gate in tests/at_size.py: {st}"
    min_identical, min_roots, max_dv, max_l1, max_dval = GATES[label]
    n = st["n"]
    assert st["identical"] >= min_identical * n - 1e-9, (label, "trees identical in every simulation", st)
    assert st["roots_equal"] >= min_roots * n - 1e-9, (label, "root visit counts equal", st)
    assert st["max_dvisits"] <= max_dv, (label, "max |d child_visits|", st)
    assert st["mean_l1"] <= max_l1, (label, "mean L1 of child_visits", st)
    assert st["max_dvalue"] <= max_dval, (label, "root value", st)
