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
GATES = {}


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
