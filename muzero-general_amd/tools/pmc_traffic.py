"""
HBM traffic per STEP of a bench workload from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs of
`python bench.py --workload W --steps N --warmup 0 --repeats 1 --also none ...`, N steps and nothing else of the search path
in the process), as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes for gfx950: bytes = 2 x FETCH_SIZE + WRITE_SIZE
(FETCH_SIZE tallies the 128-byte fabric requests of wide coalesced reads at 64 bytes; both counters are in KiB).  Sums
every dispatch of the library's kernels (mzx::*) except the once-per-set_weights packing kernels, divides by the steps.

    python muzero-general_amd/tools/pmc_traffic.py <workload> <steps> <dir with pmc_fetch/ pmc_write/ [pmc_mfma/ pmc_l2/]> <kernel tag> >> entry

Prints one JSON object {workload: {...}}; gpu_job.sh merges them into profiles/pmc_traffic.json.
"""
import collections
import csv
import glob
import json
import os
import sys

SETUP = ("RzPackOp", "RzAsumOp", "BnFoldOp", "RzCopyOp")


def rows(root, sub):
    out = []
    for path in glob.glob(os.path.join(root, sub, "**", "*_counter_collection.csv"), recursive=True):
        out += list(csv.DictReader(open(path)))
    return out


def total(rws, counter):
    per_kernel = collections.OrderedDict()
    tot = 0.0
    for r in rws:
        name = r["Kernel_Name"]
        if r["Counter_Name"] != counter or "mzx::" not in name or any(s in name for s in SETUP):
            continue
        v = float(r["Counter_Value"])
        tot += v
        k = name.split("(")[0][-70:]
        e = per_kernel.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += v
    return tot, per_kernel


def main():
    workload, steps, root, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    fetch, fk = total(rows(root, "pmc_fetch"), "FETCH_SIZE")
    write, wk = total(rows(root, "pmc_write"), "WRITE_SIZE")
    entry = {
        "fetch_kib_per_step": fetch / steps, "write_kib_per_step": write / steps,
        "bytes_per_step": int((2.0 * fetch + write) * 1024 / steps), "steps": steps, "kernel_tag": tag,
        "dispatches_per_step": sum(v[0] for v in fk.values()) / steps,
        "top_kernels_read_mb_per_step": {k: round(2.0 * v[1] * 1024 / 1e6 / steps, 2) for k, v in
                                         sorted(fk.items(), key=lambda kv: -kv[1][1])[:4]},
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of bench.py --workload %s --steps %d --warmup 0 "
                  "--repeats 1 (gpu_job.sh pmc); 2 x FETCH_SIZE + WRITE_SIZE, KiB" % (workload, steps),
    }
    mf = rows(root, "pmc_mfma")
    if mf:
        busy, _ = total(mf, "SQ_VALU_MFMA_BUSY_CYCLES")
        gui, _ = total(mf, "GRBM_GUI_ACTIVE")
        wavec, _ = total(mf, "SQ_WAVE_CYCLES")
        if gui > 0:
            entry["mfma_busy_share_of_simd_cycles"] = round(busy / (1024.0 * gui / 8.0), 4)    # 1024 SIMDs; GRBM_GUI_ACTIVE sums 8 XCDs
            entry["wave_slots_occupied_per_simd"] = round(4.0 * wavec / (1024.0 * gui / 8.0), 3)   # SQ_WAVE_CYCLES: quad-cycles
    l2 = rows(root, "pmc_l2")
    if l2:
        hit, _ = total(l2, "TCC_HIT_sum")
        miss, _ = total(l2, "TCC_MISS_sum")
        if hit + miss > 0:
            entry["l2_hit_rate"] = round(hit / (hit + miss), 4)
    print(json.dumps({workload: entry}))


if __name__ == "__main__":
    main()
