"""
Summarise rocprofv3 CSV output directories (``--output-format csv``) into a small text file that
can be committed under profiles/:  kernel-stats table + per-kernel means of every PMC counter.

    python muzero-general_amd/tools/rocprof_summary.py gpurun_out/<tag> [kernel-substring] > profiles/<name>.txt
"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    needle = sys.argv[2] if len(sys.argv) > 2 else ""
    for path in sorted(glob.glob(os.path.join(root, "**", "*_kernel_stats.csv"), recursive=True)):
        print(f"== kernel stats ({os.path.relpath(path, root)}): rocprofv3 --kernel-trace --stats")
        print("calls  total_us  avg_us  pct  name")
        for r in csv.DictReader(open(path)):
            print(f'{r["Calls"]:>5} {float(r["TotalDurationNs"]) / 1e3:>10.1f} {float(r["AverageNs"]) / 1e3:>9.2f} '
                  f'{float(r["Percentage"]):>6.2f}  {r["Name"][:150]}')
        print()
    for path in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(path)))
        agg = collections.OrderedDict()
        meta = {}
        for r in rows:
            if needle and needle not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"][:110], r["Counter_Name"])
            agg.setdefault(key, []).append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"][:110]] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                                              "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if k in r}
        if not agg:
            continue
        print(f"== PMC ({os.path.relpath(path, root)}): per-dispatch mean over n dispatches")
        for (kernel, counter), v in agg.items():
            print(f"{counter:<24} n={len(v):<4} mean={sum(v) / len(v):<16.1f} {kernel}")
        for k, m in meta.items():
            print(f"   dispatch of {k}: {m}")
        print()


if __name__ == "__main__":
    main()
