"""
ISA facts of the shipped library per kernel: registers, scratch, spills (code-object metadata, llvm-readelf --notes) and the
instruction mix of its body (llvm-objdump -d): MFMAs, LDS reads / writes, global accesses, binary64 instructions, barriers.
No GPU needed.

    python muzero-general_amd/tools/isa_summary.py [path to libmzx.so] > profiles/rNN_isa_summary.txt
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = ("search", "tower", "gemm", "network", "row_select", "wave_select", "row_expand", "ObsStack")


def code_objects(lib):
    tmp = tempfile.mkdtemp(prefix="mzx_isa_")
    copy = os.path.join(tmp, "lib.so")
    shutil.copy(lib, copy)
    subprocess.run([LLVM + "llvm-objdump", "--offloading", copy], capture_output=True, check=True)      # extracts next to the file
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)


def metadata(path):
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
    kernels, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and "name" in cur:      # (the first key of the next kernel's map)
            kernels.append(cur)
            cur = {}
        cur[k] = v
    if "name" in cur:
        kernels.append(cur)
    return [k for k in kernels if "vgpr_count" in k]


def mixes(path):
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", path], capture_output=True, text=True).stdout
    out, sym = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            sym = m.group(1)
            out[sym] = collections.Counter()
            continue
        t = line.split()
        if sym is None or not t:
            continue
        op, c = t[0], out[sym]
        c["instr"] += 1
        for key, test in (("mfma", op.startswith("v_mfma")), ("ds_rd", op.startswith(("ds_read", "ds_load"))),
                          ("ds_wr", op.startswith(("ds_write", "ds_store"))),
                          ("vmem", op.startswith(("global_", "buffer_"))), ("scratch", op.startswith("scratch_")),
                          ("f64", "_f64" in op), ("bar", op.startswith("s_barrier")), ("dpp", "dpp" in line)):
            c[key] += bool(test)
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), "mzx", "libmzx.so")
    seen = collections.OrderedDict()
    for n, path in enumerate(code_objects(lib)):
        mix = mixes(path)
        for k in metadata(path):
            name = k["name"]
            if not any(s in name for s in KEEP):
                continue
            seen.setdefault(name, (k, mix.get(name, collections.Counter()), []))[2].append(n)
    names = subprocess.run(["c++filt"], input="\n".join(seen), capture_output=True, text=True).stdout.splitlines()
    print("ISA facts of %s (%d bytes): gfx950 code objects, one per translation unit (mzx_batched.hip, mzx_lib.cpp, mzx_tower_search.inc;\n"
          "kernels defined in headers that two units include are compiled into both).  vgpr = unified VGPR + AGPR budget of a lane (512 max;\n"
          "256 -> two waves per SIMD), scratch = bytes of private memory per lane, spill = spilled VGPRs; instruction counts are static\n"
          "(whole kernel body, loops counted once)." % (os.path.basename(lib), os.path.getsize(lib)))
    print("%-88s %4s %4s %4s %7s %5s | %6s %5s %5s %5s %5s %5s %4s %4s  objects" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "spill", "instr",
                                                                                     "mfma", "ds_rd", "ds_wr", "vmem", "f64", "dpp", "bar"))
    for (name, (k, c, objs)), pretty in zip(seen.items(), names):
        pretty = re.sub(r"\(anonymous namespace\)::", "", pretty).replace("void mzx::", "").split("(mzx::")[0]
        print("%-88s %4s %4s %4s %7s %5s | %6d %5d %5d %5d %5d %5d %4d %4d  %s" % (
            pretty[:88], k.get("vgpr_count"), k.get("agpr_count"), k.get("sgpr_count"), k.get("private_segment_fixed_size"),
            k.get("vgpr_spill_count", "0"), c["instr"], c["mfma"], c["ds_rd"], c["ds_wr"], c["vmem"], c["f64"], c["dpp"], c["bar"],
            ",".join(map(str, objs))))


if __name__ == "__main__":
    main()
