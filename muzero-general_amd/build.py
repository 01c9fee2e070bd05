"""
Builds the gfx950 shared library ``mzx/libmzx.so`` from ``csrc/`` with hipcc.

In-tree and explicit (no JIT cache): the built .so travels to the GPU box with
the repository snapshot.  hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "mzx", "libmzx.so")

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    "-ffp-contract=off",      # tree statistics must not be fused (bit-exact binary64)
    "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip"))
    )


def up_to_date():
    if not os.path.isfile(OUT):
        return False
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.startswith("_")] + [os.path.join(INCLUDE, "mzx.h"), __file__]
    return all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)


def _object_up_to_date(obj, src):
    if not os.path.isfile(obj):
        return False
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [src, os.path.join(INCLUDE, "mzx.h"), __file__]
    return all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=True):
    """One object per translation unit (compiled concurrently, cached under csrc/_obj), then one link."""
    if not force and up_to_date():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # MZX_CXXFLAGS: extra flags, e.g. -DMZX_RZ_EXPERIMENT for the instrumented build that
    # tools/resnet_phase_profile.py (intra-operator stamps) and the MZX_RZ_DBG latency experiments need
    extra = os.environ.get("MZX_CXXFLAGS", "").split()
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in FLAGS if f != "-shared"]
    stamp = os.path.join(objdir, "flags.txt")
    flag_line = " ".join(flags + extra)
    if not os.path.isfile(stamp) or open(stamp).read() != flag_line:
        force = True
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not _object_up_to_date(obj, src):
            cmd = [hipcc] + flags + extra + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    with open(stamp, "w") as f:
        f.write(flag_line)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
