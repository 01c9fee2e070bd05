"""
TEST INFRASTRUCTURE ONLY.

Builds the mzx C ABI as a serial host library (g++ -DMZX_HOSTCHECK): the same
element functors and host driver the HIP kernels are made of, executed by plain
loops.  It lets the CPU test-suite (`-m "not gpu"`) check the kernel LOGIC --
tree arithmetic bit-for-bit, network operators within tolerance -- against the
oracle before any GPU time is spent.  The mzx package never loads this library.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "muzero-general_amd", "csrc")
LIB = os.path.join(HERE, "libmzx_hostcheck.so")


def build(force=False):
    srcs = [os.path.join(SRC, f) for f in os.listdir(SRC)] + [os.path.join(ROOT, "include", "mzx.h")]
    if not force and os.path.isfile(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DMZX_HOSTCHECK",
           "-x", "c++", os.path.join(SRC, "mzx_lib.cpp"), "-o", LIB]
    subprocess.run(cmd, check=True)
    return LIB


def backend():
    from mzx import _lib
    return _lib.Backend(_lib.Library(build()), "cpu")
