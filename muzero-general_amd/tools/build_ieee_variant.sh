#!/bin/bash
# A/B build of the library with libm expf and IEEE division in the soft-max / scaling / decode helpers
# (csrc/mzx_platform.h: -DMZX_IEEE_MATH) -> muzero-general_amd/mzx/libmzx_ieee.so; select with MZX_LIB=<path>.
# Used once per round to measure what the fast v_exp_f32 / v_rcp_f32 forms cost in parity (profiles/*ieee_math_ab*).
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function \
  -DMZX_IEEE_MATH -x hip "$HERE/csrc/mzx_batched.hip" "$HERE/csrc/mzx_lib.cpp" -o "$HERE/mzx/libmzx_ieee.so"
ls -la "$HERE/mzx/libmzx_ieee.so"
