// mzx_platform.h -- compile-target glue.
//
// Product build: hipcc --offload-arch=gfx950 (MZX_HOSTCHECK undefined).  The
// element functors of mzx_ops.h run as HIP kernels; there is no CPU execution
// path in the product library.
//
// tests/hostcheck build (g++ -DMZX_HOSTCHECK): the SAME functors are executed by
// plain serial loops so the CPU test-suite can check the kernel logic against
// the oracle without a GPU.  Test infrastructure only -- the mzx package never
// loads that library.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#ifdef MZX_HOSTCHECK
#define MZX_HD
#define MZX_DEVICE_ONLY 0
#else
#include <hip/hip_runtime.h>
#define MZX_HD __host__ __device__
#define MZX_DEVICE_ONLY 1
#endif

#define MZX_INF (__builtin_inf())

namespace mzx {

// e^x for the network's activations and soft-maxes (every call site has x <= 0 or small).  Device:
// v_exp_f32(x * log2 e) -- two instructions instead of libm's fifteen; relative error <= ~|x| * 2^-24
// (1e-6 at x = -16), far inside the 1e-4 contract on the heads.  The generic operators and the
// whole-search kernels share it, so they stay bit-identical to each other on the device.
// -DMZX_IEEE_MATH (A/B builds only, tools/build_ieee_variant.sh): libm expf and the IEEE division instead, to measure
// what the fast forms cost in parity (profiles/r03_ieee_math_ab.txt).
MZX_HD inline float mzx_expf(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MZX_IEEE_MATH)
  return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#else
  return expf(x);
#endif
}
// a / b where one ulp does not matter (soft-max normalisation, min-max scaling): device v_rcp_f32 + multiply
// instead of the ten-instruction IEEE expansion.  NOT used by support_inverse_transform (cancellation).
MZX_HD inline float mzx_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MZX_IEEE_MATH)
  return a * __builtin_amdgcn_rcpf(b);
#else
  return a / b;
#endif
}
MZX_HD inline float mzx_expm1f(float x) { return expm1f(x); }
// torch.nn.ELU(alpha=1): x > 0 ? x : exp(x) - 1   (models.py:635).  exp(x) - 1 rather than
// expm1(x): a third of the instructions on gfx950, absolute error <= 2^-24 (the tolerance
// contract is 1e-4 absolute on the heads).
MZX_HD inline float mzx_elu(float x) { return x > 0.f ? x : (mzx_expf(x) - 1.0f); }

}  // namespace mzx
