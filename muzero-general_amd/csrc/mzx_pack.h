// mzx_pack.h -- weight re-ordering functors shared by the two MFMA engines (mzx_resnet_fused.h,
// mzx_resnet_batched.h): B fragments of v_mfma_f32_16x16x4_f32 in lane order, zero-padded copies, the
// border-aware tap sums of the dynamics input's action plane (models.py:557-572).
#pragma once
#include "mzx_platform.h"

namespace mzx {

// ---------------------------------------------------------------------------
// weight packing (runs once per set_weights; element functor, also built by hostcheck)

struct RzPackOp {
  const float* W;   // [cout][cin_total][taps]
  float* out;       // [ntiles][wchunks][64 lanes][4], chunks >= nchunks are zero
  int32_t taps, cin, cin_total, cchunks, cout, nchunks, wchunks, ntiles;

  MZX_HD size_t size() const { return (size_t)ntiles * wchunks * 256; }
  MZX_HD void operator()(size_t i) const {
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const int c = (int)((i >> 8) % wchunks), nt = (int)((i >> 8) / wchunks);
    if (c >= nchunks) { out[i] = 0.f; return; }
    const int tap = c / cchunks, cc = c % cchunks;
    const int ci = cc * 16 + 4 * (lane >> 4) + j;      // K-step j of the chunk, B row lane >> 4
    const int n = nt * 16 + (lane & 15);
    float v = 0.f;
    if (ci < cin && n < cout) v = W[((int64_t)n * cin_total + ci) * taps + tap];
    out[i] = v;
  }
};

struct RzCopyOp {
  const float* src;
  float* dst;
  int32_t n, npad;
  MZX_HD size_t size() const { return (size_t)npad; }
  MZX_HD void operator()(size_t i) const { dst[i] = ((int)i < n) ? src[i] : 0.f; }
};

// out[co][pos] = sum over the 3x3 taps that stay inside the board of W[co][cin_total - 1][ky][kx]
struct RzAsumOp {
  const float* W;
  float* out;
  int32_t cout, cin_total, H, Wd;

  MZX_HD size_t size() const { return (size_t)cout * H * Wd; }
  MZX_HD void operator()(size_t i) const {
    const int p = (int)(i % (H * Wd)), co = (int)(i / (H * Wd));
    const int y = p / Wd, x = p % Wd;
    const float* w = W + ((int64_t)co * cin_total + (cin_total - 1)) * 9;
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = x + kx - 1;
        if (ix < 0 || ix >= Wd) continue;
        acc += w[ky * 3 + kx];
      }
    }
    out[i] = acc;
  }
};

}  // namespace mzx
