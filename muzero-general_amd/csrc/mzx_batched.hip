// mzx_batched.hip -- kernels of the STREAMED residual-network engine (plan: mzx_resnet_batched.h).
//
// rb_gemm_kernel<MT, NT>: one layer of MuZeroResidualNetwork (models.py:206-623) -- 3x3 convolution with folded
// BatchNorm / residual / ReLU / action plane, 1x1 head convolution, Linear (+ ELU) -- as an FP32-MFMA implicit GEMM
// over the whole batch.  A 512-thread workgroup (8 waves, two per SIMD) owns a tile of up to 144 GEMM rows =
// (sample, output position) pairs and up to 16 column tiles of 16 output channels:
//   * the tile's input patch (with the K x K halo, out-of-image cells zero) is staged ONCE into LDS, position-major
//     [cell][Cs] with Cs = 8 mod 16 floats -- a tap is a constant LDS offset, no bounds tests in the K loop;
//     patches that do not fit are staged in channel groups (phases), accumulators live across phases;
//   * K runs over (tap, 16-channel chunk): lane (row l & 15, group g = l >> 4) reads channels 4g..4g+3 of its row
//     with one ds_read_b128 = the A operands of four v_mfma_f32_16x16x4_f32 K-steps; B fragments come pre-packed
//     in the same lane order (RzPackOp), 16 bytes per lane per chunk, straight from L2 (every workgroup streams the
//     layer's weights once; all workgroups read the same image, so it stays L2-resident).  Weight fragments are
//     double-buffered one chunk ahead; the position fragments of the larger tilings are ONE set refilled in place
//     (tile i's read for the next chunk is issued right after its last MFMA of this one), the smaller tilings keep two
//     sets or a four-chunk ring;
//   * waves take column tiles first (each wave streams a disjoint slice of the weights), row tiles next; a wave
//     owns up to 9 x 2 accumulator tiles, one B fragment feeds up to nine MFMAs;
//   * epilogue in registers in the per-operator kernels' order (mzx_ops.h): action term, alpha * acc + beta, bias,
//     residual, ReLU / ELU.  The weight fragment is the MFMA's FIRST operand (D = (A . B)^T), so a lane ends up with
//     four consecutive output channels of ONE position: one 16-byte residual load and one 16-byte store per lane
//     and tile, position-major (NHWC), or four position-coalesced scalar stores (NCHW head tensors).
// The f32 MFMA is a k-ordered fmaf chain; only the summation ORDER differs from ATen's / Conv3x3Op's.
// Roofline: FP32 matrix pipe, 157.3 TFLOP/s dense.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "mzx_pack.h"
#include "mzx_resnet_batched.h"
#include "mzx_tower.h"

#ifndef MZX_RB_RING_MAX
#define MZX_RB_RING_MAX 4   // accumulator tiles per wave up to which the K loop keeps four chunks in flight
#endif
namespace mzx {

namespace {

struct RbGemmArgs {
  RbTensor x, res;       // res.p null: no residual
  float* y;
  int64_t y_sstride;
  const float* wpack;
  const float* alpha;    // folded BatchNorm (null: none)
  const float* beta;
  const float* bias;     // null: none
  const float* asum;     // [cout][hout * wout] tap sums of the action plane (null: none)
  const int32_t* action;
  int32_t num_actions, batch;
  int32_t cin, cout, hin, win, hout, wout, stride, taps, ksize, pad;
  int32_t kmagic, rowskip;   // ceil(2^16 / ksize): tap / ksize by multiply-shift (taps <= 64); (PW - ksize) * Cs
  int32_t T, th, tw, tiles_x, tiles_y, PH, PW, Cs, cpg, phases, cchunks, wchunks, rows, mtiles;
  int32_t ntiles, ntiles_wg, WN, WM, y_layout, act, x_vec, y_vec, r_vec;   // *_vec: 16-byte accesses are aligned
  int32_t rowsplit;      // > 0: ROW-RANGE tiles -- workgroup sp of a sample owns its row tiles [sp * rowsplit, (sp + 1) * rowsplit)
                         // (positions in raster order; stride-1 operators, T = 1); the patch is the board rows they touch
  unsigned long long* stamps;   // diagnostics (env MZX_RB_STAMPS): s_memtime of the phases, wave 0 of the first 64 workgroups
  int32_t dbg;           // latency experiments (env MZX_RB_DBG): 1 skip the K loops, 2 skip the epilogue, 4 skip the staging
  uint32_t magic_thw, magic_tw, magic_phw, magic_pw;   // ceil(2^32 / d)
  int32_t grid_x, grid_y;   // this operator's own grid (rb_gemm_multi_kernel: the launch's grid is the largest of its slices')
};

// Two workgroups share a CU (their LDS tiles are planned for it, RB_LDS_BUDGET): one stages its patch or writes its
// outputs while the other keeps the matrix pipes busy.  Four waves per SIMD = 128 registers per lane, which the
// tilings with at most nine accumulator tiles per wave fit; the 2 x 9 tilings (wide layers) run one workgroup per CU.
constexpr int RB_STAMP_WGS = 1024, RB_STAMP_SLOTS = 24;
#define RB_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (a.stamps && blockIdx.y == 0 && blockIdx.x < RB_STAMP_WGS && (tid & 63) == 0 && (slot) < RB_STAMP_SLOTS)  \
      a.stamps[((size_t)blockIdx.x * 8 + (tid >> 6)) * RB_STAMP_SLOTS + (slot)] = __builtin_readcyclecounter();   \
  } while (0)

template <int MT, int NT>
__global__ void __launch_bounds__(RB_THREADS) __attribute__((amdgpu_waves_per_eu(MT * NT <= 8 ? 4 : 2, MT * NT <= 8 ? 4 : 2)))
rb_gemm_kernel(const RbGemmArgs a) {
#define RB_BX blockIdx.x
#define RB_BY blockIdx.y
#include "mzx_batched_gemm_body.inc"
#undef RB_BX
#undef RB_BY
}

// Several INDEPENDENT operators of one shape class in one launch -- the k-th Linear layers of the head MLPs (reward, value,
// policy: csrc/mzx_resnet_batched.h rb_find_heads): slice blockIdx.z runs operator z with its own arguments, launch shape
// and grid (the launch's grid is the largest; a slice's surplus workgroups leave at once).  Same code, same shapes, same
// bits as one launch per operator.
constexpr int RB_MULTI_MAX = 3;
struct RbGemmMulti { RbGemmArgs g[RB_MULTI_MAX]; };

template <int MT, int NT>
__global__ void __launch_bounds__(RB_THREADS) __attribute__((amdgpu_waves_per_eu(MT * NT <= 8 ? 4 : 2, MT * NT <= 8 ? 4 : 2)))
rb_gemm_multi_kernel(const RbGemmMulti m) {
  const RbGemmArgs& a = m.g[blockIdx.z];
  if (blockIdx.x >= (unsigned)a.grid_x || blockIdx.y >= (unsigned)a.grid_y) return;
#define RB_BX blockIdx.x
#define RB_BY blockIdx.y
#include "mzx_batched_gemm_body.inc"
#undef RB_BX
#undef RB_BY
}

// ---------------------------------------------------------------------------------------------------------------
// rb_tower_kernel<MT, NT>: a whole TOWER -- conv3x3 + BatchNorm + ReLU followed by residual blocks (ResidualBlock.forward,
// models.py:213-229), i.e. the representation / dynamics / prediction trunks of MuZeroResidualNetwork -- in ONE launch.
// A 512-thread workgroup owns T whole samples for all layers of the run:
//   * the input patch (zero halo, every channel of the tower's width) is staged ONCE into LDS, position-major
//     [cell][Cs] as in rb_gemm_kernel;
//   * per layer: the K loop of rb_gemm_kernel over (tap, 16-channel chunk) with the layer's B fragments streamed from L2
//     -> workgroup barrier (every wave is done reading the tile) -> epilogue in registers (action term, folded
//     BatchNorm, residual, ReLU) -> the result OVERWRITES the tile's interior cells in place (the zero halo of the
//     input is the zero padding of every later layer) -> barrier.  One activation tile per sample instead of the three
//     slots of the LDS-resident engine: connect4 (64 channels, 6 x 7) holds three boards in 62 KB, so TWO workgroups
//     share a CU and hide each other's barriers and epilogues;
//   * the residual of a block never exists in memory: before a wave overwrites the interior cells with the output of
//     the block's first convolution it keeps the elements IT will add two layers later -- its own (row, channel)
//     footprint, the same lanes -- in registers;
//   * only the last layer's output goes to memory (position-major, as rb_gemm_kernel writes it).
// Per residual block this moves 2 x [positions][C] floats (stage, final store: once per TOWER, not per block) instead of
// five transfers per block of the layer-by-layer path, and pays no launch / staging / epilogue tail per layer.
// Summation order per output: taps in order, 16-channel chunks in order, the MFMA's k order -- independent of T and of
// the batch (two half-shards build the trees of the undivided run).

template <int MT, int NT>
__global__ void __launch_bounds__(RB_THREADS) __attribute__((amdgpu_waves_per_eu(MT * NT <= 4 ? 4 : 2, MT * NT <= 4 ? 4 : 2)))
rb_tower_kernel(const RbTowerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rb_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mpad = a.mtiles * 16, Tpad = (a.T + 1) & ~1;
  int* rowaddr = (int*)rb_lds;
  int* rowt = rowaddr + mpad;
  int* rowpos = rowt + mpad;
  long long* soff_x = (long long*)(rowpos + mpad);
  float* tile = (float*)(soff_x + Tpad);
  const int b0 = blockIdx.x * a.T;
  const int HW = a.H * a.W, phw = a.PH * a.PW, cells = a.T * phw;

  // ---- row tables: row m = (sample t, position r); its 3 x 3 window starts at cell (y, x) of the haloed board
  for (int m = tid; m < mpad; m += RB_THREADS) {
    const int t = rb_div(m, HW, a.magic_hw), r = m - t * HW;
    const int y = rb_div(r, a.W, a.magic_w), x = r - y * a.W;
    const bool valid = m < a.rows;
    rowaddr[m] = valid ? ((t * a.PH + y) * a.PW + x) * a.Cs : 0;
    rowt[m] = valid ? t : 0;
    rowpos[m] = (valid && b0 + t < a.batch) ? r : -1;
  }
  for (int t = tid; t < a.T; t += RB_THREADS) {
    const int b = (b0 + t < a.batch) ? b0 + t : a.batch - 1;
    soff_x[t] = ((long long)b * a.x.nodes + (a.x.node ? a.x.node[b] : 0)) * a.x.sstride;
  }
  __syncthreads();

  // ---- stage the input: EVERY channel of the tile (zero beyond cin0: later layers read all of them), zero halo
  {
    const int q = a.cchunks * 4;                      // channel quads per cell
    const int total = cells * q;
    const float* xp = a.x.p;
    if (a.x.layout == RB_NHWC) {
      const uint32_t magic_q = (uint32_t)((0x100000000ull + (uint64_t)q - 1) / (uint64_t)q);
      const bool vec = (a.cin0 & 3) == 0 && (a.x.sstride & 3) == 0 && ((uintptr_t)xp & 15) == 0;
      for (int i0 = tid; i0 < total; i0 += 4 * RB_THREADS) {
        f32x4 v[4];
        int at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = i0 + u * RB_THREADS;
          at[u] = -1;
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (idx < total) {
            const int cell = rb_div(idx, q, magic_q), k = idx - cell * q;
            const int t = rb_div(cell, phw, a.magic_phw), rem = cell - t * phw;
            const int iy = rb_div(rem, a.PW, a.magic_pw), ix = rem - iy * a.PW;
            const int gy = iy - 1, gx = ix - 1, c = 4 * k;
            at[u] = cell * a.Cs + 4 * k;
            if (b0 + t < a.batch && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.cin0) {
              const float* src = xp + soff_x[t] + ((long long)gy * a.W + gx) * a.cin0 + c;
              if (vec && c + 3 < a.cin0) v[u] = *(const f32x4*)src;
              else {
                v[u][0] = src[0];
                if (c + 1 < a.cin0) v[u][1] = src[1];
                if (c + 2 < a.cin0) v[u][2] = src[2];
                if (c + 3 < a.cin0) v[u][3] = src[3];
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (at[u] >= 0) *(f32x4*)(tile + at[u]) = v[u];
      }
    } else {
      // NCHW source (observations, hidden states of the search arena): consecutive threads take consecutive cells of
      // one channel quad (coalesced along x)
      const uint32_t magic_cells = (uint32_t)((0x100000000ull + (uint64_t)cells - 1) / (uint64_t)cells);
      const long long plane = (long long)HW;
      for (int i0 = tid; i0 < total; i0 += 2 * RB_THREADS) {
        f32x4 v[2];
        int at[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int idx = i0 + u * RB_THREADS;
          at[u] = -1;
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (idx < total) {
            const int k = rb_div(idx, cells, magic_cells), cell = idx - k * cells;
            const int t = rb_div(cell, phw, a.magic_phw), rem = cell - t * phw;
            const int iy = rb_div(rem, a.PW, a.magic_pw), ix = rem - iy * a.PW;
            const int gy = iy - 1, gx = ix - 1, c = 4 * k;
            at[u] = cell * a.Cs + 4 * k;
            if (b0 + t < a.batch && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.cin0) {
              const float* src = xp + soff_x[t] + (long long)c * plane + (long long)gy * a.W + gx;
              v[u][0] = src[0];
              if (c + 1 < a.cin0) v[u][1] = src[plane];
              if (c + 2 < a.cin0) v[u][2] = src[2 * plane];
              if (c + 3 < a.cin0) v[u][3] = src[3 * plane];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (at[u] >= 0) *(f32x4*)(tile + at[u]) = v[u];
      }
    }
  }

#define RB_TOWER_HAS_TAIL (a.scale_y != nullptr || a.n_conv > 0)
#define RB_TOWER_ACTION(b, t) a.action[b]
#define RB_TOWER_LDS_BARRIER false
#define RB_TOWER_SHORT_GROUP false
#include "mzx_tower_layers.inc"
#undef RB_TOWER_HAS_TAIL
#undef RB_TOWER_ACTION
#undef RB_TOWER_LDS_BARRIER
#undef RB_TOWER_SHORT_GROUP
  if (!has_tail) return;
  __syncthreads();                                     // the last layer's output is in the tile's interior cells
  const int in0 = (a.PW + 1) * a.Cs;                   // cell (0, 0) of a board inside its halo
  if (a.scale_y) {
    float* lo = tile + (size_t)cells * a.Cs;
    float* sc = lo + a.T * a.C;
    for (int idx = tid; idx < a.T * a.C; idx += RB_THREADS) {
      const int t = rb_div(idx, a.C, rb_magic_dev(a.C)), c = idx - t * a.C;
      const float* base = tile + (size_t)t * phw * a.Cs + in0 + c;
      float l = base[0], h = l;
      for (int y = 0; y < a.H; ++y)
        for (int x = 0; x < a.W; ++x) {
          const float v = base[(y * a.PW + x) * a.Cs];
          l = fminf(l, v);
          h = fmaxf(h, v);
        }
      float sp = h - l;
      if (sp < 1e-5f) sp += 1e-5f;
      lo[idx] = l;
      sc[idx] = sp;
    }
    __syncthreads();
    const int CHW = a.C * HW, total = a.T * CHW;
    for (int i = tid; i < total; i += RB_THREADS) {
      const int t = rb_div(i, CHW, a.magic_chw), r = i - t * CHW;
      const int c = rb_div(r, HW, a.magic_hw), p = r - c * HW;
      const int y = rb_div(p, a.W, a.magic_w), x = p - y * a.W;
      const int b = b0 + t;
      if (b >= a.batch) continue;
      const float v = tile[(size_t)t * phw * a.Cs + in0 + (y * a.PW + x) * a.Cs + c];
      float* yb = a.scale_y + ((long long)b * a.scale_nodes + (a.scale_node ? a.scale_node[b] : 0)) * a.scale_sstride;
      yb[r] = mzx_div(v - lo[t * a.C + c], sc[t * a.C + c]);
    }
  }
  for (int q = 0; q < a.n_conv; ++q) {
    // y[b][rc][p] = bias[rc] + sum_c x[b][p][c] w[rc][c], channels in order (one fmaf chain per output)
    const int R = a.conv[q].R, RHW = R * HW, total = a.T * RHW;
    const uint32_t magic_rhw = rb_magic_dev(RHW);
    for (int i = tid; i < total; i += RB_THREADS) {
      const int t = rb_div(i, RHW, magic_rhw), r = i - t * RHW;
      const int rc = rb_div(r, HW, a.magic_hw), p = r - rc * HW;
      const int y = rb_div(p, a.W, a.magic_w), x = p - y * a.W;
      const int b = b0 + t;
      if (b >= a.batch) continue;
      const float* xin = tile + (size_t)t * phw * a.Cs + in0 + (y * a.PW + x) * a.Cs;
      const float* w = a.conv[q].w + (size_t)rc * a.C;
      float acc = 0.f;
      int c = 0;
      for (; c + 3 < a.C; c += 4) {
        const f32x4 v = *(const f32x4*)(xin + c);
        acc = fmaf(v[0], w[c], acc);
        acc = fmaf(v[1], w[c + 1], acc);
        acc = fmaf(v[2], w[c + 2], acc);
        acc = fmaf(v[3], w[c + 3], acc);
      }
      for (; c < a.C; ++c) acc = fmaf(xin[c], w[c], acc);
      a.conv[q].y[(long long)b * RHW + r] = acc + a.conv[q].b[rc];
    }
  }
}

// Per-plane min-max scaling of the hidden state (models.py:527-553, :574-599; MinMaxScaleOp's arithmetic): one
// workgroup per sample, NHWC or NCHW in, NCHW out (into the search arena's node store when `node` is set).
struct RbScaleArgs {
  RbTensor x;
  float* y;
  const int32_t* y_node;
  int64_t y_sstride;
  int32_t y_nodes, C, HW;
};

__global__ void __launch_bounds__(256) rb_scale_kernel(const RbScaleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rb_lds[];
  float* lo = rb_lds;
  float* sc = lo + a.C;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = a.x.p + ((long long)b * a.x.nodes + (a.x.node ? a.x.node[b] : 0)) * a.x.sstride;
  float* yb = a.y + ((long long)b * a.y_nodes + (a.y_node ? a.y_node[b] : 0)) * a.y_sstride;
  const bool nhwc = a.x.layout == RB_NHWC;
  for (int c = tid; c < a.C; c += 256) {
    float l = nhwc ? xb[c] : xb[(long long)c * a.HW], h = l;
    for (int p = 1; p < a.HW; ++p) {
      const float v = nhwc ? xb[(long long)p * a.C + c] : xb[(long long)c * a.HW + p];
      l = fminf(l, v);
      h = fmaxf(h, v);
    }
    float s = h - l;
    if (s < 1e-5f) s += 1e-5f;
    lo[c] = l;
    sc[c] = s;
  }
  __syncthreads();
  const int total = a.C * a.HW;
  for (int i = tid; i < total; i += 256) {
    const int c = i / a.HW, p = i - c * a.HW;
    const float v = nhwc ? xb[(long long)p * a.C + c] : xb[i];
    yb[i] = mzx_div(v - lo[c], sc[c]);
  }
}

// AvgPool2d(kernel 3, stride 2, padding 1), count_include_pad=True (AvgPoolOp), position-major tensors.
struct RbPoolOp {
  const float* x;
  float* y;
  int32_t batch, C, hin, win, hout, wout, out_nchw;
  MZX_HD size_t size() const { return (size_t)batch * hout * wout * C; }
  MZX_HD void operator()(size_t i) const {
    const int c = (int)(i % C);
    const int ox = (int)((i / C) % wout), oy = (int)((i / ((size_t)C * wout)) % hout);
    const int64_t b = (int64_t)(i / ((size_t)C * wout * hout));
    const float* xb = x + b * hin * win * C + c;
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 + ky - 1;
      if (iy < 0 || iy >= hin) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 + kx - 1;
        if (ix < 0 || ix >= win) continue;
        acc += xb[((int64_t)iy * win + ix) * C];
      }
    }
    const float v = acc / 9.0f;
    if (out_nchw) y[((b * C + c) * hout + oy) * wout + ox] = v;
    else y[i] = v;
  }
};

// torch.nn.MaxPool2d(kernel_size=3, stride=2) (models.py:286, :289: no padding, floor mode), position-major tensors.
struct RbMaxPoolOp {
  const float* x;
  float* y;
  int32_t batch, C, hin, win, hout, wout;
  MZX_HD size_t size() const { return (size_t)batch * hout * wout * C; }
  MZX_HD void operator()(size_t i) const {
    const int c = (int)(i % C);
    const int ox = (int)((i / C) % wout), oy = (int)((i / ((size_t)C * wout)) % hout);
    const int64_t b = (int64_t)(i / ((size_t)C * wout * hout));
    const float* xb = x + b * hin * win * C + c;
    float m = xb[((int64_t)(oy * 2) * win + ox * 2) * C];
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) m = fmaxf(m, xb[((int64_t)(oy * 2 + ky) * win + ox * 2 + kx) * C]);
    y[i] = m;
  }
};

// torch.nn.AdaptiveAvgPool2d((hout, wout)) (models.py:291; AdaptiveAvgPoolOp's windows and order), position-major.
struct RbAdaptivePoolOp {
  const float* x;
  float* y;
  int32_t batch, C, hin, win, hout, wout;
  MZX_HD size_t size() const { return (size_t)batch * hout * wout * C; }
  MZX_HD void operator()(size_t i) const {
    const int c = (int)(i % C);
    const int ox = (int)((i / C) % wout), oy = (int)((i / ((size_t)C * wout)) % hout);
    const int64_t b = (int64_t)(i / ((size_t)C * wout * hout));
    const float* xb = x + b * hin * win * C + c;
    const int y0 = (oy * hin) / hout, y1 = ((oy + 1) * hin + hout - 1) / hout;
    const int x0 = (ox * win) / wout, x1 = ((ox + 1) * win + wout - 1) / wout;
    float acc = 0.f;
    for (int iy = y0; iy < y1; ++iy)
      for (int ix = x0; ix < x1; ++ix) acc += xb[((int64_t)iy * win + ix) * C];
    y[i] = acc / (float)((y1 - y0) * (x1 - x0));
  }
};

// [batch][HW][C] -> [batch][C][HW] (diagnostic dumps: the per-operator kernels' layout)
struct RbToNchwOp {
  const float* x;
  float* y;
  int32_t batch, C, HW;
  MZX_HD size_t size() const { return (size_t)batch * C * HW; }
  MZX_HD void operator()(size_t i) const {
    const int p = (int)(i % HW), c = (int)((i / HW) % C);
    const int64_t b = (int64_t)(i / ((size_t)HW * C));
    y[i] = x[(b * HW + p) * C + c];
  }
};

uint32_t rb_magic(int d) { return d > 1 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u; }

typedef void (*RbGemmFn)(const RbGemmArgs);

template <int NT>
RbGemmFn rb_pick_mt(int mt) {
  switch (mt) {
    case 1: return rb_gemm_kernel<1, NT>;
    case 2: return rb_gemm_kernel<2, NT>;
    case 3: return rb_gemm_kernel<3, NT>;
    case 4: return rb_gemm_kernel<4, NT>;
    case 5: return rb_gemm_kernel<5, NT>;
    case 6: return rb_gemm_kernel<6, NT>;
    case 7: return rb_gemm_kernel<7, NT>;
    case 8: return rb_gemm_kernel<8, NT>;
    default: return rb_gemm_kernel<9, NT>;
  }
}

// A launch shape of operator `o`, written into its arguments.
void rb_apply_shape(RbGemmArgs& a, const RbOp& o, const RbShape& sh) {
  const int WN = sh.WN, WM = sh.WM;
  a.T = sh.T; a.rows = sh.rows; a.mtiles = sh.mtiles;
  a.cpg = sh.cpg; a.phases = sh.phases; a.Cs = sh.Cs;
  a.kmagic = (65536 + o.ksize - 1) / o.ksize; a.rowskip = (o.PW - o.ksize) * sh.Cs;   // the launch's Cs, not the plan's
  a.rowsplit = sh.rowsplit;
  if (sh.rowsplit > 0) {   // row-range tiles: tiles_y workgroups per sample, patch = the board rows a range touches
    a.tiles_x = 1; a.tiles_y = sh.splits; a.PH = sh.PH;
    a.magic_tw = rb_magic(o.wout); a.magic_phw = rb_magic(sh.PH * o.PW);
  }
  a.ntiles_wg = sh.ntiles_wg; a.WN = WN; a.WM = WM;
  static const int dbg = exp_int("MZX_RB_DBG", 0);
  a.dbg = dbg;
  a.grid_x = sh.groups; a.grid_y = sh.nsplit;
}

// The launch shape of operator `o` at `batch` (rb_choose_shape), written into its arguments.
RbShape rb_shape_args(RbGemmArgs& a, const RbOp& o, int batch) {
  const RbShape sh = rb_choose_shape(o, batch);
  rb_apply_shape(a, o, sh);
  return sh;
}

int rb_launch_gemm(RbGemmArgs& a, const RbOp& o, int batch, stream_t stream) {
  const RbShape sh = rb_shape_args(a, o, batch);
  const int groups_m = sh.groups, nsplit = sh.nsplit, NT = sh.NT, WN = sh.WN, WM = sh.WM, MT = sh.MT;
  // MZX_RB_STAMPS=<launch number>: phase clocks of that GEMM launch (counted from the first one of the process),
  // printed to stderr after a blocking copy -- a diagnostic, never set in production
  static const int stamp_launch = exp_int("MZX_RB_STAMPS", -1);
  static std::atomic<int> launch_no{0};      // (two streams / host threads launch through here)
  static unsigned long long* d_stamps = nullptr;   // touched by the one stamped launch only
  const bool stamp = stamp_launch >= 0 && launch_no.fetch_add(1) == stamp_launch;
  const size_t stamp_words = (size_t)RB_STAMP_WGS * 8 * RB_STAMP_SLOTS;
  if (stamp) {
    if (!d_stamps && hipMalloc((void**)&d_stamps, stamp_words * 8) != hipSuccess) d_stamps = nullptr;
    if (d_stamps) (void)hipMemsetAsync(d_stamps, 0, stamp_words * 8, stream);
    a.stamps = d_stamps;
  }
  RbGemmFn fn = NT == 2 ? rb_pick_mt<2>(MT) : rb_pick_mt<1>(MT);
  static std::atomic<uint64_t> lds_attr_done[2][RB_MT + 1];   // per instantiation, one bit per device
  if (const int ae = allow_large_lds((const void*)fn, RB_LDS_MAX, lds_attr_done[NT - 1][MT])) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  hipError_t e;
  void* params[] = {(void*)&a};
  e = hipLaunchKernel((const void*)fn, dim3(groups_m, nsplit), dim3(RB_THREADS), params, (size_t)sh.lds, stream);
  if (e != hipSuccess) { set_error("streamed GEMM launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  if (stamp && d_stamps) {
    std::vector<unsigned long long> h(stamp_words);
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h.data(), d_stamps, stamp_words * 8, hipMemcpyDeviceToHost);
    const int slots = 2 + 2 * sh.phases;
    fprintf(stderr, "rb stamps: launch %d, MT %d NT %d WN %d WM %d, grid %d x %d, lds %d, phases %d, rows %d\n", stamp_launch, MT, NT,
            WN, WM, groups_m, nsplit, sh.lds, sh.phases, sh.rows);
    for (int wg = 0; wg < std::min(groups_m, 8); ++wg)
      for (int w = 0; w < 8; w += 4) {
        const unsigned long long* r = h.data() + ((size_t)wg * 8 + w) * RB_STAMP_SLOTS;
        fprintf(stderr, "  wg %d wave %d: start %llu", wg, w, r[0] - h[0]);
        for (int k = 1; k < slots; ++k) fprintf(stderr, " +%llu", r[k] - r[k - 1]);
        fprintf(stderr, "   (tables+stage0, K0, stage1, K1, ..., epilogue)\n");
      }
    // mean over the stamped workgroups, wave 0
    std::vector<double> mean(slots, 0.0);
    const int nw = std::min(groups_m, RB_STAMP_WGS);
    for (int wg = 0; wg < nw; ++wg) {
      const unsigned long long* r = h.data() + (size_t)wg * 8 * RB_STAMP_SLOTS;
      for (int k = 1; k < slots; ++k) mean[k] += (double)(r[k] - r[k - 1]) / nw;
    }
    fprintf(stderr, "  mean over %d workgroups (wave 0):", nw);
    for (int k = 1; k < slots; ++k) fprintf(stderr, " %.0f", mean[k]);
    fprintf(stderr, "\n");
    {   // when workgroups start and end on the 100 MHz reference clock (comparable across XCDs), every wave 0 / wave 4
      std::vector<double> t0, t1;
      const int nall = std::min(groups_m, RB_STAMP_WGS);
      for (int wg = 0; wg < nall; ++wg)
        for (int w = 0; w < 8; w += 4) {
          const unsigned long long* r = h.data() + ((size_t)wg * 8 + w) * RB_STAMP_SLOTS;
          if (r[RB_STAMP_SLOTS - 1] > r[RB_STAMP_SLOTS - 2]) { t0.push_back((double)r[RB_STAMP_SLOTS - 2]); t1.push_back((double)r[RB_STAMP_SLOTS - 1]); }
        }
      if (!t0.empty()) {
        const double base = *std::min_element(t0.begin(), t0.end());
        std::vector<double> start(t0), end(t1), dur(t0.size());
        for (size_t i = 0; i < t0.size(); ++i) { start[i] = (t0[i] - base) / 100.0; end[i] = (t1[i] - base) / 100.0; dur[i] = end[i] - start[i]; }
        auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        fprintf(stderr, "  %zu waves of %d workgroups: start (us) p50 %.1f p99 %.1f max %.1f | duration p1 %.1f p50 %.1f p99 %.1f max %.1f | "
                        "end p1 %.1f p50 %.1f max %.1f\n", t0.size(), nall, pct(start, .5), pct(start, .99), pct(start, 1.0), pct(dur, .01),
                pct(dur, .5), pct(dur, .99), pct(dur, 1.0), pct(end, .01), pct(end, .5), pct(end, 1.0));
      }
    }
    {   // effective shader clock: shader cycles (s_memtime) per tick of the 100 MHz reference (s_memrealtime), wave 4 of workgroup 0
      const unsigned long long* r = h.data() + (size_t)4 * RB_STAMP_SLOTS;
      const double cycles = (double)(r[slots - 1] - r[0]), ticks = (double)(r[RB_STAMP_SLOTS - 1] - r[RB_STAMP_SLOTS - 2]);
      if (ticks > 0) fprintf(stderr, "  workgroup 0 wave 4: %.0f shader cycles in %.2f us -> %.0f MHz effective shader clock\n", cycles,
                             ticks / 100.0, cycles / ticks * 100.0);
    }
  }
  return MZX_OK;
}

// n operators of one (MT, NT) class in ONE launch (rb_gemm_multi_kernel: blockIdx.z = operator).  The arguments carry
// their own shapes (rb_shape_args); returns MZX_ERR_INVALID without launching when the class has no multi instantiation.
typedef void (*RbGemmMultiFn)(const RbGemmMulti);
inline RbGemmMultiFn rb_pick_multi(int mt, int nt) {
  if (nt != 1) return nullptr;
  switch (mt) {
    case 1: return rb_gemm_multi_kernel<1, 1>;
    case 2: return rb_gemm_multi_kernel<2, 1>;
    case 3: return rb_gemm_multi_kernel<3, 1>;
    case 4: return rb_gemm_multi_kernel<4, 1>;
    default: return nullptr;
  }
}

int rb_launch_gemm_multi(const RbGemmMulti& m, const RbShape* sh, int n, stream_t stream) {
  RbGemmMultiFn fn = rb_pick_multi(sh[0].MT, sh[0].NT);
  if (!fn || n < 1 || n > RB_MULTI_MAX) { set_error("grouped GEMM launch: no instantiation"); return MZX_ERR_INVALID; }
  int gx = 0, gy = 0, lds = 0;
  for (int q = 0; q < n; ++q) { gx = std::max(gx, sh[q].groups); gy = std::max(gy, sh[q].nsplit); lds = std::max(lds, sh[q].lds); }
  static std::atomic<uint64_t> lds_attr_done[RB_MULTI_MT + 1];
  if (const int ae = allow_large_lds((const void*)fn, RB_LDS_MAX, lds_attr_done[sh[0].MT])) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  void* params[] = {(void*)&m};
  const hipError_t e = hipLaunchKernel((const void*)fn, dim3(gx, gy, n), dim3(RB_THREADS), params, (size_t)lds, stream);
  if (e != hipSuccess) { set_error("grouped GEMM launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

typedef void (*RbTowerFn)(const RbTowerArgs);

template <int NT>
RbTowerFn rb_pick_tower_mt(int mt) {
  switch (mt) {
    case 1: return rb_tower_kernel<1, NT>;
    case 2: return rb_tower_kernel<2, NT>;
    case 3: return rb_tower_kernel<3, NT>;
    case 4: return rb_tower_kernel<4, NT>;
    case 5: return rb_tower_kernel<5, NT>;
    case 6: return rb_tower_kernel<6, NT>;
    default: break;
  }
  // (two column tiles per wave: the planner stops at six row tiles, rb_tower_grid -- 7 .. 9 would not fit 256 registers with
  // the saved residual beside the accumulators and are not instantiated)
  if constexpr (NT == 1) {
    switch (mt) {
      case 7: return rb_tower_kernel<7, NT>;
      case 8: return rb_tower_kernel<8, NT>;
      case 9: return rb_tower_kernel<9, NT>;
      default: break;
    }
  }
  return nullptr;
}

// Launches layers [tw.first, tw.first + layers) of a tower (layers < tw.count: a diagnostic prefix; the last layer run
// writes its output to memory like the whole tower's last layer does).
struct RbTowerTail {
  int write_out = 1;
  RbTensor scale{};          // .p null: no scaling operator in the tail
  int n_conv = 0;
  struct { const float* w; const float* b; float* y; int R; } conv[2] = {};
};

int rb_launch_tower(const mzx_net* net, const std::vector<OpDesc>& prog, const RbProgram& R, const RbTower& tw, int layers,
                    const RbTensor& x, float* y, const int32_t* action, int batch, stream_t stream, const RbTowerTail& tail) {
  const RbTowerShape sh = rb_tower_shape(tw, batch);
  if (sh.T < 1) { set_error("tower: no launch shape"); return MZX_ERR_INVALID; }
  RbTowerArgs a;
  memset(&a, 0, sizeof(a));
  a.write_out = tail.write_out;
  a.scale_y = const_cast<float*>(tail.scale.p); a.scale_node = tail.scale.node; a.scale_sstride = tail.scale.sstride;
  a.scale_nodes = tail.scale.nodes;
  a.n_conv = tail.n_conv;
  for (int q = 0; q < tail.n_conv; ++q) { a.conv[q].w = tail.conv[q].w; a.conv[q].b = tail.conv[q].b; a.conv[q].y = tail.conv[q].y; a.conv[q].R = tail.conv[q].R; }
  a.magic_chw = rb_magic(tw.C * tw.H * tw.W);
  const RbOp& o0 = R.ops[tw.first];
  a.x = x; a.y = y; a.y_sstride = (int64_t)tw.C * tw.H * tw.W;
  a.der = net->d_derived;
  if (o0.asum_off >= 0) { a.asum = net->d_derived + o0.asum_off; a.action = action; a.num_actions = net->cfg.action_space_size; }
  a.batch = batch; a.cin0 = o0.cin; a.C = tw.C; a.H = tw.H; a.W = tw.W; a.PH = tw.H + 2; a.PW = tw.W + 2;
  a.Cs = sh.Cs; a.cchunks = tw.cchunks; a.T = sh.T; a.rows = sh.rows; a.mtiles = sh.mtiles; a.ntiles = tw.ntiles;
  a.WN = sh.WN; a.WM = sh.WM; a.layers = layers;
  static const int dbg = exp_int("MZX_RB_DBG", 0);
  a.dbg = dbg;
  a.y_vec = (tw.C % 4 == 0 && a.y_sstride % 4 == 0 && ((uintptr_t)y % 16) == 0) ? 1 : 0;
  a.rowskip = (a.PW - 3) * sh.Cs;
  a.magic_hw = rb_magic(tw.H * tw.W); a.magic_w = rb_magic(tw.W);
  a.magic_phw = rb_magic(a.PH * a.PW); a.magic_pw = rb_magic(a.PW);
  for (int l = 0; l < layers; ++l) {
    const OpDesc& d = prog[tw.first + l];
    const RbOp& o = R.ops[tw.first + l];
    RbTowerLayer& L = a.layer[l];
    L.w_off = o.w_off;
    L.bn_alpha = d.bn.channels ? d.bn.alpha : -1;
    L.bn_beta = d.bn.channels ? d.bn.beta : -1;
    L.cchunks = o.cchunks;
    L.flags = (o.act == RZ_ACT_RELU ? 1 : 0) | (d.res != -100 ? 4 : 0);
    // the NEXT layer adds this layer's input (the block input): keep it when this layer's output replaces it
    if (l + 1 < tw.count && prog[tw.first + l + 1].res != -100) L.flags |= 2;
  }
  RbTowerFn fn = sh.NT == 2 ? rb_pick_tower_mt<2>(sh.MT) : rb_pick_tower_mt<1>(sh.MT);
  if (!fn) { set_error("tower: no instantiation for %d x %d tiles per wave", sh.MT, sh.NT); return MZX_ERR_INVALID; }
  static std::atomic<uint64_t> lds_attr_done[2][RB_MT + 1];
  if (const int ae = allow_large_lds((const void*)fn, RB_LDS_MAX, lds_attr_done[sh.NT - 1][sh.MT])) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  void* params[] = {(void*)&a};
  const hipError_t e = hipLaunchKernel((const void*)fn, dim3(sh.groups), dim3(RB_THREADS), params, (size_t)sh.lds, stream);
  if (e != hipSuccess) { set_error("tower launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

}  // namespace

int rb_refresh_derived(const mzx_net* net, const float* d_flat, float* d_derived, stream_t stream) {
  for (const RzPack& p : net->rb.packs) {
    RzPackOp op;
    op.W = d_flat + p.src; op.out = d_derived + p.dst;
    op.taps = p.taps; op.cin = p.cin; op.cin_total = p.cin_total; op.cchunks = p.cchunks; op.cout = p.cout;
    op.nchunks = p.nchunks; op.wchunks = p.wchunks; op.ntiles = p.ntiles;
    if (int rc = launch<256>(op, stream)) { set_error("launch failed: %s", runtime_error_string(rc)); return MZX_ERR_RUNTIME; }
  }
  for (const RzAsum& q : net->rb.asums) {
    RzAsumOp op;
    op.W = d_flat + q.src; op.out = d_derived + q.dst; op.cout = q.cout; op.cin_total = q.cin_total; op.H = q.H; op.Wd = q.W;
    if (int rc = launch<256>(op, stream)) { set_error("launch failed: %s", runtime_error_string(rc)); return MZX_ERR_RUNTIME; }
  }
  return MZX_OK;
}

int rb_run_program(const mzx_net* net, bool recurrent, const NetBuffers& nb, int batch, stream_t stream,
                   const NetIndex* ix, int n_ops, float* dump, bool last_nchw) {
  const std::vector<OpDesc>& prog = recurrent ? net->prog_recurrent : net->prog_initial;
  const RbProgram& R = recurrent ? net->rb.recurrent : net->rb.initial;
  const float* flat = net->d_flat;
  const float* der = net->d_derived;
  const int count = n_ops < 0 ? (int)prog.size() : n_ops;
  const int64_t in_floats = recurrent ? net->hidden_size : net->input_size;
  // logical buffer -> tensor (per-sample stride `dense` for workspace temporaries and head outputs)
  auto tensor = [&](int id, int64_t dense, int layout) {
    RbTensor t;
    t.p = resolve(net, nb, id, batch); t.node = nullptr; t.nodes = 1; t.sstride = dense; t.layout = layout;
    if (id == BUF_IN) { t.sstride = in_floats; if (ix) { t.node = ix->in_node; t.nodes = ix->in_nodes; } }
    if (id == BUF_HIDDEN) { t.sstride = net->hidden_size; if (ix) { t.node = ix->out_node; t.nodes = ix->out_nodes; } }
    return t;
  };
  // ---- head chains whose levels run as grouped launches at the end of the program: the whole program runs, their
  // tower runs as a tower with its tail at this batch (the tail convolution then writes the chain's input into the
  // private region behind the temporaries, where nothing can overwrite it)
  const bool tails_on = tune(TUNE_RB_TAIL) != 0;
  bool chain_on[RB_HEADS_MAX_CHAINS] = {false, false, false};
  int chains_on = 0;
  float* head_region = nb.workspace + net->act_floats * net->n_temp * (int64_t)batch;
  // tuning "rb_heads": 0 one launch per Linear layer; 2 ONE MFMA launch per chain LEVEL (rb_gemm_multi_kernel: the k-th
  // layers of all chains) -- the same body, shapes and channel groups, hence the same bits
  const int heads_mode = rb_heads_mode();
  if (count == (int)prog.size() && tails_on && !net->rb_no_towers && heads_mode == 2)
    for (int q = 0; q < R.heads.n_chains; ++q) {
      const RbHeadChain& hc = R.heads.chain[q];
      const int t = R.ops[hc.conv_op].tower_of_tail;
      if (t >= 0 && rb_tower_use(R.towers[t], batch)) { chain_on[q] = true; ++chains_on; }
    }
  // the arguments of GEMM operator k (x_private / y_private: a head chain's input / inner output in the private region
  // behind the temporaries instead of the program's buffers -- same strides)
  auto gemm_args = [&](int k, const float* x_private, float* y_private) {
    const OpDesc& d = prog[k];
    const RbOp& o = R.ops[k];
    RbGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = tensor(d.in, (int64_t)o.cin * o.hin * o.win, o.in_layout);
    if (x_private) { a.x.p = x_private; a.x.node = nullptr; a.x.nodes = 1; }
    if (d.res != -100) a.res = tensor(d.res, (int64_t)o.cout * o.hout * o.wout, o.res_layout);
    a.y = y_private ? y_private : resolve(net, nb, d.out, batch);
    a.y_sstride = (int64_t)o.cout * o.hout * o.wout;
    a.y_layout = o.out_layout;
    a.wpack = der + o.w_off;
    if (d.kind == OP_CONV3 && d.bn.channels) { a.alpha = der + d.bn.alpha; a.beta = der + d.bn.beta; }
    if (d.kind != OP_CONV3) a.bias = flat + d.b;   // 1x1 heads, Linear, DownsampleCNN convolutions
    if (o.asum_off >= 0) { a.asum = der + o.asum_off; a.action = nb.action; a.num_actions = net->cfg.action_space_size; }
    a.batch = batch;
    a.cin = o.cin; a.cout = o.cout; a.hin = o.hin; a.win = o.win; a.hout = o.hout; a.wout = o.wout;
    a.stride = o.stride; a.taps = o.taps; a.ksize = o.ksize; a.pad = o.pad;
    a.T = o.T; a.th = o.th; a.tw = o.tw; a.tiles_x = o.tiles_x; a.tiles_y = o.tiles_y; a.PH = o.PH; a.PW = o.PW;
    a.Cs = o.Cs; a.cpg = o.cpg; a.phases = o.phases; a.cchunks = o.cchunks; a.wchunks = o.wchunks;
    a.rows = o.rows; a.mtiles = o.mtiles; a.ntiles = o.ntiles; a.act = o.act;
    a.x_vec = (o.in_layout == RB_NHWC && o.cin % 4 == 0 && a.x.sstride % 4 == 0 && ((uintptr_t)a.x.p % 16) == 0) ? 1 : 0;
    a.y_vec = (o.out_layout == RB_NHWC && o.cout % 4 == 0 && a.y_sstride % 4 == 0 && ((uintptr_t)a.y % 16) == 0) ? 1 : 0;
    a.r_vec = (a.res.p && o.res_layout == RB_NHWC && o.cout % 4 == 0 && a.res.sstride % 4 == 0 && ((uintptr_t)a.res.p % 16) == 0) ? 1 : 0;
    a.magic_thw = rb_magic(o.th * o.tw); a.magic_tw = rb_magic(o.tw);
    a.magic_phw = rb_magic(o.PH * o.PW); a.magic_pw = rb_magic(o.PW);
    return a;
  };
  for (int k = 0; k < count; ++k) {
    const OpDesc& d = prog[k];
    const RbOp& o = R.ops[k];
    int rc = 0;
    if (o.head_chain >= 0 && chain_on[o.head_chain] && d.kind == OP_LINEAR) continue;     // runs in rb_heads_kernel below
    if (o.kind == RB_GEMM && o.tower >= 0 && !net->rb_no_towers && R.towers[o.tower].first == k &&
        rb_tower_use(R.towers[o.tower], batch)) {
      // a whole tower (or, for a diagnostic prefix, its first layers) in one launch; inner outputs never reach memory
      const RbTower& tw = R.towers[o.tower];
      const int layers = std::min(tw.count, count - k);
      const OpDesc& dl = prog[k + layers - 1];
      const RbTensor x = tensor(d.in, (int64_t)o.cin * o.hin * o.win, o.in_layout);
      // the tail: the scaling operator and the small 1x1 head convolutions that directly follow the whole tower and read
      // its output run inside the tower launch, on the LDS-resident tile (a diagnostic prefix takes as many as it covers)
      RbTowerTail tail;
      int n_tail = 0;
      if (layers == tw.count && tails_on) {      // (tuning "rb_tail" = 0: towers end at their last convolution, the A/B)
        for (int m = k + layers; m < count && m < k + layers + tw.n_tail; ++m) {
          const OpDesc& dm = prog[m];
          if (dm.kind == OP_SCALE) {
            tail.scale = tensor(dm.out, (int64_t)dm.groups_per_sample * dm.len, RB_NCHW);
          } else {
            auto& cv = tail.conv[tail.n_conv++];
            cv.w = flat + dm.w; cv.b = flat + dm.b; cv.y = resolve(net, nb, dm.out, batch); cv.R = dm.cout;
            const int hq = R.ops[m].head_chain;
            if (hq >= 0 && chain_on[hq]) {      // the chain's input: [batch][in_features] in the private region
              cv.y = head_region + R.heads.chain[hq].in_off * (int64_t)batch;
            }
          }
          ++n_tail;
        }
        // does anything else still read the tower's output?
        tail.write_out = 0;
        for (int m = k + layers + n_tail; m < (int)prog.size(); ++m) {
          if (prog[m].in == dl.out || prog[m].res == dl.out) { tail.write_out = 1; break; }
          if (prog[m].out == dl.out) break;
        }
        if (n_tail < tw.n_tail || n_tail == 0) tail.write_out = 1;   // (a prefix that ends inside the tail: the rest would read it)
      }
      rc = rb_launch_tower(net, prog, R, tw, layers, x, resolve(net, nb, dl.out, batch), nb.action, batch, stream, tail);
      if (rc) return rc;
      k += layers + n_tail - 1;
      continue;
    }
    if (o.kind == RB_GEMM) {
      RbGemmArgs a = gemm_args(k, nullptr, nullptr);
      rc = rb_launch_gemm(a, o, batch, stream);
      if (rc) return rc;
      continue;
    }
    if (o.kind == RB_SCALE) {
      RbScaleArgs s;
      s.x = tensor(d.in, (int64_t)d.groups_per_sample * d.len, o.in_layout);
      const RbTensor y = tensor(d.out, (int64_t)d.groups_per_sample * d.len, RB_NCHW);
      s.y = const_cast<float*>(y.p); s.y_node = y.node; s.y_nodes = y.nodes; s.y_sstride = y.sstride;
      s.C = d.groups_per_sample; s.HW = d.len;
      hipLaunchKernelGGL(rb_scale_kernel, dim3(batch), dim3(256), (size_t)(2 * s.C * sizeof(float)), stream, s);
      rc = (int)hipGetLastError();
    } else if (o.kind == RB_POOL && o.in_layout == RB_NHWC) {
      RbPoolOp op;
      op.x = resolve(net, nb, d.in, batch); op.y = resolve(net, nb, d.out, batch);
      op.batch = batch; op.C = d.cin; op.hin = d.hin; op.win = d.win; op.hout = d.hout; op.wout = d.wout;
      op.out_nchw = (last_nchw && k == count - 1) ? 1 : 0;
      rc = launch<256>(op, stream);
    } else if (o.kind == RB_MAXPOOL) {
      RbMaxPoolOp op;
      op.x = resolve(net, nb, d.in, batch); op.y = resolve(net, nb, d.out, batch);
      op.batch = batch; op.C = d.cin; op.hin = d.hin; op.win = d.win; op.hout = d.hout; op.wout = d.wout;
      rc = launch<256>(op, stream);
    } else if (o.kind == RB_ADAPTIVE_POOL) {
      RbAdaptivePoolOp op;
      op.x = resolve(net, nb, d.in, batch); op.y = resolve(net, nb, d.out, batch);
      op.batch = batch; op.C = d.cin; op.hin = d.hin; op.win = d.win; op.hout = d.hout; op.wout = d.wout;
      rc = launch<256>(op, stream);
    } else {
      if (ix && ((d.in == BUF_IN && ix->in_nodes != 1) ||
                 ((d.in == BUF_HIDDEN || d.out == BUF_HIDDEN) && ix->out_nodes != 1))) {
        set_error("operator %d of the streamed engine does not take indexed hidden states", k);
        return MZX_ERR_INVALID;
      }
      const std::vector<OpDesc> one(prog.begin() + k, prog.begin() + k + 1);
      const int r2 = run_program(net, one, nb, batch, stream);
      if (r2) return r2;
    }
    if (rc) { set_error("kernel launch failed: %s", runtime_error_string(rc)); return MZX_ERR_RUNTIME; }
  }
  if (chains_on > 0) {
    for (int level = 0; level < RB_HEADS_MAX_LAYERS; ++level) {
      RbGemmArgs args[RB_HEADS_MAX_CHAINS];
      RbShape shapes[RB_HEADS_MAX_CHAINS];
      int ops[RB_HEADS_MAX_CHAINS], n = 0;
      for (int q = 0; q < R.heads.n_chains; ++q) {
        const RbHeadChain& hc = R.heads.chain[q];
        if (!chain_on[q] || level >= hc.count) continue;
        const float* x = head_region + (level == 0 ? hc.in_off : hc.hid_off[level - 1]) * (int64_t)batch;
        float* y = level + 1 < hc.count ? head_region + hc.hid_off[level] * (int64_t)batch : nullptr;
        ops[n] = hc.first + level;
        args[n] = gemm_args(ops[n], x, y);
        shapes[n] = rb_shape_args(args[n], R.ops[ops[n]], batch);
        ++n;
      }
      // one launch per (MT, NT) class of this level (every shipped configuration: one class); a class without a grouped
      // instantiation runs its operators one by one
      bool done[RB_HEADS_MAX_CHAINS] = {false, false, false};
      for (int q = 0; q < n; ++q) {
        if (done[q]) continue;
        RbGemmMulti m;
        memset(&m, 0, sizeof(m));
        RbShape sh[RB_MULTI_MAX];
        int member[RB_MULTI_MAX], members = 0;
        for (int z = q; z < n; ++z)
          if (!done[z] && shapes[z].MT == shapes[q].MT && shapes[z].NT == shapes[q].NT) {
            m.g[members] = args[z]; sh[members] = shapes[z]; member[members++] = z; done[z] = true;
          }
        int rc = MZX_OK;
        if (members > 1 && rb_pick_multi(sh[0].MT, sh[0].NT)) rc = rb_launch_gemm_multi(m, sh, members, stream);
        else
          for (int z = 0; z < members && !rc; ++z) rc = rb_launch_gemm(args[member[z]], R.ops[ops[member[z]]], batch, stream);
        if (rc) return rc;
      }
    }
  }
  if (dump && count > 0) {   // diagnostics: the last operator's output in the per-operator kernels' layout
    const OpDesc& d = prog[count - 1];
    const RbOp& o = R.ops[count - 1];
    const float* src = resolve(net, nb, d.out, batch);
    int rc = 0;
    const bool spatial = d.kind == OP_CONV3 || d.kind == OP_POOL || d.kind == OP_CONVK || d.kind == OP_MAXPOOL || d.kind == OP_ADAPTIVE_POOL;
    if (spatial && o.out_layout == RB_NHWC) {
      RbToNchwOp op;
      op.x = src; op.y = dump; op.batch = batch; op.C = d.cout; op.HW = d.hout * d.wout;
      rc = launch<256>(op, stream);
    } else {
      int64_t per = 0;
      switch (d.kind) {
        case OP_LINEAR: per = d.out_features; break;
        case OP_CONV1: per = (int64_t)d.cout * d.hin; break;
        case OP_SCALE: per = (int64_t)d.groups_per_sample * d.len; break;
        default: per = (int64_t)d.cout * d.hout * d.wout; break;
      }
      rc = copy_d2d(dump, src, sizeof(float) * per * batch, stream);
    }
    if (rc) { set_error("dump failed: %s", runtime_error_string(rc)); return MZX_ERR_RUNTIME; }
  }
  return MZX_OK;
}

}  // namespace mzx
