// mzx_tower_search.hip -- rt_search_kernel: EVERY simulation of MCTS.run (/root/reference/self_play.py:319-355) in ONE
// launch for wide residual networks whose trunks run as TOWERS (rb_tower_kernel, mzx_batched.hip) -- connect4-class
// networks (games/connect4.py: 64 channels x 3 residual blocks per trunk on 6 x 7).
//
// Why.  On the streamed engine a simulation of such a search is six DEPENDENT launches per half-shard (row select, two
// towers, two grouped head-MLP levels, row expand + back-propagate): 2 400 launches per 200-simulation step, every one
// with a launch gap, a tail of half-empty CUs and its own staging.  But nothing in a simulation crosses trees: a
// workgroup that owns T trees can run their whole search alone.  Here it does:
//   * the workgroup (512 threads) owns T trees for all simulations; tree t is walked by 16-lane row t with the row
//     functions of the per-simulation kernels (row_select_body / row_expand_backprop_body, mzx_row_search.h: binary64,
//     reference operation order, DPP arg-max + ballot, tape-exact tie draws) on the tree in the arena (L2-resident);
//   * recurrent_inference (models.py:620-623) runs inside: the selected parents' hidden states are staged from the
//     arena's node store into ONE LDS activation tile [cell][Cs] with a zero halo, the dynamics tower updates it in
//     place (mzx_tower_layers.inc -- the very layer loop of rb_tower_kernel), the reward 1x1 convolution and the
//     per-plane min-max scaling (models.py:574-599) read it there, the scaled state goes to the leaf's slot of the
//     node store AND stays in the tile as the prediction tower's input (it never makes the round trip through memory),
//     the value / policy 1x1 convolutions read the prediction tower's output there, and the head MLPs run on the
//     matrix pipes with one valid row per tree (the layer kernel's fragments and k order);
//   * no launch, no grid barrier, no inter-workgroup traffic: a step is root + ONE launch + result gathering.
// Two workgroups share a CU (connect4, two trees each: 52 KB of LDS, 128 registers): while one selects, stages,
// scales or back-propagates, the other feeds the matrix pipes.
//
// Same bits as the streamed route.  Every output element is produced by the same instructions on the same operands in
// the same order as on the tower route of mzx_batched.hip: the towers ARE mzx_tower_layers.inc, the tails restate
// rb_tower_kernel's tail statement by statement, a head layer is the MFMA chain of rb_gemm_kernel (weight fragment
// first, 16-input chunks in order, k = 0..3), the tree side is the row kernels' own functions.  Asserted on the device
// (tests/test_gpu_tower_search.py: every statistic of the finished trees bit-identical).
// Roofline: FP32 matrix pipe, 157.3 TFLOP/s dense.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <assert.h>

#include <atomic>
#include <memory>

#include "mzx_row_search.h"
#include "mzx_tower.h"
#include "mzx_tuning.h"

namespace mzx {

#ifndef MZX_HOSTCHECK
namespace {

constexpr int RT_MAX_LAYERS = 16;                       // layers per tower (connect4: 7 / 6, games/gomoku.py: 13 / 12)
constexpr int RT_MAX_CHAINS = RB_HEADS_MAX_CHAINS;      // reward, value, policy
constexpr int RT_MAX_LEVELS = RB_HEADS_MAX_LAYERS;
constexpr int RT_LDS_MAX = 160 * 1024;
constexpr int RT_MAX_ROWS = 256;                        // rows of a workgroup: eight row tiles per wave, two waves deep
constexpr int RT_MAX_T = 16;                            // trees per workgroup

// What mzx_tower_layers.inc reads of a tower (the field names of RbTowerArgs).
struct RtTower {
  const float* der;
  const float* asum;       // first layer: border-aware tap sums of the action plane (null: none)
  float* y;                // never written: the towers' outputs stay in the tile
  int64_t y_sstride;
  int32_t num_actions, C, Cs, PW, WN, WM, mtiles, ntiles, layers, rowskip, write_out, y_vec, dbg;
  RbTowerLayer layer[RT_MAX_LAYERS];
};

struct RtConv {            // a 1x1 head convolution in a tower's tail (conv1x1_reward / _value / _policy, models.py:369-433)
  int64_t w, b;            // flat buffer
  int32_t R;               // output channels
  int32_t out_lds;         // float offset (LDS) of its output rows [T][out_stride] = the input of its head chain
  int32_t out_stride, pad;
};

struct RtHeadLayer {       // a Linear (+ ELU) layer of a head chain (models.py:630-642)
  int64_t w_off;           // derived buffer: its packed B fragments (RzPackOp), [column tile][chunk][256]
  int64_t b_off;           // flat buffer: bias
  int32_t in_lds, out_lds; // float offsets (LDS) of the input rows / output rows
  int32_t in_stride, out_stride;
  int32_t cchunks, ntiles, out, elu;
};

struct RtSearchArgs {
  SearchParams p;          // pbc_table / sqrt_table: the handle's device tables (copied to LDS by the kernel)
  TreeLayout L;
  char* trees;             // arena: [num_trees][L.tree_bytes]
  const uint32_t* tape;    // [num_trees][tape_words]
  float* hidden;           // arena node store [num_trees][num_nodes][hidden_size], NCHW per node
  const float* flat;
  const float* der;
  int32_t num_sims, sim0, batch;
  int32_t dbg;             // timing experiments (tuning "rt_dbg", never set in production; results are WRONG with any bit set): 1 skip the
                           // K loops, 2 skip the epilogues, 4 skip select / expand, 8 skip staging and tails, 16 skip the head MLPs
  int32_t C, H, W, PH, PW, Cs, cchunks, T, rows, mtiles;
  uint32_t magic_hw, magic_w, magic_rows, magic_chw, magic_c;
  int32_t off_tables, off_rowtab, off_tile, off_scale, off_sel, off_rowsel, off_heads, heads_floats;   // LDS carve (bytes)
  int32_t value_lds, reward_lds, policy_lds, value_stride, reward_stride, policy_stride;              // final logits (floats)
  // Row r of the workgroup's MFMA tiles is (sample, position) perm[r] -- NOT raster order: rt_row_order deals the positions
  // so that the sixteen lanes a ds_read_b128 serves together read sixteen distinct 16-byte LDS slots (0x8000 | k: no
  // position, the row reads cell k); shift[t]: cells board t is moved by inside the tile (the same purpose).
  uint16_t perm[RT_MAX_ROWS];
  uint8_t shift[RT_MAX_T];
  int32_t tile_cells;      // cells of the tile: T boards with their halo + the last board's shift
  int32_t n_conv[2];       // tail convolutions of the dynamics / prediction tower: conv[0 .. n_conv[0]) / conv[n_conv[0] ..)
  int32_t n_chains, n_levels;
  int32_t levels[RT_MAX_CHAINS];
  RtConv conv[3];
  RtHeadLayer lin[RT_MAX_CHAINS][RT_MAX_LEVELS];
  RtTower tw[2];           // dynamics, prediction
};

// AW: lanes that can hold a child slot (4, 16), or 0 = wide (several slots per lane, any support size).
// NW: waves of the workgroup -- 8 (column tiles x two row groups for a 64-channel network) or 4 (one wave per column tile:
// every row tile of the workgroup on each wave).
template <int MT, int NT, int AW, int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(MT * NT <= 4 ? 4 : 2, MT * NT <= 4 ? 4 : 2)))
rt_search_kernel(const RtSearchArgs sa) {
  constexpr int RT_THREADS = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float rb_lds[];
  char* const lds = (char*)rb_lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = sa.T, mpad = sa.mtiles * 16, Tp = (T + 3) & ~3;
  double* const tables = (double*)(lds + sa.off_tables);
  int* const rowaddr = (int*)(lds + sa.off_rowtab);
  int* const rowt = rowaddr + mpad;
  int* const rowpos = rowt + mpad;
  float* const tile = (float*)(lds + sa.off_tile);
  float* const lo = (float*)(lds + sa.off_scale);
  float* const sc = lo + T * sa.C;
  int32_t* const sel_parent = (int32_t*)(lds + sa.off_sel);
  int32_t* const sel_action = sel_parent + Tp;
  int32_t* const sel_leaf = sel_action + Tp;
  int32_t* const rowsel = (int32_t*)(lds + sa.off_rowsel);
  float* const heads = (float*)(lds + sa.off_heads);
  const int b0 = blockIdx.x * T;
  const int ntree = min(T, sa.batch - b0);
  const int HW = sa.H * sa.W, phw = sa.PH * sa.PW, cells = sa.tile_cells;
  const int NN = sa.p.num_nodes;

  // ---- once per launch: the search's tables, the row tables, a ZERO tile (the halo and the channels beyond the
  // network's width are never written again), zero head rows (the padding of a chain's input up to whole 16-input chunks)
  for (int i = tid; i < 2 * (NN + 1); i += RT_THREADS) tables[i] = sa.p.pbc_table[i];   // pbc[N + 1] then sqrt[N + 1], contiguous
  for (int m = tid; m < mpad; m += RT_THREADS) {
    const unsigned pm = sa.perm[m];
    const bool valid = !(pm & 0x8000u);
    const int t = valid ? rb_div((int)pm, HW, sa.magic_hw) : 0, r = (int)pm - t * HW;
    const int y = rb_div(r, sa.W, sa.magic_w), x = r - y * sa.W;
    rowaddr[m] = valid ? ((t * sa.PH + y) * sa.PW + x + sa.shift[t]) * sa.Cs : (int)(pm & 7u) * sa.Cs;   // top-left cell of the 3 x 3 window
    rowt[m] = valid ? t : 0;
    rowpos[m] = (valid && b0 + t < sa.batch) ? r : -1;
  }
  for (int i = tid; i < cells * sa.Cs / 4; i += RT_THREADS) ((f32x4*)tile)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < sa.heads_floats; i += RT_THREADS) heads[i] = 0.f;
  if (tid < 3 * Tp) sel_parent[tid] = 0;
  SearchParams p = sa.p;
  p.pbc_table = tables;
  p.sqrt_table = tables + (NN + 1);

  // tree <-> 16-lane row
  const int sub = tid & (FUSED_ROW - 1), row = tid / FUSED_ROW, row_in_wave = row & 3;
  const bool row_valid = row < ntree;    // the other rows idle through the tree phases, every lane runs the network
  const int tree = b0 + (row_valid ? row : 0);
  TreeRef t;
  t.base = sa.trees + (size_t)tree * sa.L.tree_bytes;
  t.L = sa.L;
  const uint32_t* const tape = sa.tape + (size_t)tree * p.tape_words;
  int32_t* const rs = rowsel + (row_valid ? row : 0) * ROWSEL_INTS;
  const int ctr_cell = (sa.PW + 1) * sa.Cs;              // from a window's top-left cell to its centre = the position's own cell
  const int in0 = ctr_cell;                              // cell (0, 0) of a board inside its halo
  const int64_t Hf = (int64_t)sa.C * HW;
  __syncthreads();

  for (int sim = 0; sim < sa.num_sims; ++sim) {
    // ---- selection (self_play.py:325-334)
    if (row_valid && !(sa.dbg & 4))
      row_select_body<AW>(p, t, tape, sa.sim0 + sim, sub, row_in_wave, rs, sel_parent + row, sel_action + row, sel_leaf + row);
    __syncthreads();

    // ---- the selected parents' hidden states (node store, NCHW) -> the tile's interior cells.  Consecutive threads
    // take consecutive positions of one channel quad (coalesced along the board); rows of trees beyond the shard stay zero.
    if (!(sa.dbg & 8)) {
      const int q = sa.cchunks * 4;
      const int total = mpad * q;
      for (int i0 = tid; i0 < total; i0 += 2 * RT_THREADS) {
        f32x4 v[2];
        int at[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int idx = i0 + u * RT_THREADS;
          at[u] = -1;
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (idx < total) {
            const int k = rb_div(idx, mpad, sa.magic_rows), m = idx - k * mpad;
            const int pos = rowpos[m], c = 4 * k;
            if (pos >= 0) {
              const int tt = rowt[m];
              at[u] = rowaddr[m] + ctr_cell + c;
              if (c < sa.C) {
                const float* src = sa.hidden + ((int64_t)(b0 + tt) * NN + sel_parent[tt]) * Hf + (int64_t)c * HW + pos;
                v[u][0] = src[0];
                if (c + 1 < sa.C) v[u][1] = src[HW];
                if (c + 2 < sa.C) v[u][2] = src[2 * HW];
                if (c + 3 < sa.C) v[u][3] = src[3 * HW];
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (at[u] >= 0) *(f32x4*)(tile + at[u]) = v[u];
      }
    }

    // ---- dynamics tower (models.py:352-389): conv + BatchNorm + ReLU with the action plane, residual blocks, in place
    {
      const RtTower& a = sa.tw[0];
#define RB_TOWER_HAS_TAIL true
#define RB_TOWER_ACTION(b, t) sel_action[t]
#define RB_TOWER_LDS_BARRIER true
#include "mzx_tower_layers.inc"
#undef RB_TOWER_HAS_TAIL
#undef RB_TOWER_ACTION
#undef RB_TOWER_LDS_BARRIER
    }
    __syncthreads();                                     // the dynamics tower's output is in the tile's interior cells
    // ---- its tail, as rb_tower_kernel's: the reward head's 1x1 convolution and the per-plane (min, range) pairs read
    // the unscaled state; then the scaled state replaces it in the tile and goes to the leaf's slot of the node store
    for (int q = 0; q < ((sa.dbg & 8) ? 0 : sa.n_conv[0]); ++q) {
      // y[t][rc][p] = bias[rc] + sum_c x[t][p][c] w[rc][c], channels in order (one fmaf chain per output)
      const RtConv& cv = sa.conv[q];
      const int R = cv.R, RHW = R * HW, total = T * RHW;
      const uint32_t magic_rhw = rb_magic_dev(RHW);
      const float* wq = sa.flat + cv.w;
      const float* bq = sa.flat + cv.b;
      for (int i = tid; i < total; i += RT_THREADS) {
        const int tt = rb_div(i, RHW, magic_rhw), r = i - tt * RHW;
        const int rc = rb_div(r, HW, sa.magic_hw), pp = r - rc * HW;
        const int y = rb_div(pp, sa.W, sa.magic_w), x = pp - y * sa.W;
        if (b0 + tt >= sa.batch) continue;
        const float* xin = tile + (size_t)(tt * phw + sa.shift[tt]) * sa.Cs + in0 + (y * sa.PW + x) * sa.Cs;
        const float* w = wq + (size_t)rc * sa.C;
        float acc = 0.f;
        int c = 0;
        for (; c + 3 < sa.C; c += 4) {
          const f32x4 v = *(const f32x4*)(xin + c);
          acc = fmaf(v[0], w[c], acc);
          acc = fmaf(v[1], w[c + 1], acc);
          acc = fmaf(v[2], w[c + 2], acc);
          acc = fmaf(v[3], w[c + 3], acc);
        }
        for (; c < sa.C; ++c) acc = fmaf(xin[c], w[c], acc);
        heads[cv.out_lds + tt * cv.out_stride + r] = acc + bq[rc];
      }
    }
    for (int idx = tid; idx < ((sa.dbg & 8) ? 0 : T * sa.C); idx += RT_THREADS) {
      const int tt = rb_div(idx, sa.C, sa.magic_c), c = idx - tt * sa.C;
      const float* base = tile + (size_t)(tt * phw + sa.shift[tt]) * sa.Cs + in0 + c;
      float l = base[0], h = l;
      for (int y = 0; y < sa.H; ++y)
        for (int x = 0; x < sa.W; ++x) {
          const float v = base[(y * sa.PW + x) * sa.Cs];
          l = fminf(l, v);
          h = fmaxf(h, v);
        }
      float sp = h - l;
      if (sp < 1e-5f) sp += 1e-5f;
      lo[idx] = l;
      sc[idx] = sp;
    }
    __syncthreads();
    if (!(sa.dbg & 8)) {
      const int CHW = sa.C * HW, total = T * CHW;
      for (int i = tid; i < total; i += RT_THREADS) {
        const int tt = rb_div(i, CHW, sa.magic_chw), r = i - tt * CHW;
        const int c = rb_div(r, HW, sa.magic_hw), pp = r - c * HW;
        const int y = rb_div(pp, sa.W, sa.magic_w), x = pp - y * sa.W;
        if (b0 + tt >= sa.batch) continue;
        float* cell = tile + (size_t)(tt * phw + sa.shift[tt]) * sa.Cs + in0 + (y * sa.PW + x) * sa.Cs + c;
        const float s = mzx_div(*cell - lo[tt * sa.C + c], sc[tt * sa.C + c]);
        sa.hidden[((int64_t)(b0 + tt) * NN + sel_leaf[tt]) * Hf + r] = s;
        *cell = s;                                       // the prediction tower's input, in place
      }
    }

    // ---- prediction tower (models.py:392-433) on the scaled state
    {
      const RtTower& a = sa.tw[1];
#define RB_TOWER_HAS_TAIL true
#define RB_TOWER_ACTION(b, t) 0
#define RB_TOWER_LDS_BARRIER true
#include "mzx_tower_layers.inc"
#undef RB_TOWER_HAS_TAIL
#undef RB_TOWER_ACTION
#undef RB_TOWER_LDS_BARRIER
    }
    __syncthreads();
    for (int q = sa.n_conv[0]; q < ((sa.dbg & 8) ? 0 : sa.n_conv[0] + sa.n_conv[1]); ++q) {
      const RtConv& cv = sa.conv[q];
      const int R = cv.R, RHW = R * HW, total = T * RHW;
      const uint32_t magic_rhw = rb_magic_dev(RHW);
      const float* wq = sa.flat + cv.w;
      const float* bq = sa.flat + cv.b;
      for (int i = tid; i < total; i += RT_THREADS) {
        const int tt = rb_div(i, RHW, magic_rhw), r = i - tt * RHW;
        const int rc = rb_div(r, HW, sa.magic_hw), pp = r - rc * HW;
        const int y = rb_div(pp, sa.W, sa.magic_w), x = pp - y * sa.W;
        if (b0 + tt >= sa.batch) continue;
        const float* xin = tile + (size_t)(tt * phw + sa.shift[tt]) * sa.Cs + in0 + (y * sa.PW + x) * sa.Cs;
        const float* w = wq + (size_t)rc * sa.C;
        float acc = 0.f;
        int c = 0;
        for (; c + 3 < sa.C; c += 4) {
          const f32x4 v = *(const f32x4*)(xin + c);
          acc = fmaf(v[0], w[c], acc);
          acc = fmaf(v[1], w[c + 1], acc);
          acc = fmaf(v[2], w[c + 2], acc);
          acc = fmaf(v[3], w[c + 3], acc);
        }
        for (; c < sa.C; ++c) acc = fmaf(xin[c], w[c], acc);
        heads[cv.out_lds + tt * cv.out_stride + r] = acc + bq[rc];
      }
    }
    __syncthreads();

    // ---- head MLPs (dynamics fc = reward, prediction fc_value / fc_policy, models.py:379-433), level by level: unit
    // (chain, column tile) -> a wave; one row tile whose rows 0 .. T - 1 are the workgroup's trees.  rb_gemm_kernel's
    // chain: weight fragment first (D = (A . B)^T), 16-input chunks in order, k = 0 .. 3; bias, ELU.
    for (int level = 0; level < ((sa.dbg & 16) ? 0 : sa.n_levels); ++level) {
      int unit = wave;
      for (int q = 0; q < sa.n_chains; ++q) {
        if (level >= sa.levels[q]) continue;
        const RtHeadLayer& Ld = sa.lin[q][level];
        for (; unit < Ld.ntiles; unit += RT_THREADS / 64) {
          const int m_lane = lane & 15, g4 = 4 * (lane >> 4);
          const float* wp = sa.der + Ld.w_off + (size_t)unit * (size_t)Ld.cchunks * 256 + (unsigned)lane * 4;
          const float* xr = heads + Ld.in_lds + (m_lane < T ? m_lane : 0) * Ld.in_stride + g4;
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
          // (a layer is 4 .. 11 chunks: the weight fragments of up to eight chunks are requested together -- one L2 round
          // trip per group instead of one per chunk; a head level was 13 us of a 450 us simulation with one fragment ahead)
          for (int c0 = 0; c0 < Ld.cchunks; c0 += 8) {
            f32x4 fb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (c0 + u < Ld.cchunks) fb[u] = *(const f32x4*)(wp + (size_t)(c0 + u) * 256);
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (c0 + u < Ld.cchunks) {
                const f32x4 fa = *(const f32x4*)(xr + (c0 + u) * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[u][k], fa[k], acc, 0, 0, 0);
              }
          }
          const int n0 = unit * 16 + g4;
          if (m_lane < T && b0 + m_lane < sa.batch) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (n0 + u < Ld.out) {
                float x = acc[u];
                x += sa.flat[Ld.b_off + n0 + u];
                if (Ld.elu) x = mzx_elu(x);
                heads[Ld.out_lds + m_lane * Ld.out_stride + n0 + u] = x;
              }
          }
        }
        unit -= Ld.ntiles;                               // the next chain's column tiles continue the round-robin
      }
      __syncthreads();
    }

    // ---- decode, expand, back-propagate (self_play.py:343-353), the row that owns the tree
    if (row_valid && !(sa.dbg & 4))
      row_expand_backprop_body<AW>(p, t, sub, row_in_wave, rs, heads + sa.value_lds + row * sa.value_stride,
                                   heads + sa.reward_lds + row * sa.reward_stride, heads + sa.policy_lds + row * sa.policy_stride);
    __syncthreads();
  }
}

typedef void (*RtSearchFn)(const RtSearchArgs);

template <int AW, int NW>
RtSearchFn rt_pick_mt(int mt) {
  switch (mt) {
    case 1: return rt_search_kernel<1, 1, AW, NW>;
    case 2: return rt_search_kernel<2, 1, AW, NW>;
    case 3: return rt_search_kernel<3, 1, AW, NW>;
    case 4: return rt_search_kernel<4, 1, AW, NW>;
    case 5: return rt_search_kernel<5, 1, AW, NW>;
    case 6: return rt_search_kernel<6, 1, AW, NW>;
    case 7: return rt_search_kernel<7, 1, AW, NW>;
    case 8: return rt_search_kernel<8, 1, AW, NW>;
    default: return nullptr;
  }
}
template <int NW>
RtSearchFn rt_pick(int mt, bool wide, int num_actions) {
  return wide ? rt_pick_mt<0, NW>(mt) : (num_actions <= 4 ? rt_pick_mt<4, NW>(mt) : rt_pick_mt<16, NW>(mt));
}
constexpr int RT_MT_MAX = 8;

inline int64_t rt_align(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// ---- planning (host).  The recurrent program must be EXACTLY: dynamics tower with the scaling operator and one small
// 1x1 convolution in its tail, prediction tower reading the scaled state with two small 1x1 convolutions in its tail, and
// three head chains behind those convolutions -- every operator of the program in one of these.

bool rt_structure(const mzx_net* net, const char** why) {
  auto no = [&](const char* m) { if (why) *why = m; return false; };
  if (!net || net->cfg.network != 1 || !net->rb.ok || !net->rb.recurrent.ok || net->rb_no_towers) return no("no streamed plan with towers");
  const std::vector<OpDesc>& prog = net->prog_recurrent;
  const RbProgram& R = net->rb.recurrent;
  if (R.towers.size() != 2) return no("the recurrent program does not have exactly two towers");
  const RbTower& t0 = R.towers[0];
  const RbTower& t1 = R.towers[1];
  if (t0.first != 0 || t0.C != t1.C || t0.H != t1.H || t0.W != t1.W) return no("towers of different shapes");
  if (t0.ntiles > 8 || t0.count > RT_MAX_LAYERS || t1.count > RT_MAX_LAYERS) return no("tower too wide / too deep");
  if (prog[0].in != BUF_IN || R.ops[0].cin != t0.C || R.ops[0].asum_off < 0) return no("the dynamics tower does not start at the state + action plane");
  if (prog[t1.first].in != BUF_HIDDEN || R.ops[t1.first].cin != t1.C) return no("the prediction tower does not read the scaled state");
  if ((int64_t)t0.C * t0.H * t0.W != net->hidden_size) return no("hidden state is not the tower's board");
  std::vector<char> covered(prog.size(), 0);
  int scales = 0;
  for (int w = 0; w < 2; ++w) {
    const RbTower& tw = R.towers[w];
    for (int k = tw.first; k < tw.first + tw.count; ++k) covered[k] = 1;
    int convs = 0;
    for (int m = tw.first + tw.count; m < tw.first + tw.count + tw.n_tail; ++m) {
      const OpDesc& d = prog[m];
      if (d.kind == OP_SCALE) {
        if (w != 0 || d.out != BUF_HIDDEN) return no("scaling operator outside the dynamics tail");
        ++scales;
      } else if (d.kind == OP_CONV1) {
        if (R.ops[m].head_chain < 0) return no("tail convolution without a head chain");
        ++convs;
      } else return no("unknown tail operator");
      covered[m] = 1;
    }
    if (convs != (w == 0 ? 1 : 2)) return no("unexpected number of head convolutions");
    // nobody outside the tail reads the tower's output
    const int out = prog[tw.first + tw.count - 1].out;
    for (int m = tw.first + tw.count + tw.n_tail; m < (int)prog.size(); ++m) {
      if (prog[m].in == out || prog[m].res == out) return no("the tower's output has a reader outside its tail");
      if (prog[m].out == out) break;
    }
  }
  if (scales != 1) return no("no scaling operator in the dynamics tail");
  if (R.heads.n_chains != 3) return no("not three head chains");
  bool have[3] = {false, false, false};
  for (int q = 0; q < 3; ++q) {
    const RbHeadChain& hc = R.heads.chain[q];
    if (hc.count < 1 || hc.count > RT_MAX_LEVELS) return no("head chain depth");
    for (int k = hc.first; k < hc.first + hc.count; ++k) {
      covered[k] = 1;
      if (R.ops[k].ntiles > 8 || R.ops[k].w_off < 0) return no("head layer too wide");
    }
    const int out = prog[hc.first + hc.count - 1].out;
    if (out == BUF_VALUE) have[0] = true;
    else if (out == BUF_REWARD) have[1] = true;
    else if (out == BUF_POLICY) have[2] = true;
    else return no("head chain does not end in a network output");
  }
  if (!have[0] || !have[1] || !have[2]) return no("missing head");
  for (char c : covered)
    if (!c) return no("an operator outside towers, tails and head chains");
  return true;
}

// LDS carve for T trees per workgroup; fills the geometry and head descriptors of `a`.
// Wave grid of a workgroup of `waves` waves that owns T samples of tower `tw` (rb_tower_grid's for 8 waves, one column
// tile per wave): column tiles over WN waves, the other waves deep in rows.
bool rt_grid(const RbTower& tw, int T, int waves, RbTowerShape& c) {
  if (tw.ntiles > 8 || tw.ntiles > waves) return false;
  c.T = T;
  c.rows = T * tw.H * tw.W;
  c.mtiles = (c.rows + 15) / 16;
  c.NT = 1;
  c.WN = tw.ntiles;
  c.WM = std::max(1, std::min(waves / c.WN, c.mtiles));
  c.MT = (c.mtiles + c.WM - 1) / c.WM;
  c.Cs = 16 * tw.cchunks + 8;
  c.lds = 0; c.groups = 0; c.per_cu = 1;
  return true;
}

// Which (sample, position) every row of the workgroup's MFMA tiles is, and where the boards lie in the tile.
// ds_read_b128 serves a wave in four groups of sixteen lanes, a group in one LDS cycle when its sixteen 16-byte slots are
// distinct modulo 16 (MI355X_MICROARCH.md, LDS); lane l = (row l & 15, channel quad l >> 4) of a tile reads slot 18 cell +
// quad (Cs = 72 floats: 18 slots per cell), and a group holds rows {0-3, 12-15} with one quad and rows {4-11} with the
// next: conflict-free exactly when the cells of rows {0-3, 12-15} are distinct modulo 8 and those of rows {4-11} too.  In
// raster order they are not (a 7-wide board in a 9-wide halo: cells 0-6, 9-15, 18, 19 -- half of all LDS cycles of the K
// loops were conflict cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  No output depends on which row computes it, so the
// positions are DEALT: eight residues to every half tile, boards shifted by a few cells so that the residues of the
// workgroup's positions are evenly filled.  Rows without a position read a cell of a residue their half tile lacks.
void rt_row_order(RtSearchArgs& a, int mtiles) {
  assert(16 * mtiles <= RT_MAX_ROWS && a.T <= RT_MAX_T);      // (rt_plan rejects such candidates before carving)
  const int T = a.T, HW = a.H * a.W, phw = a.PH * a.PW, halves = 2 * mtiles;
  const bool deal = (a.Cs / 4) % 2 == 0 && (a.Cs / 4) % 4 != 0 && T <= RT_MAX_T && 16 * mtiles <= RT_MAX_ROWS;   // slot stride 2 (mod 4)
  int count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int shift = 0;
  for (int t = 0; t < T; ++t) {
    // shift of board t: the smallest (not below its predecessor's: boards must not overlap) that fills the emptiest residues
    int best = shift, best_max = 1 << 30;
    for (int sft = shift; deal && sft < shift + 8 && sft < 256; ++sft) {
      int c[8];
      for (int r = 0; r < 8; ++r) c[r] = count[r];
      for (int p = 0; p < HW; ++p) ++c[(t * phw + sft + (p / a.W) * a.PW + p % a.W) & 7];
      int mx = 0;
      for (int r = 0; r < 8; ++r) mx = std::max(mx, c[r]);
      if (mx < best_max) { best_max = mx; best = sft; }
    }
    shift = best;
    a.shift[t] = (uint8_t)shift;
    for (int p = 0; p < HW; ++p) ++count[(t * phw + shift + (p / a.W) * a.PW + p % a.W) & 7];
  }
  a.tile_cells = T * phw + shift;
  // buckets of rows by residue, dealt to the half tiles: every half takes one row of each of the (up to eight) fullest buckets
  std::vector<uint16_t> bucket[8];
  for (int t = 0; t < T; ++t)
    for (int p = 0; p < HW; ++p) bucket[(t * phw + a.shift[t] + (p / a.W) * a.PW + p % a.W) & 7].push_back((uint16_t)(t * HW + p));
  std::vector<std::vector<uint16_t>> half(halves);
  bool ok = deal;
  for (int h = 0; h < halves && ok; ++h) {
    // (a bucket with more rows than halves left cannot be emptied without a conflict: raster order then)
    for (int r = 0; r < 8; ++r) ok = ok && (int)bucket[r].size() <= halves - h;
    for (int r = 0; r < 8 && ok; ++r)
      if (!bucket[r].empty()) { half[h].push_back(bucket[r].back()); bucket[r].pop_back(); }
  }
  for (int r = 0; r < 8; ++r) ok = ok && bucket[r].empty();
  if (!ok) {      // raster order, boards unshifted
    for (int t = 0; t < T; ++t) a.shift[t] = 0;
    a.tile_cells = T * phw;
    for (int m = 0; m < 16 * mtiles; ++m) a.perm[m] = m < T * HW ? (uint16_t)m : (uint16_t)0x8000;
    return;
  }
  // the fullest halves first would leave the tail tiles empty; interleave so that every tile's two halves are balanced:
  // half h -> tile h / 2; rows {0-3, 12-15} of the tile = its even half, rows {4-11} = its odd half
  static const int slot_a[8] = {0, 1, 2, 3, 12, 13, 14, 15}, slot_b[8] = {4, 5, 6, 7, 8, 9, 10, 11};
  for (int h = 0; h < halves; ++h) {
    const int* slot = (h & 1) ? slot_b : slot_a;
    bool used[8] = {false, false, false, false, false, false, false, false};
    auto residue = [&](uint16_t pm) { const int t = pm / HW, p = pm % HW; return (t * phw + a.shift[t] + (p / a.W) * a.PW + p % a.W) & 7; };
    for (uint16_t pm : half[h]) used[residue(pm)] = true;
    int free_r = 0;
    for (int k = 0; k < 8; ++k) {
      uint16_t v;
      if (k < (int)half[h].size()) v = half[h][k];
      else {
        while (free_r < 8 && used[free_r]) ++free_r;      // a cell of a residue this half lacks (cells 0 .. 7 of the tile)
        v = (uint16_t)(0x8000 | (free_r & 7));
        if (free_r < 8) used[free_r] = true;
      }
      a.perm[(h / 2) * 16 + slot[k]] = v;
    }
  }
}

size_t rt_carve(const mzx_search* s, int T, int waves, RtSearchArgs& a) {
  const mzx_net* net = s->net;
  const std::vector<OpDesc>& prog = net->prog_recurrent;
  const RbProgram& R = net->rb.recurrent;
  const RbTower& t0 = R.towers[0];
  RbTowerShape sh;
  if (!rt_grid(t0, T, waves, sh)) return 0;
  a.C = t0.C; a.H = t0.H; a.W = t0.W; a.PH = t0.H + 2; a.PW = t0.W + 2; a.Cs = sh.Cs; a.cchunks = t0.cchunks;
  a.T = T; a.rows = sh.rows; a.mtiles = sh.mtiles;
  int64_t o = 0;
  a.off_tables = (int32_t)o; o += (int64_t)2 * (s->p.num_nodes + 1) * 8;
  o = rt_align(o, 16);
  a.off_rowtab = (int32_t)o; o += (int64_t)3 * 16 * sh.mtiles * 4;
  o = rt_align(o, 16);
  rt_row_order(a, sh.mtiles);
  a.off_tile = (int32_t)o; o += (int64_t)a.tile_cells * a.Cs * 4;
  a.off_scale = (int32_t)o; o += (int64_t)2 * T * a.C * 4;
  o = rt_align(o, 16);
  a.off_sel = (int32_t)o; o += (int64_t)3 * ((T + 3) & ~3) * 4;
  a.off_rowsel = (int32_t)o; o += (int64_t)T * ROWSEL_INTS * 4;
  o = rt_align(o, 16);
  a.off_heads = (int32_t)o;
  // head rows: a chain's input (the tail convolution's output, channel-major = view(-1, R H W)) and every layer's output,
  // each padded to whole 16-input chunks of its reader (zero, written once at kernel start)
  int64_t hf = 0;
  a.n_chains = R.heads.n_chains; a.n_levels = 0;
  int nconv[2] = {0, 0};
  for (int q = 0; q < R.heads.n_chains; ++q) {
    a.levels[q] = R.heads.chain[q].count;
    a.n_levels = std::max(a.n_levels, a.levels[q]);
  }
  int ci = 0;
  for (int w = 0; w < 2; ++w)      // (conv[] is ordered by tower: the dynamics tower's convolution first)
    for (int q = 0; q < R.heads.n_chains; ++q) {
      const RbHeadChain& hc = R.heads.chain[q];
      if (R.ops[hc.conv_op].tower_of_tail != w) continue;
      const OpDesc& dc = prog[hc.conv_op];
      RtConv& cv = a.conv[ci++];
      ++nconv[w];
      cv.w = dc.w; cv.b = dc.b; cv.R = dc.cout; cv.pad = 0;
      const int in_stride = 16 * R.ops[hc.first].cchunks;
      cv.out_lds = (int32_t)hf; cv.out_stride = in_stride;
      int in_lds = (int32_t)hf, stride = in_stride;
      hf += (int64_t)T * in_stride;
      for (int l = 0; l < hc.count; ++l) {
        const OpDesc& dl = prog[hc.first + l];
        const RbOp& ol = R.ops[hc.first + l];
        RtHeadLayer& L = a.lin[q][l];
        L.w_off = ol.w_off; L.b_off = dl.b;
        L.cchunks = ol.cchunks; L.ntiles = ol.ntiles; L.out = dl.out_features; L.elu = dl.elu;
        L.in_lds = in_lds; L.in_stride = stride;
        const int out_stride = (l + 1 < hc.count) ? 16 * R.ops[hc.first + l + 1].cchunks : ((dl.out_features + 3) & ~3);
        L.out_lds = (int32_t)hf; L.out_stride = out_stride;
        hf += (int64_t)T * out_stride;
        in_lds = L.out_lds; stride = out_stride;
        if (l + 1 == hc.count) {
          if (dl.out == BUF_VALUE) { a.value_lds = L.out_lds; a.value_stride = out_stride; }
          else if (dl.out == BUF_REWARD) { a.reward_lds = L.out_lds; a.reward_stride = out_stride; }
          else { a.policy_lds = L.out_lds; a.policy_stride = out_stride; }
        }
      }
    }
  a.n_conv[0] = nconv[0]; a.n_conv[1] = nconv[1];
  a.heads_floats = (int32_t)hf;
  o += hf * 4;
  return (size_t)rt_align(o, 16);
}

void rt_fill_tower(const mzx_net* net, const RbTower& tw, const RbTowerShape& sh, RtTower& a) {
  const std::vector<OpDesc>& prog = net->prog_recurrent;
  const RbProgram& R = net->rb.recurrent;
  memset(&a, 0, sizeof(a));
  const RbOp& o0 = R.ops[tw.first];
  a.der = net->d_derived;
  if (o0.asum_off >= 0) { a.asum = net->d_derived + o0.asum_off; a.num_actions = net->cfg.action_space_size; }
  a.C = tw.C; a.Cs = sh.Cs; a.PW = tw.W + 2; a.WN = sh.WN; a.WM = sh.WM; a.mtiles = sh.mtiles; a.ntiles = tw.ntiles;
  a.layers = tw.count; a.rowskip = (a.PW - 3) * sh.Cs; a.write_out = 0; a.y = nullptr; a.y_sstride = 0; a.y_vec = 0; a.dbg = 0;
  for (int l = 0; l < tw.count; ++l) {        // (as rb_launch_tower, mzx_batched.hip)
    const OpDesc& d = prog[tw.first + l];
    const RbOp& o = R.ops[tw.first + l];
    RbTowerLayer& L = a.layer[l];
    L.w_off = o.w_off;
    L.bn_alpha = d.bn.channels ? d.bn.alpha : -1;
    L.bn_beta = d.bn.channels ? d.bn.beta : -1;
    L.cchunks = o.cchunks;
    L.flags = (o.act == RZ_ACT_RELU ? 1 : 0) | (d.res != -100 ? 4 : 0);
    if (l + 1 < tw.count && prog[tw.first + l + 1].res != -100) L.flags |= 2;
  }
}

uint32_t rt_magic(int d) { return d > 1 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u; }

// Trees per workgroup and waves per workgroup for a shard of `batch` trees.  Whole searches run in ROUNDS of co-resident
// workgroups (a workgroup keeps its trees for all simulations: a partly filled last round costs a whole one); a round
// costs the row tiles its workgroups put on a CU's matrix pipes, over what the tiling gets out of them.
// What the measurements say (profiles/r05_rt_experiments.txt): the K loops are bound by OPERAND DELIVERY and by what a wave
// can issue between its MFMAs, not by the matrix pipe -- with the position and weight loads knocked out they run at 0.98 of
// it -- and a workgroup's phases outside the K loops (tree walks, staging, tails, head MLPs, epilogues: 15 % of a step)
// are hidden by a co-resident workgroup only in part.  So: few, fat waves.  A wave with six or eight row tiles feeds 24 /
// 32 MFMAs from one weight fragment (256 registers, two waves per SIMD); up to four row tiles per wave compile to 128
// registers and run four waves per SIMD.  Whole steps, connect4 x 200 simulations, of the FP32 MFMA peak per PADDED row
// tile (measured fraction / share of valid rows; section 9 of the experiments file):
//   <3,1> 0.71 (two trees per 512-thread workgroup, two workgroups per CU, 1024 trees), <4,1> 0.72, <6,1> 0.72, <8,1> 0.74;
//   a CU's wave slots half empty: x 0.92 (shallow tilings), x 0.83 (deep); a single workgroup on the CU (nobody runs while it
//   walks its trees): x 0.95.
struct RtPlan {
  bool ok = false;
  int T = 0, MT = 0, waves = 8, groups = 0, per_cu = 1;
  size_t lds = 0;
  RtSearchArgs a;
};

RtPlan rt_plan_compute(const mzx_search* s) {
  RtPlan best;
  if (!rt_structure(s->net, nullptr) || !row_search_supported(s->p)) return best;
  const RbProgram& R = s->net->rb.recurrent;
  const int force_t = tune(TUNE_RT_TREES), force_w = tune(TUNE_RT_WAVES);
  const int batch = s->p.num_trees;
  static const double eff_mt[RT_MT_MAX + 1] = {1.0, 0.45, 0.50, 0.706, 0.718, 0.71, 0.723, 0.73, 0.74};
  double best_cost = 1e30;
  for (int waves = 8; waves >= 4; waves >>= 1) {
    if (force_w > 0 && waves != force_w) continue;
    for (int T = 16; T >= 1; --T) {
      if (force_t > 0 && T != force_t) continue;
      RbTowerShape sh;
      if (!rt_grid(R.towers[0], T, waves, sh) || sh.MT > RT_MT_MAX) continue;
      // (RtSearchArgs::perm / ::shift are sized for these: narrow towers in forced modes reach 64 row tiles otherwise, and
      // rt_row_order's raster fallback would write past perm[] -- ADVICE r5)
      if (16 * sh.mtiles > RT_MAX_ROWS || T > RT_MAX_T) continue;
      if (T * 16 > waves * 64) continue;                     // a 16-lane row per tree
      RtPlan c;
      memset(&c.a, 0, sizeof(c.a));
      c.lds = rt_carve(s, T, waves, c.a);
      if (c.lds == 0 || c.lds > (size_t)RT_LDS_MAX) continue;
      c.T = T; c.MT = sh.MT; c.waves = waves; c.groups = (batch + T - 1) / T;
      // up to four row tiles per wave: 128-register instantiations, four waves per SIMD = sixteen per CU; deeper: two per SIMD
      const int slots = sh.MT <= 4 ? 16 : 8;
      c.per_cu = (int)std::min<size_t>((size_t)(slots / waves), (size_t)RT_LDS_MAX / c.lds);
      if (c.per_cu < 1) continue;
      const int64_t cap = (int64_t)256 * c.per_cu;
      const int64_t full = c.groups / cap, rem = c.groups % cap;
      auto round_cost = [&](int ways) {
        const double filled = (double)(ways * waves) / slots;      // share of the CU's wave slots (of this register class) in use
        const double occ = filled >= 1.0 ? 1.0 : (sh.MT <= 4 ? (filled >= 0.5 ? 0.92 : 0.72) : 0.83);
        return (double)ways * sh.mtiles / (eff_mt[sh.MT] * occ * (ways == 1 ? 0.95 : 1.0));
      };
      double cost = (double)full * round_cost(c.per_cu);
      if (rem) cost += round_cost((int)std::min<int64_t>(c.per_cu, (rem + 255) / 256));
      cost += 1e-6 * c.groups;
      c.ok = true;
      if (cost < best_cost) { best_cost = cost; best = c; }
    }
  }
  return best;
}

// The plan of a handle is a function of its shard size, the two forcing knobs and the network's structure (fixed at
// mzx_net_create): about 32 (waves, trees) candidates through rt_carve / rt_row_order, several kilobytes of argument struct
// each.  mzx_selfplay_search runs it on every move (route check + launch), so it is computed once per handle and key
// (ADVICE r5).  Pointers into the arena / weight buffers are filled per launch (rt_search_simulations works on a copy).
const RtPlan& rt_plan(const mzx_search* s) {
  const int64_t key[4] = {s->p.num_trees, tune(TUNE_RT_TREES), tune(TUNE_RT_WAVES),
                          ((int64_t)s->p.num_nodes << 8) | (s->net ? (s->net->rb_no_towers ? 1 : 0) | (s->net->rb_force ? 2 : 0) : 4)};
  if (!s->rt_plan_cache || memcmp(key, s->rt_plan_key, sizeof(key)) != 0) {
    s->rt_plan_cache = std::make_shared<RtPlan>(rt_plan_compute(s));
    memcpy(s->rt_plan_key, key, sizeof(key));
  }
  return *static_cast<const RtPlan*>(s->rt_plan_cache.get());
}

}  // namespace

bool rt_search_supported(const mzx_search* s) { return s && s->net && rt_plan(s).ok; }

// {trees per workgroup, row tiles per wave, workgroups, workgroups per CU, LDS bytes, threads per workgroup}; zeros when the
// kernel does not take the search
void rt_search_shape(const mzx_search* s, int32_t out[6]) {
  const RtPlan& P = rt_plan(s);
  out[0] = P.ok ? P.T : 0; out[1] = P.ok ? P.MT : 0; out[2] = P.ok ? P.groups : 0; out[3] = P.ok ? P.per_cu : 0;
  out[4] = P.ok ? (int32_t)P.lds : 0; out[5] = P.ok ? P.waves * 64 : 0;
}

// The simulations of a search whose roots are in the arena (RootInitOp done, root states in the node store).
int rt_search_simulations(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream) {
  RtPlan P = rt_plan(s);
  if (!P.ok) { set_error("tower whole-search kernel: configuration not supported"); return MZX_ERR_INVALID; }
  const mzx_net* net = s->net;
  const ArenaView v = arena_view(s, d_arena);
  const RbProgram& R = net->rb.recurrent;
  RtSearchArgs& a = P.a;
  a.p = v.p; a.L = s->L;
  a.trees = v.arena.trees; a.tape = io->d_tape; a.hidden = v.arena.hidden;
  a.flat = net->d_flat; a.der = net->d_derived;
  a.num_sims = s->p.num_sims; a.sim0 = 0; a.batch = s->p.num_trees;
  a.magic_hw = rt_magic(a.H * a.W); a.magic_w = rt_magic(a.W); a.magic_rows = rt_magic(16 * a.mtiles);
  a.magic_chw = rt_magic(a.C * a.H * a.W); a.magic_c = rt_magic(a.C);
  RbTowerShape sh;
  rt_grid(R.towers[0], P.T, P.waves, sh);
  rt_fill_tower(net, R.towers[0], sh, a.tw[0]);
  rt_fill_tower(net, R.towers[1], sh, a.tw[1]);
  const bool wide = s->p.num_actions > FUSED_ROW || 2 * s->p.support_size + 1 > 2 * FUSED_ROW;
  RtSearchFn fn = P.waves == 8 ? rt_pick<8>(P.MT, wide, s->p.num_actions) : rt_pick<4>(P.MT, wide, s->p.num_actions);
  const int dbg = tune(TUNE_RT_DBG);
  a.tw[0].dbg = a.tw[1].dbg = dbg & 3;
  a.dbg = dbg;
  if (!fn) { set_error("tower whole-search kernel: no instantiation for %d row tiles per wave", P.MT); return MZX_ERR_INVALID; }
  static std::atomic<uint64_t> lds_attr_done[2][3][RT_MT_MAX + 1];
  if (const int ae = allow_large_lds((const void*)fn, RT_LDS_MAX,
                                     lds_attr_done[P.waves == 8][wide ? 0 : (s->p.num_actions <= 4 ? 1 : 2)][P.MT])) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  void* params[] = {(void*)&a};
  const hipError_t e = hipLaunchKernel((const void*)fn, dim3(P.groups), dim3(P.waves * 64), params, P.lds, stream);
  if (e != hipSuccess) {
    set_error("tower whole-search launch failed: %s (grid %d, %zu bytes of LDS, %d trees per workgroup)", hipGetErrorString(e),
              P.groups, P.lds, P.T);
    return MZX_ERR_RUNTIME;
  }
  return MZX_OK;
}
#endif  // !MZX_HOSTCHECK

}  // namespace mzx
