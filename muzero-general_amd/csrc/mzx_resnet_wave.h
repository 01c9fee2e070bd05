// mzx_resnet_wave.h -- MCTS.run (/root/reference/self_play.py:319-355) for residual networks on SMALL boards
// (H * W <= 16: one MFMA row tile per tree; tic-tac-toe, BASELINE config C3): a WAVE owns a tree for the whole
// search.
//
// mzx_resnet_search.h spreads the (tree, position) rows of a workgroup's trees over its waves and separates
// the operators of recurrent_inference (models.py:555-623) by workgroup barriers; on a 3 x 3 board the layer
// GEMMs are one row tile each, so a simulation there is 9 barrier slots of ~4.6 k cycles of which the MFMAs
// are ~1.2 k.  Here the 16 rows of a row tile are ONE tree's positions: every operator of the program --
// 3x3 / 1x1 convolutions (rows = positions) and the head MLPs (a tile with one valid row) -- reads and writes
// only the wave's own LDS regions, so the four waves of a workgroup run four independent searches without a
// single workgroup barrier after set-up, each on its own SIMD and MFMA pipe:
//   * tree: the 32-byte slot / node records of mzx_fused_fc2.h in LDS (fc2_walk / fc2_expand / fc2_backprop:
//     the binary64 operations of mzx_tree.h in the reference's order), converted from / to the arena's
//     TreeLayout at the ends of the launch;
//   * network: the operator table of mzx_resnet_fused.h (rz_plan) as a table of 12-word descriptors, one scalar
//     word each; operators of a fast class (RzFastClass) run straight-line code (rzw_gemm_s: every fragment
//     requested up front, 36 MFMAs back to back, constant epilogue), the others the interpreter (rzw_gemm: the K
//     loop and epilogue of rz_gemm_tiles<1>); either way the same MFMA sequence per output element as the workgroup
//     engine, so the logits are bit-identical to its -- with the row addresses of the wave's tile held in registers
//     for the whole launch instead of being re-read per operator;
//   * weights: as many packed B-fragment images as fit beside the four trees stay in LDS (the 3x3 convolutions
//     of C3), the rest is streamed from L2.
// Second kernel of this file, rz_tile_search_kernel: boards of 17-64 positions (breakout's 6 x 6 hidden state, C5) --
// a workgroup per tree, a row tile per wave, the same operator loop with one barrier per slot.
// (Included by mzx_resnet_search.h between its kernels and its host driver.)
#pragma once

namespace mzx {

#ifndef MZX_HOSTCHECK

constexpr int RZW_WAVES = 4;
typedef int rzw_i32x4 __attribute__((ext_vector_type(4)));

struct RzWaveArgs {
  RzSearchArgs s;
  int32_t dual;             // wave kernel: 1 = two operators of a slot in flight together (MZX_RZ_DUAL=0: A/B knob)
  int32_t fc1_valu;           // tile kernel: 1 = head layers with <= 16 inputs on the vector ALUs (MZX_RZ_FC1_VALU=0: A/B knob)
  int32_t wl[RZ_MAX_OPS];   // float offset of operator o's packed weights inside the LDS weight area, -1: read from L2
  int32_t wl_floats;        // size of the LDS weight area
};

struct RzwOp {          // compact descriptor in LDS: three 16-byte reads, requested one operator ahead
  int32_t head;         // class | weights in LDS << 4 | store_hidden << 5 | column tiles << 8 | wchunks << 16: the only
                        // word that has to reach a scalar register (branches, loop count)
  int32_t in_off, out_off, res_off;
  int32_t w_at, p0, p1, asum_off;     // w_at: float offset of the packed weights in the LDS weight area / the global image;
                                      // p0, p1: alpha, beta or bias
  int32_t cout, channels, pad0, pad1;
};
static_assert(sizeof(RzwOp) == 48, "RzwOp is fetched as three 16-byte LDS reads");

struct RzWaveLayout { int optab, simg, tables, inv_y, wlds, wave0, wave_stride, o_scratch, o_reg, o_tree, total; };   // floats

// LDS of a workgroup: rowaddr[16], operator roles [RZ_MAX_OPS], compact operator table, small image, UCB tables,
// reciprocals, weights, then per wave {scaling scratch + action value, the program's regions, the tree's records}
__host__ __device__ inline RzWaveLayout rzw_layout(int n_ops, int small_floats, int NN, int Cs, int tree_floats, int rec_floats,
                                                   int wl_floats) {
  RzWaveLayout y;
  int c = 16 + RZ_MAX_OPS;
  y.optab = c; c += n_ops * 12;
  y.simg = c; c += (small_floats + 3) & ~3;
  y.tables = c; c += (4 * (NN + 1) + 3) & ~3;
  y.inv_y = c; c += (2 * (NN + 2) + 3) & ~3;
  y.wlds = c; c += (wl_floats + 3) & ~3;
  y.wave0 = c;
  int w = 0;
  y.o_scratch = w; w += (2 * Cs + 4 + 3) & ~3;
  y.o_reg = w; w += (tree_floats + 3) & ~3;
  y.o_tree = w; w += (rec_floats + 3) & ~3;
  y.wave_stride = w;
  y.total = c + RZW_WAVES * w;
  return y;
}

// operator descriptor -> scalar registers (the LDS copy of the table is read by every lane at the same address)
__device__ __forceinline__ RzOp rzw_fetch_op(const float* image, int o) {
  const int* w = (const int*)image + o * (int)(sizeof(RzOp) / 4);
  int v[sizeof(RzOp) / 4];
#pragma unroll
  for (int k = 0; k < (int)(sizeof(RzOp) / 4); ++k) v[k] = __builtin_amdgcn_readfirstlane(w[k]);
  RzOp op;
  __builtin_memcpy(&op, v, sizeof(RzOp));
  return op;
}

// One column tile of a layer GEMM on the wave's row tile: K loop + epilogue of rz_gemm_tiles<1, *, true>
// (mzx_resnet_fused.h) with T = 1 -- same MFMA order (K-steps alternate between two accumulator tiles that are
// added at the end), same epilogue arithmetic.  `ra_lane` = activation address of row lane & 15, `ra4` = those
// of the four rows 4 * (lane >> 4) + r this lane holds in the D fragment.
__device__ __forceinline__ void rzw_gemm(const RzOp& op, const RzArgs& a, float* reg, const float* simg, const float* wsrc,
                                         const float* scratch, int lane, int ra_lane, const rzw_i32x4& ra4, int nt) {
  const bool pos_rows = (op.rows == RZ_ROWS_POS);
  const int rows = pos_rows ? a.HW : 1;
  const float* in = reg + op.in_off;
  const int abase = (pos_rows ? ra_lane : 0) + 4 * (lane >> 4);
  const f32x4* wp = (const f32x4*)(wsrc) + (size_t)nt * op.wchunks * 64 + lane;
  const int* tbl = (const int*)simg + op.aoff_off;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc_odd = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const f32x4& av, const f32x4& bv) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
    acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc_odd, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc, 0, 0, 0);
    acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc_odd, 0, 0, 0);
  };
  auto load_a = [&](int off, f32x4& av) { av = *(const f32x4*)(in + abase + off); };
  const int wlast = op.wchunks - 1;
  auto load_b = [&](int c, f32x4& bv) { bv = wp[(size_t)(c < wlast ? c : wlast) * 64]; };
  // what the epilogue needs that does not depend on the accumulators, before the K loop
  const int n = nt * 16 + (lane & 15);
  const bool padded = (op.out_layout == RZ_OUT_PADDED);
  const float* res = (op.res_off >= 0) ? reg + op.res_off : nullptr;
  const int m0 = (lane >> 4) * 4;
  float rs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rs[r] = res ? res[ra4[r] + n] : 0.f;
  {
    f32x4 A0, A1, A2, A3, B0, B1, B2, B3;
    const int nch = op.nchunks;
    {
      const rzw_i32x4 q = *(const rzw_i32x4*)tbl;   // chunks 0 .. 3
      load_b(0, B0); load_a(q[0], A0);
      load_b(1, B1); load_a(q[1], A1);
      load_b(2, B2); load_a(q[2], A2);
      load_b(3, B3); load_a(q[3], A3);
    }
    for (int c = 0; c < nch; c += 4) {
      const rzw_i32x4 q = *(const rzw_i32x4*)(tbl + c + 4);   // chunks c + 4 .. c + 7 (entries past the end repeat the last)
      __builtin_amdgcn_sched_barrier(0);
      compute(A0, B0);
      __builtin_amdgcn_sched_barrier(0);
      load_b(c + 4, B0); load_a(q[0], A0);
      if (c + 1 >= nch) break;
      __builtin_amdgcn_sched_barrier(0);
      compute(A1, B1);
      __builtin_amdgcn_sched_barrier(0);
      load_b(c + 5, B1); load_a(q[1], A1);
      if (c + 2 >= nch) break;
      __builtin_amdgcn_sched_barrier(0);
      compute(A2, B2);
      __builtin_amdgcn_sched_barrier(0);
      load_b(c + 6, B2); load_a(q[2], A2);
      if (c + 3 >= nch) break;
      __builtin_amdgcn_sched_barrier(0);
      compute(A3, B3);
      __builtin_amdgcn_sched_barrier(0);
      load_b(c + 7, B3); load_a(q[3], A3);
    }
  }
  acc = acc + acc_odd;
  // ---- epilogue (rz_gemm_tiles): action term, folded BatchNorm, bias, residual, activation
  const bool nv = n < op.cout;
  float al = 1.f, be = 0.f, bi = 0.f;
  if (op.alpha_off >= 0) { al = simg[op.alpha_off + n]; be = simg[op.beta_off + n]; }
  if (op.bias_off >= 0) bi = simg[op.bias_off + n];
  const float floor_v = (op.act == RZ_ACT_RELU) ? 0.f : -MZX_INF;
  float* out = reg + op.out_off;
  const int nstride = (!padded && pos_rows) ? a.HW : 1;
  int base[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = (m0 + r < a.HW) ? m0 + r : 0;
    base[r] = padded ? ra4[r] : (pos_rows ? p : (m0 + r) * op.out_tstride);
  }
  if (op.asum_off >= 0) {   // action plane of the dynamics input (first layer only)
    const float actval = scratch[2 * a.Cs];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = (m0 + r < a.HW) ? m0 + r : 0;
      acc[r] += actval * simg[op.asum_off + n * a.HW + p];
    }
  }
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = acc[r] * al + be;
    x = x + bi;
    x = x + rs[r];
    v[r] = fmaxf(x, floor_v);
  }
  if (op.act == RZ_ACT_ELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = mzx_elu(v[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (m0 + r < rows && nv) out[base[r] + n * nstride] = v[r];
}

// per-plane min-max scaling (models.py:527-553, :574-599) of the wave's tree + hidden-state store (rz_scale, one wave)
__device__ __forceinline__ void rzw_scale(const RzOp& op, const RzArgs& a, float* reg, float* scratch, const int* rowaddr,
                                          int lane, float* hid) {
  const int C = op.channels;
  const float* in = reg + op.in_off;
  float* out = reg + op.out_off;
  const int sub = lane & 15, grp = lane >> 4;
  for (int base = 0; base < C; base += 4) {
    const int c = base + grp;
    const bool valid = c < C;
    const int cc = valid ? c : 0;
    float lo = MZX_INF, hi = -MZX_INF;
    for (int p = sub; p < a.HW; p += 16) {
      const float v = in[rowaddr[p] + cc];
      lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, m, 16));
      hi = fmaxf(hi, __shfl_xor(hi, m, 16));
    }
    if (valid && sub == 0) {
      float sc = hi - lo;
      if (sc < 1e-5f) sc += 1e-5f;
      scratch[2 * c] = lo;
      scratch[2 * c + 1] = sc;
    }
  }
  wave_sync();
  const int per_tree = C * a.HW;
  for (int rem = lane; rem < per_tree; rem += 64) {
    const int c = rz_div(rem, a.HW, a.magic_hw), p = rem - c * a.HW;
    const int ra = rowaddr[p] + c;
    const float y = (in[ra] - scratch[2 * c]) / scratch[2 * c + 1];
    out[ra] = y;
    if (hid) hid[rem] = y;
  }
}

__device__ __forceinline__ RzwOp rzw_fetch(const RzwOp* tab, int o) {
  const rzw_i32x4* w = (const rzw_i32x4*)(tab + o);
  const rzw_i32x4 w0 = w[0], w1 = w[1], w2 = w[2];
  RzwOp q;
  q.head = w0[0]; q.in_off = w0[1]; q.out_off = w0[2]; q.res_off = w0[3];
  q.w_at = w1[0]; q.p0 = w1[1]; q.p1 = w1[2]; q.asum_off = w1[3];
  q.cout = w2[0]; q.channels = w2[1]; q.pad0 = w2[2]; q.pad1 = w2[3];
  return q;
}

// lane constants of the wave's row tile, for the whole launch
struct RzwLane {
  int lane, g4, HW, Cs;
  int m0;               // position of this lane's first D row: 16 * (row tile of the wave) + g4
  int ra_lane;          // activation address of row lane & 15 (A operand)
  rzw_i32x4 ra4;        // ... of rows 4 * (lane >> 4) + r (D fragment)
  int atap[9];          // ra_lane + 4 * (lane >> 4) + offset of tap c of a 3x3 convolution with one channel chunk
  int rsc[4];           // activation address of position (lane >> 4) + 4 k (scaling operator)
};

enum { RZW_K_TAP9 = 0, RZW_K_LIN1 = 1, RZW_K_LIN9 = 2 };
// two slot-mates in flight together: classes behind RzFastClass's in the class field of the FIRST descriptor of the pair
enum { RZW_DUAL_CONV_CONV1 = 9, RZW_DUAL_CONVRES_FC9, RZW_DUAL_CONV1_CONV1, RZW_DUAL_FC9_FC9, RZW_DUAL_FC1_FC1 };
enum { RZW_EP_BN_RELU = 0, RZW_EP_BN_RELU_ASUM, RZW_EP_BN_RES_RELU, RZW_EP_BIAS_POS, RZW_EP_BIAS_ELU_TREE, RZW_EP_BIAS_TREE };

// rzw_gemm with the K structure and the epilogue options as compile-time constants, as three phases of an
// operator in flight: every fragment of the operator is requested up front (nine A + nine B quads at most), the
// MFMAs follow back to back in the order of rz_gemm_tiles<1> (chunk by chunk, K-steps alternating between the two
// accumulator tiles), the epilogue is the same sequence of fp32 operations with absent terms as literal 1.f / 0.f.
template <int KS, int EP>
struct RzwGemm {
  static constexpr bool POS = (EP <= RZW_EP_BIAS_POS);
  static constexpr bool BN = (EP <= RZW_EP_BN_RES_RELU);
  static constexpr int NCH = (KS == RZW_K_LIN1) ? 1 : 9;
  f32x4 A[NCH], B[NCH];
  f32x4 acc, acc_odd;
  float al, be, bi, actval;
  float rs[4], as[4];
  int n;

  __device__ __forceinline__ void load(const RzwOp& q, int wchunks, const float* reg, const float* simg, const f32x4* wp,
                                       const float* scratch, const RzwLane& k, int nt) {
    const float* in = reg + q.in_off;
    wp += (size_t)nt * wchunks * 64;
    n = nt * 16 + (k.lane & 15);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {   // chunk 0 first: LDS answers in request order
      const int at = (KS == RZW_K_TAP9) ? k.atap[c] : (POS ? k.ra_lane : 0) + k.g4 + c * 16;
      B[c] = wp[c * 64];
      A[c] = *(const f32x4*)(in + at);
      if (c == 0) __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue operands: their LDS round trips hide under the MFMAs
    al = 1.f; be = 0.f; bi = 0.f; actval = 0.f;
    if (BN) { al = simg[q.p0 + n]; be = simg[q.p1 + n]; } else { bi = simg[q.p0 + n]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { rs[r] = 0.f; as[r] = 0.f; }
    if (EP == RZW_EP_BN_RES_RELU) {
      const float* res = reg + q.res_off;
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[r] = res[k.ra4[r] + n];
    }
    if (EP == RZW_EP_BN_RELU_ASUM) {
      actval = scratch[2 * k.Cs];
#pragma unroll
      for (int r = 0; r < 4; ++r) as[r] = simg[q.asum_off + n * k.HW + ((k.m0 + r < k.HW) ? k.m0 + r : 0)];
    }
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc_odd = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void mfma(int c) {   // the four K-steps of chunk c
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][0], B[c][0], acc, 0, 0, 0);
    acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][1], B[c][1], acc_odd, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][2], B[c][2], acc, 0, 0, 0);
    acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][3], B[c][3], acc_odd, 0, 0, 0);
  }
  __device__ __forceinline__ void finish(const RzwOp& q, float* reg, const RzwLane& k) {
    acc = acc + acc_odd;
    const bool nv = n < q.cout;
    float* out = reg + q.out_off;
    if (EP == RZW_EP_BN_RELU_ASUM) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += actval * as[r];
    }
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = acc[r] * al + be;
      x = x + bi;
      x = x + rs[r];
      v[r] = fmaxf(x, BN ? 0.f : -MZX_INF);
    }
    if (EP == RZW_EP_BIAS_ELU_TREE) v[0] = mzx_elu(v[0]);
    if (BN) {                                   // padded position-major output
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (k.m0 + r < k.HW && nv) out[k.ra4[r] + n] = v[r];
    } else if (EP == RZW_EP_BIAS_POS) {         // flat [channel][position]
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (k.m0 + r < k.HW && nv) out[k.m0 + r + n * k.HW] = v[r];
    } else {                                    // one valid row: the tree
      if (k.g4 == 0 && nv) out[n] = v[0];   // (head layers run on one wave: row 0 of ITS tile is the tree)
    }
  }
};

template <int KS, int EP>
__device__ __forceinline__ void rzw_gemm_s(const RzwOp& q, int wchunks, float* reg, const float* simg, const f32x4* wp,
                                           const float* scratch, const RzwLane& k, int nt) {
  RzwGemm<KS, EP> g;
  g.load(q, wchunks, reg, simg, wp, scratch, k, nt);
  // every request is issued before the first MFMA (the scheduler would otherwise re-serialise them into a
  // load -> wait -> four MFMAs chain per chunk, one LDS round trip each)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < RzwGemm<KS, EP>::NCH; ++c) g.mfma(c);
  __builtin_amdgcn_sched_barrier(0);
  g.finish(q, reg, k);
}

// Two INDEPENDENT operators (members of one slot of rz_schedule: neither reads or writes a region the other writes)
// in flight together in one wave: both operators' requests, then their MFMAs chunk by chunk in turn, then both
// epilogues.  A lone wave cannot overlap the ~1 k cycles of request / drain / store latency of an operator with
// anything else; the second operator's do overlap with the first one's MFMAs.  Every output element is computed by
// the same instruction sequence as alone.
template <int KS1, int EP1, int KS2, int EP2>
__device__ __forceinline__ void rzw_gemm_dual(const RzwOp& q1, int wch1, const f32x4* wp1, const RzwOp& q2, int wch2, const f32x4* wp2,
                                              float* reg, const float* simg, const float* scratch, const RzwLane& k) {
  RzwGemm<KS1, EP1> g1;
  RzwGemm<KS2, EP2> g2;
  g1.load(q1, wch1, reg, simg, wp1, scratch, k, 0);
  g2.load(q2, wch2, reg, simg, wp2, scratch, k, 0);
  __builtin_amdgcn_sched_barrier(0);
  constexpr int N1 = RzwGemm<KS1, EP1>::NCH, N2 = RzwGemm<KS2, EP2>::NCH;
#pragma unroll
  for (int c = 0; c < (N1 > N2 ? N1 : N2); ++c) {
    if (c < N1) g1.mfma(c);
    if (c < N2) g2.mfma(c);
  }
  __builtin_amdgcn_sched_barrier(0);
  g1.finish(q1, reg, k);
  g2.finish(q2, reg, k);
}

// rzw_scale for <= 16 planes of <= 16 positions, in registers: lane (g, c) holds positions g, g + 4, g + 8, g + 12 of
// plane c (minimum / maximum are exact in any order)
__device__ __forceinline__ void rzw_scale16(const RzwOp& q, float* reg, const RzwLane& k, float* hid) {
  const float* in = reg + q.in_off;
  float* out = reg + q.out_off;
  const int c = k.lane & 15, g = k.lane >> 4;
  const bool cv = c < q.channels;
  float v[4];
  float lo = MZX_INF, hi = -MZX_INF;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool valid = cv && g + 4 * j < k.HW;
    v[j] = in[k.rsc[j] + (cv ? c : 0)];
    lo = valid ? fminf(lo, v[j]) : lo;
    hi = valid ? fmaxf(hi, v[j]) : hi;
  }
  lo = fminf(lo, __shfl_xor(lo, 16)); hi = fmaxf(hi, __shfl_xor(hi, 16));
  lo = fminf(lo, __shfl_xor(lo, 32)); hi = fmaxf(hi, __shfl_xor(hi, 32));
  float sc = hi - lo;
  if (sc < 1e-5f) sc += 1e-5f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = g + 4 * j;
    if (cv && p < k.HW) {
      const float y = (v[j] - lo) / sc;
      out[k.rsc[j] + c] = y;
      if (hid) hid[c * k.HW + p] = y;
    }
  }
}

// AW: lanes that hold a child slot (4 or 16); PROFILE: per-phase cycle counters of wave 0 (mode flag 8)
template <int AW, bool PROFILE>
__global__ void __launch_bounds__(RZW_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
rz_wave_search_kernel(const RzWaveArgs wa) {
  constexpr int NT = RZW_WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) float rz_lds[];
  const RzSearchArgs& sa = wa.s;
  const RzArgs& a = sa.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NN = sa.p.num_nodes;
  const RzWaveLayout y = rzw_layout(a.n_ops, a.small_floats, NN, a.Cs, a.tree_floats, (int)rz_trec_tree_floats(NN, AW), wa.wl_floats);
  int* rowaddr = (int*)rz_lds;            // [16]
  int* roles = rowaddr + 16;              // [RZ_MAX_OPS]: 0 an operator on its own, 1 first of a pair, 2 second of a pair
  RzwOp* optab = (RzwOp*)(rz_lds + y.optab);
  float* simg = rz_lds + y.simg;
  double* tables = (double*)(rz_lds + y.tables);
  double* inv_y = (double*)(rz_lds + y.inv_y);
  float* wlds = rz_lds + y.wlds;
  float* mine = rz_lds + y.wave0 + wave * y.wave_stride;
  float* scratch = mine + y.o_scratch;
  float* reg = mine + y.o_reg;

  // ---- once per launch, the whole workgroup: program image, tables, LDS-resident weights, zeroed regions
  {
    const f32x4* src = (const f32x4*)a.small;
    f32x4* dst = (f32x4*)simg;
    for (int i = tid; i < a.small_floats / 4; i += NT) dst[i] = src[i];
  }
  for (int i = tid; i < 2 * (NN + 1); i += NT) tables[i] = sa.p.pbc_table[i];   // pbc[NN + 1] then sqrt[NN + 1], contiguous
  for (int i = tid; i < NN + 2; i += NT) inv_y[i] = recip_refined((double)(i > 0 ? i : 1));
  // Operator table.  Two operators of one slot (rz_schedule: neither reads nor writes a region the other writes) run
  // as a PAIR when there is code that keeps both in flight (rzw_gemm_dual): the first descriptor of the pair carries
  // the pair's class, the second follows it in the table.  Decided per slot from its first operator on, in parallel.
  const RzOp* gops = (const RzOp*)a.small;
  auto desc = [&](int o) {
    const RzOp op = gops[o];
    RzwOp q;
    const int wl = wa.wl[o];
    q.head = rz_classify(op, a.HW) | ((wl >= 0 ? 1 : 0) << 4) | ((op.store_hidden ? 1 : 0) << 5) |
             (((op.cout + 15) >> 4) << 8) | ((op.wchunks & 0xFFF) << 16);
    q.in_off = op.in_off; q.out_off = op.out_off; q.res_off = op.res_off;
    q.w_at = (wl >= 0) ? wl : op.w_off;
    q.p0 = (op.alpha_off >= 0) ? op.alpha_off : op.bias_off; q.p1 = op.beta_off;
    q.asum_off = op.asum_off; q.cout = op.cout; q.channels = op.channels; q.pad0 = o; q.pad1 = 0;
    return q;
  };
  // class of the pair (i, i + 1), 0: none; `swap`: the second operator leads (its weights are the LDS-resident ones)
  auto pair_class = [&](int i, bool& swap) {
    swap = false;
    if (!wa.dual || i + 1 >= a.n_ops || ((gops[i].sched >> 16) & 1u)) return 0;
    const RzOp &x = gops[i], &z = gops[i + 1];
    const int cx_ = rz_classify(x, a.HW), cz = rz_classify(z, a.HW);
    const int ntx = (x.cout + 15) >> 4, ntz = (z.cout + 15) >> 4;
    const bool lx = wa.wl[i] >= 0, lz = wa.wl[i + 1] >= 0;
    const bool mixed_ok = lx || !lz;                 // (L2, LDS) is not instantiated
    if (cx_ == RZ_FAST_CONV && cz == RZ_FAST_CONV1 && ntx == 1 && ntz == 1 && mixed_ok) return (int)RZW_DUAL_CONV_CONV1;
    if (cx_ == RZ_FAST_CONV_RES && cz == RZ_FAST_FC9_ELU && ntx == 1 && ntz == 1 && mixed_ok) return (int)RZW_DUAL_CONVRES_FC9;
    swap = !mixed_ok;
    if (cx_ == RZ_FAST_CONV1 && cz == RZ_FAST_CONV1 && ntx == 1 && ntz == 1) return (int)RZW_DUAL_CONV1_CONV1;
    if (cx_ == RZ_FAST_FC9_ELU && cz == RZ_FAST_FC9_ELU && ntx == 1 && ntz == 1) return (int)RZW_DUAL_FC9_FC9;
    if (cx_ == RZ_FAST_FC1 && cz == RZ_FAST_FC1) return (int)RZW_DUAL_FC1_FC1;
    swap = false;
    return 0;
  };
  if (tid < a.n_ops) {
    int first = tid;                                  // first operator of this one's slot
    while (first > 0 && ((gops[first - 1].sched >> 16) & 1u) == 0) --first;
    int role = 0;
    for (int i = first; i <= tid;) {                  // greedy from the slot's first operator
      bool sw;
      if (pair_class(i, sw)) { if (i == tid) role = 1; else if (i + 1 == tid) role = 2; i += 2; }
      else ++i;
    }
    roles[tid] = role;
  }
  __syncthreads();
  if (tid < a.n_ops && roles[tid] != 2) {
    const int role = roles[tid];
    RzwOp q1 = desc(tid);
    if (role == 1) {
      bool sw;
      const int dc = pair_class(tid, sw);
      RzwOp q2 = desc(tid + 1);
      if (sw) { const RzwOp t = q1; q1 = q2; q2 = t; }
      q1.head = (q1.head & ~15) | dc;
      q1.pad1 = q2.head;                              // the scalar word of the second descriptor rides with the first
      optab[tid + 1] = q2;
    }
    optab[tid] = q1;
  }
  for (int o = 0; o < a.n_ops; ++o) {
    const int at = wa.wl[o];
    if (at < 0) continue;
    const RzOp* gop = (const RzOp*)a.small + o;
    const int nfl = ((gop->cout + 15) >> 4) * gop->wchunks * 256;
    const f32x4* src = (const f32x4*)(a.weights + gop->w_off);
    f32x4* dst = (f32x4*)(wlds + at);
    for (int i = tid; i < nfl / 4; i += NT) dst[i] = src[i];
  }
  for (int w = 0; w < RZW_WAVES; ++w) {   // halo positions, pad channels and pad words stay zero for the whole launch
    f32x4* z = (f32x4*)(rz_lds + y.wave0 + w * y.wave_stride + y.o_scratch);
    for (int i = tid; i < y.o_tree / 4; i += NT) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (tid < 16) {
    int ra = (a.PW + 1) * a.Cs;           // rows beyond the board: a valid address, never stored to
    if (tid < a.HW) {
      const int yy = rz_div(tid, a.W, a.magic_w), xx = tid - yy * a.W;
      ra = ((yy + 1) * a.PW + xx + 1) * a.Cs;
    }
    rowaddr[tid] = ra;
  }
  __syncthreads();
  const int tree = blockIdx.x * RZW_WAVES + wave;
  if (tree >= a.batch) return;            // whole waves; no workgroup barrier below

  // ---- the wave's tree: arena -> records.  All four 16-lane rows of the wave run the tree code on the SAME tree
  // (identical values, identical stores): no lane masks around the row-wide DPP / ballot operations.
  const int sub = lane & (FUSED_ROW - 1), row_in_wave = lane >> 4;
  TreeRef t;
  t.base = sa.trees + (size_t)tree * sa.L.tree_bytes;
  t.L = sa.L;
  Fc2Tree FT;
  fc2_carve<AW>(FT, (char*)(mine + y.o_tree), NN);
  FT.pbc = tables; FT.sqt = tables + (NN + 1); FT.inv_y = inv_y;
  FT.disc = sa.p.discount; FT.A = sa.p.num_actions; FT.NN = NN; FT.P = sa.p.num_players;
  FT.pb_leaf = FT.pbc[1] * div_by(FT.sqt[1], 1.0, inv_y[1]);
  Fc2Row rst;
  fc2_from_arena<AW>(FT, rst, t, sub);
  wave_sync();
  const uint32_t* tape = sa.tape + (size_t)tree * sa.p.tape_words;
  const int F = a.out_n[0], A = a.out_n[2];
  const int support = sa.p.support_size;
  // rows of the wave's tile, for the whole launch
  const int ra_lane = rowaddr[lane & 15];
  const rzw_i32x4 ra4 = *(const rzw_i32x4*)(rowaddr + (lane >> 4) * 4);
  RzwLane kl;
  kl.lane = lane; kl.g4 = 4 * (lane >> 4); kl.m0 = kl.g4; kl.HW = a.HW; kl.Cs = a.Cs; kl.ra_lane = ra_lane; kl.ra4 = ra4;
#pragma unroll
  for (int c = 0; c < 9; ++c) kl.atap[c] = ra_lane + kl.g4 + (c / 3 - 1) * a.PW * a.Cs + (c % 3 - 1) * a.Cs;   // rz_aoff_entry, one chunk per tap
#pragma unroll
  for (int j = 0; j < 4; ++j) kl.rsc[j] = rowaddr[((lane >> 4) + 4 * j) & 15];
  const int per_tree = a.in_channels * a.HW;

  uint32_t pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = 0;
  const bool prof = PROFILE && sa.prof != nullptr && tid == 0;
  if (prof) t_last = __builtin_readcyclecounter();
#define RZW_PROF(k) if (PROFILE && prof) { const unsigned long long _t = __builtin_readcyclecounter(); pc[k] += (uint32_t)(_t - t_last); t_last = _t; }
  // parent-state gather: element lane + 64 u of [C][H][W] -> its position-major LDS offset, for the whole launch
  int gat[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = lane + 64 * u;
    const int c = rz_div(i, a.HW, a.magic_hw), p = i - c * a.HW;
    gat[u] = (i < per_tree) ? rowaddr[p < a.HW ? p : 0] + c : -1;
  }

  RzwOp qnext = rzw_fetch(optab, 0);
  for (int sim = 0; sim < sa.num_sims; ++sim) {
    // ---- selection (self_play.py:325-334)
    const Fc2Walk wk = fc2_walk<AW>(FT, rst, tape, sa.p.tape_words, sub, row_in_wave);
    // ---- parent state [C][H][W] (arena) -> position-major region: the loads leave before the path fetch
    const float* src = a.in + ((int64_t)tree * a.in_nodes + wk.parent) * per_tree;
    float gv[4] = {0.f, 0.f, 0.f, 0.f};
    if (per_tree <= 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (gat[u] >= 0) gv[u] = src[lane + 64 * u];
    }
    wave_sync();
    const Fc2Lane<AW> lane_ops = fc2_load_lane<AW>(FT, wk, wk.levels >> 4, sub);   // hides behind the network
    RZW_PROF(0)
    {
      if (lane == 0) scratch[2 * a.Cs] = a.use_action ? (float)wk.action / (float)a.num_actions : 0.f;   // action value of the dynamics input
      float* dst = reg + a.in_off;
      if (per_tree <= 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (gat[u] >= 0) dst[gat[u]] = gv[u];
      } else {
        for (int i = lane; i < per_tree; i += 64) {
          const int c = rz_div(i, a.HW, a.magic_hw), p = i - c * a.HW;
          dst[rowaddr[p] + c] = src[i];
        }
      }
    }
    wave_sync();
    RZW_PROF(2)

    // ---- recurrent_inference (models.py:620-623): the operator table in schedule order, one wave
    float* hid = a.hidden_out + ((int64_t)tree * a.out_nodes + wk.leaf) * a.hidden_floats;
#define RZW_CALL(KS, EP)                                                                                          \
  for (int nt = 0; nt < nt_total; ++nt) {                                                                         \
    if (w_in_lds) rzw_gemm_s<KS, EP>(q, wchunks, reg, simg, (const f32x4*)(wlds + q.w_at) + lane, scratch, kl, nt); \
    else rzw_gemm_s<KS, EP>(q, wchunks, reg, simg, (const f32x4*)(a.weights + q.w_at) + lane, scratch, kl, nt);    \
  }
#define RZW_DUAL(K1, E1, K2, E2)                                                                                          \
  {                                                                                                                     \
    if (w_in_lds && l2) rzw_gemm_dual<K1, E1, K2, E2>(q, wchunks, (const f32x4*)(wlds + q.w_at) + lane, q2, wch2,         \
                                                      (const f32x4*)(wlds + q2.w_at) + lane, reg, simg, scratch, kl);    \
    else if (w_in_lds) rzw_gemm_dual<K1, E1, K2, E2>(q, wchunks, (const f32x4*)(wlds + q.w_at) + lane, q2, wch2,          \
                                                     (const f32x4*)(a.weights + q2.w_at) + lane, reg, simg, scratch, kl); \
    else rzw_gemm_dual<K1, E1, K2, E2>(q, wchunks, (const f32x4*)(a.weights + q.w_at) + lane, q2, wch2,                   \
                                       (const f32x4*)(a.weights + q2.w_at) + lane, reg, simg, scratch, kl);              \
  }
    for (int i = 0; i < a.n_ops;) {
      const RzwOp q = qnext;
      const int head = __builtin_amdgcn_readfirstlane(q.head);
      const int cls = head & 15;
      const int used = (cls >= RZW_DUAL_CONV_CONV1) ? 2 : 1;
      const int nxt = (i + used < a.n_ops) ? i + used : 0;
      qnext = rzw_fetch(optab, nxt);                           // in flight while this entry runs
      __builtin_amdgcn_sched_barrier(0);
      const int nt_total = (head >> 8) & 255, wchunks = (head >> 16) & 0xFFF;
      const bool w_in_lds = (head >> 4) & 1, store_hidden = (head >> 5) & 1;
      RZW_PROF(3)   // descriptor (+ the fence below)
      if (used == 2) {   // ---- two slot-mates in flight together
        const RzwOp q2 = rzw_fetch(optab, i + 1);              // its fields are first needed after the first operator's requests
        const int head2 = __builtin_amdgcn_readfirstlane(q.pad1);
        const int nt2 = (head2 >> 8) & 255, wch2 = (head2 >> 16) & 0xFFF;
        const bool l2 = (head2 >> 4) & 1;
        switch (cls) {
          case RZW_DUAL_CONV_CONV1: RZW_DUAL(RZW_K_TAP9, RZW_EP_BN_RELU, RZW_K_LIN1, RZW_EP_BIAS_POS) break;
          case RZW_DUAL_CONVRES_FC9: RZW_DUAL(RZW_K_TAP9, RZW_EP_BN_RES_RELU, RZW_K_LIN9, RZW_EP_BIAS_ELU_TREE) break;
          case RZW_DUAL_CONV1_CONV1: RZW_DUAL(RZW_K_LIN1, RZW_EP_BIAS_POS, RZW_K_LIN1, RZW_EP_BIAS_POS) break;
          case RZW_DUAL_FC9_FC9: RZW_DUAL(RZW_K_LIN9, RZW_EP_BIAS_ELU_TREE, RZW_K_LIN9, RZW_EP_BIAS_ELU_TREE) break;
          default: {   // RZW_DUAL_FC1_FC1: column tile 0 of both together, further column tiles after
            RZW_DUAL(RZW_K_LIN1, RZW_EP_BIAS_TREE, RZW_K_LIN1, RZW_EP_BIAS_TREE)
            for (int nt = 1; nt < nt_total; ++nt) {
              if (w_in_lds) rzw_gemm_s<RZW_K_LIN1, RZW_EP_BIAS_TREE>(q, wchunks, reg, simg, (const f32x4*)(wlds + q.w_at) + lane, scratch, kl, nt);
              else rzw_gemm_s<RZW_K_LIN1, RZW_EP_BIAS_TREE>(q, wchunks, reg, simg, (const f32x4*)(a.weights + q.w_at) + lane, scratch, kl, nt);
            }
            for (int nt = 1; nt < nt2; ++nt) {
              if (l2) rzw_gemm_s<RZW_K_LIN1, RZW_EP_BIAS_TREE>(q2, wch2, reg, simg, (const f32x4*)(wlds + q2.w_at) + lane, scratch, kl, nt);
              else rzw_gemm_s<RZW_K_LIN1, RZW_EP_BIAS_TREE>(q2, wch2, reg, simg, (const f32x4*)(a.weights + q2.w_at) + lane, scratch, kl, nt);
            }
            break;
          }
        }
      } else switch (cls) {
        case RZ_FAST_CONV: RZW_CALL(RZW_K_TAP9, RZW_EP_BN_RELU) break;
        case RZ_FAST_CONV_ASUM: RZW_CALL(RZW_K_TAP9, RZW_EP_BN_RELU_ASUM) break;
        case RZ_FAST_CONV_RES: RZW_CALL(RZW_K_TAP9, RZW_EP_BN_RES_RELU) break;
        case RZ_FAST_CONV1: RZW_CALL(RZW_K_LIN1, RZW_EP_BIAS_POS) break;
        case RZ_FAST_FC9_ELU: RZW_CALL(RZW_K_LIN9, RZW_EP_BIAS_ELU_TREE) break;
        case RZ_FAST_FC1: RZW_CALL(RZW_K_LIN1, RZW_EP_BIAS_TREE) break;
        case RZ_FAST_SCALE16: rzw_scale16(q, reg, kl, store_hidden ? hid : nullptr); break;
        case RZ_FAST_SCALE_GEN: {
          const int o = __builtin_amdgcn_readfirstlane(q.pad0);
          const RzOp op = rzw_fetch_op(simg, o);
          rzw_scale(op, a, reg, scratch, rowaddr, lane, op.store_hidden ? hid : nullptr);
          break;
        }
        default: {   // any other GEMM: the interpreter
          const int o = __builtin_amdgcn_readfirstlane(q.pad0);
          const RzOp op = rzw_fetch_op(simg, o);
          const int at = __builtin_amdgcn_readfirstlane(q.w_at);
          if (w_in_lds) {   // two call sites: the address space of the B fragments is known in each (ds_read / global_load)
            for (int nt = 0; nt < nt_total; ++nt) rzw_gemm(op, a, reg, simg, wlds + at, scratch, lane, ra_lane, ra4, nt);
          } else {
            for (int nt = 0; nt < nt_total; ++nt) rzw_gemm(op, a, reg, simg, a.weights + op.w_off, scratch, lane, ra_lane, ra4, nt);
          }
          break;
        }
      }
      wave_sync();
      // profile: 9-chunk GEMMs from LDS weights / from L2, short GEMMs, scaling
      if (prof) {
        const int pk = (cls == RZ_FAST_SCALE16 || cls == RZ_FAST_SCALE_GEN) ? 7 : ((cls == RZ_FAST_CONV1 || cls == RZ_FAST_FC1) ? 6 : (w_in_lds ? 1 : 5));
        RZW_PROF(pk)
      }
      i += used;
    }
#undef RZW_CALL
#undef RZW_DUAL
    // the leaf's state must have left the wave before a later simulation gathers it (what the workgroup
    // barrier of rz_search_kernel implies)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    RZW_PROF(3)

    // ---- decode, expand, back-propagate (self_play.py:343-353)
    {
      const float* vl = reg + a.out_off[0];
      const float* rl = reg + a.out_off[1];
      const float* pl = reg + a.out_off[2];
      const float value = row_decode2(sub < F ? vl[sub] : 0.f, sub + 16 < F ? vl[sub + 16] : 0.f, F, support, sub);
      const float reward = row_decode2(sub < F ? rl[sub] : 0.f, sub + 16 < F ? rl[sub + 16] : 0.f, F, support, sub);
      const bool in = sub < A;
      const float lg = in ? pl[sub] : 0.f;
      const float m = row_max(in ? lg : -MZX_INF);
      const float e = in ? mzx_expf(lg - m) : 0.f;
      const float den = row_sum(e);
      fc2_expand<AW>(FT, wk.leaf, sub, in, (double)mzx_div(e, den));
      fc2_backprop<AW>(FT, rst, wk, lane_ops, sub, (double)value, (double)reward);
    }
    wave_sync();
    RZW_PROF(4)
  }
#undef RZW_PROF
  if (prof) for (int k = 0; k < 8; ++k) sa.prof[blockIdx.x * 8 + k] = pc[k];
  wave_sync();
  fc2_to_arena<AW>(FT, rst, t, sub);
}

template <int AW, bool PROFILE>
inline int rz_wave_launch_k(const RzWaveArgs& wa, unsigned grid, size_t lds_bytes, stream_t stream) {
  static std::atomic<uint64_t> lds_attr_done{0};   // per instantiation, one bit per device
  if (const int ae = allow_large_lds((const void*)rz_wave_search_kernel<AW, PROFILE>, 160 * 1024, lds_attr_done)) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  hipLaunchKernelGGL((rz_wave_search_kernel<AW, PROFILE>), dim3(grid), dim3(RZW_WAVES * 64), lds_bytes, stream, wa);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("wave-private search kernel launch failed: %s (grid %u, %zu bytes of LDS)", hipGetErrorString(e), grid, lds_bytes);
    return MZX_ERR_RUNTIME;
  }
  return MZX_OK;
}

// Plans the launch; false: the configuration is not one for this kernel (the caller uses rz_search_kernel).
// MZX_RZ_WAVE=0: A/B knob.
inline bool rz_wave_plan(const mzx_search* s, const RzProgram& R, const RzArgs& base, RzWaveArgs& wa, unsigned& grid,
                         size_t& lds_bytes) {
  const RzGeometry& g = s->net->rz.g;
  const int A = s->p.num_actions;
  if (g.HW > 16 || A > FUSED_ROW || 2 * s->p.support_size + 1 > 2 * FUSED_ROW) return false;
  if (rz_env_int("MZX_RZ_WAVE", 1) == 0) return false;
  for (int o = 0; o < R.n_ops; ++o)     // fields of RzwOp::head
    if (R.ops[o].kind == RZ_GEMM && (R.ops[o].cout + 15) / 16 > 255) return false;
  const int AW = A <= 4 ? 4 : 16;
  const int NN = s->p.num_nodes;
  wa.s.net = base;
  RzArgs& a = wa.s.net;
  a.T = 1;
  a.mpad = 16;
  a.small_floats = R.small_floats;      // with the A-fragment offset tables
  const int rec_floats = (int)rz_trec_tree_floats(NN, AW);
  const int64_t fixed = rzw_layout(a.n_ops, a.small_floats, NN, a.Cs, a.tree_floats, rec_floats, 0).total;
  if (4 * fixed > RZ_LDS_BUDGET) return false;
  // weights into LDS, greedily in schedule order while they fit (shared packs once)
  int64_t room = RZ_LDS_BUDGET / 4 - fixed;
  int used = 0;
  for (int o = 0; o < RZ_MAX_OPS; ++o) wa.wl[o] = -1;
  for (int o = 0; o < R.n_ops; ++o) {
    const RzOp& op = R.ops[o];
    if (op.kind != RZ_GEMM) continue;
    int shared = -1;
    for (int q = 0; q < o; ++q)
      if (R.ops[q].kind == RZ_GEMM && R.ops[q].w_off == op.w_off && wa.wl[q] >= 0) shared = wa.wl[q];
    if (shared >= 0) { wa.wl[o] = shared; continue; }
    const int nfl = ((op.cout + 15) / 16) * op.wchunks * 256;
    if (used + nfl <= room) { wa.wl[o] = used; used += nfl; }
  }
  wa.wl_floats = used;
  wa.dual = rz_env_int("MZX_RZ_DUAL", 1) != 0 ? 1 : 0;
  grid = (unsigned)((s->p.num_trees + RZW_WAVES - 1) / RZW_WAVES);
  lds_bytes = (size_t)4 * rzw_layout(a.n_ops, a.small_floats, NN, a.Cs, a.tree_floats, rec_floats, used).total;
  return true;
}

inline int rz_wave_launch(const RzWaveArgs& wa, unsigned grid, size_t lds_bytes, stream_t stream) {
  if (wa.s.prof) {
    return wa.s.p.num_actions <= 4 ? rz_wave_launch_k<4, true>(wa, grid, lds_bytes, stream)
                                   : rz_wave_launch_k<16, true>(wa, grid, lds_bytes, stream);
  }
  return wa.s.p.num_actions <= 4 ? rz_wave_launch_k<4, false>(wa, grid, lds_bytes, stream)
                                 : rz_wave_launch_k<16, false>(wa, grid, lds_bytes, stream);
}

// ---------------------------------------------------------------------------
// Boards of 17 .. 64 positions (2 .. 4 row tiles per tree; breakout's 6 x 6 hidden state, BASELINE C5): a WORKGROUP
// per tree, a row tile per wave, the same lean operator loop -- compact descriptors, operator classes, row
// addresses in registers -- with one workgroup barrier per operator (a 3x3 convolution reads its neighbours' rows).
// Wave 0 walks and updates the tree (records of mzx_fused_fc2.h); head layers with many column tiles (a 601-bin
// support is 38 of them) are spread over the four waves; the value and reward supports are decoded by two waves
// side by side.  Only programs made of fast-class operators run here (rz_tile_plan), anything else stays on
// rz_search_kernel.

struct RzTileLayout { int optab, simg, tables, inv_y, wlds, scratch, reg, tree, total; };   // floats

// LDS of a workgroup: rowaddr[64] + hand-off words [16], compact operator table, small image, UCB tables,
// reciprocals, weights, scaling scratch + action value, the program's regions, the tree's records
__host__ __device__ inline RzTileLayout rzt_layout(int n_ops, int small_floats, int NN, int Cs, int tree_floats, int rec_floats,
                                                   int wl_floats) {
  RzTileLayout y;
  int c = 64 + 16;
  y.optab = c; c += n_ops * 12;
  y.simg = c; c += (small_floats + 3) & ~3;
  y.tables = c; c += (4 * (NN + 1) + 3) & ~3;
  y.inv_y = c; c += (2 * (NN + 2) + 3) & ~3;
  y.wlds = c; c += (wl_floats + 3) & ~3;
  y.scratch = c; c += (2 * Cs + 4 + 3) & ~3;
  y.reg = c; c += (tree_floats + 3) & ~3;
  y.tree = c; c += (rec_floats + 3) & ~3;
  y.total = c;
  return y;
}

// rz_scale of one tree by the whole workgroup (contains a workgroup barrier)
__device__ __forceinline__ void rzt_scale(const RzwOp& q, const RzArgs& a, float* reg, float* scratch, const int* rowaddr, int tid,
                                          float* hid) {
  constexpr int NT = RZW_WAVES * 64;
  const int C = q.channels;
  const float* in = reg + q.in_off;
  float* out = reg + q.out_off;
  const int sub = tid & 15, grp = tid >> 4;
  for (int base = 0; base < C; base += NT / 16) {
    const int c = base + grp;
    const bool valid = c < C;
    const int cc = valid ? c : 0;
    float lo = MZX_INF, hi = -MZX_INF;
    for (int p = sub; p < a.HW; p += 16) {
      const float v = in[rowaddr[p] + cc];
      lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, m, 16));
      hi = fmaxf(hi, __shfl_xor(hi, m, 16));
    }
    if (valid && sub == 0) {
      float sc = hi - lo;
      if (sc < 1e-5f) sc += 1e-5f;
      scratch[2 * c] = lo;
      scratch[2 * c + 1] = sc;
    }
  }
  __syncthreads();
  const int per_tree = C * a.HW;
  for (int rem = tid; rem < per_tree; rem += NT) {
    const int c = rz_div(rem, a.HW, a.magic_hw), p = rem - c * a.HW;
    const int ra = rowaddr[p] + c;
    const float y = (in[ra] - scratch[2 * c]) / scratch[2 * c + 1];
    out[ra] = y;
    if (hid) hid[rem] = y;
  }
}

// RZ_FAST_FC1 (a head layer with <= 16 inputs) on the vector ALUs, one output per thread: with a 601-bin support the
// layer is 38 column tiles of four MFMAs at one valid row in sixteen.  Same bits as the matrix path: the f32 MFMA
// accumulates its four products in k order with one rounding each (a k-ordered fmaf chain), the K-steps of the
// chunk alternate between the two accumulators of rz_gemm_tiles<1>, then the same epilogue.
__device__ __forceinline__ void rzt_fc1_valu(const RzwOp& q, int wchunks, float* reg, const float* simg, const f32x4* wbase, int tid) {
  const f32x4* xin = (const f32x4*)(reg + q.in_off);       // 16 inputs, row 0
  const f32x4 x0 = xin[0], x1 = xin[1], x2 = xin[2], x3 = xin[3];   // x[4 g + j] = xg[j]
  for (int n = tid; n < q.cout; n += RZW_WAVES * 64) {
    const f32x4* w = wbase + (size_t)(n >> 4) * wchunks * 64 + (n & 15);
    const f32x4 w0 = w[0], w1 = w[16], w2 = w[32], w3 = w[48];      // W[4 g + j][n] = wg[j]
    const float bi = simg[q.p0 + n];
    float acc = 0.f, odd = 0.f;
    acc = __builtin_fmaf(x0[0], w0[0], acc); acc = __builtin_fmaf(x1[0], w1[0], acc);
    acc = __builtin_fmaf(x2[0], w2[0], acc); acc = __builtin_fmaf(x3[0], w3[0], acc);
    odd = __builtin_fmaf(x0[1], w0[1], odd); odd = __builtin_fmaf(x1[1], w1[1], odd);
    odd = __builtin_fmaf(x2[1], w2[1], odd); odd = __builtin_fmaf(x3[1], w3[1], odd);
    acc = __builtin_fmaf(x0[2], w0[2], acc); acc = __builtin_fmaf(x1[2], w1[2], acc);
    acc = __builtin_fmaf(x2[2], w2[2], acc); acc = __builtin_fmaf(x3[2], w3[2], acc);
    odd = __builtin_fmaf(x0[3], w0[3], odd); odd = __builtin_fmaf(x1[3], w1[3], odd);
    odd = __builtin_fmaf(x2[3], w2[3], odd); odd = __builtin_fmaf(x3[3], w3[3], odd);
    float xv = (acc + odd) * 1.f + 0.f;
    xv = xv + bi;
    xv = xv + 0.f;
    reg[q.out_off + n] = fmaxf(xv, -MZX_INF);
  }
}

template <int AW, bool PROFILE>
__global__ void __launch_bounds__(RZW_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
rz_tile_search_kernel(const RzWaveArgs wa) {
  constexpr int NT = RZW_WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) float rz_lds[];
  const RzSearchArgs& sa = wa.s;
  const RzArgs& a = sa.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NN = sa.p.num_nodes;
  const RzTileLayout y = rzt_layout(a.n_ops, a.small_floats, NN, a.Cs, a.tree_floats, (int)rz_trec_tree_floats(NN, AW), wa.wl_floats);
  int* rowaddr = (int*)rz_lds;            // [64]
  int* hand = rowaddr + 64;               // [16]: parent, action, leaf | value, reward (float bits)
  RzwOp* optab = (RzwOp*)(rz_lds + y.optab);
  float* simg = rz_lds + y.simg;
  double* tables = (double*)(rz_lds + y.tables);
  double* inv_y = (double*)(rz_lds + y.inv_y);
  float* wlds = rz_lds + y.wlds;
  float* scratch = rz_lds + y.scratch;
  float* reg = rz_lds + y.reg;
  const int tree = blockIdx.x;

  // ---- once per launch
  {
    const f32x4* src = (const f32x4*)a.small;
    f32x4* dst = (f32x4*)simg;
    for (int i = tid; i < a.small_floats / 4; i += NT) dst[i] = src[i];
  }
  for (int i = tid; i < 2 * (NN + 1); i += NT) tables[i] = sa.p.pbc_table[i];
  for (int i = tid; i < NN + 2; i += NT) inv_y[i] = recip_refined((double)(i > 0 ? i : 1));
  if (tid < a.n_ops) {
    const RzOp op = ((const RzOp*)a.small)[tid];
    RzwOp q;
    const int wl = wa.wl[tid];
    // slots (rz_schedule): operators of a slot are independent -- no barrier between them, position-row operators on
    // the tile waves, the k-th tree-row operator (a head MLP layer) of the slot on wave 3 - k
    const RzOp* ops = (const RzOp*)a.small;
    int first = tid;
    while (first > 0 && ((ops[first - 1].sched >> 16) & 1u) == 0) --first;
    int kth = 0;
    for (int i = first; i < tid; ++i) kth += (ops[i].kind == RZ_GEMM && ops[i].rows == RZ_ROWS_TREE) ? 1 : 0;
    const int last = (int)((op.sched >> 16) & 1u), alone = (last && first == tid) ? 1 : 0;
    q.head = rz_classify(op, 0x7FFFFFFF) | ((wl >= 0 ? 1 : 0) << 4) | ((op.store_hidden ? 1 : 0) << 5) | (last << 6) | (alone << 7) |
             (((op.cout + 15) >> 4) << 8) | ((op.wchunks & 0xFFF) << 16) | (((RZW_WAVES - 1 - kth) & 3) << 28);
    q.in_off = op.in_off; q.out_off = op.out_off; q.res_off = op.res_off;
    q.w_at = (wl >= 0) ? wl : op.w_off;
    q.p0 = (op.alpha_off >= 0) ? op.alpha_off : op.bias_off; q.p1 = op.beta_off;
    q.asum_off = op.asum_off; q.cout = op.cout; q.channels = op.channels; q.pad0 = 0; q.pad1 = 0;
    optab[tid] = q;
  }
  for (int o = 0; o < a.n_ops; ++o) {
    const int at = wa.wl[o];
    if (at < 0) continue;
    const RzOp* gop = (const RzOp*)a.small + o;
    const int nfl = ((gop->cout + 15) >> 4) * gop->wchunks * 256;
    const f32x4* src = (const f32x4*)(a.weights + gop->w_off);
    f32x4* dst = (f32x4*)(wlds + at);
    for (int i = tid; i < nfl / 4; i += NT) dst[i] = src[i];
  }
  {
    f32x4* z = (f32x4*)scratch;             // scratch + regions: halo positions, pad channels and pad words stay zero
    for (int i = tid; i < (y.tree - y.scratch) / 4; i += NT) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (tid < 64) {
    int ra = (a.PW + 1) * a.Cs;
    if (tid < a.HW) {
      const int yy = rz_div(tid, a.W, a.magic_w), xx = tid - yy * a.W;
      ra = ((yy + 1) * a.PW + xx + 1) * a.Cs;
    }
    rowaddr[tid] = ra;
  }
  __syncthreads();

  const int sub = lane & (FUSED_ROW - 1), row_in_wave = lane >> 4;
  TreeRef t;
  t.base = sa.trees + (size_t)tree * sa.L.tree_bytes;
  t.L = sa.L;
  Fc2Tree FT;
  fc2_carve<AW>(FT, (char*)(rz_lds + y.tree), NN);
  FT.pbc = tables; FT.sqt = tables + (NN + 1); FT.inv_y = inv_y;
  FT.disc = sa.p.discount; FT.A = sa.p.num_actions; FT.NN = NN; FT.P = sa.p.num_players;
  FT.pb_leaf = FT.pbc[1] * div_by(FT.sqt[1], 1.0, inv_y[1]);
  Fc2Row rst;
  rst.n_nodes = 0; rst.tape_pos = 0; rst.flags = 0; rst.ties = 0; rst.max_depth = 0; rst.sum_depth = 0; rst.root_n = 0; rst.root_to_play = 0;
  if (wave == 0) { fc2_from_arena<AW>(FT, rst, t, sub); wave_sync(); }   // all four rows of wave 0 on the same tree (see rz_wave_search_kernel)
  const uint32_t* tape = sa.tape + (size_t)tree * sa.p.tape_words;
  const int F = a.out_n[0], A = a.out_n[2];
  const int support = sa.p.support_size;
  // rows of the wave's tile, for the whole launch
  const bool tile_valid = 16 * wave < a.HW;
  RzwLane kl;
  kl.lane = lane; kl.g4 = 4 * (lane >> 4); kl.m0 = 16 * wave + kl.g4; kl.HW = a.HW; kl.Cs = a.Cs;
  kl.ra_lane = rowaddr[(16 * wave + (lane & 15)) & 63];
  kl.ra4 = *(const rzw_i32x4*)(rowaddr + ((16 * wave + kl.g4) & 63));
#pragma unroll
  for (int c = 0; c < 9; ++c) kl.atap[c] = kl.ra_lane + kl.g4 + (c / 3 - 1) * a.PW * a.Cs + (c % 3 - 1) * a.Cs;
#pragma unroll
  for (int j = 0; j < 4; ++j) kl.rsc[j] = 0;
  // head layers (rows = the tree) read row 0 of the tile: their A address does not involve the tile
  const int per_tree = a.in_channels * a.HW;
  int gat[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = tid + NT * u;
    const int c = rz_div(i, a.HW, a.magic_hw), p = i - c * a.HW;
    gat[u] = (i < per_tree) ? rowaddr[p < a.HW ? p : 0] + c : -1;
  }

  uint32_t pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = 0;
  const bool prof = PROFILE && sa.prof != nullptr && tid == 0;
  if (prof) t_last = __builtin_readcyclecounter();
#define RZT_PROF(k) if (PROFILE && prof) { const unsigned long long _t = __builtin_readcyclecounter(); pc[k] += (uint32_t)(_t - t_last); t_last = _t; }

  RzwOp qnext = rzw_fetch(optab, 0);
  for (int sim = 0; sim < sa.num_sims; ++sim) {
    // ---- selection (self_play.py:325-334), wave 0
    Fc2Walk wk;
    Fc2Lane<AW> lane_ops;
    if (wave == 0) {
      wk = fc2_walk<AW>(FT, rst, tape, sa.p.tape_words, sub, row_in_wave);
      if (lane == 0) { hand[0] = wk.parent; hand[1] = wk.action; hand[2] = wk.leaf; }
      wave_sync();
      lane_ops = fc2_load_lane<AW>(FT, wk, wk.levels >> 4, sub);
    }
    RZT_PROF(0)
    __syncthreads();
    const int parent = hand[0], action = hand[1], leaf = hand[2];
    // ---- parent state [C][H][W] (arena) -> position-major region
    {
      const float* src = a.in + ((int64_t)tree * a.in_nodes + parent) * per_tree;
      float* dst = reg + a.in_off;
      if (tid == 0) scratch[2 * a.Cs] = a.use_action ? (float)action / (float)a.num_actions : 0.f;
      if (per_tree <= 4 * NT) {
        float gv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) if (gat[u] >= 0) gv[u] = src[tid + NT * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (gat[u] >= 0) dst[gat[u]] = gv[u];
      } else {
        for (int i = tid; i < per_tree; i += NT) {
          const int c = rz_div(i, a.HW, a.magic_hw), p = i - c * a.HW;
          dst[rowaddr[p] + c] = src[i];
        }
      }
    }
    __syncthreads();
    RZT_PROF(2)

    // ---- recurrent_inference: the operator table in schedule order, one barrier per operator
    float* hid = a.hidden_out + ((int64_t)tree * a.out_nodes + leaf) * a.hidden_floats;
#define RZT_CALL(KS, EP, NT0, NTSTEP)                                                                              \
  for (int nt = (NT0); nt < nt_total; nt += (NTSTEP)) {                                                           \
    if (w_in_lds) rzw_gemm_s<KS, EP>(q, wchunks, reg, simg, (const f32x4*)(wlds + q.w_at) + lane, scratch, kl, nt); \
    else rzw_gemm_s<KS, EP>(q, wchunks, reg, simg, (const f32x4*)(a.weights + q.w_at) + lane, scratch, kl, nt);    \
  }
    for (int o = 0; o < a.n_ops; ++o) {
      const RzwOp q = qnext;
      const int head = __builtin_amdgcn_readfirstlane(q.head);
      qnext = rzw_fetch(optab, o + 1 < a.n_ops ? o + 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
      const int cls = head & 15, nt_total = (head >> 8) & 255, wchunks = (head >> 16) & 0xFFF;
      const bool w_in_lds = (head >> 4) & 1, store_hidden = (head >> 5) & 1;
      const bool last = (head >> 6) & 1, alone = (head >> 7) & 1;
      const int twave = (head >> 28) & 3;
      switch (cls) {
        case RZ_FAST_CONV: if (tile_valid) { RZT_CALL(RZW_K_TAP9, RZW_EP_BN_RELU, 0, 1) } break;
        case RZ_FAST_CONV_ASUM: if (tile_valid) { RZT_CALL(RZW_K_TAP9, RZW_EP_BN_RELU_ASUM, 0, 1) } break;
        case RZ_FAST_CONV_RES: if (tile_valid) { RZT_CALL(RZW_K_TAP9, RZW_EP_BN_RES_RELU, 0, 1) } break;
        case RZ_FAST_CONV1: if (tile_valid) { RZT_CALL(RZW_K_LIN1, RZW_EP_BIAS_POS, 0, 1) } break;
        case RZ_FAST_FC9_ELU:
          if (alone) { RZT_CALL(RZW_K_LIN9, RZW_EP_BIAS_ELU_TREE, wave, RZW_WAVES) }
          else if (wave == twave) { RZT_CALL(RZW_K_LIN9, RZW_EP_BIAS_ELU_TREE, 0, 1) }
          break;
        case RZ_FAST_FC1:
          if (alone && wa.fc1_valu) {   // (MZX_RZ_FC1_VALU=0: the matrix path, A/B)
            if (w_in_lds) rzt_fc1_valu(q, wchunks, reg, simg, (const f32x4*)(wlds + q.w_at), tid);
            else rzt_fc1_valu(q, wchunks, reg, simg, (const f32x4*)(a.weights + q.w_at), tid);
          } else if (alone) {
            RZT_CALL(RZW_K_LIN1, RZW_EP_BIAS_TREE, wave, RZW_WAVES)
          } else if (wave == twave) {
            RZT_CALL(RZW_K_LIN1, RZW_EP_BIAS_TREE, 0, 1)
          }
          break;
        default: rzt_scale(q, a, reg, scratch, rowaddr, tid, store_hidden ? hid : nullptr); break;
      }
      if (last) __syncthreads();
    }
#undef RZT_CALL
    RZT_PROF(3)

    // ---- decode (value on wave 0, reward on wave 1), expand, back-propagate (self_play.py:343-353)
    if (wave <= 1) {
      const float* lg = reg + a.out_off[wave];    // 0: value, 1: reward
      const float x = (F <= 2 * FUSED_ROW) ? row_decode2(sub < F ? lg[sub] : 0.f, sub + 16 < F ? lg[sub + 16] : 0.f, F, support, sub)
                                            : row_decode_wide(lg, F, support, sub);
      if (lane == 0) hand[4 + wave] = __builtin_bit_cast(int, x);
    }
    __syncthreads();
    if (wave == 0) {
      const float value = __builtin_bit_cast(float, hand[4]), reward = __builtin_bit_cast(float, hand[5]);
      const float* pl = reg + a.out_off[2];
      const bool in = sub < A;
      const float lgp = in ? pl[sub] : 0.f;
      const float m = row_max(in ? lgp : -MZX_INF);
      const float e = in ? mzx_expf(lgp - m) : 0.f;
      const float den = row_sum(e);
      fc2_expand<AW>(FT, wk.leaf, sub, in, (double)mzx_div(e, den));
      fc2_backprop<AW>(FT, rst, wk, lane_ops, sub, (double)value, (double)reward);
      wave_sync();
    }
    RZT_PROF(4)
  }
#undef RZT_PROF
  if (prof) for (int k = 0; k < 8; ++k) sa.prof[blockIdx.x * 8 + k] = pc[k];
  if (wave == 0) { wave_sync(); fc2_to_arena<AW>(FT, rst, t, sub); }
}

template <int AW, bool PROFILE>
inline int rz_tile_launch_k(const RzWaveArgs& wa, unsigned grid, size_t lds_bytes, stream_t stream) {
  static std::atomic<uint64_t> lds_attr_done{0};   // per instantiation, one bit per device
  if (const int ae = allow_large_lds((const void*)rz_tile_search_kernel<AW, PROFILE>, 160 * 1024, lds_attr_done)) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  hipLaunchKernelGGL((rz_tile_search_kernel<AW, PROFILE>), dim3(grid), dim3(RZW_WAVES * 64), lds_bytes, stream, wa);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("tile-per-wave search kernel launch failed: %s (grid %u, %zu bytes of LDS)", hipGetErrorString(e), grid, lds_bytes);
    return MZX_ERR_RUNTIME;
  }
  return MZX_OK;
}

// Plans the launch of rz_tile_search_kernel; false: not a configuration for it.  MZX_RZ_TILE=0: A/B knob.
inline bool rz_tile_plan(const mzx_search* s, const RzProgram& R, const RzArgs& base, RzWaveArgs& wa, unsigned& grid,
                         size_t& lds_bytes) {
  const RzGeometry& g = s->net->rz.g;
  const int A = s->p.num_actions;
  if (g.HW <= 16 || g.HW > 16 * RZW_WAVES || A > FUSED_ROW) return false;
  if (rz_env_int("MZX_RZ_TILE", 1) == 0) return false;
  for (int o = 0; o < R.n_ops; ++o) {     // every operator of a class this kernel has code for
    const int cls = rz_classify(R.ops[o], 0x7FFFFFFF);
    if (cls == RZ_FAST_NONE) return false;
    if (R.ops[o].kind == RZ_GEMM && ((R.ops[o].cout + 15) / 16 > 255 || R.ops[o].wchunks > 0x7FFF)) return false;   // fields of RzwOp::head
  }
  const int AW = A <= 4 ? 4 : 16;
  const int NN = s->p.num_nodes;
  wa.s.net = base;
  RzArgs& a = wa.s.net;
  a.T = 1;
  a.mpad = 64;
  a.small_floats = R.small_floats;
  const int rec_floats = (int)rz_trec_tree_floats(NN, AW);
  const int64_t fixed = rzt_layout(a.n_ops, a.small_floats, NN, a.Cs, a.tree_floats, rec_floats, 0).total;
  if (4 * fixed > RZ_LDS_BUDGET) return false;
  int64_t room = RZ_LDS_BUDGET / 4 - fixed;
  int used = 0;
  for (int o = 0; o < RZ_MAX_OPS; ++o) wa.wl[o] = -1;
  for (int o = 0; o < R.n_ops; ++o) {
    const RzOp& op = R.ops[o];
    if (op.kind != RZ_GEMM) continue;
    int shared = -1;
    for (int q = 0; q < o; ++q)
      if (R.ops[q].kind == RZ_GEMM && R.ops[q].w_off == op.w_off && wa.wl[q] >= 0) shared = wa.wl[q];
    if (shared >= 0) { wa.wl[o] = shared; continue; }
    const int nfl = ((op.cout + 15) / 16) * op.wchunks * 256;
    if (used + nfl <= room) { wa.wl[o] = used; used += nfl; }
  }
  wa.wl_floats = used;
  wa.fc1_valu = rz_env_int("MZX_RZ_FC1_VALU", 1) != 0 ? 1 : 0;
  grid = (unsigned)s->p.num_trees;
  lds_bytes = (size_t)4 * rzt_layout(a.n_ops, a.small_floats, NN, a.Cs, a.tree_floats, rec_floats, used).total;
  return true;
}

inline int rz_tile_launch(const RzWaveArgs& wa, unsigned grid, size_t lds_bytes, stream_t stream) {
  if (wa.s.prof) {
    return wa.s.p.num_actions <= 4 ? rz_tile_launch_k<4, true>(wa, grid, lds_bytes, stream)
                                   : rz_tile_launch_k<16, true>(wa, grid, lds_bytes, stream);
  }
  return wa.s.p.num_actions <= 4 ? rz_tile_launch_k<4, false>(wa, grid, lds_bytes, stream)
                                 : rz_tile_launch_k<16, false>(wa, grid, lds_bytes, stream);
}

#endif  // !MZX_HOSTCHECK

}  // namespace mzx
