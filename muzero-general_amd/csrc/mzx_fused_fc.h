// mzx_fused_fc.h -- the whole per-move search of a fully connected MuZero network
// in ONE gfx950 kernel launch (K5 of SURVEY.md section 8a): initial_inference, root
// expansion, and num_simulations x {select, recurrent_inference, expand,
// backpropagate} with every tree resident in LDS.
//
// Reference semantics: MCTS.run /root/reference/self_play.py:260-361 and
// MuZeroFullyConnectedNetwork models.py:80-195.  The tree arithmetic is the SAME
// inline code as the generic path (mzx_tree.h, proven bit-exact against the
// reference in lock-step); the fp32 reductions follow the canonical 16-lane
// order defined there, so this kernel and the generic operators agree bit for
// bit on the device (tests compare the exported trees).
//
// Mapping (wave64, one 256-thread workgroup per CU):
//   * a tree owns a ROW of 16 lanes (one DPP row); a wavefront carries 4 trees, a
//     workgroup 16 (fewer if LDS is short).  Wavefronts never synchronise with
//     each other after the staging barrier: each wave free-runs its 4 trees.
//   * tree statistics (binary64), child links, priors and per-node hidden states
//     live in that tree's LDS slab for the whole launch; HBM sees only the
//     inputs and the final visit counts.
//   * selection: lane s scores child slot s (one LDS read of A contiguous slots per
//     level), the argmax is a binary64 DPP butterfly + a wavefront ballot whose
//     16-bit row field yields the maximiser count / index (ties -> numpy-exact
//     draw from the tape).  Expansion: lane a writes slot a.  Back-propagation:
//     lane 0 (latency chain).
//   * network, two engines behind one interface:
//       - SmallNet<...>: every weight a lane needs sits in its REGISTERS for the
//         whole launch; a layer is K x {v_mov_dpp row_newbcast:k, v_fmac}: the
//         activations never leave the register file (single-hidden-layer MLPs up to
//         16 wide: the CartPole-class networks);
//       - LdsNet: any fully connected configuration up to 64-wide layers; weights
//         staged in LDS, activations exchanged through a per-tree LDS scratch.
#pragma once
#include <stdlib.h>

#include "mzx_search.h"

namespace mzx {

constexpr int FUSED_ROW = 16;          // lanes per tree
constexpr int FUSED_MAX_WIDTH = 64;    // widest layer LdsNet handles
constexpr int FUSED_SCRATCH = 5 * FUSED_MAX_WIDTH;  // floats of per-tree exchange scratch
constexpr int FUSED_PROF_WORDS = 16;

struct FusedMlp {
  int32_t n;                                 // number of Linear layers
  int32_t sizes[MZX_MAX_LAYERS + 2];         // widths, sizes[0] = input (incl. one-hot block)
  int32_t w[MZX_MAX_LAYERS + 1], b[MZX_MAX_LAYERS + 1];  // float offsets into the flat buffer
};

struct FusedFcArgs {
  SearchParams p;
  TreeLayout L;
  FusedMlp rep, dyn, rew, pol, val;
  int32_t in_size, E, n_params, trees_per_block;
  int32_t lds_tables, lds_weights, lds_trees, tree_stride, off_hidden, off_scratch;  // byte offsets in LDS
  const float* flat;
  const double* tables;   // global: pbc[N+1] then sqrt[N+1]
  char* export_trees;     // nullable: arena tree region (debug / parity export)
  float* export_hidden;   // nullable: arena hidden region
  uint32_t* prof;         // nullable: [B][FUSED_PROF_WORDS] cycle counters (profiling build)
  mzx_search_io io;
};

#ifndef MZX_HOSTCHECK

// ---------------------------------------------------------------------------
// wave / row primitives

// Orders this wave's LDS traffic across lanes (LDS is in-order per wave; this
// only stops the compiler from moving loads/stores across the exchange point).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int DPP_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // row_half_mirror: i <-> 7-i inside each 8
constexpr int DPP_MIRROR = 0x140;      // row_mirror: i <-> 15-i
constexpr int DPP_BCAST0 = 0x150;      // row_newbcast:0 (gfx90a+): every lane reads lane k of its row

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ float bcast(float v) { return dpp_f<DPP_BCAST0 + K>(v); }

// canonical butterfly (mzx_tree.h): xor 1, xor 2, half-mirror, mirror; own + partner
__device__ __forceinline__ float row_sum(float v) {
  v = v + dpp_f<DPP_XOR1>(v);
  v = v + dpp_f<DPP_XOR2>(v);
  v = v + dpp_f<DPP_HALF_MIRROR>(v);
  v = v + dpp_f<DPP_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, dpp_f<DPP_XOR1>(v));
  v = fmaxf(v, dpp_f<DPP_XOR2>(v));
  v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
  return v;
}
__device__ __forceinline__ float row_min(float v) {
  v = fminf(v, dpp_f<DPP_XOR1>(v));
  v = fminf(v, dpp_f<DPP_XOR2>(v));
  v = fminf(v, dpp_f<DPP_HALF_MIRROR>(v));
  v = fminf(v, dpp_f<DPP_MIRROR>(v));
  return v;
}
__device__ __forceinline__ double row_max_d(double v) {
  double o;
  o = dpp_d<DPP_XOR1>(v); v = (o > v) ? o : v;
  o = dpp_d<DPP_XOR2>(v); v = (o > v) ? o : v;
  o = dpp_d<DPP_HALF_MIRROR>(v); v = (o > v) ? o : v;
  o = dpp_d<DPP_MIRROR>(v); v = (o > v) ? o : v;
  return v;
}

// support_to_scalar (models.py:645-666), canonical lane order: this lane holds
// logits l0 (index sub) and l1 (index sub + 16); indices >= F are absent.
__device__ __forceinline__ float row_decode2(float l0, float l1, int F, int support, int sub) {
  const bool v0 = sub < F, v1 = sub + 16 < F;
  const float m = row_max(fmaxf(v0 ? l0 : -MZX_INF, v1 ? l1 : -MZX_INF));
  const float e0 = v0 ? mzx_expf(l0 - m) : 0.f, e1 = v1 ? mzx_expf(l1 - m) : 0.f;
  float dl = 0.f;          // canonical lane partial: 0 + e[sub] + e[sub+16]
  dl += e0;
  if (v1) dl += e1;
  const float den = row_sum(dl);
  float num = 0.f;
  if (v0) num += (float)(sub - support) * (e0 / den);
  if (v1) num += (float)(sub + 16 - support) * (e1 / den);
  return support_inverse_transform(row_sum(num));
}

// ---------------------------------------------------------------------------
// lane-parallel selection walk (tree_select of mzx_tree.h, lane s = child slot s)
struct RowSel { SelCtx c; int action; };

__device__ __forceinline__ RowSel row_select(const TreeRef& t, const SearchParams& p, const uint32_t* tape, int sub,
                                             int row_in_wave) {
  int node = 0, depth = 0, slot = 0;
  int vtp = t.to_play(0);
  const double mn = t.mm_min(), mx = t.mm_max();
  const int root_n = t.meta(TM_ROOT_N);
  int32_t tape_pos = t.meta(TM_TAPE_POS), flags = t.meta(TM_FLAGS), ties = t.meta(TM_TIE_DRAWS);
  for (;;) {
    ++depth;
    const int nc = (node == 0) ? root_n : p.num_actions;
    const int N = t.visit(node);
    const double pbc = p.pbc_table[N], sq = p.sqrt_table[N];
    const bool valid = sub < nc;
    const int s = valid ? sub : 0;
    const double sc = valid ? ucb_from(pbc, sq, t.slot_visit(node, s), t.prior(node, s), t.slot_q(node, s), mn, mx)
                            : -MZX_INF;
    const double best = row_max_d(sc);
    const unsigned long long bal = __ballot(valid && sc == best);
    const unsigned bits = (unsigned)(bal >> (row_in_wave * FUSED_ROW)) & 0xFFFFu;
    const int nbest = __popc(bits);
    if (nbest <= 1) {
      slot = nbest ? (__ffs(bits) - 1) : 0;
    } else {  // numpy.random.choice(ties): k-th maximiser in slot order
      ++ties;
      int k = tape_draw(tape, p.tape_words, tape_pos, flags, nbest);
      unsigned b = bits;
      for (; k > 0; --k) b &= b - 1;
      slot = __ffs(b) - 1;
    }
    vtp = (vtp + 1 < p.num_players) ? vtp + 1 : 0;
    const int nxt = t.child(node, slot);
    if (nxt < 0) break;
    node = nxt;
  }
  RowSel r;
  int leaf = t.meta(TM_N_NODES);
  if (leaf >= p.num_nodes) { flags |= TF_NODE_OVERFLOW; leaf = p.num_nodes - 1; }
  if (sub == 0) {
    t.meta(TM_TAPE_POS) = tape_pos;
    t.meta(TM_FLAGS) = flags;
    t.meta(TM_TIE_DRAWS) = ties;
  }
  r.c.parent = node; r.c.slot = slot; r.c.leaf = leaf; r.c.depth = depth; r.c.to_play = vtp;
  r.action = (node == 0) ? t.root_action(slot) : slot;
  return r;
}

// ---------------------------------------------------------------------------
// Network engines.  Interface (all calls are row-collective, `sub` = lane in row):
//   stage(a, smem, tid)   workgroup-wide staging before the one __syncthreads()
//   setup(a, smem, sub)   per-lane setup after it
//   initial(obs, h_out, scr, sub, value, policy)         models.py:172-190
//   recurrent(h_in, action, h_out, scr, sub, value, reward, policy)  models.py:192-195
// value/reward come back decoded (support_to_scalar) in every lane; `policy` is
// the logit of action `sub` (undefined for sub >= A); h_out[0..E) receives the
// min-max scaled state.

struct NetOut { float value, reward, policy; };

// ---- LdsNet: any FC configuration (widths <= 64) -------------------------------
struct LdsNet {
  const float* W;
  const FusedFcArgs* a;

  __device__ __forceinline__ void stage(const FusedFcArgs& args, char* smem, int tid) {
    float* w = (float*)(smem + args.lds_weights);
    for (int i = tid; i < args.n_params; i += blockDim.x) w[i] = args.flat[i];
  }
  __device__ __forceinline__ void setup(const FusedFcArgs& args, char* smem, int) {
    W = (const float*)(smem + args.lds_weights);
    a = &args;
  }

  // One MLP (models.py:630-642): x = K0 floats readable by every lane; result in `out`.
  // Same per-neuron operation order as LinearOp (mzx_ops.h).
  __device__ __forceinline__ void mlp(const FusedMlp& m, const float* x, float* tmp0, float* tmp1, float* out, int sub,
                                      int action, int onehot) const {
    for (int l = 0; l < m.n; ++l) {
      const int K = m.sizes[l], O = m.sizes[l + 1];
      const int Kx = (l == 0) ? K - onehot : K;
      const bool last = (l == m.n - 1);
      float* y = last ? out : ((l & 1) ? tmp1 : tmp0);
      for (int o = sub; o < O; o += FUSED_ROW) {
        const float* wr = W + m.w[l] + o * K;
        float acc = 0.f;
        for (int k = 0; k < Kx; ++k) acc = fmaf(x[k], wr[k], acc);
        if (l == 0 && onehot) acc += wr[Kx + action];
        acc += W[m.b[l] + o];
        y[o] = last ? acc : mzx_elu(acc);
      }
      wave_sync();
      x = y;
    }
  }
  __device__ __forceinline__ void scale(const float* x, float* y, int E, int sub) const {
    float lo = x[0], hi = x[0];
    for (int k = 1; k < E; ++k) { lo = fminf(lo, x[k]); hi = fmaxf(hi, x[k]); }
    float sc = hi - lo;
    if (sc < 1e-5f) sc += 1e-5f;
    for (int j = sub; j < E; j += FUSED_ROW) y[j] = (x[j] - lo) / sc;
    wave_sync();
  }
  __device__ __forceinline__ float decode(const float* lg, int sub) const {
    const int F = 2 * a->p.support_size + 1, S = a->p.support_size;
    float m = -MZX_INF;
    for (int i = sub; i < F; i += FUSED_ROW) m = fmaxf(m, lg[i]);
    m = row_max(m);
    float den = 0.f;
    for (int i = sub; i < F; i += FUSED_ROW) den += mzx_expf(lg[i] - m);
    den = row_sum(den);
    float num = 0.f;
    for (int i = sub; i < F; i += FUSED_ROW) num += (float)(i - S) * (mzx_expf(lg[i] - m) / den);
    return support_inverse_transform(row_sum(num));
  }
  __device__ __forceinline__ void heads(const float* h, float* scr, int sub, NetOut& o) const {
    float* s0 = scr; float* s1 = scr + FUSED_MAX_WIDTH; float* s2 = scr + 2 * FUSED_MAX_WIDTH;
    float* s3 = scr + 3 * FUSED_MAX_WIDTH;
    mlp(a->pol, h, s0, s1, s3, sub, 0, 0);
    mlp(a->val, h, s0, s1, s2, sub, 0, 0);
    o.value = decode(s2, sub);
    o.policy = s3[sub < a->p.num_actions ? sub : 0];
  }
  __device__ __forceinline__ void initial(const float* obs, float* h_out, float* scr, int sub, NetOut& o) const {
    float* s0 = scr; float* s1 = scr + FUSED_MAX_WIDTH; float* s4 = scr + 4 * FUSED_MAX_WIDTH;
    mlp(a->rep, obs, s0, s1, s4, sub, 0, 0);
    scale(s4, h_out, a->E, sub);
    heads(h_out, scr, sub, o);
  }
  __device__ __forceinline__ void recurrent(const float* h_in, int action, float* h_out, float* scr, int sub,
                                            NetOut& o) const {
    float* s0 = scr; float* s1 = scr + FUSED_MAX_WIDTH; float* s3 = scr + 3 * FUSED_MAX_WIDTH;
    float* s4 = scr + 4 * FUSED_MAX_WIDTH;
    mlp(a->dyn, h_in, s0, s1, s4, sub, action, a->p.num_actions);   // s4 <- next state (unscaled)
    mlp(a->rew, s4, s0, s1, s3, sub, 0, 0);                         // reward head reads the UNscaled state
    o.reward = decode(s3, sub);
    scale(s4, h_out, a->E, sub);
    heads(h_out, scr, sub, o);
  }
};

// ---- SmallNet: register-resident weights, DPP-broadcast activations -------------
template <int K, int I = 0>
__device__ __forceinline__ float fma_bcast(float x, const float (&w)[K], float acc) {
  if constexpr (I < K) {
    acc = fmaf(bcast<I>(x), w[I], acc);
    return fma_bcast<K, I + 1>(x, w, acc);
  } else {
    return acc;
  }
}
template <int K>
__device__ __forceinline__ void load_row(float (&w)[K], float& b, const float* flat, int woff, int boff, int o, int O) {
  const bool ok = o < O;
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = ok ? flat[woff + o * K + k] : 0.f;
  b = ok ? flat[boff + o] : 0.f;
}

// rep: IN -> E (no hidden layer); dyn: E+A -> HD -> E; rew: E -> HR -> F; pol: E -> HP -> A; val: E -> HV -> F
template <int IN, int E, int A, int F, int HD, int HR, int HP, int HV>
struct SmallNet {
  static_assert(IN <= 64 && E <= 16 && A <= 16 && F <= 32 && HD <= 16 && HR <= 16 && HP <= 16 && HV <= 16, "shape");
  float w_rep[IN], b_rep;
  float w_d1[E + A], b_d1, w_d2[HD], b_d2;
  float w_r1[E], b_r1, w_r2a[HR], b_r2a, w_r2b[HR], b_r2b;
  float w_p1[E], b_p1, w_p2[HP], b_p2;
  float w_v1[E], b_v1, w_v2a[HV], b_v2a, w_v2b[HV], b_v2b;
  int support;

  static bool matches(const FusedFcArgs& a) {
    const int Fa = 2 * a.p.support_size + 1;
    return a.in_size == IN && a.E == E && a.p.num_actions == A && Fa == F && a.rep.n == 1 && a.dyn.n == 2 &&
           a.dyn.sizes[1] == HD && a.rew.n == 2 && a.rew.sizes[1] == HR && a.pol.n == 2 && a.pol.sizes[1] == HP &&
           a.val.n == 2 && a.val.sizes[1] == HV;
  }

  __device__ __forceinline__ void stage(const FusedFcArgs&, char*, int) {}
  __device__ __forceinline__ void setup(const FusedFcArgs& a, char*, int sub) {
    const float* f = a.flat;
    support = a.p.support_size;
    load_row(w_rep, b_rep, f, a.rep.w[0], a.rep.b[0], sub, E);
    load_row(w_d1, b_d1, f, a.dyn.w[0], a.dyn.b[0], sub, HD);
    load_row(w_d2, b_d2, f, a.dyn.w[1], a.dyn.b[1], sub, E);
    load_row(w_r1, b_r1, f, a.rew.w[0], a.rew.b[0], sub, HR);
    load_row(w_r2a, b_r2a, f, a.rew.w[1], a.rew.b[1], sub, F);
    load_row(w_r2b, b_r2b, f, a.rew.w[1], a.rew.b[1], sub + 16, F);
    load_row(w_p1, b_p1, f, a.pol.w[0], a.pol.b[0], sub, HP);
    load_row(w_p2, b_p2, f, a.pol.w[1], a.pol.b[1], sub, A);
    load_row(w_v1, b_v1, f, a.val.w[0], a.val.b[0], sub, HV);
    load_row(w_v2a, b_v2a, f, a.val.w[1], a.val.b[1], sub, F);
    load_row(w_v2b, b_v2b, f, a.val.w[1], a.val.b[1], sub + 16, F);
  }

  // min-max scale of the state held one element per lane (lanes >= E hold junk)
  __device__ __forceinline__ float scale(float s, int sub) const {
    const bool in = sub < E;
    const float lo = row_min(in ? s : MZX_INF), hi = row_max(in ? s : -MZX_INF);
    float sc = hi - lo;
    if (sc < 1e-5f) sc += 1e-5f;
    return (s - lo) / sc;
  }
  // prediction heads from the scaled state (lane k holds element k)
  __device__ __forceinline__ void heads(float hn, int sub, NetOut& o) const {
    const float p1 = mzx_elu(fma_bcast<E>(hn, w_p1, 0.f) + b_p1);
    const float v1 = mzx_elu(fma_bcast<E>(hn, w_v1, 0.f) + b_v1);
    o.policy = fma_bcast<HP>(p1, w_p2, 0.f) + b_p2;
    const float va = fma_bcast<HV>(v1, w_v2a, 0.f) + b_v2a;
    const float vb = fma_bcast<HV>(v1, w_v2b, 0.f) + b_v2b;
    o.value = row_decode2(va, vb, F, support, sub);
  }
  __device__ __forceinline__ void initial(const float* obs, float* h_out, float*, int sub, NetOut& o) const {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < IN; ++k) acc = fmaf(obs[k], w_rep[k], acc);
    const float hn = scale(acc + b_rep, sub);
    if (sub < E) h_out[sub] = hn;
    heads(hn, sub, o);
  }
  __device__ __forceinline__ void recurrent(const float* h_in, int action, float* h_out, float*, int sub,
                                            NetOut& o) const {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) acc = fmaf(h_in[k], w_d1[k], acc);
    float col = w_d1[E];
#pragma unroll
    for (int x = 1; x < A; ++x) col = (action == x) ? w_d1[E + x] : col;
    const float d1 = mzx_elu((acc + col) + b_d1);
    const float s = fma_bcast<HD>(d1, w_d2, 0.f) + b_d2;          // next state, unscaled (lane k < E)
    const float r1 = mzx_elu(fma_bcast<E>(s, w_r1, 0.f) + b_r1);   // reward head reads the UNscaled state
    const float ra = fma_bcast<HR>(r1, w_r2a, 0.f) + b_r2a;
    const float rb = fma_bcast<HR>(r1, w_r2b, 0.f) + b_r2b;
    const float hn = scale(s, sub);
    if (sub < E) h_out[sub] = hn;
    o.reward = row_decode2(ra, rb, F, support, sub);
    heads(hn, sub, o);
  }
};

// ---------------------------------------------------------------------------
// the kernel

#define MZX_PROF(k)                                             \
  if (PROFILE) {                                                \
    const unsigned long long _t = __builtin_readcyclecounter(); \
    prof[k] += (uint32_t)(_t - t_last);                         \
    t_last = _t;                                                \
  }

template <class Net, bool PROFILE>
__global__ void __launch_bounds__(256) fused_fc_search_kernel(const FusedFcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int sub = tid & (FUSED_ROW - 1);
  const int row = tid / FUSED_ROW;                        // tree slot inside the workgroup
  const int row_in_wave = row & 3;
  const int tree = blockIdx.x * a.trees_per_block + row;   // global tree index
  const int A = a.p.num_actions, E = a.E;
  uint32_t prof[FUSED_PROF_WORDS];
  unsigned long long t_last = 0;
  if (PROFILE) {
    for (int k = 0; k < FUSED_PROF_WORDS; ++k) prof[k] = 0;
    t_last = __builtin_readcyclecounter();
  }

  // ---- stage tables (+ weights) into LDS: the only workgroup-wide barrier
  Net net;
  double* tables = (double*)(smem + a.lds_tables);
  const int ntab = 2 * (a.p.num_nodes + 1);
  for (int i = tid; i < ntab; i += blockDim.x) tables[i] = a.tables[i];
  net.stage(a, smem, tid);
  __syncthreads();
  if (row >= a.trees_per_block || tree >= a.p.num_trees) return;  // whole rows exit together
  net.setup(a, smem, sub);

  char* slab = smem + a.lds_trees + (size_t)row * a.tree_stride;
  TreeRef t;
  t.base = slab;
  t.L = a.L;
  float* hidden = (float*)(slab + a.off_hidden);   // [num_nodes][E]
  float* scr = (float*)(slab + a.off_scratch);     // FUSED_SCRATCH floats

  SearchParams p = a.p;
  p.pbc_table = tables;
  p.sqrt_table = tables + (a.p.num_nodes + 1);
  const uint32_t* tape = (const uint32_t*)a.io.d_tape + (size_t)tree * p.tape_words;
  MZX_PROF(0)

  // ---- initial_inference (models.py:172-190) + root expansion (self_play.py:286-314, :467-476)
  {
    NetOut o;
    net.initial(a.io.d_observation + (size_t)tree * a.in_size, hidden, scr, sub, o);
    const int32_t* lg = a.io.d_legal_actions + (size_t)tree * A;
    const double* nz = a.io.d_noise ? a.io.d_noise + (size_t)tree * A : nullptr;
    int nroot = 0;
    while (nroot < A && lg[nroot] >= 0) ++nroot;
    if (sub < A) scr[sub] = o.policy;
    wave_sync();
    const bool in = sub < nroot;
    const float l = in ? scr[lg[sub]] : -MZX_INF;   // logits gathered in the game's legal-action order
    const float m = row_max(l);
    const float e = in ? mzx_expf(l - m) : 0.f;
    const float den = row_sum(e);
    if (sub == 0) {
      tree_init_root_record(t, p, lg, a.io.d_to_play[tree], (double)support_inverse_transform(0.0f));
      if (a.io.d_root_predicted_value) a.io.d_root_predicted_value[tree] = (double)o.value;
    }
    if (sub < A) tree_init_slot(t, 0, sub, in ? root_noisy_prior((double)(e / den), nz, sub, p.exploration_fraction) : 0.0);
    wave_sync();
  }
  MZX_PROF(1)

  // ---- simulations (self_play.py:319-355)
  for (int sim = 0; sim < p.num_sims; ++sim) {
    const RowSel sel = row_select(t, p, tape, sub, row_in_wave);
    MZX_PROF(2)
    NetOut o;
    net.recurrent(hidden + sel.c.parent * E, sel.action, hidden + sel.c.leaf * E, scr, sub, o);
    MZX_PROF(3)
    // priors = fp32 softmax over the full action space (self_play.py:460-462), lane a = action a
    const bool in = sub < A;
    const float m = row_max(in ? o.policy : -MZX_INF);
    const float e = in ? mzx_expf(o.policy - m) : 0.f;
    const float den = row_sum(e);
    if (in) tree_init_slot(t, sel.c.leaf, sub, (double)(e / den));
    MZX_PROF(4)
    if (sub == 0) {
      tree_attach_leaf(t, p, sel.c, (double)o.reward);
      tree_backprop(t, p, sel.c, (double)o.value);
    }
    wave_sync();
    MZX_PROF(5)
  }

  // ---- results (FinalizeOp)
  if (sub == 0) {
    for (int x = 0; x < A; ++x) a.io.d_visit_counts[(size_t)tree * A + x] = 0;
    const int nroot = t.meta(TM_ROOT_N);
    for (int s = 0; s < nroot; ++s) a.io.d_visit_counts[(size_t)tree * A + t.root_action(s)] = t.slot_visit(0, s);
    const int vc = t.visit(0);
    a.io.d_root_value[tree] = (vc == 0) ? 0.0 : t.value_sum(0) / (double)vc;
    a.io.d_info[tree * 4 + 0] = t.meta(TM_MAX_DEPTH);
    a.io.d_info[tree * 4 + 1] = t.meta(TM_FLAGS);
    a.io.d_info[tree * 4 + 2] = t.meta(TM_TAPE_POS);
    a.io.d_info[tree * 4 + 3] = t.meta(TM_SUM_DEPTH);
  }
  if (a.export_trees) {  // parity / diagnose export: LDS slab -> arena (same layout as the generic path)
    const int words = (int)(a.L.tree_bytes / 4);
    uint32_t* dst = (uint32_t*)(a.export_trees + (size_t)tree * a.L.tree_bytes);
    const uint32_t* src = (const uint32_t*)slab;
    for (int i = sub; i < words; i += FUSED_ROW) dst[i] = src[i];
    float* hd = a.export_hidden + (size_t)tree * p.num_nodes * E;
    for (int i = sub; i < p.num_nodes * E; i += FUSED_ROW) hd[i] = hidden[i];
  }
  MZX_PROF(6)
  if (PROFILE && sub == 0 && a.prof)
    for (int k = 0; k < FUSED_PROF_WORDS; ++k) a.prof[(size_t)tree * FUSED_PROF_WORDS + k] = prof[k];
}

// the register-resident instantiations: CartPole-class networks (games/cartpole.py:11-113)
using SmallNetCartpole = SmallNet<4, 8, 2, 21, 16, 16, 16, 16>;

// ---------------------------------------------------------------------------
// host side

constexpr int FUSED_LDS_BUDGET = 160 * 1024;

struct FusedPlan {
  FusedFcArgs args;
  int lds_bytes = 0;
  int ok = 0;
  int small = 0;   // 1: SmallNetCartpole
};

inline bool fused_take_mlp(const std::vector<OpDesc>& prog, size_t& pos, FusedMlp& m, int in_width, int onehot) {
  // consecutive OP_LINEAR ops starting at pos; the MLP ends at its first non-ELU (output) layer
  m.n = 0;
  m.sizes[0] = in_width + onehot;
  while (pos < prog.size() && prog[pos].kind == OP_LINEAR) {
    const OpDesc& d = prog[pos];
    if (m.n >= MZX_MAX_LAYERS + 1) return false;
    if (d.w_stride != m.sizes[m.n]) return false;
    if (d.out_features > FUSED_MAX_WIDTH) return false;
    m.w[m.n] = (int32_t)d.w;
    m.b[m.n] = (int32_t)d.b;
    m.sizes[m.n + 1] = d.out_features;
    ++m.n;
    ++pos;
    if (!d.elu) break;
  }
  return m.n > 0;
}

inline FusedPlan fused_plan(const mzx_search* s, bool allow_small = true) {
  FusedPlan P;
  const mzx_net* net = s->net;
  if (!net || net->cfg.network != 0) return P;
  const int A = s->p.num_actions, E = (int)net->hidden_size;
  if (A > FUSED_ROW || E > FUSED_MAX_WIDTH || net->input_size > 4096) return P;
  if (2 * s->p.support_size + 1 > FUSED_MAX_WIDTH) return P;
  if (net->num_params * 4 > 64 * 1024) return P;
  FusedFcArgs& a = P.args;
  memset(&a, 0, sizeof(a));
  // recover the five MLPs from the operator programs (initial: rep, scale, pol, val;
  // recurrent: dyn, rew, scale, pol, val)
  size_t pos = 0;
  if (!fused_take_mlp(net->prog_initial, pos, a.rep, (int)net->input_size, 0)) return P;
  if (pos >= net->prog_initial.size() || net->prog_initial[pos].kind != OP_SCALE) return P;
  ++pos;
  if (!fused_take_mlp(net->prog_initial, pos, a.pol, E, 0)) return P;
  if (!fused_take_mlp(net->prog_initial, pos, a.val, E, 0)) return P;
  pos = 0;
  if (!fused_take_mlp(net->prog_recurrent, pos, a.dyn, E, A)) return P;
  if (!fused_take_mlp(net->prog_recurrent, pos, a.rew, E, 0)) return P;
  if (a.rep.sizes[a.rep.n] != E || a.dyn.sizes[a.dyn.n] != E || a.pol.sizes[a.pol.n] != A) return P;

  a.p = s->p;
  a.L = s->L;
  a.in_size = (int32_t)net->input_size;
  a.E = E;
  a.n_params = (int32_t)net->num_params;
  P.small = (allow_small && SmallNetCartpole::matches(a)) ? 1 : 0;
  auto al16 = [](int64_t x) { return (x + 15) & ~int64_t(15); };
  int64_t o = 0;
  a.lds_tables = (int32_t)o;  o += al16(int64_t(16) * (s->p.num_nodes + 1));
  a.lds_weights = (int32_t)o; o += P.small ? 0 : al16(int64_t(4) * net->num_params);
  a.lds_trees = (int32_t)o;
  a.off_hidden = (int32_t)al16(s->L.tree_bytes);
  a.off_scratch = (int32_t)(a.off_hidden + al16(int64_t(4) * s->p.num_nodes * E));
  a.tree_stride = (int32_t)(a.off_scratch + int64_t(4) * FUSED_SCRATCH);
  int tpb = 16;
  while (tpb >= 4 && o + int64_t(tpb) * a.tree_stride > FUSED_LDS_BUDGET) tpb /= 2;
  if (tpb < 4) return P;
  a.trees_per_block = tpb;
  P.lds_bytes = (int)(o + int64_t(tpb) * a.tree_stride);
  P.ok = 1;
  return P;
}

inline int fused_fc_supported(const mzx_search* s) { return fused_plan(s).ok; }

template <class Net, bool PROFILE>
inline int fused_launch(const FusedPlan& P, unsigned grid, stream_t stream) {
  static bool attr_set = false;  // one per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)fused_fc_search_kernel<Net, PROFILE>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BUDGET);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
    attr_set = true;
  }
  hipLaunchKernelGGL((fused_fc_search_kernel<Net, PROFILE>), dim3(grid), dim3(P.args.trees_per_block * FUSED_ROW),
                     (size_t)P.lds_bytes, stream, P.args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("fused kernel launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

// mode bits: 1 = fused, 2 = export trees to the arena, 4 = force LdsNet, 8 = cycle-profile build
inline int fused_fc_run(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream) {
  FusedPlan P = fused_plan(s, !(s->mode & 4));
  if (!P.ok) { set_error("fused search kernel does not support this configuration"); return MZX_ERR_INVALID; }
  int rc = ensure_tables(s, d_arena, stream);
  if (rc) return rc;
  P.args.flat = s->net->d_flat;
  P.args.tables = (const double*)((char*)d_arena + s->off_tables);
  P.args.io = *io;
  // the cycle-profile build parks its counters in the (otherwise unused) network workspace
  const bool profile = (s->mode & 8) != 0 &&
                       s->ws_floats * 4 >= int64_t(s->p.num_trees) * FUSED_PROF_WORDS * 4;
  if (s->mode & 2) {
    P.args.export_trees = (char*)d_arena + s->off_trees;
    P.args.export_hidden = (float*)((char*)d_arena + s->off_hidden);
  }
  if (profile) P.args.prof = (uint32_t*)((char*)d_arena + s->off_ws);
  const bool small = P.small != 0;
  const int tpb = P.args.trees_per_block;
  const unsigned grid = (unsigned)((s->p.num_trees + tpb - 1) / tpb);
  if (small) return profile ? fused_launch<SmallNetCartpole, true>(P, grid, stream)
                            : fused_launch<SmallNetCartpole, false>(P, grid, stream);
  return profile ? fused_launch<LdsNet, true>(P, grid, stream) : fused_launch<LdsNet, false>(P, grid, stream);
}

#endif  // !MZX_HOSTCHECK

}  // namespace mzx
