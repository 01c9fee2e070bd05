// mzx_fused_fc.h -- the whole per-move search of a fully connected MuZero network
// in ONE gfx950 kernel launch (K5 of SURVEY.md section 8a): initial_inference, root
// expansion, and num_simulations x {select, recurrent_inference, expand,
// backpropagate} with every tree resident in LDS.
//
// Reference semantics: MCTS.run /root/reference/self_play.py:260-361 and
// MuZeroFullyConnectedNetwork models.py:80-195.  The tree arithmetic is the SAME
// inline code as the generic path (mzx_tree.h, proven bit-exact against the
// reference in lock-step); only where it executes changes.
//
// Mapping (wave64, one 256-thread workgroup per CU):
//   * a tree owns a ROW of 16 lanes; a wavefront carries 4 trees, a workgroup
//     TPB = 16 trees (fewer if LDS is short).  Wavefronts never synchronise with
//     each other after the weight preload: each wave free-runs its 4 trees.
//   * tree statistics (binary64), child links, priors and per-node hidden states
//     live in that tree's LDS slab for the whole launch; weights are staged to
//     LDS once per workgroup and shared by its trees; HBM sees only the inputs
//     and the final visit counts.
//   * the latency chain select -> expand/backprop runs on lane 0 of the row
//     (exec-masked; identical cost to running it on all lanes); the network runs
//     on all 16 lanes: lane j computes neurons j, j+16, ... of each layer, layer
//     inputs are exchanged through a per-tree LDS scratch vector, and the
//     softmax / support decode reductions use 16-lane butterflies.
#pragma once
#include "mzx_search.h"

namespace mzx {

constexpr int FUSED_ROW = 16;          // lanes per tree
constexpr int FUSED_MAX_WIDTH = 64;    // widest layer / action space the kernel handles
constexpr int FUSED_SCRATCH = 5 * FUSED_MAX_WIDTH;  // floats of per-tree exchange scratch

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
  mzx_search_io io;
};

#ifndef MZX_HOSTCHECK

// Orders this wave's LDS traffic across lanes (LDS is in-order per wave; this
// only stops the compiler from moving loads/stores across the exchange point).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 8, FUSED_ROW));
  v = fmaxf(v, __shfl_xor(v, 4, FUSED_ROW));
  v = fmaxf(v, __shfl_xor(v, 2, FUSED_ROW));
  v = fmaxf(v, __shfl_xor(v, 1, FUSED_ROW));
  return v;
}
__device__ __forceinline__ float row_sum(float v) {
  v += __shfl_xor(v, 8, FUSED_ROW);
  v += __shfl_xor(v, 4, FUSED_ROW);
  v += __shfl_xor(v, 2, FUSED_ROW);
  v += __shfl_xor(v, 1, FUSED_ROW);
  return v;
}

// One MLP (models.py:630-642) for the tree of this row.  x: K0 floats readable by
// every lane of the row (LDS or global); the result is left in `out` (LDS).
// Same per-neuron operation order as LinearOp (mzx_ops.h).
__device__ __forceinline__ void row_mlp(const FusedMlp& m, const float* W, const float* x, float* tmp0, float* tmp1,
                                        float* out, int sub, int action, int onehot) {
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

// Min-max scale of the E-vector at `x` into `y` (models.py:136-145); every lane
// scans the whole vector (broadcast reads), lane j < E writes element j.
__device__ __forceinline__ void row_scale(const float* x, float* y, int E, int sub) {
  float lo = x[0], hi = x[0];
  for (int k = 1; k < E; ++k) { lo = fminf(lo, x[k]); hi = fmaxf(hi, x[k]); }
  float scale = hi - lo;
  if (scale < 1e-5f) scale += 1e-5f;
  for (int j = sub; j < E; j += FUSED_ROW) y[j] = (x[j] - lo) / scale;
  wave_sync();
}

// support_to_scalar (models.py:645-666) of the F logits at `lg`, cooperatively.
__device__ __forceinline__ float row_support_to_scalar(const float* lg, int F, int support, int sub) {
  float m = -MZX_INF;
  for (int i = sub; i < F; i += FUSED_ROW) m = fmaxf(m, lg[i]);
  m = row_max(m);
  float den = 0.f;
  for (int i = sub; i < F; i += FUSED_ROW) den += mzx_expf(lg[i] - m);
  den = row_sum(den);
  float num = 0.f;
  for (int i = sub; i < F; i += FUSED_ROW) num += (float)(i - support) * (mzx_expf(lg[i] - m) / den);
  num = row_sum(num);
  return support_inverse_transform(num);
}

__global__ void __launch_bounds__(256) fused_fc_search_kernel(const FusedFcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int sub = tid & (FUSED_ROW - 1);
  const int row = tid / FUSED_ROW;                       // tree slot inside the workgroup
  const int tree = blockIdx.x * a.trees_per_block + row;  // global tree index
  const int A = a.p.num_actions, E = a.E, F = 2 * a.p.support_size + 1;

  // ---- stage tables + weights into LDS (whole workgroup), the only block-wide barrier
  double* tables = (double*)(smem + a.lds_tables);
  float* W = (float*)(smem + a.lds_weights);
  const int ntab = 2 * (a.p.num_nodes + 1);
  for (int i = tid; i < ntab; i += blockDim.x) tables[i] = a.tables[i];
  for (int i = tid; i < a.n_params; i += blockDim.x) W[i] = a.flat[i];
  __syncthreads();
  if (row >= a.trees_per_block || tree >= a.p.num_trees) return;  // whole rows exit together

  char* slab = smem + a.lds_trees + (size_t)row * a.tree_stride;
  TreeRef t;
  t.base = slab;
  t.L = a.L;
  float* hidden = (float*)(slab + a.off_hidden);     // [num_nodes][E]
  float* scr = (float*)(slab + a.off_scratch);       // 5 x FUSED_MAX_WIDTH floats
  float* s0 = scr;                                   // MLP ping
  float* s1 = scr + FUSED_MAX_WIDTH;                 // MLP pong
  float* s2 = scr + 2 * FUSED_MAX_WIDTH;             // value logits / unscaled root state
  float* s3 = scr + 3 * FUSED_MAX_WIDTH;             // reward / policy logits
  float* s4 = scr + 4 * FUSED_MAX_WIDTH;             // unscaled next state

  SearchParams p = a.p;
  p.pbc_table = tables;
  p.sqrt_table = tables + (a.p.num_nodes + 1);
  const uint32_t* tape = a.io.d_tape ? (const uint32_t*)a.io.d_tape + (size_t)tree * p.tape_words : nullptr;

  // ---- initial_inference (models.py:172-190) + root expansion (self_play.py:286-314)
  {
    const float* obs = a.io.d_observation + (size_t)tree * a.in_size;
    row_mlp(a.rep, W, obs, s0, s1, s2, sub, 0, 0);       // s2 <- encoded state (unscaled)
    row_scale(s2, hidden, E, sub);                        // node 0 hidden state
    row_mlp(a.pol, W, hidden, s0, s1, s3, sub, 0, 0);     // s3 <- policy logits
    row_mlp(a.val, W, hidden, s0, s1, s2, sub, 0, 0);     // s2 <- value logits
    const float v0 = row_support_to_scalar(s2, F, p.support_size, sub);
    if (sub == 0) {
      const int32_t* lg = a.io.d_legal_actions + (size_t)tree * A;
      const double* nz = a.io.d_noise ? a.io.d_noise + (size_t)tree * A : nullptr;
      int nroot = 0;
      while (nroot < A && lg[nroot] >= 0) ++nroot;
      const SoftmaxStats st = softmax_stats(nroot, [&](int s) { return s3[lg[s]]; });
      tree_init_root(t, p, lg, a.io.d_to_play[tree], (double)support_inverse_transform(0.0f),
                     [&](int s) { return (double)(mzx_expf(s3[lg[s]] - st.m) / st.den); }, nz);
      if (a.io.d_root_predicted_value) a.io.d_root_predicted_value[tree] = (double)v0;
    }
    wave_sync();
  }

  // ---- simulations (self_play.py:319-355)
  for (int sim = 0; sim < p.num_sims; ++sim) {
    if (sub == 0) tree_select(t, p, tape);
    wave_sync();
    const int parent = t.meta(TM_CUR_PARENT), slot = t.meta(TM_CUR_SLOT), leaf = t.meta(TM_CUR_LEAF);
    const int action = (parent == 0) ? t.root_action(slot) : slot;

    // recurrent_inference (models.py:147-169, :192-195)
    row_mlp(a.dyn, W, hidden + parent * E, s0, s1, s4, sub, action, A);   // s4 <- next state (unscaled)
    row_mlp(a.rew, W, s4, s0, s1, s3, sub, 0, 0);                         // s3 <- reward logits
    const float reward = row_support_to_scalar(s3, F, p.support_size, sub);
    float* hnew = hidden + leaf * E;
    row_scale(s4, hnew, E, sub);                                          // leaf hidden state
    row_mlp(a.pol, W, hnew, s0, s1, s3, sub, 0, 0);                       // s3 <- policy logits
    row_mlp(a.val, W, hnew, s0, s1, s2, sub, 0, 0);                       // s2 <- value logits
    const float value = row_support_to_scalar(s2, F, p.support_size, sub);

    // priors = fp32 softmax over the full action space (self_play.py:460-462)
    float m = -MZX_INF;
    for (int i = sub; i < A; i += FUSED_ROW) m = fmaxf(m, s3[i]);
    m = row_max(m);
    float den = 0.f;
    for (int i = sub; i < A; i += FUSED_ROW) den += mzx_expf(s3[i] - m);
    den = row_sum(den);
    if (sub == 0) {
      tree_expand_backprop(t, p, (double)value, (double)reward,
                           [&](int s) { return (double)(mzx_expf(s3[s] - m) / den); });
    }
    wave_sync();
  }

  // ---- results (FinalizeOp)
  if (sub == 0) {
    for (int x = 0; x < A; ++x) a.io.d_visit_counts[(size_t)tree * A + x] = 0;
    const int nroot = t.meta(TM_ROOT_N);
    for (int s = 0; s < nroot; ++s) {
      const int c = t.child(0, s);
      a.io.d_visit_counts[(size_t)tree * A + t.root_action(s)] = (c >= 0) ? t.visit(c) : 0;
    }
    const int vc = t.visit(0);
    a.io.d_root_value[tree] = (vc == 0) ? 0.0 : t.value_sum(0) / (double)vc;
    a.io.d_info[tree * 4 + 0] = t.meta(TM_MAX_DEPTH);
    a.io.d_info[tree * 4 + 1] = t.meta(TM_FLAGS);
    a.io.d_info[tree * 4 + 2] = t.meta(TM_TAPE_POS);
    a.io.d_info[tree * 4 + 3] = t.meta(TM_SUM_DEPTH);
  }
}

// ---------------------------------------------------------------------------
// host side

constexpr int FUSED_LDS_BUDGET = 160 * 1024;

struct FusedPlan {
  FusedFcArgs args;
  int lds_bytes = 0;
  int ok = 0;
};

inline bool fused_take_mlp(const mzx_net* net, const std::vector<OpDesc>& prog, size_t& pos, FusedMlp& m, int in_width,
                           int onehot) {
  // consecutive OP_LINEAR ops starting at pos whose first layer has the expected input width
  m.n = 0;
  m.sizes[0] = in_width + onehot;
  while (pos < prog.size() && prog[pos].kind == OP_LINEAR) {
    const OpDesc& d = prog[pos];
    if (m.n >= MZX_MAX_LAYERS + 1) return false;
    if (d.w_stride != m.sizes[m.n]) break;  // next MLP starts
    m.w[m.n] = (int32_t)d.w;
    m.b[m.n] = (int32_t)d.b;
    m.sizes[m.n + 1] = d.out_features;
    if (d.out_features > FUSED_MAX_WIDTH) return false;
    ++m.n;
    ++pos;
    if (!d.elu) break;  // output layer of this MLP
  }
  return m.n > 0;
}

inline FusedPlan fused_plan(const mzx_search* s) {
  FusedPlan P;
  const mzx_net* net = s->net;
  if (!net || net->cfg.network != 0) return P;
  const int A = s->p.num_actions, E = (int)net->hidden_size;
  if (A > FUSED_MAX_WIDTH || E > FUSED_MAX_WIDTH) return P;
  if (net->num_params * 4 > 64 * 1024) return P;
  FusedFcArgs& a = P.args;
  memset(&a, 0, sizeof(a));
  // recover the five MLPs from the operator programs (initial: rep, scale, pol, val;
  // recurrent: dyn, rew, scale, pol, val)
  size_t pos = 0;
  if (!fused_take_mlp(net, net->prog_initial, pos, a.rep, (int)net->input_size, 0)) return P;
  if (pos >= net->prog_initial.size() || net->prog_initial[pos].kind != OP_SCALE) return P;
  ++pos;
  if (!fused_take_mlp(net, net->prog_initial, pos, a.pol, E, 0)) return P;
  if (!fused_take_mlp(net, net->prog_initial, pos, a.val, E, 0)) return P;
  pos = 0;
  if (!fused_take_mlp(net, net->prog_recurrent, pos, a.dyn, E, A)) return P;
  if (!fused_take_mlp(net, net->prog_recurrent, pos, a.rew, E, 0)) return P;
  if (a.rep.sizes[a.rep.n] != E || a.dyn.sizes[a.dyn.n] != E || a.pol.sizes[a.pol.n] != A) return P;
  if (2 * s->p.support_size + 1 > FUSED_MAX_WIDTH) return P;

  a.p = s->p;
  a.L = s->L;
  a.in_size = (int32_t)net->input_size;
  a.E = E;
  a.n_params = (int32_t)net->num_params;
  auto al16 = [](int64_t x) { return (x + 15) & ~int64_t(15); };
  int64_t o = 0;
  a.lds_tables = (int32_t)o;  o += al16(int64_t(16) * (s->p.num_nodes + 1));
  a.lds_weights = (int32_t)o; o += al16(int64_t(4) * net->num_params);
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

inline int fused_fc_run(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream) {
  FusedPlan P = fused_plan(s);
  if (!P.ok) { set_error("fused search kernel does not support this configuration"); return MZX_ERR_INVALID; }
  int rc = ensure_tables(s, d_arena, stream);
  if (rc) return rc;
  P.args.flat = s->net->d_flat;
  P.args.tables = (const double*)((char*)d_arena + s->off_tables);
  P.args.io = *io;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)fused_fc_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       FUSED_LDS_BUDGET);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
    attr_set = true;
  }
  const int tpb = P.args.trees_per_block;
  const unsigned grid = (unsigned)((s->p.num_trees + tpb - 1) / tpb);
  hipLaunchKernelGGL(fused_fc_search_kernel, dim3(grid), dim3(tpb * FUSED_ROW), (size_t)P.lds_bytes, stream, P.args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("fused kernel launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

#endif  // !MZX_HOSTCHECK

}  // namespace mzx
