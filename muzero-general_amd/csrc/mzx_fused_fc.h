// mzx_fused_fc.h -- the whole per-move search of a fully connected MuZero network
// in ONE gfx950 kernel launch (K5 of SURVEY.md section 8a): initial_inference, root
// expansion, and num_simulations x {select, recurrent_inference, expand,
// backpropagate} with every tree resident in LDS.
//
// Reference semantics: MCTS.run /root/reference/self_play.py:260-361 and
// MuZeroFullyConnectedNetwork models.py:80-195.  The tree arithmetic is the SAME
// inline code as the generic path (mzx_tree.h, proven bit-exact against the
// reference in lock-step) or a lane-parallel rearrangement of it that performs
// the identical binary64 operations per node; the fp32 reductions follow the
// canonical 16-lane order defined there.  Hence this kernel and the generic
// operators agree bit for bit on the device (tests compare the exported trees).
//
// Mapping (wave64, one 256-thread workgroup per CU):
//   * a tree owns a ROW of 16 lanes (one DPP row); a wavefront carries 4 trees, a
//     workgroup 16 (fewer if LDS is short).  Wavefronts never synchronise with
//     each other after the staging barrier: each wave free-runs its 4 trees.
//   * tree statistics (binary64), child links, priors and per-node hidden states
//     live in that tree's LDS slab for the whole launch; per-tree scalars (min-max
//     bounds, node count, tape cursor) stay in registers.  HBM sees only the
//     inputs and the final visit counts.
//   * selection: lane s scores child slot s (one LDS read of A contiguous slots per
//     level); argmax = binary64 DPP butterfly + a wavefront ballot whose 16-bit
//     row field gives the maximiser count / index (ties -> numpy-exact draw from
//     the tape).  The winner's link, visit count and pb_c / sqrt table entries are
//     handed to the next level by DPP row broadcasts (ds_bpermute beyond 4
//     candidates), so a level costs ONE round trip to the tree's memory.  Lane d remembers the node at depth d of the walk.
//   * expansion: lane a writes child slot a.  Back-propagation: lane d updates the
//     path node at depth d; the discounted value chain (the only true recurrence)
//     runs as DPP-broadcast steps; min-max by a binary64 butterfly when it moves.
//   * network, two engines behind one interface:
//       - SmallNet<...>: every weight a lane needs sits in its REGISTERS for the
//         whole launch; a layer is K x v_fmac_f32_dpp row_newbcast:k -- activations
//         never leave the register file (single-hidden-layer MLPs up to 16 wide:
//         the CartPole-class networks);
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
template <int K>
__device__ __forceinline__ int bcast_i(int v) { return dpp_i<DPP_BCAST0 + K>(v); }
template <int K>
__device__ __forceinline__ double bcast_d(double v) { return dpp_d<DPP_BCAST0 + K>(v); }

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
// binary64 max / min over the first W lanes of the row (W = 2, 4 or 16; the other lanes hold
// the neutral element); exact, so the order of the butterfly does not matter
template <int W>
__device__ __forceinline__ double row_max_d(double v) {
  double o;
  o = dpp_d<DPP_XOR1>(v); v = (o > v) ? o : v;
  if constexpr (W > 2) { o = dpp_d<DPP_XOR2>(v); v = (o > v) ? o : v; }
  if constexpr (W > 4) {
    o = dpp_d<DPP_HALF_MIRROR>(v); v = (o > v) ? o : v;
    o = dpp_d<DPP_MIRROR>(v); v = (o > v) ? o : v;
  }
  return v;
}
__device__ __forceinline__ double row_min_d16(double v) {
  double o;
  o = dpp_d<DPP_XOR1>(v); v = (o < v) ? o : v;
  o = dpp_d<DPP_XOR2>(v); v = (o < v) ? o : v;
  o = dpp_d<DPP_HALF_MIRROR>(v); v = (o < v) ? o : v;
  o = dpp_d<DPP_MIRROR>(v); v = (o < v) ? o : v;
  return v;
}
// value held by lane `slot` (< W, row-uniform) of this row
template <int W>
__device__ __forceinline__ int pick_i(int v, int slot) {
  static_assert(W == 2 || W == 4, "DPP pick is unrolled for 2 or 4 candidates");
  int r = bcast_i<0>(v);
  r = (slot == 1) ? bcast_i<1>(v) : r;
  if constexpr (W > 2) {
    r = (slot == 2) ? bcast_i<2>(v) : r;
    r = (slot == 3) ? bcast_i<3>(v) : r;
  }
  return r;
}
template <int W>
__device__ __forceinline__ double pick_d(double v, int slot) {
  return __hiloint2double(pick_i<W>(__double2hiint(v), slot), pick_i<W>(__double2loint(v), slot));
}
// value held by lane `slot` (< 16, row-uniform) of this row, any width: one ds_bpermute (no memory access)
__device__ __forceinline__ int perm_i(int v, int slot, int row_in_wave) {
  return __builtin_amdgcn_ds_bpermute((row_in_wave * FUSED_ROW + slot) << 2, v);
}
__device__ __forceinline__ double perm_d(double v, int slot, int row_in_wave) {
  return __hiloint2double(perm_i(__double2hiint(v), slot, row_in_wave), perm_i(__double2loint(v), slot, row_in_wave));
}
__device__ __forceinline__ unsigned row_bits(unsigned long long ballot, int row_in_wave) {
  return (unsigned)(ballot >> (row_in_wave * FUSED_ROW)) & 0xFFFFu;
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
  if (v0) num += (float)(sub - support) * mzx_div(e0, den);
  if (v1) num += (float)(sub + 16 - support) * mzx_div(e1, den);
  return support_inverse_transform(row_sum(num));
}

// ---------------------------------------------------------------------------
// per-tree scalars kept in registers (row-uniform) for the whole launch
struct RowState {
  double mn, mx;                       // MinMaxStats
  int32_t n_nodes, tape_pos, flags, ties, max_depth, sum_depth, root_n, root_to_play;
};

__device__ __forceinline__ void store_state(const TreeRef& t, const RowState& st) {
  t.mm_min() = st.mn; t.mm_max() = st.mx;
  t.meta(TM_N_NODES) = st.n_nodes; t.meta(TM_TAPE_POS) = st.tape_pos; t.meta(TM_FLAGS) = st.flags;
  t.meta(TM_TIE_DRAWS) = st.ties; t.meta(TM_MAX_DEPTH) = st.max_depth; t.meta(TM_SUM_DEPTH) = st.sum_depth;
  t.meta(TM_ROOT_N) = st.root_n;
}

struct RowSel {
  SelCtx c;
  int action;
  // this lane's share of the search path: lane d holds the node at depth d (root = 0, new leaf = depth)
  int my_node, my_parent, my_pslot;
};

// Lane-parallel selection walk (tree_select of mzx_tree.h; lane s = child slot s).
template <int AW>
__device__ __forceinline__ RowSel row_select(const TreeRef& t, const SearchParams& p, const uint32_t* tape, int sub,
                                             int row_in_wave, int sim, RowState& st, int2* path = nullptr) {
  RowSel r;
  int node = 0, depth = 0, slot = 0;
  if (path && sub == 0) path[0] = make_int2(0, -1);      // (the whole path for row_backprop: walks deeper than a row's lanes)
  int vtp = st.root_to_play;
  int N = sim;  // every finished simulation visited the root once
  double pbc = p.pbc_table[N], sq = p.sqrt_table[N];
  r.my_node = 0; r.my_parent = -1; r.my_pslot = -1;
  // The rows of a wave walk trees of different depths: the loop is WAVE-uniform (it ends when every row
  // has reached its leaf) and a row that is done keeps executing with its state frozen by selects --
  // one straight-line body per level instead of a divergent loop with exec-mask bookkeeping.
  bool done = false;
  for (;;) {
    const int d1 = depth + 1;
    const int nc = (node == 0) ? st.root_n : p.num_actions;
    const bool valid = sub < nc;
    const int s = valid ? sub : 0;
    const int n = t.slot_visit(node, s);
    const int c = t.child(node, s);
    const double u = ucb_from(pbc, sq, n, t.prior(node, s), t.slot_q(node, s), st.mn, st.mx);
    const double sc = valid ? u : -MZX_INF;
    // tables of the child's visit count, in case this slot wins (off the critical path)
    const double pbc_c = p.pbc_table[n], sq_c = p.sqrt_table[n];
    const double best = row_max_d<AW>(sc);
    const unsigned bits = row_bits(__ballot(valid && sc == best), row_in_wave);
    const int nbest = __popc(bits);
    int sl = nbest ? (__ffs(bits) - 1) : 0;
    if (nbest > 1 && !done) {  // numpy.random.choice(ties): k-th maximiser in slot order (rare after the first level)
      ++st.ties;
      int k = tape_draw(tape, p.tape_words, st.tape_pos, st.flags, nbest);
      unsigned b = bits;
      for (; k > 0; --k) b &= b - 1;
      sl = __ffs(b) - 1;
    }
    int cw, n_w;
    double pbc_w, sq_w;
    if constexpr (AW <= 4) {
      cw = pick_i<AW>(c, sl); n_w = pick_i<AW>(n, sl); pbc_w = pick_d<AW>(pbc_c, sl); sq_w = pick_d<AW>(sq_c, sl);
    } else {
      cw = perm_i(c, sl, row_in_wave); n_w = perm_i(n, sl, row_in_wave);
      pbc_w = perm_d(pbc_c, sl, row_in_wave); sq_w = perm_d(sq_c, sl, row_in_wave);
    }
    // commit this level unless the row already stopped
    const bool act = !done;
    const bool mine = act && sub == d1;
    r.my_parent = mine ? node : r.my_parent;
    r.my_pslot = mine ? sl : r.my_pslot;
    r.my_node = mine ? cw : r.my_node;
    if (path && act && sub == 0) path[d1] = make_int2(cw, sl);
    const int nvtp = (vtp + 1 < p.num_players) ? vtp + 1 : 0;  // players turn by turn, :331-334
    depth = act ? d1 : depth;
    slot = act ? sl : slot;
    vtp = act ? nvtp : vtp;
    const bool go = act && cw >= 0;
    N = go ? n_w : N;
    pbc = go ? pbc_w : pbc;
    sq = go ? sq_w : sq;
    node = go ? cw : node;
    done = done || (cw < 0);
    if (__all(done)) break;
  }
  int leaf = st.n_nodes;
  if (leaf >= p.num_nodes) { st.flags |= TF_NODE_OVERFLOW; leaf = p.num_nodes - 1; }
  if (sub == depth) r.my_node = leaf;
  if (path && sub == 0) path[depth] = make_int2(leaf, slot);
  r.c.parent = node; r.c.slot = slot; r.c.leaf = leaf; r.c.depth = depth; r.c.to_play = vtp;
  r.action = (node == 0) ? t.root_action(slot) : slot;
  return r;
}

// Leaf attachment + back-propagation of one simulation (tree_attach_leaf + tree_backprop of
// mzx_tree.h), one path node per lane: identical binary64 operations per node, the value
// recurrence evaluated leaf -> root exactly in the reference's order.
template <int J>
__device__ __forceinline__ void chain_step(double r_eff, double disc, int depth, int sub, double& val, double& my_in) {
  if (J <= depth) {  // row-uniform
    const double rj = bcast_d<J>(r_eff);
    if (sub == J) my_in = val;
    val = rj + disc * val;
  }
  if constexpr (J > 1) chain_step<J - 1>(r_eff, disc, depth, sub, val, my_in);
}

// One chunk of sixteen path nodes of a walk deeper than a row (lane j <-> depth 16 c + j), as row_backprop's single chunk.
struct RowChunk { int n, par, ps, vc, tp; double rr, vs; bool active, is_leaf; };
__device__ __forceinline__ RowChunk row_chunk_load(const TreeRef& t, const int2* path, const RowSel& r, int c, int top, int sub,
                                                   int P, double reward) {
  RowChunk k;
  const int depth = r.c.depth, d = c * FUSED_ROW + sub, ld = (c == top) ? (depth & (FUSED_ROW - 1)) : FUSED_ROW - 1;
  k.active = c >= 0 && sub <= ld; k.is_leaf = k.active && d == depth;
  k.n = 0; k.par = -1; k.ps = -1; k.vc = 0; k.tp = r.c.to_play; k.rr = reward; k.vs = 0.0;
  if (k.active) {
    const int2 e = path[d];
    k.n = e.x; k.ps = e.y;
    if (d > 0) k.par = path[d - 1].x;
    if (!k.is_leaf) {
      k.rr = t.reward(k.n); k.vs = t.value_sum(k.n); k.vc = t.visit(k.n);
      if (P == 2) k.tp = t.to_play(k.n);
    }
  }
  return k;
}

__device__ __forceinline__ void row_backprop(const TreeRef& t, const SearchParams& p, const RowSel& r, int sub,
                                             int row_in_wave, double value, double reward, RowState& st,
                                             const int2* path = nullptr) {
  const int depth = r.c.depth, P = p.num_players;
  st.n_nodes = r.c.leaf + 1;
  if (depth > st.max_depth) st.max_depth = depth;
  st.sum_depth += depth;
  if (depth >= FUSED_ROW && path) {
    // A path longer than a row, written down level by level by the selection walk (`path`: (node, slot taken) per depth):
    // sixteen nodes at a time, the leaf's chunk first -- every node's statistics of a chunk requested together (and the
    // chunk above already on its way), the value recurrence through the chunk in the reference's order, one more step through
    // lane 0's node for the chunk above.  The reference constructor's gomoku weights dig 107-ply lines (400 plies at most):
    // the serial walk below chased 107 parent links through the arena, a quarter of a millisecond per simulation (round 6).
    const int top = depth / FUSED_ROW;
    double val = value, hi = -MZX_INF, lo = MZX_INF;
    RowChunk k = row_chunk_load(t, path, r, top, top, sub, P, reward);
    for (int c = top; c >= 0; --c) {      // (row-uniform)
      const RowChunk up = row_chunk_load(t, path, r, c - 1, top, sub, P, reward);
      const int ld = (c == top) ? (depth & (FUSED_ROW - 1)) : FUSED_ROW - 1;
      const bool same = (k.tp == r.c.to_play);
      const double r_eff = (P == 1 || !same) ? k.rr : -k.rr;
      double my_in = val;
      chain_step<FUSED_ROW - 1>(r_eff, p.discount, ld, sub, val, my_in);
      if (sub == 0) my_in = val;
      if (c > 0) val = bcast_d<0>(r_eff) + p.discount * val;      // through lane 0's node, for the chunk above
      double qv = 0.0;
      if (k.active) {
        const double vs2 = k.vs + ((P == 1 || same) ? my_in : -my_in);
        const int vc2 = k.vc + 1;
        const double mean = vs2 / (double)vc2;
        qv = k.rr + p.discount * ((P == 1) ? mean : -mean);
        t.value_sum(k.n) = vs2;
        t.visit(k.n) = vc2;
        if (k.par >= 0) { t.slot_visit(k.par, k.ps) = vc2; t.slot_q(k.par, k.ps) = qv; }
        if (k.is_leaf) {
          t.child(k.par, k.ps) = k.n;
          t.parent(k.n) = k.par; t.parent_slot(k.n) = k.ps; t.to_play(k.n) = r.c.to_play; t.reward(k.n) = reward;
        }
      }
      const double h2 = row_max_d<16>(k.active ? qv : -MZX_INF), l2 = row_min_d16(k.active ? qv : MZX_INF);
      hi = h2 > hi ? h2 : hi;
      lo = l2 < lo ? l2 : lo;
      k = up;
    }
    if (hi > st.mx) st.mx = hi;      // MinMaxStats.update over the path (pure min / max: order-free)
    if (lo < st.mn) st.mn = lo;
    return;
  }
  if (depth >= FUSED_ROW) {  // path longer than a row and no path record: serial walk on lane 0 (rare)
    if (sub == 0) {
      store_state(t, st);
      t.child(r.c.parent, r.c.slot) = r.c.leaf;
      t.parent(r.c.leaf) = r.c.parent; t.parent_slot(r.c.leaf) = r.c.slot; t.to_play(r.c.leaf) = r.c.to_play;
      t.reward(r.c.leaf) = reward; t.visit(r.c.leaf) = 0; t.value_sum(r.c.leaf) = 0.0;
      tree_backprop(t, p, r.c, value);
    }
    wave_sync();
    st.mn = t.mm_min(); st.mx = t.mm_max();
    return;
  }
  const bool active = sub <= depth, is_leaf = sub == depth;
  const int n = r.my_node;
  double rr = reward, vs = 0.0;
  int vc = 0, tp = r.c.to_play;
  if (active && !is_leaf) {
    rr = t.reward(n); vs = t.value_sum(n); vc = t.visit(n);
    if (P == 2) tp = t.to_play(n);
  }
  const bool same = (tp == r.c.to_play);
  const double r_eff = (P == 1 || !same) ? rr : -rr;  // value = (+-reward) + discount * value
  double val = value, my_in = value;
  chain_step<FUSED_ROW - 1>(r_eff, p.discount, depth, sub, val, my_in);
  if (sub == 0) my_in = val;
  double qv = 0.0;
  if (active) {
    const double vs2 = vs + ((P == 1 || same) ? my_in : -my_in);
    const int vc2 = vc + 1;
    const double mean = vs2 / (double)vc2;
    qv = rr + p.discount * ((P == 1) ? mean : -mean);
    t.value_sum(n) = vs2;
    t.visit(n) = vc2;
    if (sub > 0) { t.slot_visit(r.my_parent, r.my_pslot) = vc2; t.slot_q(r.my_parent, r.my_pslot) = qv; }
    if (is_leaf) {
      t.child(r.my_parent, r.my_pslot) = n;
      t.parent(n) = r.my_parent; t.parent_slot(n) = r.my_pslot; t.to_play(n) = r.c.to_play; t.reward(n) = reward;
    }
  }
  // MinMaxStats.update over the path (pure min / max: order-free)
  if (row_bits(__ballot(active && (qv > st.mx || qv < st.mn)), row_in_wave)) {
    const double hi = row_max_d<16>(active ? qv : -MZX_INF), lo = row_min_d16(active ? qv : MZX_INF);
    if (hi > st.mx) st.mx = hi;
    if (lo < st.mn) st.mn = lo;
  }
}

// ---------------------------------------------------------------------------
// Network engines.  Interface (all calls are row-collective, `sub` = lane in row):
//   stage(a, smem, tid)   workgroup-wide staging before the one __syncthreads()
//   setup(a, smem, sub)   per-lane setup after it
//   initial(obs, h_out, scr, sub, out)                      models.py:172-190
//   recurrent(h_in, action, h_out, scr, sub, out)           models.py:192-195
// value/reward come back decoded (support_to_scalar) in every lane; `policy` is
// the logit of action `sub` (undefined for sub >= A); h_out[0..E) receives the
// min-max scaled state.

struct NetOut { float value, reward, policy; };

// ---- LdsNet: any FC configuration (widths <= 64) -------------------------------
struct LdsNet {
  const float* W;
  const FusedFcArgs* a;

  // Weights go to LDS TRANSPOSED, [K][O] per layer at the flat buffer's own offsets: lane o of a row then reads
  // element k of ITS neuron at k * O + o -- consecutive lanes, consecutive words.  In the state_dict's [O][K]
  // order the 16 lanes of a row are K words apart: for K = 64 (games/lunarlander.py:69-74) all of them hit one
  // bank and every weight read is a 16-way conflict.
  __device__ __forceinline__ void stage_mlp(const FusedMlp& m, const float* flat, float* w, int tid) {
    for (int l = 0; l < m.n; ++l) {
      const int K = m.sizes[l], O = m.sizes[l + 1], base = m.w[l];
      for (int i = tid; i < O * K; i += blockDim.x) {
        const int o = i / K, k = i - o * K;
        w[base + k * O + o] = flat[base + i];
      }
      for (int i = tid; i < O; i += blockDim.x) w[m.b[l] + i] = flat[m.b[l] + i];
    }
  }
  __device__ __forceinline__ void stage(const FusedFcArgs& args, char* smem, int tid) {
    float* w = (float*)(smem + args.lds_weights);
    stage_mlp(args.rep, args.flat, w, tid);
    stage_mlp(args.dyn, args.flat, w, tid);
    stage_mlp(args.rew, args.flat, w, tid);
    stage_mlp(args.pol, args.flat, w, tid);
    stage_mlp(args.val, args.flat, w, tid);
  }
  __device__ __forceinline__ void setup(const FusedFcArgs& args, char* smem, int) {
    W = (const float*)(smem + args.lds_weights);
    a = &args;
  }

  // One Linear layer for the neurons o0, o0 + 16, ..., o0 + 16 (G - 1) of this lane (G = 1, 2 or 4 by the layer's
  // width): the input value x[k] is loaded once and feeds G independent fmaf chains, so wide layers (the 64-wide
  // hidden layers of games/lunarlander.py) cost a quarter of the x loads and their chains overlap.  Per neuron
  // the k order -- the sequential fmaf chain of LinearOp (mzx_ops.h) -- is unchanged.  Reads of neurons beyond O
  // stay inside LDS and their results are discarded.
  template <int G>
  __device__ __forceinline__ void layer(const float* wl, const float* bl, const float* x, float* y, int K, int Kx, int O,
                                        int sub, int action, bool onehot, bool last) const {
    for (int o0 = sub; o0 < O; o0 += G * FUSED_ROW) {
      const float* wr = wl + o0;      // transposed: element k of neuron o at k * O + o
      float acc[G];
#pragma unroll
      for (int j = 0; j < G; ++j) acc[j] = 0.f;
      int k = 0;
      constexpr int U = (G == 1) ? 8 : 4;   // k values per trip: U x loads + U * G w loads in flight
      for (; k + U <= Kx; k += U) {
        float xv[U], wv[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          xv[u] = x[k + u];
#pragma unroll
          for (int j = 0; j < G; ++j) wv[u][j] = wr[(k + u) * O + j * FUSED_ROW];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < G; ++j) acc[j] = fmaf(xv[u], wv[u][j], acc[j]);
      }
      for (; k < Kx; ++k) {
        const float xk = x[k];
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = fmaf(xk, wr[k * O + j * FUSED_ROW], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int o = o0 + j * FUSED_ROW;
        if (o < O) {
          float v = acc[j];
          if (onehot) v += wr[(Kx + action) * O + j * FUSED_ROW];
          v += bl[o];
          y[o] = last ? v : mzx_elu(v);
        }
      }
    }
  }

  // One MLP (models.py:630-642): x = K0 floats readable by every lane; result in `out`.
  __device__ __forceinline__ void mlp(const FusedMlp& m, const float* x, float* tmp0, float* tmp1, float* out, int sub,
                                      int action, int onehot) const {
    for (int l = 0; l < m.n; ++l) {
      const int K = m.sizes[l], O = m.sizes[l + 1];
      const int Kx = (l == 0) ? K - onehot : K;
      const bool last = (l == m.n - 1);
      float* y = last ? out : ((l & 1) ? tmp1 : tmp0);
      const float* wl = W + m.w[l];
      const float* bl = W + m.b[l];
      const bool oh = (l == 0 && onehot);
      if (O > 2 * FUSED_ROW) layer<4>(wl, bl, x, y, K, Kx, O, sub, action, oh, last);
      else if (O > FUSED_ROW) layer<2>(wl, bl, x, y, K, Kx, O, sub, action, oh, last);
      else layer<1>(wl, bl, x, y, K, Kx, O, sub, action, oh, last);
      wave_sync();
      x = y;
    }
  }
  __device__ __forceinline__ void scale(const float* x, float* y, int E, int sub) const {
    float lo = x[0], hi = x[0];
    for (int k = 1; k < E; ++k) { lo = fminf(lo, x[k]); hi = fmaxf(hi, x[k]); }
    float sc = hi - lo;
    if (sc < 1e-5f) sc += 1e-5f;
    for (int j = sub; j < E; j += FUSED_ROW) y[j] = mzx_div(x[j] - lo, sc);
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
    for (int i = sub; i < F; i += FUSED_ROW) num += (float)(i - S) * mzx_div(mzx_expf(lg[i] - m), den);
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
//
// acc += x[lane k of my row] * w[k] for k = 0..K-1, in that order, as one fused
// multiply-add each: v_fmac_f32_dpp with the row_newbcast:k operand swizzle, i.e.
// the sequential fmaf chain of LinearOp.  `x` is produced by plain VALU code right
// before the block, hence the leading 2 wait states (VALU write -> DPP read hazard,
// which hipcc does not track through an asm statement).
#define MZX_FD(ACC, W, K) "v_fmac_f32_dpp " ACC ", %[x], " W " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"

__device__ __forceinline__ float dot8_bcast(float x, const float (&w)[8]) {
  float acc = 0.f;
  asm("s_nop 1\n\t"
      MZX_FD("%[a]", "%[w0]", 0) MZX_FD("%[a]", "%[w1]", 1) MZX_FD("%[a]", "%[w2]", 2) MZX_FD("%[a]", "%[w3]", 3)
      MZX_FD("%[a]", "%[w4]", 4) MZX_FD("%[a]", "%[w5]", 5) MZX_FD("%[a]", "%[w6]", 6) MZX_FD("%[a]", "%[w7]", 7)
      : [a] "+v"(acc)
      : [x] "v"(x), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]),
        [w6] "v"(w[6]), [w7] "v"(w[7]));
  return acc;
}
__device__ __forceinline__ float dot16_bcast(float x, const float (&w)[16]) {
  float acc = 0.f;
  asm("s_nop 1\n\t"
      MZX_FD("%[a]", "%[w0]", 0) MZX_FD("%[a]", "%[w1]", 1) MZX_FD("%[a]", "%[w2]", 2) MZX_FD("%[a]", "%[w3]", 3)
      MZX_FD("%[a]", "%[w4]", 4) MZX_FD("%[a]", "%[w5]", 5) MZX_FD("%[a]", "%[w6]", 6) MZX_FD("%[a]", "%[w7]", 7)
      MZX_FD("%[a]", "%[w8]", 8) MZX_FD("%[a]", "%[w9]", 9) MZX_FD("%[a]", "%[w10]", 10) MZX_FD("%[a]", "%[w11]", 11)
      MZX_FD("%[a]", "%[w12]", 12) MZX_FD("%[a]", "%[w13]", 13) MZX_FD("%[a]", "%[w14]", 14) MZX_FD("%[a]", "%[w15]", 15)
      : [a] "+v"(acc)
      : [x] "v"(x), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]),
        [w6] "v"(w[6]), [w7] "v"(w[7]), [w8] "v"(w[8]), [w9] "v"(w[9]), [w10] "v"(w[10]), [w11] "v"(w[11]),
        [w12] "v"(w[12]), [w13] "v"(w[13]), [w14] "v"(w[14]), [w15] "v"(w[15]));
  return acc;
}
// two neurons of one layer sharing the input (two independent fmac chains interleaved)
__device__ __forceinline__ void dot16x2_bcast(float x, const float (&u)[16], const float (&v)[16], float& ru, float& rv) {
  float a = 0.f, b = 0.f;
#define MZX_FD2(K) MZX_FD("%[a]", "%[u" #K "]", K) MZX_FD("%[b]", "%[v" #K "]", K)
  asm("s_nop 1\n\t"
      MZX_FD2(0) MZX_FD2(1) MZX_FD2(2) MZX_FD2(3) MZX_FD2(4) MZX_FD2(5) MZX_FD2(6) MZX_FD2(7)
      : [a] "+v"(a), [b] "+v"(b)
      : [x] "v"(x), [u0] "v"(u[0]), [u1] "v"(u[1]), [u2] "v"(u[2]), [u3] "v"(u[3]), [u4] "v"(u[4]), [u5] "v"(u[5]),
        [u6] "v"(u[6]), [u7] "v"(u[7]), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]),
        [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]));
  asm("s_nop 1\n\t"
      MZX_FD2(8) MZX_FD2(9) MZX_FD2(10) MZX_FD2(11) MZX_FD2(12) MZX_FD2(13) MZX_FD2(14) MZX_FD2(15)
      : [a] "+v"(a), [b] "+v"(b)
      : [x] "v"(x), [u8] "v"(u[8]), [u9] "v"(u[9]), [u10] "v"(u[10]), [u11] "v"(u[11]), [u12] "v"(u[12]),
        [u13] "v"(u[13]), [u14] "v"(u[14]), [u15] "v"(u[15]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]),
        [v11] "v"(v[11]), [v12] "v"(v[12]), [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15]));
#undef MZX_FD2
  ru = a;
  rv = b;
}


// ---- interleaved chains ------------------------------------------------------------------------------------
// Several neurons' fmac chains issued round-robin in ONE asm statement: consecutive instructions belong to
// different accumulators, so none waits for its predecessor (a lone chain issues one dependent fmac every ~8
// cycles on a SIMD that holds a single wave).  Per chain the k order -- and hence every bit -- is unchanged.
// (An asm statement takes at most 30 operands: four K steps of five chains per statement.)
#define MZX_FD_X(ACC, X, W, K) "v_fmac_f32_dpp " ACC ", " X ", " W " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"

// first layers of the three towers that follow the dynamics state: r1 <- s ; p1, v1 <- hn  (8 inputs each)
__device__ __forceinline__ void dot8x3_bcast(float xs, float xh, const float (&wr)[8], const float (&wp)[8],
                                             const float (&wv)[8], float& r, float& p, float& v) {
  float a = 0.f, b = 0.f, c = 0.f;
#define MZX_STEP3(K, J) MZX_FD_X("%[a]", "%[xs]", "%[r" #J "]", K) MZX_FD_X("%[b]", "%[xh]", "%[p" #J "]", K) MZX_FD_X("%[c]", "%[xh]", "%[v" #J "]", K)
  asm("s_nop 1\n\t" MZX_STEP3(0, 0) MZX_STEP3(1, 1) MZX_STEP3(2, 2) MZX_STEP3(3, 3)
      : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c)
      : [xs] "v"(xs), [xh] "v"(xh), [r0] "v"(wr[0]), [r1] "v"(wr[1]), [r2] "v"(wr[2]), [r3] "v"(wr[3]),
        [p0] "v"(wp[0]), [p1] "v"(wp[1]), [p2] "v"(wp[2]), [p3] "v"(wp[3]),
        [v0] "v"(wv[0]), [v1] "v"(wv[1]), [v2] "v"(wv[2]), [v3] "v"(wv[3]));
  asm("s_nop 1\n\t" MZX_STEP3(4, 0) MZX_STEP3(5, 1) MZX_STEP3(6, 2) MZX_STEP3(7, 3)
      : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c)
      : [xs] "v"(xs), [xh] "v"(xh), [r0] "v"(wr[4]), [r1] "v"(wr[5]), [r2] "v"(wr[6]), [r3] "v"(wr[7]),
        [p0] "v"(wp[4]), [p1] "v"(wp[5]), [p2] "v"(wp[6]), [p3] "v"(wp[7]),
        [v0] "v"(wv[4]), [v1] "v"(wv[5]), [v2] "v"(wv[6]), [v3] "v"(wv[7]));
#undef MZX_STEP3
  r = a; p = b; v = c;
}

// output layers of the three towers: (ra, rb) <- r1 ; pl <- p1 ; (va, vb) <- v1  (16 inputs each)
__device__ __forceinline__ void dot16x5_bcast(float xr, float xp, float xv, const float (&ra)[16], const float (&rb)[16],
                                              const float (&pw)[16], const float (&va)[16], const float (&vb)[16],
                                              float& o_ra, float& o_rb, float& o_p, float& o_va, float& o_vb) {
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f, e = 0.f;
#define MZX_STEP5(K, J)                                                                                             \
  MZX_FD_X("%[a]", "%[xr]", "%[s" #J "]", K) MZX_FD_X("%[b]", "%[xr]", "%[t" #J "]", K) MZX_FD_X("%[c]", "%[xp]", "%[u" #J "]", K) \
  MZX_FD_X("%[d]", "%[xv]", "%[w" #J "]", K) MZX_FD_X("%[e]", "%[xv]", "%[z" #J "]", K)
#define MZX_BLOCK5(K0, K1, K2, K3, B)                                                                              \
  asm("s_nop 1\n\t" MZX_STEP5(K0, 0) MZX_STEP5(K1, 1) MZX_STEP5(K2, 2) MZX_STEP5(K3, 3)                             \
      : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [e] "+v"(e)                                             \
      : [xr] "v"(xr), [xp] "v"(xp), [xv] "v"(xv),                                                                    \
        [s0] "v"(ra[B]), [s1] "v"(ra[B + 1]), [s2] "v"(ra[B + 2]), [s3] "v"(ra[B + 3]),                               \
        [t0] "v"(rb[B]), [t1] "v"(rb[B + 1]), [t2] "v"(rb[B + 2]), [t3] "v"(rb[B + 3]),                               \
        [u0] "v"(pw[B]), [u1] "v"(pw[B + 1]), [u2] "v"(pw[B + 2]), [u3] "v"(pw[B + 3]),                               \
        [w0] "v"(va[B]), [w1] "v"(va[B + 1]), [w2] "v"(va[B + 2]), [w3] "v"(va[B + 3]),                               \
        [z0] "v"(vb[B]), [z1] "v"(vb[B + 1]), [z2] "v"(vb[B + 2]), [z3] "v"(vb[B + 3]))
  MZX_BLOCK5(0, 1, 2, 3, 0);
  MZX_BLOCK5(4, 5, 6, 7, 4);
  MZX_BLOCK5(8, 9, 10, 11, 8);
  MZX_BLOCK5(12, 13, 14, 15, 12);
#undef MZX_BLOCK5
#undef MZX_STEP5
  o_ra = a; o_rb = b; o_p = c; o_va = d; o_vb = e;
}

// two support_to_scalar decodes (value and reward heads) in one straight-line body: the two dependency chains
// -- max butterfly, exp, sum butterfly, expectation, sum butterfly, inverse transform -- interleave.
__device__ __forceinline__ void row_decode2x2(float a0, float a1, float b0, float b1, int F, int support, int sub,
                                              float& out_a, float& out_b) {
  const bool v0 = sub < F, v1 = sub + 16 < F;
  const float ma = row_max(fmaxf(v0 ? a0 : -MZX_INF, v1 ? a1 : -MZX_INF));
  const float mb = row_max(fmaxf(v0 ? b0 : -MZX_INF, v1 ? b1 : -MZX_INF));
  const float ea0 = v0 ? mzx_expf(a0 - ma) : 0.f, ea1 = v1 ? mzx_expf(a1 - ma) : 0.f;
  const float eb0 = v0 ? mzx_expf(b0 - mb) : 0.f, eb1 = v1 ? mzx_expf(b1 - mb) : 0.f;
  float da = 0.f, db = 0.f;      // canonical lane partial: 0 + e[sub] + e[sub + 16]
  da += ea0; db += eb0;
  if (v1) { da += ea1; db += eb1; }
  const float dena = row_sum(da), denb = row_sum(db);
  float na = 0.f, nb = 0.f;
  if (v0) { na += (float)(sub - support) * mzx_div(ea0, dena); nb += (float)(sub - support) * mzx_div(eb0, denb); }
  if (v1) { na += (float)(sub + 16 - support) * mzx_div(ea1, dena); nb += (float)(sub + 16 - support) * mzx_div(eb1, denb); }
  out_a = support_inverse_transform(row_sum(na));
  out_b = support_inverse_transform(row_sum(nb));
}

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
__device__ __forceinline__ float dot_bcast(float x, const float (&w)[K]) {
  if constexpr (K == 8) return dot8_bcast(x, w);
  else if constexpr (K == 16) return dot16_bcast(x, w);
  else return fma_bcast<K>(x, w, 0.f);
}
template <int K>
__device__ __forceinline__ void dot2_bcast(float x, const float (&u)[K], const float (&v)[K], float& ru, float& rv) {
  if constexpr (K == 16) dot16x2_bcast(x, u, v, ru, rv);
  else { ru = fma_bcast<K>(x, u, 0.f); rv = fma_bcast<K>(x, v, 0.f); }
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
    return mzx_div(s - lo, sc);
  }
  // prediction heads from the scaled state (lane k holds element k)
  __device__ __forceinline__ void heads(float hn, int sub, NetOut& o) const {
    const float p1 = mzx_elu(dot_bcast<E>(hn, w_p1) + b_p1);
    const float v1 = mzx_elu(dot_bcast<E>(hn, w_v1) + b_v1);
    o.policy = dot_bcast<HP>(p1, w_p2) + b_p2;
    float va, vb;
    dot2_bcast<HV>(v1, w_v2a, w_v2b, va, vb);
    o.value = row_decode2(va + b_v2a, vb + b_v2b, F, support, sub);
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
    const float s = dot_bcast<HD>(d1, w_d2) + b_d2;                // next state, unscaled (lane k < E)
    if constexpr (E == 8 && HR == 16 && HP == 16 && HV == 16) {
      // the three towers behind the dynamics state side by side (see dot8x3_bcast): same values, bit for bit
      const float hn = scale(s, sub);
      if (sub < E) h_out[sub] = hn;
      float r1, p1, v1;
      dot8x3_bcast(s, hn, w_r1, w_p1, w_v1, r1, p1, v1);          // the reward head reads the UNscaled state
      r1 = mzx_elu(r1 + b_r1); p1 = mzx_elu(p1 + b_p1); v1 = mzx_elu(v1 + b_v1);
      float ra, rb, pl, va, vb;
      dot16x5_bcast(r1, p1, v1, w_r2a, w_r2b, w_p2, w_v2a, w_v2b, ra, rb, pl, va, vb);
      o.policy = pl + b_p2;
      row_decode2x2(va + b_v2a, vb + b_v2b, ra + b_r2a, rb + b_r2b, F, support, sub, o.value, o.reward);
      return;
    }
    const float r1 = mzx_elu(dot_bcast<E>(s, w_r1) + b_r1);         // reward head reads the UNscaled state
    float ra, rb;
    dot2_bcast<HR>(r1, w_r2a, w_r2b, ra, rb);
    const float hn = scale(s, sub);
    if (sub < E) h_out[sub] = hn;
    o.reward = row_decode2(ra + b_r2a, rb + b_r2b, F, support, sub);
    heads(hn, sub, o);
  }
};

// ---------------------------------------------------------------------------
// the kernel

// Phase stamps of the profiling instantiations: outstanding LDS / memory operations are drained and the
// compiler may not move code across the stamp, so a phase is charged with its own latency (the profiling
// build is a few per cent slower than the product build for that reason).
#define MZX_PROF(k)                                             \
  if (PROFILE) {                                                \
    __builtin_amdgcn_sched_barrier(0);                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    const unsigned long long _t = __builtin_readcyclecounter(); \
    prof[k] += (uint32_t)(_t - t_last);                         \
    t_last = _t;                                                \
    __builtin_amdgcn_sched_barrier(0);                          \
  }

// AW = lanes that can hold a child slot in this instantiation: 2, 4 (DPP hand-off of the
// winner) or 16 (LDS re-read); chosen on the host from the action-space size.
template <class Net, int AW, bool PROFILE>
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
  RowState st;
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
    st.root_to_play = a.io.d_to_play[tree];
    if (sub == 0) {
      tree_init_root_record(t, p, lg, st.root_to_play, (double)support_inverse_transform(0.0f));
      if (a.io.d_root_predicted_value) a.io.d_root_predicted_value[tree] = (double)o.value;
    }
    if (sub < A) tree_init_slot(t, 0, sub, in ? root_noisy_prior((double)mzx_div(e, den), nz, sub, p.exploration_fraction) : 0.0);
    st.mn = MZX_INF; st.mx = -MZX_INF;
    st.n_nodes = 1; st.tape_pos = 0; st.flags = 0; st.ties = 0; st.max_depth = 0; st.sum_depth = 0; st.root_n = nroot;
    wave_sync();
  }
  MZX_PROF(1)

  // ---- simulations (self_play.py:319-355)
  for (int sim = 0; sim < p.num_sims; ++sim) {
    const RowSel sel = row_select<AW>(t, p, tape, sub, row_in_wave, sim, st);
    MZX_PROF(2)
    NetOut o;
    net.recurrent(hidden + sel.c.parent * E, sel.action, hidden + sel.c.leaf * E, scr, sub, o);
    MZX_PROF(3)
    // priors = fp32 softmax over the full action space (self_play.py:460-462), lane a = action a
    const bool in = sub < A;
    const float m = row_max(in ? o.policy : -MZX_INF);
    const float e = in ? mzx_expf(o.policy - m) : 0.f;
    const float den = row_sum(e);
    if (in) tree_init_slot(t, sel.c.leaf, sub, (double)mzx_div(e, den));
    MZX_PROF(4)
    row_backprop(t, p, sel, sub, row_in_wave, (double)o.value, (double)o.reward, st);
    wave_sync();
    MZX_PROF(5)
  }

  // ---- results (FinalizeOp)
  if (sub == 0) {
    store_state(t, st);
    for (int x = 0; x < A; ++x) a.io.d_visit_counts[(size_t)tree * A + x] = 0;
    for (int s = 0; s < st.root_n; ++s) a.io.d_visit_counts[(size_t)tree * A + t.root_action(s)] = t.slot_visit(0, s);
    const int vc = t.visit(0);
    a.io.d_root_value[tree] = (vc == 0) ? 0.0 : t.value_sum(0) / (double)vc;
    a.io.d_info[tree * 4 + 0] = st.max_depth;
    a.io.d_info[tree * 4 + 1] = st.flags;
    a.io.d_info[tree * 4 + 2] = st.tape_pos;
    a.io.d_info[tree * 4 + 3] = st.sum_depth;
  }
  if (a.export_trees) {  // parity / diagnose export: LDS slab -> arena (same layout as the generic path)
    wave_sync();
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

// The first-generation whole-search kernel itself (per-level UCB evaluation) is kept for A/B measurements only: it is
// launched in instrumented builds (MZX_CXXFLAGS=-DMZX_EXPERIMENT); the product library does not carry its 16
// instantiations (round 6).  What fc2_search_kernel shares with it -- the network engines, the row functions, the plan --
// is above.
#ifdef MZX_EXPERIMENT
template <class Net, int AW, bool PROFILE>
inline int fused_launch(const FusedPlan& P, unsigned grid, stream_t stream) {
  static std::atomic<uint64_t> lds_attr_done{0};   // per instantiation, one bit per device
  if (const int ae = allow_large_lds((const void*)fused_fc_search_kernel<Net, AW, PROFILE>, FUSED_LDS_BUDGET, lds_attr_done)) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  hipLaunchKernelGGL((fused_fc_search_kernel<Net, AW, PROFILE>), dim3(grid),
                     dim3(P.args.trees_per_block * FUSED_ROW), (size_t)P.lds_bytes, stream, P.args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("fused kernel launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

template <class Net, bool PROFILE>
inline int fused_launch_aw(const FusedPlan& P, unsigned grid, stream_t stream) {
  const int A = P.args.p.num_actions;
  if (A <= 2) return fused_launch<Net, 2, PROFILE>(P, grid, stream);
  if (A <= 4) return fused_launch<Net, 4, PROFILE>(P, grid, stream);
  return fused_launch<Net, 16, PROFILE>(P, grid, stream);
}

#endif  // MZX_EXPERIMENT

// mode bits: 1 = fused, 2 = export trees to the arena, 4 = force LdsNet, 8 = cycle-profile build
inline int fused_fc_run(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream) {
#ifndef MZX_EXPERIMENT
  (void)io; (void)d_arena; (void)stream; (void)s;
  set_error("the first-generation fully connected kernel (mode flag 16) is built into instrumented libraries only (-DMZX_EXPERIMENT)");
  return MZX_ERR_INVALID;
#else
  FusedPlan P = fused_plan(s, !(s->mode & 4));
  if (!P.ok) { set_error("fused search kernel does not support this configuration"); return MZX_ERR_INVALID; }
  int rc = ensure_tables(s, d_arena, stream);
  if (rc) return rc;
  P.args.flat = s->net->d_flat;
  P.args.tables = s->d_tables;
  P.args.io = *io;
  // the cycle-profile build parks its counters in the (otherwise unused) network workspace
  const bool profile = (s->mode & 8) != 0 &&
                       s->ws_floats * 4 >= int64_t(s->p.num_trees) * FUSED_PROF_WORDS * 4;
  if (s->mode & 2) {
    P.args.export_trees = (char*)d_arena + s->off_trees;
    P.args.export_hidden = (float*)((char*)d_arena + s->off_hidden);
  }
  if (profile) P.args.prof = (uint32_t*)((char*)d_arena + s->off_ws);
  const int tpb = P.args.trees_per_block;
  const unsigned grid = (unsigned)((s->p.num_trees + tpb - 1) / tpb);
  if (P.small) {
    // SmallNetCartpole has A = 2
    return profile ? fused_launch<SmallNetCartpole, 2, true>(P, grid, stream)
                   : fused_launch<SmallNetCartpole, 2, false>(P, grid, stream);
  }
  return profile ? fused_launch_aw<LdsNet, true>(P, grid, stream) : fused_launch_aw<LdsNet, false>(P, grid, stream);
#endif
}

#endif  // !MZX_HOSTCHECK

}  // namespace mzx
