// mzx_row_search.h -- per-simulation tree kernels for networks that run layer by layer (the streamed MFMA engine,
// mzx_batched.hip): select and expand + back-propagate of MCTS.run (/root/reference/self_play.py:319-355) with ONE
// 16-LANE ROW PER TREE instead of the generic path's one thread per tree.
//
// The generic operators (SelectOp / ExpandBackpropOp, mzx_ops.h) walk a tree with a single lane: with the 121
// actions of games/gomoku.py a level of the walk is 121 dependent binary64 UCB evaluations, and a shard of 512
// trees occupies eight wavefronts of the chip -- measured 3.3 ms per simulation next to 4.2 ms of network.  Here
// the row functions of the whole-search kernels (row_select / row_select_wide / row_backprop: lane s scores child
// slots s, s + 16, ..., binary64 DPP arg-max + wave ballot, tie draws from the tape, lane d back-propagates path
// node d; bit-identical to the generic operators, asserted on the device) run as two small kernels per simulation,
// four trees per wavefront, the trees staying in the arena (HBM / L2).  What one kernel hands to the other -- the
// path's nodes per lane -- goes through 64 ints per tree in the arena.
#pragma once
#include "mzx_resnet_search.h"

namespace mzx {
#ifndef MZX_HOSTCHECK

struct RowSearchArgs {
  SearchParams p;        // pbc_table / sqrt_table: the handle's device tables
  TreeLayout L;
  char* trees;           // arena: [num_trees][L.tree_bytes]
  const uint32_t* tape;  // [num_trees][tape_words]
  int32_t* sel_parent;   // [B] node whose hidden state feeds recurrent_inference
  int32_t* sel_action;   // [B]
  int32_t* sel_leaf;     // [B]
  int32_t* rowsel;       // [B][64]: SelCtx (5), action, then (node, parent, parent slot) of path depth d at 16 + 3 d
  const float* value;    // [B][F] logits of recurrent_inference
  const float* reward;   // [B][F]
  const float* policy;   // [B][A]
  int32_t sim;           // simulations finished so far (= visit count of the root)
};

constexpr int ROWSEL_INTS = 64;

inline bool row_search_supported(const SearchParams& p) { return p.num_actions <= WIDE_MAX_CHUNKS * FUSED_ROW; }

template <int AW>
__global__ void __launch_bounds__(64) row_select_kernel(const RowSearchArgs a) {
  const int tid = threadIdx.x, sub = tid & (FUSED_ROW - 1), row = tid / FUSED_ROW;
  const int tree = blockIdx.x * 4 + row;
  if (tree >= a.p.num_trees) return;     // whole rows leave: the row-level DPP / ballot steps stay row-uniform
  TreeRef t;
  t.base = a.trees + (size_t)tree * a.L.tree_bytes;
  t.L = a.L;
  RowState st;
  load_state(t, st);
  const uint32_t* tape = a.tape + (size_t)tree * a.p.tape_words;
  RowSel sel;
  if constexpr (AW == 0) {
    if (a.p.num_actions <= 8 * FUSED_ROW) sel = row_select_wide<8>(t, a.p, tape, sub, row, a.sim, st);
    else sel = row_select_wide<WIDE_MAX_CHUNKS>(t, a.p, tape, sub, row, a.sim, st);
  }
  else sel = row_select<AW>(t, a.p, tape, sub, row, a.sim, st);
  int32_t* rs = a.rowsel + (size_t)tree * ROWSEL_INTS;
  rs[16 + 3 * sub] = sel.my_node; rs[17 + 3 * sub] = sel.my_parent; rs[18 + 3 * sub] = sel.my_pslot;
  if (sub == 0) {
    a.sel_parent[tree] = sel.c.parent; a.sel_action[tree] = sel.action; a.sel_leaf[tree] = sel.c.leaf;
    rs[0] = sel.c.parent; rs[1] = sel.c.slot; rs[2] = sel.c.leaf; rs[3] = sel.c.depth; rs[4] = sel.c.to_play; rs[5] = sel.action;
    store_state(t, st);                  // tape position, tie draws, flags moved
  }
}

template <int AW>
__global__ void __launch_bounds__(64) row_expand_backprop_kernel(const RowSearchArgs a) {
  const int tid = threadIdx.x, sub = tid & (FUSED_ROW - 1), row = tid / FUSED_ROW;
  const int tree = blockIdx.x * 4 + row;
  if (tree >= a.p.num_trees) return;
  const SearchParams& p = a.p;
  TreeRef t;
  t.base = a.trees + (size_t)tree * a.L.tree_bytes;
  t.L = a.L;
  RowState st;
  load_state(t, st);
  const int32_t* rs = a.rowsel + (size_t)tree * ROWSEL_INTS;
  RowSel sel;
  sel.c.parent = rs[0]; sel.c.slot = rs[1]; sel.c.leaf = rs[2]; sel.c.depth = rs[3]; sel.c.to_play = rs[4]; sel.action = rs[5];
  sel.my_node = rs[16 + 3 * sub]; sel.my_parent = rs[17 + 3 * sub]; sel.my_pslot = rs[18 + 3 * sub];
  const int F = 2 * p.support_size + 1, A = p.num_actions;
  const float* vl = a.value + (size_t)tree * F;
  const float* rl = a.reward + (size_t)tree * F;
  const float* pl = a.policy + (size_t)tree * A;
  float value, reward;
  if constexpr (AW == 0) {
    value = row_decode_wide(vl, F, p.support_size, sub);
    reward = row_decode_wide(rl, F, p.support_size, sub);
    // priors = fp32 softmax over the full action space (self_play.py:460-462), canonical lane order
    float m = -MZX_INF;
    for (int i = sub; i < A; i += FUSED_ROW) m = fmaxf(m, pl[i]);
    m = row_max(m);
    float dl = 0.f;
    for (int i = sub; i < A; i += FUSED_ROW) dl += mzx_expf(pl[i] - m);
    const float den = row_sum(dl);
    for (int i = sub; i < A; i += FUSED_ROW) tree_init_slot(t, sel.c.leaf, i, (double)mzx_div(mzx_expf(pl[i] - m), den));
  } else {
    value = row_decode2(sub < F ? vl[sub] : 0.f, sub + 16 < F ? vl[sub + 16] : 0.f, F, p.support_size, sub);
    reward = row_decode2(sub < F ? rl[sub] : 0.f, sub + 16 < F ? rl[sub + 16] : 0.f, F, p.support_size, sub);
    const bool in = sub < A;
    const float lg = in ? pl[sub] : 0.f;
    const float m = row_max(in ? lg : -MZX_INF);
    const float e = in ? mzx_expf(lg - m) : 0.f;
    const float den = row_sum(e);
    if (in) tree_init_slot(t, sel.c.leaf, sub, (double)mzx_div(e, den));
  }
  row_backprop(t, p, sel, sub, row, (double)value, (double)reward, st);
  if (sub == 0) store_state(t, st);
}

template <int AW>
inline int row_search_step(const RowSearchArgs& a, stream_t stream) {
  const unsigned grid = (unsigned)((a.p.num_trees + 3) / 4);
  hipLaunchKernelGGL(row_select_kernel<AW>, dim3(grid), dim3(64), 0, stream, a);
  return (int)hipGetLastError();
}
template <int AW>
inline int row_search_apply(const RowSearchArgs& a, stream_t stream) {
  const unsigned grid = (unsigned)((a.p.num_trees + 3) / 4);
  hipLaunchKernelGGL(row_expand_backprop_kernel<AW>, dim3(grid), dim3(64), 0, stream, a);
  return (int)hipGetLastError();
}

// MCTS.run for B roots with a network that runs layer by layer on an engine taking indexed hidden states (the
// streamed MFMA engine): root by the generic kernels, then per simulation row-select, recurrent_inference straight
// from / into the arena's node store, row-expand + back-propagate.
inline int search_run_rows(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream,
                   const RootOverride* ov = nullptr) {
  const ArenaView v = arena_view(s, d_arena);
  mzx_net* net = s->net;
  const int B = s->p.num_trees;
  int rc = ensure_tables(s, d_arena, stream);
  if (rc) return rc;
  const bool ix_init = rz_enabled(net, false) || rb_enabled(net, false);
  NetIndex ix;
  ix.in_nodes = 1; ix.out_nodes = s->p.num_nodes;
  NetBuffers nb;
  nb.in = io->d_observation; nb.action = nullptr; nb.hidden = ix_init ? v.arena.hidden : v.dense_out;
  nb.value = v.value; nb.reward = v.reward; nb.policy = v.policy; nb.workspace = v.ws;
  if (!ov) {
    rc = run_network(net, false, nb, B, stream, ix_init ? &ix : nullptr);
    if (rc) return rc;
  }
  // roots the caller expanded itself (MCTS.run(..., override_root_with=root), self_play.py:275-277): their priors /
  // reward / hidden state replace initial_inference, the simulations run on the same kernel
  RootInitOp ri;
  ri.arena = v.arena; ri.p = v.p; ri.value_logits = v.value; ri.policy_logits = v.policy;
  ri.ext_priors = ov ? ov->priors : nullptr; ri.ext_root_reward = ov ? ov->reward : nullptr;
  ri.legal = io->d_legal_actions; ri.to_play = io->d_to_play; ri.noise = io->d_noise;
  ri.root_predicted_value = io->d_root_predicted_value;
  MZX_TRY_LAUNCH(launch<64>(ri, stream));
  if (ov || !ix_init) {
    HiddenMoveOp mv;
    mv.arena = v.arena; mv.num_trees = B; mv.num_nodes = s->p.num_nodes; mv.hidden_size = s->p.hidden_size;
    mv.dense = ov ? const_cast<float*>(ov->hidden) : v.dense_out; mv.node = nullptr; mv.to_arena = 1;
    MZX_TRY_LAUNCH(launch<256>(mv, stream));
  }
  RowSearchArgs a;
  a.p = v.p; a.L = s->L; a.trees = v.arena.trees; a.tape = io->d_tape;
  a.sel_parent = v.sel_parent; a.sel_action = v.sel_action; a.sel_leaf = v.sel_leaf;
  a.rowsel = (int32_t*)((char*)d_arena + s->off_rowsel);
  a.value = v.value; a.reward = v.reward; a.policy = v.policy;
  const bool wide = s->p.num_actions > FUSED_ROW || 2 * s->p.support_size + 1 > 2 * FUSED_ROW;
  const int aw = wide ? 0 : (s->p.num_actions <= 4 ? 4 : 16);
  nb.in = v.arena.hidden; nb.hidden = v.arena.hidden; nb.action = v.sel_action;
  ix.in_node = v.sel_parent; ix.out_node = v.sel_leaf; ix.in_nodes = s->p.num_nodes; ix.out_nodes = s->p.num_nodes;
  for (int k = 0; k < s->p.num_sims; ++k) {
    a.sim = k;
    MZX_TRY_LAUNCH(aw == 0 ? row_search_step<0>(a, stream) : aw == 4 ? row_search_step<4>(a, stream) : row_search_step<16>(a, stream));
    rc = run_network(net, true, nb, B, stream, &ix);
    if (rc) return rc;
    MZX_TRY_LAUNCH(aw == 0 ? row_search_apply<0>(a, stream) : aw == 4 ? row_search_apply<4>(a, stream) : row_search_apply<16>(a, stream));
  }
  return search_finish(s, io, d_arena, stream);
}

#endif  // !MZX_HOSTCHECK
}  // namespace mzx
