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
// path's nodes per lane -- goes through 64 ints per tree in the arena.  Shards of 1024 trees and more run as two
// half-shards on two HIP streams (search_run_rows).
#pragma once
#include "mzx_resnet_search.h"
#include "mzx_tuning.h"

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
  int2* rowpath;         // [B][num_nodes + 1]: (node, slot taken from its parent) of EVERY depth of the walk (paths beyond 16 levels)
  const float* value;    // [B][F] logits of recurrent_inference
  const float* reward;   // [B][F]
  const float* policy;   // [B][A]
  int32_t sim;           // simulations finished so far (= visit count of the root)
};

constexpr int ROWSEL_INTS = 64;

inline bool row_search_supported(const SearchParams& p) { return p.num_actions <= WIDE_MAX_CHUNKS * FUSED_ROW; }

// The tower whole-search kernel (mzx_tower_search.inc): every simulation of a search in ONE launch for wide residual
// networks whose recurrent program is two towers + tails + head chains; bit-identical trees to the per-simulation
// launches of search_run_rows on its tower route.  rt_search_shape: {trees per workgroup, row tiles per wave, workgroups,
// workgroups per CU, LDS bytes, threads per workgroup}, zeros when the kernel does not take the search.
bool rt_search_supported(const mzx_search* s);
void rt_search_shape(const mzx_search* s, int32_t out[6]);
int rt_search_simulations(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream);

// One selection walk of tree `t` by its 16-lane row (self_play.py:325-334): the walk's result for the network
// (parent node, action, leaf) and the row's share of the path -- 64 ints at `rs`: SelCtx (5), action, then (node, parent,
// parent slot) of path depth d at 16 + 3 d -- for row_expand_backprop_body.  Shared by row_select_kernel (rs in the arena)
// and rt_search_kernel (mzx_tower_search.inc: rs in LDS).
template <int AW>
__device__ __forceinline__ void row_select_body(const SearchParams& p, const TreeRef& t, const uint32_t* tape, int sim, int sub,
                                                int row_in_wave, int32_t* rs, int32_t* out_parent, int32_t* out_action,
                                                int32_t* out_leaf, int2* path = nullptr) {
  RowState st;
  load_state(t, st);
  RowSel sel;
  if constexpr (AW == 0) {
    if (p.num_actions <= 8 * FUSED_ROW) sel = row_select_wide<8>(t, p, tape, sub, row_in_wave, sim, st, path);
    else sel = row_select_wide<WIDE_MAX_CHUNKS>(t, p, tape, sub, row_in_wave, sim, st, path);
  }
  else sel = row_select<AW>(t, p, tape, sub, row_in_wave, sim, st, path);
  rs[16 + 3 * sub] = sel.my_node; rs[17 + 3 * sub] = sel.my_parent; rs[18 + 3 * sub] = sel.my_pslot;
  if (sub == 0) {
    *out_parent = sel.c.parent; *out_action = sel.action; *out_leaf = sel.c.leaf;
    rs[0] = sel.c.parent; rs[1] = sel.c.slot; rs[2] = sel.c.leaf; rs[3] = sel.c.depth; rs[4] = sel.c.to_play; rs[5] = sel.action;
    store_state(t, st);                  // tape position, tie draws, flags moved
  }
}

// Decode of the three heads, expansion of the leaf and back-propagation (self_play.py:343-353) by the tree's 16-lane row:
// vl / rl = value / reward logits [2 support + 1], pl = policy logits [A] of the leaf's recurrent_inference.
template <int AW>
__device__ __forceinline__ void row_expand_backprop_body(const SearchParams& p, const TreeRef& t, int sub, int row_in_wave,
                                                         const int32_t* rs, const float* vl, const float* rl, const float* pl,
                                                         const int2* path = nullptr) {
  RowState st;
  load_state(t, st);
  RowSel sel;
  sel.c.parent = rs[0]; sel.c.slot = rs[1]; sel.c.leaf = rs[2]; sel.c.depth = rs[3]; sel.c.to_play = rs[4]; sel.action = rs[5];
  sel.my_node = rs[16 + 3 * sub]; sel.my_parent = rs[17 + 3 * sub]; sel.my_pslot = rs[18 + 3 * sub];
  const int F = 2 * p.support_size + 1, A = p.num_actions;
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
  row_backprop(t, p, sel, sub, row_in_wave, (double)value, (double)reward, st, path);
  if (sub == 0) store_state(t, st);
}

template <int AW>
__global__ void __launch_bounds__(64) row_select_kernel(const RowSearchArgs a) {
  const int tid = threadIdx.x, sub = tid & (FUSED_ROW - 1), row = tid / FUSED_ROW;
  const int tree = blockIdx.x * 4 + row;
  if (tree >= a.p.num_trees) return;     // whole rows leave: the row-level DPP / ballot steps stay row-uniform
  TreeRef t;
  t.base = a.trees + (size_t)tree * a.L.tree_bytes;
  t.L = a.L;
  row_select_body<AW>(a.p, t, a.tape + (size_t)tree * a.p.tape_words, a.sim, sub, row, a.rowsel + (size_t)tree * ROWSEL_INTS,
                      a.sel_parent + tree, a.sel_action + tree, a.sel_leaf + tree, a.rowpath + (size_t)tree * (a.p.num_nodes + 1));
}

template <int AW>
__global__ void __launch_bounds__(64) row_expand_backprop_kernel(const RowSearchArgs a) {
  const int tid = threadIdx.x, sub = tid & (FUSED_ROW - 1), row = tid / FUSED_ROW;
  const int tree = blockIdx.x * 4 + row;
  if (tree >= a.p.num_trees) return;
  TreeRef t;
  t.base = a.trees + (size_t)tree * a.L.tree_bytes;
  t.L = a.L;
  const int F = 2 * a.p.support_size + 1, A = a.p.num_actions;
  row_expand_backprop_body<AW>(a.p, t, sub, row, a.rowsel + (size_t)tree * ROWSEL_INTS, a.value + (size_t)tree * F,
                               a.reward + (size_t)tree * F, a.policy + (size_t)tree * A, a.rowpath + (size_t)tree * (a.p.num_nodes + 1));
}

// ---------------------------------------------------------------------------
// Wide action spaces, a WAVEFRONT per tree (round 6): row_select_wide's walk (mzx_resnet_search.h: lane l scores slots l, l + 16,
// ... of a 16-lane row, eight chunks per level for the 121 actions of games/gomoku.py) with lane l of the whole wave scoring slots
// l, l + 64, ...: two chunks per level.  A level of such a walk is a trip to the tree in the arena + the binary64 UCB arithmetic of
// the lane's chunks (two divisions per slot); with 107-ply walks (the reference constructor's gomoku weights) the arithmetic of
// eight chunks was half of row_select_kernel<0> (341 us per launch alone against 169 us with two chunks' worth,
// gpurun_out r06ag).  One tree per wave: the walk's state is wave-uniform, no frozen rows.  Same operations per slot, the same
// maximum, maximisers counted in slot order, the same tape draws: the same walk (tests/test_gpu_streamed.py, row kernels against one
// thread per tree).  Hands the path over exactly as row_select_body does: 64 ints (lanes 0 .. 15) + the whole path.
__device__ __forceinline__ double wave_lane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int NCH>      // 64-slot chunks a lane holds: 2 = up to 128 actions, 4 = up to 256
__device__ __forceinline__ RowSel wave_select_wide(const TreeRef& t, const SearchParams& p, const uint32_t* tape, int lane, int sim,
                                                   RowState& st, int2* path) {
  RowSel r;
  int node = 0, depth = 0, slot = 0;
  int vtp = st.root_to_play;
  int N = sim;      // every finished simulation visited the root once
  double pbc = p.pbc_table[N], sq = p.sqrt_table[N];
  r.my_node = 0; r.my_parent = -1; r.my_pslot = -1;
  if (lane == 0) path[0] = make_int2(0, -1);
  for (;;) {
    const int d1 = depth + 1;
    const int nc = (node == 0) ? st.root_n : p.num_actions;
    int nv[NCH], cv[NCH];
    double pv[NCH], qv[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {   // slots beyond the node's children re-read slot 0 (masked below)
      const int s = ch * 64 + lane;
      const int ss = s < nc ? s : 0;
      nv[ch] = t.slot_visit(node, ss); pv[ch] = t.prior(node, ss); qv[ch] = t.slot_q(node, ss); cv[ch] = t.child(node, ss);
    }
    double sc[NCH];
    double mine = -MZX_INF;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const bool valid = ch * 64 + lane < nc;
      const double u = ucb_from(pbc, sq, nv[ch], pv[ch], qv[ch], st.mn, st.mx);
      sc[ch] = valid ? u : -MZX_INF;
      mine = (sc[ch] > mine) ? sc[ch] : mine;
    }
    // the maximum of the wave: the row butterfly, then the four rows' maxima
    const double rm = row_max_d<16>(mine);
    const double m01 = fmax(wave_lane_d(rm, 0), wave_lane_d(rm, 16)), m23 = fmax(wave_lane_d(rm, 32), wave_lane_d(rm, 48));
    const double best = fmax(m01, m23);
    // maximisers, chunk by chunk in slot order
    int nbest = 0, sl = 0;
    unsigned long long bits[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      bits[ch] = __ballot(sc[ch] == best && ch * 64 + lane < nc);
      if (nbest == 0 && bits[ch]) sl = ch * 64 + (__ffsll((long long)bits[ch]) - 1);
      nbest += __popcll(bits[ch]);
    }
    if (nbest > 1) {  // numpy.random.choice(ties): k-th maximiser in slot order
      ++st.ties;
      int k = tape_draw(tape, p.tape_words, st.tape_pos, st.flags, nbest);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int cnt = __popcll(bits[ch]);
        if (k >= 0 && k < cnt) {
          unsigned long long b = bits[ch];
          for (int q = k; q > 0; --q) b &= b - 1;
          sl = ch * 64 + (__ffsll((long long)b) - 1);
          k = -1;
        } else if (k >= cnt) {
          k -= cnt;
        }
      }
    }
    sl = __builtin_amdgcn_readfirstlane(sl);
    // the winner's child and visit count from the registers of the lane that scored it
    int cw_mine = cv[0], nw_mine = nv[0];
#pragma unroll
    for (int ch = 1; ch < NCH; ++ch) {
      const bool is = (sl >> 6) == ch;
      cw_mine = is ? cv[ch] : cw_mine;
      nw_mine = is ? nv[ch] : nw_mine;
    }
    const int cw = __builtin_amdgcn_readlane(cw_mine, sl & 63), n_w = __builtin_amdgcn_readlane(nw_mine, sl & 63);
    const bool mine_lane = lane == d1;      // (lanes 0 .. 15 are handed over, as a row's)
    r.my_parent = mine_lane ? node : r.my_parent;
    r.my_pslot = mine_lane ? sl : r.my_pslot;
    r.my_node = mine_lane ? cw : r.my_node;
    if (lane == 0) path[d1] = make_int2(cw, sl);
    vtp = (vtp + 1 < p.num_players) ? vtp + 1 : 0;      // players turn by turn, self_play.py:331-334
    depth = d1;
    slot = sl;
    if (cw < 0) break;
    N = n_w;
    node = cw;
    pbc = p.pbc_table[N]; sq = p.sqrt_table[N];
  }
  int leaf = st.n_nodes;
  if (leaf >= p.num_nodes) { st.flags |= TF_NODE_OVERFLOW; leaf = p.num_nodes - 1; }
  if (lane == depth) r.my_node = leaf;
  if (lane == 0) path[depth] = make_int2(leaf, slot);
  r.c.parent = node; r.c.slot = slot; r.c.leaf = leaf; r.c.depth = depth; r.c.to_play = vtp;
  r.action = (node == 0) ? t.root_action(slot) : slot;
  return r;
}

__global__ void __launch_bounds__(64) wave_select_kernel(const RowSearchArgs a) {
  const int lane = threadIdx.x, tree = blockIdx.x;
  TreeRef t;
  t.base = a.trees + (size_t)tree * a.L.tree_bytes;
  t.L = a.L;
  RowState st;
  load_state(t, st);
  int2* path = a.rowpath + (size_t)tree * (a.p.num_nodes + 1);
  const uint32_t* tape = a.tape + (size_t)tree * a.p.tape_words;
  const RowSel sel = (a.p.num_actions <= 128) ? wave_select_wide<2>(t, a.p, tape, lane, a.sim, st, path)
                                              : wave_select_wide<4>(t, a.p, tape, lane, a.sim, st, path);
  int32_t* rs = a.rowsel + (size_t)tree * ROWSEL_INTS;
  if (lane < FUSED_ROW) { rs[16 + 3 * lane] = sel.my_node; rs[17 + 3 * lane] = sel.my_parent; rs[18 + 3 * lane] = sel.my_pslot; }
  if (lane == 0) {
    a.sel_parent[tree] = sel.c.parent; a.sel_action[tree] = sel.action; a.sel_leaf[tree] = sel.c.leaf;
    rs[0] = sel.c.parent; rs[1] = sel.c.slot; rs[2] = sel.c.leaf; rs[3] = sel.c.depth; rs[4] = sel.c.to_play; rs[5] = sel.action;
    store_state(t, st);                  // tape position, tie draws, flags moved
  }
}

template <int AW>
inline int row_search_step(const RowSearchArgs& a, stream_t stream) {
  if (AW == 0 && tune(TUNE_WAVE_SELECT) && a.p.num_actions <= 256) {      // wide records: a wavefront per tree
    hipLaunchKernelGGL(wave_select_kernel, dim3((unsigned)a.p.num_trees), dim3(64), 0, stream, a);
    return (int)hipGetLastError();
  }
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
// from / into the arena's node store, row-expand + back-propagate -- or, where the recurrent program is two towers with
// tails and head chains (connect4-class networks), ALL simulations in one launch of rt_search_kernel.
//
// Routing of a WIDE residual network that also fits the LDS-resident whole-search kernel (connect4: 64 channels).  The
// LDS-resident engine (mzx_resnet_fused.h) and the streamed engine sum a convolution in different orders, so which of
// them a search runs on must not depend on the shard size (a tree's result would): such a network ALWAYS takes the
// tower arithmetic -- rt_search_kernel, or (tuning "rt_search" = 0, or shards above "rt_max_trees") the per-simulation
// launches with the trunks as towers, which build bit-identical trees -- unless the streamed route is switched off
// altogether ("wide_towers" = 0: rz_search_kernel at every shard size, the A/B).  Measured whole steps, connect4 x
// 200 simulations, of the FP32 MFMA peak (profiles/r04_c4_by_shard.txt, profiles/r05_c4_by_shard.txt): rz_search_kernel
// 0.53 at every shard size; per-simulation launches 0.45 / 0.58 / 0.67 / 0.70 / 0.72 at 512 / 1024 / 1536 / 3072 / 9216.
enum SearchRoute { ROUTE_RZ = 0, ROUTE_ROWS = 1, ROUTE_RT = 2 };

inline bool wide_tower_network(const mzx_search* s) {
  const mzx_net* net = s->net;
  if (!net || !net->rb.ok || !net->rb.initial.ok || !net->rb.recurrent.ok || net->rb_no_towers) return false;
  if (!row_search_supported(s->p)) return false;
  if (net->cfg.channels < 48) return false;          // narrow networks: the wave / tile whole-search kernels win by far
  bool any = false;
  for (const RbTower& tw : net->rb.recurrent.towers) any |= rb_tower_use(tw, std::max(1, s->p.num_trees));
  return any;
}

// Networks that run on the streamed engine anyway (forced by the network mode, or too large for the LDS-resident
// engine): rt_search_kernel when it takes them and the shard is within "rt_max_trees" -- the same trees either way.
inline bool streamed_whole_search(const mzx_search* s) {
  const int rt = tune(TUNE_RT_SEARCH);
  if (rt == 0 || !s->net || s->net->rb_no_towers || tune(TUNE_RB_TAIL) == 0 || tune(TUNE_RB_TOWER_T) > 0) return false;
  // (the per-simulation launches must take the tower route at this shard as well -- a tower whose shape would waste too
  // many MFMA rows launches layer by layer, rb_tower_use, with the layer kernel's channel groups: another summation order)
  for (const RbTower& tw : s->net->rb.recurrent.towers)
    if (!rb_tower_use(tw, std::max(1, s->p.num_trees))) return false;
  if (!rt_search_supported(s)) return false;
  if (rt == 1) return true;
  // automatic: networks of at most 64 channels.  A 128-channel tower (games/gomoku.py) is one board per workgroup and one
  // workgroup per CU, and nothing hides its 121-action tree walks: measured 0.735 of the peak against 0.744 launch by
  // launch on two streams (profiles/r05_rt_experiments.txt section 8)
  return s->net->rb.recurrent.towers[0].ntiles <= 4 && s->p.num_trees <= tune(TUNE_RT_MAX_TREES);
}

inline int wide_search_route(const mzx_search* s) {
  if (tune(TUNE_WIDE_TOWERS) == 0 || !wide_tower_network(s)) return ROUTE_RZ;
  return streamed_whole_search(s) ? ROUTE_RT : ROUTE_ROWS;
}

inline int search_run_rows(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream,
                   const RootOverride* ov = nullptr, bool force_streamed = false, bool whole_search = false) {
  const ArenaView v = arena_view(s, d_arena);
  mzx_net* net = s->net;
  const int B = s->p.num_trees;
  int rc = ensure_tables(s, d_arena, stream);
  if (rc) return rc;
  // (force_streamed: the network would run on the LDS-resident engine by default; this search runs it on the streamed
  // one -- row_search_preferred -- without touching the network handle's mode)
  auto run_network = [&](mzx_net* n, bool recurrent, const NetBuffers& b, int batch, stream_t st, const NetIndex* ixp) {
    return force_streamed ? rb_run_program(n, recurrent, b, batch, st, ixp) : mzx::run_network(n, recurrent, b, batch, st, ixp);
  };
  const bool ix_init = force_streamed || rz_enabled(net, false) || rb_enabled(net, false);
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
  // ---- the simulations, all of them in one launch (rt_search_kernel, mzx_tower_search.inc) ...
  if (whole_search) {
    s->last_kernel = "mzx::rt_search_kernel";
    rc = rt_search_simulations(s, io, d_arena, stream);
    if (rc) return rc;
    return search_finish(s, io, d_arena, stream);
  }
  // ---- ... or launch by launch.  Large shards run as TWO HALF-SHARDS on two HIP streams: the layers of one half start under
  // the tail of the other's (a launch ends with a few workgroups on a mostly idle chip, and ~6 us pass before a
  // dependent launch starts), and the small per-simulation kernels (select, expand, heads, scaling) of one half hide
  // under the trunk layers of the other.  Trees are independent and a sample's arithmetic does not depend on the
  // batch it runs in (the halves keep the planned launch shape), so the trees are the ones of the undivided run.
  const int split_min = tune(TUNE_ROW_SPLIT_MIN);            // 0: never
  int parts = 1;
  // (decided per run: it depends on the threshold, on the network handle's mode and on the launch-shape tuning, all of
  // which may move between two runs of one handle; a few dozen host-side shape evaluations)
  s->split_first = rb_split_first(net, B, split_min);
  const int first = s->split_first > 0 ? s->split_first : B;
  if (s->split_first > 0) {
    if (!s->side_stream) {
      hipStream_t st = nullptr;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
          hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
          hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess) {
        s->side_stream = st; s->ev_fork = e0; s->ev_join = e1;
      } else {
        if (e0) (void)hipEventDestroy(e0);
        if (st) (void)hipStreamDestroy(st);
        (void)hipGetLastError();
      }
    }
    if (s->side_stream) parts = 2;
  }
  if (parts == 2)
    s->last_kernel = "mzx::rb_tower_kernel / mzx::rb_gemm_kernel / mzx::rb_gemm_multi_kernel (streamed FP32-MFMA trunks, layers, head MLP levels) between mzx::row_select_kernel / mzx::row_expand_backprop_kernel, two half-shards on two streams";
  const bool wide = s->p.num_actions > FUSED_ROW || 2 * s->p.support_size + 1 > 2 * FUSED_ROW;
  const int aw = wide ? 0 : (s->p.num_actions <= 4 ? 4 : 16);
  const int F = 2 * s->p.support_size + 1, A = s->p.num_actions;
  const int64_t node_floats = (int64_t)s->p.num_nodes * s->p.hidden_size;
  RowSearchArgs as[2];
  NetBuffers nbs[2];
  NetIndex ixs[2];
  int count[2] = {parts == 2 ? first : B, B - first};
  stream_t streams[2] = {stream, (stream_t)s->side_stream};
  for (int h = 0; h < parts; ++h) {
    const int64_t o = h ? first : 0;      // (h = 1 only when split)
    RowSearchArgs& a = as[h];
    a.p = v.p; a.p.num_trees = count[h]; a.L = s->L;
    a.trees = v.arena.trees + o * s->L.tree_bytes; a.tape = io->d_tape + o * s->p.tape_words;
    a.sel_parent = v.sel_parent + o; a.sel_action = v.sel_action + o; a.sel_leaf = v.sel_leaf + o;
    a.rowsel = (int32_t*)((char*)d_arena + s->off_rowsel) + o * ROWSEL_INTS;
    a.rowpath = (int2*)((char*)d_arena + s->off_rowpath) + o * (s->p.num_nodes + 1);
    a.value = v.value + o * F; a.reward = v.reward + o * F; a.policy = v.policy + o * A;
    NetBuffers& n = nbs[h];
    n.in = v.arena.hidden + o * node_floats; n.hidden = v.arena.hidden + o * node_floats; n.action = a.sel_action;
    n.value = v.value + o * F; n.reward = v.reward + o * F; n.policy = v.policy + o * A;
    n.workspace = v.ws + o * net_ws_per_sample(net);             // the workspace is linear in the batch
    ixs[h].in_node = a.sel_parent; ixs[h].out_node = a.sel_leaf;
    ixs[h].in_nodes = s->p.num_nodes; ixs[h].out_nodes = s->p.num_nodes;
  }
  if (parts == 2) {
    if (hipEventRecord((hipEvent_t)s->ev_fork, stream) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)s->side_stream, (hipEvent_t)s->ev_fork, 0) != hipSuccess) {
      set_error("row search: fork onto the second stream failed: %s", hipGetErrorString(hipGetLastError()));
      return MZX_ERR_RUNTIME;
    }
  }
  // the simulations of every part; whatever happens, the caller's stream is joined with the second one afterwards
  auto simulations = [&]() -> int {
    for (int k = 0; k < s->p.num_sims; ++k) {
      for (int h = 0; h < parts; ++h) {
        RowSearchArgs& a = as[h];
        a.sim = k;
        MZX_TRY_LAUNCH(aw == 0 ? row_search_step<0>(a, streams[h]) : aw == 4 ? row_search_step<4>(a, streams[h]) : row_search_step<16>(a, streams[h]));
        const int nrc = run_network(net, true, nbs[h], count[h], streams[h], &ixs[h]);
        if (nrc) return nrc;
        MZX_TRY_LAUNCH(aw == 0 ? row_search_apply<0>(a, streams[h]) : aw == 4 ? row_search_apply<4>(a, streams[h]) : row_search_apply<16>(a, streams[h]));
      }
    }
    return 0;
  };
  rc = simulations();
  if (parts == 2) {
    if (hipEventRecord((hipEvent_t)s->ev_join, (hipStream_t)s->side_stream) != hipSuccess ||
        hipStreamWaitEvent(stream, (hipEvent_t)s->ev_join, 0) != hipSuccess) {
      if (!rc) set_error("row search: join of the second stream failed: %s", hipGetErrorString(hipGetLastError()));
      return rc ? rc : MZX_ERR_RUNTIME;
    }
  }
  if (rc) return rc;
  return search_finish(s, io, d_arena, stream);
}

#endif  // !MZX_HOSTCHECK
}  // namespace mzx
