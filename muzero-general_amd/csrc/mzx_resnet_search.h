// mzx_resnet_search.h -- all simulations of a move in ONE launch for residual networks:
// num_simulations x {select, recurrent_inference, expand, backpropagate} of
// MCTS.run (/root/reference/self_play.py:319-355) with the network of
// mzx_resnet_fused.h (models.py:555-623) inside the same kernel.
//
// A workgroup owns T trees for the whole search (no inter-workgroup traffic, no grid
// barrier).  Tree t is walked lane-parallel by 16-lane row t of the workgroup with the
// SAME code as the fully connected kernel (row_select / row_backprop of mzx_fused_fc.h,
// i.e. the binary64 operations of mzx_tree.h in the reference's order): lane s scores
// child slot s, binary64 DPP argmax + wave ballot, tie draws from the tape, lane d
// back-propagates path node d.  The trees and the per-node hidden states stay in the
// caller's arena (HBM; L2-resident at these sizes: C4 40 KB per tree + 2.1 MB of hidden
// states); the selected parents' states are gathered into the LDS activation slots, all
// four waves (or eight) then run the layer GEMMs on the MFMA pipes, the scaled next
// states go straight back to the arena and the head logits are decoded in registers
// (support_to_scalar / softmax in the canonical 16-lane order) by the row that owns the
// tree.  The operator table, epilogue parameters and -- when they fit -- the weights are
// staged into LDS once per MOVE instead of once per simulation.
//
// The root (initial_inference + root expansion) and the result gathering remain the
// kernels of the generic path: 4 launches per move instead of 3 * num_simulations + 4.
#pragma once
#include "mzx_fused_fc2.h"

namespace mzx {

#ifndef MZX_HOSTCHECK

struct RzSearchArgs {
  RzArgs net;            // recurrent program (in / hidden_out = the arena's hidden-state store)
  SearchParams p;        // pbc_table / sqrt_table: global (arena), copied to LDS by the kernel
  TreeLayout L;
  char* trees;           // arena: [num_trees][L.tree_bytes]
  const uint32_t* tape;  // [num_trees][tape_words]
  int32_t num_sims, sim0;
  uint32_t* prof;        // nullable: [grid][8] cycles per phase, accumulated over the simulations (mode flag 8)
  int32_t tree_lds;      // 1: the workgroup's trees are staged into LDS for the whole search (they fit)
};

// floats of LDS the search adds behind the network engine's image: selection hand-off, tables
inline int64_t rz_search_extra_floats(const SearchParams& p) { return 64 + 4 * (int64_t)(p.num_nodes + 1) + 4; }

__device__ __forceinline__ void load_state(const TreeRef& t, RowState& st) {
  st.mn = t.mm_min(); st.mx = t.mm_max();
  st.n_nodes = t.meta(TM_N_NODES); st.tape_pos = t.meta(TM_TAPE_POS); st.flags = t.meta(TM_FLAGS);
  st.ties = t.meta(TM_TIE_DRAWS); st.max_depth = t.meta(TM_MAX_DEPTH); st.sum_depth = t.meta(TM_SUM_DEPTH);
  st.root_n = t.meta(TM_ROOT_N); st.root_to_play = t.to_play(0);
}


// ---------------------------------------------------------------------------
// Wide action spaces (more than one 16-lane row of child slots: gomoku 121, atari 18, ...).  Lane l scores
// the slots l, l + 16, l + 32, ... of the node (self_play.py:363-404, same binary64 operations as ucb_from);
// the arg-max is the row butterfly over the lanes' own maxima, the maximisers are counted per 16-slot chunk
// with one wave ballot each so that numpy.random.choice(ties) picks the k-th one in SLOT order, and the
// winner's link / visit count are re-read from the tree (one broadcast load) instead of permuted.
constexpr int WIDE_MAX_CHUNKS = 16;   // up to 256 actions

// NCH: 16-slot chunks a lane can hold (8 covers up to 128 actions).  The slot statistics of ALL chunks of a level are
// requested before the first one is used -- the trees live in HBM in the per-simulation path (1.2 MB per gomoku tree),
// where a level costs one memory round trip this way instead of one per chunk.
template <int NCH = WIDE_MAX_CHUNKS>
__device__ __forceinline__ RowSel row_select_wide(const TreeRef& t, const SearchParams& p, const uint32_t* tape, int sub,
                                                  int row_in_wave, int sim, RowState& st, int2* path = nullptr) {
  RowSel r;
  int node = 0, depth = 0, slot = 0;
  if (path && sub == 0) path[0] = make_int2(0, -1);      // (the whole path for row_backprop: walks deeper than a row's lanes)
  int vtp = st.root_to_play;
  int N = sim;  // every finished simulation visited the root once
  double pbc = p.pbc_table[N], sq = p.sqrt_table[N];
  r.my_node = 0; r.my_parent = -1; r.my_pslot = -1;
  bool done = false;
  for (;;) {
    const int d1 = depth + 1;
    const int nc = (node == 0) ? st.root_n : p.num_actions;
    // ONE round trip to the tree per level (round 6): the child of every slot comes with its statistics, and the winner's
    // child / visit count are picked from the registers of the lane that scored it -- it was two dependent trips per
    // level (statistics; then child and visit count of the winner), 6.8 us per level of a games/gomoku.py walk on the
    // reference constructor's weights (107 levels on average).  Same values.
    int nv[NCH], cv[NCH];
    double pv[NCH], qv[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {   // slots beyond the node's children re-read slot 0 (masked below)
      const int s = ch * FUSED_ROW + sub;
      const int ss = s < nc ? s : 0;
      nv[ch] = t.slot_visit(node, ss); pv[ch] = t.prior(node, ss); qv[ch] = t.slot_q(node, ss); cv[ch] = t.child(node, ss);
    }
    double sc[NCH];
    double mine = -MZX_INF;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const bool valid = ch * FUSED_ROW + sub < nc;
      const double u = ucb_from(pbc, sq, nv[ch], pv[ch], qv[ch], st.mn, st.mx);
      sc[ch] = valid ? u : -MZX_INF;
      mine = (sc[ch] > mine) ? sc[ch] : mine;
    }
    const double best = row_max_d<16>(mine);
    // maximisers, chunk by chunk in slot order
    int nbest = 0, sl = 0;
    unsigned bits[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      bits[ch] = row_bits(__ballot(sc[ch] == best && ch * FUSED_ROW + sub < nc), row_in_wave);
      if (nbest == 0 && bits[ch]) sl = ch * FUSED_ROW + (__ffs(bits[ch]) - 1);
      nbest += __popc(bits[ch]);
    }
    if (nbest > 1 && !done) {  // numpy.random.choice(ties): k-th maximiser in slot order
      ++st.ties;
      int k = tape_draw(tape, p.tape_words, st.tape_pos, st.flags, nbest);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int cnt = __popc(bits[ch]);
        if (k >= 0 && k < cnt) {
          unsigned b = bits[ch];
          for (int q = k; q > 0; --q) b &= b - 1;
          sl = ch * FUSED_ROW + (__ffs(b) - 1);
          k = -1;
        } else if (k >= cnt) {
          k -= cnt;
        }
      }
    }
    // the winner's child and visit count: chunk sl / 16 of lane sl % 16 of this row
    int cw_mine = cv[0], nw_mine = nv[0];
#pragma unroll
    for (int ch = 1; ch < NCH; ++ch) {
      const bool is = (sl >> 4) == ch;
      cw_mine = is ? cv[ch] : cw_mine;
      nw_mine = is ? nv[ch] : nw_mine;
    }
    const int cw = perm_i(cw_mine, sl & (FUSED_ROW - 1), row_in_wave);
    const int n_w = perm_i(nw_mine, sl & (FUSED_ROW - 1), row_in_wave);
    const bool act = !done;
    const bool mine_lane = act && sub == d1;
    r.my_parent = mine_lane ? node : r.my_parent;
    r.my_pslot = mine_lane ? sl : r.my_pslot;
    r.my_node = mine_lane ? cw : r.my_node;
    if (path && act && sub == 0) path[d1] = make_int2(cw, sl);
    const int nvtp = (vtp + 1 < p.num_players) ? vtp + 1 : 0;  // players turn by turn, :331-334
    depth = act ? d1 : depth;
    slot = act ? sl : slot;
    vtp = act ? nvtp : vtp;
    const bool go = act && cw >= 0;
    N = go ? n_w : N;
    node = go ? cw : node;
    done = done || (cw < 0);
    if (__all(done)) break;
    pbc = p.pbc_table[N]; sq = p.sqrt_table[N];     // (a hot 3 kB table: an L2 hit behind the level's last instruction)
  }
  int leaf = st.n_nodes;
  if (leaf >= p.num_nodes) { st.flags |= TF_NODE_OVERFLOW; leaf = p.num_nodes - 1; }
  if (sub == depth) r.my_node = leaf;
  if (path && sub == 0) path[depth] = make_int2(leaf, slot);
  r.c.parent = node; r.c.slot = slot; r.c.leaf = leaf; r.c.depth = depth; r.c.to_play = vtp;
  r.action = (node == 0) ? t.root_action(slot) : slot;
  return r;
}

// support_to_scalar (models.py:645-666) of F logits readable by every lane, canonical lane order (mzx_tree.h):
// element i belongs to lane i % 16, accumulated in increasing i; any F.
__device__ __forceinline__ float row_decode_wide(const float* lg, int F, int support, int sub) {
  float m = -MZX_INF;
  for (int i = sub; i < F; i += FUSED_ROW) m = fmaxf(m, lg[i]);
  m = row_max(m);
  float dl = 0.f;
  for (int i = sub; i < F; i += FUSED_ROW) dl += mzx_expf(lg[i] - m);
  const float den = row_sum(dl);
  float num = 0.f;
  for (int i = sub; i < F; i += FUSED_ROW) num += (float)(i - support) * mzx_div(mzx_expf(lg[i] - m), den);
  return support_inverse_transform(row_sum(num));
}

// floats of LDS per tree / per workgroup for the record form of an LDS-resident tree (TREC kernels)
__host__ __device__ inline int64_t rz_trec_tree_floats(int N, int AW) {
  auto al16 = [](int64_t x) { return (x + 15) & ~int64_t(15); };
  return (int64_t(32) * N * AW + int64_t(32) * N + al16(int64_t(8) * (N + 1)) + al16(int64_t(4) * AW) + 16) / 4;
}
inline int64_t rz_trec_extra_floats(int N) { return 2 * (int64_t)(N + 2); }   // refined-reciprocal table

// Arena tree (TreeLayout, as RootInitOp left it: normally the root alone) -> the slot / node records of
// mzx_fused_fc2.h in LDS, and back (what FinalizeOp and mzx_search_dump read).  Called by the 16-lane row of the tree.
template <int RW>
__device__ __forceinline__ void fc2_from_arena(const Fc2Tree& FT, Fc2Row& rst, const TreeRef& t, int sub) {
  const int nn = t.meta(TM_N_NODES), rootn = t.meta(TM_ROOT_N);
  for (int n = sub; n < nn; n += FUSED_ROW) {
    Fc2Node r;
    r.value_sum = t.value_sum(n); r.reward = t.reward(n); r.visit = t.visit(n); r.to_play = t.to_play(n);
    r.parent = t.parent(n); r.parent_slot = t.parent_slot(n);
    FT.nodes[n] = r;
    const int nc = (n == 0) ? rootn : FT.A;
    for (int s2 = 0; s2 < RW; ++s2) {
      Fc2Slot q;
      const bool in = s2 < nc;
      q.prior = in ? t.prior(n, s2) : 0.0; q.q = in ? t.slot_q(n, s2) : 0.0;
      q.n = in ? t.slot_visit(n, s2) : 0; q.child = in ? t.child(n, s2) : -1;
      q.ps = in ? prior_score(FT.pbc[r.visit], FT.sqt[r.visit], q.n, FT.inv_y[q.n + 1], q.prior) : -MZX_INF;
      FT.slots[n * RW + s2] = q;
    }
  }
  if (sub < RW) FT.roota[sub] = (sub < rootn) ? t.root_action(sub) : -1;
  if (sub == 0) { FT.path[0] = make_int2(0, -1); FT.mm[0] = t.mm_min(); FT.mm[1] = t.mm_max(); }
  rst.n_nodes = nn; rst.tape_pos = t.meta(TM_TAPE_POS); rst.flags = t.meta(TM_FLAGS); rst.ties = t.meta(TM_TIE_DRAWS);
  rst.max_depth = t.meta(TM_MAX_DEPTH); rst.sum_depth = t.meta(TM_SUM_DEPTH); rst.root_n = rootn;
  rst.root_to_play = t.to_play(0);
}

template <int RW>
__device__ __forceinline__ void fc2_to_arena(const Fc2Tree& FT, const Fc2Row& rst, const TreeRef& t, int sub) {
  for (int n = sub; n < rst.n_nodes; n += FUSED_ROW) {
    const Fc2Node r = FT.nodes[n];
    t.value_sum(n) = r.value_sum; t.reward(n) = r.reward; t.visit(n) = r.visit; t.to_play(n) = r.to_play;
    t.parent(n) = r.parent; t.parent_slot(n) = r.parent_slot;
    const int nc = (n == 0) ? rst.root_n : FT.A;
    for (int s2 = 0; s2 < nc; ++s2) {
      const Fc2Slot q = FT.slots[n * RW + s2];
      t.prior(n, s2) = q.prior; t.slot_q(n, s2) = q.q; t.slot_visit(n, s2) = q.n; t.child(n, s2) = q.child;
    }
  }
  if (sub == 0) {
    t.mm_min() = FT.mm[0]; t.mm_max() = FT.mm[1];
    t.meta(TM_N_NODES) = rst.n_nodes; t.meta(TM_TAPE_POS) = rst.tape_pos; t.meta(TM_FLAGS) = rst.flags;
    t.meta(TM_TIE_DRAWS) = rst.ties; t.meta(TM_MAX_DEPTH) = rst.max_depth; t.meta(TM_SUM_DEPTH) = rst.sum_depth;
  }
}

// Carves one tree's records out of `rec` (rz_trec_tree_floats(NN, RW) floats)
template <int RW>
__device__ __forceinline__ void fc2_carve(Fc2Tree& FT, char* rec, int NN) {
  FT.slots = (Fc2Slot*)rec;
  FT.nodes = (Fc2Node*)(rec + (size_t)32 * NN * RW);
  FT.path = (int2*)((char*)FT.nodes + (size_t)32 * NN);
  FT.roota = (int32_t*)((char*)FT.path + (((size_t)8 * (NN + 1) + 15) & ~(size_t)15));
  FT.mm = (double*)((char*)FT.roota + (((size_t)4 * RW + 15) & ~(size_t)15));
}

// AW: lanes that can hold a child slot (4, 16), or 0 = wide (several slots per lane, any support size).
// TREC: the workgroup's trees live in LDS as the 32-byte slot / node records of mzx_fused_fc2.h (cached prior
// scores, hoisted division halves, operands of back-propagation fetched before the network) instead of the
// arena's struct-of-arrays slab; converted from / to the arena layout at the ends of the launch.
template <bool WLDS, int NW, int AW, int MM, bool TREC = false>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
rz_search_kernel(const RzSearchArgs sa) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float rz_lds[];
  const RzArgs& a = sa.net;
  const int tid = threadIdx.x, T = a.T;
  const int b0 = blockIdx.x * T;
  const int ntree = min(T, a.batch - b0);
  const RzCtx cx = rz_carve<NW>(a, rz_lds);
  float* extra = (float*)cx.wlds + (WLDS ? a.w_floats : 0);
  int32_t* sel_parent = (int32_t*)extra;
  int32_t* sel_action = sel_parent + 16;
  int32_t* sel_leaf = sel_action + 16;
  double* tables = (double*)(extra + 64);
  rz_setup<WLDS, NW>(a, cx);
  const int ntab = 2 * (sa.p.num_nodes + 1);
  for (int i = tid; i < ntab; i += NT) tables[i] = sa.p.pbc_table[i];   // pbc[N+1] then sqrt[N+1], contiguous
  SearchParams p = sa.p;
  p.pbc_table = tables;
  p.sqrt_table = tables + (sa.p.num_nodes + 1);

  // tree <-> 16-lane row
  const int sub = tid & (FUSED_ROW - 1), row = tid / FUSED_ROW, row_in_wave = row & 3;
  const bool row_valid = row < ntree;   // the other rows idle through the tree phases, all lanes run the network
  const int tree = b0 + (row_valid ? row : 0);
  TreeRef t;
  t.base = sa.trees + (size_t)tree * sa.L.tree_bytes;
  t.L = sa.L;
  // Small searches: the trees of the workgroup live in LDS for the whole launch (the selection walk is a
  // chain of dependent reads per level -- an LDS round trip instead of an L2 one), staged in and out once.
  double* tree_lds = tables + ntab;
  const int slab8 = (int)(sa.L.tree_bytes >> 3);      // tree_bytes is a multiple of 8
  constexpr int RW = (AW == 0) ? 16 : AW;             // record lanes per node (TREC)
  Fc2Tree FT;
  Fc2Row rst;
  if constexpr (TREC) {
    const int NNr = sa.p.num_nodes;
    double* inv_y = tree_lds;                                       // [NN + 2]
    for (int i = tid; i < NNr + 2; i += NT) inv_y[i] = recip_refined((double)(i > 0 ? i : 1));
    char* rec = (char*)(inv_y + NNr + 2) + (size_t)(row_valid ? row : 0) * (size_t)(4 * rz_trec_tree_floats(NNr, RW));
    fc2_carve<RW>(FT, rec, NNr);
    FT.pbc = tables; FT.sqt = tables + (NNr + 1); FT.inv_y = inv_y;
    FT.disc = p.discount; FT.A = p.num_actions; FT.NN = NNr; FT.P = p.num_players;
    __syncthreads();
    FT.pb_leaf = FT.pbc[1] * div_by(FT.sqt[1], 1.0, inv_y[1]);
    if (row_valid) fc2_from_arena<RW>(FT, rst, t, sub);
    __syncthreads();
  } else if (sa.tree_lds) {
    const double* src = (const double*)(sa.trees + (size_t)b0 * sa.L.tree_bytes);
    for (int i = tid; i < ntree * slab8; i += NT) tree_lds[i] = src[i];
    t.base = (char*)(tree_lds + (size_t)(row_valid ? row : 0) * slab8);
    __syncthreads();
  }
  const uint32_t* tape = sa.tape + (size_t)tree * p.tape_words;
  RowState st;
  if (!TREC && row_valid) load_state(t, st);
  const int F = a.out_n[0], A = a.out_n[2];
  __syncthreads();
  // cycle profile (mode flag 8): thread 0 accumulates the shader clock per phase
  uint32_t pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = 0;
  const bool prof = sa.prof != nullptr && tid == 0;
  unsigned long long rt0 = 0;     // 100 MHz reference clock: shader cycles / reference ticks = the effective shader clock
  if (prof) { t_last = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#define RZS_PROF(k) if (prof) { const unsigned long long _t = __builtin_readcyclecounter(); pc[k] += (uint32_t)(_t - t_last); t_last = _t; }

  for (int sim = 0; sim < sa.num_sims; ++sim) {
    // ---- selection (self_play.py:325-334)
    RowSel sel;
    Fc2Walk wk;
    Fc2Lane<RW> lane_ops;
    if (row_valid) {
      if constexpr (TREC) {
        wk = fc2_walk<RW>(FT, rst, tape, p.tape_words, sub, row_in_wave);
        if (sub == 0) { sel_parent[row] = wk.parent; sel_action[row] = wk.action; sel_leaf[row] = wk.leaf; }
        wave_sync();
        lane_ops = fc2_load_lane<RW>(FT, wk, wk.levels >> 4, sub);   // hides behind the network
      } else {
        if constexpr (AW == 0) {
          if (p.num_actions <= 8 * FUSED_ROW) sel = row_select_wide<8>(t, p, tape, sub, row_in_wave, sa.sim0 + sim, st);
          else sel = row_select_wide<WIDE_MAX_CHUNKS>(t, p, tape, sub, row_in_wave, sa.sim0 + sim, st);
        }
        else sel = row_select<AW>(t, p, tape, sub, row_in_wave, sa.sim0 + sim, st);
        if (sub == 0) { sel_parent[row] = sel.c.parent; sel_action[row] = sel.action; sel_leaf[row] = sel.c.leaf; }
      }
    }
    RZS_PROF(0)
    __syncthreads();
    RZS_PROF(1)

    // ---- recurrent_inference on the T selected (parent state, action) pairs (models.py:620-623)
    rz_load_input<NW>(a, cx, b0, ntree, sel_parent, sel_action, true);
    RZS_PROF(2)
    // slots of independent operators, one barrier per slot (rz_schedule); descriptors come from the
    // LDS-resident program image (measured faster than scalar loads from L2; fetching the next descriptor
    // ahead of the current operator costs 4 %, profiles/r01_rz_slots_ab.txt)
    for (int o = 0; o < a.n_ops;) {
      bool last;
      do {
        const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)cx.work[o * NW + cx.wave]);
        const RzOp op = rz_fetch_op(cx.simg, o);
        if (op.kind == RZ_GEMM) { if (!RZ_DBG(a, 4)) rz_gemm<WLDS, NW, MM>(op, a, cx, w, rz_op_class<NW>(a, cx, o)); }
        else rz_scale<NW>(op, a, cx, b0, ntree, sel_leaf, true);
        last = ((op.sched >> 16) & 1u) != 0;
        ++o;
      } while (!last);
      if (!RZ_DBG(a, 8)) __syncthreads();
    }
    RZS_PROF(3)

    // ---- decode, expand, back-propagate (self_play.py:343-353), the row that owns the tree
    if (row_valid) {
      const float* vl = cx.reg + T * a.out_off[0] + row * a.out_ts[0];
      const float* rl = cx.reg + T * a.out_off[1] + row * a.out_ts[1];
      const float* pl = cx.reg + T * a.out_off[2] + row * a.out_ts[2];
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
        if constexpr (TREC) fc2_expand<RW>(FT, wk.leaf, sub, in, (double)mzx_div(e, den));
        else if (in) tree_init_slot(t, sel.c.leaf, sub, (double)mzx_div(e, den));
      }
      if constexpr (TREC) fc2_backprop<RW>(FT, rst, wk, lane_ops, sub, (double)value, (double)reward);
      else row_backprop(t, p, sel, sub, row_in_wave, (double)value, (double)reward, st);
    }
    RZS_PROF(4)
    __syncthreads();
    RZS_PROF(5)
  }
#undef RZS_PROF
  if (prof) {
    pc[7] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0);
    for (int k = 0; k < 8; ++k) sa.prof[blockIdx.x * 8 + k] = pc[k];
  }
  if constexpr (TREC) {
    if (row_valid) { wave_sync(); fc2_to_arena<RW>(FT, rst, t, sub); }
    return;
  }
  if (row_valid && sub == 0) store_state(t, st);
  if (sa.tree_lds) {
    __syncthreads();
    double* dst = (double*)(sa.trees + (size_t)b0 * sa.L.tree_bytes);
    for (int i = tid; i < ntree * slab8; i += NT) dst[i] = tree_lds[i];
  }
}

template <bool WLDS, int NW, int AW, int MM, bool TREC = false>
inline int rz_search_launch_k(const RzSearchArgs& sa, unsigned grid, size_t lds_bytes, stream_t stream) {
  static std::atomic<uint64_t> lds_attr_done{0};   // per instantiation, one bit per device
  if (const int ae = allow_large_lds((const void*)rz_search_kernel<WLDS, NW, AW, MM, TREC>, 160 * 1024, lds_attr_done)) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  hipLaunchKernelGGL((rz_search_kernel<WLDS, NW, AW, MM, TREC>), dim3(grid), dim3(NW * 64), lds_bytes, stream, sa);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("residual search kernel launch failed: %s (grid %u, %d threads, %zu bytes of LDS, %d trees per workgroup, "
              "weights in LDS %d, action lanes %d)", hipGetErrorString(e), grid, NW * 64, lds_bytes, sa.net.T, (int)WLDS, AW);
    return MZX_ERR_RUNTIME;
  }
  return MZX_OK;
}

template <bool WLDS, int NW, int MM>
inline int rz_search_launch_aw(const RzSearchArgs& sa, unsigned grid, size_t lds, stream_t stream) {
  const bool wide = sa.p.num_actions > FUSED_ROW || 2 * sa.p.support_size + 1 > 2 * FUSED_ROW;
  if (wide) return rz_search_launch_k<WLDS, NW, 0, MM>(sa, grid, lds, stream);
  if constexpr (NW == 4 && MM == 3) {   // record-form LDS trees: the small-network kernels (tree side 14-18 % of a simulation)
    if (sa.tree_lds == 2)
      return sa.p.num_actions <= 4 ? rz_search_launch_k<WLDS, NW, 4, MM, true>(sa, grid, lds, stream)
                                   : rz_search_launch_k<WLDS, NW, 16, MM, true>(sa, grid, lds, stream);
  }
  if (sa.p.num_actions <= 4) return rz_search_launch_k<WLDS, NW, 4, MM>(sa, grid, lds, stream);
  return rz_search_launch_k<WLDS, NW, 16, MM>(sa, grid, lds, stream);
}

#endif  // !MZX_HOSTCHECK
}  // namespace mzx

#include "mzx_resnet_wave.h"   // small boards: a wave per tree (needs RzSearchArgs and the fc2 <-> arena conversions above)

namespace mzx {
#ifndef MZX_HOSTCHECK

inline bool rz_search_supported(const mzx_search* s) {
  const mzx_net* net = s->net;
  if (!net || !net->rz.ok || !net->rz.recurrent.ok) return false;
  if (s->p.num_actions > WIDE_MAX_CHUNKS * FUSED_ROW) return false;
  const RzProgram& R = net->rz.recurrent;
  // at least one tree per workgroup must fit beside the search's own LDS
  return 4 * (rz_lds_floats(net->rz.g, R, 1, false) + rz_search_extra_floats(s->p)) <= RZ_LDS_BUDGET;
}

// MCTS.run for B roots, residual network: root by the generic kernels, all simulations in one launch.
inline int rz_search_run(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream,
                   const RootOverride* ov = nullptr) {
  const ArenaView v = arena_view(s, d_arena);
  mzx_net* net = s->net;
  const int B = s->p.num_trees;
  int rc = ensure_tables(s, d_arena, stream);
  if (rc) return rc;

  // ---- root: initial_inference (hidden state -> arena node 0) + root expansion
  const bool ix_init = rz_enabled(net, false);
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

  // ---- every simulation, one launch
  if (s->p.num_sims > 0) {
    const RzProgram& R = net->rz.recurrent;
    NetBuffers nr;
    nr.in = v.arena.hidden; nr.action = nullptr; nr.hidden = v.arena.hidden;
    nr.value = nullptr; nr.reward = nullptr; nr.policy = nullptr; nr.workspace = nullptr;
    NetIndex ir;
    ir.in_nodes = s->p.num_nodes; ir.out_nodes = s->p.num_nodes;
    const int64_t extra = rz_search_extra_floats(s->p);
    RzLaunch L = rz_prepare(net, R, nr.in, nr, B, &ir, 0, extra);
    // trees in LDS when that costs neither trees per workgroup nor the kernel shape (MZX_RZ_TREE_LDS=0: A/B knob)
    const int64_t slab_floats = 2 * (int64_t)(s->L.tree_bytes >> 3);
    bool tree_lds = false;
    if (rz_env_int("MZX_RZ_TREE_LDS", 1) != 0) {
      const RzLaunch L1 = rz_prepare(net, R, nr.in, nr, B, &ir, slab_floats, extra);
      const bool fits = (int64_t)L1.lds + 4 * (extra + slab_floats * L1.a.T) <= RZ_LDS_BUDGET;   // (one tree per workgroup is not re-checked by rz_prepare)
      if (fits && L1.a.T == L.a.T && L1.eight == L.eight && L1.small == L.small) { L = L1; tree_lds = true; }
    }
    // record-form trees (mzx_fused_fc2.h) for the small-network kernels when they fit beside the engine's image
    bool tree_rec = false;
    size_t rec_lds = 0;
    const int A_ = s->p.num_actions;
    const bool narrow = A_ <= FUSED_ROW && 2 * s->p.support_size + 1 <= 2 * FUSED_ROW;
    if (narrow && L.small && !L.eight && rz_env_int("MZX_RZ_TREE_REC", 1) != 0) {
      const int RW = A_ <= 4 ? 4 : 16;
      const int64_t rec_floats = rz_trec_tree_floats(s->p.num_nodes, RW);
      const int64_t extra2 = extra + rz_trec_extra_floats(s->p.num_nodes);
      const RzLaunch L2 = rz_prepare(net, R, nr.in, nr, B, &ir, rec_floats, extra2);
      const bool fits = (int64_t)L2.lds + 4 * (extra2 + rec_floats * L2.a.T) <= RZ_LDS_BUDGET;
      if (fits && L2.small && !L2.eight && L2.a.T >= (tree_lds ? L.a.T : 1)) {
        L = L2; tree_rec = true; tree_lds = false;
        rec_lds = (size_t)4 * (extra2 + rec_floats * L2.a.T);
      }
    }
    {   // small boards: a wave per tree, no workgroup barriers (mzx_resnet_wave.h)
      RzWaveArgs wa;
      memset(&wa, 0, sizeof(wa));
      unsigned wgrid = 0;
      size_t wlds = 0;
      const bool wave_k = rz_wave_plan(s, R, L.a, wa, wgrid, wlds);
      const bool tile_k = !wave_k && rz_tile_plan(s, R, L.a, wa, wgrid, wlds);
      if (wave_k || tile_k) {
        wa.s.p = v.p;
        wa.s.L = s->L;
        wa.s.trees = v.arena.trees;
        wa.s.tape = io->d_tape;
        wa.s.num_sims = s->p.num_sims;
        wa.s.sim0 = 0;
        wa.s.prof = ((s->mode & 8) && s->ws_floats >= (int64_t)wgrid * 8) ? (uint32_t*)((char*)d_arena + s->off_ws) : nullptr;
        wa.s.tree_lds = 2;
        s->last_kernel = wave_k ? "mzx::rz_wave_search_kernel" : "mzx::rz_tile_search_kernel";
        rc = wave_k ? rz_wave_launch(wa, wgrid, wlds, stream) : rz_tile_launch(wa, wgrid, wlds, stream);
        if (rc) return rc;
        return search_finish(s, io, d_arena, stream);
      }
    }
    s->last_kernel = "mzx::rz_search_kernel";
    RzSearchArgs sa;
    sa.net = L.a;
    sa.p = v.p;
    sa.L = s->L;
    sa.trees = v.arena.trees;
    sa.tape = io->d_tape;
    sa.num_sims = s->p.num_sims;
    sa.sim0 = 0;
    // mode flag 8: per-workgroup phase cycle counters in the (otherwise unused) network workspace region
    sa.prof = ((s->mode & 8) && s->ws_floats >= (int64_t)L.grid * 8) ? (uint32_t*)((char*)d_arena + s->off_ws) : nullptr;
    sa.tree_lds = tree_rec ? 2 : (tree_lds ? 1 : 0);
    const size_t lds = tree_rec ? L.lds + rec_lds : L.lds + (size_t)4 * (extra + (tree_lds ? slab_floats * L.a.T : 0));
    if (L.small) rc = L.wlds ? rz_search_launch_aw<true, 4, 3>(sa, L.grid, lds, stream) : rz_search_launch_aw<false, 4, 3>(sa, L.grid, lds, stream);
    else if (L.wlds) rc = L.eight ? rz_search_launch_aw<true, 8, 8>(sa, L.grid, lds, stream) : rz_search_launch_aw<true, 4, 8>(sa, L.grid, lds, stream);
    else rc = L.eight ? rz_search_launch_aw<false, 8, 8>(sa, L.grid, lds, stream) : rz_search_launch_aw<false, 4, 8>(sa, L.grid, lds, stream);
    if (rc) return rc;
  }
  return search_finish(s, io, d_arena, stream);
}

#endif  // !MZX_HOSTCHECK

}  // namespace mzx
