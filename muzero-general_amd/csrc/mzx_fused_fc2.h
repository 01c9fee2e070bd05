// mzx_fused_fc2.h -- second generation of the whole-search kernel for fully connected networks
// (K5 of SURVEY.md section 8a): same contract as mzx_fused_fc.h -- initial_inference, root expansion and
// num_simulations x {select, recurrent_inference, expand, backpropagate} in ONE launch, every tree resident
// in LDS, bit-identical tree statistics to the generic path (mzx_tree.h) -- with the selection walk, which was
// 46 % of a simulation, rebuilt around what actually changes between two visits of a node.
//
// Reference semantics: MCTS.run /root/reference/self_play.py:260-361, ucb_score :380-404,
// backpropagate :406-430, MinMaxStats :553-570.
//
//   ucb_score(parent, child) = pb_c(N, n) * prior  +  [n > 0] normalize(reward + discount * (+-value))
//
//   * prior score  ps = (pbc[N] * (sqrt[N] / (n + 1))) * prior  depends on the parent's visit count N and the
//     child's n only: it changes exactly for the slots of the nodes on the path just back-propagated.  It is
//     CACHED per child slot and refreshed right after back-propagation by the lane that owns the path node
//     (N is in its registers), A slots per lane.  The walk no longer evaluates it: no table look-ups chained
//     behind the slot read, no division.
//   * value score  normalize(q) = (q - min) / (max - min)  depends on the tree-wide MinMaxStats, which move in
//     about every second simulation -- it cannot be cached, but its DIVISOR is the same for every node of a
//     simulation.  hipcc expands a binary64 division into div_scale, rcp, two Newton steps on the reciprocal,
//     then mul / fma / div_fmas / div_fixup on the numerator (AMDGPU LowerFDIV64); the first half depends on the
//     divisor only and is hoisted out of the walk (`recip_refined`, once per simulation); per child remain
//     mul, fma, fma, div_fixup -- the SAME instructions on the same operands, hence the same bits.  (div_scale
//     is the identity unless an operand is denormal or the exponents differ by more than 2^9 orders of
//     magnitude: tree statistics are sums of at most num_simulations fp32-origin values, so neither occurs.)
//     The two divisions by small integers (sqrt[N] / (n + 1), value_sum / visit_count) use the same split with a
//     per-integer reciprocal table filled at kernel start by the same instructions.
//   * one LDS round trip per level: a child slot is ONE 32-byte record {prior score, q, visit count, child
//     link, prior}; for two-action games the argmax needs no wave ballot (both scores are broadcast in the row).
//   * the path of a walk is kept in an LDS array instead of "lane d remembers depth d", and back-propagation
//     runs over 16-level chunks of it: searches deeper than 16 plies no longer fall back to a serial lane.
//
// Mapping, network engines (SmallNet / LdsNet) and the canonical fp32 reduction order are those of
// mzx_fused_fc.h.  The trees are converted to the arena's TreeLayout on export (mode flag 2), so
// mzx_search_dump and the bit-identity tests see the same data as with the generic path.
#pragma once
#include "mzx_fused_fc.h"

namespace mzx {

struct Fc2Args {
  // roots the caller expanded itself (MCTS.run(..., override_root_with=root), self_play.py:275-277,
  // diagnose_model.py:57-74): hidden state / child priors in legal-action order / reward replace initial_inference
  const float* ov_hidden;    // [B][E]   (null: the kernel runs initial_inference)
  const double* ov_priors;   // [B][A]
  const double* ov_reward;   // [B]
  FusedFcArgs f;   // network description, search parameters, io, export pointers (f.L = arena TreeLayout)
  int32_t off_slots, off_nodes, off_path, off_roota, off_mm, off_hidden, off_scratch;  // byte offsets inside a tree's slab
  int32_t tree_stride, lds_inv;                                               // lds_inv: reciprocal table (bytes)
};

#ifndef MZX_HOSTCHECK

struct __attribute__((aligned(16))) Fc2Slot {   // one child slot of a node
  double ps;       // cached prior score of the slot at the parent's current visit count
  double q;        // reward + discount * (+-value()) of the child (written by back-propagation)
  int32_t n;       // child's visit count
  int32_t child;   // canonical node index or -1
  double prior;
};
struct __attribute__((aligned(16))) Fc2Node {
  double value_sum, reward;
  int32_t visit, to_play, parent, parent_slot;
};
static_assert(sizeof(Fc2Slot) == 32 && sizeof(Fc2Node) == 32, "record size");

// Divisor-only half of hipcc's binary64 division expansion: v_rcp_f64 + two Newton steps.
__device__ __forceinline__ double recip_refined(double b) {
  const double y0 = __builtin_amdgcn_rcp(b);
  const double f0 = __builtin_fma(-b, y0, 1.0);
  const double f1 = __builtin_fma(y0, f0, y0);
  const double f2 = __builtin_fma(-b, f1, 1.0);
  return __builtin_fma(f1, f2, f1);
}
// Numerator half: a / b given y = recip_refined(b).
__device__ __forceinline__ double div_by(double a, double b, double y) {
  const double m = a * y;
  const double r = __builtin_fma(-b, m, a);
  const double q = __builtin_fma(r, y, m);
  return __builtin_amdgcn_div_fixup(q, b, a);
}
// prior_score of ucb_score (self_play.py:384-392), operation order of ucb_from (mzx_tree.h)
__device__ __forceinline__ double prior_score(double pbcN, double sqN, int n, double inv_n1, double prior) {
  const double pb_c = pbcN * div_by(sqN, (double)(n + 1), inv_n1);
  return pb_c * prior;
}

// Butterflies over the first W lanes of a row (the other lanes hold the neutral element): the canonical
// order xor 1, xor 2, half-mirror, mirror cut after log2(W) steps -- the skipped steps would add zeros / compare
// with the neutral element, so the result has the same bits as the 16-lane form.
template <int W>
__device__ __forceinline__ float row_sum_w(float v) {
  v = v + dpp_f<DPP_XOR1>(v);
  if constexpr (W > 2) v = v + dpp_f<DPP_XOR2>(v);
  if constexpr (W > 4) v = v + dpp_f<DPP_HALF_MIRROR>(v);
  if constexpr (W > 8) v = v + dpp_f<DPP_MIRROR>(v);
  return v;
}
template <int W>
__device__ __forceinline__ float row_max_w(float v) {
  v = fmaxf(v, dpp_f<DPP_XOR1>(v));
  if constexpr (W > 2) v = fmaxf(v, dpp_f<DPP_XOR2>(v));
  if constexpr (W > 4) v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
  if constexpr (W > 8) v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
  return v;
}
constexpr int DPP_ROW_SHL1 = 0x101;   // lane i reads lane i + 1 of its row (lane 15: keeps `old`)
__device__ __forceinline__ int shl1_i(int v, int old) {
  return __builtin_amdgcn_update_dpp(old, v, DPP_ROW_SHL1, 0xF, 0xF, false);
}
__device__ __forceinline__ double shl1_d(double v, double old) {
  return __hiloint2double(shl1_i(__double2hiint(v), __double2hiint(old)), shl1_i(__double2loint(v), __double2loint(old)));
}

// One step of the value recurrence  value = (+-reward) + discount * value  (self_play.py:417 / :424-427) at path
// lane J: the lane notes the value arriving at its node, every lane advances the row-uniform value.  Rows whose
// leaf is above lane J (J > ld) keep theirs -- a select, no exec-mask branch.
template <int J>
__device__ __forceinline__ void chain_step2(double r_eff, double disc, int ld, int sub, double& val, double& my_in) {
  const double rj = bcast_d<J>(r_eff);
  const double nv = rj + disc * val;
  my_in = (sub == J) ? val : my_in;
  val = (J <= ld) ? nv : val;
}
// Steps 15 .. 1 in groups guarded by the deepest row of the wave (wmax, wave-uniform: plain scalar branches, no
// exec masking); a step above a row's own leaf is a no-op by its select, so a group may run a step too many.
__device__ __forceinline__ void value_chain(double r_eff, double disc, int ld, int sub, int wmax, double& val, double& my_in) {
  if (wmax >= 13) {
    chain_step2<15>(r_eff, disc, ld, sub, val, my_in);
    chain_step2<14>(r_eff, disc, ld, sub, val, my_in);
    chain_step2<13>(r_eff, disc, ld, sub, val, my_in);
  }
  if (wmax >= 9) {
    chain_step2<12>(r_eff, disc, ld, sub, val, my_in);
    chain_step2<11>(r_eff, disc, ld, sub, val, my_in);
    chain_step2<10>(r_eff, disc, ld, sub, val, my_in);
    chain_step2<9>(r_eff, disc, ld, sub, val, my_in);
  }
  if (wmax >= 7) {
    chain_step2<8>(r_eff, disc, ld, sub, val, my_in);
    chain_step2<7>(r_eff, disc, ld, sub, val, my_in);
  }
  if (wmax >= 5) {
    chain_step2<6>(r_eff, disc, ld, sub, val, my_in);
    chain_step2<5>(r_eff, disc, ld, sub, val, my_in);
  }
  if (wmax >= 4) chain_step2<4>(r_eff, disc, ld, sub, val, my_in);
  if (wmax >= 3) chain_step2<3>(r_eff, disc, ld, sub, val, my_in);
  if (wmax >= 2) chain_step2<2>(r_eff, disc, ld, sub, val, my_in);
  if (wmax >= 1) chain_step2<1>(r_eff, disc, ld, sub, val, my_in);
}


// ---------------------------------------------------------------------------
// The tree side of one simulation as row-collective functions over LDS records (one 16-lane row per tree): used by
// the fully connected kernel below and by the residual whole-search kernel (mzx_resnet_search.h) when its trees
// are LDS-resident.

struct Fc2Tree {              // LDS views of one tree + the workgroup's tables
  Fc2Slot* slots;             // [NN][AW]
  Fc2Node* nodes;             // [NN]
  int2* path;                 // [NN + 1]: {node at depth d, slot taken from its parent}
  int32_t* roota;             // [AW]: action of root slot s
  double* mm;                 // MinMaxStats {minimum, maximum}: updated by LDS min / max
  const double *pbc, *sqt, *inv_y;
  double pb_leaf, disc;       // pb_c(N = 1, n = 0): prior-score factor of a fresh leaf's slots
  int A, NN, P;
};
struct Fc2Row { int32_t n_nodes, tape_pos, flags, ties, max_depth, sum_depth, root_n, root_to_play; };
struct Fc2Walk { int parent, slot, leaf, depth, levels, vtp, action; };

// Selection walk (self_play.py:325-334, :363-404).  Slots that do not exist (root slots beyond the legal actions,
// padding up to AW) carry a prior score of -inf and n = 0, so their score is -inf without a validity test.  A row
// that has reached its leaf keeps executing with its state frozen (one straight-line body per level per wave).
template <int AW>
__device__ __forceinline__ Fc2Walk fc2_walk(const Fc2Tree& T, Fc2Row& st, const uint32_t* tape, int tape_words, int sub,
                                            int row_in_wave) {
  const double2 mmv = *(const double2*)T.mm;   // MinMaxStats after the previous back-propagation
  const double mn = mmv.x, mx = mmv.y;
  const double dd = mx - mn;                 // MinMaxStats.normalize divisor, same for every node of this walk
  const double yd = recip_refined(dd);       // (garbage while max <= min: results discarded like the reference's branch)
  const bool norm_on = mx > mn;
  int node = 0, depth = 0, slot = 0;
  int levels = 0;                  // walk iterations = depth of the deepest leaf in this wave (scalar)
  bool done = false;
  for (;;) {
    ++levels;
    const Fc2Slot* rp = T.slots + (node * AW + (sub & (AW - 1)));
    const double ps = rp->ps, q = rp->q;
    const int n = rp->n, c = rp->child;
    const double nv = div_by(q - mn, dd, yd);
    const double v = norm_on ? nv : q;
    const double wv = ps + v;
    const double sc = (n > 0) ? wv : ps;
    int sl, cw;
    if constexpr (AW == 2) {
      // two candidates: every lane sees both scores; no ballot
      const double a0 = bcast_d<0>(sc), a1 = bcast_d<1>(sc);
      sl = (a1 > a0) ? 1 : 0;
      if (__builtin_expect(a0 == a1 && !done, 0)) {  // numpy.random.choice([0, 1]): first walk of a search, rare later
        ++st.ties;
        sl = tape_draw(tape, tape_words, st.tape_pos, st.flags, 2);
      }
      const int c0 = bcast_i<0>(c), c1 = bcast_i<1>(c);
      cw = sl ? c1 : c0;
    } else {
      const double best = row_max_d<AW>(sc);
      const unsigned bits = row_bits(__ballot(sc == best), row_in_wave) & ((1u << AW) - 1u);
      const int nbest = __popc(bits);
      sl = nbest ? (__ffs(bits) - 1) : 0;
      if (__builtin_expect(nbest > 1 && !done, 0)) {  // numpy.random.choice(ties): k-th maximiser in slot order
        ++st.ties;
        int k = tape_draw(tape, tape_words, st.tape_pos, st.flags, nbest);
        unsigned b = bits;
        for (; k > 0; --k) b &= b - 1;
        sl = __ffs(b) - 1;
      }
      if constexpr (AW <= 4) cw = pick_i<AW>(c, sl);
      else cw = perm_i(c, sl, row_in_wave);
    }
    const bool act = !done;
    // entry of the level just decided; a finished row rewrites the entry beyond its leaf (never read)
    if (sub == 0) T.path[depth + 1] = make_int2(cw, sl);
    depth += act ? 1 : 0;
    slot = act ? sl : slot;
    node = (act && cw >= 0) ? cw : node;
    done = done || (cw < 0);
    if (__all(done)) break;
  }
  Fc2Walk w;
  // players play turn by turn (self_play.py:331-334): the leaf's player follows from the depth
  w.vtp = (T.P == 1) ? 0 : ((st.root_to_play + depth) & 1);
  int leaf = st.n_nodes;
  if (leaf >= T.NN) { st.flags |= TF_NODE_OVERFLOW; leaf = T.NN - 1; }
  if (sub == 0) T.path[depth] = make_int2(leaf, slot);
  const int ra = T.roota[slot < AW ? slot : 0];
  w.parent = node; w.slot = slot; w.leaf = leaf; w.depth = depth; w.levels = levels;
  w.action = (node == 0) ? ra : slot;
  return w;
}

// What back-propagation needs about the path node a lane owns (lane j of chunk c <-> depth 16 c + j).
template <int AW>
struct Fc2Lane {
  int nd, pslot, par, vc, tp;
  double vs, rr, inv_vc2, inv_vc3, pb, sv;
  int sn[AW <= 4 ? AW : 1];
  double sprior[AW <= 4 ? AW : 1], sinv[AW <= 4 ? AW : 1];
};

// Everything back-propagation needs from the tree is independent of the network: fetched BEFORE it, so that the
// LDS latency hides behind recurrent_inference.  (Call after a wave_sync following fc2_walk.)
template <int AW>
__device__ __forceinline__ Fc2Lane<AW> fc2_load_lane(const Fc2Tree& T, const Fc2Walk& w, int c, int sub) {
  Fc2Lane<AW> L;
  const int d = c * 16 + sub;
  const bool active = d <= w.depth, is_leaf = d == w.depth;
  L.nd = 0; L.pslot = 0; L.par = 0; L.vc = 0; L.tp = w.vtp; L.vs = 0.0; L.rr = 0.0;
  if (active) {
    const int2 pe = T.path[d];
    L.nd = pe.x; L.pslot = pe.y;
    if (d > 0) L.par = T.path[d - 1].x;
  }
  if (active && !is_leaf) {
    const Fc2Node* np = T.nodes + L.nd;
    L.vs = np->value_sum; L.rr = np->reward; L.vc = np->visit;
    if (T.P == 2) L.tp = np->to_play;
  }
  L.inv_vc2 = T.inv_y[L.vc + 1];        // reciprocal of this node's visit count after the update
  L.inv_vc3 = T.inv_y[L.vc + 2];        // ... and of (that + 1): what its PARENT's prior score divides by
  L.pb = T.pbc[L.vc + 1]; L.sv = T.sqt[L.vc + 1];
  if constexpr (AW <= 4) {
#pragma unroll
    for (int s = 0; s < AW; ++s) {
      L.sn[s] = 0; L.sprior[s] = 0.0;
      if (active && !is_leaf && s < T.A) { L.sn[s] = T.slots[L.nd * AW + s].n; L.sprior[s] = T.slots[L.nd * AW + s].prior; }
    }
#pragma unroll
    for (int s = 0; s < AW; ++s) L.sinv[s] = T.inv_y[L.sn[s] + 1];
  }
  return L;
}

// Node.expand (self_play.py:451-465): child slot `sub` of the new leaf (lanes < AW; `in`: the slot exists).
template <int AW>
__device__ __forceinline__ void fc2_expand(const Fc2Tree& T, int leaf, int sub, bool in, double prior) {
  if (sub < AW) {
    Fc2Slot s;
    s.prior = in ? prior : 0.0;
    s.q = 0.0; s.n = 0; s.child = -1;
    s.ps = in ? T.pb_leaf * s.prior : -MZX_INF;   // prior score once the leaf has its first visit (N = 1, n = 0)
    T.slots[leaf * AW + sub] = s;
  }
}

// MCTS.backpropagate (self_play.py:406-430) + refresh of the cached prior scores along the path.  `L` = the
// lane's operands of chunk w.levels >> 4 (fc2_load_lane before the network).
template <int AW>
__device__ __forceinline__ void fc2_backprop(const Fc2Tree& T, Fc2Row& st, const Fc2Walk& w, Fc2Lane<AW> L, int sub,
                                             double value, double reward) {
  const int depth = w.depth, vtp = w.vtp, P = T.P;
  const double disc = T.disc;
  st.n_nodes = w.leaf + 1;
  if (depth > st.max_depth) st.max_depth = depth;
  st.sum_depth += depth;
  const int cmax = w.levels >> 4;    // wave-uniform number of 16-level chunks - 1 (0 unless a path is > 15 deep)
  double val = value;
  int carry_vc2 = 0, carry_pslot = -1;       // lane 0 of the chunk below (deeper), for lane 15 of this one
  double carry_inv = 0.0;
  for (int c = cmax; c >= 0; --c) {
    if (c != cmax) { wave_sync(); L = fc2_load_lane<AW>(T, w, c, sub); }
    const int d = c * 16 + sub;
    const int ld = depth - c * 16;             // row-uniform: depth of the leaf relative to this chunk
    const bool active = d <= depth, is_leaf = d == depth;
    const double rr = is_leaf ? reward : L.rr;
    const bool same = (L.tp == vtp);
    const double r_eff = (P == 1 || !same) ? rr : -rr;   // value = (+-reward) + discount * value
    double my_in = val;
    const int wm = w.levels - c * 16;           // deepest leaf of the wave relative to this chunk (wave-uniform)
    value_chain(r_eff, disc, ld, sub, wm > 15 ? 15 : (wm < 0 ? 0 : wm), val, my_in);
    if (sub == 0) my_in = val;
    if (c > 0 && ld >= 0) val = bcast_d<0>(r_eff) + disc * val;   // hand the value to the chunk above
    const int vc2 = L.vc + 1;
    const double vs2 = L.vs + ((P == 1 || same) ? my_in : -my_in);
    double qv = 0.0;
    if (active) {
      const double mean = div_by(vs2, (double)vc2, L.inv_vc2);
      qv = rr + disc * ((P == 1) ? mean : -mean);
      if (is_leaf) {
        Fc2Node r;
        r.value_sum = vs2; r.reward = reward; r.visit = vc2; r.to_play = vtp; r.parent = L.par; r.parent_slot = L.pslot;
        T.nodes[L.nd] = r;
        T.slots[L.par * AW + L.pslot].child = L.nd;
      } else {
        T.nodes[L.nd].value_sum = vs2;
        T.nodes[L.nd].visit = vc2;
      }
      if (d > 0) {
        Fc2Slot* ps = T.slots + (L.par * AW + L.pslot);
        ps->q = qv;
        ps->n = vc2;
      }
      // MinMaxStats.update (self_play.py:562-564): pure min / max over the path nodes, order-free
      __hip_atomic_fetch_min(&T.mm[0], qv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_max(&T.mm[1], qv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // prior scores of this node's child slots at its new visit count N = vc2 (see header).  The slot on the
    // path just received the child's new visit count: taken from the lane above instead of LDS.
    if constexpr (AW <= 4) {
      const int ch_vc2 = shl1_i(vc2, carry_vc2), ch_slot = shl1_i(L.pslot, carry_pslot);
      const double ch_inv = shl1_d(L.inv_vc3, carry_inv);
      if (active && !is_leaf) {
        const int nslots = (L.nd == 0) ? st.root_n : T.A;
#pragma unroll
        for (int s = 0; s < AW; ++s) {
          if (s < nslots) {
            const bool on_path = (s == ch_slot);
            const int ns = on_path ? ch_vc2 : L.sn[s];
            const double iv = on_path ? ch_inv : L.sinv[s];
            T.slots[L.nd * AW + s].ps = prior_score(L.pb, L.sv, ns, iv, L.sprior[s]);
          }
        }
      }
      carry_vc2 = bcast_i<0>(vc2); carry_pslot = bcast_i<0>(L.pslot); carry_inv = bcast_d<0>(L.inv_vc3);
    } else {
      wave_sync();
      if (active && !is_leaf) {
        Fc2Slot* sp = T.slots + L.nd * AW;
        const int nslots = (L.nd == 0) ? st.root_n : T.A;
        for (int s = 0; s < nslots; ++s) {
          const int ns = sp[s].n;
          sp[s].ps = prior_score(L.pb, L.sv, ns, T.inv_y[ns + 1], sp[s].prior);
        }
      }
    }
  }
  wave_sync();
}

template <class Net, int AW, bool PROFILE>
__global__ void __launch_bounds__(256) fc2_search_kernel(const Fc2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int sub = tid & (FUSED_ROW - 1);
  const int row = tid / FUSED_ROW;
  const int row_in_wave = row & 3;
  const int tree = blockIdx.x * a.f.trees_per_block + row;
  const int A = a.f.p.num_actions, E = a.f.E, NN = a.f.p.num_nodes, P = a.f.p.num_players;
  const double disc = a.f.p.discount;
  uint32_t prof[FUSED_PROF_WORDS];
  unsigned long long t_last = 0;
  if (PROFILE) {
    for (int k = 0; k < FUSED_PROF_WORDS; ++k) prof[k] = 0;
    t_last = __builtin_readcyclecounter();
  }

  // ---- stage tables (+ weights): the only workgroup-wide barrier
  Net net;
  double* tables = (double*)(smem + a.f.lds_tables);
  double* inv_y = (double*)(smem + a.lds_inv);   // [NN + 2]: refined reciprocal of the integer k >= 1
  const int ntab = 2 * (NN + 1);
  for (int i = tid; i < ntab; i += blockDim.x) tables[i] = a.f.tables[i];
  for (int i = tid; i < NN + 2; i += blockDim.x) inv_y[i] = recip_refined((double)(i > 0 ? i : 1));
  net.stage(a.f, smem, tid);
  __syncthreads();
  if (row >= a.f.trees_per_block || tree >= a.f.p.num_trees) return;  // whole rows exit together
  net.setup(a.f, smem, sub);
  const double* pbc = tables;
  const double* sqt = tables + (NN + 1);
  // prior-score factor of a freshly expanded leaf's slots: the leaf has ONE visit after its own
  // back-propagation and its children none -- pb_c(N = 1, n = 0)
  const double pb_leaf = pbc[1] * div_by(sqt[1], 1.0, inv_y[1]);

  char* slab = smem + a.f.lds_trees + (size_t)row * a.tree_stride;
  Fc2Slot* slots = (Fc2Slot*)(slab + a.off_slots);   // [NN][AW]
  Fc2Node* nodes = (Fc2Node*)(slab + a.off_nodes);   // [NN]
  int2* path = (int2*)(slab + a.off_path);           // [NN + 1]: {node at depth d, slot taken from its parent}
  int32_t* roota = (int32_t*)(slab + a.off_roota);   // [AW]: action of root slot s
  double* mm = (double*)(slab + a.off_mm);           // MinMaxStats {minimum, maximum}: updated by LDS min / max
  float* hidden = (float*)(slab + a.off_hidden);     // [NN][E]
  float* scr = (float*)(slab + a.off_scratch);
  const uint32_t* tape = (const uint32_t*)a.f.io.d_tape + (size_t)tree * a.f.p.tape_words;
  const int tape_words = a.f.p.tape_words;

  // per-tree scalars, row-uniform registers
  int32_t n_nodes = 1, tape_pos = 0, flags = 0, ties = 0, max_depth = 0, sum_depth = 0, root_n = 0, root_to_play = 0;
  MZX_PROF(0)

  // ---- initial_inference (models.py:172-190) + root expansion (self_play.py:286-314, :467-476)
  {
    NetOut o;
    const bool given = a.ov_hidden != nullptr;       // (launch-uniform) the caller's roots: no initial_inference
    if (!given) net.initial(a.f.io.d_observation + (size_t)tree * a.f.in_size, hidden, scr, sub, o);
    else
      for (int e = sub; e < E; e += FUSED_ROW) hidden[e] = a.ov_hidden[(size_t)tree * E + e];
    const int32_t* lg = a.f.io.d_legal_actions + (size_t)tree * A;
    const double* nz = a.f.io.d_noise ? a.f.io.d_noise + (size_t)tree * A : nullptr;
    while (root_n < A && lg[root_n] >= 0) ++root_n;
    if (!given && sub < A) scr[sub] = o.policy;
    wave_sync();
    const bool in = sub < root_n;
    const float l = (in && !given) ? scr[lg[sub]] : -MZX_INF;   // logits gathered in the game's legal-action order
    const float m = row_max_w<AW>(l);
    const float e = (in && !given) ? mzx_expf(l - m) : 0.f;
    const float den = row_sum_w<AW>(e);
    root_to_play = a.f.io.d_to_play[tree];
    if (sub == 0) {
      Fc2Node r;
      r.value_sum = 0.0; r.visit = 0; r.to_play = root_to_play;
      r.reward = given ? a.ov_reward[tree] : (double)support_inverse_transform(0.0f);
      r.parent = -1; r.parent_slot = -1;
      nodes[0] = r;
      path[0] = make_int2(0, -1);
      mm[0] = MZX_INF; mm[1] = -MZX_INF;      // MinMaxStats (self_play.py:558-560)
      if (!given && a.f.io.d_root_predicted_value) a.f.io.d_root_predicted_value[tree] = (double)o.value;
    }
    if (sub < AW) {
      Fc2Slot s;
      const double pr = !in ? 0.0 : (given ? a.ov_priors[(size_t)tree * A + sub] : (double)mzx_div(e, den));
      s.prior = in ? root_noisy_prior(pr, nz, sub, a.f.p.exploration_fraction) : 0.0;
      s.q = 0.0; s.n = 0; s.child = -1;
      s.ps = in ? prior_score(pbc[0], sqt[0], 0, inv_y[1], s.prior) : -MZX_INF;   // root visit count 0
      slots[sub] = s;
      roota[sub] = in ? lg[sub] : -1;
    }
    wave_sync();
  }
  MZX_PROF(1)

  // ---- simulations (self_play.py:319-355)
  Fc2Tree T;
  T.slots = slots; T.nodes = nodes; T.path = path; T.roota = roota; T.mm = mm;
  T.pbc = pbc; T.sqt = sqt; T.inv_y = inv_y; T.pb_leaf = pb_leaf; T.disc = disc; T.A = A; T.NN = NN; T.P = P;
  Fc2Row st;
  st.n_nodes = n_nodes; st.tape_pos = tape_pos; st.flags = flags; st.ties = ties; st.max_depth = max_depth;
  st.sum_depth = sum_depth; st.root_n = root_n; st.root_to_play = root_to_play;
  const int num_sims = a.f.p.num_sims;
  for (int sim = 0; sim < num_sims; ++sim) {
    const Fc2Walk w = fc2_walk<AW>(T, st, tape, tape_words, sub, row_in_wave);
    MZX_PROF(2)
    wave_sync();
    const Fc2Lane<AW> L = fc2_load_lane<AW>(T, w, w.levels >> 4, sub);
    MZX_PROF(7)

    // ------------------------------------------------------------- recurrent_inference (models.py:192-195)
    NetOut o;
    net.recurrent(hidden + w.parent * E, w.action, hidden + w.leaf * E, scr, sub, o);
    MZX_PROF(3)

    // ------------------------------------------------------------- expand (self_play.py:451-465)
    {
      const bool in = sub < A;
      const float m = row_max_w<AW>(in ? o.policy : -MZX_INF);
      const float e = in ? mzx_expf(o.policy - m) : 0.f;
      const float den = row_sum_w<AW>(e);
      fc2_expand<AW>(T, w.leaf, sub, in, (double)mzx_div(e, den));
    }
    MZX_PROF(4)

    // ------------------------------------------------------------- backpropagate (self_play.py:406-430)
    fc2_backprop<AW>(T, st, w, L, sub, (double)o.value, (double)o.reward);
    MZX_PROF(5)
  }
  n_nodes = st.n_nodes; tape_pos = st.tape_pos; flags = st.flags; ties = st.ties; max_depth = st.max_depth;
  sum_depth = st.sum_depth;

  // ---- results (FinalizeOp)
  if (sub == 0) {
    for (int x = 0; x < A; ++x) a.f.io.d_visit_counts[(size_t)tree * A + x] = 0;
    for (int s = 0; s < root_n; ++s) a.f.io.d_visit_counts[(size_t)tree * A + roota[s]] = slots[s].n;
    const int vc = nodes[0].visit;
    a.f.io.d_root_value[tree] = (vc == 0) ? 0.0 : nodes[0].value_sum / (double)vc;
    a.f.io.d_info[tree * 4 + 0] = max_depth;
    a.f.io.d_info[tree * 4 + 1] = flags;
    a.f.io.d_info[tree * 4 + 2] = tape_pos;
    a.f.io.d_info[tree * 4 + 3] = sum_depth;
  }
  if (a.f.export_trees) {  // parity / diagnose export: LDS records -> the arena's TreeLayout
    TreeRef t;
    t.base = a.f.export_trees + (size_t)tree * a.f.L.tree_bytes;
    t.L = a.f.L;
    for (int n = sub; n < n_nodes; n += FUSED_ROW) {
      const Fc2Node r = nodes[n];
      t.value_sum(n) = r.value_sum; t.reward(n) = r.reward; t.visit(n) = r.visit; t.to_play(n) = r.to_play;
      t.parent(n) = r.parent; t.parent_slot(n) = r.parent_slot;
      for (int s = 0; s < A; ++s) {
        const Fc2Slot q = slots[n * AW + s];
        t.prior(n, s) = q.prior; t.slot_q(n, s) = q.q; t.slot_visit(n, s) = q.n; t.child(n, s) = q.child;
      }
    }
    if (sub == 0) {
      for (int k = 0; k < TM_WORDS; ++k) t.meta(k) = 0;
      t.mm_min() = mm[0]; t.mm_max() = mm[1];
      t.meta(TM_N_NODES) = n_nodes; t.meta(TM_TAPE_POS) = tape_pos; t.meta(TM_FLAGS) = flags;
      t.meta(TM_TIE_DRAWS) = ties; t.meta(TM_MAX_DEPTH) = max_depth; t.meta(TM_SUM_DEPTH) = sum_depth;
      t.meta(TM_ROOT_N) = root_n;
      for (int s = 0; s < A; ++s) t.root_action(s) = roota[s];
    }
    float* hd = a.f.export_hidden + (size_t)tree * NN * E;
    for (int i = sub; i < n_nodes * E; i += FUSED_ROW) hd[i] = hidden[i];
  }
  MZX_PROF(6)
  if (PROFILE && sub == 0 && a.f.prof)
    for (int k = 0; k < FUSED_PROF_WORDS; ++k) a.f.prof[(size_t)tree * FUSED_PROF_WORDS + k] = prof[k];
}

// ---------------------------------------------------------------------------
// host side

struct Fc2Plan {
  Fc2Args args;
  int lds_bytes = 0, ok = 0, small = 0, aw = 0;
};

inline Fc2Plan fc2_plan(const mzx_search* s, bool allow_small = true) {
  Fc2Plan P;
  const FusedPlan base = fused_plan(s, allow_small);   // network recovery + shape limits are shared
  if (!base.ok) return P;
  Fc2Args& a = P.args;
  memset(&a, 0, sizeof(a));
  a.f = base.args;
  P.small = base.small;
  const int A = s->p.num_actions, N = s->p.num_nodes, E = a.f.E;
  const int AW = A <= 2 ? 2 : (A <= 4 ? 4 : 16);
  P.aw = AW;
  auto al16 = [](int64_t x) { return (x + 15) & ~int64_t(15); };
  int64_t o = 0;
  a.f.lds_tables = (int32_t)o;  o += al16(int64_t(16) * (N + 1));
  a.lds_inv = (int32_t)o;       o += al16(int64_t(8) * (N + 2));
  a.f.lds_weights = (int32_t)o; o += P.small ? 0 : al16(int64_t(4) * s->net->num_params);
  a.f.lds_trees = (int32_t)o;
  int64_t t = 0;
  a.off_slots = (int32_t)t;   t += int64_t(32) * N * AW;
  a.off_nodes = (int32_t)t;   t += int64_t(32) * N;
  a.off_path = (int32_t)t;    t += al16(int64_t(8) * (N + 1));
  a.off_roota = (int32_t)t;   t += al16(int64_t(4) * AW);
  a.off_mm = (int32_t)t;      t += 16;
  a.off_hidden = (int32_t)t;  t += al16(int64_t(4) * N * E);
  a.off_scratch = (int32_t)t; t += int64_t(4) * (P.small ? 16 : FUSED_SCRATCH);
  // the four rows of a wave touch the same offsets of four consecutive slabs in one instruction:
  // keep the slab stride off the multiples of 256 bytes so that they land in different bank groups
  // (bank = (address / 4) mod 64): a stride of 64 (mod 256) bytes puts the rows' records into four different
  // bank groups; with 0 or 128 (mod 256) two rows collide on every access (measured: 43 % of the LDS cycles
  // were bank conflicts at a stride of 128 mod 256, profiles/r02_rocprof_fc2_c2_v1.txt)
  t = al16(t);
  t += (64 - t % 256 + 256) % 256;
  a.tree_stride = (int32_t)t;
  int tpb = 16;
  while (tpb >= 4 && o + int64_t(tpb) * a.tree_stride > FUSED_LDS_BUDGET) tpb /= 2;
  if (tpb < 4) return P;
  a.f.trees_per_block = tpb;
  a.f.tree_stride = a.tree_stride;
  P.lds_bytes = (int)(o + int64_t(tpb) * a.tree_stride);
  P.ok = 1;
  return P;
}

template <class Net, int AW, bool PROFILE>
inline int fc2_launch(const Fc2Plan& P, unsigned grid, stream_t stream) {
  static std::atomic<uint64_t> lds_attr_done{0};   // per instantiation, one bit per device
  if (const int ae = allow_large_lds((const void*)fc2_search_kernel<Net, AW, PROFILE>, FUSED_LDS_BUDGET, lds_attr_done)) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  hipLaunchKernelGGL((fc2_search_kernel<Net, AW, PROFILE>), dim3(grid), dim3(P.args.f.trees_per_block * FUSED_ROW),
                     (size_t)P.lds_bytes, stream, P.args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("fused kernel launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

template <class Net, bool PROFILE>
inline int fc2_launch_aw(const Fc2Plan& P, unsigned grid, stream_t stream) {
  if (P.aw == 2) return fc2_launch<Net, 2, PROFILE>(P, grid, stream);
  if (P.aw == 4) return fc2_launch<Net, 4, PROFILE>(P, grid, stream);
  return fc2_launch<Net, 16, PROFILE>(P, grid, stream);
}

// mode bits: 1 = fused, 2 = export trees to the arena, 4 = force LdsNet, 8 = cycle-profile build
inline int fc2_run(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream, const RootOverride* ov = nullptr) {
  Fc2Plan P = fc2_plan(s, !(s->mode & 4));
  if (!P.ok) { set_error("fused search kernel does not support this configuration"); return MZX_ERR_INVALID; }
  int rc = ensure_tables(s, d_arena, stream);
  if (rc) return rc;
  if (ov) { P.args.ov_hidden = ov->hidden; P.args.ov_priors = ov->priors; P.args.ov_reward = ov->reward; }
  P.args.f.flat = s->net->d_flat;
  P.args.f.tables = s->d_tables;
  P.args.f.io = *io;
  const bool profile = (s->mode & 8) != 0 && s->ws_floats * 4 >= int64_t(s->p.num_trees) * FUSED_PROF_WORDS * 4;
  if (s->mode & 2) {
    P.args.f.export_trees = (char*)d_arena + s->off_trees;
    P.args.f.export_hidden = (float*)((char*)d_arena + s->off_hidden);
  }
  if (profile) P.args.f.prof = (uint32_t*)((char*)d_arena + s->off_ws);
  const int tpb = P.args.f.trees_per_block;
  const unsigned grid = (unsigned)((s->p.num_trees + tpb - 1) / tpb);
  if (P.small)   // SmallNetCartpole has A = 2
    return profile ? fc2_launch<SmallNetCartpole, 2, true>(P, grid, stream)
                   : fc2_launch<SmallNetCartpole, 2, false>(P, grid, stream);
  return profile ? fc2_launch_aw<LdsNet, true>(P, grid, stream) : fc2_launch_aw<LdsNet, false>(P, grid, stream);
}

#endif  // !MZX_HOSTCHECK

}  // namespace mzx
