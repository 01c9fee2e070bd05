// mzx_tree.h -- per-tree MCTS arithmetic (select / expand / backpropagate) as
// host+device inline functions over a struct-of-arrays tree.
//
// Replaces the per-node Python object graph of the reference:
//   MCTS.run            /root/reference/self_play.py:260-361
//   MCTS.select_child   self_play.py:363-378     MCTS.ucb_score   self_play.py:380-404
//   MCTS.backpropagate  self_play.py:406-430     Node             self_play.py:433-476
//   MinMaxStats         self_play.py:553-570
//
// Bit-exactness contract (DESIGN.md "Tree arithmetic"): every statistic is IEEE
// binary64, evaluated in the reference's operation order with contraction OFF
// (-ffp-contract=off); log/sqrt terms come from host-built tables so the device
// only executes +,-,*,/ which are correctly rounded on both sides.
//
// A node is addressed by its CANONICAL INDEX: root = 0, the leaf expanded by
// simulation k is k+1.  A child slot is (node, position in the node's action
// list).  Layout: the statistics a SELECTION needs about a child -- visit count,
// q = reward + discount * (+-value()), prior, link -- are stored in the PARENT's
// slot arrays, so one level of the walk reads A contiguous slots of one node
// (no pointer chase through child nodes); value_sum / visit / reward / to_play
// also live per node for back-propagation and for the root.
//
// The same functions are compiled (a) for gfx950 by hipcc and (b) for the host
// by g++ in tests/hostcheck (test-only build used by the CPU test-suite; the
// product never loads it).
#pragma once
#include <stdint.h>

#include "mzx_platform.h"

namespace mzx {

enum TreeMeta {
  TM_N_NODES = 0,
  TM_MAX_DEPTH = 1,
  TM_TAPE_POS = 2,
  TM_FLAGS = 3,
  TM_ROOT_N = 4,
  TM_CUR_PARENT = 5,
  TM_CUR_SLOT = 6,
  TM_CUR_DEPTH = 7,
  TM_CUR_TO_PLAY = 8,
  TM_TIE_DRAWS = 9,
  TM_SUM_DEPTH = 10,
  TM_CUR_LEAF = 11,
  TM_WORDS = 16
};

enum TreeFlags { TF_TAPE_OVERFLOW = 1, TF_NODE_OVERFLOW = 2 };

// Search-wide constants (one copy, passed by value to kernels).
struct SearchParams {
  int32_t num_trees;    // B
  int32_t num_sims;     // S
  int32_t num_actions;  // A = len(config.action_space)
  int32_t num_nodes;    // N = S + 1 node slots per tree
  int32_t num_players;  // len(config.players), 1 or 2
  int32_t hidden_size;  // Hf floats per node
  int32_t support_size;
  int32_t tape_words;   // raw MT19937 words available per tree for tie draws
  double discount;
  double exploration_fraction;
  const double* pbc_table;   // [N+1]: log((n + pb_c_base + 1) / pb_c_base) + pb_c_init, n = parent visits
  const double* sqrt_table;  // [N+1]: sqrt(n)
};

// Byte layout of one tree (all offsets multiples of 8).
struct TreeLayout {
  int32_t N, A;
  int64_t off_value_sum, off_reward, off_prior, off_sq, off_mm;                        // binary64
  int64_t off_visit, off_to_play, off_parent, off_parent_slot, off_svisit, off_child;   // int32
  int64_t off_root_actions, off_meta;
  int64_t tree_bytes;

  MZX_HD static inline int64_t align8(int64_t x) { return (x + 7) & ~int64_t(7); }

  MZX_HD static inline TreeLayout make(int32_t N, int32_t A) {
    TreeLayout L;
    L.N = N;
    L.A = A;
    int64_t o = 0;
    L.off_value_sum = o; o += int64_t(8) * N;
    L.off_reward = o;    o += int64_t(8) * N;
    L.off_prior = o;     o += int64_t(8) * N * A;
    L.off_sq = o;        o += int64_t(8) * N * A;
    L.off_mm = o;        o += 16;
    L.off_visit = o;     o += align8(int64_t(4) * N);
    L.off_to_play = o;   o += align8(int64_t(4) * N);
    L.off_parent = o;    o += align8(int64_t(4) * N);
    L.off_parent_slot = o; o += align8(int64_t(4) * N);
    L.off_svisit = o;    o += align8(int64_t(4) * N * A);
    L.off_child = o;     o += align8(int64_t(4) * N * A);
    L.off_root_actions = o; o += align8(int64_t(4) * A);
    L.off_meta = o;      o += int64_t(4) * TM_WORDS;
    L.tree_bytes = align8(o);
    return L;
  }
};

// View of ONE tree (global memory, LDS or host memory).
struct TreeRef {
  char* base;
  TreeLayout L;

  // per node
  MZX_HD inline double& value_sum(int n) const { return ((double*)(base + L.off_value_sum))[n]; }
  MZX_HD inline double& reward(int n) const { return ((double*)(base + L.off_reward))[n]; }
  MZX_HD inline int32_t& visit(int n) const { return ((int32_t*)(base + L.off_visit))[n]; }
  MZX_HD inline int32_t& to_play(int n) const { return ((int32_t*)(base + L.off_to_play))[n]; }
  MZX_HD inline int32_t& parent(int n) const { return ((int32_t*)(base + L.off_parent))[n]; }
  MZX_HD inline int32_t& parent_slot(int n) const { return ((int32_t*)(base + L.off_parent_slot))[n]; }
  // per child slot of a node
  MZX_HD inline double& prior(int n, int s) const { return ((double*)(base + L.off_prior))[n * L.A + s]; }
  MZX_HD inline double& slot_q(int n, int s) const { return ((double*)(base + L.off_sq))[n * L.A + s]; }
  MZX_HD inline int32_t& slot_visit(int n, int s) const { return ((int32_t*)(base + L.off_svisit))[n * L.A + s]; }
  MZX_HD inline int32_t& child(int n, int s) const { return ((int32_t*)(base + L.off_child))[n * L.A + s]; }
  // per tree
  MZX_HD inline double& mm_min() const { return ((double*)(base + L.off_mm))[0]; }
  MZX_HD inline double& mm_max() const { return ((double*)(base + L.off_mm))[1]; }
  MZX_HD inline int32_t& root_action(int s) const { return ((int32_t*)(base + L.off_root_actions))[s]; }
  MZX_HD inline int32_t& meta(int k) const { return ((int32_t*)(base + L.off_meta))[k]; }
};

// ---------------------------------------------------------------------------
// UCB score of one child slot -- self_play.py:380-404, same operation order.
// `pbc` = log((N+base+1)/base)+init and `sq` = sqrt(N) for the PARENT's visit
// count N come from the host tables; n / q / prior are the slot's visit count,
// cached reward + discount * (+-value()) (written by backpropagate with the very
// expression ucb_score uses, hence the same bits) and prior.
MZX_HD inline double ucb_from(double pbc, double sq, int n, double prior, double q, double mn, double mx) {
  // Straight-line form (selects, no branches: the lanes of a wave score different children): every
  // operation the reference performs is performed with the same operands in the same order; results of
  // the branches the reference does not take (e.g. the normalisation while max <= min, which may be
  // inf - inf) are computed and discarded -- IEEE arithmetic does not trap.
  const double pb_c = pbc * (sq / (double)(n + 1));            // pb_c *= sqrt(N) / (n + 1)
  const double prior_score = pb_c * prior;
  const double normalized = (q - mn) / (mx - mn);               // MinMaxStats.normalize, :566-570
  const double v = (mx > mn) ? normalized : q;
  const double with_value = prior_score + v;
  return (n > 0) ? with_value : prior_score;
}

template <class T>
MZX_HD inline double ucb_score(const T& t, int node, int slot, double pbc, double sq, double mn, double mx) {
  return ucb_from(pbc, sq, t.slot_visit(node, slot), t.prior(node, slot), t.slot_q(node, slot), mn, mx);
}

// Masked-rejection draw in [0, n) from a tape of raw MT19937 words: exactly
// numpy's legacy randint(0, n) / choice(list of n) (n >= 2; tests pin it).
MZX_HD inline int tape_draw(const uint32_t* tape, int tape_words, int32_t& pos, int32_t& flags, int n) {
  uint32_t rng = (uint32_t)(n - 1), mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  for (;;) {
    if (pos >= tape_words) { flags |= TF_TAPE_OVERFLOW; return 0; }
    uint32_t w = tape[pos++] & mask;
    if (w <= rng) return (int)w;
  }
}

// One selection walk (the `while node.expanded()` loop, self_play.py:325-334):
// descends from the root to the first unexpanded child slot, allocates the new
// leaf's canonical index and records (parent, slot, depth, virtual_to_play).
// Returns the parent node.
template <class T>
MZX_HD inline int tree_select(const T& t, const SearchParams& p, const uint32_t* tape) {
  int node = 0, depth = 0, slot = 0;
  int vtp = t.to_play(0);
  const double mn = t.mm_min(), mx = t.mm_max();
  int32_t tape_pos = t.meta(TM_TAPE_POS), flags = t.meta(TM_FLAGS), ties = t.meta(TM_TIE_DRAWS);
  for (;;) {
    ++depth;
    const int nc = (node == 0) ? t.meta(TM_ROOT_N) : p.num_actions;
    const int N = t.visit(node);
    const double pbc = p.pbc_table[N], sq = p.sqrt_table[N];
    double best = 0.0;
    int nbest = 0, first = 0;
    for (int s = 0; s < nc; ++s) {
      const double sc = ucb_score(t, node, s, pbc, sq, mn, mx);
      if (s == 0 || sc > best) { best = sc; nbest = 1; first = s; }
      else if (sc == best) { ++nbest; }
    }
    slot = first;
    if (nbest > 1) {  // numpy.random.choice(ties): k-th maximiser in slot order
      ++ties;
      int k = tape_draw(tape, p.tape_words, tape_pos, flags, nbest);
      for (int s = first; s < nc; ++s) {
        if (ucb_score(t, node, s, pbc, sq, mn, mx) == best) {
          if (k == 0) { slot = s; break; }
          --k;
        }
      }
    }
    vtp = (vtp + 1 < p.num_players) ? vtp + 1 : 0;  // players turn by turn, :331-334
    const int nxt = t.child(node, slot);
    if (nxt < 0) break;
    node = nxt;
  }
  int leaf = t.meta(TM_N_NODES);
  if (leaf >= p.num_nodes) { flags |= TF_NODE_OVERFLOW; leaf = p.num_nodes - 1; }
  t.meta(TM_TAPE_POS) = tape_pos;
  t.meta(TM_FLAGS) = flags;
  t.meta(TM_TIE_DRAWS) = ties;
  t.meta(TM_CUR_PARENT) = node;
  t.meta(TM_CUR_SLOT) = slot;
  t.meta(TM_CUR_DEPTH) = depth;
  t.meta(TM_CUR_TO_PLAY) = vtp;
  t.meta(TM_CUR_LEAF) = leaf;
  return node;
}

// What one selection walk hands to the expansion / back-propagation of the same
// simulation.  The generic path parks it in the tree's meta words between kernels;
// the fused kernels keep it in registers.
struct SelCtx { int32_t parent, slot, leaf, depth, to_play; };

template <class T>
MZX_HD inline SelCtx load_ctx(const T& t) {
  SelCtx c;
  c.parent = t.meta(TM_CUR_PARENT); c.slot = t.meta(TM_CUR_SLOT); c.leaf = t.meta(TM_CUR_LEAF);
  c.depth = t.meta(TM_CUR_DEPTH); c.to_play = t.meta(TM_CUR_TO_PLAY);
  return c;
}

// Node.expand, part 1 (self_play.py:451-458): link the selected leaf under its
// parent and initialise its node record.  Priors / child links of the leaf's own
// slots are written by tree_init_slot (one call per slot, any order / any lane).
template <class T>
MZX_HD inline void tree_attach_leaf(const T& t, const SearchParams& p, const SelCtx& c, double reward) {
  t.child(c.parent, c.slot) = c.leaf;
  t.parent(c.leaf) = c.parent;
  t.parent_slot(c.leaf) = c.slot;
  t.to_play(c.leaf) = c.to_play;
  t.reward(c.leaf) = reward;
  t.visit(c.leaf) = 0;
  t.value_sum(c.leaf) = 0.0;
  t.meta(TM_N_NODES) = c.leaf + 1;
  if (c.depth > t.meta(TM_MAX_DEPTH)) t.meta(TM_MAX_DEPTH) = c.depth;
  t.meta(TM_SUM_DEPTH) += c.depth;
}

// Node.expand, part 2 (self_play.py:460-465): child slot `s` of `node` starts
// unvisited with the given prior (fp32 softmax probability widened to binary64).
template <class T>
MZX_HD inline void tree_init_slot(const T& t, int node, int s, double prior) {
  t.prior(node, s) = prior;
  t.slot_q(node, s) = 0.0;
  t.slot_visit(node, s) = 0;
  t.child(node, s) = -1;
}

// Statistics update of ONE path node given the value arriving from below
// (self_play.py:411-417 / :420-427); returns the value to pass upward.  Mirrors
// (visit, q) into the parent's slot arrays for the next selection.
template <class T>
MZX_HD inline double tree_update_node(const T& t, const SearchParams& p, int n, double value, int leaf_to_play,
                                      double& mn, double& mx) {
  const double disc = p.discount;
  const double r = t.reward(n);
  double vs, qv, up;
  const int vc = t.visit(n) + 1;
  if (p.num_players == 1) {
    vs = t.value_sum(n) + value;
    qv = r + disc * (vs / (double)vc);
    up = r + disc * value;
  } else {
    const bool same = (t.to_play(n) == leaf_to_play);
    vs = t.value_sum(n) + (same ? value : -value);
    qv = r + disc * (-(vs / (double)vc));
    up = (same ? -r : r) + disc * value;
  }
  t.value_sum(n) = vs;
  t.visit(n) = vc;
  const int par = t.parent(n);
  if (par >= 0) {
    const int ps = t.parent_slot(n);
    t.slot_visit(par, ps) = vc;
    t.slot_q(par, ps) = qv;
  }
  if (qv > mx) mx = qv;   // MinMaxStats.update, :562-564
  if (qv < mn) mn = qv;
  return up;
}

// MCTS.backpropagate (self_play.py:406-430): walk parent links leaf -> root.
template <class T>
MZX_HD inline void tree_backprop(const T& t, const SearchParams& p, const SelCtx& c, double value) {
  double mn = t.mm_min(), mx = t.mm_max();
  for (int n = c.leaf; n >= 0; n = t.parent(n)) value = tree_update_node(t, p, n, value, c.to_play, mn, mx);
  t.mm_min() = mn;
  t.mm_max() = mx;
}

// expand + backpropagate of one simulation (self_play.py:345-353).
template <class T, class PriorFn>
MZX_HD inline void tree_expand_backprop(const T& t, const SearchParams& p, double value, double reward,
                                        PriorFn prior_of_slot) {
  const SelCtx c = load_ctx(t);
  tree_attach_leaf(t, p, c, reward);
  for (int s = 0; s < p.num_actions; ++s) tree_init_slot(t, c.leaf, s, prior_of_slot(s));
  tree_backprop(t, p, c, value);
}

// Root creation: Node(0) + root.expand(legal_actions, ...) + add_exploration_noise
// (self_play.py:276-314, :467-476).  `legal` is the game's legal_actions() list
// in ITS order (slot i = legal[i]; the reference's children dict keeps that
// order), padded with -1 up to num_actions.  `noise` (slot order) may be null.
template <class T>
MZX_HD inline int tree_init_root_record(const T& t, const SearchParams& p, const int32_t* legal, int to_play,
                                        double root_reward) {
  for (int k = 0; k < TM_WORDS; ++k) t.meta(k) = 0;
  t.mm_min() = MZX_INF;
  t.mm_max() = -MZX_INF;
  int nroot = 0;
  while (nroot < p.num_actions && legal[nroot] >= 0) { t.root_action(nroot) = legal[nroot]; ++nroot; }
  for (int s = nroot; s < p.num_actions; ++s) t.root_action(s) = -1;
  t.meta(TM_ROOT_N) = nroot;
  t.meta(TM_N_NODES) = 1;
  t.visit(0) = 0;
  t.value_sum(0) = 0.0;
  t.reward(0) = root_reward;
  t.to_play(0) = to_play;
  t.parent(0) = -1;
  t.parent_slot(0) = -1;
  return nroot;
}

// prior of root slot s after exploration noise (self_play.py:476); `pr` = softmax prior.
MZX_HD inline double root_noisy_prior(double pr, const double* noise, int s, double frac) {
  return noise ? pr * (1.0 - frac) + noise[s] * frac : pr;
}

template <class T, class PriorFn>
MZX_HD inline void tree_init_root(const T& t, const SearchParams& p, const int32_t* legal, int to_play,
                                  double root_reward, PriorFn prior_of_slot, const double* noise) {
  const int nroot = tree_init_root_record(t, p, legal, to_play, root_reward);
  for (int s = 0; s < p.num_actions; ++s)
    tree_init_slot(t, 0, s, s < nroot ? root_noisy_prior(prior_of_slot(s), noise, s, p.exploration_fraction) : 0.0);
}

// ---------------------------------------------------------------------------
// fp32 helpers that turn network heads into tree inputs.
//
// Reductions use ONE canonical order everywhere (generic per-thread operators
// and the 16-lanes-per-tree fused kernels): element i belongs to lane i % 16,
// a lane accumulates its elements in increasing i, and the 16 lane partials are
// combined by the butterfly  xor 1, xor 2, half-mirror (i <-> 7-i), mirror
// (i <-> 15-i) -- what the fused kernels do with DPP.  Hence both paths produce
// bit-identical decoded values / priors on the device.

// Lane 0's result of the 16-lane butterfly sum of v[0..15].
MZX_HD inline float butterfly16_sum(const float* v) {
  const float q0 = (v[0] + v[1]) + (v[2] + v[3]);
  const float q1 = (v[7] + v[6]) + (v[5] + v[4]);
  const float q3 = (v[15] + v[14]) + (v[13] + v[12]);
  const float q2 = (v[8] + v[9]) + (v[10] + v[11]);
  return (q0 + q1) + (q3 + q2);
}

// Inverse of the value scaling h(x) (models.py:660-665), fp32, same op order as the
// reference's tensor expression.  NOTE x = 0 gives sign(0) * (tiny negative) = -0.0f:
// the reference's root reward is NEGATIVE zero (log one-hot head, models.py:176-183),
// and the root is initialised with exactly this value to keep zero signs identical.
MZX_HD inline float support_inverse_transform(float x) {
  float t = fabsf(x) + 1.0f;
  t = t + 0.001f;
  t = 0.004f * t;
  t = 1.0f + t;
  t = sqrtf(t);
  t = (t - 1.0f) / 0.002f;
  t = t * t - 1.0f;
  const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  return sgn * t;
}

// fp32 softmax statistics over n logits (Node.expand, self_play.py:460-462).
struct SoftmaxStats { float m, den; };
template <class LogitFn>
MZX_HD inline SoftmaxStats softmax_stats(int n, LogitFn logit) {
  SoftmaxStats s;
  s.m = logit(0);
  for (int i = 1; i < n; ++i) s.m = fmaxf(s.m, logit(i));
  float lane[16];
  for (int j = 0; j < 16; ++j) {
    float acc = 0.f;
    for (int i = j; i < n; i += 16) acc += mzx_expf(logit(i) - s.m);
    lane[j] = acc;
  }
  s.den = butterfly16_sum(lane);
  return s;
}

// models.support_to_scalar (models.py:645-666) for one row of 2*support+1 logits.
MZX_HD inline float support_to_scalar(const float* logits, int support_size) {
  const int F = 2 * support_size + 1;
  const SoftmaxStats st = softmax_stats(F, [&](int i) { return logits[i]; });
  float lane[16];
  for (int j = 0; j < 16; ++j) {
    float acc = 0.f;
    for (int i = j; i < F; i += 16) acc += (float)(i - support_size) * mzx_div(mzx_expf(logits[i] - st.m), st.den);
    lane[j] = acc;
  }
  return support_inverse_transform(butterfly16_sum(lane));
}

}  // namespace mzx
