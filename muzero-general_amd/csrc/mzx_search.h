// mzx_search.h -- host-side driver of the batched search (generic path):
// carves the caller's arena, sequences the per-simulation operators and exposes
// the lock-step interface used by the parity harness.
//
// Reference: MCTS.run, /root/reference/self_play.py:260-361 -- the body of one
// simulation (:319-355) becomes select -> gather hidden -> recurrent_inference
// -> expand+backpropagate -> scatter hidden, each over all B trees at once.
#pragma once
#include <memory>
#include <vector>

#include "mzx_resnet_fused.h"

struct mzx_search {
  mzx_search_config cfg;
  mzx_net* net = nullptr;
  mzx::SearchParams p;
  mzx::TreeLayout L;
  std::vector<double> h_pbc, h_sqrt;
  // arena carve (byte offsets)
  int64_t off_tables = 0, off_trees = 0, off_hidden = 0, off_dense_in = 0, off_dense_out = 0;
  int64_t off_value = 0, off_reward = 0, off_policy = 0, off_sel = 0, off_rowsel = 0, off_rowpath = 0, off_ws = 0, arena_bytes = 0;
  int64_t ws_floats = 0;
  // pbc[N+1] then sqrt[N+1] on the device: owned by the handle (uploaded once at create), NOT carved from the
  // caller's arena -- an arena may be cleared, freed or re-allocated at the same address between calls
  double* d_tables = nullptr;
  int32_t device = -1;     // device the tables were allocated on (= the device every launch of this handle must run on)
  int32_t mode = 0;
  int32_t fused_ok = 0;
  const char* last_kernel = "";   // search kernel of the last mzx_search_run (mzx_search_kernel_name)
  // second HIP stream + fork / join events of the row-per-tree path (mzx_row_search.h: two half-shards in flight
  // together); created on first use, owned by the handle
  void* side_stream = nullptr;
  void* ev_fork = nullptr;
  void* ev_join = nullptr;
  int32_t split_first = 0;     // trees of the first half-shard of the last run (rb_split_first), 0 = undivided
  // launch plan of the tower whole-search kernel (mzx_tower_search.inc: RtPlan), computed once per (shard size, tuning)
  // and reused by every move's search; owned by the handle
  mutable std::shared_ptr<void> rt_plan_cache;
  mutable int64_t rt_plan_key[4] = {-1, -1, -1, -1};
};

namespace mzx {

inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

inline void search_plan(mzx_search* s) {
  const mzx_search_config& c = s->cfg;
  const int B = c.num_trees, N = c.num_simulations + 1, A = c.action_space_size;
  const int F = 2 * c.support_size + 1;
  const int64_t Hf = s->net ? s->net->hidden_size : 0;
  s->L = TreeLayout::make(N, A);
  SearchParams& p = s->p;
  p.num_trees = B; p.num_sims = c.num_simulations; p.num_actions = A; p.num_nodes = N;
  p.num_players = c.num_players; p.hidden_size = (int32_t)Hf; p.support_size = c.support_size;
  p.tape_words = c.tape_words; p.discount = c.discount; p.exploration_fraction = c.root_exploration_fraction;
  p.pbc_table = nullptr; p.sqrt_table = nullptr;
  int64_t o = 0;
  s->off_tables = o;    // (the tables live in a handle-owned buffer; no arena bytes)
  s->off_trees = o;     o += align256(s->L.tree_bytes * B);
  s->off_hidden = o;    o += align256(int64_t(4) * B * N * Hf);
  s->off_dense_in = o;  o += align256(int64_t(4) * B * Hf);
  s->off_dense_out = o; o += align256(int64_t(4) * B * Hf);
  s->off_value = o;     o += align256(int64_t(4) * B * F);
  s->off_reward = o;    o += align256(int64_t(4) * B * F);
  s->off_policy = o;    o += align256(int64_t(4) * B * A);
  s->off_sel = o;       o += align256(int64_t(4) * B * 3);
  s->off_rowsel = o;    o += align256(int64_t(4) * B * 64);   // path hand-off of the row kernels (mzx_row_search.h)
  s->off_rowpath = o;   o += align256(int64_t(8) * B * (N + 1));   // ... and the whole path of a walk, (node, slot taken) per depth: paths beyond a row's 16 lanes
  s->ws_floats = s->net ? net_ws_per_sample(s->net) * (int64_t)B : 0;
  s->off_ws = o;        o += align256(int64_t(4) * s->ws_floats);
  s->arena_bytes = o;
}

struct ArenaView {
  TreeArena arena;
  SearchParams p;
  float *dense_in, *dense_out, *value, *reward, *policy, *ws;
  int32_t *sel_parent, *sel_action, *sel_leaf;
};

inline ArenaView arena_view(const mzx_search* s, void* d_arena) {
  char* base = (char*)d_arena;
  ArenaView v;
  v.arena.trees = base + s->off_trees;
  v.arena.hidden = (float*)(base + s->off_hidden);
  v.arena.L = s->L;
  v.p = s->p;
  v.p.pbc_table = s->d_tables;
  v.p.sqrt_table = v.p.pbc_table + (s->p.num_nodes + 1);
  v.dense_in = (float*)(base + s->off_dense_in);
  v.dense_out = (float*)(base + s->off_dense_out);
  v.value = (float*)(base + s->off_value);
  v.reward = (float*)(base + s->off_reward);
  v.policy = (float*)(base + s->off_policy);
  v.sel_parent = (int32_t*)(base + s->off_sel);
  v.sel_action = v.sel_parent + s->p.num_trees;
  v.sel_leaf = v.sel_action + s->p.num_trees;
  v.ws = (float*)(base + s->off_ws);
  return v;
}

#define MZX_TRY_LAUNCH(expr)                                                        \
  do {                                                                              \
    int _rc = (expr);                                                               \
    if (_rc) { set_error("launch failed: %s", runtime_error_string(_rc)); return MZX_ERR_RUNTIME; } \
  } while (0)

// The handle's own device copy of the host tables (mzx_search_create); freed by mzx_search_destroy.
inline int upload_tables(mzx_search* s) {
  const int n = s->p.num_nodes + 1;
  void* d = nullptr;
  s->device = current_device();
  MZX_TRY_LAUNCH(device_alloc(&d, sizeof(double) * 2 * n));
  s->d_tables = (double*)d;
  MZX_TRY_LAUNCH(copy_h2d_blocking(s->d_tables, s->h_pbc.data(), sizeof(double) * n));
  MZX_TRY_LAUNCH(copy_h2d_blocking(s->d_tables + n, s->h_sqrt.data(), sizeof(double) * n));
  return MZX_OK;
}
inline int ensure_tables(mzx_search* s, void*, stream_t) {
  if (!s->d_tables) { set_error("search handle has no device tables"); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

inline int search_finish(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream) {
  const ArenaView v = arena_view(s, d_arena);
  FinalizeOp f;
  f.arena = v.arena; f.p = v.p;
  f.visit_counts = io->d_visit_counts; f.root_value = io->d_root_value; f.info = io->d_info;
  MZX_TRY_LAUNCH(launch<64>(f, stream));
  return MZX_OK;
}

// Roots the caller expanded itself (MCTS.run(..., override_root_with=root), self_play.py:275-277):
// prior per legal slot (binary64), reward and hidden state of each root replace initial_inference.
struct RootOverride {
  const float* hidden;    // [B][Hf]
  const double* priors;   // [B][A] slot order
  const double* reward;   // [B]
};

// Generic path: one kernel per operator (any network configuration).
inline int search_run_generic(mzx_search* s, const mzx_search_io* io, void* d_arena, stream_t stream,
                              const RootOverride* ov = nullptr) {
  const ArenaView v = arena_view(s, d_arena);
  mzx_net* net = s->net;
  const int B = s->p.num_trees;
  int rc = ensure_tables(s, d_arena, stream);
  if (rc) return rc;

  // With the fused network engine the per-node hidden states are read from / written to the arena
  // store [B][N][Hf] directly (NetIndex); the per-operator engine goes through dense staging copies.
  const bool ix_init = rz_enabled(net, false) || rb_enabled(net, false), ix_rec = rz_enabled(net, true) || rb_enabled(net, true);
  NetIndex ix;
  ix.in_nodes = 1; ix.out_nodes = s->p.num_nodes;   // root: dense observation in, node 0 out

  NetBuffers nb;
  nb.in = io->d_observation; nb.action = nullptr; nb.hidden = ix_init ? v.arena.hidden : v.dense_out;
  nb.value = v.value; nb.reward = v.reward; nb.policy = v.policy; nb.workspace = v.ws;
  if (!ov) {
    rc = run_network(net, false, nb, B, stream, ix_init ? &ix : nullptr);
    if (rc) return rc;
  }

  RootInitOp ri;
  ri.arena = v.arena; ri.p = v.p; ri.value_logits = v.value; ri.policy_logits = v.policy;
  ri.ext_priors = ov ? ov->priors : nullptr; ri.ext_root_reward = ov ? ov->reward : nullptr;
  ri.legal = io->d_legal_actions; ri.to_play = io->d_to_play; ri.noise = io->d_noise;
  ri.root_predicted_value = io->d_root_predicted_value;
  MZX_TRY_LAUNCH(launch<64>(ri, stream));

  HiddenMoveOp mv;
  mv.arena = v.arena; mv.num_trees = B; mv.num_nodes = s->p.num_nodes; mv.hidden_size = s->p.hidden_size;
  if (ov || !ix_init) {
    mv.dense = ov ? const_cast<float*>(ov->hidden) : v.dense_out; mv.node = nullptr; mv.to_arena = 1;
    MZX_TRY_LAUNCH(launch<256>(mv, stream));
  }

  SelectOp sel;
  sel.arena = v.arena; sel.p = v.p; sel.tape = io->d_tape;
  sel.sel_parent = v.sel_parent; sel.sel_action = v.sel_action; sel.sel_leaf = v.sel_leaf;
  ExpandBackpropOp eb;
  eb.arena = v.arena; eb.p = v.p; eb.value_logits = v.value; eb.reward_logits = v.reward; eb.policy_logits = v.policy;
  eb.ext_value = nullptr; eb.ext_reward = nullptr; eb.ext_priors = nullptr;
  nb.action = v.sel_action;
  if (ix_rec) {
    nb.in = v.arena.hidden; nb.hidden = v.arena.hidden;
    ix.in_node = v.sel_parent; ix.out_node = v.sel_leaf; ix.in_nodes = s->p.num_nodes; ix.out_nodes = s->p.num_nodes;
  } else {
    nb.in = v.dense_in; nb.hidden = v.dense_out;
  }

  for (int k = 0; k < s->p.num_sims; ++k) {
    MZX_TRY_LAUNCH(launch<64>(sel, stream));
    if (ix_rec) {
      rc = run_network(net, true, nb, B, stream, &ix);
      if (rc) return rc;
      MZX_TRY_LAUNCH(launch<64>(eb, stream));
      continue;
    }
    mv.dense = v.dense_in; mv.node = v.sel_parent; mv.to_arena = 0;
    MZX_TRY_LAUNCH(launch<256>(mv, stream));
    rc = run_network(net, true, nb, B, stream);
    if (rc) return rc;
    MZX_TRY_LAUNCH(launch<64>(eb, stream));
    mv.dense = v.dense_out; mv.node = v.sel_leaf; mv.to_arena = 1;
    MZX_TRY_LAUNCH(launch<256>(mv, stream));
  }
  return search_finish(s, io, d_arena, stream);
}

// Copies the trees out in canonical node order (parity tests, diagnose tooling).
struct DumpOp {
  TreeArena arena;
  SearchParams p;
  mzx_tree_dump d;
  MZX_HD size_t size() const { return (size_t)p.num_trees; }
  MZX_HD void operator()(size_t i) const {
    const int b = (int)i, N = p.num_nodes, A = p.num_actions;
    const TreeRef t = arena.tree(b);
    const int nn = t.meta(TM_N_NODES);
    if (d.d_n_nodes) d.d_n_nodes[b] = nn;
    if (d.d_minmax) { d.d_minmax[b * 2] = t.mm_min(); d.d_minmax[b * 2 + 1] = t.mm_max(); }
    for (int n = 0; n < N; ++n) {
      const bool live = n < nn;
      const int64_t o = (int64_t)b * N + n;
      if (d.d_visit) d.d_visit[o] = live ? t.visit(n) : 0;
      if (d.d_value_sum) d.d_value_sum[o] = live ? t.value_sum(n) : 0.0;
      if (d.d_reward) d.d_reward[o] = live ? t.reward(n) : 0.0;
      if (d.d_to_play) d.d_to_play[o] = live ? t.to_play(n) : -1;
      if (d.d_parent) d.d_parent[o] = live ? t.parent(n) : -1;
      for (int a = 0; a < A; ++a) {
        if (d.d_child) d.d_child[o * A + a] = live ? t.child(n, a) : -1;
        if (d.d_prior) d.d_prior[o * A + a] = live ? t.prior(n, a) : 0.0;
      }
    }
  }
};

}  // namespace mzx
