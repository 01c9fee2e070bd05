// mzx_lib.cpp -- the C ABI (include/mzx.h).  Compiled as HIP for gfx950 into
// libmzx.so (product) and, with -DMZX_HOSTCHECK, as plain C++ into the test-only
// libmzx_hostcheck.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>

#include "mzx_obs.h"
#include "mzx_replay.h"
#include "mzx_rng.h"
#include "mzx_search.h"
#ifndef MZX_HOSTCHECK
#include "mzx_fused_fc.h"
#include "mzx_fused_fc2.h"
#include "mzx_resnet_search.h"
#include "mzx_row_search.h"
// rt_search_kernel and its planner: part of THIS translation unit (round 6).  As a unit of its own it instantiated every
// header-defined kernel a second time (648 kernels in both code objects: the library was 9.3 MB, half of it duplicates).
#include "mzx_tower_search.inc"
#endif

namespace mzx {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}

}  // namespace mzx

using namespace mzx;

extern "C" {

int mzx_abi_version(void) { return MZX_ABI_VERSION; }

const char* mzx_last_error(void) { return g_error.c_str(); }

int mzx_is_device_build(void) {
#ifdef MZX_HOSTCHECK
  return 0;
#else
  return 1;
#endif
}

// ---------------------------------------------------------------- network

int mzx_net_create(const mzx_net_config* cfg, mzx_net** out) {
  if (!cfg || !out) { set_error("mzx_net_create: null argument"); return MZX_ERR_INVALID; }
  const int32_t counts[] = {cfg->n_fc_representation_layers, cfg->n_fc_dynamics_layers, cfg->n_fc_reward_layers,
                            cfg->n_fc_value_layers, cfg->n_fc_policy_layers, cfg->n_resnet_fc_reward_layers,
                            cfg->n_resnet_fc_value_layers, cfg->n_resnet_fc_policy_layers};
  for (int32_t n : counts)
    if (n < 0 || n > MZX_MAX_LAYERS) { set_error("more than %d hidden layers in an MLP", MZX_MAX_LAYERS); return MZX_ERR_INVALID; }
  mzx_net* net = new (std::nothrow) mzx_net();
  if (!net) { set_error("out of host memory"); return MZX_ERR_RUNTIME; }
  net->cfg = *cfg;
  NetBuilder b(net);
  if (!b.build()) { delete net; return MZX_ERR_INVALID; }
  rz_plan(net);
  rb_plan(net);
  *out = net;
  return MZX_OK;
}

void mzx_net_destroy(mzx_net* net) { delete net; }

int32_t mzx_net_num_tensors(const mzx_net* net) { return net ? (int32_t)net->tensors.size() : 0; }
int64_t mzx_net_num_params(const mzx_net* net) { return net ? net->num_params : 0; }
int64_t mzx_net_hidden_size(const mzx_net* net) { return net ? net->hidden_size : 0; }
int64_t mzx_net_input_size(const mzx_net* net) { return net ? net->input_size : 0; }
static int64_t derived_total(const mzx_net* net) {
  int64_t n = net->rz.ok ? net->rz.derived_floats : net->derived_floats;
  if (net->rb.ok && net->rb.derived_floats > n) n = net->rb.derived_floats;
  return n > 0 ? n : 1;
}
int64_t mzx_net_derived_floats(const mzx_net* net) { return net ? derived_total(net) : 0; }
int64_t mzx_net_workspace_floats(const mzx_net* net, int32_t max_batch) {
  if (!net || max_batch < 1) return 0;
  return net_ws_per_sample(net) * (int64_t)max_batch;
}

int mzx_net_tensor_info(const mzx_net* net, int32_t i, char* name, int32_t name_cap, int64_t* offset,
                        int64_t* numel, int32_t dims[4]) {
  if (!net || i < 0 || i >= (int32_t)net->tensors.size()) { set_error("tensor index out of range"); return MZX_ERR_INVALID; }
  const TensorInfo& t = net->tensors[i];
  if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = t.offset;
  if (numel) *numel = t.numel;
  if (dims) for (int k = 0; k < 4; ++k) dims[k] = t.dims[k];
  return MZX_OK;
}

int mzx_net_set_weights(mzx_net* net, const float* d_flat, int64_t n_floats, float* d_derived,
                        int64_t derived_floats, void* stream) {
  if (!net || !d_flat || !d_derived) { set_error("mzx_net_set_weights: null argument"); return MZX_ERR_INVALID; }
  if (n_floats != net->num_params) {
    set_error("flat weight buffer has %lld floats, network expects %lld", (long long)n_floats, (long long)net->num_params);
    return MZX_ERR_INVALID;
  }
  if (derived_floats < derived_total(net)) { set_error("derived buffer too small"); return MZX_ERR_WORKSPACE; }
  net->d_flat = d_flat;
  net->d_derived = d_derived;
  for (const BnRef& r : net->bns) {
    BnFoldOp op;
    op.weight = d_flat + r.weight; op.bias = d_flat + r.bias; op.mean = d_flat + r.mean; op.var = d_flat + r.var;
    op.alpha = d_derived + r.alpha; op.beta = d_derived + r.beta; op.channels = r.channels;
    MZX_TRY_LAUNCH(launch<256>(op, (stream_t)stream));
  }
  if (net->rz.ok) {  // MFMA-fragment-ordered weights + program tables of the fused residual engine
    for (const RzPack& p : net->rz.packs) {
      RzPackOp op;
      op.W = d_flat + p.src; op.out = d_derived + p.dst;
      op.taps = p.taps; op.cin = p.cin; op.cin_total = p.cin_total; op.cchunks = p.cchunks; op.cout = p.cout;
      op.nchunks = p.nchunks; op.wchunks = p.wchunks; op.ntiles = p.ntiles;
      MZX_TRY_LAUNCH(launch<256>(op, (stream_t)stream));
    }
    for (const RzAsum& q : net->rz.asums) {
      RzAsumOp op;
      op.W = d_flat + q.src; op.out = d_derived + q.dst; op.cout = q.cout; op.cin_total = q.cin_total; op.H = q.H; op.Wd = q.W;
      MZX_TRY_LAUNCH(launch<256>(op, (stream_t)stream));
    }
    for (const RzCopy& c : net->rz.copies) {
      RzCopyOp op;
      op.src = (c.from_derived ? (const float*)d_derived : d_flat) + c.src; op.dst = d_derived + c.dst; op.n = c.n; op.npad = c.npad;
      MZX_TRY_LAUNCH(launch<256>(op, (stream_t)stream));
    }
    for (const RzProgram* R : {&net->rz.initial, &net->rz.recurrent})
      if (R->ok) {
        MZX_TRY_LAUNCH(copy_h2d(d_derived + R->small_base, R->ops, sizeof(RzOp) * R->n_ops, (stream_t)stream));
        MZX_TRY_LAUNCH(copy_h2d(d_derived + R->small_base + R->aoff_base, R->aoff.data(), sizeof(int32_t) * R->aoff.size(),
                                (stream_t)stream));
      }
  }
#ifndef MZX_HOSTCHECK
  if (net->rb.ok) {  // streamed MFMA engine: fragment-ordered weights, action tap sums
    const int rc = rb_refresh_derived(net, d_flat, d_derived, (stream_t)stream);
    if (rc) return rc;
  }
#endif
  return MZX_OK;
}

int mzx_net_fused_supported(const mzx_net* net) {
  if (!net || !net->rz.ok) return 0;
  return (net->rz.initial.ok ? 1 : 0) | (net->rz.recurrent.ok ? 2 : 0);
}

int mzx_net_streamed_supported(const mzx_net* net) {
  if (!net || !net->rb.ok) return 0;
  const bool fi = net->rz.ok && net->rz.initial.ok, fr = net->rz.ok && net->rz.recurrent.ok;
  return ((net->rb.initial.ok && !fi) ? 1 : 0) | ((net->rb.recurrent.ok && !fr) ? 2 : 0);
}

int mzx_net_streamed_plan(const mzx_net* net, int32_t recurrent, int32_t op, int32_t out[24]) {
  if (!net || !out) { set_error("null argument"); return MZX_ERR_INVALID; }
  const RbProgram& R = recurrent ? net->rb.recurrent : net->rb.initial;
  if (!net->rb.ok || !R.ok || op < 0 || op >= (int32_t)R.ops.size()) { set_error("no streamed plan for this operator"); return MZX_ERR_INVALID; }
  const RbOp& o = R.ops[op];
  const int32_t v[24] = {o.kind, o.in_layout, o.out_layout, o.res_layout, o.taps, o.stride, o.cin, o.cout, o.hin, o.win,
                         o.hout, o.wout, o.T, o.th, o.tw, o.tiles_x, o.tiles_y, o.PH, o.PW, o.cpg, o.phases, o.rows,
                         o.mtiles, o.lds_bytes};
  for (int k = 0; k < 24; ++k) out[k] = v[k];
  return MZX_OK;
}

int mzx_net_streamed_shape(const mzx_net* net, int32_t recurrent, int32_t op, int32_t batch, int32_t out[16]) {
  if (!net || !out || batch < 1) { set_error("null argument / batch < 1"); return MZX_ERR_INVALID; }
  const RbProgram& R = recurrent ? net->rb.recurrent : net->rb.initial;
  if (!net->rb.ok || !R.ok || op < 0 || op >= (int32_t)R.ops.size() || R.ops[op].kind != RB_GEMM) {
    set_error("operator %d has no streamed GEMM plan", op);
    return MZX_ERR_INVALID;
  }
  const RbShape sh = rb_choose_shape(R.ops[op], batch);
  const int32_t v[16] = {sh.T, sh.rows, sh.mtiles, sh.lds, sh.ntiles_wg, sh.nsplit, sh.NT, sh.WN, sh.WM, sh.MT, sh.groups,
                         sh.cpg, sh.phases, sh.Cs, R.ops[op].ntiles, R.ops[op].cchunks};
  for (int k = 0; k < 16; ++k) out[k] = v[k];
  return MZX_OK;
}

int mzx_net_streamed_tower(const mzx_net* net, int32_t recurrent, int32_t index, int32_t batch, int32_t out[16]) {
  if (!net || !out || batch < 1) { set_error("null argument / batch < 1"); return MZX_ERR_INVALID; }
  const RbProgram& R = recurrent ? net->rb.recurrent : net->rb.initial;
  if (!net->rb.ok || !R.ok || net->rb_no_towers || index < 0 || index >= (int32_t)R.towers.size()) {
    set_error("no tower %d", index);
    return MZX_ERR_INVALID;
  }
  const RbTower& tw = R.towers[index];
  const RbTowerShape sh = rb_tower_shape(tw, batch);
  // (groups = 0: at this batch the tower's layers launch one by one, rb_tower_use)
  const int32_t v[16] = {tw.first, tw.count, tw.C, tw.H, tw.W, sh.T, sh.MT, sh.NT, sh.WM, sh.WN, sh.lds,
                         rb_tower_use(tw, batch) ? sh.groups : 0, tune(TUNE_RB_TAIL) == 0 ? 0 : tw.n_tail, 0, 0, 0};
  for (int k = 0; k < 16; ++k) out[k] = v[k];
  return MZX_OK;
}

int mzx_net_streamed_heads(const mzx_net* net, int32_t recurrent, int32_t batch, int32_t out[16]) {
  if (!net || !out || batch < 1) { set_error("null argument / batch < 1"); return MZX_ERR_INVALID; }
  for (int k = 0; k < 16; ++k) out[k] = 0;
  const RbProgram& R = recurrent ? net->rb.recurrent : net->rb.initial;
  if (!net->rb.ok || !R.ok) return MZX_OK;
  const int mode = rb_heads_mode();
  if (net->rb_no_towers || tune(TUNE_RB_TAIL) == 0 || mode != 2) return MZX_OK;
  // out: [0] Linear operators that leave the layer-by-layer path, [1] chains, [2 .. 13] the operators, [14] their levels
  // inside their chains (2 bits each), [15] the mode (rb_heads_mode)
  int n = 0;
  for (int q = 0; q < R.heads.n_chains; ++q) {
    const RbHeadChain& hc = R.heads.chain[q];
    const int t = R.ops[hc.conv_op].tower_of_tail;
    if (t < 0 || !rb_tower_use(R.towers[t], batch)) continue;
    ++out[1];
    for (int l = 0; l < hc.count && n < 12; ++l) { out[14] |= l << (2 * n); out[2 + n++] = hc.first + l; }
  }
  out[0] = n;
  out[15] = n ? mode : 0;
  return MZX_OK;
}

int mzx_net_streamed_split(const mzx_net* net, int32_t batch, int32_t out[2]) {
  if (!net || !out || batch < 1) { set_error("null argument / batch < 1"); return MZX_ERR_INVALID; }
  const int first = rb_split_first(net, batch, tune(TUNE_ROW_SPLIT_MIN));
  out[0] = first > 0 ? first : batch;
  out[1] = first > 0 ? batch - first : 0;
  return MZX_OK;
}

int64_t mzx_net_operator_out_floats(const mzx_net* net, int32_t recurrent, int32_t op) {
  if (!net) return 0;
  const std::vector<OpDesc>& prog = recurrent ? net->prog_recurrent : net->prog_initial;
  if (op < 0 || op >= (int32_t)prog.size()) return 0;
  const OpDesc& d = prog[op];
  switch (d.kind) {      // (the sizes run_network_prefix copies out)
    case OP_LINEAR: return d.out_features;
    case OP_CONV3: case OP_POOL: case OP_CONVK: case OP_MAXPOOL: case OP_ADAPTIVE_POOL: return (int64_t)d.cout * d.hout * d.wout;
    case OP_CONV1: return (int64_t)d.cout * d.hin;
    default: return (int64_t)d.groups_per_sample * d.len;
  }
}

int mzx_net_set_mode(mzx_net* net, int32_t mode) {
  if (!net) { set_error("null network handle"); return MZX_ERR_INVALID; }
  if (mode < 0 || mode > 5) { set_error("network mode is 0 (one kernel per operator), 1 (fused engine, streamed engine for what it cannot hold), 2 (fused, 4-wave workgroups), 3 (streamed engine for everything), 4 (as 3, layer by layer: no tower launches) or 5 (as 1, the streamed engine layer by layer)"); return MZX_ERR_INVALID; }
  if ((mode == 3 || mode == 4) && !net->rb.ok) { set_error("the streamed engine runs residual networks only"); return MZX_ERR_INVALID; }
  net->rz_mode = mode ? 1 : 0;
  net->rz_waves = (mode == 2) ? 4 : 0;
  net->rb_force = (mode == 3 || mode == 4) ? 1 : 0;
  net->rb_no_towers = (mode == 4 || mode == 5) ? 1 : 0;
  return MZX_OK;
}

int64_t mzx_net_flops(const mzx_net* net, int32_t recurrent) {
  if (!net) return 0;
  int64_t macs = 0;
  for (const OpDesc& d : (recurrent ? net->prog_recurrent : net->prog_initial)) {
    switch (d.kind) {
      case OP_CONV3: macs += (int64_t)d.cout * d.hout * d.wout * d.cin * 9; break;
      case OP_CONVK: macs += (int64_t)d.cout * d.hout * d.wout * d.cin * d.ksize * d.ksize; break;
      case OP_CONV1: macs += (int64_t)d.cout * d.hin * d.cin; break;
      case OP_LINEAR: macs += (int64_t)d.out_features * d.w_stride; break;
      default: break;
    }
  }
  return 2 * macs;
}

int mzx_net_fused_schedule(const mzx_net* net, int32_t recurrent, int32_t* slot_of_op, int32_t cap) {
  if (!net || !slot_of_op) return 0;
  const RzProgram& R = recurrent ? net->rz.recurrent : net->rz.initial;
  const int n = (int)(recurrent ? net->prog_recurrent.size() : net->prog_initial.size());
  if (!net->rz.ok || !R.ok || cap < n) return 0;
  for (int k = 0; k < n; ++k) slot_of_op[k] = -1;
  for (int k = 0; k < R.n_ops; ++k) {
    int slot = 0;
    for (int q = 0; q < R.order[k]; ++q) slot += (int)((R.ops[q].sched >> 16) & 1u);
    slot_of_op[R.first + k] = slot;
  }
  return R.n_slots;
}

int mzx_net_num_operators(const mzx_net* net, int32_t recurrent) {
  if (!net) return 0;
  return (int)(recurrent ? net->prog_recurrent.size() : net->prog_initial.size());
}

static int check_net_call(const mzx_net* net, int32_t batch, int64_t workspace_floats) {
  if (!net) { set_error("null network handle"); return MZX_ERR_INVALID; }
  if (!net->d_flat) { set_error("network has no weights bound (call mzx_net_set_weights)"); return MZX_ERR_INVALID; }
  if (batch < 1) { set_error("batch must be >= 1"); return MZX_ERR_INVALID; }
  if (workspace_floats < net_ws_per_sample(net) * (int64_t)batch) {
    set_error("workspace too small: %lld floats given, %lld needed", (long long)workspace_floats,
              (long long)(net_ws_per_sample(net) * (int64_t)batch));
    return MZX_ERR_WORKSPACE;
  }
  return MZX_OK;
}

int mzx_net_initial_inference(mzx_net* net, const float* d_observation, int32_t batch, float* d_value_logits,
                              float* d_reward_logits, float* d_policy_logits, float* d_hidden, float* d_workspace,
                              int64_t workspace_floats, void* stream) {
  int rc = check_net_call(net, batch, workspace_floats);
  if (rc) return rc;
  if (!d_observation || !d_value_logits || !d_policy_logits || !d_hidden) { set_error("null buffer"); return MZX_ERR_INVALID; }
  NetBuffers nb;
  nb.in = d_observation; nb.action = nullptr; nb.hidden = d_hidden; nb.value = d_value_logits;
  nb.reward = nullptr; nb.policy = d_policy_logits; nb.workspace = d_workspace;
  rc = run_network(net, false, nb, batch, (stream_t)stream);
  if (rc) return rc;
  if (d_reward_logits) {
    RewardFillOp f;
    f.y = d_reward_logits; f.batch = batch; f.full_support = net->full_support;
    MZX_TRY_LAUNCH(launch<256>(f, (stream_t)stream));
  }
  return MZX_OK;
}

int mzx_net_recurrent_inference(mzx_net* net, const float* d_hidden, const int32_t* d_action, int32_t batch,
                                float* d_value_logits, float* d_reward_logits, float* d_policy_logits,
                                float* d_next_hidden, float* d_workspace, int64_t workspace_floats, void* stream) {
  int rc = check_net_call(net, batch, workspace_floats);
  if (rc) return rc;
  if (!d_hidden || !d_action || !d_value_logits || !d_reward_logits || !d_policy_logits || !d_next_hidden) {
    set_error("null buffer");
    return MZX_ERR_INVALID;
  }
  NetBuffers nb;
  nb.in = d_hidden; nb.action = d_action; nb.hidden = d_next_hidden; nb.value = d_value_logits;
  nb.reward = d_reward_logits; nb.policy = d_policy_logits; nb.workspace = d_workspace;
  return run_network(net, true, nb, batch, (stream_t)stream);
}

int mzx_net_debug_prefix(mzx_net* net, int32_t recurrent, int32_t fused, int32_t n_ops, const float* d_input,
                         const int32_t* d_action, int32_t batch, float* d_out, int64_t out_floats, float* d_scratch,
                         int64_t scratch_floats, float* d_workspace, int64_t workspace_floats, void* stream) {
  int rc = check_net_call(net, batch, workspace_floats);
  if (rc) return rc;
  const int64_t need = (net->hidden_size + 2 * net->full_support + net->cfg.action_space_size) * (int64_t)batch;
  if (!d_input || !d_out || !d_scratch || scratch_floats < need) { set_error("debug_prefix: missing / short buffer (%lld scratch floats)", (long long)need); return MZX_ERR_WORKSPACE; }
  NetBuffers nb;
  nb.in = d_input; nb.action = d_action; nb.workspace = d_workspace;
  nb.hidden = d_scratch;
  nb.value = nb.hidden + net->hidden_size * batch;
  nb.reward = nb.value + (int64_t)net->full_support * batch;
  nb.policy = nb.reward + (int64_t)net->full_support * batch;
  return run_network_prefix(net, recurrent != 0, fused, n_ops, nb, batch, d_out, out_floats, (stream_t)stream);
}

// ----------------------------------------------------------------- search

int mzx_search_create(const mzx_search_config* cfg, mzx_net* net, mzx_search** out) {
  if (!cfg || !out) { set_error("mzx_search_create: null argument"); return MZX_ERR_INVALID; }
  if (cfg->num_trees < 1 || cfg->num_simulations < 0 || cfg->action_space_size < 1) {
    set_error("num_trees/num_simulations/action_space_size out of range");
    return MZX_ERR_INVALID;
  }
  if (cfg->num_players != 1 && cfg->num_players != 2) {
    // self_play.py:429-430
    set_error("More than two player mode not implemented.");
    return MZX_ERR_INVALID;
  }
  if (!cfg->h_pb_c_table || !cfg->h_sqrt_table) { set_error("pb_c / sqrt tables are required"); return MZX_ERR_INVALID; }
  if (net && net->cfg.action_space_size != cfg->action_space_size) { set_error("network/search action space mismatch"); return MZX_ERR_INVALID; }
  if (net && net->cfg.support_size != cfg->support_size) { set_error("network/search support size mismatch"); return MZX_ERR_INVALID; }
  mzx_search* s = new (std::nothrow) mzx_search();
  if (!s) { set_error("out of host memory"); return MZX_ERR_RUNTIME; }
  s->cfg = *cfg;
  s->net = net;
  const int n = cfg->num_simulations + 2;
  s->h_pbc.assign(cfg->h_pb_c_table, cfg->h_pb_c_table + (n - 1));
  s->h_sqrt.assign(cfg->h_sqrt_table, cfg->h_sqrt_table + (n - 1));
  s->h_pbc.push_back(0.0);
  s->h_sqrt.push_back(0.0);
  s->cfg.h_pb_c_table = nullptr;
  s->cfg.h_sqrt_table = nullptr;
  search_plan(s);
  if (int rc = upload_tables(s)) { mzx_search_destroy(s); return rc; }
#ifndef MZX_HOSTCHECK
  // 1: fully connected whole-search kernel, 2: residual whole-search kernels, 3: row-per-tree kernels around a
  // network that runs layer by layer on the streamed MFMA engine
  s->fused_ok = fc2_plan(s).ok ? 1 : (rz_search_supported(s) ? 2 : ((net && net->rb.ok && net->rb.recurrent.ok && row_search_supported(s->p)) ? 3 : 0));
#endif
  s->mode = s->fused_ok ? 1 : 0;
  *out = s;
  return MZX_OK;
}

void mzx_search_destroy(mzx_search* s) {
  if (s && s->d_tables) device_free(s->d_tables);
#ifndef MZX_HOSTCHECK
  if (s && s->side_stream) {      // the row-per-tree path's second stream (joined at the end of every run)
    (void)hipStreamSynchronize((hipStream_t)s->side_stream);
    if (s->ev_fork) (void)hipEventDestroy((hipEvent_t)s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy((hipEvent_t)s->ev_join);
    (void)hipStreamDestroy((hipStream_t)s->side_stream);
  }
#endif
  delete s;
}

int64_t mzx_search_arena_bytes(const mzx_search* s) { return s ? s->arena_bytes : 0; }

int mzx_search_fused_supported(const mzx_search* s) { return s ? s->fused_ok : 0; }

const char* mzx_search_kernel_name(const mzx_search* s) { return s ? s->last_kernel : ""; }

int mzx_search_arena_offsets(const mzx_search* s, int64_t out[8]) {
  if (!s || !out) { set_error("null argument"); return MZX_ERR_INVALID; }
  out[0] = s->off_tables; out[1] = s->off_trees; out[2] = s->off_hidden; out[3] = s->off_ws;
  out[4] = s->L.tree_bytes; out[5] = s->ws_floats * 4; out[6] = s->arena_bytes; out[7] = 0;
  return MZX_OK;
}

int mzx_tuning_set(const char* name, int32_t value) {
  const int k = tuning_find(name);
  if (k < 0) { set_error("unknown tuning entry '%s'", name ? name : "(null)"); return MZX_ERR_INVALID; }
  TuningEntry& e = tuning_table()[k];
  if (value < e.lo || value > e.hi) { set_error("tuning entry '%s' takes %d .. %d", e.name, e.lo, e.hi); return MZX_ERR_INVALID; }
  e.value = value;
  return MZX_OK;
}

int mzx_tuning_get(const char* name, int32_t* value, int32_t* dflt) {
  const int k = tuning_find(name);
  if (k < 0) { set_error("unknown tuning entry '%s'", name ? name : "(null)"); return MZX_ERR_INVALID; }
  if (value) *value = tuning_table()[k].value;
  if (dflt) *dflt = tuning_table()[k].dflt;
  return MZX_OK;
}

const char* mzx_tuning_name(int32_t index) { return (index >= 0 && index < TUNE_COUNT) ? tuning_table()[index].name : nullptr; }
const char* mzx_tuning_help(int32_t index) { return (index >= 0 && index < TUNE_COUNT) ? tuning_table()[index].what : nullptr; }

static void search_route_of(const mzx_search* s, int32_t out[8]) {
  for (int k = 0; k < 8; ++k) out[k] = 0;
#ifndef MZX_HOSTCHECK
  if (!s->net) return;
  int whole = 0;
  if ((s->mode & 1) && s->fused_ok == 2 && rz_enabled(s->net, true)) {
    const int route = wide_search_route(s);
    out[0] = route == ROUTE_RZ ? 1 : (route == ROUTE_ROWS ? 2 : 3);
    whole = route == ROUTE_RT;
  } else if ((s->mode & 1) && s->fused_ok == 1) {
    out[0] = 4;
  } else if ((s->mode & 1) && rb_enabled(s->net, true) && row_search_supported(s->p)) {
    whole = streamed_whole_search(s);
    out[0] = whole ? 3 : 2;
  }
  if (whole) rt_search_shape(s, out + 1);      // out[1 .. 6]
  if (out[0] == 2) {
    const int first = rb_split_first(s->net, s->p.num_trees, tune(TUNE_ROW_SPLIT_MIN));
    out[6] = first > 0 ? first : s->p.num_trees;
    out[7] = first > 0 ? s->p.num_trees - first : 0;
  }
#endif
}

int mzx_search_route(const mzx_search* s, int32_t out[8]) {
  if (!s || !out) { set_error("null argument"); return MZX_ERR_INVALID; }
  search_route_of(s, out);
  return MZX_OK;
}

int mzx_net_search_route(const mzx_net* net, int32_t num_trees, int32_t num_simulations, int32_t out[8]) {
  if (!net || !out || num_trees < 1 || num_simulations < 0) { set_error("null argument / sizes out of range"); return MZX_ERR_INVALID; }
  // a search descriptor without device tables: the planner only reads sizes (host-side, no GPU)
  mzx_search tmp;
  memset(&tmp.cfg, 0, sizeof(tmp.cfg));
  tmp.cfg.num_trees = num_trees; tmp.cfg.num_simulations = num_simulations;
  tmp.cfg.action_space_size = net->cfg.action_space_size; tmp.cfg.support_size = net->cfg.support_size;
  tmp.cfg.num_players = 1; tmp.cfg.tape_words = 16; tmp.cfg.discount = 1.0; tmp.cfg.root_exploration_fraction = 0.25;
  tmp.net = const_cast<mzx_net*>(net);
  search_plan(&tmp);
#ifndef MZX_HOSTCHECK
  tmp.fused_ok = fc2_plan(&tmp).ok ? 1 : (rz_search_supported(&tmp) ? 2 : ((net->rb.ok && net->rb.recurrent.ok && row_search_supported(tmp.p)) ? 3 : 0));
#endif
  tmp.mode = tmp.fused_ok ? 1 : 0;
  search_route_of(&tmp, out);
  return MZX_OK;
}

int mzx_search_set_mode(mzx_search* s, int32_t mode) {
  if (!s) { set_error("null search handle"); return MZX_ERR_INVALID; }
  if (mode < 0 || mode > 31) { set_error("mode is a 5-bit flag set"); return MZX_ERR_INVALID; }
#if !defined(MZX_HOSTCHECK) && defined(MZX_EXPERIMENT)
  if ((mode & 16) && !(s->fused_ok == 1 && fused_fc_supported(s))) { set_error("flag 16 selects the first-generation fully connected kernel"); return MZX_ERR_INVALID; }
#else
  if (mode & 16) { set_error("flag 16 (the first-generation fully connected kernel) needs an instrumented build (-DMZX_EXPERIMENT)"); return MZX_ERR_INVALID; }
#endif
  if ((mode & 1) && !s->fused_ok) { set_error("fused search kernel does not support this configuration"); return MZX_ERR_INVALID; }
  s->mode = mode;
  return MZX_OK;
}

static int check_search_call(const mzx_search* s, const void* d_arena, int64_t arena_bytes, bool need_net) {
  if (!s || !d_arena) { set_error("null search handle / arena"); return MZX_ERR_INVALID; }
  if (arena_bytes >= 0 && arena_bytes < s->arena_bytes) {
    set_error("arena too small: %lld bytes given, %lld needed", (long long)arena_bytes, (long long)s->arena_bytes);
    return MZX_ERR_WORKSPACE;
  }
  if (need_net && (!s->net || !s->net->d_flat)) { set_error("search needs a network with bound weights"); return MZX_ERR_INVALID; }
  if (s->device >= 0 && current_device() != s->device) {   // its tables (and the caller's arena) live on that device
    set_error("search handle was created on device %d, the current device is %d", s->device, current_device());
    return MZX_ERR_INVALID;
  }
  return MZX_OK;
}

int mzx_search_run(mzx_search* s, const mzx_search_io* io, void* d_arena, int64_t arena_bytes, void* stream) {
  int rc = check_search_call(s, d_arena, arena_bytes, true);
  if (rc) return rc;
  if (!io || !io->d_observation || !io->d_legal_actions || !io->d_to_play || !io->d_tape || !io->d_visit_counts ||
      !io->d_root_value || !io->d_info) {
    set_error("mzx_search_run: missing io buffer");
    return MZX_ERR_INVALID;
  }
#ifndef MZX_HOSTCHECK
  if ((s->mode & 1) && s->fused_ok == 1)
  {
    s->last_kernel = (s->mode & 16) ? "mzx::fused_fc_search" : "mzx::fc2_search_kernel";
    return (s->mode & 16) ? fused_fc_run(s, io, d_arena, (stream_t)stream) : fc2_run(s, io, d_arena, (stream_t)stream);
  }
  if ((s->mode & 1) && s->fused_ok == 2 && rz_enabled(s->net, true)) {
    // a wide network: the tower arithmetic at every shard size (wide_search_route, mzx_row_search.h) -- all simulations in
    // one launch of rt_search_kernel, or the trunks as towers between the row-per-tree kernels
    const int route = wide_search_route(s);
    if (route != ROUTE_RZ) {
      s->last_kernel = "mzx::rb_tower_kernel / mzx::rb_gemm_kernel / mzx::rb_gemm_multi_kernel (streamed FP32-MFMA trunks, layers, head MLP levels) between mzx::row_select_kernel / mzx::row_expand_backprop_kernel";
      return search_run_rows(s, io, d_arena, (stream_t)stream, nullptr, true, route == ROUTE_RT);
    }
    return rz_search_run(s, io, d_arena, (stream_t)stream);
  }
  if ((s->mode & 1) && rb_enabled(s->net, true) && row_search_supported(s->p)) {
    s->last_kernel = "mzx::rb_tower_kernel / mzx::rb_gemm_kernel / mzx::rb_gemm_multi_kernel (streamed FP32-MFMA trunks, layers, head MLP levels) between mzx::row_select_kernel / mzx::row_expand_backprop_kernel";
    // (renames last_kernel when it runs two half-shards or the whole-search kernel)
    return search_run_rows(s, io, d_arena, (stream_t)stream, nullptr, false, streamed_whole_search(s));
  }
#endif
  s->last_kernel = "one kernel per step of a simulation (select / network / expand + back-propagate)";
#ifndef MZX_HOSTCHECK
  if (rb_enabled(s->net, true)) s->last_kernel = "mzx::rb_tower_kernel / mzx::rb_gemm_kernel / mzx::rb_gemm_multi_kernel (streamed FP32-MFMA trunks, layers, head MLP levels) between one-thread-per-tree kernels";
#endif
  return search_run_generic(s, io, d_arena, (stream_t)stream);
}

int mzx_search_run_from_roots(mzx_search* s, const mzx_search_io* io, const float* d_root_hidden,
                              const double* d_root_priors, const double* d_root_reward, void* d_arena,
                              int64_t arena_bytes, void* stream) {
  int rc = check_search_call(s, d_arena, arena_bytes, true);
  if (rc) return rc;
  if (!io || !io->d_legal_actions || !io->d_to_play || !io->d_tape || !io->d_visit_counts || !io->d_root_value ||
      !io->d_info || !d_root_hidden || !d_root_priors || !d_root_reward) {
    set_error("mzx_search_run_from_roots: missing buffer");
    return MZX_ERR_INVALID;
  }
  RootOverride ov;
  ov.hidden = d_root_hidden; ov.priors = d_root_priors; ov.reward = d_root_reward;
#ifndef MZX_HOSTCHECK
  // the simulations run on the kernel mzx_search_run would use: the residual whole-search kernels read the roots from
  // the arena, the fully connected one (second generation) takes them as launch arguments instead of running
  // initial_inference; the first-generation kernel (flag 16, A/B only) stays on the per-operator path here
  if ((s->mode & 1) && s->fused_ok == 1 && !(s->mode & 16)) {
    s->last_kernel = "mzx::fc2_search_kernel";
    return fc2_run(s, io, d_arena, (stream_t)stream, &ov);
  }
  if ((s->mode & 1) && s->fused_ok == 2 && rz_enabled(s->net, true)) {
    const int route = wide_search_route(s);
    if (route != ROUTE_RZ) {
      s->last_kernel = "mzx::rb_tower_kernel / mzx::rb_gemm_kernel / mzx::rb_gemm_multi_kernel (streamed FP32-MFMA trunks, layers, head MLP levels) between mzx::row_select_kernel / mzx::row_expand_backprop_kernel";
      return search_run_rows(s, io, d_arena, (stream_t)stream, &ov, true, route == ROUTE_RT);
    }
    return rz_search_run(s, io, d_arena, (stream_t)stream, &ov);
  }
  if ((s->mode & 1) && rb_enabled(s->net, true) && row_search_supported(s->p)) {
    s->last_kernel = "mzx::rb_tower_kernel / mzx::rb_gemm_kernel / mzx::rb_gemm_multi_kernel (streamed FP32-MFMA trunks, layers, head MLP levels) between mzx::row_select_kernel / mzx::row_expand_backprop_kernel";
    return search_run_rows(s, io, d_arena, (stream_t)stream, &ov, false, streamed_whole_search(s));
  }
#endif
  s->last_kernel = "one kernel per step of a simulation (select / network / expand + back-propagate)";
  return search_run_generic(s, io, d_arena, (stream_t)stream, &ov);
}

int mzx_search_lockstep_begin(mzx_search* s, const mzx_search_io* io, const double* d_root_priors,
                              const double* d_root_reward, void* d_arena, int64_t arena_bytes, void* stream) {
  int rc = check_search_call(s, d_arena, arena_bytes, false);
  if (rc) return rc;
  if (!io || !io->d_legal_actions || !io->d_to_play || !d_root_priors) { set_error("lockstep_begin: missing buffer"); return MZX_ERR_INVALID; }
  rc = ensure_tables(s, d_arena, (stream_t)stream);
  if (rc) return rc;
  const ArenaView v = arena_view(s, d_arena);
  RootInitOp ri;
  ri.arena = v.arena; ri.p = v.p; ri.value_logits = nullptr; ri.policy_logits = nullptr; ri.ext_priors = d_root_priors; ri.ext_root_reward = d_root_reward;
  ri.legal = io->d_legal_actions; ri.to_play = io->d_to_play; ri.noise = io->d_noise; ri.root_predicted_value = nullptr;
  MZX_TRY_LAUNCH(launch<64>(ri, (stream_t)stream));
  return MZX_OK;
}

int mzx_search_lockstep_select(mzx_search* s, const mzx_search_io* io, int32_t* d_parent, int32_t* d_action,
                               int32_t* d_leaf, void* d_arena, void* stream) {
  int rc = check_search_call(s, d_arena, -1, false);
  if (rc) return rc;
  if (!io || !io->d_tape || !d_parent || !d_action || !d_leaf) { set_error("lockstep_select: missing buffer"); return MZX_ERR_INVALID; }
  const ArenaView v = arena_view(s, d_arena);
  SelectOp sel;
  sel.arena = v.arena; sel.p = v.p; sel.tape = io->d_tape;
  sel.sel_parent = d_parent; sel.sel_action = d_action; sel.sel_leaf = d_leaf;
  MZX_TRY_LAUNCH(launch<64>(sel, (stream_t)stream));
  return MZX_OK;
}

int mzx_search_lockstep_apply(mzx_search* s, const double* d_value, const double* d_reward, const double* d_priors,
                              void* d_arena, void* stream) {
  int rc = check_search_call(s, d_arena, -1, false);
  if (rc) return rc;
  if (!d_value || !d_reward || !d_priors) { set_error("lockstep_apply: missing buffer"); return MZX_ERR_INVALID; }
  const ArenaView v = arena_view(s, d_arena);
  ExpandBackpropOp eb;
  eb.arena = v.arena; eb.p = v.p; eb.value_logits = nullptr; eb.reward_logits = nullptr; eb.policy_logits = nullptr;
  eb.ext_value = d_value; eb.ext_reward = d_reward; eb.ext_priors = d_priors;
  MZX_TRY_LAUNCH(launch<64>(eb, (stream_t)stream));
  return MZX_OK;
}

int mzx_search_finish(mzx_search* s, const mzx_search_io* io, void* d_arena, void* stream) {
  int rc = check_search_call(s, d_arena, -1, false);
  if (rc) return rc;
  if (!io || !io->d_visit_counts || !io->d_root_value || !io->d_info) { set_error("finish: missing buffer"); return MZX_ERR_INVALID; }
  return search_finish(s, io, d_arena, (stream_t)stream);
}

int mzx_search_dump(mzx_search* s, const mzx_tree_dump* dump, void* d_arena, void* stream) {
  int rc = check_search_call(s, d_arena, -1, false);
  if (rc) return rc;
  if (!dump) { set_error("dump: null"); return MZX_ERR_INVALID; }
  const ArenaView v = arena_view(s, d_arena);
  DumpOp op;
  op.arena = v.arena; op.p = v.p; op.d = *dump;
  MZX_TRY_LAUNCH(launch<64>(op, (stream_t)stream));
  return MZX_OK;
}

// ------------------------------------------------------- observation pipeline

static bool obs_layout_ok(const mzx_obs_layout* L) {
  return L && L->channels >= 1 && L->height >= 1 && L->width >= 1 && L->stacked_observations >= 0 &&
         L->action_space_size >= 1 && L->num_games >= 1 && L->ring >= 1 &&
         (int64_t)L->channels * (L->stacked_observations + 1) + L->stacked_observations < (1 << 20);
}

int64_t mzx_obs_stacked_floats(const mzx_obs_layout* L) {
  if (!obs_layout_ok(L)) return 0;
  return ((int64_t)L->channels * (L->stacked_observations + 1) + L->stacked_observations) * L->height * L->width;
}

extern "C++" {
template <int VEC>
static int obs_stack_launch(const mzx_obs_layout* L, const float* d_frames, const int32_t* d_actions,
                            const int32_t* d_game, const int32_t* d_time, int32_t time0, int32_t n_out, float* d_out,
                            stream_t stream) {
  ObsStackOp<VEC> op;
  op.frames = d_frames; op.actions = d_actions; op.game = d_game; op.time = d_time; op.out = d_out;
  op.time0 = time0; op.C = L->channels; op.hwv = L->height * L->width / VEC; op.k = L->stacked_observations;
  op.A = L->action_space_size; op.G = L->num_games; op.ring = L->ring; op.n_out = n_out;
  op.c_out = L->channels * (L->stacked_observations + 1) + L->stacked_observations;
  op.pieces = (op.hwv + 64 * ObsStackOp<VEC>::UNROLL - 1) / (64 * ObsStackOp<VEC>::UNROLL);
  if ((int64_t)n_out * op.c_out * op.pieces >= (int64_t)1 << 26) {   // 32-bit group index; split the call
    set_error("obs_stack: %d samples x %d planes exceed one launch; stack in smaller batches", n_out, op.c_out);
    return MZX_ERR_INVALID;
  }
  MZX_TRY_LAUNCH(launch<256>(op, stream));
  return MZX_OK;
}
}  // extern "C++"

int mzx_obs_stack(const mzx_obs_layout* L, const float* d_frames, const int32_t* d_actions, const int32_t* d_game,
                  const int32_t* d_time, int32_t time0, int32_t n_out, float* d_out, void* stream) {
  if (!obs_layout_ok(L)) { set_error("obs_stack: invalid layout"); return MZX_ERR_INVALID; }
  if (n_out < 0 || time0 < 0) { set_error("obs_stack: negative n_out / time0"); return MZX_ERR_INVALID; }
  if (n_out == 0) return MZX_OK;
  if (!d_frames || !d_out || (L->stacked_observations > 0 && !d_actions)) { set_error("obs_stack: missing buffer"); return MZX_ERR_INVALID; }
  if (!d_time && L->ring < L->stacked_observations + 1 && (int64_t)time0 + (n_out - 1) / L->num_games >= L->ring) {
    set_error("obs_stack: ring %d cannot hold %d stacked observations at index %d", L->ring, L->stacked_observations, time0);
    return MZX_ERR_INVALID;
  }
  const bool vec = (L->height * L->width) % 4 == 0 && ((uintptr_t)d_frames % 16) == 0 && ((uintptr_t)d_out % 16) == 0;
  return vec ? obs_stack_launch<4>(L, d_frames, d_actions, d_game, d_time, time0, n_out, d_out, (stream_t)stream)
             : obs_stack_launch<1>(L, d_frames, d_actions, d_game, d_time, time0, n_out, d_out, (stream_t)stream);
}

int mzx_support_to_scalar(const float* d_logits, int32_t rows, int32_t support_size, float* d_out, void* stream) {
  if (rows < 0 || support_size < 0) { set_error("support_to_scalar: negative argument"); return MZX_ERR_INVALID; }
  if (rows == 0) return MZX_OK;
  if (!d_logits || !d_out) { set_error("support_to_scalar: missing buffer"); return MZX_ERR_INVALID; }
  SupportToScalarOp op;
  op.logits = d_logits; op.out = d_out; op.rows = rows; op.support_size = support_size;
  MZX_TRY_LAUNCH(launch<64>(op, (stream_t)stream));
  return MZX_OK;
}

// ------------------------------------------------------------- random streams

static int rng_check(const mzx_rng* r, const int32_t* idx, int32_t count) {
  if (!r || (count > 0 && !idx) || count < 0) { set_error("rng: null / negative argument"); return MZX_ERR_INVALID; }
  const int32_t n = (int32_t)r->streams.size();
  for (int32_t k = 0; k < count; ++k)
    if (idx[k] < 0 || idx[k] >= n) { set_error("rng: stream index %d out of range (%d streams)", idx[k], n); return MZX_ERR_INVALID; }
  return MZX_OK;
}

int mzx_rng_create(int32_t num_streams, mzx_rng** out) {
  if (num_streams < 1 || !out) { set_error("mzx_rng_create: invalid argument"); return MZX_ERR_INVALID; }
  mzx_rng* r = new (std::nothrow) mzx_rng();
  if (!r) { set_error("out of host memory"); return MZX_ERR_RUNTIME; }
  r->streams.resize(num_streams);
  for (int32_t i = 0; i < num_streams; ++i) r->streams[i].seed((uint32_t)i);
  *out = r;
  return MZX_OK;
}

void mzx_rng_destroy(mzx_rng* r) { delete r; }

int mzx_rng_seed(mzx_rng* r, int32_t first, int32_t count, const uint32_t* seeds) {
  if (!r || !seeds || first < 0 || count < 0 || first + count > (int32_t)r->streams.size()) { set_error("mzx_rng_seed: range"); return MZX_ERR_INVALID; }
  for (int32_t k = 0; k < count; ++k) r->streams[first + k].seed(seeds[k]);
  return MZX_OK;
}

int mzx_rng_get_state(const mzx_rng* r, int32_t i, uint32_t* key, int32_t* pos, int32_t* has_gauss, double* gauss) {
  if (!r || i < 0 || i >= (int32_t)r->streams.size() || !key || !pos || !has_gauss || !gauss) { set_error("mzx_rng_get_state: argument"); return MZX_ERR_INVALID; }
  const Mt19937& m = r->streams[i];
  memcpy(key, m.key, sizeof(m.key));
  *pos = m.pos; *has_gauss = m.has_gauss; *gauss = m.gauss;
  return MZX_OK;
}

int mzx_rng_set_state(mzx_rng* r, int32_t i, const uint32_t* key, int32_t pos, int32_t has_gauss, double gauss) {
  if (!r || i < 0 || i >= (int32_t)r->streams.size() || !key || pos < 0 || pos > 624) { set_error("mzx_rng_set_state: argument"); return MZX_ERR_INVALID; }
  Mt19937& m = r->streams[i];
  memcpy(m.key, key, sizeof(m.key));
  m.pos = pos; m.has_gauss = has_gauss ? 1 : 0; m.gauss = gauss;
  return MZX_OK;
}

int mzx_rng_root_draws(mzx_rng* r, const int32_t* idx, int32_t count, double alpha, const int32_t* n_legal,
                       int32_t action_space_size, double* noise, int32_t tape_words, uint32_t* tape, int32_t n_threads) {
  int rc = rng_check(r, idx, count);
  if (rc) return rc;
  if ((noise && !n_legal) || action_space_size < 1 || tape_words < 0 || (tape_words > 0 && !tape)) { set_error("mzx_rng_root_draws: argument"); return MZX_ERR_INVALID; }
  if (noise)
    for (int32_t k = 0; k < count; ++k)
      if (n_legal[k] < 1 || n_legal[k] > action_space_size) { set_error("n_legal[%d] = %d out of range", k, n_legal[k]); return MZX_ERR_INVALID; }
  rng_parallel(r, count, n_threads, [=](int lo, int hi) {
    for (int k = lo; k < hi; ++k) {
      if (k + 1 < hi) {   // the next game's generator state (2.5 KB each: a shard's bank does not stay in cache)
        const Mt19937& nx = r->streams[idx[k + 1]];
        __builtin_prefetch(&nx.pos);
        __builtin_prefetch(&nx.key[nx.pos < 624 ? nx.pos : 0]);
      }
      Mt19937& m = r->streams[idx[k]];
      if (noise) {  // RandomState.dirichlet([alpha] * n): gammas, then multiplication by 1 / sum
        double* out = noise + (size_t)k * action_space_size;
        const int n = n_legal[k];
        double acc = 0.0;
        for (int j = 0; j < n; ++j) { out[j] = m.standard_gamma(alpha); acc = acc + out[j]; }
        const double invacc = 1 / acc;
        for (int j = 0; j < n; ++j) out[j] = out[j] * invacc;
        for (int j = n; j < action_space_size; ++j) out[j] = 0.0;
      }
      if (tape_words > 0) m.peek(tape_words, tape + (size_t)k * tape_words);   // generator state untouched
    }
  });
  return MZX_OK;
}

int mzx_rng_advance(mzx_rng* r, const int32_t* idx, int32_t count, const int32_t* words) {
  int rc = rng_check(r, idx, count);
  if (rc) return rc;
  if (count > 0 && !words) { set_error("mzx_rng_advance: null"); return MZX_ERR_INVALID; }
  for (int32_t k = 0; k < count; ++k) {
    Mt19937& m = r->streams[idx[k]];
    for (int32_t j = 0; j < words[k]; ++j) m.next32();
  }
  return MZX_OK;
}

int mzx_rng_random_sample(mzx_rng* r, const int32_t* idx, int32_t count, double* out) {
  int rc = rng_check(r, idx, count);
  if (rc) return rc;
  if (count > 0 && !out) { set_error("mzx_rng_random_sample: null"); return MZX_ERR_INVALID; }
  for (int32_t k = 0; k < count; ++k) out[k] = r->streams[idx[k]].next_double();
  return MZX_OK;
}

int mzx_rng_randint(mzx_rng* r, const int32_t* idx, int32_t count, const int32_t* n, int32_t* out) {
  int rc = rng_check(r, idx, count);
  if (rc) return rc;
  if (count > 0 && (!n || !out)) { set_error("mzx_rng_randint: null"); return MZX_ERR_INVALID; }
  for (int32_t k = 0; k < count; ++k) {
    if (n[k] < 1) { set_error("randint(0, %d): empty range", n[k]); return MZX_ERR_INVALID; }
    out[k] = (int32_t)r->streams[idx[k]].bounded((uint32_t)n[k]);
  }
  return MZX_OK;
}

int mzx_rng_choice_weighted(mzx_rng* r, const int32_t* idx, int32_t count, const double* weights,
                            int32_t row_stride, const int32_t* n, int32_t* out) {
  int rc = rng_check(r, idx, count);
  if (rc) return rc;
  if (count > 0 && (!weights || !n || !out)) { set_error("mzx_rng_choice_weighted: null"); return MZX_ERR_INVALID; }
  for (int32_t k = 0; k < count; ++k)
    if (n[k] < 1 || n[k] > row_stride || n[k] > 4096) { set_error("choice: row %d has %d candidates (stride %d)", k, n[k], row_stride); return MZX_ERR_INVALID; }
  for (int32_t k = 0; k < count; ++k) {
    const double* w = weights + (size_t)k * row_stride;
    const int m = n[k];
    double total = 0.0;                       // Python's sum(dist): 0 + d0 + d1 + ...
    for (int j = 0; j < m; ++j) total = total + w[j];
    double cdf[4096];
    double acc = 0.0;                         // ndarray.cumsum of p = dist / total
    for (int j = 0; j < m; ++j) { acc = acc + w[j] / total; cdf[j] = acc; }
    const double last = cdf[m - 1];
    const double u = r->streams[idx[k]].next_double();
    int pos = 0;                              // searchsorted(u, side="right") on cdf / cdf[-1]
    for (int j = 0; j < m; ++j) pos += (cdf[j] / last <= u) ? 1 : 0;
    out[k] = pos;
  }
  return MZX_OK;
}

}  // extern "C"

namespace mzx {

// SelfPlay.select_action (self_play.py:222-245) for the games [lo, hi) of a move: the streams first consume the tie-break
// words the search used (mzx_rng_advance), then every game's action is drawn.  Shared by mzx_selfplay_select and the round
// loop of mzx_actor.h (where it runs inside the per-move region together with the game step).
struct SelectArgs {
  mzx_rng* r; const int32_t* idx; const int32_t* legal_all; const int32_t* n_legal; const int32_t* words; const int32_t* visit_counts;
  const double* temperature; const double* pow_table; int32_t table_stride; const double* table_temperatures; int32_t num_temperatures;
  int64_t* action; int32_t A;
};

inline void select_range(const SelectArgs& s, int lo, int hi) {
  mzx_rng* r = s.r;
  const int32_t* idx = s.idx;
  const int A = s.A;
  double cdf[4096];
  for (int k = lo; k < hi; ++k) {
    if (k + 1 < hi) {
      const Mt19937& nx = r->streams[idx[k + 1]];
      __builtin_prefetch(&nx.pos);
      __builtin_prefetch(&nx.key[nx.pos < 624 ? nx.pos : 0]);
    }
    Mt19937& g = r->streams[idx[k]];
    if (s.words) for (int32_t j = 0; j < s.words[k]; ++j) g.next32();        // what the search consumed (mzx_rng_advance)
    const int32_t* legal = s.legal_all + (size_t)k * A;
    const int32_t* vis = s.visit_counts + (size_t)k * A;
    const int n = s.n_legal[k];
    const double t = s.temperature[k];
    int pos = 0;
    if (t == 0.0) {                            // actions[numpy.argmax(visit_counts)]: the first maximum
      int32_t best = vis[legal[0]];
      for (int j = 1; j < n; ++j) if (vis[legal[j]] > best) { best = vis[legal[j]]; pos = j; }
    } else if (t == MZX_INF) {                 // numpy.random.choice(actions)
      pos = (int)g.bounded((uint32_t)n);
    } else {                                   // numpy.random.choice(actions, p=dist / sum(dist)): mzx_rng_choice_weighted
      int row = 0;
      for (int q = 0; q < s.num_temperatures; ++q) if (s.table_temperatures[q] == t) row = q;
      const double* tab = s.pow_table + (size_t)row * s.table_stride;
      double total = 0.0;
      for (int j = 0; j < n; ++j) total = total + tab[vis[legal[j]]];
      double acc = 0.0;
      for (int j = 0; j < n; ++j) { acc = acc + tab[vis[legal[j]]] / total; cdf[j] = acc; }
      const double last = cdf[n - 1];
      const double u = g.next_double();
      for (int j = 0; j < n; ++j) pos += (cdf[j] / last <= u) ? 1 : 0;
    }
    s.action[k] = legal[pos < n ? pos : n - 1];
  }
}

// the argument checks of mzx_selfplay_select (temperatures have a power table row, visit counts lie inside it)
inline int select_check(const int32_t* legal_all, const int32_t* n_legal, const int32_t* visit_counts, const double* temperature,
                        int B, int A, int32_t table_stride, const double* table_temperatures, int32_t num_temperatures) {
  for (int k = 0; k < B; ++k) {
    if (n_legal[k] < 1 || n_legal[k] > A) { set_error("n_legal[%d] = %d out of range", k, n_legal[k]); return MZX_ERR_INVALID; }
    const double t = temperature[k];
    if (t == 0.0 || t == MZX_INF) continue;
    int row = -1;
    for (int q = 0; q < num_temperatures; ++q) if (table_temperatures[q] == t) row = q;
    if (row < 0) { set_error("mzx_selfplay_select: no power table for temperature %g", t); return MZX_ERR_INVALID; }
    const int32_t* legal = legal_all + (size_t)k * A;
    const int32_t* vis = visit_counts + (size_t)k * A;
    for (int j = 0; j < n_legal[k]; ++j)
      if (vis[legal[j]] < 0 || vis[legal[j]] >= table_stride) { set_error("visit count %d outside the power table", vis[legal[j]]); return MZX_ERR_INVALID; }
  }
  return MZX_OK;
}

}  // namespace mzx

extern "C" {

// ------------------------------------------------------------- a self-play move of a shard behind two calls

static int move_check(const mzx_rng* r, const mzx_move* m) {
  if (!m) { set_error("mzx_move: null"); return MZX_ERR_INVALID; }
  if (m->num_games < 1 || m->action_space_size < 1 || m->tape_words < 0) { set_error("mzx_move: sizes"); return MZX_ERR_INVALID; }
  if (!m->legal_actions) { set_error("mzx_move: legal_actions missing"); return MZX_ERR_INVALID; }
  return rng_check(r, m->streams, m->num_games);
}

// host address of the field whose device address is `d` (NULL when the field lies outside the staged block)
static void* move_host(const void* d, const void* d_block, void* h_block, int64_t bytes) {
  if (!d) return nullptr;
  const ptrdiff_t off = (const char*)d - (const char*)d_block;
  if (off < 0 || off >= bytes) return nullptr;
  return (char*)h_block + off;
}

int mzx_selfplay_search(mzx_search* s, mzx_rng* r, const mzx_move* m, int32_t* n_legal, void* d_arena,
                        int64_t arena_bytes, void* stream) {
  int rc = move_check(r, m);
  if (rc) return rc;
  if (!s || !n_legal || !m->to_play || !m->h_in || !m->d_in || !m->h_out || !m->d_out || m->in_bytes < 1 || m->out_bytes < 1) {
    set_error("mzx_selfplay_search: missing buffer");
    return MZX_ERR_INVALID;
  }
  const int B = m->num_games, A = m->action_space_size, W = m->tape_words;
  const mzx_search_io& io = m->io;
  double* h_noise = (double*)move_host(io.d_noise, m->d_in, m->h_in, m->in_bytes);
  int32_t* h_legal = (int32_t*)move_host(io.d_legal_actions, m->d_in, m->h_in, m->in_bytes);
  int32_t* h_to_play = (int32_t*)move_host(io.d_to_play, m->d_in, m->h_in, m->in_bytes);
  uint32_t* h_tape = (uint32_t*)move_host(io.d_tape, m->d_in, m->h_in, m->in_bytes);
  float* h_obs = (float*)move_host(io.d_observation, m->d_in, m->h_in, m->in_bytes);
  if (!h_legal || !h_to_play || (W > 0 && !h_tape) || (m->observation && !h_obs) || (!m->observation && !io.d_observation) ||
      (m->add_exploration_noise ? !h_noise : io.d_noise != nullptr)) {
    set_error("mzx_selfplay_search: io fields must lie inside the staged input block");
    return MZX_ERR_INVALID;
  }
  // self_play.py:296-301 on the padded lists
  for (int k = 0; k < B; ++k) {
    const int32_t* row = m->legal_actions + (size_t)k * A;
    int n = 0;
    while (n < A && row[n] >= 0) ++n;
    if (n == 0) { set_error("Legal actions should not be an empty array. Got [] (game %d).", k); return MZX_ERR_INVALID; }
    for (int j = 0; j < A; ++j)
      if (j < n ? row[j] >= A : row[j] >= 0) {
        set_error("Legal actions should be a subset of the action space (padded with -1 at the end); game %d.", k);
        return MZX_ERR_INVALID;
      }
    n_legal[k] = n;
  }
  const double alpha = m->dirichlet_alpha;
  const int32_t* idx = m->streams;
  rng_parallel(r, B, m->num_threads, [=](int lo, int hi) {
    for (int k = lo; k < hi; ++k) {
      if (k + 1 < hi) {   // the next game's generator state (2.5 KB each, a shard's bank does not stay in cache)
        const Mt19937& nx = r->streams[idx[k + 1]];
        __builtin_prefetch(&nx.pos);
        __builtin_prefetch(&nx.key[nx.pos < 624 ? nx.pos : 0]);
      }
      Mt19937& g = r->streams[idx[k]];
      if (h_noise) {  // RandomState.dirichlet([alpha] * n): the statements of mzx_rng_root_draws
        double* out = h_noise + (size_t)k * A;
        const int n = n_legal[k];
        double acc = 0.0;
        for (int j = 0; j < n; ++j) { out[j] = g.standard_gamma(alpha); acc = acc + out[j]; }
        const double invacc = 1 / acc;
        for (int j = 0; j < n; ++j) out[j] = out[j] * invacc;
        for (int j = n; j < A; ++j) out[j] = 0.0;
      }
      if (W > 0) g.peek(W, h_tape + (size_t)k * W);
    }
  });
  memcpy(h_legal, m->legal_actions, sizeof(int32_t) * (size_t)B * A);
  memcpy(h_to_play, m->to_play, sizeof(int32_t) * (size_t)B);
  if (m->observation) memcpy(h_obs, m->observation, sizeof(float) * (size_t)B * m->observation_floats);
  const bool staged = m->h_in != m->d_in;
  if (staged) MZX_TRY_LAUNCH(copy_h2d(m->d_in, m->h_in, (size_t)m->in_bytes, (stream_t)stream));
  // (instrumented builds only, -DMZX_EXPERIMENT: the host envelope of a move timed without the search)
  if (!exp_int("MZX_MOVE_NO_SEARCH", 0)) {
    rc = mzx_search_run(s, &io, d_arena, arena_bytes, stream);
    if (rc) return rc;
  }
  if (m->h_out != m->d_out) MZX_TRY_LAUNCH(copy_d2h(m->h_out, m->d_out, (size_t)m->out_bytes, (stream_t)stream));
  if (!(m->flags & MZX_MOVE_NO_SYNC)) MZX_TRY_LAUNCH(stream_sync((stream_t)stream));
  return MZX_OK;
}

int mzx_selfplay_select(mzx_rng* r, const mzx_move* m, const int32_t* n_legal, const int32_t* words,
                        const int32_t* visit_counts, const double* temperature, const double* pow_table,
                        int32_t table_stride, const double* table_temperatures, int32_t num_temperatures,
                        int64_t* action) {
  int rc = move_check(r, m);
  if (rc) return rc;
  if (!n_legal || !visit_counts || !temperature || !action || num_temperatures < 0 ||
      (num_temperatures > 0 && (!pow_table || !table_temperatures || table_stride < 1))) {
    set_error("mzx_selfplay_select: missing argument");
    return MZX_ERR_INVALID;
  }
  const int B = m->num_games, A = m->action_space_size;
  if (A > 4096) { set_error("mzx_selfplay_select: more than 4096 actions"); return MZX_ERR_INVALID; }
  rc = select_check(m->legal_actions, n_legal, visit_counts, temperature, B, A, table_stride, table_temperatures, num_temperatures);
  if (rc) return rc;
  SelectArgs sa;
  sa.r = r; sa.idx = m->streams; sa.legal_all = m->legal_actions; sa.n_legal = n_legal; sa.words = words; sa.visit_counts = visit_counts;
  sa.temperature = temperature; sa.pow_table = pow_table; sa.table_stride = table_stride; sa.table_temperatures = table_temperatures;
  sa.num_temperatures = num_temperatures; sa.action = action; sa.A = A;
  rng_parallel(r, B, m->num_threads, [=](int lo, int hi) { select_range(sa, lo, hi); });
  return MZX_OK;
}

// ------------------------------------------------------------- replay hand-off: initial PER priorities on the device

int mzx_replay_priorities(const double* d_root_values, const double* d_rewards, const int32_t* d_to_play, int32_t num_games,
                          int32_t moves, int32_t td_steps, const double* d_discount_pow, double per_alpha, double* d_targets,
                          float* d_priorities, float* d_game_priority, void* stream) {
  if (num_games < 0 || moves < 0 || td_steps < 0) { set_error("mzx_replay_priorities: negative argument"); return MZX_ERR_INVALID; }
  if (num_games == 0 || moves == 0) return MZX_OK;
  if (!d_root_values || !d_rewards || !d_to_play || !d_discount_pow || !d_priorities) { set_error("mzx_replay_priorities: missing buffer"); return MZX_ERR_INVALID; }
  ReplayPriorityOp op;
  op.root_values = d_root_values; op.rewards = d_rewards; op.to_play = d_to_play; op.discount_pow = d_discount_pow;
  op.targets = d_targets; op.priorities = d_priorities; op.per_alpha = per_alpha;
  op.num_games = num_games; op.moves = moves; op.td_steps = td_steps;
  MZX_TRY_LAUNCH(launch<256>(op, (stream_t)stream));
  if (d_game_priority) {
    ReplayGameMaxOp mx;
    mx.priorities = d_priorities; mx.game_priority = d_game_priority; mx.num_games = num_games; mx.moves = moves;
    MZX_TRY_LAUNCH(launch<64>(mx, (stream_t)stream));
  }
  return MZX_OK;
}

}  // extern "C"

// ------------------------------------------------------------- natively stepped games + the round loop of a shard
#include "mzx_actor.h"

extern "C" {

int mzx_game_create(const char* kind, int32_t num_games, const uint32_t* seeds, const int32_t* observation_shape,
                    int32_t action_space_size, int32_t num_players, mzx_game** out) {
  if (!out) { set_error("mzx_game_create: null out"); return MZX_ERR_INVALID; }
  std::string err;
  mzx_game* g = game_make(kind, num_games, seeds, observation_shape, action_space_size, num_players, err);
  if (!g) { set_error("%s", err.c_str()); return MZX_ERR_INVALID; }
  *out = g;
  return MZX_OK;
}

void mzx_game_destroy(mzx_game* g) { delete g; }

int mzx_game_info(const mzx_game* g, int32_t out[8]) {
  if (!g || !out) { set_error("mzx_game_info: null"); return MZX_ERR_INVALID; }
  out[0] = g->shape[0]; out[1] = g->shape[1]; out[2] = g->shape[2]; out[3] = g->num_actions; out[4] = g->num_players;
  out[5] = g->obs_dtype; out[6] = g->reward_is_int; out[7] = g->always_all_legal;
  return MZX_OK;
}

int mzx_game_reset(mzx_game* g, const int32_t* games, int32_t count) {
  if (!g || count < 0) { set_error("mzx_game_reset: argument"); return MZX_ERR_INVALID; }
  if (!games) { g->reset(nullptr, g->num_games); return MZX_OK; }
  for (int32_t k = 0; k < count; ++k)
    if (games[k] < 0 || games[k] >= g->num_games) { set_error("mzx_game_reset: game %d out of range", games[k]); return MZX_ERR_INVALID; }
  g->reset(games, count);
  return MZX_OK;
}

int mzx_game_observe(const mzx_game* g, float* out) {
  if (!g || !out) { set_error("mzx_game_observe: null"); return MZX_ERR_INVALID; }
  g->observe(0, g->num_games, out);
  return MZX_OK;
}

int mzx_game_legal_actions(const mzx_game* g, int32_t* out) {
  if (!g || !out) { set_error("mzx_game_legal_actions: null"); return MZX_ERR_INVALID; }
  g->legal_actions(0, g->num_games, out);
  return MZX_OK;
}

int mzx_game_to_play(const mzx_game* g, int32_t* out) {
  if (!g || !out) { set_error("mzx_game_to_play: null"); return MZX_ERR_INVALID; }
  g->to_play(0, g->num_games, out);
  return MZX_OK;
}

int mzx_game_step(mzx_game* g, const int64_t* actions, const uint8_t* active, double* reward, uint8_t* done) {
  if (!g || !actions || !reward || !done) { set_error("mzx_game_step: null"); return MZX_ERR_INVALID; }
  for (int32_t k = 0; k < g->num_games; ++k)
    if ((!active || active[k]) && (actions[k] < 0 || actions[k] >= g->num_actions)) {
      set_error("mzx_game_step: action %lld of game %d outside the action space", (long long)actions[k], k);
      return MZX_ERR_INVALID;
    }
  g->step(0, g->num_games, actions, active, reward, done);
  return MZX_OK;
}

int mzx_actor_create(const mzx_actor_config* c, mzx_actor** out) {
  if (!c || !out || !c->game || !c->search || !c->bank || !c->streams || !c->d_arena) { set_error("mzx_actor_create: missing argument"); return MZX_ERR_INVALID; }
  const mzx_game* g = c->game;
  const mzx_move& m = c->move;
  if (m.num_games != g->num_games || m.action_space_size != g->num_actions || c->max_moves < 1 ||
      !m.h_in || !m.d_in || !m.h_out || !m.d_out || !m.io.d_noise) {
    set_error("mzx_actor_create: the move block does not fit the game (%d games x %d actions) or lacks a staging block / root noise",
              g->num_games, g->num_actions);
    return MZX_ERR_INVALID;
  }
  int rc = rng_check(c->bank, c->streams, g->num_games);
  if (rc) return rc;
  mzx_actor* a = new (std::nothrow) mzx_actor();
  if (!a) { set_error("out of host memory"); return MZX_ERR_RUNTIME; }
  a->game = c->game; a->search = c->search; a->bank = c->bank; a->d_arena = c->d_arena; a->arena_bytes = c->arena_bytes;
  a->move = c->move;
  a->move.add_exploration_noise = 1;
  a->B = g->num_games; a->A = g->num_actions; a->E = g->obs_elems(); a->max_moves = c->max_moves; a->first_slot = c->first_slot;
  const size_t B = (size_t)a->B, A = (size_t)a->A;
  a->streams.assign(c->streams, c->streams + B);
  a->legal.resize(B * A); a->to_play.resize(B); a->n_legal.resize(B); a->words.resize(B);
  a->cur_obs.resize(B * (size_t)a->E); a->next_obs.resize(B * (size_t)a->E);
  a->actions.resize(B); a->start.assign(B, 0);
  a->temps.assign(B, c->temperature); a->move_temps.resize(B); a->reward.resize(B); a->done.resize(B);
  a->game->reset(nullptr, a->B);
  a->game->observe(0, a->B, a->cur_obs.data());
  actor_refresh(a);
  *out = a;
  return MZX_OK;
}

void mzx_actor_destroy(mzx_actor* a) {
  if (!a) return;
  if (a->pending) (void)event_wait(a->event);
  event_destroy(a->event);
  event_destroy(a->start_event);
  stream_destroy(a->own_stream);
  delete a;
}

int mzx_selfplay_rounds(mzx_actor* const* groups, int32_t num_groups, mzx_rounds* io, void* stream) {
  if (!groups || num_groups < 1 || !io) { set_error("mzx_selfplay_rounds: missing argument"); return MZX_ERR_INVALID; }
  for (int k = 0; k < num_groups; ++k)
    if (!groups[k]) { set_error("mzx_selfplay_rounds: null group"); return MZX_ERR_INVALID; }
  const double t_call = actor_now();
  for (double& x : io->phase_seconds) x = 0.0;
  RoundsArgs args;
  args.phase = io->phase_seconds;
  args.temperature = io->temperature; args.temperature_threshold = io->temperature_threshold;
  args.pow_table = io->pow_table; args.table_stride = io->table_stride; args.table_temperatures = io->table_temperatures;
  args.num_temperatures = io->num_temperatures; args.retry = io->retry; args.retry_ctx = io->retry_ctx;
  int64_t finished = 0, rounds = 0, searches = 0, sequence = io->sequence;
  double search_seconds = 0.0;
  int rc = MZX_OK;
  const bool pipelined = num_groups > 1;
  // Streams: the first group searches on the caller's stream, every further group on a stream of its own, ordered behind
  // whatever the caller queued before this call (a weight upload).  Half a shard of a small network fills half the chip
  // (C2: 2048 trees = two waves per CU), and such a search is latency-bound -- it takes as long as the whole shard's -- so
  // the two groups' searches must run SIDE BY SIDE, not one behind the other, for the pipelining to pay.  Every search is
  // waited for on the host before its results are read and nothing is in flight when the call returns.
  std::vector<void*> streams((size_t)num_groups, stream);
  for (int k = 1; k < num_groups && rc == MZX_OK; ++k) {
    mzx_actor* a = groups[k];
    if (tune(TUNE_ROUNDS_STREAMS) == 0) break;
    if (!a->own_stream && stream_create(&a->own_stream) != 0) { a->own_stream = nullptr; continue; }
    if (!a->own_stream) continue;          // (a build without streams: the caller's)
    if (event_record(&a->start_event, (stream_t)stream) != 0 || stream_wait_event((stream_t)a->own_stream, a->start_event) != 0) {
      set_error("mzx_selfplay_rounds: ordering the group's stream behind the caller's failed");
      rc = MZX_ERR_RUNTIME;
      break;
    }
    streams[(size_t)k] = a->own_stream;
  }
  // SelfPlay._rounds_batched, statement for statement
  while (rc == MZX_OK && finished < io->min_games && (io->max_rounds < 0 || rounds < io->max_rounds)) {
    for (int k = 0; k < num_groups && rc == MZX_OK; ++k)
      if (!groups[k]->pending) rc = actor_begin(groups[k], streams[(size_t)k], &search_seconds, io->phase_seconds);
    for (int k = 0; k < num_groups && rc == MZX_OK; ++k) {
      rc = actor_consume(groups[k], k, args, &sequence, &finished, &search_seconds);
      if (rc) break;
      searches += groups[k]->B;
      int64_t later = 0;
      for (int j = k + 1; j < num_groups; ++j) later += groups[j]->B;
      // certain to be consumed by this call: runs while the host plays the groups after it
      if (pipelined && k + 1 < num_groups && finished + later < io->min_games && (io->max_rounds < 0 || rounds + 1 < io->max_rounds))
        rc = actor_begin(groups[k], streams[(size_t)k], &search_seconds, io->phase_seconds);
    }
    ++rounds;
  }
  if (rc != MZX_OK) {          // nothing stays in flight behind a failure
    for (int k = 0; k < num_groups; ++k) (void)stream_sync((stream_t)streams[(size_t)k]);
    for (int k = 0; k < num_groups; ++k) groups[k]->pending = false;
  }
  io->phase_seconds[5] = (actor_now() - t_call) - (io->phase_seconds[0] + io->phase_seconds[1] + io->phase_seconds[2] +
                                                   io->phase_seconds[3] + io->phase_seconds[4]);
  io->rounds = rounds; io->games = finished; io->searches = searches; io->search_seconds = search_seconds; io->sequence = sequence;
  return rc;
}

int mzx_actor_finished(mzx_actor* a, int64_t before_sequence, int64_t out[3]) {
  if (!a || !out) { set_error("mzx_actor_finished: null"); return MZX_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(a->finished_lock);
  out[0] = out[1] = out[2] = 0;
  for (const mzx_actor::Batch& b : a->finished) {
    if (before_sequence >= 0 && !b.seq.empty() && b.seq[0] >= before_sequence) break;
    out[0] += (int64_t)b.slot.size(); out[1] += (int64_t)b.val.size();
    if (!b.mask.empty()) out[2] = 1;
  }
  return MZX_OK;
}

int mzx_actor_take(mzx_actor* a, int64_t before_sequence, int32_t* slot, int32_t* length, int64_t* sequence, float* observations,
                   int64_t* actions, double* rewards, int64_t* to_play, int32_t* visit_counts, double* root_values,
                   uint8_t* legal_mask, int32_t* any_illegal) {
  if (!a || !slot || !length || !sequence || !observations || !actions || !rewards || !to_play || !visit_counts || !root_values) {
    set_error("mzx_actor_take: missing buffer");
    return MZX_ERR_INVALID;
  }
  // the batches to hand out leave the queue under its lock (a rounds call on another thread may be appending); the copies
  // run outside it
  std::deque<mzx_actor::Batch> mine;
  {
    std::lock_guard<std::mutex> lk(a->finished_lock);
    while (!a->finished.empty()) {
      mzx_actor::Batch& b = a->finished.front();
      if (before_sequence >= 0 && !b.seq.empty() && b.seq[0] >= before_sequence) break;
      if (!b.mask.empty() && !legal_mask) { set_error("mzx_actor_take: games with restricted legal sets need the legal_mask buffer"); return MZX_ERR_INVALID; }
      mine.push_back(std::move(b));
      a->finished.pop_front();
    }
  }
  const size_t A = (size_t)a->A, E = (size_t)a->E;
  size_t g = 0, q1 = 0, q0 = 0;
  int32_t illegal = 0;
  for (const mzx_actor::Batch& b : mine) {
    const size_t k = b.slot.size(), n1 = b.act.size(), n0 = b.val.size();
    memcpy(slot + g, b.slot.data(), sizeof(int32_t) * k);
    memcpy(length + g, b.n.data(), sizeof(int32_t) * k);
    memcpy(sequence + g, b.seq.data(), sizeof(int64_t) * k);
    memcpy(observations + q1 * E, b.obs.data(), sizeof(float) * n1 * E);
    memcpy(actions + q1, b.act.data(), sizeof(int64_t) * n1);
    memcpy(rewards + q1, b.rew.data(), sizeof(double) * n1);
    memcpy(to_play + q1, b.tp.data(), sizeof(int64_t) * n1);
    memcpy(visit_counts + q0 * A, b.vis.data(), sizeof(int32_t) * n0 * A);
    memcpy(root_values + q0, b.val.data(), sizeof(double) * n0);
    if (legal_mask) {
      if (b.mask.empty()) memset(legal_mask + q0 * A, 1, n0 * A);
      else { memcpy(legal_mask + q0 * A, b.mask.data(), n0 * A); illegal = 1; }
    }
    g += k; q1 += n1; q0 += n0;
  }
  if (any_illegal) *any_illegal = illegal;
  return MZX_OK;
}

}  // extern "C"
