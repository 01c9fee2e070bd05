// mzx_net.h -- host-side network descriptor: the reference's two inference
// networks expressed as a flat weight table + two operator programs
// (initial_inference, recurrent_inference).
//
// Mirrors the factory and module structure of /root/reference/models.py:
//   MuZeroNetwork.__new__ :7-41, MuZeroFullyConnectedNetwork :80-195,
//   RepresentationNetwork :300-349, DynamicsNetwork :352-389,
//   PredictionNetwork :392-433, MuZeroResidualNetwork :436-623, mlp :630-642.
// The weight table reproduces the reference state_dict key order (keys carry
// DataParallel's ".module." infix, models.py:98-126) so a checkpoint produced by
// the reference trainer binds without renaming.
#pragma once
#include <string>
#include <vector>

#include "../../include/mzx.h"
#include "mzx_launch.h"
#include "mzx_ops.h"
#include "mzx_resnet_plan.h"
#include "mzx_batched_plan.h"

namespace mzx {

void set_error(const char* fmt, ...);

enum BufId { BUF_IN = -1, BUF_HIDDEN = -2, BUF_VALUE = -3, BUF_REWARD = -4, BUF_POLICY = -5 };
enum OpKind { OP_LINEAR, OP_CONV3, OP_POOL, OP_CONV1, OP_SCALE, OP_CONVK, OP_MAXPOOL, OP_ADAPTIVE_POOL };

struct TensorInfo {
  std::string name;
  int64_t offset, numel;
  int32_t dims[4];
};

struct BnRef {           // offsets (floats) into the flat buffer / derived buffer
  int64_t weight = -1, bias, mean, var, alpha, beta;
  int32_t channels = 0;
};

struct OpDesc {
  int kind;
  int in, out, res;      // buffer ids (BufId or temp index >= 0); res = -100 if none
  int64_t w = -1, b = -1;  // flat offsets
  BnRef bn;
  int32_t cin = 0, cout = 0, hin = 0, win = 0, hout = 0, wout = 0, stride = 1;
  int32_t relu = 0, elu = 0, use_action = 0;
  int32_t in_features = 0, out_features = 0, w_stride = 0;
  int32_t groups_per_sample = 0, len = 0;  // OP_SCALE
  int32_t ksize = 0, pad = 0;               // OP_CONVK
};

struct LayerSpec { int32_t n; const int32_t* sizes; };

}  // namespace mzx

struct mzx_net {
  mzx_net_config cfg;
  std::vector<mzx::TensorInfo> tensors;
  std::vector<mzx::BnRef> bns;
  std::vector<mzx::OpDesc> prog_initial, prog_recurrent;
  int64_t num_params = 0, derived_floats = 0;
  int64_t hidden_size = 0, input_size = 0;
  int64_t act_floats = 0;  // per-sample capacity of one temp buffer
  int32_t n_temp = 0;
  int32_t full_support = 0;
  int32_t hc = 0, hh = 0, hw = 0;  // hidden state dims (resnet)
  const float* d_flat = nullptr;
  float* d_derived = nullptr;
  mzx::RzPlan rz;          // fused residual-network engine (mzx_resnet_fused.h)
  int32_t rz_mode = 1;     // 0: one kernel per operator, 1: fused engine where planned
  int32_t rz_waves = 0;    // 0: automatic, 4: force 256-thread workgroups (A/B measurements)
  int32_t rb_force = 0;    // 1: every program on the streamed engine (mzx_net_set_mode(3) / (4))
  int32_t rb_no_towers = 0;  // 1: the streamed engine runs layer by layer, no tower launches (mzx_net_set_mode(4): A/B)
  mzx::RbPlan rb;          // streamed MFMA engine for residual networks the fused engine cannot hold (mzx_resnet_batched.h)
};

namespace mzx {

class NetBuilder {
 public:
  explicit NetBuilder(mzx_net* n) : net(n) {}
  bool build();

 private:
  mzx_net* net;
  std::vector<OpDesc>* prog = nullptr;

  int64_t add_tensor(const std::string& name, int d0, int d1 = 0, int d2 = 0, int d3 = 0) {
    TensorInfo t;
    t.name = name;
    t.offset = net->num_params;
    t.dims[0] = d0; t.dims[1] = d1; t.dims[2] = d2; t.dims[3] = d3;
    t.numel = (int64_t)d0 * (d1 ? d1 : 1) * (d2 ? d2 : 1) * (d3 ? d3 : 1);
    net->num_params += t.numel;
    net->tensors.push_back(t);
    return t.offset;
  }
  BnRef add_bn(const std::string& prefix, int c) {
    BnRef r;
    r.channels = c;
    r.weight = add_tensor(prefix + ".weight", c);
    r.bias = add_tensor(prefix + ".bias", c);
    r.mean = add_tensor(prefix + ".running_mean", c);
    r.var = add_tensor(prefix + ".running_var", c);
    r.alpha = net->derived_floats;
    r.beta = net->derived_floats + c;
    net->derived_floats += 2 * c;
    net->bns.push_back(r);
    return r;
  }
  void note_act(int64_t per_sample) { if (per_sample > net->act_floats) net->act_floats = per_sample; }
  int fresh(std::initializer_list<int> busy) {
    for (int t = 0;; ++t) {
      bool used = false;
      for (int b : busy) used |= (b == t);
      if (!used) { if (t + 1 > net->n_temp) net->n_temp = t + 1; return t; }
    }
  }

  struct Mlp { std::vector<int64_t> w, b; std::vector<int32_t> sizes; };
  // mlp(input, layer_sizes, output): Sequential indices 0,2,4,.. are the Linear layers (models.py:630-642)
  Mlp declare_mlp(const std::string& prefix, int in, int n_hidden, const int32_t* hidden, int out) {
    Mlp m;
    m.sizes.push_back(in);
    for (int i = 0; i < n_hidden; ++i) m.sizes.push_back(hidden[i]);
    m.sizes.push_back(out);
    for (size_t i = 0; i + 1 < m.sizes.size(); ++i) {
      const std::string p = prefix + "." + std::to_string(2 * i);
      m.w.push_back(add_tensor(p + ".weight", m.sizes[i + 1], m.sizes[i]));
      m.b.push_back(add_tensor(p + ".bias", m.sizes[i + 1]));
    }
    return m;
  }
  // emit the Linear chain; `extra_onehot` = width of the one-hot action block of layer 0
  void emit_mlp(const Mlp& m, int in, int out, std::initializer_list<int> keep, int extra_onehot = 0) {
    int cur = in;
    const size_t L = m.w.size();
    for (size_t i = 0; i < L; ++i) {
      OpDesc d{};
      d.kind = OP_LINEAR;
      d.res = -100;
      d.in = cur;
      const bool last = (i + 1 == L);
      int dst;
      if (last) dst = out;
      else {
        // a temp distinct from the input and everything the caller still needs
        for (int t = 0;; ++t) {
          bool used = (t == cur);
          for (int k : keep) used |= (k == t);
          if (!used) { dst = t; if (t + 1 > net->n_temp) net->n_temp = t + 1; break; }
        }
      }
      d.out = dst;
      d.w = m.w[i];
      d.b = m.b[i];
      d.out_features = m.sizes[i + 1];
      d.w_stride = m.sizes[i];
      d.in_features = (i == 0) ? m.sizes[0] - extra_onehot : m.sizes[i];
      d.use_action = (i == 0 && extra_onehot > 0);
      d.elu = last ? 0 : 1;
      note_act(d.out_features);
      prog->push_back(d);
      cur = dst;
    }
  }

  struct Block { int64_t w1, w2; BnRef bn1, bn2; int c; };
  Block declare_block(const std::string& prefix, int c) {
    Block b;
    b.c = c;
    b.w1 = add_tensor(prefix + ".conv1.weight", c, c, 3, 3);
    b.bn1 = add_bn(prefix + ".bn1", c);
    b.w2 = add_tensor(prefix + ".conv2.weight", c, c, 3, 3);
    b.bn2 = add_bn(prefix + ".bn2", c);
    return b;
  }
  void emit_conv3(int in, int out, int res, int64_t w, const BnRef* bn, int cin, int cout, int hin, int win,
                  int stride, int relu, int use_action) {
    OpDesc d{};
    d.kind = OP_CONV3;
    d.in = in; d.out = out; d.res = res;
    d.w = w;
    if (bn) d.bn = *bn;
    d.cin = cin; d.cout = cout; d.hin = hin; d.win = win; d.stride = stride;
    d.hout = (hin + 2 - 3) / stride + 1;
    d.wout = (win + 2 - 3) / stride + 1;
    d.relu = relu;
    d.use_action = use_action;
    note_act((int64_t)cout * d.hout * d.wout);
    prog->push_back(d);
  }
  // ResidualBlock.forward (models.py:221-229); returns the buffer holding the result
  int emit_block(const Block& b, int cur, int h, int w, std::initializer_list<int> keep = {}) {
    int busy1[3] = {cur, -100, -100};
    int k = 1;
    for (int x : keep) if (k < 3) busy1[k++] = x;
    const int a = fresh({busy1[0], busy1[1], busy1[2]});
    emit_conv3(cur, a, -100, b.w1, &b.bn1, b.c, b.c, h, w, 1, 1, 0);
    const int o = fresh({busy1[0], busy1[1], busy1[2], a});
    emit_conv3(a, o, cur, b.w2, &b.bn2, b.c, b.c, h, w, 1, 1, 0);
    return o;
  }
  void emit_pool(int in, int out, int c, int hin, int win) {
    OpDesc d{};
    d.kind = OP_POOL;
    d.in = in; d.out = out; d.res = -100;
    d.cin = d.cout = c; d.hin = hin; d.win = win;
    d.hout = (hin + 2 - 3) / 2 + 1;
    d.wout = (win + 2 - 3) / 2 + 1;
    note_act((int64_t)c * d.hout * d.wout);
    prog->push_back(d);
  }
  // DownsampleCNN pieces (models.py:278-297)
  void emit_convk(int in, int out, int64_t w, int64_t b, int cin, int cout, int hin, int win, int ksize, int stride,
                  int pad) {
    OpDesc d{};
    d.kind = OP_CONVK;
    d.in = in; d.out = out; d.res = -100;
    d.w = w; d.b = b; d.cin = cin; d.cout = cout; d.hin = hin; d.win = win; d.ksize = ksize; d.stride = stride; d.pad = pad;
    d.hout = (hin + 2 * pad - ksize) / stride + 1;
    d.wout = (win + 2 * pad - ksize) / stride + 1;
    d.relu = 1;
    note_act((int64_t)cout * d.hout * d.wout);
    prog->push_back(d);
  }
  void emit_window_pool(int kind, int in, int out, int c, int hin, int win, int hout, int wout) {
    OpDesc d{};
    d.kind = kind;
    d.in = in; d.out = out; d.res = -100;
    d.cin = d.cout = c; d.hin = hin; d.win = win; d.hout = hout; d.wout = wout;
    note_act((int64_t)c * hout * wout);
    prog->push_back(d);
  }
  void emit_conv1(int in, int out, int64_t w, int64_t b, int cin, int cout, int hw) {
    OpDesc d{};
    d.kind = OP_CONV1;
    d.in = in; d.out = out; d.res = -100;
    d.w = w; d.b = b; d.cin = cin; d.cout = cout; d.hin = hw; d.win = 1;
    note_act((int64_t)cout * hw);
    prog->push_back(d);
  }
  void emit_scale(int in, int out, int groups_per_sample, int len) {
    OpDesc d{};
    d.kind = OP_SCALE;
    d.in = in; d.out = out; d.res = -100;
    d.groups_per_sample = groups_per_sample; d.len = len;
    prog->push_back(d);
  }

  bool build_fc();
  bool build_resnet();
};

inline bool NetBuilder::build_fc() {
  const mzx_net_config& c = net->cfg;
  const int A = c.action_space_size, E = c.encoding_size, F = 2 * c.support_size + 1;
  const int in = c.observation_shape[0] * c.observation_shape[1] * c.observation_shape[2] * (c.stacked_observations + 1) +
                 c.stacked_observations * c.observation_shape[1] * c.observation_shape[2];  // models.py:100-106
  net->input_size = in;
  net->hidden_size = E;
  net->hc = E; net->hh = 1; net->hw = 1;
  const Mlp rep = declare_mlp("representation_network.module", in, c.n_fc_representation_layers, c.fc_representation_layers, E);
  const Mlp dyn = declare_mlp("dynamics_encoded_state_network.module", E + A, c.n_fc_dynamics_layers, c.fc_dynamics_layers, E);
  const Mlp rew = declare_mlp("dynamics_reward_network.module", E, c.n_fc_reward_layers, c.fc_reward_layers, F);
  const Mlp pol = declare_mlp("prediction_policy_network.module", E, c.n_fc_policy_layers, c.fc_policy_layers, A);
  const Mlp val = declare_mlp("prediction_value_network.module", E, c.n_fc_value_layers, c.fc_value_layers, F);
  note_act(E); note_act(F); note_act(A);

  // initial_inference, models.py:172-190
  prog = &net->prog_initial;
  int t = fresh({});
  emit_mlp(rep, BUF_IN, t, {t});
  emit_scale(t, BUF_HIDDEN, 1, E);
  emit_mlp(pol, BUF_HIDDEN, BUF_POLICY, {});
  emit_mlp(val, BUF_HIDDEN, BUF_VALUE, {});

  // recurrent_inference, models.py:147-169, :192-195 (reward head reads the UNscaled next state)
  prog = &net->prog_recurrent;
  t = fresh({});
  emit_mlp(dyn, BUF_IN, t, {t}, A);
  emit_mlp(rew, t, BUF_REWARD, {t});
  emit_scale(t, BUF_HIDDEN, 1, E);
  emit_mlp(pol, BUF_HIDDEN, BUF_POLICY, {});
  emit_mlp(val, BUF_HIDDEN, BUF_VALUE, {});
  return true;
}

inline bool NetBuilder::build_resnet() {
  const mzx_net_config& c = net->cfg;
  const int A = c.action_space_size, F = 2 * c.support_size + 1, C = c.channels;
  const int cin = c.observation_shape[0] * (c.stacked_observations + 1) + c.stacked_observations;
  int H = c.observation_shape[1], W = c.observation_shape[2];
  net->input_size = (int64_t)cin * H * W;
  const std::string R = "representation_network.module";

  // ---- declare tensors in state_dict order ----
  int64_t ds_conv1 = -1, ds_conv2 = -1;
  std::vector<Block> ds1, ds2, ds3;
  // DownsampleCNN (models.py:278-297): features.0 / features.3 are the two biased convolutions
  int64_t cnn_w1 = -1, cnn_b1 = -1, cnn_w2 = -1, cnn_b2 = -1;
  const int cnn_mid = (cin + C) / 2, cnn_k = 2 * ((H + 15) / 16);
  if (c.downsample == 2) {
    cnn_w1 = add_tensor(R + ".downsample_net.features.0.weight", cnn_mid, cin, cnn_k, cnn_k);
    cnn_b1 = add_tensor(R + ".downsample_net.features.0.bias", cnn_mid);
    cnn_w2 = add_tensor(R + ".downsample_net.features.3.weight", C, cnn_mid, 5, 5);
    cnn_b2 = add_tensor(R + ".downsample_net.features.3.bias", C);
  }
  if (c.downsample == 1) {
    ds_conv1 = add_tensor(R + ".downsample_net.conv1.weight", C / 2, cin, 3, 3);
    for (int i = 0; i < 2; ++i) ds1.push_back(declare_block(R + ".downsample_net.resblocks1." + std::to_string(i), C / 2));
    ds_conv2 = add_tensor(R + ".downsample_net.conv2.weight", C, C / 2, 3, 3);
    for (int i = 0; i < 3; ++i) ds2.push_back(declare_block(R + ".downsample_net.resblocks2." + std::to_string(i), C));
    for (int i = 0; i < 3; ++i) ds3.push_back(declare_block(R + ".downsample_net.resblocks3." + std::to_string(i), C));
  }
  // conv/bn exist in the state_dict even when the downsample stem bypasses them (models.py:330-334)
  const int64_t rep_conv = add_tensor(R + ".conv.weight", C, cin, 3, 3);
  const BnRef rep_bn = add_bn(R + ".bn", C);
  std::vector<Block> rep_blocks, dyn_blocks, pred_blocks;
  for (int i = 0; i < c.blocks; ++i) rep_blocks.push_back(declare_block(R + ".resblocks." + std::to_string(i), C));

  // hidden-state geometry
  int h = H, w = W;
  int cnn_h[4] = {0, 0, 0, 0}, cnn_w[4] = {0, 0, 0, 0};   // after conv1, pool1, conv2, pool2
  if (c.downsample == 2) {
    cnn_h[0] = (H + 4 - cnn_k) / 4 + 1; cnn_w[0] = (W + 4 - cnn_k) / 4 + 1;
    if (H + 4 < cnn_k || W + 4 < cnn_k || cnn_h[0] < 3 || cnn_w[0] < 3) {
      set_error("downsample=\"CNN\": observation %dx%d is too small for its %dx%d stride-4 convolution", H, W, cnn_k, cnn_k);
      return false;
    }
    cnn_h[1] = (cnn_h[0] - 3) / 2 + 1; cnn_w[1] = (cnn_w[0] - 3) / 2 + 1;
    cnn_h[2] = cnn_h[1]; cnn_w[2] = cnn_w[1];                                   // 5x5, padding 2
    if (cnn_h[2] < 3 || cnn_w[2] < 3) { set_error("downsample=\"CNN\": observation %dx%d is too small for the second pooling", H, W); return false; }
    cnn_h[3] = (cnn_h[2] - 3) / 2 + 1; cnn_w[3] = (cnn_w[2] - 3) / 2 + 1;
    h = (H + 15) / 16; w = (W + 15) / 16;                                       // AdaptiveAvgPool2d target
  }
  if (c.downsample == 1) {
    h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;   // conv1 stride 2
    h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;   // conv2 stride 2
    h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;   // pooling1
    h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;   // pooling2
    // the reference sizes its heads with ceil(H/16) x ceil(W/16) (models.py:455-484)
    if (h != (H + 15) / 16 || w != (W + 15) / 16) {
      set_error("downsample geometry %dx%d -> %dx%d does not match the reference head size", H, W, h, w);
      return false;
    }
  }
  net->hc = C; net->hh = h; net->hw = w;
  net->hidden_size = (int64_t)C * h * w;

  const std::string D = "dynamics_network.module";
  const int64_t dyn_conv = add_tensor(D + ".conv.weight", C, C + 1, 3, 3);
  const BnRef dyn_bn = add_bn(D + ".bn", C);
  for (int i = 0; i < c.blocks; ++i) dyn_blocks.push_back(declare_block(D + ".resblocks." + std::to_string(i), C));
  const int Rr = c.reduced_channels_reward, Rv = c.reduced_channels_value, Rp = c.reduced_channels_policy;
  const int64_t rw = add_tensor(D + ".conv1x1_reward.weight", Rr, C, 1, 1);
  const int64_t rb = add_tensor(D + ".conv1x1_reward.bias", Rr);
  const Mlp rew = declare_mlp(D + ".fc", Rr * h * w, c.n_resnet_fc_reward_layers, c.resnet_fc_reward_layers, F);

  const std::string P = "prediction_network.module";
  for (int i = 0; i < c.blocks; ++i) pred_blocks.push_back(declare_block(P + ".resblocks." + std::to_string(i), C));
  const int64_t vw = add_tensor(P + ".conv1x1_value.weight", Rv, C, 1, 1);
  const int64_t vb = add_tensor(P + ".conv1x1_value.bias", Rv);
  const int64_t pw = add_tensor(P + ".conv1x1_policy.weight", Rp, C, 1, 1);
  const int64_t pb = add_tensor(P + ".conv1x1_policy.bias", Rp);
  const Mlp val = declare_mlp(P + ".fc_value", Rv * h * w, c.n_resnet_fc_value_layers, c.resnet_fc_value_layers, F);
  const Mlp pol = declare_mlp(P + ".fc_policy", Rp * h * w, c.n_resnet_fc_policy_layers, c.resnet_fc_policy_layers, A);
  note_act(net->hidden_size); note_act(F); note_act(A);

  // PredictionNetwork.forward (models.py:423-433) reading BUF_HIDDEN
  auto emit_prediction = [&]() {
    int cur = BUF_HIDDEN;
    for (const Block& b : pred_blocks) cur = emit_block(b, cur, h, w);
    const int v = fresh({cur});
    emit_conv1(cur, v, vw, vb, C, Rv, h * w);
    const int q = fresh({cur, v});
    emit_conv1(cur, q, pw, pb, C, Rp, h * w);
    emit_mlp(val, v, BUF_VALUE, {v, q});
    emit_mlp(pol, q, BUF_POLICY, {q});
  };

  // ---- initial_inference (models.py:522-553, :601-618) ----
  prog = &net->prog_initial;
  int cur = BUF_IN, ch = H, cw = W;
  if (c.downsample == 2) {  // DownsampleCNN.forward, models.py:293-296
    int t = fresh({});
    emit_convk(cur, t, cnn_w1, cnn_b1, cin, cnn_mid, H, W, cnn_k, 4, 2);
    cur = t; t = fresh({cur});
    emit_window_pool(OP_MAXPOOL, cur, t, cnn_mid, cnn_h[0], cnn_w[0], cnn_h[1], cnn_w[1]);
    cur = t; t = fresh({cur});
    emit_convk(cur, t, cnn_w2, cnn_b2, cnn_mid, C, cnn_h[1], cnn_w[1], 5, 1, 2);
    cur = t; t = fresh({cur});
    emit_window_pool(OP_MAXPOOL, cur, t, C, cnn_h[2], cnn_w[2], cnn_h[3], cnn_w[3]);
    cur = t; t = fresh({cur});
    emit_window_pool(OP_ADAPTIVE_POOL, cur, t, C, cnn_h[3], cnn_w[3], h, w);
    cur = t;
  } else if (c.downsample) {  // DownSample.forward, models.py:264-275
    int t = fresh({});
    emit_conv3(cur, t, -100, ds_conv1, nullptr, cin, C / 2, ch, cw, 2, 0, 0);
    cur = t; ch = (ch - 1) / 2 + 1; cw = (cw - 1) / 2 + 1;
    for (const Block& b : ds1) cur = emit_block(b, cur, ch, cw);
    t = fresh({cur});
    emit_conv3(cur, t, -100, ds_conv2, nullptr, C / 2, C, ch, cw, 2, 0, 0);
    cur = t; ch = (ch - 1) / 2 + 1; cw = (cw - 1) / 2 + 1;
    for (const Block& b : ds2) cur = emit_block(b, cur, ch, cw);
    t = fresh({cur});
    emit_pool(cur, t, C, ch, cw);
    cur = t; ch = (ch - 1) / 2 + 1; cw = (cw - 1) / 2 + 1;
    for (const Block& b : ds3) cur = emit_block(b, cur, ch, cw);
    t = fresh({cur});
    emit_pool(cur, t, C, ch, cw);
    cur = t; ch = (ch - 1) / 2 + 1; cw = (cw - 1) / 2 + 1;
  } else {
    const int t = fresh({});
    emit_conv3(cur, t, -100, rep_conv, &rep_bn, cin, C, ch, cw, 1, 1, 0);
    cur = t;
  }
  for (const Block& b : rep_blocks) cur = emit_block(b, cur, h, w);
  emit_scale(cur, BUF_HIDDEN, C, h * w);
  emit_prediction();

  // ---- recurrent_inference (models.py:555-599, :620-623) ----
  prog = &net->prog_recurrent;
  {
    int t = fresh({});
    emit_conv3(BUF_IN, t, -100, dyn_conv, &dyn_bn, C + 1, C, h, w, 1, 1, 1);
    cur = t;
    for (const Block& b : dyn_blocks) cur = emit_block(b, cur, h, w);
    // the reward head reads the UNSCALED state (models.py:574-599); the scaling is emitted first so that the
    // reward chain, which nothing else waits for, can run beside the prediction trunk (rz_schedule)
    emit_scale(cur, BUF_HIDDEN, C, h * w);
    const int r = fresh({cur});
    emit_conv1(cur, r, rw, rb, C, Rr, h * w);
    emit_mlp(rew, r, BUF_REWARD, {cur, r});
    emit_prediction();
  }
  return true;
}

inline bool NetBuilder::build() {
  const mzx_net_config& c = net->cfg;
  net->full_support = 2 * c.support_size + 1;
  if (c.action_space_size < 1 || c.support_size < 0) { set_error("invalid action_space_size/support_size"); return false; }
  for (int i = 0; i < 3; ++i)
    if (c.observation_shape[i] < 1) { set_error("observation_shape must be 3 positive ints"); return false; }
  if (c.network == 0) return build_fc();
  if (c.network == 1) {
    if (c.downsample < 0 || c.downsample > 2) {
      set_error("downsample should be \"resnet\" or \"CNN\".");   // models.py:327
      return false;
    }
    if (c.blocks < 0 || c.channels < 1) { set_error("invalid blocks/channels"); return false; }
    return build_resnet();
  }
  // models.py:38-41
  set_error("The network parameter should be \"fullyconnected\" or \"resnet\".");
  return false;
}

// ---------------------------------------------------------------------------
// Program execution

struct NetBuffers {
  const float* in;       // observation or hidden state
  const int32_t* action; // recurrent only
  float* hidden;
  float* value;
  float* reward;
  float* policy;
  float* workspace;
};

// Workspace floats per sample: the temporaries of the operator program, then the streamed engine's private head-input
// region (mzx_batched_plan.h: RbHeads).  A workspace for `batch` samples is [temporaries x batch][head region x batch].
inline int64_t net_ws_per_sample(const mzx_net* net) { return net->act_floats * net->n_temp + net->rb.head_floats; }

inline float* resolve(const mzx_net* net, const NetBuffers& nb, int id, int batch) {
  switch (id) {
    case BUF_IN: return const_cast<float*>(nb.in);
    case BUF_HIDDEN: return nb.hidden;
    case BUF_VALUE: return nb.value;
    case BUF_REWARD: return nb.reward;
    case BUF_POLICY: return nb.policy;
    default: return nb.workspace + (int64_t)id * net->act_floats * batch;
  }
}

inline int run_program(const mzx_net* net, const std::vector<OpDesc>& prog, const NetBuffers& nb, int batch,
                       stream_t stream) {
  const float* flat = net->d_flat;
  const float* der = net->d_derived;
  for (const OpDesc& d : prog) {
    float* in = resolve(net, nb, d.in, batch);
    float* out = resolve(net, nb, d.out, batch);
    int rc = 0;
    switch (d.kind) {
      case OP_LINEAR: {
        LinearOp op;
        op.x = in; op.W = flat + d.w; op.bias = flat + d.b; op.y = out;
        op.action = d.use_action ? nb.action : nullptr;
        op.x_stride = d.in_features; op.y_stride = d.out_features;
        op.batch = batch; op.in_features = d.in_features; op.out_features = d.out_features;
        op.w_stride = d.w_stride; op.elu = d.elu;
        rc = launch<256>(op, stream);
        break;
      }
      case OP_CONV3: {
        Conv3x3Op op;
        op.x = in; op.W = flat + d.w;
        op.alpha = d.bn.channels ? der + d.bn.alpha : nullptr;
        op.beta = d.bn.channels ? der + d.bn.beta : nullptr;
        op.res = (d.res == -100) ? nullptr : resolve(net, nb, d.res, batch);
        op.y = out;
        op.action = d.use_action ? nb.action : nullptr;
        op.batch = batch; op.cin = d.cin; op.cout = d.cout; op.hin = d.hin; op.win = d.win;
        op.hout = d.hout; op.wout = d.wout; op.stride = d.stride; op.relu = d.relu;
        op.num_actions = net->cfg.action_space_size;
        rc = launch<256>(op, stream);
        break;
      }
      case OP_POOL: {
        AvgPoolOp op;
        op.x = in; op.y = out; op.planes = batch * d.cin; op.hin = d.hin; op.win = d.win; op.hout = d.hout; op.wout = d.wout;
        rc = launch<256>(op, stream);
        break;
      }
      case OP_CONV1: {
        Conv1x1Op op;
        op.x = in; op.W = flat + d.w; op.bias = flat + d.b; op.y = out;
        op.batch = batch; op.cin = d.cin; op.cout = d.cout; op.hw = d.hin;
        rc = launch<256>(op, stream);
        break;
      }
      case OP_CONVK: {
        ConvKxKOp op;
        op.x = in; op.W = flat + d.w; op.bias = flat + d.b; op.y = out;
        op.batch = batch; op.cin = d.cin; op.cout = d.cout; op.hin = d.hin; op.win = d.win; op.hout = d.hout;
        op.wout = d.wout; op.ksize = d.ksize; op.stride = d.stride; op.pad = d.pad; op.relu = d.relu;
        rc = launch<256>(op, stream);
        break;
      }
      case OP_MAXPOOL: {
        MaxPoolOp op;
        op.x = in; op.y = out; op.planes = batch * d.cin; op.hin = d.hin; op.win = d.win; op.hout = d.hout; op.wout = d.wout;
        rc = launch<256>(op, stream);
        break;
      }
      case OP_ADAPTIVE_POOL: {
        AdaptiveAvgPoolOp op;
        op.x = in; op.y = out; op.planes = batch * d.cin; op.hin = d.hin; op.win = d.win; op.hout = d.hout; op.wout = d.wout;
        rc = launch<256>(op, stream);
        break;
      }
      case OP_SCALE: {
        MinMaxScaleOp op;
        op.x = in; op.y = out; op.groups = batch * d.groups_per_sample; op.len = d.len;
        rc = launch<256>(op, stream);
        break;
      }
    }
    if (rc) { set_error("kernel launch failed: %s", runtime_error_string(rc)); return MZX_ERR_RUNTIME; }
  }
  return MZX_OK;
}

// log(one-hot at the support centre): -inf everywhere, 0 at the centre (models.py:176-183)
struct RewardFillOp {
  float* y;
  int32_t batch, full_support;
  MZX_HD size_t size() const { return (size_t)batch * full_support; }
  MZX_HD void operator()(size_t i) const {
    y[i] = ((int)(i % full_support) == full_support / 2) ? 0.0f : -(float)MZX_INF;
  }
};

}  // namespace mzx
