// mzx_ops.h -- element functors of the GENERIC (any-config) path.
//
// Every functor describes one data-parallel operator as "element i of size()"
// with no cross-thread communication, so one source serves the HIP kernel
// template (mzx_launch.h) and the serial test build (tests/hostcheck).  The
// tuned gfx950 kernels (LDS-resident fused search, MFMA convolutions) live in
// their own .hip files and are verified against these.
//
// Network semantics follow /root/reference/models.py in eval mode:
//   mlp :630-642 (Linear + ELU)            ResidualBlock :213-229
//   FC representation/dynamics :128-169     DownSample :233-275 (AvgPool2d 3/2/1,
//   ResNet representation/dynamics :522-599   count_include_pad=True)
//   conv1x1 heads :369-389, :404-433
#pragma once
#include "mzx_platform.h"
#include "mzx_tree.h"

namespace mzx {

// y[b][o] = act(bias[o] + sum_i x[b][i] * W[o][i]  (+ W[o][in + action[b]]))
// The optional action term is the one-hot concat of the FC dynamics input
// (models.py:148-156): one-hot times W picks a column.
struct LinearOp {
  const float* x;
  const float* W;
  const float* bias;
  float* y;
  const int32_t* action;  // nullable
  int64_t x_stride, y_stride;
  int32_t batch, in_features, out_features, w_stride, elu;

  MZX_HD size_t size() const { return (size_t)batch * out_features; }
  MZX_HD void operator()(size_t i) const {
    const int b = (int)(i / out_features), o = (int)(i % out_features);
    const float* xr = x + (int64_t)b * x_stride;
    const float* wr = W + (int64_t)o * w_stride;
    float acc = 0.f;
    for (int k = 0; k < in_features; ++k) acc = fmaf(xr[k], wr[k], acc);
    if (action) acc += wr[in_features + action[b]];
    acc += bias[o];
    y[(int64_t)b * y_stride + o] = elu ? mzx_elu(acc) : acc;
  }
};

// 3x3 convolution, padding 1, stride 1|2, bias-free, NCHW fp32, with the
// eval-mode BatchNorm folded to y = conv * alpha[co] + beta[co] (alpha = null:
// no BN), optional residual add and ReLU.  If `action` is set the LAST input
// channel is virtual: the constant plane action[b] / num_actions of the ResNet
// dynamics input (models.py:557-572).
struct Conv3x3Op {
  const float* x;
  const float* W;      // [Cout][Cin][3][3]
  const float* alpha;  // nullable
  const float* beta;
  const float* res;    // nullable, same shape as y
  float* y;
  const int32_t* action;  // nullable
  int32_t batch, cin, cout, hin, win, hout, wout, stride, relu, num_actions;

  MZX_HD size_t size() const { return (size_t)batch * cout * hout * wout; }
  MZX_HD void operator()(size_t i) const {
    const int ox = (int)(i % wout);
    const int oy = (int)((i / wout) % hout);
    const int co = (int)((i / ((size_t)wout * hout)) % cout);
    const int b = (int)(i / ((size_t)wout * hout * cout));
    const int creal = action ? cin - 1 : cin;
    const float* xb = x + (int64_t)b * creal * hin * win;
    const float* wc = W + (int64_t)co * cin * 9;
    float acc = 0.f;
    for (int ci = 0; ci < cin; ++ci) {
      const bool virt = (ci >= creal);
      const float plane = virt ? ((float)action[b] / (float)num_actions) : 0.f;
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride + ky - 1;
        if (iy < 0 || iy >= hin) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = ox * stride + kx - 1;
          if (ix < 0 || ix >= win) continue;
          const float v = virt ? plane : xb[((int64_t)ci * hin + iy) * win + ix];
          acc = fmaf(v, wc[ci * 9 + ky * 3 + kx], acc);
        }
      }
    }
    if (alpha) acc = acc * alpha[co] + beta[co];
    if (res) acc += res[i];
    if (relu) acc = fmaxf(acc, 0.f);
    y[i] = acc;
  }
};

// AvgPool2d(kernel 3, stride 2, padding 1), count_include_pad=True: always / 9.
struct AvgPoolOp {
  const float* x;
  float* y;
  int32_t planes, hin, win, hout, wout;  // planes = batch * channels

  MZX_HD size_t size() const { return (size_t)planes * hout * wout; }
  MZX_HD void operator()(size_t i) const {
    const int ox = (int)(i % wout), oy = (int)((i / wout) % hout);
    const int64_t pl = (int64_t)(i / ((size_t)wout * hout));
    const float* xp = x + pl * hin * win;
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 + ky - 1;
      if (iy < 0 || iy >= hin) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 + kx - 1;
        if (ix < 0 || ix >= win) continue;
        acc += xp[iy * win + ix];
      }
    }
    y[i] = acc / 9.0f;
  }
};

// Square convolution with bias and optional ReLU, NCHW fp32: torch.nn.Conv2d(cin, cout, K, stride, padding)
// of DownsampleCNN.features (models.py:281-290: K = 2 * ceil(H / 16), stride 4, padding 2, then K = 5,
// stride 1, padding 2).
struct ConvKxKOp {
  const float* x;
  const float* W;     // [cout][cin][K][K]
  const float* bias;  // [cout]
  float* y;
  int32_t batch, cin, cout, hin, win, hout, wout, ksize, stride, pad, relu;

  MZX_HD size_t size() const { return (size_t)batch * cout * hout * wout; }
  MZX_HD void operator()(size_t i) const {
    const int ox = (int)(i % wout), oy = (int)((i / wout) % hout);
    const int co = (int)((i / ((size_t)wout * hout)) % cout);
    const int64_t b = (int64_t)(i / ((size_t)wout * hout * cout));
    const float* xb = x + b * cin * hin * win;
    const float* wc = W + (int64_t)co * cin * ksize * ksize;
    float acc = 0.f;
    for (int c = 0; c < cin; ++c) {
      for (int ky = 0; ky < ksize; ++ky) {
        const int iy = oy * stride + ky - pad;
        if (iy < 0 || iy >= hin) continue;
        for (int kx = 0; kx < ksize; ++kx) {
          const int ix = ox * stride + kx - pad;
          if (ix < 0 || ix >= win) continue;
          acc = fmaf(xb[((int64_t)c * hin + iy) * win + ix], wc[(c * ksize + ky) * ksize + kx], acc);
        }
      }
    }
    acc += bias[co];
    y[i] = relu ? fmaxf(acc, 0.f) : acc;
  }
};

// torch.nn.MaxPool2d(kernel_size=3, stride=2): no padding, floor mode (models.py:286, :289).
struct MaxPoolOp {
  const float* x;
  float* y;
  int32_t planes, hin, win, hout, wout;

  MZX_HD size_t size() const { return (size_t)planes * hout * wout; }
  MZX_HD void operator()(size_t i) const {
    const int ox = (int)(i % wout), oy = (int)((i / wout) % hout);
    const float* xp = x + (int64_t)(i / ((size_t)wout * hout)) * hin * win;
    float m = xp[(oy * 2) * win + ox * 2];
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) m = fmaxf(m, xp[(oy * 2 + ky) * win + ox * 2 + kx]);
    y[i] = m;
  }
};

// torch.nn.AdaptiveAvgPool2d((hout, wout)) (models.py:291): window [floor(o * in / out), ceil((o + 1) * in / out)).
struct AdaptiveAvgPoolOp {
  const float* x;
  float* y;
  int32_t planes, hin, win, hout, wout;

  MZX_HD size_t size() const { return (size_t)planes * hout * wout; }
  MZX_HD void operator()(size_t i) const {
    const int ox = (int)(i % wout), oy = (int)((i / wout) % hout);
    const float* xp = x + (int64_t)(i / ((size_t)wout * hout)) * hin * win;
    const int y0 = (oy * hin) / hout, y1 = ((oy + 1) * hin + hout - 1) / hout;
    const int x0 = (ox * win) / wout, x1 = ((ox + 1) * win + wout - 1) / wout;
    float acc = 0.f;
    for (int iy = y0; iy < y1; ++iy)
      for (int ix = x0; ix < x1; ++ix) acc += xp[iy * win + ix];
    y[i] = acc / (float)((y1 - y0) * (x1 - x0));
  }
};

// 1x1 convolution with bias: y[b][r][p] = bias[r] + sum_c x[b][c][p] * W[r][c]
struct Conv1x1Op {
  const float* x;
  const float* W;
  const float* bias;
  float* y;
  int32_t batch, cin, cout, hw;

  MZX_HD size_t size() const { return (size_t)batch * cout * hw; }
  MZX_HD void operator()(size_t i) const {
    const int p = (int)(i % hw), r = (int)((i / hw) % cout), b = (int)(i / ((size_t)hw * cout));
    const float* xb = x + (int64_t)b * cin * hw + p;
    const float* wr = W + (int64_t)r * cin;
    float acc = 0.f;
    for (int c = 0; c < cin; ++c) acc = fmaf(xb[(int64_t)c * hw], wr[c], acc);
    y[i] = acc + bias[r];
  }
};

// Min-max scaling of each group of `len` consecutive floats to [0, 1]
// (models.py:136-145 rows of the FC state; :527-553 per-channel planes of the
// ResNet state).  A scale below 1e-5 gets 1e-5 ADDED (not clamped).
struct MinMaxScaleOp {
  const float* x;
  float* y;
  int32_t groups, len;

  MZX_HD size_t size() const { return (size_t)groups * len; }
  MZX_HD void operator()(size_t i) const {
    const int64_t g = (int64_t)(i / len);
    const float* xg = x + g * len;
    float lo = xg[0], hi = xg[0];
    for (int k = 1; k < len; ++k) { lo = fminf(lo, xg[k]); hi = fmaxf(hi, xg[k]); }
    float scale = hi - lo;
    if (scale < 1e-5f) scale += 1e-5f;
    y[i] = mzx_div(x[i] - lo, scale);
  }
};

// Eval-mode BatchNorm folding, ATen's own form: alpha = weight / sqrt(var + eps),
// beta = bias - mean * alpha  (eps = 1e-5).
struct BnFoldOp {
  const float* weight;
  const float* bias;
  const float* mean;
  const float* var;
  float* alpha;
  float* beta;
  int32_t channels;

  MZX_HD size_t size() const { return (size_t)channels; }
  MZX_HD void operator()(size_t i) const {
    const float invstd = 1.0f / sqrtf(var[i] + 1e-5f);
    const float a = invstd * weight[i];
    alpha[i] = a;
    beta[i] = bias[i] - mean[i] * a;
  }
};

// ---------------------------------------------------------------------------
// Tree operators (one element = one tree).

struct TreeArena {
  char* trees;      // num_trees * L.tree_bytes
  float* hidden;    // [num_trees][num_nodes][hidden_size]
  TreeLayout L;
  MZX_HD TreeRef tree(int b) const { TreeRef t; t.base = trees + (int64_t)b * L.tree_bytes; t.L = L; return t; }
};

// Root expansion from the heads of initial_inference (self_play.py:286-314).
struct RootInitOp {
  TreeArena arena;
  SearchParams p;
  const float* value_logits;   // [B][F]   (null in lock-step mode)
  const float* policy_logits;  // [B][A]
  const double* ext_priors;    // [B][A] slot order, lock-step mode (null otherwise)
  const double* ext_root_reward;  // [B] lock-step mode, nullable
  const int32_t* legal;        // [B][A] legal action list, padded with -1
  const int32_t* to_play;      // [B]
  const double* noise;         // [B][A] slot order, nullable
  double* root_predicted_value;  // [B] out (nullable)

  MZX_HD size_t size() const { return (size_t)p.num_trees; }
  MZX_HD void operator()(size_t i) const {
    const int b = (int)i, A = p.num_actions;
    const TreeRef t = arena.tree(b);
    const int32_t* lg = legal + (int64_t)b * A;
    const double* nz = noise ? noise + (int64_t)b * A : nullptr;
    // reward head of initial_inference = log(one-hot at 0) -> decodes to -0.0f (see mzx_tree.h)
    const double r0 = ext_root_reward ? ext_root_reward[b] : (double)support_inverse_transform(0.0f);
    if (ext_priors) {
      const double* ep = ext_priors + (int64_t)b * A;
      tree_init_root(t, p, lg, to_play[b], r0, [&](int s) { return ep[s]; }, nz);
    } else {
      const float* pl = policy_logits + (int64_t)b * A;
      int nroot = 0;
      while (nroot < A && lg[nroot] >= 0) ++nroot;
      const SoftmaxStats st = softmax_stats(nroot, [&](int s) { return pl[lg[s]]; });
      tree_init_root(t, p, lg, to_play[b], r0,
                     [&](int s) { return (double)mzx_div(mzx_expf(pl[lg[s]] - st.m), st.den); }, nz);
      if (root_predicted_value)
        root_predicted_value[b] = (double)support_to_scalar(value_logits + (int64_t)b * (2 * p.support_size + 1),
                                                            p.support_size);
    }
  }
};

// Selection walk; emits what the network needs for the new leaf.
struct SelectOp {
  TreeArena arena;
  SearchParams p;
  const uint32_t* tape;   // [B][tape_words]
  int32_t* sel_parent;    // [B] node whose hidden state feeds recurrent_inference
  int32_t* sel_action;    // [B] action taken from it
  int32_t* sel_leaf;      // [B] canonical index of the new node

  MZX_HD size_t size() const { return (size_t)p.num_trees; }
  MZX_HD void operator()(size_t i) const {
    const int b = (int)i;
    const TreeRef t = arena.tree(b);
    const int parent = tree_select(t, p, tape + (int64_t)b * p.tape_words);
    const int slot = t.meta(TM_CUR_SLOT);
    sel_parent[b] = parent;
    sel_action[b] = (parent == 0) ? t.root_action(slot) : slot;
    sel_leaf[b] = t.meta(TM_CUR_LEAF);
  }
};

// Decode the heads of recurrent_inference, expand the leaf, back-propagate
// (self_play.py:343-353).
struct ExpandBackpropOp {
  TreeArena arena;
  SearchParams p;
  const float* value_logits;   // [B][F]
  const float* reward_logits;  // [B][F]
  const float* policy_logits;  // [B][A]
  const double* ext_value;     // lock-step mode: [B], [B], [B][A]
  const double* ext_reward;
  const double* ext_priors;

  MZX_HD size_t size() const { return (size_t)p.num_trees; }
  MZX_HD void operator()(size_t i) const {
    const int b = (int)i, A = p.num_actions, F = 2 * p.support_size + 1;
    const TreeRef t = arena.tree(b);
    if (ext_priors) {
      const double* ep = ext_priors + (int64_t)b * A;
      tree_expand_backprop(t, p, ext_value[b], ext_reward[b], [&](int s) { return ep[s]; });
    } else {
      const float* pl = policy_logits + (int64_t)b * A;
      const double v = (double)support_to_scalar(value_logits + (int64_t)b * F, p.support_size);
      const double r = (double)support_to_scalar(reward_logits + (int64_t)b * F, p.support_size);
      const SoftmaxStats st = softmax_stats(A, [&](int s) { return pl[s]; });
      tree_expand_backprop(t, p, v, r, [&](int s) { return (double)mzx_div(mzx_expf(pl[s] - st.m), st.den); });
    }
  }
};

// dense[b][:] = hidden[b][node[b]][:]   /   hidden[b][node[b]][:] = dense[b][:]
struct HiddenMoveOp {
  TreeArena arena;
  float* dense;            // [B][Hf]
  const int32_t* node;     // [B]; null => node 0
  int32_t num_trees, num_nodes, hidden_size, to_arena;

  MZX_HD size_t size() const { return (size_t)num_trees * hidden_size; }
  MZX_HD void operator()(size_t i) const {
    const int64_t b = (int64_t)(i / hidden_size), j = (int64_t)(i % hidden_size);
    const int n = node ? node[b] : 0;
    float* h = arena.hidden + (b * num_nodes + n) * hidden_size + j;
    if (to_arena) *h = dense[i]; else dense[i] = *h;
  }
};

// Search results per tree: visit count per ACTION (0 for illegal / unvisited),
// root value (Node.value of the root), max depth, status flags, statistics.
struct FinalizeOp {
  TreeArena arena;
  SearchParams p;
  int32_t* visit_counts;  // [B][A] by action
  double* root_value;     // [B]
  int32_t* info;          // [B][4]: max_depth, flags, tape words consumed, sum of leaf depths

  MZX_HD size_t size() const { return (size_t)p.num_trees; }
  MZX_HD void operator()(size_t i) const {
    const int b = (int)i, A = p.num_actions;
    const TreeRef t = arena.tree(b);
    for (int a = 0; a < A; ++a) visit_counts[(int64_t)b * A + a] = 0;
    const int nroot = t.meta(TM_ROOT_N);
    for (int s = 0; s < nroot; ++s) {
      const int c = t.child(0, s);
      visit_counts[(int64_t)b * A + t.root_action(s)] = (c >= 0) ? t.visit(c) : 0;
    }
    const int vc = t.visit(0);
    root_value[b] = (vc == 0) ? 0.0 : t.value_sum(0) / (double)vc;
    info[b * 4 + 0] = t.meta(TM_MAX_DEPTH);
    info[b * 4 + 1] = t.meta(TM_FLAGS);
    info[b * 4 + 2] = t.meta(TM_TAPE_POS);
    info[b * 4 + 3] = t.meta(TM_SUM_DEPTH);
  }
};

}  // namespace mzx
