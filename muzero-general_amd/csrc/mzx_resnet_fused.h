// mzx_resnet_fused.h -- the residual MuZero network (models.py:206-623) as ONE gfx950
// kernel per inference: every convolution / 1x1 head convolution / head MLP layer of
// initial_inference or recurrent_inference is an FP32 MFMA GEMM whose A operand (the
// activations of a tile of trees) never leaves LDS between layers.
//
// Reference semantics (file:line relative to /root/reference), eval mode:
//   conv3x3 :206-209, ResidualBlock :213-229, RepresentationNetwork :300-349,
//   DynamicsNetwork :352-389 (reward head :369-389), PredictionNetwork :392-433,
//   MuZeroResidualNetwork.{representation,dynamics,prediction} :522-599 (per-plane
//   min-max scaling, action plane = action / |A|), mlp :630-642.
//
// Tiling (wave64, 256-thread workgroup = 4 waves, one workgroup per tile of T trees):
//   * activations: three rotating LDS slots in a position-major, zero-haloed layout
//     [T][(H+2)*(W+2)][Cs] (channels innermost, Cs = 8 mod 16 floats so that the 16-byte
//     reads of 16 consecutive rows fall on distinct LDS slots); head tensors are small
//     flat regions.  Halo positions and pad channels are written once (zero) and never
//     again, so a 3x3 tap is a constant address offset -- no bounds tests.
//   * every layer is D[M x N] = A[M x K] . B[K x N] on v_mfma_f32_16x16x4_f32:
//       3x3 conv : M = T*H*W (tree, position), K = 9 taps x Cin, N = Cout
//       1x1 conv : same rows, K = Cin
//       Linear   : M = T (trees), K = in_features, N = out_features
//     K is walked in 16-channel chunks: lane l = (row l&15, group g = l>>4) reads channels
//     4g..4g+3 of its row with ONE ds_read_b128 and uses them as the A operand of four
//     consecutive K-steps (K-step j = channels {4g + j}); the B fragments are pre-packed
//     in the same permuted order (RzPackOp), one 16-byte load per lane per chunk --
//     from an LDS-resident copy of the program's whole weight image when it fits beside the
//     activations (small nets: every weight is read from HBM/L2 once per workgroup), else
//     straight from L2 with the next chunk in flight.  Chunks are double-buffered in
//     registers (ping-pong, no copies); N tiles are spread over the waves first (each wave
//     then streams a disjoint part of the weights exactly once), M tiles next; a wave keeps
//     up to 8 accumulator tiles so one B fragment feeds up to 8 MFMAs.
//   * the dynamics input's action plane (models.py:557-572) is not materialised: it is
//     constant inside the board, so its contribution is action/|A| x (sum of the in-board
//     taps of its weights), a [Cout][H*W] table built per set_weights and added in the
//     epilogue -- K stays a multiple of 16 (Cin = C, not C + 1).
//   * epilogue per layer in registers: action term, folded BatchNorm (alpha, beta), bias,
//     residual (read from its LDS slot), ReLU / ELU, then the D fragment (col = lane & 15,
//     row = 4 * (lane >> 4) + r) goes back to LDS (16 lanes = 16 consecutive channels).
//   * exactness: the f32 MFMA is a k-ordered fmaf chain (bitwise); only the summation
//     ORDER differs from ATen's, i.e. fp32 round-off (tests: 1e-4 absolute on all heads).
//
// MFMA is used because these are genuine dense contractions (C3 231 kFLOP, C4 40.4 MFLOP,
// C5 1.5 MFLOP per simulation); roofline: FP32 matrix peak 157.3 TFLOP/s.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>

#include "mzx_net.h"
#include "mzx_pack.h"
#include "mzx_resnet_batched.h"

namespace mzx {

constexpr int RZ_LDS_BUDGET = 160 * 1024 - 1024;

// ---------------------------------------------------------------------------
// host planner: operator program (mzx_net.h) -> fused program

inline int rz_round4(int x) { return (x + 3) & ~3; }
// ints of one A-fragment offset table: nchunks entries + 8 prefetch entries, rounded to 16 bytes
MZX_HD inline int rz_aoff_ints(int nchunks) { return (nchunks + 8 + 3) & ~3; }
// entry of chunk c (clamped to the last chunk): K chunk -> (tap, 16-channel chunk) -> LDS float offset
MZX_HD inline int rz_aoff_entry(int c, int nchunks, int taps, int cchunks, int PW, int Cs) {
  const int cc = c < nchunks - 1 ? c : nchunks - 1;
  if (taps != 9) return cc * 16;
  const int tap = cc / cchunks, ch = cc - tap * cchunks;
  const int ty = tap / 3, tx = tap - 3 * ty;
  return (ty - 1) * PW * Cs + (tx - 1) * Cs + ch * 16;
}
inline int rz_round16(int x) { return (x + 15) & ~15; }

inline bool rz_fusable(const OpDesc& d, int h, int w) {
  switch (d.kind) {
    case OP_CONV3: return d.stride == 1 && d.hin == h && d.win == w;
    case OP_CONV1: return d.hin == h * w;
    case OP_LINEAR: return !d.use_action;
    case OP_SCALE: return d.len == h * w;
    default: return false;
  }
}

struct RzInst { int id, def, last, spatial, size, slot, off, tstride; };

// packs are deduplicated inside one program; every program owns a contiguous weight image
inline int32_t rz_add_pack(RzPlan& P, size_t first_pack, int64_t w_base, int64_t src, int taps, int cin, int cin_total,
                           int cout, int64_t& cursor) {
  for (size_t k = first_pack; k < P.packs.size(); ++k) {
    const RzPack& p = P.packs[k];
    if (p.src == src && p.taps == taps && p.cin == cin && p.cout == cout) return (int32_t)(p.dst - w_base);
  }
  RzPack p;
  p.src = src; p.taps = taps; p.cin = cin; p.cin_total = cin_total; p.cout = cout;
  p.cchunks = rz_round16(cin) / 16;
  p.nchunks = taps * p.cchunks;
  p.wchunks = (p.nchunks + 1) & ~1;
  p.ntiles = (cout + 15) / 16;
  p.dst = cursor;
  cursor += (int64_t)p.ntiles * p.wchunks * 256;
  P.packs.push_back(p);
  return (int32_t)(p.dst - w_base);
}

// Builds `R` from `prog`; returns false if the program cannot be fused.
// Slots.  Operators of a program that neither read nor write a region another one writes are independent
// (the heads after the trunk: reward / value / policy chains; the reward head beside the prediction trunk);
// list scheduling in program order puts each operator into the earliest slot after everything it conflicts
// with.  A slot costs one workgroup barrier and -- on the small networks, whose layers are latency-bound -- about
// as much time as its longest operator, so fewer slots is fewer microseconds per simulation.  Only small GEMMs
// share (at most three per slot); wide trunk convolutions and the scaling operator keep the whole workgroup.
// Region offsets still hold slot indices / flat offsets here (rz_finish_program runs later), which is exactly
// the identity the conflict test needs.
inline void rz_schedule(RzProgram& R, int HW) {
  const int n = R.n_ops;
  const int Tn = std::max(1, std::min(16, (64 + HW / 2) / HW));   // nominal trees per workgroup for work estimates
  struct Use { int kind[3], id[3], nr; int wkind, wid; };            // reads (in, res), write
  std::vector<Use> use(n);
  std::vector<int> tiles(n), work(n), slot(n, 0);
  std::vector<bool> shareable(n);
  for (int i = 0; i < n; ++i) {
    const RzOp& o = R.ops[i];
    Use& u = use[i];
    u.nr = 0;
    const bool in_spatial = !(o.kind == RZ_GEMM && o.rows == RZ_ROWS_TREE);
    u.kind[u.nr] = in_spatial; u.id[u.nr] = o.in_off; ++u.nr;
    if (o.res_off >= 0) { u.kind[u.nr] = 1; u.id[u.nr] = o.res_off; ++u.nr; }
    u.wkind = (o.out_layout == RZ_OUT_PADDED); u.wid = o.out_off;
    if (o.kind == RZ_GEMM) {
      const int nt = (o.cout + 15) / 16;
      tiles[i] = (o.rows == RZ_ROWS_POS ? (Tn * HW + 15) / 16 : 1) * nt;
      work[i] = tiles[i] * (o.nchunks + 2);
      shareable[i] = tiles[i] <= 6;
    } else {
      tiles[i] = 0; work[i] = 0; shareable[i] = false;
    }
  }
  auto conflict = [&](int x, int y) {
    for (int k = 0; k < use[x].nr; ++k) if (use[x].kind[k] == use[y].wkind && use[x].id[k] == use[y].wid) return true;
    for (int k = 0; k < use[y].nr; ++k) if (use[y].kind[k] == use[x].wkind && use[y].id[k] == use[x].wid) return true;
    return use[x].wkind == use[y].wkind && use[x].wid == use[y].wid;
  };
  std::vector<std::vector<int>> slots;
  for (int i = 0; i < n; ++i) {
    int earliest = 0;
    for (int j = 0; j < i; ++j) if (conflict(i, j)) earliest = std::max(earliest, slot[j] + 1);
    int s = (int)slots.size();
    if (shareable[i]) {
      for (int q = earliest; q < (int)slots.size(); ++q) {
        bool ok = slots[q].size() < 3;
        for (int j : slots[q]) ok = ok && shareable[j];
        if (ok) { s = q; break; }
      }
    }
    if (s == (int)slots.size()) slots.push_back({});
    slots[s].push_back(i);
    slot[i] = s;
  }
  // teams + table order
  RzOp sorted[RZ_MAX_OPS];
  int pos = 0;
  for (std::vector<int>& members : slots) {
    std::stable_sort(members.begin(), members.end(), [&](int x, int y) { return work[x] > work[y]; });
    const int m = (int)members.size();
    for (int k = 0; k < 2 && m > 1; ++k) {
      const int nw = k ? 8 : 4;
      int cnt[3] = {1, 1, 1};
      for (int left = nw - m; left > 0; --left) {   // spare waves to the operator with the most work per wave
        int best = -1;
        for (int q = 0; q < m; ++q) {
          const RzOp& o = R.ops[members[q]];
          const int cap = (o.rows == RZ_ROWS_TREE) ? (o.cout + 15) / 16 : tiles[members[q]];
          if (cnt[q] >= cap) continue;
          if (best < 0 || work[members[q]] * cnt[best] > work[members[best]] * cnt[q]) best = q;
        }
        if (best < 0) break;
        ++cnt[best];
      }
      int lo = 0;
      for (int q = 0; q < m; ++q) {
        RzOp& o = R.ops[members[q]];
        o.team |= ((uint32_t)lo | ((uint32_t)cnt[q] << 8)) << (16 * k);
        lo += cnt[q];
      }
    }
    for (int q = 0; q < m; ++q) {
      R.order[members[q]] = pos;
      sorted[pos] = R.ops[members[q]];
      if (q == m - 1) sorted[pos].sched |= 1u << 16;
      ++pos;
    }
  }
  memcpy(R.ops, sorted, sizeof(RzOp) * n);
  R.n_slots = (int32_t)slots.size();
}

inline bool rz_build_program(const mzx_net* net, const std::vector<OpDesc>& prog, RzPlan& P, RzProgram& R,
                             int64_t& cursor, int& max_c) {
  const int h = net->hh, w = net->hw, HW = h * w;
  const int n = (int)prog.size();
  int first = n;
  while (first > 0 && rz_fusable(prog[first - 1], h, w)) --first;
  if (first >= n) return false;
  // the fused part may read exactly one tensor it does not produce
  std::vector<RzInst> inst;
  std::map<int, int> cur;
  auto use = [&](int id, int at) -> int {
    auto it = cur.find(id);
    if (it == cur.end()) return -1;
    inst[it->second].last = at;
    return it->second;
  };
  std::vector<int> in_i(n, -1), res_i(n, -1), out_i(n, -1);
  int ext = -1;
  for (int i = first; i < n; ++i) {
    const OpDesc& d = prog[i];
    const bool in_spatial = (d.kind != OP_LINEAR);
    int a = use(d.in, i);
    if (a < 0) {
      if (ext >= 0 || !in_spatial) return false;
      RzInst e{d.in, first - 1, i, 1, 0, -1, 0, 0};
      e.size = (d.kind == OP_CONV3) ? d.cin - (d.use_action ? 1 : 0) : (d.kind == OP_SCALE ? d.groups_per_sample : d.cin);
      inst.push_back(e);
      ext = (int)inst.size() - 1;
      cur[d.in] = ext;
      a = ext;
    }
    in_i[i] = a;
    if (d.kind == OP_CONV3 && d.res != -100) {
      const int r = use(d.res, i);
      if (r < 0) return false;
      res_i[i] = r;
    }
    RzInst o{d.out, i, i, 0, 0, -1, 0, 0};
    switch (d.kind) {
      case OP_CONV3: o.spatial = 1; o.size = d.cout; break;
      case OP_SCALE: o.spatial = 1; o.size = d.groups_per_sample; break;
      case OP_CONV1: o.spatial = 0; o.size = d.cout * HW; break;
      default: o.spatial = 0; o.size = d.out_features; break;
    }
    inst.push_back(o);
    out_i[i] = (int)inst.size() - 1;
    cur[d.out] = out_i[i];
  }
  if (ext < 0) return false;
  // outputs stay live to the end
  for (RzInst& s : inst)
    if (s.id == BUF_VALUE || s.id == BUF_REWARD || s.id == BUF_POLICY) s.last = n;
  // slot / region allocation
  int flat_floats = 0;
  for (size_t k = 0; k < inst.size(); ++k) {
    RzInst& s = inst[k];
    if (s.spatial) {
      bool busy[3] = {false, false, false};
      for (size_t j = 0; j < inst.size(); ++j) {
        const RzInst& t = inst[j];
        if (j != k && t.spatial && t.slot >= 0 && t.def < s.def && t.last >= s.def) busy[t.slot] = true;
      }
      // among the free slots take the one whose previous tenant was read last the longest ago: a freshly
      // vacated slot would chain this operator behind the vacating reader (a false dependence that keeps
      // independent heads from sharing a barrier interval, rz_schedule)
      s.slot = -1;
      int best_last = 0;
      for (int q = 0; q < 3; ++q) {
        if (busy[q]) continue;
        int last_read = -1000;
        for (size_t j = 0; j < k; ++j)
          if (inst[j].spatial && inst[j].slot == q) last_read = std::max(last_read, inst[j].last);
        if (s.slot < 0 || last_read < best_last) { s.slot = q; best_last = last_read; }
      }
      if (s.slot < 0) return false;
    } else {
      s.tstride = rz_round16(s.size) + 8;   // whole 16-deep K chunks readable; 8 mod 16 spreads the rows over LDS slots
      s.off = flat_floats;
      flat_floats += s.tstride;
    }
  }
  if (n - first > RZ_MAX_OPS) return false;
  R.first = first;
  R.ext_buf = inst[ext].id;
  R.n_ops = n - first;
  R.in_channels = inst[ext].size;
  R.use_action = (prog[first].kind == OP_CONV3 && prog[first].use_action) ? 1 : 0;
  R.flat_floats = flat_floats;
  R.w_base = cursor;
  max_c = std::max(max_c, R.in_channels);
  const size_t first_pack = P.packs.size();
  // spatial region offsets hold the SLOT index until the geometry is known (rz_finish_program)
  for (int i = first; i < n; ++i) {
    const OpDesc& d = prog[i];
    RzOp& o = R.ops[i - first];
    memset(&o, 0, sizeof(o));
    o.res_off = -1; o.alpha_off = -1; o.beta_off = -1; o.bias_off = -1; o.asum_off = -1;
    const RzInst& si = inst[in_i[i]];
    const RzInst& so = inst[out_i[i]];
    switch (d.kind) {
      case OP_CONV3: {
        const int cin = d.cin - (d.use_action ? 1 : 0);   // the action plane is folded into the epilogue
        o.kind = RZ_GEMM; o.rows = RZ_ROWS_POS; o.taps = 9;
        o.in_off = si.slot; o.out_off = so.slot; o.out_layout = RZ_OUT_PADDED;
        if (res_i[i] >= 0) o.res_off = inst[res_i[i]].slot;
        o.cchunks = rz_round16(cin) / 16; o.cout = d.cout;
        if (d.bn.channels) { o.alpha_off = (int32_t)d.bn.alpha; o.beta_off = (int32_t)d.bn.beta; }   // patched below
        o.act = d.relu ? RZ_ACT_RELU : RZ_ACT_NONE;
        o.w_off = rz_add_pack(P, first_pack, R.w_base, d.w, 9, cin, d.cin, d.cout, cursor);
        if (o.cchunks >= 256 || 9 * o.cchunks >= 4096) return false;   // reciprocal range of the kernel's chunk decode
        max_c = std::max(max_c, std::max(cin, d.cout));
        break;
      }
      case OP_CONV1:
        o.kind = RZ_GEMM; o.rows = RZ_ROWS_POS; o.taps = 1;
        o.in_off = si.slot; o.out_off = so.off; o.out_tstride = so.tstride; o.out_layout = RZ_OUT_FLAT;
        o.cchunks = rz_round16(d.cin) / 16; o.cout = d.cout;
        o.bias_off = (int32_t)d.b;
        o.w_off = rz_add_pack(P, first_pack, R.w_base, d.w, 1, d.cin, d.cin, d.cout, cursor);
        max_c = std::max(max_c, d.cin);
        break;
      case OP_LINEAR:
        if (d.w_stride != d.in_features) return false;
        o.kind = RZ_GEMM; o.rows = RZ_ROWS_TREE; o.taps = 1;
        o.in_off = si.off; o.in_tstride = si.tstride;
        o.out_off = so.off; o.out_tstride = so.tstride; o.out_layout = RZ_OUT_FLAT;
        o.cchunks = rz_round16(d.in_features) / 16; o.cout = d.out_features;
        o.bias_off = (int32_t)d.b;
        o.act = d.elu ? RZ_ACT_ELU : RZ_ACT_NONE;
        o.w_off = rz_add_pack(P, first_pack, R.w_base, d.w, 1, d.in_features, d.in_features, d.out_features, cursor);
        if (o.cchunks * 16 > si.tstride) return false;
        break;
      default:  // OP_SCALE
        o.kind = RZ_SCALE; o.rows = RZ_ROWS_POS;
        o.in_off = si.slot; o.out_off = so.slot; o.out_layout = RZ_OUT_PADDED;
        o.channels = d.groups_per_sample;
        o.store_hidden = (d.out == BUF_HIDDEN) ? 1 : 0;
        max_c = std::max(max_c, d.groups_per_sample);
        break;
    }
    if (o.kind == RZ_GEMM) {
      o.nchunks = o.taps * o.cchunks;
      o.taps |= (int32_t)((((1u << 20) + (uint32_t)o.cchunks - 1) / (uint32_t)o.cchunks) << 8);
      o.wchunks = (o.nchunks + 1) & ~1;
      const int nt_total = (o.cout + 15) / 16;
      for (int k = 0; k < 2; ++k) {   // column tiles over the waves first (a power of two of them)
        const int nw = k ? 8 : 4;
        int lg = 0;
        while ((2 << lg) <= nw && (2 << lg) <= nt_total) ++lg;
        o.sched |= (uint32_t)lg << (8 * k);
      }
    }
    const int which = (d.out == BUF_VALUE) ? 0 : (d.out == BUF_REWARD) ? 1 : (d.out == BUF_POLICY) ? 2 : -1;
    if (which >= 0) {
      if (so.spatial) return false;
      R.out_off[which] = so.off; R.out_ts[which] = so.tstride; R.out_n[which] = so.size;
    }
  }
  R.w_floats = (int32_t)(cursor - R.w_base);
  // small image: RzOp table, then the epilogue parameters of every GEMM (padded to whole column tiles), then
  // the action-plane tap sums of the first convolution
  R.small_base = cursor;
  cursor += rz_round4((int)(sizeof(RzOp) / 4) * R.n_ops);
  auto param = [&](int64_t src, int from_derived, int nvals) -> int32_t {
    RzCopy c;
    c.src = src; c.dst = cursor; c.n = nvals; c.npad = rz_round16(nvals); c.from_derived = from_derived;
    P.copies.push_back(c);
    cursor += c.npad;
    return (int32_t)(c.dst - R.small_base);
  };
  for (int i = 0; i < R.n_ops; ++i) {
    RzOp& o = R.ops[i];
    if (o.kind != RZ_GEMM) continue;
    if (o.alpha_off >= 0) { o.alpha_off = param(o.alpha_off, 1, o.cout); o.beta_off = param(o.beta_off, 1, o.cout); }
    if (o.bias_off >= 0) o.bias_off = param(o.bias_off, 0, o.cout);
  }
  if (R.use_action) {
    const OpDesc& d = prog[first];
    RzAsum s;
    s.src = d.w; s.dst = cursor; s.cout = d.cout; s.cin_total = d.cin; s.H = h; s.W = w;
    R.ops[0].asum_off = (int32_t)(cursor - R.small_base);
    cursor += rz_round4(d.cout * HW);
    P.asums.push_back(s);
  }
  // A-fragment offset tables (values need the activation geometry: rz_finish_program)
  R.aoff_base = (int32_t)(cursor - R.small_base);
  for (int i = 0; i < R.n_ops; ++i) {
    RzOp& o = R.ops[i];
    if (o.kind != RZ_GEMM) continue;
    o.aoff_off = (uint32_t)(cursor - R.small_base);
    cursor += rz_aoff_ints(o.nchunks);
  }
  R.aoff.assign((size_t)(cursor - R.small_base) - R.aoff_base, 0);
  R.small_floats = (int32_t)(cursor - R.small_base);
  R.in_off = inst[ext].slot;
  rz_schedule(R, HW);
  R.ok = 1;
  return true;
}

// min-max scratch [2 * T * Cs], actval [T], cycle stamps of the profiling mode
constexpr int RZ_STAMP_WORDS = (RZ_MAX_OPS + 4) + 8 * RZ_MAX_OPS;   // uint64 stamps: per operator + 8 intra-operator each
inline int rz_scratch_floats(const RzGeometry& g, int T) { return rz_round4(2 * T * g.Cs + T) + 2 * RZ_STAMP_WORDS; }

// LDS floats of a workgroup of T trees: row tables, scratch, regions, optionally the weight image
// (`tables` = false: without the A-fragment offset tables at the end of the small image, which only the
// 4-wave kernels read)
inline int64_t rz_lds_floats(const RzGeometry& g, const RzProgram& R, int T, bool weights_in_lds, bool tables = true) {
  const int mpad = rz_round16(T * g.HW);
  const int64_t scratch = rz_scratch_floats(g, T);
  return (int64_t)2 * mpad + rz_round4(R.n_ops * 8) + scratch + (tables ? R.small_floats : R.aoff_base) +
         (int64_t)T * (3 * g.slot_ts + R.flat_floats) + (weights_in_lds ? R.w_floats : 0);
}

inline int rz_max_trees(const RzGeometry& g, const RzProgram& R, bool weights_in_lds) {
  int best = 0;
  for (int T = 1; T <= RZ_MAX_TREES && T * g.HW <= RZ_MAX_ROWS; ++T)
    if (4 * rz_lds_floats(g, R, T, weights_in_lds) <= RZ_LDS_BUDGET) best = T;
  return best;
}

// slot indices -> per-tree float offsets, flat offsets -> behind the three slots
inline void rz_finish_program(const RzGeometry& g, RzProgram& R) {
  if (!R.ok) return;
  const int flat0 = 3 * g.slot_ts;
  for (int i = 0; i < R.n_ops; ++i) {
    RzOp& o = R.ops[i];
    const bool in_spatial = !(o.kind == RZ_GEMM && o.rows == RZ_ROWS_TREE);
    if (in_spatial) { o.in_off *= g.slot_ts; o.in_tstride = g.slot_ts; } else { o.in_off += flat0; }
    if (o.out_layout == RZ_OUT_PADDED) { o.out_off *= g.slot_ts; o.out_tstride = g.slot_ts; } else { o.out_off += flat0; }
    if (o.res_off >= 0) o.res_off *= g.slot_ts;
  }
  for (int i = 0; i < R.n_ops; ++i) {
    const RzOp& o = R.ops[i];
    if (o.kind != RZ_GEMM) continue;
    int32_t* tbl = R.aoff.data() + ((int32_t)o.aoff_off - R.aoff_base);
    for (int k = 0; k < rz_aoff_ints(o.nchunks); ++k)
      tbl[k] = rz_aoff_entry(k, o.nchunks, o.taps & 0xFF, o.cchunks, g.PW, g.Cs);
  }
  R.in_off *= g.slot_ts;
  for (int k = 0; k < 3; ++k) if (R.out_off[k] >= 0) R.out_off[k] += flat0;
  if (rz_max_trees(g, R, false) < 1) R.ok = 0;
}

// Plans both programs of a residual network; called by mzx_net_create after NetBuilder::build.
inline void rz_plan(mzx_net* net) {
  RzPlan& P = net->rz;
  P = RzPlan();
  if (net->cfg.network != 1) return;
  const int h = net->hh, w = net->hw;
  if (h < 1 || w < 1 || h * w > RZ_MAX_ROWS) return;
  int64_t cursor = (net->derived_floats + 3) & ~int64_t(3);
  int max_c = 4;
  if (!rz_build_program(net, net->prog_initial, P, P.initial, cursor, max_c)) P.initial.ok = 0;
  if (!rz_build_program(net, net->prog_recurrent, P, P.recurrent, cursor, max_c)) P.recurrent.ok = 0;
  if (!P.initial.ok && !P.recurrent.ok) { P = RzPlan(); return; }
  RzGeometry& g = P.g;
  g.H = h; g.W = w; g.HW = h * w; g.PW = w + 2; g.PP = (h + 2) * (w + 2);
  g.Cs = rz_round16(max_c) + 8;
  g.slot_ts = g.PP * g.Cs;
  rz_finish_program(g, P.initial);
  rz_finish_program(g, P.recurrent);
  if (!P.initial.ok && !P.recurrent.ok) { P = RzPlan(); return; }
  // down-sampling stem: its 3x3 convolutions (stride 1 / 2, large maps) get packed weights for rz_stem_conv_kernel
  if (P.initial.ok && P.initial.first > 0) {
    for (int i = 0; i < P.initial.first; ++i) {
      const OpDesc& d = net->prog_initial[i];
      if (d.kind != OP_CONV3 || d.use_action || d.cout > 64 || d.cin > 240) continue;
      RzStemConv sc;
      sc.op_index = i;
      sc.cchunks = rz_round16(d.cin) / 16;
      sc.nchunks = 9 * sc.cchunks;
      sc.wchunks = (sc.nchunks + 1) & ~1;
      sc.w_off = rz_add_pack(P, P.packs.size(), 0, d.w, 9, d.cin, d.cin, d.cout, cursor);
      P.stem.push_back(sc);
    }
  }
  P.derived_floats = cursor;
  P.ok = 1;
}

// Latency experiments (profiles/r01_rz_latency_experiments.txt) are compiled in only on request: even a
// never-taken uniform branch per layer costs a few percent on small networks.
#ifdef MZX_RZ_EXPERIMENT
#define RZ_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define RZ_DBG(a, bit) false
#endif

struct RzArgs {
  int32_t n_ops, T, batch, num_actions;
  int32_t H, W, HW, PW, Cs, slot_ts, tree_floats, mpad, scratch_floats;
  uint32_t magic_hw, magic_w;     // ceil(2^32 / HW), ceil(2^32 / W): divisions by multiplication
  uint32_t magic_pt;              // ceil(2^32 / (in_channels * HW))
  int32_t in_off, in_channels, use_action;
  int32_t out_off[3], out_ts[3], out_n[3];
  int32_t hidden_floats;          // C * H * W
  int32_t in_nodes, out_nodes;    // nodes per sample of the in / hidden-out tensors (1 = dense)
  int32_t dbg;                    // latency experiments (builds with -DMZX_RZ_EXPERIMENT only, env MZX_RZ_DBG):
                                  // 1 skip the K loops, 2 skip the epilogues, 4 skip the layer GEMMs, 8 skip the barriers
  int32_t dump_op;                // >= 0: stop after this op and copy its output to `dump`; -2: cycle profile
                                  // (workgroup 0 writes s_memtime stamps after staging, input load and each op)
  int32_t w_floats, small_floats; // sizes of the program's weight image / small image
  int32_t fast;                   // 1: operators of a fast class run rz_gemm_fast (MZX_RZ_FAST=0: A/B knob)
  const float* in;
  const int32_t* in_node;
  const int32_t* out_node;
  const int32_t* action;
  float* hidden_out;
  float* outs[3];                 // value, reward, policy logits (nullable)
  float* dump;
  const float* weights;           // the program's weight image (global)
  const float* small;             // the program's small image (global): RzOp table, epilogue parameters
};


// ---------------------------------------------------------------------------
// Operator classes, decided once per launch.  The layer GEMMs of the reference's residual network at <= 16
// channels get straight-line code when a wave's share of the operator is ONE row tile (small boards: C3, C5) --
// K loop fully unrolled with every fragment requested up front, every per-layer option a compile-time constant
// -- because on those networks a layer is ~1.2 k cycles of MFMA inside ~3-4 k cycles of interpretation (descriptor
// decode, offset tables, option tests executed by a lone wave per SIMD).  Same MFMA order and epilogue arithmetic
// as rz_gemm_tiles<1>: the results are bit-identical.  Anything else runs on the interpreter.
enum RzFastClass {
  RZ_FAST_NONE = 0,   // any GEMM: rz_gemm_tiles
  RZ_FAST_SCALE_GEN,  // (wave kernel) any scaling operator
  RZ_FAST_SCALE16,    // (wave kernel) scaling, <= 16 planes of <= 16 positions: in registers
  RZ_FAST_CONV,       // 3x3, <= 16 -> <= 16 channels, folded BatchNorm, ReLU
  RZ_FAST_CONV_ASUM,  // ... + the action plane of the dynamics input
  RZ_FAST_CONV_RES,   // ... + residual
  RZ_FAST_CONV1,      // 1x1 head convolution, <= 16 -> <= 16 channels, bias, flat output
  RZ_FAST_FC9_ELU,    // head MLP layer, 129 .. 144 inputs (nine 16-deep chunks), <= 16 outputs, ELU
  RZ_FAST_FC1,        // head MLP layer, <= 16 inputs, any number of outputs, no activation
};

MZX_HD inline int rz_classify(const RzOp& op, int HW) {
  if (op.kind != RZ_GEMM) return (op.channels <= 16 && HW <= 16) ? RZ_FAST_SCALE16 : RZ_FAST_SCALE_GEN;
  const bool pos = op.rows == RZ_ROWS_POS, padded = op.out_layout == RZ_OUT_PADDED;
  const int taps = op.taps & 0xFF;
  const bool bn = op.alpha_off >= 0, bias = op.bias_off >= 0, res = op.res_off >= 0, asum = op.asum_off >= 0;
  if (pos && taps == 9 && op.cchunks == 1 && op.cout <= 16 && padded && bn && !bias && op.act == RZ_ACT_RELU) {
    if (res) return asum ? RZ_FAST_NONE : RZ_FAST_CONV_RES;
    return asum ? RZ_FAST_CONV_ASUM : RZ_FAST_CONV;
  }
  if (bn || res || asum || !bias || padded) return RZ_FAST_NONE;
  if (pos && taps == 1 && op.cchunks == 1 && op.cout <= 16 && op.act == RZ_ACT_NONE) return RZ_FAST_CONV1;
  if (!pos && op.nchunks == 9 && op.cout <= 16 && op.act == RZ_ACT_ELU) return RZ_FAST_FC9_ELU;
  if (!pos && op.nchunks == 1 && op.act == RZ_ACT_NONE) return RZ_FAST_FC1;
  return RZ_FAST_NONE;
}

#ifndef MZX_HOSTCHECK

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RzCtx {
  float* reg;           // workgroup LDS regions (region r of the program starts at T * r.off)
  const int* rowaddr;   // [mpad] activation address of row m = (tree, position): t * slot_ts + ((y+1) * PW + x + 1) * Cs
  const int* rowtp;     // [mpad] (t << 16) | position (0 for rows beyond T * HW)
  const int* rowout;    // output / residual address of row m (the stem kernel's tiles differ from rowaddr); null: rowaddr
  const unsigned* work; // [n_ops][waves] work words (rz_work_word), built once per launch
  unsigned long long* fine;   // profiling: eight intra-operator clock stamps of wave 0 (null: off)
  float* scratch;       // [2 * T * Cs] min-max scratch, then actval[T] = action / |A| per tree
  const float* wlds;    // LDS copy of the weight image (WLDS kernels)
  const float* simg;    // LDS copy of the small image
  int T, lane, wave, tid;
};

// x / d for 0 <= x < 2^32 / d with magic = ceil(2^32 / d) (d >= 2), no division instruction
__device__ __forceinline__ int rz_div(int x, int d, unsigned magic) {
  return d == 1 ? x : (int)__umulhi((unsigned)x, magic);
}

// One group of MT row tiles x one column tile of a layer GEMM: K loop + epilogue.
// TBL: per-chunk A-fragment offsets come from the precomputed table (4-wave kernels: one wave per SIMD, the
// ~20 scalar instructions of the tap decode per chunk are exposed).  The 8-wave kernels sit at the 256-register
// limit of two waves per SIMD -- the 8 registers of the offset quads would spill -- and decode arithmetically;
// their second wave hides the scalar work.
template <int MT, bool WLDS, bool TBL>
__device__ __forceinline__ void rz_gemm_tiles(const RzOp& op, const RzArgs& a, const RzCtx& cx, int nt, int mt0,
                                              int mt_step) {
  const int lane = cx.lane, T = cx.T;
#ifdef MZX_RZ_EXPERIMENT
#define RZ_FINE(k) if (cx.fine && cx.tid == 0) cx.fine[k] = __builtin_readcyclecounter();
#else
#define RZ_FINE(k)
#endif
  RZ_FINE(1)
  const bool pos_rows = (op.rows == RZ_ROWS_POS);
  const int rows = pos_rows ? T * a.HW : T;
  const float* in = cx.reg + T * op.in_off;
  f32x4 acc[MT];
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int m = (mt0 + i * mt_step) * 16 + (lane & 15);
    if (m >= rows) m = 0;
    abase[i] = (pos_rows ? cx.rowaddr[m] : m * op.in_tstride) + 4 * (lane >> 4);
  }
  const f32x4* wp = (const f32x4*)((WLDS ? cx.wlds : a.weights) + op.w_off) + (size_t)nt * op.wchunks * 64 + lane;
  // chunk c -> LDS offset of its A fragment: a precomputed table (rz_aoff_entry) instead of ~20 scalar
  // instructions of tap decode per chunk, which the small layers cannot hide behind four MFMAs
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int* tbl = (const int*)cx.simg + op.aoff_off;
  // Summation order of an output element: the K-steps of a chunk alternate between two accumulator tiles that are
  // added at the end.  With a single row tile consecutive MFMAs would otherwise form one dependent chain (40-cycle
  // accumulator latency against a 32-cycle issue interval); the 4-wave kernels (TBL) do the same for every tile
  // count, so that their results -- and the wave-per-tree kernel's (mzx_resnet_wave.h), all one-tile GEMMs -- do
  // not depend on how many row tiles a wave happens to own (trees per workgroup, operators sharing a slot).  The
  // 8-wave kernels sit at the register limit and keep one accumulator per tile beyond the first.
  constexpr bool ALT = TBL || MT == 1;
  f32x4 acc_odd[ALT ? MT : 1];
#pragma unroll
  for (int i = 0; i < (ALT ? MT : 1); ++i) acc_odd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const f32x4 (&av)[MT], const f32x4& bv, bool) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        // weight fragment FIRST: the MFMA then leaves D transposed -- lane l holds row l & 15 of the tile and the four
        // consecutive output channels 4 (l >> 4) .. + 3 (same products, same k order: the same bits), which is what
        // the position-major LDS layout wants: one 16-byte residual read and one 16-byte store per lane and tile
        if (ALT && (j & 1)) acc_odd[ALT ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i][j], acc_odd[ALT ? i : 0], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i][j], acc[i], 0, 0, 0);
      }
  };
  auto load_a = [&](int off, f32x4 (&av)[MT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) av[i] = *(const f32x4*)(in + abase[i] + off);
  };
  // arithmetic decode: chunk c -> (tap, 16-channel chunk of the tap), tap = c / cchunks by a 20-bit reciprocal
  // (exact for c < 4096 and cchunks < 256, the planner checks both)
  const unsigned magic = (unsigned)op.taps >> 8;
  const bool nine = (op.taps & 0xFF) == 9;
  const int pw9 = nine ? a.PW * a.Cs : 0, one9 = nine ? a.Cs : 0;
  const int last = op.nchunks - 1;
  auto decode = [&](int c) {
    c = c < last ? c : last;
    const int tap = nine ? (int)(((unsigned)c * magic) >> 20) : 0;
    const int cc = c - tap * op.cchunks;
    const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;                 // tap / 3, tap % 3 for tap < 9
    return (ty - 1) * pw9 + (tx - 1) * one9 + cc * 16;
  };
  // an odd chunk count is stored with one trailing zero chunk; prefetches beyond it re-read the last chunk
  const int wlast = op.wchunks - 1;
  auto load_b = [&](int c, f32x4& bv) { bv = wp[(size_t)(c < wlast ? c : wlast) * 64]; };
  // Fetch everything the epilogue needs that does not depend on the accumulators -- the lane's row address and its
  // residual quad -- BEFORE the K loop, so that these LDS round trips hide under the MFMAs instead of following them
  // (six registers per tile: up to four row tiles in the 4-wave kernels; the 8-wave kernels sit at the 256-register
  // limit of two waves per SIMD and prefetch for at most two).
  constexpr bool EARLY = TBL ? (MT <= 4) : (MT <= 2);
  const int m_lane = lane & 15;
  const int n0 = nt * 16 + 4 * (lane >> 4);          // this lane's four output channels n0 .. n0 + 3
  const int* rowo = cx.rowout ? cx.rowout : cx.rowaddr;
  const bool padded = (op.out_layout == RZ_OUT_PADDED);
  const bool need_tp = pos_rows && (!padded || op.asum_off >= 0);
  const float* res = (op.res_off >= 0) ? cx.reg + T * op.res_off : nullptr;   // padded layout: row address + channel
  int e_ra[EARLY ? MT : 1], e_tp[EARLY ? MT : 1];
  f32x4 e_rs[EARLY ? MT : 1];
  if (EARLY) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = (mt0 + i * mt_step) * 16 + m_lane;
      e_ra[i] = pos_rows ? rowo[m] : 0;
      e_tp[i] = need_tp ? cx.rowtp[m] : 0;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) e_rs[i] = res ? *(const f32x4*)(res + e_ra[i] + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  RZ_FINE(2)
  if (!RZ_DBG(a, 1)) {
  if (TBL) {
    // one wave per SIMD: nothing else hides the LDS round trip of a fragment, so four chunks are resident and
    // every buffer is refilled (four chunks ahead) as soon as its MFMAs have issued
    f32x4 A0[MT], A1[MT], A2[MT], A3[MT], B0, B1, B2, B3;
    const int n = op.nchunks;
    {
      const i32x4 q = *(const i32x4*)tbl;          // chunks 0 .. 3
      load_b(0, B0); load_a(q[0], A0);
      load_b(1, B1); load_a(q[1], A1);
      load_b(2, B2); load_a(q[2], A2);
      load_b(3, B3); load_a(q[3], A3);
    }
    for (int c = 0; c < n; c += 4) {
      const i32x4 q = *(const i32x4*)(tbl + c + 4);   // chunks c + 4 .. c + 7 (entries past the end repeat the last chunk)
      __builtin_amdgcn_sched_barrier(0);
      compute(A0, B0, false);
      __builtin_amdgcn_sched_barrier(0);
      load_b(c + 4, B0); load_a(q[0], A0);
      if (c + 1 >= n) break;
      __builtin_amdgcn_sched_barrier(0);
      compute(A1, B1, false);
      __builtin_amdgcn_sched_barrier(0);
      load_b(c + 5, B1); load_a(q[1], A1);
      if (c + 2 >= n) break;
      __builtin_amdgcn_sched_barrier(0);
      compute(A2, B2, false);
      __builtin_amdgcn_sched_barrier(0);
      load_b(c + 6, B2); load_a(q[2], A2);
      if (c + 3 >= n) break;
      __builtin_amdgcn_sched_barrier(0);
      compute(A3, B3, false);
      __builtin_amdgcn_sched_barrier(0);
      load_b(c + 7, B3); load_a(q[3], A3);
    }
  } else {
    // software pipeline: A fragments (LDS) one chunk ahead, B fragments (L2) two chunks ahead of the 4 * MT
    // MFMAs being issued; ping-pong registers, four chunks per trip: no register copies in the rotation
    f32x4 a0[MT], a1[MT], b0, b1, b2, b3;
    load_b(0, b0);
    load_b(1, b1);
    load_a(decode(0), a0);
    for (int c = 0; c < op.nchunks; c += 4) {
      load_b(c + 2, b2);
      load_b(c + 3, b3);
      load_a(decode(c + 1), a1);
      __builtin_amdgcn_sched_barrier(0);
      compute(a0, b0, false);
      __builtin_amdgcn_sched_barrier(0);
      load_a(decode(c + 2), a0);
      __builtin_amdgcn_sched_barrier(0);
      compute(a1, b1, true);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 >= op.nchunks) break;
      load_b(c + 4, b0);
      load_b(c + 5, b1);
      load_a(decode(c + 3), a1);
      __builtin_amdgcn_sched_barrier(0);
      compute(a0, b2, false);
      __builtin_amdgcn_sched_barrier(0);
      load_a(decode(c + 4), a0);
      __builtin_amdgcn_sched_barrier(0);
      compute(a1, b3, true);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }
  if (ALT) {
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = acc[i] + acc_odd[ALT ? i : 0];
  }
  RZ_FINE(3)
  if (RZ_DBG(a, 2)) return;
  // ---- epilogue: every per-layer variation (BatchNorm or not, bias or not, residual or not, ReLU or not) is turned
  // into DATA before the element loop (identity scale / zero bias / -inf floor), so the loop body is straight-line
  // code.  Lane l owns row m_lane of every tile and channels n0 .. n0 + 3 (the epilogue parameters are padded to
  // whole column tiles, channel rows of an activation slot to Cs >= 16 c + 8 floats: the quads are always readable).
  __builtin_amdgcn_sched_barrier(0);   // the parameter reads stay behind the K loop (its registers are all in use)
  float al[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
  if (op.alpha_off >= 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { al[u] = cx.simg[op.alpha_off + n0 + u]; be[u] = cx.simg[op.beta_off + n0 + u]; }
  }
  if (op.bias_off >= 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) bi[u] = cx.simg[op.bias_off + n0 + u];
  }
  const float floor_v = (op.act == RZ_ACT_RELU) ? 0.f : -MZX_INF;
  float* out = cx.reg + T * op.out_off;
  const float* actval = cx.scratch + 2 * T * a.Cs;
  const int nstride = (!padded && pos_rows) ? a.HW : 1;       // address step per output channel
  const bool quad_ok = n0 + 3 < op.cout;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = (mt0 + i * mt_step) * 16 + m_lane;
    int ra = 0, tp = 0;
    f32x4 rs = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EARLY) {
      ra = e_ra[i]; tp = e_tp[i]; rs = e_rs[i];
    } else {
      if (pos_rows) ra = rowo[m];
      if (need_tp) tp = cx.rowtp[m];
      if (res) rs = *(const f32x4*)(res + ra + n0);
    }
    int base;
    if (padded) base = ra;
    else if (pos_rows) base = (tp >> 16) * op.out_tstride + (tp & 0xFFFF);
    else base = m * op.out_tstride;
    if (op.asum_off >= 0) {   // action plane of the dynamics input (first layer only)
      const float av = actval[tp >> 16];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[i][u] += av * cx.simg[op.asum_off + (n0 + u) * a.HW + (tp & 0xFFFF)];
    }
    f32x4 v;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float x = acc[i][u] * al[u] + be[u];
      x = x + bi[u];
      x = x + rs[u];
      v[u] = fmaxf(x, floor_v);
    }
    if (op.act == RZ_ACT_ELU) {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = mzx_elu(v[u]);
    }
    if (m < rows) {
      if (padded && quad_ok) *(f32x4*)(out + base + n0) = v;
      else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (n0 + u < op.cout) out[base + (n0 + u) * nstride] = v[u];
      }
    }
  }
  RZ_FINE(4)
#undef RZ_FINE
}

// operator descriptor o of the program image (24 words at a wave-uniform address).  Measured on C3/C4/C5:
// reading the LDS-resident copy beats scalar loads of the global copy by 3-7 % of the whole search.
__device__ __forceinline__ RzOp rz_fetch_op(const float* image, int o) {
  return ((const RzOp*)image)[o];
}

// What one wave does for one operator, decided once per launch (rz_setup) instead of per operator and
// simulation: the team / tile arithmetic and its branches were ~1 000 cycles of exposed scalar code per
// operator on the small networks.  Bits 0-3 / 4-7: row tiles of the wave's first / second group (0 = none;
// first 0 = the wave idles), 8-11 first column tile, 12-15 column tile step, 16-19 first row tile, 20-23 row
// tile step.  Column tiles go to the waves first (a power of two of them), row tiles to the rest; an
// operator that shares its slot runs on its team only (rz_schedule): per-tree GEMMs spread column tiles,
// per-position GEMMs row tiles over the team.
__device__ __forceinline__ unsigned rz_work_word(const RzOp& op, int T, int HW, int nw, int wave) {
  if (op.kind != RZ_GEMM) return 0u;
  const int rows = (op.rows == RZ_ROWS_POS) ? T * HW : T;
  const int mt_total = (rows + 15) >> 4, nt_total = (op.cout + 15) >> 4;
  const unsigned team = (nw == 8) ? (op.team >> 16) : (op.team & 0xFFFFu);
  const int t_cnt = (int)(team >> 8);
  int wn, wm, waves_n, waves_m;
  if (t_cnt == 0) {
    const int lg_n = (int)((op.sched >> (nw == 8 ? 8 : 0)) & 0xFFu);
    waves_n = 1 << lg_n; waves_m = nw >> lg_n;
    wn = wave & (waves_n - 1); wm = wave >> lg_n;
  } else {
    const int tw = wave - (int)(team & 0xFFu);
    if (tw < 0 || tw >= t_cnt) return 0u;
    if (op.rows == RZ_ROWS_TREE) { waves_n = t_cnt; waves_m = 1; wn = tw; wm = 0; }
    else { waves_n = 1; waves_m = t_cnt; wn = 0; wm = tw; }
  }
  if (wn >= nt_total || wm >= mt_total) return 0u;
  const int mine = (mt_total - wm + waves_m - 1) / waves_m;     // row tiles wm, wm + waves_m, ...
  const int cnt0 = mine < 8 ? mine : 8, cnt1 = mine - cnt0 < 8 ? mine - cnt0 : 8;
  return (unsigned)cnt0 | ((unsigned)cnt1 << 4) | ((unsigned)wn << 8) | ((unsigned)waves_n << 12) |
         ((unsigned)wm << 16) | ((unsigned)waves_m << 20);
}

enum { RZ_K_TAP9 = 0, RZ_K_LIN1 = 1, RZ_K_LIN9 = 2 };
enum { RZ_EP_BN_RELU = 0, RZ_EP_BN_RELU_ASUM, RZ_EP_BN_RES_RELU, RZ_EP_BIAS_POS, RZ_EP_BIAS_ELU_TREE, RZ_EP_BIAS_TREE };

// One row tile x one column tile of an operator of a fast class (see RzFastClass): rz_gemm_tiles<1> with the K
// structure and the epilogue options as template parameters.
template <int KS, int EP, bool WLDS>
__device__ __forceinline__ void rz_gemm_fast(const RzOp& op, const RzArgs& a, const RzCtx& cx, int nt, int mt0) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr bool POS = (EP <= RZ_EP_BIAS_POS);
  constexpr bool BN = (EP <= RZ_EP_BN_RES_RELU);
  constexpr int NCH = (KS == RZ_K_LIN1) ? 1 : 9;
  const int lane = cx.lane, T = cx.T, g4 = 4 * (lane >> 4);
  const int rows = POS ? T * a.HW : T;
  const float* in = cx.reg + T * op.in_off;
  int m = mt0 * 16 + (lane & 15);
  if (m >= rows) m = 0;
  const int abase = (POS ? cx.rowaddr[m] : m * op.in_tstride) + g4;
  const f32x4* wp = (const f32x4*)((WLDS ? cx.wlds : a.weights) + op.w_off) + (size_t)nt * op.wchunks * 64 + lane;
  const int n = nt * 16 + (lane & 15);
  const int m0 = mt0 * 16 + g4;
  const int pwcs = a.PW * a.Cs;
  f32x4 A[NCH], B[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {   // chunk 0 first: LDS answers in request order
    const int off = (KS == RZ_K_TAP9) ? (c / 3 - 1) * pwcs + (c % 3 - 1) * a.Cs : c * 16;   // rz_aoff_entry, one chunk per tap
    B[c] = wp[c * 64];
    A[c] = *(const f32x4*)(in + abase + off);
    if (c == 0) __builtin_amdgcn_sched_barrier(0);
  }
  // epilogue operands: their LDS round trips hide under the MFMAs
  const int* rowo = cx.rowout ? cx.rowout : cx.rowaddr;
  i32x4 ra4 = i32x4{0, 0, 0, 0}, tp4 = i32x4{0, 0, 0, 0};
  if (POS && EP != RZ_EP_BIAS_POS) ra4 = *(const i32x4*)(rowo + m0);
  if (EP == RZ_EP_BIAS_POS || EP == RZ_EP_BN_RELU_ASUM) tp4 = *(const i32x4*)(cx.rowtp + m0);
  float al = 1.f, be = 0.f, bi = 0.f;
  if (BN) { al = cx.simg[op.alpha_off + n]; be = cx.simg[op.beta_off + n]; } else { bi = cx.simg[op.bias_off + n]; }
  float rs[4] = {0.f, 0.f, 0.f, 0.f}, as[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
  if (EP == RZ_EP_BN_RES_RELU) {
    const float* res = cx.reg + T * op.res_off;
#pragma unroll
    for (int r = 0; r < 4; ++r) rs[r] = res[ra4[r] + n];
  }
  if (EP == RZ_EP_BN_RELU_ASUM) {
    const float* actval = cx.scratch + 2 * T * a.Cs;
#pragma unroll
    for (int r = 0; r < 4; ++r) { av[r] = actval[tp4[r] >> 16]; as[r] = cx.simg[op.asum_off + n * a.HW + (tp4[r] & 0xFFFF)]; }
  }
  // every request above is issued before the first MFMA (the scheduler would otherwise re-serialise them into a
  // load -> wait -> four MFMAs chain per chunk, one LDS round trip each)
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc_odd = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][0], B[c][0], acc, 0, 0, 0);
    acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][1], B[c][1], acc_odd, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][2], B[c][2], acc, 0, 0, 0);
    acc_odd = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][3], B[c][3], acc_odd, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  acc = acc + acc_odd;
  const bool nv = n < op.cout;
  float* out = cx.reg + T * op.out_off;
  if (EP == RZ_EP_BN_RELU_ASUM) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += av[r] * as[r];
  }
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = acc[r] * al + be;
    x = x + bi;
    x = x + rs[r];
    v[r] = fmaxf(x, BN ? 0.f : -MZX_INF);
  }
  if (EP == RZ_EP_BIAS_ELU_TREE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = mzx_elu(v[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int at;
    if (BN) at = ra4[r] + n;                                                          // padded position-major output
    else if (EP == RZ_EP_BIAS_POS) at = (tp4[r] >> 16) * op.out_tstride + (tp4[r] & 0xFFFF) + n * a.HW;   // flat [channel][position]
    else at = (m0 + r) * op.out_tstride + n;                                          // rows = trees
    if (m0 + r < rows && nv) out[at] = v[r];
  }
}

template <int NW>
__device__ __forceinline__ int rz_op_class(const RzArgs& a, const RzCtx& cx, int o) {
  if (NW != 4 || a.fast == 0) return RZ_FAST_NONE;
  return __builtin_amdgcn_readfirstlane((int)cx.work[a.n_ops * NW + o]);
}

// MM: the most row tiles one wave ever gets in this launch (host-checked).  Kernels for small activation
// matrices are instantiated with MM = 3: without the code of the 4..8-tile variants the kernel is half the
// size and the per-operator dispatch -- exposed on the latency-bound small networks -- is shorter.
template <bool WLDS, int NW, int MM>
__device__ __forceinline__ void rz_gemm(const RzOp& op, const RzArgs& a, const RzCtx& cx, unsigned w, int cls = RZ_FAST_NONE) {
#ifdef MZX_RZ_EXPERIMENT
  if (cx.fine && cx.tid == 0) cx.fine[7] = __builtin_readcyclecounter() + (unsigned long long)(op.kind & 0);   // after the descriptor fetch
#endif
  const int cnt0 = (int)(w & 15u);
  if (cnt0 == 0) return;
  const int cnt1 = (int)((w >> 4) & 15u), wn = (int)((w >> 8) & 15u), waves_n = (int)((w >> 12) & 15u);
  const int wm = (int)((w >> 16) & 15u), waves_m = (int)((w >> 20) & 15u);
  const int nt_total = (op.cout + 15) >> 4;
  if constexpr (NW == 4 && MM == 3) {   // the kernels of the small networks
    if (cls >= RZ_FAST_CONV && cnt0 == 1 && cnt1 == 0) {
      for (int nt = wn; nt < nt_total; nt += waves_n) {
        switch (cls) {
          case RZ_FAST_CONV: rz_gemm_fast<RZ_K_TAP9, RZ_EP_BN_RELU, WLDS>(op, a, cx, nt, wm); break;
          case RZ_FAST_CONV_ASUM: rz_gemm_fast<RZ_K_TAP9, RZ_EP_BN_RELU_ASUM, WLDS>(op, a, cx, nt, wm); break;
          case RZ_FAST_CONV_RES: rz_gemm_fast<RZ_K_TAP9, RZ_EP_BN_RES_RELU, WLDS>(op, a, cx, nt, wm); break;
          case RZ_FAST_CONV1: rz_gemm_fast<RZ_K_LIN1, RZ_EP_BIAS_POS, WLDS>(op, a, cx, nt, wm); break;
          case RZ_FAST_FC9_ELU: rz_gemm_fast<RZ_K_LIN9, RZ_EP_BIAS_ELU_TREE, WLDS>(op, a, cx, nt, wm); break;
          default: rz_gemm_fast<RZ_K_LIN1, RZ_EP_BIAS_TREE, WLDS>(op, a, cx, nt, wm); break;
        }
      }
      return;
    }
  }
  for (int nt = wn; nt < nt_total; nt += waves_n) {
    for (int g = 0; g < 2; ++g) {
      const int cnt = g ? cnt1 : cnt0;   // wave-uniform
      if (cnt == 0) break;
      const int mt0 = wm + g * 8 * waves_m;
      switch (cnt) {
        case 1: if constexpr (MM >= 1) rz_gemm_tiles<1, WLDS, NW == 4>(op, a, cx, nt, mt0, waves_m); break;
        case 2: if constexpr (MM >= 2) rz_gemm_tiles<2, WLDS, NW == 4>(op, a, cx, nt, mt0, waves_m); break;
        case 3: if constexpr (MM >= 3) rz_gemm_tiles<3, WLDS, NW == 4>(op, a, cx, nt, mt0, waves_m); break;
        case 4: if constexpr (MM >= 4) rz_gemm_tiles<4, WLDS, NW == 4>(op, a, cx, nt, mt0, waves_m); break;
        case 5: if constexpr (MM >= 5) rz_gemm_tiles<5, WLDS, NW == 4>(op, a, cx, nt, mt0, waves_m); break;
        case 6: if constexpr (MM >= 6) rz_gemm_tiles<6, WLDS, NW == 4>(op, a, cx, nt, mt0, waves_m); break;
        case 7: if constexpr (MM >= 7) rz_gemm_tiles<7, WLDS, NW == 4>(op, a, cx, nt, mt0, waves_m); break;
        default: if constexpr (MM >= 8) rz_gemm_tiles<8, WLDS, NW == 4>(op, a, cx, nt, mt0, waves_m); break;
      }
    }
  }
}

// per-plane min-max scaling (models.py:527-553, :574-599) + hidden-state store
template <int NW>
__device__ __forceinline__ void rz_scale(const RzOp& op, const RzArgs& a, const RzCtx& cx, int b0, int ntree,
                                         const int32_t* out_node, bool local) {
  constexpr int NT = NW * 64;
  const int T = cx.T, C = op.channels;
  const float* in = cx.reg + T * op.in_off;
  float* out = cx.reg + T * op.out_off;
  // plane minimum / maximum: one 16-lane row per (tree, channel) plane, lanes stride over the positions, then
  // a four-step butterfly (min and max are exact in any order)
  {
    const int sub = cx.tid & 15, grp = cx.tid >> 4, planes = T * C;
    for (int base = 0; base < planes; base += NT / 16) {   // uniform trip count: the shuffles need the whole wave
      const int pr = base + grp;
      const bool valid = pr < planes;
      const int t = valid ? pr / C : 0, c = valid ? pr - t * C : 0;
      float lo = MZX_INF, hi = -MZX_INF;
      for (int p = sub; p < a.HW; p += 16) {
        const float v = in[cx.rowaddr[t * a.HW + p] + c];
        lo = fminf(lo, v); hi = fmaxf(hi, v);
      }
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, m, 16));
        hi = fmaxf(hi, __shfl_xor(hi, m, 16));
      }
      if (valid && sub == 0) {
        float sc = hi - lo;
        if (sc < 1e-5f) sc += 1e-5f;
        cx.scratch[2 * pr] = lo;
        cx.scratch[2 * pr + 1] = sc;
      }
    }
  }
  __syncthreads();
  const int per_tree = C * a.HW;
  for (int t = 0; t < T; ++t) {
    const int64_t s = b0 + t;
    const int64_t node = (op.store_hidden && t < ntree && out_node) ? out_node[local ? t : s] : 0;
    float* hid = a.hidden_out + (s * a.out_nodes + node) * a.hidden_floats;
    for (int rem = cx.tid; rem < per_tree; rem += NT) {
      const int c = rz_div(rem, a.HW, a.magic_hw), p = rem - c * a.HW;
      const int ra = cx.rowaddr[t * a.HW + p] + c;
      const float y = (in[ra] - cx.scratch[2 * (t * C + c)]) / cx.scratch[2 * (t * C + c) + 1];
      out[ra] = y;
      if (op.store_hidden && t < ntree && a.hidden_out) hid[rem] = y;
    }
  }
}

// One workgroup per CU by design (LDS-resident activations): tell the scheduler that registers are
// free (1 wave per SIMD) so that it keeps the prefetch distance of the software pipeline.
// LDS carve of a workgroup (offsets follow rz_lds_floats)
template <int NW>
__device__ __forceinline__ RzCtx rz_carve(const RzArgs& a, float* lds) {
  RzCtx cx;
  int* rowaddr = (int*)lds;
  int* rowtp = rowaddr + a.mpad;
  unsigned* work = (unsigned*)(rowtp + a.mpad);
  float* scratch = (float*)(work + ((a.n_ops * 8 + 3) & ~3));
  float* simg = scratch + a.scratch_floats;
  float* reg = simg + a.small_floats;
  cx.reg = reg; cx.rowaddr = rowaddr; cx.rowtp = rowtp; cx.scratch = scratch; cx.simg = simg; cx.rowout = nullptr;
  cx.work = work;
  cx.fine = nullptr;
  cx.wlds = reg + a.T * a.tree_floats;
  cx.T = a.T; cx.tid = threadIdx.x; cx.lane = threadIdx.x & 63; cx.wave = threadIdx.x >> 6;
  return cx;
}

// Once per launch: program image -> LDS, regions zeroed, row tables built.  Ends with a barrier.
template <bool WLDS, int NW>
__device__ __forceinline__ void rz_setup(const RzArgs& a, const RzCtx& cx) {
  constexpr int NT = NW * 64;
  const int tid = cx.tid, T = cx.T;
  // small image (operator table, epilogue parameters) -> LDS, always: no layer waits on HBM/L2 latency
  {
    const f32x4* src = (const f32x4*)a.small;
    f32x4* dst = (f32x4*)cx.simg;
    for (int i = tid; i < a.small_floats / 4; i += NT) dst[i] = src[i];
  }
  // weight image -> LDS (every weight leaves L2 once per workgroup), 16 bytes per lane per load
  if (WLDS) {
    const f32x4* src = (const f32x4*)a.weights;
    f32x4* dst = (f32x4*)cx.wlds;
    for (int i = tid; i < a.w_floats / 4; i += NT) dst[i] = src[i];
  }
  // zero every region (halo positions, pad channels and pad words stay zero for the whole launch)
  {
    f32x4* z = (f32x4*)cx.reg;
    for (int i = tid; i < T * a.tree_floats / 4; i += NT) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // work words of every (operator, wave)
  {
    unsigned* work = (unsigned*)cx.work;
    const RzOp* ops = (const RzOp*)a.small;
    for (int i = tid; i < a.n_ops * NW; i += NT) work[i] = rz_work_word(ops[i / NW], T, a.HW, NW, i % NW);
    // operator classes (RzFastClass) behind the work words of the 4-wave kernels (the table is sized for 8 waves)
    if (NW == 4) for (int i = tid; i < a.n_ops; i += NT) work[a.n_ops * NW + i] = (unsigned)rz_classify(ops[i], a.HW);
  }
  int* rowaddr = (int*)cx.rowaddr;
  int* rowtp = (int*)cx.rowtp;
  for (int m = tid; m < a.mpad; m += NT) {
    if (m < T * a.HW) {
      const int t = rz_div(m, a.HW, a.magic_hw), p = m - t * a.HW;
      const int y = rz_div(p, a.W, a.magic_w), x = p - y * a.W;
      rowaddr[m] = t * a.slot_ts + ((y + 1) * a.PW + x + 1) * a.Cs;
      rowtp[m] = (t << 16) | p;
    } else {
      rowaddr[m] = (a.PW + 1) * a.Cs;   // rows beyond the matrix: a valid address, never stored to
      rowtp[m] = 0;
    }
  }
  __syncthreads();
}

// Input tensors [in_channels][H][W] of the workgroup's trees -> position-major LDS layout, and the
// per-tree action value action / |A| of the dynamics input.  `in_node` / `action` are indexed by the
// tree's slot in the workgroup when `local` (LDS arrays of the search kernel), else by sample.
// Ends with a barrier.
template <int NW>
__device__ __forceinline__ void rz_load_input(const RzArgs& a, const RzCtx& cx, int b0, int ntree, const int32_t* in_node,
                                              const int32_t* action, bool local) {
  constexpr int NT = NW * 64;
  const int tid = cx.tid, T = cx.T;
  if (tid < T) {
    const int k = local ? tid : b0 + tid;
    cx.scratch[2 * T * a.Cs + tid] = (a.use_action && tid < ntree) ? (float)action[k] / (float)a.num_actions : 0.f;
  }
  float* dst = cx.reg + T * a.in_off;
  const int per_tree = a.in_channels * a.HW;
  // (tree, element) pairs are spread over the whole workgroup and every thread keeps four independent global
  // loads in flight per trip: the states sit in L2 / HBM, so the gather costs one round trip per trip instead
  // of one per tree
  const int total = ntree * per_tree;
  for (int i0 = tid; i0 < total; i0 += 4 * NT) {
    float v[4];
    int at[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT;
      v[u] = 0.f; at[u] = -1;
      if (i < total) {
        const int t = rz_div(i, per_tree, a.magic_pt), rem = i - t * per_tree;
        const int64_t s = b0 + t;
        const int64_t node = in_node ? in_node[local ? t : s] : 0;
        v[u] = a.in[(s * a.in_nodes + node) * per_tree + rem];
        const int c = rz_div(rem, a.HW, a.magic_hw), p = rem - c * a.HW;
        at[u] = cx.rowaddr[t * a.HW + p] + c;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (at[u] >= 0) dst[at[u]] = v[u];
  }
  __syncthreads();
}

template <bool WLDS, int NW, int MM>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
rz_network_kernel(const RzArgs a) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float rz_lds[];
  const unsigned long long t_entry = __builtin_readcyclecounter();
  const int tid = threadIdx.x, T = a.T;
  const int b0 = blockIdx.x * T;
  const int ntree = min(T, a.batch - b0);
  const RzCtx cx = rz_carve<NW>(a, rz_lds);
  float* reg = cx.reg;
  const int* rowaddr = cx.rowaddr;
  rz_setup<WLDS, NW>(a, cx);
  unsigned long long* stamps = (unsigned long long*)(cx.scratch + a.scratch_floats - 2 * RZ_STAMP_WORDS);   // LDS
  const bool prof = (a.dump_op == -2) && blockIdx.x == 0 && tid == 0;
  if (prof) { stamps[0] = t_entry; stamps[1] = __builtin_readcyclecounter(); }

  rz_load_input<NW>(a, cx, b0, ntree, a.in_node, a.action, false);
  if (prof) stamps[2] = __builtin_readcyclecounter();

  // ---- the layers: slots of independent operators (rz_schedule), one barrier per slot
  for (int o = 0; o < a.n_ops;) {
    const int slot_first = o;
    RzCtx cxo = cx;
    unsigned long long* fine = stamps + (RZ_MAX_OPS + 4) + 8 * o;   // [8] per operator, LDS
    if (a.dump_op == -2 && blockIdx.x == 0) { cxo.fine = fine; if (tid == 0) fine[0] = __builtin_readcyclecounter(); }
    bool last;
    do {
      const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)cx.work[o * NW + cx.wave]);
      const RzOp op = rz_fetch_op(cx.simg, o);   // from the LDS-resident program image
      if (op.kind == RZ_GEMM) rz_gemm<WLDS, NW, MM>(op, a, cxo, w, rz_op_class<NW>(a, cx, o));
      else rz_scale<NW>(op, a, cx, b0, ntree, a.out_node, false);
      last = ((op.sched >> 16) & 1u) != 0;
      ++o;
    } while (!last);
    if (cxo.fine && tid == 0) fine[5] = __builtin_readcyclecounter();
    __syncthreads();
    if (cxo.fine && tid == 0) fine[6] = __builtin_readcyclecounter();
    if (prof) for (int k = slot_first; k < o; ++k) stamps[3 + k] = __builtin_readcyclecounter();
    if (a.dump_op >= slot_first && a.dump_op < o) {  // diagnostics: the output tensor of one operator, dense per sample
      const RzOp op = rz_fetch_op(cx.simg, a.dump_op);
      const float* src = reg + T * op.out_off;
      if (op.out_layout == RZ_OUT_PADDED) {
        const int C = (op.kind == RZ_GEMM) ? op.cout : op.channels;
        const int per_tree = C * a.HW;
        for (int t = 0; t < ntree; ++t)
          for (int rem = tid; rem < per_tree; rem += NT) {
            const int c = rz_div(rem, a.HW, a.magic_hw), p = rem - c * a.HW;
            a.dump[(int64_t)(b0 + t) * per_tree + rem] = src[rowaddr[t * a.HW + p] + c];
          }
      } else {
        const int nfl = (op.rows == RZ_ROWS_POS) ? op.cout * a.HW : op.cout;
        for (int t = 0; t < ntree; ++t)
          for (int rem = tid; rem < nfl; rem += NT) a.dump[(int64_t)(b0 + t) * nfl + rem] = src[t * op.out_tstride + rem];
      }
      return;
    }
  }

  // ---- head logits
  for (int k = 0; k < 3; ++k) {
    if (!a.outs[k] || a.out_off[k] < 0) continue;
    const float* src = reg + T * a.out_off[k];
    const int nfl = a.out_n[k];
    for (int t = 0; t < ntree; ++t)
      for (int rem = tid; rem < nfl; rem += NT) a.outs[k][(int64_t)(b0 + t) * nfl + rem] = src[t * a.out_ts[k] + rem];
  }
  if (prof) {
    stamps[3 + a.n_ops] = __builtin_readcyclecounter();
    for (int k = 0; k < a.n_ops + 4; ++k) ((unsigned long long*)a.dump)[k] = stamps[k];
    for (int k = 0; k < 8 * a.n_ops; ++k) ((unsigned long long*)a.dump)[RZ_MAX_OPS + 4 + k] = stamps[RZ_MAX_OPS + 4 + k];
  }
}

// Trees per workgroup and whether the weight image is LDS-resident, for a batch.
inline void rz_choose(const RzGeometry& g, const RzProgram& R, int batch, int& T, bool& wlds) {
  T = (batch + 255) / 256;          // one workgroup per CU when the batch allows
  if (T < 1) T = 1;
  const int tmax = rz_max_trees(g, R, false);
  if (T > tmax) T = tmax;
  wlds = 4 * rz_lds_floats(g, R, T, true) <= RZ_LDS_BUDGET;
}

template <bool WLDS, int NW, int MM>
inline int rz_launch_k(const RzArgs& a, unsigned grid, size_t lds_bytes, stream_t stream) {
  static std::atomic<uint64_t> lds_attr_done{0};   // per instantiation, one bit per device
  if (const int ae = allow_large_lds((const void*)rz_network_kernel<WLDS, NW, MM>, 160 * 1024, lds_attr_done)) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  hipLaunchKernelGGL((rz_network_kernel<WLDS, NW, MM>), dim3(grid), dim3(NW * 64), lds_bytes, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("fused network launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

struct RzLaunch { RzArgs a; bool wlds; bool eight; bool small; unsigned grid; size_t lds; };

inline int rz_env_int(const char* name, int dflt) { return exp_int(name, dflt); }   // (instrumented builds only: mzx_tuning.h)

// Everything of a launch that depends on (network, program, batch); `extra_lds_floats` = LDS the caller
// adds behind the engine's own image (search kernel).
inline RzLaunch rz_prepare(const mzx_net* net, const RzProgram& R, const float* in, const NetBuffers& nb, int batch,
                           const NetIndex* ix, int64_t extra_lds_floats_per_tree = 0, int64_t extra_lds_floats = 0) {
  const RzGeometry& g = net->rz.g;
  RzLaunch L;
  RzArgs& a = L.a;
  memset(&a, 0, sizeof(a));
  bool wlds = false;
  rz_choose(g, R, batch, a.T, wlds);
  // tuning knobs for A/B measurements (bench / profiling only): trees per workgroup, waves per workgroup.
  // Measured on MI355X (profiles/r01_rz_tiling_ab.txt): the defaults below win on C3 and C4; splitting a CU
  // between two smaller workgroups loses more M-tile efficiency and weight reuse than it hides.
  const int force_T = rz_env_int("MZX_RZ_TREES", 0);
  if (force_T > 0 && force_T < a.T) a.T = force_T;
  // Two waves per SIMD (8-wave kernels) when the layer GEMMs have enough tiles to feed eight waves (big boards /
  // wide networks): latencies of one wave (weight prefetch, epilogue, barriers) hide behind the other's MFMAs.
  // Those kernels decode chunk offsets arithmetically and do not stage the offset tables, so the trees per
  // workgroup are sized without them first.
  const int force_w = rz_env_int("MZX_RZ_WAVES", 0);
  bool tables = true;
  {
    auto fits = [&](int T, bool w, bool tb) {
      return 4 * (rz_lds_floats(g, R, T, w, tb) + extra_lds_floats_per_tree * T + extra_lds_floats) <= RZ_LDS_BUDGET;
    };
    auto eight_for = [&](int T) {
      const int tiles = ((T * g.HW + 15) / 16) * ((net->cfg.channels + 15) / 16);
      return force_w == 8 || (force_w != 4 && tiles >= 16 && net->rz_waves != 4);
    };
    int T8 = a.T;
    while (T8 > 1 && !fits(T8, false, false)) --T8;
    L.eight = eight_for(T8);
    if (L.eight) {
      a.T = T8; tables = false;
    } else {
      while (a.T > 1 && !fits(a.T, false, true)) --a.T;
    }
    wlds = fits(a.T, true, tables);
  }
  a.n_ops = R.n_ops;
  a.batch = batch;
  a.num_actions = net->cfg.action_space_size;
  a.H = g.H; a.W = g.W; a.HW = g.HW; a.PW = g.PW; a.Cs = g.Cs; a.slot_ts = g.slot_ts;
  a.tree_floats = 3 * g.slot_ts + R.flat_floats;
  a.mpad = rz_round16(a.T * g.HW);
  a.scratch_floats = rz_scratch_floats(g, a.T);
  a.in_off = R.in_off; a.in_channels = R.in_channels; a.use_action = R.use_action;
  for (int k = 0; k < 3; ++k) { a.out_off[k] = R.out_off[k]; a.out_ts[k] = R.out_ts[k]; a.out_n[k] = R.out_n[k]; }
  a.hidden_floats = (int32_t)net->hidden_size;
  a.in_nodes = ix ? ix->in_nodes : 1;
  a.out_nodes = ix ? ix->out_nodes : 1;
  a.dump_op = -1;
  a.dbg = rz_env_int("MZX_RZ_DBG", 0);
  a.fast = rz_env_int("MZX_RZ_FAST", 1) != 0 ? 1 : 0;
  a.w_floats = R.w_floats;
  a.small_floats = tables ? R.small_floats : R.aoff_base;
  a.in = in;
  a.in_node = ix ? ix->in_node : nullptr;
  a.out_node = ix ? ix->out_node : nullptr;
  a.action = nb.action;
  a.hidden_out = nb.hidden;
  a.outs[0] = nb.value; a.outs[1] = nb.reward; a.outs[2] = nb.policy;
  a.dump = nullptr;
  a.weights = net->d_derived + R.w_base;
  a.small = net->d_derived + R.small_base;
  L.grid = (unsigned)((batch + a.T - 1) / a.T);
  L.lds = (size_t)4 * rz_lds_floats(g, R, a.T, wlds, tables);
  a.magic_hw = (uint32_t)((0x100000000ull + (uint64_t)g.HW - 1) / (uint64_t)g.HW);
  a.magic_w = (uint32_t)((0x100000000ull + (uint64_t)g.W - 1) / (uint64_t)g.W);
  {
    const uint64_t pt = (uint64_t)R.in_channels * (uint64_t)g.HW;
    a.magic_pt = pt > 1 ? (uint32_t)((0x100000000ull + pt - 1) / pt) : 0u;
  }
  if (rz_env_int("MZX_RZ_WLDS", 1) == 0 && wlds) {   // A/B knob: weights from L2 although they would fit in LDS
    wlds = false;
    L.lds = (size_t)4 * rz_lds_floats(g, R, a.T, false, tables);
  }
  L.wlds = wlds;
  // at most three row tiles in any layer GEMM: the kernels without the 4..8-tile code (4-wave kernels only)
  L.small = !L.eight && (a.T * g.HW + 15) / 16 <= 3 && rz_env_int("MZX_RZ_SMALL", 1) != 0;
  return L;
}

// Launches the fused part of a program.  `in` = the tensor feeding it (observation, parent hidden
// state or the stem's output).
inline int rz_launch(const mzx_net* net, const RzProgram& R, const float* in, const NetBuffers& nb, int batch,
                     const NetIndex* ix, stream_t stream, int dump_op = -1, float* dump = nullptr) {
  RzLaunch L = rz_prepare(net, R, in, nb, batch, ix);
  L.a.dump_op = dump_op;
  L.a.dump = dump;
  if (L.small) return L.wlds ? rz_launch_k<true, 4, 3>(L.a, L.grid, L.lds, stream) : rz_launch_k<false, 4, 3>(L.a, L.grid, L.lds, stream);
  if (L.wlds) return L.eight ? rz_launch_k<true, 8, 8>(L.a, L.grid, L.lds, stream) : rz_launch_k<true, 4, 8>(L.a, L.grid, L.lds, stream);
  return L.eight ? rz_launch_k<false, 8, 8>(L.a, L.grid, L.lds, stream) : rz_launch_k<false, 4, 8>(L.a, L.grid, L.lds, stream);
}

// ---------------------------------------------------------------------------
// Down-sampling stem (DownSample, models.py:233-275): one 3x3 convolution (stride 1 or 2, optional folded
// BatchNorm / residual / ReLU) on LARGE feature maps as an MFMA implicit GEMM.  A workgroup computes a
// TH x TW tile of output positions for every output channel: the input tile with its halo is gathered
// from the NCHW tensor into the position-major LDS layout (out-of-image cells zero), the K loop and
// epilogue are rz_gemm_tiles, the result is staged in LDS and written back coalesced (NCHW).
struct RzStemArgs {
  RzOp op;                // one RZ_GEMM (rows = tile positions), in_off / out_off / res_off = LDS float offsets
  const float* x;         // [batch][cin][hin][win]
  const float* res;       // [batch][cout][hout][wout] or null
  float* y;               // [batch][cout][hout][wout]
  const float* weights;   // packed B fragments
  const float* alpha;     // folded BatchNorm (null: none)
  const float* beta;
  int32_t cin, cout, hin, win, hout, wout, stride;
  int32_t TH, TW, tiles_x, PWin, PHin, Cs, mpad;
  uint32_t magic_pwin, magic_tw, magic_cells, magic_rows;   // ceil(2^32 / d): divisions by multiplication (rz_div)
};

// Occupancy: the kernel is a chain of memory phases (tile gather, residual gather, write-back) around one GEMM
// phase, separated by workgroup barriers -- alone on a CU it exposes every one of those latencies (measured
// ~21 us per 16 x 16 tile at 48 x 48 x 8, 89 % of a breakout Reanalyse pass).  Registers are capped so that two or
// three workgroups share a CU (their LDS tiles are sized for it, rz_stem_launch) and overlap each other's phases.
static __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) rz_stem_conv_kernel(const RzStemArgs sa) {
  extern __shared__ __attribute__((aligned(16))) float rz_lds[];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x, b = blockIdx.y;
  const int ty0 = (tile / sa.tiles_x) * sa.TH, tx0 = (tile % sa.tiles_x) * sa.TW;
  const int rows = sa.TH * sa.TW;
  int* rowaddr = (int*)rz_lds;
  int* rowtp = rowaddr + sa.mpad;
  int* rowout = rowtp + sa.mpad;
  float* params = (float*)(rowout + sa.mpad);          // alpha[64], beta[64], A-fragment offset table
  const int aoff_ints = rz_aoff_ints(sa.op.nchunks);
  float* reg = params + 128 + aoff_ints;
  const int in_floats = sa.PHin * sa.PWin * sa.Cs;
  // ---- tables, parameters, input tile (+ halo), residual tile
  for (int m = tid; m < sa.mpad; m += 256) {
    const int ty = m / sa.TW, tx = m - ty * sa.TW;
    const bool v = m < rows;
    rowaddr[m] = v ? ((ty * sa.stride + 1) * sa.PWin + tx * sa.stride + 1) * sa.Cs : (sa.PWin + 1) * sa.Cs;
    rowout[m] = v ? m * sa.Cs : 0;
    rowtp[m] = v ? m : 0;
  }
  if (tid < 128) params[tid] = (tid < 64) ? ((sa.alpha && tid < sa.cout) ? sa.alpha[tid] : 1.f)
                                          : ((sa.beta && tid - 64 < sa.cout) ? sa.beta[tid - 64] : 0.f);
  for (int k = tid; k < aoff_ints; k += 256)
    ((int*)params)[128 + k] = rz_aoff_entry(k, sa.op.nchunks, 9, sa.op.cchunks, sa.PWin, sa.Cs);
  {
    // only the input tile needs zeros (halo cells outside the image, pad channels of the last K chunk); every
    // output / residual element that is read back is written first
    f32x4* z = (f32x4*)reg;
    for (int i = tid; i < in_floats / 4; i += 256) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  {
    // LDS cell (iy, ix) of the haloed tile = image pixel (ty0 * stride - 1 + iy, tx0 * stride - 1 + ix).
    // (channel, cell) pairs are spread over the workgroup with four global loads in flight per thread: the
    // gather costs a few memory round trips, not one per channel and pass.
    const float* xb = sa.x + (size_t)b * sa.cin * sa.hin * sa.win;
    const int gy0 = ty0 * sa.stride - 1, gx0 = tx0 * sa.stride - 1;
    const int cells = sa.PHin * sa.PWin, total = sa.cin * cells;
    for (int i0 = tid; i0 < total; i0 += 4 * 256) {
      float v[4];
      int at[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * 256;
        at[u] = -1; v[u] = 0.f;
        if (idx < total) {
          const int c = rz_div(idx, cells, sa.magic_cells), i = idx - c * cells;
          const int iy = rz_div(i, sa.PWin, sa.magic_pwin), ix = i - iy * sa.PWin;
          const int gy = gy0 + iy, gx = gx0 + ix;
          if (gy >= 0 && gy < sa.hin && gx >= 0 && gx < sa.win) {
            v[u] = xb[((size_t)c * sa.hin + gy) * sa.win + gx];
            at[u] = i * sa.Cs + c;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (at[u] >= 0) reg[at[u]] = v[u];
    }
    if (sa.res) {   // the residual tile lands in the OUTPUT region: the epilogue adds and overwrites it element by element
      float* rr = reg + in_floats;
      const float* rb = sa.res + (size_t)b * sa.cout * sa.hout * sa.wout;
      const int rtotal = sa.cout * rows;
      for (int i0 = tid; i0 < rtotal; i0 += 4 * 256) {
        float v[4];
        int at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = i0 + u * 256;
          at[u] = -1; v[u] = 0.f;
          if (idx < rtotal) {
            const int c = rz_div(idx, rows, sa.magic_rows), m = idx - c * rows;
            const int ty = rz_div(m, sa.TW, sa.magic_tw), tx = m - ty * sa.TW;
            const int oy = ty0 + ty, ox = tx0 + tx;
            at[u] = m * sa.Cs + c;
            if (oy < sa.hout && ox < sa.wout) v[u] = rb[((size_t)c * sa.hout + oy) * sa.wout + ox];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (at[u] >= 0) rr[at[u]] = v[u];
      }
    }
  }
  __syncthreads();
  // ---- the GEMM: rows = tile positions, K = 9 taps x cin, N = cout
  RzArgs a;
  a.T = 1; a.HW = rows; a.PW = sa.PWin; a.Cs = sa.Cs; a.weights = sa.weights;
  RzCtx cx;
  cx.reg = reg; cx.rowaddr = rowaddr; cx.rowtp = rowtp; cx.rowout = rowout; cx.scratch = params; cx.simg = params;
  cx.wlds = nullptr; cx.T = 1; cx.tid = tid; cx.lane = tid & 63; cx.wave = tid >> 6; cx.fine = nullptr;
  cx.work = nullptr;
  rz_gemm<false, 4, 4>(sa.op, a, cx, rz_work_word(sa.op, 1, rows, 4, tid >> 6));   // at most 256 rows = 4 row tiles per wave
  __syncthreads();
  // ---- write back, coalesced along x
  {
    const float* out = reg + in_floats;
    float* yb = sa.y + (size_t)b * sa.cout * sa.hout * sa.wout;
    const int total = sa.cout * rows;
    for (int idx = tid; idx < total; idx += 256) {
      const int c = rz_div(idx, rows, sa.magic_rows), m = idx - c * rows;
      const int ty = rz_div(m, sa.TW, sa.magic_tw), tx = m - ty * sa.TW;
      const int oy = ty0 + ty, ox = tx0 + tx;
      if (oy < sa.hout && ox < sa.wout) yb[((size_t)c * sa.hout + oy) * sa.wout + ox] = out[m * sa.Cs + c];
    }
  }
}

// Launches one stem convolution; returns MZX_ERR_INVALID if the shape does not fit the kernel's LDS tiling
// (the caller then uses the per-operator kernel).
inline int rz_stem_launch(const mzx_net* net, const RzStemConv& sc, const OpDesc& d, const float* x, const float* res,
                          float* y, int batch, stream_t stream) {
  static std::atomic<uint64_t> lds_attr_done{0};   // per instantiation, one bit per device
  if (const int ae = allow_large_lds((const void*)rz_stem_conv_kernel, 160 * 1024, lds_attr_done)) {
    set_error("hipFuncSetAttribute: %s", runtime_error_string(ae));
    return MZX_ERR_RUNTIME;
  }
  RzStemArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.cin = d.cin; sa.cout = d.cout; sa.hin = d.hin; sa.win = d.win; sa.hout = d.hout; sa.wout = d.wout; sa.stride = d.stride;
  // floats per position: whole 16-channel K chunks + 4.  A row stride of 16 c + 4 words keeps the 16-byte reads of
  // 16 consecutive rows on distinct banks (20 k mod 64, k = 0..15, are 16 disjoint 4-word ranges) at 5/6 of the
  // LDS of the engine's 16 c + 8 layout: three workgroups per CU fit at the 8- and 16-channel stages
  sa.Cs = rz_round16(std::max(d.cin, d.cout)) + 4;
  // output tile: up to 256 positions (16 MFMA row tiles over 4 waves), halo'ed input tile beside it in LDS
  sa.TW = std::min(16, d.wout);
  sa.TH = std::min(256 / sa.TW, d.hout);
  for (;;) {
    sa.PHin = (sa.TH - 1) * d.stride + 3;
    sa.PWin = (sa.TW - 1) * d.stride + 3;
    sa.mpad = rz_round16(sa.TH * sa.TW);
    const int64_t floats = 3 * (int64_t)sa.mpad + 128 + rz_aoff_ints(sc.nchunks) + (int64_t)sa.PHin * sa.PWin * sa.Cs +
                           (int64_t)sa.TH * sa.TW * sa.Cs;
    if (4 * floats <= RZ_LDS_BUDGET) break;
    if (sa.TH <= 1) return MZX_ERR_INVALID;
    sa.TH = (sa.TH + 1) / 2;
  }
  sa.tiles_x = (d.wout + sa.TW - 1) / sa.TW;
  const int tiles_y = (d.hout + sa.TH - 1) / sa.TH;
  RzOp& o = sa.op;
  o.kind = RZ_GEMM; o.rows = RZ_ROWS_POS;
  o.taps = 9 | (int32_t)((((1u << 20) + (uint32_t)sc.cchunks - 1) / (uint32_t)sc.cchunks) << 8);
  o.in_off = 0; o.in_tstride = 0;
  o.out_off = sa.PHin * sa.PWin * sa.Cs; o.out_tstride = 0; o.out_layout = RZ_OUT_PADDED;
  o.res_off = res ? o.out_off : -1;   // in place: the residual tile is loaded into the output region
  o.cchunks = sc.cchunks; o.cout = d.cout; o.nchunks = sc.nchunks; o.wchunks = sc.wchunks; o.aoff_off = 128;
  o.w_off = 0;
  o.alpha_off = d.bn.channels ? 0 : -1; o.beta_off = d.bn.channels ? 64 : -1; o.bias_off = -1; o.asum_off = -1;
  o.act = d.relu ? RZ_ACT_RELU : RZ_ACT_NONE;
  sa.x = x; sa.res = res; sa.y = y;
  sa.weights = net->d_derived + sc.w_off;
  sa.alpha = d.bn.channels ? net->d_derived + d.bn.alpha : nullptr;
  sa.beta = d.bn.channels ? net->d_derived + d.bn.beta : nullptr;
  const size_t lds = 4 * (size_t)(3 * sa.mpad + 128 + rz_aoff_ints(sc.nchunks) + sa.PHin * sa.PWin * sa.Cs + sa.TH * sa.TW * sa.Cs);
  auto magic = [](int d) { return d > 1 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u; };
  sa.magic_pwin = magic(sa.PWin); sa.magic_tw = magic(sa.TW);
  sa.magic_cells = magic(sa.PHin * sa.PWin); sa.magic_rows = magic(sa.TH * sa.TW);
  hipLaunchKernelGGL(rz_stem_conv_kernel, dim3(sa.tiles_x * tiles_y, batch), dim3(256), lds, stream, sa);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("stem convolution launch failed: %s", hipGetErrorString(e)); return MZX_ERR_RUNTIME; }
  return MZX_OK;
}

// The operators [0, count) of initial_inference that precede the fused part: 3x3 convolutions on the MFMA
// stem kernel, everything else (average pooling) on the per-operator kernels.
inline int rz_run_stem(const mzx_net* net, const std::vector<OpDesc>& prog, int count, const NetBuffers& nb, int batch,
                       stream_t stream) {
  // Round 3: the stem's convolutions run on the streamed engine's implicit-GEMM kernel (position-major activations,
  // 16-byte staging and stores, no residual / write-back phases behind barriers) when the whole stem is planned
  // there and ends in a pooling, which then writes the NCHW tensor the LDS-resident engine gathers (MZX_RZ_STEM=rz:
  // the round-1 stem kernel, A/B).
  static const bool rb_stem = exp_int("MZX_RZ_STEM_RZ", 0) == 0;      // (1: the LDS-resident engine's own stem kernel, the A/B)
  if (rb_stem && net->rz_mode == 1 && &prog == &net->prog_initial && net->rb.ok && net->rb.initial.ok && count > 0 &&
      prog[count - 1].kind == OP_POOL) {
    bool all = true;
    for (int i = 0; i < count; ++i) all = all && net->rb.initial.ops[i].kind != RB_FUNCTOR;
    if (all) return rb_run_program(net, false, nb, batch, stream, nullptr, count, nullptr, true);
  }
  for (int i = 0; i < count; ++i) {
    const OpDesc& d = prog[i];
    const RzStemConv* sc = nullptr;
    if (net->rz_mode == 1) for (const RzStemConv& c : net->rz.stem) if (c.op_index == i) sc = &c;
    int rc = MZX_ERR_INVALID;
    if (sc && d.kind == OP_CONV3) {
      const float* res = (d.res == -100) ? nullptr : resolve(net, nb, d.res, batch);
      rc = rz_stem_launch(net, *sc, d, resolve(net, nb, d.in, batch), res, resolve(net, nb, d.out, batch), batch, stream);
      if (rc == MZX_ERR_RUNTIME) return rc;
    }
    if (rc != MZX_OK) {
      const std::vector<OpDesc> one(prog.begin() + i, prog.begin() + i + 1);
      rc = run_program(net, one, nb, batch, stream);
      if (rc) return rc;
    }
  }
  return MZX_OK;
}

#endif  // !MZX_HOSTCHECK


// True when inference `recurrent` of `net` runs on the fused engine.
inline bool rz_enabled(const mzx_net* net, bool recurrent) {
#ifdef MZX_HOSTCHECK
  (void)net; (void)recurrent;
  return false;
#else
  if (!net->rz.ok || !net->rz_mode) return false;
  if (net->rb_force && net->rb.ok && (recurrent ? net->rb.recurrent.ok : net->rb.initial.ok)) return false;
  return recurrent ? net->rz.recurrent.ok != 0 : net->rz.initial.ok != 0;
#endif
}

// initial_inference / recurrent_inference of any network: the fused engine where it applies
// (preceded by the per-operator kernels of a down-sampling stem), else one kernel per operator.
// With `ix`, sample b reads hidden-state node ix->in_node[b] of nb.in ([batch][ix->in_nodes][..])
// and writes node ix->out_node[b] of nb.hidden ([batch][ix->out_nodes][hidden]); only the fused
// engine implements that, callers check rz_enabled first.
inline int run_network(const mzx_net* net, bool recurrent, const NetBuffers& nb, int batch, stream_t stream,
                       const NetIndex* ix = nullptr) {
  const std::vector<OpDesc>& prog = recurrent ? net->prog_recurrent : net->prog_initial;
#ifndef MZX_HOSTCHECK
  if (rz_enabled(net, recurrent)) {
    const RzProgram& R = recurrent ? net->rz.recurrent : net->rz.initial;
    const float* in = nb.in;
    if (R.first > 0) {
      const int rc = rz_run_stem(net, prog, R.first, nb, batch, stream);
      if (rc) return rc;
      in = resolve(net, nb, R.ext_buf, batch);
    }
    return rz_launch(net, R, in, nb, batch, ix, stream);
  }
  if (rb_enabled(net, recurrent)) return rb_run_program(net, recurrent, nb, batch, stream, ix);
#endif
  if (ix && (ix->in_nodes != 1 || ix->out_nodes != 1)) { set_error("indexed inference needs an MFMA engine"); return MZX_ERR_INVALID; }
  return run_program(net, prog, nb, batch, stream);
}

// Diagnostics: run the first n_ops operators of a program on either engine and copy the output
// tensor of the last one (dense per sample) to d_out.
inline int run_network_prefix(const mzx_net* net, bool recurrent, int fused, int n_ops, const NetBuffers& nb, int batch,
                              float* d_out, int64_t out_floats, stream_t stream) {
  const std::vector<OpDesc>& prog = recurrent ? net->prog_recurrent : net->prog_initial;
  if (n_ops < 1 || n_ops > (int)prog.size()) { set_error("n_ops out of range"); return MZX_ERR_INVALID; }
  const OpDesc& last = prog[n_ops - 1];
  int64_t per = 0;
  switch (last.kind) {
    case OP_LINEAR: per = last.out_features; break;
    case OP_CONV3: case OP_POOL: case OP_CONVK: case OP_MAXPOOL: case OP_ADAPTIVE_POOL:
      per = (int64_t)last.cout * last.hout * last.wout; break;
    case OP_CONV1: per = (int64_t)last.cout * last.hin; break;
    default: per = (int64_t)last.groups_per_sample * last.len; break;
  }
  if (fused == 2) per = 0;
  if (out_floats < per * batch) { set_error("prefix output buffer too small (%lld floats per sample)", (long long)per); return MZX_ERR_WORKSPACE; }
#ifndef MZX_HOSTCHECK
  if (fused && rb_enabled(net, recurrent)) {   // streamed engine: output of operator n_ops - 1 in the per-operator layout
    if (fused == 2) { set_error("no cycle profile on the streamed engine"); return MZX_ERR_INVALID; }
    return rb_run_program(net, recurrent, nb, batch, stream, nullptr, n_ops, d_out);
  }
  if (fused) {
    const RzProgram& R = recurrent ? net->rz.recurrent : net->rz.initial;
    if (!net->rz.ok || !R.ok) { set_error("fused engine does not cover this operator"); return MZX_ERR_INVALID; }
    if (n_ops <= R.first) {  // an operator of the down-sampling stem (MFMA stem kernel / per-operator pooling)
      const int rc = rz_run_stem(net, prog, n_ops, nb, batch, stream);
      if (rc) return rc;
      return copy_d2d(d_out, resolve(net, nb, last.out, batch), sizeof(float) * per * batch, stream) ? MZX_ERR_RUNTIME : MZX_OK;
    }
    const float* in = nb.in;
    if (R.first > 0) {
      const int rc = rz_run_stem(net, prog, R.first, nb, batch, stream);
      if (rc) return rc;
      in = resolve(net, nb, R.ext_buf, batch);
    }
    if (fused == 2) {  // cycle profile of the whole fused program: stamps (uint64) land in d_out
      if (out_floats < 2 * RZ_STAMP_WORDS) { set_error("profile buffer too small"); return MZX_ERR_WORKSPACE; }
      return rz_launch(net, R, in, nb, batch, nullptr, stream, -2, d_out);
    }
    return rz_launch(net, R, in, nb, batch, nullptr, stream, R.order[n_ops - 1 - R.first], d_out);
  }
#else
  if (fused) { set_error("no fused engine in this build"); return MZX_ERR_INVALID; }
#endif
  const std::vector<OpDesc> head(prog.begin(), prog.begin() + n_ops);
  const int rc = run_program(net, head, nb, batch, stream);
  if (rc) return rc;
  const float* src = resolve(net, nb, last.out, batch);
  return copy_d2d(d_out, src, sizeof(float) * per * batch, stream) ? MZX_ERR_RUNTIME : MZX_OK;
}

}  // namespace mzx
