// mzx_resnet_batched.h -- host side of the STREAMED residual-network engine: planner and entry points.
//
// Which networks come here: every MuZeroResidualNetwork (models.py:436-623) whose inference the LDS-resident
// engine (mzx_resnet_fused.h) cannot hold -- the reference's games/gomoku.py (128 ch x 6 blocks, 11 x 11) and
// games/atari.py (256 ch x 16 blocks, "resnet" down-sampling stem, 256-wide heads) as shipped.  Each operator of
// the program built by mzx_net.h becomes one launch over the whole batch:
//   conv3x3 (:206-209, stride 1 | 2, folded BatchNorm, residual, ReLU, action plane)   rb_gemm_kernel, 9 taps
//   conv1x1 head (:369-389, :404-433), Linear (:630-642)                               rb_gemm_kernel, 1 tap
//   DownsampleCNN convolutions (:281-290: K x K stride 4 / 5 x 5, bias, ReLU)          rb_gemm_kernel, K * K taps
//   per-plane min-max scaling (:527-553, :574-599)                                     rb_scale_kernel
//   AvgPool2d of the stem (:233-275), MaxPool2d / AdaptiveAvgPool2d (:286-291)         element kernels, NHWC
// Activations between operators are position-major ([sample][y][x][channel]) so that a workgroup's input patch is
// staged into LDS with 16-byte copies; observations and hidden states keep the reference's NCHW layout at the
// boundary (the first convolution gathers, the scaling operator writes NCHW -- straight into / out of the search
// arena's per-node store, NetIndex).
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <tuple>

#include "mzx_net.h"

namespace mzx {

inline int rb_round16(int x) { return (x + 15) & ~15; }

// LDS bytes of a workgroup: row tables (3 ints per padded row), per-sample offsets (2 x int64), the staged patch
inline int64_t rb_lds_bytes(int T, int mtiles, int cells, int Cs) {
  return (int64_t)3 * 16 * mtiles * 4 + (int64_t)2 * ((T + 1) & ~1) * 8 + (int64_t)cells * Cs * 4;
}

// Workgroup tile of a GEMM operator: T whole samples per workgroup when a sample's output positions fit RB_MAX_ROWS
// rows, else one th x tw patch of one sample; the input patch (with the 3x3 halo) is staged in `phases` channel
// groups of `cpg` 16-channel chunks.  Maximises MFMA row-tile occupancy, then prefers fewer phases and more rows
// (every workgroup streams the whole B matrix once: traffic per flop ~ 1 / rows).
inline int rb_lds_budget() {   // MZX_RB_LDS: A/B knob (bytes), e.g. 159744 = one workgroup per CU
  static const int v = [] {
    const int x = exp_int("MZX_RB_LDS", RB_LDS_BUDGET);
    return std::min(std::max(x, 16 * 1024), RB_LDS_MAX);
  }();
  return v;
}

// Rows a workgroup can own: nine row tiles per wave, and the waves the column tiles leave over split the rows -- a
// 64-channel layer has four column tiles -> two waves deep in rows -> 288 rows (connect4: six boards per workgroup
// instead of three, in two channel groups: the per-workgroup costs -- tables, first weight fetch, barriers, pipeline
// fill -- are paid half as often; 0.615 -> 0.65 of the MFMA peak for the whole step at 9216 trees).
// MZX_RB_ROWS_WIDE=0: nine row tiles per workgroup whatever the width (A/B).
inline int rb_max_rows(const RbOp& o) {
  static const int wide = exp_int("MZX_RB_ROWS_WIDE", 1);
  if (!wide || o.taps == 1 || o.ntiles < 2) return RB_MAX_ROWS;      // trunk convolutions only: head layers are tiny
  const int nt = o.ntiles > 8 ? 2 : 1, wn = std::min(8, (std::min(o.ntiles, 16) + nt - 1) / nt);
  return RB_MAX_ROWS * std::min(4, std::max(1, 8 / wn));
}

inline bool rb_choose_tile(RbOp& o) {
  const int HWo = o.hout * o.wout;
  const int max_rows = rb_max_rows(o);
  double best = -1.0;
  RbOp pick = o;
  auto consider = [&](int T, int th, int tw) {
    const int PH = (th - 1) * o.stride + o.ksize, PW = (tw - 1) * o.stride + o.ksize;   // the windows' extent
    const int cells = T * PH * PW;
    const int rows = T * th * tw, mtiles = (rows + 15) / 16;
    if (rows > max_rows) return;
    int cpg = o.cchunks;
    while (cpg >= 1 && rb_lds_bytes(T, mtiles, cells, 16 * cpg + 8) > rb_lds_budget()) --cpg;
    if (cpg < 1) return;
    const int phases = (o.cchunks + cpg - 1) / cpg;
    cpg = (o.cchunks + phases - 1) / phases;
    const int tiles_x = (o.wout + tw - 1) / tw, tiles_y = (o.hout + th - 1) / th;
    const double eff = (double)(T * HWo) / ((double)tiles_x * tiles_y * mtiles * 16);
    const double score = eff * (1.0 - 0.02 * (phases - 1)) * ((double)rows / (rows + 8.0)) -
                         1e-7 * (double)cells * phases;   // ties: the smaller staged patch
    if (score > best) {
      best = score;
      pick.T = T; pick.th = th; pick.tw = tw; pick.tiles_x = tiles_x; pick.tiles_y = tiles_y;
      pick.PH = PH; pick.PW = PW; pick.cpg = cpg; pick.phases = phases; pick.Cs = 16 * cpg + 8;
      pick.rows = rows; pick.mtiles = mtiles;
      pick.lds_bytes = (int32_t)rb_lds_bytes(T, mtiles, cells, pick.Cs);
    }
  };
  if (HWo <= max_rows) {
    for (int T = 1; T * HWo <= max_rows; ++T) consider(T, o.hout, o.wout);
  } else {
    for (int tw = 1; tw <= std::min(o.wout, max_rows); ++tw)
      for (int th = 1; th <= o.hout && th * tw <= max_rows; ++th) consider(1, th, tw);
  }
  if (best < 0.0) return false;
  o = pick;
  return true;
}

// Launch shape of one GEMM operator for THIS batch.  The plan fixes the largest tile (T samples, all column tiles in
// one workgroup); a small batch would leave most of the chip idle with it, so fewer samples per workgroup and / or a
// split of the column tiles over more workgroups are considered, by a cost model in units of "accumulator tiles on the
// busiest wave": rounds of co-resident workgroups x (row tiles x column tiles of that wave), plus a small charge per
// workgroup for staging its patch again.  Large batches keep the planned tile (fewer, larger workgroups win ties).
struct RbShape { int T, rows, mtiles, lds, ntiles_wg, nsplit, NT, WN, WM, MT, groups, cpg, phases, Cs, rowsplit, splits, PH; };

inline RbShape rb_choose_shape(const RbOp& o, int batch) {
  static const int fixed = exp_int("MZX_RB_SHAPE", 0);   // 1: always the planned tile (A/B)
  RbShape best{};
  double best_cost = 1e30;
  const int spatial = o.tiles_x * o.tiles_y;
  const int t_lo = (spatial == 1 && !fixed) ? 1 : o.T;
  for (int T = o.T; T >= t_lo; --T) {
    RbShape c;
    c.T = T;
    c.rows = T * o.th * o.tw;
    c.mtiles = (c.rows + 15) / 16;
    c.lds = (int)rb_lds_bytes(T, c.mtiles, T * o.PH * o.PW, o.Cs);
    c.groups = ((batch + T - 1) / T) * spatial;
    for (int ntiles_wg = std::min(o.ntiles, 16);; ntiles_wg = (ntiles_wg + 1) / 2) {
      c.ntiles_wg = ntiles_wg;
      c.nsplit = (o.ntiles + ntiles_wg - 1) / ntiles_wg;
      // two column tiles per wave when the workgroup has more than eight -- or, MZX_RB_NT=2 (experiment), whenever it has
      // an even number: half the A-fragment LDS reads per MFMA, the waves split the row tiles instead
      static const int want_nt = exp_int("MZX_RB_NT", 0);
      c.NT = (ntiles_wg > 8 || (want_nt == 2 && ntiles_wg >= 2 && ntiles_wg % 2 == 0)) ? 2 : 1;
      c.WN = std::min(8, (ntiles_wg + c.NT - 1) / c.NT);
      c.WM = std::max(1, std::min(8 / c.WN, c.mtiles));
      c.MT = (c.mtiles + c.WM - 1) / c.WM;
      const int per_cu = (c.lds <= RB_LDS_BUDGET && c.MT * c.NT <= 8) ? 2 : 1;   // 128-register instantiations
      const int64_t wgs = (int64_t)c.groups * c.nsplit, cap = 256 * per_cu;
      // a full round of co-resident workgroups shares the matrix pipes per_cu ways; the last, partial round as
      // many ways as it has workgroups per CU
      const int64_t rem = wgs % cap;
      const double ways = (double)(wgs / cap) * per_cu + (rem ? (double)std::min<int64_t>(per_cu, (rem + 255) / 256) : 0.0);
      const double cost = ways * (c.MT * c.NT) * (1.0 + 0.02 * (c.nsplit - 1)) + 1e-4 * (double)wgs;
      static const int force_ntw = exp_int("MZX_RB_NTILES_WG", 0);   // experiment knob
      if (force_ntw > 0 && T == o.T && ntiles_wg == std::min(force_ntw, std::min(o.ntiles, 16))) { best_cost = -1.0; best = c; }
      if (cost < best_cost) { best_cost = cost; best = c; }
      if (ntiles_wg == 1 || fixed) break;
    }
  }
  // Row-range tiles (stride-1 operators on whole samples, one sample per workgroup): a sample's row tiles are dealt to
  // `splits` workgroups, each staging only the board rows its positions touch.  More, shorter-lived workgroups per CU
  // keep two of them overlapping to the end of the launch (with ONE round of two the younger one, which loses the
  // matrix pipes to the older by age, runs alone for the last quarter: measured 115 / 165 us for the two of a CU).
  best.rowsplit = 0; best.splits = 1; best.PH = o.PH;
  static const int want_splits = exp_int("MZX_RB_ROWSPLIT", 0);
  if (!fixed && want_splits > 1 && o.tiles_x * o.tiles_y == 1 && best.T == 1 && o.stride == 1 && o.ksize == 2 * o.pad + 1 &&
      best.mtiles >= 2 * want_splits) {
    const int per = (best.mtiles + want_splits - 1) / want_splits;
    const int splits = (best.mtiles + per - 1) / per;
    const int pad = o.pad, HWo = o.hout * o.wout;
    int PH = 1;
    for (int sp = 0; sp < splits; ++sp) {
      const int p0 = sp * per * 16, p1 = std::min(p0 + per * 16, HWo) - 1;
      PH = std::max(PH, p1 / o.wout - p0 / o.wout + 1 + 2 * pad);
    }
    best.rowsplit = per; best.splits = splits; best.PH = PH;
    best.mtiles = per; best.rows = per * 16;
    best.groups *= splits;
    best.WM = std::max(1, std::min(8 / best.WN, best.mtiles));
    best.MT = (best.mtiles + best.WM - 1) / best.WM;
    best.lds = (int)rb_lds_bytes(1, best.mtiles, PH * o.PW, o.Cs);
  }
  // Channel phases: the plan splits the patch so that TWO workgroups fit a CU.  A launch with at most one workgroup
  // per CU has no partner to overlap its staging with: it takes the whole LDS and as few phases as fit.
  best.cpg = o.cpg; best.phases = o.phases; best.Cs = o.Cs;
  if (!fixed && (int64_t)best.groups * best.nsplit <= 256 && o.phases > 1) {
    const int cells = best.T * best.PH * o.PW;
    int cpg = o.cchunks;
    while (cpg > o.cpg && rb_lds_bytes(best.T, best.mtiles, cells, 16 * cpg + 8) > RB_LDS_MAX) --cpg;
    const int phases = (o.cchunks + cpg - 1) / cpg;
    cpg = (o.cchunks + phases - 1) / phases;
    if (phases < o.phases) {
      best.cpg = cpg; best.phases = phases; best.Cs = 16 * cpg + 8;
      best.lds = (int)rb_lds_bytes(best.T, best.mtiles, cells, best.Cs);
    }
  }
  return best;
}

// ---- towers (rb_tower_kernel): runs of same-width stride-1 3x3 convolutions, one launch, activations in LDS in place.
// Wave grid of a tower workgroup (512 threads): column tiles over WN waves (NT = 2 per wave above eight tiles), the
// remaining 8 / WN waves deep in rows, at most nine row tiles per wave.
struct RbTowerShape { int T, rows, mtiles, lds, NT, WN, WM, MT, groups, Cs, per_cu; };

// row tables, per-sample offsets, the tile, and the per-plane (min, range) pairs of a scaling operator in the tail
inline int64_t rb_tower_lds_bytes(int T, int mtiles, int cells, int Cs) {
  return (int64_t)3 * 16 * mtiles * 4 + (int64_t)((T + 1) & ~1) * 8 + (int64_t)cells * Cs * 4 + (int64_t)2 * T * (Cs - 8) * 4;
}

inline bool rb_tower_grid(const RbTower& tw, int T, RbTowerShape& c) {
  if (tw.ntiles > 16) return false;
  c.T = T;
  c.rows = T * tw.H * tw.W;
  c.mtiles = (c.rows + 15) / 16;
  c.NT = tw.ntiles > 8 ? 2 : 1;
  c.WN = std::min(8, (tw.ntiles + c.NT - 1) / c.NT);
  c.WM = std::max(1, std::min(8 / c.WN, c.mtiles));
  c.MT = (c.mtiles + c.WM - 1) / c.WM;
  c.Cs = 16 * tw.cchunks + 8;
  c.lds = (int)rb_tower_lds_bytes(T, c.mtiles, T * (tw.H + 2) * (tw.W + 2), c.Cs);
  // two workgroups per CU: half the LDS each and the 128-register instantiations (accumulators + saved residual of at
  // most four tiles per wave)
  c.per_cu = (c.lds <= RB_LDS_BUDGET && c.MT * c.NT <= 4) ? 2 : 1;
  // (the tower kernel keeps a saved residual beside the accumulators: the 2-column tilings compile without scratch up
  // to six row tiles per wave, ISA checked)
  return c.MT <= (c.NT == 2 ? 6 : RB_MT) && rb_tower_lds_bytes(T, c.mtiles, T * (tw.H + 2) * (tw.W + 2), c.Cs) <= RB_LDS_MAX;
}

// Samples per workgroup for THIS batch: the cost model of rb_choose_shape (accumulator tiles on the busiest wave x rounds
// of co-resident workgroups; ties: fewer, larger workgroups).  Calibrated on connect4 (profiles/r04_tower_experiments.txt:
// samples per workgroup 2 .. 6 at 1024 / 2048 / 3072 / 4608 trees): the model ranks the measured rates correctly in all
// four cases -- whole rounds of workgroups matter more than anything else (4608 trees: six boards = 768 workgroups =
// three rounds of 256 -> 0.663; four boards = 4.5 rounds -> 0.545).  MZX_RB_TOWER_T=<n>: force (A/B).
inline RbTowerShape rb_tower_shape(const RbTower& tw, int batch) {
  const int force_t = tune(TUNE_RB_TOWER_T);             // > 0: forced (tests, A/B)
  RbTowerShape best{};
  double best_cost = 1e30;
  for (int T = tw.t_max; T >= 1; --T) {
    RbTowerShape c;
    if (!rb_tower_grid(tw, T, c)) continue;
    c.groups = (batch + T - 1) / T;
    const int64_t wgs = c.groups, cap = 256 * c.per_cu;
    const int64_t rem = wgs % cap;
    const double ways = (double)(wgs / cap) * c.per_cu + (rem ? (double)std::min<int64_t>(c.per_cu, (rem + 255) / 256) : 0.0);
    const double cost = ways * (c.MT * c.NT) + 1e-4 * (double)wgs;
    if (force_t > 0 && T == std::min(force_t, tw.t_max)) { best = c; break; }
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// Does the tower run as one launch?  A property of the NETWORK, not of the batch (a tower and the layer kernels
// sum a convolution in different orders -- the layer kernel stages its channels in groups --, and the LDS-resident engine in a
// third; whichever runs must not depend on how many trees share the launch, or a tree's result would depend on its shard
// and the two slot groups of a pipelined self-play shard would play other games than the undivided shard).  Usable: SOME
// samples-per-workgroup count the LDS admits wastes at most a fifth of the MFMA rows (connect4: 42 of 48 rows with one
// board; a 5 x 5 board: 75 of 80 with three; games/atari.py: 72 of 80 with two 6 x 6 samples).  The shape for a given batch
// is still rb_tower_shape's (a small batch may take fewer samples per workgroup to fill the chip: games/atari.py at 256
// trees runs one sample per workgroup at 36 of 48 rows, measured 0.58 of the peak against 0.62 layer by layer -- the price
// of one arithmetic per network).
inline bool rb_tower_rows_ok(const RbTowerShape& c) { return c.rows * 5 >= c.mtiles * 16 * 4; }

inline bool rb_tower_use(const RbTower& tw, int /*batch*/ = 0) {
  if (tw.t_max < 1) return false;
  if (tune(TUNE_RB_TOWER_T) > 0) return true;     // a forced shape (tests, A/B) is taken as given
  for (int T = 1; T <= tw.t_max; ++T) {
    RbTowerShape c;
    if (rb_tower_grid(tw, T, c) && rb_tower_rows_ok(c)) return true;
  }
  return false;
}

// Finds the towers of a planned program.  A layer joins the tower of its predecessor when it is a stride-1 3x3 GEMM of
// the tower's width on the tower's board that reads the predecessor's output, its residual (if any) is the
// predecessor's INPUT (the block input: ResidualBlock.forward, models.py:221-229), and the predecessor's output has no
// reader outside the run (it never reaches memory).  MZX_RB_TOWER=0 plans none (A/B).
inline void rb_find_towers(const std::vector<OpDesc>& prog, RbProgram& R) {
  R.towers.clear();
  for (RbOp& o : R.ops) { o.tower = -1; o.tower_of_tail = -1; }
  static const int enabled = exp_int("MZX_RB_TOWER", 1);
  if (!enabled) return;
  const int n = (int)prog.size();
  auto trunk_conv = [&](int k) {
    const OpDesc& d = prog[k];
    const RbOp& o = R.ops[k];
    return d.kind == OP_CONV3 && o.kind == RB_GEMM && o.stride == 1 && o.taps == 9 && o.hin == o.hout && o.win == o.wout &&
           o.cout <= 256 && o.out_layout == RB_NHWC;
  };
  // readers of the output of operator j before that buffer is written again must all lie in (j, last]
  auto private_output = [&](int j, int last) {
    const int id = prog[j].out;
    for (int m = j + 1; m < n; ++m) {
      if ((prog[m].in == id || prog[m].res == id) && m > last) return false;
      if (prog[m].out == id) break;
    }
    return true;
  };
  int k = 0;
  while (k < n) {
    if (!trunk_conv(k) || prog[k].res != -100) { ++k; continue; }   // (a tower starts at a layer without a residual)
    const RbOp& o0 = R.ops[k];
    int last = k;
    while (last + 1 < n && last + 1 - k < RB_TOWER_MAX_LAYERS) {
      const int j = last + 1;
      if (!trunk_conv(j)) break;
      const OpDesc& d = prog[j];
      const RbOp& o = R.ops[j];
      if (o.cin != o0.cout || o.cout != o0.cout || o.hin != o0.hin || o.win != o0.win || d.use_action) break;
      if (d.in != prog[j - 1].out) break;
      if (d.res != -100 && d.res != prog[j - 1].in) break;
      if (d.res != -100 && j - 1 == k && R.ops[k].cin != o0.cout) break;   // the first layer's input has another width
      last = j;
    }
    // cut the run where an inner output is needed elsewhere (to a fixed point: a shorter run moves readers outside)
    for (bool changed = true; changed;) {
      changed = false;
      for (int j = k; j < last; ++j)
        if (!private_output(j, last)) { last = j; changed = true; break; }
    }
    if (last - k + 1 >= 2) {
      RbTower tw;
      tw.first = k; tw.count = last - k + 1;
      tw.C = o0.cout; tw.H = o0.hout; tw.W = o0.wout;
      tw.cchunks = rb_round16(std::max(o0.cout, o0.cin)) / 16; tw.ntiles = (o0.cout + 15) / 16;
      tw.t_max = 0;
      for (int T = 1; T <= 64; ++T) {
        RbTowerShape c;
        if (rb_tower_grid(tw, T, c)) tw.t_max = T;
        else if (T * tw.H * tw.W > 16 * RB_MT * 8) break;
      }
      // the tail: operators directly behind the tower that read only its output -- at most one scaling operator and two
      // small 1x1 head convolutions, in program order
      for (int m = last + 1, scales = 0, convs = 0; m < n; ++m) {
        const OpDesc& dm = prog[m];
        const bool scale = dm.kind == OP_SCALE && R.ops[m].kind == RB_SCALE && dm.in == prog[last].out &&
                           dm.groups_per_sample == tw.C && dm.len == tw.H * tw.W;
        const bool conv1 = dm.kind == OP_CONV1 && R.ops[m].kind == RB_GEMM && dm.in == prog[last].out && dm.cin == tw.C &&
                           dm.hin == tw.H * tw.W && dm.cout <= RB_TAIL_MAX_R;
        if (scale && scales == 0) ++scales;
        else if (conv1 && convs < 2) ++convs;
        else break;
        ++tw.n_tail;
      }
      if (tw.t_max >= 1) {
        for (int j = k; j <= last; ++j) R.ops[j].tower = (int)R.towers.size();
        for (int j = last + 1; j <= last + tw.n_tail; ++j) R.ops[j].tower_of_tail = (int)R.towers.size();
        R.towers.push_back(tw);
      }
    }
    k = last + 1;
  }
}

// Finds the head chains of a program (after rb_find_towers): Linear chains fed by a tower's tail convolution.
inline void rb_find_heads(const std::vector<OpDesc>& prog, RbProgram& R) {
  R.heads = RbHeads();
  for (RbOp& o : R.ops) o.head_chain = -1;
  // (always planned; how a run launches the chains' layers is decided per call: tuning "rb_heads", mzx_batched_plan.h)
  const int n = (int)prog.size();
  for (const RbTower& tw : R.towers) {
    for (int m = tw.first + tw.count; m < tw.first + tw.count + tw.n_tail && m < n; ++m) {
      if (prog[m].kind != OP_CONV1 || R.heads.n_chains >= RB_HEADS_MAX_CHAINS) continue;
      const int src = prog[m].out;
      // the chain: the first Linear that reads the convolution's output, then every Linear reading its predecessor
      int first = -1;
      for (int q = m + 1; q < n; ++q) {
        if (prog[q].out == src) break;
        if (prog[q].in == src || prog[q].res == src) { first = q; break; }
      }
      if (first < 0 || prog[first].kind != OP_LINEAR || prog[first].use_action) continue;
      // the convolution's output must have no other reader
      bool sole = true;
      for (int q = m + 1; q < n && sole; ++q) {
        if (q != first && (prog[q].in == src || prog[q].res == src)) sole = false;
        if (prog[q].out == src) break;
      }
      if (!sole) continue;
      int count = 1;
      while (first + count < n && count < RB_HEADS_MAX_LAYERS && prog[first + count].kind == OP_LINEAR &&
             !prog[first + count].use_action && prog[first + count].in == prog[first + count - 1].out)
        ++count;
      // a longer chain than the kernel takes, or an inner output somebody else reads: leave it to the layer launches
      if (first + count < n && prog[first + count].kind == OP_LINEAR && prog[first + count].in == prog[first + count - 1].out) continue;
      bool ok = prog[first].in_features == prog[m].cout * prog[m].hin && prog[first].in_features <= RB_HEADS_MAX_IN;
      for (int q = first; q < first + count && ok; ++q) {
        ok = prog[q].out_features <= RB_HEADS_MAX_WIDTH && prog[q].w_stride == prog[q].in_features;
        if (q + 1 < first + count)      // inner outputs: read by the next layer only
          for (int z = q + 2; z < n; ++z) {
            if (prog[z].in == prog[q].out || prog[z].res == prog[q].out) { ok = false; break; }
            if (prog[z].out == prog[q].out) break;
          }
      }
      if (!ok) continue;
      RbHeadChain& c = R.heads.chain[R.heads.n_chains];
      c.conv_op = m; c.first = first; c.count = count; c.in_features = prog[first].in_features;
      c.in_off = R.heads.floats_per_sample;
      R.heads.floats_per_sample += (c.in_features + 3) & ~3;
      for (int l = 0; l + 1 < count; ++l) {
        c.hid_off[l] = R.heads.floats_per_sample;
        R.heads.floats_per_sample += (prog[first + l].out_features + 3) & ~3;
      }
      R.ops[m].head_chain = R.heads.n_chains;
      for (int q = first; q < first + count; ++q) R.ops[q].head_chain = R.heads.n_chains;
      ++R.heads.n_chains;
    }
  }
}

// Two half-shards on two streams (mzx_row_search.h): size of the FIRST half for a shard of `batch` trees, 0 = the shard
// runs undivided.  16-tree aligned (every per-tree array of the second half stays 16-byte aligned).  The summation
// order of a layer depends on the launch shape only through its channel groups (phases x chunks per group): a shard is
// split only when both halves run every layer with the groups of the undivided launch (true of every shipped
// configuration from 512 trees per half on: the planned tile), so the halves build the trees of the undivided run.
constexpr int RB_SPLIT_MIN_DEFAULT = 1024;
inline int rb_split_first(const mzx_net* net, int batch, int split_min = RB_SPLIT_MIN_DEFAULT) {
  if (!net || !net->rb.ok || !net->rb.recurrent.ok) return 0;
  if (split_min <= 0 || batch < split_min || batch < 32) return 0;
  const int first = ((batch / 2 + 15) / 16) * 16;
  if (first <= 0 || first >= batch) return 0;
  // a tower sums in (tap, chunk) order whatever its shape; the layer kernel in (channel group, tap, chunk) order: the
  // halves must take the same path as the undivided shard
  const RbProgram& R = net->rb.recurrent;
  std::vector<char> in_tower(R.ops.size(), 0);      // operators that run inside a tower launch (layers and tail)
  if (!net->rb_no_towers)
    for (const RbTower& tw : R.towers) {
      const bool w = rb_tower_use(tw, batch);
      if (rb_tower_use(tw, first) != w || rb_tower_use(tw, batch - first) != w) return 0;
      if (w)
        for (int k = tw.first; k < tw.first + tw.count + tw.n_tail && k < (int)R.ops.size(); ++k) in_tower[k] = 1;
    }
  for (size_t k = 0; k < R.ops.size(); ++k) {
    const RbOp& o = R.ops[k];
    if (o.kind != RB_GEMM || in_tower[k]) continue;    // (only the layers that launch on their own have channel groups)
    const RbShape w = rb_choose_shape(o, batch), h0 = rb_choose_shape(o, first), h1 = rb_choose_shape(o, batch - first);
    if (h0.phases != w.phases || h0.cpg != w.cpg || h1.phases != w.phases || h1.cpg != w.cpg) return 0;
  }
  return first;
}

// Plans one program; `layout` carries the layout of every logical buffer written so far.
inline bool rb_build_program(mzx_net* net, const std::vector<OpDesc>& prog, RbPlan& P, RbProgram& R, int64_t& cursor,
                             std::map<std::tuple<int64_t, int, int>, int64_t>& packed) {
  R.ops.clear();
  std::map<int, int> layout;
  auto layout_of = [&](int id) {
    if (id == BUF_IN || id == BUF_HIDDEN) return (int)RB_NCHW;
    auto it = layout.find(id);
    return it == layout.end() ? (int)RB_NCHW : it->second;
  };
  auto add_pack = [&](int64_t src, int taps, int cin, int cin_total, int cout) {
    const auto key = std::make_tuple(src, cin, taps);
    auto it = packed.find(key);
    if (it != packed.end()) return it->second;
    RzPack p;
    p.src = src; p.taps = taps; p.cin = cin; p.cin_total = cin_total;
    p.cchunks = rb_round16(cin) / 16; p.cout = cout;
    p.nchunks = taps * p.cchunks; p.wchunks = p.nchunks; p.ntiles = (cout + 15) / 16;
    p.dst = cursor;
    cursor += (int64_t)p.ntiles * p.wchunks * 256;
    P.packs.push_back(p);
    packed[key] = p.dst;
    return p.dst;
  };
  for (const OpDesc& d : prog) {
    RbOp o;
    o.in_layout = layout_of(d.in);
    o.res_layout = (d.res == -100) ? (int)RB_NCHW : layout_of(d.res);
    switch (d.kind) {
      case OP_CONV3: {
        o.kind = RB_GEMM; o.taps = 9; o.ksize = 3; o.pad = 1; o.stride = d.stride;
        o.cin_total = d.cin; o.cin = d.use_action ? d.cin - 1 : d.cin; o.cout = d.cout;
        o.hin = d.hin; o.win = d.win; o.hout = d.hout; o.wout = d.wout;
        o.act = d.relu ? RZ_ACT_RELU : RZ_ACT_NONE;
        o.out_layout = RB_NHWC;
        if (d.use_action) {
          RzAsum q;
          q.src = d.w; q.dst = cursor; q.cout = d.cout; q.cin_total = d.cin; q.H = d.hin; q.W = d.win;
          cursor += (int64_t)d.cout * d.hin * d.win;
          cursor = (cursor + 3) & ~int64_t(3);
          P.asums.push_back(q);
          o.asum_off = q.dst;
        }
        break;
      }
      case OP_CONV1: {
        o.kind = RB_GEMM; o.taps = 1; o.stride = 1;
        o.cin_total = o.cin = d.cin; o.cout = d.cout;
        o.hin = o.hout = d.hin; o.win = o.wout = 1;
        o.act = RZ_ACT_NONE;
        o.out_layout = RB_NCHW;          // the head MLP reads view(-1, R * H * W): (channel, position) order
        break;
      }
      case OP_LINEAR: {
        if (d.use_action) return false;  // fully connected dynamics only; never part of a residual network
        o.kind = RB_GEMM; o.taps = 1; o.stride = 1;
        o.cin_total = d.w_stride; o.cin = d.in_features; o.cout = d.out_features;
        o.act = d.elu ? RZ_ACT_ELU : RZ_ACT_NONE;
        o.in_layout = RB_NHWC; o.out_layout = RB_NHWC;   // one position per sample: both layouts coincide
        break;
      }
      case OP_SCALE: o.kind = RB_SCALE; o.out_layout = RB_NCHW; break;
      case OP_POOL: o.kind = RB_POOL; o.out_layout = o.in_layout; break;
      case OP_CONVK: {   // DownsampleCNN.features.0 / .3 (models.py:281-290): K x K, stride, padding, bias, ReLU
        o.kind = RB_GEMM; o.taps = d.ksize * d.ksize; o.ksize = d.ksize; o.pad = d.pad; o.stride = d.stride;
        o.cin_total = o.cin = d.cin; o.cout = d.cout;
        o.hin = d.hin; o.win = d.win; o.hout = d.hout; o.wout = d.wout;
        o.act = d.relu ? RZ_ACT_RELU : RZ_ACT_NONE;
        o.out_layout = RB_NHWC;
        break;
      }
      case OP_MAXPOOL: o.kind = (o.in_layout == RB_NHWC) ? RB_MAXPOOL : RB_FUNCTOR; o.out_layout = o.in_layout; break;
      case OP_ADAPTIVE_POOL: o.kind = (o.in_layout == RB_NHWC) ? RB_ADAPTIVE_POOL : RB_FUNCTOR; o.out_layout = o.in_layout; break;
      default:
        if (o.in_layout != RB_NCHW) return false;
        o.kind = RB_FUNCTOR; o.out_layout = RB_NCHW;
        break;
    }
    if (o.kind == RB_GEMM) {
      o.cchunks = rb_round16(o.cin) / 16;
      o.nchunks = o.taps * o.cchunks; o.wchunks = o.nchunks; o.ntiles = (o.cout + 15) / 16;
      if (!rb_choose_tile(o)) return false;
      o.w_off = add_pack(d.w, o.taps, o.cin, o.cin_total, o.cout);
    }
    if (d.out >= 0) layout[d.out] = o.out_layout;
    R.ops.push_back(o);
  }
  R.ok = 1;
  rb_find_towers(prog, R);
  rb_find_heads(prog, R);
  return true;
}

// Plans both programs of every residual network; called by mzx_net_create after rz_plan.  By default a program
// runs here when the fused engine does not take it; mzx_net_set_mode(3) routes every program here (A/B, parity of
// this engine against the reference's small configurations).
inline void rb_plan(mzx_net* net) {
  RbPlan& P = net->rb;
  P = RbPlan();
  if (net->cfg.network != 1) return;
  int64_t cursor = net->rz.ok ? net->rz.derived_floats : net->derived_floats;
  cursor = (cursor + 63) & ~int64_t(63);     // 256-byte aligned fragment images
  std::map<std::tuple<int64_t, int, int>, int64_t> packed;
  if (!rb_build_program(net, net->prog_initial, P, P.initial, cursor, packed)) P.initial.ok = 0;
  if (!rb_build_program(net, net->prog_recurrent, P, P.recurrent, cursor, packed)) P.recurrent.ok = 0;
  if (!P.initial.ok && !P.recurrent.ok) { P = RbPlan(); return; }
  P.derived_floats = cursor;
  P.head_floats = std::max(P.initial.ok ? P.initial.heads.floats_per_sample : 0, P.recurrent.ok ? P.recurrent.heads.floats_per_sample : 0);
  P.ok = 1;
}

// True when inference `recurrent` of `net` runs on the streamed engine (mode 0 = one element kernel per operator
// stays available as the A/B reference).
inline bool rb_enabled(const mzx_net* net, bool recurrent) {
#ifdef MZX_HOSTCHECK
  (void)net; (void)recurrent;
  return false;
#else
  if (!net->rb.ok || !net->rz_mode) return false;
  if (!(recurrent ? net->rb.recurrent.ok : net->rb.initial.ok)) return false;
  if (net->rb_force) return true;
  return !(net->rz.ok && (recurrent ? net->rz.recurrent.ok : net->rz.initial.ok));
#endif
}

#ifndef MZX_HOSTCHECK
// Defined in mzx_batched.hip.  Runs operators [0, n_ops) of the program (n_ops < 0: all of it).  With `ix`, sample b
// reads hidden-state node ix->in_node[b] of nb.in and writes node ix->out_node[b] of nb.hidden.
// `last_nchw`: when the last operator run is a pooling whose output would be position-major, it writes NCHW instead
// (the down-sampling stem in front of the LDS-resident engine, which gathers NCHW).
int rb_run_program(const mzx_net* net, bool recurrent, const NetBuffers& nb, int batch, stream_t stream,
                   const NetIndex* ix, int n_ops = -1, float* dump = nullptr, bool last_nchw = false);
// Packs the B fragments / action tap sums of the plan into the derived buffer (mzx_net_set_weights).
int rb_refresh_derived(const mzx_net* net, const float* d_flat, float* d_derived, stream_t stream);
#endif

}  // namespace mzx
