// mzx_batched_plan.h -- descriptor types of the STREAMED residual-network engine
// (mzx_resnet_batched.h / mzx_batched.hip); kept apart so that `struct mzx_net` can embed them.
//
// The LDS-resident engine (mzx_resnet_fused.h) keeps a tile of trees inside ONE workgroup for a whole
// inference; it needs three activation slots of the network's width in 160 KB of LDS.  The reference's
// larger configurations -- games/gomoku.py:56-64 (128 channels x 6 blocks on 11 x 11), games/atari.py:61-69
// (256 channels x 16 blocks, 256-wide heads) -- do not fit.  This engine runs them layer by layer: every
// 3x3 convolution, 1x1 head convolution and Linear layer of models.py:206-623 is one FP32-MFMA implicit GEMM
// launch over ALL samples (M = batch x positions, N = Cout, K = taps x Cin), activations streaming through
// L2 / HBM between layers in a position-major (NHWC) layout.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "mzx_resnet_plan.h"
#include "mzx_tuning.h"

namespace mzx {

constexpr int RB_MAX_ROWS = 144;              // GEMM rows of a workgroup: nine 16-row MFMA tiles
constexpr int RB_MT = 9;                      // row tiles a wave can own
constexpr int RB_LDS_MAX = 156 * 1024;        // bytes of dynamic LDS a workgroup can get (one workgroup per CU)
constexpr int RB_LDS_BUDGET = 78 * 1024;      // what the planner asks for: TWO workgroups per CU, so that one stages its
                                              // patch / writes its outputs while the other feeds the matrix pipes

enum RbLayout { RB_NHWC = 0, RB_NCHW = 1 };   // [sample][position][channel] / [sample][channel][position]
enum RbKind { RB_GEMM = 0, RB_SCALE = 1, RB_POOL = 2, RB_FUNCTOR = 3, RB_MAXPOOL = 4, RB_ADAPTIVE_POOL = 5 };

// One operator of a program (index-aligned with mzx_net::prog_initial / prog_recurrent).
struct RbOp {
  int32_t kind = RB_FUNCTOR;
  int32_t in_layout = RB_NCHW, out_layout = RB_NCHW, res_layout = RB_NCHW;
  // ---- RB_GEMM: D[rows x cout] = A[rows x taps*cin] . B, rows = (sample, output position)
  int32_t taps = 1, stride = 1, cin = 0, cin_total = 0, cout = 0, hin = 1, win = 1, hout = 1, wout = 1;
  int32_t ksize = 1, pad = 0;   // square kernel: taps = ksize * ksize (3 / 1: the residual trunk; 5, 2 ceil(H / 16): DownsampleCNN)
  int32_t cchunks = 0;     // 16-channel K chunks per tap (cin padded to a multiple of 16)
  int32_t nchunks = 0, wchunks = 0, ntiles = 0;
  // workgroup tile: T whole samples (small maps) or one th x tw patch of output positions of one sample
  int32_t T = 1, th = 1, tw = 1, tiles_x = 1, tiles_y = 1;
  int32_t PH = 1, PW = 1;  // input patch incl. the halo of a 3x3 kernel, cells per sample
  int32_t cpg = 1;         // K chunks (of 16 channels) staged in LDS per phase
  int32_t phases = 1;      // ceil(cchunks / cpg): the input patch is staged in channel groups when it does not fit
  int32_t Cs = 24;         // LDS floats per cell: 16 * cpg + 8 (16-byte reads of 16 consecutive cells on distinct banks)
  int32_t rows = 1, mtiles = 1, lds_bytes = 0;
  int64_t w_off = -1;      // derived buffer: B fragments in v_mfma_f32_16x16x4_f32 lane order (RzPackOp)
  int64_t asum_off = -1;   // derived buffer: border-aware tap sums of the action plane's weights [cout][H*W]
  int32_t act = 0;         // RzAct
  int32_t tower = -1;      // index into RbProgram::towers when this operator is a layer of one
  int32_t head_chain = -1; // index into RbProgram::heads.chain: a tail convolution feeding it / a Linear layer of it
  int32_t tower_of_tail = -1;  // for an operator in a tower's tail: that tower
};

// A TOWER: a run of consecutive stride-1 3x3 convolutions of one width on one board size in which every operator reads
// what the previous one wrote -- the representation / dynamics trunk (conv + residual blocks, models.py:300-389) and the
// prediction trunk (:392-433).  rb_tower_kernel runs the whole run in ONE launch with the activations resident in LDS,
// updated IN PLACE (csrc/mzx_batched.hip); `first` / `count` index the program's operators.
constexpr int RB_TOWER_MAX_LAYERS = 40;
struct RbTower {
  int32_t first = 0, count = 0;
  int32_t C = 0, H = 0, W = 0;          // channels (= cout of every layer), board
  int32_t cchunks = 0, ntiles = 0;      // ceil(C / 16)
  int32_t t_max = 1;                    // most samples per workgroup that fit the LDS and the wave grid
  int32_t n_tail = 0;                   // operators directly behind the tower that run inside its launch (the per-plane
                                        // scaling, 1x1 head convolutions with at most RB_TAIL_MAX_R output channels)
};
constexpr int RB_TAIL_MAX_R = 8;

// HEADS: the MLPs behind the small 1x1 head convolutions (dynamics fc = reward, prediction fc_value / fc_policy,
// models.py:379-389, :418-433) -- chains of Linear (+ ELU) layers a few dozen neurons wide.  Launched one layer at a time
// they are latency-bound launches of a few microseconds each (six per connect4 / gomoku inference).  A chain qualifies
// for grouping when its input is written by a tower's tail (which then writes it into a private region of the workspace:
// nothing else can overwrite it before the end of the program) and its layers are at most RB_HEADS_MAX_WIDTH wide.
// How the chains run (tuning "rb_heads", mzx_tuning.h): 0 one rb_gemm_kernel launch per layer; 2 (default) one
// rb_gemm_multi_kernel launch per LEVEL -- the k-th layers of all chains as blockIdx.z slices, inputs and inner outputs in
// the private region.  Both run the same kernel body on the same shapes' channel groups: the same bits.  Measured
// (connect4, 512 samples): 0.292 / 0.269 ms per recurrent_inference (profiles/r04_tower_experiments.txt section 10; two
// more variants measured there -- all chains and levels in one MFMA launch with fences between the levels, all chains on
// the vector ALUs -- were slower at every shard size and left the tree in round 5).
inline int rb_heads_mode() { return tune(TUNE_RB_HEADS); }
constexpr int RB_MULTI_MT = 4;     // grouped / chained launches are instantiated for NT = 1, MT <= 4 (head layers: <1,1> or <2,1>)
constexpr int RB_HEADS_MAX_CHAINS = 3, RB_HEADS_MAX_LAYERS = 3, RB_HEADS_MAX_WIDTH = 128, RB_HEADS_MAX_IN = 1024;
struct RbHeadChain {
  int32_t conv_op = -1;      // the tail convolution that writes the chain's input
  int32_t first = -1, count = 0;   // its Linear operators (consecutive in the program)
  int32_t in_features = 0;
  int64_t in_off = 0;        // per-sample offset of its input inside the private region
  int64_t hid_off[RB_HEADS_MAX_LAYERS] = {0, 0, 0};   // ... of its inner outputs (grouped launches: one level of all chains at a time)
};
struct RbHeads {
  int32_t n_chains = 0;
  RbHeadChain chain[RB_HEADS_MAX_CHAINS];
  int64_t floats_per_sample = 0;   // private region: the chains' inputs and inner outputs
};

struct RbProgram {
  int32_t ok = 0;
  std::vector<RbOp> ops;
  std::vector<RbTower> towers;
  RbHeads heads;
};

struct RbPlan {
  int32_t ok = 0;
  int64_t head_floats = 0;      // per sample: the larger of the two programs' private head-input regions
  RbProgram initial, recurrent;
  std::vector<RzPack> packs;
  std::vector<RzAsum> asums;
  int64_t derived_floats = 0;   // end of this engine's part of the derived buffer
};

}  // namespace mzx
