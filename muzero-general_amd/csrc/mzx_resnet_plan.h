// mzx_resnet_plan.h -- descriptor types of the fused residual-network engine
// (mzx_resnet_fused.h); kept apart so that `struct mzx_net` can embed them.
#pragma once
#include <stdint.h>

#include <vector>

namespace mzx {

constexpr int RZ_MAX_OPS = 48;
constexpr int RZ_MAX_ROWS = 256;        // (trees per workgroup) x (board positions) rows of the activation matrix
constexpr int RZ_MAX_TREES = 16;        // trees per workgroup (one MFMA row tile for the per-tree GEMMs)

enum RzKind { RZ_GEMM = 0, RZ_SCALE = 1 };
enum RzRows { RZ_ROWS_POS = 0, RZ_ROWS_TREE = 1 };    // GEMM rows: (tree, position) pairs, or trees
enum RzOut { RZ_OUT_PADDED = 0, RZ_OUT_FLAT = 1 };
enum RzAct { RZ_ACT_NONE = 0, RZ_ACT_RELU = 1, RZ_ACT_ELU = 2 };

// One step of a fused program.  Region offsets are in floats PER TREE: a region
// starts at trees_per_workgroup * off inside the workgroup's LDS image, and tree t
// of the region at + t * (its per-tree stride).
struct RzOp {
  int32_t kind;          // RzKind
  int32_t rows;          // RzRows
  int32_t in_off, in_tstride;
  int32_t out_off, out_tstride, out_layout;   // RzOut
  int32_t res_off;       // residual input (padded layout), -1: none
  int32_t taps;          // 9 (3x3, padding 1) or 1
  int32_t cin4;          // input channels padded to a multiple of 4, / 4  (= K-steps per tap)
  int32_t cout;
  int32_t nchunks;       // ceil(taps * cin4 / 4): 16-deep K chunks of packed B fragments
  int32_t w_off;         // float offset of the packed B fragments in the derived buffer
  int32_t alpha_off, beta_off;   // folded BatchNorm (derived buffer), -1: none
  int32_t bias_off;      // bias (flat weight buffer), -1: none
  int32_t act;           // RzAct
  int32_t channels;      // RZ_SCALE: planes per tree
  int32_t store_hidden;  // RZ_SCALE: also write the scaled state to the caller's hidden-state output
  int32_t pad_;
};

// One packed weight tensor (deduplicated by source offset): B fragments of a GEMM
// in v_mfma_f32_16x16x4_f32 lane order, K = (tap, channel) tap-major.
struct RzPack {
  int64_t src;           // flat-buffer offset of W[cout][cin][taps]
  int64_t dst;           // derived-buffer offset
  int32_t taps, cin, cin4, cout, nchunks, ntiles;
};

struct RzProgram {
  int32_t ok = 0;
  int32_t first = 0;            // ops [0, first) of the operator program run as separate kernels (down-sampling stem)
  int32_t ext_buf = 0;          // logical buffer id feeding the fused part (BUF_IN or a stem temp)
  int32_t n_ops = 0;
  RzOp ops[RZ_MAX_OPS];
  int32_t in_off = 0;           // region receiving the input tensor
  int32_t in_channels = 0;      // channels of the input tensor
  int32_t use_action = 0;       // 1: plane `in_channels` of the input region = action / |A|
  int32_t out_off[3] = {-1, -1, -1};   // value, reward, policy logits (flat regions, per-tree stride = out_ts)
  int32_t out_ts[3] = {0, 0, 0};
  int32_t out_n[3] = {0, 0, 0};
  int64_t dev_off = 0;          // derived-buffer float offset of the uploaded RzOp table
};

struct RzGeometry {
  int32_t H = 0, W = 0, HW = 0, PW = 0;   // board, padded row width W + 2
  int32_t PS = 0;                         // plane stride (floats) of the padded layout
  int32_t Cbuf = 0;                       // planes per tree of a spatial slot
  int32_t slot_ts = 0;                    // Cbuf * PS
  int32_t tree_floats = 0;                // LDS floats per tree (3 slots + flat regions)
  int32_t max_trees = 0;                  // trees per workgroup the LDS budget admits
};

struct RzPlan {
  int32_t ok = 0;
  RzGeometry g;
  RzProgram initial, recurrent;
  std::vector<RzPack> packs;
  int64_t derived_floats = 0;             // packed weights + program tables appended to the derived buffer
};

}  // namespace mzx
