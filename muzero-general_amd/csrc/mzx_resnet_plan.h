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
  int32_t taps;          // low byte: 9 (3x3, padding 1) or 1; upper bits: ceil(2^20 / cchunks), the reciprocal of the
                         // arithmetic chunk -> tap decode (kernels that do not use the offset table)
  int32_t cchunks;       // 16-channel K chunks per tap (input channels padded to a multiple of 16)
  int32_t cout;
  int32_t nchunks;       // taps * cchunks
  int32_t wchunks;       // chunks stored per column tile: nchunks rounded up to even (the pad chunk is zero) --
                         // the pipeline consumes chunks in pairs
  int32_t w_off;         // float offset of the packed B fragments inside the program's weight image
  // epilogue parameters: float offsets inside the program's SMALL image (LDS-resident), -1: none
  int32_t alpha_off, beta_off;   // folded BatchNorm, padded to whole column tiles
  int32_t bias_off;      // bias, padded to whole column tiles
  int32_t asum_off;      // dynamics input convolution: border-aware tap sums of the action plane's weights [cout][H*W]
  int32_t act;           // RzAct
  int32_t channels;      // RZ_SCALE: planes per tree
  int32_t store_hidden;  // RZ_SCALE: also write the scaled state to the caller's hidden-state output
  uint32_t aoff_off;     // A-fragment offset table of this GEMM inside the small image (ints): chunk c at [3 + c],
                         // = (tap row - 1) * PW * Cs + (tap column - 1) * Cs + 16 * (channel chunk); the entries
                         // after the last chunk repeat it (the software pipeline prefetches past the end), the
                         // three leading pads put chunks 1..4 on a 16-byte boundary
  // Scheduling.  Operators without data dependences between them share a SLOT (one barrier per slot, not per
  // operator): the slot's operators are consecutive in the table, each runs on its own TEAM of waves.
  uint32_t sched;        // bits 0-7 / 8-15: log2 of the waves the column tiles are spread over when the whole
                         // 4- / 8-wave workgroup runs the operator; bit 16: last operator of its slot
  uint32_t team;         // bytes: first wave, wave count (4-wave kernels), first wave, wave count (8-wave kernels);
                         // count 0 = the whole workgroup
};
static_assert(sizeof(RzOp) == 96, "RzOp is fetched as 24 scalar words");

// One packed weight tensor: B fragments of a GEMM in v_mfma_f32_16x16x4_f32 lane order,
// K = (tap, 16-channel chunk); inside a chunk K-step j holds channels {4 g + j : g = 0..3}
// so that ONE 16-byte LDS read per lane (channels 4g .. 4g+3 of its row) feeds four K-steps.
struct RzPack {
  int64_t src;           // flat-buffer offset of W[cout][cin_total][taps]
  int64_t dst;           // derived-buffer offset
  int32_t taps, cin, cin_total, cchunks, cout, nchunks, wchunks, ntiles;   // cin = channels packed (<= cin_total)
};

// n floats copied (zero-padded to npad) into a program's small image: folded BatchNorm terms, biases
struct RzCopy {
  int64_t src;           // offset in the flat weight buffer (from_derived = 0) or the derived buffer (1)
  int64_t dst;           // derived-buffer offset
  int32_t n, npad, from_derived;
};

// sum over the in-board taps of W[co][action channel][ky][kx], per output position
struct RzAsum {
  int64_t src;           // flat-buffer offset of W[cout][cin_total][3][3]
  int64_t dst;           // derived-buffer offset of [cout][H*W]
  int32_t cout, cin_total, H, W;
};

struct RzProgram {
  int32_t ok = 0;
  int32_t first = 0;            // ops [0, first) of the operator program run as separate kernels (down-sampling stem)
  int32_t ext_buf = 0;          // logical buffer id feeding the fused part (BUF_IN or a stem temp)
  int32_t n_ops = 0;
  RzOp ops[RZ_MAX_OPS];         // in SCHEDULE order (slots); order[k] = position of operator first + k of the program
  int32_t order[RZ_MAX_OPS];
  int32_t n_slots = 0;
  int32_t aoff_base = 0;        // A-fragment offset tables: start inside the small image, host copy (filled by
  std::vector<int32_t> aoff;    // rz_finish_program once the activation geometry is known)
  int32_t in_off = 0;           // region receiving the input tensor
  int32_t in_channels = 0;      // channels of the input tensor
  int32_t use_action = 0;       // 1: the first op adds the action plane's contribution (asum_off)
  int32_t out_off[3] = {-1, -1, -1};   // value, reward, policy logits (flat regions, per-tree stride = out_ts)
  int32_t out_ts[3] = {0, 0, 0};
  int32_t out_n[3] = {0, 0, 0};
  // program image in the derived buffer: [packed weights: w_floats][small image: RzOp table, epilogue
  // parameters, action tap sums: small_floats].  The small image is always staged into LDS.
  int64_t w_base = 0;
  int32_t w_floats = 0;
  int64_t small_base = 0;
  int32_t small_floats = 0;
  int32_t flat_floats = 0;      // per-tree floats of the flat regions
};

struct RzGeometry {
  int32_t H = 0, W = 0, HW = 0, PW = 0;   // board, padded row width W + 2
  int32_t PP = 0;                         // padded positions per tree (H + 2) * (W + 2)
  int32_t Cs = 0;                         // channel stride of the position-major (NHWC) activation layout
  int32_t slot_ts = 0;                    // PP * Cs
};

// A 3x3 convolution of the down-sampling stem (large feature maps), run as its own MFMA kernel
struct RzStemConv {
  int32_t op_index;      // index in prog_initial
  int64_t w_off;         // derived-buffer offset of the packed weights
  int32_t cchunks, nchunks, wchunks;
};

struct RzPlan {
  int32_t ok = 0;
  RzGeometry g;
  RzProgram initial, recurrent;
  std::vector<RzPack> packs;
  std::vector<RzAsum> asums;
  std::vector<RzCopy> copies;
  std::vector<RzStemConv> stem;
  int64_t derived_floats = 0;             // packed weights + tables appended to the derived buffer
};

// Per-sample indirection through the search arena's hidden-state store [B][nodes][Hf]:
// sample b reads node in_node[b] and writes node out_node[b] (null: dense [B][Hf] tensors).
struct NetIndex {
  const int32_t* in_node = nullptr;    // null: node 0
  const int32_t* out_node = nullptr;   // null: node 0
  int32_t in_nodes = 1, out_nodes = 1; // nodes per sample of nb.in / nb.hidden (1 = dense tensors)
};

}  // namespace mzx
