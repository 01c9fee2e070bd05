// mzx_tower.h -- what the two users of the TOWER body share: rb_tower_kernel (mzx_batched.hip: a whole trunk of
// MuZeroResidualNetwork, models.py:300-433, as one launch over a batch) and rt_search_kernel (mzx_tower_search.inc: every
// simulation of MCTS.run, self_play.py:319-355, in one launch with the same trunks inside).  The layer loop itself -- K loops,
// barriers, in-place epilogues -- is mzx_tower_layers.inc, included textually by both kernels: the same instructions on
// the same operands in the same order, hence the same bits on both routes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "mzx_resnet_batched.h"

namespace mzx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RB_THREADS = 512;

// Row tiles per wave from which the tower kernels hide loop-invariant address arithmetic from the compiler (an empty asm
// on the thread index / tree base / lane offsets, mzx_tower_search.inc, mzx_tower_layers.inc): the deep tilings keep
// those values in scratch across the K loops otherwise; the shallow ones have the registers and only pay the recomputation.
#ifndef MZX_OPAQUE_MIN_MT
#define MZX_OPAQUE_MIN_MT 5
#endif

struct RbTensor {
  const float* p;
  const int32_t* node;   // node of sample b inside [batch][nodes][sstride] (null: node 0)
  int64_t sstride;       // floats per node
  int32_t nodes;
  int32_t layout;        // RbLayout
};

// ReLU of a tower's epilogue: the maximum instruction fmaxf(x, 0.f) ends in, without the canonicalising v_max_f32 x, x
// the compiler puts in front of it for signalling NaNs (there are none: same result for every other operand, -0 included).
__device__ __forceinline__ float rb_relu(float x) {
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}

__device__ __forceinline__ int rb_div(int x, int d, uint32_t magic) {
  return d == 1 ? x : (int)__umulhi((unsigned)x, magic);
}

__device__ __forceinline__ uint32_t rb_magic_dev(int d) {      // ceil(2^32 / d), as rb_magic on the host
  return d > 1 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u;
}

struct RbTowerLayer {
  int64_t w_off;       // derived buffer: packed B fragments (RzPackOp) of the layer
  int64_t bn_alpha;    // derived buffer offsets of the folded BatchNorm terms (-1: none)
  int64_t bn_beta;
  int32_t cchunks;     // 16-channel chunks of the layer's input
  int32_t flags;       // 1: ReLU, 2: the layer's INPUT is the residual of the next layer (keep it), 4: add the kept residual
};

struct RbTowerArgs {
  RbTensor x;
  float* y;
  int64_t y_sstride;
  const float* der;
  const float* asum;       // first layer only: border-aware tap sums of the action plane (null: none)
  const int32_t* action;
  int32_t num_actions, batch;
  int32_t cin0, C, H, W, PH, PW, Cs, cchunks, T, rows, mtiles, ntiles, WN, WM, layers, y_vec, rowskip;
  int32_t dbg;             // latency experiments (env MZX_RB_DBG, never set in production): 1 skip the K loops, 2 skip the epilogues
  uint32_t magic_hw, magic_w, magic_phw, magic_pw, magic_chw;
  // ---- the tower's TAIL: operators that read nothing but the tower's output run on the LDS-resident tile before the
  // workgroup retires (a workgroup owns whole samples, so a per-plane reduction is an intra-workgroup one)
  int32_t write_out;       // 0: nobody else reads the tower's output -- it never goes to memory
  float* scale_y;          // per-plane min-max scaling (models.py:527-553, :574-599; MinMaxScaleOp's arithmetic) of the output,
  const int32_t* scale_node;   // written NCHW into the search arena's node store (null scale_y: none)
  int64_t scale_sstride;
  int32_t scale_nodes;
  int32_t n_conv;          // 1x1 head convolutions with few output channels (conv1x1_reward / _value / _policy, models.py:369-433)
  struct { const float* w; const float* b; float* y; int32_t R; int32_t pad; } conv[2];
  RbTowerLayer layer[RB_TOWER_MAX_LAYERS];
};

}  // namespace mzx
