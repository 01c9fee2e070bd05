// mzx_obs.h -- observation pipeline on the device.
//
// GameHistory.get_stacked_observations (/root/reference/self_play.py:513-550) builds the network
// input of a position from the game's history: the current observation followed, for the k =
// config.stacked_observations previous positions (most recent first), by that observation and a
// constant plane  action / len(action_space)  -- or zeros before the start of the game.  With
// config.stacked_observations = 32 (games/atari.py:30) that is 131 planes of 96x96: 4.8 MB per
// position, 36x the one new frame a move actually adds.  The host mirror therefore keeps the frames
// of a shard in HBM (one upload of the NEW frame per move) and this operator assembles the stacked
// inputs there: pure HBM-bound byte movement in 16-byte loads and stores.
//
// Frame store layout: frames[ring][G][C][H][W] fp32, actions[ring][G] int32; frame / action of game
// g at history index t live in slot t % ring (ring >= k + 1 for self-play; ring = len(history) for
// whole-game use such as Reanalyse, replay_buffer.py:343-360).
// Output n: [C * (k + 1) + k][H][W], sample n = (game[n] or n % G, time[n] or time0 + n / G).
//
// Values: a frame element is copied; the action plane is float(double(action) / double(A)) -- numpy
// evaluates ones_like(plane) * action / A in binary64 and the caller's torch.tensor(...).float()
// rounds once to fp32 (self_play.py:280-285).
#pragma once
#include "mzx_platform.h"
#include "mzx_tree.h"

namespace mzx {

// Work split: one wavefront-sized group of 64 elements moves one 4 KB piece (UNROLL x 64 vectors) of
// one output plane, so the (sample, plane) decode -- four 32-bit divisions -- is shared by UNROLL
// 16-byte moves per thread and every load / store instruction of a wave covers 1 KB contiguous.
template <int VEC>
struct ObsStackOp {
  static constexpr int UNROLL = 4;
  const float* frames;
  const int32_t* actions;
  const int32_t* game;   // nullable
  const int32_t* time;   // nullable
  float* out;
  int32_t time0, C, hwv /* H*W / VEC */, k, A, G, ring, n_out, c_out, pieces /* per plane */;

  MZX_HD size_t size() const { return (size_t)n_out * c_out * pieces * 64; }
  MZX_HD void operator()(size_t i) const {
    const uint32_t lane = (uint32_t)i & 63u, grp = (uint32_t)(i >> 6);
    const uint32_t plane = grp / (uint32_t)pieces, piece = grp % (uint32_t)pieces;
    const int c = (int)(plane % (uint32_t)c_out), n = (int)(plane / (uint32_t)c_out);
    const int g = game ? game[n] : n % G;
    const int t = time ? time[n] : time0 + n / G;
    const int64_t plane_floats = (int64_t)hwv * VEC;
    const int slot_t = t % ring;
    const float* src = nullptr;
    float fill = 0.f;
    if (c < C) {
      src = frames + (((int64_t)slot_t * G + g) * C + c) * plane_floats;
    } else {
      const int j = (c - C) / (C + 1), r = (c - C) % (C + 1);
      if (t - 1 - j >= 0) {                 // history index t - 1 - j; before the game: zeros
        int slot = slot_t - 1 - j;          // its slot: (t - 1 - j) % ring without the division
        if (slot < 0) slot += ring;         //   (j + 1 <= k < ring, or the ring holds the whole game)
        if (r < C) {
          src = frames + (((int64_t)slot * G + g) * C + r) * plane_floats;
        } else {
          int aslot = slot + 1;             // action_history[t - j]
          if (aslot >= ring) aslot -= ring;
          fill = (float)((double)actions[(int64_t)aslot * G + g] / (double)A);
        }
      }
    }
    float* dst = out + (int64_t)plane * plane_floats;
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t p = piece * (64u * UNROLL) + (uint32_t)u * 64u + lane;
      if (p >= (uint32_t)hwv) break;
      if (VEC == 4) {
#ifdef MZX_HOSTCHECK
        struct alignas(16) F4 { float x, y, z, w; };
        F4 v = {fill, fill, fill, fill};
        if (src) v = *(const F4*)(src + (int64_t)p * 4);
        *(F4*)(dst + (int64_t)p * 4) = v;
#else
        // streamed once in, once out: non-temporal 16-byte accesses keep the frames from displacing each other in L2
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 v = {fill, fill, fill, fill};
        if (src) v = __builtin_nontemporal_load((const v4*)(src + (int64_t)p * 4));
        __builtin_nontemporal_store(v, (v4*)(dst + (int64_t)p * 4));
#endif
      } else {
        dst[p] = src ? src[p] : fill;
      }
    }
  }
};

// models.support_to_scalar (models.py:645-666) of `rows` logit rows, one row per thread, in the
// search kernels' canonical summation order (mzx_tree.h) -- Reanalyse's value decode
// (replay_buffer.py:361-367).
struct SupportToScalarOp {
  const float* logits;
  float* out;
  int32_t rows, support_size;

  MZX_HD size_t size() const { return (size_t)rows; }
  MZX_HD void operator()(size_t i) const {
    out[i] = support_to_scalar(logits + (int64_t)i * (2 * support_size + 1), support_size);
  }
};

}  // namespace mzx
