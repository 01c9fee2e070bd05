// mzx_replay.h -- hand-off of finished games to the replay buffer: the INITIAL prioritised-replay priorities of a
// shard's games on the device.
//
// Reference: ReplayBuffer.save_game, /root/reference/replay_buffer.py:39-51 -- for every position i of a game
//     priority_i = |root_value_i - compute_target_value(game, i)| ** PER_alpha        (numpy.float64, then float32)
//     game_priority = max_i priority_i
// with compute_target_value (:230-262): the root value td_steps ahead (sign by whose turn it is there) times
// discount ** td_steps -- or the integer 0 past the end of the game --, plus the rewards of the next td_steps moves, each
// signed by whose turn it was and times discount ** i, accumulated in that order in binary64.
//
// One element per (game, position); G games of T positions each (a shard record: games that started and ended together;
// ragged shards go record by record).  Bit-exactness: the target value and the gap |root - target| use only binary64
// multiplications and additions in the reference's order (the library is compiled with -ffp-contract=off), with
// discount ** i taken from a table the HOST fills with Python's own float pow -- identical bit patterns to the reference
// (`d_targets` exposes them to the tests).  The final `** PER_alpha` is libm's pow in the reference: here binary64 sqrt
// for PER_alpha = 0.5 (every shipped configuration), the identity for 1, the device pow otherwise -- correctly rounded /
// within an ulp in binary64, i.e. the same float32 except when the binary64 result sits within ~1e-16 (relative) of a
// float32 rounding boundary (probability ~1e-9 per position); tests compare the float32 priorities bit for bit.
#pragma once
#include "mzx_platform.h"

namespace mzx {

struct ReplayPriorityOp {
  const double* root_values;    // [G][T]     root.value() of every searched position (0 for an unvisited root)
  const double* rewards;        // [G][T + 1] reward_history (leading 0)
  const int32_t* to_play;       // [G][T + 1] to_play_history
  const double* discount_pow;   // [td_steps + 1] discount ** i, host-computed
  double* targets;              // [G][T] nullable: compute_target_value
  float* priorities;            // [G][T]
  double per_alpha;
  int32_t num_games, moves, td_steps;

  MZX_HD size_t size() const { return (size_t)num_games * moves; }
  MZX_HD void operator()(size_t e) const {
    const int T = moves;
    const int g = (int)(e / T), index = (int)(e % T);
    const double* rv = root_values + (size_t)g * T;
    const double* rw = rewards + (size_t)g * (T + 1);
    const int32_t* tp = to_play + (size_t)g * (T + 1);
    const int me = tp[index];
    const int b = index + td_steps;
    double value = 0.0;
    if (b < T) {
      const double last = tp[b] == me ? rv[b] : -rv[b];
      value = last * discount_pow[td_steps];
    }
    const int stop = b < T ? b : T;       // reward_history[index + 1 : bootstrap_index + 1] has T + 1 entries
    for (int i = 0; index + 1 + i <= stop; ++i) {
      const double r = rw[index + 1 + i];
      const double s = me == tp[index + i] ? r : -r;
      value = value + s * discount_pow[i];
    }
    if (targets) targets[e] = value;
    const double gap = fabs(rv[index] - value);
    double p;
    if (per_alpha == 0.5) p = sqrt(gap);
    else if (per_alpha == 1.0) p = gap;
    else p = pow(gap, per_alpha);
    priorities[e] = (float)p;
  }
};

struct ReplayGameMaxOp {       // game_priority = numpy.max(priorities)
  const float* priorities;
  float* game_priority;
  int32_t num_games, moves;
  MZX_HD size_t size() const { return (size_t)num_games; }
  MZX_HD void operator()(size_t g) const {
    const float* p = priorities + g * (size_t)moves;
    float m = p[0];
    for (int i = 1; i < moves; ++i) m = p[i] > m ? p[i] : m;
    game_priority[g] = m;
  }
};

}  // namespace mzx
