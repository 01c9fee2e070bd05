// mzx_tuning.h -- the library's tuning surface: ONE process-wide table of named integers behind the C ABI
// (include/mzx.h: mzx_tuning_set / mzx_tuning_get / mzx_tuning_name), instead of environment variables read inside
// launch paths.  Every entry is a routing or launch-shape choice that never changes WHAT is computed beyond the
// summation orders documented per entry; the defaults are the measured best, tests and bench.py
// move them for A/B runs.  Reads are plain loads (set a value before the calls it should affect, from the thread that
// makes them -- the same rule as for handles, mzx.h "Conventions").
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace mzx {

struct TuningEntry {
  const char* name;
  int32_t value, dflt, lo, hi;
  const char* what;
};

enum TuningKey {
  TUNE_RB_HEADS = 0,          // head MLPs behind a tower's tail: 0 one launch per Linear layer, 2 one grouped launch per level
  TUNE_RB_TAIL,               // 1: scaling / small 1x1 head convolutions run inside the tower launch
  TUNE_RB_TOWER_T,            // > 0: samples per tower workgroup (0: the planner's cost model)
  TUNE_ROW_SPLIT_MIN,         // row-per-tree path: shards of at least this many trees run as two half-shards on two streams (0: never)
  TUNE_WIDE_TOWERS,           // 1: wide residual networks that also fit the LDS-resident engine search on the tower arithmetic (0: rz_search_kernel)
  TUNE_RT_SEARCH,             // tower whole-search kernel (rt_search_kernel): -1 automatic, 0 never, 1 whenever the network fits
  TUNE_RT_TREES,              // > 0: trees per workgroup of rt_search_kernel (0: the planner's cost model)
  TUNE_RT_WAVES,              // 4 / 8: waves per workgroup of rt_search_kernel (0: the planner's cost model)
  TUNE_RT_MAX_TREES,          // automatic routing: shards above this many trees stay on the two-stream streamed path
  TUNE_RT_DBG,                // timing experiments of rt_search_kernel (results are wrong with any bit set; never set in production)
  TUNE_RT_SHORT,              // 1: a last row group of MT - 1 row tiles runs the short K loop (0: every wave multiplies MT tiles, A/B)
  TUNE_ROUNDS_STREAMS,        // mzx_selfplay_rounds: 1 = every slot group behind the first searches on a stream of its own (0: one stream, A/B)
  TUNE_WAVE_SELECT,           // per-simulation launches, wide child records: 1 = a wavefront per tree walks (wave_select_kernel), 0 = a 16-lane row (row_select_kernel<0>, A/B); same walks
  TUNE_COUNT
};

inline TuningEntry* tuning_table() {
  static TuningEntry t[TUNE_COUNT] = {
      {"rb_heads", 2, 2, 0, 2, "head MLP levels: 0 = one launch per Linear layer, 2 = one grouped MFMA launch per level"},
      {"rb_tail", 1, 1, 0, 1, "tower tails (scaling, 1x1 head convolutions) inside the tower launch"},
      {"rb_tower_t", 0, 0, 0, 64, "samples per tower workgroup (0 = cost model)"},
      {"row_split_min", 1024, 1024, 0, 1 << 30, "two half-shards on two streams from this many trees (0 = never)"},
      {"wide_towers", 1, 1, 0, 1, "wide residual networks search on the tower arithmetic (0 = the LDS-resident whole-search kernel)"},
      {"rt_search", -1, -1, -1, 1, "tower whole-search kernel: -1 automatic, 0 never, 1 whenever supported"},
      {"rt_trees", 0, 0, 0, 16, "trees per workgroup of the tower whole-search kernel (0 = cost model)"},
      {"rt_waves", 0, 0, 0, 8, "waves per workgroup of the tower whole-search kernel: 4 or 8 (0 = cost model)"},
      {"rt_max_trees", 1 << 30, 1 << 30, 0, 1 << 30, "automatic routing: largest shard sent to the tower whole-search kernel"},
      {"rt_dbg", 0, 0, 0, 31, "timing experiments: 1 no K loops, 2 no epilogues, 4 no tree phases, 8 no staging / tails, 16 no head MLPs (wrong results)"},
      {"rt_short", 1, 1, 0, 1, "tower whole-search kernel: waves of a row group one tile short skip that tile's products (0 = multiply it, A/B)"},
      {"rounds_streams", 1, 1, 0, 1, "mzx_selfplay_rounds: slot groups behind the first search on streams of their own (0 = the caller's stream for all)"},
      {"wave_select", 1, 1, 0, 1, "per-simulation launches, more than 16 actions: the selection walk by a wavefront per tree (0 = a 16-lane row per tree, A/B; same walks)"},
  };
  return t;
}

inline int32_t tune(TuningKey k) { return tuning_table()[k].value; }

// Latency / layout EXPERIMENTS of earlier rounds (knock-outs, forced tilings, phase stamps; profiles/r0*_experiments*.txt)
// are environment variables -- in instrumented builds only (MZX_CXXFLAGS=-DMZX_EXPERIMENT, muzero-general_amd/build.py).  The
// product library never reads the environment: every such knob is its default there.
inline int exp_int(const char* name, int dflt) {
#ifdef MZX_EXPERIMENT
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

inline int tuning_find(const char* name) {
  if (!name) return -1;
  TuningEntry* t = tuning_table();
  for (int i = 0; i < TUNE_COUNT; ++i)
    if (strcmp(t[i].name, name) == 0) return i;
  return -1;
}

}  // namespace mzx
