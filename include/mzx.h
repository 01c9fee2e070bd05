/*
 * mzx.h -- C ABI of the MI355X-native MuZero self-play hot path (libmzx.so).
 *
 * The reference (werner-duvaud/muzero-general) has no FFI: its boundary for this
 * path is the duck-typed Python surface of models.py / self_play.py.  Every
 * entry point below names the reference interface it stands behind
 * (file:line relative to /root/reference); the Python host package
 * (muzero-general_amd/mzx) binds them with ctypes and re-exposes the reference's
 * own names (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C types only; every `d_*` pointer is a DEVICE pointer (e.g.
 *     torch.Tensor.data_ptr()); `h_*` pointers are host pointers read during
 *     the call.  The library never allocates or frees device memory: the
 *     caller owns every buffer (sizes via the *_bytes / *_floats queries).
 *   - `stream` is a hipStream_t (0 = default stream).  Calls only ENQUEUE work
 *     on it; they never synchronise.
 *   - return value: MZX_OK or a negative MZX_ERR_* code; mzx_last_error() gives
 *     the message of the calling thread's last failure.
 *   - handles are host-side descriptors; one handle may be used by one thread
 *     at a time (the reference runs one actor per process, self_play.py:11-29).
 */
#ifndef MZX_H
#define MZX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZX_ABI_VERSION 1

#define MZX_OK 0
#define MZX_ERR_INVALID (-1)      /* bad argument / unsupported configuration */
#define MZX_ERR_RUNTIME (-2)      /* HIP runtime error */
#define MZX_ERR_WORKSPACE (-3)    /* caller-provided buffer too small */

#define MZX_MAX_LAYERS 8          /* hidden layers per MLP */

int mzx_abi_version(void);
const char* mzx_last_error(void);
/* 1 if the library was built for the GPU (always, for libmzx.so). */
int mzx_is_device_build(void);

/* ------------------------------------------------------------------------- *
 * Tuning: the library's routing / launch-shape choices as ONE process-wide table of named integers
 * (csrc/mzx_tuning.h), in place of environment variables.  No entry changes WHAT is computed (the reference
 * has no counterpart: it has one code path); the defaults are the measured best, tests
 * and bench.py move them for A/B runs.  Set a value before the calls it should affect, from the calling thread.
 *   rb_heads       head MLP levels of the streamed engine: 0 one launch per Linear layer, 2 one grouped launch per level
 *   rb_tail        1: per-plane scaling / small 1x1 head convolutions run inside the tower launch
 *   rb_tower_t     > 0: samples per tower workgroup (0: cost model)
 *   row_split_min  per-simulation launches: two half-shards on two HIP streams from this many trees (0: never)
 *   wide_towers    1: wide residual networks that would also fit the LDS-resident whole-search kernel search on the
 *                  tower arithmetic at EVERY shard size (a tree's result must not depend on the shard it is searched
 *                  in); 0: mzx::rz_search_kernel (the A/B)
 *   rt_search      the tower whole-search kernel (every simulation in one launch): -1 automatic, 0 never, 1 whenever
 *                  the network fits;  rt_trees > 0: trees per workgroup;  rt_waves 4 | 8: waves per workgroup (0: cost
 *                  model);  rt_max_trees: largest shard routed to it;  rt_dbg: timing knock-outs (wrong results);
 *                  rt_short 1 | 0: waves of a row group one tile short skip that tile's products (0: A/B, same trees)
 *   rounds_streams mzx_selfplay_rounds: 1 = every slot group behind the first searches on a stream of its own (two half-shard
 *                  searches of a small network run side by side); 0 = all on the caller's stream (the A/B)
 *   wave_select    per-simulation launches, more than 16 actions: 1 = the selection walk by a wavefront per tree (lane l scores
 *                  child slots l, l + 64, ...), 0 = by a 16-lane row per tree (the A/B); the same walks either way
 * mzx_tuning_set / _get return MZX_ERR_INVALID for an unknown name or a value out of range; mzx_tuning_name /
 * _help enumerate the table (NULL past its end).
 * ------------------------------------------------------------------------- */
int mzx_tuning_set(const char* name, int32_t value);
int mzx_tuning_get(const char* name, int32_t* value /* nullable */, int32_t* default_value /* nullable */);
const char* mzx_tuning_name(int32_t index);
const char* mzx_tuning_help(int32_t index);

/* ------------------------------------------------------------------------- *
 * Network: replaces models.MuZeroNetwork(config) -- models.py:7-41 (factory),
 * :80-195 (MuZeroFullyConnectedNetwork), :436-623 (MuZeroResidualNetwork).
 * Field names are the MuZeroConfig attributes the factory reads.
 * ------------------------------------------------------------------------- */
typedef struct mzx_net_config {
  int32_t network;               /* 0 = "fullyconnected", 1 = "resnet" */
  int32_t observation_shape[3];  /* (channels, height, width) */
  int32_t stacked_observations;
  int32_t action_space_size;     /* len(config.action_space) */
  int32_t support_size;
  /* fullyconnected (models.py:80-126) */
  int32_t encoding_size;
  int32_t n_fc_representation_layers, fc_representation_layers[MZX_MAX_LAYERS];
  int32_t n_fc_dynamics_layers, fc_dynamics_layers[MZX_MAX_LAYERS];
  int32_t n_fc_reward_layers, fc_reward_layers[MZX_MAX_LAYERS];
  int32_t n_fc_value_layers, fc_value_layers[MZX_MAX_LAYERS];
  int32_t n_fc_policy_layers, fc_policy_layers[MZX_MAX_LAYERS];
  /* resnet (models.py:436-520) */
  int32_t downsample;            /* 0 = False, 1 = "resnet" (DownSample, models.py:233-275), 2 = "CNN" (DownsampleCNN, :278-297) */
  int32_t blocks, channels;
  int32_t reduced_channels_reward, reduced_channels_value, reduced_channels_policy;
  int32_t n_resnet_fc_reward_layers, resnet_fc_reward_layers[MZX_MAX_LAYERS];
  int32_t n_resnet_fc_value_layers, resnet_fc_value_layers[MZX_MAX_LAYERS];
  int32_t n_resnet_fc_policy_layers, resnet_fc_policy_layers[MZX_MAX_LAYERS];
} mzx_net_config;

typedef struct mzx_net mzx_net;

/* models.MuZeroNetwork.__new__ (models.py:7-41): validates, builds the operator program. */
int mzx_net_create(const mzx_net_config* cfg, mzx_net** out);
void mzx_net_destroy(mzx_net* net);

/* The flat weight buffer = every floating-point tensor of the reference
 * state_dict (AbstractNetwork.get_weights, models.py:69-70) concatenated in
 * state_dict order (BatchNorm running stats included, num_batches_tracked
 * skipped).  This one buffer is what RCCL broadcasts. */
int32_t mzx_net_num_tensors(const mzx_net* net);
int64_t mzx_net_num_params(const mzx_net* net);
/* i-th tensor: reference state_dict key (with the ".module." infix), offset/numel
 * in floats, up to 4 dims (unused dims = 0). */
int mzx_net_tensor_info(const mzx_net* net, int32_t i, char* name, int32_t name_cap,
                        int64_t* offset, int64_t* numel, int32_t dims[4]);
int64_t mzx_net_hidden_size(const mzx_net* net);     /* floats per encoded state */
int64_t mzx_net_input_size(const mzx_net* net);      /* floats per stacked observation */
int64_t mzx_net_derived_floats(const mzx_net* net);  /* folded-BatchNorm buffer size */
int64_t mzx_net_workspace_floats(const mzx_net* net, int32_t max_batch);
/* 2 x multiply-accumulates of one sample's initial_inference (recurrent = 0) or
 * recurrent_inference (1): convolutions and linear layers, as the reference's modules count. */
int64_t mzx_net_flops(const mzx_net* net, int32_t recurrent);

/* AbstractNetwork.set_weights (models.py:72-73): bind the flat device buffer
 * (kept alive by the caller) and derive folded BatchNorm terms into d_derived. */
int mzx_net_set_weights(mzx_net* net, const float* d_flat, int64_t n_floats,
                        float* d_derived, int64_t derived_floats, void* stream);

/* Residual networks run on the fused MFMA engine (one launch per inference, activations
 * resident in LDS) where the configuration allows it.  mzx_net_fused_supported: bit 0 =
 * initial_inference, bit 1 = recurrent_inference.  mzx_net_set_mode(0) forces one kernel per
 * operator (the correctness reference of the tuned engine), 1 (default) re-enables it. */
int mzx_net_fused_supported(const mzx_net* net);
int mzx_net_set_mode(mzx_net* net, int32_t mode);

/* Residual networks the fused engine cannot hold in LDS (reference games/gomoku.py:56-64 -- 128 channels x 6
 * blocks on 11 x 11 -- and games/atari.py:61-69 -- 256 channels x 16 blocks, 256-wide heads; models.py:436-623)
 * run on the STREAMED MFMA engine: one FP32-MFMA implicit-GEMM launch per convolution / Linear layer over the
 * whole batch.  mzx_net_streamed_supported: bit 0 = initial_inference, bit 1 = recurrent_inference run there
 * by default (mode 0 still forces one element kernel per operator; mzx_net_set_mode(3) routes EVERY residual
 * program there, fused-capable ones included -- the A/B and parity knob; 4 = as 3 and 5 = as 1, but the streamed
 * engine launches every layer on its own, no towers).  mzx_net_streamed_plan: the workgroup tiling of
 * operator `op`: {kind, in_layout, out_layout, res_layout, taps, stride, cin, cout, hin, win, hout, wout, T, th,
 * tw, tiles_x, tiles_y, PH, PW, chunks per phase, phases, rows, row tiles, LDS bytes}. */
int mzx_net_streamed_supported(const mzx_net* net);
int mzx_net_streamed_plan(const mzx_net* net, int32_t recurrent, int32_t op, int32_t out[24]);
/* The launch shape the streamed engine picks for GEMM operator `op` at `batch` samples (small batches take fewer samples
 * per workgroup, split the column tiles over more workgroups and use fewer channel phases): {T, rows, row tiles, LDS
 * bytes, column tiles per workgroup, column splits, column tiles per wave, waves along N, waves along M, row tiles per
 * wave, row groups (grid.x), chunks per phase, phases, LDS floats per cell, column tiles, 16-channel chunks per tap}. */
int mzx_net_streamed_shape(const mzx_net* net, int32_t recurrent, int32_t op, int32_t batch, int32_t out[16]);
/* TOWERS: runs of stride-1 3x3 convolutions of one width (the representation / dynamics / prediction trunks,
 * models.py:300-433) run as ONE launch of rb_tower_kernel with the activations resident in LDS, updated in place
 * (mzx_net_set_mode(4) / (5): layer by layer instead, the A/B).  Tower `index` of the program at `batch` samples:
 * {first operator, operators, channels, H, W, samples per workgroup, row tiles per wave, column tiles per wave, waves
 * along M, waves along N, LDS bytes, workgroups (0: at this batch the tower's shape would waste more than a fifth of the
 * MFMA rows and its layers launch one by one), operators directly behind the tower that run INSIDE its launch on the
 * LDS-resident output (the per-plane min-max scaling, 1x1 head convolutions with few output channels: the tower's
 * output then never reaches memory), 0, 0, 0}; an error when the program has no such tower.  The operators of a
 * tower are not launched one by one (mzx_net_streamed_shape still describes what the layer-by-layer path would do). */
int mzx_net_streamed_tower(const mzx_net* net, int32_t recurrent, int32_t index, int32_t batch, int32_t out[16]);
/* HEADS: the Linear chains behind the small 1x1 head convolutions (dynamics fc, prediction fc_value / fc_policy,
 * models.py:379-433), when their tower runs with its tail, leave the one-launch-per-layer path: by default the k-th
 * layers of all chains run as slices of ONE rb_gemm_multi_kernel launch per level (tuning "rb_heads" = 2; the layer
 * kernel's own body and shapes: the same bits); "rb_heads" = 0: one rb_gemm_kernel launch per layer.  out = {[0] Linear operators covered at `batch` samples, [1] chains, [2..13] their operator indices,
 * [14] their levels inside their chains (2 bits each, operator k at bits 2k), [15] the mode in effect}; all zero when
 * off or when no chain qualifies. */
int mzx_net_streamed_heads(const mzx_net* net, int32_t recurrent, int32_t batch, int32_t out[16]);
/* The row-per-tree search runs large shards as two half-shards on two HIP streams (csrc/mzx_row_search.h; from 1024
 * trees, and only when both halves keep the channel groups -- the summation order -- of the undivided launch):
 * out = {trees of the first half, trees of the second half}; {batch, 0} when a shard of `batch` trees runs undivided.
 * recurrent_inference is then launched at THESE batch sizes (tests / bench.py derive the kernel instantiations a
 * workload launches from it, host-side). */
int mzx_net_streamed_split(const mzx_net* net, int32_t batch, int32_t out[2]);
/* Floats per sample of the output tensor of operator `op` (what mzx_net_debug_prefix copies out for a prefix ending
 * there); 0 for an unknown operator. */
int64_t mzx_net_operator_out_floats(const mzx_net* net, int32_t recurrent, int32_t op);

/* initial_inference(observation) (models.py:172-190 / :601-618).
 * d_observation [batch][input_size]; outputs value_logits [batch][2s+1],
 * policy_logits [batch][A], hidden [batch][hidden_size].  The reward of the
 * root is log(one-hot at the support centre), i.e. scalar 0 -- d_reward_logits
 * (nullable) is filled with -inf / 0 accordingly. */
int mzx_net_initial_inference(mzx_net* net, const float* d_observation, int32_t batch,
                              float* d_value_logits, float* d_reward_logits,
                              float* d_policy_logits, float* d_hidden,
                              float* d_workspace, int64_t workspace_floats, void* stream);

/* recurrent_inference(encoded_state, action) (models.py:192-195 / :620-623). */
int mzx_net_recurrent_inference(mzx_net* net, const float* d_hidden, const int32_t* d_action,
                                int32_t batch, float* d_value_logits, float* d_reward_logits,
                                float* d_policy_logits, float* d_next_hidden,
                                float* d_workspace, int64_t workspace_floats, void* stream);

/* Diagnostics (parity bisection, no reference counterpart): run the first n_ops operators of
 * initial_inference (recurrent = 0) or recurrent_inference (1) on the per-operator (fused = 0)
 * or fused (1) engine and copy the LAST operator's output tensor, dense per sample, to d_out.
 * fused = 2: run the whole fused program and write workgroup 0's shader-clock stamps (uint64: after
 * staging, after the input load, after every operator, at the end) to d_out instead.
 * d_scratch: (hidden_size + 2 * (2 * support_size + 1) + action_space_size) * batch floats. */
int mzx_net_num_operators(const mzx_net* net, int32_t recurrent);
/* Schedule of the fused residual engine: slot_of_op[k] = slot (barrier interval) operator k of the program
 * runs in, -1 for operators outside the fused part (down-sampling stem); operators sharing a slot have no
 * data dependence and run concurrently on disjoint wave teams.  Returns the number of slots, 0 if the
 * program is not fused.  `cap` = entries available in slot_of_op. */
int mzx_net_fused_schedule(const mzx_net* net, int32_t recurrent, int32_t* slot_of_op, int32_t cap);
int mzx_net_debug_prefix(mzx_net* net, int32_t recurrent, int32_t fused, int32_t n_ops, const float* d_input,
                         const int32_t* d_action, int32_t batch, float* d_out, int64_t out_floats,
                         float* d_scratch, int64_t scratch_floats, float* d_workspace, int64_t workspace_floats,
                         void* stream);

/* ------------------------------------------------------------------------- *
 * Batched search: replaces MCTS(config).run(...) -- self_play.py:249-430 -- for
 * num_trees independent roots at once (the reference runs one).
 * ------------------------------------------------------------------------- */
typedef struct mzx_search_config {
  int32_t num_trees;           /* B parallel roots (one per game of the shard) */
  int32_t num_simulations;     /* config.num_simulations */
  int32_t action_space_size;   /* len(config.action_space) */
  int32_t num_players;         /* len(config.players): 1 or 2 */
  int32_t support_size;        /* config.support_size */
  int32_t tape_words;          /* raw MT19937 words per tree for argmax tie draws */
  double discount;             /* config.discount */
  double root_exploration_fraction;
  /* host tables, entries 0..num_simulations (index = parent visit count), built
   * by the caller with the language's own math library so that they are the
   * reference's values bit for bit (self_play.py:384-391):
   *   h_pb_c_table[n] = log((n + pb_c_base + 1) / pb_c_base) + pb_c_init
   *   h_sqrt_table[n] = sqrt(n) */
  const double* h_pb_c_table;
  const double* h_sqrt_table;
} mzx_search_config;

typedef struct mzx_search mzx_search;

int mzx_search_create(const mzx_search_config* cfg, mzx_net* net /* nullable: lock-step only */,
                      mzx_search** out);
void mzx_search_destroy(mzx_search* s);
/* Device memory the caller must provide as `d_arena` (trees + per-node hidden
 * states + per-simulation scratch + network workspace).  The arena is scratch: its contents need
 * NOT persist between calls (only mzx_search_dump / the lock-step calls read what the preceding
 * call of the same handle left there); the pb_c / sqrt tables live in a small device buffer owned
 * by the handle (allocated on the device current at mzx_search_create, freed by destroy). */
int64_t mzx_search_arena_bytes(const mzx_search* s);

/* Per-move inputs / outputs (all device pointers, all [num_trees] leading). */
typedef struct mzx_search_io {
  const float* d_observation;    /* [B][input_size] stacked observations (self_play.py:138-140) */
  const int32_t* d_legal_actions;/* [B][A] game.legal_actions() in the game's order, padded with -1 */
  const int32_t* d_to_play;      /* [B] game.to_play() */
  const double* d_noise;         /* [B][A] Dirichlet sample per root child slot; NULL = no exploration noise */
  const uint32_t* d_tape;        /* [B][tape_words] raw MT19937 words following the Dirichlet draw */
  int32_t* d_visit_counts;       /* out [B][A] root child visit counts by action (0 if not a child) */
  double* d_root_value;          /* out [B] root.value() */
  double* d_root_predicted_value;/* out [B] support_to_scalar(initial value head) */
  int32_t* d_info;               /* out [B][4]: max_tree_depth, flags (1 tape overflow, 2 node overflow),
                                    tape words consumed, sum of leaf depths */
} mzx_search_io;

/* MCTS.run for B roots: initial_inference, root expansion (+noise), then
 * num_simulations x {select, recurrent_inference, expand, backpropagate}. */
int mzx_search_run(mzx_search* s, const mzx_search_io* io, void* d_arena, int64_t arena_bytes, void* stream);

/* MCTS.run(..., override_root_with=root) (self_play.py:275-277) for roots the caller expanded itself,
 * as diagnose_model.py:57-74 does after a recurrent_inference: d_root_hidden [B][hidden_size],
 * d_root_priors [B][A] (binary64 Node.prior of the root's children in slot order = order of
 * io->d_legal_actions), d_root_reward [B] (Node.reward).  io->d_observation is not read and
 * io->d_root_predicted_value not written (the reference reports None).  Exploration noise, the
 * simulations and the outputs are those of mzx_search_run, on the same kernels: the residual whole-search kernels /
 * streamed engine read the given roots from the arena, the fully connected whole-search kernel (fc2_search_kernel)
 * takes them as launch arguments in place of its initial_inference. */
int mzx_search_run_from_roots(mzx_search* s, const mzx_search_io* io, const float* d_root_hidden,
                              const double* d_root_priors, const double* d_root_reward, void* d_arena,
                              int64_t arena_bytes, void* stream);

/* Which implementation mzx_search_run uses, a set of flags:
 *   1  whole-search kernel: every simulation of the move in one launch.  mzx_search_fused_supported
 *      returns 1 for the LDS-resident fully connected kernel, 2 for the residual-network kernel
 *      (trees in the arena, network on the fused MFMA engine), 3 for networks that run layer by layer on
 *      the streamed MFMA engine (per simulation: a 16-lane row per tree selects, one launch per layer,
 *      a row per tree expands and back-propagates; csrc/mzx_row_search.h), 0 if none applies;
 *      0 = generic path (select / network / expand+backpropagate launches per simulation, one thread per tree)
 *   2  (fused) also export the finished trees to the arena so mzx_search_dump works
 *   4  (fused) force the LDS-weight engine even when a register-resident
 *      specialisation matches the network shape
 *   8  cycle profile: phase cycle counters are left in the network-workspace region of the
 *      arena (mzx_search_arena_offsets): [tree][16] for the fully connected kernel (a
 *      separate profiling instantiation), [workgroup][8] for the residual kernel
 *  16  (fully connected whole-search kernel) run the first-generation kernel (per-level UCB evaluation,
 *      csrc/mzx_fused_fc.h) instead of the cached-prior-score kernel (csrc/mzx_fused_fc2.h): A/B measurements
 * Default: 1 when supported, else 0. */
int mzx_search_fused_supported(const mzx_search* s);
/* Name of the search kernel the last mzx_search_run of this handle launched ("" before the first run): which of the
 * whole-search kernels a configuration is routed to is decided per launch (network family, board size, LDS fit). */
const char* mzx_search_kernel_name(const mzx_search* s);
int mzx_search_set_mode(mzx_search* s, int32_t mode);
/* What the NEXT mzx_search_run of this handle would launch (host-side, no GPU): out[0] = 0 the generic path, 1
 * mzx::rz_search_kernel or one of its small-board siblings, 2 per-simulation launches around the streamed engine, 3
 * mzx::rt_search_kernel (every simulation in one launch, csrc/mzx_tower_search.inc), 4 the fully connected whole-search
 * kernel; for 3: out[1..6] = {trees per workgroup, row tiles per wave, workgroups, workgroups per CU, LDS bytes, threads per
 * workgroup}; for 2: out[6..7] = trees of the two half-shards (second 0: undivided). */
int mzx_search_route(const mzx_search* s, int32_t out[8]);
/* The same for a search of `num_trees` roots x `num_simulations` on `net` in the default mode, WITHOUT a search handle (no
 * device needed: tests/test_streamed_coverage.py keeps bench.py's workloads inside the GPU-tested launch shapes with it). */
int mzx_net_search_route(const mzx_net* net, int32_t num_trees, int32_t num_simulations, int32_t out[8]);

/* Byte offsets inside the arena (diagnostics): out[0..6] = (unused, 0), trees, hidden states,
 * network workspace, bytes per tree, workspace bytes, total bytes. */
int mzx_search_arena_offsets(const mzx_search* s, int64_t out[8]);

/* Lock-step interface (parity harness, SURVEY.md section 8c'): the tree arithmetic
 * alone, with the network outputs supplied by the caller in binary64.
 *   begin : root expansion from d_root_priors [B][A] (slot order) and d_root_reward (+ io->d_noise)
 *   select: one selection walk per tree -> d_parent/d_action/d_leaf [B]
 *   apply : expand the selected leaf with (value, reward, priors [B][A]) and back-propagate
 *   finish: write io outputs */
int mzx_search_lockstep_begin(mzx_search* s, const mzx_search_io* io, const double* d_root_priors,
                              const double* d_root_reward /* [B], NULL = decoded zero-reward head */,
                              void* d_arena, int64_t arena_bytes, void* stream);
int mzx_search_lockstep_select(mzx_search* s, const mzx_search_io* io, int32_t* d_parent, int32_t* d_action,
                               int32_t* d_leaf, void* d_arena, void* stream);
int mzx_search_lockstep_apply(mzx_search* s, const double* d_value, const double* d_reward,
                              const double* d_priors, void* d_arena, void* stream);
int mzx_search_finish(mzx_search* s, const mzx_search_io* io, void* d_arena, void* stream);

/* Canonical-order dump of the trees (parity tests / diagnose tooling):
 * copies node statistics of every tree into caller arrays sized for
 * num_simulations+1 nodes: visit [B][N] i32, value_sum [B][N] f64, reward [B][N] f64,
 * to_play [B][N] i32, parent [B][N] i32, child [B][N][A] i32, prior [B][N][A] f64,
 * minmax [B][2] f64, n_nodes [B] i32.  Any pointer may be NULL. */
typedef struct mzx_tree_dump {
  int32_t* d_visit; double* d_value_sum; double* d_reward; int32_t* d_to_play; int32_t* d_parent;
  int32_t* d_child; double* d_prior; double* d_minmax; int32_t* d_n_nodes;
} mzx_tree_dump;
int mzx_search_dump(mzx_search* s, const mzx_tree_dump* dump, void* d_arena, void* stream);

/* ------------------------------------------------------------------------- *
 * Observation pipeline on the device (csrc/mzx_obs.h).
 * mzx_obs_stack replaces GameHistory.get_stacked_observations (self_play.py:513-550) for n_out
 * positions at once, reading a device-resident frame store instead of the Python history lists:
 *   d_frames  [ring][num_games][channels][height][width] fp32,  d_actions [ring][num_games] int32;
 *   the observation / action of game g at history index t live in slot t % ring
 *   (action_history convention of the reference: index 0 holds the leading 0, self_play.py:118).
 * Sample n is (game, index) = (d_game[n], d_time[n]); a NULL d_game means n % num_games, a NULL
 * d_time means time0 + n / num_games.  The caller guarantees 0 <= index and that the frames
 * index - stacked_observations .. index are present (ring >= stacked_observations + 1).
 *   d_out [n_out][channels * (stacked + 1) + stacked][height][width] fp32 = the reference's array
 *   after torch.tensor(...).float() (self_play.py:280-285), bit for bit.
 * mzx_support_to_scalar = models.support_to_scalar (models.py:645-666) of `rows` rows of
 * 2 * support_size + 1 logits -> d_out[rows] (Reanalyse's value decode, replay_buffer.py:361-367).
 * ------------------------------------------------------------------------- */
typedef struct mzx_obs_layout {
  int32_t channels, height, width; /* config.observation_shape */
  int32_t stacked_observations;    /* config.stacked_observations */
  int32_t action_space_size;       /* len(config.action_space) */
  int32_t num_games;               /* games interleaved in the frame store */
  int32_t ring;                    /* history slots per game */
} mzx_obs_layout;
int64_t mzx_obs_stacked_floats(const mzx_obs_layout* layout); /* floats per stacked sample; 0 if invalid */
int mzx_obs_stack(const mzx_obs_layout* layout, const float* d_frames, const int32_t* d_actions,
                  const int32_t* d_game, const int32_t* d_time, int32_t time0, int32_t n_out, float* d_out,
                  void* stream);
int mzx_support_to_scalar(const float* d_logits, int32_t rows, int32_t support_size, float* d_out, void* stream);

/* ------------------------------------------------------------------------- *
 * Per-game random streams (host side, no GPU): replaces the numpy.random calls of one
 * self-play actor -- numpy.random.seed (self_play.py:22), numpy.random.dirichlet (:473), the
 * numpy.random.choice(ties) of the search (:371, drawn on the device from the tape this produces)
 * and SelfPlay.select_action's numpy.random.choice (:229-243) -- for a SHARD of games: stream i is
 * numpy.random.RandomState(seed_i), bit for bit (legacy MT19937 algorithms, csrc/mzx_rng.h).
 * All arrays are host arrays; `idx[k]` selects the stream of the k-th entry.
 * ------------------------------------------------------------------------- */
typedef struct mzx_rng mzx_rng;
int mzx_rng_create(int32_t num_streams, mzx_rng** out);
void mzx_rng_destroy(mzx_rng* r);
/* RandomState(seeds[k]) for streams first .. first + count - 1 (0 <= seed < 2^32). */
int mzx_rng_seed(mzx_rng* r, int32_t first, int32_t count, const uint32_t* seeds);
/* RandomState.get_state() / set_state() of one stream: key[624], pos, has_gauss, cached_gaussian. */
int mzx_rng_get_state(const mzx_rng* r, int32_t i, uint32_t* key, int32_t* pos, int32_t* has_gauss, double* gauss);
int mzx_rng_set_state(mzx_rng* r, int32_t i, const uint32_t* key, int32_t pos, int32_t has_gauss, double gauss);
/* Per-move root draws of `count` games: noise[k][0..n_legal[k]) = dirichlet([alpha] * n_legal[k])
 * (noise == NULL: no exploration noise, nothing drawn), then tape[k][0..tape_words) = the NEXT raw
 * 32-bit words of the stream WITHOUT consuming them (the search consumes some; see mzx_rng_advance). */
int mzx_rng_root_draws(mzx_rng* r, const int32_t* idx, int32_t count, double alpha, const int32_t* n_legal,
                       int32_t action_space_size, double* noise, int32_t tape_words, uint32_t* tape, int32_t n_threads);
/* Consume words[k] raw words of stream idx[k] (what the search reported in info[2]). */
int mzx_rng_advance(mzx_rng* r, const int32_t* idx, int32_t count, const int32_t* words);
/* One RandomState.random_sample() per stream (numpy.random.choice(a, p=...) draws exactly one). */
int mzx_rng_random_sample(mzx_rng* r, const int32_t* idx, int32_t count, double* out);
/* One RandomState.randint(0, n[k]) per stream (numpy.random.choice(list of n) draws exactly that). */
int mzx_rng_randint(mzx_rng* r, const int32_t* idx, int32_t count, const int32_t* n, int32_t* out);
/* numpy.random.choice(actions, p=dist / sum(dist)) of SelfPlay.select_action (self_play.py:236-243) for `count`
 * streams at once: weights [count][row_stride] binary64 (the caller's visit_counts ** (1 / temperature), first
 * n[k] entries of a row are used), out[k] = the chosen POSITION in 0..n[k]-1.  Same arithmetic as the Python
 * statement + numpy's legacy choice: left-to-right sum, division, cumulative sum, division by its last
 * element, one random_sample(), right bisection. */
int mzx_rng_choice_weighted(mzx_rng* r, const int32_t* idx, int32_t count, const double* weights,
                            int32_t row_stride, const int32_t* n, int32_t* out);

/* ------------------------------------------------------------------------- *
 * One self-play MOVE of a shard of games in two host calls (self_play.py:138-181 for B games at once): what an actor
 * does around MCTS.run on the host -- root noise draw (:473), tie-break words, the upload of the move's inputs, the
 * search, the download, the streams' advance, SelfPlay.select_action (:222-245) -- without a Python statement per
 * field in between.  `mzx_move` describes a move: host arrays as the game plugin returns them, ONE pinned host
 * block + ONE device block of the same layout for the inputs and for the outputs (io's device pointers point into
 * d_in / d_out; a field's host copy is at the same offset of h_in / h_out; io.d_observation may point elsewhere when
 * `observation` is NULL = the stacked observations are already on the device; h_in == d_in on a build without a
 * device).  Same arithmetic, same draws in the same order as the calls it replaces (mzx_rng_root_draws,
 * mzx_search_run, mzx_rng_advance, mzx_rng_choice_weighted / mzx_rng_randint), pinned by tests/test_selfplay_move.py.
 * ------------------------------------------------------------------------- */
typedef struct mzx_move {
  int32_t num_games;               /* B */
  int32_t action_space_size;       /* A */
  int32_t tape_words;              /* raw words per game handed to the search for its tie breaks */
  int32_t num_threads;             /* host threads for the per-game draws (persistent pool of the bank) */
  const int32_t* streams;          /* [B] stream of game k in the bank */
  const int32_t* legal_actions;    /* host [B][A], game.legal_actions() padded with -1 (validated: self_play.py:296-301) */
  const int32_t* to_play;          /* host [B] */
  const float* observation;        /* host [B][observation_floats], or NULL (io.d_observation holds them already) */
  int64_t observation_floats;
  double dirichlet_alpha;          /* config.root_dirichlet_alpha */
  int32_t add_exploration_noise;   /* 0: no noise drawn (io.d_noise must be NULL) */
  int32_t flags;                   /* MZX_MOVE_NO_SYNC: return once the download is queued; the caller waits for `stream`
                                      (an event of its own behind the call) before it reads h_out */
  void* h_in;  void* d_in;  int64_t in_bytes;
  void* h_out; void* d_out; int64_t out_bytes;
  mzx_search_io io;
} mzx_move;
#define MZX_MOVE_NO_SYNC 1
/* Draws + upload + mzx_search_run + download + stream synchronisation.  n_legal [B] (out): legal actions per game.
 * On return the outputs are in h_out; the streams have consumed the noise draws but NOT the tie-break words
 * (info[k][2]): a caller that sees the tape-overflow flag (info[k][1] & 1) searches those games again on a longer
 * tape before mzx_selfplay_select. */
int mzx_selfplay_search(mzx_search* s, mzx_rng* r, const mzx_move* m, int32_t* n_legal, void* d_arena,
                        int64_t arena_bytes, void* stream);
/* mzx_rng_advance by words[k], then SelfPlay.select_action for every game: temperature[k] == 0 -> first maximum of
 * the visit counts in legal order; +inf -> randint(0, n_legal[k]); else numpy.random.choice(p = dist / sum(dist)) with
 * dist[j] = pow_table[row(temperature[k])][count_j] -- the caller fills pow_table [num_temperatures][table_stride]
 * with numpy's own integer ** (1 / T) for the distinct finite non-zero temperatures `table_temperatures` (so that
 * the power is the reference's, whatever libm this library was linked with).  visit_counts [B][A] by action (as the
 * search wrote them), action [B] (out): the chosen action. */
int mzx_selfplay_select(mzx_rng* r, const mzx_move* m, const int32_t* n_legal, const int32_t* words,
                        const int32_t* visit_counts, const double* temperature, const double* pow_table,
                        int32_t table_stride, const double* table_temperatures, int32_t num_temperatures,
                        int64_t* action);

/* ------------------------------------------------------------------------- *
 * Replay hand-off: the INITIAL prioritised-replay priorities of finished games on the device --
 * ReplayBuffer.save_game, replay_buffer.py:39-51, with compute_target_value (:230-262) -- for G games of T searched
 * positions each (a shard record: games that started and ended together).  Device arrays:
 *   d_root_values [G][T] f64 (root.value(), 0 for an unvisited root), d_rewards [G][T+1] f64 (reward_history, leading 0),
 *   d_to_play [G][T+1] i32, d_discount_pow [td_steps+1] f64 = config.discount ** i computed by the CALLER with the host
 *   language's own pow (the reference's `self.config.discount**i`);
 *   out: d_targets [G][T] f64 (nullable) = compute_target_value, bit for bit (binary64 multiply / add in the reference's
 *   order); d_priorities [G][T] f32 = float32(|root - target| ** per_alpha) (binary64 sqrt for 0.5, identity for 1, device
 *   pow otherwise); d_game_priority [G] f32 (nullable) = numpy.max(priorities).
 * ------------------------------------------------------------------------- */
int mzx_replay_priorities(const double* d_root_values, const double* d_rewards, const int32_t* d_to_play, int32_t num_games,
                          int32_t moves, int32_t td_steps, const double* d_discount_pow, double per_alpha, double* d_targets,
                          float* d_priorities, float* d_game_priority, void* stream);

/* ------------------------------------------------------------------------- *
 * Games that step NATIVELY for a whole shard (host side, no GPU; csrc/mzx_games.h): the plugin surface of
 * games/abstract_game.py:9-105 -- reset / step / legal_actions / to_play -- for num_games games at once, so that a
 * self-play shard need not return to the interpreter per move (mzx_selfplay_rounds below).  kind: "synthetic" (the
 * fixed-shape environment of the metric, mzx/synthetic.py: observation_shape [C,H,W], action_space_size and num_players
 * are read), "tictactoe" / "connect4" / "gomoku" (games/tictactoe.py:125-310, games/connect4.py:125-300,
 * games/gomoku.py:130-300: geometry fixed, the three size arguments are ignored).  seeds [num_games] (nullable = 0).
 * Observations are float32 [num_games][C*H*W] (what torch.tensor(obs).float() makes of the reference game's arrays);
 * mzx_game_info: out[0..7] = C, H, W, actions, players, dtype of the reference game's observation arrays (0 float32,
 * 1 int32, 2 float64), rewards are integers (1 / 0), every action always legal (1 / 0).  legal_actions: int32
 * [num_games][actions], increasing, padded with -1.  mzx_game_reset(games = NULL): every game.
 * ------------------------------------------------------------------------- */
typedef struct mzx_game mzx_game;
int mzx_game_create(const char* kind, int32_t num_games, const uint32_t* seeds, const int32_t* observation_shape,
                    int32_t action_space_size, int32_t num_players, mzx_game** out);
void mzx_game_destroy(mzx_game* g);
int mzx_game_info(const mzx_game* g, int32_t out[8]);
int mzx_game_reset(mzx_game* g, const int32_t* games, int32_t count);
int mzx_game_observe(const mzx_game* g, float* out);
int mzx_game_legal_actions(const mzx_game* g, int32_t* out);
int mzx_game_to_play(const mzx_game* g, int32_t* out);
int mzx_game_step(mzx_game* g, const int64_t* actions, const uint8_t* active /* nullable: games with 0 stay untouched */,
                  double* reward, uint8_t* done);

/* ------------------------------------------------------------------------- *
 * The self-play ROUND LOOP of a shard of natively stepped games in one call (self_play.py:110-183 for every slot of the
 * shard + the actor loop :31-52; csrc/mzx_actor.h).  An `mzx_actor` is one SLOT GROUP: its game object, search handle,
 * streams of the bank, staging blocks (`move`: as for mzx_selfplay_search; the host-array fields are the actor's own)
 * and the log of its games in progress.  mzx_selfplay_rounds plays rounds -- one move of every slot: search
 * (mzx_selfplay_search), action draw (mzx_selfplay_select, temperature_threshold gate self_play.py:151-157), Game.step,
 * GameHistory row -- until at least `min_games` games have finished or `max_rounds` (< 0: unbounded) rounds were played;
 * a finished game (done, or max_moves moves) is queued for mzx_actor_take and its slot restarts at once on the same
 * stream.  Two groups take turns on the GPU (one searched while the host consumes the other); nothing is in flight when
 * the call returns.  Temperature: `temperature` applies to the games that START during the call; pow_table as for
 * mzx_selfplay_select (every finite non-zero temperature a running game may carry must have a row).  `retry`: called with
 * the games of a group whose search exhausted its tie-break tape (info flag 1); it must search them again on a longer
 * tape and write visit counts / root value / info into the group's output block (rare; NULL: such a search is an error).
 * In / out: `sequence` numbers finished games in the order they finish (across groups and calls).
 * mzx_actor_take hands out and forgets the finished games of a group, game-major and ragged: game j has length[j] moves;
 * observations / actions / rewards / to_play hold length[j] + 1 entries per game (action_history / reward_history with
 * their leading 0, the final position's observation and side to move), visit_counts / root_values / legal_mask length[j];
 * sizes from mzx_actor_finished (out[0] games, out[1] sum of lengths, out[2] = 1 when a game had a restricted legal set:
 * the legal_mask [sum][A] u8 buffer is then required).  `before_sequence` >= 0 restricts both to the games numbered below it
 * (whole rounds: what the call that returned that `sequence` had finished) -- a caller may take the games of call k from
 * one thread while call k + 1 is running on another (mzx.self_play.SelfPlay.continuous_self_play does).
 * ------------------------------------------------------------------------- */
typedef struct mzx_actor mzx_actor;
typedef struct mzx_actor_config {
  mzx_game* game; mzx_search* search; mzx_rng* bank;
  const int32_t* streams;          /* [num_games] stream of game k in the bank (copied) */
  int32_t first_slot;              /* slot of game 0 in the shard (finished games are reported by shard slot) */
  int32_t max_moves;               /* config.max_moves */
  double temperature;              /* of the games that start with the actor */
  void* d_arena; int64_t arena_bytes;
  mzx_move move;                   /* num_games, action_space_size, tape_words, num_threads, dirichlet_alpha, staging blocks, io */
} mzx_actor_config;
int mzx_actor_create(const mzx_actor_config* config, mzx_actor** out);
void mzx_actor_destroy(mzx_actor* a);
typedef int (*mzx_retry_fn)(void* ctx, int32_t group, int32_t count, const int32_t* games);
typedef struct mzx_rounds {
  double temperature;
  int32_t temperature_threshold;   /* config.temperature_threshold, 0 = none */
  int32_t table_stride;
  const double* pow_table; const double* table_temperatures; int32_t num_temperatures; int32_t reserved;
  int64_t min_games, max_rounds;
  int64_t sequence;                /* in / out */
  mzx_retry_fn retry; void* retry_ctx;
  int64_t rounds, games, searches; /* out: rounds played, games finished, searches run (one per slot and round) */
  double search_seconds;           /* out: host time queueing searches + waiting for them */
  double phase_seconds[6];         /* out: where the host time went -- 0 mzx_selfplay_search (draws, staging, upload, launch,
                                      download queued), 1 waiting for a search's event, 2 the per-move region (action draws, log
                                      row, Game.step, next position), 3 finished games copied out + slots restarted, 4 tape
                                      retries, 5 the rest of the call */
} mzx_rounds;
int mzx_selfplay_rounds(mzx_actor* const* groups, int32_t num_groups, mzx_rounds* io, void* stream);
int mzx_actor_finished(mzx_actor* a, int64_t before_sequence, int64_t out[3]);
int mzx_actor_take(mzx_actor* a, int64_t before_sequence, int32_t* slot, int32_t* length, int64_t* sequence, float* observations, int64_t* actions,
                   double* rewards, int64_t* to_play, int32_t* visit_counts, double* root_values, uint8_t* legal_mask,
                   int32_t* any_illegal);

#ifdef __cplusplus
}
#endif
#endif /* MZX_H */
