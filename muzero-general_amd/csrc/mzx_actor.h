// mzx_actor.h -- the self-play ROUND LOOP of a shard of natively stepped games (host code; one call plays many moves).
//
// Reference: SelfPlay.play_game, /root/reference/self_play.py:110-183 -- per move: stacked observation, MCTS.run,
// select_action (:222-245, temperature_threshold gate :151-157), Game.step, GameHistory appends (:159-177), until done or
// max_moves (:129) -- and the actor loop around it (:31-52: the next game starts the moment one ends).  mzx.self_play
// plays B such actors as one shard (SelfPlay.play_rounds: every slot is one actor with its own numpy stream, finished
// games hand their slot to the next one, every search runs at full width); through a Python plugin game that is ~75
// interpreter statements per slot group and round around two library calls.  For a game that steps natively (mzx_games.h)
// the whole loop runs here: per round and slot group
//     begin    legal_actions / to_play / observation of the group's games -> mzx_selfplay_search (root noise + tie words
//              straight into the pinned staging block, ONE upload, the search, ONE download, not waited for)
//     consume  wait for that search's event; tape-overflow retries (callback); mzx_selfplay_select (the streams consume
//              the search's tie words, every game's action is drawn); the move is logged as one row per field of a ring
//              [round % capacity][slot]; Game.step; games that ended (done, or max_moves moves) are copied out of the ring
//              into the finished queue, their slots restart (Game.reset on the slot, same stream)
// and two slot groups take turns on the GPU exactly as SelfPlay._rounds_batched schedules them (the search of one group
// is in flight while the host consumes the other; a search is only queued ahead when the call cannot end before it is
// consumed, so nothing is in flight between calls -- weights may change there).  Same draws in the same order, same
// games field for field as the Python loop: tests/test_native_rounds.py.
//
// Not here (the Python loop keeps them): stacked observations (the device frame store), Python plugin games, opponents.
#pragma once
#include <chrono>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

#include "mzx_games.h"

struct mzx_actor {
  mzx_game* game = nullptr;
  mzx_search* search = nullptr;
  mzx_rng* bank = nullptr;
  void* d_arena = nullptr;
  int64_t arena_bytes = 0;
  mzx_move move{};                       // staging blocks + io; the host-array fields point into this object's vectors
  int32_t B = 0, A = 0, max_moves = 0;
  int64_t E = 0;                         // observation elements per game
  int32_t first_slot = 0;                // slot of game 0 in the shard (finished games report shard slots)
  std::vector<int32_t> streams, legal, to_play, n_legal, words;
  std::vector<float> cur_obs, next_obs;
  std::vector<int64_t> actions, start;
  std::vector<double> temps, move_temps, reward;
  std::vector<uint8_t> done;
  int64_t round = 0;
  bool pending = false;
  void* event = nullptr;                 // recorded behind the queued download of the search in flight
  void* own_stream = nullptr;            // groups behind the first search on a stream of their own (created on first use)
  void* start_event = nullptr;           // orders that stream behind what the caller queued before the call
  // the log of the games in progress: ring rows [round % cap][slot]
  int64_t cap = 0;
  std::vector<float> r_obs;
  std::vector<int32_t> r_tp, r_vis;
  std::vector<int64_t> r_act;
  std::vector<double> r_rew, r_val;
  std::vector<uint8_t> r_mask;           // [cap][B][A], only for games whose legal sets vary
  // finished games, batch by batch (the games one consume() ended), game-major ragged arrays
  struct Batch {
    std::vector<int32_t> slot, n;
    std::vector<int64_t> seq;
    std::vector<float> obs;              // sum(n + 1) x E
    std::vector<int64_t> act, tp;        // sum(n + 1)
    std::vector<double> rew;             // sum(n + 1)
    std::vector<int32_t> vis;            // sum(n) x A
    std::vector<double> val;             // sum(n)
    std::vector<uint8_t> mask;           // sum(n) x A, empty when every action was legal throughout
  };
  // (taken by mzx_actor_take, possibly from another thread while a rounds call is appending: the queue has its own lock)
  std::deque<Batch> finished;
  std::mutex finished_lock;
};

namespace mzx {

inline double actor_now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline void actor_grow_ring(mzx_actor* a, int64_t need) {
  if (need <= a->cap) return;
  const int64_t limit = (int64_t)a->max_moves + 1;
  int64_t cap = std::max<int64_t>(std::max<int64_t>(2 * a->cap, need), std::min<int64_t>(limit, 32));
  cap = std::min<int64_t>(cap, std::max<int64_t>(limit, need));
  const int64_t B = a->B, A = a->A, E = a->E;
  std::vector<float> obs((size_t)(cap * B * E));
  std::vector<int32_t> tp((size_t)(cap * B)), vis((size_t)(cap * B * A));
  std::vector<int64_t> act((size_t)(cap * B));
  std::vector<double> rew((size_t)(cap * B)), val((size_t)(cap * B));
  std::vector<uint8_t> mask(a->game->always_all_legal ? 0 : (size_t)(cap * B * A));
  if (a->cap > 0) {
    int64_t lo = a->round;
    for (int64_t s = 0; s < B; ++s) lo = std::min(lo, a->start[s]);
    for (int64_t q = lo; q < a->round; ++q) {
      const int64_t o = q % a->cap, n = q % cap;
      memcpy(&obs[(size_t)(n * B * E)], &a->r_obs[(size_t)(o * B * E)], sizeof(float) * (size_t)(B * E));
      memcpy(&tp[(size_t)(n * B)], &a->r_tp[(size_t)(o * B)], sizeof(int32_t) * (size_t)B);
      memcpy(&vis[(size_t)(n * B * A)], &a->r_vis[(size_t)(o * B * A)], sizeof(int32_t) * (size_t)(B * A));
      memcpy(&act[(size_t)(n * B)], &a->r_act[(size_t)(o * B)], sizeof(int64_t) * (size_t)B);
      memcpy(&rew[(size_t)(n * B)], &a->r_rew[(size_t)(o * B)], sizeof(double) * (size_t)B);
      memcpy(&val[(size_t)(n * B)], &a->r_val[(size_t)(o * B)], sizeof(double) * (size_t)B);
      if (!mask.empty()) memcpy(&mask[(size_t)(n * B * A)], &a->r_mask[(size_t)(o * B * A)], (size_t)(B * A));
    }
  }
  a->r_obs.swap(obs); a->r_tp.swap(tp); a->r_vis.swap(vis); a->r_act.swap(act); a->r_rew.swap(rew); a->r_val.swap(val);
  a->r_mask.swap(mask);
  a->cap = cap;
}

// positions of the group's games as the next search reads them
inline void actor_refresh(mzx_actor* a) {
  mzx_game* g = a->game;
  rng_parallel(a->bank, a->B, a->move.num_threads, [=](int lo, int hi) {
    g->legal_actions(lo, hi, a->legal.data());
    g->to_play(lo, hi, a->to_play.data());
  });
}

inline int actor_begin(mzx_actor* a, void* stream, double* search_seconds, double* phase) {
  const double t0 = actor_now();
  mzx_move& m = a->move;
  m.streams = a->streams.data();
  m.legal_actions = a->legal.data();
  m.to_play = a->to_play.data();
  m.observation = a->cur_obs.data();
  m.observation_floats = a->E;
  m.flags = MZX_MOVE_NO_SYNC;
  int rc = mzx_selfplay_search(a->search, a->bank, &m, a->n_legal.data(), a->d_arena, a->arena_bytes, stream);
  if (rc) return rc;
  rc = event_record(&a->event, (stream_t)stream);
  if (rc) { set_error("mzx_selfplay_rounds: event record failed: %s", runtime_error_string(rc)); return MZX_ERR_RUNTIME; }
  a->pending = true;
  const double dt = actor_now() - t0;
  *search_seconds += dt;
  phase[0] += dt;
  return MZX_OK;
}

// copies the games of slots idx (ascending) that ended with round r out of the ring
inline void actor_harvest(mzx_actor* a, const std::vector<int32_t>& idx, int64_t r, int64_t* sequence) {
  const int64_t B = a->B, A = a->A, E = a->E, cap = a->cap;
  mzx_actor::Batch b;
  int64_t rows = 0;
  for (int32_t s : idx) rows += r - a->start[s] + 1;
  const int64_t k = (int64_t)idx.size();
  b.slot.resize(k); b.n.resize(k); b.seq.resize(k);
  b.obs.resize((size_t)((rows + k) * E)); b.act.resize((size_t)(rows + k)); b.tp.resize((size_t)(rows + k));
  b.rew.resize((size_t)(rows + k)); b.vis.resize((size_t)(rows * A)); b.val.resize((size_t)rows);
  const bool masks = !a->r_mask.empty();
  bool any_illegal = false;
  if (masks) b.mask.resize((size_t)(rows * A));
  std::vector<int64_t> off1((size_t)k), off0((size_t)k);     // offsets of game j in the (n + 1)-long and n-long arrays
  int64_t o1 = 0, o0 = 0;
  for (int64_t j = 0; j < k; ++j) {
    const int32_t s = idx[j];
    const int64_t n = r - a->start[s] + 1;
    b.slot[j] = a->first_slot + s; b.n[j] = (int32_t)n; b.seq[j] = (*sequence)++;
    off1[j] = o1; off0[j] = o0;
    o1 += n + 1; o0 += n;
  }
  mzx_actor::Batch* bp = &b;
  std::vector<uint8_t> flag((size_t)k, 0);
  uint8_t* flagp = flag.data();
  const int64_t* p1 = off1.data();
  const int64_t* p0 = off0.data();
  const int32_t* idxp = idx.data();
  rng_parallel(a->bank, (int)k, a->move.num_threads, [=](int lo, int hi) {
    for (int j = lo; j < hi; ++j) {
      const int32_t s = idxp[j];
      const int64_t st = a->start[s], n = r - st + 1, q1 = p1[j], q0 = p0[j];
      bp->act[(size_t)q1] = 0; bp->rew[(size_t)q1] = 0.0;        // action_history / reward_history start with 0 (self_play.py:118-120)
      for (int64_t t = 0; t < n; ++t) {
        const int64_t c = (st + t) % cap;
        memcpy(&bp->obs[(size_t)((q1 + t) * E)], &a->r_obs[(size_t)((c * B + s) * E)], sizeof(float) * (size_t)E);
        bp->tp[(size_t)(q1 + t)] = a->r_tp[(size_t)(c * B + s)];
        bp->act[(size_t)(q1 + t + 1)] = a->r_act[(size_t)(c * B + s)];
        bp->rew[(size_t)(q1 + t + 1)] = a->r_rew[(size_t)(c * B + s)];
        memcpy(&bp->vis[(size_t)((q0 + t) * A)], &a->r_vis[(size_t)((c * B + s) * A)], sizeof(int32_t) * (size_t)A);
        bp->val[(size_t)(q0 + t)] = a->r_val[(size_t)(c * B + s)];
        if (masks) {
          const uint8_t* mk = &a->r_mask[(size_t)((c * B + s) * A)];
          memcpy(&bp->mask[(size_t)((q0 + t) * A)], mk, (size_t)A);
          for (int64_t x = 0; x < A; ++x) if (!mk[x]) flagp[j] = 1;
        }
      }
      // the position after the last move: its observation and side to move close the record (self_play.py:166-168)
      memcpy(&bp->obs[(size_t)((q1 + n) * E)], &a->next_obs[(size_t)(s * E)], sizeof(float) * (size_t)E);
      bp->tp[(size_t)(q1 + n)] = a->to_play[s];
    }
  });
  for (int64_t j = 0; j < k; ++j) any_illegal = any_illegal || flag[(size_t)j];
  if (masks && !any_illegal) { b.mask.clear(); b.mask.shrink_to_fit(); }
  std::lock_guard<std::mutex> lk(a->finished_lock);
  a->finished.push_back(std::move(b));
}

typedef int (*actor_retry_fn)(void* ctx, int32_t group, int32_t count, const int32_t* games);

struct RoundsArgs {
  double* phase;               // [6] accumulators (mzx_rounds::phase_seconds)
  double temperature;
  int32_t temperature_threshold;
  const double* pow_table; int32_t table_stride; const double* table_temperatures; int32_t num_temperatures;
  actor_retry_fn retry; void* retry_ctx;
};

inline int actor_consume(mzx_actor* a, int group, const RoundsArgs& args, int64_t* sequence, int64_t* games_done,
                         double* search_seconds) {
  const double t0 = actor_now();
  if (!a->pending) { set_error("mzx_selfplay_rounds: no search in flight for group %d", group); return MZX_ERR_INVALID; }
  int rc = event_wait(a->event);
  if (rc) { set_error("mzx_selfplay_rounds: waiting for the search failed: %s", runtime_error_string(rc)); return MZX_ERR_RUNTIME; }
  a->pending = false;
  double t1 = actor_now();
  *search_seconds += t1 - t0;
  args.phase[1] += t1 - t0;
  const int B = a->B, A = a->A;
  const int64_t E = a->E;
  const mzx_move& m = a->move;
  const char* h_out = (const char*)m.h_out;
  auto host_of = [&](const void* d) { return h_out + ((const char*)d - (const char*)m.d_out); };
  int32_t* info = (int32_t*)host_of(m.io.d_info);
  const int32_t* visits = (const int32_t*)host_of(m.io.d_visit_counts);
  const double* root_value = (const double*)host_of(m.io.d_root_value);
  // a tree that exhausted its tie-break tape is searched again on a longer one (rare: equal priors at every level); the
  // host mirror owns that path (another staging geometry) and writes the results into this group's output block
  std::vector<int32_t> redo;
  for (int k = 0; k < B; ++k) if (info[4 * k + 1] & 1) redo.push_back(k);
  if (!redo.empty()) {
    if (!args.retry) { set_error("mzx_selfplay_rounds: a search exhausted its tie-break tape and no retry callback was given"); return MZX_ERR_RUNTIME; }
    rc = args.retry(args.retry_ctx, group, (int32_t)redo.size(), redo.data());
    if (rc) { set_error("mzx_selfplay_rounds: the tape-retry callback failed (%d)", rc); return MZX_ERR_RUNTIME; }
    const double t2 = actor_now();
    args.phase[4] += t2 - t1;
    t1 = t2;
  }
  for (int k = 0; k < B; ++k)
    if (info[4 * k + 1] != 0) { set_error("search flagged tree %d (flags %d): node arena exhausted", k, info[4 * k + 1]); return MZX_ERR_RUNTIME; }
  const int64_t r = a->round;
  for (int k = 0; k < B; ++k) {
    a->words[k] = info[4 * k + 2];
    const int64_t moves_before = r - a->start[k];
    a->move_temps[k] = (args.temperature_threshold && moves_before + 1 >= args.temperature_threshold) ? 0.0 : a->temps[k];   // self_play.py:151-157
  }
  if (A > 4096) { set_error("mzx_selfplay_rounds: more than 4096 actions"); return MZX_ERR_INVALID; }
  rc = select_check(a->legal.data(), a->n_legal.data(), visits, a->move_temps.data(), B, A, args.table_stride, args.table_temperatures,
                    args.num_temperatures);
  if (rc) return rc;
  int64_t lo_start = r;
  for (int k = 0; k < B; ++k) lo_start = std::min(lo_start, a->start[k]);
  actor_grow_ring(a, r - lo_start + 1);
  const int64_t c = r % a->cap;
  SelectArgs sa;
  sa.r = a->bank; sa.idx = a->streams.data(); sa.legal_all = a->legal.data(); sa.n_legal = a->n_legal.data(); sa.words = a->words.data();
  sa.visit_counts = visits; sa.temperature = a->move_temps.data(); sa.pow_table = args.pow_table; sa.table_stride = args.table_stride;
  sa.table_temperatures = args.table_temperatures; sa.num_temperatures = args.num_temperatures; sa.action = a->actions.data(); sa.A = A;
  mzx_game* g = a->game;
  const bool masks = !a->r_mask.empty();
  const int max_moves = a->max_moves;
  // ONE region per move and slot group: per game [lo, hi) the action draw (the stream consumes the search's tie words first),
  // the move's row of the log -- position searched (observation, side to move), action, search statistics, legal set --,
  // Game.step, the reward, and the next position (observation, side to move, legal actions) for the next search
  rng_parallel(a->bank, B, m.num_threads, [=](int lo, int hi) {
    select_range(sa, lo, hi);
    const size_t n = (size_t)(hi - lo);
    memcpy(&a->r_obs[(size_t)((c * B + lo) * E)], &a->cur_obs[(size_t)lo * E], sizeof(float) * n * (size_t)E);
    memcpy(&a->r_tp[(size_t)(c * B + lo)], &a->to_play[lo], sizeof(int32_t) * n);
    memcpy(&a->r_act[(size_t)(c * B + lo)], &a->actions[lo], sizeof(int64_t) * n);
    memcpy(&a->r_vis[(size_t)((c * B + lo) * A)], visits + (size_t)lo * A, sizeof(int32_t) * n * (size_t)A);
    memcpy(&a->r_val[(size_t)(c * B + lo)], root_value + lo, sizeof(double) * n);
    if (masks) {
      uint8_t* mk = &a->r_mask[(size_t)((c * B + lo) * A)];
      memset(mk, 0, n * (size_t)A);
      for (int k = lo; k < hi; ++k)
        for (int j = 0; j < a->n_legal[k]; ++j) mk[(size_t)(k - lo) * A + a->legal[(size_t)k * A + j]] = 1;
    }
    g->step(lo, hi, a->actions.data(), nullptr, a->reward.data(), a->done.data());
    memcpy(&a->r_rew[(size_t)(c * B + lo)], &a->reward[lo], sizeof(double) * n);
    g->observe(lo, hi, a->next_obs.data());
    g->to_play(lo, hi, a->to_play.data());
    g->legal_actions(lo, hi, a->legal.data());
    for (int k = lo; k < hi; ++k)
      if ((r - a->start[k]) + 2 > max_moves) a->done[k] = 1;        // len(action_history) <= max_moves, self_play.py:129
  });
  {
    const double t2 = actor_now();
    args.phase[2] += t2 - t1;
    t1 = t2;
  }
  std::vector<int32_t> over;
  for (int k = 0; k < B; ++k)
    if (a->done[k]) over.push_back(k);
  if (!over.empty()) {
    actor_harvest(a, over, r, sequence);
    *games_done += (int64_t)over.size();
    g->reset(over.data(), (int32_t)over.size());          // the slots' next games begin (self_play.py:31-52)
    // first positions of the restarted slots, run by run of consecutive slots
    for (size_t q = 0; q < over.size();) {
      size_t e = q + 1;
      while (e < over.size() && over[e] == over[e - 1] + 1) ++e;
      const int lo = over[q], hi = over[e - 1] + 1;
      g->observe(lo, hi, a->next_obs.data());
      g->to_play(lo, hi, a->to_play.data());
      g->legal_actions(lo, hi, a->legal.data());
      q = e;
    }
    for (int32_t s : over) {
      a->start[s] = r + 1;
      a->temps[s] = args.temperature;
    }
  }
  a->cur_obs.swap(a->next_obs);
  args.phase[3] += actor_now() - t1;
  a->round = r + 1;
  return MZX_OK;
}

}  // namespace mzx
