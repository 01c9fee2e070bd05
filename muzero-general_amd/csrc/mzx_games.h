// mzx_games.h -- games that step NATIVELY for a whole shard (host code, no GPU): the environments a self-play shard plays
// without returning to the interpreter per move (mzx_selfplay_rounds, mzx_actor.h).
//
// What they restate.  The reference steps one `Game` object per actor in Python (self_play.py:129-181:
// legal_actions / step / to_play per move, games/abstract_game.py:9-105).  mzx.games ships three of its board games in
// the batched plugin protocol (numpy arithmetic over [B][cells]); these are the same games -- and the synthetic
// fixed-shape environment of the metric (mzx/synthetic.py) -- behind a C ABI:
//   "synthetic"  mzx.synthetic.make_synthetic_batched_game: next observation = counter hash of (key, action, t), reward =
//                one hash bit, every action always legal, never ends before max_moves;
//   "tictactoe"  games/tictactoe.py:125-310   3 x 3, three in a row, int32 planes, a win pays 20;
//   "connect4"   games/connect4.py:125-300    6 x 7 with gravity, four in a row, float64 planes, a win pays 10;
//   "gomoku"     games/gomoku.py:130-300      11 x 11, five in a row, float64 planes, 1 is paid whenever the game ends.
// Observations are emitted as float32 (what torch.tensor(obs).float() makes of them, self_play.py:280-285; board planes
// hold -1 / 0 / 1: exact); `obs_dtype` names the dtype the reference game returns so that the host mirror hands out
// GameHistory.observation_history in it.  Game for game identical to the Python classes: tests/test_native_games.py.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

struct mzx_game {
  int32_t num_games = 0, num_actions = 0, num_players = 1;
  int32_t shape[3] = {0, 0, 0};
  int32_t obs_dtype = 0;       // 0 float32, 1 int32, 2 float64 (the reference game's array dtype)
  int32_t reward_is_int = 1;   // every shipped game pays integers
  int32_t always_all_legal = 0;
  virtual ~mzx_game() {}
  int64_t obs_elems() const { return (int64_t)shape[0] * shape[1] * shape[2]; }
  // games [lo, hi) -- every method is safe to call on disjoint ranges from several threads
  virtual void reset(const int32_t* idx, int32_t count) = 0;                  // idx == nullptr: games 0 .. count - 1
  virtual void observe(int lo, int hi, float* out) const = 0;                  // out: row g at out + g * obs_elems()
  virtual void legal_actions(int lo, int hi, int32_t* out) const = 0;          // [B][A] increasing, padded with -1
  virtual void to_play(int lo, int hi, int32_t* out) const = 0;
  // active (nullable): games with active[g] == 0 are left untouched (reward 0, not done) -- the lock-step loop of
  // SelfPlay.play_games keeps stepping a shard whose shorter games have ended
  virtual void step(int lo, int hi, const int64_t* actions, const uint8_t* active, double* reward, uint8_t* done) = 0;
};

namespace mzx {

inline uint32_t game_hash_u32(uint32_t x) {     // mzx/synthetic.py _hash_u32 (wrapping 32-bit arithmetic)
  x = (x ^ 61u) ^ (x >> 16);
  x *= 9u;
  x ^= x >> 4;
  x *= 0x27D4EB2Du;
  x ^= x >> 15;
  return x;
}

// mzx/synthetic.py make_synthetic_batched_game
struct SyntheticGame : mzx_game {
  std::vector<uint32_t> seeds, key, t;
  std::vector<int32_t> player;

  uint32_t first_key(uint32_t seed64_low) const { return game_hash_u32(seed64_low * 7919u + 17u); }

  void reset(const int32_t* idx, int32_t count) override {
    for (int32_t k = 0; k < count; ++k) {
      const int g = idx ? idx[k] : k;
      t[g] = 0; player[g] = 0;
      // (seeds * 7919 + 17 in uint64, truncated to uint32: the same value modulo 2**32)
      key[g] = first_key(seeds[g]);
    }
  }
  void observe(int lo, int hi, float* out) const override {
    const int64_t E = obs_elems();
    for (int g = lo; g < hi; ++g) {
      float* o = out + (int64_t)g * E;
      for (int64_t j = 0; j < E; ++j) {
        const uint32_t lane = (uint32_t)((uint64_t)j * 2654435761ull);
        const uint32_t h = game_hash_u32(lane + key[g]);
        o[j] = (float)((double)h / 4294967296.0);
      }
    }
  }
  void legal_actions(int lo, int hi, int32_t* out) const override {
    for (int g = lo; g < hi; ++g)
      for (int a = 0; a < num_actions; ++a) out[(int64_t)g * num_actions + a] = a;
  }
  void to_play(int lo, int hi, int32_t* out) const override {
    for (int g = lo; g < hi; ++g) out[g] = player[g];
  }
  void step(int lo, int hi, const int64_t* actions, const uint8_t* /*active: the Python class steps every game*/, double* reward,
            uint8_t* done) override {
    for (int g = lo; g < hi; ++g) {
      t[g] += 1u;
      uint32_t k = key[g] * 31u;
      k += (uint32_t)actions[g] * 131u;
      k += t[g];
      key[g] = game_hash_u32(k);
      if (num_players > 1) player[g] = (player[g] + 1) % num_players;
      reward[g] = (double)(key[g] & 1u);
      done[g] = 0;
    }
  }
};

// mzx/games.py _KInARowBatched
struct KInARowGame : mzx_game {
  int rows = 0, cols = 0, k = 0;
  bool gravity = false, end_pays = false;
  int reward_scale = 1;
  std::vector<int8_t> board;        // [B][rows * cols]: 0 empty, 1 / -1
  std::vector<int8_t> player;       // [B]: 1 / -1
  std::vector<int32_t> lines;       // [L][k] flat cell indices

  void build_lines() {
    static const int dirs[4][2] = {{0, 1}, {1, 0}, {1, 1}, {1, -1}};
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c)
        for (const auto& d : dirs) {
          const int rr = r + (k - 1) * d[0], cc = c + (k - 1) * d[1];
          if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
          for (int i = 0; i < k; ++i) lines.push_back((r + i * d[0]) * cols + (c + i * d[1]));
        }
  }
  void reset(const int32_t* idx, int32_t count) override {
    const int cells = rows * cols;
    for (int32_t q = 0; q < count; ++q) {
      const int g = idx ? idx[q] : q;
      memset(&board[(size_t)g * cells], 0, (size_t)cells);
      player[g] = 1;
    }
  }
  void observe(int lo, int hi, float* out) const override {
    const int cells = rows * cols;
    for (int g = lo; g < hi; ++g) {
      float* o = out + (int64_t)g * 3 * cells;
      const int8_t* b = &board[(size_t)g * cells];
      for (int j = 0; j < cells; ++j) {
        o[j] = b[j] == 1 ? 1.f : 0.f;
        o[cells + j] = b[j] == -1 ? 1.f : 0.f;
        o[2 * cells + j] = (float)player[g];
      }
    }
  }
  void legal_actions(int lo, int hi, int32_t* out) const override {
    const int cells = rows * cols, A = num_actions;
    for (int g = lo; g < hi; ++g) {
      const int8_t* b = &board[(size_t)g * cells];
      int32_t* o = out + (int64_t)g * A;
      int n = 0;
      for (int a = 0; a < A; ++a) {
        const bool free_ = gravity ? b[(rows - 1) * cols + a] == 0 : b[a] == 0;
        if (free_) o[n++] = a;
      }
      for (; n < A; ++n) o[n] = -1;
    }
  }
  void to_play(int lo, int hi, int32_t* out) const override {
    for (int g = lo; g < hi; ++g) out[g] = player[g] == 1 ? 0 : 1;
  }
  void step(int lo, int hi, const int64_t* actions, const uint8_t* active, double* reward, uint8_t* done) override {
    const int cells = rows * cols;
    const int L = (int)(lines.size() / (size_t)k);
    for (int g = lo; g < hi; ++g) {
      if (active && !active[g]) { reward[g] = 0.0; done[g] = 0; continue; }
      int8_t* b = &board[(size_t)g * cells];
      const int a = (int)actions[g];
      const int8_t me = player[g];
      if (gravity) {             // the lowest empty cell of the column; the reference's loop places nothing in a full one
        for (int r = 0; r < rows; ++r)
          if (b[r * cols + a] == 0) { b[r * cols + a] = me; break; }
      } else {
        b[a] = me;
      }
      bool won = false;
      for (int l = 0; l < L && !won; ++l) {
        bool all = true;
        for (int i = 0; i < k && all; ++i) all = b[lines[(size_t)l * k + i]] == me;
        won = all;
      }
      bool no_move = true;
      if (gravity) { for (int c = 0; c < cols && no_move; ++c) no_move = b[(rows - 1) * cols + c] != 0; }
      else { for (int j = 0; j < cells && no_move; ++j) no_move = b[j] != 0; }
      const bool over = won || no_move;
      done[g] = over ? 1 : 0;
      reward[g] = ((end_pays ? over : won) ? (double)reward_scale : 0.0);
      player[g] = (int8_t)-me;
    }
  }
};

// kind, geometry -> a game object (nullptr + message on a bad argument)
inline mzx_game* game_make(const char* kind, int32_t num_games, const uint32_t* seeds, const int32_t shape[3], int32_t num_actions,
                           int32_t num_players, std::string& err) {
  if (!kind || num_games < 1) { err = "mzx_game_create: kind / num_games"; return nullptr; }
  const std::string name(kind);
  if (name == "synthetic") {
    if (!shape || shape[0] < 1 || shape[1] < 1 || shape[2] < 1 || num_actions < 1 || num_players < 1) {
      err = "mzx_game_create(synthetic): observation shape, actions and players must be positive";
      return nullptr;
    }
    SyntheticGame* g = new SyntheticGame();
    g->num_games = num_games; g->num_actions = num_actions; g->num_players = num_players;
    g->shape[0] = shape[0]; g->shape[1] = shape[1]; g->shape[2] = shape[2];
    g->obs_dtype = 0; g->always_all_legal = 1;
    g->seeds.resize(num_games); g->key.resize(num_games); g->t.assign(num_games, 0); g->player.assign(num_games, 0);
    for (int i = 0; i < num_games; ++i) g->seeds[i] = seeds ? seeds[i] : 0u;
    g->reset(nullptr, num_games);
    return g;
  }
  KInARowGame* g = nullptr;
  if (name == "tictactoe") { g = new KInARowGame(); g->rows = 3; g->cols = 3; g->k = 3; g->reward_scale = 20; g->obs_dtype = 1; }
  else if (name == "connect4") { g = new KInARowGame(); g->rows = 6; g->cols = 7; g->k = 4; g->gravity = true; g->reward_scale = 10; g->obs_dtype = 2; }
  else if (name == "gomoku") { g = new KInARowGame(); g->rows = 11; g->cols = 11; g->k = 5; g->reward_scale = 1; g->obs_dtype = 2; g->end_pays = true; }
  else { err = "mzx_game_create: unknown game '" + name + "' (synthetic, tictactoe, connect4, gomoku)"; return nullptr; }
  g->num_games = num_games; g->num_players = 2;
  g->num_actions = g->gravity ? g->cols : g->rows * g->cols;
  g->shape[0] = 3; g->shape[1] = g->rows; g->shape[2] = g->cols;
  g->board.assign((size_t)num_games * g->rows * g->cols, 0);
  g->player.assign(num_games, 1);
  g->build_lines();
  return g;
}

}  // namespace mzx
