// mzx_rng.h -- host-side bank of numpy-compatible random streams (one per game of a shard).
//
// The reference draws from numpy's process-global legacy stream (MT19937 `RandomState`), once per
// move in this order (self_play.py): Dirichlet root noise (:473, numpy.random.dirichlet), the
// arg-max tie breaks of the search (:371, numpy.random.choice(list)) and the action sample
// (:229-243, numpy.random.choice(actions[, p])).  A shard of B games needs B such streams; driving
// B Python `RandomState` objects costs ~100 us per game and move, three orders of magnitude more
// than the search itself.  This bank holds the B generator states natively and produces, for all
// games of a move at once (optionally on several host threads), exactly the numbers the reference's
// calls would produce: a restatement of numpy's LEGACY algorithms (numpy/random/mtrand.pyx,
// src/legacy/legacy-distributions.c, src/mt19937/mt19937.c -- third-party, not vendored by the
// reference; the legacy stream is frozen by NumPy policy, NEP 19):
//   seeding            RandomState(seed) for 0 <= seed < 2^32: init_genrand
//   double             (a >> 5, b >> 6) -> (a * 67108864 + b) / 9007199254740992
//   standard_exponential  -log(1 - double)
//   standard_gamma     shape < 1: Ahrens-Dieter / Best rejection; shape > 1: Marsaglia-Tsang with the
//                      cached polar gauss; shape == 1: exponential
//   dirichlet          gammas, then multiplication by 1 / sum
//   randint(0, n) / choice(list of n)   masked rejection on 32-bit words
// libm's log / pow / sqrt are the same functions numpy itself calls on this host.
// tests/test_rng_bank.py pins every entry point against numpy.random.RandomState.
#pragma once
#include <unistd.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace mzx {

struct Mt19937 {
  uint32_t key[624];
  int32_t pos;
  int32_t has_gauss;
  double gauss;

  void seed(uint32_t s) {  // mt19937_seed / init_genrand
    s &= 0xffffffffu;
    for (int i = 0; i < 624; ++i) {
      key[i] = s;
      s = (1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u) & 0xffffffffu;
    }
    pos = 624;
    has_gauss = 0;
    gauss = 0.0;
  }
  void gen() {  // mt19937_gen
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int i;
    uint32_t y;
    for (i = 0; i < 624 - 397; ++i) {
      y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; i < 623; ++i) {
      y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (key[623] & UPPER) | (key[0] & LOWER);
    key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    pos = 0;
  }
  inline uint32_t next32() {
    if (pos == 624) gen();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  static inline uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  // The next `n` raw words WITHOUT advancing the stream (the device draws the tie breaks of a search
  // from them; the host then advances by what was consumed).  No copy of the 2.5 KB state unless the
  // window crosses a regeneration of the key array.
  void peek(int n, uint32_t* out) const {
    if (pos + n <= 624) {
      for (int j = 0; j < n; ++j) out[j] = temper(key[pos + j]);
      return;
    }
    Mt19937 copy = *this;
    for (int j = 0; j < n; ++j) out[j] = copy.next32();
  }
  inline double next_double() {
    const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  inline double standard_exponential() { return -log(1.0 - next_double()); }
  double gauss_polar() {  // legacy_gauss
    if (has_gauss) {
      const double t = gauss;
      gauss = 0.0;
      has_gauss = 0;
      return t;
    }
    double f, x1, x2, r2;
    do {
      x1 = 2.0 * next_double() - 1.0;
      x2 = 2.0 * next_double() - 1.0;
      r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    gauss = f * x1;
    has_gauss = 1;
    return f * x2;
  }
  double standard_gamma(double shape) {  // legacy_standard_gamma
    double b, c, U, V, X, Y;
    if (shape == 1.0) return standard_exponential();
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
      for (;;) {
        U = next_double();
        V = standard_exponential();
        if (U <= 1.0 - shape) {
          X = pow(U, 1. / shape);
          if (X <= V) return X;
        } else {
          Y = -log((1 - U) / shape);
          X = pow(1.0 - shape + shape * Y, 1. / shape);
          if (X <= (V + Y)) return X;
        }
      }
    }
    b = shape - 1. / 3.;
    c = 1. / sqrt(9 * b);
    for (;;) {
      do {
        X = gauss_polar();
        V = 1.0 + c * X;
      } while (V <= 0.0);
      V = V * V * V;
      U = next_double();
      if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return (b * V);
      if (log(U) < 0.5 * X * X + b * (1. - V + log(V))) return (b * V);
    }
  }
  // randint(0, n) / choice(list of n), n >= 1: masked rejection (n == 1 consumes nothing)
  inline uint32_t bounded(uint32_t n) {
    const uint32_t rng = n - 1;
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (next32() & mask)) > rng) {}
    return v;
  }
};

}  // namespace mzx

namespace mzx {

// A few persistent worker threads: the per-move draws of a shard (4 096 Dirichlet samples, ~2 us each) are worth
// spreading over cores, but creating and joining std::threads on every move costs more than the draws themselves
// (~30 us per thread).  A self-play round is two or three such regions of 20 - 100 us each, a fraction of a millisecond
// apart (mzx_actor.h), so a worker that finished its piece SPINS on the generation counter for a short while before it goes
// to sleep on the condition variable: waking sixteen sleepers through the futex cost more than the region they were woken
// for (round 6: the native round loop measured 0.5 ms of a 0.69 ms round in pool hand-offs).  Between calls (no region for
// ~100 us) the workers sleep.
class RngPool {
 public:
  explicit RngPool(int n) {
    for (int t = 0; t < n; ++t) workers_.emplace_back([this, t]() { loop(t); });
  }
  ~RngPool() {
    stop_.store(true, std::memory_order_seq_cst);
    generation_.fetch_add(1, std::memory_order_seq_cst);
    { std::lock_guard<std::mutex> lk(m_); }
    cv_.notify_all();
    for (std::thread& th : workers_) th.join();
  }
  int size() const { return (int)workers_.size(); }
  // fn(lo, hi) over [0, count) split into size() + 1 pieces (the caller's thread takes one)
  void run(int count, const std::function<void(int, int)>& fn) {
    const int parts = size() + 1, per = (count + parts - 1) / parts;
    fn_ = &fn; count_ = count; per_ = per;
    pending_.store(size(), std::memory_order_relaxed);
    generation_.fetch_add(1, std::memory_order_seq_cst);       // publishes fn_ / count_ / per_
    { std::lock_guard<std::mutex> lk(m_); }                      // (a worker between its last check and its wait holds m_)
    cv_.notify_all();
    const int lo = size() * per;
    if (lo < count) fn(lo, count);
    for (int spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins) {
      if (spins < 4096) cpu_relax();
      else std::this_thread::yield();
    }
  }

 private:
  static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
  }
  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      // wait for the next region: spin first, then sleep
      int spins = 0;
      while (generation_.load(std::memory_order_acquire) == seen) {
        if (++spins < 20000) { cpu_relax(); continue; }
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&]() { return generation_.load(std::memory_order_seq_cst) != seen; });
      }
      seen = generation_.load(std::memory_order_acquire);
      if (stop_.load(std::memory_order_acquire)) return;
      const int lo = t * per_, hi = lo + per_ < count_ ? lo + per_ : count_;
      if (lo < hi) (*fn_)(lo, hi);
      pending_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<bool> stop_{false};
  std::atomic<uint64_t> generation_{0};
  std::atomic<int> pending_{0};
  int count_ = 0, per_ = 0;
  const std::function<void(int, int)>* fn_ = nullptr;
};

}  // namespace mzx

struct mzx_rng {
  std::vector<mzx::Mt19937> streams;
  std::unique_ptr<mzx::RngPool> pool;   // created on the first large parallel call
  long pool_pid = 0;                    // process that created it: worker threads do not survive fork()
  std::mutex pool_owner;                // ONE caller at a time drives the pool (two slot groups of a shard call into the
                                        // bank from two threads: the search's draws on a worker, the action draw on the main one)
};

namespace mzx {

template <class Fn>
inline void rng_parallel(mzx_rng* r, int count, int n_threads, Fn fn) {
  if (n_threads <= 1 || count < 256) { fn(0, count); return; }
  // the pool serves one call at a time; a caller that finds it busy (another thread's call on OTHER streams of the bank)
  // does its own range itself instead of waiting -- disjoint streams, no shared state
  std::unique_lock<std::mutex> owner(r->pool_owner, std::try_to_lock);
  if (!owner.owns_lock()) { fn(0, count); return; }
  if (n_threads > 64) n_threads = 64;
  const long pid = (long)getpid();
  if (r->pool && r->pool_pid != pid) (void)r->pool.release();   // forked child: the parent's threads are not here; never join them
  if (!r->pool || r->pool->size() != n_threads - 1) { r->pool.reset(new RngPool(n_threads - 1)); r->pool_pid = pid; }
  const std::function<void(int, int)> f = fn;
  r->pool->run(count, f);
}

}  // namespace mzx
