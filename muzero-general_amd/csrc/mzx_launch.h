// mzx_launch.h -- how an element functor (mzx_ops.h) is executed.
//
// Product: one HIP kernel template, one thread per element; 256-thread blocks
// for dense operators, 64-thread (one wavefront) blocks for per-tree operators
// so that B trees spread over B/64 compute units instead of B/256.
// tests/hostcheck: a serial loop (test infrastructure only).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "mzx_platform.h"

namespace mzx {

#ifdef MZX_HOSTCHECK

typedef void* stream_t;

template <int BLOCK, class Op>
inline int launch(const Op& op, stream_t) {
  const size_t n = op.size();
  for (size_t i = 0; i < n; ++i) op(i);
  return 0;
}

inline int copy_h2d(void* dst, const void* src, size_t bytes, stream_t) {
  memcpy(dst, src, bytes);
  return 0;
}

inline int copy_d2d(void* dst, const void* src, size_t bytes, stream_t) {
  memcpy(dst, src, bytes);
  return 0;
}
inline int copy_d2h(void* dst, const void* src, size_t bytes, stream_t) {
  memcpy(dst, src, bytes);
  return 0;
}
inline int stream_sync(stream_t) { return 0; }
// an event recorded behind the work queued on a stream so far (mzx_actor.h: one search in flight per slot group)
inline int event_record(void** ev, stream_t) { *ev = (void*)1; return 0; }
inline int event_wait(void*) { return 0; }
inline void event_destroy(void*) {}
inline int stream_create(void** out) { *out = nullptr; return 0; }
inline void stream_destroy(void*) {}
inline int stream_wait_event(stream_t, void*) { return 0; }

inline int device_alloc(void** out, size_t bytes) {
  *out = malloc(bytes);
  return *out ? 0 : 1;
}
inline void device_free(void* p) { free(p); }
inline int copy_h2d_blocking(void* dst, const void* src, size_t bytes) {
  memcpy(dst, src, bytes);
  return 0;
}

inline const char* runtime_error_string(int) { return "hostcheck"; }
inline int current_device() { return 0; }

#else

typedef hipStream_t stream_t;

template <int BLOCK, class Op>
__global__ void __launch_bounds__(BLOCK) op_kernel(const Op op, const size_t n) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) op(i);
}

template <int BLOCK, class Op>
inline int launch(const Op& op, stream_t stream) {
  const size_t n = op.size();
  if (n == 0) return 0;
  const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL((op_kernel<BLOCK, Op>), dim3(grid), dim3(BLOCK), 0, stream, op, n);
  return (int)hipGetLastError();
}

inline int copy_h2d(void* dst, const void* src, size_t bytes, stream_t stream) {
  return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
}

inline int copy_d2d(void* dst, const void* src, size_t bytes, stream_t stream) {
  return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
}
inline int copy_d2h(void* dst, const void* src, size_t bytes, stream_t stream) {
  return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
}
inline int stream_sync(stream_t stream) { return (int)hipStreamSynchronize(stream); }
// an event recorded behind the work queued on a stream so far (mzx_actor.h: one search in flight per slot group, both
// groups on the caller's stream -- waiting for the stream would wait for the OTHER group's search queued behind)
inline int event_record(void** ev, stream_t stream) {
  if (!*ev) {
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) return (int)rc;
    *ev = (void*)e;
  }
  return (int)hipEventRecord((hipEvent_t)*ev, stream);
}
inline int event_wait(void* ev) { return ev ? (int)hipEventSynchronize((hipEvent_t)ev) : 0; }
inline void event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
// a library-owned stream (the second slot group of a natively played shard searches on its own: two searches of half a shard
// each fill half the chip -- they run side by side instead of one after the other)
inline int stream_create(void** out) {
  hipStream_t st = nullptr;
  const hipError_t rc = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  *out = (void*)st;
  return (int)rc;
}
inline void stream_destroy(void* st) { if (st) (void)hipStreamDestroy((hipStream_t)st); }
inline int stream_wait_event(stream_t stream, void* ev) { return ev ? (int)hipStreamWaitEvent(stream, (hipEvent_t)ev, 0) : 0; }

// small library-owned device allocations (the search's pb_c / sqrt tables); every large buffer is the caller's
inline int device_alloc(void** out, size_t bytes) { return (int)hipMalloc(out, bytes); }
inline void device_free(void* p) { (void)hipFree(p); }
inline int copy_h2d_blocking(void* dst, const void* src, size_t bytes) {
  return (int)hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}

inline const char* runtime_error_string(int e) { return hipGetErrorString((hipError_t)e); }
inline int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }

// Kernels that ask for more than 64 KB of dynamic LDS need hipFuncSetAttribute once PER DEVICE (a process may drive
// several GPUs); `done` is the instantiation's own bit set of devices already served.  Returns a hipError_t.
inline int allow_large_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = uint64_t(1) << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

#endif

}  // namespace mzx
