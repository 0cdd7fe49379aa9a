// lld_common.h — shared host-side plumbing of liblld_amd.so (gfx950 only, no CUDA shims, no CPU fallback).
#ifndef LLD_COMMON_H
#define LLD_COMMON_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lld_amd.h"

#define LLD_HIP_TRY(expr)                                                                          \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      std::fprintf(stderr, "[lld_amd] %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__,        \
                   hipGetErrorString(_e));                                                         \
      return LLD_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)

struct lld_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_cu = 256;
  // scratch reused by the single-shot entry points (grown on demand, freed with the context)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // pinned host staging for entry points that move many small arrays in one copy
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // small pinned block the single-window BA polls its progress counters through
  void* poll = nullptr;
  // kernel attributes are per device: remembered per context, not in a process-wide static (a process may hold contexts on several GPUs)
  bool orb_lds_raised = false;
  unsigned pose_lds_raised = 0;            // one bit per pose_opt_kernel instantiation (lld_pose.hip pose_launch_as)
  // Resources of the batched local BA that outlive a batch.  A pipelined caller creates one batch after another on the same
  // context (one context per host thread): allocating a multi-gigabyte slab per batch and, worse, freeing it (hipFree synchronises
  // the device, stalling every other context's solve) was most of the host-buffer rate, and so were pageable uploads.  At most one
  // live batch per context owns the cached set (a second concurrent batch on the same context allocates its own: `busy` is taken with
  // an atomic exchange, so two host threads that create batches on one context at the same time cannot both borrow the slab - a
  // context is still meant to be driven by ONE host thread, see include/lld_amd.h).  lld_ctx_release_cache gives the memory back.
  struct BACache {
    std::atomic<bool> busy{false};           // a live batch holds slab / streams / events / poll block
    void* slab = nullptr; size_t slab_bytes = 0;
    void* stage[2] = {nullptr, nullptr}; size_t stage_bytes[2] = {0, 0};   // pinned upload arenas: [0] flattened inputs, [1] Schur / task structures
    hipEvent_t stage_free = nullptr; bool stage_pending = false;            // the arenas may be rewritten once this event has completed
    void* rec = nullptr; size_t rec_bytes = 0;                              // pinned landing buffer of the result records
    hipStream_t streams[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // group streams 1..7 (group 0 runs on ctx->stream)
    hipEvent_t events[8][4][6] = {};           // [group][super-step of a queued chunk][phase boundary]
    bool attrs_set = false;                  // hipFuncSetAttribute(max dynamic LDS) done for this device
  } ba;
};

// Grow-only pinned host staging on the context.
static inline int lld_ctx_pinned(lld_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->pinned_bytes) {
    if (ctx->pinned) LLD_HIP_TRY(hipHostFree(ctx->pinned));
    ctx->pinned = nullptr; ctx->pinned_bytes = 0;
    size_t want = bytes + (bytes >> 2) + 4096;
    LLD_HIP_TRY(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
  }
  *out = ctx->pinned;
  return LLD_OK;
}

// Grow-only device scratch on the context.
static inline int lld_ctx_scratch(lld_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->scratch_bytes) {
    if (ctx->scratch) LLD_HIP_TRY(hipFree(ctx->scratch));
    ctx->scratch = nullptr; ctx->scratch_bytes = 0;
    size_t want = bytes + (bytes >> 2) + 4096;
    LLD_HIP_TRY(hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return LLD_OK;
}

// Simple bump allocator over one hipMalloc'd slab (all sub-buffers 256-B aligned).
struct lld_slab {
  char* base = nullptr;
  size_t size = 0, used = 0;
  template <class T>
  T* take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    T* p = reinterpret_cast<T*>(base + used);
    used += bytes;
    return p;
  }
  static size_t pad(size_t bytes) { return (bytes + 255) & ~size_t(255); }
};

#endif
