// lld_ba_multi.hip — the multi-GPU split of the batched local BA behind the C ABI (north_star: "independent
// local-BA windows shard embarrassingly across the 8 GPUs of one node ... only for the final gather ... while the Tracking / LocalMapping
// host threads stay C++").  One process, one host thread + one context (lld_ctx: device, stream, cached slab) per shard:
//   create   block partition of the windows over the shards (lld_ba_multi_shard = shard(strong=True) of lld_slam_amd/dist.py), every shard
//            flattens and uploads its windows to its own device, concurrently
//   solve    every shard runs lld_ba_batch_solve on its own host thread (nothing is exchanged during the solve); when a shard is done its
//            fixed-stride result records travel to ONE buffer on the first shard's device with hipMemcpyPeerAsync - the peer-to-peer copy
//            over xGMI that an RCCL send / recv pair of the same size performs, without a second RCCL in processes that already hold
//            torch's - record k of the whole batch at k x stride: the only exchange
//   verify   the first device checks what arrived (dist.verify_gathered_records): win_index and point-edge count of every header against
//            the window the partition put there, a finished protocol
// A device may be listed more than once (two shards share it): that is how the 1-GPU test box exercises partition, threads and gather.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <thread>

#include "lld_common.h"

namespace {
struct RecHeader { double chi2_round1, chi2_final; int lm_iterations[2], lm_trials[2]; int pcg_iterations, aborted, win_index, n_pt_obs; };   // = lldba::BARecordHeader
static_assert(sizeof(RecHeader) == 48, "record header layout (lld_ba_kernels.h, lld_slam_amd/dist.py)");
}

struct lld_ba_multi {
  struct Shard { int device = 0, first = 0, count = 0; lld_ctx* ctx = nullptr; lld_ba_batch* batch = nullptr; void* rec = nullptr; uint64_t stride = 0; int status = LLD_OK; };
  std::vector<Shard> shards;
  int n_windows = 0;
  std::vector<int> n_pt_obs;            // per window, for the identity check of a gathered record
  void* gathered = nullptr;             // on shards[0].device: n_windows x stride
  uint64_t stride = 0;
  double solve_ms = 0.0, gather_ms = 0.0;
};

extern "C" {

int lld_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

void lld_ba_multi_shard(int32_t n_windows, int32_t n_parts, int32_t part, int32_t* first, int32_t* count) {
  int32_t lo = 0, n = 0;
  if (n_windows >= 0 && n_parts >= 1 && part >= 0 && part < n_parts) {
    lo = (int32_t)((int64_t)n_windows * part / n_parts);
    n = (int32_t)((int64_t)n_windows * (part + 1) / n_parts) - lo;
  }
  if (first) *first = lo;
  if (count) *count = n;
}

void lld_ba_multi_destroy(lld_ba_multi* m) {
  if (!m) return;
  for (auto& s : m->shards) {
    if (s.batch) lld_ba_batch_destroy(s.batch);
    if (s.ctx) lld_ctx_destroy(s.ctx);
  }
  if (m->gathered && !m->shards.empty()) { (void)hipSetDevice(m->shards[0].device); (void)hipFree(m->gathered); }
  delete m;
}

int lld_ba_multi_create(int32_t n_devices, const int32_t* devices, int32_t n_windows, const lld_ba_window* windows, const lld_ba_params* params, lld_ba_multi** out) {
  if (!out || n_devices < 1 || n_devices > 64 || !devices || n_windows < n_devices || !windows) return LLD_ERR_INVALID;
  *out = nullptr;
  lld_ba_multi* m = new lld_ba_multi();
  m->n_windows = n_windows;
  m->shards.resize((size_t)n_devices);
  m->n_pt_obs.resize((size_t)n_windows);
  for (int w = 0; w < n_windows; w++) m->n_pt_obs[(size_t)w] = windows[w].n_pt_obs;
  std::vector<std::thread> pool;
  for (int d = 0; d < n_devices; d++) {
    lld_ba_multi::Shard& s = m->shards[(size_t)d];
    s.device = devices[d];
    lld_ba_multi_shard(n_windows, n_devices, d, &s.first, &s.count);
    pool.emplace_back([&s, windows, params]() {
      s.status = lld_ctx_create(s.device, &s.ctx);
      if (s.status == LLD_OK) s.status = lld_ba_batch_create(s.ctx, s.count, windows + s.first, params, &s.batch);
      if (s.status == LLD_OK) s.status = lld_ba_batch_result_records(s.batch, &s.rec, &s.stride);
    });
  }
  for (auto& t : pool) t.join();
  for (auto& s : m->shards) {
    if (s.status != LLD_OK) { const int st = s.status; lld_ba_multi_destroy(m); return st; }
    m->stride = std::max(m->stride, s.stride);
  }
  // every other shard's device may write into the first one's memory (xGMI peer access; a shard on the same device needs none, and where
  // the platform refuses, hipMemcpyPeerAsync stages through the host by itself)
  const int dev0 = m->shards[0].device;
  for (auto& s : m->shards)
    if (s.device != dev0) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, s.device, dev0) == hipSuccess && can) { (void)hipSetDevice(s.device); (void)hipDeviceEnablePeerAccess(dev0, 0); (void)hipGetLastError(); }
    }
  if (hipSetDevice(dev0) != hipSuccess || hipMalloc(&m->gathered, (size_t)m->stride * (size_t)n_windows + 256) != hipSuccess) { lld_ba_multi_destroy(m); return LLD_ERR_ALLOC; }
  *out = m;
  return LLD_OK;
}

int lld_ba_multi_solve(lld_ba_multi* m, volatile const int* abort_flag) {
  if (!m) return LLD_ERR_INVALID;
  const int dev0 = m->shards[0].device;
  char* dst0 = static_cast<char*>(m->gathered);
  const uint64_t stride = m->stride;
  std::vector<std::thread> pool;
  std::vector<double> t_solve(m->shards.size(), 0.0), t_gather(m->shards.size(), 0.0);
  for (size_t d = 0; d < m->shards.size(); d++) {
    lld_ba_multi::Shard& s = m->shards[d];
    pool.emplace_back([&s, &t_solve, &t_gather, d, abort_flag, dev0, dst0, stride]() {
      const auto t0 = std::chrono::steady_clock::now();
      s.status = lld_ba_batch_solve(s.batch, abort_flag);
      const auto t1 = std::chrono::steady_clock::now();
      if (s.status != LLD_OK) return;
      if (hipSetDevice(s.device) != hipSuccess) { s.status = LLD_ERR_HIP; return; }
      hipStream_t st = static_cast<hipStream_t>(lld_ctx_stream(s.ctx));
      hipError_t e = hipSuccess;
      char* dst = dst0 + (size_t)stride * (size_t)s.first;
      if (s.stride == stride) e = hipMemcpyPeerAsync(dst, dev0, s.rec, s.device, (size_t)s.stride * (size_t)s.count, st);
      else                                                 // a shard whose largest record is smaller than the batch's: record by record at the common stride
        for (int k = 0; k < s.count && e == hipSuccess; k++)
          e = hipMemcpyPeerAsync(dst + (size_t)stride * (size_t)k, dev0, static_cast<char*>(s.rec) + (size_t)s.stride * (size_t)k, s.device, (size_t)s.stride, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) { std::fprintf(stderr, "[lld_amd] gather of shard %zu failed: %s\n", d, hipGetErrorString(e)); s.status = LLD_ERR_HIP; return; }
      const auto t2 = std::chrono::steady_clock::now();
      t_solve[d] = std::chrono::duration<double, std::milli>(t1 - t0).count(); t_gather[d] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    });
  }
  for (auto& t : pool) t.join();
  m->solve_ms = 0.0; m->gather_ms = 0.0;
  for (size_t d = 0; d < m->shards.size(); d++) {
    if (m->shards[d].status != LLD_OK) return m->shards[d].status;
    m->solve_ms = std::max(m->solve_ms, t_solve[d]); m->gather_ms = std::max(m->gather_ms, t_gather[d]);
  }
  return LLD_OK;
}

int lld_ba_multi_result_records(lld_ba_multi* m, void** dev_ptr, uint64_t* stride_bytes, int32_t* device) {
  if (!m) return LLD_ERR_INVALID;
  if (dev_ptr) *dev_ptr = m->gathered;
  if (stride_bytes) *stride_bytes = m->stride;
  if (device) *device = m->shards[0].device;
  return LLD_OK;
}

int lld_ba_multi_times_ms(lld_ba_multi* m, double* slowest_solve_ms, double* slowest_gather_ms) {
  if (!m) return LLD_ERR_INVALID;
  if (slowest_solve_ms) *slowest_solve_ms = m->solve_ms;
  if (slowest_gather_ms) *slowest_gather_ms = m->gather_ms;
  return LLD_OK;
}

int lld_ba_multi_download(lld_ba_multi* m, int32_t window, lld_ba_result* out) {
  if (!m || window < 0 || window >= m->n_windows || !out) return LLD_ERR_INVALID;
  for (auto& s : m->shards)
    if (window >= s.first && window < s.first + s.count) return lld_ba_batch_download(s.batch, window - s.first, out);
  return LLD_ERR_INVALID;
}

int lld_ba_multi_verify_gathered(lld_ba_multi* m, int32_t* n_checked) {
  if (!m) return LLD_ERR_INVALID;
  if (n_checked) *n_checked = 0;
  std::vector<RecHeader> h((size_t)m->n_windows);
  LLD_HIP_TRY(hipSetDevice(m->shards[0].device));
  LLD_HIP_TRY(hipMemcpy2D(h.data(), sizeof(RecHeader), m->gathered, (size_t)m->stride, sizeof(RecHeader), (size_t)m->n_windows, hipMemcpyDeviceToHost));
  for (auto& s : m->shards)
    for (int k = 0; k < s.count; k++) {
      const RecHeader& r = h[(size_t)(s.first + k)];
      const bool finished = r.aborted != 0 || (std::isfinite(r.chi2_final) && r.chi2_final > 0 && r.lm_iterations[0] >= 1);
      if (r.win_index != k || r.n_pt_obs != m->n_pt_obs[(size_t)(s.first + k)] || !finished) {
        std::fprintf(stderr, "[lld_amd] gathered record %d is not window %d of shard on device %d (win_index %d, n_pt_obs %d, chi2 %g)\n", s.first + k, k, s.device, r.win_index, r.n_pt_obs, r.chi2_final);
        return LLD_ERR_INVALID;
      }
      if (n_checked) (*n_checked)++;
    }
  return LLD_OK;
}

}  // extern "C"
