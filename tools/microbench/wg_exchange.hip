// wg_exchange.hip - what does it cost G workgroups of ONE kernel to exchange 28 doubles each and all see the sum?  (round 6: could one frame's
// PoseOptimization spread over several CUs?  It needs such an exchange per linearisation and per LM trial.)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/wg_exchange.hip -o tools/microbench/wg_exchange && tools/microbench/wg_exchange
// Pattern: every workgroup writes its 28 partials (agent-scope relaxed atomic stores: they bypass the non-coherent caches), waits for the
// stores, raises its flag to the epoch number; every workgroup polls the G flags, then reads the G x 28 partials and adds them in a fixed
// order.  Slots are double buffered by epoch parity.  `stride`: workgroup w of the cluster is block w * stride (8: all on one XCD, 1: spread
// over the XCDs); the other blocks of the grid exit at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int K = 28;
__global__ __launch_bounds__(512) void exchange_kernel(double* slots, int* flags, int G, int stride, int epochs, long long* cycles, double* out) {
  if (blockIdx.x % stride != 0) return;
  const int w = blockIdx.x / stride;
  if (w >= G) return;
  const int tid = threadIdx.x;
  __shared__ double tot[K];
  double acc = 0.0;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int e = 1; e <= epochs; e++) {
    double* my = slots + ((size_t)(e & 1) * G + w) * K;
    if (tid < K) __hip_atomic_store(&my[tid], (double)(w + 1) * e + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);                               // the partials have left this CU
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flags[w * 32], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < G) { while (__hip_atomic_load(&flags[tid * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {} }
    __syncthreads();
    if (tid < K) {
      double s = 0.0;
      for (int g = 0; g < G; g++) s += __hip_atomic_load(&slots[((size_t)(e & 1) * G + g) * K + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tot[tid] = s;
    }
    __syncthreads();
    acc += tot[tid % K];
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  if (tid == 0) { cycles[w] = t1 - t0; out[w] = acc; }
}

int main() {
  double* slots; int* flags; long long* cycles; double* out;
  hipMalloc(&slots, sizeof(double) * 2 * 64 * K); hipMalloc(&flags, sizeof(int) * 64 * 32); hipMalloc(&cycles, sizeof(long long) * 64); hipMalloc(&out, sizeof(double) * 64);
  const int epochs = 2000;
  for (int stride : {8, 1}) {
    for (int G : {1, 2, 4, 8, 16}) {
      hipMemset(flags, 0, sizeof(int) * 64 * 32); hipMemset(slots, 0, sizeof(double) * 2 * 64 * K);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(exchange_kernel, dim3((G - 1) * stride + 1), dim3(512), 0, 0, slots, flags, G, stride, epochs, cycles, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms = 0; hipEventElapsedTime(&ms, a, b);
      std::vector<long long> c(G); hipMemcpy(c.data(), cycles, sizeof(long long) * G, hipMemcpyDeviceToHost);
      std::vector<double> o(G); hipMemcpy(o.data(), out, sizeof(double) * G, hipMemcpyDeviceToHost);
      bool same = true; for (int g = 1; g < G; g++) same = same && o[g] == o[0];
      std::printf("stride %d  G %2d : %.3f us per exchange by HIP events (%d epochs), %.0f s_memtime ticks per exchange on workgroup 0; all workgroups saw the same sums: %s\n",
                  stride, G, 1e3 * ms / epochs, epochs, (double)c[0] / epochs, same ? "yes" : "NO");
    }
  }
  return 0;
}
