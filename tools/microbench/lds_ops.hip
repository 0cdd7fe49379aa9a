// LDS-pipe cost of the cross-lane and atomic operations of the linearise kernels, per CU:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_ops.hip -o build/lds_ops && build/lds_ops
// Each test: 256 workgroups x 1024 threads (16 wavefronts per CU, one workgroup per CU), every wavefront issues the operation `reps` times.
// Reported: CU clocks per wavefront-instruction = elapsed * clock / (16 * reps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kReps = 2000;
__device__ __forceinline__ int dpp_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }

template <int kMode>
__global__ __launch_bounds__(1024) void bench(double* out, int cams, int copies_shift) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  double acc = threadIdx.x;
  int iv = threadIdx.x * 7 + 1;
  // a pseudo-random camera per lane, changing per repetition (like co-visible edges: 6 lanes of a point on 6 different cameras)
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
  for (int r = 0; r < kReps; r++) {
    h = h * 1664525u + 1013904223u;
    const int cam = (h >> 8) % cams;
    const int copy = (threadIdx.x >> 3) & ((1 << copies_shift) - 1);
    if (kMode == 0) {                    // 8 x ds_bpermute_b32 (variable source)
#pragma unroll
      for (int k = 0; k < 8; k++) iv = __builtin_amdgcn_ds_bpermute(((lane + k + 1 + (iv & 1)) & 63) << 2, iv);
    } else if (kMode == 1) {             // 8 x ds_add_f64 to the accumulator row of the lane's camera
      double* a = lds + (copy * cams + cam) * 27;
#pragma unroll
      for (int k = 0; k < 8; k++) atomicAdd(&a[k], acc);
    } else if (kMode == 2) {             // 8 x v_mov_dpp wave_shl:1
#pragma unroll
      for (int k = 0; k < 8; k++) iv = dpp_shl1(iv) + 1;
    } else if (kMode == 3) {             // 7 doubles of the lane's camera (the pose gather) as ds_read
      const double* p = lds + cam * 7;
      double s = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) s += p[k];
      acc += s;
    } else if (kMode == 4) {             // 8 x ds_add_f64, every lane its own address (no collisions, no bank conflicts beyond the width)
      double* a = lds + threadIdx.x % 64 + (threadIdx.x >> 6) * 64;
#pragma unroll
      for (int k = 0; k < 8; k++) atomicAdd(&a[k * 1024], acc);
    } else if (kMode == 5) {             // 8 x ds_bpermute_b32 with a constant rotation (what __shfl_down compiles to)
#pragma unroll
      for (int k = 0; k < 8; k++) iv = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, iv) + 1;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + iv;
}

// 8 x ds_add_f64 at a per-lane address given by the host (doubles), the same for every repetition: the bank-conflict model
__global__ __launch_bounds__(1024) void bench_table(double* out, const int* __restrict__ addr) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  double* a = lds + addr[threadIdx.x & 63];
  const double acc = threadIdx.x;
  for (int r = 0; r < kReps; r++) {
#pragma unroll
    for (int k = 0; k < 8; k++) atomicAdd(&a[k], acc);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int run_table(const char* name, const std::vector<int>& addr, double clock_ghz, double* d) {
  int* da; CHECK(hipMalloc(&da, 64 * sizeof(int)));
  CHECK(hipMemcpy(da, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench_table), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  bench_table<<<256, 1024, 131072>>>(d, da);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  bench_table<<<256, 1024, 131072>>>(d, da);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  printf("%-58s %8.3f ms  %7.2f CU-clk per wavefront instruction\n", name, ms, ms * 1e-3 * clock_ghz * 1e9 / (16.0 * kReps * 8));
  CHECK(hipFree(da));
  return 0;
}

template <int kMode>
int run(const char* name, int per_iter, int cams, int copies_shift, double clock_ghz, double* d) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench<kMode>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  bench<kMode><<<256, 1024, 131072>>>(d, cams, copies_shift);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  bench<kMode><<<256, 1024, 131072>>>(d, cams, copies_shift);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double clk = ms * 1e-3 * clock_ghz * 1e9 / (16.0 * kReps * per_iter);
  printf("%-58s %8.3f ms  %7.2f CU-clk per wavefront instruction\n", name, ms, clk);
  return 0;
}

int main() {
  double* d; CHECK(hipMalloc(&d, 256 * 1024 * sizeof(double)));
  int khz = 0; CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double ghz = khz * 1e-6;
  printf("clock %.3f GHz (nominal; the figures scale with the real clock)\n", ghz);
  if (run<0>("ds_bpermute_b32, variable source", 8, 50, 2, ghz, d)) return 1;
  if (run<5>("ds_bpermute_b32, rotation by one", 8, 50, 2, ghz, d)) return 1;
  if (run<2>("v_mov_b32_dpp wave_shl:1 (+ v_add)", 8, 50, 2, ghz, d)) return 1;
  if (run<4>("ds_add_f64, one address per lane", 8, 50, 2, ghz, d)) return 1;
  if (run<1>("ds_add_f64, 50 cameras x 4 copies (8 lanes per copy)", 8, 50, 2, ghz, d)) return 1;
  if (run<1>("ds_add_f64, 50 cameras x 1 copy", 8, 50, 0, ghz, d)) return 1;
  if (run<1>("ds_add_f64, 50 cameras x 8 copies", 8, 50, 3, ghz, d)) return 1;
  if (run<3>("ds_read_b64 x7, pose of a random camera", 7, 60, 2, ghz, d)) return 1;
  {
    std::vector<int> t(64);
    for (int j = 0; j < 64; j++) t[j] = j * 9;                      if (run_table("table: 64 addresses, 2 lanes per 8-byte bank (stride 9)", t, ghz, d)) return 1;
    for (int j = 0; j < 64; j++) t[j] = (j % 16) * 9 + 32 * 9 * (j / 16);   if (run_table("table: 4 lanes per bank, all addresses distinct", t, ghz, d)) return 1;
    for (int j = 0; j < 64; j++) t[j] = (j % 8) * 9 + 32 * 9 * (j / 8);     if (run_table("table: 8 lanes per bank, all addresses distinct", t, ghz, d)) return 1;
    for (int j = 0; j < 64; j++) t[j] = 32 * 9 * j % 16000;          if (run_table("table: 64 lanes in one bank, all addresses distinct", t, ghz, d)) return 1;
    for (int j = 0; j < 64; j++) t[j] = 0;                          if (run_table("table: one address for all lanes", t, ghz, d)) return 1;
    for (int j = 0; j < 64; j++) t[j] = (j / 2) * 9;                 if (run_table("table: pairs of lanes share an address, 32 banks", t, ghz, d)) return 1;
    for (int j = 0; j < 64; j++) t[j] = (j / 4) * 9;                 if (run_table("table: 4 lanes share an address, 16 banks used", t, ghz, d)) return 1;
    for (int j = 0; j < 64; j++) t[j] = (j % 32) * 9;                if (run_table("table: lanes j and j+32 share an address", t, ghz, d)) return 1;
    for (int j = 0; j < 64; j++) t[j] = j * 9 + (j / 32) * 16 * 9;   if (run_table("table: 64 addresses in 64 4-byte-bank pairs?", t, ghz, d)) return 1;
  }
  return 0;
}
