// Cycles of one call of chol_tile_factor (the panel wave's factorisation of a 16x16 diagonal tile + right-hand side + inverse, the serial
// part of ba_chol_mfma_kernel) on one wavefront, alone on its CU:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -I lld_slam_amd/csrc tools/microbench/chol_panel.hip -o build/chol_panel && build/chol_panel
#include <cstdio>
#include "lld_ba_kernels.h"
using namespace lldba;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void panel(const double* tile, long long* cycles, double* out, int reps) {
  __shared__ double Dg[16 * kCholMStride], Li[16 * kCholMStride], y[16], T0[16 * kCholMStride];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) T0[(i >> 4) * kCholMStride + (i & 15)] = tile[i];
  __syncthreads();
  long long total = 0; bool ok = true;
  for (int r = 0; r < reps; r++) {
    for (int i = lane; i < 256; i += 64) Dg[(i >> 4) * kCholMStride + (i & 15)] = T0[(i >> 4) * kCholMStride + (i & 15)];
    if (lane < 16) y[lane] = 1.0 + lane;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    ok = chol_tile_factor(Dg, Li, y, lane) && ok;
    __syncthreads();
    total += __builtin_readcyclecounter() - t0;
  }
  if (lane == 0) { cycles[0] = total; cycles[1] = ok ? 1 : 0; }
  if (lane < 16) out[lane] = y[lane];
  for (int i = lane; i < 256; i += 64) out[16 + i] = Li[(i >> 4) * kCholMStride + (i & 15)];
}

int main() {
  double h[256];
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1.0 + i + j);
  double* d; long long* c; double* o;
  CHECK(hipMalloc(&d, sizeof(h))); CHECK(hipMalloc(&c, 16)); CHECK(hipMalloc(&o, (16 + 256) * 8));
  CHECK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
  const int reps = 2000;
  for (int k = 0; k < 2; k++) { hipLaunchKernelGGL(panel, dim3(1), dim3(64), 0, 0, d, c, o, reps); CHECK(hipDeviceSynchronize()); }
  long long hc[2]; double ho[16 + 256];
  CHECK(hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost));
  // check: L^-1 (L^-1)^T = S^-1  <=>  S * Linv^T * Linv = I; report the residual of row 0
  double worst = 0.0;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
    double acc = 0.0;                                        // (S Linv^T Linv)[i][j]
    for (int k = 0; k < 16; k++) { double t = 0.0; for (int m = 0; m < 16; m++) t += ho[16 + m * 16 + k] * ho[16 + m * 16 + j]; acc += h[i * 16 + k] * t; }
    const double e = acc - (i == j ? 1.0 : 0.0); if (e < 0 ? -e > worst : e > worst) worst = e < 0 ? -e : e;
  }
  printf("chol_tile_factor: %.0f cycles per call (s_memtime ticks; %d calls, ok %lld), |S Linv^T Linv - I|max %.1e, y[15] %.15g\n", (double)hc[0] / reps, reps, hc[1], worst, ho[15]);
  return 0;
}
