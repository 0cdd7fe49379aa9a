// Issue rate of non-packed 32-bit VALU instructions (v_xor_b32, v_bcnt_u32_b32), fp32 FMA, packed fp32 FMA and fp64 FMA, chip-wide:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rate.hip -o build/valu_rate && build/valu_rate
// 2048 workgroups x 256 threads (8 wavefronts per SIMD in flight), every lane runs kReps iterations of 16 independent chains.
// Reported: T lane-operations per second (= wave-instructions x 64 / time) - the peak the bench's VALU rulers should use.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kReps = 4096;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int kMode>
__global__ __launch_bounds__(256) void bench(unsigned* out, unsigned seed) {
  unsigned a[16]; float f[16]; double d[8]; f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 2654435761u + i * seed; f[i] = 1.0f + 1e-3f * (threadIdx.x + i); }
#pragma unroll
  for (int i = 0; i < 8; i++) { d[i] = 1.0 + 1e-3 * (threadIdx.x + i); p[i] = {f[i], f[i + 8]}; }
  const unsigned k = seed | 1u; const float fk = 1.0f + 1e-7f * seed; const double dk = 1.0 + 1e-9 * seed; const f32x2 pk = {fk, fk};
  for (int r = 0; r < kReps; r++) {
    if (kMode == 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(k));
    } else if (kMode == 1) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(k));
    } else if (kMode == 2) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fk));
    } else if (kMode == 3) {
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pk));
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dk));
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i] + (unsigned)f[i];
#pragma unroll
  for (int i = 0; i < 8; i++) s += (unsigned)d[i] + (unsigned)p[i].x + (unsigned)p[i].y;
  if (s == 0x12345678u) out[0] = s;
}

template <int kMode>
int run(const char* name, int per_iter, int lane_factor) {
  unsigned* out; CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 2048;
  hipLaunchKernelGGL(bench<kMode>, dim3(blocks), dim3(256), 0, 0, out, 3u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(bench<kMode>, dim3(blocks), dim3(256), 0, 0, out, 5u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_instr = (double)blocks * 4 * kReps * per_iter;
  printf("%-16s %8.3f ms  %7.1f G wave-instructions/s  %6.1f T lane-operations/s\n", name, ms, wave_instr / ms / 1e6, wave_instr * 64 * lane_factor / ms / 1e9);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  if (run<0>("v_xor_b32", 16, 1)) return 1;
  if (run<1>("v_bcnt_u32_b32", 16, 1)) return 1;
  if (run<2>("v_fma_f32", 16, 1)) return 1;
  if (run<3>("v_pk_fma_f32", 8, 2)) return 1;
  if (run<4>("v_fma_f64", 8, 1)) return 1;
  return 0;
}
