// counter_calib.hip - what do rocprofv3's FETCH_SIZE / WRITE_SIZE say for access patterns with a KNOWN byte count?  (the
// roofline's counter ruler doubles FETCH_SIZE everywhere, the guide calibrates that factor for 16-B-per-lane streams only.)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/counter_calib.hip -o tools/microbench/counter_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o f -- tools/microbench/counter_calib     (and once more with WRITE_SIZE)
// Every kernel touches 2 GiB (8x the Infinity Cache) exactly once; the kernel name says the pattern and tools/counter_calib_summary.py
// divides the counter by the bytes the pattern must move:
//   read_stream16 / 8 / 4      coalesced streaming reads, 16 / 8 / 4 bytes per lane
//   read_gather8_blocks        doubles gathered through an index list (the BA kernels' pattern: a 4-byte index, then an 8-byte double) where
//                              the list is a random permutation of aligned 128-byte blocks: every fetched line is used in full - 12 B per element
//   read_gather8_random        the same with a fully random list: 8 useful bytes per fetched sector (reported as the gather's inflation)
//   write_stream16 / 8         coalesced streaming writes
//   write_pairs16_blocks       16-byte pairs written through an index list that permutes aligned 128-byte blocks (+ 4 B of index read per pair)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void read_stream16(const double2* a, size_t n, double* sink) { double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; } if (s == 12345.678) *sink = s; }
__global__ void read_stream8(const double* a, size_t n, double* sink) { double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i]; if (s == 12345.678) *sink = s; }
__global__ void read_stream4(const float* a, size_t n, double* sink) { float s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i]; if (s == 12345.678f) *sink = s; }
__global__ void read_gather8_blocks(const double* a, const uint32_t* idx, size_t n, double* sink) { double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[idx[i]]; if (s == 12345.678) *sink = s; }
__global__ void read_gather8_random(const double* a, const uint32_t* idx, size_t n, double* sink) { double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[idx[i]]; if (s == 12345.678) *sink = s; }
__global__ void write_stream16(double2* a, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_double2((double)i, 1.0); }
__global__ void write_stream8(double* a, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i; }
__global__ void write_pairs16_blocks(double2* a, const uint32_t* idx, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[idx[i]] = make_double2((double)i, 2.0); }

int main() {
  const size_t bytes = (size_t)2 << 30;
  void* buf; CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 0, bytes));
  double* sink; CHECK(hipMalloc(&sink, 8));
  const size_t n8 = bytes / 8, n16 = bytes / 16;
  // index lists: block permutations (16 doubles = 128 B per block; 8 pairs = 128 B per block) and a fully random one
  std::mt19937_64 rng(1);
  std::vector<uint32_t> h8(n8), h8r(n8), h16(n16);
  { std::vector<uint32_t> blocks(n8 / 16); for (size_t b = 0; b < blocks.size(); b++) blocks[b] = (uint32_t)b; std::shuffle(blocks.begin(), blocks.end(), rng);
    for (size_t i = 0; i < n8; i++) h8[i] = blocks[i / 16] * 16 + (uint32_t)(i % 16); }
  { for (size_t i = 0; i < n8; i++) h8r[i] = (uint32_t)(rng() % n8); }
  { std::vector<uint32_t> blocks(n16 / 8); for (size_t b = 0; b < blocks.size(); b++) blocks[b] = (uint32_t)b; std::shuffle(blocks.begin(), blocks.end(), rng);
    for (size_t i = 0; i < n16; i++) h16[i] = blocks[i / 8] * 8 + (uint32_t)(i % 8); }
  uint32_t *d8, *d8r, *d16;
  CHECK(hipMalloc(&d8, n8 * 4)); CHECK(hipMalloc(&d8r, n8 * 4)); CHECK(hipMalloc(&d16, n16 * 4));
  CHECK(hipMemcpy(d8, h8.data(), n8 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d8r, h8r.data(), n8 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d16, h16.data(), n16 * 4, hipMemcpyHostToDevice));
  const dim3 g(256 * 16), b(256);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(read_stream16, g, b, 0, 0, (const double2*)buf, n16, sink);
    hipLaunchKernelGGL(read_stream8, g, b, 0, 0, (const double*)buf, n8, sink);
    hipLaunchKernelGGL(read_stream4, g, b, 0, 0, (const float*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(read_gather8_blocks, g, b, 0, 0, (const double*)buf, d8, n8, sink);
    hipLaunchKernelGGL(read_gather8_random, g, b, 0, 0, (const double*)buf, d8r, n8 / 8, sink);      // (an eighth of the elements: every one costs a whole sector)
    hipLaunchKernelGGL(write_stream16, g, b, 0, 0, (double2*)buf, n16);
    hipLaunchKernelGGL(write_stream8, g, b, 0, 0, (double*)buf, n8);
    hipLaunchKernelGGL(write_pairs16_blocks, g, b, 0, 0, (double2*)buf, d16, n16);
  }
  CHECK(hipDeviceSynchronize());
  printf("counter_calib: 2 GiB per kernel, two repetitions; read_gather8_random touches %zu elements\n", n8 / 8);
  return 0;
}
