// lld_match.hip — descriptor matching kernels (gfx950).
//
//   hamming256_best2   ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663) + the best/second-best loops of
//                      the Search* family (e.g. :76-114): lane <-> query (two queries per lane), the four waves of a
//                      workgroup split a train tile that is staged through LDS and read back as wave-uniform
//                      (broadcast) ds_read_b128; xor + v_bcnt accumulate per word; running (best, second) kept as
//                      packed (distance, order) keys so that the lexicographic minimum reproduces the reference's
//                      strict '<' ("first candidate wins ties"); cross-wave merge through LDS.
//   l2f32_best2        LineMatcher::MatchLineDescriptors argmin loops (src/TwoFrameLineMatcher.cc:112, Tracking.cc:1092,
//                      1532) with the build-defined float-L2 distance (parity unpinned: LBDMOD is not vendored).
//   line_greedy        sequential masking of TwoFrameLineMatcher::MatchLines (src/TwoFrameLineMatcher.cc:39-67).
#include <algorithm>
#include <cmath>

#include "lld_common.h"
#include "lld_track_internal.h"

namespace {

constexpr int kWave = 64;
constexpr int kMatchThreads = 256;             // 4 waves
constexpr int kQPerLane = 4;                   // queries per lane: one pair of broadcast LDS reads feeds kQPerLane distance computations
constexpr int kQPerBlock = kQPerLane * 64;     // all 4 waves see the same 256 queries and split the train rows
constexpr int kTileRows = 256;                 // train rows staged per LDS tile (8 KiB)
constexpr unsigned kIdxBits = 22;              // packed key = dist << 22 | order  (order < 4 Mi)
constexpr unsigned kKeyEmpty = (256u << kIdxBits) | ((1u << kIdxBits) - 1u);

__device__ __forceinline__ void key_update(unsigned key, unsigned& best, unsigned& second) {
  // lexicographic top-2: second = min(second, max(best, key)); best = min(best, key)
  // with best <= second the new second is the median of the three (one v_med3_u32 instead of a max and a min)
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(second) : "v"(best), "v"(second), "v"(key));
  best = min(best, key);
}

// v_bcnt_u32_b32 D = popcount(S0) + S1: the eight words of a descriptor pair are counted in one accumulating chain.  Written as asm
// because the compiler prefers eight independent counts and three v_add3_u32 (shorter chains, 3 more instructions per pair); the brute-
// force kernel has four independent queries per lane to fill the chain's latency and is bound by VALU issue.
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
  unsigned d;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
  return d;
}
__device__ __forceinline__ unsigned hamming8(const uint4& qa, const uint4& qb, const uint4& ta, const uint4& tb) {
  unsigned d = __popc(qa.x ^ ta.x);
  d = bcnt_acc(qa.y ^ ta.y, d); d = bcnt_acc(qa.z ^ ta.z, d); d = bcnt_acc(qa.w ^ ta.w, d);
  d = bcnt_acc(qb.x ^ tb.x, d); d = bcnt_acc(qb.y ^ tb.y, d); d = bcnt_acc(qb.z ^ tb.z, d); d = bcnt_acc(qb.w ^ tb.w, d);
  return d;
}

// grid (ceil(nq / kQPerBlock), batch); block 256.
__global__ __launch_bounds__(kMatchThreads) void hamming256_best2_kernel(
    const uint4* __restrict__ q, int nq, const uint4* __restrict__ t, int nt, const uint8_t* __restrict__ mask,
    int* __restrict__ best_idx, int* __restrict__ best_dist, int* __restrict__ second_idx, int* __restrict__ second_dist) {
  __shared__ uint4 tile[kTileRows * 2];                       // 8 KiB train tile
  __shared__ unsigned merge[4][kQPerBlock][2];                // per-wave (best, second) keys
  const int pair = blockIdx.y;
  q += (size_t)pair * nq * 2; t += (size_t)pair * nt * 2;
  const size_t out_off = (size_t)pair * nq;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  uint4 qa[kQPerLane], qb[kQPerLane];
  unsigned best[kQPerLane], sec[kQPerLane];
  const uint8_t* mrow[kQPerLane];
#pragma unroll
  for (int k = 0; k < kQPerLane; k++) {
    const int qk = blockIdx.x * kQPerBlock + k * kWave + lane;
    const bool v = qk < nq;
    qa[k] = v ? q[qk * 2] : make_uint4(0, 0, 0, 0); qb[k] = v ? q[qk * 2 + 1] : make_uint4(0, 0, 0, 0);
    best[k] = kKeyEmpty; sec[k] = kKeyEmpty;
    mrow[k] = mask ? mask + ((size_t)pair * nq + (v ? qk : 0)) * nt : nullptr;
  }
  for (int base = 0; base < nt; base += kTileRows) {
    const int rows = min(kTileRows, nt - base);
    __syncthreads();                                           // previous tile fully consumed
    for (int i = threadIdx.x; i < rows * 2; i += kMatchThreads) tile[i] = t[(size_t)base * 2 + i];   // 16 B / lane, coalesced
    __syncthreads();
    // wave w takes rows w, w+4, ... of the tile; LDS reads are wave-uniform (broadcast, conflict-free)
    for (int r = wave; r < rows; r += 4) {
      const uint4 ta = tile[r * 2], tb = tile[r * 2 + 1];
      const unsigned j = (unsigned)(base + r);
#pragma unroll
      for (int k = 0; k < kQPerLane; k++) {
        unsigned key = (hamming8(qa[k], qb[k], ta, tb) << kIdxBits) | j;
        if (mask && !mrow[k][j]) key = kKeyEmpty;
        key_update(key, best[k], sec[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kQPerLane; k++) { merge[wave][k * kWave + lane][0] = best[k]; merge[wave][k * kWave + lane][1] = sec[k]; }
  __syncthreads();
  for (int qo = threadIdx.x; qo < kQPerBlock; qo += kMatchThreads) {
    const int qi = blockIdx.x * kQPerBlock + qo;
    if (qi < nq) {
      unsigned b = kKeyEmpty, s = kKeyEmpty;
      for (int w = 0; w < 4; w++) { key_update(merge[w][qo][0], b, s); key_update(merge[w][qo][1], b, s); }
      const unsigned idx_mask = (1u << kIdxBits) - 1u;
      best_idx[out_off + qi] = (b == kKeyEmpty) ? -1 : (int)(b & idx_mask);
      best_dist[out_off + qi] = (int)(b >> kIdxBits);
      second_idx[out_off + qi] = (s == kKeyEmpty) ? -1 : (int)(s & idx_mask);
      second_dist[out_off + qi] = (int)(s >> kIdxBits);
    }
  }
}

// Candidate-list form: one lane per query walks its own list (Frame::GetFeaturesInArea / BoW node order); the list
// position is the tie-break key, the output is the train index.
__global__ __launch_bounds__(256) void hamming256_csr_kernel(
    const uint4* __restrict__ q, int nq, const uint4* __restrict__ t, const int* __restrict__ cand_start, const int* __restrict__ cand_idx,
    int* __restrict__ best_idx, int* __restrict__ best_dist, int* __restrict__ second_idx, int* __restrict__ second_dist) {
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= nq) return;
  const uint4 a = q[qi * 2], b = q[qi * 2 + 1];
  int bd = 256, bi = -1, sd = 256, si = -1;
  const int s = cand_start[qi], e = cand_start[qi + 1];
  for (int k = s; k < e; k++) {
    const int j = cand_idx[k];
    const int d = (int)hamming8(a, b, t[j * 2], t[j * 2 + 1]);
    if (d < bd) { sd = bd; si = bi; bd = d; bi = j; }
    else if (d < sd) { sd = d; si = j; }
  }
  best_idx[qi] = bi; best_dist[qi] = bd; second_idx[qi] = si; second_dist[qi] = sd;
}

// ------------------------------------------------------------------ float L2 (LBD)
// d = sqrt( sum_i (double)(a_i - b_i)^2 ), float difference, double accumulation in ascending i (the product of two
// floats is exact in double, so the fused multiply-add below rounds exactly like mul-then-add on the CPU).
// One wavefront per workgroup: the exact kernel holds a 72-float query and two rows' sums in 192 VGPRs (two wavefronts per SIMD), and the
// 300 queries of a frame pair are 4.7 wavefronts - in workgroups of 256 lanes the second one kept three idle wavefronts resident beside its
// single working one (5 working wavefronts in 8 slots).  Each wavefront now stages the train tile for itself (86 KB per pair and
// wavefront out of L2): 826 k -> 1.2 M frame pairs/s.
constexpr int kL2Threads = 64;
constexpr int kL2TileRows = 64;

// kExact: dim == DIM_MAX, known at compile time.  With a run-time dim every component of the unrolled sums sits behind its own `i < dim`
// branch: one ds_read_b32 per component, scalar registers spilled into lanes, 3158 instructions of which 225 are the FMAs.  The
// descriptor lengths in use (72: LBD, 32) therefore get the exact form, other lengths the guarded one.
template <int DIM_MAX, bool kExact>
__global__ __launch_bounds__(kL2Threads) void l2f32_best2_kernel(
    const float* __restrict__ q, int nq, const float* __restrict__ t, int nt, int dim_arg, const uint8_t* __restrict__ mask,
    int* __restrict__ best_idx, double* __restrict__ best_dist, int* __restrict__ second_idx, double* __restrict__ second_dist,
    double* __restrict__ dist_matrix /* optional [nq][nt] */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);                        // [kL2TileRows][dim]
  const int dim = kExact ? DIM_MAX : dim_arg;
  const int pair = blockIdx.y;
  q += (size_t)pair * nq * dim; t += (size_t)pair * nt * dim;
  const size_t out_off = (size_t)pair * nq;
  const int qi = blockIdx.x * kL2Threads + threadIdx.x;
  const bool valid = qi < nq;
  float qa[DIM_MAX];
#pragma unroll
  for (int i = 0; i < DIM_MAX; i++) qa[i] = (valid && (kExact || i < dim)) ? q[(size_t)qi * dim + i] : 0.f;
  double bd = 1.7976931348623157e308, sd = 1.7976931348623157e308;
  int bi = -1, si = -1;
  // (with workgroups of more than one wavefront: a wavefront without a single query only helps to stage the tiles)
  const bool wave_has_queries = blockIdx.x * kL2Threads + (threadIdx.x & ~63) < nq;
#define LLD_L2_TAKE(DIST, J)                                                                              \
  do {                                                                                                    \
    const double dist_ = (DIST); const int j_ = (J);                                                      \
    if (valid) {                                                                                          \
      if (dist_matrix) dist_matrix[((size_t)pair * nq + qi) * nt + j_] = dist_;                           \
      const bool cand_ = !mask || mask[((size_t)pair * nq + qi) * nt + j_];                               \
      if (cand_) {                                                                                        \
        if (dist_ < bd) { sd = bd; si = bi; bd = dist_; bi = j_; }                                        \
        else if (dist_ < sd) { sd = dist_; si = j_; }                                                     \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
  for (int base = 0; base < nt; base += kL2TileRows) {
    const int rows = min(kL2TileRows, nt - base);
    __syncthreads();
    for (int i = threadIdx.x; i < rows * dim; i += kL2Threads) tile[i] = t[(size_t)base * dim + i];
    __syncthreads();
    if (!wave_has_queries) continue;
    const float* tile_rows = tile;
    int r = 0;
    // two train rows at a time: two independent accumulation chains (each row's sum keeps its own index order, so the result is
    // bit-identical to the one-row loop and to the oracle)
    for (; r + 1 < rows; r += 2) {
      const float* tr0 = tile_rows + r * dim; const float* tr1 = tr0 + dim;
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int i = 0; i < DIM_MAX; i++) {
        if (kExact || i < dim) {
          const float d0 = qa[i] - tr0[i], d1 = qa[i] - tr1[i];
          acc0 = fma((double)d0, (double)d0, acc0); acc1 = fma((double)d1, (double)d1, acc1);
        }
      }
      LLD_L2_TAKE(sqrt(acc0), base + r); LLD_L2_TAKE(sqrt(acc1), base + r + 1);
    }
    for (; r < rows; r++) {
      const float* tr = tile_rows + r * dim;
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < DIM_MAX; i++) {
        if (kExact || i < dim) { const float d = qa[i] - tr[i]; acc = fma((double)d, (double)d, acc); }
      }
      LLD_L2_TAKE(sqrt(acc), base + r);
    }
  }
#undef LLD_L2_TAKE
  if (valid) {
    best_idx[out_off + qi] = bi; best_dist[out_off + qi] = bd; second_idx[out_off + qi] = si; second_dist[out_off + qi] = sd;
  }
}

// Latency form of the same search for small problems (one frame pair): one wavefront per QUERY row, lanes stride over the train
// rows, so a 300 x 300 x 72 problem is 300 short wavefronts instead of two workgroups walking 21 600 dependent FMAs per lane.
// Same arithmetic per pair and the same lexicographic (distance, index) order, hence identical outputs.
__global__ __launch_bounds__(64) void l2f32_best2_row_kernel(const float* __restrict__ q, const float* __restrict__ t, int nt, int dim,
                                                            const uint8_t* __restrict__ mask, int* __restrict__ best_idx, double* __restrict__ best_dist,
                                                            int* __restrict__ second_idx, double* __restrict__ second_dist, double* __restrict__ dist_matrix) {
  extern __shared__ __attribute__((aligned(16))) float qrow_l2[];
  const int qi = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < dim; i += 64) qrow_l2[i] = q[(size_t)qi * dim + i];
  __syncthreads();
  double bd = 1.7976931348623157e308, sd = 1.7976931348623157e308;
  int bi = 0x7fffffff, si = 0x7fffffff;
  for (int j = lane; j < nt; j += 64) {
    const float* tr = t + (size_t)j * dim;
    double acc = 0.0;
    for (int i = 0; i < dim; i++) { const float d = qrow_l2[i] - tr[i]; acc = fma((double)d, (double)d, acc); }
    const double dist = sqrt(acc);
    if (dist_matrix) dist_matrix[(size_t)qi * nt + j] = dist;
    if (mask && !mask[(size_t)qi * nt + j]) continue;
    if (dist < bd) { sd = bd; si = bi; bd = dist; bi = j; }            // ascending j per lane: strict '<' keeps the lower index
    else if (dist < sd) { sd = dist; si = j; }
  }
  auto less = [](double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); };
  for (int off = 32; off > 0; off >>= 1) {
    const double obd = __shfl_xor(bd, off), osd = __shfl_xor(sd, off);
    const int obi = __shfl_xor(bi, off), osi = __shfl_xor(si, off);
    // merge two sorted pairs (b, s) and (ob, os): the two smallest of the four in (distance, index) order
    if (less(obd, obi, bd, bi)) {
      if (less(bd, bi, osd, osi)) { sd = bd; si = bi; } else { sd = osd; si = osi; }
      bd = obd; bi = obi;
    } else if (less(obd, obi, sd, si)) { sd = obd; si = obi; }
  }
  if (lane == 0) {
    best_idx[qi] = bi == 0x7fffffff ? -1 : bi; best_dist[qi] = bd;
    second_idx[qi] = si == 0x7fffffff ? -1 : si; second_dist[qi] = sd;
  }
}

// ------------------------------------------------------------------ TwoFrameLineMatcher::MatchLines (src/TwoFrameLineMatcher.cc:26-77)
// The reference walks the left lines in order; each takes the untaken, gated right line with the smallest distance below tau
// (strict '<' while scanning in index order = lexicographic (distance, index) minimum).  Only `taken` is order dependent, so:
//   line_candidates_kernel  one wavefront per LEFT line: gates + float-L2 distances of its row, then the row's kLineTopK smallest
//                           (distance, index) candidates by repeated wavefront argmin - all left lines in parallel;
//   line_resolve_kernel     the order-dependent assignment by fixed-point rounds over all left lines at once (see there).
constexpr int kLineTopK = 8;
struct LineCand { double d[kLineTopK]; int idx[kLineTopK]; int n, pad; };
constexpr double kInfD = 1.7976931348623157e308;

// Geometric gates of TwoFrameLineMatcher::CheckLinePair (src/TwoFrameLineMatcher.cc:79-109) for the stereo pair of one frame.
// T = identity, T_right = [I | (b,0,0)] (GetTForRight, src/LineMatching.cc:228-237).
// The 3x3 system of vgl::TriangulateLine (src/vgl.cc:78-108) has rows n1, n2, d = n1 x n2 / |n1 x n2|, so its determinant is
// |n1 x n2| > 0 once the 0.975 parallelism test passed (rank 3 always) and X0 = (b1 (d x n1)) / det in closed form.  The 3x2
// least squares of vgl::ReprojectLinePointTo3D (src/vgl.cc:336-346) is solved the way the reference solves it, by a column-pivoted
// Householder QR (reproject_param_qr): its normal equations square the condition number, and when the viewing ray of an end point is
// parallel to the triangulated line to ~1e-8 (two unrelated segments: one pair in ~1e8) they return a line parameter of the wrong
// sign, and the depth test `p.z < 0` with it (found by tools/fuzz_matchers.py, FUZZ_BIG=1, seed 9, scene 13712).
struct LineGateParams { double K[9]; double b; double min_len; int is_stereo; };

// Eigen::ColPivHouseholderQR::solve of the 3x2 system [a | c] (depth, param)^T = r, restated as oracle/lldo_linematch.cpp restates it
// (pivot on the larger column norm, makeHouseholderInPlace, rank by |R_kk| > 2 eps max|R_kk|, dropped unknowns = 0); returns the
// line parameter.  No contraction: the same operations as the host code, one rounding each.
__device__ __forceinline__ double reproject_param_qr(const double a[3], const double c[3], const double r[3]) {
#pragma clang fp contract(off)
  double A[3][2] = {{a[0], c[0]}, {a[1], c[1]}, {a[2], c[2]}};
  double b[3] = {r[0], r[1], r[2]};
  int perm[2] = {0, 1};
  double diag[2] = {0.0, 0.0}, maxpivot = 0.0;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    int best = k; double best_norm = -1.0;
    for (int cc = k; cc < 2; cc++) {
      double sn = 0.0; for (int rr = k; rr < 3; rr++) sn += A[rr][cc] * A[rr][cc];
      if (sn > best_norm) { best_norm = sn; best = cc; }
    }
    if (best != k) { for (int rr = 0; rr < 3; rr++) { const double t = A[rr][k]; A[rr][k] = A[rr][best]; A[rr][best] = t; } const int t = perm[k]; perm[k] = perm[best]; perm[best] = t; }
    const double c0 = A[k][k];
    double tail = 0.0; for (int rr = k + 1; rr < 3; rr++) tail += A[rr][k] * A[rr][k];
    double beta = c0, tau = 0.0; double v[3] = {0.0, 0.0, 0.0};
    if (tail > 2.2250738585072014e-308) {
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= 0) beta = -beta;
      for (int rr = k + 1; rr < 3; rr++) v[rr] = A[rr][k] / (c0 - beta);
      tau = (beta - c0) / beta;
      for (int cc = k + 1; cc < 2; cc++) {
        double w = A[k][cc]; for (int rr = k + 1; rr < 3; rr++) w += v[rr] * A[rr][cc];
        A[k][cc] -= tau * w; for (int rr = k + 1; rr < 3; rr++) A[rr][cc] -= tau * w * v[rr];
      }
      double w = b[k]; for (int rr = k + 1; rr < 3; rr++) w += v[rr] * b[rr];
      b[k] -= tau * w; for (int rr = k + 1; rr < 3; rr++) b[rr] -= tau * w * v[rr];
    }
    A[k][k] = beta; for (int rr = k + 1; rr < 3; rr++) A[rr][k] = 0.0;
    diag[k] = fabs(beta);
    if (diag[k] > maxpivot) maxpivot = diag[k];
  }
  const double threshold = 2.220446049250313e-16 * 2.0;                         // NumTraits::epsilon() * diagonalSize()
  int rank = 0;
  for (int k = 0; k < 2; k++) if (diag[k] > threshold * maxpivot) rank++;
  double y[2] = {0.0, 0.0};
  for (int k = rank - 1; k >= 0; k--) {
    double sacc = b[k]; for (int cc = k + 1; cc < rank; cc++) sacc -= A[k][cc] * y[cc];
    y[k] = sacc / A[k][k];
  }
  double x[2] = {0.0, 0.0};
  for (int k = 0; k < 2; k++) x[perm[k]] = (k < rank) ? y[k] : 0.0;
  return x[1];
}

__device__ __forceinline__ void normalized_line_eq(const float* kl, const double* K, double* l) {
  const double sx = kl[0], sy = kl[1], ex = kl[2], ey = kl[3];
  const double ix = sy - ey, iy = ex - sx, iz = sx * ey - sy * ex;             // (sx,sy,1) x (ex,ey,1)
  const double a = K[0] * ix + K[3] * iy + K[6] * iz, b = K[1] * ix + K[4] * iy + K[7] * iz, c = K[2] * ix + K[5] * iy + K[8] * iz;
  const double n = sqrt(a * a + b * b);
  l[0] = a / n; l[1] = b / n; l[2] = c / n;
}

__device__ __forceinline__ bool line_pair_gate(const LineGateParams& P, const float* k1, int o1, const float* k2, int o2) {
  if (P.is_stereo && o1 != o2) return false;
  const double d1x = (double)k1[0] - (double)k1[2], d1y = (double)k1[1] - (double)k1[3];
  const double d2x = (double)k2[0] - (double)k2[2], d2y = (double)k2[1] - (double)k2[3];
  if (sqrt(d1x * d1x + d1y * d1y) < P.min_len || sqrt(d2x * d2x + d2y * d2y) < P.min_len) return false;
  double n1[3], n2[3];
  normalized_line_eq(k1, P.K, n1); normalized_line_eq(k2, P.K, n2);
  const double nn1 = sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]), nn2 = sqrt(n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]);
  if (fabs(n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2]) / nn1 / nn2 > 0.975) return false;
  double d[3] = {n1[1] * n2[2] - n1[2] * n2[1], n1[2] * n2[0] - n1[0] * n2[2], n1[0] * n2[1] - n1[1] * n2[0]};
  const double det = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  d[0] /= det; d[1] /= det; d[2] /= det;
  const double b1 = n2[0] * P.b;                                               // n2 . t2, t2 = (b,0,0); n1 . t1 = 0
  const double X0[3] = {b1 * (d[1] * n1[2] - d[2] * n1[1]) / det, b1 * (d[2] * n1[0] - d[0] * n1[2]) / det, b1 * (d[0] * n1[1] - d[1] * n1[0]) / det};
  if (sqrt(X0[0] * X0[0] + X0[1] * X0[1] + X0[2] * X0[2]) < 0.5) return false;
  const double c[3] = {-(P.K[0] * d[0] + P.K[1] * d[1] + P.K[2] * d[2]), -(P.K[3] * d[0] + P.K[4] * d[1] + P.K[5] * d[2]),
                       -(P.K[6] * d[0] + P.K[7] * d[1] + P.K[8] * d[2])};
  const double r[3] = {P.K[0] * X0[0] + P.K[1] * X0[1] + P.K[2] * X0[2], P.K[3] * X0[0] + P.K[4] * X0[1] + P.K[5] * X0[2],
                       P.K[6] * X0[0] + P.K[7] * X0[1] + P.K[8] * X0[2]};
  bool front = true;
  for (int e = 0; e < 2; e++) {
    const double a[3] = {(double)k1[2 * e], (double)k1[2 * e + 1], 1.0};
    const double p = reproject_param_qr(a, c, r);                               // line parameter of the re-projected endpoint
    if (X0[2] + p * d[2] < 0) front = false;
  }
  return front;
}

// grid nq, block 64; dynamic LDS: nt doubles (the row's admissible distances) + dim floats (the left descriptor).
// kGeom: the gate is CheckLinePair's geometry (written to gate_mat); otherwise gate_in is the caller's matrix (or NULL = all pass).
template <bool kGeom>
__global__ __launch_bounds__(64) void line_candidates_kernel(LineGateParams P, const float* __restrict__ left, const int* __restrict__ loct,
                                                            const float* __restrict__ right, const int* __restrict__ roct,
                                                            const float* __restrict__ q, const float* __restrict__ t, int dim, int nt,
                                                            const uint8_t* __restrict__ gate_in, double tau, uint8_t* __restrict__ gate_mat,
                                                            double* __restrict__ dmat, LineCand* __restrict__ cand) {
  extern __shared__ __attribute__((aligned(16))) double lds_row[];
  double* drow = lds_row;
  float* qrow = reinterpret_cast<float*>(drow + nt);
  const int j = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < dim; i += 64) qrow[i] = q[(size_t)j * dim + i];
  __syncthreads();
  float k1[4] = {0.f, 0.f, 0.f, 0.f}; int o1 = 0;
  if (kGeom) { for (int e = 0; e < 4; e++) k1[e] = left[4 * j + e]; o1 = loct[j]; }
  for (int oi = lane; oi < nt; oi += 64) {
    bool g;
    if (kGeom) { float k2[4]; for (int e = 0; e < 4; e++) k2[e] = right[4 * oi + e]; g = line_pair_gate(P, k1, o1, k2, roct[oi]); gate_mat[(size_t)j * nt + oi] = g ? 1 : 0; }
    else g = !gate_in || gate_in[(size_t)j * nt + oi] != 0;
    double dist = kInfD;
    if (g) {                                                           // LineMatcher::MatchLineDescriptors: float difference, double accumulation
      const float* tr = t + (size_t)oi * dim;
      double acc = 0.0;
      for (int i = 0; i < dim; i++) { const float df = qrow[i] - tr[i]; acc = fma((double)df, (double)df, acc); }
      dist = sqrt(acc);
    }
    dmat[(size_t)j * nt + oi] = dist;
    drow[oi] = (g && dist < tau) ? dist : kInfD;
  }
  __syncthreads();
  // the kLineTopK smallest (distance, index) pairs of the row, in order: every round takes the smallest pair greater than the last
  double pd = -1.0; int pi = -1, n = 0;
  double my_d = kInfD; int my_i = -1;
  for (int r = 0; r < kLineTopK; r++) {
    double bd = kInfD; int bi = 0x7fffffff;
    for (int oi = lane; oi < nt; oi += 64) {
      const double d = drow[oi];
      if (d < tau && (d > pd || (d == pd && oi > pi)) && d < bd) { bd = d; bi = oi; }    // ascending oi per lane: lowest index on ties
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double od = __shfl_xor(bd, off); const int oidx = __shfl_xor(bi, off);
      if (od < bd || (od == bd && oidx < bi)) { bd = od; bi = oidx; }
    }
    if (bi == 0x7fffffff) break;
    if (lane == r) { my_d = bd; my_i = bi; }
    pd = bd; pi = bi; n++;
  }
  if (lane < kLineTopK) { cand[j].d[lane] = my_d; cand[j].idx[lane] = my_i; }
  if (lane == 0) { cand[j].n = n; cand[j].pad = 0; }
}

// The greedy, order-dependent assignment ("left line j takes its best right line that no EARLIER left line took").  Two regimes:
//   * fixed-point rounds, as the guided ORB search resolves its occupancy (lld_orb_search.hip): a round lets every line pick, in parallel, the
//     first entry of its short list that no line with a smaller index picked in the round before; blk[c] = the smallest index that picks c.
//     Line 0 is final after one round, line j after j + 1 at the latest, and a round that changes nothing has reached the sequential answer.
//     AddLinesFrom's sparse candidate sets converge in 3 - 6 rounds of a few hundred cycles (the one-wavefront walk of rounds 2 - 5 paid
//     ~400 cycles per LINE: 45 us for 260 map lines, now 5);
//   * where the rounds do NOT converge quickly - the stereo matcher with its wide tau: unrelated lines take over partners of later lines and
//     the corrections ripple down one line per round (300 x 300: ~300 rounds, 340 us) - the kernel stops after kResolveRounds rounds and ONE
//     wavefront walks the rest in order: every line below the smallest index that changed in the last round is final (its inputs, the picks of
//     the lines before it, did not change, nor did its own), so the walk starts there with their picks as the taken set.
// A list that is full and completely taken falls back to a scan of the stored row like the reference does.  One workgroup; LDS: blk[nt] + pick[nq].
constexpr int kResolveThreads = 256;
constexpr int kResolveRounds = 8;
inline size_t line_resolve_lds(int nq, int nt) { return ((size_t)nq + (size_t)nt) * 4 + 16; }
static int line_resolve_prepare(int nq, int nt);      // raises the kernel's dynamic-LDS ceiling when blk + pick need more than the default (defined after the kernel)
__global__ __launch_bounds__(kResolveThreads) void line_resolve_kernel(const LineCand* __restrict__ cand, const double* __restrict__ dmat, const uint8_t* __restrict__ gate,
                                                                      int nq, int nt, double tau, int* __restrict__ matches, double* __restrict__ match_dist,
                                                                      lld_track::LineApplyDev ap, const double* __restrict__ map_x0, const double* __restrict__ map_dir) {
  extern __shared__ __attribute__((aligned(16))) int res_lds[];
  __shared__ int changed_min, scan_min;
  __shared__ int clist[kResolveThreads][kLineTopK + 1];  // the ordered walk's candidate lists, a chunk of lines at a time
  int* blk = res_lds; int* pick = res_lds + nt;
  const int tid = threadIdx.x;
  constexpr int kFree = 0x7fffffff;
  for (int i = tid; i < nt; i += kResolveThreads) blk[i] = kFree;
  for (int j = tid; j < nq; j += kResolveThreads) pick[j] = -1;
  if (tid == 0) { changed_min = kFree; scan_min = kFree; }
  __syncthreads();
  int walk_from = kFree;                                  // < kFree: lines from this index on are walked in order
  for (int round = 0;; round++) {
    for (int j = tid; j < nq; j += kResolveThreads) {
      const LineCand& C = cand[j];
      const int n = C.n;
      int bi = -1;
      for (int k = 0; k < n; k++) { const int c = C.idx[k]; if (blk[c] >= j) { bi = c; break; } }
      // every listed candidate taken and the list was cut: this line needs a scan of its stored row, which only the ordered walk does
      // (with the wavefront's 64 lanes and the FINAL taken set); the line and everything after it is left to the walk
      if (bi < 0 && n == kLineTopK) atomicMin(&scan_min, j);
      if (bi != pick[j]) { pick[j] = bi; atomicMin(&changed_min, j); }
    }
    __syncthreads();
    const int first_changed = changed_min, first_scan = scan_min;
    if (first_changed == kFree || first_changed >= first_scan) { walk_from = first_scan; break; }     // converged (below the first line that needs a row scan)
    if (round + 1 >= kResolveRounds) walk_from = min(first_changed, first_scan);
    __syncthreads();
    for (int i = tid; i < nt; i += kResolveThreads) blk[i] = kFree;
    if (tid == 0) { changed_min = kFree; scan_min = kFree; }
    __syncthreads();
    for (int j = tid; j < nq; j += kResolveThreads) { const int c = pick[j]; if (c >= 0) atomicMin(&blk[c], j); }
    __syncthreads();
    if (walk_from != kFree) break;
  }
  if (walk_from != kFree) {
    // the taken set of the walk: the (final) picks of the lines below walk_from
    __syncthreads();
    for (int i = tid; i < nt; i += kResolveThreads) blk[i] = kFree;
    __syncthreads();
    for (int j = tid; j < min(walk_from, nq); j += kResolveThreads) { const int c = pick[j]; if (c >= 0) blk[c] = j; }
    for (int j0 = walk_from; j0 < nq; j0 += kResolveThreads) {
      const int nj = min(kResolveThreads, nq - j0);
      __syncthreads();
      if (tid < nj) {
        const LineCand& C = cand[j0 + tid];
#pragma unroll
        for (int k = 0; k < kLineTopK; k++) clist[tid][k] = C.idx[k];
        clist[tid][kLineTopK] = C.n;
      }
      __syncthreads();
      if (tid < 64) {
        const int lane = tid;
        for (int jj = 0; jj < nj; jj++) {
          const int j = j0 + jj;
          const int n = clist[jj][kLineTopK];
          const int my_i = lane < n ? clist[jj][lane] : 0;
          const bool free_ = lane < n && blk[my_i] == kFree;
          const unsigned long long mask = __ballot(free_);
          int bi = -1;
          if (mask) bi = __builtin_amdgcn_readlane(my_i, __ffsll((long long)mask) - 1);
          else if (n == kLineTopK) {
            double sd = kInfD; int si = kFree;
            for (int oi = lane; oi < nt; oi += 64) {
              if (blk[oi] != kFree) continue;
              if (gate && !gate[(size_t)j * nt + oi]) continue;
              const double d = dmat[(size_t)j * nt + oi];
              if (d < tau && d < sd) { sd = d; si = oi; }
            }
            for (int off = 32; off > 0; off >>= 1) {
              const double od = __shfl_xor(sd, off); const int oidx = __shfl_xor(si, off);
              if (od < sd || (od == sd && oidx < si)) { sd = od; si = oidx; }
            }
            if (si != kFree) bi = si;
          }
          if (lane == 0) { pick[j] = bi; if (bi >= 0) blk[bi] = j; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (one wavefront: its LDS operations retire in order; this keeps the compiler from moving the next reads up)
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
  __syncthreads();
  for (int j = tid; j < nq; j += kResolveThreads) {
    const int bi = pick[j];
    matches[j] = bi;
    if (match_dist) match_dist[j] = bi >= 0 ? dmat[(size_t)j * nt + bi] : kInfD;
    // the device-resident chain (lld_frame_track_*): the frame line takes the map line here (every frame line has at most one taker)
    if (ap.ln_has && bi >= 0) {
      ap.ln_has[bi] = 1; ap.ln_id[bi] = ap.map_id[j];
      for (int c = 0; c < 3; c++) { ap.ln_x0[3 * bi + c] = map_x0[3 * j + c]; ap.ln_dir[3 * bi + c] = map_dir[3 * j + c]; }
      const int at = atomicAdd(ap.n_tracked, 1);
      if (at < ap.tracked_cap) ap.tracked[at] = ap.map_id[j];
    }
  }
}

static int line_resolve_prepare(int nq, int nt) {
  const size_t lds = line_resolve_lds(nq, nt);
  if (lds > 150 * 1024) return LLD_ERR_UNSUPPORTED;                          // (about 38 000 lines on both sides together)
  if (lds > 48 * 1024) LLD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&line_resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return LLD_OK;
}

// ------------------------------------------------------------------ Tracking::AddLinesFrom (src/Tracking.cc:996-1124)
// Gate matrix of the map-line -> frame-line association: row = map line, column = line of the current frame.  Everything that depends
// on the map line only is computed once per row (projected image lines in the left / right camera for vgl::LineReprojErrorL1, the Hough
// cell neighbourhood of SubselectWithGrid, the depth test of the main points); a column contributes its cell, its occupancy, its stereo
// partner and four dot products.  The greedy, order-dependent part is line_resolve_kernel's, as for TwoFrameLineMatcher.
constexpr int kHoughDist = 50, kHoughAng = 50;           // FRAME_DIST_CELLS, FRAME_ANG_CELLS (include/Frame.h:45-46)
#define LLD_HOUGH_PI 3.14159265                          /* the literal of src/LineMatching.cc:61 */
struct HoughCell { int dist_ind, ang_ind, shift_dist, shift_ang; };
// centre cell + shifts of GetHoughCoordinates (src/LineMatching.cc:63-110) for the homogeneous image line leq (pixels)
__host__ __device__ inline HoughCell hough_cell(double lx, double ly, double lz, double sx, double sy) {
  lx /= sx; ly /= sy;
  const double n = sqrt(lx * lx + ly * ly);
  lx /= n; ly /= n; lz /= n;
  if (ly < 0) { lx = -lx; ly = -ly; lz = -lz; }
  HoughCell c;
  // a degenerate line (coincident end points): the reference would index its grid with the integer cast of a NaN; the build treats it as the line y = 0
  if (!(lx - lx == 0.0) || !(ly - ly == 0.0) || !(lz - lz == 0.0)) { lx = 0.0; ly = 1.0; lz = 0.0; }      // x - x == 0 only for a finite x
  const double dist_level = fabs(lz / (sqrt(2.0))) * kHoughDist;
  int di = (int)floor(dist_level + 0.5);
  di = di < kHoughDist - 1 ? di : kHoughDist - 1; di = di > 0 ? di : 0;
  c.dist_ind = di; c.shift_dist = (dist_level - di < 0) ? 1 : -1;
  const double ang = atan2(ly, lx);
  const double ang_level = ang / LLD_HOUGH_PI * kHoughAng;
  int ai = (int)floor(ang_level + 0.5);
  ai = ai < kHoughAng - 1 ? ai : kHoughAng - 1; ai = ai > 0 ? ai : 0;
  c.ang_ind = ai; c.shift_ang = (ang_level - ai < 0) ? 1 : -1;
  return c;
}
struct LineTrackParams { double K[9]; double R[9]; double t[3]; double tr[3]; double thr_base, sx, sy; int monocular, use_grid; };

// one thread per frame line: its grid cell (the fill the reference lacks: the centre cell of the line through the left KeyLine)
__global__ void line_cells_kernel(const float* __restrict__ lines, int n, double sx, double sy, int* __restrict__ cell) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double xs = lines[4 * i], ys = lines[4 * i + 1], xe = lines[4 * i + 2], ye = lines[4 * i + 3];
  const HoughCell c = hough_cell(ys - ye, xe - xs, xs * ye - ys * xe, sx, sy);          // GetLineEq: (xs,ys,1) x (xe,ye,1)
  cell[i] = c.dist_ind * kHoughAng + c.ang_ind;
}

__device__ __forceinline__ void track_map_point(const LineTrackParams& P, const double* t, const double* X, double* o) {   // vgl::MapPoint: R^T (X - t)
  const double d0 = X[0] - t[0], d1 = X[1] - t[1], d2 = X[2] - t[2];
  for (int c = 0; c < 3; c++) o[c] = P.R[0 * 3 + c] * d0 + P.R[1 * 3 + c] * d1 + P.R[2 * 3 + c] * d2;
}
__device__ __forceinline__ void track_k_mul(const double* K, const double* X, double* o) {
  for (int r = 0; r < 3; r++) o[r] = K[3 * r] * X[0] + K[3 * r + 1] * X[1] + K[3 * r + 2] * X[2];
}
// image line of the 3D line (X0, dir) in the camera with centre t: K MapPoint(X0) x K MapPoint(X0 + dir), first two components normalised
__device__ __forceinline__ void track_image_line(const LineTrackParams& P, const double* t, const double* X0, const double* dir, double* l) {
  double a[3], b[3], Xa[3], Xb[3];
  const double X0d[3] = {X0[0] + dir[0], X0[1] + dir[1], X0[2] + dir[2]};
  track_map_point(P, t, X0, a); track_map_point(P, t, X0d, b);
  track_k_mul(P.K, a, Xa); track_k_mul(P.K, b, Xb);
  {
    // no fused multiply-add here: a map line with a zero direction has Xa == Xb and the reference's cross product is exactly zero (the line
    // becomes NaN and, `NaN > thr` being false, passes the reprojection gate); a contracted a*b - c*d would leave a rounding residue instead
#pragma clang fp contract(off)
    l[0] = Xa[1] * Xb[2] - Xa[2] * Xb[1]; l[1] = Xa[2] * Xb[0] - Xa[0] * Xb[2]; l[2] = Xa[0] * Xb[1] - Xa[1] * Xb[0];
  }
  const double n = sqrt(l[0] * l[0] + l[1] * l[1]);
  l[0] /= n; l[1] /= n; l[2] /= n;
}

// grid n_map, block 64
__global__ __launch_bounds__(64) void line_track_gate_kernel(LineTrackParams P, const double* __restrict__ x0, const double* __restrict__ dir,
                                                            const double* __restrict__ x1, const double* __restrict__ x2, const uint8_t* __restrict__ skip,
                                                            int n_cur, const float* __restrict__ left, const int* __restrict__ loct,
                                                            const float* __restrict__ right, const int* __restrict__ lmatch,
                                                            const uint8_t* __restrict__ occupied, const int* __restrict__ cell, uint8_t* __restrict__ gate,
                                                            const LineTrackParams* __restrict__ P_dev) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (P_dev) P = *P_dev;                                                      // the pose came out of a kernel (lld_frame_track_*): camera in device memory
  uint8_t* grow = gate + (size_t)i * n_cur;
  bool row_ok = !(skip && skip[i]);
  double X1c[3], X2c[3];
  track_map_point(P, P.t, x1 + 3 * i, X1c); track_map_point(P, P.t, x2 + 3 * i, X2c);
  if (X1c[2] < 0 || X2c[2] < 0) row_ok = false;                               // Tracking.cc:1066-1074
  double ll[3], lr[3];
  track_image_line(P, P.t, x0 + 3 * i, dir + 3 * i, ll);
  track_image_line(P, P.tr, x0 + 3 * i, dir + 3 * i, lr);
  // SubselectWithGrid: the cell neighbourhood of the projected line (GetHoughCoordinates with step 3: six consecutive angle cells
  // around the centre, modulo the grid, and up to six distance cells, the last distance row excluded as in the reference)
  const HoughCell hc = hough_cell(ll[0], ll[1], ll[2], P.sx, P.sy);
  const int ang_min = hc.ang_ind < hc.ang_ind + hc.shift_ang ? hc.ang_ind : hc.ang_ind + hc.shift_ang;
  const int dist_min = hc.dist_ind < hc.dist_ind + hc.shift_dist ? hc.dist_ind : hc.dist_ind + hc.shift_dist;
  for (int si = lane; si < n_cur; si += 64) {
    bool g = row_ok && !(occupied && occupied[si]);
    const int ri = lmatch[si];
    if (ri < 0 && !P.monocular) g = false;
    if (g && P.use_grid) {
      const int cd = cell[si] / kHoughAng, ca = cell[si] - cd * kHoughAng;
      int da = ca - (ang_min - 2); da %= kHoughAng; if (da < 0) da += kHoughAng;
      const bool in_ang = da < 6;
      const bool in_dist = cd >= dist_min - 2 && cd <= dist_min + 3 && cd >= 0 && cd < kHoughDist - 1;
      g = in_ang && in_dist;
    }
    if (g) {
      double thr = P.thr_base;                                                // GetReprojThrPyramid
      for (int o = 0; o < loct[si]; o++) thr *= 1.44;
      const float* kl = left + 4 * si;
      const double se = fabs((double)kl[0] * ll[0] + (double)kl[1] * ll[1] + ll[2]) + fabs((double)kl[2] * ll[0] + (double)kl[3] * ll[1] + ll[2]);
      double se2 = 0.0;
      if (!P.monocular) {
        const float* kr = right + 4 * ri;
        se2 = fabs((double)kr[0] * lr[0] + (double)kr[1] * lr[1] + lr[2]) + fabs((double)kr[2] * lr[0] + (double)kr[3] * lr[1] + lr[2]);
      }
      if (se > thr || se2 > thr) g = false;                                   // Tracking.cc:1085
    }
    grow[si] = g ? 1 : 0;
  }
}

// ------------------------------------------------------------------ Tracking::MatchLinesLastKF (src/Tracking.cc:1449-1611)
struct LastKfParams { double K[9]; double R[9], t[3], tr[3]; double Rl[9], tl[3], tlr[3]; double thr_base, md_thr, sx, sy; int use_grid; };
__device__ __forceinline__ void d3_cross(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
__device__ __forceinline__ double d3_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void d3_rmul(const double* R, const double* v, double* o) { for (int r = 0; r < 3; r++) o[r] = R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2]; }
__device__ __forceinline__ void d3_rtmul(const double* R, const double* v, double* o) { for (int c = 0; c < 3; c++) o[c] = R[c] * v[0] + R[3 + c] * v[1] + R[6 + c] * v[2]; }
// vgl::TriangulateLine (src/vgl.cc:78-108) for two cameras with one rotation R (camera-to-world) and centres t1, t2.  The 3x3 system with
// rows n1, n2, d = n1 x n2 / |n1 x n2| has determinant |n1 x n2| > 0 once the parallelism test passed, so Eigen's rank() < 3 cannot fire
// and the solution is Cramer's rule with b = (n1.t1, n2.t2, 0).
__device__ __forceinline__ bool tri_line(const double* R, const double* t1, const double* t2, const double* l1, const double* l2, double* X0, double* dir) {
  double n1[3], n2[3];
  d3_rmul(R, l1, n1); d3_rmul(R, l2, n2);
  if (fabs(d3_dot(n1, n2)) / sqrt(d3_dot(n1, n1)) / sqrt(d3_dot(n2, n2)) > 0.975) return false;
  d3_cross(n1, n2, dir);
  const double dn = sqrt(d3_dot(dir, dir));
  dir[0] /= dn; dir[1] /= dn; dir[2] /= dn;
  double c23[3], c31[3];
  d3_cross(n2, dir, c23); d3_cross(dir, n1, c31);
  const double det = d3_dot(n1, c23), b1 = d3_dot(n1, t1), b2 = d3_dot(n2, t2);
  for (int k = 0; k < 3; k++) X0[k] = (b1 * c23[k] + b2 * c31[k]) / det;
  return true;
}
// symmetric 3x3 eigen-decomposition (cyclic Jacobi): A -> eigenvalues w[3], eigenvectors as the columns of V
__device__ __forceinline__ void sym3_eig(double A[3][3], double V[3][3], double* w) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; sweep++) {
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (A[p][q] == 0.0) continue;
      const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
      const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
      for (int k = 0; k < 3; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
      for (int k = 0; k < 3; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
      for (int k = 0; k < 3; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
    }
  }
  for (int i = 0; i < 3; i++) w[i] = A[i][i];
}
// vgl::MultiTriangulateLine (src/vgl.cc:28-76) for four views.  The direction is the eigenvector of the smallest eigenvalue of M^T M (the
// right singular vector of the smallest singular value of the normal matrix M), its largest component positive.  The reference solves
// M X0 = b by a pivoted QR and then removes X0's component along that direction: with M^T M = sum lambda_k v_k v_k^T this is
// X0 = sum over the two LARGER eigenpairs of v_k (v_k . M^T b) / lambda_k - the ill-conditioned third term never formed.
__device__ __forceinline__ bool multi_tri_line(const double* const* Rs, const double* const* ts, const double (*leqs)[3], double* X0, double* dir) {
  double nrm[4][3];
  for (int i = 0; i < 4; i++) {
    const double ln = sqrt(d3_dot(leqs[i], leqs[i]));
    const double u[3] = {leqs[i][0] / ln, leqs[i][1] / ln, leqs[i][2] / ln};
    d3_rmul(Rs[i], u, nrm[i]);
  }
  for (int i = 1; i < 4; i++) if (fabs(d3_dot(nrm[0], nrm[i])) / sqrt(d3_dot(nrm[0], nrm[0])) / sqrt(d3_dot(nrm[i], nrm[i])) > 0.975) return false;
  double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Mtb[3] = {0, 0, 0};
  for (int i = 0; i < 4; i++) {
    const double bi = d3_dot(nrm[i], ts[i]);
    for (int r = 0; r < 3; r++) { Mtb[r] += nrm[i][r] * bi; for (int c = 0; c < 3; c++) A[r][c] += nrm[i][r] * nrm[i][c]; }
  }
  double V[3][3], w[3];
  sym3_eig(A, V, w);
  int m = 0; for (int i = 1; i < 3; i++) if (w[i] < w[m]) m = i;
  double d[3] = {V[0][m], V[1][m], V[2][m]};
  const double dn = sqrt(d3_dot(d, d));
  int big = 0; if (fabs(d[1]) > fabs(d[big])) big = 1; if (fabs(d[2]) > fabs(d[big])) big = 2;
  const double sg = d[big] < 0 ? -1.0 / dn : 1.0 / dn;
  for (int k = 0; k < 3; k++) { dir[k] = d[k] * sg; X0[k] = 0.0; }
  for (int e = 0; e < 3; e++) {
    if (e == m) continue;
    const double v[3] = {V[0][e], V[1][e], V[2][e]};
    const double cf = d3_dot(v, Mtb) / w[e];
    for (int k = 0; k < 3; k++) X0[k] += cf * v[k];
  }
  return true;
}
// the line parameter of vgl::ReprojectLinePointTo3D (src/vgl.cc:336-346) through the normal equations of its 3x2 least squares
__device__ __forceinline__ double reproject_param(const double* K, const double* X0rot, const double* dirRot, double px, double py) {
  double c[3], r[3];
  for (int q = 0; q < 3; q++) { c[q] = -(K[3 * q] * dirRot[0] + K[3 * q + 1] * dirRot[1] + K[3 * q + 2] * dirRot[2]); r[q] = K[3 * q] * X0rot[0] + K[3 * q + 1] * X0rot[1] + K[3 * q + 2] * X0rot[2]; }
  const double aa = px * px + py * py + 1.0, ac = px * c[0] + py * c[1] + c[2], ar = px * r[0] + py * r[1] + r[2];
  const double cc = d3_dot(c, c), cr = d3_dot(c, r);
  return (aa * cr - ac * ar) / (aa * cc - ac * ac);
}
__device__ __forceinline__ void image_line_of(const double* K, const double* R, const double* t, const double* X0, const double* dir, double* l) {
  double a[3], b[3], Xa[3], Xb[3];
  const double d0[3] = {X0[0] - t[0], X0[1] - t[1], X0[2] - t[2]}, d1[3] = {X0[0] + dir[0] - t[0], X0[1] + dir[1] - t[1], X0[2] + dir[2] - t[2]};
  d3_rtmul(R, d0, a); d3_rtmul(R, d1, b);
  for (int r = 0; r < 3; r++) { Xa[r] = K[3 * r] * a[0] + K[3 * r + 1] * a[1] + K[3 * r + 2] * a[2]; Xb[r] = K[3 * r] * b[0] + K[3 * r + 1] * b[1] + K[3 * r + 2] * b[2]; }
  d3_cross(Xa, Xb, l);
  const double n = sqrt(l[0] * l[0] + l[1] * l[1]);
  l[0] /= n; l[1] /= n; l[2] /= n;
}

// grid n_cur, block 64; dynamic LDS: dim floats (the current line's descriptor)
__global__ __launch_bounds__(64) void line_lastkf_kernel(LastKfParams P, int n_cur, const float* __restrict__ cur_left, const float* __restrict__ cur_right,
                                                        const int* __restrict__ cur_lm, const uint8_t* __restrict__ cur_occ, const float* __restrict__ cur_desc,
                                                        int n_last, const float* __restrict__ last_left, const int* __restrict__ last_oct,
                                                        const float* __restrict__ last_right, const int* __restrict__ last_lm,
                                                        const uint8_t* __restrict__ last_skip, const float* __restrict__ last_desc, const int* __restrict__ last_cell,
                                                        int dim, int* __restrict__ match_last, uint8_t* __restrict__ created, double* __restrict__ x0_out,
                                                        double* __restrict__ dir_out) {
  extern __shared__ __attribute__((aligned(16))) float qrow[];
  const int i = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) { match_last[i] = -1; created[i] = 0; for (int k = 0; k < 3; k++) { x0_out[3 * i + k] = 0.0; dir_out[3 * i + k] = 0.0; } }
  if (cur_occ && cur_occ[i]) return;
  const int ri = cur_lm[i];
  if (ri < 0) return;
  for (int k = lane; k < dim; k += 64) qrow[k] = cur_desc[(size_t)i * dim + k];
  __syncthreads();
  double l1[3], l2[3], X0[3], ld[3];
  normalized_line_eq(cur_left + 4 * i, P.K, l1); normalized_line_eq(cur_right + 4 * ri, P.K, l2);
  if (!tri_line(P.R, P.t, P.tr, l1, l2, X0, ld)) return;
  double ll[3], lr[3];
  image_line_of(P.K, P.Rl, P.tl, X0, ld, ll);
  image_line_of(P.K, P.Rl, P.tlr, X0, ld, lr);
  const HoughCell hc = hough_cell(ll[0], ll[1], ll[2], P.sx, P.sy);
  const int ang_min = hc.ang_ind < hc.ang_ind + hc.shift_ang ? hc.ang_ind : hc.ang_ind + hc.shift_ang;
  const int dist_min = hc.dist_ind < hc.dist_ind + hc.shift_dist ? hc.dist_ind : hc.dist_ind + hc.shift_dist;
  double bd = kInfD; int bi = 0x7fffffff;
  for (int li = lane; li < n_last; li += 64) {
    const int pri = last_lm[li];
    if (pri < 0) continue;
    if (last_skip && last_skip[li]) continue;
    if (P.use_grid) {
      const int cd = last_cell[li] / kHoughAng, ca = last_cell[li] - cd * kHoughAng;
      int da = ca - (ang_min - 2); da %= kHoughAng; if (da < 0) da += kHoughAng;
      if (!(da < 6 && cd >= dist_min - 2 && cd <= dist_min + 3 && cd >= 0 && cd < kHoughDist - 1)) continue;
    }
    double thr = P.thr_base;
    for (int o = 0; o < last_oct[li]; o++) thr *= 1.44;
    const float* kl = last_left + 4 * li; const float* kr = last_right + 4 * pri;
    const double se = fabs((double)kl[0] * ll[0] + (double)kl[1] * ll[1] + ll[2]) + fabs((double)kl[2] * ll[0] + (double)kl[3] * ll[1] + ll[2]);
    const double se2 = fabs((double)kr[0] * lr[0] + (double)kr[1] * lr[1] + lr[2]) + fabs((double)kr[2] * lr[0] + (double)kr[3] * lr[1] + lr[2]);
    if (se > thr && se2 > thr) continue;                                    // Tracking.cc:1526
    const float* tr = last_desc + (size_t)li * dim;
    double acc = 0.0;
    for (int k = 0; k < dim; k++) { const float df = tr[k] - qrow[k]; acc = fma((double)df, (double)df, acc); }
    const double dist = sqrt(acc);
    if (dist < bd) { bd = dist; bi = li; }                                  // ascending li per lane: the lowest index wins ties
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double od = __shfl_xor(bd, off); const int oidx = __shfl_xor(bi, off);
    if (od < bd || (od == bd && oidx < bi)) { bd = od; bi = oidx; }
  }
  if (lane != 0 || bi == 0x7fffffff || bd > P.md_thr) return;
  match_last[i] = bi;
  const int pri = last_lm[bi];
  double leqs[4][3];
  for (int k = 0; k < 3; k++) { leqs[0][k] = l1[k]; leqs[1][k] = l2[k]; }
  normalized_line_eq(last_left + 4 * bi, P.K, leqs[2]); normalized_line_eq(last_right + 4 * pri, P.K, leqs[3]);
  const double* Rs[4] = {P.R, P.R, P.Rl, P.Rl}; const double* ts[4] = {P.t, P.tr, P.tl, P.tlr};
  if (!multi_tri_line(Rs, ts, leqs, X0, ld)) return;
  // ReprojectKeyLineTo3D of the current left KeyLine in the current pose
  double X0rot[3], dirRot[3];
  { const double dd[3] = {X0[0] - P.t[0], X0[1] - P.t[1], X0[2] - P.t[2]}; d3_rtmul(P.R, dd, X0rot); d3_rtmul(P.R, ld, dirRot); }
  const float* kc = cur_left + 4 * i;
  const double p_s = reproject_param(P.K, X0rot, dirRot, kc[0], kc[1]), p_e = reproject_param(P.K, X0rot, dirRot, kc[2], kc[3]);
  const double p1[3] = {X0[0] + p_s * ld[0], X0[1] + p_s * ld[1], X0[2] + p_s * ld[2]}, p2[3] = {X0[0] + p_e * ld[0], X0[1] + p_e * ld[1], X0[2] + p_e * ld[2]};
  bool behind = false;
  for (int v = 0; v < 4; v++) {
    const double a1[3] = {p1[0] - ts[v][0], p1[1] - ts[v][1], p1[2] - ts[v][2]}, a2[3] = {p2[0] - ts[v][0], p2[1] - ts[v][1], p2[2] - ts[v][2]};
    const double z1 = Rs[v][2] * a1[0] + Rs[v][5] * a1[1] + Rs[v][8] * a1[2], z2 = Rs[v][2] * a2[0] + Rs[v][5] * a2[1] + Rs[v][8] * a2[2];
    if (z1 < 0 || z2 < 0) behind = true;
  }
  if (behind) return;
  created[i] = 1;
  for (int k = 0; k < 3; k++) { x0_out[3 * i + k] = X0[k]; dir_out[3 * i + k] = ld[k]; }
}

int launch_hamming(lld_ctx* ctx, int batch, const uint32_t* q, int nq, const uint32_t* t, int nt, const uint8_t* mask,
                   int* bi, int* bd, int* si, int* sd) {
  if (nt >= (1 << kIdxBits)) return LLD_ERR_UNSUPPORTED;
  dim3 grid((nq + kQPerBlock - 1) / kQPerBlock, batch);
  hipLaunchKernelGGL(hamming256_best2_kernel, grid, dim3(kMatchThreads), 0, ctx->stream, reinterpret_cast<const uint4*>(q), nq,
                     reinterpret_cast<const uint4*>(t), nt, mask, bi, bd, si, sd);
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}

int launch_l2(lld_ctx* ctx, int batch, const float* q, int nq, const float* t, int nt, int dim, const uint8_t* mask,
              int* bi, double* bd, int* si, double* sd, double* dist_matrix) {
  if (batch == 1 && nq <= 8192 && dim <= 128) {                       // one frame pair: the wavefront-per-row form
    hipLaunchKernelGGL(l2f32_best2_row_kernel, dim3(nq), dim3(64), (size_t)dim * sizeof(float), ctx->stream, q, t, nt, dim, mask, bi, bd, si, sd, dist_matrix);
    LLD_HIP_TRY(hipGetLastError());
    return LLD_OK;
  }
  dim3 grid((nq + kL2Threads - 1) / kL2Threads, batch);
  const size_t lds = (size_t)kL2TileRows * dim * sizeof(float);
#define LLD_L2_LAUNCH(N, EXACT)                                                                                                 \
  hipLaunchKernelGGL((l2f32_best2_kernel<N, EXACT>), grid, dim3(kL2Threads), lds, ctx->stream, q, nq, t, nt, dim, mask, bi, bd, si, sd, \
                     dist_matrix)
  if (dim == 32) LLD_L2_LAUNCH(32, true);
  else if (dim == 72) LLD_L2_LAUNCH(72, true);
  else if (dim <= 32) LLD_L2_LAUNCH(32, false);
  else if (dim <= 72) LLD_L2_LAUNCH(72, false);
  else if (dim <= 128) LLD_L2_LAUNCH(128, false);
  else return LLD_ERR_UNSUPPORTED;
#undef LLD_L2_LAUNCH
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}

}  // namespace

extern "C" {

int lld_match_hamming256(lld_ctx* ctx, const uint32_t* q, int nq, const uint32_t* t, int nt, const uint8_t* mask,
                         int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist) {
  if (!ctx || !q || !t || nq < 0 || nt < 0 || !best_idx || !best_dist || !second_idx || !second_dist) return LLD_ERR_INVALID;
  if (nq == 0) return LLD_OK;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  const size_t qb = lld_slab::pad((size_t)nq * 32), tb = lld_slab::pad((size_t)nt * 32 + 32), mb = mask ? lld_slab::pad((size_t)nq * nt) : 0;
  const size_t ob = lld_slab::pad((size_t)nq * 4);
  void* base; int st = lld_ctx_scratch(ctx, qb + tb + mb + 4 * ob, &base); if (st) return st;
  lld_slab s; s.base = (char*)base;
  uint32_t* dq = s.take<uint32_t>((size_t)nq * 8); uint32_t* dt = s.take<uint32_t>((size_t)nt * 8 + 8);
  uint8_t* dm = mask ? s.take<uint8_t>((size_t)nq * nt) : nullptr;
  int *dbi = s.take<int>(nq), *dbd = s.take<int>(nq), *dsi = s.take<int>(nq), *dsd = s.take<int>(nq);
  LLD_HIP_TRY(hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
  if (nt) LLD_HIP_TRY(hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
  if (mask) LLD_HIP_TRY(hipMemcpyAsync(dm, mask, (size_t)nq * nt, hipMemcpyHostToDevice, ctx->stream));
  st = launch_hamming(ctx, 1, dq, nq, dt, nt, dm, dbi, dbd, dsi, dsd); if (st) return st;
  LLD_HIP_TRY(hipMemcpyAsync(best_idx, dbi, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(best_dist, dbd, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(second_idx, dsi, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(second_dist, dsd, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return LLD_OK;
}

int lld_match_hamming256_csr(lld_ctx* ctx, const uint32_t* q, int nq, const uint32_t* t, int nt, const int32_t* cand_start,
                             const int32_t* cand_idx, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist) {
  if (!ctx || !q || !t || !cand_start || !cand_idx || nq < 0 || nt < 0) return LLD_ERR_INVALID;
  if (nq == 0) return LLD_OK;
  const int ncand = cand_start[nq];
  for (int i = 0; i < nq; i++) if (cand_start[i + 1] < cand_start[i]) return LLD_ERR_INVALID;
  for (int k = 0; k < ncand; k++) if (cand_idx[k] < 0 || cand_idx[k] >= nt) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  const size_t need = lld_slab::pad((size_t)nq * 32) + lld_slab::pad((size_t)nt * 32 + 32) + lld_slab::pad((size_t)(nq + 1) * 4) +
                      lld_slab::pad((size_t)ncand * 4 + 4) + 4 * lld_slab::pad((size_t)nq * 4);
  void* base; int st = lld_ctx_scratch(ctx, need, &base); if (st) return st;
  lld_slab s; s.base = (char*)base;
  uint32_t* dq = s.take<uint32_t>((size_t)nq * 8); uint32_t* dt = s.take<uint32_t>((size_t)nt * 8 + 8);
  int* dcs = s.take<int>(nq + 1); int* dci = s.take<int>(ncand + 1);
  int *dbi = s.take<int>(nq), *dbd = s.take<int>(nq), *dsi = s.take<int>(nq), *dsd = s.take<int>(nq);
  LLD_HIP_TRY(hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
  if (nt) LLD_HIP_TRY(hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(dcs, cand_start, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (ncand) LLD_HIP_TRY(hipMemcpyAsync(dci, cand_idx, (size_t)ncand * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(hamming256_csr_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4*>(dq), nq,
                     reinterpret_cast<const uint4*>(dt), dcs, dci, dbi, dbd, dsi, dsd);
  LLD_HIP_TRY(hipGetLastError());
  LLD_HIP_TRY(hipMemcpyAsync(best_idx, dbi, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(best_dist, dbd, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(second_idx, dsi, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(second_dist, dsd, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return LLD_OK;
}

int lld_match_hamming256_batch_dev(lld_ctx* ctx, int batch, const uint32_t* q_dev, int nq, const uint32_t* t_dev, int nt,
                                   int32_t* best_idx_dev, int32_t* best_dist_dev, int32_t* second_idx_dev, int32_t* second_dist_dev) {
  if (!ctx || batch <= 0 || nq <= 0 || nt < 0 || !q_dev || !t_dev) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  return launch_hamming(ctx, batch, q_dev, nq, t_dev, nt, nullptr, best_idx_dev, best_dist_dev, second_idx_dev, second_dist_dev);
}

int lld_match_l2f32(lld_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim, const uint8_t* mask,
                    int32_t* best_idx, double* best_dist, int32_t* second_idx, double* second_dist) {
  if (!ctx || !q || !t || nq < 0 || nt < 0 || dim <= 0) return LLD_ERR_INVALID;
  if (dim > 128) return LLD_ERR_UNSUPPORTED;
  if (nq == 0) return LLD_OK;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  const size_t need = lld_slab::pad((size_t)nq * dim * 4) + lld_slab::pad((size_t)nt * dim * 4 + 16) + (mask ? lld_slab::pad((size_t)nq * nt) : 0) +
                      2 * lld_slab::pad((size_t)nq * 4) + 2 * lld_slab::pad((size_t)nq * 8);
  void* base; int st = lld_ctx_scratch(ctx, need, &base); if (st) return st;
  lld_slab s; s.base = (char*)base;
  float* dq = s.take<float>((size_t)nq * dim); float* dt = s.take<float>((size_t)nt * dim + 4);
  uint8_t* dm = mask ? s.take<uint8_t>((size_t)nq * nt) : nullptr;
  int *dbi = s.take<int>(nq), *dsi = s.take<int>(nq); double *dbd = s.take<double>(nq), *dsd = s.take<double>(nq);
  LLD_HIP_TRY(hipMemcpyAsync(dq, q, (size_t)nq * dim * 4, hipMemcpyHostToDevice, ctx->stream));
  if (nt) LLD_HIP_TRY(hipMemcpyAsync(dt, t, (size_t)nt * dim * 4, hipMemcpyHostToDevice, ctx->stream));
  if (mask) LLD_HIP_TRY(hipMemcpyAsync(dm, mask, (size_t)nq * nt, hipMemcpyHostToDevice, ctx->stream));
  st = launch_l2(ctx, 1, dq, nq, dt, nt, dim, dm, dbi, dbd, dsi, dsd, nullptr); if (st) return st;
  LLD_HIP_TRY(hipMemcpyAsync(best_idx, dbi, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(best_dist, dbd, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(second_idx, dsi, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipMemcpyAsync(second_dist, dsd, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return LLD_OK;
}

int lld_match_l2f32_batch_dev(lld_ctx* ctx, int batch, const float* q_dev, int nq, const float* t_dev, int nt, int dim,
                              int32_t* best_idx_dev, double* best_dist_dev, int32_t* second_idx_dev, double* second_dist_dev) {
  if (!ctx || batch <= 0 || nq <= 0 || nt < 0 || dim <= 0) return LLD_ERR_INVALID;
  if (dim > 128) return LLD_ERR_UNSUPPORTED;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  return launch_l2(ctx, batch, q_dev, nq, t_dev, nt, dim, nullptr, best_idx_dev, best_dist_dev, second_idx_dev, second_dist_dev, nullptr);
}

// Shared body of the two line matchers: one pinned-staged H2D copy of every input, candidate lists for all left lines in parallel,
// the in-order resolve on one wavefront, one D2H copy.
static int line_match_core(lld_ctx* ctx, const lld_line_stereo_params* geom, const float* left_lines, const int32_t* left_octave,
                           const float* dl, int nq, const float* right_lines, const int32_t* right_octave, const float* dr, int nt, int dim,
                           const uint8_t* gate_in, double tau, int32_t* matches, double* match_dist, uint8_t* gate_out) {
  if (dim > 128) return LLD_ERR_UNSUPPORTED;
  if (nq == 0) return LLD_OK;
  if (nt == 0) { for (int i = 0; i < nq; i++) { matches[i] = -1; if (match_dist) match_dist[i] = 1.7976931348623157e308; } return LLD_OK; }
  if ((size_t)nt * 8 + (size_t)dim * 4 > 150 * 1024) return LLD_ERR_UNSUPPORTED;         // one row of distances lives in LDS (nt <= ~19 000)
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  const size_t pairs = (size_t)nq * nt;
  auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
  // input region
  size_t in = 0;
  const size_t o_q = in; in += pad((size_t)nq * dim * 4);
  const size_t o_t = in; in += pad((size_t)nt * dim * 4);
  size_t o_ll = 0, o_rl = 0, o_lo = 0, o_ro = 0, o_g = 0;
  if (geom) { o_ll = in; in += pad((size_t)nq * 16); o_rl = in; in += pad((size_t)nt * 16); o_lo = in; in += pad((size_t)nq * 4); o_ro = in; in += pad((size_t)nt * 4); }
  else if (gate_in) { o_g = in; in += pad(pairs); }
  // output region (copied back), then device-only scratch
  size_t out = 0;
  const size_t r_m = out; out += pad((size_t)nq * 4);
  const size_t r_d = out; out += pad((size_t)nq * 8);
  const size_t r_g = out; if (geom && gate_out) out += pad(pairs);
  size_t dev = 0;
  const size_t s_g = dev; if (geom && !gate_out) dev += pad(pairs);
  const size_t s_mat = dev; dev += pad(pairs * 8);
  const size_t s_c = dev; dev += pad((size_t)nq * sizeof(LineCand));
  void* hb; int st = lld_ctx_pinned(ctx, in + out, &hb); if (st) return st;
  void* db; st = lld_ctx_scratch(ctx, in + out + dev + 256, &db); if (st) return st;
  char* h = (char*)hb; char* d = (char*)db; char* h_out = h + in; char* d_out = d + in; char* d_dev = d_out + out;
  std::memcpy(h + o_q, dl, (size_t)nq * dim * 4); std::memcpy(h + o_t, dr, (size_t)nt * dim * 4);
  if (geom) {
    std::memcpy(h + o_ll, left_lines, (size_t)nq * 16); std::memcpy(h + o_rl, right_lines, (size_t)nt * 16);
    std::memcpy(h + o_lo, left_octave, (size_t)nq * 4); std::memcpy(h + o_ro, right_octave, (size_t)nt * 4);
  } else if (gate_in) std::memcpy(h + o_g, gate_in, pairs);
  hipStream_t sm = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, in, hipMemcpyHostToDevice, sm));
  LineGateParams P; std::memset(&P, 0, sizeof P);
  uint8_t* dgate = nullptr;
  const uint8_t* dgate_in = nullptr;
  if (geom) {
    for (int i = 0; i < 9; i++) P.K[i] = geom->K[i];
    P.b = geom->b; P.min_len = (double)geom->min_line_length; P.is_stereo = geom->is_stereo;
    dgate = reinterpret_cast<uint8_t*>(gate_out ? d_out + r_g : d_dev + s_g);
  } else if (gate_in) dgate_in = reinterpret_cast<const uint8_t*>(d + o_g);
  double* dmat = reinterpret_cast<double*>(d_dev + s_mat);
  LineCand* dc = reinterpret_cast<LineCand*>(d_dev + s_c);
  const size_t lds = (size_t)nt * 8 + (size_t)dim * 4 + 16;
  if (lds > 48 * 1024) {
    LLD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&line_candidates_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    LLD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&line_candidates_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  
  if (geom)
    hipLaunchKernelGGL(line_candidates_kernel<true>, dim3(nq), dim3(64), lds, sm, P, reinterpret_cast<const float*>(d + o_ll), reinterpret_cast<const int*>(d + o_lo),
                       reinterpret_cast<const float*>(d + o_rl), reinterpret_cast<const int*>(d + o_ro), reinterpret_cast<const float*>(d + o_q),
                       reinterpret_cast<const float*>(d + o_t), dim, nt, nullptr, tau, dgate, dmat, dc);
  else
    hipLaunchKernelGGL(line_candidates_kernel<false>, dim3(nq), dim3(64), lds, sm, P, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const float*>(d + o_q),
                       reinterpret_cast<const float*>(d + o_t), dim, nt, dgate_in, tau, nullptr, dmat, dc);
  { const int rs = line_resolve_prepare(nq, nt); if (rs) return rs; }
  hipLaunchKernelGGL(line_resolve_kernel, dim3(1), dim3(kResolveThreads), line_resolve_lds(nq, nt), sm, dc, dmat, geom ? dgate : dgate_in, nq, nt, tau,
                     reinterpret_cast<int*>(d_out + r_m), reinterpret_cast<double*>(d_out + r_d), lld_track::LineApplyDev{}, nullptr, nullptr);
  LLD_HIP_TRY(hipGetLastError());
  LLD_HIP_TRY(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, sm));
  LLD_HIP_TRY(hipStreamSynchronize(sm));
  std::memcpy(matches, h_out + r_m, (size_t)nq * 4);
  if (match_dist) std::memcpy(match_dist, h_out + r_d, (size_t)nq * 8);
  if (geom && gate_out) std::memcpy(gate_out, h_out + r_g, pairs);
  return LLD_OK;
}

int lld_line_match_greedy(lld_ctx* ctx, const float* dl, int nq, const float* dr, int nt, int dim, const uint8_t* gate, double tau,
                          int32_t* matches, double* match_dist) {
  if (!ctx || !dl || !dr || nq < 0 || nt < 0 || dim <= 0 || !matches) return LLD_ERR_INVALID;
  return line_match_core(ctx, nullptr, nullptr, nullptr, dl, nq, nullptr, nullptr, dr, nt, dim, gate, tau, matches, match_dist, nullptr);
}

int lld_line_match_stereo(lld_ctx* ctx, const lld_line_stereo_params* params, const float* left_lines, const int32_t* left_octave,
                          const float* dl, int nq, const float* right_lines, const int32_t* right_octave, const float* dr, int nt, int dim,
                          int32_t* matches, double* match_dist, uint8_t* gate_out) {
  if (!ctx || !params || nq < 0 || nt < 0 || dim <= 0 || !matches) return LLD_ERR_INVALID;
  if (nq > 0 && (!left_lines || !left_octave || !dl)) return LLD_ERR_INVALID;
  if (nt > 0 && (!right_lines || !right_octave || !dr)) return LLD_ERR_INVALID;
  return line_match_core(ctx, params, left_lines, left_octave, dl, nq, right_lines, right_octave, dr, nt, dim, nullptr, params->tau, matches, match_dist, gate_out);
}

int lld_line_hough_cells(const float* lines, int n, double sx, double sy, int32_t* cell) {
  if (n < 0 || (n > 0 && (!lines || !cell)) || !(sx > 0) || !(sy > 0)) return LLD_ERR_INVALID;
  for (int i = 0; i < n; i++) {
    const double xs = lines[4 * i], ys = lines[4 * i + 1], xe = lines[4 * i + 2], ye = lines[4 * i + 3];
    const HoughCell c = hough_cell(ys - ye, xe - xs, xs * ye - ys * xe, sx, sy);
    cell[i] = c.dist_ind * kHoughAng + c.ang_ind;
  }
  return LLD_OK;
}

int lld_line_track_match(lld_ctx* ctx, const lld_line_track_params* prm, int n_map, const double* map_x0, const double* map_dir, const double* map_x1,
                         const double* map_x2, const uint8_t* map_skip, const float* map_desc, int n_cur, const float* left_lines,
                         const int32_t* left_octave, int n_right, const float* right_lines, const int32_t* line_matches, const uint8_t* occupied,
                         const float* cur_desc, int dim, int32_t* matches, double* match_dist, uint8_t* gate_out) {
  if (!ctx || !prm || n_map < 0 || n_cur < 0 || n_right < 0 || dim <= 0 || !matches) return LLD_ERR_INVALID;
  if (n_map > 0 && (!map_x0 || !map_dir || !map_x1 || !map_x2 || !map_desc)) return LLD_ERR_INVALID;
  if (n_cur > 0 && (!left_lines || !left_octave || !line_matches || !cur_desc)) return LLD_ERR_INVALID;
  if (!prm->monocular && n_right > 0 && !right_lines) return LLD_ERR_INVALID;
  if (!(prm->sx > 0) || !(prm->sy > 0)) return LLD_ERR_INVALID;
  for (int si = 0; si < n_cur; si++) {
    if (left_octave[si] < 0 || left_octave[si] > 64) return LLD_ERR_INVALID;
    if (line_matches[si] >= n_right) return LLD_ERR_INVALID;
  }
  if (dim > 128) return LLD_ERR_UNSUPPORTED;
  if (n_map == 0) return LLD_OK;
  if (n_cur == 0) { for (int i = 0; i < n_map; i++) { matches[i] = -1; if (match_dist) match_dist[i] = 1.7976931348623157e308; } return LLD_OK; }
  if ((size_t)n_cur * 8 + (size_t)dim * 4 > 150 * 1024) return LLD_ERR_UNSUPPORTED;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  const size_t pairs = (size_t)n_map * n_cur;
  auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
  size_t in = 0;
  const size_t o_q = in; in += pad((size_t)n_map * dim * 4);
  const size_t o_t = in; in += pad((size_t)n_cur * dim * 4);
  const size_t o_x0 = in; in += pad((size_t)n_map * 24); const size_t o_dr = in; in += pad((size_t)n_map * 24);
  const size_t o_x1 = in; in += pad((size_t)n_map * 24); const size_t o_x2 = in; in += pad((size_t)n_map * 24);
  const size_t o_sk = in; in += pad((size_t)n_map);
  const size_t o_ll = in; in += pad((size_t)n_cur * 16); const size_t o_lo = in; in += pad((size_t)n_cur * 4);
  const size_t o_rl = in; in += pad((size_t)std::max(n_right, 1) * 16);
  const size_t o_lm = in; in += pad((size_t)n_cur * 4); const size_t o_oc = in; in += pad((size_t)n_cur);
  size_t out = 0;
  const size_t r_m = out; out += pad((size_t)n_map * 4);
  const size_t r_d = out; out += pad((size_t)n_map * 8);
  const size_t r_g = out; out += pad(pairs);
  size_t dev = 0;
  const size_t s_cell = dev; dev += pad((size_t)n_cur * 4);
  const size_t s_mat = dev; dev += pad(pairs * 8);
  const size_t s_c = dev; dev += pad((size_t)n_map * sizeof(LineCand));
  void* hb; int st = lld_ctx_pinned(ctx, in + out, &hb); if (st) return st;
  void* db; st = lld_ctx_scratch(ctx, in + out + dev + 256, &db); if (st) return st;
  char* h = (char*)hb; char* d = (char*)db; char* h_out = h + in; char* d_out = d + in; char* d_dev = d_out + out;
  std::memcpy(h + o_q, map_desc, (size_t)n_map * dim * 4); std::memcpy(h + o_t, cur_desc, (size_t)n_cur * dim * 4);
  std::memcpy(h + o_x0, map_x0, (size_t)n_map * 24); std::memcpy(h + o_dr, map_dir, (size_t)n_map * 24);
  std::memcpy(h + o_x1, map_x1, (size_t)n_map * 24); std::memcpy(h + o_x2, map_x2, (size_t)n_map * 24);
  if (map_skip) std::memcpy(h + o_sk, map_skip, (size_t)n_map); else std::memset(h + o_sk, 0, (size_t)n_map);
  std::memcpy(h + o_ll, left_lines, (size_t)n_cur * 16); std::memcpy(h + o_lo, left_octave, (size_t)n_cur * 4);
  if (n_right > 0 && right_lines) std::memcpy(h + o_rl, right_lines, (size_t)n_right * 16);
  std::memcpy(h + o_lm, line_matches, (size_t)n_cur * 4);
  if (occupied) std::memcpy(h + o_oc, occupied, (size_t)n_cur); else std::memset(h + o_oc, 0, (size_t)n_cur);
  hipStream_t sm = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, in, hipMemcpyHostToDevice, sm));
  LineTrackParams P; std::memset(&P, 0, sizeof P);
  for (int i = 0; i < 9; i++) P.K[i] = prm->K[i];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) P.R[3 * r + c] = prm->T_curr[4 * r + c]; P.t[r] = prm->T_curr[4 * r + 3]; }
  for (int r = 0; r < 3; r++) P.tr[r] = P.t[r] + P.R[3 * r] * prm->b;            // GetTForRight: t + R (b, 0, 0)
  P.thr_base = prm->thr_reproj_base; P.sx = prm->sx; P.sy = prm->sy; P.monocular = prm->monocular; P.use_grid = prm->use_grid;
  int* dcell = reinterpret_cast<int*>(d_dev + s_cell);
  uint8_t* dgate = reinterpret_cast<uint8_t*>(d_out + r_g);
  double* dmat = reinterpret_cast<double*>(d_dev + s_mat);
  LineCand* dc = reinterpret_cast<LineCand*>(d_dev + s_c);
  hipLaunchKernelGGL(line_cells_kernel, dim3((n_cur + 255) / 256), dim3(256), 0, sm, reinterpret_cast<const float*>(d + o_ll), n_cur, P.sx, P.sy, dcell);
  hipLaunchKernelGGL(line_track_gate_kernel, dim3(n_map), dim3(64), 0, sm, P, reinterpret_cast<const double*>(d + o_x0), reinterpret_cast<const double*>(d + o_dr),
                     reinterpret_cast<const double*>(d + o_x1), reinterpret_cast<const double*>(d + o_x2), reinterpret_cast<const uint8_t*>(d + o_sk), n_cur,
                     reinterpret_cast<const float*>(d + o_ll), reinterpret_cast<const int*>(d + o_lo), reinterpret_cast<const float*>(d + o_rl),
                     reinterpret_cast<const int*>(d + o_lm), reinterpret_cast<const uint8_t*>(d + o_oc), dcell, dgate, nullptr);
  // `md > mdThr` rejects (Tracking.cc:1099): distances up to and including md_thr pass, where the stereo matcher's tau is strict
  const double tau = std::nextafter(prm->md_thr, 1.7976931348623157e308);
  const size_t lds = (size_t)n_cur * 8 + (size_t)dim * 4 + 16;
  if (lds > 48 * 1024) LLD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&line_candidates_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  
  LineGateParams G0; std::memset(&G0, 0, sizeof G0);
  hipLaunchKernelGGL(line_candidates_kernel<false>, dim3(n_map), dim3(64), lds, sm, G0, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const float*>(d + o_q),
                     reinterpret_cast<const float*>(d + o_t), dim, n_cur, dgate, tau, nullptr, dmat, dc);
  { const int rs = line_resolve_prepare(n_map, n_cur); if (rs) return rs; }
  hipLaunchKernelGGL(line_resolve_kernel, dim3(1), dim3(kResolveThreads), line_resolve_lds(n_map, n_cur), sm, dc, dmat, dgate, n_map, n_cur, tau,
                     reinterpret_cast<int*>(d_out + r_m), reinterpret_cast<double*>(d_out + r_d), lld_track::LineApplyDev{}, nullptr, nullptr);
  LLD_HIP_TRY(hipGetLastError());
  LLD_HIP_TRY(hipMemcpyAsync(h_out, d_out, gate_out ? out : r_g, hipMemcpyDeviceToHost, sm));
  LLD_HIP_TRY(hipStreamSynchronize(sm));
  std::memcpy(matches, h_out + r_m, (size_t)n_map * 4);
  if (match_dist) std::memcpy(match_dist, h_out + r_d, (size_t)n_map * 8);
  if (gate_out) std::memcpy(gate_out, h_out + r_g, pairs);
  return LLD_OK;
}

int lld_line_match_last_frame(lld_ctx* ctx, const lld_line_lastkf_params* prm, int n_cur, const float* cur_left, int n_cur_right, const float* cur_right,
                              const int32_t* cur_line_matches, const uint8_t* cur_occupied, const float* cur_desc, int n_last, const float* last_left,
                              const int32_t* last_left_octave, int n_last_right, const float* last_right, const int32_t* last_line_matches,
                              const uint8_t* last_skip, const float* last_desc, int dim, int32_t* match_last, uint8_t* created, double* x0, double* dir) {
  if (!ctx || !prm || n_cur < 0 || n_last < 0 || n_cur_right < 0 || n_last_right < 0 || dim <= 0 || !match_last || !created || !x0 || !dir) return LLD_ERR_INVALID;
  if (n_cur > 0 && (!cur_left || !cur_line_matches || !cur_desc)) return LLD_ERR_INVALID;
  if (n_last > 0 && (!last_left || !last_left_octave || !last_line_matches || !last_desc)) return LLD_ERR_INVALID;
  if ((n_cur_right > 0 && !cur_right) || (n_last_right > 0 && !last_right)) return LLD_ERR_INVALID;
  if (!(prm->sx > 0) || !(prm->sy > 0)) return LLD_ERR_INVALID;
  for (int i = 0; i < n_cur; i++) if (cur_line_matches[i] >= n_cur_right) return LLD_ERR_INVALID;
  for (int i = 0; i < n_last; i++) if (last_line_matches[i] >= n_last_right || last_left_octave[i] < 0 || last_left_octave[i] > 64) return LLD_ERR_INVALID;
  if (dim > 4096) return LLD_ERR_UNSUPPORTED;
  if (n_cur == 0) return LLD_OK;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
  size_t in = 0;
  const size_t o_cl = in; in += pad((size_t)n_cur * 16); const size_t o_cr = in; in += pad((size_t)std::max(n_cur_right, 1) * 16);
  const size_t o_clm = in; in += pad((size_t)n_cur * 4); const size_t o_co = in; in += pad((size_t)n_cur);
  const size_t o_cd = in; in += pad((size_t)n_cur * dim * 4);
  const size_t o_ll = in; in += pad((size_t)std::max(n_last, 1) * 16); const size_t o_lo = in; in += pad((size_t)std::max(n_last, 1) * 4);
  const size_t o_lr = in; in += pad((size_t)std::max(n_last_right, 1) * 16); const size_t o_llm = in; in += pad((size_t)std::max(n_last, 1) * 4);
  const size_t o_ls = in; in += pad((size_t)std::max(n_last, 1)); const size_t o_ld = in; in += pad((size_t)std::max(n_last, 1) * dim * 4);
  size_t out = 0;
  const size_t r_m = out; out += pad((size_t)n_cur * 4); const size_t r_c = out; out += pad((size_t)n_cur);
  const size_t r_x = out; out += pad((size_t)n_cur * 24); const size_t r_d = out; out += pad((size_t)n_cur * 24);
  const size_t s_cell = pad((size_t)std::max(n_last, 1) * 4);
  void* hb; int st = lld_ctx_pinned(ctx, in + out, &hb); if (st) return st;
  void* db; st = lld_ctx_scratch(ctx, in + out + s_cell + 256, &db); if (st) return st;
  char* h = (char*)hb; char* d = (char*)db; char* h_out = h + in; char* d_out = d + in; char* d_dev = d_out + out;
  std::memcpy(h + o_cl, cur_left, (size_t)n_cur * 16); if (n_cur_right) std::memcpy(h + o_cr, cur_right, (size_t)n_cur_right * 16);
  std::memcpy(h + o_clm, cur_line_matches, (size_t)n_cur * 4);
  if (cur_occupied) std::memcpy(h + o_co, cur_occupied, (size_t)n_cur); else std::memset(h + o_co, 0, (size_t)n_cur);
  std::memcpy(h + o_cd, cur_desc, (size_t)n_cur * dim * 4);
  if (n_last) {
    std::memcpy(h + o_ll, last_left, (size_t)n_last * 16); std::memcpy(h + o_lo, last_left_octave, (size_t)n_last * 4);
    if (n_last_right) std::memcpy(h + o_lr, last_right, (size_t)n_last_right * 16);
    std::memcpy(h + o_llm, last_line_matches, (size_t)n_last * 4);
    if (last_skip) std::memcpy(h + o_ls, last_skip, (size_t)n_last); else std::memset(h + o_ls, 0, (size_t)n_last);
    std::memcpy(h + o_ld, last_desc, (size_t)n_last * dim * 4);
  }
  hipStream_t sm = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, in, hipMemcpyHostToDevice, sm));
  LastKfParams P; std::memset(&P, 0, sizeof P);
  for (int i = 0; i < 9; i++) P.K[i] = prm->K[i];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { P.R[3 * r + c] = prm->T_curr[4 * r + c]; P.Rl[3 * r + c] = prm->T_last[4 * r + c]; } P.t[r] = prm->T_curr[4 * r + 3]; P.tl[r] = prm->T_last[4 * r + 3]; }
  for (int r = 0; r < 3; r++) { P.tr[r] = P.t[r] + P.R[3 * r] * prm->b; P.tlr[r] = P.tl[r] + P.Rl[3 * r] * prm->b; }      // GetTForRight
  P.thr_base = prm->thr_reproj_base; P.md_thr = prm->md_thr; P.sx = prm->sx; P.sy = prm->sy; P.use_grid = prm->use_grid;
  int* dcell = reinterpret_cast<int*>(d_dev);
  if (n_last > 0) hipLaunchKernelGGL(line_cells_kernel, dim3((n_last + 255) / 256), dim3(256), 0, sm, reinterpret_cast<const float*>(d + o_ll), n_last, P.sx, P.sy, dcell);
  hipLaunchKernelGGL(line_lastkf_kernel, dim3(n_cur), dim3(64), (size_t)dim * 4 + 16, sm, P, n_cur, reinterpret_cast<const float*>(d + o_cl),
                     reinterpret_cast<const float*>(d + o_cr), reinterpret_cast<const int*>(d + o_clm), reinterpret_cast<const uint8_t*>(d + o_co),
                     reinterpret_cast<const float*>(d + o_cd), n_last, reinterpret_cast<const float*>(d + o_ll), reinterpret_cast<const int*>(d + o_lo),
                     reinterpret_cast<const float*>(d + o_lr), reinterpret_cast<const int*>(d + o_llm), reinterpret_cast<const uint8_t*>(d + o_ls),
                     reinterpret_cast<const float*>(d + o_ld), dcell, dim, reinterpret_cast<int*>(d_out + r_m), reinterpret_cast<uint8_t*>(d_out + r_c),
                     reinterpret_cast<double*>(d_out + r_x), reinterpret_cast<double*>(d_out + r_d));
  LLD_HIP_TRY(hipGetLastError());
  LLD_HIP_TRY(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, sm));
  LLD_HIP_TRY(hipStreamSynchronize(sm));
  std::memcpy(match_last, h_out + r_m, (size_t)n_cur * 4); std::memcpy(created, h_out + r_c, (size_t)n_cur);
  std::memcpy(x0, h_out + r_x, (size_t)n_cur * 24); std::memcpy(dir, h_out + r_d, (size_t)n_cur * 24);
  return LLD_OK;
}

}  // extern "C"

// ================================================================ Tracking::AddLinesFrom on device arrays (lld_track_internal.h): the kernels of
// lld_line_track_match with every operand already in HBM and the camera read from device memory; nothing is copied, nothing waits.
namespace lld_track {
static_assert(sizeof(LineTrackDevParams) == sizeof(LineTrackParams), "lld_track_internal.h restates LineTrackParams");
static inline size_t lt_pad(size_t b) { return (b + 255) & ~size_t(255); }
size_t line_track_work_bytes(int n_map, int n_cur) {
  const size_t pairs = (size_t)n_map * (size_t)n_cur;
  return lt_pad(pairs) + lt_pad(pairs * 8) + lt_pad((size_t)n_map * sizeof(LineCand)) + 256;
}
int line_cells_dev(hipStream_t st, const float* d_left, int n, double sx, double sy, int32_t* d_cell) {
  if (n > 0) hipLaunchKernelGGL(line_cells_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_left, n, sx, sy, d_cell);
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}
int line_track_launch_dev(lld_ctx* ctx, hipStream_t st, const LineTrackDevParams* params_d, const LineMapDev& map, const LineFrameDev& cur, double md_thr,
                          void* d_work, int32_t* matches_d, const LineApplyDev& apply) {
  (void)ctx;
  const int n_map = map.n, n_cur = cur.n_cur, dim = cur.dim;
  if (n_map <= 0 || n_cur <= 0) return LLD_OK;                               // (nothing to associate: the caller does not read matches_d then)
  if (dim > 128 || (size_t)n_cur * 8 + (size_t)dim * 4 > 150 * 1024) return LLD_ERR_UNSUPPORTED;
  const size_t pairs = (size_t)n_map * (size_t)n_cur;
  char* w = static_cast<char*>(d_work);
  uint8_t* dgate = reinterpret_cast<uint8_t*>(w);
  double* dmat = reinterpret_cast<double*>(w + lt_pad(pairs));
  LineCand* dc = reinterpret_cast<LineCand*>(w + lt_pad(pairs) + lt_pad(pairs * 8));
  LineTrackParams P0; std::memset(&P0, 0, sizeof P0);
  hipLaunchKernelGGL(line_track_gate_kernel, dim3(n_map), dim3(64), 0, st, P0, map.x0, map.dir, map.x1, map.x2, map.skip, n_cur, cur.left, cur.loct, cur.right,
                     cur.lmatch, cur.occupied, cur.cell, dgate, reinterpret_cast<const LineTrackParams*>(params_d));
  const double tau = std::nextafter(md_thr, 1.7976931348623157e308);         // `md > mdThr` rejects (Tracking.cc:1099)
  const size_t lds = (size_t)n_cur * 8 + (size_t)dim * 4 + 16;
  if (lds > 48 * 1024) LLD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&line_candidates_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  
  LineGateParams G0; std::memset(&G0, 0, sizeof G0);
  hipLaunchKernelGGL(line_candidates_kernel<false>, dim3(n_map), dim3(64), lds, st, G0, nullptr, nullptr, nullptr, nullptr, map.desc, cur.desc, dim, n_cur, dgate, tau,
                     nullptr, dmat, dc);
  { const int rs = line_resolve_prepare(n_map, n_cur); if (rs) return rs; }
  hipLaunchKernelGGL(line_resolve_kernel, dim3(1), dim3(kResolveThreads), line_resolve_lds(n_map, n_cur), st, dc, dmat, dgate, n_map, n_cur, tau, matches_d, nullptr, apply, map.x0, map.dir);
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}
}  // namespace lld_track
