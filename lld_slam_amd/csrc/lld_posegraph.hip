// lld_posegraph.hip — Optimizer::OptimizeEssentialGraph (src/Optimizer.cc:1391-1654), the optimisation proper: Sim3 vertices, EdgeSim3
// edges (error = log(Sji * Siw * Sjw^-1), identity information, numeric Jacobians for both vertices as g2o computes them),
// Levenberg-Marquardt with a user lambda (1e-16) on the 7N x 7N system.
//   pg_errors     lane <-> edge: the 7-vector errors at the current estimate, chi2
//   pg_linearize  lane <-> (edge, vertex, dof): central differences on the vertex's oplus (delta 1e-9) -> J[e][v] (7x7)
//   pg_diag       lane group <-> unknown vertex: b_v = -sum J^T e and H_vv = sum J^T J over its incident edges (vertex CSR, fixed order)
// Linear solve, default: the 7nu x 7nu system is formed DENSE in HBM (a 4000-keyframe map is 6 GB of 288) and factored on the
// fp64 matrix cores - a pose graph is a chain with a few loops, the worst case for a block-Jacobi PCG (thousands of iterations),
// while n^3/3 flops are nothing for this chip:
//   pg_dense_fill   lower triangle of H + lambda I from the edge Jacobians, the right-hand side as one more matrix row
//   pg_chol_col     left-looking Cholesky on 16x16 tiles, one launch per tile column: workgroup <-> row tile I >= J computes
//                   A_IJ - sum_K L_IK L_JK^T with v_mfma_f64_16x16x4_f64 and multiplies by L_JJ^-T (every workgroup factors the
//                   diagonal tile itself, so a column needs no grid-wide synchronisation); the right-hand side row comes out as
//                   y = L^-1 b
//   pg_chol_back    L^T x = y, one launch per tile column from the last: x_J = L_JJ^-T y_J, y_I -= L_JI^T x_J for I < J
// Beyond 2048 tiles (n > 32768) or on request the system is instead applied edge by edge (H p = sum_e J_e^T (J_e p)) through
// the vertex CSR, memory O(edges), every sum in a fixed order:
//   pg_pcg_init / pg_matvec / pg_pcg_update   block-Jacobi PCG ((H_vv + lambda I)^-1 per vertex) with the host polling `done`
//   pg_update     V <- exp(x_v) V (the scale update zeroed in place when fix_scale), scale = sum x (lambda x + b)
// The LM control (push / pop, rho, lambda schedule, stop rules of optimization_algorithm_levenberg.cpp:61-164) runs on the host:
// a loop closure happens once in a while, latency of a few polls is irrelevant.
#include <algorithm>
#include <vector>
#include "lld_common.h"
#include "lld_device_math.h"
#include "lld_sim3_math.h"

namespace {

constexpr int kPgThreads = 256;

struct PgArrays {
  int N, E, nu;
  double* V;                 // [N][8] current estimate
  const int* hidx;           // [N] unknown index or -1
  const int* ei; const int* ej;
  const double* C;           // [E][8]
  double* err;               // [E][7]
  double* J;                 // [E][2][49] row-major 7x7 (error row, dof)
  const int* vstart;         // [nu+1] CSR: incident (edge, role) of every unknown vertex, in edge order
  const int* vinc;           // entry = edge * 2 + role (0: the vertex is edge_i, 1: edge_j)
  const int* uvert;          // [nu] vertex id of an unknown
  double* Hd;                // [nu][49]  H_vv
  double* Mi;                // [nu][49]  (H_vv + lambda I)^-1
  double* b; double* x; double* r; double* z; double* p; double* ap;   // [7 nu]
  double* sc;                // scalars: 0 rz, 1 stop, 2 iterations, 3 done, 4 ok, 5 chi2, 6 scale
  int fix_scale;
};

__device__ __forceinline__ void edge_error(const Sim3& C, const Sim3& v1, const Sim3& v2, double* e) {
  sim3_log(sim3_mul(sim3_mul(C, v1), sim3_inverse(v2)), e);            // Sim3 error_ = C * v1 * v2.inverse(); _error = error_.log()
}
__device__ __forceinline__ Sim3 oplus(const Sim3& S, const double* upd, int fix_scale) {
  double u[7];
#pragma unroll
  for (int i = 0; i < 7; i++) u[i] = upd[i];
  if (fix_scale) u[6] = 0;
  return sim3_mul(sim3_exp(u), S);
}

// one workgroup: errors of all edges, chi2 = sum e^T e in a fixed order
__global__ __launch_bounds__(1024) void pg_errors_kernel(PgArrays A) {
  __shared__ double scratch[32];
  const int tid = threadIdx.x;
  double part = 0.0;
  for (int e = tid; e < A.E; e += 1024) {
    double er[7];
    edge_error(sim3_load(A.C + 8 * (size_t)e), sim3_load(A.V + 8 * (size_t)A.ei[e]), sim3_load(A.V + 8 * (size_t)A.ej[e]), er);
    double c = 0.0;
#pragma unroll
    for (int k = 0; k < 7; k++) { A.err[7 * (size_t)e + k] = er[k]; c += er[k] * er[k]; }
    part += c;
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) { double s = 0.0; for (int w = 0; w < 16; w++) s += scratch[w]; A.sc[5] = s; }
}

// lane <-> (edge, vertex role, dof)
__global__ __launch_bounds__(kPgThreads) void pg_linearize_kernel(PgArrays A) {
  const long long id = (long long)blockIdx.x * kPgThreads + threadIdx.x;
  if (id >= (long long)A.E * 14) return;
  const int e = (int)(id / 14), pq = (int)(id % 14), role = pq / 7, d = pq % 7;
  const int vi = A.ei[e], vj = A.ej[e];
  const int vid = role == 0 ? vi : vj;
  if (A.hidx[vid] < 0) return;
  const Sim3 C = sim3_load(A.C + 8 * (size_t)e);
  const Sim3 S1 = sim3_load(A.V + 8 * (size_t)vi), S2 = sim3_load(A.V + 8 * (size_t)vj);
  double add_v[7] = {0, 0, 0, 0, 0, 0, 0}, ep[7], em[7];
  add_v[d] = 1e-9;
  const Sim3 Pp = oplus(role == 0 ? S1 : S2, add_v, A.fix_scale);
  add_v[d] = -1e-9;
  const Sim3 Pm = oplus(role == 0 ? S1 : S2, add_v, A.fix_scale);
  if (role == 0) { edge_error(C, Pp, S2, ep); edge_error(C, Pm, S2, em); }
  else { edge_error(C, S1, Pp, ep); edge_error(C, S1, Pm, em); }
  const double scalar = 1.0 / (2 * 1e-9);
  double* Jd = A.J + ((size_t)e * 2 + role) * 49;
#pragma unroll
  for (int r = 0; r < 7; r++) Jd[r * 7 + d] = scalar * (ep[r] - em[r]);
}

// lane <-> (unknown vertex, entry of H_vv or of b_v): 56 lanes of a wavefront per vertex
__global__ __launch_bounds__(64) void pg_diag_kernel(PgArrays A) {
  const int u = blockIdx.x, t = threadIdx.x;
  if (u >= A.nu || t >= 56) return;
  double acc = 0.0;
  for (int q = A.vstart[u]; q < A.vstart[u + 1]; q++) {
    const int e = A.vinc[q] >> 1, role = A.vinc[q] & 1;
    const double* Jd = A.J + ((size_t)e * 2 + role) * 49;
    if (t < 49) {
      const int r = t / 7, c = t % 7;
#pragma unroll
      for (int k = 0; k < 7; k++) acc += Jd[k * 7 + r] * Jd[k * 7 + c];
    } else {
      const int r = t - 49;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 7; k++) s += Jd[k * 7 + r] * A.err[7 * (size_t)e + k];
      acc -= s;
    }
  }
  if (t < 49) A.Hd[(size_t)u * 49 + t] = acc; else A.b[(size_t)u * 7 + (t - 49)] = acc;
}

// (H_vv + lambda I)^-1 by LDL^T; ok = 0 when a pivot is not positive.  lane <-> vertex
__device__ bool inv7(const double* Hm, double lambda, double* out) {
  double L[7][7], D[7];
  for (int j = 0; j < 7; j++) {
    double d = Hm[j * 7 + j] + lambda;
    for (int p = 0; p < j; p++) d -= L[j][p] * L[j][p] * D[p];
    if (!(d > 0.0) || !isfinite(d)) return false;
    D[j] = d;
    for (int i = j + 1; i < 7; i++) {
      double s = Hm[i * 7 + j];
      for (int p = 0; p < j; p++) s -= L[i][p] * L[j][p] * D[p];
      L[i][j] = s / d;
    }
  }
  for (int c = 0; c < 7; c++) {                       // solve for the unit vectors
    double y[7];
    for (int i = 0; i < 7; i++) { double s = i == c ? 1.0 : 0.0; for (int p = 0; p < i; p++) s -= L[i][p] * y[p]; y[i] = s; }
    for (int i = 0; i < 7; i++) y[i] /= D[i];
    double xs[7];
    for (int i = 6; i >= 0; i--) { double s = y[i]; for (int p = i + 1; p < 7; p++) s -= L[p][i] * xs[p]; xs[i] = s; }
    for (int i = 0; i < 7; i++) out[i * 7 + c] = xs[i];
  }
  return true;
}

__global__ __launch_bounds__(1024) void pg_pcg_init_kernel(PgArrays A, double lambda, double tol) {
  __shared__ double scratch[32];
  __shared__ int ok_s;
  const int tid = threadIdx.x, n = 7 * A.nu;
  if (tid == 0) ok_s = 1;
  __syncthreads();
  for (int u = tid; u < A.nu; u += 1024) if (!inv7(A.Hd + (size_t)u * 49, lambda, A.Mi + (size_t)u * 49)) ok_s = 0;
  for (int i = tid; i < n; i += 1024) { A.x[i] = 0.0; A.r[i] = A.b[i]; }
  __syncthreads();
  double part = 0.0;
  for (int i = tid; i < n; i += 1024) {
    const int u = i / 7, rr = i - u * 7;
    double zv = 0.0;
#pragma unroll
    for (int c = 0; c < 7; c++) zv += A.Mi[(size_t)u * 49 + rr * 7 + c] * A.r[u * 7 + c];
    A.z[i] = zv; A.p[i] = zv;
    part += A.r[i] * zv;
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) {
    double rz0 = 0.0; for (int w = 0; w < 16; w++) rz0 += scratch[w];
    const bool ok = ok_s != 0 && isfinite(rz0);
    A.sc[0] = rz0; A.sc[1] = tol * tol * rz0; A.sc[2] = 0.0; A.sc[3] = (ok && rz0 > 0.0) ? 0.0 : 1.0; A.sc[4] = ok ? 1.0 : 0.0;
  }
}

// ap_v = lambda p_v + sum_{e incident to v} J_{e,v}^T (J_{e,i} p_i + J_{e,j} p_j): 8 lanes per unknown vertex (7 used)
__global__ __launch_bounds__(kPgThreads) void pg_matvec_kernel(PgArrays A, double lambda) {
  if (A.sc[3] != 0.0) return;
  const int g = (blockIdx.x * kPgThreads + threadIdx.x) >> 3, l = threadIdx.x & 7;
  if (g >= A.nu) return;
  double acc = l < 7 ? lambda * A.p[(size_t)g * 7 + l] : 0.0;
  for (int q = A.vstart[g]; q < A.vstart[g + 1]; q++) {
    const int e = A.vinc[q] >> 1, role = A.vinc[q] & 1;
    const int hi = A.hidx[A.ei[e]], hj = A.hidx[A.ej[e]];
    const double* Ji = A.J + ((size_t)e * 2) * 49; const double* Jj = Ji + 49;
    double w = 0.0;                                      // lane l: component l of J_i p_i + J_j p_j
    if (l < 7) {
      if (hi >= 0) for (int d = 0; d < 7; d++) w += Ji[l * 7 + d] * A.p[(size_t)hi * 7 + d];
      if (hj >= 0) for (int d = 0; d < 7; d++) w += Jj[l * 7 + d] * A.p[(size_t)hj * 7 + d];
    }
    const double* Jv = role == 0 ? Ji : Jj;
#pragma unroll
    for (int k = 0; k < 7; k++) {
      const double wk = __shfl(w, k, 8);
      if (l < 7) acc += Jv[k * 7 + l] * wk;
    }
  }
  if (l < 7) A.ap[(size_t)g * 7 + l] = acc;
}

__global__ __launch_bounds__(1024) void pg_pcg_update_kernel(PgArrays A, int max_iter) {
  __shared__ double scratch[32];
  if (A.sc[3] != 0.0) return;
  const int tid = threadIdx.x, n = 7 * A.nu;
  const double rz = A.sc[0], stop = A.sc[1];
  auto bsum = [&](double part) {
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = part;
    __syncthreads();
    double s = 0.0; for (int w = 0; w < 16; w++) s += scratch[w];
    return s;
  };
  double part = 0.0;
  for (int i = tid; i < n; i += 1024) part += A.p[i] * A.ap[i];
  const double pAp = bsum(part);
  if (!(pAp > 0.0) || !isfinite(pAp)) { if (tid == 0) { A.sc[3] = 1.0; A.sc[4] = 0.0; } return; }
  const double alpha = rz / pAp;
  for (int i = tid; i < n; i += 1024) { A.x[i] += alpha * A.p[i]; A.r[i] -= alpha * A.ap[i]; }
  __syncthreads();
  part = 0.0;
  for (int i = tid; i < n; i += 1024) {
    const int u = i / 7, rr = i - u * 7;
    double zv = 0.0;
#pragma unroll
    for (int c = 0; c < 7; c++) zv += A.Mi[(size_t)u * 49 + rr * 7 + c] * A.r[u * 7 + c];
    A.z[i] = zv;
    part += A.r[i] * zv;
  }
  const double rz_new = bsum(part);
  const double iters = A.sc[2] + 1.0;
  bool done = false, ok = true;
  if (!isfinite(rz_new)) { done = true; ok = false; }
  else if (rz_new <= stop || iters >= (double)max_iter) done = true;
  if (!done) { const double beta = rz_new / rz; for (int i = tid; i < n; i += 1024) A.p[i] = A.z[i] + beta * A.p[i]; }
  __syncthreads();
  if (tid == 0) { A.sc[0] = rz_new; A.sc[2] = iters; if (done) A.sc[3] = 1.0; if (!ok) A.sc[4] = 0.0; }
}

// V <- exp(x_v) V for the unknown vertices; scale = sum_j x_j (lambda x_j + b_j) with the zeroed scale components
__global__ __launch_bounds__(1024) void pg_update_kernel(PgArrays A, double lambda) {
  __shared__ double scratch[32];
  const int tid = threadIdx.x;
  double part = 0.0;
  for (int u = tid; u < A.nu; u += 1024) {
    double* xu = A.x + (size_t)u * 7;
    if (A.fix_scale) xu[6] = 0;                               // oplusImpl zeroes update[6] in the solver's x
    const int v = A.uvert[u];
    sim3_store(oplus(sim3_load(A.V + 8 * (size_t)v), xu, A.fix_scale), A.V + 8 * (size_t)v);
#pragma unroll
    for (int k = 0; k < 7; k++) part += xu[k] * (lambda * xu[k] + A.b[(size_t)u * 7 + k]);
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) { double s = 0.0; for (int w = 0; w < 16; w++) s += scratch[w]; A.sc[6] = s; }
}


// ---------------------------------------------------------------- dense Cholesky of H + lambda I on the matrix cores
constexpr int kPgTS = 17;                                  // padded LDS row of a 16x16 tile
constexpr int kPgMaxTiles = 2048;
typedef double pg_v4d __attribute__((ext_vector_type(4)));

struct PgDense {
  double* M;                 // [(NT+1)*16][ld] row-major, lower triangle: H + lambda I, then L below the diagonal tiles (those stay
                             // as assembled); row 16*NT: b, then y = L^-1 b
  double* Linv;              // [NT][16][16] inverses of the diagonal tiles of L
  int NT, n;                 // tile columns, unknowns (<= 16*NT; the padding rows are identity)
  size_t ld;                 // 16*NT
  // tile pattern of L from the host's symbolic factorisation (a pose graph in keyframe order is a band plus the fill of a few
  // loop edges): the kernels only visit non-zero tiles, the zero ones stay as the memset left them
  const int* col_start; const int* col_rows;     // per tile column J: the rows I > J with L_IJ != 0, ascending
  const int* row_start; const int* row_cols;     // per tile row J: the columns K < J with L_JK != 0, ascending
};

__device__ __forceinline__ double pg_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// One wavefront: factor the tile T (LDS, lower triangle used) in place, L^-1 -> Li.  Lane r < 16 holds row r; lanes 16..31 carry
// the identity as extra rows, which the same right-looking recurrence turns into the columns of L^-1 (readlane broadcasts only).
__device__ __forceinline__ bool pg_tile_factor(double* T, double* Li, int lane) {
  const int r = lane & 31;
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { const double v = T[(r & 15) * kPgTS + c]; a[c] = r < 16 ? v : (r - 16 == c ? 1.0 : 0.0); }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const double d = pg_readlane(a[c], c);
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    double inv = __builtin_amdgcn_rsq(d);                           // 1/sqrt(d): seed + two Newton steps
    inv = inv * (1.5 - (0.5 * d) * (inv * inv));
    inv = inv * (1.5 - (0.5 * d) * (inv * inv));
    const double lc = a[c] * inv;
    a[c] = lc;
#pragma unroll
    for (int c2 = c + 1; c2 < 16; c2++) a[c2] -= lc * pg_readlane(lc, c2);
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; c++) T[r * kPgTS + c] = (c <= r) ? a[c] : 0.0;
  } else if (lane < 32) {                                           // lane 16+k holds column k of L^-1
#pragma unroll
    for (int c = 0; c < 16; c++) Li[c * kPgTS + (lane - 16)] = a[c];
  }
  return ok;
}

// lane <-> entry of an off-diagonal 7x7 block (per edge), of a diagonal block (per unknown), of the right-hand side / the padding
__global__ __launch_bounds__(kPgThreads) void pg_dense_fill_kernel(PgArrays A, PgDense D, double lambda) {
  const long long id = (long long)blockIdx.x * kPgThreads + threadIdx.x;
  const long long n_off = (long long)A.E * 49, n_diag = (long long)A.nu * 49, N = 16ll * D.NT;
  if (id == 0) { A.sc[4] = 1.0; A.sc[2] = 0.0; }
  if (id < n_off) {
    const int e = (int)(id / 49), t = (int)(id % 49), r = t / 7, c = t % 7;
    const int hi = A.hidx[A.ei[e]], hj = A.hidx[A.ej[e]];
    if (hi < 0 || hj < 0 || hi == hj) return;
    const double* Ji = A.J + ((size_t)e * 2) * 49; const double* Jj = Ji + 49;
    const double* Jr = hi > hj ? Ji : Jj; const double* Jc = hi > hj ? Jj : Ji;       // block (max, min) = J_max^T J_min
    const int ur = hi > hj ? hi : hj, uc = hi > hj ? hj : hi;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 7; k++) v += Jr[k * 7 + r] * Jc[k * 7 + c];
    atomicAdd(D.M + (size_t)(7 * ur + r) * D.ld + 7 * uc + c, v);
  } else if (id < n_off + n_diag) {
    const long long q = id - n_off;
    const int u = (int)(q / 49), t = (int)(q % 49), r = t / 7, c = t % 7;
    D.M[(size_t)(7 * u + r) * D.ld + 7 * u + c] = A.Hd[(size_t)u * 49 + t] + (r == c ? lambda : 0.0);
  } else if (id < n_off + n_diag + N) {
    const long long i = id - n_off - n_diag;
    if (i < D.n) D.M[(size_t)N * D.ld + i] = A.b[i];
    else D.M[(size_t)i * D.ld + i] = 1.0;
  }
}

// grid (2 + #rows of column J) x 256 lanes: workgroup <-> row tile I = J, NT (the right-hand side) or a non-zero row of the
// column.  K runs over the non-zero tiles of row J (a zero L_JK contributes nothing whatever L_IK is).  The K loop of both
// sums (diagonal tile and own tile) is split over the four wavefronts, four tile columns in flight per wavefront: the loop is a
// chain of L2 round trips, not of matrix-core work.
__global__ __launch_bounds__(256) void pg_chol_col_kernel(PgDense D, int J, double* sc) {
  __shared__ __attribute__((aligned(16))) double Dg[16 * kPgTS], Li[16 * kPgTS], T[16 * kPgTS], red_d[4][256], red_t[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane >> 4, lcol = lane & 15;
  const int I = blockIdx.x == 0 ? J : (blockIdx.x == 1 ? D.NT : D.col_rows[D.col_start[J] + (int)blockIdx.x - 2]);
  const bool own = I != J;
  const int* Ks = D.row_cols + D.row_start[J];
  const int nK = D.row_start[J + 1] - D.row_start[J];
  const double* rowJ = D.M + (size_t)(16 * J + lcol) * D.ld + 4 * lrow;      // MFMA operand: row lcol of tile row J, k = 4 lrow + kk
  const double* rowI = D.M + (size_t)(16 * I + lcol) * D.ld + 4 * lrow;
  pg_v4d cd = {0.0, 0.0, 0.0, 0.0}, ct = {0.0, 0.0, 0.0, 0.0};
  const pg_v4d zero = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = wave; k0 < nK; k0 += 16) {
    pg_v4d a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bool valid = k0 + 4 * q < nK;
      const int K = valid ? Ks[k0 + 4 * q] : 0;
      b[q] = valid ? *reinterpret_cast<const pg_v4d*>(rowJ + 16 * K) : zero;
      a[q] = (valid && own) ? *reinterpret_cast<const pg_v4d*>(rowI + 16 * K) : zero;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        cd = __builtin_amdgcn_mfma_f64_16x16x4f64(b[q][kk], b[q][kk], cd, 0, 0, 0);
        ct = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], b[q][kk], ct, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 4; g++) { red_d[wave][g * 64 + lane] = cd[g]; red_t[wave][g * 64 + lane] = ct[g]; }
  __syncthreads();
  if (wave == 0) {                                                   // diagonal tile A_JJ - sum_K L_JK L_JK^T
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const double sum = red_d[0][g * 64 + lane] + red_d[1][g * 64 + lane] + red_d[2][g * 64 + lane] + red_d[3][g * 64 + lane];
      Dg[(lrow + 4 * g) * kPgTS + lcol] = D.M[(size_t)(16 * J + lrow + 4 * g) * D.ld + 16 * J + lcol] - sum;
    }
  } else if (wave == 1 && own) {                                     // own tile A_IJ - sum_K L_IK L_JK^T
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const double sum = red_t[0][g * 64 + lane] + red_t[1][g * 64 + lane] + red_t[2][g * 64 + lane] + red_t[3][g * 64 + lane];
      T[(lrow + 4 * g) * kPgTS + lcol] = D.M[(size_t)(16 * I + lrow + 4 * g) * D.ld + 16 * J + lcol] - sum;
    }
  }
  __syncthreads();
  bool ok = true;
  if (wave == 0) ok = pg_tile_factor(Dg, Li, lane);
  __syncthreads();
  if (wave != 0) return;
  if (!own) {
    // publish the inverse of L_JJ (all the back substitution needs); the tile itself stays as assembled: the other workgroups
    // of this launch are still reading it for their own copy of the factorisation
#pragma unroll
    for (int g = 0; g < 4; g++) {
      D.Linv[(size_t)J * 256 + (lrow + 4 * g) * 16 + lcol] = (lcol <= lrow + 4 * g) ? Li[(lrow + 4 * g) * kPgTS + lcol] : 0.0;
    }
    if (!ok && lane == 0) sc[4] = 0.0;
    return;
  }
  // L_IJ = T L_JJ^-T; T is read back from LDS in the operand layout
  pg_v4d o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; kk++) o = __builtin_amdgcn_mfma_f64_16x16x4f64(T[lcol * kPgTS + 4 * lrow + kk], Li[lcol * kPgTS + 4 * lrow + kk], o, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 4; g++) D.M[(size_t)(16 * I + lrow + 4 * g) * D.ld + 16 * J + lcol] = o[g];
}

// grid ceil((1 + #non-zero tiles of row J) / 4) x 256 lanes: wavefront <-> tile I = J or a non-zero column of row J.
// x_J = L_JJ^-T y_J (every wavefront), y_I -= L_JI^T x_J
__global__ __launch_bounds__(256) void pg_chol_back_kernel(PgDense D, int J, double* x) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, part = lane >> 4, c = lane & 15;
  const int slot = 4 * (int)blockIdx.x + wave, nK = D.row_start[J + 1] - D.row_start[J];
  if (slot > nK) return;
  const int I = slot == 0 ? J : D.row_cols[D.row_start[J] + slot - 1];
  double* y = D.M + (size_t)(16 * D.NT) * D.ld;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 4; q++) s += D.Linv[(size_t)J * 256 + (4 * part + q) * 16 + c] * y[16 * J + 4 * part + q];
  s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);                     // x_J[c] on every lane with this c
  if (I == J) { if (lane < 16 && 16 * J + c < D.n) x[16 * J + c] = s; return; }
  double t = 0.0;
#pragma unroll
  for (int q = 0; q < 4; q++) t += D.M[(size_t)(16 * J + 4 * part + q) * D.ld + 16 * I + c] * __shfl(s, 4 * part + q);
  t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
  if (lane < 16) y[16 * I + c] -= t;
}

inline size_t pad256(size_t b) { return (b + 255) & ~size_t(255); }

}  // namespace

extern "C" void lld_pose_graph_params_default(lld_pose_graph_params* p) {
  if (!p) return;
  p->iterations = 15; p->fix_scale = 1; p->lambda_init = 1e-16; p->max_trials = 10; p->pcg_max_iter = 0; p->pcg_rel_tol = 1e-12; p->solver = 0; p->reserved = 0;
}

extern "C" int lld_optimize_essential_graph(lld_ctx* ctx, const lld_pose_graph* g, const lld_pose_graph_params* params, lld_pose_graph_result* out) {
  if (!ctx || !g || !out || !out->sim3) return LLD_ERR_INVALID;
  lld_pose_graph_params prm; if (params) prm = *params; else lld_pose_graph_params_default(&prm);
  const int N = g->n_vertices, E = g->n_edges;
  if (N < 0 || E < 0 || (N > 0 && !g->sim3) || (E > 0 && (!g->edge_i || !g->edge_j || !g->edge_sji))) return LLD_ERR_INVALID;
  if (prm.iterations < 0 || prm.max_trials <= 0 || !(prm.pcg_rel_tol > 0) || prm.solver < 0 || prm.solver > 2) return LLD_ERR_INVALID;
  for (int e = 0; e < E; e++) if (g->edge_i[e] < 0 || g->edge_i[e] >= N || g->edge_j[e] < 0 || g->edge_j[e] >= N) return LLD_ERR_INVALID;
  out->chi2 = 0; out->lm_iterations = 0; out->lm_trials = 0; out->pcg_iterations = 0; out->solver_used = 0;
  if (N) std::memcpy(out->sim3, g->sim3, sizeof(double) * 8 * (size_t)N);
  // unknowns in vertex order (g2o: buildIndexMapping over the vertices sorted by id), vertex CSR in edge order
  std::vector<int> hidx(N, -1), uvert;
  for (int v = 0; v < N; v++) if (!(g->fixed && g->fixed[v])) { hidx[v] = (int)uvert.size(); uvert.push_back(v); }
  const int nu = (int)uvert.size();
  if (E == 0 || nu == 0) return LLD_OK;                      // optimize() has nothing to do
  std::vector<int> vstart(nu + 1, 0), vinc;
  for (int e = 0; e < E; e++) { if (hidx[g->edge_i[e]] >= 0) vstart[hidx[g->edge_i[e]] + 1]++; if (hidx[g->edge_j[e]] >= 0) vstart[hidx[g->edge_j[e]] + 1]++; }
  for (int u = 0; u < nu; u++) vstart[u + 1] += vstart[u];
  vinc.resize(vstart[nu]);
  { std::vector<int> cur(vstart.begin(), vstart.end() - 1);
    for (int e = 0; e < E; e++) { const int a = hidx[g->edge_i[e]], b = hidx[g->edge_j[e]]; if (a >= 0) vinc[cur[a]++] = e * 2; if (b >= 0) vinc[cur[b]++] = e * 2 + 1; } }
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  const size_t n = 7 * (size_t)nu;
  // one scratch slab
  size_t off = 0; auto take = [&](size_t bytes) { const size_t o = off; off += pad256(bytes); return o; };
  const size_t o_V = take((size_t)N * 64), o_Vbk = take((size_t)N * 64), o_h = take((size_t)N * 4), o_ei = take((size_t)E * 4), o_ej = take((size_t)E * 4), o_C = take((size_t)E * 64),
               o_err = take((size_t)E * 56), o_J = take((size_t)E * 2 * 49 * 8), o_vs = take((size_t)(nu + 1) * 4), o_vi = take(vinc.size() * 4 + 4), o_uv = take((size_t)nu * 4),
               o_Hd = take((size_t)nu * 49 * 8), o_Mi = take((size_t)nu * 49 * 8), o_vec = take(6 * n * 8 + 256), o_sc = take(64);
  // dense Cholesky unless the system is beyond kPgMaxTiles tile columns (or the caller asks for the PCG)
  const int NT = (int)((n + 15) / 16);
  if (prm.solver == 1 && NT > kPgMaxTiles) return LLD_ERR_UNSUPPORTED;
  const bool dense = prm.solver == 1 || (prm.solver == 0 && NT <= kPgMaxTiles);
  const size_t dense_bytes = dense ? (size_t)(NT + 1) * 16 * ((size_t)NT * 16) * 8 : 0;
  // symbolic factorisation on 16x16 tiles: the pattern of column J is merged into the column of its first sub-diagonal row
  std::vector<int> col_start, col_rows, row_start, row_cols;
  if (dense) {
    std::vector<std::vector<int>> col(NT), row(NT);
    auto mark = [&](int ur, int uc) {                                 // unknowns ur >= uc: every tile their 7x7 block touches
      for (int tr = (7 * ur) / 16; tr <= (7 * ur + 6) / 16; tr++) for (int tc = (7 * uc) / 16; tc <= (7 * uc + 6) / 16; tc++) if (tr > tc) col[tc].push_back(tr);
    };
    for (int u = 0; u < nu; u++) mark(u, u);
    for (int e = 0; e < E; e++) { const int a = hidx[g->edge_i[e]], b = hidx[g->edge_j[e]]; if (a >= 0 && b >= 0 && a != b) mark(std::max(a, b), std::min(a, b)); }
    for (int J = 0; J < NT; J++) {
      std::vector<int>& c = col[J];
      std::sort(c.begin(), c.end()); c.erase(std::unique(c.begin(), c.end()), c.end());
      if (c.size() > 1) col[c[0]].insert(col[c[0]].end(), c.begin() + 1, c.end());
      for (int I : c) row[I].push_back(J);
    }
    col_start.assign(1, 0); row_start.assign(1, 0);
    for (int J = 0; J < NT; J++) { col_rows.insert(col_rows.end(), col[J].begin(), col[J].end()); col_start.push_back((int)col_rows.size()); }
    for (int J = 0; J < NT; J++) { row_cols.insert(row_cols.end(), row[J].begin(), row[J].end()); row_start.push_back((int)row_cols.size()); }
  }
  const size_t o_M = take(dense_bytes), o_Li = take(dense ? (size_t)NT * 256 * 8 : 0);
  const size_t o_cs = take(col_start.size() * 4), o_cr = take(col_rows.size() * 4 + 4), o_rs = take(row_start.size() * 4), o_rc = take(row_cols.size() * 4 + 4);
  out->solver_used = dense ? 1 : 2;
  void* db; int st = lld_ctx_scratch(ctx, off + 256, &db); if (st) return st;
  char* d = (char*)db;
  hipStream_t sm = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d + o_V, g->sim3, (size_t)N * 64, hipMemcpyHostToDevice, sm));
  LLD_HIP_TRY(hipMemcpyAsync(d + o_h, hidx.data(), (size_t)N * 4, hipMemcpyHostToDevice, sm));
  LLD_HIP_TRY(hipMemcpyAsync(d + o_ei, g->edge_i, (size_t)E * 4, hipMemcpyHostToDevice, sm));
  LLD_HIP_TRY(hipMemcpyAsync(d + o_ej, g->edge_j, (size_t)E * 4, hipMemcpyHostToDevice, sm));
  LLD_HIP_TRY(hipMemcpyAsync(d + o_C, g->edge_sji, (size_t)E * 64, hipMemcpyHostToDevice, sm));
  LLD_HIP_TRY(hipMemcpyAsync(d + o_vs, vstart.data(), (size_t)(nu + 1) * 4, hipMemcpyHostToDevice, sm));
  LLD_HIP_TRY(hipMemcpyAsync(d + o_vi, vinc.data(), vinc.size() * 4, hipMemcpyHostToDevice, sm));
  LLD_HIP_TRY(hipMemcpyAsync(d + o_uv, uvert.data(), (size_t)nu * 4, hipMemcpyHostToDevice, sm));
  LLD_HIP_TRY(hipMemsetAsync(d + o_J, 0, (size_t)E * 2 * 49 * 8, sm));
  PgArrays A; std::memset(&A, 0, sizeof A);
  A.N = N; A.E = E; A.nu = nu; A.V = reinterpret_cast<double*>(d + o_V); A.hidx = reinterpret_cast<const int*>(d + o_h);
  A.ei = reinterpret_cast<const int*>(d + o_ei); A.ej = reinterpret_cast<const int*>(d + o_ej); A.C = reinterpret_cast<const double*>(d + o_C);
  A.err = reinterpret_cast<double*>(d + o_err); A.J = reinterpret_cast<double*>(d + o_J);
  A.vstart = reinterpret_cast<const int*>(d + o_vs); A.vinc = reinterpret_cast<const int*>(d + o_vi); A.uvert = reinterpret_cast<const int*>(d + o_uv);
  A.Hd = reinterpret_cast<double*>(d + o_Hd); A.Mi = reinterpret_cast<double*>(d + o_Mi);
  double* vec = reinterpret_cast<double*>(d + o_vec);
  A.b = vec; A.x = vec + n; A.r = vec + 2 * n; A.z = vec + 3 * n; A.p = vec + 4 * n; A.ap = vec + 5 * n;
  A.sc = reinterpret_cast<double*>(d + o_sc); A.fix_scale = prm.fix_scale;
  double* dVbk = reinterpret_cast<double*>(d + o_Vbk);
  if (dense) {
    LLD_HIP_TRY(hipMemcpyAsync(d + o_cs, col_start.data(), col_start.size() * 4, hipMemcpyHostToDevice, sm));
    if (!col_rows.empty()) LLD_HIP_TRY(hipMemcpyAsync(d + o_cr, col_rows.data(), col_rows.size() * 4, hipMemcpyHostToDevice, sm));
    LLD_HIP_TRY(hipMemcpyAsync(d + o_rs, row_start.data(), row_start.size() * 4, hipMemcpyHostToDevice, sm));
    if (!row_cols.empty()) LLD_HIP_TRY(hipMemcpyAsync(d + o_rc, row_cols.data(), row_cols.size() * 4, hipMemcpyHostToDevice, sm));
  }
  PgDense D; D.col_start = reinterpret_cast<const int*>(d + o_cs); D.col_rows = reinterpret_cast<const int*>(d + o_cr);
  D.row_start = reinterpret_cast<const int*>(d + o_rs); D.row_cols = reinterpret_cast<const int*>(d + o_rc);
  D.M = reinterpret_cast<double*>(d + o_M); D.Linv = reinterpret_cast<double*>(d + o_Li); D.NT = NT; D.n = (int)n; D.ld = (size_t)NT * 16;
  double hsc[8];
  auto read_sc = [&]() -> int { LLD_HIP_TRY(hipMemcpyAsync(hsc, A.sc, sizeof hsc, hipMemcpyDeviceToHost, sm)); LLD_HIP_TRY(hipStreamSynchronize(sm)); return LLD_OK; };
  auto errors = [&](double* chi) -> int { hipLaunchKernelGGL(pg_errors_kernel, dim3(1), dim3(1024), 0, sm, A); int s = read_sc(); if (s) return s; *chi = hsc[5]; return LLD_OK; };
  const int pcg_limit = prm.pcg_max_iter > 0 ? prm.pcg_max_iter : 10 * (int)n;

  // ---- SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg::solve (lambda from setUserLambdaInit)
  double lambda = -1.0, ni = 2.0; int nBad = 0;
  bool ok = true;
  for (int it = 0; it < prm.iterations && ok; it++) {
    double currentChi; st = errors(&currentChi); if (st) return st;
    const double iniChi = currentChi;
    hipLaunchKernelGGL(pg_linearize_kernel, dim3((unsigned)(((long long)E * 14 + kPgThreads - 1) / kPgThreads)), dim3(kPgThreads), 0, sm, A);
    hipLaunchKernelGGL(pg_diag_kernel, dim3(nu), dim3(64), 0, sm, A);
    if (it == 0) { lambda = prm.lambda_init > 0 ? prm.lambda_init : 1e-5; ni = 2.0; nBad = 0; }   // computeLambdaInit with a user lambda
    double rho = 0.0; int q = 0;
    do {
      LLD_HIP_TRY(hipMemcpyAsync(dVbk, A.V, (size_t)N * 64, hipMemcpyDeviceToDevice, sm));          // push
      if (dense) {
        LLD_HIP_TRY(hipMemsetAsync(D.M, 0, dense_bytes, sm));
        const long long fill = (long long)E * 49 + (long long)nu * 49 + 16ll * NT;
        hipLaunchKernelGGL(pg_dense_fill_kernel, dim3((unsigned)((fill + kPgThreads - 1) / kPgThreads)), dim3(kPgThreads), 0, sm, A, D, lambda);
        for (int J = 0; J < NT; J++) hipLaunchKernelGGL(pg_chol_col_kernel, dim3(2 + col_start[J + 1] - col_start[J]), dim3(256), 0, sm, D, J, A.sc);
        for (int J = NT - 1; J >= 0; J--) hipLaunchKernelGGL(pg_chol_back_kernel, dim3((1 + row_start[J + 1] - row_start[J] + 3) / 4), dim3(256), 0, sm, D, J, A.x);
      } else {
        hipLaunchKernelGGL(pg_pcg_init_kernel, dim3(1), dim3(1024), 0, sm, A, lambda, prm.pcg_rel_tol);
        for (int k = 0; k < pcg_limit;) {
          for (int c = 0; c < 16 && k < pcg_limit; c++, k++) {
            hipLaunchKernelGGL(pg_matvec_kernel, dim3((nu * 8 + kPgThreads - 1) / kPgThreads), dim3(kPgThreads), 0, sm, A, lambda);
            hipLaunchKernelGGL(pg_pcg_update_kernel, dim3(1), dim3(1024), 0, sm, A, pcg_limit);
          }
          st = read_sc(); if (st) return st;
          if (hsc[3] != 0.0) break;
        }
      }
      st = read_sc(); if (st) return st;
      const bool ok2 = hsc[4] != 0.0;
      out->pcg_iterations += (int)hsc[2];
      hipLaunchKernelGGL(pg_update_kernel, dim3(1), dim3(1024), 0, sm, A, lambda);
      double tempChi; st = errors(&tempChi); if (st) return st;                                      // also brings sc[6] = scale
      double scale = hsc[6];
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2; currentChi = tempChi;                                                                // discardTop
      } else {
        lambda *= ni; ni *= 2;
        LLD_HIP_TRY(hipMemcpyAsync(A.V, dVbk, (size_t)N * 64, hipMemcpyDeviceToDevice, sm));        // pop
      }
      q++; out->lm_trials++;
    } while (rho < 0 && q < prm.max_trials);
    out->chi2 = currentChi;
    out->lm_iterations++;
    if (q == prm.max_trials || rho == 0) ok = false;
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) ok = false;
    }
  }
  LLD_HIP_TRY(hipMemcpyAsync(out->sim3, A.V, (size_t)N * 64, hipMemcpyDeviceToHost, sm));
  LLD_HIP_TRY(hipStreamSynchronize(sm));
  return LLD_OK;
}
