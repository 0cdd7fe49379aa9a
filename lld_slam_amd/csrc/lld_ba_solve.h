// lld_ba_solve.h - Reduced camera system of the batched bundle adjustment: PCG (ba_pcg*, ba_pcgm_*), exact LDL^T / Cholesky (ba_chol_kernel), fp64 matrix-core Cholesky (ba_chol_mfma_kernel), structure-following Cholesky (lld_ba_chol_sparse.h).
// Part of lld_ba_kernels.h (split by kernel family in round 6; no behaviour change): included from there INSIDE namespace lldba, after the shared types and helpers.
// Not a stand-alone header.

// ================================================================== PCG on the reduced camera system
// grid (nW); block kPcgThreads; dynamic LDS: 4n + kPcgThreads + nf*36 + 32 doubles.  Block-Jacobi preconditioner (inverse 6x6 diagonal
// blocks), fixed reduction trees, stops at |r|_M <= tol |b|_M.  On exit it applies VertexSE3Expmap::oplusImpl to the free
// cameras (trial buffer) and leaves sum x(lambda x + b) of the camera part for computeScale.
// ------------------------------------------------------------------ PCG across the whole GPU (few, larger windows)
// The same block-Jacobi PCG as ba_pcg_kernel, cut into kernels so that the matrix-vector product of ONE window runs on every CU:
//   init    (1 workgroup per window)  Minv, x = 0, r = b, z = Minv r, p = z, rz
//   matvec  (wavefront per row)       Sp = S p                                   -- S symmetric, stored in full
//   update  (1 workgroup per window)  alpha, x, r, z, rz, stop test, beta, p
//   final   (1 workgroup per window)  the common epilogue (x -> xp, scale, trial cameras)
// The host launches matvec/update pairs in chunks and looks at the `done` scalars between chunks; finished windows return at once.
__global__ __launch_bounds__(kPcgThreads) void ba_pcgm_init_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st, double tol) {
  __shared__ double scratch[32];
  const BAWin W = wins[blockIdx.x];
  const BAState& S = st[blockIdx.x];
  double* sc = A.pcg_sc + 8 * (size_t)W.win_index;
  if (S.phase != PH_RUN) { if (threadIdx.x == 0) sc[3] = 1.0; return; }
  const int nf = W.n_free, n = 6 * nf, tid = threadIdx.x;
  const double* Sg = A.S + W.S_off;
  double* Mi = A.pcg_mi + (size_t)W.hpp_off * 36;
  double* x = A.xp + W.x_off;
  double* r = A.pcg_vec + W.x_off; double* z = r + A.x_total; double* p = z + A.x_total;
  double* ok_s = scratch + 31;
  if (tid == 0) *ok_s = 1.0;
  __syncthreads();
  for (int cb = tid; cb < nf; cb += kPcgThreads) {
    double F[36], Fi[36];
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
#pragma unroll
      for (int c = 0; c < 6; c++) F[rr * 6 + c] = Sg[(size_t)(cb * 6 + rr) * n + cb * 6 + c];
    if (!spd_inverse<6>(F, Fi)) *ok_s = 0.0;
#pragma unroll
    for (int i = 0; i < 36; i++) Mi[cb * 36 + i] = Fi[i];
  }
  for (int i = tid; i < n; i += kPcgThreads) { x[i] = 0.0; r[i] = A.bschur[W.x_off + i]; }
  __syncthreads();                                       // Mi and r of this workgroup are visible to it
  double part = 0.0;
  for (int i = tid; i < n; i += kPcgThreads) {
    const int b = i / 6, rr = i - b * 6;
    double zv = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) zv += Mi[b * 36 + rr * 6 + c] * r[b * 6 + c];
    z[i] = zv; p[i] = zv;
    part += r[i] * zv;
  }
  const double rz0 = block_sum(part, scratch);
  if (tid == 0) {
    const bool ok = *ok_s != 0.0 && isfinite(rz0);
    sc[0] = rz0; sc[1] = tol * tol * rz0; sc[2] = 0.0; sc[3] = (ok && rz0 > 0.0) ? 0.0 : 1.0; sc[4] = ok ? 1.0 : 0.0;
  }
}

// grid (ceil(n_max / 4), nW), block 256: one wavefront per row of S
__global__ __launch_bounds__(256) void ba_pcgm_matvec_kernel(BAArrays A, const BAWin* __restrict__ wins) {
  const BAWin W = wins[blockIdx.y];
  const double* sc = A.pcg_sc + 8 * (size_t)W.win_index;
  if (sc[3] != 0.0) return;
  const int n = 6 * W.n_free, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
  if (row >= n) return;
  const double* Sr = A.S + W.S_off + (size_t)row * n;
  const double* p = A.pcg_vec + 2 * A.x_total + W.x_off;
  double acc = 0.0;
  for (int c = lane; c < n; c += 64) acc = fma(Sr[c], p[c], acc);
  acc = wave_sum(acc);
  if (lane == 0) A.pcg_vec[3 * A.x_total + W.x_off + row] = acc;
}

__global__ __launch_bounds__(kPcgThreads) void ba_pcgm_update_kernel(BAArrays A, const BAWin* __restrict__ wins, int max_iter_param) {
  __shared__ double scratch[32];
  const BAWin W = wins[blockIdx.x];
  double* sc = A.pcg_sc + 8 * (size_t)W.win_index;
  if (sc[3] != 0.0) return;
  const int nf = W.n_free, n = 6 * nf, tid = threadIdx.x;
  const double* Mi = A.pcg_mi + (size_t)W.hpp_off * 36;
  double* x = A.xp + W.x_off;
  double* r = A.pcg_vec + W.x_off; double* z = r + A.x_total; double* p = z + A.x_total; const double* ap = p + A.x_total;
  const double rz = sc[0], stop = sc[1];
  const int max_iter = max_iter_param > 0 ? max_iter_param : 10 * n;
  double part = 0.0;
  for (int i = tid; i < n; i += kPcgThreads) part += p[i] * ap[i];
  const double pAp = block_sum(part, scratch);
  if (!(pAp > 0.0) || !isfinite(pAp)) { if (tid == 0) { sc[3] = 1.0; sc[4] = 0.0; } return; }
  const double alpha = rz / pAp;
  for (int i = tid; i < n; i += kPcgThreads) { x[i] += alpha * p[i]; r[i] -= alpha * ap[i]; }
  __syncthreads();
  part = 0.0;
  for (int i = tid; i < n; i += kPcgThreads) {
    const int b = i / 6, rr = i - b * 6;
    double zv = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) zv += Mi[b * 36 + rr * 6 + c] * r[b * 6 + c];
    z[i] = zv;
    part += r[i] * zv;
  }
  const double rz_new = block_sum(part, scratch);
  const double iters = sc[2] + 1.0;
  bool done = false, ok = true;
  if (!isfinite(rz_new)) { done = true; ok = false; }
  else if (rz_new <= stop || iters >= (double)max_iter) done = true;
  if (!done) { const double beta = rz_new / rz; for (int i = tid; i < n; i += kPcgThreads) p[i] = z[i] + beta * p[i]; }
  __syncthreads();                                        // every lane has read sc[] before lane 0 rewrites it
  if (tid == 0) { sc[0] = rz_new; sc[2] = iters; if (done) sc[3] = 1.0; if (!ok) sc[4] = 0.0; }
}

// the epilogue of solve_epilogue for any number of unknowns: scale of the step, trial cameras
__global__ __launch_bounds__(kPcgThreads) void ba_pcgm_final_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  __shared__ double scratch[32];
  const BAWin W = wins[blockIdx.x];
  BAState& S = st[blockIdx.x];
  if (S.phase != PH_RUN) return;
  const double* sc = A.pcg_sc + 8 * (size_t)W.win_index;
  const int tid = threadIdx.x, nf = W.n_free, n = 6 * nf;
  const double lambda = S.lambda;
  const double* x = A.xp + W.x_off;
  const double* bpv = A.bp + (size_t)W.hpp_off * 6;
  double part = 0.0;
  for (int i = tid; i < n; i += kPcgThreads) part += x[i] * (lambda * x[i] + bpv[i]);
  const double sc_t = block_sum(part, scratch);
  const int cur = S.cur, nxt = cur ^ 1;
  for (int c = tid; c < W.n_cams; c += kPcgThreads) {
    const Pose T = load_cam(A, cur, W.cam_off + c);
    Pose Tn = T;
    if (c < nf) Tn = pose_oplus(T, x + c * 6);
    pose_store(Tn, A.cam_qt + ((size_t)nxt * A.NC + W.cam_off + c) * 7);
  }
  if (tid == 0) { S.scale_cam = sc_t; S.pcg_ok = sc[4] != 0.0 ? 1 : 0; S.pcg_iterations += (int)sc[2]; }
}

__global__ __launch_bounds__(kPcgThreads) void ba_pcg_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st, double tol,
                                                            int max_iter_param) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const BAWin W = wins[blockIdx.x];
  BAState& S = st[blockIdx.x];
  if (S.phase != PH_RUN) return;
  const int nf = W.n_free, n = 6 * nf;
  double* x = lds; double* r = x + n; double* z = r + n; double* p = z + n; double* Ap = p + n;   // Ap: kPcgThreads doubles
  double* Mi = Ap + kPcgThreads;       // nf * 36
  double* scratch = Mi + nf * 36;      // 32
  const double* Sg = A.S + W.S_off;
  const double* bs = A.bschur + W.x_off;
  const int tid = threadIdx.x;
  double* ok_s = scratch + 31;         // keeps every LDS object inside the (16-B aligned) dynamic region
  if (tid == 0) *ok_s = 1.0;
  __syncthreads();
  if (tid < nf) {
    double F[36], Fi[36];
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
#pragma unroll
      for (int c = 0; c < 6; c++) F[rr * 6 + c] = Sg[(size_t)(tid * 6 + rr) * n + tid * 6 + c];
    if (!spd_inverse<6>(F, Fi)) *ok_s = 0.0;
#pragma unroll
    for (int i = 0; i < 36; i++) Mi[tid * 36 + i] = Fi[i];
  }
  if (tid < n) { x[tid] = 0.0; r[tid] = bs[tid]; }
  __syncthreads();
  auto precond = [&]() {               // z = M^-1 r
    if (tid < n) {
      const int b = tid / 6, rr = tid - b * 6;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 6; c++) s += Mi[b * 36 + rr * 6 + c] * r[b * 6 + c];
      z[tid] = s;
    }
  };
  precond();
  __syncthreads();
  if (tid < n) p[tid] = z[tid];
  double rz = block_sum(tid < n ? r[tid] * z[tid] : 0.0, scratch);
  const double rz0 = rz;
  const int max_iter = max_iter_param > 0 ? max_iter_param : 10 * n;
  int iters = 0;
  bool ok = *ok_s != 0.0 && isfinite(rz0);
  // S is symmetric and stored in full, so y = S p is computed column-wise: lane <-> column c, K row slices per column,
  // every load is a coalesced 512-B row segment, independent of its neighbours (deep memory-level parallelism), and no
  // cross-lane reduction is needed: y[c] = sum_k part[k][c] in a fixed order.
  const int K = max(1, min(8, kPcgThreads / max(n, 1)));
  const int rows_per = (n + K - 1) / K;
  const int mv_c = tid % max(n, 1), mv_k = tid / max(n, 1);
  const bool mv_on = n > 0 && mv_k < K;
  const int mv_r0 = mv_k * rows_per, mv_r1 = min(n, mv_r0 + rows_per);
  double* part = Ap;                    // [K][n] partial products live in the Ap..Mi gap: K*n <= kPcgThreads doubles
  if (ok && rz0 > 0.0) {
    const double stop = tol * tol * rz0;
    for (; iters < max_iter;) {
      if (mv_on) {
        double acc = 0.0;
        const double* Sc = Sg + mv_c;
#pragma unroll 4
        for (int row = mv_r0; row < mv_r1; row++) acc += Sc[(size_t)row * n] * p[row];
        part[mv_k * n + mv_c] = acc;
      }
      __syncthreads();
      double ap = 0.0;
      if (tid < n) { for (int k = 0; k < K; k++) ap += part[k * n + tid]; }
      const double pAp = block_sum(tid < n ? p[tid] * ap : 0.0, scratch);
      if (!(pAp > 0.0) || !isfinite(pAp)) { ok = false; break; }
      const double alpha = rz / pAp;
      if (tid < n) { x[tid] += alpha * p[tid]; r[tid] -= alpha * ap; }
      __syncthreads();
      precond();
      __syncthreads();
      const double rz_new = block_sum(tid < n ? r[tid] * z[tid] : 0.0, scratch);
      iters++;
      if (!isfinite(rz_new)) { ok = false; break; }
      if (rz_new <= stop) break;
      const double beta = rz_new / rz;
      rz = rz_new;
      if (tid < n) p[tid] = z[tid] + beta * p[tid];
      __syncthreads();
    }
  }
  __syncthreads();
  solve_epilogue(A, W, S, x, scratch, ok, iters);
}

// ================================================================== exact solve of the reduced camera system
// grid (nW); block kPcgThreads; dynamic LDS: (2*nf*36 + 2*n + 32 + lds_tri_doubles) doubles.
// Right-looking block Cholesky (6x6 camera blocks) in place on the LOWER block triangle of S, the right-hand side carried
// along as an extra block row (forward substitution for free), then block back-substitution.  This is the counterpart of
// the reference's exact factorisation (Eigen::SimplicialLDLT, solvers/linear_solver_eigen.h:94-124): S is read once from
// HBM instead of once per PCG iteration, and the result does not depend on an iteration tolerance.  A non-positive pivot
// reports failure, which Levenberg–Marquardt turns into a rejected trial (optimization_algorithm_levenberg.cpp:126-127).
__global__ __launch_bounds__(kPcgThreads) void ba_chol_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st, int lds_tri_doubles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.x);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  const int nf = W.n_free, n = 6 * nf;
  double* linv = lds;                    // [nf][36] inverse of the diagonal Cholesky blocks (lower)
  double* panel = linv + nf * 36;        // [nf][36] current block column of L
  double* y = panel + nf * 36;           // [n] right-hand side -> forward solution
  double* x = y + n;                     // [n] solution
  double* scratch = x + n;               // [32]
  double* okf = scratch + 31;
  double* tri = scratch + 32;            // LDS-resident trailing block triangle (cameras >= m0), 36 doubles per block
  double* Sg = A.S + W.S_off;
  const int tid = threadIdx.x;
  // largest trailing triangle that fits: blocks (i, j), i >= j >= m0, live in LDS for the whole factorisation, so their
  // read-modify-write updates never wait for HBM/L2; only the first m0 block columns are updated in global memory
  int mt = 0;
  while (mt < nf && (mt + 1) * (mt + 2) / 2 * 36 <= lds_tri_doubles) mt++;
  const int m0 = nf - mt;
  auto tri_blk = [&](int i, int j) { const int ii = i - m0, jj = j - m0; return tri + (size_t)(ii * (ii + 1) / 2 + jj) * 36; };
  if (tid == 0) *okf = 1.0;
  if (tid < n) y[tid] = A.bschur[W.x_off + tid];
  for (int t = tid; t < mt * (mt + 1) / 2 * 6; t += kPcgThreads) {        // lane <-> one row of one block
    const int blk = t / 6, r = t - blk * 6;
    int ii = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= blk) ii++;
    while (ii * (ii + 1) / 2 > blk) ii--;
    const int jj = blk - ii * (ii + 1) / 2;
    const double* src = Sg + (size_t)(6 * (m0 + ii) + r) * n + 6 * (m0 + jj);
    double* dst = tri + (size_t)blk * 36 + r * 6;
#pragma unroll
    for (int c = 0; c < 6; c++) dst[c] = src[c];
  }
  __syncthreads();
  for (int k = 0; k < nf; k++) {
    const bool k_lds = k >= m0;
    // (1) diagonal block: L_kk = chol(A_kk), Linv_kk, y_k = Linv_kk b_k
    if (tid == 0) {
      double a[6][6], L[6][6], Li[6][6];
      if (k_lds) { const double* d = tri_blk(k, k); for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) a[r][c] = d[r * 6 + c]; }
      else for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) a[r][c] = Sg[(size_t)(6 * k + r) * n + 6 * k + c];
      bool ok = true;
      for (int j = 0; j < 6; j++) {
        double d = a[j][j];
        for (int m = 0; m < j; m++) d -= L[j][m] * L[j][m];
        if (!(d > 0.0) || !isfinite(d)) ok = false;
        const double ljj = sqrt(d), inv = 1.0 / ljj;
        L[j][j] = ljj;
        for (int i = j + 1; i < 6; i++) {
          double sacc = a[i][j];
          for (int m = 0; m < j; m++) sacc -= L[i][m] * L[j][m];
          L[i][j] = sacc * inv;
        }
      }
      for (int j = 0; j < 6; j++) {
        Li[j][j] = 1.0 / L[j][j];
        for (int i = j + 1; i < 6; i++) {
          double sacc = 0.0;
          for (int m = j; m < i; m++) sacc -= L[i][m] * Li[m][j];
          Li[i][j] = sacc / L[i][i];
        }
      }
      for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) linv[k * 36 + r * 6 + c] = c <= r ? Li[r][c] : 0.0;
      double yk[6];
      for (int r = 0; r < 6; r++) { double sacc = 0.0; for (int c = 0; c <= r; c++) sacc += Li[r][c] * y[6 * k + c]; yk[r] = sacc; }
      for (int r = 0; r < 6; r++) y[6 * k + r] = yk[r];
      if (!ok) *okf = 0.0;
    }
    __syncthreads();
    // (2) panel: L_ik = A_ik Linv_kk^T for i > k (lane <-> one row of one block), b_i -= L_ik y_k.  The back-substitution
    //     reads L from global memory, so the panel is stored there as well (plain stores, nothing waits for them).
    const int m_rows = (nf - k - 1) * 6;
    for (int t = tid; t < m_rows; t += kPcgThreads) {
      const int i = k + 1 + t / 6, r = t % 6;
      double* grow = Sg + (size_t)(6 * i + r) * n + 6 * k;
      double arow[6], lrow[6];
      if (k_lds) { const double* d = tri_blk(i, k) + r * 6;
#pragma unroll
        for (int c = 0; c < 6; c++) arow[c] = d[c];
      } else {
#pragma unroll
        for (int c = 0; c < 6; c++) arow[c] = grow[c];
      }
      const double* Li = linv + k * 36;
      double dotv = 0.0;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double sacc = 0.0;
#pragma unroll
        for (int m = 0; m <= c; m++) sacc += arow[m] * Li[c * 6 + m];
        lrow[c] = sacc;
        dotv += sacc * y[6 * k + c];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) { grow[c] = lrow[c]; panel[i * 36 + r * 6 + c] = lrow[c]; }
      y[6 * i + r] -= dotv;
    }
    __syncthreads();
    // (3) trailing update: A_ij -= L_ik L_jk^T for k < j <= i (lane <-> block; lower triangle only)
    const int m = nf - k - 1;
    const int nblk = m * (m + 1) / 2;
    for (int t = tid; t < nblk; t += kPcgThreads) {
      int ii = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
      while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
      while (ii * (ii + 1) / 2 > t) ii--;
      const int jj = t - ii * (ii + 1) / 2;
      const int i = k + 1 + ii, j = k + 1 + jj;
      const double* Pi = panel + i * 36;
      const double* Pj = panel + j * 36;
      double pj[36];
#pragma unroll
      for (int q = 0; q < 36; q++) pj[q] = Pj[q];
      const bool in_lds = j >= m0;
      double* blk = in_lds ? tri_blk(i, j) : nullptr;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        double* row = in_lds ? blk + r * 6 : Sg + (size_t)(6 * i + r) * n + 6 * j;
        double pr[6];
#pragma unroll
        for (int q = 0; q < 6; q++) pr[q] = Pi[r * 6 + q];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          double sacc = 0.0;
#pragma unroll
          for (int q = 0; q < 6; q++) sacc += pr[q] * pj[c * 6 + q];
          row[c] -= sacc;
        }
      }
    }
    __syncthreads();
  }
  // back substitution: L^T x = y
  for (int k = nf - 1; k >= 0; k--) {
    if (tid < 6) {
      const double* Li = linv + k * 36;
      double sacc = 0.0;
      for (int mm = tid; mm < 6; mm++) sacc += Li[mm * 6 + tid] * y[6 * k + mm];      // x_k = Linv_kk^T y_k
      x[6 * k + tid] = sacc;
    }
    __syncthreads();
    for (int t = tid; t < 6 * k; t += kPcgThreads) {                                  // y_j -= L_kj^T x_k, j < k (coalesced along the row)
      double sacc = 0.0;
#pragma unroll
      for (int r = 0; r < 6; r++) sacc += Sg[(size_t)(6 * k + r) * n + t] * x[6 * k + r];
      y[t] -= sacc;
    }
    __syncthreads();
  }
  const bool ok = *okf != 0.0;
  solve_epilogue(A, W, S, x, scratch, ok, 0);
}

// ================================================================== exact solve on the fp64 matrix cores
// grid (nW); block kCholMThreads (8 wavefronts); dynamic LDS kCholMLdsDoubles doubles.  For n = 6*n_free <= 304.
// Right-looking Cholesky on 16x16 tiles with the WHOLE lower tile triangle held in registers for the entire factorisation
// (<= 190 tiles, 28 per wavefront, 4 doubles per lane each): S is read from HBM exactly once, L never leaves the chip.
// Wavefronts 1..7 own the tiles; wavefront 0 (the panel wave) owns no tile and does the serial work, so its 16-double row
// buffers never compete with the accumulator tiles for registers.  Per tile column J (two barriers):
//   (c) tile waves: L_IJ = A_IJ L_JJ^-T as four v_mfma_f64_16x16x4_f64 per tile of column J, result kept in the registers (it is
//       the L the back substitution needs) and written to the LDS panel buffer as the operand of (d);
//   (d) tile waves: every tile (I,K), K > J, takes T -= L_IJ L_KJ^T on the matrix cores — operands are read once per 1024 FMAs
//       instead of once per 1.5 as in the 6x6 register-blocked kernel above, whose trailing update is LDS-bandwidth bound;
//       tiles of column J+1 are then final and go to the other panel buffer, the diagonal tile J+2 to its slot;
//       LOOKAHEAD, same phase: the panel wave forward-substitutes the right-hand side, applies column J's update to the
//       diagonal tile J+1 itself (four MFMAs on the published copy) and factors it right-looking with one lane per row — lane 16
//       carries the right-hand side and lanes 17..32 the identity as extra rows, which yields y_{J+1} and L^-1 from the same
//       recurrence (v_readlane broadcasts, no LDS traffic inside it).  The serial factorisation is off the critical path.
// Tile element layout of v_mfma_f64_16x16x4_f64: C/D lane l, register g -> (row (l>>4) + 4g, col l&15); A[i][k] and B[k][j]
// come from lane i + 16k resp. j + 16k.
constexpr int kCholMThreads = 512;
constexpr int kCholMTileWaves = kCholMThreads / 64 - 1;
constexpr int kCholMMaxTiles = 19;                                   // 19 * 16 = 304 >= 6 * 50
constexpr int kCholMSlots = 28;                                      // ceil(190 / 7)
constexpr int kCholMStride = 17;                                     // padded LDS row of 16 doubles
constexpr int kCholMN = kCholMMaxTiles * 16;
constexpr int kCholMLdsDoubles = 2 * kCholMN * kCholMStride + kCholMMaxTiles * 16 * kCholMStride + 16 * kCholMStride + 3 * kCholMN + 32;
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Panel wave: factor the 16x16 tile in Dg (lower triangle used); L^-1 -> Li (operand of the column's L_IJ = A_IJ L_JJ^-T) and -> Dg
// (kept for the back substitution, which only needs the inverse); rhs y[0..15] -> L^-1 y.
__device__ __forceinline__ bool chol_tile_factor(double* Dg, double* Li, double* y, int lane) {
  const int r = lane < 32 ? lane : 32;                               // 0..15 tile rows, 16 rhs, 17..32 identity rows
  // one load path for all lanes: tile rows and the right-hand side are read through a per-lane pointer, the identity rows read
  // the (finite) right-hand side too and are overwritten
  const double* src = (r < 16) ? Dg + r * kCholMStride : y;
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { const double v = src[c]; a[c] = (r <= 16) ? v : (r - 17 == c ? 1.0 : 0.0); }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const double d = readlane_f64(a[c], c);
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    // 1/sqrt(d): v_rsq_f64 seed + two Newton steps (the library sqrt and divide are ~45 dependent instructions, which is what
    // bounds this serial recurrence; the result is within an ulp or two, L L^T = S to rounding either way)
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * (1.5 - (0.5 * d) * (inv * inv));
    inv = inv * (1.5 - (0.5 * d) * (inv * inv));
    const double lc = a[c] * inv;                                    // lane c: sqrt(d); below: L[r][c]; rhs lane: y_c
    a[c] = lc;
#pragma unroll
    for (int c2 = c + 1; c2 < 16; c2++) a[c2] -= lc * readlane_f64(lc, c2);   // A[r][c2] -= L[r][c] L[c2][c]
  }
  if (lane == 16) {
#pragma unroll
    for (int c = 0; c < 16; c++) y[c] = a[c];
  } else if (lane > 16 && lane <= 32) {                              // lane 17+k holds column k of L^-1
#pragma unroll
    for (int c = 0; c < 16; c++) { Li[c * kCholMStride + (lane - 17)] = a[c]; Dg[c * kCholMStride + (lane - 17)] = a[c]; }
  }
  return ok;
}

__global__ __launch_bounds__(kCholMThreads) void ba_chol_mfma_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.x);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  if (A.chol_plan && A.chol_plan[W.win_index].mode == 1) return;      // this window is ba_chol_sparse_kernel's (launched next to this one when a group holds both kinds)
  const int nf = W.n_free, n = 6 * nf, NT = (n + 15) >> 4, N = NT << 4;
  double* Lp0 = lds;                                       // [2][N][17] panel buffers: column J in buffer J & 1 (raw, then L)
  double* Dall = Lp0 + 2 * kCholMN * kCholMStride;         // [NT][16][17] diagonal tiles: raw until factored, then L_JJ
  double* Li = Dall + kCholMMaxTiles * 16 * kCholMStride;  // [16][17] inverse of the current diagonal factor
  double* colsum = Li + 16 * kCholMStride;                 // [7][16] per tile wave: column sums of the back substitution (room for N)
  double* y = colsum + kCholMN;                            // [N] right-hand side -> forward solution
  double* x = y + kCholMN;                                 // [N] solution
  double* scratch = x + kCholMN;                           // [32]
  double* okf = scratch + 31;
  const double* Sg = A.S + W.S_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 4, lcol = lane & 15;
#ifdef LLD_EXPERIMENTS
  long long* stamp_base = A.chol_stamps ? A.chol_stamps + ((size_t)W.win_index * kCholStampWaves + wave) * kCholStampSlots : nullptr;
#endif
  LLD_CHOL_STAMP(0);
  if (tid < N) y[tid] = (tid < n) ? A.bschur[W.x_off + tid] : 0.0;
  if (tid == 0) *okf = 1.0;

  if (wave == 0) {
    // ================================================================ panel wave
    // The diagonal tile 0 does not wait for the tile wavefronts (round 4): this wavefront fetches its 256 values itself and factors it while
    // the other seven still load their 28 tiles each - the prologue's publish of that tile and its 2.3 us factorisation leave the chain.
    if (NT > 0) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int idx = lane + 64 * q, row = idx >> 4, col = idx & 15;
        const bool inside = row < n && col < n, lower = inside && col <= row;
        const double v = Sg[lower ? row * n + col : 0];
        Dall[row * kCholMStride + col] = lower ? v : ((!inside && row == col) ? 1.0 : 0.0);
      }
      if (!chol_tile_factor(Dall, Li, y, lane) && lane == 0) *okf = 0.0;      // (y[0..15] and *okf were written by this wavefront's own lanes above)
    }
    __syncthreads();                                                   // tiles loaded, y staged
    LLD_CHOL_STAMP(1);
    __syncthreads();                                                   // prologue publish done: column 0, diagonal tile 1
    LLD_CHOL_STAMP(2);
    LLD_CHOL_STAMP(3);
    __syncthreads();                                                   // (diagonal tile 0 factored: long since)
    for (int J = 0; J < NT; J++) {
      const double* Lp = Lp0 + (J & 1) * kCholMN * kCholMStride;
      LLD_CHOL_STAMP(8 + 6 * J);
      __syncthreads();                                                 // (c) done: Lp holds L(:,J)
      LLD_CHOL_STAMP(9 + 6 * J);
      if (lane < 16 && J + 1 < NT) {                                   // y_(J+1) -= L_(J+1)J y_J: the sixteen rows the next tile factor carries along;
        const double* pr = Lp + (16 * (J + 1) + lane) * kCholMStride;  // the rows below are the tile wavefronts' (they wait for this wavefront
        double dotv = 0.0;                                             // in the late columns: round 4, 10.9 us off its path)
#pragma unroll
        for (int c = 0; c < 16; c++) dotv += pr[c] * y[16 * J + c];
        y[16 * (J + 1) + lane] -= dotv;
      }
      LLD_CHOL_STAMP(10 + 6 * J);
      if (J + 1 < NT) {
        // lookahead: diagonal tile J+1 (published with the updates of columns < J) takes column J's update here, then is factored
        double* Dg = Dall + (J + 1) * 16 * kCholMStride;
        const double* pa = Lp + (16 * (J + 1) + lcol) * kCholMStride + lrow;
        v4d c;
#pragma unroll
        for (int g = 0; g < 4; g++) c[g] = Dg[(lrow + 4 * g) * kCholMStride + lcol];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk], pa[4 * kk], c, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++) Dg[(lrow + 4 * g) * kCholMStride + lcol] = c[g];
        LLD_CHOL_STAMP(11 + 6 * J);
        if (!chol_tile_factor(Dg, Li, y + 16 * (J + 1), lane) && lane == 0) *okf = 0.0;
      }
      LLD_CHOL_STAMP(12 + 6 * J);
      __syncthreads();                                                 // (d) + lookahead done
      LLD_CHOL_STAMP(13 + 6 * J);
    }
    LLD_CHOL_STAMP(4);
    // back substitution L^T x = y: x_J = L_JJ^-T (y_J - s_J), s_J = the tile waves' column sums of L_IJ^T x_I (I > J); two barriers per tile
    for (int J = NT - 1; J >= 0; J--) {
      __syncthreads();                                                 // column sums of J complete
      const double* Di = Dall + J * 16 * kCholMStride;                 // L_JJ^-1
      const int c = lane & 15, part = lane >> 4;
      double xc = 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 4 * part + q;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kCholMTileWaves; w++) sum += colsum[w * 16 + r];
        xc += Di[r * kCholMStride + c] * (y[16 * J + r] - sum);
      }
      xc += __shfl_xor(xc, 16); xc += __shfl_xor(xc, 32);
      if (lane < 16) x[16 * J + c] = xc;
      __syncthreads();                                                 // x_J ready
    }
    LLD_CHOL_STAMP(5);
  } else {
    // ================================================================ tile waves
    // tile coordinates of this wavefront's slots (wave-uniform, integer-only: scalar registers).  Tile (I, K) belongs to tile wave
    // (I + 2K) mod 7: the tiles of one COLUMN (I consecutive) and of one row spread evenly over the seven waves, so the
    // L_IJ = A_IJ L_JJ^-T phase of a column is at most ceil(rows / 7) tiles deep (a round-robin over the packed index
    // I (I + 1) / 2 + K puts a column on four of the seven waves only); <= 28 tiles per wave for 19 tile rows.
    int tI[kCholMSlots], tK[kCholMSlots];
    {
      const int w0 = wave - 1;
      int I = 0, K = (4 * w0) % 7;                                     // in row I: K = 4 (w0 - I) mod 7 (4 = 2^-1 mod 7), then every 7th column
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        while (I < NT && K > I) { I++; K = (4 * (((w0 - I) % 7) + 7)) % 7; }
        const bool valid = I < NT;
        tI[sl] = valid ? I : -1;
        tK[sl] = valid ? K : -1;
        K += 7;
      }
    }
    // S -> registers (lower triangle; the padding rows/columns carry an identity so that L is the identity there).  All 28 tiles
    // (112 loads per lane, the hardware queues what it cannot keep in flight) go out before the first value is touched: one slot at a time, the 28 slots were 28 dependent round
    // trips to another XCD's L2 (18 us of a 160 us kernel).  Tile base in scalar registers, four per-lane offsets shared by all slots.
    unsigned* cmask = reinterpret_cast<unsigned*>(colsum + kCholMTileWaves * 16) + (wave - 1) * kCholMMaxTiles;    // per-column slot masks, see below (the tail of the column-sum area: 112 of its 304 doubles are used)
    v4d acc[kCholMSlots];
    int offg[4];
#pragma unroll
    for (int g = 0; g < 4; g++) offg[g] = (lrow + 4 * g) * n + lcol;
    constexpr int kLoadGroup = 28;
#pragma unroll
    for (int s0 = 0; s0 < kCholMSlots; s0 += kLoadGroup) {
#pragma unroll
      for (int sl = s0; sl < s0 + kLoadGroup; sl++) {
        v4d v = {0.0, 0.0, 0.0, 0.0};
        if (tI[sl] >= 0) {
          const double* base = Sg + (16 * tI[sl]) * n + 16 * tK[sl];
          if (tK[sl] < tI[sl] && 16 * tI[sl] + 16 <= n) {               // interior tile (wave-uniform): scalar base + the shared lane offsets
#pragma unroll
            for (int g = 0; g < 4; g++) v[g] = base[offg[g]];
          } else {
            const int col = 16 * tK[sl] + lcol;
#pragma unroll
            for (int g = 0; g < 4; g++) {
              const int row = 16 * tI[sl] + lrow + 4 * g;
              const bool lower = row < n && col < n && col <= row;
              v[g] = base[lower ? offg[g] : 0];
            }
          }
        }
        acc[sl] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s0 == 0) {                                                   // (while the loads are in flight)
        // Per column J the slots of this wavefront's OFF-DIAGONAL tiles of that column, as a bit mask in LDS (round 4).  The L_IJ phase and the back
        // substitution touch at most three tiles per column and wavefront but walked all 28 slots for them, and the tile coordinates live in
        // spilled scalar registers (~35 cycles per slot looked at: the walk, not the barriers, is what a column of the back substitution costs -
        // DESIGN.md section 7): with the mask a slot that is not in the column costs one scalar bit test.
        if (lane < kCholMMaxTiles) cmask[lane] = 0u;
#pragma unroll
        for (int sl = 0; sl < kCholMSlots; sl++)
          if (tI[sl] > tK[sl] && lane == 0) atomicOr(&cmask[tK[sl]], 1u << sl);
        // (the trailing update takes most slots in the columns where it is on the critical path; masks for it measured no gain)
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sl = s0; sl < s0 + kLoadGroup; sl++) {
        if (tI[sl] >= 0 && !(tK[sl] < tI[sl] && 16 * tI[sl] + 16 <= n)) {
          const int col = 16 * tK[sl] + lcol;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int row = 16 * tI[sl] + lrow + 4 * g;
            const bool inside = row < n && col < n, lower = inside && col <= row;
            acc[sl][g] = lower ? acc[sl][g] : ((!inside && row == col) ? 1.0 : 0.0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    LLD_CHOL_STAMP(1);
    __syncthreads();
    {
      // prologue publish: column 0 (raw) -> panel buffer 0, diagonal tile 1 (raw) -> its slot (tile 0 is the panel wavefront's own business)
      int off_cd = lrow * kCholMStride + lcol;
      asm volatile("" : "+v"(off_cd));
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        const bool diag01 = tI[sl] == tK[sl] && tI[sl] == 1;
        if (diag01 || (tK[sl] == 0 && tI[sl] > 0)) {
          double* dst = (diag01 ? Dall + tI[sl] * 16 * kCholMStride : Lp0 + 16 * tI[sl] * kCholMStride) + off_cd;
#pragma unroll
          for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = acc[sl][g];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    LLD_CHOL_STAMP(2);
    __syncthreads();                                                   // prologue publish done
    __syncthreads();                                                   // diagonal tile 0 factored: Li = L_00^-1
    LLD_CHOL_STAMP(3);
    for (int J = 0; J < NT; J++) {
      // Per-lane LDS offsets, made opaque once per iteration: otherwise the per-slot addresses are hoisted out of the J loop as
      // loop invariants and push the accumulator tiles out of the register file.
      int off_cd = lrow * kCholMStride + lcol, off_ab = lcol * kCholMStride + lrow;
      asm volatile("" : "+v"(off_cd), "+v"(off_ab));
      double* Lp = Lp0 + (J & 1) * kCholMN * kCholMStride;
      double* Lnext = Lp0 + ((J + 1) & 1) * kCholMN * kCholMStride;
      LLD_CHOL_STAMP(8 + 6 * J);
      // (c) L_IJ = A_IJ L_JJ^-T on the matrix cores; keep it (back substitution) and publish it (operand of d)
      const unsigned mcol = __builtin_amdgcn_readfirstlane(cmask[J]);
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        if (mcol & (1u << sl)) {
          const double* pa = Lp + 16 * tI[sl] * kCholMStride + off_ab;
          const double* pb = Li + off_ab;
          v4d c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pb[4 * kk], c, 0, 0, 0);
          acc[sl] = c;
          double* dst = Lp + 16 * tI[sl] * kCholMStride + off_cd;
#pragma unroll
          for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = c[g];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      LLD_CHOL_STAMP(9 + 6 * J);
      __syncthreads();                                                 // (c) done
      LLD_CHOL_STAMP(10 + 6 * J);
      // (d) trailing update; column J+1 and the diagonal tile J+2 are final afterwards and are published for the next steps.
      //     The diagonal tile J+1 is not touched: the panel wave finishes it from its published copy (lookahead).
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        if (tK[sl] > J && !(tI[sl] == J + 1 && tK[sl] == J + 1)) {
          const double* pa = Lp + 16 * tI[sl] * kCholMStride + off_ab;
          const double* pb = Lp + 16 * tK[sl] * kCholMStride + off_ab;
          v4d c = acc[sl];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk], pb[4 * kk], c, 0, 0, 0);
          acc[sl] = c;
          const bool next_col = tK[sl] == J + 1, next_diag = tI[sl] == J + 2 && tK[sl] == J + 2;
          if (next_col || next_diag) {
            double* dst = (next_diag ? Dall + (J + 2) * 16 * kCholMStride : Lnext + 16 * tI[sl] * kCholMStride) + off_cd;
#pragma unroll
            for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = c[g];
          }
        }
        if (sl & 1) __builtin_amdgcn_sched_barrier(0);                 // let the loads of one tile overlap the MFMAs of its neighbour, not more
      }
      // forward substitution of the right-hand side below tile row J + 1: y_I -= L_IJ y_J for the tile rows I = J + 2 + (wave - 1), + 7, ...
      // (lane = row of the tile + 16 x quarter of the columns; L(:,J) is in the panel buffer, y_J is final since the previous column)
      for (int I = J + 1 + wave; I < NT; I += kCholMTileWaves) {
        const double* pr = Lp + (16 * I + lcol) * kCholMStride + 4 * lrow;
        const double* yj = y + 16 * J + 4 * lrow;
        double dotv = pr[0] * yj[0] + pr[1] * yj[1] + pr[2] * yj[2] + pr[3] * yj[3];
        dotv += __shfl_xor(dotv, 16); dotv += __shfl_xor(dotv, 32);
        if (lane < 16) y[16 * I + lane] -= dotv;
      }
      LLD_CHOL_STAMP(12 + 6 * J);
      __syncthreads();                                                 // (d) + lookahead done
      LLD_CHOL_STAMP(13 + 6 * J);
    }
    LLD_CHOL_STAMP(4);
    // back substitution L^T x = y: L lives in the register tiles, s_c = sum_{i below tile J} L[i][16J + c] x_i
    for (int J = NT - 1; J >= 0; J--) {
      double part = 0.0; bool any = false;
      const unsigned mcol = __builtin_amdgcn_readfirstlane(cmask[J]);
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        if (mcol & (1u << sl)) {
#pragma unroll
          for (int g = 0; g < 4; g++) part += acc[sl][g] * x[16 * tI[sl] + lrow + 4 * g];
          any = true;
        }
      }
      if (any) { part += __shfl_xor(part, 16); part += __shfl_xor(part, 32); }   // wave-uniform; sum over the 4 row groups of a column
      if (lane < 16) colsum[(wave - 1) * 16 + lane] = part;            // one row of partial sums per tile wave: no atomics
      __syncthreads();                                                 // column sums of J complete
      __syncthreads();                                                 // x_J ready
    }
    LLD_CHOL_STAMP(5);
  }
  const bool ok = *okf != 0.0;
  solve_epilogue(A, W, S, x, scratch, ok, 0);
  LLD_CHOL_STAMP(6);
}


#include "lld_ba_chol_sparse.h"   // round 5: the same factorisation along the structure of S, two panel wavefronts where the plan has two chains


