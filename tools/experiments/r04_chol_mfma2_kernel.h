// lld_ba_chol_exp.h - EXPERIMENTS BUILD ONLY (included by lld_ba_kernels.h under LLD_EXPERIMENTS, inside namespace lldba).
// Round 4's attempt at the dependent chain of ba_chol_mfma_kernel: built, parity-tested (tests/test_gpu_ba.py passes with
// LLD_BA_CHOL_V2=1 on the experiments library), timed stage by stage - and NOT faster: 146 - 152 us per launch against 130 for round 3's
// kernel (profiles/r04_chol_stage_budget_*.txt).  Kept for the measurements, not shipped in the product library.
// ------------------------------------------------------------------ the same factorisation with a shorter dependent chain (round 4)
// profiles/r04_chol_stage_budget.txt (s_memtime stamps inside ba_chol_mfma_kernel, one LBA-B window, 142 us): loading S 12.8 us, the 19 tile
// columns 97 us - of which the panel wave's tile factorisation 44 (2.0 - 2.9 us per tile: 16 pivots x ~45 readlane / FMA instructions of one
// wavefront), the L_IJ phase it waits for 21, its forward substitution of y 11, its update of the next diagonal tile 8, and in the first six
// columns 10 more of waiting for the trailing update - and 25 us of back substitution (two barriers per tile).  ba_chol_mfma2_kernel keeps the
// data layout (whole lower tile triangle in registers, 7 tile wavefronts + 1 panel wavefront) and changes what sits on the chain:
//   * the diagonal tile is factored IN the accumulator layout of v_mfma_f64_16x16x4_f64: pivot c is one v_readlane, a v_rsq_f64 + Newton
//     chain and ONE matrix-core rank-1 update  T -= v v^T  whose two operands are the same register (row c of the symmetric tile sits in
//     lanes 16 (c & 3) .. + 15, exactly where the K-slice c & 3 of both operands lives); a second rank-1 update per pivot carries the
//     identity along and leaves L^-1 (what L_IJ = A_IJ L_JJ^-T and the back substitution need) - no row-per-lane copy, no 15 broadcasts
//     per pivot;
//   * the panel wavefront no longer waits for the tile wavefronts' L_IJ phase: it forms its own copy of L_(J+1)J from the raw published
//     column (4 matrix-core instructions), updates the next diagonal tile in registers and starts factoring while the others still
//     compute their L_IJ (the owner of tile (J+1, J) publishes its copy in a side buffer so that the raw values stay readable);
//   * the forward substitution of the right-hand side leaves the panel wavefront: the owner of tile (I, J) updates y_I on the matrix
//     cores (y as a one-column B operand) during the trailing update; the panel only touches y_(J+1);
//   * back substitution with ONE barrier per tile row: x_J is computed by the wavefront that holds L_(J+1)J in its registers (the only
//     term that needs x_(J+1)), while all wavefronts already sum the other tiles' contributions to column J - 1.
constexpr int kChol2Slots = 25;                                      // off-diagonal tiles per tile wave for 19 tile rows (171 / 7, rounded up)
constexpr int kChol2LdsDoubles = 2 * kCholMN * kCholMStride + kCholMMaxTiles * 16 * kCholMStride + 3 * 16 * kCholMStride + 3 * kCholMN + 16 + 32;

// Cholesky of the symmetric 16x16 tile t (C/D layout: lane l, register g <-> row (l >> 4) + 4 g, column l & 15; FULL tile, both
// triangles) on the matrix core; on return F = L^-1 in the same layout (t is consumed).  False if a pivot is not positive.
__device__ __forceinline__ bool chol_tile_factor_mfma(v4d& t, v4d& F, int lrow, int lcol) {
  bool ok = true;
#pragma unroll
  for (int g = 0; g < 4; g++) F[g] = (lrow + 4 * g == lcol) ? 1.0 : 0.0;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int q = c & 3, g = c >> 2;
    const double d = readlane_f64(t[g], 16 * q + c);                 // T[c][c]
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    const double inv = rsqrt_nr(d);
    const bool rowc = lrow == q;                                     // the lanes that hold row c of a tile in register g
    const double v = (rowc && lcol >= c) ? t[g] * inv : 0.0;         // L[j][c], j = lcol >= c (v[c] = sqrt(d)); K-slice q of BOTH operands
    const double u = rowc ? F[g] * inv : 0.0;                        // row c of L^-1 once scaled
    const double vp = (lcol == c) ? 0.0 : v;                         // ... which is set, not updated
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, t, 0, 0, 0);     // T[i][j] -= L[i][c] L[j][c]
    F = __builtin_amdgcn_mfma_f64_16x16x4f64(-vp, u, F, 0, 0, 0);    // F[m][n] -= L[m][c] F[c][n] / sqrt(d), m > c
    F[g] = rowc ? u : F[g];
  }
  return ok;
}

// The same factorisation with FOUR pivots per matrix-core update.  The four rows of a diagonal 4x4 block step are first replicated into
// every lane group (8 ds_bpermute pairs, issued together), the four pivots then run on the vector ALU alone - per pivot one v_readlane of
// the diagonal entry, the v_rsq_f64 + Newton chain, and for the (at most three) later rows of the block one v_readlane + two FMAs each -
// and ONE rank-4 update per block brings the rest of the tile (and of L^-1) up to date: the K-slices 0..3 of the operand register are the
// four pivot rows.  8 matrix-core instructions per tile instead of 32, none of them on the pivot-to-pivot chain.
__device__ __forceinline__ bool chol_tile_factor_blk(v4d& t, v4d& F, int lrow, int lcol) {
  bool ok = true;
#pragma unroll
  for (int g = 0; g < 4; g++) F[g] = (lrow + 4 * g == lcol) ? 1.0 : 0.0;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    double P[4], Q[4];                                                  // P[q] = T[4 g + q][lcol], Q[q] = F[4 g + q][lcol] in EVERY lane group
#pragma unroll
    for (int q = 0; q < 4; q++) { P[q] = __shfl(t[g], 16 * q + lcol); Q[q] = __shfl(F[g], 16 * q + lcol); }
    double Vop = 0.0, Uop = 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = 4 * g + q;
      const double d = readlane_f64(P[q], c);
      if (!(d > 0.0) || !isfinite(d)) ok = false;
      const double inv = rsqrt_nr(d);
      const double v = (lcol >= c) ? P[q] * inv : 0.0;                  // L[j][c], j = lcol
      const double u = Q[q] * inv;                                      // row c of L^-1
#pragma unroll
      for (int q2 = q + 1; q2 < 4; q2++) {
        const double sc = readlane_f64(v, 4 * g + q2);                  // L[c2][c]
        P[q2] -= sc * v; Q[q2] -= sc * u;
      }
      Vop = (lrow == q) ? v : Vop; Uop = (lrow == q) ? u : Uop;
    }
    F[g] = Uop;                                                         // rows 4 g .. 4 g + 3 of L^-1 are final
    if (g < 3) {
      const double Vp = (lcol > 4 * g + 3) ? Vop : 0.0;                 // only the rows below the block take the update
      t = __builtin_amdgcn_mfma_f64_16x16x4f64(-Vop, Vop, t, 0, 0, 0);
      F = __builtin_amdgcn_mfma_f64_16x16x4f64(-Vp, Uop, F, 0, 0, 0);
    }
  }
  return ok;
}

template <bool kBlockedFactor>
__global__ __launch_bounds__(kCholMThreads) void ba_chol_mfma2_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.x);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  const int nf = W.n_free, n = 6 * nf, NT = (n + 15) >> 4, N = NT << 4;
  constexpr int TS = 16 * kCholMStride;                    // doubles of one staged tile
  double* Lp0 = lds;                                       // [2][N][17] panel buffers: column J in buffer J & 1 (raw, then L; rows of tile J + 1 stay raw)
  double* Dall = Lp0 + 2 * kCholMN * kCholMStride;         // [NT][16][17] diagonal tiles: raw (full symmetric) until factored, then L_JJ^-1
  double* Li = Dall + kCholMMaxTiles * TS;                 // [16][17] L_JJ^-1 of the current column
  double* Lsub = Li + TS;                                  // [16][17] L_(J+1)J as published by its owner
  double* Ps = Lsub + TS;                                  // [16][17] panel wavefront's private scratch (accumulator layout -> operand layout)
  double* colsum = Ps + TS;                                // [2][7][16] back substitution: per tile wavefront partial sums (room for N)
  double* y = colsum + kCholMN;                            // [N] right-hand side -> forward solution
  double* x = y + kCholMN;                                 // [N] solution
  double* rb = x + kCholMN;                                // [16] right-hand side of one back-substitution step
  double* scratch = rb + 16;                               // [32]
  double* okf = scratch + 31;
  const double* Sg = A.S + W.S_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 4, lcol = lane & 15;
#ifdef LLD_EXPERIMENTS
  long long* stamp_base = A.chol_stamps ? A.chol_stamps + ((size_t)W.win_index * 8 + wave) * kCholStampSlots : nullptr;
#endif
  LLD_CHOL_STAMP(0);
  if (tid < N) y[tid] = (tid < n) ? A.bschur[W.x_off + tid] : 0.0;
  if (tid == 0) *okf = 1.0;
  const int off_c = lrow * kCholMStride + lcol;            // accumulator layout: + 4 g rows
  const int off_o = lcol * kCholMStride + lrow;            // operand layout: + 4 kk columns

  if (wave == 0) {
    // ================================================================ panel wave
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();                                                   // B0: tiles loaded, y staged
    LLD_CHOL_STAMP(1);
    __syncthreads();                                                   // B1: prologue publish done: column 0, diagonal tiles 0 and 1
    LLD_CHOL_STAMP(2);
    v4d t, F;
    bool ok = true;
    // store F = L_JJ^-1 (Li for the column's L_IJ, Dall for the back substitution), then y_J <- L_JJ^-1 yv (yv: lane's lcol-th entry)
    auto finish_tile = [&](int J, double yv) {
      double* Dg = Dall + J * TS;
#pragma unroll
      for (int g = 0; g < 4; g++) { Li[off_c + 4 * g * kCholMStride] = F[g]; Dg[off_c + 4 * g * kCholMStride] = F[g]; }
      if (lane < 16) rb[lane] = yv;
      double sacc = 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) sacc += Li[lcol * kCholMStride + 4 * lrow + q] * rb[4 * lrow + q];
      sacc += __shfl_xor(sacc, 16); sacc += __shfl_xor(sacc, 32);
      if (lane < 16) y[16 * J + lane] = sacc;
    };
    if (NT > 0) {
#pragma unroll
      for (int g = 0; g < 4; g++) t[g] = Dall[off_c + 4 * g * kCholMStride];
      ok = (kBlockedFactor ? chol_tile_factor_blk(t, F, lrow, lcol) : chol_tile_factor_mfma(t, F, lrow, lcol)) && ok;
      finish_tile(0, y[lcol]);
    }
    LLD_CHOL_STAMP(3);
    __syncthreads();                                                   // B2: diagonal tile 0 factored
    for (int J = 0; J < NT; J++) {
      const double* Lp = Lp0 + (J & 1) * kCholMN * kCholMStride;
      double pa[4] = {0.0, 0.0, 0.0, 0.0};
      LLD_CHOL_STAMP(8 + 6 * J);
      if (J + 1 < NT) {
        // own copy of L_(J+1)J = A_(J+1)J L_JJ^-T from the raw column (the tile waves compute theirs meanwhile) ...
        const double* praw = Lp + 16 * (J + 1) * kCholMStride + off_o;
        const double* pb = Li + off_o;
        v4d c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(praw[4 * kk], pb[4 * kk], c, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++) Ps[off_c + 4 * g * kCholMStride] = c[g];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) pa[kk] = Ps[off_o + 4 * kk];    // operand layout: lane (i = lcol, k' = lrow) holds L[i][4 kk + k']
        // ... the next diagonal tile (published with the updates of columns < J) takes column J's update in registers
        const double* Dg = Dall + (J + 1) * TS;
#pragma unroll
        for (int g = 0; g < 4; g++) t[g] = Dg[off_c + 4 * g * kCholMStride];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) t = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[kk], pa[kk], t, 0, 0, 0);
      }
      LLD_CHOL_STAMP(9 + 6 * J);
      __syncthreads();                                                 // Bc: the tile waves' L(:,J) is published
      LLD_CHOL_STAMP(10 + 6 * J);
      if (J + 1 < NT) {
        // y_(J+1) -= L_(J+1)J y_J (every other y_I is updated by the owner of tile (I, J)), factor, y_(J+1) <- L^-1 y_(J+1)
        double sacc = 0.0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) sacc += pa[kk] * y[16 * J + 4 * kk + lrow];
        sacc += __shfl_xor(sacc, 16); sacc += __shfl_xor(sacc, 32);
        const double yv = y[16 * (J + 1) + lcol] - sacc;
        LLD_CHOL_STAMP(11 + 6 * J);
        ok = (kBlockedFactor ? chol_tile_factor_blk(t, F, lrow, lcol) : chol_tile_factor_mfma(t, F, lrow, lcol)) && ok;
        finish_tile(J + 1, yv);
      }
      LLD_CHOL_STAMP(12 + 6 * J);
      __syncthreads();                                                 // Bd: trailing update done, column J + 1 published
      LLD_CHOL_STAMP(13 + 6 * J);
    }
    if (!ok && lane == 0) *okf = 0.0;
    LLD_CHOL_STAMP(4);
    for (int J = NT - 1; J >= 0; J--) __syncthreads();                 // back substitution: the tile waves' work
    LLD_CHOL_STAMP(5);
  } else {
    // ================================================================ tile waves
    // Tile (I, K) belongs to tile wave (I + 2 K) mod 7.  The OFF-DIAGONAL tiles of a wave live in registers (<= 25 slots of 4 doubles per
    // lane for 19 tile rows) in COLUMN-major order - column K holds the rows first_off(K), + 7, ... < NT in consecutive slots - so that
    // every phase touches a contiguous slot range (column J for L_IJ, everything from column J + 1 on for the trailing update): a slot
    // is tested with one or two scalar compares against the range and its (I, K) follows from scalar arithmetic, where round 3's
    // row-major order kept (I, K) per slot in spilled scalar registers and paid ~35 cycles of v_readlane / compares / branches per slot
    // and phase.  (A switch on the slot number that jumps into the unrolled bodies was tried: the merged register webs of the 25
    // accumulator tiles made the compiler copy and spill them.)
    // The DIAGONAL tiles stay in LDS (Dall), where the panel wave needs them anyway: their owners update them in place; that frees
    // three register slots per wave, without which the kernel spilled two tiles to scratch (profiles/r04_chol_stage_budget*).
    const int w0 = wave - 1;
    auto first_row = [&](int K) { int r = (w0 - 3 * K) % 7; if (r < 0) r += 7; return K + r; };      // smallest I >= K with (I + 2 K) mod 7 == w0
    auto first_off = [&](int K) { const int I = first_row(K); return I == K ? I + 7 : I; };            // ... smallest I > K
    auto off_count = [&](int K) { const int f = first_off(K); return f < NT ? (NT - 1 - f) / 7 + 1 : 0; };
    auto next_tile = [&](int& I, int& K) {                             // column-major successor; K >= NT: none left
      I += 7;
      if (I >= NT) { do { K++; I = first_off(K); } while (K < NT && I >= NT); }
    };
    const int diag0 = (5 * w0) % 7;                                    // this wave owns the diagonal tiles diag0, diag0 + 7, ... (3 K == w0 mod 7)
    // S -> registers / LDS.  All loads go out before the first value is touched.
    v4d acc[kChol2Slots];
    int offg[4];
#pragma unroll
    for (int g = 0; g < 4; g++) offg[g] = (lrow + 4 * g) * n + lcol;
    {
      int K = 0, I = first_off(0);
      while (K < NT && I >= NT) { K++; I = first_off(K); }
#pragma unroll
      for (int sl = 0; sl < kChol2Slots; sl++) {
        v4d v = {0.0, 0.0, 0.0, 0.0};
        if (K < NT) {
          const double* base = Sg + (16 * I) * n + 16 * K;
          if (16 * I + 16 <= n) {                                       // interior tile (wave-uniform): scalar base + the shared lane offsets
#pragma unroll
            for (int g = 0; g < 4; g++) v[g] = base[offg[g]];
          } else {                                                     // last tile row of a padded system: rows >= n are zero
#pragma unroll
            for (int g = 0; g < 4; g++) { const bool inside = 16 * I + lrow + 4 * g < n; const double t_ = base[inside ? offg[g] : 0]; v[g] = inside ? t_ : 0.0; }
          }
          next_tile(I, K);
        }
        acc[sl] = v;
      }
    }
    // diagonal tiles -> LDS as FULL symmetric tiles (the factorisation reads both triangles; S holds the lower block triangle only),
    // identity in the padding rows / columns
    for (int K = diag0; K < NT; K += 7) {
      const double* base = Sg + (16 * K) * n + 16 * K;
      const int col = 16 * K + lcol;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int row = 16 * K + lrow + 4 * g;
        const bool inside = row < n && col < n;
        const int off = col <= row ? offg[g] : lcol * n + lrow + 4 * g;          // upper triangle: the mirrored element
        const double t_ = base[inside ? off : 0];
        Dall[K * TS + off_c + 4 * g * kCholMStride] = inside ? t_ : (row == col ? 1.0 : 0.0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // column 0 (raw) -> panel buffer 0: its tiles are the slots 0, 1, 2
    {
      int I = first_off(0);
#pragma unroll
      for (int sl = 0; sl < 3; sl++) {
        if (I < NT) {
          double* dst = Lp0 + 16 * I * kCholMStride + off_c;
#pragma unroll
          for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = acc[sl][g];
        }
        I += 7;
      }
    }
    LLD_CHOL_STAMP(1);
    __syncthreads();                                                   // B0: y staged
    LLD_CHOL_STAMP(2);
    __syncthreads();                                                   // B1: column 0 and the diagonal tiles published
    __syncthreads();                                                   // B2: diagonal tile 0 factored: Li = L_00^-1, y_0 final
    LLD_CHOL_STAMP(3);
    int cs = 0;                                                        // first slot of column J
    for (int J = 0; J < NT; J++) {
      // Per-lane LDS offsets, made opaque once per iteration: otherwise per-slot addresses are hoisted out of the J loop as loop
      // invariants and push the accumulator tiles out of the register file.
      int off_cd = off_c, off_ab = off_o, off_y = lrow;
      asm volatile("" : "+v"(off_cd), "+v"(off_ab), "+v"(off_y));
      const bool col0 = lcol == 0;
      double* Lp = Lp0 + (J & 1) * kCholMN * kCholMStride;
      double* Lnext = Lp0 + ((J + 1) & 1) * kCholMN * kCholMStride;
      const int cntJ = off_count(J);
      LLD_CHOL_STAMP(8 + 6 * J);
      // (c) L_IJ = A_IJ L_JJ^-T on the matrix cores; keep it (back substitution) and publish it (operand of d).  Tile (J + 1, J) goes to
      //     the side buffer: the panel wave reads the raw rows of tile J + 1 in this phase.
      {
        const double* pbv = Li + off_ab;
        const int fI = first_off(J);
#pragma unroll
        for (int sl = 0; sl < kChol2Slots; sl++) {
          const int I = fI + 7 * (sl - cs);
          if (sl >= cs && I < NT) {                                     // the slots cs, cs + 1, ... of column J (wave-uniform)
            const double* pa = Lp + 16 * I * kCholMStride + off_ab;
            v4d c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pbv[4 * kk], c, 0, 0, 0);
            acc[sl] = c;
            double* dst = (I == J + 1 ? Lsub : Lp + 16 * I * kCholMStride) + off_cd;
#pragma unroll
            for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = c[g];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      LLD_CHOL_STAMP(9 + 6 * J);
      __syncthreads();                                                 // Bc: (c) done
      LLD_CHOL_STAMP(10 + 6 * J);
      // (d) the trailing update.  First the diagonal tiles K >= J + 2 this wave owns, in place in LDS (K = J + 1 is the panel wave's) ...
      {
        int K = diag0;
        while (K < J + 2) K += 7;
        for (; K < NT; K += 7) {
          const double* pbp = Lp + 16 * K * kCholMStride + off_ab;
          double* Dg = Dall + K * TS + off_cd;
          v4d c;
#pragma unroll
          for (int g = 0; g < 4; g++) c[g] = Dg[4 * g * kCholMStride];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) { const double b = pbp[4 * kk]; c = __builtin_amdgcn_mfma_f64_16x16x4f64(-b, b, c, 0, 0, 0); }
#pragma unroll
          for (int g = 0; g < 4; g++) Dg[4 * g * kCholMStride] = c[g];
        }
      }
      //     ... then every off-diagonal tile from column J + 1 on.  The tiles of column J + 1 are final afterwards and are published raw
      //     for the next column; the owner of tile (I, J + 1) also carries the forward substitution of the right-hand side,
      //     y_I -= L_IJ y_J, on the matrix cores (y_J as a one-column B operand; the operand L_IJ is in registers for the update anyway).
      {
        int K = J + 1, I = first_off(K);
        while (K < NT && I >= NT) { K++; I = first_off(K); }
        const double* yJ = y + 16 * J + off_y;
        const int s0 = cs + cntJ;                                       // first slot of column J + 1
#pragma unroll
        for (int sl = 0; sl < kChol2Slots; sl++) {
          if (sl >= s0 && K < NT) {
            const double* pap = Lp + 16 * I * kCholMStride + off_ab;
            double pa[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) pa[kk] = pap[4 * kk];
            const double* pbp = (K == J + 1 ? Lsub : Lp + 16 * K * kCholMStride) + off_ab;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[kk], pbp[4 * kk], acc[sl], 0, 0, 0);
            if (K == J + 1) {
              double* yI = y + 16 * I + off_y;
              v4d cy;
#pragma unroll
              for (int g = 0; g < 4; g++) { const double yv = yI[4 * g]; cy[g] = col0 ? yv : 0.0; }
#pragma unroll
              for (int kk = 0; kk < 4; kk++) { const double yv = yJ[4 * kk]; cy = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[kk], col0 ? yv : 0.0, cy, 0, 0, 0); }
              double* dst = Lnext + 16 * I * kCholMStride + off_cd;
#pragma unroll
              for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = acc[sl][g];
              if (col0) {
#pragma unroll
                for (int g = 0; g < 4; g++) yI[4 * g] = cy[g];
              }
            }
            next_tile(I, K);
          }
          if (sl & 1) __builtin_amdgcn_sched_barrier(0);               // let the loads of one tile overlap the MFMAs of its neighbour, not more
        }
      }
      cs += cntJ;
      LLD_CHOL_STAMP(12 + 6 * J);
      __syncthreads();                                                 // Bd: (d) done
      LLD_CHOL_STAMP(13 + 6 * J);
    }
    LLD_CHOL_STAMP(4);
    // back substitution L^T x = y, one barrier per tile row.  x_J = L_JJ^-T (y_J - sum_{I > J} L_IJ^T x_I): the term I = J + 1 is the only one
    // that needs x_(J+1), and the wavefront that owns tile (J + 1, J) adds it itself when it computes x_J; the other terms (colsum') were
    // summed by all wavefronts one step earlier, while x_(J+1) was being computed.  (cs = number of this wavefront's tiles here.)
    for (int J = NT - 1; J >= 0; J--) {
      double* cur = colsum + (J & 1) * 7 * 16;
      double* nxt = colsum + ((J + 1) & 1) * 7 * 16;
      int off_x = lrow;
      asm volatile("" : "+v"(off_x));                                   // (no per-slot addresses hoisted out of the loop, see above)
      cs -= off_count(J);                                               // first slot of column J
      const bool exec = J + 1 < NT ? first_off(J) == J + 1 : w0 == 0;   // the owner of tile (J + 1, J); the last tile row: wave 1
      if (exec) {
        double sacc = 0.0;
        if (J + 1 < NT) {
          double xv[4];
#pragma unroll
          for (int g = 0; g < 4; g++) xv[g] = x[16 * (J + 1) + off_x + 4 * g];
#pragma unroll
          for (int sl = 0; sl < kChol2Slots; sl++) {
            if (sl == cs) {
#pragma unroll
              for (int g = 0; g < 4; g++) sacc += acc[sl][g] * xv[g];
            }
          }
          sacc += __shfl_xor(sacc, 16); sacc += __shfl_xor(sacc, 32);
#pragma unroll
          for (int w = 0; w < kCholMTileWaves; w++) sacc += cur[w * 16 + lcol];
        }
        if (lane < 16) rb[lane] = y[16 * J + lane] - sacc;
        const double* Di = Dall + J * TS;                               // L_JJ^-1
        double xc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) xc += Di[(4 * lrow + q) * kCholMStride + lcol] * rb[4 * lrow + q];
        xc += __shfl_xor(xc, 16); xc += __shfl_xor(xc, 32);
        if (lane < 16) x[16 * J + lane] = xc;
      }
      if (J >= 1) {
        // colsum' of column J - 1: its tiles (I, J - 1) with I >= J + 1 (x_I known since the previous barrier)
        int fI = first_off(J - 1), s0 = cs - off_count(J - 1);
        if (fI == J) { fI += 7; s0++; }                                 // tile (J, J - 1) waits for x_J: its owner adds it in the next step
        double part = 0.0;
#pragma unroll
        for (int sl = 0; sl < kChol2Slots; sl++) {
          const int I = fI + 7 * (sl - s0);
          if (sl >= s0 && I < NT) {
#pragma unroll
            for (int g = 0; g < 4; g++) part += acc[sl][g] * x[16 * I + off_x + 4 * g];
          }
        }
        part += __shfl_xor(part, 16); part += __shfl_xor(part, 32);
        if (lane < 16) nxt[w0 * 16 + lane] = part;
      }
      __syncthreads();
    }
    LLD_CHOL_STAMP(5);
  }
  const bool okk = *okf != 0.0;
  solve_epilogue(A, W, S, x, scratch, okk, 0);
  LLD_CHOL_STAMP(6);
}

