// lld_ba_chol_sparse.h — the reduced camera system solved along its structure (round 5): right-looking Cholesky on 16x16 tiles of the fp64
// matrix cores like ba_chol_mfma_kernel, but only over the tiles of the host's symbolic factorisation (lld_ba_chol_plan.h), in the host's
// elimination order, and with TWO panel wavefronts where the plan splits the cameras into two chains that meet in a separator.
// Stands in for LinearSolverEigen::solve with its symbolic decomposition (Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124, :147-232).
//
// Included by lld_ba_kernels.h after the dense kernel (chol tile layout, readlane_f64, v4d, solve_epilogue are defined there).
//
// One workgroup of 512 lanes per window.  Wavefront 0 / 1: panel wavefront of chain 0 / 1 (the second one idles in a one-chain plan);
// wavefronts 2..7: tile wavefronts, each with up to 24 tiles of the factor resident in registers (lane l of a tile wavefront also keeps the
// coordinates of slot l in three registers: v_readlane hands them to the scalar unit when a slot is touched - no tables in scalar registers).
// A step eliminates one tile column per chain:
//   (c)  tile wavefronts: L_IJ = A_IJ L_JJ^-T for the column's non-zero tiles (raw tile in the step's panel buffer -> L in place + registers)
//   ---- barrier
//   (d)  tile wavefronts: T_IK -= L_IJ L_KJ^T for the tiles the plan lists, then publish the next step's columns (raw) into the other panel buffer
//        and the diagonal tiles of the step after into Dall; forward substitution of the right-hand side below the panel's rows;
//        panel wavefronts (lookahead): the next column's diagonal tile takes this step's update(s) from the panel buffer, its right-hand side
//        rows likewise, then the serial 16x16 factor (one lane per row, right-hand side and identity as extra rows: y_J and L_JJ^-1)
//   ---- barrier
// then the back substitution in reverse step order (tile wavefronts: column sums of L_IJ^T x_I per chain; panel wavefronts: x_J = L_JJ^-T (y_J - sum)).
// The permutation lives in the plan's row map only: S is gathered through it, x is scattered back through it.
#ifndef LLD_BA_CHOL_SPARSE_H
#define LLD_BA_CHOL_SPARSE_H

constexpr int kSpTile = 16 * kCholMStride;                          // doubles of one padded LDS tile
constexpr int kSpN = kSpMaxT * 16;
constexpr int kSpLdsDoubles = 2 * kSpPos * kSpTile + kSpMaxT * kSpTile + 2 * kSpTileWaves * 16 + 3 * kSpN + 32;
// (experiments build: s_memtime stamps are parked in LDS and leave for HBM when the kernel is done - a global store per stamp put a
// store round trip under the next s_waitcnt vmcnt(0) of every phase it was meant to time, round 5)
constexpr int kSpStampSlots = 144;      // 13 + 6 x 21 = 139 is the last one a plan of 22 steps leaves; 32-bit (a launch is 1e5 ticks)
#ifdef LLD_EXPERIMENTS
constexpr size_t kSpLdsBytes = kSpLdsDoubles * sizeof(double) + sizeof(CholPlan) + (kSpThreads / 64) * kSpStampSlots * sizeof(unsigned);
#define LLD_SP_STAMP(k) do { if (stamp_lds && lane == 0) stamp_lds[(k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
constexpr size_t kSpLdsBytes = kSpLdsDoubles * sizeof(double) + sizeof(CholPlan);
#define LLD_SP_STAMP(k) do { } while (0)
#endif

// Sum over the four lanes of a quad through the VALU (DPP quad_perm), not through the LDS pipe: the right-hand-side products below map a tile
// row to a quad (lane = 4 x row + quarter of the columns), so their reductions are two DPP adds where a lane = row + 16 x quarter layout needs
// two ds_bpermute round trips per add (a chain of eight dependent LDS round trips per tile row was 1 us of every step's tile-wavefront time).
__device__ __forceinline__ double quad_sum(double v) {
  double o = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += o;
  o = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x4E, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(v), 0x4E, 0xf, 0xf, true));          // quad_perm [2,3,0,1]
  return v + o;
}

// Panel wavefront: factor the 16x16 tile in Dg (lower triangle used) in place into L^-1 - all the column's L_IJ = A_IJ L_JJ^-T, the forward
// substitution y_J = L_JJ^-1 (...) and the back substitution need.  Same recurrence as chol_tile_factor (one lane per row, the identity as
// sixteen extra rows), without the right-hand side: that is the tile wavefronts' business here, off the chain of the factors.
__device__ __forceinline__ bool chol_tile_factor_inplace(double* Dg, int lane) {
  const int r = lane < 32 ? lane : 31;                               // 0..15 tile rows, 16..31 identity rows
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { const double v = Dg[(r & 15) * kCholMStride + c]; a[c] = (r < 16) ? v : (r - 16 == c ? 1.0 : 0.0); }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const double d = readlane_f64(a[c], c);
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * (1.5 - (0.5 * d) * (inv * inv));
    inv = inv * (1.5 - (0.5 * d) * (inv * inv));
    const double lc = a[c] * inv;
    a[c] = lc;
#pragma unroll
    for (int c2 = c + 1; c2 < 16; c2++) a[c2] -= lc * readlane_f64(lc, c2);
  }
  if (lane >= 16 && lane < 32) {                                     // lane 16+k holds column k of L^-1
#pragma unroll
    for (int c = 0; c < 16; c++) Dg[c * kCholMStride + (lane - 16)] = a[c];
  }
  return ok;
}

__global__ __launch_bounds__(kSpThreads) void ba_chol_sparse_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.x);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  const CholPlan* __restrict__ Pg = A.chol_plan + W.win_index;
  if (Pg->mode != 1) return;                                         // this window is the dense kernel's
  double* Lp0 = lds;                                                 // [2][kSpPos] panel buffers: the tiles of step s's columns in buffer s & 1 (raw, then L)
  double* Dall = Lp0 + 2 * kSpPos * kSpTile;                         // [NT] diagonal tiles: raw until factored, then L_JJ^-1
  double* colsum = Dall + kSpMaxT * kSpTile;                         // [2 chains][6 tile wavefronts][16] column sums of the back substitution
  double* y = colsum + 2 * kSpTileWaves * 16;                        // [N] right-hand side -> forward solution (permuted order)
  double* x = y + kSpN;                                              // [N] solution (permuted order)
  double* xo = x + kSpN;                                             // [n] solution in S's order
  double* scratch = xo + kSpN;                                       // [32]
  double* okf = scratch + 31;
  CholPlan* P = reinterpret_cast<CholPlan*>(scratch + 32);
  const double* Sg = A.S + W.S_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 4, lcol = lane & 15;
  const int n = 6 * W.n_free;
  {
    const int4* src = reinterpret_cast<const int4*>(Pg); int4* dst = reinterpret_cast<int4*>(P);
    for (int i = tid; i < (int)(sizeof(CholPlan) / 16); i += kSpThreads) dst[i] = src[i];
  }
  if (tid == 0) *okf = 1.0;
#ifdef LLD_EXPERIMENTS
  unsigned* stamp_lds = A.chol_stamps ? reinterpret_cast<unsigned*>(reinterpret_cast<char*>(P) + sizeof(CholPlan)) + wave * kSpStampSlots : nullptr;
  if (stamp_lds) for (int i = lane; i < kSpStampSlots; i += 64) stamp_lds[i] = 0;
#endif
  LLD_SP_STAMP(0);
#ifdef LLD_EXPERIMENTS
  if (stamp_lds && lane == 0) stamp_lds[7] = (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: SIMD_ID in bits 5:4
#endif
  __syncthreads();                                                   // B0: the plan is in LDS
  const int NT = P->NT, T = P->T, N = NT << 4;

  if (wave < 2) {
    // ================================================================ panel wavefront of chain `wave`
    const int ch = wave;
    const int J0 = P->cols[0][ch];
    if (J0 != kSpNone) {                                             // step 0's diagonal tile: fetched and factored here while the tile wavefronts load
      double* Dg = Dall + J0 * kSpTile;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int idx = lane + 64 * q, row = idx >> 4, col = idx & 15;
        const int R = P->rowmap[16 * J0 + row], C = P->rowmap[16 * J0 + col];
        const bool inside = R >= 0 && C >= 0;
        const int hi = R > C ? R : C, lo = R > C ? C : R;
        const double v = Sg[inside ? hi * n + lo : 0];
        Dg[row * kCholMStride + col] = inside ? v : (row == col ? 1.0 : 0.0);
      }
      if (!chol_tile_factor_inplace(Dg, lane) && lane == 0) *okf = 0.0;
    }
    LLD_SP_STAMP(1);
    __syncthreads();                                                 // B1: tiles loaded, y staged
    LLD_SP_STAMP(2);
    __syncthreads();                                                 // B2: prologue publish done
    LLD_SP_STAMP(3);
    for (int s = 0; s < T; s++) {
      double* Lp = Lp0 + (s & 1) * kSpPos * kSpTile;
      LLD_SP_STAMP(8 + 6 * s);
      const int Jn = P->cols[s + 1][ch];
      double* Dg = Dall + (Jn != kSpNone ? Jn : 0) * kSpTile;
      if (Jn != kSpNone) {
        // The next column's diagonal tile (published with the updates of the steps before s) takes step s's update(s) HERE, from L_(Jn)J tiles this
        // wavefront forms itself (raw tile x L_JJ^-T, both in LDS since the last barrier) - its chain does not wait for the tile wavefronts' (c).
        // L goes back in place: it is the tile wavefronts' operand in (d), and the slot's owner fetches it from there.
        int off_cd = lrow * kCholMStride + lcol, off_ab = lcol * kCholMStride + lrow;
        asm volatile("" : "+v"(off_cd), "+v"(off_ab));
        v4d c;
#pragma unroll
        for (int g = 0; g < 4; g++) c[g] = Dg[4 * g * kCholMStride + off_cd];
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
          const int J = P->cols[s][c2];
          if (J == kSpNone) continue;
          const int p = P->pos[J][Jn];
          if (p == kSpNone) continue;
          double* Lt = Lp + p * kSpTile;
          const double* pa = Lt + off_ab;
          const double* pb = Dall + J * kSpTile + off_ab;
          v4d l = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; kk++) l = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pb[4 * kk], l, 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; g++) Lt[4 * g * kCholMStride + off_cd] = l[g];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk], pa[4 * kk], c, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) Dg[4 * g * kCholMStride + off_cd] = c[g];
      }
      LLD_SP_STAMP(9 + 6 * s);
      __syncthreads();                                               // (c) done: Lp holds L of step s's columns
      LLD_SP_STAMP(10 + 6 * s);
      if (Jn != kSpNone && !chol_tile_factor_inplace(Dg, lane) && lane == 0) *okf = 0.0;
      LLD_SP_STAMP(12 + 6 * s);
      __syncthreads();                                               // (d) + lookahead done
      LLD_SP_STAMP(13 + 6 * s);
    }
    LLD_SP_STAMP(4);
    // back substitution L^T x = y in reverse step order: x_J = L_JJ^-T (y_J - s_J), s_J = the tile wavefronts' column sums of L_IJ^T x_I
    for (int s = T - 1; s >= 0; s--) {
      __syncthreads();                                               // column sums of step s complete
      const int J = P->cols[s][ch];
      if (J != kSpNone) {
        const double* Di = Dall + J * kSpTile;                       // L_JJ^-1
        const double* cs = colsum + ch * kSpTileWaves * 16;
        const int c = lane & 15, part = lane >> 4;
        double xc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int r = 4 * part + q;
          double sum = 0.0;
#pragma unroll
          for (int w = 0; w < kSpTileWaves; w++) sum += cs[w * 16 + r];
          xc += Di[r * kCholMStride + c] * (y[16 * J + r] - sum);
        }
        xc += __shfl_xor(xc, 16); xc += __shfl_xor(xc, 32);
        if (lane < 16) x[16 * J + c] = xc;
      }
      __syncthreads();                                               // x of step s ready
    }
    LLD_SP_STAMP(5);
  } else {
    // ================================================================ tile wavefronts
    const int w = wave - 2;
    // lane l <-> slot l: tile row / column and the tile's position in its own column's panel buffer (0x80 | row for a diagonal tile)
    int myI = kSpNone, myK = kSpNone, selfpos = kSpNone;
    if (lane < kSpSlots) {
      myI = P->slotI[w][lane]; myK = P->slotK[w][lane];
      if (myI != kSpNone) selfpos = myI == myK ? (0x80 | myI) : P->pos[myK][myI];
    }
    // right-hand side in permuted order
    {
      const int t = tid - 128;
      if (t < N) { const int R = P->rowmap[t]; y[t] = R >= 0 ? A.bschur[W.x_off + R] : 0.0; }
    }
    // S -> registers through the row map; every load goes out before the first value is touched
    v4d acc[kSpSlots];
#pragma unroll
    for (int sl = 0; sl < kSpSlots; sl++) {
      v4d v = {0.0, 0.0, 0.0, 0.0};
      const int I = __builtin_amdgcn_readlane(myI, sl);
      if (I != kSpNone) {
        const int K = __builtin_amdgcn_readlane(myK, sl);
        const int C = P->rowmap[16 * K + lcol];
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int R = P->rowmap[16 * I + lrow + 4 * g];
          const bool inside = R >= 0 && C >= 0;
          const int hi = R > C ? R : C, lo = R > C ? C : R;
          v[g] = Sg[inside ? hi * n + lo : 0];
        }
      }
      acc[sl] = v;
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const unsigned pm = __builtin_amdgcn_readfirstlane(P->padmask[w]);
#pragma unroll
      for (int sl = 0; sl < kSpSlots; sl++) {
        if (pm & (1u << sl)) {
          const int I = __builtin_amdgcn_readlane(myI, sl), K = __builtin_amdgcn_readlane(myK, sl);
          const int C = P->rowmap[16 * K + lcol];
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int R = P->rowmap[16 * I + lrow + 4 * g];
            const bool inside = R >= 0 && C >= 0;
            acc[sl][g] = inside ? acc[sl][g] : ((16 * I + lrow + 4 * g == 16 * K + lcol) ? 1.0 : 0.0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    LLD_SP_STAMP(1);
    __syncthreads();                                                 // B1
    int off_cd0 = lrow * kCholMStride + lcol;
    asm volatile("" : "+v"(off_cd0));
    // publish the slots of `mask`: an off-diagonal tile into its column's panel buffer `Lto`, a diagonal tile into Dall
#define LLD_SP_PUBLISH(mask_, Lto_, off_cd_)                                                                         \
    do {                                                                                                             \
      const unsigned pmask_ = (mask_);                                                                               \
      _Pragma("unroll") for (int sl = 0; sl < kSpSlots; sl++) {                                                      \
        if (pmask_ & (1u << sl)) {                                                                                   \
          const int sp = __builtin_amdgcn_readlane(selfpos, sl);                                                     \
          double* dst = ((sp & 0x80) ? Dall + (sp & 0x7f) * kSpTile : (Lto_) + sp * kSpTile) + (off_cd_);            \
          _Pragma("unroll") for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = acc[sl][g];                      \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
      }                                                                                                              \
    } while (0)
    LLD_SP_PUBLISH(__builtin_amdgcn_readfirstlane(P->pub0[w]), Lp0, off_cd0);
    LLD_SP_STAMP(2);
    __syncthreads();                                                 // B2: prologue publish done
    LLD_SP_STAMP(3);
    for (int s = 0; s < T; s++) {
      LLD_SP_STAMP(8 + 6 * s);
      int off_cd = lrow * kCholMStride + lcol, off_ab = lcol * kCholMStride + lrow;
      asm volatile("" : "+v"(off_cd), "+v"(off_ab));
      double* Lp = Lp0 + (s & 1) * kSpPos * kSpTile;
      double* Lnext = Lp0 + ((s + 1) & 1) * kSpPos * kSpTile;
      const int JA = P->cols[s][0], JB = P->cols[s][1];
      // (c) L_IJ = A_IJ L_JJ^-T on the matrix cores; keep it (back substitution) and leave it in the panel buffer (operand of d)
      const unsigned mOwn = __builtin_amdgcn_readfirstlane(P->own[w][s]);           // the next column's panel wavefront forms these
      {
        const unsigned mA = __builtin_amdgcn_readfirstlane(P->cA[w][s]) & ~mOwn, mB = __builtin_amdgcn_readfirstlane(P->cB[w][s]) & ~mOwn;
#pragma unroll
        for (int sl = 0; sl < kSpSlots; sl++) {
          if ((mA | mB) & (1u << sl)) {
            const int sp = __builtin_amdgcn_readlane(selfpos, sl);
            const double* pa = Lp + sp * kSpTile + off_ab;
            const double* pb = Dall + ((mA & (1u << sl)) ? JA : JB) * kSpTile + off_ab;
            v4d c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pb[4 * kk], c, 0, 0, 0);
            acc[sl] = c;
            double* dst = Lp + sp * kSpTile + off_cd;
#pragma unroll
            for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = c[g];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // forward solution of this step's column(s): y_J = L_JJ^-1 y_J (every earlier column's term is in; the factor is in Dall since the last barrier)
#pragma unroll
      for (int c2 = 0; c2 < 2; c2++) {
        const int J = c2 == 0 ? JA : JB;
        if (J != kSpNone && J % kSpTileWaves == w) {
          const double* Li = Dall + J * kSpTile + (lane >> 2) * kCholMStride + 4 * (lane & 3);     // lane = 4 x row of the tile + quarter of the columns
          const double* yj = y + 16 * J + 4 * (lane & 3);
          const double dotv = quad_sum(Li[0] * yj[0] + Li[1] * yj[1] + Li[2] * yj[2] + Li[3] * yj[3]);
          if ((lane & 3) == 0) y[16 * J + (lane >> 2)] = dotv;
        }
      }
      LLD_SP_STAMP(9 + 6 * s);
      __syncthreads();                                               // (c) done
      LLD_SP_STAMP(10 + 6 * s);
      // the tiles the panel wavefronts formed: L into this wavefront's registers (the back substitution reads it there)
#pragma unroll
      for (int sl = 0; sl < kSpSlots; sl++) {
        if (mOwn & (1u << sl)) {
          const int sp = __builtin_amdgcn_readlane(selfpos, sl);
          const double* src = Lp + sp * kSpTile + off_cd;
#pragma unroll
          for (int g = 0; g < 4; g++) acc[sl][g] = src[4 * g * kCholMStride];
        }
      }
      // (d) trailing updates of this step's column(s), then the publishes
      {
        const unsigned mA = __builtin_amdgcn_readfirstlane(P->dA[w][s]), mB = __builtin_amdgcn_readfirstlane(P->dB[w][s]);
        // lane l: the panel-buffer positions of L(I, JA), L(K, JA), L(I, JB), L(K, JB) for slot l's tile (I, K)
        unsigned opnd = 0;
        if (lane < kSpSlots && myI != kSpNone) {
          if (JA != kSpNone) opnd |= (unsigned)P->pos[JA][myI] | ((unsigned)P->pos[JA][myK] << 8);
          if (JB != kSpNone) opnd |= ((unsigned)P->pos[JB][myI] << 16) | ((unsigned)P->pos[JB][myK] << 24);
        }
#pragma unroll
        for (int sl = 0; sl < kSpSlots; sl++) {
          if ((mA | mB) & (1u << sl)) {
            const unsigned o = __builtin_amdgcn_readlane(opnd, sl);
            v4d c = acc[sl];
            if (mA & (1u << sl)) {
              const double* pa = Lp + (o & 0xff) * kSpTile + off_ab;
              const double* pb = Lp + ((o >> 8) & 0xff) * kSpTile + off_ab;
#pragma unroll
              for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk], pb[4 * kk], c, 0, 0, 0);
            }
            if (mB & (1u << sl)) {
              const double* pa = Lp + ((o >> 16) & 0xff) * kSpTile + off_ab;
              const double* pb = Lp + (o >> 24) * kSpTile + off_ab;
#pragma unroll
              for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk], pb[4 * kk], c, 0, 0, 0);
            }
            acc[sl] = c;
          }
          if (sl & 1) __builtin_amdgcn_sched_barrier(0);
        }
        LLD_SP_STAMP(11 + 6 * s);
        LLD_SP_PUBLISH(__builtin_amdgcn_readfirstlane(P->pub[w][s]), Lnext, off_cd);
      }
      // forward substitution of the right-hand side below the panel wavefronts' rows: y_I -= L_IJ y_J, tile rows I = w, w + 6, ...
      {
        const unsigned yA = __builtin_amdgcn_readfirstlane(P->yrows[s][0]), yB = __builtin_amdgcn_readfirstlane(P->yrows[s][1]);
        for (int I = w; I < NT; I += kSpTileWaves) {
          if (!(((yA | yB) >> I) & 1)) continue;
          double dotv = 0.0;                                 // lane = 4 x row of the tile + quarter of the columns
          if ((yA >> I) & 1) {
            const double* pr = Lp + P->pos[JA][I] * kSpTile + (lane >> 2) * kCholMStride + 4 * (lane & 3);
            const double* yj = y + 16 * JA + 4 * (lane & 3);
            dotv += pr[0] * yj[0] + pr[1] * yj[1] + pr[2] * yj[2] + pr[3] * yj[3];
          }
          if ((yB >> I) & 1) {
            const double* pr = Lp + P->pos[JB][I] * kSpTile + (lane >> 2) * kCholMStride + 4 * (lane & 3);
            const double* yj = y + 16 * JB + 4 * (lane & 3);
            dotv += pr[0] * yj[0] + pr[1] * yj[1] + pr[2] * yj[2] + pr[3] * yj[3];
          }
          dotv = quad_sum(dotv);
          if ((lane & 3) == 0) y[16 * I + (lane >> 2)] -= dotv;
        }
      }
      LLD_SP_STAMP(12 + 6 * s);
      __syncthreads();                                               // (d) + lookahead done
      LLD_SP_STAMP(13 + 6 * s);
    }
    LLD_SP_STAMP(4);
    // back substitution: L lives in the register tiles, s_c = sum over the column's tiles of L[i][c] x_i
    for (int s = T - 1; s >= 0; s--) {
      const unsigned mA = __builtin_amdgcn_readfirstlane(P->cA[w][s]), mB = __builtin_amdgcn_readfirstlane(P->cB[w][s]);
      double partA = 0.0, partB = 0.0;
#pragma unroll
      for (int sl = 0; sl < kSpSlots; sl++) {
        if ((mA | mB) & (1u << sl)) {
          const int I = __builtin_amdgcn_readlane(myI, sl);
          double t = 0.0;
#pragma unroll
          for (int g = 0; g < 4; g++) t += acc[sl][g] * x[16 * I + lrow + 4 * g];
          if (mA & (1u << sl)) partA += t; else partB += t;
        }
      }
      if (mA) { partA += __shfl_xor(partA, 16); partA += __shfl_xor(partA, 32); }
      if (mB) { partB += __shfl_xor(partB, 16); partB += __shfl_xor(partB, 32); }
      if (lane < 16) { colsum[w * 16 + lane] = partA; colsum[kSpTileWaves * 16 + w * 16 + lane] = partB; }
      __syncthreads();                                               // column sums of step s complete
      __syncthreads();                                               // x of step s ready
    }
    LLD_SP_STAMP(5);
  }
  // back to S's row order, then the common epilogue
  if (tid < N) { const int R = P->rowmap[tid]; if (R >= 0) xo[R] = x[tid]; }
  __syncthreads();
  const bool ok = *okf != 0.0;
  solve_epilogue(A, W, S, xo, scratch, ok, 0);
  LLD_SP_STAMP(6);
#ifdef LLD_EXPERIMENTS
  if (stamp_lds) {
    long long* dst = A.chol_stamps + ((size_t)W.win_index * kCholStampWaves + wave) * kCholStampSlots;
    for (int i = lane; i < kSpStampSlots; i += 64) dst[i] = (long long)stamp_lds[i];
  }
#endif
}
#undef LLD_SP_STAMP
#undef LLD_SP_PUBLISH

#endif
