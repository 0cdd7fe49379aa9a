// lld_ba_schur.h - Schur complement of the batched bundle adjustment: ba_schur_items* (chunk products), ba_schur_wide, ba_schur_reduce, ba_symmetrize.
// Part of lld_ba_kernels.h (split by kernel family in round 6; no behaviour change): included from there INSIDE namespace lldba, after the shared types and helpers.
// Not a stand-alone header.

// ================================================================== Schur complement
// (1) ba_schur_items: one wavefront per chunk of landmarks that share one set of free cameras.  With
//     Hll + lambda I = L L^T (setLambda + the inverse of block_solver.hpp:391 folded into a Cholesky factor) and Z_a = W_a L^-T,
//     the reference's  Y_a W_b^T = W_a (Hll + lambda I)^-1 W_b^T  (block_solver.hpp:395-428) is Z_a Z_b^T and
//     Y_a b_l = Z_a (L^-1 b_l).  The chunk is swept in sub-batches of up to 64/k landmarks that are staged through LDS:
//       stage    lane (landmark, slot) rebuilds its 6xD Hpl block (points: closed form from pose, point and weight; lines: the
//                stored block), factors Hll + lambda I, and leaves Z (and t = L^-1 b_l, once per landmark) in LDS.  Its global
//                loads run ahead: indices two sub-batches ahead, landmark data one.
//       product  lane (slot pair (a,b), interleave) keeps the WHOLE 6x6 product Z_a Z_b^T of its pair in 36 registers over the
//                chunk: 36 LDS doubles (16-byte reads) per 36*D FMAs.  (An fp64 FMA of a wavefront takes 4 cycles on one of the four
//                SIMDs, the CU's single LDS pipe moves 32 doubles per cycle with ds_read_b128 and 16 with ds_read2_b64: the 2-row
//                blocks this replaces read 24 doubles per 12*D FMAs through ds_read2_b64 and ran at the speed of the LDS pipe.)
//     The chunk's partials are stored once (plain stores).
// (2) ba_schur_reduce: S = blockdiag(Hpp + lambda I) - sum of partials, bschur = b_p - sum c, through a host-built CSR
//     (lower block -> contributing partials).  Only the LOWER block triangle is produced.  No atomics, fixed order.
constexpr int kSchurWideK = 64;            // more free observations of one landmark than this: schur_chunk_wide

// sub-batch geometry shared by host (LDS size) and device: landmarks per sub-batch, LDS doubles
__host__ __device__ inline int schur_nb(int k) {
  int NB = 64 / k; if (NB < 1) NB = 1;
  const int np = k * (k + 1) / 2, units = np < 64 ? np : 64, q = 64 / units;
  if (NB >= q) NB -= NB % q;               // every interleave lane gets the same number of landmarks
  return NB;
}
__host__ __device__ inline int schur_lds_doubles(int k, int D) {
  const int WS = (D == 3) ? 18 : 26;
  return k > kSchurWideK ? k * WS + D + 1 : ((schur_nb(k) * (k * WS + D) + 1) & ~1);
}

// factor + stage one (landmark, slot): Z = W L^-T into zl (6 x D), t = L^-1 b_l into tl
template <int D>
__device__ __forceinline__ void schur_stage_one(bool a, const double* v, double lambda, const double* w, double* zl, double* tl, bool write_t) {
  constexpr int HU = (D == 3) ? 6 : 10;
  double L[D * (D + 1) / 2], idg[D];
  if (!a) {                                                 // inactive landmark (rare): contributes nothing - explicit zeros, so that a
#pragma unroll                                              // non-finite stale block cannot turn into 0 * NaN
    for (int i = 0; i < 6 * D; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(0.0, 0.0);
    if (write_t) {
#pragma unroll
      for (int c = 0; c < D; c++) tl[c] = 0.0;
    }
    return;
  }
  chol_packed<D>(v, lambda, L, idg);
  double z[6 * D];
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < D; c++) {
      double sacc = w[r * D + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= z[r * D + m] * L[c * (c + 1) / 2 + m];
      z[r * D + c] = sacc * idg[c];
    }
#pragma unroll
  for (int i = 0; i < 6 * D; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(z[i], z[i + 1]);
  if (write_t) {
    double t[D];
#pragma unroll
    for (int c = 0; c < D; c++) {
      double sacc = v[HU + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= t[m] * L[c * (c + 1) / 2 + m];
      t[c] = sacc * idg[c];
    }
#pragma unroll
    for (int c = 0; c < D; c++) tl[c] = t[c];
  }
}

// The same for a POINT edge, from the structure of its block (round 5).  W = ws Jc^T Jp = [ [Xc]x G ; G ] with G = ws (A^T A) R
// (point_hpl_closed_iz), hence Z = W L^-T = [ [Xc]x H ; H ] with H = G L^-T: the triangular solve runs on three rows instead of six and the
// three cross products act on H instead of on G - 18 multiply-adds less per (landmark, slot) of the ~200 a staging lane spends.
// `G`: rows g0, g1, g2 of G (G[r * 3 + j]); Xc: the point in the camera frame.
__device__ __forceinline__ void schur_stage_point(bool a, const double* v, double lambda, const double* G, const Vec3& Xc, double* zl, double* tl, bool write_t) {
  double L[6], idg[3];
  if (!a) {
#pragma unroll
    for (int i = 0; i < 18; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(0.0, 0.0);
    if (write_t) { tl[0] = 0.0; tl[1] = 0.0; tl[2] = 0.0; }
    return;
  }
  chol_packed<3>(v, lambda, L, idg);
  double z[18];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double sacc = G[r * 3 + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= z[(3 + r) * 3 + m] * L[c * (c + 1) / 2 + m];
      z[(3 + r) * 3 + c] = sacc * idg[c];
    }
#pragma unroll
  for (int c = 0; c < 3; c++) {                              // Xc x (column c of H)
    const double h0 = z[9 + c], h1 = z[12 + c], h2 = z[15 + c];
    z[c] = Xc.y * h2 - Xc.z * h1;
    z[3 + c] = Xc.z * h0 - Xc.x * h2;
    z[6 + c] = Xc.x * h1 - Xc.y * h0;
  }
#pragma unroll
  for (int i = 0; i < 18; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(z[i], z[i + 1]);
  if (write_t) {
    double t[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double sacc = v[6 + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= t[m] * L[c * (c + 1) / 2 + m];
      t[c] = sacc * idg[c];
    }
    tl[0] = t[0]; tl[1] = t[1]; tl[2] = t[2];
  }
}

// H form of a staged point block (round 6): only H (3 x 3, row major) and Xc go to LDS - 12 doubles instead of the 18 of Z = [ [Xc]x H ; H ];
// the product loop rebuilds the four 3 x 3 blocks of Z_a Z_b^T from M = H_a H_b^T (schur_chunk_wave).  kSchurHS: LDS stride in doubles.
#ifndef LLD_SCHUR_ZFORM
#define LLD_SCHUR_HFORM 1
#endif
constexpr int kSchurHS = 14;
__device__ __forceinline__ void schur_stage_point_h(bool a, const double* v, double lambda, const double* G, const Vec3& Xc, double* zl, double* tl, bool write_t) {
  double L[6], idg[3];
  if (!a) {
#pragma unroll
    for (int i = 0; i < 12; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(0.0, 0.0);
    if (write_t) { tl[0] = 0.0; tl[1] = 0.0; tl[2] = 0.0; }
    return;
  }
  chol_packed<3>(v, lambda, L, idg);
  double h[12];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double sacc = G[r * 3 + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= h[r * 3 + m] * L[c * (c + 1) / 2 + m];
      h[r * 3 + c] = sacc * idg[c];
    }
  h[9] = Xc.x; h[10] = Xc.y; h[11] = Xc.z;
#pragma unroll
  for (int i = 0; i < 12; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(h[i], h[i + 1]);
  if (write_t) {
    double t[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double sacc = v[6 + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= t[m] * L[c * (c + 1) / 2 + m];
      t[c] = sacc * idg[c];
    }
    tl[0] = t[0]; tl[1] = t[1]; tl[2] = t[2];
  }
}

// A landmark with more than kSchurWideK free observations (global BA of a long track): no pipelining and no register accumulators -
// the whole workgroup stages the landmark's k blocks, then thread t adds the products of the pairs t, t + 256, ... into the chunk's
// partials in HBM (one writer per pair, landmarks in order: deterministic).  Such chunks hold a handful of landmarks.
template <int D>
__device__ __forceinline__ void schur_chunk_wide(const BAArrays& A, const BAWin& W, const SChunk& C, double lambda, int cur, double* lds) {
  constexpr int VN = (D == 3) ? 9 : 14, WN = 6 * D, WS = (D == 3) ? 18 : 26;
  const double* __restrict__ Vbase = (D == 3) ? A.pt_V : A.ln_V;
  const uint8_t* __restrict__ act = (D == 3) ? A.pt_active : A.ln_active;
  const int k = C.k, np = k * (k + 1) / 2;
  double* Zl = lds; double* tl = lds + ((k * WS + 1) & ~1);
  for (int t0 = 0; t0 < C.n_lm; t0++) {
    const int g = A.sg_lm[C.lm_off + t0];
    const bool a = act[g] != 0;
    double v[VN];
#pragma unroll
    for (int i = 0; i < VN; i++) v[i] = Vbase[(size_t)g * VN + i];
    __syncthreads();
    for (int sl = threadIdx.x; sl < k; sl += kSchurWideThreads) {
      const int id = A.sg_tab[C.tab_off + (size_t)t0 * k + sl];
      double w[WN];
      if constexpr (D == 3) {
        const Pose Ts = load_cam(A, cur, W.cam_off + A.sg_cams[C.cams_off + sl]);
        point_hpl_closed(W.cam, Ts, quat_rotation(Ts.q), load_pt(A, cur, g), signbit(A.pe_ws[id]), fabs(A.pe_ws[id]), w);
      } else {
#pragma unroll
        for (int i = 0; i < WN; i++) w[i] = A.lo_W[(size_t)id * WN + i];
      }
      schur_stage_one<D>(a, v, lambda, w, Zl + sl * WS, tl, sl == 0);
    }
    __syncthreads();
    for (int pr = threadIdx.x; pr < np; pr += kSchurWideThreads) {
      int sa = 0, rem = pr;
      while (rem >= k - sa) { rem -= k - sa; sa++; }
      const int sb = sa + rem;
      const double* za = Zl + sa * WS; const double* zb = Zl + sb * WS;
      double* dst = A.sp_part + (size_t)(C.part_off + pr) * 36;
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) {
          double sacc = t0 == 0 ? 0.0 : dst[r * 6 + c];
#pragma unroll
          for (int m = 0; m < D; m++) sacc = fma(za[r * D + m], zb[c * D + m], sacc);
          dst[r * 6 + c] = sacc;
        }
      if (sa == sb) {
        double* cd = A.sp_cpart + (size_t)(C.cpart_off + sa) * 6;
        for (int r = 0; r < 6; r++) {
          double sacc = t0 == 0 ? 0.0 : cd[r];
#pragma unroll
          for (int m = 0; m < D; m++) sacc = fma(za[r * D + m], tl[m], sacc);
          cd[r] = sacc;
        }
      }
    }
  }
}

template <int D>
__device__ __forceinline__ void schur_chunk_wave(const BAArrays& A, const BAWin& W, const SChunk& C, double lambda, int cur, double* lds) {
  constexpr int VN = (D == 3) ? 9 : 14, WN = 6 * D;
  // LDS stride of a staged 6xD block: 144 B for points (conflict-free as is); 192 B would put slots 0 and 4 of a line on the
  // same banks, so line blocks are padded to 208 B (still 16-B aligned)
#ifdef LLD_SCHUR_HFORM
  constexpr bool kH = (D == 3);
  constexpr int WS = (D == 3) ? kSchurHS : 26;
#else
  constexpr bool kH = false;
  constexpr int WS = (D == 3) ? 18 : 26;
#endif
  const double* __restrict__ Vbase = (D == 3) ? A.pt_V : A.ln_V;
  const uint8_t* __restrict__ act = (D == 3) ? A.pt_active : A.ln_active;
  const int lane = threadIdx.x;
  const int k = C.k, np = k * (k + 1) / 2;
  const int NB = schur_nb(k);
  double* Zl = lds;
  double* tl = lds + NB * k * WS;
  const int* __restrict__ lm = A.sg_lm + C.lm_off;
  const int* __restrict__ tab = A.sg_tab + C.tab_off;
  // stage lane <-> (landmark ej, slot esl) of a sub-batch
  const int ej = lane / k, esl = lane - ej * k;
  const bool stager = lane < NB * k;
  // the lanes of the per-slot vector pass: slot cslot, interleave ci of cq
  const int cq = (64 / k) < NB ? (64 / k) : NB;
  const int ci = lane / k, cslot = lane - ci * k;
  const bool con = ci < cq;
  double cacc[6];
#pragma unroll
  for (int i = 0; i < 6; i++) cacc[i] = 0.0;
  for (int pass0 = 0; pass0 < np; pass0 += 64) {
    const int units = (np - pass0) < 64 ? (np - pass0) : 64;        // slot pairs of this pass
    const int q = 64 / units;                                       // landmarks worked on at a time
    const int pl = lane % units, qq = lane / units;
    const bool on = qq < q;
    int sa = 0, rem = pass0 + pl;
    while (rem >= k - sa) { rem -= k - sa; sa++; }
    const int sb = sa + rem;
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0.0;
    // The staging runs one sub-batch AHEAD of its loads' latency: the camera pose of a lane's slot is loop-invariant, the landmark /
    // edge indices are fetched two sub-batches ahead and the landmark data one ahead, so the two dependent HBM/L2 round trips
    // (index -> data) of a sub-batch overlap the block products of the previous one.  Bytes stay as loaded (a_n, fl_n): turning
    // them into flags where they are fetched would wait for the loads right there.
    Pose T; Mat3 Rt;
    if constexpr (D == 3) { if (stager) { T = load_cam(A, cur, W.cam_off + A.sg_cams[C.cams_off + esl]); Rt = quat_rotation(T.q); } }
    int g_n = 0, id_n = 0, g_nn = 0, id_nn = 0, a_n = 0;
    double v_n[VN];
    double ws_n = 0.0; Vec3 X_n = vec3(0, 0, 1);
    // Every lane issues every load, with the landmark index clamped into the chunk (a lane past the end or outside the staging range
    // re-reads the last landmark and never uses it): loads under a divergent `if` cannot be counted and the compiler waits for
    // vmcnt(0) at the first use of any loaded value (14 such waits in this kernel before, 7 now; the loads are a whole sub-batch of
    // arithmetic ahead either way, so the launch time did not move).
    auto fetch_idx = [&](int t0, int& g, int& id) {
      const int tj = min(t0 + ej, C.n_lm - 1);
      g = lm[tj]; id = tab[(size_t)tj * k + esl];
    };
    auto fetch_data = [&](int t0, int g, int id) {
      a_n = act[g];
      const double* V = Vbase + (size_t)g * VN;
#pragma unroll
      for (int i = 0; i < VN; i++) v_n[i] = V[i];
      if constexpr (D == 3) { ws_n = A.pe_ws[id]; X_n = load_pt(A, cur, g); }      // (the stereo flag rides in the weight's sign: no gather of the edge's flag byte)      // (the stereo flag rides in the weight's sign: no gather of the edge's flag byte)
    };
    fetch_idx(0, g_n, id_n);
    fetch_idx(NB, g_nn, id_nn);
    fetch_data(0, g_n, id_n);
    for (int t0 = 0; t0 < C.n_lm; t0 += NB) {
      const int nb = (C.n_lm - t0) < NB ? (C.n_lm - t0) : NB;
      // this sub-batch's operands (arrived while the previous products ran) -> locals; then put the next loads in flight
      double w[WN], v[VN];
#pragma unroll
      for (int i = 0; i < VN; i++) v[i] = v_n[i];
      const double ws = ws_n; const Vec3 X = X_n; const int a_raw = a_n;
      const int id_cur = id_n;
      g_n = g_nn; id_n = id_nn;
      fetch_data(t0 + NB, g_n, id_n);
      fetch_idx(t0 + 2 * NB, g_nn, id_nn);
      if constexpr (D == 4) {
        // line observation: the summed 6x4 block was stored by the linearisation (fetched here, not a sub-batch ahead: holding two
        // of them would halve the occupancy); by every lane, like the prefetch (id_cur is a valid observation in all of them)
        const double* Wg = A.lo_W + (size_t)id_cur * WN;
#pragma unroll
        for (int i = 0; i < WN; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(Wg + i); w[i] = t2.x; w[i + 1] = t2.y; }
      }
      __syncthreads();                                      // the previous sub-batch has been consumed
      if (stager && ej < nb) {
        // point edge: the Hpl block is a function of the linearisation-point pose, point and one weight
        if constexpr (D == 3) {
          const Vec3 Xc = mat_mul(Rt, X) + T.t;
          double G[9];
          point_g_closed_iz(W.cam, Xc, rcp_nr(Xc.z), Rt, signbit(ws), fabs(ws), G);
          if constexpr (kH) schur_stage_point_h(a_raw != 0, v, lambda, G, Xc, Zl + lane * WS, tl + ej * D, esl == 0);
          else schur_stage_point(a_raw != 0, v, lambda, G, Xc, Zl + lane * WS, tl + ej * D, esl == 0);
        } else schur_stage_one<D>(a_raw != 0, v, lambda, w, Zl + lane * WS, tl + ej * D, esl == 0);
      }
      __syncthreads();
      if (on) {
        for (int j = qq; j < nb; j += q) {
          const double* za = Zl + (j * k + sa) * WS;
          const double* zb = Zl + (j * k + sb) * WS;
          if constexpr (kH) {
            // Z_a Z_b^T = [ a M b^T, a M ; M b^T, M ] with M = H_a H_b^T, a = [Xa]x, b = [Xb]x: 27 + 9 + 18 + 9 + 18 + 18 = 99 operations against the
            // 108 FMAs of the full 6x3 . 3x6 product, and 24 LDS doubles per (pair, landmark) instead of 36.  (Reading the next landmark's operands
            // ahead of this one's products by hand measured 1 % slower: tools/experiments/r06_schur_hform_prefetch.patch.)
            double ha[12], hb[12];
#pragma unroll
            for (int i = 0; i < 12; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(zb + i); hb[i] = t2.x; hb[i + 1] = t2.y; }
#pragma unroll
            for (int i = 0; i < 12; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(za + i); ha[i] = t2.x; ha[i + 1] = t2.y; }
            double M[9], P[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int c = 0; c < 3; c++) M[r * 3 + c] = fma(ha[r * 3 + 2], hb[c * 3 + 2], fma(ha[r * 3 + 1], hb[c * 3 + 1], ha[r * 3] * hb[c * 3]));
            const double xa = ha[9], ya = ha[10], za_ = ha[11], xb = hb[9], yb = hb[10], zb_ = hb[11];
#pragma unroll
            for (int r = 0; r < 3; r++) {                          // P = M b^T
              const double m0 = M[r * 3], m1 = M[r * 3 + 1], m2 = M[r * 3 + 2];
              P[r * 3] = fma(yb, m2, -(zb_ * m1)); P[r * 3 + 1] = fma(zb_, m0, -(xb * m2)); P[r * 3 + 2] = fma(xb, m1, -(yb * m0));
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
              acc[18 + 3 + c] += M[c]; acc[24 + 3 + c] += M[3 + c]; acc[30 + 3 + c] += M[6 + c];        // lower right: M
              acc[18 + c] += P[c]; acc[24 + c] += P[3 + c]; acc[30 + c] += P[6 + c];                    // lower left: M b^T
              // upper right: a M, upper left: a P   (row 0 = ya * row 2 - za * row 1, row 1 = za * row 0 - xa * row 2, row 2 = xa * row 1 - ya * row 0)
              acc[3 + c] = fma(ya, M[6 + c], fma(-za_, M[3 + c], acc[3 + c]));
              acc[6 + 3 + c] = fma(za_, M[c], fma(-xa, M[6 + c], acc[6 + 3 + c]));
              acc[12 + 3 + c] = fma(xa, M[3 + c], fma(-ya, M[c], acc[12 + 3 + c]));
              acc[c] = fma(ya, P[6 + c], fma(-za_, P[3 + c], acc[c]));
              acc[6 + c] = fma(za_, P[c], fma(-xa, P[6 + c], acc[6 + c]));
              acc[12 + c] = fma(xa, P[3 + c], fma(-ya, P[c], acc[12 + c]));
            }
            continue;
          }
          double b[WN];
#pragma unroll
          for (int i = 0; i < WN; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(zb + i); b[i] = t2.x; b[i + 1] = t2.y; }
#pragma unroll
          for (int rp = 0; rp < 3; rp++) {                   // two rows of Z_a at a time: 16-byte LDS reads
            double a2[2 * D];
#pragma unroll
            for (int i = 0; i < 2 * D; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(za + rp * 2 * D + i); a2[i] = t2.x; a2[i + 1] = t2.y; }
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
              const int r = 2 * rp + rr;
#pragma unroll
              for (int cc = 0; cc < 6; cc++) {
                double s0 = acc[r * 6 + cc];
#pragma unroll
                for (int m = 0; m < D; m++) s0 = fma(a2[rr * D + m], b[cc * D + m], s0);
                acc[r * 6 + cc] = s0;
              }
            }
          }
        }
      }
      // Y_a b_l = Z_a (L^-1 b_l), one 6-vector per slot: lane <-> (slot, interleave) in a pass of its own (round 5).  Inside the product loop
      // every lane carried these 6 x D multiply-adds per landmark although only the k diagonal pairs of the k (k + 1) / 2 keep them.
      if (pass0 == 0 && con) {
        for (int j = ci; j < nb; j += cq) {
          const double* za = Zl + (j * k + cslot) * WS;
          if constexpr (kH) {                                      // Z t = [ Xc x (H t) ; H t ]
            double ha[12];
#pragma unroll
            for (int i = 0; i < 12; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(za + i); ha[i] = t2.x; ha[i + 1] = t2.y; }
            const double t0_ = tl[j * 3], t1_ = tl[j * 3 + 1], t2_ = tl[j * 3 + 2];
            const double h0 = fma(ha[2], t2_, fma(ha[1], t1_, ha[0] * t0_)), h1 = fma(ha[5], t2_, fma(ha[4], t1_, ha[3] * t0_)), h2 = fma(ha[8], t2_, fma(ha[7], t1_, ha[6] * t0_));
            cacc[3] += h0; cacc[4] += h1; cacc[5] += h2;
            cacc[0] = fma(ha[10], h2, fma(-ha[11], h1, cacc[0]));
            cacc[1] = fma(ha[11], h0, fma(-ha[9], h2, cacc[1]));
            cacc[2] = fma(ha[9], h1, fma(-ha[10], h0, cacc[2]));
            continue;
          }
          double a[WN], tv[D];
#pragma unroll
          for (int i = 0; i < WN; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(za + i); a[i] = t2.x; a[i + 1] = t2.y; }
#pragma unroll
          for (int m = 0; m < D; m++) tv[m] = tl[j * D + m];
#pragma unroll
          for (int r = 0; r < 6; r++) {
            double s1 = cacc[r];
#pragma unroll
            for (int m = 0; m < D; m++) s1 = fma(a[r * D + m], tv[m], s1);
            cacc[r] = s1;
          }
        }
      }
    }
    // sum over the interleave (lanes qq * units + pl): a fixed shuffle tree, result in the lanes qq == 0
    for (int sft = 1; sft < q; sft <<= 1) {
      const bool take = (qq % (2 * sft)) == 0 && qq + sft < q;
#pragma unroll
      for (int i = 0; i < 36; i++) { const double o = __shfl_down(acc[i], sft * units); if (take) acc[i] += o; }
    }
    if (on && qq == 0) {
      // plain stores of the chunk's partial products; ba_schur_reduce sums them into S in a fixed order (no atomics)
      double* dst = A.sp_part + (size_t)(C.part_off + pass0 + pl) * 36;
#pragma unroll
      for (int i = 0; i < 36; i += 2) *reinterpret_cast<double2*>(dst + i) = make_double2(acc[i], acc[i + 1]);
    }
  }
  // the slot vectors: sum over the interleave (lanes cslot + k * ci), a fixed shuffle tree, result in the lanes ci == 0
  for (int sft = 1; sft < cq; sft <<= 1) {
    const bool take = (ci % (2 * sft)) == 0 && ci + sft < cq;
#pragma unroll
    for (int i = 0; i < 6; i++) { const double o = __shfl_down(cacc[i], sft * k); if (take) cacc[i] += o; }
  }
  if (ci == 0) {                                            // (lanes 0 .. k - 1)
    double* cd = A.sp_cpart + (size_t)(C.cpart_off + cslot) * 6;
#pragma unroll
    for (int i = 0; i < 6; i += 2) *reinterpret_cast<double2*>(cd + i) = make_double2(cacc[i], cacc[i + 1]);
  }
}

// grid (max chunks of this landmark type, nW), block 64 = one wavefront per chunk; dynamic LDS sized by the host.
template <int D>
__global__ __launch_bounds__(kSchurThreads) void ba_schur_items_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  const int first = (D == 3) ? W.item_off : W.item_off + W.n_items_pt;
  const int count = (D == 3) ? W.n_items_pt : W.n_items - W.n_items_pt;
  if ((int)blockIdx.x >= count) return;
  const SChunk C = A.sg_chunks[first + blockIdx.x];
  if (C.k <= kSchurWideK) schur_chunk_wave<D>(A, W, C, S.lambda, S.cur, lds);   // (wider chunks: ba_schur_wide_kernel)
}

// Point and line chunks in ONE launch: grid (nW * (n_pt_blocks + max line chunks)), see the dispatch order below; chunks are stored heaviest first (stage_chunks).  The line chunks fill the tail of the point
// chunks instead of waiting for it - for a single window the two kernels were two dependent 17 us launches on an otherwise idle GPU.
__global__ __launch_bounds__(kSchurThreads) void ba_schur_items_both_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st, int n_pt_blocks, int win_tile, int nrow, int nch) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // dispatch order (x fastest) -> (window row, chunk): tiles of win_tile windows, inside a tile the chunk index runs slowest.  Workgroups go to
  // the eight XCDs round-robin, so with a tile of 8 (any multiple of 8) ALL chunks of a window run on ONE XCD, close together in time: the sectors
  // of the landmark arrays that several of them touch (a 128-byte line of positions or V serves landmarks of different camera sets) are fetched
  // into one L2 once instead of into up to eight.  FETCH_SIZE of this kernel per launch of 256 windows, tools/experiments/exp_schur_tile.sh:
  // windows one after the other 671 k KiB, tile 8: 360 k, 16: 373 k, 32: 389 k, 64: 504 k, all 256: 766 k; the time is within noise from 8 to 32
  // and 3 - 4 % better than either extreme.  A window's chunks are stored heaviest first, so every tile drains on its light chunks.
  const int lin = (int)blockIdx.x;                             // one-dimensional grid of nrow x nch workgroups (a map of thousands of keyframes has more chunks than gridDim.y may be)
  const int tile = lin / (win_tile * nch), row0 = tile * win_tile, tsz = min(win_tile, nrow - row0), rem = lin - tile * win_tile * nch;
  const int ci = rem / tsz, row = row0 + rem - ci * tsz;
  const int wrow = LLD_ROW_WINDOW(A, st, row);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  if (ci < n_pt_blocks) {
    if (ci >= W.n_items_pt) return;
    const SChunk C = A.sg_chunks[W.item_off + ci];
    if (C.k <= kSchurWideK) schur_chunk_wave<3>(A, W, C, S.lambda, S.cur, lds);
  } else {
    const int i = ci - n_pt_blocks;
    if (i >= W.n_items - W.n_items_pt) return;
    const SChunk C = A.sg_chunks[W.item_off + W.n_items_pt + i];
    if (C.k <= kSchurWideK) schur_chunk_wave<4>(A, W, C, S.lambda, S.cur, lds);
  }
}

// The chunks the kernels above skip (a landmark with more than kSchurWideK free observations); launched only for batches that
// have one.  grid (max chunks, nW) over all chunks of a window, block kSchurThreads.
__global__ __launch_bounds__(kSchurWideThreads) void ba_schur_wide_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN || (int)blockIdx.x >= W.n_items) return;
  const SChunk C = A.sg_chunks[W.item_off + blockIdx.x];
  if (C.k <= kSchurWideK) return;
  if (C.D == 3) schur_chunk_wide<3>(A, W, C, S.lambda, S.cur, lds);
  else schur_chunk_wide<4>(A, W, C, S.lambda, S.cur, lds);
}

// grid (ceil(nblk_max * 6 / 256) + 2, nW): lane <-> one row of one lower 6x6 block of S.  S_ij = [i == j](Hpp_i + lambda I)
// - sum of the chunk partials listed for the block (fixed order -> deterministic), written once with a plain store.
// blk_src = part_index * 4 + mode; mode 0: partial is Y_a W_b^T for cameras a < b -> transposed into block (b, a);
// mode 1: same observation on the diagonal; mode 2: two observations by one camera -> P + P^T.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void ba_schur_reduce_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  const int nf = W.n_free, n = 6 * nf, nblk = nf * (nf + 1) / 2;
  // the LAST TWO workgroups of a window do the right-hand side (half of the rows each), the others the blocks: for a single window both are
  // chains of dependent cross-XCD loads and must not queue behind one another
  const int rhs_part = (int)gridDim.x - 1 - (int)blockIdx.x;   // 0 / 1: a right-hand-side workgroup
  const bool rhs_block = rhs_part < 2;
  // lane <-> one row of one lower 6x6 block, blocks in blk_perm order (longest partial lists first: the lanes of one wavefront walk lists of
  // one length).  The wavefronts of that order are dealt round-robin to the window's workgroups, so that the few long-list wavefronts of a
  // window pull their partials through different CUs (a single window: all of them in one workgroup cost 5 us per launch).
  const int nslot6 = (A.s_skip_empty ? W.n_blk_nz : nblk) * 6, nwg = ((nslot6 + 63) / 64 + 3) / 4;
  const int idx = (((int)threadIdx.x >> 6) * nwg + (int)blockIdx.x) * 64 + ((int)threadIdx.x & 63);
  if (!rhs_block && (int)blockIdx.x < nwg && idx < nslot6) {
    const int slot = idx / 6, rr = idx - slot * 6;
    const int blk = A.blk_perm[W.blk_csr_off + slot];
    int i = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= blk) i++;
    while (i * (i + 1) / 2 > blk) i--;
    const int j = blk - i * (i + 1) / 2;
    double v[6] = {0, 0, 0, 0, 0, 0};
    if (i == j) {
      const double* Hp = A.Hpp + ((size_t)W.hpp_off + i) * 21;
#pragma unroll
      for (int cc = 0; cc < 6; cc++) {
        const int lo = rr < cc ? rr : cc, hi = rr < cc ? cc : rr;
        v[cc] = Hp[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
      }
      v[rr] += S.lambda;
    }
    const int* bst = A.blk_start + W.blk_csr_off;
    const int q0 = bst[blk], q1 = bst[blk + 1];
    // A diagonal block collects one partial from every chunk that sees its camera (~50), and each list entry is a chain of two
    // dependent loads (index -> partial): four entries are kept in flight and the three modes are folded into weights (row part
    // w_r, column part w_c in {0,1}) so that the loads do not sit behind a branch.  0*x + y is exact and mode 2 keeps its
    // P + P^T order, so the result is bit-identical to the entry-by-entry loop.
    // the indices of the next four entries are fetched while the partials of the current four are in flight (two dependent
    // round trips per group otherwise; for a single window every one of them leaves the XCD)
    int nxt[4];
#pragma unroll
    for (int uu = 0; uu < 4; uu++) nxt[uu] = (q0 + uu < q1) ? A.blk_src[q0 + uu] : -1;
    for (int q = q0; q < q1; q += 4) {
      int src[4];
#pragma unroll
      for (int uu = 0; uu < 4; uu++) src[uu] = nxt[uu];
      // ONE strided load per entry: the lane's row of the partial (modes 1, 2) or its column (mode 0: the transposed block) - round 4;
      // before, every entry fetched both (12 doubles, 96 registers in flight, three wavefronts per SIMD for a kernel that only waits)
      double pv[4][6];
#pragma unroll
      for (int uu = 0; uu < 4; uu++) {
        const int sidx = src[uu] < 0 ? 0 : src[uu];
        const bool col = (sidx & 3) == 0;
        const double* P = A.sp_part + (size_t)(sidx >> 2) * 36 + (col ? rr : rr * 6);
        const int stp = col ? 6 : 1;
#pragma unroll
        for (int cc = 0; cc < 6; cc++) pv[uu][cc] = P[cc * stp];
      }
#pragma unroll
      for (int uu = 0; uu < 4; uu++) nxt[uu] = (q + 4 + uu < q1) ? A.blk_src[q + 4 + uu] : -1;
#pragma unroll
      for (int uu = 0; uu < 4; uu++) {
        if (src[uu] >= 0) {
#pragma unroll
          for (int cc = 0; cc < 6; cc++) v[cc] -= pv[uu][cc];
          if ((src[uu] & 3) == 2) {                        // two observations of one landmark by the same camera: P + P^T (no local-BA window has one)
            const double* P = A.sp_part + (size_t)(src[uu] >> 2) * 36;
#pragma unroll
            for (int cc = 0; cc < 6; cc++) v[cc] -= P[cc * 6 + rr];
          }
        }
      }
    }
    double* dst = A.S + W.S_off + (size_t)(6 * i + rr) * n + 6 * j;
#pragma unroll
    for (int cc = 0; cc < 6; cc += 2) *reinterpret_cast<double2*>(dst + cc) = make_double2(v[cc], v[cc + 1]);
  }
  if (rhs_block) {
    const int* cst = A.cam_start + W.cam_csr_off;
    const int half = (n + 1) / 2, t_end = rhs_part == 0 ? half : n;
    for (int t = (rhs_part == 0 ? 0 : half) + (int)threadIdx.x; t < t_end; t += 256) {
      const int c = t / 6, r = t - c * 6;
      double v = A.bp[(size_t)W.hpp_off * 6 + t];
      // a camera appears in ~50 chunks and every list entry is two dependent loads (index -> partial): sixteen entries are kept in
      // flight; they are still subtracted one by one in list order, so the sum is bit-identical to the plain loop
      const int q0 = cst[c], q1 = cst[c + 1];
      for (int q = q0; q < q1; q += 16) {
        int src[16]; double pv[16];
#pragma unroll
        for (int uu = 0; uu < 16; uu++) src[uu] = (q + uu < q1) ? A.cam_src[q + uu] : -1;
#pragma unroll
        for (int uu = 0; uu < 16; uu++) pv[uu] = (src[uu] >= 0) ? A.sp_cpart[(size_t)src[uu] * 6 + r] : 0.0;
#pragma unroll
        for (int uu = 0; uu < 16; uu++) if (src[uu] >= 0) v -= pv[uu];
      }
      A.bschur[W.x_off + t] = v;
    }
  }
}

// PCG only: mirror the lower block triangle into the upper one (the column-wise matvec wants the full matrix).  grid (16, nW)
__global__ __launch_bounds__(256) void ba_symmetrize_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  const BAWin W = wins[blockIdx.y];
  if (st[blockIdx.y].phase != PH_RUN) return;
  const int n = 6 * W.n_free;
  double* Sg = A.S + W.S_off;
  // 64-bit element index: n * n passes 2^31 from 7724 free cameras on (n = 46344), and the limit is 8192
  const long long total = (long long)n * n, stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int row = (int)(i / n), col = (int)(i - (long long)row * n);
    if (col / 6 > row / 6) Sg[i] = Sg[(size_t)col * n + row];
  }
}

// Shared tail of the reduced-system solvers: publish x_p, apply VertexSE3Expmap::oplusImpl to the free cameras (trial
// buffer), leave sum x (lambda x + b) of the camera part for computeScale (optimization_algorithm_levenberg.cpp:182-189).
__device__ __forceinline__ void solve_epilogue(const BAArrays& A, const BAWin& W, BAState& S, const double* x, double* scratch, bool ok, int iters) {
  const int tid = threadIdx.x, nf = W.n_free, n = 6 * nf;
  const double lambda = S.lambda;
  const double* bpv = A.bp + (size_t)W.hpp_off * 6;
  double sc = 0.0;
  if (tid < n) { A.xp[W.x_off + tid] = x[tid]; sc = x[tid] * (lambda * x[tid] + bpv[tid]); }
  const double sc_t = block_sum(sc, scratch);
  const int cur = S.cur, nxt = cur ^ 1;
  if (tid < W.n_cams) {
    const Pose T = load_cam(A, cur, W.cam_off + tid);
    Pose Tn = T;
    if (tid < nf) Tn = pose_oplus(T, x + tid * 6);
    pose_store(Tn, A.cam_qt + ((size_t)nxt * A.NC + W.cam_off + tid) * 7);
  }
  if (tid == 0) { S.scale_cam = sc_t; S.pcg_ok = ok ? 1 : 0; S.pcg_iterations += iters; }
}

