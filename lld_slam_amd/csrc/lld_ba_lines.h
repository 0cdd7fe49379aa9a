// lld_ba_lines.h - Line landmark kernels of the batched bundle adjustment: linearisation (ba_linearize_ln_*, ba_linearize_both) and back-substitution + trial errors (ba_backsub_ln_*), one lane per (line, KF) OBSERVATION.
// Part of lld_ba_kernels.h (split by kernel family in round 6; no behaviour change): included from there INSIDE namespace lldba, after the shared types and helpers.
// Not a stand-alone header.

// ================================================================== line landmarks: one lane per (line, KF) OBSERVATION
// Same scheme as the point kernels with the observation as the unit: a lane linearises the left and (if present) right image
// edge of its observation and keeps their summed Hpl block; Hll/b_l (14 values) are combined over the line's lanes.
struct LineGeom { Vec3 c0, c1, X1, X2; double alpha; };
__device__ __forceinline__ LineGeom line_geom(const LineQ& L) {
  const Mat3 Rl = line_rotation_t<true>(L);
  LineGeom G; G.c0 = mat_col(Rl, 0); G.c1 = mat_col(Rl, 1); G.alpha = L.alpha; G.X1 = L.alpha * G.c1; G.X2 = G.X1 + G.c0;
  return G;
}
// The two image edges of one (line, KF) observation as loaded: flags and (xs, ys, xe, ye, info) of the left and right slot.  Loaded for
// both slots at once and before anything is decided on them: one memory round trip per observation instead of one per slot behind a
// branch on the slot's flags (b_x needs no load: it is 0 for the left and CamK::bx_right for the right slot).
struct LnObsIn { uint8_t fl[2]; double xs[2], ys[2], xe[2], ye[2], s[2]; };
template <int kPk>
__device__ __forceinline__ void line_obs_load(const BAArrays& A, int o, LnObsIn& I) {
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int e = 2 * o + side;
    I.fl[side] = A.le_flags[e];
    const LnSeg g = ln_seg_of<kPk>(A, e);
    I.xs[side] = g.xs; I.ys[side] = g.ys; I.xe[side] = g.xe; I.ye[side] = g.ye;
  }
  if (obs_packed<kPk>(A)) { const unsigned oc = A.lo_oct[o]; I.s[0] = A.ln_info[oc & 255u]; I.s[1] = A.ln_info[oc >> 8]; }
  else { I.s[0] = A.le_s[2 * o]; I.s[1] = A.le_s[2 * o + 1]; }
}
// linearise one observation: hb (10 + 4) and the summed 6x4 Hpl block; returns the robust cost of its active edges
__device__ __forceinline__ double line_obs_linearize(const BAArrays& A, const BAWin& W, const Pose& T, int o, int c, const LineGeom& G, const LnObsIn& I,
                                                     double* hb, double* acc_lds) {
  const bool free_cam = c < W.n_free;
  double Wo[24], Jc0[12], r0[2] = {0.0, 0.0}, ws0 = 0.0;
#pragma unroll
  for (int i = 0; i < 24; i++) Wo[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 12; i++) Jc0[i] = 0.0;
  double chi = 0.0;
  const Mat3 Rc = quat_rotation(T.q);
  const Vec3 X1m = pose_map(T, G.X1), X2m = pose_map(T, G.X2);
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int e = 2 * o + side;
    const uint8_t fl = I.fl[side];
    if (!(fl & EF_VALID) || (fl & EF_LEVEL1)) continue;
    double r[2]; LineAdj adj;
    line_residual_t<true>(W.cam, side == 1 ? W.cam.bx_right : 0.0, X1m, X2m, I.xs[side], I.ys[side], I.xe[side], I.ye[side], r, &adj);
    const double s = I.s[side];
    const double c2 = chi2_of(r, 2, s);
    A.le_chi2[e] = c2;
    double w = 1.0, rho0 = c2;
    if (fl & EF_ROBUST) rho0 = huber_nr(c2, (fl & EF_PAIRSTEREO) ? W.th_ln_stereo : W.th_ln_mono, &w);
    chi += rho0;
    const double ws = w * s;
    double Jc[12], Jl[8];
    line_jac_pose(adj, X1m, X2m, Jc);
    line_jac_line(adj, Rc, G.c0, G.c1, G.alpha, Jl);
    int k = 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
      hb[10 + a] -= ws * (Jl[a] * r[0] + Jl[4 + a] * r[1]);
#pragma unroll
      for (int d = a; d < 4; d++) hb[k++] += ws * (Jl[a] * Jl[d] + Jl[4 + a] * Jl[4 + d]);
    }
    if (free_cam) {
#pragma unroll
      for (int rr = 0; rr < 6; rr++)
#pragma unroll
        for (int a = 0; a < 4; a++) Wo[rr * 4 + a] += ws * (Jc[rr] * Jl[a] + Jc[6 + rr] * Jl[4 + a]);
      // the camera block of BOTH image edges goes to the accumulators in one pass of LDS atomics (8 ... 24 CU clocks each,
      // tools/microbench/lds_ops.hip): the left edge only keeps its Jacobian, the right edge adds the sum
      if (side == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) Jc0[i] = Jc[i];
        r0[0] = r[0]; r0[1] = r[1]; ws0 = ws;
      } else {
        double* ac = acc_lds + c * 27;
        int kk = 0;
#pragma unroll
        for (int rr = 0; rr < 6; rr++) {
          atomicAdd(&ac[21 + rr], -(ws0 * (Jc0[rr] * r0[0] + Jc0[6 + rr] * r0[1]) + ws * (Jc[rr] * r[0] + Jc[6 + rr] * r[1])));
#pragma unroll
          for (int cc = rr; cc < 6; cc++)
            atomicAdd(&ac[kk++], ws0 * (Jc0[rr] * Jc0[cc] + Jc0[6 + rr] * Jc0[6 + cc]) + ws * (Jc[rr] * Jc[cc] + Jc[6 + rr] * Jc[6 + cc]));
        }
        ws0 = 0.0;                                           // added
      }
    }
  }
  if (free_cam) {
    double* Wb = A.lo_W + (size_t)o * 24;
#pragma unroll
    for (int i = 0; i < 24; i += 2) *reinterpret_cast<double2*>(Wb + i) = make_double2(Wo[i], Wo[i + 1]);
    if (ws0 != 0.0) {                                        // a left edge without an active right edge
      double* ac = acc_lds + c * 27;
      int kk = 0;
#pragma unroll
      for (int rr = 0; rr < 6; rr++) {
        atomicAdd(&ac[21 + rr], -ws0 * (Jc0[rr] * r0[0] + Jc0[6 + rr] * r0[1]));
#pragma unroll
        for (int cc = rr; cc < 6; cc++) atomicAdd(&ac[kk++], ws0 * (Jc0[rr] * Jc0[cc] + Jc0[6 + rr] * Jc0[6 + cc]));
      }
    }
  }
  return chi;
}

// grid (nl_ln, nW), block 512 = 8 wavefronts, BAWin::rounds tasks per wavefront; dynamic LDS: kAccCopies*n_free_max*27 doubles + 8 scratch.
template <bool kBig, int kPk>
__device__ __forceinline__ void ba_linearize_ln_body(const BAArrays& A, const BAWin* __restrict__ wins, BAState* __restrict__ st, const int bx) {   // bx: the workgroup's index along x
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN || !S.need_lin) return;
  if ((int)bx >= W.nl_ln) return;
  const int nacc = W.n_free * 27;
  double* acc_all = kBig ? A.hpp_part + W.hpart_off : lds;   // kAccCopies x [n_free][21 Hpp upper + 6 bp]; kBig: see ba_linearize_pt_body
  const int copies = W.acc_copies[1];
  double* scratch = kBig ? lds : lds + copies * nacc;
  double* acc = acc_all;
  const int cur = S.cur;
  double* cams_l = scratch + 8;                              // [n_cams][7] poses of the linearisation point
  const double* cams = kBig ? A.cam_qt + ((size_t)cur * A.NC + W.cam_off) * 7 : cams_l;
  const int nthr = blockDim.x, nwv = W.lin_waves[1];
  if (!kBig) {
    for (int i = threadIdx.x; i < copies * nacc; i += nthr) acc_all[i] = 0.0;
    acc = acc_all + (W.det ? (int)(threadIdx.x >> 6) : (int)((threadIdx.x >> 3) & (copies - 1))) * nacc;      // see ba_linearize_pt_body
    for (int i = threadIdx.x; i < W.n_cams * 7; i += nthr) cams_l[i] = A.cam_qt[((size_t)cur * A.NC + W.cam_off) * 7 + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double chi = 0.0, maxd = 0.0;
  for (int rnd = 0; rnd < W.rounds[1]; rnd++) {
    const int ti = (bx * W.rounds[1] + rnd) * nwv + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
    if (ti >= W.n_ltasks) break;
    const PTask T = A.ltasks[W.ltask_off + ti];
    double hb[14];
#pragma unroll
    for (int i = 0; i < 14; i++) hb[i] = 0.0;
    if (T.nl > 1) {
      // two dependent memory levels only: (1) everything addressed by the observation - line index, camera, both edge slots -
      // (2) the line's state; the camera pose comes from the workgroup's LDS copy
      const bool has = lane < T.ne;
      const int o = T.e0 + (has ? lane : 0);
      int c, l_raw;
      ln_cam_lm_of<kPk>(A, o, T.l0, c, l_raw);
      LnObsIn I;
      line_obs_load<kPk>(A, o, I);
      const int l = has ? l_raw : -1 - lane;
      const int g = W.ln_off + (has ? l : T.l0);
      const LineQ Lq = load_ln(A, cur, g);
      const bool lm_act = has && A.ln_active[g];
      if (lm_act) chi += line_obs_linearize(A, W, pose_load(cams + c * 7), o, c, line_geom(Lq), I, hb, acc);
      seg_sum<14>(hb, l, lane, T.ms);
      if (lm_act && o == A.ln_obs_start[g]) {
        double* V = A.ln_V + (size_t)g * 14;
#pragma unroll
        for (int i = 0; i < 14; i++) V[i] = hb[i];
        maxd = fmax(maxd, fmax(fmax(fabs(hb[0]), fabs(hb[4])), fmax(fabs(hb[7]), fabs(hb[9]))));
      }
    } else {
      const int g = W.ln_off + T.l0;
      if (A.ln_active[g]) {
        const LineGeom G = line_geom(load_ln(A, cur, g));
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int o = T.e0 + sidx, c = ln_cam_of<kPk>(A, o);
          LnObsIn I;
          line_obs_load<kPk>(A, o, I);
          chi += line_obs_linearize(A, W, pose_load(cams + c * 7), o, c, G, I, hb, acc);
        }
        wave_sum_n<14>(hb);
        if (lane == 0) {
          double* V = A.ln_V + (size_t)g * 14;
#pragma unroll
          for (int i = 0; i < 14; i++) V[i] = hb[i];
          maxd = fmax(maxd, fmax(fmax(fabs(hb[0]), fabs(hb[4])), fmax(fabs(hb[7]), fabs(hb[9]))));
        }
      }
    }
  }
  const double chi_t = block_sum(chi, scratch);
  const double max_t = block_max(maxd, scratch);
  if (threadIdx.x == 0) {
    A.chi_part[W.part_off + W.nl_pt + bx] = chi_t;
    atomicMax(&S.maxdiag_bits, (unsigned long long)__double_as_longlong(max_t));
  }
  __syncthreads();
  if (kBig) return;
  // plain stores of this workgroup's camera partials; ba_hpp_reduce sums them in a fixed order (no global atomics)
  double* dst = A.hpp_part + W.hpart_off + (size_t)(W.nl_pt + bx) * nacc;
  for (int i = threadIdx.x; i < nacc; i += nthr) {
    double v = 0.0;
    for (int q = 0; q < copies; q++) v += acc_all[q * nacc + i];
    dst[i] = v;
  }
}
__global__ __launch_bounds__(kLinThreads) void ba_linearize_ln_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_ln_body<false, 1>(A, wins, st, (int)blockIdx.x); }
__global__ __launch_bounds__(kLinThreads) void ba_linearize_ln_f64_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_ln_body<false, 0>(A, wins, st, (int)blockIdx.x); }      // observations as given (BAArrays::packed = 0)
__global__ __launch_bounds__(kLinThreads) void ba_linearize_ln_big_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_ln_body<true, kPkRuntime>(A, wins, st, (int)blockIdx.x); }

__device__ __forceinline__ void line_obs_wtx(const BAArrays& A, const BAWin& W, int o, int c, const double* xp, double* t) {
  const double* Wb = A.lo_W + (size_t)o * 24;              // zero when both image edges are inactive
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; r++) s += Wb[r * 4 + k] * xp[c * 6 + r];
    t[k] = s;
  }
}
__device__ __forceinline__ double line_obs_trial(const BAArrays& A, const BAWin& W, const Pose& T, int o, const LineGeom& G, const LnObsIn& I) {
  double chi = 0.0;
  const Vec3 X1m = pose_map(T, G.X1), X2m = pose_map(T, G.X2);
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int e = 2 * o + side;
    const uint8_t fl = I.fl[side];
    if (!(fl & EF_VALID) || (fl & EF_LEVEL1)) continue;
    double r[2];
    line_residual_t<true>(W.cam, side == 1 ? W.cam.bx_right : 0.0, X1m, X2m, I.xs[side], I.ys[side], I.xe[side], I.ye[side], r, nullptr);
    const double c2 = chi2_of(r, 2, I.s[side]);
    A.le_chi2[e] = c2;
    double w, rho0 = c2;
    if (fl & EF_ROBUST) rho0 = huber_nr(c2, (fl & EF_PAIRSTEREO) ? W.th_ln_stereo : W.th_ln_mono, &w);
    chi += rho0;
  }
  return chi;
}
__device__ __forceinline__ double line_backsub(const double* V, double lambda, const double* wtx, const LineQ& L, LineQ& Ln) {
  double t[4], xl[4], sc = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) t[i] = V[10 + i] - wtx[i];
  chol_solve<4>(V, lambda, t, xl);
#pragma unroll
  for (int i = 0; i < 4; i++) sc += xl[i] * (lambda * xl[i] + V[10 + i]);
  Ln = line_oplus(L, xl);
  return sc;
}

// grid (nt_ln, nW), block 256 = 4 wavefronts, BAWin::rounds tasks per wavefront
// dynamic LDS: 8 + 7 n_cams + 6 n_free doubles (poses of the trial state, x_c); kBig: read from HBM instead (see ba_linearize_pt_body)
template <bool kBig, int kPk>
__device__ __forceinline__ void ba_backsub_ln_body(const BAArrays& A, const BAWin* __restrict__ wins, const BAState* __restrict__ st, const int bx) {   // bx: the workgroup's index along x
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* scratch = lds;
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  if ((int)bx >= W.nt_ln) return;
  const int cur = S.cur, nxt = cur ^ 1;
  const double lambda = S.lambda;
  double* camB_l = lds + 8; double* xps_l = camB_l + W.n_cams * 7;
  const double* camB = kBig ? A.cam_qt + ((size_t)nxt * A.NC + W.cam_off) * 7 : camB_l;
  const double* xp = kBig ? A.xp + W.x_off : xps_l;
  // (tasks fetched ahead: see ba_linearize_pt_body)
  const int task_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PTask T_next = A.ltasks[W.ltask_off + min((bx * W.rounds[3]) * 4 + task_wave, W.n_ltasks - 1)];
  if (!kBig) {
    for (int i = threadIdx.x; i < W.n_cams * 7; i += kLmThreads) camB_l[i] = A.cam_qt[((size_t)nxt * A.NC + W.cam_off) * 7 + i];
    for (int i = threadIdx.x; i < 6 * W.n_free; i += kLmThreads) xps_l[i] = A.xp[W.x_off + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double chi = 0.0, sc = 0.0;
  for (int rnd = 0; rnd < W.rounds[3]; rnd++) {
    const int ti = (bx * W.rounds[3] + rnd) * 4 + task_wave;      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
    if (ti >= W.n_ltasks) break;
    const PTask T = T_next;
    T_next = A.ltasks[W.ltask_off + min((bx * W.rounds[3] + rnd + 1) * 4 + task_wave, W.n_ltasks - 1)];
    if (T.nl > 1) {
      // two dependent memory levels, like the point kernel: (1) the task, (2) every global operand - the observation's arrays by observation
      // lane, the line's state, active byte and observation range by LANDMARK lane (lane i <-> line l0 + i: no trip through the line index
      // the observation carries); line data reaches the observation lanes by shuffle.  Until round 4 the line state hung off the
      // observation's line index (a third level) and Hll / b_l off the head lane's test against ln_obs_start (a fourth).
      const bool has = lane < T.ne;
      const int o = T.e0 + (has ? lane : 0);
      int c, l_raw;
      ln_cam_lm_of<kPk>(A, o, T.l0, c, l_raw);
      LnObsIn I;
      line_obs_load<kPk>(A, o, I);                              // (used after the back-substitution: in flight meanwhile)
      const bool lmk = lane < T.nl;
      const int g2 = W.ln_off + T.l0 + (lmk ? lane : 0);
      const LineQ L2 = load_ln(A, cur, g2);
      const int act2 = lmk ? (int)A.ln_active[g2] : 0;
      const int start2 = A.ln_obs_start[g2], end2 = A.ln_obs_start[g2 + 1];
      const int l = has ? l_raw : -1 - lane;
      const int slot = has ? l - T.l0 : 0;
      const bool lm_act = has && __shfl(act2, slot) != 0;
      double wtx[4] = {0, 0, 0, 0};
      if (lm_act && c < W.n_free) line_obs_wtx(A, W, o, c, xp, wtx);
      double V2[14];                                            // (issued once the 24 doubles of the Hpl block are consumed)
#pragma unroll
      for (int i = 0; i < 14; i++) V2[i] = A.ln_V[(size_t)g2 * 14 + i];
      seg_sum<4>(wtx, l, lane, T.ms);
      // landmark lane: back-substitution and oplus of its line (inactive / observation-less lines keep their state)
      const int first = (lmk && end2 > start2) ? start2 - T.e0 : 0;
      double wl[4];
#pragma unroll
      for (int i = 0; i < 4; i++) wl[i] = __shfl(wtx[i], first);
      LineQ Ln2 = L2;
      if (lmk) {
        if (act2 && end2 > start2) sc += line_backsub(V2, lambda, wl, L2, Ln2);
        store_ln(A, nxt, g2, Ln2);
      }
      LineQ Ln;
      Ln.q.x = __shfl(Ln2.q.x, slot); Ln.q.y = __shfl(Ln2.q.y, slot); Ln.q.z = __shfl(Ln2.q.z, slot); Ln.q.w = __shfl(Ln2.q.w, slot); Ln.alpha = __shfl(Ln2.alpha, slot);
      if (lm_act) chi += line_obs_trial(A, W, pose_load(camB + c * 7), o, line_geom(Ln), I);
    } else {
      const int g = W.ln_off + T.l0;
      const LineQ L = load_ln(A, cur, g);
      if (!A.ln_active[g]) { if (lane == 0) store_ln(A, nxt, g, L); }
      else {
        double wtx[4] = {0, 0, 0, 0};
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int o = T.e0 + sidx, c = ln_cam_of<kPk>(A, o);
          if (c >= W.n_free) continue;
          double t1[4];
          line_obs_wtx(A, W, o, c, xp, t1);
          wtx[0] += t1[0]; wtx[1] += t1[1]; wtx[2] += t1[2]; wtx[3] += t1[3];
        }
        wave_sum_n<4>(wtx);
        LineQ Ln;
        const double s1 = line_backsub(A.ln_V + (size_t)g * 14, lambda, wtx, L, Ln);
        if (lane == 0) { sc += s1; store_ln(A, nxt, g, Ln); }
        const LineGeom G = line_geom(Ln);
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int o = T.e0 + sidx;
          LnObsIn I;
          line_obs_load<kPk>(A, o, I);
          chi += line_obs_trial(A, W, pose_load(camB + ln_cam_of<kPk>(A, o) * 7), o, G, I);
        }
      }
    }
  }
  const double chi_t = block_sum(chi, scratch);
  const double sc_t = block_sum(sc, scratch);
  if (threadIdx.x == 0) { xwg_store(&A.chi_part2[W.part_off + W.nt_pt + bx], chi_t); xwg_store(&A.scale_part[W.part_off + W.nt_pt + bx], sc_t); xwg_stores_done(); }
}
__global__ __launch_bounds__(kLmThreads, 4) void ba_backsub_ln_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_ln_body<false, 1>(A, wins, st, (int)blockIdx.x); }
__global__ __launch_bounds__(kLmThreads, 4) void ba_backsub_ln_f64_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_ln_body<false, 0>(A, wins, st, (int)blockIdx.x); }      // observations as given (BAArrays::packed = 0)
__global__ __launch_bounds__(kLmThreads) void ba_backsub_ln_big_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_ln_body<true, kPkRuntime>(A, wins, st, (int)blockIdx.x); }

// Point and line landmarks in one launch, for batches too small to fill the GPU (a single window above all): there the two
// kernels of a pair are dependent launches of 8-16 us each on idle hardware.  Not for large batches: the fused kernel gets the
// register budget of the line body (223 VGPRs), which would halve the occupancy of the point body.
__global__ __launch_bounds__(kLinThreads) void ba_linearize_both_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st, int n_pt_blocks) {
  if ((int)blockIdx.x < n_pt_blocks) ba_linearize_pt_body<false, 1>(A, wins, st, (int)blockIdx.x);      // (packed observations only: the host launches the
  else ba_linearize_ln_body<false, 1>(A, wins, st, (int)blockIdx.x - n_pt_blocks);                     //  two kernels of the pair otherwise)
}

// LM iteration head of one window, by ONE wavefront: chi2 of the current state, lambda initialisation at iteration 0
// (optimization_algorithm_levenberg.cpp:75-99,166-180).
__device__ __forceinline__ void ba_begin_body(const BAArrays& A, const BAWin& W, BAState& S, int lane) {
  // chi2 of the current state: lanes sum interleaved partials, then a fixed shuffle tree (deterministic)
  double chi = 0.0;
  const int nb = W.nl_pt + W.nl_ln;
  for (int i = lane; i < nb; i += 64) chi += A.chi_part[W.part_off + i];
  chi = wave_sum(chi);
  double md = 0.0;
  if (S.it == 0) {
    // computeLambdaInit: tau * max |H_kk| over cameras and landmarks (optimization_algorithm_levenberg.cpp:166-180)
    const double* H = A.Hpp + (size_t)W.hpp_off * 21;
    for (int c = lane; c < W.n_free; c += 64) {
      const double* h = H + c * 21;
      md = fmax(md, fmax(fmax(fabs(xwg_load(h)), fabs(xwg_load(h + 6))), fmax(fmax(fabs(xwg_load(h + 11)), fabs(xwg_load(h + 15))), fmax(fabs(xwg_load(h + 18)), fabs(xwg_load(h + 20))))));
    }
    md = wave_max(md);
  }
  if (lane == 0) {
    S.currentChi = chi; S.iniChi = chi;
    if (S.it == 0) {
      md = fmax(md, __longlong_as_double((long long)S.maxdiag_bits));
      S.lambda = 1e-5 * md; S.ni = 2.0; S.nBad = 0;
    }
    S.q = 0; S.need_lin = 0;
  }
}

// Hpp / b_p = sum over the linearise workgroups' partials, fixed order; the window's LAST workgroup to finish then runs the LM iteration
// head (it was a launch of its own until round 4: one dependent launch less per linearisation).  grid (ceil(n_free_max*27 / 256), nW)
__global__ __launch_bounds__(256) void ba_hpp_reduce_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  __shared__ int is_last;
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN || !S.need_lin) return;             // (uniform over the window's workgroups: need_lin is only cleared behind the ticket)
  const int nacc = W.n_free * 27;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nacc) {
    const double* src = A.hpp_part + W.hpart_off + i;
    const int nb = W.big ? 1 : W.nl_pt + W.nl_ln;          // big: the linearise kernels added into one row directly
    // a small batch has ~140 partial rows per window and every row sits in another XCD's L2: eight independent loads in flight,
    // summed in row order (bit-identical to the plain loop)
    double v = 0.0;
    int b = 0;
    for (; b + 8 <= nb; b += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = src[(size_t)(b + u) * nacc];
#pragma unroll
      for (int u = 0; u < 8; u++) v += t[u];
    }
    for (; b < nb; b++) v += src[(size_t)b * nacc];
    const int c = i / 27, k = i - c * 27;
    if (k < 21) xwg_store(&A.Hpp[((size_t)W.hpp_off + c) * 21 + k], v); else A.bp[((size_t)W.hpp_off + c) * 6 + (k - 21)] = v;      // (the LM head reads Hpp's diagonal)
  }
  xwg_stores_done();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&S.ticket_lin, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (is_last && threadIdx.x < 64) {
    if (threadIdx.x == 0) S.ticket_lin = 0;
    ba_begin_body(A, W, S, threadIdx.x);
  }
}

