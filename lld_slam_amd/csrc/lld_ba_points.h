// lld_ba_points.h - Point landmark kernels of the batched bundle adjustment: linearisation (ba_linearize_pt_*) and back-substitution + trial errors (ba_backsub_pt_*), one lane per EDGE.
// Part of lld_ba_kernels.h (split by kernel family in round 6; no behaviour change): included from there INSIDE namespace lldba, after the shared types and helpers.
// Not a stand-alone header.

// ================================================================== point landmarks: one lane per EDGE
// Edge SoA arrays are read fully coalesced (lane i <-> edge e0 + i); what belongs to a landmark (Hll, b_l, the back-substituted
// update) is combined over the landmark's lanes with a segmented shuffle reduction, the landmark's first lane ("head") does the
// per-landmark work, and results travel back to the lanes with one shuffle.
// Shifts by one lane over the whole wavefront go through the VALU (v_mov_b32_dpp wave_shl:1 / wave_shr:1), not through the LDS pipe:
// tools/microbench/lds_ops.hip measures 6.3 CU clocks per ds_bpermute_b32 against 1.3 for a DPP move, and the LDS pipe is what bounds
// the linearise kernels (it also carries their fp64 atomics).  Shifts by 2 and 4 are chains of single shifts.
// dpp_down1: lane i <- lane i + 1, dpp_up1: lane i <- lane i - 1; the lane without a source receives 0 (bound_ctrl), so no register has
// to be preset with a fill value
__device__ __forceinline__ int dpp_down1(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_up1(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }
template <int OFF>
__device__ __forceinline__ int dpp_down(int v) {
#pragma unroll
  for (int h = 0; h < OFF; h++) v = dpp_down1(v);
  return v;
}
template <int OFF>
__device__ __forceinline__ double dpp_down(double v) {
  return __hiloint2double(dpp_down<OFF>(__double2hiint(v)), dpp_down<OFF>(__double2loint(v)));
}
__device__ __forceinline__ bool seg_step(int seg, int lane, int off) {
  const int so = __shfl_down(seg, off);
  return (lane + off < 64) && so == seg;
}
// one step of the segmented sum: v += (value OFF lanes up, if that lane is in the same segment).  seg1 = segment id + 1 is never 0 in a
// lane that can receive the zero fill (ids are >= 0, or -1 - lane in idle lanes, and lane 0 receives no fill).  The condition enters as
// a factor 0.0 / 1.0 of one fused multiply-add per value (the partial sums are finite: idle lanes hold zeros).
template <int N, int OFF>
__device__ __forceinline__ void seg_sum_step(double* v, int seg1) {
  const bool ok = dpp_down<OFF>(seg1) == seg1;
  double o[N];
#pragma unroll
  for (int i = 0; i < N; i++) o[i] = dpp_down<OFF>(v[i]);
  // one predicated block of adds (EXEC = the lanes that continue their segment): a non-finite partial of a NEIGHBOURING landmark
  // cannot leak in, which a 0.0 / 1.0 factor in an FMA would let it do (0 * NaN)
  if (ok) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += o[i];
  }
}
template <int N>
__device__ __forceinline__ void seg_sum(double* v, int seg, int lane, int max_len) {      // valid in the first lane of every segment
  if (max_len > 1) seg_sum_step<N, 1>(v, seg + 1);
  if (max_len > 2) seg_sum_step<N, 2>(v, seg + 1);
  if (max_len > 4) seg_sum_step<N, 4>(v, seg + 1);
  for (int off = 8; off < max_len; off <<= 1) {
    const bool ok = seg_step(seg, lane, off);
#pragma unroll
    for (int i = 0; i < N; i++) { const double o = __shfl_down(v[i], off); if (ok) v[i] += o; }
  }
}
template <int N>
__device__ __forceinline__ void wave_sum_n(double* v) {
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = wave_sum(v[i]);
}

struct PtEdgeLin { double r[3], Jp[9], Jc[18], ws, rho0; bool stereo; };

// residual, chi2 (stored), Huber weight, Jacobians of one active point edge at the linearisation point
template <int kPk>
__device__ __forceinline__ void point_edge_linearize(const BAArrays& A, const BAWin& W, int cur, int e, uint8_t fl, int c, const Vec3& X, PtEdgeLin& L) {
  const Pose T = load_cam(A, cur, W.cam_off + c);
  const Vec3 Xc = pose_map(T, X);
  const PtObs ob = pt_obs_of<kPk>(A, e);
  L.stereo = !(ob.ur < 0);
  point_residual_iz(W.cam, Xc, rcp_nr(Xc.z), ob.u, ob.v, ob.ur, L.stereo, true, L.r);
  const double s = ob.s;
  const double c2 = chi2_of(L.r, L.stereo ? 3 : 2, s);
  A.pe_chi2[e] = c2;
  double w = 1.0;
  L.rho0 = c2;
  if (fl & EF_ROBUST) L.rho0 = huber_nr(c2, L.stereo ? W.th_stereo : W.th_mono, &w);
  L.ws = w * s;
  A.pe_ws[e] = L.stereo ? -L.ws : L.ws;                   // (sign bit = stereo edge, see BAArrays::pe_ws)
  point_jac_point(W.cam, Xc, quat_rotation(T.q), L.stereo, L.Jp);
  point_jac_pose(W.cam, Xc, L.stereo, L.Jc);
}
// landmark side Hll (6 upper) + b_l (3) of one edge
__device__ __forceinline__ void point_edge_hll(const PtEdgeLin& L, double* hb) {
  int k = 0;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int d = a; d < 3; d++) hb[k++] = L.ws * (L.Jp[a] * L.Jp[d] + L.Jp[3 + a] * L.Jp[3 + d] + L.Jp[6 + a] * L.Jp[6 + d]);
#pragma unroll
  for (int a = 0; a < 3; a++) hb[6 + a] = -L.ws * (L.Jp[a] * L.r[0] + L.Jp[3 + a] * L.r[1] + L.Jp[6 + a] * L.r[2]);
}
// camera side: Hpp (21 upper) and b_p (6) into the LDS-staged per-camera accumulators
__device__ __forceinline__ void point_edge_hpp(const PtEdgeLin& L, double* ac) {
  int kk = 0;
#pragma unroll
  for (int rr = 0; rr < 6; rr++) {
    atomicAdd(&ac[21 + rr], -L.ws * (L.Jc[rr] * L.r[0] + L.Jc[6 + rr] * L.r[1] + L.Jc[12 + rr] * L.r[2]));
#pragma unroll
    for (int cc = rr; cc < 6; cc++) atomicAdd(&ac[kk++], L.ws * (L.Jc[rr] * L.Jc[cc] + L.Jc[6 + rr] * L.Jc[6 + cc] + L.Jc[12 + rr] * L.Jc[12 + cc]));
  }
}

// Everything one active point edge adds to the normal equations, in closed form (same algebra as point_hpl_closed): with
// A = d(u,v,uR)/dXc, M = ws A^T A, g = A^T (ws r), P = [Xc]x M:
//   Hll = R^T M R, b_l = R^T g;  Hpp = [[ Xc x P_i (rows) , P ], [ . , M ]], b_p = [ Xc x g ; g ]
// instead of forming Jp (3x3) and Jc (3x6) and contracting them.  hb: 6 upper of Hll + b_l; hp: 21 upper of Hpp + b_p (row-major
// packed like point_edge_hpp).  Returns chi2 of the edge; ws and rho0 through the references.
__device__ __forceinline__ double point_edge_blocks_closed(const BAWin& W, const Pose& T, const Vec3& X, const PtObs& ob, uint8_t fl, double& ws_out,
                                                           double& rho0_out, double* hb, double* hp) {
  const CamK& k = W.cam;
  const Vec3 Xc = pose_map(T, X);
  const bool stereo = !(ob.ur < 0);
  double r[3];
  const double iz = rcp_nr(Xc.z), iz2 = iz * iz;             // one reciprocal for the residual and the Jacobian entries
  point_residual_iz(k, Xc, iz, ob.u, ob.v, ob.ur, stereo, true, r);
  const double c2e = chi2_of(r, stereo ? 3 : 2, ob.s);
  double w = 1.0, rho0 = c2e;
  if (fl & EF_ROBUST) rho0 = huber_nr(c2e, stereo ? W.th_stereo : W.th_mono, &w);
  const double ws = w * ob.s;
  ws_out = ws; rho0_out = rho0;
  const Mat3 R = quat_rotation(T.q);
  const double a = k.fx * iz, b = k.fy * iz;
  const double c0 = -k.fx * Xc.x * iz2, c1 = -k.fy * Xc.y * iz2, c2 = c0 + k.bf * iz2;
  const double m00 = ws * (stereo ? 2.0 * a * a : a * a);
  const double m02 = ws * (stereo ? a * (c0 + c2) : a * c0);
  const double m11 = ws * (b * b), m12 = ws * (b * c1);
  const double m22 = ws * (stereo ? c0 * c0 + c1 * c1 + c2 * c2 : c0 * c0 + c1 * c1);
  const double wr0 = ws * r[0], wr1 = ws * r[1], wr2 = stereo ? ws * r[2] : 0.0;
  const double g0 = a * (wr0 + wr2), g1 = b * wr1, g2 = c0 * wr0 + c1 * wr1 + c2 * wr2;
  // landmark side
  double G[3][3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    G[0][j] = m00 * R.m[0][j] + m02 * R.m[2][j];
    G[1][j] = m11 * R.m[1][j] + m12 * R.m[2][j];
    G[2][j] = m02 * R.m[0][j] + m12 * R.m[1][j] + m22 * R.m[2][j];
  }
  int kk = 0;
#pragma unroll
  for (int p = 0; p < 3; p++)
#pragma unroll
    for (int d = p; d < 3; d++) hb[kk++] = R.m[0][p] * G[0][d] + R.m[1][p] * G[1][d] + R.m[2][p] * G[2][d];
#pragma unroll
  for (int p = 0; p < 3; p++) hb[6 + p] = R.m[0][p] * g0 + R.m[1][p] * g1 + R.m[2][p] * g2;
  // camera side: P = [Xc]x M (column j = Xc x M[:,j]); M is symmetric with m01 = 0
  const double x = Xc.x, y = Xc.y, z = Xc.z;
  const double P[3][3] = {{y * m02 - z * 0.0, y * m12 - z * m11, y * m22 - z * m12},
                          {z * m00 - x * m02, z * 0.0 - x * m12, z * m02 - x * m22},
                          {x * 0.0 - y * m00, x * m11 - y * 0.0, x * m12 - y * m02}};
  // rotation-rotation block: row i = Xc x P[i,:]
  const double Q[3][3] = {{y * P[0][2] - z * P[0][1], z * P[0][0] - x * P[0][2], x * P[0][1] - y * P[0][0]},
                          {y * P[1][2] - z * P[1][1], z * P[1][0] - x * P[1][2], x * P[1][1] - y * P[1][0]},
                          {y * P[2][2] - z * P[2][1], z * P[2][0] - x * P[2][2], x * P[2][1] - y * P[2][0]}};
  // packed upper triangle, rows 0..5: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
  hp[0] = Q[0][0]; hp[1] = Q[0][1]; hp[2] = Q[0][2]; hp[3] = P[0][0]; hp[4] = P[0][1]; hp[5] = P[0][2];
  hp[6] = Q[1][1]; hp[7] = Q[1][2]; hp[8] = P[1][0]; hp[9] = P[1][1]; hp[10] = P[1][2];
  hp[11] = Q[2][2]; hp[12] = P[2][0]; hp[13] = P[2][1]; hp[14] = P[2][2];
  hp[15] = m00; hp[16] = 0.0; hp[17] = m02; hp[18] = m11; hp[19] = m12; hp[20] = m22;
  hp[21] = y * g2 - z * g1; hp[22] = z * g0 - x * g2; hp[23] = x * g1 - y * g0; hp[24] = g0; hp[25] = g1; hp[26] = g2;
  return c2e;
}

// grid (nl_pt, nW), block 512 = 8 wavefronts, BAWin::rounds tasks per wavefront; dynamic LDS: kAccCopies*n_free_max*27 doubles + 8 scratch.
// kBig (a map with more cameras than the LDS holds accumulators and poses for, BAWin::big): the camera accumulators are ONE row in HBM
// (zeroed by ba_init / ba_control / ba_round2, added to with global fp64 atomics) and the poses are read from HBM.
template <bool kBig, int kPk>
__device__ __forceinline__ void ba_linearize_pt_body(const BAArrays& A, const BAWin* __restrict__ wins, BAState* __restrict__ st, const int bx) {   // bx: the workgroup's index along x
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN || !S.need_lin) return;
  if ((int)bx >= W.nl_pt) return;
  const int nacc = W.n_free * 27;
  const int cur = S.cur;
  double* acc_all = kBig ? A.hpp_part + W.hpart_off : lds;   // kAccCopies x [n_free][21 Hpp upper + 6 bp]
  const int copies = W.acc_copies[0];
  double* scratch = kBig ? lds : lds + copies * nacc;
  double* cams_l = scratch + 8;                              // [n_cams][7] poses of the linearisation point
  const double* cams = kBig ? A.cam_qt + ((size_t)cur * A.NC + W.cam_off) * 7 : cams_l;
  double* acc = acc_all;
  const int nthr = blockDim.x, nwv = W.lin_waves[0];         // 512 / 8; bit-reproducible mode: one wavefront per accumulator copy
  // the wavefront's tasks: the first one is fetched while the workgroup stages its LDS copies, the next one while the current one is
  // worked on (scalar loads: a task index never waits for a dependent load of its own inside the loop).  Fetching the per-lane operands
  // of the next task as well (13 registers: edge record, camera word, flags, landmark) was measured and dropped: 128 VGPRs with 10
  // spilled, ba_linearize 13.75 -> 15.2 ms per step.
  const int task_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PTask T_next = A.ptasks[W.ptask_off + min((bx * W.rounds[0]) * nwv + task_wave, W.n_ptasks - 1)];
  if (!kBig) {
    for (int i = threadIdx.x; i < copies * nacc; i += nthr) acc_all[i] = 0.0;
    // Default: the lanes of a wavefront are spread over the copies (same-address LDS atomics serialise) and every copy is shared by all
    // wavefronts - the order of the adds varies from run to run.  Deterministic mode: copy = wavefront, so a copy only ever sees ONE
    // wavefront's adds, in program order (lanes of one instruction that hit the same camera are serialised by the LDS in lane order).
    acc = acc_all + (W.det ? (int)(threadIdx.x >> 6) : (int)((threadIdx.x >> 3) & (copies - 1))) * nacc;
    for (int i = threadIdx.x; i < W.n_cams * 7; i += nthr) cams_l[i] = A.cam_qt[((size_t)cur * A.NC + W.cam_off) * 7 + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double chi = 0.0, maxd = 0.0;
  for (int rnd = 0; rnd < W.rounds[0]; rnd++) {
    const int ti = (bx * W.rounds[0] + rnd) * nwv + task_wave;      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
    if (ti >= W.n_ptasks) break;
    const PTask T = T_next;
    T_next = A.ptasks[W.ptask_off + min((bx * W.rounds[0] + rnd + 1) * nwv + task_wave, W.n_ptasks - 1)];
    if (T.nl > 1) {
      // two dependent memory levels only: (1) the task, (2) every global operand - edge arrays by edge lane, landmark
      // state by landmark lane (lane i <-> landmark l0 + i); camera poses come from the workgroup's LDS copy and landmark
      // data reaches the edge lanes by shuffle.
      const bool has = lane < T.ne;
      const int e = T.e0 + (has ? lane : 0);
      int c, l_raw;
      pt_cam_lm_of<kPk>(A, e, T.l0, c, l_raw);
      const int l = has ? l_raw : -1 - lane;
      const uint8_t fl = A.pe_flags[e];
      const PtObs ob = pt_obs_of<kPk>(A, e);
      const bool lmk = lane < T.nl;
      const int g2 = W.pt_off + T.l0 + (lmk ? lane : 0);
      const Vec3 X2 = load_pt(A, cur, g2);
      const int act2 = lmk ? (int)A.pt_active[g2] : 0;
      const int slot = has ? l - T.l0 : 0;
      Vec3 X; X.x = __shfl(X2.x, slot); X.y = __shfl(X2.y, slot); X.z = __shfl(X2.z, slot);
      const bool lm_act = has && __shfl(act2, slot) != 0;
      const bool head = lm_act && dpp_up1(l + 1) != l + 1;               // the first edge lane of an active landmark (lane 0 receives 0)
      double hb[9];
#pragma unroll
      for (int i = 0; i < 9; i++) hb[i] = 0.0;
      if (has && (fl & EF_LEVEL1)) A.pe_ws[e] = 0.0;
      if (lm_act && !(fl & EF_LEVEL1)) {
        double hp[27], ws_e, rho0_e;
        A.pe_chi2[e] = point_edge_blocks_closed(W, pose_load(cams + c * 7), X, ob, fl, ws_e, rho0_e, hb, hp);
        A.pe_ws[e] = (fl & EF_STEREO) ? -ws_e : ws_e;
        chi += rho0_e;
        if (c < W.n_free) {
          double* ac = acc + c * 27;
#pragma unroll
          for (int i = 0; i < 27; i++) if (i != 16) atomicAdd(&ac[i], hp[i]);          // entry 16 is the structural zero of M
        }
      }
      seg_sum<9>(hb, l, lane, T.ms);
      // the head lane of a landmark holds the sums: it writes Hll / b_l itself (no trip back to the landmark lane)
      if (head) {
        double* V = A.pt_V + (size_t)(W.pt_off + l) * 9;
#pragma unroll
        for (int i = 0; i < 9; i++) V[i] = hb[i];
        maxd = fmax(maxd, fmax(fabs(hb[0]), fmax(fabs(hb[3]), fabs(hb[5]))));
      }
    } else {                                             // a single landmark, any number of edges
      const int g = W.pt_off + T.l0;
      if (A.pt_active[g]) {
        const Vec3 X = load_pt(A, cur, g);
        double hb[9];
#pragma unroll
        for (int i = 0; i < 9; i++) hb[i] = 0.0;
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int e = T.e0 + sidx;
          const uint8_t fl = A.pe_flags[e];
          if (fl & EF_LEVEL1) { A.pe_ws[e] = 0.0; continue; }
          const int c = pt_cam_of<kPk>(A, e);
          PtEdgeLin L;
          point_edge_linearize<kPk>(A, W, cur, e, fl, c, X, L);
          chi += L.rho0;
          double h1[9];
          point_edge_hll(L, h1);
#pragma unroll
          for (int i = 0; i < 9; i++) hb[i] += h1[i];
          if (c < W.n_free) point_edge_hpp(L, acc + c * 27);
        }
        wave_sum_n<9>(hb);
        if (lane == 0) {
          double* V = A.pt_V + (size_t)g * 9;
#pragma unroll
          for (int i = 0; i < 9; i++) V[i] = hb[i];
          maxd = fmax(maxd, fmax(fabs(hb[0]), fmax(fabs(hb[3]), fabs(hb[5]))));
        }
      }
    }
  }
  const double chi_t = block_sum(chi, scratch);
  const double max_t = block_max(maxd, scratch);
  if (threadIdx.x == 0) {
    A.chi_part[W.part_off + bx] = chi_t;
    atomicMax(&S.maxdiag_bits, (unsigned long long)__double_as_longlong(max_t));
  }
  __syncthreads();
  if (kBig) return;
  // plain stores of this workgroup's camera partials; ba_hpp_reduce sums them in a fixed order (no global atomics)
  double* dst = A.hpp_part + W.hpart_off + (size_t)(bx) * nacc;
  for (int i = threadIdx.x; i < nacc; i += nthr) {
    double v = 0.0;
    for (int q = 0; q < copies; q++) v += acc_all[q * nacc + i];
    dst[i] = v;
  }
}
__global__ __launch_bounds__(kLinThreads, 4) void ba_linearize_pt_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_pt_body<false, 1>(A, wins, st, (int)blockIdx.x); }
__global__ __launch_bounds__(kLinThreads, 4) void ba_linearize_pt_f64_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_pt_body<false, 0>(A, wins, st, (int)blockIdx.x); }      // observations as given (BAArrays::packed = 0)
__global__ __launch_bounds__(kLinThreads, 4) void ba_linearize_pt_big_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_pt_body<true, kPkRuntime>(A, wins, st, (int)blockIdx.x); }

// W_e^T x_c = ws * Jp^T (Jc x_c) of one point edge with the Jacobians of the linearisation point
__device__ __forceinline__ void point_edge_wtx(const BAArrays& A, const BAWin& W, int cur, int e, uint8_t fl, int c, const Vec3& X, const double* xp, double* t) {
  const double ws = fabs(A.pe_ws[e]);
  const bool stereo = (fl & EF_STEREO) != 0;
  const Pose T = load_cam(A, cur, W.cam_off + c);
  const Vec3 Xc = pose_map(T, X);
  double Jp[9], Jc[18];
  point_jac_point(W.cam, Xc, quat_rotation(T.q), stereo, Jp);
  point_jac_pose(W.cam, Xc, stereo, Jc);
  double uu[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; r++) s += Jc[i * 6 + r] * xp[c * 6 + r];
    uu[i] = ws * s;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) t[k] = Jp[k] * uu[0] + Jp[3 + k] * uu[1] + Jp[6 + k] * uu[2];
}
// trial-state residual of one active point edge: stores chi2, returns its (robust) cost
template <int kPk>
__device__ __forceinline__ double point_edge_trial(const BAArrays& A, const BAWin& W, int nxt, int e, uint8_t fl, int c, const Vec3& Xn) {
  const Pose T = load_cam(A, nxt, W.cam_off + c);
  const Vec3 Xc = pose_map(T, Xn);
  const PtObs ob = pt_obs_of<kPk>(A, e);
  const bool stereo = !(ob.ur < 0);
  double r[3];
  point_residual_iz(W.cam, Xc, rcp_nr(Xc.z), ob.u, ob.v, ob.ur, stereo, true, r);
  const double c2 = chi2_of(r, stereo ? 3 : 2, ob.s);
  A.pe_chi2[e] = c2;
  double w, rho0 = c2;
  if (fl & EF_ROBUST) rho0 = huber_nr(c2, stereo ? W.th_stereo : W.th_mono, &w);
  return rho0;
}
// x_l = (Hll + lambda I)^-1 (b_l - sum W^T x_c), oplus; returns the landmark's part of computeScale
__device__ __forceinline__ double point_backsub(const double* V, double lambda, const double* wtx, const Vec3& X, Vec3& Xn) {
  const double t[3] = {V[6] - wtx[0], V[7] - wtx[1], V[8] - wtx[2]};
  double xl[3], sc = 0.0;
  chol_solve<3>(V, lambda, t, xl);
#pragma unroll
  for (int i = 0; i < 3; i++) sc += xl[i] * (lambda * xl[i] + V[6 + i]);
  Xn = vec3(X.x + xl[0], X.y + xl[1], X.z + xl[2]);      // VertexSBAPointXYZ::oplusImpl
  return sc;
}

// grid (nt_pt, nW), block 256 = 4 wavefronts, BAWin::rounds tasks per wavefront; dynamic LDS: 8 + 14 n_cams + 6 n_free doubles
template <bool kBig, int kPk>
__device__ __forceinline__ void ba_backsub_pt_body(const BAArrays& A, const BAWin* __restrict__ wins, const BAState* __restrict__ st, const int bx) {   // bx: the workgroup's index along x
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  if ((int)bx >= W.nt_pt) return;
  const int cur = S.cur, nxt = cur ^ 1;
  const double lambda = S.lambda;
  const double* xp = A.xp + W.x_off;
  // workgroup copies of what every edge lane gathers: poses of the linearisation point (camA) and of the trial state (camB),
  // and the camera part of the solution
  double* scratch = lds;
  double* camA_l = lds + 8;
  double* camB_l = camA_l + W.n_cams * 7;
  double* xps_l = camB_l + W.n_cams * 7;
  // kBig: no LDS copies, the poses and x_c are read from HBM (see ba_linearize_pt_body)
  const double* camA = kBig ? A.cam_qt + ((size_t)cur * A.NC + W.cam_off) * 7 : camA_l;
  const double* camB = kBig ? A.cam_qt + ((size_t)nxt * A.NC + W.cam_off) * 7 : camB_l;
  const double* xps = kBig ? xp : xps_l;
  // the wavefront's tasks: the first one is fetched while the workgroup stages its LDS copies, the next one while the current one is
  // worked on (scalar loads: a task index never waits for a dependent load of its own inside the loop)
  const int task_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PTask T_next = A.ptasks[W.ptask_off + min((bx * W.rounds[2]) * 4 + task_wave, W.n_ptasks - 1)];
  if (!kBig) {
    for (int i = threadIdx.x; i < W.n_cams * 7; i += kLmThreads) {
      camA_l[i] = A.cam_qt[((size_t)cur * A.NC + W.cam_off) * 7 + i];
      camB_l[i] = A.cam_qt[((size_t)nxt * A.NC + W.cam_off) * 7 + i];
    }
    for (int i = threadIdx.x; i < 6 * W.n_free; i += kLmThreads) xps_l[i] = xp[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double chi = 0.0, sc = 0.0;
  for (int rnd = 0; rnd < W.rounds[2]; rnd++) {
    const int ti = (bx * W.rounds[2] + rnd) * 4 + task_wave;      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
    if (ti >= W.n_ptasks) break;
    const PTask T = T_next;
    T_next = A.ptasks[W.ptask_off + min((bx * W.rounds[2] + rnd + 1) * 4 + task_wave, W.n_ptasks - 1)];
    if (T.nl > 1) {
      const bool has = lane < T.ne;
      const int e = T.e0 + (has ? lane : 0);
      int c, l_raw;
      pt_cam_lm_of<kPk>(A, e, T.l0, c, l_raw);
      const int l = has ? l_raw : -1 - lane;
      const uint8_t fl = A.pe_flags[e];
      const double ws = fabs(A.pe_ws[e]);
      const PtObs ob = pt_obs_of<kPk>(A, e);
      const bool lmk = lane < T.nl;
      const int g2 = W.pt_off + T.l0 + (lmk ? lane : 0);
      const Vec3 X2 = load_pt(A, cur, g2);
      const int act2 = lmk ? (int)A.pt_active[g2] : 0;
      const int start2 = A.pt_obs_start[g2], end2 = A.pt_obs_start[g2 + 1];
      double V2[9];
#pragma unroll
      for (int i = 0; i < 9; i++) V2[i] = A.pt_V[(size_t)g2 * 9 + i];
      const int slot = has ? l - T.l0 : 0;
      Vec3 X; X.x = __shfl(X2.x, slot); X.y = __shfl(X2.y, slot); X.z = __shfl(X2.z, slot);
      const bool e_act = has && __shfl(act2, slot) != 0 && !(fl & EF_LEVEL1);
      double wtx[3] = {0, 0, 0};
      if (e_act && c < W.n_free) {
        // W_e^T x_c = ws * Jp^T (Jc x_c) with the Jacobians of the linearisation point
        // closed form (see point_hpl_closed): Jc x = A (Xc x x_w - x_t), Jp^T u = -R^T A^T u
        const Pose Tc = pose_load(camA + c * 7);
        const Vec3 Xc = pose_map(Tc, X);
        const bool stereo = (fl & EF_STEREO) != 0;
        const Mat3 R = quat_rotation(Tc.q);
        const double* xc = xps + c * 6;
        const Vec3 v = cross(Xc, vec3(xc[0], xc[1], xc[2])) - vec3(xc[3], xc[4], xc[5]);
        const double iz = rcp_nr(Xc.z), iz2 = iz * iz;
        const double a = W.cam.fx * iz, b = W.cam.fy * iz;
        const double c0 = -W.cam.fx * Xc.x * iz2, c1 = -W.cam.fy * Xc.y * iz2, c2 = c0 + W.cam.bf * iz2;
        const double u0 = ws * (a * v.x + c0 * v.z), u1 = ws * (b * v.y + c1 * v.z), u2 = stereo ? ws * (a * v.x + c2 * v.z) : 0.0;
        const double h0 = a * (u0 + u2), h1 = b * u1, h2 = c0 * u0 + c1 * u1 + c2 * u2;
#pragma unroll
        for (int k = 0; k < 3; k++) wtx[k] = -(R.m[0][k] * h0 + R.m[1][k] * h1 + R.m[2][k] * h2);
      }
      seg_sum<3>(wtx, l, lane, T.ms);
      // landmark lane: back-substitution and oplus of its landmark (inactive / edge-less landmarks keep their state)
      const int first = (lmk && end2 > start2) ? start2 - T.e0 : 0;
      double wl[3];
#pragma unroll
      for (int i = 0; i < 3; i++) wl[i] = __shfl(wtx[i], first);
      Vec3 Xn2 = X2;
      if (lmk) {
        if (act2 && end2 > start2) sc += point_backsub(V2, lambda, wl, X2, Xn2);
        store_pt(A, nxt, g2, Xn2);
      }
      Vec3 Xn; Xn.x = __shfl(Xn2.x, slot); Xn.y = __shfl(Xn2.y, slot); Xn.z = __shfl(Xn2.z, slot);
      if (e_act) {
        const Vec3 Xc = pose_map(pose_load(camB + c * 7), Xn);
        const bool stereo = !(ob.ur < 0);
        double r[3];
        point_residual_iz(W.cam, Xc, rcp_nr(Xc.z), ob.u, ob.v, ob.ur, stereo, true, r);
        const double c2 = chi2_of(r, stereo ? 3 : 2, ob.s);
        A.pe_chi2[e] = c2;
        double w, rho0 = c2;
        if (fl & EF_ROBUST) rho0 = huber_nr(c2, stereo ? W.th_stereo : W.th_mono, &w);
        chi += rho0;
      }
    } else {
      const int g = W.pt_off + T.l0;
      const Vec3 X = load_pt(A, cur, g);
      if (!A.pt_active[g]) { if (lane == 0) store_pt(A, nxt, g, X); }
      else {
        double wtx[3] = {0, 0, 0};
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int e = T.e0 + sidx;
          const uint8_t fl = A.pe_flags[e];
          const int c = pt_cam_of<kPk>(A, e);
          if ((fl & EF_LEVEL1) || c >= W.n_free) continue;
          double t1[3];
          point_edge_wtx(A, W, cur, e, fl, c, X, xp, t1);
          wtx[0] += t1[0]; wtx[1] += t1[1]; wtx[2] += t1[2];
        }
        wave_sum_n<3>(wtx);
        Vec3 Xn;
        const double s1 = point_backsub(A.pt_V + (size_t)g * 9, lambda, wtx, X, Xn);      // every lane, same value
        if (lane == 0) { sc += s1; store_pt(A, nxt, g, Xn); }
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int e = T.e0 + sidx;
          const uint8_t fl = A.pe_flags[e];
          if (fl & EF_LEVEL1) continue;
          chi += point_edge_trial<kPk>(A, W, nxt, e, fl, pt_cam_of<kPk>(A, e), Xn);
        }
      }
    }
  }
  const double chi_t = block_sum(chi, scratch);
  const double sc_t = block_sum(sc, scratch);
  if (threadIdx.x == 0) { xwg_store(&A.chi_part2[W.part_off + bx], chi_t); xwg_store(&A.scale_part[W.part_off + bx], sc_t); xwg_stores_done(); }
}
__global__ __launch_bounds__(kLmThreads) void ba_backsub_pt_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_pt_body<false, 1>(A, wins, st, (int)blockIdx.x); }
__global__ __launch_bounds__(kLmThreads) void ba_backsub_pt_f64_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_pt_body<false, 0>(A, wins, st, (int)blockIdx.x); }      // observations as given (BAArrays::packed = 0)
__global__ __launch_bounds__(kLmThreads) void ba_backsub_pt_big_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_pt_body<true, kPkRuntime>(A, wins, st, (int)blockIdx.x); }

