// lld_pose.hip — Optimizer::PoseOptimization (src/Optimizer.cc:653-932, AddLineMinOnlyPose :562-650) as ONE kernel:
// one workgroup per frame runs the whole protocol (4 rounds x 10 Levenberg–Marquardt iterations, outlier
// classification between rounds) without host round trips.  Per LM iteration the 256 lanes sweep the frame's point and
// line edges (SoA in HBM, L2-resident), build the 6x6 normal equations in registers (21+6+1 fp64 partials per lane),
// reduce them with a fixed shuffle tree + LDS, and every lane solves the damped 6x6 system redundantly (LDL^T) so no
// broadcast is needed.  g2o semantics kept: Huber weights use rho' only, chi2 is float-compared against 5.991f/7.815f,
// per-edge errors go stale exactly as in the reference (only outliers are re-evaluated before classification).
#include "lld_common.h"
#include "lld_device_math.h"

namespace {

using namespace lld;

constexpr int kPoseThreads = 256;

struct PoseFrameDev {            // per-frame header in HBM
  CamK cam;
  double T0[7];
  int pt_off, n_pt;              // into the point SoA
  int le_off, n_le;              // into the expanded line-edge SoA
  int ln_off, n_ln;              // into per-line arrays
  double delta_mono, delta_stereo;          // (double)(float)sqrt(5.991 / 7.815)
  double delta_ln_stereo, delta_ln_mono;    // (double)(float)(delta * gamma)
  double thr_ln_stereo, thr_ln_mono;        // float*float products, widened
};

struct PoseSoA {
  // points
  const double *px, *py, *pz, *u, *v, *ur, *s;
  double* pt_chi2; uint8_t* pt_level; uint8_t* pt_robust; uint8_t* pt_outlier;
  // line edges
  const double *x1x, *x1y, *x1z, *x2x, *x2y, *x2z, *xs, *ys, *xe, *ye, *ls, *lbx;
  const int* le_line; const uint8_t* le_last;
  double* le_chi2; uint8_t* le_level; uint8_t* le_robust;
  // lines
  const uint8_t* ln_has_right; uint8_t* ln_outlier;
};

struct PoseOut { double qt[7]; double chi2; int n_inliers, lm_iterations, lm_trials, pad; };

// Fixed-tree block reduction of N doubles per lane; every lane returns with the totals in `v`.
template <int N>
__device__ __forceinline__ void block_sum(double* v, double* lds /* [4][N] + [N] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; i++) {
    double x = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    v[i] = x;
  }
  __syncthreads();
  if (lane == 0) { for (int i = 0; i < N; i++) lds[wave * N + i] = v[i]; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = ((lds[i] + lds[N + i]) + lds[2 * N + i]) + lds[3 * N + i];
}

// Dense LDL^T of the 6x6 system (LinearSolverDense, solvers/linear_solver_dense.h:65-113): fails unless all pivots > 0.
__device__ __forceinline__ bool solve6(const double* Hu /*21 upper, row-major packed*/, double lambda, const double* b, double* x) {
  double A[6][6];
  int k = 0;
  for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) { A[i][j] = Hu[k]; A[j][i] = Hu[k]; k++; }
  for (int i = 0; i < 6; i++) A[i][i] += lambda;
  double L[6][6], D[6];
  bool ok = true;
  for (int j = 0; j < 6; j++) {
    double d = A[j][j];
    for (int p = 0; p < j; p++) d -= L[j][p] * L[j][p] * D[p];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    D[j] = d;
    for (int i = j + 1; i < 6; i++) {
      double s = A[i][j];
      for (int p = 0; p < j; p++) s -= L[i][p] * L[j][p] * D[p];
      L[i][j] = s / d;
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) { double s = b[i]; for (int p = 0; p < i; p++) s -= L[i][p] * y[p]; y[i] = s; }
  for (int i = 0; i < 6; i++) y[i] /= D[i];
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int p = i + 1; p < 6; p++) s -= L[p][i] * x[p]; x[i] = s; }
  return ok;
}

__device__ __forceinline__ void accum_unary(const double* J, int D, const double* e, double s, double w, double* H, double* b) {
  // BaseUnaryEdge::constructQuadraticForm (core/base_unary_edge.hpp:42-72): b -= w A^T (s e), H += A^T (w s) A
  const double ws = w * s;
  int k = 0;
  for (int r = 0; r < 6; r++) {
    double br = 0;
    for (int i = 0; i < D; i++) br += J[i * 6 + r] * e[i];
    b[r] -= ws * br;
    for (int c = r; c < 6; c++) {
      double h = 0;
      for (int i = 0; i < D; i++) h += J[i * 6 + r] * J[i * 6 + c];
      H[k++] += ws * h;
    }
  }
}

__global__ __launch_bounds__(kPoseThreads) void pose_opt_kernel(const PoseFrameDev* __restrict__ frames, PoseSoA a, PoseOut* __restrict__ out,
                                                               int n_rounds, int its_per_round, int max_trials) {
  __shared__ double red[5 * 28];
  const PoseFrameDev& F = frames[blockIdx.x];
  const CamK cam = F.cam;
  const int tid = threadIdx.x;
  const Pose T0 = pose_load(F.T0);
  Pose T = T0;
  const int n_pt = F.n_pt, n_le = F.n_le;
  const int po = F.pt_off, lo = F.le_off, no = F.ln_off;
  int lm_iterations = 0, lm_trials = 0;
  double last_chi = 0.0;
  int nBad_pts = 0;
  const bool enough = n_pt >= 3;                     // if(nInitialCorrespondences<3) return 0;  (Optimizer.cc:809-810)

  // error evaluation at pose P for the lane's edges; optionally accumulates the normal equations
  auto sweep = [&](const Pose& P, bool build, double* acc /*28: H21,b6,chi*/) {
    for (int i = 0; i < 28; i++) acc[i] = 0.0;
    for (int i = tid; i < n_pt; i += kPoseThreads) {
      if (a.pt_level[po + i] != 0) continue;
      const Vec3 Xc = pose_map(P, vec3(a.px[po + i], a.py[po + i], a.pz[po + i]));
      const double urv = a.ur[po + i];
      const bool stereo = !(urv < 0);
      double e[3];
      point_residual(cam, Xc, a.u[po + i], a.v[po + i], urv, stereo, false, e);
      const double s = a.s[po + i];
      const double chi = e[0] * (s * e[0]) + e[1] * (s * e[1]) + (stereo ? e[2] * (s * e[2]) : 0.0);
      a.pt_chi2[po + i] = chi;
      double w = 1.0, rho0 = chi;
      if (a.pt_robust[po + i]) rho0 = huber(chi, stereo ? F.delta_stereo : F.delta_mono, &w);
      acc[27] += rho0;
      if (build) { double J[18]; point_jac_pose(cam, Xc, stereo, J); accum_unary(J, stereo ? 3 : 2, e, s, w, acc, acc + 21); }
    }
    for (int i = tid; i < n_le; i += kPoseThreads) {
      if (a.le_level[lo + i] != 0) continue;
      const Vec3 X1m = pose_map(P, vec3(a.x1x[lo + i], a.x1y[lo + i], a.x1z[lo + i]));
      const Vec3 X2m = pose_map(P, vec3(a.x2x[lo + i], a.x2y[lo + i], a.x2z[lo + i]));
      double e[2]; LineAdj adj;
      line_residual(cam, a.lbx[lo + i], X1m, X2m, a.xs[lo + i], a.ys[lo + i], a.xe[lo + i], a.ye[lo + i], e, build ? &adj : nullptr);
      const double s = a.ls[lo + i];
      const double chi = e[0] * (s * e[0]) + e[1] * (s * e[1]);
      a.le_chi2[lo + i] = chi;
      double w = 1.0, rho0 = chi;
      if (a.le_robust[lo + i]) rho0 = huber(chi, a.ln_has_right[no + a.le_line[lo + i]] ? F.delta_ln_stereo : F.delta_ln_mono, &w);
      acc[27] += rho0;
      if (build) { double J[12]; line_jac_pose(adj, X1m, X2m, J); accum_unary(J, 2, e, s, w, acc, acc + 21); }
    }
  };

  double lambda = -1.0, ni = 2.0;
  int nBadLM = 0;
  if (enough) {
    for (int round = 0; round < n_rounds; round++) {
      T = T0;                                                        // vSE3->setEstimate(toSE3Quat(pFrame->mTcw))  (:823)
      // initializeOptimization(0): active = level-0 edges.  optimize() returns -1 when nothing is active.
      double cnt[1] = {0.0};
      for (int i = tid; i < n_pt; i += kPoseThreads) cnt[0] += a.pt_level[po + i] == 0 ? 1.0 : 0.0;
      for (int i = tid; i < n_le; i += kPoseThreads) cnt[0] += a.le_level[lo + i] == 0 ? 1.0 : 0.0;
      block_sum<1>(cnt, red);
      if (cnt[0] > 0.5) {
        bool ok = true;
        for (int it = 0; it < its_per_round && ok; it++) {
          // ---- OptimizationAlgorithmLevenberg::solve(it)  (optimization_algorithm_levenberg.cpp:61-164)
          double acc[28];
          sweep(T, true, acc);
          block_sum<28>(acc, red);
          double currentChi = acc[27];
          const double iniChi = currentChi;
          if (it == 0) {
            double md = 0.0; int k = 0;
            for (int r = 0; r < 6; r++) { md = fmax(fabs(acc[k]), md); k += 6 - r; }
            lambda = 1e-5 * md; ni = 2.0; nBadLM = 0;
          }
          double rho = 0.0; int q = 0;
          do {
            double x[6];
            const bool ok2 = solve6(acc, lambda, acc + 21, x);
            const Pose Tn = pose_oplus(T, x);
            double ev[28];
            sweep(Tn, false, ev);
            double tmp[1] = {ev[27]};
            block_sum<1>(tmp, red);
            double tempChi = ok2 ? tmp[0] : 1.7976931348623157e308;
            double scale = 0.0;
            for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + acc[21 + j]);
            scale += 1e-3;
            rho = (currentChi - tempChi) / scale;
            if (rho > 0 && isfinite(tempChi)) {
              double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
              alpha = fmin(alpha, 2. / 3.);
              lambda *= fmax(1. / 3., alpha);
              ni = 2; currentChi = tempChi; T = Tn;
            } else { lambda *= ni; ni *= 2; }
            q++; lm_trials++;
          } while (rho < 0 && q < max_trials);
          last_chi = currentChi;
          lm_iterations++;
          if (q == max_trials || rho == 0) ok = false;
          else {
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) ok = false;
          }
        }
      }
      // ---- classification (Optimizer.cc:827-913)
      __syncthreads();
      double nb[1] = {0.0};
      for (int i = tid; i < n_pt; i += kPoseThreads) {
        const double urv = a.ur[po + i];
        const bool stereo = !(urv < 0);
        double chi = a.pt_chi2[po + i];
        if (a.pt_outlier[po + i]) {                                 // if(pFrame->mvbOutlier[idx]) e->computeError();
          const Vec3 Xc = pose_map(T, vec3(a.px[po + i], a.py[po + i], a.pz[po + i]));
          double e[3]; point_residual(cam, Xc, a.u[po + i], a.v[po + i], urv, stereo, false, e);
          const double s = a.s[po + i];
          chi = e[0] * (s * e[0]) + e[1] * (s * e[1]) + (stereo ? e[2] * (s * e[2]) : 0.0);
          a.pt_chi2[po + i] = chi;
        }
        const float chif = (float)chi;
        const bool bad = chif > (stereo ? 7.815f : 5.991f);
        a.pt_outlier[po + i] = bad; a.pt_level[po + i] = bad;
        nb[0] += bad ? 1.0 : 0.0;
        if (round == 2) a.pt_robust[po + i] = 0;
      }
      block_sum<1>(nb, red);
      nBad_pts = (int)(nb[0] + 0.5);
      if (n_pt + n_le < 10) break;                                   // if(optimizer.edges().size()<10) break;
      for (int i = tid; i < n_le; i += kPoseThreads) {
        const Vec3 X1m = pose_map(T, vec3(a.x1x[lo + i], a.x1y[lo + i], a.x1z[lo + i]));
        const Vec3 X2m = pose_map(T, vec3(a.x2x[lo + i], a.x2y[lo + i], a.x2z[lo + i]));
        double e[2];
        line_residual(cam, a.lbx[lo + i], X1m, X2m, a.xs[lo + i], a.ys[lo + i], a.xe[lo + i], a.ye[lo + i], e, nullptr);
        const double s = a.ls[lo + i];
        const double chi = e[0] * (s * e[0]) + e[1] * (s * e[1]);
        a.le_chi2[lo + i] = chi;
        const float chif = (float)chi;
        const int idx = a.le_line[lo + i];
        // vnStereoLines is filled per EDGE but indexed by the LINE index (Optimizer.cc:643-648 vs :898)
        const bool st = idx < n_le ? a.ln_has_right[no + a.le_line[lo + idx]] != 0 : true;
        const double thr = st ? F.thr_ln_stereo : F.thr_ln_mono;
        const bool bad = (double)chif > thr;
        a.le_level[lo + i] = bad;
        if (a.le_last[lo + i]) a.ln_outlier[no + idx] = bad;        // the right-image edge overwrites the left one
        if (round == 2) a.le_robust[lo + i] = 0;
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    PoseOut& o = out[blockIdx.x];
    pose_store(T, o.qt);
    o.chi2 = last_chi; o.n_inliers = enough ? n_pt - nBad_pts : 0; o.lm_iterations = lm_iterations; o.lm_trials = lm_trials; o.pad = 0;
  }
}

}  // namespace

struct lld_pose_batch {
  lld_ctx* ctx = nullptr;
  int n_frames = 0;
  lld_pose_params params;
  void* slab = nullptr;
  PoseFrameDev* d_frames = nullptr;
  PoseSoA soa;
  PoseOut* d_out = nullptr;
  std::vector<PoseFrameDev> h_frames;
  std::vector<PoseOut> h_out;
  size_t n_pt_total = 0, n_le_total = 0, n_ln_total = 0;
  uint8_t *d_pt_level = nullptr, *d_pt_robust = nullptr, *d_pt_outlier = nullptr, *d_le_level = nullptr, *d_le_robust = nullptr, *d_ln_outlier = nullptr;
  std::vector<uint8_t> h_pt_outlier, h_ln_outlier;
  bool solved = false;
};

extern "C" {

int lld_pose_batch_create(lld_ctx* ctx, int n_frames, const lld_pose_problem* frames, const lld_pose_params* params, lld_pose_batch** out) {
  if (!ctx || n_frames <= 0 || !frames || !out) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  lld_pose_batch* B = new lld_pose_batch();
  B->ctx = ctx; B->n_frames = n_frames;
  if (params) B->params = *params; else lld_pose_params_default(&B->params);
  const double gamma = B->params.gamma;
  // host-side expansion (AddLineMinOnlyPose): one left edge per line and a right edge when the line has a stereo match
  std::vector<double> px, py, pz, u, v, ur, s, x1x, x1y, x1z, x2x, x2y, x2z, xs, ys, xe, ye, ls, lbx;
  std::vector<int> le_line; std::vector<uint8_t> le_last, has_right;
  B->h_frames.resize(n_frames);
  const float dMono = (float)std::sqrt(5.991), dStereo = (float)std::sqrt(7.815);
  float dLnS = dStereo, dLnM = dMono;
  dLnS *= gamma; dLnM *= gamma;                                      // float *= double (Optimizer.cc:706-707)
  for (int f = 0; f < n_frames; f++) {
    const lld_pose_problem& P = frames[f];
    if (P.n_points < 0 || P.n_lines < 0) { delete B; return LLD_ERR_INVALID; }
    PoseFrameDev& F = B->h_frames[f];
    F.cam = lld::make_camk(P.cam);
    std::memcpy(F.T0, P.pose_qt, sizeof F.T0);
    F.pt_off = (int)px.size(); F.n_pt = P.n_points;
    F.le_off = (int)ls.size(); F.ln_off = (int)has_right.size(); F.n_ln = P.n_lines;
    F.delta_mono = (double)dMono; F.delta_stereo = (double)dStereo;
    F.delta_ln_stereo = (double)dLnS; F.delta_ln_mono = (double)dLnM;
    F.thr_ln_stereo = (double)(dLnS * dLnS); F.thr_ln_mono = (double)(dLnM * dLnM);
    for (int i = 0; i < P.n_points; i++) {
      px.push_back(P.pt_xw[3 * i]); py.push_back(P.pt_xw[3 * i + 1]); pz.push_back(P.pt_xw[3 * i + 2]);
      u.push_back(P.pt_uvr[3 * i]); v.push_back(P.pt_uvr[3 * i + 1]); ur.push_back(P.pt_uvr[3 * i + 2]);
      s.push_back(P.pt_inv_sigma2[i]);
    }
    for (int l = 0; l < P.n_lines; l++) {
      const double* L = P.ln_left + 4 * l; const double* R = P.ln_right + 4 * l;
      const bool hr = !(R[0] < 0);
      has_right.push_back(hr ? 1 : 0);
      for (int si = 0; si < 2; si++) {
        if (si == 1 && !hr) continue;
        const double* kl = si == 0 ? L : R;
        x1x.push_back(P.ln_x0[3 * l]); x1y.push_back(P.ln_x0[3 * l + 1]); x1z.push_back(P.ln_x0[3 * l + 2]);
        x2x.push_back(P.ln_x0[3 * l] + P.ln_dir[3 * l]); x2y.push_back(P.ln_x0[3 * l + 1] + P.ln_dir[3 * l + 1]); x2z.push_back(P.ln_x0[3 * l + 2] + P.ln_dir[3 * l + 2]);
        xs.push_back(kl[0]); ys.push_back(kl[1]); xe.push_back(kl[2]); ye.push_back(kl[3]);
        ls.push_back(lld::line_info(gamma, P.ln_octave[2 * l + si]));
        lbx.push_back(si == 1 ? F.cam.bx_right : 0.0);
        le_line.push_back(l);
        le_last.push_back((si == 1 || !hr) ? 1 : 0);
      }
    }
    F.n_le = (int)ls.size() - F.le_off;
  }
  B->n_pt_total = px.size(); B->n_le_total = ls.size(); B->n_ln_total = has_right.size();
  const size_t NP = B->n_pt_total + 1, NE = B->n_le_total + 1, NL = B->n_ln_total + 1;
  size_t bytes = lld_slab::pad(sizeof(PoseFrameDev) * n_frames) + lld_slab::pad(sizeof(PoseOut) * n_frames) + 8 * lld_slab::pad(NP * 8) + 4 * lld_slab::pad(NP) +
                 13 * lld_slab::pad(NE * 8) + lld_slab::pad(NE * 4) + 3 * lld_slab::pad(NE) + 2 * lld_slab::pad(NL) + 4096;
  if (hipMalloc(&B->slab, bytes) != hipSuccess) { delete B; return LLD_ERR_ALLOC; }
  lld_slab sl; sl.base = (char*)B->slab;
  B->d_frames = sl.take<PoseFrameDev>(n_frames); B->d_out = sl.take<PoseOut>(n_frames);
  hipStream_t st = ctx->stream;
  auto up = [&](const std::vector<double>& h, size_t n) { double* d = sl.take<double>(n); if (!h.empty()) (void)hipMemcpyAsync(d, h.data(), h.size() * 8, hipMemcpyHostToDevice, st); return (const double*)d; };
  PoseSoA& A = B->soa;
  A.px = up(px, NP); A.py = up(py, NP); A.pz = up(pz, NP); A.u = up(u, NP); A.v = up(v, NP); A.ur = up(ur, NP); A.s = up(s, NP);
  A.pt_chi2 = sl.take<double>(NP);
  B->d_pt_level = sl.take<uint8_t>(NP); B->d_pt_robust = sl.take<uint8_t>(NP); B->d_pt_outlier = sl.take<uint8_t>(NP);
  A.pt_level = B->d_pt_level; A.pt_robust = B->d_pt_robust; A.pt_outlier = B->d_pt_outlier;
  (void)sl.take<uint8_t>(NP);
  A.x1x = up(x1x, NE); A.x1y = up(x1y, NE); A.x1z = up(x1z, NE); A.x2x = up(x2x, NE); A.x2y = up(x2y, NE); A.x2z = up(x2z, NE);
  A.xs = up(xs, NE); A.ys = up(ys, NE); A.xe = up(xe, NE); A.ye = up(ye, NE); A.ls = up(ls, NE); A.lbx = up(lbx, NE);
  A.le_chi2 = sl.take<double>(NE);
  int* dl = sl.take<int>(NE); if (!le_line.empty()) (void)hipMemcpyAsync(dl, le_line.data(), le_line.size() * 4, hipMemcpyHostToDevice, st); A.le_line = dl;
  uint8_t* dlast = sl.take<uint8_t>(NE); if (!le_last.empty()) (void)hipMemcpyAsync(dlast, le_last.data(), le_last.size(), hipMemcpyHostToDevice, st); A.le_last = dlast;
  B->d_le_level = sl.take<uint8_t>(NE); B->d_le_robust = sl.take<uint8_t>(NE); A.le_level = B->d_le_level; A.le_robust = B->d_le_robust;
  uint8_t* dhr = sl.take<uint8_t>(NL); if (!has_right.empty()) (void)hipMemcpyAsync(dhr, has_right.data(), has_right.size(), hipMemcpyHostToDevice, st); A.ln_has_right = dhr;
  B->d_ln_outlier = sl.take<uint8_t>(NL); A.ln_outlier = B->d_ln_outlier;
  LLD_HIP_TRY(hipMemcpyAsync(B->d_frames, B->h_frames.data(), sizeof(PoseFrameDev) * n_frames, hipMemcpyHostToDevice, st));
  LLD_HIP_TRY(hipStreamSynchronize(st));       // the host vectors die with this scope
  *out = B;
  return LLD_OK;
}

int lld_pose_batch_solve(lld_pose_batch* B) {
  if (!B) return LLD_ERR_INVALID;
  lld_ctx* ctx = B->ctx;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // every solve restarts from the uploaded state: level 0, Huber kernels on, nothing flagged
  LLD_HIP_TRY(hipMemsetAsync(B->d_pt_level, 0, B->n_pt_total + 1, st));
  LLD_HIP_TRY(hipMemsetAsync(B->d_pt_robust, 1, B->n_pt_total + 1, st));
  LLD_HIP_TRY(hipMemsetAsync(B->d_pt_outlier, 0, B->n_pt_total + 1, st));
  LLD_HIP_TRY(hipMemsetAsync(B->d_le_level, 0, B->n_le_total + 1, st));
  LLD_HIP_TRY(hipMemsetAsync(B->d_le_robust, 1, B->n_le_total + 1, st));
  LLD_HIP_TRY(hipMemsetAsync(B->d_ln_outlier, 0, B->n_ln_total + 1, st));
  hipLaunchKernelGGL(pose_opt_kernel, dim3(B->n_frames), dim3(kPoseThreads), 0, st, B->d_frames, B->soa, B->d_out, B->params.n_rounds,
                     B->params.its_per_round, B->params.max_trials);
  LLD_HIP_TRY(hipGetLastError());
  B->solved = false;
  return LLD_OK;
}

static int pose_batch_fetch(lld_pose_batch* B) {
  if (B->solved) return LLD_OK;
  hipStream_t st = B->ctx->stream;
  B->h_out.resize(B->n_frames); B->h_pt_outlier.resize(B->n_pt_total + 1); B->h_ln_outlier.resize(B->n_ln_total + 1);
  LLD_HIP_TRY(hipMemcpyAsync(B->h_out.data(), B->d_out, sizeof(PoseOut) * B->n_frames, hipMemcpyDeviceToHost, st));
  LLD_HIP_TRY(hipMemcpyAsync(B->h_pt_outlier.data(), B->d_pt_outlier, B->n_pt_total + 1, hipMemcpyDeviceToHost, st));
  LLD_HIP_TRY(hipMemcpyAsync(B->h_ln_outlier.data(), B->d_ln_outlier, B->n_ln_total + 1, hipMemcpyDeviceToHost, st));
  LLD_HIP_TRY(hipStreamSynchronize(st));
  B->solved = true;
  return LLD_OK;
}

int lld_pose_batch_download(lld_pose_batch* B, int frame, lld_pose_result* out) {
  if (!B || !out || frame < 0 || frame >= B->n_frames) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(B->ctx->device));
  int st = pose_batch_fetch(B); if (st) return st;
  const PoseOut& o = B->h_out[frame]; const PoseFrameDev& F = B->h_frames[frame];
  std::memcpy(out->pose_qt, o.qt, sizeof o.qt);
  out->n_inliers = o.n_inliers; out->lm_iterations = o.lm_iterations; out->lm_trials = o.lm_trials; out->reserved = 0; out->chi2 = o.chi2;
  if (out->pt_outlier && F.n_pt) std::memcpy(out->pt_outlier, B->h_pt_outlier.data() + F.pt_off, F.n_pt);
  if (out->ln_outlier && F.n_ln) std::memcpy(out->ln_outlier, B->h_ln_outlier.data() + F.ln_off, F.n_ln);
  return LLD_OK;
}

void lld_pose_batch_destroy(lld_pose_batch* B) {
  if (!B) return;
  (void)hipSetDevice(B->ctx->device);
  (void)hipStreamSynchronize(B->ctx->stream);
  if (B->slab) (void)hipFree(B->slab);
  delete B;
}

int lld_pose_opt(lld_ctx* ctx, const lld_pose_problem* in, const lld_pose_params* params, lld_pose_result* out) {
  if (!ctx || !in || !out) return LLD_ERR_INVALID;
  lld_pose_batch* B = nullptr;
  int st = lld_pose_batch_create(ctx, 1, in, params, &B); if (st) return st;
  st = lld_pose_batch_solve(B);
  if (!st) st = lld_pose_batch_download(B, 0, out);
  lld_pose_batch_destroy(B);
  return st;
}

}  // extern "C"
