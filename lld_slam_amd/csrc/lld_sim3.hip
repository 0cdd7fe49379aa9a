// lld_sim3.hip — Optimizer::OptimizeSim3 (src/Optimizer.cc:1656-1851) as one kernel: a workgroup per (KF1, KF2) candidate runs
// optimize(5), the chi2 check that drops correspondences, optimize(10 | 5) and the final inlier count without host round trips.
// One Sim3 vertex (g2o::Sim3, types/sim3.h), two projection edges per correspondence with NUMERIC Jacobians exactly as g2o
// computes them for these edge types (central differences with delta 1e-9 on the vertex's oplus, core/base_binary_edge.hpp:131-197):
// the 14 perturbed Sim3 states (and their inverses) of a linearisation are built once by 14 lanes and shared through LDS, every
// lane then evaluates its edges against them.  Dense 7x7 LDL^T on one lane; LM rules as in lld_pose.hip.
#include "lld_common.h"
#include "lld_device_math.h"
#include "lld_sim3_math.h"

namespace {

using namespace lld;

constexpr int kSimThreads = 256;
constexpr int kSimWaves = kSimThreads / 64;

struct SimProblemDev {
  double f1[2], pp1[2], f2[2], pp2[2];
  double S0[8];
  int n, off;                  // correspondences, offset into the SoA
  double th2, delta;
  int fix_scale, its_first, its_more_bad, its_more_clean, min_inliers, max_trials;
};
struct SimArrays {
  const double *p1c, *p2c, *obs1, *obs2, *s1, *s2;     // [N][3], [N][3], [N][2], [N][2], [N], [N]
  double* err;                                           // [N][4] e12 (2), e21 (2) of the last computeActiveErrors
  uint8_t* dropped;                                      // [N]
};
struct SimOut { double S[8]; double chi2; int n_inliers, n_bad_first, its[2], trials[2]; };

// Dense LDL^T of the 7x7 system (LinearSolverDense): fails unless all pivots > 0.
__device__ bool solve7(const double* Hu /*28 upper, row-major packed*/, double lambda, const double* b, double* x) {
  double A[7][7];
  { int k = 0; for (int i = 0; i < 7; i++) for (int j = i; j < 7; j++) { A[j][i] = Hu[k]; k++; } }
  for (int i = 0; i < 7; i++) A[i][i] += lambda;
  bool ok = true;
  double inv[7];
  for (int j = 0; j < 7; j++) {
    double d = A[j][j];
    for (int p = 0; p < j; p++) d -= A[j][p] * A[j][p] * A[p][p];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    inv[j] = 1.0 / d;
    for (int i = j + 1; i < 7; i++) {
      double s = A[i][j];
      for (int p = 0; p < j; p++) s -= A[i][p] * A[j][p] * A[p][p];
      A[i][j] = s * inv[j];
    }
    A[j][j] = d;
  }
  double y[7];
  for (int i = 0; i < 7; i++) { double s = b[i]; for (int p = 0; p < i; p++) s -= A[i][p] * y[p]; y[i] = s; }
  for (int i = 0; i < 7; i++) y[i] *= inv[i];
  for (int i = 6; i >= 0; i--) { double s = y[i]; for (int p = i + 1; p < 7; p++) s -= A[p][i] * x[p]; x[i] = s; }
  return ok;
}

template <int N>
__device__ __forceinline__ void block_sum_n(double* v, double* lds /* [kSimWaves][N] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; i++) {
    double x = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    v[i] = x;
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; i++) lds[wave * N + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; i++) { double s = lds[i]; for (int w = 1; w < kSimWaves; w++) s += lds[w * N + i]; v[i] = s; }
}

__global__ __launch_bounds__(kSimThreads) void sim3_opt_kernel(const SimProblemDev* __restrict__ probs, SimArrays a, SimOut* __restrict__ outs) {
  __shared__ double red[kSimWaves * 36];
  __shared__ double pert[14 * 8], pert_inv[14 * 8];     // plus[0..6], minus[0..6] and their inverses
  __shared__ double cur[8], cur_inv[8], sol[20];
  const SimProblemDev& P = probs[blockIdx.x];
  const int tid = threadIdx.x, n = P.n, off = P.off;
  Sim3 S = sim3_load(P.S0);
  for (int i = tid; i < n; i += kSimThreads) { a.dropped[off + i] = 0; for (int k = 0; k < 4; k++) a.err[4 * (size_t)(off + i) + k] = 0.0; }
  __syncthreads();

  // error of edge (i, inv) at a state given with its inverse
  auto edge_err = [&](const double* st, const double* st_inv, int i, bool inv, double* e) {
    if (!inv) {                                            // EdgeSim3ProjectXYZ: obs1 - cam_map1(project(S12.map(X2)))
      const Vec3 m = sim3_map(sim3_load(st), vec3(a.p2c[3 * (size_t)(off + i)], a.p2c[3 * (size_t)(off + i) + 1], a.p2c[3 * (size_t)(off + i) + 2]));
      e[0] = a.obs1[2 * (size_t)(off + i)] - (m.x / m.z * P.f1[0] + P.pp1[0]); e[1] = a.obs1[2 * (size_t)(off + i) + 1] - (m.y / m.z * P.f1[1] + P.pp1[1]);
    } else {                                               // EdgeInverseSim3ProjectXYZ: obs2 - cam_map2(project(S12.inverse().map(X1)))
      const Vec3 m = sim3_map(sim3_load(st_inv), vec3(a.p1c[3 * (size_t)(off + i)], a.p1c[3 * (size_t)(off + i) + 1], a.p1c[3 * (size_t)(off + i) + 2]));
      e[0] = a.obs2[2 * (size_t)(off + i)] - (m.x / m.z * P.f2[0] + P.pp2[0]); e[1] = a.obs2[2 * (size_t)(off + i) + 1] - (m.y / m.z * P.f2[1] + P.pp2[1]);
    }
  };
  auto huber_w = [&](double chi, double& w) { return huber(chi, P.delta, &w); };

  double lambda = -1.0, ni = 2.0; int nBadLM = 0;
  double last_chi = 0.0;
  int its[2] = {0, 0}, trials[2] = {0, 0};
  int n_bad_first = 0, n_in = 0;
  bool finished_early = false;

  for (int round = 0; round < 2; round++) {
    const int n_its = round == 0 ? P.its_first : (n_bad_first > 0 ? P.its_more_bad : P.its_more_clean);
    // initializeOptimization(): the alive correspondences
    double cnt[1] = {0.0};
    for (int i = tid; i < n; i += kSimThreads) cnt[0] += a.dropped[off + i] ? 0.0 : 1.0;
    block_sum_n<1>(cnt, red);
    bool ok = cnt[0] > 0.5;
    for (int it = 0; it < n_its && ok; it++) {
      // ---- perturbed states of the numeric Jacobian + the current state with its inverse
      __syncthreads();
      if (tid < 14) {
        double u[7] = {0, 0, 0, 0, 0, 0, 0};
        u[tid % 7] = tid < 7 ? 1e-9 : -1e-9;
        if (P.fix_scale) u[6] = 0;
        const Sim3 Sp = sim3_mul(sim3_exp(u), S);
        sim3_store(Sp, pert + 8 * tid); sim3_store(sim3_inverse(Sp), pert_inv + 8 * tid);
      }
      if (tid == 14) { sim3_store(S, cur); sim3_store(sim3_inverse(S), cur_inv); }
      __syncthreads();
      double acc[36];
      for (int i = 0; i < 36; i++) acc[i] = 0.0;
      const double scalar = 1.0 / (2 * 1e-9);
      for (int i = tid; i < n; i += kSimThreads) {
        if (a.dropped[off + i]) continue;
        for (int side = 0; side < 2; side++) {             // e12 then e21, the reference's insertion order per correspondence
          double e[2]; edge_err(cur, cur_inv, i, side == 1, e);
          a.err[4 * (size_t)(off + i) + 2 * side] = e[0]; a.err[4 * (size_t)(off + i) + 2 * side + 1] = e[1];
          const double sg = side == 0 ? a.s1[off + i] : a.s2[off + i];
          const double chi = e[0] * (sg * e[0]) + e[1] * (sg * e[1]);
          double w; acc[35] += huber_w(chi, w);
          double J[2][7];
          for (int d = 0; d < 7; d++) {
            double ep[2], em[2];
            edge_err(pert + 8 * d, pert_inv + 8 * d, i, side == 1, ep); edge_err(pert + 8 * (7 + d), pert_inv + 8 * (7 + d), i, side == 1, em);
            J[0][d] = scalar * (ep[0] - em[0]); J[1][d] = scalar * (ep[1] - em[1]);
          }
          int k = 0;
          for (int r = 0; r < 7; r++) {
            acc[28 + r] -= w * (J[0][r] * (sg * e[0]) + J[1][r] * (sg * e[1]));
            for (int c = r; c < 7; c++) acc[k++] += J[0][r] * (w * sg) * J[0][c] + J[1][r] * (w * sg) * J[1][c];
          }
        }
      }
      block_sum_n<36>(acc, red);
      double currentChi = acc[35];
      const double iniChi = currentChi;
      if (it == 0) {
        double md = 0.0; int k = 0;
        for (int r = 0; r < 7; r++) { md = fmax(fabs(acc[k]), md); k += 7 - r; }
        lambda = 1e-5 * md; ni = 2.0; nBadLM = 0;
      }
      double rho = 0.0; int q = 0;
      do {
        __syncthreads();
        if (tid == 0) {
          double x[7];
          const bool ok2 = solve7(acc, lambda, acc + 28, x);
          if (P.fix_scale) x[6] = 0;                        // VertexSim3Expmap::oplusImpl zeroes it in place, in the solver's x, before computeScale
          const Sim3 Sn = sim3_mul(sim3_exp(x), S);
          double scale = 0.0;
          for (int j = 0; j < 7; j++) scale += x[j] * (lambda * x[j] + acc[28 + j]);
          scale += 1e-3;
          sim3_store(Sn, sol); sim3_store(sim3_inverse(Sn), sol + 8);
          sol[16] = scale; sol[17] = ok2 ? 1.0 : 0.0;
        }
        __syncthreads();
        const Sim3 Sn = sim3_load(sol);
        const double scale = sol[16]; const bool ok2 = sol[17] != 0.0;
        double c1[1] = {0.0};
        for (int i = tid; i < n; i += kSimThreads) {
          if (a.dropped[off + i]) continue;
          for (int side = 0; side < 2; side++) {
            double e[2]; edge_err(sol, sol + 8, i, side == 1, e);
            a.err[4 * (size_t)(off + i) + 2 * side] = e[0]; a.err[4 * (size_t)(off + i) + 2 * side + 1] = e[1];
            const double sg = side == 0 ? a.s1[off + i] : a.s2[off + i];
            double w; c1[0] += huber_w(e[0] * (sg * e[0]) + e[1] * (sg * e[1]), w);
          }
        }
        block_sum_n<1>(c1, red);
        const double tempChi = ok2 ? c1[0] : 1.7976931348623157e308;
        rho = (currentChi - tempChi) / scale;
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2; currentChi = tempChi; S = Sn;
        } else { lambda *= ni; ni *= 2; }
        q++; trials[round]++;
      } while (rho < 0 && q < P.max_trials);
      last_chi = currentChi;
      its[round]++;
      if (q == P.max_trials || rho == 0) ok = false;
      else {
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) ok = false;
      }
    }
    // ---- chi2 check on the errors of the last computeActiveErrors (Optimizer.cc:1795-1813 / :1830-1845)
    __syncthreads();
    double cc[2] = {0.0, 0.0};                               // newly dropped, still in
    for (int i = tid; i < n; i += kSimThreads) {
      if (a.dropped[off + i]) continue;
      const double* e = a.err + 4 * (size_t)(off + i);
      const double c12 = e[0] * (a.s1[off + i] * e[0]) + e[1] * (a.s1[off + i] * e[1]);
      const double c21 = e[2] * (a.s2[off + i] * e[2]) + e[3] * (a.s2[off + i] * e[3]);
      if (c12 > P.th2 || c21 > P.th2) { a.dropped[off + i] = 1; cc[0] += 1.0; } else cc[1] += 1.0;
    }
    block_sum_n<2>(cc, red);
    if (round == 0) {
      n_bad_first = (int)(cc[0] + 0.5);
      if (n - n_bad_first < P.min_inliers) { finished_early = true; break; }       // return 0: g2oS12 is not written back
    } else n_in = (int)(cc[1] + 0.5);
    __syncthreads();
  }
  if (tid == 0) {
    SimOut& o = outs[blockIdx.x];
    if (finished_early) { for (int k = 0; k < 8; k++) o.S[k] = P.S0[k]; o.n_inliers = 0; }
    else { sim3_store(S, o.S); o.n_inliers = n_in; }
    o.chi2 = last_chi; o.n_bad_first = n_bad_first; o.its[0] = its[0]; o.its[1] = its[1]; o.trials[0] = trials[0]; o.trials[1] = trials[1];
  }
}

inline size_t al64(size_t b) { return (b + 63) & ~size_t(63); }

}  // namespace

extern "C" void lld_sim3_params_default(lld_sim3_params* p) {
  if (!p) return;
  p->th2 = 10.0; p->fix_scale = 1; p->its_first = 5; p->its_more_bad = 10; p->its_more_clean = 5; p->min_inliers = 10; p->max_trials = 10;
}

extern "C" int lld_optimize_sim3_batch(lld_ctx* ctx, int n, const lld_sim3_problem* problems, const lld_sim3_params* params, lld_sim3_result* outs) {
  if (!ctx || n < 0 || (n > 0 && (!problems || !outs))) return LLD_ERR_INVALID;
  if (n == 0) return LLD_OK;
  lld_sim3_params prm; if (params) prm = *params; else lld_sim3_params_default(&prm);
  if (!(prm.th2 > 0) || prm.its_first < 0 || prm.its_more_bad < 0 || prm.its_more_clean < 0 || prm.max_trials <= 0) return LLD_ERR_INVALID;
  size_t N = 0;
  for (int i = 0; i < n; i++) {
    const lld_sim3_problem& p = problems[i];
    if (p.n < 0 || (!outs[i].dropped && p.n > 0)) return LLD_ERR_INVALID;
    if (p.n > 0 && (!p.p1c || !p.p2c || !p.obs1 || !p.obs2 || !p.inv_sigma2_1 || !p.inv_sigma2_2)) return LLD_ERR_INVALID;
    N += (size_t)p.n;
  }
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  const size_t NN = N + 1;
  size_t in = 0, outb = 0, dev = 0;
  const size_t o_p = in; in += al64(sizeof(SimProblemDev) * n);
  const size_t o_p1 = in; in += al64(NN * 24); const size_t o_p2 = in; in += al64(NN * 24);
  const size_t o_o1 = in; in += al64(NN * 16); const size_t o_o2 = in; in += al64(NN * 16);
  const size_t o_s1 = in; in += al64(NN * 8); const size_t o_s2 = in; in += al64(NN * 8);
  const size_t r_o = outb; outb += al64(sizeof(SimOut) * n);
  const size_t r_d = outb; outb += al64(NN);
  const size_t s_e = dev; dev += al64(NN * 32);
  void* hb; int st = lld_ctx_pinned(ctx, in + outb, &hb); if (st) return st;
  void* db; st = lld_ctx_scratch(ctx, in + outb + dev + 256, &db); if (st) return st;
  char* h = (char*)hb; char* d = (char*)db; char* h_out = h + in; char* d_out = d + in; char* d_dev = d_out + outb;
  SimProblemDev* hp = reinterpret_cast<SimProblemDev*>(h + o_p);
  size_t off = 0;
  for (int i = 0; i < n; i++) {
    const lld_sim3_problem& p = problems[i]; SimProblemDev& D = hp[i];
    std::memset(&D, 0, sizeof D);
    D.f1[0] = p.fx1; D.f1[1] = p.fy1; D.pp1[0] = p.cx1; D.pp1[1] = p.cy1; D.f2[0] = p.fx2; D.f2[1] = p.fy2; D.pp2[0] = p.cx2; D.pp2[1] = p.cy2;
    for (int k = 0; k < 4; k++) D.S0[k] = p.s12_q[k];
    for (int k = 0; k < 3; k++) D.S0[4 + k] = p.s12_t[k];
    D.S0[7] = p.s12_s;
    D.n = p.n; D.off = (int)off; D.th2 = prm.th2; D.delta = (double)(float)std::sqrt(prm.th2);         // const float deltaHuber = sqrt(th2)
    D.fix_scale = prm.fix_scale; D.its_first = prm.its_first; D.its_more_bad = prm.its_more_bad; D.its_more_clean = prm.its_more_clean;
    D.min_inliers = prm.min_inliers; D.max_trials = prm.max_trials;
    if (p.n) {
      std::memcpy(h + o_p1 + off * 24, p.p1c, (size_t)p.n * 24); std::memcpy(h + o_p2 + off * 24, p.p2c, (size_t)p.n * 24);
      std::memcpy(h + o_o1 + off * 16, p.obs1, (size_t)p.n * 16); std::memcpy(h + o_o2 + off * 16, p.obs2, (size_t)p.n * 16);
      std::memcpy(h + o_s1 + off * 8, p.inv_sigma2_1, (size_t)p.n * 8); std::memcpy(h + o_s2 + off * 8, p.inv_sigma2_2, (size_t)p.n * 8);
    }
    off += (size_t)p.n;
  }
  SimArrays A;
  A.p1c = reinterpret_cast<const double*>(d + o_p1); A.p2c = reinterpret_cast<const double*>(d + o_p2);
  A.obs1 = reinterpret_cast<const double*>(d + o_o1); A.obs2 = reinterpret_cast<const double*>(d + o_o2);
  A.s1 = reinterpret_cast<const double*>(d + o_s1); A.s2 = reinterpret_cast<const double*>(d + o_s2);
  A.err = reinterpret_cast<double*>(d_dev + s_e); A.dropped = reinterpret_cast<uint8_t*>(d_out + r_d);
  hipStream_t sm = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, in, hipMemcpyHostToDevice, sm));
  hipLaunchKernelGGL(sim3_opt_kernel, dim3(n), dim3(kSimThreads), 0, sm, reinterpret_cast<const SimProblemDev*>(d + o_p), A, reinterpret_cast<SimOut*>(d_out + r_o));
  LLD_HIP_TRY(hipGetLastError());
  LLD_HIP_TRY(hipMemcpyAsync(h_out, d_out, outb, hipMemcpyDeviceToHost, sm));
  LLD_HIP_TRY(hipStreamSynchronize(sm));
  const SimOut* ho = reinterpret_cast<const SimOut*>(h_out + r_o);
  off = 0;
  for (int i = 0; i < n; i++) {
    lld_sim3_result& o = outs[i];
    for (int k = 0; k < 4; k++) o.s12_q[k] = ho[i].S[k];
    for (int k = 0; k < 3; k++) o.s12_t[k] = ho[i].S[4 + k];
    o.s12_s = ho[i].S[7];
    o.n_inliers = ho[i].n_inliers; o.n_bad_first = ho[i].n_bad_first; o.chi2 = ho[i].chi2;
    o.lm_iterations[0] = ho[i].its[0]; o.lm_iterations[1] = ho[i].its[1]; o.lm_trials[0] = ho[i].trials[0]; o.lm_trials[1] = ho[i].trials[1];
    if (problems[i].n) std::memcpy(o.dropped, h_out + r_d + off, (size_t)problems[i].n);
    off += (size_t)problems[i].n;
  }
  return LLD_OK;
}

extern "C" int lld_optimize_sim3(lld_ctx* ctx, const lld_sim3_problem* in, const lld_sim3_params* params, lld_sim3_result* out) {
  if (!in || !out) return LLD_ERR_INVALID;
  return lld_optimize_sim3_batch(ctx, 1, in, params, out);
}
