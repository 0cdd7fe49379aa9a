// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
// (lld_slam_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the reference cannot be built here (no Eigen / OpenCV) and has no tests or
// golden vectors; this CPU restatement is pinned by the known-answer tests under tests/.
//
// lld_oracle.cpp — single-threaded CPU restatement of the reference hot path behind the same
// C structs as include/lld_amd.h (symbols lldo_*):
//   Optimizer::LocalBundleAdjustment   src/Optimizer.cc:936-1388  + src/LineOptimizer.cc
//   Optimizer::PoseOptimization        src/Optimizer.cc:562-932
//   g2o LM / block solver / Schur      Thirdparty/g2o/g2o/core/{optimization_algorithm_levenberg.cpp,
//                                      block_solver.hpp, sparse_optimizer.cpp, base_*_edge.hpp}
//   exact reduced solve                Thirdparty/g2o/g2o/solvers/linear_solver_{eigen,dense}.h (dense LDLT here)
//   ORB / LBD matching                 src/ORBmatcher.cc:1647-1663 (+ best/second loops), src/TwoFrameLineMatcher.cc:26-124
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/lld_amd.h"
#include "lldo_edges.h"
#include "lldo_lm.h"
#include "lldo_math.h"

using namespace lldo;

namespace {

// ------------------------------------------------------------------ small dense helpers
// Inverse of a d x d matrix (d <= 4) by Gauss-Jordan with partial pivoting.  The reference calls
// D->inverse() on a dynamic Eigen::MatrixXd (BlockSolverX, block_solver.hpp:391), i.e. PartialPivLU.
// Test knob (lldo_set_landmark_inverse): 1 = the same inverse through a Cholesky factor, A^-1 = L^-T L^-1 - equal in exact
// arithmetic, rounded differently.  The difference between the two is the sensitivity of a landmark to HOW (Hll + lambda I)^-1 is
// rounded; the device (Cholesky solve) is held to that spread on nearly singular blocks (tests/test_gpu_ba.py, "ill-conditioned Hll").
static int g_landmark_inverse = 0;
static bool invert_small_chol(const double* A, int d, double* Ainv) {
  double L[4][4] = {{0}}, Li[4][4] = {{0}};
  for (int j = 0; j < d; j++) {
    double s = A[j * d + j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    if (!(s > 0.0)) return false;
    L[j][j] = std::sqrt(s);
    for (int i = j + 1; i < d; i++) { double t = A[i * d + j]; for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k]; L[i][j] = t / L[j][j]; }
  }
  for (int c = 0; c < d; c++)                       // Li = L^-1, column by column (forward substitution on the identity)
    for (int i = c; i < d; i++) { double t = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; k++) t -= L[i][k] * Li[k][c]; Li[i][c] = t / L[i][i]; }
  for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) { double t = 0; for (int k = (i > j ? i : j); k < d; k++) t += Li[k][i] * Li[k][j]; Ainv[i * d + j] = t; }
  return true;
}
static bool invert_small(const double* A, int d, double* Ainv) {
  if (g_landmark_inverse == 1 && invert_small_chol(A, d, Ainv)) return true;
  double M[4][8];
  for (int i = 0; i < d; i++) {
    for (int j = 0; j < d; j++) { M[i][j] = A[i * d + j]; M[i][d + j] = (i == j) ? 1.0 : 0.0; }
  }
  for (int c = 0; c < d; c++) {
    int piv = c; double best = std::fabs(M[c][c]);
    for (int r = c + 1; r < d; r++) if (std::fabs(M[r][c]) > best) { best = std::fabs(M[r][c]); piv = r; }
    if (piv != c) for (int j = 0; j < 2 * d; j++) std::swap(M[c][j], M[piv][j]);
    const double p = M[c][c];
    for (int j = 0; j < 2 * d; j++) M[c][j] /= p;
    for (int r = 0; r < d; r++) if (r != c) {
      const double f = M[r][c];
      if (f != 0.0) for (int j = 0; j < 2 * d; j++) M[r][j] -= f * M[c][j];
    }
  }
  for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) Ainv[i * d + j] = M[i][d + j];
  return true;
}

// Dense LDL^T without pivoting on a symmetric matrix given by its upper triangle (row-major n x n,
// only j>=i read).  Stands in for Eigen::SimplicialLDLT<Upper> (linear_solver_eigen.h:94-124):
// fails only on a zero / non-finite pivot.  When `require_positive`, also fails on a negative pivot
// (Eigen::LDLT::isPositive(), linear_solver_dense.h:104-112).
static bool ldlt_solve(std::vector<double>& A, int n, const double* b, double* x, bool require_positive) {
  // in-place: L strictly lower (stored in lower part), D on diagonal.  Mirror upper into lower first.
  for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) A[(size_t)j * n + i] = A[(size_t)i * n + j];
  std::vector<double> v(n);
  for (int j = 0; j < n; j++) {
    double dj = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) { v[k] = A[(size_t)j * n + k] * A[(size_t)k * n + k]; dj -= A[(size_t)j * n + k] * v[k]; }
    if (!(std::isfinite(dj)) || dj == 0.0) return false;
    if (require_positive && dj < 0.0) return false;
    A[(size_t)j * n + j] = dj;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * v[k];
      A[(size_t)i * n + j] = s / dj;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * y[k]; y[i] = s; }
  for (int i = 0; i < n; i++) y[i] /= A[(size_t)i * n + i];
  for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k]; x[i] = s; }
  return true;
}

// ================================================================== local BA system
struct PtEdge {
  int cam, pt; double obs[3]; bool stereo; double s;
  int level; bool robust; Huber hub; double err[3];
};
struct LnEdge {
  int cam, line; double x1[2], x2[2]; double s; double bx; bool pair_stereo;
  int level; bool robust; Huber hub; double err[2]; bool removed; int obs, side;
};

struct BASystem {
  Cam cam; double lf, lcx, lcy;   // line edges: f = K(0,0), cx, cy
  int n_cams = 0, n_free = 0;
  std::vector<SE3> cams; std::vector<V3> pts; std::vector<Line> lines; std::vector<char> line_removed;
  std::vector<PtEdge> pe; std::vector<LnEdge> le;
  volatile const int* abort_flag = nullptr;

  // active set / index mapping (SparseOptimizer::initializeOptimization, buildIndexMapping)
  std::vector<int> act_pe, act_le;
  std::vector<int> cam_h, pt_h, ln_h;          // hessian block index or -1
  std::vector<int> acams, apts, alns;          // active free vertices in index order
  int np = 0, nl = 0;                          // scalar sizes (poses, landmarks)
  std::vector<int> pt_off, ln_off;             // scalar offset of landmark inside the landmark part
  // system storage
  std::vector<double> Hpp;                     // [acams][36]
  std::vector<double> Hll_p;                   // [apts][9]
  std::vector<double> Hll_l;                   // [alns][16]
  struct Blk { int cam_h; double w[24]; };     // Hpl block (6 x d), row-major 6 rows
  std::vector<std::vector<Blk>> W_p, W_l;      // per active landmark, sorted by cam_h
  std::vector<double> b, x;
  std::vector<double> diagBackupPose, diagBackupLm;
  std::vector<double> Dinv_p, Dinv_l;
  // estimate backup stack (depth 1 is enough for LM)
  std::vector<SE3> bk_cams; std::vector<V3> bk_pts; std::vector<Line> bk_lines;

  // lld_ba_params::abort_after_trials (test hook): the flag counts as raised once that many LM trials are done, over both rounds
  int abort_after = 0; const int* trials_a = nullptr; const int* trials_b = nullptr;
  bool terminate() {
    if (abort_flag && *abort_flag) return true;
    return abort_after > 0 && (trials_a ? *trials_a : 0) + (trials_b ? *trials_b : 0) >= abort_after;
  }
  size_t numUnknownVertices() { return acams.size() + apts.size() + alns.size(); }

  bool cam_fixed(int c) const { return c >= n_free; }

  void initializeOptimization(int level) {
    act_pe.clear(); act_le.clear();
    cam_h.assign(n_cams, -1); pt_h.assign(pts.size(), -1); ln_h.assign(lines.size(), -1);
    std::vector<char> cam_act(n_cams, 0), pt_act(pts.size(), 0), ln_act(lines.size(), 0);
    // landmarks are never fixed, so allVerticesFixed() is always false for binary edges
    for (size_t i = 0; i < pe.size(); i++) {
      if (level < 0 || pe[i].level == level) { act_pe.push_back((int)i); cam_act[pe[i].cam] = 1; pt_act[pe[i].pt] = 1; }
    }
    for (size_t i = 0; i < le.size(); i++) {
      if (le[i].removed) continue;
      if (level < 0 || le[i].level == level) { act_le.push_back((int)i); cam_act[le[i].cam] = 1; ln_act[le[i].line] = 1; }
    }
    acams.clear(); apts.clear(); alns.clear();
    int idx = 0;
    for (int c = 0; c < n_free; c++) if (cam_act[c]) { cam_h[c] = idx++; acams.push_back(c); }
    np = 6 * (int)acams.size();
    pt_off.assign(pts.size(), -1); ln_off.assign(lines.size(), -1);
    int off = 0;
    int li = 0;
    for (size_t p = 0; p < pts.size(); p++) if (pt_act[p]) { pt_h[p] = li++; apts.push_back((int)p); pt_off[p] = off; off += 3; }
    li = 0;
    for (size_t l = 0; l < lines.size(); l++) if (ln_act[l] && !line_removed[l]) { ln_h[l] = li++; alns.push_back((int)l); ln_off[l] = off; off += 4; }
    nl = off;
  }

  bool buildStructure() {
    Hpp.assign(acams.size() * 36, 0.0);
    Hll_p.assign(apts.size() * 9, 0.0);
    Hll_l.assign(alns.size() * 16, 0.0);
    W_p.assign(apts.size(), {}); W_l.assign(alns.size(), {});
    auto add_blk = [](std::vector<Blk>& col, int ch) {
      for (auto& bk : col) if (bk.cam_h == ch) return;
      Blk nb; nb.cam_h = ch; std::memset(nb.w, 0, sizeof nb.w);
      col.push_back(nb);
    };
    for (int ei : act_pe) { const PtEdge& e = pe[ei]; if (cam_h[e.cam] >= 0) add_blk(W_p[pt_h[e.pt]], cam_h[e.cam]); }
    for (int ei : act_le) { const LnEdge& e = le[ei]; if (cam_h[e.cam] >= 0) add_blk(W_l[ln_h[e.line]], cam_h[e.cam]); }
    auto by_cam = [](const Blk& a, const Blk& c) { return a.cam_h < c.cam_h; };
    for (auto& col : W_p) std::sort(col.begin(), col.end(), by_cam);
    for (auto& col : W_l) std::sort(col.begin(), col.end(), by_cam);
    b.assign(np + nl, 0.0); x.assign(np + nl, 0.0);
    Dinv_p.assign(apts.size() * 9, 0.0); Dinv_l.assign(alns.size() * 16, 0.0);
    return true;
  }

  void pt_edge_error(PtEdge& e) {
    const V3 Xc = se3_map(cams[e.cam], pts[e.pt]);
    if (e.stereo) stereo_error_binary(cam, Xc, e.obs, e.err);
    else mono_error(cam, Xc, e.obs, e.err);
  }
  void ln_edge_error(LnEdge& e) {
    const Line& L = lines[e.line];
    const M3 R = line_getR(L);
    const V3 X1 = scale(m3_col(R, 1), L.alpha);
    const V3 X2 = add(X1, m3_col(R, 0));
    line_error_cam(lf, lcx, lcy, e.bx, se3_map(cams[e.cam], X1), se3_map(cams[e.cam], X2), e.x1, e.x2, e.err);
  }
  void computeActiveErrors() {
    for (int ei : act_pe) pt_edge_error(pe[ei]);
    for (int ei : act_le) ln_edge_error(le[ei]);
  }
  double pe_chi2(const PtEdge& e) const { return chi2_iso(e.err, e.stereo ? 3 : 2, e.s); }
  double le_chi2(const LnEdge& e) const { return chi2_iso(e.err, 2, e.s); }
  double activeRobustChi2() {
    double chi = 0.0, rho[3];
    for (int ei : act_pe) { const PtEdge& e = pe[ei]; if (e.robust) { huber_robustify(e.hub, pe_chi2(e), rho); chi += rho[0]; } else chi += pe_chi2(e); }
    for (int ei : act_le) { const LnEdge& e = le[ei]; if (e.robust) { huber_robustify(e.hub, le_chi2(e), rho); chi += rho[0]; } else chi += le_chi2(e); }
    return chi;
  }

  // BaseBinaryEdge::constructQuadraticForm (core/base_binary_edge.hpp:54-120) with Omega = s*I.
  // A: D x da (landmark, vertex 0), B: D x 6 (camera, vertex 1)
  void quadratic_form(const double* A, int da, const double* B, int D, const double* err, double s, bool robust, const Huber& hub,
                      double* Hll, double* bl, double* Hpp_blk /*or null*/, double* bp /*or null*/, double* Wblk /*6 x da or null*/) {
    double w = 1.0;
    double omega_r[3];
    for (int i = 0; i < D; i++) omega_r[i] = -(s * err[i]);
    if (robust) {
      double rho[3];
      huber_robustify(hub, chi2_iso(err, D, s), rho);
      w = rho[1];
      for (int i = 0; i < D; i++) omega_r[i] *= rho[1];
    }
    const double ws = w * s;
    // from (landmark) is never fixed
    for (int a = 0; a < da; a++) {
      double acc = 0; for (int i = 0; i < D; i++) acc += A[i * da + a] * omega_r[i];
      bl[a] += acc;
      for (int c = 0; c < da; c++) { double h = 0; for (int i = 0; i < D; i++) h += A[i * da + a] * ws * A[i * da + c]; Hll[a * da + c] += h; }
    }
    if (Hpp_blk) {
      if (Wblk) {
        // _hessian (da x 6) = A^T wOmega B; stored here transposed as the pose-landmark block (6 x da)
        for (int r = 0; r < 6; r++) for (int a = 0; a < da; a++) { double h = 0; for (int i = 0; i < D; i++) h += B[i * 6 + r] * ws * A[i * da + a]; Wblk[r * da + a] += h; }
      }
      for (int r = 0; r < 6; r++) {
        double acc = 0; for (int i = 0; i < D; i++) acc += B[i * 6 + r] * omega_r[i];
        bp[r] += acc;
        for (int c = 0; c < 6; c++) { double h = 0; for (int i = 0; i < D; i++) h += B[i * 6 + r] * ws * B[i * 6 + c]; Hpp_blk[r * 6 + c] += h; }
      }
    }
  }

  void buildSystem() {
    std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll_p.begin(), Hll_p.end(), 0.0); std::fill(Hll_l.begin(), Hll_l.end(), 0.0);
    for (auto& col : W_p) for (auto& bk : col) std::memset(bk.w, 0, sizeof bk.w);
    for (auto& col : W_l) for (auto& bk : col) std::memset(bk.w, 0, sizeof bk.w);
    std::fill(b.begin(), b.end(), 0.0);
    double Jp[9], Jc[18], Jl[8];
    for (int ei : act_pe) {
      PtEdge& e = pe[ei];
      point_jac_binary(cam, cams[e.cam], pts[e.pt], e.stereo, Jp, Jc);
      const int D = e.stereo ? 3 : 2;
      const int ph = pt_h[e.pt], ch = cam_h[e.cam];
      double* Wb = nullptr;
      if (ch >= 0) for (auto& bk : W_p[ph]) if (bk.cam_h == ch) Wb = bk.w;
      quadratic_form(Jp, 3, Jc, D, e.err, e.s, e.robust, e.hub, &Hll_p[ph * 9], &b[np + pt_off[e.pt]],
                     ch >= 0 ? &Hpp[ch * 36] : nullptr, ch >= 0 ? &b[ch * 6] : nullptr, Wb);
    }
    for (int ei : act_le) {
      LnEdge& e = le[ei];
      line_jac_binary(lf, lcx, lcy, e.bx, cams[e.cam], lines[e.line], e.x1, e.x2, Jl, Jc);
      const int lh = ln_h[e.line], ch = cam_h[e.cam];
      double* Wb = nullptr;
      if (ch >= 0) for (auto& bk : W_l[lh]) if (bk.cam_h == ch) Wb = bk.w;
      quadratic_form(Jl, 4, Jc, 2, e.err, e.s, e.robust, e.hub, &Hll_l[lh * 16], &b[np + ln_off[e.line]],
                     ch >= 0 ? &Hpp[ch * 36] : nullptr, ch >= 0 ? &b[ch * 6] : nullptr, Wb);
    }
  }

  double maxDiagonal() {
    double m = 0.;
    for (size_t c = 0; c < acams.size(); c++) for (int j = 0; j < 6; j++) m = std::max(std::fabs(Hpp[c * 36 + j * 6 + j]), m);
    for (size_t p = 0; p < apts.size(); p++) for (int j = 0; j < 3; j++) m = std::max(std::fabs(Hll_p[p * 9 + j * 3 + j]), m);
    for (size_t l = 0; l < alns.size(); l++) for (int j = 0; j < 4; j++) m = std::max(std::fabs(Hll_l[l * 16 + j * 4 + j]), m);
    return m;
  }

  void push() { bk_cams = cams; bk_pts = pts; bk_lines = lines;
    // BaseVertex::push copies the estimate through LineParams' copy-ctor, which normalises q (types_sba.cpp:64-68)
    for (auto& l : bk_lines) l.q = quat_normalized(l.q); }
  void pop() { for (int c : acams) cams[c] = bk_cams[c]; for (int p : apts) pts[p] = bk_pts[p]; for (int l : alns) lines[l] = bk_lines[l]; }
  void discardTop() {}

  void setLambda(double lambda) {
    diagBackupPose.resize(acams.size() * 6); diagBackupLm.resize(apts.size() * 3 + alns.size() * 4);
    for (size_t c = 0; c < acams.size(); c++) for (int j = 0; j < 6; j++) { diagBackupPose[c * 6 + j] = Hpp[c * 36 + j * 6 + j]; Hpp[c * 36 + j * 6 + j] += lambda; }
    size_t k = 0;
    for (size_t p = 0; p < apts.size(); p++) for (int j = 0; j < 3; j++) { diagBackupLm[k++] = Hll_p[p * 9 + j * 3 + j]; Hll_p[p * 9 + j * 3 + j] += lambda; }
    for (size_t l = 0; l < alns.size(); l++) for (int j = 0; j < 4; j++) { diagBackupLm[k++] = Hll_l[l * 16 + j * 4 + j]; Hll_l[l * 16 + j * 4 + j] += lambda; }
  }
  void restoreDiagonal() {
    for (size_t c = 0; c < acams.size(); c++) for (int j = 0; j < 6; j++) Hpp[c * 36 + j * 6 + j] = diagBackupPose[c * 6 + j];
    size_t k = 0;
    for (size_t p = 0; p < apts.size(); p++) for (int j = 0; j < 3; j++) Hll_p[p * 9 + j * 3 + j] = diagBackupLm[k++];
    for (size_t l = 0; l < alns.size(); l++) for (int j = 0; j < 4; j++) Hll_l[l * 16 + j * 4 + j] = diagBackupLm[k++];
  }

  // BlockSolver::solve, Schur branch (core/block_solver.hpp:354-486)
  template <int d>
  void schur_landmark(const double* D, const double* bl, const std::vector<Blk>& col, double* Dinv, std::vector<double>& S, std::vector<double>& coeff) {
    invert_small(D, d, Dinv);
    double db[4];
    for (int i = 0; i < d; i++) { double a = 0; for (int j = 0; j < d; j++) a += Dinv[i * d + j] * bl[j]; db[i] = a; }
    const int n = np;
    for (size_t o = 0; o < col.size(); o++) {
      const Blk& Bi = col[o];
      double BDinv[24];
      for (int r = 0; r < 6; r++) for (int c = 0; c < d; c++) { double a = 0; for (int k = 0; k < d; k++) a += Bi.w[r * d + k] * Dinv[k * d + c]; BDinv[r * d + c] = a; }
      for (int r = 0; r < 6; r++) { double a = 0; for (int k = 0; k < d; k++) a += Bi.w[r * d + k] * db[k]; coeff[Bi.cam_h * 6 + r] += a; }
      for (size_t q = o; q < col.size(); q++) {
        const Blk& Bj = col[q];
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
          double a = 0; for (int k = 0; k < d; k++) a += BDinv[r * d + k] * Bj.w[c * d + k];
          S[(size_t)(Bi.cam_h * 6 + r) * n + (Bj.cam_h * 6 + c)] -= a;
        }
      }
    }
  }
  bool solve() {
    const int n = np;
    std::vector<double> S((size_t)n * n, 0.0), coeff(n, 0.0);
    for (size_t c = 0; c < acams.size(); c++) for (int r = 0; r < 6; r++) for (int k = 0; k < 6; k++) S[(size_t)(c * 6 + r) * n + (c * 6 + k)] = Hpp[c * 36 + r * 6 + k];
    for (size_t p = 0; p < apts.size(); p++) schur_landmark<3>(&Hll_p[p * 9], &b[np + pt_off[apts[p]]], W_p[p], &Dinv_p[p * 9], S, coeff);
    for (size_t l = 0; l < alns.size(); l++) schur_landmark<4>(&Hll_l[l * 16], &b[np + ln_off[alns[l]]], W_l[l], &Dinv_l[l * 16], S, coeff);
    std::vector<double> bschur(n);
    for (int i = 0; i < n; i++) bschur[i] = b[i] - coeff[i];
    if (n > 0) { if (!ldlt_solve(S, n, bschur.data(), x.data(), false)) return false; }
    // landmarks: xl = Dinv * (bl - B^T xp)
    for (size_t p = 0; p < apts.size(); p++) {
      double cl[3]; const int off = np + pt_off[apts[p]];
      for (int k = 0; k < 3; k++) cl[k] = b[off + k];
      for (const Blk& B : W_p[p]) for (int k = 0; k < 3; k++) { double a = 0; for (int r = 0; r < 6; r++) a += B.w[r * 3 + k] * (-x[B.cam_h * 6 + r]); cl[k] += a; }
      for (int i = 0; i < 3; i++) { double a = 0; for (int k = 0; k < 3; k++) a += Dinv_p[p * 9 + i * 3 + k] * cl[k]; x[off + i] = a; }
    }
    for (size_t l = 0; l < alns.size(); l++) {
      double cl[4]; const int off = np + ln_off[alns[l]];
      for (int k = 0; k < 4; k++) cl[k] = b[off + k];
      for (const Blk& B : W_l[l]) for (int k = 0; k < 4; k++) { double a = 0; for (int r = 0; r < 6; r++) a += B.w[r * 4 + k] * (-x[B.cam_h * 6 + r]); cl[k] += a; }
      for (int i = 0; i < 4; i++) { double a = 0; for (int k = 0; k < 4; k++) a += Dinv_l[l * 16 + i * 4 + k] * cl[k]; x[off + i] = a; }
    }
    return true;
  }
  // SparseOptimizer::update (sparse_optimizer.cpp:422-435) -> oplusImpl of each vertex
  void update() {
    for (size_t c = 0; c < acams.size(); c++) cams[acams[c]] = se3_mul(se3_exp(&x[c * 6]), cams[acams[c]]);
    for (int p : apts) { const double* u = &x[np + pt_off[p]]; pts[p] = V3{pts[p].x + u[0], pts[p].y + u[1], pts[p].z + u[2]}; }
    for (int l : alns) line_oplus(lines[l], &x[np + ln_off[l]]);
  }
  double computeScale(double lambda) {
    double scale = 0.;
    for (size_t j = 0; j < x.size(); j++) scale += x[j] * (lambda * x[j] + b[j]);
    return scale;
  }
};



}  // namespace

// ------------------------------------------------------------------ exported helpers (KATs)
extern "C" {

void lldo_se3_from_tcw_f32(const float* T, double* qt) {
  M3 R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = T[i * 4 + j];
  SE3 s = se3_from_Rt(R, V3{T[3], T[7], T[11]});
  qt[0] = s.q.x; qt[1] = s.q.y; qt[2] = s.q.z; qt[3] = s.q.w; qt[4] = s.t.x; qt[5] = s.t.y; qt[6] = s.t.z;
}
void lldo_se3_to_tcw_f32(const double* qt, float* T) {
  const M3 R = quat_to_R(Quat{qt[0], qt[1], qt[2], qt[3]});
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R.m[i][j]; }
  T[3] = (float)qt[4]; T[7] = (float)qt[5]; T[11] = (float)qt[6];
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}
void lldo_orb_inv_level_sigma2(float scale_factor, int n_levels, float* out) {
  std::vector<float> sf(n_levels), s2(n_levels);
  sf[0] = 1.0f; s2[0] = 1.0f;
  for (int i = 1; i < n_levels; i++) { sf[i] = sf[i - 1] * scale_factor; s2[i] = sf[i] * sf[i]; }
  for (int i = 0; i < n_levels; i++) out[i] = 1.0f / s2[i];
}

static SE3 qt_to_se3(const double* qt) { return SE3{Quat{qt[0], qt[1], qt[2], qt[3]}, V3{qt[4], qt[5], qt[6]}}; }
static void se3_to_qt(const SE3& s, double* qt) { qt[0] = s.q.x; qt[1] = s.q.y; qt[2] = s.q.z; qt[3] = s.q.w; qt[4] = s.t.x; qt[5] = s.t.y; qt[6] = s.t.z; }

void lldo_se3_exp(const double* u6, double* qt7) { se3_to_qt(se3_exp(u6), qt7); }
void lldo_se3_mul(const double* a, const double* b, double* out) { se3_to_qt(se3_mul(qt_to_se3(a), qt_to_se3(b)), out); }
void lldo_se3_map(const double* qt, const double* X, double* out) { V3 r = se3_map(qt_to_se3(qt), V3{X[0], X[1], X[2]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void lldo_se3_oplus(const double* qt, const double* u6, double* out) { se3_to_qt(se3_mul(se3_exp(u6), qt_to_se3(qt)), out); }
void lldo_quat_to_R(const double* q, double* R9) { M3 R = quat_to_R(Quat{q[0], q[1], q[2], q[3]}); std::memcpy(R9, R.m, sizeof R.m); }
void lldo_quat_from_R(const double* R9, double* q) { M3 R; std::memcpy(R.m, R9, sizeof R.m); Quat r = quat_from_R(R); q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
void lldo_huber(double delta, double e, double* rho3) { huber_robustify(huber_make(delta), e, rho3); }

// line5: qx,qy,qz,qw,alpha
void lldo_line_from_x0_dir(const double* X0, const double* dir, double* line5) {
  Line l = line_from_x0_dir(V3{X0[0], X0[1], X0[2]}, V3{dir[0], dir[1], dir[2]});
  line5[0] = l.q.x; line5[1] = l.q.y; line5[2] = l.q.z; line5[3] = l.q.w; line5[4] = l.alpha;
}
void lldo_line_oplus(const double* line5, const double* u4, double* out5) {
  Line l{Quat{line5[0], line5[1], line5[2], line5[3]}, line5[4]};
  line_oplus(l, u4);
  out5[0] = l.q.x; out5[1] = l.q.y; out5[2] = l.q.z; out5[3] = l.q.w; out5[4] = l.alpha;
}
void lldo_line_to_x0_dir(const double* line5, double* X0, double* dir) {
  Line l{Quat{line5[0], line5[1], line5[2], line5[3]}, line5[4]};
  const M3 R = line_getR(l);
  const V3 c0 = m3_col(R, 0), c1 = m3_col(R, 1);
  dir[0] = c0.x; dir[1] = c0.y; dir[2] = c0.z;
  X0[0] = l.alpha * c1.x; X0[1] = l.alpha * c1.y; X0[2] = l.alpha * c1.z;
}

// Point edge (binary): err[3], Jp[9], Jc[18] (rows used: 2 mono / 3 stereo)
void lldo_edge_point(const lld_camera* c, const double* qt, const double* Xw, const double* obs, int stereo, double* err, double* Jp, double* Jc) {
  Cam cam{c->fx, c->fy, c->cx, c->cy, c->bf};
  const SE3 T = qt_to_se3(qt); const V3 X{Xw[0], Xw[1], Xw[2]};
  const V3 Xc = se3_map(T, X);
  if (stereo) stereo_error_binary(cam, Xc, obs, err); else mono_error(cam, Xc, obs, err);
  if (Jp && Jc) point_jac_binary(cam, T, X, stereo != 0, Jp, Jc);
}
// Pose-only point edge: err[3], Jc[18]
void lldo_edge_point_posonly(const lld_camera* c, const double* qt, const double* Xw, const double* obs, int stereo, double* err, double* Jc) {
  Cam cam{c->fx, c->fy, c->cx, c->cy, c->bf};
  const V3 Xc = se3_map(qt_to_se3(qt), V3{Xw[0], Xw[1], Xw[2]});
  if (stereo) stereo_error_posonly(cam, Xc, obs, err); else mono_error(cam, Xc, obs, err);
  if (Jc) point_jac_posonly(cam, Xc, stereo != 0, Jc);
}
// Line edge (binary): seg = xs,ys,xe,ye ; err[2], Jl[8], Jc[12], depth_ok
void lldo_edge_line(const lld_camera* c, double bx, const double* qt, const double* line5, const double* seg, double* err, double* Jl, double* Jc, int* depth_ok) {
  const SE3 T = qt_to_se3(qt);
  Line L{Quat{line5[0], line5[1], line5[2], line5[3]}, line5[4]};
  const M3 R = line_getR(L);
  const V3 X1 = scale(m3_col(R, 1), L.alpha), X2 = add(X1, m3_col(R, 0));
  line_error_cam(c->fx, c->cx, c->cy, bx, se3_map(T, X1), se3_map(T, X2), seg, seg + 2, err);
  if (Jl && Jc) line_jac_binary(c->fx, c->cx, c->cy, bx, T, L, seg, seg + 2, Jl, Jc);
  if (depth_ok) *depth_ok = line_depth_positive(c->fx, c->cx, c->cy, bx, T, L, seg, seg + 2) ? 1 : 0;
}
void lldo_edge_line_posonly(const lld_camera* c, double bx, const double* qt, const double* X1w, const double* X2w, const double* seg, double* err, double* Jc) {
  const SE3 T = qt_to_se3(qt);
  const V3 X1{X1w[0], X1w[1], X1w[2]}, X2{X2w[0], X2w[1], X2w[2]};
  line_error_cam(c->fx, c->cx, c->cy, bx, se3_map(T, X1), se3_map(T, X2), seg, seg + 2, err);
  if (Jc) line_jac_posonly(c->fx, c->cx, c->cy, bx, T, X1, X2, seg, seg + 2, Jc);
}
void lldo_reproject_line_point(const double* X0, const double* ldir, double px, double py, double f, double cx, double cy, double* depth, double* param) {
  reproject_line_point(V3{X0[0], X0[1], X0[2]}, V3{ldir[0], ldir[1], ldir[2]}, px, py, f, cx, cy, depth, param);
}

void lldo_set_landmark_inverse(int how) { g_landmark_inverse = how; }      // test knob, see invert_small

// test hook: record the LM trajectory of the calls that follow into buf[3 * cap] (lambda used, trial chi2, accepted); returns the
// number of trials recorded so far; buf == nullptr switches it off
static lldo::LMTrace g_trace_store{nullptr, 0, 0};
int lldo_lm_trace(double* buf, int cap) {
  const int n = g_trace_store.n;
  if (!buf) { lldo::g_lm_trace = nullptr; return n; }
  g_trace_store = lldo::LMTrace{buf, cap, 0};
  lldo::g_lm_trace = &g_trace_store;
  return n;
}

// test hook: how close the last lldo_local_ba came to a classification threshold - min over the edges of |chi2 - threshold| / threshold
// at the classification between the rounds [0] and at the final one [1] (tests pick oracle-checked windows whose decisions do not hang
// on the last digits of a chi2, and assert the margin)
static double g_cls_margin[2] = {1e300, 1e300};
void lldo_last_classification_margin(double* out2) { out2[0] = g_cls_margin[0]; out2[1] = g_cls_margin[1]; }
static inline void cls_margin(int which, double chi2, double thr) {
  const double m = std::fabs(chi2 - thr) / thr;
  if (m < g_cls_margin[which]) g_cls_margin[which] = m;
}
// test hook: the OTHER side of a decision that hangs on the last digits of a chi2.  With flip[which] set, an edge whose chi2 lies within
// `margin` (relative) of its threshold at classification `which` is classified the other way - the result a run takes whose sums are
// rounded differently (the reference's follow pointer order).  tests/test_gpu_ba.py holds the device to ONE of the two results, in full.
static int g_cls_flip[2] = {0, 0};
static double g_cls_flip_margin = 0.0;
void lldo_set_classification_flip(int which, int on, double margin) { if (which >= 0 && which < 2) g_cls_flip[which] = on; g_cls_flip_margin = margin; }
static inline bool cls_over(int which, double chi2, double thr) {
  const bool over = chi2 > thr;
  return (g_cls_flip[which] && std::fabs(chi2 - thr) / thr < g_cls_flip_margin) ? !over : over;
}

void lldo_ba_params_default(lld_ba_params* p) {
  p->gamma = 1.0; p->its_round1 = 5; p->its_round2 = 15; p->ln_filter = 4; p->max_trials = 10;
  p->pcg_rel_tol = 1e-12; p->pcg_max_iter = 0; p->reduced_solver = 0; p->protocol = 0; p->robust_points = 1; p->abort_after_trials = 0; p->deterministic = 2;      // (the oracle is sequential: deterministic either way)
}
void lldo_pose_params_default(lld_pose_params* p) { p->gamma = 0.5; p->n_rounds = 4; p->its_per_round = 10; p->max_trials = 10; p->reserved = 0; }

}  // extern "C"

// ================================================================== LocalBundleAdjustment protocol
namespace {

static void ba_setup(BASystem& S, const lld_ba_window* in, const lld_ba_params* prm) {
  S.cam = Cam{in->cam.fx, in->cam.fy, in->cam.cx, in->cam.cy, in->cam.bf};
  S.lf = in->cam.fx; S.lcx = in->cam.cx; S.lcy = in->cam.cy;   // K_eig(0,0), K_eig(0,2), K_eig(1,2)  (LineOptimizer.cc:66-68)
  S.n_cams = in->n_cams; S.n_free = in->n_free_cams;
  S.cams.resize(in->n_cams);
  for (int c = 0; c < in->n_cams; c++) S.cams[c] = qt_to_se3(in->cam_qt + 7 * c);
  S.pts.resize(in->n_points);
  for (int p = 0; p < in->n_points; p++) S.pts[p] = V3{in->pt_xyz[3 * p], in->pt_xyz[3 * p + 1], in->pt_xyz[3 * p + 2]};
  S.lines.resize(in->n_lines); S.line_removed.assign(in->n_lines, 0);
  for (int l = 0; l < in->n_lines; l++)
    S.lines[l] = line_from_x0_dir(V3{in->line_x0[3 * l], in->line_x0[3 * l + 1], in->line_x0[3 * l + 2]}, V3{in->line_dir[3 * l], in->line_dir[3 * l + 1], in->line_dir[3 * l + 2]});
  // const float thHuberMono = sqrt(5.991); const float thHuberStereo = sqrt(7.815);  (Optimizer.cc:1088-1089)
  const double thHuberMono = (double)(float)std::sqrt(5.991), thHuberStereo = (double)(float)std::sqrt(7.815);
  S.pe.clear();
  for (int p = 0; p < in->n_points; p++)
    for (int o = in->pt_obs_start[p]; o < in->pt_obs_start[p + 1]; o++) {
      PtEdge e; e.cam = in->pt_obs_cam[o]; e.pt = p;
      e.obs[0] = in->pt_obs_uvr[3 * o]; e.obs[1] = in->pt_obs_uvr[3 * o + 1]; e.obs[2] = in->pt_obs_uvr[3 * o + 2];
      e.stereo = !(e.obs[2] < 0);           // if(pKFi->mvuRight[...]<0) -> mono (Optimizer.cc:1119)
      e.s = in->pt_obs_inv_sigma2[o];
      e.level = 0; e.robust = prm->protocol == 1 ? prm->robust_points != 0 : true;      // if(bRobust)  (Optimizer.cc:405,437)
      e.hub = huber_make(e.stereo ? thHuberStereo : thHuberMono);
      e.err[0] = e.err[1] = e.err[2] = 0;
      S.pe.push_back(e);
    }
  // LineOptimizer ctor (LineOptimizer.cc:28-37)
  const double thLinesStereo = thHuberStereo * prm->gamma, thLinesMono = thHuberMono * prm->gamma;
  const double infoLines = 1.0 * (prm->gamma * prm->gamma);
  // stereo_b = pKF->mbf / pKF->mK.at<float>(0,0): float / float (Optimizer.cc:1216)
  const double stereo_b = (double)((float)in->cam.bf / (float)in->cam.fx);
  S.le.clear();
  for (int l = 0; l < in->n_lines; l++)
    for (int o = in->ln_obs_start[l]; o < in->ln_obs_start[l + 1]; o++) {
      const double* L = in->ln_obs_left + 4 * o; const double* Rr = in->ln_obs_right + 4 * o;
      const bool has_right = !(Rr[0] < 0);
      for (int si = 0; si < 2; si++) {
        if (si == 1 && !has_right) continue;
        LnEdge e; e.cam = in->ln_obs_cam[o]; e.line = l; e.obs = o; e.side = si;
        const double* kl = si == 0 ? L : Rr;
        e.x1[0] = kl[0]; e.x1[1] = kl[1]; e.x2[0] = kl[2]; e.x2[1] = kl[3];
        e.bx = si == 1 ? -stereo_b : 0.0;
        e.pair_stereo = has_right;
        e.hub = huber_make(has_right ? thLinesStereo : thLinesMono);
        const double thr = reproj_thr_pyramid(1.0, in->ln_obs_octave[2 * o + si]);
        double info = infoLines; info /= thr * thr;
        e.s = info;
        if (prm->protocol == 1) {            // AddLineMinimalGlobal (Optimizer.cc:149-240): identity information, delta = thHuber3D/2.0
          e.s = 1.0;
          e.hub = huber_make((double)(float)std::sqrt(7.815) / 2.0);
        }
        e.level = 0; e.robust = true; e.removed = false;
        S.le.push_back(e);
        S.ln_edge_error(S.le.back());   // e->computeError() before addEdge (LineOptimizer.cc:114)
      }
    }
}

}  // namespace

extern "C" int lldo_local_ba(void* /*ctx*/, const lld_ba_window* in, const lld_ba_params* prm_in, volatile const int* abort_flag, lld_ba_result* out) {
  lld_ba_params prm; if (prm_in) prm = *prm_in; else lldo_ba_params_default(&prm);
  BASystem S; S.abort_flag = abort_flag;
  ba_setup(S, in, &prm);
  std::memset(&out->stats, 0, sizeof out->stats);
  auto write_back = [&](bool untouched) {
    for (int c = 0; c < in->n_cams; c++) se3_to_qt(S.cams[c], out->cam_qt + 7 * c);
    for (int p = 0; p < in->n_points; p++) { out->pt_xyz[3 * p] = S.pts[p].x; out->pt_xyz[3 * p + 1] = S.pts[p].y; out->pt_xyz[3 * p + 2] = S.pts[p].z; }
    (void)untouched;
  };
  std::memset(out->pt_obs_outlier, 0, in->n_pt_obs);
  std::memset(out->ln_edge_outlier, 0, (size_t)in->n_ln_obs * 2);
  std::memset(out->line_removed, 0, in->n_lines);
  std::memcpy(out->line_x0, in->line_x0, sizeof(double) * 3 * in->n_lines);
  std::memcpy(out->line_dir, in->line_dir, sizeof(double) * 3 * in->n_lines);
  if (abort_flag && *abort_flag) {            // Optimizer.cc:1220-1222: return before touching anything
    for (int c = 0; c < in->n_cams; c++) std::memcpy(out->cam_qt + 7 * c, in->cam_qt + 7 * c, 7 * sizeof(double));
    std::memcpy(out->pt_xyz, in->pt_xyz, sizeof(double) * 3 * in->n_points);
    out->stats.aborted = 1;
    return LLD_OK;
  }
  g_cls_margin[0] = g_cls_margin[1] = 1e300;
  LMData lm; lm.maxTrials = prm.max_trials;
  S.abort_after = prm.abort_after_trials; S.trials_a = &lm.trials;
  S.initializeOptimization(0);
  lm_optimize(S, lm, prm.its_round1);
  out->stats.chi2_round1 = lm.lastChi; out->stats.chi2_final = lm.lastChi;
  out->stats.lm_iterations[0] = lm.iterations; out->stats.lm_trials[0] = lm.trials;
  bool bDoMore = true;
  if (S.terminate()) { bDoMore = false; out->stats.aborted = 1; }        // if(pbStopFlag) if(*pbStopFlag) bDoMore = false;  (Optimizer.cc:1230-1232)
  const bool global = prm.protocol == 1;     // Optimizer::BundleAdjustment: optimize(nIterations), then "Recover optimized data" (:493-558)
  if (global) bDoMore = false;
  const double thLinesStereo = (double)(float)std::sqrt(7.815) * prm.gamma, thLinesMono = (double)(float)std::sqrt(5.991) * prm.gamma;
  if (bDoMore) {
    for (auto& e : S.pe) {
      const bool depth_pos = se3_map(S.cams[e.cam], S.pts[e.pt]).z > 0.0;
      if (cls_over(0, S.pe_chi2(e), e.stereo ? 7.815 : 5.991) || !depth_pos) e.level = 1;
      cls_margin(0, S.pe_chi2(e), e.stereo ? 7.815 : 5.991);
      e.robust = false;
    }
    // LineOptimizer::DisableOutliers (LineOptimizer.cc:129-170)
    std::vector<int> cnt(in->n_lines, 0); std::vector<char> has_edge(in->n_lines, 0);
    for (auto& e : S.le) {
      has_edge[e.line] = 1;
      double thr = thLinesStereo * thLinesStereo;
      if (!e.pair_stereo) thr = thLinesMono * thLinesMono;
      const bool depth_pos = line_depth_positive(S.lf, S.lcx, S.lcy, e.bx, S.cams[e.cam], S.lines[e.line], e.x1, e.x2);
      if (cls_over(0, S.le_chi2(e), thr) || !depth_pos) e.level = 1; else cnt[e.line] += 2;
      cls_margin(0, S.le_chi2(e), thr);
      e.robust = false;
    }
    for (int l = 0; l < in->n_lines; l++) if (has_edge[l] && cnt[l] <= prm.ln_filter) { S.line_removed[l] = 1; out->stats.n_lines_removed++; }
    for (auto& e : S.le) if (S.line_removed[e.line]) e.removed = true;
    LMData lm2; lm2.maxTrials = prm.max_trials;
    lm2.lambda = lm.lambda; lm2.ni = lm.ni; lm2.nBad = lm.nBad;   // same algorithm object; all reset at iteration 0
    S.trials_b = &lm2.trials;
    S.initializeOptimization(0);
    const int r = lm_optimize(S, lm2, prm.its_round2);
    if (r >= 0) out->stats.chi2_final = lm2.lastChi;
    out->stats.lm_iterations[1] = lm2.iterations; out->stats.lm_trials[1] = lm2.trials;
    // lld_ba_stats::aborted is the stop flag at the protocol's last poll (include/lld_amd.h); the reference itself returns void.
    // An empty active set makes optimize() return before it polls (sparse_optimizer.cpp:356-359).
    if (r >= 0 && S.terminate()) out->stats.aborted = 1;
    S.trials_b = nullptr; S.trials_a = nullptr;       // (lm2 goes out of scope)
    S.abort_after = 0;
  }
  // final classification (Optimizer.cc:1278-1329)
  if (!global) {
    int o = 0;
    for (auto& e : S.pe) {
      const bool depth_pos = se3_map(S.cams[e.cam], S.pts[e.pt]).z > 0.0;
      if (cls_over(1, S.pe_chi2(e), e.stereo ? 7.815 : 5.991) || !depth_pos) { out->pt_obs_outlier[o] = 1; out->stats.n_pt_obs_outlier++; }
      cls_margin(1, S.pe_chi2(e), e.stereo ? 7.815 : 5.991);
      o++;
    }
  }
  for (auto& e : S.le) {
    if (global) break;
    if (S.line_removed[e.line]) continue;      // GetLineData returns false: vertex deleted
    const bool depth_pos = line_depth_positive(S.lf, S.lcx, S.lcy, e.bx, S.cams[e.cam], S.lines[e.line], e.x1, e.x2);
    S.ln_edge_error(e);
    double thr = thLinesStereo * thLinesStereo;
    if (!e.pair_stereo) thr = thLinesMono * thLinesMono;
    if (cls_over(1, S.le_chi2(e), thr) || !depth_pos) { out->ln_edge_outlier[2 * e.obs + e.side] = 1; out->stats.n_ln_edge_outlier++; }
    cls_margin(1, S.le_chi2(e), thr);
  }
  write_back(false);
  for (int l = 0; l < in->n_lines; l++) {
    out->line_removed[l] = S.line_removed[l];
    if (S.line_removed[l]) continue;
    const M3 R = line_getR(S.lines[l]);
    const V3 c0 = m3_col(R, 0), c1 = m3_col(R, 1);
    out->line_dir[3 * l] = c0.x; out->line_dir[3 * l + 1] = c0.y; out->line_dir[3 * l + 2] = c0.z;
    out->line_x0[3 * l] = S.lines[l].alpha * c1.x; out->line_x0[3 * l + 1] = S.lines[l].alpha * c1.y; out->line_x0[3 * l + 2] = S.lines[l].alpha * c1.z;
  }
  return LLD_OK;
}

// One LM trial's linear algebra exposed for KAT (6): builds the system at the initial state of `in` with all
// edges active and kernels on, applies lambda, returns x (cams then points then lines, index order) and b.
extern "C" int lldo_ba_one_step(const lld_ba_window* in, const lld_ba_params* prm_in, double lambda, double* x_out, double* b_out, int* n_out, double* chi_out, double* maxdiag_out) {
  lld_ba_params prm; if (prm_in) prm = *prm_in; else lldo_ba_params_default(&prm);
  BASystem S; ba_setup(S, in, &prm);
  S.initializeOptimization(0); S.buildStructure(); S.computeActiveErrors();
  if (chi_out) *chi_out = S.activeRobustChi2();
  S.buildSystem();
  if (maxdiag_out) *maxdiag_out = S.maxDiagonal();
  S.setLambda(lambda);
  const bool ok = S.solve();
  *n_out = (int)S.x.size();
  std::memcpy(x_out, S.x.data(), sizeof(double) * S.x.size());
  std::memcpy(b_out, S.b.data(), sizeof(double) * S.b.size());
  return ok ? 0 : 1;
}

// ================================================================== PoseOptimization protocol
namespace {

struct PoseSystem {
  Cam cam; SE3 T; SE3 bk;
  struct PE { V3 Xw; double obs[3]; bool stereo; double s; int level; bool robust; Huber hub; double err[3]; };
  struct LE { V3 X1, X2; double x1[2], x2[2]; double s; double bx; int line; int level; bool robust; Huber hub; double err[2]; };
  std::vector<PE> pe; std::vector<LE> le;
  std::vector<int> act_pe, act_le;
  double H[36], b[6], x[6], diagBackup[6];
  bool terminate() { return false; }
  size_t numUnknownVertices() { return (act_pe.size() + act_le.size()) > 0 ? 1 : 0; }
  void initializeOptimization(int level) {
    act_pe.clear(); act_le.clear();
    for (size_t i = 0; i < pe.size(); i++) if (pe[i].level == level) act_pe.push_back((int)i);
    for (size_t i = 0; i < le.size(); i++) if (le[i].level == level) act_le.push_back((int)i);
  }
  bool buildStructure() { return true; }
  void pe_error(PE& e) { const V3 Xc = se3_map(T, e.Xw); if (e.stereo) stereo_error_posonly(cam, Xc, e.obs, e.err); else mono_error(cam, Xc, e.obs, e.err); }
  void le_error(LE& e) { line_error_cam(cam.fx, cam.cx, cam.cy, e.bx, se3_map(T, e.X1), se3_map(T, e.X2), e.x1, e.x2, e.err); }
  void computeActiveErrors() { for (int i : act_pe) pe_error(pe[i]); for (int i : act_le) le_error(le[i]); }
  double pe_chi2(const PE& e) const { return chi2_iso(e.err, e.stereo ? 3 : 2, e.s); }
  double le_chi2(const LE& e) const { return chi2_iso(e.err, 2, e.s); }
  double activeRobustChi2() {
    double chi = 0, rho[3];
    // edges are active in insertion order: point edges (feature order) then line edges
    for (int i : act_pe) { const PE& e = pe[i]; if (e.robust) { huber_robustify(e.hub, pe_chi2(e), rho); chi += rho[0]; } else chi += pe_chi2(e); }
    for (int i : act_le) { const LE& e = le[i]; if (e.robust) { huber_robustify(e.hub, le_chi2(e), rho); chi += rho[0]; } else chi += le_chi2(e); }
    return chi;
  }
  // BaseUnaryEdge::constructQuadraticForm (core/base_unary_edge.hpp:42-72)
  void qf(const double* A, int D, const double* err, double s, bool robust, const Huber& hub) {
    double w = 1.0;
    if (robust) { double rho[3]; huber_robustify(hub, chi2_iso(err, D, s), rho); w = rho[1]; }
    for (int r = 0; r < 6; r++) {
      double acc = 0; for (int i = 0; i < D; i++) acc += A[i * 6 + r] * (s * err[i]);
      b[r] -= w * acc;
      for (int c = 0; c < 6; c++) { double h = 0; for (int i = 0; i < D; i++) h += A[i * 6 + r] * (w * s) * A[i * 6 + c]; H[r * 6 + c] += h; }
    }
  }
  void buildSystem() {
    std::memset(H, 0, sizeof H); std::memset(b, 0, sizeof b);
    double Jc[18];
    for (int i : act_pe) { PE& e = pe[i]; point_jac_posonly(cam, se3_map(T, e.Xw), e.stereo, Jc); qf(Jc, e.stereo ? 3 : 2, e.err, e.s, e.robust, e.hub); }
    for (int i : act_le) { LE& e = le[i]; line_jac_posonly(cam.fx, cam.cx, cam.cy, e.bx, T, e.X1, e.X2, e.x1, e.x2, Jc); qf(Jc, 2, e.err, e.s, e.robust, e.hub); }
  }
  double maxDiagonal() { double m = 0; for (int j = 0; j < 6; j++) m = std::max(std::fabs(H[j * 6 + j]), m); return m; }
  void push() { bk = T; } void pop() { T = bk; } void discardTop() {}
  void setLambda(double l) { for (int j = 0; j < 6; j++) { diagBackup[j] = H[j * 6 + j]; H[j * 6 + j] += l; } }
  void restoreDiagonal() { for (int j = 0; j < 6; j++) H[j * 6 + j] = diagBackup[j]; }
  bool solve() { std::vector<double> A(H, H + 36); return ldlt_solve(A, 6, b, x, true); }   // LinearSolverDense (linear_solver_dense.h:65-113)
  void update() { T = se3_mul(se3_exp(x), T); }
  double computeScale(double lambda) { double sc = 0; for (int j = 0; j < 6; j++) sc += x[j] * (lambda * x[j] + b[j]); return sc; }
};

}  // namespace

extern "C" int lldo_pose_opt(void* /*ctx*/, const lld_pose_problem* in, const lld_pose_params* prm_in, lld_pose_result* out) {
  lld_pose_params prm; if (prm_in) prm = *prm_in; else lldo_pose_params_default(&prm);
  PoseSystem S;
  S.cam = Cam{in->cam.fx, in->cam.fy, in->cam.cx, in->cam.cy, in->cam.bf};
  const SE3 T0 = qt_to_se3(in->pose_qt);
  S.T = T0;
  // const float deltaMono = sqrt(5.991); const float deltaStereo = sqrt(7.815);  (Optimizer.cc:697-698)
  const float deltaMono = (float)std::sqrt(5.991), deltaStereo = (float)std::sqrt(7.815);
  float deltaLinesStereo = deltaStereo, deltaLinesMono = deltaMono;
  double info_lines = 1.0;
  deltaLinesStereo *= prm.gamma; deltaLinesMono *= prm.gamma; info_lines *= prm.gamma * prm.gamma;   // :706-708
  int nInitialCorrespondences = 0;
  for (int i = 0; i < in->n_points; i++) {
    PoseSystem::PE e;
    e.Xw = V3{in->pt_xw[3 * i], in->pt_xw[3 * i + 1], in->pt_xw[3 * i + 2]};
    e.obs[0] = in->pt_uvr[3 * i]; e.obs[1] = in->pt_uvr[3 * i + 1]; e.obs[2] = in->pt_uvr[3 * i + 2];
    e.stereo = !(e.obs[2] < 0);
    e.s = in->pt_inv_sigma2[i];
    e.level = 0; e.robust = true; e.hub = huber_make(e.stereo ? (double)deltaStereo : (double)deltaMono);
    e.err[0] = e.err[1] = e.err[2] = 0;
    S.pe.push_back(e);
    nInitialCorrespondences++;
    out->pt_outlier[i] = 0;
  }
  // AddLineMinOnlyPose (Optimizer.cc:562-650)
  std::vector<char> vnStereoLines;      // pushed per EDGE, later indexed by LINE index (hazard 8, :643-648 vs :898)
  const double bright = (double)(-(float)in->cam.bf / (float)in->cam.fx);   // b = -pFrame->mbf / pFrame->mK.at<float>(0,0)
  for (int i = 0; i < in->n_lines; i++) {
    const double* L = in->ln_left + 4 * i; const double* Rr = in->ln_right + 4 * i;
    const bool has_right = !(Rr[0] < 0);
    double deltaLines = deltaLinesStereo; if (!has_right) deltaLines = deltaLinesMono;
    for (int si = 0; si < 2; si++) {
      if (si == 1 && !has_right) continue;
      PoseSystem::LE e;
      const V3 X0{in->ln_x0[3 * i], in->ln_x0[3 * i + 1], in->ln_x0[3 * i + 2]}, d{in->ln_dir[3 * i], in->ln_dir[3 * i + 1], in->ln_dir[3 * i + 2]};
      e.X1 = X0; e.X2 = add(X0, d);
      const double* kl = si == 0 ? L : Rr;
      e.x1[0] = kl[0]; e.x1[1] = kl[1]; e.x2[0] = kl[2]; e.x2[1] = kl[3];
      double info = info_lines; const double thr = reproj_thr_pyramid(1.0, in->ln_octave[2 * i + si]); info /= thr * thr;
      e.s = info; e.bx = si == 1 ? bright : 0.0; e.line = i; e.level = 0; e.robust = true; e.hub = huber_make(deltaLines);
      e.err[0] = e.err[1] = 0;
      S.le.push_back(e);
      vnStereoLines.push_back(has_right ? 1 : 0);
    }
  }
  for (int i = 0; i < in->n_lines; i++) out->ln_outlier[i] = 0;
  out->lm_iterations = 0; out->lm_trials = 0; out->chi2 = 0; out->reserved = 0;
  if (nInitialCorrespondences < 3) { se3_to_qt(T0, out->pose_qt); out->n_inliers = 0; return LLD_OK; }
  const float chi2Mono[4] = {5.991f, 5.991f, 5.991f, 5.991f}, chi2Stereo[4] = {7.815f, 7.815f, 7.815f, 7.815f};
  int nBad = 0;
  LMData lm; lm.maxTrials = prm.max_trials;
  const size_t n_edges_total = S.pe.size() + S.le.size();
  for (int it = 0; it < prm.n_rounds; it++) {
    S.T = T0;                                   // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw))
    S.initializeOptimization(0);
    lm_optimize(S, lm, prm.its_per_round);
    nBad = 0;
    const int itc = it < 4 ? it : 3;
    for (size_t i = 0; i < S.pe.size(); i++) {
      PoseSystem::PE& e = S.pe[i];
      if (out->pt_outlier[i]) S.pe_error(e);
      const float chi2 = (float)S.pe_chi2(e);
      if (chi2 > (e.stereo ? chi2Stereo[itc] : chi2Mono[itc])) { out->pt_outlier[i] = 1; e.level = 1; nBad++; }
      else { out->pt_outlier[i] = 0; e.level = 0; }
      if (it == 2) e.robust = false;
    }
    if (n_edges_total < 10) break;
    for (size_t i = 0; i < S.le.size(); i++) {
      PoseSystem::LE& e = S.le[i];
      const int idx = e.line;
      S.le_error(e);
      const float chi2 = (float)S.le_chi2(e);
      double thr = deltaLinesStereo * deltaLinesStereo;       // float * float
      // int idx = vnIndexLines[i]: the line's index in the FRAME (Optimizer.cc:893), used for mvbOutlierLines and for vnStereoLines alike
      const long long fi = in->ln_frame_index ? (long long)in->ln_frame_index[idx] : (long long)idx;
      const bool st = (fi >= 0 && (size_t)fi < vnStereoLines.size()) ? vnStereoLines[(size_t)fi] != 0 : true;   // out of range = UB in the reference
      if (!st) thr = deltaLinesMono * deltaLinesMono;
      if (chi2 > thr) { out->ln_outlier[idx] = 1; e.level = 1; } else { out->ln_outlier[idx] = 0; e.level = 0; }
      if (it == 2) e.robust = false;
    }
  }
  se3_to_qt(S.T, out->pose_qt);
  out->n_inliers = nInitialCorrespondences - nBad;
  out->lm_iterations = lm.iterations; out->lm_trials = lm.trials; out->chi2 = lm.lastChi;
  return LLD_OK;
}

// ================================================================== matching
extern "C" {

// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663)
int lldo_descriptor_distance(const uint32_t* a, const uint32_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    unsigned int v = a[i] ^ b[i];
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// best / second-best loop of the Search* family (e.g. ORBmatcher.cc:76-114): strict '<', first wins ties
static inline void best2_update(int dist, int idx, int& bestDist, int& bestIdx, int& bestDist2, int& bestIdx2) {
  if (dist < bestDist) { bestDist2 = bestDist; bestIdx2 = bestIdx; bestDist = dist; bestIdx = idx; }
  else if (dist < bestDist2) { bestDist2 = dist; bestIdx2 = idx; }
}

int lldo_match_hamming256(void*, const uint32_t* q, int nq, const uint32_t* t, int nt, const uint8_t* mask,
                          int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist) {
  for (int i = 0; i < nq; i++) {
    int bd = 256, bi = -1, bd2 = 256, bi2 = -1;
    for (int j = 0; j < nt; j++) {
      if (mask && !mask[(size_t)i * nt + j]) continue;
      best2_update(lldo_descriptor_distance(q + 8 * i, t + 8 * j), j, bd, bi, bd2, bi2);
    }
    best_idx[i] = bi; best_dist[i] = bd; second_idx[i] = bi2; second_dist[i] = bd2;
  }
  return LLD_OK;
}
int lldo_match_hamming256_csr(void*, const uint32_t* q, int nq, const uint32_t* t, int /*nt*/, const int32_t* cs, const int32_t* ci,
                              int32_t* best_idx, int32_t* best_dist, int32_t* second_idx, int32_t* second_dist) {
  for (int i = 0; i < nq; i++) {
    int bd = 256, bi = -1, bd2 = 256, bi2 = -1;
    for (int k = cs[i]; k < cs[i + 1]; k++) best2_update(lldo_descriptor_distance(q + 8 * i, t + 8 * ci[k]), ci[k], bd, bi, bd2, bi2);
    best_idx[i] = bi; best_dist[i] = bd; second_idx[i] = bi2; second_dist[i] = bd2;
  }
  return LLD_OK;
}

// Build-defined LBD distance (LBDMOD is un-vendored): cv::norm(a-b) arithmetic, src/MapLine.cc:175
double lldo_l2f32(const float* a, const float* b, int dim) {
  double acc = 0;
  for (int i = 0; i < dim; i++) { const float d = a[i] - b[i]; acc += (double)d * (double)d; }
  return std::sqrt(acc);
}
int lldo_match_l2f32(void*, const float* q, int nq, const float* t, int nt, int dim, const uint8_t* mask,
                     int32_t* best_idx, double* best_dist, int32_t* second_idx, double* second_dist) {
  for (int i = 0; i < nq; i++) {
    double bd = DBL_MAX, bd2 = DBL_MAX; int bi = -1, bi2 = -1;
    for (int j = 0; j < nt; j++) {
      if (mask && !mask[(size_t)i * nt + j]) continue;
      const double d = lldo_l2f32(q + (size_t)dim * i, t + (size_t)dim * j, dim);
      if (d < bd) { bd2 = bd; bi2 = bi; bd = d; bi = j; } else if (d < bd2) { bd2 = d; bi2 = j; }
    }
    best_idx[i] = bi; best_dist[i] = bd; second_idx[i] = bi2; second_dist[i] = bd2;
  }
  return LLD_OK;
}
// TwoFrameLineMatcher::MatchLines + the descriptor part of CheckLinePair (src/TwoFrameLineMatcher.cc:26-77,112-123)
int lldo_line_match_greedy(void*, const float* dl, int nq, const float* dr, int nt, int dim, const uint8_t* gate, double tau,
                           int32_t* matches, double* match_dist) {
  std::vector<char> other(nt, 0);
  for (int j = 0; j < nq; j++) {
    double min_d = DBL_MAX, sec = DBL_MAX; int min_j = -1;
    for (int oi = 0; oi < nt; oi++) {
      if (other[oi]) continue;
      if (gate && !gate[(size_t)j * nt + oi]) continue;
      const double d = lldo_l2f32(dl + (size_t)dim * j, dr + (size_t)dim * oi, dim);
      double m = min_d; if (d < m) m = d;
      if (m < min_d && m < tau) { sec = min_d; min_d = m; min_j = oi; }
    }
    (void)sec;
    if (min_j >= 0) other[min_j] = 1;
    matches[j] = min_j;
    if (match_dist) match_dist[j] = min_j >= 0 ? min_d : DBL_MAX;
  }
  return LLD_OK;
}

}  // extern "C"

// ================================================================== Optimizer::OptimizeSim3 (src/Optimizer.cc:1656-1851)
// g2o::Sim3 (types/sim3.h), VertexSim3Expmap / EdgeSim3ProjectXYZ / EdgeInverseSim3ProjectXYZ (types/types_seven_dof_expmap.h:48-171),
// numeric Jacobians of BaseBinaryEdge (core/base_binary_edge.hpp:131-197; the MapPoint vertices are fixed), dense 7x7 LM.
namespace {

struct Sim3T { Quat r; V3 t; double s; };

// Sim3(const Vector7d& update)  (sim3.h:64-131): update = (omega, upsilon, sigma)
Sim3T sim3_exp(const double* u) {
  const V3 omega{u[0], u[1], u[2]}, upsilon{u[3], u[4], u[5]};
  const double sigma = u[6];
  const double theta = norm(omega);
  const M3 Omega = skew(omega);
  const double s = std::exp(sigma);
  const M3 Omega2 = m3_mul(Omega, Omega);
  const M3 I = m3_identity();
  M3 R;
  const double eps = 0.00001;
  double A, B, C;
  auto small_R = [&]() { const M3 OO = m3_mul(Omega, Omega); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = (I.m[i][j] + Omega.m[i][j]) + OO.m[i][j]; };
  auto big_R = [&]() {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = (I.m[i][j] + a * Omega.m[i][j]) + b * Omega2.m[i][j];
  };
  if (std::fabs(sigma) < eps) {
    C = 1;
    if (theta < eps) { A = 1. / 2.; B = 1. / 6.; small_R(); }
    else {
      const double theta2 = theta * theta;
      A = (1 - std::cos(theta)) / (theta2);
      B = (theta - std::sin(theta)) / (theta2 * theta);
      big_R();
    }
  } else {
    C = (s - 1) / sigma;
    if (theta < eps) {
      const double sigma2 = sigma * sigma;
      A = ((sigma - 1) * s + 1) / sigma2;
      B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma);
      small_R();
    } else {
      big_R();
      const double a = s * std::sin(theta), b = s * std::cos(theta);
      const double theta2 = theta * theta, sigma2 = sigma * sigma;
      const double c = theta2 + sigma2;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  Sim3T r; r.r = quat_from_R(R); r.s = s;
  M3 W;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W.m[i][j] = (A * Omega.m[i][j] + B * Omega2.m[i][j]) + C * I.m[i][j];
  r.t = m3_mulv(W, upsilon);
  return r;
}
V3 sim3_map(const Sim3T& S, const V3& x) { return add(scale(quat_rot(S.r, x), S.s), S.t); }                       // s*(r*xyz) + t
Sim3T sim3_mul(const Sim3T& a, const Sim3T& b) { Sim3T r; r.r = quat_mul(a.r, b.r); r.t = add(scale(quat_rot(a.r, b.t), a.s), a.t); r.s = a.s * b.s; return r; }
Sim3T sim3_inverse(const Sim3T& a) {
  const Quat c{-a.r.x, -a.r.y, -a.r.z, a.r.w};
  Sim3T r; r.r = c; r.t = quat_rot(c, scale(a.t, -1. / a.s)); r.s = 1. / a.s;
  return r;
}

struct Sim3System {
  double f1[2], pp1[2], f2[2], pp2[2];
  bool fix_scale;
  Sim3T S, bk;
  struct Edge { V3 X; double obs[2]; double s; bool inverse; bool alive; Huber hub; double err[2]; };
  std::vector<Edge> e;              // e12_0, e21_0, e12_1, e21_1, ...
  std::vector<int> act;
  double H[49], b[7], x[7], diagBackup[7];
  bool terminate() { return false; }
  size_t numUnknownVertices() { return act.empty() ? 0 : 1; }
  void initializeOptimization() { act.clear(); for (size_t i = 0; i < e.size(); i++) if (e[i].alive) act.push_back((int)i); }
  bool buildStructure() { return true; }
  void edge_error(const Sim3T& T, const Edge& ed, double* err) const {
    const V3 m = ed.inverse ? sim3_map(sim3_inverse(T), ed.X) : sim3_map(T, ed.X);
    const double px = m.x / m.z, py = m.y / m.z;                                                                 // project()
    const double* f = ed.inverse ? f2 : f1; const double* pp = ed.inverse ? pp2 : pp1;
    err[0] = ed.obs[0] - (px * f[0] + pp[0]); err[1] = ed.obs[1] - (py * f[1] + pp[1]);
  }
  void computeActiveErrors() { for (int i : act) edge_error(S, e[i], e[i].err); }
  double activeRobustChi2() {
    double chi = 0, rho[3];
    for (int i : act) { huber_robustify(e[i].hub, chi2_iso(e[i].err, 2, e[i].s), rho); chi += rho[0]; }
    return chi;
  }
  Sim3T oplus(const double* upd) const { double u[7]; std::memcpy(u, upd, sizeof u); if (fix_scale) u[6] = 0; return sim3_mul(sim3_exp(u), S); }
  void buildSystem() {
    std::memset(H, 0, sizeof H); std::memset(b, 0, sizeof b);
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    Sim3T plus[7], minus[7];
    for (int d = 0; d < 7; d++) { double add_v[7] = {0, 0, 0, 0, 0, 0, 0}; add_v[d] = delta; plus[d] = oplus(add_v); add_v[d] = -delta; minus[d] = oplus(add_v); }
    for (int i : act) {
      const Edge& ed = e[i];
      double J[2][7];
      for (int d = 0; d < 7; d++) {
        double ep[2], em[2];
        edge_error(plus[d], ed, ep); edge_error(minus[d], ed, em);
        J[0][d] = scalar * (ep[0] - em[0]); J[1][d] = scalar * (ep[1] - em[1]);
      }
      double rho[3]; huber_robustify(ed.hub, chi2_iso(ed.err, 2, ed.s), rho);
      const double w = rho[1];
      for (int r = 0; r < 7; r++) {
        double acc = 0; for (int k = 0; k < 2; k++) acc += J[k][r] * (ed.s * ed.err[k]);
        b[r] -= w * acc;
        for (int c = 0; c < 7; c++) { double h = 0; for (int k = 0; k < 2; k++) h += J[k][r] * (w * ed.s) * J[k][c]; H[r * 7 + c] += h; }
      }
    }
  }
  double maxDiagonal() { double m = 0; for (int j = 0; j < 7; j++) m = std::max(std::fabs(H[j * 7 + j]), m); return m; }
  void push() { bk = S; } void pop() { S = bk; } void discardTop() {}
  void setLambda(double l) { for (int j = 0; j < 7; j++) { diagBackup[j] = H[j * 7 + j]; H[j * 7 + j] += l; } }
  void restoreDiagonal() { for (int j = 0; j < 7; j++) H[j * 7 + j] = diagBackup[j]; }
  bool solve() { std::vector<double> A(H, H + 49); return ldlt_solve(A, 7, b, x, true); }
  void update() { if (fix_scale) x[6] = 0; S = oplus(x); }   // oplusImpl zeroes update[6] in place, i.e. in the solver's x (types_seven_dof_expmap.h:62-65)
  double computeScale(double lambda) { double sc = 0; for (int j = 0; j < 7; j++) sc += x[j] * (lambda * x[j] + b[j]); return sc; }
};

}  // namespace

extern "C" void lldo_sim3_params_default(lld_sim3_params* p) {
  p->th2 = 10.0; p->fix_scale = 1; p->its_first = 5; p->its_more_bad = 10; p->its_more_clean = 5; p->min_inliers = 10; p->max_trials = 10;
}
extern "C" void lldo_sim3_exp(const double* u7, double* qts8) {
  const Sim3T s = sim3_exp(u7);
  qts8[0] = s.r.x; qts8[1] = s.r.y; qts8[2] = s.r.z; qts8[3] = s.r.w; qts8[4] = s.t.x; qts8[5] = s.t.y; qts8[6] = s.t.z; qts8[7] = s.s;
}

extern "C" int lldo_optimize_sim3(void* /*ctx*/, const lld_sim3_problem* in, const lld_sim3_params* prm_in, lld_sim3_result* out) {
  lld_sim3_params prm; if (prm_in) prm = *prm_in; else lldo_sim3_params_default(&prm);
  Sim3System Y;
  Y.f1[0] = in->fx1; Y.f1[1] = in->fy1; Y.pp1[0] = in->cx1; Y.pp1[1] = in->cy1;
  Y.f2[0] = in->fx2; Y.f2[1] = in->fy2; Y.pp2[0] = in->cx2; Y.pp2[1] = in->cy2;
  Y.fix_scale = prm.fix_scale != 0;
  Y.S.r = Quat{in->s12_q[0], in->s12_q[1], in->s12_q[2], in->s12_q[3]}; Y.S.t = V3{in->s12_t[0], in->s12_t[1], in->s12_t[2]}; Y.S.s = in->s12_s;
  const Sim3T S0 = Y.S;
  const double deltaHuber = (double)(float)std::sqrt(prm.th2);                     // const float deltaHuber = sqrt(th2)
  const int n = in->n;
  for (int i = 0; i < n; i++) {
    Sim3System::Edge a; a.X = V3{in->p2c[3 * i], in->p2c[3 * i + 1], in->p2c[3 * i + 2]}; a.obs[0] = in->obs1[2 * i]; a.obs[1] = in->obs1[2 * i + 1];
    a.s = in->inv_sigma2_1[i]; a.inverse = false; a.alive = true; a.hub = huber_make(deltaHuber); a.err[0] = a.err[1] = 0;
    Sim3System::Edge c; c.X = V3{in->p1c[3 * i], in->p1c[3 * i + 1], in->p1c[3 * i + 2]}; c.obs[0] = in->obs2[2 * i]; c.obs[1] = in->obs2[2 * i + 1];
    c.s = in->inv_sigma2_2[i]; c.inverse = true; c.alive = true; c.hub = huber_make(deltaHuber); c.err[0] = c.err[1] = 0;
    Y.e.push_back(a); Y.e.push_back(c);
  }
  auto store = [&](const Sim3T& S) {
    out->s12_q[0] = S.r.x; out->s12_q[1] = S.r.y; out->s12_q[2] = S.r.z; out->s12_q[3] = S.r.w;
    out->s12_t[0] = S.t.x; out->s12_t[1] = S.t.y; out->s12_t[2] = S.t.z; out->s12_s = S.s;
  };
  for (int i = 0; i < n; i++) out->dropped[i] = 0;
  out->n_inliers = 0; out->n_bad_first = 0; out->lm_iterations[0] = out->lm_iterations[1] = 0; out->lm_trials[0] = out->lm_trials[1] = 0; out->chi2 = 0;
  store(S0);
  LMData lm; lm.maxTrials = prm.max_trials;
  Y.initializeOptimization();
  lm_optimize(Y, lm, prm.its_first);
  out->lm_iterations[0] = lm.iterations; out->lm_trials[0] = lm.trials; out->chi2 = lm.lastChi;
  int nBad = 0;
  for (int i = 0; i < n; i++) {
    Sim3System::Edge& e12 = Y.e[2 * i]; Sim3System::Edge& e21 = Y.e[2 * i + 1];
    if (chi2_iso(e12.err, 2, e12.s) > prm.th2 || chi2_iso(e21.err, 2, e21.s) > prm.th2) { out->dropped[i] = 1; e12.alive = false; e21.alive = false; nBad++; }
  }
  out->n_bad_first = nBad;
  const int nMoreIterations = nBad > 0 ? prm.its_more_bad : prm.its_more_clean;
  if (n - nBad < prm.min_inliers) return LLD_OK;                                     // return 0; g2oS12 untouched
  LMData lm2; lm2.maxTrials = prm.max_trials; lm2.lambda = lm.lambda; lm2.ni = lm.ni; lm2.nBad = lm.nBad;
  Y.initializeOptimization();
  lm_optimize(Y, lm2, nMoreIterations);
  out->lm_iterations[1] = lm2.iterations; out->lm_trials[1] = lm2.trials; if (lm2.iterations > 0) out->chi2 = lm2.lastChi;
  int nIn = 0;
  for (int i = 0; i < n; i++) {
    if (out->dropped[i]) continue;
    const Sim3System::Edge& e12 = Y.e[2 * i]; const Sim3System::Edge& e21 = Y.e[2 * i + 1];
    if (chi2_iso(e12.err, 2, e12.s) > prm.th2 || chi2_iso(e21.err, 2, e21.s) > prm.th2) out->dropped[i] = 1; else nIn++;
  }
  store(Y.S);
  out->n_inliers = nIn;
  return LLD_OK;
}

// ================================================================== Optimizer::OptimizeEssentialGraph (src/Optimizer.cc:1391-1654)
// EdgeSim3 (types_seven_dof_expmap.h:99-127): error = log(C * v1 * v2^-1), numeric Jacobians for both vertices, identity information;
// Sim3::log (sim3.h:137-212) incl. the 3x3 partial-pivoting LU of W.lu().solve(t); dense LDL^T of the 7N x 7N system.
namespace {

V3 solve3_lu(const M3& Win, const V3& rhs) {          // Eigen::PartialPivLU<Matrix3d>::solve
  double A[3][4];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = Win.m[i][j]; A[i][3] = at(rhs, i); }
  for (int k = 0; k < 3; k++) {
    int piv = k; double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < 3; i++) if (std::fabs(A[i][k]) > best) { best = std::fabs(A[i][k]); piv = i; }
    if (piv != k) for (int j = 0; j < 4; j++) std::swap(A[k][j], A[piv][j]);
    for (int i = k + 1; i < 3; i++) { const double f = A[i][k] / A[k][k]; for (int j = k; j < 4; j++) A[i][j] -= f * A[k][j]; }
  }
  double x[3];
  for (int i = 2; i >= 0; i--) { double s0 = A[i][3]; for (int j = i + 1; j < 3; j++) s0 -= A[i][j] * x[j]; x[i] = s0 / A[i][i]; }
  return V3{x[0], x[1], x[2]};
}

void sim3_log(const Sim3T& S, double* res) {
  const double sigma = std::log(S.s);
  const M3 R = quat_to_R(S.r);
  const double d = 0.5 * (R.m[0][0] + R.m[1][1] + R.m[2][2] - 1);
  const V3 dR{R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]};
  const double eps = 0.00001;
  V3 omega; double A, B, C;
  if (std::fabs(sigma) < eps) {
    C = 1;
    if (d > 1 - eps) { omega = scale(dR, 0.5); A = 1. / 2.; B = 1. / 6.; }
    else {
      const double theta = std::acos(d), theta2 = theta * theta;
      omega = scale(dR, theta / (2 * std::sqrt(1 - d * d)));
      A = (1 - std::cos(theta)) / (theta2);
      B = (theta - std::sin(theta)) / (theta2 * theta);
    }
  } else {
    C = (S.s - 1) / sigma;
    if (d > 1 - eps) {
      const double sigma2 = sigma * sigma;
      omega = scale(dR, 0.5);
      A = ((sigma - 1) * S.s + 1) / (sigma2);
      B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
    } else {
      const double theta = std::acos(d);
      omega = scale(dR, theta / (2 * std::sqrt(1 - d * d)));
      const double theta2 = theta * theta;
      const double a = S.s * std::sin(theta), b = S.s * std::cos(theta);
      const double c = theta2 + sigma * sigma;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  const M3 Omega = skew(omega), OO = m3_mul(Omega, Omega), I = m3_identity();
  M3 W;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W.m[i][j] = (A * Omega.m[i][j] + B * OO.m[i][j]) + C * I.m[i][j];
  const V3 upsilon = solve3_lu(W, S.t);
  res[0] = omega.x; res[1] = omega.y; res[2] = omega.z; res[3] = upsilon.x; res[4] = upsilon.y; res[5] = upsilon.z; res[6] = sigma;
}

struct PoseGraphSystem {
  bool fix_scale = true;
  std::vector<Sim3T> V, bk;
  std::vector<int> hidx;                                  // hessian index of a vertex, -1 = fixed
  struct Edge { int i, j; Sim3T C; double err[7]; };
  std::vector<Edge> E;
  int nu = 0;                                             // unknown vertices
  std::vector<double> H, b, x, diagBackup;
  bool terminate() { return false; }
  size_t numUnknownVertices() { return (size_t)nu; }
  bool buildStructure() { return true; }
  void edge_error(const Sim3T& v1, const Sim3T& v2, const Sim3T& C, double* err) const { sim3_log(sim3_mul(sim3_mul(C, v1), sim3_inverse(v2)), err); }
  void computeActiveErrors() { for (auto& e : E) edge_error(V[e.i], V[e.j], e.C, e.err); }
  double activeRobustChi2() { double chi = 0; for (auto& e : E) { double c = 0; for (int k = 0; k < 7; k++) c += e.err[k] * e.err[k]; chi += c; } return chi; }
  Sim3T oplus(const Sim3T& S, const double* upd) const { double u[7]; std::memcpy(u, upd, sizeof u); if (fix_scale) u[6] = 0; return sim3_mul(sim3_exp(u), S); }
  void buildSystem() {
    const int n = 7 * nu;
    std::fill(H.begin(), H.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    for (auto& e : E) {
      double J[2][7][7];                                  // [vertex][error row][dof]
      for (int v = 0; v < 2; v++) {
        const int vid = v == 0 ? e.i : e.j;
        if (hidx[vid] < 0) continue;
        for (int d = 0; d < 7; d++) {
          double add_v[7] = {0, 0, 0, 0, 0, 0, 0}, ep[7], em[7];
          add_v[d] = delta; const Sim3T P = oplus(V[vid], add_v);
          add_v[d] = -delta; const Sim3T M = oplus(V[vid], add_v);
          if (v == 0) { edge_error(P, V[e.j], e.C, ep); edge_error(M, V[e.j], e.C, em); }
          else { edge_error(V[e.i], P, e.C, ep); edge_error(V[e.i], M, e.C, em); }
          for (int r = 0; r < 7; r++) J[v][r][d] = scalar * (ep[r] - em[r]);
        }
      }
      // BaseBinaryEdge::constructQuadraticForm without a robust kernel, Omega = I (core/base_binary_edge.hpp:54-120)
      for (int v = 0; v < 2; v++) {
        const int hv = hidx[v == 0 ? e.i : e.j];
        if (hv < 0) continue;
        for (int r = 0; r < 7; r++) {
          double acc = 0; for (int k = 0; k < 7; k++) acc += J[v][k][r] * e.err[k];
          b[7 * hv + r] -= acc;
          for (int c = 0; c < 7; c++) { double h = 0; for (int k = 0; k < 7; k++) h += J[v][k][r] * J[v][k][c]; H[(size_t)(7 * hv + r) * n + 7 * hv + c] += h; }
        }
      }
      const int hi = hidx[e.i], hj = hidx[e.j];
      if (hi >= 0 && hj >= 0)
        for (int r = 0; r < 7; r++)
          for (int c = 0; c < 7; c++) {
            double h = 0; for (int k = 0; k < 7; k++) h += J[0][k][r] * J[1][k][c];
            H[(size_t)(7 * hi + r) * n + 7 * hj + c] += h; H[(size_t)(7 * hj + c) * n + 7 * hi + r] += h;
          }
    }
  }
  double maxDiagonal() { double m = 0; const int n = 7 * nu; for (int j = 0; j < n; j++) m = std::max(std::fabs(H[(size_t)j * n + j]), m); return m; }
  void push() { bk = V; } void pop() { V = bk; } void discardTop() {}
  void setLambda(double l) { const int n = 7 * nu; for (int j = 0; j < n; j++) { diagBackup[j] = H[(size_t)j * n + j]; H[(size_t)j * n + j] += l; } }
  void restoreDiagonal() { const int n = 7 * nu; for (int j = 0; j < n; j++) H[(size_t)j * n + j] = diagBackup[j]; }
  bool solve() { std::vector<double> A(H); return ldlt_solve(A, 7 * nu, b.data(), x.data(), true); }
  void update() { for (size_t v = 0; v < V.size(); v++) if (hidx[v] >= 0) { double* u = &x[7 * hidx[v]]; if (fix_scale) u[6] = 0; V[v] = oplus(V[v], u); } }
  double computeScale(double lambda) { double sc = 0; for (int j = 0; j < 7 * nu; j++) sc += x[j] * (lambda * x[j] + b[j]); return sc; }
};

}  // namespace

extern "C" void lldo_pose_graph_params_default(lld_pose_graph_params* p) {
  p->iterations = 15; p->fix_scale = 1; p->lambda_init = 1e-16; p->max_trials = 10; p->pcg_max_iter = 0; p->pcg_rel_tol = 1e-12;
}
extern "C" void lldo_sim3_log(const double* qts8, double* u7) {
  Sim3T s; s.r = Quat{qts8[0], qts8[1], qts8[2], qts8[3]}; s.t = V3{qts8[4], qts8[5], qts8[6]}; s.s = qts8[7];
  sim3_log(s, u7);
}
extern "C" int lldo_optimize_essential_graph(void* /*ctx*/, const lld_pose_graph* g, const lld_pose_graph_params* prm_in, lld_pose_graph_result* out) {
  lld_pose_graph_params prm; if (prm_in) prm = *prm_in; else lldo_pose_graph_params_default(&prm);
  PoseGraphSystem Y; Y.fix_scale = prm.fix_scale != 0;
  const int N = g->n_vertices;
  Y.V.resize(N); Y.hidx.assign(N, -1);
  for (int v = 0; v < N; v++) {
    const double* s = g->sim3 + 8 * v;
    Y.V[v].r = Quat{s[0], s[1], s[2], s[3]}; Y.V[v].t = V3{s[4], s[5], s[6]}; Y.V[v].s = s[7];
    if (!(g->fixed && g->fixed[v])) Y.hidx[v] = Y.nu++;
  }
  for (int e = 0; e < g->n_edges; e++) {
    PoseGraphSystem::Edge ed; ed.i = g->edge_i[e]; ed.j = g->edge_j[e];
    const double* s = g->edge_sji + 8 * e;
    ed.C.r = Quat{s[0], s[1], s[2], s[3]}; ed.C.t = V3{s[4], s[5], s[6]}; ed.C.s = s[7];
    std::memset(ed.err, 0, sizeof ed.err);
    Y.E.push_back(ed);
  }
  const size_t n = 7 * (size_t)Y.nu;
  Y.H.assign(n * n, 0.0); Y.b.assign(n, 0.0); Y.x.assign(n, 0.0); Y.diagBackup.assign(n, 0.0);
  LMData lm; lm.maxTrials = prm.max_trials; lm.userLambdaInit = prm.lambda_init;
  if (!Y.E.empty()) lm_optimize(Y, lm, prm.iterations);
  for (int v = 0; v < N; v++) {
    double* s = out->sim3 + 8 * v;
    s[0] = Y.V[v].r.x; s[1] = Y.V[v].r.y; s[2] = Y.V[v].r.z; s[3] = Y.V[v].r.w; s[4] = Y.V[v].t.x; s[5] = Y.V[v].t.y; s[6] = Y.V[v].t.z; s[7] = Y.V[v].s;
  }
  out->chi2 = lm.lastChi; out->lm_iterations = lm.iterations; out->lm_trials = lm.trials; out->pcg_iterations = 0; out->solver_used = 0;
  return LLD_OK;
}

// ------------------------------------------------------------------ struct-free entry points (TEST INFRASTRUCTURE, round 4)
// lldo_local_ba / lldo_pose_opt take the ABI's structs, and until round 4 the Python tests filled those structs for the oracle with the
// SAME marshalling code as for the device (lld_slam_amd/host.py: Window.to_c, PoseFrame.to_c): a wrong stride or a swapped array there
// would have reached both sides alike.  These two take every array and every count as a plain argument and build the structs HERE, from
// the header: oracle/oracle_py.py passes the caller's numpy arrays one by one, with code of its own.
extern "C" int lldo_local_ba_flat(const double* cam5, int n_cams, int n_free_cams, const double* cam_qt,
                                  int n_points, const double* pt_xyz, const int32_t* pt_obs_start, int n_pt_obs, const int32_t* pt_obs_cam,
                                  const double* pt_obs_uvr, const double* pt_obs_inv_sigma2,
                                  int n_lines, const double* line_x0, const double* line_dir, const int32_t* ln_obs_start, int n_ln_obs,
                                  const int32_t* ln_obs_cam, const double* ln_obs_left, const double* ln_obs_right, const int32_t* ln_obs_octave,
                                  double gamma, int its_round1, int its_round2, int ln_filter, int max_trials, int protocol, int robust_points,
                                  int abort_after_trials, int abort_flag_value,
                                  double* o_cam_qt, double* o_pt_xyz, double* o_line_x0, double* o_line_dir, uint8_t* o_pt_obs_outlier,
                                  uint8_t* o_ln_edge_outlier, uint8_t* o_line_removed, double* o_stats12) {
  lld_ba_window w; std::memset(&w, 0, sizeof w);
  w.cam.fx = cam5[0]; w.cam.fy = cam5[1]; w.cam.cx = cam5[2]; w.cam.cy = cam5[3]; w.cam.bf = cam5[4];
  w.n_cams = n_cams; w.n_free_cams = n_free_cams; w.cam_qt = cam_qt;
  w.n_points = n_points; w.pt_xyz = pt_xyz; w.pt_obs_start = pt_obs_start; w.n_pt_obs = n_pt_obs; w.pt_obs_cam = pt_obs_cam;
  w.pt_obs_uvr = pt_obs_uvr; w.pt_obs_inv_sigma2 = pt_obs_inv_sigma2;
  w.n_lines = n_lines; w.line_x0 = line_x0; w.line_dir = line_dir; w.ln_obs_start = ln_obs_start; w.n_ln_obs = n_ln_obs;
  w.ln_obs_cam = ln_obs_cam; w.ln_obs_left = ln_obs_left; w.ln_obs_right = ln_obs_right; w.ln_obs_octave = ln_obs_octave;
  lld_ba_params p; lldo_ba_params_default(&p);
  p.gamma = gamma;
  if (its_round1 >= 0) p.its_round1 = its_round1;
  if (its_round2 >= 0) p.its_round2 = its_round2;
  if (ln_filter >= 0) p.ln_filter = ln_filter;
  if (max_trials >= 0) p.max_trials = max_trials;
  if (protocol >= 0) p.protocol = protocol;
  if (robust_points >= 0) p.robust_points = robust_points;
  if (abort_after_trials >= 0) p.abort_after_trials = abort_after_trials;
  lld_ba_result r; std::memset(&r, 0, sizeof r);
  r.cam_qt = o_cam_qt; r.pt_xyz = o_pt_xyz; r.line_x0 = o_line_x0; r.line_dir = o_line_dir;
  r.pt_obs_outlier = o_pt_obs_outlier; r.ln_edge_outlier = o_ln_edge_outlier; r.line_removed = o_line_removed;
  volatile int flag = abort_flag_value;
  const int st = lldo_local_ba(nullptr, &w, &p, &flag, &r);
  o_stats12[0] = r.stats.chi2_round1; o_stats12[1] = r.stats.chi2_final;
  o_stats12[2] = r.stats.lm_iterations[0]; o_stats12[3] = r.stats.lm_iterations[1]; o_stats12[4] = r.stats.lm_trials[0]; o_stats12[5] = r.stats.lm_trials[1];
  o_stats12[6] = r.stats.pcg_iterations; o_stats12[7] = r.stats.n_pt_obs_outlier; o_stats12[8] = r.stats.n_ln_edge_outlier;
  o_stats12[9] = r.stats.n_lines_removed; o_stats12[10] = r.stats.aborted; o_stats12[11] = 0.0;
  return st;
}

extern "C" int lldo_pose_opt_flat(const double* cam5, const double* pose_qt, int n_points, const double* pt_xw, const double* pt_uvr, const double* pt_inv_sigma2,
                                  int n_lines, const double* ln_x0, const double* ln_dir, const double* ln_left, const double* ln_right,
                                  const int32_t* ln_octave, const int32_t* ln_frame_index /* may be null */,
                                  double gamma, int n_rounds, int its_per_round, int max_trials,
                                  double* o_pose_qt, uint8_t* o_pt_outlier, uint8_t* o_ln_outlier, double* o_stats4 /* n_inliers, lm_iterations, lm_trials, chi2 */) {
  lld_pose_problem f; std::memset(&f, 0, sizeof f);
  f.cam.fx = cam5[0]; f.cam.fy = cam5[1]; f.cam.cx = cam5[2]; f.cam.cy = cam5[3]; f.cam.bf = cam5[4];
  for (int i = 0; i < 7; i++) f.pose_qt[i] = pose_qt[i];
  f.n_points = n_points; f.pt_xw = pt_xw; f.pt_uvr = pt_uvr; f.pt_inv_sigma2 = pt_inv_sigma2;
  f.n_lines = n_lines; f.ln_x0 = ln_x0; f.ln_dir = ln_dir; f.ln_left = ln_left; f.ln_right = ln_right; f.ln_octave = ln_octave; f.ln_frame_index = ln_frame_index;
  lld_pose_params p; lldo_pose_params_default(&p);
  p.gamma = gamma;
  if (n_rounds >= 0) p.n_rounds = n_rounds;
  if (its_per_round >= 0) p.its_per_round = its_per_round;
  if (max_trials >= 0) p.max_trials = max_trials;
  lld_pose_result r; std::memset(&r, 0, sizeof r);
  r.pt_outlier = o_pt_outlier; r.ln_outlier = o_ln_outlier;
  const int st = lldo_pose_opt(nullptr, &f, &p, &r);
  for (int i = 0; i < 7; i++) o_pose_qt[i] = r.pose_qt[i];
  o_stats4[0] = r.n_inliers; o_stats4[1] = r.lm_iterations; o_stats4[2] = r.lm_trials; o_stats4[3] = r.chi2;
  return st;
}
