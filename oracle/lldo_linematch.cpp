// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
// (lld_slam_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the reference cannot be built here (no Eigen / OpenCV / LBDMOD) and has no tests or golden
// vectors; this restatement is pinned by the known-answer tests under tests/.
//
// lldo_linematch.cpp — literal restatement of the stereo line association:
//   TwoFrameLineMatcher::{MatchLines, CheckLinePair}      src/TwoFrameLineMatcher.cc:26-124
//   GetNormalizedLineEq / vgl::NormalizedLineEquation      src/LineMatching.cc:249-254, src/vgl.cc:578-585
//   LineLength                                             src/LineMatching.cc:50-59
//   GetTForRight                                           src/LineMatching.cc:228-237
//   vgl::TriangulateLine                                   src/vgl.cc:78-108
//   ReprojectKeyLineTo3D / vgl::MapPoint                   src/LineMatching.cc:277-292, src/vgl.cc:587-590
//   vgl::ReprojectLinePointTo3D                            src/vgl.cc:336-346
//   Tracking::AddLinesFrom                                src/Tracking.cc:996-1124
//   Tracking::MatchLinesLastKF                            src/Tracking.cc:1449-1611
//   vgl::MultiTriangulateLine                             src/vgl.cc:28-76  (JacobiSVD is NOT restated: the right singular vector of the
//                                                         smallest singular value comes from a Jacobi eigen-decomposition of M^T M, sign fixed
//                                                         so that its largest component is positive; the rest is literal)
//   SubselectWithGrid / GetHoughCoordinates               src/LineMatching.cc:63-180
//   GetReprojThrPyramid, GetLineEq, GetReprojErrPixelsL1  src/LineMatching.cc:239-275;  vgl::LineReprojErrorL1  src/vgl.cc:548-559
// Eigen is absent, so ColPivHouseholderQR (rank(), solve()) is restated: column pivoting on the largest remaining
// column norm, Householder reflections, rank = #{ |R_kk| > eps * size * max|R_kk| }, minimum-norm-free solve on the
// leading rank block (Eigen zeroes the remaining unknowns).
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../include/lld_amd.h"
#include "lldo_math.h"

using namespace lldo;

extern "C" double lldo_l2f32(const float* a, const float* b, int dim);

namespace {

struct M34 { M3 R; V3 t; };   // T.block<3,4>(0,0) of a 4x4 pose

// Eigen::ColPivHouseholderQR of an m x n matrix (m >= n, m <= 4, n <= 3): solves A x = b in the least-squares sense.
// Returns the rank; x gets zeros for the dropped unknowns.
int colpiv_qr_solve(int m, int n, double A[4][3], const double* b_in, double* x) {
  double b[4]; for (int i = 0; i < m; i++) b[i] = b_in[i];
  int perm[3] = {0, 1, 2};
  double maxpivot = 0.0; double diag[3] = {0, 0, 0};
  for (int k = 0; k < n; k++) {
    // pivot: largest squared norm of the remaining part of the remaining columns (first maximum wins)
    int best = k; double best_norm = -1.0;
    for (int c = k; c < n; c++) {
      double s = 0.0; for (int r = k; r < m; r++) s += A[r][c] * A[r][c];
      if (s > best_norm) { best_norm = s; best = c; }
    }
    if (best != k) { for (int r = 0; r < m; r++) std::swap(A[r][k], A[r][best]); std::swap(perm[k], perm[best]); }
    // Householder: makeHouseholderInPlace on A[k..m-1][k]
    const double c0 = A[k][k];
    double tail = 0.0; for (int r = k + 1; r < m; r++) tail += A[r][k] * A[r][k];
    double beta = c0, tau = 0.0; double v[4] = {0, 0, 0, 0};
    if (tail > DBL_MIN) {
      beta = std::sqrt(c0 * c0 + tail);
      if (c0 >= 0) beta = -beta;
      for (int r = k + 1; r < m; r++) v[r] = A[r][k] / (c0 - beta);
      tau = (beta - c0) / beta;
      for (int c = k + 1; c < n; c++) {
        double w = A[k][c]; for (int r = k + 1; r < m; r++) w += v[r] * A[r][c];
        A[k][c] -= tau * w; for (int r = k + 1; r < m; r++) A[r][c] -= tau * w * v[r];
      }
      double w = b[k]; for (int r = k + 1; r < m; r++) w += v[r] * b[r];
      b[k] -= tau * w; for (int r = k + 1; r < m; r++) b[r] -= tau * w * v[r];
    }
    A[k][k] = beta; for (int r = k + 1; r < m; r++) A[r][k] = 0.0;
    diag[k] = std::fabs(beta);
    if (diag[k] > maxpivot) maxpivot = diag[k];
  }
  const double threshold = DBL_EPSILON * (double)n;      // NumTraits::epsilon() * diagonalSize()
  int rank = 0;
  for (int k = 0; k < n; k++) if (diag[k] > threshold * maxpivot) rank++;
  double y[3] = {0, 0, 0};
  for (int k = rank - 1; k >= 0; k--) {
    double s = b[k]; for (int c = k + 1; c < rank; c++) s -= A[k][c] * y[c];
    y[k] = s / A[k][k];
  }
  for (int k = 0; k < n; k++) x[perm[k]] = (k < rank) ? y[k] : 0.0;
  return rank;
}

V3 kt_mul(const double* K, const V3& l) {   // K.transpose() * l
  return V3{K[0] * l.x + K[3] * l.y + K[6] * l.z, K[1] * l.x + K[4] * l.y + K[7] * l.z, K[2] * l.x + K[5] * l.y + K[8] * l.z};
}
V3 k_mul(const double* K, const V3& X) {    // K * X
  return V3{K[0] * X.x + K[1] * X.y + K[2] * X.z, K[3] * X.x + K[4] * X.y + K[5] * X.z, K[6] * X.x + K[7] * X.y + K[8] * X.z};
}

// vgl::NormalizedLineEquation
V3 normalized_line_eq(const float* kl, const double* K) {
  const V3 Xs{kl[0], kl[1], 1.0}, Xe{kl[2], kl[3], 1.0};
  const V3 lineImg = cross(Xs, Xe);
  V3 leq = kt_mul(K, lineImg);
  const double n = std::sqrt(leq.x * leq.x + leq.y * leq.y);
  return V3{leq.x / n, leq.y / n, leq.z / n};
}

double line_length(const float* kl) {
  const double dx = (double)kl[0] - (double)kl[2], dy = (double)kl[1] - (double)kl[3];
  return std::sqrt(dx * dx + dy * dy);
}

// vgl::TriangulateLine
bool triangulate_line(const M34& T1, const M34& T2, const V3& l1, const V3& l2, V3* X0, V3* line_dir) {
  const V3 normal_1 = m3_mulv(T1.R, l1), normal_2 = m3_mulv(T2.R, l2);
  if (std::fabs(dot(normal_1, normal_2)) / norm(normal_1) / norm(normal_2) > 0.975) return false;
  V3 d = cross(normal_1, normal_2);
  d = scale(d, 1.0 / norm(d));
  *line_dir = d;
  double M[4][3] = {{normal_1.x, normal_1.y, normal_1.z}, {normal_2.x, normal_2.y, normal_2.z}, {d.x, d.y, d.z}, {0, 0, 0}};
  const double b[3] = {dot(normal_1, T1.t), dot(normal_2, T2.t), 0.0};
  double x[3];
  if (colpiv_qr_solve(3, 3, M, b, x) < 3) return false;
  *X0 = V3{x[0], x[1], x[2]};
  return true;
}

// vgl::ReprojectLinePointTo3D
void reproject_line_point_to_3d(const V3& X0, const V3& line_dir, double px, double py, const double* K, double* depth, double* line_param) {
  const V3 kd = k_mul(K, line_dir);
  double M[4][3] = {{px, -kd.x, 0}, {py, -kd.y, 0}, {1.0, -kd.z, 0}, {0, 0, 0}};
  const V3 rhs = k_mul(K, X0);
  const double b[3] = {rhs.x, rhs.y, rhs.z};
  double sol[3];
  colpiv_qr_solve(3, 2, M, b, sol);
  *depth = sol[0]; *line_param = sol[1];
}

// ReprojectKeyLineTo3D with T = [R|t] camera-to-world as vgl::MapPoint reads it
void reproject_keyline_to_3d(const float* kl, const M34& T, const double* K, const V3& X0, const V3& lineDir, V3* X1, V3* X2) {
  const V3 dX = sub(X0, T.t);
  const V3 X0rot{T.R.m[0][0] * dX.x + T.R.m[1][0] * dX.y + T.R.m[2][0] * dX.z, T.R.m[0][1] * dX.x + T.R.m[1][1] * dX.y + T.R.m[2][1] * dX.z,
                 T.R.m[0][2] * dX.x + T.R.m[1][2] * dX.y + T.R.m[2][2] * dX.z};
  const V3 dirRot{T.R.m[0][0] * lineDir.x + T.R.m[1][0] * lineDir.y + T.R.m[2][0] * lineDir.z,
                  T.R.m[0][1] * lineDir.x + T.R.m[1][1] * lineDir.y + T.R.m[2][1] * lineDir.z,
                  T.R.m[0][2] * lineDir.x + T.R.m[1][2] * lineDir.y + T.R.m[2][2] * lineDir.z};
  double d, p;
  reproject_line_point_to_3d(X0rot, dirRot, kl[0], kl[1], K, &d, &p);
  *X1 = add(X0, scale(lineDir, p));
  reproject_line_point_to_3d(X0rot, dirRot, kl[2], kl[3], K, &d, &p);
  *X2 = add(X0, scale(lineDir, p));
}

// the geometric part of CheckLinePair (everything before MatchLineDescriptors)
bool pair_gate(const lld_line_stereo_params& P, const M34& T, const M34& T_right, const float* kl1, int oct1, const float* kl2, int oct2,
               V3* X0_out) {
  if (P.is_stereo && oct1 != oct2) return false;
  const double len_thr = P.min_line_length;
  if (line_length(kl1) < len_thr || line_length(kl2) < len_thr) return false;
  V3 X0, line_dir;
  const V3 leftEq = normalized_line_eq(kl1, P.K), rightEq = normalized_line_eq(kl2, P.K);
  if (!triangulate_line(T, T_right, leftEq, rightEq, &X0, &line_dir) || norm(X0) < 0.5) return false;
  V3 p1, p2;
  reproject_keyline_to_3d(kl1, T, P.K, X0, line_dir, &p1, &p2);
  if (p1.z < 0 || p2.z < 0) return false;
  if (X0_out) *X0_out = X0;
  return true;
}

void make_poses(const lld_line_stereo_params& P, M34* T, M34* T_right) {
  T->R = m3_identity(); T->t = V3{0, 0, 0};                       // T.setIdentity()
  *T_right = *T;                                                  // GetTForRight
  T_right->t = add(T_right->t, m3_mulv(T_right->R, V3{P.b, 0, 0}));
}

}  // namespace

extern "C" {

// test hook: vgl::TriangulateLine + the two re-projected endpoints for one pair; out = X0(3) dir(3) p1(3) p2(3); returns the gate
int lldo_line_pair_geometry(const lld_line_stereo_params* P, const float* kl1, const float* kl2, double* out) {
  M34 T, Tr; make_poses(*P, &T, &Tr);
  V3 X0{0, 0, 0}, dir{0, 0, 0}, p1{0, 0, 0}, p2{0, 0, 0};
  const V3 l1 = normalized_line_eq(kl1, P->K), l2 = normalized_line_eq(kl2, P->K);
  const bool ok = triangulate_line(T, Tr, l1, l2, &X0, &dir);
  if (ok) reproject_keyline_to_3d(kl1, T, P->K, X0, dir, &p1, &p2);
  const double v[12] = {X0.x, X0.y, X0.z, dir.x, dir.y, dir.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
  for (int i = 0; i < 12; i++) out[i] = v[i];
  return ok ? 1 : 0;
}

// test hook: the restated ColPivHouseholderQR
int lldo_colpiv_qr_solve(int m, int n, const double* A_rowmajor, const double* b, double* x) {
  if (m < 1 || m > 4 || n < 1 || n > 3 || m < n) return -1;
  double A[4][3] = {{0}};
  for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) A[i][j] = A_rowmajor[i * n + j];
  return colpiv_qr_solve(m, n, A, b, x);
}

// TwoFrameLineMatcher::MatchLines, whole routine
int lldo_line_match_stereo(void*, const lld_line_stereo_params* P, const float* left_lines, const int32_t* left_octave, const float* desc_left, int nq,
                           const float* right_lines, const int32_t* right_octave, const float* desc_right, int nt, int dim,
                           int32_t* matches, double* match_dist, uint8_t* gate_out) {
  M34 T, Tr; make_poses(*P, &T, &Tr);
  std::vector<bool> is_matched(nq, false), is_other_matched(nt, false);
  if (gate_out)
    for (int j = 0; j < nq; j++) for (int oi = 0; oi < nt; oi++)
      gate_out[(size_t)j * nt + oi] = pair_gate(*P, T, Tr, left_lines + 4 * j, left_octave[j], right_lines + 4 * oi, right_octave[oi], nullptr);
  for (int j = 0; j < nq; j++) {
    matches[j] = -1; if (match_dist) match_dist[j] = DBL_MAX;
    if (is_matched[j]) continue;
    double min_d = DBL_MAX, sec_min_d = DBL_MAX; int min_j = -1;
    for (int oi = 0; oi < nt; oi++) {
      if (is_other_matched[oi]) continue;
      // CheckLinePair
      if (!pair_gate(*P, T, Tr, left_lines + 4 * j, left_octave[j], right_lines + 4 * oi, right_octave[oi], nullptr)) continue;
      const double d = lldo_l2f32(desc_left + (size_t)dim * j, desc_right + (size_t)dim * oi, dim);
      double m = min_d; if (d < m) m = d;
      if (m < min_d && m < P->tau) { sec_min_d = min_d; min_d = m; min_j = oi; }
    }
    (void)sec_min_d;
    if (min_j >= 0) is_other_matched[min_j] = true;
    matches[j] = min_j;
    if (match_dist && min_j >= 0) match_dist[j] = min_d;
  }
  return LLD_OK;
}


// ---------------------------------------------------------------- Tracking::AddLinesFrom
namespace {
constexpr int kDistCells = 50, kAngCells = 50;          // FRAME_DIST_CELLS, FRAME_ANG_CELLS (include/Frame.h:45-46)
#define LLDO_PI 3.14159265                               /* the literal of src/LineMatching.cc:61 */

// GetHoughCoordinates (src/LineMatching.cc:63-152), literal; also returns the centre cell
void hough_coordinates(V3 leq, double sx, double sy, std::vector<int>* dist_inds, std::vector<int>* ang_inds, int step_dist, int step_ang,
                       int* dist_c, int* ang_c) {
  dist_inds->clear(); ang_inds->clear();
  leq.x /= sx; leq.y /= sy;
  { const double n = std::sqrt(leq.x * leq.x + leq.y * leq.y); leq.x /= n; leq.y /= n; leq.z /= n; }
  if (leq.y < 0) { leq.x = -leq.x; leq.y = -leq.y; leq.z = -leq.z; }
  if (!std::isfinite(leq.x) || !std::isfinite(leq.y) || !std::isfinite(leq.z)) {
    // a degenerate line (coincident end points): the reference would cast a NaN to int and index its grid with it; the build treats it as
    // the line y = 0 - same rule as the device code
    leq = V3{0.0, 1.0, 0.0};
  }
  const int dist_cell_num = kDistCells, ang_cell_num = kAngCells;
  const double dist_level = std::fabs(leq.z / (std::sqrt(2.0))) * dist_cell_num;
  int dist_ind = (int)std::floor(dist_level + 0.5);
  dist_ind = std::min(dist_ind, dist_cell_num - 1); dist_ind = std::max(dist_ind, 0);
  int shift_dist = -1;
  if (dist_level - dist_ind < 0) shift_dist = 1;
  const double ang = std::atan2(leq.y, leq.x);
  const double ang_level = ang / LLDO_PI * ang_cell_num;
  int ang_ind = (int)std::floor(ang_level + 0.5);
  ang_ind = std::min(ang_ind, ang_cell_num - 1); ang_ind = std::max(ang_ind, 0);
  int shift_ang = -1;
  if (ang_level - ang_ind < 0) shift_ang = 1;
  if (dist_c) *dist_c = dist_ind;
  if (ang_c) *ang_c = ang_ind;
  const int ang_max = std::max(ang_ind, ang_ind + shift_ang);
  for (int i = ang_max; i < ang_max + step_ang; i++) { int a = i; if (i < 0) a += ang_cell_num; a = a % ang_cell_num; ang_inds->push_back(a); }
  const int ang_min = std::min(ang_ind, ang_ind + shift_ang);
  for (int i = ang_min; i > ang_min - step_ang; i--) { int a = i; if (i < 0) a += ang_cell_num; a = a % ang_cell_num; ang_inds->push_back(a); }
  const int dist_max = std::max(dist_ind, dist_ind + shift_dist);
  for (int i = dist_max; i < dist_max + step_dist; i++) if (i >= 0 && i < dist_cell_num - 1) dist_inds->push_back(i);
  const int dist_min = std::min(dist_ind, dist_ind + shift_dist);
  for (int i = dist_min; i > dist_min - step_dist; i--) if (i >= 0 && i < dist_cell_num - 1) dist_inds->push_back(i);
}
V3 line_eq_px(const float* kl) { return cross(V3{kl[0], kl[1], 1.0}, V3{kl[2], kl[3], 1.0}); }   // GetLineEq
struct M44 { M3 R; V3 t; };
M44 pose44(const double* T) { M44 P; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) P.R.m[r][c] = T[4 * r + c]; P.t = V3{T[3], T[7], T[11]}; return P; }
V3 m3t_mulv(const M3& a, const V3& v) {                    // R^T v
  return V3{a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z, a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z, a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z};
}
V3 map_point(const M44& T, const V3& X) { return m3t_mulv(T.R, sub(X, T.t)); }                    // vgl::MapPoint: R^T (X - t)
// vgl::LineReprojErrorL1
double line_reproj_err_l1(double xs, double ys, double xe, double ye, const M44& T, const V3& X0, const V3& dir, const double* K) {
  const V3 Xc1 = k_mul(K, map_point(T, X0)), Xc2 = k_mul(K, map_point(T, add(X0, dir)));
  V3 l = cross(Xc1, Xc2);
  const double n = std::sqrt(l.x * l.x + l.y * l.y);
  l.x /= n; l.y /= n; l.z /= n;
  return std::fabs(xs * l.x + ys * l.y + l.z) + std::fabs(xe * l.x + ye * l.y + l.z);
}
}  // namespace

int lld_line_hough_cells_oracle(const float* lines, int n, double sx, double sy, int32_t* cell) {
  std::vector<int> d, a;
  for (int i = 0; i < n; i++) { int dc, ac; hough_coordinates(line_eq_px(lines + 4 * i), sx, sy, &d, &a, 0, 0, &dc, &ac); cell[i] = dc * kAngCells + ac; }
  return LLD_OK;
}
int lldo_line_hough_cells(const float* lines, int n, double sx, double sy, int32_t* cell) { return lld_line_hough_cells_oracle(lines, n, sx, sy, cell); }
// the two index lists of GetHoughCoordinates for a homogeneous image line (known-answer tests)
int lldo_hough_coordinates(const double* leq, double sx, double sy, int step_dist, int step_ang, int32_t* dist_inds, int32_t* n_dist, int32_t* ang_inds, int32_t* n_ang) {
  std::vector<int> d, a;
  hough_coordinates(V3{leq[0], leq[1], leq[2]}, sx, sy, &d, &a, step_dist, step_ang, nullptr, nullptr);
  *n_dist = (int)d.size(); *n_ang = (int)a.size();
  for (size_t i = 0; i < d.size(); i++) dist_inds[i] = d[i];
  for (size_t i = 0; i < a.size(); i++) ang_inds[i] = a[i];
  return LLD_OK;
}

int lldo_line_track_match(void*, const lld_line_track_params* P, int n_map, const double* map_x0, const double* map_dir, const double* map_x1,
                          const double* map_x2, const uint8_t* map_skip, const float* map_desc, int n_cur, const float* left_lines,
                          const int32_t* left_octave, int n_right, const float* right_lines, const int32_t* line_matches, const uint8_t* occupied0,
                          const float* cur_desc, int dim, int32_t* matches, double* match_dist, uint8_t* gate_out) {
  const M44 T_curr = pose44(P->T_curr);
  M44 T_right = T_curr;                                   // GetTForRight
  T_right.t = add(T_curr.t, m3_mulv(T_curr.R, V3{P->b, 0.0, 0.0}));
  // the line grid of the frame (the fill the reference lacks: every line in its centre cell)
  std::vector<std::vector<std::vector<int>>> lines_grid(kDistCells, std::vector<std::vector<int>>(kAngCells));
  {
    std::vector<int32_t> cell(n_cur);
    lld_line_hough_cells_oracle(left_lines, n_cur, P->sx, P->sy, cell.data());
    for (int si = 0; si < n_cur; si++) lines_grid[cell[si] / kAngCells][cell[si] % kAngCells].push_back(si);
  }
  std::vector<char> occupied(n_cur, 0);
  if (occupied0) for (int si = 0; si < n_cur; si++) occupied[si] = occupied0[si] != 0;
  if (gate_out) std::fill(gate_out, gate_out + (size_t)n_map * n_cur, (uint8_t)0);
  for (int i = 0; i < n_map; i++) {
    matches[i] = -1; if (match_dist) match_dist[i] = DBL_MAX;
    if (map_skip && map_skip[i]) continue;
    const V3 X0m{map_x0[3 * i], map_x0[3 * i + 1], map_x0[3 * i + 2]}, dirm{map_dir[3 * i], map_dir[3 * i + 1], map_dir[3 * i + 2]};
    std::vector<int> sub_inds;
    if (P->use_grid) {
      // SubselectWithGrid
      const V3 Xl1 = k_mul(P->K, m3t_mulv(T_curr.R, sub(X0m, T_curr.t))), Xl2 = k_mul(P->K, m3t_mulv(T_curr.R, sub(add(X0m, dirm), T_curr.t)));
      V3 leq = cross(Xl1, Xl2);
      { const double n = std::sqrt(leq.x * leq.x + leq.y * leq.y); leq.x /= n; leq.y /= n; leq.z /= n; }
      std::vector<int> dis, ais;
      hough_coordinates(leq, P->sx, P->sy, &dis, &ais, 3, 3, nullptr, nullptr);
      std::vector<char> in(n_cur, 0);                    // std::set<int>: unique, ascending
      for (int ai : ais) for (int di : dis) for (int oi : lines_grid[di][ai]) in[oi] = 1;
      for (int oi = 0; oi < n_cur; oi++) if (in[oi]) sub_inds.push_back(oi);
    } else for (int oi = 0; oi < n_cur; oi++) sub_inds.push_back(oi);
    int match_id = -1; double md = 1e10;
    const float* ml_desc = map_desc + (size_t)dim * i;
    for (int si : sub_inds) {
      if (occupied[si]) continue;
      const int ri = line_matches[si];
      if (ri < 0 && !P->monocular) continue;
      const V3 X1{map_x1[3 * i], map_x1[3 * i + 1], map_x1[3 * i + 2]}, X2{map_x2[3 * i], map_x2[3 * i + 1], map_x2[3 * i + 2]};
      const V3 X1c = map_point(T_curr, X1), X2c = map_point(T_curr, X2);
      if (X1c.z < 0 || X2c.z < 0) continue;
      double thr = P->thr_reproj_base;                  // GetReprojThrPyramid
      for (int oi = 0; oi < left_octave[si]; oi++) thr *= 1.44;
      const float* kl1 = left_lines + 4 * si;
      const double se = line_reproj_err_l1(kl1[0], kl1[1], kl1[2], kl1[3], T_curr, X0m, dirm, P->K);
      double se2 = 0;
      if (!P->monocular) { const float* kr = right_lines + 4 * ri; se2 = line_reproj_err_l1(kr[0], kr[1], kr[2], kr[3], T_right, X0m, dirm, P->K); }
      if (se > thr || se2 > thr) continue;
      if (gate_out) gate_out[(size_t)i * n_cur + si] = 1;
      const double cd = lldo_l2f32(ml_desc, cur_desc + (size_t)dim * si, dim);
      if (cd < md) { md = cd; match_id = si; }
    }
    if (md > P->md_thr) continue;
    if (match_id >= 0) {
      if (occupied[match_id]) continue;
      occupied[match_id] = 1;
      matches[i] = match_id; if (match_dist) match_dist[i] = md;
    }
  }
  (void)n_right;
  return LLD_OK;
}

// ---------------------------------------------------------------- Tracking::MatchLinesLastKF
namespace {
// eigenvector of the smallest eigenvalue of the symmetric 3x3 A (cyclic Jacobi in long double), largest component positive
V3 smallest_eigvec(const double A_in[3][3]) {
  long double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = A_in[i][j];
  for (int sweep = 0; sweep < 60; sweep++) {
    long double off = fabsl(A[0][1]) + fabsl(A[0][2]) + fabsl(A[1][2]);
    if (off == 0.0L) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (A[p][q] == 0.0L) continue;
      const long double theta = (A[q][q] - A[p][p]) / (2.0L * A[p][q]);
      const long double t = (theta >= 0 ? 1.0L : -1.0L) / (fabsl(theta) + sqrtl(theta * theta + 1.0L));
      const long double c = 1.0L / sqrtl(t * t + 1.0L), sn = t * c;
      for (int k = 0; k < 3; k++) { const long double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
      for (int k = 0; k < 3; k++) { const long double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
      for (int k = 0; k < 3; k++) { const long double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
    }
  }
  int m = 0; for (int i = 1; i < 3; i++) if (A[i][i] < A[m][m]) m = i;
  V3 v{(double)V[0][m], (double)V[1][m], (double)V[2][m]};
  const double n = norm(v); v = scale(v, 1.0 / n);
  int big = 0; if (std::fabs(v.y) > std::fabs(at(v, big))) big = 1; if (std::fabs(v.z) > std::fabs(at(v, big))) big = 2;
  if (at(v, big) < 0) v = scale(v, -1.0);
  return v;
}
// vgl::MultiTriangulateLine (src/vgl.cc:28-76)
bool multi_triangulate_line(int n, const M44* Ts, const V3* lines, V3* X0_p, V3* line_dir_p) {
  if (n < 3) return false;
  if (n > 4) return false;                                          // (the reference takes any count; its only call site passes four views)
  V3 normals[4];
  for (int i = 0; i < n; i++) { const V3 leq = scale(lines[i], 1.0 / norm(lines[i])); normals[i] = m3_mulv(Ts[i].R, leq); }
  for (int i = 1; i < n; i++) if (std::fabs(dot(normals[0], normals[i])) / norm(normals[0]) / norm(normals[i]) > 0.975) return false;
  double MtM[3][3] = {{0}};
  for (int i = 0; i < n; i++) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) MtM[r][c] += at(normals[i], r) * at(normals[i], c);
  const V3 line_dir = smallest_eigvec(MtM);                         // Vm.col(2) of M2.jacobiSvd(...)
  double M1[4][3], b1[4], x[3];
  for (int i = 0; i < n; i++) { M1[i][0] = normals[i].x; M1[i][1] = normals[i].y; M1[i][2] = normals[i].z; b1[i] = dot(normals[i], Ts[i].t); }
  colpiv_qr_solve(n, 3, M1, b1, x);                                 // M1.colPivHouseholderQr().solve(b1)
  V3 X0{x[0], x[1], x[2]};
  X0 = sub(X0, scale(line_dir, dot(X0, line_dir)));
  *line_dir_p = line_dir; *X0_p = X0;
  return true;
}
M34 m34_of(const M44& T) { M34 r; r.R = T.R; r.t = T.t; return r; }
}  // namespace

int lldo_multi_triangulate_line(int n, const double* Ts /*[n][16]*/, const double* lines /*[n][3]*/, double* x0, double* dir) {
  M44 T[4]; V3 l[4];
  if (n > 4) return 0;
  for (int i = 0; i < n; i++) { T[i] = pose44(Ts + 16 * i); l[i] = V3{lines[3 * i], lines[3 * i + 1], lines[3 * i + 2]}; }
  V3 X0, d;
  if (!multi_triangulate_line(n, T, l, &X0, &d)) return 0;
  x0[0] = X0.x; x0[1] = X0.y; x0[2] = X0.z; dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
  return 1;
}

int lldo_line_match_last_frame(void*, const lld_line_lastkf_params* P, int n_cur, const float* cur_left, int n_cur_right, const float* cur_right,
                               const int32_t* cur_line_matches, const uint8_t* cur_occupied, const float* cur_desc, int n_last, const float* last_left,
                               const int32_t* last_left_octave, int n_last_right, const float* last_right, const int32_t* last_line_matches,
                               const uint8_t* last_skip, const float* last_desc, int dim, int32_t* match_last, uint8_t* created, double* x0_out, double* dir_out) {
  const M44 T = pose44(P->T_curr), T_last = pose44(P->T_last);
  M44 T_right = T, T_last_right = T_last;                                 // GetTForRight
  T_right.t = add(T.t, m3_mulv(T.R, V3{P->b, 0.0, 0.0}));
  T_last_right.t = add(T_last.t, m3_mulv(T_last.R, V3{P->b, 0.0, 0.0}));
  std::vector<std::vector<std::vector<int>>> grid(kDistCells, std::vector<std::vector<int>>(kAngCells));
  {
    std::vector<int32_t> cell(n_last);
    lld_line_hough_cells_oracle(last_left, n_last, P->sx, P->sy, cell.data());
    for (int li = 0; li < n_last; li++) grid[cell[li] / kAngCells][cell[li] % kAngCells].push_back(li);
  }
  for (int i = 0; i < n_cur; i++) {
    match_last[i] = -1; created[i] = 0;
    for (int k = 0; k < 3; k++) { x0_out[3 * i + k] = 0.0; dir_out[3 * i + k] = 0.0; }
    if (cur_occupied && cur_occupied[i]) continue;
    const int ri = cur_line_matches[i];
    if (ri < 0) continue;                                               // (the stereo system; the reference would index mvLinesRight[-1] otherwise)
    const V3 l1 = normalized_line_eq(cur_left + 4 * i, P->K), l2 = normalized_line_eq(cur_right + 4 * ri, P->K);
    V3 X0, line_dir;
    if (!triangulate_line(m34_of(T), m34_of(T_right), l1, l2, &X0, &line_dir)) continue;
    std::vector<int> lines_inds;
    if (P->use_grid) {
      const V3 Xl1 = k_mul(P->K, m3t_mulv(T_last.R, sub(X0, T_last.t))), Xl2 = k_mul(P->K, m3t_mulv(T_last.R, sub(add(X0, line_dir), T_last.t)));
      V3 leq = cross(Xl1, Xl2);
      { const double n = std::sqrt(leq.x * leq.x + leq.y * leq.y); leq.x /= n; leq.y /= n; leq.z /= n; }
      std::vector<int> dis, ais;
      hough_coordinates(leq, P->sx, P->sy, &dis, &ais, 3, 3, nullptr, nullptr);
      std::vector<char> in(n_last, 0);
      for (int ai : ais) for (int di : dis) for (int oi : grid[di][ai]) in[oi] = 1;
      for (int oi = 0; oi < n_last; oi++) if (in[oi]) lines_inds.push_back(oi);
    } else for (int oi = 0; oi < n_last; oi++) lines_inds.push_back(oi);
    int match_id = -1; double md = 1e10;
    for (int li : lines_inds) {
      const int pri = last_line_matches[li];
      if (pri < 0) continue;
      if (last_skip && last_skip[li]) continue;
      double thr = P->thr_reproj_base;
      for (int oi = 0; oi < last_left_octave[li]; oi++) thr *= 1.44;
      const float* kl = last_left + 4 * li; const float* kr = last_right + 4 * pri;
      const double se = line_reproj_err_l1(kl[0], kl[1], kl[2], kl[3], T_last, X0, line_dir, P->K);
      const double se2 = line_reproj_err_l1(kr[0], kr[1], kr[2], kr[3], T_last_right, X0, line_dir, P->K);
      if (se > thr && se2 > thr) continue;
      const double cd = lldo_l2f32(last_desc + (size_t)dim * li, cur_desc + (size_t)dim * i, dim);
      if (cd < md) { md = cd; match_id = li; }
    }
    if (match_id < 0) continue;
    const int pi = match_id;
    if (md > P->md_thr) continue;
    match_last[i] = pi;
    const M44 Ts[4] = {T, T_right, T_last, T_last_right};
    const int pri = last_line_matches[pi];
    const V3 leqs[4] = {l1, l2, normalized_line_eq(last_left + 4 * pi, P->K), normalized_line_eq(last_right + 4 * pri, P->K)};
    if (!multi_triangulate_line(4, Ts, leqs, &X0, &line_dir)) continue;
    V3 p1, p2;
    reproject_keyline_to_3d(cur_left + 4 * i, m34_of(T), P->K, X0, line_dir, &p1, &p2);
    bool is_behind = false;
    for (const M44& Ti : Ts) { const V3 p1c = map_point(Ti, p1), p2c = map_point(Ti, p2); if (p1c.z < 0 || p2c.z < 0) is_behind = true; }
    if (is_behind) continue;
    created[i] = 1;
    x0_out[3 * i] = X0.x; x0_out[3 * i + 1] = X0.y; x0_out[3 * i + 2] = X0.z;
    dir_out[3 * i] = line_dir.x; dir_out[3 * i + 1] = line_dir.y; dir_out[3 * i + 2] = line_dir.z;
  }
  (void)n_cur_right; (void)n_last_right;
  return LLD_OK;
}

}  // extern "C"
