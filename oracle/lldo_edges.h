// ORACLE — TEST INFRASTRUCTURE ONLY (see lldo_math.h header).  PARITY UNPINNED.
//
// lldo_edges.h — residuals, Jacobians, depth tests and the Huber kernel, restated from
//   Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h,cpp}
//   Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp
//   src/vgl.cc:336-346
#ifndef LLDO_EDGES_H
#define LLDO_EDGES_H

#include "lldo_math.h"

namespace lldo {

struct Cam { double fx, fy, cx, cy, bf; };

// ---------------------------------------------------------------- point edges: residuals
// EdgeSE3ProjectXYZ::cam_project (types_six_dof_expmap.cpp:143-149) and the OnlyPose twin (:296-302)
static inline void mono_error(const Cam& c, const V3& Xc, const double* obs, double* e) {
  const double px = Xc.x / Xc.z, py = Xc.y / Xc.z;     // project2d
  e[0] = obs[0] - (px * c.fx + c.cx);
  e[1] = obs[1] - (py * c.fy + c.cy);
}
// EdgeStereoSE3ProjectXYZ::cam_project(trans_xyz, const float& bf) (types_six_dof_expmap.cpp:152-159):
// invz and bf are FLOAT; bf*invz is a float product.
static inline void stereo_error_binary(const Cam& c, const V3& Xc, const double* obs, double* e) {
  const float invz = 1.0f / Xc.z;            // double division, rounded to float
  const float bff = (float)c.bf;             // 'const float &bf' bound to the double member
  const double r0 = Xc.x * invz * c.fx + c.cx;
  const double r1 = Xc.y * invz * c.fy + c.cy;
  const float bfz = bff * invz;              // float * float
  const double r2 = r0 - bfz;
  e[0] = obs[0] - r0; e[1] = obs[1] - r1; e[2] = obs[2] - r2;
}
// EdgeStereoSE3ProjectXYZOnlyPose::cam_project (types_six_dof_expmap.cpp:305-312): invz float, bf DOUBLE
static inline void stereo_error_posonly(const Cam& c, const V3& Xc, const double* obs, double* e) {
  const float invz = 1.0f / Xc.z;
  const double r0 = Xc.x * invz * c.fx + c.cx;
  const double r1 = Xc.y * invz * c.fy + c.cy;
  const double r2 = r0 - c.bf * invz;
  e[0] = obs[0] - r0; e[1] = obs[1] - r1; e[2] = obs[2] - r2;
}

// ---------------------------------------------------------------- point edges: Jacobians
// Binary edges (EdgeSE3ProjectXYZ::linearizeOplus :111-141, EdgeStereoSE3ProjectXYZ::linearizeOplus :188-236).
// Jp: rows x 3 (point), Jc: rows x 6 (pose, rotation columns first).  rows = 2 (mono) or 3 (stereo).
static inline void point_jac_binary(const Cam& c, const SE3& T, const V3& Xw, bool stereo, double* Jp, double* Jc) {
  const V3 Xc = se3_map(T, Xw);
  const M3 R = quat_to_R(T.q);
  const double x = Xc.x, y = Xc.y, z = Xc.z, z_2 = z * z;
  const double fx = c.fx, fy = c.fy, bf = c.bf;
  if (!stereo) {
    // _jacobianOplusXi = -1./z * tmp * R  with tmp = [[fx,0,-x/z*fx],[0,fy,-y/z*fy]]
    double tmp[2][3] = {{fx, 0, -x / z * fx}, {0, fy, -y / z * fy}};
    const double s = -1. / z;
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) {
        // Eigen evaluates (scalar * tmp) * R
        double acc = (s * tmp[i][0]) * R.m[0][j] + (s * tmp[i][1]) * R.m[1][j] + (s * tmp[i][2]) * R.m[2][j];
        Jp[i * 3 + j] = acc;
      }
  } else {
    for (int j = 0; j < 3; j++) {
      Jp[0 * 3 + j] = -fx * R.m[0][j] / z + fx * x * R.m[2][j] / z_2;
      Jp[1 * 3 + j] = -fy * R.m[1][j] / z + fy * y * R.m[2][j] / z_2;
      Jp[2 * 3 + j] = Jp[0 * 3 + j] - bf * R.m[2][j] / z_2;
    }
  }
  Jc[0] = x * y / z_2 * fx;
  Jc[1] = -(1 + (x * x / z_2)) * fx;
  Jc[2] = y / z * fx;
  Jc[3] = -1. / z * fx;
  Jc[4] = 0;
  Jc[5] = x / z_2 * fx;
  Jc[6 + 0] = (1 + y * y / z_2) * fy;
  Jc[6 + 1] = -x * y / z_2 * fy;
  Jc[6 + 2] = -x / z * fy;
  Jc[6 + 3] = 0;
  Jc[6 + 4] = -1. / z * fy;
  Jc[6 + 5] = y / z_2 * fy;
  if (stereo) {
    Jc[12 + 0] = Jc[0] - bf * y / z_2;
    Jc[12 + 1] = Jc[1] + bf * x / z_2;
    Jc[12 + 2] = Jc[2];
    Jc[12 + 3] = Jc[3];
    Jc[12 + 4] = 0;
    Jc[12 + 5] = Jc[5] - bf / z_2;
  }
}
// Pose-only edges (EdgeSE3ProjectXYZOnlyPose::linearizeOplus :272-293, Stereo twin :337-366): invz in double
static inline void point_jac_posonly(const Cam& c, const V3& Xc, bool stereo, double* Jc) {
  const double x = Xc.x, y = Xc.y;
  const double invz = 1.0 / Xc.z, invz_2 = invz * invz;
  const double fx = c.fx, fy = c.fy, bf = c.bf;
  Jc[0] = x * y * invz_2 * fx;
  Jc[1] = -(1 + (x * x * invz_2)) * fx;
  Jc[2] = y * invz * fx;
  Jc[3] = -invz * fx;
  Jc[4] = 0;
  Jc[5] = x * invz_2 * fx;
  Jc[6 + 0] = (1 + y * y * invz_2) * fy;
  Jc[6 + 1] = -x * y * invz_2 * fy;
  Jc[6 + 2] = -x * invz * fy;
  Jc[6 + 3] = 0;
  Jc[6 + 4] = -invz * fy;
  Jc[6 + 5] = y * invz_2 * fy;
  if (stereo) {
    Jc[12 + 0] = Jc[0] - bf * y * invz_2;
    Jc[12 + 1] = Jc[1] + bf * x * invz_2;
    Jc[12 + 2] = Jc[2];
    Jc[12 + 3] = Jc[3];
    Jc[12 + 4] = 0;
    Jc[12 + 5] = Jc[5] - bf * invz_2;
  }
}

// ---------------------------------------------------------------- line edges
// K = [[f,0,cx],[0,f,cy],[0,0,1]] with the single focal f = K(0,0) (types_six_dof_expmap.h:358-363)
static inline V3 Kmul(double f, double cx, double cy, const V3& X) {
  return V3{f * X.x + 0.0 * X.y + cx * X.z, 0.0 * X.x + f * X.y + cy * X.z, 0.0 * X.x + 0.0 * X.y + 1.0 * X.z};
}
static inline M3 Kmat(double f, double cx, double cy) {
  M3 K = m3_identity();
  K.m[0][0] = f; K.m[1][1] = f; K.m[0][2] = cx; K.m[1][2] = cy;
  return K;
}

// EdgeSE3ProjectLine::computeError (types_six_dof_expmap.h:344-375) / EdgeSE3ProjectLineOnlyPose (:403-418).
// X1m/X2m: endpoints already mapped into the camera frame (without b).  x1/x2: detected endpoints (xs,ys),(xe,ye).
static inline void line_error_cam(double f, double cx, double cy, double bx, const V3& X1c, const V3& X2c,
                                  const double* x1, const double* x2, double* e) {
  const V3 b{bx, 0, 0};
  const V3 P1 = Kmul(f, cx, cy, add(X1c, b));
  const V3 P2 = Kmul(f, cx, cy, add(X2c, b));
  const V3 lt = cross(P1, P2);
  const double n = std::sqrt(lt.x * lt.x + lt.y * lt.y);
  const V3 l{lt.x / n, lt.y / n, lt.z / n};
  e[0] = x1[0] * l.x + x1[1] * l.y + 1.0 * l.z;
  e[1] = x2[0] * l.x + x2[1] * l.y + 1.0 * l.z;
}

struct M36 { double m[3][6]; };
struct M34 { double m[3][4]; };

// FormJacobianLineWRTCam (types_six_dof_expmap.cpp:472-497): J_l (3x6) = D * (cp(P1)*A2 - cp(P2)*A1), D (3x3)
static inline void line_jac_wrt_cam(double f, double cx, double cy, double bx, const V3& X1m, const V3& X2m, M36* J_lp, M3* Dp) {
  const V3 b{bx, 0, 0};
  const M3 K = Kmat(f, cx, cy);
  const V3 P1 = Kmul(f, cx, cy, add(X1m, b));
  const V3 P2 = Kmul(f, cx, cy, add(X2m, b));
  const V3 lt = cross(P1, P2);
  const double n = std::sqrt(lt.x * lt.x + lt.y * lt.y);
  const double dn[3] = {-lt.x / (n * n * n), -lt.y / (n * n * n), 0};
  M3 D;
  const double ltv[3] = {lt.x, lt.y, lt.z};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) D.m[i][j] = ltv[i] * dn[j] + (1.0 / n) * (i == j ? 1.0 : 0.0);
  *Dp = D;
  // J_l_k = [ -K*cpmat(Xkm) | K*I ]
  M36 A1, A2;
  {
    const M3 KS1 = m3_mul(K, skew(X1m)), KS2 = m3_mul(K, skew(X2m));
    const M3 KI = m3_mul(K, m3_identity());
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        A1.m[i][j] = -KS1.m[i][j]; A1.m[i][j + 3] = KI.m[i][j];
        A2.m[i][j] = -KS2.m[i][j]; A2.m[i][j + 3] = KI.m[i][j];
      }
  }
  const M3 C1 = skew(P1), C2 = skew(P2);
  M36 J;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 6; j++) {
      const double a = C1.m[i][0] * A2.m[0][j] + C1.m[i][1] * A2.m[1][j] + C1.m[i][2] * A2.m[2][j];
      const double c = C2.m[i][0] * A1.m[0][j] + C2.m[i][1] * A1.m[1][j] + C2.m[i][2] * A1.m[2][j];
      J.m[i][j] = a - c;
    }
  M36 DJ;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 6; j++) DJ.m[i][j] = D.m[i][0] * J.m[0][j] + D.m[i][1] * J.m[1][j] + D.m[i][2] * J.m[2][j];
  *J_lp = DJ;
}

// EdgeSE3ProjectLine::linearize (types_six_dof_expmap.cpp:505-581): Jl 2x4 (line), Jc 2x6 (pose)
static inline void line_jac_binary(double f, double cx, double cy, double bx, const SE3& T, const Line& L,
                                   const double* x1, const double* x2, double* Jl, double* Jc) {
  const M3 R = line_getR(L);
  const double alpha = L.alpha;
  const V3 c0 = m3_col(R, 0), c1 = m3_col(R, 1);
  const V3 X1 = scale(c1, alpha);
  const V3 X2 = add(X1, c0);
  const V3 X1m = se3_map(T, X1), X2m = se3_map(T, X2);
  M36 J_l; M3 D;
  line_jac_wrt_cam(f, cx, cy, bx, X1m, X2m, &J_l, &D);
  for (int j = 0; j < 6; j++) {
    Jc[j] = x1[0] * J_l.m[0][j] + x1[1] * J_l.m[1][j] + 1.0 * J_l.m[2][j];
    Jc[6 + j] = x2[0] * J_l.m[0][j] + x2[1] * J_l.m[1][j] + 1.0 * J_l.m[2][j];
  }
  // dX1 = [ 2 * (-cpmat(alpha*R.col(1))) | R.col(1) ],  dX2 = dX1 with rotation block - 2*cpmat(R.col(0))
  const M3 S1 = skew(scale(c1, alpha));
  const M3 S0 = skew(c0);
  M34 dX1, dX2;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      dX1.m[i][j] = 2 * (-S1.m[i][j]);
      dX2.m[i][j] = dX1.m[i][j] - 2 * S0.m[i][j];
    }
    dX1.m[i][3] = at(c1, i);
    dX2.m[i][3] = at(c1, i);
  }
  const M3 K = Kmat(f, cx, cy);
  const M3 Rc = quat_to_R(T.q);     // to_homogeneous_matrix().block<3,3>(0,0)
  const V3 b{bx, 0, 0};
  const V3 P1 = Kmul(f, cx, cy, add(X1m, b));
  const V3 P2 = Kmul(f, cx, cy, add(X2m, b));
  // d_l_tilde = cpmat(P1)*K*R_cam*dX2 - cpmat(P2)*K*R_cam*dX1  (evaluated left to right)
  const M3 M1 = m3_mul(m3_mul(skew(P1), K), Rc);
  const M3 M2 = m3_mul(m3_mul(skew(P2), K), Rc);
  M34 dlt;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) {
      const double a = M1.m[i][0] * dX2.m[0][j] + M1.m[i][1] * dX2.m[1][j] + M1.m[i][2] * dX2.m[2][j];
      const double c = M2.m[i][0] * dX1.m[0][j] + M2.m[i][1] * dX1.m[1][j] + M2.m[i][2] * dX1.m[2][j];
      dlt.m[i][j] = a - c;
    }
  M34 Dl;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) Dl.m[i][j] = D.m[i][0] * dlt.m[0][j] + D.m[i][1] * dlt.m[1][j] + D.m[i][2] * dlt.m[2][j];
  for (int j = 0; j < 4; j++) {
    Jl[j] = x1[0] * Dl.m[0][j] + x1[1] * Dl.m[1][j] + 1.0 * Dl.m[2][j];
    Jl[4 + j] = x2[0] * Dl.m[0][j] + x2[1] * Dl.m[1][j] + 1.0 * Dl.m[2][j];
  }
}

// EdgeSE3ProjectLineOnlyPose::linearize (types_six_dof_expmap.cpp:583-613): fixed world endpoints
static inline void line_jac_posonly(double f, double cx, double cy, double bx, const SE3& T, const V3& X1w, const V3& X2w,
                                    const double* x1, const double* x2, double* Jc) {
  const V3 X1m = se3_map(T, X1w), X2m = se3_map(T, X2w);
  M36 J_l; M3 D;
  line_jac_wrt_cam(f, cx, cy, bx, X1m, X2m, &J_l, &D);
  for (int j = 0; j < 6; j++) {
    Jc[j] = x1[0] * J_l.m[0][j] + x1[1] * J_l.m[1][j] + 1.0 * J_l.m[2][j];
    Jc[6 + j] = x2[0] * J_l.m[0][j] + x2[1] * J_l.m[1][j] + 1.0 * J_l.m[2][j];
  }
}

// vgl::ReprojectLinePointTo3D (src/vgl.cc:336-346): least-squares solve of the 3x2 system
//   [ (px,py,1) | -K*line_dir ] (depth, param)^T = K*X0   via Eigen colPivHouseholderQr.
// Restated as Householder QR with column pivoting (largest column norm first).  The matrix is
// assumed to have full column rank (Eigen would zero the second unknown otherwise).
static inline void reproject_line_point(const V3& X0, const V3& ldir, double px, double py, double f, double cx, double cy,
                                        double* depth, double* param) {
  double A[3][2];
  const V3 kd = Kmul(f, cx, cy, ldir);
  A[0][0] = px; A[1][0] = py; A[2][0] = 1.0;
  A[0][1] = -kd.x; A[1][1] = -kd.y; A[2][1] = -kd.z;
  const V3 rhsv = Kmul(f, cx, cy, X0);
  double rhs[3] = {rhsv.x, rhsv.y, rhsv.z};
  int perm[2] = {0, 1};
  const double n0 = A[0][0] * A[0][0] + A[1][0] * A[1][0] + A[2][0] * A[2][0];
  const double n1 = A[0][1] * A[0][1] + A[1][1] * A[1][1] + A[2][1] * A[2][1];
  if (n1 > n0) {
    perm[0] = 1; perm[1] = 0;
    for (int i = 0; i < 3; i++) { double t = A[i][0]; A[i][0] = A[i][1]; A[i][1] = t; }
  }
  // Householder on column 0 (rows 0..2)
  {
    const double c0 = A[0][0];
    const double tail = A[1][0] * A[1][0] + A[2][0] * A[2][0];
    if (tail != 0.0) {
      double beta = std::sqrt(c0 * c0 + tail);
      if (c0 >= 0) beta = -beta;
      const double v1 = A[1][0] / (c0 - beta), v2 = A[2][0] / (c0 - beta);
      const double tau = (beta - c0) / beta;
      // apply H = I - tau * v v^T (v = (1, v1, v2)) to column 1 and rhs
      double w = A[0][1] + v1 * A[1][1] + v2 * A[2][1];
      A[0][1] -= tau * w; A[1][1] -= tau * w * v1; A[2][1] -= tau * w * v2;
      w = rhs[0] + v1 * rhs[1] + v2 * rhs[2];
      rhs[0] -= tau * w; rhs[1] -= tau * w * v1; rhs[2] -= tau * w * v2;
      A[0][0] = beta; A[1][0] = 0; A[2][0] = 0;
    }
  }
  // Householder on column 1 (rows 1..2)
  {
    const double c0 = A[1][1];
    const double tail = A[2][1] * A[2][1];
    if (tail != 0.0) {
      double beta = std::sqrt(c0 * c0 + tail);
      if (c0 >= 0) beta = -beta;
      const double v1 = A[2][1] / (c0 - beta);
      const double tau = (beta - c0) / beta;
      const double w = rhs[1] + v1 * rhs[2];
      rhs[1] -= tau * w; rhs[2] -= tau * w * v1;
      A[1][1] = beta; A[2][1] = 0;
    }
  }
  double y[2];
  y[1] = rhs[1] / A[1][1];
  y[0] = (rhs[0] - A[0][1] * y[1]) / A[0][0];
  double sol[2];
  sol[perm[0]] = y[0];
  sol[perm[1]] = y[1];
  *depth = sol[0];
  *param = sol[1];
}

// EdgeSE3ProjectLine::IsDepthPositive (types_six_dof_expmap.h:312-342)
static inline bool line_depth_positive(double f, double cx, double cy, double bx, const SE3& T, const Line& L,
                                       const double* x1, const double* x2) {
  const M3 R = line_getR(L);
  const V3 X0 = scale(m3_col(R, 1), L.alpha);
  const V3 ld = m3_col(R, 0);
  const V3 b{bx, 0, 0};
  const V3 X0l = add(se3_map(T, X0), b);
  const V3 ldl = sub(add(se3_map(T, add(X0, ld)), b), X0l);
  double d1, d2, p;
  reproject_line_point(X0l, ldl, x1[0], x1[1], f, cx, cy, &d1, &p);
  reproject_line_point(X0l, ldl, x2[0], x2[1], f, cx, cy, &d2, &p);
  return !(d1 < 0 || d2 < 0);
}

// ---------------------------------------------------------------- Huber (core/robust_kernel_impl.cpp:65-91)
struct Huber { double delta, dsqr; };
static inline Huber huber_make(double delta) { return Huber{delta, delta * delta}; }
static inline void huber_robustify(const Huber& h, double e, double* rho) {
  if (e <= h.dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
  else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * h.delta - h.dsqr;
    rho[1] = h.delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

// BaseEdge::chi2 with information = s*I (core/base_edge.h:58-61): e . (s*e)
static inline double chi2_iso(const double* e, int dim, double s) {
  double c = 0;
  for (int i = 0; i < dim; i++) c += e[i] * (s * e[i]);
  return c;
}

// GetReprojThrPyramid (src/LineMatching.cc:239-247) with LinePyrFactor = 1.44 (:27)
static inline double reproj_thr_pyramid(double base, int lev) {
  double t = base;
  for (int i = 0; i < lev; i++) t *= 1.44;
  return t;
}

}  // namespace lldo
#endif
