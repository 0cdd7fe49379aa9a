// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
// (lld_slam_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the reference (ORB-SLAM2 + vendored g2o) cannot be compiled in this
// environment (no Eigen, no OpenCV) and ships no tests or golden vectors, so this restatement
// is pinned only by the known-answer tests in tests/test_oracle_*.py (scipy expm, finite
// differences, dense normal-equation solves, closed forms).
//
// lldo_math.h — small fixed-size linear algebra that restates the Eigen operations the
// reference's hot path relies on.  Every function names the reference / Eigen behaviour it
// follows.  Plain scalar double code, no FMA contraction (built with -ffp-contract=off).
#ifndef LLDO_MATH_H
#define LLDO_MATH_H

#include <cmath>
#include <cstring>

namespace lldo {

struct V3 { double x, y, z; };
struct M3 { double m[3][3]; };
struct Quat { double x, y, z, w; };   // Eigen coeffs() order: x,y,z,w

static inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
static inline V3 add(const V3& a, const V3& b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 sub(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 scale(const V3& a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
// Eigen cross(): (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
static inline V3 cross(const V3& a, const V3& b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline double at(const V3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static inline M3 m3_identity() {
  M3 r; std::memset(&r, 0, sizeof r); r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0; return r;
}
static inline M3 m3_mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
static inline V3 m3_mulv(const M3& a, const V3& v) {
  return V3{a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
            a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
static inline V3 m3_col(const M3& a, int c) { return V3{a.m[0][c], a.m[1][c], a.m[2][c]}; }

// g2o skew (types/se3_ops.hpp:27-38) == cpmat (types/types_six_dof_expmap.cpp:36-43)
static inline M3 skew(const V3& v) {
  M3 r; std::memset(&r, 0, sizeof r);
  r.m[0][1] = -v.z; r.m[0][2] = v.y; r.m[1][2] = -v.x;
  r.m[1][0] = v.z;  r.m[2][0] = -v.y; r.m[2][1] = v.x;
  return r;
}

// Eigen::Quaterniond(Matrix3d)  (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>)
static inline Quat quat_from_R(const M3& R) {
  Quat q;
  double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R.m[2][1] - R.m[1][2]) * t;
    q.y = (R.m[0][2] - R.m[2][0]) * t;
    q.z = (R.m[1][0] - R.m[0][1]) * t;
  } else {
    int i = 0;
    if (R.m[1][1] > R.m[0][0]) i = 1;
    if (R.m[2][2] > R.m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
    double c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R.m[k][j] - R.m[j][k]) * t;
    c[j] = (R.m[j][i] + R.m[i][j]) * t;
    c[k] = (R.m[k][i] + R.m[i][k]) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}

// Eigen QuaternionBase::toRotationMatrix
static inline M3 quat_to_R(const Quat& q) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 R;
  R.m[0][0] = 1.0 - (tyy + tzz); R.m[0][1] = txy - twz;         R.m[0][2] = txz + twy;
  R.m[1][0] = txy + twz;         R.m[1][1] = 1.0 - (txx + tzz); R.m[1][2] = tyz - twx;
  R.m[2][0] = txz - twy;         R.m[2][1] = tyz + twx;         R.m[2][2] = 1.0 - (txx + tyy);
  return R;
}

// Eigen QuaternionBase::_transformVector: uv = 2 * (vec x v); v + w*uv + vec x uv
static inline V3 quat_rot(const Quat& q, const V3& v) {
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = add(uv, uv);
  V3 c2 = cross(qv, uv);
  return V3{v.x + q.w * uv.x + c2.x, v.y + q.w * uv.y + c2.y, v.z + q.w * uv.z + c2.z};
}

// Eigen quaternion product (Hamilton)
static inline Quat quat_mul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

static inline Quat quat_normalized(const Quat& q) {
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / n, q.y / n, q.z / n, q.w / n};
}

// ------------------------------------------------------------------ SE3Quat (types/se3quat.h)
struct SE3 { Quat q; V3 t; };

// SE3Quat::normalizeRotation (se3quat.h:280-285)
static inline void se3_normalize_rotation(SE3& T) {
  if (T.q.w < 0) { T.q.x *= -1; T.q.y *= -1; T.q.z *= -1; T.q.w *= -1; }
  T.q = quat_normalized(T.q);
}
// SE3Quat(const Matrix3d&, const Vector3d&) (se3quat.h:58-60)
static inline SE3 se3_from_Rt(const M3& R, const V3& t) {
  SE3 T{quat_from_R(R), t};
  se3_normalize_rotation(T);
  return T;
}
// SE3Quat::map (se3quat.h:217-220)
static inline V3 se3_map(const SE3& T, const V3& X) { return add(quat_rot(T.q, X), T.t); }
// SE3Quat::operator* (se3quat.h:104-110)
static inline SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 r = a;
  r.t = add(r.t, quat_rot(a.q, b.t));
  r.q = quat_mul(a.q, b.q);
  se3_normalize_rotation(r);
  return r;
}
// SE3Quat::exp (se3quat.h:223-257); update = (omega, upsilon)
static inline SE3 se3_exp(const double* u) {
  V3 omega{u[0], u[1], u[2]}, upsilon{u[3], u[4], u[5]};
  const double theta = norm(omega);
  const M3 Omega = skew(omega);
  M3 R, V;
  if (theta < 0.00001) {
    // "TODO: CHECK WHETHER THIS IS CORRECT!!!" in the reference: R = I + Omega + Omega*Omega; V = R
    const M3 O2 = m3_mul(Omega, Omega);
    const M3 I = m3_identity();
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R.m[i][j] = (I.m[i][j] + Omega.m[i][j]) + O2.m[i][j];
    V = R;
  } else {
    const M3 O2 = m3_mul(Omega, Omega);
    const M3 I = m3_identity();
    const double a = std::sin(theta) / theta;
    const double b = (1 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / (std::pow(theta, 3));
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        R.m[i][j] = (I.m[i][j] + a * Omega.m[i][j]) + b * O2.m[i][j];
        V.m[i][j] = (I.m[i][j] + b * Omega.m[i][j]) + c * O2.m[i][j];
      }
  }
  SE3 T{quat_from_R(R), m3_mulv(V, upsilon)};
  se3_normalize_rotation(T);    // SE3Quat(const Quaterniond&, const Vector3d&) ctor, se3quat.h:62-64
  return T;
}

// ------------------------------------------------------------------ LineParams (types/types_sba.h:62-108, types_sba.cpp:58-92)
struct Line { Quat q; double alpha; };   // q stored un-normalised; every read normalises

static inline Quat line_getq(const Line& l) { return quat_normalized(l.q); }
static inline M3 line_getR(const Line& l) { return quat_to_R(line_getq(l)); }
// LineOptimizer::AddLineMinimal initialisation (src/LineOptimizer.cc:44-50)
static inline Line line_from_x0_dir(const V3& X0, const V3& dir) {
  const double n = norm(X0);
  M3 R;
  const V3 c1 = V3{X0.x / n, X0.y / n, X0.z / n};
  const V3 cr = cross(dir, X0);
  const V3 c2 = V3{cr.x / n, cr.y / n, cr.z / n};
  R.m[0][0] = dir.x; R.m[1][0] = dir.y; R.m[2][0] = dir.z;
  R.m[0][1] = c1.x;  R.m[1][1] = c1.y;  R.m[2][1] = c1.z;
  R.m[0][2] = c2.x;  R.m[1][2] = c2.y;  R.m[2][2] = c2.z;
  return Line{quat_from_R(R), n};
}
// VertexSBALine::oplusImpl (types_sba.h:93-104)
static inline void line_oplus(Line& l, const double* u) {
  Quat qr;
  qr.x = u[0]; qr.y = u[1]; qr.z = u[2];
  qr.w = std::sqrt(1.0 - (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
  l.q = quat_mul(qr, line_getq(l));
  l.alpha = l.alpha + u[3];
}

}  // namespace lldo
#endif
