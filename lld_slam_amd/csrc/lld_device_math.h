// lld_device_math.h — fp64 geometry for the HIP kernels (gfx950).  Host+device so the tiny host-side
// conversions of the ABI share it.  Each routine names the reference behaviour it reproduces; the arithmetic
// is organised for registers (closed-form adjoint Jacobians instead of the reference's 3x3/3x6 matrix
// chains), so results agree with the reference to rounding, not bitwise.
#ifndef LLD_DEVICE_MATH_H
#define LLD_DEVICE_MATH_H

#include <hip/hip_runtime.h>

#define LLD_HD __host__ __device__ __forceinline__

namespace lld {

struct Vec3 { double x, y, z; };
struct Mat3 { double m[3][3]; };
struct Quat { double x, y, z, w; };
struct Pose { Quat q; Vec3 t; };          // g2o::SE3Quat: world -> camera
struct LineQ { Quat q; double alpha; };   // g2o::LineParams

LLD_HD Vec3 vec3(double x, double y, double z) { Vec3 v; v.x = x; v.y = y; v.z = z; return v; }
LLD_HD Vec3 operator+(const Vec3& a, const Vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
LLD_HD Vec3 operator-(const Vec3& a, const Vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
LLD_HD Vec3 operator*(double s, const Vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
LLD_HD double dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LLD_HD Vec3 cross(const Vec3& a, const Vec3& b) {
  return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Eigen::Quaterniond(Matrix3d) — trace branch / largest-diagonal branch (Eigen Quaternion.h).
LLD_HD Quat quat_from_rotation(const Mat3& R) {
  Quat q;
  double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R.m[2][1] - R.m[1][2]) * t;
    q.y = (R.m[0][2] - R.m[2][0]) * t;
    q.z = (R.m[1][0] - R.m[0][1]) * t;
  } else if (R.m[0][0] >= R.m[1][1] && R.m[0][0] >= R.m[2][2]) {        // i = 0 (ties keep the lower index)
    t = sqrt(R.m[0][0] - R.m[1][1] - R.m[2][2] + 1.0);
    q.x = 0.5 * t; t = 0.5 / t;
    q.w = (R.m[2][1] - R.m[1][2]) * t;
    q.y = (R.m[1][0] + R.m[0][1]) * t;
    q.z = (R.m[2][0] + R.m[0][2]) * t;
  } else if (R.m[1][1] >= R.m[2][2]) {                                    // i = 1
    t = sqrt(R.m[1][1] - R.m[2][2] - R.m[0][0] + 1.0);
    q.y = 0.5 * t; t = 0.5 / t;
    q.w = (R.m[0][2] - R.m[2][0]) * t;
    q.z = (R.m[2][1] + R.m[1][2]) * t;
    q.x = (R.m[0][1] + R.m[1][0]) * t;
  } else {                                                                // i = 2
    t = sqrt(R.m[2][2] - R.m[0][0] - R.m[1][1] + 1.0);
    q.z = 0.5 * t; t = 0.5 / t;
    q.w = (R.m[1][0] - R.m[0][1]) * t;
    q.x = (R.m[0][2] + R.m[2][0]) * t;
    q.y = (R.m[1][2] + R.m[2][1]) * t;
  }
  return q;
}

// Eigen QuaternionBase::toRotationMatrix
LLD_HD Mat3 quat_rotation(const Quat& q) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  Mat3 R;
  R.m[0][0] = 1.0 - (tyy + tzz); R.m[0][1] = txy - twz;         R.m[0][2] = txz + twy;
  R.m[1][0] = txy + twz;         R.m[1][1] = 1.0 - (txx + tzz); R.m[1][2] = tyz - twx;
  R.m[2][0] = txz - twy;         R.m[2][1] = tyz + twx;         R.m[2][2] = 1.0 - (txx + tyy);
  return R;
}

LLD_HD Vec3 mat_col(const Mat3& R, int c) { return vec3(R.m[0][c], R.m[1][c], R.m[2][c]); }
LLD_HD Vec3 mat_mul(const Mat3& R, const Vec3& v) {
  return vec3(R.m[0][0] * v.x + R.m[0][1] * v.y + R.m[0][2] * v.z, R.m[1][0] * v.x + R.m[1][1] * v.y + R.m[1][2] * v.z,
              R.m[2][0] * v.x + R.m[2][1] * v.y + R.m[2][2] * v.z);
}
LLD_HD Vec3 mat_tmul(const Mat3& R, const Vec3& v) {   // R^T v
  return vec3(R.m[0][0] * v.x + R.m[1][0] * v.y + R.m[2][0] * v.z, R.m[0][1] * v.x + R.m[1][1] * v.y + R.m[2][1] * v.z,
              R.m[0][2] * v.x + R.m[1][2] * v.y + R.m[2][2] * v.z);
}

// Eigen q * v: v + w*(2 q.vec x v) + q.vec x (2 q.vec x v)
LLD_HD Vec3 quat_rotate(const Quat& q, const Vec3& v) {
  const Vec3 qv = vec3(q.x, q.y, q.z);
  Vec3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}

LLD_HD Quat quat_mul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

// 1 / z and 1 / sqrt(d) for the kernels' inner loops.  On the device: v_rcp_f64 / v_rsq_f64 seed + two Newton steps - within an ulp or two of
// the correctly rounded quotient, ~8 instructions where the IEEE division and square root the compiler emits for `1.0 / z` and `sqrt(d)` are
// ~15 and ~20 dependent ones.  On the host (the few host-side uses of this header): the plain expressions.
LLD_HD double rcp_nr(double z) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(z);
  r = r * (2.0 - z * r);
  r = r * (2.0 - z * r);
  return r;
#else
  return 1.0 / z;
#endif
}
LLD_HD double rsqrt_nr(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - (0.5 * d) * (y * y));
  y = y * (1.5 - (0.5 * d) * (y * y));
  return y;
#else
  return 1.0 / sqrt(d);
#endif
}

// kFast (the batched BA kernels): one reciprocal square root and four products instead of a square root and four divisions
template <bool kFast>
LLD_HD Quat quat_unit_t(const Quat& q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  Quat r;
  if (kFast) { const double in = rsqrt_nr(n2); r.x = q.x * in; r.y = q.y * in; r.z = q.z * in; r.w = q.w * in; }
  else { const double n = sqrt(n2); r.x = q.x / n; r.y = q.y / n; r.z = q.z / n; r.w = q.w / n; }
  return r;
}
LLD_HD Quat quat_unit(const Quat& q) {
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  Quat r; r.x = q.x / n; r.y = q.y / n; r.z = q.z / n; r.w = q.w / n;
  return r;
}

// SE3Quat::normalizeRotation (types/se3quat.h:280-285)
LLD_HD void pose_normalize(Pose& p) {
  if (p.q.w < 0) { p.q.x = -p.q.x; p.q.y = -p.q.y; p.q.z = -p.q.z; p.q.w = -p.q.w; }
  p.q = quat_unit(p.q);
}
LLD_HD Pose pose_load(const double* qt) {
  Pose p; p.q.x = qt[0]; p.q.y = qt[1]; p.q.z = qt[2]; p.q.w = qt[3]; p.t = vec3(qt[4], qt[5], qt[6]);
  return p;
}
LLD_HD void pose_store(const Pose& p, double* qt) {
  qt[0] = p.q.x; qt[1] = p.q.y; qt[2] = p.q.z; qt[3] = p.q.w; qt[4] = p.t.x; qt[5] = p.t.y; qt[6] = p.t.z;
}
// SE3Quat::map (types/se3quat.h:217-220)
LLD_HD Vec3 pose_map(const Pose& p, const Vec3& X) { return quat_rotate(p.q, X) + p.t; }

// VertexSE3Expmap::oplusImpl: T <- SE3Quat::exp(update) * T  (types_six_dof_expmap.h:76-79, se3quat.h:223-257,104-110).
// update = (omega, upsilon).  The theta < 1e-5 branch keeps the reference's R = V = I + Omega + Omega^2.
LLD_HD Pose pose_oplus(const Pose& T, const double* u) {
  const Vec3 w = vec3(u[0], u[1], u[2]), v = vec3(u[3], u[4], u[5]);
  const double theta = sqrt(dot(w, w));
  double a, b, c;          // R = I + a*W + b*W^2 ; V = I + b'*W + c*W^2
  double bv;
  if (theta < 0.00001) { a = 1.0; b = 1.0; bv = 1.0; c = 1.0; }
  else {
    const double s = sin(theta), co = cos(theta);
    a = s / theta; b = (1 - co) / (theta * theta); bv = b; c = (theta - s) / (theta * theta * theta);
  }
  // W^2 = w w^T - |w|^2 I
  const double ww = dot(w, w);
  Mat3 R;
  const double W2[3][3] = {{w.x * w.x - ww, w.x * w.y, w.x * w.z}, {w.y * w.x, w.y * w.y - ww, w.y * w.z}, {w.z * w.x, w.z * w.y, w.z * w.z - ww}};
  const double W1[3][3] = {{0, -w.z, w.y}, {w.z, 0, -w.x}, {-w.y, w.x, 0}};
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) R.m[i][j] = ((i == j ? 1.0 : 0.0) + a * W1[i][j]) + b * W2[i][j];
  }
  // V * upsilon = v + bv * (w x v) + c * (w x (w x v))
  const Vec3 wv = cross(w, v);
  const Vec3 wwv = cross(w, wv);
  Pose E;
  E.q = quat_from_rotation(R);
  E.t = v + bv * wv + c * wwv;
  pose_normalize(E);
  Pose r;
  r.t = E.t + quat_rotate(E.q, T.t);
  r.q = quat_mul(E.q, T.q);
  pose_normalize(r);
  return r;
}

// LineOptimizer::AddLineMinimal (src/LineOptimizer.cc:44-50): R = [dir | X0/|X0| | dir x X0/|X0|]
LLD_HD LineQ line_from_x0_dir(const Vec3& X0, const Vec3& d) {
  const double n = sqrt(dot(X0, X0));
  const Vec3 c1 = vec3(X0.x / n, X0.y / n, X0.z / n);
  const Vec3 cr = cross(d, X0);
  const Vec3 c2 = vec3(cr.x / n, cr.y / n, cr.z / n);
  Mat3 R;
  R.m[0][0] = d.x; R.m[1][0] = d.y; R.m[2][0] = d.z;
  R.m[0][1] = c1.x; R.m[1][1] = c1.y; R.m[2][1] = c1.z;
  R.m[0][2] = c2.x; R.m[1][2] = c2.y; R.m[2][2] = c2.z;
  LineQ l; l.q = quat_from_rotation(R); l.alpha = n;
  return l;
}
// LineParams::GetR — every read normalises q (types/types_sba.cpp:77-79,89-92)
LLD_HD Mat3 line_rotation(const LineQ& l) { return quat_rotation(quat_unit(l.q)); }
template <bool kFast> LLD_HD Mat3 line_rotation_t(const LineQ& l) { return quat_rotation(quat_unit_t<kFast>(l.q)); }
// VertexSBALine::oplusImpl (types/types_sba.h:93-104)
LLD_HD LineQ line_oplus(const LineQ& l, const double* u) {
  Quat qr; qr.x = u[0]; qr.y = u[1]; qr.z = u[2];
  qr.w = sqrt(1.0 - (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
  LineQ r; r.q = quat_mul(qr, quat_unit(l.q)); r.alpha = l.alpha + u[3];
  return r;
}

struct CamK { double fx, fy, cx, cy, bf; float bf_f; double bx_right; };
// bx_right = -(float)bf / (float)fx: KeyFrame::mbf and mK are floats (src/Optimizer.cc:1216, :632-637)
LLD_HD CamK make_camk(const lld_camera& c) {
  CamK k; k.fx = c.fx; k.fy = c.fy; k.cx = c.cx; k.cy = c.cy; k.bf = c.bf; k.bf_f = (float)c.bf;
  k.bx_right = -(double)((float)c.bf / (float)c.fx);
  return k;
}

// ---------------------------------------------------------------- point edges
// Residuals.  EdgeStereoSE3ProjectXYZ::cam_project keeps `invz` (and, in the binary edge, `bf`) in float
// (types_six_dof_expmap.cpp:152-159, :305-312); `binary` selects float bf*invz (LBA) vs double (pose-only).
// The same with 1 / Xc.z supplied by the caller (the batched BA kernels need it for the Jacobians anyway and pass rcp_nr(Xc.z): the
// float `invz` of a stereo edge then differs from the reference's only where the double quotient sits within an ulp of a float rounding
// boundary, the mono projection by an ulp of its double product).
LLD_HD void point_residual_iz(const CamK& k, const Vec3& Xc, double iz, double u, double v, double ur, bool stereo, bool binary, double* e) {
  if (stereo) {
    const float invz = (float)iz;
    const double r0 = Xc.x * (double)invz * k.fx + k.cx;
    const double r1 = Xc.y * (double)invz * k.fy + k.cy;
    const double shift = binary ? (double)(k.bf_f * invz) : k.bf * (double)invz;
    e[0] = u - r0; e[1] = v - r1; e[2] = ur - (r0 - shift);
  } else {
    e[0] = u - (Xc.x * iz * k.fx + k.cx);
    e[1] = v - (Xc.y * iz * k.fy + k.cy);
    e[2] = 0.0;
  }
}
LLD_HD void point_residual(const CamK& k, const Vec3& Xc, double u, double v, double ur, bool stereo, bool binary, double* e) {
  if (stereo) {
    const float invz = (float)(1.0 / Xc.z);
    const double r0 = Xc.x * (double)invz * k.fx + k.cx;
    const double r1 = Xc.y * (double)invz * k.fy + k.cy;
    const double shift = binary ? (double)(k.bf_f * invz) : k.bf * (double)invz;
    e[0] = u - r0; e[1] = v - r1; e[2] = ur - (r0 - shift);
  } else {
    e[0] = u - (Xc.x / Xc.z * k.fx + k.cx);
    e[1] = v - (Xc.y / Xc.z * k.fy + k.cy);
    e[2] = 0.0;
  }
}

// Pose Jacobian rows (types_six_dof_expmap.cpp:119-141 / :212-236); Jc[r*6+c], rows 0..2 (row 2 only if stereo)
LLD_HD void point_jac_pose(const CamK& k, const Vec3& Xc, bool stereo, double* Jc) {
  const double x = Xc.x, y = Xc.y, iz = 1.0 / Xc.z, iz2 = iz * iz;
  Jc[0] = x * y * iz2 * k.fx;
  Jc[1] = -(1 + x * x * iz2) * k.fx;
  Jc[2] = y * iz * k.fx;
  Jc[3] = -iz * k.fx;
  Jc[4] = 0;
  Jc[5] = x * iz2 * k.fx;
  Jc[6] = (1 + y * y * iz2) * k.fy;
  Jc[7] = -x * y * iz2 * k.fy;
  Jc[8] = -x * iz * k.fy;
  Jc[9] = 0;
  Jc[10] = -iz * k.fy;
  Jc[11] = y * iz2 * k.fy;
  if (stereo) {
    Jc[12] = Jc[0] - k.bf * y * iz2;
    Jc[13] = Jc[1] + k.bf * x * iz2;
    Jc[14] = Jc[2];
    Jc[15] = Jc[3];
    Jc[16] = 0;
    Jc[17] = Jc[5] - k.bf * iz2;
  } else {
    Jc[12] = Jc[13] = Jc[14] = Jc[15] = Jc[16] = Jc[17] = 0;
  }
}
// Point Jacobian rows (types_six_dof_expmap.cpp:196-210 / :124): Jp[r*3+c]
LLD_HD void point_jac_point(const CamK& k, const Vec3& Xc, const Mat3& R, bool stereo, double* Jp) {
  const double iz = 1.0 / Xc.z, iz2 = iz * iz;
  for (int j = 0; j < 3; j++) {
    Jp[j] = -k.fx * R.m[0][j] * iz + k.fx * Xc.x * R.m[2][j] * iz2;
    Jp[3 + j] = -k.fy * R.m[1][j] * iz + k.fy * Xc.y * R.m[2][j] * iz2;
    Jp[6 + j] = stereo ? Jp[j] - k.bf * R.m[2][j] * iz2 : 0.0;
  }
}

// Hpl block of a point edge, W (6x3, row-major) = ws * Jc^T Jp, recomputed from the linearisation-point state instead of being stored
// (BaseBinaryEdge::constructQuadraticForm's `_hessian`, core/base_binary_edge.hpp:84-105, transposed), in closed form:
// with A = d(u, v, uR)/dXc (3x3, third row only for stereo) the two Jacobians are Jp = -A R and
// Jc = [A [Xc]x | -A], hence W = ws Jc^T Jp = [ [Xc]x G ; G ] with G = ws (A^T A) R: one symmetric 3x3 (with a structural zero),
// one 3x3 product and three cross products instead of two full Jacobians and a 6x3x3 contraction.  R = quat_rotation(T.q) is passed in because the callers keep it per lane.
// `iz` = 1 / Xc.z as the caller computes it: the Schur kernel passes v_rcp_f64 + two Newton steps (a Jacobian entry, like the linearisation's:
// an IEEE division is ~40 dependent instructions of the ~200 its staging lane spends per (landmark, camera)).
LLD_HD void point_hpl_closed_iz(const CamK& k, const Vec3& Xc, double iz, const Mat3& R, bool stereo, double ws, double* W);
LLD_HD void point_hpl_closed(const CamK& k, const Pose& T, const Mat3& R, const Vec3& X, bool stereo, double ws, double* W) {
  const Vec3 Xc = mat_mul(R, X) + T.t;
  point_hpl_closed_iz(k, Xc, 1.0 / Xc.z, R, stereo, ws, W);
}
LLD_HD void point_hpl_closed_iz(const CamK& k, const Vec3& Xc, double iz, const Mat3& R, bool stereo, double ws, double* W) {
  const double iz2 = iz * iz;
  const double a = k.fx * iz, b = k.fy * iz;
  const double c0 = -k.fx * Xc.x * iz2, c1 = -k.fy * Xc.y * iz2, c2 = c0 + k.bf * iz2;
  // ws * A^T A
  const double m00 = ws * (stereo ? 2.0 * a * a : a * a);
  const double m02 = ws * (stereo ? a * (c0 + c2) : a * c0);
  const double m11 = ws * (b * b), m12 = ws * (b * c1);
  const double m22 = ws * (stereo ? c0 * c0 + c1 * c1 + c2 * c2 : c0 * c0 + c1 * c1);
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const double g0 = m00 * R.m[0][j] + m02 * R.m[2][j];
    const double g1 = m11 * R.m[1][j] + m12 * R.m[2][j];
    const double g2 = m02 * R.m[0][j] + m12 * R.m[1][j] + m22 * R.m[2][j];
    W[0 * 3 + j] = Xc.y * g2 - Xc.z * g1;               // Xc x g
    W[1 * 3 + j] = Xc.z * g0 - Xc.x * g2;
    W[2 * 3 + j] = Xc.x * g1 - Xc.y * g0;
    W[3 * 3 + j] = g0; W[4 * 3 + j] = g1; W[5 * 3 + j] = g2;
  }
}

// G = ws (A^T A) R of the block above alone (rows g0, g1, g2: G[r * 3 + j]); W = [ [Xc]x G ; G ].  The Schur staging applies L^-T to G first
// and the cross products afterwards (lld_ba_kernels.h schur_stage_point).
LLD_HD void point_g_closed_iz(const CamK& k, const Vec3& Xc, double iz, const Mat3& R, bool stereo, double ws, double* G) {
  const double iz2 = iz * iz;
  const double a = k.fx * iz, b = k.fy * iz;
  const double c0 = -k.fx * Xc.x * iz2, c1 = -k.fy * Xc.y * iz2, c2 = c0 + k.bf * iz2;
  const double m00 = ws * (stereo ? 2.0 * a * a : a * a);
  const double m02 = ws * (stereo ? a * (c0 + c2) : a * c0);
  const double m11 = ws * (b * b), m12 = ws * (b * c1);
  const double m22 = ws * (stereo ? c0 * c0 + c1 * c1 + c2 * c2 : c0 * c0 + c1 * c1);
#pragma unroll
  for (int j = 0; j < 3; j++) {
    G[j] = m00 * R.m[0][j] + m02 * R.m[2][j];
    G[3 + j] = m11 * R.m[1][j] + m12 * R.m[2][j];
    G[6 + j] = m02 * R.m[0][j] + m12 * R.m[1][j] + m22 * R.m[2][j];
  }
}

// ---------------------------------------------------------------- line edges
// Residual of EdgeSE3ProjectLine / OnlyPose (types_six_dof_expmap.h:344-375, :403-418) plus, optionally, the adjoint
// vectors a1,a2 with  d r_k = a1[k] . dX1m + a2[k] . dX2m  (X1m, X2m = endpoints in the camera frame, without b).
// K1 = [[f,0,cx],[0,f,cy],[0,0,1]] with the single focal f = fx.
struct LineAdj { Vec3 a1[2], a2[2]; };
template <bool kFast>
LLD_HD void line_residual_t(const CamK& k, double bx, const Vec3& X1m, const Vec3& X2m, double xs, double ys, double xe, double ye,
                            double* e, LineAdj* adj) {
  const double f = k.fx;
  const Vec3 P1 = vec3(f * (X1m.x + bx) + k.cx * X1m.z, f * X1m.y + k.cy * X1m.z, X1m.z);
  const Vec3 P2 = vec3(f * (X2m.x + bx) + k.cx * X2m.z, f * X2m.y + k.cy * X2m.z, X2m.z);
  const Vec3 lt = cross(P1, P2);
  const double n2 = lt.x * lt.x + lt.y * lt.y;
  // kFast (the batched BA kernels): 1 / n by rsqrt_nr and 1 / n^3 as its cube instead of a square root and two divisions
  double in;
  if (kFast) in = rsqrt_nr(n2); else { const double n = sqrt(n2); in = 1.0 / n; }
  const double d1 = xs * lt.x + ys * lt.y + lt.z, d2 = xe * lt.x + ye * lt.y + lt.z;   // x_k . l~
  e[0] = d1 * in; e[1] = d2 * in;
  if (adj) {
    const double in3 = kFast ? in * in * in : in / n2;
    const double px[2] = {xs, xe}, py[2] = {ys, ye}, dd[2] = {d1, d2};
    for (int q = 0; q < 2; q++) {
      // g = D^T x_k = x_k/n - (x_k . l~)/n^3 * (l~x, l~y, 0)
      const Vec3 g = vec3(px[q] * in - dd[q] * in3 * lt.x, py[q] * in - dd[q] * in3 * lt.y, in);
      const Vec3 h1 = cross(P2, g), h2 = cross(g, P1);    // d r = h1 . dP1 + h2 . dP2
      adj->a1[q] = vec3(f * h1.x, f * h1.y, k.cx * h1.x + k.cy * h1.y + h1.z);   // K1^T h
      adj->a2[q] = vec3(f * h2.x, f * h2.y, k.cx * h2.x + k.cy * h2.y + h2.z);
    }
  }
}
LLD_HD void line_residual(const CamK& k, double bx, const Vec3& X1m, const Vec3& X2m, double xs, double ys, double xe, double ye,
                          double* e, LineAdj* adj) { line_residual_t<false>(k, bx, X1m, X2m, xs, ys, xe, ye, e, adj); }
// Pose Jacobian (2x6) from the adjoints: d r / d omega = X1m x a1 + X2m x a2, d r / d upsilon = a1 + a2
// (FormJacobianLineWRTCam, types_six_dof_expmap.cpp:472-497, A_i = [-K skew(X_im) | K]).
LLD_HD void line_jac_pose(const LineAdj& adj, const Vec3& X1m, const Vec3& X2m, double* Jc) {
  for (int q = 0; q < 2; q++) {
    const Vec3 r = cross(X1m, adj.a1[q]) + cross(X2m, adj.a2[q]);
    const Vec3 s = adj.a1[q] + adj.a2[q];
    Jc[q * 6 + 0] = r.x; Jc[q * 6 + 1] = r.y; Jc[q * 6 + 2] = r.z;
    Jc[q * 6 + 3] = s.x; Jc[q * 6 + 4] = s.y; Jc[q * 6 + 5] = s.z;
  }
}
// Line Jacobian (2x4) (EdgeSE3ProjectLine::linearize, types_six_dof_expmap.cpp:523-541):
// dX1 = [-2 skew(alpha c1) | c1] delta, dX2 = dX1 + [-2 skew(c0) | 0] delta, mapped by R_cam.
LLD_HD void line_jac_line(const LineAdj& adj, const Mat3& Rcam, const Vec3& c0, const Vec3& c1, double alpha, double* Jl) {
  const Vec3 X1 = alpha * c1;
  for (int q = 0; q < 2; q++) {
    const Vec3 b1 = mat_tmul(Rcam, adj.a1[q]), b2 = mat_tmul(Rcam, adj.a2[q]);
    const Vec3 bs = b1 + b2;
    const Vec3 r = 2.0 * (cross(X1, bs) + cross(c0, b2));
    Jl[q * 4 + 0] = r.x; Jl[q * 4 + 1] = r.y; Jl[q * 4 + 2] = r.z; Jl[q * 4 + 3] = dot(bs, c1);
  }
}

// vgl::ReprojectLinePointTo3D (src/vgl.cc:336-346): least-squares (depth, param) of
//   [ (px,py,1) | -K ldir ] (depth, param)^T = K X0 ; solved here through the 2x2 normal equations (the 3x2 matrix
// has full column rank whenever the pixel ray is not parallel to the projected direction).  Only the sign of depth is used.
LLD_HD double reproject_depth(const CamK& k, const Vec3& X0, const Vec3& ld, double px, double py) {
  const double f = k.fx;
  const Vec3 a = vec3(px, py, 1.0);
  const Vec3 b = vec3(-(f * ld.x + k.cx * ld.z), -(f * ld.y + k.cy * ld.z), -ld.z);
  const Vec3 r = vec3(f * X0.x + k.cx * X0.z, f * X0.y + k.cy * X0.z, X0.z);
  const double aa = dot(a, a), ab = dot(a, b), bb = dot(b, b), ar = dot(a, r), br = dot(b, r);
  return (bb * ar - ab * br) / (aa * bb - ab * ab);
}
// EdgeSE3ProjectLine::IsDepthPositive (types_six_dof_expmap.h:312-342)
LLD_HD bool line_depth_positive(const CamK& k, double bx, const Pose& T, const Vec3& c0, const Vec3& c1, double alpha,
                                double xs, double ys, double xe, double ye) {
  const Vec3 X0 = alpha * c1;
  Vec3 X0l = pose_map(T, X0); X0l.x += bx;
  Vec3 X1l = pose_map(T, X0 + c0); X1l.x += bx;
  const Vec3 ldl = X1l - X0l;
  const double d1 = reproject_depth(k, X0l, ldl, xs, ys);
  const double d2 = reproject_depth(k, X0l, ldl, xe, ye);
  return !(d1 < 0 || d2 < 0);
}

// RobustKernelHuber::robustify (core/robust_kernel_impl.cpp:78-91): returns rho0, writes rho1
// The same without a branch and without the square root / division pair (the batched BA kernels): nearly every wavefront holds an outlier edge,
// so all of them took the branch; sqrt(e) = e / sqrt(e).
LLD_HD double huber_nr(double e, double delta, double* rho1) {
  const double dsqr = delta * delta;
  const bool in = e <= dsqr;
  const double is = rsqrt_nr(in ? 1.0 : e);
  *rho1 = in ? 1.0 : delta * is;
  return in ? e : 2 * (e * is) * delta - dsqr;
}
LLD_HD double huber(double e, double delta, double* rho1) {
  const double dsqr = delta * delta;
  if (e <= dsqr) { *rho1 = 1.0; return e; }
  const double s = sqrt(e);
  *rho1 = delta / s;
  return 2 * s * delta - dsqr;
}

// GetReprojThrPyramid(1.0, lev)^2 -> information divisor (src/LineMatching.cc:239-247, LinePyrFactor 1.44)
LLD_HD double line_info(double gamma, int octave) {
  double t = 1.0;
  for (int i = 0; i < octave; i++) t *= 1.44;
  double info = 1.0 * (gamma * gamma);
  info /= t * t;
  return info;
}

}  // namespace lld
#endif
