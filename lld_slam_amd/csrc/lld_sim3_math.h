// lld_sim3_math.h — g2o::Sim3 (Thirdparty/g2o/g2o/types/sim3.h) for the device kernels: exp (the constructor from a 7-vector), map,
// product, inverse and log, branch for branch as the reference writes them.
#ifndef LLD_SIM3_MATH_H
#define LLD_SIM3_MATH_H

#include "lld_device_math.h"

namespace {

using namespace lld;

struct Sim3 { Quat r; Vec3 t; double s; };

__device__ __forceinline__ Mat3 skew3(const Vec3& v) {
  Mat3 r;
  r.m[0][0] = 0; r.m[0][1] = -v.z; r.m[0][2] = v.y;
  r.m[1][0] = v.z; r.m[1][1] = 0; r.m[1][2] = -v.x;
  r.m[2][0] = -v.y; r.m[2][1] = v.x; r.m[2][2] = 0;
  return r;
}
__device__ __forceinline__ Mat3 mat_mat(const Mat3& a, const Mat3& b) {
  Mat3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}

// Sim3(const Vector7d& update)  (types/sim3.h:64-131): update = (omega, upsilon, sigma)
__device__ Sim3 sim3_exp(const double* u) {
  const Vec3 omega = vec3(u[0], u[1], u[2]), upsilon = vec3(u[3], u[4], u[5]);
  const double sigma = u[6];
  const double theta = sqrt(dot(omega, omega));
  const Mat3 Omega = skew3(omega);
  const double s = exp(sigma);
  const Mat3 Omega2 = mat_mat(Omega, Omega);
  Mat3 R;
  const double eps = 0.00001;
  double A, B, C;
  const bool small = theta < eps;
  if (small) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) R.m[i][j] = ((i == j ? 1.0 : 0.0) + Omega.m[i][j]) + Omega2.m[i][j];
  } else {
    const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) R.m[i][j] = ((i == j ? 1.0 : 0.0) + a * Omega.m[i][j]) + b * Omega2.m[i][j];
  }
  if (fabs(sigma) < eps) {
    C = 1;
    if (small) { A = 1. / 2.; B = 1. / 6.; }
    else {
      const double theta2 = theta * theta;
      A = (1 - cos(theta)) / (theta2);
      B = (theta - sin(theta)) / (theta2 * theta);
    }
  } else {
    C = (s - 1) / sigma;
    if (small) {
      const double sigma2 = sigma * sigma;
      A = ((sigma - 1) * s + 1) / sigma2;
      B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma);
    } else {
      const double a = s * sin(theta), b = s * cos(theta);
      const double theta2 = theta * theta, sigma2 = sigma * sigma;
      const double c = theta2 + sigma2;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  Sim3 r; r.r = quat_from_rotation(R); r.s = s;
  Mat3 W;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) W.m[i][j] = (A * Omega.m[i][j] + B * Omega2.m[i][j]) + C * (i == j ? 1.0 : 0.0);
  r.t = mat_mul(W, upsilon);
  return r;
}
__device__ __forceinline__ Vec3 sim3_map(const Sim3& S, const Vec3& x) { return S.s * quat_rotate(S.r, x) + S.t; }       // s*(r*xyz) + t
__device__ __forceinline__ Sim3 sim3_mul(const Sim3& a, const Sim3& b) {
  Sim3 r; r.r = quat_mul(a.r, b.r); r.t = a.s * quat_rotate(a.r, b.t) + a.t; r.s = a.s * b.s; return r;
}
__device__ __forceinline__ Sim3 sim3_inverse(const Sim3& a) {
  Quat c; c.x = -a.r.x; c.y = -a.r.y; c.z = -a.r.z; c.w = a.r.w;
  Sim3 r; r.r = c; r.t = quat_rotate(c, (-1. / a.s) * a.t); r.s = 1. / a.s;
  return r;
}
__device__ __forceinline__ void sim3_store(const Sim3& S, double* d) { d[0] = S.r.x; d[1] = S.r.y; d[2] = S.r.z; d[3] = S.r.w; d[4] = S.t.x; d[5] = S.t.y; d[6] = S.t.z; d[7] = S.s; }
__device__ __forceinline__ Sim3 sim3_load(const double* d) { Sim3 S; S.r.x = d[0]; S.r.y = d[1]; S.r.z = d[2]; S.r.w = d[3]; S.t = vec3(d[4], d[5], d[6]); S.s = d[7]; return S; }


// Sim3::log (types/sim3.h:137-212); upsilon = W.lu().solve(t) is Eigen's 3x3 partial-pivoting LU
__device__ __forceinline__ Vec3 solve3_lu(const Mat3& Win, const Vec3& rhs) {
  double A[3][4] = {{Win.m[0][0], Win.m[0][1], Win.m[0][2], rhs.x}, {Win.m[1][0], Win.m[1][1], Win.m[1][2], rhs.y}, {Win.m[2][0], Win.m[2][1], Win.m[2][2], rhs.z}};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    // pivot = first row of largest magnitude, ONE swap (selects: k is a compile-time constant after unrolling, the arrays stay in registers)
    int piv = k; double best = fabs(A[k][k]);
#pragma unroll
    for (int i = k + 1; i < 3; i++) if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); piv = i; }
#pragma unroll
    for (int i = k + 1; i < 3; i++) {
      const bool sw = piv == i;
#pragma unroll
      for (int j = 0; j < 4; j++) { const double a = A[k][j], b = A[i][j]; A[k][j] = sw ? b : a; A[i][j] = sw ? a : b; }
    }
#pragma unroll
    for (int i = k + 1; i < 3; i++) {
      const double f = A[i][k] / A[k][k];
#pragma unroll
      for (int j = k; j < 4; j++) A[i][j] -= f * A[k][j];
    }
  }
  const double z = A[2][3] / A[2][2];
  const double y = (A[1][3] - A[1][2] * z) / A[1][1];
  const double x = (A[0][3] - A[0][1] * y - A[0][2] * z) / A[0][0];
  return vec3(x, y, z);
}
__device__ void sim3_log(const Sim3& S, double* res) {
  const double sigma = log(S.s);
  const Mat3 R = quat_rotation(S.r);
  const double d = 0.5 * (R.m[0][0] + R.m[1][1] + R.m[2][2] - 1);
  const Vec3 dR = vec3(R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]);
  const double eps = 0.00001;
  Vec3 omega; double A, B, C;
  const bool near = d > 1 - eps;
  if (near) omega = 0.5 * dR;
  else { const double theta = acos(d); omega = (theta / (2 * sqrt(1 - d * d))) * dR; }
  if (fabs(sigma) < eps) {
    C = 1;
    if (near) { A = 1. / 2.; B = 1. / 6.; }
    else {
      const double theta = acos(d), theta2 = theta * theta;
      A = (1 - cos(theta)) / (theta2);
      B = (theta - sin(theta)) / (theta2 * theta);
    }
  } else {
    C = (S.s - 1) / sigma;
    if (near) {
      const double sigma2 = sigma * sigma;
      A = ((sigma - 1) * S.s + 1) / (sigma2);
      B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
    } else {
      const double theta = acos(d), theta2 = theta * theta;
      const double a = S.s * sin(theta), b = S.s * cos(theta);
      const double c = theta2 + sigma * sigma;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  const Mat3 Omega = skew3(omega), OO = mat_mat(Omega, Omega);
  Mat3 W;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) W.m[i][j] = (A * Omega.m[i][j] + B * OO.m[i][j]) + C * (i == j ? 1.0 : 0.0);
  const Vec3 ups = solve3_lu(W, S.t);
  res[0] = omega.x; res[1] = omega.y; res[2] = omega.z; res[3] = ups.x; res[4] = ups.y; res[5] = ups.z; res[6] = sigma;
}

}  // namespace
#endif
