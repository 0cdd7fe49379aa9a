// lld_pose.hip — Optimizer::PoseOptimization (src/Optimizer.cc:653-932, AddLineMinOnlyPose :562-650) as ONE kernel:
// one workgroup per frame runs the whole protocol (4 rounds x 10 Levenberg–Marquardt iterations, outlier
// classification between rounds) without host round trips.  The frame's observations are copied ONCE from HBM into LDS together with
// the per-edge working state (chi2, level, robust kernel, outlier flag), so the ~80 sweeps of the protocol never leave the CU.
// LDS image of a frame (round 4): world points and per-LINE end points as doubles, the image observations as FLOAT records when every
// one of them is a widened float (what the reference's key points, uRight, key lines and level sigmas are - checked on the host, the
// doubles as given otherwise), chi2 as the float the classification compares (Optimizer.cc:849-860 casts it), no chi2 for line edges
// (never read: the classification re-evaluates every line edge, :893-911): 45 B per point + 48 B per line + 29 B per line edge =
// 66 KB for 1000 points + 200 stereo lines instead of 106 KB, i.e. TWO frames per CU.  Batches with more frames than CUs run
// 256 lanes per frame (one wavefront per SIMD and frame, two frames per CU: while one frame waits for its reductions, barriers and
// the one-lane 6x6 solve, the other one sweeps), smaller batches and single calls 512 lanes (shortest chain for one frame).  The lanes
// sweep the edges, build the 6x6 normal equations in registers (21+6+1 fp64 partials per lane), reduce them with a fixed shuffle
// tree + LDS; ONE lane solves the damped 6x6 system (LDL^T) and applies the update, the trial pose travels through LDS.  g2o semantics
// kept: Huber weights use rho' only, chi2 is float-compared against 5.991f/7.815f, per-edge errors go stale exactly as in the reference
// (only outliers are re-evaluated before classification).  Frames too large for LDS run the same code on HBM-resident working arrays.
#include "lld_common.h"
#include "lld_device_math.h"
#include "lld_track_internal.h"

namespace {

using namespace lld;

constexpr int kPoseThreadsMax = 512;
constexpr size_t kPoseLdsBudget = 150 * 1024;              // dynamic LDS available to the staged frame (160 KB per CU)
constexpr size_t kPoseLdsBudgetPair = 78 * 1024;           // ... to each of two co-resident frames (the kernel's static LDS is 2.2 KB per workgroup)
constexpr uint8_t PF_LEVEL = 1, PF_ROBUST = 2, PF_OUTLIER = 4;             // point working flags
constexpr uint8_t LF_LEVEL = 1, LF_ROBUST = 2, LF_LAST = 4, LF_STEREO = 8, LF_THR_STEREO = 16; // line-edge flags (LAST / STEREO / THR_STEREO are inputs)

#ifdef LLD_EXPERIMENTS
#define LLD_PO_T0() long long po_t_ = (long long)__builtin_amdgcn_s_memtime()
#define LLD_PO_LAP(k) do { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); po_acc[(k)] += n_ - po_t_; po_t_ = n_; } while (0)
#define LLD_PO_COUNT(k) do { po_acc[(k)] += 1; } while (0)
#else
#define LLD_PO_T0() do {} while (0)
#define LLD_PO_LAP(k) do {} while (0)
#define LLD_PO_COUNT(k) do {} while (0)
#endif
enum { PO_STAGE = 0, PO_BUILD, PO_REDUCE, PO_SOLVE, PO_TRIAL_SWEEP, PO_TRIAL_SUM, PO_CLASSIFY, PO_ROUND_HEAD, PO_N_ITS, PO_N_TRIALS, PO_N_SOLVES, PO_TOTAL, PO_N_STAMPS };

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

struct PoseArrays {
  const double* pt[7];           // px, py, pz, u, v, ur, s                    [n_pt_total]
  const double* le[12];          // x1x,x1y,x1z, x2x,x2y,x2z, xs,ys,xe,ye, info, bx   [n_le_total]
  const int* le_line;            // line of the edge (frame-local)
  const uint8_t* le_fl0;         // LF_LAST: last edge of its line; LF_STEREO: the line has a right-image edge; LF_THR_STEREO: the
                                 // classification threshold the reference applies to this edge is the stereo one (see pose_pack)
  // working state of the HBM mode (frames that do not fit LDS)
  double *pt_chi2, *le_chi2; uint8_t *pt_fl, *le_fl;
  // results
  uint8_t *pt_outlier, *ln_outlier;
  // device-resident frames (lld_frame_track_*): edge e belongs to keypoint pt_kp[e], line l of the problem to frame line ln_fi[l]; the flags go
  // straight into Frame::mvbOutlier / mvbOutlierLines (kp_outlier / fl_outlier), the result also to track_out.  All null otherwise.
  const int* pt_kp = nullptr; const int* ln_fi = nullptr;
  uint8_t* kp_outlier = nullptr; uint8_t* fl_outlier = nullptr;
  double* track_out = nullptr;
  long long* stamps = nullptr;      // experiments build: per-stage s_memtime sums of frame 0's wavefront 0 (tools/pose_stage_budget.py)
};

struct PoseOut { double qt[7]; double chi2; int n_inliers, lm_iterations, lm_trials, pad; };

// ---- cross-lane sums without the LDS (round 6).  __shfl_xor compiles to ds_bpermute_b32 - two LDS instructions and a round trip of the CU's
// one LDS pipe per double and step; a frame's kernel does 28-value sums twenty times and one-value sums forty-five times with eight wavefronts
// queueing for that pipe.  gfx950 can pair lanes in the vector ALU instead: v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and
// even rows of two registers, DPP moves pair lanes inside a row of sixteen.  The six pairings used - xor 32, xor 16, xor 8 (row_ror:8), xor 7
// (row_half_mirror), xor 2 and xor 1 (quad_perm) - generate all 64 lanes, so after the six steps every lane has met every other one exactly once.
typedef unsigned int lld_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dlo(double x) { return (unsigned)__double2loint(x); }
__device__ __forceinline__ unsigned dhi(double x) { return (unsigned)__double2hiint(x); }
__device__ __forceinline__ double dmk(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
// x + (x of lane ^ 32) in the lanes below 32 and y + (y of lane ^ 32) in the lanes from 32 up: v_permlane32_swap exchanges lanes [32, 64) of its
// first operand with lanes [0, 32) of its second, after which both registers hold one own and one partner value of the kind the lane keeps.
__device__ __forceinline__ double pair32_sum(double x, double y) {
  const lld_u2 l = __builtin_amdgcn_permlane32_swap(dlo(x), dlo(y), false, false), h = __builtin_amdgcn_permlane32_swap(dhi(x), dhi(y), false, false);
  return dmk(l[0], h[0]) + dmk(l[1], h[1]);
}
// the same with lane ^ 16 (v_permlane16_swap: the odd rows of the first operand against the even rows of the second): x in the even rows, y in the odd ones
__device__ __forceinline__ double pair16_sum(double x, double y) {
  const lld_u2 l = __builtin_amdgcn_permlane16_swap(dlo(x), dlo(y), false, false), h = __builtin_amdgcn_permlane16_swap(dhi(x), dhi(y), false, false);
  return dmk(l[0], h[0]) + dmk(l[1], h[1]);
}
constexpr int kDppRor8 = 0x128, kDppHalfMirror = 0x141, kDppQuadXor2 = 0x4E, kDppQuadXor1 = 0xB1;
// the value of the partner lane under one DPP pairing (all lanes)
template <int kCtrl>
__device__ __forceinline__ double dpp_partner(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)dlo(x), kCtrl, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, (int)dhi(x), kCtrl, 0xf, 0xf, false);
  return dmk((unsigned)lo, (unsigned)hi);
}
// lanes whose selector bit is clear receive the partner's x, the others the partner's y; the selector is a whole bank of four lanes (bit 8: banks 2, 3;
// bit 4: banks 1, 3), so the two DPP moves write complementary banks of one register - no select
template <int kCtrl, int kBanksClear>
__device__ __forceinline__ double dpp_partner_xy(double x, double y) {
  int lo = __builtin_amdgcn_update_dpp(0, (int)dlo(x), kCtrl, 0xf, kBanksClear, false), hi = __builtin_amdgcn_update_dpp(0, (int)dhi(x), kCtrl, 0xf, kBanksClear, false);
  lo = __builtin_amdgcn_update_dpp(lo, (int)dlo(y), kCtrl, 0xf, 0xf ^ kBanksClear, false); hi = __builtin_amdgcn_update_dpp(hi, (int)dhi(y), kCtrl, 0xf, 0xf ^ kBanksClear, false);
  return dmk((unsigned)lo, (unsigned)hi);
}
// every lane returns the wavefront's total (fixed tree)
__device__ __forceinline__ double wave_sum_all1(double x) {
  x = pair32_sum(x, x);
  x = pair16_sum(x, x);
  x += dpp_partner<kDppRor8>(x);
  x += dpp_partner<kDppHalfMirror>(x);
  x += dpp_partner<kDppQuadXor2>(x);
  x += dpp_partner<kDppQuadXor1>(x);
  return x;
}
template <int N>
__device__ __forceinline__ void wave_sum_all(double* v) {
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = wave_sum_all1(v[i]);
}
// Fixed-tree block sum of one double per lane; every lane returns the total.  Consecutive calls alternate between two LDS
// buffers (`flip`), so ONE barrier per call is enough: a buffer is rewritten only after every lane has passed the barrier of the
// call in between, i.e. after it has read the previous contents.
template <int kWaves>
__device__ __forceinline__ double block_sum1(double x, double* lds /* [2][kWaves] */, int& flip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  x = wave_sum_all1(x);
  double* buf = lds + flip * kWaves;
  flip ^= 1;
  if (lane == 0) buf[wave] = x;
  __syncthreads();
  double s = buf[0];
#pragma unroll
  for (int w = 1; w < kWaves; w++) s += buf[w];
  return s;
}

// Sum of 28 values per lane over the wavefront with a halving butterfly: at every step a lane keeps one half of its values and hands the other half to
// its partner, so 14 + 7 + 4 + 2 + 1 + 1 = 29 exchanges do the work of 28 x 6.  The totals land in dst[0..27] (LDS).  Fixed tree, hence deterministic.
// Which lane ends with which total: index = 14 [lane & 32] + 7 [lane & 16] + 4 [lane & 8] + 2 [lane & 4] + [lane & 2], the even lanes store.
__device__ __forceinline__ void wave_sum28(const double* v, double* dst) {
  const int lane = threadIdx.x & 63;
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
  double s[14], t[8], u[4], w[2];
#pragma unroll
  for (int i = 0; i < 14; i++) s[i] = pair32_sum(v[i], v[i + 14]);
#pragma unroll
  for (int i = 0; i < 7; i++) t[i] = pair16_sum(s[i], s[i + 7]);
  t[7] = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) u[i] = (b3 ? t[i + 4] : t[i]) + dpp_partner_xy<kDppRor8, 0x3>(t[i], t[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; i++) w[i] = (b2 ? u[i + 2] : u[i]) + dpp_partner_xy<kDppHalfMirror, 0x5>(u[i], u[i + 2]);
  double r = (b1 ? w[1] : w[0]) + dpp_partner<kDppQuadXor2>(b1 ? w[0] : w[1]);
  r += dpp_partner<kDppQuadXor1>(r);
  const int sub = (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
  if (!(lane & 1) && sub < 7) dst[(b5 ? 14 : 0) + (b4 ? 7 : 0) + sub] = r;
}

// pose_oplus of lld_device_math.h with the two quaternion normalisations done by one reciprocal each instead of four divisions:
// the solve runs on ONE lane while the workgroup waits, and fp64 divisions (~30 dependent instructions each) were most of it.
// Differs from the division form by at most one rounding per component, far inside the 1e-5 parity band.
__device__ __forceinline__ void pose_normalize_rcp(Pose& p) {
  if (p.q.w < 0) { p.q.x = -p.q.x; p.q.y = -p.q.y; p.q.z = -p.q.z; p.q.w = -p.q.w; }
  const double inv = 1.0 / sqrt(p.q.x * p.q.x + p.q.y * p.q.y + p.q.z * p.q.z + p.q.w * p.q.w);
  p.q.x *= inv; p.q.y *= inv; p.q.z *= inv; p.q.w *= inv;
}
__device__ __forceinline__ Pose pose_oplus_rcp(const Pose& T, const double* u) {
  const Vec3 w = vec3(u[0], u[1], u[2]), v = vec3(u[3], u[4], u[5]);
  const double ww = dot(w, w);
  const double theta = sqrt(ww);
  double a, b, c;
  if (theta < 0.00001) { a = 1.0; b = 1.0; c = 1.0; }
  else {
    double s, co; sincos(theta, &s, &co);
    const double it = 1.0 / theta, it2 = it * it;
    a = s * it; b = (1 - co) * it2; c = (theta - s) * it2 * it;
  }
  Mat3 R;
  const double W2[3][3] = {{w.x * w.x - ww, w.x * w.y, w.x * w.z}, {w.y * w.x, w.y * w.y - ww, w.y * w.z}, {w.z * w.x, w.z * w.y, w.z * w.z - ww}};
  const double W1[3][3] = {{0, -w.z, w.y}, {w.z, 0, -w.x}, {-w.y, w.x, 0}};
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) R.m[i][j] = ((i == j ? 1.0 : 0.0) + a * W1[i][j]) + b * W2[i][j];
  }
  const Vec3 wv = cross(w, v);
  const Vec3 wwv = cross(w, wv);
  Pose E;
  E.q = quat_from_rotation(R);
  E.t = v + b * wv + c * wwv;
  pose_normalize_rcp(E);
  Pose r;
  r.t = E.t + quat_rotate(E.q, T.t);
  r.q = quat_mul(E.q, T.q);
  pose_normalize_rcp(r);
  return r;
}

// (Round 6 tried exp(omega) as a quaternion directly, with series for sin / cos below 1/4 rad - 10 us off a frame's kernel.  Not kept: the
// series are MORE accurate than the reference's closed forms (1 - cos t) / t^2, (t - sin t) / t^3, whose cancellation noise decides whether
// a trial at convergence "improves" chi2: a monocular frame that the oracle walks in 20 iterations / 52 trials took 20 / 20 with the accurate
// map and 17 / 44 with this one.  Same pose to 5e-13 either way; the reference's arithmetic stays.  profiles/NOTES_r06.md.)
// Dense LDL^T of the 6x6 system (LinearSolverDense, solvers/linear_solver_dense.h:65-113): fails unless all pivots > 0.
__device__ __forceinline__ bool solve6(const double* Hu /*21 upper, row-major packed*/, double lambda, const double* b, double* x) {
  double A[6][6];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
      for (int j = i; j < 6; j++) { A[j][i] = Hu[k]; k++; }         // lower triangle; becomes L below the diagonal, D on it
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) A[i][i] += lambda;
  bool ok = true;
  double inv[6];                                                    // one division per pivot (the columns and y are scaled by it)
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double d = A[j][j];
#pragma unroll
    for (int p = 0; p < j; p++) d -= A[j][p] * A[j][p] * A[p][p];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    inv[j] = rcp_nr(d);                                              // (v_rcp_f64 + two Newton steps: this chain runs on one wavefront while the workgroup waits)
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double s = A[i][j];
#pragma unroll
      for (int p = 0; p < j; p++) s -= A[i][p] * A[j][p] * A[p][p];
      A[i][j] = s * inv[j];
    }
    A[j][j] = d;
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double s = b[i];
#pragma unroll
    for (int p = 0; p < i; p++) s -= A[i][p] * y[p];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] *= inv[i];
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
#pragma unroll
    for (int p = i + 1; p < 6; p++) s -= A[p][i] * x[p];
    x[i] = s;
  }
  return ok;
}

// BaseUnaryEdge::constructQuadraticForm (core/base_unary_edge.hpp:42-72): b -= w A^T (s e), H += A^T (w s) A.  D is a
// compile-time row count so that J stays in registers (a monocular point passes a zero third row and e[2] = 0).
template <int D>
__device__ __forceinline__ void accum_unary(const double* J, const double* e, double s, double w, double* H, double* b) {
  const double ws = w * s;
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    double br = 0;
#pragma unroll
    for (int i = 0; i < D; i++) br += J[i * 6 + r] * e[i];
    b[r] -= ws * br;
#pragma unroll
    for (int c = r; c < 6; c++) {
      double h = 0;
#pragma unroll
      for (int i = 0; i < D; i++) h += J[i * 6 + r] * J[i * 6 + c];
      H[k++] += ws * h;
    }
  }
}

// A pose as rotation matrix + translation: the sweeps map two or three points per lane with it (9 multiply-adds each instead of the 24
// operations of Eigen's q * v; the matrix is formed once per sweep).  Differs from the quaternion form by rounding, like the closed forms
// of the BA kernels.
struct PoseRt { Mat3 R; Vec3 t; };
__device__ __forceinline__ PoseRt pose_rt(const Pose& p) { PoseRt r; r.R = quat_rotation(p.q); r.t = p.t; return r; }

// One staged observation record: (u, v, uR, invSigma2) of a point, (xs, ys, xe, ye) of a line edge - floats when every observation of the
// batch is a widened float (PoseArrays::f32, the host checks), the caller's doubles otherwise.
template <typename OT> struct alignas(16) Obs4 { OT a, b, c, d; };

// LDS bytes of one staged frame: world points (3) + per-line end points (6) + per-edge information (1) as doubles, the observation
// records, the points' float chi2 and the edges' line index, two flag byte arrays.
__host__ __device__ inline size_t pose_lds_bytes(int n_pt, int n_ln, int n_le, bool f32) {
  size_t b = 8 * (3 * (size_t)n_pt + 6 * (size_t)n_ln + (size_t)n_le);
  b = (b + 15) & ~(size_t)15;
  b += (f32 ? 16 : 32) * ((size_t)n_pt + (size_t)n_le);
  b += 4 * ((size_t)n_pt + (size_t)n_le);
  b += (((size_t)n_pt + 15) & ~(size_t)15) + (((size_t)n_le + 15) & ~(size_t)15) + 32;
  return b;
}

// kLds: the frame is staged in LDS (OT = type of its observation records); otherwise the HBM arrays themselves are swept (OT unused).
// kThreads: 512 (one frame per CU) or 256 (two).  launch bound: two wavefronts per SIMD either way (<= 256 registers).
template <bool kLds, typename OT, int kThreads>
__global__ __launch_bounds__(kThreads, 2) void pose_opt_kernel(const PoseFrameDev* __restrict__ frames, PoseArrays a, PoseOut* __restrict__ out,
                                                                  int n_rounds, int its_per_round, int max_trials) {
  constexpr int kWaves = kThreads / 64;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  __shared__ double red[kWaves * 28];           // per-wavefront partials of the normal equations
  __shared__ double tot[28];                    // H (21 upper), b (6), robust chi2
  constexpr int kCand = 16;                     // damped solutions computed at once (see the trial loop)
  __shared__ double sol[kCand * 10];            // per candidate: trial pose (7), scale, solver ok, lambda
  __shared__ double red1[2 * kWaves];
  int flip = 0;
#ifdef LLD_EXPERIMENTS
  long long po_acc[PO_N_STAMPS];
  for (int i = 0; i < PO_N_STAMPS; i++) po_acc[i] = 0;
  const long long po_begin = (long long)__builtin_amdgcn_s_memtime();
#endif
  LLD_PO_T0();
  const PoseFrameDev& F = frames[blockIdx.x];
  const CamK cam = F.cam;
  const int tid = threadIdx.x;
  const Pose T0 = pose_load(F.T0);
  Pose T = T0;
  const int n_pt = F.n_pt, n_le = F.n_le, n_ln = F.n_ln;
  const int po = F.pt_off, lo = F.le_off, no = F.ln_off;

  // ---- the frame's view: LDS copies (staged once) or the HBM arrays themselves
  const double *PX, *PY, *PZ;                   // world points
  const double* LX[6];                          // line end points: per LINE in LDS, per EDGE in HBM
  const double* linfo;                          // information of the edge
  const Obs4<OT>* pobs = nullptr; const Obs4<OT>* lseg = nullptr;
  float* pchi_f = nullptr; double* pchi_d = nullptr;
  uint8_t *pfl, *lfl; const int* lline;
  if constexpr (kLds) {
    double* d = dyn;
    double* px = d; d += n_pt; double* py = d; d += n_pt; double* pz = d; d += n_pt;
    double* lx[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { lx[k] = d; d += n_ln; }
    double* li_ = d; d += n_le;
    Obs4<OT>* po4 = reinterpret_cast<Obs4<OT>*>(reinterpret_cast<char*>(dyn) + ((8 * (3 * (size_t)n_pt + 6 * (size_t)n_ln + (size_t)n_le) + 15) & ~(size_t)15));
    Obs4<OT>* ls4 = po4 + n_pt;
    float* pc = reinterpret_cast<float*>(ls4 + n_le);
    int* ll = reinterpret_cast<int*>(pc + n_pt);
    pfl = reinterpret_cast<uint8_t*>(ll + n_le);
    lfl = pfl + ((n_pt + 15) & ~15);
    for (int i = tid; i < n_pt; i += kThreads) {
      px[i] = a.pt[0][po + i]; py[i] = a.pt[1][po + i]; pz[i] = a.pt[2][po + i];
      Obs4<OT> o; o.a = (OT)a.pt[3][po + i]; o.b = (OT)a.pt[4][po + i]; o.c = (OT)a.pt[5][po + i]; o.d = (OT)a.pt[6][po + i];
      po4[i] = o;
    }
    for (int i = tid; i < n_le; i += kThreads) {
      const int l = a.le_line[lo + i];
      ll[i] = l;
#pragma unroll
      for (int k = 0; k < 6; k++) lx[k][l] = a.le[k][lo + i];        // both edges of a stereo line carry the same end points: equal values, either write
      Obs4<OT> o; o.a = (OT)a.le[6][lo + i]; o.b = (OT)a.le[7][lo + i]; o.c = (OT)a.le[8][lo + i]; o.d = (OT)a.le[9][lo + i];
      ls4[i] = o;
      li_[i] = a.le[10][lo + i];
    }
    PX = px; PY = py; PZ = pz;
#pragma unroll
    for (int k = 0; k < 6; k++) LX[k] = lx[k];
    linfo = li_; pobs = po4; lseg = ls4; pchi_f = pc; lline = ll;
  } else {
    PX = a.pt[0] + po; PY = a.pt[1] + po; PZ = a.pt[2] + po;
#pragma unroll
    for (int k = 0; k < 6; k++) LX[k] = a.le[k] + lo;
    linfo = a.le[10] + lo;
    pchi_d = a.pt_chi2 + po; pfl = a.pt_fl + po; lfl = a.le_fl + lo; lline = a.le_line + lo;
  }
  // every solve starts from: level 0, Huber kernels on, nothing flagged
  for (int i = tid; i < n_pt; i += kThreads) { pfl[i] = PF_ROBUST; if constexpr (kLds) pchi_f[i] = 0.f; else pchi_d[i] = 0.0; }
  for (int i = tid; i < n_le; i += kThreads) lfl[i] = a.le_fl0[lo + i] | LF_ROBUST;
  for (int i = tid; i < n_ln; i += kThreads) a.ln_outlier[no + i] = 0;       // (Frame::mvbOutlierLines itself is NOT reset by the reference: a.fl_outlier keeps what it holds)
  __syncthreads();
  LLD_PO_LAP(PO_STAGE);

  int lm_iterations = 0, lm_trials = 0;
  double last_chi = 0.0;
  int nBad_pts = 0;
  const bool enough = n_pt >= 3;                     // if(nInitialCorrespondences<3) return 0;  (Optimizer.cc:809-810)

  auto pt_obs = [&](int i, double& u, double& v, double& ur, double& s) {
    if constexpr (kLds) { const Obs4<OT> o = pobs[i]; u = (double)o.a; v = (double)o.b; ur = (double)o.c; s = (double)o.d; }
    else { u = a.pt[3][po + i]; v = a.pt[4][po + i]; ur = a.pt[5][po + i]; s = a.pt[6][po + i]; }
  };
  auto put_chi = [&](int i, double chi) { if constexpr (kLds) pchi_f[i] = (float)chi; else pchi_d[i] = chi; };
  // residuals of one edge at pose P (operands are loaded before the level test so the loads of a lane's edges overlap)
  auto point_eval = [&](const PoseRt& Pq, int i, Vec3& Xc, double* e, double& s, bool& stereo) {
    Xc = mat_mul(Pq.R, vec3(PX[i], PY[i], PZ[i])) + Pq.t;
    double u, v, urv; pt_obs(i, u, v, urv, s);
    stereo = !(urv < 0);
    point_residual(cam, Xc, u, v, urv, stereo, false, e);
    return e[0] * (s * e[0]) + e[1] * (s * e[1]) + (stereo ? e[2] * (s * e[2]) : 0.0);
  };
  // (fl: the edge's flag byte - the right-image edge of a stereo line, LAST and STEREO, projects with the baseline)
  auto line_eval = [&](const PoseRt& Pq, int i, uint8_t fl, Vec3& X1m, Vec3& X2m, double* e, double& s, LineAdj* adj) {
    const int l = kLds ? lline[i] : i;
    X1m = mat_mul(Pq.R, vec3(LX[0][l], LX[1][l], LX[2][l])) + Pq.t;
    X2m = mat_mul(Pq.R, vec3(LX[3][l], LX[4][l], LX[5][l])) + Pq.t;
    double xs, ys, xe, ye, bx;
    if constexpr (kLds) {
      const Obs4<OT> o = lseg[i]; xs = (double)o.a; ys = (double)o.b; xe = (double)o.c; ye = (double)o.d;
      bx = ((fl & (LF_LAST | LF_STEREO)) == (LF_LAST | LF_STEREO)) ? cam.bx_right : 0.0;
    } else { xs = a.le[6][lo + i]; ys = a.le[7][lo + i]; xe = a.le[8][lo + i]; ye = a.le[9][lo + i]; bx = a.le[11][lo + i]; }
    line_residual(cam, bx, X1m, X2m, xs, ys, xe, ye, e, adj);
    s = linfo[i];
    return e[0] * (s * e[0]) + e[1] * (s * e[1]);
  };
  // linearisation sweep: chi2 of every active edge + the lane's share of the normal equations
  auto sweep_build = [&](const Pose& Pose_q, double* acc /*28: H21,b6,chi*/) {
    const PoseRt Pq = pose_rt(Pose_q);
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = 0.0;
    for (int i = tid; i < n_pt; i += kThreads) {
      const uint8_t fl = pfl[i];
      if (fl & PF_LEVEL) continue;
      Vec3 Xc; double e[3], s; bool stereo;
      const double chi = point_eval(Pq, i, Xc, e, s, stereo);
      put_chi(i, chi);
      double w = 1.0, rho0 = chi;
      if (fl & PF_ROBUST) rho0 = huber(chi, stereo ? F.delta_stereo : F.delta_mono, &w);
      acc[27] += rho0;
      double J[18]; point_jac_pose(cam, Xc, stereo, J); accum_unary<3>(J, e, s, w, acc, acc + 21);
    }
    for (int i = tid; i < n_le; i += kThreads) {
      const uint8_t fl = lfl[i];
      if (fl & LF_LEVEL) continue;
      Vec3 X1m, X2m; double e[2], s; LineAdj adj;
      const double chi = line_eval(Pq, i, fl, X1m, X2m, e, s, &adj);
      double w = 1.0, rho0 = chi;
      if (fl & LF_ROBUST) rho0 = huber(chi, (fl & LF_STEREO) ? F.delta_ln_stereo : F.delta_ln_mono, &w);
      acc[27] += rho0;
      double J[12]; line_jac_pose(adj, X1m, X2m, J); accum_unary<2>(J, e, s, w, acc, acc + 21);
    }
  };
  // trial sweep: computeActiveErrors + activeRobustChi2 at the trial pose
  auto sweep_chi = [&](const Pose& Pose_q) {
    const PoseRt Pq = pose_rt(Pose_q);
    double c = 0.0;
    for (int i = tid; i < n_pt; i += kThreads) {
      const uint8_t fl = pfl[i];
      if (fl & PF_LEVEL) continue;
      Vec3 Xc; double e[3], s; bool stereo;
      const double chi = point_eval(Pq, i, Xc, e, s, stereo);
      put_chi(i, chi);
      double w, rho0 = chi;
      if (fl & PF_ROBUST) rho0 = huber(chi, stereo ? F.delta_stereo : F.delta_mono, &w);
      c += rho0;
    }
    for (int i = tid; i < n_le; i += kThreads) {
      const uint8_t fl = lfl[i];
      if (fl & LF_LEVEL) continue;
      Vec3 X1m, X2m; double e[2], s;
      const double chi = line_eval(Pq, i, fl, X1m, X2m, e, s, nullptr);
      double w, rho0 = chi;
      if (fl & LF_ROBUST) rho0 = huber(chi, (fl & LF_STEREO) ? F.delta_ln_stereo : F.delta_ln_mono, &w);
      c += rho0;
    }
    return c;
  };

  double lambda = -1.0, ni = 2.0;
  int nBadLM = 0;
  if (enough) {
    for (int round = 0; round < n_rounds; round++) {
      T = T0;                                                        // vSE3->setEstimate(toSE3Quat(pFrame->mTcw))  (:823)
      // initializeOptimization(0): active = level-0 edges.  optimize() returns -1 when nothing is active.
      double cnt = 0.0;
      for (int i = tid; i < n_pt; i += kThreads) cnt += (pfl[i] & PF_LEVEL) ? 0.0 : 1.0;
      for (int i = tid; i < n_le; i += kThreads) cnt += (lfl[i] & LF_LEVEL) ? 0.0 : 1.0;
      cnt = block_sum1<kWaves>(cnt, red1, flip);
      LLD_PO_LAP(PO_ROUND_HEAD);
      if (cnt > 0.5) {
        bool ok = true;
        for (int it = 0; it < its_per_round && ok; it++) {
          // ---- OptimizationAlgorithmLevenberg::solve(it)  (optimization_algorithm_levenberg.cpp:61-164)
          {
            double acc[28];
            sweep_build(T, acc);
            LLD_PO_LAP(PO_BUILD);
            wave_sum28(acc, red + (tid >> 6) * 28);               // red / tot were last read before the previous trial's barrier
            __syncthreads();
            if (tid < 28) {
              double sv = red[tid];
#pragma unroll
              for (int w = 1; w < kWaves; w++) sv += red[w * 28 + tid];
              tot[tid] = sv;
            }
          }
          // lanes 0..27 belong to wavefront 0, which runs in lockstep: its lane 0 may read `tot` without a workgroup barrier; the
          // other wavefronts read tot[27] only after the barrier that follows the first solve
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          LLD_PO_LAP(PO_REDUCE);
          bool first_trial = true;
          double currentChi = 0.0, iniChi = 0.0;
          double rho = 0.0; int q = 0;
          // The candidates of this iteration.  A rejected trial changes lambda alone (lambda *= ni, ni *= 2; ni = 2 at the first trial of every
          // iteration): the damped solutions of the NEXT kCand - 1 rejections are known as soon as H and b are, and more than half of a frame's
          // trials are rejected (config PO: 20 iterations, 30 - 49 trials).  Lane j of wavefront 0 therefore solves for lambda after j rejections
          // - the same instruction stream in every lane, one solve's time - and a rejected trial picks up the next candidate instead of waiting
          // for one lane's LDL^T, exponential map and two normalisations again (~900 dependent fp64 instructions).
          int q_base = 0;
          do {
            if (q == q_base) {
              if (tid < kCand) {
                double Hb[27], x[6];
#pragma unroll
                for (int i = 0; i < 27; i++) Hb[i] = tot[i];
                double lam = lambda, nij = ni;
                if (first_trial && it == 0) {                        // computeLambdaInit: 1e-5 * max diagonal, iteration 0 of every optimize()
                  double md = 0.0; int k = 0;
#pragma unroll
                  for (int r = 0; r < 6; r++) { md = fmax(fabs(Hb[k]), md); k += 6 - r; }
                  lam = 1e-5 * md; nij = 2.0;
                }
                // lambda after `tid` rejections: lam * prod_{k < tid} (nij 2^k).  nij is 2 whenever candidates are formed (an accepted trial and
                // computeLambdaInit both reset it; a round that ran out of trials ends), so the products are exact powers of two - in any order
                if (nij == 2.0) lam = ldexp(lam, tid * (tid + 1) / 2);
                else for (int j = 0; j < tid; j++) { lam *= nij; nij *= 2; }
                const bool ok2 = solve6(Hb, lam, Hb + 21, x);
                const Pose Tn = pose_oplus_rcp(T, x);
                double scale = 0.0;
                for (int j = 0; j < 6; j++) scale += x[j] * (lam * x[j] + Hb[21 + j]);
                scale += 1e-3;
                double* so = sol + tid * 10;
                pose_store(Tn, so);
                so[7] = scale; so[8] = ok2 ? 1.0 : 0.0; so[9] = lam;
              }
              __syncthreads();
              LLD_PO_LAP(PO_SOLVE); LLD_PO_COUNT(PO_N_SOLVES);
            }
            const double* so = sol + (q - q_base) * 10;
            const Pose Tn = pose_load(so);
            const double scale = so[7];
            const bool ok2 = so[8] != 0.0;
            if (first_trial) {
              currentChi = tot[27]; iniChi = currentChi;
              if (it == 0) { lambda = so[9]; ni = 2.0; nBadLM = 0; }
              first_trial = false;
            }
            const double my_chi = sweep_chi(Tn);
            LLD_PO_LAP(PO_TRIAL_SWEEP);
            const double tmp = block_sum1<kWaves>(my_chi, red1, flip);   // its barrier also fences `sol` and `tot` against the next trial
            LLD_PO_LAP(PO_TRIAL_SUM); LLD_PO_COUNT(PO_N_TRIALS);
            const double tempChi = ok2 ? tmp : 1.7976931348623157e308;
            rho = (currentChi - tempChi) / scale;
            if (rho > 0 && isfinite(tempChi)) {
              double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
              alpha = fmin(alpha, 2. / 3.);
              lambda *= fmax(1. / 3., alpha);
              ni = 2; currentChi = tempChi; T = Tn;
            } else { lambda *= ni; ni *= 2; }
            q++; lm_trials++;
            if (q - q_base == kCand) q_base = q;                       // (more than kCand rejections in a row: the next block of candidates)
          } while (rho < 0 && q < max_trials);
          last_chi = currentChi;
          lm_iterations++; LLD_PO_COUNT(PO_N_ITS);
          if (q == max_trials || rho == 0) ok = false;
          else {
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) ok = false;
          }
        }
      }
      // ---- classification (Optimizer.cc:827-913)
      __syncthreads();
      LLD_PO_LAP(PO_ROUND_HEAD);
      const PoseRt Trt = pose_rt(T);
      double nb = 0.0;
      for (int i = tid; i < n_pt; i += kThreads) {
        uint8_t fl = pfl[i];
        float chif;
        if constexpr (kLds) chif = pchi_f[i]; else chif = (float)pchi_d[i];
        double u, v, urv, s0; pt_obs(i, u, v, urv, s0);
        bool stereo = !(urv < 0);
        if (fl & PF_OUTLIER) {                                       // if(pFrame->mvbOutlier[idx]) e->computeError();
          Vec3 Xc; double e[3], s;
          const double chi = point_eval(Trt, i, Xc, e, s, stereo);
          put_chi(i, chi);
          chif = (float)chi;
        }
        const bool bad = chif > (stereo ? 7.815f : 5.991f);
        fl = (uint8_t)((fl & PF_ROBUST) | (bad ? (PF_LEVEL | PF_OUTLIER) : 0));
        if (round == 2) fl &= (uint8_t)~PF_ROBUST;
        pfl[i] = fl;
        nb += bad ? 1.0 : 0.0;
      }
      nb = block_sum1<kWaves>(nb, red1, flip);
      nBad_pts = (int)(nb + 0.5);
      if (n_pt + n_le < 10) break;                                   // if(optimizer.edges().size()<10) break;
      // vnStereoLines is filled per EDGE but indexed by the LINE's index in the frame (Optimizer.cc:643-648 vs :898): which entry
      // that is was resolved on the host (LF_THR_STEREO, pose_pack)
      for (int i = tid; i < n_le; i += kThreads) {
        uint8_t fl = lfl[i];
        Vec3 X1m, X2m; double e[2], s;
        const double chi = line_eval(Trt, i, fl, X1m, X2m, e, s, nullptr);
        const float chif = (float)chi;
        const int idx = lline[i];
        const bool st = (fl & LF_THR_STEREO) != 0;
        const double thr = st ? F.thr_ln_stereo : F.thr_ln_mono;
        const bool bad = (double)chif > thr;
        fl = (uint8_t)((fl & ~LF_LEVEL) | (bad ? LF_LEVEL : 0));
        if (fl & LF_LAST) { a.ln_outlier[no + idx] = bad; if (a.ln_fi) a.fl_outlier[a.ln_fi[idx]] = bad; }   // the right-image edge overwrites the left one
        if (round == 2) fl &= (uint8_t)~LF_ROBUST;
        lfl[i] = fl;
      }
      __syncthreads();
      LLD_PO_LAP(PO_CLASSIFY);
    }
  }
  for (int i = tid; i < n_pt; i += kThreads) {
    const uint8_t bad = (pfl[i] & PF_OUTLIER) ? 1 : 0;
    a.pt_outlier[po + i] = bad;
    if (a.pt_kp) a.kp_outlier[a.pt_kp[i]] = bad;
  }
  if (tid == 0) {
    PoseOut& o = out[blockIdx.x];
    // PoseOptimization returns before it touches the frame when it has fewer than three points (Optimizer.cc:809-810): the pose stays
    pose_store(T, o.qt);
    o.chi2 = last_chi; o.n_inliers = enough ? n_pt - nBad_pts : 0; o.lm_iterations = lm_iterations; o.lm_trials = lm_trials; o.pad = 0;
#ifdef LLD_EXPERIMENTS
    if (a.stamps && blockIdx.x == 0) {
      po_acc[PO_TOTAL] = (long long)__builtin_amdgcn_s_memtime() - po_begin;
      for (int i = 0; i < PO_N_STAMPS; i++) a.stamps[i] = po_acc[i];
    }
#endif
    if (a.track_out) {
      pose_store(T, a.track_out); a.track_out[7] = last_chi;
      int* ti = reinterpret_cast<int*>(a.track_out + 8);
      ti[0] = o.n_inliers; ti[1] = lm_iterations; ti[2] = lm_trials; ti[3] = n_pt + n_le; ti[4] = n_pt; ti[5] = n_le;
    }
  }
}

}  // namespace

// Byte layout of a batch image: [input region | output region | HBM-mode working arrays], identical on host and device.
struct PoseLayout {
  size_t NP = 1, NE = 1, NL = 1;
  size_t frames = 0, pt = 0, le = 0, le_line = 0, le_fl0 = 0, in_bytes = 0;
  size_t out = 0, pt_outlier = 0, ln_outlier = 0, out_bytes = 0;           // offsets from the start of the image
  size_t pt_chi2 = 0, le_chi2 = 0, pt_fl = 0, le_fl = 0, total = 0;
};
static PoseLayout pose_layout(int n_frames, size_t np, size_t ne, size_t nl, bool hbm_state) {
  PoseLayout Y; Y.NP = np + 1; Y.NE = ne + 1; Y.NL = nl + 1;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += lld_slab::pad(bytes); return at; };
  Y.frames = take(sizeof(PoseFrameDev) * n_frames);
  Y.pt = take(7 * Y.NP * 8); Y.le = take(12 * Y.NE * 8); Y.le_line = take(Y.NE * 4); Y.le_fl0 = take(Y.NE);
  Y.in_bytes = o;
  Y.out = take(sizeof(PoseOut) * n_frames); Y.pt_outlier = take(Y.NP); Y.ln_outlier = take(Y.NL);
  Y.out_bytes = o - Y.in_bytes;
  if (hbm_state) { Y.pt_chi2 = take(Y.NP * 8); Y.le_chi2 = take(Y.NE * 8); Y.pt_fl = take(Y.NP); Y.le_fl = take(Y.NE); }
  Y.total = o;
  return Y;
}

// How a batch runs: observation records as floats (every observation is a widened float), frames staged in LDS, lanes per frame.
struct PoseMode { bool f32 = true, use_lds = true; int threads = kPoseThreadsMax; size_t lds_bytes = 0; };
struct PoseCounts { size_t np = 0, ne = 0, nl = 0; bool f32 = true; int max_pt = 0, max_ln = 0, max_le = 0; size_t lds_f32 = 0, lds_f64 = 0; };
static inline bool pose_is_float(double x) { return (double)(float)x == x; }   // (NaN: false; a float's infinities and subnormals convert back exactly)
static int pose_count(int n_frames, const lld_pose_problem* frames, PoseCounts* C) {
  for (int f = 0; f < n_frames; f++) {
    const lld_pose_problem& P = frames[f];
    if (P.n_points < 0 || P.n_lines < 0) return LLD_ERR_INVALID;
    if ((P.n_points > 0 && (!P.pt_xw || !P.pt_uvr || !P.pt_inv_sigma2)) ||
        (P.n_lines > 0 && (!P.ln_x0 || !P.ln_dir || !P.ln_left || !P.ln_right || !P.ln_octave))) return LLD_ERR_INVALID;
    size_t ne = 0;
    bool fl = C->f32;
    for (int l = 0; l < P.n_lines; l++) {
      const bool hr = !(P.ln_right[4 * l] < 0);
      ne += hr ? 2 : 1;
      if (fl) {
        for (int k = 0; k < 4; k++) fl = fl && pose_is_float(P.ln_left[4 * l + k]) && (!hr || pose_is_float(P.ln_right[4 * l + k]));
      }
    }
    if (fl) {
      for (int i = 0; i < P.n_points && fl; i++)
        fl = pose_is_float(P.pt_uvr[3 * i]) && pose_is_float(P.pt_uvr[3 * i + 1]) && pose_is_float(P.pt_uvr[3 * i + 2]) && pose_is_float(P.pt_inv_sigma2[i]);
    }
    C->f32 = fl;
    C->np += P.n_points; C->ne += ne; C->nl += P.n_lines;
    C->lds_f32 = std::max(C->lds_f32, pose_lds_bytes(P.n_points, P.n_lines, (int)ne, true));
    C->lds_f64 = std::max(C->lds_f64, pose_lds_bytes(P.n_points, P.n_lines, (int)ne, false));
    if (C->np > 0x3fffffff || C->ne > 0x3fffffff) return LLD_ERR_UNSUPPORTED;
  }
  return LLD_OK;
}
// Two frames per CU (256 lanes each) when there are more frames than CUs and two staged frames fit the CU's LDS; otherwise one
// frame per CU on 512 lanes.  The two forms sum a frame's edges in different (each fixed) orders: results agree to rounding.
static PoseMode pose_mode(const lld_ctx* ctx, int n_frames, const PoseCounts& C) {
  PoseMode M;
  M.f32 = C.f32;
  const size_t lds = C.f32 ? C.lds_f32 : C.lds_f64;
  M.use_lds = lds <= kPoseLdsBudget;
  M.lds_bytes = M.use_lds ? lds : 0;
  M.threads = (M.use_lds && lds <= kPoseLdsBudgetPair && n_frames > ctx->n_cu) ? 256 : kPoseThreadsMax;
  return M;
}

// Host-side expansion (AddLineMinOnlyPose): one left edge per line and a right edge when the line has a stereo match.
static void pose_pack(int n_frames, const lld_pose_problem* frames, double gamma, const PoseLayout& Y, char* img, PoseFrameDev* hf) {
  const float dMono = (float)std::sqrt(5.991), dStereo = (float)std::sqrt(7.815);
  float dLnS = dStereo, dLnM = dMono;
  dLnS *= gamma; dLnM *= gamma;                                      // float *= double (Optimizer.cc:706-707)
  double* pt = reinterpret_cast<double*>(img + Y.pt);
  double* le = reinterpret_cast<double*>(img + Y.le);
  int* le_line = reinterpret_cast<int*>(img + Y.le_line);
  uint8_t* le_fl0 = reinterpret_cast<uint8_t*>(img + Y.le_fl0);
  size_t ip = 0, ie = 0, il = 0;
  for (int f = 0; f < n_frames; f++) {
    const lld_pose_problem& P = frames[f];
    PoseFrameDev& F = hf[f];
    F.cam = lld::make_camk(P.cam);
    std::memcpy(F.T0, P.pose_qt, sizeof F.T0);
    F.pt_off = (int)ip; F.n_pt = P.n_points;
    F.le_off = (int)ie; F.ln_off = (int)il; F.n_ln = P.n_lines;
    F.delta_mono = (double)dMono; F.delta_stereo = (double)dStereo;
    F.delta_ln_stereo = (double)dLnS; F.delta_ln_mono = (double)dLnM;
    F.thr_ln_stereo = (double)(dLnS * dLnS); F.thr_ln_mono = (double)(dLnM * dLnM);
    for (int i = 0; i < P.n_points; i++, ip++) {
      pt[0 * Y.NP + ip] = P.pt_xw[3 * i]; pt[1 * Y.NP + ip] = P.pt_xw[3 * i + 1]; pt[2 * Y.NP + ip] = P.pt_xw[3 * i + 2];
      pt[3 * Y.NP + ip] = P.pt_uvr[3 * i]; pt[4 * Y.NP + ip] = P.pt_uvr[3 * i + 1]; pt[5 * Y.NP + ip] = P.pt_uvr[3 * i + 2];
      pt[6 * Y.NP + ip] = P.pt_inv_sigma2[i];
    }
    // vnStereoLines of this frame: one entry per edge, in the order AddLineMinOnlyPose adds them (Optimizer.cc:643-648)
    std::vector<uint8_t> vnStereoLines;
    for (int l = 0; l < P.n_lines; l++) { const bool hr = !(P.ln_right[4 * l] < 0); vnStereoLines.push_back(hr); if (hr) vnStereoLines.push_back(1); }
    for (int l = 0; l < P.n_lines; l++) {
      const double* Lk = P.ln_left + 4 * l; const double* Rk = P.ln_right + 4 * l;
      const bool hr = !(Rk[0] < 0);
      // ... and the entry the classification reads for this line: vnStereoLines[idx], idx = vnIndexLines[.] = the line's index in the frame (:893-898)
      const long long fi = P.ln_frame_index ? (long long)P.ln_frame_index[l] : (long long)l;
      const bool thr_stereo = (fi >= 0 && fi < (long long)vnStereoLines.size()) ? vnStereoLines[(size_t)fi] != 0 : true;
      for (int si = 0; si < 2; si++) {
        if (si == 1 && !hr) continue;
        const double* kl = si == 0 ? Lk : Rk;
        le[0 * Y.NE + ie] = P.ln_x0[3 * l]; le[1 * Y.NE + ie] = P.ln_x0[3 * l + 1]; le[2 * Y.NE + ie] = P.ln_x0[3 * l + 2];
        le[3 * Y.NE + ie] = P.ln_x0[3 * l] + P.ln_dir[3 * l]; le[4 * Y.NE + ie] = P.ln_x0[3 * l + 1] + P.ln_dir[3 * l + 1];
        le[5 * Y.NE + ie] = P.ln_x0[3 * l + 2] + P.ln_dir[3 * l + 2];
        le[6 * Y.NE + ie] = kl[0]; le[7 * Y.NE + ie] = kl[1]; le[8 * Y.NE + ie] = kl[2]; le[9 * Y.NE + ie] = kl[3];
        le[10 * Y.NE + ie] = lld::line_info(gamma, P.ln_octave[2 * l + si]);
        le[11 * Y.NE + ie] = si == 1 ? F.cam.bx_right : 0.0;
        le_line[ie] = l;
        le_fl0[ie] = (uint8_t)(((si == 1 || !hr) ? LF_LAST : 0) | (hr ? LF_STEREO : 0) | (thr_stereo ? LF_THR_STEREO : 0));
        ie++;
      }
    }
    F.n_le = (int)ie - F.le_off;
    il += P.n_lines;
  }
  std::memcpy(img + Y.frames, hf, sizeof(PoseFrameDev) * n_frames);
}

static PoseArrays pose_arrays(char* d, const PoseLayout& Y) {
  PoseArrays A;
  for (int k = 0; k < 7; k++) A.pt[k] = reinterpret_cast<const double*>(d + Y.pt) + k * Y.NP;
  for (int k = 0; k < 12; k++) A.le[k] = reinterpret_cast<const double*>(d + Y.le) + k * Y.NE;
  A.le_line = reinterpret_cast<const int*>(d + Y.le_line);
  A.le_fl0 = reinterpret_cast<const uint8_t*>(d + Y.le_fl0);
  A.pt_chi2 = reinterpret_cast<double*>(d + Y.pt_chi2); A.le_chi2 = reinterpret_cast<double*>(d + Y.le_chi2);
  A.pt_fl = reinterpret_cast<uint8_t*>(d + Y.pt_fl); A.le_fl = reinterpret_cast<uint8_t*>(d + Y.le_fl);
  A.pt_outlier = reinterpret_cast<uint8_t*>(d + Y.pt_outlier); A.ln_outlier = reinterpret_cast<uint8_t*>(d + Y.ln_outlier);
  return A;
}

template <bool kLds, typename OT, int kThreads>
static int pose_launch_as(lld_ctx* ctx, hipStream_t st, int n_frames, size_t lds_bytes, const PoseFrameDev* fr, const PoseArrays& A, PoseOut* po, const lld_pose_params& prm) {
  // the dynamic-LDS ceiling of an instantiation is raised ONCE per context (a driver call on the Tracking thread's latency path otherwise: every lld_pose_opt)
  constexpr unsigned kBit = 1u << ((sizeof(OT) == sizeof(float) ? 1 : 0) | (kThreads == 256 ? 2 : 0));
  if (kLds && !(ctx->pose_lds_raised & kBit)) {
    LLD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pose_opt_kernel<kLds, OT, kThreads>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPoseLdsBudget));
    ctx->pose_lds_raised |= kBit;
  }
  hipLaunchKernelGGL((pose_opt_kernel<kLds, OT, kThreads>), dim3(n_frames), dim3(kThreads), lds_bytes, st, fr, A, po, prm.n_rounds, prm.its_per_round, prm.max_trials);
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}
static int pose_launch(lld_ctx* ctx, int n_frames, char* d_img, const PoseLayout& Y, const PoseMode& M, const lld_pose_params& prm) {
  hipStream_t st = ctx->stream;
  const PoseFrameDev* fr = reinterpret_cast<const PoseFrameDev*>(d_img + Y.frames);
  PoseOut* po = reinterpret_cast<PoseOut*>(d_img + Y.out);
  const PoseArrays A = pose_arrays(d_img, Y);
  if (!M.use_lds) return pose_launch_as<false, double, kPoseThreadsMax>(ctx, st, n_frames, 0, fr, A, po, prm);
  if (M.f32) return M.threads == 256 ? pose_launch_as<true, float, 256>(ctx, st, n_frames, M.lds_bytes, fr, A, po, prm)
                                     : pose_launch_as<true, float, kPoseThreadsMax>(ctx, st, n_frames, M.lds_bytes, fr, A, po, prm);
  return M.threads == 256 ? pose_launch_as<true, double, 256>(ctx, st, n_frames, M.lds_bytes, fr, A, po, prm)
                          : pose_launch_as<true, double, kPoseThreadsMax>(ctx, st, n_frames, M.lds_bytes, fr, A, po, prm);
}

static void pose_fill_result(const PoseLayout& Y, const char* h_out /* start of the output region */, const PoseFrameDev& F, int frame, lld_pose_result* out) {
  const PoseOut& o = reinterpret_cast<const PoseOut*>(h_out + (Y.out - Y.in_bytes))[frame];
  std::memcpy(out->pose_qt, o.qt, sizeof o.qt);
  out->n_inliers = o.n_inliers; out->lm_iterations = o.lm_iterations; out->lm_trials = o.lm_trials; out->reserved = 0; out->chi2 = o.chi2;
  if (out->pt_outlier && F.n_pt) std::memcpy(out->pt_outlier, h_out + (Y.pt_outlier - Y.in_bytes) + F.pt_off, F.n_pt);
  if (out->ln_outlier && F.n_ln) std::memcpy(out->ln_outlier, h_out + (Y.ln_outlier - Y.in_bytes) + F.ln_off, F.n_ln);
}

// ================================================================ the resident frame's PoseOptimization (lld_frame_track_*, lld_track_internal.h)
// The edges are gathered ON THE DEVICE from what the frame holds, in the order Optimizer::PoseOptimization adds them: one point edge per
// keypoint with a MapPoint in keypoint order (stereo iff mvuRight[i] >= 0, src/Optimizer.cc:739-792), then AddLineMinOnlyPose for every
// frame line with a MapLine in line order (:796-804).  One workgroup: an exclusive scan numbers the edges, every lane writes its keypoints'
// / lines' rows of the image pose_opt_kernel reads.  pt_kp / ln_fi map the edges back to the frame for the outlier flags.
namespace {
constexpr int kAsmThreads = 1024;
struct PoseAsmConsts { double delta_mono, delta_stereo, delta_ln_stereo, delta_ln_mono, thr_ln_stereo, thr_ln_mono; };

__device__ __forceinline__ int block_excl_scan(int v, int* lds /*[kAsmThreads/64 + 1]*/, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
  __syncthreads();                                             // (lds may still be read by the previous scan)
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < kAsmThreads / 64; w++) { const int t = lds[w]; if (w < wave) base += t; tot += t; }
  total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(kAsmThreads) void pose_assemble_kernel(lld_track::PoseTrackDev in, PoseAsmConsts dc, char* img, PoseLayout Y, int* pt_kp, int* ln_fi) {
  __shared__ int scan_lds[kAsmThreads / 64 + 1];
  extern __shared__ unsigned char edge_stereo[];               // vnStereoLines: one entry per line EDGE, in the order they are added
  const int tid = threadIdx.x;
  const CamK cam = lld::make_camk(in.cam);
  double* pt = reinterpret_cast<double*>(img + Y.pt);
  double* le = reinterpret_cast<double*>(img + Y.le);
  int* le_line = reinterpret_cast<int*>(img + Y.le_line);
  uint8_t* le_fl0 = reinterpret_cast<uint8_t*>(img + Y.le_fl0);
  // ---- points
  const int per = (in.nt + kAsmThreads - 1) / kAsmThreads, k0 = min(tid * per, in.nt), k1 = min(k0 + per, in.nt);
  int cnt = 0;
  for (int k = k0; k < k1; k++) cnt += in.kp_has[k] ? 1 : 0;
  int n_pt;
  int e = block_excl_scan(cnt, scan_lds, n_pt);
  for (int k = k0; k < k1; k++) {
    if (!in.kp_has[k]) continue;
    const float ur = in.t_uright ? in.t_uright[k] : -1.f;
    pt[0 * Y.NP + e] = (double)in.kp_world[3 * k]; pt[1 * Y.NP + e] = (double)in.kp_world[3 * k + 1]; pt[2 * Y.NP + e] = (double)in.kp_world[3 * k + 2];
    pt[3 * Y.NP + e] = (double)in.t_xy[2 * k]; pt[4 * Y.NP + e] = (double)in.t_xy[2 * k + 1];
    pt[5 * Y.NP + e] = ur < 0.f ? -1.0 : (double)ur;
    pt[6 * Y.NP + e] = (double)in.inv_sigma2[in.t_octave[k]];
    pt_kp[e] = k;
    e++;
  }
  // ---- lines: rank of the line among those with a MapLine, and the index of its first edge
  const int lper = (in.nl + kAsmThreads - 1) / kAsmThreads, l0 = min(tid * lper, in.nl), l1 = min(l0 + lper, in.nl);
  int lc = 0, ec = 0;
  for (int i = l0; i < l1; i++) if (in.ln_has[i]) { lc++; ec += in.ln_match[i] >= 0 ? 2 : 1; }
  int n_ln, n_le;
  int l = block_excl_scan(lc, scan_lds, n_ln);
  int ie = block_excl_scan(ec, scan_lds, n_le);
  {
    int ie2 = ie;
    for (int i = l0; i < l1; i++) if (in.ln_has[i]) { const bool hr = in.ln_match[i] >= 0; edge_stereo[ie2++] = hr; if (hr) edge_stereo[ie2++] = 1; }
  }
  __syncthreads();
  for (int i = l0; i < l1; i++) {
    if (!in.ln_has[i]) continue;
    const int ri = in.ln_match[i];
    const bool hr = ri >= 0;
    // vnStereoLines is filled per EDGE and read with the LINE's index in the frame (Optimizer.cc:643-648 vs :898); beyond the edge count: stereo
    const bool thr_stereo = i < n_le ? edge_stereo[i] != 0 : true;
    const double x0[3] = {in.ln_x0[3 * i], in.ln_x0[3 * i + 1], in.ln_x0[3 * i + 2]}, dr[3] = {in.ln_dir[3 * i], in.ln_dir[3 * i + 1], in.ln_dir[3 * i + 2]};
    for (int si = 0; si < 2; si++) {
      if (si == 1 && !hr) continue;
      const float* kl = si == 0 ? in.ln_left + 4 * i : in.ln_right + 4 * ri;
      const int oct = si == 0 ? in.ln_loct[i] : in.ln_roct[ri];
      le[0 * Y.NE + ie] = x0[0]; le[1 * Y.NE + ie] = x0[1]; le[2 * Y.NE + ie] = x0[2];
      le[3 * Y.NE + ie] = x0[0] + dr[0]; le[4 * Y.NE + ie] = x0[1] + dr[1]; le[5 * Y.NE + ie] = x0[2] + dr[2];
      le[6 * Y.NE + ie] = (double)kl[0]; le[7 * Y.NE + ie] = (double)kl[1]; le[8 * Y.NE + ie] = (double)kl[2]; le[9 * Y.NE + ie] = (double)kl[3];
      le[10 * Y.NE + ie] = lld::line_info(in.gamma, oct);
      le[11 * Y.NE + ie] = si == 1 ? cam.bx_right : 0.0;
      le_line[ie] = l;
      le_fl0[ie] = (uint8_t)(((si == 1 || !hr) ? LF_LAST : 0) | (hr ? LF_STEREO : 0) | (thr_stereo ? LF_THR_STEREO : 0));
      ie++;
    }
    ln_fi[l] = i;
    l++;
  }
  if (tid == 0) {
    PoseFrameDev F;
    F.cam = cam;
    for (int k = 0; k < 7; k++) F.T0[k] = in.pose_qt[k];
    F.pt_off = 0; F.n_pt = n_pt; F.le_off = 0; F.n_le = n_le; F.ln_off = 0; F.n_ln = n_ln;
    F.delta_mono = dc.delta_mono; F.delta_stereo = dc.delta_stereo; F.delta_ln_stereo = dc.delta_ln_stereo; F.delta_ln_mono = dc.delta_ln_mono;
    F.thr_ln_stereo = dc.thr_ln_stereo; F.thr_ln_mono = dc.thr_ln_mono;
    *reinterpret_cast<PoseFrameDev*>(img + Y.frames) = F;
  }
}
}  // namespace

namespace lld_track {
static PoseLayout track_layout(int nt, int nl) { return pose_layout(1, (size_t)nt, 2 * (size_t)nl, (size_t)nl, true); }
size_t pose_track_work_bytes(int nt, int nl) {
  const PoseLayout Y = track_layout(nt, nl);
  return Y.total + lld_slab::pad(4 * ((size_t)nt + 1)) + lld_slab::pad(4 * ((size_t)nl + 1)) + 256;
}
int pose_track_launch(lld_ctx* ctx, hipStream_t st, const PoseTrackDev& in, const lld_pose_params& prm, void* d_work) {
  if (in.nl > 16 * 1024) return LLD_ERR_UNSUPPORTED;           // (the edge flags of the line scan live in LDS)
  char* d_img = static_cast<char*>(d_work);
  const PoseLayout Y = track_layout(in.nt, in.nl);
  int* pt_kp = reinterpret_cast<int*>(d_img + Y.total);
  int* ln_fi = reinterpret_cast<int*>(d_img + Y.total + lld_slab::pad(4 * ((size_t)in.nt + 1)));
  const float dMono = (float)std::sqrt(5.991), dStereo = (float)std::sqrt(7.815);
  float dLnS = dStereo, dLnM = dMono;
  dLnS *= prm.gamma; dLnM *= prm.gamma;                        // float *= double (Optimizer.cc:706-707)
  PoseAsmConsts dc{(double)dMono, (double)dStereo, (double)dLnS, (double)dLnM, (double)(dLnS * dLnS), (double)(dLnM * dLnM)};
  PoseTrackDev inq = in; inq.gamma = prm.gamma;
  hipLaunchKernelGGL(pose_assemble_kernel, dim3(1), dim3(kAsmThreads), (size_t)2 * in.nl + 16, st, inq, dc, d_img, Y, pt_kp, ln_fi);
  PoseArrays A = pose_arrays(d_img, Y);
  A.pt_kp = pt_kp; A.ln_fi = ln_fi; A.kp_outlier = in.kp_outlier; A.fl_outlier = in.ln_outlier; A.track_out = in.pose_out;
  // the edge counts are the device's: the launch reserves the LDS of the largest frame this handle can hold (frame data are floats)
  const size_t lds = pose_lds_bytes(in.nt, in.nl, 2 * in.nl, true);
  const PoseFrameDev* fr = reinterpret_cast<const PoseFrameDev*>(d_img + Y.frames);
  PoseOut* po = reinterpret_cast<PoseOut*>(d_img + Y.out);
  int s;
  if (lds <= kPoseLdsBudget) s = pose_launch_as<true, float, kPoseThreadsMax>(ctx, st, 1, lds, fr, A, po, prm);
  else s = pose_launch_as<false, double, kPoseThreadsMax>(ctx, st, 1, 0, fr, A, po, prm);
  return s;
}
}  // namespace lld_track

struct lld_pose_batch {
  lld_ctx* ctx = nullptr;
  int n_frames = 0;
  lld_pose_params params;
  char* slab = nullptr;
  PoseLayout lay;
  PoseMode mode;
  std::vector<PoseFrameDev> h_frames;
  std::vector<char> h_out;
  bool fetched = false;
};

extern "C" {

int lld_pose_batch_create(lld_ctx* ctx, int n_frames, const lld_pose_problem* frames, const lld_pose_params* params, lld_pose_batch** out) {
  if (!ctx || n_frames <= 0 || !frames || !out) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  PoseCounts C;
  int st = pose_count(n_frames, frames, &C); if (st) return st;
  lld_pose_batch* B = new lld_pose_batch();
  B->ctx = ctx; B->n_frames = n_frames;
  if (params) B->params = *params; else lld_pose_params_default(&B->params);
  B->mode = pose_mode(ctx, n_frames, C);
  B->lay = pose_layout(n_frames, C.np, C.ne, C.nl, !B->mode.use_lds);
  B->h_frames.resize(n_frames);
  std::vector<char> img(B->lay.in_bytes);
  pose_pack(n_frames, frames, B->params.gamma, B->lay, img.data(), B->h_frames.data());
  if (hipMalloc(reinterpret_cast<void**>(&B->slab), B->lay.total) != hipSuccess) { delete B; return LLD_ERR_ALLOC; }
  if (hipMemcpy(B->slab, img.data(), B->lay.in_bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(B->slab); delete B; return LLD_ERR_HIP; }
  *out = B;
  return LLD_OK;
}

int lld_pose_batch_solve(lld_pose_batch* B) {
  if (!B) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(B->ctx->device));
  B->fetched = false;
  return pose_launch(B->ctx, B->n_frames, B->slab, B->lay, B->mode, B->params);   // the kernel resets its own working state
}

int lld_pose_batch_download(lld_pose_batch* B, int frame, lld_pose_result* out) {
  if (!B || !out || frame < 0 || frame >= B->n_frames) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(B->ctx->device));
  if (!B->fetched) {
    B->h_out.resize(B->lay.out_bytes);
    LLD_HIP_TRY(hipMemcpyAsync(B->h_out.data(), B->slab + B->lay.in_bytes, B->lay.out_bytes, hipMemcpyDeviceToHost, B->ctx->stream));
    LLD_HIP_TRY(hipStreamSynchronize(B->ctx->stream));
    B->fetched = true;
  }
  pose_fill_result(B->lay, B->h_out.data(), B->h_frames[frame], frame, out);
  return LLD_OK;
}

void lld_pose_batch_destroy(lld_pose_batch* B) {
  if (!B) return;
  (void)hipSetDevice(B->ctx->device);
  (void)hipStreamSynchronize(B->ctx->stream);
  if (B->slab) (void)hipFree(B->slab);
  delete B;
}

// One frame, host buffers in and out (what the Tracking thread calls three times per frame): the packed image goes through the
// context's pinned staging and device scratch, i.e. one H2D copy, one kernel, one D2H copy and no allocation in steady state.
int lld_pose_opt(lld_ctx* ctx, const lld_pose_problem* in, const lld_pose_params* params, lld_pose_result* out) {
  if (!ctx || !in || !out) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  lld_pose_params prm;
  if (params) prm = *params; else lld_pose_params_default(&prm);
  PoseCounts C;
  int st = pose_count(1, in, &C); if (st) return st;
  const PoseMode M = pose_mode(ctx, 1, C);
  const PoseLayout Y = pose_layout(1, C.np, C.ne, C.nl, !M.use_lds);
  void* hb; st = lld_ctx_pinned(ctx, Y.in_bytes + Y.out_bytes, &hb); if (st) return st;
  void* db; st = lld_ctx_scratch(ctx, Y.total, &db); if (st) return st;
  char* h_img = static_cast<char*>(hb); char* d_img = static_cast<char*>(db);
  PoseFrameDev F;
  pose_pack(1, in, prm.gamma, Y, h_img, &F);
  hipStream_t s = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d_img, h_img, Y.in_bytes, hipMemcpyHostToDevice, s));
  st = pose_launch(ctx, 1, d_img, Y, M, prm); if (st) return st;
  LLD_HIP_TRY(hipMemcpyAsync(h_img + Y.in_bytes, d_img + Y.in_bytes, Y.out_bytes, hipMemcpyDeviceToHost, s));
  LLD_HIP_TRY(hipStreamSynchronize(s));
  pose_fill_result(Y, h_img + Y.in_bytes, F, 0, out);
  return LLD_OK;
}

}  // extern "C"

#ifdef LLD_EXPERIMENTS
// experiments build only: one lld_pose_opt with the per-stage s_memtime sums of the frame's first wavefront (tools/pose_stage_budget.py)
extern "C" __attribute__((visibility("default"))) int lld_exp_pose_stamps(lld_ctx* ctx, const lld_pose_problem* in, const lld_pose_params* params, lld_pose_result* out, long long* stamps16) {
  if (!ctx || !in || !out || !stamps16) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  lld_pose_params prm;
  if (params) prm = *params; else lld_pose_params_default(&prm);
  PoseCounts C;
  int st = pose_count(1, in, &C); if (st) return st;
  const PoseMode M = pose_mode(ctx, 1, C);
  if (!M.use_lds || !M.f32) return LLD_ERR_UNSUPPORTED;
  const PoseLayout Y = pose_layout(1, C.np, C.ne, C.nl, false);
  void* hb; st = lld_ctx_pinned(ctx, Y.in_bytes + Y.out_bytes, &hb); if (st) return st;
  void* db; st = lld_ctx_scratch(ctx, Y.total + 512, &db); if (st) return st;
  char* h_img = static_cast<char*>(hb); char* d_img = static_cast<char*>(db);
  PoseFrameDev F;
  pose_pack(1, in, prm.gamma, Y, h_img, &F);
  hipStream_t s = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d_img, h_img, Y.in_bytes, hipMemcpyHostToDevice, s));
  PoseArrays A = pose_arrays(d_img, Y);
  A.stamps = reinterpret_cast<long long*>(d_img + lld_slab::pad(Y.total));
  st = pose_launch_as<true, float, kPoseThreadsMax>(ctx, s, 1, M.lds_bytes, reinterpret_cast<const PoseFrameDev*>(d_img + Y.frames), A, reinterpret_cast<PoseOut*>(d_img + Y.out), prm);
  if (st) return st;
  LLD_HIP_TRY(hipMemcpyAsync(h_img + Y.in_bytes, d_img + Y.in_bytes, Y.out_bytes, hipMemcpyDeviceToHost, s));
  LLD_HIP_TRY(hipStreamSynchronize(s));
  pose_fill_result(Y, h_img + Y.in_bytes, F, 0, out);
  LLD_HIP_TRY(hipMemcpy(stamps16, A.stamps, sizeof(long long) * PO_N_STAMPS, hipMemcpyDeviceToHost));
  return LLD_OK;
}
#endif

#ifdef LLD_EXPERIMENTS
// experiments build only: the cross-lane sums of pose_opt_kernel on their own (tests/test_gpu_pose.py): `in` holds [64][28] doubles (lane-major);
// out28[28] = wave_sum28 of them, out1[64] = wave_sum_all1 of every lane's first value as each lane sees it.
namespace {
__global__ void pose_wave_sums_kernel(const double* in, double* out28, double* out1) {
  __shared__ double dst[28];
  double v[28];
  for (int i = 0; i < 28; i++) v[i] = in[threadIdx.x * 28 + i];
  wave_sum28(v, dst);
  out1[threadIdx.x] = wave_sum_all1(v[0]);
  __syncthreads();
  if (threadIdx.x < 28) out28[threadIdx.x] = dst[threadIdx.x];
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int lld_exp_pose_wave_sums(lld_ctx* ctx, const double* in, double* out28, double* out1) {
  if (!ctx || !in || !out28 || !out1) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  void* db; int st = lld_ctx_scratch(ctx, (64 * 28 + 28 + 64) * sizeof(double), &db); if (st) return st;
  double* d = static_cast<double*>(db);
  hipStream_t s = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, in, 64 * 28 * sizeof(double), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(pose_wave_sums_kernel, dim3(1), dim3(64), 0, s, d, d + 64 * 28, d + 64 * 28 + 28);
  LLD_HIP_TRY(hipGetLastError());
  LLD_HIP_TRY(hipMemcpyAsync(out28, d + 64 * 28, 28 * sizeof(double), hipMemcpyDeviceToHost, s));
  LLD_HIP_TRY(hipMemcpyAsync(out1, d + 64 * 28 + 28, 64 * sizeof(double), hipMemcpyDeviceToHost, s));
  LLD_HIP_TRY(hipStreamSynchronize(s));
  return LLD_OK;
}
#endif
