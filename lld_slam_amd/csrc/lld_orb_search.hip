// lld_orb_search.hip — guided ORB search on the device: one kernel runs the whole body of an ORBmatcher::Search* /
// Fuse / Frame::ComputeStereoMatches routine (reference lines cited at each rule below).
//
// One workgroup owns one (query set, keypoint set) problem and one LANE owns one query (a query has only tens of candidates, so
// the search is latency-bound: 1024 queries in flight per workgroup hide it; a batch puts one problem on each CU):
//   * the frame's 64x48 keypoint grid (Frame::AssignFeaturesToGrid, src/Frame.cc:294-313) is rebuilt in LDS by a counting
//     sort, so a window query touches only the grid columns GetFeaturesInArea (src/Frame.cc:391-444) would visit;
//   * the keypoint records and (when they fit) the descriptors stay in LDS for all rounds;
//   * every candidate gets the 64-bit key  dist<<32 | visit-order ; the lane keeps the two smallest keys, which is exactly the
//     reference's strict-'<' best / second-best bookkeeping (first visited wins ties) without visiting in order;
//   * the order-dependent rule "keypoint already taken by an earlier query" (the reference writes mvpMapPoints / vpMatched
//     inside its loop) is solved by fixed-point rounds: round r sees the keypoints claimed in round r-1 by queries with a
//     smaller index; when a round changes no match the result is the sequential one (query i is final after i+1 rounds at
//     the latest, in practice 2-4 rounds);
//   * float gates use explicit round-to-nearest intrinsics so no FMA contraction can change a comparison.
#include "lld_common.h"
#include "lld_track_internal.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kHisto = 30;                 // HISTO_LENGTH, src/ORBmatcher.cc:39
constexpr unsigned long long kNone = ~0ull;

struct TKey {        // one keypoint of the searched frame, LDS resident
  float x, y, ur;
  int32_t meta;      // octave [0,4) | index [4,16) | cell+1 [16,29) (0 = outside the grid) | occupied bit 29
};
__device__ __forceinline__ int tk_oct(int m) { return m & 15; }
__device__ __forceinline__ int tk_idx(int m) { return (m >> 4) & 4095; }
__device__ __forceinline__ int tk_cell(int m) { return ((m >> 16) & 8191) - 1; }
__device__ __forceinline__ bool tk_occ(int m) { return (m >> 29) & 1; }

struct QRec {        // one query, packed on the host
  float u, v, radius, ur, stereo_radius, angle;
  float ea, eb, ec;
  int32_t level_min, level_max;
  int32_t flags;     // 1 valid | 2 blocks | 4 stereo (bStereo1)
  int32_t cs, ce;    // CSR range
  int32_t pad0, pad1;
};

struct Problem {
  int nt, nq;
  const uint32_t* t_desc; const float* t_xy; const int32_t* t_octave; const float* t_uright; const float* t_angle; const uint8_t* t_occupied;
  const uint32_t* q_desc; const QRec* q; const int32_t* cand_idx;
  float min_x, min_y, winv, hinv; int cols, rows;
  int n_levels; float scale[LLD_ORB_MAX_LEVELS], sigma2[LLD_ORB_MAX_LEVELS], inv_sigma2[LLD_ORB_MAX_LEVELS];
  float disp_min, disp_max, epi_x, epi_y; int only_stereo;
  int candidates, gates, tie_last, accept_max, ratio_mode; float nnratio; int sequential, check_orientation;
  int32_t* match; int32_t* best_dist; int32_t* second_dist; uint8_t* removed; int32_t* owner; int32_t* summary;   // summary: n_matches, rounds
  int desc_in_lds, want_owner;
  unsigned long long* cache;   // [nq][kTopK] device scratch of sequential problems: the smallest candidate keys of round 1
  // device-resident chains (lld_frame_track_*): the whole search is skipped unless (*run_if < run_if_below) == (run_if_want != 0)
  const int32_t* run_if; int run_if_below, run_if_want;
  const uint8_t* t_occ_obs;    // non-null: a keypoint is occupied only if t_occupied[k] AND t_occ_obs[k] (its MapPoint has observations, ORBmatcher.cc:98-100, :1409-1411)
  // ... and hands the frame what it matched: CurrentFrame.mvpMapPoints[bestIdx] = pMP (src/ORBmatcher.cc:124, :1427) for every keypoint the search
  // leaves with an owner, if it accepted at least ap_min_matches (Tracking.cc:907: a first search below 20 is thrown away and repeated wider)
  uint8_t* ap_has; float* ap_world; int32_t* ap_id; uint8_t* ap_obs;
  const float* ap_q_pos; const int32_t* ap_q_id; const uint8_t* ap_q_obs;
  int32_t* ap_counts; int ap_min_matches, ap_is_retry;      // ap_counts: [0] n of the first search, [1] n of the search whose matches were taken, [2] retry used
};

template <class Ptr>
__device__ __forceinline__ int hamming256(const uint32_t (&a)[8], Ptr b) {
  const uint4 b0 = *reinterpret_cast<const uint4*>(b), b1 = *reinterpret_cast<const uint4*>(b + 4);
  return __popc(a[0] ^ b0.x) + __popc(a[1] ^ b0.y) + __popc(a[2] ^ b0.z) + __popc(a[3] ^ b0.w) +
         __popc(a[4] ^ b1.x) + __popc(a[5] ^ b1.y) + __popc(a[6] ^ b1.z) + __popc(a[7] ^ b1.w);
}

__device__ __forceinline__ void top2_insert(unsigned long long c, unsigned long long& b1, unsigned long long& b2) {
  if (c < b1) { b2 = b1; b1 = c; } else if (c < b2) { b2 = c; }
}

// the kTopK smallest keys, ascending
constexpr int kTopK = 8;
__device__ __forceinline__ void topk_insert(unsigned long long k, unsigned long long (&c)[kTopK]) {
  if (k < c[kTopK - 1]) {
    c[kTopK - 1] = k;
#pragma unroll
    for (int i = kTopK - 1; i > 0; i--) {
      const unsigned long long lo = c[i] < c[i - 1] ? c[i] : c[i - 1], hi = c[i] < c[i - 1] ? c[i - 1] : c[i];
      c[i - 1] = lo; c[i] = hi;
    }
  }
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  const int lo = __shfl_xor((int)(v & 0xffffffffu), m), hi = __shfl_xor((int)(v >> 32), m);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// Per-candidate skip rules that do not depend on how the candidate was produced.
__device__ __forceinline__ bool gates_pass(const Problem& P, const QRec& Q, const TKey& T, int q, const int* blk, bool use_blk, const float* lvl /* LDS: scale[16], sigma2[16], inv_sigma2[16] */) {
  const int m = T.meta;
  if (tk_occ(m)) return false;
  if (use_blk && blk[tk_idx(m)] < q) return false;
  const int oct = tk_oct(m);
  if (P.gates & LLD_ORB_GATE_LEVEL) {                                     // Frame.cc:422-430, ORBmatcher.cc:386,907
    if (oct < Q.level_min) return false;
    if (Q.level_max >= 0 && oct > Q.level_max) return false;
  }
  if (P.gates & LLD_ORB_GATE_STEREO) {                                    // ORBmatcher.cc:90-95, 1400-1406
    if (T.ur > 0.f && fabsf(__fsub_rn(Q.ur, T.ur)) > Q.stereo_radius) return false;
  }
  if (P.gates & LLD_ORB_GATE_CHI2) {                                      // ORBmatcher.cc:912-936
    const float ex = __fsub_rn(Q.u, T.x), ey = __fsub_rn(Q.v, T.y);
    float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
    if (T.ur >= 0.f) {
      const float er = __fsub_rn(Q.ur, T.ur);
      e2 = __fadd_rn(e2, __fmul_rn(er, er));
      if ((double)__fmul_rn(e2, lvl[32 + oct]) > 7.8) return false;
    } else {
      if ((double)__fmul_rn(e2, lvl[32 + oct]) > 5.99) return false;
    }
  }
  if (P.gates & LLD_ORB_GATE_EPIPOLAR) {                                  // ORBmatcher.cc:720-751, 138-157
    const bool s1 = (Q.flags & 4) != 0, s2 = T.ur >= 0.f;
    if (P.only_stereo && !s2) return false;
    if (!s1 && !s2) {
      const float dx = __fsub_rn(P.epi_x, T.x), dy = __fsub_rn(P.epi_y, T.y);
      if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.f, lvl[oct])) return false;
    }
    const float num = __fadd_rn(__fadd_rn(__fmul_rn(Q.ea, T.x), __fmul_rn(Q.eb, T.y)), Q.ec);
    const float den = __fadd_rn(__fmul_rn(Q.ea, Q.ea), __fmul_rn(Q.eb, Q.eb));
    if (den == 0.f) return false;
    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
    if (!((double)dsqr < 3.84 * (double)lvl[16 + oct])) return false;
  }
  return true;
}

// ---- the reference's OpenCV calls on float data, as this build restates them (see include/lld_amd.h) --------------------------
// `R*P + t` is ONE cv::gemm: double accumulation in k order, one rounding to float
__device__ __forceinline__ void cv_transform(const lld_frame_view& V, const float* P, float* Pc) {
#pragma unroll
  for (int r = 0; r < 3; r++)
    Pc[r] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)V.Rcw[3 * r], (double)P[0]), __dmul_rn((double)V.Rcw[3 * r + 1], (double)P[1])),
                                       __dmul_rn((double)V.Rcw[3 * r + 2], (double)P[2])), (double)V.tcw[r]);
}
// cv::norm(a) and a.dot(b) of CV_32F vectors accumulate in double
__device__ __forceinline__ double cv_dot3(const float* a, const float* b) {
  return __dadd_rn(__dadd_rn(__dmul_rn((double)a[0], (double)b[0]), __dmul_rn((double)a[1], (double)b[1])), __dmul_rn((double)a[2], (double)b[2]));
}
__device__ __forceinline__ float cv_norm3(const float* a) { return (float)__dsqrt_rn(cv_dot3(a, a)); }
// log(float) as the reference's libm computes it.  MapPoint::PredictScale takes ceil(log(ratio) / mfLogScaleFactor): where the quotient
// lands on an integer, a logarithm that is one unit in the last place off moves the predicted level by one, and the device library's
// logf and glibc's are both "within an ulp" without being the same function (found by tools/fuzz_matchers.py with FUZZ_BIG=1: one
// level in 6128 in-view points of one scene in 100 000).  This is glibc's algorithm (sysdeps/ieee754/flt-32/e_logf.c and
// logf_data.c, glibc >= 2.27, taken from ARM's optimized routines; constants checked against the libm.so.6 of this image, glibc 2.35):
// x = 2^k z with z in [0x1.66p-1, 0x1.66p0), sixteen sub-intervals with 1/c and log(c) tabulated, log1p(z/c - 1) by a cubic, all in
// double, rounded to float once.  tests/test_oracle_kat.py holds the same restatement (oracle/) against std::log(float) on this host;
// contraction of its multiply-adds does not change the float result (0 differences in 2e8 random arguments either way).
__device__ __forceinline__ float glibc_logf(float x) {
  const double T[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
      {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
      {0x1p+0, 0x0p+0},                              {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
      {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},   {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
      {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
  const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  const uint32_t ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return logf(x);            // zero, subnormal, negative, inf, nan: not a distance ratio
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u), k = (int)tmp >> 23;                   // arithmetic shift
  const double z = (double)__uint_as_float(ix - (tmp & 0xff800000u));
  const double r = __dsub_rn(__dmul_rn(z, T[i][0]), 1.0), y0 = __dadd_rn(T[i][1], __dmul_rn((double)k, Ln2)), r2 = __dmul_rn(r, r);
  double y = __dadd_rn(__dmul_rn(A1, r), A2);
  y = __dadd_rn(__dmul_rn(A0, r2), y);
  y = __dadd_rn(__dmul_rn(y, r2), __dadd_rn(y0, r));
  return (float)y;
}

// MapPoint::PredictScale (src/MapPoint.cc:402-417): float log, float division, ceil, clamp
__device__ __forceinline__ int predict_scale(float max_distance, float dist, const lld_frame_view& V) {
  const float ratio = __fdiv_rn(max_distance, dist);
  int n = (int)ceilf(__fdiv_rn(glibc_logf(ratio), V.log_scale_factor));
  if (n < 0) n = 0; else if (n >= V.n_levels) n = V.n_levels - 1;
  return n;
}

// Frame::isInFrustum (src/Frame.cc:333-389) for one MapPoint per lane, written straight into the query record of the search
// kernel.  Float / double mixture as the reference's OpenCV calls (see include/lld_amd.h); every float operation is an explicit
// round-to-nearest intrinsic, so nothing is contracted into an FMA.
struct FrustumArgs {
  lld_frame_view V;
  int n;
  const float* pos; const float* nrm; const float* maxd; const float* mind; const uint8_t* has_obs; const uint8_t* skip;
  float scale[LLD_ORB_MAX_LEVELS];
  float cos_limit, th;
  QRec* q; uint8_t* in_view; float* uvr; int32_t* level; float* view_cos;
  const lld_frame_view* view_d;          // non-null: the view lives in device memory (the pose was optimised on the device)
  int32_t* n_in_view;                    // non-null: += the number of points inside the frustum
};

__device__ __forceinline__ bool frustum_one(const FrustumArgs& F, int i) {
  QRec Q; memset(&Q, 0, sizeof(Q));
  Q.level_min = -1; Q.level_max = -1;
  Q.flags = (!F.has_obs || F.has_obs[i]) ? 2 : 0;
  bool ok = !(F.skip && F.skip[i]);
  float u = 0.f, v = 0.f, ur = 0.f, vc = 0.f; int lvl = 0;
  do {
    if (!ok) break;
    const float P[3] = {F.pos[3 * i], F.pos[3 * i + 1], F.pos[3 * i + 2]};
    float Pc[3]; cv_transform(F.V, P, Pc);
    ok = false;
    if (Pc[2] < 0.0f) break;
    const float invz = __fdiv_rn(1.0f, Pc[2]);
    u = __fadd_rn(__fmul_rn(__fmul_rn(F.V.fx, Pc[0]), invz), F.V.cx);
    v = __fadd_rn(__fmul_rn(__fmul_rn(F.V.fy, Pc[1]), invz), F.V.cy);
    if (u < F.V.min_x || u > F.V.max_x) break;
    if (v < F.V.min_y || v > F.V.max_y) break;
    const float maxDistance = __fmul_rn(1.2f, F.maxd[i]), minDistance = __fmul_rn(0.8f, F.mind[i]);
    const float PO[3] = {__fsub_rn(P[0], F.V.Ow[0]), __fsub_rn(P[1], F.V.Ow[1]), __fsub_rn(P[2], F.V.Ow[2])};
    const float dist = cv_norm3(PO);
    if (dist < minDistance || dist > maxDistance) break;
    vc = (float)__ddiv_rn(cv_dot3(PO, F.nrm + 3 * i), (double)dist);             // PO.dot(Pn)/dist
    if (vc < F.cos_limit) break;
    lvl = predict_scale(F.maxd[i], dist, F.V);
    ur = __fsub_rn(u, __fmul_rn(F.V.bf, invz));
    ok = true;
  } while (false);
  if (ok) {
    float r = ((double)vc > 0.998) ? 2.5f : 4.0f;                                // RadiusByViewingCos, src/ORBmatcher.cc:131-137
    if (F.th != 1.0f) r = __fmul_rn(r, F.th);
    const float radius = __fmul_rn(r, F.scale[lvl]);
    Q.u = u; Q.v = v; Q.radius = radius; Q.ur = ur; Q.stereo_radius = radius;
    Q.level_min = lvl - 1; Q.level_max = lvl;
    Q.flags |= 1;
  }
  F.q[i] = Q;
  if (F.in_view) F.in_view[i] = ok ? 1 : 0;
  if (F.uvr) { F.uvr[3 * i] = u; F.uvr[3 * i + 1] = v; F.uvr[3 * i + 2] = ur; }
  if (F.level) F.level[i] = lvl;
  if (F.view_cos) F.view_cos[i] = vc;
  return ok;
}
__global__ __launch_bounds__(256) void frustum_kernel(FrustumArgs F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < F.n) {
    if (F.view_d) F.V = *F.view_d;
    ok = frustum_one(F, i);
  }
  if (F.n_in_view) {                                                            // nToMatch (src/Tracking.cc:1647), one atomic per wavefront
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(F.n_in_view, __popcll(m));
  }
}

// Projection loop of ORBmatcher::SearchByProjection(Current, Last, th, bMono) (src/ORBmatcher.cc:1352-1386), one lane per keypoint of
// the last frame, written straight into the query record of the search kernel.
struct LastFrameArgs {
  lld_frame_view V;
  int n, direction;
  const float* pos; const uint8_t* valid; const int32_t* octave; const float* angle; const uint8_t* has_obs;
  float scale[LLD_ORB_MAX_LEVELS];
  float th;
  QRec* q; float* uvr;
  const lld_frame_view* view_d;
  const int32_t* run_if; int run_if_below, run_if_want;
  // ... and hands the frame what it matched: CurrentFrame.mvpMapPoints[bestIdx] = pMP (src/ORBmatcher.cc:124, :1427) for every keypoint the search
  // leaves with an owner, if it accepted at least ap_min_matches (Tracking.cc:907: a first search below 20 is thrown away and repeated wider)
  uint8_t* ap_has; float* ap_world; int32_t* ap_id; uint8_t* ap_obs;
  const float* ap_q_pos; const int32_t* ap_q_id; const uint8_t* ap_q_obs;
  int32_t* ap_counts; int ap_min_matches, ap_is_retry;      // ap_counts: [0] n of the first search, [1] n of the search whose matches were taken, [2] retry used
};

__device__ __forceinline__ void project_last_one(const LastFrameArgs& F, int i) {
  QRec Q; memset(&Q, 0, sizeof(Q));
  Q.level_min = -1; Q.level_max = -1;
  Q.flags = (!F.has_obs || F.has_obs[i]) ? 2 : 0;
  Q.angle = F.angle ? F.angle[i] : 0.f;
  float u = 0.f, v = 0.f, ur = 0.f;
  if (F.valid[i]) {
    const float P[3] = {F.pos[3 * i], F.pos[3 * i + 1], F.pos[3 * i + 2]};
    float Pc[3]; cv_transform(F.V, P, Pc);                                       // x3Dc = Rcw*x3Dw+tcw
    const float invzc = (float)__ddiv_rn(1.0, (double)Pc[2]);                   // const float invzc = 1.0/x3Dc.at<float>(2);
    if (!(invzc < 0.f)) {
      u = __fadd_rn(__fmul_rn(__fmul_rn(F.V.fx, Pc[0]), invzc), F.V.cx);
      v = __fadd_rn(__fmul_rn(__fmul_rn(F.V.fy, Pc[1]), invzc), F.V.cy);
      if (!(u < F.V.min_x || u > F.V.max_x) && !(v < F.V.min_y || v > F.V.max_y)) {
        const int oct = F.octave[i];
        const float radius = __fmul_rn(F.th, F.scale[oct]);
        ur = __fsub_rn(u, __fmul_rn(F.V.bf, invzc));                             // :1402
        Q.u = u; Q.v = v; Q.radius = radius; Q.ur = ur; Q.stereo_radius = radius;
        if (F.direction > 0) { Q.level_min = oct; Q.level_max = -1; }            // GetFeaturesInArea(u,v,radius,nLastOctave)
        else if (F.direction < 0) { Q.level_min = 0; Q.level_max = oct; }        // (u,v,radius,0,nLastOctave)
        else { Q.level_min = oct - 1; Q.level_max = oct + 1; }
        Q.flags |= 1;
      }
    }
  }
  F.q[i] = Q;
  if (F.uvr) { F.uvr[3 * i] = u; F.uvr[3 * i + 1] = v; F.uvr[3 * i + 2] = ur; }
}
__global__ __launch_bounds__(256) void project_last_frame_kernel(LastFrameArgs F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n) return;
  if (F.run_if && ((*F.run_if < F.run_if_below) != (F.run_if_want != 0))) return;
  if (F.view_d) F.V = *F.view_d;
  project_last_one(F, i);
}

__global__ __launch_bounds__(kThreads) void orb_search_kernel(const Problem* __restrict__ problems) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const Problem& P = problems[blockIdx.x];
  if (P.run_if && ((*P.run_if < P.run_if_below) != (P.run_if_want != 0))) return;
  const int nt = P.nt, nq = P.nq, tid = threadIdx.x;
  const int n_cells = P.cols * P.rows;
  const bool grid = P.candidates == LLD_ORB_CAND_GRID, rows = P.candidates == LLD_ORB_CAND_ROWS;
  const bool bucketed = grid || rows;                                           // keypoints sorted by grid cell / by image row
  TKey* tk = reinterpret_cast<TKey*>(lds_raw);                                  // [nt]   (bucketed: sorted by cell)
  uint32_t* dsc = reinterpret_cast<uint32_t*>(tk + nt);                         // [nt][8] descriptors in tk order (when they fit)
  int* blk = reinterpret_cast<int*>(dsc + (P.desc_in_lds ? (size_t)nt * 8 : 0));  // [nt]   first blocking query per keypoint index
  int* cell_start = blk + nt;                                                   // [n_cells + 1]
  int* hist = cell_start + n_cells + 1;                                         // [32]
  int* ctl = hist + 32;                                                         // [8]: 0 changed, 1 accepted, 2 removed, 3..5 kept bins
  int* scan = ctl + 8;                                                          // [kThreads]
  float* lvl = reinterpret_cast<float*>(scan + kThreads);                       // [48] level tables: scale, sigma2, 1/sigma2
  int* holder = reinterpret_cast<int*>(lvl + 48);                               // [nt]   sequential == 2: the query that holds keypoint k (vnMatches21)
  unsigned long long* lkeys = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(holder + nt) + 15) & ~uintptr_t(15));   // [64][kTopK], 16-byte aligned

  // ---------------------------------------------------------------- keypoints into LDS (+ grid counting sort)
  auto load_key = [&](int k, int& cell) -> TKey {
    TKey T; T.x = P.t_xy[2 * k]; T.y = P.t_xy[2 * k + 1]; T.ur = P.t_uright ? P.t_uright[k] : -1.f;
    cell = -1;
    if (grid) {                                                                 // Frame::PosInGrid, src/Frame.cc:446-456
      const int px = (int)roundf(__fmul_rn(__fsub_rn(T.x, P.min_x), P.winv)), py = (int)roundf(__fmul_rn(__fsub_rn(T.y, P.min_y), P.hinv));
      if (px >= 0 && px < P.cols && py >= 0 && py < P.rows) cell = px * P.rows + py;
    } else if (rows) {                                                          // bucket = (octave, image row) of the keypoint (search aid only)
      cell = min(P.t_octave[k], P.cols - 1) * P.rows + min(max((int)floorf(T.y), 0), P.rows - 1);
    }
    T.meta = (P.t_octave[k] & 15) | (k << 4) | ((cell + 1) << 16) | ((P.t_occupied && P.t_occupied[k] && (!P.t_occ_obs || P.t_occ_obs[k])) ? (1 << 29) : 0);
    return T;
  };
  auto place = [&](int pos, int k, const TKey& T) {
    tk[pos] = T;
    if (P.desc_in_lds) {
      const uint4* src = reinterpret_cast<const uint4*>(P.t_desc + 8 * (size_t)k);
      uint4* dst = reinterpret_cast<uint4*>(dsc + 8 * (size_t)pos);
      dst[0] = src[0]; dst[1] = src[1];
    }
  };
  for (int c = tid; c <= n_cells; c += kThreads) cell_start[c] = 0;
  for (int k = tid; k < nt; k += kThreads) blk[k] = 0x7fffffff;
  if (tid < 32) hist[tid] = 0;
  if (tid < 8) ctl[tid] = 0;
  if (tid < LLD_ORB_MAX_LEVELS) { lvl[tid] = P.scale[tid]; lvl[16 + tid] = P.sigma2[tid]; lvl[32 + tid] = P.inv_sigma2[tid]; }
  __syncthreads();
  for (int k = tid; k < nt; k += kThreads) {
    int cell; const TKey T = load_key(k, cell);
    if (!bucketed) place(k, k, T);
    else if (cell >= 0) atomicAdd(&cell_start[cell], 1);
  }
  __syncthreads();
  if (bucketed) {
    // scan of the cell counts (each thread owns a contiguous run of cells) into cell ENDS, then an unordered placement that
    // counts every cell's end down to its start - one array serves as counter, cursor and final cell_start (end of cell c =
    // start of cell c + 1).  The position inside a cell is irrelevant: ties are broken by the key, not by the visit position.
    const int per = (n_cells + kThreads - 1) / kThreads, c0 = min(tid * per, n_cells), c1 = min(c0 + per, n_cells);
    int sum = 0;
    for (int c = c0; c < c1; c++) sum += cell_start[c];
    // block-wide exclusive prefix of the per-thread sums: shuffle scan inside the wavefront + the 16 wavefront totals
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if ((tid & 63) >= off) incl += t; }
    if ((tid & 63) == 63) scan[tid >> 6] = incl;
    __syncthreads();
    int run = incl - sum, total = 0;
    for (int w = 0; w < kThreads / 64; w++) { const int t = scan[w]; if (w < (tid >> 6)) run += t; total += t; }
    for (int c = c0; c < c1; c++) { run += cell_start[c]; cell_start[c] = run; }
    if (tid == kThreads - 1) cell_start[n_cells] = total;
    __syncthreads();
    for (int k = tid; k < nt; k += kThreads) {
      int cell; const TKey T = load_key(k, cell);
      if (cell >= 0) place(atomicSub(&cell_start[cell], 1) - 1, k, T);
    }
  }
  for (int q = tid; q < nq; q += kThreads) P.match[q] = -2;
  __syncthreads();

  // every candidate of query q that passes the gates, its kTopK smallest keys (distance, then visit order) into c.  `steal`: the
  // SearchForInitialization rule - a keypoint held by an earlier query with a distance <= this one is no candidate (blk = vMatchedDistance)
  auto scan_candidates = [&](int q, const QRec& Q, unsigned long long (&c)[kTopK], bool use_blk, bool steal) {
#pragma unroll
        for (int i = 0; i < kTopK; i++) c[i] = kNone;
        uint32_t qd[8];
        {
          const uint4 a = *reinterpret_cast<const uint4*>(P.q_desc + 8 * (size_t)q), b = *reinterpret_cast<const uint4*>(P.q_desc + 8 * (size_t)q + 4);
          qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w; qd[4] = b.x; qd[5] = b.y; qd[6] = b.z; qd[7] = b.w;
        }
        auto visit = [&](int pos, const TKey& T, unsigned key) {
          if (!gates_pass(P, Q, T, q, blk, use_blk, lvl)) return;
          const int d = P.desc_in_lds ? hamming256(qd, dsc + 8 * (size_t)pos) : hamming256(qd, P.t_desc + 8 * (size_t)tk_idx(T.meta));
          if (steal && blk[tk_idx(T.meta)] <= d) return;                                                           // ORBmatcher.cc:443-444
          if (P.tie_last) key = ~key;
          topk_insert(((unsigned long long)d << 32) | key, c);
        };
        if (grid) {
          // GetFeaturesInArea cell range, src/Frame.cc:396-410 (float arithmetic, floor/ceil, clamps and early returns)
          const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.u, P.min_x), Q.radius), P.winv)));
          const int maxCX = min(P.cols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.u, P.min_x), Q.radius), P.winv)));
          const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.v, P.min_y), Q.radius), P.hinv)));
          const int maxCY = min(P.rows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.v, P.min_y), Q.radius), P.hinv)));
          if (minCX < P.cols && maxCX >= 0 && minCY < P.rows && maxCY >= 0 && maxCY >= minCY) {
            for (int ix = minCX; ix <= maxCX; ix++) {
              const int j0 = cell_start[ix * P.rows + minCY], j1 = cell_start[ix * P.rows + maxCY + 1];   // one column = one contiguous run
              for (int j = j0; j < j1; j++) {
                const TKey T = tk[j];
                if (!(fabsf(__fsub_rn(T.x, Q.u)) < Q.radius && fabsf(__fsub_rn(T.y, Q.v)) < Q.radius)) continue;   // Frame.cc:433-437
                visit(j, T, ((unsigned)tk_cell(T.meta) << 12) | (unsigned)tk_idx(T.meta));
              }
            }
          }
        } else if (P.candidates == LLD_ORB_CAND_CSR) {
          for (int p = Q.cs; p < Q.ce; p++) {
            const int k = P.cand_idx[p];
            visit(k, tk[k], (unsigned)(p - Q.cs));
          }
        } else if (rows) {
          // Frame::ComputeStereoMatches: vRowIndices[vL] holds right keypoint iR iff floor(yR-r) <= (int)vL <= ceil(yR+r), r = 2*scale[octave]
          // (src/Frame.cc:546-556).  Only the row buckets that can satisfy this are scanned; the exact test follows.
          const float minU = __fsub_rn(Q.u, P.disp_max), maxU = __fsub_rn(Q.u, P.disp_min);                       // Frame.cc:574-575
          const long long row = (long long)Q.v;                                                                    // vRowIndices[vL], :569
          if (!(maxU < 0.f)) {                                                                                     // :577-578
            // buckets are (octave, row): only the octaves the level gate admits are scanned, each with its own row margin
            int o_lo = 0, o_hi = P.cols - 1;
            if (P.gates & LLD_ORB_GATE_LEVEL) { o_lo = min(max(Q.level_min, 0), P.cols - 1); if (Q.level_max >= 0) o_hi = min(Q.level_max, P.cols - 1); }
            for (int o = o_lo; o <= o_hi; o++) {
              float rmax = __fmul_rn(2.0f, lvl[o]);
              if (o == P.cols - 1) for (int l = o + 1; l < LLD_ORB_MAX_LEVELS; l++) rmax = fmaxf(rmax, __fmul_rn(2.0f, lvl[l]));   // clamped octaves
              const long long margin = (long long)ceilf(rmax) + 2;
              const int lo = (int)min(max(row - margin, 0ll), (long long)P.rows - 1), hi = (int)min(max(row + margin, 0ll), (long long)P.rows - 1);
              for (int j = cell_start[o * P.rows + lo]; j < cell_start[o * P.rows + hi + 1]; j++) {
                const TKey T = tk[j];
                const float r = __fmul_rn(2.0f, lvl[tk_oct(T.meta)]);
                const long long maxr = (long long)ceilf(__fadd_rn(T.y, r)), minr = (long long)floorf(__fsub_rn(T.y, r));
                if (row < minr || row > maxr) continue;
                if (!(T.x >= minU && T.x <= maxU)) continue;                                                       // :594-596
                visit(j, T, (unsigned)tk_idx(T.meta));
              }
            }
          }
        } else {
          for (int k = 0; k < nt; k++) visit(k, tk[k], (unsigned)k);
        }
  };

  // ---------------------------------------------------------------- fixed-point rounds, one lane per query
  int rounds = 0;
  for (;;) {
    rounds++;
    for (int q = tid; q < nq; q += kThreads) {
      const QRec Q = P.q[q];
      unsigned long long b1 = kNone, b2 = kNone;
      auto decode = [&](unsigned long long c) -> int {
        unsigned key = (unsigned)(c & 0xffffffffu);
        if (P.tie_last) key = ~key;
        if (grid) return (int)(key & 4095u);
        if (P.candidates == LLD_ORB_CAND_CSR) return P.cand_idx[Q.cs + (int)key];
        return (int)key;
      };
      if (Q.flags & 1) {
        // The candidates of a query, sorted by key, do not depend on the round; only which of them are blocked does.  Round 1
        // keeps the kTopK smallest keys; later rounds take the first two unblocked ones from that list and rescan only when
        // the list was full and fewer than two of its entries are still free.
        unsigned long long c[kTopK];
#pragma unroll
        for (int i = 0; i < kTopK; i++) c[i] = kNone;
        auto pick = [&]() {
          int found = 0; b1 = kNone; b2 = kNone;
          auto take = [&](unsigned long long k) {
            if (k != kNone && !(P.sequential && blk[decode(k)] < q)) {
              b1 = found == 0 ? k : b1; b2 = found == 1 ? k : b2;          // selects, so that b1 / b2 stay in registers
              found++;
            }
          };
#pragma unroll
          for (int i = 0; i < kTopK; i++) take(c[i]);
          return found;
        };
        bool need_scan = rounds == 1;
        if (rounds > 1) {
#pragma unroll
          for (int i = 0; i < kTopK; i += 2) {
            const ulonglong2 t2 = *reinterpret_cast<const ulonglong2*>(P.cache + kTopK * (size_t)q + i);
            c[i] = t2.x; c[i + 1] = t2.y;
          }
          need_scan = pick() < 2 && c[kTopK - 1] != kNone;
        }
        if (need_scan) {
          scan_candidates(q, Q, c, rounds > 1, false);
        if (rounds == 1 && P.sequential) {
#pragma unroll
          for (int i = 0; i < kTopK; i += 2) *reinterpret_cast<ulonglong2*>(P.cache + kTopK * (size_t)q + i) = make_ulonglong2(c[i], c[i + 1]);
        }
        pick();
        }
      }
      int m = -1, bd = 256, sd = 256;
      if (b1 != kNone) {
        bd = (int)(b1 >> 32);
        const int bi = decode(b1);
        bool ok = bd <= P.accept_max;
        if (b2 != kNone) sd = (int)(b2 >> 32);
        if (ok && P.ratio_mode == 1) ok = (float)bd < __fmul_rn(P.nnratio, (float)sd);                            // ORBmatcher.cc:226-228
        if (ok && P.ratio_mode == 2) {                                                                            // :118-121
          const int lvl1 = P.t_octave[bi], lvl2 = (b2 != kNone) ? P.t_octave[decode(b2)] : -1;
          if (lvl1 == lvl2 && (float)bd > __fmul_rn(P.nnratio, (float)sd)) ok = false;
        }
        if (ok) m = bi;
      }
      if (P.match[q] != m) { P.match[q] = m; ctl[0] = 1; }
      P.best_dist[q] = bd; P.second_dist[q] = sd;
    }
    __syncthreads();
    const int changed = ctl[0];
    __syncthreads();
    if (P.sequential == 2) {
      // ---- ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:417-476): the one order-dependent rule that is not an occupancy.  A
      // candidate is skipped when an EARLIER query holds it with a distance <= this one (vMatchedDistance[i2] <= dist, :443), and an
      // accepted query takes the keypoint away from its holder (:458-465).  Round 1 above left every query's kTopK smallest candidates
      // (distance, then visit order) in P.cache; wavefront 0 walks the queries in order with lane i on list entry i: the first two
      // entries still free are bestDist / bestDist2.  Only when fewer than two of a FULL list are free is the window scanned again (by
      // one lane, with the rule applied).  blk = vMatchedDistance (INT_MAX so far), holder = vnMatches21; the rotation histogram
      // counts every acceptance, also those stolen later (rotHist is never cleaned, :466-476).
      for (int k = tid; k < nt; k += kThreads) holder[k] = -1;
      __syncthreads();
      if (tid < 64) {
        // No global load sits on the query-to-query chain: a chunk of 64 queries is loaded lane-wise (lane l: flags and list of query
        // q0 + l), the walk takes query i's values out of lane i's registers with v_readlane, results stay in lane registers until the
        // chunk is done, and the rotation bins (which need both angles) are counted afterwards by all lanes from the recorded acceptances.
        auto dec = [&](unsigned long long c) -> int { const unsigned kk = (unsigned)(c & 0xffffffffu); return grid ? (int)(kk & 4095u) : (int)kk; };
        auto rl64 = [&](unsigned long long v, int lane) -> unsigned long long {
          const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffu), lane), hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
          return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
        };
        // the window of query q once more, with the take-over rule, by the whole wavefront: lanes stride over the candidates of a grid
        // column (or over all keypoints), keep their two smallest keys, and a butterfly merges the 64 pairs
        auto wave_rescan = [&](int q, unsigned long long& r1, unsigned long long& r2) {
          const QRec Q = P.q[q];
          uint32_t qd[8];
          {
            const uint4 a = *reinterpret_cast<const uint4*>(P.q_desc + 8 * (size_t)q), b = *reinterpret_cast<const uint4*>(P.q_desc + 8 * (size_t)q + 4);
            qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w; qd[4] = b.x; qd[5] = b.y; qd[6] = b.z; qd[7] = b.w;
          }
          unsigned long long t1 = kNone, t2 = kNone;
          auto visit = [&](int pos, const TKey& T, unsigned key) {
            if (!gates_pass(P, Q, T, q, blk, false, lvl)) return;
            const int d = P.desc_in_lds ? hamming256(qd, dsc + 8 * (size_t)pos) : hamming256(qd, P.t_desc + 8 * (size_t)tk_idx(T.meta));
            if (blk[tk_idx(T.meta)] <= d) return;                                                                  // ORBmatcher.cc:443-444
            top2_insert(((unsigned long long)d << 32) | key, t1, t2);
          };
          if (grid) {
            const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.u, P.min_x), Q.radius), P.winv)));
            const int maxCX = min(P.cols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.u, P.min_x), Q.radius), P.winv)));
            const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.v, P.min_y), Q.radius), P.hinv)));
            const int maxCY = min(P.rows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.v, P.min_y), Q.radius), P.hinv)));
            if (minCX < P.cols && maxCX >= 0 && minCY < P.rows && maxCY >= 0 && maxCY >= minCY) {
              for (int ix = minCX; ix <= maxCX; ix++) {
                const int j0 = cell_start[ix * P.rows + minCY], j1 = cell_start[ix * P.rows + maxCY + 1];
                for (int j = j0 + tid; j < j1; j += 64) {
                  const TKey T = tk[j];
                  if (!(fabsf(__fsub_rn(T.x, Q.u)) < Q.radius && fabsf(__fsub_rn(T.y, Q.v)) < Q.radius)) continue;
                  visit(j, T, ((unsigned)tk_cell(T.meta) << 12) | (unsigned)tk_idx(T.meta));
                }
              }
            }
          } else {
            for (int k = tid; k < nt; k += 64) visit(k, tk[k], (unsigned)k);
          }
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o1 = shfl_xor_u64(t1, off), o2 = shfl_xor_u64(t2, off);
            const unsigned long long lo = t1 < o1 ? t1 : o1, hi = t1 < o1 ? o1 : t1, s2 = t2 < o2 ? t2 : o2;
            t1 = lo; t2 = hi < s2 ? hi : s2;
          }
          r1 = t1; r2 = t2;
        };
        for (int q0 = 0; q0 < nq; q0 += 64) {
          const int ql = q0 + tid;
          int fl = 0;
          {
            unsigned long long kk[kTopK];
#pragma unroll
            for (int t = 0; t < kTopK; t++) kk[t] = kNone;
            if (ql < nq) {
              fl = P.q[ql].flags;
#pragma unroll
              for (int t = 0; t < kTopK; t += 2) {
                const ulonglong2 t2 = *reinterpret_cast<const ulonglong2*>(P.cache + kTopK * (size_t)ql + t);
                kk[t] = t2.x; kk[t + 1] = t2.y;
              }
            }
#pragma unroll
            for (int t = 0; t < kTopK; t += 2) *reinterpret_cast<ulonglong2*>(lkeys + kTopK * tid + t) = make_ulonglong2(kk[t], kk[t + 1]);
          }
          int m_l = -1, bd_l = 256, sd_l = 256, acc_l = -1;                      // this lane's query: match, distances, keypoint accepted at its turn
          const int cnt = min(64, nq - q0);
          const int tl = tid < kTopK ? tid : 0;
          unsigned long long key_n = lkeys[tl];                                 // query 0 of the chunk; the next one is always in flight
          for (int i = 0; i < cnt; i++) {
            unsigned long long key = tid < kTopK ? key_n : kNone;
            const bool full = __builtin_amdgcn_readlane((int)(key_n >> 32), kTopK - 1) != -1 || __builtin_amdgcn_readlane((int)key_n, kTopK - 1) != -1;
            key_n = lkeys[kTopK * min(i + 1, 63) + tl];
            const int flags = __builtin_amdgcn_readlane(fl, i);
            if (!(flags & 1)) continue;
            const int q = q0 + i;
            const int kp = key != kNone ? dec(key) : 0;
            const int vmd = blk[kp], hold = holder[kp];                         // vMatchedDistance and vnMatches21 of this lane's candidate
            const bool free_ = key != kNone && vmd > (int)(key >> 32);
            const unsigned long long fm = __ballot(free_) & ((1ull << kTopK) - 1);
            unsigned long long b1 = kNone, b2 = kNone;
            int prev = -1;
            if (__popcll(fm) < 2 && full) {
              rounds++;                                                         // reported: 1 + the number of rescans
              wave_rescan(q, b1, b2);
              if (b1 != kNone) prev = holder[dec(b1)];
            } else {
              const unsigned long long fm2 = fm & (fm - 1);
              if (fm) { const int l1 = __ffsll((long long)fm) - 1; b1 = rl64(key, l1); prev = __builtin_amdgcn_readlane(hold, l1); }
              if (fm2) b2 = rl64(key, __ffsll((long long)fm2) - 1);
            }
            int m = -1, bd = 256, sd = 256;
            if (b1 != kNone) {
              bd = (int)(b1 >> 32);
              const int bi = dec(b1);
              bool ok = bd <= P.accept_max;
              if (b2 != kNone) sd = (int)(b2 >> 32);
              if (ok && P.ratio_mode == 1) ok = (float)bd < __fmul_rn(P.nnratio, (float)sd);                      // ORBmatcher.cc:456
              if (ok) m = bi;
            }
            if (m >= 0) {                                                       // uniform: every lane holds the same m
              if (prev >= q0) { if (tid == prev - q0) m_l = -1; }               // :460-461, the holder is in this chunk
              else if (prev >= 0 && tid == 0) P.match[prev] = -1;
              if (tid == 0) { holder[m] = q; blk[m] = bd; }
            }
            if (tid == i) { m_l = m; bd_l = bd; sd_l = sd; acc_l = m; }
          }
          if (ql < nq) {
            P.match[ql] = m_l; P.best_dist[ql] = bd_l; P.second_dist[ql] = sd_l;
            P.cache[kTopK * (size_t)ql] = (unsigned long long)(long long)acc_l;   // the list is spent: its first word records the acceptance
          }
        }
      }
      __syncthreads();
      if (P.check_orientation) {                                                // rotHist of every acceptance (:468-476)
        for (int q = tid; q < nq; q += kThreads) {
          const int acc = (int)(long long)P.cache[kTopK * (size_t)q];
          if (acc < 0 || !(P.q[q].flags & 1)) continue;
          float rot = __fsub_rn(P.q[q].angle, P.t_angle[acc]);
          if (rot < 0.f) rot = __fadd_rn(rot, 360.0f);
          int bin = (int)roundf(__fmul_rn(rot, 1.0f / kHisto));
          if (bin == kHisto) bin = 0;
          atomicAdd(&hist[min(max(bin, 0), kHisto - 1)], 1);
        }
      }
      __syncthreads();
      break;
    }
    if (!P.sequential || !changed) break;
    if (tid == 0) ctl[0] = 0;
    for (int k = tid; k < nt; k += kThreads) blk[k] = 0x7fffffff;
    __syncthreads();
    for (int q = tid; q < nq; q += kThreads) {
      const int m = P.match[q];
      if (m >= 0 && (P.q[q].flags & 2)) atomicMin(&blk[m], q);
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- rotation histogram, owners, counts
  int* owner = blk;                                                             // blockers are dead now
  for (int k = tid; k < nt; k += kThreads) owner[k] = -1;
  __syncthreads();
  for (int q = tid; q < nq; q += kThreads) {
    const int m = P.match[q];
    int bin = 255;
    if (m >= 0) {
      atomicAdd(&ctl[1], 1);
      atomicMax(&owner[m], q);
      if (P.check_orientation) {                                                // ORBmatcher.cc:1431-1440
        float rot = __fsub_rn(P.q[q].angle, P.t_angle[m]);
        if (rot < 0.f) rot = __fadd_rn(rot, 360.0f);
        bin = (int)roundf(__fmul_rn(rot, 1.0f / kHisto));
        if (bin == kHisto) bin = 0;
        bin = min(max(bin, 0), kHisto - 1);                                     // the reference asserts the range
        if (P.sequential != 2) atomicAdd(&hist[bin], 1);                        // (steal mode counted every acceptance in its pass)
      }
    }
    P.removed[q] = (uint8_t)bin;
  }
  __syncthreads();
  if (tid == 0) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    if (P.check_orientation) {                                                  // ComputeThreeMaxima, ORBmatcher.cc:1601-1642
      int max1 = 0, max2 = 0, max3 = 0;
      for (int i = 0; i < kHisto; i++) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
      }
      if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
      else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
    }
    ctl[3] = ind1; ctl[4] = ind2; ctl[5] = ind3;
  }
  __syncthreads();
  for (int q = tid; q < nq; q += kThreads) {
    const int bin = P.removed[q];
    uint8_t rem = 0;
    if (P.check_orientation && bin != 255 && bin != ctl[3] && bin != ctl[4] && bin != ctl[5]) {
      rem = 1;
      atomicAdd(&ctl[2], 1);
      owner[P.match[q]] = -2;                                                   // slot NULLed, ORBmatcher.cc:1452-1460
    }
    P.removed[q] = rem;
  }
  __syncthreads();
  if (P.want_owner) for (int k = tid; k < nt; k += kThreads) P.owner[k] = owner[k];
  if (tid == 0) { P.summary[0] = ctl[1] - ctl[2]; P.summary[1] = rounds; }
  if (P.ap_has) {
    const int n_matches = ctl[1] - ctl[2];
    const bool take = n_matches >= P.ap_min_matches;
    if (tid == 0) {
      if (!P.ap_is_retry) { P.ap_counts[0] = n_matches; if (take) { P.ap_counts[1] = n_matches; P.ap_counts[2] = 0; } }
      else { P.ap_counts[1] = n_matches; P.ap_counts[2] = 1; }
    }
    if (take) {
      for (int k = tid; k < nt; k += kThreads) {
        const int q = owner[k];
        if (q < 0) continue;
        P.ap_has[k] = 1; P.ap_id[k] = P.ap_q_id[q]; P.ap_obs[k] = P.ap_q_obs ? P.ap_q_obs[q] : 1;
        P.ap_world[3 * k] = P.ap_q_pos[3 * q]; P.ap_world[3 * k + 1] = P.ap_q_pos[3 * q + 1]; P.ap_world[3 * k + 2] = P.ap_q_pos[3 * q + 2];
      }
    }
  }
}

// Projection loop of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:841-890), one lane per MapPoint.
struct FuseArgs {
  lld_frame_view V;
  int n;
  const float* pos; const float* nrm; const float* maxd; const float* mind; const uint8_t* skip;
  float scale[LLD_ORB_MAX_LEVELS];
  float th;
  QRec* q; float* uvr;
};

__global__ __launch_bounds__(256) void project_fuse_kernel(FuseArgs F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n) return;
  QRec Q; memset(&Q, 0, sizeof(Q));
  Q.level_min = -1; Q.level_max = -1;
  float u = 0.f, v = 0.f, ur = 0.f;
  do {
    if (F.skip && F.skip[i]) break;
    const float P[3] = {F.pos[3 * i], F.pos[3 * i + 1], F.pos[3 * i + 2]};
    float Pc[3]; cv_transform(F.V, P, Pc);
    if (Pc[2] < 0.0f) break;
    const float invz = __fdiv_rn(1.0f, Pc[2]);
    const float x = __fmul_rn(Pc[0], invz), y = __fmul_rn(Pc[1], invz);
    u = __fadd_rn(__fmul_rn(F.V.fx, x), F.V.cx);
    v = __fadd_rn(__fmul_rn(F.V.fy, y), F.V.cy);
    if (!(u >= F.V.min_x && u < F.V.max_x && v >= F.V.min_y && v < F.V.max_y)) break;      // KeyFrame::IsInImage
    ur = __fsub_rn(u, __fmul_rn(F.V.bf, invz));
    const float maxDistance = __fmul_rn(1.2f, F.maxd[i]), minDistance = __fmul_rn(0.8f, F.mind[i]);
    const float PO[3] = {__fsub_rn(P[0], F.V.Ow[0]), __fsub_rn(P[1], F.V.Ow[1]), __fsub_rn(P[2], F.V.Ow[2])};
    const float dist3D = cv_norm3(PO);
    if (dist3D < minDistance || dist3D > maxDistance) break;
    if (cv_dot3(PO, F.nrm + 3 * i) < __dmul_rn(0.5, (double)dist3D)) break;                   // PO.dot(Pn)<0.5*dist3D
    const int lvl = predict_scale(F.maxd[i], dist3D, F.V);
    Q.u = u; Q.v = v; Q.ur = ur; Q.radius = __fmul_rn(F.th, F.scale[lvl]);
    Q.level_min = lvl - 1; Q.level_max = lvl;
    Q.flags = 1;
  } while (false);
  F.q[i] = Q;
  if (F.uvr) { F.uvr[3 * i] = u; F.uvr[3 * i + 1] = v; F.uvr[3 * i + 2] = ur; }
}

// Projection loops of the relocalisation / loop-closing matchers (lld_orb_search_projected), one lane per MapPoint; `routine` selects
// the reference's sequence of tests and its float arithmetic (see include/lld_amd.h).
struct ProjGenArgs {
  lld_frame_view V;
  int n, routine;
  float sR[9], t2[3];
  const float* pos; const float* nrm; const float* maxd; const float* mind; const uint8_t* skip; const float* angle;
  float scale[LLD_ORB_MAX_LEVELS];
  float th;
  QRec* q; float* uv; int32_t* level;
};

__global__ __launch_bounds__(256) void project_general_kernel(ProjGenArgs F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.n) return;
  QRec Q; memset(&Q, 0, sizeof(Q));
  Q.level_min = -1; Q.level_max = -1;
  Q.angle = F.angle ? F.angle[i] : 0.f;
  float u = 0.f, v = 0.f; int lvl = 0;
  do {
    if (F.skip && F.skip[i]) break;
    const float P[3] = {F.pos[3 * i], F.pos[3 * i + 1], F.pos[3 * i + 2]};
    float Pc[3]; cv_transform(F.V, P, Pc);                                       // Rcw*p3Dw+tcw: one cv::gemm
    if (F.routine == LLD_ORB_PROJ_SIM3_DIR) {                                    // p3Dc2 = sR21*p3Dc1 + t21: a second cv::gemm
      float P2[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
        P2[r] = (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)F.sR[3 * r], (double)Pc[0]), __dmul_rn((double)F.sR[3 * r + 1], (double)Pc[1])),
                                           __dmul_rn((double)F.sR[3 * r + 2], (double)Pc[2])), (double)F.t2[r]);
      Pc[0] = P2[0]; Pc[1] = P2[1]; Pc[2] = P2[2];
    }
    float dist;
    if (F.routine == LLD_ORB_PROJ_RELOC) {
      const float invzc = (float)__ddiv_rn(1.0, (double)Pc[2]);                 // const float invzc = 1.0/x3Dc.at<float>(2);  (no depth test, :1499-1503)
      u = __fadd_rn(__fmul_rn(__fmul_rn(F.V.fx, Pc[0]), invzc), F.V.cx);
      v = __fadd_rn(__fmul_rn(__fmul_rn(F.V.fy, Pc[1]), invzc), F.V.cy);
      if (u < F.V.min_x || u > F.V.max_x) break;
      if (v < F.V.min_y || v > F.V.max_y) break;
    } else {
      if (Pc[2] < 0.0f) break;
      // `1/z` is a float division in :328, `1.0/z` a double one rounded to float in :1019, :1168, :1246
      const float invz = F.routine == LLD_ORB_PROJ_KF_SIM3 ? __fdiv_rn(1.0f, Pc[2]) : (float)__ddiv_rn(1.0, (double)Pc[2]);
      const float x = __fmul_rn(Pc[0], invz), y = __fmul_rn(Pc[1], invz);
      u = __fadd_rn(__fmul_rn(F.V.fx, x), F.V.cx);
      v = __fadd_rn(__fmul_rn(F.V.fy, y), F.V.cy);
      if (!(u >= F.V.min_x && u < F.V.max_x && v >= F.V.min_y && v < F.V.max_y)) break;      // KeyFrame::IsInImage
    }
    const float maxDistance = __fmul_rn(1.2f, F.maxd[i]), minDistance = __fmul_rn(0.8f, F.mind[i]);
    if (F.routine == LLD_ORB_PROJ_SIM3_DIR) dist = cv_norm3(Pc);                  // cv::norm(p3Dc2)
    else {
      const float PO[3] = {__fsub_rn(P[0], F.V.Ow[0]), __fsub_rn(P[1], F.V.Ow[1]), __fsub_rn(P[2], F.V.Ow[2])};
      dist = cv_norm3(PO);
      if (dist < minDistance || dist > maxDistance) break;
      if (F.routine != LLD_ORB_PROJ_RELOC && cv_dot3(PO, F.nrm + 3 * i) < __dmul_rn(0.5, (double)dist)) break;   // PO.dot(Pn)<0.5*dist
    }
    if (dist < minDistance || dist > maxDistance) break;
    lvl = predict_scale(F.maxd[i], dist, F.V);
    Q.u = u; Q.v = v; Q.radius = __fmul_rn(F.th, F.scale[lvl]);
    Q.level_min = lvl - 1; Q.level_max = F.routine == LLD_ORB_PROJ_RELOC ? lvl + 1 : lvl;
    Q.flags = 1 | 2;                                                              // valid; a match blocks the keypoint for later points (routines with occupancy)
  } while (false);
  F.q[i] = Q;
  if (F.uv) { F.uv[2 * i] = u; F.uv[2 * i + 1] = v; }
  if (F.level) F.level[i] = lvl;
}

constexpr size_t kLdsLimit = 160 * 1024 - 512;
constexpr int kRowBuckets = 512;           // ROWS mode: one bucket per (octave, image row); rows beyond are clamped into the last bucket of the octave

size_t lds_bytes(int nt, int n_cells, bool grid, bool desc) {
  return (size_t)nt * sizeof(TKey) + (desc ? (size_t)nt * 32 : 0) + (size_t)nt * 4 + (size_t)(n_cells + 1) * 4 + 32 * 4 + 8 * 4 + kThreads * 4 +
         48 * 4 + (size_t)nt * 4 + 64 * 8 * 8 + 32;
}

int validate(const lld_orb_search* s, const lld_orb_search_result* out) {
  if (!s || !out) return LLD_ERR_INVALID;
  const int nt = s->nt, nq = s->nq;
  if (nt < 0 || nq < 0) return LLD_ERR_INVALID;
  if (nt > LLD_ORB_MAX_KEYPOINTS) return LLD_ERR_UNSUPPORTED;
  if (!out->match || !out->best_dist || !out->second_dist || !out->removed) return LLD_ERR_INVALID;
  if (nq > 0 && !s->q_desc) return LLD_ERR_INVALID;
  if (nt > 0 && (!s->t_desc || !s->t_xy || !s->t_octave)) return LLD_ERR_INVALID;
  if (s->candidates < LLD_ORB_CAND_ALL || s->candidates > LLD_ORB_CAND_ROWS) return LLD_ERR_INVALID;
  if (s->ratio_mode < 0 || s->ratio_mode > 2) return LLD_ERR_INVALID;
  if (s->sequential < 0 || s->sequential > 2) return LLD_ERR_INVALID;
  if (s->sequential == 2 && (s->tie_last || s->ratio_mode == 2 || (s->candidates != LLD_ORB_CAND_GRID && s->candidates != LLD_ORB_CAND_ALL))) return LLD_ERR_INVALID;
  const bool grid = s->candidates == LLD_ORB_CAND_GRID;
  if (grid && (!s->q_uv || !s->q_radius || s->grid_cols <= 0 || s->grid_rows <= 0 || s->grid_cols * s->grid_rows > 8191)) return LLD_ERR_INVALID;
  if (s->candidates == LLD_ORB_CAND_CSR && (!s->cand_range || s->n_cand < 0 || (s->n_cand > 0 && !s->cand_idx))) return LLD_ERR_INVALID;
  if (s->candidates == LLD_ORB_CAND_ROWS && (!s->q_uv || !s->level_scale)) return LLD_ERR_INVALID;
  if ((s->gates & LLD_ORB_GATE_LEVEL) && (!s->q_level_min || !s->q_level_max)) return LLD_ERR_INVALID;
  if ((s->gates & LLD_ORB_GATE_STEREO) && (!s->q_uright || !s->q_stereo_radius)) return LLD_ERR_INVALID;
  if ((s->gates & LLD_ORB_GATE_CHI2) && (!s->q_uv || !s->q_uright || !s->level_inv_sigma2)) return LLD_ERR_INVALID;
  if ((s->gates & LLD_ORB_GATE_EPIPOLAR) && (!s->q_epiline || !s->q_stereo || !s->level_scale || !s->level_sigma2)) return LLD_ERR_INVALID;
  if (s->check_orientation && (!s->q_angle || (nt > 0 && !s->t_angle))) return LLD_ERR_INVALID;
  if (s->n_levels < 0 || s->n_levels > LLD_ORB_MAX_LEVELS) return LLD_ERR_UNSUPPORTED;
  for (int k = 0; k < nt; k++) if (s->t_octave[k] < 0 || s->t_octave[k] >= LLD_ORB_MAX_LEVELS) return LLD_ERR_INVALID;
  if (s->candidates == LLD_ORB_CAND_CSR) {
    for (int q = 0; q < nq; q++)
      if (s->cand_range[2 * q] < 0 || s->cand_range[2 * q + 1] < s->cand_range[2 * q] || s->cand_range[2 * q + 1] > s->n_cand) return LLD_ERR_INVALID;
    for (int p = 0; p < s->n_cand; p++) if (s->cand_idx[p] < 0 || s->cand_idx[p] >= nt) return LLD_ERR_INVALID;
  }
  return LLD_OK;
}

inline size_t al(size_t b) { return (b + 63) & ~size_t(63); }

// Byte layout of one problem inside the packed input / output regions (identical in pinned host memory and in HBM).
struct Layout {
  size_t q, q_desc, t_desc, t_xy, t_oct, t_ur, t_ang, t_occ, cand, in_end;      // offsets from the start of the input region
  size_t match, bd, sd, owner, sum, rem, out_end;                               // offsets from the start of the output region
};

}  // namespace

extern "C" int lld_orb_search_batch(lld_ctx* ctx, int n, const lld_orb_search* problems, lld_orb_search_result* outs) {
  if (!ctx || n < 0 || (n > 0 && (!problems || !outs))) return LLD_ERR_INVALID;
  for (int i = 0; i < n; i++) { const int st = validate(&problems[i], &outs[i]); if (st) return st; }
  if (n == 0) return LLD_OK;
  LLD_HIP_TRY(hipSetDevice(ctx->device));

  // ---- lay the batch out: [Problem x n | per-problem inputs ...] and [per-problem outputs ...]
  std::vector<Layout> lay((size_t)n);
  size_t in_off = al(sizeof(Problem) * (size_t)n), out_off = 0, lds_max = 0;
  for (int i = 0; i < n; i++) {
    const lld_orb_search& s = problems[i]; Layout& L = lay[i];
    const size_t nt = s.nt, nq = s.nq, nc = (s.candidates == LLD_ORB_CAND_CSR) ? s.n_cand : 0;
    L.q = in_off; in_off += al(nq * sizeof(QRec));
    L.q_desc = in_off; in_off += al(nq * 32);
    L.t_desc = in_off; in_off += al(nt * 32);
    L.t_xy = in_off; in_off += al(nt * 8);
    L.t_oct = in_off; in_off += al(nt * 4);
    L.t_ur = in_off; in_off += s.t_uright ? al(nt * 4) : 0;
    L.t_ang = in_off; in_off += s.t_angle ? al(nt * 4) : 0;
    L.t_occ = in_off; in_off += s.t_occupied ? al(nt) : 0;
    L.cand = in_off; in_off += al(nc * 4);
    L.in_end = in_off;
    L.match = out_off; out_off += al(nq * 4);
    L.bd = out_off; out_off += al(nq * 4);
    L.sd = out_off; out_off += al(nq * 4);
    L.owner = out_off; out_off += al(nt * 4);
    L.sum = out_off; out_off += al(16);
    L.rem = out_off; out_off += al(nq);
    L.out_end = out_off;
  }
  const size_t in_bytes = in_off, out_bytes = out_off;
  // device-only scratch behind the output region: the round-1 candidate cache of the sequential problems
  std::vector<size_t> cache_at((size_t)n);
  size_t cache_bytes = 0;
  for (int i = 0; i < n; i++) { cache_at[i] = cache_bytes; if (problems[i].sequential) cache_bytes += al((size_t)problems[i].nq * 8 * kTopK); }
  void* hbase; int st = lld_ctx_pinned(ctx, in_bytes + out_bytes, &hbase); if (st) return st;
  void* dbase; st = lld_ctx_scratch(ctx, in_bytes + out_bytes + cache_bytes + 256, &dbase); if (st) return st;
  char* h = (char*)hbase; char* d = (char*)dbase;
  char* d_out = d + in_bytes; char* h_out = h + in_bytes;
  char* d_cache = d_out + out_bytes;

  for (int i = 0; i < n; i++) {
    const lld_orb_search& s = problems[i]; const Layout& L = lay[i];
    const int nt = s.nt, nq = s.nq;
    const bool grid = s.candidates == LLD_ORB_CAND_GRID;
    QRec* qr = reinterpret_cast<QRec*>(h + L.q);
    for (int q = 0; q < nq; q++) {
      QRec& Q = qr[q]; std::memset(&Q, 0, sizeof(Q));
      if (s.q_uv) { Q.u = s.q_uv[2 * q]; Q.v = s.q_uv[2 * q + 1]; }
      if (s.q_radius) Q.radius = s.q_radius[q];
      if (s.q_uright) Q.ur = s.q_uright[q];
      if (s.q_stereo_radius) Q.stereo_radius = s.q_stereo_radius[q];
      if (s.q_angle) Q.angle = s.q_angle[q];
      if (s.q_epiline) { Q.ea = s.q_epiline[3 * q]; Q.eb = s.q_epiline[3 * q + 1]; Q.ec = s.q_epiline[3 * q + 2]; }
      Q.level_min = s.q_level_min ? s.q_level_min[q] : -1;
      Q.level_max = s.q_level_max ? s.q_level_max[q] : -1;
      Q.flags = ((!s.q_valid || s.q_valid[q]) ? 1 : 0) | ((!s.q_blocks || s.q_blocks[q]) ? 2 : 0) | ((s.q_stereo && s.q_stereo[q]) ? 4 : 0);
      if (s.candidates == LLD_ORB_CAND_CSR) { Q.cs = s.cand_range[2 * q]; Q.ce = s.cand_range[2 * q + 1]; }
    }
    if (nq) std::memcpy(h + L.q_desc, s.q_desc, (size_t)nq * 32);
    if (nt) {
      std::memcpy(h + L.t_desc, s.t_desc, (size_t)nt * 32);
      std::memcpy(h + L.t_xy, s.t_xy, (size_t)nt * 8);
      std::memcpy(h + L.t_oct, s.t_octave, (size_t)nt * 4);
      if (s.t_uright) std::memcpy(h + L.t_ur, s.t_uright, (size_t)nt * 4);
      if (s.t_angle) std::memcpy(h + L.t_ang, s.t_angle, (size_t)nt * 4);
      if (s.t_occupied) std::memcpy(h + L.t_occ, s.t_occupied, (size_t)nt);
    }
    if (s.candidates == LLD_ORB_CAND_CSR && s.n_cand) std::memcpy(h + L.cand, s.cand_idx, (size_t)s.n_cand * 4);

    Problem& P = reinterpret_cast<Problem*>(h)[i]; std::memset(&P, 0, sizeof(P));
    P.nt = nt; P.nq = nq;
    P.t_desc = reinterpret_cast<const uint32_t*>(d + L.t_desc); P.t_xy = reinterpret_cast<const float*>(d + L.t_xy);
    P.t_octave = reinterpret_cast<const int32_t*>(d + L.t_oct);
    P.t_uright = s.t_uright ? reinterpret_cast<const float*>(d + L.t_ur) : nullptr;
    P.t_angle = s.t_angle ? reinterpret_cast<const float*>(d + L.t_ang) : nullptr;
    P.t_occupied = s.t_occupied ? reinterpret_cast<const uint8_t*>(d + L.t_occ) : nullptr;
    P.q_desc = reinterpret_cast<const uint32_t*>(d + L.q_desc); P.q = reinterpret_cast<const QRec*>(d + L.q);
    P.cand_idx = reinterpret_cast<const int32_t*>(d + L.cand);
    P.min_x = s.grid_min_x; P.min_y = s.grid_min_y; P.winv = s.grid_width_inv; P.hinv = s.grid_height_inv;
    const int n_lv = std::max(1, s.n_levels);
    P.cols = grid ? s.grid_cols : (s.candidates == LLD_ORB_CAND_ROWS ? n_lv : 1);
    P.rows = grid ? s.grid_rows : (s.candidates == LLD_ORB_CAND_ROWS ? std::min(kRowBuckets, 8190 / n_lv) : 1);
    P.n_levels = s.n_levels;
    for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) {
      P.scale[l] = (s.level_scale && l < s.n_levels) ? s.level_scale[l] : 1.f;
      P.sigma2[l] = (s.level_sigma2 && l < s.n_levels) ? s.level_sigma2[l] : 1.f;
      P.inv_sigma2[l] = (s.level_inv_sigma2 && l < s.n_levels) ? s.level_inv_sigma2[l] : 1.f;
    }
    P.disp_min = s.disp_min; P.disp_max = s.disp_max; P.epi_x = s.epipole_x; P.epi_y = s.epipole_y; P.only_stereo = s.only_stereo;
    P.candidates = s.candidates; P.gates = s.gates; P.tie_last = s.tie_last; P.accept_max = s.accept_max; P.ratio_mode = s.ratio_mode;
    P.nnratio = s.nnratio; P.sequential = s.sequential; P.check_orientation = s.check_orientation;
    P.match = reinterpret_cast<int32_t*>(d_out + L.match); P.best_dist = reinterpret_cast<int32_t*>(d_out + L.bd);
    P.second_dist = reinterpret_cast<int32_t*>(d_out + L.sd); P.removed = reinterpret_cast<uint8_t*>(d_out + L.rem);
    P.owner = reinterpret_cast<int32_t*>(d_out + L.owner); P.summary = reinterpret_cast<int32_t*>(d_out + L.sum);
    P.want_owner = outs[i].owner != nullptr;
    P.cache = reinterpret_cast<unsigned long long*>(d_cache + cache_at[i]);
    const bool bucketed = grid || s.candidates == LLD_ORB_CAND_ROWS;
    P.desc_in_lds = lds_bytes(nt, P.cols * P.rows, bucketed, true) <= kLdsLimit;
    lds_max = std::max(lds_max, lds_bytes(nt, P.cols * P.rows, bucketed, P.desc_in_lds != 0));
  }

  hipStream_t sm = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, sm));
  if (!ctx->orb_lds_raised) { LLD_HIP_TRY(hipFuncSetAttribute((const void*)orb_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit)); ctx->orb_lds_raised = true; }
  hipLaunchKernelGGL(orb_search_kernel, dim3(n), dim3(kThreads), lds_max, sm, reinterpret_cast<const Problem*>(d));
  LLD_HIP_TRY(hipGetLastError());
  LLD_HIP_TRY(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, sm));
  LLD_HIP_TRY(hipStreamSynchronize(sm));

  for (int i = 0; i < n; i++) {
    const Layout& L = lay[i]; lld_orb_search_result& o = outs[i];
    const size_t nt = problems[i].nt, nq = problems[i].nq;
    if (nq) {
      std::memcpy(o.match, h_out + L.match, nq * 4); std::memcpy(o.best_dist, h_out + L.bd, nq * 4);
      std::memcpy(o.second_dist, h_out + L.sd, nq * 4); std::memcpy(o.removed, h_out + L.rem, nq);
    }
    if (o.owner && nt) std::memcpy(o.owner, h_out + L.owner, nt * 4);
    const int32_t* sum = reinterpret_cast<const int32_t*>(h_out + L.sum);
    o.n_matches = sum[0]; o.rounds = sum[1];
  }
  return LLD_OK;
}

extern "C" int lld_orb_search_run(lld_ctx* ctx, const lld_orb_search* s, lld_orb_search_result* out) {
  return lld_orb_search_batch(ctx, 1, s, out);
}

namespace {

}  // namespace

// (struct lld_frame: lld_track_internal.h)

namespace {

// Shared plumbing of the "project on the device, then search" entry points: one packed input region
// [Problem | QRec[nq] (written by the projection kernel) | q_desc | frame keypoints | caller inputs] and one output region.
struct ProjSearch {
  lld_ctx* ctx; const lld_orb_search* frame; int nt, nq; bool need_angle;
  const lld_frame* resident = nullptr;     // the keypoint side is already on the device (lld_frame_*): only t_occupied travels
  size_t in = 0, out = 0;
  size_t o_q = 0, o_qd = 0, o_td = 0, o_txy = 0, o_toct = 0, o_tur = 0, o_tang = 0, o_tocc = 0;
  size_t r_match = 0, r_bd = 0, r_sd = 0, r_owner = 0, r_sum = 0, r_rem = 0;
  char *h = nullptr, *d = nullptr, *h_out = nullptr, *d_out = nullptr;
  size_t add_in(size_t bytes) { const size_t o = in; in += al(bytes); return o; }
  size_t add_out(size_t bytes) { const size_t o = out; out += al(bytes); return o; }

  int check(const lld_orb_search_result* res) const {
    if (nt < 0 || nq < 0) return LLD_ERR_INVALID;
    if (nt > LLD_ORB_MAX_KEYPOINTS) return LLD_ERR_UNSUPPORTED;
    if (!res->match || !res->best_dist || !res->second_dist || !res->removed) return LLD_ERR_INVALID;
    if (resident) { if (need_angle && nt > 0 && !resident->has_angle) return LLD_ERR_INVALID; }
    else if (nt > 0 && (!frame->t_desc || !frame->t_xy || !frame->t_octave || (need_angle && !frame->t_angle))) return LLD_ERR_INVALID;
    if (!frame->level_scale || frame->n_levels <= 0 || frame->n_levels > LLD_ORB_MAX_LEVELS) return LLD_ERR_INVALID;
    if (frame->grid_cols <= 0 || frame->grid_rows <= 0 || frame->grid_cols * frame->grid_rows > 8191) return LLD_ERR_INVALID;
    if (!resident) for (int k = 0; k < nt; k++) if (frame->t_octave[k] < 0 || frame->t_octave[k] >= LLD_ORB_MAX_LEVELS) return LLD_ERR_INVALID;
    return LLD_OK;
  }
  void layout() {
    in = al(sizeof(Problem));
    o_q = add_in((size_t)nq * sizeof(QRec)); o_qd = add_in((size_t)nq * 32);
    if (!resident) {
      o_td = add_in((size_t)nt * 32); o_txy = add_in((size_t)nt * 8); o_toct = add_in((size_t)nt * 4);
      o_tur = frame->t_uright ? add_in((size_t)nt * 4) : 0; o_tang = need_angle ? add_in((size_t)nt * 4) : 0;
    }
    o_tocc = frame->t_occupied ? add_in((size_t)nt) : 0;
    r_match = add_out((size_t)nq * 4); r_bd = add_out((size_t)nq * 4); r_sd = add_out((size_t)nq * 4); r_owner = add_out((size_t)nt * 4);
    r_sum = add_out(16); r_rem = add_out((size_t)nq);
  }
  int alloc() {
    void* hb; int st = lld_ctx_pinned(ctx, in + out, &hb); if (st) return st;
    void* db; st = lld_ctx_scratch(ctx, in + out + al((size_t)nq * 8 * kTopK) + 256, &db); if (st) return st;   // + the round-1 candidate cache
    h = (char*)hb; d = (char*)db; h_out = h + in; d_out = d + in;
    return LLD_OK;
  }
  // frame keypoints + query descriptors into the staging buffer, Problem with everything but the matching rules
  Problem& pack(const uint32_t* q_desc, bool want_owner) {
    if (nq) std::memcpy(h + o_qd, q_desc, (size_t)nq * 32);
    if (nt && !resident) {
      std::memcpy(h + o_td, frame->t_desc, (size_t)nt * 32); std::memcpy(h + o_txy, frame->t_xy, (size_t)nt * 8);
      std::memcpy(h + o_toct, frame->t_octave, (size_t)nt * 4);
      if (frame->t_uright) std::memcpy(h + o_tur, frame->t_uright, (size_t)nt * 4);
      if (need_angle) std::memcpy(h + o_tang, frame->t_angle, (size_t)nt * 4);
    }
    if (nt && frame->t_occupied) std::memcpy(h + o_tocc, frame->t_occupied, (size_t)nt);
    Problem& P = *reinterpret_cast<Problem*>(h); std::memset(&P, 0, sizeof(P));
    P.nt = nt; P.nq = nq;
    if (resident) {
      P.t_desc = reinterpret_cast<const uint32_t*>(resident->d + resident->o_td); P.t_xy = reinterpret_cast<const float*>(resident->d + resident->o_txy);
      P.t_octave = reinterpret_cast<const int32_t*>(resident->d + resident->o_toct);
      P.t_uright = resident->has_uright ? reinterpret_cast<const float*>(resident->d + resident->o_tur) : nullptr;
      P.t_angle = (need_angle && resident->has_angle) ? reinterpret_cast<const float*>(resident->d + resident->o_tang) : nullptr;
    } else {
      P.t_desc = reinterpret_cast<const uint32_t*>(d + o_td); P.t_xy = reinterpret_cast<const float*>(d + o_txy);
      P.t_octave = reinterpret_cast<const int32_t*>(d + o_toct);
      P.t_uright = frame->t_uright ? reinterpret_cast<const float*>(d + o_tur) : nullptr;
      P.t_angle = need_angle ? reinterpret_cast<const float*>(d + o_tang) : nullptr;
    }
    P.t_occupied = frame->t_occupied ? reinterpret_cast<const uint8_t*>(d + o_tocc) : nullptr;
    P.q_desc = reinterpret_cast<const uint32_t*>(d + o_qd); P.q = reinterpret_cast<const QRec*>(d + o_q);
    P.min_x = frame->grid_min_x; P.min_y = frame->grid_min_y; P.winv = frame->grid_width_inv; P.hinv = frame->grid_height_inv;
    P.cols = frame->grid_cols; P.rows = frame->grid_rows; P.n_levels = frame->n_levels;
    for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) { P.scale[l] = l < frame->n_levels ? frame->level_scale[l] : 1.f; P.sigma2[l] = 1.f; P.inv_sigma2[l] = 1.f; }
    P.candidates = LLD_ORB_CAND_GRID;
    P.match = reinterpret_cast<int32_t*>(d_out + r_match); P.best_dist = reinterpret_cast<int32_t*>(d_out + r_bd);
    P.second_dist = reinterpret_cast<int32_t*>(d_out + r_sd); P.removed = reinterpret_cast<uint8_t*>(d_out + r_rem);
    P.owner = reinterpret_cast<int32_t*>(d_out + r_owner); P.summary = reinterpret_cast<int32_t*>(d_out + r_sum);
    P.want_owner = want_owner;
    P.cache = reinterpret_cast<unsigned long long*>(d_out + out);
    P.desc_in_lds = lds_bytes(nt, P.cols * P.rows, true, true) <= kLdsLimit;
    return P;
  }
  int upload() { LLD_HIP_TRY(hipMemcpyAsync(d, h, in, hipMemcpyHostToDevice, ctx->stream)); return LLD_OK; }
  int search_and_fetch(lld_orb_search_result* res) {
    const Problem& P = *reinterpret_cast<const Problem*>(h);
    const size_t lds = lds_bytes(nt, P.cols * P.rows, true, P.desc_in_lds != 0);
    if (!ctx->orb_lds_raised) { LLD_HIP_TRY(hipFuncSetAttribute((const void*)orb_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit)); ctx->orb_lds_raised = true; }
    hipLaunchKernelGGL(orb_search_kernel, dim3(1), dim3(kThreads), lds, ctx->stream, reinterpret_cast<const Problem*>(d));
    LLD_HIP_TRY(hipGetLastError());
    LLD_HIP_TRY(hipMemcpyAsync(h_out, d_out, out, hipMemcpyDeviceToHost, ctx->stream));
    LLD_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (nq) {
      std::memcpy(res->match, h_out + r_match, (size_t)nq * 4); std::memcpy(res->best_dist, h_out + r_bd, (size_t)nq * 4);
      std::memcpy(res->second_dist, h_out + r_sd, (size_t)nq * 4); std::memcpy(res->removed, h_out + r_rem, (size_t)nq);
    }
    if (res->owner && nt) std::memcpy(res->owner, h_out + r_owner, (size_t)nt * 4);
    const int32_t* sum = reinterpret_cast<const int32_t*>(h_out + r_sum);
    res->n_matches = sum[0]; res->rounds = sum[1];
    return LLD_OK;
  }
};

}  // namespace

static int local_points_impl(lld_ctx* ctx, const lld_orb_search* frame, const lld_frame* resident, const lld_frame_view* view, const lld_map_points* mp,
                             float viewing_cos_limit, float th, float nnratio, lld_frustum_result* fr, lld_orb_search_result* out) {
  if (!ctx || !frame || !view || !mp || !out) return LLD_ERR_INVALID;
  ProjSearch S{ctx, frame, frame->nt, mp->n, false, resident};
  int st = S.check(out); if (st) return st;
  const int nq = S.nq;
  if (nq > 0 && (!mp->world_pos || !mp->normal || !mp->max_distance || !mp->min_distance || !mp->desc)) return LLD_ERR_INVALID;
  if (view->n_levels != frame->n_levels) return LLD_ERR_INVALID;
  out->n_matches = 0; out->rounds = 0;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  S.layout();
  const size_t o_pos = S.add_in((size_t)nq * 12), o_nrm = S.add_in((size_t)nq * 12), o_maxd = S.add_in((size_t)nq * 4), o_mind = S.add_in((size_t)nq * 4);
  const size_t o_obs = mp->has_obs ? S.add_in((size_t)nq) : 0, o_skip = mp->skip ? S.add_in((size_t)nq) : 0;
  const size_t r_inv = S.add_out((size_t)nq), r_uvr = S.add_out((size_t)nq * 12), r_lvl = S.add_out((size_t)nq * 4), r_vc = S.add_out((size_t)nq * 4);
  st = S.alloc(); if (st) return st;
  if (nq) {
    std::memcpy(S.h + o_pos, mp->world_pos, (size_t)nq * 12); std::memcpy(S.h + o_nrm, mp->normal, (size_t)nq * 12);
    std::memcpy(S.h + o_maxd, mp->max_distance, (size_t)nq * 4); std::memcpy(S.h + o_mind, mp->min_distance, (size_t)nq * 4);
    if (mp->has_obs) std::memcpy(S.h + o_obs, mp->has_obs, (size_t)nq);
    if (mp->skip) std::memcpy(S.h + o_skip, mp->skip, (size_t)nq);
  }
  Problem& P = S.pack(mp->desc, out->owner != nullptr);
  P.gates = LLD_ORB_GATE_LEVEL | LLD_ORB_GATE_STEREO; P.accept_max = 100;       // TH_HIGH, src/ORBmatcher.cc:37,117
  P.ratio_mode = 2; P.nnratio = nnratio; P.sequential = 1;
  st = S.upload(); if (st) return st;
  if (nq) {
    FrustumArgs F; std::memset(&F, 0, sizeof(F));
    F.V = *view; F.n = nq;
    F.pos = reinterpret_cast<const float*>(S.d + o_pos); F.nrm = reinterpret_cast<const float*>(S.d + o_nrm);
    F.maxd = reinterpret_cast<const float*>(S.d + o_maxd); F.mind = reinterpret_cast<const float*>(S.d + o_mind);
    F.has_obs = mp->has_obs ? reinterpret_cast<const uint8_t*>(S.d + o_obs) : nullptr;
    F.skip = mp->skip ? reinterpret_cast<const uint8_t*>(S.d + o_skip) : nullptr;
    for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) F.scale[l] = P.scale[l];
    F.cos_limit = viewing_cos_limit; F.th = th;
    F.q = reinterpret_cast<QRec*>(S.d + S.o_q);
    F.in_view = reinterpret_cast<uint8_t*>(S.d_out + r_inv); F.uvr = reinterpret_cast<float*>(S.d_out + r_uvr);
    F.level = reinterpret_cast<int32_t*>(S.d_out + r_lvl); F.view_cos = reinterpret_cast<float*>(S.d_out + r_vc);
    hipLaunchKernelGGL(frustum_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, F);
    LLD_HIP_TRY(hipGetLastError());
  }
  st = S.search_and_fetch(out); if (st) return st;
  if (nq && fr) {
    if (fr->in_view) std::memcpy(fr->in_view, S.h_out + r_inv, (size_t)nq);
    if (fr->proj_uvr) std::memcpy(fr->proj_uvr, S.h_out + r_uvr, (size_t)nq * 12);
    if (fr->level) std::memcpy(fr->level, S.h_out + r_lvl, (size_t)nq * 4);
    if (fr->view_cos) std::memcpy(fr->view_cos, S.h_out + r_vc, (size_t)nq * 4);
  }
  return LLD_OK;
}

extern "C" int lld_orb_search_local_points(lld_ctx* ctx, const lld_orb_search* frame, const lld_frame_view* view, const lld_map_points* mp,
                                           float viewing_cos_limit, float th, float nnratio, lld_frustum_result* fr, lld_orb_search_result* out) {
  return local_points_impl(ctx, frame, nullptr, view, mp, viewing_cos_limit, th, nnratio, fr, out);
}

static int last_frame_impl(lld_ctx* ctx, const lld_orb_search* frame, const lld_frame* resident, const lld_frame_view* view, const lld_last_frame_points* last,
                           int direction, float th, int check_orientation, float* proj_uvr, lld_orb_search_result* out) {
  if (!ctx || !frame || !view || !last || !out) return LLD_ERR_INVALID;
  ProjSearch S{ctx, frame, frame->nt, last->n, check_orientation != 0, resident};
  int st = S.check(out); if (st) return st;
  const int nq = S.nq;
  if (nq > 0 && (!last->world_pos || !last->valid || !last->octave || !last->desc || (check_orientation && !last->angle))) return LLD_ERR_INVALID;
  for (int i = 0; i < nq; i++) if (last->octave[i] < 0 || last->octave[i] >= frame->n_levels) return LLD_ERR_INVALID;
  out->n_matches = 0; out->rounds = 0;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  S.layout();
  const size_t o_pos = S.add_in((size_t)nq * 12), o_val = S.add_in((size_t)nq), o_oct = S.add_in((size_t)nq * 4);
  const size_t o_ang = last->angle ? S.add_in((size_t)nq * 4) : 0, o_obs = last->has_obs ? S.add_in((size_t)nq) : 0;
  const size_t r_uvr = S.add_out((size_t)nq * 12);
  st = S.alloc(); if (st) return st;
  if (nq) {
    std::memcpy(S.h + o_pos, last->world_pos, (size_t)nq * 12); std::memcpy(S.h + o_val, last->valid, (size_t)nq);
    std::memcpy(S.h + o_oct, last->octave, (size_t)nq * 4);
    if (last->angle) std::memcpy(S.h + o_ang, last->angle, (size_t)nq * 4);
    if (last->has_obs) std::memcpy(S.h + o_obs, last->has_obs, (size_t)nq);
  }
  Problem& P = S.pack(last->desc, out->owner != nullptr);
  P.gates = LLD_ORB_GATE_LEVEL | LLD_ORB_GATE_STEREO; P.accept_max = 100;       // TH_HIGH, src/ORBmatcher.cc:1418
  P.ratio_mode = 0; P.sequential = 1; P.check_orientation = check_orientation != 0;
  st = S.upload(); if (st) return st;
  if (nq) {
    LastFrameArgs F; std::memset(&F, 0, sizeof(F));
    F.V = *view; F.n = nq; F.direction = direction;
    F.pos = reinterpret_cast<const float*>(S.d + o_pos); F.valid = reinterpret_cast<const uint8_t*>(S.d + o_val);
    F.octave = reinterpret_cast<const int32_t*>(S.d + o_oct);
    F.angle = last->angle ? reinterpret_cast<const float*>(S.d + o_ang) : nullptr;
    F.has_obs = last->has_obs ? reinterpret_cast<const uint8_t*>(S.d + o_obs) : nullptr;
    for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) F.scale[l] = P.scale[l];
    F.th = th;
    F.q = reinterpret_cast<QRec*>(S.d + S.o_q); F.uvr = reinterpret_cast<float*>(S.d_out + r_uvr);
    hipLaunchKernelGGL(project_last_frame_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, F);
    LLD_HIP_TRY(hipGetLastError());
  }
  st = S.search_and_fetch(out); if (st) return st;
  if (nq && proj_uvr) std::memcpy(proj_uvr, S.h_out + r_uvr, (size_t)nq * 12);
  return LLD_OK;
}

extern "C" int lld_orb_search_last_frame(lld_ctx* ctx, const lld_orb_search* frame, const lld_frame_view* view, const lld_last_frame_points* last,
                                         int direction, float th, int check_orientation, float* proj_uvr, lld_orb_search_result* out) {
  return last_frame_impl(ctx, frame, nullptr, view, last, direction, th, check_orientation, proj_uvr, out);
}

// ---- the resident frame (lld_frame_*)
extern "C" int lld_frame_create(lld_ctx* ctx, const lld_orb_search* kp, lld_frame** out) {
  if (!ctx || !kp || !out) return LLD_ERR_INVALID;
  *out = nullptr;
  const int nt = kp->nt;
  if (nt < 0) return LLD_ERR_INVALID;
  if (nt > LLD_ORB_MAX_KEYPOINTS) return LLD_ERR_UNSUPPORTED;
  if (nt > 0 && (!kp->t_desc || !kp->t_xy || !kp->t_octave)) return LLD_ERR_INVALID;
  if (!kp->level_scale || kp->n_levels <= 0 || kp->n_levels > LLD_ORB_MAX_LEVELS) return LLD_ERR_INVALID;
  if (kp->grid_cols <= 0 || kp->grid_rows <= 0 || kp->grid_cols * kp->grid_rows > 8191) return LLD_ERR_INVALID;
  for (int k = 0; k < nt; k++) if (kp->t_octave[k] < 0 || kp->t_octave[k] >= LLD_ORB_MAX_LEVELS) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  lld_frame* f = new lld_frame();
  f->ctx = ctx; f->nt = nt; f->has_uright = kp->t_uright != nullptr; f->has_angle = kp->t_angle != nullptr;
  f->consts = *kp;
  f->consts.t_desc = nullptr; f->consts.t_xy = nullptr; f->consts.t_octave = nullptr; f->consts.t_uright = nullptr; f->consts.t_angle = nullptr; f->consts.t_occupied = nullptr;
  f->consts.nq = 0; f->consts.q_desc = nullptr;
  for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) {
    f->scale[l] = l < kp->n_levels ? kp->level_scale[l] : 1.f;
    f->sigma2[l] = (l < kp->n_levels && kp->level_sigma2) ? kp->level_sigma2[l] : 1.f;
    f->inv_sigma2[l] = (l < kp->n_levels && kp->level_inv_sigma2) ? kp->level_inv_sigma2[l] : 1.f;
  }
  f->consts.level_scale = f->scale; f->consts.level_sigma2 = f->sigma2; f->consts.level_inv_sigma2 = f->inv_sigma2;
  size_t bytes = 0;
  auto add = [&](size_t b) { const size_t o = bytes; bytes += al(b); return o; };
  f->o_td = add((size_t)nt * 32); f->o_txy = add((size_t)nt * 8); f->o_toct = add((size_t)nt * 4);
  f->o_tur = f->has_uright ? add((size_t)nt * 4) : 0; f->o_tang = f->has_angle ? add((size_t)nt * 4) : 0;
  if (hipMalloc(reinterpret_cast<void**>(&f->d), bytes + 256) != hipSuccess) { delete f; return LLD_ERR_ALLOC; }
  void* hb = nullptr;
  int st = lld_ctx_pinned(ctx, bytes + 256, &hb);
  if (st) { (void)hipFree(f->d); delete f; return st; }
  char* h = static_cast<char*>(hb);
  if (nt) {
    std::memcpy(h + f->o_td, kp->t_desc, (size_t)nt * 32); std::memcpy(h + f->o_txy, kp->t_xy, (size_t)nt * 8); std::memcpy(h + f->o_toct, kp->t_octave, (size_t)nt * 4);
    if (f->has_uright) std::memcpy(h + f->o_tur, kp->t_uright, (size_t)nt * 4);
    if (f->has_angle) std::memcpy(h + f->o_tang, kp->t_angle, (size_t)nt * 4);
  }
  // (the context's pinned staging is reused by the next call on this context: the copy must have left it before this one returns)
  if (hipMemcpyAsync(f->d, h, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipFree(f->d); delete f; return LLD_ERR_HIP; }
  *out = f;
  return LLD_OK;
}

extern "C" void lld_frame_destroy(lld_frame* f) {
  if (!f) return;
  (void)hipSetDevice(f->ctx->device);
  (void)hipStreamSynchronize(f->ctx->stream);
  lld_track::state_free(f);
  if (f->d) (void)hipFree(f->d);
  delete f;
}

extern "C" int lld_frame_search_last_frame(lld_frame* f, const uint8_t* t_occupied, const lld_frame_view* view, const lld_last_frame_points* last,
                                           int direction, float th, int check_orientation, float* proj_uvr, lld_orb_search_result* out) {
  if (!f) return LLD_ERR_INVALID;
  lld_orb_search fs = f->consts; fs.nt = f->nt; fs.t_occupied = t_occupied;
  return last_frame_impl(f->ctx, &fs, f, view, last, direction, th, check_orientation, proj_uvr, out);
}

extern "C" int lld_frame_search_local_points(lld_frame* f, const uint8_t* t_occupied, const lld_frame_view* view, const lld_map_points* mp,
                                             float viewing_cos_limit, float th, float nnratio, lld_frustum_result* fr, lld_orb_search_result* out) {
  if (!f) return LLD_ERR_INVALID;
  lld_orb_search fs = f->consts; fs.nt = f->nt; fs.t_occupied = t_occupied;
  return local_points_impl(f->ctx, &fs, f, view, mp, viewing_cos_limit, th, nnratio, fr, out);
}

extern "C" int lld_orb_fuse_search(lld_ctx* ctx, const lld_orb_search* keyframe, const lld_frame_view* view, const lld_map_points* mp, float th,
                                   float* proj_uvr, lld_orb_search_result* out) {
  if (!ctx || !keyframe || !view || !mp || !out) return LLD_ERR_INVALID;
  ProjSearch S{ctx, keyframe, keyframe->nt, mp->n, false};
  int st = S.check(out); if (st) return st;
  const int nq = S.nq;
  if (nq > 0 && (!mp->world_pos || !mp->normal || !mp->max_distance || !mp->min_distance || !mp->desc)) return LLD_ERR_INVALID;
  if (view->n_levels != keyframe->n_levels || !keyframe->level_inv_sigma2) return LLD_ERR_INVALID;
  out->n_matches = 0; out->rounds = 0;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  S.layout();
  const size_t o_pos = S.add_in((size_t)nq * 12), o_nrm = S.add_in((size_t)nq * 12), o_maxd = S.add_in((size_t)nq * 4), o_mind = S.add_in((size_t)nq * 4);
  const size_t o_skip = mp->skip ? S.add_in((size_t)nq) : 0;
  const size_t r_uvr = S.add_out((size_t)nq * 12);
  st = S.alloc(); if (st) return st;
  if (nq) {
    std::memcpy(S.h + o_pos, mp->world_pos, (size_t)nq * 12); std::memcpy(S.h + o_nrm, mp->normal, (size_t)nq * 12);
    std::memcpy(S.h + o_maxd, mp->max_distance, (size_t)nq * 4); std::memcpy(S.h + o_mind, mp->min_distance, (size_t)nq * 4);
    if (mp->skip) std::memcpy(S.h + o_skip, mp->skip, (size_t)nq);
  }
  Problem& P = S.pack(mp->desc, out->owner != nullptr);
  for (int l = 0; l < keyframe->n_levels; l++) P.inv_sigma2[l] = keyframe->level_inv_sigma2[l];
  P.gates = LLD_ORB_GATE_LEVEL | LLD_ORB_GATE_CHI2; P.accept_max = 50;          // TH_LOW, src/ORBmatcher.cc:38,934
  st = S.upload(); if (st) return st;
  if (nq) {
    FuseArgs F; std::memset(&F, 0, sizeof(F));
    F.V = *view; F.n = nq;
    F.pos = reinterpret_cast<const float*>(S.d + o_pos); F.nrm = reinterpret_cast<const float*>(S.d + o_nrm);
    F.maxd = reinterpret_cast<const float*>(S.d + o_maxd); F.mind = reinterpret_cast<const float*>(S.d + o_mind);
    F.skip = mp->skip ? reinterpret_cast<const uint8_t*>(S.d + o_skip) : nullptr;
    for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) F.scale[l] = P.scale[l];
    F.th = th;
    F.q = reinterpret_cast<QRec*>(S.d + S.o_q); F.uvr = reinterpret_cast<float*>(S.d_out + r_uvr);
    hipLaunchKernelGGL(project_fuse_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, F);
    LLD_HIP_TRY(hipGetLastError());
  }
  st = S.search_and_fetch(out); if (st) return st;
  if (nq && proj_uvr) std::memcpy(proj_uvr, S.h_out + r_uvr, (size_t)nq * 12);
  return LLD_OK;
}

extern "C" int lld_orb_search_projected(lld_ctx* ctx, const lld_orb_search* frame, const lld_frame_view* view, const lld_map_points* mp,
                                        const float* angle, const lld_orb_projection* proj, float* proj_uv, int32_t* level, lld_orb_search_result* out) {
  if (!ctx || !frame || !view || !mp || !proj || !out) return LLD_ERR_INVALID;
  const int routine = proj->routine;
  if (routine < LLD_ORB_PROJ_KF_SIM3 || routine > LLD_ORB_PROJ_SIM3_DIR) return LLD_ERR_INVALID;
  const bool reloc = routine == LLD_ORB_PROJ_RELOC, orient = reloc && proj->check_orientation != 0;
  const bool need_normal = routine == LLD_ORB_PROJ_KF_SIM3 || routine == LLD_ORB_PROJ_FUSE_SIM3;
  ProjSearch S{ctx, frame, frame->nt, mp->n, orient};
  int st = S.check(out); if (st) return st;
  const int nq = S.nq;
  if (nq > 0 && (!mp->world_pos || !mp->max_distance || !mp->min_distance || !mp->desc || (need_normal && !mp->normal) || (orient && !angle))) return LLD_ERR_INVALID;
  if (view->n_levels != frame->n_levels) return LLD_ERR_INVALID;
  out->n_matches = 0; out->rounds = 0;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  S.layout();
  const size_t o_pos = S.add_in((size_t)nq * 12), o_nrm = need_normal ? S.add_in((size_t)nq * 12) : 0, o_maxd = S.add_in((size_t)nq * 4), o_mind = S.add_in((size_t)nq * 4);
  const size_t o_skip = mp->skip ? S.add_in((size_t)nq) : 0, o_ang = angle ? S.add_in((size_t)nq * 4) : 0;
  const size_t r_uv = S.add_out((size_t)nq * 8), r_lvl = S.add_out((size_t)nq * 4);
  st = S.alloc(); if (st) return st;
  if (nq) {
    std::memcpy(S.h + o_pos, mp->world_pos, (size_t)nq * 12);
    if (need_normal) std::memcpy(S.h + o_nrm, mp->normal, (size_t)nq * 12);
    std::memcpy(S.h + o_maxd, mp->max_distance, (size_t)nq * 4); std::memcpy(S.h + o_mind, mp->min_distance, (size_t)nq * 4);
    if (mp->skip) std::memcpy(S.h + o_skip, mp->skip, (size_t)nq);
    if (angle) std::memcpy(S.h + o_ang, angle, (size_t)nq * 4);
  }
  Problem& P = S.pack(mp->desc, out->owner != nullptr);
  P.gates = LLD_ORB_GATE_LEVEL;
  switch (routine) {
    case LLD_ORB_PROJ_KF_SIM3: P.accept_max = 50; P.sequential = 1; break;                                   // TH_LOW (:394); vpMatched[bestIdx]=pMP inside the loop
    case LLD_ORB_PROJ_RELOC: P.accept_max = proj->accept_max; P.sequential = 1; P.check_orientation = orient; break;   // ORBdist (:1555)
    case LLD_ORB_PROJ_FUSE_SIM3: P.accept_max = 50; break;                                                   // TH_LOW (:1078)
    default: P.accept_max = 100; break;                                                                      // TH_HIGH (:1220, :1300)
  }
  st = S.upload(); if (st) return st;
  if (nq) {
    ProjGenArgs F; std::memset(&F, 0, sizeof(F));
    F.V = *view; F.n = nq; F.routine = routine;
    for (int k = 0; k < 9; k++) F.sR[k] = proj->sR[k];
    for (int k = 0; k < 3; k++) F.t2[k] = proj->t[k];
    F.pos = reinterpret_cast<const float*>(S.d + o_pos); F.nrm = need_normal ? reinterpret_cast<const float*>(S.d + o_nrm) : nullptr;
    F.maxd = reinterpret_cast<const float*>(S.d + o_maxd); F.mind = reinterpret_cast<const float*>(S.d + o_mind);
    F.skip = mp->skip ? reinterpret_cast<const uint8_t*>(S.d + o_skip) : nullptr;
    F.angle = angle ? reinterpret_cast<const float*>(S.d + o_ang) : nullptr;
    for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) F.scale[l] = P.scale[l];
    F.th = proj->th;
    F.q = reinterpret_cast<QRec*>(S.d + S.o_q); F.uv = reinterpret_cast<float*>(S.d_out + r_uv); F.level = reinterpret_cast<int32_t*>(S.d_out + r_lvl);
    hipLaunchKernelGGL(project_general_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, F);
    LLD_HIP_TRY(hipGetLastError());
  }
  st = S.search_and_fetch(out); if (st) return st;
  if (nq && proj_uv) std::memcpy(proj_uv, S.h_out + r_uv, (size_t)nq * 8);
  if (nq && level) std::memcpy(level, S.h_out + r_lvl, (size_t)nq * 4);
  return LLD_OK;
}

extern "C" int lld_orb_search_by_sim3(lld_ctx* ctx, const lld_orb_search* kf1, const lld_frame_view* view1, const lld_map_points* points1,
                                      const lld_orb_search* kf2, const lld_frame_view* view2, const lld_map_points* points2,
                                      const float* sR12, const float* t12, const float* sR21, const float* t21, float th, int32_t* match12, int32_t* n_found) {
  if (!ctx || !kf1 || !kf2 || !view1 || !view2 || !points1 || !points2 || !sR12 || !t12 || !sR21 || !t21 || !match12 || !n_found) return LLD_ERR_INVALID;
  const int n1 = points1->n, n2 = points2->n;
  if (n1 != kf1->nt || n2 != kf2->nt) return LLD_ERR_INVALID;                   // vpMapPoints1 / 2 are per keypoint of their keyframe
  std::vector<int32_t> m1((size_t)std::max(n1, 1)), m2((size_t)std::max(n2, 1)), bd((size_t)std::max(std::max(n1, n2), 1)), sd(bd.size());
  std::vector<uint8_t> rem(bd.size());
  lld_orb_projection pr; std::memset(&pr, 0, sizeof pr);
  pr.routine = LLD_ORB_PROJ_SIM3_DIR; pr.th = th;
  // KF1's MapPoints into KF2 (:1147-1224): camera 1 from world, then sR21 / t21; searched among KF2's keypoints
  for (int k = 0; k < 9; k++) pr.sR[k] = sR21[k];
  for (int k = 0; k < 3; k++) pr.t[k] = t21[k];
  // pose R1w, t1w; fx, fy, cx, cy are pKF1's in BOTH directions (:1105-1108); IsInImage and PredictScale are pKF2's (:1176, :1188)
  lld_frame_view v = *view1;
  v.min_x = view2->min_x; v.max_x = view2->max_x; v.min_y = view2->min_y; v.max_y = view2->max_y;
  v.log_scale_factor = view2->log_scale_factor; v.n_levels = view2->n_levels;
  lld_orb_search_result r1{m1.data(), bd.data(), sd.data(), rem.data(), nullptr, 0, 0};
  int st = lld_orb_search_projected(ctx, kf2, &v, points1, nullptr, &pr, nullptr, nullptr, &r1); if (st) return st;
  // KF2's MapPoints into KF1 (:1227-1304)
  for (int k = 0; k < 9; k++) pr.sR[k] = sR12[k];
  for (int k = 0; k < 3; k++) pr.t[k] = t12[k];
  v = *view2;                                                                   // pose R2w, t2w; pKF1's intrinsics, bounds and scale pyramid (:1256, :1268)
  v.fx = view1->fx; v.fy = view1->fy; v.cx = view1->cx; v.cy = view1->cy; v.min_x = view1->min_x; v.max_x = view1->max_x; v.min_y = view1->min_y; v.max_y = view1->max_y;
  v.log_scale_factor = view1->log_scale_factor; v.n_levels = view1->n_levels;
  lld_orb_search_result r2{m2.data(), bd.data(), sd.data(), rem.data(), nullptr, 0, 0};
  st = lld_orb_search_projected(ctx, kf1, &v, points2, nullptr, &pr, nullptr, nullptr, &r2); if (st) return st;
  // agreement (:1306-1322)
  int found = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    const int idx2 = m1[i1];
    match12[i1] = -1;
    if (idx2 >= 0 && idx2 < n2 && m2[idx2] == i1) { match12[i1] = idx2; found++; }
  }
  *n_found = found;
  return LLD_OK;
}

// ================================================================ launchers for the device-resident Tracking chain (lld_track_internal.h)
namespace lld_track {

size_t orbs_problem_bytes() { return al(sizeof(Problem)); }
size_t orbs_qrec_bytes(int nq) { return al((size_t)nq * sizeof(QRec)); }
size_t orbs_cache_bytes(int nq) { return al((size_t)nq * 8 * kTopK); }
void orbs_fill_problem(const lld_frame* f, int mode, int nq, const uint8_t* d_occupied, const void* d_qrec, const uint32_t* d_qdesc, const SearchOut& out,
                       void* d_cache, float nnratio, int check_orientation, RunIf run_if, const ApplyDev& ap, void* problem_h) {
  Problem& P = *static_cast<Problem*>(problem_h); std::memset(&P, 0, sizeof(P));
  const lld_orb_search& c = f->consts;
  P.nt = f->nt; P.nq = nq;
  P.t_desc = reinterpret_cast<const uint32_t*>(f->d + f->o_td); P.t_xy = reinterpret_cast<const float*>(f->d + f->o_txy);
  P.t_octave = reinterpret_cast<const int32_t*>(f->d + f->o_toct);
  P.t_uright = f->has_uright ? reinterpret_cast<const float*>(f->d + f->o_tur) : nullptr;
  P.t_angle = (check_orientation && f->has_angle) ? reinterpret_cast<const float*>(f->d + f->o_tang) : nullptr;
  P.t_occupied = d_occupied;
  P.q_desc = d_qdesc; P.q = static_cast<const QRec*>(d_qrec);
  P.min_x = c.grid_min_x; P.min_y = c.grid_min_y; P.winv = c.grid_width_inv; P.hinv = c.grid_height_inv;
  P.cols = c.grid_cols; P.rows = c.grid_rows; P.n_levels = c.n_levels;
  for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) { P.scale[l] = l < c.n_levels ? f->scale[l] : 1.f; P.sigma2[l] = 1.f; P.inv_sigma2[l] = 1.f; }
  P.candidates = LLD_ORB_CAND_GRID;
  P.gates = LLD_ORB_GATE_LEVEL | LLD_ORB_GATE_STEREO; P.accept_max = 100;       // TH_HIGH (src/ORBmatcher.cc:117, :1418)
  if (mode == 0) { P.ratio_mode = 0; P.check_orientation = check_orientation != 0; }
  else { P.ratio_mode = 2; P.nnratio = nnratio; }
  P.sequential = 1;
  P.match = out.match; P.best_dist = out.best_dist; P.second_dist = out.second_dist; P.removed = out.removed; P.owner = out.owner; P.summary = out.summary;
  P.want_owner = 1;
  P.cache = static_cast<unsigned long long*>(d_cache);
  P.desc_in_lds = lds_bytes(P.nt, P.cols * P.rows, true, true) <= kLdsLimit;
  P.run_if = run_if.flag; P.run_if_below = run_if.below; P.run_if_want = run_if.want;
  P.t_occ_obs = ap.kp_obs;                                                      // (the frame's own flags: only a MapPoint with observations blocks its keypoint)
  P.ap_has = ap.kp_has; P.ap_world = ap.kp_world; P.ap_id = ap.kp_id; P.ap_obs = ap.kp_obs; P.ap_q_pos = ap.q_pos; P.ap_q_id = ap.q_id; P.ap_q_obs = ap.q_obs;
  P.ap_counts = ap.counts; P.ap_min_matches = ap.min_matches; P.ap_is_retry = ap.is_retry;
}

// (Round 6 also tried the projection loops as a prologue of the search kernel: one launch less, but 2500 frustum tests on ONE compute unit take
// 38 us where ten workgroups of frustum_kernel take 11 - the chain got 30 us slower.  The projections keep their own launches.)
int orbs_project_last_frame(hipStream_t st, const lld_frame* f, const lld_frame_view* view_h, const lld_frame_view* view_d, const LastFrameDev& last, int direction, float th,
                            void* d_qrec, RunIf run_if) {
  if (last.n <= 0) return LLD_OK;
  LastFrameArgs F; std::memset(&F, 0, sizeof(F));
  if (view_h) F.V = *view_h;
  F.view_d = view_d; F.n = last.n; F.direction = direction;
  F.pos = last.pos; F.valid = last.valid; F.octave = last.octave; F.angle = last.angle; F.has_obs = last.has_obs;
  for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) F.scale[l] = l < f->consts.n_levels ? f->scale[l] : 1.f;
  F.th = th; F.q = static_cast<QRec*>(d_qrec); F.uvr = nullptr;
  F.run_if = run_if.flag; F.run_if_below = run_if.below; F.run_if_want = run_if.want;
  hipLaunchKernelGGL(project_last_frame_kernel, dim3((last.n + 255) / 256), dim3(256), 0, st, F);
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}

int orbs_project_local_points(hipStream_t st, const lld_frame* f, const lld_frame_view* view_h, const lld_frame_view* view_d, const MapPointsDev& mp, float cos_limit, float th,
                              void* d_qrec, uint8_t* d_in_view, int32_t* d_n_in_view) {
  if (mp.n <= 0) return LLD_OK;
  FrustumArgs F; std::memset(&F, 0, sizeof(F));
  if (view_h) F.V = *view_h;
  F.view_d = view_d; F.n = mp.n;
  F.pos = mp.pos; F.nrm = mp.normal; F.maxd = mp.maxd; F.mind = mp.mind; F.has_obs = mp.has_obs; F.skip = mp.skip;
  for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) F.scale[l] = l < f->consts.n_levels ? f->scale[l] : 1.f;
  F.cos_limit = cos_limit; F.th = th; F.q = static_cast<QRec*>(d_qrec); F.in_view = d_in_view; F.n_in_view = d_n_in_view;
  hipLaunchKernelGGL(frustum_kernel, dim3((mp.n + 255) / 256), dim3(256), 0, st, F);
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}

int orbs_launch(lld_ctx* ctx, hipStream_t st, const lld_frame* f, const void* problem_d) {
  const lld_orb_search& c = f->consts;
  const bool desc_in_lds = lds_bytes(f->nt, c.grid_cols * c.grid_rows, true, true) <= kLdsLimit;
  const size_t lds = lds_bytes(f->nt, c.grid_cols * c.grid_rows, true, desc_in_lds);
  if (!ctx->orb_lds_raised) { LLD_HIP_TRY(hipFuncSetAttribute((const void*)orb_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit)); ctx->orb_lds_raised = true; }
  hipLaunchKernelGGL(orb_search_kernel, dim3(1), dim3(kThreads), lds, st, static_cast<const Problem*>(problem_d));
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}

}  // namespace lld_track
