// lld_frame_track.hip — the Tracking thread's per-frame chain as ONE device-resident sequence (round 6, lld_frame_track_*).
//
// The reference runs on one Frame (stereo): TrackWithMotionModel = SearchByProjection(Current, Last) [twice if < 20 matches]
// (src/Tracking.cc:904-911) -> AddLinesFrom(mLastFrame.mvpMapLines) (:924) -> Optimizer::PoseOptimization (:937) -> discard of the outliers
// (:940-975); TrackLocalMap = SearchLocalPoints (:1133, :1613-1664) -> AddLinesFrom(local_lines) (:1140) -> PoseOptimization (:1152) ->
// statistics / discard (:1155-1187).  Between those calls the Frame carries mvpMapPoints, mvbOutlier, mvpMapLines, mvbOutlierLines and mTcw.
// Here they are device arrays of the resident lld_frame (lld_track_internal.h): every stage reads what the stage before it left in HBM
// and NOTHING travels to the host between the stages; the host uploads a stage's map-side inputs once, queues its kernels and returns.
// lld_frame_track_download fetches the per-stage records (one copy, one synchronisation).
//
//   keypoint k   kp_has[k] (mvpMapPoints[k] != NULL), kp_world[k] (its GetWorldPos), kp_id[k] (the caller's MapPoint id), kp_obs[k]
//                (Observations() > 0), kp_outlier[k] (mvbOutlier)
//   line i       ln_has[i], ln_x0 / ln_dir[i] (MapLine::GetMinimalPos), ln_id[i], ln_outlier[i]
//   per frame    pose_qt (double; what Converter::toSE3Quat(mTcw) hands PoseOptimization), the float view (Frame::UpdatePoseMatrices), the
//                ids of the MapPoints the outlier discard marked (mnLastFrameSeen = mnId, :949) and of the MapLines this frame already
//                tracked (tracked_last_id = mnId, :1117): SearchLocalPoints / AddLinesFrom skip them by id.
//
// Kernels of this file: the bookkeeping between the library's search / line / pose kernels (lld_orb_search.hip, lld_match.hip, lld_pose.hip),
// i.e. what the reference does in the few lines of C++ between its calls.
#include "lld_common.h"
#include "lld_device_math.h"
#include "lld_track_internal.h"

namespace {

using namespace lld_track;

constexpr int kRecInts = 16;
struct RecHeader { double pose_qt[7]; double chi2; int32_t i[kRecInts]; };
// i[]: 0 n_inliers, 1 lm_iterations, 2 lm_trials, 3 n_edges, 4 n_search_first, 5 n_search, 6 used_wide, 7 n_points, 8 n_points_map,
//      9 n_lines_matched, 10 n_lines, 11 n_discarded, 12 n_point_edges, 13 n_in_view
enum { RI_INL = 0, RI_ITS, RI_TRIALS, RI_EDGES, RI_SEARCH1, RI_SEARCH, RI_WIDE, RI_POINTS, RI_POINTS_MAP, RI_LINES_MATCHED, RI_LINES, RI_DISCARDED, RI_POINT_EDGES, RI_IN_VIEW };

struct TrackDev {                 // device pointers of the frame's tracking state (all inside lld_frame_track_state::d_state)
  int nt, nl, nr, dim;
  uint8_t* kp_has; float* kp_world; int32_t* kp_id; uint8_t* kp_obs; uint8_t* kp_outlier;
  int32_t* discard; int32_t* n_discard;
  uint8_t* ln_has; double* ln_x0; double* ln_dir; int32_t* ln_id; uint8_t* ln_outlier;
  int32_t* tracked; int32_t* n_tracked; int tracked_cap;
  double* pose_qt; double* pose_out; lld_frame_view* view; LineTrackDevParams* line_params;
  // per-stage records
  RecHeader* rec_h[2]; int32_t* rec_kp_id[2]; uint8_t* rec_kp_out[2]; int32_t* rec_ln_id[2]; uint8_t* rec_ln_out[2];
};

struct ViewConsts { float fx, fy, cx, cy, bf, min_x, max_x, min_y, max_y, log_scale_factor; int n_levels; double b, thr_base, sx, sy; int monocular, use_grid; };

// Frame::SetPose + UpdatePoseMatrices (src/Frame.cc:318-331) from the optimised SE3Quat: Converter::toCvMat narrows to_homogeneous_matrix
// to float (src/Converter.cc:49-70); mOw = -mRcw.t()*mtcw is one cv::gemm (double accumulation, one rounding).  No contraction: the same
// operations, one rounding each, as lld_se3_to_tcw_f32 performs on the host.
__device__ void view_from_pose(const double* qt, const ViewConsts& C, lld_frame_view* V, LineTrackDevParams* L, double* qt_of_float_matrix) {
#pragma clang fp contract(off)
  const lld::Pose p = lld::pose_load(qt);
  const lld::Mat3 R = lld::quat_rotation(p.q);
  float Rf[9], tf[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rf[3 * i + j] = (float)R.m[i][j];
  tf[0] = (float)p.t.x; tf[1] = (float)p.t.y; tf[2] = (float)p.t.z;
  {
    // The Frame keeps the FLOAT matrix only: the next PoseOptimization starts from Converter::toSE3Quat(pFrame->mTcw) (Optimizer.cc:823),
    // i.e. from lld_se3_from_tcw_f32 of these floats, not from the double SE3Quat that was just optimised.
    lld::Mat3 Rd;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rd.m[i][j] = (double)Rf[3 * i + j];
    lld::Pose pf; pf.q = lld::quat_from_rotation(Rd); pf.t = lld::vec3((double)tf[0], (double)tf[1], (double)tf[2]);
    lld::pose_normalize(pf);
    lld::pose_store(pf, qt_of_float_matrix);
  }
  for (int i = 0; i < 9; i++) V->Rcw[i] = Rf[i];
  for (int i = 0; i < 3; i++) {
    V->tcw[i] = tf[i];
    const double acc = ((double)Rf[0 + i] * (double)tf[0] + (double)Rf[3 + i] * (double)tf[1]) + (double)Rf[6 + i] * (double)tf[2];
    V->Ow[i] = (float)(-acc);
  }
  V->fx = C.fx; V->fy = C.fy; V->cx = C.cx; V->cy = C.cy; V->bf = C.bf;
  V->min_x = C.min_x; V->max_x = C.max_x; V->min_y = C.min_y; V->max_y = C.max_y; V->log_scale_factor = C.log_scale_factor; V->n_levels = C.n_levels;
  // AddLinesFrom's camera (src/Tracking.cc:920-923, :1136-1139): T_curr = mTcw.inv() widened to double.  The build takes the frame's own
  // Rwc = Rcw^T and Ow for it (equal to OpenCV's float LU inverse up to float rounding: include/lld_amd.h); the right camera is GetTForRight.
  for (int i = 0; i < 9; i++) L->K[i] = 0.0;
  L->K[0] = (double)C.fx; L->K[2] = (double)C.cx; L->K[4] = (double)C.fy; L->K[5] = (double)C.cy; L->K[8] = 1.0;
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) L->R[3 * r + c] = (double)Rf[3 * c + r]; L->t[r] = (double)V->Ow[r]; }
  for (int r = 0; r < 3; r++) L->tr[r] = L->t[r] + L->R[3 * r] * C.b;
  L->thr_base = C.thr_base; L->sx = C.sx; L->sy = C.sy; L->monocular = C.monocular; L->use_grid = C.use_grid;
}

// A new frame enters TrackWithMotionModel: no MapPoints (fill(..., NULL), :897), no outlier flags, nothing discarded or tracked yet.
__global__ void track_reset_kernel(TrackDev D, const double* pose_guess, const lld_frame_view* view_up, const LineTrackDevParams* lp_up) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { *D.view = *view_up; *D.line_params = *lp_up; }                 // the caller's view of the predicted pose (kept if PoseOptimization does not run)
  if (i < D.nt) { D.kp_has[i] = 0; D.kp_id[i] = -1; D.kp_obs[i] = 0; D.kp_outlier[i] = 0; }
  if (i < D.nl) { D.ln_has[i] = 0; D.ln_id[i] = -1; D.ln_outlier[i] = 0; }
  if (i < 7) D.pose_qt[i] = pose_guess[i];
  if (i == 0) { *D.n_discard = 0; *D.n_tracked = 0; }
  if (i < 2 * (int)(sizeof(RecHeader) / 4)) {
    int32_t* h = reinterpret_cast<int32_t*>(i < (int)(sizeof(RecHeader) / 4) ? D.rec_h[0] : D.rec_h[1]);
    h[i % (int)(sizeof(RecHeader) / 4)] = 0;
  }
}

// The frame enters TrackLocalMap from a stage 1 that ran elsewhere (lld_frame_track_set_state): what that routine left in the frame.
struct HeldUp { const int32_t* kp_id; const float* kp_world; const uint8_t* kp_obs; const uint8_t* kp_out; const int32_t* seen; int n_seen;
                const int32_t* ln_id; const double* ln_x0; const double* ln_dir; const uint8_t* ln_out; const int32_t* tracked; int n_tracked; };
__global__ void track_load_kernel(TrackDev D, HeldUp U, const double* pose_qt, const lld_frame_view* view_up, const LineTrackDevParams* lp_up) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { *D.view = *view_up; *D.line_params = *lp_up; *D.n_discard = U.n_seen; }
  if (i < D.nt) {
    const int32_t id = U.kp_id[i];
    const bool has = id >= 0;
    D.kp_has[i] = has; D.kp_id[i] = has ? id : -1; D.kp_obs[i] = has ? U.kp_obs[i] : 0; D.kp_outlier[i] = U.kp_out[i];
    for (int c = 0; c < 3; c++) D.kp_world[3 * i + c] = has ? U.kp_world[3 * i + c] : 0.f;
  }
  if (i < U.n_seen) D.discard[i] = U.seen[i];
  if (i < D.nl) {
    const int32_t id = U.ln_id[i];
    const bool has = id >= 0;
    D.ln_has[i] = has; D.ln_id[i] = has ? id : -1; D.ln_outlier[i] = U.ln_out[i];
    for (int c = 0; c < 3; c++) { D.ln_x0[3 * i + c] = has ? U.ln_x0[3 * i + c] : 0.0; D.ln_dir[3 * i + c] = has ? U.ln_dir[3 * i + c] : 0.0; }
  }
  if (i < U.n_tracked) D.tracked[i] = U.tracked[i];
  if (i == 0) *D.n_tracked = U.n_tracked;
  if (i < 7) D.pose_qt[i] = pose_qt[i];
  if (i < 2 * (int)(sizeof(RecHeader) / 4)) {
    int32_t* h = reinterpret_cast<int32_t*>(i < (int)(sizeof(RecHeader) / 4) ? D.rec_h[0] : D.rec_h[1]);
    h[i % (int)(sizeof(RecHeader) / 4)] = 0;
  }
  if (i < D.nt) { D.rec_kp_id[0][i] = -1; D.rec_kp_out[0][i] = 0; }
  if (i < D.nl) { D.rec_ln_id[0][i] = -1; D.rec_ln_out[0][i] = 0; }
}

// What follows PoseOptimization.  stage 0 = TrackWithMotionModel (src/Tracking.cc:940-975): an outlier point leaves the frame, its flag is
// cleared, its MapPoint is marked seen (-> the discard list); an outlier line leaves, its flag STAYS (the reference does not clear
// mvbOutlierLines).  stage 1 = TrackLocalMap (:1155-1187): outlier points leave (STEREO) with their flag kept, outlier lines leave.
// Before any of it the stage's record is written: ids as matched, flags as PoseOptimization left them.  One workgroup.
__global__ __launch_bounds__(1024) void track_after_pose_kernel(TrackDev D, ViewConsts C, int stage) {
  __shared__ int cnt[8];
  const int tid = threadIdx.x;
  if (tid < 8) cnt[tid] = 0;
  __syncthreads();
  // the last wavefront's first lane turns the optimised pose into the frame's float view while the other fifteen walk the keypoints and lines
  // (a chain of ~500 dependent fp64 instructions behind one global load: it would otherwise follow the walk's own chain of loads)
  constexpr int kWalk = 960;
  if (tid == kWalk) {
    const int* pi = reinterpret_cast<const int*>(D.pose_out + 8);
    RecHeader& H = *D.rec_h[stage];
    for (int c = 0; c < 7; c++) H.pose_qt[c] = D.pose_out[c];
    H.chi2 = D.pose_out[7];
    H.i[RI_INL] = pi[0]; H.i[RI_ITS] = pi[1]; H.i[RI_TRIALS] = pi[2]; H.i[RI_EDGES] = pi[3]; H.i[RI_POINT_EDGES] = pi[4];
    // pFrame->SetPose(pose) - unless PoseOptimization returned before it optimised (fewer than three points, Optimizer.cc:809-810)
    if (pi[4] >= 3) view_from_pose(D.pose_out, C, D.view, D.line_params, D.pose_qt);
  }
  int n_pts = 0, n_map = 0, n_disc = 0, n_lm = 0, n_ln = 0;
  for (int k0 = 0; k0 < D.nt && tid < kWalk; k0 += 2 * kWalk) {
    // two keypoints per lane, every load issued before the first use
    const int ka = k0 + tid, kb = ka + kWalk;
    const bool ina = ka < D.nt, inb = kb < D.nt;
    const int kca = ina ? ka : 0, kcb = inb ? kb : 0;
    const uint8_t ha = D.kp_has[kca], hb = D.kp_has[kcb], oa = D.kp_outlier[kca], ob = D.kp_outlier[kcb], va = D.kp_obs[kca], vb = D.kp_obs[kcb];
    const int32_t ia = D.kp_id[kca], ib = D.kp_id[kcb];
    auto one = [&](int k, bool in, uint8_t has_, uint8_t out_, uint8_t obs_, int32_t id_) {
      if (!in) return;
      const bool has = has_ != 0;
      const uint8_t bad = has ? out_ : 0;
      D.rec_kp_id[stage][k] = has ? id_ : -1;
      D.rec_kp_out[stage][k] = bad;
      if (!has) return;
      if (bad) {
        if (stage == 0) { const int at = atomicAdd(D.n_discard, 1); D.discard[at] = id_; D.kp_outlier[k] = 0; }
        D.kp_has[k] = 0; D.kp_id[k] = -1;
        n_disc++;
      } else { n_pts++; if (obs_) n_map++; }
    };
    one(ka, ina, ha, oa, va, ia); one(kb, inb, hb, ob, vb, ib);
  }
  for (int i = tid; i < D.nl && tid < kWalk; i += kWalk) {
    const uint8_t has_ = D.ln_has[i], out_ = D.ln_outlier[i]; const int32_t id_ = D.ln_id[i];
    const bool has = has_ != 0;
    D.rec_ln_id[stage][i] = has ? id_ : -1;
    D.rec_ln_out[stage][i] = has ? out_ : 0;
    if (!has) continue;
    n_lm++;
    if (out_) { D.ln_has[i] = 0; D.ln_id[i] = -1; } else n_ln++;
  }
  // wavefront sums first: five LDS atomics per wavefront instead of per lane
  for (int off = 32; off > 0; off >>= 1) {
    n_pts += __shfl_xor(n_pts, off); n_map += __shfl_xor(n_map, off); n_disc += __shfl_xor(n_disc, off); n_lm += __shfl_xor(n_lm, off); n_ln += __shfl_xor(n_ln, off);
  }
  if ((tid & 63) == 0) { atomicAdd(&cnt[0], n_pts); atomicAdd(&cnt[1], n_map); atomicAdd(&cnt[2], n_disc); atomicAdd(&cnt[3], n_lm); atomicAdd(&cnt[4], n_ln); }
  __syncthreads();
  if (tid == 0) {
    RecHeader& H = *D.rec_h[stage];
    H.i[RI_POINTS] = cnt[0]; H.i[RI_POINTS_MAP] = cnt[1]; H.i[RI_DISCARDED] = cnt[2]; H.i[RI_LINES_MATCHED] = cnt[3]; H.i[RI_LINES] = cnt[4];
  }
}

// SearchLocalPoints' "already seen in this frame" (src/Tracking.cc:1616-1643): pMP->mnLastFrameSeen == mCurrentFrame.mnId holds for the
// MapPoints the frame holds (:1629) and for those the outlier discard marked (:949).  Ids are the caller's (>= 0).  Lines: tracked_last_id ==
// mnId (:1023) against the list of lines assigned so far.
constexpr int kSeenThreads = 1024;
__device__ __forceinline__ unsigned seen_hash(int32_t id, unsigned mask) { return ((unsigned)id * 2654435761u >> 7) & mask; }
__device__ __forceinline__ void seen_insert(int32_t* tab, unsigned mask, int32_t id) {
  unsigned h = seen_hash(id, mask);
  for (;;) {
    const int32_t old = atomicCAS(&tab[h], -1, id);
    if (old == -1 || old == id) return;
    h = (h + 1) & mask;
  }
}
__device__ __forceinline__ bool seen_lookup(const int32_t* tab, unsigned mask, int32_t id) {
  unsigned h = seen_hash(id, mask);
  for (;;) {
    const int32_t v = tab[h];
    if (v == id) return true;
    if (v == -1) return false;
    h = (h + 1) & mask;
  }
}
// One workgroup: the ids the frame holds or discarded (and the lines it tracked) go into two open-addressing hash sets in LDS (at most half
// full: ids >= 0, -1 = empty), then every local MapPoint / MapLine probes its id.
__global__ __launch_bounds__(kSeenThreads) void track_mark_seen_kernel(TrackDev D, int n_mp, const int32_t* mp_id, const uint8_t* mp_skip, uint8_t* mp_skip_out,
                                                                      int n_ml, const int32_t* ml_id, const uint8_t* ml_skip, uint8_t* ml_skip_out, unsigned pmask, unsigned lmask) {
  extern __shared__ int32_t seen_tab[];              // [pmask + 1] point ids, then [lmask + 1] line ids
  int32_t* ptab = seen_tab; int32_t* ltab = seen_tab + pmask + 1;
  const int tid = threadIdx.x;
  for (unsigned k = tid; k < pmask + 1 + lmask + 1; k += kSeenThreads) seen_tab[k] = -1;
  const int nd = *D.n_discard, ntr = min(*D.n_tracked, D.tracked_cap);
  __syncthreads();
  for (int k = tid; k < D.nt; k += kSeenThreads) if (D.kp_has[k]) seen_insert(ptab, pmask, D.kp_id[k]);
  for (int k = tid; k < nd; k += kSeenThreads) seen_insert(ptab, pmask, D.discard[k]);
  for (int k = tid; k < ntr; k += kSeenThreads) seen_insert(ltab, lmask, D.tracked[k]);
  __syncthreads();
  for (int q = tid; q < n_mp; q += kSeenThreads) mp_skip_out[q] = ((mp_skip && mp_skip[q]) || seen_lookup(ptab, pmask, mp_id[q])) ? 1 : 0;
  for (int j = tid; j < n_ml; j += kSeenThreads) ml_skip_out[j] = ((ml_skip && ml_skip[j]) || seen_lookup(ltab, lmask, ml_id[j])) ? 1 : 0;
}

inline size_t al(size_t b) { return (b + 255) & ~size_t(255); }

}  // namespace

struct lld_frame_track_state {
  TrackDev D{};
  char* d_state = nullptr;                 // per-frame state + records (sized at lld_frame_set_lines / first track call)
  size_t rec_off = 0, rec_bytes = 0;       // the two records, contiguous (one download)
  // frame lines
  int nl = 0, nr = 0, dim = 0; double sx = 0, sy = 0;
  const float* ln_left = nullptr; const int32_t* ln_loct = nullptr; const float* ln_right = nullptr; const int32_t* ln_roct = nullptr;
  const int32_t* ln_match = nullptr; const float* ln_desc = nullptr; int32_t* ln_cell = nullptr;
  // per-call work: uploaded inputs + search / line / pose scratch (grow-only), one pinned staging region per stage
  char* d_work = nullptr; size_t work_bytes = 0;
  char* h_stage[2] = {nullptr, nullptr}; size_t h_stage_bytes[2] = {0, 0};
  hipEvent_t uploaded[2] = {nullptr, nullptr}; bool upload_pending[2] = {false, false};
  char* h_rec = nullptr; size_t h_rec_bytes = 0;
  ViewConsts consts{};
  bool stage1_queued = false;
  size_t in_view_off = 0; int n_in_view = 0;   // Frame::isInFrustum flags of stage 2's local MapPoints, inside d_work
};

namespace lld_track {
void state_free(lld_frame* f) {
  lld_frame_track_state* S = f->track;
  if (!S) return;
  if (S->d_state) (void)hipFree(S->d_state);
  if (S->d_work) (void)hipFree(S->d_work);
  for (int s = 0; s < 2; s++) { if (S->h_stage[s]) (void)hipHostFree(S->h_stage[s]); if (S->uploaded[s]) (void)hipEventDestroy(S->uploaded[s]); }
  if (S->h_rec) (void)hipHostFree(S->h_rec);
  delete S;
  f->track = nullptr;
}
}  // namespace lld_track

namespace {

// (Re)builds the device state of a frame: keypoint-side arrays, the frame's lines (copied from `L`, may be null: a frame without lines),
// the records.  Synchronous (it is part of uploading a new frame, like lld_frame_create).
int state_build(lld_frame* f, const lld_frame_lines* L) {
  lld_ctx* ctx = f->ctx;
  lld_track::state_free(f);
  lld_frame_track_state* S = new lld_frame_track_state();
  f->track = S;
  const int nt = f->nt, nl = L ? L->n_left : 0, nr = L ? L->n_right : 0, dim = L ? L->dim : 1;
  S->nl = nl; S->nr = nr; S->dim = dim; S->sx = L ? L->sx : 1.0; S->sy = L ? L->sy : 1.0;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const size_t o_has = take(nt), o_world = take((size_t)nt * 12), o_id = take((size_t)nt * 4), o_obs = take(nt), o_out = take(nt), o_disc = take((size_t)nt * 4 + 4);
  const size_t o_lhas = take(nl), o_lx0 = take((size_t)nl * 24), o_ldir = take((size_t)nl * 24), o_lid = take((size_t)nl * 4), o_lout = take(nl);
  const int tracked_cap = 2 * nl + 16;
  const size_t o_trk = take((size_t)tracked_cap * 4), o_cnt = take(64), o_pose = take(7 * 8), o_pout = take(12 * 8), o_view = take(sizeof(lld_frame_view)), o_lp = take(sizeof(LineTrackDevParams));
  const size_t o_ll = take((size_t)nl * 16), o_lo = take((size_t)nl * 4), o_lr = take((size_t)std::max(nr, 1) * 16), o_ro = take((size_t)std::max(nr, 1) * 4), o_lm = take((size_t)nl * 4);
  const size_t o_ld = take((size_t)nl * dim * 4), o_lc = take((size_t)nl * 4);
  S->rec_off = o;
  size_t o_rh[2], o_rk[2], o_rko[2], o_rl[2], o_rlo[2];
  for (int s = 0; s < 2; s++) { o_rh[s] = take(sizeof(RecHeader)); o_rk[s] = take((size_t)nt * 4); o_rko[s] = take(nt); o_rl[s] = take((size_t)nl * 4); o_rlo[s] = take(nl); }
  S->rec_bytes = o - S->rec_off;
  if (hipMalloc(reinterpret_cast<void**>(&S->d_state), o + 256) != hipSuccess) { lld_track::state_free(f); return LLD_ERR_ALLOC; }
  char* d = S->d_state;
  TrackDev& D = S->D;
  D.nt = nt; D.nl = nl; D.nr = nr; D.dim = dim;
  D.kp_has = reinterpret_cast<uint8_t*>(d + o_has); D.kp_world = reinterpret_cast<float*>(d + o_world); D.kp_id = reinterpret_cast<int32_t*>(d + o_id);
  D.kp_obs = reinterpret_cast<uint8_t*>(d + o_obs); D.kp_outlier = reinterpret_cast<uint8_t*>(d + o_out);
  D.discard = reinterpret_cast<int32_t*>(d + o_disc); D.n_discard = reinterpret_cast<int32_t*>(d + o_cnt);
  D.ln_has = reinterpret_cast<uint8_t*>(d + o_lhas); D.ln_x0 = reinterpret_cast<double*>(d + o_lx0); D.ln_dir = reinterpret_cast<double*>(d + o_ldir);
  D.ln_id = reinterpret_cast<int32_t*>(d + o_lid); D.ln_outlier = reinterpret_cast<uint8_t*>(d + o_lout);
  D.tracked = reinterpret_cast<int32_t*>(d + o_trk); D.n_tracked = reinterpret_cast<int32_t*>(d + o_cnt) + 1; D.tracked_cap = tracked_cap;
  D.pose_qt = reinterpret_cast<double*>(d + o_pose); D.pose_out = reinterpret_cast<double*>(d + o_pout);
  D.view = reinterpret_cast<lld_frame_view*>(d + o_view); D.line_params = reinterpret_cast<LineTrackDevParams*>(d + o_lp);
  for (int s = 0; s < 2; s++) {
    D.rec_h[s] = reinterpret_cast<RecHeader*>(d + o_rh[s]); D.rec_kp_id[s] = reinterpret_cast<int32_t*>(d + o_rk[s]); D.rec_kp_out[s] = reinterpret_cast<uint8_t*>(d + o_rko[s]);
    D.rec_ln_id[s] = reinterpret_cast<int32_t*>(d + o_rl[s]); D.rec_ln_out[s] = reinterpret_cast<uint8_t*>(d + o_rlo[s]);
  }
  S->ln_left = reinterpret_cast<const float*>(d + o_ll); S->ln_loct = reinterpret_cast<const int32_t*>(d + o_lo); S->ln_right = reinterpret_cast<const float*>(d + o_lr);
  S->ln_roct = reinterpret_cast<const int32_t*>(d + o_ro); S->ln_match = reinterpret_cast<const int32_t*>(d + o_lm); S->ln_desc = reinterpret_cast<const float*>(d + o_ld);
  S->ln_cell = reinterpret_cast<int32_t*>(d + o_lc);
  LLD_HIP_TRY(hipMemsetAsync(d, 0, o, ctx->stream));
  if (nl > 0) {
    // the lines travel through the context's pinned staging in one copy (synchronous: the staging is the context's)
    const size_t span = (o_lc) - o_ll;
    void* hb; int st = lld_ctx_pinned(ctx, span, &hb); if (st) return st;
    char* h = static_cast<char*>(hb);
    std::memset(h, 0, span);
    std::memcpy(h + (o_ll - o_ll), L->left, (size_t)nl * 16); std::memcpy(h + (o_lo - o_ll), L->left_octave, (size_t)nl * 4);
    if (nr > 0) { std::memcpy(h + (o_lr - o_ll), L->right, (size_t)nr * 16); std::memcpy(h + (o_ro - o_ll), L->right_octave, (size_t)nr * 4); }
    std::memcpy(h + (o_lm - o_ll), L->line_matches, (size_t)nl * 4); std::memcpy(h + (o_ld - o_ll), L->desc, (size_t)nl * dim * 4);
    LLD_HIP_TRY(hipMemcpyAsync(d + o_ll, h, span, hipMemcpyHostToDevice, ctx->stream));
    int st2 = line_cells_dev(ctx->stream, S->ln_left, nl, S->sx, S->sy, S->ln_cell); if (st2) return st2;
  }
  LLD_HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int s = 0; s < 2; s++) LLD_HIP_TRY(hipEventCreateWithFlags(&S->uploaded[s], hipEventDisableTiming));
  return LLD_OK;
}

int ensure_state(lld_frame* f) { return f->track ? LLD_OK : state_build(f, nullptr); }

int ensure_work(lld_frame_track_state* S, lld_ctx* ctx, size_t bytes) {
  if (bytes <= S->work_bytes) return LLD_OK;
  LLD_HIP_TRY(hipStreamSynchronize(ctx->stream));                // kernels of an earlier stage may still read the old block
  if (S->d_work) LLD_HIP_TRY(hipFree(S->d_work));
  S->d_work = nullptr; S->work_bytes = 0;
  const size_t want = bytes + (bytes >> 2) + 4096;
  if (hipMalloc(reinterpret_cast<void**>(&S->d_work), want) != hipSuccess) return LLD_ERR_ALLOC;
  S->work_bytes = want;
  return LLD_OK;
}
int ensure_stage(lld_frame_track_state* S, int s, size_t bytes) {
  if (S->upload_pending[s]) { LLD_HIP_TRY(hipEventSynchronize(S->uploaded[s])); S->upload_pending[s] = false; }   // (long complete: the previous frame's copy)
  if (bytes <= S->h_stage_bytes[s]) return LLD_OK;
  if (S->h_stage[s]) LLD_HIP_TRY(hipHostFree(S->h_stage[s]));
  S->h_stage[s] = nullptr; S->h_stage_bytes[s] = 0;
  const size_t want = bytes + (bytes >> 2) + 4096;
  LLD_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_stage[s]), want, hipHostMallocDefault));
  S->h_stage_bytes[s] = want;
  return LLD_OK;
}

void fill_consts(lld_frame_track_state* S, const lld_frame* f, const lld_track_params* P, const lld_frame_view* view) {
  ViewConsts& C = S->consts;
  C.fx = view->fx; C.fy = view->fy; C.cx = view->cx; C.cy = view->cy; C.bf = view->bf;
  C.min_x = view->min_x; C.max_x = view->max_x; C.min_y = view->min_y; C.max_y = view->max_y; C.log_scale_factor = view->log_scale_factor; C.n_levels = view->n_levels;
  C.b = (double)(view->bf / view->fx);                                      // mb = mbf / fx, floats (src/Frame.cc:97)
  C.thr_base = P->line_thr_reproj_base; C.sx = S->sx; C.sy = S->sy; C.monocular = P->monocular; C.use_grid = P->line_use_grid;
  (void)f;
}

// the host-side twin of view_from_pose's line camera for stage 1, where the view is the caller's (mVelocity * mLastFrame.mTcw is a float product)
void line_params_from_view(const ViewConsts& C, const lld_frame_view& V, LineTrackDevParams* L) {
  std::memset(L, 0, sizeof *L);
  L->K[0] = (double)C.fx; L->K[2] = (double)C.cx; L->K[4] = (double)C.fy; L->K[5] = (double)C.cy; L->K[8] = 1.0;
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) L->R[3 * r + c] = (double)V.Rcw[3 * c + r]; L->t[r] = (double)V.Ow[r]; }
  for (int r = 0; r < 3; r++) L->tr[r] = L->t[r] + L->R[3 * r] * C.b;
  L->thr_base = C.thr_base; L->sx = C.sx; L->sy = C.sy; L->monocular = C.monocular; L->use_grid = C.use_grid;
}

struct LinesUp { size_t x0, dir, x1, x2, skip, desc, id, skip2, matches; };

// lines half of a stage: AddLinesFrom on the uploaded map lines (skip bytes at `d_skip`), then the assignment into the frame
int run_lines(lld_frame* f, hipStream_t st, const lld_track_params* P, const lld_map_lines* ML, char* d_in, const LinesUp& U, const uint8_t* d_skip, char* d_line_work, int stage) {
  lld_frame_track_state* S = f->track;
  const int n_map = ML ? ML->n : 0;
  if (n_map <= 0 || S->nl <= 0) return LLD_OK;
  int32_t* d_matches = reinterpret_cast<int32_t*>(d_in + U.matches);
  LineMapDev M{n_map, reinterpret_cast<const double*>(d_in + U.x0), reinterpret_cast<const double*>(d_in + U.dir), reinterpret_cast<const double*>(d_in + U.x1),
               reinterpret_cast<const double*>(d_in + U.x2), d_skip, reinterpret_cast<const float*>(d_in + U.desc)};
  LineFrameDev Cur{S->nl, S->ln_left, S->ln_loct, S->ln_right, S->ln_match, S->D.ln_has, S->ln_cell, S->ln_desc, S->dim};
  const LineApplyDev ap{S->D.ln_has, S->D.ln_x0, S->D.ln_dir, S->D.ln_id, S->D.tracked, S->D.n_tracked, S->D.tracked_cap, reinterpret_cast<const int32_t*>(d_in + U.id)};
  (void)stage;
  return line_track_launch_dev(f->ctx, st, S->D.line_params, M, Cur, P->line_md_thr, d_line_work, d_matches, ap);
}

int run_pose(lld_frame* f, hipStream_t st, const lld_track_params* P, char* d_pose_work, int stage) {
  lld_frame_track_state* S = f->track;
  PoseTrackDev in{};
  in.nt = f->nt; in.nl = S->nl;
  in.t_xy = reinterpret_cast<const float*>(f->d + f->o_txy); in.t_uright = f->has_uright ? reinterpret_cast<const float*>(f->d + f->o_tur) : nullptr;
  in.t_octave = reinterpret_cast<const int32_t*>(f->d + f->o_toct);
  in.kp_has = S->D.kp_has; in.kp_world = S->D.kp_world;
  in.ln_left = S->ln_left; in.ln_loct = S->ln_loct; in.ln_right = S->ln_right; in.ln_roct = S->ln_roct; in.ln_match = S->ln_match;
  in.ln_has = S->D.ln_has; in.ln_x0 = S->D.ln_x0; in.ln_dir = S->D.ln_dir;
  in.pose_qt = S->D.pose_qt; in.cam = P->cam; in.gamma = P->pose.gamma;
  for (int l = 0; l < LLD_ORB_MAX_LEVELS; l++) in.inv_sigma2[l] = f->inv_sigma2[l];
  in.kp_outlier = S->D.kp_outlier; in.ln_outlier = S->D.ln_outlier; in.pose_out = S->D.pose_out;
  int s = pose_track_launch(f->ctx, st, in, P->pose, d_pose_work); if (s) return s;
  hipLaunchKernelGGL(track_after_pose_kernel, dim3(1), dim3(1024), 0, st, S->D, S->consts, stage);
  LLD_HIP_TRY(hipGetLastError());
  return LLD_OK;
}

int check_lines(const lld_frame_track_state* S, const lld_map_lines* ML) {
  if (!ML || ML->n == 0) return LLD_OK;
  if (ML->n < 0 || !ML->x0 || !ML->dir || !ML->x1 || !ML->x2 || !ML->desc || !ML->id) return LLD_ERR_INVALID;
  (void)S;
  return LLD_OK;
}

size_t lay_lines(size_t& o, int n_map, int dim, LinesUp* U, size_t* dev_only) {
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  U->x0 = take((size_t)n_map * 24); U->dir = take((size_t)n_map * 24); U->x1 = take((size_t)n_map * 24); U->x2 = take((size_t)n_map * 24);
  U->skip = take(n_map); U->desc = take((size_t)n_map * dim * 4); U->id = take((size_t)n_map * 4);
  (void)dev_only;
  return o;
}
void pack_lines(char* h, const LinesUp& U, const lld_map_lines* ML, int dim) {
  const size_t n = ML->n;
  std::memcpy(h + U.x0, ML->x0, n * 24); std::memcpy(h + U.dir, ML->dir, n * 24); std::memcpy(h + U.x1, ML->x1, n * 24); std::memcpy(h + U.x2, ML->x2, n * 24);
  if (ML->skip) std::memcpy(h + U.skip, ML->skip, n); else std::memset(h + U.skip, 0, n);
  std::memcpy(h + U.desc, ML->desc, n * dim * 4); std::memcpy(h + U.id, ML->id, n * 4);
}

}  // namespace

extern "C" {

void lld_track_params_default(lld_track_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  lld_pose_params_default(&p->pose);
  p->th_motion = 7.f; p->th_local = 1.f; p->nnratio_local = 0.8f; p->viewing_cos_limit = 0.5f;
  p->direction = 0; p->check_orientation = 1; p->wide_retry = 1; p->monocular = 0;
  p->line_thr_reproj_base = 2.0; p->line_md_thr = 0.9; p->line_use_grid = 1;
}

int lld_frame_set_lines(lld_frame* f, const lld_frame_lines* L) {
  if (!f) return LLD_ERR_INVALID;
  if (L) {
    if (L->n_left < 0 || L->n_right < 0 || L->dim <= 0 || L->dim > 128 || !(L->sx > 0) || !(L->sy > 0)) return LLD_ERR_INVALID;
    if (L->n_left > 0 && (!L->left || !L->left_octave || !L->line_matches || !L->desc)) return LLD_ERR_INVALID;
    if (L->n_right > 0 && (!L->right || !L->right_octave)) return LLD_ERR_INVALID;
    for (int i = 0; i < L->n_left; i++) {
      if (L->left_octave[i] < 0 || L->left_octave[i] > 64 || L->line_matches[i] >= L->n_right) return LLD_ERR_INVALID;
    }
    for (int i = 0; i < L->n_right; i++) if (L->right_octave[i] < 0 || L->right_octave[i] > 64) return LLD_ERR_INVALID;
  }
  LLD_HIP_TRY(hipSetDevice(f->ctx->device));
  LLD_HIP_TRY(hipStreamSynchronize(f->ctx->stream));
  return state_build(f, (L && L->n_left > 0) ? L : nullptr);
}

int lld_frame_track_motion_model(lld_frame* f, const lld_track_params* P, const lld_frame_view* view, const double* pose_qt, const lld_last_frame_points* last,
                                 const int32_t* last_point_id, const lld_map_lines* last_lines) {
  if (!f || !P || !view || !pose_qt || !last) return LLD_ERR_INVALID;
  const int nq = last->n, nt = f->nt;
  if (nq < 0 || (nq > 0 && (!last->world_pos || !last->valid || !last->octave || !last->desc || !last_point_id || (P->check_orientation && !last->angle)))) return LLD_ERR_INVALID;
  if (P->check_orientation && nt > 0 && !f->has_angle) return LLD_ERR_INVALID;
  if (view->n_levels != f->consts.n_levels) return LLD_ERR_INVALID;
  for (int i = 0; i < nq; i++) if (last->octave[i] < 0 || last->octave[i] >= f->consts.n_levels) return LLD_ERR_INVALID;
  lld_ctx* ctx = f->ctx;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  int s = ensure_state(f); if (s) return s;
  lld_frame_track_state* S = f->track;
  s = check_lines(S, last_lines); if (s) return s;
  const int n_map = (last_lines && S->nl > 0) ? last_lines->n : 0;
  fill_consts(S, f, P, view);
  // ---- layout of the uploaded block, then the device-only scratch behind it
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const size_t o_prob1 = take(orbs_problem_bytes()), o_prob2 = take(orbs_problem_bytes());
  const size_t o_pose = take(7 * 8), o_lp = take(sizeof(LineTrackDevParams)), o_view = take(sizeof(lld_frame_view));
  const size_t o_pos = take((size_t)nq * 12), o_val = take(nq), o_oct = take((size_t)nq * 4), o_ang = take((size_t)nq * 4), o_obs = take(nq), o_desc = take((size_t)nq * 32), o_id = take((size_t)nq * 4);
  LinesUp U{}; lay_lines(o, n_map, S->dim, &U, nullptr);
  const size_t up_bytes = o;
  U.matches = take((size_t)std::max(n_map, 1) * 4);
  size_t o_so[2][6];
  for (int k = 0; k < 2; k++) { o_so[k][0] = take((size_t)nq * 4); o_so[k][1] = take((size_t)nq * 4); o_so[k][2] = take((size_t)nq * 4); o_so[k][3] = take(nq); o_so[k][4] = take((size_t)nt * 4); o_so[k][5] = take(16); }
  const size_t o_qrec = take(orbs_qrec_bytes(nq)), o_cache = take(orbs_cache_bytes(nq));
  const size_t o_lwork = take(line_track_work_bytes(n_map, S->nl)), o_pwork = take(pose_track_work_bytes(nt, S->nl));
  s = ensure_work(S, ctx, o); if (s) return s;
  s = ensure_stage(S, 0, up_bytes); if (s) return s;
  char* h = S->h_stage[0]; char* d = S->d_work;
  // ---- pack
  if (nq) {
    std::memcpy(h + o_pos, last->world_pos, (size_t)nq * 12); std::memcpy(h + o_val, last->valid, nq); std::memcpy(h + o_oct, last->octave, (size_t)nq * 4);
    if (last->angle) std::memcpy(h + o_ang, last->angle, (size_t)nq * 4); else std::memset(h + o_ang, 0, (size_t)nq * 4);
    if (last->has_obs) std::memcpy(h + o_obs, last->has_obs, nq); else std::memset(h + o_obs, 1, nq);
    std::memcpy(h + o_desc, last->desc, (size_t)nq * 32); std::memcpy(h + o_id, last_point_id, (size_t)nq * 4);
  }
  if (n_map) pack_lines(h, U, last_lines, S->dim);
  std::memcpy(h + o_pose, pose_qt, 7 * 8);
  line_params_from_view(S->consts, *view, reinterpret_cast<LineTrackDevParams*>(h + o_lp));
  std::memcpy(h + o_view, view, sizeof(lld_frame_view));
  SearchOut so[2];
  for (int k = 0; k < 2; k++)
    so[k] = SearchOut{reinterpret_cast<int32_t*>(d + o_so[k][0]), reinterpret_cast<int32_t*>(d + o_so[k][1]), reinterpret_cast<int32_t*>(d + o_so[k][2]),
                      reinterpret_cast<uint8_t*>(d + o_so[k][3]), reinterpret_cast<int32_t*>(d + o_so[k][4]), reinterpret_cast<int32_t*>(d + o_so[k][5])};
  const bool wide = P->wide_retry != 0;
  const RunIf gate{so[0].summary, 20, 1};                                     // if(nmatches<20) (src/Tracking.cc:907)
  const LastFrameDev LF{nq, reinterpret_cast<const float*>(d + o_pos), reinterpret_cast<const uint8_t*>(d + o_val), reinterpret_cast<const int32_t*>(d + o_oct),
                        reinterpret_cast<const float*>(d + o_ang), reinterpret_cast<const uint8_t*>(d + o_obs)};
  int32_t* counts = S->D.rec_h[0]->i + RI_SEARCH1;                            // RI_SEARCH1, RI_SEARCH, RI_WIDE are consecutive
  ApplyDev ap{S->D.kp_has, S->D.kp_world, S->D.kp_id, S->D.kp_obs, LF.pos, reinterpret_cast<const int32_t*>(d + o_id), LF.has_obs, counts, wide ? 20 : 0, 0};
  orbs_fill_problem(f, 0, nq, S->D.kp_has, d + o_qrec, reinterpret_cast<const uint32_t*>(d + o_desc), so[0], d + o_cache, 0.f, P->check_orientation, RunIf{}, ap, h + o_prob1);
  ap.min_matches = 0; ap.is_retry = 1;
  orbs_fill_problem(f, 0, nq, S->D.kp_has, d + o_qrec, reinterpret_cast<const uint32_t*>(d + o_desc), so[1], d + o_cache, 0.f, P->check_orientation, gate, ap, h + o_prob2);
  hipStream_t st = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, up_bytes, hipMemcpyHostToDevice, st));
  LLD_HIP_TRY(hipEventRecord(S->uploaded[0], st)); S->upload_pending[0] = true;
  // ---- kernels
  const int nmax = std::max(std::max(nt, S->nl), 64);
  hipLaunchKernelGGL(track_reset_kernel, dim3((nmax + 255) / 256), dim3(256), 0, st, S->D, reinterpret_cast<const double*>(d + o_pose),
                     reinterpret_cast<const lld_frame_view*>(d + o_view), reinterpret_cast<const LineTrackDevParams*>(d + o_lp));
  s = orbs_project_last_frame(st, f, view, nullptr, LF, P->direction, P->th_motion, d + o_qrec, RunIf{}); if (s) return s;
  s = orbs_launch(ctx, st, f, d + o_prob1); if (s) return s;
  if (wide) {
    s = orbs_project_last_frame(st, f, view, nullptr, LF, P->direction, 2.f * P->th_motion, d + o_qrec, gate); if (s) return s;
    s = orbs_launch(ctx, st, f, d + o_prob2); if (s) return s;
  }
  s = run_lines(f, st, P, n_map ? last_lines : nullptr, d, U, reinterpret_cast<const uint8_t*>(d + U.skip), d + o_lwork, 0); if (s) return s;
  s = run_pose(f, st, P, d + o_pwork, 0); if (s) return s;
  S->stage1_queued = true; S->n_in_view = 0;
  return LLD_OK;
}

int lld_frame_track_set_state(lld_frame* f, const lld_track_params* P, const lld_frame_view* view, const double* pose_qt, const lld_frame_held* held) {
  if (!f || !P || !view || !pose_qt || !held) return LLD_ERR_INVALID;
  const int nt = f->nt;
  if (view->n_levels != f->consts.n_levels) return LLD_ERR_INVALID;
  if (nt > 0 && (!held->kp_point_id || !held->kp_world_pos)) return LLD_ERR_INVALID;
  if (held->n_seen < 0 || held->n_seen > nt || (held->n_seen > 0 && !held->seen_point_id)) return LLD_ERR_INVALID;   // the outlier discard marks at most one MapPoint per keypoint
  lld_ctx* ctx = f->ctx;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  int s = ensure_state(f); if (s) return s;
  lld_frame_track_state* S = f->track;
  const int nl = S->nl;
  if (nl > 0 && held->ln_line_id && (!held->ln_x0 || !held->ln_dir)) return LLD_ERR_INVALID;
  int n_held_lines = 0;
  if (nl > 0 && held->ln_line_id) for (int i = 0; i < nl; i++) n_held_lines += held->ln_line_id[i] >= 0;
  if (held->n_tracked < 0 || (held->n_tracked > 0 && !held->tracked_line_id) || n_held_lines + held->n_tracked > S->D.tracked_cap - nl) return LLD_ERR_INVALID;
  for (int k = 0; k < held->n_seen; k++) if (held->seen_point_id[k] < 0) return LLD_ERR_INVALID;
  for (int k = 0; k < held->n_tracked; k++) if (held->tracked_line_id[k] < 0) return LLD_ERR_INVALID;
  fill_consts(S, f, P, view);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  const int n_trk = n_held_lines + held->n_tracked;
  const size_t o_pose = take(7 * 8), o_lp = take(sizeof(LineTrackDevParams)), o_view = take(sizeof(lld_frame_view));
  const size_t o_id = take((size_t)nt * 4), o_w = take((size_t)nt * 12), o_obs = take(nt), o_out = take(nt), o_seen = take((size_t)std::max(held->n_seen, 1) * 4);
  const size_t o_lid = take((size_t)nl * 4), o_lx0 = take((size_t)nl * 24), o_ldir = take((size_t)nl * 24), o_lout = take(nl), o_trk = take((size_t)std::max(n_trk, 1) * 4);
  s = ensure_work(S, ctx, o); if (s) return s;
  s = ensure_stage(S, 0, o); if (s) return s;
  char* h = S->h_stage[0]; char* d = S->d_work;
  std::memcpy(h + o_pose, pose_qt, 7 * 8);
  line_params_from_view(S->consts, *view, reinterpret_cast<LineTrackDevParams*>(h + o_lp));
  std::memcpy(h + o_view, view, sizeof(lld_frame_view));
  if (nt) {
    std::memcpy(h + o_id, held->kp_point_id, (size_t)nt * 4); std::memcpy(h + o_w, held->kp_world_pos, (size_t)nt * 12);
    if (held->kp_has_obs) std::memcpy(h + o_obs, held->kp_has_obs, nt); else std::memset(h + o_obs, 1, nt);
    if (held->kp_outlier) std::memcpy(h + o_out, held->kp_outlier, nt); else std::memset(h + o_out, 0, nt);
  }
  if (held->n_seen) std::memcpy(h + o_seen, held->seen_point_id, (size_t)held->n_seen * 4);
  if (nl) {
    int32_t* trk = reinterpret_cast<int32_t*>(h + o_trk); int at = 0;
    if (held->ln_line_id) {
      std::memcpy(h + o_lid, held->ln_line_id, (size_t)nl * 4); std::memcpy(h + o_lx0, held->ln_x0, (size_t)nl * 24); std::memcpy(h + o_ldir, held->ln_dir, (size_t)nl * 24);
      for (int i = 0; i < nl; i++) if (held->ln_line_id[i] >= 0) trk[at++] = held->ln_line_id[i];      // a line the frame holds was tracked by it (tracked_last_id, :1117)
    } else {
      std::memset(h + o_lid, 0xff, (size_t)nl * 4); std::memset(h + o_lx0, 0, (size_t)nl * 24); std::memset(h + o_ldir, 0, (size_t)nl * 24);
    }
    if (held->ln_outlier) std::memcpy(h + o_lout, held->ln_outlier, nl); else std::memset(h + o_lout, 0, nl);
    for (int k = 0; k < held->n_tracked; k++) trk[at++] = held->tracked_line_id[k];
  }
  hipStream_t st = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, o, hipMemcpyHostToDevice, st));
  LLD_HIP_TRY(hipEventRecord(S->uploaded[0], st)); S->upload_pending[0] = true;
  const HeldUp U{reinterpret_cast<const int32_t*>(d + o_id), reinterpret_cast<const float*>(d + o_w), reinterpret_cast<const uint8_t*>(d + o_obs), reinterpret_cast<const uint8_t*>(d + o_out),
                 reinterpret_cast<const int32_t*>(d + o_seen), held->n_seen, reinterpret_cast<const int32_t*>(d + o_lid), reinterpret_cast<const double*>(d + o_lx0),
                 reinterpret_cast<const double*>(d + o_ldir), reinterpret_cast<const uint8_t*>(d + o_lout), reinterpret_cast<const int32_t*>(d + o_trk), nl ? n_trk : 0};
  const int nmax = std::max(std::max(std::max(nt, nl), n_trk), 64);
  hipLaunchKernelGGL(track_load_kernel, dim3((nmax + 255) / 256), dim3(256), 0, st, S->D, U, reinterpret_cast<const double*>(d + o_pose),
                     reinterpret_cast<const lld_frame_view*>(d + o_view), reinterpret_cast<const LineTrackDevParams*>(d + o_lp));
  LLD_HIP_TRY(hipGetLastError());
  S->stage1_queued = true; S->n_in_view = 0;
  return LLD_OK;
}

int lld_frame_track_local_map(lld_frame* f, const lld_track_params* P, const lld_map_points* mp, const int32_t* point_id, const lld_map_lines* local_lines) {
  if (!f || !P || !mp) return LLD_ERR_INVALID;
  lld_frame_track_state* S = f->track;
  if (!S || !S->stage1_queued) return LLD_ERR_INVALID;                        // the frame's pose and MapPoints come from lld_frame_track_motion_model
  const int nq = mp->n, nt = f->nt;
  if (nq < 0 || (nq > 0 && (!mp->world_pos || !mp->normal || !mp->max_distance || !mp->min_distance || !mp->desc || !point_id))) return LLD_ERR_INVALID;
  lld_ctx* ctx = f->ctx;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  int s = check_lines(S, local_lines); if (s) return s;
  const int n_map = (local_lines && S->nl > 0) ? local_lines->n : 0;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
  // stage 1's kernels may still be reading their part of the work block: stage 2 lives behind the high-water mark of stage 1's
  // layout (both stages are laid out from the frame's sizes and this call's, so the block is simply split in two halves here)
  const size_t o_prob = take(orbs_problem_bytes());
  const size_t o_pos = take((size_t)nq * 12), o_nrm = take((size_t)nq * 12), o_maxd = take((size_t)nq * 4), o_mind = take((size_t)nq * 4), o_obs = take(nq), o_skip = take(nq);
  const size_t o_desc = take((size_t)nq * 32), o_id = take((size_t)nq * 4);
  LinesUp U{}; lay_lines(o, n_map, S->dim, &U, nullptr);
  const size_t up_bytes = o;
  U.matches = take((size_t)std::max(n_map, 1) * 4);
  const size_t o_skip2 = take(nq), o_lskip2 = take(std::max(n_map, 1)), o_inview = take(std::max(nq, 1));
  size_t o_so[6];
  o_so[0] = take((size_t)nq * 4); o_so[1] = take((size_t)nq * 4); o_so[2] = take((size_t)nq * 4); o_so[3] = take(nq); o_so[4] = take((size_t)nt * 4); o_so[5] = take(16);
  const size_t o_qrec = take(orbs_qrec_bytes(nq)), o_cache = take(orbs_cache_bytes(nq));
  const size_t o_lwork = take(line_track_work_bytes(n_map, S->nl)), o_pwork = take(pose_track_work_bytes(nt, S->nl));
  // (one work block serves both stages one after the other: the stream orders stage 2's upload behind stage 1's last kernel)
  s = ensure_work(S, ctx, o); if (s) return s;
  s = ensure_stage(S, 1, up_bytes); if (s) return s;
  char* h = S->h_stage[1]; char* d = S->d_work;
  if (nq) {
    std::memcpy(h + o_pos, mp->world_pos, (size_t)nq * 12); std::memcpy(h + o_nrm, mp->normal, (size_t)nq * 12);
    std::memcpy(h + o_maxd, mp->max_distance, (size_t)nq * 4); std::memcpy(h + o_mind, mp->min_distance, (size_t)nq * 4);
    if (mp->has_obs) std::memcpy(h + o_obs, mp->has_obs, nq); else std::memset(h + o_obs, 1, nq);
    if (mp->skip) std::memcpy(h + o_skip, mp->skip, nq); else std::memset(h + o_skip, 0, nq);
    std::memcpy(h + o_desc, mp->desc, (size_t)nq * 32); std::memcpy(h + o_id, point_id, (size_t)nq * 4);
  }
  if (n_map) pack_lines(h, U, local_lines, S->dim);
  const SearchOut so{reinterpret_cast<int32_t*>(d + o_so[0]), reinterpret_cast<int32_t*>(d + o_so[1]), reinterpret_cast<int32_t*>(d + o_so[2]),
                     reinterpret_cast<uint8_t*>(d + o_so[3]), reinterpret_cast<int32_t*>(d + o_so[4]), reinterpret_cast<int32_t*>(d + o_so[5])};
  const MapPointsDev MP{nq, reinterpret_cast<const float*>(d + o_pos), reinterpret_cast<const float*>(d + o_nrm), reinterpret_cast<const float*>(d + o_maxd),
                        reinterpret_cast<const float*>(d + o_mind), reinterpret_cast<const uint8_t*>(d + o_obs), reinterpret_cast<const uint8_t*>(d + o_skip2)};
  const ApplyDev ap{S->D.kp_has, S->D.kp_world, S->D.kp_id, S->D.kp_obs, MP.pos, reinterpret_cast<const int32_t*>(d + o_id), MP.has_obs, S->D.rec_h[1]->i + RI_SEARCH1, 0, 0};
  orbs_fill_problem(f, 1, nq, S->D.kp_has, d + o_qrec, reinterpret_cast<const uint32_t*>(d + o_desc), so, d + o_cache, P->nnratio_local, 0, RunIf{}, ap, h + o_prob);
  hipStream_t st = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, up_bytes, hipMemcpyHostToDevice, st));
  LLD_HIP_TRY(hipEventRecord(S->uploaded[1], st)); S->upload_pending[1] = true;
  // ---- kernels
  {
    auto pow2 = [](unsigned n) { unsigned p = 64; while (p < n) p <<= 1; return p; };
    const unsigned psize = pow2(4u * (unsigned)std::max(nt, 1)), lsize = pow2(2u * (unsigned)S->D.tracked_cap);     // held + discarded <= 2 nt ids: at most half full
    const size_t lds = ((size_t)psize + lsize) * 4;
    if (lds > 150 * 1024) return LLD_ERR_UNSUPPORTED;
    if (lds > 48 * 1024) LLD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&track_mark_seen_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (nq + n_map > 0)
      hipLaunchKernelGGL(track_mark_seen_kernel, dim3(1), dim3(kSeenThreads), lds, st, S->D, nq, reinterpret_cast<const int32_t*>(d + o_id),
                         reinterpret_cast<const uint8_t*>(d + o_skip), reinterpret_cast<uint8_t*>(d + o_skip2), n_map, reinterpret_cast<const int32_t*>(d + U.id),
                         reinterpret_cast<const uint8_t*>(d + U.skip), reinterpret_cast<uint8_t*>(d + o_lskip2), psize - 1, lsize - 1);
  }
  S->in_view_off = o_inview; S->n_in_view = nq;
  s = orbs_project_local_points(st, f, nullptr, S->D.view, MP, P->viewing_cos_limit, P->th_local, d + o_qrec, reinterpret_cast<uint8_t*>(d + o_inview),
                                S->D.rec_h[1]->i + RI_IN_VIEW); if (s) return s;
  s = orbs_launch(ctx, st, f, d + o_prob); if (s) return s;
  s = run_lines(f, st, P, n_map ? local_lines : nullptr, d, U, reinterpret_cast<const uint8_t*>(d + o_lskip2), d + o_lwork, 1); if (s) return s;
  s = run_pose(f, st, P, d + o_pwork, 1); if (s) return s;
  return LLD_OK;
}

int lld_frame_track_download(lld_frame* f, lld_track_result* r1, lld_track_result* r2) {
  if (!f || !f->track || !f->track->stage1_queued) return LLD_ERR_INVALID;
  lld_frame_track_state* S = f->track; lld_ctx* ctx = f->ctx;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  const bool want_view = r2 && r2->mp_in_view && S->n_in_view > 0;
  const size_t need = S->rec_bytes + (want_view ? (size_t)S->n_in_view : 0);
  if (need > S->h_rec_bytes) {
    if (S->h_rec) LLD_HIP_TRY(hipHostFree(S->h_rec));
    S->h_rec = nullptr; S->h_rec_bytes = 0;
    LLD_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S->h_rec), need + 4096, hipHostMallocDefault));
    S->h_rec_bytes = need + 4096;
  }
  LLD_HIP_TRY(hipMemcpyAsync(S->h_rec, S->d_state + S->rec_off, S->rec_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (want_view) LLD_HIP_TRY(hipMemcpyAsync(S->h_rec + S->rec_bytes, S->d_work + S->in_view_off, (size_t)S->n_in_view, hipMemcpyDeviceToHost, ctx->stream));
  LLD_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (want_view) std::memcpy(r2->mp_in_view, S->h_rec + S->rec_bytes, (size_t)S->n_in_view);
  const TrackDev& D = S->D;
  lld_track_result* rr[2] = {r1, r2};
  for (int s = 0; s < 2; s++) {
    lld_track_result* r = rr[s];
    if (!r) continue;
    auto at = [&](const void* dev) { return S->h_rec + (static_cast<const char*>(dev) - (S->d_state + S->rec_off)); };
    const RecHeader* H = reinterpret_cast<const RecHeader*>(at(D.rec_h[s]));
    std::memcpy(r->pose_qt, H->pose_qt, sizeof r->pose_qt); r->chi2 = H->chi2;
    r->n_inliers = H->i[RI_INL]; r->lm_iterations = H->i[RI_ITS]; r->lm_trials = H->i[RI_TRIALS]; r->n_edges = H->i[RI_EDGES];
    r->n_search_first = H->i[RI_SEARCH1]; r->n_search = H->i[RI_SEARCH]; r->used_wide = H->i[RI_WIDE]; r->n_points = H->i[RI_POINTS];
    r->n_points_map = H->i[RI_POINTS_MAP]; r->n_lines_matched = H->i[RI_LINES_MATCHED]; r->n_lines = H->i[RI_LINES]; r->n_discarded = H->i[RI_DISCARDED];
    r->n_point_edges = H->i[RI_POINT_EDGES]; r->n_in_view = H->i[RI_IN_VIEW];
    if (r->kp_point_id && D.nt) std::memcpy(r->kp_point_id, at(D.rec_kp_id[s]), (size_t)D.nt * 4);
    if (r->kp_outlier && D.nt) std::memcpy(r->kp_outlier, at(D.rec_kp_out[s]), D.nt);
    if (r->ln_line_id && D.nl) std::memcpy(r->ln_line_id, at(D.rec_ln_id[s]), (size_t)D.nl * 4);
    if (r->ln_outlier && D.nl) std::memcpy(r->ln_outlier, at(D.rec_ln_out[s]), D.nl);
  }
  return LLD_OK;
}

}  // extern "C"

