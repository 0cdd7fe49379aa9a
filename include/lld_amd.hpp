// lld_amd.hpp — header-only C++ host layer over the C ABI (include/lld_amd.h).
//
// The reference's hot path is entered through C++ static/member functions on live SLAM objects
// (include/Optimizer.h:49-50, include/ORBmatcher.h:41-83, include/TwoFrameLineMatcher.h:31-42).  This header mirrors those
// names and argument meanings on flat, owning containers, so the adapter in INTEGRATION.md shrinks to "fill the vectors,
// call, scatter".  It needs nothing but the C++11 standard library and liblld_amd.so.
#ifndef LLD_AMD_HPP
#define LLD_AMD_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "lld_amd.h"

namespace lld_amd {

inline void check(int status, const char* what) {
  if (status != LLD_OK) throw std::runtime_error(std::string(what) + ": " + lld_status_string(status));
}

// One per host thread (Tracking, LocalMapping): a HIP device + stream.  Throws when no GPU is present — there is no CPU fallback.
class Context {
 public:
  explicit Context(int device = 0) { check(lld_ctx_create(device, &h_), "lld_ctx_create"); }
  ~Context() { lld_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  lld_ctx* get() const { return h_; }
 private:
  lld_ctx* h_ = nullptr;
};

// Flat local-BA window (what Optimizer.cc:938-1218 gathers); see lld_ba_window for the meaning of every array.
struct BAWindow {
  lld_camera cam{};
  int n_free_cams = 0;
  std::vector<double> cam_qt, pt_xyz, pt_obs_uvr, pt_obs_inv_sigma2, line_x0, line_dir, ln_obs_left, ln_obs_right;
  std::vector<int32_t> pt_obs_start{0}, pt_obs_cam, ln_obs_start{0}, ln_obs_cam, ln_obs_octave;
  int n_cams() const { return (int)(cam_qt.size() / 7); }
  int n_points() const { return (int)(pt_xyz.size() / 3); }
  int n_lines() const { return (int)(line_x0.size() / 3); }
  lld_ba_window view() const {
    lld_ba_window w{};
    w.cam = cam; w.n_cams = n_cams(); w.n_free_cams = n_free_cams; w.cam_qt = cam_qt.data();
    w.n_points = n_points(); w.pt_xyz = pt_xyz.data(); w.pt_obs_start = pt_obs_start.data();
    w.n_pt_obs = (int)pt_obs_cam.size(); w.pt_obs_cam = pt_obs_cam.data(); w.pt_obs_uvr = pt_obs_uvr.data();
    w.pt_obs_inv_sigma2 = pt_obs_inv_sigma2.data();
    w.n_lines = n_lines(); w.line_x0 = line_x0.data(); w.line_dir = line_dir.data(); w.ln_obs_start = ln_obs_start.data();
    w.n_ln_obs = (int)ln_obs_cam.size(); w.ln_obs_cam = ln_obs_cam.data(); w.ln_obs_left = ln_obs_left.data();
    w.ln_obs_right = ln_obs_right.data(); w.ln_obs_octave = ln_obs_octave.data();
    return w;
  }
};

struct BAOutput {
  std::vector<double> cam_qt, pt_xyz, line_x0, line_dir;
  std::vector<uint8_t> pt_obs_outlier, ln_edge_outlier, line_removed;   // vToErase / GetLineData outliers / deleted lines
  lld_ba_stats stats{};
};

struct PoseFrame {
  lld_camera cam{};
  double pose_qt[7] = {0, 0, 0, 1, 0, 0, 0};
  std::vector<double> pt_xw, pt_uvr, pt_inv_sigma2, ln_x0, ln_dir, ln_left, ln_right;
  std::vector<int32_t> ln_octave;
  std::vector<int32_t> ln_frame_index;                // index of each line in the frame's mvLinesLeft (empty: 0..n-1), see lld_pose_problem
  std::vector<uint8_t> mvbOutlier, mvbOutlierLines;   // filled by PoseOptimization
};

// Mirror of the reference's `class Optimizer` (include/Optimizer.h:43-61), hot-path members only.
class Optimizer {
 public:
  // void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, double gamma = 1.0)
  static BAOutput LocalBundleAdjustment(Context& ctx, const BAWindow& win, const bool* pbStopFlag = nullptr, double gamma = 1.0) {
    lld_ba_params p; lld_ba_params_default(&p); p.gamma = gamma;
    return Solve(ctx, win, p, pbStopFlag);
  }
  // void static GlobalBundleAdjustment(Map* pMap, int nIterations=5, bool* pbStopFlag=NULL, const unsigned long nLoopKF=0, const bool bRobust = true)
  // `win` holds the whole map (every keyframe but mnId==0 free); one optimize(nIterations), nothing is erased (src/Optimizer.cc:312-559)
  static BAOutput GlobalBundleAdjustment(Context& ctx, const BAWindow& win, int nIterations = 5, const bool* pbStopFlag = nullptr, bool bRobust = true) {
    lld_ba_params p; lld_ba_params_default(&p); p.protocol = 1; p.its_round1 = nIterations; p.robust_points = bRobust ? 1 : 0;
    return Solve(ctx, win, p, pbStopFlag);
  }
  // int static OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12, const float th2, const bool bFixScale)
  // `pair` holds the correspondences that pass the loop's tests (src/Optimizer.cc:1704-1786); S12 is updated in place, dropped[i] = 1
  // means vpMatches1[idx] = NULL; returns nIn
  static int OptimizeSim3(Context& ctx, lld_sim3_problem& pair, std::vector<uint8_t>& dropped, float th2, bool bFixScale) {
    lld_sim3_params p; lld_sim3_params_default(&p); p.th2 = th2; p.fix_scale = bFixScale ? 1 : 0;
    dropped.assign(pair.n > 0 ? pair.n : 1, 0);
    lld_sim3_result r{}; r.dropped = dropped.data();
    check(lld_optimize_sim3(ctx.get(), &pair, &p, &r), "lld_optimize_sim3");
    dropped.resize(pair.n);
    for (int k = 0; k < 4; k++) pair.s12_q[k] = r.s12_q[k];
    for (int k = 0; k < 3; k++) pair.s12_t[k] = r.s12_t[k];
    pair.s12_s = r.s12_s;
    return r.n_inliers;
  }
  // void static OptimizeEssentialGraph(Map*, KeyFrame* pLoopKF, KeyFrame* pCurKF, NonCorrectedSim3, CorrectedSim3, LoopConnections, bFixScale)
  // `graph` is what the reference hands to g2o (src/Optimizer.cc:1413-1585: vertices vScw, fixed[pLoopKF] = 1, edges (nIDi, nIDj, Sji));
  // returns CorrectedSiw per vertex (8 doubles each), from which the caller recovers the SE3 poses and corrects the MapPoints (:1593-1653)
  static std::vector<double> OptimizeEssentialGraph(Context& ctx, const lld_pose_graph& graph, bool bFixScale) {
    lld_pose_graph_params p; lld_pose_graph_params_default(&p); p.fix_scale = bFixScale ? 1 : 0;
    std::vector<double> corrected(8 * (size_t)(graph.n_vertices > 0 ? graph.n_vertices : 1));
    lld_pose_graph_result r{}; r.sim3 = corrected.data();
    check(lld_optimize_essential_graph(ctx.get(), &graph, &p, &r), "lld_optimize_essential_graph");
    corrected.resize(8 * (size_t)graph.n_vertices);
    return corrected;
  }
  static BAOutput Solve(Context& ctx, const BAWindow& win, const lld_ba_params& p, const bool* pbStopFlag = nullptr) {
    const lld_ba_window w = win.view();
    BAOutput o;
    o.cam_qt.resize(7 * (size_t)w.n_cams); o.pt_xyz.resize(3 * (size_t)w.n_points);
    o.line_x0.resize(3 * (size_t)w.n_lines); o.line_dir.resize(3 * (size_t)w.n_lines);
    o.pt_obs_outlier.resize(w.n_pt_obs); o.ln_edge_outlier.resize(2 * (size_t)w.n_ln_obs); o.line_removed.resize(w.n_lines);
    lld_ba_result r{};
    r.cam_qt = o.cam_qt.data(); r.pt_xyz = o.pt_xyz.data(); r.line_x0 = o.line_x0.data(); r.line_dir = o.line_dir.data();
    r.pt_obs_outlier = o.pt_obs_outlier.data(); r.ln_edge_outlier = o.ln_edge_outlier.data(); r.line_removed = o.line_removed.data();
    volatile int stop = (pbStopFlag && *pbStopFlag) ? 1 : 0;      // a live caller refreshes this from its bool
    check(lld_local_ba(ctx.get(), &w, &p, &stop, &r), "lld_local_ba");
    o.stats = r.stats;
    return o;
  }
  // int static PoseOptimization(Frame* pFrame, double gamma = 1.0): returns the inlier count, writes the pose and the flags
  static int PoseOptimization(Context& ctx, PoseFrame& f, double gamma = 1.0) {
    lld_pose_problem q{};
    q.cam = f.cam;
    for (int i = 0; i < 7; i++) q.pose_qt[i] = f.pose_qt[i];
    q.n_points = (int)(f.pt_xw.size() / 3); q.pt_xw = f.pt_xw.data(); q.pt_uvr = f.pt_uvr.data(); q.pt_inv_sigma2 = f.pt_inv_sigma2.data();
    q.n_lines = (int)(f.ln_x0.size() / 3); q.ln_x0 = f.ln_x0.data(); q.ln_dir = f.ln_dir.data(); q.ln_left = f.ln_left.data();
    q.ln_right = f.ln_right.data(); q.ln_octave = f.ln_octave.data();
    q.ln_frame_index = f.ln_frame_index.empty() ? nullptr : f.ln_frame_index.data();
    lld_pose_params p; lld_pose_params_default(&p); p.gamma = gamma;
    f.mvbOutlier.assign(q.n_points, 0); f.mvbOutlierLines.assign(q.n_lines, 0);
    lld_pose_result r{};
    r.pt_outlier = f.mvbOutlier.data(); r.ln_outlier = f.mvbOutlierLines.data();
    check(lld_pose_opt(ctx.get(), &q, &p, &r), "lld_pose_opt");
    for (int i = 0; i < 7; i++) f.pose_qt[i] = r.pose_qt[i];
    return r.n_inliers;
  }
};

// Mirror of `class ORBmatcher` (include/ORBmatcher.h:41-83): the distance + best/second-best core; the accept rules
// (TH_LOW / TH_HIGH / mfNNratio) stay with the caller as in the reference.
class ORBmatcher {
 public:
  static constexpr int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;     // src/ORBmatcher.cc:37-39
  ORBmatcher(Context& ctx, float nnratio = 0.6f, bool checkOri = true) : ctx_(ctx), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  struct Best2 { std::vector<int32_t> best_idx, best_dist, second_idx, second_dist; };
  // descriptors: nq x 8 / nt x 8 uint32 rows (cv::Mat CV_8U 32 bytes per row); mask: nq x nt bytes or empty
  Best2 BestTwo(const uint32_t* q, int nq, const uint32_t* t, int nt, const std::vector<uint8_t>& mask = {}) const {
    Best2 b; b.best_idx.resize(nq); b.best_dist.resize(nq); b.second_idx.resize(nq); b.second_dist.resize(nq);
    check(lld_match_hamming256(ctx_.get(), q, nq, t, nt, mask.empty() ? nullptr : mask.data(), b.best_idx.data(), b.best_dist.data(),
                               b.second_idx.data(), b.second_dist.data()), "lld_match_hamming256");
    return b;
  }
  // candidate lists in the reference's own order (Frame::GetFeaturesInArea, BoW nodes)
  Best2 BestTwo(const uint32_t* q, int nq, const uint32_t* t, int nt, const std::vector<int32_t>& cand_start, const std::vector<int32_t>& cand_idx) const {
    Best2 b; b.best_idx.resize(nq); b.best_dist.resize(nq); b.second_idx.resize(nq); b.second_dist.resize(nq);
    static const int32_t none = 0;
    check(lld_match_hamming256_csr(ctx_.get(), q, nq, t, nt, cand_start.data(), cand_idx.empty() ? &none : cand_idx.data(), b.best_idx.data(),
                                   b.best_dist.data(), b.second_idx.data(), b.second_dist.data()), "lld_match_hamming256_csr");
    return b;
  }
  // One whole Search* / Fuse / ComputeStereoMatches routine on the device: fill an lld_orb_search as INTEGRATION.md §4b shows
  // (candidate generator, gates, accept rule, `sequential`, `check_orientation`), get the matches and the routine's return value.
  struct SearchResult {
    std::vector<int32_t> match, best_dist, second_dist, owner;
    std::vector<uint8_t> removed;
    int n_matches = 0, rounds = 0;
  };
  SearchResult Search(lld_orb_search s) const {
    if (s.nnratio == 0.f) s.nnratio = mfNNratio;
    SearchResult r;
    r.match.resize(s.nq); r.best_dist.resize(s.nq); r.second_dist.resize(s.nq); r.removed.resize(s.nq); r.owner.resize(s.nt);
    lld_orb_search_result o{};
    o.match = r.match.data(); o.best_dist = r.best_dist.data(); o.second_dist = r.second_dist.data(); o.removed = r.removed.data();
    o.owner = r.owner.data();
    check(lld_orb_search_run(ctx_.get(), &s, &o), "lld_orb_search_run");
    r.n_matches = o.n_matches; r.rounds = o.rounds;
    return r;
  }
  // Tracking::SearchLocalPoints: Frame::isInFrustum for every local MapPoint + SearchByProjection(F, vpMapPoints, th), one call.
  // `frame` carries the keypoint side (nt, t_*), the grid constants and the scale table; in_view (may be null) receives mbTrackInView.
  SearchResult SearchLocalPoints(const lld_orb_search& frame, const lld_frame_view& view, const lld_map_points& points, float th = 1.0f,
                                 std::vector<uint8_t>* in_view = nullptr, float viewingCosLimit = 0.5f) const {
    SearchResult r;
    r.match.resize(points.n); r.best_dist.resize(points.n); r.second_dist.resize(points.n); r.removed.resize(points.n); r.owner.resize(frame.nt);
    lld_orb_search_result o{};
    o.match = r.match.data(); o.best_dist = r.best_dist.data(); o.second_dist = r.second_dist.data(); o.removed = r.removed.data();
    o.owner = r.owner.data();
    lld_frustum_result fr{};
    if (in_view) { in_view->assign(points.n, 0); fr.in_view = in_view->data(); }
    check(lld_orb_search_local_points(ctx_.get(), &frame, &view, &points, viewingCosLimit, th, mfNNratio, &fr, &o), "lld_orb_search_local_points");
    r.n_matches = o.n_matches; r.rounds = o.rounds;
    return r;
  }
  // SearchByProjection(Frame& Current, const Frame& Last, th, bMono) with the projection of the last frame's points on the device;
  // direction: +1 bForward, -1 bBackward, 0 neither (src/ORBmatcher.cc:1343-1350)
  SearchResult SearchByProjection(const lld_orb_search& currentFrame, const lld_frame_view& view, const lld_last_frame_points& last, int direction,
                                  float th) const {
    SearchResult r;
    r.match.resize(last.n); r.best_dist.resize(last.n); r.second_dist.resize(last.n); r.removed.resize(last.n); r.owner.resize(currentFrame.nt);
    lld_orb_search_result o{};
    o.match = r.match.data(); o.best_dist = r.best_dist.data(); o.second_dist = r.second_dist.data(); o.removed = r.removed.data();
    o.owner = r.owner.data();
    check(lld_orb_search_last_frame(ctx_.get(), &currentFrame, &view, &last, direction, th, mbCheckOrientation ? 1 : 0, nullptr, &o),
          "lld_orb_search_last_frame");
    r.n_matches = o.n_matches; r.rounds = o.rounds;
    return r;
  }
  // The relocalisation / loop-closing matchers with their projection loops on the device (include/lld_amd.h, LLD_ORB_PROJ_*):
  //   SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)   src/ORBmatcher.cc:1472-1599   (kfAngle = pKF->mvKeysUn[i].angle)
  SearchResult SearchByProjection(const lld_orb_search& currentFrame, const lld_frame_view& view, const lld_map_points& kfPoints,
                                  const float* kfAngle, float th, int ORBdist) const {
    lld_orb_projection pr{}; pr.routine = LLD_ORB_PROJ_RELOC; pr.th = th; pr.accept_max = ORBdist; pr.check_orientation = mbCheckOrientation ? 1 : 0;
    return Projected(currentFrame, view, kfPoints, kfAngle, pr);
  }
  //   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)         :290-403   (view = the decomposed Scw, frame.t_occupied = vpMatched[idx] != NULL)
  SearchResult SearchByProjection(const lld_orb_search& keyFrame, const lld_frame_view& scwView, const lld_map_points& points, int th) const {
    lld_orb_projection pr{}; pr.routine = LLD_ORB_PROJ_KF_SIM3; pr.th = (float)th;
    return Projected(keyFrame, scwView, points, nullptr, pr);
  }
  //   Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)                  :977-1100  (match[i] = bestIdx; the bookkeeping of :1078-1093 stays here)
  SearchResult Fuse(const lld_orb_search& keyFrame, const lld_frame_view& scwView, const lld_map_points& points, float th) const {
    lld_orb_projection pr{}; pr.routine = LLD_ORB_PROJ_FUSE_SIM3; pr.th = th;
    return Projected(keyFrame, scwView, points, nullptr, pr);
  }
  //   SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)            :1102-1326 (sR12 = s12*R12, sR21 = (1.0/s12)*R12.t(), t21 = -sR21*t12
  //   formed by the caller, :1121-1124); vnMatch12[i1] = KF2 keypoint index or -1; returns nFound
  int SearchBySim3(const lld_orb_search& kf1, const lld_frame_view& view1, const lld_map_points& points1, const lld_orb_search& kf2,
                   const lld_frame_view& view2, const lld_map_points& points2, const float sR12[9], const float t12[3], const float sR21[9],
                   const float t21[3], float th, std::vector<int32_t>& vnMatch12) const {
    vnMatch12.assign(points1.n, -1);
    int32_t found = 0, none = -1;
    check(lld_orb_search_by_sim3(ctx_.get(), &kf1, &view1, &points1, &kf2, &view2, &points2, sR12, t12, sR21, t21, th,
                                 points1.n ? vnMatch12.data() : &none, &found), "lld_orb_search_by_sim3");
    return found;
  }
  // SearchForInitialization(Frame& F1, Frame& F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:405-520).  `f2` carries
  // the keypoints of F2 (nt, t_desc, t_xy, t_octave, t_angle) and the grid constants; the arrays of F1 have n1 rows.  vbPrevMatched
  // ([n1][2], in / out) and vnMatches12 are updated as the reference does; returns nmatches.
  int SearchForInitialization(lld_orb_search f2, int n1, const uint32_t* desc1, const int32_t* octave1, const float* angle1,
                              std::vector<float>& vbPrevMatched, std::vector<int32_t>& vnMatches12, int windowSize = 10) const {
    std::vector<uint8_t> valid(n1); std::vector<float> radius(n1, (float)windowSize); std::vector<int32_t> level(n1, 0);
    for (int i = 0; i < n1; i++) valid[i] = octave1[i] <= 0;                    // `if(level1>0) continue;` (:423-425)
    f2.nq = n1; f2.q_desc = desc1; f2.q_valid = valid.data(); f2.q_uv = vbPrevMatched.data(); f2.q_radius = radius.data();
    f2.q_level_min = level.data(); f2.q_level_max = level.data(); f2.q_angle = angle1;
    f2.candidates = LLD_ORB_CAND_GRID; f2.gates = LLD_ORB_GATE_LEVEL; f2.accept_max = TH_LOW; f2.ratio_mode = 1; f2.nnratio = mfNNratio;
    f2.sequential = 2; f2.check_orientation = mbCheckOrientation ? 1 : 0; f2.tie_last = 0;
    const SearchResult r = Search(f2);
    vnMatches12.assign(n1, -1);
    for (int i = 0; i < n1; i++)
      if (r.match[i] >= 0 && !r.removed[i]) {
        vnMatches12[i] = r.match[i];
        vbPrevMatched[2 * i] = f2.t_xy[2 * r.match[i]]; vbPrevMatched[2 * i + 1] = f2.t_xy[2 * r.match[i] + 1];   // :513-516
      }
    return r.n_matches;
  }
  // Frame::ComputeStereoMatches as a whole (src/Frame.cc:530-704): row-band Hamming search, 11x11 SAD refinement on the image
  // pyramids, median cut.  Fills mvuRight / mvDepth; returns the number of stereo keypoints kept.
  int ComputeStereoMatches(const lld_keypoints& left, const lld_keypoints& right, const lld_stereo_pyramids& pyramids, float mb, float mbf,
                           std::vector<float>& mvuRight, std::vector<float>& mvDepth) const {
    mvuRight.assign(left.n, -1.0f); mvDepth.assign(left.n, -1.0f);
    lld_stereo_result o{};
    o.u_right = mvuRight.data(); o.depth = mvDepth.data();
    check(lld_compute_stereo_matches(ctx_.get(), &left, &right, &pyramids, mb, mbf, &o), "lld_compute_stereo_matches");
    return o.n_matches;
  }
 private:
  SearchResult Projected(const lld_orb_search& frame, const lld_frame_view& view, const lld_map_points& points, const float* angle,
                         const lld_orb_projection& pr) const {
    SearchResult r;
    r.match.resize(points.n); r.best_dist.resize(points.n); r.second_dist.resize(points.n); r.removed.resize(points.n); r.owner.resize(frame.nt);
    lld_orb_search_result o{};
    o.match = r.match.data(); o.best_dist = r.best_dist.data(); o.second_dist = r.second_dist.data(); o.removed = r.removed.data();
    o.owner = r.owner.data();
    check(lld_orb_search_projected(ctx_.get(), &frame, &view, &points, angle, &pr, nullptr, nullptr, &o), "lld_orb_search_projected");
    r.n_matches = o.n_matches; r.rounds = o.rounds;
    return r;
  }
  Context& ctx_;
 public:
  float mfNNratio; bool mbCheckOrientation;
};

// Mirror of `class TwoFrameLineMatcher` (include/TwoFrameLineMatcher.h:31-42): the caller supplies CheckLinePair's geometric
// gates as a byte matrix; the descriptor distance, the running strict minimum under tau and the greedy masking run on the GPU.
class TwoFrameLineMatcher {
 public:
  TwoFrameLineMatcher(Context& ctx, double tau) : ctx_(ctx), tau_(tau) {}
  // TwoFrameLineMatcher(K, b, tau, minLineLength, lineMatcher): the whole MatchLines, gates included, on the device
  TwoFrameLineMatcher(Context& ctx, const double K[9], double b, double tau, int minLineLength) : ctx_(ctx), tau_(tau) {
    for (int i = 0; i < 9; i++) p_.K[i] = K[i];
    p_.b = b; p_.tau = tau; p_.min_line_length = minLineLength; p_.is_stereo = 1; has_geometry_ = true;
  }
  // lines / other_lines: [n][4] startPointX, startPointY, endPointX, endPointY of the KeyLines
  void MatchLines(const float* lines, const int32_t* octaves, int nLeft, const float* other_lines, const int32_t* other_octaves, int nRight,
                  const float* descsLeft, const float* descsRight, int dim, std::vector<int>* desc_matches) const {
    if (!has_geometry_) throw std::logic_error("TwoFrameLineMatcher constructed without K / b");
    desc_matches->assign(nLeft, -1);
    check(lld_line_match_stereo(ctx_.get(), &p_, lines, octaves, descsLeft, nLeft, other_lines, other_octaves, descsRight, nRight, dim,
                                desc_matches->data(), nullptr, nullptr), "lld_line_match_stereo");
  }
  void MatchLines(const float* descsLeft, int nLeft, const float* descsRight, int nRight, int dim, const std::vector<uint8_t>& gate,
                  std::vector<int>* desc_matches) const {
    desc_matches->assign(nLeft, -1);
    check(lld_line_match_greedy(ctx_.get(), descsLeft, nLeft, descsRight, nRight, dim, gate.empty() ? nullptr : gate.data(), tau_,
                                desc_matches->data(), nullptr), "lld_line_match_greedy");
  }
 private:
  Context& ctx_;
  double tau_;
  lld_line_stereo_params p_{};
  bool has_geometry_ = false;
};

// The line half of `class Tracking` (include/Tracking.h) that runs on the device: the two temporal line matchers.
struct MapLineSet {            // what AddLinesFrom reads off its MapLine* list (GetMinimalPos, GetMainPoints3D, descriptor row, skip rule :1023-1034)
  std::vector<double> X0, dir, X1, X2;      // [n][3] each
  std::vector<uint8_t> skip;                 // [n] or empty
  std::vector<float> desc;                   // [n][dim]
  int size() const { return (int)(X0.size() / 3); }
};
struct FrameLines {            // the line members of a stereo Frame
  std::vector<float> left, right;            // [n][4] / [n_right][4] startPointX, startPointY, endPointX, endPointY (mvLinesLeft / mvLinesRight)
  std::vector<int32_t> left_octave, line_matches;   // [n]
  std::vector<uint8_t> occupied;             // [n] mvpMapLines[i] != NULL (AddLinesFrom) / tracked-in-this-frame (last frame of MatchLinesLastKF), or empty
  std::vector<float> desc;                   // [n][dim] mDescriptorsLines
  int size() const { return (int)(left.size() / 4); }
};
class Tracking {
 public:
  Tracking(Context& ctx, const double K[9], double b, double mnMaxX, double mnMaxY, double mdThr, bool monocular = false)
      : ctx_(ctx), b_(b), sx_(1.0 / mnMaxX), sy_(1.0 / mnMaxY), mdThr_(mdThr), mono_(monocular) { for (int i = 0; i < 9; i++) K_[i] = K[i]; }
  // Tracking::AddLinesFrom (src/Tracking.cc:996-1124): matches[i] = line of `frame` given to map line i, or -1 (the caller sets
  // frame->mvpMapLines[matches[i]] and tracked_last_id).  T_curr: camera-to-world, row-major 4x4.
  void AddLinesFrom(const MapLineSet& lines_last, const double T_curr[16], double thrReprojLineBase, const FrameLines& frame, int dim,
                    std::vector<int>* matches, bool use_grid = true) const {
    lld_line_track_params p{};
    for (int i = 0; i < 9; i++) p.K[i] = K_[i];
    for (int i = 0; i < 16; i++) p.T_curr[i] = T_curr[i];
    p.b = b_; p.thr_reproj_base = thrReprojLineBase; p.md_thr = mdThr_; p.sx = sx_; p.sy = sy_; p.monocular = mono_; p.use_grid = use_grid;
    matches->assign(lines_last.size(), -1);
    check(lld_line_track_match(ctx_.get(), &p, lines_last.size(), lines_last.X0.data(), lines_last.dir.data(), lines_last.X1.data(), lines_last.X2.data(),
                               lines_last.skip.empty() ? nullptr : lines_last.skip.data(), lines_last.desc.data(), frame.size(), frame.left.data(),
                               frame.left_octave.data(), (int)(frame.right.size() / 4), frame.right.data(), frame.line_matches.data(),
                               frame.occupied.empty() ? nullptr : frame.occupied.data(), frame.desc.data(), dim, matches->data(), nullptr, nullptr),
          "lld_line_track_match");
  }
  // Tracking::MatchLinesLastKF (src/Tracking.cc:1449-1611): created[i] != 0 -> the reference constructs MapLine(X0[i], dir[i]) for line i
  // of the current frame; match_last[i] is the line of the last frame it was matched with.  last.occupied plays last_skip (:1517-1520).
  void MatchLinesLastKF(const double T_curr[16], const double T_last[16], const FrameLines& current, const FrameLines& last, int dim,
                        std::vector<int>* match_last, std::vector<uint8_t>* created, std::vector<double>* X0, std::vector<double>* dir,
                        double thrReprojLineBase = 6.0, bool use_grid = true) const {
    lld_line_lastkf_params p{};
    for (int i = 0; i < 9; i++) p.K[i] = K_[i];
    for (int i = 0; i < 16; i++) { p.T_curr[i] = T_curr[i]; p.T_last[i] = T_last[i]; }
    p.b = b_; p.thr_reproj_base = thrReprojLineBase; p.md_thr = mdThr_; p.sx = sx_; p.sy = sy_; p.use_grid = use_grid;
    const int n = current.size();
    match_last->assign(n, -1); created->assign(n, 0); X0->assign(3 * (size_t)n, 0.0); dir->assign(3 * (size_t)n, 0.0);
    check(lld_line_match_last_frame(ctx_.get(), &p, n, current.left.data(), (int)(current.right.size() / 4), current.right.data(), current.line_matches.data(),
                                    current.occupied.empty() ? nullptr : current.occupied.data(), current.desc.data(), last.size(), last.left.data(),
                                    last.left_octave.data(), (int)(last.right.size() / 4), last.right.data(), last.line_matches.data(),
                                    last.occupied.empty() ? nullptr : last.occupied.data(), last.desc.data(), dim, match_last->data(), created->data(),
                                    X0->data(), dir->data()), "lld_line_match_last_frame");
  }
 private:
  Context& ctx_;
  double K_[9]; double b_, sx_, sy_, mdThr_; bool mono_;
};

// The Tracking thread's per-frame chain on ONE device-resident Frame (lld_frame_track_*, include/lld_amd.h): what Tracking::TrackWithMotionModel
// (src/Tracking.cc:885-994) and Tracking::TrackLocalMap (:1126-1220) do to mCurrentFrame, with mvpMapPoints / mvbOutlier / mvpMapLines /
// mvbOutlierLines / mTcw living in HBM between the calls.  Both Track* calls only queue work; Download() is the one synchronisation.
struct TrackRecord {
  lld_track_result r{};
  std::vector<int32_t> kp_point_id, ln_line_id;
  std::vector<uint8_t> kp_outlier, ln_outlier;
  void bind(int nt, int nl) {
    kp_point_id.assign(nt, -1); kp_outlier.assign(nt, 0); ln_line_id.assign(nl, -1); ln_outlier.assign(nl, 0);
    r.kp_point_id = kp_point_id.data(); r.kp_outlier = kp_outlier.data(); r.ln_line_id = ln_line_id.data(); r.ln_outlier = ln_outlier.data();
  }
};
class TrackedFrame {
 public:
  // keypoints: the keypoint side of an lld_orb_search (as for lld_frame_create); lines: NULL for a frame without lines
  TrackedFrame(Context& ctx, const lld_orb_search& keypoints, const lld_frame_lines* lines) : nt_(keypoints.nt), nl_(lines ? lines->n_left : 0) {
    check(lld_frame_create(ctx.get(), &keypoints, &f_), "lld_frame_create");
    const int st = lld_frame_set_lines(f_, lines);
    if (st != LLD_OK) { lld_frame_destroy(f_); f_ = nullptr; check(st, "lld_frame_set_lines"); }
    lld_track_params_default(&params);
  }
  ~TrackedFrame() { if (f_) lld_frame_destroy(f_); }
  TrackedFrame(const TrackedFrame&) = delete;
  TrackedFrame& operator=(const TrackedFrame&) = delete;
  lld_track_params params;
  // Tcw: the predicted pose mVelocity * mLastFrame.mTcw as the float matrix the Frame holds; `view`: its UpdatePoseMatrices
  void TrackWithMotionModel(const lld_frame_view& view, const float Tcw[16], const lld_last_frame_points& last, const int32_t* last_ids, const lld_map_lines* last_lines) {
    double qt[7];
    lld_se3_from_tcw_f32(Tcw, qt);                       // Converter::toSE3Quat(pFrame->mTcw)
    check(lld_frame_track_motion_model(f_, &params, &view, qt, &last, last_ids, last_lines), "lld_frame_track_motion_model");
  }
  // stage 1 ran elsewhere (TrackReferenceKeyFrame / Relocalization): the frame's pose and what it holds, then TrackLocalMap as usual
  void SetState(const lld_frame_view& view, const float Tcw[16], const lld_frame_held& held) {
    double qt[7];
    lld_se3_from_tcw_f32(Tcw, qt);
    check(lld_frame_track_set_state(f_, &params, &view, qt, &held), "lld_frame_track_set_state");
  }
  void TrackLocalMap(const lld_map_points& local_points, const int32_t* ids, const lld_map_lines* local_lines) {
    check(lld_frame_track_local_map(f_, &params, &local_points, ids, local_lines), "lld_frame_track_local_map");
  }
  void Download(TrackRecord* stage1, TrackRecord* stage2) {
    if (stage1) stage1->bind(nt_, nl_);
    if (stage2) stage2->bind(nt_, nl_);
    check(lld_frame_track_download(f_, stage1 ? &stage1->r : nullptr, stage2 ? &stage2->r : nullptr), "lld_frame_track_download");
  }
 private:
  lld_frame* f_ = nullptr;
  int nt_, nl_;
};

}  // namespace lld_amd
#endif
