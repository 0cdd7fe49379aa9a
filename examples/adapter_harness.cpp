// adapter_harness.cpp — builds a live object graph (KeyFrame / MapPoint / MapLine / Frame, adapters/lld_slam_objects.h) from a flat
// problem file, runs the COMPILED host adapters (adapters/lld_optimizer_adapter.cc) on it the way LocalMapping::Run / Tracking would,
// and dumps (1) the flat problem the adapter gathered, (2) what the library returned, (3) the state of the objects after the
// write-back, with the bookkeeping the test needs to map gathered indices to the input's (tests/test_cpp_adapter.py).
//
//   adapter_harness ba   <in> <out> [seed]    Optimizer::LocalBundleAdjustment(pKF, &mbAbortBA, pMap, gamma)
//   adapter_harness pose <in> <out> [seed]    Optimizer::PoseOptimization(&mCurrentFrame, gamma)
//
// The object graph is deliberately awkward: keyframes are allocated in shuffled order (std::map<KeyFrame*, ...> iterates by ADDRESS),
// their mnIds are a permutation of the input's camera order, one covisible keyframe has mnId 0 (fixed although local), and there are
// objects the reference skips (a bad MapPoint, a MapLine with two observations, a bad observing KeyFrame, keypoints / lines without
// a landmark).  Input files are the ones examples/harness.cpp reads (`ba` / `pose` modes).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <string>

#include "../adapters/lld_optimizer_adapter.h"
#include "../adapters/lld_matcher_adapter.h"
#include "../adapters/lld_line_adapter.h"
#include "../adapters/lld_tracking_adapter.h"

std::mutex lld_slam::MapPoint::mGlobalMutex;        // the doubles' static member (the real class defines its own, MapPoint.cc:30)

using namespace lld_slam;

namespace {

struct Reader {
  FILE* f;
  explicit Reader(const char* path) : f(std::fopen(path, "rb")) { if (!f) throw std::runtime_error(std::string("cannot open ") + path); }
  ~Reader() { std::fclose(f); }
  template <class T> void get(T* p, size_t n) { if (n && std::fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short read"); }
  template <class T> void get(std::vector<T>& v, size_t n) { v.resize(n); get(v.data(), n); }
};
struct Writer {
  FILE* f;
  explicit Writer(const char* path) : f(std::fopen(path, "wb")) { if (!f) throw std::runtime_error(std::string("cannot open ") + path); }
  ~Writer() { std::fclose(f); }
  template <class T> void put(const T* p, size_t n) { if (n && std::fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("short write"); }
  template <class T> void put(const std::vector<T>& v) { put(v.data(), v.size()); }
};

std::vector<float> level_table() { std::vector<float> t(8); lld_orb_inv_level_sigma2(1.2f, 8, t.data()); return t; }
int octave_of(const std::vector<float>& table, double inv_sigma2) {
  int best = 0; double d = 1e300;
  for (int o = 0; o < (int)table.size(); o++) { const double e = std::abs((double)table[o] - inv_sigma2); if (e < d) { d = e; best = o; } }
  return best;
}
Mat pose_mat(const double* qt7) { float T[16]; lld_se3_to_tcw_f32(qt7, T); return Mat(4, 4, T); }
Mat k_mat(float fx, float fy, float cx, float cy) { Mat K(3, 3); K.at<float>(0, 0) = fx; K.at<float>(1, 1) = fy; K.at<float>(0, 2) = cx; K.at<float>(1, 2) = cy; K.at<float>(2, 2) = 1.f; return K; }
KeyLine key_line(const double* seg, int octave) { KeyLine k; k.startPointX = (float)seg[0]; k.startPointY = (float)seg[1]; k.endPointX = (float)seg[2]; k.endPointY = (float)seg[3]; k.octave = octave; return k; }

int run_ba(const char* in, const char* out, unsigned seed) {
  Reader r(in);
  int32_t h[8]; r.get(h, 8);                   // n_cams n_free n_points n_pt_obs n_lines n_ln_obs stop 0
  double camg[6]; r.get(camg, 6);              // fx fy cx cy bf gamma
  lld_amd::BAWindow w;
  w.n_free_cams = h[1];
  r.get(w.cam_qt, 7 * (size_t)h[0]);
  r.get(w.pt_xyz, 3 * (size_t)h[2]); r.get(w.pt_obs_start, (size_t)h[2] + 1); r.get(w.pt_obs_cam, h[3]);
  r.get(w.pt_obs_uvr, 3 * (size_t)h[3]); r.get(w.pt_obs_inv_sigma2, h[3]);
  r.get(w.line_x0, 3 * (size_t)h[4]); r.get(w.line_dir, 3 * (size_t)h[4]); r.get(w.ln_obs_start, (size_t)h[4] + 1);
  r.get(w.ln_obs_cam, h[5]); r.get(w.ln_obs_left, 4 * (size_t)h[5]); r.get(w.ln_obs_right, 4 * (size_t)h[5]);
  r.get(w.ln_obs_octave, 2 * (size_t)h[5]);
  const int n_cams = h[0], n_free = h[1], n_pts = h[2], n_lns = h[4];
  if (n_free < 1) throw std::runtime_error("adapter-ba needs a free camera (pKF)");
  std::mt19937 rng(seed);
  const std::vector<float> table = level_table();

  // ---- keyframes: allocated in shuffled order, mnIds a permutation of the input order; the first fixed camera becomes mnId 0 and local
  std::vector<int> alloc_order(n_cams + 1);
  for (int i = 0; i <= n_cams; i++) alloc_order[i] = i;
  std::shuffle(alloc_order.begin(), alloc_order.end(), rng);
  std::vector<std::unique_ptr<KeyFrame> > kf_store(n_cams + 1);
  for (int i : alloc_order) kf_store[i].reset(new KeyFrame());
  std::vector<KeyFrame*> kf(n_cams);
  for (int c = 0; c < n_cams; c++) kf[c] = kf_store[c].get();
  KeyFrame* bad_kf = kf_store[n_cams].get();                                   // an observer the reference skips
  std::vector<int> perm(n_free);
  for (int i = 0; i < n_free; i++) perm[i] = i;
  std::shuffle(perm.begin(), perm.end(), rng);
  std::map<const KeyFrame*, int> kf_orig;
  for (int c = 0; c < n_cams; c++) {
    KeyFrame& K = *kf[c];
    K.mnId = c < n_free ? (unsigned long)(5 + 2 * perm[c]) : (c == n_free ? 0ul : (unsigned long)(1000 + c));
    K.fx = (float)camg[0]; K.fy = (float)camg[1]; K.cx = (float)camg[2]; K.cy = (float)camg[3]; K.mbf = (float)camg[4];
    K.mK = k_mat(K.fx, K.fy, K.cx, K.cy);
    K.mvInvLevelSigma2 = table;
    K.Tcw = pose_mat(&w.cam_qt[7 * (size_t)c]);
    kf_orig[&K] = c;
  }
  bad_kf->mnId = 777; bad_kf->mbBad = true; bad_kf->mvInvLevelSigma2 = table; bad_kf->Tcw = pose_mat(&w.cam_qt[0]);
  bad_kf->fx = (float)camg[0]; bad_kf->mK = k_mat((float)camg[0], (float)camg[1], (float)camg[2], (float)camg[3]);
  // pKF = the free camera with the largest mnId; every other free camera and the mnId-0 one are covisible (in shuffled order)
  KeyFrame* pKF = kf[0];
  for (int c = 1; c < n_free; c++) if (kf[c]->mnId > pKF->mnId) pKF = kf[c];
  for (int c = 0; c < n_cams; c++) if (kf[c] != pKF && (c < n_free || c == n_free)) pKF->mvpOrderedConnectedKeyFrames.push_back(kf[c]);
  std::shuffle(pKF->mvpOrderedConnectedKeyFrames.begin(), pKF->mvpOrderedConnectedKeyFrames.end(), rng);
  pKF->mvpOrderedConnectedKeyFrames.push_back(bad_kf);                         // a bad covisible keyframe: marked local, never a camera

  // ---- map points and their observations
  std::vector<std::unique_ptr<MapPoint> > mp_store(n_pts + 1);
  std::map<const MapPoint*, int> mp_orig;
  std::map<std::pair<const KeyFrame*, const MapPoint*>, int> ptobs_orig;
  for (int p = 0; p < n_pts; p++) {
    mp_store[p].reset(new MapPoint());
    MapPoint& M = *mp_store[p];
    M.mnId = (unsigned long)(3 * p + 1);
    M.mWorldPos = Mat(3, 1);
    for (int k = 0; k < 3; k++) M.mWorldPos.at<float>(k) = (float)w.pt_xyz[3 * (size_t)p + k];
    mp_orig[&M] = p;
    for (int o = w.pt_obs_start[p]; o < w.pt_obs_start[p + 1]; o++) {
      KeyFrame& K = *kf[w.pt_obs_cam[o]];
      KeyPoint kp; kp.pt.x = (float)w.pt_obs_uvr[3 * (size_t)o]; kp.pt.y = (float)w.pt_obs_uvr[3 * (size_t)o + 1]; kp.octave = octave_of(table, w.pt_obs_inv_sigma2[o]);
      if (K.mvKeysUn.size() % 5 == 2) { K.mvKeysUn.push_back(kp); K.mvuRight.push_back(-1.f); K.mvpMapPoints.push_back(nullptr); }   // a keypoint without a MapPoint in between
      M.mObservations[&K] = K.mvKeysUn.size();
      K.mvKeysUn.push_back(kp); K.mvuRight.push_back((float)w.pt_obs_uvr[3 * (size_t)o + 2]); K.mvpMapPoints.push_back(&M);
      ptobs_orig[std::make_pair((const KeyFrame*)&K, (const MapPoint*)&M)] = o;
    }
    if (p % 7 == 3) {                                                           // the bad keyframe also observes this point: skipped by the reference
      KeyPoint kp; kp.pt.x = 10.f; kp.pt.y = 10.f; kp.octave = 0;
      M.mObservations[bad_kf] = bad_kf->mvKeysUn.size();
      bad_kf->mvKeysUn.push_back(kp); bad_kf->mvuRight.push_back(5.f); bad_kf->mvpMapPoints.push_back(&M);
    }
  }
  mp_store[n_pts].reset(new MapPoint());                                         // a bad MapPoint among pKF's matches: never enters the window
  MapPoint& bad_mp = *mp_store[n_pts];
  bad_mp.mnId = 999999; bad_mp.mbBad = true; bad_mp.mWorldPos = Mat(3, 1);
  { KeyPoint kp; kp.pt.x = 1.f; kp.pt.y = 2.f; kp.octave = 0; bad_mp.mObservations[pKF] = pKF->mvKeysUn.size(); pKF->mvKeysUn.push_back(kp); pKF->mvuRight.push_back(1.f); pKF->mvpMapPoints.push_back(&bad_mp); }

  // ---- map lines
  std::vector<std::unique_ptr<MapLine> > ml_store(n_lns + 1);
  std::map<const MapLine*, int> ml_orig;
  std::map<std::pair<const KeyFrame*, const MapLine*>, int> lnobs_orig;
  for (int l = 0; l < n_lns; l++) {
    ml_store[l].reset(new MapLine());
    MapLine& L = *ml_store[l];
    L.mnId = (unsigned long)(2 * l + 7);
    L.mX0 = Vector3d(w.line_x0[3 * (size_t)l], w.line_x0[3 * (size_t)l + 1], w.line_x0[3 * (size_t)l + 2]);
    L.mDir = Vector3d(w.line_dir[3 * (size_t)l], w.line_dir[3 * (size_t)l + 1], w.line_dir[3 * (size_t)l + 2]);
    ml_orig[&L] = l;
    for (int o = w.ln_obs_start[l]; o < w.ln_obs_start[l + 1]; o++) {
      KeyFrame& K = *kf[w.ln_obs_cam[o]];
      const bool has_right = !(w.ln_obs_right[4 * (size_t)o] < 0);
      L.mObservations[&K] = K.mvLinesLeft.size();
      K.mvLinesLeft.push_back(key_line(&w.ln_obs_left[4 * (size_t)o], w.ln_obs_octave[2 * (size_t)o]));
      if (has_right) { K.line_matches.push_back((int)K.mvLinesRight.size()); K.mvLinesRight.push_back(key_line(&w.ln_obs_right[4 * (size_t)o], w.ln_obs_octave[2 * (size_t)o + 1])); }
      else K.line_matches.push_back(-1);
      K.mvpMapLines.push_back(&L);
      lnobs_orig[std::make_pair((const KeyFrame*)&K, (const MapLine*)&L)] = o;
    }
  }
  ml_store[n_lns].reset(new MapLine());                                          // a MapLine with two observations: below the `Observations() < 4` bar (:972)
  MapLine& short_ml = *ml_store[n_lns];
  short_ml.mnId = 888888; short_ml.mX0 = Vector3d(1, 2, 3); short_ml.mDir = Vector3d(1, 0, 0);
  for (int c = 0; c < std::min(2, n_free); c++) {
    KeyFrame& K = *kf[c];
    short_ml.mObservations[&K] = K.mvLinesLeft.size();
    const double seg[4] = {10, 10, 50, 50};
    K.mvLinesLeft.push_back(key_line(seg, 0)); K.line_matches.push_back(-1); K.mvpMapLines.push_back(&short_ml);
  }

  // ---- the call, as LocalMapping::Run makes it (LocalMapping.cc:76-82)
  Map map;
  bool mbAbortBA = h[6] != 0;
  lld_amd::Context ctx(0);
  lld_adapter::LbaTrace tr;
  lld_adapter::LocalBundleAdjustment(ctx.get(), pKF, &mbAbortBA, &map, camg[5], &tr);

  // ---- dump: gathered window (same layout as the input), raw output, object state
  const lld_amd::BAWindow& g = tr.window;
  const lld_amd::BAOutput& o = tr.output;
  Writer wr(out);
  const int32_t gh[8] = {g.n_cams(), g.n_free_cams, g.n_points(), (int32_t)g.pt_obs_cam.size(), g.n_lines(), (int32_t)g.ln_obs_cam.size(), tr.returned_before_optimising ? 1 : 0,
                         (int32_t)pKF->mnId};
  wr.put(gh, 8);
  const double gcam[6] = {g.cam.fx, g.cam.fy, g.cam.cx, g.cam.cy, g.cam.bf, camg[5]};
  wr.put(gcam, 6);
  wr.put(g.cam_qt); wr.put(g.pt_xyz); wr.put(g.pt_obs_start); wr.put(g.pt_obs_cam); wr.put(g.pt_obs_uvr); wr.put(g.pt_obs_inv_sigma2);
  wr.put(g.line_x0); wr.put(g.line_dir); wr.put(g.ln_obs_start); wr.put(g.ln_obs_cam); wr.put(g.ln_obs_left); wr.put(g.ln_obs_right); wr.put(g.ln_obs_octave);
  wr.put(o.cam_qt); wr.put(o.pt_xyz); wr.put(o.line_x0); wr.put(o.line_dir); wr.put(o.pt_obs_outlier); wr.put(o.ln_edge_outlier); wr.put(o.line_removed);
  const double chi2[2] = {o.stats.chi2_round1, o.stats.chi2_final};
  wr.put(chi2, 2);
  const int32_t st[4] = {o.stats.lm_iterations[0], o.stats.lm_iterations[1], o.stats.aborted, o.stats.n_lines_removed};
  wr.put(st, 4);
  // cameras
  for (KeyFrame* K : tr.cams) { const int32_t v[3] = {kf_orig.count(K) ? kf_orig[K] : -1, (int32_t)K->mnId, K->n_set_pose}; wr.put(v, 3); wr.put(K->Tcw.ptr<float>(), 16); }
  // points
  for (MapPoint* M : tr.points) { const int32_t v[3] = {mp_orig.count(M) ? mp_orig[M] : -1, M->n_set_pos, M->n_update_normal}; wr.put(v, 3); wr.put(M->mWorldPos.ptr<float>(), 3); }
  for (size_t k = 0; k < tr.pt_obs_owner.size(); k++) {
    KeyFrame* K = tr.pt_obs_owner[k].first; MapPoint* M = tr.pt_obs_owner[k].second;
    const auto it = ptobs_orig.find(std::make_pair((const KeyFrame*)K, (const MapPoint*)M));
    bool in_kf = false; for (MapPoint* q : K->mvpMapPoints) in_kf = in_kf || q == M;
    const int32_t v[3] = {it == ptobs_orig.end() ? -1 : it->second, M->mObservations.count(K) ? 1 : 0, in_kf ? 1 : 0};
    wr.put(v, 3);
  }
  // lines
  for (MapLine* L : tr.lines) { const int32_t v[2] = {ml_orig.count(L) ? ml_orig[L] : -1, L->n_set_pos}; wr.put(v, 2); wr.put(L->mX0.v, 3); wr.put(L->mDir.v, 3); }
  for (size_t k = 0; k < tr.ln_obs_owner.size(); k++) {
    KeyFrame* K = tr.ln_obs_owner[k].first; MapLine* L = tr.ln_obs_owner[k].second;
    const auto it = lnobs_orig.find(std::make_pair((const KeyFrame*)K, (const MapLine*)L));
    bool in_kf = false; for (MapLine* q : K->mvpMapLines) in_kf = in_kf || q == L;
    const int32_t v[3] = {it == lnobs_orig.end() ? -1 : it->second, L->mObservations.count(K) ? 1 : 0, in_kf ? 1 : 0};
    wr.put(v, 3);
  }
  // the objects the reference skips must be untouched; std::map<KeyFrame*> order must really differ from the mnId order for the test to mean something
  int addr_inversions = 0;
  for (size_t i = 1; i < tr.cams.size(); i++) if ((tr.cams[i] < tr.cams[i - 1]) != (tr.cams[i]->mnId < tr.cams[i - 1]->mnId)) addr_inversions++;
  const int32_t extra[8] = {bad_mp.n_set_pos + bad_mp.n_update_normal, short_ml.n_set_pos, bad_kf->n_set_pose, (int32_t)tr.vToErase.size(), (int32_t)tr.vToEraseLines.size(),
                            addr_inversions, (int32_t)short_ml.mObservations.size(), (int32_t)bad_mp.mObservations.size()};
  wr.put(extra, 8);
  std::printf("adapter-ba: pKF %lu, %d cameras (%d free), %d points, %d lines gathered; chi2 %.9g -> %.9g; %zu + %zu erased\n", pKF->mnId, g.n_cams(), g.n_free_cams, g.n_points(),
              g.n_lines(), chi2[0], chi2[1], tr.vToErase.size(), tr.vToEraseLines.size());
  return 0;
}

int run_pose(const char* in, const char* out, unsigned seed) {
  Reader r(in);
  int32_t h[2]; r.get(h, 2);                   // n_points n_lines
  double camg[6]; r.get(camg, 6);
  lld_amd::PoseFrame f;
  r.get(f.pose_qt, 7);
  r.get(f.pt_xw, 3 * (size_t)h[0]); r.get(f.pt_uvr, 3 * (size_t)h[0]); r.get(f.pt_inv_sigma2, h[0]);
  r.get(f.ln_x0, 3 * (size_t)h[1]); r.get(f.ln_dir, 3 * (size_t)h[1]); r.get(f.ln_left, 4 * (size_t)h[1]);
  r.get(f.ln_right, 4 * (size_t)h[1]); r.get(f.ln_octave, 2 * (size_t)h[1]);
  std::mt19937 rng(seed);
  const std::vector<float> table = level_table();
  Frame F;
  F.fx = (float)camg[0]; F.fy = (float)camg[1]; F.cx = (float)camg[2]; F.cy = (float)camg[3]; F.mbf = (float)camg[4];
  F.mK = k_mat(F.fx, F.fy, F.cx, F.cy);
  F.mvInvLevelSigma2 = table;
  F.mTcw = pose_mat(f.pose_qt);
  std::vector<std::unique_ptr<MapPoint> > mps; std::vector<std::unique_ptr<MapLine> > mls;
  for (int p = 0; p < h[0]; p++) {
    if (rng() % 3 == 0) { KeyPoint kp; kp.pt.x = 5.f; kp.pt.y = 6.f; kp.octave = 1; F.mvKeysUn.push_back(kp); F.mvuRight.push_back(-1.f); F.mvpMapPoints.push_back(nullptr); }   // keypoint without a MapPoint
    mps.emplace_back(new MapPoint());
    mps.back()->mWorldPos = Mat(3, 1);
    for (int k = 0; k < 3; k++) mps.back()->mWorldPos.at<float>(k) = (float)f.pt_xw[3 * (size_t)p + k];
    KeyPoint kp; kp.pt.x = (float)f.pt_uvr[3 * (size_t)p]; kp.pt.y = (float)f.pt_uvr[3 * (size_t)p + 1]; kp.octave = octave_of(table, f.pt_inv_sigma2[p]);
    F.mvKeysUn.push_back(kp); F.mvuRight.push_back((float)f.pt_uvr[3 * (size_t)p + 2]); F.mvpMapPoints.push_back(mps.back().get());
  }
  F.N = (int)F.mvKeysUn.size();
  F.mvbOutlier.assign(F.N, true);                                                  // stale flags: the call must reset the ones it uses and leave the rest
  for (int l = 0; l < h[1]; l++) {
    if (rng() % 2 == 0) { const double seg[4] = {1, 1, 9, 9}; F.mvLinesLeft.push_back(key_line(seg, 0)); F.line_matches.push_back(-1); F.mvpMapLines.push_back(nullptr); }      // a line without a MapLine
    mls.emplace_back(new MapLine());
    mls.back()->mX0 = Vector3d(f.ln_x0[3 * (size_t)l], f.ln_x0[3 * (size_t)l + 1], f.ln_x0[3 * (size_t)l + 2]);
    mls.back()->mDir = Vector3d(f.ln_dir[3 * (size_t)l], f.ln_dir[3 * (size_t)l + 1], f.ln_dir[3 * (size_t)l + 2]);
    const bool has_right = !(f.ln_right[4 * (size_t)l] < 0);
    F.mvLinesLeft.push_back(key_line(&f.ln_left[4 * (size_t)l], f.ln_octave[2 * (size_t)l]));
    if (has_right) { F.line_matches.push_back((int)F.mvLinesRight.size()); F.mvLinesRight.push_back(key_line(&f.ln_right[4 * (size_t)l], f.ln_octave[2 * (size_t)l + 1])); }
    else F.line_matches.push_back(-1);
    F.mvpMapLines.push_back(mls.back().get());
  }
  F.mvbOutlierLines.assign(F.mvLinesLeft.size(), true);
  lld_amd::Context ctx(0);
  lld_adapter::PoseTrace tr;
  const int32_t n_in = lld_adapter::PoseOptimization(ctx.get(), &F, camg[5], &tr);
  const lld_amd::PoseFrame& g = tr.frame;
  Writer wr(out);
  const int32_t gh[4] = {(int32_t)tr.vnIndexEdge.size(), (int32_t)tr.vnIndexLines.size(), F.N, (int32_t)F.mvLinesLeft.size()};
  wr.put(gh, 4);
  // gathered problem (the pose_qt field of the trace has been overwritten with the result: the INPUT pose is the frame's old mTcw, rebuilt by the test)
  wr.put(g.pt_xw); wr.put(g.pt_uvr); wr.put(g.pt_inv_sigma2); wr.put(g.ln_x0); wr.put(g.ln_dir); wr.put(g.ln_left); wr.put(g.ln_right); wr.put(g.ln_octave); wr.put(g.ln_frame_index);
  wr.put(g.pose_qt, 7); wr.put(&n_in, 1); wr.put(g.mvbOutlier); wr.put(g.mvbOutlierLines);
  std::vector<int32_t> ie(tr.vnIndexEdge.begin(), tr.vnIndexEdge.end()), il(tr.vnIndexLines.begin(), tr.vnIndexLines.end());
  wr.put(ie); wr.put(il);
  wr.put(F.mTcw.ptr<float>(), 16);
  const int32_t nset = F.n_set_pose; wr.put(&nset, 1);
  std::vector<uint8_t> fo(F.N), fl(F.mvbOutlierLines.size());
  for (int i = 0; i < F.N; i++) fo[i] = F.mvbOutlier[i] ? 1 : 0;
  for (size_t i = 0; i < fl.size(); i++) fl[i] = F.mvbOutlierLines[i] ? 1 : 0;
  wr.put(fo); wr.put(fl);
  std::printf("adapter-pose: %d of %d keypoints and %d of %zu lines carry a landmark; %d inliers\n", gh[0], F.N, gh[1], F.mvLinesLeft.size(), n_in);
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- matchers
// The keypoint side shared by the frames and the keyframe of the matcher scene (as Frame::Frame / KeyFrame::KeyFrame fill it)
struct KeypointData {
  int N, nl;
  float cam[6], bounds[6], logsf;
  std::vector<float> scale, sigma2, inv_sigma2, xy, angle, uright;
  std::vector<int32_t> octave;
  std::vector<uint32_t> desc;
  void read(Reader& r, int n, int levels) {
    N = n; nl = levels;
    r.get(cam, 6); r.get(bounds, 6);           // fx fy cx cy bf mb | mnMinX mnMaxX mnMinY mnMaxY mfGridElementWidthInv mfGridElementHeightInv
    r.get(scale, nl); r.get(sigma2, nl); r.get(inv_sigma2, nl); r.get(&logsf, 1);
    r.get(xy, 2 * (size_t)N); r.get(octave, N); r.get(angle, N); r.get(uright, N); r.get(desc, 8 * (size_t)N);
  }
  template <class F> void keys(F& f) const {
    f.fx = cam[0]; f.fy = cam[1]; f.cx = cam[2]; f.cy = cam[3]; f.mbf = cam[4];
    f.mvKeysUn.resize(N);
    for (int k = 0; k < N; k++) { f.mvKeysUn[k].pt.x = xy[2 * k]; f.mvKeysUn[k].pt.y = xy[2 * k + 1]; f.mvKeysUn[k].octave = octave[k]; f.mvKeysUn[k].angle = angle[k]; }
    f.mvuRight = uright;
    f.mDescriptors = MatU8(N, 32); std::memcpy(f.mDescriptors.template ptr<unsigned char>(), desc.data(), 32 * (size_t)N);
    f.mnMinX = bounds[0]; f.mnMaxX = bounds[1]; f.mnMinY = bounds[2]; f.mnMaxY = bounds[3];
    f.mfGridElementWidthInv = bounds[4]; f.mfGridElementHeightInv = bounds[5];
    f.mnScaleLevels = nl; f.mfScaleFactor = scale[1]; f.mfLogScaleFactor = logsf;
    f.mvScaleFactors = scale; f.mvLevelSigma2 = sigma2; f.mvInvLevelSigma2 = inv_sigma2;
  }
  void frame(Frame& F, const float* Tcw, unsigned long id) const {
    keys(F); F.N = N; F.mnId = id; F.mb = cam[5]; F.mvKeys = F.mvKeysUn;
    F.mvpMapPoints.assign(N, nullptr); F.mvbOutlier.assign(N, false);
    F.SetPose(Mat(4, 4, Tcw));
  }
};
struct PointData {
  int n;
  std::vector<float> pos, nrm, maxd, mind;
  std::vector<uint32_t> desc;
  std::vector<int32_t> nobs;
  std::vector<uint8_t> bad;
  void read(Reader& r, int count) {
    n = count;
    r.get(pos, 3 * (size_t)n); r.get(nrm, 3 * (size_t)n); r.get(maxd, n); r.get(mind, n); r.get(desc, 8 * (size_t)n); r.get(nobs, n); r.get(bad, n);
  }
  void make(std::vector<std::unique_ptr<MapPoint> >& own, std::vector<MapPoint*>& out, unsigned long id0) const {
    for (int i = 0; i < n; i++) {
      own.emplace_back(new MapPoint()); MapPoint* p = own.back().get();
      p->mnId = id0 + i; p->mWorldPos = Mat(3, 1, &pos[3 * i]); p->mNormalVector = Mat(3, 1, &nrm[3 * i]);
      p->mfMaxDistance = maxd[i]; p->mfMinDistance = mind[i];
      p->mDescriptor = MatU8(1, 32); std::memcpy(p->mDescriptor.ptr<unsigned char>(), &desc[8 * (size_t)i], 32);
      p->nObs = nobs[i]; p->mbBad = bad[i] != 0;
      out.push_back(p);
    }
  }
};

// The three per-frame / per-keyframe matchers through adapters/lld_matcher_adapter.cc on an object graph built from a scene file
int run_match(const char* in, const char* out) {
  Reader r(in);
  int32_t h[8]; r.get(h, 8);                   // N, levels, n_local, n_fuse, th_local, bMono, checkOrientation, 0
  KeypointData K; K.read(r, h[0], h[1]);
  float T[16], Tlast[16], th[2]; r.get(T, 16); r.get(Tlast, 16); r.get(th, 2);   // current pose, last pose, th (last frame), th (fuse)
  std::vector<uint8_t> occupied, last_valid, last_outlier, kf_has; std::vector<int32_t> kf_nobs;
  r.get(occupied, K.N);                        // keypoints of the current frame that already hold a MapPoint with observations
  PointData local; local.read(r, h[2]);
  PointData last; last.read(r, K.N); r.get(last_valid, K.N); r.get(last_outlier, K.N);
  std::vector<int32_t> last_octave; std::vector<float> last_angle;
  r.get(last_octave, K.N); r.get(last_angle, K.N);   // LastFrame.mvKeys[i].octave, LastFrame.mvKeysUn[i].angle
  PointData fuse; fuse.read(r, h[3]); r.get(kf_has, K.N); r.get(kf_nobs, K.N);
  lld_amd::Context ctx(0);
  std::vector<std::unique_ptr<MapPoint> > own;
  Writer wr(out);
  auto holder = [](const std::vector<MapPoint*>& slots, unsigned long id0, int n, std::vector<int32_t>& idx) {
    idx.assign(slots.size(), -1);
    for (size_t k = 0; k < slots.size(); k++)
      if (slots[k]) idx[k] = (slots[k]->mnId >= id0 && slots[k]->mnId < id0 + (unsigned long)n) ? (int32_t)(slots[k]->mnId - id0) : -2;
  };
  {  // ---- Tracking::SearchLocalPoints
    Frame F; K.frame(F, T, 7);
    std::vector<MapPoint*> occ, pts;
    for (int k = 0; k < K.N; k++)
      if (occupied[k]) { own.emplace_back(new MapPoint()); own.back()->mnId = 900000 + k; own.back()->nObs = 2; F.mvpMapPoints[k] = own.back().get(); }
    local.make(own, pts, 1000);
    lld_adapter::MatchTrace tr;
    const int nm = lld_adapter::SearchLocalPoints(ctx.get(), F, pts, h[4], &tr);
    std::vector<int32_t> idx; holder(F.mvpMapPoints, 1000, local.n, idx);
    std::vector<int32_t> lvl(local.n), vis(local.n); std::vector<float> uvr(3 * (size_t)local.n), vc(local.n); std::vector<uint8_t> inview(local.n);
    for (int i = 0; i < local.n; i++) {
      inview[i] = pts[i]->mbTrackInView; lvl[i] = pts[i]->mnTrackScaleLevel; vis[i] = pts[i]->mnVisible; vc[i] = pts[i]->mTrackViewCos;
      uvr[3 * i] = pts[i]->mTrackProjX; uvr[3 * i + 1] = pts[i]->mTrackProjY; uvr[3 * i + 2] = pts[i]->mTrackProjXR;
    }
    const int32_t c[2] = {nm, tr.nToMatch};
    wr.put(c, 2); wr.put(idx); wr.put(inview); wr.put(lvl); wr.put(vis); wr.put(uvr); wr.put(vc);
    int seen = 0; for (int k = 0; k < K.N; k++) if (occupied[k] && F.mvpMapPoints[k] && F.mvpMapPoints[k]->mnLastFrameSeen == 7) seen++;
    std::printf("SearchLocalPoints: %d matches of %d in view (%d local points, %d occupied keypoints marked seen)\n", nm, tr.nToMatch, local.n, seen);
  }
  {  // ---- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
    Frame Cur, Last; K.frame(Cur, T, 8); K.frame(Last, Tlast, 7);
    for (int k = 0; k < K.N; k++)
      if (occupied[k]) { own.emplace_back(new MapPoint()); own.back()->mnId = 910000 + k; own.back()->nObs = 2; Cur.mvpMapPoints[k] = own.back().get(); }
    std::vector<MapPoint*> pts; last.make(own, pts, 200000);
    for (int i = 0; i < K.N; i++) {
      Last.mvpMapPoints[i] = last_valid[i] ? pts[i] : nullptr; Last.mvbOutlier[i] = last_outlier[i] != 0;
      Last.mvKeys[i].octave = last_octave[i]; Last.mvKeysUn[i].angle = last_angle[i];
    }
    lld_adapter::MatchTrace tr;
    const int nm = lld_adapter::ORBmatcher(ctx.get(), 0.9f, h[6] != 0).SearchByProjection(Cur, Last, th[0], h[5] != 0, &tr);
    std::vector<int32_t> idx; holder(Cur.mvpMapPoints, 200000, K.N, idx);
    const int32_t c[2] = {nm, tr.direction};
    wr.put(c, 2); wr.put(idx); wr.put(tr.removed);
    std::printf("SearchByProjection(Current, Last): %d matches, direction %d\n", nm, tr.direction);
  }
  {  // ---- ORBmatcher::Fuse(pKF, vpMapPoints, th)
    KeyFrame KF; K.keys(KF); KF.mnId = 3; KF.Tcw = Mat(4, 4, T);
    { Frame tmp; tmp.SetPose(Mat(4, 4, T)); KF.Ow = tmp.mOw; }                // KeyFrame::SetPose computes Ow the same way (KeyFrame.cc:74-88)
    KF.mvpMapPoints.assign(K.N, nullptr);
    std::vector<MapPoint*> inkf(K.N, nullptr);
    for (int k = 0; k < K.N; k++)
      if (kf_has[k]) { own.emplace_back(new MapPoint()); MapPoint* p = own.back().get(); p->mnId = 920000 + k; p->nObs = kf_nobs[k]; p->mObservations[&KF] = k; KF.mvpMapPoints[k] = p; inkf[k] = p; }
    std::vector<MapPoint*> pts; fuse.make(own, pts, 300000);
    lld_adapter::MatchTrace tr;
    const int nf = lld_adapter::ORBmatcher(ctx.get()).Fuse(&KF, pts, th[1], &tr);
    std::vector<int32_t> idx; holder(KF.mvpMapPoints, 300000, fuse.n, idx);
    std::vector<uint8_t> pbad(fuse.n), kbad(K.N, 0); std::vector<int32_t> pobs(fuse.n);
    for (int i = 0; i < fuse.n; i++) { pbad[i] = pts[i]->isBad(); pobs[i] = pts[i]->Observations(); }
    for (int k = 0; k < K.N; k++) if (inkf[k]) kbad[k] = inkf[k]->isBad();
    const int32_t c[1] = {nf};
    wr.put(c, 1); wr.put(tr.match); wr.put(idx); wr.put(pbad); wr.put(pobs); wr.put(kbad);
    std::printf("Fuse: %d fused of %d points\n", nf, fuse.n);
  }
  return 0;
}

// The relocalisation / loop-closing matchers through adapters/lld_matcher_adapter.cc: SearchByProjection(Frame&, KeyFrame*, ...),
// SearchByProjection(KeyFrame*, Scw, ...), Fuse(KeyFrame*, Scw, ...) on one keyframe whose keypoint i holds MapPoint i, and SearchBySim3 on a pair.
int run_loopmatch(const char* in, const char* out) {
  Reader r(in);
  int32_t h[8]; r.get(h, 8);                   // N1, levels, N2, ORBdist, checkOrientation, 0, 0, 0
  KeypointData K1, K2; K1.read(r, h[0], h[1]);
  float T[16], Scw[16], th[4]; r.get(T, 16); r.get(Scw, 16); r.get(th, 4);   // th: relocalisation, KF / Scw, Fuse / Scw, SearchBySim3
  PointData pts; pts.read(r, K1.N);
  std::vector<float> kf_angle; std::vector<uint8_t> cur_occupied, kf_matched, kf_has, found;
  r.get(kf_angle, K1.N); r.get(cur_occupied, K1.N); r.get(kf_matched, K1.N); r.get(kf_has, K1.N); r.get(found, K1.N);
  K2.read(r, h[2], h[1]);
  float T1[16], T2[16], sim[13]; r.get(T1, 16); r.get(T2, 16); r.get(sim, 13);   // s12, R12 (9), t12 (3)
  PointData p1, p2; p1.read(r, K1.N); p2.read(r, K2.N);
  std::vector<uint8_t> has1, has2; std::vector<int32_t> pre12; r.get(has1, K1.N); r.get(has2, K2.N); r.get(pre12, K1.N);
  lld_amd::Context ctx(0);
  std::vector<std::unique_ptr<MapPoint> > own;
  Writer wr(out);
  auto index_of = [](const std::vector<MapPoint*>& slots, unsigned long id0, int n, std::vector<int32_t>& idx) {
    idx.assign(slots.size(), -1);
    for (size_t k = 0; k < slots.size(); k++)
      if (slots[k]) idx[k] = (slots[k]->mnId >= id0 && slots[k]->mnId < id0 + (unsigned long)n) ? (int32_t)(slots[k]->mnId - id0) : -2;
  };
  auto keyframe = [](KeyFrame& KF, const KeypointData& K, const float* Tcw, unsigned long id) {
    K.keys(KF); KF.mnId = id; KF.Tcw = Mat(4, 4, Tcw);
    Frame tmp; tmp.SetPose(Mat(4, 4, Tcw)); KF.Ow = tmp.mOw;
    KF.mvpMapPoints.assign(K.N, nullptr);
  };
  {  // ---- relocalisation: the keyframe's MapPoints into the current frame
    Frame Cur; K1.frame(Cur, T, 20);
    KeyFrame KF; keyframe(KF, K1, T, 5);
    for (int k = 0; k < K1.N; k++) KF.mvKeysUn[k].angle = kf_angle[k];
    std::vector<MapPoint*> mps; pts.make(own, mps, 400000);
    std::set<MapPoint*> sFound;
    for (int k = 0; k < K1.N; k++) { KF.mvpMapPoints[k] = mps[k]; if (found[k]) sFound.insert(mps[k]); }
    for (int k = 0; k < K1.N; k++) if (cur_occupied[k]) { own.emplace_back(new MapPoint()); own.back()->mnId = 930000 + k; Cur.mvpMapPoints[k] = own.back().get(); }
    lld_adapter::MatchTrace tr;
    const int n = lld_adapter::ORBmatcher(ctx.get(), 0.9f, h[4] != 0).SearchByProjection(Cur, &KF, sFound, th[0], h[3], &tr);
    std::vector<int32_t> idx; index_of(Cur.mvpMapPoints, 400000, K1.N, idx);
    const int32_t c = n; wr.put(&c, 1); wr.put(idx); wr.put(tr.removed);
    std::printf("SearchByProjection(Frame, KeyFrame): %d matches\n", n);
  }
  {  // ---- SearchByProjection(KeyFrame, Scw)
    KeyFrame KF; keyframe(KF, K1, T, 6);
    std::vector<MapPoint*> mps; pts.make(own, mps, 500000);
    std::vector<MapPoint*> vpMatched(K1.N, nullptr);
    for (int k = 0; k < K1.N; k++) if (kf_matched[k]) { own.emplace_back(new MapPoint()); own.back()->mnId = 940000 + k; vpMatched[k] = own.back().get(); }
    const int n = lld_adapter::ORBmatcher(ctx.get(), 0.75f).SearchByProjection(&KF, Mat(4, 4, Scw), mps, vpMatched, (int)th[1]);
    std::vector<int32_t> idx; index_of(vpMatched, 500000, K1.N, idx);
    const int32_t c = n; wr.put(&c, 1); wr.put(idx);
    std::printf("SearchByProjection(KeyFrame, Scw): %d matches\n", n);
  }
  {  // ---- Fuse(KeyFrame, Scw)
    KeyFrame KF; keyframe(KF, K1, T, 7);
    std::vector<MapPoint*> mps; pts.make(own, mps, 600000);
    for (int k = 0; k < K1.N; k++) if (kf_has[k]) { own.emplace_back(new MapPoint()); own.back()->mnId = 950000 + k; KF.mvpMapPoints[k] = own.back().get(); }
    std::vector<MapPoint*> vpReplace(K1.N, nullptr);
    lld_adapter::MatchTrace tr;
    const int n = lld_adapter::ORBmatcher(ctx.get()).Fuse(&KF, Mat(4, 4, Scw), mps, th[2], vpReplace, &tr);
    std::vector<int32_t> slots, rep; index_of(KF.mvpMapPoints, 600000, K1.N, slots); index_of(vpReplace, 600000, K1.N, rep);
    std::vector<int32_t> pobs(K1.N); for (int i = 0; i < K1.N; i++) pobs[i] = mps[i]->Observations();
    const int32_t c = n; wr.put(&c, 1); wr.put(tr.match); wr.put(slots); wr.put(rep); wr.put(pobs);
    std::printf("Fuse(KeyFrame, Scw): %d fused\n", n);
  }
  {  // ---- SearchBySim3
    KeyFrame KFa, KFb; keyframe(KFa, K1, T1, 8); keyframe(KFb, K2, T2, 9);
    std::vector<MapPoint*> m1, m2; p1.make(own, m1, 700000); p2.make(own, m2, 800000);
    for (int k = 0; k < K1.N; k++) KFa.mvpMapPoints[k] = has1[k] ? m1[k] : nullptr;
    for (int k = 0; k < K2.N; k++) { KFb.mvpMapPoints[k] = has2[k] ? m2[k] : nullptr; m2[k]->mObservations[&KFb] = k; }
    std::vector<MapPoint*> vpMatches12(K1.N, nullptr);
    for (int k = 0; k < K1.N; k++) if (pre12[k] >= 0) vpMatches12[k] = m2[pre12[k]];
    const int n = lld_adapter::ORBmatcher(ctx.get()).SearchBySim3(&KFa, &KFb, vpMatches12, sim[0], Mat(3, 3, sim + 1), Mat(3, 1, sim + 10), th[3]);
    std::vector<int32_t> idx; index_of(vpMatches12, 800000, K2.N, idx);
    const int32_t c = n; wr.put(&c, 1); wr.put(idx);
    std::printf("SearchBySim3: %d found\n", n);
  }
  return 0;
}

// The vocabulary-guided matchers through adapters/lld_matcher_adapter.cc: SearchByBoW(KeyFrame*, Frame&, ...) and SearchByBoW(KeyFrame*, KeyFrame*, ...)
// with FeatureVectors that also hold nodes only one side has (the lower_bound branches of the merge loop).
int run_bow(const char* in, const char* out) {
  Reader r(in);
  int32_t h[4]; r.get(h, 4);                   // N1, N2, levels, checkOrientation
  float nn[2]; r.get(nn, 2);                   // mfNNratio of the two calls
  KeypointData K1, K2; K1.read(r, h[0], h[2]); K2.read(r, h[1], h[2]);
  auto read_featvec = [&](DBoW2::FeatureVector& fv) {
    int32_t n; r.get(&n, 1);
    std::vector<int32_t> ids, start, idx; r.get(ids, n); r.get(start, (size_t)n + 1); r.get(idx, (size_t)start[n]);
    for (int k = 0; k < n; k++) { std::vector<unsigned int>& v = fv[(unsigned)ids[k]]; for (int j = start[k]; j < start[k + 1]; j++) v.push_back((unsigned)idx[j]); }
  };
  DBoW2::FeatureVector fv1, fv2; read_featvec(fv1); read_featvec(fv2);
  std::vector<uint8_t> has1, bad1, has2, bad2; r.get(has1, K1.N); r.get(bad1, K1.N); r.get(has2, K2.N); r.get(bad2, K2.N);
  float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  lld_amd::Context ctx(0);
  std::vector<std::unique_ptr<MapPoint> > own;
  auto points = [&](KeyFrame& KF, const std::vector<uint8_t>& has, const std::vector<uint8_t>& bad, unsigned long id0) {
    KF.mvpMapPoints.assign(has.size(), nullptr);
    for (size_t k = 0; k < has.size(); k++) if (has[k]) { own.emplace_back(new MapPoint()); own.back()->mnId = id0 + k; own.back()->mbBad = bad[k] != 0; KF.mvpMapPoints[k] = own.back().get(); }
  };
  Writer wr(out);
  {
    KeyFrame KF; K1.keys(KF); KF.mFeatVec = fv1; points(KF, has1, bad1, 100000);
    Frame F; K2.frame(F, T, 1); F.mFeatVec = fv2;
    std::vector<MapPoint*> vpMapPointMatches;
    const int n = lld_adapter::ORBmatcher(ctx.get(), nn[0], h[3] != 0).SearchByBoW(&KF, F, vpMapPointMatches);
    std::vector<int32_t> idx(K2.N, -1);
    for (int k = 0; k < K2.N; k++) if (vpMapPointMatches[k]) idx[k] = (int32_t)(vpMapPointMatches[k]->mnId - 100000);
    const int32_t c = n; wr.put(&c, 1); wr.put(idx);
    std::printf("SearchByBoW(KeyFrame, Frame): %d matches\n", n);
  }
  {
    KeyFrame KFa, KFb; K1.keys(KFa); K2.keys(KFb); KFa.mFeatVec = fv1; KFb.mFeatVec = fv2;
    points(KFa, has1, bad1, 100000); points(KFb, has2, bad2, 200000);
    std::vector<MapPoint*> vpMatches12;
    const int n = lld_adapter::ORBmatcher(ctx.get(), nn[1], h[3] != 0).SearchByBoW(&KFa, &KFb, vpMatches12);
    std::vector<int32_t> idx(K1.N, -1);
    for (int k = 0; k < K1.N; k++) if (vpMatches12[k]) idx[k] = (int32_t)(vpMatches12[k]->mnId - 200000);
    const int32_t c = n; wr.put(&c, 1); wr.put(idx);
    std::printf("SearchByBoW(KeyFrame, KeyFrame): %d matches\n", n);
  }
  {  // ---- SearchForTriangulation: the keypoints WITHOUT a MapPoint, epipolar gate from F12, epipole from the two poses
    float T1[16], T2[16], F12[9]; int32_t only; r.get(T1, 16); r.get(T2, 16); r.get(F12, 9); r.get(&only, 1);
    KeyFrame KFa, KFb; K1.keys(KFa); K2.keys(KFb); KFa.mFeatVec = fv1; KFb.mFeatVec = fv2;
    KFa.Tcw = Mat(4, 4, T1); KFb.Tcw = Mat(4, 4, T2);
    { Frame tmp; tmp.SetPose(Mat(4, 4, T1)); KFa.Ow = tmp.mOw; }
    std::vector<uint8_t> nobad1(K1.N, 0), nobad2(K2.N, 0);
    points(KFa, has1, nobad1, 100000); points(KFb, has2, nobad2, 200000);
    std::vector<std::pair<size_t, size_t> > pairs;
    const int n = lld_adapter::ORBmatcher(ctx.get(), 0.6f, h[3] != 0).SearchForTriangulation(&KFa, &KFb, Mat(3, 3, F12), pairs, only != 0);
    std::vector<int32_t> m12(K1.N, -1);
    for (size_t i = 0; i < pairs.size(); i++) m12[pairs[i].first] = (int32_t)pairs[i].second;
    const int32_t c[2] = {n, (int32_t)pairs.size()}; wr.put(c, 2); wr.put(m12);
    std::printf("SearchForTriangulation: %d matches\n", n);
  }
  return 0;
}

// The line matchers through adapters/lld_line_adapter.cc: Tracking::AddLinesFrom on MapLine / Frame objects built from the first scene of
// the `harness lines` input (same file format), then TwoFrameLineMatcher::MatchLines on the KeyLines / descriptors of a second block.
int run_lines(const char* in, const char* out) {
  Reader r(in);
  int32_t h[3]; r.get(h, 3);                   // dim, n_map, use_grid
  double k[9 + 16 + 16 + 5]; r.get(k, 46);     // K, T_curr, T_last (unused), b, mnMaxX, mnMaxY, mdThr, thrReprojLineBase
  const int dim = h[0], n_map = h[1];
  std::vector<double> X0, dir, X1, X2; std::vector<uint8_t> skip; std::vector<float> ldesc;
  r.get(X0, 3 * (size_t)n_map); r.get(dir, 3 * (size_t)n_map); r.get(X1, 3 * (size_t)n_map); r.get(X2, 3 * (size_t)n_map);
  r.get(skip, n_map); r.get(ldesc, (size_t)n_map * dim);
  int32_t nf[2]; r.get(nf, 2);
  std::vector<float> left, right, fdesc; std::vector<int32_t> octave, lm; std::vector<uint8_t> occ;
  r.get(left, 4 * (size_t)nf[0]); r.get(right, 4 * (size_t)nf[1]); r.get(octave, nf[0]); r.get(lm, nf[0]); r.get(occ, nf[0]); r.get(fdesc, (size_t)nf[0] * dim);
  lld_amd::Context ctx(0);
  Frame F; F.mnId = 42;
  auto kl = [](const float* s4, int oct) { KeyLine q; q.startPointX = s4[0]; q.startPointY = s4[1]; q.endPointX = s4[2]; q.endPointY = s4[3]; q.octave = oct; return q; };
  for (int i = 0; i < nf[0]; i++) F.mvLinesLeft.push_back(kl(&left[4 * i], octave[i]));
  for (int i = 0; i < nf[1]; i++) F.mvLinesRight.push_back(kl(&right[4 * i], 0));
  F.line_matches.assign(lm.begin(), lm.end());
  F.mDescriptorsLines = Mat(nf[0], dim, fdesc.data());
  std::vector<std::unique_ptr<MapLine> > own;
  F.mvpMapLines.assign(nf[0], nullptr);
  for (int i = 0; i < nf[0]; i++) if (occ[i]) { own.emplace_back(new MapLine()); own.back()->mnId = 990000 + i; F.mvpMapLines[i] = own.back().get(); }
  // the skipped map lines take the three forms the reference tests for (:1015-1026) in turn
  std::vector<MapLine*> lines_last(n_map, nullptr); std::vector<Mat> descs;
  int kind = 0;
  for (int i = 0; i < n_map; i++) {
    descs.push_back(Mat(1, dim, &ldesc[(size_t)i * dim]));
    if (skip[i] && (kind++ % 3) == 0) continue;                              // NULL
    own.emplace_back(new MapLine()); MapLine* p = own.back().get(); p->mnId = i;
    p->mX0 = Vector3d(X0[3 * i], X0[3 * i + 1], X0[3 * i + 2]); p->mDir = Vector3d(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    p->mX1 = Vector3d(X1[3 * i], X1[3 * i + 1], X1[3 * i + 2]); p->mX2 = Vector3d(X2[3 * i], X2[3 * i + 1], X2[3 * i + 2]);
    if (skip[i]) { if ((kind % 3) == 1) p->tracked_last_id = 42; else p->mbBad = true; }
    lines_last[i] = p;
  }
  lld_adapter::TrackingLines T;
  for (int i = 0; i < 9; i++) T.K[i] = k[i];
  T.mb = k[41]; T.mnMaxX = k[42]; T.mnMaxY = k[43]; T.mdThr = k[44]; T.monocular = false;
  lld_adapter::AddLinesFrom(ctx, T, lines_last, k + 9, descs, Mat(), k[45], &F);
  std::vector<int32_t> m(n_map, -1), tracked(n_map, 0);
  for (int i = 0; i < nf[0]; i++) if (F.mvpMapLines[i] && F.mvpMapLines[i]->mnId < 990000) m[F.mvpMapLines[i]->mnId] = i;
  for (int i = 0; i < n_map; i++) tracked[i] = lines_last[i] && !skip[i] && lines_last[i]->tracked_last_id == 42;
  Writer wr(out);
  wr.put(m); wr.put(tracked);
  int na = 0; for (int i = 0; i < n_map; i++) na += m[i] >= 0;
  std::printf("AddLinesFrom: %d of %d map lines placed in the frame\n", na, n_map);
  // TwoFrameLineMatcher::MatchLines on a stereo pair of line sets
  double K[9], bt[3]; int32_t n2[3]; r.get(K, 9); r.get(bt, 3); r.get(n2, 3);   // K; b, tau, minLineLength; nl, nr, dim
  std::vector<float> sl, sr, dl, dr; std::vector<int32_t> ol, orr;
  r.get(sl, 4 * (size_t)n2[0]); r.get(ol, n2[0]); r.get(dl, (size_t)n2[0] * n2[2]); r.get(sr, 4 * (size_t)n2[1]); r.get(orr, n2[1]); r.get(dr, (size_t)n2[1] * n2[2]);
  std::vector<KeyLine> kl1, kl2;
  for (int i = 0; i < n2[0]; i++) kl1.push_back(kl(&sl[4 * i], ol[i]));
  for (int i = 0; i < n2[1]; i++) kl2.push_back(kl(&sr[4 * i], orr[i]));
  std::vector<int> dm;
  lld_adapter::TwoFrameLineMatcher(ctx, K, bt[0], bt[1], (int)bt[2]).MatchLines(kl1, kl2, Mat(n2[0], n2[2], dl.data()), Mat(n2[1], n2[2], dr.data()), &dm);
  std::vector<int32_t> dm32(dm.begin(), dm.end());
  wr.put(dm32);
  int ns = 0; for (size_t i = 0; i < dm.size(); i++) ns += dm[i] >= 0;
  std::printf("TwoFrameLineMatcher::MatchLines: %d of %d left lines matched\n", ns, n2[0]);
  return 0;
}

// ORBmatcher::SearchForInitialization through adapters/lld_matcher_adapter.cc on two Frames built from a scene file
int run_init(const char* in, const char* out) {
  Reader r(in);
  int32_t h[6]; r.get(h, 6);                   // N1, levels, N2, windowSize, checkOrientation, 0
  float nn; r.get(&nn, 1);
  KeypointData K1, K2; K1.read(r, h[0], h[1]); K2.read(r, h[2], h[1]);
  std::vector<float> prev; r.get(prev, 2 * (size_t)h[0]);
  float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  Frame F1, F2; K1.frame(F1, I4, 1); K2.frame(F2, I4, 2);
  std::vector<lld_slam::Point2f> vbPrevMatched(h[0]);
  for (int i = 0; i < h[0]; i++) { vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }
  std::vector<int> vnMatches12(3, 77);         // (the routine resizes and resets it, :408)
  lld_amd::Context ctx(0);
  const int nm = lld_adapter::ORBmatcher(ctx.get(), nn, h[4] != 0).SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, h[3]);
  Writer wr(out);
  const int32_t c[2] = {nm, (int32_t)vnMatches12.size()};
  std::vector<int32_t> m(vnMatches12.begin(), vnMatches12.end());
  for (int i = 0; i < h[0]; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
  wr.put(c, 2); wr.put(m); wr.put(prev);
  std::printf("SearchForInitialization: %d matches of %d keypoints\n", nm, h[0]);
  return 0;
}

// Tracking::MatchLinesLastKF through adapters/lld_line_adapter.cc: two stereo Frames with lines, a KeyFrame and a Map
int run_lastkf(const char* in, const char* out) {
  Reader r(in);
  int32_t h[2]; r.get(h, 2);                   // dim, 0
  double k[9 + 16 + 16 + 4]; r.get(k, 45);     // K, T_curr, T_last, b, mnMaxX, mnMaxY, mdThr
  const int dim = h[0];
  auto kl = [](const float* s4, int oct) { KeyLine q; q.startPointX = s4[0]; q.startPointY = s4[1]; q.endPointX = s4[2]; q.endPointY = s4[3]; q.octave = oct; return q; };
  struct Lines { std::vector<float> left, right, desc; std::vector<int32_t> octave, lm; std::vector<uint8_t> flag; int32_t n[2]; };
  auto read_lines = [&](Lines& L) {
    r.get(L.n, 2);
    r.get(L.left, 4 * (size_t)L.n[0]); r.get(L.right, 4 * (size_t)L.n[1]); r.get(L.octave, L.n[0]); r.get(L.lm, L.n[0]); r.get(L.flag, L.n[0]); r.get(L.desc, (size_t)L.n[0] * dim);
  };
  Lines C, L; read_lines(C); read_lines(L);
  auto fill = [&](Frame& F, const Lines& S, unsigned long id) {
    F.mnId = id;
    for (int i = 0; i < S.n[0]; i++) F.mvLinesLeft.push_back(kl(&S.left[4 * i], S.octave[i]));
    for (int i = 0; i < S.n[1]; i++) F.mvLinesRight.push_back(kl(&S.right[4 * i], 0));
    F.line_matches.assign(S.lm.begin(), S.lm.end());
    F.mDescriptorsLines = Mat(S.n[0], dim, S.desc.data());
    F.mvpMapLines.assign(S.n[0], nullptr);
  };
  Frame Cur, Last; fill(Cur, C, 42); fill(Last, L, 41);
  std::vector<std::unique_ptr<MapLine> > own;
  for (int i = 0; i < C.n[0]; i++) if (C.flag[i]) { own.emplace_back(new MapLine()); own.back()->mnId = 990000 + i; Cur.mvpMapLines[i] = own.back().get(); }
  // lines of the last frame: flagged ones hold a MapLine this frame already tracks (:1517-1520); of the others every third holds one it does not, the rest none
  for (int i = 0; i < L.n[0]; i++) {
    if (!L.flag[i] && i % 3) continue;
    own.emplace_back(new MapLine()); own.back()->mnId = 980000 + i; own.back()->tracked_last_id = L.flag[i] ? 42 : 41; Last.mvpMapLines[i] = own.back().get();
  }
  lld_adapter::TrackingLines T;
  for (int i = 0; i < 9; i++) T.K[i] = k[i];
  T.mb = k[41]; T.mnMaxX = k[42]; T.mnMaxY = k[43]; T.mdThr = k[44]; T.monocular = false;
  lld_amd::Context ctx(0);
  KeyFrame KF; KF.mnId = 5; lld_slam::Map map;
  std::vector<MapLine*> created; std::vector<int> match;
  const int ret = lld_adapter::MatchLinesLastKF(ctx, T, Cur, Last, k + 9, k + 25, &KF, &map, &created, &match);
  const int nc = C.n[0];
  std::vector<uint8_t> made(nc, 0), wired(nc, 0); std::vector<double> x0(3 * (size_t)nc, 0.0), dr(3 * (size_t)nc, 0.0);
  for (int i = 0; i < nc; i++) {
    MapLine* ml = created[i];
    if (!ml) { wired[i] = (Cur.mvpMapLines[i] != nullptr) == (C.flag[i] != 0); continue; }        // untouched slots keep what they held
    made[i] = 1;
    for (int c = 0; c < 3; c++) { x0[3 * i + c] = ml->mX0(c); dr[3 * i + c] = ml->mDir(c); }
    const std::map<KeyFrame*, size_t> obs = ml->GetObservations();
    wired[i] = Cur.mvpMapLines[i] == ml && ml->tracked_last_id == 42 && KF.mvpMapLines.size() > (size_t)i && KF.mvpMapLines[i] == ml && obs.size() == 1 && obs.count(&KF) &&
               obs.find(&KF)->second == (size_t)i && ml->n_distinctive == 1 && map.mspMapLines.count(ml) == 1 && ml->mpRefKF == &KF && ml->mpMap == &map && ml->ref_idx == i;
  }
  Writer wr(out);
  const int32_t c[2] = {ret, (int32_t)map.mspMapLines.size()};
  std::vector<int32_t> m32(match.begin(), match.end());
  wr.put(c, 2); wr.put(m32); wr.put(made); wr.put(wired); wr.put(x0); wr.put(dr);
  for (MapLine* ml : created) delete ml;
  std::printf("MatchLinesLastKF: returned %d, %d new map lines\n", ret, c[1]);
  return 0;
}

// Tracking::TrackWithMotionModel + Tracking::TrackLocalMap through adapters/lld_tracking_adapter.cc on a live object graph: the scene file of
// `harness track` (lld_slam_amd/tracking.py write_harness_scene) plus a tail {Tlast[16] f32, mb f32, mbOnlyTracking i32, half_outliers i32,
// via_set_state i32, 0}.  via_set_state: TrackLocalMap runs on a SECOND device frame that was handed the object graph's state (FrameOnDevice::SetFrameState),
// the way it follows TrackReferenceKeyFrame in the running system.
// One MapPoint / MapLine object per id, shared by the last frame and the local map as in the running system.
int run_track(const char* in, const char* out) {
  Reader r(in);
  int32_t h[16]; r.get(h, 16);      // nt n_levels n_last n_mp nl nr dim n_last_lines n_local_lines repeats download_between
  const int nt = h[0], n_levels = h[1], n_last = h[2], n_mp = h[3], nl = h[4], nr = h[5], dim = h[6], n_ll = h[7], n_ml = h[8];
  float fc[6]; r.get(fc, 6);        // min_x min_y max_x max_y grid_width_inv grid_height_inv
  std::vector<float> scale, inv_sigma2; r.get(scale, n_levels); r.get(inv_sigma2, n_levels);
  double dc[8]; r.get(dc, 8);       // fx fy cx cy bf gamma line_thr_base md_thr
  std::vector<uint32_t> t_desc; std::vector<float> t_xy, t_ur, t_ang; std::vector<int32_t> t_oct;
  r.get(t_desc, 8 * (size_t)nt); r.get(t_xy, 2 * (size_t)nt); r.get(t_oct, nt); r.get(t_ur, nt); r.get(t_ang, nt);
  lld_frame_view view; r.get(&view, 1);
  float Tcw[16]; r.get(Tcw, 16);
  std::vector<float> l_pos, l_ang; std::vector<uint8_t> l_valid, l_obs; std::vector<int32_t> l_oct, l_id; std::vector<uint32_t> l_desc;
  r.get(l_pos, 3 * (size_t)n_last); r.get(l_valid, n_last); r.get(l_oct, n_last); r.get(l_ang, n_last); r.get(l_desc, 8 * (size_t)n_last); r.get(l_obs, n_last); r.get(l_id, n_last);
  std::vector<float> m_pos, m_nrm, m_maxd, m_mind; std::vector<uint32_t> m_desc; std::vector<uint8_t> m_obs, m_skip; std::vector<int32_t> m_id;
  r.get(m_pos, 3 * (size_t)n_mp); r.get(m_nrm, 3 * (size_t)n_mp); r.get(m_maxd, n_mp); r.get(m_mind, n_mp); r.get(m_desc, 8 * (size_t)n_mp); r.get(m_obs, n_mp); r.get(m_skip, n_mp);
  r.get(m_id, n_mp);
  std::vector<float> ln_left, ln_right, ln_desc; std::vector<int32_t> ln_lo, ln_ro, ln_lm;
  r.get(ln_left, 4 * (size_t)nl); r.get(ln_lo, nl); r.get(ln_right, 4 * (size_t)nr); r.get(ln_ro, nr); r.get(ln_lm, nl); r.get(ln_desc, (size_t)nl * dim);
  struct LineSet { std::vector<double> x0, dir, x1, x2; std::vector<uint8_t> skip; std::vector<float> desc; std::vector<int32_t> id; };
  auto read_lines = [&](LineSet& L, int n) {
    r.get(L.x0, 3 * (size_t)n); r.get(L.dir, 3 * (size_t)n); r.get(L.x1, 3 * (size_t)n); r.get(L.x2, 3 * (size_t)n); r.get(L.skip, n); r.get(L.desc, (size_t)n * dim); r.get(L.id, n);
  };
  LineSet last_lines, local_lines; read_lines(last_lines, n_ll); read_lines(local_lines, n_ml);
  float Tlast[16], mb; int32_t tail[4]; r.get(Tlast, 16); r.get(&mb, 1); r.get(tail, 4);     // ..., mbOnlyTracking, half_outliers, via_set_state, 0

  // ---- the current frame as Frame::Frame leaves it
  Frame Cur;
  Cur.N = nt; Cur.mnId = 42; Cur.fx = (float)dc[0]; Cur.fy = (float)dc[1]; Cur.cx = (float)dc[2]; Cur.cy = (float)dc[3]; Cur.mbf = (float)dc[4]; Cur.mb = mb;
  Cur.mvKeysUn.resize(nt);
  for (int k = 0; k < nt; k++) { Cur.mvKeysUn[k].pt.x = t_xy[2 * k]; Cur.mvKeysUn[k].pt.y = t_xy[2 * k + 1]; Cur.mvKeysUn[k].octave = t_oct[k]; Cur.mvKeysUn[k].angle = t_ang[k]; }
  Cur.mvKeys = Cur.mvKeysUn; Cur.mvuRight = t_ur;
  Cur.mDescriptors = MatU8(nt, 32); std::memcpy(Cur.mDescriptors.ptr<unsigned char>(), t_desc.data(), 32 * (size_t)nt);
  Cur.mnMinX = fc[0]; Cur.mnMinY = fc[1]; Cur.mnMaxX = fc[2]; Cur.mnMaxY = fc[3]; Cur.mfGridElementWidthInv = fc[4]; Cur.mfGridElementHeightInv = fc[5];
  Cur.mnScaleLevels = n_levels; Cur.mfScaleFactor = scale[1]; Cur.mfLogScaleFactor = view.log_scale_factor; Cur.mvScaleFactors = scale; Cur.mvInvLevelSigma2 = inv_sigma2;
  Cur.mvLevelSigma2.resize(n_levels); for (int l = 0; l < n_levels; l++) Cur.mvLevelSigma2[l] = scale[l] * scale[l];
  Cur.mvpMapPoints.assign(nt, nullptr); Cur.mvbOutlier.assign(nt, false);
  for (int i = 0; i < nl; i++) { KeyLine q; q.startPointX = ln_left[4 * i]; q.startPointY = ln_left[4 * i + 1]; q.endPointX = ln_left[4 * i + 2]; q.endPointY = ln_left[4 * i + 3]; q.octave = ln_lo[i]; Cur.mvLinesLeft.push_back(q); }
  for (int i = 0; i < nr; i++) { KeyLine q; q.startPointX = ln_right[4 * i]; q.startPointY = ln_right[4 * i + 1]; q.endPointX = ln_right[4 * i + 2]; q.endPointY = ln_right[4 * i + 3]; q.octave = ln_ro[i]; Cur.mvLinesRight.push_back(q); }
  Cur.line_matches.assign(ln_lm.begin(), ln_lm.end());
  if (nl > 0) Cur.mDescriptorsLines = Mat(nl, dim, ln_desc.data());
  Cur.mvpMapLines.assign(nl, nullptr); Cur.mvbOutlierLines.assign(nl, false);
  Cur.SetPose(Mat(4, 4, Tcw));                 // mCurrentFrame.SetPose(mVelocity*mLastFrame.mTcw)

  // ---- the map: one object per id
  std::map<int32_t, std::unique_ptr<MapPoint> > points;
  std::vector<MapPoint*> mvpLocalMapPoints;
  for (int i = 0; i < n_mp; i++) {
    std::unique_ptr<MapPoint>& p = points[m_id[i]];
    p.reset(new MapPoint());
    p->mnId = (unsigned long)m_id[i]; p->mWorldPos = Mat(3, 1, &m_pos[3 * i]); p->mNormalVector = Mat(3, 1, &m_nrm[3 * i]); p->mfMaxDistance = m_maxd[i]; p->mfMinDistance = m_mind[i];
    p->mDescriptor = MatU8(1, 32); std::memcpy(p->mDescriptor.ptr<unsigned char>(), &m_desc[8 * (size_t)i], 32);
    p->nObs = m_obs[i] ? 3 : 0; p->mbBad = m_skip[i] != 0;
    mvpLocalMapPoints.push_back(p.get());
  }
  Frame Last;
  Last.N = n_last; Last.mnId = 41; Last.mvKeys.resize(n_last); Last.mvKeysUn.resize(n_last); Last.mvpMapPoints.assign(n_last, nullptr); Last.mvbOutlier.assign(n_last, false);
  Last.SetPose(Mat(4, 4, Tlast));
  for (int i = 0; i < n_last; i++) {
    Last.mvKeys[i].octave = l_oct[i]; Last.mvKeysUn[i].angle = l_ang[i];
    std::unique_ptr<MapPoint>& p = points[l_id[i]];
    if (!p) {                                   // a point of the last frame outside the local map (UpdateLastFrame's temporal points are such)
      p.reset(new MapPoint());
      p->mnId = (unsigned long)l_id[i]; p->mWorldPos = Mat(3, 1, &l_pos[3 * i]); p->mDescriptor = MatU8(1, 32); std::memcpy(p->mDescriptor.ptr<unsigned char>(), &l_desc[8 * (size_t)i], 32);
      p->nObs = l_obs[i] ? 3 : 0;
    }
    if (l_valid[i]) Last.mvpMapPoints[i] = p.get();
    else if (tail[1] && (i & 1)) { Last.mvpMapPoints[i] = p.get(); Last.mvbOutlier[i] = true; }          // the two ways the reference skips an entry (:1356-1358)
  }
  std::map<int32_t, std::unique_ptr<MapLine> > lines;
  auto line_obj = [&](const LineSet& L, int i) {
    std::unique_ptr<MapLine>& q = lines[L.id[i]];
    if (!q) {
      q.reset(new MapLine()); q->mnId = (unsigned long)L.id[i];
      for (int c = 0; c < 3; c++) { q->mX0(c) = L.x0[3 * i + c]; q->mDir(c) = L.dir[3 * i + c]; q->mX1(c) = L.x1[3 * i + c]; q->mX2(c) = L.x2[3 * i + c]; }
    }
    if (L.skip[i]) q->mbBad = true;
    return q.get();
  };
  Last.mvpMapLines.assign(n_ll, nullptr);
  for (int i = 0; i < n_ll; i++) Last.mvpMapLines[i] = line_obj(last_lines, i);
  if (n_ll > 0) Last.mDescriptorsLines = Mat(n_ll, dim, last_lines.desc.data());
  std::vector<MapLine*> tracking_local_lines; std::vector<Mat> local_line_descs;
  for (int i = 0; i < n_ml; i++) { tracking_local_lines.push_back(line_obj(local_lines, i)); local_line_descs.push_back(Mat(1, dim, &local_lines.desc[(size_t)i * dim])); }

  lld_amd::Context ctx(0);
  lld_adapter::TrackingMembers tm; tm.gamma = dc[5]; tm.mdThr = dc[7]; tm.mbOnlyTracking = tail[0] != 0;
  lld_adapter::FrameOnDevice dev(ctx.get(), Cur);
  lld_adapter::TrackTrace t1, t2;
  Writer w(out);
  auto put_trace = [&](const lld_adapter::TrackTrace& t) {
    w.put(t.r.pose_qt, 7); w.put(&t.r.chi2, 1);
    const int32_t c[12] = {t.r.n_inliers, t.r.lm_iterations, t.r.lm_trials, t.r.n_edges, t.r.n_search_first, t.r.n_search, t.r.used_wide,
                           t.r.n_points, t.r.n_points_map, t.r.n_lines_matched, t.r.n_lines, t.r.n_discarded};
    w.put(c, 12); w.put(t.kp_point_id); w.put(t.kp_outlier); w.put(t.ln_line_id); w.put(t.ln_outlier);
  };
  auto put_objects = [&]() {                    // what a reader of the object graph sees now
    std::vector<int32_t> kp(nt), ln(nl), pt; std::vector<uint8_t> ko(nt), lo(nl);
    for (int k = 0; k < nt; k++) { kp[k] = Cur.mvpMapPoints[k] ? (int32_t)Cur.mvpMapPoints[k]->mnId : -1; ko[k] = Cur.mvbOutlier[k]; }
    for (int i = 0; i < nl; i++) { ln[i] = Cur.mvpMapLines[i] ? (int32_t)Cur.mvpMapLines[i]->mnId : -1; lo[i] = Cur.mvbOutlierLines[i]; }
    w.put(kp); w.put(ko); w.put(ln); w.put(lo); w.put(Cur.mTcw.ptr<float>(), 16);
    const int32_t n[2] = {(int32_t)points.size(), (int32_t)lines.size()};
    w.put(n, 2);
    for (const auto& e : points) { const int32_t v[5] = {e.first, e.second->mnVisible, e.second->mnFound, (int32_t)e.second->mbTrackInView, (int32_t)e.second->mnLastFrameSeen}; w.put(v, 5); }
    for (const auto& e : lines) { const int32_t v[2] = {e.first, (int32_t)e.second->tracked_last_id}; w.put(v, 2); }
  };
  bool mbVO = false; int mnMatchesInliers = -1;
  const bool ok = dev.TrackWithMotionModel(tm, Cur, Last, &mbVO, &t1);
  put_trace(t1); put_objects();
  if (tail[2]) {
    lld_adapter::FrameOnDevice dev2(ctx.get(), Cur);
    dev2.SetFrameState(tm, Cur);
    dev2.TrackLocalMap(tm, Cur, mvpLocalMapPoints, tracking_local_lines, local_line_descs, &mnMatchesInliers, &t2);
  } else {
    dev.TrackLocalMap(tm, Cur, mvpLocalMapPoints, tracking_local_lines, local_line_descs, &mnMatchesInliers, &t2);
  }
  put_trace(t2); put_objects();
  w.put(t2.mp_in_view);
  const int32_t fin[4] = {(int32_t)ok, (int32_t)mbVO, mnMatchesInliers, Cur.n_set_pose};
  w.put(fin, 4);
  std::printf("adapter-track: TrackWithMotionModel %s (%d points, %d lines), TrackLocalMap %d inliers (%d lines)\n", ok ? "ok" : "lost", t1.r.n_points, t1.r.n_lines, mnMatchesInliers, t2.r.n_lines);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: adapter_harness ba|pose|match|loopmatch|bow|lines|init|lastkf|track <in> <out> [seed]\n"); return 2; }
  const unsigned seed = argc > 4 ? (unsigned)std::atoi(argv[4]) : 1u;
  try {
    const std::string mode = argv[1];
    if (mode == "ba") return run_ba(argv[2], argv[3], seed);
    if (mode == "pose") return run_pose(argv[2], argv[3], seed);
    if (mode == "match") return run_match(argv[2], argv[3]);
    if (mode == "loopmatch") return run_loopmatch(argv[2], argv[3]);
    if (mode == "bow") return run_bow(argv[2], argv[3]);
    if (mode == "lines") return run_lines(argv[2], argv[3]);
    if (mode == "init") return run_init(argv[2], argv[3]);
    if (mode == "lastkf") return run_lastkf(argv[2], argv[3]);
    if (mode == "track") return run_track(argv[2], argv[3]);
    std::fprintf(stderr, "unknown mode %s\n", argv[1]);
    return 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "adapter_harness: %s\n", e.what());
    return 1;
  }
}
