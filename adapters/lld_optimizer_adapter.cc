// lld_optimizer_adapter.cc — see lld_optimizer_adapter.h.  Citations are src/Optimizer.cc of the reference unless stated otherwise.
#include "lld_optimizer_adapter.h"

#include <algorithm>
#include <list>
#include <map>
#include <mutex>

namespace lld_adapter {

using lld_slam::KeyLine;
using lld_slam::KeyPoint;
using lld_slam::Mat;
using lld_slam::Vector3d;

namespace {

// Converter::toCvMat(SE3Quat): 4x4 CV_32F
Mat pose_to_mat(const double* qt7) {
  float T[16];
  lld_se3_to_tcw_f32(qt7, T);
  return Mat(4, 4, T);
}

}  // namespace

void LocalBundleAdjustment(lld_ctx* ctx, KeyFrame* pKF, bool* pbStopFlag, Map* pMap, double gamma, LbaTrace* trace) {
  LbaTrace local_trace;
  LbaTrace& tr = trace ? *trace : local_trace;
  tr = LbaTrace();

  // ---- :938-950  local keyframes: pKF and its covisible keyframes
  std::list<KeyFrame*> lLocalKeyFrames;
  lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (int i = 0, iend = (int)vNeighKFs.size(); i < iend; i++) {
    KeyFrame* pKFi = vNeighKFs[i];
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
  }
  // ---- :953-984  local MapPoints / MapLines seen in the local keyframes (lines need >= 4 observations)
  std::list<MapPoint*> lLocalMapPoints;
  std::list<MapLine*> lLocalMapLines;
  for (std::list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); lit++) {
    const std::vector<MapPoint*> vpMPs = (*lit)->GetMapPointMatches();
    const std::vector<MapLine*> vpMLs = (*lit)->GetMapLineMatches();
    for (MapPoint* pMP : vpMPs)
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
    for (MapLine* pML : vpMLs) {
      if (!pML || pML->isBad()) continue;
      if (pML->Observations() < 4) continue;
      if (pML->mnBALocalForKF != pKF->mnId) { lLocalMapLines.push_back(pML); pML->mnBALocalForKF = pKF->mnId; }
    }
  }
  // ---- :989-1018  fixed keyframes: see local landmarks, are not local
  std::list<KeyFrame*> lFixedCameras;
  for (MapPoint* pMP : lLocalMapPoints) {
    const std::map<KeyFrame*, size_t> observations = pMP->GetObservations();
    for (std::map<KeyFrame*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
      KeyFrame* pKFi = mit->first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
      }
    }
  }
  for (MapLine* pML : lLocalMapLines) {
    const std::map<KeyFrame*, size_t> observations = pML->GetObservations();
    for (std::map<KeyFrame*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
      KeyFrame* pKFi = mit->first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
      }
    }
  }

  // ---- gather: cameras.  g2o numbers the unknowns by vertex id = KeyFrame::mnId (sparse_optimizer.cpp:166-190), so the free cameras go
  //      in ascending mnId; the local keyframe with mnId == 0 is a fixed vertex (:1045), like every lFixedCameras member (:1058)
  lld_amd::BAWindow& w = tr.window;
  std::vector<KeyFrame*>& cams = tr.cams;
  for (KeyFrame* kf : lLocalKeyFrames) if (kf->mnId != 0) cams.push_back(kf);
  std::sort(cams.begin(), cams.end(), [](const KeyFrame* a, const KeyFrame* b) { return a->mnId < b->mnId; });
  w.n_free_cams = (int)cams.size();
  for (KeyFrame* kf : lLocalKeyFrames) if (kf->mnId == 0) cams.push_back(kf);
  for (KeyFrame* kf : lFixedCameras) cams.push_back(kf);
  std::map<KeyFrame*, int> cam_index;
  w.cam_qt.resize(7 * cams.size());
  for (size_t i = 0; i < cams.size(); i++) {
    cam_index[cams[i]] = (int)i;
    const Mat T = cams[i]->GetPose();                                           // CV_32F 4x4
    lld_se3_from_tcw_f32(T.ptr<float>(), &w.cam_qt[7 * i]);                     // == Converter::toSE3Quat
  }
  w.cam = lld_camera{pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf};              // e->fx .. e->bf (:1139-1170); lines: K_eig / mbf of pKF (:1211-1216)

  // ---- gather: points and their observations (:1093-1178), in the reference's insertion order: list order, then the
  //      std::map<KeyFrame*, size_t> of every point, i.e. pointer order
  tr.points.assign(lLocalMapPoints.begin(), lLocalMapPoints.end());
  for (MapPoint* pMP : tr.points) {
    const Mat X = pMP->GetWorldPos();
    for (int k = 0; k < 3; k++) w.pt_xyz.push_back((double)X.at<float>(k));      // Converter::toVector3d
    const std::map<KeyFrame*, size_t> observations = pMP->GetObservations();
    for (std::map<KeyFrame*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
      KeyFrame* pKFi = mit->first;
      if (pKFi->isBad()) continue;
      const std::map<KeyFrame*, int>::const_iterator ci = cam_index.find(pKFi);
      if (ci == cam_index.end()) continue;       // (cannot happen: every observer that is not bad is local or fixed; the reference would dereference a null vertex here)
      const KeyPoint& kpUn = pKFi->mvKeysUn[mit->second];
      w.pt_obs_cam.push_back(ci->second);
      w.pt_obs_uvr.push_back((double)kpUn.pt.x); w.pt_obs_uvr.push_back((double)kpUn.pt.y);
      w.pt_obs_uvr.push_back((double)pKFi->mvuRight[mit->second]);              // < 0: monocular edge (:1119)
      w.pt_obs_inv_sigma2.push_back((double)pKFi->mvInvLevelSigma2[kpUn.octave]);
      tr.pt_obs_owner.push_back(std::make_pair(pKFi, pMP));
    }
    w.pt_obs_start.push_back((int32_t)w.pt_obs_cam.size());
  }
  // ---- gather: lines (:1185-1218, LineOptimizer::AddLineMinimal): proj_map is a std::map<int, ...> keyed by KeyFrame::mnId, so the
  //      edges of a line are created in ascending mnId
  tr.lines.assign(lLocalMapLines.begin(), lLocalMapLines.end());
  for (MapLine* pML : tr.lines) {
    Vector3d X0, line_dir;
    pML->GetMinimalPos(&X0, &line_dir);
    for (int k = 0; k < 3; k++) { w.line_x0.push_back(X0(k)); w.line_dir.push_back(line_dir(k)); }
    const std::map<KeyFrame*, size_t> observations = pML->GetObservations();
    std::map<int, std::pair<KeyFrame*, size_t> > by_id;
    for (std::map<KeyFrame*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); mit++)
      if (!mit->first->isBad()) by_id.insert(std::make_pair((int)mit->first->mnId, std::make_pair(mit->first, mit->second)));
    for (std::map<int, std::pair<KeyFrame*, size_t> >::const_iterator it = by_id.begin(); it != by_id.end(); it++) {
      KeyFrame* pKFcurr = it->second.first; const size_t idx = it->second.second;
      const std::map<KeyFrame*, int>::const_iterator ci = cam_index.find(pKFcurr);
      if (ci == cam_index.end()) continue;
      const KeyLine& kl = pKFcurr->mvLinesLeft[idx];
      w.ln_obs_cam.push_back(ci->second);
      const double left[4] = {kl.startPointX, kl.startPointY, kl.endPointX, kl.endPointY};
      w.ln_obs_left.insert(w.ln_obs_left.end(), left, left + 4);
      int oct_r = 0;
      if (pKFcurr->line_matches[idx] >= 0) {
        const KeyLine& kr = pKFcurr->mvLinesRight[pKFcurr->line_matches[idx]];
        const double right[4] = {kr.startPointX, kr.startPointY, kr.endPointX, kr.endPointY};   // a right line with startPointX < 0 counts as "no stereo" there too (LineOptimizer.cc:60)
        w.ln_obs_right.insert(w.ln_obs_right.end(), right, right + 4);
        oct_r = kr.octave;
      } else {
        const double none[4] = {-1, -1, -1, -1};                                  // kl_empty (:1203-1209)
        w.ln_obs_right.insert(w.ln_obs_right.end(), none, none + 4);
      }
      w.ln_obs_octave.push_back(kl.octave); w.ln_obs_octave.push_back(oct_r);     // information = gamma^2 / GetReprojThrPyramid(1, octave)^2 per edge
      tr.ln_obs_owner.push_back(std::make_pair(pKFcurr, pML));
    }
    w.ln_obs_start.push_back((int32_t)w.ln_obs_cam.size());
  }

  // ---- :1020-1329  graph, optimize(5), classification, optimize(15), erase lists: one call
  const lld_ba_window cw = w.view();
  lld_amd::BAOutput& o = tr.output;
  o.cam_qt.resize(7 * (size_t)cw.n_cams); o.pt_xyz.resize(3 * (size_t)cw.n_points);
  o.line_x0.resize(3 * (size_t)cw.n_lines); o.line_dir.resize(3 * (size_t)cw.n_lines);
  o.pt_obs_outlier.resize(cw.n_pt_obs); o.ln_edge_outlier.resize(2 * (size_t)cw.n_ln_obs); o.line_removed.resize(cw.n_lines);
  lld_ba_result r{};
  r.cam_qt = o.cam_qt.data(); r.pt_xyz = o.pt_xyz.data(); r.line_x0 = o.line_x0.data(); r.line_dir = o.line_dir.data();
  r.pt_obs_outlier = o.pt_obs_outlier.data(); r.ln_edge_outlier = o.ln_edge_outlier.data(); r.line_removed = o.line_removed.data();
  lld_ba_params p; lld_ba_params_default(&p); p.gamma = gamma;
  // pbStopFlag is LocalMapping's plain bool (LocalMapping.cc:119 writes it from the Tracking thread): the byte form of the entry point reads it in place
  lld_amd::check(lld_local_ba_stopflag(ctx, &cw, &p, reinterpret_cast<volatile const unsigned char*>(pbStopFlag), &r), "lld_local_ba_stopflag");
  o.stats = r.stats;
  tr.solved = true;
  if (o.stats.aborted && o.stats.lm_iterations[0] == 0 && o.stats.lm_trials[0] == 0) {      // :1220-1222 return before optimising: the map is not touched
    tr.returned_before_optimising = true;
    return;
  }

  // ---- :1278-1329  erase lists
  for (size_t k = 0; k < tr.pt_obs_owner.size(); k++) if (o.pt_obs_outlier[k]) tr.vToErase.push_back(tr.pt_obs_owner[k]);
  for (size_t k = 0; k < tr.ln_obs_owner.size(); k++)
    for (int si = 0; si < 2; si++) if (o.ln_edge_outlier[2 * k + si]) tr.vToEraseLines.push_back(tr.ln_obs_owner[k]);   // GetLineData lists the keyframe once per outlier edge

  // ---- :1334-1386  write-back under the map mutex
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
  for (size_t i = 0; i < tr.vToErase.size(); i++) {
    KeyFrame* pKFi = tr.vToErase[i].first; MapPoint* pMPi = tr.vToErase[i].second;
    pKFi->EraseMapPointMatch(pMPi);
    pMPi->EraseObservation(pKFi);
  }
  for (size_t i = 0; i < tr.vToEraseLines.size(); i++) {
    KeyFrame* pKFi = tr.vToEraseLines[i].first; MapLine* pMLi = tr.vToEraseLines[i].second;
    pKFi->EraseMapLineMatch(pMLi);
    pMLi->EraseObservation(pKFi);
  }
  // keyframes: EVERY local keyframe is written, the fixed mnId == 0 one included (its estimate makes the round trip through SE3Quat)
  for (KeyFrame* kf : lLocalKeyFrames) kf->SetPose(pose_to_mat(&o.cam_qt[7 * (size_t)cam_index[kf]]));
  for (size_t k = 0; k < tr.points.size(); k++) {
    Mat X(3, 1);
    for (int c = 0; c < 3; c++) X.at<float>(c) = (float)o.pt_xyz[3 * k + c];      // Converter::toCvMat(Vector3d)
    tr.points[k]->SetWorldPos(X);
    tr.points[k]->UpdateNormalAndDepth();
  }
  for (size_t k = 0; k < tr.lines.size(); k++) {
    if (o.line_removed[k]) continue;                                              // GetLineData returned false: vertex deleted by DisableOutliers
    tr.lines[k]->SetMinimalPos(Vector3d(o.line_x0[3 * k], o.line_x0[3 * k + 1], o.line_x0[3 * k + 2]),
                               Vector3d(o.line_dir[3 * k], o.line_dir[3 * k + 1], o.line_dir[3 * k + 2]));
  }
}

int PoseOptimization(lld_ctx* ctx, Frame* pFrame, double gamma, PoseTrace* trace) {
  PoseTrace local_trace;
  PoseTrace& tr = trace ? *trace : local_trace;
  tr = PoseTrace();
  lld_amd::PoseFrame& f = tr.frame;
  f.cam = lld_camera{pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, pFrame->mbf};
  lld_se3_from_tcw_f32(pFrame->mTcw.ptr<float>(), f.pose_qt);                      // Converter::toSE3Quat(pFrame->mTcw)  (:669)
  int nInitialCorrespondences = 0;
  {
    std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);                     // :715
    const int N = pFrame->N;
    for (int i = 0; i < N; i++) {
      MapPoint* pMP = pFrame->mvpMapPoints[i];
      if (!pMP) continue;
      nInitialCorrespondences++;
      pFrame->mvbOutlier[i] = false;                                               // :723,755
      const KeyPoint& kpUn = pFrame->mvKeysUn[i];
      const Mat Xw = pMP->GetWorldPos();
      for (int k = 0; k < 3; k++) f.pt_xw.push_back((double)Xw.at<float>(k));
      f.pt_uvr.push_back((double)kpUn.pt.x); f.pt_uvr.push_back((double)kpUn.pt.y); f.pt_uvr.push_back((double)pFrame->mvuRight[i]);   // < 0: monocular edge
      f.pt_inv_sigma2.push_back((double)pFrame->mvInvLevelSigma2[kpUn.octave]);
      tr.vnIndexEdge.push_back(i);
    }
    for (size_t i = 0; i < pFrame->mvpMapLines.size(); i++) {                      // :796-804, AddLineMinOnlyPose :562-650
      MapLine* pML = pFrame->mvpMapLines[i];
      if (!pML) continue;
      Vector3d X0, line_dir;
      pML->GetMinimalPos(&X0, &line_dir);
      for (int k = 0; k < 3; k++) { f.ln_x0.push_back(X0(k)); f.ln_dir.push_back(line_dir(k)); }
      const KeyLine& kl = pFrame->mvLinesLeft[i];
      const double left[4] = {kl.startPointX, kl.startPointY, kl.endPointX, kl.endPointY};
      f.ln_left.insert(f.ln_left.end(), left, left + 4);
      int oct_r = 0;
      if (pFrame->line_matches[i] >= 0) {
        const KeyLine& kr = pFrame->mvLinesRight[pFrame->line_matches[i]];
        // the library tells "stereo line" by a right start point >= 0; here the reference asks line_matches (:575,:581), so a matched right
        // line whose detector put its start point left of the image border would lose its right edge: clamp it (never seen on undistorted KITTI lines)
        const double right[4] = {kr.startPointX < 0 ? 0.0 : (double)kr.startPointX, kr.startPointY, kr.endPointX, kr.endPointY};
        f.ln_right.insert(f.ln_right.end(), right, right + 4);
        oct_r = kr.octave;
      } else {
        const double none[4] = {-1, -1, -1, -1};
        f.ln_right.insert(f.ln_right.end(), none, none + 4);
      }
      f.ln_octave.push_back(kl.octave); f.ln_octave.push_back(oct_r);
      f.ln_frame_index.push_back((int32_t)i);                                      // vnIndexLines (:640): mvbOutlierLines AND vnStereoLines are indexed by it (:893-898)
      tr.vnIndexLines.push_back((int)i);
    }
  }
  if (nInitialCorrespondences < 3) { tr.too_few = true; return 0; }                // :809-810
  size_t n_line_edges = 0;
  for (size_t k = 0; k < tr.vnIndexLines.size(); k++) n_line_edges += f.ln_right[4 * k] < 0 ? 1 : 2;
  lld_pose_problem q{};
  q.cam = f.cam;
  for (int i = 0; i < 7; i++) q.pose_qt[i] = f.pose_qt[i];
  q.n_points = (int)tr.vnIndexEdge.size(); q.pt_xw = f.pt_xw.data(); q.pt_uvr = f.pt_uvr.data(); q.pt_inv_sigma2 = f.pt_inv_sigma2.data();
  q.n_lines = (int)tr.vnIndexLines.size(); q.ln_x0 = f.ln_x0.data(); q.ln_dir = f.ln_dir.data(); q.ln_left = f.ln_left.data(); q.ln_right = f.ln_right.data();
  q.ln_octave = f.ln_octave.data(); q.ln_frame_index = f.ln_frame_index.data();
  lld_pose_params pp; lld_pose_params_default(&pp); pp.gamma = gamma;
  f.mvbOutlier.assign(q.n_points > 0 ? q.n_points : 1, 0); f.mvbOutlierLines.assign(q.n_lines > 0 ? q.n_lines : 1, 0);
  lld_pose_result pr{};
  pr.pt_outlier = f.mvbOutlier.data(); pr.ln_outlier = f.mvbOutlierLines.data();
  lld_amd::check(lld_pose_opt(ctx, &q, &pp, &pr), "lld_pose_opt");
  f.mvbOutlier.resize(q.n_points); f.mvbOutlierLines.resize(q.n_lines);
  for (int i = 0; i < 7; i++) f.pose_qt[i] = pr.pose_qt[i];
  const int n_in = pr.n_inliers;
  for (size_t k = 0; k < tr.vnIndexEdge.size(); k++) pFrame->mvbOutlier[tr.vnIndexEdge[k]] = f.mvbOutlier[k] != 0;
  // the line flags are assigned inside the rounds only when the graph has at least 10 edges (:887-888 breaks before the line loop)
  if ((size_t)nInitialCorrespondences + n_line_edges >= 10)
    for (size_t k = 0; k < tr.vnIndexLines.size(); k++) pFrame->mvbOutlierLines[tr.vnIndexLines[k]] = f.mvbOutlierLines[k] != 0;
  float T[16];
  lld_se3_to_tcw_f32(f.pose_qt, T);
  pFrame->SetPose(Mat(4, 4, T));                                                   // :916-919
  return n_in;                                                                     // :931
}

}  // namespace lld_adapter
