// lld_tracking_adapter.cc — see lld_tracking_adapter.h
#include "lld_tracking_adapter.h"

#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>

namespace lld_adapter {

namespace {

void check(int status, const char* what) {
  if (status != LLD_OK) throw std::runtime_error(std::string(what) + ": " + lld_status_string(status));
}

void key_lines(const std::vector<lld_slam::KeyLine>& kl, std::vector<float>& seg, std::vector<int32_t>& octave) {
  seg.resize(4 * kl.size()); octave.resize(kl.size());
  for (size_t i = 0; i < kl.size(); i++) {
    seg[4 * i] = kl[i].startPointX; seg[4 * i + 1] = kl[i].startPointY; seg[4 * i + 2] = kl[i].endPointX; seg[4 * i + 3] = kl[i].endPointY;
    octave[i] = kl[i].octave;
  }
}

// Frame::UpdatePoseMatrices as the frame holds it
lld_frame_view view_of(const Frame& F) {
  lld_frame_view v; std::memset(&v, 0, sizeof v);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) v.Rcw[3 * r + c] = F.mRcw.at<float>(r, c);
    v.tcw[r] = F.mtcw.at<float>(r); v.Ow[r] = F.mOw.at<float>(r);
  }
  v.fx = F.fx; v.fy = F.fy; v.cx = F.cx; v.cy = F.cy; v.bf = F.mbf;
  v.min_x = F.mnMinX; v.max_x = F.mnMaxX; v.min_y = F.mnMinY; v.max_y = F.mnMaxY;
  v.log_scale_factor = F.mfLogScaleFactor; v.n_levels = F.mnScaleLevels;
  return v;
}

// MapLine list -> lld_map_lines (GetMinimalPos, GetMainPoints3D, descriptor row, skip = NULL || tracked_last_id == mnId || isBad:
// src/Tracking.cc:1013-1025; the device applies the "already tracked" rule by id as well)
struct LineSide {
  std::vector<double> x0, dir, x1, x2; std::vector<uint8_t> skip; std::vector<float> desc; std::vector<int32_t> id;
  lld_map_lines m;
  LineSide(const std::vector<MapLine*>& lines, const std::vector<lld_slam::Mat>* descs, const lld_slam::Mat* rows, int dim, unsigned long frame_id) {
    const int n = (int)lines.size();
    x0.assign(3 * (size_t)n + 3, 0.0); dir = x0; x1 = x0; x2 = x0; skip.assign(n + 1, 1); desc.assign((size_t)n * dim + 1, 0.f); id.assign(n + 1, 0);
    for (int i = 0; i < n; i++) {
      MapLine* pML = lines[i];
      if (!pML) continue;
      id[i] = (int32_t)pML->mnId;
      if ((unsigned int)pML->tracked_last_id == frame_id || pML->isBad()) continue;                       // :1018-1025
      skip[i] = 0;
      lld_slam::Vector3d a, d, p1, p2;
      pML->GetMinimalPos(&a, &d); pML->GetMainPoints3D(&p1, &p2);
      for (int k = 0; k < 3; k++) { x0[3 * i + k] = a(k); dir[3 * i + k] = d(k); x1[3 * i + k] = p1(k); x2[3 * i + k] = p2(k); }
      const float* row = (descs && !descs->empty()) ? (*descs)[i].ptr<float>() : rows->ptr<float>(i);       // :1043-1048
      for (int k = 0; k < dim; k++) desc[(size_t)i * dim + k] = row[k];
    }
    m.n = n; m.x0 = x0.data(); m.dir = dir.data(); m.x1 = x1.data(); m.x2 = x2.data(); m.skip = skip.data(); m.desc = desc.data(); m.id = id.data();
  }
};

struct Outputs {
  std::vector<int32_t> kp_id, ln_id; std::vector<uint8_t> kp_out, ln_out, in_view;
  lld_track_result r;
  Outputs(int nt, int nl, int n_mp) : kp_id(nt + 1, -1), ln_id(nl + 1, -1), kp_out(nt + 1, 0), ln_out(nl + 1, 0), in_view(n_mp + 1, 0) {
    std::memset(&r, 0, sizeof r);
    r.kp_point_id = kp_id.data(); r.kp_outlier = kp_out.data(); r.ln_line_id = ln_id.data(); r.ln_outlier = ln_out.data(); r.mp_in_view = n_mp > 0 ? in_view.data() : nullptr;
  }
  void to(TrackTrace* t, int nt, int nl, int n_mp) const {
    if (!t) return;
    t->r = r; t->r.kp_point_id = nullptr; t->r.kp_outlier = nullptr; t->r.ln_line_id = nullptr; t->r.ln_outlier = nullptr; t->r.mp_in_view = nullptr;
    t->kp_point_id.assign(kp_id.begin(), kp_id.begin() + nt); t->kp_outlier.assign(kp_out.begin(), kp_out.begin() + nt);
    t->ln_line_id.assign(ln_id.begin(), ln_id.begin() + nl); t->ln_outlier.assign(ln_out.begin(), ln_out.begin() + nl);
    t->mp_in_view.assign(in_view.begin(), in_view.begin() + n_mp);
  }
};

// pFrame->SetPose(Converter::toCvMat(SE3quat_recov)) (src/Optimizer.cc:915-918) - when PoseOptimization got past its early return
void set_pose(Frame& F, const lld_track_result& r) {
  if (r.n_point_edges < 3) return;
  float T[16];
  lld_se3_to_tcw_f32(r.pose_qt, T);
  F.SetPose(lld_slam::Mat(4, 4, T));
}

}  // namespace

FrameOnDevice::FrameOnDevice(lld_ctx* ctx, const Frame& F) : ctx_(ctx), nt_(F.N), nl_((int)F.mvLinesLeft.size()) {
  std::vector<float> xy(2 * (size_t)F.N + 2), angle(F.N + 1); std::vector<int32_t> octave(F.N + 1);
  for (int k = 0; k < F.N; k++) { xy[2 * k] = F.mvKeysUn[k].pt.x; xy[2 * k + 1] = F.mvKeysUn[k].pt.y; octave[k] = F.mvKeysUn[k].octave; angle[k] = F.mvKeysUn[k].angle; }
  lld_orb_search kp; std::memset(&kp, 0, sizeof kp);
  kp.nt = F.N; kp.t_desc = F.mDescriptors.ptr<uint32_t>(); kp.t_xy = xy.data(); kp.t_octave = octave.data(); kp.t_uright = F.mvuRight.data(); kp.t_angle = angle.data();
  kp.grid_min_x = F.mnMinX; kp.grid_min_y = F.mnMinY; kp.grid_width_inv = F.mfGridElementWidthInv; kp.grid_height_inv = F.mfGridElementHeightInv;
  kp.grid_cols = 64; kp.grid_rows = 48;
  kp.n_levels = F.mnScaleLevels; kp.level_scale = F.mvScaleFactors.data(); kp.level_sigma2 = F.mvLevelSigma2.data(); kp.level_inv_sigma2 = F.mvInvLevelSigma2.data();
  check(lld_frame_create(ctx, &kp, &f_), "lld_frame_create");
  std::vector<float> left, right; std::vector<int32_t> lo, ro, lm(F.line_matches.begin(), F.line_matches.end());
  key_lines(F.mvLinesLeft, left, lo); key_lines(F.mvLinesRight, right, ro);
  dim_ = nl_ > 0 ? F.mDescriptorsLines.cols : 1;
  lld_frame_lines fl; std::memset(&fl, 0, sizeof fl);
  fl.n_left = nl_; fl.left = left.data(); fl.left_octave = lo.data(); fl.n_right = (int)F.mvLinesRight.size(); fl.right = right.data(); fl.right_octave = ro.data();
  fl.line_matches = lm.data(); fl.desc = nl_ > 0 ? F.mDescriptorsLines.ptr<float>() : nullptr; fl.dim = dim_;
  fl.sx = 1.0 / F.mnMaxX; fl.sy = 1.0 / F.mnMaxY;
  const int st = lld_frame_set_lines(f_, nl_ > 0 ? &fl : nullptr);
  if (st != LLD_OK) { lld_frame_destroy(f_); f_ = nullptr; check(st, "lld_frame_set_lines"); }
  lld_track_params_default(&params_);
  params_.cam = lld_camera{F.fx, F.fy, F.cx, F.cy, F.mbf};
}

FrameOnDevice::~FrameOnDevice() { if (f_) lld_frame_destroy(f_); }

bool FrameOnDevice::TrackWithMotionModel(const TrackingMembers& tr, Frame& Cur, const Frame& Last, bool* mbVO, TrackTrace* trace) {
  params_.pose.gamma = tr.gamma; params_.line_md_thr = tr.mdThr;
  // ---- gather LastFrame.mvpMapPoints (src/ORBmatcher.cc:1352-1358, :1381, :1435) and the direction test (:1338-1350)
  const int n = Last.N;
  std::vector<float> pos(3 * (size_t)n + 3, 0.f), angle(n + 1, 0.f); std::vector<uint8_t> valid(n + 1, 0), has_obs(n + 1, 0);
  std::vector<int32_t> octave(n + 1, 0), ids(n + 1, 0); std::vector<uint32_t> desc(8 * (size_t)n + 8, 0u);
  std::unordered_map<int32_t, MapPoint*> point_of;
  for (int i = 0; i < n; i++) {
    MapPoint* pMP = Last.mvpMapPoints[i];
    if (!pMP || Last.mvbOutlier[i]) continue;
    valid[i] = 1; ids[i] = (int32_t)pMP->mnId; point_of[ids[i]] = pMP;
    const lld_slam::Mat x3Dw = pMP->GetWorldPos();
    for (int k = 0; k < 3; k++) pos[3 * i + k] = x3Dw.at<float>(k);
    octave[i] = Last.mvKeys[i].octave; angle[i] = Last.mvKeysUn[i].angle;
    const lld_slam::MatU8 d = pMP->GetDescriptor();
    std::memcpy(&desc[8 * (size_t)i], d.ptr<unsigned char>(), 32);
    has_obs[i] = pMP->Observations() > 0;
  }
  lld_last_frame_points last; std::memset(&last, 0, sizeof last);
  last.n = n; last.world_pos = pos.data(); last.valid = valid.data(); last.octave = octave.data(); last.angle = angle.data(); last.desc = desc.data(); last.has_obs = has_obs.data();
  {
    double accz = 0.0;                                                        // tlc = Rlw*twc+tlw, twc = mOw of the current frame
    for (int k = 0; k < 3; k++) accz += (double)Last.mTcw.at<float>(2, k) * (double)Cur.mOw.at<float>(k);
    const float tlc2 = (float)(accz + (double)Last.mTcw.at<float>(2, 3));
    params_.direction = (tlc2 > Cur.mb) ? 1 : ((-tlc2 > Cur.mb) ? -1 : 0);
  }
  LineSide lines(Last.mvpMapLines, nullptr, &Last.mDescriptorsLines, dim_, Cur.mnId);
  std::unordered_map<int32_t, MapLine*> line_of;
  for (size_t i = 0; i < Last.mvpMapLines.size(); i++) if (Last.mvpMapLines[i]) line_of[(int32_t)Last.mvpMapLines[i]->mnId] = Last.mvpMapLines[i];
  // ---- the stage
  const lld_frame_view view = view_of(Cur);
  double qt[7];
  lld_se3_from_tcw_f32(Cur.mTcw.ptr<float>(), qt);                            // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw))
  check(lld_frame_track_motion_model(f_, &params_, &view, qt, &last, ids.data(), nl_ > 0 ? &lines.m : nullptr), "lld_frame_track_motion_model");
  Outputs out(nt_, nl_, 0);
  check(lld_frame_track_download(f_, &out.r, nullptr), "lld_frame_track_download");
  out.to(trace, nt_, nl_, 0);
  // ---- write-back: matches, PoseOptimization's flags, the outlier discard (:940-975)
  for (int k = 0; k < nt_; k++) {
    Cur.mvpMapPoints[k] = static_cast<MapPoint*>(NULL); Cur.mvbOutlier[k] = false;
    if (out.kp_id[k] < 0) continue;
    MapPoint* pMP = point_of[out.kp_id[k]];
    if (out.kp_out[k]) { pMP->mbTrackInView = false; pMP->mnLastFrameSeen = Cur.mnId; }
    else Cur.mvpMapPoints[k] = pMP;
  }
  for (int i = 0; i < nl_; i++) {
    if (out.ln_id[i] < 0) continue;
    MapLine* pML = line_of[out.ln_id[i]];
    pML->tracked_last_id = (long)Cur.mnId;                                    // AddLinesFrom (:1117); stays when the line is thrown out below
    Cur.mvbOutlierLines[i] = out.ln_out[i] != 0;
    Cur.mvpMapLines[i] = out.ln_out[i] ? static_cast<MapLine*>(NULL) : pML;
  }
  set_pose(Cur, out.r);
  if (out.r.n_search < 10) return false;                                      // :913-917 (the chain has run on; the caller falls back to TrackReferenceKeyFrame)
  if (tr.mbOnlyTracking) { if (mbVO) *mbVO = out.r.n_points_map < 10; return out.r.n_points > 20; }
  return out.r.n_points_map >= 7;
}

void FrameOnDevice::SetFrameState(const TrackingMembers& tr, const Frame& Cur) {
  params_.pose.gamma = tr.gamma; params_.line_md_thr = tr.mdThr;
  std::vector<int32_t> kp_id(nt_ + 1, -1), ln_id(nl_ + 1, -1); std::vector<float> world(3 * (size_t)nt_ + 3, 0.f); std::vector<uint8_t> obs(nt_ + 1, 0), out(nt_ + 1, 0), lout(nl_ + 1, 0);
  std::vector<double> x0(3 * (size_t)nl_ + 3, 0.0), dir(3 * (size_t)nl_ + 3, 0.0);
  for (int k = 0; k < nt_; k++) {
    MapPoint* pMP = Cur.mvpMapPoints[k];
    out[k] = Cur.mvbOutlier[k];
    if (!pMP) continue;
    kp_id[k] = (int32_t)pMP->mnId; obs[k] = pMP->Observations() > 0;
    const lld_slam::Mat P = pMP->GetWorldPos();
    for (int c = 0; c < 3; c++) world[3 * k + c] = P.at<float>(c);
  }
  for (int i = 0; i < nl_; i++) {
    MapLine* pML = i < (int)Cur.mvpMapLines.size() ? Cur.mvpMapLines[i] : static_cast<MapLine*>(NULL);
    lout[i] = i < (int)Cur.mvbOutlierLines.size() && Cur.mvbOutlierLines[i];
    if (!pML) continue;
    ln_id[i] = (int32_t)pML->mnId;
    lld_slam::Vector3d a, d;
    pML->GetMinimalPos(&a, &d);
    for (int c = 0; c < 3; c++) { x0[3 * i + c] = a(c); dir[3 * i + c] = d(c); }
  }
  lld_frame_held held; std::memset(&held, 0, sizeof held);
  held.kp_point_id = kp_id.data(); held.kp_world_pos = world.data(); held.kp_has_obs = obs.data(); held.kp_outlier = out.data();
  held.ln_line_id = ln_id.data(); held.ln_x0 = x0.data(); held.ln_dir = dir.data(); held.ln_outlier = lout.data();
  // (MapPoints / MapLines this frame marked without holding them - mnLastFrameSeen, tracked_last_id - are skipped by TrackLocalMap's own gather)
  const lld_frame_view view = view_of(Cur);
  double qt[7];
  lld_se3_from_tcw_f32(Cur.mTcw.ptr<float>(), qt);
  check(lld_frame_track_set_state(f_, &params_, &view, qt, &held), "lld_frame_track_set_state");
}

void FrameOnDevice::TrackLocalMap(const TrackingMembers& tr, Frame& Cur, const std::vector<MapPoint*>& mvpLocalMapPoints, const std::vector<MapLine*>& local_lines,
                                  const std::vector<lld_slam::Mat>& local_line_descs, int* mnMatchesInliers, TrackTrace* trace) {
  params_.pose.gamma = tr.gamma; params_.line_md_thr = tr.mdThr;
  params_.th_local = tr.just_relocalised ? 5.f : 1.f;
  // ---- SearchLocalPoints, first loop (:1616-1633): host bookkeeping on the MapPoints the frame holds
  std::unordered_map<int32_t, MapPoint*> point_of;
  for (int k = 0; k < nt_; k++) {
    MapPoint* pMP = Cur.mvpMapPoints[k];
    if (!pMP) continue;
    pMP->IncreaseVisible(); pMP->mnLastFrameSeen = Cur.mnId; pMP->mbTrackInView = false;
    point_of[(int32_t)pMP->mnId] = pMP;
  }
  // ---- gather the local map (as lld_orb_search_local_points)
  const int n = (int)mvpLocalMapPoints.size();
  std::vector<float> pos(3 * (size_t)n + 3), nrm(3 * (size_t)n + 3), maxd(n + 1), mind(n + 1); std::vector<uint32_t> desc(8 * (size_t)n + 8);
  std::vector<uint8_t> has_obs(n + 1, 0), skip(n + 1, 1); std::vector<int32_t> ids(n + 1, 0);
  for (int i = 0; i < n; i++) {
    MapPoint* pMP = mvpLocalMapPoints[i];
    ids[i] = (int32_t)pMP->mnId; point_of[ids[i]] = pMP;
    if (pMP->isBad() || pMP->mnLastFrameSeen == Cur.mnId) continue;          // :1639-1642 (the device skips the same ones by id; the objects are the authority)
    skip[i] = 0;
    const lld_slam::Mat P = pMP->GetWorldPos(), Pn = pMP->GetNormal();
    for (int k = 0; k < 3; k++) { pos[3 * i + k] = P.at<float>(k); nrm[3 * i + k] = Pn.at<float>(k); }
    maxd[i] = pMP->GetMaxDistance(); mind[i] = pMP->GetMinDistance();
    const lld_slam::MatU8 d = pMP->GetDescriptor();
    std::memcpy(&desc[8 * (size_t)i], d.ptr<unsigned char>(), 32);
    has_obs[i] = pMP->Observations() > 0;
  }
  lld_map_points mp; std::memset(&mp, 0, sizeof mp);
  mp.n = n; mp.world_pos = pos.data(); mp.normal = nrm.data(); mp.max_distance = maxd.data(); mp.min_distance = mind.data(); mp.desc = desc.data(); mp.has_obs = has_obs.data();
  mp.skip = skip.data();
  LineSide lines(local_lines, &local_line_descs, nullptr, dim_, Cur.mnId);
  std::unordered_map<int32_t, MapLine*> line_of;
  for (int i = 0; i < nl_; i++) if (Cur.mvpMapLines[i]) line_of[(int32_t)Cur.mvpMapLines[i]->mnId] = Cur.mvpMapLines[i];
  for (size_t i = 0; i < local_lines.size(); i++) if (local_lines[i]) line_of[(int32_t)local_lines[i]->mnId] = local_lines[i];
  // ---- the stage
  check(lld_frame_track_local_map(f_, &params_, &mp, ids.data(), nl_ > 0 ? &lines.m : nullptr), "lld_frame_track_local_map");
  Outputs out(nt_, nl_, n);
  check(lld_frame_track_download(f_, nullptr, &out.r), "lld_frame_track_download");
  out.to(trace, nt_, nl_, n);
  // ---- what Frame::isInFrustum and the loop around it leave in the MapPoints (:1637-1650)
  for (int i = 0; i < n; i++) {
    if (skip[i]) continue;
    MapPoint* pMP = mvpLocalMapPoints[i];
    pMP->mbTrackInView = out.in_view[i] != 0;
    if (out.in_view[i]) pMP->IncreaseVisible();
  }
  // ---- matches, PoseOptimization's flags, the statistics loop (:1155-1187)
  int inliers = 0;
  for (int k = 0; k < nt_; k++) {
    Cur.mvpMapPoints[k] = static_cast<MapPoint*>(NULL);
    if (out.kp_id[k] < 0) { continue; }
    MapPoint* pMP = point_of[out.kp_id[k]];
    Cur.mvbOutlier[k] = out.kp_out[k] != 0;
    if (!out.kp_out[k]) {
      Cur.mvpMapPoints[k] = pMP;
      pMP->IncreaseFound();
      if (!tr.mbOnlyTracking) { if (pMP->Observations() > 0) inliers++; } else inliers++;
    }                                                                          // else: STEREO -> mvpMapPoints[k] = NULL, the flag stays (:1170-1171)
  }
  for (int i = 0; i < nl_; i++) {
    if (out.ln_id[i] < 0) { Cur.mvpMapLines[i] = static_cast<MapLine*>(NULL); continue; }
    MapLine* pML = line_of[out.ln_id[i]];
    pML->tracked_last_id = (long)Cur.mnId;
    Cur.mvbOutlierLines[i] = out.ln_out[i] != 0;
    Cur.mvpMapLines[i] = out.ln_out[i] ? static_cast<MapLine*>(NULL) : pML;
  }
  set_pose(Cur, out.r);
  if (mnMatchesInliers) *mnMatchesInliers = inliers;
}

}  // namespace lld_adapter
