// lld_tracking_adapter.h — host adapter for the Tracking thread's per-frame chain on live SLAM objects (round 6):
//   bool Tracking::TrackWithMotionModel()   src/Tracking.cc:885-994   (from `mCurrentFrame.SetPose(mVelocity*mLastFrame.mTcw)` on)
//   bool Tracking::TrackLocalMap()          src/Tracking.cc:1126-1220 (after UpdateLocalMap())
// Both run as ONE device-resident sequence (lld_frame_track_*, include/lld_amd.h): the frame's keypoints and lines go to the device once
// (FrameOnDevice's constructor - the last step of the reference's Frame constructor), each routine gathers its map-side inputs, queues
// its stage, fetches the stage's record and writes it back into the objects the way the reference's loops do (mvpMapPoints / mvbOutlier /
// mvpMapLines / mvbOutlierLines / mTcw of the frame; mnLastFrameSeen, mbTrackInView, IncreaseVisible / IncreaseFound of the MapPoints;
// tracked_last_id of the MapLines).  Same object model switch as lld_optimizer_adapter.h (LLD_ADAPTER_OBJECTS_HEADER).
// MapPoints and MapLines are named on the device by their mnId (ids below 2^31).
#ifndef LLD_TRACKING_ADAPTER_H
#define LLD_TRACKING_ADAPTER_H

#include <vector>

#include "../include/lld_amd.h"

#ifndef LLD_ADAPTER_OBJECTS_HEADER
#define LLD_ADAPTER_OBJECTS_HEADER "lld_slam_objects.h"
#endif
#include LLD_ADAPTER_OBJECTS_HEADER

namespace lld_adapter {

using lld_slam::Frame;
using lld_slam::MapLine;
using lld_slam::MapPoint;

struct TrackingMembers {          // what the two routines read off `this` (include/Tracking.h)
  double gamma = 0.5;             // yaml `gamma`
  double mdThr = 0.9;             // yaml mdThr
  bool mbOnlyTracking = false;
  bool just_relocalised = false;  // mCurrentFrame.mnId < mnLastRelocFrameId + 2: SearchLocalPoints uses th = 5 (:1656-1658)
};

// What one stage got back (optional; the tests read it, a live system passes nullptr).
struct TrackTrace { lld_track_result r; std::vector<int32_t> kp_point_id, ln_line_id; std::vector<uint8_t> kp_outlier, ln_outlier, mp_in_view; };

class FrameOnDevice {
 public:
  // the frame's own data: mDescriptors, mvKeysUn, mvuRight, the grid and level tables, mvLinesLeft / mvLinesRight / line_matches / mDescriptorsLines
  FrameOnDevice(lld_ctx* ctx, const Frame& mCurrentFrame);
  ~FrameOnDevice();
  FrameOnDevice(const FrameOnDevice&) = delete;
  FrameOnDevice& operator=(const FrameOnDevice&) = delete;

  // bool Tracking::TrackWithMotionModel(): the caller has run UpdateLastFrame() and mCurrentFrame.SetPose(mVelocity*mLastFrame.mTcw), and
  // cleared mCurrentFrame.mvpMapPoints (:897).  *mbVO as the reference sets it in localisation mode (:985).
  bool TrackWithMotionModel(const TrackingMembers& tr, Frame& mCurrentFrame, const Frame& mLastFrame, bool* mbVO = nullptr, TrackTrace* trace = nullptr);
  // The frame's pose and matches came from another routine (Tracking::TrackReferenceKeyFrame, src/Tracking.cc:770-816, or Relocalization): hands the
  // device what mCurrentFrame holds now (mTcw, mvpMapPoints / mvbOutlier, mvpMapLines / mvbOutlierLines) so that TrackLocalMap can follow.
  void SetFrameState(const TrackingMembers& tr, const Frame& mCurrentFrame);
  // bool Tracking::TrackLocalMap() from SearchLocalPoints() on; local_line_descs[i]: the descriptor AddLinesFrom compares local_lines[i] with.
  // Returns mnMatchesInliers through the pointer; the two final tests of the reference (:1212-1219) stay with the caller.
  void TrackLocalMap(const TrackingMembers& tr, Frame& mCurrentFrame, const std::vector<MapPoint*>& mvpLocalMapPoints, const std::vector<MapLine*>& local_lines,
                     const std::vector<lld_slam::Mat>& local_line_descs, int* mnMatchesInliers, TrackTrace* trace = nullptr);

 private:
  lld_ctx* ctx_;
  lld_frame* f_ = nullptr;
  int nt_ = 0, nl_ = 0, dim_ = 1;
  lld_track_params params_;
};

}  // namespace lld_adapter
#endif
