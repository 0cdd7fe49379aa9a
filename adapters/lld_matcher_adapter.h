// lld_matcher_adapter.h — host adapters for the three ORB matchers that run every frame / keyframe, on live SLAM objects:
//   Tracking::SearchLocalPoints                     src/Tracking.cc:1613-1664  (Frame::isInFrustum + ORBmatcher::SearchByProjection(F, points, th))
//   ORBmatcher::SearchByProjection(Current, Last)   src/ORBmatcher.cc:1328-1470 (the matcher of Tracking::TrackWithMotionModel)
//   ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th)    src/ORBmatcher.cc:825-958   (the matcher of LocalMapping::SearchInNeighbors)
// Each is gather -> ONE call of liblld_amd.so (projection loop and search on the device) -> the reference's bookkeeping on the
// objects.  Same object model switch as lld_optimizer_adapter.h (LLD_ADAPTER_OBJECTS_HEADER).  Against the real classes the patch
// adds two trivial getters to MapPoint (GetMinDistance / GetMaxDistance: mfMinDistance and mfMaxDistance are protected and
// MapPoint::PredictScale, which reads them, now runs on the device).
#ifndef LLD_MATCHER_ADAPTER_H
#define LLD_MATCHER_ADAPTER_H

#include <vector>

#include "../include/lld_amd.h"

#ifndef LLD_ADAPTER_OBJECTS_HEADER
#define LLD_ADAPTER_OBJECTS_HEADER "lld_slam_objects.h"
#endif
#include LLD_ADAPTER_OBJECTS_HEADER

namespace lld_adapter {

using lld_slam::Frame;
using lld_slam::KeyFrame;
using lld_slam::MapPoint;

// What one call got back from the device (optional; the tests read it, a live system passes nullptr).
struct MatchTrace {
  std::vector<int32_t> match, best_dist;                           // per query: keypoint index or -1, its distance
  std::vector<uint8_t> removed;                                    // per query: dropped by the rotation histogram
  std::vector<uint8_t> in_view;                                    // SearchLocalPoints: Frame::isInFrustum per local MapPoint
  int nToMatch = 0;
  int direction = 0;                                               // SearchByProjection(Current, Last): +1 bForward, -1 bBackward
};

// void Tracking::SearchLocalPoints() with mCurrentFrame / mvpLocalMapPoints as arguments; th = 1, 3 (RGBD) or 5 (just relocalised)
// as the caller decides (Tracking.cc:1654-1660).  Returns what matcher.SearchByProjection returned (0 when nToMatch == 0).
int SearchLocalPoints(lld_ctx* ctx, Frame& mCurrentFrame, const std::vector<MapPoint*>& mvpLocalMapPoints, int th, MatchTrace* trace = nullptr);

class ORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // src/ORBmatcher.cc:37-39
  ORBmatcher(lld_ctx* ctx, float nnratio = 0.6f, bool checkOri = true) : ctx_(ctx), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  // int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono, MatchTrace* trace = nullptr);
  // int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint *> &vpMapPoints, const float th)
  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0f, MatchTrace* trace = nullptr);

 private:
  lld_ctx* ctx_;
  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace lld_adapter
#endif
