// lld_optimizer_adapter.h — the host adapters of SURVEY.md §8(f) rank 1: Optimizer::LocalBundleAdjustment and
// Optimizer::PoseOptimization with their reference signatures (include/Optimizer.h:49-50) on live SLAM objects, implemented as
// gather -> liblld_amd.so -> scatter.  The bodies (lld_optimizer_adapter.cc) keep the reference's window assembly
// (src/Optimizer.cc:938-1018), replace graph construction + optimisation + classification (:1020-1329) by ONE call of the C ABI,
// and keep the write-back (:1334-1386) under the same mutex; PoseOptimization likewise (:653-932).
//
// The object model is whatever LLD_ADAPTER_OBJECTS_HEADER names (default: this repository's test doubles, lld_slam_objects.h; the
// real classes once OpenCV / Eigen are on the include path).  Everything else is the C++11 standard library and include/lld_amd.hpp.
#ifndef LLD_OPTIMIZER_ADAPTER_H
#define LLD_OPTIMIZER_ADAPTER_H

#include <utility>
#include <vector>

#include "../include/lld_amd.hpp"

#ifndef LLD_ADAPTER_OBJECTS_HEADER
#define LLD_ADAPTER_OBJECTS_HEADER "lld_slam_objects.h"
#endif
#include LLD_ADAPTER_OBJECTS_HEADER

namespace lld_adapter {

using lld_slam::Frame;
using lld_slam::KeyFrame;
using lld_slam::Map;
using lld_slam::MapLine;
using lld_slam::MapPoint;

// What one LocalBundleAdjustment call gathered, got back and scattered (optional; the tests read it, a live system passes nullptr).
struct LbaTrace {
  lld_amd::BAWindow window;                                        // the flat window handed to lld_local_ba_stopflag
  lld_amd::BAOutput output;                                        // what came back
  std::vector<KeyFrame*> cams;                                     // window camera index -> keyframe (free by mnId, then mnId==0, then fixed)
  std::vector<MapPoint*> points;                                   // window point index -> MapPoint (lLocalMapPoints order)
  std::vector<MapLine*> lines;                                     // window line index -> MapLine (lLocalMapLines order)
  std::vector<std::pair<KeyFrame*, MapPoint*> > pt_obs_owner;      // per point observation
  std::vector<std::pair<KeyFrame*, MapLine*> > ln_obs_owner;       // per (line, KF) observation
  std::vector<std::pair<KeyFrame*, MapPoint*> > vToErase;          // Optimizer.cc:1281-1307
  std::vector<std::pair<KeyFrame*, MapLine*> > vToEraseLines;      // :1311-1329 (one entry per outlier EDGE, as there)
  bool returned_before_optimising = false;                         // :1220-1222
  bool solved = false;
};

// void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, double gamma)   src/Optimizer.cc:936-1388
void LocalBundleAdjustment(lld_ctx* ctx, KeyFrame* pKF, bool* pbStopFlag, Map* pMap, double gamma = 1.0, LbaTrace* trace = nullptr);

struct PoseTrace {
  lld_amd::PoseFrame frame;                                        // the flat problem (pose_qt / mvbOutlier* updated by the call)
  std::vector<int> vnIndexEdge;                                    // flat point index -> keypoint index i of the frame
  std::vector<int> vnIndexLines;                                   // flat line index -> line index i of the frame
  bool too_few = false;                                            // nInitialCorrespondences < 3 (:809-810)
};

// int Optimizer::PoseOptimization(Frame* pFrame, double gamma)   src/Optimizer.cc:653-932; returns nInitialCorrespondences - nBad
int PoseOptimization(lld_ctx* ctx, Frame* pFrame, double gamma = 1.0, PoseTrace* trace = nullptr);

}  // namespace lld_adapter
#endif
