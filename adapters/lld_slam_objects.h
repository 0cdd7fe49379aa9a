// lld_slam_objects.h — a minimal object model for COMPILING and RUNNING the host adapters (adapters/lld_optimizer_adapter.cc)
// without the reference's dependencies.
//
// These are this repository's own test doubles, not reference headers: each class carries only the members the adapter touches,
// under the names the reference uses (include/KeyFrame.h, MapPoint.h, MapLine.h, Frame.h, Map.h), so that the adapter source reads
// like a patch of src/Optimizer.cc and compiles unchanged against the real classes once OpenCV / Eigen are present
// (build with -DLLD_ADAPTER_OBJECTS_HEADER='"your_header.h"').  `Mat` stands in for the CV_32F cv::Mat the reference stores poses and
// points in (same accessors: ptr<float>(), at<float>(i), at<float>(r, c), clone()), `Vector3d` for Eigen::Vector3d (operator()(i)).
//
// What is deliberately kept from the reference's data model, because the adapter's correctness depends on it:
//   * observations are std::map<KeyFrame*, size_t>: iteration is in POINTER order (SURVEY hazard 13);
//   * poses and points are float32, lines are double (Converter.cc defines the f32 <-> f64 boundary);
//   * KeyFrame::mnBALocalForKF / mnBAFixedForKF and MapPoint / MapLine::mnBALocalForKF are the visit marks of Optimizer.cc:938-1018.
#ifndef LLD_SLAM_OBJECTS_H
#define LLD_SLAM_OBJECTS_H

#include <cstddef>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace lld_slam {

// ---- cv::Mat (CV_32F) stand-in: row-major floats with shared-nothing value semantics (the reference clones on every Get/Set)
class Mat {
 public:
  Mat() : rows(0), cols(0) {}
  Mat(int r, int c) : rows(r), cols(c), d_((size_t)r * c, 0.f) {}
  Mat(int r, int c, const float* src) : rows(r), cols(c), d_(src, src + (size_t)r * c) {}
  int rows, cols;
  bool empty() const { return d_.empty(); }
  Mat clone() const { return *this; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(d_.data()) + (size_t)r * cols; }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(d_.data()) + (size_t)r * cols; }
  template <class T> T& at(int i) { return d_[(size_t)i]; }
  template <class T> const T& at(int i) const { return d_[(size_t)i]; }
  template <class T> T& at(int r, int c) { return d_[(size_t)r * cols + c]; }
  template <class T> const T& at(int r, int c) const { return d_[(size_t)r * cols + c]; }
 private:
  std::vector<float> d_;
};

// ---- Eigen::Vector3d stand-in
struct Vector3d {
  double v[3];
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};

struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; int octave; };                       // cv::KeyPoint: the adapter reads pt and octave
struct KeyLine {                                                   // cv::line_descriptor::KeyLine (LineMatching.h:27): end points and octave
  float startPointX, startPointY, endPointX, endPointY; int octave;
  KeyLine() : startPointX(0), startPointY(0), endPointX(0), endPointY(0), octave(0) {}
};

class MapPoint;
class MapLine;

class KeyFrame {
 public:
  unsigned long mnId = 0;
  unsigned long mnBALocalForKF = 0, mnBAFixedForKF = 0;            // KeyFrame.h: visit marks of LocalBundleAdjustment
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  Mat mK;                                                          // 3x3 CV_32F
  std::vector<KeyPoint> mvKeysUn;
  std::vector<float> mvuRight;                                     // negative: monocular keypoint
  std::vector<float> mvInvLevelSigma2;
  std::vector<KeyLine> mvLinesLeft, mvLinesRight;
  std::vector<int> line_matches;                                   // index into mvLinesRight, -1: no stereo partner

  Mat GetPose() const { return Tcw.clone(); }
  void SetPose(const Mat& T) { Tcw = T.clone(); n_set_pose++; }
  bool isBad() const { return mbBad; }
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() const { return mvpOrderedConnectedKeyFrames; }
  std::vector<MapPoint*> GetMapPointMatches() const { return mvpMapPoints; }
  std::vector<MapLine*> GetMapLineMatches() const { return mvpMapLines; }
  void EraseMapPointMatch(MapPoint* p) { for (auto& q : mvpMapPoints) if (q == p) q = nullptr; }
  void EraseMapLineMatch(MapLine* l) { for (auto& q : mvpMapLines) if (q == l) q = nullptr; }

  // (test access)
  Mat Tcw;
  bool mbBad = false;
  int n_set_pose = 0;
  std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
  std::vector<MapPoint*> mvpMapPoints;                             // per keypoint, nullptr = none
  std::vector<MapLine*> mvpMapLines;                               // per left line
};

class MapPoint {
 public:
  unsigned long mnId = 0;
  unsigned long mnBALocalForKF = 0;
  static std::mutex mGlobalMutex;
  Mat GetWorldPos() const { return mWorldPos.clone(); }
  void SetWorldPos(const Mat& p) { mWorldPos = p.clone(); n_set_pos++; }
  std::map<KeyFrame*, size_t> GetObservations() const { return mObservations; }
  void EraseObservation(KeyFrame* kf) { mObservations.erase(kf); }
  bool isBad() const { return mbBad; }
  void UpdateNormalAndDepth() { n_update_normal++; }

  Mat mWorldPos;                                                   // 3x1 CV_32F
  std::map<KeyFrame*, size_t> mObservations;                       // pointer-ordered, as in the reference
  bool mbBad = false;
  int n_set_pos = 0, n_update_normal = 0;
};

class MapLine {
 public:
  unsigned long mnId = 0;
  unsigned long mnBALocalForKF = 0;
  void GetMinimalPos(Vector3d* X0, Vector3d* dir) const { *X0 = mX0; *dir = mDir; }
  void SetMinimalPos(const Vector3d& X0, const Vector3d& dir) { mX0 = X0; mDir = dir; n_set_pos++; }
  std::map<KeyFrame*, size_t> GetObservations() const { return mObservations; }
  int Observations() const { return (int)mObservations.size(); }
  void EraseObservation(KeyFrame* kf) { mObservations.erase(kf); }
  bool isBad() const { return mbBad; }

  Vector3d mX0, mDir;
  std::map<KeyFrame*, size_t> mObservations;
  bool mbBad = false;
  int n_set_pos = 0;
};

class Frame {                                                      // what PoseOptimization reads and writes (include/Frame.h)
 public:
  int N = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  Mat mK, mTcw;
  std::vector<KeyPoint> mvKeysUn;
  std::vector<float> mvuRight;
  std::vector<float> mvInvLevelSigma2;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<KeyLine> mvLinesLeft, mvLinesRight;
  std::vector<int> line_matches;
  std::vector<MapLine*> mvpMapLines;
  std::vector<bool> mvbOutlierLines;
  void SetPose(const Mat& T) { mTcw = T.clone(); n_set_pose++; }
  int n_set_pose = 0;
};

class Map {
 public:
  std::mutex mMutexMapUpdate;
};

}  // namespace lld_slam
#endif
