// lld_track_internal.h — what the translation units of liblld_amd.so share for the device-resident Tracking-thread chain (lld_frame_track_*,
// lld_frame_track.hip): the resident frame, and launchers that run the searches, the line association and the pose optimisation of the
// single-call entry points on DEVICE arrays, on a stream, without touching the host.  Nothing here is exported (-fvisibility=hidden).
#ifndef LLD_TRACK_INTERNAL_H
#define LLD_TRACK_INTERNAL_H

#include "lld_common.h"

struct lld_frame_track_state;

// A frame whose keypoint side lives on the device for as long as the Tracking thread works on it (round 5, lld_frame_*): descriptors,
// undistorted positions, octaves, right coordinates and angles are uploaded ONCE; the per-frame routines that search this frame
// (lld_frame_search_last_frame: Tracking.cc:904, lld_frame_search_local_points: :1133) then move only their queries and the occupancy bytes.
// Round 6: `track` holds what the reference keeps IN the Frame between those calls (mvpMapPoints, mvbOutlier, mvpMapLines, mvbOutlierLines, mTcw)
// as device arrays, so that the whole sequence runs without a trip through the host (lld_frame_track.hip).
struct lld_frame {
  lld_ctx* ctx = nullptr;
  int nt = 0; bool has_uright = false, has_angle = false;
  char* d = nullptr;                       // one device allocation: desc | xy | octave | uright | angle
  size_t o_td = 0, o_txy = 0, o_toct = 0, o_tur = 0, o_tang = 0;
  lld_orb_search consts;                   // grid constants, n_levels; the level tables are copied below
  float scale[LLD_ORB_MAX_LEVELS], sigma2[LLD_ORB_MAX_LEVELS], inv_sigma2[LLD_ORB_MAX_LEVELS];
  std::vector<int32_t> octave;             // host copy (validation of queries needs none of the rest)
  lld_frame_track_state* track = nullptr;  // owned by lld_frame_track.hip (lld_track::state_free)
};

namespace lld_track {

void state_free(lld_frame* f);             // lld_frame_destroy calls it before the keypoints go

// ---------------------------------------------------------------- guided ORB search on device arrays (lld_orb_search.hip)
// One search = a projection kernel that writes the query records + orb_search_kernel on one workgroup, which also hands the frame what it
// matched (mvpMapPoints[bestIdx] = pMP as an epilogue).  Everything the kernels read or write is a device pointer; the problem is filled on the
// host into pinned memory the caller uploads before the launch.  run_if: when non-null the kernels return at once unless
// (*flag < below) == (want != 0) - the reference's "if(nmatches<20) search again with 2*th" (src/Tracking.cc:907-911) without a host decision.
size_t orbs_problem_bytes();
size_t orbs_qrec_bytes(int nq);
size_t orbs_cache_bytes(int nq);
struct SearchOut { int32_t* match; int32_t* best_dist; int32_t* second_dist; uint8_t* removed; int32_t* owner; int32_t* summary; };
struct RunIf { const int32_t* flag = nullptr; int below = 0; int want = 1; };
// the frame's mvpMapPoints and where the queries' MapPoints are described; counts: [0] n of the first search, [1] n of the search taken, [2] retry used
struct ApplyDev { uint8_t* kp_has; float* kp_world; int32_t* kp_id; uint8_t* kp_obs; const float* q_pos; const int32_t* q_id; const uint8_t* q_obs;
                  int32_t* counts; int min_matches; int is_retry; };
struct LastFrameDev { int n; const float* pos; const uint8_t* valid; const int32_t* octave; const float* angle; const uint8_t* has_obs; };
struct MapPointsDev { int n; const float* pos; const float* normal; const float* maxd; const float* mind; const uint8_t* has_obs; const uint8_t* skip; };
// mode 0: SearchByProjection(Current, Last) rules (TH_HIGH, no ratio, rotation histogram optional); mode 1: SearchByProjection(F, MapPoints) rules
void orbs_fill_problem(const lld_frame* f, int mode, int nq, const uint8_t* d_occupied, const void* d_qrec, const uint32_t* d_qdesc, const SearchOut& out,
                       void* d_cache, float nnratio, int check_orientation, RunIf run_if, const ApplyDev& ap, void* problem_h);
int orbs_project_last_frame(hipStream_t st, const lld_frame* f, const lld_frame_view* view_h, const lld_frame_view* view_d, const LastFrameDev& last, int direction, float th,
                            void* d_qrec, RunIf run_if);
int orbs_project_local_points(hipStream_t st, const lld_frame* f, const lld_frame_view* view_h, const lld_frame_view* view_d, const MapPointsDev& mp, float cos_limit, float th,
                              void* d_qrec, uint8_t* d_in_view, int32_t* d_n_in_view);
int orbs_launch(lld_ctx* ctx, hipStream_t st, const lld_frame* f, const void* problem_d);

// ---------------------------------------------------------------- Tracking::AddLinesFrom on device arrays (lld_match.hip)
struct LineTrackDevParams { double K[9]; double R[9]; double t[3]; double tr[3]; double thr_base, sx, sy; int monocular, use_grid; };   // = LineTrackParams of lld_match.hip
struct LineMapDev { int n; const double* x0; const double* dir; const double* x1; const double* x2; const uint8_t* skip; const float* desc; };
struct LineFrameDev { int n_cur; const float* left; const int32_t* loct; const float* right; const int32_t* lmatch; const uint8_t* occupied; const int32_t* cell;
                      const float* desc; int dim; };
// mCurrentFrame.mvpMapLines[mi] = pML; pML->tracked_last_id = mnId (src/Tracking.cc:1116-1117), done by the assignment kernel itself
struct LineApplyDev { uint8_t* ln_has; double* ln_x0; double* ln_dir; int32_t* ln_id; int32_t* tracked; int32_t* n_tracked; int tracked_cap; const int32_t* map_id; };
size_t line_track_work_bytes(int n_map, int n_cur);
int line_cells_dev(hipStream_t st, const float* d_left, int n, double sx, double sy, int32_t* d_cell);
// params_d: LineTrackDevParams in device memory (written by the tracker's pose kernel); matches_d [n_map]
int line_track_launch_dev(lld_ctx* ctx, hipStream_t st, const LineTrackDevParams* params_d, const LineMapDev& map, const LineFrameDev& cur, double md_thr,
                          void* d_work, int32_t* matches_d, const LineApplyDev& apply);

// ---------------------------------------------------------------- Optimizer::PoseOptimization on the frame's device state (lld_pose.hip)
struct PoseTrackDev {
  int nt, nl;                                          // keypoints / left lines of the frame (upper bounds of the edge counts)
  const float* t_xy; const float* t_uright; const int32_t* t_octave;
  const uint8_t* kp_has; const float* kp_world;
  const float* ln_left; const int32_t* ln_loct; const float* ln_right; const int32_t* ln_roct; const int32_t* ln_match;
  const uint8_t* ln_has; const double* ln_x0; const double* ln_dir;
  const double* pose_qt;                               // [7] device: the estimate the optimisation starts from (Converter::toSE3Quat(pFrame->mTcw))
  lld_camera cam; double gamma; float inv_sigma2[LLD_ORB_MAX_LEVELS];
  // outputs (device): per keypoint / per line outlier flags as PoseOptimization leaves mvbOutlier / mvbOutlierLines for the entries it touched
  uint8_t* kp_outlier; uint8_t* ln_outlier;
  double* pose_out;                                    // [7] + chi2 + (n_inliers, lm_iterations, lm_trials, n_edges, n_points, n_line_edges) as int32: 11 doubles
};
size_t pose_track_work_bytes(int nt, int nl);
int pose_track_launch(lld_ctx* ctx, hipStream_t st, const PoseTrackDev& in, const lld_pose_params& prm, void* d_work);

}  // namespace lld_track

#endif
