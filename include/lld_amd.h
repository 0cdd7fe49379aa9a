/*
 * lld_amd.h — C ABI of the MI355X-native point+line local-BA / pose-optimisation /
 * descriptor-matching core.
 *
 * This is the drop-in boundary for the hot path of alexandervakhitov/lld-slam.  The
 * reference has no FFI layer: its boundary is a set of C++ static/member functions that
 * take live SLAM objects (include/Optimizer.h:49-50, include/ORBmatcher.h:41-83,
 * include/TwoFrameLineMatcher.h:31-42).  Each entry point below names the reference
 * function it stands in for; the host adapter that gathers KeyFrame/MapPoint/MapLine
 * state into these flat structs is sketched in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, caller-allocated buffers, `int` status return (0 ok, <0 error), no globals;
 *   - a context is bound to one HIP device and one stream; it is re-entrant per handle
 *     (one handle per host thread, as Tracking / LocalMapping each would own one);
 *   - all pointers in the *input* structs are HOST pointers unless the function name ends
 *     in `_dev`; batch handles keep their inputs resident in HBM between solves;
 *   - poses are world->camera, stored as 7 doubles (qx,qy,qz,qw,tx,ty,tz) exactly as
 *     g2o::SE3Quat holds them (Thirdparty/g2o/g2o/types/se3quat.h:47-48);
 *   - floating point parity target: 1e-5 relative on final chi2 / poses / landmarks,
 *     identical outlier sets; matcher indices and integer distances bit-exact.
 *
 * The same structs are consumed by the CPU oracle (oracle/lld_oracle.cpp, symbols
 * `lldo_*`), which is test infrastructure only.
 */
#ifndef LLD_AMD_H
#define LLD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* liblld_amd.so is built with -fvisibility=hidden: only the declarations of this header are exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ------------------------------------------------------------------ status codes */
#define LLD_OK               0
#define LLD_ERR_INVALID     -1   /* bad argument / inconsistent sizes               */
#define LLD_ERR_NO_DEVICE   -2   /* no HIP device: the product path never falls back */
#define LLD_ERR_HIP         -3   /* a HIP runtime call failed                        */
#define LLD_ERR_ALLOC       -4
#define LLD_ERR_UNSUPPORTED -5   /* size outside the compiled limits                 */

const char* lld_status_string(int status);

/* ------------------------------------------------------------------ context */
typedef struct lld_ctx lld_ctx;

/* Binds to HIP device `device`, creates a private stream.  Fails with LLD_ERR_NO_DEVICE
 * when no GPU is visible (there is no CPU fallback). */
int  lld_ctx_create(int device, lld_ctx** out);
void lld_ctx_destroy(lld_ctx* ctx);
/* Stream the context launches on (hipStream_t as void*), so callers can record events. */
void* lld_ctx_stream(lld_ctx* ctx);
int  lld_ctx_synchronize(lld_ctx* ctx);
/* Threading and memory contract of a context.
 *   - ONE host thread drives a context at a time (Tracking and LocalMapping each own one).  Two threads on one handle are not
 *     supported; the one case the library guards is two lld_ba_batch_create calls racing for the cached resources below (the flag
 *     is taken with an atomic exchange: the loser gets private resources and frees them with its batch).
 *   - A context KEEPS what the batched local BA needs between batches, grow-only: the device slab of the largest batch created so
 *     far (3.4 GB for 256 LBA-B windows, 13 MB for one), two pinned upload arenas (1.1 GB for that batch), the pinned landing buffer
 *     of the result records (100 MB), the group streams and events.  lld_ba_batch_destroy does NOT return them - allocating and,
 *     worse, freeing them per batch (hipFree synchronises the device) was most of the cost of a pipelined caller.  At most one live
 *     batch per context borrows the cached set; a second live batch on the same context allocates its own and frees it on destroy.
 *   - lld_ctx_release_cache gives the cached memory back without destroying the context (e.g. after a one-off global BA, or when a
 *     pipelined caller goes idle).  It fails with LLD_ERR_INVALID while a live batch borrows the set; the next batch re-grows it.
 *   - Environment: the library reads exactly one variable, LLD_HOST_THREADS (host threads that flatten / unpack a batch, 1..64,
 *     default min(cores, 16)).  Experiment knobs exist only in the experiments build (make -C lld_slam_amd/csrc exp). */
int  lld_ctx_release_cache(lld_ctx* ctx);

/* ------------------------------------------------------------------ shared types */
typedef struct {
  double fx, fy, cx, cy;   /* pinhole; line edges use fx for both axes (LineOptimizer.cc:66-68) */
  double bf;               /* baseline * fx (KeyFrame::mbf)                                      */
} lld_camera;

/* Converter::toSE3Quat (src/Converter.cc:37-47): float 4x4 row-major Tcw -> SE3Quat 7-vector. */
void lld_se3_from_tcw_f32(const float* tcw16, double* qt7);
/* Converter::toCvMat(SE3Quat) (src/Converter.cc:49-70): SE3Quat -> float 4x4 row-major. */
void lld_se3_to_tcw_f32(const double* qt7, float* tcw16);
/* ORBextractor level table mvInvLevelSigma2 (src/ORBextractor.cc:416-430), float arithmetic. */
void lld_orb_inv_level_sigma2(float scale_factor, int n_levels, float* out);

/* ================================================================== local bundle adjustment
 * Stands in for Optimizer::LocalBundleAdjustment (src/Optimizer.cc:936-1388) from the point
 * where the local window has been collected (:938-1018) to the point where results are
 * written back (:1334-1386), including LineOptimizer::{AddLineMinimal,DisableOutliers,
 * GetLineData} (src/LineOptimizer.cc:39-201) and everything g2o does underneath.
 *
 * Window layout (the order is the reference's insertion order, so sums run the same way):
 *   cameras   [0,n_free_cams) are optimised, in ascending KeyFrame::mnId order (g2o orders
 *             unknowns by vertex id, sparse_optimizer.cpp:166-190); [n_free_cams,n_cams) are
 *             fixed (lFixedCameras and the mnId==0 keyframe, Optimizer.cc:1037-1063).
 *   points    each point owns a contiguous run of observations pt_obs_start[p]..[p+1]
 *             (the loop over MapPoint::GetObservations, Optimizer.cc:1107-1178).
 *             uR < 0 marks a monocular observation (Optimizer.cc:1119).
 *   lines     each line owns a run of (line,KF) observations ln_obs_start[l]..[l+1]
 *             (proj_map, Optimizer.cc:1189-1218); every observation yields a left-image edge
 *             and, when right xs >= 0, a right-image edge (LineOptimizer.cc:58-65).
 */
typedef struct {
  lld_camera cam;
  int32_t n_cams;
  int32_t n_free_cams;
  const double*  cam_qt;             /* [n_cams][7]                                           */

  int32_t n_points;
  const double*  pt_xyz;             /* [n_points][3]  (Converter::toVector3d of the f32 pos) */
  const int32_t* pt_obs_start;       /* [n_points+1]                                          */
  int32_t n_pt_obs;
  const int32_t* pt_obs_cam;         /* [n_pt_obs] camera index                               */
  const double*  pt_obs_uvr;         /* [n_pt_obs][3] u, v, uR (uR<0: mono)                   */
  const double*  pt_obs_inv_sigma2;  /* [n_pt_obs] mvInvLevelSigma2[octave] widened           */

  int32_t n_lines;
  const double*  line_x0;            /* [n_lines][3]  MapLine::GetMinimalPos                  */
  const double*  line_dir;           /* [n_lines][3]                                          */
  const int32_t* ln_obs_start;       /* [n_lines+1]                                           */
  int32_t n_ln_obs;
  const int32_t* ln_obs_cam;         /* [n_ln_obs]                                            */
  const double*  ln_obs_left;        /* [n_ln_obs][4] xs,ys,xe,ye of the left KeyLine         */
  const double*  ln_obs_right;       /* [n_ln_obs][4] right KeyLine; xs<0 -> no stereo match  */
  const int32_t* ln_obs_octave;      /* [n_ln_obs][2] octave of left / right KeyLine          */
} lld_ba_window;

typedef struct {
  double  gamma;            /* line weight; LocalMapping passes 1.0 (Optimizer.h:49)           */
  int32_t its_round1;       /* 5  (Optimizer.cc:1224); >= 1: optimize(0) evaluates no error, the classification that follows would read g2o's uninitialised _error (undefined in the reference) -> LLD_ERR_INVALID */
  int32_t its_round2;       /* 15 (Optimizer.cc:1273); >= 1 for protocol 0 */
  int32_t ln_filter;        /* 4  (LineOptimizer.h:90)                                         */
  int32_t max_trials;       /* 10 (maxTrialsAfterFailure, optimization_algorithm_levenberg.cpp:50) */
  double  pcg_rel_tol;      /* reduced-system PCG stops at |r|_M / |b|_M <= tol (GPU only)     */
  int32_t pcg_max_iter;     /* 0 -> 10 * 6 * n_free_cams                                        */
  int32_t reduced_solver;   /* GPU only: 0 = exact Cholesky (default; the reference factorises exactly,
                               linear_solver_eigen.h:94-124) on the fp64 matrix cores when 6*n_free <= 304 - along the
                               block structure of S in an elimination order chosen per window, as the reference's sparse
                               LDL^T does after computeSymbolicDecomposition (linear_solver_eigen.h:147-232), wherever the
                               symbolic factor fits the kernel (lld_ba_chol_plan), dense otherwise,
                               1 = block-Jacobi PCG, 2 = exact 6x6-block Cholesky on the vector ALUs,
                               3 = the matrix-core Cholesky over all tiles in the caller's camera order (round 4's default);
                               diagnostic: 4 = structure-following in the caller's camera order as ONE chain of tile columns,
                               5 = structure-following with the two-chain (separator) plan only, dense where none exists */
  int32_t protocol;         /* 0 = Optimizer::LocalBundleAdjustment: optimize(its_round1), outlier protocol, optimize(its_round2).
                               1 = Optimizer::BundleAdjustment / GlobalBundleAdjustment (src/Optimizer.cc:312-559) on the same
                                   kernels: ONE optimize(its_round1) call and nothing else - no classification, no line removal, all
                                   result flags 0; line edges carry identity information and the Huber delta thHuber3D/2
                                   (AddLineMinimalGlobal, :149-240), `gamma` and `ln_filter` are ignored.  The window holds the whole
                                   map: every keyframe but mnId==0 free.  Limits of this build: n_free_cams <= 170 per window in general,
                                   <= 8192 when the batch has at most 8 windows (then the reduced system - dense, 6 n_free squared
                                   doubles of HBM - is solved by the multi-workgroup PCG whatever `reduced_solver` says, except 2;
                                   beyond 590 cameras the camera accumulators and pose copies of the landmark kernels live in HBM
                                   instead of LDS and are summed with global fp64 atomics: results of such maps are reproducible
                                   to rounding, not bit for bit, from run to run; parity-tested to 600 free cameras, timed to 4000) */
  int32_t robust_points;    /* protocol 1 only: bRobust (Huber kernels on the point edges, default 1); lines are always robust */
  int32_t abort_after_trials; /* TEST HOOK, 0 = off: behave as if *abort_flag had been raised right after the k-th LM trial of the
                               window (trials counted over both rounds) and stayed up - a deterministic stand-in for the asynchronous
                               pbStopFlag, honoured identically by the library and by the CPU oracle (tests/test_gpu_ba.py)          */
  int32_t deterministic;    /* 2 (default): bit-reproducible wherever the build can be - every wavefront of a linearisation workgroup adds the
                               per-camera sums Hpp / b_p into its OWN accumulator copy in program order, and the copies, workgroup partials
                               and everything downstream are summed in a fixed order: two solves of the same input on the same build are
                               BIT-IDENTICAL, as the reference is within a run (it walks its edges in a fixed order,
                               sparse_optimizer.cpp:482-487).  Maps beyond 590 free cameras keep their accumulators in HBM under global
                               atomics and fall back to "agree to rounding" silently.
                               1: the same, but such a map is refused (LLD_ERR_UNSUPPORTED) instead of falling back.
                               0: all wavefronts of a workgroup share the accumulator copies (LDS fp64 atomics whose order varies from run
                               to run): two solves agree to rounding (1e-16 per sum, amplified by 20 LM iterations to 1e-9 .. 1e-6 on the
                               weakest landmarks, and an observation whose chi2 ends within that of a threshold may come out on either side).
                               Faster by 0.4 - 0.7 % on 256 LBA-B windows (`secondary.deterministic` of the bench line); the default
                               until round 4. */
} lld_ba_params;

/* ALWAYS start from lld_ba_params_default(): a zero-initialised struct is NOT the default (deterministic = 0 selects the shared-accumulator
 * mode, its_* = 0 is refused), and fields added by later versions get their defaults here. */
void lld_ba_params_default(lld_ba_params* p);

typedef struct {
  double  chi2_round1;      /* LM cost (robust) after optimize(its_round1)                      */
  double  chi2_final;       /* LM cost after optimize(its_round2) (kernels removed)             */
  int32_t lm_iterations[2]; /* outer iterations executed per round                              */
  int32_t lm_trials[2];     /* linear solves (trials) executed per round                        */
  int32_t pcg_iterations;   /* total PCG iterations (0 for the oracle's direct solve)           */
  int32_t n_pt_obs_outlier; /* size of vToErase                                                 */
  int32_t n_ln_edge_outlier;
  int32_t n_lines_removed;
  int32_t aborted;          /* 1 iff the stop flag was up at the protocol's LAST poll: the check before optimising (Optimizer.cc:1220,
                               nothing is touched then), the check after optimize(its_round1) (:1230, round 2 is skipped, the final
                               classification still runs on the round-1 state), or - when round 2 ran - the last terminate() of
                               optimize(its_round2) (sparse_optimizer.cpp:376, levenberg.cpp:149).  The reference returns void; the
                               adapter needs only "aborted && lm_iterations[0]==0 -> leave the map alone"                          */
  int32_t reserved;
} lld_ba_stats;

typedef struct {
  double*  cam_qt;          /* [n_cams][7]    optimised poses (fixed ones copied through)        */
  double*  pt_xyz;          /* [n_points][3]                                                     */
  double*  line_x0;         /* [n_lines][3]   LineOptimizer::GetLineData; removed lines keep input */
  double*  line_dir;        /* [n_lines][3]                                                      */
  uint8_t* pt_obs_outlier;  /* [n_pt_obs]     1 -> (KF,MapPoint) goes to vToErase (Optimizer.cc:1281-1307) */
  uint8_t* ln_edge_outlier; /* [n_ln_obs][2]  1 -> kf id pushed by GetLineData for left/right edge */
  uint8_t* line_removed;    /* [n_lines]      1 -> vertex deleted by DisableOutliers            */
  lld_ba_stats stats;
} lld_ba_result;

/* One window, synchronous.  `abort_flag` may be NULL; it is the reference's pbStopFlag
 * (Optimizer.cc:1030-1031, sparse_optimizer.h:188), polled between LM trials. */
int lld_local_ba(lld_ctx* ctx, const lld_ba_window* in, const lld_ba_params* params,
                 volatile const int* abort_flag, lld_ba_result* out);

/* The same call with the stop flag as a BYTE: the reference's pbStopFlag is a `bool*` (LocalMapping::mbAbortBA, Optimizer.h:49); an
 * adapter passes it as `(volatile const unsigned char*)pbStopFlag` - character types may alias any object, an `int*` may not. */
int lld_local_ba_stopflag(lld_ctx* ctx, const lld_ba_window* in, const lld_ba_params* params,
                          volatile const unsigned char* stop_flag, lld_ba_result* out);

/* Batched, HBM-resident form: windows are uploaded once, then solved any number of times
 * (each solve restarts from the uploaded initial state).  This is the throughput path:
 * independent windows are what shards across GPUs (one batch per rank). */
typedef struct lld_ba_batch lld_ba_batch;
int  lld_ba_batch_create(lld_ctx* ctx, int n_windows, const lld_ba_window* windows,
                         const lld_ba_params* params, lld_ba_batch** out);
/* Error contract of a solve: when lld_ba_batch_solve returns anything but LLD_OK (a HIP call failed, a launch the build cannot express),
 * every stream the solve used has been drained before the call returns, and the batch is FAILED: its device state is somewhere inside an
 * LM trial, so solve / download / download_range / stats / result_records / phase_ms return LLD_ERR_INVALID from then on.  Destroy it and
 * create it again; the context stays usable. */
int  lld_ba_batch_solve(lld_ba_batch* batch, volatile const int* abort_flag); /* async on ctx stream until the final sync */
int  lld_ba_batch_download(lld_ba_batch* batch, int window, lld_ba_result* out);
/* Windows [first, first + count) into out[0..count): the same as `count` calls of lld_ba_batch_download, unpacked by several host
 * threads (a caller that wants every result of a 256-window batch moves 100 MB out of the landing buffer). */
int  lld_ba_batch_download_range(lld_ba_batch* batch, int first, int count, lld_ba_result* out /* [count] */);
int  lld_ba_batch_stats(lld_ba_batch* batch, lld_ba_stats* stats /* [n_windows] */);
/* Device buffer holding the fixed-stride result records of all windows (for the RCCL
 * gather): returns base pointer and record stride in bytes. */
int  lld_ba_batch_result_records(lld_ba_batch* batch, void** dev_ptr, uint64_t* stride_bytes);
/* Per-phase device time of the last solve in ms (HIP events on the stream the kernels are launched on):
 * [0] linearise (residuals + Jacobians + Hll/Hpl/Hpp), [1] Schur complement, [2] PCG on the reduced system,
 * [3] back-substitution + update + chi2, [4] LM control / outlier classification, [5] whole solve.
 * Mirrors G2OBatchStatistics (core/batch_stats.h:41-70), and like it ([0..4]) is OFF unless asked for: lld_ba_batch_set_phase_timing(batch, 1)
 * makes the following solves record an event at every phase boundary of every super-step.  An event between two dependent kernels costs
 * ~4 us of device time: 15 % of one window's solve, 10 % of a 32-window batch's, 1.6 % at 256 windows.  [5] is always measured. */
#define LLD_BA_N_PHASES 6
int  lld_ba_batch_set_phase_timing(lld_ba_batch* batch, int on);
int  lld_ba_batch_phase_ms(lld_ba_batch* batch, double* ms6);
/* Launch count (always) and summed HIP-event time (phase timing on) of one kernel family in the last solve; `kernel` uses the phase ids 0..4. */
int  lld_ba_batch_kernel_stats(lld_ba_batch* batch, int kernel, int64_t* launches, double* total_ms);
/* Tuning: number of window groups solved concurrently on separate HIP streams (1..8; 0 restores the default: one group below 8 windows,
 * three from 8, four from 16 - the device runs four streams side by side and time-slices a fifth -, and two for a batch of >= 64 windows
 * that was CREATED WHILE ANOTHER BATCH'S SOLVE RAN on the device: such a batch belongs to a pipelined caller whose other contexts' uploads
 * and downloads need streams of their own during its solve).  One group makes the HIP-event times of lld_ba_batch_phase_ms disjoint,
 * which is what a roofline measurement wants; several groups hide the latency-bound reduced solve and the per-super-step host poll.
 * Results do not depend on the grouping (bit-identical in the default deterministic mode). */
int  lld_ba_batch_set_groups(lld_ba_batch* batch, int n_groups);
void lld_ba_batch_destroy(lld_ba_batch* batch);

/* ------------------------------------------------------------------ the batch over several GPUs of one node
 * SURVEY.md 7 step 7 / 8e, north_star: independent windows shard across the GPUs of one node, the only exchange is the final gather of the
 * fixed-stride result records.  For a C++ host (LocalMapping / a relocalisation service keeps its threads): ONE process, one host thread
 * and one context per shard, created and driven by the library.  devices[] lists the HIP device of every shard; a device may appear
 * more than once (two shards share it).  Shard d owns the windows lld_ba_multi_shard(n_windows, n_devices, d) - the block partition of
 * lld_slam_amd/dist.py's shard(strong=True).  lld_ba_multi_solve runs every shard's lld_ba_batch_solve concurrently and, as each shard
 * finishes, copies its records into one buffer on devices[0] with hipMemcpyPeerAsync (a peer-to-peer copy over xGMI; a DEVIATION from north_star's
 * "RCCL for the final gather", chosen so that a torch process does not host a second RCCL instance - the torchrun path of bench.py uses RCCL itself): record k of the whole batch at k * stride.
 * lld_ba_multi_verify_gathered is the receiver's check that every record IS the window the partition put there (win_index and edge count
 * in the header) with a finished protocol.  Results per window are those of lld_ba_batch_* on that shard's batch, bit for bit. */
int  lld_device_count(void);                                          /* visible HIP devices; 0 without a GPU */
void lld_ba_multi_shard(int32_t n_windows, int32_t n_parts, int32_t part, int32_t* first, int32_t* count);   /* host only */
typedef struct lld_ba_multi lld_ba_multi;
int  lld_ba_multi_create(int32_t n_devices, const int32_t* devices, int32_t n_windows, const lld_ba_window* windows,
                         const lld_ba_params* params, lld_ba_multi** out);   /* n_windows >= n_devices */
int  lld_ba_multi_solve(lld_ba_multi* m, volatile const int* abort_flag);
int  lld_ba_multi_result_records(lld_ba_multi* m, void** dev_ptr, uint64_t* stride_bytes, int32_t* device);   /* the gathered buffer, on `device` = devices[0] */
int  lld_ba_multi_verify_gathered(lld_ba_multi* m, int32_t* n_checked);
int  lld_ba_multi_download(lld_ba_multi* m, int32_t window, lld_ba_result* out);   /* window of the whole batch, from the shard that solved it */
int  lld_ba_multi_times_ms(lld_ba_multi* m, double* slowest_solve_ms, double* slowest_gather_ms);   /* host clocks of the last lld_ba_multi_solve */
void lld_ba_multi_destroy(lld_ba_multi* m);

/* Diagnostic, host only (no device needed): the symbolic factorisation `reduced_solver = 0` runs per window - the stand-in for
 * LinearSolverEigen::computeSymbolicDecomposition (linear_solver_eigen.h:147-232).  block_nz[a * n_free_cams + b] != 0: free cameras a and
 * b share a landmark (symmetric; the diagonal is implied).  force: 0 = the plan a batch would use, 1 = the caller's camera order as one
 * chain of tile columns, 2 = the best two-chain (separator) plan only.  Writes the kernel's schedule (struct CholPlan of
 * lld_slam_amd/csrc/lld_ba_chol_plan.h, *plan_size bytes; its first byte is 1 when the structure-following kernel takes the window, 0
 * when it goes to the dense kernel) to plan_out if plan_bytes suffices.  tests/test_chol_plan.py executes such plans in numpy. */
int lld_ba_chol_plan(int32_t n_free_cams, const uint8_t* block_nz, int32_t force, void* plan_out, uint64_t plan_bytes, uint64_t* plan_size);

/* ================================================================== pose optimisation
 * Stands in for Optimizer::PoseOptimization (src/Optimizer.cc:653-932) including
 * AddLineMinOnlyPose (:562-650): 4 rounds x 10 LM iterations on one SE3 vertex. */
typedef struct {
  lld_camera cam;
  double pose_qt[7];                /* Converter::toSE3Quat(pFrame->mTcw)                       */
  int32_t n_points;                 /* matched keypoints with a MapPoint                        */
  const double*  pt_xw;             /* [n_points][3] world position widened from f32            */
  const double*  pt_uvr;            /* [n_points][3] u,v,uR (uR<0: mono)                        */
  const double*  pt_inv_sigma2;     /* [n_points]                                               */
  int32_t n_lines;
  const double*  ln_x0;             /* [n_lines][3]                                             */
  const double*  ln_dir;            /* [n_lines][3]                                             */
  const double*  ln_left;           /* [n_lines][4]                                             */
  const double*  ln_right;          /* [n_lines][4] xs<0 -> no stereo match                     */
  const int32_t* ln_octave;         /* [n_lines][2]                                             */
  const int32_t* ln_frame_index;    /* [n_lines] index i of the line in pFrame->mvLinesLeft (what AddLineMinOnlyPose pushes to
                                       vnIndexLines, Optimizer.cc:640); NULL = 0..n_lines-1 (every frame line has a MapLine).
                                       The reference classifies the edges of line i against the stereo or the mono threshold by
                                       vnStereoLines[i] (:898) although vnStereoLines is filled per EDGE (:643-648): the entry read
                                       is the stereo flag of the i-th edge added, whichever line that edge belongs to.  Reproduced
                                       as is; an index beyond the edge count (undefined behaviour there) counts as stereo.      */
} lld_pose_problem;

typedef struct {
  double  gamma;                    /* yaml `gamma` (0.5 for KITTI04-12_LBD.yaml:71)            */
  int32_t n_rounds;                 /* 4                                                         */
  int32_t its_per_round;            /* 10                                                        */
  int32_t max_trials;               /* 10                                                        */
  int32_t reserved;
} lld_pose_params;

void lld_pose_params_default(lld_pose_params* p);

typedef struct {
  double   pose_qt[7];
  int32_t  n_inliers;               /* return value of PoseOptimization (:931); 0 if <3 points  */
  int32_t  lm_iterations;           /* summed over rounds                                        */
  int32_t  lm_trials;
  int32_t  reserved;
  double   chi2;                    /* LM cost at the end of the last round                      */
  uint8_t* pt_outlier;              /* [n_points] pFrame->mvbOutlier                             */
  uint8_t* ln_outlier;              /* [n_lines]  pFrame->mvbOutlierLines                        */
} lld_pose_result;

int lld_pose_opt(lld_ctx* ctx, const lld_pose_problem* in, const lld_pose_params* params,
                 lld_pose_result* out);

/* Many frames in ONE launch (a relocalisation's candidates, a benchmark): one workgroup per frame.  A batch with more frames than the
 * device has compute units whose frames are small enough for two of them to share a CU's LDS (up to about 1100 points + 250 stereo
 * lines when the image observations are widened floats, as the reference's are) runs 256 lanes per frame and two frames per CU;
 * smaller batches and lld_pose_opt run 512 lanes per frame.  The two forms add a frame's edges in different (each fixed) orders: a
 * frame's result agrees between them to rounding (poses to 1e-9, identical inlier / outlier sets in every test and fuzz campaign),
 * and is bit-reproducible within a form - repeat solves, equal frames at other positions of a batch. */
typedef struct lld_pose_batch lld_pose_batch;
int  lld_pose_batch_create(lld_ctx* ctx, int n_frames, const lld_pose_problem* frames,
                           const lld_pose_params* params, lld_pose_batch** out);
int  lld_pose_batch_solve(lld_pose_batch* batch);
int  lld_pose_batch_download(lld_pose_batch* batch, int frame, lld_pose_result* out);
void lld_pose_batch_destroy(lld_pose_batch* batch);

/* ================================================================== Optimizer::OptimizeSim3 (src/Optimizer.cc:1656-1851)
 * One Sim3 vertex (S12, 7 dof, `_fix_scale` zeroes the scale update), the matched MapPoints of the two keyframes as FIXED points
 * in their own camera frames, two edges per correspondence (EdgeSim3ProjectXYZ: x1 = K1 proj(S12 X2); EdgeInverseSim3ProjectXYZ:
 * x2 = K2 proj(S12^-1 X1)) with Huber delta sqrt(th2), LM on the dense 7x7 system.  g2o differentiates these edges NUMERICALLY
 * (central differences, delta 1e-9, core/base_binary_edge.hpp:131-197): so do the oracle and the device.  Protocol: optimize(5);
 * a correspondence whose e12 or e21 chi2 exceeds th2 is dropped (both edges; vpMatches1[idx] = NULL); nMoreIterations = 10 if any
 * was dropped else 5; fewer than 10 correspondences left -> return 0 WITHOUT updating S12; optimize(nMoreIterations); count the
 * correspondences still within th2 (the others are NULLed too); S12 <- estimate.  `n`, the arrays and their order are the loop
 * :1704-1786 restricted to the correspondences that pass its tests (pMP1 && pMP2, neither bad, i2 >= 0). */
typedef struct {
  double fx1, fy1, cx1, cy1;        /* pKF1->mK (floats widened)                                   */
  double fx2, fy2, cx2, cy2;        /* pKF2->mK                                                    */
  double s12_q[4];                  /* g2oS12.rotation().coeffs(): x, y, z, w                      */
  double s12_t[3];
  double s12_s;
  int32_t n;
  int32_t reserved;
  const double* p1c;                /* [n][3] P3D1c = R1w*P3D1w + t1w (Converter::toVector3d)      */
  const double* p2c;                /* [n][3] P3D2c                                                */
  const double* obs1;               /* [n][2] pKF1->mvKeysUn[i].pt                                 */
  const double* obs2;               /* [n][2] pKF2->mvKeysUn[i2].pt                                */
  const double* inv_sigma2_1;       /* [n] pKF1->mvInvLevelSigma2[kpUn1.octave]                    */
  const double* inv_sigma2_2;       /* [n]                                                         */
} lld_sim3_problem;
typedef struct {
  double  th2;                      /* float th2 of the caller, widened (LoopClosing passes 10)    */
  int32_t fix_scale;                /* bFixScale (true for stereo / RGB-D)                         */
  int32_t its_first;                /* 5                                                           */
  int32_t its_more_bad;             /* 10 (when the first round dropped something)                 */
  int32_t its_more_clean;           /* 5                                                           */
  int32_t min_inliers;              /* 10                                                          */
  int32_t max_trials;               /* 10                                                          */
} lld_sim3_params;
void lld_sim3_params_default(lld_sim3_params* p);
typedef struct {
  double   s12_q[4], s12_t[3], s12_s;  /* g2oS12 on return (unchanged when the function returns 0 early) */
  uint8_t* dropped;                 /* [n] 1 -> vpMatches1[idx] = NULL                             */
  int32_t  n_inliers;               /* the return value (nIn, or 0)                                */
  int32_t  n_bad_first;             /* nBad of the first check                                     */
  int32_t  lm_iterations[2];
  int32_t  lm_trials[2];
  double   chi2;                    /* LM cost at the end of the last optimize()                   */
} lld_sim3_result;
int lld_optimize_sim3(lld_ctx* ctx, const lld_sim3_problem* in, const lld_sim3_params* params, lld_sim3_result* out);
/* several loop / relocalisation candidates in one launch (one workgroup each) */
int lld_optimize_sim3_batch(lld_ctx* ctx, int n, const lld_sim3_problem* problems, const lld_sim3_params* params, lld_sim3_result* outs);

/* ================================================================== Optimizer::OptimizeEssentialGraph (src/Optimizer.cc:1391-1654)
 * The pose graph itself: one Sim3 vertex per keyframe (Siw; `fixed[k]` for pLoopKF), one EdgeSim3 per loop / spanning-tree / old
 * loop / covisibility (>= 100) connection in the reference's insertion order, error = log(Sji * Siw * Sjw^-1)
 * (types_seven_dof_expmap.h:99-127), identity information, no robust kernel, NUMERIC Jacobians for both vertices like g2o
 * (core/base_binary_edge.hpp:131-197), Levenberg-Marquardt with setUserLambdaInit(1e-16), optimize(15).  Nothing is marginalised:
 * H is the 7N x 7N system, solved by the block-Jacobi PCG spread over the GPU (g2o: sparse Cholesky).
 * Building the edge list from the map (:1447-1585) and the write-back (:1593-1653: SE3 recovery [R t/s], MapPoint correction through
 * the reference keyframe) stay with the adapter. */
typedef struct {
  int32_t n_vertices;
  int32_t n_edges;
  const double*  sim3;          /* [n_vertices][8] Siw: rotation x, y, z, w, translation, scale      */
  const uint8_t* fixed;         /* [n_vertices] 1 = setFixed(true)                                    */
  const int32_t* edge_i;        /* [n_edges] vertex 0 of the edge (nIDi)                              */
  const int32_t* edge_j;        /* [n_edges] vertex 1 of the edge (nIDj)                              */
  const double*  edge_sji;      /* [n_edges][8] measurement Sji                                       */
} lld_pose_graph;
typedef struct {
  int32_t iterations;           /* 15                                                                 */
  int32_t fix_scale;            /* bFixScale                                                          */
  double  lambda_init;          /* 1e-16 (solver->setUserLambdaInit)                                  */
  int32_t max_trials;           /* 10                                                                 */
  int32_t pcg_max_iter;         /* 0 -> 10 * 7 * n_vertices                                           */
  double  pcg_rel_tol;          /* |r|_M / |b|_M of the PCG                                           */
  int32_t solver;               /* 0 auto (dense Cholesky up to 7*unknowns <= 32768, else PCG), 1 dense Cholesky, 2 PCG */
  int32_t reserved;
} lld_pose_graph_params;
void lld_pose_graph_params_default(lld_pose_graph_params* p);
typedef struct {
  double* sim3;                 /* [n_vertices][8] CorrectedSiw                                       */
  double  chi2;                 /* active chi2 after the last accepted step                           */
  int32_t lm_iterations, lm_trials, pcg_iterations, solver_used;   /* solver_used: 1 dense Cholesky, 2 PCG, 0 nothing to solve */
} lld_pose_graph_result;
int lld_optimize_essential_graph(lld_ctx* ctx, const lld_pose_graph* graph, const lld_pose_graph_params* params, lld_pose_graph_result* out);

/* ================================================================== descriptor matching
 * lld_match_hamming256*: ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1663) plus
 * the best / second-best loops of the Search* family (e.g. :76-125, :201-249).  Strict '<'
 * everywhere, so the FIRST candidate in iteration order wins ties.
 *   - brute force: candidates are all train rows in index order (mask NULL);
 *   - mask: byte matrix [nq][nt], non-zero = candidate (index order);
 *   - csr : explicit candidate lists in the reference's own list order
 *           (Frame::GetFeaturesInArea / BoW node lists); outputs are train indices.
 * Unmatched queries (no candidate) return idx -1 and dist 256 (the reference initialises
 * bestDist=256, ORBmatcher.cc:72).
 */
int lld_match_hamming256(lld_ctx* ctx, const uint32_t* q, int nq, const uint32_t* t, int nt,
                         const uint8_t* mask_or_null,
                         int32_t* best_idx, int32_t* best_dist,
                         int32_t* second_idx, int32_t* second_dist);
int lld_match_hamming256_csr(lld_ctx* ctx, const uint32_t* q, int nq, const uint32_t* t, int nt,
                             const int32_t* cand_start /*[nq+1]*/, const int32_t* cand_idx,
                             int32_t* best_idx, int32_t* best_dist,
                             int32_t* second_idx, int32_t* second_dist);
/* `batch` independent frame pairs of identical shape, inputs/outputs are DEVICE pointers:
 * q [batch][nq][8], t [batch][nt][8], outputs [batch][nq]. */
int lld_match_hamming256_batch_dev(lld_ctx* ctx, int batch, const uint32_t* q_dev, int nq,
                                   const uint32_t* t_dev, int nt,
                                   int32_t* best_idx_dev, int32_t* best_dist_dev,
                                   int32_t* second_idx_dev, int32_t* second_dist_dev);

/* lld_match_l2f32*: LineMatcher::MatchLineDescriptors call sites
 * (src/TwoFrameLineMatcher.cc:112, src/Tracking.cc:1092,1532).  The function itself lives in
 * the un-vendored LBDMOD library (parity unpinned); this build defines it as
 * d = sqrt( sum_i (double)(a_i - b_i)^2 ), the float difference squared and accumulated in
 * double in ascending i — the arithmetic of cv::norm(a - b) used at src/MapLine.cc:175. */
int lld_match_l2f32(lld_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim,
                    const uint8_t* mask_or_null,
                    int32_t* best_idx, double* best_dist,
                    int32_t* second_idx, double* second_dist);
int lld_match_l2f32_batch_dev(lld_ctx* ctx, int batch, const float* q_dev, int nq,
                              const float* t_dev, int nt, int dim,
                              int32_t* best_idx_dev, double* best_dist_dev,
                              int32_t* second_idx_dev, double* second_dist_dev);

/* TwoFrameLineMatcher::MatchLines (src/TwoFrameLineMatcher.cc:26-77): sequential greedy
 * assignment.  For left line j = 0..nq-1 in order: over right lines oi not yet taken and
 * with gate[j][oi] != 0 (CheckLinePair's geometric gates, :81-109, computed by the caller),
 * pick the strict running minimum of the descriptor distance below `tau`; the winner is
 * masked for all later j.  matches[j] = oi or -1. */
int lld_line_match_greedy(lld_ctx* ctx, const float* desc_left, int nq, const float* desc_right,
                          int nt, int dim, const uint8_t* gate /*[nq][nt]*/, double tau,
                          int32_t* matches /*[nq]*/, double* match_dist /*[nq] or NULL*/);

/* TwoFrameLineMatcher::MatchLines with CheckLinePair's geometric gates computed ON THE DEVICE
 * (src/TwoFrameLineMatcher.cc:26-124; the only caller is the left/right line association of the
 * Frame constructor, src/Frame.cc:121-122, so T = identity and T_right = GetTForRight(T, b),
 * src/LineMatching.cc:228-237).  For every (left j, right oi) pair the gate is
 *   same octave (:81-84)  &&  both lengths >= min_line_length (:86-91)
 *   && vgl::TriangulateLine succeeds (src/vgl.cc:78-108: back-projected plane normals
 *      n = R*GetNormalizedLineEq(kl,K) not closer than |cos| 0.975, direction n1 x n2, X0 from
 *      the 3x3 system [n1; n2; dir] X0 = [n1.t1; n2.t2; 0])  &&  |X0| >= 0.5 (:100-103)
 *   && both detected endpoints of the LEFT line, re-projected onto the 3D line by the 3x2 least
 *      squares of vgl::ReprojectLinePointTo3D (src/vgl.cc:336-346, src/LineMatching.cc:277-292),
 *      have z >= 0 (:104-109);
 * then the greedy, order-dependent descriptor assignment of lld_line_match_greedy.
 * lines: [n][4] float startPointX, startPointY, endPointX, endPointY of the KeyLines.
 * K: row-major 3x3 (Frame.cc:118-120).  b: mbf / fx.  gate_out: [nq][nt] bytes or NULL. */
typedef struct {
  double K[9];
  double b;
  double tau;               /* thrDD / mdThr */
  int32_t min_line_length;  /* minLineLen */
  int32_t is_stereo;        /* 1: octaves must be equal (:81) */
} lld_line_stereo_params;
int lld_line_match_stereo(lld_ctx* ctx, const lld_line_stereo_params* params,
                          const float* left_lines, const int32_t* left_octave, const float* desc_left, int nq,
                          const float* right_lines, const int32_t* right_octave, const float* desc_right, int nt,
                          int dim, int32_t* matches /*[nq]*/, double* match_dist /*[nq] or NULL*/,
                          uint8_t* gate_out /*[nq][nt] or NULL*/);

/* Tracking::AddLinesFrom (src/Tracking.cc:996-1124): the per-frame association of map lines (lines of the last frame /
 * of the local map) with the lines of the current frame, candidate selection and geometric gates ON THE DEVICE.
 * For map line i = 0..n_map-1 in order (skip[i] != 0: NULL, already tracked in this frame or bad, :1023-1034):
 *   candidates = SubselectWithGrid (src/LineMatching.cc:154-180): the projected line's Hough cell neighbourhood
 *     (GetHoughCoordinates, :63-152, 3 cells to each side in distance and angle) looked up in the frame's 50 x 50
 *     line grid, in ascending line index (the reference collects them in a std::set);
 *   a candidate si is dropped if it already holds a map line (:1054), has no stereo partner (:1059-1063, unless
 *     monocular), if one of the map line's main points X1, X2 lies behind the camera (:1066-1074), or if the L1
 *     reprojection error of its left / right detected endpoints against the projected 3D line
 *     (GetReprojErrPixelsL1 = vgl::LineReprojErrorL1, src/vgl.cc:548-559) exceeds thr_reproj_base * 1.44^octave
 *     (GetReprojThrPyramid, src/LineMatching.cc:239-247) in EITHER image (:1085);
 *   the strict running minimum of MatchLineDescriptors (float L2, see lld_match_l2f32) wins if it is <= md_thr
 *     (:1099) and then occupies its line for all later i (:1117).
 * The reference allocates the line grid (Frame.cc:746-755) but never fills it (SURVEY hazard 10), so its loop never
 * sees a candidate; this build DEFINES the fill it lacks: a frame line sits in the cell (dist_ind, ang_ind) that
 * GetHoughCoordinates computes for the image line through its left KeyLine's endpoints (GetLineEq, :255-268).
 * use_grid = 0 makes every line of the frame a candidate (brute force under the same gates).
 * T_curr: camera-to-world 4x4, row-major (the callers pass mTcw.inv()); the right camera is GetTForRight(T_curr, b).
 * map_x0 / map_dir: MapLine::GetMinimalPos; map_x1 / map_x2: GetMainPoints3D.  lines: [n][4] float startPointX,
 * startPointY, endPointX, endPointY.  line_matches[si]: index of the right line matched to left line si or -1.
 * matches[i] = frame line or -1; gate_out [n_map][n_cur] (optional): 1 where a pair passed every test but the
 * descriptor threshold. */
typedef struct {
  double K[9];
  double T_curr[16];
  double b;
  double thr_reproj_base;   /* thrReprojLineBase */
  double md_thr;            /* mdThr (KITTI04-12_LBD.yaml:70) */
  double sx, sy;            /* 1 / mnMaxX, 1 / mnMaxY */
  int32_t monocular;
  int32_t use_grid;
} lld_line_track_params;
int lld_line_track_match(lld_ctx* ctx, const lld_line_track_params* params,
                         int n_map, const double* map_x0, const double* map_dir, const double* map_x1, const double* map_x2,
                         const uint8_t* map_skip /*[n_map] or NULL*/, const float* map_desc,
                         int n_cur, const float* left_lines, const int32_t* left_octave,
                         int n_right, const float* right_lines, const int32_t* line_matches /*[n_cur]*/,
                         const uint8_t* occupied /*[n_cur] or NULL*/, const float* cur_desc, int dim,
                         int32_t* matches /*[n_map]*/, double* match_dist /*[n_map] or NULL*/,
                         uint8_t* gate_out /*[n_map][n_cur] or NULL*/);
/* Tracking::MatchLinesLastKF (src/Tracking.cc:1449-1611): new map lines from the lines the current and the last
 * stereo frame share.  For every line i of the current frame that holds no map line and has a stereo partner
 * (:1477-1487): vgl::TriangulateLine of its left / right KeyLine (src/vgl.cc:78-108) in the current pose; the Hough
 * cell neighbourhood of that 3D line projected into the LAST frame (SubselectWithGrid, :1504) gives the candidates
 * li; li is dropped without stereo partner (:1511-1515), if last_skip[li] (its map line was tracked in this frame,
 * :1517-1520) or if the L1 reprojection error exceeds 6 px * 1.44^octave in BOTH images of the last frame (:1526 -
 * `&&`, where AddLinesFrom has `||`); strict running minimum of MatchLineDescriptors, accepted if <= md_thr
 * (:1558); then vgl::MultiTriangulateLine over the four views (src/vgl.cc:28-76: unit plane normals, |cos| <= 0.975
 * against the first one, direction = right singular vector of the smallest singular value of the normal matrix,
 * X0 = least-squares point of the four planes minus its component along the direction), ReprojectKeyLineTo3D of
 * the current left KeyLine (src/LineMatching.cc:277-292) and the depth test of both end points in all four views
 * (:1580-1592).  No step depends on another line: rows run in parallel.
 * match_last[i]: the accepted line of the last frame or -1; created[i]: 1 if the reference would construct the
 * MapLine (x0 / dir [n_cur][3] valid).  The sign of dir is the build's (largest component positive): Eigen's JacobiSVD
 * is not restated and a 3D line has no orientation.  T_curr, T_last: camera-to-world, row-major 4x4. */
typedef struct {
  double K[9];
  double T_curr[16], T_last[16];
  double b;
  double thr_reproj_base;   /* 6 px (Tracking.cc:1451) */
  double md_thr;
  double sx, sy;
  int32_t use_grid;
  int32_t pad;
} lld_line_lastkf_params;
int lld_line_match_last_frame(lld_ctx* ctx, const lld_line_lastkf_params* params,
                              int n_cur, const float* cur_left, int n_cur_right, const float* cur_right,
                              const int32_t* cur_line_matches, const uint8_t* cur_occupied /*or NULL*/, const float* cur_desc,
                              int n_last, const float* last_left, const int32_t* last_left_octave, int n_last_right,
                              const float* last_right, const int32_t* last_line_matches, const uint8_t* last_skip /*or NULL*/,
                              const float* last_desc, int dim,
                              int32_t* match_last /*[n_cur]*/, uint8_t* created /*[n_cur]*/, double* x0 /*[n_cur][3]*/,
                              double* dir /*[n_cur][3]*/);
/* The grid cell of that fill: cell[si] = dist_ind * 50 + ang_ind (host helper, no device work). */
int lld_line_hough_cells(const float* lines, int n, double sx, double sy, int32_t* cell);

/* ================================================================== guided ORB search
 * lld_orb_search: the complete body of one ORBmatcher::Search* / Fuse / ComputeStereoMatches
 * routine for one (query set, keypoint set) pair: candidate generation ON THE DEVICE, the
 * per-candidate skip rules, DescriptorDistance, best / second-best, the accept rules, the
 * order-dependent "keypoint already taken" rule and the rotation-histogram filter
 * (SURVEY Appendix B).  The caller (the reference-side adapter) supplies what the reference
 * computes per query BEFORE its inner loop - projected position, window radius, predicted
 * level range, predicted right coordinate - and per keypoint the frame's own vectors.
 *
 * Candidate sets (lld_orb_search.candidates), always visited in the reference's order, which
 * decides ties (strict '<': the first candidate in order wins, unless tie_last):
 *   LLD_ORB_CAND_ALL   every keypoint, index order.
 *   LLD_ORB_CAND_GRID  Frame::GetFeaturesInArea (src/Frame.cc:391-444) / KeyFrame::
 *                      GetFeaturesInArea (src/KeyFrame.cc:592-631) over the 64x48 grid built
 *                      by Frame::AssignFeaturesToGrid + PosInGrid (:294-313, :446-456): cells
 *                      ix in [nMinCellX,nMaxCellX] outer, iy inner, keypoints of a cell in index
 *                      order; |dx|<r && |dy|<r; keypoints whose cell falls outside the grid are
 *                      never candidates.  All in float, as the reference.
 *   LLD_ORB_CAND_CSR   explicit lists: query q visits cand_idx[cand_range[q][0] .. cand_range[q][1]) in that order;
 *                      ranges may be shared between queries (all keypoints of one BoW node search the same
 *                      node list of the other frame: src/ORBmatcher.cc:185-187,561,700).
 *   LLD_ORB_CAND_ROWS  Frame::ComputeStereoMatches (src/Frame.cc:541-613): right keypoint iR is a
 *                      candidate of left keypoint iL iff (int)vL lies in [floor(yR-r), ceil(yR+r)],
 *                      r = 2*scale[octave_R]; index order; uR in [uL-disp_max, uL-disp_min];
 *                      a query with uL-disp_min < 0 is skipped (:577-578).
 * Gates (lld_orb_search.gates, OR of LLD_ORB_GATE_*), each skips a candidate:
 *   LEVEL    octave < q_level_min || (q_level_max >= 0 && octave > q_level_max)
 *   STEREO   t_uright > 0 && fabs(q_uright - t_uright) > q_stereo_radius   (ORBmatcher.cc:90-95,1400-1406)
 *   CHI2     Fuse's reprojection gate (ORBmatcher.cc:912-936): stereo keypoints (t_uright>=0)
 *            e2*invSigma2[octave] > 7.8, others > 5.99, e = (q_uv - t_xy [, q_uright - t_uright])
 *   EPIPOLAR SearchForTriangulation (ORBmatcher.cc:720-751): only_stereo filter; epipole distance
 *            when neither side is stereo; CheckDistEpipolarLine (:138-157) with the query's line
 *   t_occupied[k] != 0 always skips; with `sequential` a keypoint taken by an accepted EARLIER
 *   query whose q_blocks flag is set is skipped too (the reference writes mvpMapPoints /
 *   vpMatched / vbMatched2 inside its loop).  The device reaches the sequential answer by
 *   fixed-point rounds over all queries and reports the number of rounds.
 *   sequential = 2 is the rule of ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520), the one routine whose order
 *   dependence is not an occupancy: a candidate is skipped when an EARLIER query holds it with a distance <= this one
 *   (`vMatchedDistance[i2]<=dist`, :443), an accepted query takes the keypoint away from its holder, whose match is cleared
 *   (:458-465), and the rotation histogram counts every acceptance, also those taken away later (:466-476).  match[] is
 *   vnMatches12 before the orientation filter, owner[] vnMatches21, `rounds` 1 + the number of queries whose cached candidate
 *   list ran dry; GRID or ALL candidates, ratio_mode 0 or 1, no tie_last.
 * Accept: best <= accept_max, then ratio_mode 0 none | 1 (float)best < nn*(float)second
 * (:226-228) | 2 reject iff level(best)==level(second) && best > nn*second (:118-121).
 * check_orientation: 30-bin histogram of q_angle - t_angle, factor 1/30, ComputeThreeMaxima
 * (:1601-1642); matches outside the three kept bins are flagged `removed`.
 * Outputs (host arrays): match[nq] accepted keypoint or -1 (before the orientation filter);
 * removed[nq]; best/second distances (256 = none); owner[nt] = the query that holds keypoint k at
 * the end of the routine (last accepted writer; -1 if none or if any writer was removed by the
 * orientation filter - the reference NULLs the slot, :1452-1460); n_matches = the routine's
 * return value. */
enum { LLD_ORB_CAND_ALL = 0, LLD_ORB_CAND_GRID = 1, LLD_ORB_CAND_CSR = 2, LLD_ORB_CAND_ROWS = 3 };
enum { LLD_ORB_GATE_LEVEL = 1, LLD_ORB_GATE_STEREO = 2, LLD_ORB_GATE_CHI2 = 4, LLD_ORB_GATE_EPIPOLAR = 8 };
#define LLD_ORB_MAX_KEYPOINTS 4096
#define LLD_ORB_MAX_LEVELS 16

typedef struct {
  /* keypoints searched (Frame / KeyFrame; the right image for ROWS) */
  int32_t nt;
  const uint32_t* t_desc;       /* [nt][8]                                                       */
  const float*    t_xy;         /* [nt][2] mvKeysUn[k].pt (ROWS: mvKeysRight[k].pt)              */
  const int32_t*  t_octave;     /* [nt]                                                          */
  const float*    t_uright;     /* [nt] mvuRight; NULL = all -1                                  */
  const float*    t_angle;      /* [nt] degrees; needed with check_orientation                   */
  const uint8_t*  t_occupied;   /* [nt] or NULL                                                  */
  /* queries, in the order the reference's outer loop visits them */
  int32_t nq;
  const uint32_t* q_desc;       /* [nq][8]                                                       */
  const uint8_t*  q_valid;      /* [nq] or NULL; 0 = the reference `continue`s before the search */
  const uint8_t*  q_blocks;     /* [nq] or NULL (= all 1); see `sequential`                      */
  const float*    q_uv;         /* [nq][2] GRID: window centre; ROWS: left keypoint; CHI2: projection */
  const float*    q_radius;     /* [nq]    GRID                                                  */
  const int32_t*  q_level_min;  /* [nq]    LEVEL                                                 */
  const int32_t*  q_level_max;  /* [nq]    LEVEL                                                 */
  const float*    q_uright;     /* [nq]    STEREO / CHI2                                         */
  const float*    q_stereo_radius; /* [nq] STEREO                                                */
  const float*    q_angle;      /* [nq]    check_orientation                                     */
  const float*    q_epiline;    /* [nq][3] EPIPOLAR: a,b,c of x1'F12 (ORBmatcher.cc:141-143)     */
  const uint8_t*  q_stereo;     /* [nq]    EPIPOLAR: bStereo1                                    */
  const int32_t*  cand_range;   /* [nq][2] CSR: begin, end into cand_idx                         */
  const int32_t*  cand_idx;     /* [n_cand] CSR                                                  */
  int32_t         n_cand;
  /* frame constants */
  float grid_min_x, grid_min_y, grid_width_inv, grid_height_inv;   /* mnMinX, mnMinY, mfGridElement*Inv */
  int32_t grid_cols, grid_rows;                                    /* 64, 48 (Frame.h:43-44)     */
  int32_t n_levels;
  const float* level_scale;       /* [n_levels] mvScaleFactors     (ROWS, EPIPOLAR)              */
  const float* level_sigma2;      /* [n_levels] mvLevelSigma2      (EPIPOLAR)                    */
  const float* level_inv_sigma2;  /* [n_levels] mvInvLevelSigma2   (CHI2)                        */
  float disp_min, disp_max;       /* ROWS: minD, maxD (Frame.cc:558-560)                         */
  float epipole_x, epipole_y;     /* EPIPOLAR (ORBmatcher.cc:669-671)                            */
  int32_t only_stereo;            /* EPIPOLAR: bOnlyStereo                                       */
  /* rules */
  int32_t candidates;             /* LLD_ORB_CAND_*                                              */
  int32_t gates;                  /* OR of LLD_ORB_GATE_*                                        */
  int32_t tie_last;               /* 1: `dist>bestDist -> skip`, later equal distance replaces (ORBmatcher.cc:733) */
  int32_t accept_max;             /* accept iff best <= accept_max                               */
  int32_t ratio_mode;
  float   nnratio;                /* mfNNratio                                                   */
  int32_t sequential;             /* 0 | 1 occupancy by earlier queries | 2 SearchForInitialization's take-over rule        */
  int32_t check_orientation;
} lld_orb_search;

typedef struct {
  int32_t* match;        /* [nq] */
  int32_t* best_dist;    /* [nq] */
  int32_t* second_dist;  /* [nq] */
  uint8_t* removed;      /* [nq] */
  int32_t* owner;        /* [nt] or NULL */
  int32_t  n_matches;
  int32_t  rounds;       /* fixed-point rounds executed (1 without `sequential`) */
} lld_orb_search_result;

int lld_orb_search_run(lld_ctx* ctx, const lld_orb_search* s, lld_orb_search_result* out);
/* Tracking::SearchLocalPoints (src/Tracking.cc:1613-1664) in one call: Frame::isInFrustum (src/Frame.cc:333-389) for every
 * local MapPoint ON THE DEVICE - camera transform, projection, image bounds, scale-invariance distance band, viewing angle,
 * MapPoint::PredictScale (src/MapPoint.cc:402-417), mTrackProjXR - and then ORBmatcher::SearchByProjection(F, vpMapPoints, th)
 * (src/ORBmatcher.cc:45-129) on the projected points without a trip through the host.
 * The float / double mixture follows the reference's OpenCV calls as this build restates them (OpenCV is not in the image:
 * parity unpinned): `mRcw*P+mtcw` is one cv::gemm, i.e. float(sum_k double(R_ik) double(P_k) + double(t_i)); `P-mOw` a float
 * subtraction; cv::norm and Mat::dot accumulate in double; everything else is float arithmetic in source order; log() of a
 * float is the float overload - computed the way glibc (>= 2.27) computes logf, so that PredictScale's ceil() flips at the same
 * float as on the reference's host (lld_orb_search.hip glibc_logf; a libm with another logf moves one level in ~1e8).
 * `frame`: only the keypoint side (nt, t_*), the grid constants and n_levels / level_scale are read.
 * frustum outputs (any may be NULL): in_view[n] (mbTrackInView), proj_uvr[n][3] (mTrackProjX, mTrackProjY, mTrackProjXR),
 * level[n] (mnTrackScaleLevel), view_cos[n] (mTrackViewCos); values of points outside the frustum are unspecified. */
typedef struct {
  float Rcw[9], tcw[3], Ow[3];          /* Frame::mRcw, mtcw, mOw (UpdatePoseMatrices, src/Frame.cc:325-331) */
  float fx, fy, cx, cy, bf;             /* mbf = baseline * fx */
  float min_x, max_x, min_y, max_y;     /* mnMinX ... */
  float log_scale_factor;               /* mfLogScaleFactor */
  int32_t n_levels;                     /* mnScaleLevels */
} lld_frame_view;
typedef struct {
  int32_t n;
  const float*    world_pos;            /* [n][3] MapPoint::GetWorldPos */
  const float*    normal;               /* [n][3] GetNormal */
  const float*    max_distance;         /* [n] mfMaxDistance (GetMaxDistanceInvariance = 1.2f * it) */
  const float*    min_distance;         /* [n] mfMinDistance (0.8f * it) */
  const uint32_t* desc;                 /* [n][8] GetDescriptor */
  const uint8_t*  has_obs;              /* [n] Observations()>0, or NULL = all 1 */
  const uint8_t*  skip;                 /* [n] or NULL: mnLastFrameSeen == frame id || isBad (src/Tracking.cc:1640-1643) */
} lld_map_points;
typedef struct { uint8_t* in_view; float* proj_uvr; int32_t* level; float* view_cos; } lld_frustum_result;
int lld_orb_search_local_points(lld_ctx* ctx, const lld_orb_search* frame, const lld_frame_view* view, const lld_map_points* points,
                                float viewing_cos_limit, float th, float nnratio,
                                lld_frustum_result* frustum_or_null, lld_orb_search_result* out);
/* ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono) (src/ORBmatcher.cc:1328-1470, the matcher of
 * Tracking::TrackWithMotionModel) in one call: the projection of the last frame's MapPoints into the current frame (:1358-1377:
 * cv::gemm transform, invzc = float(1.0 / double(z)), image bounds), the window radius th*scale[octave], the octave range chosen by
 * `direction` (+1 bForward, -1 bBackward, 0 neither; the caller evaluates :1343-1350 once per frame pair), ur = u - mbf*invzc, then
 * the search with occupancy and the rotation histogram.  last_valid[i] = LastFrame.mvpMapPoints[i] && !LastFrame.mvbOutlier[i].
 * proj_uvr (may be NULL): [n][3] u, v, ur of the projected points. */
typedef struct {
  int32_t n;
  const float*    world_pos;    /* [n][3] pMP->GetWorldPos() of LastFrame.mvpMapPoints[i] */
  const uint8_t*  valid;        /* [n] */
  const int32_t*  octave;       /* [n] LastFrame.mvKeys[i].octave */
  const float*    angle;        /* [n] LastFrame.mvKeysUn[i].angle */
  const uint32_t* desc;         /* [n][8] pMP->GetDescriptor() */
  const uint8_t*  has_obs;      /* [n] Observations()>0, or NULL = all 1 */
} lld_last_frame_points;
int lld_orb_search_last_frame(lld_ctx* ctx, const lld_orb_search* frame, const lld_frame_view* view, const lld_last_frame_points* last,
                              int direction, float th, int check_orientation, float* proj_uvr_or_null, lld_orb_search_result* out);
/* A frame RESIDENT on the device for the time the Tracking thread works on it (round 5).  The reference runs, on one Frame, the matcher of
 * TrackWithMotionModel (src/Tracking.cc:904), PoseOptimization (:937), SearchLocalPoints (:1133), PoseOptimization (:1152); the two
 * matchers above re-send the frame's 2000 keypoints (descriptors, positions, octaves, right coordinates, angles: 106 KB through pinned
 * memory) on every call.  lld_frame_create uploads the keypoint side of `keypoints` (nt, t_desc, t_xy, t_octave, t_uright, t_angle, the grid
 * constants and level tables; the query side is ignored) ONCE; lld_frame_search_last_frame / lld_frame_search_local_points are
 * lld_orb_search_last_frame / lld_orb_search_local_points on that frame - same arguments, same results bit for bit - and move only their
 * queries and the per-call occupancy bytes (t_occupied[nt] or NULL: Frame::mvpMapPoints[k] != NULL at the time of the call).  The handle
 * belongs to the context (and host thread) it was created on; destroy it before the context. */
typedef struct lld_frame lld_frame;
int  lld_frame_create(lld_ctx* ctx, const lld_orb_search* keypoints, lld_frame** out);
int  lld_frame_search_last_frame(lld_frame* frame, const uint8_t* t_occupied, const lld_frame_view* view, const lld_last_frame_points* last,
                                 int direction, float th, int check_orientation, float* proj_uvr_or_null, lld_orb_search_result* out);
int  lld_frame_search_local_points(lld_frame* frame, const uint8_t* t_occupied, const lld_frame_view* view, const lld_map_points* points,
                                   float viewing_cos_limit, float th, float nnratio, lld_frustum_result* frustum_or_null, lld_orb_search_result* out);
void lld_frame_destroy(lld_frame* frame);
/* ------------------------------------------------------------------ the Tracking thread's per-frame chain, device resident (round 6)
 * On one stereo Frame the reference runs
 *   TrackWithMotionModel (src/Tracking.cc:885-994): SearchByProjection(Current, Last, th) - again with 2*th when it finds fewer than 20
 *     (:904-911) -> AddLinesFrom(mLastFrame.mvpMapLines) (:924) -> Optimizer::PoseOptimization (:937) -> outlier discard (:940-975);
 *   TrackLocalMap (:1126-1220): SearchLocalPoints (:1133, :1613-1664) -> AddLinesFrom(local_lines) (:1140) -> PoseOptimization (:1152)
 *     -> statistics / discard (:1155-1187),
 * and between those calls the Frame carries mvpMapPoints, mvbOutlier, mvpMapLines, mvbOutlierLines and mTcw.  lld_frame_track_* keeps
 * exactly that state in HBM next to the resident keypoints: every stage reads what the stage before it left on the device, the edges of
 * PoseOptimization are gathered from the match tables by a kernel (what Optimizer.cc:683-804 does from mvpMapPoints / mvpMapLines), and
 * NOTHING travels to the host between the stages.  Both calls only upload their map-side inputs (one copy), queue their kernels on the
 * context's stream and return; lld_frame_track_download fetches both stages' records in one copy and is the only synchronisation.
 * (The reference's host needs the MapPoints of stage 1 for UpdateLocalMap before it can name the local map of stage 2: such a caller
 * downloads between the two calls - 10 KB - and still has no search -> PoseOptimization hand-over through the host.)
 *
 * MapPoints / MapLines are named by caller-chosen ids >= 0 (MapPoint::mnId / MapLine ids, or indices into the caller's own tables):
 *   - "pMP->mnLastFrameSeen == mCurrentFrame.mnId" (:1640), which makes SearchLocalPoints skip the points the frame already holds (:1629)
 *     AND those the outlier discard of stage 1 marked (:949), is an id look-up against the frame's held + discarded ids;
 *   - "pML->tracked_last_id == mCurrentFrame.mnId" (:1023) likewise against the lines stage 1 assigned (kept even when PoseOptimization
 *     then threw the line out, as in the reference, which never resets tracked_last_id).
 * Stale state the reference keeps is kept: mvbOutlierLines is not cleared when an outlier line leaves the frame (:962-975), so a line
 * edge that never reaches a classification (fewer than 10 edges, Optimizer.cc:878) reads the old flag.
 * Deviations (both inside the "OpenCV restated" caveat of the searches above): the camera-to-world matrix AddLinesFrom receives is the
 * frame's own Rwc = Rcw^T, Ow (Frame::UpdatePoseMatrices) widened to double, where the reference inverts mTcw with cv::Mat::inv() (a float
 * LU, equal up to float rounding); stage 2's float view is formed on the device from the optimised SE3Quat exactly as
 * lld_se3_to_tcw_f32 + UpdatePoseMatrices form it on the host.
 * `params` of the two calls of one frame must agree.  The handle belongs to one context / host thread like every lld_frame call. */
typedef struct {
  int32_t n_left;  const float* left;  const int32_t* left_octave;    /* mvLinesLeft: [n][4] startPointX, startPointY, endPointX, endPointY; octave */
  int32_t n_right; const float* right; const int32_t* right_octave;   /* mvLinesRight                                                              */
  const int32_t* line_matches;                                         /* [n_left] Frame::line_matches: right line of left line i, or -1            */
  const float* desc; int32_t dim;                                      /* mDescriptorsLines [n_left][dim], dim <= 128                                */
  int32_t reserved;
  double sx, sy;                                                       /* 1 / mnMaxX, 1 / mnMaxY (the Hough grid of lld_line_track_match)            */
} lld_frame_lines;
/* Uploads the frame's lines ONCE (and fills the 50 x 50 Hough grid cells on the device); NULL or n_left = 0: a frame without lines.  Resets the
 * tracking state of the frame.  Call it after lld_frame_create and before lld_frame_track_motion_model; synchronous. */
int  lld_frame_set_lines(lld_frame* frame, const lld_frame_lines* lines);
typedef struct {
  int32_t n;
  const double* x0; const double* dir;                                 /* [n][3] MapLine::GetMinimalPos                                              */
  const double* x1; const double* x2;                                  /* [n][3] GetMainPoints3D                                                     */
  const uint8_t* skip;                                                 /* [n] or NULL: NULL entry / isBad (src/Tracking.cc:1018-1034)                */
  const float* desc;                                                   /* [n][dim] descs[i] / mLastFrame.mDescriptorsLines.row(i)                    */
  const int32_t* id;                                                   /* [n] >= 0                                                                   */
} lld_map_lines;
typedef struct {
  lld_camera cam;                   /* PoseOptimization's intrinsics (fx, fy, cx, cy, bf as doubles of the Frame's floats)                           */
  lld_pose_params pose;             /* gamma, 4 x 10 iterations                                                                                      */
  float   th_motion;                /* 7 (stereo) / 15 (src/Tracking.cc:899-903)                                                                     */
  float   th_local;                 /* 1; 3 RGBD; 5 after a relocalisation (:1652-1658)                                                              */
  float   nnratio_local;            /* 0.8 (:1651)                                                                                                   */
  float   viewing_cos_limit;        /* 0.5 (:1646)                                                                                                   */
  int32_t direction;                /* lld_orb_search_last_frame's: +1 bForward, -1 bBackward, 0 neither                                             */
  int32_t check_orientation;        /* 1: ORBmatcher(0.9, true) (:888)                                                                               */
  int32_t wide_retry;               /* 1: search again with 2 * th_motion when the first search finds < 20 (:907-911), decided on the device          */
  int32_t monocular;                /* 0 (the chain is the stereo system's)                                                                          */
  double  line_thr_reproj_base;     /* 2 (:924, :1140)                                                                                               */
  double  line_md_thr;              /* mdThr                                                                                                         */
  int32_t line_use_grid;            /* as lld_line_track_params.use_grid                                                                             */
  int32_t reserved;
} lld_track_params;
void lld_track_params_default(lld_track_params* p);
typedef struct {
  double  pose_qt[7];               /* mTcw after the stage's PoseOptimization (pFrame->SetPose, Optimizer.cc:918)                                   */
  double  chi2;
  int32_t n_inliers;                /* PoseOptimization's return value                                                                               */
  int32_t lm_iterations, lm_trials, n_edges;
  int32_t n_search_first;           /* nmatches of the (first) search                                                                                */
  int32_t n_search;                 /* nmatches of the search whose matches the frame took (stage 1: the wide one if used_wide)                      */
  int32_t used_wide;
  int32_t n_points;                 /* MapPoints the frame holds after the stage's discard (stage 1: `nmatches`, :952)                               */
  int32_t n_points_map;             /* ... of which Observations() > 0 (stage 1: nmatchesMap, :955; stage 2: mnMatchesInliers, :1164)                */
  int32_t n_lines_matched;          /* MapLines the frame held when PoseOptimization started (lcnt_init / lcnt, :926-932, :1142-1149)                */
  int32_t n_lines;                  /* ... and after the outlier lines left (:962-975, :1176-1187)                                                   */
  int32_t n_discarded;
  int32_t n_point_edges;            /* nInitialCorrespondences of the stage's PoseOptimization (below 3 it returned without touching the pose, Optimizer.cc:809) */
  int32_t n_in_view;                /* stage 2: nToMatch of SearchLocalPoints (local MapPoints inside the frustum, src/Tracking.cc:1645-1649)              */
  /* caller-allocated, any may be NULL.  Ids / flags as the stage's PoseOptimization saw them, BEFORE its discard:                                   */
  int32_t* kp_point_id;             /* [nt] id of mvpMapPoints[k] or -1                                                                              */
  uint8_t* kp_outlier;              /* [nt] mvbOutlier[k] of those (the discard removes exactly the flagged ones)                                    */
  int32_t* ln_line_id;              /* [n_left] id of mvpMapLines[i] or -1                                                                           */
  uint8_t* ln_outlier;              /* [n_left] mvbOutlierLines[i] of those                                                                          */
  uint8_t* mp_in_view;              /* stage 2 only, [local_points->n] or NULL: Frame::isInFrustum of every local MapPoint that was not skipped (what the
                                       reference leaves in pMP->mbTrackInView and counts with IncreaseVisible, src/Tracking.cc:1645-1649)            */
} lld_track_result;
/* Stage 1.  `view`: Frame::UpdatePoseMatrices of the predicted pose mVelocity * mLastFrame.mTcw (as for lld_frame_search_last_frame);
 * `pose_qt`: Converter::toSE3Quat of the same matrix (lld_se3_from_tcw_f32).  last / last_point_id: LastFrame.mvpMapPoints as for
 * lld_frame_search_last_frame plus the id of every entry; last_lines: mLastFrame.mvpMapLines (NULL: none). */
int  lld_frame_track_motion_model(lld_frame* frame, const lld_track_params* params, const lld_frame_view* view, const double* pose_qt,
                                  const lld_last_frame_points* last, const int32_t* last_point_id, const lld_map_lines* last_lines);
/* Stage 1 ran elsewhere: Tracking::TrackReferenceKeyFrame (src/Tracking.cc:770-816) or Tracking::Relocalization end with the same
 * PoseOptimization + outlier discard but find their matches by bag of words / PnP.  This call hands the device what such a routine left in
 * the frame, so that lld_frame_track_local_map can follow on the same handle (Tracking::Track runs TrackLocalMap after whichever routine
 * produced the pose, :401-407).  The record of stage 1 reads as empty afterwards. */
typedef struct {
  const int32_t* kp_point_id;       /* [nt] id of mvpMapPoints[k] or -1                                                                              */
  const float*   kp_world_pos;      /* [nt][3] GetWorldPos() of those (ignored where the id is -1)                                                   */
  const uint8_t* kp_has_obs;        /* [nt] Observations() > 0, or NULL: all                                                                         */
  const uint8_t* kp_outlier;        /* [nt] mvbOutlier, or NULL: none                                                                                */
  int32_t        n_seen;            /* MapPoints with mnLastFrameSeen == mCurrentFrame.mnId the frame does not hold (the discard's, :805-808); <= nt */
  const int32_t* seen_point_id;
  const int32_t* ln_line_id;        /* [n_left] id of mvpMapLines[i] or -1; NULL: the frame holds no lines (TrackReferenceKeyFrame adds none)        */
  const double*  ln_x0;             /* [n_left][3] GetMinimalPos of those                                                                            */
  const double*  ln_dir;
  const uint8_t* ln_outlier;        /* [n_left] mvbOutlierLines, or NULL: none                                                                       */
  int32_t        n_tracked;         /* further MapLines with tracked_last_id == mCurrentFrame.mnId (<= n_left + 16 together with the held ones)      */
  const int32_t* tracked_line_id;
} lld_frame_held;
/* view / pose_qt: Frame::UpdatePoseMatrices and Converter::toSE3Quat of the frame's mTcw, as for lld_frame_track_motion_model. */
int  lld_frame_track_set_state(lld_frame* frame, const lld_track_params* params, const lld_frame_view* view, const double* pose_qt, const lld_frame_held* held);
/* Stage 2, on the pose and the MapPoints / MapLines stage 1 left in the frame.  local_points->skip: isBad only - what the frame holds or
 * discarded is skipped by id on the device.  local_lines: Tracking::local_lines with their descriptors (NULL: none). */
int  lld_frame_track_local_map(lld_frame* frame, const lld_track_params* params, const lld_map_points* local_points, const int32_t* local_point_id,
                               const lld_map_lines* local_lines);
/* Waits for the queued stages and fetches their records (either may be NULL; stage2 is meaningful only after lld_frame_track_local_map). */
int  lld_frame_track_download(lld_frame* frame, lld_track_result* stage1, lld_track_result* stage2);
/* ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:825-958; the loop of LocalMapping::SearchInNeighbors) with the
 * projection loop (:841-890) on the device: cv::gemm transform, z >= 0, invz = 1/z, u = fx*(x*invz)+cx, KeyFrame::IsInImage
 * (upper bounds strict, src/KeyFrame.cc:633-636), ur = u - bf*invz, scale-invariance band, PO.dot(Pn) >= 0.5*dist3D, PredictScale;
 * then the window search with the level and reprojection-chi2 gates and `bestDist <= TH_LOW`.  `points`: as for
 * lld_orb_search_local_points; skip[i] = !pMP || isBad || IsInKeyFrame(pKF); has_obs is ignored (no occupancy in Fuse).
 * out->match[i] = bestIdx or -1, out->n_matches = nFused; the replace / add bookkeeping (:936-954) stays with the caller. */
int lld_orb_fuse_search(lld_ctx* ctx, const lld_orb_search* keyframe, const lld_frame_view* view, const lld_map_points* points,
                        float th, float* proj_uvr_or_null, lld_orb_search_result* out);
/* The matchers of relocalisation and loop closing WITH their projection loops on the device (round 3; before, the searches were
 * mirrored and the per-query projections stayed with the adapter).  One entry point, four routines:
 *   LLD_ORB_PROJ_KF_SIM3    ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)      src/ORBmatcher.cc:290-403
 *                           (LoopClosing::ComputeSim3): z >= 0, invz = 1/z, u = fx*(x*invz)+cx, KeyFrame::IsInImage, distance band,
 *                           PO.dot(Pn) >= 0.5*dist, PredictScale, radius = th*scale[level], levels [l-1, l], keypoints with
 *                           vpMatched[idx] skipped (frame->t_occupied) and taken ones blocking later points, bestDist <= TH_LOW.
 *   LLD_ORB_PROJ_RELOC      ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)  :1472-1599
 *                           (Tracking::Relocalization): NO depth test, invzc = float(1.0/z), u = fx*xc*invzc+cx, frame bounds
 *                           (u<min || u>max), distance band, PredictScale, radius = th*scale, levels [l-1, l+1], occupied keypoints
 *                           (CurrentFrame.mvpMapPoints[i2], frame->t_occupied) skipped and blocking, bestDist <= accept_max (ORBdist),
 *                           rotation histogram over `angle` (pKF->mvKeysUn[i].angle) when check_orientation.
 *   LLD_ORB_PROJ_FUSE_SIM3  ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)                :977-1100
 *                           (LoopClosing::SearchAndFuse): projection as KF_SIM3, levels [l-1, l], no occupancy, bestDist <= TH_LOW;
 *                           match[i] = bestIdx, the replace / add bookkeeping (:1078-1093) stays with the caller.
 *   LLD_ORB_PROJ_SIM3_DIR   one direction of ORBmatcher::SearchBySim3                                     :1147-1224 / :1227-1304
 *                           p3Dc1 = R1w*p3Dw+t1w (view), p3Dc2 = sR*p3Dc1+t (second cv::gemm, `sR`, `t` below), z >= 0, IsInImage,
 *                           dist3D = cv::norm(p3Dc2), band, PredictScale, levels [l-1, l], bestDist <= TH_HIGH.  view->fx.. are pKF1's in
 *                           both directions (:1105-1108), the bounds and the scale pyramid those of the keyframe searched in.
 * `view` carries the DECOMPOSED transform (Rcw = sRcw/scw, tcw = Scw.col(3)/scw, Ow = -Rcw.t()*tcw, :298-303 - three OpenCV calls
 * that stay with the adapter) and the intrinsics / image bounds of the keyframe or frame searched in; `points`: as for
 * lld_orb_search_local_points (normal may be NULL for RELOC and SIM3_DIR; skip[i] = isBad / already found / vbAlreadyMatched;
 * has_obs is ignored).  proj_uv [n][2] and level [n] (either may be NULL) return u, v and nPredictedLevel of the points that reach
 * the window search.  Same OpenCV restatement as above (parity unpinned), bit-identical between device and oracle. */
#define LLD_ORB_PROJ_KF_SIM3   0
#define LLD_ORB_PROJ_RELOC     1
#define LLD_ORB_PROJ_FUSE_SIM3 2
#define LLD_ORB_PROJ_SIM3_DIR  3
typedef struct {
  int32_t routine;
  float   th;
  int32_t accept_max;         /* RELOC: ORBdist; ignored by the other routines (TH_LOW / TH_HIGH as listed) */
  int32_t check_orientation;  /* RELOC only */
  float   sR[9], t[3];        /* SIM3_DIR: sR21, t21 (KF1 -> KF2) or sR12, t12 (KF2 -> KF1), row-major */
} lld_orb_projection;
int lld_orb_search_projected(lld_ctx* ctx, const lld_orb_search* frame, const lld_frame_view* view, const lld_map_points* points,
                             const float* angle_or_null, const lld_orb_projection* proj, float* proj_uv_or_null, int32_t* level_or_null,
                             lld_orb_search_result* out);
/* ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1102-1326) as a whole: both directions (two LLD_ORB_PROJ_SIM3_DIR searches) and the
 * agreement check (:1306-1322).  kf1 / view1 / points1: KF1's keypoints, its pose (R1w, t1w) and its MapPoints per keypoint
 * (skip[i] = !pMP || vbAlreadyMatched1[i] || isBad), likewise KF2; sR12, t12, sR21, t21 as the reference forms them (:1121-1124).
 * match12[i1] = index of the KF2 keypoint whose MapPoint becomes vpMatches12[i1], or -1; returns nFound in *n_found. */
int lld_orb_search_by_sim3(lld_ctx* ctx, const lld_orb_search* kf1, const lld_frame_view* view1, const lld_map_points* points1,
                           const lld_orb_search* kf2, const lld_frame_view* view2, const lld_map_points* points2,
                           const float* sR12, const float* t12, const float* sR21, const float* t21, float th,
                           int32_t* match12 /* [points1->n] */, int32_t* n_found);
/* ------------------------------------------------------------------ Frame::ComputeStereoMatches, whole routine
 * src/Frame.cc:530-704: (1) the row-band Hamming search (:536-613, the ROWS problem of lld_orb_search_run with the level gate
 * octave +-1, disparity range [0, mbf/mb] and bestDist < (TH_HIGH+TH_LOW)/2), (2) the sub-pixel refinement (:615-688): 11x11
 * patch of the left pyramid level around the rounded scaled keypoint, centre pixel subtracted, L1 distance to the right patch slid
 * over incR = -5..5, first minimum (int bestDist, strict <), parabola through the three distances around it, |deltaR| <= 1,
 * bestuR = scale * (scaleduR0 + bestincR + deltaR), 0 <= disparity < mbf/mb (disparity <= 0 -> 0.01), (3) the outlier cut
 * (:690-703): median of the SAD distances, entries with dist >= 1.5f*1.4f*median are cleared.
 * All of it is integer / single-rounding float work: results are bit-exact against the CPU restatement.
 * Deviation: the reference slices cv::Mat ranges unchecked (OpenCV aborts when a patch leaves the image; ORB keeps keypoints
 * 19 px inside); here such a keypoint simply gets no stereo match. */
typedef struct {
  int32_t n;
  const float*    xy;           /* [n][2] mvKeys / mvKeysRight .pt */
  const int32_t*  octave;       /* [n]                              */
  const uint32_t* desc;         /* [n][8]                           */
} lld_keypoints;
typedef struct {
  int32_t n_levels;
  const uint8_t* const* left;   /* [n_levels] mpORBextractorLeft->mvImagePyramid[l].data (CV_8U)  */
  const uint8_t* const* right;  /* [n_levels] mpORBextractorRight->mvImagePyramid[l].data         */
  const int32_t* cols;          /* [n_levels]                                                       */
  const int32_t* rows;          /* [n_levels]                                                       */
  const int32_t* left_step;     /* [n_levels] bytes per image row (cv::Mat::step)                   */
  const int32_t* right_step;
  const float*   scale_factors;     /* [n_levels] mvScaleFactors    */
  const float*   inv_scale_factors; /* [n_levels] mvInvScaleFactors */
  int32_t on_device;            /* 1: the image pointers are HBM pointers (e.g. of a device ORB extractor), nothing is uploaded */
  int32_t reserved;
} lld_stereo_pyramids;
typedef struct {
  float*   u_right;             /* [n_left] mvuRight (-1 = none)                                   */
  float*   depth;               /* [n_left] mvDepth  (-1 = none)                                   */
  int32_t* best_r;              /* [n_left] or NULL: bestIdxR of the Hamming stage, -1 = none      */
  int32_t* sad;                 /* [n_left] or NULL: bestDist of the refinement as pushed into vDistIdx, -1 = not pushed */
  int32_t  n_matches;           /* entries of vDistIdx that survive the median cut                 */
  int32_t  reserved;
} lld_stereo_result;
int lld_compute_stereo_matches(lld_ctx* ctx, const lld_keypoints* left, const lld_keypoints* right, const lld_stereo_pyramids* pyr,
                               float mb, float mbf, lld_stereo_result* out);
/* `n` independent problems (e.g. one relocalisation / loop candidate keyframe each, or the searches of several frames) in one
 * launch: one workgroup per problem, all inputs moved in one host-to-device copy and all outputs in one copy back. */
int lld_orb_search_batch(lld_ctx* ctx, int n, const lld_orb_search* problems, lld_orb_search_result* outs);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* LLD_AMD_H */
