// lld_ba_kernels.h — device side of the batched local bundle adjustment (gfx950).
//
// One launch of each kernel sweeps ALL windows of a batch (grid.y = window); every window carries its own
// Levenberg–Marquardt state machine in HBM (BAState) and each kernel starts by reading it, so windows advance
// independently ("super-steps"): a window that rejected its trial only re-runs Schur/PCG/back-substitution with the
// new lambda, a window that accepted re-linearises, finished windows fall through.  The host only launches and reads
// back three counters per super-step.
//
// Kernel <-> reference map
//   ba_linearize   computeActiveErrors + activeRobustChi2 + buildSystem   sparse_optimizer.cpp:61-114, block_solver.hpp:502-560,
//                  (per-edge residual, Huber weight, closed-form blocks:  base_binary_edge.hpp:54-120, types_six_dof_expmap.cpp
//                   Hll/bl by a segmented shuffle sum, the weight of a
//                   point edge / the Hpl block of a line observation,
//                   Hpp/bp through LDS-staged per-camera accumulators)
//   ba_hpp_reduce  camera-block sums; its last workgroup per window runs the  optimization_algorithm_levenberg.cpp:75-99,166-180
//                  LM iteration head (chi2, lambda init)
//   ba_schur       setLambda + Schur complement: Z = W L^-T per (landmark,    block_solver.hpp:373-439,564-589
//                  slot) through LDS, whole 6x6 products Z_a Z_b^T in
//                  registers per chunk, fixed-order reduction into S
//   ba_chol_mfma   reduced camera system solve (exact Cholesky on the      linear_solver_eigen.h:94-124 (sparse LDLT there)
//   ba_chol/ba_pcg fp64 matrix cores; vector-ALU Cholesky and block-
//                  Jacobi PCG as alternatives) + camera oplus
//   ba_backsub     landmark back-substitution, oplus, trial chi2           block_solver.hpp:459-483, sparse_optimizer.cpp:422-435
//   ba_control     accept / reject, lambda update, stop rules              optimization_algorithm_levenberg.cpp:102-164
//   ba_classify    outlier levels between the two rounds                   Optimizer.cc:1239-1267, LineOptimizer.cc:129-170
//   ba_finalize    erase lists + read-back                                 Optimizer.cc:1278-1329, LineOptimizer.cc:172-201
#ifndef LLD_BA_KERNELS_H
#define LLD_BA_KERNELS_H

#include "lld_common.h"
#include "lld_device_math.h"
#include "lld_ba_chol_plan.h"

namespace lldba {

using namespace lld;

constexpr int kLmThreads = 256;        // landmark-parallel kernels: one lane per landmark
constexpr int kSchurThreads = 64;     // Schur kernel: one wavefront per landmark chunk
constexpr int kSchurWideThreads = 256; // ... and its variant for landmarks with more than 64 free observations
constexpr int kPcgThreads = 1024;
constexpr int kCtlThreads = 64;
constexpr int kLinThreads = 512;        // linearise kernels: 8 tasks per workgroup (fewer per-workgroup Hpp partials to reduce)
// Tasks swept per wavefront of a linearise / back-substitution workgroup (BAWin::rounds, chosen per batch by the host):
// sweeping several tasks amortises zeroing / flushing the LDS accumulators, the pose copies and x_p, and divides the number of
// per-workgroup Hpp partials (and ba_hpp_reduce's work) by the same factor - worth it when many windows fill the chip (throughput),
// not when a handful of windows need every workgroup they can get (latency).  Order: linearise pt, linearise ln, backsub pt, backsub ln.
constexpr int kRoundsThroughput[4] = {8, 2, 4, 1};
// (fewer than kRoundsThroughputMinWindows windows.  Until round 4 this was {1, 1, 1, 1}: one task per wavefront made eight times the workgroups,
// each zeroing and flushing its accumulator copies and leaving a partial for ba_hpp_reduce - swept again with the fused small-group kernels
// (tools/exp_small_rounds.sh, gpurun_out/s4_small_rounds*.txt): 16 windows 1420 -> 1740 windows/s, 8: 870 -> 940, 4: 454 -> 490, 1: 179 -> 183)
constexpr int kRoundsLatency[4] = {4, 2, 4, 1};
constexpr int kRoundsThroughputMinWindows = 32;
// batches that fill the GPU many times over: twice / four times the tasks per workgroup (fewer, longer workgroups, fewer per-workgroup partials to
// reduce).  Swept at the end of round 3 (LLD_BA_ROUNDS): 256 LBA-B windows 5320 -> 5430 windows/s; 128 windows and fewer, and the smaller LBA-A
// windows, are 1 - 5 % faster with the setting above.
constexpr int kRoundsThroughputBig[4] = {16, 4, 16, 1};
constexpr int kRoundsBigMinWindows = 192;
constexpr int kAccCopies = 4;          // (default; BAWin::acc_copies drops to 2 or 1 when a window's cameras would not fit LDS otherwise)
constexpr int kAccCopiesDoc = 4;          // LDS copies of the per-camera Hpp/bp accumulators: lanes of one wavefront that hit the
                                       // same camera are spread over them (same-address LDS atomics serialise)
constexpr int kMaxFreeCams = 8192;       // (S is dense: 6 n_free squared doubles); up to 590 the linearise accumulators (216 B + 56 B per camera) live in LDS,
constexpr int kMaxFreeCamsLds = 590;     // beyond that in HBM (BAWin::big); the one-workgroup reduced solvers
constexpr int kMaxFreeCamsOneWg = 170;   // map one lane to one unknown (6 * 170 <= 1024), larger windows need the multi-workgroup PCG

enum Phase : int { PH_RUN = 0, PH_TRANSITION = 2, PH_FINALIZE = 3, PH_DONE = 4 };
constexpr uint8_t EF_LEVEL1 = 1, EF_ROBUST = 2, EF_VALID = 4, EF_PAIRSTEREO = 8, EF_STEREO = 16;

struct BAWin {                 // immutable per-window header
  CamK cam;
  int n_cams, n_free;
  int cam_off;                 // cameras
  int pt_off, n_pt;            // point landmarks
  int ln_off, n_ln;            // line landmarks
  int pe_off, n_pe;            // point edges
  int le_off, n_le;            // line edge slots (2 per observation)
  int hpp_off;                 // free-camera accumulators
  int x_off;                   // reduced-system vectors (doubles)
  long long S_off;             // reduced-system matrix (doubles)
  int item_off, n_items, n_items_pt;   // Schur chunks of this window (range in sg_chunks): n_items_pt point chunks, then line chunks
  int lo_off, n_lo;            // line observations (= le_off / 2)
  int blk_csr_off, cam_csr_off; // CSR (per lower S block / per free camera) of the chunk partials that add into it
  int n_blk_nz;                // the first n_blk_nz blocks of blk_perm are structurally non-zero (diagonal, or some chunk adds to them)
  int nb_pt, nb_ln;            // landmark blocks (kLmThreads landmarks each)
  int ptask_off, n_ptasks, nt_pt;   // point tasks (4 * rounds[2] per back-substitution workgroup -> nt_pt workgroups)
  int ltask_off, n_ltasks, nt_ln;   // line tasks; partial-sum slots of a window: nt_pt + nt_ln
  int nl_pt, nl_ln;            // workgroups of the linearise kernels (rounds[0|1] * kLinThreads / 64 tasks each)
  int rounds[4];               // tasks per wavefront: linearise pt / ln, backsub pt / ln
  long long hpart_off;         // per-workgroup Hpp/bp partials of the linearise kernels (doubles): [nl_pt + nl_ln][n_free * 27]
  int part_off;                // per-block partial sums
  int its[2];                  // LM iterations per round
  int max_trials, ln_filter;
  int abort_after;             // lld_ba_params::abort_after_trials (test hook: the stop flag counts as raised once this many LM trials are done; 0 = off)
  int big;                     // more cameras than the LDS of the linearise / back-substitution kernels holds: accumulators and poses in HBM
  int protocol, robust_pts;
  int acc_copies[2];           // LDS copies of the per-camera accumulators in the point / line linearise kernel (4, 2 or 1; bit-reproducible mode: = lin_waves)
  int det, lin_waves[2];       // bit-reproducible mode (lld_ba_params::deterministic resolved); wavefronts per point / line linearise workgroup
  int win_index;               // index of the window in its batch (slot of the multi-workgroup PCG scalars)    // lld_ba_params::protocol / robust_points (1 = global BA: one round, no classification)
  double th_mono, th_stereo;   // Huber deltas of point edges  ((double)(float)sqrt(5.991 / 7.815))
  double th_ln_mono, th_ln_stereo;   // Huber deltas of line edges (x gamma)
  long long rec_off;           // result record (bytes)
};

struct BAState {               // mutable per-window LM state
  int phase, need_lin, round, it, q, cur, nBad, pcg_ok;
  double lambda, ni, currentChi, iniChi, scale_cam;
  unsigned long long maxdiag_bits;
  double chi2_round1, chi2_final;
  int lm_iterations[2], lm_trials[2];
  int pcg_iterations, aborted, n_active_edges, pad;
  int ticket_lin, ticket_bs, ticket_cls, pad2;   // "last workgroup of the window" tickets of ba_hpp_reduce / ba_backsub_ctl / ba_classify (zero between launches)
};

// Schur work decomposition (built once per window on the host from the camera sets of the landmarks):
// landmarks that are seen by the SAME set of free cameras are sorted together and cut into chunks; one wavefront owns a
// chunk and accumulates -Y_a W_b^T for every camera-slot pair over the chunk's landmarks in registers before it touches S.
// Lane-per-edge point kernels: one wavefront per task = a run of consecutive point landmarks whose edges fit in 64 lanes
// (or a single landmark with any number of edges).
struct PTask { int l0, nl, e0, ne, ms, pad0, pad1, pad2; };   // local first landmark, landmark count, global first edge / observation,
                                                             // edge count, longest run of one landmark (bounds the segmented reductions)

struct SChunk { int lm_off, n_lm, tab_off, cams_off, k, D, part_off, cpart_off; };   // part_off: 36-double blocks, cpart_off: 6-double vectors

struct BAArrays {
  long long NC, NP, NL;        // totals (stride of the double-buffered state arrays)
  // state, double buffered: [2][N]
  double* cam_qt;              // [2][NC*7]
  double *ptx, *pty, *ptz;     // [2][NP]
  double *lqx, *lqy, *lqz, *lqw, *lal;   // [2][NL]
  // inputs
  const double* cam_qt0;       // [NC*7]
  const double* pt0;           // [NP*3]
  const double *ln_x0, *ln_dir;          // [NL*3]
  const int* pt_obs_start;     // [NP+1] global point-edge index
  const int* ln_obs_start;     // [NL+1] global line-observation index (slots = 2*obs + side)
  const int* pe_pt;            // [NPE] window-local landmark of a point edge (the kernels that sweep edges without a task: classify, finalize)
  // Observations, one of two layouts (`packed`, chosen per batch by the host):
  // packed = 1 - every image coordinate / information value the caller handed over is a float widened to double (what the reference's
  //   cv::KeyPoint::pt, mvuRight, mvInvLevelSigma2 and KeyLine end points are) and every line octave is in 0 .. 254: 16-byte float
  //   records, camera and landmark slot in one word, the line information by octave from a 256-entry table.  A point edge is 20 B
  //   where it was 40, a line observation 42 B where it was 114 - the back-substitution kernels run at 70 % of the copy bandwidth
  //   and the upload of a 256-window batch halves.  Widening a float is exact: the arithmetic sees the same doubles.
  // packed = 0 - anything else (the ABI takes doubles): the arrays as the caller gave them.
  int packed;
  // Grid rows of the per-super-step kernels map to windows through slot_map (set per launch by the host, offset to the launch's GROUP of
  // windows: slot_map[y] = the group-relative window grid row y works on, -1: none) or directly (null: row y = window y).  With the map a
  // super-step is launched over the windows that are still at work - the host knows their number from the previous poll - instead of
  // over the whole group: in the last third of a batch's solve only the windows with rejected trials are left, and a launch over 64
  // windows of which 3 are running spends its time dispatching workgroups that look at their window's phase and leave (a tail
  // super-step of a 64-window group took 700 us where one window alone takes 250).  The LM control rebuilds the map after every
  // super-step, in window order, from the "still running / in transition" bits its wavefronts publish in active_pub (agent-scope stores,
  // read by the group's last control wavefront of the same launch).
  int* slot_map;               // the buffer the LM control WRITES (the next super-step's rows); double buffered by the host, never the one slot_rd reads in the same launch
  int* active_pub;
  const int* slot_rd;          // what the kernels READ rows through: slot_map, or null (row y = window y) while every window of the group is still at work - the map is
                               // the identity then, and a launch spares its workgroups one dependent load (4 us of a single window's 225 us super-step)
  const float4* pe_obs;        // [NPE] u, v, uR (< 0: monocular), invSigma2
  const int* pe_cs;            // [NPE] camera | (landmark - first landmark of the edge's task) << 24
  const float4* lo_seg;        // [2 NLO] slot 2 o + side: startPointX, startPointY, endPointX, endPointY
  const int* lo_cs;            // [NLO] camera | (line - first line of the observation's task) << 24
  const int* lo_ln;            // [NLO] window-local line (classify, finalize)
  const unsigned short* lo_oct;// [NLO] left octave | right octave << 8; 255: no such edge
  const double* ln_info;       // [256] information of a line edge by octave byte (lld::line_info; [255] = 0)
  const int* pe_cam;
  const double *pe_u, *pe_v, *pe_ur, *pe_s;
  const int* le_cam; const int* le_ln;                     // [2 NLO] per edge slot
  const double *le_xs, *le_ys, *le_xe, *le_ye, *le_s;      // (b_x of a slot is 0 for the left and CamK::bx_right for the right image; its validity follows from startPointX of the right segment)
  // per-edge mutable
  uint8_t *pe_flags, *le_flags;
  double *pe_chi2, *le_chi2;
  double *pe_ws;               // [NPE] rho' * invSigma2 of the edge at the linearisation point (0 for level-1 edges), NEGATED for a stereo edge (the weight is >= 0: its sign
                               //       bit carries the one flag the Schur staging needs, which spares it a gather of the flag byte - round 5): the 6x3 Hpl
                               //       block of a point edge is recomputed from it (point_hpl) instead of being stored
  double *lo_W;                // [NLO*24]  Hpl block 6x4 per (line, KF) observation: left + right edge summed
  // per-landmark mutable
  uint8_t *pt_active, *ln_active, *ln_removed;
  double *pt_V;                // [NP*9]   Hll upper (6) + bl (3)
  double *ln_V;                // [NL*14]  Hll upper (10) + bl (4)
  // per-window reduced system
  double *hpp_part;            // per linearise workgroup: [n_free][21 + 6]
  double *Hpp;                 // [NF*21]
  double *bp;                  // [NF*6]
  double *S, *bschur, *xp;
  double *pcg_vec;             // multi-workgroup PCG: r, z, p, Sp of every window ([4][x_total])
  double *pcg_mi;              // its block-Jacobi preconditioner ([NF][36])
  double *pcg_sc;              // its scalars per window: rz, stop, iterations, done, ok, 3 spare
  long long x_total;
  double *chi_part, *chi_part2, *scale_part;
  // Schur work items
  const SChunk* sg_chunks;
  const PTask* ptasks; const PTask* ltasks;
  double *sp_part, *sp_cpart;  // per (chunk, slot pair) 6x6 partial products / per (chunk, slot) 6-vectors
  const int *blk_start, *blk_src, *cam_start, *cam_src;
  int s_skip_empty;            // the batch's solvers only read S: ba_schur_reduce leaves the structurally empty blocks at the zeros of the batch's memset
  const int *blk_perm;         // per window (at blk_csr_off): the lower blocks of S ordered by the length of their partial lists (ba_schur_reduce)
  const int *sg_lm, *sg_tab, *sg_cams;
  const CholPlan* chol_plan;   // [windows of the batch] schedule of the structure-following reduced solve (mode 0: the dense kernel's window); null: none
  // results
  unsigned char* records;
#ifdef LLD_EXPERIMENTS
  long long* chol_stamps;      // experiments build: [window][kCholStampWaves][kCholStampSlots] s_memtime stamps of ba_chol_mfma_kernel's stages
#endif
};
constexpr int kCholStampSlots = 256;
constexpr int kCholStampWaves = 16;      // wavefront rows per window in BAArrays::chol_stamps (the dense kernel has 8 wavefronts, the structure-following one 12)
#ifdef LLD_EXPERIMENTS
#define LLD_CHOL_STAMP(k) do { if (stamp_base && lane == 0) stamp_base[(k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define LLD_CHOL_STAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------------ small helpers
// Values one workgroup hands to ANOTHER workgroup of the same launch (the "last workgroup of the window" fusions: ba_hpp_reduce ->
// LM head, ba_backsub_ctl -> LM control).  The 8 XCDs of the part have private L2s: a plain store may sit dirty in the writer's L2 and a
// plain load may hit a stale line in the reader's.  An agent-scope fence per workgroup would fix that by writing back / invalidating
// the WHOLE L2 - measured: batches of 32 windows ran at half speed with one __threadfence() per back-substitution workgroup - so the
// handful of exchanged values travel with agent-scope atomic stores / loads instead (write-through / bypass), and the writer waits
// for its stores to be acknowledged before it takes its ticket.
__device__ __forceinline__ void xwg_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double xwg_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xwg_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void xwg_store_i32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xwg_load_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The window grid row `y` of a per-super-step kernel works on (see BAArrays::slot_map); < 0: none, the workgroup leaves.
#define LLD_ROW_WINDOW(A, st, y) ((A).slot_rd ? (A).slot_rd[(y)] : (int)(y))
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
  return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off));
  return x;
}
// Fixed-tree block sum; result valid in every lane.  `scratch` holds blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double x, double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  x = wave_sum(x);
  __syncthreads();
  if (lane == 0) scratch[wave] = x;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; i++) t += scratch[i];
  return t;
}
__device__ __forceinline__ double block_max(double x, double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  x = wave_max(x);
  __syncthreads();
  if (lane == 0) scratch[wave] = x;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; i++) t = fmax(t, scratch[i]);
  return t;
}

// Inverse of a symmetric positive definite D x D matrix (full row-major in/out) through Cholesky; false if not SPD.
template <int D>
__device__ __forceinline__ bool spd_inverse(const double* A, double* Ainv) {
  double L[D][D];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < D; j++) {
    double d = A[j * D + j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
    if (!(d > 0.0)) ok = false;
    const double ljj = sqrt(d);
    L[j][j] = ljj;
    const double inv = 1.0 / ljj;
#pragma unroll
    for (int i = j + 1; i < D; i++) {
      double s = A[i * D + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      L[i][j] = s * inv;
    }
  }
  // Linv (lower)
  double Li[D][D];
#pragma unroll
  for (int j = 0; j < D; j++) {
    Li[j][j] = 1.0 / L[j][j];
#pragma unroll
    for (int i = j + 1; i < D; i++) {
      double s = 0.0;
#pragma unroll
      for (int k = j; k < i; k++) s -= L[i][k] * Li[k][j];
      Li[i][j] = s / L[i][i];
    }
  }
  // Ainv = Linv^T Linv
#pragma unroll
  for (int i = 0; i < D; i++)
#pragma unroll
    for (int j = i; j < D; j++) {
      double s = 0.0;
#pragma unroll
      for (int k = j; k < D; k++) s += Li[k][i] * Li[k][j];
      Ainv[i * D + j] = s; Ainv[j * D + i] = s;
    }
  return ok;
}

// (rcp_nr / rsqrt_nr: lld_device_math.h)
// lower Cholesky factor of (packed upper U) + lambda I, D x D: L packed row-major lower (L[i][j] at i(i+1)/2 + j, diagonal entries
// unused), idiag[i] = 1 / L[i][i]
template <int D>
__device__ __forceinline__ void chol_packed(const double* U, double lambda, double* L, double* idiag) {
  double F[D][D];
  int kk = 0;
#pragma unroll
  for (int i = 0; i < D; i++)
#pragma unroll
    for (int j = i; j < D; j++) { F[j][i] = U[kk++]; }
#pragma unroll
  for (int i = 0; i < D; i++) F[i][i] += lambda;
#pragma unroll
  for (int j = 0; j < D; j++) {
    double d = F[j][j];
#pragma unroll
    for (int m = 0; m < j; m++) d -= L[j * (j + 1) / 2 + m] * L[j * (j + 1) / 2 + m];
    const double inv = rsqrt_nr(d);
    idiag[j] = inv;
#pragma unroll
    for (int i = j + 1; i < D; i++) {
      double sacc = F[i][j];
#pragma unroll
      for (int m = 0; m < j; m++) sacc -= L[i * (i + 1) / 2 + m] * L[j * (j + 1) / 2 + m];
      L[i * (i + 1) / 2 + j] = sacc * inv;
    }
  }
}
// x = (U + lambda I)^-1 t through that factor (forward, then backward substitution)
template <int D>
__device__ __forceinline__ void chol_solve(const double* U, double lambda, const double* t, double* x) {
  double L[D * (D + 1) / 2], idg[D], y[D];
  chol_packed<D>(U, lambda, L, idg);
#pragma unroll
  for (int c = 0; c < D; c++) {
    double sacc = t[c];
#pragma unroll
    for (int m = 0; m < c; m++) sacc -= y[m] * L[c * (c + 1) / 2 + m];
    y[c] = sacc * idg[c];
  }
#pragma unroll
  for (int c = D - 1; c >= 0; c--) {
    double sacc = y[c];
#pragma unroll
    for (int m = c + 1; m < D; m++) sacc -= x[m] * L[m * (m + 1) / 2 + c];
    x[c] = sacc * idg[c];
  }
}

__device__ __forceinline__ Pose load_cam(const BAArrays& A, int buf, int cam_global) {
  return pose_load(A.cam_qt + ((size_t)buf * A.NC + cam_global) * 7);
}
// Point positions: SoA x, y, z [2][NP].  (Round 5 measured 32-byte records x, y, z, active - one sector for the Schur staging's gather where
// the three arrays are three: ba_schur 16.65 -> 16.43 ms per step, but the landmark lanes of the point kernels then read 24 of every 32
// bytes: linearise 13.74 -> 14.14, back-substitution 9.45 -> 9.75, value - 1.1 % on the same box.  Dropped; the stereo flag in the sign of
// pe_ws, which spares the Schur staging a gather and costs nobody, stayed.)
__device__ __forceinline__ Vec3 load_pt(const BAArrays& A, int buf, int g) {
  const size_t o = (size_t)buf * A.NP + g;
  return vec3(A.ptx[o], A.pty[o], A.ptz[o]);
}
__device__ __forceinline__ void store_pt(const BAArrays& A, int buf, int g, const Vec3& X) {
  const size_t o = (size_t)buf * A.NP + g;
  A.ptx[o] = X.x; A.pty[o] = X.y; A.ptz[o] = X.z;
}
__device__ __forceinline__ LineQ load_ln(const BAArrays& A, int buf, int g) {
  const size_t o = (size_t)buf * A.NL + g;
  LineQ l; l.q.x = A.lqx[o]; l.q.y = A.lqy[o]; l.q.z = A.lqz[o]; l.q.w = A.lqw[o]; l.alpha = A.lal[o];
  return l;
}
__device__ __forceinline__ void store_ln(const BAArrays& A, int buf, int g, const LineQ& l) {
  const size_t o = (size_t)buf * A.NL + g;
  A.lqx[o] = l.q.x; A.lqy[o] = l.q.y; A.lqz[o] = l.q.z; A.lqw[o] = l.q.w; A.lal[o] = l.alpha;
}

// ---- observations: the two layouts of BAArrays behind one set of accessors.  kPk: 1 = packed, 0 = as given (the hot kernels exist in
// both forms and the host launches the one that matches BAArrays::packed), 2 = ask BAArrays::packed at run time (a scalar branch; the
// once-per-solve kernels and the maps whose accumulators live in HBM).  Not a run-time branch in the hot kernels: a load under a branch
// cannot be counted, the compiler waits for it right behind the branch (s_waitcnt vmcnt(0)), and the handful of operand loads of a task
// became as many dependent round trips - the linearisation of 256 windows ran 12 % SLOWER on half the bytes.
constexpr int kPkRuntime = 2;
template <int kPk> __device__ __forceinline__ bool obs_packed(const BAArrays& A) { return kPk == kPkRuntime ? A.packed != 0 : kPk == 1; }
struct PtObs { double u, v, ur, s; };
template <int kPk>
__device__ __forceinline__ PtObs pt_obs_of(const BAArrays& A, int e) {
  PtObs ob;
  if (obs_packed<kPk>(A)) { const float4 o = A.pe_obs[e]; ob.u = (double)o.x; ob.v = (double)o.y; ob.ur = (double)o.z; ob.s = (double)o.w; }
  else { ob.u = A.pe_u[e]; ob.v = A.pe_v[e]; ob.ur = A.pe_ur[e]; ob.s = A.pe_s[e]; }
  return ob;
}
template <int kPk> __device__ __forceinline__ bool pt_obs_stereo(const BAArrays& A, int e) { return obs_packed<kPk>(A) ? !(A.pe_obs[e].z < 0.f) : !(A.pe_ur[e] < 0); }
template <int kPk> __device__ __forceinline__ int pt_cam_of(const BAArrays& A, int e) { return obs_packed<kPk>(A) ? (A.pe_cs[e] & 0xffffff) : A.pe_cam[e]; }
// camera and window-local landmark of edge e of a task whose first landmark is l0 (no load of pe_pt in the packed layout)
template <int kPk>
__device__ __forceinline__ void pt_cam_lm_of(const BAArrays& A, int e, int l0, int& c, int& l) {
  if (obs_packed<kPk>(A)) { const unsigned w = (unsigned)A.pe_cs[e]; c = (int)(w & 0xffffffu); l = l0 + (int)(w >> 24); }
  else { c = A.pe_cam[e]; l = A.pe_pt[e]; }
}
// line observation o (both image edges) / edge slot e = 2 o + side
template <int kPk>
__device__ __forceinline__ void ln_cam_lm_of(const BAArrays& A, int o, int l0, int& c, int& l) {
  if (obs_packed<kPk>(A)) { const unsigned w = (unsigned)A.lo_cs[o]; c = (int)(w & 0xffffffu); l = l0 + (int)(w >> 24); }
  else { c = A.le_cam[2 * o]; l = A.le_ln[2 * o]; }
}
template <int kPk> __device__ __forceinline__ int ln_cam_of(const BAArrays& A, int o) { return obs_packed<kPk>(A) ? (A.lo_cs[o] & 0xffffff) : A.le_cam[2 * o]; }
template <int kPk> __device__ __forceinline__ int ln_line_of(const BAArrays& A, int o) { return obs_packed<kPk>(A) ? A.lo_ln[o] : A.le_ln[2 * o]; }
struct LnSeg { double xs, ys, xe, ye; };
template <int kPk>
__device__ __forceinline__ LnSeg ln_seg_of(const BAArrays& A, int e) {
  LnSeg g;
  if (obs_packed<kPk>(A)) { const float4 q = A.lo_seg[e]; g.xs = (double)q.x; g.ys = (double)q.y; g.xe = (double)q.z; g.ye = (double)q.w; }
  else { g.xs = A.le_xs[e]; g.ys = A.le_ys[e]; g.xe = A.le_xe[e]; g.ye = A.le_ye[e]; }
  return g;
}
template <int kPk>
__device__ __forceinline__ double ln_info_of(const BAArrays& A, int e) {
  return obs_packed<kPk>(A) ? A.ln_info[(A.lo_oct[e >> 1] >> (8 * (e & 1))) & 255] : A.le_s[e];
}
// the flags an edge slot starts with: valid (the left slot always, the right one if the observation has a right segment) | stereo pair
template <int kPk>
__device__ __forceinline__ uint8_t ln_flags0_of(const BAArrays& A, int e) {
  const int right = e | 1;
  const bool has_right = obs_packed<kPk>(A) ? !(A.lo_seg[right].x < 0.f) : !(A.le_xs[right] < 0);      // startPointX >= 0 (LineOptimizer.cc:60)
  const bool valid = !(e & 1) || has_right;
  return (uint8_t)((valid ? 4 : 0) | (has_right ? 8 : 0));                                    // EF_VALID | EF_PAIRSTEREO
}

__device__ __forceinline__ double chi2_of(const double* e, int D, double s) {
  double c = e[0] * (s * e[0]) + e[1] * (s * e[1]);
  if (D == 3) c += e[2] * (s * e[2]);
  return c;
}

// ================================================================== init
// grid (blocks, nW): resets the working state of every window from the uploaded inputs.
__global__ __launch_bounds__(kLmThreads) void ba_init_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  const BAWin W = wins[blockIdx.y];
  const int gid = blockIdx.x * kLmThreads + threadIdx.x, stride = gridDim.x * kLmThreads;
  for (int c = gid; c < W.n_cams * 7; c += stride) {
    const double v = A.cam_qt0[(size_t)W.cam_off * 7 + c];
    A.cam_qt[(size_t)W.cam_off * 7 + c] = v; A.cam_qt[(size_t)(A.NC + W.cam_off) * 7 + c] = v;
  }
  for (int p = gid; p < W.n_pt; p += stride) {
    const int g = W.pt_off + p;
    const Vec3 X = vec3(A.pt0[(size_t)g * 3], A.pt0[(size_t)g * 3 + 1], A.pt0[(size_t)g * 3 + 2]);
    store_pt(A, 0, g, X); store_pt(A, 1, g, X);
    A.pt_active[g] = A.pt_obs_start[g + 1] > A.pt_obs_start[g];
  }
  for (int l = gid; l < W.n_ln; l += stride) {
    const int g = W.ln_off + l;
    const LineQ L = line_from_x0_dir(vec3(A.ln_x0[(size_t)g * 3], A.ln_x0[(size_t)g * 3 + 1], A.ln_x0[(size_t)g * 3 + 2]),
                                     vec3(A.ln_dir[(size_t)g * 3], A.ln_dir[(size_t)g * 3 + 1], A.ln_dir[(size_t)g * 3 + 2]));
    store_ln(A, 0, g, L); store_ln(A, 1, g, L);
    A.ln_active[g] = A.ln_obs_start[g + 1] > A.ln_obs_start[g];
    A.ln_removed[g] = 0;
  }
  for (int e = gid; e < W.n_pe; e += stride) {
    A.pe_flags[W.pe_off + e] = (uint8_t)(EF_VALID | (W.robust_pts ? EF_ROBUST : 0) | (pt_obs_stereo<kPkRuntime>(A, W.pe_off + e) ? EF_STEREO : 0));
    A.pe_chi2[W.pe_off + e] = 0.0; A.pe_ws[W.pe_off + e] = 0.0;
  }
  for (int e = gid; e < W.n_le; e += stride) {
    const uint8_t f0 = ln_flags0_of<kPkRuntime>(A, W.le_off + e);
    A.le_flags[W.le_off + e] = (f0 & EF_VALID) ? (uint8_t)(f0 | EF_ROBUST) : (uint8_t)0;
    A.le_chi2[W.le_off + e] = 0.0;
  }
  for (int i = gid; i < W.n_free * 21; i += stride) A.Hpp[(size_t)W.hpp_off * 21 + i] = 0.0;
  for (int i = gid; i < W.n_free * 6; i += stride) A.bp[(size_t)W.hpp_off * 6 + i] = 0.0;
  if (W.big) for (int i = gid; i < W.n_free * 27; i += stride) A.hpp_part[W.hpart_off + i] = 0.0;
  if (gid == 0) {
    BAState s;
    memset(&s, 0, sizeof s);
    const int n_edges = W.n_pe + W.n_le;     // an empty graph skips straight to the read-back
    s.phase = n_edges > 0 ? PH_RUN : PH_FINALIZE;
    s.need_lin = 1; s.lambda = -1.0; s.ni = 2.0;
    st[blockIdx.y] = s;
    if (A.slot_map) { A.slot_map[blockIdx.y] = (int)blockIdx.y; A.active_pub[blockIdx.y] = 1; }   // every window has its own grid row until the first LM control rebuilds the map
  }
}


// ---- the kernel families (each file continues namespace lldba; order matters: later families use helpers of earlier ones)
#include "lld_ba_points.h"     // ba_linearize_pt_*, ba_backsub_pt_*
#include "lld_ba_lines.h"      // ba_linearize_ln_*, ba_linearize_both, ba_backsub_ln_*
#include "lld_ba_schur.h"      // ba_schur_items*, ba_schur_wide, ba_schur_reduce, ba_symmetrize
#include "lld_ba_solve.h"      // ba_pcg*, ba_chol*, lld_ba_chol_sparse.h
#include "lld_ba_control.h"    // ba_control, ba_backsub_ctl, ba_classify, ba_finalize

}  // namespace lldba
#endif
