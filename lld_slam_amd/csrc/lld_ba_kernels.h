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


// ================================================================== point landmarks: one lane per EDGE
// Edge SoA arrays are read fully coalesced (lane i <-> edge e0 + i); what belongs to a landmark (Hll, b_l, the back-substituted
// update) is combined over the landmark's lanes with a segmented shuffle reduction, the landmark's first lane ("head") does the
// per-landmark work, and results travel back to the lanes with one shuffle.
// Shifts by one lane over the whole wavefront go through the VALU (v_mov_b32_dpp wave_shl:1 / wave_shr:1), not through the LDS pipe:
// tools/microbench/lds_ops.hip measures 6.3 CU clocks per ds_bpermute_b32 against 1.3 for a DPP move, and the LDS pipe is what bounds
// the linearise kernels (it also carries their fp64 atomics).  Shifts by 2 and 4 are chains of single shifts.
// dpp_down1: lane i <- lane i + 1, dpp_up1: lane i <- lane i - 1; the lane without a source receives 0 (bound_ctrl), so no register has
// to be preset with a fill value
__device__ __forceinline__ int dpp_down1(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_up1(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }
template <int OFF>
__device__ __forceinline__ int dpp_down(int v) {
#pragma unroll
  for (int h = 0; h < OFF; h++) v = dpp_down1(v);
  return v;
}
template <int OFF>
__device__ __forceinline__ double dpp_down(double v) {
  return __hiloint2double(dpp_down<OFF>(__double2hiint(v)), dpp_down<OFF>(__double2loint(v)));
}
__device__ __forceinline__ bool seg_step(int seg, int lane, int off) {
  const int so = __shfl_down(seg, off);
  return (lane + off < 64) && so == seg;
}
// one step of the segmented sum: v += (value OFF lanes up, if that lane is in the same segment).  seg1 = segment id + 1 is never 0 in a
// lane that can receive the zero fill (ids are >= 0, or -1 - lane in idle lanes, and lane 0 receives no fill).  The condition enters as
// a factor 0.0 / 1.0 of one fused multiply-add per value (the partial sums are finite: idle lanes hold zeros).
template <int N, int OFF>
__device__ __forceinline__ void seg_sum_step(double* v, int seg1) {
  const bool ok = dpp_down<OFF>(seg1) == seg1;
  double o[N];
#pragma unroll
  for (int i = 0; i < N; i++) o[i] = dpp_down<OFF>(v[i]);
  // one predicated block of adds (EXEC = the lanes that continue their segment): a non-finite partial of a NEIGHBOURING landmark
  // cannot leak in, which a 0.0 / 1.0 factor in an FMA would let it do (0 * NaN)
  if (ok) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += o[i];
  }
}
template <int N>
__device__ __forceinline__ void seg_sum(double* v, int seg, int lane, int max_len) {      // valid in the first lane of every segment
  if (max_len > 1) seg_sum_step<N, 1>(v, seg + 1);
  if (max_len > 2) seg_sum_step<N, 2>(v, seg + 1);
  if (max_len > 4) seg_sum_step<N, 4>(v, seg + 1);
  for (int off = 8; off < max_len; off <<= 1) {
    const bool ok = seg_step(seg, lane, off);
#pragma unroll
    for (int i = 0; i < N; i++) { const double o = __shfl_down(v[i], off); if (ok) v[i] += o; }
  }
}
template <int N>
__device__ __forceinline__ void wave_sum_n(double* v) {
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = wave_sum(v[i]);
}

struct PtEdgeLin { double r[3], Jp[9], Jc[18], ws, rho0; bool stereo; };

// residual, chi2 (stored), Huber weight, Jacobians of one active point edge at the linearisation point
template <int kPk>
__device__ __forceinline__ void point_edge_linearize(const BAArrays& A, const BAWin& W, int cur, int e, uint8_t fl, int c, const Vec3& X, PtEdgeLin& L) {
  const Pose T = load_cam(A, cur, W.cam_off + c);
  const Vec3 Xc = pose_map(T, X);
  const PtObs ob = pt_obs_of<kPk>(A, e);
  L.stereo = !(ob.ur < 0);
  point_residual_iz(W.cam, Xc, rcp_nr(Xc.z), ob.u, ob.v, ob.ur, L.stereo, true, L.r);
  const double s = ob.s;
  const double c2 = chi2_of(L.r, L.stereo ? 3 : 2, s);
  A.pe_chi2[e] = c2;
  double w = 1.0;
  L.rho0 = c2;
  if (fl & EF_ROBUST) L.rho0 = huber_nr(c2, L.stereo ? W.th_stereo : W.th_mono, &w);
  L.ws = w * s;
  A.pe_ws[e] = L.stereo ? -L.ws : L.ws;                   // (sign bit = stereo edge, see BAArrays::pe_ws)
  point_jac_point(W.cam, Xc, quat_rotation(T.q), L.stereo, L.Jp);
  point_jac_pose(W.cam, Xc, L.stereo, L.Jc);
}
// landmark side Hll (6 upper) + b_l (3) of one edge
__device__ __forceinline__ void point_edge_hll(const PtEdgeLin& L, double* hb) {
  int k = 0;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int d = a; d < 3; d++) hb[k++] = L.ws * (L.Jp[a] * L.Jp[d] + L.Jp[3 + a] * L.Jp[3 + d] + L.Jp[6 + a] * L.Jp[6 + d]);
#pragma unroll
  for (int a = 0; a < 3; a++) hb[6 + a] = -L.ws * (L.Jp[a] * L.r[0] + L.Jp[3 + a] * L.r[1] + L.Jp[6 + a] * L.r[2]);
}
// camera side: Hpp (21 upper) and b_p (6) into the LDS-staged per-camera accumulators
__device__ __forceinline__ void point_edge_hpp(const PtEdgeLin& L, double* ac) {
  int kk = 0;
#pragma unroll
  for (int rr = 0; rr < 6; rr++) {
    atomicAdd(&ac[21 + rr], -L.ws * (L.Jc[rr] * L.r[0] + L.Jc[6 + rr] * L.r[1] + L.Jc[12 + rr] * L.r[2]));
#pragma unroll
    for (int cc = rr; cc < 6; cc++) atomicAdd(&ac[kk++], L.ws * (L.Jc[rr] * L.Jc[cc] + L.Jc[6 + rr] * L.Jc[6 + cc] + L.Jc[12 + rr] * L.Jc[12 + cc]));
  }
}

// Everything one active point edge adds to the normal equations, in closed form (same algebra as point_hpl_closed): with
// A = d(u,v,uR)/dXc, M = ws A^T A, g = A^T (ws r), P = [Xc]x M:
//   Hll = R^T M R, b_l = R^T g;  Hpp = [[ Xc x P_i (rows) , P ], [ . , M ]], b_p = [ Xc x g ; g ]
// instead of forming Jp (3x3) and Jc (3x6) and contracting them.  hb: 6 upper of Hll + b_l; hp: 21 upper of Hpp + b_p (row-major
// packed like point_edge_hpp).  Returns chi2 of the edge; ws and rho0 through the references.
__device__ __forceinline__ double point_edge_blocks_closed(const BAWin& W, const Pose& T, const Vec3& X, const PtObs& ob, uint8_t fl, double& ws_out,
                                                           double& rho0_out, double* hb, double* hp) {
  const CamK& k = W.cam;
  const Vec3 Xc = pose_map(T, X);
  const bool stereo = !(ob.ur < 0);
  double r[3];
  const double iz = rcp_nr(Xc.z), iz2 = iz * iz;             // one reciprocal for the residual and the Jacobian entries
  point_residual_iz(k, Xc, iz, ob.u, ob.v, ob.ur, stereo, true, r);
  const double c2e = chi2_of(r, stereo ? 3 : 2, ob.s);
  double w = 1.0, rho0 = c2e;
  if (fl & EF_ROBUST) rho0 = huber_nr(c2e, stereo ? W.th_stereo : W.th_mono, &w);
  const double ws = w * ob.s;
  ws_out = ws; rho0_out = rho0;
  const Mat3 R = quat_rotation(T.q);
  const double a = k.fx * iz, b = k.fy * iz;
  const double c0 = -k.fx * Xc.x * iz2, c1 = -k.fy * Xc.y * iz2, c2 = c0 + k.bf * iz2;
  const double m00 = ws * (stereo ? 2.0 * a * a : a * a);
  const double m02 = ws * (stereo ? a * (c0 + c2) : a * c0);
  const double m11 = ws * (b * b), m12 = ws * (b * c1);
  const double m22 = ws * (stereo ? c0 * c0 + c1 * c1 + c2 * c2 : c0 * c0 + c1 * c1);
  const double wr0 = ws * r[0], wr1 = ws * r[1], wr2 = stereo ? ws * r[2] : 0.0;
  const double g0 = a * (wr0 + wr2), g1 = b * wr1, g2 = c0 * wr0 + c1 * wr1 + c2 * wr2;
  // landmark side
  double G[3][3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    G[0][j] = m00 * R.m[0][j] + m02 * R.m[2][j];
    G[1][j] = m11 * R.m[1][j] + m12 * R.m[2][j];
    G[2][j] = m02 * R.m[0][j] + m12 * R.m[1][j] + m22 * R.m[2][j];
  }
  int kk = 0;
#pragma unroll
  for (int p = 0; p < 3; p++)
#pragma unroll
    for (int d = p; d < 3; d++) hb[kk++] = R.m[0][p] * G[0][d] + R.m[1][p] * G[1][d] + R.m[2][p] * G[2][d];
#pragma unroll
  for (int p = 0; p < 3; p++) hb[6 + p] = R.m[0][p] * g0 + R.m[1][p] * g1 + R.m[2][p] * g2;
  // camera side: P = [Xc]x M (column j = Xc x M[:,j]); M is symmetric with m01 = 0
  const double x = Xc.x, y = Xc.y, z = Xc.z;
  const double P[3][3] = {{y * m02 - z * 0.0, y * m12 - z * m11, y * m22 - z * m12},
                          {z * m00 - x * m02, z * 0.0 - x * m12, z * m02 - x * m22},
                          {x * 0.0 - y * m00, x * m11 - y * 0.0, x * m12 - y * m02}};
  // rotation-rotation block: row i = Xc x P[i,:]
  const double Q[3][3] = {{y * P[0][2] - z * P[0][1], z * P[0][0] - x * P[0][2], x * P[0][1] - y * P[0][0]},
                          {y * P[1][2] - z * P[1][1], z * P[1][0] - x * P[1][2], x * P[1][1] - y * P[1][0]},
                          {y * P[2][2] - z * P[2][1], z * P[2][0] - x * P[2][2], x * P[2][1] - y * P[2][0]}};
  // packed upper triangle, rows 0..5: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
  hp[0] = Q[0][0]; hp[1] = Q[0][1]; hp[2] = Q[0][2]; hp[3] = P[0][0]; hp[4] = P[0][1]; hp[5] = P[0][2];
  hp[6] = Q[1][1]; hp[7] = Q[1][2]; hp[8] = P[1][0]; hp[9] = P[1][1]; hp[10] = P[1][2];
  hp[11] = Q[2][2]; hp[12] = P[2][0]; hp[13] = P[2][1]; hp[14] = P[2][2];
  hp[15] = m00; hp[16] = 0.0; hp[17] = m02; hp[18] = m11; hp[19] = m12; hp[20] = m22;
  hp[21] = y * g2 - z * g1; hp[22] = z * g0 - x * g2; hp[23] = x * g1 - y * g0; hp[24] = g0; hp[25] = g1; hp[26] = g2;
  return c2e;
}

// grid (nl_pt, nW), block 512 = 8 wavefronts, BAWin::rounds tasks per wavefront; dynamic LDS: kAccCopies*n_free_max*27 doubles + 8 scratch.
// kBig (a map with more cameras than the LDS holds accumulators and poses for, BAWin::big): the camera accumulators are ONE row in HBM
// (zeroed by ba_init / ba_control / ba_round2, added to with global fp64 atomics) and the poses are read from HBM.
template <bool kBig, int kPk>
__device__ __forceinline__ void ba_linearize_pt_body(const BAArrays& A, const BAWin* __restrict__ wins, BAState* __restrict__ st, const int bx) {   // bx: the workgroup's index along x
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN || !S.need_lin) return;
  if ((int)bx >= W.nl_pt) return;
  const int nacc = W.n_free * 27;
  const int cur = S.cur;
  double* acc_all = kBig ? A.hpp_part + W.hpart_off : lds;   // kAccCopies x [n_free][21 Hpp upper + 6 bp]
  const int copies = W.acc_copies[0];
  double* scratch = kBig ? lds : lds + copies * nacc;
  double* cams_l = scratch + 8;                              // [n_cams][7] poses of the linearisation point
  const double* cams = kBig ? A.cam_qt + ((size_t)cur * A.NC + W.cam_off) * 7 : cams_l;
  double* acc = acc_all;
  const int nthr = blockDim.x, nwv = W.lin_waves[0];         // 512 / 8; bit-reproducible mode: one wavefront per accumulator copy
  // the wavefront's tasks: the first one is fetched while the workgroup stages its LDS copies, the next one while the current one is
  // worked on (scalar loads: a task index never waits for a dependent load of its own inside the loop).  Fetching the per-lane operands
  // of the next task as well (13 registers: edge record, camera word, flags, landmark) was measured and dropped: 128 VGPRs with 10
  // spilled, ba_linearize 13.75 -> 15.2 ms per step.
  const int task_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PTask T_next = A.ptasks[W.ptask_off + min((bx * W.rounds[0]) * nwv + task_wave, W.n_ptasks - 1)];
  if (!kBig) {
    for (int i = threadIdx.x; i < copies * nacc; i += nthr) acc_all[i] = 0.0;
    // Default: the lanes of a wavefront are spread over the copies (same-address LDS atomics serialise) and every copy is shared by all
    // wavefronts - the order of the adds varies from run to run.  Deterministic mode: copy = wavefront, so a copy only ever sees ONE
    // wavefront's adds, in program order (lanes of one instruction that hit the same camera are serialised by the LDS in lane order).
    acc = acc_all + (W.det ? (int)(threadIdx.x >> 6) : (int)((threadIdx.x >> 3) & (copies - 1))) * nacc;
    for (int i = threadIdx.x; i < W.n_cams * 7; i += nthr) cams_l[i] = A.cam_qt[((size_t)cur * A.NC + W.cam_off) * 7 + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double chi = 0.0, maxd = 0.0;
  for (int rnd = 0; rnd < W.rounds[0]; rnd++) {
    const int ti = (bx * W.rounds[0] + rnd) * nwv + task_wave;      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
    if (ti >= W.n_ptasks) break;
    const PTask T = T_next;
    T_next = A.ptasks[W.ptask_off + min((bx * W.rounds[0] + rnd + 1) * nwv + task_wave, W.n_ptasks - 1)];
    if (T.nl > 1) {
      // two dependent memory levels only: (1) the task, (2) every global operand - edge arrays by edge lane, landmark
      // state by landmark lane (lane i <-> landmark l0 + i); camera poses come from the workgroup's LDS copy and landmark
      // data reaches the edge lanes by shuffle.
      const bool has = lane < T.ne;
      const int e = T.e0 + (has ? lane : 0);
      int c, l_raw;
      pt_cam_lm_of<kPk>(A, e, T.l0, c, l_raw);
      const int l = has ? l_raw : -1 - lane;
      const uint8_t fl = A.pe_flags[e];
      const PtObs ob = pt_obs_of<kPk>(A, e);
      const bool lmk = lane < T.nl;
      const int g2 = W.pt_off + T.l0 + (lmk ? lane : 0);
      const Vec3 X2 = load_pt(A, cur, g2);
      const int act2 = lmk ? (int)A.pt_active[g2] : 0;
      const int slot = has ? l - T.l0 : 0;
      Vec3 X; X.x = __shfl(X2.x, slot); X.y = __shfl(X2.y, slot); X.z = __shfl(X2.z, slot);
      const bool lm_act = has && __shfl(act2, slot) != 0;
      const bool head = lm_act && dpp_up1(l + 1) != l + 1;               // the first edge lane of an active landmark (lane 0 receives 0)
      double hb[9];
#pragma unroll
      for (int i = 0; i < 9; i++) hb[i] = 0.0;
      if (has && (fl & EF_LEVEL1)) A.pe_ws[e] = 0.0;
      if (lm_act && !(fl & EF_LEVEL1)) {
        double hp[27], ws_e, rho0_e;
        A.pe_chi2[e] = point_edge_blocks_closed(W, pose_load(cams + c * 7), X, ob, fl, ws_e, rho0_e, hb, hp);
        A.pe_ws[e] = (fl & EF_STEREO) ? -ws_e : ws_e;
        chi += rho0_e;
        if (c < W.n_free) {
          double* ac = acc + c * 27;
#pragma unroll
          for (int i = 0; i < 27; i++) if (i != 16) atomicAdd(&ac[i], hp[i]);          // entry 16 is the structural zero of M
        }
      }
      seg_sum<9>(hb, l, lane, T.ms);
      // the head lane of a landmark holds the sums: it writes Hll / b_l itself (no trip back to the landmark lane)
      if (head) {
        double* V = A.pt_V + (size_t)(W.pt_off + l) * 9;
#pragma unroll
        for (int i = 0; i < 9; i++) V[i] = hb[i];
        maxd = fmax(maxd, fmax(fabs(hb[0]), fmax(fabs(hb[3]), fabs(hb[5]))));
      }
    } else {                                             // a single landmark, any number of edges
      const int g = W.pt_off + T.l0;
      if (A.pt_active[g]) {
        const Vec3 X = load_pt(A, cur, g);
        double hb[9];
#pragma unroll
        for (int i = 0; i < 9; i++) hb[i] = 0.0;
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int e = T.e0 + sidx;
          const uint8_t fl = A.pe_flags[e];
          if (fl & EF_LEVEL1) { A.pe_ws[e] = 0.0; continue; }
          const int c = pt_cam_of<kPk>(A, e);
          PtEdgeLin L;
          point_edge_linearize<kPk>(A, W, cur, e, fl, c, X, L);
          chi += L.rho0;
          double h1[9];
          point_edge_hll(L, h1);
#pragma unroll
          for (int i = 0; i < 9; i++) hb[i] += h1[i];
          if (c < W.n_free) point_edge_hpp(L, acc + c * 27);
        }
        wave_sum_n<9>(hb);
        if (lane == 0) {
          double* V = A.pt_V + (size_t)g * 9;
#pragma unroll
          for (int i = 0; i < 9; i++) V[i] = hb[i];
          maxd = fmax(maxd, fmax(fabs(hb[0]), fmax(fabs(hb[3]), fabs(hb[5]))));
        }
      }
    }
  }
  const double chi_t = block_sum(chi, scratch);
  const double max_t = block_max(maxd, scratch);
  if (threadIdx.x == 0) {
    A.chi_part[W.part_off + bx] = chi_t;
    atomicMax(&S.maxdiag_bits, (unsigned long long)__double_as_longlong(max_t));
  }
  __syncthreads();
  if (kBig) return;
  // plain stores of this workgroup's camera partials; ba_hpp_reduce sums them in a fixed order (no global atomics)
  double* dst = A.hpp_part + W.hpart_off + (size_t)(bx) * nacc;
  for (int i = threadIdx.x; i < nacc; i += nthr) {
    double v = 0.0;
    for (int q = 0; q < copies; q++) v += acc_all[q * nacc + i];
    dst[i] = v;
  }
}
__global__ __launch_bounds__(kLinThreads, 4) void ba_linearize_pt_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_pt_body<false, 1>(A, wins, st, (int)blockIdx.x); }
__global__ __launch_bounds__(kLinThreads, 4) void ba_linearize_pt_f64_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_pt_body<false, 0>(A, wins, st, (int)blockIdx.x); }      // observations as given (BAArrays::packed = 0)
__global__ __launch_bounds__(kLinThreads, 4) void ba_linearize_pt_big_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_pt_body<true, kPkRuntime>(A, wins, st, (int)blockIdx.x); }

// W_e^T x_c = ws * Jp^T (Jc x_c) of one point edge with the Jacobians of the linearisation point
__device__ __forceinline__ void point_edge_wtx(const BAArrays& A, const BAWin& W, int cur, int e, uint8_t fl, int c, const Vec3& X, const double* xp, double* t) {
  const double ws = fabs(A.pe_ws[e]);
  const bool stereo = (fl & EF_STEREO) != 0;
  const Pose T = load_cam(A, cur, W.cam_off + c);
  const Vec3 Xc = pose_map(T, X);
  double Jp[9], Jc[18];
  point_jac_point(W.cam, Xc, quat_rotation(T.q), stereo, Jp);
  point_jac_pose(W.cam, Xc, stereo, Jc);
  double uu[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; r++) s += Jc[i * 6 + r] * xp[c * 6 + r];
    uu[i] = ws * s;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) t[k] = Jp[k] * uu[0] + Jp[3 + k] * uu[1] + Jp[6 + k] * uu[2];
}
// trial-state residual of one active point edge: stores chi2, returns its (robust) cost
template <int kPk>
__device__ __forceinline__ double point_edge_trial(const BAArrays& A, const BAWin& W, int nxt, int e, uint8_t fl, int c, const Vec3& Xn) {
  const Pose T = load_cam(A, nxt, W.cam_off + c);
  const Vec3 Xc = pose_map(T, Xn);
  const PtObs ob = pt_obs_of<kPk>(A, e);
  const bool stereo = !(ob.ur < 0);
  double r[3];
  point_residual_iz(W.cam, Xc, rcp_nr(Xc.z), ob.u, ob.v, ob.ur, stereo, true, r);
  const double c2 = chi2_of(r, stereo ? 3 : 2, ob.s);
  A.pe_chi2[e] = c2;
  double w, rho0 = c2;
  if (fl & EF_ROBUST) rho0 = huber_nr(c2, stereo ? W.th_stereo : W.th_mono, &w);
  return rho0;
}
// x_l = (Hll + lambda I)^-1 (b_l - sum W^T x_c), oplus; returns the landmark's part of computeScale
__device__ __forceinline__ double point_backsub(const double* V, double lambda, const double* wtx, const Vec3& X, Vec3& Xn) {
  const double t[3] = {V[6] - wtx[0], V[7] - wtx[1], V[8] - wtx[2]};
  double xl[3], sc = 0.0;
  chol_solve<3>(V, lambda, t, xl);
#pragma unroll
  for (int i = 0; i < 3; i++) sc += xl[i] * (lambda * xl[i] + V[6 + i]);
  Xn = vec3(X.x + xl[0], X.y + xl[1], X.z + xl[2]);      // VertexSBAPointXYZ::oplusImpl
  return sc;
}

// grid (nt_pt, nW), block 256 = 4 wavefronts, BAWin::rounds tasks per wavefront; dynamic LDS: 8 + 14 n_cams + 6 n_free doubles
template <bool kBig, int kPk>
__device__ __forceinline__ void ba_backsub_pt_body(const BAArrays& A, const BAWin* __restrict__ wins, const BAState* __restrict__ st, const int bx) {   // bx: the workgroup's index along x
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  if ((int)bx >= W.nt_pt) return;
  const int cur = S.cur, nxt = cur ^ 1;
  const double lambda = S.lambda;
  const double* xp = A.xp + W.x_off;
  // workgroup copies of what every edge lane gathers: poses of the linearisation point (camA) and of the trial state (camB),
  // and the camera part of the solution
  double* scratch = lds;
  double* camA_l = lds + 8;
  double* camB_l = camA_l + W.n_cams * 7;
  double* xps_l = camB_l + W.n_cams * 7;
  // kBig: no LDS copies, the poses and x_c are read from HBM (see ba_linearize_pt_body)
  const double* camA = kBig ? A.cam_qt + ((size_t)cur * A.NC + W.cam_off) * 7 : camA_l;
  const double* camB = kBig ? A.cam_qt + ((size_t)nxt * A.NC + W.cam_off) * 7 : camB_l;
  const double* xps = kBig ? xp : xps_l;
  // the wavefront's tasks: the first one is fetched while the workgroup stages its LDS copies, the next one while the current one is
  // worked on (scalar loads: a task index never waits for a dependent load of its own inside the loop)
  const int task_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PTask T_next = A.ptasks[W.ptask_off + min((bx * W.rounds[2]) * 4 + task_wave, W.n_ptasks - 1)];
  if (!kBig) {
    for (int i = threadIdx.x; i < W.n_cams * 7; i += kLmThreads) {
      camA_l[i] = A.cam_qt[((size_t)cur * A.NC + W.cam_off) * 7 + i];
      camB_l[i] = A.cam_qt[((size_t)nxt * A.NC + W.cam_off) * 7 + i];
    }
    for (int i = threadIdx.x; i < 6 * W.n_free; i += kLmThreads) xps_l[i] = xp[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double chi = 0.0, sc = 0.0;
  for (int rnd = 0; rnd < W.rounds[2]; rnd++) {
    const int ti = (bx * W.rounds[2] + rnd) * 4 + task_wave;      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
    if (ti >= W.n_ptasks) break;
    const PTask T = T_next;
    T_next = A.ptasks[W.ptask_off + min((bx * W.rounds[2] + rnd + 1) * 4 + task_wave, W.n_ptasks - 1)];
    if (T.nl > 1) {
      const bool has = lane < T.ne;
      const int e = T.e0 + (has ? lane : 0);
      int c, l_raw;
      pt_cam_lm_of<kPk>(A, e, T.l0, c, l_raw);
      const int l = has ? l_raw : -1 - lane;
      const uint8_t fl = A.pe_flags[e];
      const double ws = fabs(A.pe_ws[e]);
      const PtObs ob = pt_obs_of<kPk>(A, e);
      const bool lmk = lane < T.nl;
      const int g2 = W.pt_off + T.l0 + (lmk ? lane : 0);
      const Vec3 X2 = load_pt(A, cur, g2);
      const int act2 = lmk ? (int)A.pt_active[g2] : 0;
      const int start2 = A.pt_obs_start[g2], end2 = A.pt_obs_start[g2 + 1];
      double V2[9];
#pragma unroll
      for (int i = 0; i < 9; i++) V2[i] = A.pt_V[(size_t)g2 * 9 + i];
      const int slot = has ? l - T.l0 : 0;
      Vec3 X; X.x = __shfl(X2.x, slot); X.y = __shfl(X2.y, slot); X.z = __shfl(X2.z, slot);
      const bool e_act = has && __shfl(act2, slot) != 0 && !(fl & EF_LEVEL1);
      double wtx[3] = {0, 0, 0};
      if (e_act && c < W.n_free) {
        // W_e^T x_c = ws * Jp^T (Jc x_c) with the Jacobians of the linearisation point
        // closed form (see point_hpl_closed): Jc x = A (Xc x x_w - x_t), Jp^T u = -R^T A^T u
        const Pose Tc = pose_load(camA + c * 7);
        const Vec3 Xc = pose_map(Tc, X);
        const bool stereo = (fl & EF_STEREO) != 0;
        const Mat3 R = quat_rotation(Tc.q);
        const double* xc = xps + c * 6;
        const Vec3 v = cross(Xc, vec3(xc[0], xc[1], xc[2])) - vec3(xc[3], xc[4], xc[5]);
        const double iz = rcp_nr(Xc.z), iz2 = iz * iz;
        const double a = W.cam.fx * iz, b = W.cam.fy * iz;
        const double c0 = -W.cam.fx * Xc.x * iz2, c1 = -W.cam.fy * Xc.y * iz2, c2 = c0 + W.cam.bf * iz2;
        const double u0 = ws * (a * v.x + c0 * v.z), u1 = ws * (b * v.y + c1 * v.z), u2 = stereo ? ws * (a * v.x + c2 * v.z) : 0.0;
        const double h0 = a * (u0 + u2), h1 = b * u1, h2 = c0 * u0 + c1 * u1 + c2 * u2;
#pragma unroll
        for (int k = 0; k < 3; k++) wtx[k] = -(R.m[0][k] * h0 + R.m[1][k] * h1 + R.m[2][k] * h2);
      }
      seg_sum<3>(wtx, l, lane, T.ms);
      // landmark lane: back-substitution and oplus of its landmark (inactive / edge-less landmarks keep their state)
      const int first = (lmk && end2 > start2) ? start2 - T.e0 : 0;
      double wl[3];
#pragma unroll
      for (int i = 0; i < 3; i++) wl[i] = __shfl(wtx[i], first);
      Vec3 Xn2 = X2;
      if (lmk) {
        if (act2 && end2 > start2) sc += point_backsub(V2, lambda, wl, X2, Xn2);
        store_pt(A, nxt, g2, Xn2);
      }
      Vec3 Xn; Xn.x = __shfl(Xn2.x, slot); Xn.y = __shfl(Xn2.y, slot); Xn.z = __shfl(Xn2.z, slot);
      if (e_act) {
        const Vec3 Xc = pose_map(pose_load(camB + c * 7), Xn);
        const bool stereo = !(ob.ur < 0);
        double r[3];
        point_residual_iz(W.cam, Xc, rcp_nr(Xc.z), ob.u, ob.v, ob.ur, stereo, true, r);
        const double c2 = chi2_of(r, stereo ? 3 : 2, ob.s);
        A.pe_chi2[e] = c2;
        double w, rho0 = c2;
        if (fl & EF_ROBUST) rho0 = huber_nr(c2, stereo ? W.th_stereo : W.th_mono, &w);
        chi += rho0;
      }
    } else {
      const int g = W.pt_off + T.l0;
      const Vec3 X = load_pt(A, cur, g);
      if (!A.pt_active[g]) { if (lane == 0) store_pt(A, nxt, g, X); }
      else {
        double wtx[3] = {0, 0, 0};
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int e = T.e0 + sidx;
          const uint8_t fl = A.pe_flags[e];
          const int c = pt_cam_of<kPk>(A, e);
          if ((fl & EF_LEVEL1) || c >= W.n_free) continue;
          double t1[3];
          point_edge_wtx(A, W, cur, e, fl, c, X, xp, t1);
          wtx[0] += t1[0]; wtx[1] += t1[1]; wtx[2] += t1[2];
        }
        wave_sum_n<3>(wtx);
        Vec3 Xn;
        const double s1 = point_backsub(A.pt_V + (size_t)g * 9, lambda, wtx, X, Xn);      // every lane, same value
        if (lane == 0) { sc += s1; store_pt(A, nxt, g, Xn); }
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int e = T.e0 + sidx;
          const uint8_t fl = A.pe_flags[e];
          if (fl & EF_LEVEL1) continue;
          chi += point_edge_trial<kPk>(A, W, nxt, e, fl, pt_cam_of<kPk>(A, e), Xn);
        }
      }
    }
  }
  const double chi_t = block_sum(chi, scratch);
  const double sc_t = block_sum(sc, scratch);
  if (threadIdx.x == 0) { xwg_store(&A.chi_part2[W.part_off + bx], chi_t); xwg_store(&A.scale_part[W.part_off + bx], sc_t); xwg_stores_done(); }
}
__global__ __launch_bounds__(kLmThreads) void ba_backsub_pt_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_pt_body<false, 1>(A, wins, st, (int)blockIdx.x); }
__global__ __launch_bounds__(kLmThreads) void ba_backsub_pt_f64_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_pt_body<false, 0>(A, wins, st, (int)blockIdx.x); }      // observations as given (BAArrays::packed = 0)
__global__ __launch_bounds__(kLmThreads) void ba_backsub_pt_big_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_pt_body<true, kPkRuntime>(A, wins, st, (int)blockIdx.x); }

// ================================================================== line landmarks: one lane per (line, KF) OBSERVATION
// Same scheme as the point kernels with the observation as the unit: a lane linearises the left and (if present) right image
// edge of its observation and keeps their summed Hpl block; Hll/b_l (14 values) are combined over the line's lanes.
struct LineGeom { Vec3 c0, c1, X1, X2; double alpha; };
__device__ __forceinline__ LineGeom line_geom(const LineQ& L) {
  const Mat3 Rl = line_rotation_t<true>(L);
  LineGeom G; G.c0 = mat_col(Rl, 0); G.c1 = mat_col(Rl, 1); G.alpha = L.alpha; G.X1 = L.alpha * G.c1; G.X2 = G.X1 + G.c0;
  return G;
}
// The two image edges of one (line, KF) observation as loaded: flags and (xs, ys, xe, ye, info) of the left and right slot.  Loaded for
// both slots at once and before anything is decided on them: one memory round trip per observation instead of one per slot behind a
// branch on the slot's flags (b_x needs no load: it is 0 for the left and CamK::bx_right for the right slot).
struct LnObsIn { uint8_t fl[2]; double xs[2], ys[2], xe[2], ye[2], s[2]; };
template <int kPk>
__device__ __forceinline__ void line_obs_load(const BAArrays& A, int o, LnObsIn& I) {
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int e = 2 * o + side;
    I.fl[side] = A.le_flags[e];
    const LnSeg g = ln_seg_of<kPk>(A, e);
    I.xs[side] = g.xs; I.ys[side] = g.ys; I.xe[side] = g.xe; I.ye[side] = g.ye;
  }
  if (obs_packed<kPk>(A)) { const unsigned oc = A.lo_oct[o]; I.s[0] = A.ln_info[oc & 255u]; I.s[1] = A.ln_info[oc >> 8]; }
  else { I.s[0] = A.le_s[2 * o]; I.s[1] = A.le_s[2 * o + 1]; }
}
// linearise one observation: hb (10 + 4) and the summed 6x4 Hpl block; returns the robust cost of its active edges
__device__ __forceinline__ double line_obs_linearize(const BAArrays& A, const BAWin& W, const Pose& T, int o, int c, const LineGeom& G, const LnObsIn& I,
                                                     double* hb, double* acc_lds) {
  const bool free_cam = c < W.n_free;
  double Wo[24], Jc0[12], r0[2] = {0.0, 0.0}, ws0 = 0.0;
#pragma unroll
  for (int i = 0; i < 24; i++) Wo[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 12; i++) Jc0[i] = 0.0;
  double chi = 0.0;
  const Mat3 Rc = quat_rotation(T.q);
  const Vec3 X1m = pose_map(T, G.X1), X2m = pose_map(T, G.X2);
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int e = 2 * o + side;
    const uint8_t fl = I.fl[side];
    if (!(fl & EF_VALID) || (fl & EF_LEVEL1)) continue;
    double r[2]; LineAdj adj;
    line_residual_t<true>(W.cam, side == 1 ? W.cam.bx_right : 0.0, X1m, X2m, I.xs[side], I.ys[side], I.xe[side], I.ye[side], r, &adj);
    const double s = I.s[side];
    const double c2 = chi2_of(r, 2, s);
    A.le_chi2[e] = c2;
    double w = 1.0, rho0 = c2;
    if (fl & EF_ROBUST) rho0 = huber_nr(c2, (fl & EF_PAIRSTEREO) ? W.th_ln_stereo : W.th_ln_mono, &w);
    chi += rho0;
    const double ws = w * s;
    double Jc[12], Jl[8];
    line_jac_pose(adj, X1m, X2m, Jc);
    line_jac_line(adj, Rc, G.c0, G.c1, G.alpha, Jl);
    int k = 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
      hb[10 + a] -= ws * (Jl[a] * r[0] + Jl[4 + a] * r[1]);
#pragma unroll
      for (int d = a; d < 4; d++) hb[k++] += ws * (Jl[a] * Jl[d] + Jl[4 + a] * Jl[4 + d]);
    }
    if (free_cam) {
#pragma unroll
      for (int rr = 0; rr < 6; rr++)
#pragma unroll
        for (int a = 0; a < 4; a++) Wo[rr * 4 + a] += ws * (Jc[rr] * Jl[a] + Jc[6 + rr] * Jl[4 + a]);
      // the camera block of BOTH image edges goes to the accumulators in one pass of LDS atomics (8 ... 24 CU clocks each,
      // tools/microbench/lds_ops.hip): the left edge only keeps its Jacobian, the right edge adds the sum
      if (side == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) Jc0[i] = Jc[i];
        r0[0] = r[0]; r0[1] = r[1]; ws0 = ws;
      } else {
        double* ac = acc_lds + c * 27;
        int kk = 0;
#pragma unroll
        for (int rr = 0; rr < 6; rr++) {
          atomicAdd(&ac[21 + rr], -(ws0 * (Jc0[rr] * r0[0] + Jc0[6 + rr] * r0[1]) + ws * (Jc[rr] * r[0] + Jc[6 + rr] * r[1])));
#pragma unroll
          for (int cc = rr; cc < 6; cc++)
            atomicAdd(&ac[kk++], ws0 * (Jc0[rr] * Jc0[cc] + Jc0[6 + rr] * Jc0[6 + cc]) + ws * (Jc[rr] * Jc[cc] + Jc[6 + rr] * Jc[6 + cc]));
        }
        ws0 = 0.0;                                           // added
      }
    }
  }
  if (free_cam) {
    double* Wb = A.lo_W + (size_t)o * 24;
#pragma unroll
    for (int i = 0; i < 24; i += 2) *reinterpret_cast<double2*>(Wb + i) = make_double2(Wo[i], Wo[i + 1]);
    if (ws0 != 0.0) {                                        // a left edge without an active right edge
      double* ac = acc_lds + c * 27;
      int kk = 0;
#pragma unroll
      for (int rr = 0; rr < 6; rr++) {
        atomicAdd(&ac[21 + rr], -ws0 * (Jc0[rr] * r0[0] + Jc0[6 + rr] * r0[1]));
#pragma unroll
        for (int cc = rr; cc < 6; cc++) atomicAdd(&ac[kk++], ws0 * (Jc0[rr] * Jc0[cc] + Jc0[6 + rr] * Jc0[6 + cc]));
      }
    }
  }
  return chi;
}

// grid (nl_ln, nW), block 512 = 8 wavefronts, BAWin::rounds tasks per wavefront; dynamic LDS: kAccCopies*n_free_max*27 doubles + 8 scratch.
template <bool kBig, int kPk>
__device__ __forceinline__ void ba_linearize_ln_body(const BAArrays& A, const BAWin* __restrict__ wins, BAState* __restrict__ st, const int bx) {   // bx: the workgroup's index along x
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN || !S.need_lin) return;
  if ((int)bx >= W.nl_ln) return;
  const int nacc = W.n_free * 27;
  double* acc_all = kBig ? A.hpp_part + W.hpart_off : lds;   // kAccCopies x [n_free][21 Hpp upper + 6 bp]; kBig: see ba_linearize_pt_body
  const int copies = W.acc_copies[1];
  double* scratch = kBig ? lds : lds + copies * nacc;
  double* acc = acc_all;
  const int cur = S.cur;
  double* cams_l = scratch + 8;                              // [n_cams][7] poses of the linearisation point
  const double* cams = kBig ? A.cam_qt + ((size_t)cur * A.NC + W.cam_off) * 7 : cams_l;
  const int nthr = blockDim.x, nwv = W.lin_waves[1];
  if (!kBig) {
    for (int i = threadIdx.x; i < copies * nacc; i += nthr) acc_all[i] = 0.0;
    acc = acc_all + (W.det ? (int)(threadIdx.x >> 6) : (int)((threadIdx.x >> 3) & (copies - 1))) * nacc;      // see ba_linearize_pt_body
    for (int i = threadIdx.x; i < W.n_cams * 7; i += nthr) cams_l[i] = A.cam_qt[((size_t)cur * A.NC + W.cam_off) * 7 + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double chi = 0.0, maxd = 0.0;
  for (int rnd = 0; rnd < W.rounds[1]; rnd++) {
    const int ti = (bx * W.rounds[1] + rnd) * nwv + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
    if (ti >= W.n_ltasks) break;
    const PTask T = A.ltasks[W.ltask_off + ti];
    double hb[14];
#pragma unroll
    for (int i = 0; i < 14; i++) hb[i] = 0.0;
    if (T.nl > 1) {
      // two dependent memory levels only: (1) everything addressed by the observation - line index, camera, both edge slots -
      // (2) the line's state; the camera pose comes from the workgroup's LDS copy
      const bool has = lane < T.ne;
      const int o = T.e0 + (has ? lane : 0);
      int c, l_raw;
      ln_cam_lm_of<kPk>(A, o, T.l0, c, l_raw);
      LnObsIn I;
      line_obs_load<kPk>(A, o, I);
      const int l = has ? l_raw : -1 - lane;
      const int g = W.ln_off + (has ? l : T.l0);
      const LineQ Lq = load_ln(A, cur, g);
      const bool lm_act = has && A.ln_active[g];
      if (lm_act) chi += line_obs_linearize(A, W, pose_load(cams + c * 7), o, c, line_geom(Lq), I, hb, acc);
      seg_sum<14>(hb, l, lane, T.ms);
      if (lm_act && o == A.ln_obs_start[g]) {
        double* V = A.ln_V + (size_t)g * 14;
#pragma unroll
        for (int i = 0; i < 14; i++) V[i] = hb[i];
        maxd = fmax(maxd, fmax(fmax(fabs(hb[0]), fabs(hb[4])), fmax(fabs(hb[7]), fabs(hb[9]))));
      }
    } else {
      const int g = W.ln_off + T.l0;
      if (A.ln_active[g]) {
        const LineGeom G = line_geom(load_ln(A, cur, g));
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int o = T.e0 + sidx, c = ln_cam_of<kPk>(A, o);
          LnObsIn I;
          line_obs_load<kPk>(A, o, I);
          chi += line_obs_linearize(A, W, pose_load(cams + c * 7), o, c, G, I, hb, acc);
        }
        wave_sum_n<14>(hb);
        if (lane == 0) {
          double* V = A.ln_V + (size_t)g * 14;
#pragma unroll
          for (int i = 0; i < 14; i++) V[i] = hb[i];
          maxd = fmax(maxd, fmax(fmax(fabs(hb[0]), fabs(hb[4])), fmax(fabs(hb[7]), fabs(hb[9]))));
        }
      }
    }
  }
  const double chi_t = block_sum(chi, scratch);
  const double max_t = block_max(maxd, scratch);
  if (threadIdx.x == 0) {
    A.chi_part[W.part_off + W.nl_pt + bx] = chi_t;
    atomicMax(&S.maxdiag_bits, (unsigned long long)__double_as_longlong(max_t));
  }
  __syncthreads();
  if (kBig) return;
  // plain stores of this workgroup's camera partials; ba_hpp_reduce sums them in a fixed order (no global atomics)
  double* dst = A.hpp_part + W.hpart_off + (size_t)(W.nl_pt + bx) * nacc;
  for (int i = threadIdx.x; i < nacc; i += nthr) {
    double v = 0.0;
    for (int q = 0; q < copies; q++) v += acc_all[q * nacc + i];
    dst[i] = v;
  }
}
__global__ __launch_bounds__(kLinThreads) void ba_linearize_ln_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_ln_body<false, 1>(A, wins, st, (int)blockIdx.x); }
__global__ __launch_bounds__(kLinThreads) void ba_linearize_ln_f64_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_ln_body<false, 0>(A, wins, st, (int)blockIdx.x); }      // observations as given (BAArrays::packed = 0)
__global__ __launch_bounds__(kLinThreads) void ba_linearize_ln_big_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) { ba_linearize_ln_body<true, kPkRuntime>(A, wins, st, (int)blockIdx.x); }

__device__ __forceinline__ void line_obs_wtx(const BAArrays& A, const BAWin& W, int o, int c, const double* xp, double* t) {
  const double* Wb = A.lo_W + (size_t)o * 24;              // zero when both image edges are inactive
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; r++) s += Wb[r * 4 + k] * xp[c * 6 + r];
    t[k] = s;
  }
}
__device__ __forceinline__ double line_obs_trial(const BAArrays& A, const BAWin& W, const Pose& T, int o, const LineGeom& G, const LnObsIn& I) {
  double chi = 0.0;
  const Vec3 X1m = pose_map(T, G.X1), X2m = pose_map(T, G.X2);
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int e = 2 * o + side;
    const uint8_t fl = I.fl[side];
    if (!(fl & EF_VALID) || (fl & EF_LEVEL1)) continue;
    double r[2];
    line_residual_t<true>(W.cam, side == 1 ? W.cam.bx_right : 0.0, X1m, X2m, I.xs[side], I.ys[side], I.xe[side], I.ye[side], r, nullptr);
    const double c2 = chi2_of(r, 2, I.s[side]);
    A.le_chi2[e] = c2;
    double w, rho0 = c2;
    if (fl & EF_ROBUST) rho0 = huber_nr(c2, (fl & EF_PAIRSTEREO) ? W.th_ln_stereo : W.th_ln_mono, &w);
    chi += rho0;
  }
  return chi;
}
__device__ __forceinline__ double line_backsub(const double* V, double lambda, const double* wtx, const LineQ& L, LineQ& Ln) {
  double t[4], xl[4], sc = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) t[i] = V[10 + i] - wtx[i];
  chol_solve<4>(V, lambda, t, xl);
#pragma unroll
  for (int i = 0; i < 4; i++) sc += xl[i] * (lambda * xl[i] + V[10 + i]);
  Ln = line_oplus(L, xl);
  return sc;
}

// grid (nt_ln, nW), block 256 = 4 wavefronts, BAWin::rounds tasks per wavefront
// dynamic LDS: 8 + 7 n_cams + 6 n_free doubles (poses of the trial state, x_c); kBig: read from HBM instead (see ba_linearize_pt_body)
template <bool kBig, int kPk>
__device__ __forceinline__ void ba_backsub_ln_body(const BAArrays& A, const BAWin* __restrict__ wins, const BAState* __restrict__ st, const int bx) {   // bx: the workgroup's index along x
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* scratch = lds;
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  if ((int)bx >= W.nt_ln) return;
  const int cur = S.cur, nxt = cur ^ 1;
  const double lambda = S.lambda;
  double* camB_l = lds + 8; double* xps_l = camB_l + W.n_cams * 7;
  const double* camB = kBig ? A.cam_qt + ((size_t)nxt * A.NC + W.cam_off) * 7 : camB_l;
  const double* xp = kBig ? A.xp + W.x_off : xps_l;
  // (tasks fetched ahead: see ba_linearize_pt_body)
  const int task_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PTask T_next = A.ltasks[W.ltask_off + min((bx * W.rounds[3]) * 4 + task_wave, W.n_ltasks - 1)];
  if (!kBig) {
    for (int i = threadIdx.x; i < W.n_cams * 7; i += kLmThreads) camB_l[i] = A.cam_qt[((size_t)nxt * A.NC + W.cam_off) * 7 + i];
    for (int i = threadIdx.x; i < 6 * W.n_free; i += kLmThreads) xps_l[i] = A.xp[W.x_off + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double chi = 0.0, sc = 0.0;
  for (int rnd = 0; rnd < W.rounds[3]; rnd++) {
    const int ti = (bx * W.rounds[3] + rnd) * 4 + task_wave;      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
    if (ti >= W.n_ltasks) break;
    const PTask T = T_next;
    T_next = A.ltasks[W.ltask_off + min((bx * W.rounds[3] + rnd + 1) * 4 + task_wave, W.n_ltasks - 1)];
    if (T.nl > 1) {
      // two dependent memory levels, like the point kernel: (1) the task, (2) every global operand - the observation's arrays by observation
      // lane, the line's state, active byte and observation range by LANDMARK lane (lane i <-> line l0 + i: no trip through the line index
      // the observation carries); line data reaches the observation lanes by shuffle.  Until round 4 the line state hung off the
      // observation's line index (a third level) and Hll / b_l off the head lane's test against ln_obs_start (a fourth).
      const bool has = lane < T.ne;
      const int o = T.e0 + (has ? lane : 0);
      int c, l_raw;
      ln_cam_lm_of<kPk>(A, o, T.l0, c, l_raw);
      LnObsIn I;
      line_obs_load<kPk>(A, o, I);                              // (used after the back-substitution: in flight meanwhile)
      const bool lmk = lane < T.nl;
      const int g2 = W.ln_off + T.l0 + (lmk ? lane : 0);
      const LineQ L2 = load_ln(A, cur, g2);
      const int act2 = lmk ? (int)A.ln_active[g2] : 0;
      const int start2 = A.ln_obs_start[g2], end2 = A.ln_obs_start[g2 + 1];
      const int l = has ? l_raw : -1 - lane;
      const int slot = has ? l - T.l0 : 0;
      const bool lm_act = has && __shfl(act2, slot) != 0;
      double wtx[4] = {0, 0, 0, 0};
      if (lm_act && c < W.n_free) line_obs_wtx(A, W, o, c, xp, wtx);
      double V2[14];                                            // (issued once the 24 doubles of the Hpl block are consumed)
#pragma unroll
      for (int i = 0; i < 14; i++) V2[i] = A.ln_V[(size_t)g2 * 14 + i];
      seg_sum<4>(wtx, l, lane, T.ms);
      // landmark lane: back-substitution and oplus of its line (inactive / observation-less lines keep their state)
      const int first = (lmk && end2 > start2) ? start2 - T.e0 : 0;
      double wl[4];
#pragma unroll
      for (int i = 0; i < 4; i++) wl[i] = __shfl(wtx[i], first);
      LineQ Ln2 = L2;
      if (lmk) {
        if (act2 && end2 > start2) sc += line_backsub(V2, lambda, wl, L2, Ln2);
        store_ln(A, nxt, g2, Ln2);
      }
      LineQ Ln;
      Ln.q.x = __shfl(Ln2.q.x, slot); Ln.q.y = __shfl(Ln2.q.y, slot); Ln.q.z = __shfl(Ln2.q.z, slot); Ln.q.w = __shfl(Ln2.q.w, slot); Ln.alpha = __shfl(Ln2.alpha, slot);
      if (lm_act) chi += line_obs_trial(A, W, pose_load(camB + c * 7), o, line_geom(Ln), I);
    } else {
      const int g = W.ln_off + T.l0;
      const LineQ L = load_ln(A, cur, g);
      if (!A.ln_active[g]) { if (lane == 0) store_ln(A, nxt, g, L); }
      else {
        double wtx[4] = {0, 0, 0, 0};
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int o = T.e0 + sidx, c = ln_cam_of<kPk>(A, o);
          if (c >= W.n_free) continue;
          double t1[4];
          line_obs_wtx(A, W, o, c, xp, t1);
          wtx[0] += t1[0]; wtx[1] += t1[1]; wtx[2] += t1[2]; wtx[3] += t1[3];
        }
        wave_sum_n<4>(wtx);
        LineQ Ln;
        const double s1 = line_backsub(A.ln_V + (size_t)g * 14, lambda, wtx, L, Ln);
        if (lane == 0) { sc += s1; store_ln(A, nxt, g, Ln); }
        const LineGeom G = line_geom(Ln);
        for (int sidx = lane; sidx < T.ne; sidx += 64) {
          const int o = T.e0 + sidx;
          LnObsIn I;
          line_obs_load<kPk>(A, o, I);
          chi += line_obs_trial(A, W, pose_load(camB + ln_cam_of<kPk>(A, o) * 7), o, G, I);
        }
      }
    }
  }
  const double chi_t = block_sum(chi, scratch);
  const double sc_t = block_sum(sc, scratch);
  if (threadIdx.x == 0) { xwg_store(&A.chi_part2[W.part_off + W.nt_pt + bx], chi_t); xwg_store(&A.scale_part[W.part_off + W.nt_pt + bx], sc_t); xwg_stores_done(); }
}
__global__ __launch_bounds__(kLmThreads, 4) void ba_backsub_ln_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_ln_body<false, 1>(A, wins, st, (int)blockIdx.x); }
__global__ __launch_bounds__(kLmThreads, 4) void ba_backsub_ln_f64_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_ln_body<false, 0>(A, wins, st, (int)blockIdx.x); }      // observations as given (BAArrays::packed = 0)
__global__ __launch_bounds__(kLmThreads) void ba_backsub_ln_big_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) { ba_backsub_ln_body<true, kPkRuntime>(A, wins, st, (int)blockIdx.x); }

// Point and line landmarks in one launch, for batches too small to fill the GPU (a single window above all): there the two
// kernels of a pair are dependent launches of 8-16 us each on idle hardware.  Not for large batches: the fused kernel gets the
// register budget of the line body (223 VGPRs), which would halve the occupancy of the point body.
__global__ __launch_bounds__(kLinThreads) void ba_linearize_both_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st, int n_pt_blocks) {
  if ((int)blockIdx.x < n_pt_blocks) ba_linearize_pt_body<false, 1>(A, wins, st, (int)blockIdx.x);      // (packed observations only: the host launches the
  else ba_linearize_ln_body<false, 1>(A, wins, st, (int)blockIdx.x - n_pt_blocks);                     //  two kernels of the pair otherwise)
}

// LM iteration head of one window, by ONE wavefront: chi2 of the current state, lambda initialisation at iteration 0
// (optimization_algorithm_levenberg.cpp:75-99,166-180).
__device__ __forceinline__ void ba_begin_body(const BAArrays& A, const BAWin& W, BAState& S, int lane) {
  // chi2 of the current state: lanes sum interleaved partials, then a fixed shuffle tree (deterministic)
  double chi = 0.0;
  const int nb = W.nl_pt + W.nl_ln;
  for (int i = lane; i < nb; i += 64) chi += A.chi_part[W.part_off + i];
  chi = wave_sum(chi);
  double md = 0.0;
  if (S.it == 0) {
    // computeLambdaInit: tau * max |H_kk| over cameras and landmarks (optimization_algorithm_levenberg.cpp:166-180)
    const double* H = A.Hpp + (size_t)W.hpp_off * 21;
    for (int c = lane; c < W.n_free; c += 64) {
      const double* h = H + c * 21;
      md = fmax(md, fmax(fmax(fabs(xwg_load(h)), fabs(xwg_load(h + 6))), fmax(fmax(fabs(xwg_load(h + 11)), fabs(xwg_load(h + 15))), fmax(fabs(xwg_load(h + 18)), fabs(xwg_load(h + 20))))));
    }
    md = wave_max(md);
  }
  if (lane == 0) {
    S.currentChi = chi; S.iniChi = chi;
    if (S.it == 0) {
      md = fmax(md, __longlong_as_double((long long)S.maxdiag_bits));
      S.lambda = 1e-5 * md; S.ni = 2.0; S.nBad = 0;
    }
    S.q = 0; S.need_lin = 0;
  }
}

// Hpp / b_p = sum over the linearise workgroups' partials, fixed order; the window's LAST workgroup to finish then runs the LM iteration
// head (it was a launch of its own until round 4: one dependent launch less per linearisation).  grid (ceil(n_free_max*27 / 256), nW)
__global__ __launch_bounds__(256) void ba_hpp_reduce_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  __shared__ int is_last;
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN || !S.need_lin) return;             // (uniform over the window's workgroups: need_lin is only cleared behind the ticket)
  const int nacc = W.n_free * 27;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nacc) {
    const double* src = A.hpp_part + W.hpart_off + i;
    const int nb = W.big ? 1 : W.nl_pt + W.nl_ln;          // big: the linearise kernels added into one row directly
    // a small batch has ~140 partial rows per window and every row sits in another XCD's L2: eight independent loads in flight,
    // summed in row order (bit-identical to the plain loop)
    double v = 0.0;
    int b = 0;
    for (; b + 8 <= nb; b += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = src[(size_t)(b + u) * nacc];
#pragma unroll
      for (int u = 0; u < 8; u++) v += t[u];
    }
    for (; b < nb; b++) v += src[(size_t)b * nacc];
    const int c = i / 27, k = i - c * 27;
    if (k < 21) xwg_store(&A.Hpp[((size_t)W.hpp_off + c) * 21 + k], v); else A.bp[((size_t)W.hpp_off + c) * 6 + (k - 21)] = v;      // (the LM head reads Hpp's diagonal)
  }
  xwg_stores_done();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&S.ticket_lin, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (is_last && threadIdx.x < 64) {
    if (threadIdx.x == 0) S.ticket_lin = 0;
    ba_begin_body(A, W, S, threadIdx.x);
  }
}

// ================================================================== Schur complement
// (1) ba_schur_items: one wavefront per chunk of landmarks that share one set of free cameras.  With
//     Hll + lambda I = L L^T (setLambda + the inverse of block_solver.hpp:391 folded into a Cholesky factor) and Z_a = W_a L^-T,
//     the reference's  Y_a W_b^T = W_a (Hll + lambda I)^-1 W_b^T  (block_solver.hpp:395-428) is Z_a Z_b^T and
//     Y_a b_l = Z_a (L^-1 b_l).  The chunk is swept in sub-batches of up to 64/k landmarks that are staged through LDS:
//       stage    lane (landmark, slot) rebuilds its 6xD Hpl block (points: closed form from pose, point and weight; lines: the
//                stored block), factors Hll + lambda I, and leaves Z (and t = L^-1 b_l, once per landmark) in LDS.  Its global
//                loads run ahead: indices two sub-batches ahead, landmark data one.
//       product  lane (slot pair (a,b), interleave) keeps the WHOLE 6x6 product Z_a Z_b^T of its pair in 36 registers over the
//                chunk: 36 LDS doubles (16-byte reads) per 36*D FMAs.  (An fp64 FMA of a wavefront takes 4 cycles on one of the four
//                SIMDs, the CU's single LDS pipe moves 32 doubles per cycle with ds_read_b128 and 16 with ds_read2_b64: the 2-row
//                blocks this replaces read 24 doubles per 12*D FMAs through ds_read2_b64 and ran at the speed of the LDS pipe.)
//     The chunk's partials are stored once (plain stores).
// (2) ba_schur_reduce: S = blockdiag(Hpp + lambda I) - sum of partials, bschur = b_p - sum c, through a host-built CSR
//     (lower block -> contributing partials).  Only the LOWER block triangle is produced.  No atomics, fixed order.
constexpr int kSchurWideK = 64;            // more free observations of one landmark than this: schur_chunk_wide

// sub-batch geometry shared by host (LDS size) and device: landmarks per sub-batch, LDS doubles
__host__ __device__ inline int schur_nb(int k) {
  int NB = 64 / k; if (NB < 1) NB = 1;
  const int np = k * (k + 1) / 2, units = np < 64 ? np : 64, q = 64 / units;
  if (NB >= q) NB -= NB % q;               // every interleave lane gets the same number of landmarks
  return NB;
}
__host__ __device__ inline int schur_lds_doubles(int k, int D) {
  const int WS = (D == 3) ? 18 : 26;
  return k > kSchurWideK ? k * WS + D + 1 : ((schur_nb(k) * (k * WS + D) + 1) & ~1);
}

// factor + stage one (landmark, slot): Z = W L^-T into zl (6 x D), t = L^-1 b_l into tl
template <int D>
__device__ __forceinline__ void schur_stage_one(bool a, const double* v, double lambda, const double* w, double* zl, double* tl, bool write_t) {
  constexpr int HU = (D == 3) ? 6 : 10;
  double L[D * (D + 1) / 2], idg[D];
  if (!a) {                                                 // inactive landmark (rare): contributes nothing - explicit zeros, so that a
#pragma unroll                                              // non-finite stale block cannot turn into 0 * NaN
    for (int i = 0; i < 6 * D; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(0.0, 0.0);
    if (write_t) {
#pragma unroll
      for (int c = 0; c < D; c++) tl[c] = 0.0;
    }
    return;
  }
  chol_packed<D>(v, lambda, L, idg);
  double z[6 * D];
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < D; c++) {
      double sacc = w[r * D + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= z[r * D + m] * L[c * (c + 1) / 2 + m];
      z[r * D + c] = sacc * idg[c];
    }
#pragma unroll
  for (int i = 0; i < 6 * D; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(z[i], z[i + 1]);
  if (write_t) {
    double t[D];
#pragma unroll
    for (int c = 0; c < D; c++) {
      double sacc = v[HU + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= t[m] * L[c * (c + 1) / 2 + m];
      t[c] = sacc * idg[c];
    }
#pragma unroll
    for (int c = 0; c < D; c++) tl[c] = t[c];
  }
}

// The same for a POINT edge, from the structure of its block (round 5).  W = ws Jc^T Jp = [ [Xc]x G ; G ] with G = ws (A^T A) R
// (point_hpl_closed_iz), hence Z = W L^-T = [ [Xc]x H ; H ] with H = G L^-T: the triangular solve runs on three rows instead of six and the
// three cross products act on H instead of on G - 18 multiply-adds less per (landmark, slot) of the ~200 a staging lane spends.
// `G`: rows g0, g1, g2 of G (G[r * 3 + j]); Xc: the point in the camera frame.
__device__ __forceinline__ void schur_stage_point(bool a, const double* v, double lambda, const double* G, const Vec3& Xc, double* zl, double* tl, bool write_t) {
  double L[6], idg[3];
  if (!a) {
#pragma unroll
    for (int i = 0; i < 18; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(0.0, 0.0);
    if (write_t) { tl[0] = 0.0; tl[1] = 0.0; tl[2] = 0.0; }
    return;
  }
  chol_packed<3>(v, lambda, L, idg);
  double z[18];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double sacc = G[r * 3 + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= z[(3 + r) * 3 + m] * L[c * (c + 1) / 2 + m];
      z[(3 + r) * 3 + c] = sacc * idg[c];
    }
#pragma unroll
  for (int c = 0; c < 3; c++) {                              // Xc x (column c of H)
    const double h0 = z[9 + c], h1 = z[12 + c], h2 = z[15 + c];
    z[c] = Xc.y * h2 - Xc.z * h1;
    z[3 + c] = Xc.z * h0 - Xc.x * h2;
    z[6 + c] = Xc.x * h1 - Xc.y * h0;
  }
#pragma unroll
  for (int i = 0; i < 18; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(z[i], z[i + 1]);
  if (write_t) {
    double t[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double sacc = v[6 + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= t[m] * L[c * (c + 1) / 2 + m];
      t[c] = sacc * idg[c];
    }
    tl[0] = t[0]; tl[1] = t[1]; tl[2] = t[2];
  }
}

// H form of a staged point block (round 6): only H (3 x 3, row major) and Xc go to LDS - 12 doubles instead of the 18 of Z = [ [Xc]x H ; H ];
// the product loop rebuilds the four 3 x 3 blocks of Z_a Z_b^T from M = H_a H_b^T (schur_chunk_wave).  kSchurHS: LDS stride in doubles.
#ifndef LLD_SCHUR_ZFORM
#define LLD_SCHUR_HFORM 1
#endif
constexpr int kSchurHS = 14;
__device__ __forceinline__ void schur_stage_point_h(bool a, const double* v, double lambda, const double* G, const Vec3& Xc, double* zl, double* tl, bool write_t) {
  double L[6], idg[3];
  if (!a) {
#pragma unroll
    for (int i = 0; i < 12; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(0.0, 0.0);
    if (write_t) { tl[0] = 0.0; tl[1] = 0.0; tl[2] = 0.0; }
    return;
  }
  chol_packed<3>(v, lambda, L, idg);
  double h[12];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double sacc = G[r * 3 + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= h[r * 3 + m] * L[c * (c + 1) / 2 + m];
      h[r * 3 + c] = sacc * idg[c];
    }
  h[9] = Xc.x; h[10] = Xc.y; h[11] = Xc.z;
#pragma unroll
  for (int i = 0; i < 12; i += 2) *reinterpret_cast<double2*>(zl + i) = make_double2(h[i], h[i + 1]);
  if (write_t) {
    double t[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double sacc = v[6 + c];
#pragma unroll
      for (int m = 0; m < c; m++) sacc -= t[m] * L[c * (c + 1) / 2 + m];
      t[c] = sacc * idg[c];
    }
    tl[0] = t[0]; tl[1] = t[1]; tl[2] = t[2];
  }
}

// A landmark with more than kSchurWideK free observations (global BA of a long track): no pipelining and no register accumulators -
// the whole workgroup stages the landmark's k blocks, then thread t adds the products of the pairs t, t + 256, ... into the chunk's
// partials in HBM (one writer per pair, landmarks in order: deterministic).  Such chunks hold a handful of landmarks.
template <int D>
__device__ __forceinline__ void schur_chunk_wide(const BAArrays& A, const BAWin& W, const SChunk& C, double lambda, int cur, double* lds) {
  constexpr int VN = (D == 3) ? 9 : 14, WN = 6 * D, WS = (D == 3) ? 18 : 26;
  const double* __restrict__ Vbase = (D == 3) ? A.pt_V : A.ln_V;
  const uint8_t* __restrict__ act = (D == 3) ? A.pt_active : A.ln_active;
  const int k = C.k, np = k * (k + 1) / 2;
  double* Zl = lds; double* tl = lds + ((k * WS + 1) & ~1);
  for (int t0 = 0; t0 < C.n_lm; t0++) {
    const int g = A.sg_lm[C.lm_off + t0];
    const bool a = act[g] != 0;
    double v[VN];
#pragma unroll
    for (int i = 0; i < VN; i++) v[i] = Vbase[(size_t)g * VN + i];
    __syncthreads();
    for (int sl = threadIdx.x; sl < k; sl += kSchurWideThreads) {
      const int id = A.sg_tab[C.tab_off + (size_t)t0 * k + sl];
      double w[WN];
      if constexpr (D == 3) {
        const Pose Ts = load_cam(A, cur, W.cam_off + A.sg_cams[C.cams_off + sl]);
        point_hpl_closed(W.cam, Ts, quat_rotation(Ts.q), load_pt(A, cur, g), signbit(A.pe_ws[id]), fabs(A.pe_ws[id]), w);
      } else {
#pragma unroll
        for (int i = 0; i < WN; i++) w[i] = A.lo_W[(size_t)id * WN + i];
      }
      schur_stage_one<D>(a, v, lambda, w, Zl + sl * WS, tl, sl == 0);
    }
    __syncthreads();
    for (int pr = threadIdx.x; pr < np; pr += kSchurWideThreads) {
      int sa = 0, rem = pr;
      while (rem >= k - sa) { rem -= k - sa; sa++; }
      const int sb = sa + rem;
      const double* za = Zl + sa * WS; const double* zb = Zl + sb * WS;
      double* dst = A.sp_part + (size_t)(C.part_off + pr) * 36;
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) {
          double sacc = t0 == 0 ? 0.0 : dst[r * 6 + c];
#pragma unroll
          for (int m = 0; m < D; m++) sacc = fma(za[r * D + m], zb[c * D + m], sacc);
          dst[r * 6 + c] = sacc;
        }
      if (sa == sb) {
        double* cd = A.sp_cpart + (size_t)(C.cpart_off + sa) * 6;
        for (int r = 0; r < 6; r++) {
          double sacc = t0 == 0 ? 0.0 : cd[r];
#pragma unroll
          for (int m = 0; m < D; m++) sacc = fma(za[r * D + m], tl[m], sacc);
          cd[r] = sacc;
        }
      }
    }
  }
}

template <int D>
__device__ __forceinline__ void schur_chunk_wave(const BAArrays& A, const BAWin& W, const SChunk& C, double lambda, int cur, double* lds) {
  constexpr int VN = (D == 3) ? 9 : 14, WN = 6 * D;
  // LDS stride of a staged 6xD block: 144 B for points (conflict-free as is); 192 B would put slots 0 and 4 of a line on the
  // same banks, so line blocks are padded to 208 B (still 16-B aligned)
#ifdef LLD_SCHUR_HFORM
  constexpr bool kH = (D == 3);
  constexpr int WS = (D == 3) ? kSchurHS : 26;
#else
  constexpr bool kH = false;
  constexpr int WS = (D == 3) ? 18 : 26;
#endif
  const double* __restrict__ Vbase = (D == 3) ? A.pt_V : A.ln_V;
  const uint8_t* __restrict__ act = (D == 3) ? A.pt_active : A.ln_active;
  const int lane = threadIdx.x;
  const int k = C.k, np = k * (k + 1) / 2;
  const int NB = schur_nb(k);
  double* Zl = lds;
  double* tl = lds + NB * k * WS;
  const int* __restrict__ lm = A.sg_lm + C.lm_off;
  const int* __restrict__ tab = A.sg_tab + C.tab_off;
  // stage lane <-> (landmark ej, slot esl) of a sub-batch
  const int ej = lane / k, esl = lane - ej * k;
  const bool stager = lane < NB * k;
  // the lanes of the per-slot vector pass: slot cslot, interleave ci of cq
  const int cq = (64 / k) < NB ? (64 / k) : NB;
  const int ci = lane / k, cslot = lane - ci * k;
  const bool con = ci < cq;
  double cacc[6];
#pragma unroll
  for (int i = 0; i < 6; i++) cacc[i] = 0.0;
  for (int pass0 = 0; pass0 < np; pass0 += 64) {
    const int units = (np - pass0) < 64 ? (np - pass0) : 64;        // slot pairs of this pass
    const int q = 64 / units;                                       // landmarks worked on at a time
    const int pl = lane % units, qq = lane / units;
    const bool on = qq < q;
    int sa = 0, rem = pass0 + pl;
    while (rem >= k - sa) { rem -= k - sa; sa++; }
    const int sb = sa + rem;
    const bool diag = sa == sb;
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0.0;
    // The staging runs one sub-batch AHEAD of its loads' latency: the camera pose of a lane's slot is loop-invariant, the landmark /
    // edge indices are fetched two sub-batches ahead and the landmark data one ahead, so the two dependent HBM/L2 round trips
    // (index -> data) of a sub-batch overlap the block products of the previous one.  Bytes stay as loaded (a_n, fl_n): turning
    // them into flags where they are fetched would wait for the loads right there.
    Pose T; Mat3 Rt;
    if constexpr (D == 3) { if (stager) { T = load_cam(A, cur, W.cam_off + A.sg_cams[C.cams_off + esl]); Rt = quat_rotation(T.q); } }
    int g_n = 0, id_n = 0, g_nn = 0, id_nn = 0, a_n = 0;
    double v_n[VN];
    double ws_n = 0.0; Vec3 X_n = vec3(0, 0, 1);
    // Every lane issues every load, with the landmark index clamped into the chunk (a lane past the end or outside the staging range
    // re-reads the last landmark and never uses it): loads under a divergent `if` cannot be counted and the compiler waits for
    // vmcnt(0) at the first use of any loaded value (14 such waits in this kernel before, 7 now; the loads are a whole sub-batch of
    // arithmetic ahead either way, so the launch time did not move).
    auto fetch_idx = [&](int t0, int& g, int& id) {
      const int tj = min(t0 + ej, C.n_lm - 1);
      g = lm[tj]; id = tab[(size_t)tj * k + esl];
    };
    auto fetch_data = [&](int t0, int g, int id) {
      a_n = act[g];
      const double* V = Vbase + (size_t)g * VN;
#pragma unroll
      for (int i = 0; i < VN; i++) v_n[i] = V[i];
      if constexpr (D == 3) { ws_n = A.pe_ws[id]; X_n = load_pt(A, cur, g); }      // (the stereo flag rides in the weight's sign: no gather of the edge's flag byte)      // (the stereo flag rides in the weight's sign: no gather of the edge's flag byte)
    };
    fetch_idx(0, g_n, id_n);
    fetch_idx(NB, g_nn, id_nn);
    fetch_data(0, g_n, id_n);
    for (int t0 = 0; t0 < C.n_lm; t0 += NB) {
      const int nb = (C.n_lm - t0) < NB ? (C.n_lm - t0) : NB;
      // this sub-batch's operands (arrived while the previous products ran) -> locals; then put the next loads in flight
      double w[WN], v[VN];
#pragma unroll
      for (int i = 0; i < VN; i++) v[i] = v_n[i];
      const double ws = ws_n; const Vec3 X = X_n; const int a_raw = a_n;
      const int id_cur = id_n;
      g_n = g_nn; id_n = id_nn;
      fetch_data(t0 + NB, g_n, id_n);
      fetch_idx(t0 + 2 * NB, g_nn, id_nn);
      if constexpr (D == 4) {
        // line observation: the summed 6x4 block was stored by the linearisation (fetched here, not a sub-batch ahead: holding two
        // of them would halve the occupancy); by every lane, like the prefetch (id_cur is a valid observation in all of them)
        const double* Wg = A.lo_W + (size_t)id_cur * WN;
#pragma unroll
        for (int i = 0; i < WN; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(Wg + i); w[i] = t2.x; w[i + 1] = t2.y; }
      }
      __syncthreads();                                      // the previous sub-batch has been consumed
      if (stager && ej < nb) {
        // point edge: the Hpl block is a function of the linearisation-point pose, point and one weight
        if constexpr (D == 3) {
          const Vec3 Xc = mat_mul(Rt, X) + T.t;
          double G[9];
          point_g_closed_iz(W.cam, Xc, rcp_nr(Xc.z), Rt, signbit(ws), fabs(ws), G);
          if constexpr (kH) schur_stage_point_h(a_raw != 0, v, lambda, G, Xc, Zl + lane * WS, tl + ej * D, esl == 0);
          else schur_stage_point(a_raw != 0, v, lambda, G, Xc, Zl + lane * WS, tl + ej * D, esl == 0);
        } else schur_stage_one<D>(a_raw != 0, v, lambda, w, Zl + lane * WS, tl + ej * D, esl == 0);
      }
      __syncthreads();
      if (on) {
        for (int j = qq; j < nb; j += q) {
          const double* za = Zl + (j * k + sa) * WS;
          const double* zb = Zl + (j * k + sb) * WS;
          if constexpr (kH) {
            // Z_a Z_b^T = [ a M b^T, a M ; M b^T, M ] with M = H_a H_b^T, a = [Xa]x, b = [Xb]x: 27 + 9 + 18 + 9 + 18 + 18 = 99 operations against the
            // 108 FMAs of the full 6x3 . 3x6 product, and 24 LDS doubles per (pair, landmark) instead of 36.  (Reading the next landmark's operands
            // ahead of this one's products by hand measured 1 % slower: tools/experiments/r06_schur_hform_prefetch.patch.)
            double ha[12], hb[12];
#pragma unroll
            for (int i = 0; i < 12; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(zb + i); hb[i] = t2.x; hb[i + 1] = t2.y; }
#pragma unroll
            for (int i = 0; i < 12; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(za + i); ha[i] = t2.x; ha[i + 1] = t2.y; }
            double M[9], P[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int c = 0; c < 3; c++) M[r * 3 + c] = fma(ha[r * 3 + 2], hb[c * 3 + 2], fma(ha[r * 3 + 1], hb[c * 3 + 1], ha[r * 3] * hb[c * 3]));
            const double xa = ha[9], ya = ha[10], za_ = ha[11], xb = hb[9], yb = hb[10], zb_ = hb[11];
#pragma unroll
            for (int r = 0; r < 3; r++) {                          // P = M b^T
              const double m0 = M[r * 3], m1 = M[r * 3 + 1], m2 = M[r * 3 + 2];
              P[r * 3] = fma(yb, m2, -(zb_ * m1)); P[r * 3 + 1] = fma(zb_, m0, -(xb * m2)); P[r * 3 + 2] = fma(xb, m1, -(yb * m0));
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
              acc[18 + 3 + c] += M[c]; acc[24 + 3 + c] += M[3 + c]; acc[30 + 3 + c] += M[6 + c];        // lower right: M
              acc[18 + c] += P[c]; acc[24 + c] += P[3 + c]; acc[30 + c] += P[6 + c];                    // lower left: M b^T
              // upper right: a M, upper left: a P   (row 0 = ya * row 2 - za * row 1, row 1 = za * row 0 - xa * row 2, row 2 = xa * row 1 - ya * row 0)
              acc[3 + c] = fma(ya, M[6 + c], fma(-za_, M[3 + c], acc[3 + c]));
              acc[6 + 3 + c] = fma(za_, M[c], fma(-xa, M[6 + c], acc[6 + 3 + c]));
              acc[12 + 3 + c] = fma(xa, M[3 + c], fma(-ya, M[c], acc[12 + 3 + c]));
              acc[c] = fma(ya, P[6 + c], fma(-za_, P[3 + c], acc[c]));
              acc[6 + c] = fma(za_, P[c], fma(-xa, P[6 + c], acc[6 + c]));
              acc[12 + c] = fma(xa, P[3 + c], fma(-ya, P[c], acc[12 + c]));
            }
            continue;
          }
          double b[WN];
#pragma unroll
          for (int i = 0; i < WN; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(zb + i); b[i] = t2.x; b[i + 1] = t2.y; }
#pragma unroll
          for (int rp = 0; rp < 3; rp++) {                   // two rows of Z_a at a time: 16-byte LDS reads
            double a2[2 * D];
#pragma unroll
            for (int i = 0; i < 2 * D; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(za + rp * 2 * D + i); a2[i] = t2.x; a2[i + 1] = t2.y; }
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
              const int r = 2 * rp + rr;
#pragma unroll
              for (int cc = 0; cc < 6; cc++) {
                double s0 = acc[r * 6 + cc];
#pragma unroll
                for (int m = 0; m < D; m++) s0 = fma(a2[rr * D + m], b[cc * D + m], s0);
                acc[r * 6 + cc] = s0;
              }
            }
          }
        }
      }
      // Y_a b_l = Z_a (L^-1 b_l), one 6-vector per slot: lane <-> (slot, interleave) in a pass of its own (round 5).  Inside the product loop
      // every lane carried these 6 x D multiply-adds per landmark although only the k diagonal pairs of the k (k + 1) / 2 keep them.
      if (pass0 == 0 && con) {
        for (int j = ci; j < nb; j += cq) {
          const double* za = Zl + (j * k + cslot) * WS;
          if constexpr (kH) {                                      // Z t = [ Xc x (H t) ; H t ]
            double ha[12];
#pragma unroll
            for (int i = 0; i < 12; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(za + i); ha[i] = t2.x; ha[i + 1] = t2.y; }
            const double t0_ = tl[j * 3], t1_ = tl[j * 3 + 1], t2_ = tl[j * 3 + 2];
            const double h0 = fma(ha[2], t2_, fma(ha[1], t1_, ha[0] * t0_)), h1 = fma(ha[5], t2_, fma(ha[4], t1_, ha[3] * t0_)), h2 = fma(ha[8], t2_, fma(ha[7], t1_, ha[6] * t0_));
            cacc[3] += h0; cacc[4] += h1; cacc[5] += h2;
            cacc[0] = fma(ha[10], h2, fma(-ha[11], h1, cacc[0]));
            cacc[1] = fma(ha[11], h0, fma(-ha[9], h2, cacc[1]));
            cacc[2] = fma(ha[9], h1, fma(-ha[10], h0, cacc[2]));
            continue;
          }
          double a[WN], tv[D];
#pragma unroll
          for (int i = 0; i < WN; i += 2) { const double2 t2 = *reinterpret_cast<const double2*>(za + i); a[i] = t2.x; a[i + 1] = t2.y; }
#pragma unroll
          for (int m = 0; m < D; m++) tv[m] = tl[j * D + m];
#pragma unroll
          for (int r = 0; r < 6; r++) {
            double s1 = cacc[r];
#pragma unroll
            for (int m = 0; m < D; m++) s1 = fma(a[r * D + m], tv[m], s1);
            cacc[r] = s1;
          }
        }
      }
    }
    // sum over the interleave (lanes qq * units + pl): a fixed shuffle tree, result in the lanes qq == 0
    for (int sft = 1; sft < q; sft <<= 1) {
      const bool take = (qq % (2 * sft)) == 0 && qq + sft < q;
#pragma unroll
      for (int i = 0; i < 36; i++) { const double o = __shfl_down(acc[i], sft * units); if (take) acc[i] += o; }
    }
    if (on && qq == 0) {
      // plain stores of the chunk's partial products; ba_schur_reduce sums them into S in a fixed order (no atomics)
      double* dst = A.sp_part + (size_t)(C.part_off + pass0 + pl) * 36;
#pragma unroll
      for (int i = 0; i < 36; i += 2) *reinterpret_cast<double2*>(dst + i) = make_double2(acc[i], acc[i + 1]);
    }
  }
  // the slot vectors: sum over the interleave (lanes cslot + k * ci), a fixed shuffle tree, result in the lanes ci == 0
  for (int sft = 1; sft < cq; sft <<= 1) {
    const bool take = (ci % (2 * sft)) == 0 && ci + sft < cq;
#pragma unroll
    for (int i = 0; i < 6; i++) { const double o = __shfl_down(cacc[i], sft * k); if (take) cacc[i] += o; }
  }
  if (ci == 0) {                                            // (lanes 0 .. k - 1)
    double* cd = A.sp_cpart + (size_t)(C.cpart_off + cslot) * 6;
#pragma unroll
    for (int i = 0; i < 6; i += 2) *reinterpret_cast<double2*>(cd + i) = make_double2(cacc[i], cacc[i + 1]);
  }
}

// grid (max chunks of this landmark type, nW), block 64 = one wavefront per chunk; dynamic LDS sized by the host.
template <int D>
__global__ __launch_bounds__(kSchurThreads) void ba_schur_items_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  const int first = (D == 3) ? W.item_off : W.item_off + W.n_items_pt;
  const int count = (D == 3) ? W.n_items_pt : W.n_items - W.n_items_pt;
  if ((int)blockIdx.x >= count) return;
  const SChunk C = A.sg_chunks[first + blockIdx.x];
  if (C.k <= kSchurWideK) schur_chunk_wave<D>(A, W, C, S.lambda, S.cur, lds);   // (wider chunks: ba_schur_wide_kernel)
}

// Point and line chunks in ONE launch: grid (nW * (n_pt_blocks + max line chunks)), see the dispatch order below; chunks are stored heaviest first (stage_chunks).  The line chunks fill the tail of the point
// chunks instead of waiting for it - for a single window the two kernels were two dependent 17 us launches on an otherwise idle GPU.
__global__ __launch_bounds__(kSchurThreads) void ba_schur_items_both_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st, int n_pt_blocks, int win_tile, int nrow, int nch) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // dispatch order (x fastest) -> (window row, chunk): tiles of win_tile windows, inside a tile the chunk index runs slowest.  Workgroups go to
  // the eight XCDs round-robin, so with a tile of 8 (any multiple of 8) ALL chunks of a window run on ONE XCD, close together in time: the sectors
  // of the landmark arrays that several of them touch (a 128-byte line of positions or V serves landmarks of different camera sets) are fetched
  // into one L2 once instead of into up to eight.  FETCH_SIZE of this kernel per launch of 256 windows, tools/experiments/exp_schur_tile.sh:
  // windows one after the other 671 k KiB, tile 8: 360 k, 16: 373 k, 32: 389 k, 64: 504 k, all 256: 766 k; the time is within noise from 8 to 32
  // and 3 - 4 % better than either extreme.  A window's chunks are stored heaviest first, so every tile drains on its light chunks.
  const int lin = (int)blockIdx.x;                             // one-dimensional grid of nrow x nch workgroups (a map of thousands of keyframes has more chunks than gridDim.y may be)
  const int tile = lin / (win_tile * nch), row0 = tile * win_tile, tsz = min(win_tile, nrow - row0), rem = lin - tile * win_tile * nch;
  const int ci = rem / tsz, row = row0 + rem - ci * tsz;
  const int wrow = LLD_ROW_WINDOW(A, st, row);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  if (ci < n_pt_blocks) {
    if (ci >= W.n_items_pt) return;
    const SChunk C = A.sg_chunks[W.item_off + ci];
    if (C.k <= kSchurWideK) schur_chunk_wave<3>(A, W, C, S.lambda, S.cur, lds);
  } else {
    const int i = ci - n_pt_blocks;
    if (i >= W.n_items - W.n_items_pt) return;
    const SChunk C = A.sg_chunks[W.item_off + W.n_items_pt + i];
    if (C.k <= kSchurWideK) schur_chunk_wave<4>(A, W, C, S.lambda, S.cur, lds);
  }
}

// The chunks the kernels above skip (a landmark with more than kSchurWideK free observations); launched only for batches that
// have one.  grid (max chunks, nW) over all chunks of a window, block kSchurThreads.
__global__ __launch_bounds__(kSchurWideThreads) void ba_schur_wide_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN || (int)blockIdx.x >= W.n_items) return;
  const SChunk C = A.sg_chunks[W.item_off + blockIdx.x];
  if (C.k <= kSchurWideK) return;
  if (C.D == 3) schur_chunk_wide<3>(A, W, C, S.lambda, S.cur, lds);
  else schur_chunk_wide<4>(A, W, C, S.lambda, S.cur, lds);
}

// grid (ceil(nblk_max * 6 / 256) + 2, nW): lane <-> one row of one lower 6x6 block of S.  S_ij = [i == j](Hpp_i + lambda I)
// - sum of the chunk partials listed for the block (fixed order -> deterministic), written once with a plain store.
// blk_src = part_index * 4 + mode; mode 0: partial is Y_a W_b^T for cameras a < b -> transposed into block (b, a);
// mode 1: same observation on the diagonal; mode 2: two observations by one camera -> P + P^T.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void ba_schur_reduce_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  const BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  const int nf = W.n_free, n = 6 * nf, nblk = nf * (nf + 1) / 2;
  // the LAST TWO workgroups of a window do the right-hand side (half of the rows each), the others the blocks: for a single window both are
  // chains of dependent cross-XCD loads and must not queue behind one another
  const int rhs_part = (int)gridDim.x - 1 - (int)blockIdx.x;   // 0 / 1: a right-hand-side workgroup
  const bool rhs_block = rhs_part < 2;
  // lane <-> one row of one lower 6x6 block, blocks in blk_perm order (longest partial lists first: the lanes of one wavefront walk lists of
  // one length).  The wavefronts of that order are dealt round-robin to the window's workgroups, so that the few long-list wavefronts of a
  // window pull their partials through different CUs (a single window: all of them in one workgroup cost 5 us per launch).
  const int nslot6 = (A.s_skip_empty ? W.n_blk_nz : nblk) * 6, nwg = ((nslot6 + 63) / 64 + 3) / 4;
  const int idx = (((int)threadIdx.x >> 6) * nwg + (int)blockIdx.x) * 64 + ((int)threadIdx.x & 63);
  if (!rhs_block && (int)blockIdx.x < nwg && idx < nslot6) {
    const int slot = idx / 6, rr = idx - slot * 6;
    const int blk = A.blk_perm[W.blk_csr_off + slot];
    int i = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= blk) i++;
    while (i * (i + 1) / 2 > blk) i--;
    const int j = blk - i * (i + 1) / 2;
    double v[6] = {0, 0, 0, 0, 0, 0};
    if (i == j) {
      const double* Hp = A.Hpp + ((size_t)W.hpp_off + i) * 21;
#pragma unroll
      for (int cc = 0; cc < 6; cc++) {
        const int lo = rr < cc ? rr : cc, hi = rr < cc ? cc : rr;
        v[cc] = Hp[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
      }
      v[rr] += S.lambda;
    }
    const int* bst = A.blk_start + W.blk_csr_off;
    const int q0 = bst[blk], q1 = bst[blk + 1];
    // A diagonal block collects one partial from every chunk that sees its camera (~50), and each list entry is a chain of two
    // dependent loads (index -> partial): four entries are kept in flight and the three modes are folded into weights (row part
    // w_r, column part w_c in {0,1}) so that the loads do not sit behind a branch.  0*x + y is exact and mode 2 keeps its
    // P + P^T order, so the result is bit-identical to the entry-by-entry loop.
    // the indices of the next four entries are fetched while the partials of the current four are in flight (two dependent
    // round trips per group otherwise; for a single window every one of them leaves the XCD)
    int nxt[4];
#pragma unroll
    for (int uu = 0; uu < 4; uu++) nxt[uu] = (q0 + uu < q1) ? A.blk_src[q0 + uu] : -1;
    for (int q = q0; q < q1; q += 4) {
      int src[4];
#pragma unroll
      for (int uu = 0; uu < 4; uu++) src[uu] = nxt[uu];
      // ONE strided load per entry: the lane's row of the partial (modes 1, 2) or its column (mode 0: the transposed block) - round 4;
      // before, every entry fetched both (12 doubles, 96 registers in flight, three wavefronts per SIMD for a kernel that only waits)
      double pv[4][6];
#pragma unroll
      for (int uu = 0; uu < 4; uu++) {
        const int sidx = src[uu] < 0 ? 0 : src[uu];
        const bool col = (sidx & 3) == 0;
        const double* P = A.sp_part + (size_t)(sidx >> 2) * 36 + (col ? rr : rr * 6);
        const int stp = col ? 6 : 1;
#pragma unroll
        for (int cc = 0; cc < 6; cc++) pv[uu][cc] = P[cc * stp];
      }
#pragma unroll
      for (int uu = 0; uu < 4; uu++) nxt[uu] = (q + 4 + uu < q1) ? A.blk_src[q + 4 + uu] : -1;
#pragma unroll
      for (int uu = 0; uu < 4; uu++) {
        if (src[uu] >= 0) {
#pragma unroll
          for (int cc = 0; cc < 6; cc++) v[cc] -= pv[uu][cc];
          if ((src[uu] & 3) == 2) {                        // two observations of one landmark by the same camera: P + P^T (no local-BA window has one)
            const double* P = A.sp_part + (size_t)(src[uu] >> 2) * 36;
#pragma unroll
            for (int cc = 0; cc < 6; cc++) v[cc] -= P[cc * 6 + rr];
          }
        }
      }
    }
    double* dst = A.S + W.S_off + (size_t)(6 * i + rr) * n + 6 * j;
#pragma unroll
    for (int cc = 0; cc < 6; cc += 2) *reinterpret_cast<double2*>(dst + cc) = make_double2(v[cc], v[cc + 1]);
  }
  if (rhs_block) {
    const int* cst = A.cam_start + W.cam_csr_off;
    const int half = (n + 1) / 2, t_end = rhs_part == 0 ? half : n;
    for (int t = (rhs_part == 0 ? 0 : half) + (int)threadIdx.x; t < t_end; t += 256) {
      const int c = t / 6, r = t - c * 6;
      double v = A.bp[(size_t)W.hpp_off * 6 + t];
      // a camera appears in ~50 chunks and every list entry is two dependent loads (index -> partial): sixteen entries are kept in
      // flight; they are still subtracted one by one in list order, so the sum is bit-identical to the plain loop
      const int q0 = cst[c], q1 = cst[c + 1];
      for (int q = q0; q < q1; q += 16) {
        int src[16]; double pv[16];
#pragma unroll
        for (int uu = 0; uu < 16; uu++) src[uu] = (q + uu < q1) ? A.cam_src[q + uu] : -1;
#pragma unroll
        for (int uu = 0; uu < 16; uu++) pv[uu] = (src[uu] >= 0) ? A.sp_cpart[(size_t)src[uu] * 6 + r] : 0.0;
#pragma unroll
        for (int uu = 0; uu < 16; uu++) if (src[uu] >= 0) v -= pv[uu];
      }
      A.bschur[W.x_off + t] = v;
    }
  }
}

// PCG only: mirror the lower block triangle into the upper one (the column-wise matvec wants the full matrix).  grid (16, nW)
__global__ __launch_bounds__(256) void ba_symmetrize_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  const BAWin W = wins[blockIdx.y];
  if (st[blockIdx.y].phase != PH_RUN) return;
  const int n = 6 * W.n_free;
  double* Sg = A.S + W.S_off;
  // 64-bit element index: n * n passes 2^31 from 7724 free cameras on (n = 46344), and the limit is 8192
  const long long total = (long long)n * n, stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int row = (int)(i / n), col = (int)(i - (long long)row * n);
    if (col / 6 > row / 6) Sg[i] = Sg[(size_t)col * n + row];
  }
}

// Shared tail of the reduced-system solvers: publish x_p, apply VertexSE3Expmap::oplusImpl to the free cameras (trial
// buffer), leave sum x (lambda x + b) of the camera part for computeScale (optimization_algorithm_levenberg.cpp:182-189).
__device__ __forceinline__ void solve_epilogue(const BAArrays& A, const BAWin& W, BAState& S, const double* x, double* scratch, bool ok, int iters) {
  const int tid = threadIdx.x, nf = W.n_free, n = 6 * nf;
  const double lambda = S.lambda;
  const double* bpv = A.bp + (size_t)W.hpp_off * 6;
  double sc = 0.0;
  if (tid < n) { A.xp[W.x_off + tid] = x[tid]; sc = x[tid] * (lambda * x[tid] + bpv[tid]); }
  const double sc_t = block_sum(sc, scratch);
  const int cur = S.cur, nxt = cur ^ 1;
  if (tid < W.n_cams) {
    const Pose T = load_cam(A, cur, W.cam_off + tid);
    Pose Tn = T;
    if (tid < nf) Tn = pose_oplus(T, x + tid * 6);
    pose_store(Tn, A.cam_qt + ((size_t)nxt * A.NC + W.cam_off + tid) * 7);
  }
  if (tid == 0) { S.scale_cam = sc_t; S.pcg_ok = ok ? 1 : 0; S.pcg_iterations += iters; }
}

// ================================================================== PCG on the reduced camera system
// grid (nW); block kPcgThreads; dynamic LDS: 4n + kPcgThreads + nf*36 + 32 doubles.  Block-Jacobi preconditioner (inverse 6x6 diagonal
// blocks), fixed reduction trees, stops at |r|_M <= tol |b|_M.  On exit it applies VertexSE3Expmap::oplusImpl to the free
// cameras (trial buffer) and leaves sum x(lambda x + b) of the camera part for computeScale.
// ------------------------------------------------------------------ PCG across the whole GPU (few, larger windows)
// The same block-Jacobi PCG as ba_pcg_kernel, cut into kernels so that the matrix-vector product of ONE window runs on every CU:
//   init    (1 workgroup per window)  Minv, x = 0, r = b, z = Minv r, p = z, rz
//   matvec  (wavefront per row)       Sp = S p                                   -- S symmetric, stored in full
//   update  (1 workgroup per window)  alpha, x, r, z, rz, stop test, beta, p
//   final   (1 workgroup per window)  the common epilogue (x -> xp, scale, trial cameras)
// The host launches matvec/update pairs in chunks and looks at the `done` scalars between chunks; finished windows return at once.
__global__ __launch_bounds__(kPcgThreads) void ba_pcgm_init_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st, double tol) {
  __shared__ double scratch[32];
  const BAWin W = wins[blockIdx.x];
  const BAState& S = st[blockIdx.x];
  double* sc = A.pcg_sc + 8 * (size_t)W.win_index;
  if (S.phase != PH_RUN) { if (threadIdx.x == 0) sc[3] = 1.0; return; }
  const int nf = W.n_free, n = 6 * nf, tid = threadIdx.x;
  const double* Sg = A.S + W.S_off;
  double* Mi = A.pcg_mi + (size_t)W.hpp_off * 36;
  double* x = A.xp + W.x_off;
  double* r = A.pcg_vec + W.x_off; double* z = r + A.x_total; double* p = z + A.x_total;
  double* ok_s = scratch + 31;
  if (tid == 0) *ok_s = 1.0;
  __syncthreads();
  for (int cb = tid; cb < nf; cb += kPcgThreads) {
    double F[36], Fi[36];
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
#pragma unroll
      for (int c = 0; c < 6; c++) F[rr * 6 + c] = Sg[(size_t)(cb * 6 + rr) * n + cb * 6 + c];
    if (!spd_inverse<6>(F, Fi)) *ok_s = 0.0;
#pragma unroll
    for (int i = 0; i < 36; i++) Mi[cb * 36 + i] = Fi[i];
  }
  for (int i = tid; i < n; i += kPcgThreads) { x[i] = 0.0; r[i] = A.bschur[W.x_off + i]; }
  __syncthreads();                                       // Mi and r of this workgroup are visible to it
  double part = 0.0;
  for (int i = tid; i < n; i += kPcgThreads) {
    const int b = i / 6, rr = i - b * 6;
    double zv = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) zv += Mi[b * 36 + rr * 6 + c] * r[b * 6 + c];
    z[i] = zv; p[i] = zv;
    part += r[i] * zv;
  }
  const double rz0 = block_sum(part, scratch);
  if (tid == 0) {
    const bool ok = *ok_s != 0.0 && isfinite(rz0);
    sc[0] = rz0; sc[1] = tol * tol * rz0; sc[2] = 0.0; sc[3] = (ok && rz0 > 0.0) ? 0.0 : 1.0; sc[4] = ok ? 1.0 : 0.0;
  }
}

// grid (ceil(n_max / 4), nW), block 256: one wavefront per row of S
__global__ __launch_bounds__(256) void ba_pcgm_matvec_kernel(BAArrays A, const BAWin* __restrict__ wins) {
  const BAWin W = wins[blockIdx.y];
  const double* sc = A.pcg_sc + 8 * (size_t)W.win_index;
  if (sc[3] != 0.0) return;
  const int n = 6 * W.n_free, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // the wavefront index as a scalar: the task and every address built on it stay in SGPRs
  if (row >= n) return;
  const double* Sr = A.S + W.S_off + (size_t)row * n;
  const double* p = A.pcg_vec + 2 * A.x_total + W.x_off;
  double acc = 0.0;
  for (int c = lane; c < n; c += 64) acc = fma(Sr[c], p[c], acc);
  acc = wave_sum(acc);
  if (lane == 0) A.pcg_vec[3 * A.x_total + W.x_off + row] = acc;
}

__global__ __launch_bounds__(kPcgThreads) void ba_pcgm_update_kernel(BAArrays A, const BAWin* __restrict__ wins, int max_iter_param) {
  __shared__ double scratch[32];
  const BAWin W = wins[blockIdx.x];
  double* sc = A.pcg_sc + 8 * (size_t)W.win_index;
  if (sc[3] != 0.0) return;
  const int nf = W.n_free, n = 6 * nf, tid = threadIdx.x;
  const double* Mi = A.pcg_mi + (size_t)W.hpp_off * 36;
  double* x = A.xp + W.x_off;
  double* r = A.pcg_vec + W.x_off; double* z = r + A.x_total; double* p = z + A.x_total; const double* ap = p + A.x_total;
  const double rz = sc[0], stop = sc[1];
  const int max_iter = max_iter_param > 0 ? max_iter_param : 10 * n;
  double part = 0.0;
  for (int i = tid; i < n; i += kPcgThreads) part += p[i] * ap[i];
  const double pAp = block_sum(part, scratch);
  if (!(pAp > 0.0) || !isfinite(pAp)) { if (tid == 0) { sc[3] = 1.0; sc[4] = 0.0; } return; }
  const double alpha = rz / pAp;
  for (int i = tid; i < n; i += kPcgThreads) { x[i] += alpha * p[i]; r[i] -= alpha * ap[i]; }
  __syncthreads();
  part = 0.0;
  for (int i = tid; i < n; i += kPcgThreads) {
    const int b = i / 6, rr = i - b * 6;
    double zv = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) zv += Mi[b * 36 + rr * 6 + c] * r[b * 6 + c];
    z[i] = zv;
    part += r[i] * zv;
  }
  const double rz_new = block_sum(part, scratch);
  const double iters = sc[2] + 1.0;
  bool done = false, ok = true;
  if (!isfinite(rz_new)) { done = true; ok = false; }
  else if (rz_new <= stop || iters >= (double)max_iter) done = true;
  if (!done) { const double beta = rz_new / rz; for (int i = tid; i < n; i += kPcgThreads) p[i] = z[i] + beta * p[i]; }
  __syncthreads();                                        // every lane has read sc[] before lane 0 rewrites it
  if (tid == 0) { sc[0] = rz_new; sc[2] = iters; if (done) sc[3] = 1.0; if (!ok) sc[4] = 0.0; }
}

// the epilogue of solve_epilogue for any number of unknowns: scale of the step, trial cameras
__global__ __launch_bounds__(kPcgThreads) void ba_pcgm_final_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  __shared__ double scratch[32];
  const BAWin W = wins[blockIdx.x];
  BAState& S = st[blockIdx.x];
  if (S.phase != PH_RUN) return;
  const double* sc = A.pcg_sc + 8 * (size_t)W.win_index;
  const int tid = threadIdx.x, nf = W.n_free, n = 6 * nf;
  const double lambda = S.lambda;
  const double* x = A.xp + W.x_off;
  const double* bpv = A.bp + (size_t)W.hpp_off * 6;
  double part = 0.0;
  for (int i = tid; i < n; i += kPcgThreads) part += x[i] * (lambda * x[i] + bpv[i]);
  const double sc_t = block_sum(part, scratch);
  const int cur = S.cur, nxt = cur ^ 1;
  for (int c = tid; c < W.n_cams; c += kPcgThreads) {
    const Pose T = load_cam(A, cur, W.cam_off + c);
    Pose Tn = T;
    if (c < nf) Tn = pose_oplus(T, x + c * 6);
    pose_store(Tn, A.cam_qt + ((size_t)nxt * A.NC + W.cam_off + c) * 7);
  }
  if (tid == 0) { S.scale_cam = sc_t; S.pcg_ok = sc[4] != 0.0 ? 1 : 0; S.pcg_iterations += (int)sc[2]; }
}

__global__ __launch_bounds__(kPcgThreads) void ba_pcg_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st, double tol,
                                                            int max_iter_param) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const BAWin W = wins[blockIdx.x];
  BAState& S = st[blockIdx.x];
  if (S.phase != PH_RUN) return;
  const int nf = W.n_free, n = 6 * nf;
  double* x = lds; double* r = x + n; double* z = r + n; double* p = z + n; double* Ap = p + n;   // Ap: kPcgThreads doubles
  double* Mi = Ap + kPcgThreads;       // nf * 36
  double* scratch = Mi + nf * 36;      // 32
  const double* Sg = A.S + W.S_off;
  const double* bs = A.bschur + W.x_off;
  const int tid = threadIdx.x;
  double* ok_s = scratch + 31;         // keeps every LDS object inside the (16-B aligned) dynamic region
  if (tid == 0) *ok_s = 1.0;
  __syncthreads();
  if (tid < nf) {
    double F[36], Fi[36];
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
#pragma unroll
      for (int c = 0; c < 6; c++) F[rr * 6 + c] = Sg[(size_t)(tid * 6 + rr) * n + tid * 6 + c];
    if (!spd_inverse<6>(F, Fi)) *ok_s = 0.0;
#pragma unroll
    for (int i = 0; i < 36; i++) Mi[tid * 36 + i] = Fi[i];
  }
  if (tid < n) { x[tid] = 0.0; r[tid] = bs[tid]; }
  __syncthreads();
  auto precond = [&]() {               // z = M^-1 r
    if (tid < n) {
      const int b = tid / 6, rr = tid - b * 6;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 6; c++) s += Mi[b * 36 + rr * 6 + c] * r[b * 6 + c];
      z[tid] = s;
    }
  };
  precond();
  __syncthreads();
  if (tid < n) p[tid] = z[tid];
  double rz = block_sum(tid < n ? r[tid] * z[tid] : 0.0, scratch);
  const double rz0 = rz;
  const int max_iter = max_iter_param > 0 ? max_iter_param : 10 * n;
  int iters = 0;
  bool ok = *ok_s != 0.0 && isfinite(rz0);
  // S is symmetric and stored in full, so y = S p is computed column-wise: lane <-> column c, K row slices per column,
  // every load is a coalesced 512-B row segment, independent of its neighbours (deep memory-level parallelism), and no
  // cross-lane reduction is needed: y[c] = sum_k part[k][c] in a fixed order.
  const int K = max(1, min(8, kPcgThreads / max(n, 1)));
  const int rows_per = (n + K - 1) / K;
  const int mv_c = tid % max(n, 1), mv_k = tid / max(n, 1);
  const bool mv_on = n > 0 && mv_k < K;
  const int mv_r0 = mv_k * rows_per, mv_r1 = min(n, mv_r0 + rows_per);
  double* part = Ap;                    // [K][n] partial products live in the Ap..Mi gap: K*n <= kPcgThreads doubles
  if (ok && rz0 > 0.0) {
    const double stop = tol * tol * rz0;
    for (; iters < max_iter;) {
      if (mv_on) {
        double acc = 0.0;
        const double* Sc = Sg + mv_c;
#pragma unroll 4
        for (int row = mv_r0; row < mv_r1; row++) acc += Sc[(size_t)row * n] * p[row];
        part[mv_k * n + mv_c] = acc;
      }
      __syncthreads();
      double ap = 0.0;
      if (tid < n) { for (int k = 0; k < K; k++) ap += part[k * n + tid]; }
      const double pAp = block_sum(tid < n ? p[tid] * ap : 0.0, scratch);
      if (!(pAp > 0.0) || !isfinite(pAp)) { ok = false; break; }
      const double alpha = rz / pAp;
      if (tid < n) { x[tid] += alpha * p[tid]; r[tid] -= alpha * ap; }
      __syncthreads();
      precond();
      __syncthreads();
      const double rz_new = block_sum(tid < n ? r[tid] * z[tid] : 0.0, scratch);
      iters++;
      if (!isfinite(rz_new)) { ok = false; break; }
      if (rz_new <= stop) break;
      const double beta = rz_new / rz;
      rz = rz_new;
      if (tid < n) p[tid] = z[tid] + beta * p[tid];
      __syncthreads();
    }
  }
  __syncthreads();
  solve_epilogue(A, W, S, x, scratch, ok, iters);
}

// ================================================================== exact solve of the reduced camera system
// grid (nW); block kPcgThreads; dynamic LDS: (2*nf*36 + 2*n + 32 + lds_tri_doubles) doubles.
// Right-looking block Cholesky (6x6 camera blocks) in place on the LOWER block triangle of S, the right-hand side carried
// along as an extra block row (forward substitution for free), then block back-substitution.  This is the counterpart of
// the reference's exact factorisation (Eigen::SimplicialLDLT, solvers/linear_solver_eigen.h:94-124): S is read once from
// HBM instead of once per PCG iteration, and the result does not depend on an iteration tolerance.  A non-positive pivot
// reports failure, which Levenberg–Marquardt turns into a rejected trial (optimization_algorithm_levenberg.cpp:126-127).
__global__ __launch_bounds__(kPcgThreads) void ba_chol_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st, int lds_tri_doubles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.x);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  const int nf = W.n_free, n = 6 * nf;
  double* linv = lds;                    // [nf][36] inverse of the diagonal Cholesky blocks (lower)
  double* panel = linv + nf * 36;        // [nf][36] current block column of L
  double* y = panel + nf * 36;           // [n] right-hand side -> forward solution
  double* x = y + n;                     // [n] solution
  double* scratch = x + n;               // [32]
  double* okf = scratch + 31;
  double* tri = scratch + 32;            // LDS-resident trailing block triangle (cameras >= m0), 36 doubles per block
  double* Sg = A.S + W.S_off;
  const int tid = threadIdx.x;
  // largest trailing triangle that fits: blocks (i, j), i >= j >= m0, live in LDS for the whole factorisation, so their
  // read-modify-write updates never wait for HBM/L2; only the first m0 block columns are updated in global memory
  int mt = 0;
  while (mt < nf && (mt + 1) * (mt + 2) / 2 * 36 <= lds_tri_doubles) mt++;
  const int m0 = nf - mt;
  auto tri_blk = [&](int i, int j) { const int ii = i - m0, jj = j - m0; return tri + (size_t)(ii * (ii + 1) / 2 + jj) * 36; };
  if (tid == 0) *okf = 1.0;
  if (tid < n) y[tid] = A.bschur[W.x_off + tid];
  for (int t = tid; t < mt * (mt + 1) / 2 * 6; t += kPcgThreads) {        // lane <-> one row of one block
    const int blk = t / 6, r = t - blk * 6;
    int ii = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= blk) ii++;
    while (ii * (ii + 1) / 2 > blk) ii--;
    const int jj = blk - ii * (ii + 1) / 2;
    const double* src = Sg + (size_t)(6 * (m0 + ii) + r) * n + 6 * (m0 + jj);
    double* dst = tri + (size_t)blk * 36 + r * 6;
#pragma unroll
    for (int c = 0; c < 6; c++) dst[c] = src[c];
  }
  __syncthreads();
  for (int k = 0; k < nf; k++) {
    const bool k_lds = k >= m0;
    // (1) diagonal block: L_kk = chol(A_kk), Linv_kk, y_k = Linv_kk b_k
    if (tid == 0) {
      double a[6][6], L[6][6], Li[6][6];
      if (k_lds) { const double* d = tri_blk(k, k); for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) a[r][c] = d[r * 6 + c]; }
      else for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) a[r][c] = Sg[(size_t)(6 * k + r) * n + 6 * k + c];
      bool ok = true;
      for (int j = 0; j < 6; j++) {
        double d = a[j][j];
        for (int m = 0; m < j; m++) d -= L[j][m] * L[j][m];
        if (!(d > 0.0) || !isfinite(d)) ok = false;
        const double ljj = sqrt(d), inv = 1.0 / ljj;
        L[j][j] = ljj;
        for (int i = j + 1; i < 6; i++) {
          double sacc = a[i][j];
          for (int m = 0; m < j; m++) sacc -= L[i][m] * L[j][m];
          L[i][j] = sacc * inv;
        }
      }
      for (int j = 0; j < 6; j++) {
        Li[j][j] = 1.0 / L[j][j];
        for (int i = j + 1; i < 6; i++) {
          double sacc = 0.0;
          for (int m = j; m < i; m++) sacc -= L[i][m] * Li[m][j];
          Li[i][j] = sacc / L[i][i];
        }
      }
      for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) linv[k * 36 + r * 6 + c] = c <= r ? Li[r][c] : 0.0;
      double yk[6];
      for (int r = 0; r < 6; r++) { double sacc = 0.0; for (int c = 0; c <= r; c++) sacc += Li[r][c] * y[6 * k + c]; yk[r] = sacc; }
      for (int r = 0; r < 6; r++) y[6 * k + r] = yk[r];
      if (!ok) *okf = 0.0;
    }
    __syncthreads();
    // (2) panel: L_ik = A_ik Linv_kk^T for i > k (lane <-> one row of one block), b_i -= L_ik y_k.  The back-substitution
    //     reads L from global memory, so the panel is stored there as well (plain stores, nothing waits for them).
    const int m_rows = (nf - k - 1) * 6;
    for (int t = tid; t < m_rows; t += kPcgThreads) {
      const int i = k + 1 + t / 6, r = t % 6;
      double* grow = Sg + (size_t)(6 * i + r) * n + 6 * k;
      double arow[6], lrow[6];
      if (k_lds) { const double* d = tri_blk(i, k) + r * 6;
#pragma unroll
        for (int c = 0; c < 6; c++) arow[c] = d[c];
      } else {
#pragma unroll
        for (int c = 0; c < 6; c++) arow[c] = grow[c];
      }
      const double* Li = linv + k * 36;
      double dotv = 0.0;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double sacc = 0.0;
#pragma unroll
        for (int m = 0; m <= c; m++) sacc += arow[m] * Li[c * 6 + m];
        lrow[c] = sacc;
        dotv += sacc * y[6 * k + c];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) { grow[c] = lrow[c]; panel[i * 36 + r * 6 + c] = lrow[c]; }
      y[6 * i + r] -= dotv;
    }
    __syncthreads();
    // (3) trailing update: A_ij -= L_ik L_jk^T for k < j <= i (lane <-> block; lower triangle only)
    const int m = nf - k - 1;
    const int nblk = m * (m + 1) / 2;
    for (int t = tid; t < nblk; t += kPcgThreads) {
      int ii = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
      while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
      while (ii * (ii + 1) / 2 > t) ii--;
      const int jj = t - ii * (ii + 1) / 2;
      const int i = k + 1 + ii, j = k + 1 + jj;
      const double* Pi = panel + i * 36;
      const double* Pj = panel + j * 36;
      double pj[36];
#pragma unroll
      for (int q = 0; q < 36; q++) pj[q] = Pj[q];
      const bool in_lds = j >= m0;
      double* blk = in_lds ? tri_blk(i, j) : nullptr;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        double* row = in_lds ? blk + r * 6 : Sg + (size_t)(6 * i + r) * n + 6 * j;
        double pr[6];
#pragma unroll
        for (int q = 0; q < 6; q++) pr[q] = Pi[r * 6 + q];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          double sacc = 0.0;
#pragma unroll
          for (int q = 0; q < 6; q++) sacc += pr[q] * pj[c * 6 + q];
          row[c] -= sacc;
        }
      }
    }
    __syncthreads();
  }
  // back substitution: L^T x = y
  for (int k = nf - 1; k >= 0; k--) {
    if (tid < 6) {
      const double* Li = linv + k * 36;
      double sacc = 0.0;
      for (int mm = tid; mm < 6; mm++) sacc += Li[mm * 6 + tid] * y[6 * k + mm];      // x_k = Linv_kk^T y_k
      x[6 * k + tid] = sacc;
    }
    __syncthreads();
    for (int t = tid; t < 6 * k; t += kPcgThreads) {                                  // y_j -= L_kj^T x_k, j < k (coalesced along the row)
      double sacc = 0.0;
#pragma unroll
      for (int r = 0; r < 6; r++) sacc += Sg[(size_t)(6 * k + r) * n + t] * x[6 * k + r];
      y[t] -= sacc;
    }
    __syncthreads();
  }
  const bool ok = *okf != 0.0;
  solve_epilogue(A, W, S, x, scratch, ok, 0);
}

// ================================================================== exact solve on the fp64 matrix cores
// grid (nW); block kCholMThreads (8 wavefronts); dynamic LDS kCholMLdsDoubles doubles.  For n = 6*n_free <= 304.
// Right-looking Cholesky on 16x16 tiles with the WHOLE lower tile triangle held in registers for the entire factorisation
// (<= 190 tiles, 28 per wavefront, 4 doubles per lane each): S is read from HBM exactly once, L never leaves the chip.
// Wavefronts 1..7 own the tiles; wavefront 0 (the panel wave) owns no tile and does the serial work, so its 16-double row
// buffers never compete with the accumulator tiles for registers.  Per tile column J (two barriers):
//   (c) tile waves: L_IJ = A_IJ L_JJ^-T as four v_mfma_f64_16x16x4_f64 per tile of column J, result kept in the registers (it is
//       the L the back substitution needs) and written to the LDS panel buffer as the operand of (d);
//   (d) tile waves: every tile (I,K), K > J, takes T -= L_IJ L_KJ^T on the matrix cores — operands are read once per 1024 FMAs
//       instead of once per 1.5 as in the 6x6 register-blocked kernel above, whose trailing update is LDS-bandwidth bound;
//       tiles of column J+1 are then final and go to the other panel buffer, the diagonal tile J+2 to its slot;
//       LOOKAHEAD, same phase: the panel wave forward-substitutes the right-hand side, applies column J's update to the
//       diagonal tile J+1 itself (four MFMAs on the published copy) and factors it right-looking with one lane per row — lane 16
//       carries the right-hand side and lanes 17..32 the identity as extra rows, which yields y_{J+1} and L^-1 from the same
//       recurrence (v_readlane broadcasts, no LDS traffic inside it).  The serial factorisation is off the critical path.
// Tile element layout of v_mfma_f64_16x16x4_f64: C/D lane l, register g -> (row (l>>4) + 4g, col l&15); A[i][k] and B[k][j]
// come from lane i + 16k resp. j + 16k.
constexpr int kCholMThreads = 512;
constexpr int kCholMTileWaves = kCholMThreads / 64 - 1;
constexpr int kCholMMaxTiles = 19;                                   // 19 * 16 = 304 >= 6 * 50
constexpr int kCholMSlots = 28;                                      // ceil(190 / 7)
constexpr int kCholMStride = 17;                                     // padded LDS row of 16 doubles
constexpr int kCholMN = kCholMMaxTiles * 16;
constexpr int kCholMLdsDoubles = 2 * kCholMN * kCholMStride + kCholMMaxTiles * 16 * kCholMStride + 16 * kCholMStride + 3 * kCholMN + 32;
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Panel wave: factor the 16x16 tile in Dg (lower triangle used); L^-1 -> Li (operand of the column's L_IJ = A_IJ L_JJ^-T) and -> Dg
// (kept for the back substitution, which only needs the inverse); rhs y[0..15] -> L^-1 y.
__device__ __forceinline__ bool chol_tile_factor(double* Dg, double* Li, double* y, int lane) {
  const int r = lane < 32 ? lane : 32;                               // 0..15 tile rows, 16 rhs, 17..32 identity rows
  // one load path for all lanes: tile rows and the right-hand side are read through a per-lane pointer, the identity rows read
  // the (finite) right-hand side too and are overwritten
  const double* src = (r < 16) ? Dg + r * kCholMStride : y;
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { const double v = src[c]; a[c] = (r <= 16) ? v : (r - 17 == c ? 1.0 : 0.0); }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const double d = readlane_f64(a[c], c);
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    // 1/sqrt(d): v_rsq_f64 seed + two Newton steps (the library sqrt and divide are ~45 dependent instructions, which is what
    // bounds this serial recurrence; the result is within an ulp or two, L L^T = S to rounding either way)
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * (1.5 - (0.5 * d) * (inv * inv));
    inv = inv * (1.5 - (0.5 * d) * (inv * inv));
    const double lc = a[c] * inv;                                    // lane c: sqrt(d); below: L[r][c]; rhs lane: y_c
    a[c] = lc;
#pragma unroll
    for (int c2 = c + 1; c2 < 16; c2++) a[c2] -= lc * readlane_f64(lc, c2);   // A[r][c2] -= L[r][c] L[c2][c]
  }
  if (lane == 16) {
#pragma unroll
    for (int c = 0; c < 16; c++) y[c] = a[c];
  } else if (lane > 16 && lane <= 32) {                              // lane 17+k holds column k of L^-1
#pragma unroll
    for (int c = 0; c < 16; c++) { Li[c * kCholMStride + (lane - 17)] = a[c]; Dg[c * kCholMStride + (lane - 17)] = a[c]; }
  }
  return ok;
}

__global__ __launch_bounds__(kCholMThreads) void ba_chol_mfma_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.x);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_RUN) return;
  if (A.chol_plan && A.chol_plan[W.win_index].mode == 1) return;      // this window is ba_chol_sparse_kernel's (launched next to this one when a group holds both kinds)
  const int nf = W.n_free, n = 6 * nf, NT = (n + 15) >> 4, N = NT << 4;
  double* Lp0 = lds;                                       // [2][N][17] panel buffers: column J in buffer J & 1 (raw, then L)
  double* Dall = Lp0 + 2 * kCholMN * kCholMStride;         // [NT][16][17] diagonal tiles: raw until factored, then L_JJ
  double* Li = Dall + kCholMMaxTiles * 16 * kCholMStride;  // [16][17] inverse of the current diagonal factor
  double* colsum = Li + 16 * kCholMStride;                 // [7][16] per tile wave: column sums of the back substitution (room for N)
  double* y = colsum + kCholMN;                            // [N] right-hand side -> forward solution
  double* x = y + kCholMN;                                 // [N] solution
  double* scratch = x + kCholMN;                           // [32]
  double* okf = scratch + 31;
  const double* Sg = A.S + W.S_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 4, lcol = lane & 15;
#ifdef LLD_EXPERIMENTS
  long long* stamp_base = A.chol_stamps ? A.chol_stamps + ((size_t)W.win_index * kCholStampWaves + wave) * kCholStampSlots : nullptr;
#endif
  LLD_CHOL_STAMP(0);
  if (tid < N) y[tid] = (tid < n) ? A.bschur[W.x_off + tid] : 0.0;
  if (tid == 0) *okf = 1.0;

  if (wave == 0) {
    // ================================================================ panel wave
    // The diagonal tile 0 does not wait for the tile wavefronts (round 4): this wavefront fetches its 256 values itself and factors it while
    // the other seven still load their 28 tiles each - the prologue's publish of that tile and its 2.3 us factorisation leave the chain.
    if (NT > 0) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int idx = lane + 64 * q, row = idx >> 4, col = idx & 15;
        const bool inside = row < n && col < n, lower = inside && col <= row;
        const double v = Sg[lower ? row * n + col : 0];
        Dall[row * kCholMStride + col] = lower ? v : ((!inside && row == col) ? 1.0 : 0.0);
      }
      if (!chol_tile_factor(Dall, Li, y, lane) && lane == 0) *okf = 0.0;      // (y[0..15] and *okf were written by this wavefront's own lanes above)
    }
    __syncthreads();                                                   // tiles loaded, y staged
    LLD_CHOL_STAMP(1);
    __syncthreads();                                                   // prologue publish done: column 0, diagonal tile 1
    LLD_CHOL_STAMP(2);
    LLD_CHOL_STAMP(3);
    __syncthreads();                                                   // (diagonal tile 0 factored: long since)
    for (int J = 0; J < NT; J++) {
      const double* Lp = Lp0 + (J & 1) * kCholMN * kCholMStride;
      LLD_CHOL_STAMP(8 + 6 * J);
      __syncthreads();                                                 // (c) done: Lp holds L(:,J)
      LLD_CHOL_STAMP(9 + 6 * J);
      if (lane < 16 && J + 1 < NT) {                                   // y_(J+1) -= L_(J+1)J y_J: the sixteen rows the next tile factor carries along;
        const double* pr = Lp + (16 * (J + 1) + lane) * kCholMStride;  // the rows below are the tile wavefronts' (they wait for this wavefront
        double dotv = 0.0;                                             // in the late columns: round 4, 10.9 us off its path)
#pragma unroll
        for (int c = 0; c < 16; c++) dotv += pr[c] * y[16 * J + c];
        y[16 * (J + 1) + lane] -= dotv;
      }
      LLD_CHOL_STAMP(10 + 6 * J);
      if (J + 1 < NT) {
        // lookahead: diagonal tile J+1 (published with the updates of columns < J) takes column J's update here, then is factored
        double* Dg = Dall + (J + 1) * 16 * kCholMStride;
        const double* pa = Lp + (16 * (J + 1) + lcol) * kCholMStride + lrow;
        v4d c;
#pragma unroll
        for (int g = 0; g < 4; g++) c[g] = Dg[(lrow + 4 * g) * kCholMStride + lcol];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk], pa[4 * kk], c, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++) Dg[(lrow + 4 * g) * kCholMStride + lcol] = c[g];
        LLD_CHOL_STAMP(11 + 6 * J);
        if (!chol_tile_factor(Dg, Li, y + 16 * (J + 1), lane) && lane == 0) *okf = 0.0;
      }
      LLD_CHOL_STAMP(12 + 6 * J);
      __syncthreads();                                                 // (d) + lookahead done
      LLD_CHOL_STAMP(13 + 6 * J);
    }
    LLD_CHOL_STAMP(4);
    // back substitution L^T x = y: x_J = L_JJ^-T (y_J - s_J), s_J = the tile waves' column sums of L_IJ^T x_I (I > J); two barriers per tile
    for (int J = NT - 1; J >= 0; J--) {
      __syncthreads();                                                 // column sums of J complete
      const double* Di = Dall + J * 16 * kCholMStride;                 // L_JJ^-1
      const int c = lane & 15, part = lane >> 4;
      double xc = 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 4 * part + q;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kCholMTileWaves; w++) sum += colsum[w * 16 + r];
        xc += Di[r * kCholMStride + c] * (y[16 * J + r] - sum);
      }
      xc += __shfl_xor(xc, 16); xc += __shfl_xor(xc, 32);
      if (lane < 16) x[16 * J + c] = xc;
      __syncthreads();                                                 // x_J ready
    }
    LLD_CHOL_STAMP(5);
  } else {
    // ================================================================ tile waves
    // tile coordinates of this wavefront's slots (wave-uniform, integer-only: scalar registers).  Tile (I, K) belongs to tile wave
    // (I + 2K) mod 7: the tiles of one COLUMN (I consecutive) and of one row spread evenly over the seven waves, so the
    // L_IJ = A_IJ L_JJ^-T phase of a column is at most ceil(rows / 7) tiles deep (a round-robin over the packed index
    // I (I + 1) / 2 + K puts a column on four of the seven waves only); <= 28 tiles per wave for 19 tile rows.
    int tI[kCholMSlots], tK[kCholMSlots];
    {
      const int w0 = wave - 1;
      int I = 0, K = (4 * w0) % 7;                                     // in row I: K = 4 (w0 - I) mod 7 (4 = 2^-1 mod 7), then every 7th column
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        while (I < NT && K > I) { I++; K = (4 * (((w0 - I) % 7) + 7)) % 7; }
        const bool valid = I < NT;
        tI[sl] = valid ? I : -1;
        tK[sl] = valid ? K : -1;
        K += 7;
      }
    }
    // S -> registers (lower triangle; the padding rows/columns carry an identity so that L is the identity there).  All 28 tiles
    // (112 loads per lane, the hardware queues what it cannot keep in flight) go out before the first value is touched: one slot at a time, the 28 slots were 28 dependent round
    // trips to another XCD's L2 (18 us of a 160 us kernel).  Tile base in scalar registers, four per-lane offsets shared by all slots.
    unsigned* cmask = reinterpret_cast<unsigned*>(colsum + kCholMTileWaves * 16) + (wave - 1) * kCholMMaxTiles;    // per-column slot masks, see below (the tail of the column-sum area: 112 of its 304 doubles are used)
    v4d acc[kCholMSlots];
    int offg[4];
#pragma unroll
    for (int g = 0; g < 4; g++) offg[g] = (lrow + 4 * g) * n + lcol;
    constexpr int kLoadGroup = 28;
#pragma unroll
    for (int s0 = 0; s0 < kCholMSlots; s0 += kLoadGroup) {
#pragma unroll
      for (int sl = s0; sl < s0 + kLoadGroup; sl++) {
        v4d v = {0.0, 0.0, 0.0, 0.0};
        if (tI[sl] >= 0) {
          const double* base = Sg + (16 * tI[sl]) * n + 16 * tK[sl];
          if (tK[sl] < tI[sl] && 16 * tI[sl] + 16 <= n) {               // interior tile (wave-uniform): scalar base + the shared lane offsets
#pragma unroll
            for (int g = 0; g < 4; g++) v[g] = base[offg[g]];
          } else {
            const int col = 16 * tK[sl] + lcol;
#pragma unroll
            for (int g = 0; g < 4; g++) {
              const int row = 16 * tI[sl] + lrow + 4 * g;
              const bool lower = row < n && col < n && col <= row;
              v[g] = base[lower ? offg[g] : 0];
            }
          }
        }
        acc[sl] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s0 == 0) {                                                   // (while the loads are in flight)
        // Per column J the slots of this wavefront's OFF-DIAGONAL tiles of that column, as a bit mask in LDS (round 4).  The L_IJ phase and the back
        // substitution touch at most three tiles per column and wavefront but walked all 28 slots for them, and the tile coordinates live in
        // spilled scalar registers (~35 cycles per slot looked at: the walk, not the barriers, is what a column of the back substitution costs -
        // DESIGN.md section 7): with the mask a slot that is not in the column costs one scalar bit test.
        if (lane < kCholMMaxTiles) cmask[lane] = 0u;
#pragma unroll
        for (int sl = 0; sl < kCholMSlots; sl++)
          if (tI[sl] > tK[sl] && lane == 0) atomicOr(&cmask[tK[sl]], 1u << sl);
        // (the trailing update takes most slots in the columns where it is on the critical path; masks for it measured no gain)
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sl = s0; sl < s0 + kLoadGroup; sl++) {
        if (tI[sl] >= 0 && !(tK[sl] < tI[sl] && 16 * tI[sl] + 16 <= n)) {
          const int col = 16 * tK[sl] + lcol;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int row = 16 * tI[sl] + lrow + 4 * g;
            const bool inside = row < n && col < n, lower = inside && col <= row;
            acc[sl][g] = lower ? acc[sl][g] : ((!inside && row == col) ? 1.0 : 0.0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    LLD_CHOL_STAMP(1);
    __syncthreads();
    {
      // prologue publish: column 0 (raw) -> panel buffer 0, diagonal tile 1 (raw) -> its slot (tile 0 is the panel wavefront's own business)
      int off_cd = lrow * kCholMStride + lcol;
      asm volatile("" : "+v"(off_cd));
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        const bool diag01 = tI[sl] == tK[sl] && tI[sl] == 1;
        if (diag01 || (tK[sl] == 0 && tI[sl] > 0)) {
          double* dst = (diag01 ? Dall + tI[sl] * 16 * kCholMStride : Lp0 + 16 * tI[sl] * kCholMStride) + off_cd;
#pragma unroll
          for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = acc[sl][g];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    LLD_CHOL_STAMP(2);
    __syncthreads();                                                   // prologue publish done
    __syncthreads();                                                   // diagonal tile 0 factored: Li = L_00^-1
    LLD_CHOL_STAMP(3);
    for (int J = 0; J < NT; J++) {
      // Per-lane LDS offsets, made opaque once per iteration: otherwise the per-slot addresses are hoisted out of the J loop as
      // loop invariants and push the accumulator tiles out of the register file.
      int off_cd = lrow * kCholMStride + lcol, off_ab = lcol * kCholMStride + lrow;
      asm volatile("" : "+v"(off_cd), "+v"(off_ab));
      double* Lp = Lp0 + (J & 1) * kCholMN * kCholMStride;
      double* Lnext = Lp0 + ((J + 1) & 1) * kCholMN * kCholMStride;
      LLD_CHOL_STAMP(8 + 6 * J);
      // (c) L_IJ = A_IJ L_JJ^-T on the matrix cores; keep it (back substitution) and publish it (operand of d)
      const unsigned mcol = __builtin_amdgcn_readfirstlane(cmask[J]);
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        if (mcol & (1u << sl)) {
          const double* pa = Lp + 16 * tI[sl] * kCholMStride + off_ab;
          const double* pb = Li + off_ab;
          v4d c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pb[4 * kk], c, 0, 0, 0);
          acc[sl] = c;
          double* dst = Lp + 16 * tI[sl] * kCholMStride + off_cd;
#pragma unroll
          for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = c[g];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      LLD_CHOL_STAMP(9 + 6 * J);
      __syncthreads();                                                 // (c) done
      LLD_CHOL_STAMP(10 + 6 * J);
      // (d) trailing update; column J+1 and the diagonal tile J+2 are final afterwards and are published for the next steps.
      //     The diagonal tile J+1 is not touched: the panel wave finishes it from its published copy (lookahead).
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        if (tK[sl] > J && !(tI[sl] == J + 1 && tK[sl] == J + 1)) {
          const double* pa = Lp + 16 * tI[sl] * kCholMStride + off_ab;
          const double* pb = Lp + 16 * tK[sl] * kCholMStride + off_ab;
          v4d c = acc[sl];
#pragma unroll
          for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk], pb[4 * kk], c, 0, 0, 0);
          acc[sl] = c;
          const bool next_col = tK[sl] == J + 1, next_diag = tI[sl] == J + 2 && tK[sl] == J + 2;
          if (next_col || next_diag) {
            double* dst = (next_diag ? Dall + (J + 2) * 16 * kCholMStride : Lnext + 16 * tI[sl] * kCholMStride) + off_cd;
#pragma unroll
            for (int g = 0; g < 4; g++) dst[4 * g * kCholMStride] = c[g];
          }
        }
        if (sl & 1) __builtin_amdgcn_sched_barrier(0);                 // let the loads of one tile overlap the MFMAs of its neighbour, not more
      }
      // forward substitution of the right-hand side below tile row J + 1: y_I -= L_IJ y_J for the tile rows I = J + 2 + (wave - 1), + 7, ...
      // (lane = row of the tile + 16 x quarter of the columns; L(:,J) is in the panel buffer, y_J is final since the previous column)
      for (int I = J + 1 + wave; I < NT; I += kCholMTileWaves) {
        const double* pr = Lp + (16 * I + lcol) * kCholMStride + 4 * lrow;
        const double* yj = y + 16 * J + 4 * lrow;
        double dotv = pr[0] * yj[0] + pr[1] * yj[1] + pr[2] * yj[2] + pr[3] * yj[3];
        dotv += __shfl_xor(dotv, 16); dotv += __shfl_xor(dotv, 32);
        if (lane < 16) y[16 * I + lane] -= dotv;
      }
      LLD_CHOL_STAMP(12 + 6 * J);
      __syncthreads();                                                 // (d) + lookahead done
      LLD_CHOL_STAMP(13 + 6 * J);
    }
    LLD_CHOL_STAMP(4);
    // back substitution L^T x = y: L lives in the register tiles, s_c = sum_{i below tile J} L[i][16J + c] x_i
    for (int J = NT - 1; J >= 0; J--) {
      double part = 0.0; bool any = false;
      const unsigned mcol = __builtin_amdgcn_readfirstlane(cmask[J]);
#pragma unroll
      for (int sl = 0; sl < kCholMSlots; sl++) {
        if (mcol & (1u << sl)) {
#pragma unroll
          for (int g = 0; g < 4; g++) part += acc[sl][g] * x[16 * tI[sl] + lrow + 4 * g];
          any = true;
        }
      }
      if (any) { part += __shfl_xor(part, 16); part += __shfl_xor(part, 32); }   // wave-uniform; sum over the 4 row groups of a column
      if (lane < 16) colsum[(wave - 1) * 16 + lane] = part;            // one row of partial sums per tile wave: no atomics
      __syncthreads();                                                 // column sums of J complete
      __syncthreads();                                                 // x_J ready
    }
    LLD_CHOL_STAMP(5);
  }
  const bool ok = *okf != 0.0;
  solve_epilogue(A, W, S, x, scratch, ok, 0);
  LLD_CHOL_STAMP(6);
}


#include "lld_ba_chol_sparse.h"   // round 5: the same factorisation along the structure of S, two panel wavefronts where the plan has two chains


// ================================================================== LM control
// One wavefront per window: lane 0 takes the accept / reject decision of the trial that just ran
// (optimization_algorithm_levenberg.cpp:118-163) and advances the window's state machine; all lanes then clear the
// camera accumulators when a new linearisation is due; the wavefront of the LAST window to get here publishes the group's totals.
// Runs as ba_control_kernel (grid nW, block 64) or, for small groups, inside ba_backsub_ctl_kernel as the last act of the window's last
// workgroup (one dependent launch less per super-step).  `lane` 0..63, all lanes of ONE wavefront; no block-level barrier inside.
__device__ __forceinline__ void ba_control_body(const BAArrays& A, const BAWin& W, BAState& S, int lane, int n_windows, int abort_flag,
                                                int wrow /* the window's index in its group; < 0: a grid row without a window - it only takes its ticket */, int nw_group /* windows of the group */,
                                                int* __restrict__ counters /* [4]: running, transition, finalize, ticket (all zero on entry) */,
                                                int* __restrict__ host_counters /* pinned host memory: the group's totals */,
                                                const int* __restrict__ host_abort /* pinned host memory: the live stop flag, forwarded by the polling host thread (null: only the launch-time sample counts) */) {
  double tempChi = 0.0, scale_l = 0.0;
  int do_clear = 0;
  const bool valid = wrow >= 0;
  if (valid) {
  if (S.phase == PH_RUN) {                           // interleaved partial sums + fixed shuffle tree (deterministic)
    const int nb = W.nt_pt + W.nt_ln;
    for (int i = lane; i < nb; i += 64) { tempChi += xwg_load(&A.chi_part2[W.part_off + i]); scale_l += xwg_load(&A.scale_part[W.part_off + i]); }
    tempChi = wave_sum(tempChi); scale_l = wave_sum(scale_l);
  }
  if (lane == 0 && S.phase == PH_RUN) {
    double scale = S.scale_cam + scale_l;
    if (!S.pcg_ok) tempChi = 1.7976931348623157e308;
    double rho = (S.currentChi - tempChi);
    scale += 1e-3;
    rho /= scale;
    if (rho > 0 && isfinite(tempChi)) {
      double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
      alpha = fmin(alpha, 2. / 3.);
      S.lambda *= fmax(1. / 3., alpha);
      S.ni = 2;
      S.currentChi = tempChi;
      S.cur ^= 1;                                  // discardTop: the trial buffer becomes the state
    } else {
      S.lambda *= S.ni; S.ni *= 2;                 // pop: keep the old buffer
    }
    S.q++;
    const int round = S.round;
    S.lm_trials[round]++;
    // terminate(): the host's sample of *abort_flag at the launch of this super-step, its live forward, or the deterministic test hook
    const bool stop = abort_flag || (host_abort && __hip_atomic_load(host_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) ||
                      (W.abort_after > 0 && S.lm_trials[0] + S.lm_trials[1] >= W.abort_after);
    const bool again = (rho < 0 && S.q < W.max_trials && !stop);
    if (!again) {
      bool term = (S.q == W.max_trials || rho == 0);
      if (!term) {
        if ((S.iniChi - S.currentChi) * 1e3 < S.iniChi) S.nBad++; else S.nBad = 0;
        if (S.nBad >= 3) term = true;
      }
      S.it++;
      S.lm_iterations[round]++;
      if (!term && S.it < W.its[round] && !stop) { S.need_lin = 1; S.maxdiag_bits = 0ull; do_clear = 1; }
      else if (round == 0) {
        S.chi2_round1 = S.currentChi; S.chi2_final = S.currentChi;
        if (stop) { S.aborted = 1; S.phase = PH_FINALIZE; }           // Optimizer.cc:1230-1232: bDoMore = false, the final classification still runs
        else if (W.protocol == 1) S.phase = PH_FINALIZE;              // global BA: optimize(nIterations) and nothing else
        else S.phase = PH_TRANSITION;
      } else { S.chi2_final = S.currentChi; S.aborted = stop ? 1 : 0; S.phase = PH_FINALIZE; }   // lld_ba_stats::aborted = the flag at the last poll
    }
  }
  do_clear = __builtin_amdgcn_readfirstlane(do_clear);
  if (do_clear) {
    for (int i = lane; i < W.n_free * 21; i += 64) A.Hpp[(size_t)W.hpp_off * 21 + i] = 0.0;
    for (int i = lane; i < W.n_free * 6; i += 64) A.bp[(size_t)W.hpp_off * 6 + i] = 0.0;
    if (W.big) for (int i = lane; i < W.n_free * 27; i += 64) A.hpp_part[W.hpart_off + i] = 0.0;
  }
  }
  int last = 0;
  if (lane == 0) {
    if (valid) {
      const int ph = S.phase;
      if (ph == PH_RUN) atomicAdd(&counters[0], 1);
      else if (ph == PH_TRANSITION) atomicAdd(&counters[1], 1);
      else if (ph == PH_FINALIZE) atomicAdd(&counters[2], 1);
      // the window's "still at work" bit for the rebuild of the row -> window map below: write-through, acknowledged before the ticket
      if (A.slot_map) { xwg_store_i32(&A.active_pub[wrow], (ph == PH_RUN || ph == PH_TRANSITION) ? 1 : 0); xwg_stores_done(); }
    }
    // The last window's wavefront publishes the totals straight into pinned host memory and leaves the device counters at zero for the
    // next super-step: no 16-byte device-to-host copy (a blit kernel of its own, 30 - 40 us on the dependent chain of every
    // super-step, 110 us while another context's upload holds the link) and no memset.  The host reads after the event that
    // follows this kernel.  (Only atomics travel between the windows' wavefronts here: no fence - an agent-scope fence writes back the L2.)
    if (atomicAdd(&counters[3], 1) == n_windows - 1) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int v = atomicExch(&counters[i], 0);
        __hip_atomic_store(&host_counters[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      counters[3] = 0;
      __threadfence_system();
      last = 1;
    }
  }
  // The group's last control wavefront rebuilds the row -> window map for the next super-step: the windows still at work, in window
  // order, then -1 (a queued super-step may be launched with more rows than windows are left).  Plain stores: read by the next launch.
  last = __builtin_amdgcn_readfirstlane(last);
  if (last && A.slot_map) {
    int cnt = 0;
    for (int base = 0; base < nw_group; base += 64) {
      const int w = base + lane;
      const bool act = w < nw_group && xwg_load_i32(&A.active_pub[w]) != 0;
      const unsigned long long m = __ballot(act);
      if (act) A.slot_map[cnt + __popcll(m & ((1ull << lane) - 1ull))] = w;
      cnt += __popcll(m);
    }
    for (int k = cnt + lane; k < nw_group; k += 64) A.slot_map[k] = -1;
  }
}
__global__ __launch_bounds__(kCtlThreads) void ba_control_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st, int abort_flag, int nw_group,
                                                                 int* __restrict__ counters, int* __restrict__ host_counters, const int* __restrict__ host_abort) {
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.x);
  const int wi = wrow < 0 ? 0 : wrow;
  ba_control_body(A, wins[wi], st[wi], threadIdx.x, (int)gridDim.x, abort_flag, wrow, nw_group, counters, host_counters, host_abort);
}

// Point and line back-substitution in one launch AND the LM control behind it, for groups too small to fill the GPU: the window's last
// workgroup to finish (a ticket in BAState) runs ba_control_body; a window that is not running sends its first workgroup straight there
// (it still has to be counted).  grid (n_pt_blocks + max line blocks, nW), block kLmThreads.
__global__ __launch_bounds__(kLmThreads) void ba_backsub_ctl_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st, int n_pt_blocks, int abort_flag, int nw_group,
                                                                   int* __restrict__ counters, int* __restrict__ host_counters, const int* __restrict__ host_abort) {
  __shared__ int is_last;
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  const int bx = (int)blockIdx.x;
  if (wrow < 0) {                                         // a grid row without a window (queued super-step, fewer windows left): only its ticket
    if (bx == 0 && threadIdx.x < 64) ba_control_body(A, wins[0], st[0], threadIdx.x, (int)gridDim.y, abort_flag, -1, nw_group, counters, host_counters, host_abort);
    return;
  }
  const BAWin& W = wins[wrow];
  BAState& S = st[wrow];
  const bool running = S.phase == PH_RUN;                 // (uniform over the window's WORKING workgroups: the phase only changes behind their ticket.  A padding workgroup - bx beyond the
                                                          //  window's own count - may be dispatched after the control ran: it reads its row through the map of THIS launch, which the control does not
                                                          //  touch (it writes the other buffer, see BAArrays::slot_map), finds `works` false whatever the phase says, and leaves)
  const bool is_pt = bx < n_pt_blocks;
  const bool works = running && (is_pt ? bx < W.nt_pt : bx - n_pt_blocks < W.nt_ln);
  if (works) {
    if (is_pt) ba_backsub_pt_body<false, 1>(A, wins, st, bx); else ba_backsub_ln_body<false, 1>(A, wins, st, bx - n_pt_blocks);      // (packed observations only, like ba_linearize_both_kernel)
  }
  if (!running) {
    if (bx == 0 && threadIdx.x < 64) ba_control_body(A, W, S, threadIdx.x, (int)gridDim.y, abort_flag, wrow, nw_group, counters, host_counters, host_abort);
    return;
  }
  if (!works) return;
  // (thread 0 wrote the workgroup's two partial sums with write-through stores and waited for them: see xwg_store)
  if (threadIdx.x == 0) is_last = atomicAdd(&S.ticket_bs, 1) == W.nt_pt + W.nt_ln - 1;
  __syncthreads();
  if (is_last && threadIdx.x < 64) {
    if (threadIdx.x == 0) S.ticket_bs = 0;
    ba_control_body(A, W, S, threadIdx.x, (int)gridDim.y, abort_flag, wrow, nw_group, counters, host_counters, host_abort);
  }
}

// ================================================================== classification between the rounds
// grid (nb_pt + nb_ln, nW), windows in PH_TRANSITION only.
__global__ __launch_bounds__(kLmThreads) void ba_classify_kernel(BAArrays A, const BAWin* __restrict__ wins, BAState* __restrict__ st) {
  __shared__ double scratch[8];
  const int wrow = LLD_ROW_WINDOW(A, st, blockIdx.y);
  if (wrow < 0) return;
  const BAWin W = wins[wrow];
  BAState& S = st[wrow];
  if (S.phase != PH_TRANSITION) return;
  if ((int)blockIdx.x >= W.nb_pt + W.nb_ln) return;
  const int cur = S.cur;
  const CamK cam = W.cam;
  double n_active = 0.0;
  // Two phases per workgroup: the edges of the workgroup's landmarks (a contiguous range) are classified by EDGE lanes - coalesced reads,
  // one camera gather and one map per lane - then, behind a barrier, each landmark lane counts the flags of its own edges.
  if ((int)blockIdx.x < W.nb_pt) {
    const int p0 = blockIdx.x * kLmThreads, np = min(kLmThreads, W.n_pt - p0);
    const int eb = A.pt_obs_start[W.pt_off + p0], ee = A.pt_obs_start[W.pt_off + p0 + np];
    for (int e = eb + (int)threadIdx.x; e < ee; e += kLmThreads) {
      uint8_t fl = A.pe_flags[e];
      const Vec3 X = load_pt(A, cur, W.pt_off + A.pe_pt[e]);
      const Pose T = load_cam(A, cur, W.cam_off + pt_cam_of<kPkRuntime>(A, e));
      const bool depth_pos = pose_map(T, X).z > 0.0;
      const bool stereo = pt_obs_stereo<kPkRuntime>(A, e);
      if (A.pe_chi2[e] > (stereo ? 7.815 : 5.991) || !depth_pos) fl |= EF_LEVEL1;      // Optimizer.cc:1246,1260
      fl &= (uint8_t)~EF_ROBUST;                                                       // e->setRobustKernel(0)
      A.pe_flags[e] = fl;
    }
    __syncthreads();
    const int p = p0 + threadIdx.x;
    if (p < W.n_pt) {
      const int g = W.pt_off + p;
      int act = 0;
      for (int e = A.pt_obs_start[g]; e < A.pt_obs_start[g + 1]; e++) act += !(A.pe_flags[e] & EF_LEVEL1);
      A.pt_active[g] = act > 0;
      n_active += act;
    }
  } else {
    const int l0 = (blockIdx.x - W.nb_pt) * kLmThreads, nl = min(kLmThreads, W.n_ln - l0);
    const int eb = 2 * A.ln_obs_start[W.ln_off + l0], ee = 2 * A.ln_obs_start[W.ln_off + l0 + nl];
    for (int e = eb + (int)threadIdx.x; e < ee; e += kLmThreads) {
      uint8_t fl = A.le_flags[e];
      if (!(fl & EF_VALID)) continue;
      const LineQ L = load_ln(A, cur, W.ln_off + ln_line_of<kPkRuntime>(A, e >> 1));
      const Mat3 Rl = line_rotation(L);
      const Vec3 c0 = mat_col(Rl, 0), c1 = mat_col(Rl, 1);
      const double th = (fl & EF_PAIRSTEREO) ? W.th_ln_stereo : W.th_ln_mono;
      const Pose T = load_cam(A, cur, W.cam_off + ln_cam_of<kPkRuntime>(A, e >> 1));
      const LnSeg sg = ln_seg_of<kPkRuntime>(A, e);
      const bool depth_pos = line_depth_positive(cam, (e & 1) ? cam.bx_right : 0.0, T, c0, c1, L.alpha, sg.xs, sg.ys, sg.xe, sg.ye);
      if (A.le_chi2[e] > th * th || !depth_pos) fl |= EF_LEVEL1;                       // LineOptimizer.cc:141-153
      fl &= (uint8_t)~EF_ROBUST;
      A.le_flags[e] = fl;
    }
    __syncthreads();
    const int l = l0 + threadIdx.x;
    if (l < W.n_ln) {
      const int g = W.ln_off + l;
      const int e0 = 2 * A.ln_obs_start[g], e1 = 2 * A.ln_obs_start[g + 1];
      int cnt = 0, act = 0; bool has_edge = false;
      for (int e = e0; e < e1; e++) {
        const uint8_t fl = A.le_flags[e];
        if (!(fl & EF_VALID)) continue;
        has_edge = true;
        if (!(fl & EF_LEVEL1)) { cnt += 2; act++; }
      }
      const bool removed = has_edge && cnt <= W.ln_filter;                                          // LineOptimizer.cc:156-168
      if (removed) {
        for (int e = e0; e < e1; e++) if (A.le_flags[e] & EF_VALID) A.le_flags[e] |= EF_LEVEL1;
        act = 0;
      }
      A.ln_removed[g] = removed;
      A.ln_active[g] = act > 0;
      n_active += act;
    }
  }
  const double t = block_sum(n_active, scratch);
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    if (t > 0.0) atomicAdd(&S.n_active_edges, (int)(t + 0.5));
    is_last = atomicAdd(&S.ticket_cls, 1) == W.nb_pt + W.nb_ln - 1;
  }
  __syncthreads();
  if (!is_last) return;
  // The window's last workgroup starts round 2 (initializeOptimization(0); optimize(its[1])) - a kernel of its own until round 4.
  // Every other workgroup of the window has left its phase test behind (the ticket comes after all its work); what it needs from them
  // is the atomic edge count alone (their flag stores are for the next launch).
  for (int i = threadIdx.x; i < W.n_free * 21; i += kLmThreads) A.Hpp[(size_t)W.hpp_off * 21 + i] = 0.0;
  for (int i = threadIdx.x; i < W.n_free * 6; i += kLmThreads) A.bp[(size_t)W.hpp_off * 6 + i] = 0.0;
  if (W.big) for (int i = threadIdx.x; i < W.n_free * 27; i += kLmThreads) A.hpp_part[W.hpart_off + i] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    S.ticket_cls = 0;
    S.round = 1; S.it = 0; S.q = 0; S.need_lin = 1; S.maxdiag_bits = 0ull;
    S.phase = __hip_atomic_load(&S.n_active_edges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 ? PH_RUN : PH_FINALIZE;        // optimize() returns -1 on an empty active set
  }
}

// ================================================================== final classification + read-back
struct BARecordHeader {
  double chi2_round1, chi2_final;
  int lm_iterations[2], lm_trials[2];
  int pcg_iterations, aborted, win_index, n_pt_obs;      // win_index: position in the batch; n_pt_obs: point edges of the window (identity of a gathered record, lld_slam_amd/dist.py)
};
// record layout (bytes from W.rec_off): header | cam_qt[7*n_cams] | pt[3*n_pt] | x0[3*n_ln] | dir[3*n_ln] |
//                                       pt_obs_outlier[n_pe] | ln_edge_outlier[n_le] | line_removed[n_ln]
__device__ __forceinline__ double* rec_cam(unsigned char* r) { return reinterpret_cast<double*>(r + sizeof(BARecordHeader)); }

__global__ __launch_bounds__(kLmThreads) void ba_finalize_kernel(BAArrays A, const BAWin* __restrict__ wins, const BAState* __restrict__ st) {
  const BAWin W = wins[blockIdx.y];
  const BAState& S = st[blockIdx.y];
  if (S.phase != PH_FINALIZE) return;
  if ((int)blockIdx.x >= W.nb_pt + W.nb_ln + 1) return;
  const int cur = S.cur;
  const CamK cam = W.cam;
  // Optimizer.cc:1220-1222: a stop request before the first optimize() returns without classifying or writing anything
  const bool global = W.protocol == 1;                     // Optimizer::BundleAdjustment erases nothing: all flags stay 0
  const bool untouched = S.aborted && S.lm_trials[0] == 0;
  unsigned char* rec = A.records + W.rec_off;
  double* o_cam = rec_cam(rec);
  double* o_pt = o_cam + 7 * W.n_cams;
  double* o_x0 = o_pt + 3 * W.n_pt;
  double* o_dir = o_x0 + 3 * W.n_ln;
  unsigned char* o_pe = reinterpret_cast<unsigned char*>(o_dir + 3 * W.n_ln);
  unsigned char* o_le = o_pe + W.n_pe;
  unsigned char* o_rm = o_le + W.n_le;
  if ((int)blockIdx.x == W.nb_pt + W.nb_ln) {            // cameras + header
    for (int i = threadIdx.x; i < W.n_cams * 7; i += kLmThreads) o_cam[i] = A.cam_qt[((size_t)cur * A.NC + W.cam_off) * 7 + i];
    if (threadIdx.x == 0) {
      BARecordHeader h;
      h.chi2_round1 = S.chi2_round1; h.chi2_final = S.chi2_final;
      h.lm_iterations[0] = S.lm_iterations[0]; h.lm_iterations[1] = S.lm_iterations[1];
      h.lm_trials[0] = S.lm_trials[0]; h.lm_trials[1] = S.lm_trials[1];
      h.pcg_iterations = S.pcg_iterations; h.aborted = S.aborted; h.win_index = W.win_index; h.n_pt_obs = W.n_pe;
      *reinterpret_cast<BARecordHeader*>(rec) = h;
    }
    return;
  }
  // Landmark state by landmark lane, edge flags by EDGE lane (the window's edges strided over the workgroups of their landmark type:
  // coalesced reads and byte stores).  One lane per landmark walking its own edges - the first version - ran the read-back of 64 windows
  // in 0.59 ms, at a tenth of the linearisation's edge rate, and this launch ends every group's chain.
  if ((int)blockIdx.x < W.nb_pt) {
    const int p = blockIdx.x * kLmThreads + threadIdx.x;
    if (p < W.n_pt) {
      const Vec3 X = load_pt(A, cur, W.pt_off + p);
      o_pt[3 * p] = X.x; o_pt[3 * p + 1] = X.y; o_pt[3 * p + 2] = X.z;
    }
    for (int e = W.pe_off + p; e < W.pe_off + W.n_pe; e += W.nb_pt * kLmThreads) {
      const Vec3 X = load_pt(A, cur, W.pt_off + A.pe_pt[e]);
      const Pose T = load_cam(A, cur, W.cam_off + pt_cam_of<kPkRuntime>(A, e));
      const bool depth_pos = pose_map(T, X).z > 0.0;
      const bool stereo = pt_obs_stereo<kPkRuntime>(A, e);
      o_pe[e - W.pe_off] = (!untouched && !global && (A.pe_chi2[e] > (stereo ? 7.815 : 5.991) || !depth_pos)) ? 1 : 0;      // Optimizer.cc:1290,1305
    }
  } else {
    const int l = (blockIdx.x - W.nb_pt) * kLmThreads + threadIdx.x;
    if (l < W.n_ln) {
      const int g = W.ln_off + l;
      const bool removed = A.ln_removed[g] != 0;
      o_rm[l] = removed;
      if (removed || untouched) {                                       // GetLineData returns false: nothing updated, nothing erased
        for (int k = 0; k < 3; k++) { o_x0[3 * l + k] = A.ln_x0[(size_t)g * 3 + k]; o_dir[3 * l + k] = A.ln_dir[(size_t)g * 3 + k]; }
      } else {
        const LineQ L = load_ln(A, cur, g);
        const Mat3 Rl = line_rotation(L);
        const Vec3 c0 = mat_col(Rl, 0), c1 = mat_col(Rl, 1);
        const Vec3 X1 = L.alpha * c1;
        o_dir[3 * l] = c0.x; o_dir[3 * l + 1] = c0.y; o_dir[3 * l + 2] = c0.z;
        o_x0[3 * l] = X1.x; o_x0[3 * l + 1] = X1.y; o_x0[3 * l + 2] = X1.z;
      }
    }
    for (int e = W.le_off + l; e < W.le_off + W.n_le; e += W.nb_ln * kLmThreads) {
      const int g = W.ln_off + ln_line_of<kPkRuntime>(A, e >> 1);
      const uint8_t fl = A.le_flags[e];
      if (A.ln_removed[g] != 0 || untouched || !(fl & EF_VALID)) { o_le[e - W.le_off] = 0; continue; }
      const LineQ L = load_ln(A, cur, g);
      const Mat3 Rl = line_rotation(L);
      const Vec3 c0 = mat_col(Rl, 0), c1 = mat_col(Rl, 1);
      const Vec3 X1 = L.alpha * c1, X2 = X1 + c0;
      const Pose T = load_cam(A, cur, W.cam_off + ln_cam_of<kPkRuntime>(A, e >> 1));
      const LnSeg sg = ln_seg_of<kPkRuntime>(A, e);
      const double bx = (e & 1) ? cam.bx_right : 0.0;
      const bool depth_pos = line_depth_positive(cam, bx, T, c0, c1, L.alpha, sg.xs, sg.ys, sg.xe, sg.ye);
      double r[2];
      line_residual(cam, bx, pose_map(T, X1), pose_map(T, X2), sg.xs, sg.ys, sg.xe, sg.ye, r, nullptr);
      const double c2 = chi2_of(r, 2, ln_info_of<kPkRuntime>(A, e));
      A.le_chi2[e] = c2;
      const double th = (fl & EF_PAIRSTEREO) ? W.th_ln_stereo : W.th_ln_mono;
      o_le[e - W.le_off] = (!global && (c2 > th * th || !depth_pos)) ? 1 : 0;                    // LineOptimizer.cc:185-196
    }
  }
}

__global__ void ba_mark_done_kernel(BAState* __restrict__ st, int n_windows) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_windows && st[w].phase == PH_FINALIZE) st[w].phase = PH_DONE;
}

}  // namespace lldba
#endif
