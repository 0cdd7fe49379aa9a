// lld_ba.hip — host side of the batched local bundle adjustment: HBM layout, upload, the super-step launch loop and
// read-back.  Kernels live in lld_ba_kernels.h.  Stands in for Optimizer::LocalBundleAdjustment (src/Optimizer.cc:936-1388).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "lld_ba_kernels.h"

using namespace lldba;

namespace {
constexpr int kNumPhases = 5;
constexpr int kFusePairsBelowWindows = 24; // a GROUP of fewer windows than this: the point + line kernels of a pair share one launch (a dependent
                                           // launch less per pair on a chain that is latency-bound anyway: 32 windows = two groups of 16: 2520 -> 2610
                                           // windows/s).  It was 64 until the end of round 3; swept then (LLD_BA_FUSE_BELOW / LLD_BA_CHUNK_FROM): groups of
                                           // 16 - 21 windows do not care, groups of 27 - 48 lose 2.5 - 6 % to the fused kernels and the queued super-steps
                                           // (96 windows = 3 x 32: 4220 -> 4410 windows/s without them, 112: 4400 -> 4560, 192 = 4 x 48: 4850 -> 5120)
constexpr int kMaxSuperSteps = 512;      // hard stop: 2 rounds x 15 iterations x 10 trials is the protocol's own bound (300)
// Super-steps queued per host poll.  A group of few windows is a chain of dependent, latency-bound kernels (a 32-window batch: 390 us per
// super-step, of which ~30 us are the host's wait-read-launch round trip and more are launch gaps): such groups queue kChunkSmall
// super-steps at once - every kernel looks at its window's state first, so a super-step queued for a window that is already done
// falls through - with the round transition (ba_classify + ba_round2) inside every queued super-step.  Groups of >= kChunkFromWindows
// windows keep one poll per super-step: their polls hide behind the other groups' kernels, and two more launches per super-step
// over that many windows cost more than they save (same sweep as kFusePairsBelowWindows).
constexpr int kChunkSmall = 4, kChunkFromWindows = 24;
constexpr int kSchurWindowTile = 8;      // ba_schur_items_both: windows per dispatch tile = XCDs of the device (see the kernel)
// Experiment / debug knobs read from the environment exist only in the experiments build (make exp: -DLLD_EXPERIMENTS ->
// liblld_amd_exp.so, loaded by tools/ and by the two tests that need LLD_BA_FORCE_BIG / LLD_BA_TIMING through LLD_AMD_LIB or abi.Lib).
// The product library reads ONE variable, LLD_HOST_THREADS (host staging threads), documented in include/lld_amd.h.
#ifdef LLD_EXPERIMENTS
static int exp_int(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }
static bool exp_flag(const char* name) { return std::getenv(name) != nullptr; }
static const char* exp_str(const char* name) { return std::getenv(name); }
#else
static constexpr int exp_int(const char*, int dflt) { return dflt; }
static constexpr bool exp_flag(const char*) { return false; }
static constexpr const char* exp_str(const char*) { return nullptr; }
#endif
// A batch of this many windows fills the GPU on its own (four stream groups in flight).  Two such solves interleaved from two
// contexts ran 20 - 30 % slower in aggregate than one after the other (2 lanes: 2930 windows/s host buffers in and out, 3910 with the
// solves taking turns; tools/exp_e2e_lanes.py), so solves of large batches take turns per device; everything else of a pipelined
// caller - the next batch's flattening and upload, the previous batch's download - overlaps the solve in flight.  Small batches and
// single windows (lld_local_ba on the LocalMapping thread next to lld_pose_opt on the Tracking thread) never wait.
constexpr int kSerialiseSolvesFromWindows = 64;
std::mutex g_big_solve[64];               // per HIP device
std::atomic<int> g_big_solves_running[64];   // per HIP device: solves of >= kSerialiseSolvesFromWindows windows in flight
// Host threads that flatten a batch while such a solve runs on the same device (a pipelined caller: the next batch is staged while this one
// solves).  Rounds 4 - 5 capped them at four: next to a solve on four stream groups with six event records per super-step, sixteen staging
// threads made every latency-bound kernel 10 - 100 % slower (3 lanes at 4 / 8 / 16 threads = 5050 / 4690 / 4560 windows/s in steady state).
// With the events gone and two groups for a pipelined batch (ba_make_groups) the solve no longer notices them, and the cap was what held a
// two-lane caller back - tools/experiments/exp_e2e_lanes.py, same box, end to end / steady state: 2 lanes 4020 / 4390 windows/s at 4 threads,
// 4940 / 5700 at 8, 5590 / 6600 at 16; 3 lanes 5810 / 6440, 5960 / 6560, 5950 / 6620.  The constant stays as the experiments build's knob.
constexpr int kStagingThreadsUnderSolve = 16;
static int staging_cap() { static const int c = std::max(1, exp_int("LLD_BA_STAGING_CAP", kStagingThreadsUnderSolve)); return c; }
}

struct lld_ba_batch {
  lld_ctx* ctx = nullptr;
  int n_windows = 0;
  lld_ba_params params;
  std::vector<BAWin> h_wins;
  void* slab = nullptr; bool borrowed = false; size_t slab_bytes = 0;      // borrowed: slab, streams, events and poll block are the context's cached set
  BAArrays A;
  BAWin* d_wins = nullptr; BAState* d_state = nullptr;
  // window groups solved concurrently, each on its own stream (hides the latency-bound reduced solve, the per-super-step
  // host poll and kernel tails behind the other groups' work)
  struct Group { int w0 = 0, nw = 0; hipStream_t st = nullptr; bool own_stream = false; int* d_counters = nullptr; int* h_counters = nullptr;
                 hipEvent_t ev[kChunkSmall][kNumPhases + 1] = {}; int chunk = 1, chunk0 = 1, chunk_from = 0; hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr; int steps = 0; bool active = false; int rows = 0; int map_parity = 0; int max_nt_pt = 0, max_nb_ln = 0, max_nl_pt = 0, max_nl_ln = 0, max_lblocks = 0, max_items_pt = 0, max_items_ln = 0, max_blk = 0; bool any_sparse = false, any_dense = false; };
  std::vector<Group> groups;
  int* d_counters = nullptr; int* h_counters = nullptr;               // device / pinned, 4 ints per group
  int* d_slot_map = nullptr; int* d_active_pub = nullptr;            // per window: grid row -> window map of its group, published "still at work" bits (BAArrays::slot_map)
  int* h_abort = nullptr;                                             // pinned, host-written / device-read: the live stop flag as the control kernel sees it
  int max_lblocks = 0, max_items_pt = 0, max_items_ln = 0, max_free = 0, max_cams = 0, max_blk = 0, acc_copies[2] = {4, 4}, lin_waves[2] = {kLinThreads / 64, kLinThreads / 64};   // [point, line] linearise kernel
  bool pcg_multi = false;
  bool pipelined = false;                                 // created while another batch's large solve ran on this device (ba_make_groups)
  bool failed = false;                                    // a solve returned an error: the device state is mid-trial, every entry point but destroy refuses the batch
  bool big = false;                                       // a map beyond kMaxFreeCamsLds cameras: accumulators and poses of the linearise / back-substitution kernels in HBM
  size_t schur_lds[2] = {0, 0}; size_t schur_wide_lds = 0;
  int chunk_landmarks = 32;
  size_t S_total = 0, x_total = 0;
  size_t rec_stride = 0;
  unsigned char* h_records = nullptr; bool records_pinned_own = false; std::vector<unsigned char> h_records_pageable; bool records_valid = false;
  double phase_ms[LLD_BA_N_PHASES] = {};
  bool phase_events = false;               // lld_ba_batch_set_phase_timing: HIP events at the phase boundaries of every super-step.  Off by default - an event record is a
                                           // barrier packet of ~4 us between two dependent kernels: six per super-step were 15 % of a single window's solve, 10 % at 32
                                           // windows, 1.6 % at 256 (tools/experiments/exp_phase_events.sh)
  int64_t launches[kNumPhases] = {};
  int super_steps = 0;
  std::vector<uint8_t> plan_mode;                         // per window: CholPlan::mode (1: ba_chol_sparse_kernel, 0: ba_chol_mfma_kernel)
};

namespace {

// deep = false: the O(1) part only (counts and pointers), what the layout needs before the windows are walked
int validate_window(const lld_ba_window& w, bool deep = true) {
  if (w.n_cams <= 0 || w.n_free_cams < 0 || w.n_free_cams > w.n_cams || w.n_points < 0 || w.n_lines < 0 || w.n_pt_obs < 0 || w.n_ln_obs < 0)
    return LLD_ERR_INVALID;
  if (!w.cam_qt) return LLD_ERR_INVALID;
  if (w.n_free_cams > kMaxFreeCams) return LLD_ERR_UNSUPPORTED;
  if (w.n_points > 0 && (!w.pt_xyz || !w.pt_obs_start)) return LLD_ERR_INVALID;
  if (w.n_pt_obs > 0 && (!w.pt_obs_cam || !w.pt_obs_uvr || !w.pt_obs_inv_sigma2)) return LLD_ERR_INVALID;
  if (w.n_lines > 0 && (!w.line_x0 || !w.line_dir || !w.ln_obs_start)) return LLD_ERR_INVALID;
  if (w.n_ln_obs > 0 && (!w.ln_obs_cam || !w.ln_obs_left || !w.ln_obs_right || !w.ln_obs_octave)) return LLD_ERR_INVALID;
  if (!deep) return LLD_OK;
  if (w.n_points > 0) {
    if (w.pt_obs_start[0] != 0 || w.pt_obs_start[w.n_points] != w.n_pt_obs) return LLD_ERR_INVALID;
    for (int p = 0; p < w.n_points; p++) if (w.pt_obs_start[p + 1] < w.pt_obs_start[p]) return LLD_ERR_INVALID;
  } else if (w.n_pt_obs != 0) return LLD_ERR_INVALID;
  if (w.n_lines > 0) {
    if (w.ln_obs_start[0] != 0 || w.ln_obs_start[w.n_lines] != w.n_ln_obs) return LLD_ERR_INVALID;
    for (int l = 0; l < w.n_lines; l++) if (w.ln_obs_start[l + 1] < w.ln_obs_start[l]) return LLD_ERR_INVALID;
  } else if (w.n_ln_obs != 0) return LLD_ERR_INVALID;
  for (int o = 0; o < w.n_pt_obs; o++) if (w.pt_obs_cam[o] < 0 || w.pt_obs_cam[o] >= w.n_cams) return LLD_ERR_INVALID;
  for (int o = 0; o < w.n_ln_obs; o++) if (w.ln_obs_cam[o] < 0 || w.ln_obs_cam[o] >= w.n_cams) return LLD_ERR_INVALID;
  return LLD_OK;
}

size_t record_bytes(const BAWin& W) {
  size_t b = sizeof(BARecordHeader) + sizeof(double) * (7 * (size_t)W.n_cams + 3 * (size_t)W.n_pt + 6 * (size_t)W.n_ln) + (size_t)W.n_pe + (size_t)W.n_le + (size_t)W.n_ln;
  return (b + 255) & ~size_t(255);
}


// ---- host staging (see ba_batch_create_impl) ------------------------------------------------------------------------------------
// A host staging array: a view into the pinned upload arena (the staging threads write every element).
template <class T> struct HostBuf {
  T* p = nullptr; size_t n = 0;
  void view(T* at, size_t count) { p = at; n = count; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
};

struct HostArrays {      // the flattened inputs, batch-global indexing (BAArrays' input section); observations in ONE of the two layouts
  bool packed = true;
  HostBuf<double> cam_qt0, pt0, ln_x0, ln_dir, ln_info;
  HostBuf<int> pt_obs_start, ln_obs_start, pe_pt;
  // packed (BAArrays::packed = 1)
  HostBuf<float4> pe_obs, lo_seg;
  HostBuf<int> pe_cs, lo_cs, lo_ln;
  HostBuf<unsigned short> lo_oct;
  // as given
  HostBuf<double> pe_u, pe_v, pe_ur, pe_s, le_xs, le_ys, le_xe, le_ye, le_s;
  HostBuf<int> pe_cam, le_cam, le_ln;
};

struct WinBases { long long NC, NP, NL, NPE, NLO, NF; size_t S_total, x_total; };   // totals of the windows before this one

// Schur chunks of one window and one landmark kind, offsets local to this stage
struct ChunkStage {
  std::vector<SChunk> chunks;
  std::vector<int> sg_lm, sg_tab, sg_cams;
  std::vector<int> blk_key, blk_val, cam_key, cam_val;     // (block | camera, partial index [*4 + mode]) in generation order
  size_t n_part = 0, n_cpart = 0, lds_need = 0, wide_lds_need = 0;
};

struct WinStage {
  std::vector<PTask> ptasks, ltasks;
  std::vector<uint8_t> pt_slot, ln_slot;                   // per landmark: its position in its task (landmark - PTask::l0; 0 for a task of one)
  ChunkStage cs[2];                                        // points, lines
  std::vector<int> blk_start, blk_src, cam_start, cam_src; // window-local CSRs over both kinds (stage_csr)
  std::vector<int> blk_perm;                               // the blocks of S in the order ba_schur_reduce visits them: longest partial list first
  int n_blk_nz = 0;                                        // ... of which the first n_blk_nz are structurally non-zero (the others stay the zeros the batch starts with)
  CholPlan plan;                                           // schedule of the structure-following reduced solve (stage_chol_plan; mode 0: dense kernel)
};

// The slab starts with two sections that come from the host: A = the flattened inputs, B = the Schur / task structures and the
// window headers.  The pinned upload arenas are carved by the SAME sequence of takes, so host and device offsets agree and each
// section travels in ONE host-to-device copy (section A while the host still places section B).
struct SecA {
  double *cam_qt0, *pt0, *ln_x0, *ln_dir, *ln_info; int *pt_obs_start, *ln_obs_start, *pe_pt;
  float4 *pe_obs, *lo_seg; int *pe_cs, *lo_cs, *lo_ln; unsigned short* lo_oct;
  int* pe_cam; double *pe_u, *pe_v, *pe_ur, *pe_s;
  int *le_cam, *le_ln; double *le_xs, *le_ys, *le_xe, *le_ye, *le_s;
};
void carve_a(lld_slab& sl, bool packed, long long NC, long long NP, long long NL, long long NPE, size_t NLE, SecA& a) {
  a = SecA{};
  a.cam_qt0 = sl.take<double>(NC * 7 + 1); a.pt0 = sl.take<double>(NP * 3 + 1); a.ln_x0 = sl.take<double>(NL * 3 + 1); a.ln_dir = sl.take<double>(NL * 3 + 1);
  a.ln_info = sl.take<double>(256);
  a.pt_obs_start = sl.take<int>(NP + 2); a.ln_obs_start = sl.take<int>(NL + 2);
  a.pe_pt = sl.take<int>(NPE + 1);
  if (packed) {
    a.pe_obs = sl.take<float4>(NPE + 1); a.lo_seg = sl.take<float4>(NLE + 2);
    a.pe_cs = sl.take<int>(NPE + 1); a.lo_cs = sl.take<int>(NLE / 2 + 1); a.lo_ln = sl.take<int>(NLE / 2 + 1);
    a.lo_oct = sl.take<unsigned short>(NLE / 2 + 1);
    return;
  }
  a.pe_cam = sl.take<int>(NPE + 1);
  a.pe_u = sl.take<double>(NPE + 1); a.pe_v = sl.take<double>(NPE + 1); a.pe_ur = sl.take<double>(NPE + 1); a.pe_s = sl.take<double>(NPE + 1);
  a.le_cam = sl.take<int>(NLE + 1); a.le_ln = sl.take<int>(NLE + 1);
  a.le_xs = sl.take<double>(NLE + 1); a.le_ys = sl.take<double>(NLE + 1); a.le_xe = sl.take<double>(NLE + 1); a.le_ye = sl.take<double>(NLE + 1);
  a.le_s = sl.take<double>(NLE + 1);
}
struct SecBSizes { size_t blk_start, blk_src, cam_start, cam_src, lm, tab, cams, chunk, ptask, ltask; int n_windows; };
struct SecB { int *blk_start, *blk_src, *cam_start, *cam_src, *blk_perm, *sg_lm, *sg_tab, *sg_cams; SChunk* chunks; PTask *ptasks, *ltasks; BAWin* wins; CholPlan* plans; };
void carve_b(lld_slab& sl, const SecBSizes& z, SecB& b) {
  b.blk_start = sl.take<int>(z.blk_start + 2); b.blk_perm = sl.take<int>(z.blk_start + 2); b.blk_src = sl.take<int>(z.blk_src + 1); b.cam_start = sl.take<int>(z.cam_start + 2); b.cam_src = sl.take<int>(z.cam_src + 1);
  b.sg_lm = sl.take<int>(z.lm + 1); b.sg_tab = sl.take<int>(z.tab + 1); b.sg_cams = sl.take<int>(z.cams + 1);
  b.chunks = sl.take<SChunk>(z.chunk + 1); b.ptasks = sl.take<PTask>(z.ptask + 1); b.ltasks = sl.take<PTask>(z.ltask + 1);
  b.wins = sl.take<BAWin>((size_t)z.n_windows);
  b.plans = sl.take<CholPlan>((size_t)z.n_windows);
}

// Grow-only pinned arena `which` of the context's cache (or, for a batch that does not own the cache, a private one the caller frees).
int stage_arena(lld_ctx* ctx, bool cached, int which, size_t bytes, void** out) {
  if (!cached) { LLD_HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault)); return LLD_OK; }
  lld_ctx::BACache& c = ctx->ba;
  if (c.stage_pending) { LLD_HIP_TRY(hipEventSynchronize(c.stage_free)); c.stage_pending = false; }     // the previous batch's uploads have left
  if (bytes > c.stage_bytes[which]) {
    if (c.stage[which]) LLD_HIP_TRY(hipHostFree(c.stage[which]));
    c.stage[which] = nullptr; c.stage_bytes[which] = 0;
    const size_t want = bytes + (bytes >> 3) + 4096;
    LLD_HIP_TRY(hipHostMalloc(&c.stage[which], want, hipHostMallocDefault));
    c.stage_bytes[which] = want;
  }
  *out = c.stage[which];
  return LLD_OK;
}

// wavefront tasks of the lane-per-edge kernels + everything in BAWin that follows from the window sizes
void stage_tasks(const lld_ba_window& w, const lld_ba_params& P, const WinBases& b, int n_windows, const int* lin_waves, bool det, BAWin& W, WinStage& S) {
  std::memset(&W, 0, sizeof W);
  W.cam = lld::make_camk(w.cam);
  W.n_cams = w.n_cams; W.n_free = w.n_free_cams;
  W.cam_off = (int)b.NC; W.pt_off = (int)b.NP; W.n_pt = w.n_points; W.ln_off = (int)b.NL; W.n_ln = w.n_lines;
  W.pe_off = (int)b.NPE; W.n_pe = w.n_pt_obs; W.le_off = (int)(2 * b.NLO); W.n_le = 2 * w.n_ln_obs;
  W.lo_off = (int)b.NLO; W.n_lo = w.n_ln_obs;
  W.hpp_off = (int)b.NF; W.x_off = (int)b.x_total; W.S_off = (long long)b.S_total;
  W.nb_pt = (w.n_points + kLmThreads - 1) / kLmThreads; W.nb_ln = (w.n_lines + kLmThreads - 1) / kLmThreads;
  // a task = consecutive landmarks while their edges fit into one wavefront; a landmark with more than 64 edges is a task of its own
  auto build = [](int n_lm, const int32_t* start, long long e_base, std::vector<PTask>& out, std::vector<uint8_t>& slot) {
    out.reserve((size_t)(start ? start[n_lm] : 0) / 48 + 8);
    slot.resize((size_t)n_lm);
    for (int l = 0; l < n_lm;) {
      PTask T; std::memset(&T, 0, sizeof T); T.l0 = l; T.e0 = (int)e_base + start[l];
      while (l < n_lm) {
        const int ne = start[l + 1] - start[l];
        if (T.nl > 0 && (T.ne + ne > 64 || T.nl >= 64)) break;
        slot[l] = (uint8_t)(l - T.l0);
        T.nl++; T.ne += ne; T.ms = std::max(T.ms, ne); l++;
        if (T.ne > 64) break;
      }
      out.push_back(T);
    }
  };
  build(w.n_points, w.pt_obs_start, b.NPE, S.ptasks, S.pt_slot);      // lane <-> point edge
  build(w.n_lines, w.ln_obs_start, b.NLO, S.ltasks, S.ln_slot);       // lane <-> (line, KF) observation
  W.n_ptasks = (int)S.ptasks.size(); W.n_ltasks = (int)S.ltasks.size();
  const int* R = n_windows >= kRoundsBigMinWindows ? kRoundsThroughputBig : (n_windows >= kRoundsThroughputMinWindows ? kRoundsThroughput : kRoundsLatency);
  for (int i = 0; i < 4; i++) W.rounds[i] = R[i];
  // experiments: LLD_BA_ROUNDS="lin_pt,lin_ln,backsub_pt,backsub_ln" tasks per wavefront (parsed once per process)
  static const struct RoundsEnv { int r[4]; bool set; RoundsEnv() : r{0, 0, 0, 0}, set(false) {
    if (const char* e = exp_str("LLD_BA_ROUNDS")) set = std::sscanf(e, "%d,%d,%d,%d", &r[0], &r[1], &r[2], &r[3]) == 4; } } rounds_env;
  if (rounds_env.set) for (int i = 0; i < 4; i++) if (rounds_env.r[i] >= 1 && rounds_env.r[i] <= 64) W.rounds[i] = rounds_env.r[i];
  W.nt_pt = (W.n_ptasks + 4 * W.rounds[2] - 1) / (4 * W.rounds[2]); W.nl_pt = (W.n_ptasks + W.rounds[0] * lin_waves[0] - 1) / (W.rounds[0] * lin_waves[0]);
  W.nt_ln = (W.n_ltasks + 4 * W.rounds[3] - 1) / (4 * W.rounds[3]); W.nl_ln = (W.n_ltasks + W.rounds[1] * lin_waves[1] - 1) / (W.rounds[1] * lin_waves[1]);
  W.lin_waves[0] = lin_waves[0]; W.lin_waves[1] = lin_waves[1]; W.det = det ? 1 : 0;
  const double thMono = (double)(float)std::sqrt(5.991), thStereo = (double)(float)std::sqrt(7.815);   // Optimizer.cc:1088-1089
  W.its[0] = P.its_round1; W.its[1] = P.its_round2; W.max_trials = P.max_trials; W.ln_filter = P.ln_filter; W.abort_after = P.abort_after_trials;
  W.th_mono = thMono; W.th_stereo = thStereo;
  W.th_ln_mono = thMono * P.gamma; W.th_ln_stereo = thStereo * P.gamma;            // LineOptimizer.cc:33-35
  W.protocol = P.protocol; W.robust_pts = P.protocol == 1 ? (P.robust_points != 0) : 1;
  if (P.protocol == 1) { W.its[1] = 0; W.th_ln_mono = W.th_ln_stereo = thStereo / 2.0; }   // double thHuberLines = thHuber3D/2.0  (Optimizer.cc:358)
}

// A double that is a float widened (what the reference's image coordinates and level sigmas are): its float, exactly.  Zero and normal
// floats only - NaN, values beyond the float range and subnormals make the batch keep the caller's doubles (BAArrays::packed = 0).
inline bool narrow(double x, float& f) {
  f = (float)x;
  return (double)f == x && (x == 0.0 || std::fabs(x) >= (double)FLT_MIN);
}

// the window's vertices and edges into their slots of the batch-global arrays; false: H.packed and an observation that is no widened float
bool stage_edges(const lld_ba_window& w, const lld_ba_params& P, const WinBases& b, const BAWin& W, const WinStage& S, HostArrays& H) {
  std::copy(w.cam_qt, w.cam_qt + 7 * (size_t)w.n_cams, H.cam_qt0.p + 7 * (size_t)b.NC);
  if (w.n_points) std::copy(w.pt_xyz, w.pt_xyz + 3 * (size_t)w.n_points, H.pt0.p + 3 * (size_t)b.NP);
  if (w.n_lines) {
    std::copy(w.line_x0, w.line_x0 + 3 * (size_t)w.n_lines, H.ln_x0.p + 3 * (size_t)b.NL);
    std::copy(w.line_dir, w.line_dir + 3 * (size_t)w.n_lines, H.ln_dir.p + 3 * (size_t)b.NL);
  }
  if (H.packed && w.n_cams > 0xffffff) return false;
  bool ok = true;
  {
    int* os = H.pt_obs_start.p + b.NP; int* pt = H.pe_pt.p + b.NPE;
    if (H.packed) {
      float4* ob = H.pe_obs.p + b.NPE; int* cs = H.pe_cs.p + b.NPE;
      for (int p = 0; p < w.n_points; p++) {
        os[p] = (int)b.NPE + w.pt_obs_start[p];
        const int slot = (int)S.pt_slot[p] << 24;
        for (int o = w.pt_obs_start[p]; o < w.pt_obs_start[p + 1]; o++) {
          pt[o] = p; cs[o] = w.pt_obs_cam[o] | slot;
          float4 q;
          const bool n0 = narrow(w.pt_obs_uvr[3 * (size_t)o], q.x), n1 = narrow(w.pt_obs_uvr[3 * (size_t)o + 1], q.y), n2 = narrow(w.pt_obs_uvr[3 * (size_t)o + 2], q.z);
          const bool n3 = narrow(w.pt_obs_inv_sigma2[o], q.w);      // (all four are narrowed whatever the others say: q is stored either way)
          ok &= n0 && n1 && n2 && n3;
          ob[o] = q;
        }
      }
    } else {
      int* cam = H.pe_cam.p + b.NPE;
      double* u = H.pe_u.p + b.NPE; double* v = H.pe_v.p + b.NPE; double* ur = H.pe_ur.p + b.NPE; double* s = H.pe_s.p + b.NPE;
      for (int p = 0; p < w.n_points; p++) {
        os[p] = (int)b.NPE + w.pt_obs_start[p];
        for (int o = w.pt_obs_start[p]; o < w.pt_obs_start[p + 1]; o++) {
          cam[o] = w.pt_obs_cam[o]; pt[o] = p;
          u[o] = w.pt_obs_uvr[3 * (size_t)o]; v[o] = w.pt_obs_uvr[3 * (size_t)o + 1]; ur[o] = w.pt_obs_uvr[3 * (size_t)o + 2];
          s[o] = w.pt_obs_inv_sigma2[o];
        }
      }
    }
  }
  if (!ok) return false;
  {
    int* os = H.ln_obs_start.p + b.NL;
    if (H.packed) {
      float4* seg = H.lo_seg.p + 2 * (size_t)b.NLO; int* cs = H.lo_cs.p + b.NLO; int* ln = H.lo_ln.p + b.NLO; unsigned short* oct = H.lo_oct.p + b.NLO;
      for (int l = 0; l < w.n_lines; l++) {
        os[l] = (int)b.NLO + w.ln_obs_start[l];
        const int slot = (int)S.ln_slot[l] << 24;
        for (int o = w.ln_obs_start[l]; o < w.ln_obs_start[l + 1]; o++) {
          const double* Lf = w.ln_obs_left + 4 * (size_t)o; const double* Rt = w.ln_obs_right + 4 * (size_t)o;
          const bool has_right = !(Rt[0] < 0);                                        // startPointX >= 0 (LineOptimizer.cc:60)
          float4 a, c;
          bool all = true;                                                           // (every component is narrowed whatever the others say: a and c are stored either way)
          const double in8[8] = {Lf[0], Lf[1], Lf[2], Lf[3], Rt[0], Rt[1], Rt[2], Rt[3]};
          float* out8[8] = {&a.x, &a.y, &a.z, &a.w, &c.x, &c.y, &c.z, &c.w};
          for (int q = 0; q < 8; q++) { const bool nq = narrow(in8[q], *out8[q]); all = all && nq; }
          ok &= all;
          seg[2 * (size_t)o] = a; seg[2 * (size_t)o + 1] = c;
          cs[o] = w.ln_obs_cam[o] | slot; ln[o] = l;
          const int ol = w.ln_obs_octave[2 * (size_t)o], orr = w.ln_obs_octave[2 * (size_t)o + 1];
          ok &= ol <= 254 && orr <= 254;
          oct[o] = (unsigned short)(std::max(ol, 0) & 255) | (unsigned short)((has_right ? (std::max(orr, 0) & 255) : 255) << 8);       // (line_info of a negative octave = of octave 0)
        }
      }
    } else {
      const size_t e0 = 2 * (size_t)b.NLO;
      int* cam = H.le_cam.p + e0; int* ln = H.le_ln.p + e0;
      double* xs = H.le_xs.p + e0; double* ys = H.le_ys.p + e0; double* xe = H.le_xe.p + e0; double* ye = H.le_ye.p + e0; double* s = H.le_s.p + e0;
      for (int l = 0; l < w.n_lines; l++) {
        os[l] = (int)b.NLO + w.ln_obs_start[l];
        for (int o = w.ln_obs_start[l]; o < w.ln_obs_start[l + 1]; o++) {
          const double* Lf = w.ln_obs_left + 4 * (size_t)o; const double* Rt = w.ln_obs_right + 4 * (size_t)o;
          const bool has_right = !(Rt[0] < 0);
          for (int si = 0; si < 2; si++) {
            const double* kl = si == 0 ? Lf : Rt;
            const bool valid = si == 0 || has_right;
            const size_t e = 2 * (size_t)o + si;
            cam[e] = w.ln_obs_cam[o]; ln[e] = l;
            xs[e] = kl[0]; ys[e] = kl[1]; xe[e] = kl[2]; ye[e] = kl[3];
            s[e] = valid ? (P.protocol == 1 ? 1.0 : lld::line_info(P.gamma, w.ln_obs_octave[2 * (size_t)o + si])) : 0.0;   // AddLineMinimalGlobal: identity
          }
        }
      }
    }
  }
  (void)W;
  return ok;
}

// Schur work items of one landmark kind: sort the landmarks by their set of free cameras, cut the runs into chunks, one item per
// (chunk, slot pair).  Structure only: outlier levels are handled through zeroed Hpl blocks at run time.
void stage_chunks(const lld_ba_window& w, int D, const WinBases& b, int chunk_landmarks, ChunkStage& out) {
  const int n_lm = D == 3 ? w.n_points : w.n_lines;
  if (n_lm == 0) return;
  const int32_t* start = D == 3 ? w.pt_obs_start : w.ln_obs_start;
  const int32_t* ocam = D == 3 ? w.pt_obs_cam : w.ln_obs_cam;
  const long long id_base = D == 3 ? b.NPE : b.NLO, lm_base = D == 3 ? b.NP : b.NL;
  // a landmark's signature = its free cameras in ascending order (ties keep the observation order) with the ids of the
  // matching observations; flat arrays, no per-landmark allocation
  std::vector<int> soff(1, 0), scam, sid, sigs;                    // sigs: landmarks that touch a free camera
  soff.reserve(n_lm + 1); scam.reserve(start[n_lm] - start[0]); sid.reserve(start[n_lm] - start[0]); sigs.reserve(n_lm);
  for (int l = 0; l < n_lm; l++) {
    const int b0 = (int)scam.size();
    for (int o = start[l]; o < start[l + 1]; o++) {
      if (ocam[o] >= w.n_free_cams) continue;
      int at = (int)scam.size();
      scam.push_back(ocam[o]); sid.push_back((int)(id_base + o));
      while (at > b0 && scam[at - 1] > scam[at]) { std::swap(scam[at - 1], scam[at]); std::swap(sid[at - 1], sid[at]); at--; }   // stable insertion
    }
    soff.push_back((int)scam.size());
    if ((int)scam.size() > b0) sigs.push_back(l);
  }
  auto sig_k = [&](int l) { return soff[l + 1] - soff[l]; };
  auto same_cams = [&](int a, int c) {
    if (sig_k(a) != sig_k(c)) return false;
    return std::equal(scam.begin() + soff[a], scam.begin() + soff[a + 1], scam.begin() + soff[c]);
  };
  // Landmarks with equal camera sets must end up adjacent, in landmark order; which set comes first is immaterial.  With at most 64 free
  // cameras and no camera twice in a landmark (every local-BA window) the set is a 64-bit mask and up to eight stable 8-bit radix passes
  // sort the landmarks by it (a comparison sort on the camera lists was half of this function's time); otherwise by the lists.
  bool by_mask = w.n_free_cams <= 64;
  std::vector<unsigned long long> mask;
  if (by_mask) {
    mask.resize(sigs.size());
    for (size_t i = 0; i < sigs.size() && by_mask; i++) {
      unsigned long long m = 0;
      for (int j = soff[sigs[i]]; j < soff[sigs[i] + 1]; j++) { const unsigned long long bit = 1ull << scam[j]; by_mask &= !(m & bit); m |= bit; }
      mask[i] = m;
    }
  }
  if (by_mask) {
    std::vector<int> tmp(sigs.size()); std::vector<unsigned long long> tmask(sigs.size());
    for (int pass = 0; pass < 8; pass++) {
      size_t cnt[257] = {0};
      const int sh = 8 * pass;
      for (size_t i = 0; i < sigs.size(); i++) cnt[((mask[i] >> sh) & 0xff) + 1]++;
      if (cnt[1] == sigs.size()) continue;                       // this digit is zero everywhere (fewer than 8 * pass cameras)
      for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
      for (size_t i = 0; i < sigs.size(); i++) { const size_t at = cnt[(mask[i] >> sh) & 0xff]++; tmp[at] = sigs[i]; tmask[at] = mask[i]; }
      sigs.swap(tmp); mask.swap(tmask);
    }
  } else
  std::stable_sort(sigs.begin(), sigs.end(), [&](int a, int c) {
    const int ka = sig_k(a), kc = sig_k(c);
    if (ka != kc) return ka < kc;
    const int* pa = scam.data() + soff[a]; const int* pc = scam.data() + soff[c];
    for (int i = 0; i < ka; i++) if (pa[i] != pc[i]) return pa[i] < pc[i];
    return false;
  });
  out.sg_lm.reserve(sigs.size()); out.sg_tab.reserve(scam.size());
  size_t i0 = 0;
  while (i0 < sigs.size()) {
    size_t i1 = i0 + 1;
    while (i1 < sigs.size() && i1 - i0 < (size_t)chunk_landmarks && same_cams(sigs[i0], sigs[i1])) i1++;
    SChunk C; std::memset(&C, 0, sizeof C);
    const int* c0cams = scam.data() + soff[sigs[i0]];
    C.k = sig_k(sigs[i0]); C.D = D; C.n_lm = (int)(i1 - i0);
    C.lm_off = (int)out.sg_lm.size(); C.tab_off = (int)out.sg_tab.size(); C.cams_off = (int)out.sg_cams.size();
    out.sg_cams.insert(out.sg_cams.end(), c0cams, c0cams + C.k);
    for (size_t i = i0; i < i1; i++) { out.sg_lm.push_back((int)(lm_base + sigs[i])); out.sg_tab.insert(out.sg_tab.end(), sid.begin() + soff[sigs[i]], sid.begin() + soff[sigs[i] + 1]); }
    C.part_off = (int)out.n_part; C.cpart_off = (int)out.n_cpart;
    int pidx = 0;
    for (int sa = 0; sa < C.k; sa++) {
      const int ca = c0cams[sa];
      out.cam_key.push_back(ca); out.cam_val.push_back((int)out.n_cpart + sa);
      for (int sb = sa; sb < C.k; sb++, pidx++) {
        const int cb = c0cams[sb];                               // cb >= ca (slots are sorted by camera)
        const int mode = ca != cb ? 0 : (sa == sb ? 1 : 2);
        out.blk_key.push_back(cb * (cb + 1) / 2 + ca); out.blk_val.push_back(((int)out.n_part + pidx) * 4 + mode);
      }
    }
    out.n_part += (size_t)C.k * (C.k + 1) / 2; out.n_cpart += C.k;
    if (C.k > kSchurWideK) out.wide_lds_need = std::max(out.wide_lds_need, (size_t)schur_lds_doubles(C.k, D) * sizeof(double));
    else out.lds_need = std::max(out.lds_need, (size_t)schur_lds_doubles(C.k, D) * sizeof(double));   // two staged sub-batches
    out.chunks.push_back(C);
    i0 = i1;
  }
  // Longest first: ba_schur_items takes one wavefront per chunk and its grid runs over the windows fastest, so every window's heaviest chunks are
  // dispatched first and the launch drains on the light ones (a chunk is self-contained - its offsets travel with it - so the order is free).
  // (a map of thousands of keyframes has ~1e5 chunks of one or two landmarks each: nothing to order there)
  if (out.chunks.size() <= 4096) std::stable_sort(out.chunks.begin(), out.chunks.end(), [](const SChunk& a, const SChunk& b) {
    return (long long)a.n_lm * a.k * (a.k + 1) > (long long)b.n_lm * b.k * (b.k + 1); });
}

// Per reduced-system block / per camera: which partials to sum, in generation order (points before lines).  Counting sort of
// the (key, value) pairs both kinds left; the line chunks' partials are numbered after the point chunks'.
void stage_csr(int n_free, WinStage& S) {
  const int nblk = n_free * (n_free + 1) / 2;
  const size_t part3 = S.cs[0].n_part, cpart3 = S.cs[0].n_cpart;
  for (SChunk& c : S.cs[1].chunks) { c.part_off += (int)part3; c.cpart_off += (int)cpart3; }
  auto csr = [&](int n_keys, std::vector<int> ChunkStage::*key, std::vector<int> ChunkStage::*val, size_t off1, std::vector<int>& start, std::vector<int>& src) {
    start.assign((size_t)n_keys, 0);
    std::vector<int> fill((size_t)n_keys + 1, 0);
    for (int d = 0; d < 2; d++) for (int k : S.cs[d].*key) fill[(size_t)k + 1]++;
    for (int k = 0; k < n_keys; k++) { fill[k + 1] += fill[k]; start[k] = fill[k]; }
    src.resize((size_t)fill[n_keys]);
    for (int d = 0; d < 2; d++) {
      const std::vector<int>& K = S.cs[d].*key; const std::vector<int>& V = S.cs[d].*val;
      const int add = d == 0 ? 0 : (int)off1;
      for (size_t i = 0; i < K.size(); i++) src[(size_t)fill[K[i]]++] = V[i] + add;
    }
    for (int d = 0; d < 2; d++) { std::vector<int>().swap(S.cs[d].*key); std::vector<int>().swap(S.cs[d].*val); }
  };
  csr(nblk, &ChunkStage::blk_key, &ChunkStage::blk_val, part3 * 4, S.blk_start, S.blk_src);
  // ba_schur_reduce gives a lane one row of one block and walks that block's partial list: a diagonal block has ~50 entries, most
  // off-diagonal ones none.  In block-index order every wavefront (ten blocks) holds about one diagonal block and waits for its list with a
  // tenth of its lanes (round 5); sorted by list length - a counting sort, stable - the long lists share wavefronts and the empty ones too.
  {
    std::vector<int> len((size_t)nblk);
    int max_len = 0;
    // (a diagonal block is never empty: it carries Hpp + lambda I even when no landmark adds to it)
    for (int b = 0, k = 0; b < n_free; b++)
      for (int a = 0; a <= b; a++, k++) {
        len[k] = (k + 1 < nblk ? S.blk_start[k + 1] : (int)S.blk_src.size()) - S.blk_start[k];
        if (a == b) len[k] = std::max(len[k], 1);
        max_len = std::max(max_len, len[k]);
      }
    S.n_blk_nz = 0;
    for (int k = 0; k < nblk; k++) S.n_blk_nz += len[k] > 0;
    std::vector<int> at((size_t)max_len + 2, 0);
    for (int k = 0; k < nblk; k++) at[(size_t)(max_len - len[k]) + 1]++;
    for (int l = 0; l <= max_len; l++) at[(size_t)l + 1] += at[l];
    S.blk_perm.resize((size_t)nblk);
    for (int k = 0; k < nblk; k++) S.blk_perm[(size_t)at[max_len - len[k]]++] = k;
  }
  csr(n_free, &ChunkStage::cam_key, &ChunkStage::cam_val, cpart3, S.cam_start, S.cam_src);
}

// The symbolic factorisation of the window's reduced camera system (lld_ba_chol_plan.h): block (a, b) of S is structurally non-zero iff some
// landmark is seen by both free cameras, i.e. iff the Schur reduce has a partial to sum into it.  `force`: 0 = the best plan, 1 = natural
// order / one chain, 2 = two chains only, 3 = none (the dense kernel); experiments and tests.
void stage_chol_plan(int n_free, int force, WinStage& S) {
  std::memset(&S.plan, 0, sizeof S.plan);
  if (force == 3 || n_free < 1 || 6 * n_free > kCholMN || n_free > 64) return;
  uint64_t adj[64] = {0};
  const int nblk = n_free * (n_free + 1) / 2;
  for (int b = 0; b < n_free; b++)
    for (int a = 0; a <= b; a++) {
      const int k = b * (b + 1) / 2 + a;
      const int cnt = (k + 1 < nblk ? S.blk_start[k + 1] : (int)S.blk_src.size()) - S.blk_start[k];
      if (cnt > 0 || a == b) { adj[a] |= 1ull << b; adj[b] |= 1ull << a; }
    }
  if (!cholplan::build(n_free, adj, force, S.plan)) std::memset(&S.plan, 0, sizeof S.plan);
}

}  // namespace


static void ba_drop_groups(lld_ba_batch* B) {
  for (auto& G : B->groups) {
    if (G.own_stream && G.st) (void)hipStreamSynchronize(G.st);
    if (G.gexec) { (void)hipGraphExecDestroy(G.gexec); G.gexec = nullptr; }      // (experiments build: LLD_BA_GRAPH)
    if (G.graph) { (void)hipGraphDestroy(G.graph); G.graph = nullptr; }
    if (B->borrowed) continue;                         // streams and events belong to the context's cache
    for (auto& row : G.ev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
    if (G.own_stream && G.st) (void)hipStreamDestroy(G.st);
  }
  B->groups.clear();
}

// Partition the windows into groups, each with its own stream, counters and events.  n_groups 0 -> default: 1 group for tiny
// batches, up to 4 for large ones (the environment variable LLD_BA_GROUPS overrides the default, for experiments).
static int ba_make_groups(lld_ba_batch* B, int n_groups) {
  const int n_windows = B->n_windows;
  int G = n_groups;
  if (G <= 0) {
    // Four groups from 16 windows on, three from 8: swept again in round 5 once the per-phase events had left the solve (they had made every
    // additional chain pay six barrier packets per super-step), tools/experiments/exp_small_sweep2.sh, windows/s with 2 / 3 / 4 groups:
    // 16 windows 2590 / 2680 / 2700, 24: 3330 / 3550 / 3600, 32: 3820 / 3960 / 3890 - 4020, 48: 4370 / 4460 / 4460, 64: 4790 / 4730 / 4760,
    // 96: 5430 / 5530 / 5540, 128: 5730 / 5770 / 5830, 192: 6350 / 6390 / 6410; 8 windows 1510 / 1550 / 1580 (one group: 1450).
    // Five and more fall off a cliff at every size (128 windows: 5820 -> 5020, 256: 6530 -> 5940): the streams then share hardware queues.
    G = n_windows >= 16 ? 4 : (n_windows >= 8 ? 3 : 1);
    // A pipelined caller - several contexts, each looping create -> solve -> download on the same device - has the other contexts' uploads and
    // downloads in flight during this batch's solve.  The part runs four streams side by side and time-slices the rest: with four groups the
    // copies compete with the solve's own chains, with two they have room.  Three lanes x 8 batches of 256 windows, host buffers in and results
    // out (tools/experiments/exp_e2e_lanes.py, same box): 4800 - 4940 windows/s with four groups, 5120 - 5400 with three, 5450 - 5560 with two,
    // 5440 with one; a resident batch alone keeps four (6530 against 6350 with two).  The sign of such a caller: the batch was created while
    // another batch's solve ran on the device.
    if (B->pipelined) G = std::min(G, 2);
    static const int groups_exp = exp_int("LLD_BA_GROUPS", 0);
    if (groups_exp >= 1 && groups_exp <= 8) G = groups_exp;
  }
  G = std::max(1, std::min(std::min(G, 8), n_windows));
  ba_drop_groups(B);
  B->groups.resize(G);
  for (int g = 0; g < G; g++) {
    lld_ba_batch::Group& Gr = B->groups[g];
    Gr.w0 = (int)((long long)n_windows * g / G); Gr.nw = (int)((long long)n_windows * (g + 1) / G) - Gr.w0;
    lld_ctx::BACache& cache = B->ctx->ba;
    if (g == 0) Gr.st = B->ctx->stream;
    else if (B->borrowed) {                            // created once per context (stream / event creation was 18 ms of every batch create)
      if (!cache.streams[g - 1]) LLD_HIP_TRY(hipStreamCreateWithFlags(&cache.streams[g - 1], hipStreamNonBlocking));
      Gr.st = cache.streams[g - 1]; Gr.own_stream = true;
    } else { LLD_HIP_TRY(hipStreamCreateWithFlags(&Gr.st, hipStreamNonBlocking)); Gr.own_stream = true; }
    Gr.d_counters = B->d_counters + 4 * g; Gr.h_counters = B->h_counters + 4 * g;
    static const int chunk_from = exp_int("LLD_BA_CHUNK_FROM", kChunkFromWindows);
    Gr.chunk0 = (B->pcg_multi || Gr.nw >= chunk_from) ? 1 : kChunkSmall; Gr.chunk = Gr.chunk0;
    Gr.chunk_from = chunk_from;
    for (int q = 0; q < Gr.chunk; q++)
      for (int k = 0; k < kNumPhases + 1; k++) {
        if (B->borrowed) { if (!cache.events[g][q][k]) LLD_HIP_TRY(hipEventCreate(&cache.events[g][q][k])); Gr.ev[q][k] = cache.events[g][q][k]; }
        else LLD_HIP_TRY(hipEventCreate(&Gr.ev[q][k]));
      }
    for (int wi = Gr.w0; wi < Gr.w0 + Gr.nw; wi++) {
      const BAWin& W = B->h_wins[wi];
      Gr.max_lblocks = std::max(Gr.max_lblocks, W.nb_pt + W.nb_ln);
      Gr.max_nt_pt = std::max(Gr.max_nt_pt, W.nt_pt); Gr.max_nb_ln = std::max(Gr.max_nb_ln, W.nt_ln);
      Gr.max_nl_pt = std::max(Gr.max_nl_pt, W.nl_pt); Gr.max_nl_ln = std::max(Gr.max_nl_ln, W.nl_ln);
      Gr.max_items_pt = std::max(Gr.max_items_pt, W.n_items_pt); Gr.max_items_ln = std::max(Gr.max_items_ln, W.n_items - W.n_items_pt);
      Gr.max_blk = std::max(Gr.max_blk, B->A.s_skip_empty ? W.n_blk_nz : W.n_free * (W.n_free + 1) / 2);
      const bool sp = (size_t)wi < B->plan_mode.size() && B->plan_mode[wi] == 1;
      Gr.any_sparse = Gr.any_sparse || sp; Gr.any_dense = Gr.any_dense || !sp;
    }
  }
  return LLD_OK;
}

extern "C" {

// The slab, the pinned upload arenas, the group streams / events and the pinned poll block come from the context's cache (grow-only,
// reused by the next batch on this context) unless a live batch already holds them - see lld_ctx::BACache.
// `packed`: flatten the observations as float records (BAArrays::packed); kNotPacked comes back if one of them is no widened float or
// an octave / camera index does not fit the record, and the caller builds the batch again with the doubles as given.
constexpr int kNotPacked = -1000;
static int ba_batch_create_impl(lld_ctx* ctx, int n_windows, const lld_ba_window* wins, const lld_ba_params* params, bool packed, lld_ba_batch** out) {
  if (!ctx || n_windows <= 0 || !wins || !out) return LLD_ERR_INVALID;
  *out = nullptr;
  for (int w = 0; w < n_windows; w++) { int st = validate_window(wins[w], false); if (st) return st; }
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  static const bool timing = exp_flag("LLD_BA_TIMING");
  const auto tc0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { if (timing) std::fprintf(stderr, "[ba_create] %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count()); };
  lld_ba_batch* B = new lld_ba_batch();
  B->ctx = ctx; B->n_windows = n_windows;
  if (params) B->params = *params; else lld_ba_params_default(&B->params);
  const lld_ba_params& P = B->params;
  // (optimize(0) would evaluate no error at all: the classification that follows would read g2o's uninitialised _error vectors)
  if (P.its_round1 < 1 || (P.protocol == 0 && P.its_round2 < 1) || P.its_round2 < 0 || P.max_trials <= 0 || !(P.pcg_rel_tol > 0) || P.reduced_solver < 0 || P.reduced_solver > 5 || P.protocol < 0 || P.protocol > 1 || P.abort_after_trials < 0) { delete B; return LLD_ERR_INVALID; }

  // ---- layout + host staging: the windows are flattened by a few host threads straight into their final positions
  //      (every offset that depends only on the window sizes is known up front), the variable-length Schur structures are
  //      staged per window and placed by a second parallel pass once their sizes are known
  B->h_wins.resize(n_windows);
  // landmarks per Schur chunk: long chunks mean fewer partials to reduce (256: 566 us per Schur launch of 256 windows, 128 and 512: 607),
  // short ones more wavefronts for small batches
  B->chunk_landmarks = n_windows >= 64 ? 256 : (n_windows >= 8 ? 64 : 32);
  { static const int chunk_exp = exp_int("LLD_BA_CHUNK", 0); if (chunk_exp >= 1 && chunk_exp <= 4096) B->chunk_landmarks = chunk_exp; }
  std::vector<WinBases> bases(n_windows + 1);
  {
    WinBases b{};
    for (int wi = 0; wi < n_windows; wi++) {
      bases[wi] = b;
      const lld_ba_window& w = wins[wi];
      b.NC += w.n_cams; b.NP += w.n_points; b.NL += w.n_lines; b.NPE += w.n_pt_obs; b.NLO += w.n_ln_obs; b.NF += w.n_free_cams;
      const size_t n = 6 * (size_t)w.n_free_cams;
      b.S_total += n * n; b.x_total += n;
    }
    bases[n_windows] = b;
  }
  const long long NC = bases[n_windows].NC, NP = bases[n_windows].NP, NL = bases[n_windows].NL, NPE = bases[n_windows].NPE, NLO = bases[n_windows].NLO, NF = bases[n_windows].NF;
  if (NPE > 0x3fffffffll || NLO > 0x1fffffffll || NC * 7 > 0x7fffffffll) { delete B; return LLD_ERR_UNSUPPORTED; }   // 32-bit edge indices
  const size_t S_total = bases[n_windows].S_total, x_total = bases[n_windows].x_total;
  const size_t NLE = 2 * (size_t)NLO;
  // ---- resources: the context's cached set if no live batch holds it, private ones otherwise
  lld_ctx::BACache& cache = ctx->ba;
  const bool cached = !cache.busy.exchange(true);          // test-and-set: a second thread on the same context gets private resources
  void* priv_stage[2] = {nullptr, nullptr};
  bool uploads_queued = false;                             // a DMA out of the pinned arenas is (or may be) in flight on ctx->stream
  auto fail = [&](int st) {
    // nothing may be rewritten or freed under a copy in flight: wait for the stream first
    if (uploads_queued) {
      (void)hipStreamSynchronize(ctx->stream);
      if (B->borrowed) cache.stage_pending = false;        // the arenas are free (an earlier batch's stage_free sits on the same stream)
    }
    for (void* q : priv_stage) if (q) (void)hipHostFree(q);
    ba_drop_groups(B);                                     // private streams / events of a batch that does not own the cache
    if (B->borrowed) cache.busy = false;
    else {
      if (B->h_counters) (void)hipHostFree(B->h_counters);
      if (B->slab) (void)hipFree(B->slab);
    }
    delete B; return st;
  };
  if (cached) B->borrowed = true;
  // ---- section A (flattened inputs) in the pinned arena
  SecA hA{}, dA{};
  lld_slab dryA; dryA.base = reinterpret_cast<char*>(256);
  carve_a(dryA, packed, NC, NP, NL, NPE, NLE, hA);
  const size_t bytesA = dryA.used;
  void* arenaA = nullptr;
  { const int gs = stage_arena(ctx, cached, 0, bytesA, &arenaA); if (gs) return fail(gs); if (!cached) priv_stage[0] = arenaA; }
  { lld_slab sl; sl.base = static_cast<char*>(arenaA); sl.size = bytesA; carve_a(sl, packed, NC, NP, NL, NPE, NLE, hA); }
  HostArrays H;
  H.packed = packed;
  H.cam_qt0.view(hA.cam_qt0, 7 * (size_t)NC); H.pt0.view(hA.pt0, 3 * (size_t)NP); H.ln_x0.view(hA.ln_x0, 3 * (size_t)NL); H.ln_dir.view(hA.ln_dir, 3 * (size_t)NL);
  H.ln_info.view(hA.ln_info, 256);
  H.pt_obs_start.view(hA.pt_obs_start, (size_t)NP + 1); H.ln_obs_start.view(hA.ln_obs_start, (size_t)NL + 1);
  H.pe_pt.view(hA.pe_pt, NPE);
  if (packed) {
    H.pe_obs.view(hA.pe_obs, NPE); H.pe_cs.view(hA.pe_cs, NPE); H.lo_seg.view(hA.lo_seg, NLE); H.lo_cs.view(hA.lo_cs, NLO); H.lo_ln.view(hA.lo_ln, NLO); H.lo_oct.view(hA.lo_oct, NLO);
  } else {
    H.pe_cam.view(hA.pe_cam, NPE); H.pe_u.view(hA.pe_u, NPE); H.pe_v.view(hA.pe_v, NPE); H.pe_ur.view(hA.pe_ur, NPE); H.pe_s.view(hA.pe_s, NPE);
    H.le_cam.view(hA.le_cam, NLE); H.le_ln.view(hA.le_ln, NLE); H.le_xs.view(hA.le_xs, NLE); H.le_ys.view(hA.le_ys, NLE); H.le_xe.view(hA.le_xe, NLE); H.le_ye.view(hA.le_ye, NLE);
    H.le_s.view(hA.le_s, NLE);
  }
  H.pt_obs_start.p[NP] = (int)NPE; H.ln_obs_start.p[NL] = (int)NLO;
  for (int i = 0; i < 255; i++) H.ln_info.p[i] = P.protocol == 1 ? 1.0 : lld::line_info(P.gamma, i);      // (protocol 1: AddLineMinimalGlobal, identity information)
  H.ln_info.p[255] = 0.0;                                                                                   // the right slot of an observation without a right segment
  // ---- what follows from the camera counts alone: where the accumulators of the linearise kernels live and how its workgroups are shaped
  // more cameras than the LDS holds accumulators and pose copies for (a global BA of a long sequence): those live in HBM (BAWin::big)
  bool big_map = false;
  for (int wi = 0; wi < n_windows; wi++) {
    const size_t nf = (size_t)wins[wi].n_free_cams, nc = (size_t)wins[wi].n_cams;
    if (nf > (size_t)kMaxFreeCamsLds || (nf * 27 + 8 + nc * 7) * sizeof(double) > 158 * 1024 || (8 + nc * 14 + nf * 6) * sizeof(double) > 158 * 1024) big_map = true;
    B->max_free = std::max(B->max_free, wins[wi].n_free_cams); B->max_cams = std::max(B->max_cams, wins[wi].n_cams);
  }
  const bool force_big = exp_flag("LLD_BA_FORCE_BIG");            // experiments build, tests: the HBM path on windows of any size
  if (force_big) big_map = true;
  B->big = big_map;
  if (P.deterministic < 0 || P.deterministic > 2) return fail(LLD_ERR_INVALID);
  if (B->big && P.deterministic == 1) return fail(LLD_ERR_UNSUPPORTED);  // HBM accumulators are summed with global atomics (see lld_ba_params::deterministic)
  const bool det = P.deterministic != 0 && !B->big;                 // 2 (the default): wherever the accumulators live in LDS
  // LDS copies of the per-camera accumulators in the linearise kernels: as many as fit (4 for local windows)
  auto lin_lds_bytes = [&](int copies) { return ((size_t)B->max_free * 27 * copies + 8 + (size_t)B->max_cams * 7) * sizeof(double); };
  int copies = B->big ? 1 : kAccCopies;
  while (!B->big && copies > 1 && lin_lds_bytes(copies) > 150 * 1024) copies >>= 1;
  if (!B->big && lin_lds_bytes(copies) > 158 * 1024) return fail(LLD_ERR_UNSUPPORTED);
  B->acc_copies[0] = B->acc_copies[1] = copies;
  // Wavefronts per linearise workgroup.  Bit-reproducible mode: one per accumulator copy (each copy sees one wavefront's adds, in program
  // order).  Shared-accumulator mode: 8 for the point kernel (fewer per-workgroup partials to reduce), and for the line kernel - two
  // wavefronts per SIMD at 256 registers whatever the workgroup size - 4 in large batches: 192 us per launch of 256 windows against 216
  // (profiles/r04_kernel_stats_bench256_1group.txt; small groups share ONE launch between both kinds and keep a common size).
  B->lin_waves[0] = det ? copies : kLinThreads / 64;
  B->lin_waves[1] = det ? copies : (n_windows >= 64 ? 4 : kLinThreads / 64);
  if (const char* e = exp_str("LLD_BA_LIN_WAVES")) {                // experiments: "pt,ln" wavefronts (bit-reproducible mode: = copies)
    int a = 0, c = 0;
    if (std::sscanf(e, "%d,%d", &a, &c) == 2 && a >= 1 && a <= 8 && c >= 1 && c <= 8) {
      B->lin_waves[0] = a; B->lin_waves[1] = c;
      if (det) { B->acc_copies[0] = a; B->acc_copies[1] = c; if (lin_lds_bytes(std::max(a, c)) > 158 * 1024) return fail(LLD_ERR_UNSUPPORTED); }
    }
  }
  std::vector<WinStage> stages(n_windows);
  int n_threads = 1;
  const bool staged_under_solve = g_big_solves_running[ctx->device & 63].load(std::memory_order_relaxed) > 0;
  if (n_windows >= 4) {
    n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    if (g_big_solves_running[ctx->device & 63].load(std::memory_order_relaxed) > 0) n_threads = std::min(n_threads, staging_cap());
    if (const char* e = std::getenv("LLD_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) n_threads = v; }
    // (the cap that protects a solve in flight holds under the override too: bench.py sets LLD_HOST_THREADS for every rank of an N > 1 run)
    if (g_big_solves_running[ctx->device & 63].load(std::memory_order_relaxed) > 0) n_threads = std::min(n_threads, staging_cap());
    n_threads = std::min(n_threads, n_windows);
  }
  std::atomic<int> first_error{LLD_OK};
  // run body(wi) for every window on n_threads host threads (the calling thread is one of them)
  auto for_windows = [&](auto&& body) {
    std::atomic<int> next{0};
    auto worker = [&]() {
      for (;;) {
        const int wi = next.fetch_add(1);
        if (wi >= n_windows || first_error.load() != LLD_OK) break;
        try { body(wi); } catch (...) { int ok = LLD_OK; first_error.compare_exchange_strong(ok, LLD_ERR_ALLOC); }
      }
    };
    std::vector<std::thread> pool;
    try { for (int t = 1; t < n_threads; t++) pool.emplace_back(worker); } catch (...) {}      // fewer threads is fine
    worker();
    for (auto& t : pool) t.join();
  };
  // ---- Section A travels WHILE it is written (round 5).  Its place in the slab - the start - and its size follow from the window sizes alone, and a
  // context that has solved a batch before already owns a slab: the windows are cut into up to eight ranges, and the staging thread that finishes
  // the last window of a range queues that range's slices of every section-A array on the context's stream.  For 256 LBA-B windows the 590 MB
  // upload (12 ms at link speed) used to start when the last window was staged; now it ends about when staging does (one synchronous lane, host
  // buffers in and results out: 3.8 k -> see profiles/NOTES_r05.md).  If the slab turns out too small for the rest of the batch it is re-grown
  // and the whole section goes again, as before.
  void* const early_slab = (cached && n_windows >= 32 && cache.slab && cache.slab_bytes >= bytesA + 4096 && !exp_flag("LLD_BA_POISON") && !exp_flag("LLD_BA_NO_EARLY_UPLOAD")) ? cache.slab : nullptr;
  const int n_ranges = early_slab ? std::min(8, n_windows / 16) : 0;
  std::unique_ptr<std::atomic<int>[]> range_left(n_ranges ? new std::atomic<int>[(size_t)n_ranges] : nullptr);
  auto range_lo = [&](int r) { return (int)((long long)n_windows * r / std::max(1, n_ranges)); };
  for (int r = 0; r < n_ranges; r++) range_left[(size_t)r].store(range_lo(r + 1) - range_lo(r));
  auto copy_rows = [&](const void* hbase, size_t elt, long long lo, long long hi) -> bool {
    if (!hbase || hi <= lo) return true;
    const char* h = static_cast<const char*>(hbase) + elt * (size_t)lo;
    char* d = static_cast<char*>(early_slab) + (h - static_cast<const char*>(arenaA));
    return hipMemcpyAsync(d, h, elt * (size_t)(hi - lo), hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
  };
  auto early_copy = [&](int r) {
    if (hipSetDevice(ctx->device) != hipSuccess) { int ok = LLD_OK; first_error.compare_exchange_strong(ok, LLD_ERR_HIP); return; }
    const WinBases& a = bases[(size_t)range_lo(r)]; const WinBases& b = bases[(size_t)range_lo(r + 1)];
    bool ok = copy_rows(hA.cam_qt0, 7 * sizeof(double), a.NC, b.NC) && copy_rows(hA.pt0, 3 * sizeof(double), a.NP, b.NP) &&
              copy_rows(hA.ln_x0, 3 * sizeof(double), a.NL, b.NL) && copy_rows(hA.ln_dir, 3 * sizeof(double), a.NL, b.NL) &&
              copy_rows(hA.pt_obs_start, sizeof(int), a.NP, b.NP) && copy_rows(hA.ln_obs_start, sizeof(int), a.NL, b.NL) && copy_rows(hA.pe_pt, sizeof(int), a.NPE, b.NPE);
    if (packed)
      ok = ok && copy_rows(hA.pe_obs, sizeof(float4), a.NPE, b.NPE) && copy_rows(hA.pe_cs, sizeof(int), a.NPE, b.NPE) && copy_rows(hA.lo_seg, sizeof(float4), 2 * a.NLO, 2 * b.NLO) &&
           copy_rows(hA.lo_cs, sizeof(int), a.NLO, b.NLO) && copy_rows(hA.lo_ln, sizeof(int), a.NLO, b.NLO) && copy_rows(hA.lo_oct, sizeof(unsigned short), a.NLO, b.NLO);
    else
      ok = ok && copy_rows(hA.pe_cam, sizeof(int), a.NPE, b.NPE) && copy_rows(hA.pe_u, sizeof(double), a.NPE, b.NPE) && copy_rows(hA.pe_v, sizeof(double), a.NPE, b.NPE) &&
           copy_rows(hA.pe_ur, sizeof(double), a.NPE, b.NPE) && copy_rows(hA.pe_s, sizeof(double), a.NPE, b.NPE) &&
           copy_rows(hA.le_cam, sizeof(int), 2 * a.NLO, 2 * b.NLO) && copy_rows(hA.le_ln, sizeof(int), 2 * a.NLO, 2 * b.NLO) &&
           copy_rows(hA.le_xs, sizeof(double), 2 * a.NLO, 2 * b.NLO) && copy_rows(hA.le_ys, sizeof(double), 2 * a.NLO, 2 * b.NLO) && copy_rows(hA.le_xe, sizeof(double), 2 * a.NLO, 2 * b.NLO) &&
           copy_rows(hA.le_ye, sizeof(double), 2 * a.NLO, 2 * b.NLO) && copy_rows(hA.le_s, sizeof(double), 2 * a.NLO, 2 * b.NLO);
    if (!ok) { int e_ = LLD_OK; first_error.compare_exchange_strong(e_, LLD_ERR_HIP); }
  };
  auto window_staged = [&](int wi) {            // (every write into section A of window wi has been made)
    if (!n_ranges) return;
    int r = (int)(((long long)wi * n_ranges) / n_windows);
    while (r + 1 < n_ranges && wi >= range_lo(r + 1)) r++;
    while (r > 0 && wi < range_lo(r)) r--;
    if (range_left[(size_t)r].fetch_sub(1) == 1) early_copy(r);
  };
  if (n_ranges) uploads_queued = true;
  const auto lap1 = [&](const char* what) { if (n_windows == 1) lap(what); };
  for_windows([&](int wi) {
    const int st = validate_window(wins[wi]);
    if (st) { int ok = LLD_OK; first_error.compare_exchange_strong(ok, st); return; }
    BAWin& W = B->h_wins[wi];
    WinStage& S = stages[wi];
    stage_tasks(wins[wi], P, bases[wi], n_windows, B->lin_waves, det, W, S);
    lap1("tasks built");
    if (n_windows == 1 && wins[wi].n_pt_obs + wins[wi].n_ln_obs > 20000) {
      // a single large window: the point chunks on a helper thread, edges and line chunks here
      std::thread helper;
      bool helped = false;
      try { helper = std::thread([&]() { try { stage_chunks(wins[wi], 3, bases[wi], B->chunk_landmarks, S.cs[0]); } catch (...) { int ok = LLD_OK; first_error.compare_exchange_strong(ok, LLD_ERR_ALLOC); } }); helped = true; } catch (...) {}
      if (!stage_edges(wins[wi], P, bases[wi], W, S, H)) { int ok = LLD_OK; first_error.compare_exchange_strong(ok, kNotPacked); }
      lap1("edges flattened");
      stage_chunks(wins[wi], 4, bases[wi], B->chunk_landmarks, S.cs[1]);
      if (helped) helper.join(); else stage_chunks(wins[wi], 3, bases[wi], B->chunk_landmarks, S.cs[0]);
    } else {
      if (!stage_edges(wins[wi], P, bases[wi], W, S, H)) { int ok = LLD_OK; first_error.compare_exchange_strong(ok, kNotPacked); return; }
      lap1("edges flattened");
      window_staged(wi);
      stage_chunks(wins[wi], 3, bases[wi], B->chunk_landmarks, S.cs[0]);
      stage_chunks(wins[wi], 4, bases[wi], B->chunk_landmarks, S.cs[1]);
    }
    stage_csr(wins[wi].n_free_cams, S);
    W.n_blk_nz = S.n_blk_nz;
    {
      static const int plan_force = exp_int("LLD_BA_CHOL_FORCE", 0);          // experiments: 1 natural order / one chain, 2 two chains only, 3 dense kernel
      stage_chol_plan(wins[wi].n_free_cams, P.reduced_solver == 0 ? plan_force : (P.reduced_solver == 4 ? 1 : (P.reduced_solver == 5 ? 2 : 3)), S);
    }
    lap1("chunks built");
  });
  if (first_error.load() != LLD_OK) return fail(first_error.load());
  // ---- where each window's variable-length pieces go
  struct Place { size_t ptask, ltask, chunk, lm, tab, cams, blk_start, blk_src, cam_start, cam_src, part, cpart; };
  std::vector<Place> place(n_windows + 1);
  size_t n_hpart = 0; long long NPART = 0; int max_blk = 0;
  {
    Place q{};
    for (int wi = 0; wi < n_windows; wi++) {
      place[wi] = q;
      const WinStage& S = stages[wi];
      BAWin& W = B->h_wins[wi];
      W.ptask_off = (int)q.ptask; W.ltask_off = (int)q.ltask; W.item_off = (int)q.chunk;
      W.n_items_pt = (int)S.cs[0].chunks.size(); W.n_items = W.n_items_pt + (int)S.cs[1].chunks.size();
      W.blk_csr_off = (int)q.blk_start; W.cam_csr_off = (int)q.cam_start;
      W.hpart_off = (long long)n_hpart; n_hpart += (size_t)(big_map ? 1 : W.nl_pt + W.nl_ln) * W.n_free * 27;
      W.part_off = (int)NPART; NPART += W.nt_pt + W.nt_ln;
      q.ptask += S.ptasks.size(); q.ltask += S.ltasks.size(); q.chunk += (size_t)W.n_items;
      q.lm += S.cs[0].sg_lm.size() + S.cs[1].sg_lm.size(); q.tab += S.cs[0].sg_tab.size() + S.cs[1].sg_tab.size(); q.cams += S.cs[0].sg_cams.size() + S.cs[1].sg_cams.size();
      q.blk_start += S.blk_start.size(); q.blk_src += S.blk_src.size(); q.cam_start += S.cam_start.size(); q.cam_src += S.cam_src.size();
      q.part += S.cs[0].n_part + S.cs[1].n_part; q.cpart += S.cs[0].n_cpart + S.cs[1].n_cpart;
      if (q.part * 4 > 0x7fffffffull || q.tab > 0x7fffffffull) return fail(LLD_ERR_UNSUPPORTED);
      for (int d = 0; d < 2; d++) B->schur_lds[d] = std::max(B->schur_lds[d], S.cs[d].lds_need);
      for (int d = 0; d < 2; d++) B->schur_wide_lds = std::max(B->schur_wide_lds, S.cs[d].wide_lds_need);
      if (B->schur_wide_lds > 64 * 1024) return fail(LLD_ERR_UNSUPPORTED);       // a landmark with > ~450 free observations
      max_blk = std::max(max_blk, W.n_free * (W.n_free + 1) / 2);
      B->max_items_pt = std::max(B->max_items_pt, W.n_items_pt); B->max_items_ln = std::max(B->max_items_ln, W.n_items - W.n_items_pt);
      B->max_lblocks = std::max(B->max_lblocks, W.nb_pt + W.nb_ln);
      B->max_free = std::max(B->max_free, W.n_free);
      B->max_cams = std::max(B->max_cams, W.n_cams);
      B->rec_stride = std::max(B->rec_stride, record_bytes(W));
    }
    place[n_windows] = q;
  }
  const Place& tot = place[n_windows];
  const size_t n_part = tot.part, n_cpart = tot.cpart;
  // few windows whose reduced system is beyond the matrix-core Cholesky: the PCG runs across the whole GPU (see ba_pcgm_*)
  B->pcg_multi = n_windows <= 8 && B->max_free * 6 > kCholMN && P.reduced_solver != 2;
  // The matrix-core solvers only READ S: its structurally empty blocks can stay the zeros the batch starts with (one memset per batch) instead of
  // being rewritten by every ba_schur_reduce launch - three quarters of an LBA-B window's 1275 blocks (round 5).  The vector-ALU Cholesky factors in
  // place and the PCG paths mirror the triangle: they keep the full rewrite.
  const bool s_skip_empty = (P.reduced_solver == 0 || P.reduced_solver >= 3) && B->max_free * 6 <= kCholMN && !B->pcg_multi;
  if (B->max_free > kMaxFreeCamsOneWg && !B->pcg_multi) return fail(LLD_ERR_UNSUPPORTED);   // batches of huge windows: not in this build
  if (B->max_cams > kPcgThreads && !B->pcg_multi) return fail(LLD_ERR_UNSUPPORTED);           // (the one-workgroup solvers move one camera per lane)
  for (int wi = 0; wi < n_windows; wi++) { B->h_wins[wi].acc_copies[0] = B->acc_copies[0]; B->h_wins[wi].acc_copies[1] = B->acc_copies[1]; B->h_wins[wi].win_index = wi; B->h_wins[wi].big = B->big ? 1 : 0; }
  B->max_blk = max_blk;
  // fixed-stride result records (what an RCCL gather of the batch moves)
  for (int wi = 0; wi < n_windows; wi++) B->h_wins[wi].rec_off = (long long)(B->rec_stride * (size_t)wi);
  const size_t rec_total = B->rec_stride * (size_t)n_windows;
  B->S_total = S_total; B->x_total = x_total;
  const SecBSizes zB{tot.blk_start, tot.blk_src, tot.cam_start, tot.cam_src, tot.lm, tot.tab, tot.cams, tot.chunk, tot.ptask, tot.ltask, n_windows};
  // ---- one slab: a dry run of the carve sizes it exactly, the second run assigns the pointers
  hipStream_t st = ctx->stream;
  BAArrays& A = B->A;
  SecB dB{};
  size_t offA = 0, offB = 0, bytesB = 0;
  auto carve = [&](lld_slab& sl) {
    std::memset(&A, 0, sizeof A);
    offA = sl.used; carve_a(sl, packed, NC, NP, NL, NPE, NLE, dA);
    offB = sl.used; carve_b(sl, zB, dB); bytesB = sl.used - offB;
    A.cam_qt0 = dA.cam_qt0; A.pt0 = dA.pt0; A.ln_x0 = dA.ln_x0; A.ln_dir = dA.ln_dir; A.pt_obs_start = dA.pt_obs_start; A.ln_obs_start = dA.ln_obs_start;
    A.pe_pt = dA.pe_pt; A.ln_info = dA.ln_info; A.packed = packed ? 1 : 0;
    A.pe_obs = dA.pe_obs; A.pe_cs = dA.pe_cs; A.lo_seg = dA.lo_seg; A.lo_cs = dA.lo_cs; A.lo_ln = dA.lo_ln; A.lo_oct = dA.lo_oct;
    A.pe_cam = dA.pe_cam; A.pe_u = dA.pe_u; A.pe_v = dA.pe_v; A.pe_ur = dA.pe_ur; A.pe_s = dA.pe_s;
    A.le_cam = dA.le_cam; A.le_ln = dA.le_ln; A.le_xs = dA.le_xs; A.le_ys = dA.le_ys; A.le_xe = dA.le_xe; A.le_ye = dA.le_ye; A.le_s = dA.le_s;
    A.blk_start = dB.blk_start; A.blk_perm = dB.blk_perm; A.blk_src = dB.blk_src; A.cam_start = dB.cam_start; A.cam_src = dB.cam_src;
    A.sg_lm = dB.sg_lm; A.sg_tab = dB.sg_tab; A.sg_cams = dB.sg_cams; A.sg_chunks = dB.chunks; A.ptasks = dB.ptasks; A.ltasks = dB.ltasks;
    B->d_wins = dB.wins; A.chol_plan = dB.plans;
    B->d_state = sl.take<BAState>(n_windows);
    A.NC = NC; A.NP = NP; A.NL = NL;
    A.cam_qt = sl.take<double>(2 * NC * 7 + 1);
    A.ptx = sl.take<double>(2 * NP + 1); A.pty = sl.take<double>(2 * NP + 1); A.ptz = sl.take<double>(2 * NP + 1);
    A.lqx = sl.take<double>(2 * NL + 1); A.lqy = sl.take<double>(2 * NL + 1); A.lqz = sl.take<double>(2 * NL + 1); A.lqw = sl.take<double>(2 * NL + 1); A.lal = sl.take<double>(2 * NL + 1);
    A.pe_flags = sl.take<uint8_t>(NPE + 1); A.le_flags = sl.take<uint8_t>(NLE + 1);
    A.pe_chi2 = sl.take<double>(NPE + 1); A.le_chi2 = sl.take<double>(NLE + 1);
    A.pe_ws = sl.take<double>((size_t)NPE + 1); A.lo_W = sl.take<double>((size_t)NLO * 24 + 1);
    A.pt_active = sl.take<uint8_t>(NP + 1); A.ln_active = sl.take<uint8_t>(NL + 1); A.ln_removed = sl.take<uint8_t>(NL + 1);
    A.pt_V = sl.take<double>((size_t)NP * 9 + 1); A.ln_V = sl.take<double>((size_t)NL * 14 + 1);
    A.hpp_part = sl.take<double>(n_hpart + 2);
    A.Hpp = sl.take<double>((size_t)NF * 21 + 1); A.bp = sl.take<double>((size_t)NF * 6 + 1);
    A.S = sl.take<double>(S_total + 1); A.bschur = sl.take<double>(x_total + 1); A.xp = sl.take<double>(x_total + 1);
    A.x_total = (long long)x_total; A.s_skip_empty = s_skip_empty ? 1 : 0;
    if (B->pcg_multi) { A.pcg_vec = sl.take<double>(4 * x_total + 4); A.pcg_mi = sl.take<double>((size_t)NF * 36 + 1); A.pcg_sc = sl.take<double>(8 * (size_t)n_windows + 8); }
    A.chi_part = sl.take<double>(NPART + 1); A.chi_part2 = sl.take<double>(NPART + 1); A.scale_part = sl.take<double>(NPART + 1);
    A.sp_part = sl.take<double>(n_part * 36 + 2); A.sp_cpart = sl.take<double>(n_cpart * 6 + 2);
    A.records = sl.take<unsigned char>(rec_total + 256);
#ifdef LLD_EXPERIMENTS
    A.chol_stamps = exp_flag("LLD_BA_CHOL_STAMPS") ? sl.take<long long>((size_t)n_windows * kCholStampWaves * kCholStampSlots) : nullptr;
#endif
    B->d_counters = sl.take<int>(4 * 8);
    B->d_slot_map = sl.take<int>(2 * ((size_t)n_windows + 1)); B->d_active_pub = sl.take<int>((size_t)n_windows + 1);
  };
  lap("host staging done");
  lld_slab dry; dry.base = reinterpret_cast<char*>(256);
  carve(dry);
  const size_t bytes = dry.used + 4096;
  bool slab_regrown = false;                          // the early ranges went into an allocation that no longer exists (never inferred from
                                                      // pointer equality: hipMalloc may hand the base just freed back)
  if (cached) {
    if (bytes > cache.slab_bytes) {                   // grow-only (hipFree synchronises the device: it happens only while a context warms up)
      slab_regrown = true;
      void* old_slab = cache.slab;
      cache.slab = nullptr; cache.slab_bytes = 0;          // (before the free: a failure must not leave a dangling pointer in the cache)
      if (old_slab && hipFree(old_slab) != hipSuccess) return fail(LLD_ERR_HIP);
      const size_t want = bytes + (bytes >> 4);
      if (hipMalloc(&cache.slab, want) != hipSuccess) return fail(LLD_ERR_ALLOC);
      cache.slab_bytes = want;
    }
    B->slab = cache.slab;
  } else if (hipMalloc(&B->slab, bytes) != hipSuccess) return fail(LLD_ERR_ALLOC);
  B->slab_bytes = bytes;
  lap("slab ready");
  // debugging aid: every byte of the slab starts as 0xFF (NaN doubles, -1 indices), so a kernel that reads what nothing wrote shows up in the results
  const bool poison = exp_flag("LLD_BA_POISON");
  if (poison && hipMemsetAsync(B->slab, 0xFF, bytes, st) != hipSuccess) return fail(LLD_ERR_HIP);
  lld_slab sl; sl.base = (char*)B->slab; sl.size = bytes;
  carve(sl);
#ifdef LLD_EXPERIMENTS
  if (A.chol_stamps && hipMemsetAsync(A.chol_stamps, 0, sizeof(long long) * (size_t)n_windows * kCholStampWaves * kCholStampSlots, st) != hipSuccess) return fail(LLD_ERR_HIP);
#endif
  if (s_skip_empty && S_total > 0 && hipMemsetAsync(A.S, 0, S_total * sizeof(double), st) != hipSuccess) return fail(LLD_ERR_HIP);
  // section A leaves now and travels while the host places section B - unless its ranges left while they were staged (above): then only what no
  // window owns is still to go (the level table of the line information, the closing entries of the two observation CSRs)
  uploads_queued = true;
  if (n_ranges && !slab_regrown && B->slab == early_slab && offA == 0) {
    bool ok = copy_rows(hA.ln_info, sizeof(double), 0, 256) && copy_rows(hA.pt_obs_start, sizeof(int), NP, NP + 1) && copy_rows(hA.ln_obs_start, sizeof(int), NL, NL + 1);
    if (!ok) return fail(LLD_ERR_HIP);
  } else if (hipMemcpyAsync((char*)B->slab + offA, arenaA, bytesA, hipMemcpyHostToDevice, st) != hipSuccess) return fail(LLD_ERR_HIP);
  lap("inputs queued");
  // ---- section B in its pinned arena: every window writes its pieces straight into their final places
  void* arenaB = nullptr;
  { const int gs = stage_arena(ctx, cached, 1, bytesB, &arenaB); if (gs) return fail(gs); if (!cached) priv_stage[1] = arenaB; }
  SecB hB{};
  B->plan_mode.assign((size_t)n_windows, 0);
  { lld_slab slb; slb.base = static_cast<char*>(arenaB); slb.size = bytesB; carve_b(slb, zB, hB); }
  hB.blk_start[tot.blk_start] = (int)tot.blk_src; hB.cam_start[tot.cam_start] = (int)tot.cam_src;
  for_windows([&](int wi) {
    const Place& q = place[wi];
    WinStage& S = stages[wi];
    std::copy(S.ptasks.begin(), S.ptasks.end(), hB.ptasks + q.ptask);
    std::copy(S.ltasks.begin(), S.ltasks.end(), hB.ltasks + q.ltask);
    size_t at_chunk = q.chunk, at_lm = q.lm, at_tab = q.tab, at_cams = q.cams;
    for (int d = 0; d < 2; d++) {
      const ChunkStage& C = S.cs[d];
      for (SChunk c : C.chunks) {
        c.lm_off += (int)at_lm; c.tab_off += (int)at_tab; c.cams_off += (int)at_cams; c.part_off += (int)q.part; c.cpart_off += (int)q.cpart;      // stage_csr numbered both kinds within the window
        hB.chunks[at_chunk++] = c;
      }
      std::copy(C.sg_lm.begin(), C.sg_lm.end(), hB.sg_lm + at_lm); at_lm += C.sg_lm.size();
      std::copy(C.sg_tab.begin(), C.sg_tab.end(), hB.sg_tab + at_tab); at_tab += C.sg_tab.size();
      std::copy(C.sg_cams.begin(), C.sg_cams.end(), hB.sg_cams + at_cams); at_cams += C.sg_cams.size();
    }
    // the window-local CSRs number their partials from the window's first one
    const int part4 = (int)(q.part * 4), cpart0 = (int)q.cpart, bsrc0 = (int)q.blk_src, csrc0 = (int)q.cam_src;
    for (size_t i = 0; i < S.blk_start.size(); i++) hB.blk_start[q.blk_start + i] = S.blk_start[i] + bsrc0;
    for (size_t i = 0; i < S.blk_perm.size(); i++) hB.blk_perm[q.blk_start + i] = S.blk_perm[i];
    for (size_t i = 0; i < S.blk_src.size(); i++) hB.blk_src[q.blk_src + i] = S.blk_src[i] + part4;
    for (size_t i = 0; i < S.cam_start.size(); i++) hB.cam_start[q.cam_start + i] = S.cam_start[i] + csrc0;
    for (size_t i = 0; i < S.cam_src.size(); i++) hB.cam_src[q.cam_src + i] = S.cam_src[i] + cpart0;
    hB.wins[wi] = B->h_wins[wi];
    hB.plans[wi] = S.plan; B->plan_mode[wi] = S.plan.mode;
    S = WinStage();
  });
  if (first_error.load() != LLD_OK) return fail(first_error.load());
  stages.clear();
  if (hipMemcpyAsync((char*)B->slab + offB, arenaB, bytesB, hipMemcpyHostToDevice, st) != hipSuccess) return fail(LLD_ERR_HIP);
  lap("structures queued");
  if (!cache.attrs_set || !cached) {
    // dynamic LDS ceilings of the kernels that may ask for more than the 64 KiB default: raised once per context to what a workgroup
    // can own on gfx950 (a launch still states the bytes it needs)
    const int lds_max = 159 * 1024;
    const void* fns[] = {reinterpret_cast<const void*>(ba_pcg_kernel), reinterpret_cast<const void*>(ba_backsub_pt_kernel), reinterpret_cast<const void*>(ba_backsub_ln_kernel),
                         reinterpret_cast<const void*>(ba_backsub_ctl_kernel), reinterpret_cast<const void*>(ba_linearize_pt_kernel), reinterpret_cast<const void*>(ba_linearize_ln_kernel),
                         reinterpret_cast<const void*>(ba_linearize_both_kernel), reinterpret_cast<const void*>(ba_backsub_pt_f64_kernel), reinterpret_cast<const void*>(ba_backsub_ln_f64_kernel),
                         reinterpret_cast<const void*>(ba_linearize_pt_f64_kernel), reinterpret_cast<const void*>(ba_linearize_ln_f64_kernel), reinterpret_cast<const void*>(ba_chol_kernel), reinterpret_cast<const void*>(ba_chol_mfma_kernel), reinterpret_cast<const void*>(ba_chol_sparse_kernel),
    };
    for (const void* f : fns) if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess) return fail(LLD_ERR_HIP);
    if (cached) cache.attrs_set = true;
  }
  lap("attributes set");
  if (B->borrowed) {
    if (!ctx->poll && hipHostMalloc(&ctx->poll, 256, hipHostMallocDefault) != hipSuccess) return fail(LLD_ERR_HIP);
    B->h_counters = static_cast<int*>(ctx->poll);
  } else if (hipHostMalloc((void**)&B->h_counters, 256, hipHostMallocDefault) != hipSuccess) return fail(LLD_ERR_HIP);
  B->h_abort = B->h_counters + 48;                            // (the pinned block is 256 bytes: 4 ints per group x 8 groups, then the live stop word)
  lap("pinned counters");
  // a batch that was staged under another batch's solve belongs to a pipelined caller (see ba_make_groups)
  B->pipelined = n_windows >= kSerialiseSolvesFromWindows && (staged_under_solve || g_big_solves_running[ctx->device & 63].load(std::memory_order_relaxed) > 0);
  { int gs = ba_make_groups(B, 0); if (gs) return fail(gs); }
  lap("groups made");
  if (cached) {
    // the arenas are free again once both copies have left; the next create on this context waits for that, not this one
    if (!cache.stage_free && hipEventCreateWithFlags(&cache.stage_free, hipEventDisableTiming) != hipSuccess) return fail(LLD_ERR_HIP);
    if (hipEventRecord(cache.stage_free, st) != hipSuccess) return fail(LLD_ERR_HIP);
    cache.stage_pending = true;
  } else {
    if (hipStreamSynchronize(st) != hipSuccess) return fail(LLD_ERR_HIP);
    for (void*& q : priv_stage) if (q) { (void)hipHostFree(q); q = nullptr; }
  }
  lap("done");
  *out = B;
  return LLD_OK;
}

// Observations as float records first (every value the reference hands over is a float widened to double); a batch with an observation
// that is not one is built again from the doubles as given (experiments build: LLD_BA_OBS_F64=1 goes there directly).
static int ba_batch_create_any(lld_ctx* ctx, int n_windows, const lld_ba_window* wins, const lld_ba_params* params, lld_ba_batch** out) {
  const bool f64_only = exp_flag("LLD_BA_OBS_F64");              // (read per call: tests/test_gpu_ba.py compares the two layouts in one process)
  int st = f64_only ? kNotPacked : ba_batch_create_impl(ctx, n_windows, wins, params, true, out);
  if (st == kNotPacked) st = ba_batch_create_impl(ctx, n_windows, wins, params, false, out);
  return st;
}

int lld_ba_batch_create(lld_ctx* ctx, int n_windows, const lld_ba_window* wins, const lld_ba_params* params, lld_ba_batch** out) {
  return ba_batch_create_any(ctx, n_windows, wins, params, out);
}

// The reference's pbStopFlag is a plain `bool*` (LocalMapping::mbAbortBA, written by the Tracking thread): the byte form of the
// flag lets an adapter pass that pointer as it is (character types may alias any object).
struct StopFlag {
  volatile const int* i32; volatile const unsigned char* u8;
  bool up() const { return (i32 && *i32) || (u8 && *u8); }
};

// The solve proper.  t_begin / t_end belong to the caller (ba_batch_solve_impl), which also owns what happens when this returns an error.
static int ba_batch_solve_body(lld_ba_batch* B, StopFlag abort_flag, hipEvent_t t_begin, hipEvent_t t_end) {
  lld_ctx* ctx = B->ctx;
  std::unique_lock<std::mutex> turn(g_big_solve[ctx->device & 63], std::defer_lock);
  struct RunningGuard { std::atomic<int>* c; ~RunningGuard() { if (c) c->fetch_sub(1, std::memory_order_relaxed); } } running{nullptr};
  if (B->n_windows >= kSerialiseSolvesFromWindows) {
    turn.lock();
    running.c = &g_big_solves_running[ctx->device & 63]; running.c->fetch_add(1, std::memory_order_relaxed);
  }
  BAArrays& A = B->A;
  B->records_valid = false;
  for (auto& m : B->phase_ms) m = 0.0;
  for (auto& l : B->launches) l = 0;
  B->super_steps = 0;
  LLD_HIP_TRY(hipEventRecord(t_begin, ctx->stream));
  const size_t lin_lds_pt = B->big ? 64 : ((size_t)B->max_free * 27 * B->acc_copies[0] + 8 + (size_t)B->max_cams * 7) * sizeof(double);
  const size_t lin_lds_ln = B->big ? 64 : ((size_t)B->max_free * 27 * B->acc_copies[1] + 8 + (size_t)B->max_cams * 7) * sizeof(double);
  const bool same_lin_shape = B->lin_waves[0] == B->lin_waves[1] && B->acc_copies[0] == B->acc_copies[1];      // the fused point + line launch needs ONE workgroup shape
  const size_t bs_lds = B->big ? 64 : (8 + (size_t)B->max_cams * 14 + (size_t)B->max_free * 6) * sizeof(double);
  const size_t pcg_lds = ((size_t)B->max_free * 6 * 4 + kPcgThreads + (size_t)B->max_free * 36 + 32) * sizeof(double);
  const size_t chol_fixed = ((size_t)B->max_free * 36 * 2 + (size_t)B->max_free * 6 * 2 + 32) * sizeof(double);
  // whatever LDS is left (a workgroup may own up to 160 KiB) holds the trailing block triangle of S
  const size_t chol_tri = std::min((size_t)B->max_free * (B->max_free + 1) / 2 * 36 * sizeof(double), (size_t)(156 * 1024) - chol_fixed) / 288 * 288;
  const size_t chol_lds = chol_fixed + chol_tri;
  // Optimizer.cc:1220-1222: a stop request before optimising returns without touching the map -> the read-back kernel copies
  // the (untouched) working state and every flag stays clear.
  const bool abort_at_start = abort_flag.up();
  *B->h_abort = 0;
  const bool live_flag = abort_flag.i32 != nullptr || abort_flag.u8 != nullptr;
  using Group = lld_ba_batch::Group;

  auto finalize_group = [&](Group& G) {
    const BAWin* dw = B->d_wins + G.w0; BAState* ds = B->d_state + G.w0;
    hipLaunchKernelGGL(ba_finalize_kernel, dim3(G.max_lblocks + 1, G.nw), dim3(kLmThreads), 0, G.st, A, dw, ds);
    hipLaunchKernelGGL(ba_mark_done_kernel, dim3((G.nw + 63) / 64), dim3(64), 0, G.st, ds, G.nw);
  };
  // one super-step of one group: linearise (windows that need it) -> Schur -> reduced solve -> back-substitution + trial chi2
  // -> LM control; then the three phase counters travel to pinned host memory and ev[5] marks the end.
  // Grid rows of a super-step = the windows of the group that were still at work at the last poll, mapped to windows on the device
  // (BAArrays::slot_map); the PCG paths keep one row per window.
  const bool use_slots = !(B->params.reduced_solver == 1 || B->pcg_multi);
  // The row -> window map is double buffered: the kernels of a super-step READ the map the previous super-step's control wrote, the control
  // of this super-step WRITES the other buffer (workgroups of the fused back-substitution + control launch may still be dispatched - and read
  // their row - after the group's last control wavefront has rebuilt the map: nothing orders them).  The parity flips with every launch.
  auto group_arrays = [&](const Group& G) { BAArrays Ag = B->A; int* rd = B->d_slot_map + (size_t)G.map_parity * ((size_t)B->n_windows + 1) + G.w0;
                                            int* wr = B->d_slot_map + (size_t)(G.map_parity ^ 1) * ((size_t)B->n_windows + 1) + G.w0;
                                            Ag.slot_map = use_slots ? wr : nullptr; Ag.active_pub = use_slots ? B->d_active_pub + G.w0 : nullptr;
                                            Ag.slot_rd = (use_slots && G.rows < G.nw) ? rd : nullptr; return Ag; };
  // Experiments build, LLD_BA_GRAPH=1: the queued super-steps of a small group (28 launches) are captured once per batch and group into a
  // hipGraph and replayed per poll (all rows every time: the grid must not change).  Measured, not the default: profiles/NOTES_r05.md.
  static const bool use_graph = exp_flag("LLD_BA_GRAPH");
  bool capturing = false;
  int n_superstep_launches = 0;
  const int fail_at = exp_int("LLD_BA_FAIL_AT_SUPERSTEP", -1);      // (read per solve, so that one test process can set and clear it)
  auto launch_superstep = [&](Group& G, int q) -> int {
    const BAWin* dw = B->d_wins + G.w0; BAState* ds = B->d_state + G.w0;
    // experiments build: LLD_BA_FAIL_AT_SUPERSTEP=n makes the n-th super-step launch of a solve fail the way an inexpressible grid does,
    // with the other groups' kernels in flight (tests/test_gpu_ba.py: the error contract of a solve)
    if (fail_at >= 0 && n_superstep_launches++ == fail_at) return LLD_ERR_UNSUPPORTED;
    const BAArrays A = group_arrays(G);                                // (shadows the batch's arrays: every launch below is per group)
    const int nw = use_slots ? std::max(1, std::min(G.rows, G.nw)) : G.nw; hipStream_t st = G.st;
    const int abort_now = abort_flag.up() ? 1 : 0;
    hipEvent_t* ev = G.ev[q];
    const bool tev = B->phase_events && !capturing;                    // per-phase HIP events (lld_ba_batch_phase_ms); ev[5], the end of the super-step, is always recorded
    if (tev) LLD_HIP_TRY(hipEventRecord(ev[0], st));
    static const int fuse_below = exp_int("LLD_BA_FUSE_BELOW", kFusePairsBelowWindows);
    const bool packed = A.packed != 0;                                 // the layout of the observations picks the kernel variant (lld_ba_kernels.h: kPk)
    const bool fuse_pairs = nw < fuse_below && !B->big && packed;      // see ba_linearize_both_kernel
    static const int fuse_bs_below = exp_int("LLD_BA_FUSE_BS_BELOW", -1);   // experiments: the back-substitution pair (+ control) fused up to another group size than the linearisation pair
    const bool fuse_bs = fuse_bs_below >= 0 ? (nw < fuse_bs_below && !B->big && packed) : fuse_pairs;
    if (B->big) {
      if (G.max_nl_pt > 0) hipLaunchKernelGGL(ba_linearize_pt_big_kernel, dim3(G.max_nl_pt, nw), dim3(64 * B->lin_waves[0]), lin_lds_pt, st, A, dw, ds);
      if (G.max_nl_ln > 0) hipLaunchKernelGGL(ba_linearize_ln_big_kernel, dim3(G.max_nl_ln, nw), dim3(64 * B->lin_waves[1]), lin_lds_ln, st, A, dw, ds);
    } else if (fuse_pairs && same_lin_shape && G.max_nl_pt > 0 && G.max_nl_ln > 0) hipLaunchKernelGGL(ba_linearize_both_kernel, dim3(G.max_nl_pt + G.max_nl_ln, nw), dim3(64 * B->lin_waves[0]), lin_lds_pt, st, A, dw, ds, G.max_nl_pt);
    else {
      if (G.max_nl_pt > 0) hipLaunchKernelGGL(packed ? ba_linearize_pt_kernel : ba_linearize_pt_f64_kernel, dim3(G.max_nl_pt, nw), dim3(64 * B->lin_waves[0]), lin_lds_pt, st, A, dw, ds);
      if (G.max_nl_ln > 0) hipLaunchKernelGGL(packed ? ba_linearize_ln_kernel : ba_linearize_ln_f64_kernel, dim3(G.max_nl_ln, nw), dim3(64 * B->lin_waves[1]), lin_lds_ln, st, A, dw, ds);
    }
    hipLaunchKernelGGL(ba_hpp_reduce_kernel, dim3(std::max(1, (B->max_free * 27 + 255) / 256), nw), dim3(256), 0, st, A, dw, ds);
    if (tev) LLD_HIP_TRY(hipEventRecord(ev[1], st));
    static const int schur_tile = std::max(1, exp_int("LLD_BA_SCHUR_TILE", kSchurWindowTile));
    static const bool split_schur = exp_flag("LLD_BA_SPLIT_SCHUR");             // experiments: the two launches of before
    if (split_schur) {
      if (G.max_items_pt > 0) hipLaunchKernelGGL(ba_schur_items_kernel<3>, dim3(G.max_items_pt, nw), dim3(kSchurThreads), B->schur_lds[0], st, A, dw, ds);
      if (G.max_items_ln > 0) hipLaunchKernelGGL(ba_schur_items_kernel<4>, dim3(G.max_items_ln, nw), dim3(kSchurThreads), B->schur_lds[1], st, A, dw, ds);
    } else if (G.max_items_pt + G.max_items_ln > 0) {
      if ((long long)nw * (G.max_items_pt + G.max_items_ln) * kSchurThreads > 0xffffffffll) return LLD_ERR_UNSUPPORTED;      // (one workgroup per window and chunk; an AQL packet carries the grid as a 32-bit count of work-ITEMS: 2^26 workgroups of 64)
      hipLaunchKernelGGL(ba_schur_items_both_kernel, dim3((unsigned)((long long)nw * (G.max_items_pt + G.max_items_ln))), dim3(kSchurThreads), std::max(B->schur_lds[0], B->schur_lds[1]), st, A, dw, ds, G.max_items_pt, schur_tile, nw, G.max_items_pt + G.max_items_ln);
    }
    if (B->schur_wide_lds > 0) hipLaunchKernelGGL(ba_schur_wide_kernel, dim3(G.max_items_pt + G.max_items_ln, nw), dim3(kSchurWideThreads), B->schur_wide_lds, st, A, dw, ds);
    hipLaunchKernelGGL(ba_schur_reduce_kernel, dim3((std::max(1, G.max_blk) * 6 + 255) / 256 + 2, nw), dim3(256), 0, st, A, dw, ds);
    if (B->params.reduced_solver == 1 || B->pcg_multi) hipLaunchKernelGGL(ba_symmetrize_kernel, dim3(B->pcg_multi ? 256 : 16, nw), dim3(256), 0, st, A, dw, ds);
    if (tev) LLD_HIP_TRY(hipEventRecord(ev[2], st));
    if (B->pcg_multi) {
      // block-Jacobi PCG with the matrix-vector product spread over the GPU; the host looks at the `done` scalars every 16 iterations
      hipLaunchKernelGGL(ba_pcgm_init_kernel, dim3(nw), dim3(kPcgThreads), 0, st, A, dw, ds, B->params.pcg_rel_tol);
      const int n_max = B->max_free * 6, limit = B->params.pcg_max_iter > 0 ? B->params.pcg_max_iter : 10 * n_max;
      std::vector<double> hsc(8 * (size_t)nw);
      for (int it = 0; it < limit;) {
        for (int k = 0; k < 16 && it < limit; k++, it++) {
          hipLaunchKernelGGL(ba_pcgm_matvec_kernel, dim3((n_max + 3) / 4, nw), dim3(256), 0, st, A, dw);
          hipLaunchKernelGGL(ba_pcgm_update_kernel, dim3(nw), dim3(kPcgThreads), 0, st, A, dw, B->params.pcg_max_iter);
        }
        LLD_HIP_TRY(hipMemcpyAsync(hsc.data(), A.pcg_sc + 8 * (size_t)G.w0, hsc.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        LLD_HIP_TRY(hipStreamSynchronize(st));
        bool all_done = true;
        for (int wI = 0; wI < nw; wI++) all_done = all_done && hsc[8 * (size_t)wI + 3] != 0.0;
        if (all_done) break;
      }
      hipLaunchKernelGGL(ba_pcgm_final_kernel, dim3(nw), dim3(kPcgThreads), 0, st, A, dw, ds);
    } else if (B->params.reduced_solver == 1)
      hipLaunchKernelGGL(ba_pcg_kernel, dim3(nw), dim3(kPcgThreads), pcg_lds, st, A, dw, ds, B->params.pcg_rel_tol, B->params.pcg_max_iter);
    else if ((B->params.reduced_solver == 0 || B->params.reduced_solver >= 3) && B->max_free * 6 <= kCholMN) {    // register-resident tiles on the fp64 matrix cores
      // windows with a plan (lld_ba_chol_plan.h) factor along the structure of S; a group that holds both kinds launches both kernels and each
      // leaves the other's windows alone
      if (G.any_sparse) hipLaunchKernelGGL(ba_chol_sparse_kernel, dim3(nw), dim3(kSpThreads), kSpLdsBytes, st, A, dw, ds);
      if (G.any_dense) hipLaunchKernelGGL(ba_chol_mfma_kernel, dim3(nw), dim3(kCholMThreads), kCholMLdsDoubles * sizeof(double), st, A, dw, ds);
    }
    else
      hipLaunchKernelGGL(ba_chol_kernel, dim3(nw), dim3(kPcgThreads), chol_lds, st, A, dw, ds, (int)(chol_tri / sizeof(double)));
    if (tev) LLD_HIP_TRY(hipEventRecord(ev[3], st));
    bool control_fused = false;
    if (B->big) {
      if (G.max_nt_pt > 0) hipLaunchKernelGGL(ba_backsub_pt_big_kernel, dim3(G.max_nt_pt, nw), dim3(kLmThreads), bs_lds, st, A, dw, ds);
      if (G.max_nb_ln > 0) hipLaunchKernelGGL(ba_backsub_ln_big_kernel, dim3(G.max_nb_ln, nw), dim3(kLmThreads), bs_lds, st, A, dw, ds);
    } else if (fuse_bs && G.max_nt_pt > 0 && G.max_nb_ln > 0) {
      // small groups: both landmark kinds AND the LM control (run by each window's last workgroup) in one launch
      hipLaunchKernelGGL(ba_backsub_ctl_kernel, dim3(G.max_nt_pt + G.max_nb_ln, nw), dim3(kLmThreads), bs_lds, st, A, dw, ds, G.max_nt_pt, abort_now, G.nw, G.d_counters, G.h_counters,
                         (live_flag && G.chunk > 1) ? B->h_abort : nullptr);
      control_fused = true;
    }
    else {
      if (G.max_nt_pt > 0) hipLaunchKernelGGL(packed ? ba_backsub_pt_kernel : ba_backsub_pt_f64_kernel, dim3(G.max_nt_pt, nw), dim3(kLmThreads), bs_lds, st, A, dw, ds);
      if (G.max_nb_ln > 0) hipLaunchKernelGGL(packed ? ba_backsub_ln_kernel : ba_backsub_ln_f64_kernel, dim3(G.max_nb_ln, nw), dim3(kLmThreads), bs_lds, st, A, dw, ds);
    }
    if (tev) LLD_HIP_TRY(hipEventRecord(ev[4], st));
    if (!control_fused)
      hipLaunchKernelGGL(ba_control_kernel, dim3(nw), dim3(kCtlThreads), 0, st, A, dw, ds, abort_now, G.nw, G.d_counters, G.h_counters, (live_flag && G.chunk > 1) ? B->h_abort : nullptr);   // totals land in pinned host memory
    if (G.chunk > 1) {                                    // the round transition rides along (windows in PH_TRANSITION only)
      hipLaunchKernelGGL(ba_classify_kernel, dim3(std::max(1, G.max_lblocks), nw), dim3(kLmThreads), 0, st, A, dw, ds);      // (its last workgroup per window starts round 2)
    }
    LLD_HIP_TRY(hipGetLastError());
    if (!capturing && (tev || q == G.chunk - 1)) LLD_HIP_TRY(hipEventRecord(ev[5], st));
    G.map_parity ^= 1;                                   // the next launch reads the map this one's control wrote
    return LLD_OK;
  };
  auto launch_chunk = [&](Group& G) -> int {
    if (use_graph && G.chunk > 1 && (G.chunk & 1) == 0) {
      if (!G.gexec) {
        G.rows = G.nw;
        LLD_HIP_TRY(hipStreamBeginCapture(G.st, hipStreamCaptureModeThreadLocal));
        capturing = true;
        int s_ = LLD_OK;
        for (int q = 0; q < G.chunk && s_ == LLD_OK; q++) s_ = launch_superstep(G, q);
        capturing = false;
        LLD_HIP_TRY(hipStreamEndCapture(G.st, &G.graph));
        if (s_) return s_;
        LLD_HIP_TRY(hipGraphInstantiate(&G.gexec, G.graph, nullptr, nullptr, 0));
      }
      LLD_HIP_TRY(hipGraphLaunch(G.gexec, G.st));
      LLD_HIP_TRY(hipEventRecord(G.ev[G.chunk - 1][5], G.st));
      return LLD_OK;
    }
    for (int q = 0; q < G.chunk; q++) { const int s_ = launch_superstep(G, q); if (s_) return s_; }
    return LLD_OK;
  };

  for (Group& G : B->groups) {
    const BAWin* dw = B->d_wins + G.w0; BAState* ds = B->d_state + G.w0;
    if (G.own_stream) LLD_HIP_TRY(hipStreamWaitEvent(G.st, t_begin, 0));
    G.steps = 0; G.active = !abort_at_start; G.rows = G.nw; G.chunk = G.chunk0; G.map_parity = 0;
    LLD_HIP_TRY(hipMemsetAsync(G.d_counters, 0, 4 * sizeof(int), G.st));     // the control kernel leaves them at zero after every super-step
    hipLaunchKernelGGL(ba_init_kernel, dim3(std::max(1, std::min(64, G.max_lblocks + 1)), G.nw), dim3(kLmThreads), 0, G.st, group_arrays(G), dw, ds);
    LLD_HIP_TRY(hipGetLastError());
    if (abort_at_start) {
      // every window: phase FINALIZE with aborted = 1; the read-back emits untouched states with clear flags
      std::vector<BAState> hs(G.nw);
      std::memset(hs.data(), 0, sizeof(BAState) * G.nw);
      for (auto& s : hs) { s.phase = PH_FINALIZE; s.aborted = 1; }
      LLD_HIP_TRY(hipMemcpyAsync(ds, hs.data(), sizeof(BAState) * G.nw, hipMemcpyHostToDevice, G.st));
      LLD_HIP_TRY(hipStreamSynchronize(G.st));
      finalize_group(G);
    } else {
      int s = launch_chunk(G); if (s) return s;
    }
  }
  // round-robin over the groups: wait for a group's super-step, read its counters, queue its next one; the other groups'
  // kernels keep the GPU busy meanwhile
  for (bool any = !abort_at_start; any;) {
    any = false;
    for (Group& G : B->groups) {
      if (!G.active) continue;
      // A group that queues several super-steps per poll samples *abort_flag only once per chunk at launch time; the reference polls
      // terminate() on every LM trial (optimization_algorithm_levenberg.cpp:149).  While this thread waits for the chunk it therefore
      // keeps looking at the caller's flag and forwards it through a pinned word that every ba_control_kernel reads: a raised flag
      // is honoured by the NEXT control kernel that runs, chunked or not (the word is device-visible host memory, like the counters).
      if (live_flag && G.chunk > 1) {
        for (;;) {
          const hipError_t q = hipEventQuery(G.ev[G.chunk - 1][5]);
          if (q == hipSuccess) break;
          if (q != hipErrorNotReady) LLD_HIP_TRY(q);
          if (abort_flag.up()) __atomic_store_n(B->h_abort, 1, __ATOMIC_RELEASE);
          std::this_thread::yield();
        }
      } else LLD_HIP_TRY(hipEventSynchronize(G.ev[G.chunk - 1][5]));
      for (int q = 0; q < G.chunk; q++)
        for (int k = 0; k < kNumPhases; k++) {
          float ms = 0.f;
          if (B->phase_events && hipEventElapsedTime(&ms, G.ev[q][k], G.ev[q][k + 1]) == hipSuccess) B->phase_ms[k] += ms;
          B->launches[k]++;
        }
      G.steps += G.chunk; B->super_steps += G.chunk;
      const int n_run = G.h_counters[0], n_trans = G.h_counters[1], n_fin = G.h_counters[2];
      const BAWin* dw = B->d_wins + G.w0; BAState* ds = B->d_state + G.w0;
      if (use_slots && !(use_graph && G.gexec)) G.rows = n_run + n_trans;      // the control kernel left exactly these windows in the group's row map
      // The tail of a large solve - the few windows with rejected trials, a tenth of the solve's time at a tenth of the chip - no longer
      // holds the device's turn: the next lane's solve starts under it (host-buffer pipeline: lld_ba_batch_solve calls from other contexts).
      if (turn.owns_lock() && use_slots) {
        int left = 0;
        for (const Group& Gq : B->groups) left += Gq.active ? Gq.rows : 0;
        if (left * 8 <= B->n_windows) turn.unlock();      // (an eighth, a quarter or half of the windows left: 5550 - 5700 windows/s in steady state either way, 5300 without - tools/exp_ab_e2e.sh)
      }
      if (n_trans > 0 && G.chunk == 1) {
        hipLaunchKernelGGL(ba_classify_kernel, dim3(std::max(1, G.max_lblocks), use_slots ? std::max(1, G.rows) : G.nw), dim3(kLmThreads), 0, G.st, group_arrays(G), dw, ds);
      }
      (void)n_fin;                             // finished windows wait for the group's trailing read-back (one launch instead of one per super-step that finished a window)
      LLD_HIP_TRY(hipGetLastError());
      // a window whose classification leaves an empty active set goes straight to FINALIZE: the trailing read-back picks it up
      // (the tail of a large group - the few windows with rejected trials - already runs the fused launches of a small group: launch_superstep
      // looks at the row count.  Queueing kChunkSmall super-steps per poll there as well was measured and dropped: 5735 against 5800 windows/s,
      // same box, tools/exp_ab_libs.sh - the four groups' polls hide behind each other's kernels.)
      if ((n_run + n_trans) > 0 && G.steps < kMaxSuperSteps) { int s = launch_chunk(G); if (s) return s; any = true; }
      else G.active = false;
    }
  }
  for (Group& G : B->groups) {
    finalize_group(G);                       // windows sent straight to FINALIZE by ba_round2, or stopped by the hard limit
    LLD_HIP_TRY(hipGetLastError());
    if (G.own_stream) { LLD_HIP_TRY(hipEventRecord(G.ev[0][0], G.st)); LLD_HIP_TRY(hipStreamWaitEvent(ctx->stream, G.ev[0][0], 0)); }
  }
  LLD_HIP_TRY(hipEventRecord(t_end, ctx->stream));
  LLD_HIP_TRY(hipEventSynchronize(t_end));
  float tot = 0.f;
  (void)hipEventElapsedTime(&tot, t_begin, t_end);
  B->phase_ms[kNumPhases] = tot;
  return LLD_OK;
}

// Error contract of a solve (include/lld_amd.h): when anything inside fails - a HIP call, a launch the build cannot express - the other
// groups' streams may still hold kernels of this batch.  They are drained before the status goes back to the caller, and the batch is marked
// failed: its device state is somewhere inside an LM trial, so lld_ba_batch_solve / _download / _phase_ms refuse it (LLD_ERR_INVALID) until it
// is destroyed and created again.  The two events are owned here so that no return path leaks them.
static int ba_batch_solve_impl(lld_ba_batch* B, StopFlag abort_flag) {
  if (!B || B->failed) return LLD_ERR_INVALID;
  lld_ctx* ctx = B->ctx;
  LLD_HIP_TRY(hipSetDevice(ctx->device));
  struct Events {
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  LLD_HIP_TRY(hipEventCreate(&ev.a)); LLD_HIP_TRY(hipEventCreate(&ev.b));
  const int st = ba_batch_solve_body(B, abort_flag, ev.a, ev.b);
  if (st != LLD_OK) {
    for (auto& G : B->groups) if (G.st) (void)hipStreamSynchronize(G.st);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipGetLastError();                              // (a sticky launch error was reported through `st`)
    B->failed = true; B->records_valid = false;
  }
  return st;
}

int lld_ba_batch_solve(lld_ba_batch* B, volatile const int* abort_flag) { return ba_batch_solve_impl(B, StopFlag{abort_flag, nullptr}); }

static int ba_fetch_records(lld_ba_batch* B) {
  if (B->records_valid) return LLD_OK;
  const size_t total = B->rec_stride * (size_t)B->n_windows;
  if (B->borrowed) {                                  // pinned landing buffer kept by the context (grow-only): one DMA at link speed
    lld_ctx::BACache& c = B->ctx->ba;
    if (total > c.rec_bytes) {
      if (c.rec) LLD_HIP_TRY(hipHostFree(c.rec));
      c.rec = nullptr; c.rec_bytes = 0;
      const size_t want = total + (total >> 3) + 4096;
      LLD_HIP_TRY(hipHostMalloc(&c.rec, want, hipHostMallocDefault));
      c.rec_bytes = want;
    }
    B->h_records = static_cast<unsigned char*>(c.rec);
  } else { B->h_records_pageable.resize(total); B->h_records = B->h_records_pageable.data(); }
  LLD_HIP_TRY(hipMemcpyAsync(B->h_records, B->A.records, total, hipMemcpyDeviceToHost, B->ctx->stream));
  LLD_HIP_TRY(hipStreamSynchronize(B->ctx->stream));
  B->records_valid = true;
  return LLD_OK;
}

static void fill_stats(const lld_ba_batch* B, int wi, lld_ba_stats* s) {
  const BAWin& W = B->h_wins[wi];
  const unsigned char* rec = B->h_records + W.rec_off;
  const BARecordHeader* h = reinterpret_cast<const BARecordHeader*>(rec);
  std::memset(s, 0, sizeof *s);
  s->chi2_round1 = h->chi2_round1; s->chi2_final = h->chi2_final;
  s->lm_iterations[0] = h->lm_iterations[0]; s->lm_iterations[1] = h->lm_iterations[1];
  s->lm_trials[0] = h->lm_trials[0]; s->lm_trials[1] = h->lm_trials[1];
  s->pcg_iterations = h->pcg_iterations; s->aborted = h->aborted;
  const unsigned char* flags = rec + sizeof(BARecordHeader) + sizeof(double) * (7 * (size_t)W.n_cams + 3 * (size_t)W.n_pt + 6 * (size_t)W.n_ln);
  for (int i = 0; i < W.n_pe; i++) s->n_pt_obs_outlier += flags[i];
  for (int i = 0; i < W.n_le; i++) s->n_ln_edge_outlier += flags[W.n_pe + i];
  for (int i = 0; i < W.n_ln; i++) s->n_lines_removed += flags[W.n_pe + W.n_le + i];
}

static void ba_unpack_record(const lld_ba_batch* B, int wi, lld_ba_result* out);

int lld_ba_batch_download(lld_ba_batch* B, int wi, lld_ba_result* out) {
  if (!B || B->failed || !out || wi < 0 || wi >= B->n_windows) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(B->ctx->device));
  int st = ba_fetch_records(B); if (st) return st;
  ba_unpack_record(B, wi, out);
  return LLD_OK;
}

int lld_ba_batch_download_range(lld_ba_batch* B, int first, int count, lld_ba_result* out) {
  if (!B || B->failed || !out || first < 0 || count < 0 || first + count > B->n_windows) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(B->ctx->device));
  int st = ba_fetch_records(B); if (st) return st;
  int n_threads = count >= 8 ? (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u) : 1;
  if (g_big_solves_running[B->ctx->device & 63].load(std::memory_order_relaxed) > 0) n_threads = std::min(n_threads, staging_cap());   // see there
  if (const char* e = std::getenv("LLD_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) n_threads = std::min(n_threads, v); }
  n_threads = std::max(1, std::min(n_threads, count / 4));
  std::atomic<int> next{0};
  auto worker = [&]() { for (;;) { const int i = next.fetch_add(1); if (i >= count) break; ba_unpack_record(B, first + i, out + i); } };
  std::vector<std::thread> pool;
  try { for (int t = 1; t < n_threads; t++) pool.emplace_back(worker); } catch (...) {}
  worker();
  for (auto& t : pool) t.join();
  return LLD_OK;
}

static void ba_unpack_record(const lld_ba_batch* B, int wi, lld_ba_result* out) {
  const BAWin& W = B->h_wins[wi];
  const unsigned char* rec = B->h_records + W.rec_off;
  const double* d = reinterpret_cast<const double*>(rec + sizeof(BARecordHeader));
  if (out->cam_qt) std::memcpy(out->cam_qt, d, sizeof(double) * 7 * W.n_cams);
  d += 7 * (size_t)W.n_cams;
  if (out->pt_xyz && W.n_pt) std::memcpy(out->pt_xyz, d, sizeof(double) * 3 * W.n_pt);
  d += 3 * (size_t)W.n_pt;
  if (out->line_x0 && W.n_ln) std::memcpy(out->line_x0, d, sizeof(double) * 3 * W.n_ln);
  d += 3 * (size_t)W.n_ln;
  if (out->line_dir && W.n_ln) std::memcpy(out->line_dir, d, sizeof(double) * 3 * W.n_ln);
  d += 3 * (size_t)W.n_ln;
  const unsigned char* f = reinterpret_cast<const unsigned char*>(d);
  if (out->pt_obs_outlier && W.n_pe) std::memcpy(out->pt_obs_outlier, f, W.n_pe);
  if (out->ln_edge_outlier && W.n_le) std::memcpy(out->ln_edge_outlier, f + W.n_pe, W.n_le);
  if (out->line_removed && W.n_ln) std::memcpy(out->line_removed, f + W.n_pe + W.n_le, W.n_ln);
  fill_stats(B, wi, &out->stats);
}

int lld_ba_batch_stats(lld_ba_batch* B, lld_ba_stats* stats) {
  if (!B || B->failed || !stats) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(B->ctx->device));
  int st = ba_fetch_records(B); if (st) return st;
  for (int w = 0; w < B->n_windows; w++) fill_stats(B, w, stats + w);
  return LLD_OK;
}

int lld_ba_batch_result_records(lld_ba_batch* B, void** dev_ptr, uint64_t* stride_bytes) {
  if (!B || B->failed || !dev_ptr || !stride_bytes) return LLD_ERR_INVALID;
  *dev_ptr = B->A.records; *stride_bytes = B->rec_stride;
  return LLD_OK;
}

int lld_ba_batch_phase_ms(lld_ba_batch* B, double* ms6) {
  if (!B || B->failed || !ms6) return LLD_ERR_INVALID;
  for (int i = 0; i < LLD_BA_N_PHASES; i++) ms6[i] = B->phase_ms[i];
  return LLD_OK;
}

int lld_ba_batch_kernel_stats(lld_ba_batch* B, int kernel, int64_t* launches, double* total_ms) {
  if (!B || kernel < 0 || kernel >= kNumPhases || !launches || !total_ms) return LLD_ERR_INVALID;
  *launches = B->launches[kernel]; *total_ms = B->phase_ms[kernel];
  return LLD_OK;
}

int lld_ba_batch_set_phase_timing(lld_ba_batch* B, int on) {
  if (!B) return LLD_ERR_INVALID;
  B->phase_events = on != 0;
  return LLD_OK;
}

int lld_ba_batch_set_groups(lld_ba_batch* B, int n_groups) {
  if (!B || n_groups < 0 || n_groups > 8) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(B->ctx->device));
  LLD_HIP_TRY(hipStreamSynchronize(B->ctx->stream));
  return ba_make_groups(B, n_groups);
}

#ifdef LLD_EXPERIMENTS
// experiments build only (LLD_BA_CHOL_STAMPS=1 at create): the s_memtime stamps the LAST ba_chol_mfma_kernel launch of each window left,
// [n_windows][kCholStampWaves][kCholStampSlots] (tools/chol_stage_budget.py)
__attribute__((visibility("default"))) int lld_exp_chol_stamps(lld_ba_batch* B, long long* out) {
  if (!B || !out || !B->A.chol_stamps) return LLD_ERR_INVALID;
  LLD_HIP_TRY(hipSetDevice(B->ctx->device));
  LLD_HIP_TRY(hipMemcpy(out, B->A.chol_stamps, sizeof(long long) * (size_t)B->n_windows * kCholStampWaves * kCholStampSlots, hipMemcpyDeviceToHost));
  return LLD_OK;
}
// experiments build only, no device needed: host time of every staging stage of ONE window, as lld_ba_batch_create runs them for a batch of
// `n_windows_hint` windows (chunk size, tasks per wavefront); ms per call, the minimum over `reps` (tools/time_host_staging.py)
__attribute__((visibility("default"))) int lld_exp_stage_timing(const lld_ba_window* w, int n_windows_hint, int reps, double* ms6) {
  if (!w || !ms6 || validate_window(*w)) return LLD_ERR_INVALID;
  lld_ba_params P; lld_ba_params_default(&P);
  WinBases b{};
  const int lin_waves[2] = {4, 4};
  const int chunk_landmarks = n_windows_hint >= 64 ? 256 : (n_windows_hint >= 8 ? 64 : 32);
  const size_t NPE = (size_t)w->n_pt_obs, NLO = (size_t)w->n_ln_obs;
  std::vector<double> d7((size_t)w->n_cams * 7 + 1), d3p((size_t)w->n_points * 3 + 1), d3l((size_t)w->n_lines * 3 + 1), d3l2((size_t)w->n_lines * 3 + 1);
  std::vector<int> ps((size_t)w->n_points + 2), ls((size_t)w->n_lines + 2), pept(NPE + 1), pecs(NPE + 1), locs(NLO + 1), loln(NLO + 1);
  std::vector<float4> peobs(NPE + 1), loseg(2 * NLO + 2);
  std::vector<unsigned short> looct(NLO + 1);
  HostArrays H; H.packed = true;
  H.cam_qt0.view(d7.data(), d7.size()); H.pt0.view(d3p.data(), d3p.size()); H.ln_x0.view(d3l.data(), d3l.size()); H.ln_dir.view(d3l2.data(), d3l2.size());
  H.pt_obs_start.view(ps.data(), ps.size()); H.ln_obs_start.view(ls.data(), ls.size()); H.pe_pt.view(pept.data(), NPE);
  H.pe_obs.view(peobs.data(), NPE); H.pe_cs.view(pecs.data(), NPE); H.lo_seg.view(loseg.data(), 2 * NLO); H.lo_cs.view(locs.data(), NLO); H.lo_ln.view(loln.data(), NLO); H.lo_oct.view(looct.data(), NLO);
  for (int k = 0; k < 6; k++) ms6[k] = 1e30;
  for (int r = 0; r < std::max(1, reps); r++) {
    BAWin W; WinStage S;
    auto t0 = std::chrono::steady_clock::now();
    auto lapms = [&](int k) { const auto t1 = std::chrono::steady_clock::now(); ms6[k] = std::min(ms6[k], std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; };
    stage_tasks(*w, P, b, n_windows_hint, lin_waves, true, W, S); lapms(0);
    (void)stage_edges(*w, P, b, W, S, H); lapms(1);
    stage_chunks(*w, 3, b, chunk_landmarks, S.cs[0]); lapms(2);
    stage_chunks(*w, 4, b, chunk_landmarks, S.cs[1]); lapms(3);
    stage_csr(w->n_free_cams, S); lapms(4);
    stage_chol_plan(w->n_free_cams, 0, S); lapms(5);
  }
  return LLD_OK;
}
#endif

int lld_ba_chol_plan(int32_t n_free_cams, const uint8_t* block_nz, int32_t force, void* plan_out, uint64_t plan_bytes, uint64_t* plan_size) {
  if (plan_size) *plan_size = sizeof(CholPlan);
  if (n_free_cams < 1 || n_free_cams > 64 || !block_nz || force < 0 || force > 2) return LLD_ERR_INVALID;
  uint64_t adj[64] = {0};
  for (int a = 0; a < n_free_cams; a++)
    for (int b = 0; b < n_free_cams; b++)
      if (a == b || block_nz[a * n_free_cams + b] || block_nz[b * n_free_cams + a]) adj[a] |= 1ull << b;
  CholPlan P;
  if (6 * n_free_cams > kCholMN || !cholplan::build(n_free_cams, adj, force, P)) std::memset(&P, 0, sizeof P);
  if (plan_out && plan_bytes >= sizeof(CholPlan)) std::memcpy(plan_out, &P, sizeof P);
  else if (plan_out) return LLD_ERR_INVALID;
  return LLD_OK;
}

void lld_ba_batch_destroy(lld_ba_batch* B) {
  if (!B) return;
  (void)hipSetDevice(B->ctx->device);
  (void)hipStreamSynchronize(B->ctx->stream);
  ba_drop_groups(B);
  if (B->borrowed) B->ctx->ba.busy = false;          // slab, arenas, streams, events and poll block stay with the context for the next batch
  else {
    if (B->h_counters) (void)hipHostFree(B->h_counters);
    if (B->slab) (void)hipFree(B->slab);
  }
  delete B;
}

static int local_ba_impl(lld_ctx* ctx, const lld_ba_window* in, const lld_ba_params* params, StopFlag abort_flag, lld_ba_result* out) {
  if (!ctx || !in || !out) return LLD_ERR_INVALID;
  lld_ba_batch* B = nullptr;
  static const bool timing = exp_flag("LLD_BA_TIMING");      // experiments build: prints where a single call spends its time
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  int st = ba_batch_create_any(ctx, 1, in, params, &B); if (st) return st;
  const auto t1 = now();
  st = ba_batch_solve_impl(B, abort_flag);
  const auto t2 = now();
  if (!st) st = lld_ba_batch_download(B, 0, out);
  const auto t3 = now();
  lld_ba_batch_destroy(B);
  if (timing) std::fprintf(stderr, "[lld_local_ba] create %.3f ms, solve %.3f ms, download %.3f ms, destroy %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
  return st;
}

int lld_local_ba(lld_ctx* ctx, const lld_ba_window* in, const lld_ba_params* params, volatile const int* abort_flag, lld_ba_result* out) {
  return local_ba_impl(ctx, in, params, StopFlag{abort_flag, nullptr}, out);
}

int lld_local_ba_stopflag(lld_ctx* ctx, const lld_ba_window* in, const lld_ba_params* params, volatile const unsigned char* stop_flag, lld_ba_result* out) {
  return local_ba_impl(ctx, in, params, StopFlag{nullptr, stop_flag}, out);
}

}  // extern "C"
