// lld_ba_control.h - LM control of the batched bundle adjustment: ba_control, ba_backsub_ctl, classification between the rounds, final classification + read-back.
// Part of lld_ba_kernels.h (split by kernel family in round 6; no behaviour change): included from there INSIDE namespace lldba, after the shared types and helpers.
// Not a stand-alone header.

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

