// lld_stereo.hip — Frame::ComputeStereoMatches (src/Frame.cc:530-704) behind lld_compute_stereo_matches:
//   stage 1  row-band Hamming search            -> lld_orb_search_run (ROWS problem, lld_orb_search.hip)
//   stage 2  11x11 SAD refinement + parabola    -> stereo_refine_kernel: 16 lanes per left keypoint, lane s < 11 owns the shift
//                                                  incR = s - 5 (121 byte pairs), the group's first lane does the scalar tail
//   stage 3  median cut of the SAD distances    -> stereo_median_kernel: two-level 256-bin radix select in LDS (the distances are
//                                                  integers below 2^16), then one pass that clears the outliers
// Integer sums and single-rounding float operations (explicit _rn intrinsics, nothing contracts), so the outputs are bit-exact
// against the CPU restatement.  The two image pyramids travel in one pinned-staged copy unless the caller already has them in HBM.
#include "lld_common.h"

namespace {

constexpr int kW = 5, kL = 5;                       // w and L of Frame.cc:627,633
constexpr int kMaxLevels = LLD_ORB_MAX_LEVELS;

struct RefineArgs {
  int n_left;
  const float* left_xy; const int32_t* left_octave; const float* right_xy; const int32_t* best_r;
  const uint8_t* left_img[kMaxLevels]; const uint8_t* right_img[kMaxLevels];
  int cols[kMaxLevels], rows[kMaxLevels], lstep[kMaxLevels], rstep[kMaxLevels];
  float scale[kMaxLevels], inv_scale[kMaxLevels];
  float min_d, max_d, mbf;
  float* u_right; float* depth; int32_t* sad;
};

__global__ __launch_bounds__(256) void stereo_refine_kernel(RefineArgs A) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int iL = gid >> 4, s = gid & 15;
  if (iL >= A.n_left) return;                       // whole 16-lane groups leave together
  const int bR = A.best_r[iL];
  bool ok = bR >= 0;
  int o = 0, x0 = 0, y0 = 0, xr = 0, cols = 0;
  float uL = 0.f, suR0 = 0.f;
  if (ok) {
    o = A.left_octave[iL];
    uL = A.left_xy[2 * iL];
    const float vL = A.left_xy[2 * iL + 1], uR0 = A.right_xy[2 * bR];
    const float sf = A.inv_scale[o];
    const float suL = roundf(__fmul_rn(uL, sf)), svL = roundf(__fmul_rn(vL, sf));
    suR0 = roundf(__fmul_rn(uR0, sf));
    cols = A.cols[o];
    const float iniu = __fsub_rn(__fadd_rn(suR0, (float)kL), (float)kW);                          // scaleduR0+L-w
    const float endu = __fadd_rn(__fadd_rn(__fadd_rn(suR0, (float)kL), (float)kW), 1.0f);         // scaleduR0+L+w+1
    if (iniu < 0.f || endu >= (float)cols) ok = false;                                            // Frame.cc:640-641
    x0 = (int)suL; y0 = (int)svL; xr = (int)suR0;
    // the reference slices these ranges unchecked; a patch that leaves the image gets no match here
    if (x0 - kW < 0 || x0 + kW >= cols || y0 - kW < 0 || y0 + kW >= A.rows[o] || xr - kL - kW < 0 || xr + kL + kW >= cols) ok = false;
  }
  int dist = 0x7fffffff;
  if (ok && s <= 2 * kL) {
    const int inc = s - kL;
    const uint8_t* Lp = A.left_img[o] + (size_t)(y0 - kW) * A.lstep[o] + (x0 - kW);
    const uint8_t* Rp = A.right_img[o] + (size_t)(y0 - kW) * A.rstep[o] + (xr + inc - kW);
    const int lc = Lp[kW * A.lstep[o] + kW], rc = Rp[kW * A.rstep[o] + kW];
    int acc = 0;
    for (int r = 0; r <= 2 * kW; r++) {
      const uint8_t* lr = Lp + (size_t)r * A.lstep[o];
      const uint8_t* rr = Rp + (size_t)r * A.rstep[o];
#pragma unroll
      for (int c = 0; c <= 2 * kW; c++) acc += abs(((int)lr[c] - lc) - ((int)rr[c] - rc));        // cv::norm(IL,IR,NORM_L1), exact
    }
    dist = acc;
  }
  int d[2 * kL + 1];
#pragma unroll
  for (int i = 0; i <= 2 * kL; i++) d[i] = __shfl(dist, i, 16);
  if (s != 0) return;
  float ur = -1.0f, dep = -1.0f; int sad = -1;
  if (ok) {
    int best = 0x7fffffff, binc = 0;                                                              // int bestDist = INT_MAX
#pragma unroll
    for (int i = 0; i <= 2 * kL; i++) if (d[i] < best) { best = d[i]; binc = i - kL; }
    if (binc != -kL && binc != kL) {
      float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
      for (int i = 1; i < 2 * kL; i++) if (i - kL == binc) { d1 = (float)d[i - 1]; d2 = (float)d[i]; d3 = (float)d[i + 1]; }
      // deltaR = (dist1-dist3)/(2.0f*(dist1+dist3-2.0f*dist2))
      const float delta = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
      if (!(delta < -1.f || delta > 1.f)) {
        float bestuR = __fmul_rn(A.scale[o], __fadd_rn(__fadd_rn(suR0, (float)binc), delta));
        float disparity = __fsub_rn(uL, bestuR);
        if (disparity >= A.min_d && disparity < A.max_d) {
          if (disparity <= 0.f) { disparity = 0.01f; bestuR = (float)__dsub_rn((double)uL, 0.01); }
          dep = __fdiv_rn(A.mbf, disparity);
          ur = bestuR; sad = best;
        }
      }
    }
  }
  A.u_right[iL] = ur; A.depth[iL] = dep; A.sad[iL] = sad;
}

// sort(vDistIdx); median = vDistIdx[size/2].first; thDist = 1.5f*1.4f*median; every entry with first >= thDist is cleared
// (Frame.cc:690-703).  One workgroup; distances < 2^16 (121 * 510).
__global__ __launch_bounds__(1024) void stereo_median_kernel(int n, float* __restrict__ u_right, float* __restrict__ depth, const int32_t* __restrict__ sad,
                                                            int32_t* __restrict__ summary) {
  __shared__ int hist[256];
  __shared__ int sel[4];       // 0: count, 1: high byte of the median, 2: rank inside that bin, 3: median
  const int tid = threadIdx.x;
  if (tid < 256) hist[tid] = 0;
  if (tid < 4) sel[tid] = 0;
  __syncthreads();
  int cnt = 0;
  for (int i = tid; i < n; i += 1024) { const int v = sad[i]; if (v >= 0) { atomicAdd(&hist[(v >> 8) & 255], 1); cnt++; } }
  if (cnt) atomicAdd(&sel[0], cnt);
  __syncthreads();
  const int total = sel[0];
  if (total == 0) { if (tid == 0) { summary[0] = 0; summary[1] = -1; } return; }   // the reference reads vDistIdx[0] of an empty vector here
  if (tid == 0) {
    int rank = total / 2, b = 0;
    while (rank >= hist[b]) { rank -= hist[b]; b++; }
    sel[1] = b; sel[2] = rank;
  }
  __syncthreads();
  const int hb = sel[1];
  __syncthreads();
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) { const int v = sad[i]; if (v >= 0 && ((v >> 8) & 255) == hb) atomicAdd(&hist[v & 255], 1); }
  __syncthreads();
  if (tid == 0) {
    int rank = sel[2], b = 0;
    while (rank >= hist[b]) { rank -= hist[b]; b++; }
    sel[3] = (hb << 8) | b;
  }
  __syncthreads();
  const float median = (float)sel[3];
  const float th = __fmul_rn(1.5f * 1.4f, median);
  int kept = 0;
  for (int i = tid; i < n; i += 1024) {
    const int v = sad[i];
    if (v < 0) continue;
    if ((float)v < th) kept++; else { u_right[i] = -1.0f; depth[i] = -1.0f; }
  }
  __syncthreads();
  if (tid == 0) sel[0] = 0;
  __syncthreads();
  if (kept) atomicAdd(&sel[0], kept);
  __syncthreads();
  if (tid == 0) { summary[0] = sel[0]; summary[1] = sel[3]; }
}

inline size_t al64(size_t b) { return (b + 63) & ~size_t(63); }

}  // namespace

extern "C" int lld_compute_stereo_matches(lld_ctx* ctx, const lld_keypoints* left, const lld_keypoints* right, const lld_stereo_pyramids* pyr,
                                          float mb, float mbf, lld_stereo_result* out) {
  if (!ctx || !left || !right || !pyr || !out) return LLD_ERR_INVALID;
  const int nl = left->n, nr = right->n, nlv = pyr->n_levels;
  if (nl < 0 || nr < 0 || nlv <= 0 || nlv > kMaxLevels) return LLD_ERR_INVALID;
  if (nl > LLD_ORB_MAX_KEYPOINTS || nr > LLD_ORB_MAX_KEYPOINTS) return LLD_ERR_UNSUPPORTED;
  if (!out->u_right || !out->depth) return LLD_ERR_INVALID;
  if ((nl > 0 && (!left->xy || !left->octave || !left->desc)) || (nr > 0 && (!right->xy || !right->octave || !right->desc))) return LLD_ERR_INVALID;
  if (!pyr->left || !pyr->right || !pyr->cols || !pyr->rows || !pyr->left_step || !pyr->right_step || !pyr->scale_factors || !pyr->inv_scale_factors)
    return LLD_ERR_INVALID;
  for (int l = 0; l < nlv; l++)
    if (!pyr->left[l] || !pyr->right[l] || pyr->cols[l] <= 0 || pyr->rows[l] <= 0 || pyr->left_step[l] < pyr->cols[l] || pyr->right_step[l] < pyr->cols[l])
      return LLD_ERR_INVALID;
  for (int i = 0; i < nl; i++) if (left->octave[i] < 0 || left->octave[i] >= nlv) return LLD_ERR_INVALID;
  for (int i = 0; i < nr; i++) if (right->octave[i] < 0 || right->octave[i] >= nlv) return LLD_ERR_INVALID;
  out->n_matches = 0;
  if (nl == 0) return LLD_OK;
  LLD_HIP_TRY(hipSetDevice(ctx->device));

  // ---- stage 1: Frame.cc:536-613 as the ROWS problem (right keypoints searched, one query per left keypoint)
  const float minD = 0.0f, maxD = mbf / mb;                                                      // minZ = mb; maxD = mbf/minZ (:559-561)
  std::vector<int32_t> lmin((size_t)nl), lmax((size_t)nl), match((size_t)nl), bd((size_t)nl), sd((size_t)nl);
  std::vector<uint8_t> removed((size_t)nl);
  for (int i = 0; i < nl; i++) { lmin[i] = left->octave[i] - 1; lmax[i] = left->octave[i] + 1; }
  lld_orb_search S; std::memset(&S, 0, sizeof(S));
  S.nt = nr; S.t_desc = right->desc; S.t_xy = right->xy; S.t_octave = right->octave;
  S.nq = nl; S.q_desc = left->desc; S.q_uv = left->xy; S.q_level_min = lmin.data(); S.q_level_max = lmax.data();
  S.n_levels = nlv; S.level_scale = pyr->scale_factors;
  S.disp_min = minD; S.disp_max = maxD;
  S.candidates = LLD_ORB_CAND_ROWS; S.gates = LLD_ORB_GATE_LEVEL; S.accept_max = (100 + 50) / 2 - 1;   // bestDist < thOrbDist
  lld_orb_search_result R; std::memset(&R, 0, sizeof(R));
  R.match = match.data(); R.best_dist = bd.data(); R.second_dist = sd.data(); R.removed = removed.data();
  int st = lld_orb_search_run(ctx, &S, &R); if (st) return st;

  // ---- stages 2 + 3: one input region [best_r | left xy, octave | right xy | images], one output region [u_right | depth | sad | summary]
  size_t in = 0, outb = 0;
  auto add_in = [&](size_t b) { const size_t o = in; in += al64(b); return o; };
  auto add_out = [&](size_t b) { const size_t o = outb; outb += al64(b); return o; };
  const size_t o_br = add_in((size_t)nl * 4), o_lxy = add_in((size_t)nl * 8), o_loct = add_in((size_t)nl * 4), o_rxy = add_in((size_t)nr * 8 + 8);
  size_t o_limg[kMaxLevels] = {}, o_rimg[kMaxLevels] = {};
  if (!pyr->on_device)
    for (int l = 0; l < nlv; l++) { o_limg[l] = add_in((size_t)pyr->cols[l] * pyr->rows[l]); o_rimg[l] = add_in((size_t)pyr->cols[l] * pyr->rows[l]); }
  const size_t r_ur = add_out((size_t)nl * 4), r_dep = add_out((size_t)nl * 4), r_sad = add_out((size_t)nl * 4), r_sum = add_out(16);
  void* hb; st = lld_ctx_pinned(ctx, in + outb, &hb); if (st) return st;
  void* db; st = lld_ctx_scratch(ctx, in + outb + 256, &db); if (st) return st;
  char* h = (char*)hb; char* d = (char*)db; char* h_out = h + in; char* d_out = d + in;
  std::memcpy(h + o_br, match.data(), (size_t)nl * 4);
  std::memcpy(h + o_lxy, left->xy, (size_t)nl * 8); std::memcpy(h + o_loct, left->octave, (size_t)nl * 4);
  if (nr) std::memcpy(h + o_rxy, right->xy, (size_t)nr * 8);
  RefineArgs A; std::memset(&A, 0, sizeof(A));
  for (int l = 0; l < nlv; l++) {
    A.cols[l] = pyr->cols[l]; A.rows[l] = pyr->rows[l]; A.scale[l] = pyr->scale_factors[l]; A.inv_scale[l] = pyr->inv_scale_factors[l];
    if (pyr->on_device) { A.left_img[l] = pyr->left[l]; A.right_img[l] = pyr->right[l]; A.lstep[l] = pyr->left_step[l]; A.rstep[l] = pyr->right_step[l]; }
    else {
      // rows are packed tightly in the staging buffer whatever the caller's step
      for (int r = 0; r < pyr->rows[l]; r++) {
        std::memcpy(h + o_limg[l] + (size_t)r * pyr->cols[l], pyr->left[l] + (size_t)r * pyr->left_step[l], (size_t)pyr->cols[l]);
        std::memcpy(h + o_rimg[l] + (size_t)r * pyr->cols[l], pyr->right[l] + (size_t)r * pyr->right_step[l], (size_t)pyr->cols[l]);
      }
      A.left_img[l] = reinterpret_cast<const uint8_t*>(d + o_limg[l]); A.right_img[l] = reinterpret_cast<const uint8_t*>(d + o_rimg[l]);
      A.lstep[l] = pyr->cols[l]; A.rstep[l] = pyr->cols[l];
    }
  }
  A.n_left = nl;
  A.left_xy = reinterpret_cast<const float*>(d + o_lxy); A.left_octave = reinterpret_cast<const int32_t*>(d + o_loct);
  A.right_xy = reinterpret_cast<const float*>(d + o_rxy); A.best_r = reinterpret_cast<const int32_t*>(d + o_br);
  A.min_d = minD; A.max_d = maxD; A.mbf = mbf;
  A.u_right = reinterpret_cast<float*>(d_out + r_ur); A.depth = reinterpret_cast<float*>(d_out + r_dep); A.sad = reinterpret_cast<int32_t*>(d_out + r_sad);
  hipStream_t sm = ctx->stream;
  LLD_HIP_TRY(hipMemcpyAsync(d, h, in, hipMemcpyHostToDevice, sm));
  hipLaunchKernelGGL(stereo_refine_kernel, dim3((nl * 16 + 255) / 256), dim3(256), 0, sm, A);
  hipLaunchKernelGGL(stereo_median_kernel, dim3(1), dim3(1024), 0, sm, nl, A.u_right, A.depth, A.sad, reinterpret_cast<int32_t*>(d_out + r_sum));
  LLD_HIP_TRY(hipGetLastError());
  LLD_HIP_TRY(hipMemcpyAsync(h_out, d_out, outb, hipMemcpyDeviceToHost, sm));
  LLD_HIP_TRY(hipStreamSynchronize(sm));
  std::memcpy(out->u_right, h_out + r_ur, (size_t)nl * 4); std::memcpy(out->depth, h_out + r_dep, (size_t)nl * 4);
  if (out->best_r) std::memcpy(out->best_r, match.data(), (size_t)nl * 4);
  if (out->sad) std::memcpy(out->sad, h_out + r_sad, (size_t)nl * 4);
  out->n_matches = reinterpret_cast<const int32_t*>(h_out + r_sum)[0];
  return LLD_OK;
}
