// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
// (lld_slam_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the reference cannot be built here (no OpenCV / DBoW2) and has no tests or golden
// vectors for its matchers; these restatements are pinned by the known-answer tests under tests/.
//
// lldo_orbsearch.cpp — literal, sequential restatements of the reference's guided ORB searches on flat
// arrays.  Each routine keeps the reference's loop structure, its order of candidate visits and its
// float arithmetic (compiled with -ffp-contract=off); what the reference computes per query BEFORE
// its inner loop from cv::Mat poses (projection, predicted level, view cosine) comes in as input.
//   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea     src/Frame.cc:294-313,446-456,391-444
//   KeyFrame::GetFeaturesInArea                                     src/KeyFrame.cc:592-631
//   ORBmatcher::SearchByProjection (local map)                      src/ORBmatcher.cc:45-129
//   ORBmatcher::SearchByProjection (frame to frame)                 src/ORBmatcher.cc:1328-1470
//   ORBmatcher::SearchByProjection (relocalisation)                 src/ORBmatcher.cc:1472-1599
//   ORBmatcher::SearchByProjection (KF, Scw)                        src/ORBmatcher.cc:290-403
//   ORBmatcher::SearchByBoW (KF, Frame) / (KF, KF)                  src/ORBmatcher.cc:159-288, 522-655
//   ORBmatcher::SearchForTriangulation                              src/ORBmatcher.cc:657-823
//   ORBmatcher::Fuse / SearchBySim3 inner searches                  src/ORBmatcher.cc:825-1100, 1102-1326
//   ORBmatcher::ComputeThreeMaxima                                  src/ORBmatcher.cc:1601-1642
//   Frame::ComputeStereoMatches (Hamming search)                    src/Frame.cc:530-613
//   Frame::isInFrustum, MapPoint::PredictScale                      src/Frame.cc:333-389, src/MapPoint.cc:402-417
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/lld_amd.h"

extern "C" int lldo_descriptor_distance(const uint32_t* a, const uint32_t* b);

namespace {

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // ORBmatcher.cc:37-39
constexpr int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;      // Frame.h:43-44

struct Grid {
  std::vector<std::vector<int>> cell;   // [ix * ROWS + iy] in insertion (= keypoint index) order
};

}  // namespace

extern "C" {

// One frame's keypoints and constants (Frame / KeyFrame members of the same names).
struct lldo_frame {
  int32_t n;
  const uint32_t* desc;    // mDescriptors rows
  const float* xy;         // mvKeysUn[i].pt
  const int32_t* octave;   // mvKeysUn[i].octave
  const float* uright;     // mvuRight
  const float* angle;      // mvKeysUn[i].angle
  float min_x, min_y, width_inv, height_inv;   // mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
  int32_t n_levels;
  const float* scale;      // mvScaleFactors
  const float* sigma2;     // mvLevelSigma2
  const float* inv_sigma2; // mvInvLevelSigma2
};

}  // extern "C"

namespace {

// Frame::AssignFeaturesToGrid + PosInGrid
Grid build_grid(const lldo_frame& F) {
  Grid g; g.cell.resize((size_t)FRAME_GRID_COLS * FRAME_GRID_ROWS);
  for (int i = 0; i < F.n; i++) {
    const int posX = (int)std::round((F.xy[2 * i] - F.min_x) * F.width_inv);
    const int posY = (int)std::round((F.xy[2 * i + 1] - F.min_y) * F.height_inv);
    if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
    g.cell[(size_t)posX * FRAME_GRID_ROWS + posY].push_back(i);
  }
  return g;
}

// Frame::GetFeaturesInArea; KeyFrame::GetFeaturesInArea is the same with minLevel=-1,maxLevel=-1
std::vector<int> features_in_area(const lldo_frame& F, const Grid& g, float x, float y, float r, int minLevel = -1, int maxLevel = -1) {
  std::vector<int> v;
  const int nMinCellX = std::max(0, (int)std::floor((x - F.min_x - r) * F.width_inv));
  if (nMinCellX >= FRAME_GRID_COLS) return v;
  const int nMaxCellX = std::min(FRAME_GRID_COLS - 1, (int)std::ceil((x - F.min_x + r) * F.width_inv));
  if (nMaxCellX < 0) return v;
  const int nMinCellY = std::max(0, (int)std::floor((y - F.min_y - r) * F.height_inv));
  if (nMinCellY >= FRAME_GRID_ROWS) return v;
  const int nMaxCellY = std::min(FRAME_GRID_ROWS - 1, (int)std::ceil((y - F.min_y + r) * F.height_inv));
  if (nMaxCellY < 0) return v;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const std::vector<int>& vCell = g.cell[(size_t)ix * FRAME_GRID_ROWS + iy];
      for (size_t j = 0; j < vCell.size(); j++) {
        const int k = vCell[j];
        if (bCheckLevels) {
          if (F.octave[k] < minLevel) continue;
          if (maxLevel >= 0 && F.octave[k] > maxLevel) continue;
        }
        const float distx = F.xy[2 * k] - x, disty = F.xy[2 * k + 1] - y;
        if (std::fabs(distx) < r && std::fabs(disty) < r) v.push_back(k);
      }
    }
  return v;
}

// ORBmatcher::ComputeThreeMaxima
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

inline int rot_bin(float a1, float a2) {
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

// the tail shared by every routine with mbCheckOrientation: NULL the slots of all but the three largest bins
int apply_rotation(std::vector<int>* rotHist, int32_t* slots, int nmatches) {
  int ind1 = -1, ind2 = -1, ind3 = -1;
  three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
  for (int i = 0; i < HISTO_LENGTH; i++) {
    if (i == ind1 || i == ind2 || i == ind3) continue;
    for (size_t j = 0; j < rotHist[i].size(); j++) { slots[rotHist[i][j]] = -1; nmatches--; }
  }
  return nmatches;
}

}  // namespace

extern "C" {

// test hook: Frame::GetFeaturesInArea on its own; returns the count, writes the indices in visit order
int lldo_features_in_area(const lldo_frame* F, float x, float y, float r, int minLevel, int maxLevel, int32_t* out) {
  const Grid g = build_grid(*F);
  const std::vector<int> v = features_in_area(*F, g, x, y, r, minLevel, maxLevel);
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  return (int)v.size();
}

// test hook: ComputeThreeMaxima on bin counts
void lldo_three_maxima(const int32_t* counts, int32_t* ind) {
  std::vector<int> h[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) h[i].resize(counts[i]);
  ind[0] = ind[1] = ind[2] = -1;
  three_maxima(h, HISTO_LENGTH, ind[0], ind[1], ind[2]);
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)   ORBmatcher.cc:45-129
//   in_view[i]   = pMP->mbTrackInView && !pMP->isBad();  proj = (mTrackProjX, mTrackProjY), proj_xr = mTrackProjXR
//   mp_obs[i]    = pMP->Observations()>0;  f_slot[k] in: >=0 iff F.mvpMapPoints[k] && Observations()>0 (value ignored), out: index of
//                  the MapPoint written to F.mvpMapPoints[k] (entries that were occupied on entry keep their input value)
int lldo_search_by_projection_map(const lldo_frame* F, int n_mp, const uint32_t* mp_desc, const uint8_t* in_view, const float* proj,
                                  const float* proj_xr, const int32_t* pred_level, const float* view_cos, const uint8_t* mp_obs,
                                  float th, float mfNNratio, int32_t* f_slot, uint8_t* f_slot_obs) {
  const Grid g = build_grid(*F);
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < n_mp; iMP++) {
    if (!in_view[iMP]) continue;
    const int nPredictedLevel = pred_level[iMP];
    float r = (view_cos[iMP] > 0.998) ? 2.5f : 4.0f;        // RadiusByViewingCos
    if (bFactor) r *= th;
    const std::vector<int> vIndices = features_in_area(*F, g, proj[2 * iMP], proj[2 * iMP + 1], r * F->scale[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
    if (vIndices.empty()) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (size_t c = 0; c < vIndices.size(); c++) {
      const int idx = vIndices[c];
      if (f_slot[idx] >= 0 && f_slot_obs[idx]) continue;
      if (F->uright[idx] > 0) {
        const float er = std::fabs(proj_xr[iMP] - F->uright[idx]);
        if (er > r * F->scale[nPredictedLevel]) continue;
      }
      const int dist = lldo_descriptor_distance(mp_desc + 8 * iMP, F->desc + 8 * idx);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F->octave[idx]; bestIdx = idx; }
      else if (dist < bestDist2) { bestLevel2 = F->octave[idx]; bestDist2 = dist; }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
      f_slot[bestIdx] = iMP; f_slot_obs[bestIdx] = mp_obs[iMP];
      nmatches++;
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono)   ORBmatcher.cc:1328-1470
//   valid[i] = LastFrame.mvpMapPoints[i] && !mvbOutlier[i] && invzc>=0 && (u,v) inside the image bounds; uv, ur = u - mbf*invzc
//   and last_octave = LastFrame.mvKeys[i].octave as the reference computes them; direction: 1 forward, -1 backward, 0 neither
int lldo_search_by_projection_frame(const lldo_frame* Cur, int n_last, const uint32_t* last_desc, const uint8_t* valid, const float* uv,
                                    const float* ur, const int32_t* last_octave, const float* last_angle, const uint8_t* mp_obs,
                                    int direction, float th, int check_orientation, int32_t* cur_slot, uint8_t* cur_slot_obs) {
  const Grid g = build_grid(*Cur);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < n_last; i++) {
    if (!valid[i]) continue;
    const float u = uv[2 * i], v = uv[2 * i + 1];
    const int nLastOctave = last_octave[i];
    const float radius = th * Cur->scale[nLastOctave];
    std::vector<int> vIndices2;
    if (direction > 0) vIndices2 = features_in_area(*Cur, g, u, v, radius, nLastOctave);
    else if (direction < 0) vIndices2 = features_in_area(*Cur, g, u, v, radius, 0, nLastOctave);
    else vIndices2 = features_in_area(*Cur, g, u, v, radius, nLastOctave - 1, nLastOctave + 1);
    if (vIndices2.empty()) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (size_t c = 0; c < vIndices2.size(); c++) {
      const int i2 = vIndices2[c];
      if (cur_slot[i2] >= 0 && cur_slot_obs[i2]) continue;
      if (Cur->uright[i2] > 0) {
        const float er = std::fabs(ur[i] - Cur->uright[i2]);
        if (er > radius) continue;
      }
      const int dist = lldo_descriptor_distance(last_desc + 8 * i, Cur->desc + 8 * i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      cur_slot[bestIdx2] = i; cur_slot_obs[bestIdx2] = mp_obs[i];
      nmatches++;
      if (check_orientation) rotHist[rot_bin(last_angle[i], Cur->angle[bestIdx2])].push_back(bestIdx2);
    }
  }
  if (check_orientation) nmatches = apply_rotation(rotHist, cur_slot, nmatches);
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& Current, KeyFrame*, sAlreadyFound, th, ORBdist)   ORBmatcher.cc:1472-1599
//   valid[i] = pMP && !isBad && !sAlreadyFound.count(pMP) && in image && dist3D in range;  cur_slot[k] >= 0 iff CurrentFrame.mvpMapPoints[k]
int lldo_search_by_projection_reloc(const lldo_frame* Cur, int n, const uint32_t* desc, const uint8_t* valid, const float* uv,
                                    const int32_t* pred_level, const float* kf_angle, float th, int ORBdist, int check_orientation,
                                    int32_t* cur_slot) {
  const Grid g = build_grid(*Cur);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < n; i++) {
    if (!valid[i]) continue;
    const int nPredictedLevel = pred_level[i];
    const float radius = th * Cur->scale[nPredictedLevel];
    const std::vector<int> vIndices2 = features_in_area(*Cur, g, uv[2 * i], uv[2 * i + 1], radius, nPredictedLevel - 1, nPredictedLevel + 1);
    if (vIndices2.empty()) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (size_t c = 0; c < vIndices2.size(); c++) {
      const int i2 = vIndices2[c];
      if (cur_slot[i2] >= 0) continue;
      const int dist = lldo_descriptor_distance(desc + 8 * i, Cur->desc + 8 * i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= ORBdist) {
      cur_slot[bestIdx2] = i;
      nmatches++;
      if (check_orientation) rotHist[rot_bin(kf_angle[i], Cur->angle[bestIdx2])].push_back(bestIdx2);
    }
  }
  if (check_orientation) nmatches = apply_rotation(rotHist, cur_slot, nmatches);
  return nmatches;
}

// ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)   ORBmatcher.cc:290-403
//   valid[i] = every `continue` before the search (:316-361) not taken;  matched[k] >= 0 iff vpMatched[k]
int lldo_search_by_projection_kf(const lldo_frame* KF, int n, const uint32_t* desc, const uint8_t* valid, const float* uv,
                                 const int32_t* pred_level, int th, int32_t* matched) {
  const Grid g = build_grid(*KF);
  int nmatches = 0;
  for (int iMP = 0; iMP < n; iMP++) {
    if (!valid[iMP]) continue;
    const int nPredictedLevel = pred_level[iMP];
    const float radius = th * KF->scale[nPredictedLevel];
    const std::vector<int> vIndices = features_in_area(*KF, g, uv[2 * iMP], uv[2 * iMP + 1], radius);
    if (vIndices.empty()) continue;
    int bestDist = 256, bestIdx = -1;
    for (size_t c = 0; c < vIndices.size(); c++) {
      const int idx = vIndices[c];
      if (matched[idx] >= 0) continue;
      const int kpLevel = KF->octave[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist = lldo_descriptor_distance(desc + 8 * iMP, KF->desc + 8 * idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_LOW) { matched[bestIdx] = iMP; nmatches++; }
  }
  return nmatches;
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)   ORBmatcher.cc:405-520
//   prev_matched [n1][2] = vbPrevMatched (in: where each F1 keypoint was last matched; out: updated with the F2 position of every match)
//   matches12 [n1] = vnMatches12.  INT_MAX of the reference is 256 here (no distance exceeds it).
int lldo_search_for_initialization(const lldo_frame* F1, const lldo_frame* F2, float* prev_matched, int windowSize, float nnratio,
                                   int check_orientation, int32_t* matches12) {
  const Grid g = build_grid(*F2);
  int nmatches = 0;
  for (int i = 0; i < F1->n; i++) matches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> vMatchedDistance((size_t)F2->n, 0x7fffffff), vnMatches21((size_t)F2->n, -1);
  for (int i1 = 0; i1 < F1->n; i1++) {
    const int level1 = F1->octave[i1];
    if (level1 > 0) continue;
    const std::vector<int> vIndices2 = features_in_area(*F2, g, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
    for (size_t c = 0; c < vIndices2.size(); c++) {
      const int i2 = vIndices2[c];
      const int dist = lldo_descriptor_distance(F1->desc + 8 * i1, F2->desc + 8 * i2);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        matches12[i1] = bestIdx2; vnMatches21[bestIdx2] = i1; vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (check_orientation) rotHist[rot_bin(F1->angle[i1], F2->angle[bestIdx2])].push_back(i1);
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0; j < rotHist[i].size(); j++) {
        const int idx1 = rotHist[i][j];
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (int i1 = 0; i1 < F1->n; i1++)
    if (matches12[i1] >= 0) { prev_matched[2 * i1] = F2->xy[2 * matches12[i1]]; prev_matched[2 * i1 + 1] = F2->xy[2 * matches12[i1] + 1]; }
  return nmatches;
}

// Inner search of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th)   ORBmatcher.cc:825-958 (the Scw variant :960-1100 has the
// same loop without the stereo branch inputs).  best_idx[i] = bestIdx when bestDist<=TH_LOW else -1; the replace/add
// bookkeeping on the map (:936-954) is the caller's.  Returns nFused.
int lldo_fuse_search(const lldo_frame* KF, int n, const uint32_t* desc, const uint8_t* valid, const float* uv, const float* ur,
                     const int32_t* pred_level, float th, int32_t* best_idx) {
  const Grid g = build_grid(*KF);
  int nFused = 0;
  for (int i = 0; i < n; i++) {
    best_idx[i] = -1;
    if (!valid[i]) continue;
    const float u = uv[2 * i], v = uv[2 * i + 1];
    const int nPredictedLevel = pred_level[i];
    const float radius = th * KF->scale[nPredictedLevel];
    const std::vector<int> vIndices = features_in_area(*KF, g, u, v, radius);
    if (vIndices.empty()) continue;
    int bestDist = 256, bestIdx = -1;
    for (size_t c = 0; c < vIndices.size(); c++) {
      const int idx = vIndices[c];
      const int kpLevel = KF->octave[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const float kpx = KF->xy[2 * idx], kpy = KF->xy[2 * idx + 1];
      if (KF->uright[idx] >= 0) {
        const float kpr = KF->uright[idx];
        const float ex = u - kpx, ey = v - kpy, er = ur[i] - kpr;
        const float e2 = ex * ex + ey * ey + er * er;
        if (e2 * KF->inv_sigma2[kpLevel] > 7.8) continue;
      } else {
        const float ex = u - kpx, ey = v - kpy;
        const float e2 = ex * ex + ey * ey;
        if (e2 * KF->inv_sigma2[kpLevel] > 5.99) continue;
      }
      const int dist = lldo_descriptor_distance(desc + 8 * i, KF->desc + 8 * idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; nFused++; }
  }
  return nFused;
}

// One direction of ORBmatcher::SearchBySim3 (ORBmatcher.cc:1147-1224 / :1227-1304): vnMatch[i] = bestIdx or -1
void lldo_search_sim3_direction(const lldo_frame* KF2, int n, const uint32_t* desc, const uint8_t* valid, const float* uv,
                                const int32_t* pred_level, float th, int32_t* vnMatch) {
  const Grid g = build_grid(*KF2);
  for (int i1 = 0; i1 < n; i1++) {
    vnMatch[i1] = -1;
    if (!valid[i1]) continue;
    const int nPredictedLevel = pred_level[i1];
    const float radius = th * KF2->scale[nPredictedLevel];
    const std::vector<int> vIndices = features_in_area(*KF2, g, uv[2 * i1], uv[2 * i1 + 1], radius);
    if (vIndices.empty()) continue;
    int bestDist = INT_MAX, bestIdx = -1;
    for (size_t c = 0; c < vIndices.size(); c++) {
      const int idx = vIndices[c];
      if (KF2->octave[idx] < nPredictedLevel - 1 || KF2->octave[idx] > nPredictedLevel) continue;
      const int dist = lldo_descriptor_distance(desc + 8 * i1, KF2->desc + 8 * idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_HIGH) vnMatch[i1] = bestIdx;
  }
}

// Inner search of ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:1050-1093): no chi2 gate, no
// occupancy; best[i] = bestIdx when bestDist<=TH_LOW, else -1; returns nFused
int lldo_fuse_search_sim3(const lldo_frame* KF, int n, const uint32_t* desc, const uint8_t* valid, const float* uv,
                          const int32_t* pred_level, float th, int32_t* best) {
  const Grid g = build_grid(*KF);
  int nFused = 0;
  for (int iMP = 0; iMP < n; iMP++) {
    best[iMP] = -1;
    if (!valid[iMP]) continue;
    const int nPredictedLevel = pred_level[iMP];
    const float radius = th * KF->scale[nPredictedLevel];
    const std::vector<int> vIndices = features_in_area(*KF, g, uv[2 * iMP], uv[2 * iMP + 1], radius);
    if (vIndices.empty()) continue;
    int bestDist = INT_MAX, bestIdx = -1;
    for (size_t c = 0; c < vIndices.size(); c++) {
      const int idx = vIndices[c];
      const int kpLevel = KF->octave[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist = lldo_descriptor_distance(desc + 8 * iMP, KF->desc + 8 * idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_LOW) { best[iMP] = bestIdx; nFused++; }
  }
  return nFused;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)   ORBmatcher.cc:159-288
//   the two FeatureVectors arrive as the list of COMMON nodes in ascending node id (what the merge loop :183-251 visits):
//   node n holds KF indices kf_idx[kf_start[n]..kf_start[n+1]) and F indices f_idx[f_start[n]..).  kf_valid[k] = pMP && !isBad.
//   f_match[k] out: KF keypoint index whose MapPoint is written to vpMapPointMatches[k], or -1.
int lldo_search_by_bow_frame(const lldo_frame* KF, const lldo_frame* F, int n_nodes, const int32_t* kf_start, const int32_t* kf_idx,
                             const int32_t* f_start, const int32_t* f_idx, const uint8_t* kf_valid, float mfNNratio,
                             int check_orientation, int32_t* f_match) {
  for (int k = 0; k < F->n; k++) f_match[k] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int node = 0; node < n_nodes; node++) {
    for (int a = kf_start[node]; a < kf_start[node + 1]; a++) {
      const int realIdxKF = kf_idx[a];
      if (!kf_valid[realIdxKF]) continue;
      int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
      for (int b = f_start[node]; b < f_start[node + 1]; b++) {
        const int realIdxF = f_idx[b];
        if (f_match[realIdxF] >= 0) continue;
        const int dist = lldo_descriptor_distance(KF->desc + 8 * realIdxKF, F->desc + 8 * realIdxF);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
        else if (dist < bestDist2) { bestDist2 = dist; }
      }
      if (bestDist1 <= TH_LOW) {
        if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
          f_match[bestIdxF] = realIdxKF;
          if (check_orientation) rotHist[rot_bin(KF->angle[realIdxKF], F->angle[bestIdxF])].push_back(bestIdxF);
          nmatches++;
        }
      }
    }
  }
  if (check_orientation) nmatches = apply_rotation(rotHist, f_match, nmatches);
  return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)   ORBmatcher.cc:522-655
//   valid1[k] = pMP1 && !isBad;  valid2[k] = pMP2 && !isBad;  matches12[idx1] = idx2 or -1
int lldo_search_by_bow_kf(const lldo_frame* KF1, const lldo_frame* KF2, int n_nodes, const int32_t* start1, const int32_t* idx1v,
                          const int32_t* start2, const int32_t* idx2v, const uint8_t* valid1, const uint8_t* valid2, float mfNNratio,
                          int check_orientation, int32_t* matches12) {
  for (int k = 0; k < KF1->n; k++) matches12[k] = -1;
  std::vector<bool> vbMatched2(KF2->n, false);
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  for (int node = 0; node < n_nodes; node++) {
    for (int a = start1[node]; a < start1[node + 1]; a++) {
      const int idx1 = idx1v[a];
      if (!valid1[idx1]) continue;
      int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
      for (int b = start2[node]; b < start2[node + 1]; b++) {
        const int idx2 = idx2v[b];
        if (vbMatched2[idx2] || !valid2[idx2]) continue;
        const int dist = lldo_descriptor_distance(KF1->desc + 8 * idx1, KF2->desc + 8 * idx2);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
        else if (dist < bestDist2) { bestDist2 = dist; }
      }
      if (bestDist1 < TH_LOW) {
        if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
          matches12[idx1] = bestIdx2;
          vbMatched2[bestIdx2] = true;
          if (check_orientation) rotHist[rot_bin(KF1->angle[idx1], KF2->angle[bestIdx2])].push_back(idx1);
          nmatches++;
        }
      }
    }
  }
  if (check_orientation) nmatches = apply_rotation(rotHist, matches12, nmatches);
  return nmatches;
}

// ORBmatcher::SearchForTriangulation   ORBmatcher.cc:657-823 (CheckDistEpipolarLine :138-157 inlined)
//   has_mp1[k] = pKF1->GetMapPoint(k) != NULL, has_mp2 likewise; F12 row-major 3x3 float; (ex,ey) the epipole (:669-671).
//   vbMatched2 is never set by the reference, so it is omitted.  matches12[idx1] = idx2 or -1.
int lldo_search_for_triangulation(const lldo_frame* KF1, const lldo_frame* KF2, int n_nodes, const int32_t* start1, const int32_t* idx1v,
                                  const int32_t* start2, const int32_t* idx2v, const uint8_t* has_mp1, const uint8_t* has_mp2,
                                  const float* F12, float ex, float ey, int bOnlyStereo, int check_orientation, int32_t* matches12) {
  for (int k = 0; k < KF1->n; k++) matches12[k] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  for (int node = 0; node < n_nodes; node++) {
    for (int a = start1[node]; a < start1[node + 1]; a++) {
      const int idx1 = idx1v[a];
      if (has_mp1[idx1]) continue;
      const bool bStereo1 = KF1->uright[idx1] >= 0;
      if (bOnlyStereo && !bStereo1) continue;
      const float kp1x = KF1->xy[2 * idx1], kp1y = KF1->xy[2 * idx1 + 1];
      int bestDist = TH_LOW, bestIdx2 = -1;
      for (int b = start2[node]; b < start2[node + 1]; b++) {
        const int idx2 = idx2v[b];
        if (has_mp2[idx2]) continue;
        const bool bStereo2 = KF2->uright[idx2] >= 0;
        if (bOnlyStereo && !bStereo2) continue;
        const int dist = lldo_descriptor_distance(KF1->desc + 8 * idx1, KF2->desc + 8 * idx2);
        if (dist > TH_LOW || dist > bestDist) continue;
        const float kp2x = KF2->xy[2 * idx2], kp2y = KF2->xy[2 * idx2 + 1];
        if (!bStereo1 && !bStereo2) {
          const float distex = ex - kp2x, distey = ey - kp2y;
          if (distex * distex + distey * distey < 100 * KF2->scale[KF2->octave[idx2]]) continue;
        }
        // CheckDistEpipolarLine
        const float la = kp1x * F12[0] + kp1y * F12[3] + F12[6];
        const float lb = kp1x * F12[1] + kp1y * F12[4] + F12[7];
        const float lc = kp1x * F12[2] + kp1y * F12[5] + F12[8];
        const float num = la * kp2x + lb * kp2y + lc;
        const float den = la * la + lb * lb;
        if (den == 0) continue;
        const float dsqr = num * num / den;
        if (dsqr < 3.84 * KF2->sigma2[KF2->octave[idx2]]) { bestIdx2 = idx2; bestDist = dist; }
      }
      if (bestIdx2 >= 0) {
        matches12[idx1] = bestIdx2;
        nmatches++;
        if (check_orientation) rotHist[rot_bin(KF1->angle[idx1], KF2->angle[bestIdx2])].push_back(idx1);
      }
    }
  }
  if (check_orientation) nmatches = apply_rotation(rotHist, matches12, nmatches);
  return nmatches;
}

// The query's epipolar line as CheckDistEpipolarLine forms it (:141-143); the adapter hands these to the device
void lldo_epipolar_line(const float* F12, float x, float y, float* abc) {
  abc[0] = x * F12[0] + y * F12[3] + F12[6];
  abc[1] = x * F12[1] + y * F12[4] + F12[7];
  abc[2] = x * F12[2] + y * F12[5] + F12[8];
}

// Hamming search of Frame::ComputeStereoMatches   Frame.cc:530-613: best_r[iL] = bestIdxR when bestDist < thOrbDist else -1;
// best_dist[iL] = bestDist (TH_HIGH when nothing beat it).  n_rows = image rows (row table size).
void lldo_stereo_search(const lldo_frame* L, const lldo_frame* R, int n_rows, float minD, float maxD, int32_t* best_r, int32_t* best_dist) {
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  std::vector<std::vector<int>> vRowIndices(n_rows);
  for (int iR = 0; iR < R->n; iR++) {
    const float kpY = R->xy[2 * iR + 1];
    const float r = 2.0f * R->scale[R->octave[iR]];
    const int maxr = (int)std::ceil(kpY + r);
    const int minr = (int)std::floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < n_rows) vRowIndices[yi].push_back(iR);   // the reference indexes unchecked
  }
  for (int iL = 0; iL < L->n; iL++) {
    best_r[iL] = -1; best_dist[iL] = TH_HIGH;
    const int levelL = L->octave[iL];
    const float vL = L->xy[2 * iL + 1], uL = L->xy[2 * iL];
    const std::vector<int>& vCandidates = vRowIndices[(size_t)vL];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH, bestIdxR = 0;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const int iR = vCandidates[iC];
      if (R->octave[iR] < levelL - 1 || R->octave[iR] > levelL + 1) continue;
      const float uR = R->xy[2 * iR];
      if (uR >= minU && uR <= maxU) {
        const int dist = lldo_descriptor_distance(L->desc + 8 * iL, R->desc + 8 * iR);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    best_dist[iL] = bestDist;
    if (bestDist < thOrbDist) best_r[iL] = bestIdxR;
  }
}

// Frame::ComputeStereoMatches, whole routine (src/Frame.cc:530-704), through the structs of include/lld_amd.h.  cv::Mat slicing and
// cv::norm(IL,IR,NORM_L1) on CV_32F patches of 8-bit pixels are exact integer arithmetic; patches that would leave the image (where
// OpenCV aborts) get no match, as in the product.  Returns the number of entries of vDistIdx that survive the median cut.
int lldo_compute_stereo_matches(const lld_keypoints* left, const lld_keypoints* right, const lld_stereo_pyramids* pyr, float mb, float mbf,
                                float* mvuRight, float* mvDepth, int32_t* best_r, int32_t* sad) {
  const int N = left->n, Nr = right->n;
  for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; if (best_r) best_r[i] = -1; if (sad) sad[i] = -1; }
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = pyr->rows[0];
  std::vector<std::vector<size_t>> vRowIndices(nRows);
  for (int iR = 0; iR < Nr; iR++) {
    const float kpY = right->xy[2 * iR + 1];
    const float r = 2.0f * pyr->scale_factors[right->octave[iR]];
    const int maxr = (int)std::ceil(kpY + r);
    const int minr = (int)std::floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);   // the reference indexes unchecked
  }
  const float minZ = mb;
  const float minD = 0;
  const float maxD = mbf / minZ;
  std::vector<std::pair<int, int>> vDistIdx;
  vDistIdx.reserve(N);
  for (int iL = 0; iL < N; iL++) {
    const int levelL = left->octave[iL];
    const float vL = left->xy[2 * iL + 1];
    const float uL = left->xy[2 * iL];
    if (!((long long)vL >= 0 && (long long)vL < nRows)) continue;                                       // unchecked in the reference
    const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD;
    const float maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const size_t iR = vCandidates[iC];
      if (right->octave[iR] < levelL - 1 || right->octave[iR] > levelL + 1) continue;
      const float uR = right->xy[2 * iR];
      if (uR >= minU && uR <= maxU) {
        const int dist = lldo_descriptor_distance(left->desc + 8 * iL, right->desc + 8 * iR);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      if (best_r) best_r[iL] = (int32_t)bestIdxR;
      const float uR0 = right->xy[2 * bestIdxR];
      const float scaleFactor = pyr->inv_scale_factors[levelL];
      const float scaleduL = std::round(uL * scaleFactor);
      const float scaledvL = std::round(vL * scaleFactor);
      const float scaleduR0 = std::round(uR0 * scaleFactor);
      const int w = 5;
      const int cols = pyr->cols[levelL], rows = pyr->rows[levelL];
      const uint8_t* imL = pyr->left[levelL]; const uint8_t* imR = pyr->right[levelL];
      const int sL = pyr->left_step[levelL], sR = pyr->right_step[levelL];
      const int x0 = (int)scaleduL, y0 = (int)scaledvL, xr = (int)scaleduR0;
      int bestDist = INT_MAX;
      int bestincR = 0;
      const int L = 5;
      std::vector<float> vDists;
      vDists.resize(2 * L + 1);
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= cols) continue;
      // rowRange / colRange outside the image: cv::Mat would abort; no match here
      if (x0 - w < 0 || x0 + w >= cols || y0 - w < 0 || y0 + w >= rows || xr - L - w < 0 || xr + L + w >= cols) continue;
      float IL[11][11];
      const float cL = (float)imL[(size_t)y0 * sL + x0];
      for (int r = 0; r < 2 * w + 1; r++) for (int c = 0; c < 2 * w + 1; c++) IL[r][c] = (float)imL[(size_t)(y0 - w + r) * sL + (x0 - w + c)] - cL;
      for (int incR = -L; incR <= +L; incR++) {
        const float cR = (float)imR[(size_t)y0 * sR + (xr + incR)];
        double acc = 0.0;                                                                               // cv::norm accumulates in double
        for (int r = 0; r < 2 * w + 1; r++)
          for (int c = 0; c < 2 * w + 1; c++) {
            const float ir = (float)imR[(size_t)(y0 - w + r) * sR + (xr + incR - w + c)] - cR;
            acc += (double)std::fabs(IL[r][c] - ir);
          }
        float dist = (float)acc;
        if (dist < bestDist) { bestDist = dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = pyr->scale_factors[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
        mvDepth[iL] = mbf / disparity;
        mvuRight[iL] = bestuR;
        vDistIdx.push_back(std::pair<int, int>(bestDist, iL));
        if (sad) sad[iL] = bestDist;
      }
    }
  }
  if (vDistIdx.empty()) return 0;                                                                       // the reference reads vDistIdx[0] regardless
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  int kept = (int)vDistIdx.size();
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist) break;
    mvuRight[vDistIdx[i].second] = -1;
    mvDepth[vDistIdx[i].second] = -1;
    kept--;
  }
  return kept;
}

// Frame::isInFrustum for every MapPoint (src/Frame.cc:333-389) with the per-point part of Tracking::SearchLocalPoints
// (src/Tracking.cc:1637-1649).  OpenCV is absent, so its three calls are restated as this build reads them: `mRcw*P+mtcw` is one
// cv::gemm = float(sum_k double(R_ik) double(P_k) + double(t_i)); cv::norm and Mat::dot accumulate in double; the rest is float
// arithmetic in source order (-ffp-contract=off).  Returns nToMatch.
int lldo_is_in_frustum(const lld_frame_view* V, const lld_map_points* mp, float viewingCosLimit, uint8_t* in_view, float* proj_uvr,
                       int32_t* level, float* view_cos) {
  int nToMatch = 0;
  for (int i = 0; i < mp->n; i++) {
    in_view[i] = 0;                                                  // pMP->mbTrackInView = false
    if (mp->skip && mp->skip[i]) continue;
    const float* P = mp->world_pos + 3 * i;
    float Pc[3];
    for (int r = 0; r < 3; r++) {
      double s0 = 0.0;
      for (int k = 0; k < 3; k++) s0 += (double)V->Rcw[3 * r + k] * (double)P[k];
      Pc[r] = (float)(s0 + (double)V->tcw[r]);
    }
    const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
    if (PcZ < 0.0f) continue;
    const float invz = 1.0f / PcZ;
    const float u = V->fx * PcX * invz + V->cx;
    const float v = V->fy * PcY * invz + V->cy;
    if (u < V->min_x || u > V->max_x) continue;
    if (v < V->min_y || v > V->max_y) continue;
    const float maxDistance = 1.2f * mp->max_distance[i], minDistance = 0.8f * mp->min_distance[i];
    const float PO[3] = {P[0] - V->Ow[0], P[1] - V->Ow[1], P[2] - V->Ow[2]};
    double n2 = 0.0; for (int k = 0; k < 3; k++) n2 += (double)PO[k] * (double)PO[k];
    const float dist = (float)std::sqrt(n2);
    if (dist < minDistance || dist > maxDistance) continue;
    double dotv = 0.0; for (int k = 0; k < 3; k++) dotv += (double)PO[k] * (double)mp->normal[3 * i + k];
    const float viewCos = (float)(dotv / dist);
    if (viewCos < viewingCosLimit) continue;
    const float ratio = mp->max_distance[i] / dist;                   // MapPoint::PredictScale
    int nScale = (int)std::ceil(std::log(ratio) / V->log_scale_factor);
    if (nScale < 0) nScale = 0; else if (nScale >= V->n_levels) nScale = V->n_levels - 1;
    in_view[i] = 1;
    proj_uvr[3 * i] = u; proj_uvr[3 * i + 1] = v; proj_uvr[3 * i + 2] = u - V->bf * invz;
    level[i] = nScale; view_cos[i] = viewCos;
    nToMatch++;
  }
  return nToMatch;
}

// Projection loop of ORBmatcher::SearchByProjection(Current, Last, th, bMono) (src/ORBmatcher.cc:1352-1377): valid_out[i] = the
// point survives every `continue` before the window search; uv / ur as the reference computes them (ur = u - mbf*invzc, :1402).
void lldo_project_last_frame(const lld_frame_view* V, const lld_last_frame_points* L, uint8_t* valid_out, float* uv, float* ur) {
  for (int i = 0; i < L->n; i++) {
    valid_out[i] = 0; uv[2 * i] = uv[2 * i + 1] = 0.f; ur[i] = 0.f;
    if (!L->valid[i]) continue;
    const float* P = L->world_pos + 3 * i;
    float x3Dc[3];
    for (int r = 0; r < 3; r++) {
      double s0 = 0.0;
      for (int k = 0; k < 3; k++) s0 += (double)V->Rcw[3 * r + k] * (double)P[k];
      x3Dc[r] = (float)(s0 + (double)V->tcw[r]);
    }
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = 1.0 / x3Dc[2];
    if (invzc < 0) continue;
    const float u = V->fx * xc * invzc + V->cx;
    const float v = V->fy * yc * invzc + V->cy;
    if (u < V->min_x || u > V->max_x) continue;
    if (v < V->min_y || v > V->max_y) continue;
    valid_out[i] = 1; uv[2 * i] = u; uv[2 * i + 1] = v; ur[i] = u - V->bf * invzc;
  }
}

// Projection loop of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:841-890): valid_out, uv, ur, predicted level.
void lldo_project_fuse(const lld_frame_view* V, const lld_map_points* mp, uint8_t* valid_out, float* uv, float* ur_out, int32_t* level) {
  for (int i = 0; i < mp->n; i++) {
    valid_out[i] = 0; uv[2 * i] = uv[2 * i + 1] = 0.f; ur_out[i] = 0.f; level[i] = 0;
    if (mp->skip && mp->skip[i]) continue;
    const float* P = mp->world_pos + 3 * i;
    float p3Dc[3];
    for (int r = 0; r < 3; r++) {
      double s0 = 0.0;
      for (int k = 0; k < 3; k++) s0 += (double)V->Rcw[3 * r + k] * (double)P[k];
      p3Dc[r] = (float)(s0 + (double)V->tcw[r]);
    }
    if (p3Dc[2] < 0.0f) continue;
    const float invz = 1 / p3Dc[2];
    const float x = p3Dc[0] * invz;
    const float y = p3Dc[1] * invz;
    const float u = V->fx * x + V->cx;
    const float v = V->fy * y + V->cy;
    if (!(u >= V->min_x && u < V->max_x && v >= V->min_y && v < V->max_y)) continue;      // KeyFrame::IsInImage
    const float ur = u - V->bf * invz;
    const float maxDistance = 1.2f * mp->max_distance[i], minDistance = 0.8f * mp->min_distance[i];
    const float PO[3] = {P[0] - V->Ow[0], P[1] - V->Ow[1], P[2] - V->Ow[2]};
    double n2 = 0.0; for (int k = 0; k < 3; k++) n2 += (double)PO[k] * (double)PO[k];
    const float dist3D = (float)std::sqrt(n2);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    double dotv = 0.0; for (int k = 0; k < 3; k++) dotv += (double)PO[k] * (double)mp->normal[3 * i + k];
    if (dotv < 0.5 * dist3D) continue;
    const float ratio = mp->max_distance[i] / dist3D;
    int nScale = (int)std::ceil(std::log(ratio) / V->log_scale_factor);
    if (nScale < 0) nScale = 0; else if (nScale >= V->n_levels) nScale = V->n_levels - 1;
    valid_out[i] = 1; uv[2 * i] = u; uv[2 * i + 1] = v; ur_out[i] = ur; level[i] = nScale;
  }
}

// Projection loops of the relocalisation / loop-closing matchers, `routine` as LLD_ORB_PROJ_* (include/lld_amd.h):
//   0 SearchByProjection(KeyFrame*, Scw, ...) src/ORBmatcher.cc:311-358      2 Fuse(KeyFrame*, Scw, ...) :1000-1048
//   1 SearchByProjection(Frame&, KeyFrame*, ...) :1487-1531                   3 one direction of SearchBySim3 :1152-1191 / :1232-1271
// valid_out, uv, predicted level of the points that reach GetFeaturesInArea.  sR / t: second transform of routine 3 only.
void lldo_project_general(const lld_frame_view* V, const lld_map_points* mp, int routine, const float* sR, const float* t,
                          uint8_t* valid_out, float* uv, int32_t* level) {
  for (int i = 0; i < mp->n; i++) {
    valid_out[i] = 0; uv[2 * i] = uv[2 * i + 1] = 0.f; level[i] = 0;
    if (mp->skip && mp->skip[i]) continue;
    const float* P = mp->world_pos + 3 * i;
    float p3Dc[3];
    for (int r = 0; r < 3; r++) {                                              // Rcw*p3Dw+tcw
      double s0 = 0.0;
      for (int k = 0; k < 3; k++) s0 += (double)V->Rcw[3 * r + k] * (double)P[k];
      p3Dc[r] = (float)(s0 + (double)V->tcw[r]);
    }
    if (routine == 3) {                                                        // p3Dc2 = sR21*p3Dc1 + t21
      float q[3];
      for (int r = 0; r < 3; r++) {
        double s0 = 0.0;
        for (int k = 0; k < 3; k++) s0 += (double)sR[3 * r + k] * (double)p3Dc[k];
        q[r] = (float)(s0 + (double)t[r]);
      }
      p3Dc[0] = q[0]; p3Dc[1] = q[1]; p3Dc[2] = q[2];
    }
    float u, v;
    if (routine == 1) {
      const float xc = p3Dc[0];
      const float yc = p3Dc[1];
      const float invzc = 1.0 / p3Dc[2];
      u = V->fx * xc * invzc + V->cx;
      v = V->fy * yc * invzc + V->cy;
      if (u < V->min_x || u > V->max_x) continue;
      if (v < V->min_y || v > V->max_y) continue;
    } else {
      if (p3Dc[2] < 0.0) continue;
      float invz;
      if (routine == 0) invz = 1 / p3Dc[2]; else invz = 1.0 / p3Dc[2];
      const float x = p3Dc[0] * invz;
      const float y = p3Dc[1] * invz;
      u = V->fx * x + V->cx;
      v = V->fy * y + V->cy;
      if (!(u >= V->min_x && u < V->max_x && v >= V->min_y && v < V->max_y)) continue;      // KeyFrame::IsInImage
    }
    const float maxDistance = 1.2f * mp->max_distance[i], minDistance = 0.8f * mp->min_distance[i];
    float dist;
    if (routine == 3) {
      double n2 = 0.0; for (int k = 0; k < 3; k++) n2 += (double)p3Dc[k] * (double)p3Dc[k];
      dist = (float)std::sqrt(n2);
      if (dist < minDistance || dist > maxDistance) continue;
    } else {
      const float PO[3] = {P[0] - V->Ow[0], P[1] - V->Ow[1], P[2] - V->Ow[2]};
      double n2 = 0.0; for (int k = 0; k < 3; k++) n2 += (double)PO[k] * (double)PO[k];
      dist = (float)std::sqrt(n2);
      if (dist < minDistance || dist > maxDistance) continue;
      if (routine != 1) {
        double dotv = 0.0; for (int k = 0; k < 3; k++) dotv += (double)PO[k] * (double)mp->normal[3 * i + k];
        if (dotv < 0.5 * dist) continue;
      }
    }
    const float ratio = mp->max_distance[i] / dist;
    int nScale = (int)std::ceil(std::log(ratio) / V->log_scale_factor);
    if (nScale < 0) nScale = 0; else if (nScale >= V->n_levels) nScale = V->n_levels - 1;
    valid_out[i] = 1; uv[2 * i] = u; uv[2 * i + 1] = v; level[i] = nScale;
  }
}

// The libm under MapPoint::PredictScale's log(ratio), restated: glibc's logf (sysdeps/ieee754/flt-32/e_logf.c + logf_data.c, glibc >= 2.27;
// constants read back from this image's libm.so.6, glibc 2.35).  The routines above keep calling std::log - the reference's behaviour on
// this platform; the DEVICE cannot call it and carries this algorithm (lld_orb_search.hip, glibc_logf).  lldo_glibc_logf_differences pins
// the restatement to std::log(float) of the host the tests run on: if that libm ever computes logf another way, the test says so.
float lldo_glibc_logf(float x) {
  static const double T[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
      {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
      {0x1p+0, 0x0p+0},                              {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
      {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},   {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
      {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
  static const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix; std::memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return std::log(x);       // zero, subnormal, negative, inf, nan: not a distance ratio
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u), k = (int)((int32_t)tmp >> 23);
  const uint32_t iz = ix - (tmp & 0xff800000u);
  float zf; std::memcpy(&zf, &iz, 4);
  const double z = zf;
  volatile double zi = z * T[i][0];                                             // volatile: no contraction, whatever the flags (liblld_oracle_fma.so)
  const double r = zi - 1.0;
  volatile double kl = (double)k * Ln2, r2 = r * r, a1r = A1 * r;
  const double y0 = T[i][1] + kl;
  double y = a1r + A2;
  volatile double a0r2 = A0 * r2;
  y = a0r2 + y;
  volatile double yr2 = y * r2;
  y = yr2 + (y0 + r);
  return (float)y;
}

// how many of n pseudo-random positive normal floats (+ the neighbourhoods of 1.2^k, PredictScale's boundaries) give another bit pattern
// than std::log(float); *first_bad = the first such argument
long lldo_glibc_logf_differences(long n, unsigned long long seed, float* first_bad) {
  long bad = 0;
  unsigned long long s = seed * 6364136223846793005ull + 1442695040888963407ull;
  auto check = [&](float x) {
    volatile float xv = x;
    const float ref = std::log(xv), got = lldo_glibc_logf(x);
    if (std::memcmp(&ref, &got, 4) != 0) { if (!bad && first_bad) *first_bad = x; bad++; }
  };
  for (long t = 0; t < n; t++) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t b = (uint32_t)(s >> 33) & 0x7fffffffu;
    if ((b >> 23) == 0 || (b >> 23) == 255) continue;
    float x; std::memcpy(&x, &b, 4);
    check(x);
  }
  for (int k = -8; k <= 16; k++) {
    float c = (float)std::pow(1.2, k); uint32_t cb; std::memcpy(&cb, &c, 4);
    for (int d = -4096; d <= 4096; d++) { const uint32_t b = cb + (uint32_t)d; float x; std::memcpy(&x, &b, 4); check(x); }
  }
  return bad;
}

}  // extern "C"
