// lld_ba_chol_plan.h — the symbolic side of the structure-following reduced solve (round 5).
//
// The reference factors the reduced camera system with a SPARSE LDL^T after an elimination ordering of its block pattern
// (Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:60 CholmodSupport-less `SimplicialLDLT` with `computeSymbolicDecomposition`, :147-232:
// AMD ordering on the 6x6-block matrix, then `analyzePatternWithPermutation`; :94-124 the numeric factorisation and solve; chosen at
// src/Optimizer.cc:1023).  This file is that symbolic step for the matrix-core kernel ba_chol_sparse_kernel (lld_ba_chol_sparse.h):
// per window, once per batch, the host
//   1. reads the block pattern of S (which camera pairs share a landmark: the CSR the Schur reduce sums from),
//   2. picks an elimination order of the cameras - natural or reverse Cuthill-McKee, as ONE chain of 16x16 tile columns or, where a
//      vertex separator splits the cameras, as TWO chains that are eliminated side by side by two panel wavefronts and meet in the
//      separator's columns (a two-leaf elimination tree: the serial chain of tile factors is what bounds the kernel),
//   3. runs the symbolic factorisation on the tile graph (fill included) and
//   4. writes the kernel's schedule as tables: per step the tile columns, per tile wavefront the register slots of its tiles and, per
//      step, bit masks of the slots that form L_IJ, take a trailing update or are published, and the positions of a column's tiles in
//      the LDS panel buffers.
// A plan that does not fit the kernel (more than 144 non-zero tiles after fill: a dense 50-camera window) leaves mode = 0 and the window
// goes to the dense kernel ba_chol_mfma_kernel.  Plain C++ (no HIP): compiled into liblld_amd.so and, through lld_ba_chol_plan, driven
// by tests/test_chol_plan.py, which executes the tables tile by tile in numpy against numpy.linalg.solve.
#ifndef LLD_BA_CHOL_PLAN_H
#define LLD_BA_CHOL_PLAN_H

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace lldba {

constexpr int kSpMaxT = 22;          // tile rows of a plan (two padded chains + separator of 50 cameras: at most 22)
constexpr int kSpStride = 24;        // row length of the per-step tables
constexpr int kSpTileWaves = 10;     // wavefronts 2..11 hold tiles, wavefronts 0 and 1 are the panel wavefronts of the two chains
constexpr int kSpSlots = 14;         // register tiles per tile wavefront (14 x 4 doubles per lane; 153 VGPRs: three wavefronts per SIMD)
constexpr int kSpPos = 20;           // tiles per LDS panel buffer (all off-diagonal tiles of the columns of one step)
constexpr int kSpThreads = 768;
constexpr int kSpNone = 255;

struct alignas(16) CholPlan {          // (16-byte multiple: the kernel copies the plan into LDS sixteen bytes at a time)
  uint8_t mode;                                   // 0: dense kernel, 1: structure-following kernel
  uint8_t NT, T, chains;                          // tile rows, steps, 1 or 2
  uint8_t cols[kSpStride][2];                     // [step][chain] tile column eliminated in that step (kSpNone: none); steps >= T: none
  uint8_t slotI[kSpTileWaves][kSpSlots];          // tile (I, K) of a wavefront's register slot (kSpNone: unused)
  uint8_t slotK[kSpTileWaves][kSpSlots];
  uint8_t pos[kSpMaxT][kSpStride];                // pos[J][I]: position of L(I, J) in the panel buffer of J's step (kSpNone: structural zero)
  uint32_t cA[kSpTileWaves][kSpStride];           // per step: slots of the off-diagonal tiles of the step's chain-0 column (L_IJ = A_IJ L_JJ^-T, back substitution)
  uint32_t cB[kSpTileWaves][kSpStride];           // ... chain-1 column
  uint32_t dA[kSpTileWaves][kSpStride];           // per step: slots that take the trailing update of the chain-0 column
  uint32_t dB[kSpTileWaves][kSpStride];
  uint32_t pub[kSpTileWaves][kSpStride];          // per step: slots published after the update (next step's columns -> panel buffer, the diagonal tiles of the step after -> Dall)
  uint32_t own[kSpTileWaves][kSpStride];          // per step: the subset of cA | cB whose L_IJ the next column's PANEL wavefront forms (tile (Jn, J), Jn a column of the next
                                                  // step): the panel's chain does not wait for the tile wavefronts; the slot's wavefront fetches L from the panel buffer afterwards
  uint32_t pub0[kSpTileWaves];                    // ... in the prologue (step 0's columns, step 1's diagonal tiles)
  uint32_t padmask[kSpTileWaves];                 // slots whose tile touches padding rows (identity there)
  uint32_t yrows[kSpStride][2];                   // per step and chain: tile rows I with L(I, J) != 0: y_I -= L_IJ y_J on the tile wavefronts (the right-hand side is not on the panel's chain)
  int16_t rowmap[kSpMaxT * 16];                   // permuted scalar row -> row of S (-1: padding)
  int32_t n_tiles, n_updates, n_cams, est_ns;     // non-zero tiles incl. fill, tile updates, cameras, estimated time (diagnostics)
};
static_assert(sizeof(CholPlan) % 16 == 0, "the kernel copies the plan into LDS sixteen bytes at a time");

struct CholPlanChoice { int order; int cut; int side; };     // diagnostics: 0 natural / 1 RCM; cut position; separator taken from the left / right part

namespace cholplan {

inline int tiles_of(int cams) { return (6 * cams + 15) / 16; }

// Symbolic factorisation + tables for the segments A | B | Sep (B empty: one chain).  false: does not fit the kernel.
inline bool build_from_segments(int nf, const uint64_t* adj, const std::vector<int>& segA, const std::vector<int>& segB, const std::vector<int>& segS, CholPlan& P) {
  std::memset(&P, 0, sizeof P);
  std::memset(P.cols, kSpNone, sizeof P.cols); std::memset(P.slotI, kSpNone, sizeof P.slotI); std::memset(P.slotK, kSpNone, sizeof P.slotK);
  std::memset(P.pos, kSpNone, sizeof P.pos);
  for (auto& r : P.rowmap) r = -1;
  const bool two = !segB.empty();
  int camrow[64];
  int r = 0;
  auto place = [&](const std::vector<int>& seg, bool pad) {
    for (int c : seg) { camrow[c] = r; for (int k = 0; k < 6; k++) { if (r >= kSpMaxT * 16) return false; P.rowmap[r++] = (int16_t)(6 * c + k); } }
    if (pad) r = (r + 15) & ~15;
    return r <= kSpMaxT * 16;
  };
  if (!place(segA, two)) return false;
  const int na = two ? r / 16 : 0;
  if (two && !place(segB, true)) return false;
  const int nb = two ? r / 16 - na : 0;
  if (!place(segS, true)) return false;
  const int NT = r / 16;
  if (NT < 1 || NT > kSpMaxT) return false;
  const int ns = NT - na - nb;
  if ((int)(segA.size() + segB.size() + segS.size()) != nf) return false;
  // tile pattern (bit K of nzt[I]: tile (I, K), K <= I)
  uint32_t nzt[kSpMaxT] = {0};
  for (int I = 0; I < NT; I++) nzt[I] |= 1u << I;
  for (int a = 0; a < nf; a++)
    for (int b = 0; b <= a; b++) {
      if (a != b && !((adj[a] >> b) & 1)) continue;
      const int ra = camrow[a], rb = camrow[b];
      for (int ta = ra / 16; ta <= (ra + 5) / 16; ta++)
        for (int tb = rb / 16; tb <= (rb + 5) / 16; tb++) { const int I = std::max(ta, tb), K = std::min(ta, tb); nzt[I] |= 1u << K; }
    }
  if (two)                                         // the two chains must not touch (the caller's separator guarantees it)
    for (int I = na; I < na + nb; I++) if (nzt[I] & ((1u << na) - 1)) return false;
  int n_updates = 0;
  for (int J = 0; J < NT; J++) {
    int rows[kSpMaxT], nr = 0;
    for (int I = J + 1; I < NT; I++) if ((nzt[I] >> J) & 1) rows[nr++] = I;
    for (int a = 0; a < nr; a++) for (int b = 0; b <= a; b++) { nzt[rows[a]] |= 1u << rows[b]; n_updates++; }
  }
  auto nz = [&](int I, int K) { return ((nzt[I] >> K) & 1) != 0; };
  // steps: chain 0 = A's columns then the separator's, chain 1 = B's columns; both chains end in the same step
  const int lead = std::max(na, nb);
  const int T = two ? lead + ns : NT;
  if (T > kSpMaxT) return false;
  int step_of[kSpMaxT];
  if (two) {
    for (int j = 0; j < na; j++) { P.cols[lead - na + j][0] = (uint8_t)j; step_of[j] = lead - na + j; }
    for (int j = 0; j < nb; j++) { P.cols[lead - nb + j][1] = (uint8_t)(na + j); step_of[na + j] = lead - nb + j; }
    for (int j = 0; j < ns; j++) { P.cols[lead + j][0] = (uint8_t)(na + nb + j); step_of[na + nb + j] = lead + j; }
  } else
    for (int j = 0; j < NT; j++) { P.cols[j][0] = (uint8_t)j; step_of[j] = j; }
  // every non-zero L(I, J) must be eliminated before its row's column: step_of[J] < step_of[I]
  for (int J = 0; J < NT; J++) for (int I = J + 1; I < NT; I++) if (nz(I, J) && !(step_of[J] < step_of[I])) return false;
  // panel-buffer positions
  for (int s = 0; s < T; s++) {
    int p = 0;
    for (int ch = 0; ch < 2; ch++) {
      const int J = P.cols[s][ch];
      if (J == kSpNone) continue;
      for (int I = J + 1; I < NT; I++) if (nz(I, J)) { if (p >= kSpPos) return false; P.pos[J][I] = (uint8_t)p++; }
    }
  }
  // Register slots.  What a tile costs its wavefront and when: one L_IJ product in its column's step, one trailing update in the step of
  // every earlier column J with L(I, J), L(K, J) != 0, one publish.  The tile wavefronts run a step in lock-step (two barriers), so the step
  // is as slow as its most loaded wavefront: tiles go, busiest first, to the wavefront whose load in THEIR steps is smallest (ties: fewest
  // tiles).  The diagonal tiles of step 0's columns belong to the panel wavefronts alone.
  // The load that binds is the fp64 matrix pipe of the wavefront's SIMD (64 cycles per v_mfma_f64_16x16x4, one pipe per SIMD): the workgroup's
  // eight wavefronts sit two per SIMD (wavefront w on SIMD w mod 4: checked by the stage-budget tool through HW_ID), so tile wavefronts 2 and 3
  // (hardware wavefronts 4, 5) share their SIMD with a PANEL wavefront, which hardly uses the pipe, while 0 / 4 and 1 / 5 share a pipe between
  // them: a tile costs the latter twice what it costs the former (profiles/r05_chol_sparse_stage_budget_*: the tile wavefronts' update phase
  // was the longest stage of a step before).
  // MEASURED, round 5: weighting by pipe sharing ({2, 2, 1, 1, 2, 2}: 22 tiles on the two wavefronts next to the panels, 11 on the others) is
  // SLOWER - one launch 58 -> 62 - 69 us: what a tile wavefront's time is made of is its own serial chain of LDS read -> four dependent
  // matrix-core instructions per update, not the pipe's throughput, so equal counts win.  The weights stay as the knob they are.
  static const int kSlow[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  auto in_step = [&](int K, int s) { return s >= 0 && s < T && (P.cols[s][0] == K || P.cols[s][1] == K); };
  struct TileJob { int I, K, n_ev; uint8_t ev_step[kSpMaxT + 2], ev_cost[kSpMaxT + 2]; };
  std::vector<TileJob> jobs;
  int n_tiles = 0;
  for (int K = 0; K < NT; K++)
    for (int I = K; I < NT; I++) {
      if (!nz(I, K)) continue;
      n_tiles++;
      if (I == K && step_of[K] == 0) continue;
      TileJob j; j.I = I; j.K = K; j.n_ev = 0;
      for (int J = 0; J < K; J++) {
        if (!nz(I, J) || !nz(K, J)) continue;
        if (I == K && in_step(K, step_of[J] + 1)) continue;
        int at = -1;
        for (int q = 0; q < j.n_ev; q++) if (j.ev_step[q] == step_of[J]) at = q;
        if (at < 0) { at = j.n_ev++; j.ev_step[at] = (uint8_t)step_of[J]; j.ev_cost[at] = 0; }
        j.ev_cost[at] += 4;                                   // four matrix-core instructions + their operand reads
      }
      if (I > K) { j.ev_step[j.n_ev] = (uint8_t)step_of[K]; j.ev_cost[j.n_ev++] = 5; }      // L_IJ: product + write-back
      jobs.push_back(j);
    }
  std::stable_sort(jobs.begin(), jobs.end(), [](const TileJob& a, const TileJob& b) { return a.n_ev > b.n_ev; });
  int used[kSpTileWaves] = {0};
  int load[kSpTileWaves][kSpMaxT + 2];
  std::memset(load, 0, sizeof load);
  int8_t wave_of[kSpMaxT][kSpMaxT], slot_of[kSpMaxT][kSpMaxT];
  std::memset(wave_of, -1, sizeof wave_of); std::memset(slot_of, -1, sizeof slot_of);
  for (const TileJob& j : jobs) {
    int best = -1; long best_cost = 0;
    for (int w = 0; w < kSpTileWaves; w++) {
      if (used[w] >= kSpSlots) continue;
      long c = 0;
      for (int q = 0; q < j.n_ev; q++) { const long l = load[w][j.ev_step[q]] + j.ev_cost[q] * kSlow[w]; c += l * l; }
      c = c * 64 + used[w];
      if (best < 0 || c < best_cost) { best = w; best_cost = c; }
    }
    if (best < 0) return false;
    for (int q = 0; q < j.n_ev; q++) load[best][j.ev_step[q]] += j.ev_cost[q] * kSlow[best];
    P.slotI[best][used[best]] = (uint8_t)j.I; P.slotK[best][used[best]] = (uint8_t)j.K;
    wave_of[j.I][j.K] = (int8_t)best; slot_of[j.I][j.K] = (int8_t)used[best];
    used[best]++;
  }
  for (int K = 0; K < NT; K++)
    for (int I = K; I < NT; I++) {
      const int w = wave_of[I][K];
      if (w < 0) continue;
      const uint32_t bit = 1u << slot_of[I][K];
      bool padded = false;
      for (int q = 0; q < 16; q++) padded = padded || P.rowmap[16 * I + q] < 0 || P.rowmap[16 * K + q] < 0;
      if (padded) P.padmask[w] |= bit;
      const int sK = step_of[K];
      if (I > K) {
        (P.cols[sK][0] == K ? P.cA : P.cB)[w][sK] |= bit;
        if (in_step(I, sK + 1)) P.own[w][sK] |= bit;          // L(Jn, J) of the NEXT step's column: formed by that column's panel wavefront, fetched here after the barrier
        if (sK == 0) P.pub0[w] |= bit; else P.pub[w][sK - 1] |= bit;
      } else {
        if (sK == 1) P.pub0[w] |= bit; else if (sK >= 2) P.pub[w][sK - 2] |= bit;
      }
      // trailing updates: column J of step s updates (I, K) iff L(I, J) and L(K, J) are non-zero; the diagonal tile of a column of step
      // s + 1 takes step s's update on the panel wavefront instead
      for (int J = 0; J < K; J++) {
        if (!nz(I, J) || !nz(K, J)) continue;
        const int s = step_of[J];
        if (I == K && in_step(K, s + 1)) continue;
        (P.cols[s][0] == J ? P.dA : P.dB)[w][s] |= bit;
      }
    }
  for (int s = 0; s < T; s++)
    for (int ch = 0; ch < 2; ch++) {
      const int J = P.cols[s][ch];
      if (J == kSpNone) continue;
      for (int I = J + 1; I < NT; I++) if (nz(I, J)) P.yrows[s][ch] |= 1u << I;
    }
  P.mode = 1; P.NT = (uint8_t)NT; P.T = (uint8_t)T; P.chains = two ? 2 : 1;
  P.n_tiles = n_tiles; P.n_updates = n_updates; P.n_cams = nf;
  // what a step costs is the panel wavefront's tile factor (about 2.4 us with the column's L_IJ behind a barrier); the tile wavefronts'
  // updates hide behind it unless a step has many (7 us of one CU's matrix pipes for 100 tile updates); load and back substitution
  P.est_ns = 3500 * T + 12 * n_updates + 60 * n_tiles + 8000;
  return true;
}

// reverse Cuthill-McKee order of the camera graph (components one after the other, each from a pseudo-peripheral start)
inline std::vector<int> rcm_order(int nf, const uint64_t* adj) {
  auto deg = [&](int v) { return __builtin_popcountll(adj[v] & ~(1ull << v)); };
  std::vector<int> order; order.reserve(nf);
  uint64_t seen = 0;
  auto bfs = [&](int start, uint64_t allowed, std::vector<int>& out) {
    out.clear();
    uint64_t vis = 1ull << start;
    out.push_back(start);
    for (size_t h = 0; h < out.size(); h++) {
      std::vector<int> nb;
      const uint64_t m = adj[out[h]] & allowed & ~vis;
      for (int v = 0; v < nf; v++) if ((m >> v) & 1) nb.push_back(v);
      std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return deg(a) < deg(b); });
      for (int v : nb) { vis |= 1ull << v; out.push_back(v); }
    }
    return vis;
  };
  while ((int)order.size() < nf) {
    int start = -1;
    for (int v = 0; v < nf; v++) if (!((seen >> v) & 1) && (start < 0 || deg(v) < deg(start))) start = v;
    std::vector<int> lv;
    const uint64_t allowed = ~seen;
    // pseudo-peripheral: restart the search from the last vertex reached, a few times
    for (int it = 0; it < 3; it++) { bfs(start, allowed, lv); const int far = lv.back(); if (far == start) break; start = far; }
    const uint64_t vis = bfs(start, allowed, lv);
    seen |= vis;
    order.insert(order.end(), lv.begin(), lv.end());
  }
  std::reverse(order.begin(), order.end());
  return order;
}

// force: 0 = best of everything, 1 = natural order, one chain, 2 = best two-chain plan only (tests)
inline bool build(int nf, const uint64_t* adj, int force, CholPlan& best, CholPlanChoice* choice = nullptr) {
  std::memset(&best, 0, sizeof best);
  if (nf < 1 || nf > 64) return false;
  bool have = false;
  CholPlan P;
  auto consider = [&](const std::vector<int>& a, const std::vector<int>& b, const std::vector<int>& s, int order, int cut, int side) {
    if (!build_from_segments(nf, adj, a, b, s, P)) return;
    if (!have || P.est_ns < best.est_ns) { best = P; have = true; if (choice) *choice = CholPlanChoice{order, cut, side}; }
  };
  std::vector<int> nat(nf);
  for (int i = 0; i < nf; i++) nat[i] = i;
  std::vector<std::vector<int>> orders;
  orders.push_back(nat);
  struct Cand { int T, NT, order, m, side; };
  std::vector<Cand> cands;
  // two chains: cut the order at m; the separator is the side of the cut that touches the other one.  Every cut is priced by its chain
  // length first (bit operations), the shortest are built
  auto price_cuts = [&](int oi) {
    const std::vector<int>& pi = orders[(size_t)oi];
    uint64_t left = 0, right = 0;
    for (int i = 0; i < nf; i++) right |= 1ull << pi[i];
    for (int m = 1; m < nf; m++) {
      left |= 1ull << pi[m - 1]; right &= ~(1ull << pi[m - 1]);
      uint64_t sl = 0, sr = 0;                     // cameras of the left / right part that touch the other part
      for (int i = 0; i < nf; i++) {
        const int v = pi[i];
        if (adj[v] & ~(1ull << v) & (i < m ? right : left)) (i < m ? sl : sr) |= 1ull << v;
      }
      for (int side = 0; side < 2; side++) {
        const int nsep = __builtin_popcountll(side == 0 ? sl : sr);
        const int na = m - (side == 0 ? nsep : 0), nb = nf - m - (side == 1 ? nsep : 0);
        if (na < 1 || nb < 1) continue;
        const int Tc = std::max(tiles_of(na), tiles_of(nb)) + tiles_of(nsep), NTc = tiles_of(na) + tiles_of(nb) + tiles_of(nsep);
        if (Tc >= tiles_of(nf) || NTc > kSpMaxT) continue;     // worth it only if the chain gets shorter than one chain over everything
        cands.push_back(Cand{Tc, NTc, oi, m, side});
      }
    }
  };
  if (force != 1) {
    price_cuts(0);
    int best_T = 1 << 30;
    for (const Cand& c : cands) best_T = std::min(best_T, c.T);
    // the caller's order already splits well (a trajectory: keyframes in time order): no need to look for a hidden band
    if (best_T * 4 > tiles_of(nf) * 3) { std::vector<int> rc = rcm_order(nf, adj); if (rc != nat) { orders.push_back(rc); price_cuts(1); } }
  }
  std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.T != b.T ? a.T < b.T : a.NT < b.NT; });
  int built = 0;
  for (const Cand& c : cands) {
    if (built >= 1) break;                               // (the price is the chain length: the first one that fits the kernel is the plan)
    const std::vector<int>& pi = orders[(size_t)c.order];
    uint64_t left = 0, right = 0;
    for (int i = 0; i < nf; i++) (i < c.m ? left : right) |= 1ull << pi[i];
    std::vector<int> a, b, s;
    for (int i = 0; i < nf; i++) {
      const int v = pi[i];
      const bool in_left = i < c.m;
      const bool touches = (adj[v] & ~(1ull << v) & (in_left ? right : left)) != 0;
      if (touches && (in_left ? c.side == 0 : c.side == 1)) s.push_back(v);
      else (in_left ? a : b).push_back(v);
    }
    consider(a, b, s, c.order, c.m, c.side);
    if (P.mode == 1) built++;
  }
  if (!have && force != 2)                               // one chain: the caller's order, or the band an RCM order finds, whichever is estimated faster
    for (size_t oi = 0; oi < orders.size(); oi++) consider(std::vector<int>(), std::vector<int>(), orders[oi], (int)oi, 0, 0);
  return have;
}

}  // namespace cholplan
}  // namespace lldba

#endif
