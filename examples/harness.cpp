// harness.cpp — a C++ caller of liblld_amd.so through include/lld_amd.hpp, the way a patched reference would call it.
// It reads one problem from a flat binary file (written by tests/test_cpp_harness.py), runs it on GPU 0 and writes the
// outputs back, so the tests can compare the C++ route with the golden vectors.
//
//   harness ba   <in> <out>     Optimizer::LocalBundleAdjustment
//   harness gba  <in> <out>     Optimizer::GlobalBundleAdjustment (same file layout, header[7] = nIterations)
//   harness sim3 <in> <out>     Optimizer::OptimizeSim3
//   harness pose <in> <out>     Optimizer::PoseOptimization
//   harness orb  <in> <out>     ORBmatcher::BestTwo (brute force)
//   harness lines <in> <out>    Tracking::AddLinesFrom + Tracking::MatchLinesLastKF
//   harness init <in> <out>     ORBmatcher::SearchForInitialization
//   harness loop <in> <out>     the relocalisation / Scw / SearchBySim3 matchers with their projection loops
//   harness track <in> <out>    the Tracking thread's per-frame chain on one device-resident Frame (TrackWithMotionModel + TrackLocalMap,
//                               lines included), repeated header[9] times with the host wall clock around every repeat
//
// File layout: int32 header (counts), then the arrays in the order they appear in the structs, native endianness.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "lld_amd.hpp"

namespace {

struct Reader {
  FILE* f;
  explicit Reader(const char* path) : f(std::fopen(path, "rb")) { if (!f) throw std::runtime_error(std::string("cannot open ") + path); }
  ~Reader() { std::fclose(f); }
  template <class T> void get(T* p, size_t n) { if (n && std::fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short read"); }
  template <class T> void get(std::vector<T>& v, size_t n) { v.resize(n); get(v.data(), n); }
};
struct Writer {
  FILE* f;
  explicit Writer(const char* path) : f(std::fopen(path, "wb")) { if (!f) throw std::runtime_error(std::string("cannot open ") + path); }
  ~Writer() { std::fclose(f); }
  template <class T> void put(const T* p, size_t n) { if (n && std::fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("short write"); }
  template <class T> void put(const std::vector<T>& v) { put(v.data(), v.size()); }
};

int run_ba(const char* in, const char* out, bool global) {
  Reader r(in);
  int32_t h[8]; r.get(h, 8);                   // n_cams n_free n_points n_pt_obs n_lines n_ln_obs stop 0
  double camg[6]; r.get(camg, 6);              // fx fy cx cy bf gamma
  lld_amd::BAWindow w;
  w.cam = lld_camera{camg[0], camg[1], camg[2], camg[3], camg[4]};
  w.n_free_cams = h[1];
  r.get(w.cam_qt, 7 * (size_t)h[0]);
  r.get(w.pt_xyz, 3 * (size_t)h[2]); r.get(w.pt_obs_start, (size_t)h[2] + 1); r.get(w.pt_obs_cam, h[3]);
  r.get(w.pt_obs_uvr, 3 * (size_t)h[3]); r.get(w.pt_obs_inv_sigma2, h[3]);
  r.get(w.line_x0, 3 * (size_t)h[4]); r.get(w.line_dir, 3 * (size_t)h[4]); r.get(w.ln_obs_start, (size_t)h[4] + 1);
  r.get(w.ln_obs_cam, h[5]); r.get(w.ln_obs_left, 4 * (size_t)h[5]); r.get(w.ln_obs_right, 4 * (size_t)h[5]);
  r.get(w.ln_obs_octave, 2 * (size_t)h[5]);
  lld_amd::Context ctx(0);
  bool stop = h[6] != 0;
  const lld_amd::BAOutput o = global ? lld_amd::Optimizer::GlobalBundleAdjustment(ctx, w, h[7], &stop, true)
                                     : lld_amd::Optimizer::LocalBundleAdjustment(ctx, w, &stop, camg[5]);
  Writer wr(out);
  wr.put(o.cam_qt); wr.put(o.pt_xyz); wr.put(o.line_x0); wr.put(o.line_dir);
  wr.put(o.pt_obs_outlier); wr.put(o.ln_edge_outlier); wr.put(o.line_removed);
  const double chi2[2] = {o.stats.chi2_round1, o.stats.chi2_final};
  wr.put(chi2, 2);
  const int32_t st[4] = {o.stats.lm_iterations[0], o.stats.lm_iterations[1], o.stats.aborted, o.stats.n_lines_removed};
  wr.put(st, 4);
  std::printf("ba: chi2 %.9g -> %.9g, %d+%d LM iterations, %d point / %d line-edge outliers, %d lines removed\n", chi2[0], chi2[1],
              st[0], st[1], o.stats.n_pt_obs_outlier, o.stats.n_ln_edge_outlier, o.stats.n_lines_removed);
  return 0;
}

int run_pose(const char* in, const char* out) {
  Reader r(in);
  int32_t h[2]; r.get(h, 2);                   // n_points n_lines
  double camg[6]; r.get(camg, 6);
  lld_amd::PoseFrame f;
  f.cam = lld_camera{camg[0], camg[1], camg[2], camg[3], camg[4]};
  r.get(f.pose_qt, 7);
  r.get(f.pt_xw, 3 * (size_t)h[0]); r.get(f.pt_uvr, 3 * (size_t)h[0]); r.get(f.pt_inv_sigma2, h[0]);
  r.get(f.ln_x0, 3 * (size_t)h[1]); r.get(f.ln_dir, 3 * (size_t)h[1]); r.get(f.ln_left, 4 * (size_t)h[1]);
  r.get(f.ln_right, 4 * (size_t)h[1]); r.get(f.ln_octave, 2 * (size_t)h[1]);
  lld_amd::Context ctx(0);
  const int32_t n_in = lld_amd::Optimizer::PoseOptimization(ctx, f, camg[5]);
  Writer wr(out);
  wr.put(f.pose_qt, 7); wr.put(&n_in, 1); wr.put(f.mvbOutlier); wr.put(f.mvbOutlierLines);
  std::printf("pose: %d inliers\n", n_in);
  return 0;
}

int run_orb(const char* in, const char* out) {
  Reader r(in);
  int32_t h[2]; r.get(h, 2);                   // nq nt
  std::vector<uint32_t> q, t; r.get(q, 8 * (size_t)h[0]); r.get(t, 8 * (size_t)h[1]);
  lld_amd::Context ctx(0);
  lld_amd::ORBmatcher m(ctx, 0.6f, true);
  const lld_amd::ORBmatcher::Best2 b = m.BestTwo(q.data(), h[0], t.data(), h[1]);
  Writer wr(out);
  wr.put(b.best_idx); wr.put(b.best_dist); wr.put(b.second_idx); wr.put(b.second_dist);
  std::printf("orb: %d x %d\n", h[0], h[1]);
  return 0;
}

// one stereo frame's lines: n, n_right, then left[4n] right[4 n_right] left_octave[n] line_matches[n] occupied[n] desc[n dim]
void read_frame_lines(Reader& r, int dim, lld_amd::FrameLines& F) {
  int32_t h[2]; r.get(h, 2);
  r.get(F.left, 4 * (size_t)h[0]); r.get(F.right, 4 * (size_t)h[1]); r.get(F.left_octave, h[0]); r.get(F.line_matches, h[0]);
  r.get(F.occupied, h[0]); r.get(F.desc, (size_t)h[0] * dim);
}

int run_lines(const char* in, const char* out) {
  Reader r(in);
  int32_t h[3]; r.get(h, 3);                   // dim, n_map, use_grid
  double k[9 + 16 + 16 + 5]; r.get(k, 46);     // K, T_curr, T_last (unused here), b, mnMaxX, mnMaxY, mdThr, thrReprojLineBase
  const int dim = h[0], n_map = h[1];
  lld_amd::MapLineSet L;
  r.get(L.X0, 3 * (size_t)n_map); r.get(L.dir, 3 * (size_t)n_map); r.get(L.X1, 3 * (size_t)n_map); r.get(L.X2, 3 * (size_t)n_map);
  r.get(L.skip, n_map); r.get(L.desc, (size_t)n_map * dim);
  lld_amd::FrameLines F, cur, last;
  read_frame_lines(r, dim, F);
  double k2[9 + 16 + 16 + 5]; r.get(k2, 46);   // the second scene: K, T_curr, T_last, b, mnMaxX, mnMaxY, mdThr, thr
  read_frame_lines(r, dim, cur); read_frame_lines(r, dim, last);
  lld_amd::Context ctx(0);
  std::vector<int> m;
  lld_amd::Tracking(ctx, k, k[41], k[42], k[43], k[44]).AddLinesFrom(L, k + 9, k[45], F, dim, &m, h[2] != 0);
  std::vector<int> ml; std::vector<uint8_t> created; std::vector<double> X0, dir;
  lld_amd::Tracking(ctx, k2, k2[41], k2[42], k2[43], k2[44]).MatchLinesLastKF(k2 + 9, k2 + 25, cur, last, dim, &ml, &created, &X0, &dir, k2[45], h[2] != 0);
  Writer wr(out);
  wr.put(m); wr.put(ml); wr.put(created); wr.put(X0); wr.put(dir);
  std::printf("lines: %d map lines x %d frame lines, %d x %d\n", n_map, F.size(), cur.size(), last.size());
  return 0;
}

int run_init(const char* in, const char* out) {
  Reader r(in);
  int32_t h[4]; r.get(h, 4);                   // n1, n2, windowSize, checkOrientation
  float g[5]; r.get(g, 5);                     // mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv, mfNNratio
  std::vector<uint32_t> d1, d2; std::vector<int32_t> o1, o2; std::vector<float> a1, a2, xy2, prev;
  r.get(d1, 8 * (size_t)h[0]); r.get(o1, h[0]); r.get(a1, h[0]); r.get(prev, 2 * (size_t)h[0]);
  r.get(d2, 8 * (size_t)h[1]); r.get(o2, h[1]); r.get(a2, h[1]); r.get(xy2, 2 * (size_t)h[1]);
  lld_orb_search f2{};
  f2.nt = h[1]; f2.t_desc = d2.data(); f2.t_xy = xy2.data(); f2.t_octave = o2.data(); f2.t_angle = a2.data();
  f2.grid_min_x = g[0]; f2.grid_min_y = g[1]; f2.grid_width_inv = g[2]; f2.grid_height_inv = g[3]; f2.grid_cols = 64; f2.grid_rows = 48;
  lld_amd::Context ctx(0);
  std::vector<int32_t> m12;
  const int n = lld_amd::ORBmatcher(ctx, g[4], h[3] != 0).SearchForInitialization(f2, h[0], d1.data(), o1.data(), a1.data(), prev, m12, h[2]);
  Writer wr(out);
  const int32_t n32 = n; wr.put(&n32, 1); wr.put(m12); wr.put(prev);
  std::printf("init: %d matches of %d\n", n, h[0]);
  return 0;
}

// One keyframe / frame side of a projected search: keypoints, grid constants, scale table, occupancy
struct OrbSide {
  std::vector<uint32_t> desc; std::vector<float> xy, angle, scale; std::vector<int32_t> octave; std::vector<uint8_t> occupied;
  lld_frame_view view; lld_orb_search s;
  void read(Reader& r) {
    int32_t h[2]; r.get(h, 2);                 // n keypoints, n levels
    float g[4]; r.get(g, 4);                   // mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
    r.get(&view, 1);
    r.get(desc, 8 * (size_t)h[0]); r.get(xy, 2 * (size_t)h[0]); r.get(octave, h[0]); r.get(angle, h[0]); r.get(occupied, h[0]); r.get(scale, h[1]);
    s = lld_orb_search{};
    s.nt = h[0]; s.t_desc = desc.data(); s.t_xy = xy.data(); s.t_octave = octave.data(); s.t_angle = angle.data();
    s.grid_min_x = g[0]; s.grid_min_y = g[1]; s.grid_width_inv = g[2]; s.grid_height_inv = g[3]; s.grid_cols = 64; s.grid_rows = 48;
    s.n_levels = h[1]; s.level_scale = scale.data();
  }
};
struct OrbPoints {
  std::vector<float> pos, nrm, maxd, mind, angle; std::vector<uint32_t> desc; std::vector<uint8_t> skip;
  lld_map_points m;
  void read(Reader& r) {
    int32_t n; r.get(&n, 1);
    r.get(pos, 3 * (size_t)n); r.get(nrm, 3 * (size_t)n); r.get(maxd, n); r.get(mind, n); r.get(desc, 8 * (size_t)n); r.get(skip, n); r.get(angle, n);
    m = lld_map_points{};
    m.n = n; m.world_pos = pos.data(); m.normal = nrm.data(); m.max_distance = maxd.data(); m.min_distance = mind.data(); m.desc = desc.data();
    m.skip = skip.data();
  }
};

// The relocalisation / loop-closing matchers through lld_amd::ORBmatcher: SearchByProjection(Frame&, KeyFrame*, ...),
// SearchByProjection(KeyFrame*, Scw, ...), Fuse(KeyFrame*, Scw, ...) on one keyframe and SearchBySim3 on a pair.
int run_loop(const char* in, const char* out) {
  Reader r(in);
  float p[4]; r.get(p, 4);                     // th (reloc), ORBdist, th (KF / Scw), th (Fuse)
  OrbSide kf; kf.read(r);
  OrbPoints pts; pts.read(r);
  OrbSide k1, k2; OrbPoints p1, p2;
  k1.read(r); p1.read(r); k2.read(r); p2.read(r);
  float sim[25]; r.get(sim, 25);               // sR12 (9) t12 (3) sR21 (9) t21 (3) th
  lld_amd::Context ctx(0);
  lld_amd::ORBmatcher m(ctx, 0.9f, true);
  lld_orb_search occupied = kf.s; occupied.t_occupied = kf.occupied.data();
  const lld_amd::ORBmatcher::SearchResult reloc = m.SearchByProjection(occupied, kf.view, pts.m, pts.angle.data(), p[0], (int)p[1]);
  const lld_amd::ORBmatcher::SearchResult scw = m.SearchByProjection(occupied, kf.view, pts.m, (int)p[2]);
  const lld_amd::ORBmatcher::SearchResult fuse = m.Fuse(kf.s, kf.view, pts.m, p[3]);
  std::vector<int32_t> m12;
  const int found = m.SearchBySim3(k1.s, k1.view, p1.m, k2.s, k2.view, p2.m, sim, sim + 9, sim + 12, sim + 21, sim[24], m12);
  Writer wr(out);
  const int32_t counts[4] = {reloc.n_matches, scw.n_matches, fuse.n_matches, found};
  wr.put(counts, 4);
  wr.put(reloc.match); wr.put(reloc.removed); wr.put(scw.match); wr.put(fuse.match); wr.put(m12);
  std::printf("loop: reloc %d, Scw projection %d, Scw fuse %d, SearchBySim3 %d\n", counts[0], counts[1], counts[2], counts[3]);
  return 0;
}

int run_sim3(const char* in, const char* out) {
  Reader r(in);
  int32_t h[2]; r.get(h, 2);                   // n, bFixScale
  double k[17]; r.get(k, 17);                  // K1 (4), K2 (4), S12 q (4) t (3) s (1), th2
  std::vector<double> p1c, p2c, obs1, obs2, s1, s2;
  r.get(p1c, 3 * (size_t)h[0]); r.get(p2c, 3 * (size_t)h[0]); r.get(obs1, 2 * (size_t)h[0]); r.get(obs2, 2 * (size_t)h[0]); r.get(s1, h[0]); r.get(s2, h[0]);
  lld_sim3_problem P{};
  P.fx1 = k[0]; P.fy1 = k[1]; P.cx1 = k[2]; P.cy1 = k[3]; P.fx2 = k[4]; P.fy2 = k[5]; P.cx2 = k[6]; P.cy2 = k[7];
  for (int i = 0; i < 4; i++) P.s12_q[i] = k[8 + i];
  for (int i = 0; i < 3; i++) P.s12_t[i] = k[12 + i];
  P.s12_s = k[15]; P.n = h[0];
  P.p1c = p1c.data(); P.p2c = p2c.data(); P.obs1 = obs1.data(); P.obs2 = obs2.data(); P.inv_sigma2_1 = s1.data(); P.inv_sigma2_2 = s2.data();
  lld_amd::Context ctx(0);
  std::vector<uint8_t> dropped;
  const int32_t nIn = lld_amd::Optimizer::OptimizeSim3(ctx, P, dropped, (float)k[16], h[1] != 0);
  Writer wr(out);
  const double S[8] = {P.s12_q[0], P.s12_q[1], P.s12_q[2], P.s12_q[3], P.s12_t[0], P.s12_t[1], P.s12_t[2], P.s12_s};
  wr.put(S, 8); wr.put(&nIn, 1); wr.put(dropped);
  std::printf("sim3: %d inliers of %d\n", nIn, h[0]);
  return 0;
}

}  // namespace

// The frame-rate path, driven the way a patched Tracking thread would drive it: the Frame's keypoints and lines are uploaded once
// (Frame constructor), then TrackWithMotionModel and TrackLocalMap are two calls that only queue work, and one download ends the frame.
// header[10] != 0: the host also fetches stage 1's record between the two calls (what the reference's UpdateLocalMap needs).
int run_track(const char* in, const char* out) {
  Reader r(in);
  int32_t h[16]; r.get(h, 16);      // nt n_levels n_last n_mp nl nr dim n_last_lines n_local_lines repeats download_between
  const int nt = h[0], n_levels = h[1], n_last = h[2], n_mp = h[3], nl = h[4], nr = h[5], dim = h[6], n_ll = h[7], n_ml = h[8], repeats = h[9];
  float fc[6]; r.get(fc, 6);        // min_x min_y max_x max_y grid_width_inv grid_height_inv
  std::vector<float> scale, inv_sigma2; r.get(scale, n_levels); r.get(inv_sigma2, n_levels);
  double dc[8]; r.get(dc, 8);       // fx fy cx cy bf gamma line_thr_base md_thr
  std::vector<uint32_t> t_desc; std::vector<float> t_xy, t_ur, t_ang; std::vector<int32_t> t_oct;
  r.get(t_desc, 8 * (size_t)nt); r.get(t_xy, 2 * (size_t)nt); r.get(t_oct, nt); r.get(t_ur, nt); r.get(t_ang, nt);
  lld_frame_view view; r.get(&view, 1);
  float Tcw[16]; r.get(Tcw, 16);
  std::vector<float> l_pos, l_ang; std::vector<uint8_t> l_valid, l_obs; std::vector<int32_t> l_oct, l_id; std::vector<uint32_t> l_desc;
  r.get(l_pos, 3 * (size_t)n_last); r.get(l_valid, n_last); r.get(l_oct, n_last); r.get(l_ang, n_last); r.get(l_desc, 8 * (size_t)n_last); r.get(l_obs, n_last); r.get(l_id, n_last);
  std::vector<float> m_pos, m_nrm, m_maxd, m_mind; std::vector<uint32_t> m_desc; std::vector<uint8_t> m_obs, m_skip; std::vector<int32_t> m_id;
  r.get(m_pos, 3 * (size_t)n_mp); r.get(m_nrm, 3 * (size_t)n_mp); r.get(m_maxd, n_mp); r.get(m_mind, n_mp); r.get(m_desc, 8 * (size_t)n_mp); r.get(m_obs, n_mp); r.get(m_skip, n_mp);
  r.get(m_id, n_mp);
  std::vector<float> ln_left, ln_right, ln_desc; std::vector<int32_t> ln_lo, ln_ro, ln_lm;
  r.get(ln_left, 4 * (size_t)nl); r.get(ln_lo, nl); r.get(ln_right, 4 * (size_t)nr); r.get(ln_ro, nr); r.get(ln_lm, nl); r.get(ln_desc, (size_t)nl * dim);
  struct LineSet { std::vector<double> x0, dir, x1, x2; std::vector<uint8_t> skip; std::vector<float> desc; std::vector<int32_t> id; lld_map_lines c; };
  auto read_lines = [&](LineSet& L, int n) {
    r.get(L.x0, 3 * (size_t)n); r.get(L.dir, 3 * (size_t)n); r.get(L.x1, 3 * (size_t)n); r.get(L.x2, 3 * (size_t)n); r.get(L.skip, n); r.get(L.desc, (size_t)n * dim); r.get(L.id, n);
    L.c = lld_map_lines{n, L.x0.data(), L.dir.data(), L.x1.data(), L.x2.data(), L.skip.data(), L.desc.data(), L.id.data()};
  };
  LineSet last_lines, local_lines; read_lines(last_lines, n_ll); read_lines(local_lines, n_ml);

  lld_orb_search kp{};
  kp.nt = nt; kp.t_desc = t_desc.data(); kp.t_xy = t_xy.data(); kp.t_octave = t_oct.data(); kp.t_uright = t_ur.data(); kp.t_angle = t_ang.data();
  kp.grid_min_x = fc[0]; kp.grid_min_y = fc[1]; kp.grid_width_inv = fc[4]; kp.grid_height_inv = fc[5]; kp.grid_cols = 64; kp.grid_rows = 48;
  kp.n_levels = n_levels; kp.level_scale = scale.data(); kp.level_inv_sigma2 = inv_sigma2.data();
  lld_frame_lines fl{};
  fl.n_left = nl; fl.left = ln_left.data(); fl.left_octave = ln_lo.data(); fl.n_right = nr; fl.right = ln_right.data(); fl.right_octave = ln_ro.data();
  fl.line_matches = ln_lm.data(); fl.desc = ln_desc.data(); fl.dim = dim; fl.sx = 1.0 / fc[2]; fl.sy = 1.0 / fc[3];
  lld_last_frame_points last{n_last, l_pos.data(), l_valid.data(), l_oct.data(), l_ang.data(), l_desc.data(), l_obs.data()};
  lld_map_points mp{n_mp, m_pos.data(), m_nrm.data(), m_maxd.data(), m_mind.data(), m_desc.data(), m_obs.data(), m_skip.data()};

  lld_amd::Context ctx(0);
  lld_amd::TrackedFrame frame(ctx, kp, nl > 0 ? &fl : nullptr);
  frame.params.cam = lld_camera{dc[0], dc[1], dc[2], dc[3], dc[4]};
  frame.params.pose.gamma = dc[5]; frame.params.line_thr_reproj_base = dc[6]; frame.params.line_md_thr = dc[7];
  lld_amd::TrackRecord r1, r2, r1_between;
  std::vector<double> ms_total(repeats), ms_queue1(repeats), ms_queue2(repeats);
  typedef std::chrono::steady_clock clk;
  for (int it = 0; it < repeats; it++) {
    const clk::time_point t0 = clk::now();
    frame.TrackWithMotionModel(view, Tcw, last, l_id.data(), n_ll > 0 ? &last_lines.c : nullptr);
    const clk::time_point t1 = clk::now();
    if (h[10]) frame.Download(&r1_between, nullptr);
    const clk::time_point t1b = clk::now();
    frame.TrackLocalMap(mp, m_id.data(), n_ml > 0 ? &local_lines.c : nullptr);
    const clk::time_point t2 = clk::now();
    frame.Download(&r1, &r2);
    const clk::time_point t3 = clk::now();
    ms_total[it] = std::chrono::duration<double, std::milli>(t3 - t0).count();
    ms_queue1[it] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    ms_queue2[it] = std::chrono::duration<double, std::milli>(t2 - t1b).count();
  }
  Writer w(out);
  for (const lld_amd::TrackRecord* rec : {&r1, &r2}) {
    w.put(rec->r.pose_qt, 7); w.put(&rec->r.chi2, 1);
    const int32_t c[12] = {rec->r.n_inliers, rec->r.lm_iterations, rec->r.lm_trials, rec->r.n_edges, rec->r.n_search_first, rec->r.n_search, rec->r.used_wide,
                           rec->r.n_points, rec->r.n_points_map, rec->r.n_lines_matched, rec->r.n_lines, rec->r.n_discarded};
    w.put(c, 12);
    w.put(rec->kp_point_id); w.put(rec->kp_outlier); w.put(rec->ln_line_id); w.put(rec->ln_outlier);
  }
  w.put(ms_total); w.put(ms_queue1); w.put(ms_queue2);
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 4) { std::fprintf(stderr, "usage: harness ba|gba|pose|orb|sim3|lines|init|loop|track <in> <out>\n"); return 2; }
  try {
    if (!std::strcmp(argv[1], "ba")) return run_ba(argv[2], argv[3], false);
    if (!std::strcmp(argv[1], "gba")) return run_ba(argv[2], argv[3], true);
    if (!std::strcmp(argv[1], "sim3")) return run_sim3(argv[2], argv[3]);
    if (!std::strcmp(argv[1], "pose")) return run_pose(argv[2], argv[3]);
    if (!std::strcmp(argv[1], "orb")) return run_orb(argv[2], argv[3]);
    if (!std::strcmp(argv[1], "lines")) return run_lines(argv[2], argv[3]);
    if (!std::strcmp(argv[1], "init")) return run_init(argv[2], argv[3]);
    if (!std::strcmp(argv[1], "loop")) return run_loop(argv[2], argv[3]);
    if (!std::strcmp(argv[1], "track")) return run_track(argv[2], argv[3]);
    std::fprintf(stderr, "unknown mode %s\n", argv[1]);
    return 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "harness: %s\n", e.what());   // e.g. "lld_ctx_create: no HIP device (this library has no CPU fallback)"
    return 1;
  }
}
