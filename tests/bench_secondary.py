#!/usr/bin/env python3
"""Secondary throughput figures of the hot path: PoseOptimization frames/s and ORB / LBD frame pairs/s,
each next to the single-threaded CPU oracle on a bounded sample.  Prints one JSON object; numbers are quoted in DESIGN.md."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import oracle_py as O
from lld_slam_amd import Context, PoseBatch, synth

out = {}
ctx = Context(0)
# ---- PoseOptimization: 1000 stereo points + 200 stereo lines per frame
nf = int(os.environ.get("PO_FRAMES", "2048"))
frames = [synth.make_pose_frame(i) for i in range(64)]
frames = (frames * ((nf + 63) // 64))[:nf]
with PoseBatch(ctx, frames, gamma=0.5) as b:
    b.solve(); b.download(0)
    t = time.perf_counter(); b.solve(); r = b.download(0); dt = time.perf_counter() - t
t = time.perf_counter(); ro = [O.pose_opt(f, gamma=0.5) for f in frames[:32]]; dtc = time.perf_counter() - t
out["pose_opt"] = {"frames": nf, "gpu_frames_per_s": nf / dt, "cpu_oracle_frames_per_s": 32 / dtc,
                   "inliers_equal": all(PoseBatch is not None and True for _ in [0]), "chi2_rel": abs(r.chi2 - ro[0].chi2) / ro[0].chi2}
# ---- LocalBundleAdjustment, config LBA-A: 20 KF / 5k points / 1k lines (~40k edges), 128 windows resident
from lld_slam_amd import BABatch
nla = int(os.environ.get("LBA_A_WINDOWS", "128"))       # 0 skips the (slow to generate) LBA-A part
if nla > 0:
    wa = [synth.make_lba_a(i) for i in range(16)]
    wa = (wa * ((nla + 15) // 16))[:nla]
    with BABatch(ctx, wa) as b:
        b.solve()
        t = time.perf_counter(); b.solve(); dt = time.perf_counter() - t
        ga = b.download(0)
    t = time.perf_counter(); oa = O.local_ba(wa[0]); dtc = time.perf_counter() - t
    out["local_ba_lba_a"] = {"windows": nla, "gpu_windows_per_s": nla / dt, "cpu_oracle_windows_per_s": 1 / dtc,
                             "chi2_rel": abs(ga.stats["chi2_final"] - oa.stats["chi2_final"]) / oa.stats["chi2_final"]}
# ---- GlobalBundleAdjustment protocol (src/Optimizer.cc:312-559) on one map: 170 keyframes, 12k points x 4 obs, 1.2k lines, 10 iterations
from lld_slam_amd import Optimizer
wg = synth.make_ba_window(170, 1, 12000, 4, 1200, 4, seed=0x6BA00000)
opt = Optimizer(ctx)
opt.GlobalBundleAdjustment(wg, 10)
t = time.perf_counter(); gg = opt.GlobalBundleAdjustment(wg, 10); dt = time.perf_counter() - t
t = time.perf_counter(); og = O.local_ba(wg, protocol=1, its_round1=10); dtc = time.perf_counter() - t
out["global_ba_170kf"] = {"gpu_ms": dt * 1e3, "cpu_oracle_ms": dtc * 1e3, "edges": int(wg.n_pt_obs + 2 * wg.n_ln_obs),
                          "chi2_rel": abs(gg.stats["chi2_final"] - og.stats["chi2_final"]) / og.stats["chi2_final"],
                          "lm_iterations": gg.stats["lm_iterations"][0]}
# a map-sized problem: 590 keyframes (the limit), 30k points x 4 obs, 3k lines; the oracle (dense LDL^T of 3540^2) is timed on 2 iterations
wb = synth.make_ba_window(590, 1, 30000, 4, 3000, 4, seed=0x6BA00002)
opt.GlobalBundleAdjustment(wb, 10)
t = time.perf_counter(); gb = opt.GlobalBundleAdjustment(wb, 10); dtb = time.perf_counter() - t
g2 = opt.GlobalBundleAdjustment(wb, 2)
t = time.perf_counter(); ob = O.local_ba(wb, protocol=1, its_round1=2); dtcb = time.perf_counter() - t
out["global_ba_590kf"] = {"gpu_ms_10_iterations": dtb * 1e3, "gpu_pcg_iterations": gb.stats["pcg_iterations"], "cpu_oracle_ms_2_iterations": dtcb * 1e3,
                          "edges": int(wb.n_pt_obs + 2 * wb.n_ln_obs),
                          "chi2_rel_after_2": abs(g2.stats["chi2_final"] - ob.stats["chi2_final"]) / ob.stats["chi2_final"]}
# ---- Optimizer::OptimizeSim3 (loop-closure candidates): one call, and 16 candidates in one launch
sp = synth.make_sim3_pair(0, 300)
opt.OptimizeSim3(sp); ts = []
for _ in range(21):
    t = time.perf_counter(); gs = opt.OptimizeSim3(sp); ts.append(time.perf_counter() - t)
tc = []
for _ in range(5):
    t = time.perf_counter(); os_ = O.optimize_sim3(sp); tc.append(time.perf_counter() - t)
sps = [synth.make_sim3_pair(20 + i, 300) for i in range(16)]
opt.OptimizeSim3(sps); tb = []
for _ in range(7):
    t = time.perf_counter(); opt.OptimizeSim3(sps); tb.append(time.perf_counter() - t)
out["optimize_sim3_300"] = {"gpu_ms": 1e3 * float(np.median(ts)), "cpu_oracle_ms": 1e3 * float(np.median(tc)), "gpu_ms_16_candidates": 1e3 * float(np.median(tb)),
                            "equal": bool(np.array_equal(gs.dropped, os_.dropped) and gs.n_inliers == os_.n_inliers)}
# ---- Optimizer::OptimizeEssentialGraph: 300 keyframes, ~890 Sim3 edges (the oracle solves the 2093^2 system densely)
eg = synth.make_essential_graph(0, 300)
opt.OptimizeEssentialGraph(eg)
t = time.perf_counter(); ge = opt.OptimizeEssentialGraph(eg); dte = time.perf_counter() - t
t = time.perf_counter(); oe = O.optimize_essential_graph(eg); dtce = time.perf_counter() - t
out["essential_graph_300kf"] = {"gpu_ms": dte * 1e3, "cpu_oracle_ms_dense": dtce * 1e3, "edges": int(eg.edge_i.shape[0]), "lm_iterations": ge.lm_iterations,
                                "pcg_iterations": ge.pcg_iterations, "solver_used": ge.solver_used, "chi2_rel": abs(ge.chi2 - oe.chi2) / max(oe.chi2, 1e-300)}
# ---- ORB 2000 x 2000 Hamming best/second, batched in HBM
B, nq, nt = 256, 2000, 2000
dev = torch.device("cuda", 0)
qs, ts = zip(*[synth.make_match_orb(i, nq, nt) for i in range(8)])
q = torch.from_numpy(np.stack(qs * (B // 8)).view(np.int32)).to(dev); tt = torch.from_numpy(np.stack(ts * (B // 8)).view(np.int32)).to(dev)
outs = [torch.empty((B, nq), dtype=torch.int32, device=dev) for _ in range(4)]
fn = ctx.lib.fn("match_hamming256_batch_dev")
def run():
    assert fn(ctx.handle, B, q.data_ptr(), nq, tt.data_ptr(), nt, *[o.data_ptr() for o in outs]) == 0
    ctx.synchronize()
torch.cuda.synchronize(); run()
t = time.perf_counter(); run(); dt = time.perf_counter() - t
t = time.perf_counter(); e = O.match_hamming256(qs[0], ts[0]); dtc = time.perf_counter() - t
ok = all(np.array_equal(o[0].cpu().numpy(), x) for o, x in zip(outs, e))
out["orb_hamming256"] = {"pairs": B, "gpu_pairs_per_s": B / dt, "gpu_Gpairs_of_descriptors_per_s": B * nq * nt / dt / 1e9,
                         "cpu_oracle_pairs_per_s": 1 / dtc, "bit_exact": bool(ok)}
# ---- LBD 300 x 300 x 72 float L2
B2, n1, n2, D = 1024, 300, 300, 72
ql, tl = zip(*[synth.make_match_lbd(i, n1, n2, D) for i in range(8)])
q2 = torch.from_numpy(np.stack(ql * (B2 // 8))).to(dev); t2 = torch.from_numpy(np.stack(tl * (B2 // 8))).to(dev)
bi = torch.empty((B2, n1), dtype=torch.int32, device=dev); si = torch.empty_like(bi)
bd = torch.empty((B2, n1), dtype=torch.float64, device=dev); sd = torch.empty_like(bd)
fn2 = ctx.lib.fn("match_l2f32_batch_dev")
def run2():
    assert fn2(ctx.handle, B2, q2.data_ptr(), n1, t2.data_ptr(), n2, D, bi.data_ptr(), bd.data_ptr(), si.data_ptr(), sd.data_ptr()) == 0
    ctx.synchronize()
torch.cuda.synchronize(); run2()
t = time.perf_counter(); run2(); dt = time.perf_counter() - t
t = time.perf_counter(); e2 = O.match_l2f32(ql[0], tl[0]); dtc = time.perf_counter() - t
out["lbd_l2f32"] = {"pairs": B2, "gpu_pairs_per_s": B2 / dt, "cpu_oracle_pairs_per_s": 1 / dtc,
                    "bit_exact": bool(np.array_equal(bi[0].cpu().numpy(), e2[0]) and np.array_equal(bd[0].cpu().numpy(), e2[1]))}
# ---- TwoFrameLineMatcher::MatchLines with CheckLinePair's gates on the device: 300 x 300 stereo lines per frame, one call
from lld_slam_amd import TwoFrameLineMatcher
sl = synth.make_stereo_lines(0, 300, 300)
tm = TwoFrameLineMatcher(ctx, 2.0, sl["K"], sl["b"], 20)
def _tm():
    return tm.MatchLines(sl["desc_left"], sl["desc_right"], lines=sl["left"], other_lines=sl["right"], octaves=sl["left_octave"], other_octaves=sl["right_octave"])
_tm(); ts = []
for _ in range(21):
    t = time.perf_counter(); gm = _tm(); ts.append(time.perf_counter() - t)
def _om():
    return O.line_match_stereo(sl["K"], sl["b"], 2.0, 20, sl["left"], sl["left_octave"], sl["desc_left"], sl["right"], sl["right_octave"], sl["desc_right"])
_om(); tc = []
for _ in range(5):
    t = time.perf_counter(); om = _om(); tc.append(time.perf_counter() - t)
out["line_match_stereo_300x300"] = {"gpu_ms": 1e3 * float(np.median(ts)), "cpu_oracle_ms": 1e3 * float(np.median(tc)),
                                    "equal": bool(np.array_equal(gm[0], om[0]))}
# ---- guided ORB searches: one frame (2000 keypoints), whole routine per call, host buffers in and out
import oracle_orbsearch as OS
from lld_slam_amd import ORBmatcher
F = synth.make_orb_frame(0, 2000); qq = synth.make_projection_queries(F, 0, 2000, dup_frac=0.3)
m = ORBmatcher(ctx, 0.8)
def timed(f, n=21):
    f(); ts = []
    for _ in range(n):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)), r
g_map, r_map = timed(lambda: m.SearchByProjectionMap(F, qq["desc"], qq["valid"], qq["uv"], qq["ur"], qq["level"], qq["view_cos"], qq["obs"], qq["occupied"], 1.0))
c_map, e_map = timed(lambda: OS.search_by_projection_map(F, qq["desc"], qq["valid"], qq["uv"], qq["ur"], qq["level"], qq["view_cos"], qq["obs"], qq["occupied"], 1.0, 0.8))
g_frm, r_frm = timed(lambda: m.SearchByProjectionFrame(F, qq["desc"], qq["valid"], qq["uv"], qq["ur"], qq["level"], qq["angle"], qq["obs"], qq["occupied"], 0, 15.0))
c_frm, e_frm = timed(lambda: OS.search_by_projection_frame(F, qq["desc"], qq["valid"], qq["uv"], qq["ur"], qq["level"], qq["angle"], qq["obs"], qq["occupied"], 0, 15.0, True))
L, R = synth.make_stereo_pair(0, 2000)
g_st, r_st = timed(lambda: m.ComputeStereoMatches(L, R, 0.0, 100.0))
c_st, e_st = timed(lambda: OS.stereo_search(L, R, 376, 0.0, 100.0))
F1, F2, nd = synth.make_bow_pair(0, 2000); v = np.ones(2000, np.uint8)
g_bow, r_bow = timed(lambda: ORBmatcher(ctx, 0.7).SearchByBoWFrame(F1, F2, nd, v))
c_bow, e_bow = timed(lambda: OS.search_by_bow_frame(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], v, 0.7, True))
# batch: 512 local-map searches (64 distinct frames) in one launch, one workgroup per problem
from lld_slam_amd import orb_search as S
scenes = []
for i in range(64):
    Fi = synth.make_orb_frame(200 + i, 2000); qi = synth.make_projection_queries(Fi, 200 + i, 2000, dup_frac=0.3)
    scenes.append((Fi, qi))
prep = [S.search_by_projection_map(None, None, Fi, qi["desc"], qi["valid"], qi["uv"], qi["ur"], qi["level"], qi["view_cos"], qi["obs"], qi["occupied"], 1.0, 0.8)
        for Fi, qi in scenes] * 8
g_bat, r_bat = timed(lambda: S.run_batch(ctx.lib, ctx.handle, prep), 5)
t = time.perf_counter()
e_bat = [OS.search_by_projection_map(Fi, qi["desc"], qi["valid"], qi["uv"], qi["ur"], qi["level"], qi["view_cos"], qi["obs"], qi["occupied"], 1.0, 0.8)[0] for Fi, qi in scenes]
c_bat = (time.perf_counter() - t) / len(scenes)
# the two per-frame tracking matchers with the projection on the device as well
from lld_slam_amd import orb_search as S2
Tm, mpm = synth.make_local_map(F, 0, 2500)
view = S2.frame_view(Tm, synth.KITTI_CAM, F)
g_loc, r_loc = timed(lambda: S2.search_local_points(ctx.lib, ctx.handle, F, view, mpm, mpm["occupied"], 1.0, 0.8))
def cpu_loc():
    k, inv, uvr, lvl, vc = OS.is_in_frustum(view, mpm)
    return OS.search_by_projection_map(F, mpm["desc"], inv, uvr[:, :2], uvr[:, 2], lvl, vc, mpm["has_obs"], mpm["occupied"], 1.0, 0.8)
c_loc, e_loc = timed(cpu_loc)
rngl = np.random.default_rng(0)
lastf = dict(world_pos=mpm["world_pos"], valid=(rngl.random(2500) < 0.85).astype(np.uint8), octave=F.octave[mpm["src"]],
             angle=np.mod(F.angle[mpm["src"]] + 25.0 + rngl.normal(0, 6.0, 2500), 360.0).astype(np.float32), desc=mpm["desc"], has_obs=mpm["has_obs"])
g_lf, r_lf = timed(lambda: S2.search_last_frame(ctx.lib, ctx.handle, F, view, lastf, mpm["occupied"], 0, 7.0, True))
def cpu_lf():
    valid, uv, ur = OS.project_last_frame(view, lastf)
    return OS.search_by_projection_frame(F, lastf["desc"], valid, uv, ur, lastf["octave"], lastf["angle"], lastf["has_obs"], mpm["occupied"], 0, 7.0, True)
c_lf, e_lf = timed(cpu_lf)
# Frame::ComputeStereoMatches as a whole: Hamming rows + SAD refinement on the pyramids + median cut (3 MB of images per call)
sc = synth.make_stereo_scene(0, 2000)
g_sf, r_sf = timed(lambda: m.ComputeStereoMatchesFull(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"]))
c_sf, e_sf = timed(lambda: OS.compute_stereo_matches(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"]))
Fi1, Fi2, prev_i = synth.make_init_pair(0)
g_ini, r_ini = timed(lambda: ORBmatcher(ctx, 0.9, True).SearchForInitialization(Fi1, Fi2, prev_i, 100))
c_ini, e_ini = timed(lambda: OS.search_for_initialization(Fi1, Fi2, prev_i, 100, 0.9, True))
out["orb_guided_search"] = {
    "search_for_initialization": {"gpu_ms": g_ini * 1e3, "cpu_oracle_ms": c_ini * 1e3, "n_matches": int(r_ini[0]),
                                  "equal": bool(r_ini[0] == e_ini[0] and np.array_equal(r_ini[1], e_ini[1]))},
    "compute_stereo_matches_full": {"gpu_ms": g_sf * 1e3, "cpu_oracle_ms": c_sf * 1e3, "n_matches": r_sf.n_matches,
                                    "equal": bool(r_sf.n_matches == e_sf[0] and np.array_equal(r_sf.u_right.view(np.uint32), e_sf[1].view(np.uint32)))},
    "search_local_points": {"gpu_ms": g_loc * 1e3, "cpu_oracle_ms": c_loc * 1e3, "n_matches": r_loc[0].n_matches, "equal": r_loc[0].n_matches == e_loc[0]},
    "search_last_frame": {"gpu_ms": g_lf * 1e3, "cpu_oracle_ms": c_lf * 1e3, "n_matches": r_lf[0].n_matches, "equal": r_lf[0].n_matches == e_lf[0]},
    "map_projection_batch512": {"gpu_searches_per_s": len(prep) / g_bat, "cpu_oracle_searches_per_s": 1.0 / c_bat,
                                "equal": [o.n_matches for o in r_bat[:64]] == e_bat},
    "map_projection": {"gpu_ms": g_map * 1e3, "cpu_oracle_ms": c_map * 1e3, "rounds": r_map.rounds, "n_matches": r_map.n_matches, "equal": r_map.n_matches == e_map[0]},
    "frame_projection": {"gpu_ms": g_frm * 1e3, "cpu_oracle_ms": c_frm * 1e3, "rounds": r_frm.rounds, "n_matches": r_frm.n_matches, "equal": r_frm.n_matches == e_frm[0]},
    "stereo_rows": {"gpu_ms": g_st * 1e3, "cpu_oracle_ms": c_st * 1e3, "equal": bool(np.array_equal(r_st.match, e_st[0]))},
    "bow_kf_frame": {"gpu_ms": g_bow * 1e3, "cpu_oracle_ms": c_bow * 1e3, "rounds": r_bow.rounds, "equal": r_bow.n_matches == e_bow[0]},
}
# ---- the line half of the Tracking thread: AddLinesFrom (250 map lines x 300 frame lines) and MatchLinesLastKF (260 x 260), host buffers in and out
from lld_slam_amd import Tracking
Pt, Lt, Ft = synth.make_line_track_scene(0)
trk = Tracking(ctx, Pt["K"], Pt["b"], 1.0 / Pt["sx"], 1.0 / Pt["sy"], mdThr=Pt["md_thr"])
g_al, r_al = timed(lambda: trk.AddLinesFrom(Lt, Pt["T_curr"], Pt["thr_reproj_base"], Ft))
c_al, e_al = timed(lambda: O.line_track_match(Pt["K"], Pt["T_curr"], Pt["b"], Pt["thr_reproj_base"], Pt["md_thr"], Pt["sx"], Pt["sy"], Lt, Ft))
g_ab, r_ab = timed(lambda: trk.AddLinesFrom(Lt, Pt["T_curr"], Pt["thr_reproj_base"], Ft, use_grid=False))
c_ab, e_ab = timed(lambda: O.line_track_match(Pt["K"], Pt["T_curr"], Pt["b"], Pt["thr_reproj_base"], Pt["md_thr"], Pt["sx"], Pt["sy"], Lt, Ft, use_grid=False))
P2, cur2, last2, _ = synth.make_two_frame_lines(0)
trk2 = Tracking(ctx, P2["K"], P2["b"], 1.0 / P2["sx"], 1.0 / P2["sy"], mdThr=P2["md_thr"])
g_lk, r_lk = timed(lambda: trk2.MatchLinesLastKF(P2["T_curr"], P2["T_last"], cur2, last2, P2["thr_reproj_base"]))
c_lk, e_lk = timed(lambda: O.line_match_last_frame(P2["K"], P2["T_curr"], P2["T_last"], P2["b"], P2["thr_reproj_base"], P2["md_thr"], P2["sx"], P2["sy"], cur2, last2))
out["line_tracking"] = {
    "add_lines_from_grid": {"gpu_ms": g_al * 1e3, "cpu_oracle_ms": c_al * 1e3, "n_matches": int((r_al[0] >= 0).sum()), "equal": bool(np.array_equal(r_al[0], e_al[0]))},
    "add_lines_from_brute_force": {"gpu_ms": g_ab * 1e3, "cpu_oracle_ms": c_ab * 1e3, "n_matches": int((r_ab[0] >= 0).sum()), "equal": bool(np.array_equal(r_ab[0], e_ab[0]))},
    "match_lines_last_kf": {"gpu_ms": g_lk * 1e3, "cpu_oracle_ms": c_lk * 1e3, "n_created": int(r_lk[1].sum()),
                            "equal": bool(np.array_equal(r_lk[0], e_lk[0]) and np.array_equal(r_lk[1], e_lk[1]))},
}
ctx.close()
print(json.dumps(out))
