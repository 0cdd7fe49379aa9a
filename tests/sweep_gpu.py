"""Randomised GPU-vs-oracle sweep over every optimiser / matcher entry point (more seeds and sizes than the pytest cases; run by hand on the
GPU box: python tests/sweep_gpu.py).  Prints the number of mismatches per family; rounds 1 and 2: 0 everywhere."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import oracle_py as O, oracle_orbsearch as OS
from lld_slam_amd import Context, Optimizer, ORBmatcher, PoseBatch, synth
ctx = Context(0); opt = Optimizer(ctx)
rng = np.random.default_rng(1)
bad = 0
frames = [synth.make_pose_frame(1000 + i, n_points=int(rng.integers(3, 1200)), n_lines=int(rng.integers(0, 250)), outlier_frac=float(rng.uniform(0, 0.4)),
                                mono_frac=float(rng.choice([0, 0, 0.2])), mono_line_frac=float(rng.choice([0, 0.3]))) for i in range(150)]
with PoseBatch(ctx, frames, gamma=0.5) as b:
    b.solve()
    for i, f in enumerate(frames):
        g = b.download(i); o = O.pose_opt(f, gamma=0.5)
        ok = g.n_inliers == o.n_inliers and np.array_equal(g.pt_outlier, o.pt_outlier) and np.array_equal(g.ln_outlier, o.ln_outlier) and np.allclose(g.pose_qt, o.pose_qt, rtol=1e-5, atol=1e-7)
        if not ok: bad += 1; print("pose mismatch", i, g.n_inliers, o.n_inliers, np.abs(g.pose_qt - o.pose_qt).max())
print("pose frames checked", len(frames), "mismatches", bad)
bad = 0
pairs = [synth.make_sim3_pair(2000 + i, int(rng.integers(12, 600)), outlier_frac=float(rng.uniform(0, 0.4))) for i in range(80)]
gs = opt.OptimizeSim3(pairs)
for i, (g, p) in enumerate(zip(gs, pairs)):
    o = O.optimize_sim3(p)
    ok = g.n_inliers == o.n_inliers and np.array_equal(g.dropped, o.dropped) and np.allclose(g.s12_t, o.s12_t, rtol=1e-5, atol=1e-6)
    if not ok: bad += 1; print("sim3 mismatch", i, g.n_inliers, o.n_inliers, (g.dropped != o.dropped).sum())
print("sim3 pairs checked", len(pairs), "mismatches", bad)
bad = 0
m = ORBmatcher(ctx)
for i in range(12):
    sc = synth.make_stereo_scene(100 + i, int(rng.integers(200, 3000)))
    g = m.ComputeStereoMatchesFull(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    n, ur, dep, br, sad = OS.compute_stereo_matches(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    ok = g.n_matches == n and np.array_equal(g.u_right.view(np.uint32), ur.view(np.uint32)) and np.array_equal(g.sad, sad) and np.array_equal(g.best_r, br)
    if not ok: bad += 1; print("stereo mismatch", i)
print("stereo scenes checked 12 mismatches", bad)
from lld_slam_amd import BABatch
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_ba import check_ba
from test_gpu_orbsearch import expect_slots
rng = np.random.default_rng(7)
ws = [synth.make_lba_small(500 + i, n_free=int(rng.integers(1, 40)), n_fixed=int(rng.integers(6, 9)), n_points=int(rng.integers(20, 900)), n_lines=int(rng.integers(0, 150)),
                           mono_frac=float(rng.choice([0, 0.2])), mono_line_frac=float(rng.choice([0, 0.3])), outlier_frac=float(rng.uniform(0, 0.2))) for i in range(48)]
bad = 0
with BABatch(ctx, ws) as b:
    b.solve()
    for i, w in enumerate(ws):
        try: check_ba(b.download(i), O.local_ba(w), w)
        except AssertionError as e: bad += 1; print("ba mismatch", i, w.n_free_cams, w.n_points, str(e)[:200])
print("ba windows checked", len(ws), "mismatches", bad)
bad = 0
for i in range(4):
    w = ws[i * 5]
    try: check_ba(opt.GlobalBundleAdjustment(w, 7), O.local_ba(w, protocol=1, its_round1=7), w)
    except AssertionError as e: bad += 1; print("gba mismatch", i, str(e)[:200])
print("gba checked 4 mismatches", bad)
# essential graphs, dense tile Cholesky and PCG vs the oracle's dense LDL^T (fixed scale: the two sides stop together)
from test_gpu_posegraph import _check as check_pg
bad = 0; n_pg = 0
for i in range(8):
    g = synth.make_essential_graph(700 + i, int(rng.integers(12, 260)))
    o = O.optimize_essential_graph(g, bFixScale=True)
    for solver in (1, 2):
        n_pg += 1
        try: check_pg(opt.OptimizeEssentialGraph(g, bFixScale=True, solver=solver), o, 2e-5)
        except AssertionError as e: bad += 1; print("essential graph mismatch", i, g.sim3.shape[0], solver, str(e)[:200])
print("essential graphs checked", n_pg, "mismatches", bad)
bad = 0
m = ORBmatcher(ctx, 0.8)
for seed in range(100, 125):
    F = synth.make_orb_frame(seed, int(rng.integers(300, 4000)))
    q = synth.make_projection_queries(F, seed, int(rng.integers(200, 3500)), dup_frac=float(rng.uniform(0, 0.5)))
    out = m.SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0)
    n_exp, slot = OS.search_by_projection_map(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0, 0.8)
    if out.n_matches != n_exp or not np.array_equal(expect_slots(out, q["occupied"]), slot): bad += 1; print("map mismatch", seed)
    out = ORBmatcher(ctx, 0.9, True).SearchByProjectionFrame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], 0, 15.0)
    n_exp, slot = OS.search_by_projection_frame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], 0, 15.0, True)
    if out.n_matches != n_exp or not np.array_equal(expect_slots(out, q["occupied"]), slot): bad += 1; print("frame mismatch", seed)
print("orb searches checked 50 mismatches", bad)
bad = 0
for pid in range(40, 60):
    F1, F2, prev = synth.make_init_pair(pid, n=int(rng.integers(200, 3500)), rival_frac=float(rng.uniform(0, 0.4)))
    win, nn, ori = int(rng.integers(5, 160)), float(rng.uniform(0.6, 0.95)), bool(rng.integers(0, 2))
    on, om, opm = OS.search_for_initialization(F1, F2, prev, win, nn, ori)
    gn, gm, gpm = ORBmatcher(ctx, nn, ori).SearchForInitialization(F1, F2, prev, win)
    if gn != on or not np.array_equal(gm, om) or not np.array_equal(gpm, opm): bad += 1; print("initialisation search mismatch", pid, win, nn, ori)
print("initialisation searches checked 20 mismatches", bad)
bad = 0
from lld_slam_amd import Tracking
for seed in range(200, 240):
    P, L, F = synth.make_line_track_scene(seed, n_map=int(rng.integers(1, 400)), n_cur=int(rng.integers(1, 500)), related_frac=float(rng.uniform(0.2, 0.9)),
                                          occupied_frac=float(rng.uniform(0, 0.3)), no_partner_frac=float(rng.uniform(0, 0.4)))
    mono, grid = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    trk = Tracking(ctx, P["K"], P["b"], 1.0 / P["sx"], 1.0 / P["sy"], mdThr=P["md_thr"], monocular=mono)
    g = trk.AddLinesFrom(L, P["T_curr"], P["thr_reproj_base"], F, use_grid=grid)
    o = O.line_track_match(P["K"], P["T_curr"], P["b"], P["thr_reproj_base"], P["md_thr"], P["sx"], P["sy"], L, F, monocular=mono, use_grid=grid)
    if not (np.array_equal(g[0], o[0]) and np.array_equal(g[1][g[0] >= 0], o[1][o[0] >= 0])): bad += 1; print("AddLinesFrom mismatch", seed)
    P2, cur, last, _ = synth.make_two_frame_lines(seed, n_lines=int(rng.integers(8, 400)), shared_frac=float(rng.uniform(0.2, 0.9)), baseline=float(rng.uniform(0.5, 3.0)))
    trk2 = Tracking(ctx, P2["K"], P2["b"], 1.0 / P2["sx"], 1.0 / P2["sy"], mdThr=P2["md_thr"])
    gm, gc, gx, gd = trk2.MatchLinesLastKF(P2["T_curr"], P2["T_last"], cur, last, P2["thr_reproj_base"], grid)
    om, oc, ox, od = O.line_match_last_frame(P2["K"], P2["T_curr"], P2["T_last"], P2["b"], P2["thr_reproj_base"], P2["md_thr"], P2["sx"], P2["sy"], cur, last, grid)
    okc = oc.astype(bool)
    if not (np.array_equal(gm, om) and np.array_equal(gc, oc) and np.allclose(gx[okc], ox[okc], rtol=1e-6, atol=1e-6) and np.allclose(gd[okc], od[okc], atol=1e-7)):
        bad += 1; print("MatchLinesLastKF mismatch", seed)
print("line tracking scenes checked 80 mismatches", bad)
