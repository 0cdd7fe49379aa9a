"""Random scenes for every matcher entry point with integer / index outputs that tools/fuzz_orb_projected.py does not cover: brute-force
Hamming / L2 (masks, candidate lists), the guided ORB searches with the projection on the host or on the device (local map, last frame,
Fuse), both SearchByBoW, SearchForTriangulation, SearchForInitialization, Frame::ComputeStereoMatches with images, the line matchers
(greedy, stereo gates, AddLinesFrom, MatchLinesLastKF).  Random sizes (around the workgroup / tile / LDS boundaries too), thresholds,
ratios, occupancy - device vs the oracle's sequential restatements, every integer output bit for bit (floats of the stereo routine by
their bit patterns).      python tools/fuzz_matchers.py [n=400] [seed=0] [only=routine]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from lld_slam_amd import Context, ORBmatcher, TwoFrameLineMatcher, Tracking, orb_search as S, synth
import oracle_orbsearch as OS
import oracle_py as O

f32 = np.float32


def eq(*pairs):
    return all(np.array_equal(a, b) for a, b in pairs)


def expect_slots(out, occupied, token=1 << 20):
    slot = np.where(np.asarray(occupied) != 0, token, -1).astype(np.int32)
    slot = np.where(out.owner >= 0, out.owner, slot)
    return np.where(out.owner == -2, -1, slot).astype(np.int32)


QUANT = os.environ.get("FUZZ_QUANT") == "1"   # FUZZ_QUANT=1: keypoint and projection coordinates on a half-pixel lattice, so that |dx| == r, ceil / floor of an
                                              # integer and round(x.5) - the strict / non-strict comparisons and rounding modes of the reference - actually occur


def quant(a):
    return (np.round(np.asarray(a, np.float64) * 2.0) / 2.0).astype(np.float32)


def quant_frame(F):
    if QUANT:
        F.xy = quant(F.xy); F.uright = np.where(F.uright > 0, quant(F.uright), F.uright).astype(np.float32)
    return F


BIG = os.environ.get("FUZZ_BIG") == "1"       # FUZZ_BIG=1: most sizes in the top tenth of their range, three times as many queries / MapPoints


def size(rng, hi, lo=1):
    """Mostly uniform, sometimes right at a power of two or one off it."""
    if BIG and rng.random() < 0.7:
        return int(rng.integers(max(lo, int(0.9 * hi)), hi + 1))
    if rng.random() < 0.25:
        p = 1 << int(rng.integers(0, max(1, int(np.log2(hi))) + 1))
        return int(np.clip(p + int(rng.integers(-1, 2)), lo, hi))
    return int(rng.integers(lo, hi + 1))


def r_hamming(ctx, rng, sid):
    nq, nt = size(rng, 2500), size(rng, 2500)
    q, t = synth.make_match_orb(sid, nq, nt, n_corr=int(rng.uniform(0, 1) * min(nq, nt)), n_dup=int(rng.integers(0, 20)))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        return eq(*zip(ORBmatcher(ctx).BestTwo(q, t), O.match_hamming256(q, t))), nq
    if kind == 1:
        mask = (rng.random((nq, nt)) < rng.choice([0.01, 0.1, 0.6])).astype(np.uint8)
        return eq(*zip(ORBmatcher(ctx).BestTwo(q, t, mask), O.match_hamming256(q, t, mask))), nq
    lens = rng.integers(0, int(rng.choice([3, 40, 200])), nq)
    cs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32); ci = rng.integers(0, nt, cs[-1]).astype(np.int32)
    return eq(*zip(ORBmatcher(ctx).BestTwoCandidates(q, t, cs, ci), O.match_hamming256_csr(q, t, cs, ci))), nq


def r_l2(ctx, rng, sid):
    nq, nt, dim = size(rng, 600), size(rng, 600), int(rng.choice([72, 72, 32, size(rng, 128)]))
    if rng.random() < 0.5:
        q, t = synth.make_match_lbd(sid, nq, nt, dim, n_corr=int(rng.uniform(0, 1) * min(nq, nt)))
    else:
        q = rng.normal(size=(nq, dim)).astype(f32); t = rng.normal(size=(nt, dim)).astype(f32)
        if nt > 2: t[nt // 2] = t[0]
    mask = None if rng.random() < 0.5 else (rng.random((nq, nt)) < rng.choice([0.05, 0.5])).astype(np.uint8)
    g = TwoFrameLineMatcher(ctx, 2.0).BestTwo(q, t, mask); o = O.match_l2f32(q, t, mask)
    ok = eq((g[0], o[0]), (g[1], o[1]))
    if nt > 1 and mask is None: ok = ok and eq((g[2], o[2]), (g[3], o[3]))
    elif mask is not None: ok = ok and eq((g[2], o[2]))
    return ok, nq


def r_line_greedy(ctx, rng, sid):
    nq, nt, dim = size(rng, 400), size(rng, 400), int(rng.choice([72, 32]))
    q, t = synth.make_match_lbd(sid, nq, nt, dim, n_corr=int(rng.uniform(0, 1) * min(nq, nt)))
    if rng.random() < 0.5 and nq > 8:                                   # groups of identical left lines: pure order dependence
        a = int(rng.integers(0, nq - 4)); q[a:a + int(rng.integers(2, 30))] = q[a]
    if rng.random() < 0.3 and nt > 4: t[int(rng.integers(0, nt))] = t[int(rng.integers(0, nt))]
    gate = None if rng.random() < 0.3 else (rng.random((nq, nt)) < rng.choice([0.05, 0.6, 1.0])).astype(np.uint8)
    tau = float(rng.choice([2.0, 0.5, 1.6, 1e9]))
    gm, gd = TwoFrameLineMatcher(ctx, tau).MatchLines(q, t, gate); om, od = O.line_match_greedy(q, t, gate, tau)
    return eq((gm, om), (gd[gm >= 0], od[om >= 0])), int((gm >= 0).sum())


def r_line_stereo(ctx, rng, sid):
    nl, nr = size(rng, 400), size(rng, 400)
    s = synth.make_stereo_lines(sid, nl, nr, related_frac=float(rng.uniform(0.2, 0.95)), pixel_noise=float(rng.choice([0.2, 0.4, 2.0])))
    tau, ml = float(rng.choice([2.0, 1.2])), float(rng.choice([20, 5, 60]))
    tm = TwoFrameLineMatcher(ctx, tau, K=s["K"], b=s["b"], minLineLength=ml)
    m, d, gate = tm.MatchLines(s["desc_left"], s["desc_right"], lines=s["left"], other_lines=s["right"], octaves=s["left_octave"],
                               other_octaves=s["right_octave"], want_gate=True)
    me, de, ge = O.line_match_stereo(s["K"], s["b"], tau, ml, s["left"], s["left_octave"], s["desc_left"], s["right"], s["right_octave"], s["desc_right"],
                                     want_gate=True)
    return eq((gate, ge), (m, me), (d[m >= 0], de[me >= 0])), int((m >= 0).sum())


def r_line_track(ctx, rng, sid):
    P, L, F = synth.make_line_track_scene(sid, n_map=size(rng, 400, 5), n_cur=size(rng, 500, 5), related_frac=float(rng.uniform(0.2, 0.9)),
                                          pixel_noise=float(rng.choice([0.3, 0.5, 3.0])))
    kw = dict(use_grid=bool(rng.integers(0, 2)), monocular=bool(rng.integers(0, 2)))
    trk = Tracking(ctx, P["K"], P["b"], 1.0 / P["sx"], 1.0 / P["sy"], mdThr=P["md_thr"], monocular=kw["monocular"])
    thr = float(P["thr_reproj_base"] * rng.choice([1.0, 0.5, 3.0]))
    gm, gd, gg = trk.AddLinesFrom(L, P["T_curr"], thr, F, use_grid=kw["use_grid"], want_gate=True)
    om, od, og = O.line_track_match(P["K"], P["T_curr"], P["b"], thr, P["md_thr"], P["sx"], P["sy"], L, F, want_gate=True, **kw)
    return eq((gm, om), (gd[gm >= 0], od[om >= 0])) and bool(np.all(gg[og.astype(bool)] == 1)), int((gm >= 0).sum())


def r_line_lastkf(ctx, rng, sid):
    P, cur, last, _ = synth.make_two_frame_lines(sid, n_lines=size(rng, 400, 2), shared_frac=float(rng.uniform(0.2, 0.9)))
    trk = Tracking(ctx, P["K"], P["b"], 1.0 / P["sx"], 1.0 / P["sy"], mdThr=P["md_thr"])
    ug = bool(rng.integers(0, 2)); thr = float(P["thr_reproj_base"] * rng.choice([1.0, 0.5, 2.0]))
    gm, gc, gx, gdd = trk.MatchLinesLastKF(P["T_curr"], P["T_last"], cur, last, thr, ug)
    om, oc, ox, od = O.line_match_last_frame(P["K"], P["T_curr"], P["T_last"], P["b"], thr, P["md_thr"], P["sx"], P["sy"], cur, last, ug)
    ok = eq((gm, om), (gc, oc))
    c = oc.astype(bool)
    if ok and c.any(): ok = bool(np.allclose(gdd[c], od[c], atol=1e-7) and np.allclose(gx[c], ox[c], rtol=1e-6, atol=1e-6))
    return ok, int((gm >= 0).sum())


def _frame_and_queries(rng, sid):
    n = size(rng, S.MAX_KEYPOINTS, 1)
    F = synth.make_orb_frame(sid, n, n_clusters=int(rng.choice([0, 10, 60])))
    q = synth.make_projection_queries(F, sid, size(rng, 10500 if BIG else 3500), dup_frac=float(rng.choice([0.0, 0.3, 0.5])), pos_sigma=float(rng.choice([1.2, 2.5, 6.0])))
    if QUANT:
        quant_frame(F); q["uv"] = quant(q["uv"]); q["ur"] = quant(q["ur"])
    return F, q


def r_proj_map(ctx, rng, sid):
    F, q = _frame_and_queries(rng, sid)
    th, nn = float(rng.choice([1.0, 3.0, 5.0])), float(rng.choice([0.6, 0.8, 1.0]))
    out = ORBmatcher(ctx, nn).SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], th)
    ne, slot = OS.search_by_projection_map(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], th, nn)
    return out.n_matches == ne and eq((expect_slots(out, q["occupied"]), slot)), ne


def r_proj_frame(ctx, rng, sid):
    F, q = _frame_and_queries(rng, sid)
    q["obs"][::int(rng.integers(2, 9))] = 0
    d, th, chk = int(rng.integers(-1, 2)), float(rng.choice([7.0, 15.0, 3.0])), bool(rng.integers(0, 2))
    out = ORBmatcher(ctx, 0.9, chk).SearchByProjectionFrame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], d, th)
    ne, slot = OS.search_by_projection_frame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], d, th, chk)
    return out.n_matches == ne and eq((expect_slots(out, q["occupied"]), slot)), ne


def r_fuse_inner(ctx, rng, sid):
    F, q = _frame_and_queries(rng, sid)
    th = float(rng.choice([3.0, 4.0, 2.0]))
    out = ORBmatcher(ctx).Fuse(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], th)
    ne, best = OS.fuse_search(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], th)
    return out.n_matches == ne and eq((out.match, best)), ne


def _map_scene(rng, sid):
    F = synth.make_orb_frame(sid, size(rng, S.MAX_KEYPOINTS, 1))
    nm = size(rng, 10500 if BIG else 3500)
    T, mp = synth.make_local_map(F, sid, nm, related_frac=float(rng.uniform(0.2, 0.95)))
    return F, T, mp, nm


def r_local_points(ctx, rng, sid):
    F, T, mp, nm = _map_scene(rng, sid)
    view = S.frame_view(T, synth.KITTI_CAM, F)
    th, nn = float(rng.choice([1.0, 3.0, 5.0])), float(rng.choice([0.8, 0.6]))
    out, fr = S.search_local_points(ctx.lib, ctx.handle, F, view, mp, mp["occupied"], th, nn)
    k, inv, uvr, lvl, vc = OS.is_in_frustum(view, mp)
    m = inv != 0
    ok = eq((fr["in_view"], inv), (fr["proj_uvr"][m], uvr[m]), (fr["view_cos"][m], vc[m]), (fr["level"][m], lvl[m]))
    ne, slot = OS.search_by_projection_map(F, mp["desc"], inv, uvr[:, :2], uvr[:, 2], lvl, vc, mp["has_obs"], mp["occupied"], th, nn)
    return ok and out.n_matches == ne and eq((expect_slots(out, mp["occupied"]), slot)), ne


def r_last_frame(ctx, rng, sid):
    F, T, mp, nm = _map_scene(rng, sid)
    ang = np.mod(F.angle[mp["src"]] + 25.0 + rng.normal(0, 6.0, nm), 360.0)
    wild = rng.random(nm) < 0.15; ang[wild] = rng.uniform(0, 360, int(wild.sum()))
    last = dict(world_pos=mp["world_pos"], valid=(rng.random(nm) < 0.85).astype(np.uint8), octave=F.octave[mp["src"]], angle=ang.astype(f32),
                desc=mp["desc"], has_obs=mp["has_obs"])
    view = S.frame_view(T, synth.KITTI_CAM, F)
    d, th, chk = int(rng.integers(-1, 2)), float(rng.choice([7.0, 15.0])), bool(rng.integers(0, 2))
    out, uvr = S.search_last_frame(ctx.lib, ctx.handle, F, view, last, mp["occupied"], d, th, chk)
    valid, uv, ur = OS.project_last_frame(view, last)
    m = valid != 0
    ok = eq((uvr[m, :2], uv[m]), (uvr[m, 2], ur[m]))
    ne, slot = OS.search_by_projection_frame(F, last["desc"], valid, uv, ur, last["octave"], last["angle"], last["has_obs"], mp["occupied"], d, th, chk)
    return ok and out.n_matches == ne and eq((expect_slots(out, mp["occupied"]), slot)), ne


def r_fuse(ctx, rng, sid):
    KF, T, mp, nm = _map_scene(rng, sid)
    view = S.frame_view(T, synth.KITTI_CAM, KF)
    th = float(rng.choice([3.0, 4.0]))
    out, uvr = S.fuse_search_points(ctx.lib, ctx.handle, KF, view, mp, th)
    valid, uv, ur, lvl = OS.project_fuse(view, mp)
    m = valid != 0
    ne, best = OS.fuse_search(KF, mp["desc"], valid, uv, ur, lvl, th)
    return eq((uvr[m, :2], uv[m]), (uvr[m, 2], ur[m]), (out.match, best)) and out.n_matches == ne, ne


def _bow(rng, sid, **kw):
    n = size(rng, S.MAX_KEYPOINTS, 2)
    return synth.make_bow_pair(sid, n, n_nodes=int(rng.choice([1, 20, 400, 2000])), related_frac=float(rng.uniform(0.2, 0.9)), **kw)


def r_bow_frame(ctx, rng, sid):
    F1, F2, nd = _bow(rng, sid)
    valid = (rng.random(F1.n) < rng.choice([0.3, 0.85, 1.0])).astype(np.uint8)
    nn, chk = float(rng.choice([0.7, 0.9])), bool(rng.integers(0, 2))
    out = ORBmatcher(ctx, nn, chk).SearchByBoWFrame(F1, F2, nd, valid)
    ne, fm = OS.search_by_bow_frame(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], valid, nn, chk)
    got = np.where(out.owner >= 0, out.query_kp[np.maximum(out.owner, 0)], -1) if out.query_kp.size else -np.ones(F2.n, np.int32)
    return out.n_matches == ne and eq((got, fm)), ne


def r_bow_kf(ctx, rng, sid):
    F1, F2, nd = _bow(rng, sid)
    v1 = (rng.random(F1.n) < 0.85).astype(np.uint8); v2 = (rng.random(F2.n) < rng.choice([0.4, 0.85])).astype(np.uint8)
    nn, chk = float(rng.choice([0.75, 0.9])), bool(rng.integers(0, 2))
    out = ORBmatcher(ctx, nn, chk).SearchByBoWKF(F1, F2, nd, v1, v2)
    ne, m12 = OS.search_by_bow_kf(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], v1, v2, nn, chk)
    got = -np.ones(F1.n, np.int32); got[out.query_kp] = out.final_match()
    return out.n_matches == ne and eq((got, m12)), ne


def r_triangulation(ctx, rng, sid):
    F1, F2, nd = _bow(rng, sid, pos_sigma=(25.0, 1.5))
    if QUANT: quant_frame(F1); quant_frame(F2)
    F12 = (np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0.0]]) + rng.normal(0, 2e-6, (3, 3))).astype(f32)
    has1 = (rng.random(F1.n) < rng.choice([0.3, 0.6])).astype(np.uint8); has2 = (rng.random(F2.n) < 0.3).astype(np.uint8)
    only = bool(rng.integers(0, 2)); chk = bool(rng.integers(0, 2))
    if not only: F1.uright[::2] = -1; F2.uright[::3] = -1
    epipole = (float(rng.uniform(0, 1241)), float(rng.uniform(0, 376)))
    out = ORBmatcher(ctx, 0.6, chk).SearchForTriangulation(F1, F2, nd, has1, has2, OS.epipolar_lines(F12, F1.xy), epipole, only)
    ne, m12 = OS.search_for_triangulation(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], has1, has2, F12, epipole, only, chk)
    got = -np.ones(F1.n, np.int32); got[out.query_kp] = out.final_match()
    return out.n_matches == ne and eq((got, m12)), ne


def r_init(ctx, rng, sid):
    n = size(rng, S.MAX_KEYPOINTS, 2)
    F1, F2, prev = synth.make_init_pair(sid, n=n, rival_frac=float(rng.choice([0.0, 0.15, 0.5])), flow_sigma=float(rng.choice([4.0, 12.0, 40.0])))
    if QUANT: quant_frame(F1); quant_frame(F2); prev = quant(prev)
    w, nn, chk = int(rng.choice([10, 30, 60, 100, 150])), float(rng.choice([0.7, 0.9, 0.95])), bool(rng.integers(0, 2))
    on, om, opm = OS.search_for_initialization(F1, F2, prev, w, nn, chk)
    gn, gm, gpm = ORBmatcher(ctx, nn, chk).SearchForInitialization(F1, F2, prev, w)
    return gn == on and eq((gm, om), (gpm, opm)), on


def r_stereo(ctx, rng, sid):
    sc = synth.make_stereo_scene(sid, size(rng, S.MAX_KEYPOINTS, 1))
    if QUANT: quant_frame(sc["L"]); quant_frame(sc["R"])
    g = ORBmatcher(ctx).ComputeStereoMatchesFull(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    n, ur, dep, br, sad = OS.compute_stereo_matches(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    return g.n_matches == n and eq((g.best_r, br), (g.sad, sad), (g.u_right.view(np.uint32), ur.view(np.uint32)), (g.depth.view(np.uint32), dep.view(np.uint32))), n


def r_batch(ctx, rng, sid):
    """lld_orb_search_batch: 2..10 problems of mixed routines and sizes in ONE launch; every problem against its own oracle."""
    prepared, expect = [], []
    for k in range(int(rng.integers(2, 11))):
        kind = int(rng.integers(0, 4)); sub = int(rng.integers(0, 1 << 30))
        if kind == 0:
            F, q = _frame_and_queries(rng, sub); th, nn = float(rng.choice([1.0, 3.0])), float(rng.choice([0.6, 0.8]))
            prepared.append(S.search_by_projection_map(None, None, F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], th, nn))
            expect.append(("slots", q["occupied"]) + OS.search_by_projection_map(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], th, nn))
        elif kind == 1:
            F, q = _frame_and_queries(rng, sub); d, th, chk = int(rng.integers(-1, 2)), float(rng.choice([7.0, 15.0])), bool(rng.integers(0, 2))
            prepared.append(S.search_by_projection_frame(None, None, F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], d, th, chk))
            expect.append(("slots", q["occupied"]) + OS.search_by_projection_frame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], d, th, chk))
        elif kind == 2:
            L, R = synth.make_stereo_pair(sub, size(rng, S.MAX_KEYPOINTS, 1))
            prepared.append(S.stereo_search(None, None, L, R, 0.0, 100.0)); expect.append(("stereo",) + OS.stereo_search(L, R, 376, 0.0, 100.0))
        else:
            F1, F2, nd = _bow(rng, sub); v = (rng.random(F1.n) < 0.85).astype(np.uint8); nn, chk = float(rng.choice([0.7, 0.9])), bool(rng.integers(0, 2))
            prepared.append(S.search_by_bow_frame(None, None, F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], v, nn, chk))
            expect.append(("bow",) + OS.search_by_bow_frame(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], v, nn, chk))
    outs = S.run_batch(ctx.lib, ctx.handle, prepared)
    ok = True; nm = 0
    for out, e in zip(outs, expect):
        if e[0] == "slots": ok = ok and out.n_matches == e[2] and eq((expect_slots(out, e[1]), e[3])); nm += int(e[2])
        elif e[0] == "stereo": ok = ok and eq((out.match, e[1])); nm += int((e[1] >= 0).sum())
        else:
            got = np.where(out.owner >= 0, out.query_kp[np.maximum(out.owner, 0)], -1) if out.query_kp.size else -np.ones_like(e[2])
            ok = ok and out.n_matches == e[1] and eq((got, e[2])); nm += int(e[1])
    return ok, nm


ROUTINES = dict(batch=r_batch, hamming=r_hamming, l2=r_l2, line_greedy=r_line_greedy, line_stereo=r_line_stereo, line_track=r_line_track, line_lastkf=r_line_lastkf,
                proj_map=r_proj_map, proj_frame=r_proj_frame, fuse_inner=r_fuse_inner, local_points=r_local_points, last_frame=r_last_frame, fuse=r_fuse,
                bow_frame=r_bow_frame, bow_kf=r_bow_kf, triangulation=r_triangulation, init=r_init, stereo=r_stereo)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    ctx = Context(0); O.lib()
    names = [only] if only else sorted(ROUTINES)
    stats = {k: [0, 0, 0, 0.0] for k in names}                         # scenes, mismatches, matches, seconds
    t_all = time.time()
    for it in range(n):
        name = names[it % len(names)]
        sid = int(rng.integers(0, 1 << 30)); sub = np.random.default_rng([seed, it])
        t0 = time.time()
        try:
            ok, nm = ROUTINES[name](ctx, sub, sid)
        except Exception as e:                                          # a refusal or a crash of either side is a finding too
            ok, nm = False, 0; print(f"EXC {name} seed={seed} it={it} sid={sid}: {type(e).__name__}: {e}", flush=True)
        st = stats[name]; st[0] += 1; st[1] += 0 if ok else 1; st[2] += int(nm); st[3] += time.time() - t0
        if not ok: print(f"MISMATCH {name} seed={seed} it={it} sid={sid}", flush=True)
    print(f"# tools/fuzz_matchers.py {n} {seed}: random scenes per matcher, device vs oracle, every integer output bit for bit")
    print(f"{'routine':<14} {'scenes':>7} {'mismatches':>10} {'matches':>10} {'s':>8}")
    for k in names:
        s = stats[k]; print(f"{k:<14} {s[0]:>7} {s[1]:>10} {s[2]:>10} {s[3]:>8.1f}")
    tot = [sum(s[i] for s in stats.values()) for i in range(3)]
    print(f"{'total':<14} {tot[0]:>7} {tot[1]:>10} {tot[2]:>10} {time.time() - t_all:>8.1f}")
    ctx.close()
    return 1 if tot[1] else 0


if __name__ == "__main__":
    sys.exit(main())
