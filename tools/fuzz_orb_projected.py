"""Random scenes for the matchers whose projection loops moved to the device in round 3 (lld_orb_search_projected / lld_orb_search_by_sim3):
random frame sizes, map sizes, poses, Sim3 scales, radii, thresholds, occupancy, skip sets - device vs the oracle's literal restatements,
every output bit for bit.   python tools/fuzz_orb_projected.py [n=400] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from lld_slam_amd import Context, orb_search as S, synth
import oracle_orbsearch as OS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
bad = 0; checked = {0: 0, 1: 0, 2: 0, 3: 0, "sim3": 0}; matches = 0


def expect_slots(out, occupied, token=1 << 20):
    slot = OS.slots_from_occupied(occupied)
    ok = (out.match >= 0) & (out.removed == 0)
    slot[out.match[ok]] = np.nonzero(ok)[0]
    return slot


for it in range(n):
    nk = int(rng.integers(1, 2600)); nm = int(rng.integers(0, 3000))
    F = synth.make_orb_frame(int(rng.integers(0, 1 << 30)), nk).normalise()
    T, mp = synth.make_local_map(F, int(rng.integers(0, 1 << 30)), max(nm, 1))
    if nm == 0: mp = {k: (v[:0] if k != "occupied" else v) for k, v in mp.items()}
    scale = float(rng.choice([1.0, rng.uniform(0.3, 3.0)]))
    Scw = np.array(T, np.float32, copy=True); Scw[:3, :] = (np.float64(scale) * T[:3, :].astype(np.float64)).astype(np.float32)
    routine = int(rng.integers(0, 4))
    th = float(rng.choice([3.0, 4.0, 7.5, 10.0, 15.0]))
    try:
        if routine == S.PROJ_RELOC:
            view = S.frame_view(T, synth.KITTI_CAM, F)
            ang = rng.uniform(0, 360, nm).astype(np.float32); orb = int(rng.choice([50, 64, 100])); chk = bool(rng.integers(0, 2))
            out, uv, lvl = S.search_projected(lib, h, F, view, mp, routine, th, accept_max=orb, check_orientation=chk, angle=ang, occupied=mp["occupied"])
            v, uvo, lo = OS.project_general(view, mp, routine)
            ne, slot = OS.search_by_projection_reloc(F, mp["desc"], v, uvo, lo, ang, mp["occupied"], th, orb, chk)
            ok = out.n_matches == ne and np.array_equal(expect_slots(out, mp["occupied"]), slot)
        elif routine == S.PROJ_KF_SIM3:
            view = S.sim3_view(Scw, synth.KITTI_CAM, F); thi = int(th)
            out, uv, lvl = S.search_projected(lib, h, F, view, mp, routine, thi, occupied=mp["occupied"])
            v, uvo, lo = OS.project_general(view, mp, routine)
            ne, slot = OS.search_by_projection_kf(F, mp["desc"], v, uvo, lo, mp["occupied"], thi)
            ok = out.n_matches == ne and np.array_equal(expect_slots(out, mp["occupied"]), slot)
        elif routine == S.PROJ_FUSE_SIM3:
            view = S.sim3_view(Scw, synth.KITTI_CAM, F)
            out, uv, lvl = S.search_projected(lib, h, F, view, mp, routine, th)
            v, uvo, lo = OS.project_general(view, mp, routine)
            ne, best = OS.fuse_search_sim3(F, mp["desc"], v, uvo, lo, th)
            ok = out.n_matches == ne and np.array_equal(out.match, best)
        else:
            view = S.frame_view(T, synth.KITTI_CAM, F)
            sR = (rng.uniform(0.8, 1.25) * synth._rodrigues(rng.normal(0, 0.01, 3))).astype(np.float32); t2 = rng.normal(0, 0.05, 3).astype(np.float32)
            out, uv, lvl = S.search_projected(lib, h, F, view, mp, routine, th, sR=sR, t=t2)
            v, uvo, lo = OS.project_general(view, mp, routine, sR, t2)
            m = OS.search_sim3_direction(F, mp["desc"], v, uvo, lo, th)
            ok = np.array_equal(out.match, m)
        k = v != 0
        ok = ok and np.array_equal(uv[k], uvo[k]) and np.array_equal(lvl[k], lo[k])
        checked[routine] += 1; matches += int(out.n_matches)
    except Exception as e:                                      # noqa: BLE001
        ok = False; print("ERROR", it, routine, repr(e)[:200])
    if not ok:
        bad += 1; print("MISMATCH", it, "routine", routine, "keypoints", nk, "points", nm, "scale", scale, "th", th)
    if it % 8 == 0 and nk > 10:                                 # SearchBySim3 as a whole on every eighth scene
        K2 = synth.make_orb_frame(int(rng.integers(0, 1 << 30)), int(rng.integers(10, 2000))).normalise()
        T2, mp1 = synth.make_local_map(K2, int(rng.integers(0, 1 << 30)), F.n); T1, mp2 = synth.make_local_map(F, int(rng.integers(0, 1 << 30)), K2.n)
        s12 = float(rng.uniform(0.8, 1.25))
        R1, t1, R2, t2 = T1[:3, :3].astype(np.float64), T1[:3, 3].astype(np.float64), T2[:3, :3].astype(np.float64), T2[:3, 3].astype(np.float64)
        R12 = (R1 @ R2.T).astype(np.float32); t12 = (t1 - s12 * (R1 @ R2.T) @ t2).astype(np.float32)
        v1 = S.frame_view(T1, synth.KITTI_CAM, F); v2 = S.frame_view(T2, synth.KITTI_CAM, K2)
        sR12, t12f, sR21, t21 = S.sim3_transforms(s12, R12, t12)
        m12, found = S.search_by_sim3_points(lib, h, F, v1, mp1, K2, v2, mp2, sR12, t12f, sR21, t21, 7.5)
        mixa = S.FrameView.from_buffer_copy(v1); mixb = S.FrameView.from_buffer_copy(v2)
        mixa.min_x, mixa.max_x, mixa.min_y, mixa.max_y, mixa.log_scale_factor, mixa.n_levels = v2.min_x, v2.max_x, v2.min_y, v2.max_y, v2.log_scale_factor, v2.n_levels
        mixb.fx, mixb.fy, mixb.cx, mixb.cy = v1.fx, v1.fy, v1.cx, v1.cy
        mixb.min_x, mixb.max_x, mixb.min_y, mixb.max_y, mixb.log_scale_factor, mixb.n_levels = v1.min_x, v1.max_x, v1.min_y, v1.max_y, v1.log_scale_factor, v1.n_levels
        va, uva, la = OS.project_general(mixa, mp1, 3, sR21, t21); vb, uvb, lb = OS.project_general(mixb, mp2, 3, sR12, t12f)
        a = OS.search_sim3_direction(K2, mp1["desc"], va, uva, la, 7.5); b = OS.search_sim3_direction(F, mp2["desc"], vb, uvb, lb, 7.5)
        exp = np.array([a[i] if a[i] >= 0 and b[a[i]] == i else -1 for i in range(F.n)], np.int32)
        checked["sim3"] += 1
        if not (np.array_equal(m12, exp) and found == int((exp >= 0).sum())):
            bad += 1; print("MISMATCH", it, "SearchBySim3", F.n, K2.n, s12)
print("fuzzed %d scenes: per routine %s, %d matches in total, %d mismatches / errors" % (n, checked, matches, bad))
