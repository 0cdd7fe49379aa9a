"""The C++ route: examples/harness.cpp calls liblld_amd.so through include/lld_amd.hpp (the header a patched reference would
include, mirroring include/Optimizer.h:49-50 / ORBmatcher.h / TwoFrameLineMatcher.h) and must reproduce the golden vectors."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HARNESS = os.path.join(ROOT, "examples", "harness")

BA_ARRAYS = (("cam_qt", np.float64), ("pt_xyz", np.float64), ("pt_obs_start", np.int32), ("pt_obs_cam", np.int32),
             ("pt_obs_uvr", np.float64), ("pt_obs_inv_sigma2", np.float64), ("line_x0", np.float64), ("line_dir", np.float64),
             ("ln_obs_start", np.int32), ("ln_obs_cam", np.int32), ("ln_obs_left", np.float64), ("ln_obs_right", np.float64),
             ("ln_obs_octave", np.int32))
POSE_ARRAYS = (("pose_qt", np.float64), ("pt_xw", np.float64), ("pt_uvr", np.float64), ("pt_inv_sigma2", np.float64),
               ("ln_x0", np.float64), ("ln_dir", np.float64), ("ln_left", np.float64), ("ln_right", np.float64), ("ln_octave", np.int32))


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists(os.path.join(ROOT, "lld_slam_amd", "csrc", "liblld_amd.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "lld_slam_amd", "csrc")], check=True, capture_output=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    assert os.path.exists(HARNESS)
    return HARNESS


def run(harness, mode, tmp_path):
    return subprocess.run([harness, mode, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)


def write_ba(path, d, stop=0, iterations=0):
    dims = (d["cam_qt"].shape[0], d["pt_xyz"].shape[0], d["pt_obs_cam"].shape[0], d["line_x0"].shape[0], d["ln_obs_cam"].shape[0])
    with open(path, "wb") as f:
        np.array([dims[0], int(d["n_free_cams"]), dims[1], dims[2], dims[3], dims[4], stop, iterations], np.int32).tofile(f)
        np.concatenate([d["cam"], [float(d["gamma"]) if "gamma" in d else 1.0]]).astype(np.float64).tofile(f)
        for k, t in BA_ARRAYS:
            np.ascontiguousarray(d[k], t).tofile(f)
    return dims


def read_ba(path, n_cams, n_pts, n_po, n_ln, n_lo):
    with open(path, "rb") as f:
        return {"cam_qt": np.fromfile(f, np.float64, 7 * n_cams).reshape(-1, 7), "pt_xyz": np.fromfile(f, np.float64, 3 * n_pts).reshape(-1, 3),
                "line_x0": np.fromfile(f, np.float64, 3 * n_ln).reshape(-1, 3), "line_dir": np.fromfile(f, np.float64, 3 * n_ln).reshape(-1, 3),
                "pt_obs_outlier": np.fromfile(f, np.uint8, n_po), "ln_edge_outlier": np.fromfile(f, np.uint8, 2 * n_lo).reshape(-1, 2),
                "line_removed": np.fromfile(f, np.uint8, n_ln), "chi2": np.fromfile(f, np.float64, 2), "st": np.fromfile(f, np.int32, 4)}


def test_harness_builds_and_refuses_without_gpu(harness, tmp_path):
    """CPU: the C++ layer compiles with plain g++ -std=c++11 and, with no device, fails loudly (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    write_ba(tmp_path / "in.bin", np.load(os.path.join(GOLD, "lba_small.npz")))
    r = run(harness, "ba", tmp_path)
    assert r.returncode == 1
    assert "no HIP device" in r.stderr


@pytest.mark.gpu
def test_harness_local_ba_golden(harness, tmp_path):
    d = np.load(os.path.join(GOLD, "lba_small.npz"))
    dims = write_ba(tmp_path / "in.bin", d)
    r = run(harness, "ba", tmp_path)
    assert r.returncode == 0, r.stderr
    o = read_ba(tmp_path / "out.bin", *dims)
    rtol = 1e-5     # north_star tolerance
    assert abs(o["chi2"][0] - float(d["out_chi2_round1"])) <= rtol * float(d["out_chi2_round1"])
    assert abs(o["chi2"][1] - float(d["out_chi2_final"])) <= rtol * float(d["out_chi2_final"])
    assert np.array_equal(o["pt_obs_outlier"], d["out_pt_obs_outlier"])
    assert np.array_equal(o["ln_edge_outlier"], d["out_ln_edge_outlier"])
    assert np.array_equal(o["line_removed"], d["out_line_removed"])
    assert np.max(np.abs(o["cam_qt"] - d["out_cam_qt"])) <= rtol * np.max(np.abs(d["out_cam_qt"]))
    err = np.linalg.norm(o["pt_xyz"] - d["out_pt_xyz"], axis=1) / np.linalg.norm(d["out_pt_xyz"], axis=1)
    assert np.median(err) <= rtol and err.max() <= 10 * rtol


@pytest.mark.gpu
def test_harness_abort_at_start_leaves_window_untouched(harness, tmp_path):
    """pbStopFlag already set: Optimizer.cc:1220-1222 returns before optimize(); nothing moves, nothing is flagged."""
    d = np.load(os.path.join(GOLD, "lba_small.npz"))
    dims = write_ba(tmp_path / "in.bin", d, stop=1)
    r = run(harness, "ba", tmp_path)
    assert r.returncode == 0, r.stderr
    o = read_ba(tmp_path / "out.bin", *dims)
    assert o["st"][2] == 1 and o["st"][0] == 0
    assert np.array_equal(o["pt_xyz"], d["pt_xyz"])
    assert not o["pt_obs_outlier"].any() and not o["ln_edge_outlier"].any() and not o["line_removed"].any()


@pytest.mark.gpu
def test_harness_pose_golden(harness, tmp_path):
    d = np.load(os.path.join(GOLD, "pose_small.npz"))
    n_pts, n_ln = d["pt_xw"].shape[0], d["ln_x0"].shape[0]
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([n_pts, n_ln], np.int32).tofile(f)
        np.concatenate([d["cam"], [float(d["gamma"]) if "gamma" in d else 1.0]]).astype(np.float64).tofile(f)
        for k, t in POSE_ARRAYS:
            np.ascontiguousarray(d[k], t).tofile(f)
    r = run(harness, "pose", tmp_path)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        qt = np.fromfile(f, np.float64, 7)
        n_in = np.fromfile(f, np.int32, 1)[0]
        po, lo = np.fromfile(f, np.uint8, n_pts), np.fromfile(f, np.uint8, n_ln)
    assert n_in == int(d["out_n_inliers"])
    assert np.array_equal(po, d["out_pt_outlier"]) and np.array_equal(lo, d["out_ln_outlier"])
    assert np.max(np.abs(qt - d["out_pose_qt"])) <= 1e-5 * np.max(np.abs(d["out_pose_qt"]))


@pytest.mark.gpu
def test_harness_orb_golden(harness, tmp_path):
    d = np.load(os.path.join(GOLD, "match_orb.npz"))
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([d["q"].shape[0], d["t"].shape[0]], np.int32).tofile(f)
        np.ascontiguousarray(d["q"], np.uint32).tofile(f)
        np.ascontiguousarray(d["t"], np.uint32).tofile(f)
    r = run(harness, "orb", tmp_path)
    assert r.returncode == 0, r.stderr
    out = np.fromfile(tmp_path / "out.bin", np.int32).reshape(4, -1)
    for got, key in zip(out, ("best_idx", "best_dist", "second_idx", "second_dist")):
        assert np.array_equal(got, d[key]), key


@pytest.mark.gpu
def test_harness_global_ba_golden(harness, tmp_path):
    d = np.load(os.path.join(GOLD, "gba_small.npz"))
    dims = write_ba(tmp_path / "in.bin", d, iterations=int(d["iterations"]))
    r = run(harness, "gba", tmp_path)
    assert r.returncode == 0, r.stderr
    o = read_ba(tmp_path / "out.bin", *dims)
    rtol = 1e-5
    assert abs(o["chi2"][1] - float(d["out_chi2_final"])) <= rtol * float(d["out_chi2_final"])
    assert o["st"][1] == 0 and not o["pt_obs_outlier"].any() and not o["line_removed"].any()
    assert np.max(np.abs(o["cam_qt"] - d["out_cam_qt"])) <= rtol * np.max(np.abs(d["out_cam_qt"]))


@pytest.mark.gpu
def test_harness_optimize_sim3(harness, tmp_path, oracle):
    from lld_slam_amd import synth
    p = synth.make_sim3_pair(30, 200)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([p.n, 1], np.int32).tofile(f)
        np.concatenate([np.asarray(p.K1, np.float64), np.asarray(p.K2, np.float64), p.s12_q, p.s12_t, [p.s12_s], [10.0]]).astype(np.float64).tofile(f)
        for a in (p.p1c, p.p2c, p.obs1, p.obs2, p.inv_sigma2_1, p.inv_sigma2_2):
            np.ascontiguousarray(a, np.float64).tofile(f)
    r = run(harness, "sim3", tmp_path)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        S = np.fromfile(f, np.float64, 8); n_in = int(np.fromfile(f, np.int32, 1)[0]); dropped = np.fromfile(f, np.uint8, p.n)
    o = oracle.optimize_sim3(p)
    assert n_in == o.n_inliers and np.array_equal(dropped, o.dropped)
    np.testing.assert_allclose(S[:4], o.s12_q, rtol=1e-5, atol=1e-7); np.testing.assert_allclose(S[4:7], o.s12_t, rtol=1e-5, atol=1e-6)


def _write_frame_lines(f, left, right, octave, matches, occupied, desc):
    np.array([left.shape[0], right.shape[0]], np.int32).tofile(f)
    np.ascontiguousarray(left, np.float32).tofile(f); np.ascontiguousarray(right, np.float32).tofile(f)
    np.ascontiguousarray(octave, np.int32).tofile(f); np.ascontiguousarray(matches, np.int32).tofile(f)
    np.ascontiguousarray(occupied, np.uint8).tofile(f); np.ascontiguousarray(desc, np.float32).tofile(f)


@pytest.mark.gpu
def test_harness_tracking_line_matchers_golden(harness, tmp_path):
    """lld_amd::Tracking::AddLinesFrom / MatchLinesLastKF (include/lld_amd.hpp) on tests/golden/line_track.npz."""
    d = np.load(os.path.join(GOLD, "line_track.npz"))
    dim = d["l_desc"].shape[1]; n_map = d["l_X0"].shape[0]
    def params(pre, T_last=None):
        return np.concatenate([d[pre + "K"].reshape(-1), d[pre + "T_curr"].reshape(-1), (np.zeros(16) if T_last is None else T_last.reshape(-1)),
                               [float(d[pre + "b"]), 1.0 / float(d[pre + "sx"]), 1.0 / float(d[pre + "sy"]), float(d[pre + "md_thr"]), float(d[pre + "thr_reproj_base"])]]).astype(np.float64)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([dim, n_map, 1], np.int32).tofile(f)
        params("p_").tofile(f)
        for k in ("l_X0", "l_dir", "l_X1", "l_X2"): np.ascontiguousarray(d[k], np.float64).tofile(f)
        np.ascontiguousarray(d["l_skip"], np.uint8).tofile(f); np.ascontiguousarray(d["l_desc"], np.float32).tofile(f)
        _write_frame_lines(f, d["f_left_lines"], d["f_right_lines"], d["f_left_octave"], d["f_line_matches"], d["f_occupied"], d["f_desc"])
        params("p2_", d["p2_T_last"]).tofile(f)
        nc = d["c_left_lines"].shape[0]
        _write_frame_lines(f, d["c_left_lines"], d["c_right_lines"], np.zeros(nc, np.int32), d["c_line_matches"], d["c_occupied"], d["c_desc"])
        _write_frame_lines(f, d["k_left_lines"], d["k_right_lines"], d["k_left_octave"], d["k_line_matches"], d["k_skip"], d["k_desc"])
    r = run(harness, "lines", tmp_path)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        m = np.fromfile(f, np.int32, n_map); ml = np.fromfile(f, np.int32, nc); created = np.fromfile(f, np.uint8, nc)
        x0 = np.fromfile(f, np.float64, 3 * nc).reshape(-1, 3); dr = np.fromfile(f, np.float64, 3 * nc).reshape(-1, 3)
    np.testing.assert_array_equal(m, d["out_m_grid"])
    np.testing.assert_array_equal(ml, d["out_k_match"]); np.testing.assert_array_equal(created, d["out_k_created"])
    ok = created.astype(bool)
    np.testing.assert_allclose(x0[ok], d["out_k_x0"][ok], rtol=1e-6, atol=1e-6); np.testing.assert_allclose(dr[ok], d["out_k_dir"][ok], atol=1e-7)


@pytest.mark.gpu
def test_harness_search_for_initialization(harness, tmp_path):
    """lld_amd::ORBmatcher::SearchForInitialization (include/lld_amd.hpp) against the oracle."""
    import oracle_orbsearch as OS
    from lld_slam_amd import synth
    F1, F2, prev = synth.make_init_pair(12, n=900)
    on, om, opm = OS.search_for_initialization(F1, F2, prev, 60, 0.9, True)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([F1.n, F2.n, 60, 1], np.int32).tofile(f)
        np.array([F2.min_x, F2.min_y, F2.width_inv, F2.height_inv, 0.9], np.float32).tofile(f)
        np.ascontiguousarray(F1.desc, np.uint32).tofile(f); np.ascontiguousarray(F1.octave, np.int32).tofile(f); np.ascontiguousarray(F1.angle, np.float32).tofile(f)
        np.ascontiguousarray(prev, np.float32).tofile(f)
        np.ascontiguousarray(F2.desc, np.uint32).tofile(f); np.ascontiguousarray(F2.octave, np.int32).tofile(f); np.ascontiguousarray(F2.angle, np.float32).tofile(f)
        np.ascontiguousarray(F2.xy, np.float32).tofile(f)
    r = run(harness, "init", tmp_path)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        n = int(np.fromfile(f, np.int32, 1)[0]); m = np.fromfile(f, np.int32, F1.n); pm = np.fromfile(f, np.float32, 2 * F1.n).reshape(-1, 2)
    assert n == on and on > 100
    np.testing.assert_array_equal(m, om); np.testing.assert_array_equal(pm, opm)


def _write_side(f, F, view, occupied):
    import ctypes
    np.array([F.n, F.scale.shape[0]], np.int32).tofile(f)
    np.array([F.min_x, F.min_y, F.width_inv, F.height_inv], np.float32).tofile(f)
    f.write(bytes(memoryview(ctypes.string_at(ctypes.addressof(view), ctypes.sizeof(view)))))
    np.ascontiguousarray(F.desc, np.uint32).tofile(f); np.ascontiguousarray(F.xy, np.float32).tofile(f); np.ascontiguousarray(F.octave, np.int32).tofile(f)
    np.ascontiguousarray(F.angle, np.float32).tofile(f); np.ascontiguousarray(occupied, np.uint8).tofile(f); np.ascontiguousarray(F.scale, np.float32).tofile(f)


def _write_points(f, mp, angle):
    n = mp["world_pos"].shape[0]
    np.array([n], np.int32).tofile(f)
    for k, t in (("world_pos", np.float32), ("normal", np.float32), ("max_distance", np.float32), ("min_distance", np.float32), ("desc", np.uint32),
                 ("skip", np.uint8)):
        np.ascontiguousarray(mp[k], t).tofile(f)
    np.ascontiguousarray(angle, np.float32).tofile(f)


@pytest.mark.gpu
def test_harness_relocalisation_and_loop_closing_matchers(harness, tmp_path):
    """lld_amd::ORBmatcher's SearchByProjection(Frame&, KeyFrame*) / SearchByProjection(KeyFrame*, Scw) / Fuse(KeyFrame*, Scw) /
    SearchBySim3 (include/lld_amd.hpp) with the projection loops on the device, against the oracle."""
    import oracle_orbsearch as OS
    from lld_slam_amd import orb_search, synth
    KF = synth.make_orb_frame(300, 1500).normalise()
    T, mp = synth.make_local_map(KF, 300, 1800)
    view = orb_search.frame_view(T, synth.KITTI_CAM, KF)
    ang = np.mod(KF.angle[mp["src"]] + 30.0, 360.0).astype(np.float32)
    K1 = synth.make_orb_frame(301, 1200).normalise(); K2 = synth.make_orb_frame(302, 1100).normalise()
    T2, mp1 = synth.make_local_map(K2, 301, K1.n); T1, mp2 = synth.make_local_map(K1, 302, K2.n)
    v1 = orb_search.frame_view(T1, synth.KITTI_CAM, K1); v2 = orb_search.frame_view(T2, synth.KITTI_CAM, K2)
    R1, t1, R2, t2 = T1[:3, :3].astype(np.float64), T1[:3, 3].astype(np.float64), T2[:3, :3].astype(np.float64), T2[:3, 3].astype(np.float64)
    R12 = R1 @ R2.T; t12 = t1 - R12 @ t2
    sR12, t12f, sR21, t21 = orb_search.sim3_transforms(1.0, R12, t12)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([10.0, 100.0, 10.0, 4.0], np.float32).tofile(f)
        _write_side(f, KF, view, mp["occupied"]); _write_points(f, mp, ang)
        _write_side(f, K1, v1, np.zeros(K1.n, np.uint8)); _write_points(f, mp1, np.zeros(K1.n, np.float32))
        _write_side(f, K2, v2, np.zeros(K2.n, np.uint8)); _write_points(f, mp2, np.zeros(K2.n, np.float32))
        np.concatenate([sR12.reshape(9), t12f, sR21.reshape(9), t21, [7.5]]).astype(np.float32).tofile(f)
    r = run(harness, "loop", tmp_path)
    assert r.returncode == 0, r.stderr
    n = mp["world_pos"].shape[0]
    with open(tmp_path / "out.bin", "rb") as f:
        counts = np.fromfile(f, np.int32, 4); reloc = np.fromfile(f, np.int32, n); removed = np.fromfile(f, np.uint8, n)
        scw = np.fromfile(f, np.int32, n); fuse = np.fromfile(f, np.int32, n); m12 = np.fromfile(f, np.int32, K1.n)

    def slots(match, rem):
        s = OS.slots_from_occupied(mp["occupied"])
        ok = (match >= 0) & (rem == 0)
        s[match[ok]] = np.nonzero(ok)[0]
        return s
    va, uva, la = OS.project_general(view, mp, orb_search.PROJ_RELOC)
    n_exp, slot = OS.search_by_projection_reloc(KF, mp["desc"], va, uva, la, ang, mp["occupied"], 10.0, 100, True)
    assert counts[0] == n_exp and n_exp > 100
    np.testing.assert_array_equal(slots(reloc, removed), slot)
    vb, uvb, lb = OS.project_general(view, mp, orb_search.PROJ_KF_SIM3)
    n_exp, slot = OS.search_by_projection_kf(KF, mp["desc"], vb, uvb, lb, mp["occupied"], 10)
    assert counts[1] == n_exp and n_exp > 100
    np.testing.assert_array_equal(slots(scw, np.zeros(n, np.uint8)), slot)
    vc, uvc, lc = OS.project_general(view, mp, orb_search.PROJ_FUSE_SIM3)
    n_exp, best = OS.fuse_search_sim3(KF, mp["desc"], vc, uvc, lc, 4.0)
    assert counts[2] == n_exp and n_exp > 100
    np.testing.assert_array_equal(fuse, best)
    mix = orb_search.FrameView.from_buffer_copy(v1)
    mix.min_x, mix.max_x, mix.min_y, mix.max_y, mix.log_scale_factor, mix.n_levels = v2.min_x, v2.max_x, v2.min_y, v2.max_y, v2.log_scale_factor, v2.n_levels
    vd, uvd, ld = OS.project_general(mix, mp1, orb_search.PROJ_SIM3_DIR, sR21, t21)
    mix2 = orb_search.FrameView.from_buffer_copy(v2)
    mix2.fx, mix2.fy, mix2.cx, mix2.cy = v1.fx, v1.fy, v1.cx, v1.cy
    mix2.min_x, mix2.max_x, mix2.min_y, mix2.max_y, mix2.log_scale_factor, mix2.n_levels = v1.min_x, v1.max_x, v1.min_y, v1.max_y, v1.log_scale_factor, v1.n_levels
    ve, uve, le = OS.project_general(mix2, mp2, orb_search.PROJ_SIM3_DIR, sR12, t12f)
    a = OS.search_sim3_direction(K2, mp1["desc"], vd, uvd, ld, 7.5); b = OS.search_sim3_direction(K1, mp2["desc"], ve, uve, le, 7.5)
    exp = np.array([a[i] if a[i] >= 0 and b[a[i]] == i else -1 for i in range(K1.n)], np.int32)
    np.testing.assert_array_equal(m12, exp)
    assert counts[3] == int((exp >= 0).sum())



@pytest.mark.gpu
@pytest.mark.parametrize("between", [False, True])
def test_harness_tracking_chain(harness, tmp_path, oracle, between):
    """`harness track`: the Tracking thread's per-frame chain (TrackWithMotionModel + TrackLocalMap, lines included) driven from compiled
    C++ through lld_amd::TrackedFrame, several repeats on one handle - against the oracle's own run of the sequence; with and without the
    host fetching stage 1's record between the two calls."""
    import oracle_tracking as OT
    from lld_slam_amd import synth, tracking
    sc = synth.make_tracking_scene(8)
    repeats = 3
    nl = tracking.write_harness_scene(tmp_path / "in.bin", sc, repeats=repeats, download_between=between)
    p = run(harness, "track", tmp_path)
    assert p.returncode == 0, p.stderr
    g1, g2, ms = tracking.read_harness_result(tmp_path / "out.bin", sc["frame"].n, nl, repeats)
    e1, e2 = OT.track_frame(sc)
    for g, e in ((g1, e1), (g2, e2)):
        for k in ("kp_point_id", "kp_outlier", "ln_line_id", "ln_outlier"):
            np.testing.assert_array_equal(g[k], e[k], err_msg=k)
        for k in ("n_inliers", "n_edges", "n_search_first", "n_search", "used_wide", "n_points", "n_points_map", "n_lines_matched", "n_lines", "n_discarded"):
            assert g[k] == e[k], k
        np.testing.assert_allclose(g["pose_qt"], e["pose_qt"], rtol=1e-5, atol=1e-8)
    assert ms["total"].shape == (repeats,) and np.all(ms["total"] > 0)
