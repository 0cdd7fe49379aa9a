"""The Tracking thread's per-frame sequence on one resident frame (lld_frame_*, lld_slam_amd/tracking.py): TrackWithMotionModel's matcher
(src/Tracking.cc:904) -> PoseOptimization (:937) -> outlier discard -> SearchLocalPoints (:1133) -> PoseOptimization (:1152), each stage
against the oracle ON THE INPUTS THE DEVICE CHAIN HANDED IT (indices bit-exact, poses to 1e-5 with identical outlier flags), and the
resident form against the per-call-upload form bit for bit."""
import numpy as np
import pytest

import oracle_orbsearch as OS
from lld_slam_amd import orb_search, synth
from lld_slam_amd.tracking import TrackedFrame
from test_gpu_orbsearch import expect_slots

pytestmark = pytest.mark.gpu


def run_chain(gpu_ctx, sc, resident):
    with TrackedFrame(gpu_ctx, sc["frame"], sc["cam"], resident=resident) as tf:
        p1 = tf.track_with_motion_model(sc["pose_guess"], sc["last"], sc["last_ids"], th=7.0)
        p2 = tf.track_local_map(p1, sc["map_points"], sc["map_ids"], th=1.0)
        return p1, p2, tf.stages, tf.kp_has.copy(), tf.kp_point.copy()


@pytest.mark.parametrize("scene", [0, 1, 2])
def test_tracking_sequence_stage_by_stage_against_the_oracle(gpu_ctx, oracle, scene):
    sc = synth.make_tracking_scene(scene)
    F = sc["frame"]
    p1, p2, st, has, point = run_chain(gpu_ctx, sc, True)
    # stage 1: projection of the last frame's points + search, on the predicted pose
    s = st["search_last_frame"]
    valid, uv, ur = OS.project_last_frame(s["view"], sc["last"])
    m = valid != 0
    np.testing.assert_array_equal(s["uvr"][m, :2], uv[m]); np.testing.assert_array_equal(s["uvr"][m, 2], ur[m])
    n_exp, slot = OS.search_by_projection_frame(F, sc["last"]["desc"], valid, uv, ur, sc["last"]["octave"], sc["last"]["angle"], sc["last"]["has_obs"],
                                                s["occupied"], 0, 7.0, True)
    assert s["out"].n_matches == n_exp and n_exp > 300
    np.testing.assert_array_equal(expect_slots(s["out"], s["occupied"]), slot)
    # stages 2 and 4: PoseOptimization on exactly the edges the device chain built
    for tag in ("pose_after_motion_model", "pose_after_local_map"):
        g, o = st[tag]["out"], oracle.pose_opt(st[tag]["problem"], 0.5)
        np.testing.assert_allclose(g.pose_qt, o.pose_qt, rtol=1e-5, atol=1e-8)
        np.testing.assert_array_equal(g.pt_outlier, o.pt_outlier)
        assert g.n_inliers == o.n_inliers
    # stage 3: frustum + local-map search on the pose of stage 2, skipping what the frame already holds
    s = st["search_local_points"]
    k, inv, uvr, lvl, vc = OS.is_in_frustum(s["view"], s["points"])
    np.testing.assert_array_equal(s["frustum"]["in_view"], inv)
    mm = inv != 0
    np.testing.assert_array_equal(s["frustum"]["proj_uvr"][mm], uvr[mm]); np.testing.assert_array_equal(s["frustum"]["level"][mm], lvl[mm])
    n_exp, slot = OS.search_by_projection_map(F, s["points"]["desc"], inv, uvr[:, :2], uvr[:, 2], lvl, vc, s["points"]["has_obs"], s["occupied"], 1.0, 0.8)
    assert s["out"].n_matches == n_exp and n_exp > 100
    np.testing.assert_array_equal(expect_slots(s["out"], s["occupied"]), slot)
    # the sequence did its job: more MapPoints after the local map than after the motion model, and a pose near the truth
    assert has.sum() > 500
    err0 = np.linalg.norm(np.asarray(sc["pose_guess"])[4:] - np.asarray(sc["pose_true"])[4:])
    err2 = np.linalg.norm(np.asarray(p2)[4:] - np.asarray(sc["pose_true"])[4:])
    assert err2 < 0.5 * err0, (err0, err2)


def test_resident_frame_equals_per_call_upload(gpu_ctx):
    sc = synth.make_tracking_scene(3)
    a = run_chain(gpu_ctx, sc, True); b = run_chain(gpu_ctx, sc, False)
    np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[3], b[3]); np.testing.assert_array_equal(a[4], b[4])
    for tag in ("search_last_frame", "search_local_points"):
        for f in ("match", "best_dist", "second_dist", "removed", "owner"):
            np.testing.assert_array_equal(getattr(a[2][tag]["out"], f), getattr(b[2][tag]["out"], f))


def test_frame_handle_edge_cases(gpu_ctx):
    """An empty frame, an empty query set, and a handle that outlives several calls with different occupancy."""
    F = synth.make_orb_frame(260, 300)
    T, mp = synth.make_local_map(F, 260, 200)
    view = orb_search.frame_view(T, synth.KITTI_CAM, F)
    with orb_search.ResidentFrame(gpu_ctx.lib, gpu_ctx.handle, F) as R:
        empty = {k: (v[:0] if k != "occupied" else v) for k, v in mp.items()}
        out, fr = R.search_local_points(view, empty, mp["occupied"])
        assert out.n_matches == 0 and out.match.shape == (0,)
        a, _ = R.search_local_points(view, mp, mp["occupied"])
        b, _ = orb_search.search_local_points(gpu_ctx.lib, gpu_ctx.handle, F, view, mp, mp["occupied"])
        np.testing.assert_array_equal(a.match, b.match); np.testing.assert_array_equal(a.owner, b.owner)
        full = np.ones(F.n, np.uint8)
        c, _ = R.search_local_points(view, mp, full)
        assert c.n_matches == 0                                   # every keypoint occupied: nothing to take
