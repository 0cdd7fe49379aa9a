"""Two host threads, each with its own context, at the same time - the reference's Tracking and LocalMapping threads
(src/Tracking.cc, src/LocalMapping.cc:82 run concurrently): LocalBundleAdjustment on one, PoseOptimization + guided ORB search + the
line matchers on the other.  Every result must equal what the same call gives alone (the contract of include/lld_amd.h: re-entrant
per handle, one handle per host thread)."""
import threading

import numpy as np
import pytest

from lld_slam_amd import Context, Optimizer, ORBmatcher, Tracking, synth

pytestmark = pytest.mark.gpu


def test_tracking_and_local_mapping_threads_do_not_disturb_each_other(gpu_ctx, oracle):
    import oracle_orbsearch as OS
    w = synth.make_lba_small(41, n_free=8, n_fixed=2, n_points=600, n_lines=80)
    f = synth.make_pose_frame(61, n_points=400, n_lines=80)
    F = synth.make_orb_frame(71, 1500); q = synth.make_projection_queries(F, 71, 1200)
    P, L, FL = synth.make_line_track_scene(31, n_map=120, n_cur=160)
    ref_ba = oracle.local_ba(w)
    ref_pose = oracle.pose_opt(f, gamma=0.5)
    ref_n, ref_slot = OS.search_by_projection_map(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0, 0.8)
    ref_lines = oracle.line_track_match(P["K"], P["T_curr"], P["b"], P["thr_reproj_base"], P["md_thr"], P["sx"], P["sy"], L, FL)[0]
    errors = []
    stop = threading.Event()

    def local_mapping():
        try:
            ctx = Context(0)
            try:
                for _ in range(12):
                    g = Optimizer(ctx).LocalBundleAdjustment(w)
                    assert g.stats["chi2_final"] == pytest.approx(ref_ba.stats["chi2_final"], rel=1e-5)
                    np.testing.assert_array_equal(g.pt_obs_outlier, ref_ba.pt_obs_outlier)
                    np.testing.assert_allclose(g.cam_qt, ref_ba.cam_qt, rtol=1e-5, atol=1e-7)
            finally:
                ctx.close()
        except BaseException as e:                                              # noqa: BLE001 - reported by the main thread
            errors.append(("local mapping", repr(e)[:500]))
        finally:
            stop.set()

    def tracking():
        try:
            ctx = Context(0)
            try:
                n = 0
                while not stop.is_set() or n < 5:
                    g = Optimizer(ctx).PoseOptimization(f, gamma=0.5)
                    assert g.n_inliers == ref_pose.n_inliers
                    np.testing.assert_array_equal(g.pt_outlier, ref_pose.pt_outlier)
                    np.testing.assert_allclose(g.pose_qt, ref_pose.pose_qt, rtol=1e-5, atol=1e-7)
                    out = ORBmatcher(ctx, 0.8).SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0)
                    assert out.n_matches == ref_n
                    m, _ = Tracking(ctx, P["K"], P["b"], 1.0 / P["sx"], 1.0 / P["sy"], mdThr=P["md_thr"]).AddLinesFrom(L, P["T_curr"], P["thr_reproj_base"], FL)
                    np.testing.assert_array_equal(m, ref_lines)
                    n += 1
                    if n > 400: break
            finally:
                ctx.close()
        except BaseException as e:                                              # noqa: BLE001
            errors.append(("tracking", repr(e)[:500]))

    ta = threading.Thread(target=local_mapping); tb = threading.Thread(target=tracking)
    ta.start(); tb.start(); ta.join(timeout=300); tb.join(timeout=300)
    assert not ta.is_alive() and not tb.is_alive()
    assert not errors, errors
