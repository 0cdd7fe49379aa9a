"""Known-answer tests for the oracle's solver and protocols (SURVEY.md §8c items 5, 6 and the protocol rules)."""
import numpy as np
import pytest

from lld_slam_amd import synth

CAM = synth.KITTI_CAM


def _full_normal_equations(O, w, gamma=1.0):
    """Independent assembly of J, W, r for every edge of a window with numpy, using only the per-edge oracle calls."""
    nf = w.n_free_cams; n = 6 * nf + 3 * w.n_points + 4 * w.n_lines
    H = np.zeros((n, n)); b = np.zeros(n)
    d_mono = float(np.float32(np.sqrt(5.991))); d_st = float(np.float32(np.sqrt(7.815)))
    po = 6 * nf; lo = po + 3 * w.n_points

    def add(cols_l, Jl, cols_c, Jc, e, s, delta):
        chi = s * float(e @ e)
        wgt = 1.0 if chi <= delta * delta else delta / np.sqrt(chi)
        J = np.zeros((len(e), n)); J[:, cols_l] = Jl
        if cols_c is not None:
            J[:, cols_c] = Jc
        nonlocal H, b
        H += wgt * s * J.T @ J; b += -wgt * s * J.T @ e

    for p in range(w.n_points):
        for o in range(w.pt_obs_start[p], w.pt_obs_start[p + 1]):
            c = w.pt_obs_cam[o]; st = w.pt_obs_uvr[o, 2] >= 0
            e, Jp, Jc = O.edge_point(CAM, w.cam_qt[c], w.pt_xyz[p], w.pt_obs_uvr[o], st)
            add(slice(po + 3 * p, po + 3 * p + 3), Jp, slice(6 * c, 6 * c + 6) if c < nf else None, Jc, e,
                w.pt_obs_inv_sigma2[o], d_st if st else d_mono)
    bxr = -float(np.float32(CAM[4]) / np.float32(CAM[0]))
    for l in range(w.n_lines):
        l5 = O.line_from_x0_dir(w.line_x0[l], w.line_dir[l])
        for o in range(w.ln_obs_start[l], w.ln_obs_start[l + 1]):
            c = w.ln_obs_cam[o]; has_r = w.ln_obs_right[o, 0] >= 0
            for si, seg in enumerate((w.ln_obs_left[o], w.ln_obs_right[o])):
                if si == 1 and not has_r:
                    continue
                e, Jl, Jc, _ = O.edge_line(CAM, bxr if si else 0.0, w.cam_qt[c], l5, seg)
                info = gamma * gamma / (1.44 ** w.ln_obs_octave[o, si]) ** 2
                add(slice(lo + 4 * l, lo + 4 * l + 4), Jl, slice(6 * c, 6 * c + 6) if c < nf else None, Jc, e, info,
                    gamma * (d_st if has_r else d_mono))
    return H, b


def test_schur_solve_matches_dense_normal_equations(oracle):
    """(6) Schur complement + reduced solve + back-substitution == dense solve of the damped full system."""
    w = synth.make_lba_small(3, n_free=4, n_fixed=2, n_points=60, n_lines=15, mono_frac=0.2, mono_line_frac=0.2)
    H, b = _full_normal_equations(oracle, w)
    lam = 1e-5 * np.max(np.abs(np.diag(H)))
    st, x, bo, chi, md = oracle.ba_one_step(w, lam)
    assert st == 0
    np.testing.assert_allclose(md, np.max(np.abs(np.diag(H))), rtol=1e-12)
    np.testing.assert_allclose(bo, b, rtol=1e-9, atol=1e-9 * np.abs(b).max())
    xd = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
    np.testing.assert_allclose(x, xd, rtol=1e-6, atol=1e-8 * np.abs(xd).max())


def test_noise_free_window_has_zero_cost_and_lm_recovers_ground_truth(oracle):
    """(5) chi2 = 0 at the ground truth; LM from a perturbed start returns to it."""
    kw = dict(n_free=5, n_fixed=2, n_points=200, n_lines=40, outlier_frac=0.0, noise=0.0)
    w0 = synth.make_lba_small(7, pose_sigma=(0, 0), point_sigma=0, line_sigma=(0, 0), **kw)
    st, x, b, chi, md = oracle.ba_one_step(w0, 1.0)
    assert chi < 1e-3          # float32 storage of poses/points/observations leaves ~1e-3 px residuals
    w = synth.make_lba_small(7, **kw)
    r = oracle.local_ba(w)
    assert r.stats["chi2_final"] < 1e-2     # float32 rounding of the fixed poses and observations
    assert r.stats["n_pt_obs_outlier"] == 0 and r.stats["n_lines_removed"] == 0
    Rcw, tcw = w.meta["gt_Rcw"], w.meta["gt_tcw"]
    for c in range(w.n_free_cams):
        np.testing.assert_allclose(oracle.quat_to_R(r.cam_qt[c, :4]), Rcw[c], atol=2e-6)
        np.testing.assert_allclose(r.cam_qt[c, 4:], tcw[c], atol=2e-4)
    # fixed cameras are copied through untouched
    np.testing.assert_array_equal(r.cam_qt[w.n_free_cams:], w.cam_qt[w.n_free_cams:])


def test_local_ba_protocol_rules(oracle):
    w = synth.make_lba_small(11, mono_frac=0.1, mono_line_frac=0.1)
    r = oracle.local_ba(w)
    s = r.stats
    assert 1 <= s["lm_iterations"][0] <= 5 and 1 <= s["lm_iterations"][1] <= 15
    assert s["lm_trials"][0] >= s["lm_iterations"][0]
    assert s["n_pt_obs_outlier"] == int(r.pt_obs_outlier.sum()) > 0
    assert s["n_lines_removed"] == int(r.line_removed.sum())
    # removed lines keep their input parameters and report no outlier edges (GetLineData returns false)
    rem = r.line_removed.astype(bool)
    np.testing.assert_array_equal(r.line_x0[rem], w.line_x0[rem])
    for l in np.nonzero(rem)[0]:
        assert r.ln_edge_outlier[w.ln_obs_start[l]:w.ln_obs_start[l + 1]].sum() == 0
    # surviving lines come back as (X0 perpendicular to dir, unit dir)
    keep = ~rem
    np.testing.assert_allclose(np.linalg.norm(r.line_dir[keep], axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(np.sum(r.line_dir[keep] * r.line_x0[keep], 1), 0.0, atol=1e-9)
    # mono line observations never flag their (absent) right edge
    mono = w.ln_obs_right[:, 0] < 0
    assert r.ln_edge_outlier[mono, 1].sum() == 0
    # gross outliers injected by the generator are (almost all) caught
    assert s["n_pt_obs_outlier"] >= 0.03 * w.n_pt_obs


def test_local_ba_abort_flag(oracle):
    w = synth.make_lba_small(12)
    r = oracle.local_ba(w, abort=True)
    assert r.stats["aborted"] == 1 and r.stats["lm_iterations"] == [0, 0]
    np.testing.assert_array_equal(r.cam_qt, w.cam_qt); np.testing.assert_array_equal(r.pt_xyz, w.pt_xyz)


def test_local_ba_is_deterministic(oracle):
    w = synth.make_lba_small(13)
    a = oracle.local_ba(w); b = oracle.local_ba(w)
    np.testing.assert_array_equal(a.cam_qt, b.cam_qt); assert a.stats == b.stats


def test_pose_optimization_recovers_pose(oracle):
    f = synth.make_pose_frame(0, n_points=300, n_lines=60)
    r = oracle.pose_opt(f, gamma=0.5)
    gt = f.meta["gt_qt"]
    assert np.linalg.norm(f.pose_qt[4:] - gt[4:]) > 0.1
    assert np.linalg.norm(r.pose_qt[4:] - gt[4:]) < 0.03
    assert min(np.linalg.norm(r.pose_qt[:4] - gt[:4]), np.linalg.norm(r.pose_qt[:4] + gt[:4])) < 2e-3
    assert r.n_inliers == f.n_points - int(r.pt_outlier.sum())
    assert 0.05 * f.n_points < r.pt_outlier.sum() < 0.2 * f.n_points     # ~10 % injected outliers
    assert r.lm_iterations <= 40


def test_pose_optimization_needs_three_points(oracle):
    f = synth.make_pose_frame(1, n_points=2, n_lines=5)
    r = oracle.pose_opt(f)
    assert r.n_inliers == 0
    np.testing.assert_allclose(r.pose_qt, f.pose_qt)


def test_pose_optimization_mixed_mono_and_stereo(oracle):
    f = synth.make_pose_frame(2, n_points=200, n_lines=40, mono_frac=0.3, mono_line_frac=0.3)
    r = oracle.pose_opt(f)
    gt = f.meta["gt_qt"]
    assert np.linalg.norm(r.pose_qt[4:] - gt[4:]) < 0.05


def test_order_sensitivity_is_the_noise_floor(oracle):
    """The reference iterates MapPoint::GetObservations(), a std::map<KeyFrame*, size_t>, i.e. in ADDRESS order (SURVEY.md
    hazard 13): the order of a point's edges changes from run to run.  Re-ordering them here moves the oracle's own result
    by up to a few 1e-6 relative on the weakest landmarks of an LBA-A window while chi2 and poses move by < 1e-8.  This is
    the floor any other summation order (the GPU's) is compared against in tests/test_gpu_ba.py."""
    import copy
    w = synth.make_lba_a(3)
    rng = np.random.default_rng(1)
    idx = np.arange(w.n_pt_obs)
    for p in range(w.n_points):
        s, e = w.pt_obs_start[p], w.pt_obs_start[p + 1]
        idx[s:e] = s + rng.permutation(e - s)
    w2 = copy.deepcopy(w)
    w2.pt_obs_cam = w.pt_obs_cam[idx]; w2.pt_obs_uvr = w.pt_obs_uvr[idx]; w2.pt_obs_inv_sigma2 = w.pt_obs_inv_sigma2[idx]
    a = oracle.local_ba(w); b = oracle.local_ba(w2.normalise())
    np.testing.assert_array_equal(a.pt_obs_outlier[idx], b.pt_obs_outlier)
    np.testing.assert_array_equal(a.line_removed, b.line_removed)
    assert abs(a.stats["chi2_final"] - b.stats["chi2_final"]) <= 1e-7 * a.stats["chi2_final"]
    np.testing.assert_allclose(a.cam_qt, b.cam_qt, rtol=0, atol=1e-6)
    rel = np.linalg.norm(a.pt_xyz - b.pt_xyz, axis=1) / np.linalg.norm(a.pt_xyz, axis=1)
    assert rel.max() < 1e-4


def test_global_ba_protocol_rules(oracle):
    """Optimizer::BundleAdjustment (src/Optimizer.cc:321-559): one optimize(n) call, nothing classified or erased, line edges with
    identity information and delta thHuber3D/2, point kernels only when bRobust."""
    w = synth.make_lba_small(30, n_free=7, n_fixed=1, n_points=250, n_lines=40, outlier_frac=0.1)
    g = oracle.local_ba(w, protocol=1, its_round1=6)
    assert g.stats["lm_iterations"][1] == 0 and g.stats["lm_trials"][1] == 0 and 1 <= g.stats["lm_iterations"][0] <= 6
    assert g.stats["chi2_final"] == g.stats["chi2_round1"]
    assert not g.pt_obs_outlier.any() and not g.ln_edge_outlier.any() and not g.line_removed.any()
    assert g.stats["n_pt_obs_outlier"] == 0 and g.stats["n_lines_removed"] == 0
    np.testing.assert_array_equal(g.cam_qt[w.n_free_cams:], w.cam_qt[w.n_free_cams:])
    # gamma and ln_filter play no role
    g2 = oracle.local_ba(w, gamma=0.3, protocol=1, its_round1=6, ln_filter=99)
    np.testing.assert_array_equal(g.cam_qt, g2.cam_qt); np.testing.assert_array_equal(g.line_x0, g2.line_x0)
    # bRobust = false is a different (plain least squares) problem: gross outliers pull harder
    g3 = oracle.local_ba(w, protocol=1, its_round1=6, robust_points=0)
    assert g3.stats["chi2_final"] > g.stats["chi2_final"] and not np.allclose(g3.cam_qt, g.cam_qt)
    # more iterations never raise the cost (LM accepts only decreasing steps)
    g10 = oracle.local_ba(w, protocol=1, its_round1=12)
    assert g10.stats["chi2_final"] <= g.stats["chi2_final"] * (1 + 1e-12)
    # and it is not the local protocol
    loc = oracle.local_ba(w)
    assert loc.stats["lm_iterations"][1] > 0


def test_abort_after_k_trials_hook(oracle):
    """lld_ba_params.abort_after_trials: the stop flag counts as raised right after the k-th LM trial (both rounds counted) and stays up.
    Reference semantics: terminate() ends the trial loop (levenberg.cpp:149) and the iteration loop (sparse_optimizer.cpp:376);
    a flag up after optimize(5) skips classification + round 2 (Optimizer.cc:1230-1232) but NOT the final classification / write-back
    (:1278-1386).  `aborted` = the flag at the protocol's last poll."""
    w = synth.make_lba_small(12)
    full = oracle.local_ba(w)
    t1, t2 = full.stats["lm_trials"]
    assert full.stats["aborted"] == 0 and t1 >= 5 and t2 >= 2
    # k = 1: one trial of one iteration, then everything stops; the erase lists come from the round-1 state with kernels on
    r = oracle.local_ba(w, abort_after_trials=1)
    assert r.stats["lm_trials"] == [1, 0] and r.stats["lm_iterations"] == [1, 0] and r.stats["aborted"] == 1
    assert r.stats["n_lines_removed"] == 0 and r.line_removed.sum() == 0            # DisableOutliers never ran
    assert r.stats["chi2_final"] == r.stats["chi2_round1"]
    assert not np.array_equal(r.cam_qt[:w.n_free_cams], w.cam_qt[:w.n_free_cams])   # the map IS updated (unlike the abort before the start)
    assert r.pt_obs_outlier.sum() > 0
    # k in the middle of round 1
    r = oracle.local_ba(w, abort_after_trials=3)
    assert r.stats["lm_trials"] == [3, 0] and r.stats["aborted"] == 1 and r.stats["lm_iterations"][0] <= 3
    # k = the last trial of round 1: optimize(5) ends on its own, the check after it sees the flag
    r = oracle.local_ba(w, abort_after_trials=t1)
    assert r.stats["lm_trials"] == [t1, 0] and r.stats["lm_iterations"] == [full.stats["lm_iterations"][0], 0] and r.stats["aborted"] == 1
    assert r.stats["chi2_round1"] == full.stats["chi2_round1"]
    # k = first trial of round 2: classification and line removal happened, round 2 is one iteration of one trial
    r = oracle.local_ba(w, abort_after_trials=t1 + 1)
    assert r.stats["lm_trials"] == [t1, 1] and r.stats["lm_iterations"] == [full.stats["lm_iterations"][0], 1] and r.stats["aborted"] == 1
    np.testing.assert_array_equal(r.line_removed, full.line_removed)
    # k = the very last trial: nothing is cut short, but the flag is up at the last poll
    r = oracle.local_ba(w, abort_after_trials=t1 + t2)
    assert r.stats["lm_trials"] == [t1, t2] and r.stats["aborted"] == 1
    np.testing.assert_array_equal(r.cam_qt, full.cam_qt)
    # k beyond the end: never raised
    r = oracle.local_ba(w, abort_after_trials=t1 + t2 + 1)
    assert r.stats == full.stats


def test_landmark_inverse_rounding_moves_nearly_singular_points(oracle):
    """Two-view points under fixed cameras, a third of the observations monocular: Hll is nearly singular and the point depends on the
    ROUNDING of (Hll + lambda I)^-1.  Gauss-Jordan (default, standing in for Eigen's MatrixXd::inverse(), block_solver.hpp:391) against
    the same inverse through a Cholesky factor: equal sets and chi2, a handful of points apart by up to 1e-3, the rest untouched.  This
    spread is the allowance the GPU parity test grants such points (tests/test_gpu_ba.py::test_nearly_singular_landmark_blocks)."""
    w = synth.make_ba_window(n_free=0, n_fixed=3, n_points=400, obs_per_point=2, n_lines=5, obs_per_line=1, seed=48, outlier_frac=0.5, mono_frac=0.3, noise=1.0)
    a = oracle.local_ba(w)
    try:
        oracle.set_landmark_inverse(1)
        b = oracle.local_ba(w)
    finally:
        oracle.set_landmark_inverse(0)
    np.testing.assert_array_equal(a.pt_obs_outlier, b.pt_obs_outlier)
    assert a.stats["chi2_final"] == pytest.approx(b.stats["chi2_final"], rel=1e-5)
    r = np.linalg.norm(a.pt_xyz - b.pt_xyz, axis=1) / np.maximum(np.linalg.norm(a.pt_xyz, axis=1), 1e-3)
    assert 1e-5 < r.max() < 1e-2 and (r > 1e-5).sum() <= 8 and np.median(r) < 1e-9
    # a well-conditioned window does not care
    w = synth.make_lba_small(3)
    a = oracle.local_ba(w)
    try:
        oracle.set_landmark_inverse(1)
        b = oracle.local_ba(w)
    finally:
        oracle.set_landmark_inverse(0)
    np.testing.assert_allclose(a.pt_xyz, b.pt_xyz, rtol=1e-8, atol=1e-10); assert a.stats["lm_trials"] == b.stats["lm_trials"]
