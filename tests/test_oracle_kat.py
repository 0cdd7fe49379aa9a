"""Known-answer tests that pin the CPU oracle.

The reference has no tests or golden vectors and cannot be compiled here, so the restatement is checked against
independent mathematics: bit counting, the Huber closed form, scipy's matrix exponential, and central differences
of the restated error functions (g2o's own numeric-Jacobian definition, core/base_binary_edge.hpp:131-205).
"""
import numpy as np
import pytest
from scipy.linalg import expm

from lld_slam_amd import synth

CAM = synth.KITTI_CAM


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def _qt_to_T(O, qt):
    T = np.eye(4); T[:3, :3] = O.quat_to_R(qt[:4]); T[:3, 3] = qt[4:]
    return T


# ---------------------------------------------------------------- (1) DescriptorDistance
def test_descriptor_distance_matches_popcount(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.integers(0, 2 ** 32, 8, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 2 ** 32, 8, dtype=np.uint64).astype(np.uint32)
        ref = sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b))
        assert oracle.descriptor_distance(a, b) == ref
    z = np.zeros(8, np.uint32); o = np.full(8, 0xFFFFFFFF, np.uint32)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, o) == 256
    for bit in (0, 31, 32, 255):
        h = np.zeros(8, np.uint32); h[bit // 32] = np.uint32(1) << np.uint32(bit % 32)
        assert oracle.descriptor_distance(z, h) == 1
        assert oracle.descriptor_distance(o, h) == 255


# ---------------------------------------------------------------- (2) Huber closed form
@pytest.mark.parametrize("delta", [np.sqrt(5.991), np.sqrt(7.815), 0.5 * np.sqrt(7.815)])
def test_huber_closed_form(oracle, delta):
    d2 = delta * delta
    for e in (0.0, 0.3 * d2, d2, np.nextafter(d2, 10), 4 * d2, 1e6):
        rho = oracle.huber(delta, e)
        if e <= d2:
            assert rho[0] == e and rho[1] == 1.0 and rho[2] == 0.0
        else:
            assert rho[0] == pytest.approx(2 * delta * np.sqrt(e) - d2, rel=1e-15)
            assert rho[1] == pytest.approx(delta / np.sqrt(e), rel=1e-15)
    # continuity at the threshold
    assert oracle.huber(delta, np.nextafter(d2, 10))[0] == pytest.approx(d2, rel=1e-12)


# ---------------------------------------------------------------- (3) SE3 exp
def test_se3_exp_matches_expm(oracle):
    rng = np.random.default_rng(3)
    for _ in range(50):
        w = rng.normal(0, 0.5, 3); v = rng.normal(0, 1.0, 3)
        qt = oracle.se3_exp(np.concatenate([w, v]))          # reference ordering: (omega, upsilon)
        A = np.zeros((4, 4)); A[:3, :3] = _skew(w); A[:3, 3] = v
        np.testing.assert_allclose(_qt_to_T(oracle, qt), expm(A), atol=1e-12)
        assert qt[3] >= 0 and abs(np.linalg.norm(qt[:4]) - 1) < 1e-15


def test_se3_exp_small_angle_branch_is_literal(oracle):
    """theta < 1e-5: the reference uses R = V = I + Omega + Omega^2 (se3quat.h:237-243), not the Taylor series."""
    w = np.array([3e-6, -2e-6, 1e-6]); v = np.array([0.1, -0.2, 0.3])
    qt = oracle.se3_exp(np.concatenate([w, v]))
    Om = _skew(w); R = np.eye(3) + Om + Om @ Om
    T = _qt_to_T(oracle, qt)
    np.testing.assert_allclose(T[:3, 3], R @ v, rtol=0, atol=1e-16)
    np.testing.assert_allclose(T[:3, :3], R, atol=1e-11)      # through Quaterniond(R) + normalise


def test_quaternion_round_trip_all_branches(oracle):
    rng = np.random.default_rng(4)
    for k in range(200):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if k % 4 == 1: q[3] = 1e-3 * q[3]                    # small w -> trace <= 0 branches
        q /= np.linalg.norm(q)
        R = oracle.quat_to_R(q)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)
        q2 = oracle.quat_from_R(R)
        assert min(np.linalg.norm(q2 - q), np.linalg.norm(q2 + q)) < 1e-12


def test_se3_mul_and_map(oracle):
    rng = np.random.default_rng(5)
    a = oracle.se3_exp(rng.normal(0, 0.4, 6)); b = oracle.se3_exp(rng.normal(0, 0.4, 6))
    ab = oracle.se3_mul(a, b)
    np.testing.assert_allclose(_qt_to_T(oracle, ab), _qt_to_T(oracle, a) @ _qt_to_T(oracle, b), atol=1e-13)
    X = rng.normal(size=3)
    np.testing.assert_allclose(oracle.se3_map(a, X), (_qt_to_T(oracle, a) @ np.append(X, 1))[:3], atol=1e-14)


def test_converter_round_trip(oracle):
    from lld_slam_amd import host
    rng = np.random.default_rng(6)
    T = _qt_to_T(oracle, oracle.se3_exp(rng.normal(0, 0.5, 6))).astype(np.float32)
    qt = host.se3_from_tcw_f32(oracle.lib(), T)
    np.testing.assert_allclose(qt, synth._tcw_to_qt(T), atol=1e-15)
    T2 = host.se3_to_tcw_f32(oracle.lib(), qt)
    np.testing.assert_allclose(T2, T, atol=2e-7)
    assert T2[3].tolist() == [0, 0, 0, 1]


def test_inv_level_sigma2_is_float_arithmetic(oracle):
    from lld_slam_amd import host
    tab = host.orb_inv_level_sigma2(oracle.lib(), 1.2, 8)
    np.testing.assert_array_equal(tab, synth.inv_level_sigma2(1.2, 8))
    np.testing.assert_allclose(tab, 1.2 ** (-2.0 * np.arange(8)), rtol=1e-6)
    assert tab[3] != np.float32(1.2 ** -6)       # the cumulative float products are not the closed form


# ---------------------------------------------------------------- (4) Jacobians vs central differences
def _num_jac(f, n, delta=1e-6):
    cols = []
    for i in range(n):
        d = np.zeros(n); d[i] = delta
        cols.append((f(d) - f(-d)) / (2 * delta))
    return np.stack(cols, 1)


def _rand_pose(oracle, rng):
    return oracle.se3_exp(np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 1.0, 3)]))


@pytest.mark.parametrize("stereo", [True, False])
def test_point_edge_jacobians(oracle, stereo):
    rng = np.random.default_rng(10 + stereo)
    for _ in range(20):
        qt = _rand_pose(oracle, rng)
        Xc = np.array([rng.uniform(-3, 3), rng.uniform(-1, 1), rng.uniform(4, 40)])
        Xw = oracle.quat_to_R(qt[:4]).T @ (Xc - qt[4:])
        obs = np.array([600.0, 180.0, 590.0])
        e, Jp, Jc = oracle.edge_point(CAM, qt, Xw, obs, stereo)
        fp = lambda d: oracle.edge_point(CAM, qt, Xw + d, obs, stereo, jac=False)[0]
        fc = lambda d: oracle.edge_point(CAM, oracle.se3_oplus(qt, d), Xw, obs, stereo, jac=False)[0]
        # stereo residuals carry float32 noise from `const float invz` (hazard 2) -> ~1e-4 px; use a larger step
        step = 1e-3 if stereo else 1e-6
        tol = 2e-2 if stereo else 1e-5
        np.testing.assert_allclose(_num_jac(fp, 3, step), Jp, rtol=tol, atol=tol * 10)
        np.testing.assert_allclose(_num_jac(fc, 6, step), Jc, rtol=tol, atol=tol * 10)


def test_stereo_residual_has_float_cast(oracle):
    """EdgeStereoSE3ProjectXYZ::cam_project keeps invz and bf in float (types_six_dof_expmap.cpp:158-165)."""
    qt = np.array([0, 0, 0, 1, 0, 0, 0.0]); Xw = np.array([1.234567, -0.7654321, 13.37])
    obs = np.zeros(3)
    e, _, _ = oracle.edge_point(CAM, qt, Xw, obs, True)
    fx, fy, cx, cy, bf = CAM
    invz = np.float32(1.0 / Xw[2])
    u = Xw[0] * float(invz) * fx + cx
    v = Xw[1] * float(invz) * fy + cy
    ur = u - float(np.float32(bf) * invz)
    np.testing.assert_array_equal(-e, [u, v, ur])
    exact = fx * Xw[0] / Xw[2] + cx
    assert u != exact and abs(u - exact) < 1e-4
    # the pose-only twin multiplies bf (double) by the float invz
    e2, _ = oracle.edge_point_posonly(CAM, qt, Xw, obs, True)
    np.testing.assert_array_equal(-e2, [u, v, u - bf * float(invz)])


@pytest.mark.parametrize("stereo", [True, False])
def test_point_posonly_jacobian(oracle, stereo):
    rng = np.random.default_rng(20 + stereo)
    for _ in range(20):
        qt = _rand_pose(oracle, rng)
        Xc = np.array([rng.uniform(-3, 3), rng.uniform(-1, 1), rng.uniform(4, 40)])
        Xw = oracle.quat_to_R(qt[:4]).T @ (Xc - qt[4:])
        obs = np.array([600.0, 180.0, 590.0])
        e, Jc = oracle.edge_point_posonly(CAM, qt, Xw, obs, stereo)
        fc = lambda d: oracle.edge_point_posonly(CAM, oracle.se3_oplus(qt, d), Xw, obs, stereo)[0]
        step = 1e-3 if stereo else 1e-6
        tol = 2e-2 if stereo else 1e-5
        np.testing.assert_allclose(_num_jac(fc, 6, step), Jc, rtol=tol, atol=tol * 10)
        # the binary edge's pose block is the same function of (x,y,z)
        _, _, Jcb = oracle.edge_point(CAM, qt, Xw, obs, stereo)
        np.testing.assert_allclose(Jc, Jcb, rtol=1e-12, atol=1e-12)


def _rand_line_in_view(oracle, rng, qt):
    R = oracle.quat_to_R(qt[:4]); t = qt[4:]
    Mc = np.array([rng.uniform(-3, 3), rng.uniform(-1, 1), rng.uniform(6, 30)])
    dc = rng.normal(size=3); dc[2] *= 0.3; dc /= np.linalg.norm(dc)
    A = R.T @ (Mc - dc - t); B = R.T @ (Mc + dc - t)
    d = (B - A) / np.linalg.norm(B - A)
    X0 = A - (A @ d) * d
    fx, fy, cx, cy, bf = CAM
    def proj(P, bx):
        Pc = R @ P + t
        return np.array([fx * (Pc[0] + bx) / Pc[2] + cx, fx * Pc[1] / Pc[2] + cy])
    return X0, d, A, B, proj


@pytest.mark.parametrize("bx", [0.0, -CAM[4] / CAM[0]])
def test_line_edge_residual_and_jacobians(oracle, bx):
    rng = np.random.default_rng(30)
    for _ in range(20):
        qt = _rand_pose(oracle, rng)
        X0, d, A, B, proj = _rand_line_in_view(oracle, rng, qt)
        l5 = oracle.line_from_x0_dir(X0, d)
        X0b, db = oracle.line_to_x0_dir(l5)
        np.testing.assert_allclose(X0b, X0, atol=1e-12); np.testing.assert_allclose(db, d, atol=1e-12)
        seg_true = np.concatenate([proj(A, bx), proj(B, bx)])
        e, Jl, Jc, ok = oracle.edge_line(CAM, bx, qt, l5, seg_true)
        np.testing.assert_allclose(e, 0, atol=1e-9)          # endpoints on the projected line
        assert ok
        seg = seg_true + rng.normal(0, 2.0, 4)
        e, Jl, Jc, ok = oracle.edge_line(CAM, bx, qt, l5, seg)
        # residual = signed point-to-line distance in pixels
        pa, pb = proj(A, bx), proj(B, bx)
        n = np.array([pa[1] - pb[1], pb[0] - pa[0]]); n /= np.linalg.norm(n)
        np.testing.assert_allclose(np.abs(e), [abs(n @ (seg[:2] - pa)), abs(n @ (seg[2:] - pa))], rtol=1e-9, atol=1e-9)
        fc = lambda u: oracle.edge_line(CAM, bx, oracle.se3_oplus(qt, u), l5, seg)[0]
        fl = lambda u: oracle.edge_line(CAM, bx, qt, oracle.line_oplus(l5, u), seg)[0]
        np.testing.assert_allclose(_num_jac(fc, 6), Jc, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(_num_jac(fl, 4), Jl, rtol=2e-5, atol=2e-5)
        # pose-only twin: same residual and pose Jacobian for the fixed world endpoints X0, X0+dir
        e2, Jc2 = oracle.edge_line_posonly(CAM, bx, qt, X0, X0 + d, seg)
        np.testing.assert_allclose(e2, e, atol=1e-9); np.testing.assert_allclose(Jc2, Jc, rtol=1e-8, atol=1e-8)


def test_line_oplus_is_left_quaternion_update(oracle):
    rng = np.random.default_rng(31)
    X0 = np.array([1.0, 2.0, -0.5]); d = np.array([2.0, -1.0, 0.0]); d /= np.linalg.norm(d)
    X0 = X0 - (X0 @ d) * d
    l5 = oracle.line_from_x0_dir(X0, d)
    u = np.array([0.01, -0.02, 0.015, 0.25])
    l2 = oracle.line_oplus(l5, u)
    qr = np.array([u[0], u[1], u[2], np.sqrt(1 - u[:3] @ u[:3])])
    Rr = oracle.quat_to_R(qr)
    X0n, dn = oracle.line_to_x0_dir(l2)
    np.testing.assert_allclose(dn, Rr @ d, atol=1e-12)
    np.testing.assert_allclose(X0n, (np.linalg.norm(X0) + u[3]) * (Rr @ (X0 / np.linalg.norm(X0))), atol=1e-12)


def test_reproject_line_point_is_least_squares(oracle):
    rng = np.random.default_rng(32)
    fx, fy, cx, cy, bf = CAM
    K = np.array([[fx, 0, cx], [0, fx, cy], [0, 0, 1.0]])
    for _ in range(50):
        X0 = np.array([rng.uniform(-3, 3), rng.uniform(-1, 1), rng.uniform(4, 30)]); ld = rng.normal(size=3)
        px, py = rng.uniform(0, 1241), rng.uniform(0, 376)
        M = np.stack([np.array([px, py, 1.0]), -K @ ld], 1)
        sol = np.linalg.lstsq(M, K @ X0, rcond=None)[0]
        dpt, prm = oracle.reproject_line_point(X0, ld, px, py, fx, cx, cy)
        np.testing.assert_allclose([dpt, prm], sol, rtol=1e-9, atol=1e-9)


def test_line_depth_test_flags_lines_behind_camera(oracle):
    rng = np.random.default_rng(33)
    qt = _rand_pose(oracle, rng)
    X0, d, A, B, proj = _rand_line_in_view(oracle, rng, qt)
    seg = np.concatenate([proj(A, 0.0), proj(B, 0.0)])
    l5 = oracle.line_from_x0_dir(X0, d)
    assert oracle.edge_line(CAM, 0.0, qt, l5, seg)[3]
    # mirror the camera: rotate by pi about its x axis so the line is behind it
    flip = oracle.se3_mul(np.array([1.0, 0, 0, 0, 0, 0, 0]), qt)
    assert not oracle.edge_line(CAM, 0.0, flip, l5, seg)[3]


def test_line_threshold_is_looked_up_by_the_frame_index_of_the_line(oracle):
    """Optimizer.cc:893-898: `int idx = vnIndexLines[i]` is the line's index in the FRAME; the classification threshold comes from
    vnStereoLines[idx] although vnStereoLines holds one entry per EDGE (:643-648).  Known answer: a mono line (one edge) whose two
    detected end points sit 1.85 px off the projected 3D line has chi2 = gamma^2 * 2 * 1.85^2 = gamma^2 * 6.845, between the mono
    (5.991 gamma^2) and the stereo (7.815 gamma^2) threshold.  Which one it is compared with depends only on the stereo flag of the
    edge that happens to sit at position `frame index` of the edge list."""
    from lld_slam_amd import host, synth
    f = synth.make_pose_frame(11, n_points=600, n_lines=6, outlier_frac=0.0)
    gt = f.meta["gt_qt"]; Rcw, tcw = f.meta["gt_Rcw"], f.meta["gt_tcw"]
    fx, fy, cx, cy, bf = f.cam
    # noise-free point observations at the ground-truth pose: the optimum is the ground truth, the line edges barely matter
    Xc = f.pt_xw @ Rcw.T + tcw
    uvr = np.stack([fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy, fx * Xc[:, 0] / Xc[:, 2] + cx - bf / Xc[:, 2]], 1)
    # line 0: mono, both end points 1.85 px off its projection; lines 1..5: stereo, exact
    def project(P, bx=0.0):
        Pc = P @ Rcw.T + tcw
        return np.stack([fx * (Pc[:, 0] + bx) / Pc[:, 2] + cx, fx * Pc[:, 1] / Pc[:, 2] + cy], 1)       # line edges use fx on both axes
    A, B = f.ln_x0, f.ln_x0 + f.ln_dir
    a, b = project(A), project(B)
    left = np.concatenate([a, b], 1); ar, br = project(A, -bf / fx), project(B, -bf / fx); right = np.concatenate([ar, br], 1)
    n = np.stack([-(b - a)[:, 1], (b - a)[:, 0]], 1); n /= np.linalg.norm(n, axis=1, keepdims=True)
    left[0, :2] += 1.85 * n[0]; left[0, 2:] += 1.85 * n[0]
    right[0] = -1.0
    base = dict(cam=f.cam, pose_qt=gt, pt_xw=f.pt_xw, pt_uvr=uvr, pt_inv_sigma2=np.ones(f.n_points), ln_x0=f.ln_x0, ln_dir=f.ln_dir, ln_left=left, ln_right=right,
                ln_octave=np.zeros((6, 2), np.int32))
    # edge list: [L0 left (mono), L1 left, L1 right, L2 left, L2 right, ...] -> vnStereoLines = [0, 1, 1, 1, 1, ...]
    r = oracle.pose_opt(host.PoseFrame(**base), gamma=0.5)                               # frame index 0 -> entry 0 = mono -> 6.845 > 5.991: outlier
    assert r.ln_outlier.tolist() == [1, 0, 0, 0, 0, 0]
    r = oracle.pose_opt(host.PoseFrame(ln_frame_index=np.array([0, 1, 2, 3, 4, 5], np.int32), **base), gamma=0.5)
    assert r.ln_outlier.tolist() == [1, 0, 0, 0, 0, 0]                                    # NULL and the identity are the same thing
    r = oracle.pose_opt(host.PoseFrame(ln_frame_index=np.array([1, 2, 3, 4, 5, 6], np.int32), **base), gamma=0.5)
    assert r.ln_outlier.tolist() == [0, 0, 0, 0, 0, 0]                                    # frame index 1 -> entry 1 = L1's left edge = stereo -> 6.845 < 7.815: inlier
    r = oracle.pose_opt(host.PoseFrame(ln_frame_index=np.array([40, 41, 42, 43, 44, 45], np.int32), **base), gamma=0.5)
    assert r.ln_outlier.tolist() == [0, 0, 0, 0, 0, 0]                                    # beyond the edge list (undefined in the reference): stereo
    # and a stereo line read through the mono line's entry: line 1 gets the 1.85 px on its RIGHT edge (mvbOutlierLines[idx] keeps what the
    # line's last edge says, :899-907), and the frame indices put it on entry 0
    left2 = np.concatenate([a, b], 1); right2 = right.copy()
    nr = np.stack([-(br - ar)[:, 1], (br - ar)[:, 0]], 1); nr /= np.linalg.norm(nr, axis=1, keepdims=True)
    right2[1, :2] += 1.85 * nr[1]; right2[1, 2:] += 1.85 * nr[1]
    base2 = dict(base, ln_left=left2, ln_right=right2)
    r = oracle.pose_opt(host.PoseFrame(ln_frame_index=np.array([3, 0, 4, 5, 6, 7], np.int32), **base2), gamma=0.5)
    assert r.ln_outlier.tolist() == [0, 1, 0, 0, 0, 0]                                    # L1 (stereo) classified with the MONO threshold
    r = oracle.pose_opt(host.PoseFrame(**base2), gamma=0.5)
    assert r.ln_outlier.tolist() == [0, 0, 0, 0, 0, 0]
