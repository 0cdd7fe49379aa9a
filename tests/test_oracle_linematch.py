"""Known-answer tests pinning oracle/lldo_linematch.cpp (stereo line association, TwoFrameLineMatcher + vgl): the restated
Eigen::ColPivHouseholderQR against numpy's least squares, vgl::TriangulateLine against lines whose 3D position is known by
construction, and the whole MatchLines against a naive numpy restatement.  No GPU."""
import numpy as np
import pytest

from lld_slam_amd import synth


def test_colpiv_qr_matches_numpy_lstsq(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        A = rng.normal(size=(3, 3)) * rng.uniform(0.1, 100, (1, 3)); b = rng.normal(size=3)
        r, x = oracle.colpiv_qr_solve(A, b)
        assert r == 3
        np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
        A2 = rng.normal(size=(3, 2)) * rng.uniform(0.1, 1000, (1, 2))
        r, x = oracle.colpiv_qr_solve(A2, b)
        assert r == 2
        np.testing.assert_allclose(x, np.linalg.lstsq(A2, b, rcond=None)[0], rtol=1e-9, atol=1e-12)
    # rank deficiency is detected (Eigen: |R_kk| <= eps*size*max|R_kk|), the dropped unknown is zeroed
    A = np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 6.0], [1.0, 0.0, 1.0]])
    r, x = oracle.colpiv_qr_solve(A, np.array([1.0, 2.0, 3.0]))
    assert r == 2 and np.count_nonzero(x == 0.0) >= 1


def test_triangulate_line_recovers_a_known_segment(oracle):
    fx, fy, cx, cy, bf = synth.KITTI_CAM
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]); b = bf / fx
    rng = np.random.default_rng(1)
    for _ in range(100):
        A = np.array([rng.uniform(-5, 5), rng.uniform(-2, 2), rng.uniform(4, 30)]); d = rng.normal(size=3); d[1] += np.sign(d[1]) * 0.5
        d /= np.linalg.norm(d); B = A + rng.uniform(0.5, 3.0) * d

        def px(X, s): return [fx * (X[0] - s) / X[2] + cx, fy * X[1] / X[2] + cy]
        kl1 = np.array(px(A, 0) + px(B, 0), np.float64); kl2 = np.array(px(A, b) + px(B, b), np.float64)
        ok, X0, ld, p1, p2 = oracle.line_pair_geometry(K, b, kl1, kl2)
        if not ok:
            continue                                                    # near-epipolar segment: the 0.975 gate
        # the direction matches up to sign, X0 lies on the line, the re-projected endpoints are A and B (float32 pixel rounding)
        assert abs(abs(ld @ d) - 1.0) < 1e-6
        assert np.linalg.norm(np.cross(X0 - A, d)) < 2e-2 * A[2]
        assert abs(X0 @ ld) < 1e-9                                      # third row of the system: X0 . dir = 0
        assert np.linalg.norm(p1 - A) < 5e-2 * A[2] and np.linalg.norm(p2 - B) < 5e-2 * B[2]
    # a horizontal segment is epipolar-degenerate: plane normals coincide -> rejected
    ok = oracle.line_pair_geometry(K, b, [100, 200, 300, 200], [80, 200, 280, 200])[0]
    assert not ok


def naive_match(s, tau, min_len):
    """TwoFrameLineMatcher::MatchLines with numpy linear algebra (np.linalg.solve / lstsq instead of the QR restatement)."""
    K, b = s["K"], s["b"]
    def leq(kl):
        l = K.T @ np.cross([kl[0], kl[1], 1.0], [kl[2], kl[3], 1.0]); return l / np.linalg.norm(l[:2])
    nL, nR = s["left"].shape[0], s["right"].shape[0]
    L = s["left"].astype(np.float64); R = s["right"].astype(np.float64)
    gate = np.zeros((nL, nR), np.uint8)
    eqL = [leq(k) for k in L]; eqR = [leq(k) for k in R]
    lenL = np.hypot(L[:, 0] - L[:, 2], L[:, 1] - L[:, 3]); lenR = np.hypot(R[:, 0] - R[:, 2], R[:, 1] - R[:, 3])
    for j in range(nL):
        for oi in range(nR):
            if s["left_octave"][j] != s["right_octave"][oi] or lenL[j] < min_len or lenR[oi] < min_len: continue
            n1, n2 = eqL[j], eqR[oi]
            if abs(n1 @ n2) / np.linalg.norm(n1) / np.linalg.norm(n2) > 0.975: continue
            d = np.cross(n1, n2); d /= np.linalg.norm(d)
            X0 = np.linalg.solve(np.stack([n1, n2, d]), np.array([0.0, n2 @ np.array([b, 0, 0]), 0.0]))
            if np.linalg.norm(X0) < 0.5: continue
            ok = True
            for e in (0, 2):
                M = np.stack([np.array([L[j, e], L[j, e + 1], 1.0]), -K @ d], 1)
                p = np.linalg.lstsq(M, K @ X0, rcond=None)[0][1]
                if (X0 + p * d)[2] < 0: ok = False
            gate[j, oi] = ok
    taken = np.zeros(nR, bool); out = -np.ones(nL, np.int64)
    for j in range(nL):
        best, bj = np.inf, -1
        for oi in range(nR):
            if taken[oi] or not gate[j, oi]: continue
            diff = (s["desc_left"][j] - s["desc_right"][oi]).astype(np.float32)
            dd = float(np.sqrt(np.sum(diff.astype(np.float64) ** 2)))
            if dd < best and dd < tau: best, bj = dd, oi
        if bj >= 0: taken[bj] = True
        out[j] = bj
    return out, gate


@pytest.mark.parametrize("seed", [0, 1])
def test_match_lines_equals_the_naive_restatement(oracle, seed):
    s = synth.make_stereo_lines(seed, 90, 80)
    m, d, gate = oracle.line_match_stereo(s["K"], s["b"], 2.0, 20, s["left"], s["left_octave"], s["desc_left"], s["right"], s["right_octave"],
                                          s["desc_right"], want_gate=True)
    m_py, gate_py = naive_match(s, 2.0, 20)
    np.testing.assert_array_equal(gate, gate_py)
    np.testing.assert_array_equal(m, m_py)
    assert (m >= 0).sum() >= 25 and 0.01 < gate.mean() < 0.5
