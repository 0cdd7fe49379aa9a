"""CPU: the restatement of Tracking::AddLinesFrom (oracle/lldo_linematch.cpp) against an independent numpy restatement of the same
reference lines, plus known answers for GetHoughCoordinates."""
import numpy as np
import pytest

from lld_slam_amd import synth


def hough_naive(leq, sx, sy, step_dist=3, step_ang=3):
    """src/LineMatching.cc:63-152 read line by line."""
    l = np.array(leq, float); l[0] /= sx; l[1] /= sy
    l = l / np.hypot(l[0], l[1])
    if l[1] < 0:
        l = -l
    dl = abs(l[2] / np.sqrt(2.0)) * 50
    di = int(np.floor(dl + 0.5)); di = max(min(di, 49), 0)
    sd = 1 if dl - di < 0 else -1
    al = np.arctan2(l[1], l[0]) / 3.14159265 * 50
    ai = int(np.floor(al + 0.5)); ai = max(min(ai, 49), 0)
    sa = 1 if al - ai < 0 else -1
    ang = []
    amax = max(ai, ai + sa)
    for i in range(amax, amax + step_ang):
        ang.append((i + 50 if i < 0 else i) % 50)
    amin = min(ai, ai + sa)
    for i in range(amin, amin - step_ang, -1):
        ang.append((i + 50 if i < 0 else i) % 50)
    dist = []
    dmax = max(di, di + sd)
    for i in range(dmax, dmax + step_dist):
        if 0 <= i < 49:
            dist.append(i)
    dmin = min(di, di + sd)
    for i in range(dmin, dmin - step_dist, -1):
        if 0 <= i < 49:
            dist.append(i)
    return dist, ang, di, ai


def track_naive(P, L, F, monocular=False, use_grid=True):
    K, T = P["K"], P["T_curr"]; R, t = T[:3, :3], T[:3, 3]
    tr = t + R @ np.array([P["b"], 0, 0])
    n_map, n_cur = L["X0"].shape[0], F["left_lines"].shape[0]
    ll = F["left_lines"].astype(np.float64)
    grid = {}
    for si in range(n_cur):
        leq = np.cross([ll[si, 0], ll[si, 1], 1.0], [ll[si, 2], ll[si, 3], 1.0])
        _, _, di, ai = hough_naive(leq, P["sx"], P["sy"], 0, 0)
        grid.setdefault((di, ai), []).append(si)
    occ = F["occupied"].astype(bool).copy()
    matches = -np.ones(n_map, np.int64)

    def img_line(tt, X0, d):
        a = K @ (R.T @ (X0 - tt)); b = K @ (R.T @ (X0 + d - tt))
        l = np.cross(a, b)
        return l / np.hypot(l[0], l[1])
    for i in range(n_map):
        if L["skip"][i]:
            continue
        lleft = img_line(t, L["X0"][i], L["dir"][i]); lright = img_line(tr, L["X0"][i], L["dir"][i])
        if use_grid:
            dist, ang, _, _ = hough_naive(lleft, P["sx"], P["sy"])
            cand = sorted({si for a in ang for d in dist for si in grid.get((d, a), [])})
        else:
            cand = range(n_cur)
        md, mid = 1e10, -1
        for si in cand:
            if occ[si]:
                continue
            ri = F["line_matches"][si]
            if ri < 0 and not monocular:
                continue
            if (R.T @ (L["X1"][i] - t))[2] < 0 or (R.T @ (L["X2"][i] - t))[2] < 0:
                continue
            thr = P["thr_reproj_base"] * 1.44 ** int(F["left_octave"][si])
            se = abs(ll[si, 0] * lleft[0] + ll[si, 1] * lleft[1] + lleft[2]) + abs(ll[si, 2] * lleft[0] + ll[si, 3] * lleft[1] + lleft[2])
            se2 = 0.0
            if not monocular:
                kr = F["right_lines"][ri].astype(np.float64)
                se2 = abs(kr[0] * lright[0] + kr[1] * lright[1] + lright[2]) + abs(kr[2] * lright[0] + kr[3] * lright[1] + lright[2])
            if se > thr or se2 > thr:
                continue
            df = L["desc"][i] - F["desc"][si]
            cd = np.sqrt(np.sum(df.astype(np.float64) ** 2))
            if cd < md:
                md, mid = cd, si
        if md > P["md_thr"] or mid < 0:
            continue
        occ[mid] = True; matches[i] = mid
    return matches


def test_hough_coordinates_known_answers(oracle):
    sx, sy = 1.0 / 1241.0, 1.0 / 376.0
    # the image diagonal y = (376/1241) x: normalised coordinates make it the line x' = y', angle 3*pi/4 -> ang cell 37/38, distance 0
    d, a = oracle.hough_coordinates([376.0, -1241.0, 0.0], sx, sy)
    assert set(d) <= set(range(0, 4)) and 0 in d
    assert all(0 <= x < 50 for x in a) and len(a) == 6 and len(set(a)) == 6
    rng = np.random.default_rng(5)
    for _ in range(300):
        p, q = rng.uniform(0, 1241, 2), rng.uniform(0, 376, 2)
        leq = np.cross([p[0], q[0], 1.0], [p[1], q[1], 1.0])
        dn, an, _, _ = hough_naive(leq, sx, sy)
        d, a = oracle.hough_coordinates(leq, sx, sy)
        assert list(d) == dn and list(a) == an
    # the angle window wraps around the grid, the last distance row is never listed (i < dist_cell_num - 1)
    d, a = oracle.hough_coordinates([1e-9, 1.0, -0.999 * 376.0], sx, sy)
    assert 49 not in d


@pytest.mark.parametrize("scene,kw", [(0, {}), (1, dict(monocular=True)), (2, dict(use_grid=False)), (3, dict(use_grid=False, monocular=True))])
def test_add_lines_from_matches_the_naive_restatement(oracle, scene, kw):
    P, L, F = synth.make_line_track_scene(scene, n_map=120, n_cur=150)
    m, d = oracle.line_track_match(P["K"], P["T_curr"], P["b"], P["thr_reproj_base"], P["md_thr"], P["sx"], P["sy"], L, F, **kw)
    ref = track_naive(P, L, F, **kw)
    np.testing.assert_array_equal(m, ref)
    assert (m >= 0).sum() > (10 if kw.get("use_grid", True) else 40)       # the scene exercises the accept path
    assert len(set(m[m >= 0])) == (m >= 0).sum()                            # a frame line is given away once
    assert not np.any(F["occupied"][m[m >= 0]].astype(bool))


def test_grid_cells_and_rivals(oracle):
    P, L, F = synth.make_line_track_scene(4, n_map=200, n_cur=220)
    cells = oracle.line_hough_cells(F["left_lines"], P["sx"], P["sy"])
    assert cells.min() >= 0 and cells.max() < 2500
    m_grid, _ = oracle.line_track_match(P["K"], P["T_curr"], P["b"], P["thr_reproj_base"], P["md_thr"], P["sx"], P["sy"], L, F)
    m_all, _ = oracle.line_track_match(P["K"], P["T_curr"], P["b"], P["thr_reproj_base"], P["md_thr"], P["sx"], P["sy"], L, F, use_grid=False)
    # a map line that projects next to its frame line shares (or neighbours) its Hough cell: the grid loses few of the brute-force matches
    both = (m_all >= 0)
    assert both.sum() > 60 and np.mean(m_grid[both] == m_all[both]) > 0.6


# ---------------------------------------------------------------- Tracking::MatchLinesLastKF / vgl::MultiTriangulateLine
def test_multi_triangulate_line_against_numpy(oracle):
    """Direction = right singular vector of the smallest singular value (numpy SVD, sign made canonical), X0 = least-squares point of
    the planes (numpy lstsq) minus its component along the direction; degenerate inputs refused like the reference."""
    rng = np.random.default_rng(11)
    for trial in range(40):
        X0t = rng.normal(0, 3, 3) + np.array([0, 0, 12.0]); dt = rng.normal(size=3); dt /= np.linalg.norm(dt)
        Ts, ls = [], []
        for v in range(4):
            T = np.eye(4); T[:3, :3] = synth._rodrigues(rng.normal(0, 0.08, 3)); T[:3, 3] = rng.normal(0, 0.8, 3)
            a = T[:3, :3].T @ (X0t - T[:3, 3]); bb = T[:3, :3].T @ (X0t + dt - T[:3, 3])
            l = np.cross(a, bb) + rng.normal(0, 1e-3, 3); l /= np.hypot(l[0], l[1])
            Ts.append(T); ls.append(l)
        ok, x0, dr = oracle.multi_triangulate_line(Ts, ls)
        N = np.stack([T[:3, :3] @ (l / np.linalg.norm(l)) for T, l in zip(Ts, ls)])
        par = any(abs(N[0] @ N[i]) / np.linalg.norm(N[0]) / np.linalg.norm(N[i]) > 0.975 for i in range(1, 4))
        assert ok == (not par)
        if not ok:
            continue
        v = np.linalg.svd(N)[2][2]; v = v * np.sign(v[np.argmax(np.abs(v))])
        bvec = np.array([N[i] @ Ts[i][:3, 3] for i in range(4)])
        xl = np.linalg.lstsq(N, bvec, rcond=None)[0]; xl = xl - (xl @ v) * v
        np.testing.assert_allclose(dr, v, atol=1e-9)
        np.testing.assert_allclose(x0, xl, rtol=1e-6, atol=1e-6)
        assert abs(abs(dr @ dt) - 1) < 1e-2                                   # and it is the line that was projected
    assert oracle.multi_triangulate_line(Ts[:2], ls[:2])[0] is False            # fewer than three views (src/vgl.cc:32-35)


def test_match_lines_last_kf_recovers_the_shared_lines(oracle):
    P, cur, last, truth = synth.make_two_frame_lines(0)
    for use_grid in (True, False):
        m, cre, x0, dr = oracle.line_match_last_frame(P["K"], P["T_curr"], P["T_last"], P["b"], P["thr_reproj_base"], P["md_thr"], P["sx"], P["sy"],
                                                      cur, last, use_grid)
        hit = m >= 0
        assert hit.sum() > 50 and np.mean(m[hit] == truth["last_of_cur"][hit]) > 0.97
        assert not np.any(hit & cur["occupied"].astype(bool)) and not np.any(hit & (cur["line_matches"] < 0))
        assert not np.any(last["skip"][m[hit]].astype(bool)) and np.all(last["line_matches"][m[hit]] >= 0)
        assert cre.sum() > 20 and np.all(cre[~hit] == 0)
        ok = cre.astype(bool)
        # the created lines are the 3D segments that were projected: direction parallel, X0 on the line
        src = truth["src"][ok]; A, B = truth["Aw"][src], truth["Bw"][src]
        dt = (B - A) / np.linalg.norm(B - A, axis=1, keepdims=True)
        assert np.median(np.abs(np.abs(np.sum(dr[ok] * dt, axis=1)) - 1)) < 1e-3
        off = x0[ok] - A; off -= np.sum(off * dt, axis=1, keepdims=True) * dt
        assert np.median(np.linalg.norm(off, axis=1)) < 0.2
        np.testing.assert_allclose(np.sum(x0[ok] * dr[ok], axis=1), 0, atol=1e-6)         # X0 is the point closest to the origin


def test_degenerate_keyline_does_not_index_outside_the_grid(oracle):
    """Coincident end points: no line equation. The oracle (and the device) file it as the line y = 0: cell 25."""
    P, L, F = synth.make_line_track_scene(11, n_map=60, n_cur=80)
    F["left_lines"][::7, 2:] = F["left_lines"][::7, :2]
    L["dir"][::5] = 0.0
    cells = oracle.line_hough_cells(F["left_lines"], P["sx"], P["sy"])
    assert np.all(cells[::7] == 25) and cells.min() >= 0 and cells.max() < 2500
    m, d = oracle.line_track_match(P["K"], P["T_curr"], P["b"], P["thr_reproj_base"], P["md_thr"], P["sx"], P["sy"], L, F)
    assert m.shape == (60,)
