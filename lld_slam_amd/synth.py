"""Deterministic synthetic inputs for the configs of BASELINE.json / SURVEY.md §8(d).

The reference ships no data (KITTI, LBD detections and the ORB vocabulary are absent), so the parity tests and
the bench run on synthetic keyframe graphs of the named shapes:

  LBA-B  50 free + 10 fixed KFs, 10 000 points x 6 stereo obs, 2 000 lines x 5 KFs x (left+right)   = 80 000 edges
  LBA-A  20 free +  5 fixed KFs,  5 000 points x 6,            1 000 lines x 5 x 2                   = 40 000 edges
  PO     1 frame, 1 000 stereo point matches + 200 stereo lines (400 line edges), gamma 0.5
  MATCH  2 000 x 2 000 ORB (256 bit) + 300 x 300 LBD (72 x f32)

States are stored float32 and widened, mirroring the reference's float32 map (src/Converter.cc).
Random numbers come from numpy's PCG64 seeded per problem id; generation is pure numpy (no reference code).
"""
from __future__ import annotations

import numpy as np

from . import abi
from .host import PoseFrame, Window

# KITTI04-12_LBD.yaml:8-25, rounded to float32 because KeyFrame::fx/fy/cx/cy/mbf are floats.
_F32 = lambda v: float(np.float32(v))
KITTI_CAM = (_F32(707.0912), _F32(707.0912), _F32(601.8873), _F32(183.1104), _F32(379.8145))
IMG_W, IMG_H = 1241.0, 376.0

SEED_LBA_B = 0xBA5E0000
SEED_LBA_A = 0xBA5EA000
SEED_PO = 0x90530000
SEED_MATCH = 0x0AB00000


def inv_level_sigma2(scale=1.2, n=8):
    """mvInvLevelSigma2 (src/ORBextractor.cc:416-430) in float arithmetic (numpy float32)."""
    sf = np.ones(n, np.float32); s2 = np.ones(n, np.float32)
    for i in range(1, n):
        sf[i] = np.float32(sf[i - 1] * np.float32(scale)); s2[i] = np.float32(sf[i] * sf[i])
    return (np.float32(1.0) / s2).astype(np.float32)


# ------------------------------------------------------------------ small SE3 helpers (generator only)
def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3) + _skew(w)
    K = _skew(w / th)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _tcw_to_qt(T):
    """Converter::toSE3Quat restated in numpy for the generator (float32 4x4 -> q,t)."""
    T = np.asarray(T, np.float32).astype(np.float64)
    R = T[:3, :3]; t = T[:3, 3]
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0); w = 0.5 * s; s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    else:
        i = 0
        if R[1, 1] > R[0, 0]: i = 1
        if R[2, 2] > R[i, i]: i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4); q[i] = 0.5 * s; s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0: q = -q
    q = q / np.sqrt(np.sum(q * q))
    return np.concatenate([q, t])


def _project(cam, Rcw, tcw, X, bx=0.0):
    """Pinhole projection of world points X [N,3] into one camera; returns u, v, z."""
    Xc = X @ Rcw.T + tcw
    z = Xc[:, 2]
    u = cam[0] * (Xc[:, 0] + bx) / z + cam[2]
    v = cam[1] * Xc[:, 1] / z + cam[3]
    return u, v, z


def _trajectory(rng, n, origin):
    """Forward motion along +z, 1 m steps, yaw random walk sigma 1 deg, lateral jitter 5 cm.  Returns Rcw, tcw lists."""
    yaw = np.cumsum(rng.normal(0, np.deg2rad(1.0), n))
    Rs, ts = [], []
    pos = np.array(origin, float)
    for i in range(n):
        Rwc = _rodrigues(np.array([0, yaw[i], 0.0]))
        p = pos + np.array([rng.normal(0, 0.05), rng.normal(0, 0.05), 0.0])
        Rs.append(Rwc.T); ts.append(-Rwc.T @ p)
        pos = pos + Rwc @ np.array([0, 0, 1.0])
    return np.array(Rs), np.array(ts)


def _f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


def make_ba_window(n_free=50, n_fixed=10, n_points=10000, obs_per_point=6, n_lines=2000, obs_per_line=5,
                   seed=SEED_LBA_B, outlier_frac=0.05, mono_frac=0.0, mono_line_frac=0.0, cam=KITTI_CAM,
                   pose_sigma=(0.5, 0.05), point_sigma=0.10, line_sigma=(1.0, 0.05), noise=1.0) -> Window:
    rng = np.random.default_rng(seed)
    n_cams = n_free + n_fixed
    fx, fy, cx, cy, bf = cam
    b = bf / fx
    # trajectory: the n_fixed poses precede the window; camera index order = free first, then fixed
    Rt, tt = _trajectory(rng, n_cams, origin=(20.0, -10.0, 5.0))
    order = list(range(n_fixed, n_cams)) + list(range(0, n_fixed))       # cam index -> trajectory index
    traj_of_cam = np.array(order)
    Rcw = Rt[traj_of_cam]; tcw = tt[traj_of_cam]
    inv_s2 = inv_level_sigma2().astype(np.float64)

    def visible(X, need_right=True):
        Xc = np.matmul(Rcw, X.T) + tcw[:, :, None]                           # [cams, 3, n]
        z = Xc[:, 2, :].T
        with np.errstate(divide='ignore', invalid='ignore'):
            u = fx * Xc[:, 0, :].T / z + cx
            v = fy * Xc[:, 1, :].T / z + cy
            ok = (z > 1.0) & (u >= 0) & (u < IMG_W) & (v >= 0) & (v < IMG_H)
            if need_right:
                ok &= (u - bf / z) >= 0
        return ok

    def pick_nearest(vis, k_seed, n_obs):
        """For every candidate row: the n_obs visible cameras nearest (in trajectory index) to its seeding camera,
        ascending camera index; rows with fewer than n_obs visible cameras are dropped.  Returns (row ids, cams)."""
        ok = vis.sum(1) >= n_obs
        rows = np.nonzero(ok)[0]
        if rows.size == 0:
            return rows, np.zeros((0, n_obs), np.int64)
        dist = np.abs(traj_of_cam[None, :] - traj_of_cam[k_seed[rows], None]).astype(np.float64)
        dist = dist + 1e-3 * np.arange(n_cams)[None, :] / n_cams          # stable tie-break: lower camera index first
        dist[~vis[rows]] = np.inf
        sel = np.argsort(dist, axis=1, kind='stable')[:, :n_obs]
        return rows, np.sort(sel, axis=1)

    # ---------------- points
    pts = np.zeros((0, 3)); pts_cams = np.zeros((0, obs_per_point), np.int64)
    if n_points > 0 and obs_per_point > n_cams:
        raise ValueError(f"{obs_per_point} observations per point need at least as many cameras (have {n_cams})")
    while pts.shape[0] < n_points:
        m = int((n_points - pts.shape[0]) * 1.6) + 64
        k = rng.integers(0, n_cams, m)
        u = rng.uniform(0, IMG_W, m); v = rng.uniform(0, IMG_H, m); d = rng.uniform(4.0, 60.0, m)
        Xc = np.stack([(u - cx) / fx * d, (v - cy) / fy * d, d], 1)
        X = np.einsum('nij,nj->ni', np.transpose(Rcw[k], (0, 2, 1)), Xc - tcw[k])
        rows, sel = pick_nearest(visible(X), k, obs_per_point)
        take = min(rows.size, n_points - pts.shape[0])
        pts = np.vstack([pts, X[rows[:take]]]); pts_cams = np.vstack([pts_cams, sel[:take]])
    pt_obs_start = np.arange(0, (n_points + 1) * obs_per_point, obs_per_point, dtype=np.int32)
    pt_obs_cam = pts_cams.reshape(-1).astype(np.int32)
    n_pt_obs = pt_obs_cam.size
    pt_of_obs = np.repeat(np.arange(n_points), obs_per_point)
    Xo = pts[pt_of_obs]
    Xc = np.einsum('nij,nj->ni', Rcw[pt_obs_cam], Xo) + tcw[pt_obs_cam]
    octv = rng.integers(0, 8, n_pt_obs)
    sig = 1.2 ** octv
    sig = sig * noise
    u = fx * Xc[:, 0] / Xc[:, 2] + cx + rng.normal(0, 1, n_pt_obs) * sig
    v = fy * Xc[:, 1] / Xc[:, 2] + cy + rng.normal(0, 1, n_pt_obs) * sig
    ur = fx * Xc[:, 0] / Xc[:, 2] + cx - bf / Xc[:, 2] + rng.normal(0, 1, n_pt_obs) * sig
    out = rng.random(n_pt_obs) < outlier_frac
    u = np.where(out, rng.uniform(0, IMG_W, n_pt_obs), u)
    v = np.where(out, rng.uniform(0, IMG_H, n_pt_obs), v)
    ur = np.where(out, u - rng.uniform(0, 80.0, n_pt_obs), ur)
    ur = np.maximum(ur, 0.0)
    mono = rng.random(n_pt_obs) < mono_frac
    ur = np.where(mono, -1.0, ur)
    pt_obs_uvr = _f32(np.stack([u, v, ur], 1))
    pt_obs_inv_sigma2 = inv_s2[octv]

    # ---------------- lines
    lA = np.zeros((0, 3)); lB = np.zeros((0, 3)); ln_cams = np.zeros((0, obs_per_line), np.int64)
    if n_lines > 0 and obs_per_line > n_cams:
        raise ValueError(f"{obs_per_line} observations per line need at least as many cameras (have {n_cams})")
    while lA.shape[0] < n_lines:
        m = int((n_lines - lA.shape[0]) * 2.0) + 64
        k = rng.integers(0, n_cams, m)
        u0 = rng.uniform(0, IMG_W, m); v0 = rng.uniform(0, IMG_H, m); d = rng.uniform(4.0, 40.0, m)
        Xc = np.stack([(u0 - cx) / fx * d, (v0 - cy) / fy * d, d], 1)
        M = np.einsum('nij,nj->ni', np.transpose(Rcw[k], (0, 2, 1)), Xc - tcw[k])
        dirv = rng.normal(size=(m, 3)); dirv /= np.linalg.norm(dirv, axis=1, keepdims=True)
        L = rng.uniform(1.0, 5.0, m)
        A = M - 0.5 * L[:, None] * dirv; B = M + 0.5 * L[:, None] * dirv
        rows, sel = pick_nearest(visible(A) & visible(B), k, obs_per_line)
        take = min(rows.size, n_lines - lA.shape[0])
        lA = np.vstack([lA, A[rows[:take]]]); lB = np.vstack([lB, B[rows[:take]]]); ln_cams = np.vstack([ln_cams, sel[:take]])
    ln_obs_start = np.arange(0, (n_lines + 1) * obs_per_line, obs_per_line, dtype=np.int32)
    ln_obs_cam = ln_cams.reshape(-1).astype(np.int32)
    n_ln_obs = ln_obs_cam.size
    ln_of_obs = np.repeat(np.arange(n_lines), obs_per_line)

    def seg_obs(bx):
        """Detected segment for every (line,KF): true endpoints slid along the line by +-20 % and jittered 1 px."""
        A = lA[ln_of_obs]; B = lB[ln_of_obs]
        s0 = rng.uniform(-0.2, 0.2, n_ln_obs); s1 = rng.uniform(-0.2, 0.2, n_ln_obs)
        A2 = A + s0[:, None] * (B - A); B2 = B + s1[:, None] * (B - A)
        segs = []
        for P in (A2, B2):
            Xc = np.einsum('nij,nj->ni', Rcw[ln_obs_cam], P) + tcw[ln_obs_cam]
            uu = fx * (Xc[:, 0] + bx) / Xc[:, 2] + cx + rng.normal(0, 1, n_ln_obs) * noise
            vv = fy * Xc[:, 1] / Xc[:, 2] + cy + rng.normal(0, 1, n_ln_obs) * noise
            segs += [uu, vv]
        return np.stack(segs, 1)

    left = seg_obs(0.0); right = seg_obs(-b)
    lout = rng.random(n_ln_obs) < outlier_frac
    for seg in (left, right):
        rnd = np.stack([rng.uniform(0, IMG_W, n_ln_obs), rng.uniform(0, IMG_H, n_ln_obs),
                        rng.uniform(0, IMG_W, n_ln_obs), rng.uniform(0, IMG_H, n_ln_obs)], 1)
        seg[lout] = rnd[lout]
    right[:, 0] = np.maximum(right[:, 0], 0.0)     # xs >= 0 is the "has stereo match" flag
    mono_l = rng.random(n_ln_obs) < mono_line_frac
    right[mono_l] = -1.0
    loct = rng.integers(0, 3, n_ln_obs)
    ln_obs_octave = np.stack([loct, loct], 1).astype(np.int32)

    # ---------------- initial state: perturbed, stored float32 (poses, points) / double (lines)
    cam_qt = np.zeros((n_cams, 7))
    for c in range(n_cams):
        T = np.eye(4); T[:3, :3] = Rcw[c]; T[:3, 3] = tcw[c]
        if c < n_free:
            w = rng.normal(0, np.deg2rad(pose_sigma[0]), 3); dt = rng.normal(0, pose_sigma[1], 3)
            dT = np.eye(4); dT[:3, :3] = _rodrigues(w); dT[:3, 3] = dt
            T = dT @ T
        cam_qt[c] = _tcw_to_qt(T)
    pt_xyz = _f32(pts + rng.normal(0, point_sigma, pts.shape))
    # lines: perturb direction by ~1 deg and position by 5 cm, then re-derive (X0 perpendicular foot, unit dir)
    dirs = lB - lA
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True) if n_lines else dirs
    dirs_p = dirs + rng.normal(0, np.deg2rad(line_sigma[0]), dirs.shape)
    dirs_p = dirs_p / np.linalg.norm(dirs_p, axis=1, keepdims=True) if n_lines else dirs_p
    P0 = lA + rng.normal(0, line_sigma[1], lA.shape)
    X0 = P0 - np.sum(P0 * dirs_p, 1, keepdims=True) * dirs_p

    w = Window(cam=cam, n_free_cams=n_free, cam_qt=cam_qt, pt_xyz=pt_xyz, pt_obs_start=pt_obs_start,
               pt_obs_cam=pt_obs_cam, pt_obs_uvr=pt_obs_uvr, pt_obs_inv_sigma2=pt_obs_inv_sigma2,
               line_x0=X0, line_dir=dirs_p, ln_obs_start=ln_obs_start, ln_obs_cam=ln_obs_cam,
               ln_obs_left=_f32(left), ln_obs_right=_f32(right), ln_obs_octave=ln_obs_octave,
               meta=dict(seed=seed, gt_Rcw=Rcw, gt_tcw=tcw, gt_pts=pts, gt_lA=lA, gt_lB=lB))
    return w.normalise()


def make_lba_b(window_id=0, **kw) -> Window:
    return make_ba_window(50, 10, 10000, 6, 2000, 5, seed=SEED_LBA_B + window_id, **kw)


def make_lba_a(window_id=0, **kw) -> Window:
    return make_ba_window(20, 5, 5000, 6, 1000, 5, seed=SEED_LBA_A + window_id, **kw)


def generate_windows(first_id: int, count: int, workers: int = 0, maker=make_lba_b) -> list:
    """`count` synthetic windows maker(first_id + i), generated on `workers` processes (0: min(cores, 16)).  Spawned, never forked:
    the caller may already have touched the GPU."""
    import os
    if workers <= 0:
        workers = int(os.environ.get("LLD_GEN_WORKERS", "0")) or max(1, min(16, os.cpu_count() or 1))     # LLD_GEN_WORKERS=1 under rocprofv3: no child processes
    if workers <= 1 or count < 4:
        return [maker(first_id + i) for i in range(count)]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(workers, count)) as pool:
        return pool.map(maker, range(first_id, first_id + count), chunksize=max(1, count // (4 * workers)))


def make_lba_small(window_id=0, n_free=6, n_fixed=2, n_points=300, n_lines=60, **kw) -> Window:
    """Tiny window for fast CPU tests."""
    return make_ba_window(n_free, n_fixed, n_points, 4, n_lines, 4, seed=0x5A110000 + window_id, **kw)


def make_sim3_pair(pair_id=0, n=300, outlier_frac=0.15, scale=1.0, noise=1.0, cam=KITTI_CAM):
    """One loop-closure candidate for Optimizer::OptimizeSim3: n MapPoints seen by both keyframes, each known in its own camera
    frame (X1 = S12 X2 up to 3-D noise), keypoints with per-octave pixel noise, a fraction of wrong correspondences, and a start
    value of S12 a few degrees / decimetres / percent off."""
    from .host import Sim3Pair
    rng = np.random.default_rng(0x51300000 + pair_id)
    fx, fy, cx, cy, _ = cam
    R12 = _rodrigues(rng.normal(0, 0.25, 3)); t12 = rng.normal(0, 1.5, 3); s12 = float(scale)
    X2 = np.stack([rng.uniform(-12, 12, n), rng.uniform(-3, 3, n), rng.uniform(5, 45, n)], 1)
    X1 = s12 * (X2 @ R12.T) + t12
    keep = X1[:, 2] > 2.0
    X1, X2 = X1[keep], X2[keep]; n = X1.shape[0]
    octv = np.minimum(rng.geometric(0.35, n) - 1, 7)
    inv_s2 = inv_level_sigma2().astype(np.float64)
    sig = 1.2 ** octv
    def proj(X): return np.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1)
    obs1 = proj(X1) + noise * sig[:, None] * rng.normal(size=(n, 2))
    obs2 = proj(X2) + noise * sig[:, None] * rng.normal(size=(n, 2))
    bad = rng.random(n) < outlier_frac
    obs2[bad] += rng.uniform(-60, 60, (int(bad.sum()), 2))
    p1c = _f32(X1 + rng.normal(0, 0.02, X1.shape)).astype(np.float64); p2c = _f32(X2 + rng.normal(0, 0.02, X2.shape)).astype(np.float64)
    R0 = _rodrigues(rng.normal(0, 0.03, 3)) @ R12
    from scipy.spatial.transform import Rotation
    q0 = Rotation.from_matrix(R0).as_quat()                       # x, y, z, w
    return Sim3Pair(K1=(np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy)), K2=(np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy)),
                    s12_q=q0, s12_t=t12 + rng.normal(0, 0.15, 3), s12_s=s12 * (1.0 + (0.0 if scale == 1.0 else rng.normal(0, 0.03))),
                    p1c=p1c, p2c=p2c, obs1=_f32(obs1).astype(np.float64), obs2=_f32(obs2).astype(np.float64),
                    inv_sigma2_1=inv_s2[octv], inv_sigma2_2=inv_s2[octv], meta=dict(R12=R12, t12=t12, s12=s12, bad=bad))


def make_essential_graph(graph_id=0, n_kf=120, drift=(0.002, 0.03), covis=3, n_corrected=6):
    """A loop for Optimizer::OptimizeEssentialGraph: keyframes on a closed path, odometry that drifts (rotation / translation noise
    per step), so the last keyframes do not meet the first ones.  Vertices = the drifted Siw, except the last `n_corrected`
    keyframes, which carry their loop-corrected Sim3 (CorrectedSim3); edges: spanning tree (i -> i-1), covisibility (i -> i-2 ..
    i-covis) - both measured on the NON-corrected poses - and the loop edge last -> first measured on the corrected ones; keyframe
    0 (the loop keyframe) is fixed.  Returns EssentialGraph with meta = ground-truth poses."""
    from scipy.spatial.transform import Rotation
    from .host import EssentialGraph
    rng = np.random.default_rng(0xE5500000 + graph_id)
    ang = np.linspace(0, 2 * np.pi, n_kf, endpoint=False)
    Rw = [Rotation.from_euler("y", -a).as_matrix() for a in ang]                  # camera-to-world rotation: heading along the circle
    tw = np.stack([40 * np.cos(ang), 0.3 * np.sin(3 * ang), 40 * np.sin(ang)], 1)
    Tgt = [np.block([[Rw[k].T, (-Rw[k].T @ tw[k])[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]) for k in range(n_kf)]   # Tiw
    # drifted odometry: T_k = (noisy relative) * T_{k-1}
    Td = [Tgt[0]]
    for k in range(1, n_kf):
        rel = Tgt[k] @ np.linalg.inv(Tgt[k - 1])
        N = np.eye(4); N[:3, :3] = _rodrigues(rng.normal(0, drift[0], 3)); N[:3, 3] = rng.normal(0, drift[1], 3)
        Td.append(N @ rel @ Td[k - 1])
    def to_sim3(T, s=1.0):
        q = Rotation.from_matrix(T[:3, :3]).as_quat()
        return np.concatenate([q, T[:3, 3], [s]])
    def mat(S): M = np.eye(4); M[:3, :3] = S[7] * Rotation.from_quat(S[:4]).as_matrix(); M[:3, 3] = S[4:7]; return M
    def from_mat(M):
        s = np.cbrt(np.linalg.det(M[:3, :3])); return np.concatenate([Rotation.from_matrix(M[:3, :3] / s).as_quat(), M[:3, 3], [s]])
    non_corr = np.stack([to_sim3(T) for T in Td])
    verts = non_corr.copy()
    # loop correction of the tail: the corrected pose of the last keyframe is its ground truth (what ComputeSim3 found), the
    # keyframes before it get the same correction transform (CorrectLoop propagates g2oCorrectedSiw = Sic * Scw)
    corr = mat(to_sim3(Tgt[-1])) @ np.linalg.inv(mat(non_corr[-1]))              # Scw_corrected * Swc_old ... applied on the world side
    for k in range(n_kf - n_corrected, n_kf):
        verts[k] = from_mat(mat(non_corr[k]) @ np.linalg.inv(mat(non_corr[-1])) @ mat(to_sim3(Tgt[-1])))
    ei, ej, meas = [], [], []
    def add(i, j, Si, Sj):                                                        # Sji = Sjw * Swi
        ei.append(i); ej.append(j); meas.append(from_mat(mat(Sj) @ np.linalg.inv(mat(Si))))
    add(n_kf - 1, 0, verts[n_kf - 1], verts[0])                                   # loop edge, corrected poses (LoopConnections)
    for i in range(1, n_kf):
        add(i, i - 1, non_corr[i], non_corr[i - 1])                               # spanning tree
        for d in range(2, covis + 1):
            if i - d >= 0: add(i, i - d, non_corr[i], non_corr[i - d])            # covisibility >= 100
    fixed = np.zeros(n_kf, np.uint8); fixed[0] = 1
    return EssentialGraph(sim3=verts, fixed=fixed, edge_i=np.array(ei, np.int32), edge_j=np.array(ej, np.int32), edge_sji=np.stack(meas),
                          meta=dict(gt=np.stack([to_sim3(T) for T in Tgt]), drifted=non_corr))


def make_pose_frame(frame_id=0, n_points=1000, n_lines=200, outlier_frac=0.10, mono_frac=0.0, mono_line_frac=0.0,
                    cam=KITTI_CAM, seed=None) -> PoseFrame:
    rng = np.random.default_rng(SEED_PO + frame_id if seed is None else seed)
    fx, fy, cx, cy, bf = cam
    b = bf / fx
    Rcw = _rodrigues(rng.normal(0, 0.2, 3)); pos = rng.uniform(-20, 20, 3)
    tcw = -Rcw @ pos
    inv_s2 = inv_level_sigma2().astype(np.float64)
    # points in the frustum
    u = rng.uniform(0, IMG_W, n_points); v = rng.uniform(0, IMG_H, n_points); d = rng.uniform(4, 60, n_points)
    d = np.maximum(d, bf / np.maximum(u, 1.0) + 0.5)        # keep uR >= 0
    Xc = np.stack([(u - cx) / fx * d, (v - cy) / fy * d, d], 1)
    Xw = _f32((Xc - tcw) @ Rcw)                                # Rcw^T (Xc - t), stored float32
    Xc = Xw @ Rcw.T + tcw
    octv = rng.integers(0, 8, n_points); sig = 1.2 ** octv
    uu = fx * Xc[:, 0] / Xc[:, 2] + cx + rng.normal(0, 1, n_points) * sig
    vv = fy * Xc[:, 1] / Xc[:, 2] + cy + rng.normal(0, 1, n_points) * sig
    ur = fx * Xc[:, 0] / Xc[:, 2] + cx - bf / Xc[:, 2] + rng.normal(0, 1, n_points) * sig
    out = rng.random(n_points) < outlier_frac
    uu = np.where(out, rng.uniform(0, IMG_W, n_points), uu); vv = np.where(out, rng.uniform(0, IMG_H, n_points), vv)
    ur = np.where(out, uu - rng.uniform(0, 80, n_points), ur)
    ur = np.maximum(ur, 0.0)
    ur = np.where(rng.random(n_points) < mono_frac, -1.0, ur)
    # lines
    u0 = rng.uniform(100, IMG_W - 100, n_lines); v0 = rng.uniform(50, IMG_H - 50, n_lines); d = rng.uniform(6, 40, n_lines)
    Mc = np.stack([(u0 - cx) / fx * d, (v0 - cy) / fy * d, d], 1)
    dirc = rng.normal(size=(n_lines, 3)); dirc[:, 2] *= 0.3; dirc /= np.linalg.norm(dirc, axis=1, keepdims=True)
    L = rng.uniform(1, 4, n_lines)
    Ac = Mc - 0.5 * L[:, None] * dirc; Bc = Mc + 0.5 * L[:, None] * dirc
    A = (Ac - tcw) @ Rcw; B = (Bc - tcw) @ Rcw
    dirw = (B - A) / np.linalg.norm(B - A, axis=1, keepdims=True)
    X0 = A - np.sum(A * dirw, 1, keepdims=True) * dirw

    def seg(bx):
        s0 = rng.uniform(-0.2, 0.2, n_lines); s1 = rng.uniform(-0.2, 0.2, n_lines)
        cols = []
        for P in (Ac + s0[:, None] * (Bc - Ac), Bc + s1[:, None] * (Bc - Ac)):
            cols += [fx * (P[:, 0] + bx) / P[:, 2] + cx + rng.normal(0, 1, n_lines),
                     fy * P[:, 1] / P[:, 2] + cy + rng.normal(0, 1, n_lines)]
        return np.stack(cols, 1)

    left = seg(0.0); right = seg(-b)
    lout = rng.random(n_lines) < outlier_frac
    for s in (left, right):
        rnd = np.stack([rng.uniform(0, IMG_W, n_lines), rng.uniform(0, IMG_H, n_lines),
                        rng.uniform(0, IMG_W, n_lines), rng.uniform(0, IMG_H, n_lines)], 1)
        s[lout] = rnd[lout]
    right[:, 0] = np.maximum(right[:, 0], 0.0)
    right[rng.random(n_lines) < mono_line_frac] = -1.0
    loct = rng.integers(0, 3, n_lines)
    # initial pose = exp(2 deg, 0.3 m) * GT, stored float32
    w = rng.normal(size=3); w *= np.deg2rad(2.0) / np.linalg.norm(w)
    dt = rng.normal(size=3); dt *= 0.3 / np.linalg.norm(dt)
    T = np.eye(4); T[:3, :3] = Rcw; T[:3, 3] = tcw
    dT = np.eye(4); dT[:3, :3] = _rodrigues(w); dT[:3, 3] = dt
    pose_qt = _tcw_to_qt(dT @ T)
    f = PoseFrame(cam=cam, pose_qt=pose_qt, pt_xw=Xw, pt_uvr=_f32(np.stack([uu, vv, ur], 1)),
                  pt_inv_sigma2=inv_s2[octv], ln_x0=X0, ln_dir=dirw, ln_left=_f32(left), ln_right=_f32(right),
                  ln_octave=np.stack([loct, loct], 1).astype(np.int32),
                  meta=dict(gt_Rcw=Rcw, gt_tcw=tcw, gt_qt=_tcw_to_qt(T)))
    return f.normalise()


def make_match_orb(pair_id=0, nq=2000, nt=2000, n_corr=1600, flip_p=0.08, n_dup=16):
    """ORB descriptors: n_corr train rows are permuted query rows with bits flipped w.p. flip_p, the rest unrelated;
    n_dup exact duplicates of earlier train rows exercise ties (the lowest index must win)."""
    rng = np.random.default_rng(SEED_MATCH + pair_id)
    q = rng.integers(0, 2 ** 32, (nq, 8), dtype=np.uint64).astype(np.uint32)
    t = rng.integers(0, 2 ** 32, (nt, 8), dtype=np.uint64).astype(np.uint32)
    n_corr = min(n_corr, nq, nt)
    src = rng.permutation(nq)[:n_corr]; dst = rng.permutation(nt)[:n_corr]
    flips = rng.random((n_corr, 256)) < flip_p
    fl = np.packbits(flips.reshape(n_corr, 8, 32)[:, :, ::-1], axis=2, bitorder='big').view('>u4').reshape(n_corr, 8).astype(np.uint32)
    t[dst] = q[src] ^ fl
    if n_dup and nt > 2 * n_dup:
        a = rng.permutation(nt)[:2 * n_dup]
        t[a[n_dup:]] = t[a[:n_dup]]
    return q, t


def make_match_lbd(pair_id=0, nq=300, nt=300, dim=72, n_corr=240, noise=0.05):
    rng = np.random.default_rng(SEED_MATCH + 0x8000 + pair_id)
    q = rng.normal(size=(nq, dim)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.normal(size=(nt, dim)); t /= np.linalg.norm(t, axis=1, keepdims=True)
    n_corr = min(n_corr, nq, nt)
    src = rng.permutation(nq)[:n_corr]; dst = rng.permutation(nt)[:n_corr]
    t[dst] = q[src] + rng.normal(0, noise, (n_corr, dim))
    return q.astype(np.float32), t.astype(np.float32)


# ====================================================================== guided ORB search scenes
SEED_SEARCH = 0x5EA2C400


def _flip_bits(rng, desc, p, keep_word0_low=0):
    """XOR each of the 256 bits with probability p; optionally keep the low bits of word 0 (the synthetic 'vocabulary node')."""
    n = desc.shape[0]
    flips = rng.random((n, 256)) < p
    fl = np.packbits(flips.reshape(n, 8, 32)[:, :, ::-1], axis=2, bitorder='big').view('>u4').reshape(n, 8).astype(np.uint32)
    if keep_word0_low:
        fl[:, 0] &= np.uint32(~((1 << keep_word0_low) - 1) & 0xFFFFFFFF)
    return desc ^ fl


def make_orb_frame(frame_id=0, n=2000, width=1241.0, height=376.0, stereo_frac=0.75, n_clusters=60):
    """One frame's keypoints as ORBextractor produces them: integer pixel positions at the keypoint's pyramid level scaled
    back by mvScaleFactor[octave] (so equal coordinates and exact |dx| == r ties do occur), octaves skewed to the fine
    levels, angles in [0,360), stereo keypoints with uR = x - disparity, the rest uR = -1.  `n_clusters` tight clusters of
    near-duplicate descriptors make several queries compete for the same keypoints."""
    from .orb_search import Frame, orb_levels
    rng = np.random.default_rng(SEED_SEARCH + frame_id)
    scale, _, _ = orb_levels()
    octave = np.minimum(rng.geometric(0.35, n) - 1, 7).astype(np.int32)
    lx = rng.integers(16, np.floor((width - 16) / scale[octave]).astype(np.int64))
    ly = rng.integers(16, np.floor((height - 16) / scale[octave]).astype(np.int64))
    xy = np.stack([lx.astype(np.float32) * scale[octave], ly.astype(np.float32) * scale[octave]], 1).astype(np.float32)
    desc = rng.integers(0, 2 ** 32, (n, 8), dtype=np.uint64).astype(np.uint32)
    if n_clusters and n > 8 * n_clusters:
        centres = rng.permutation(n)[:n_clusters]
        for c in centres:
            members = rng.permutation(n)[:3]
            xy[members] = xy[c] + (rng.integers(-3, 4, (3, 2)) * scale[octave[c]]).astype(np.float32)
            octave[members] = octave[c]
            desc[members] = _flip_bits(rng, np.repeat(desc[c:c + 1], 3, 0), 0.01)
            desc[members[0]] = desc[c]                                             # one exact duplicate: a tie
        xy[:, 0] = np.clip(xy[:, 0], 0, width - 1); xy[:, 1] = np.clip(xy[:, 1], 0, height - 1)
    disparity = rng.uniform(2.0, 90.0, n).astype(np.float32)
    uright = np.where(rng.random(n) < stereo_frac, xy[:, 0] - disparity, np.float32(-1.0)).astype(np.float32)
    angle = rng.uniform(0.0, 360.0, n).astype(np.float32)
    return Frame(desc=desc, xy=xy, octave=octave, uright=uright, angle=angle, min_x=0.0, min_y=0.0, max_x=width, max_y=height).normalise()


def make_projection_queries(F, scene_id=0, nq=1500, related_frac=0.85, flip_p=0.06, pos_sigma=2.5, rotation=35.0, rot_outliers=0.15,
                            dup_frac=0.2):
    """Queries (MapPoints / last-frame keypoints) projected into frame F: most are noisy copies of F's keypoints (position,
    descriptor, octave +-1, right coordinate, angle = keypoint angle + a common rotation), `dup_frac` of them share their
    source keypoint with an earlier query (competition -> the order-dependent rule matters), the rest are unrelated."""
    rng = np.random.default_rng(SEED_SEARCH + 0x1000 + scene_id)
    n = F.n
    src = rng.integers(0, n, nq)
    n_dup = int(dup_frac * nq)
    if n_dup:
        a = rng.permutation(nq)[:2 * n_dup]
        src[a[n_dup:]] = src[a[:n_dup]]
    related = rng.random(nq) < related_frac
    desc = _flip_bits(rng, F.desc[src], flip_p)
    hard = rng.random(nq) < 0.15                                                   # distances around the accept thresholds
    desc[hard] = _flip_bits(rng, F.desc[src[hard]], 0.2)
    desc[~related] = rng.integers(0, 2 ** 32, (int((~related).sum()), 8), dtype=np.uint64).astype(np.uint32)
    uv = (F.xy[src] + rng.normal(0, pos_sigma, (nq, 2))).astype(np.float32)
    uv[~related] = np.stack([rng.uniform(-30, F.max_x + 30, int((~related).sum())), rng.uniform(-30, F.max_y + 30, int((~related).sum()))], 1)
    level = np.clip(F.octave[src] + rng.integers(-1, 2, nq), 0, 7).astype(np.int32)
    ur = np.where(F.uright[src] > 0, F.uright[src] + rng.normal(0, 2.0, nq), uv[:, 0] - rng.uniform(2, 90, nq)).astype(np.float32)
    far = rng.random(nq) < 0.1
    ur[far] += rng.uniform(20, 60, int(far.sum())).astype(np.float32)               # stereo gate rejects these
    ang = F.angle[src] + np.float32(rotation) + rng.normal(0, 6.0, nq)
    out = rng.random(nq) < rot_outliers
    ang[out] = rng.uniform(0, 360, int(out.sum()))
    ang = np.mod(ang, 360.0).astype(np.float32)
    return dict(desc=desc, uv=uv, ur=ur, level=level, angle=ang, src=src.astype(np.int32),
                valid=(rng.random(nq) < 0.95).astype(np.uint8), obs=(rng.random(nq) < 0.9).astype(np.uint8),
                view_cos=rng.uniform(0.99, 1.0, nq).astype(np.float32), occupied=(rng.random(n) < 0.05).astype(np.uint8))


def make_bow_pair(pair_id=0, n=2000, n_nodes=400, related_frac=0.7, flip_p=0.06, pos_sigma=(6.0, 6.0)):
    """Two frames sharing scene content plus their vocabulary-node lists (a stand-in for DBoW2::FeatureVector: the node of a
    keypoint is the low bits of its first descriptor word, preserved by the noise, so corresponding keypoints share nodes).
    Returns (F1, F2, nodes) with nodes = dict(n_nodes, start1, idx1, start2, idx2) over the COMMON nodes in ascending id."""
    from .orb_search import Frame
    rng = np.random.default_rng(SEED_SEARCH + 0x2000 + pair_id)
    F1 = make_orb_frame(1000 + 2 * pair_id, n)
    F2 = make_orb_frame(1001 + 2 * pair_id, n)
    m = int(related_frac * n)
    s1 = rng.permutation(n)[:m]; s2 = rng.permutation(n)[:m]
    bits = max(1, int(np.ceil(np.log2(n_nodes))))
    F2.desc[s2] = _flip_bits(rng, F1.desc[s1], flip_p, keep_word0_low=bits)
    F2.xy[s2] = (F1.xy[s1] + rng.normal(0, 1.0, (m, 2)) * np.asarray(pos_sigma)).astype(np.float32)
    F2.octave[s2] = F1.octave[s1]
    F2.angle[s2] = np.mod(F1.angle[s1] - 20.0 + rng.normal(0, 5.0, m), 360.0).astype(np.float32)
    node1 = (F1.desc[:, 0] & np.uint32((1 << bits) - 1)).astype(np.int64) % n_nodes
    node2 = (F2.desc[:, 0] & np.uint32((1 << bits) - 1)).astype(np.int64) % n_nodes
    common = np.intersect1d(node1, node2)
    start1, idx1, start2, idx2 = [0], [], [0], []
    for nd in common:
        a = np.nonzero(node1 == nd)[0]; b = np.nonzero(node2 == nd)[0]          # ascending keypoint index inside a node
        idx1.extend(a.tolist()); start1.append(len(idx1)); idx2.extend(b.tolist()); start2.append(len(idx2))
    nodes = dict(n_nodes=len(common), start1=np.array(start1, np.int32), idx1=np.array(idx1, np.int32),
                 start2=np.array(start2, np.int32), idx2=np.array(idx2, np.int32))
    return F1, F2, nodes


def make_init_pair(pair_id=0, n=2000, related_frac=0.8, flip_p=0.05, flow_sigma=12.0, rival_frac=0.15):
    """Two consecutive monocular frames for ORBmatcher::SearchForInitialization: F2 shows `related_frac` of F1's keypoints again, moved
    by an optical flow of a few pixels, with descriptor noise and a common rotation; a share of F1's keypoints are near-duplicates
    of an earlier one (same place, descriptor a few bits off), so that several queries want the same keypoint of F2 and the later,
    better one takes it away from its holder.  vbPrevMatched starts as F1's own positions (Tracking::MonocularInitialization).
    Returns (F1, F2, prev_matched)."""
    rng = np.random.default_rng(SEED_SEARCH + 0x9000 + pair_id)
    F1 = make_orb_frame(3000 + 2 * pair_id, n, n_clusters=0)
    F2 = make_orb_frame(3001 + 2 * pair_id, n, n_clusters=0)
    F1.octave[:] = np.where(rng.random(n) < 0.6, 0, F1.octave); F2.octave[:] = np.where(rng.random(n) < 0.6, 0, F2.octave)   # mostly the finest level
    n_riv = int(rival_frac * n)
    src = rng.integers(0, n, n_riv); dst = rng.permutation(n)[:n_riv]
    keep = dst > src                                                             # the rival comes later in the loop than its original
    src, dst = src[keep], dst[keep]
    F1.xy[dst] = F1.xy[src] + rng.integers(-2, 3, (src.size, 2)).astype(np.float32)
    F1.octave[dst] = F1.octave[src]
    F1.desc[dst] = _flip_bits(rng, F1.desc[src], 0.01)
    m = int(related_frac * n)
    s1 = rng.permutation(n)[:m]; s2 = rng.permutation(n)[:m]
    F2.desc[s2] = _flip_bits(rng, F1.desc[s1], flip_p)
    F2.xy[s2] = (F1.xy[s1] + rng.normal(0, flow_sigma, (m, 2))).astype(np.float32)
    F2.xy[:, 0] = np.clip(F2.xy[:, 0], 0, 1240); F2.xy[:, 1] = np.clip(F2.xy[:, 1], 0, 375)
    F2.octave[s2] = F1.octave[s1]
    F2.angle[s2] = np.mod(F1.angle[s1] - 15.0 + rng.normal(0, 4.0, m), 360.0).astype(np.float32)
    wrong = rng.random(m) < 0.1                                                  # some with an unrelated orientation: the histogram removes them
    F2.angle[s2[wrong]] = rng.uniform(0, 360, int(wrong.sum())).astype(np.float32)
    return F1.normalise(), F2.normalise(), F1.xy.copy()


def make_stereo_pair(pair_id=0, n=2000, flip_p=0.06):
    """Left / right keypoints of one stereo frame: right keypoints are the left ones shifted by a disparity (rows within the
    +-2*scale band), plus unrelated ones."""
    rng = np.random.default_rng(SEED_SEARCH + 0x3000 + pair_id)
    L = make_orb_frame(2000 + pair_id, n, n_clusters=40)
    R = make_orb_frame(2500 + pair_id, n, n_clusters=0)
    m = int(0.75 * n)
    sl = rng.permutation(n)[:m]; sr = rng.permutation(n)[:m]
    disp = rng.uniform(-5.0, 110.0, m).astype(np.float32)                         # some negative / too large: rejected by [minU,maxU]
    R.xy[sr, 0] = L.xy[sl, 0] - disp
    R.xy[sr, 1] = L.xy[sl, 1] + rng.normal(0, 1.2, m).astype(np.float32)
    R.xy[:, 1] = np.clip(R.xy[:, 1], 8.0, 360.0)
    R.octave[sr] = np.clip(L.octave[sl] + rng.integers(-2, 3, m), 0, 7)
    R.desc[sr] = _flip_bits(rng, L.desc[sl], flip_p)
    return L, R


def make_stereo_scene(pair_id=0, n=2000, width=1241, height=376, flip_p=0.05, edge=19):
    """A rectified stereo frame WITH images, for Frame::ComputeStereoMatches as a whole: a smooth random texture as the left image,
    the right image = the left one warped by a smooth disparity field d(x,y) in [4, 70] px, both as 8-level pyramids (level sizes
    cvRound(size * invScale) like ORBextractor, bilinear resampling); left keypoints on integer pixels of their level at least
    `edge` px inside it, right keypoints at x - d (+ sub-pixel jitter), a few rows off, octave +-1, descriptors with flipped bits,
    plus unrelated right keypoints and some pairs with a wrong disparity (caught by the SAD stage or the median cut).
    Returns dict(L, R, left, right, inv_scale, mb, mbf)."""
    from scipy import ndimage
    from .orb_search import Frame, orb_levels
    rng = np.random.default_rng(SEED_SEARCH + 0x5000 + pair_id)
    scale, _, _ = orb_levels()
    inv_scale = (np.float32(1.0) / scale).astype(np.float32)
    tex = ndimage.gaussian_filter(rng.normal(size=(height, width + 96)), 1.6) + 0.6 * ndimage.gaussian_filter(rng.normal(size=(height, width + 96)), 5.0)
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    big = np.clip(255.0 * tex, 0, 255)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    disp = 4.0 + 60.0 * (yy / height) + 6.0 * np.sin(xx / 170.0)                                  # nearer towards the bottom of the image
    left0 = ndimage.map_coordinates(big, [yy, xx + 8.0], order=1)
    right0 = ndimage.map_coordinates(big, [yy, xx + 8.0 + disp], order=1, mode="nearest")       # R(x) = L(x + d)
    def pyramid(img0):
        out = []
        for l in range(scale.shape[0]):
            w, h = int(np.round(width * inv_scale[l])), int(np.round(height * inv_scale[l]))
            y, x = np.mgrid[0:h, 0:w].astype(np.float64)
            out.append(np.clip(np.round(ndimage.map_coordinates(img0, [y * float(scale[l]), x * float(scale[l])], order=1, mode="nearest")), 0, 255).astype(np.uint8))
        return out
    left, right = pyramid(left0), pyramid(right0)
    octave = np.minimum(rng.geometric(0.35, n) - 1, 7).astype(np.int32)
    lw = np.array([left[o].shape[1] for o in octave]); lh = np.array([left[o].shape[0] for o in octave])
    lx = rng.integers(edge, lw - edge); ly = rng.integers(edge, lh - edge)
    xy = np.stack([lx.astype(np.float32) * scale[octave], ly.astype(np.float32) * scale[octave]], 1).astype(np.float32)
    desc = rng.integers(0, 2 ** 32, (n, 8), dtype=np.uint64).astype(np.uint32)
    L = Frame(desc=desc, xy=xy, octave=octave, uright=np.full(n, -1, np.float32), angle=np.zeros(n, np.float32), min_x=0.0, min_y=0.0,
              max_x=float(width), max_y=float(height)).normalise()
    m = int(0.8 * n)
    src = rng.permutation(n)[:m]
    d_true = ndimage.map_coordinates(disp, [xy[src, 1].astype(np.float64), xy[src, 0].astype(np.float64)], order=1, mode="nearest")
    wrong = rng.random(m) < 0.06
    d_used = np.where(wrong, d_true + rng.uniform(-25, 25, m), d_true) + rng.normal(0, 0.4, m)
    r_oct = np.clip(octave[src] + rng.integers(-1, 2, m), 0, 7).astype(np.int32)
    rxy = np.stack([xy[src, 0] - d_used, xy[src, 1] + rng.normal(0, 0.8, m)], 1).astype(np.float32)
    # unrelated right keypoints
    k = n - m
    uo = np.minimum(rng.geometric(0.35, k) - 1, 7).astype(np.int32)
    uxy = np.stack([rng.uniform(20, width - 20, k), rng.uniform(20, height - 20, k)], 1).astype(np.float32)
    R = Frame(desc=np.concatenate([_flip_bits(rng, desc[src], flip_p), rng.integers(0, 2 ** 32, (k, 8), dtype=np.uint64).astype(np.uint32)]),
              xy=np.concatenate([rxy, uxy]), octave=np.concatenate([r_oct, uo]), uright=np.full(n, -1, np.float32), angle=np.zeros(n, np.float32),
              min_x=0.0, min_y=0.0, max_x=float(width), max_y=float(height)).normalise()
    R.xy[:, 0] = np.clip(R.xy[:, 0], 0.0, width - 1.0); R.xy[:, 1] = np.clip(R.xy[:, 1], 0.0, height - 1.0)
    perm = rng.permutation(n)                                                                    # right keypoints in no particular order
    R.desc, R.xy, R.octave = np.ascontiguousarray(R.desc[perm]), np.ascontiguousarray(R.xy[perm]), np.ascontiguousarray(R.octave[perm])
    fx, _, _, _, bf = KITTI_CAM
    mbf = np.float32(bf); mb = np.float32(mbf / np.float32(fx))                                   # mb = mbf / fx (src/Frame.cc:100)
    return dict(L=L, R=R, left=left, right=right, inv_scale=inv_scale, mb=float(mb), mbf=float(mbf), src=src, perm=perm, d_true=d_true, wrong=wrong)


# ====================================================================== stereo line association (TwoFrameLineMatcher)
def make_stereo_lines(frame_id=0, n_left=300, n_right=300, dim=72, related_frac=0.8, pixel_noise=0.4, desc_noise=0.05):
    """Left / right KeyLines of one stereo frame: random 3D segments (depth 3-40 m, some nearly parallel to the baseline so the
    triangulation-angle gate rejects them, some partly behind the camera after noise) projected with the KITTI intrinsics
    and baseline, endpoints slid along the line in the right image, plus unrelated lines, short lines and octave mismatches.
    Returns dict(K, b, left [n,4] f32, left_octave, right, right_octave, desc_left, desc_right)."""
    rng = np.random.default_rng(SEED_SEARCH + 0x4000 + frame_id)
    fx, fy, cx, cy, bf = KITTI_CAM
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    b = float(np.float32(np.float32(bf) / np.float32(fx)))               # mbf/mK.at<float>(0,0), src/Frame.cc:121
    m = int(related_frac * min(n_left, n_right))

    def segs(n):
        z = rng.uniform(3.0, 40.0, n)
        c = np.stack([(rng.uniform(60, 1180, n) - cx) * z / fx, (rng.uniform(30, 340, n) - cy) * z / fy, z], 1)
        d = rng.normal(size=(n, 3)); d[:, 2] *= 0.3
        flat = rng.random(n) < 0.15
        d[flat, 1] *= 0.02                                               # nearly parallel to the baseline: epipolar-degenerate
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        half = rng.uniform(0.15, 2.5, n)[:, None] * 0.5
        return c - half * d, c + half * d

    def proj(X, shift):
        Xc = X - np.array([shift, 0, 0])
        zz = np.maximum(Xc[:, 2], 0.3)
        return np.stack([fx * Xc[:, 0] / zz + cx, fy * Xc[:, 1] / zz + cy], 1)
    A, B = segs(m)
    la, lb = proj(A, 0.0), proj(B, 0.0)
    s = rng.uniform(-0.2, 0.2, (m, 2))                                   # endpoints slide along the 3D line in the right image
    Ar, Br = A + s[:, :1] * (B - A), B + s[:, 1:] * (B - A)
    ra, rb = proj(Ar, b), proj(Br, b)
    left = np.concatenate([la, lb], 1) + rng.normal(0, pixel_noise, (m, 4))
    right = np.concatenate([ra, rb], 1) + rng.normal(0, pixel_noise, (m, 4))
    octave = rng.integers(0, 3, m)

    def unrelated(n):
        p = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n)], 1)
        q = p + rng.normal(0, 40, (n, 2))
        return np.concatenate([p, q], 1)
    left = np.concatenate([left, unrelated(n_left - m)]); right = np.concatenate([right, unrelated(n_right - m)])
    lo = np.concatenate([octave, rng.integers(0, 3, n_left - m)]); ro = np.concatenate([octave, rng.integers(0, 3, n_right - m)])
    mism = rng.random(m) < 0.08; ro[:m][mism] = (ro[:m][mism] + 1) % 3  # octave mismatch -> gate fails
    dl = rng.normal(size=(n_left, dim)); dl /= np.linalg.norm(dl, axis=1, keepdims=True)
    dr = rng.normal(size=(n_right, dim)); dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    dr[:m] = dl[:m] + rng.normal(0, desc_noise, (m, dim))
    pl, pr = rng.permutation(n_left), rng.permutation(n_right)            # shuffle so correspondences are not index-aligned
    return dict(K=K, b=b, left=left[pl].astype(np.float32), left_octave=lo[pl].astype(np.int32), right=right[pr].astype(np.float32),
                right_octave=ro[pr].astype(np.int32), desc_left=dl[pl].astype(np.float32), desc_right=dr[pr].astype(np.float32))



# ====================================================================== map-line tracking (Tracking::AddLinesFrom)
def make_line_track_scene(scene_id=0, n_map=250, n_cur=300, dim=72, related_frac=0.75, pixel_noise=0.5, desc_noise=0.05, behind_frac=0.05,
                          occupied_frac=0.05, no_partner_frac=0.1, skip_frac=0.08):
    """Map lines and the lines of the current stereo frame for Tracking::AddLinesFrom.  A camera pose T_curr (camera-to-world), 3D
    segments in front of it (a few behind), their noisy left / right projections among unrelated frame lines, left -> right
    partner indices (some missing), pre-occupied frame lines, skipped map lines, near-duplicate descriptors so that several map lines
    compete for one frame line.  Returns (params dict, lines_last dict, frame dict)."""
    rng = np.random.default_rng(SEED_SEARCH + 0x7000 + scene_id)
    fx, fy, cx, cy, bf = KITTI_CAM
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    b = float(np.float32(np.float32(bf) / np.float32(fx)))
    R = _rodrigues(rng.normal(0, 0.3, 3)); t = rng.normal(0, 3.0, 3)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t                               # camera-to-world: X_cam = R^T (X - t)
    z = rng.uniform(3.0, 40.0, n_map)
    c = np.stack([(rng.uniform(60, 1180, n_map) - cx) * z / fx, (rng.uniform(30, 340, n_map) - cy) * z / fy, z], 1)
    behind = rng.random(n_map) < behind_frac
    c[behind, 2] = -rng.uniform(1.0, 10.0, int(behind.sum()))
    d = rng.normal(size=(n_map, 3)); d[:, 2] *= 0.3; d /= np.linalg.norm(d, axis=1, keepdims=True)
    half = rng.uniform(0.3, 2.5, n_map)[:, None] * 0.5
    A, B = c - half * d, c + half * d                                        # camera frame

    def proj(X, shift):
        Xc = X - np.array([shift, 0, 0]); zz = np.maximum(Xc[:, 2], 0.3)
        return np.stack([fx * Xc[:, 0] / zz + cx, fy * Xc[:, 1] / zz + cy], 1)
    m = int(related_frac * min(n_map, n_cur))
    left = np.concatenate([proj(A[:m], 0.0), proj(B[:m], 0.0)], 1) + rng.normal(0, pixel_noise, (m, 4))
    s = rng.uniform(-0.2, 0.2, (m, 2))
    Ar, Br = A[:m] + s[:, :1] * (B[:m] - A[:m]), B[:m] + s[:, 1:] * (B[:m] - A[:m])
    right = np.concatenate([proj(Ar, b), proj(Br, b)], 1) + rng.normal(0, pixel_noise, (m, 4))
    far = rng.random(m) < 0.1                                                # wrong geometry: the reprojection gate must reject
    left[far] += rng.normal(0, 25.0, (int(far.sum()), 4))

    def unrelated(n):
        p = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n)], 1)
        return np.concatenate([p, p + rng.normal(0, 40, (n, 2))], 1)
    left = np.concatenate([left, unrelated(n_cur - m)]); right = np.concatenate([right, unrelated(n_cur - m)])
    lo = rng.integers(0, 3, n_cur)
    to_w = lambda X: (R @ X.T).T + t
    Aw, Bw = to_w(A), to_w(B)
    dirw = (Bw - Aw) / np.linalg.norm(Bw - Aw, axis=1, keepdims=True)
    X0 = Aw - np.sum(Aw * dirw, axis=1, keepdims=True) * dirw                 # minimal position: the point of the line closest to the origin
    dm = rng.normal(size=(n_map, dim)); dm /= np.linalg.norm(dm, axis=1, keepdims=True)
    dup = rng.integers(0, m, max(1, n_map // 12)); tgt = rng.integers(0, m, dup.size)
    dm[dup] = dm[tgt] + rng.normal(0, 0.01, (dup.size, dim))               # rivals for the same frame line
    dc = rng.normal(size=(n_cur, dim)); dc /= np.linalg.norm(dc, axis=1, keepdims=True)
    dc[:m] = dm[:m] + rng.normal(0, desc_noise, (m, dim))
    perm = rng.permutation(n_cur); rperm = rng.permutation(n_cur)          # right lines stored in another order
    inv_r = np.empty(n_cur, np.int64); inv_r[rperm] = np.arange(n_cur)
    line_matches = inv_r[perm].astype(np.int32)                             # left line (after perm) -> index of its right partner
    line_matches[rng.random(n_cur) < no_partner_frac] = -1
    params = dict(K=K, T_curr=T, b=b, thr_reproj_base=2.0, md_thr=0.9, sx=1.0 / 1241.0, sy=1.0 / 376.0)
    lines_last = dict(X0=X0, dir=dirw, X1=Aw, X2=Bw, desc=dm.astype(np.float32), skip=(rng.random(n_map) < skip_frac).astype(np.uint8))
    frame = dict(left_lines=left[perm].astype(np.float32), left_octave=lo[perm].astype(np.int32), right_lines=right[rperm].astype(np.float32),
                 line_matches=line_matches, occupied=(rng.random(n_cur) < occupied_frac).astype(np.uint8), desc=dc[perm].astype(np.float32))
    return params, lines_last, frame


def make_two_frame_lines(scene_id=0, n_lines=260, dim=72, shared_frac=0.7, pixel_noise=0.4, desc_noise=0.05, baseline=2.0):
    """Two consecutive stereo frames that see a set of 3D segments (Tracking::MatchLinesLastKF): poses T_last, T_curr
    (camera-to-world, about one metre apart), left / right KeyLines of both frames with noise, stereo partner indices, unrelated and
    occupied lines, a few lines of the last frame already tracked.  vgl::TriangulateLine refuses two views whose back-projected
    planes meet under less than acos(0.975) = 12.8 degrees, i.e. lines further away than ~4.4 baselines: the rig here has a wide baseline
    and near lines so that true stereo pairs pass (with KITTI's 0.54 m only lines within 2.4 m would).
    Returns (params, current dict, last dict, truth)."""
    rng = np.random.default_rng(SEED_SEARCH + 0x8000 + scene_id)
    fx, fy, cx, cy, bf = KITTI_CAM
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    b = float(baseline)

    def pose(w, t):
        T = np.eye(4); T[:3, :3] = _rodrigues(w); T[:3, 3] = t; return T
    T_last = pose(rng.normal(0, 0.05, 3), rng.normal(0, 0.5, 3))
    # (vgl::MultiTriangulateLine wants every other view's plane more than 12.8 degrees away from the first one: a sideways / vertical move)
    T_curr = pose(rng.normal(0, 0.05, 3), T_last[:3, 3] + T_last[:3, :3] @ np.array([rng.normal(-0.9, 0.1), rng.normal(1.6, 0.1), rng.normal(0.3, 0.1)]))
    m = int(shared_frac * n_lines)
    z = rng.uniform(2.5, 9.0, m)                                            # (beyond ~4.4 baselines the triangulation-angle gate refuses)
    c = np.stack([(rng.uniform(350, 1000, m) - cx) * z / fx, (rng.uniform(80, 290, m) - cy) * z / fy, z], 1)
    d = rng.normal(size=(m, 3)); d[:, 2] *= 0.3
    d[rng.random(m) < 0.12, 1] *= 0.02                                      # nearly parallel to the baseline: the stereo triangulation refuses
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    half = rng.uniform(0.3, 1.2, m)[:, None] * 0.5
    Aw = (T_last[:3, :3] @ (c - half * d).T).T + T_last[:3, 3]; Bw = (T_last[:3, :3] @ (c + half * d).T).T + T_last[:3, 3]

    def frame(T, seed_shift):
        R, t = T[:3, :3], T[:3, 3]

        def proj(X, shift):
            Xc = (R.T @ (X - t).T).T - np.array([shift, 0, 0]); zz = np.maximum(Xc[:, 2], 0.3)
            return np.stack([fx * Xc[:, 0] / zz + cx, fy * Xc[:, 1] / zz + cy], 1)
        left = np.concatenate([proj(Aw, 0.0), proj(Bw, 0.0)], 1) + rng.normal(0, pixel_noise, (m, 4))
        s = rng.uniform(-0.15, 0.15, (m, 2))
        right = np.concatenate([proj(Aw + s[:, :1] * (Bw - Aw), b), proj(Bw + s[:, 1:] * (Bw - Aw), b)], 1) + rng.normal(0, pixel_noise, (m, 4))
        p = np.stack([rng.uniform(0, 1241, n_lines - m), rng.uniform(0, 376, n_lines - m)], 1)
        un = np.concatenate([p, p + rng.normal(0, 40, (n_lines - m, 2))], 1)
        p2 = np.stack([rng.uniform(0, 1241, n_lines - m), rng.uniform(0, 376, n_lines - m)], 1)
        un2 = np.concatenate([p2, p2 + rng.normal(0, 40, (n_lines - m, 2))], 1)
        return np.concatenate([left, un]), np.concatenate([right, un2])
    cl, cr = frame(T_curr, 0); ll, lr = frame(T_last, 1)
    base = rng.normal(size=(n_lines, dim)); base /= np.linalg.norm(base, axis=1, keepdims=True)
    dc = base + rng.normal(0, desc_noise, base.shape); dl = base + rng.normal(0, desc_noise, base.shape)
    dl[m:] = rng.normal(size=(n_lines - m, dim)) / np.sqrt(dim)

    def shuffle(left, right, desc):
        pl, pr = rng.permutation(n_lines), rng.permutation(n_lines)
        inv = np.empty(n_lines, np.int64); inv[pr] = np.arange(n_lines)
        lm = inv[pl].astype(np.int32); lm[rng.random(n_lines) < 0.08] = -1
        return left[pl].astype(np.float32), right[pr].astype(np.float32), lm, desc[pl].astype(np.float32), pl
    cl, cr, clm, dc, pc = shuffle(cl, cr, dc); ll, lr, llm, dl, pl_ = shuffle(ll, lr, dl)
    params = dict(K=K, T_curr=T_curr, T_last=T_last, b=b, thr_reproj_base=6.0, md_thr=0.9, sx=1.0 / 1241.0, sy=1.0 / 376.0)
    cur = dict(left_lines=cl, right_lines=cr, line_matches=clm, desc=dc, occupied=(rng.random(n_lines) < 0.06).astype(np.uint8))
    last = dict(left_lines=ll, right_lines=lr, line_matches=llm, desc=dl, left_octave=rng.integers(0, 3, n_lines).astype(np.int32),
                skip=(rng.random(n_lines) < 0.06).astype(np.uint8))
    inv_last = np.empty(n_lines, np.int64); inv_last[pl_] = np.arange(n_lines)
    truth = dict(last_of_cur=np.where(pc < m, inv_last[pc], -1), Aw=Aw, Bw=Bw, src=pc)
    return params, cur, last, truth


def make_local_map(F, scene_id=0, n=2000, related_frac=0.8, flip_p=0.06, pos_noise=0.01):
    """A frame pose and local MapPoints for Tracking::SearchLocalPoints: most points are back-projections of F's keypoints (depth
    from the stereo disparity or drawn, position perturbed a little), with their observation normals, scale-invariance distances
    consistent with the keypoint octave, and noisy copies of the descriptors; the rest lie anywhere around the camera (behind it,
    outside the image, too far / too close, seen from behind).  Returns (Tcw float32 4x4, map-point dict)."""
    rng = np.random.default_rng(SEED_SEARCH + 0x5000 + scene_id)
    fx, fy, cx, cy, bf = [np.float64(np.float32(c)) for c in KITTI_CAM]
    w = _rodrigues(rng.normal(0, 0.2, 3)); t = rng.normal(0, 2.0, 3)
    T = np.eye(4); T[:3, :3] = w; T[:3, 3] = t
    T = T.astype(np.float32)
    R = T[:3, :3].astype(np.float64); tt = T[:3, 3].astype(np.float64); Ow = -R.T @ tt
    src = rng.integers(0, F.n, n)
    z = np.where(F.uright[src] > 0, bf / np.maximum(F.xy[src, 0] - F.uright[src], 0.5), rng.uniform(4, 60, n))
    Xc = np.stack([(F.xy[src, 0] - cx) * z / fx, (F.xy[src, 1] - cy) * z / fy, z], 1) + rng.normal(0, pos_noise, (n, 3)) * z[:, None]
    related = rng.random(n) < related_frac
    far = ~related
    Xc[far] = np.stack([rng.uniform(-40, 40, int(far.sum())), rng.uniform(-15, 15, int(far.sum())), rng.uniform(-20, 90, int(far.sum()))], 1)
    Xw = (R.T @ (Xc - tt).T).T
    view_dir = Xw - Ow; dist = np.linalg.norm(view_dir, axis=1)
    normal = view_dir / dist[:, None] + rng.normal(0, 0.25, (n, 3))
    normal[rng.random(n) < 0.08] *= -1.0                                   # seen from behind: the viewing-angle test rejects
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    scale = 1.2 ** F.octave[src].astype(np.float64)
    maxd = dist * scale * rng.uniform(0.9, 1.1, n)                          # mfMaxDistance = dist * scaleFactor^level at creation
    mind = maxd / 1.2 ** 7
    off = rng.random(n) < 0.06
    maxd[off] *= rng.choice([0.3, 4.0], int(off.sum()))                     # outside the scale-invariance band
    mind[off] = maxd[off] / 1.2 ** 7
    desc = _flip_bits(rng, F.desc[src], flip_p)
    desc[far] = rng.integers(0, 2 ** 32, (int(far.sum()), 8), dtype=np.uint64).astype(np.uint32)
    return T, dict(world_pos=Xw.astype(np.float32), normal=normal.astype(np.float32), max_distance=maxd.astype(np.float32),
                   min_distance=mind.astype(np.float32), desc=desc, has_obs=(rng.random(n) < 0.9).astype(np.uint8),
                   skip=(rng.random(n) < 0.1).astype(np.uint8), occupied=(rng.random(F.n) < 0.05).astype(np.uint8), src=src.astype(np.int32))


def make_tracking_scene(scene_id=0, n_kp=2000, n_map=2500, n_last=1200, rot_deg=0.25, trans=0.04, n_lines=300, n_map_lines=260, n_last_lines=140):
    """One consistent little world for the Tracking thread's per-frame sequence (lld_slam_amd/tracking.py): a frame with `n_kp` keypoints,
    its true pose, `n_map` local MapPoints most of which are back-projections of the keypoints (make_local_map), the first `n_last` of them
    also being the last frame's tracked points (with that frame's octaves / angles), and the motion model's prediction of the pose - the
    true pose moved by `rot_deg` degrees and `trans` metres.  Returns a dict."""
    rng = np.random.default_rng(SEED_SEARCH + 0x7000 + scene_id)
    F = make_orb_frame(200 + scene_id, n_kp)
    T, mp = make_local_map(F, 200 + scene_id, n_map, pos_noise=0.0004)          # MapPoints that re-project within a pixel: PoseOptimization keeps them
    mp = dict(mp, skip=np.zeros(n_map, np.uint8), occupied=np.zeros(F.n, np.uint8))      # a fresh frame: nothing held, nothing seen yet
    src = mp["src"][:n_last]
    ang = np.mod(F.angle[src] + rng.normal(0, 4.0, n_last), 360.0).astype(np.float32)
    last = dict(world_pos=mp["world_pos"][:n_last], valid=(rng.random(n_last) < 0.95).astype(np.uint8), octave=F.octave[src].astype(np.int32),
                angle=ang, desc=mp["desc"][:n_last], has_obs=mp["has_obs"][:n_last])
    w = rng.normal(size=3); w *= np.deg2rad(rot_deg) / np.linalg.norm(w)
    dt = rng.normal(size=3); dt *= trans / np.linalg.norm(dt)
    dT = np.eye(4); dT[:3, :3] = _rodrigues(w); dT[:3, 3] = dt
    Tg = (dT @ T.astype(np.float64)).astype(np.float32)
    sc = dict(frame=F, cam=KITTI_CAM, Tcw_true=T, Tcw_guess=Tg, pose_true=_tcw_to_qt(T.astype(np.float64)), pose_guess=_tcw_to_qt(Tg.astype(np.float64)),
              last=last, last_ids=np.arange(n_last), map_points=mp, map_ids=np.arange(n_map))
    if n_lines > 0:
        sc.update(make_tracking_lines(scene_id, T.astype(np.float64), n_map_lines, n_lines, n_last_lines, Tcw_guess=Tg.astype(np.float64)))
    return sc


def make_tracking_lines(scene_id, Tcw, n_map=260, n_cur=300, n_last=140, dim=72, related_frac=0.75, pixel_noise=0.4, desc_noise=0.05, no_partner_frac=0.1,
                        outlier_frac=0.12, Tcw_guess=None, trap_frac=0.12):
    """The line half of make_tracking_scene: `n_cur` stereo lines of the frame at the true pose Tcw (world -> camera), `n_map` MapLines of which
    the first `related_frac * min(n_map, n_cur)` project onto frame lines (a few with a pose-inconsistent 3D position that passes the 2-pixel
    association gate of the predicted pose only sometimes, and PoseOptimization must then throw out), the rest elsewhere or behind the camera.
    The first `n_last` MapLines are the last frame's (mLastFrame.mvpMapLines), all of them the local map's - with the same ids, so that
    TrackLocalMap meets the lines TrackWithMotionModel already tracked.  Returns dict(lines=..., last_lines=..., local_lines=...)."""
    rng = np.random.default_rng(SEED_SEARCH + 0x9000 + scene_id)
    fx, fy, cx, cy, bf = [float(np.float32(c)) for c in KITTI_CAM]
    b = float(np.float32(np.float32(bf) / np.float32(fx)))
    Rcw = Tcw[:3, :3]; tcw = Tcw[:3, 3]
    z = rng.uniform(3.0, 40.0, n_map)
    c = np.stack([(rng.uniform(60, 1180, n_map) - cx) * z / fx, (rng.uniform(30, 340, n_map) - cy) * z / fy, z], 1)
    behind = rng.random(n_map) < 0.04
    c[behind, 2] = -rng.uniform(1.0, 10.0, int(behind.sum()))
    d = rng.normal(size=(n_map, 3)); d[:, 2] *= 0.3; d /= np.linalg.norm(d, axis=1, keepdims=True)
    half = rng.uniform(0.3, 2.5, n_map)[:, None] * 0.5
    A, B = c - half * d, c + half * d                                        # camera frame of the TRUE pose

    def proj(X, shift):
        Xc = X - np.array([shift, 0, 0]); zz = np.maximum(Xc[:, 2], 0.3)
        return np.stack([fx * Xc[:, 0] / zz + cx, fy * Xc[:, 1] / zz + cy], 1)
    m = int(related_frac * min(n_map, n_cur))
    to_w = lambda X: (Rcw.T @ (X - tcw).T).T
    Aw, Bw = to_w(A), to_w(B)
    if Tcw_guess is not None:
        # traps: lines observed where the PREDICTED pose projects them - they pass AddLinesFrom's gate in TrackWithMotionModel and are
        # outliers of the PoseOptimization that follows (the discard of src/Tracking.cc:962-975 needs something to discard)
        trap = np.nonzero(rng.random(m) < trap_frac)[0]
        Rg = Tcw_guess[:3, :3]; tg = Tcw_guess[:3, 3]
        A = A.copy(); B = B.copy()
        A[trap] = (Rg @ Aw[trap].T).T + tg; B[trap] = (Rg @ Bw[trap].T).T + tg
    left = np.concatenate([proj(A[:m], 0.0), proj(B[:m], 0.0)], 1) + rng.normal(0, pixel_noise, (m, 4))
    sft = rng.uniform(-0.2, 0.2, (m, 2))
    Ar, Br = A[:m] + sft[:, :1] * (B[:m] - A[:m]), B[:m] + sft[:, 1:] * (B[:m] - A[:m])
    right = np.concatenate([proj(Ar, b), proj(Br, b)], 1) + rng.normal(0, pixel_noise, (m, 4))
    off = rng.random(m) < outlier_frac                                       # observed a little off: inside the association gate now and then, outside PoseOptimization's
    left[off] += rng.normal(0, 1.2, (int(off.sum()), 4)); right[off] += rng.normal(0, 1.2, (int(off.sum()), 4))

    def unrelated(n):
        p = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n)], 1)
        return np.concatenate([p, p + rng.normal(0, 40, (n, 2))], 1)
    left = np.concatenate([left, unrelated(n_cur - m)]); right = np.concatenate([right, unrelated(n_cur - m)])
    lo = rng.integers(0, 3, n_cur); ro = lo.copy(); flip = rng.random(n_cur) < 0.1; ro[flip] = rng.integers(0, 3, int(flip.sum()))
    dirw = (Bw - Aw) / np.linalg.norm(Bw - Aw, axis=1, keepdims=True)
    X0 = Aw - np.sum(Aw * dirw, axis=1, keepdims=True) * dirw
    dm = rng.normal(size=(n_map, dim)); dm /= np.linalg.norm(dm, axis=1, keepdims=True)
    dup = rng.integers(0, m, max(1, n_map // 12)); tgt = rng.integers(0, m, dup.size)
    dm[dup] = dm[tgt] + rng.normal(0, 0.01, (dup.size, dim))               # rivals for the same frame line
    dc = rng.normal(size=(n_cur, dim)); dc /= np.linalg.norm(dc, axis=1, keepdims=True)
    dc[:m] = dm[:m] + rng.normal(0, desc_noise, (m, dim))
    perm = rng.permutation(n_cur); rperm = rng.permutation(n_cur)
    inv_r = np.empty(n_cur, np.int64); inv_r[rperm] = np.arange(n_cur)
    line_matches = inv_r[perm].astype(np.int32)
    line_matches[rng.random(n_cur) < no_partner_frac] = -1
    # a detected KeyLine lies inside its image: no negative coordinates (lld_pose_problem marks "no right line" by a negative xs, as the
    # adapters of the reference's Frame do)
    left = np.maximum(left, 0.5); right = np.maximum(right, 0.5)
    lines = dict(left_lines=left[perm].astype(np.float32), left_octave=lo[perm].astype(np.int32), right_lines=right[rperm].astype(np.float32),
                 right_octave=ro[rperm].astype(np.int32), line_matches=line_matches, desc=dc[perm].astype(np.float32))
    order = rng.permutation(n_map)                                           # the map's own order: not the frame's
    ids = (1000 + np.arange(n_map)).astype(np.int32)

    def pick(sel, skip_frac):
        return dict(X0=X0[sel], dir=dirw[sel], X1=Aw[sel], X2=Bw[sel], desc=dm[sel].astype(np.float32), id=ids[sel],
                    skip=(rng.random(len(sel)) < skip_frac).astype(np.uint8))
    return dict(lines=lines, last_lines=pick(order[:n_last], 0.05), local_lines=pick(rng.permutation(n_map), 0.03))
