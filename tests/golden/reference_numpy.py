#!/usr/bin/env python3
"""An INDEPENDENT whole-protocol reference of Optimizer::LocalBundleAdjustment in numpy / scipy, written from SURVEY.md Appendix A.1-A.7
(not from oracle/): its purpose is to break the circle "device == oracle, both written from the same reading" with a second
implementation that shares no code and as few decisions as possible with the C++ oracle.

What is deliberately different from the oracle (and from the reference's own way of computing the same mathematics):
  * the linear system: the FULL damped normal equations over cameras + points + lines, dense, solved by scipy.linalg.cho_factor /
    cho_solve - no Schur complement, no per-landmark inverse, no back-substitution;
  * the Jacobians: complex-step differentiation of the residual DEFINITION under the DEFINITION of the two oplus updates
    (d e / d delta = Im e(i h) / h, h = 1e-30: exact to rounding, no subtraction) - not the hand-derived blocks of
    types_six_dof_expmap.cpp that the oracle restates;
  * the line depth test through numpy.linalg.lstsq; quaternions only at the state boundary (rotation matrices inside).
What must be the same, because it IS the algorithm (Appendix A.4-A.7): float casts of the stereo projection, Huber with rho' only,
lambda_0 = 1e-5 max diag, the rho / lambda / nBad rules, stale per-edge chi2 in the classification, the count <= 4 line rule.

    python tests/golden/reference_numpy.py            # regenerates tests/golden/independent_lba.npz (three small windows)

tests/test_oracle_independent.py holds the oracle to this file: one LM step and short protocols to rounding, the full protocol to
identical decisions and a bounded drift (see its docstring for why two correct implementations cannot do better over 20 iterations)."""
from __future__ import annotations

import os
import sys

import numpy as np
from scipy.linalg import cho_factor, cho_solve

H_STEP = 1e-30


# ---------------------------------------------------------------------------------------------------- A.1 SE3
def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_from_R(R):                       # Eigen's Quaterniond(Matrix3d)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if tr > 0:
        s = np.sqrt(tr + 1.0); q[3] = 0.5 * s; s = 0.5 / s
        q[0] = (R[2, 1] - R[1, 2]) * s; q[1] = (R[0, 2] - R[2, 0]) * s; q[2] = (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]: i = 1
        if R[2, 2] > R[i, i]: i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0); q[i] = 0.5 * s; s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    return q


def normalize_rotation(q):
    if q[3] < 0: q = -q
    return q / np.sqrt(q @ q)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def se3_exp(xi):
    om, up = xi[:3], xi[3:]
    th = np.sqrt(om @ om); Om = skew(om); O2 = Om @ Om
    if th < 1e-5:
        R = np.eye(3) + Om + O2; V = R                      # the reference's small-angle branch, literally
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / (th * th) * O2
        V = np.eye(3) + (1 - np.cos(th)) / (th * th) * Om + (th - np.sin(th)) / th ** 3 * O2
    return normalize_rotation(quat_from_R(R)), V @ up


def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def pose_oplus(qt, xi):                                      # T <- exp(xi) * T
    q1, t1 = se3_exp(xi)
    R1 = quat_to_R(q1)
    return np.concatenate([normalize_rotation(quat_mul(q1, qt[:4])), t1 + R1 @ qt[4:]])


# ---------------------------------------------------------------------------------------------------- A.2 landmarks
def line_init(X0, d):
    n = np.sqrt(X0 @ X0)
    R = np.stack([d, X0 / n, np.cross(d, X0) / n], 1)
    return np.concatenate([quat_from_R(R), [n]])


def line_R(l5):
    q = l5[:4] / np.sqrt(l5[:4] @ l5[:4])
    return quat_to_R(q)


def line_oplus(l5, d4):
    qr = np.array([d4[0], d4[1], d4[2], np.sqrt(1.0 - d4[:3] @ d4[:3])])
    q = l5[:4] / np.sqrt(l5[:4] @ l5[:4])
    return np.concatenate([quat_mul(qr, q), [l5[4] + d4[3]]])


# ---------------------------------------------------------------------------------------------------- A.3 residuals (definitions)
def point_residual(cam, R, t, X, obs, stereo, float_casts=True):
    fx, fy, cx, cy, bf = cam
    Xc = R @ X + t
    if stereo and float_casts:
        invz32 = np.float32(1.0 / Xc[2]); invz = float(invz32)
        u = Xc[0] * invz * fx + cx; v = Xc[1] * invz * fy + cy
        bfz = float(np.float32(bf) * invz32)                  # `const float& bf` times `const float invz`: a FLOAT product in C++ (types_six_dof_expmap.cpp:158-165)
        return np.array([obs[0] - u, obs[1] - v, obs[2] - (u - bfz)])
    invz = 1.0 / Xc[2]
    u = fx * Xc[0] * invz + cx; v = fy * Xc[1] * invz + cy
    return np.array([obs[0] - u, obs[1] - v, obs[2] - (u - bf * invz)]) if stereo else np.array([obs[0] - u, obs[1] - v])


def line_residual(f, cx, cy, bx, R, t, X1, X2, seg):
    K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1.0]])
    b = np.array([bx, 0, 0.0])
    P1 = K @ (R @ X1 + t + b); P2 = K @ (R @ X2 + t + b)
    lt = np.array([P1[1] * P2[2] - P1[2] * P2[1], P1[2] * P2[0] - P1[0] * P2[2], P1[0] * P2[1] - P1[1] * P2[0]])
    l = lt / np.sqrt(lt[0] * lt[0] + lt[1] * lt[1])
    return np.array([seg[0] * l[0] + seg[1] * l[1] + l[2], seg[2] * l[0] + seg[3] * l[1] + l[2]])


def line_depth_positive(f, cx, cy, bx, R, t, X0, d, seg):
    K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1.0]]); b = np.array([bx, 0, 0.0])
    X0l = R @ X0 + t + b; ldl = R @ (X0 + d) + t + b - X0l
    for px, py in (seg[:2], seg[2:]):
        A = np.stack([np.array([px, py, 1.0]), -(K @ ldl)], 1)
        sol = np.linalg.lstsq(A, K @ X0l, rcond=None)[0]
        if sol[0] < 0: return False
    return True


# Jacobians by complex-step differentiation of the definitions (first-order exp is exact for the derivative at zero)
def _cskew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=complex)


def point_jacobians(cam, R, t, X, stereo):
    D = 3 if stereo else 2
    Jp = np.zeros((D, 3)); Jc = np.zeros((D, 6)); obs = np.zeros(3)
    for k in range(3):
        d = np.zeros(3, complex); d[k] = 1j * H_STEP
        Jp[:, k] = point_residual(cam, R.astype(complex), t.astype(complex), X + d, obs, stereo, False).imag / H_STEP
    for k in range(6):
        xi = np.zeros(6, complex); xi[k] = 1j * H_STEP
        W = np.eye(3) + _cskew(xi[:3])
        Jc[:, k] = point_residual(cam, W @ R, W @ t + xi[3:], X.astype(complex), obs, stereo, False).imag / H_STEP
    return Jp, Jc


def _cquat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=complex)


def line_jacobians(f, cx, cy, bx, R, t, l5, seg):
    Jl = np.zeros((2, 4)); Jc = np.zeros((2, 6))
    q = (l5[:4] / np.sqrt(l5[:4] @ l5[:4])).astype(complex)
    for k in range(4):
        d = np.zeros(4, complex); d[k] = 1j * H_STEP
        qr = np.array([d[0], d[1], d[2], np.sqrt(1.0 - (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))])
        ax, ay, az, aw = qr; bx_, by, bz, bw = q
        q2 = np.array([aw * bx_ + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx_, aw * bz + ax * by - ay * bx_ + az * bw, aw * bw - ax * bx_ - ay * by - az * bz])
        q2 = q2 / np.sqrt(q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2] + q2[3] * q2[3])
        Rl = _cquat_R(q2); al = l5[4] + d[3]
        X1 = al * Rl[:, 1]; X2 = X1 + Rl[:, 0]
        Jl[:, k] = line_residual(f, cx, cy, bx, R.astype(complex), t.astype(complex), X1, X2, seg).imag / H_STEP
    Rl = line_R(l5); X1 = l5[4] * Rl[:, 1]; X2 = X1 + Rl[:, 0]
    for k in range(6):
        xi = np.zeros(6, complex); xi[k] = 1j * H_STEP
        W = np.eye(3) + _cskew(xi[:3])
        Jc[:, k] = line_residual(f, cx, cy, bx, W @ R, W @ t + xi[3:], X1.astype(complex), X2.astype(complex), seg).imag / H_STEP
    return Jl, Jc


# ---------------------------------------------------------------------------------------------------- A.4 kernel
def huber(chi2, delta):
    if chi2 <= delta * delta: return chi2, 1.0
    s = np.sqrt(chi2)
    return 2 * delta * s - delta * delta, delta / s


# ---------------------------------------------------------------------------------------------------- the protocol
def local_ba(w, gamma=1.0, its=(5, 15), ln_filter=4, max_trials=10):
    """w: a lld_slam_amd.host.Window (flat inputs).  Returns a dict with the final state, erase lists and the LM trajectory."""
    cam = tuple(float(v) for v in w.cam); fx, fy, cx, cy, bf = cam
    nf, nc, npt, nl = w.n_free_cams, w.n_cams, w.n_points, w.n_lines
    cams = [w.cam_qt[c].copy() for c in range(nc)]
    pts = [w.pt_xyz[p].copy() for p in range(npt)]
    lines = [line_init(w.line_x0[l], w.line_dir[l]) for l in range(nl)]
    d_mono, d_stereo = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
    stereo_b = float(np.float32(bf) / np.float32(fx))
    pe = []                                                            # point edges in insertion order
    for p in range(npt):
        for o in range(w.pt_obs_start[p], w.pt_obs_start[p + 1]):
            st = not (w.pt_obs_uvr[o, 2] < 0)
            pe.append(dict(cam=int(w.pt_obs_cam[o]), pt=p, obs=w.pt_obs_uvr[o].copy(), stereo=st, s=float(w.pt_obs_inv_sigma2[o]), level=0, robust=True,
                           delta=d_stereo if st else d_mono, err=None))
    le = []
    for l in range(nl):
        for o in range(w.ln_obs_start[l], w.ln_obs_start[l + 1]):
            has_right = not (w.ln_obs_right[o, 0] < 0)
            for si in range(2):
                if si == 1 and not has_right: continue
                seg = (w.ln_obs_left if si == 0 else w.ln_obs_right)[o].copy()
                thr = 1.44 ** int(w.ln_obs_octave[o, si])
                e = dict(cam=int(w.ln_obs_cam[o]), line=l, seg=seg, s=gamma * gamma / (thr * thr), bx=-stereo_b if si == 1 else 0.0, pair_stereo=has_right, level=0, robust=True,
                         delta=(d_stereo if has_right else d_mono) * gamma, err=None, obs=o, side=si, removed=False)
                le.append(e)
    trace = []                                                         # (round, iteration, trial, lambda used, chi2 of the trial, accepted)

    def cam_Rt(c): return quat_to_R(cams[c][:4]), cams[c][4:]
    def pe_err(e):
        R, t = cam_Rt(e["cam"]); e["err"] = point_residual(cam, R, t, pts[e["pt"]], e["obs"], e["stereo"])
    def le_err(e):
        R, t = cam_Rt(e["cam"]); Rl = line_R(lines[e["line"]]); X1 = lines[e["line"]][4] * Rl[:, 1]
        e["err"] = line_residual(fx, cx, cy, e["bx"], R, t, X1, X1 + Rl[:, 0], e["seg"])
    def chi2_of(e): return e["s"] * float(e["err"] @ e["err"])
    for e in le: le_err(e)                                             # e->computeError() before addEdge (LineOptimizer.cc:114)

    def optimize(n_its, rnd):
        act_pe = [e for e in pe if e["level"] == 0]; act_le = [e for e in le if e["level"] == 0 and not e["removed"]]
        if not act_pe and not act_le: return None
        # index mapping: free cameras that have an active edge, then active points, then active lines
        cam_on = sorted({e["cam"] for e in act_pe + act_le if e["cam"] < nf})
        pt_on = sorted({e["pt"] for e in act_pe}); ln_on = sorted({e["line"] for e in act_le})
        ci = {c: 6 * k for k, c in enumerate(cam_on)}; base_p = 6 * len(cam_on)
        pi = {p: base_p + 3 * k for k, p in enumerate(pt_on)}; base_l = base_p + 3 * len(pt_on)
        li = {l: base_l + 4 * k for k, l in enumerate(ln_on)}; n = base_l + 4 * len(ln_on)
        def errors():
            for e in act_pe: pe_err(e)
            for e in act_le: le_err(e)
        def robust_chi2():
            tot = 0.0
            for e in act_pe + act_le:
                c2 = chi2_of(e); tot += huber(c2, e["delta"])[0] if e["robust"] else c2
            return tot
        lam = ni = None; nbad = 0; chi = 0.0; iterations = trials = 0
        for it in range(n_its):
            errors(); chi = robust_chi2(); ini = chi
            H = np.zeros((n, n)); b = np.zeros(n)
            for e in act_pe:
                R, t = cam_Rt(e["cam"]); Jp, Jc = point_jacobians(cam, R, t, pts[e["pt"]], e["stereo"])
                wgt = huber(chi2_of(e), e["delta"])[1] if e["robust"] else 1.0
                blocks = [(pi[e["pt"]], Jp)] + ([(ci[e["cam"]], Jc)] if e["cam"] < nf else [])
                for ia, Ja in blocks:
                    b[ia:ia + Ja.shape[1]] += Ja.T @ (-wgt * e["s"] * e["err"])
                    for ib, Jb in blocks:
                        H[ia:ia + Ja.shape[1], ib:ib + Jb.shape[1]] += wgt * e["s"] * (Ja.T @ Jb)
            for e in act_le:
                R, t = cam_Rt(e["cam"]); Jl, Jc = line_jacobians(fx, cx, cy, e["bx"], R, t, lines[e["line"]], e["seg"])
                wgt = huber(chi2_of(e), e["delta"])[1] if e["robust"] else 1.0
                blocks = [(li[e["line"]], Jl)] + ([(ci[e["cam"]], Jc)] if e["cam"] < nf else [])
                for ia, Ja in blocks:
                    b[ia:ia + Ja.shape[1]] += Ja.T @ (-wgt * e["s"] * e["err"])
                    for ib, Jb in blocks:
                        H[ia:ia + Ja.shape[1], ib:ib + Jb.shape[1]] += wgt * e["s"] * (Ja.T @ Jb)
            if it == 0:
                lam = 1e-5 * np.max(np.abs(np.diag(H))); ni = 2.0; nbad = 0
            q = 0; rho = 0.0
            while True:
                backup = ([c.copy() for c in cams], [p.copy() for p in pts], [l.copy() for l in lines])
                try:
                    x = cho_solve(cho_factor(H + lam * np.eye(n), lower=True), b); ok = bool(np.all(np.isfinite(x)))
                except np.linalg.LinAlgError:
                    x = np.zeros(n); ok = False
                for c in cam_on: cams[c] = pose_oplus(cams[c], x[ci[c]:ci[c] + 6])
                for p in pt_on: pts[p] = pts[p] + x[pi[p]:pi[p] + 3]
                for l in ln_on: lines[l] = line_oplus(lines[l], x[li[l]:li[l] + 4])
                errors(); tmp = robust_chi2()
                if not ok: tmp = np.finfo(float).max
                scale = float(x @ (lam * x + b)) + 1e-3
                rho = (chi - tmp) / scale
                used = lam
                if rho > 0 and np.isfinite(tmp):
                    alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                    lam *= max(1.0 / 3.0, alpha); ni = 2.0; chi = tmp; accepted = True
                else:
                    lam *= ni; ni *= 2
                    cams[:], pts[:], lines[:] = backup
                    accepted = False
                q += 1; trials += 1
                trace.append((rnd, it, q, used, tmp, accepted))
                if not (rho < 0 and q < max_trials): break
            iterations += 1
            if q == max_trials or rho == 0: break
            nbad = nbad + 1 if (ini - chi) * 1e3 < ini else 0
            if nbad >= 3: break
        return dict(chi=chi, iterations=iterations, trials=trials)

    r1 = optimize(its[0], 0)
    # classification (A.7 step 4): per-edge chi2 is whatever the last evaluation left
    for e in pe:
        R, t = cam_Rt(e["cam"])
        if chi2_of(e) > (7.815 if e["stereo"] else 5.991) or not ((R @ pts[e["pt"]] + t)[2] > 0): e["level"] = 1
        e["robust"] = False
    count = {}
    for e in le:
        R, t = cam_Rt(e["cam"]); Rl = line_R(lines[e["line"]])
        ok = line_depth_positive(fx, cx, cy, e["bx"], R, t, lines[e["line"]][4] * Rl[:, 1], Rl[:, 0], e["seg"])
        count.setdefault(e["line"], 0)
        if chi2_of(e) > e["delta"] ** 2 or not ok: e["level"] = 1
        else: count[e["line"]] += 2
        e["robust"] = False
    removed = np.zeros(nl, np.uint8)
    for l, c in count.items():
        if c <= ln_filter: removed[l] = 1
    for e in le:
        if removed[e["line"]]: e["removed"] = True
    r2 = optimize(its[1], 1)
    # final classification (A.7 step 6)
    pt_out = np.zeros(len(pe), np.uint8)
    for k, e in enumerate(pe):
        R, t = cam_Rt(e["cam"])
        pt_out[k] = chi2_of(e) > (7.815 if e["stereo"] else 5.991) or not ((R @ pts[e["pt"]] + t)[2] > 0)
    ln_out = np.zeros((w.n_ln_obs, 2), np.uint8)
    for e in le:
        if removed[e["line"]]: continue
        R, t = cam_Rt(e["cam"]); Rl = line_R(lines[e["line"]])
        ok = line_depth_positive(fx, cx, cy, e["bx"], R, t, lines[e["line"]][4] * Rl[:, 1], Rl[:, 0], e["seg"])
        le_err(e)
        ln_out[e["obs"], e["side"]] = chi2_of(e) > e["delta"] ** 2 or not ok
    x0 = w.line_x0.copy(); dr = w.line_dir.copy()
    for l in range(nl):
        if removed[l]: continue
        Rl = line_R(lines[l]); dr[l] = Rl[:, 0]; x0[l] = lines[l][4] * Rl[:, 1]
    return dict(cam_qt=np.array(cams), pt_xyz=np.array(pts).reshape(-1, 3), line_x0=x0, line_dir=dr, pt_obs_outlier=pt_out, ln_edge_outlier=ln_out, line_removed=removed,
                chi2_round1=r1["chi"], chi2_final=(r2 or r1)["chi"], lm_iterations=[r1["iterations"], (r2 or dict(iterations=0))["iterations"]],
                lm_trials=[r1["trials"], (r2 or dict(trials=0))["trials"]], trace=np.array(trace, np.float64).reshape(-1, 6))


def normal_equations(w, gamma=1.0):
    """H, b (cameras, then points, then lines) and the robust chi2 at the INITIAL state of `w`, every edge active with its kernel on:
    what one call of buildSystem leaves behind (Appendix A.4)."""
    cam = tuple(float(v) for v in w.cam); fx, fy, cx, cy, bf = cam
    nf = w.n_free_cams
    d_mono, d_stereo = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
    stereo_b = float(np.float32(bf) / np.float32(fx))
    base_p = 6 * nf; base_l = base_p + 3 * w.n_points; n = base_l + 4 * w.n_lines
    H = np.zeros((n, n)); b = np.zeros(n); chi = 0.0
    def add(blocks, wgt, s, err):
        for ia, Ja in blocks:
            b[ia:ia + Ja.shape[1]] += Ja.T @ (-wgt * s * err)
            for ib, Jb in blocks:
                H[ia:ia + Ja.shape[1], ib:ib + Jb.shape[1]] += wgt * s * (Ja.T @ Jb)
    for p in range(w.n_points):
        for o in range(w.pt_obs_start[p], w.pt_obs_start[p + 1]):
            c = int(w.pt_obs_cam[o]); st = not (w.pt_obs_uvr[o, 2] < 0); s = float(w.pt_obs_inv_sigma2[o])
            R = quat_to_R(w.cam_qt[c][:4]); t = w.cam_qt[c][4:]
            err = point_residual(cam, R, t, w.pt_xyz[p], w.pt_obs_uvr[o], st)
            rho0, wgt = huber(s * float(err @ err), d_stereo if st else d_mono); chi += rho0
            Jp, Jc = point_jacobians(cam, R, t, w.pt_xyz[p], st)
            add([(base_p + 3 * p, Jp)] + ([(6 * c, Jc)] if c < nf else []), wgt, s, err)
    for l in range(w.n_lines):
        l5 = line_init(w.line_x0[l], w.line_dir[l]); Rl = line_R(l5); X1 = l5[4] * Rl[:, 1]
        for o in range(w.ln_obs_start[l], w.ln_obs_start[l + 1]):
            has_right = not (w.ln_obs_right[o, 0] < 0); c = int(w.ln_obs_cam[o])
            R = quat_to_R(w.cam_qt[c][:4]); t = w.cam_qt[c][4:]
            for si in range(2):
                if si == 1 and not has_right: continue
                seg = (w.ln_obs_left if si == 0 else w.ln_obs_right)[o]; thr = 1.44 ** int(w.ln_obs_octave[o, si]); s = gamma * gamma / (thr * thr)
                bx = -stereo_b if si == 1 else 0.0
                err = line_residual(fx, cx, cy, bx, R, t, X1, X1 + Rl[:, 0], seg)
                rho0, wgt = huber(s * float(err @ err), (d_stereo if has_right else d_mono) * gamma); chi += rho0
                Jl, Jc = line_jacobians(fx, cx, cy, bx, R, t, l5, seg)
                add([(base_l + 4 * l, Jl)] + ([(6 * c, Jc)] if c < nf else []), wgt, s, err)
    return H, b, chi


CASES = {     # name -> synth.make_lba_small arguments (small enough for dense normal equations in pure numpy)
    "tiny": dict(window_id=300, n_free=3, n_fixed=1, n_points=30, n_lines=6),
    "lines_and_mono": dict(window_id=301, n_free=5, n_fixed=2, n_points=90, n_lines=20, mono_frac=0.2, mono_line_frac=0.3, outlier_frac=0.1),
    "ten_cameras": dict(window_id=302, n_free=8, n_fixed=2, n_points=220, n_lines=36, outlier_frac=0.08),
}


def make_case(name):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path: sys.path.insert(0, root)
    from lld_slam_amd import synth
    return synth.make_lba_small(**CASES[name])


if __name__ == "__main__":
    out = {}
    for name in CASES:
        r = local_ba(make_case(name))
        print(name, "chi2 %.9g -> %.9g" % (r["chi2_round1"], r["chi2_final"]), "iterations", r["lm_iterations"], "trials", r["lm_trials"], "outliers", int(r["pt_obs_outlier"].sum()),
              int(r["ln_edge_outlier"].sum()), "removed", int(r["line_removed"].sum()), flush=True)
        for k, v in r.items():
            out[f"{name}__{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "independent_lba.npz"), **out)
