"""the analytic Jacobians of the line edge, derived SYMBOLICALLY from the definition of the residual and of the two
`oplus` updates, against what the oracle restates from the reference's hand-written code (types_six_dof_expmap.cpp:472-581).

Definition (types_six_dof_expmap.h:344-375, types_sba.h:62-108, se3quat.h:223-257):
  line      q (unit quaternion, R = R(q)), alpha:   X1 = alpha R.col(1),  X2 = X1 + R.col(0)
  camera    T = (qc, t):                            Xkc = R(qc) Xk + t
  image     l~ = K (X1c + b) x K (X2c + b),  l = l~ / |(l~x, l~y)|,  K = [[f,0,cx],[0,f,cy],[0,0,1]],  b = (bx, 0, 0)
  residual  e = (x1 . l, x2 . l),  xk = (detected end point, 1)
  updates   line:  q <- (r, sqrt(1 - |r|^2)) * q,  alpha <- alpha + d          (VertexSBALine::oplusImpl)
            pose:  T <- exp(omega, upsilon) * T                                 (VertexSE3Expmap::oplusImpl)
The Jacobians are d e / d (r, d) and d e / d (omega, upsilon) at zero.  For the pose update the derivative at zero only sees the
first-order part of exp - R = I + [omega]x, t' = t + omega x t + upsilon - which both branches of SE3Quat::exp share.
No finite differences here: sympy differentiates the expression tree."""
import numpy as np
import pytest

sp = pytest.importorskip("sympy")


def _quat_R(x, y, z, w):
    return sp.Matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _quat_mul(a, b):                      # Hamilton product, components (x, y, z, w)
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz)


def _skew(v):
    return sp.Matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


@pytest.fixture(scope="module")
def symbolic():
    lq = sp.symbols("lx ly lz lw"); alpha = sp.Symbol("alpha")
    cq = sp.symbols("cx_ cy_ cz_ cw_"); ct = sp.symbols("tx ty tz")
    f, pcx, pcy, bx = sp.symbols("f pcx pcy bx")
    seg = sp.symbols("xs ys xe ye")
    r = sp.symbols("r1 r2 r3"); da = sp.Symbol("da"); om = sp.symbols("o1 o2 o3"); up = sp.symbols("u1 u2 u3")
    # updated line
    dq = (r[0], r[1], r[2], sp.sqrt(1 - (r[0] ** 2 + r[1] ** 2 + r[2] ** 2)))
    q2 = _quat_mul(dq, lq)
    n2 = sp.sqrt(sum(c * c for c in q2))
    Rl = _quat_R(*[c / n2 for c in q2])                       # GetQ() normalises on every read (types_sba.cpp:58-92)
    a2 = alpha + da
    X1 = a2 * Rl[:, 1]; X2 = X1 + Rl[:, 0]
    # updated camera (first-order exp is exact for the derivative at zero)
    Rc = _quat_R(*cq); t = sp.Matrix(ct)
    W = _skew(om)
    Rc2 = (sp.eye(3) + W) * Rc; t2 = t + W * t + sp.Matrix(up)
    K = sp.Matrix([[f, 0, pcx], [0, f, pcy], [0, 0, 1]]); b = sp.Matrix([bx, 0, 0])
    P1 = K * (Rc2 * X1 + t2 + b); P2 = K * (Rc2 * X2 + t2 + b)
    lt = P1.cross(P2)
    l = lt / sp.sqrt(lt[0] ** 2 + lt[1] ** 2)
    e = sp.Matrix([seg[0] * l[0] + seg[1] * l[1] + l[2], seg[2] * l[0] + seg[3] * l[1] + l[2]])
    upd = list(r) + [da] + list(om) + list(up)
    J = e.jacobian(upd).subs({s: 0 for s in upd})
    e0 = e.subs({s: 0 for s in upd})
    args = list(lq) + [alpha] + list(cq) + list(ct) + [f, pcx, pcy, bx] + list(seg)
    return sp.lambdify(args, [e0, J], modules="numpy", cse=True)


def test_line_edge_jacobians_equal_their_symbolic_derivation(oracle, symbolic):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(7)
    cam = (707.0912, 707.0912, 601.8873, 183.1104, 379.8145)
    worst = 0.0
    for it in range(200):
        # a line 5 - 40 m in front of a camera with a random pose, seen near the image; left (bx = 0) and right (bx = -b) images
        Rcw = Rotation.from_rotvec(rng.normal(0, 0.3, 3)); tcw = rng.normal(0, 3, 3)
        Xc0 = np.array([rng.uniform(-8, 8), rng.uniform(-2, 2), rng.uniform(5, 40)])
        d = rng.normal(size=3); d[2] *= 0.3; d /= np.linalg.norm(d)
        A = Rcw.inv().apply(Xc0 - tcw); dw = Rcw.inv().apply(d)
        X0 = A - (A @ dw) * dw
        l5 = oracle.line_from_x0_dir(X0, dw)
        if it % 3 == 0:
            l5 = oracle.line_oplus(l5, rng.normal(0, 0.05, 4))                  # after an update the stored quaternion is generic
        qt = np.concatenate([Rcw.as_quat(), tcw])
        bx = 0.0 if it % 2 == 0 else -float(np.float32(cam[4]) / np.float32(cam[0]))
        Pa, Pb = Rcw.apply(X0) + tcw, Rcw.apply(X0 + dw) + tcw
        px = lambda P: np.array([cam[0] * (P[0] + bx) / P[2] + cam[2], cam[0] * P[1] / P[2] + cam[3]])
        seg = np.concatenate([px(Pa), px(Pb)]) + rng.normal(0, 2.0, 4)
        e, Jl, Jc, _ = oracle.edge_line(cam, bx, qt, l5, seg)
        lq = l5[:4] / np.linalg.norm(l5[:4])
        e_s, J_s = symbolic(*lq, l5[4], *qt[:4], *qt[4:], cam[0], cam[2], cam[3], bx, *seg)
        e_s = np.array(e_s, float).reshape(2); J_s = np.array(J_s, float).reshape(2, 10)
        np.testing.assert_allclose(e, e_s, rtol=1e-10, atol=1e-9)
        scale = max(1.0, np.abs(J_s).max())
        np.testing.assert_allclose(Jl, J_s[:, :4], rtol=1e-9, atol=1e-9 * scale)
        np.testing.assert_allclose(Jc, J_s[:, 4:], rtol=1e-9, atol=1e-9 * scale)
        worst = max(worst, np.abs(np.concatenate([Jl, Jc], 1) - J_s).max() / scale)
    assert worst < 1e-9
