"""ORACLE — TEST INFRASTRUCTURE ONLY.

Loader for oracle/liblld_oracle.so (the CPU restatement of the reference algorithm, symbols `lldo_*`).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
lld_slam_amd/ does.  PARITY UNPINNED (see lld_oracle.cpp).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from lld_slam_amd import abi, host

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_FMA = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liblld_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("lld_oracle.cpp", "lldo_orbsearch.cpp", "lldo_linematch.cpp", "lldo_math.h", "lldo_edges.h", "lldo_lm.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "lld_amd.h"))
    if force or not os.path.exists(so) or not os.path.exists(os.path.join(_HERE, "liblld_oracle_fma.so")) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib_fma() -> abi.Lib:
    """The same sources compiled with -ffp-contract=fast (oracle/Makefile): used ONLY to measure how far rounding alone moves a result."""
    global _LIB_FMA
    if _LIB_FMA is None:
        build()
        _LIB_FMA = abi.Lib(os.path.join(_HERE, "liblld_oracle_fma.so"), "lldo_")
    return _LIB_FMA


def lib() -> abi.Lib:
    global _LIB
    if _LIB is None:
        _LIB = abi.Lib(build(), "lldo_")
        d = _LIB.dll
        dp = abi.c_double_p
        d.lldo_se3_exp.argtypes = [dp, dp]; d.lldo_se3_mul.argtypes = [dp, dp, dp]; d.lldo_se3_map.argtypes = [dp, dp, dp]
        d.lldo_se3_oplus.argtypes = [dp, dp, dp]; d.lldo_quat_to_R.argtypes = [dp, dp]; d.lldo_quat_from_R.argtypes = [dp, dp]
        d.lldo_huber.argtypes = [C.c_double, C.c_double, dp]
        d.lldo_line_from_x0_dir.argtypes = [dp, dp, dp]; d.lldo_line_oplus.argtypes = [dp, dp, dp]
        d.lldo_line_to_x0_dir.argtypes = [dp, dp, dp]
        cp = C.POINTER(abi.Camera)
        d.lldo_edge_point.argtypes = [cp, dp, dp, dp, C.c_int, dp, dp, dp]
        d.lldo_edge_point_posonly.argtypes = [cp, dp, dp, dp, C.c_int, dp, dp]
        d.lldo_edge_line.argtypes = [cp, C.c_double, dp, dp, dp, dp, dp, dp, C.POINTER(C.c_int)]
        d.lldo_edge_line_posonly.argtypes = [cp, C.c_double, dp, dp, dp, dp, dp, dp]
        d.lldo_reproject_line_point.argtypes = [dp, dp] + [C.c_double] * 5 + [dp, dp]
        d.lldo_descriptor_distance.argtypes = [abi.c_uint32_p, abi.c_uint32_p]; d.lldo_descriptor_distance.restype = C.c_int
        d.lldo_l2f32.argtypes = [abi.c_float_p, abi.c_float_p, C.c_int]; d.lldo_l2f32.restype = C.c_double
        d.lldo_ba_one_step.argtypes = [C.POINTER(abi.BAWindow), C.POINTER(abi.BAParams), C.c_double, dp, dp,
                                       C.POINTER(C.c_int), dp, dp]
        d.lldo_ba_one_step.restype = C.c_int
        for n in ("lldo_se3_exp", "lldo_se3_mul", "lldo_se3_map", "lldo_se3_oplus", "lldo_quat_to_R", "lldo_quat_from_R",
                  "lldo_huber", "lldo_line_from_x0_dir", "lldo_line_oplus", "lldo_line_to_x0_dir", "lldo_edge_point",
                  "lldo_edge_point_posonly", "lldo_edge_line", "lldo_edge_line_posonly", "lldo_reproject_line_point"):
            getattr(d, n).restype = None
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _dp(a):
    return a.ctypes.data_as(abi.c_double_p)


def cam_struct(cam):
    return abi.Camera(*[float(v) for v in cam])


# ---- fine-grained entry points for the known-answer tests
def se3_exp(u):
    u = _d(u); o = np.zeros(7); lib().dll.lldo_se3_exp(_dp(u), _dp(o)); return o


def se3_mul(a, b):
    a = _d(a); b = _d(b); o = np.zeros(7); lib().dll.lldo_se3_mul(_dp(a), _dp(b), _dp(o)); return o


def se3_map(qt, X):
    qt = _d(qt); X = _d(X); o = np.zeros(3); lib().dll.lldo_se3_map(_dp(qt), _dp(X), _dp(o)); return o


def se3_oplus(qt, u):
    qt = _d(qt); u = _d(u); o = np.zeros(7); lib().dll.lldo_se3_oplus(_dp(qt), _dp(u), _dp(o)); return o


def quat_to_R(q):
    q = _d(q); o = np.zeros(9); lib().dll.lldo_quat_to_R(_dp(q), _dp(o)); return o.reshape(3, 3)


def quat_from_R(R):
    R = _d(R).reshape(9); o = np.zeros(4); lib().dll.lldo_quat_from_R(_dp(R), _dp(o)); return o


def huber(delta, e):
    o = np.zeros(3); lib().dll.lldo_huber(float(delta), float(e), _dp(o)); return o


def line_from_x0_dir(X0, d):
    X0 = _d(X0); d = _d(d); o = np.zeros(5); lib().dll.lldo_line_from_x0_dir(_dp(X0), _dp(d), _dp(o)); return o


def line_oplus(l5, u4):
    l5 = _d(l5); u4 = _d(u4); o = np.zeros(5); lib().dll.lldo_line_oplus(_dp(l5), _dp(u4), _dp(o)); return o


def line_to_x0_dir(l5):
    l5 = _d(l5); a = np.zeros(3); b = np.zeros(3); lib().dll.lldo_line_to_x0_dir(_dp(l5), _dp(a), _dp(b)); return a, b


def edge_point(cam, qt, Xw, obs, stereo, jac=True):
    c = cam_struct(cam); qt = _d(qt); Xw = _d(Xw); obs = _d(obs)
    e = np.zeros(3); Jp = np.zeros(9); Jc = np.zeros(18)
    lib().dll.lldo_edge_point(C.byref(c), _dp(qt), _dp(Xw), _dp(obs), int(stereo), _dp(e), _dp(Jp) if jac else None,
                              _dp(Jc) if jac else None)
    D = 3 if stereo else 2
    return e[:D], Jp.reshape(3, 3)[:D], Jc.reshape(3, 6)[:D]


def edge_point_posonly(cam, qt, Xw, obs, stereo):
    c = cam_struct(cam); qt = _d(qt); Xw = _d(Xw); obs = _d(obs)
    e = np.zeros(3); Jc = np.zeros(18)
    lib().dll.lldo_edge_point_posonly(C.byref(c), _dp(qt), _dp(Xw), _dp(obs), int(stereo), _dp(e), _dp(Jc))
    D = 3 if stereo else 2
    return e[:D], Jc.reshape(3, 6)[:D]


def edge_line(cam, bx, qt, l5, seg):
    c = cam_struct(cam); qt = _d(qt); l5 = _d(l5); seg = _d(seg)
    e = np.zeros(2); Jl = np.zeros(8); Jc = np.zeros(12); ok = C.c_int(0)
    lib().dll.lldo_edge_line(C.byref(c), float(bx), _dp(qt), _dp(l5), _dp(seg), _dp(e), _dp(Jl), _dp(Jc), C.byref(ok))
    return e, Jl.reshape(2, 4), Jc.reshape(2, 6), bool(ok.value)


def edge_line_posonly(cam, bx, qt, X1, X2, seg):
    c = cam_struct(cam); qt = _d(qt); X1 = _d(X1); X2 = _d(X2); seg = _d(seg)
    e = np.zeros(2); Jc = np.zeros(12)
    lib().dll.lldo_edge_line_posonly(C.byref(c), float(bx), _dp(qt), _dp(X1), _dp(X2), _dp(seg), _dp(e), _dp(Jc))
    return e, Jc.reshape(2, 6)


def reproject_line_point(X0, ld, px, py, f, cx, cy):
    X0 = _d(X0); ld = _d(ld); a = np.zeros(1); b = np.zeros(1)
    lib().dll.lldo_reproject_line_point(_dp(X0), _dp(ld), px, py, f, cx, cy, _dp(a), _dp(b))
    return float(a[0]), float(b[0])


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint32); b = np.ascontiguousarray(b, np.uint32)
    return lib().dll.lldo_descriptor_distance(a.ctypes.data_as(abi.c_uint32_p), b.ctypes.data_as(abi.c_uint32_p))


def l2f32(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return lib().dll.lldo_l2f32(a.ctypes.data_as(abi.c_float_p), b.ctypes.data_as(abi.c_float_p), a.size)


def ba_one_step(win: host.Window, lam: float, gamma=1.0):
    cw = win.to_c(); p = host.ba_params(lib(), gamma)
    n = 6 * win.n_free_cams + 3 * win.n_points + 4 * win.n_lines
    x = np.zeros(n); b = np.zeros(n); nn = C.c_int(0); chi = np.zeros(1); md = np.zeros(1)
    st = lib().dll.lldo_ba_one_step(C.byref(cw), C.byref(p), float(lam), _dp(x), _dp(b), C.byref(nn), _dp(chi), _dp(md))
    return st, x[:nn.value], b[:nn.value], float(chi[0]), float(md[0])


# ---- protocol-level entry points
# LocalBundleAdjustment and PoseOptimization reach the oracle through lldo_local_ba_flat / lldo_pose_opt_flat (round 4): every array and
# count is a plain argument, taken from the caller's fields by THIS module's code; the ABI structs are filled in C++ from the header.
# Nothing of lld_slam_amd/host.py's marshalling (Window.to_c, PoseFrame.to_c, ba_call, pose_call - what the device side of every Python test
# goes through) is shared any more: a wrong stride or a swapped array there shows up as a device / oracle difference.
_I32P = C.POINTER(C.c_int32); _U8P = C.POINTER(C.c_uint8); _F64P = C.POINTER(C.c_double)
_BA_PARAM_ORDER = ("its_round1", "its_round2", "ln_filter", "max_trials", "protocol", "robust_points", "abort_after_trials")
_BA_DEVICE_ONLY = ("reduced_solver", "deterministic", "pcg_rel_tol", "pcg_max_iter")       # no meaning for the CPU restatement


def _f64(a, cols=None):
    a = np.array(a, dtype=np.float64, order="C", copy=True)           # (a private copy: the oracle never sees the caller's buffer)
    return a.reshape(-1, cols) if cols else a.reshape(-1)


def _i32(a, cols=None):
    a = np.array(a, dtype=np.int32, order="C", copy=True)
    return a.reshape(-1, cols) if cols else a.reshape(-1)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _local_ba_flat(the_lib, win, gamma, abort, params):
    unknown = [k for k in params if k not in _BA_PARAM_ORDER + _BA_DEVICE_ONLY]
    if unknown: raise TypeError(f"local_ba: unknown parameter(s) {unknown}")
    cam = _f64([float(v) for v in win.cam]); assert cam.size == 5
    cam_qt = _f64(win.cam_qt, 7); pt = _f64(win.pt_xyz, 3); x0 = _f64(win.line_x0, 3); dr = _f64(win.line_dir, 3)
    ps = _i32(win.pt_obs_start); pc = _i32(win.pt_obs_cam); uvr = _f64(win.pt_obs_uvr, 3); isg = _f64(win.pt_obs_inv_sigma2)
    ls = _i32(win.ln_obs_start); lc = _i32(win.ln_obs_cam); ll = _f64(win.ln_obs_left, 4); lr = _f64(win.ln_obs_right, 4); lo = _i32(win.ln_obs_octave, 2)
    n_cams, n_pt, n_ln, n_po, n_lo = cam_qt.shape[0], pt.shape[0], x0.shape[0], pc.size, lc.size
    assert ps.size == n_pt + 1 and ls.size == n_ln + 1 and uvr.shape[0] == n_po and isg.size == n_po and ll.shape[0] == n_lo and lr.shape[0] == n_lo and lo.shape[0] == n_lo
    out = host.BAOutput(np.zeros((n_cams, 7)), np.zeros((n_pt, 3)), np.zeros((n_ln, 3)), np.zeros((n_ln, 3)), np.zeros(n_po, np.uint8),
                        np.zeros((n_lo, 2), np.uint8), np.zeros(n_ln, np.uint8), {})
    st12 = np.zeros(12)
    fn = the_lib.dll.lldo_local_ba_flat
    fn.restype = C.c_int
    fn.argtypes = ([_F64P, C.c_int, C.c_int, _F64P, C.c_int, _F64P, _I32P, C.c_int, _I32P, _F64P, _F64P, C.c_int, _F64P, _F64P, _I32P, C.c_int, _I32P, _F64P, _F64P, _I32P,
                    C.c_double] + [C.c_int] * 8 + [_F64P, _F64P, _F64P, _F64P, _U8P, _U8P, _U8P, _F64P])
    rc = fn(_ptr(cam, _F64P), n_cams, int(win.n_free_cams), _ptr(cam_qt, _F64P), n_pt, _ptr(pt, _F64P), _ptr(ps, _I32P), n_po, _ptr(pc, _I32P), _ptr(uvr, _F64P), _ptr(isg, _F64P),
            n_ln, _ptr(x0, _F64P), _ptr(dr, _F64P), _ptr(ls, _I32P), n_lo, _ptr(lc, _I32P), _ptr(ll, _F64P), _ptr(lr, _F64P), _ptr(lo, _I32P),
            float(gamma), *[int(params.get(k, -1)) for k in _BA_PARAM_ORDER], 1 if abort else 0,
            _ptr(out.cam_qt, _F64P), _ptr(out.pt_xyz, _F64P), _ptr(out.line_x0, _F64P), _ptr(out.line_dir, _F64P), _ptr(out.pt_obs_outlier, _U8P),
            _ptr(out.ln_edge_outlier, _U8P), _ptr(out.line_removed, _U8P), _ptr(st12, _F64P))
    if rc != 0: raise RuntimeError(f"lldo_local_ba_flat failed with status {rc}")
    out.stats = dict(chi2_round1=float(st12[0]), chi2_final=float(st12[1]), lm_iterations=[int(st12[2]), int(st12[3])], lm_trials=[int(st12[4]), int(st12[5])],
                     pcg_iterations=int(st12[6]), n_pt_obs_outlier=int(st12[7]), n_ln_edge_outlier=int(st12[8]), n_lines_removed=int(st12[9]), aborted=int(st12[10]))
    return out


def local_ba(win: host.Window, gamma=1.0, abort=False, **params):
    return _local_ba_flat(lib(), win, gamma, abort, params)


def local_ba_traced(win: host.Window, gamma=1.0, **params):
    """local_ba plus the LM trajectory: [n_trials, 3] = (lambda used, robust chi2 of the trial, accepted)."""
    buf = np.zeros(3 * 512)
    d = lib().dll
    d.lldo_lm_trace.argtypes = [abi.c_double_p, C.c_int]; d.lldo_lm_trace.restype = C.c_int
    d.lldo_lm_trace(_dp(buf), 512)
    try:
        r = local_ba(win, gamma, **params)
    finally:
        n = d.lldo_lm_trace(None, 0)
    return r, buf[:3 * n].reshape(-1, 3).copy()


class lm_trace:
    """with lm_trace() as t: ...oracle calls...; t.rows -> [n_trials, 3] = (lambda used, robust chi2 of the trial, accepted) of every LM
    trial the calls inside ran (local_ba, pose_opt, optimize_sim3 all go through lldo_lm.h's lm_solve).  One thread at a time."""

    def __enter__(self):
        self.buf = np.zeros(3 * 2048)
        d = lib().dll
        d.lldo_lm_trace.argtypes = [abi.c_double_p, C.c_int]; d.lldo_lm_trace.restype = C.c_int
        d.lldo_lm_trace(_dp(self.buf), 2048)
        self.rows = None
        return self

    def __exit__(self, *a):
        n = lib().dll.lldo_lm_trace(None, 0)
        self.rows = self.buf[:3 * n].reshape(-1, 3).copy()


def last_classification_margin():
    """(between the rounds, final): min over the edges of |chi2 - threshold| / threshold in the last local_ba call of this process."""
    out = np.zeros(2)
    d = lib().dll
    d.lldo_last_classification_margin.argtypes = [abi.c_double_p]; d.lldo_last_classification_margin.restype = None
    d.lldo_last_classification_margin(_dp(out))
    return float(out[0]), float(out[1])


def set_classification_flip(which: int, on: bool, margin: float = 1e-6):
    """Test hook: classification `which` (0 between the rounds, 1 final) takes the OTHER side for edges whose chi2 lies within `margin`
    (relative) of their threshold.  Not thread-safe: a process-wide knob, reset it in a finally."""
    d = lib().dll
    d.lldo_set_classification_flip.argtypes = [C.c_int, C.c_int, C.c_double]; d.lldo_set_classification_flip.restype = None
    d.lldo_set_classification_flip(int(which), 1 if on else 0, float(margin))


def set_landmark_inverse(how: int):
    """0: (Hll + lambda I)^-1 by Gauss-Jordan with partial pivoting (default; the reference calls MatrixXd::inverse(), block_solver.hpp:391),
    1: the same inverse through a Cholesky factor - equal in exact arithmetic.  Not thread-safe: a process-wide test knob."""
    lib().dll.lldo_set_landmark_inverse(int(how))


def optimize_sim3(pair, th2=10.0, bFixScale=True, fma=False, **params):
    """Optimizer::OptimizeSim3, literal restatement (numeric Jacobians as g2o).  `fma=True`: the FMA-contracted build (sensitivity only)."""
    many = isinstance(pair, (list, tuple))
    pairs = list(pair) if many else [pair]
    L = lib_fma() if fma else lib()
    prm = host.sim3_params(L, th2, bFixScale, **params)
    cs, drops, res = host.sim3_pack(pairs)
    fn = L.fn("optimize_sim3"); fn.argtypes = [C.c_void_p, C.POINTER(abi.Sim3Problem), C.POINTER(abi.Sim3Params), C.POINTER(abi.Sim3Result)]; fn.restype = C.c_int
    for c, r in zip(cs, res):                                      # the oracle has no batch entry point: one candidate at a time
        host.check(fn(None, C.byref(c), C.byref(prm), C.byref(r)), "lldo_optimize_sim3")
    outs = host.sim3_unpack(pairs, drops, res)
    return outs if many else outs[0]


def optimize_essential_graph(graph, bFixScale=True, fma=False, **params):
    """`fma=True`: through the FMA-contracted build (rounding-sensitivity measurements only)."""
    return host.essential_graph_call(lib_fma() if fma else lib(), None, graph, bFixScale, **params)


def sim3_log(qts8):
    d = lib().dll
    d.lldo_sim3_log.argtypes = [abi.c_double_p, abi.c_double_p]; d.lldo_sim3_log.restype = None
    q = _d(qts8); o = np.zeros(7)
    d.lldo_sim3_log(_dp(q), _dp(o))
    return o


def sim3_exp(u7):
    d = lib().dll
    d.lldo_sim3_exp.argtypes = [abi.c_double_p, abi.c_double_p]; d.lldo_sim3_exp.restype = None
    u = _d(u7); o = np.zeros(8)
    d.lldo_sim3_exp(_dp(u), _dp(o))
    return o


def pose_opt(frame: host.PoseFrame, gamma=0.5, **params):
    """PoseOptimization through lldo_pose_opt_flat (see local_ba above: no marshalling shared with the device side)."""
    unknown = [k for k in params if k not in ("n_rounds", "its_per_round", "max_trials")]
    if unknown: raise TypeError(f"pose_opt: unknown parameter(s) {unknown}")
    cam = _f64([float(v) for v in frame.cam]); qt = _f64(frame.pose_qt); assert cam.size == 5 and qt.size == 7
    xw = _f64(frame.pt_xw, 3); uvr = _f64(frame.pt_uvr, 3); isg = _f64(frame.pt_inv_sigma2)
    x0 = _f64(frame.ln_x0, 3); dr = _f64(frame.ln_dir, 3); le = _f64(frame.ln_left, 4); ri = _f64(frame.ln_right, 4); oc = _i32(frame.ln_octave, 2)
    fi = None if frame.ln_frame_index is None else _i32(frame.ln_frame_index)
    n_pt, n_ln = xw.shape[0], x0.shape[0]
    assert uvr.shape[0] == n_pt and isg.size == n_pt and dr.shape[0] == n_ln and le.shape[0] == n_ln and ri.shape[0] == n_ln and oc.shape[0] == n_ln and (fi is None or fi.size == n_ln)
    o_qt = np.zeros(7); po = np.zeros(n_pt, np.uint8); lo = np.zeros(n_ln, np.uint8); st4 = np.zeros(4)
    fn = lib().dll.lldo_pose_opt_flat
    fn.restype = C.c_int
    fn.argtypes = [_F64P, _F64P, C.c_int, _F64P, _F64P, _F64P, C.c_int, _F64P, _F64P, _F64P, _F64P, _I32P, _I32P, C.c_double, C.c_int, C.c_int, C.c_int, _F64P, _U8P, _U8P, _F64P]
    rc = fn(_ptr(cam, _F64P), _ptr(qt, _F64P), n_pt, _ptr(xw, _F64P), _ptr(uvr, _F64P), _ptr(isg, _F64P), n_ln, _ptr(x0, _F64P), _ptr(dr, _F64P), _ptr(le, _F64P), _ptr(ri, _F64P),
            _ptr(oc, _I32P), None if fi is None else _ptr(fi, _I32P), float(gamma), int(params.get("n_rounds", -1)), int(params.get("its_per_round", -1)), int(params.get("max_trials", -1)),
            _ptr(o_qt, _F64P), _ptr(po, _U8P), _ptr(lo, _U8P), _ptr(st4, _F64P))
    if rc != 0: raise RuntimeError(f"lldo_pose_opt_flat failed with status {rc}")
    return host.PoseOutput(o_qt, int(st4[0]), po, lo, int(st4[1]), int(st4[2]), float(st4[3]))


def match_hamming256(q, t, mask=None):
    return host.hamming_call(lib(), None, q, t, mask)


def match_hamming256_csr(q, t, cs, ci):
    return host.hamming_csr_call(lib(), None, q, t, cs, ci)


def match_l2f32(q, t, mask=None):
    return host.l2_call(lib(), None, q, t, mask)


def line_match_greedy(dl, dr, gate, tau):
    return host.greedy_call(lib(), None, dl, dr, gate, tau)


def line_match_stereo(K, b, tau, min_line_length, left_lines, left_octave, dl, right_lines, right_octave, dr, want_gate=False):
    """TwoFrameLineMatcher::MatchLines, literal (oracle/lldo_linematch.cpp)."""
    return host.line_stereo_call(lib(), None, K, b, tau, min_line_length, left_lines, left_octave, dl, right_lines, right_octave, dr, True, want_gate)


def line_track_match(K, T_curr, b, thr_reproj_base, md_thr, sx, sy, lines_last, frame, monocular=False, use_grid=True, want_gate=False):
    """Tracking::AddLinesFrom, literal (oracle/lldo_linematch.cpp)."""
    return host.line_track_call(lib(), None, K, T_curr, b, thr_reproj_base, md_thr, sx, sy, lines_last["X0"], lines_last["dir"], lines_last["X1"],
                                lines_last["X2"], lines_last.get("skip"), lines_last["desc"], frame["left_lines"], frame["left_octave"],
                                frame["right_lines"], frame["line_matches"], frame.get("occupied"), frame["desc"], monocular, use_grid, want_gate)


def line_match_last_frame(K, T_curr, T_last, b, thr_reproj_base, md_thr, sx, sy, cur, last, use_grid=True):
    """Tracking::MatchLinesLastKF, literal (oracle/lldo_linematch.cpp)."""
    return host.line_lastkf_call(lib(), None, K, T_curr, T_last, b, thr_reproj_base, md_thr, sx, sy, cur, last, use_grid)


def multi_triangulate_line(Ts, lines):
    """vgl::MultiTriangulateLine: (ok, X0, dir) for n <= 4 views (Ts [n,4,4] camera-to-world, lines [n,3] normalised image lines)."""
    d = lib().dll
    d.lldo_multi_triangulate_line.argtypes = [C.c_int, abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p]; d.lldo_multi_triangulate_line.restype = C.c_int
    T = _d(np.asarray(Ts, np.float64).reshape(-1, 16)); l = _d(np.asarray(lines, np.float64).reshape(-1, 3)); x0 = np.zeros(3); dr = np.zeros(3)
    ok = d.lldo_multi_triangulate_line(T.shape[0], _dp(T), _dp(l), _dp(x0), _dp(dr))
    return bool(ok), x0, dr


def hough_coordinates(leq, sx, sy, step_dist=3, step_ang=3):
    """GetHoughCoordinates, literal: (dist_inds, ang_inds) in the order the reference pushes them."""
    d = lib().dll
    d.lldo_hough_coordinates.argtypes = [abi.c_double_p, C.c_double, C.c_double, C.c_int, C.c_int, abi.c_int32_p, abi.c_int32_p, abi.c_int32_p, abi.c_int32_p]
    d.lldo_hough_coordinates.restype = C.c_int
    l = _d(leq); di = np.zeros(64, np.int32); ai = np.zeros(64, np.int32); nd = np.zeros(1, np.int32); na = np.zeros(1, np.int32)
    d.lldo_hough_coordinates(_dp(l), sx, sy, step_dist, step_ang, di.ctypes.data_as(abi.c_int32_p), nd.ctypes.data_as(abi.c_int32_p),
                             ai.ctypes.data_as(abi.c_int32_p), na.ctypes.data_as(abi.c_int32_p))
    return di[:nd[0]].copy(), ai[:na[0]].copy()


def line_hough_cells(lines, sx, sy):
    d = lib().dll
    ll = np.ascontiguousarray(lines, np.float32).reshape(-1, 4); cell = np.empty(ll.shape[0], np.int32)
    d.lldo_line_hough_cells(ll.ctypes.data_as(abi.c_float_p), ll.shape[0], sx, sy, cell.ctypes.data_as(abi.c_int32_p))
    return cell


def line_pair_geometry(K, b, kl1, kl2):
    """vgl::TriangulateLine + ReprojectKeyLineTo3D for one pair: (ok, X0, dir, p1, p2)."""
    d = lib().dll
    d.lldo_line_pair_geometry.argtypes = [C.POINTER(abi.LineStereoParams), abi.c_float_p, abi.c_float_p, abi.c_double_p]
    d.lldo_line_pair_geometry.restype = C.c_int
    P = abi.LineStereoParams()
    for i, v in enumerate(np.asarray(K, np.float64).reshape(9)):
        P.K[i] = float(v)
    P.b = float(b); P.is_stereo = 1
    a = np.ascontiguousarray(kl1, np.float32); bb = np.ascontiguousarray(kl2, np.float32); out = np.zeros(12)
    ok = d.lldo_line_pair_geometry(C.byref(P), a.ctypes.data_as(abi.c_float_p), bb.ctypes.data_as(abi.c_float_p), _dp(out))
    return bool(ok), out[0:3], out[3:6], out[6:9], out[9:12]


def colpiv_qr_solve(A, b):
    """The restated Eigen::ColPivHouseholderQR: (rank, x)."""
    d = lib().dll
    d.lldo_colpiv_qr_solve.argtypes = [C.c_int, C.c_int, abi.c_double_p, abi.c_double_p, abi.c_double_p]; d.lldo_colpiv_qr_solve.restype = C.c_int
    A = _d(A); b = _d(b); x = np.zeros(A.shape[1])
    r = d.lldo_colpiv_qr_solve(A.shape[0], A.shape[1], _dp(A), _dp(b), _dp(x))
    return r, x
