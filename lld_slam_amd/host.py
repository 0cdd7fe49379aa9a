"""Host-side mirror of the reference interface over the C ABI (Python flavour, used by tests and bench).

Names follow the reference (`Optimizer::LocalBundleAdjustment`, `Optimizer::PoseOptimization`,
`ORBmatcher`, `TwoFrameLineMatcher`); the arguments are the flat windows of include/lld_amd.h instead
of live KeyFrame/MapPoint/MapLine objects (INTEGRATION.md shows the gather/scatter adapter).

The marshalling helpers (`ba_call`, `pose_call`, ...) take the library object as an argument so the
test-only oracle loader can reuse them; the public classes below always use the HIP library and raise
when it is missing or no GPU is present.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import abi
from .abi import (BAParams, BAResult, BAStats, BAWindow, Camera, PoseParams, PoseProblem, PoseResult, as_f64,
                  as_i32, _p)


# ------------------------------------------------------------------ flat problem containers
@dataclass
class Window:
    """One local-BA window (see lld_ba_window)."""
    cam: tuple                      # fx, fy, cx, cy, bf
    n_free_cams: int
    cam_qt: np.ndarray              # [n_cams,7]
    pt_xyz: np.ndarray              # [n_points,3]
    pt_obs_start: np.ndarray        # [n_points+1]
    pt_obs_cam: np.ndarray          # [n_pt_obs]
    pt_obs_uvr: np.ndarray          # [n_pt_obs,3]
    pt_obs_inv_sigma2: np.ndarray   # [n_pt_obs]
    line_x0: np.ndarray             # [n_lines,3]
    line_dir: np.ndarray            # [n_lines,3]
    ln_obs_start: np.ndarray        # [n_lines+1]
    ln_obs_cam: np.ndarray          # [n_ln_obs]
    ln_obs_left: np.ndarray         # [n_ln_obs,4]
    ln_obs_right: np.ndarray        # [n_ln_obs,4]
    ln_obs_octave: np.ndarray       # [n_ln_obs,2]
    meta: dict = field(default_factory=dict)

    def normalise(self):
        self.cam_qt = as_f64(self.cam_qt, (-1, 7))
        self.pt_xyz = as_f64(self.pt_xyz, (-1, 3))
        self.pt_obs_start = as_i32(self.pt_obs_start)
        self.pt_obs_cam = as_i32(self.pt_obs_cam)
        self.pt_obs_uvr = as_f64(self.pt_obs_uvr, (-1, 3))
        self.pt_obs_inv_sigma2 = as_f64(self.pt_obs_inv_sigma2)
        self.line_x0 = as_f64(self.line_x0, (-1, 3))
        self.line_dir = as_f64(self.line_dir, (-1, 3))
        self.ln_obs_start = as_i32(self.ln_obs_start)
        self.ln_obs_cam = as_i32(self.ln_obs_cam)
        self.ln_obs_left = as_f64(self.ln_obs_left, (-1, 4))
        self.ln_obs_right = as_f64(self.ln_obs_right, (-1, 4))
        self.ln_obs_octave = as_i32(self.ln_obs_octave).reshape(-1, 2)
        return self

    @property
    def n_cams(self): return self.cam_qt.shape[0]
    @property
    def n_points(self): return self.pt_xyz.shape[0]
    @property
    def n_lines(self): return self.line_x0.shape[0]
    @property
    def n_pt_obs(self): return self.pt_obs_cam.shape[0]
    @property
    def n_ln_obs(self): return self.ln_obs_cam.shape[0]

    def n_edges(self):
        """g2o edges this window expands to: one per point observation, 1-2 per line observation."""
        return int(self.n_pt_obs + self.n_ln_obs + np.count_nonzero(self.ln_obs_right[:, 0] >= 0))

    def to_c(self) -> BAWindow:
        self.normalise()
        w = BAWindow()
        w.cam = Camera(*[float(v) for v in self.cam])
        w.n_cams = self.n_cams; w.n_free_cams = int(self.n_free_cams)
        w.cam_qt = _p(self.cam_qt, C.c_double)
        w.n_points = self.n_points
        w.pt_xyz = _p(self.pt_xyz, C.c_double)
        w.pt_obs_start = _p(self.pt_obs_start, C.c_int32)
        w.n_pt_obs = self.n_pt_obs
        w.pt_obs_cam = _p(self.pt_obs_cam, C.c_int32)
        w.pt_obs_uvr = _p(self.pt_obs_uvr, C.c_double)
        w.pt_obs_inv_sigma2 = _p(self.pt_obs_inv_sigma2, C.c_double)
        w.n_lines = self.n_lines
        w.line_x0 = _p(self.line_x0, C.c_double)
        w.line_dir = _p(self.line_dir, C.c_double)
        w.ln_obs_start = _p(self.ln_obs_start, C.c_int32)
        w.n_ln_obs = self.n_ln_obs
        w.ln_obs_cam = _p(self.ln_obs_cam, C.c_int32)
        w.ln_obs_left = _p(self.ln_obs_left, C.c_double)
        w.ln_obs_right = _p(self.ln_obs_right, C.c_double)
        w.ln_obs_octave = _p(self.ln_obs_octave, C.c_int32)
        return w


@dataclass
class BAOutput:
    cam_qt: np.ndarray
    pt_xyz: np.ndarray
    line_x0: np.ndarray
    line_dir: np.ndarray
    pt_obs_outlier: np.ndarray
    ln_edge_outlier: np.ndarray
    line_removed: np.ndarray
    stats: dict

    @staticmethod
    def alloc(win: Window):
        return BAOutput(np.zeros((win.n_cams, 7)), np.zeros((win.n_points, 3)), np.zeros((win.n_lines, 3)),
                        np.zeros((win.n_lines, 3)), np.zeros(win.n_pt_obs, np.uint8),
                        np.zeros((win.n_ln_obs, 2), np.uint8), np.zeros(win.n_lines, np.uint8), {})

    def to_c(self) -> BAResult:
        r = BAResult()
        r.cam_qt = _p(self.cam_qt, C.c_double); r.pt_xyz = _p(self.pt_xyz, C.c_double)
        r.line_x0 = _p(self.line_x0, C.c_double); r.line_dir = _p(self.line_dir, C.c_double)
        r.pt_obs_outlier = _p(self.pt_obs_outlier, C.c_uint8)
        r.ln_edge_outlier = _p(self.ln_edge_outlier, C.c_uint8)
        r.line_removed = _p(self.line_removed, C.c_uint8)
        return r


def stats_dict(s: BAStats) -> dict:
    return dict(chi2_round1=s.chi2_round1, chi2_final=s.chi2_final, lm_iterations=list(s.lm_iterations),
                lm_trials=list(s.lm_trials), pcg_iterations=s.pcg_iterations, n_pt_obs_outlier=s.n_pt_obs_outlier,
                n_ln_edge_outlier=s.n_ln_edge_outlier, n_lines_removed=s.n_lines_removed, aborted=s.aborted)


def ba_params(lib: abi.Lib, gamma=1.0, **kw) -> BAParams:
    p = BAParams()
    lib.fn("ba_params_default")(C.byref(p))
    p.gamma = gamma
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@dataclass
class Sim3Pair:
    """Inputs of Optimizer::OptimizeSim3 for one (KF1, KF2) candidate: the correspondences that pass the loop's tests, in order."""
    K1: tuple                    # fx, fy, cx, cy of pKF1->mK
    K2: tuple
    s12_q: np.ndarray            # g2oS12 rotation (x, y, z, w)
    s12_t: np.ndarray
    s12_s: float
    p1c: np.ndarray              # [n,3] MapPoints of KF1 in KF1's camera frame
    p2c: np.ndarray              # [n,3] MapPoints of KF2 in KF2's camera frame
    obs1: np.ndarray             # [n,2]
    obs2: np.ndarray             # [n,2]
    inv_sigma2_1: np.ndarray     # [n]
    inv_sigma2_2: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def n(self): return int(np.asarray(self.p1c).shape[0])

    def to_c(self):
        keep = [np.ascontiguousarray(a, np.float64) for a in (self.p1c, self.p2c, self.obs1, self.obs2, self.inv_sigma2_1, self.inv_sigma2_2)]
        P = abi.Sim3Problem()
        P.fx1, P.fy1, P.cx1, P.cy1 = [float(v) for v in self.K1]; P.fx2, P.fy2, P.cx2, P.cy2 = [float(v) for v in self.K2]
        for i in range(4): P.s12_q[i] = float(self.s12_q[i])
        for i in range(3): P.s12_t[i] = float(self.s12_t[i])
        P.s12_s = float(self.s12_s); P.n = self.n
        P.p1c, P.p2c, P.obs1, P.obs2, P.inv_sigma2_1, P.inv_sigma2_2 = [a.ctypes.data_as(abi.c_double_p) for a in keep]
        P._keep = keep
        return P


@dataclass
class Sim3Output:
    s12_q: np.ndarray
    s12_t: np.ndarray
    s12_s: float
    dropped: np.ndarray
    n_inliers: int
    n_bad_first: int
    lm_iterations: list
    lm_trials: list
    chi2: float


def sim3_params(lib: abi.Lib, th2=10.0, bFixScale=True, **kw) -> abi.Sim3Params:
    p = abi.Sim3Params()
    lib.fn("sim3_params_default")(C.byref(p))
    p.th2 = float(np.float32(th2)); p.fix_scale = 1 if bFixScale else 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def sim3_pack(pairs: list):
    """C structs + result buffers of a list of Sim3Pair (shared with the test-only oracle loader, which calls one pair at a time)."""
    cs = [p.to_c() for p in pairs]
    drops = [np.zeros(max(1, p.n), np.uint8) for p in pairs]
    res = (abi.Sim3Result * len(pairs))()
    for r, d in zip(res, drops): r.dropped = d.ctypes.data_as(abi.c_uint8_p)
    return cs, drops, res


def sim3_unpack(pairs, drops, res) -> list:
    return [Sim3Output(np.array(r.s12_q[:]), np.array(r.s12_t[:]), float(r.s12_s), d[:p.n].copy(), int(r.n_inliers), int(r.n_bad_first),
                       list(r.lm_iterations), list(r.lm_trials), float(r.chi2)) for r, d, p in zip(res, drops, pairs)]


def sim3_call(lib: abi.Lib, ctx, pairs: list, params: abi.Sim3Params) -> list:
    """lld_optimize_sim3_batch: all candidates in one launch."""
    cs, drops, res = sim3_pack(pairs)
    arr = (abi.Sim3Problem * len(pairs))(*cs)
    fn = lib.fn("optimize_sim3_batch"); fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.Sim3Problem), C.POINTER(abi.Sim3Params), C.POINTER(abi.Sim3Result)]; fn.restype = C.c_int
    check(fn(ctx, len(pairs), arr, C.byref(params), res), "optimize_sim3_batch")
    return sim3_unpack(pairs, drops, res)


@dataclass
class EssentialGraph:
    """The pose graph of Optimizer::OptimizeEssentialGraph: Sim3 vertices (Siw as x, y, z, w, tx, ty, tz, s), EdgeSim3 list."""
    sim3: np.ndarray             # [n,8]
    fixed: np.ndarray            # [n] uint8
    edge_i: np.ndarray           # [e] int32
    edge_j: np.ndarray
    edge_sji: np.ndarray         # [e,8]
    meta: dict = field(default_factory=dict)


@dataclass
class EssentialGraphOutput:
    sim3: np.ndarray
    chi2: float
    lm_iterations: int
    lm_trials: int
    pcg_iterations: int
    solver_used: int = 0          # 1 dense Cholesky on the matrix cores, 2 matrix-free PCG


def essential_graph_call(lib: abi.Lib, ctx, g: EssentialGraph, bFixScale=True, **kw) -> EssentialGraphOutput:
    keep = [np.ascontiguousarray(g.sim3, np.float64).reshape(-1, 8), np.ascontiguousarray(g.fixed, np.uint8), np.ascontiguousarray(g.edge_i, np.int32),
            np.ascontiguousarray(g.edge_j, np.int32), np.ascontiguousarray(g.edge_sji, np.float64).reshape(-1, 8)]
    G = abi.PoseGraph(); G.n_vertices = keep[0].shape[0]; G.n_edges = keep[2].shape[0]
    G.sim3 = keep[0].ctypes.data_as(abi.c_double_p); G.fixed = keep[1].ctypes.data_as(abi.c_uint8_p)
    G.edge_i = keep[2].ctypes.data_as(abi.c_int32_p); G.edge_j = keep[3].ctypes.data_as(abi.c_int32_p); G.edge_sji = keep[4].ctypes.data_as(abi.c_double_p)
    P = abi.PoseGraphParams(); lib.fn("pose_graph_params_default")(C.byref(P)); P.fix_scale = 1 if bFixScale else 0
    for k, v in kw.items(): setattr(P, k, v)
    o = np.zeros_like(keep[0]); R = abi.PoseGraphResult(); R.sim3 = o.ctypes.data_as(abi.c_double_p)
    fn = lib.fn("optimize_essential_graph"); fn.argtypes = [C.c_void_p, C.POINTER(abi.PoseGraph), C.POINTER(abi.PoseGraphParams), C.POINTER(abi.PoseGraphResult)]; fn.restype = C.c_int
    check(fn(ctx, C.byref(G), C.byref(P), C.byref(R)), "optimize_essential_graph")
    return EssentialGraphOutput(o, float(R.chi2), int(R.lm_iterations), int(R.lm_trials), int(R.pcg_iterations), int(R.solver_used))


def pose_params(lib: abi.Lib, gamma=0.5, **kw) -> PoseParams:
    p = PoseParams()
    lib.fn("pose_params_default")(C.byref(p))
    p.gamma = gamma
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def check(status: int, what: str):
    if status != abi.LLD_OK:
        raise RuntimeError(f"{what} failed with status {status}")


def ba_call(lib: abi.Lib, ctx, win: Window, params: BAParams | None = None, abort: bool = False) -> BAOutput:
    cw = win.to_c()
    out = BAOutput.alloc(win)
    cr = out.to_c()
    if params is None:
        params = ba_params(lib)
    flag = C.c_int(1 if abort else 0)
    check(lib.fn("local_ba")(ctx, C.byref(cw), C.byref(params), C.byref(flag), C.byref(cr)), "local_ba")
    out.stats = stats_dict(cr.stats)
    return out


@dataclass
class PoseFrame:
    """One PoseOptimization problem (see lld_pose_problem)."""
    cam: tuple
    pose_qt: np.ndarray
    pt_xw: np.ndarray
    pt_uvr: np.ndarray
    pt_inv_sigma2: np.ndarray
    ln_x0: np.ndarray
    ln_dir: np.ndarray
    ln_left: np.ndarray
    ln_right: np.ndarray
    ln_octave: np.ndarray
    ln_frame_index: np.ndarray | None = None     # [n_lines] index of each line in the frame's mvLinesLeft; None = 0..n_lines-1
    meta: dict = field(default_factory=dict)

    def normalise(self):
        self.pose_qt = as_f64(self.pose_qt, (7,))
        self.pt_xw = as_f64(self.pt_xw, (-1, 3)); self.pt_uvr = as_f64(self.pt_uvr, (-1, 3))
        self.pt_inv_sigma2 = as_f64(self.pt_inv_sigma2)
        self.ln_x0 = as_f64(self.ln_x0, (-1, 3)); self.ln_dir = as_f64(self.ln_dir, (-1, 3))
        self.ln_left = as_f64(self.ln_left, (-1, 4)); self.ln_right = as_f64(self.ln_right, (-1, 4))
        self.ln_octave = as_i32(self.ln_octave).reshape(-1, 2)
        if self.ln_frame_index is not None:
            self.ln_frame_index = as_i32(self.ln_frame_index)
        return self

    @property
    def n_points(self): return self.pt_xw.shape[0]
    @property
    def n_lines(self): return self.ln_x0.shape[0]

    def to_c(self) -> PoseProblem:
        self.normalise()
        p = PoseProblem()
        p.cam = Camera(*[float(v) for v in self.cam])
        for i in range(7):
            p.pose_qt[i] = float(self.pose_qt[i])
        p.n_points = self.n_points
        p.pt_xw = _p(self.pt_xw, C.c_double); p.pt_uvr = _p(self.pt_uvr, C.c_double)
        p.pt_inv_sigma2 = _p(self.pt_inv_sigma2, C.c_double)
        p.n_lines = self.n_lines
        p.ln_x0 = _p(self.ln_x0, C.c_double); p.ln_dir = _p(self.ln_dir, C.c_double)
        p.ln_left = _p(self.ln_left, C.c_double); p.ln_right = _p(self.ln_right, C.c_double)
        p.ln_octave = _p(self.ln_octave, C.c_int32)
        p.ln_frame_index = None if self.ln_frame_index is None else _p(self.ln_frame_index, C.c_int32)
        return p


@dataclass
class PoseOutput:
    pose_qt: np.ndarray
    n_inliers: int
    pt_outlier: np.ndarray
    ln_outlier: np.ndarray
    lm_iterations: int
    lm_trials: int
    chi2: float


def pose_result_alloc(n_points, n_lines):
    po = np.zeros(n_points, np.uint8); lo = np.zeros(n_lines, np.uint8)
    r = PoseResult()
    r.pt_outlier = _p(po, C.c_uint8); r.ln_outlier = _p(lo, C.c_uint8)
    return r, po, lo


def pose_call(lib: abi.Lib, ctx, frame: PoseFrame, params: PoseParams | None = None) -> PoseOutput:
    cp = frame.to_c()
    r, po, lo = pose_result_alloc(frame.n_points, frame.n_lines)
    if params is None:
        params = pose_params(lib)
    check(lib.fn("pose_opt")(ctx, C.byref(cp), C.byref(params), C.byref(r)), "pose_opt")
    return PoseOutput(np.array(list(r.pose_qt)), r.n_inliers, po, lo, r.lm_iterations, r.lm_trials, r.chi2)


def hamming_call(lib: abi.Lib, ctx, q: np.ndarray, t: np.ndarray, mask: np.ndarray | None = None):
    q = np.ascontiguousarray(q, np.uint32).reshape(-1, 8); t = np.ascontiguousarray(t, np.uint32).reshape(-1, 8)
    nq, nt = q.shape[0], t.shape[0]
    bi = np.empty(nq, np.int32); bd = np.empty(nq, np.int32); si = np.empty(nq, np.int32); sd = np.empty(nq, np.int32)
    mp = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8).reshape(nq, nt)
        mp = _p(mask, C.c_uint8)
    check(lib.fn("match_hamming256")(ctx, _p(q, C.c_uint32), nq, _p(t, C.c_uint32), nt, mp, _p(bi, C.c_int32),
                                     _p(bd, C.c_int32), _p(si, C.c_int32), _p(sd, C.c_int32)), "match_hamming256")
    return bi, bd, si, sd


def hamming_csr_call(lib: abi.Lib, ctx, q, t, cand_start, cand_idx):
    q = np.ascontiguousarray(q, np.uint32).reshape(-1, 8); t = np.ascontiguousarray(t, np.uint32).reshape(-1, 8)
    cs = as_i32(cand_start); ci = as_i32(cand_idx)
    if ci.size == 0:
        ci = np.zeros(1, np.int32)
    nq, nt = q.shape[0], t.shape[0]
    bi = np.empty(nq, np.int32); bd = np.empty(nq, np.int32); si = np.empty(nq, np.int32); sd = np.empty(nq, np.int32)
    check(lib.fn("match_hamming256_csr")(ctx, _p(q, C.c_uint32), nq, _p(t, C.c_uint32), nt, _p(cs, C.c_int32),
                                         _p(ci, C.c_int32), _p(bi, C.c_int32), _p(bd, C.c_int32), _p(si, C.c_int32),
                                         _p(sd, C.c_int32)), "match_hamming256_csr")
    return bi, bd, si, sd


def l2_call(lib: abi.Lib, ctx, q: np.ndarray, t: np.ndarray, mask: np.ndarray | None = None):
    q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
    nq, dim = q.shape; nt = t.shape[0]
    bi = np.empty(nq, np.int32); bd = np.empty(nq, np.float64); si = np.empty(nq, np.int32); sd = np.empty(nq, np.float64)
    mp = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8).reshape(nq, nt)
        mp = _p(mask, C.c_uint8)
    check(lib.fn("match_l2f32")(ctx, _p(q, C.c_float), nq, _p(t, C.c_float), nt, dim, mp, _p(bi, C.c_int32),
                                _p(bd, C.c_double), _p(si, C.c_int32), _p(sd, C.c_double)), "match_l2f32")
    return bi, bd, si, sd


def greedy_call(lib: abi.Lib, ctx, dl: np.ndarray, dr: np.ndarray, gate: np.ndarray | None, tau: float):
    dl = np.ascontiguousarray(dl, np.float32); dr = np.ascontiguousarray(dr, np.float32)
    nq, dim = dl.shape; nt = dr.shape[0]
    m = np.empty(nq, np.int32); d = np.empty(nq, np.float64)
    gp = None
    if gate is not None:
        gate = np.ascontiguousarray(gate, np.uint8).reshape(nq, nt)
        gp = _p(gate, C.c_uint8)
    check(lib.fn("line_match_greedy")(ctx, _p(dl, C.c_float), nq, _p(dr, C.c_float), nt, dim, gp, float(tau),
                                      _p(m, C.c_int32), _p(d, C.c_double)), "line_match_greedy")
    return m, d


def line_stereo_call(lib: abi.Lib, ctx, K, b, tau, min_line_length, left_lines, left_octave, dl, right_lines, right_octave, dr,
                     is_stereo=True, want_gate=False):
    """lld_line_match_stereo: CheckLinePair's gates on the device, then the greedy assignment."""
    dl = np.ascontiguousarray(dl, np.float32); dr = np.ascontiguousarray(dr, np.float32)
    ll = np.ascontiguousarray(left_lines, np.float32).reshape(-1, 4); rl = np.ascontiguousarray(right_lines, np.float32).reshape(-1, 4)
    lo = np.ascontiguousarray(left_octave, np.int32); ro = np.ascontiguousarray(right_octave, np.int32)
    nq, nt = ll.shape[0], rl.shape[0]
    dim = dl.shape[1] if dl.ndim == 2 else dr.shape[1]
    P = abi.LineStereoParams()
    for i, v in enumerate(np.asarray(K, np.float64).reshape(9)):
        P.K[i] = float(v)
    P.b = float(b); P.tau = float(tau); P.min_line_length = int(min_line_length); P.is_stereo = int(is_stereo)
    m = np.empty(nq, np.int32); d = np.empty(nq, np.float64)
    gate = np.empty((nq, nt), np.uint8) if want_gate else None
    check(lib.fn("line_match_stereo")(ctx, C.byref(P), _p(ll, C.c_float), _p(lo, C.c_int32), _p(dl, C.c_float), nq, _p(rl, C.c_float),
                                      _p(ro, C.c_int32), _p(dr, C.c_float), nt, dim, _p(m, C.c_int32), _p(d, C.c_double),
                                      None if gate is None else _p(gate, C.c_uint8)), "line_match_stereo")
    return (m, d, gate) if want_gate else (m, d)


def line_track_call(lib: abi.Lib, ctx, K, T_curr, b, thr_reproj_base, md_thr, sx, sy, map_x0, map_dir, map_x1, map_x2, map_skip, map_desc,
                    left_lines, left_octave, right_lines, line_matches, occupied, cur_desc, monocular=False, use_grid=True, want_gate=False):
    """lld_line_track_match: Tracking::AddLinesFrom with the Hough-cell candidates and the reprojection gates on the device."""
    P = abi.LineTrackParams()
    for i, v in enumerate(np.asarray(K, np.float64).reshape(9)):
        P.K[i] = float(v)
    for i, v in enumerate(np.asarray(T_curr, np.float64).reshape(16)):
        P.T_curr[i] = float(v)
    P.b = float(b); P.thr_reproj_base = float(thr_reproj_base); P.md_thr = float(md_thr); P.sx = float(sx); P.sy = float(sy)
    P.monocular = int(monocular); P.use_grid = int(use_grid)
    x0 = np.ascontiguousarray(map_x0, np.float64).reshape(-1, 3); dr = np.ascontiguousarray(map_dir, np.float64).reshape(-1, 3)
    x1 = np.ascontiguousarray(map_x1, np.float64).reshape(-1, 3); x2 = np.ascontiguousarray(map_x2, np.float64).reshape(-1, 3)
    n_map = x0.shape[0]
    sk = None if map_skip is None else np.ascontiguousarray(map_skip, np.uint8)
    md = np.ascontiguousarray(map_desc, np.float32); cd = np.ascontiguousarray(cur_desc, np.float32)
    ll = np.ascontiguousarray(left_lines, np.float32).reshape(-1, 4); lo = np.ascontiguousarray(left_octave, np.int32)
    rl = np.ascontiguousarray(right_lines, np.float32).reshape(-1, 4); lm = np.ascontiguousarray(line_matches, np.int32)
    oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    n_cur, n_right = ll.shape[0], rl.shape[0]
    dim = md.shape[1] if md.ndim == 2 else cd.shape[1]
    m = np.empty(n_map, np.int32); d = np.empty(n_map, np.float64)
    gate = np.empty((n_map, n_cur), np.uint8) if want_gate else None
    check(lib.fn("line_track_match")(ctx, C.byref(P), n_map, _p(x0, C.c_double), _p(dr, C.c_double), _p(x1, C.c_double), _p(x2, C.c_double),
                                     None if sk is None else _p(sk, C.c_uint8), _p(md, C.c_float), n_cur, _p(ll, C.c_float), _p(lo, C.c_int32), n_right,
                                     _p(rl, C.c_float), _p(lm, C.c_int32), None if oc is None else _p(oc, C.c_uint8), _p(cd, C.c_float), dim,
                                     _p(m, C.c_int32), _p(d, C.c_double), None if gate is None else _p(gate, C.c_uint8)), "line_track_match")
    return (m, d, gate) if want_gate else (m, d)


def line_lastkf_call(lib: abi.Lib, ctx, K, T_curr, T_last, b, thr_reproj_base, md_thr, sx, sy, cur: dict, last: dict, use_grid=True):
    """lld_line_match_last_frame: Tracking::MatchLinesLastKF.  cur / last: left_lines, right_lines, line_matches, desc, plus occupied (cur)
    and left_octave, skip (last).  Returns (match_last, created, X0, dir)."""
    P = abi.LineLastKfParams()
    for i, v in enumerate(np.asarray(K, np.float64).reshape(9)):
        P.K[i] = float(v)
    for i, v in enumerate(np.asarray(T_curr, np.float64).reshape(16)):
        P.T_curr[i] = float(v)
    for i, v in enumerate(np.asarray(T_last, np.float64).reshape(16)):
        P.T_last[i] = float(v)
    P.b = float(b); P.thr_reproj_base = float(thr_reproj_base); P.md_thr = float(md_thr); P.sx = float(sx); P.sy = float(sy); P.use_grid = int(use_grid)
    cl = np.ascontiguousarray(cur["left_lines"], np.float32).reshape(-1, 4); cr = np.ascontiguousarray(cur["right_lines"], np.float32).reshape(-1, 4)
    clm = np.ascontiguousarray(cur["line_matches"], np.int32); cd = np.ascontiguousarray(cur["desc"], np.float32)
    co = None if cur.get("occupied") is None else np.ascontiguousarray(cur["occupied"], np.uint8)
    ll = np.ascontiguousarray(last["left_lines"], np.float32).reshape(-1, 4); lr = np.ascontiguousarray(last["right_lines"], np.float32).reshape(-1, 4)
    lo = np.ascontiguousarray(last["left_octave"], np.int32); llm = np.ascontiguousarray(last["line_matches"], np.int32)
    ls = None if last.get("skip") is None else np.ascontiguousarray(last["skip"], np.uint8); ld = np.ascontiguousarray(last["desc"], np.float32)
    n_cur = cl.shape[0]; dim = cd.shape[1] if cd.ndim == 2 else ld.shape[1]
    m = np.empty(n_cur, np.int32); cre = np.empty(n_cur, np.uint8); x0 = np.zeros((n_cur, 3)); dr = np.zeros((n_cur, 3))
    check(lib.fn("line_match_last_frame")(ctx, C.byref(P), n_cur, _p(cl, C.c_float), cr.shape[0], _p(cr, C.c_float), _p(clm, C.c_int32),
                                          None if co is None else _p(co, C.c_uint8), _p(cd, C.c_float), ll.shape[0], _p(ll, C.c_float), _p(lo, C.c_int32),
                                          lr.shape[0], _p(lr, C.c_float), _p(llm, C.c_int32), None if ls is None else _p(ls, C.c_uint8), _p(ld, C.c_float), dim,
                                          _p(m, C.c_int32), _p(cre, C.c_uint8), _p(x0, C.c_double), _p(dr, C.c_double)), "line_match_last_frame")
    return m, cre, x0, dr


def se3_from_tcw_f32(lib: abi.Lib, tcw: np.ndarray) -> np.ndarray:
    t = np.ascontiguousarray(tcw, np.float32).reshape(16)
    out = np.zeros(7)
    lib.fn("se3_from_tcw_f32")(_p(t, C.c_float), _p(out, C.c_double))
    return out


def se3_to_tcw_f32(lib: abi.Lib, qt: np.ndarray) -> np.ndarray:
    q = as_f64(qt, (7,))
    out = np.zeros(16, np.float32)
    lib.fn("se3_to_tcw_f32")(_p(q, C.c_double), _p(out, C.c_float))
    return out.reshape(4, 4)


def orb_inv_level_sigma2(lib: abi.Lib, scale_factor=1.2, n_levels=8) -> np.ndarray:
    out = np.zeros(n_levels, np.float32)
    lib.fn("orb_inv_level_sigma2")(C.c_float(scale_factor), n_levels, _p(out, C.c_float))
    return out


# ------------------------------------------------------------------ product-side API (HIP only)
class Context:
    """One lld_ctx: a HIP device + stream.  Raises when no GPU is present (no CPU fallback)."""

    def __init__(self, device: int = 0, lib: "abi.Lib | None" = None):
        self.lib = lib if lib is not None else abi.product()      # `lib`: another build of the same ABI (the experiments build in two tests)
        h = C.c_void_p()
        st = self.lib.fn("ctx_create")(device, C.byref(h))
        if st != abi.LLD_OK:
            msg = self.lib.fn("status_string")(st).decode()
            raise RuntimeError(f"lld_ctx_create(device={device}) failed: {msg} (status {st})")
        self.handle = h
        self.device = device

    def stream(self) -> int:
        return int(self.lib.fn("ctx_stream")(self.handle) or 0)

    def synchronize(self):
        check(self.lib.fn("ctx_synchronize")(self.handle), "ctx_synchronize")

    def release_cache(self):
        """Give back the slab / pinned arenas the batched local BA keeps on the context (lld_ctx_release_cache)."""
        check(self.lib.fn("ctx_release_cache")(self.handle), "ctx_release_cache")

    def close(self):
        if self.handle:
            self.lib.fn("ctx_destroy")(self.handle)
            self.handle = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()


class Optimizer:
    """Mirror of the reference's `Optimizer` static interface (include/Optimizer.h:49-50)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.lib = ctx.lib

    def LocalBundleAdjustment(self, window: Window, pbStopFlag: bool = False, gamma: float = 1.0, **params) -> BAOutput:
        return ba_call(self.lib, self.ctx.handle, window, ba_params(self.lib, gamma, **params), pbStopFlag)

    def GlobalBundleAdjustment(self, window: Window, nIterations: int = 5, pbStopFlag: bool = False, bRobust: bool = True, **params) -> BAOutput:
        """Optimizer::GlobalBundleAdjustment / BundleAdjustment (src/Optimizer.cc:312-559): the window is the whole map (every
        keyframe but mnId==0 free), ONE optimize(nIterations), no outlier handling, identity line information."""
        return ba_call(self.lib, self.ctx.handle, window,
                       ba_params(self.lib, 1.0, protocol=1, its_round1=nIterations, robust_points=1 if bRobust else 0, **params), pbStopFlag)

    def OptimizeEssentialGraph(self, graph, bFixScale: bool = True, **params):
        """Optimizer::OptimizeEssentialGraph (src/Optimizer.cc:1391-1654), the optimisation proper: 15 LM iterations on the Sim3 pose graph."""
        return essential_graph_call(self.lib, self.ctx.handle, graph, bFixScale, **params)

    def OptimizeSim3(self, pair, th2: float = 10.0, bFixScale: bool = True, **params):
        """Optimizer::OptimizeSim3 (src/Optimizer.cc:1656-1851); a list of pairs goes through one launch (one workgroup per candidate)."""
        many = isinstance(pair, (list, tuple))
        outs = sim3_call(self.lib, self.ctx.handle, list(pair) if many else [pair], sim3_params(self.lib, th2, bFixScale, **params))
        return outs if many else outs[0]

    def PoseOptimization(self, frame: PoseFrame, gamma: float = 1.0, **params) -> PoseOutput:
        return pose_call(self.lib, self.ctx.handle, frame, pose_params(self.lib, gamma, **params))


class BABatch:
    """HBM-resident batch of windows (lld_ba_batch_*): upload once, solve many times."""

    def __init__(self, ctx: Context, windows: list[Window], gamma: float = 1.0, **params):
        self.ctx = ctx; self.lib = ctx.lib; self.windows = windows
        self._c = (BAWindow * len(windows))(*[w.to_c() for w in windows])
        self.params = ba_params(self.lib, gamma, **params)
        h = C.c_void_p()
        check(self.lib.fn("ba_batch_create")(ctx.handle, len(windows), self._c, C.byref(self.params), C.byref(h)),
              "ba_batch_create")
        self.handle = h

    def solve(self, abort: bool = False):
        flag = C.c_int(1 if abort else 0)
        check(self.lib.fn("ba_batch_solve")(self.handle, C.byref(flag)), "ba_batch_solve")

    def solve_with_flag(self, flag: "C.c_int"):
        """`flag` is the live pbStopFlag: another host thread may raise it while this call runs (ctypes releases the GIL)."""
        check(self.lib.fn("ba_batch_solve")(self.handle, C.byref(flag)), "ba_batch_solve")

    def download(self, i: int) -> BAOutput:
        out = BAOutput.alloc(self.windows[i]); cr = out.to_c()
        check(self.lib.fn("ba_batch_download")(self.handle, i, C.byref(cr)), "ba_batch_download")
        out.stats = stats_dict(cr.stats)
        return out

    def download_all(self) -> list[BAOutput]:
        """Every window's result through one lld_ba_batch_download_range call (unpacked by several host threads)."""
        outs = [BAOutput.alloc(w) for w in self.windows]
        arr = (BAResult * len(outs))(*[o.to_c() for o in outs])
        check(self.lib.fn("ba_batch_download_range")(self.handle, 0, len(outs), arr), "ba_batch_download_range")
        for o, r in zip(outs, arr):
            o.stats = stats_dict(r.stats)
        return outs

    def stats(self) -> list[dict]:
        arr = (BAStats * len(self.windows))()
        check(self.lib.fn("ba_batch_stats")(self.handle, arr), "ba_batch_stats")
        return [stats_dict(s) for s in arr]

    def set_phase_timing(self, on: bool = True):
        """HIP events at the phase boundaries of every super-step of the following solves (phase_ms()[0:5], kernel_stats()[1]); off by default:
        an event between two dependent kernels costs ~4 us of device time."""
        check(self.lib.fn("ba_batch_set_phase_timing")(self.handle, 1 if on else 0), "ba_batch_set_phase_timing")

    def phase_ms(self) -> np.ndarray:
        ms = np.zeros(6)
        check(self.lib.fn("ba_batch_phase_ms")(self.handle, _p(ms, C.c_double)), "ba_batch_phase_ms")
        return ms

    def kernel_stats(self, kernel: int = 1):
        """(launches, total ms) of one kernel family of the last solve: 0 linearise, 1 Schur, 2 PCG, 3 backsub, 4 control."""
        n = C.c_int64(0); ms = np.zeros(1)
        check(self.lib.fn("ba_batch_kernel_stats")(self.handle, kernel, C.byref(n), _p(ms, C.c_double)), "ba_batch_kernel_stats")
        return int(n.value), float(ms[0])

    def set_groups(self, n: int):
        """Window groups solved concurrently on separate streams (0 = default for the batch size)."""
        check(self.lib.fn("ba_batch_set_groups")(self.handle, int(n)), "ba_batch_set_groups")

    def result_records(self):
        p = C.c_void_p(); stride = C.c_uint64(0)
        check(self.lib.fn("ba_batch_result_records")(self.handle, C.byref(p), C.byref(stride)), "ba_batch_result_records")
        return int(p.value or 0), int(stride.value)

    def close(self):
        if self.handle:
            self.lib.fn("ba_batch_destroy")(self.handle)
            self.handle = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()


class PoseBatch:
    def __init__(self, ctx: Context, frames: list[PoseFrame], gamma: float = 0.5, **params):
        self.ctx = ctx; self.lib = ctx.lib; self.frames = frames
        self._c = (PoseProblem * len(frames))(*[f.to_c() for f in frames])
        self.params = pose_params(self.lib, gamma, **params)
        h = C.c_void_p()
        check(self.lib.fn("pose_batch_create")(ctx.handle, len(frames), self._c, C.byref(self.params), C.byref(h)),
              "pose_batch_create")
        self.handle = h

    def solve(self):
        check(self.lib.fn("pose_batch_solve")(self.handle), "pose_batch_solve")

    def download(self, i: int) -> PoseOutput:
        f = self.frames[i]
        r, po, lo = pose_result_alloc(f.n_points, f.n_lines)
        check(self.lib.fn("pose_batch_download")(self.handle, i, C.byref(r)), "pose_batch_download")
        return PoseOutput(np.array(list(r.pose_qt)), r.n_inliers, po, lo, r.lm_iterations, r.lm_trials, r.chi2)

    def close(self):
        if self.handle:
            self.lib.fn("pose_batch_destroy")(self.handle)
            self.handle = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()


class ORBmatcher:
    """Distance + best/second-best core of the reference's ORBmatcher (include/ORBmatcher.h:41-83).

    TH_HIGH / TH_LOW / the ratio test are the reference's accept rules (src/ORBmatcher.cc:37-39, e.g. :118-121,
    :228-230) applied on the host to the kernel's (best, second) output.
    """
    TH_HIGH = 100
    TH_LOW = 50
    HISTO_LENGTH = 30

    def __init__(self, ctx: Context, nnratio: float = 0.6, checkOri: bool = True):
        self.ctx = ctx; self.lib = ctx.lib
        self.mfNNratio = np.float32(nnratio); self.mbCheckOrientation = checkOri

    def BestTwo(self, q, t, mask=None):
        return hamming_call(self.lib, self.ctx.handle, q, t, mask)

    def BestTwoCandidates(self, q, t, cand_start, cand_idx):
        return hamming_csr_call(self.lib, self.ctx.handle, q, t, cand_start, cand_idx)

    # ---- the reference's search routines, each ONE device call (candidate generation, skip rules, accept rules, the
    # ---- order-dependent occupancy and the rotation histogram all run in lld_orb_search_run); see orb_search.py
    def SearchByProjectionMap(self, F, mp_desc, in_view, proj, proj_xr, pred_level, view_cos, mp_obs, f_occupied, th=1.0):
        """SearchByProjection(Frame&, const vector<MapPoint*>&, th)  (src/ORBmatcher.cc:45-129)."""
        from . import orb_search as S
        return S.search_by_projection_map(self.lib, self.ctx.handle, F, mp_desc, in_view, proj, proj_xr, pred_level, view_cos, mp_obs,
                                          f_occupied, th, float(self.mfNNratio))

    def SearchLocalPoints(self, F, view, map_points: dict, f_occupied=None, th=1.0, viewing_cos_limit=0.5):
        """Tracking::SearchLocalPoints (src/Tracking.cc:1613-1664): Frame::isInFrustum + SearchByProjection, both on the device."""
        from . import orb_search as S
        return S.search_local_points(self.lib, self.ctx.handle, F, view, map_points, f_occupied, th, float(self.mfNNratio), viewing_cos_limit)

    def SearchLastFrame(self, Cur, view, last: dict, cur_occupied=None, direction=0, th=7.0):
        """SearchByProjection(Frame& Current, const Frame& Last, th, bMono) with the projection on the device too."""
        from . import orb_search as S
        return S.search_last_frame(self.lib, self.ctx.handle, Cur, view, last, cur_occupied, direction, th, self.mbCheckOrientation)

    def SearchByProjectionFrame(self, Cur, last_desc, valid, uv, ur, last_octave, last_angle, mp_obs, cur_occupied, direction=0, th=7.0):
        """SearchByProjection(Frame& Current, const Frame& Last, th, bMono)  (src/ORBmatcher.cc:1328-1470)."""
        from . import orb_search as S
        return S.search_by_projection_frame(self.lib, self.ctx.handle, Cur, last_desc, valid, uv, ur, last_octave, last_angle, mp_obs,
                                            cur_occupied, direction, th, self.mbCheckOrientation)

    def SearchByProjectionReloc(self, Cur, desc, valid, uv, pred_level, kf_angle, cur_occupied, th, ORBdist):
        """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)  (src/ORBmatcher.cc:1472-1599)."""
        from . import orb_search as S
        return S.search_by_projection_reloc(self.lib, self.ctx.handle, Cur, desc, valid, uv, pred_level, kf_angle, cur_occupied, th, ORBdist,
                                            self.mbCheckOrientation)

    def SearchByProjectionKF(self, KF, desc, valid, uv, pred_level, matched, th):
        """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)  (src/ORBmatcher.cc:290-403)."""
        from . import orb_search as S
        return S.search_by_projection_kf(self.lib, self.ctx.handle, KF, desc, valid, uv, pred_level, matched, th)

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        """SearchForInitialization(Frame&, Frame&, vbPrevMatched, vnMatches12, windowSize)  (src/ORBmatcher.cc:405-520):
        returns (nmatches, vnMatches12, the updated vbPrevMatched)."""
        from . import orb_search as S
        return S.search_for_initialization(self.lib, self.ctx.handle, F1, F2, vbPrevMatched, windowSize, float(self.mfNNratio), self.mbCheckOrientation)

    def Fuse(self, KF, desc, valid, uv, ur, pred_level, th=3.0):
        """Inner search of Fuse (src/ORBmatcher.cc:825-1100)."""
        from . import orb_search as S
        return S.fuse_search(self.lib, self.ctx.handle, KF, desc, valid, uv, ur, pred_level, th)

    def FusePoints(self, KF, view, map_points: dict, th=3.0):
        """Fuse(KeyFrame*, vpMapPoints, th) with the projection loop on the device as well (src/ORBmatcher.cc:825-958)."""
        from . import orb_search as S
        return S.fuse_search_points(self.lib, self.ctx.handle, KF, view, map_points, th)

    def SearchByProjectionRelocPoints(self, Cur, view, map_points: dict, kf_angle, cur_occupied, th, ORBdist):
        """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) with its projection loop on the device
        (src/ORBmatcher.cc:1472-1599).  Returns (SearchOutput, uv, level)."""
        from . import orb_search as S
        return S.search_projected(self.lib, self.ctx.handle, Cur, view, map_points, S.PROJ_RELOC, th, accept_max=ORBdist,
                                  check_orientation=self.mbCheckOrientation, angle=kf_angle, occupied=cur_occupied)

    def SearchByProjectionKFPoints(self, KF, view, map_points: dict, matched, th):
        """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) with its projection loop on the device
        (src/ORBmatcher.cc:290-403); `view` = orb_search.sim3_view(Scw, ...)."""
        from . import orb_search as S
        return S.search_projected(self.lib, self.ctx.handle, KF, view, map_points, S.PROJ_KF_SIM3, int(th), occupied=matched)

    def FuseSim3Points(self, KF, view, map_points: dict, th=4.0):
        """Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) with its projection loop on the device (src/ORBmatcher.cc:977-1100)."""
        from . import orb_search as S
        return S.search_projected(self.lib, self.ctx.handle, KF, view, map_points, S.PROJ_FUSE_SIM3, th)

    def SearchBySim3Points(self, KF1, view1, points1, KF2, view2, points2, s12, R12, t12, th=7.5):
        """SearchBySim3 (src/ORBmatcher.cc:1102-1326), projections and both searches on the device: (match12, nFound)."""
        from . import orb_search as S
        sR12, t12, sR21, t21 = S.sim3_transforms(s12, R12, t12)
        return S.search_by_sim3_points(self.lib, self.ctx.handle, KF1, view1, points1, KF2, view2, points2, sR12, t12, sR21, t21, th)

    def SearchBySim3(self, KF1, KF2, q1, q2, th=7.5):
        """SearchBySim3 (src/ORBmatcher.cc:1102-1326)."""
        from . import orb_search as S
        return S.search_by_sim3(self.lib, self.ctx.handle, KF1, KF2, q1, q2, th)

    def SearchByBoWFrame(self, KF, F, nodes, kf_valid):
        """SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)  (src/ORBmatcher.cc:159-288)."""
        from . import orb_search as S
        return S.search_by_bow_frame(self.lib, self.ctx.handle, KF, F, nodes["n_nodes"], nodes["start1"], nodes["idx1"], nodes["start2"],
                                     nodes["idx2"], kf_valid, float(self.mfNNratio), self.mbCheckOrientation)

    def SearchByBoWKF(self, KF1, KF2, nodes, valid1, valid2):
        """SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)  (src/ORBmatcher.cc:522-655)."""
        from . import orb_search as S
        return S.search_by_bow_kf(self.lib, self.ctx.handle, KF1, KF2, nodes["n_nodes"], nodes["start1"], nodes["idx1"], nodes["start2"],
                                  nodes["idx2"], valid1, valid2, float(self.mfNNratio), self.mbCheckOrientation)

    def SearchForTriangulation(self, KF1, KF2, nodes, has_mp1, has_mp2, epilines, epipole, bOnlyStereo=False):
        """SearchForTriangulation (src/ORBmatcher.cc:657-823)."""
        from . import orb_search as S
        return S.search_for_triangulation(self.lib, self.ctx.handle, KF1, KF2, nodes["n_nodes"], nodes["start1"], nodes["idx1"],
                                          nodes["start2"], nodes["idx2"], has_mp1, has_mp2, epilines, epipole, bOnlyStereo,
                                          self.mbCheckOrientation)

    def ComputeStereoMatches(self, L, R, min_d, max_d):
        """Hamming search of Frame::ComputeStereoMatches (src/Frame.cc:530-613)."""
        from . import orb_search as S
        return S.stereo_search(self.lib, self.ctx.handle, L, R, min_d, max_d)

    def ComputeStereoMatchesFull(self, L, R, left_levels, right_levels, inv_scale, mb, mbf):
        """Frame::ComputeStereoMatches incl. the SAD sub-pixel refinement and the median cut (src/Frame.cc:530-704)."""
        from . import orb_search as S
        return S.compute_stereo_matches(self.lib, self.ctx.handle, L, R, left_levels, right_levels, inv_scale, mb, mbf)

    def AcceptByRatio(self, best_idx, best_dist, second_dist, th):
        """`bestDist<=th` and `bestDist < mfNNratio*bestDist2` as in SearchByBoW (src/ORBmatcher.cc:226-230)."""
        ok = (best_idx >= 0) & (best_dist <= th) & (best_dist.astype(np.float32) < self.mfNNratio * second_dist.astype(np.float32))
        return np.where(ok, best_idx, -1)


class Tracking:
    """The line half of the Tracking thread that runs on the device: Tracking::AddLinesFrom (src/Tracking.cc:996-1124), the per-frame
    association of map lines with the lines of the current frame.  K, b (= mbf / fx), the image size and mdThr are the members the
    reference's Tracking object holds."""

    def __init__(self, ctx: "Context", K, b: float, mnMaxX: float, mnMaxY: float, mdThr: float = 2.0, monocular: bool = False):
        self.ctx = ctx; self.lib = ctx.lib
        self.K = np.asarray(K, np.float64).reshape(3, 3); self.b = b; self.sx = 1.0 / mnMaxX; self.sy = 1.0 / mnMaxY
        self.mdThr = mdThr; self.monocular = monocular

    def AddLinesFrom(self, lines_last: dict, T_curr, thrReprojLineBase: float, frame: dict, use_grid: bool = True, want_gate: bool = False):
        """lines_last: X0, dir (GetMinimalPos), X1, X2 (GetMainPoints3D), desc, skip; frame: left_lines, left_octave, right_lines,
        line_matches, occupied (mvpMapLines != NULL), desc.  Returns (matches, distances[, gates])."""
        return line_track_call(self.lib, self.ctx.handle, self.K, T_curr, self.b, thrReprojLineBase, self.mdThr, self.sx, self.sy,
                               lines_last["X0"], lines_last["dir"], lines_last["X1"], lines_last["X2"], lines_last.get("skip"), lines_last["desc"],
                               frame["left_lines"], frame["left_octave"], frame["right_lines"], frame["line_matches"], frame.get("occupied"),
                               frame["desc"], self.monocular, use_grid, want_gate)

    def MatchLinesLastKF(self, T_curr, T_last, current: dict, last: dict, thrReprojLineBase: float = 6.0, use_grid: bool = True):
        """Tracking::MatchLinesLastKF (src/Tracking.cc:1449-1611): per unmatched stereo line of the current frame the best line of the
        last frame under the Hough-cell / reprojection gates, then vgl::MultiTriangulateLine over the four views.  Returns
        (match_last, created, X0, line_dir); the reference constructs a MapLine(X0, line_dir) where created is set."""
        return line_lastkf_call(self.lib, self.ctx.handle, self.K, T_curr, T_last, self.b, thrReprojLineBase, self.mdThr, self.sx, self.sy,
                                current, last, use_grid)

    def HoughCells(self, lines):
        """The frame's line grid as this build fills it: cell index dist_ind * 50 + ang_ind per line (lld_line_hough_cells)."""
        ll = np.ascontiguousarray(lines, np.float32).reshape(-1, 4); cell = np.empty(ll.shape[0], np.int32)
        check(self.lib.fn("line_hough_cells")(_p(ll, C.c_float), ll.shape[0], self.sx, self.sy, _p(cell, C.c_int32)), "line_hough_cells")
        return cell


class TwoFrameLineMatcher:
    """Mirror of TwoFrameLineMatcher (include/TwoFrameLineMatcher.h:31-42).  Constructed like the reference
    (K, b, tau, minLineLength) it runs the whole MatchLines on the device, CheckLinePair's geometric gates included;
    constructed with tau only, the caller supplies the gates as a byte matrix."""

    def __init__(self, ctx: Context, tau: float, K=None, b: float = 0.0, minLineLength: int = 0):
        self.ctx = ctx; self.lib = ctx.lib; self.tau = tau
        self.K = None if K is None else np.asarray(K, np.float64).reshape(3, 3); self.b = b; self.minLineLength = minLineLength

    def MatchLines(self, descsLeft, descsRight, gate=None, lines=None, other_lines=None, octaves=None, other_octaves=None, want_gate=False):
        """MatchLines(lines, other_lines, descsLeft, descsRight, desc_matches): with `lines` [n,4] / `other_lines` [m,4]
        (startPointX, startPointY, endPointX, endPointY) and their octaves the gates are computed on the device."""
        if lines is not None:
            if self.K is None:
                raise ValueError("TwoFrameLineMatcher was constructed without K / b")
            return line_stereo_call(self.lib, self.ctx.handle, self.K, self.b, self.tau, self.minLineLength, lines, octaves, descsLeft,
                                    other_lines, other_octaves, descsRight, True, want_gate)
        return greedy_call(self.lib, self.ctx.handle, descsLeft, descsRight, gate, self.tau)

    def BestTwo(self, q, t, mask=None):
        return l2_call(self.lib, self.ctx.handle, q, t, mask)
