"""The Tracking thread's per-frame sequence on one Frame, over the C ABI (host-side mirror; no oracle, no CPU fallback).

The reference runs, on ONE `Frame` (stereo): `ORBmatcher::SearchByProjection(Current, Last, th, bMono)` (src/Tracking.cc:904),
`Optimizer::PoseOptimization` (:937) and the outlier discard (:940-958) of `TrackWithMotionModel`; then `SearchLocalPoints` (:1133) and
`Optimizer::PoseOptimization` (:1152) of `TrackLocalMap`.  `TrackedFrame` keeps what the reference keeps in the Frame between those calls -
`mvpMapPoints` (here: per keypoint the world position of its MapPoint and an id) and `mvbOutlier` - and issues the same calls; with
`resident=True` the frame's keypoints are uploaded once (`lld_frame_create`) and the two matchers move only their queries.
The line half of the sequence (`AddLinesFrom`, :924) and `ComputeStereoMatches` / `MatchLines` (src/Frame.cc:113,122) have their own entry
points (`lld_line_track_match`, `lld_compute_stereo_matches`, `lld_line_match_stereo`) and are not chained here."""
from __future__ import annotations

import numpy as np

from . import orb_search
from .host import Optimizer, PoseFrame


def pose_frame_from_matches(F: orb_search.Frame, cam, pose_qt, kp_world, kp_has) -> tuple[PoseFrame, np.ndarray]:
    """Optimizer::PoseOptimization's point edges from Frame::mvpMapPoints (src/Optimizer.cc:683-760): one edge per keypoint with a
    MapPoint, in keypoint order; stereo iff mvuRight[i] >= 0.  Returns the problem and the keypoint index of every edge."""
    idx = np.nonzero(kp_has)[0]
    uvr = np.stack([F.xy[idx, 0], F.xy[idx, 1], np.where(F.uright[idx] >= 0, F.uright[idx], -1.0)], 1).astype(np.float64)
    e = np.zeros((0, 3)); e4 = np.zeros((0, 4))
    f = PoseFrame(cam=cam, pose_qt=np.asarray(pose_qt, np.float64), pt_xw=kp_world[idx].astype(np.float64), pt_uvr=uvr,
                  pt_inv_sigma2=F.inv_sigma2[F.octave[idx]].astype(np.float64), ln_x0=e, ln_dir=e, ln_left=e4, ln_right=e4,
                  ln_octave=np.zeros((0, 2), np.int32))
    return f.normalise(), idx


def qt_to_tcw_f32(lib, qt) -> np.ndarray:
    """Converter::toCvMat(SE3Quat) (src/Converter.cc:49-70) through the ABI's own conversion."""
    import ctypes as C
    from .abi import c_double_p, c_float_p
    q = np.ascontiguousarray(qt, np.float64); T = np.zeros(16, np.float32)
    lib.fn("se3_to_tcw_f32")(q.ctypes.data_as(c_double_p), T.ctypes.data_as(c_float_p))
    return T.reshape(4, 4)


class TrackedFrame:
    def __init__(self, ctx, F: orb_search.Frame, cam, resident: bool = True):
        self.ctx, self.lib, self.F, self.cam = ctx, ctx.lib, F, cam
        self.res = orb_search.ResidentFrame(ctx.lib, ctx.handle, F) if resident else None
        self.kp_has = np.zeros(F.n, bool)                        # mvpMapPoints[i] != NULL
        self.kp_world = np.zeros((F.n, 3), np.float32)           # its GetWorldPos()
        self.kp_point = np.full(F.n, -1, np.int64)               # an id of the MapPoint (caller's numbering)
        self.kp_obs = np.zeros(F.n, bool)                        # its Observations() > 0: only such a MapPoint blocks its keypoint (src/ORBmatcher.cc:98-100)
        self.discarded = np.zeros(0, np.int64)                   # MapPoints dropped as outliers in this frame: mnLastFrameSeen = mnId (src/Tracking.cc:949)
        self.stages = {}                                         # what every stage returned, for the checker

    def close(self):
        if self.res is not None:
            self.res.close(); self.res = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def _view(self, pose_qt):
        return orb_search.frame_view(qt_to_tcw_f32(self.lib, pose_qt), self.cam, self.F)

    def _optimise(self, pose_qt, tag):
        prob, idx = pose_frame_from_matches(self.F, self.cam, pose_qt, self.kp_world, self.kp_has)
        out = Optimizer(self.ctx).PoseOptimization(prob, 0.5)
        self.stages[tag] = dict(problem=prob, edge_keypoint=idx, out=out)
        # the discard of src/Tracking.cc:940-958 / :1160-1178: an outlier edge's MapPoint leaves the frame
        bad = idx[out.pt_outlier != 0]
        self.discarded = np.concatenate([self.discarded, self.kp_point[bad]])
        self.kp_has[bad] = False; self.kp_point[bad] = -1
        return out.pose_qt

    def track_with_motion_model(self, pose_qt_guess, last: dict, last_ids, th=7.0, direction=0):
        """SearchByProjection(Current, Last) from the predicted pose, PoseOptimization on the matches, outlier discard."""
        view = self._view(pose_qt_guess)
        occ = (self.kp_has & self.kp_obs).astype(np.uint8)
        if self.res is not None:
            out, uvr = self.res.search_last_frame(view, last, occ, direction, th, True)
        else:
            out, uvr = orb_search.search_last_frame(self.lib, self.ctx.handle, self.F, view, last, occ, direction, th, True)
        self.stages["search_last_frame"] = dict(view=view, occupied=occ, out=out, uvr=uvr)
        ok = (out.match >= 0) & (out.removed == 0)
        # CurrentFrame.mvpMapPoints[bestIdx2] = pMP, later queries see it occupied; the orientation filter NULLs removed ones (:1452-1460)
        for q in np.nonzero(ok)[0]:
            k = int(out.match[q])
            if out.owner[k] == q:
                self.kp_has[k] = True; self.kp_world[k] = last["world_pos"][q]; self.kp_point[k] = last_ids[q]
                self.kp_obs[k] = True if last.get("has_obs") is None else bool(last["has_obs"][q])
        return self._optimise(pose_qt_guess, "pose_after_motion_model")

    def track_local_map(self, pose_qt, mp: dict, mp_ids, th=1.0, nnratio=0.8):
        """SearchLocalPoints (points the frame already holds are skipped, src/Tracking.cc:1620-1632) + PoseOptimization."""
        view = self._view(pose_qt)
        # mnLastFrameSeen == mCurrentFrame.mnId (:1640): the points the frame holds (:1629) AND those the discard after the first
        # PoseOptimization marked (:949) - the latter must not be projected and matched again in this frame
        held = np.isin(mp_ids, self.kp_point[self.kp_has]) | np.isin(mp_ids, self.discarded)
        skip = (np.asarray(mp["skip"]) != 0) | held
        mp2 = dict(mp, skip=skip.astype(np.uint8))
        occ = (self.kp_has & self.kp_obs).astype(np.uint8)
        if self.res is not None:
            out, fr = self.res.search_local_points(view, mp2, occ, th, nnratio)
        else:
            out, fr = orb_search.search_local_points(self.lib, self.ctx.handle, self.F, view, mp2, occ, th, nnratio)
        self.stages["search_local_points"] = dict(view=view, occupied=occ, points=mp2, out=out, frustum=fr)
        for q in np.nonzero(out.match >= 0)[0]:
            k = int(out.match[q])
            if out.owner[k] == q:
                self.kp_has[k] = True; self.kp_world[k] = mp["world_pos"][q]; self.kp_point[k] = mp_ids[q]
                self.kp_obs[k] = True if mp.get("has_obs") is None else bool(mp["has_obs"][q])
        return self._optimise(pose_qt, "pose_after_local_map")


# ------------------------------------------------------------------------------------------------ the device-resident chain (round 6)
import ctypes as C

from . import abi
from .abi import c_double_p, c_float_p, c_int32_p, c_uint8_p


class FrameLines(C.Structure):
    _fields_ = [("n_left", C.c_int32), ("left", c_float_p), ("left_octave", c_int32_p), ("n_right", C.c_int32), ("right", c_float_p),
                ("right_octave", c_int32_p), ("line_matches", c_int32_p), ("desc", c_float_p), ("dim", C.c_int32), ("reserved", C.c_int32),
                ("sx", C.c_double), ("sy", C.c_double)]


class MapLines(C.Structure):
    _fields_ = [("n", C.c_int32), ("x0", c_double_p), ("dir", c_double_p), ("x1", c_double_p), ("x2", c_double_p), ("skip", c_uint8_p),
                ("desc", c_float_p), ("id", c_int32_p)]


class TrackParams(C.Structure):
    _fields_ = [("cam", abi.Camera), ("pose", abi.PoseParams), ("th_motion", C.c_float), ("th_local", C.c_float), ("nnratio_local", C.c_float),
                ("viewing_cos_limit", C.c_float), ("direction", C.c_int32), ("check_orientation", C.c_int32), ("wide_retry", C.c_int32),
                ("monocular", C.c_int32), ("line_thr_reproj_base", C.c_double), ("line_md_thr", C.c_double), ("line_use_grid", C.c_int32),
                ("reserved", C.c_int32)]


class TrackResult(C.Structure):
    _fields_ = [("pose_qt", C.c_double * 7), ("chi2", C.c_double), ("n_inliers", C.c_int32), ("lm_iterations", C.c_int32), ("lm_trials", C.c_int32),
                ("n_edges", C.c_int32), ("n_search_first", C.c_int32), ("n_search", C.c_int32), ("used_wide", C.c_int32), ("n_points", C.c_int32),
                ("n_points_map", C.c_int32), ("n_lines_matched", C.c_int32), ("n_lines", C.c_int32), ("n_discarded", C.c_int32),
                ("n_point_edges", C.c_int32), ("n_in_view", C.c_int32),
                ("kp_point_id", c_int32_p), ("kp_outlier", c_uint8_p), ("ln_line_id", c_int32_p), ("ln_outlier", c_uint8_p), ("mp_in_view", c_uint8_p)]


class FrameHeld(C.Structure):
    """lld_frame_held (include/lld_amd.h): what a stage 1 that ran elsewhere left in the frame."""
    _fields_ = [("kp_point_id", c_int32_p), ("kp_world_pos", c_float_p), ("kp_has_obs", c_uint8_p), ("kp_outlier", c_uint8_p), ("n_seen", C.c_int32),
                ("seen_point_id", c_int32_p), ("ln_line_id", c_int32_p), ("ln_x0", c_double_p), ("ln_dir", c_double_p), ("ln_outlier", c_uint8_p),
                ("n_tracked", C.c_int32), ("tracked_line_id", c_int32_p)]


_COUNTERS = ("n_inliers", "lm_iterations", "lm_trials", "n_edges", "n_search_first", "n_search", "used_wide", "n_points", "n_points_map",
             "n_lines_matched", "n_lines", "n_discarded", "n_point_edges", "n_in_view")


def map_lines_struct(ml: dict | None):
    """lld_map_lines from a dict with X0, dir, X1, X2 [n,3], desc [n,dim], id [n] and optionally skip [n]; returns (struct, arrays kept alive)."""
    m = MapLines()
    if ml is None:
        return m, {}
    keep = dict(x0=np.ascontiguousarray(ml["X0"], np.float64).reshape(-1, 3), dir=np.ascontiguousarray(ml["dir"], np.float64).reshape(-1, 3),
                x1=np.ascontiguousarray(ml["X1"], np.float64).reshape(-1, 3), x2=np.ascontiguousarray(ml["X2"], np.float64).reshape(-1, 3),
                desc=np.ascontiguousarray(ml["desc"], np.float32), id=np.ascontiguousarray(ml["id"], np.int32),
                skip=None if ml.get("skip") is None else np.ascontiguousarray(ml["skip"], np.uint8))
    m.n = keep["x0"].shape[0]
    m.x0 = keep["x0"].ctypes.data_as(c_double_p); m.dir = keep["dir"].ctypes.data_as(c_double_p)
    m.x1 = keep["x1"].ctypes.data_as(c_double_p); m.x2 = keep["x2"].ctypes.data_as(c_double_p)
    m.skip = None if keep["skip"] is None else keep["skip"].ctypes.data_as(c_uint8_p)
    m.desc = keep["desc"].ctypes.data_as(c_float_p); m.id = keep["id"].ctypes.data_as(c_int32_p)
    return m, keep


class DeviceTrackedFrame:
    """lld_frame_track_*: the whole per-frame sequence on the device (include/lld_amd.h).  `lines`: dict with left_lines [n,4], left_octave,
    right_lines, right_octave, line_matches, desc (the frame's mvLinesLeft / mvLinesRight / line_matches / mDescriptorsLines) or None."""

    def __init__(self, ctx, F: orb_search.Frame, cam, lines: dict | None = None, gamma=0.5, **params):
        self.ctx, self.lib, self.F, self.cam = ctx, ctx.lib, F, cam
        self.res = orb_search.ResidentFrame(ctx.lib, ctx.handle, F)
        lib = self.lib
        lib.fn("frame_set_lines").argtypes = [C.c_void_p, C.POINTER(FrameLines)]; lib.fn("frame_set_lines").restype = C.c_int
        lib.fn("track_params_default").argtypes = [C.POINTER(TrackParams)]; lib.fn("track_params_default").restype = None
        lib.fn("frame_track_motion_model").argtypes = [C.c_void_p, C.POINTER(TrackParams), C.POINTER(orb_search.FrameView), c_double_p,
                                                       C.POINTER(orb_search.LastFramePoints), c_int32_p, C.POINTER(MapLines)]
        lib.fn("frame_track_motion_model").restype = C.c_int
        lib.fn("frame_track_local_map").argtypes = [C.c_void_p, C.POINTER(TrackParams), C.POINTER(orb_search.MapPoints), c_int32_p, C.POINTER(MapLines)]
        lib.fn("frame_track_local_map").restype = C.c_int
        lib.fn("frame_track_download").argtypes = [C.c_void_p, C.POINTER(TrackResult), C.POINTER(TrackResult)]; lib.fn("frame_track_download").restype = C.c_int
        self.n_lines = 0; self.n_local_points = 0
        if lines is not None:
            k = dict(left=np.ascontiguousarray(lines["left_lines"], np.float32).reshape(-1, 4), lo=np.ascontiguousarray(lines["left_octave"], np.int32),
                     right=np.ascontiguousarray(lines["right_lines"], np.float32).reshape(-1, 4), ro=np.ascontiguousarray(lines["right_octave"], np.int32),
                     lm=np.ascontiguousarray(lines["line_matches"], np.int32), desc=np.ascontiguousarray(lines["desc"], np.float32))
            L = FrameLines()
            L.n_left = k["left"].shape[0]; L.left = k["left"].ctypes.data_as(c_float_p); L.left_octave = k["lo"].ctypes.data_as(c_int32_p)
            L.n_right = k["right"].shape[0]; L.right = k["right"].ctypes.data_as(c_float_p); L.right_octave = k["ro"].ctypes.data_as(c_int32_p)
            L.line_matches = k["lm"].ctypes.data_as(c_int32_p); L.desc = k["desc"].ctypes.data_as(c_float_p); L.dim = k["desc"].shape[1]
            L.sx = 1.0 / float(F.max_x); L.sy = 1.0 / float(F.max_y)
            self._check(lib.fn("frame_set_lines")(self.res.handle, C.byref(L)), "lld_frame_set_lines")
            self.n_lines = int(L.n_left)
        self.params = TrackParams()
        lib.fn("track_params_default")(C.byref(self.params))
        self.params.cam = abi.Camera(*[float(np.float32(c)) for c in cam])
        self.params.pose.gamma = gamma
        for k_, v in params.items():
            setattr(self.params, k_, v)

    def _check(self, st, what):
        if st != abi.LLD_OK:
            raise RuntimeError(f"{what} failed: {self.lib.fn('status_string')(st).decode()}")

    def close(self):
        if self.res is not None:
            self.res.close(); self.res = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def track_with_motion_model(self, Tcw_f32, last: dict, last_ids, last_lines: dict | None = None):
        """Queues stage 1 from the predicted float pose matrix (mVelocity * mLastFrame.mTcw); returns nothing (see download())."""
        from .host import se3_from_tcw_f32
        T = np.ascontiguousarray(Tcw_f32, np.float32).reshape(4, 4)
        view = orb_search.frame_view(T, self.cam, self.F)
        qt = np.ascontiguousarray(se3_from_tcw_f32(self.lib, T), np.float64)
        m, keep = orb_search.last_frame_struct(last)
        ids = np.ascontiguousarray(last_ids, np.int32)
        ml, keep2 = map_lines_struct(last_lines)
        self._check(self.lib.fn("frame_track_motion_model")(self.res.handle, C.byref(self.params), C.byref(view), qt.ctypes.data_as(c_double_p), C.byref(m),
                                                             ids.ctypes.data_as(c_int32_p), C.byref(ml) if last_lines is not None else None), "lld_frame_track_motion_model")
        return view, qt

    def set_state(self, Tcw_f32, kp_point_id, kp_world_pos, kp_has_obs=None, kp_outlier=None, seen_point_id=(), ln_line_id=None, ln_x0=None, ln_dir=None,
                  ln_outlier=None, tracked_line_id=()):
        """lld_frame_track_set_state: stage 1 ran elsewhere (TrackReferenceKeyFrame / Relocalization); the frame's float pose and what it holds."""
        from .host import se3_from_tcw_f32
        fn = self.lib.fn("frame_track_set_state")
        fn.argtypes = [C.c_void_p, C.POINTER(TrackParams), C.POINTER(orb_search.FrameView), c_double_p, C.POINTER(FrameHeld)]; fn.restype = C.c_int
        T = np.ascontiguousarray(Tcw_f32, np.float32).reshape(4, 4)
        view = orb_search.frame_view(T, self.cam, self.F)
        qt = np.ascontiguousarray(se3_from_tcw_f32(self.lib, T), np.float64)
        H = FrameHeld(); keep = []

        def arr(a, dt, ptr):
            if a is None: return None
            a = np.ascontiguousarray(a, dt); keep.append(a)
            return a.ctypes.data_as(ptr)
        H.kp_point_id = arr(kp_point_id, np.int32, c_int32_p); H.kp_world_pos = arr(kp_world_pos, np.float32, c_float_p)
        H.kp_has_obs = arr(kp_has_obs, np.uint8, c_uint8_p); H.kp_outlier = arr(kp_outlier, np.uint8, c_uint8_p)
        H.n_seen = len(seen_point_id); H.seen_point_id = arr(seen_point_id if len(seen_point_id) else None, np.int32, c_int32_p)
        H.ln_line_id = arr(ln_line_id, np.int32, c_int32_p); H.ln_x0 = arr(ln_x0, np.float64, c_double_p); H.ln_dir = arr(ln_dir, np.float64, c_double_p)
        H.ln_outlier = arr(ln_outlier, np.uint8, c_uint8_p)
        H.n_tracked = len(tracked_line_id); H.tracked_line_id = arr(tracked_line_id if len(tracked_line_id) else None, np.int32, c_int32_p)
        self._check(fn(self.res.handle, C.byref(self.params), C.byref(view), qt.ctypes.data_as(c_double_p), C.byref(H)), "lld_frame_track_set_state")
        return view, qt

    def track_local_map(self, mp: dict, mp_ids, local_lines: dict | None = None):
        m, keep = orb_search.map_points_struct(mp)
        ids = np.ascontiguousarray(mp_ids, np.int32)
        self.n_local_points = int(m.n)
        ml, keep2 = map_lines_struct(local_lines)
        self._check(self.lib.fn("frame_track_local_map")(self.res.handle, C.byref(self.params), C.byref(m), ids.ctypes.data_as(c_int32_p),
                                                          C.byref(ml) if local_lines is not None else None), "lld_frame_track_local_map")

    def download(self, stage2=True):
        """One copy, one synchronisation: the records of stage 1 and (if queued) stage 2 as dicts."""
        nt, nl = self.F.n, self.n_lines
        outs = []
        for st_ in range(2):
            r = TrackResult()
            a = dict(kp_point_id=np.empty(nt, np.int32), kp_outlier=np.empty(nt, np.uint8), ln_line_id=np.empty(nl, np.int32), ln_outlier=np.empty(nl, np.uint8))
            r.kp_point_id = a["kp_point_id"].ctypes.data_as(c_int32_p); r.kp_outlier = a["kp_outlier"].ctypes.data_as(c_uint8_p)
            r.ln_line_id = a["ln_line_id"].ctypes.data_as(c_int32_p); r.ln_outlier = a["ln_outlier"].ctypes.data_as(c_uint8_p)
            if st_ == 1 and stage2:
                a["mp_in_view"] = np.zeros(self.n_local_points, np.uint8); r.mp_in_view = a["mp_in_view"].ctypes.data_as(c_uint8_p)
            outs.append((r, a))
        self._check(self.lib.fn("frame_track_download")(self.res.handle, C.byref(outs[0][0]), C.byref(outs[1][0]) if stage2 else None), "lld_frame_track_download")
        res = []
        for r, a in outs[:2 if stage2 else 1]:
            d = dict(a, pose_qt=np.array(list(r.pose_qt)), chi2=float(r.chi2))
            for c in _COUNTERS: d[c] = int(getattr(r, c))
            res.append(d)
        return res


# ------------------------------------------------------------------------------------------------ flat files of examples/harness.cpp `track`
def write_harness_scene(path, sc: dict, repeats=1, download_between=False, gamma=0.5, thr_base=2.0, md_thr=0.9):
    """A make_tracking_scene dict as the flat binary `examples/harness track` reads (layout: run_track in examples/harness.cpp)."""
    F = sc["frame"]; lines = sc.get("lines")
    T = np.ascontiguousarray(sc["Tcw_guess"], np.float32).reshape(4, 4)
    view = orb_search.frame_view(T, sc["cam"], F)
    last, mp = sc["last"], sc["map_points"]
    n_last, n_mp = len(sc["last_ids"]), len(sc["map_ids"])
    nl = 0 if lines is None else np.asarray(lines["left_lines"]).reshape(-1, 4).shape[0]
    nr = 0 if lines is None else np.asarray(lines["right_lines"]).reshape(-1, 4).shape[0]
    dim = 1 if lines is None else np.asarray(lines["desc"]).shape[1]
    ll, ml = (sc.get("last_lines"), sc.get("local_lines")) if lines is not None else (None, None)
    n_ll = 0 if ll is None else len(ll["id"]); n_ml = 0 if ml is None else len(ml["id"])
    f32 = lambda a: np.ascontiguousarray(a, np.float32); i32 = lambda a: np.ascontiguousarray(a, np.int32); u8 = lambda a: np.ascontiguousarray(a, np.uint8)
    with open(path, "wb") as f:
        i32([F.n, F.scale.shape[0], n_last, n_mp, nl, nr, dim, n_ll, n_ml, repeats, int(download_between), 0, 0, 0, 0, 0]).tofile(f)
        f32([F.min_x, F.min_y, F.max_x, F.max_y, F.width_inv, F.height_inv]).tofile(f)
        f32(F.scale).tofile(f); f32(F.inv_sigma2).tofile(f)
        np.array([float(np.float32(c)) for c in sc["cam"]] + [gamma, thr_base, md_thr], np.float64).tofile(f)
        np.ascontiguousarray(F.desc, np.uint32).tofile(f); f32(F.xy).tofile(f); i32(F.octave).tofile(f); f32(F.uright).tofile(f); f32(F.angle).tofile(f)
        f.write(bytes(view)); f32(T).tofile(f)
        obs = last.get("has_obs") if last.get("has_obs") is not None else np.ones(n_last, np.uint8)
        f32(last["world_pos"]).tofile(f); u8(last["valid"]).tofile(f); i32(last["octave"]).tofile(f); f32(last["angle"]).tofile(f)
        np.ascontiguousarray(last["desc"], np.uint32).tofile(f); u8(obs).tofile(f); i32(sc["last_ids"]).tofile(f)
        mobs = mp.get("has_obs") if mp.get("has_obs") is not None else np.ones(n_mp, np.uint8)
        f32(mp["world_pos"]).tofile(f); f32(mp["normal"]).tofile(f); f32(mp["max_distance"]).tofile(f); f32(mp["min_distance"]).tofile(f)
        np.ascontiguousarray(mp["desc"], np.uint32).tofile(f); u8(mobs).tofile(f); u8(mp["skip"]).tofile(f); i32(sc["map_ids"]).tofile(f)
        if lines is not None:
            f32(lines["left_lines"]).tofile(f); i32(lines["left_octave"]).tofile(f); f32(lines["right_lines"]).tofile(f); i32(lines["right_octave"]).tofile(f)
            i32(lines["line_matches"]).tofile(f); f32(lines["desc"]).tofile(f)
        for L in (ll, ml):
            if L is None or len(L["id"]) == 0: continue
            for k in ("X0", "dir", "X1", "X2"): np.ascontiguousarray(L[k], np.float64).tofile(f)
            u8(L["skip"] if L.get("skip") is not None else np.zeros(len(L["id"]), np.uint8)).tofile(f); f32(L["desc"]).tofile(f); i32(L["id"]).tofile(f)
    return nl


def read_harness_result(path, nt, nl, repeats):
    """(record of stage 1, record of stage 2, dict of per-repeat host milliseconds) written by `examples/harness track`."""
    recs = []
    with open(path, "rb") as f:
        for _ in range(2):
            d = dict(pose_qt=np.fromfile(f, np.float64, 7), chi2=float(np.fromfile(f, np.float64, 1)[0]))
            c = np.fromfile(f, np.int32, 12)
            for k, v in zip(_COUNTERS[:12], c): d[k] = int(v)
            d["kp_point_id"] = np.fromfile(f, np.int32, nt); d["kp_outlier"] = np.fromfile(f, np.uint8, nt)
            d["ln_line_id"] = np.fromfile(f, np.int32, nl); d["ln_outlier"] = np.fromfile(f, np.uint8, nl)
            recs.append(d)
        ms = dict(total=np.fromfile(f, np.float64, repeats), queue_motion_model=np.fromfile(f, np.float64, repeats), queue_local_map=np.fromfile(f, np.float64, repeats))
    return recs[0], recs[1], ms
