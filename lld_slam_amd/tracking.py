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
        self.kp_has[bad] = False; self.kp_point[bad] = -1
        return out.pose_qt

    def track_with_motion_model(self, pose_qt_guess, last: dict, last_ids, th=7.0, direction=0):
        """SearchByProjection(Current, Last) from the predicted pose, PoseOptimization on the matches, outlier discard."""
        view = self._view(pose_qt_guess)
        occ = self.kp_has.astype(np.uint8)
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
        return self._optimise(pose_qt_guess, "pose_after_motion_model")

    def track_local_map(self, pose_qt, mp: dict, mp_ids, th=1.0, nnratio=0.8):
        """SearchLocalPoints (points the frame already holds are skipped, src/Tracking.cc:1620-1632) + PoseOptimization."""
        view = self._view(pose_qt)
        held = np.isin(mp_ids, self.kp_point[self.kp_has])
        skip = (np.asarray(mp["skip"]) != 0) | held
        mp2 = dict(mp, skip=skip.astype(np.uint8))
        occ = self.kp_has.astype(np.uint8)
        if self.res is not None:
            out, fr = self.res.search_local_points(view, mp2, occ, th, nnratio)
        else:
            out, fr = orb_search.search_local_points(self.lib, self.ctx.handle, self.F, view, mp2, occ, th, nnratio)
        self.stages["search_local_points"] = dict(view=view, occupied=occ, points=mp2, out=out, frustum=fr)
        for q in np.nonzero(out.match >= 0)[0]:
            k = int(out.match[q])
            if out.owner[k] == q:
                self.kp_has[k] = True; self.kp_world[k] = mp["world_pos"][q]; self.kp_point[k] = mp_ids[q]
        return self._optimise(pose_qt, "pose_after_local_map")
