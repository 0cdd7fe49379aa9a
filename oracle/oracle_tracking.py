"""TEST INFRASTRUCTURE ONLY: the Tracking thread's per-frame sequence on ONE Frame, restated on the CPU from the oracle's routines.

TrackWithMotionModel (src/Tracking.cc:885-994) and TrackLocalMap (:1126-1220) for a stereo frame, with the Frame's members as numpy
arrays - mvpMapPoints (id + world position per keypoint), mvbOutlier, mvpMapLines, mvbOutlierLines, mTcw - and every library call of
the reference replaced by the oracle's restatement of it (oracle_orbsearch: the projection loops and the two SearchByProjection;
oracle_py: AddLinesFrom, PoseOptimization, the Converter).  Nothing of the device chain's intermediate state enters: the checker runs
the whole sequence on its own and only its per-stage records are compared with lld_frame_track_download's (tests/test_gpu_track_chain.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

import oracle_orbsearch as OS
import oracle_py as O
from lld_slam_amd import host, orb_search


def frame_lines_camera(view):
    """T_curr of AddLinesFrom as this build defines it (include/lld_amd.h, deviations): [Rwc | Ow] of the frame widened to double."""
    R = np.array(view.Rcw, np.float32).reshape(3, 3)
    T = np.eye(4)
    T[:3, :3] = R.T.astype(np.float64); T[:3, 3] = np.array(view.Ow, np.float32).astype(np.float64)
    return T


class Frame:
    """The members of Frame the sequence reads and writes."""

    def __init__(self, F: orb_search.Frame, cam, lines: dict | None):
        self.F, self.cam, self.lines = F, cam, lines
        nt = F.n; nl = 0 if lines is None else np.asarray(lines["left_lines"]).reshape(-1, 4).shape[0]
        self.nt, self.nl = nt, nl
        self.kp_has = np.zeros(nt, bool); self.kp_world = np.zeros((nt, 3), np.float32); self.kp_id = np.full(nt, -1, np.int64)
        self.kp_obs = np.zeros(nt, np.uint8); self.kp_out = np.zeros(nt, np.uint8)
        self.ln_has = np.zeros(nl, bool); self.ln_x0 = np.zeros((nl, 3)); self.ln_dir = np.zeros((nl, 3)); self.ln_id = np.full(nl, -1, np.int64)
        self.ln_out = np.zeros(nl, np.uint8)
        self.seen_points: set[int] = set()          # MapPoints with mnLastFrameSeen == this frame's id
        self.tracked_lines: set[int] = set()        # MapLines with tracked_last_id == this frame's id
        self.view = None; self.pose_qt = None; self.problems = []

    def set_pose_matrix(self, Tcw_f32):
        """Frame::SetPose + UpdatePoseMatrices."""
        T = np.ascontiguousarray(Tcw_f32, np.float32).reshape(4, 4)
        self.view = orb_search.frame_view(T, self.cam, self.F)
        self.pose_qt = host.se3_from_tcw_f32(O.lib(), T)                  # Converter::toSE3Quat(pFrame->mTcw)

    # ---------------------------------------------------------------- AddLinesFrom (src/Tracking.cc:996-1124)
    def add_lines_from(self, ml: dict | None, thr_base, md_thr, use_grid):
        if ml is None or self.nl == 0 or len(ml["id"]) == 0:
            return
        fx, fy, cx, cy, bf = [float(np.float32(c)) for c in self.cam]
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
        b = float(np.float32(np.float32(bf) / np.float32(fx)))
        ids = np.asarray(ml["id"])
        skip = (np.zeros(len(ids), np.uint8) if ml.get("skip") is None else np.asarray(ml["skip"], np.uint8).copy())
        skip |= np.isin(ids, list(self.tracked_lines)).astype(np.uint8)   # (unsigned int) pML->tracked_last_id == mCurrentFrame.mnId
        last = dict(X0=ml["X0"], dir=ml["dir"], X1=ml["X1"], X2=ml["X2"], desc=ml["desc"], skip=skip)
        fr = dict(left_lines=self.lines["left_lines"], left_octave=self.lines["left_octave"], right_lines=self.lines["right_lines"],
                  line_matches=self.lines["line_matches"], occupied=self.ln_has.astype(np.uint8), desc=self.lines["desc"])
        res = O.line_track_match(K, frame_lines_camera(self.view), b, thr_base, md_thr, 1.0 / float(self.F.max_x), 1.0 / float(self.F.max_y), last, fr,
                                 monocular=False, use_grid=use_grid)
        matches = np.asarray(res[0] if isinstance(res, tuple) else res)
        for i in np.nonzero(matches >= 0)[0]:
            mi = int(matches[i])
            self.ln_has[mi] = True; self.ln_x0[mi] = np.asarray(ml["X0"], np.float64)[i]; self.ln_dir[mi] = np.asarray(ml["dir"], np.float64)[i]
            self.ln_id[mi] = int(ids[i]); self.tracked_lines.add(int(ids[i]))

    # ---------------------------------------------------------------- Optimizer::PoseOptimization (src/Optimizer.cc:653-932)
    def pose_optimization(self, gamma):
        F = self.F
        idx = np.nonzero(self.kp_has)[0]
        self.kp_out[idx] = 0                                              # pFrame->mvbOutlier[i] = false for every point with a MapPoint
        li = np.nonzero(self.ln_has)[0]
        uvr = np.stack([F.xy[idx, 0], F.xy[idx, 1], np.where(F.uright[idx] >= 0, F.uright[idx], -1.0)], 1).astype(np.float64).reshape(-1, 3)
        L = self.lines
        if len(li):
            left = np.asarray(L["left_lines"], np.float32).reshape(-1, 4)[li].astype(np.float64)
            lm = np.asarray(L["line_matches"])[li]
            rl = np.asarray(L["right_lines"], np.float32).reshape(-1, 4); ro = np.asarray(L["right_octave"])
            right = np.where((lm >= 0)[:, None], rl[np.maximum(lm, 0)].astype(np.float64), -1.0)
            octs = np.stack([np.asarray(L["left_octave"])[li], np.where(lm >= 0, ro[np.maximum(lm, 0)], 0)], 1).astype(np.int32)
        else:
            left = np.zeros((0, 4)); right = np.zeros((0, 4)); octs = np.zeros((0, 2), np.int32)
        prob = host.PoseFrame(cam=self.cam, pose_qt=np.asarray(self.pose_qt, np.float64), pt_xw=self.kp_world[idx].astype(np.float64).reshape(-1, 3), pt_uvr=uvr,
                              pt_inv_sigma2=F.inv_sigma2[F.octave[idx]].astype(np.float64), ln_x0=self.ln_x0[li].reshape(-1, 3), ln_dir=self.ln_dir[li].reshape(-1, 3),
                              ln_left=left, ln_right=right, ln_octave=octs, ln_frame_index=li.astype(np.int32)).normalise()
        out = O.pose_opt(prob, gamma)
        self.problems.append(prob)                                        # (for tools/experiments: the edges as PoseOptimization saw them)
        n_le = int(len(li) + np.count_nonzero(np.asarray(L["line_matches"])[li] >= 0)) if len(li) else 0
        self.kp_out[idx] = out.pt_outlier
        if len(idx) >= 3 and len(idx) + n_le >= 10:                       # the line classification is reached (Optimizer.cc:809, :878)
            self.ln_out[li] = out.ln_outlier
        # pFrame->SetPose(Converter::toCvMat(SE3quat_recov)): only when the optimisation ran
        if len(idx) >= 3:
            self.set_pose_from_qt(out.pose_qt)
        return out, len(idx) + n_le

    def set_pose_from_qt(self, qt):
        T = host.se3_to_tcw_f32(O.lib(), np.asarray(qt, np.float64))
        self.view = orb_search.frame_view(T, self.cam, self.F)
        self.pose_qt = host.se3_from_tcw_f32(O.lib(), T)                  # the next PoseOptimization starts from toSE3Quat(mTcw), mTcw being floats (Optimizer.cc:823)

    def record(self, out, n_edges, extra):
        return dict(kp_point_id=np.where(self.kp_has, self.kp_id, -1).astype(np.int32), kp_outlier=np.where(self.kp_has, self.kp_out, 0).astype(np.uint8),
                    ln_line_id=np.where(self.ln_has, self.ln_id, -1).astype(np.int32), ln_outlier=np.where(self.ln_has, self.ln_out, 0).astype(np.uint8),
                    pose_qt=np.asarray(out.pose_qt, np.float64).copy(), chi2=float(out.chi2), n_inliers=int(out.n_inliers), lm_iterations=int(out.lm_iterations),
                    lm_trials=int(out.lm_trials), n_edges=int(n_edges), n_lines_matched=int(self.ln_has.sum()), **extra)


def track_frame(sc: dict, gamma=0.5, th_motion=7.0, th_local=1.0, nnratio=0.8, wide_retry=True, thr_base=2.0, md_thr=0.9, use_grid=True, direction=0):
    """Both stages on the scene `sc` (lld_slam_amd.synth.make_tracking_scene): returns (record of stage 1, record of stage 2) with the fields of
    lld_track_result."""
    F = sc["frame"]
    fr = Frame(F, sc["cam"], sc.get("lines"))
    # ================= TrackWithMotionModel
    fr.set_pose_matrix(sc["Tcw_guess"])
    last = sc["last"]; last_ids = np.asarray(sc["last_ids"])
    valid, uv, ur = OS.project_last_frame(fr.view, last)
    has_obs = last.get("has_obs") if last.get("has_obs") is not None else np.ones(len(last_ids), np.uint8)
    occ = np.zeros(F.n, np.uint8)

    def search(th):
        return OS.search_by_projection_frame(F, last["desc"], valid, uv, ur, last["octave"], last["angle"], has_obs, occ, direction, th, True)
    n1, slot = search(th_motion)
    n_used, used_wide = n1, 0
    if wide_retry and n1 < 20:
        n_used, slot = search(2.0 * th_motion); used_wide = 1
    for k in np.nonzero(slot >= 0)[0]:
        q = int(slot[k])
        fr.kp_has[k] = True; fr.kp_world[k] = np.asarray(last["world_pos"], np.float32)[q]; fr.kp_id[k] = int(last_ids[q]); fr.kp_obs[k] = int(has_obs[q])
    fr.add_lines_from(sc.get("last_lines"), thr_base, md_thr, use_grid)
    out, n_edges = fr.pose_optimization(gamma)
    rec1 = fr.record(out, n_edges, dict(n_search_first=int(n1), n_search=int(n_used), used_wide=used_wide, n_point_edges=int(fr.problems[-1].n_points), n_in_view=0))
    # discard (:940-975)
    bad = fr.kp_has & (fr.kp_out != 0)
    fr.seen_points.update(int(i) for i in fr.kp_id[bad])
    fr.kp_has[bad] = False; fr.kp_id[bad] = -1; fr.kp_out[bad] = 0
    rec1.update(n_points=int(fr.kp_has.sum()), n_points_map=int((fr.kp_has & (fr.kp_obs != 0)).sum()), n_discarded=int(bad.sum()))
    lbad = fr.ln_has & (fr.ln_out != 0)
    fr.ln_has[lbad] = False; fr.ln_id[lbad] = -1                          # (mvbOutlierLines keeps its value)
    rec1["n_lines"] = int(fr.ln_has.sum())
    # ================= TrackLocalMap
    mp = sc["map_points"]; mp_ids = np.asarray(sc["map_ids"])
    fr.seen_points.update(int(i) for i in fr.kp_id[fr.kp_has])           # SearchLocalPoints: pMP->mnLastFrameSeen = mnId for what the frame holds
    skip = (np.asarray(mp["skip"]) != 0) | np.isin(mp_ids, list(fr.seen_points))
    mp2 = dict(mp, skip=skip.astype(np.uint8))
    k, inv, uvr, lvl, vc = OS.is_in_frustum(fr.view, mp2)
    occ = (fr.kp_has & (fr.kp_obs != 0)).astype(np.uint8)                # if(F.mvpMapPoints[idx]) if(F.mvpMapPoints[idx]->Observations()>0) continue;  (ORBmatcher.cc:98-100)
    mp_obs = mp2.get("has_obs") if mp2.get("has_obs") is not None else np.ones(len(mp_ids), np.uint8)
    n2, slot = OS.search_by_projection_map(F, mp2["desc"], inv, uvr[:, :2], uvr[:, 2], lvl, vc, mp_obs, occ, th_local, nnratio)
    for kk in np.nonzero((slot >= 0) & (slot < (1 << 20)))[0]:          # (a keypoint whose MapPoint has no observations may be taken over: the last writer stays)
        q = int(slot[kk])
        fr.kp_has[kk] = True; fr.kp_world[kk] = np.asarray(mp["world_pos"], np.float32)[q]; fr.kp_id[kk] = int(mp_ids[q]); fr.kp_obs[kk] = int(mp_obs[q])
    fr.add_lines_from(sc.get("local_lines"), thr_base, md_thr, use_grid)
    out, n_edges = fr.pose_optimization(gamma)
    rec2 = fr.record(out, n_edges, dict(n_search_first=int(n2), n_search=int(n2), used_wide=0, n_point_edges=int(fr.problems[-1].n_points), n_in_view=int(np.count_nonzero(inv)),
                                        mp_in_view=np.asarray(inv, np.uint8).copy()))
    bad = fr.kp_has & (fr.kp_out != 0)
    fr.kp_has[bad] = False; fr.kp_id[bad] = -1                            # STEREO: mvpMapPoints[i] = NULL, mvbOutlier stays (:1170-1171)
    rec2.update(n_points=int(fr.kp_has.sum()), n_points_map=int((fr.kp_has & (fr.kp_obs != 0)).sum()), n_discarded=int(bad.sum()))
    lbad = fr.ln_has & (fr.ln_out != 0)
    fr.ln_has[lbad] = False; fr.ln_id[lbad] = -1
    rec2["n_lines"] = int(fr.ln_has.sum())
    track_frame.last_problems = fr.problems
    return rec1, rec2
