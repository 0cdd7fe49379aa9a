"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/lldo_orbsearch.cpp: literal sequential restatements of the reference's guided ORB searches
(src/ORBmatcher.cc, src/Frame.cc).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under lld_slam_amd/ does.  PARITY UNPINNED (see lldo_orbsearch.cpp).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle_py
from lld_slam_amd.abi import c_float_p, c_int32_p, c_uint32_p, c_uint8_p
from lld_slam_amd.orb_search import Frame


class OFrame(C.Structure):
    _fields_ = [("n", C.c_int32), ("desc", c_uint32_p), ("xy", c_float_p), ("octave", c_int32_p), ("uright", c_float_p),
                ("angle", c_float_p), ("min_x", C.c_float), ("min_y", C.c_float), ("width_inv", C.c_float), ("height_inv", C.c_float),
                ("n_levels", C.c_int32), ("scale", c_float_p), ("sigma2", c_float_p), ("inv_sigma2", c_float_p)]


def oframe(F: Frame) -> OFrame:
    F.normalise()
    o = OFrame()
    o.n = F.n; o.desc = F.desc.ctypes.data_as(c_uint32_p); o.xy = F.xy.ctypes.data_as(c_float_p)
    o.octave = F.octave.ctypes.data_as(c_int32_p); o.uright = F.uright.ctypes.data_as(c_float_p); o.angle = F.angle.ctypes.data_as(c_float_p)
    o.min_x = float(np.float32(F.min_x)); o.min_y = float(np.float32(F.min_y)); o.width_inv = float(F.width_inv); o.height_inv = float(F.height_inv)
    o.n_levels = F.scale.shape[0]; o.scale = F.scale.ctypes.data_as(c_float_p); o.sigma2 = F.sigma2.ctypes.data_as(c_float_p)
    o.inv_sigma2 = F.inv_sigma2.ctypes.data_as(c_float_p)
    o._keep = F
    return o


def _dll():
    d = oracle_py.lib().dll
    if not getattr(d, "_orbsearch_bound", False):
        fp = C.POINTER(OFrame)
        d.lldo_features_in_area.argtypes = [fp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, c_int32_p]; d.lldo_features_in_area.restype = C.c_int
        d.lldo_three_maxima.argtypes = [c_int32_p, c_int32_p]; d.lldo_three_maxima.restype = None
        d.lldo_search_by_projection_map.argtypes = [fp, C.c_int, c_uint32_p, c_uint8_p, c_float_p, c_float_p, c_int32_p, c_float_p, c_uint8_p,
                                                    C.c_float, C.c_float, c_int32_p, c_uint8_p]
        d.lldo_search_by_projection_frame.argtypes = [fp, C.c_int, c_uint32_p, c_uint8_p, c_float_p, c_float_p, c_int32_p, c_float_p, c_uint8_p,
                                                      C.c_int, C.c_float, C.c_int, c_int32_p, c_uint8_p]
        d.lldo_search_by_projection_reloc.argtypes = [fp, C.c_int, c_uint32_p, c_uint8_p, c_float_p, c_int32_p, c_float_p, C.c_float, C.c_int,
                                                      C.c_int, c_int32_p]
        d.lldo_search_by_projection_kf.argtypes = [fp, C.c_int, c_uint32_p, c_uint8_p, c_float_p, c_int32_p, C.c_int, c_int32_p]
        d.lldo_fuse_search.argtypes = [fp, C.c_int, c_uint32_p, c_uint8_p, c_float_p, c_float_p, c_int32_p, C.c_float, c_int32_p]
        d.lldo_search_for_initialization.argtypes = [fp, fp, c_float_p, C.c_int, C.c_float, C.c_int, c_int32_p]
        d.lldo_search_for_initialization.restype = C.c_int
        d.lldo_search_sim3_direction.argtypes = [fp, C.c_int, c_uint32_p, c_uint8_p, c_float_p, c_int32_p, C.c_float, c_int32_p]
        d.lldo_search_sim3_direction.restype = None
        d.lldo_search_by_bow_frame.argtypes = [fp, fp, C.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_uint8_p, C.c_float, C.c_int, c_int32_p]
        d.lldo_search_by_bow_kf.argtypes = [fp, fp, C.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_uint8_p, c_uint8_p, C.c_float, C.c_int,
                                            c_int32_p]
        d.lldo_search_for_triangulation.argtypes = [fp, fp, C.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_uint8_p, c_uint8_p, c_float_p,
                                                    C.c_float, C.c_float, C.c_int, C.c_int, c_int32_p]
        d.lldo_epipolar_line.argtypes = [c_float_p, C.c_float, C.c_float, c_float_p]; d.lldo_epipolar_line.restype = None
        d.lldo_stereo_search.argtypes = [fp, fp, C.c_int, C.c_float, C.c_float, c_int32_p, c_int32_p]; d.lldo_stereo_search.restype = None
        for n in ("lldo_search_by_projection_map", "lldo_search_by_projection_frame", "lldo_search_by_projection_reloc",
                  "lldo_search_by_projection_kf", "lldo_fuse_search", "lldo_search_by_bow_frame", "lldo_search_by_bow_kf",
                  "lldo_search_for_triangulation"):
            getattr(d, n).restype = C.c_int
        d._orbsearch_bound = True
    return d


def _u32(a): return np.ascontiguousarray(a, np.uint32)
def _f32(a): return np.ascontiguousarray(a, np.float32)
def _i32(a): return np.ascontiguousarray(a, np.int32)
def _u8(a): return np.ascontiguousarray(a, np.uint8)
def _p(a, t): return a.ctypes.data_as(t)


def features_in_area(F, x, y, r, min_level=-1, max_level=-1):
    out = np.empty(F.n, np.int32)
    n = _dll().lldo_features_in_area(C.byref(oframe(F)), x, y, r, min_level, max_level, _p(out, c_int32_p))
    return out[:n].copy()


def three_maxima(counts):
    counts = _i32(counts); ind = np.empty(3, np.int32)
    _dll().lldo_three_maxima(_p(counts, c_int32_p), _p(ind, c_int32_p))
    return ind


def slots_from_occupied(occ, token=1 << 20):
    """Frame slot vector for the oracle: >=0 where a MapPoint with observations sits on entry."""
    return np.where(_u8(occ) != 0, token, -1).astype(np.int32)


def search_by_projection_map(F, mp_desc, in_view, proj, proj_xr, pred_level, view_cos, mp_obs, f_occupied, th=1.0, nnratio=0.6):
    a = [_u32(mp_desc), _u8(in_view), _f32(proj), _f32(proj_xr), _i32(pred_level), _f32(view_cos), _u8(mp_obs)]
    slot = slots_from_occupied(f_occupied); slot_obs = _u8(f_occupied).copy()
    n = _dll().lldo_search_by_projection_map(C.byref(oframe(F)), a[0].shape[0], _p(a[0], c_uint32_p), _p(a[1], c_uint8_p), _p(a[2], c_float_p),
                                             _p(a[3], c_float_p), _p(a[4], c_int32_p), _p(a[5], c_float_p), _p(a[6], c_uint8_p), th, nnratio,
                                             _p(slot, c_int32_p), _p(slot_obs, c_uint8_p))
    return n, slot


def search_by_projection_frame(Cur, last_desc, valid, uv, ur, last_octave, last_angle, mp_obs, cur_occupied, direction=0, th=7.0,
                               check_orientation=True):
    a = [_u32(last_desc), _u8(valid), _f32(uv), _f32(ur), _i32(last_octave), _f32(last_angle), _u8(mp_obs)]
    slot = slots_from_occupied(cur_occupied); slot_obs = _u8(cur_occupied).copy()
    n = _dll().lldo_search_by_projection_frame(C.byref(oframe(Cur)), a[0].shape[0], _p(a[0], c_uint32_p), _p(a[1], c_uint8_p), _p(a[2], c_float_p),
                                               _p(a[3], c_float_p), _p(a[4], c_int32_p), _p(a[5], c_float_p), _p(a[6], c_uint8_p), direction, th,
                                               int(check_orientation), _p(slot, c_int32_p), _p(slot_obs, c_uint8_p))
    return n, slot


def search_by_projection_reloc(Cur, desc, valid, uv, pred_level, kf_angle, cur_occupied, th, ORBdist, check_orientation=True):
    a = [_u32(desc), _u8(valid), _f32(uv), _i32(pred_level), _f32(kf_angle)]
    slot = slots_from_occupied(cur_occupied)
    n = _dll().lldo_search_by_projection_reloc(C.byref(oframe(Cur)), a[0].shape[0], _p(a[0], c_uint32_p), _p(a[1], c_uint8_p), _p(a[2], c_float_p),
                                               _p(a[3], c_int32_p), _p(a[4], c_float_p), th, ORBdist, int(check_orientation), _p(slot, c_int32_p))
    return n, slot


def search_by_projection_kf(KF, desc, valid, uv, pred_level, matched, th):
    a = [_u32(desc), _u8(valid), _f32(uv), _i32(pred_level)]
    slot = slots_from_occupied(matched)
    n = _dll().lldo_search_by_projection_kf(C.byref(oframe(KF)), a[0].shape[0], _p(a[0], c_uint32_p), _p(a[1], c_uint8_p), _p(a[2], c_float_p),
                                            _p(a[3], c_int32_p), int(th), _p(slot, c_int32_p))
    return n, slot


def search_for_initialization(F1, F2, prev_matched, window_size=10, nnratio=0.9, check_orientation=True):
    """ORBmatcher::SearchForInitialization: (nmatches, vnMatches12, updated vbPrevMatched)."""
    pm = np.array(prev_matched, np.float32, copy=True).reshape(-1, 2)
    m = np.empty(F1.n, np.int32)
    n = _dll().lldo_search_for_initialization(C.byref(oframe(F1)), C.byref(oframe(F2)), _p(pm, c_float_p), int(window_size), np.float32(nnratio),
                                              int(check_orientation), _p(m, c_int32_p))
    return n, m, pm


def fuse_search(KF, desc, valid, uv, ur, pred_level, th=3.0):
    a = [_u32(desc), _u8(valid), _f32(uv), _f32(ur), _i32(pred_level)]
    best = np.empty(a[0].shape[0], np.int32)
    n = _dll().lldo_fuse_search(C.byref(oframe(KF)), a[0].shape[0], _p(a[0], c_uint32_p), _p(a[1], c_uint8_p), _p(a[2], c_float_p), _p(a[3], c_float_p),
                                _p(a[4], c_int32_p), th, _p(best, c_int32_p))
    return n, best


def fuse_search_sim3(KF, desc, valid, uv, pred_level, th=4.0):
    """Inner search of Fuse(KeyFrame*, Scw, ...): (nFused, bestIdx or -1 per point)."""
    a = [_u32(desc), _u8(valid), _f32(uv), _i32(pred_level)]
    best = np.empty(a[0].shape[0], np.int32)
    d = _dll()
    d.lldo_fuse_search_sim3.argtypes = [C.POINTER(OFrame), C.c_int, c_uint32_p, c_uint8_p, c_float_p, c_int32_p, C.c_float, c_int32_p]
    d.lldo_fuse_search_sim3.restype = C.c_int
    n = d.lldo_fuse_search_sim3(C.byref(oframe(KF)), a[0].shape[0], _p(a[0], c_uint32_p), _p(a[1], c_uint8_p), _p(a[2], c_float_p), _p(a[3], c_int32_p),
                                th, _p(best, c_int32_p))
    return n, best


def search_sim3_direction(KF2, desc, valid, uv, pred_level, th=7.5):
    a = [_u32(desc), _u8(valid), _f32(uv), _i32(pred_level)]
    m = np.empty(a[0].shape[0], np.int32)
    _dll().lldo_search_sim3_direction(C.byref(oframe(KF2)), a[0].shape[0], _p(a[0], c_uint32_p), _p(a[1], c_uint8_p), _p(a[2], c_float_p),
                                      _p(a[3], c_int32_p), th, _p(m, c_int32_p))
    return m


def search_by_bow_frame(KF, F, n_nodes, kf_start, kf_idx, f_start, f_idx, kf_valid, nnratio=0.7, check_orientation=True):
    a = [_i32(kf_start), _i32(kf_idx), _i32(f_start), _i32(f_idx), _u8(kf_valid)]
    fm = np.empty(F.n, np.int32)
    n = _dll().lldo_search_by_bow_frame(C.byref(oframe(KF)), C.byref(oframe(F)), n_nodes, _p(a[0], c_int32_p), _p(a[1], c_int32_p), _p(a[2], c_int32_p),
                                        _p(a[3], c_int32_p), _p(a[4], c_uint8_p), nnratio, int(check_orientation), _p(fm, c_int32_p))
    return n, fm


def search_by_bow_kf(KF1, KF2, n_nodes, start1, idx1, start2, idx2, valid1, valid2, nnratio=0.75, check_orientation=True):
    a = [_i32(start1), _i32(idx1), _i32(start2), _i32(idx2), _u8(valid1), _u8(valid2)]
    m12 = np.empty(KF1.n, np.int32)
    n = _dll().lldo_search_by_bow_kf(C.byref(oframe(KF1)), C.byref(oframe(KF2)), n_nodes, _p(a[0], c_int32_p), _p(a[1], c_int32_p), _p(a[2], c_int32_p),
                                     _p(a[3], c_int32_p), _p(a[4], c_uint8_p), _p(a[5], c_uint8_p), nnratio, int(check_orientation), _p(m12, c_int32_p))
    return n, m12


def epipolar_lines(F12, xy):
    F12 = _f32(F12).reshape(9); xy = _f32(xy).reshape(-1, 2)
    out = np.empty((xy.shape[0], 3), np.float32)
    d = _dll()
    for i in range(xy.shape[0]):
        d.lldo_epipolar_line(_p(F12, c_float_p), float(xy[i, 0]), float(xy[i, 1]), out[i].ctypes.data_as(c_float_p))
    return out


def search_for_triangulation(KF1, KF2, n_nodes, start1, idx1, start2, idx2, has_mp1, has_mp2, F12, epipole, only_stereo=False,
                             check_orientation=True):
    a = [_i32(start1), _i32(idx1), _i32(start2), _i32(idx2), _u8(has_mp1), _u8(has_mp2), _f32(F12).reshape(9)]
    m12 = np.empty(KF1.n, np.int32)
    n = _dll().lldo_search_for_triangulation(C.byref(oframe(KF1)), C.byref(oframe(KF2)), n_nodes, _p(a[0], c_int32_p), _p(a[1], c_int32_p),
                                             _p(a[2], c_int32_p), _p(a[3], c_int32_p), _p(a[4], c_uint8_p), _p(a[5], c_uint8_p), _p(a[6], c_float_p),
                                             float(np.float32(epipole[0])), float(np.float32(epipole[1])), int(only_stereo), int(check_orientation),
                                             _p(m12, c_int32_p))
    return n, m12


def stereo_search(L, R, n_rows, min_d, max_d):
    br = np.empty(L.n, np.int32); bd = np.empty(L.n, np.int32)
    _dll().lldo_stereo_search(C.byref(oframe(L)), C.byref(oframe(R)), n_rows, min_d, max_d, _p(br, c_int32_p), _p(bd, c_int32_p))
    return br, bd


def compute_stereo_matches(L, R, left_levels, right_levels, inv_scale, mb, mbf):
    """Frame::ComputeStereoMatches, whole routine (literal restatement): (n_kept, mvuRight, mvDepth, best_r, sad)."""
    from lld_slam_amd.orb_search import Keypoints, StereoPyramids, keypoints_struct, pyramids_struct
    d = _dll()
    d.lldo_compute_stereo_matches.argtypes = [C.POINTER(Keypoints), C.POINTER(Keypoints), C.POINTER(StereoPyramids), C.c_float, C.c_float,
                                              c_float_p, c_float_p, c_int32_p, c_int32_p]
    d.lldo_compute_stereo_matches.restype = C.c_int
    kl, kr = keypoints_struct(L), keypoints_struct(R)
    P, keep = pyramids_struct(left_levels, right_levels, L.scale, inv_scale)
    ur = np.empty(L.n, np.float32); dep = np.empty(L.n, np.float32); br = np.empty(L.n, np.int32); sad = np.empty(L.n, np.int32)
    n = d.lldo_compute_stereo_matches(C.byref(kl), C.byref(kr), C.byref(P), float(np.float32(mb)), float(np.float32(mbf)),
                                      _p(ur, c_float_p), _p(dep, c_float_p), _p(br, c_int32_p), _p(sad, c_int32_p))
    return n, ur, dep, br, sad


def is_in_frustum(view, mp: dict, viewing_cos_limit=0.5):
    """Frame::isInFrustum over all map points (literal restatement): (nToMatch, in_view, proj_uvr, level, view_cos)."""
    from lld_slam_amd.orb_search import FrameView, MapPoints, map_points_struct
    d = _dll()
    d.lldo_is_in_frustum.argtypes = [C.POINTER(FrameView), C.POINTER(MapPoints), C.c_float, c_uint8_p, c_float_p, c_int32_p, c_float_p]
    d.lldo_is_in_frustum.restype = C.c_int
    m, keep = map_points_struct(mp)
    n = m.n
    in_view = np.zeros(n, np.uint8); uvr = np.zeros((n, 3), np.float32); level = np.zeros(n, np.int32); vc = np.zeros(n, np.float32)
    k = d.lldo_is_in_frustum(C.byref(view), C.byref(m), float(np.float32(viewing_cos_limit)), _p(in_view, c_uint8_p), _p(uvr, c_float_p),
                             _p(level, c_int32_p), _p(vc, c_float_p))
    return k, in_view, uvr, level, vc


def project_last_frame(view, last: dict):
    """Projection loop of SearchByProjection(Current, Last): (valid, uv [n,2], ur [n])."""
    from lld_slam_amd.orb_search import FrameView, LastFramePoints, last_frame_struct
    d = _dll()
    d.lldo_project_last_frame.argtypes = [C.POINTER(FrameView), C.POINTER(LastFramePoints), c_uint8_p, c_float_p, c_float_p]
    d.lldo_project_last_frame.restype = None
    m, keep = last_frame_struct(last)
    n = m.n
    valid = np.zeros(n, np.uint8); uv = np.zeros((n, 2), np.float32); ur = np.zeros(n, np.float32)
    d.lldo_project_last_frame(C.byref(view), C.byref(m), _p(valid, c_uint8_p), _p(uv, c_float_p), _p(ur, c_float_p))
    return valid, uv, ur


def project_fuse(view, mp: dict):
    """Projection loop of Fuse(KeyFrame*, vpMapPoints, th): (valid, uv [n,2], ur [n], level [n])."""
    from lld_slam_amd.orb_search import FrameView, MapPoints, map_points_struct
    d = _dll()
    d.lldo_project_fuse.argtypes = [C.POINTER(FrameView), C.POINTER(MapPoints), c_uint8_p, c_float_p, c_float_p, c_int32_p]
    d.lldo_project_fuse.restype = None
    m, keep = map_points_struct(mp)
    n = m.n
    valid = np.zeros(n, np.uint8); uv = np.zeros((n, 2), np.float32); ur = np.zeros(n, np.float32); lvl = np.zeros(n, np.int32)
    d.lldo_project_fuse(C.byref(view), C.byref(m), _p(valid, c_uint8_p), _p(uv, c_float_p), _p(ur, c_float_p), _p(lvl, c_int32_p))
    return valid, uv, ur, lvl


def project_general(view, mp: dict, routine: int, sR=None, t=None):
    """Projection loops of SearchByProjection(KeyFrame*, Scw) / SearchByProjection(Frame&, KeyFrame*) / Fuse(KeyFrame*, Scw) /
    one direction of SearchBySim3 (routine 0..3): (valid, uv [n,2], level [n])."""
    from lld_slam_amd.orb_search import FrameView, MapPoints, map_points_struct
    d = _dll()
    d.lldo_project_general.argtypes = [C.POINTER(FrameView), C.POINTER(MapPoints), C.c_int, c_float_p, c_float_p, c_uint8_p, c_float_p, c_int32_p]
    d.lldo_project_general.restype = None
    m, keep = map_points_struct(mp)
    n = m.n
    sR = np.ascontiguousarray(np.zeros(9) if sR is None else sR, np.float32).reshape(9)
    t = np.ascontiguousarray(np.zeros(3) if t is None else t, np.float32).reshape(3)
    valid = np.zeros(n, np.uint8); uv = np.zeros((n, 2), np.float32); lvl = np.zeros(n, np.int32)
    d.lldo_project_general(C.byref(view), C.byref(m), int(routine), _p(sR, c_float_p), _p(t, c_float_p), _p(valid, c_uint8_p), _p(uv, c_float_p),
                           _p(lvl, c_int32_p))
    return valid, uv, lvl


def glibc_logf_differences(n: int, seed: int = 1):
    """(count, first argument) where the restated glibc logf (what the device carries for MapPoint::PredictScale) differs from this host's
    std::log(float), over n pseudo-random positive floats plus 8193 neighbours of every 1.2^k, k = -8..16."""
    d = _dll()
    d.lldo_glibc_logf_differences.argtypes = [C.c_long, C.c_ulonglong, c_float_p]; d.lldo_glibc_logf_differences.restype = C.c_long
    first = np.zeros(1, np.float32)
    return int(d.lldo_glibc_logf_differences(int(n), int(seed), _p(first, c_float_p))), float(first[0])
