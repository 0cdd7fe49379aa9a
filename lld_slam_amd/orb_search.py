"""Host-side mirror of the reference's guided ORB searches over ``lld_orb_search_run`` (include/lld_amd.h).

Every function here configures ONE device call so that it reproduces one routine of the reference
(src/ORBmatcher.cc, src/Frame.cc:530-613): which candidate generator, which skip rules, which accept rule, whether earlier
matches occupy keypoints and whether the rotation histogram runs.  What the reference computes per query BEFORE its inner
loop from cv::Mat poses (projection, predicted octave, viewing cosine ...) is an input, in float32 like the reference's.
No CPU fallback: without the HIP library / a GPU these raise.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import abi
from .abi import c_float_p, c_int32_p, c_uint32_p, c_uint8_p

CAND_ALL, CAND_GRID, CAND_CSR, CAND_ROWS = 0, 1, 2, 3
GATE_LEVEL, GATE_STEREO, GATE_CHI2, GATE_EPIPOLAR = 1, 2, 4, 8
FRAME_GRID_ROWS, FRAME_GRID_COLS = 48, 64          # include/Frame.h:43-44
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30        # src/ORBmatcher.cc:37-39
MAX_KEYPOINTS, MAX_LEVELS = 4096, 16


class OrbSearch(C.Structure):
    _fields_ = [
        ("nt", C.c_int32), ("t_desc", c_uint32_p), ("t_xy", c_float_p), ("t_octave", c_int32_p), ("t_uright", c_float_p),
        ("t_angle", c_float_p), ("t_occupied", c_uint8_p),
        ("nq", C.c_int32), ("q_desc", c_uint32_p), ("q_valid", c_uint8_p), ("q_blocks", c_uint8_p), ("q_uv", c_float_p),
        ("q_radius", c_float_p), ("q_level_min", c_int32_p), ("q_level_max", c_int32_p), ("q_uright", c_float_p),
        ("q_stereo_radius", c_float_p), ("q_angle", c_float_p), ("q_epiline", c_float_p), ("q_stereo", c_uint8_p),
        ("cand_range", c_int32_p), ("cand_idx", c_int32_p), ("n_cand", C.c_int32),
        ("grid_min_x", C.c_float), ("grid_min_y", C.c_float), ("grid_width_inv", C.c_float), ("grid_height_inv", C.c_float),
        ("grid_cols", C.c_int32), ("grid_rows", C.c_int32),
        ("n_levels", C.c_int32), ("level_scale", c_float_p), ("level_sigma2", c_float_p), ("level_inv_sigma2", c_float_p),
        ("disp_min", C.c_float), ("disp_max", C.c_float), ("epipole_x", C.c_float), ("epipole_y", C.c_float),
        ("only_stereo", C.c_int32),
        ("candidates", C.c_int32), ("gates", C.c_int32), ("tie_last", C.c_int32), ("accept_max", C.c_int32),
        ("ratio_mode", C.c_int32), ("nnratio", C.c_float), ("sequential", C.c_int32), ("check_orientation", C.c_int32),
    ]


class OrbSearchResult(C.Structure):
    _fields_ = [("match", c_int32_p), ("best_dist", c_int32_p), ("second_dist", c_int32_p), ("removed", c_uint8_p),
                ("owner", c_int32_p), ("n_matches", C.c_int32), ("rounds", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float), ("log_scale_factor", C.c_float), ("n_levels", C.c_int32)]


class MapPoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("world_pos", c_float_p), ("normal", c_float_p), ("max_distance", c_float_p),
                ("min_distance", c_float_p), ("desc", c_uint32_p), ("has_obs", c_uint8_p), ("skip", c_uint8_p)]


class LastFramePoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("world_pos", c_float_p), ("valid", c_uint8_p), ("octave", c_int32_p), ("angle", c_float_p),
                ("desc", c_uint32_p), ("has_obs", c_uint8_p)]


class FrustumResult(C.Structure):
    _fields_ = [("in_view", c_uint8_p), ("proj_uvr", c_float_p), ("level", c_int32_p), ("view_cos", c_float_p)]


class Keypoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("xy", c_float_p), ("octave", c_int32_p), ("desc", c_uint32_p)]


class StereoPyramids(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("left", C.POINTER(c_uint8_p)), ("right", C.POINTER(c_uint8_p)), ("cols", c_int32_p), ("rows", c_int32_p),
                ("left_step", c_int32_p), ("right_step", c_int32_p), ("scale_factors", c_float_p), ("inv_scale_factors", c_float_p),
                ("on_device", C.c_int32), ("reserved", C.c_int32)]


class StereoResult(C.Structure):
    _fields_ = [("u_right", c_float_p), ("depth", c_float_p), ("best_r", c_int32_p), ("sad", c_int32_p), ("n_matches", C.c_int32),
                ("reserved", C.c_int32)]


def orb_levels(scale_factor=1.2, n_levels=8):
    """mvScaleFactor / mvLevelSigma2 / mvInvLevelSigma2 exactly as ORBextractor builds them (src/ORBextractor.cc:416-430):
    cumulative float products."""
    scale = np.ones(n_levels, np.float32)
    for i in range(1, n_levels):
        scale[i] = np.float32(scale[i - 1] * np.float32(scale_factor))
    sigma2 = (scale * scale).astype(np.float32)
    return scale, sigma2, (np.float32(1.0) / sigma2).astype(np.float32)


@dataclass
class Frame:
    """The members of Frame / KeyFrame the searches read (same names modulo the mv prefix)."""
    desc: np.ndarray            # [n,8] uint32  mDescriptors
    xy: np.ndarray              # [n,2] float32 mvKeysUn[i].pt
    octave: np.ndarray          # [n]   int32
    uright: np.ndarray          # [n]   float32 mvuRight (-1: mono)
    angle: np.ndarray           # [n]   float32 mvKeysUn[i].angle
    min_x: float = 0.0
    min_y: float = 0.0
    max_x: float = 1241.0
    max_y: float = 376.0
    scale: np.ndarray = field(default_factory=lambda: orb_levels()[0])
    sigma2: np.ndarray = field(default_factory=lambda: orb_levels()[1])
    inv_sigma2: np.ndarray = field(default_factory=lambda: orb_levels()[2])

    def normalise(self):
        self.desc = np.ascontiguousarray(self.desc, np.uint32).reshape(-1, 8)
        self.xy = np.ascontiguousarray(self.xy, np.float32).reshape(-1, 2)
        self.octave = np.ascontiguousarray(self.octave, np.int32)
        self.uright = np.ascontiguousarray(self.uright, np.float32)
        self.angle = np.ascontiguousarray(self.angle, np.float32)
        for name in ("scale", "sigma2", "inv_sigma2"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), np.float32))
        return self

    @property
    def n(self): return self.desc.shape[0]

    @property
    def width_inv(self):        # mfGridElementWidthInv, src/Frame.cc:272
        return np.float32(FRAME_GRID_COLS) / (np.float32(self.max_x) - np.float32(self.min_x))

    @property
    def height_inv(self):
        return np.float32(FRAME_GRID_ROWS) / (np.float32(self.max_y) - np.float32(self.min_y))


@dataclass
class SearchOutput:
    match: np.ndarray           # [nq] accepted keypoint (before the orientation filter) or -1
    best_dist: np.ndarray
    second_dist: np.ndarray
    removed: np.ndarray         # [nq] 1: dropped by the rotation histogram
    owner: np.ndarray           # [nt] query holding keypoint k at the end; -1 untouched, -2 NULLed by the rotation filter
    n_matches: int              # the reference routine's return value
    rounds: int

    def final_match(self):
        """Per query: the keypoint it still holds after the orientation filter (query-space view, e.g. vpMatches12)."""
        return np.where(self.removed != 0, -1, self.match).astype(np.int32)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a): return None if a is None else np.ascontiguousarray(a, np.float32)
def _i32(a): return None if a is None else np.ascontiguousarray(a, np.int32)
def _u8(a): return None if a is None else np.ascontiguousarray(a, np.uint8)


@dataclass
class Prepared:
    """One configured problem: the C struct, the arrays it points into and the output holder."""
    s: OrbSearch
    r: OrbSearchResult
    out: SearchOutput
    keep: tuple


def prepare(T: Frame, q_desc, *, candidates, gates=0, accept_max, ratio_mode=0, nnratio=0.0, sequential=False,
            check_orientation=False, tie_last=False, t_occupied=None, q_valid=None, q_blocks=None, q_uv=None, q_radius=None,
            q_level_min=None, q_level_max=None, q_uright=None, q_stereo_radius=None, q_angle=None, q_epiline=None, q_stereo=None,
            cand_range=None, cand_idx=None, disp_min=0.0, disp_max=0.0, epipole=(0.0, 0.0), only_stereo=False) -> Prepared:
    T.normalise()
    q_desc = np.ascontiguousarray(q_desc, np.uint32).reshape(-1, 8)
    nq, nt = q_desc.shape[0], T.n
    keep = dict(t_occupied=_u8(t_occupied), q_valid=_u8(q_valid), q_blocks=_u8(q_blocks), q_uv=_f32(q_uv), q_radius=_f32(q_radius),
                q_level_min=_i32(q_level_min), q_level_max=_i32(q_level_max), q_uright=_f32(q_uright),
                q_stereo_radius=_f32(q_stereo_radius), q_angle=_f32(q_angle), q_epiline=_f32(q_epiline), q_stereo=_u8(q_stereo),
                cand_range=_i32(cand_range), cand_idx=_i32(cand_idx))
    n_cand = 0 if keep["cand_idx"] is None else int(keep["cand_idx"].size)
    if keep["cand_idx"] is not None and n_cand == 0:
        keep["cand_idx"] = np.zeros(1, np.int32)
    s = OrbSearch()
    s.nt = nt; s.t_desc = _p(T.desc, c_uint32_p); s.t_xy = _p(T.xy, c_float_p); s.t_octave = _p(T.octave, c_int32_p)
    s.t_uright = _p(T.uright, c_float_p); s.t_angle = _p(T.angle, c_float_p); s.t_occupied = _p(keep["t_occupied"], c_uint8_p)
    s.nq = nq; s.q_desc = _p(q_desc, c_uint32_p)
    s.q_valid = _p(keep["q_valid"], c_uint8_p); s.q_blocks = _p(keep["q_blocks"], c_uint8_p); s.q_uv = _p(keep["q_uv"], c_float_p)
    s.q_radius = _p(keep["q_radius"], c_float_p); s.q_level_min = _p(keep["q_level_min"], c_int32_p)
    s.q_level_max = _p(keep["q_level_max"], c_int32_p); s.q_uright = _p(keep["q_uright"], c_float_p)
    s.q_stereo_radius = _p(keep["q_stereo_radius"], c_float_p); s.q_angle = _p(keep["q_angle"], c_float_p)
    s.q_epiline = _p(keep["q_epiline"], c_float_p); s.q_stereo = _p(keep["q_stereo"], c_uint8_p)
    s.cand_range = _p(keep["cand_range"], c_int32_p); s.cand_idx = _p(keep["cand_idx"], c_int32_p); s.n_cand = n_cand
    s.grid_min_x = float(np.float32(T.min_x)); s.grid_min_y = float(np.float32(T.min_y))
    s.grid_width_inv = float(T.width_inv); s.grid_height_inv = float(T.height_inv)
    s.grid_cols = FRAME_GRID_COLS; s.grid_rows = FRAME_GRID_ROWS
    s.n_levels = T.scale.shape[0]; s.level_scale = _p(T.scale, c_float_p); s.level_sigma2 = _p(T.sigma2, c_float_p)
    s.level_inv_sigma2 = _p(T.inv_sigma2, c_float_p)
    s.disp_min = float(np.float32(disp_min)); s.disp_max = float(np.float32(disp_max))
    s.epipole_x = float(np.float32(epipole[0])); s.epipole_y = float(np.float32(epipole[1])); s.only_stereo = int(only_stereo)
    s.candidates = candidates; s.gates = gates; s.tie_last = int(tie_last); s.accept_max = int(accept_max)
    s.ratio_mode = ratio_mode; s.nnratio = float(np.float32(nnratio)); s.sequential = int(sequential)
    s.check_orientation = int(check_orientation)
    out = SearchOutput(np.empty(nq, np.int32), np.empty(nq, np.int32), np.empty(nq, np.int32), np.empty(nq, np.uint8),
                       np.empty(nt, np.int32), 0, 0)
    r = OrbSearchResult()
    r.match = _p(out.match, c_int32_p); r.best_dist = _p(out.best_dist, c_int32_p); r.second_dist = _p(out.second_dist, c_int32_p)
    r.removed = _p(out.removed, c_uint8_p); r.owner = _p(out.owner, c_int32_p)
    return Prepared(s, r, out, (T, q_desc, keep))


def run_batch(lib: abi.Lib, ctx, prepared: list) -> list:
    """lld_orb_search_batch over already configured problems (pass ``lib=None`` to any routine below to get a Prepared
    instead of running it): one launch, one workgroup per problem."""
    n = len(prepared)
    S = (OrbSearch * n)(*[p.s for p in prepared]); R = (OrbSearchResult * n)(*[p.r for p in prepared])
    fn = lib.fn("orb_search_batch")
    fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrbSearch), C.POINTER(OrbSearchResult)]; fn.restype = C.c_int
    st = fn(ctx, n, S, R)
    if st != abi.LLD_OK:
        raise RuntimeError(f"lld_orb_search_batch failed: {lib.fn('status_string')(st).decode()}")
    for p, r in zip(prepared, R):
        p.out.n_matches, p.out.rounds = r.n_matches, r.rounds
    return [p.out for p in prepared]


def run(lib, ctx, T: Frame, q_desc, **kw):
    """One lld_orb_search_run call (or, with lib None, the configured problem for run_batch)."""
    p = prepare(T, q_desc, **kw)
    if lib is None:
        return p
    fn = lib.fn("orb_search_run")
    fn.argtypes = [C.c_void_p, C.POINTER(OrbSearch), C.POINTER(OrbSearchResult)]; fn.restype = C.c_int
    st = fn(ctx, C.byref(p.s), C.byref(p.r))
    if st != abi.LLD_OK:
        raise RuntimeError(f"lld_orb_search_run failed: {lib.fn('status_string')(st).decode()}")
    p.out.n_matches, p.out.rounds = p.r.n_matches, p.r.rounds
    return p.out


def bow_queries(start1, idx1, start2, n_nodes):
    """Flatten the merge loop over two FeatureVectors (src/ORBmatcher.cc:183-251): the query order is node-major, and the
    candidates of a query are the second frame's indices of the same node.  Returns (query keypoint index per query,
    cand_range [nq,2]) with cand_idx = the second frame's flat index array."""
    start1, start2 = np.asarray(start1), np.asarray(start2)
    order = np.ascontiguousarray(idx1, np.int32)
    node_of = np.repeat(np.arange(n_nodes), np.diff(start1))
    cr = np.empty((order.shape[0], 2), np.int32)
    cr[:, 0] = start2[node_of]; cr[:, 1] = start2[node_of + 1]
    return order, cr


# ====================================================================== the reference's routines, one device call each
def _radius_by_viewing_cos(view_cos):
    """ORBmatcher::RadiusByViewingCos (src/ORBmatcher.cc:131-137): the float is compared with the double literal 0.998."""
    return np.where(np.asarray(view_cos, np.float32).astype(np.float64) > 0.998, np.float32(2.5), np.float32(4.0)).astype(np.float32)


def search_by_projection_map(lib, ctx, F: Frame, mp_desc, in_view, proj, proj_xr, pred_level, view_cos, mp_obs, f_occupied,
                             th=1.0, nnratio=0.6) -> SearchOutput:
    """ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)  (src/ORBmatcher.cc:45-129).
    in_view = mbTrackInView && !isBad; proj = (mTrackProjX, mTrackProjY); proj_xr = mTrackProjXR; pred_level =
    mnTrackScaleLevel; view_cos = mTrackViewCos; mp_obs = Observations()>0; f_occupied[k] = F.mvpMapPoints[k] has obs>0."""
    F.normalise()
    pred_level = np.asarray(pred_level, np.int32)
    r = _radius_by_viewing_cos(view_cos)
    if np.float32(th) != np.float32(1.0):                       # bFactor
        r = (r * np.float32(th)).astype(np.float32)
    radius = (r * F.scale[pred_level]).astype(np.float32)       # r*F.mvScaleFactors[nPredictedLevel]
    return run(lib, ctx, F, mp_desc, candidates=CAND_GRID, gates=GATE_LEVEL | GATE_STEREO, accept_max=TH_HIGH, ratio_mode=2,
               nnratio=nnratio, sequential=True, t_occupied=f_occupied, q_valid=in_view, q_blocks=mp_obs, q_uv=proj, q_radius=radius,
               q_level_min=pred_level - 1, q_level_max=pred_level, q_uright=proj_xr, q_stereo_radius=radius)


def search_by_projection_frame(lib, ctx, Cur: Frame, last_desc, valid, uv, ur, last_octave, last_angle, mp_obs, cur_occupied,
                               direction=0, th=7.0, check_orientation=True) -> SearchOutput:
    """ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono)  (src/ORBmatcher.cc:1328-1470).
    direction: +1 bForward, -1 bBackward, 0 neither (:1349-1350); ur = u - mbf*invzc (:1402)."""
    Cur.normalise()
    oct_ = np.asarray(last_octave, np.int32)
    radius = (np.float32(th) * Cur.scale[oct_]).astype(np.float32)
    if direction > 0:
        lmin, lmax = oct_, np.full_like(oct_, -1)               # GetFeaturesInArea(u,v,radius,nLastOctave)
    elif direction < 0:
        lmin, lmax = np.zeros_like(oct_), oct_                  # (u,v,radius,0,nLastOctave)
    else:
        lmin, lmax = oct_ - 1, oct_ + 1
    return run(lib, ctx, Cur, last_desc, candidates=CAND_GRID, gates=GATE_LEVEL | GATE_STEREO, accept_max=TH_HIGH, sequential=True,
               check_orientation=check_orientation, t_occupied=cur_occupied, q_valid=valid, q_blocks=mp_obs, q_uv=uv, q_radius=radius,
               q_level_min=lmin, q_level_max=lmax, q_uright=ur, q_stereo_radius=radius, q_angle=last_angle)


def search_by_projection_reloc(lib, ctx, Cur: Frame, desc, valid, uv, pred_level, kf_angle, cur_occupied, th, ORBdist,
                               check_orientation=True) -> SearchOutput:
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)  (src/ORBmatcher.cc:1472-1599).
    cur_occupied[k] = CurrentFrame.mvpMapPoints[k] != NULL (any MapPoint blocks, :1542-1543)."""
    Cur.normalise()
    lvl = np.asarray(pred_level, np.int32)
    radius = (np.float32(th) * Cur.scale[lvl]).astype(np.float32)
    return run(lib, ctx, Cur, desc, candidates=CAND_GRID, gates=GATE_LEVEL, accept_max=int(ORBdist), sequential=True,
               check_orientation=check_orientation, t_occupied=cur_occupied, q_valid=valid, q_uv=uv, q_radius=radius,
               q_level_min=lvl - 1, q_level_max=lvl + 1, q_angle=kf_angle)


def search_by_projection_kf(lib, ctx, KF: Frame, desc, valid, uv, pred_level, matched, th: int) -> SearchOutput:
    """ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)  (src/ORBmatcher.cc:290-403)."""
    KF.normalise()
    lvl = np.asarray(pred_level, np.int32)
    radius = (np.float32(int(th)) * KF.scale[lvl]).astype(np.float32)
    return run(lib, ctx, KF, desc, candidates=CAND_GRID, gates=GATE_LEVEL, accept_max=TH_LOW, sequential=True, t_occupied=matched,
               q_valid=valid, q_uv=uv, q_radius=radius, q_level_min=lvl - 1, q_level_max=lvl)


def search_for_initialization(lib, ctx, F1: Frame, F2: Frame, prev_matched, window_size=10, nnratio=0.9, check_orientation=True, info=None):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520): octave-0 keypoints of F1 against the octave-0 keypoints of F2
    inside a square window around vbPrevMatched; a keypoint of F2 goes to the query with the smallest distance so far (`sequential = 2`:
    an earlier holder with a distance <= this one blocks, a better one steals).  Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
    F1.normalise(); F2.normalise()
    pm = np.array(prev_matched, np.float32, copy=True).reshape(-1, 2)
    lvl = np.zeros(F1.n, np.int32)
    out = run(lib, ctx, F2, F1.desc, candidates=CAND_GRID, gates=GATE_LEVEL, accept_max=TH_LOW, ratio_mode=1, nnratio=nnratio, sequential=2,
              check_orientation=check_orientation, q_valid=(F1.octave <= 0).astype(np.uint8), q_uv=pm, q_radius=np.full(F1.n, np.float32(int(window_size))),
              q_level_min=lvl, q_level_max=lvl, q_angle=F1.angle)
    if info is not None: info["rescans"] = out.rounds - 1                     # queries whose cached candidate list ran dry
    m12 = np.where(out.removed != 0, -1, out.match).astype(np.int32)
    ok = m12 >= 0
    pm[ok] = F2.xy[m12[ok]]
    return out.n_matches, m12, pm


def fuse_search(lib, ctx, KF: Frame, desc, valid, uv, ur, pred_level, th=3.0) -> SearchOutput:
    """Inner search of ORBmatcher::Fuse (src/ORBmatcher.cc:825-958; the Scw overload :960-1100 has the same loop):
    match[i] = bestIdx when bestDist<=TH_LOW; n_matches = nFused.  The replace / add bookkeeping stays with the caller."""
    KF.normalise()
    lvl = np.asarray(pred_level, np.int32)
    radius = (np.float32(th) * KF.scale[lvl]).astype(np.float32)
    return run(lib, ctx, KF, desc, candidates=CAND_GRID, gates=GATE_LEVEL | GATE_CHI2, accept_max=TH_LOW, q_valid=valid, q_uv=uv,
               q_radius=radius, q_level_min=lvl - 1, q_level_max=lvl, q_uright=ur)


def search_sim3_direction(lib, ctx, KF2: Frame, desc, valid, uv, pred_level, th=7.5) -> SearchOutput:
    """One direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1147-1224 / 1227-1304): match = vnMatch."""
    KF2.normalise()
    lvl = np.asarray(pred_level, np.int32)
    radius = (np.float32(th) * KF2.scale[lvl]).astype(np.float32)
    return run(lib, ctx, KF2, desc, candidates=CAND_GRID, gates=GATE_LEVEL, accept_max=TH_HIGH, q_valid=valid, q_uv=uv,
               q_radius=radius, q_level_min=lvl - 1, q_level_max=lvl)


def search_by_sim3(lib, ctx, KF1: Frame, KF2: Frame, q1, q2, th=7.5):
    """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1102-1326): both directions on the device, then the agreement check
    (:1306-1322).  q1 / q2 = dicts(desc, valid, uv, pred_level) for the KF1 points projected into KF2 and vice versa.
    Returns (vpMatches12 as KF2 keypoint index per KF1 keypoint or -1, nFound)."""
    m1 = search_sim3_direction(lib, ctx, KF2, q1["desc"], q1["valid"], q1["uv"], q1["pred_level"], th).match
    m2 = search_sim3_direction(lib, ctx, KF1, q2["desc"], q2["valid"], q2["uv"], q2["pred_level"], th).match
    out = np.full(m1.shape[0], -1, np.int32)
    ok = m1 >= 0
    ok[ok] = m2[m1[ok]] == np.nonzero(ok)[0]
    out[ok] = m1[ok]
    return out, int(ok.sum())


def search_by_bow_frame(lib, ctx, KF: Frame, F: Frame, n_nodes, kf_start, kf_idx, f_start, f_idx, kf_valid, nnratio=0.7,
                        check_orientation=True) -> SearchOutput:
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)  (src/ORBmatcher.cc:159-288).  The common vocabulary
    nodes arrive as CSR lists over both frames; queries run node-major like the merge loop.  owner[k] = position of the
    winning query in that order; use `query_kp[owner]` for the KF keypoint index."""
    order, cr = bow_queries(kf_start, kf_idx, f_start, n_nodes)
    kf_valid = np.asarray(kf_valid, np.uint8)
    out = run(lib, ctx, F, KF.normalise().desc[order], candidates=CAND_CSR, accept_max=TH_LOW, ratio_mode=1, nnratio=nnratio,
              sequential=True, check_orientation=check_orientation, q_valid=kf_valid[order], q_angle=KF.angle[order], cand_range=cr,
              cand_idx=f_idx)
    (out.out if isinstance(out, Prepared) else out).query_kp = order
    return out


def search_by_bow_kf(lib, ctx, KF1: Frame, KF2: Frame, n_nodes, start1, idx1, start2, idx2, valid1, valid2, nnratio=0.75,
                     check_orientation=True) -> SearchOutput:
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)  (src/ORBmatcher.cc:522-655): `bestDist1<TH_LOW` is strict
    (:586), KF2 keypoints without a good MapPoint are never candidates (:571-575)."""
    order, cr = bow_queries(start1, idx1, start2, n_nodes)
    out = run(lib, ctx, KF2, KF1.normalise().desc[order], candidates=CAND_CSR, accept_max=TH_LOW - 1, ratio_mode=1, nnratio=nnratio,
              sequential=True, check_orientation=check_orientation, t_occupied=1 - np.asarray(valid2, np.uint8),
              q_valid=np.asarray(valid1, np.uint8)[order], q_angle=KF1.angle[order], cand_range=cr, cand_idx=idx2)
    (out.out if isinstance(out, Prepared) else out).query_kp = order
    return out


def search_for_triangulation(lib, ctx, KF1: Frame, KF2: Frame, n_nodes, start1, idx1, start2, idx2, has_mp1, has_mp2, epilines,
                             epipole, only_stereo=False, check_orientation=True) -> SearchOutput:
    """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-823).  epilines[k] = (a,b,c) of KF1 keypoint k as
    CheckDistEpipolarLine forms them (:141-143).  A later candidate with an equal distance replaces the earlier one
    (`dist>bestDist -> continue`, :733), hence tie_last; vbMatched2 is never written by the reference, so no occupancy."""
    order, cr = bow_queries(start1, idx1, start2, n_nodes)
    KF1.normalise()
    stereo1 = (KF1.uright >= 0).astype(np.uint8)
    valid = (1 - np.asarray(has_mp1, np.uint8))
    if only_stereo:
        valid = valid & stereo1
    out = run(lib, ctx, KF2, KF1.desc[order], candidates=CAND_CSR, gates=GATE_EPIPOLAR, accept_max=TH_LOW, tie_last=True,
              check_orientation=check_orientation, t_occupied=has_mp2, q_valid=valid[order], q_angle=KF1.angle[order],
              q_epiline=np.asarray(epilines, np.float32)[order], q_stereo=stereo1[order], cand_range=cr, cand_idx=idx2,
              epipole=epipole, only_stereo=only_stereo)
    (out.out if isinstance(out, Prepared) else out).query_kp = order
    return out


def stereo_search(lib, ctx, L: Frame, R: Frame, min_d, max_d) -> SearchOutput:
    """Hamming search of Frame::ComputeStereoMatches (src/Frame.cc:530-613): match[iL] = bestIdxR when bestDist <
    (TH_HIGH+TH_LOW)/2, and bestDist starts at TH_HIGH (:580), i.e. accept iff best <= 74."""
    L.normalise()
    return run(lib, ctx, R, L.desc, candidates=CAND_ROWS, gates=GATE_LEVEL, accept_max=(TH_HIGH + TH_LOW) // 2 - 1, q_uv=L.xy,
               q_level_min=L.octave - 1, q_level_max=L.octave + 1, disp_min=min_d, disp_max=max_d)



@dataclass
class StereoMatches:
    u_right: np.ndarray      # mvuRight
    depth: np.ndarray        # mvDepth
    best_r: np.ndarray       # bestIdxR of the Hamming stage
    sad: np.ndarray          # SAD bestDist of the entries pushed into vDistIdx
    n_matches: int


def keypoints_struct(F: Frame):
    F.normalise()
    k = Keypoints(); k.n = F.n
    k.xy = _p(F.xy, c_float_p); k.octave = _p(F.octave, c_int32_p); k.desc = _p(F.desc, c_uint32_p)
    return k


def pyramids_struct(left_levels, right_levels, scale, inv_scale):
    """left_levels / right_levels: lists of 2-D uint8 arrays (mvImagePyramid); rows may be strided (cv::Mat::step)."""
    n = len(left_levels)
    keep = dict(left=[np.asarray(a, np.uint8) for a in left_levels], right=[np.asarray(a, np.uint8) for a in right_levels])
    for side in ("left", "right"):
        keep[side] = [a if a.strides[1] == 1 else np.ascontiguousarray(a) for a in keep[side]]
    keep["cols"] = np.array([a.shape[1] for a in keep["left"]], np.int32); keep["rows"] = np.array([a.shape[0] for a in keep["left"]], np.int32)
    keep["lstep"] = np.array([a.strides[0] for a in keep["left"]], np.int32); keep["rstep"] = np.array([a.strides[0] for a in keep["right"]], np.int32)
    keep["scale"] = _f32(scale); keep["inv"] = _f32(inv_scale)
    keep["lp"] = (c_uint8_p * n)(*[a.ctypes.data_as(c_uint8_p) for a in keep["left"]])
    keep["rp"] = (c_uint8_p * n)(*[a.ctypes.data_as(c_uint8_p) for a in keep["right"]])
    P = StereoPyramids(); P.n_levels = n
    P.left = C.cast(keep["lp"], C.POINTER(c_uint8_p)); P.right = C.cast(keep["rp"], C.POINTER(c_uint8_p))
    P.cols = _p(keep["cols"], c_int32_p); P.rows = _p(keep["rows"], c_int32_p)
    P.left_step = _p(keep["lstep"], c_int32_p); P.right_step = _p(keep["rstep"], c_int32_p)
    P.scale_factors = _p(keep["scale"], c_float_p); P.inv_scale_factors = _p(keep["inv"], c_float_p)
    P.on_device = 0
    return P, keep


def compute_stereo_matches(lib, ctx, L: Frame, R: Frame, left_levels, right_levels, inv_scale, mb, mbf) -> StereoMatches:
    """Frame::ComputeStereoMatches, whole routine (src/Frame.cc:530-704): lld_compute_stereo_matches."""
    kl, kr = keypoints_struct(L), keypoints_struct(R)
    P, keep = pyramids_struct(left_levels, right_levels, L.scale, inv_scale)
    out = StereoMatches(np.empty(L.n, np.float32), np.empty(L.n, np.float32), np.empty(L.n, np.int32), np.empty(L.n, np.int32), 0)
    r = StereoResult(); r.u_right = _p(out.u_right, c_float_p); r.depth = _p(out.depth, c_float_p)
    r.best_r = _p(out.best_r, c_int32_p); r.sad = _p(out.sad, c_int32_p)
    fn = lib.fn("compute_stereo_matches")
    fn.argtypes = [C.c_void_p, C.POINTER(Keypoints), C.POINTER(Keypoints), C.POINTER(StereoPyramids), C.c_float, C.c_float, C.POINTER(StereoResult)]
    fn.restype = C.c_int
    st = fn(ctx, C.byref(kl), C.byref(kr), C.byref(P), float(np.float32(mb)), float(np.float32(mbf)), C.byref(r))
    if st != abi.LLD_OK:
        raise RuntimeError(f"lld_compute_stereo_matches failed: {lib.fn('status_string')(st).decode()}")
    out.n_matches = r.n_matches
    return out


# ====================================================================== Tracking::SearchLocalPoints: frustum test + search on the device
def frame_view(Tcw, cam, F: Frame, scale_factor=1.2) -> FrameView:
    """Frame::UpdatePoseMatrices (src/Frame.cc:325-331) from a float32 4x4 Tcw: mRcw, mtcw, mOw = -mRcw.t()*mtcw (one cv::gemm:
    double accumulation, one rounding), plus the intrinsics, image bounds and scale constants the frustum test reads."""
    T = np.asarray(Tcw, np.float32).reshape(4, 4)
    R = T[:3, :3]; t = T[:3, 3]
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    v = FrameView()
    for i, x in enumerate(R.reshape(9)): v.Rcw[i] = float(x)
    for i in range(3): v.tcw[i] = float(t[i]); v.Ow[i] = float(Ow[i])
    v.fx, v.fy, v.cx, v.cy, v.bf = [float(np.float32(c)) for c in cam]
    v.min_x, v.max_x, v.min_y, v.max_y = float(np.float32(F.min_x)), float(np.float32(F.max_x)), float(np.float32(F.min_y)), float(np.float32(F.max_y))
    v.log_scale_factor = float(np.log(np.float32(scale_factor)))        # mfLogScaleFactor = log(mfScaleFactor): float log
    v.n_levels = int(F.scale.shape[0])
    return v


def map_points_struct(mp: dict):
    nrm = _f32(mp.get("normal"))
    keep = dict(world_pos=_f32(mp["world_pos"]).reshape(-1, 3), normal=None if nrm is None else nrm.reshape(-1, 3), max_distance=_f32(mp["max_distance"]),
                min_distance=_f32(mp["min_distance"]), desc=np.ascontiguousarray(mp["desc"], np.uint32).reshape(-1, 8),
                has_obs=_u8(mp.get("has_obs")), skip=_u8(mp.get("skip")))
    m = MapPoints()
    m.n = keep["world_pos"].shape[0]
    m.world_pos = _p(keep["world_pos"], c_float_p); m.normal = _p(keep["normal"], c_float_p); m.max_distance = _p(keep["max_distance"], c_float_p)
    m.min_distance = _p(keep["min_distance"], c_float_p); m.desc = _p(keep["desc"], c_uint32_p)
    m.has_obs = _p(keep["has_obs"], c_uint8_p); m.skip = _p(keep["skip"], c_uint8_p)
    return m, keep


def search_local_points(lib, ctx, F: Frame, view: FrameView, mp: dict, f_occupied=None, th=1.0, nnratio=0.8, viewing_cos_limit=0.5):
    """Tracking::SearchLocalPoints (src/Tracking.cc:1613-1664): Frame::isInFrustum for every local MapPoint and
    ORBmatcher::SearchByProjection(F, vpMapPoints, th), both on the device, one call.  Returns (SearchOutput, frustum dict)."""
    p = prepare(F, np.zeros((0, 8), np.uint32), candidates=CAND_GRID, accept_max=TH_HIGH, t_occupied=f_occupied)
    m, keep = map_points_struct(mp)
    n, nt = m.n, F.n
    out = SearchOutput(np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(nt, np.int32), 0, 0)
    r = OrbSearchResult()
    r.match = _p(out.match, c_int32_p); r.best_dist = _p(out.best_dist, c_int32_p); r.second_dist = _p(out.second_dist, c_int32_p)
    r.removed = _p(out.removed, c_uint8_p); r.owner = _p(out.owner, c_int32_p)
    fr = dict(in_view=np.zeros(n, np.uint8), proj_uvr=np.zeros((n, 3), np.float32), level=np.zeros(n, np.int32), view_cos=np.zeros(n, np.float32))
    fs = FrustumResult()
    fs.in_view = _p(fr["in_view"], c_uint8_p); fs.proj_uvr = _p(fr["proj_uvr"], c_float_p); fs.level = _p(fr["level"], c_int32_p)
    fs.view_cos = _p(fr["view_cos"], c_float_p)
    fn = lib.fn("orb_search_local_points")
    fn.argtypes = [C.c_void_p, C.POINTER(OrbSearch), C.POINTER(FrameView), C.POINTER(MapPoints), C.c_float, C.c_float, C.c_float,
                   C.POINTER(FrustumResult), C.POINTER(OrbSearchResult)]
    fn.restype = C.c_int
    st = fn(ctx, C.byref(p.s), C.byref(view), C.byref(m), float(np.float32(viewing_cos_limit)), float(np.float32(th)), float(np.float32(nnratio)),
            C.byref(fs), C.byref(r))
    if st != abi.LLD_OK:
        raise RuntimeError(f"lld_orb_search_local_points failed: {lib.fn('status_string')(st).decode()}")
    out.n_matches, out.rounds = r.n_matches, r.rounds
    return out, fr


def last_frame_struct(last: dict):
    keep = dict(world_pos=_f32(last["world_pos"]).reshape(-1, 3), valid=_u8(last["valid"]), octave=_i32(last["octave"]), angle=_f32(last.get("angle")),
                desc=np.ascontiguousarray(last["desc"], np.uint32).reshape(-1, 8), has_obs=_u8(last.get("has_obs")))
    m = LastFramePoints()
    m.n = keep["world_pos"].shape[0]
    m.world_pos = _p(keep["world_pos"], c_float_p); m.valid = _p(keep["valid"], c_uint8_p); m.octave = _p(keep["octave"], c_int32_p)
    m.angle = _p(keep["angle"], c_float_p); m.desc = _p(keep["desc"], c_uint32_p); m.has_obs = _p(keep["has_obs"], c_uint8_p)
    return m, keep


def search_last_frame(lib, ctx, Cur: Frame, view: FrameView, last: dict, cur_occupied=None, direction=0, th=7.0, check_orientation=True):
    """ORBmatcher::SearchByProjection(Current, Last, th, bMono) (src/ORBmatcher.cc:1328-1470) with the projection of the last
    frame's MapPoints done on the device.  Returns (SearchOutput, proj_uvr [n,3])."""
    p = prepare(Cur, np.zeros((0, 8), np.uint32), candidates=CAND_GRID, accept_max=TH_HIGH, t_occupied=cur_occupied)
    m, keep = last_frame_struct(last)
    n, nt = m.n, Cur.n
    out = SearchOutput(np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(nt, np.int32), 0, 0)
    r = OrbSearchResult()
    r.match = _p(out.match, c_int32_p); r.best_dist = _p(out.best_dist, c_int32_p); r.second_dist = _p(out.second_dist, c_int32_p)
    r.removed = _p(out.removed, c_uint8_p); r.owner = _p(out.owner, c_int32_p)
    uvr = np.zeros((n, 3), np.float32)
    fn = lib.fn("orb_search_last_frame")
    fn.argtypes = [C.c_void_p, C.POINTER(OrbSearch), C.POINTER(FrameView), C.POINTER(LastFramePoints), C.c_int, C.c_float, C.c_int, c_float_p,
                   C.POINTER(OrbSearchResult)]
    fn.restype = C.c_int
    st = fn(ctx, C.byref(p.s), C.byref(view), C.byref(m), int(direction), float(np.float32(th)), int(check_orientation), _p(uvr, c_float_p), C.byref(r))
    if st != abi.LLD_OK:
        raise RuntimeError(f"lld_orb_search_last_frame failed: {lib.fn('status_string')(st).decode()}")
    out.n_matches, out.rounds = r.n_matches, r.rounds
    return out, uvr


class ResidentFrame:
    """lld_frame_*: the keypoint side of a Frame uploaded once; search_last_frame / search_local_points as the module-level functions of
    the same names, on the resident copy (only the queries and the occupancy bytes travel per call)."""

    def __init__(self, lib, ctx, F: Frame):
        self.lib, self.ctx, self.F = lib, ctx, F
        p = prepare(F, np.zeros((0, 8), np.uint32), candidates=CAND_GRID, accept_max=TH_HIGH)
        h = C.c_void_p()
        fn = lib.fn("frame_create"); fn.argtypes = [C.c_void_p, C.POINTER(OrbSearch), C.POINTER(C.c_void_p)]; fn.restype = C.c_int
        st = fn(ctx, C.byref(p.s), C.byref(h))
        if st != abi.LLD_OK:
            raise RuntimeError(f"lld_frame_create failed: {lib.fn('status_string')(st).decode()}")
        self.handle = h

    def close(self):
        if self.handle:
            fn = self.lib.fn("frame_destroy"); fn.argtypes = [C.c_void_p]; fn.restype = None
            fn(self.handle); self.handle = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def _result(self, n):
        nt = self.F.n
        out = SearchOutput(np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(nt, np.int32), 0, 0)
        r = OrbSearchResult()
        r.match = _p(out.match, c_int32_p); r.best_dist = _p(out.best_dist, c_int32_p); r.second_dist = _p(out.second_dist, c_int32_p)
        r.removed = _p(out.removed, c_uint8_p); r.owner = _p(out.owner, c_int32_p)
        return out, r

    def search_last_frame(self, view: FrameView, last: dict, cur_occupied=None, direction=0, th=7.0, check_orientation=True):
        m, keep = last_frame_struct(last)
        out, r = self._result(m.n)
        uvr = np.zeros((m.n, 3), np.float32)
        occ = _u8(cur_occupied)
        fn = self.lib.fn("frame_search_last_frame")
        fn.argtypes = [C.c_void_p, c_uint8_p, C.POINTER(FrameView), C.POINTER(LastFramePoints), C.c_int, C.c_float, C.c_int, c_float_p, C.POINTER(OrbSearchResult)]
        fn.restype = C.c_int
        st = fn(self.handle, _p(occ, c_uint8_p), C.byref(view), C.byref(m), int(direction), float(np.float32(th)), int(check_orientation), _p(uvr, c_float_p), C.byref(r))
        if st != abi.LLD_OK:
            raise RuntimeError(f"lld_frame_search_last_frame failed: {self.lib.fn('status_string')(st).decode()}")
        out.n_matches, out.rounds = r.n_matches, r.rounds
        return out, uvr

    def search_local_points(self, view: FrameView, mp: dict, f_occupied=None, th=1.0, nnratio=0.8, viewing_cos_limit=0.5):
        m, keep = map_points_struct(mp)
        n = m.n
        out, r = self._result(n)
        fr = dict(in_view=np.zeros(n, np.uint8), proj_uvr=np.zeros((n, 3), np.float32), level=np.zeros(n, np.int32), view_cos=np.zeros(n, np.float32))
        fs = FrustumResult()
        fs.in_view = _p(fr["in_view"], c_uint8_p); fs.proj_uvr = _p(fr["proj_uvr"], c_float_p); fs.level = _p(fr["level"], c_int32_p)
        fs.view_cos = _p(fr["view_cos"], c_float_p)
        occ = _u8(f_occupied)
        fn = self.lib.fn("frame_search_local_points")
        fn.argtypes = [C.c_void_p, c_uint8_p, C.POINTER(FrameView), C.POINTER(MapPoints), C.c_float, C.c_float, C.c_float, C.POINTER(FrustumResult), C.POINTER(OrbSearchResult)]
        fn.restype = C.c_int
        st = fn(self.handle, _p(occ, c_uint8_p), C.byref(view), C.byref(m), float(np.float32(viewing_cos_limit)), float(np.float32(th)), float(np.float32(nnratio)),
                C.byref(fs), C.byref(r))
        if st != abi.LLD_OK:
            raise RuntimeError(f"lld_frame_search_local_points failed: {self.lib.fn('status_string')(st).decode()}")
        out.n_matches, out.rounds = r.n_matches, r.rounds
        return out, fr


def fuse_search_points(lib, ctx, KF: Frame, view: FrameView, mp: dict, th=3.0):
    """ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:825-958) with the projection loop on the device too.
    Returns (SearchOutput, proj_uvr [n,3]); match[i] = bestIdx or -1, n_matches = nFused."""
    p = prepare(KF, np.zeros((0, 8), np.uint32), candidates=CAND_GRID, accept_max=TH_LOW)
    m, keep = map_points_struct(mp)
    n, nt = m.n, KF.n
    out = SearchOutput(np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(nt, np.int32), 0, 0)
    r = OrbSearchResult()
    r.match = _p(out.match, c_int32_p); r.best_dist = _p(out.best_dist, c_int32_p); r.second_dist = _p(out.second_dist, c_int32_p)
    r.removed = _p(out.removed, c_uint8_p); r.owner = _p(out.owner, c_int32_p)
    uvr = np.zeros((n, 3), np.float32)
    fn = lib.fn("orb_fuse_search")
    fn.argtypes = [C.c_void_p, C.POINTER(OrbSearch), C.POINTER(FrameView), C.POINTER(MapPoints), C.c_float, c_float_p, C.POINTER(OrbSearchResult)]
    fn.restype = C.c_int
    st = fn(ctx, C.byref(p.s), C.byref(view), C.byref(m), float(np.float32(th)), _p(uvr, c_float_p), C.byref(r))
    if st != abi.LLD_OK:
        raise RuntimeError(f"lld_orb_fuse_search failed: {lib.fn('status_string')(st).decode()}")
    out.n_matches, out.rounds = r.n_matches, r.rounds
    return out, uvr


# ---------------------------------------------------------------------- relocalisation / loop closing with the projection on the device
PROJ_KF_SIM3, PROJ_RELOC, PROJ_FUSE_SIM3, PROJ_SIM3_DIR = 0, 1, 2, 3


class OrbProjection(C.Structure):
    _fields_ = [("routine", C.c_int32), ("th", C.c_float), ("accept_max", C.c_int32), ("check_orientation", C.c_int32),
                ("sR", C.c_float * 9), ("t", C.c_float * 3)]


def sim3_view(Scw, cam, F: Frame, scale_factor=1.2) -> FrameView:
    """The decomposition at the head of SearchByProjection(KeyFrame*, Scw, ...) and Fuse(KeyFrame*, Scw, ...)
    (src/ORBmatcher.cc:298-303, :984-989) from a float32 4x4 Scw: scw = sqrt(sRcw.row(0).dot(sRcw.row(0))) (dot in double, sqrt of
    the double, rounded to float), Rcw = sRcw/scw and tcw = Scw.col(3)/scw (cv::divide by a double scalar, rounded to float),
    Ow = -Rcw.t()*tcw (one gemm).  These are the adapter's OpenCV calls; what the device needs is their result."""
    S = np.asarray(Scw, np.float32).reshape(4, 4)
    sR = S[:3, :3]
    scw = np.float32(np.sqrt(np.dot(sR[0].astype(np.float64), sR[0].astype(np.float64))))
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = (sR.astype(np.float64) / np.float64(scw)).astype(np.float32)
    T[:3, 3] = (S[:3, 3].astype(np.float64) / np.float64(scw)).astype(np.float32)
    return frame_view(T, cam, F, scale_factor)


def search_projected(lib, ctx, F: Frame, view: FrameView, mp: dict, routine: int, th, accept_max=0, check_orientation=False, angle=None,
                     occupied=None, sR=None, t=None):
    """lld_orb_search_projected: the matchers of relocalisation and loop closing (src/ORBmatcher.cc:290-403, :977-1100, :1147-1304,
    :1472-1599) with their projection loops on the device.  Returns (SearchOutput, uv [n,2], level [n])."""
    p = prepare(F, np.zeros((0, 8), np.uint32), candidates=CAND_GRID, accept_max=TH_HIGH, t_occupied=occupied,
                check_orientation=bool(check_orientation) and routine == PROJ_RELOC)
    m, keep = map_points_struct(mp)
    n, nt = m.n, F.n
    out = SearchOutput(np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint8), np.empty(nt, np.int32), 0, 0)
    r = OrbSearchResult()
    r.match = _p(out.match, c_int32_p); r.best_dist = _p(out.best_dist, c_int32_p); r.second_dist = _p(out.second_dist, c_int32_p)
    r.removed = _p(out.removed, c_uint8_p); r.owner = _p(out.owner, c_int32_p)
    pr = OrbProjection()
    pr.routine, pr.th, pr.accept_max, pr.check_orientation = int(routine), float(np.float32(th)), int(accept_max), int(bool(check_orientation))
    if sR is not None:
        for i, x in enumerate(np.asarray(sR, np.float32).reshape(9)): pr.sR[i] = float(x)
        for i, x in enumerate(np.asarray(t, np.float32).reshape(3)): pr.t[i] = float(x)
    ang = _f32(angle)
    uv = np.zeros((n, 2), np.float32); lvl = np.zeros(n, np.int32)
    fn = lib.fn("orb_search_projected")
    fn.argtypes = [C.c_void_p, C.POINTER(OrbSearch), C.POINTER(FrameView), C.POINTER(MapPoints), c_float_p, C.POINTER(OrbProjection), c_float_p,
                   c_int32_p, C.POINTER(OrbSearchResult)]
    fn.restype = C.c_int
    st = fn(ctx, C.byref(p.s), C.byref(view), C.byref(m), _p(ang, c_float_p), C.byref(pr), _p(uv, c_float_p), _p(lvl, c_int32_p), C.byref(r))
    if st != abi.LLD_OK:
        raise RuntimeError(f"lld_orb_search_projected failed: {lib.fn('status_string')(st).decode()}")
    out.n_matches, out.rounds = r.n_matches, r.rounds
    return out, uv, lvl


def sim3_transforms(s12, R12, t12):
    """sR12 = s12*R12, sR21 = (1.0/s12)*R12.t(), t21 = -sR21*t12 (src/ORBmatcher.cc:1121-1124) as OpenCV evaluates them on CV_32F
    matrices: scalar factors are doubles applied element-wise with one rounding, the product is one gemm."""
    R12 = np.asarray(R12, np.float32).reshape(3, 3); t12 = np.asarray(t12, np.float32).reshape(3)
    s12 = np.float32(s12)
    sR12 = (np.float64(s12) * R12.astype(np.float64)).astype(np.float32)
    sR21 = ((1.0 / np.float64(s12)) * R12.T.astype(np.float64)).astype(np.float32)
    t21 = (-(sR21.astype(np.float64) @ t12.astype(np.float64))).astype(np.float32)
    return sR12, t12, sR21, t21


def search_by_sim3_points(lib, ctx, KF1: Frame, view1: FrameView, mp1: dict, KF2: Frame, view2: FrameView, mp2: dict, sR12, t12, sR21, t21, th=7.5):
    """lld_orb_search_by_sim3: ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1102-1326) in one call.  Returns (match12 [N1], nFound)."""
    p1 = prepare(KF1, np.zeros((0, 8), np.uint32), candidates=CAND_GRID, accept_max=TH_HIGH)
    p2 = prepare(KF2, np.zeros((0, 8), np.uint32), candidates=CAND_GRID, accept_max=TH_HIGH)
    m1, keep1 = map_points_struct(mp1)
    m2, keep2 = map_points_struct(mp2)
    arr = [np.ascontiguousarray(a, np.float32).reshape(-1) for a in (sR12, t12, sR21, t21)]
    match12 = np.full(m1.n, -1, np.int32); nf = C.c_int32(0)
    fn = lib.fn("orb_search_by_sim3")
    fn.argtypes = [C.c_void_p, C.POINTER(OrbSearch), C.POINTER(FrameView), C.POINTER(MapPoints), C.POINTER(OrbSearch), C.POINTER(FrameView),
                   C.POINTER(MapPoints), c_float_p, c_float_p, c_float_p, c_float_p, C.c_float, c_int32_p, C.POINTER(C.c_int32)]
    fn.restype = C.c_int
    st = fn(ctx, C.byref(p1.s), C.byref(view1), C.byref(m1), C.byref(p2.s), C.byref(view2), C.byref(m2), _p(arr[0], c_float_p), _p(arr[1], c_float_p),
            _p(arr[2], c_float_p), _p(arr[3], c_float_p), float(np.float32(th)), _p(match12, c_int32_p), C.byref(nf))
    if st != abi.LLD_OK:
        raise RuntimeError(f"lld_orb_search_by_sim3 failed: {lib.fn('status_string')(st).decode()}")
    return match12, int(nf.value)

