"""GPU parity: the guided ORB searches (one lld_orb_search_run call per reference routine) vs the literal sequential CPU
restatements in oracle/lldo_orbsearch.cpp.  Everything is integer / index work: bit-exact."""
import numpy as np
import pytest

import oracle_orbsearch as OS
from lld_slam_amd import ORBmatcher, orb_search, synth
from lld_slam_amd.orb_search import Frame

pytestmark = pytest.mark.gpu
f32 = np.float32


def expect_slots(out, occupied, token=1 << 20):
    """Frame slot vector implied by the device's `owner`: untouched slots keep their input, -2 = NULLed by the rotation filter."""
    slot = np.where(np.asarray(occupied) != 0, token, -1).astype(np.int32)
    slot = np.where(out.owner >= 0, out.owner, slot)
    return np.where(out.owner == -2, -1, slot).astype(np.int32)


@pytest.mark.parametrize("seed,n,nq,th", [(0, 2000, 1500, 1.0), (1, 2000, 2000, 3.0), (2, 300, 700, 1.0), (3, 4096, 3000, 1.0)])
def test_search_by_projection_local_map(gpu_ctx, seed, n, nq, th):
    F = synth.make_orb_frame(seed, n)
    q = synth.make_projection_queries(F, seed, nq, dup_frac=0.3)
    out = ORBmatcher(gpu_ctx, 0.8).SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], th)
    n_exp, slot = OS.search_by_projection_map(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], th, 0.8)
    assert out.n_matches == n_exp and n_exp > nq // 5
    np.testing.assert_array_equal(expect_slots(out, q["occupied"]), slot)
    assert out.rounds >= 2                                            # competing queries: the fixed point needed more than one round


@pytest.mark.parametrize("seed,direction,check", [(0, 0, True), (1, 1, True), (2, -1, True), (3, 0, False)])
def test_search_by_projection_last_frame(gpu_ctx, seed, direction, check):
    F = synth.make_orb_frame(30 + seed, 2000)
    q = synth.make_projection_queries(F, 30 + seed, 1800, dup_frac=0.3)
    q["obs"][::4] = 0                                                 # temporal points: they do not block later queries
    out = ORBmatcher(gpu_ctx, 0.9, check).SearchByProjectionFrame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"],
                                                                    q["occupied"], direction, 7.0)
    n_exp, slot = OS.search_by_projection_frame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"],
                                                direction, 7.0, check)
    assert out.n_matches == n_exp and n_exp > 300
    np.testing.assert_array_equal(expect_slots(out, q["occupied"]), slot)
    if check:
        assert out.removed.sum() > 0                                  # the rotation histogram did drop matches


@pytest.mark.parametrize("seed,th,orbdist", [(0, 10.0, 100), (1, 3.0, 64)])
def test_search_by_projection_relocalisation(gpu_ctx, seed, th, orbdist):
    F = synth.make_orb_frame(40 + seed, 2000)
    q = synth.make_projection_queries(F, 40 + seed, 1200, dup_frac=0.3)
    out = ORBmatcher(gpu_ctx, 0.9, True).SearchByProjectionReloc(F, q["desc"], q["valid"], q["uv"], q["level"], q["angle"], q["occupied"], th, orbdist)
    n_exp, slot = OS.search_by_projection_reloc(F, q["desc"], q["valid"], q["uv"], q["level"], q["angle"], q["occupied"], th, orbdist, True)
    assert out.n_matches == n_exp and n_exp > 200
    np.testing.assert_array_equal(expect_slots(out, q["occupied"]), slot)


def test_search_by_projection_keyframe_sim3(gpu_ctx):
    F = synth.make_orb_frame(50, 2000)
    q = synth.make_projection_queries(F, 50, 1500, dup_frac=0.3)
    out = ORBmatcher(gpu_ctx, 0.75).SearchByProjectionKF(F, q["desc"], q["valid"], q["uv"], q["level"], q["occupied"], 10)
    n_exp, slot = OS.search_by_projection_kf(F, q["desc"], q["valid"], q["uv"], q["level"], q["occupied"], 10)
    assert out.n_matches == n_exp and n_exp > 200
    np.testing.assert_array_equal(expect_slots(out, q["occupied"]), slot)


@pytest.mark.parametrize("seed", [0, 1])
def test_fuse_inner_search(gpu_ctx, seed):
    F = synth.make_orb_frame(60 + seed, 2000)
    q = synth.make_projection_queries(F, 60 + seed, 1500, pos_sigma=1.2)
    out = ORBmatcher(gpu_ctx).Fuse(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], 3.0)
    n_exp, best = OS.fuse_search(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], 3.0)
    assert out.n_matches == n_exp and n_exp > 100 and out.rounds == 1
    np.testing.assert_array_equal(out.match, best)


def test_search_by_sim3_mutual_agreement(gpu_ctx):
    F1, F2, _ = synth.make_bow_pair(3, 1500)
    q1 = synth.make_projection_queries(F2, 70, F1.n); q2 = synth.make_projection_queries(F1, 71, F2.n)
    a = dict(desc=q1["desc"], valid=q1["valid"], uv=q1["uv"], pred_level=q1["level"])
    b = dict(desc=q2["desc"], valid=q2["valid"], uv=q2["uv"], pred_level=q2["level"])
    # make the two directions agree for a subset: the queries of direction 2 that come from keypoint k of F1 search F1 and
    # must find k; craft them as exact copies
    m12, found = ORBmatcher(gpu_ctx).SearchBySim3(F1, F2, a, b, 7.5)
    m1 = OS.search_sim3_direction(F2, a["desc"], a["valid"], a["uv"], a["pred_level"], 7.5)
    m2 = OS.search_sim3_direction(F1, b["desc"], b["valid"], b["uv"], b["pred_level"], 7.5)
    exp = np.array([m1[i] if m1[i] >= 0 and m2[m1[i]] == i else -1 for i in range(F1.n)], np.int32)
    np.testing.assert_array_equal(m12, exp)
    assert found == int((exp >= 0).sum())


@pytest.mark.parametrize("seed,check", [(0, True), (1, False)])
def test_search_by_bow_keyframe_to_frame(gpu_ctx, seed, check):
    F1, F2, nd = synth.make_bow_pair(seed, 2000)
    valid = (np.random.default_rng(seed).random(F1.n) < 0.85).astype(np.uint8)
    out = ORBmatcher(gpu_ctx, 0.7, check).SearchByBoWFrame(F1, F2, nd, valid)
    n_exp, fm = OS.search_by_bow_frame(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], valid, 0.7, check)
    assert out.n_matches == n_exp and n_exp > 400
    got = np.where(out.owner >= 0, out.query_kp[np.maximum(out.owner, 0)], -1)
    np.testing.assert_array_equal(got, fm)


@pytest.mark.parametrize("seed", [0, 1])
def test_search_by_bow_keyframe_to_keyframe(gpu_ctx, seed):
    F1, F2, nd = synth.make_bow_pair(10 + seed, 2000)
    rng = np.random.default_rng(seed)
    v1 = (rng.random(F1.n) < 0.85).astype(np.uint8); v2 = (rng.random(F2.n) < 0.85).astype(np.uint8)
    out = ORBmatcher(gpu_ctx, 0.75, True).SearchByBoWKF(F1, F2, nd, v1, v2)
    n_exp, m12 = OS.search_by_bow_kf(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], v1, v2, 0.75, True)
    assert out.n_matches == n_exp and n_exp > 300
    got = -np.ones(F1.n, np.int32); got[out.query_kp] = out.final_match()
    np.testing.assert_array_equal(got, m12)


@pytest.mark.parametrize("seed,only_stereo", [(0, False), (1, True)])
def test_search_for_triangulation(gpu_ctx, seed, only_stereo):
    F1, F2, nd = synth.make_bow_pair(20 + seed, 2000, pos_sigma=(25.0, 1.5))
    rng = np.random.default_rng(seed)
    # nearly pure x-translation between the two keyframes: epipolar lines are (almost) the image rows
    F12 = (np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0.0]]) + rng.normal(0, 2e-6, (3, 3))).astype(f32)
    has1 = (rng.random(F1.n) < 0.3).astype(np.uint8); has2 = (rng.random(F2.n) < 0.3).astype(np.uint8)
    if not only_stereo:
        F1.uright[::2] = -1; F2.uright[::3] = -1                    # mono keypoints: the epipole-distance rule applies
    epi = OS.epipolar_lines(F12, F1.xy)
    epipole = (620.0, 180.0)
    out = ORBmatcher(gpu_ctx, 0.6, True).SearchForTriangulation(F1, F2, nd, has1, has2, epi, epipole, only_stereo)
    n_exp, m12 = OS.search_for_triangulation(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], has1, has2, F12, epipole,
                                             only_stereo, True)
    got = -np.ones(F1.n, np.int32); got[out.query_kp] = out.final_match()
    np.testing.assert_array_equal(got, m12)
    assert out.n_matches == n_exp and n_exp > 150


def test_triangulation_tie_goes_to_the_later_candidate(gpu_ctx):
    d = np.zeros((1, 8), np.uint32); t = np.zeros((3, 8), np.uint32); t[:, 1] = 1
    KF1 = Frame(desc=d, xy=np.array([[600, 180]], f32), octave=np.zeros(1, np.int32), uright=np.array([550], f32), angle=np.zeros(1, f32))
    KF2 = Frame(desc=t, xy=np.array([[500, 180], [520, 180], [540, 180]], f32), octave=np.zeros(3, np.int32), uright=np.array([450, 470, 490], f32),
                angle=np.zeros(3, f32))
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], f32)
    nd = dict(n_nodes=1, start1=[0, 1], idx1=[0], start2=[0, 3], idx2=[0, 1, 2])
    out = ORBmatcher(gpu_ctx, 0.6, False).SearchForTriangulation(KF1, KF2, nd, [0], [0, 0, 0], OS.epipolar_lines(F12, KF1.xy), (1e6, 180.0))
    assert out.match.tolist() == [2]


@pytest.mark.parametrize("seed,n", [(0, 2000), (1, 700)])
def test_stereo_matches_hamming_search(gpu_ctx, seed, n):
    L, R = synth.make_stereo_pair(seed, n)
    out = ORBmatcher(gpu_ctx).ComputeStereoMatches(L, R, 0.0, 100.0)
    br, bd = OS.stereo_search(L, R, 376, 0.0, 100.0)
    np.testing.assert_array_equal(out.match, br)
    m = br >= 0
    np.testing.assert_array_equal(out.best_dist[m], bd[m])
    assert m.sum() > n // 5


def test_batch_of_mixed_problems_equals_single_calls(gpu_ctx):
    """lld_orb_search_batch: different routines, sizes and LDS footprints in one launch; every problem equals its own oracle."""
    S = orb_search
    prepared, expect = [], []
    for i in range(6):
        F = synth.make_orb_frame(100 + i, [2000, 300, 4096, 1200, 2000, 64][i])
        q = synth.make_projection_queries(F, 100 + i, [1500, 700, 2500, 64, 2000, 300][i], dup_frac=0.3)
        prepared.append(S.search_by_projection_map(None, None, F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0, 0.8))
        expect.append(("slots", q["occupied"]) + OS.search_by_projection_map(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0, 0.8))
        prepared.append(S.search_by_projection_frame(None, None, F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], 0, 7.0, True))
        expect.append(("slots", q["occupied"]) + OS.search_by_projection_frame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], 0, 7.0, True))
    L, R = synth.make_stereo_pair(5, 1500)
    prepared.append(S.stereo_search(None, None, L, R, 0.0, 100.0)); expect.append(("stereo",) + OS.stereo_search(L, R, 376, 0.0, 100.0))
    F1, F2, nd = synth.make_bow_pair(5, 1800); v = np.ones(F1.n, np.uint8)
    prepared.append(S.search_by_bow_frame(None, None, F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], v, 0.7, True))
    expect.append(("bow",) + OS.search_by_bow_frame(F1, F2, nd["n_nodes"], nd["start1"], nd["idx1"], nd["start2"], nd["idx2"], v, 0.7, True))
    outs = S.run_batch(gpu_ctx.lib, gpu_ctx.handle, prepared)
    for out, e in zip(outs, expect):
        if e[0] == "slots":
            assert out.n_matches == e[2]; np.testing.assert_array_equal(expect_slots(out, e[1]), e[3])
        elif e[0] == "stereo":
            np.testing.assert_array_equal(out.match, e[1])
        else:
            assert out.n_matches == e[1]
            np.testing.assert_array_equal(np.where(out.owner >= 0, out.query_kp[np.maximum(out.owner, 0)], -1), e[2])


def test_long_dependency_chain_still_reaches_the_sequential_answer(gpu_ctx):
    """Worst case for the fixed-point rounds: every query wants the same few keypoints, so query i's answer depends on all
    earlier ones."""
    F = synth.make_orb_frame(90, 64, n_clusters=0)
    F.xy[:] = np.array([600.0, 180.0], f32) + np.random.default_rng(0).integers(-3, 4, (64, 2)).astype(f32)
    F.octave[:] = 0; F.uright[:] = -1
    base = F.desc[0].copy()
    for k in range(64):
        F.desc[k] = base; F.desc[k, 7] ^= np.uint32((1 << (k % 20)) - 1)          # distance k%20 from the base
    nq = 200
    q = dict(desc=np.repeat(base[None], nq, 0), valid=np.ones(nq, np.uint8), uv=np.tile(np.array([[600.0, 180.0]], f32), (nq, 1)),
             ur=np.zeros(nq, f32), level=np.zeros(nq, np.int32), view_cos=np.ones(nq, f32), obs=np.ones(nq, np.uint8), occupied=np.zeros(64, np.uint8))
    out = ORBmatcher(gpu_ctx, 1.0).SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 3.0)
    n_exp, slot = OS.search_by_projection_map(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 3.0, 1.0)
    assert out.n_matches == n_exp == 64 and out.rounds > 30
    np.testing.assert_array_equal(expect_slots(out, q["occupied"]), slot)


def test_edge_cases(gpu_ctx):
    F = synth.make_orb_frame(95, 100)
    q = synth.make_projection_queries(F, 95, 50)
    m = ORBmatcher(gpu_ctx)
    # no queries
    out = m.SearchByProjectionMap(F, np.zeros((0, 8), np.uint32), [], np.zeros((0, 2), f32), [], np.zeros(0, np.int32), [], [], q["occupied"])
    assert out.n_matches == 0 and out.match.shape == (0,) and (out.owner == -1).all()
    # no keypoints
    E = Frame(desc=np.zeros((0, 8), np.uint32), xy=np.zeros((0, 2), f32), octave=np.zeros(0, np.int32), uright=np.zeros(0, f32), angle=np.zeros(0, f32))
    out = m.SearchByProjectionMap(E, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], np.zeros(0, np.uint8))
    assert out.n_matches == 0 and (out.match == -1).all() and (out.best_dist == 256).all()
    # everything occupied / nothing valid
    out = m.SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], np.ones(F.n, np.uint8))
    assert out.n_matches == 0
    out = m.SearchByProjectionMap(F, q["desc"], np.zeros(50, np.uint8), q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"])
    assert out.n_matches == 0
    # more keypoints than the LDS-resident limit: refused loudly, not truncated
    big = synth.make_orb_frame(96, orb_search.MAX_KEYPOINTS + 1, n_clusters=0)
    with pytest.raises(RuntimeError, match="supported limits"):
        m.SearchByProjectionMap(big, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], np.zeros(big.n, np.uint8))


@pytest.mark.parametrize("seed,th", [(0, 1.0), (1, 3.0), (2, 5.0)])
def test_search_local_points_frustum_and_search_on_device(gpu_ctx, seed, th):
    """Tracking::SearchLocalPoints: Frame::isInFrustum on the device feeds the projection search without a host round trip; the
    frustum outputs equal the CPU restatement bit for bit and the matches equal the sequential search on them."""
    F = synth.make_orb_frame(120 + seed, 2000)
    T, mp = synth.make_local_map(F, 120 + seed, 2500)
    view = orb_search.frame_view(T, synth.KITTI_CAM, F)
    out, fr = orb_search.search_local_points(gpu_ctx.lib, gpu_ctx.handle, F, view, mp, mp["occupied"], th, 0.8)
    k, inv, uvr, lvl, vc = OS.is_in_frustum(view, mp)
    np.testing.assert_array_equal(fr["in_view"], inv)
    m = inv != 0
    assert 500 < k < 2400 and (~m).sum() > 100
    np.testing.assert_array_equal(fr["proj_uvr"][m], uvr[m])
    np.testing.assert_array_equal(fr["view_cos"][m], vc[m])
    np.testing.assert_array_equal(fr["level"][m], lvl[m])
    n_exp, slot = OS.search_by_projection_map(F, mp["desc"], inv, uvr[:, :2], uvr[:, 2], lvl, vc, mp["has_obs"], mp["occupied"], th, 0.8)
    assert out.n_matches == n_exp and n_exp > 150
    np.testing.assert_array_equal(expect_slots(out, mp["occupied"]), slot)


def test_predict_scale_at_its_ceil_boundaries(gpu_ctx):
    """MapPoint::PredictScale = ceil(log(maxDistance / dist) / logScaleFactor): 400 float neighbours on either side of every 1.2^k, k = 1..7,
    as distance ratios.  The level flips where the quotient crosses an integer, and it flips at the same float on the device and on the
    host only if both compute the same logf: the device carries glibc's algorithm (lld_orb_search.hip glibc_logf; the device library's
    own logf differs from it by an ulp often enough to move one level in ~1e8 - tools/fuzz_matchers.py, FUZZ_BIG=1, seed 9)."""
    F = synth.make_orb_frame(160, 600)
    T, mp = synth.make_local_map(F, 160, 9000, related_frac=1.0)
    view = orb_search.frame_view(T, synth.KITTI_CAM, F)
    _, inv0, _, _, _ = OS.is_in_frustum(view, mp)
    idx = np.nonzero(inv0)[0]
    PO = mp["world_pos"][idx] - np.array(view.Ow[:], f32)                       # float subtraction, double norm, float: as Frame::isInFrustum
    dist = np.sqrt((PO.astype(np.float64) ** 2).sum(1)).astype(f32)
    lsf = np.float64(f32(view.log_scale_factor))
    ratios = []
    for k in range(1, 8):
        c = f32(np.exp(k * lsf))
        ratios.append((np.array([c]).view(np.uint32)[0] + np.arange(-400, 401)).astype(np.uint32).view(f32))
    ratios = np.concatenate(ratios)
    assert idx.size >= ratios.size
    idx = idx[:ratios.size]; dist = dist[:ratios.size]
    mp["max_distance"][idx] = ratios * dist                                        # float product: the ratio the routine forms is within an ulp of the target
    mp["min_distance"][idx] = mp["max_distance"][idx] / f32(8.0)
    out, fr = orb_search.search_local_points(gpu_ctx.lib, gpu_ctx.handle, F, view, mp, mp["occupied"], 1.0, 0.8)
    _, inv, uvr, lvl, vc = OS.is_in_frustum(view, mp)
    np.testing.assert_array_equal(fr["in_view"], inv)
    m = inv != 0
    assert m[idx].sum() > 0.9 * idx.size
    np.testing.assert_array_equal(fr["level"][m], lvl[m])
    got = np.unique(lvl[idx][m[idx]])
    assert got.min() <= 1 and got.max() >= 7                                       # both sides of every boundary were reached


def test_search_local_points_edge_cases(gpu_ctx):
    F = synth.make_orb_frame(130, 300)
    T, mp = synth.make_local_map(F, 130, 200)
    view = orb_search.frame_view(T, synth.KITTI_CAM, F)
    empty = {k: (v[:0] if k != "occupied" else v) for k, v in mp.items()}
    out, fr = orb_search.search_local_points(gpu_ctx.lib, gpu_ctx.handle, F, view, empty, mp["occupied"])
    assert out.n_matches == 0 and out.match.shape == (0,)
    allskip = dict(mp, skip=np.ones(200, np.uint8))
    out, fr = orb_search.search_local_points(gpu_ctx.lib, gpu_ctx.handle, F, view, allskip, mp["occupied"])
    assert out.n_matches == 0 and not fr["in_view"].any()


@pytest.mark.parametrize("seed,direction,th", [(0, 0, 7.0), (1, 1, 15.0), (2, -1, 7.0)])
def test_search_last_frame_projection_and_search_on_device(gpu_ctx, seed, direction, th):
    """Tracking::TrackWithMotionModel's matcher: the last frame's MapPoints are projected on the device (cv::gemm transform,
    invzc = float(1.0/z), bounds), then searched with occupancy and the rotation histogram."""
    F = synth.make_orb_frame(140 + seed, 2000)
    T, mp = synth.make_local_map(F, 140 + seed, 2000)
    rng = np.random.default_rng(seed)
    ang = np.mod(F.angle[mp["src"]] + 25.0 + rng.normal(0, 6.0, 2000), 360.0)
    wild = rng.random(2000) < 0.15; ang[wild] = rng.uniform(0, 360, int(wild.sum()))
    last = dict(world_pos=mp["world_pos"], valid=(rng.random(2000) < 0.85).astype(np.uint8), octave=F.octave[mp["src"]],
                angle=ang.astype(np.float32), desc=mp["desc"], has_obs=mp["has_obs"])
    view = orb_search.frame_view(T, synth.KITTI_CAM, F)
    out, uvr = orb_search.search_last_frame(gpu_ctx.lib, gpu_ctx.handle, F, view, last, mp["occupied"], direction, th, True)
    valid, uv, ur = OS.project_last_frame(view, last)
    m = valid != 0
    assert 800 < m.sum() < 1900
    np.testing.assert_array_equal(uvr[m, :2], uv[m]); np.testing.assert_array_equal(uvr[m, 2], ur[m])
    n_exp, slot = OS.search_by_projection_frame(F, last["desc"], valid, uv, ur, last["octave"], last["angle"], last["has_obs"], mp["occupied"],
                                                direction, th, True)
    assert out.n_matches == n_exp and n_exp > 100
    np.testing.assert_array_equal(expect_slots(out, mp["occupied"]), slot)


@pytest.mark.parametrize("seed,th", [(0, 3.0), (1, 4.0)])
def test_fuse_projection_and_search_on_device(gpu_ctx, seed, th):
    """LocalMapping::SearchInNeighbors' matcher: Fuse's projection loop (IsInImage with strict upper bounds, 60-degree viewing
    test, PredictScale) on the device, then the window search with the reprojection-chi2 gate."""
    KF = synth.make_orb_frame(150 + seed, 2000)
    T, mp = synth.make_local_map(KF, 150 + seed, 2500)
    view = orb_search.frame_view(T, synth.KITTI_CAM, KF)
    out, uvr = orb_search.fuse_search_points(gpu_ctx.lib, gpu_ctx.handle, KF, view, mp, th)
    valid, uv, ur, lvl = OS.project_fuse(view, mp)
    m = valid != 0
    assert 800 < m.sum() < 2400
    np.testing.assert_array_equal(uvr[m, :2], uv[m]); np.testing.assert_array_equal(uvr[m, 2], ur[m])
    n_exp, best = OS.fuse_search(KF, mp["desc"], valid, uv, ur, lvl, th)
    assert out.n_matches == n_exp and n_exp > 100
    np.testing.assert_array_equal(out.match, best)


# ---------------------------------------------------------------------- relocalisation / loop-closing matchers, projection on the device
def _sim3_of(T, s):
    """Scw whose decomposition (src/ORBmatcher.cc:298-303) gives back T's rotation and translation up to float rounding."""
    S = np.array(T, np.float32, copy=True)
    S[:3, :] = (np.float64(s) * T[:3, :].astype(np.float64)).astype(np.float32)
    return S


@pytest.mark.parametrize("seed,th,orbdist", [(0, 10.0, 100), (1, 3.0, 64)])
def test_relocalisation_projection_and_search_on_device(gpu_ctx, seed, th, orbdist):
    """Tracking::Relocalization's matcher: no depth test, invzc = float(1.0/z), inclusive frame bounds, three octaves, occupancy
    by CurrentFrame.mvpMapPoints and the rotation histogram over the keyframe's keypoint angles."""
    F = synth.make_orb_frame(160 + seed, 2000)
    T, mp = synth.make_local_map(F, 160 + seed, 2200)
    rng = np.random.default_rng(160 + seed)
    ang = np.mod(F.angle[mp["src"]] + 40.0 + rng.normal(0, 6.0, 2200), 360.0)
    wild = rng.random(2200) < 0.15; ang[wild] = rng.uniform(0, 360, int(wild.sum()))
    ang = ang.astype(np.float32)
    view = orb_search.frame_view(T, synth.KITTI_CAM, F)
    m = ORBmatcher(gpu_ctx, 0.9, True)
    out, uv, lvl = m.SearchByProjectionRelocPoints(F, view, mp, ang, mp["occupied"], th, orbdist)
    valid, uv_o, lvl_o = OS.project_general(view, mp, orb_search.PROJ_RELOC)
    k = valid != 0
    assert 800 < k.sum() < 2100
    np.testing.assert_array_equal(uv[k], uv_o[k]); np.testing.assert_array_equal(lvl[k], lvl_o[k])
    n_exp, slot = OS.search_by_projection_reloc(F, mp["desc"], valid, uv_o, lvl_o, ang, mp["occupied"], th, orbdist, True)
    assert out.n_matches == n_exp and n_exp > 150 and out.removed.sum() > 0
    np.testing.assert_array_equal(expect_slots(out, mp["occupied"]), slot)
    # without the normals (the routine never reads them) and without the orientation check
    bare = {k2: v for k2, v in mp.items() if k2 != "normal"}
    out2, _, _ = ORBmatcher(gpu_ctx, 0.9, False).SearchByProjectionRelocPoints(F, view, bare, None, mp["occupied"], th, orbdist)
    n2, slot2 = OS.search_by_projection_reloc(F, mp["desc"], valid, uv_o, lvl_o, ang, mp["occupied"], th, orbdist, False)
    assert out2.n_matches == n2
    np.testing.assert_array_equal(expect_slots(out2, mp["occupied"]), slot2)


@pytest.mark.parametrize("seed,scale,th", [(0, 1.0, 10), (1, 1.37, 10), (2, 0.61, 6)])
def test_keyframe_sim3_projection_and_search_on_device(gpu_ctx, seed, scale, th):
    """LoopClosing::ComputeSim3's matcher: SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)."""
    KF = synth.make_orb_frame(170 + seed, 2000)
    T, mp = synth.make_local_map(KF, 170 + seed, 2500)
    view = orb_search.sim3_view(_sim3_of(T, scale), synth.KITTI_CAM, KF)
    out, uv, lvl = ORBmatcher(gpu_ctx, 0.75).SearchByProjectionKFPoints(KF, view, mp, mp["occupied"], th)
    valid, uv_o, lvl_o = OS.project_general(view, mp, orb_search.PROJ_KF_SIM3)
    k = valid != 0
    assert 800 < k.sum() < 2400
    np.testing.assert_array_equal(uv[k], uv_o[k]); np.testing.assert_array_equal(lvl[k], lvl_o[k])
    n_exp, slot = OS.search_by_projection_kf(KF, mp["desc"], valid, uv_o, lvl_o, mp["occupied"], th)
    assert out.n_matches == n_exp and n_exp > 150
    np.testing.assert_array_equal(expect_slots(out, mp["occupied"]), slot)


@pytest.mark.parametrize("seed,scale,th", [(0, 1.0, 4.0), (1, 0.83, 3.0)])
def test_fuse_sim3_projection_and_search_on_device(gpu_ctx, seed, scale, th):
    """LoopClosing::SearchAndFuse's matcher: Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) - no chi2 gate, no occupancy."""
    KF = synth.make_orb_frame(180 + seed, 2000)
    T, mp = synth.make_local_map(KF, 180 + seed, 2500)
    view = orb_search.sim3_view(_sim3_of(T, scale), synth.KITTI_CAM, KF)
    out, uv, lvl = ORBmatcher(gpu_ctx).FuseSim3Points(KF, view, mp, th)
    valid, uv_o, lvl_o = OS.project_general(view, mp, orb_search.PROJ_FUSE_SIM3)
    k = valid != 0
    np.testing.assert_array_equal(uv[k], uv_o[k]); np.testing.assert_array_equal(lvl[k], lvl_o[k])
    n_exp, best = OS.fuse_search_sim3(KF, mp["desc"], valid, uv_o, lvl_o, th)
    assert out.n_matches == n_exp and n_exp > 100 and out.rounds == 1
    np.testing.assert_array_equal(out.match, best)


def _sim3_pair(seed, s12):
    """Two keyframes with MapPoints per keypoint: KF1's points project onto KF2's keypoints through S21 and vice versa; the entries
    are ordered so that a good share of the two directions agrees (src/ORBmatcher.cc:1306-1322)."""
    KF1 = synth.make_orb_frame(190 + seed, 1800); KF2 = synth.make_orb_frame(195 + seed, 1700)
    T2, mp1 = synth.make_local_map(KF2, 190 + seed, KF1.n)                  # KF1's MapPoints, seen by KF2 at its keypoints src
    T1, mp2 = synth.make_local_map(KF1, 195 + seed, KF2.n)                  # KF2's MapPoints, seen by KF1
    want = mp2["src"][mp1["src"]]                                             # KF1 keypoint that the partner of entry e was drawn from
    order = np.full(KF1.n, -1, np.int64); used = np.zeros(KF1.n, bool)
    for e in range(KF1.n):
        if order[want[e]] < 0: order[want[e]] = e; used[e] = True
    order[order < 0] = np.nonzero(~used)[0]
    mp1 = {k: (v[order] if k != "occupied" else v) for k, v in mp1.items()}
    R1, t1 = T1[:3, :3].astype(np.float64), T1[:3, 3].astype(np.float64)
    R2, t2 = T2[:3, :3].astype(np.float64), T2[:3, 3].astype(np.float64)
    R12 = R1 @ R2.T                                                           # cam1 = s12*R12*cam2 + t12
    t12 = t1 - s12 * R12 @ t2
    return KF1, T1, mp1, KF2, T2, mp2, R12.astype(np.float32), t12.astype(np.float32)


@pytest.mark.parametrize("seed,s12,th", [(0, 1.0, 7.5), (1, 1.04, 7.5)])
def test_search_by_sim3_whole_routine_on_device(gpu_ctx, seed, s12, th):
    """LoopClosing::ComputeSim3's guided matcher: both projection loops (two gemms each), both searches and the agreement check."""
    KF1, T1, mp1, KF2, T2, mp2, R12, t12 = _sim3_pair(seed, s12)
    v1 = orb_search.frame_view(T1, synth.KITTI_CAM, KF1); v2 = orb_search.frame_view(T2, synth.KITTI_CAM, KF2)
    m12, found = ORBmatcher(gpu_ctx).SearchBySim3Points(KF1, v1, mp1, KF2, v2, mp2, s12, R12, t12, th)
    sR12, t12f, sR21, t21 = orb_search.sim3_transforms(s12, R12, t12)
    va, uva, la = OS.project_general(_mixed_view(v1, v1, v2), mp1, orb_search.PROJ_SIM3_DIR, sR21, t21)
    vb, uvb, lb = OS.project_general(_mixed_view(v2, v1, v1), mp2, orb_search.PROJ_SIM3_DIR, sR12, t12f)
    assert va.sum() > 600 and vb.sum() > 600
    m1 = OS.search_sim3_direction(KF2, mp1["desc"], va, uva, la, th)
    m2 = OS.search_sim3_direction(KF1, mp2["desc"], vb, uvb, lb, th)
    exp = np.array([m1[i] if m1[i] >= 0 and m2[m1[i]] == i else -1 for i in range(KF1.n)], np.int32)
    np.testing.assert_array_equal(m12, exp)
    assert found == int((exp >= 0).sum()) and found > 100
    # one direction alone, with its projections returned
    out, uv, lvl = orb_search.search_projected(gpu_ctx.lib, gpu_ctx.handle, KF2, _mixed_view(v1, v1, v2), mp1, orb_search.PROJ_SIM3_DIR, th,
                                               sR=sR21, t=t21)
    k = va != 0
    np.testing.assert_array_equal(uv[k], uva[k]); np.testing.assert_array_equal(lvl[k], la[k])
    np.testing.assert_array_equal(out.match, m1)


def _mixed_view(pose, intr, searched):
    """SearchBySim3 projects with pKF1's fx..cy in both directions and tests IsInImage / PredictScale on the keyframe searched in."""
    v = orb_search.FrameView.from_buffer_copy(pose)
    v.fx, v.fy, v.cx, v.cy = intr.fx, intr.fy, intr.cx, intr.cy
    v.min_x, v.max_x, v.min_y, v.max_y = searched.min_x, searched.max_x, searched.min_y, searched.max_y
    v.log_scale_factor, v.n_levels = searched.log_scale_factor, searched.n_levels
    return v


def test_projected_search_edge_cases(gpu_ctx):
    KF = synth.make_orb_frame(199, 300)
    T, mp = synth.make_local_map(KF, 199, 200)
    view = orb_search.frame_view(T, synth.KITTI_CAM, KF)
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    empty = {k: (v[:0] if k != "occupied" else v) for k, v in mp.items()}
    for routine in range(4):
        out, uv, lvl = orb_search.search_projected(lib, h, KF, view, empty, routine, 5.0, accept_max=100, sR=np.eye(3), t=np.zeros(3))
        assert out.n_matches == 0 and out.match.shape == (0,)
        allskip = dict(mp, skip=np.ones(200, np.uint8))
        out, uv, lvl = orb_search.search_projected(lib, h, KF, view, allskip, routine, 5.0, accept_max=100, sR=np.eye(3), t=np.zeros(3))
        assert out.n_matches == 0 and (out.match == -1).all()
    with pytest.raises(RuntimeError, match="invalid"):
        orb_search.search_projected(lib, h, KF, view, mp, 7, 5.0)
    bare = {k: v for k, v in mp.items() if k != "normal"}
    with pytest.raises(RuntimeError, match="invalid"):                    # the viewing-angle test of :350 needs the normals
        orb_search.search_projected(lib, h, KF, view, bare, orb_search.PROJ_KF_SIM3, 5.0)
    with pytest.raises(RuntimeError, match="invalid"):                    # orientation check without the keyframe's angles
        orb_search.search_projected(lib, h, KF, view, mp, orb_search.PROJ_RELOC, 5.0, accept_max=100, check_orientation=True)
    # a point exactly behind the camera plane: z = 0 passes `z<0.0`, 1/z = inf, u = nan fails IsInImage; RELOC rejects it on the bounds
    P0 = (T[:3, :3].astype(np.float64).T @ (np.array([0.3, 0.2, 0.0]) - T[:3, 3].astype(np.float64))).astype(np.float32)
    one = {k: (v[:1].copy() if k != "occupied" else v) for k, v in mp.items()}
    one["world_pos"][0] = P0; one["skip"][0] = 0
    for routine in range(3):
        out, _, _ = orb_search.search_projected(lib, h, KF, view, one, routine, 5.0, accept_max=100)
        valid, _, _ = OS.project_general(view, one, routine)
        assert out.n_matches == 0 and valid[0] == 0


# ---------------------------------------------------------------------- Frame::ComputeStereoMatches, whole routine (with images)
def _check_stereo(g, ref):
    n, ur, dep, br, sad = ref
    np.testing.assert_array_equal(g.best_r, br)
    np.testing.assert_array_equal(g.sad, sad)
    np.testing.assert_array_equal(g.u_right.view(np.uint32), ur.view(np.uint32))     # float results bit for bit
    np.testing.assert_array_equal(g.depth.view(np.uint32), dep.view(np.uint32))
    assert g.n_matches == n


@pytest.mark.parametrize("seed,n", [(0, 2000), (1, 700), (2, 4096)])
def test_compute_stereo_matches_whole_routine(gpu_ctx, seed, n):
    sc = synth.make_stereo_scene(seed, n)
    g = ORBmatcher(gpu_ctx).ComputeStereoMatchesFull(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    ref = OS.compute_stereo_matches(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    _check_stereo(g, ref)
    assert g.n_matches > n // 3


def test_compute_stereo_matches_strided_images_and_borders(gpu_ctx):
    sc = synth.make_stereo_scene(3, 600)
    L, R = sc["L"], sc["R"]
    L.xy[:50, 0] = 2.0; R.xy[:50, 0] = 1.0                           # patches leave the image: no match, no fault
    L.xy[50:80, 1] = 374.0
    pad = lambda a: np.pad(a, ((0, 0), (0, 13)))[:, :a.shape[1]]     # views with a row step larger than the width (cv::Mat::step)
    left, right = [pad(a) for a in sc["left"]], [pad(a) for a in sc["right"]]
    assert left[0].strides[0] > left[0].shape[1]
    g = ORBmatcher(gpu_ctx).ComputeStereoMatchesFull(L, R, left, right, sc["inv_scale"], sc["mb"], sc["mbf"])
    _check_stereo(g, OS.compute_stereo_matches(L, R, left, right, sc["inv_scale"], sc["mb"], sc["mbf"]))
    assert (g.u_right[:50] < 0).all()


def test_compute_stereo_matches_empty_sides(gpu_ctx):
    sc = synth.make_stereo_scene(4, 300)
    E = Frame(desc=np.zeros((0, 8), np.uint32), xy=np.zeros((0, 2), np.float32), octave=np.zeros(0, np.int32), uright=np.zeros(0, np.float32),
              angle=np.zeros(0, np.float32))
    m = ORBmatcher(gpu_ctx)
    g = m.ComputeStereoMatchesFull(sc["L"], E, sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    assert g.n_matches == 0 and (g.u_right < 0).all() and (g.depth < 0).all()
    g = m.ComputeStereoMatchesFull(E, sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    assert g.n_matches == 0 and g.u_right.shape == (0,)


def test_compute_stereo_matches_device_resident_pyramids(gpu_ctx):
    """on_device = 1: the image pointers are HBM pointers (here torch tensors), only keypoints travel."""
    import ctypes as C
    import torch
    from lld_slam_amd import orb_search as S
    from lld_slam_amd.abi import c_uint8_p
    sc = synth.make_stereo_scene(5, 1500)
    ref = OS.compute_stereo_matches(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
    dl = [torch.from_numpy(a).cuda() for a in sc["left"]]; dr = [torch.from_numpy(a).cuda() for a in sc["right"]]
    torch.cuda.synchronize()
    kl, kr = S.keypoints_struct(sc["L"]), S.keypoints_struct(sc["R"])
    P, keep = S.pyramids_struct(sc["left"], sc["right"], sc["L"].scale, sc["inv_scale"])
    lp = (c_uint8_p * len(dl))(*[C.cast(t.data_ptr(), c_uint8_p) for t in dl]); rp = (c_uint8_p * len(dr))(*[C.cast(t.data_ptr(), c_uint8_p) for t in dr])
    P.left = C.cast(lp, C.POINTER(c_uint8_p)); P.right = C.cast(rp, C.POINTER(c_uint8_p)); P.on_device = 1
    n = sc["L"].n
    g = S.StereoMatches(np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.int32), np.empty(n, np.int32), 0)
    r = S.StereoResult(); r.u_right = g.u_right.ctypes.data_as(S.c_float_p); r.depth = g.depth.ctypes.data_as(S.c_float_p)
    r.best_r = g.best_r.ctypes.data_as(S.c_int32_p); r.sad = g.sad.ctypes.data_as(S.c_int32_p)
    fn = gpu_ctx.lib.fn("compute_stereo_matches")
    fn.argtypes = [C.c_void_p, C.POINTER(S.Keypoints), C.POINTER(S.Keypoints), C.POINTER(S.StereoPyramids), C.c_float, C.c_float, C.POINTER(S.StereoResult)]
    fn.restype = C.c_int
    assert fn(gpu_ctx.handle, C.byref(kl), C.byref(kr), C.byref(P), sc["mb"], sc["mbf"], C.byref(r)) == 0
    g.n_matches = r.n_matches
    _check_stereo(g, ref)


# ------------------------------------------------------------------ ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520)
@pytest.mark.parametrize("pid,n,window,nn,check", [(0, 2000, 100, 0.9, True), (1, 2000, 30, 0.9, True), (2, 1200, 60, 0.7, False), (3, 3000, 150, 0.9, True),
                                                   (4, 300, 10, 0.9, True)])
def test_search_for_initialization(gpu_ctx, pid, n, window, nn, check):
    """The one matcher whose order dependence is not an occupancy: a keypoint of F2 goes to the query with the smallest distance so far
    (an earlier holder with a distance <= this one blocks, a better query takes it over).  vnMatches12, nmatches and the updated
    vbPrevMatched equal the sequential oracle."""
    F1, F2, prev = synth.make_init_pair(pid, n=n)
    on, om, opm = OS.search_for_initialization(F1, F2, prev, window, nn, check)
    gn, gm, gpm = ORBmatcher(gpu_ctx, nn, check).SearchForInitialization(F1, F2, prev, window)
    assert gn == on and (on > 20 or n < 500)
    np.testing.assert_array_equal(gm, om); np.testing.assert_array_equal(gpm, opm)


def test_search_for_initialization_with_crowded_windows(gpu_ctx):
    """Every level-0 keypoint of F1 is a near-duplicate of one of 40 prototypes sitting in one spot: hundreds of queries compete for
    the same few keypoints of F2 and most cached candidate lists run dry (fewer than two free entries of a full list), which sends
    the query through the rescan with the take-over rule."""
    F1, F2, prev = synth.make_init_pair(7, n=1500, rival_frac=0.0)
    rng = np.random.default_rng(5)
    proto = rng.integers(0, 1500, 40)
    for i in range(1500):
        if i in proto: continue
        p = proto[i % 40]
        F1.desc[i] = synth._flip_bits(rng, F1.desc[p:p + 1], 0.004)[0]; F1.xy[i] = F1.xy[p] + rng.integers(-3, 4, 2).astype(np.float32); F1.octave[i] = 0
    F2.xy[:400] = F1.xy[proto[np.arange(400) % 40]] + rng.integers(-6, 7, (400, 2)).astype(np.float32); F2.octave[:400] = 0
    F2.desc[:400] = synth._flip_bits(rng, F1.desc[proto[np.arange(400) % 40]], 0.02)
    F1.normalise(); F2.normalise()
    prev = F1.xy.copy()
    on, om, opm = OS.search_for_initialization(F1, F2, prev, 40, 0.95, True)
    info = {}
    gn, gm, gpm = orb_search.search_for_initialization(gpu_ctx.lib, gpu_ctx.handle, F1, F2, prev, 40, 0.95, True, info=info)
    assert gn == on and info["rescans"] > 20
    np.testing.assert_array_equal(gm, om); np.testing.assert_array_equal(gpm, opm)


def test_search_for_initialization_edge_cases(gpu_ctx):
    F1, F2, prev = synth.make_init_pair(9, n=200)
    F1.octave[:] = 1; F1.normalise()                                          # no level-0 keypoint: nothing is searched
    n, m, pm = ORBmatcher(gpu_ctx, 0.9, True).SearchForInitialization(F1, F2, prev, 100)
    assert n == 0 and np.all(m == -1) and np.array_equal(pm, prev)
    F1, F2, prev = synth.make_init_pair(9, n=200)
    far = prev + np.float32(5000.0)                                           # windows outside the image
    n, m, pm = ORBmatcher(gpu_ctx, 0.9, True).SearchForInitialization(F1, F2, far, 100)
    on, om, _ = OS.search_for_initialization(F1, F2, far, 100, 0.9, True)
    assert n == on == 0 and np.all(m == -1)
