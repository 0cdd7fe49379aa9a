"""Error behaviour of the newer entry points: malformed arguments are refused with a status, nothing is launched."""
import ctypes as C

import numpy as np
import pytest

from lld_slam_amd import abi, orb_search as S, synth

pytestmark = pytest.mark.gpu


def test_null_and_malformed_arguments_return_a_status(gpu_ctx):
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    f = lib.fn("optimize_sim3"); f.argtypes = [C.c_void_p] * 4; f.restype = C.c_int
    assert f(h, None, None, None) != 0
    p = synth.make_sim3_pair(0, 30).to_c(); r = abi.Sim3Result()
    assert f(h, C.addressof(p), None, C.addressof(r)) != 0                      # result.dropped missing
    p.n = -1
    d = np.zeros(30, np.uint8); r.dropped = d.ctypes.data_as(abi.c_uint8_p)
    assert f(h, C.addressof(p), None, C.addressof(r)) != 0
    g = lib.fn("optimize_essential_graph"); g.argtypes = [C.c_void_p] * 4; g.restype = C.c_int
    assert g(h, None, None, None) != 0
    G = abi.PoseGraph(); G.n_vertices = 3; G.n_edges = 1                        # arrays missing
    o = np.zeros((3, 8)); R = abi.PoseGraphResult(); R.sim3 = o.ctypes.data_as(abi.c_double_p)
    assert g(h, C.addressof(G), None, C.addressof(R)) != 0
    k = lib.fn("compute_stereo_matches"); k.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]; k.restype = C.c_int
    assert k(h, None, None, None, 0.5, 380.0, None) != 0
    sc = synth.make_stereo_scene(6, 200, width=320, height=150)
    kl, kr = S.keypoints_struct(sc["L"]), S.keypoints_struct(sc["R"])
    P, keep = S.pyramids_struct(sc["left"], sc["right"], sc["L"].scale, sc["inv_scale"])
    res = S.StereoResult()                                                     # u_right / depth missing
    assert k(h, C.addressof(kl), C.addressof(kr), C.addressof(P), sc["mb"], sc["mbf"], C.addressof(res)) != 0
    bad_oct = sc["L"].octave.copy(); bad_oct[0] = 99
    kl.octave = bad_oct.ctypes.data_as(abi.c_int32_p)
    ur = np.zeros(200, np.float32); dp = np.zeros(200, np.float32)
    res.u_right = ur.ctypes.data_as(abi.c_float_p); res.depth = dp.ctypes.data_as(abi.c_float_p)
    assert k(h, C.addressof(kl), C.addressof(kr), C.addressof(P), sc["mb"], sc["mbf"], C.addressof(res)) != 0   # octave outside the pyramid
    # the context is still usable afterwards
    from lld_slam_amd import Optimizer
    assert Optimizer(gpu_ctx).OptimizeSim3(synth.make_sim3_pair(1, 60)).n_inliers > 10


def test_zero_iteration_rounds_are_refused(gpu_ctx):
    """optimize(0) would evaluate no error at all, and the outlier classification that follows would read g2o's uninitialised _error
    vectors (undefined behaviour in the reference): the parameter set is refused instead of being given a meaning of our own."""
    from lld_slam_amd import Optimizer
    w = synth.make_lba_small(12, n_free=4, n_fixed=2, n_points=60, n_lines=10)
    for kw in (dict(its_round1=0), dict(its_round2=0), dict(its_round1=0, its_round2=0)):
        with pytest.raises(RuntimeError):
            Optimizer(gpu_ctx).LocalBundleAdjustment(w, **kw)
    assert Optimizer(gpu_ctx).LocalBundleAdjustment(w, its_round1=1, its_round2=1).stats["lm_iterations"] == [1, 1]
