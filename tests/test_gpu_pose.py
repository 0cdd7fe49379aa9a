"""GPU parity: Optimizer::PoseOptimization through the C ABI vs the CPU oracle.

Tolerance (BASELINE.json north_star): final pose and chi2 within 1e-5 relative, identical inlier/outlier sets.
"""
import numpy as np
import pytest

from lld_slam_amd import Optimizer, PoseBatch, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _check(g, o, n_points, weak=False):
    assert g.n_inliers == o.n_inliers
    np.testing.assert_array_equal(g.pt_outlier, o.pt_outlier)
    np.testing.assert_array_equal(g.ln_outlier, o.ln_outlier)
    np.testing.assert_allclose(g.pose_qt, o.pose_qt, rtol=RTOL, atol=1e-7)
    assert g.chi2 == pytest.approx(o.chi2, rel=RTOL)
    # LM accept/reject decisions at convergence hinge on chi2 differences at rounding level (rho ~ 0/0), so the trial
    # count may differ by a few while the result does not: the differences actually seen are logged (gpurun_out/lm_counts_test_gpu_pose.txt
    # -> profiles/r06_parity_margins.txt) and held to their maximum
    _LM_LOG.append((g.lm_iterations - o.lm_iterations, g.lm_trials - o.lm_trials))
    assert abs(g.lm_iterations - o.lm_iterations) <= MAX_IT_DIFF and abs(g.lm_trials - o.lm_trials) <= (MAX_TRIAL_DIFF_WEAK if weak else MAX_TRIAL_DIFF), _LM_LOG[-1]


# Seen on HEAD over the 66 calls of this file (round 6, after the wavefront sums left the LDS - another summation tree, so other last bits than before):
# 62 calls equal, iterations differ by 1 in one call, trials by <= 2 in three.  The one frame with monocular observations only is weakly constrained along
# the viewing direction: its converged rounds spend their ten rejected trials or not on the last bits of two sums (device 20 iterations / 70 trials, oracle
# 20 / 52, same pose to 1e-11; profiles/NOTES_r06.md walks such a frame round by round) - it gets its own bound instead of loosening everyone's.
MAX_IT_DIFF, MAX_TRIAL_DIFF, MAX_TRIAL_DIFF_WEAK = 1, 2, 20
_LM_LOG = []


@pytest.fixture(scope="module", autouse=True)
def _write_lm_log():
    yield
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if _LM_LOG and os.path.isdir(d):
        with open(os.path.join(d, "lm_counts_test_gpu_pose.txt"), "w") as f:
            f.write("# per _check call of tests/test_gpu_pose.py: device minus oracle, LM iterations and trials\n")
            for a, b in _LM_LOG: f.write(f"{a:+d} {b:+d}\n")
            f.write(f"# {len(_LM_LOG)} calls, max |iterations| {max(abs(a) for a, _ in _LM_LOG)}, max |trials| {max(abs(b) for _, b in _LM_LOG)}, calls with any difference {sum(1 for a, b in _LM_LOG if a or b)}\n")


@pytest.mark.parametrize("fid,kw", [
    (0, dict()),                                            # config PO: 1000 stereo points + 200 stereo lines
    (1, dict(n_points=300, n_lines=60)),
    (2, dict(n_points=200, n_lines=40, mono_frac=0.3, mono_line_frac=0.3)),
    (3, dict(n_points=50, n_lines=0)),
    (4, dict(n_points=5, n_lines=2)),                       # fewer than 10 edges: early break before line classification
    (5, dict(n_points=2, n_lines=5)),                       # fewer than 3 points: returns 0
    (6, dict(n_points=400, n_lines=80, outlier_frac=0.3)),
    (30, dict(n_points=0, n_lines=0)),                      # nothing to optimise
    (30, dict(n_points=200, n_lines=40, mono_frac=1.0, mono_line_frac=1.0)),   # a monocular frame: 2-D point edges, left line edges only
    (30, dict(n_points=0, n_lines=50)),                     # lines only: nInitialCorrespondences < 3 -> 0
    (30, dict(n_points=400, n_lines=80, outlier_frac=0.9)),   # almost everything is rejected in round 1
])
def test_pose_optimization_matches_oracle(gpu_ctx, oracle, fid, kw):
    f = synth.make_pose_frame(fid, **kw)
    g = Optimizer(gpu_ctx).PoseOptimization(f, gamma=0.5)
    o = oracle.pose_opt(f, gamma=0.5)
    _check(g, o, f.n_points, weak=kw.get("mono_frac", 0.0) == 1.0)
    if f.n_points >= 50 and kw.get("mono_frac", 0.0) < 1.0 and kw.get("outlier_frac", 0.0) < 0.5:
        gt = f.meta["gt_qt"]
        assert np.linalg.norm(g.pose_qt[4:] - gt[4:]) < 0.05


def test_pose_gamma_one(gpu_ctx, oracle):
    f = synth.make_pose_frame(7, n_points=300, n_lines=60)
    _check(Optimizer(gpu_ctx).PoseOptimization(f, gamma=1.0), oracle.pose_opt(f, gamma=1.0), f.n_points)


def test_pose_batch_ragged_frames(gpu_ctx, oracle):
    frames = [synth.make_pose_frame(20 + i, n_points=100 + 37 * i, n_lines=10 * i, mono_frac=0.1 * (i % 3)) for i in range(9)]
    with PoseBatch(gpu_ctx, frames, gamma=0.5) as b:
        for _ in range(2):                                   # a second solve restarts from the uploaded state
            b.solve()
            for i, f in enumerate(frames):
                _check(b.download(i), oracle.pose_opt(f, gamma=0.5), f.n_points)


def test_pose_frame_larger_than_lds(gpu_ctx, oracle):
    """3000 points + 500 lines do not fit the CU's LDS: the kernel runs on its HBM-resident working arrays instead."""
    f = synth.make_pose_frame(40, n_points=3000, n_lines=500)
    _check(Optimizer(gpu_ctx).PoseOptimization(f, gamma=0.5), oracle.pose_opt(f, gamma=0.5), f.n_points)
    # a batch takes the mode of its largest frame
    frames = [synth.make_pose_frame(41, n_points=200, n_lines=30), f]
    with PoseBatch(gpu_ctx, frames, gamma=0.5) as b:
        b.solve()
        for i, fr in enumerate(frames):
            _check(b.download(i), oracle.pose_opt(fr, gamma=0.5), fr.n_points)


def test_pose_batch_larger_than_the_gpu_runs_two_frames_per_cu(gpu_ctx, oracle):
    """More frames than compute units (300 > 256) and frames whose LDS image fits twice into a CU: the batch runs 256 lanes per frame, two
    frames per CU (lld_pose.hip, pose_mode) - another (fixed) summation order than the 512-lane form of small batches and single calls.
    Every frame against the single call (rounding-level agreement, identical sets), a spread of them against the oracle, and a second
    solve bit for bit."""
    kinds = [dict(n_points=1000, n_lines=200), dict(n_points=300, n_lines=60, mono_frac=0.3, mono_line_frac=0.3), dict(n_points=5, n_lines=2),
             dict(n_points=400, n_lines=80, outlier_frac=0.6), dict(n_points=0, n_lines=20), dict(n_points=120, n_lines=0)]
    distinct = [synth.make_pose_frame(60 + i, **kinds[i % len(kinds)]) for i in range(12)]
    frames = [distinct[i % 12] for i in range(300)]
    opt = Optimizer(gpu_ctx)
    single = [opt.PoseOptimization(f, gamma=0.5) for f in distinct]
    with PoseBatch(gpu_ctx, frames, gamma=0.5) as b:
        b.solve()
        first = [b.download(i) for i in range(300)]
        for i, g in enumerate(first):
            s = single[i % 12]
            assert g.n_inliers == s.n_inliers
            np.testing.assert_array_equal(g.pt_outlier, s.pt_outlier); np.testing.assert_array_equal(g.ln_outlier, s.ln_outlier)
            np.testing.assert_allclose(g.pose_qt, s.pose_qt, rtol=1e-9, atol=1e-10)
            if i >= 12:                                      # the same frame in another workgroup of the same launch: the same bits
                np.testing.assert_array_equal(g.pose_qt, first[i % 12].pose_qt); assert g.chi2 == first[i % 12].chi2
        for i in range(12):
            _check(first[i], oracle.pose_opt(distinct[i], gamma=0.5), distinct[i].n_points)
        b.solve()
        for i in (0, 13, 299):
            g = b.download(i)
            np.testing.assert_array_equal(g.pose_qt, first[i].pose_qt); assert g.chi2 == first[i].chi2


def test_pose_observations_that_are_no_widened_floats_keep_their_doubles(gpu_ctx, oracle):
    """The LDS image holds the image observations as floats when each of them is a widened float (what key points, uRight, key lines and
    level sigmas are); one observation with more mantissa than a float makes the whole batch keep the doubles as given - and still
    matches the oracle, which computes on the doubles."""
    import dataclasses
    f = synth.make_pose_frame(70, n_points=500, n_lines=100)
    assert np.all(f.pt_uvr.astype(np.float32).astype(np.float64) == f.pt_uvr)            # the generator makes widened floats ...
    uvr = f.pt_uvr.copy(); uvr[3, 0] += 1e-9; left = f.ln_left.copy(); left[5, 1] += 1e-9
    f2 = dataclasses.replace(f, pt_uvr=uvr, ln_left=left)                                 # ... these two are none
    _check(Optimizer(gpu_ctx).PoseOptimization(f2, gamma=0.5), oracle.pose_opt(f2, gamma=0.5), f2.n_points)
    _check(Optimizer(gpu_ctx).PoseOptimization(f, gamma=0.5), oracle.pose_opt(f, gamma=0.5), f.n_points)


def test_pose_single_calls_reuse_context_buffers(gpu_ctx, oracle):
    """lld_pose_opt stages through the context's pinned / device scratch: sizes going up and down must not leak state."""
    opt = Optimizer(gpu_ctx)
    for fid, n, m in [(50, 300, 40), (51, 1000, 200), (52, 20, 0), (50, 300, 40)]:
        f = synth.make_pose_frame(fid, n_points=n, n_lines=m)
        _check(opt.PoseOptimization(f, gamma=0.5), oracle.pose_opt(f, gamma=0.5), f.n_points)


@pytest.mark.parametrize("fid,kw", [(40, dict(n_points=300, n_lines=80, mono_line_frac=0.4)), (41, dict(n_points=200, n_lines=50, mono_frac=0.3, mono_line_frac=0.6, outlier_frac=0.2))])
def test_frame_line_indices_select_the_threshold(gpu_ctx, oracle, fid, kw):
    """lld_pose_problem::ln_frame_index (vnIndexLines): the reference reads vnStereoLines - one entry per EDGE - with the line's index in
    the frame (Optimizer.cc:893-898).  Random increasing frame indices (lines without a MapLine in between), device vs oracle, and the
    known-answer frame of tests/test_oracle_kat.py."""
    import dataclasses
    rng = np.random.default_rng(fid)
    f = synth.make_pose_frame(fid, **kw)
    f = dataclasses.replace(f, ln_frame_index=np.cumsum(rng.integers(1, 4, f.n_lines)).astype(np.int32) - 1)
    _check(Optimizer(gpu_ctx).PoseOptimization(f, gamma=0.5), oracle.pose_opt(f, gamma=0.5), f.n_points)
    with PoseBatch(gpu_ctx, [f, dataclasses.replace(f, ln_frame_index=None)], gamma=0.5) as b:
        b.solve()
        _check(b.download(0), oracle.pose_opt(f, gamma=0.5), f.n_points)
        _check(b.download(1), oracle.pose_opt(dataclasses.replace(f, ln_frame_index=None), gamma=0.5), f.n_points)


@pytest.mark.parametrize("what", ["nan_keypoint", "inf_map_point", "nan_pose"])
def test_non_finite_frame_terminates_and_leaves_the_context_clean(gpu_ctx, oracle, what):
    """A NaN / Inf in one frame: the call returns (4 rounds of at most 10 iterations x 10 trials), the other frames of a batch are the
    oracle's answers, and so is the next call on the same context."""
    import copy
    clean = synth.make_pose_frame(8, n_points=300, n_lines=60)
    bad = copy.deepcopy(clean)
    if what == "nan_keypoint": bad.pt_uvr = bad.pt_uvr.copy(); bad.pt_uvr[11, 1] = np.nan
    elif what == "inf_map_point": bad.pt_xw = bad.pt_xw.copy(); bad.pt_xw[3, 0] = np.inf
    else: bad.pose_qt = bad.pose_qt.copy(); bad.pose_qt[5] = np.nan
    g = Optimizer(gpu_ctx).PoseOptimization(bad, gamma=0.5)
    assert g.lm_iterations <= 40 and g.lm_trials <= 400
    ref = oracle.pose_opt(clean, gamma=0.5)
    with PoseBatch(gpu_ctx, [clean, bad, clean], gamma=0.5) as b:
        b.solve()
        _check(b.download(0), ref, clean.n_points); _check(b.download(2), ref, clean.n_points)
    _check(Optimizer(gpu_ctx).PoseOptimization(clean, gamma=0.5), ref, clean.n_points)



def test_cross_lane_sums_of_the_pose_kernel():
    """pose_opt_kernel's wavefront sums without the LDS (round 6: v_permlane32_swap / v_permlane16_swap / DPP pairings instead of ds_bpermute): every
    one of the 28 totals and the one-value sum as every lane sees it, against numpy on random and on adversarial inputs (one lane carries everything;
    a different value in every lane and slot) - and identical from run to run."""
    import ctypes as C, os
    from lld_slam_amd import Context, abi
    lib = abi.Lib(os.path.join(os.path.dirname(abi.product_library_path()), "liblld_amd_exp.so"), "lld_")
    ctx = Context(0, lib=lib)
    try:
        fn = lib.fn("exp_pose_wave_sums")
        dp = C.POINTER(C.c_double)
        fn.argtypes = [C.c_void_p, dp, dp, dp]; fn.restype = C.c_int
        rng = np.random.default_rng(5)
        cases = [rng.normal(size=(64, 28)), np.arange(64 * 28, dtype=np.float64).reshape(64, 28) + 1.0, np.zeros((64, 28))]
        cases[2][37] = rng.normal(size=28) * 1e6
        for v in cases:
            v = np.ascontiguousarray(v)
            outs = []
            for _ in range(2):
                o28 = np.zeros(28); o1 = np.zeros(64)
                assert fn(ctx.handle, v.ctypes.data_as(dp), o28.ctypes.data_as(dp), o1.ctypes.data_as(dp)) == 0
                outs.append((o28, o1))
            np.testing.assert_array_equal(outs[0][0], outs[1][0]); np.testing.assert_array_equal(outs[0][1], outs[1][1])
            scale = np.abs(v).sum(0) + 1e-300
            assert np.all(np.abs(outs[0][0] - v.sum(0)) <= 1e-14 * scale), (outs[0][0], v.sum(0))
            assert np.all(outs[0][1] == outs[0][1][0]) and abs(outs[0][1][0] - v[:, 0].sum()) <= 1e-14 * scale[0]
    finally:
        ctx.close()
