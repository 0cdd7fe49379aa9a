"""GPU parity: Optimizer::OptimizeSim3 through the C ABI vs the CPU oracle (both differentiate numerically like g2o).
Tolerance: S12 within 1e-5 relative, identical dropped sets and inlier counts."""
import numpy as np
import pytest

from lld_slam_amd import Optimizer, synth

pytestmark = pytest.mark.gpu


def _check(g, o):
    np.testing.assert_array_equal(g.dropped, o.dropped)
    assert g.n_inliers == o.n_inliers and g.n_bad_first == o.n_bad_first
    np.testing.assert_allclose(g.s12_q, o.s12_q, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(g.s12_t, o.s12_t, rtol=1e-5, atol=1e-6)
    assert g.s12_s == pytest.approx(o.s12_s, rel=1e-6)
    if o.chi2 > 0:
        assert g.chi2 == pytest.approx(o.chi2, rel=1e-4)
    assert abs(sum(g.lm_iterations) - sum(o.lm_iterations)) <= 2


@pytest.mark.parametrize("pid,kw,fix", [
    (0, dict(n=300), True),
    (1, dict(n=250, scale=1.08), False),
    (2, dict(n=120, outlier_frac=0.0, noise=0.2), True),
    (4, dict(n=900, outlier_frac=0.3), True),
    (5, dict(n=40, outlier_frac=0.1), True),
])
def test_optimize_sim3_matches_oracle(gpu_ctx, oracle, pid, kw, fix):
    p = synth.make_sim3_pair(pid, **kw)
    g = Optimizer(gpu_ctx).OptimizeSim3(p, bFixScale=fix)
    _check(g, oracle.optimize_sim3(p, bFixScale=fix))
    assert g.n_inliers > p.n // 3


def test_optimize_sim3_early_return_and_empty(gpu_ctx, oracle):
    few = synth.make_sim3_pair(3, 14, outlier_frac=0.6)
    g = Optimizer(gpu_ctx).OptimizeSim3(few); o = oracle.optimize_sim3(few)
    _check(g, o)
    if few.n - o.n_bad_first < 10:
        np.testing.assert_array_equal(g.s12_q, few.s12_q); assert g.n_inliers == 0
    import dataclasses
    empty = dataclasses.replace(few, p1c=np.zeros((0, 3)), p2c=np.zeros((0, 3)), obs1=np.zeros((0, 2)), obs2=np.zeros((0, 2)),
                                inv_sigma2_1=np.zeros(0), inv_sigma2_2=np.zeros(0))
    g = Optimizer(gpu_ctx).OptimizeSim3(empty)
    assert g.n_inliers == 0 and g.dropped.shape == (0,)


def test_optimize_sim3_batch_of_candidates(gpu_ctx, oracle):
    pairs = [synth.make_sim3_pair(10 + i, 80 + 60 * i, outlier_frac=0.05 * i) for i in range(6)]
    gs = Optimizer(gpu_ctx).OptimizeSim3(pairs, th2=10.0)
    for g, p in zip(gs, pairs):
        _check(g, oracle.optimize_sim3(p))
