"""A fixed-seed slice of the fuzz campaigns (tools/fuzz_matchers.py; the full logs are profiles/r03_fuzz_*): every matcher entry point on
random scenes of random sizes, device vs the oracle's sequential restatements, every integer output bit for bit."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_matchers as FZ  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(FZ.ROUTINES))
def test_random_scenes_bit_for_bit(gpu_ctx, oracle, name):
    matches = 0
    for it in range(12):
        sub = np.random.default_rng([2026, sorted(FZ.ROUTINES).index(name), it])
        sid = int(sub.integers(0, 1 << 30))
        ok, nm = FZ.ROUTINES[name](gpu_ctx, sub, sid)
        assert ok, (name, it, sid)
        matches += int(nm)
    assert matches > 0                                                 # the scenes are not degenerate: something was matched


def test_the_scene_that_found_the_logf_difference(gpu_ctx, oracle, monkeypatch):
    """tools/fuzz_matchers.py, FUZZ_BIG=1, seed 9, scene 3701 (1682 keypoints, 9507 MapPoints, 6128 in view): one predicted level differed
    while the device called its library's logf."""
    monkeypatch.setattr(FZ, "BIG", True)
    ok, nm = FZ.ROUTINES["local_points"](gpu_ctx, np.random.default_rng([9, 3701]), 88745057)
    assert ok and nm > 1000


def test_the_scene_that_found_the_near_singular_line_gate(gpu_ctx, oracle, monkeypatch):
    """tools/fuzz_matchers.py, FUZZ_BIG=1, seed 9, scene 13712 (361 x 391 lines): for one pair of unrelated segments the viewing ray of an end
    point is parallel to the triangulated line to ~1e-8; the normal equations the device used to solve returned a line parameter of the
    other sign than the reference's column-pivoted QR (-9e5 against +2e8) and CheckLinePair's depth test with it."""
    monkeypatch.setattr(FZ, "BIG", True)
    ok, nm = FZ.ROUTINES["line_stereo"](gpu_ctx, np.random.default_rng([9, 13712]), 519931807)
    assert ok and nm > 20
