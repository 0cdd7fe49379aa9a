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
