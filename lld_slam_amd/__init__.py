"""MI355X-native point+line local-BA / pose-optimisation / descriptor-matching core.

Only the hot path of alexandervakhitov/lld-slam lives here: hand-written HIP kernels for gfx950 behind the C ABI
of include/lld_amd.h (csrc/), plus the host-side mirror of the reference interface (host.py) and the synthetic
problem generators (synth.py).  There is no CPU fallback: everything that computes goes through
csrc/liblld_amd.so and fails loudly when the library or the GPU is missing.
"""
from . import abi  # noqa: F401
from .host import (BABatch, Context, Optimizer, ORBmatcher, PoseBatch, PoseFrame, Tracking, TwoFrameLineMatcher,  # noqa: F401
                   Window)

__all__ = ["abi", "Context", "Optimizer", "ORBmatcher", "TwoFrameLineMatcher", "Tracking", "BABatch", "PoseBatch", "Window",
           "PoseFrame"]
