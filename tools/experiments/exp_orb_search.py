#!/usr/bin/env python3
"""Times the guided ORB search routines call by call (kernel time comes from a rocprofv3 kernel trace of this script)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from lld_slam_amd import Context, ORBmatcher, synth

import lld_slam_amd.abi as _abi
_orig_fn = _abi.Lib.fn
CT = {"t": 0.0, "n": 0}
def _timed_fn(self, name):                       # time spent inside the C ABI call alone (the numpy packing of the mirror excluded)
    f = _orig_fn(self, name)
    if not (name.startswith("orb_") or name.startswith("compute_stereo")): return f
    class W:
        argtypes = None; restype = None
        def __call__(s, *a):
            f.argtypes = s.argtypes; f.restype = s.restype
            t = time.perf_counter(); r = f(*a); CT["t"] += time.perf_counter() - t; CT["n"] += 1
            return r
    return W()
_abi.Lib.fn = _timed_fn

ctx = Context(0)
F = synth.make_orb_frame(0, 2000); q = synth.make_projection_queries(F, 0, 2000, dup_frac=0.3)
L, R = synth.make_stereo_pair(0, 2000)
F1, F2, nd = synth.make_bow_pair(0, 2000); v = np.ones(2000, np.uint8)
m = ORBmatcher(ctx, 0.8)
for name, f in (("map", lambda: m.SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0)),
                ("frame", lambda: m.SearchByProjectionFrame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], 0, 15.0)),
                ("fuse", lambda: m.Fuse(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], 3.0)),
                ("stereo", lambda: m.ComputeStereoMatches(L, R, 0.0, 100.0)),
                ("bow", lambda: m.SearchByBoWFrame(F1, F2, nd, v)),
                ("stereo_full", None)):
    if f is None:
        sc = synth.make_stereo_scene(0, 2000)
        f = lambda: m.ComputeStereoMatchesFull(sc["L"], sc["R"], sc["left"], sc["right"], sc["inv_scale"], sc["mb"], sc["mbf"])
        f(); ts = []
        for _ in range(20):
            CT["t"] = 0.0; t = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t, CT["t"]))
        ts = np.median(np.array(ts), 0)
        print(name, "ms/call %.3f (inside the C ABI %.3f)" % (ts[0] * 1e3, ts[1] * 1e3), "matches", r.n_matches, flush=True)
        continue
    f(); ts = []
    for _ in range(20):
        CT["t"] = 0.0; t = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t, CT["t"]))
    ts = np.median(np.array(ts), 0)
    print(name, "ms/call %.3f (inside the C ABI %.3f)" % (ts[0] * 1e3, ts[1] * 1e3), "rounds", r.rounds, "matches", r.n_matches, flush=True)
ctx.close()
