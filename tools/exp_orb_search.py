#!/usr/bin/env python3
"""Times the guided ORB search routines call by call (kernel time comes from a rocprofv3 kernel trace of this script)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lld_slam_amd import Context, ORBmatcher, synth

ctx = Context(0)
F = synth.make_orb_frame(0, 2000); q = synth.make_projection_queries(F, 0, 2000, dup_frac=0.3)
L, R = synth.make_stereo_pair(0, 2000)
F1, F2, nd = synth.make_bow_pair(0, 2000); v = np.ones(2000, np.uint8)
m = ORBmatcher(ctx, 0.8)
for name, f in (("map", lambda: m.SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0)),
                ("frame", lambda: m.SearchByProjectionFrame(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["angle"], q["obs"], q["occupied"], 0, 15.0)),
                ("fuse", lambda: m.Fuse(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], 3.0)),
                ("stereo", lambda: m.ComputeStereoMatches(L, R, 0.0, 100.0)),
                ("bow", lambda: m.SearchByBoWFrame(F1, F2, nd, v))):
    f(); t = time.perf_counter()
    for _ in range(10): r = f()
    print(name, "ms/call %.3f" % ((time.perf_counter() - t) * 100), "rounds", r.rounds, "matches", r.n_matches, flush=True)
ctx.close()
