import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, synth, orb_search as S
ctx = Context(0)
for win in (10, 30, 100):
    F1, F2, prev = synth.make_init_pair(0)
    info = {}
    S.search_for_initialization(ctx.lib, ctx.handle, F1, F2, prev, win, 0.9, True, info=info)
    ts = []
    for _ in range(11):
        t = time.perf_counter(); r = S.search_for_initialization(ctx.lib, ctx.handle, F1, F2, prev, win, 0.9, True); ts.append(time.perf_counter() - t)
    print("window", win, "rescans", info["rescans"], "valid queries", int((F1.octave == 0).sum()), "ms", 1e3 * np.median(ts), "matches", r[0])
