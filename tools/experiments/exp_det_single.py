import sys, time, numpy as np
sys.path.insert(0, ".")
from lld_slam_amd import Context, Optimizer, synth
w = synth.make_lba_b(0)
with Context(0) as ctx:
    opt = Optimizer(ctx)
    for det in (0, 1):
        opt.LocalBundleAdjustment(w, deterministic=det)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); opt.LocalBundleAdjustment(w, deterministic=det); ts.append((time.perf_counter() - t0) * 1e3)
        print("single call deterministic", det, "ms", np.round(np.sort(ts), 3).tolist())
