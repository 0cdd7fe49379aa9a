"""Latency of ONE lld_local_ba call on an LBA-B window, host buffers in and out (LLD_BA_TIMING=1 prints the create / solve / download split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lld_slam_amd import Context, Optimizer, synth
ctx = Context(0); opt = Optimizer(ctx)
w = synth.make_lba_b(0)
for _ in range(3): opt.LocalBundleAdjustment(w)
ts = []
for _ in range(8):
    t = time.perf_counter(); opt.LocalBundleAdjustment(w); ts.append(time.perf_counter() - t)
print("single LBA-B call ms median", 1e3 * float(np.median(ts)))
