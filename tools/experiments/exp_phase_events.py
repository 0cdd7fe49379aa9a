#!/usr/bin/env python3
"""Wall clock of lld_ba_batch_solve on a resident batch with and without the per-phase HIP events (lld_ba_batch_set_phase_timing).
   python tools/experiments/exp_phase_events.py [windows=1] [repeats=30]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from lld_slam_amd import BABatch, Context, abi, synth

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lib = abi.product()
ON = os.environ.get("LLD_PHASE_EVENTS") == "1"
ws = [synth.make_lba_b(i) for i in range(nw)]
with Context(0, lib=lib) as ctx, BABatch(ctx, ws) as b:
    b.set_phase_timing(ON)
    for _ in range(3): b.solve()
    t = []
    for _ in range(rep):
        t0 = time.perf_counter(); b.solve(); t.append((time.perf_counter() - t0) * 1e3)
    ph = b.phase_ms()
    st = b.stats()
print("%-8s phase events  %3d windows  solve wall ms: min %.3f median %.3f   device total %.3f ms   trials of window 0: %d  chi2 %.12g%s" %
      ("with" if ON else "without", nw, min(t), float(np.median(t)), ph[5], sum(st[0]["lm_trials"]), st[0]["chi2_final"], "  [hipGraph]" if os.environ.get("LLD_BA_GRAPH") else ""))
