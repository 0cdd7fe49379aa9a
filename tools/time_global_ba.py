"""Optimizer::GlobalBundleAdjustment protocol on maps beyond the LDS limit of the landmark kernels (BAWin::big, multi-workgroup PCG):
   python tools/time_global_ba.py [n_keyframes ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lld_slam_amd import Context, Optimizer, synth
ctx = Context(0); opt = Optimizer(ctx)
for n in [int(a) for a in sys.argv[1:]] or [1000, 2000]:
    t = time.perf_counter(); w = synth.make_ba_window(n, 2, 25 * n, 4, 2 * n, 4, seed=0x6BA01000 + n); tg = time.perf_counter() - t
    opt.GlobalBundleAdjustment(w, 2)
    t = time.perf_counter(); g = opt.GlobalBundleAdjustment(w, 10); dt = time.perf_counter() - t
    print(f"{n} keyframes, {w.n_edges()} edges (generated in {tg:.1f} s): 10 iterations {1e3 * dt:.0f} ms, chi2 {g.stats['chi2_final']:.6g}, "
          f"LM {g.stats['lm_iterations']} trials {g.stats['lm_trials']}, PCG iterations {g.stats['pcg_iterations']}")
