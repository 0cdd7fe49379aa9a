#!/usr/bin/env python3
"""How long lld_local_ba runs on after the caller raises pbStopFlag (groups of < 24 windows queue four super-steps per host
poll).  Round 4 forwards the live flag through a pinned word that every control step reads, so a raised flag is honoured by the next LM
trial, queued or not.  Prints, for one LBA-B window and for a batch of 8 LBA-A windows, the time from raising the flag (another host
thread, at a random moment of the solve) to the return of the call, over 40 repeats.
    python tools/exp_abort_latency.py >> profiles/r04_abort_latency.txt"""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from lld_slam_amd import BABatch, Context, synth


def run(ctx, ws, label, reps=40):
    rng = np.random.default_rng(1)
    with BABatch(ctx, ws) as b:
        t0 = time.perf_counter(); b.solve(); full_ms = (time.perf_counter() - t0) * 1e3
        full = [sum(s["lm_trials"]) for s in b.stats()]
        lat = []; cut = []
        for _ in range(reps):
            flag = ctypes.c_int(0); t_raise = [0.0]
            delay = float(rng.uniform(0.15, 0.85)) * full_ms * 1e-3
            def raiser():
                time.sleep(delay); t_raise[0] = time.perf_counter(); flag.value = 1
            th = threading.Thread(target=raiser); th.start()
            b.solve_with_flag(flag); t_ret = time.perf_counter(); th.join()
            st = b.stats()
            if t_raise[0] > 0 and t_raise[0] < t_ret and any(s["aborted"] for s in st):
                lat.append((t_ret - t_raise[0]) * 1e3); cut.append(np.mean([sum(s["lm_trials"]) for s in st]) / np.mean(full))
        lat = np.array(lat)
        print(f"{label}: full solve {full_ms:.2f} ms, {len(lat)} of {reps} repeats raised in time; raise -> return ms: min {lat.min():.3f} median {np.median(lat):.3f} "
              f"p90 {np.quantile(lat, 0.9):.3f} max {lat.max():.3f}; trials done / full: median {np.median(cut):.2f}")


with Context(0) as ctx:
    run(ctx, [synth.make_lba_b(0)], "one LBA-B window (queued super-steps, fused point / line kernels)")
    run(ctx, [synth.make_lba_a(i) for i in range(8)], "8 LBA-A windows (two groups of 4)")
    run(ctx, [synth.make_lba_b(i) for i in range(64)], "64 LBA-B windows (groups of >= 24: one poll per super-step)", reps=12)
