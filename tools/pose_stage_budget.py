#!/usr/bin/env python3
"""Per-stage time budget of pose_opt_kernel for ONE frame (config PO: 1000 points + 200 stereo lines), round 6.

Runs on the GPU box against the EXPERIMENTS build (liblld_amd_exp.so: lld_pose.hip under -DLLD_EXPERIMENTS sums s_memtime differences of the
frame's first wavefront per stage).  Ticks are taken as 2.4 GHz core clocks unless LLD_TICK_NS says otherwise; the kernel total is printed
next to the wall clock of lld_pose_opt so the unit can be checked.

    python tools/pose_stage_budget.py [frame_id=0] > profiles/r06_pose_stage_budget.txt
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from lld_slam_amd import Context, Optimizer, abi, synth
from lld_slam_amd import host

NAMES = ["stage frame into LDS", "linearisation sweep", "28-value reduction + barrier", "16-candidate solve + oplus", "trial sweep", "trial sum + barrier",
         "classification", "round head / tail", "LM iterations", "LM trials", "candidate solves", "kernel total"]


def main():
    fid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    lib = abi.Lib(os.path.join(ROOT, "lld_slam_amd", "csrc", "liblld_amd_exp.so"), "lld_")
    tick_ns = float(os.environ.get("LLD_TICK_NS", str(1 / 2.4)))
    f = synth.make_pose_frame(fid)
    with Context(0, lib=lib) as ctx:
        opt = Optimizer(ctx)
        for _ in range(3): out = opt.PoseOptimization(f, 0.5)
        t = []
        for _ in range(20):
            t0 = time.perf_counter(); opt.PoseOptimization(f, 0.5); t.append(time.perf_counter() - t0)
        st = np.zeros(16, np.int64)
        fn = lib.dll.lld_exp_pose_stamps
        fn.restype = C.c_int
        cf = f.to_c()
        cr, po, lo = host.pose_result_alloc(f.n_points, f.n_lines)
        prm = host.pose_params(lib, 0.5)
        rc = fn(ctx.handle, C.byref(cf), C.byref(prm), C.byref(cr), st.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
    print(f"pose_opt_kernel stage budget, frame {fid}: {f.pt_xw.shape[0]} points, {f.ln_x0.shape[0]} lines; one 512-lane workgroup; times of wavefront 0, tick = {tick_ns:.4f} ns")
    print(f"lld_pose_opt wall clock (host, pack + H2D + kernel + D2H): min {min(t) * 1e6:.1f} us, median {np.median(t) * 1e6:.1f} us")
    n_it, n_tr, n_sol = int(st[8]), int(st[9]), int(st[10])
    tot = st[11] * tick_ns / 1e3
    print(f"LM iterations {n_it}, trials {n_tr}, candidate solves {n_sol}; kernel total {tot:.1f} us")
    for k in range(8):
        us = st[k] * tick_ns / 1e3
        per = ""
        if k in (1, 2): per = f"   {us / max(n_it, 1):.2f} us per iteration"
        if k == 3: per = f"   {us / max(n_sol, 1):.2f} us per solve"
        if k in (4, 5): per = f"   {us / max(n_tr, 1):.2f} us per trial"
        print(f"  {NAMES[k]:32s} {us:8.1f} us  {100 * us / max(tot, 1e-9):5.1f} %{per}")
    print(f"  {'sum of the stages':32s} {st[:8].sum() * tick_ns / 1e3:8.1f} us")


if __name__ == "__main__":
    main()
