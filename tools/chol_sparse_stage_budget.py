#!/usr/bin/env python3
"""Per-stage time budget of ba_chol_sparse_kernel (round 5), the structure-following reduced solve.

Runs on the GPU box against the EXPERIMENTS build (liblld_amd_exp.so, LLD_BA_CHOL_STAMPS=1): every wavefront leaves s_memtime stamps at
its stage boundaries; this prints, for the last launch of window 0, the stages of every step for the two panel wavefronts and for the
slowest tile wavefront.  Times in ns at 2.4 GHz (LLD_TICK_NS overrides).

    python tools/chol_sparse_stage_budget.py [n_windows=1] > profiles/r05_chol_sparse_stage_budget_1window.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["LLD_BA_CHOL_STAMPS"] = "1"

import numpy as np

from lld_slam_amd import BABatch, Context, abi, synth

SLOTS = 256


def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    lib = abi.Lib(os.path.join(ROOT, "lld_slam_amd", "csrc", "liblld_amd_exp.so"), "lld_")
    ws = [synth.make_lba_b(i) for i in range(nw)]
    import test_chol_plan as TP
    plan = TP.get_plan(TP.window_pattern(ws[0]), int(os.environ.get("LLD_BA_CHOL_FORCE", "0")))
    with Context(0, lib=lib) as ctx, BABatch(ctx, ws) as b:
        b.set_phase_timing(True); b.solve(); b.solve()
        st = np.zeros((nw, 16, SLOTS), dtype=np.int64)
        fn = lib.dll.lld_exp_chol_stamps
        fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        assert fn(b.handle, st.ctypes.data) == 0
        ph = b.phase_ms(); n_solve, ms_solve = b.kernel_stats(2)
    tick_ns = float(os.environ.get("LLD_TICK_NS", str(1 / 2.4)))
    st = st[:, :12]                                             # 2 panel + 10 tile wavefronts; stamps are the low 32 bits of s_memtime
    s = st[0].astype(np.float64) * tick_ns
    t0 = s[:, 0][st[0][:, 0] != 0].min()
    s = np.where(st[0] != 0, s - t0, np.nan)
    T = int(plan["T"])
    PA, PB, TW = s[0], s[1], s[2:]
    print(f"ba_chol_sparse_kernel stage budget, window 0 of a batch of {nw} LBA-B windows; plan: {int(plan['chains'])} chain(s), {int(plan['NT'])} tile rows, {T} steps, "
          f"{int(plan['n_tiles'])} tiles, {int(plan['n_updates'])} tile updates; times in ns since the first wavefront's entry")
    print(f"HIP-event time of the reduced-solve phase: {ms_solve / max(n_solve, 1) * 1e3:.1f} us per launch ({n_solve} launches)")
    print(f"kernel (stamp 0 -> 6, slowest wavefront): {np.nanmax(s[:, 6]) / 1e3:.1f} us")
    print(f"  plan -> LDS + S -> registers (tile waves, slowest, stamp 1): {np.nanmax(TW[:, 1]) / 1e3:.2f} us   panel A's own first factor done at {PA[1] / 1e3:.2f} us   prologue publish done at {np.nanmax(TW[:, 2]) / 1e3:.2f} us")
    print(f"  all steps (panel A stamp 3 -> 4): {(PA[4] - PA[3]) / 1e3:.2f} us    back substitution (4 -> 5): {(PA[5] - PA[4]) / 1e3:.2f} us    epilogue (5 -> 6): {(np.nanmax(s[:, 6]) - PA[5]) / 1e3:.2f} us")
    print("  SIMD of the wavefronts 0..11 (HW_ID bits 5:4):", [int(st[0][w, 7]) >> 4 & 3 for w in range(12)], "  slots used per tile wavefront:", [int((plan["slotI"][w] != 255).sum()) for w in range(10)])
    print()
    print("per step (ns):  columns | panel A: own L_(Jn)J + diag update | wait(c) | factor | wait(d)   panel B: same   tile waves (slowest): L_IJ + y_J | wait | fetch + updates | publish + y | wait    step total")
    for k in range(T):
        b0 = 8 + 6 * k
        def pan(P):
            return [P[b0 + 1] - P[b0], P[b0 + 2] - P[b0 + 1], P[b0 + 4] - P[b0 + 2], P[b0 + 5] - P[b0 + 4]]
        a, bq = pan(PA), pan(PB)
        tw = [np.nanmax(TW[:, b0 + 1] - TW[:, b0]), np.nanmin(TW[:, b0 + 2] - TW[:, b0 + 1]), np.nanmax(TW[:, b0 + 3] - TW[:, b0 + 2]), np.nanmax(TW[:, b0 + 4] - TW[:, b0 + 3]), np.nanmin(TW[:, b0 + 5] - TW[:, b0 + 4])]
        cols = tuple(int(c) for c in plan["cols"][k])
        print(f"  s={k:2d} {str(cols):10s} A {a[0]:6.0f} {a[1]:6.0f} {a[2]:6.0f} {a[3]:6.0f}   B {bq[0]:6.0f} {bq[1]:6.0f} {bq[2]:6.0f} {bq[3]:6.0f}   tiles {tw[0]:6.0f} {tw[1]:6.0f} {tw[2]:6.0f} {tw[3]:6.0f} {tw[4]:6.0f}   step {PA[b0 + 5] - PA[b0]:6.0f}")
    print()
    print("phase_ms of the last solve:", np.round(ph, 3).tolist())


if __name__ == "__main__":
    main()
