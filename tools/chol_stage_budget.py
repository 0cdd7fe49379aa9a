#!/usr/bin/env python3
"""Per-stage time budget of ba_chol_mfma_kernel ("commit a per-stage cycle budget before coding").

Runs on the GPU box against the EXPERIMENTS build (liblld_amd_exp.so, LLD_BA_CHOL_STAMPS=1): every wavefront of the kernel leaves
s_memtime stamps at its stage boundaries; this prints, for the last launch of window 0, the time of each stage per tile column for the
panel wavefront and for the slowest tile wavefront, and the totals.  An s_memtime tick is a shader cycle on gfx950 (MI355X_MICROARCH.md); the
table is in nanoseconds at 2.4 GHz (LLD_TICK_NS overrides), cycles = ns x 2.4, and the kernel's stamp-to-stamp time is printed next to its
HIP-event time so that the clock assumption can be checked.

    python tools/chol_stage_budget.py [n_windows=1] > profiles/r04_chol_stage_budget.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LLD_BA_CHOL_STAMPS"] = "1"
os.environ.setdefault("LLD_BA_CHOL_FORCE", "3")          # the dense kernel (round 5: windows with a plan go to ba_chol_sparse_kernel by default)

import numpy as np

from lld_slam_amd import BABatch, Context, abi, synth

SLOTS = 256


def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    lib = abi.Lib(os.path.join(ROOT, "lld_slam_amd", "csrc", "liblld_amd_exp.so"), "lld_")
    ws = [synth.make_lba_b(i) for i in range(nw)]
    with Context(0, lib=lib) as ctx, BABatch(ctx, ws) as b:
        b.set_phase_timing(True); b.solve(); b.solve()
        st = np.zeros((nw, 16, SLOTS), dtype=np.int64)
        fn = lib.dll.lld_exp_chol_stamps
        fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        assert fn(b.handle, st.ctypes.data) == 0
        ph = b.phase_ms(); n_solve, ms_solve = b.kernel_stats(2)
    tick_ns = float(os.environ.get("LLD_TICK_NS", str(1 / 2.4)))       # s_memtime tick = shader cycle
    st = st[:, :8]
    s = st[0].astype(np.float64) * tick_ns
    t0 = s[:, 0].min()
    s = np.where(st[0] != 0, s - t0, np.nan)
    NT = 19
    P = s[0]; T = s[1:]
    print(f"ba_chol_mfma_kernel stage budget, window 0 of a batch of {nw} LBA-B windows (n = 300, 19 tile columns); times in ns since the first wavefront's entry")
    print(f"HIP-event time of the reduced-solve phase: {ms_solve / max(n_solve, 1) * 1e3:.1f} us per launch ({n_solve} launches)")
    print(f"kernel (stamp 0 -> 6, slowest wavefront): {np.nanmax(s[:, 6]) / 1e3:.1f} us")
    print(f"  load S -> registers (tile waves, slowest): {np.nanmax(T[:, 1]) / 1e3:.2f} us      prologue publish: {(np.nanmax(T[:, 2]) - np.nanmax(T[:, 1])) / 1e3:.2f} us")
    print(f"  panel: factor of diagonal tile 0: {(P[3] - P[2]) / 1e3:.2f} us")
    print(f"  all tile columns (panel stamp 3 -> 4): {(P[4] - P[3]) / 1e3:.2f} us")
    print(f"  back substitution (panel 4 -> 5): {(P[5] - P[4]) / 1e3:.2f} us      epilogue (5 -> 6): {(np.nanmax(s[:, 6]) - P[5]) / 1e3:.2f} us")
    print()
    which = "ba_chol_mfma_kernel"
    print(f"{which}.  per tile column J (ns):  panel wave: own L_(J+1)J + diag J+1 update | wait(c) | y_(J+1) update | factor J+1 | rest of y + wait(d)      tile waves (slowest): L_IJ | wait | trailing update | wait      column total")
    tot = np.zeros(9)
    for J in range(NT):
        b0 = 8 + 6 * J
        pw = [P[b0 + 1] - P[b0], P[b0 + 2] - P[b0 + 1], P[b0 + 3] - P[b0 + 2], P[b0 + 4] - P[b0 + 3], P[b0 + 5] - P[b0 + 4]]
        if J == NT - 1:
            pw[2] = 0.0; pw[3] = P[b0 + 4] - P[b0 + 2]
        tw = [np.nanmax(T[:, b0 + 1] - T[:, b0]), np.nanmin(T[:, b0 + 2] - T[:, b0 + 1]), np.nanmax(T[:, b0 + 4] - T[:, b0 + 2]), np.nanmin(T[:, b0 + 5] - T[:, b0 + 4])]
        col = P[b0 + 5] - P[b0]
        row = pw + tw
        tot[:9] += np.nan_to_num(np.array(row))
        print(f"  J={J:2d}  panel {pw[0]:7.0f} {pw[1]:7.0f} {pw[2]:7.0f} {pw[3]:7.0f} {pw[4]:7.0f}     tiles {tw[0]:7.0f} {tw[1]:7.0f} {tw[2]:7.0f} {tw[3]:7.0f}     column {col:7.0f}")
    print(f"  sum   panel {tot[0]:7.0f} {tot[1]:7.0f} {tot[2]:7.0f} {tot[3]:7.0f} {tot[4]:7.0f}     tiles {tot[5]:7.0f} {tot[6]:7.0f} {tot[7]:7.0f} {tot[8]:7.0f}")
    print()
    print("phase_ms of the last solve (1 stream group if the batch is small):", np.round(ph, 3).tolist())


if __name__ == "__main__":
    main()
