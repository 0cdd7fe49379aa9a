"""Per-family HIP-event times of the batched local BA on N resident LBA-B windows, one window group (disjoint event brackets).
   python tools/time_lba_phases.py [n_windows=128]      (LLD_AMD_LIB=<path> selects an experimental build of the library)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lld_slam_amd import Context, BABatch, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = Context(0)
ws = [synth.make_lba_b(i) for i in range(n)]
with BABatch(ctx, ws) as b:
    b.solve()
    t = time.perf_counter(); b.solve(); dt = time.perf_counter() - t
    b.set_groups(1); b.set_phase_timing(True); b.solve()
    ph = b.phase_ms(); la = [b.kernel_stats(k)[0] for k in range(5)]
    st = b.stats()
names = ["linearize", "schur", "solve", "backsub", "control"]
print("lib", os.environ.get("LLD_AMD_LIB", "default"), "windows", n, "windows/s %.0f" % (n / dt), "chi2[0] %.9g" % st[0]["chi2_final"],
      "trials %.2f" % np.mean([sum(s["lm_trials"]) for s in st]))
print("  ".join("%s %.0f us/launch (%d)" % (names[k], 1e3 * ph[k] / max(la[k], 1), la[k]) for k in range(5)), " total ms %.2f" % ph[5])
