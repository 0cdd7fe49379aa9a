"""Run-to-run spread of one small window against the oracle (atomics make the accumulation order vary): how far do the weakest
landmarks move, and does the LM trajectory ever fork?   python tools/exp_flake.py [n_free] [runs]"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, Optimizer, synth
import oracle_py as O
ctx = Context(0); O.lib()
def rel(a, b): return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)
n_free = int(sys.argv[1]) if len(sys.argv) > 1 else 16
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
w = synth.make_lba_small(40 + n_free, n_free=n_free, n_fixed=max(2, 7 - n_free), n_points=60 * n_free + 80, n_lines=8 * n_free + 10)
o = O.local_ba(w)
print("oracle stats", o.stats)
rows = []
for i in range(runs):
    g = Optimizer(ctx).LocalBundleAdjustment(w)
    rp = rel(g.pt_xyz, o.pt_xyz); rl = rel(g.line_x0, o.line_x0)
    rows.append((rp.max(), rl.max(), np.linalg.norm(g.line_dir - o.line_dir, axis=1).max(), abs(g.stats["chi2_final"] / o.stats["chi2_final"] - 1), sum(g.stats["lm_trials"]),
                 np.abs(g.cam_qt - o.cam_qt).max(), (rp > 1e-5).sum(), (rl > 1e-5).sum()))
    if rows[-1][0] > 1e-4 or rows[-1][1] > 1e-4 or rows[-1][2] > 1e-4:
        print("run", i, rows[-1], "stats", g.stats, "worst pt", int(rp.argmax()), "worst ln", int(rl.argmax()))
rows = np.array(rows)
for k, name in enumerate(["pt", "ln_x0", "ln_dir", "chi2", "trials", "cam", "n_pt_above_1e-5", "n_ln_above_1e-5"]):
    print(name, "max", rows[:, k].max(), "p99", np.quantile(rows[:, k], 0.99), "median", np.median(rows[:, k]), "min", rows[:, k].min())
