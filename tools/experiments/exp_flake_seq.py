"""Alternating window shapes through one context (buffer reuse between calls), every result held to the oracle.
python tools/exp_flake_seq.py [reps]"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, Optimizer, synth
import oracle_py as O
ctx = Context(0); O.lib()
def rel(a, b): return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cases = []
for n_free in [1, 2, 3, 5, 8, 11, 16, 27, 50]:
    w = synth.make_lba_small(40 + n_free, n_free=n_free, n_fixed=max(2, 7 - n_free), n_points=60 * n_free + 80, n_lines=8 * n_free + 10)
    cases.append((f"pad{n_free}", w, O.local_ba(w)))
w = synth.make_lba_small(5); cases.append(("small5", w, O.local_ba(w)))
w = synth.make_lba_a(1); cases.append(("lbaA", w, O.local_ba(w)))
rng = np.random.default_rng(0)
bad = 0
for r in range(reps):
    order = rng.permutation(len(cases))
    for ci in order:
        name, w, o = cases[ci]
        g = Optimizer(ctx).LocalBundleAdjustment(w)
        rp = rel(g.pt_xyz, o.pt_xyz); rl = rel(g.line_x0, o.line_x0)
        fp, fl = np.mean(rp <= 1e-5), np.mean(rl <= 1e-5)
        chi = abs(g.stats["chi2_final"] / o.stats["chi2_final"] - 1)
        same_out = np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier) and np.array_equal(g.ln_edge_outlier, o.ln_edge_outlier)
        if fp < 0.99 or fl < 0.99 or rp.max() > 1e-4 or rl.max() > 1e-4 or chi > 1e-5 or not same_out or g.stats["lm_trials"] != o.stats["lm_trials"]:
            bad += 1
            print(f"rep {r} {name}: frac pt {fp:.4f} ln {fl:.4f} max pt {rp.max():.2e} ln {rl.max():.2e} chi {chi:.2e} outliers same {same_out} trials {g.stats['lm_trials']} vs {o.stats['lm_trials']}"
                  f" cam {np.abs(g.cam_qt - o.cam_qt).max():.2e} n(pt>1e-5) {(rp > 1e-5).sum()} n(ln>1e-5) {(rl > 1e-5).sum()}", flush=True)
print("runs", reps * len(cases), "bad", bad)
