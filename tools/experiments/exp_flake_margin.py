"""How close do repeated solves come to the parity bar of tests/test_gpu_ba.py::check_ba?  Prints, per window, the worst ratio
deviation / allowed over `runs` solves (1.0 = the assertion would fire).   python tools/exp_flake_margin.py [runs]"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, Optimizer, synth
import oracle_py as O
ctx = Context(0); O.lib()
def rel(a, b): return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cases = []
for n_free in [1, 2, 3, 5, 8, 11, 16, 27, 50]:
    cases.append((f"pad{n_free}", synth.make_lba_small(40 + n_free, n_free=n_free, n_fixed=max(2, 7 - n_free), n_points=60 * n_free + 80, n_lines=8 * n_free + 10)))
for s in range(8): cases.append((f"small{s}", synth.make_lba_small(s)))
cases.append(("lbaA", synth.make_lba_a(1)))
for name, w in cases:
    o = O.local_ba(w)
    worst = dict(cam=0.0, chi=0.0, pt_max=0.0, ln_max=0.0, dir=0.0, n_pt=0, n_ln=0, med_pt=0.0, med_ln=0.0); forks = 0
    for i in range(runs if w.n_points < 5000 else max(20, runs // 10)):
        g = Optimizer(ctx).LocalBundleAdjustment(w)
        worst["cam"] = max(worst["cam"], (np.abs(g.cam_qt - o.cam_qt) / (1e-7 + 1e-5 * np.abs(o.cam_qt))).max())
        worst["chi"] = max(worst["chi"], abs(g.stats["chi2_final"] - o.stats["chi2_final"]) / (1e-5 * o.stats["chi2_final"]))
        rp = rel(g.pt_xyz, o.pt_xyz); worst["pt_max"] = max(worst["pt_max"], rp.max() / 1e-4); worst["n_pt"] = max(worst["n_pt"], int((rp > 1e-5).sum()))
        worst["med_pt"] = max(worst["med_pt"], np.median(rp) / 1e-5)
        if w.n_lines:
            rl = rel(g.line_x0, o.line_x0); worst["ln_max"] = max(worst["ln_max"], rl.max() / 1e-4); worst["n_ln"] = max(worst["n_ln"], int((rl > 1e-5).sum()))
            worst["dir"] = max(worst["dir"], np.linalg.norm(g.line_dir - o.line_dir, axis=1).max() / 1e-4); worst["med_ln"] = max(worst["med_ln"], np.median(rl) / 1e-5)
        same = np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier) and np.array_equal(g.ln_edge_outlier, o.ln_edge_outlier) and np.array_equal(g.line_removed, o.line_removed)
        forks += (not same)
    print(name, f"n_pt {w.n_points} n_ln {w.n_lines}", " ".join(f"{k} {v:.3g}" for k, v in worst.items()), "outlier-set forks", forks, flush=True)
