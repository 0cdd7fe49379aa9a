import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, Optimizer, synth
import oracle_py as O
from test_gpu_ba_structures import CASES
ctx = Context(0); O.lib()
for name, w, kw in CASES:
    o = O.local_ba(w, **kw)
    mc = mp = 0.0; forks = 0
    for i in range(300):
        g = Optimizer(ctx).LocalBundleAdjustment(w, **kw)
        mc = max(mc, (np.abs(g.cam_qt - o.cam_qt) / (1e-7 + 1e-5 * np.abs(o.cam_qt))).max())
        if w.n_points: mp = max(mp, (np.linalg.norm(g.pt_xyz - o.pt_xyz, axis=1) / np.maximum(np.linalg.norm(o.pt_xyz, axis=1), 1e-3)).max() / 1e-4)
        if w.n_lines: mp = max(mp, (np.linalg.norm(g.line_x0 - o.line_x0, axis=1) / np.maximum(np.linalg.norm(o.line_x0, axis=1), 1e-3)).max() / 1e-4)
        forks += g.stats["lm_trials"] != o.stats["lm_trials"] or not np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier) or not np.array_equal(g.ln_edge_outlier, o.ln_edge_outlier)
    print(f"{name:40s} worst cam ratio (1 = default bar) {mc:.3g}  landmark ratio {mp:.3g}  forks {forks}")
