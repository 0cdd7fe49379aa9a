"""One fuzz window many times: how often does the device differ from the oracle, in what, and do the oracle's rounding twins differ the same way?
   python tools/exp_fuzz_window.py ["<window kwargs dict>" "<parameter dict>"]   (both as printed by tools/fuzz_ba.py; default: the PCG window of round 3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from lld_slam_amd import Context, Optimizer, synth, host
import oracle_py as O
kw = {'n_free': 36, 'n_fixed': 4, 'n_points': 149, 'obs_per_point': 5, 'n_lines': 0, 'obs_per_line': 5, 'seed': 1424471088, 'outlier_frac': 0.05, 'mono_frac': 0.3, 'mono_line_frac': 0.0, 'noise': 0.5,
      'pose_sigma': (0.3965166224649508, 0.05875704279060679), 'point_sigma': 0.29794398544307243}
par = {'gamma': 1.0, 'its_round1': 1, 'its_round2': 17}
if len(sys.argv) > 2:
    import ast
    kw, par = ast.literal_eval(sys.argv[1]), ast.literal_eval(sys.argv[2])
w = synth.make_ba_window(**kw)
o = O.local_ba(w, **par)
def rel(a, b): return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)
def show(tag, g):
    print(tag, "trials", g.stats["lm_trials"], "its", g.stats["lm_iterations"], "chi2 %.9g" % g.stats["chi2_final"], "outliers", int(g.pt_obs_outlier.sum()),
          "sets equal", np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier), "n diff", int((g.pt_obs_outlier != o.pt_obs_outlier).sum()),
          "cam %.1e pt %.1e" % (np.abs(g.cam_qt - o.cam_qt).max(), rel(g.pt_xyz, o.pt_xyz).max()), flush=True)
show("oracle      ", o)
O.set_landmark_inverse(1); show("oracle chol ", O.local_ba(w, **par)); O.set_landmark_inverse(0)
show("oracle fma  ", host.ba_call(O.lib_fma(), None, w, host.ba_params(O.lib_fma(), **par)))
_, tr = O.local_ba_traced(w, **par)
print("oracle trace (lambda, chi2, accepted):"); print(np.array2string(tr, precision=6, max_line_width=200, formatter={"float_kind": lambda v: "%.6g" % v}))
ctx = Context(0)
from collections import Counter
for solver in (0, 1, 2):
    c = Counter()
    for i in range(60):
        g = Optimizer(ctx).LocalBundleAdjustment(w, reduced_solver=solver, **par)
        key = (tuple(g.stats["lm_trials"]), bool(np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier)), "cam<=1e-5" if np.abs(g.cam_qt - o.cam_qt).max() <= 1e-5 else "cam %.0e" % np.abs(g.cam_qt - o.cam_qt).max())
        c[key] += 1
    print("solver", solver, dict(c), flush=True)
