"""Re-run given fuzz_ba windows on the library named by LLD_AMD_LIB over several reduced-solver modes and compare with the oracle and its twins.
   python tools/experiments/exp_fuzz_cases.py <fuzz log> [case numbers ...]      (the MISMATCH / FLOOR lines of tools/fuzz_ba.py carry the window's spec)"""
import ast, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np
from lld_slam_amd import Context, Optimizer, synth, host
import oracle_py as O

def rel(a, b): return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)
log = open(sys.argv[1]).read().split("\n")
want = set(int(a) for a in sys.argv[2:])
ctx = Context(0)
for line in log:
    m = re.match(r"^(MISMATCH|FLOOR)\s+(\d+) reduced_solver (\d+) ", line)
    if not m or (want and int(m.group(2)) not in want): continue
    dicts = re.findall(r"\{[^{}]*\}", line)
    kw, par = ast.literal_eval(dicts[-2]), ast.literal_eval(dicts[-1])
    w = synth.make_ba_window(**kw)
    o = O.local_ba(w, **par)
    O.set_landmark_inverse(1); o2 = O.local_ba(w, **par); O.set_landmark_inverse(0)
    o3 = host.ba_call(O.lib_fma(), None, w, host.ba_params(O.lib_fma(), **par))
    print("case", m.group(2), "fuzzed with solver", m.group(3), kw, par)
    for tag, g in (("oracle chol-inverse twin", o2), ("oracle fma twin", o3)):
        print("   %-26s trials %s chi2 %.12g  cam %.1e pt %.1e ln %.1e" % (tag, g.stats["lm_trials"], g.stats["chi2_final"], np.abs(g.cam_qt - o.cam_qt).max(), rel(g.pt_xyz, o.pt_xyz).max() if w.n_points else 0, rel(g.line_x0, o.line_x0).max() if w.n_lines else 0))
    print("   %-26s trials %s chi2 %.12g round1 %.6g" % ("oracle", o.stats["lm_trials"], o.stats["chi2_final"], o.stats["chi2_round1"]))
    for solver in (0, 3, 2, 1):
        g = Optimizer(ctx).LocalBundleAdjustment(w, reduced_solver=solver, **par)
        print("   %-26s trials %s chi2 %.12g  cam %.1e pt %.1e ln %.1e  sets equal %s" % ("device, reduced_solver %d" % solver, g.stats["lm_trials"], g.stats["chi2_final"], np.abs(g.cam_qt - o.cam_qt).max(),
              rel(g.pt_xyz, o.pt_xyz).max() if w.n_points else 0, rel(g.line_x0, o.line_x0).max() if w.n_lines else 0, np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier)), flush=True)
