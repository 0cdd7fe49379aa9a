import ast, os, re, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
from lld_slam_amd import Context, Optimizer, synth, host
import oracle_py as O
log = open("profiles/r05_fuzz_ba_20000_seed99_final.txt").read().split("\n")
for line in log:
    m = re.match(r"^(MISMATCH|FLOOR)\s+(\d+) reduced_solver (\d+) ", line)
    if not m or int(m.group(2)) != 892: continue
    dicts = re.findall(r"\{[^{}]*\}", line)
    kw, par = ast.literal_eval(dicts[-2]), ast.literal_eval(dicts[-1])
w = synth.make_ba_window(**kw)
ctx = Context(0)
for r2 in (0, 1, 2, 3, 4, 6, 8, 10, 14, 20, 30):
    p = dict(par, its_round2=r2)
    if r2 == 0: p = dict(par, its_round2=1, abort_after_trials=0)
    o = O.local_ba(w, **p); g = Optimizer(ctx).LocalBundleAdjustment(w, **p)
    print("its_round2 %2d  oracle: its %s trials %s chi2_round1 %.12g chi2 %.12g | device: its %s trials %s chi2_round1 %.12g chi2 %.12g | cam %.1e" % (p["its_round2"], o.stats["lm_iterations"], o.stats["lm_trials"], o.stats["chi2_round1"], o.stats["chi2_final"],
          g.stats["lm_iterations"], g.stats["lm_trials"], g.stats["chi2_round1"], g.stats["chi2_final"], np.abs(g.cam_qt - o.cam_qt).max()), flush=True)
_, tr = O.local_ba_traced(w, **par)
print(np.array2string(tr, precision=9, max_line_width=200, formatter={"float_kind": lambda v: "%.9g" % v}))
