import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import oracle_orbsearch as OS
from lld_slam_amd import Context, ORBmatcher, synth
ctx = Context(0)
seed, n, nq, th = 0, 2000, 1500, 1.0
F = synth.make_orb_frame(seed, n)
q = synth.make_projection_queries(F, seed, nq, dup_frac=0.3)
out = ORBmatcher(ctx, 0.8).SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], th)
n_exp, slot = OS.search_by_projection_map(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], th, 0.8)
print("gpu", out.n_matches, "exp", n_exp, "rounds", out.rounds)
# expected per-query match from slot
exp_match = -np.ones(nq, np.int32)
for k, s in enumerate(slot):
    if 0 <= s < nq: exp_match[s] = k
gm = out.match.copy()
diff = np.nonzero((gm >= 0) != (exp_match >= 0))[0]
print("queries with differing matched-ness:", len(diff), diff[:20])
for qi in diff[:8]:
    print(qi, "gpu", gm[qi], out.best_dist[qi], out.second_dist[qi], "exp", exp_match[qi], "obs", q["obs"][qi], "valid", q["valid"][qi])
d2 = np.nonzero((gm >= 0) & (exp_match >= 0) & (gm != exp_match))[0]
print("both matched but differ (may be overwritten):", len(d2))
