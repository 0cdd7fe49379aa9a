"""What non-finite input does: device vs oracle statistics for a small window with one NaN / Inf planted.   python tools/exp_non_finite.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from lld_slam_amd import Context, Optimizer, synth
import oracle_py as O
ctx = Context(0)
clean = synth.make_lba_small(21)
for what in ("nan_observation", "inf_point", "nan_pose", "nan_line_endpoint"):
    bad = copy.deepcopy(clean)
    if what == "nan_observation": bad.pt_obs_uvr = bad.pt_obs_uvr.copy(); bad.pt_obs_uvr[17, 0] = np.nan
    elif what == "inf_point": bad.pt_xyz = bad.pt_xyz.copy(); bad.pt_xyz[5, 2] = np.inf
    elif what == "nan_pose": bad.cam_qt = bad.cam_qt.copy(); bad.cam_qt[1, 5] = np.nan
    else: bad.ln_obs_left = bad.ln_obs_left.copy(); bad.ln_obs_left[3, 1] = np.nan
    g = Optimizer(ctx).LocalBundleAdjustment(bad); o = O.local_ba(bad)
    for tag, r in (("gpu", g), ("oracle", o)):
        print(what, tag, "its", r.stats["lm_iterations"], "trials", r.stats["lm_trials"], "chi2 %.6g -> %.6g" % (r.stats["chi2_round1"], r.stats["chi2_final"]),
              "outliers", int(r.pt_obs_outlier.sum()), "nan poses", int(np.isnan(r.cam_qt).any(axis=1).sum()), "nan points", int(np.isnan(r.pt_xyz).any(axis=1).sum()), flush=True)
