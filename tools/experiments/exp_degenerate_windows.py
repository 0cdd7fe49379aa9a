"""Structurally awkward windows, GPU vs oracle: a free camera without any observation (singular reduced system), points seen once,
all observations of a camera gross outliers, a window whose every camera is free, one landmark only.
   python tools/exp_degenerate_windows.py"""
import sys, copy, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, Optimizer, synth, host
import oracle_py as O
ctx = Context(0); O.lib()
def rel(a, b): return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)

def rebuild(w, keep_pt_obs=None, keep_ln_obs=None, **over):
    """copy of w with a subset of its observations"""
    d = {k: getattr(w, k) for k in ("cam", "n_free_cams", "cam_qt", "pt_xyz", "pt_obs_start", "pt_obs_cam", "pt_obs_uvr", "pt_obs_inv_sigma2", "line_x0", "line_dir",
                                    "ln_obs_start", "ln_obs_cam", "ln_obs_left", "ln_obs_right", "ln_obs_octave")}
    if keep_pt_obs is not None:
        cnt = np.add.reduceat(keep_pt_obs.astype(np.int64), d["pt_obs_start"][:-1]) if len(keep_pt_obs) else np.zeros(0, np.int64)
        cnt[np.diff(d["pt_obs_start"]) == 0] = 0
        d["pt_obs_start"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        for k in ("pt_obs_cam", "pt_obs_inv_sigma2"): d[k] = d[k][keep_pt_obs]
        d["pt_obs_uvr"] = d["pt_obs_uvr"].reshape(-1, 3)[keep_pt_obs]
    if keep_ln_obs is not None:
        cnt = np.add.reduceat(keep_ln_obs.astype(np.int64), d["ln_obs_start"][:-1])
        cnt[np.diff(d["ln_obs_start"]) == 0] = 0
        d["ln_obs_start"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        d["ln_obs_cam"] = d["ln_obs_cam"][keep_ln_obs]
        d["ln_obs_left"] = d["ln_obs_left"].reshape(-1, 4)[keep_ln_obs]; d["ln_obs_right"] = d["ln_obs_right"].reshape(-1, 4)[keep_ln_obs]
        d["ln_obs_octave"] = d["ln_obs_octave"].reshape(-1, 2)[keep_ln_obs]
    d.update(over)
    return host.Window(**d)

def check(name, w, **kw):
    try:
        o = O.local_ba(w, **kw)
    except Exception as e:
        print(f"{name:42s} oracle raised {e!r}"); o = None
    try:
        g = Optimizer(ctx).LocalBundleAdjustment(w, **kw)
    except Exception as e:
        print(f"{name:42s} gpu raised {e!r}"); return
    if o is None: return
    same = np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier) and np.array_equal(g.ln_edge_outlier, o.ln_edge_outlier) and np.array_equal(g.line_removed, o.line_removed)
    fin = np.isfinite(g.cam_qt).all() and np.isfinite(g.pt_xyz).all()
    print(f"{name:42s} chi2 gpu {g.stats['chi2_final']:.6g} oracle {o.stats['chi2_final']:.6g} trials {g.stats['lm_trials']} / {o.stats['lm_trials']} aborted {g.stats['aborted']}/{o.stats['aborted']}"
          f" outliers same {same} finite {fin} cam {np.abs(g.cam_qt - o.cam_qt).max():.1e} pt {rel(g.pt_xyz, o.pt_xyz).max() if w.n_points else 0:.1e}")

w = synth.make_lba_small(3)
check("plain", w)
# 1. free camera 2 without observations
check("free camera without observations", rebuild(w, keep_pt_obs=w.pt_obs_cam != 2, keep_ln_obs=w.ln_obs_cam != 2))
# 2. every point seen once
first = np.zeros(w.n_pt_obs, bool); first[w.pt_obs_start[:-1][np.diff(w.pt_obs_start) > 0]] = True
check("every point seen once", rebuild(w, keep_pt_obs=first))
# 3. one camera's observations all gross outliers
uvr = w.pt_obs_uvr.reshape(-1, 3).copy(); sel = w.pt_obs_cam == 1; uvr[sel, :2] += 300.0
check("camera 1: every point observation off by 300 px", rebuild(w, pt_obs_uvr=uvr))
# 4. all cameras free (gauge freedom: nothing fixes the frame)
check("all cameras free", rebuild(w, n_free_cams=w.n_cams))
# 5. one point, no lines
w1 = synth.make_lba_small(4, n_points=1, n_lines=0)
check("one point, no lines", w1)
# 6. lines only
check("lines only", synth.make_lba_small(5, n_points=0))
# 7. points only, mono observations only
uvr = w.pt_obs_uvr.reshape(-1, 3).copy(); uvr[:, 2] = -1.0
check("mono points only", rebuild(w, pt_obs_uvr=uvr, keep_ln_obs=np.zeros(w.n_ln_obs, bool)))
# 8. a point behind its cameras at the start
X = w.pt_xyz.copy(); X[:5] = -X[:5]
check("five points behind the cameras", rebuild(w, pt_xyz=X))
# 9. identical duplicate observation of a point (same camera twice)
check("gamma 0.1", w, gamma=0.1)
