"""Experiment: does the ORDER of the landmarks inside a window matter to the linearisation kernels?  Their camera accumulators are LDS
atomics; a wavefront holds the edges of ~10 consecutive landmarks.  `stratified` deals the landmarks so that consecutive ones see
different cameras (sorted by first camera, then read column-wise from a G-row table), `sorted` puts equal camera sets next to each other.
   python tools/exp_stratified_order.py [n_windows=128]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lld_slam_amd import Context, BABatch, synth


def reorder(w, mode):
    def perm_of(start, cam, G):
        n = len(start) - 1
        key = np.array([cam[start[i]:start[i + 1]].min() if start[i + 1] > start[i] else 10 ** 6 for i in range(n)])
        order = np.argsort(key, kind="stable")
        if mode == "sorted":
            return order
        cols = (n + G - 1) // G
        idx = np.arange(G * cols).reshape(G, cols).T.reshape(-1)            # walk down the columns of a G x cols table
        return order[idx[idx < n]]
    def apply(start, perm, arrays):
        cnt = np.diff(start)[perm]
        new_start = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        src = np.concatenate([np.arange(start[p], start[p + 1]) for p in perm]) if len(perm) else np.zeros(0, int)
        return new_start, [a[src] for a in arrays]
    pp = perm_of(w.pt_obs_start, w.pt_obs_cam, 10)
    w.pt_xyz = w.pt_xyz[pp]
    w.pt_obs_start, (w.pt_obs_cam, w.pt_obs_uvr, w.pt_obs_inv_sigma2) = apply(w.pt_obs_start, pp, [w.pt_obs_cam, w.pt_obs_uvr, w.pt_obs_inv_sigma2])
    lp = perm_of(w.ln_obs_start, w.ln_obs_cam, 12)
    w.line_x0 = w.line_x0[lp]; w.line_dir = w.line_dir[lp]
    w.ln_obs_start, (w.ln_obs_cam, w.ln_obs_left, w.ln_obs_right, w.ln_obs_octave) = apply(w.ln_obs_start, lp, [w.ln_obs_cam, w.ln_obs_left, w.ln_obs_right, w.ln_obs_octave])
    return w.normalise()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = Context(0)
names = ["linearize", "schur", "solve", "backsub", "control"]
for mode in ("caller", "stratified", "sorted"):
    ws = [synth.make_lba_b(i) for i in range(n)]
    if mode != "caller":
        ws = [reorder(w, mode) for w in ws]
    with BABatch(ctx, ws) as b:
        b.solve()
        t = time.perf_counter(); b.solve(); dt = time.perf_counter() - t
        b.set_groups(1); b.set_phase_timing(True); b.solve()
        ph = b.phase_ms(); la = [b.kernel_stats(k)[0] for k in range(5)]
        st = b.stats()
    print(mode, "windows/s %.0f" % (n / dt), "chi2[0] %.9g" % st[0]["chi2_final"],
          "  ".join("%s %.0f" % (names[k], 1e3 * ph[k] / max(la[k], 1)) for k in range(5)))
