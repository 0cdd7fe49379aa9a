"""Run-to-run spread of a resident batch: 256 LBA-B windows solved N times from the same uploaded state.  Reports, per repeat against the first
solve, the largest relative difference of chi2_final, how many windows exceed 1e-5 / 1e-4, and how many windows change an outlier set -
the quantities tests/test_gpu_ba.py::test_batch_config_256_lba_b_windows bounds.   python tools/exp_restart_noise.py [repeats=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lld_slam_amd import Context, BABatch, synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ws = synth.generate_windows(0, 256)
    ctx = Context(0)
    worst = 0.0
    with BABatch(ctx, ws) as b:
        b.solve()
        first = [b.download(i) for i in range(256)]
        for rep in range(n):
            b.solve()
            rel = np.zeros(256); sets = 0; cam = 0.0
            for i in range(256):
                c = b.download(i)
                rel[i] = abs(c.stats["chi2_final"] - first[i].stats["chi2_final"]) / first[i].stats["chi2_final"]
                d = (int((c.pt_obs_outlier != first[i].pt_obs_outlier).sum()), int((c.line_removed != first[i].line_removed).sum()),
                     int((c.ln_edge_outlier != first[i].ln_edge_outlier).sum()))
                if any(d):
                    sets += 1
                    print("    window %3d differs: point flags %d, removed lines %d, line-edge flags %d, trials %s vs %s, rel chi2 %.2e"
                          % (i, d[0], d[1], d[2], c.stats["lm_trials"], first[i].stats["lm_trials"], rel[i]))
                cam = max(cam, float(np.abs(c.cam_qt - first[i].cam_qt).max()))
            worst = max(worst, rel.max())
            print("repeat %2d  max rel chi2 %.2e (window %3d)  > 1e-5: %d  > 1e-4: %d  windows with a changed set: %d  max |cam| %.1e"
                  % (rep, rel.max(), int(rel.argmax()), int((rel > 1e-5).sum()), int((rel > 1e-4).sum()), sets, cam), flush=True)
    print("worst over %d repeats: %.2e" % (n, worst))


if __name__ == "__main__":      # synth.generate_windows spawns worker processes that re-import this file
    main()
