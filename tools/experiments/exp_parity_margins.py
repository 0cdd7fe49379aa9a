#!/usr/bin/env python3
"""How close the device sits to the oracle on the windows of tests/test_gpu_ba.py, default and deterministic mode: the measured basis of
check_ba's landmark bar ("tighten check_ba to what the deterministic path actually achieves").
    python tools/exp_parity_margins.py > profiles/r04_parity_margins.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as O
from lld_slam_amd import Context, Optimizer, synth


def rel(a, b):
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)


def main():
    cases = [("small0", synth.make_lba_small(0)), ("small1", synth.make_lba_small(1, mono_frac=0.15, mono_line_frac=0.2)),
             ("small4", synth.make_lba_small(4, n_free=10, n_fixed=3, n_points=700, n_lines=120, outlier_frac=0.15)),
             ("small5", synth.make_lba_small(5, n_free=5, n_fixed=0, n_points=200, n_lines=30)),
             ("s9", synth.make_lba_small(9, n_free=12, n_fixed=3, n_points=500, n_lines=80))]
    cases += [(f"pad{nf}", synth.make_lba_small(40 + nf, n_free=nf, n_fixed=max(2, 7 - nf), n_points=60 * nf + 80, n_lines=8 * nf + 10)) for nf in (1, 3, 8, 16, 27, 50)]
    cases += [(f"lba_a{i}", synth.make_lba_a(i)) for i in range(4)] + [(f"lba_b{i}", synth.make_lba_b(i)) for i in (0, 64, 192, 255)]
    print("window            mode   chi2 rel      cam max abs   pts: max rel / median / beyond 1e-5      lines x0: max rel / beyond 1e-5     dir max")
    with Context(0) as ctx:
        opt = Optimizer(ctx)
        for name, w in cases:
            o = O.local_ba(w)
            for det in (0, 1):
                g = opt.LocalBundleAdjustment(w, deterministic=det)
                same = np.array_equal(g.pt_obs_outlier, o.pt_obs_outlier) and np.array_equal(g.ln_edge_outlier, o.ln_edge_outlier) and np.array_equal(g.line_removed, o.line_removed)
                rp = rel(g.pt_xyz, o.pt_xyz) if w.n_points else np.zeros(1)
                rl = rel(g.line_x0, o.line_x0) if w.n_lines else np.zeros(1)
                dd = np.linalg.norm(g.line_dir - o.line_dir, axis=1).max() if w.n_lines else 0.0
                print(f"{name:16s}  {'det' if det else 'dflt'}   {abs(g.stats['chi2_final'] / o.stats['chi2_final'] - 1):.2e}   {np.abs(g.cam_qt - o.cam_qt).max():.2e}     "
                      f"{rp.max():.2e} / {np.median(rp):.2e} / {int((rp > 1e-5).sum()):4d} of {rp.size:5d}      {rl.max():.2e} / {int((rl > 1e-5).sum()):3d} of {rl.size:4d}    {dd:.2e}   sets {'equal' if same else 'DIFFER'}"
                      f"   its {g.stats['lm_iterations']} vs {o.stats['lm_iterations']}")


if __name__ == "__main__":
    main()
