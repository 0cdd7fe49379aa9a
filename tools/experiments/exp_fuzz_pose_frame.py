"""One frame of a tools/fuzz_pose.py campaign again (same generator state), round by round: device vs oracle after 1, 2, 3, 4 rounds of
PoseOptimization - where does a reported difference enter?   python tools/exp_fuzz_pose_frame.py <seed> <index>"""
import dataclasses, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from lld_slam_amd import Context, Optimizer, synth
import oracle_py as O


def main():
    seed, want = int(sys.argv[1]), int(sys.argv[2])
    ctx = Context(0); O.lib()
    rng = np.random.default_rng(seed)
    for it in range(want + 1):                                   # the draws of tools/fuzz_pose.py, in its order
        kw = dict(n_points=int(rng.choice([0, 2, 5, 30, 200, 1000, 1500])), n_lines=int(rng.choice([0, 2, 20, 200, 400])), outlier_frac=float(rng.choice([0.0, 0.1, 0.3, 0.6, 0.9])),
                  mono_frac=float(rng.choice([0.0, 0.0, 0.3, 1.0])), mono_line_frac=float(rng.choice([0.0, 0.0, 0.3, 1.0])))
        gamma = float(rng.choice([0.5, 0.5, 1.0, 0.1]))
        fseed = int(rng.integers(1, 2 ** 31))
        n_lines = kw["n_lines"]
        idx = None
        if n_lines and rng.random() < 0.5:
            idx = (np.cumsum(rng.integers(1, 4, n_lines)) - 1).astype(np.int32)
    f = synth.make_pose_frame(5000 + want, seed=fseed, **kw)
    if idx is not None and f.n_lines: f = dataclasses.replace(f, ln_frame_index=idx)
    print(kw, "gamma", gamma, "ln_frame_index", idx is not None)
    for rounds in (1, 2, 3, 4):
        o = O.pose_opt(f, gamma=gamma, n_rounds=rounds); g = Optimizer(ctx).PoseOptimization(f, gamma=gamma, n_rounds=rounds)
        print("rounds", rounds, "chi2 %.9e %.9e" % (g.chi2, o.chi2), "its", g.lm_iterations, o.lm_iterations, "trials", g.lm_trials, o.lm_trials, "inliers", g.n_inliers, o.n_inliers,
              "pose diff %.1e" % float(np.abs(g.pose_qt - o.pose_qt).max()), "pt flags differing", int((g.pt_outlier != o.pt_outlier).sum()), "ln flags differing",
              int((g.ln_outlier != o.ln_outlier).sum()), "outliers", int(g.pt_outlier.sum()), int(g.ln_outlier.sum()))


if __name__ == "__main__":
    main()
