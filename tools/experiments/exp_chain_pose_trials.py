"""LM iteration / trial counts of the two PoseOptimization problems of a tracking scene: oracle, device lld_pose_opt, device chain.
python tools/experiments/exp_chain_pose_trials.py <mono|scene id>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_tracking as OT
import oracle_py as O
from lld_slam_amd import Context, Optimizer, synth
from lld_slam_amd.tracking import DeviceTrackedFrame
arg = sys.argv[1] if len(sys.argv) > 1 else "mono"
if arg == "mono":
    sc = synth.make_tracking_scene(20, n_kp=600, n_map=700, n_last=300)
    F = sc["frame"]; F.uright = np.full_like(F.uright, -1.0)
    sc["lines"]["line_matches"] = np.full_like(sc["lines"]["line_matches"], -1)
else:
    sc = synth.make_tracking_scene(int(arg))
e1, e2 = OT.track_frame(sc)
probs = OT.track_frame.last_problems
with Context(0) as ctx:
    for k, prob in enumerate(probs):
        o = O.pose_opt(prob, 0.5)
        g = Optimizer(ctx).PoseOptimization(prob, 0.5)
        print(f"stage {k + 1}: edges {prob.n_points}+{prob.n_lines} oracle its {o.lm_iterations} trials {o.lm_trials} chi2 {o.chi2:.12g} | device pose_opt its {g.lm_iterations} trials {g.lm_trials} chi2 {g.chi2:.12g} "
              f"| dpose {np.max(np.abs(g.pose_qt - o.pose_qt)):.2e} outliers equal {np.array_equal(g.pt_outlier, o.pt_outlier)}")
        import oracle_py
        for nr in (1,):
            o = O.pose_opt(prob, 0.5, n_rounds=nr); g = Optimizer(ctx).PoseOptimization(prob, 0.5, n_rounds=nr)
            print(f"     rounds {nr}: oracle {o.lm_iterations}/{o.lm_trials} device {g.lm_iterations}/{g.lm_trials}")
    with DeviceTrackedFrame(ctx, sc["frame"], sc["cam"], sc.get("lines")) as tf:
        tf.track_with_motion_model(sc["Tcw_guess"], sc["last"], sc["last_ids"], sc.get("last_lines"))
        tf.track_local_map(sc["map_points"], sc["map_ids"], sc.get("local_lines"))
        r1, r2 = tf.download()
    print("chain  :", r1["lm_iterations"], r1["lm_trials"], r1["chi2"], "|", r2["lm_iterations"], r2["lm_trials"], r2["chi2"])
    print("oracle :", e1["lm_iterations"], e1["lm_trials"], e1["chi2"], "|", e2["lm_iterations"], e2["lm_trials"], e2["chi2"])
