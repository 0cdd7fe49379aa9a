#!/usr/bin/env python3
"""Random Tracking-thread frames through lld_frame_track_* against the oracle's OWN run of the sequence (oracle/oracle_tracking.py).
    python tools/fuzz_track_chain.py [n=300] [seed=0]
Every frame: random sizes (keypoints, last-frame points, local map, lines on / off), random prediction error (some far enough for the
wide retry), random stereo fraction.  Per frame, both stages: EQUAL = every id, outlier flag and counter equal, the pose inside 1e-7
(quaternion per component, translation against its norm) and chi2 inside 1e-7 relative; of those, PATH = the LM iteration / trial counts
differ (a converged round ended at another iteration: see tests/test_gpu_track_chain.py).  MISMATCH = anything else."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_tracking as OT
from lld_slam_amd import Context, synth
from lld_slam_amd.tracking import DeviceTrackedFrame

KEYS = ("kp_point_id", "kp_outlier", "ln_line_id", "ln_outlier")
CNT = ("n_inliers", "n_edges", "n_search_first", "n_search", "used_wide", "n_points", "n_points_map", "n_lines_matched", "n_lines", "n_discarded", "n_point_edges", "n_in_view")


def compare(g, e):
    if not all(np.array_equal(g[k], e[k]) for k in KEYS + (("mp_in_view",) if "mp_in_view" in g and "mp_in_view" in e else ())): return "MISMATCH ids/flags", 0, 0, 0
    if not all(g[k] == e[k] for k in CNT): return "MISMATCH counters", 0, 0, 0
    dq = float(np.max(np.abs(g["pose_qt"][:4] - e["pose_qt"][:4])))
    dt = float(np.linalg.norm(g["pose_qt"][4:] - e["pose_qt"][4:]) / max(1.0, np.linalg.norm(e["pose_qt"][4:])))
    dc = abs(g["chi2"] - e["chi2"]) / max(abs(e["chi2"]), 1e-12)
    same = g["lm_iterations"] == e["lm_iterations"] and g["lm_trials"] == e["lm_trials"]
    if max(dq, dt) > 1e-7: return "MISMATCH pose", dq, dt, dc
    if dc > 1e-7 and e["chi2"] > 1e-9: return "MISMATCH chi2", dq, dt, dc
    return ("EQUAL" if same else "PATH"), dq, dt, dc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    tally = {}; worst = [0.0, 0.0, 0.0]; wide = 0; t0 = time.time()
    with Context(0) as ctx:
        for i in range(n):
            n_kp = int(rng.choice([300, 800, 1500, 2000, 3000])); n_map = int(rng.integers(200, 3500)); n_last = int(rng.integers(0, min(n_map, 1800)))
            lines = rng.random() < 0.8
            rot, trans = (float(rng.uniform(0.05, 0.6)), float(rng.uniform(0.01, 0.15))) if rng.random() < 0.85 else (float(rng.uniform(1.5, 3.0)), float(rng.uniform(0.5, 1.5)))
            kw = dict(n_kp=n_kp, n_map=n_map, n_last=n_last, rot_deg=rot, trans=trans)
            if lines: kw.update(n_lines=int(rng.integers(20, 500)), n_map_lines=int(rng.integers(10, 400)))
            else: kw.update(n_lines=0)
            if lines: kw["n_last_lines"] = int(rng.integers(0, kw["n_map_lines"] + 1))
            sc = synth.make_tracking_scene(1000 + seed * 100000 + i, **kw)
            if rng.random() < 0.15: sc["frame"].uright = np.where(rng.random(sc["frame"].n) < 0.7, -1.0, sc["frame"].uright).astype(np.float32)
            with DeviceTrackedFrame(ctx, sc["frame"], sc["cam"], sc.get("lines")) as tf:
                tf.track_with_motion_model(sc["Tcw_guess"], sc["last"], sc["last_ids"], sc.get("last_lines"))
                tf.track_local_map(sc["map_points"], sc["map_ids"], sc.get("local_lines"))
                g = tf.download()
            e = OT.track_frame(sc)
            wide += e[0]["used_wide"]
            if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) == i:
                # the same two PoseOptimization problems (as the ORACLE's chain built them) through lld_pose_opt: does the difference sit in the
                # chain's edge assembly or in the optimiser's path?
                import oracle_py as O
                from lld_slam_amd import Optimizer
                for k, prob in enumerate(OT.track_frame.last_problems):
                    o = O.pose_opt(prob, 0.5); d = Optimizer(ctx).PoseOptimization(prob, 0.5)
                    print(f"  stage {k + 1} problem {prob.n_points}+{prob.n_lines}: oracle {o.lm_iterations}/{o.lm_trials} chi2 {o.chi2:.10g} | lld_pose_opt {d.lm_iterations}/{d.lm_trials} chi2 {d.chi2:.10g} "
                          f"| chain {g[k]['lm_iterations']}/{g[k]['lm_trials']} chi2 {g[k]['chi2']:.10g} | dpose(pose_opt, oracle) {np.max(np.abs(d.pose_qt - o.pose_qt)):.2e} dpose(chain, oracle) {np.max(np.abs(g[k]['pose_qt'] - o.pose_qt)):.2e}")
                    if k == 1:
                        from lld_slam_amd import host
                        T = host.se3_to_tcw_f32(ctx.lib, g[0]["pose_qt"]); q0 = host.se3_from_tcw_f32(ctx.lib, T)
                        print("      start pose of stage 2: device-derived vs oracle problem", np.max(np.abs(q0 - prob.pose_qt)), " stage-1 pose device vs oracle", np.max(np.abs(g[0]["pose_qt"] - e[0]["pose_qt"])))
                        import copy
                        p2 = copy.copy(prob); p2.pose_qt = q0
                        d2 = Optimizer(ctx).PoseOptimization(p2, 0.5)
                        print(f"      lld_pose_opt on the oracle's edges from the device-derived start: {d2.lm_iterations}/{d2.lm_trials} chi2 {d2.chi2:.10g}")
                    for nr in (1, 2, 3, 4):
                        o = O.pose_opt(prob, 0.5, n_rounds=nr); d = Optimizer(ctx).PoseOptimization(prob, 0.5, n_rounds=nr)
                        print(f"      rounds {nr}: oracle {o.lm_iterations}/{o.lm_trials} chi2 {o.chi2:.12g} inl {o.n_inliers} | device {d.lm_iterations}/{d.lm_trials} chi2 {d.chi2:.12g} inl {d.n_inliers}")
            for s in range(2):
                r, dq, dt, dc = compare(g[s], e[s])
                tally[r] = tally.get(r, 0) + 1
                if r in ("EQUAL", "PATH"):
                    worst[0] = max(worst[0], dq); worst[1] = max(worst[1], dt)
                    worst[2] = max(worst[2], dc)
                if r.startswith("MISMATCH"):
                    print(f"{r:20s} frame {i} stage {s + 1} {kw} dq {dq:.2e} dt {dt:.2e} dchi2 {dc:.2e} device its/trials {g[s]['lm_iterations']}/{g[s]['lm_trials']} oracle {e[s]['lm_iterations']}/{e[s]['lm_trials']} "
                          f"points {g[s]['n_points']}/{e[s]['n_points']} search {g[s]['n_search']}/{e[s]['n_search']}", flush=True)
    tot = sum(tally.values())
    print(f"fuzzed {n} frames ({tot} stage records, {wide} with the wide retry) in {time.time() - t0:.0f} s: " + ", ".join(f"{k} {v}" for k, v in sorted(tally.items())) +
          f"; worst |dq| {worst[0]:.2e}, |dt|/max(1,|t|) {worst[1]:.2e}, rel chi2 {worst[2]:.2e}")


if __name__ == "__main__":
    main()
