#!/usr/bin/env python3
"""Host wall clock of the Tracking thread's per-frame chain driven from compiled C++ (examples/harness track): one frame of 2000 keypoints +
300 stereo lines, 1200 last-frame points + 140 last-frame lines, 2500 local MapPoints + 260 local MapLines; `repeats` frames on one handle.
    python tools/time_track_chain.py [repeats=200] [scene=0]          (prints one JSON object; used by bench.py's secondary.tracking_frame)"""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lld_slam_amd import synth, tracking


def run(repeats=200, scene=0, between=False, **scene_kw):
    sc = synth.make_tracking_scene(scene, **scene_kw)
    with tempfile.TemporaryDirectory() as d:
        nl = tracking.write_harness_scene(os.path.join(d, "in.bin"), sc, repeats=repeats, download_between=between)
        p = subprocess.run([os.path.join(ROOT, "examples", "harness"), "track", os.path.join(d, "in.bin"), os.path.join(d, "out.bin")], capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            raise RuntimeError(p.stderr)
        r1, r2, ms = tracking.read_harness_result(os.path.join(d, "out.bin"), sc["frame"].n, nl, repeats)
    t = ms["total"][min(5, repeats - 1):]                        # the first frames pay allocations and kernel-attribute calls
    return dict(ms_per_frame=dict(min=round(float(t.min()), 4), median=round(float(np.median(t)), 4), p90=round(float(np.percentile(t, 90)), 4), repeats=int(t.size)),
                queue_motion_model_ms=round(float(np.median(ms["queue_motion_model"][5:])), 4), queue_local_map_ms=round(float(np.median(ms["queue_local_map"][5:])), 4),
                download_between_the_stages=bool(between),
                matches=dict(motion_model_search=r1["n_search"], after_motion_model=r1["n_points"], lines_after_motion_model=r1["n_lines"], local_map_search=r2["n_search"],
                             points_at_the_end=r2["n_points"], lines_at_the_end=r2["n_lines"]),
                lm=dict(stage1=[r1["lm_iterations"], r1["lm_trials"]], stage2=[r2["lm_iterations"], r2["lm_trials"]], edges=[r1["n_edges"], r2["n_edges"]])), sc, (r1, r2)


if __name__ == "__main__":
    rep = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    scn = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    out = dict(one_download=run(rep, scn, False)[0], download_between=run(rep, scn, True)[0], points_only=run(rep, scn, False, n_lines=0)[0])
    print(json.dumps(out, indent=1))
