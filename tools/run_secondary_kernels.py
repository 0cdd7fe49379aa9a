#!/usr/bin/env python3
"""The secondary configs of bench.py (PoseOptimization 4096 frames, ORB / LBD 1024 frame pairs, LBA-A 128 windows, one lld_local_ba
call) as a program of their own, for rocprofv3:  python3 tools/run_secondary_kernels.py  (prints bench.py's `secondary` object)"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
if __name__ == "__main__":
    import torch
    import bench
    from lld_slam_amd import Context
    torch.cuda.set_device(0)
    with Context(0) as ctx:
        print(json.dumps(bench.secondary_block(ctx, torch.device("cuda", 0), repeats=3)))
        # round 6: the frame-rate path as well - 20 single-frame PoseOptimization calls (config PO) and 20 frames of the device-resident
        # Tracking chain (lld_frame_track_*: orb_search_kernel, line kernels, pose_assemble / pose_opt_kernel<512>, track_* kernels)
        from lld_slam_amd import Optimizer, synth
        from lld_slam_amd.tracking import DeviceTrackedFrame
        f = synth.make_pose_frame(0)
        for _ in range(20): Optimizer(ctx).PoseOptimization(f, gamma=0.5)
        sc = synth.make_tracking_scene(0)
        with DeviceTrackedFrame(ctx, sc["frame"], sc["cam"], sc["lines"]) as tf:
            for _ in range(20):
                tf.track_with_motion_model(sc["Tcw_guess"], sc["last"], sc["last_ids"], sc["last_lines"])
                tf.track_local_map(sc["map_points"], sc["map_ids"], sc["local_lines"])
                tf.download()
