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
