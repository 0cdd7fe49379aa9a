"""Does a second resident frame per CU pay for the pose kernel?  Batched PoseOptimization throughput against the frame size: at 1000 points +
200 stereo lines a frame's LDS image is 106 KB (one workgroup per CU), at 700 + 140 it is 74 KB (two).   python tools/exp_pose_occupancy.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from lld_slam_amd import Context, PoseBatch, synth

ctx = Context(0)
nf = 4096
for n_points, n_lines in ((1000, 200), (760, 150), (740, 148), (700, 140), (500, 100), (350, 70)):
    frames = [synth.make_pose_frame(i, n_points=n_points, n_lines=n_lines) for i in range(32)]
    frames = [frames[i % 32] for i in range(nf)]
    with PoseBatch(ctx, frames, gamma=0.5) as b:
        b.solve(); b.download(0)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); b.solve(); b.download(0); ts.append(time.perf_counter() - t)
    med = float(np.median(ts)); edges = n_points + 2 * n_lines
    print("points %4d lines %3d  edges %4d  %.3f ms  %8.0f frames/s  %6.2f M edges/s per sweep-equivalent" % (n_points, n_lines, edges, 1e3 * med, nf / med, nf * edges / med / 1e6), flush=True)
