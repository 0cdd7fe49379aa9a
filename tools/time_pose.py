"""Latency / throughput of PoseOptimization on the GPU box: single-frame lld_pose_opt (host buffers in and out), a resident
one-frame batch (kernel + fetch), and a 4096-frame batch.  python tools/time_pose.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (loads the HIP runtime the library links against)
from lld_slam_amd import Context, Optimizer, PoseBatch, synth

ctx = Context(0)
opt = Optimizer(ctx)
f = synth.make_pose_frame(0)
out = {}
for _ in range(5): opt.PoseOptimization(f, gamma=0.5)
ts = []
for _ in range(50):
    t = time.perf_counter(); opt.PoseOptimization(f, gamma=0.5); ts.append(time.perf_counter() - t)
out["single_call_ms_median"] = 1e3 * float(np.median(ts)); out["single_call_ms_min"] = 1e3 * min(ts)
with PoseBatch(ctx, [f], gamma=0.5) as b:
    for _ in range(3): b.solve(); b.download(0)
    ts = []
    for _ in range(50):
        t = time.perf_counter(); b.solve(); b.download(0); ts.append(time.perf_counter() - t)
    out["resident_one_frame_ms_median"] = 1e3 * float(np.median(ts))
nf = int(os.environ.get("NF", "4096"))
frames = [synth.make_pose_frame(i % 64) for i in range(64)]
frames = [frames[i % 64] for i in range(nf)]
with PoseBatch(ctx, frames, gamma=0.5) as b:
    b.solve(); b.download(0)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); b.solve(); b.download(0); ts.append(time.perf_counter() - t)
    out["batch_frames"] = nf; out["batch_ms_median"] = 1e3 * float(np.median(ts)); out["batch_frames_per_s"] = nf / float(np.median(ts))
print(json.dumps(out))
