"""Free device memory before / after a few thousand single calls of every kind through one context (nothing may accumulate)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from lld_slam_amd import Context, Optimizer, ORBmatcher, Tracking, BABatch, synth
ctx = Context(0)
w = synth.make_lba_small(1); f = synth.make_pose_frame(1, n_points=300, n_lines=60)
F = synth.make_orb_frame(2, 1000); q = synth.make_projection_queries(F, 2, 800)
P, L, FL = synth.make_line_track_scene(3, n_map=100, n_cur=120)
ws = [synth.make_lba_small(10 + i) for i in range(6)]
def once():
    Optimizer(ctx).LocalBundleAdjustment(w); Optimizer(ctx).PoseOptimization(f, gamma=0.5)
    ORBmatcher(ctx, 0.8).SearchByProjectionMap(F, q["desc"], q["valid"], q["uv"], q["ur"], q["level"], q["view_cos"], q["obs"], q["occupied"], 1.0)
    Tracking(ctx, P["K"], P["b"], 1.0 / P["sx"], 1.0 / P["sy"], mdThr=P["md_thr"]).AddLinesFrom(L, P["T_curr"], P["thr_reproj_base"], FL)
    with BABatch(ctx, ws) as b: b.solve()
for _ in range(20): once()
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
import resource
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for _ in range(600): once()
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("device memory change over 600 rounds of calls: %.2f MB; host peak RSS change: %.1f MB" % ((free0 - free1) / 1e6, (rss1 - rss0) / 1e3))
