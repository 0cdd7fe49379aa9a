"""Throughput of 512 local-map projection searches in one lld_orb_search_batch call (LLD_AMD_LIB selects the library)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from lld_slam_amd import Context, synth, orb_search as S
ctx = Context(0)
scenes = []
for i in range(64):
    F = synth.make_orb_frame(i, 2000); q = synth.make_projection_queries(F, i, 2000); scenes.append((F, q))
prep = [S.search_by_projection_map(None, None, Fi, qi["desc"], qi["valid"], qi["uv"], qi["ur"], qi["level"], qi["view_cos"], qi["obs"], qi["occupied"], 1.0, 0.8) for Fi, qi in scenes] * 8
S.run_batch(ctx.lib, ctx.handle, prep)
ts = []
for _ in range(7):
    t = time.perf_counter(); S.run_batch(ctx.lib, ctx.handle, prep); ts.append(time.perf_counter() - t)
print("searches/s", len(prep) / np.median(ts))
