import os, sys
sys.path.insert(0, ".")
from lld_slam_amd import Context, BABatch, synth
ctx = Context(0)
ws = [synth.make_lba_b(i) for i in range(256)]
with BABatch(ctx, ws) as b:
    b.set_groups(1); b.solve()
