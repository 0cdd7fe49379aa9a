import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lld_slam_amd import Context, BABatch, synth
ctx = Context(0)
ws = [synth.make_lba_b(i) for i in range(8)]
with BABatch(ctx, ws) as b:
    b.solve()
    out = (C.c_ulonglong * 8)()
    print(ctx.lib.dll.lld_debug_chol_cycles(out))
    v = list(out)
    print("wall_clock64 ticks (100 MHz => x10 ns): A publish %d  B diag %d  C trsm %d  D mfma %d  loop %d  backsub %d" % tuple(v[:6]))
