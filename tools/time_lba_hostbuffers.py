"""Host buffers in, host buffers out: BABatch create (flatten + H2D) + solve + download of the 256 bench windows.
The PCIe-inclusive rate DESIGN.md section 7 quotes; never bench.py's `value`.  Run on the GPU box: python tools/time_lba_hostbuffers.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
sys.argv = ["bench.py"]
import bench
from lld_slam_amd import BABatch, Context
def main():
    ws = bench.generate_windows(0, 256, 16)
    ctx = Context(0)
    b = BABatch(ctx, ws); b.solve(); b.close()
    t0 = time.perf_counter(); b = BABatch(ctx, ws); t1 = time.perf_counter(); b.solve(); t2 = time.perf_counter()
    outs = [b.download(i) for i in range(len(ws))]; t3 = time.perf_counter()
    print("create(host flatten + H2D) %.1f ms, solve %.1f ms, download+unpack %.1f ms -> %.0f windows/s host buffers in and out" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 256/(t3-t0)))
    b.close()

if __name__ == '__main__':
    main()
