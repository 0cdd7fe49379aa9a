"""lld_line_match_stereo 300 x 300, 40 calls (for rocprofv3 --kernel-trace): which kernel holds the call's time"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
from lld_slam_amd import Context, TwoFrameLineMatcher, synth
with Context(0) as ctx:
    sl = synth.make_stereo_lines(0, 300, 300)
    tm = TwoFrameLineMatcher(ctx, 2.0, sl["K"], sl["b"], 20)
    ts = []
    for _ in range(40):
        t = time.perf_counter(); tm.MatchLines(sl["desc_left"], sl["desc_right"], lines=sl["left"], other_lines=sl["right"], octaves=sl["left_octave"], other_octaves=sl["right_octave"]); ts.append(time.perf_counter() - t)
    print("median ms", 1e3 * float(np.median(ts[5:])))
