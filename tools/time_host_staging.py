#!/usr/bin/env python3
"""Host time of every staging stage of lld_ba_batch_create for ONE LBA-B window (experiments build, no GPU needed):
tasks | edges -> packed records | point chunks | line chunks | CSRs | Cholesky plan.   python tools/time_host_staging.py [n_windows_hint=256]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lld_slam_amd import abi, synth, host
lib = abi.Lib(os.path.join(ROOT, "lld_slam_amd", "csrc", "liblld_amd_exp.so"), "lld_")
hint = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for wid in range(3):
    w = synth.make_lba_b(wid)
    cw = w.to_c()
    ms = np.zeros(6)
    fn = lib.dll.lld_exp_stage_timing; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert fn(C.byref(cw), hint, 20, ms.ctypes.data) == 0
    print(f"window {wid}: tasks {ms[0]:.3f}  edges {ms[1]:.3f}  point chunks {ms[2]:.3f}  line chunks {ms[3]:.3f}  csr {ms[4]:.3f}  chol plan {ms[5]:.3f}   sum {ms.sum():.3f} ms (one host thread)")
