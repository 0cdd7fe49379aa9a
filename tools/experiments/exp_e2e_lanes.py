"""Host buffers in -> results out, several independent lanes (one context + one host thread each): create -> solve -> download all -> destroy.
   python tools/exp_e2e_lanes.py [windows=256] [batches_per_lane=4] [lanes=1,2,3,4]"""
import ctypes as C, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lld_slam_amd import Context, synth, host, abi



SOLVE_LOCK = threading.Lock() if os.environ.get("LLD_EXP_SOLVE_LOCK") == "1" else None      # one solve at a time, creates / downloads of the other lanes overlap it


def lane(k, res, t_done, start, lib, ws, cw, params, nw, nb, t_first):
    ctx = Context(0)
    outs = [host.BAOutput.alloc(w) for w in ws]; crs = (abi.BAResult * nw)(*[o.to_c() for o in outs])
    start.wait()
    laps = np.zeros(4); gpu_ms = np.zeros(6); ms6 = (C.c_double * 6)()
    for b in range(nb):
        h = C.c_void_p()
        t0 = time.perf_counter()
        host.check(lib.fn("ba_batch_create")(ctx.handle, nw, cw, C.byref(params), C.byref(h)), "create")
        t1 = time.perf_counter()
        flag = C.c_int(0)
        if SOLVE_LOCK is not None:
            with SOLVE_LOCK: host.check(lib.fn("ba_batch_solve")(h, C.byref(flag)), "solve")
        else: host.check(lib.fn("ba_batch_solve")(h, C.byref(flag)), "solve")
        t2 = time.perf_counter()
        lib.fn("ba_batch_phase_ms")(h, ms6)
        host.check(lib.fn("ba_batch_download_range")(h, 0, nw, crs), "download")
        t3 = time.perf_counter()
        lib.fn("ba_batch_destroy")(h)
        t4 = time.perf_counter()
        if b > 0: laps += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]; gpu_ms += np.array(ms6[:])
        else: t_first[k] = t4
    if k == 0:
        print("  lane 0, mean ms per batch after the first: create %.1f solve %.1f download %.1f destroy %.1f" % tuple(1e3 * laps / max(1, nb - 1)), flush=True)
        print("  lane 0, the solve on the device (HIP events, after its turn came): %.1f ms; sum of the per-phase event times %.1f ms" % (gpu_ms[5] / max(1, nb - 1), gpu_ms[:5].sum() / max(1, nb - 1)), flush=True)
    t_done[k] = time.perf_counter()
    res[k] = crs[0].stats.chi2_final
    ctx.close()


if __name__ == "__main__":            # (generate_windows spawns worker processes that re-import this file)
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lanes_list = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3,4").split(",")]
    ws = synth.generate_windows(0, nw, int(os.environ.get('LLD_GEN_WORKERS', '0')))
    lib = abi.product()
    cw = (abi.BAWindow * nw)(*[w.to_c() for w in ws])
    params = host.ba_params(lib)
    
    for L in lanes_list:
        res = [None] * L; t_done = [0.0] * L; t_first = [0.0] * L; start = threading.Barrier(L + 1)
        th = [threading.Thread(target=lane, args=(k, res, t_done, start, lib, ws, cw, params, nw, nb, t_first)) for k in range(L)]
        for t in th: t.start()
        time.sleep(2.0)                       # contexts up, outputs allocated
        start.wait(); t0 = time.perf_counter()
        for t in th: t.join()
        el = max(t_done) - t0
        steady = max(t_done) - max(t_first)
        print(f"lanes {L}: {L * nb} batches of {nw} windows in {el * 1e3:.1f} ms -> {L * nb * nw / el:.0f} windows/s end to end (chi2[0] {res[0]:.6f}); "
              f"after every lane's first batch: {L * (nb - 1) * nw / steady:.0f} windows/s (GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')})", flush=True)
