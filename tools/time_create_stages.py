#!/usr/bin/env python3
"""Where lld_ba_batch_create spends its time on the GPU box (experiments build, LLD_BA_TIMING=1: lap times on stderr) for 256 LBA-B windows,
at 16 / 8 / 4 staging threads, plus the host's core count and one thread's per-stage times (tools/time_host_staging.py).
    python tools/time_create_stages.py 2> gpurun_out/create_stages.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["LLD_BA_TIMING"] = "1"


def main():
    from lld_slam_amd import BABatch, Context, abi, synth
    lib = abi.Lib(os.path.join(ROOT, "lld_slam_amd", "csrc", "liblld_amd_exp.so"), "lld_")
    print("cores:", os.cpu_count(), file=sys.stderr)
    ws = synth.generate_windows(0, 256)
    with Context(0, lib=lib) as ctx:
        b = BABatch(ctx, ws); b.solve(); b.close()
        for thr in (16, 8, 4):
            os.environ["LLD_HOST_THREADS"] = str(thr)
            for rep in range(2):
                print(f"--- LLD_HOST_THREADS={thr} rep {rep}", file=sys.stderr)
                t0 = time.perf_counter(); b = BABatch(ctx, ws); t1 = time.perf_counter()
                print(f"create wall {1e3 * (t1 - t0):.1f} ms (Python marshalling included)", file=sys.stderr)
                b.close()


if __name__ == "__main__":          # (generate_windows spawns worker processes that re-import this module)
    main()
