#!/usr/bin/env python3
"""bench.py — throughput of the batched local bundle adjustment (the metric of BASELINE.json) on N MI355X.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (Optimizer::LocalBundleAdjustment, both LM rounds, outlier protocol, read-back) over
one batch of synthetic LBA-B windows (50 free + 10 fixed KFs, 10 000 points x 6 stereo observations, 2 000 lines x 5 KFs x
left/right = 80 000 edges) that is already resident in HBM; every step restarts from the uploaded initial state.  Windows
are independent, so ranks shard the window list (weak scaling: --windows-per-gpu windows on every GPU) and the only
collective is the gather of the fixed-stride result records to rank 0 (RCCL), inside the timed region.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the step (by summed HIP-event time):
achieved = algorithmic bytes / event time, peak = 8 TB/s HBM3E.  `cpu_baseline` is the single-threaded CPU
oracle (a port of the reference algorithm, not the reference binary) timed on a bounded sample of the same windows.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# The solve keeps up to four window groups in flight on separate HIP streams.  ROCm maps the streams of a process onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); once RCCL has created its own streams, several of ours share a queue and
# serialise (measured: 4240 -> 3580 windows/s on one GPU with an initialised process group).  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

METRIC = "local-BA windows/sec (50 KF, 10k pts, 2k lines) at matched chi2; 1/2/4/8 GPU"
HBM_PEAK_GBS = 8000.0
PHASES = ["ba_linearize", "ba_schur", "ba_pcg", "ba_backsub", "ba_control"]


def algorithmic_bytes(w, n_trials, n_iters, pcg_iters):
    """Algorithmic HBM bytes of each kernel family for ONE window over a whole solve (SURVEY.md §8(d) record sizes).

    E_in: 44 B per stereo point edge, 36 B mono, 60 B per line edge; W: 144 B / 192 B per edge with a free camera;
    V: 72 B per point, 128 B per line; X: 56 B camera, 24 B point, 40 B line; S: 288 B per 6x6 block.
    """
    import numpy as np
    nf = w.n_free_cams
    mono = int(np.count_nonzero(w.pt_obs_uvr[:, 2] < 0))
    es = w.n_pt_obs - mono
    el = w.n_ln_obs + int(np.count_nonzero(w.ln_obs_right[:, 0] >= 0))
    e_in = 44 * es + 36 * mono + 60 * el
    free_p = int(np.count_nonzero(w.pt_obs_cam < nf))
    free_l = int(np.count_nonzero(np.repeat(w.ln_obs_cam < nf, 2)[(np.stack([np.ones(w.n_ln_obs, bool), w.ln_obs_right[:, 0] >= 0], 1)).reshape(-1)]))
    wb = 144 * free_p + 192 * free_l
    v = 72 * w.n_points + 128 * w.n_lines
    x = 56 * w.n_cams + 24 * w.n_points + 40 * w.n_lines
    s_blocks = nf * nf                                   # the kernel stores the full symmetric matrix
    return {
        "ba_linearize": n_iters * (e_in + wb + v + x // 2),
        "ba_schur": n_trials * (wb + v + 288 * s_blocks),
        "ba_pcg": pcg_iters * (288 * s_blocks + 5 * 48 * nf),
        "ba_backsub": n_trials * (wb + v + e_in + x),
        "ba_control": 0,
    }


def schur_fmas(w, n_trials):
    """fp64 FMAs of the Schur family for ONE window over a whole solve (the arithmetic that actually bounds it): per landmark with k
    free-camera observations, k x (rebuild the 6xD Hpl block in closed form ~70 for a point / 0 for a line whose block is stored,
    Z = W L^-T 6*D*(D+1)/2, Z (L^-1 b) 6*D) + one D x D Cholesky (~10 / ~20) + k(k+1)/2 block products Z_a Z_b^T of 6*6*D."""
    import numpy as np
    nf = w.n_free_cams
    kp = np.add.reduceat((w.pt_obs_cam < nf).astype(np.int64), w.pt_obs_start[:-1]) if w.n_points else np.zeros(0, np.int64)
    kp = np.where(np.diff(w.pt_obs_start) > 0, kp, 0)
    kl = np.add.reduceat((w.ln_obs_cam < nf).astype(np.int64), w.ln_obs_start[:-1]) if w.n_lines else np.zeros(0, np.int64)
    kl = np.where(np.diff(w.ln_obs_start) > 0, kl, 0)
    pts = np.sum(kp * (70 + 36 + 18) + 10 * (kp > 0) + kp * (kp + 1) // 2 * 108)
    lns = np.sum(kl * (60 + 24) + 20 * (kl > 0) + kl * (kl + 1) // 2 * 144)
    return int(n_trials * (pts + lns))


def _make(wid):
    from lld_slam_amd import synth
    return synth.make_lba_b(wid)


def generate_windows(first_id, count, workers):
    if workers <= 1 or count < 4:
        return [_make(first_id + i) for i in range(count)]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(workers) as pool:      # spawn: never fork a process that may have touched the GPU
        return pool.map(_make, range(first_id, first_id + count), chunksize=max(1, count // (4 * workers)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows-per-gpu", type=int, default=256)
    ap.add_argument("--cpu-sample", type=int, default=10, help="LBA-B windows timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gen-workers", type=int, default=0, help="processes generating the synthetic windows (0 = auto; use 1 under rocprofv3)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and world != 1:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    n_gpus = world

    # ---- synthetic windows for this rank (generated before anything touches the GPU)
    wpg = args.windows_per_gpu
    ncpu = os.cpu_count() or 1
    workers = args.gen_workers if args.gen_workers > 0 else max(1, min(16, ncpu // max(1, n_gpus)))
    t0 = time.time()
    windows = generate_windows(rank * wpg, wpg, workers)
    gen_s = time.time() - t0

    import numpy as np
    import torch
    import torch.distributed as dist
    from lld_slam_amd import BABatch, Context

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LLD_BENCH_FORCE_DIST=1 takes the RCCL path even with one rank (checks the collective plumbing on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("LLD_BENCH_FORCE_DIST") == "1"
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the one JSON line
        sys.stdout.flush(); saved_stdout = os.dup(1); os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)          # RCCL

    ctx = Context(local_rank)
    batch = BABatch(ctx, windows, gamma=1.0)
    rec_ptr, rec_stride = batch.result_records()
    rec_bytes = rec_stride * wpg

    class _Dev:                                                   # zero-copy torch view of the library's record buffer
        __cuda_array_interface__ = {"shape": (rec_bytes,), "typestr": "|u1", "data": (rec_ptr, False), "version": 2}
    records = torch.as_tensor(_Dev(), device=dev)
    gathered = [torch.empty(rec_bytes, dtype=torch.uint8, device=dev) for _ in range(world)] if (use_dist and rank == 0) else None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The gather of one step's records overlaps the solve of the next one: the records are copied to a staging tensor (108 MB,
    # device to device) and gathered from there asynchronously; the solve does not use xGMI, so the two do not compete.
    stage = torch.empty_like(records) if use_dist else None
    pending = [None]

    def step():
        batch.solve()                     # synchronous on the library's stream (it polls the LM state every super-step)
        if use_dist:
            if pending[0] is not None:
                pending[0].wait()
            stage.copy_(records)
            pending[0] = dist.gather(stage, gathered, dst=0, async_op=True)   # the final gather over xGMI: the only collective

    def drain():
        if pending[0] is not None:
            pending[0].wait(); pending[0] = None

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                               # the last gather is inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    # Roofline pass (untimed): the timed steps keep several window groups in flight on separate streams, so their HIP-event
    # brackets overlap; one more step with a single group gives disjoint per-kernel-family event times on that stream.
    phase = np.zeros(6); launches = np.zeros(5)
    prof_steps = 1
    batch.set_groups(1)
    batch.solve()
    phase += batch.phase_ms()
    launches += np.array([batch.kernel_stats(k)[0] for k in range(5)])
    batch.set_groups(0)
    # Measured stream ceiling of this GPU (SURVEY.md §8d asks for it next to the nominal 8 TB/s): a device-to-device copy of 2 GiB,
    # bytes read + bytes written over the HIP-event time, best of 5.
    stream_gbs = None
    if rank == 0:
        try:
            a = torch.empty(1 << 31, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
            b.copy_(a); torch.cuda.synchronize()
            best = 1e30
            for _ in range(5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            stream_gbs = round(2.0 * a.numel() / (best * 1e-3) / 1e9, 1)
            del a, b
        except Exception:
            stream_gbs = None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = batch.stats()
    result = None
    if rank == 0:
        total_windows = wpg * world * args.steps
        value = total_windows / elapsed
        # ---- roofline of the dominant kernel family (HIP events recorded on the library's own stream)
        per = {k: 0 for k in PHASES}
        for w, s in zip(windows, stats):
            ab = algorithmic_bytes(w, sum(s["lm_trials"]), sum(s["lm_iterations"]), s["pcg_iterations"])
            for k in PHASES:
                per[k] += ab[k]
        kdom = int(np.argmax(phase[:5])); kname = PHASES[kdom]
        dom_ms = phase[kdom] / prof_steps; dom_launches = launches[kdom] / prof_steps
        achieved = per[kname] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kname)
            except Exception:
                traffic = None
        fma_total = sum(schur_fmas(w, sum(st_["lm_trials"])) for w, st_ in zip(windows, stats)) if kname == "ba_schur" else None
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "measured_copy_ceiling_GBps": stream_gbs,
                    "launches_per_step": dom_launches, "avg_launch_ms": round(dom_ms / max(dom_launches, 1), 4),
                    "algorithmic_bytes_per_launch": int(per[kname] / max(dom_launches, 1)),
                    "phase_ms_single_stream_step": {k: round(phase[i] / prof_steps, 3) for i, k in enumerate(PHASES)},
                    "solve_ms_single_stream_step": round(phase[5] / prof_steps, 3)}
        if fma_total is not None and dom_ms > 0:      # the arithmetic roofline that actually binds this family (fp64 vector FMA, 39.3 T FMA/s)
            tf = fma_total / (dom_ms * 1e-3) / 1e12
            roofline["fp64_fma"] = {"achieved_TFMA_per_s": round(tf, 2), "peak_TFMA_per_s": 39.3, "frac": round(tf / 39.3, 4)}
        # ---- CPU baseline + matched-chi2 check on a bounded sample (N=1 only)
        cpu = None; parity = None
        if world == 1 and not args.no_cpu_baseline and args.cpu_sample > 0:
            import oracle_py as O
            O.lib()
            ns = min(args.cpu_sample, wpg)
            tc = time.perf_counter(); ores = [O.local_ba(windows[i]) for i in range(ns)]; cpu_s = time.perf_counter() - tc
            rel = max(abs(stats[i]["chi2_final"] - ores[i].stats["chi2_final"]) / max(ores[i].stats["chi2_final"], 1e-300) for i in range(ns))
            same = all(np.array_equal(batch.download(i).pt_obs_outlier, ores[i].pt_obs_outlier) and
                       np.array_equal(batch.download(i).line_removed, ores[i].line_removed) for i in range(ns))
            # the same port with one window per host thread (the reference's g2o is single-threaded per window; ctypes drops the GIL)
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, min(os.cpu_count() or 1, 32, wpg))
            tp = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(lambda i: O.local_ba(windows[i]), range(nthr)))
            par_s = time.perf_counter() - tp
            cpu = {"value": round(ns / cpu_s, 4), "unit": "windows/s", "cores": 1, "kind": "port",
                   "sample": f"{ns} LBA-B windows (ids 0..{ns - 1}) through oracle/liblld_oracle.so, 1 thread, exact dense LDLT of the reduced system",
                   "all_cores": {"value": round(nthr / par_s, 4), "cores": nthr, "sample": f"{nthr} windows, one per thread, concurrently"}}
            gres = [batch.download(i) for i in range(ns)]
            pose_err = max(float(np.max(np.abs(gres[i].cam_qt - ores[i].cam_qt)) / np.max(np.abs(ores[i].cam_qt))) for i in range(ns))
            pt_err = [np.linalg.norm(gres[i].pt_xyz - ores[i].pt_xyz, axis=1) / np.linalg.norm(ores[i].pt_xyz, axis=1) for i in range(ns)]
            parity = {"windows_checked": ns, "max_rel_chi2_final": float(rel), "outlier_sets_identical": bool(same),
                      "max_rel_pose": pose_err, "median_rel_point": float(np.median(np.concatenate(pt_err))),
                      "points_within_1e-5": float(np.mean(np.concatenate(pt_err) <= 1e-5))}
        result = {
            "metric": METRIC, "value": round(value, 3), "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batched LocalBundleAdjustment, {wpg} LBA-B windows per GPU (50 free + 10 fixed KF, 10k points x 6 stereo obs, "
                                   f"2k lines x 5 KF x 2 images = 80k edges), 5+15 LM iterations, gamma=1",
                       "windows_per_gpu": wpg, "edges_per_window": int(windows[0].n_edges()), "parallelism": f"{world} x independent window batches, RCCL gather of result records",
                       "result_record_bytes": int(rec_stride), "generate_s": round(gen_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
            "lm": {"mean_trials_per_window": float(np.mean([sum(s["lm_trials"]) for s in stats])),
                   "mean_pcg_iterations_per_trial": float(np.sum([s["pcg_iterations"] for s in stats]) / max(1, np.sum([sum(s["lm_trials"]) for s in stats])))},
        }
    batch.close(); ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if saved_stdout is not None:
        sys.stdout.flush(); os.dup2(saved_stdout, 1); os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
