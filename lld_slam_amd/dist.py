"""Multi-GPU plumbing of the batched local BA: windows shard across ranks, nothing is exchanged during the solve,
and the only collective is the gather of each rank's fixed-stride result records on rank 0.

Everything `bench.py` does between "which windows are mine" and "rank 0 holds every record" lives here, on torch tensors of whatever
device the process group's backend serves ("nccl" = RCCL over xGMI on the GPU box, "gloo" on host tensors in tests/test_distributed_cpu.py,
which imports this module - not a copy of it - with records produced by the CPU oracle).

Record layout (lld_ba_kernels.h, `BARecordHeader` + ba_finalize_kernel; the stride of a batch is the largest record, rounded to 256 B):

    header  48 B   chi2_round1, chi2_final (f64) | lm_iterations[2], lm_trials[2] (i32) | pcg_iterations, aborted, win_index, n_pt_obs (i32)
                   (win_index = position of the window in its rank's batch, n_pt_obs = its point-edge count: with the fixed cameras the record
                   carries, enough for rank 0 to check that a gathered record IS the window shard() assigned - verify_gathered_records)
    cam_qt  [n_cams][7] f64 | pt_xyz [n_points][3] f64 | line_x0 [n_lines][3] f64 | line_dir [n_lines][3] f64
    pt_obs_outlier [n_pt_obs] u8 | ln_edge_outlier [n_ln_obs][2] u8 | line_removed [n_lines] u8
"""
from __future__ import annotations

import os
import struct

import numpy as np

from .host import BAOutput, Window

RECORD_HEADER = struct.Struct("<2d4i4i")          # BARecordHeader
RECORD_ALIGN = 256


# ---------------------------------------------------------------------------------------------------------------- sharding
def shard(n_windows: int, world: int, rank: int, strong: bool) -> tuple[int, int]:
    """(first window id, count) of `rank`.  Weak scaling (the default of bench.py): every rank owns `n_windows` windows, ids
    [rank*n, (rank+1)*n).  Strong scaling: the SAME `n_windows` ids 0..n-1 split into `world` contiguous blocks (the 256
    windows of the BATCH config at 32 per GPU on 8 GPUs)."""
    if world < 1 or not 0 <= rank < world or n_windows < 0:
        raise ValueError("bad shard request")
    if not strong:
        return rank * n_windows, n_windows
    lo = n_windows * rank // world
    return lo, n_windows * (rank + 1) // world - lo


def host_thread_budget(world: int, cap: int = 16) -> int:
    """Host threads one rank may use for staging / generation: the node's cores are shared by `world` ranks, each of which also runs a
    polling solve loop (one thread) - eight ranks x 16 staging threads on one node was the unbudgeted default of round 2."""
    per_rank = (os.cpu_count() or 1) // max(1, world)
    return max(1, min(cap, per_rank - 1))


# ---------------------------------------------------------------------------------------------------------------- records
def record_bytes(w: Window) -> int:
    b = RECORD_HEADER.size + 8 * (7 * w.n_cams + 3 * w.n_points + 6 * w.n_lines) + w.n_pt_obs + 2 * w.n_ln_obs + w.n_lines
    return (b + RECORD_ALIGN - 1) // RECORD_ALIGN * RECORD_ALIGN


def record_stride(windows) -> int:
    """Stride lld_ba_batch_result_records reports for a batch of these windows (checked against the library in tests/test_gpu_ba.py)."""
    return max(record_bytes(w) for w in windows)


def unpack_record(rec: np.ndarray, w: Window) -> BAOutput:
    """One record (uint8 array of at least record_bytes(w)) -> BAOutput, exactly what lld_ba_batch_download fills."""
    rec = np.ascontiguousarray(rec, np.uint8)
    h = RECORD_HEADER.unpack_from(rec[:RECORD_HEADER.size].tobytes())
    off = RECORD_HEADER.size
    def f64(n, shape):
        nonlocal off
        a = rec[off:off + 8 * n].view(np.float64).reshape(shape).copy(); off += 8 * n
        return a
    def u8(n, shape):
        nonlocal off
        a = rec[off:off + n].reshape(shape).copy(); off += n
        return a
    cam = f64(7 * w.n_cams, (w.n_cams, 7)); pt = f64(3 * w.n_points, (w.n_points, 3))
    x0 = f64(3 * w.n_lines, (w.n_lines, 3)); dr = f64(3 * w.n_lines, (w.n_lines, 3))
    po = u8(w.n_pt_obs, (w.n_pt_obs,)); lo = u8(2 * w.n_ln_obs, (w.n_ln_obs, 2)); rm = u8(w.n_lines, (w.n_lines,))
    stats = dict(chi2_round1=h[0], chi2_final=h[1], lm_iterations=[h[2], h[3]], lm_trials=[h[4], h[5]], pcg_iterations=h[6], aborted=h[7],
                 n_pt_obs_outlier=int(po.sum()), n_ln_edge_outlier=int(lo.sum()), n_lines_removed=int(rm.sum()))
    return BAOutput(cam, pt, x0, dr, po, lo, rm, stats)


def pack_record(out: BAOutput, w: Window, stride: int, win_index: int = 0) -> np.ndarray:
    """Inverse of unpack_record (used where records do not come from the HIP library: the gloo test packs oracle results)."""
    rec = np.zeros(stride, np.uint8)
    s = out.stats
    rec[:RECORD_HEADER.size] = np.frombuffer(RECORD_HEADER.pack(s["chi2_round1"], s["chi2_final"], *s["lm_iterations"], *s["lm_trials"],
                                                                s.get("pcg_iterations", 0), s["aborted"], int(win_index), int(w.n_pt_obs)), np.uint8)
    off = RECORD_HEADER.size
    for a in (out.cam_qt, out.pt_xyz, out.line_x0, out.line_dir):
        b = np.ascontiguousarray(a, np.float64).reshape(-1).view(np.uint8); rec[off:off + b.size] = b; off += b.size
    for a in (out.pt_obs_outlier, out.ln_edge_outlier, out.line_removed):
        b = np.ascontiguousarray(a, np.uint8).reshape(-1); rec[off:off + b.size] = b; off += b.size
    assert off <= stride
    return rec


def record_identity(rec: np.ndarray, w: Window):
    """What a record says about which window it belongs to: (win_index, n_pt_obs, fixed-camera poses).  The fixed cameras travel through
    the solve untouched (Optimizer.cc:1037-1063), so they are a fingerprint of the INPUT window inside its result."""
    rec = np.ascontiguousarray(rec, np.uint8)
    h = RECORD_HEADER.unpack_from(rec[:RECORD_HEADER.size].tobytes())
    o = RECORD_HEADER.size + 8 * 7 * w.n_free_cams
    fixed = rec[o:o + 8 * 7 * (w.n_cams - w.n_free_cams)].view(np.float64).reshape(-1, 7)
    return int(h[8]), int(h[9]), fixed


def verify_gathered_records(rank_records, counts, stride: int, n_windows: int, world: int, strong: bool, make_window, per_rank: int = 2):
    """Rank 0, after the gather: `per_rank` records of every rank (first, last, ...) are checked against the window id shard() assigned
    to that position - win_index and edge count in the header, the fixed cameras bit for bit against make_window(id) (the synthetic
    generator is a pure function of the id), and a finished protocol (finite chi2, >= 1 iteration).  A rank that solved the wrong shard,
    or records that landed in the wrong slot of the gather, fail here.  Returns the number of records checked."""
    checked = 0
    for r in range(world):
        first, cnt = shard(n_windows, world, r, strong)
        if cnt != counts[r]:
            raise AssertionError(f"rank {r} reports {counts[r]} windows, shard() gives it {cnt}")
        buf = rank_records(r)
        ks = sorted({int(round(i * (cnt - 1) / max(1, per_rank - 1))) for i in range(min(per_rank, cnt))})
        for k in ks:
            w = make_window(first + k)
            rec = buf[k * stride:(k + 1) * stride]
            idx, n_pe, fixed = record_identity(rec, w)
            h = RECORD_HEADER.unpack_from(rec[:RECORD_HEADER.size].tobytes())
            if idx != k or n_pe != w.n_pt_obs or not np.array_equal(fixed, np.asarray(w.cam_qt[w.n_free_cams:], np.float64)):
                raise AssertionError(f"record {k} of rank {r} is not window {first + k}")
            if not (np.isfinite(h[1]) and h[1] > 0 and h[2] >= 1):
                raise AssertionError(f"record {k} of rank {r}: unfinished protocol {h[:6]}")
            checked += 1
    return checked


# ---------------------------------------------------------------------------------------------------------------- collective
def gather_rows(local, counts, world: int, rank: int, enabled: bool = True):
    """Fixed-stride gather of per-item result rows to rank 0: `local` is this rank's [n_local, ...] tensor (pose rows of a PoseOptimization
    shard: 7 f64 + inlier count; match rows: best / second index and distance per query, int32), `counts` the items per rank.  Uneven
    shards pad to the largest.  Returns the list of per-rank tensors (trimmed to their counts) on rank 0, None elsewhere; without a
    process group, [local].  Synchronous: these results are small (PO: 64 B per frame) next to the BA records."""
    if not enabled:
        return [local]
    import torch
    import torch.distributed as dist
    n_max = max(counts)
    if local.shape[0] != counts[rank]:
        raise ValueError("local rows do not match this rank's count")
    pad = local
    if local.shape[0] < n_max:
        pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device); pad[:local.shape[0]] = local
    pad = pad.contiguous()
    out = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, out, dst=0)
    return [o[:c] for o, c in zip(out, counts)] if rank == 0 else None


def max_count_stride(rec_stride: int, device, use_dist: bool) -> int:
    """The gather moves max(counts) x stride bytes from every rank: all ranks must agree on the stride (a rank whose windows have a
    smaller largest record would otherwise enter the collective with a different size)."""
    if not use_dist:
        return int(rec_stride)
    import torch
    import torch.distributed as dist
    t = torch.tensor([rec_stride], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())



class RecordGather:
    """The final gather, overlapped with the next solve: step() copies this rank's record buffer to a staging tensor and starts an
    asynchronous `dist.gather` to rank 0 from there; the solve that follows does not touch the interconnect, so the two do not compete.
    drain() waits for the gather in flight (bench.py calls it inside the timed region after the last step).

    `records` is a flat uint8 tensor (the library's record buffer viewed zero-copy on the GPU; a host tensor under gloo).  Every rank's
    buffer must have the same size: a rank that owns fewer windows (strong scaling, remainder) pads - `n_bytes` is the common size."""

    def __init__(self, records, world: int, rank: int, n_bytes: int | None = None, enabled: bool = True, local_stride: int | None = None,
                 common_stride: int | None = None):
        """`local_stride` / `common_stride`: record stride of this rank's batch and the largest over the ranks (max_count_stride); when they
        differ the records are re-strided into the staging buffer so that record k of every rank sits at k * common_stride."""
        import torch
        self.records = records; self.world = world; self.rank = rank; self.enabled = enabled
        self.n_bytes = int(n_bytes if n_bytes is not None else records.numel())
        self.local_stride = local_stride; self.common_stride = common_stride if common_stride is not None else local_stride
        restride = local_stride is not None and self.common_stride != local_stride
        if self.n_bytes < (records.numel() if not restride else records.numel() // local_stride * self.common_stride):
            raise ValueError("common gather size smaller than this rank's records")
        self.stage = torch.zeros(self.n_bytes, dtype=torch.uint8, device=records.device) if enabled else None
        self.gathered = [torch.empty(self.n_bytes, dtype=torch.uint8, device=records.device) for _ in range(world)] if (enabled and rank == 0) else None
        self.pending = None

    def step(self):
        if not self.enabled:
            return
        import torch
        import torch.distributed as dist
        self.drain()
        if self.local_stride is not None and self.common_stride != self.local_stride:
            n = self.records.numel() // self.local_stride
            self.stage[:n * self.common_stride].view(n, self.common_stride)[:, :self.local_stride].copy_(self.records.view(n, self.local_stride))
        else:
            self.stage[:self.records.numel()].copy_(self.records)
        # The copy runs on torch's stream; the library's next solve rewrites `records` on ITS streams (non-blocking, unordered with
        # torch's): the staging copy must have left the record buffer before step() returns.
        if self.stage.is_cuda:
            torch.cuda.current_stream(self.stage.device).synchronize()
        self.pending = dist.gather(self.stage, self.gathered, dst=0, async_op=True)

    def drain(self):
        if self.pending is not None:
            self.pending.wait(); self.pending = None

    def rank_records(self, r: int) -> np.ndarray:
        """Rank 0 only, after drain(): the bytes rank `r` sent (host copy)."""
        return self.gathered[r].cpu().numpy()


def barrier(use_dist: bool, cuda: bool):
    import torch
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
    if cuda:
        torch.cuda.synchronize()


def max_over_ranks(value: float, device, use_dist: bool) -> float:
    """The bench's time is the slowest rank's."""
    if not use_dist:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks_ok(ok: bool, device, use_dist: bool) -> bool:
    """True iff EVERY rank passes ok = True.  The one collective a rank reaches whether or not its local work succeeded: the ranks use it to
    agree on skipping the collectives that follow (a rank alone in a gather hangs the job until the backend's timeout)."""
    if not use_dist:
        return bool(ok)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()) == 1)


def gather_counts(count: int, device, use_dist: bool, world: int) -> list[int]:
    """Windows solved per rank (strong scaling shards unevenly when world does not divide the batch)."""
    if not use_dist:
        return [count]
    import torch
    import torch.distributed as dist
    t = torch.zeros(world, dtype=torch.int64, device=device); t[dist.get_rank()] = count
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]
