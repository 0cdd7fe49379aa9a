"""Multi-GPU plumbing of the batched local BA: windows shard across ranks, nothing is exchanged during the solve,
and the only collective is the gather of each rank's fixed-stride result records on rank 0.

Everything `bench.py` does between "which windows are mine" and "rank 0 holds every record" lives here, on torch tensors of whatever
device the process group's backend serves ("nccl" = RCCL over xGMI on the GPU box, "gloo" on host tensors in tests/test_distributed_cpu.py,
which imports this module - not a copy of it - with records produced by the CPU oracle).

Record layout (lld_ba_kernels.h, `BARecordHeader` + ba_finalize_kernel; the stride of a batch is the largest record, rounded to 256 B):

    header  48 B   chi2_round1, chi2_final (f64) | lm_iterations[2], lm_trials[2] (i32) | pcg_iterations, aborted, pad, pad (i32)
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
    h = RECORD_HEADER.unpack_from(rec.tobytes()[:RECORD_HEADER.size])
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


def pack_record(out: BAOutput, w: Window, stride: int) -> np.ndarray:
    """Inverse of unpack_record (used where records do not come from the HIP library: the gloo test packs oracle results)."""
    rec = np.zeros(stride, np.uint8)
    s = out.stats
    rec[:RECORD_HEADER.size] = np.frombuffer(RECORD_HEADER.pack(s["chi2_round1"], s["chi2_final"], *s["lm_iterations"], *s["lm_trials"],
                                                                s.get("pcg_iterations", 0), s["aborted"], 0, 0), np.uint8)
    off = RECORD_HEADER.size
    for a in (out.cam_qt, out.pt_xyz, out.line_x0, out.line_dir):
        b = np.ascontiguousarray(a, np.float64).reshape(-1).view(np.uint8); rec[off:off + b.size] = b; off += b.size
    for a in (out.pt_obs_outlier, out.ln_edge_outlier, out.line_removed):
        b = np.ascontiguousarray(a, np.uint8).reshape(-1); rec[off:off + b.size] = b; off += b.size
    assert off <= stride
    return rec


# ---------------------------------------------------------------------------------------------------------------- collective
class RecordGather:
    """The final gather, overlapped with the next solve: step() copies this rank's record buffer to a staging tensor and starts an
    asynchronous `dist.gather` to rank 0 from there; the solve that follows does not touch the interconnect, so the two do not compete.
    drain() waits for the gather in flight (bench.py calls it inside the timed region after the last step).

    `records` is a flat uint8 tensor (the library's record buffer viewed zero-copy on the GPU; a host tensor under gloo).  Every rank's
    buffer must have the same size: a rank that owns fewer windows (strong scaling, remainder) pads - `n_bytes` is the common size."""

    def __init__(self, records, world: int, rank: int, n_bytes: int | None = None, enabled: bool = True):
        import torch
        self.records = records; self.world = world; self.rank = rank; self.enabled = enabled
        self.n_bytes = int(n_bytes if n_bytes is not None else records.numel())
        if self.n_bytes < records.numel():
            raise ValueError("common gather size smaller than this rank's records")
        self.stage = torch.zeros(self.n_bytes, dtype=torch.uint8, device=records.device) if enabled else None
        self.gathered = [torch.empty(self.n_bytes, dtype=torch.uint8, device=records.device) for _ in range(world)] if (enabled and rank == 0) else None
        self.pending = None

    def step(self):
        if not self.enabled:
            return
        import torch.distributed as dist
        self.drain()
        self.stage[:self.records.numel()].copy_(self.records)
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


def gather_counts(count: int, device, use_dist: bool, world: int) -> list[int]:
    """Windows solved per rank (strong scaling shards unevenly when world does not divide the batch)."""
    if not use_dist:
        return [count]
    import torch
    import torch.distributed as dist
    t = torch.zeros(world, dtype=torch.int64, device=device); t[dist.get_rank()] = count
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]
