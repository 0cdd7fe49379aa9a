"""N>1 path on CPU: window sharding and the final gather of fixed-stride result records, world_size 2 over gloo.

The GPU bench shards `--windows-per-gpu` windows to every rank and gathers each rank's record buffer on rank 0
(bench.py); here the same plumbing runs with host tensors, the records being produced by the CPU oracle instead of the
HIP library (there is no GPU in this container)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_py as O
    from lld_slam_amd import synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    wpg = 3                                            # windows per rank (weak scaling: every rank gets wpg windows)
    ids = [rank * wpg + i for i in range(wpg)]
    wins = [synth.make_lba_small(i, n_free=3, n_fixed=1, n_points=40, n_lines=6) for i in ids]
    res = [O.local_ba(w) for w in wins]
    stride = 7 * 4 + 1                                 # fixed-stride record: 4 camera poses + chi2
    rec = torch.zeros(wpg * stride, dtype=torch.float64)
    for k, r in enumerate(res):
        rec[k * stride:k * stride + 28] = torch.from_numpy(r.cam_qt.reshape(-1)); rec[k * stride + 28] = r.stats["chi2_final"]
    gathered = [torch.zeros_like(rec) for _ in range(world)] if rank == 0 else None
    dist.gather(rec, gathered, dst=0)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)           # the bench takes the MAX elapsed time over ranks
    assert float(t) == 0.5 + (world - 1)
    if rank == 0:
        allrec = torch.cat(gathered).numpy().reshape(world * wpg, stride)
        for wid in range(world * wpg):                 # every window of every rank arrived, in window-id order
            ref = O.local_ba(synth.make_lba_small(wid, n_free=3, n_fixed=1, n_points=40, n_lines=6))
            assert np.array_equal(allrec[wid, :28], ref.cam_qt.reshape(-1)) and allrec[wid, 28] == ref.stats["chi2_final"]
        print("GATHER_OK", world * wpg)
    dist.destroy_process_group()
''')


def test_two_rank_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "GATHER_OK 6" in out.stdout
