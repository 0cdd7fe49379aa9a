"""N>1 path on CPU: window sharding and the final gather of fixed-stride result records, world_size 2 over gloo - and, since round 4, the
same for the two other shardable legs of the bench line (PoseOptimization frames, frame-pair match batches: dist.gather_rows).

The workers import `lld_slam_amd.dist` - the module bench.py runs (shard, record layout, the staged asynchronous gather, the MAX over
ranks of the elapsed time) - and drive it with host tensors; the records are packed from CPU-oracle results in the library's record
layout (there is no GPU in this container).  tests/test_gpu_ba.py checks the same layout against the HIP library's buffer."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_py as O
    from lld_slam_amd import synth, dist as D
    strong = %(strong)r; n = %(n)d
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    first, count = D.shard(n, world, rank, strong)
    make = lambda i: synth.make_lba_small(i, n_free=3, n_fixed=1 + i %% 2, n_points=40 + 7 * i, n_lines=6 + i)      # ragged: the stride is the largest record
    total = n if strong else n * world
    everyone = [make(i) for i in range(total)]
    mine = everyone[first:first + count]
    my_stride = D.record_stride(mine)                   # a batch's stride comes from ITS windows (what lld_ba_batch_result_records reports) ...
    stride = D.max_count_stride(my_stride, torch.device("cpu"), True)      # ... the gather needs ONE: the largest over the ranks
    assert stride == D.record_stride(everyone) and (rank == world - 1 or my_stride < stride)     # the ragged windows do make the ranks differ
    res = [O.local_ba(w) for w in mine]
    records = torch.from_numpy(np.concatenate([D.pack_record(r, w, my_stride, k) for k, (r, w) in enumerate(zip(res, mine))])) if count else torch.zeros(0, dtype=torch.uint8)
    counts = D.gather_counts(count, torch.device("cpu"), True, world)
    assert sum(counts) == total and counts[rank] == count
    g = D.RecordGather(records, world, rank, n_bytes=max(counts) * stride, local_stride=my_stride, common_stride=stride)
    for step in range(2):                               # two steps: the second waits for the first gather in flight, as in bench.py
        g.step()
    g.drain()
    D.barrier(True, False)
    assert D.max_over_ranks(0.5 + rank, torch.device("cpu"), True) == 0.5 + (world - 1)
    # the agreement every leg of bench.sharded_secondary starts with: one rank's failure is every rank's "skip the collectives"
    assert D.all_ranks_ok(True, torch.device("cpu"), True) is True
    assert D.all_ranks_ok(rank != 1, torch.device("cpu"), True) is False
    # ---- the other two shardable legs of the bench line: PoseOptimization frames and frame-pair match batches (fixed-stride rows)
    nfr = 7; f0, fc = D.shard(nfr, world, rank, True)
    frames = [synth.make_pose_frame(100 + i, n_points=60, n_lines=12) for i in range(nfr)]
    po = [O.pose_opt(f, gamma=0.5) for f in frames[f0:f0 + fc]]
    rows = torch.from_numpy(np.array([list(r.pose_qt) + [float(r.n_inliers)] for r in po], np.float64).reshape(fc, 8))
    fcounts = D.gather_counts(fc, torch.device("cpu"), True, world)
    got_po = D.gather_rows(rows, fcounts, world, rank)
    npairs = 5; p0, pc = D.shard(npairs, world, rank, True)
    pairs = [synth.make_match_orb(200 + i, 40, 50) for i in range(npairs)]
    mrows = torch.from_numpy(np.stack([np.stack(O.match_hamming256(q, t), 1) for q, t in pairs[p0:p0 + pc]]).astype(np.int32)) if pc else torch.zeros((0, 40, 4), dtype=torch.int32)
    pcounts = D.gather_counts(pc, torch.device("cpu"), True, world)
    got_m = D.gather_rows(mrows, pcounts, world, rank)
    if rank == 0:
        allpo = torch.cat(got_po).numpy(); allm = torch.cat(got_m).numpy()
        assert allpo.shape == (nfr, 8) and allm.shape == (npairs, 40, 4)
        for i, f in enumerate(frames):
            r = O.pose_opt(f, gamma=0.5)
            assert np.array_equal(allpo[i, :7], r.pose_qt) and allpo[i, 7] == r.n_inliers
        for i, (q, t) in enumerate(pairs):
            assert np.array_equal(allm[i], np.stack(O.match_hamming256(q, t), 1))
        # every rank's records are the windows shard() gave it (header index + edge count + fixed cameras against the generator)
        assert D.verify_gathered_records(g.rank_records, counts, stride, n, world, strong, make) == sum(min(2, c) for c in counts)
        swapped = lambda r: g.rank_records(1 - r)
        try:
            D.verify_gathered_records(swapped, counts[::-1], stride, n, world, strong, make); raise SystemExit("a swapped gather went unnoticed")
        except AssertionError:
            pass
        wid = 0
        for r in range(world):
            buf = g.rank_records(r)
            for k in range(counts[r]):
                w = everyone[wid]
                out = D.unpack_record(buf[k * stride:(k + 1) * stride], w)
                ref = O.local_ba(w)
                assert np.array_equal(out.cam_qt, ref.cam_qt) and np.array_equal(out.pt_xyz, ref.pt_xyz) and np.array_equal(out.line_x0, ref.line_x0)
                assert np.array_equal(out.pt_obs_outlier, ref.pt_obs_outlier) and np.array_equal(out.ln_edge_outlier, ref.ln_edge_outlier)
                assert np.array_equal(out.line_removed, ref.line_removed)
                for key in ("chi2_final", "chi2_round1", "lm_iterations", "lm_trials", "aborted", "n_pt_obs_outlier", "n_lines_removed"):
                    assert out.stats[key] == ref.stats[key], key
                wid += 1
        assert wid == total
        print("GATHER_OK", total, counts)
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("strong,n,port,expect", [(False, 3, 29533, "GATHER_OK 6 [3, 3]"), (True, 5, 29534, "GATHER_OK 5 [2, 3]")])
def test_two_rank_shard_and_gather(tmp_path, strong, n, port, expect):
    """weak: 3 windows on each of 2 ranks; strong: 5 windows split 2 + 3 (uneven shards pad to the common gather size)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "strong": strong, "n": n})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert expect in out.stdout


def test_shard_and_thread_budget():
    from lld_slam_amd import dist as D
    assert [D.shard(256, 8, r, False) for r in (0, 7)] == [(0, 256), (1792, 256)]
    blocks = [D.shard(256, 8, r, True) for r in range(8)]
    assert blocks == [(32 * r, 32) for r in range(8)]
    odd = [D.shard(10, 4, r, True) for r in range(4)]
    assert sum(c for _, c in odd) == 10 and [f for f, _ in odd] == [0, 2, 5, 7] and all(odd[i][0] + odd[i][1] == odd[i + 1][0] for i in range(3))
    with pytest.raises(ValueError):
        D.shard(4, 2, 2, True)
    cores = os.cpu_count() or 1
    assert D.host_thread_budget(1) == max(1, min(16, cores - 1)) and D.host_thread_budget(10 ** 6) == 1
    assert D.host_thread_budget(8) * 8 <= max(8, cores)


def test_record_pack_unpack_round_trip(oracle):
    from lld_slam_amd import synth, dist as D
    w = synth.make_lba_small(3)
    r = oracle.local_ba(w)
    stride = D.record_stride([w, synth.make_lba_small(4, n_points=500)])
    assert stride % 256 == 0 and stride >= D.record_bytes(w)
    back = D.unpack_record(D.pack_record(r, w, stride), w)
    for a, b in ((back.cam_qt, r.cam_qt), (back.pt_xyz, r.pt_xyz), (back.line_dir, r.line_dir), (back.ln_edge_outlier, r.ln_edge_outlier), (back.line_removed, r.line_removed)):
        np.testing.assert_array_equal(a, b)
    assert back.stats["chi2_final"] == r.stats["chi2_final"] and back.stats["lm_trials"] == r.stats["lm_trials"]
