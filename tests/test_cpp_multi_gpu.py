"""The multi-GPU split behind the C ABI (lld_ba_multi_*, lld_slam_amd/csrc/lld_ba_multi.hip) driven by a compiled C++ host,
examples/multi_gpu_harness.cpp: one process, one host thread + one context per shard, block partition, gather of the result records on the
first device.  CPU: the partition equals lld_slam_amd/dist.py's shard(strong=True) and the harness fails loudly
without a device.  GPU: every visible device - and, on a 1-GPU box, two and three shards on device 0 - against the same shards solved
as plain batches through the Python mirror, bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from lld_slam_amd import abi, dist as D, synth
from test_cpp_harness import BA_ARRAYS, read_ba

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "examples", "multi_gpu_harness")


@pytest.fixture(scope="module")
def harness():
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    assert os.path.exists(HARNESS)
    return HARNESS


def write_windows(path, ws, gamma=1.0):
    with open(path, "wb") as f:
        np.array([len(ws)], np.int32).tofile(f)
        for w in ws:
            np.array([w.n_cams, w.n_free_cams, w.n_points, w.n_pt_obs, w.n_lines, w.n_ln_obs, 0, 0], np.int32).tofile(f)
            np.concatenate([np.asarray(w.cam, np.float64).reshape(-1)[:5], [gamma]]).astype(np.float64).tofile(f)
            for k, t in BA_ARRAYS:
                np.ascontiguousarray(getattr(w, k), t).tofile(f)


def read_outputs(path, ws):
    outs = []
    with open(path, "rb") as f:
        for w in ws:
            outs.append({"cam_qt": np.fromfile(f, np.float64, 7 * w.n_cams).reshape(-1, 7), "pt_xyz": np.fromfile(f, np.float64, 3 * w.n_points).reshape(-1, 3),
                         "line_x0": np.fromfile(f, np.float64, 3 * w.n_lines).reshape(-1, 3), "line_dir": np.fromfile(f, np.float64, 3 * w.n_lines).reshape(-1, 3),
                         "pt_obs_outlier": np.fromfile(f, np.uint8, w.n_pt_obs), "ln_edge_outlier": np.fromfile(f, np.uint8, 2 * w.n_ln_obs).reshape(-1, 2),
                         "line_removed": np.fromfile(f, np.uint8, w.n_lines), "chi2": np.fromfile(f, np.float64, 2), "st": np.fromfile(f, np.int32, 4)})
        ns = int(np.fromfile(f, np.int32, 1)[0])
        shards = np.fromfile(f, np.int32, 2 * ns).reshape(-1, 2)
        verified = int(np.fromfile(f, np.int32, 1)[0])
    return outs, shards, verified


def test_partition_is_the_block_partition_of_dist_shard():
    lib = abi.product()
    fn = lib.fn("ba_multi_shard"); fn.restype = None
    fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    for n in (1, 2, 7, 8, 31, 32, 100, 255, 256, 257, 1000):
        for parts in (1, 2, 3, 4, 7, 8):
            covered = 0
            for p in range(parts):
                first, count = C.c_int32(-1), C.c_int32(-1)
                fn(n, parts, p, C.byref(first), C.byref(count))
                assert (first.value, count.value) == D.shard(n, parts, p, True)
                assert first.value == covered
                covered += count.value
            assert covered == n
    first, count = C.c_int32(-1), C.c_int32(-1)
    fn(10, 4, 4, C.byref(first), C.byref(count))                  # a part that does not exist owns nothing
    assert count.value == 0


def test_harness_refuses_without_gpu(harness, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    write_windows(tmp_path / "in.bin", [synth.make_lba_small(0)])
    r = subprocess.run([harness, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    assert abi.product().fn("device_count")() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["all", "0,0", "0,0,0"])
def test_shards_solve_and_gather_like_their_own_batches(harness, tmp_path, gpu_ctx, devices):
    """Seven ragged windows over 1 (every visible device), 2 and 3 shards: the harness' results per window equal the same shard solved as a
    plain batch (same batch size -> same kernels and chunking -> same bits), the partition it reports is shard()'s, and every gathered
    record was verified on the first device."""
    from lld_slam_amd import BABatch
    ws = [synth.make_lba_small(500 + i, n_free=4 + i, n_fixed=2, n_points=150 + 40 * i, n_lines=20 + 6 * i) for i in range(7)]
    write_windows(tmp_path / "in.bin", ws)
    r = subprocess.run([harness, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), devices, "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("7 gathered records verified") == 2, r.stdout
    outs, shards, verified = read_outputs(tmp_path / "out.bin", ws)
    n_shards = abi.product().fn("device_count")() if devices == "all" else devices.count(",") + 1
    assert verified == 7 and len(shards) == n_shards
    for p, (first, count) in enumerate(shards):
        assert (int(first), int(count)) == D.shard(7, n_shards, p, True)
        with BABatch(gpu_ctx, ws[first:first + count]) as b:
            b.solve()
            ref = b.download_all()
        for k, g in enumerate(ref):
            o = outs[first + k]
            for name in ("cam_qt", "pt_xyz", "line_x0", "line_dir", "pt_obs_outlier", "ln_edge_outlier", "line_removed"):
                np.testing.assert_array_equal(o[name], getattr(g, name), err_msg=f"window {first + k} {name}")
            assert o["chi2"][1] == g.stats["chi2_final"] and list(o["st"][:2]) == list(g.stats["lm_iterations"])
