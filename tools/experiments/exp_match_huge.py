"""Brute-force matchers far beyond a frame's size (20 000 x 20 000 ORB descriptors, 1 x 100 000, 100 000 x 1, 5000 x 5000 LBD): device vs oracle,
every output bit for bit.   python tools/exp_match_huge.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from lld_slam_amd import Context, ORBmatcher, TwoFrameLineMatcher
import oracle_py as O


def main():
    ctx = Context(0); O.lib()
    rng = np.random.default_rng(5)
    for nq, nt in ((20000, 20000), (1, 100000), (100000, 1), (65537, 63), (3, 65537)):
        q = rng.integers(0, 2 ** 32, (nq, 8), dtype=np.uint64).astype(np.uint32); t = rng.integers(0, 2 ** 32, (nt, 8), dtype=np.uint64).astype(np.uint32)
        k = min(nq, nt) // 2
        if k: t[:k] = q[:k] ^ (rng.random((k, 8)) < 0.3).astype(np.uint32)             # some close pairs
        t0 = time.time(); g = ORBmatcher(ctx).BestTwo(q, t); t1 = time.time(); o = O.match_hamming256(q, t); t2 = time.time()
        print("hamming", nq, "x", nt, "equal", all(np.array_equal(a, b) for a, b in zip(g, o)), "device %.3f s oracle %.1f s" % (t1 - t0, t2 - t1), flush=True)
    for nq, nt, dim in ((5000, 5000, 72), (1, 50000, 72), (50000, 1, 32), (4097, 129, 128)):
        q = rng.normal(size=(nq, dim)).astype(np.float32); t = rng.normal(size=(nt, dim)).astype(np.float32)
        t0 = time.time(); g = TwoFrameLineMatcher(ctx, 2.0).BestTwo(q, t); t1 = time.time(); o = O.match_l2f32(q, t); t2 = time.time()
        same = np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]) and (nt < 2 or (np.array_equal(g[2], o[2]) and np.array_equal(g[3], o[3])))
        print("l2", nq, "x", nt, "x", dim, "equal", same, "device %.3f s oracle %.1f s" % (t1 - t0, t2 - t1), flush=True)


if __name__ == "__main__":
    main()
