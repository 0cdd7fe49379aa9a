"""The symbolic side of the structure-following reduced solve (lld_slam_amd/csrc/lld_ba_chol_plan.h, `lld_ba_chol_plan` of the C ABI) without a GPU.

The plan a window gets is a set of tables that ba_chol_sparse_kernel executes blindly: which tile column(s) a step eliminates, which
register slots form L_IJ, take a trailing update or are published in that step, where a column's tiles sit in the LDS panel buffers.
`run_plan` below executes exactly those tables, tile by tile, in numpy - panel buffers and the diagonal-tile store are dictionaries, so a
tile that is read before it was published, updated after it was published or published twice fails loudly - and the result must equal
numpy.linalg.solve on the same system.  Stands in for the reference's trust in Eigen's analyzePattern / factorize pair
(linear_solver_eigen.h:94-124, :147-232)."""
import ctypes

import numpy as np
import pytest

from lld_slam_amd import abi, synth

MAXT, STRIDE, WAVES, SLOTS, POS, NONE = 22, 24, 10, 14, 20, 255          # kSpMaxT, kSpStride, kSpTileWaves, kSpSlots, kSpPos, kSpNone of lld_ba_chol_plan.h
_FIELDS = [("mode", "u1"), ("NT", "u1"), ("T", "u1"), ("chains", "u1"), ("cols", "u1", (STRIDE, 2)),
           ("slotI", "u1", (WAVES, SLOTS)), ("slotK", "u1", (WAVES, SLOTS)), ("pos", "u1", (MAXT, STRIDE)),
           ("cA", "<u4", (WAVES, STRIDE)), ("cB", "<u4", (WAVES, STRIDE)), ("dA", "<u4", (WAVES, STRIDE)), ("dB", "<u4", (WAVES, STRIDE)),
           ("pub", "<u4", (WAVES, STRIDE)), ("own", "<u4", (WAVES, STRIDE)), ("pub0", "<u4", (WAVES,)), ("padmask", "<u4", (WAVES,)), ("yrows", "<u4", (STRIDE, 2)),
           ("rowmap", "<i2", (MAXT * 16,)), ("n_tiles", "<i4"), ("n_updates", "<i4"), ("n_cams", "<i4"), ("est_ns", "<i4")]
_raw = np.dtype(_FIELDS, align=True)
PLAN = np.dtype(_FIELDS + [("tail_pad", "u1", ((-_raw.itemsize) % 16,))] if _raw.itemsize % 16 else _FIELDS, align=True)      # struct alignas(16)


def get_plan(nz, force=0):
    lib = abi.product()
    nf = nz.shape[0]
    size = ctypes.c_uint64(0)
    buf = np.zeros(1, PLAN)
    blk = np.ascontiguousarray(nz, np.uint8)
    fn = lib.fn("ba_chol_plan")
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    st = fn(nf, blk.ctypes.data, force, buf.ctypes.data, buf.nbytes, ctypes.byref(size))
    assert st == 0 and size.value == PLAN.itemsize, (st, size.value, PLAN.itemsize)
    return buf[0]


def bits(m):
    return [i for i in range(32) if (int(m) >> i) & 1]


def run_plan(P, S, b):
    """Executes the plan's tables the way ba_chol_sparse_kernel does; returns x in S's row order."""
    NT, T = int(P["NT"]), int(P["T"])
    N, n = 16 * NT, S.shape[0]
    rowmap = P["rowmap"][:N].astype(int)
    real = rowmap >= 0
    assert sorted(rowmap[real]) == list(range(n)), "the row map is not a permutation of S's rows"
    Sp = np.eye(N)
    Sp[np.ix_(real, real)] = S[np.ix_(rowmap[real], rowmap[real])]
    y = np.zeros(N); y[real] = b[rowmap[real]]
    cols, pos = P["cols"].astype(int), P["pos"].astype(int)
    tile = lambda M, I, K: M[16 * I:16 * I + 16, 16 * K:16 * K + 16]
    acc, where = {}, {}
    for w in range(WAVES):
        for sl in range(SLOTS):
            I, K = int(P["slotI"][w, sl]), int(P["slotK"][w, sl])
            if I == NONE:
                continue
            assert K <= I < NT and (I, K) not in where, "a tile twice in the slot tables"
            acc[(w, sl)] = tile(Sp, I, K).copy(); where[(I, K)] = (w, sl)
            touches_padding = (~real[16 * I:16 * I + 16]).any() or (~real[16 * K:16 * K + 16]).any()
            assert bool((int(P["padmask"][w]) >> sl) & 1) == bool(touches_padding)
    # every non-zero tile of S must be resident (or be a step-0 diagonal tile, which the panel wavefronts fetch themselves)
    for I in range(NT):
        for K in range(I + 1):
            if np.any(tile(Sp, I, K) != 0) and (I, K) not in where:
                assert I == K and K in cols[0], f"tile ({I},{K}) of S is non-zero and nobody holds it"
    Lp, Dall, published, frozen = [{}, {}], {}, set(), set()

    def factor(J):
        D = Dall[J]
        L = np.linalg.cholesky(np.tril(D) + np.tril(D, -1).T)
        Dall[J] = np.linalg.inv(L)

    def publish(masks, parity, forbidden_diag):
        Lp[parity] = {}
        for w in range(WAVES):
            for sl in bits(masks[w]):
                I, K = int(P["slotI"][w, sl]), int(P["slotK"][w, sl])
                assert (I, K) not in published, "published twice"
                published.add((I, K)); frozen.add((w, sl))
                if I == K:
                    assert I not in forbidden_diag, "a diagonal tile is published while a panel wavefront works on it"
                    Dall[I] = acc[(w, sl)].copy()
                else:
                    p = pos[K][I]
                    assert p != NONE and p < POS and p not in Lp[parity], "panel-buffer position missing or taken twice"
                    Lp[parity][p] = acc[(w, sl)].copy()

    for ch in range(2):
        J0 = cols[0][ch]
        if J0 != NONE:
            Dall[J0] = tile(Sp, J0, J0).copy(); factor(J0)
    publish(P["pub0"], 0, set(c for c in cols[0] if c != NONE))
    for s in range(T):
        par, (JA, JB) = s & 1, cols[s]
        nxt = set(c for c in cols[s + 1] if c != NONE)
        # panel wavefronts, before the barrier: their own L_(Jn)J (in place) and the update of the next diagonal tile
        panel_tiles = set()
        for ch in range(2):
            Jn = cols[s + 1][ch]
            if Jn == NONE:
                continue
            D = Dall[Jn]
            for J in (JA, JB):
                if J != NONE and pos[J][Jn] != NONE:
                    L = Lp[par][pos[J][Jn]] @ Dall[J].T
                    Lp[par][pos[J][Jn]] = L; panel_tiles.add((Jn, J))
                    D = D - L @ L.T
            Dall[Jn] = D
        own_tiles = set()
        for w in range(WAVES):                                      # (c) on the tile wavefronts: everything but the panel's tiles
            own = int(P["own"][w, s])
            assert own & ~(int(P["cA"][w, s]) | int(P["cB"][w, s])) == 0
            for sl in bits(own):
                own_tiles.add((int(P["slotI"][w, sl]), int(P["slotK"][w, sl])))
            for masks, J in ((P["cA"], JA), (P["cB"], JB)):
                for sl in bits(int(masks[w, s]) & ~own):
                    I, K = int(P["slotI"][w, sl]), int(P["slotK"][w, sl])
                    assert K == J and I > K and (I, K) not in panel_tiles
                    L = Lp[par][pos[K][I]] @ Dall[J].T
                    acc[(w, sl)] = L; Lp[par][pos[K][I]] = L
        assert own_tiles == panel_tiles, (own_tiles, panel_tiles)
        for J in (JA, JB):                                          # forward solution of the step's columns
            if J != NONE:
                y[16 * J:16 * J + 16] = Dall[J] @ y[16 * J:16 * J + 16]
        # ---- barrier
        for w in range(WAVES):
            for sl in bits(P["own"][w, s]):
                I, K = int(P["slotI"][w, sl]), int(P["slotK"][w, sl])
                acc[(w, sl)] = Lp[par][pos[K][I]].copy()
        for w in range(WAVES):                                      # (d) trailing updates
            for sl in sorted(set(bits(P["dA"][w, s])) | set(bits(P["dB"][w, s]))):
                assert (w, sl) not in frozen, "a tile is updated after it was published"
                I, K = int(P["slotI"][w, sl]), int(P["slotK"][w, sl])
                for masks, J in ((P["dA"], JA), (P["dB"], JB)):
                    if (int(masks[w, s]) >> sl) & 1:
                        acc[(w, sl)] = acc[(w, sl)] - Lp[par][pos[J][I]] @ Lp[par][pos[J][K]].T
        Lp_read = Lp[par]
        publish(P["pub"][:, s], (s + 1) & 1, nxt)
        ynew = y.copy()                                              # forward substitution of every row below, on the tile wavefronts
        for I in range(NT):
            for ch, J in ((0, JA), (1, JB)):
                if (int(P["yrows"][s, ch]) >> I) & 1:
                    ynew[16 * I:16 * I + 16] -= Lp_read[pos[J][I]] @ y[16 * J:16 * J + 16]
        y[:] = ynew
        for ch in range(2):                                          # panel wavefronts, after the barrier: the serial factor
            if cols[s + 1][ch] != NONE:
                factor(cols[s + 1][ch])
    x = np.zeros(N)
    for s in range(T - 1, -1, -1):
        for ch, masks in ((0, P["cA"]), (1, P["cB"])):
            J = cols[s][ch]
            if J == NONE:
                assert not any(int(masks[w, s]) for w in range(WAVES))
                continue
            sm = np.zeros(16)
            for w in range(WAVES):
                for sl in bits(masks[w, s]):
                    I = int(P["slotI"][w, sl])
                    sm += acc[(w, sl)].T @ x[16 * I:16 * I + 16]
            x[16 * J:16 * J + 16] = Dall[J].T @ (y[16 * J:16 * J + 16] - sm)
    xo = np.zeros(n)
    xo[rowmap[real]] = x[real]
    return xo


def spd_with_pattern(nz, seed):
    """A well-conditioned SPD matrix whose 6x6 blocks are dense exactly where `nz` says."""
    rng = np.random.default_rng(seed)
    nf = nz.shape[0]
    M = np.zeros((6 * nf, 6 * nf))
    for a in range(nf):
        for c in range(a):
            if nz[a, c] or nz[c, a]:
                B = rng.uniform(-1, 1, (6, 6))
                M[6 * a:6 * a + 6, 6 * c:6 * c + 6] = B; M[6 * c:6 * c + 6, 6 * a:6 * a + 6] = B.T
    M += np.diag(np.abs(M).sum(1) + rng.uniform(1, 2, 6 * nf))
    return M, rng.uniform(-1, 1, 6 * nf)


def band(nf, half):
    i = np.arange(nf)
    return np.abs(i[:, None] - i[None, :]) <= half


def window_pattern(w):
    nf = w.n_free_cams
    nz = np.eye(nf, dtype=bool)
    for start, cam in ((w.pt_obs_start, w.pt_obs_cam), (w.ln_obs_start, w.ln_obs_cam)):
        for l in range(len(start) - 1):
            cs = [c for c in cam[start[l]:start[l + 1]] if c < nf]
            for a in cs:
                nz[a, cs] = True
    return nz


def check(nz, seed=0, force=0, expect_mode=None):
    P = get_plan(nz, force)
    if expect_mode is not None:
        assert int(P["mode"]) == expect_mode
    if not P["mode"]:
        return P
    S, b = spd_with_pattern(nz, seed)
    x = run_plan(P, S, b)
    ref = np.linalg.solve(S, b)
    assert np.allclose(x, ref, rtol=1e-9, atol=1e-12), np.abs(x - ref).max()
    return P


def test_plan_layout_matches_the_library():
    P = get_plan(np.ones((3, 3), bool))
    assert int(P["mode"]) == 1 and int(P["NT"]) == 2 and int(P["T"]) == 2 and int(P["chains"]) == 1


@pytest.mark.parametrize("nf", [1, 2, 3, 5, 8, 11, 16, 20, 27, 37, 50])
@pytest.mark.parametrize("force", [0, 1, 2])
def test_banded_systems_every_size_and_every_kind_of_plan(nf, force):
    for half in (1, 3, 7):
        check(band(nf, half), seed=nf * 10 + half, force=force)


def test_the_metric_window_gets_two_chains():
    """LBA-B (50 keyframes, block bandwidth 13, a quarter dense): two chains that meet in a separator, 12 or 13 steps instead of 19."""
    nz = window_pattern(synth.make_lba_b(0))
    P = check(nz, seed=1, expect_mode=1)
    assert int(P["chains"]) == 2 and int(P["T"]) <= 13, (int(P["chains"]), int(P["T"]))
    one = check(nz, seed=1, force=1, expect_mode=1)
    assert int(one["chains"]) == 1 and int(one["T"]) == 19 and int(one["n_tiles"]) < 100


def test_dense_systems_go_to_the_dense_kernel_when_they_do_not_fit():
    assert int(check(np.ones((50, 50), bool), expect_mode=0)["mode"]) == 0        # 190 tiles: more than the 140 register slots
    P = check(np.ones((40, 40), bool), seed=3, expect_mode=1)                        # 120 tiles: fits, one chain
    assert int(P["chains"]) == 1 and int(P["NT"]) == 15


def test_two_unconnected_camera_groups_are_two_chains_without_a_separator():
    nz = np.zeros((50, 50), bool)
    nz[:25, :25] = True; nz[25:, 25:] = True
    P = check(nz, seed=4, expect_mode=1)
    assert int(P["chains"]) == 2 and int(P["T"]) == 10


@pytest.mark.parametrize("seed", range(40))
def test_random_covisibility_graphs(seed):
    """Random structures: bands with long-range links, star cameras (a keyframe that sees everything), shuffled camera orders."""
    rng = np.random.default_rng(1000 + seed)
    nf = int(rng.integers(2, 51))
    nz = band(nf, int(rng.integers(1, 9)))
    for _ in range(int(rng.integers(0, 4))):
        a, c = rng.integers(0, nf, 2); nz[a, c] = nz[c, a] = True
    if seed % 5 == 0:
        star = int(rng.integers(0, nf)); nz[star, :] = True; nz[:, star] = True
    if seed % 3 == 0:
        perm = rng.permutation(nf); nz = nz[np.ix_(perm, perm)]
    if seed % 7 == 0:
        nz &= rng.uniform(size=nz.shape) < 0.7; nz |= nz.T; np.fill_diagonal(nz, True)
    for force in (0, 1, 2):
        check(nz, seed=seed, force=force)


def test_a_shuffled_band_is_found_again():
    """Reverse Cuthill-McKee recovers the band a shuffled camera order hides: the plan is as short as for the ordered window."""
    rng = np.random.default_rng(7)
    nz = band(50, 6)
    ordered = check(nz, seed=5)
    perm = rng.permutation(50)
    shuffled = check(nz[np.ix_(perm, perm)], seed=6)
    assert int(shuffled["mode"]) == 1 and int(shuffled["T"]) <= int(ordered["T"]) + 2
