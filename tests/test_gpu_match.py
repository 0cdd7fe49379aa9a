"""GPU parity: descriptor matching through the C ABI vs the CPU oracle — indices and integer distances bit-exact."""
import numpy as np
import pytest

from lld_slam_amd import ORBmatcher, TwoFrameLineMatcher, host, synth

pytestmark = pytest.mark.gpu


def _same(a, b):
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("nq,nt", [(2000, 2000), (1, 1), (130, 7), (64, 257), (129, 1025)])
def test_hamming_brute_force_bit_exact(gpu_ctx, oracle, nq, nt):
    q, t = synth.make_match_orb(1, nq, nt, n_corr=int(0.8 * min(nq, nt)), n_dup=16)
    _same(ORBmatcher(gpu_ctx).BestTwo(q, t), oracle.match_hamming256(q, t))


def test_hamming_ties_pick_the_lowest_index(gpu_ctx, oracle):
    rng = np.random.default_rng(5)
    q = rng.integers(0, 2 ** 32, (70, 8), dtype=np.uint64).astype(np.uint32)
    t = np.repeat(q[:5], 60, axis=0)                       # every query 0..4 has 60 exact duplicates
    t = t[rng.permutation(t.shape[0])]
    bi, bd, si, sd = ORBmatcher(gpu_ctx).BestTwo(q, t)
    _same((bi, bd, si, sd), oracle.match_hamming256(q, t))
    for i in range(5):
        dup = np.nonzero((t == q[i]).all(1))[0]
        assert bi[i] == dup[0] and si[i] == dup[1] and bd[i] == 0 and sd[i] == 0


def test_hamming_extremes(gpu_ctx, oracle):
    z = np.zeros((3, 8), np.uint32); o = np.full((2, 8), 0xFFFFFFFF, np.uint32)
    bi, bd, si, sd = ORBmatcher(gpu_ctx).BestTwo(z, o)
    assert bd.tolist() == [256] * 3 and bi.tolist() == [0] * 3 and si.tolist() == [1] * 3
    # no train rows at all -> unmatched sentinel of the reference (bestDist = 256, no index)
    bi, bd, si, sd = ORBmatcher(gpu_ctx).BestTwo(z, np.zeros((0, 8), np.uint32))
    assert bi.tolist() == [-1] * 3 and bd.tolist() == [256] * 3


def test_hamming_mask(gpu_ctx, oracle):
    q, t = synth.make_match_orb(2, 300, 500, n_corr=200)
    rng = np.random.default_rng(6)
    mask = (rng.random((300, 500)) < 0.05).astype(np.uint8)
    mask[7] = 0                                             # a query without candidates
    res = ORBmatcher(gpu_ctx).BestTwo(q, t, mask)
    _same(res, oracle.match_hamming256(q, t, mask))
    assert res[0][7] == -1 and res[1][7] == 256


def test_hamming_candidate_lists_first_in_list_wins(gpu_ctx, oracle):
    q, t = synth.make_match_orb(3, 400, 600, n_corr=300)
    t[10] = t[500]                                          # duplicate rows: list order decides
    rng = np.random.default_rng(7)
    lens = rng.integers(0, 40, 400)
    cs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = rng.integers(0, 600, cs[-1]).astype(np.int32)
    ci[cs[3]:cs[3] + 2] = [500, 10] if lens[3] >= 2 else ci[cs[3]:cs[3] + 2]
    res = ORBmatcher(gpu_ctx).BestTwoCandidates(q, t, cs, ci)
    _same(res, oracle.match_hamming256_csr(q, t, cs, ci))


def test_orb_accept_rule_matches_reference_semantics(gpu_ctx, oracle):
    q, t = synth.make_match_orb(4, 500, 500, n_corr=400)
    m = ORBmatcher(gpu_ctx, nnratio=0.7)
    bi, bd, si, sd = m.BestTwo(q, t)
    acc = m.AcceptByRatio(bi, bd, sd, ORBmatcher.TH_LOW)
    obi, obd, osi, osd = oracle.match_hamming256(q, t)
    exp = np.where((obd <= 50) & (obd.astype(np.float32) < np.float32(0.7) * osd.astype(np.float32)), obi, -1)
    np.testing.assert_array_equal(acc, exp)
    assert (acc >= 0).sum() > 300


@pytest.mark.parametrize("nq,nt,dim", [(300, 300, 72), (5, 3, 72), (257, 33, 32), (64, 100, 128)])
def test_l2_best2_bit_exact(gpu_ctx, oracle, nq, nt, dim):
    q, t = synth.make_match_lbd(1, nq, nt, dim, n_corr=int(0.8 * min(nq, nt)))
    g = TwoFrameLineMatcher(gpu_ctx, 2.0).BestTwo(q, t)
    o = oracle.match_l2f32(q, t)
    np.testing.assert_array_equal(g[0], o[0]); np.testing.assert_array_equal(g[2], o[2])
    # distances: identical accumulation order; the final sqrt is correctly rounded on both sides
    np.testing.assert_array_equal(g[1], o[1]); np.testing.assert_array_equal(g[3], o[3])


def test_l2_mask(gpu_ctx, oracle):
    q, t = synth.make_match_lbd(2, 100, 120, 72, n_corr=80)
    mask = (np.random.default_rng(8).random((100, 120)) < 0.1).astype(np.uint8)
    g = TwoFrameLineMatcher(gpu_ctx, 2.0).BestTwo(q, t, mask)
    o = oracle.match_l2f32(q, t, mask)
    np.testing.assert_array_equal(g[0], o[0]); np.testing.assert_array_equal(g[2], o[2])


def test_line_greedy_matches_sequential_reference(gpu_ctx, oracle):
    q, t = synth.make_match_lbd(3, 300, 300, 72, n_corr=240)
    t[5] = t[17]                                            # duplicates: order dependence matters (hazard 12)
    q[40] = q[41]
    rng = np.random.default_rng(9)
    gate = (rng.random((300, 300)) < 0.6).astype(np.uint8)
    for tau in (2.0, 0.5):
        gm, gd = TwoFrameLineMatcher(gpu_ctx, tau).MatchLines(q, t, gate)
        om, od = oracle.line_match_greedy(q, t, gate, tau)
        np.testing.assert_array_equal(gm, om)
        np.testing.assert_array_equal(gd[gm >= 0], od[om >= 0])
        used = gm[gm >= 0]
        assert len(set(used.tolist())) == used.size          # each right line taken at most once
    gm, _ = TwoFrameLineMatcher(gpu_ctx, 2.0).MatchLines(q, t, None)
    np.testing.assert_array_equal(gm, oracle.line_match_greedy(q, t, None, 2.0)[0])


def test_line_greedy_exhausted_candidate_lists_fall_back_to_the_row_scan(gpu_ctx, oracle):
    """The device keeps the 8 best candidates per left line; 20 identical left lines compete for the same right lines, so from the
    9th on every listed candidate is taken and the stored row is scanned like the reference does."""
    q, t = synth.make_match_lbd(4, 120, 150, 72, n_corr=100)
    q[10:30] = q[10]
    q[60:75] = q[61]
    gate = np.ones((120, 150), np.uint8); gate[12, :] = 0; gate[:, 7] = 0
    for tau in (1e9, 1.6):
        gm, gd = TwoFrameLineMatcher(gpu_ctx, tau).MatchLines(q, t, gate)
        om, od = oracle.line_match_greedy(q, t, gate, tau)
        np.testing.assert_array_equal(gm, om)
        np.testing.assert_array_equal(gd[gm >= 0], od[om >= 0])
    assert (gm[10:30] >= 0).sum() >= 18


def test_line_greedy_many_right_lines(gpu_ctx, oracle):
    """9000 right lines: the row of distances (72 KB) needs more than the default dynamic LDS."""
    q, t = synth.make_match_lbd(5, 40, 9000, 32, n_corr=30)
    gm, gd = TwoFrameLineMatcher(gpu_ctx, 2.0).MatchLines(q, t, None)
    om, od = oracle.line_match_greedy(q, t, None, 2.0)
    np.testing.assert_array_equal(gm, om)
    np.testing.assert_array_equal(gd[gm >= 0], od[om >= 0])


def test_batched_hamming_device_entry_point(gpu_ctx, oracle):
    import torch
    B, nq, nt = 6, 333, 450
    qs, ts = zip(*[synth.make_match_orb(10 + b, nq, nt, n_corr=300) for b in range(B)])
    dev = torch.device("cuda", gpu_ctx.device)
    q = torch.from_numpy(np.stack(qs).view(np.int32)).to(dev); t = torch.from_numpy(np.stack(ts).view(np.int32)).to(dev)
    outs = [torch.empty((B, nq), dtype=torch.int32, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    st = gpu_ctx.lib.fn("match_hamming256_batch_dev")(gpu_ctx.handle, B, q.data_ptr(), nq, t.data_ptr(), nt,
                                                      *[o.data_ptr() for o in outs])
    assert st == 0
    gpu_ctx.synchronize()
    for b in range(B):
        exp = oracle.match_hamming256(qs[b], ts[b])
        for o, e in zip(outs, exp):
            np.testing.assert_array_equal(o[b].cpu().numpy(), e)


@pytest.mark.parametrize("B,nq,nt,dim", [(5, 300, 300, 72), (3, 65, 131, 72), (4, 257, 33, 32), (2, 70, 45, 40), (2, 64, 100, 128), (3, 10, 7, 7),
                                         (2, 1, 300, 72), (2, 129, 1, 72)])
def test_batched_l2_device_entry_point(gpu_ctx, oracle, B, nq, nt, dim):
    """lld_match_l2f32_batch_dev (the MATCH / LBD config's kernel): descriptor lengths with a compile-time form (72, 32) and without,
    query counts around the 64-lane workgroup, train counts around the 64-row tile - every pair bit-exact against the oracle, all four
    outputs."""
    import torch
    dev = torch.device("cuda", gpu_ctx.device)
    rng = np.random.default_rng(1000 * B + nq + dim)
    if dim == 72 and nq >= 65:
        qs, ts = zip(*[synth.make_match_lbd(20 + b, nq, nt, dim, n_corr=int(0.5 * min(nq, nt))) for b in range(B)])
    else:
        qs = [rng.normal(size=(nq, dim)).astype(np.float32) for _ in range(B)]; ts = [rng.normal(size=(nt, dim)).astype(np.float32) for _ in range(B)]
        ts[0][nt // 2] = ts[0][0]                                                       # a tie: the lower index must win
    q = torch.from_numpy(np.stack(qs)).to(dev); t = torch.from_numpy(np.stack(ts)).to(dev)
    bi = torch.empty((B, nq), dtype=torch.int32, device=dev); si = torch.empty_like(bi)
    bd = torch.empty((B, nq), dtype=torch.float64, device=dev); sd = torch.empty_like(bd)
    torch.cuda.synchronize()
    assert gpu_ctx.lib.fn("match_l2f32_batch_dev")(gpu_ctx.handle, B, q.data_ptr(), nq, t.data_ptr(), nt, dim, bi.data_ptr(), bd.data_ptr(),
                                                   si.data_ptr(), sd.data_ptr()) == 0
    gpu_ctx.synchronize()
    for b in range(B):
        ebi, ebd, esi, esd = oracle.match_l2f32(qs[b], ts[b])
        np.testing.assert_array_equal(bi[b].cpu().numpy(), ebi); np.testing.assert_array_equal(bd[b].cpu().numpy(), ebd)
        if nt > 1:
            np.testing.assert_array_equal(si[b].cpu().numpy(), esi); np.testing.assert_array_equal(sd[b].cpu().numpy(), esd)


@pytest.mark.parametrize("seed,nl,nr", [(0, 300, 300), (1, 257, 130), (2, 64, 400)])
def test_stereo_line_association_gates_on_device(gpu_ctx, oracle, seed, nl, nr):
    """TwoFrameLineMatcher::MatchLines incl. CheckLinePair's triangulation / depth gates (src/TwoFrameLineMatcher.cc:26-124): the gate
    matrix and the greedy assignment equal the literal CPU restatement."""
    s = synth.make_stereo_lines(seed, nl, nr)
    tm = TwoFrameLineMatcher(gpu_ctx, 2.0, K=s["K"], b=s["b"], minLineLength=20)
    m, d, gate = tm.MatchLines(s["desc_left"], s["desc_right"], lines=s["left"], other_lines=s["right"], octaves=s["left_octave"],
                               other_octaves=s["right_octave"], want_gate=True)
    me, de, ge = oracle.line_match_stereo(s["K"], s["b"], 2.0, 20, s["left"], s["left_octave"], s["desc_left"], s["right"], s["right_octave"],
                                          s["desc_right"], want_gate=True)
    np.testing.assert_array_equal(gate, ge)
    np.testing.assert_array_equal(m, me)
    np.testing.assert_array_equal(d[m >= 0], de[me >= 0])
    assert (m >= 0).sum() > min(nl, nr) // 4


def test_stereo_line_association_edge_cases(gpu_ctx, oracle):
    s = synth.make_stereo_lines(3, 20, 20)
    tm = TwoFrameLineMatcher(gpu_ctx, 2.0, K=s["K"], b=s["b"], minLineLength=20)
    m, d = tm.MatchLines(s["desc_left"], np.zeros((0, 72), np.float32), lines=s["left"], other_lines=np.zeros((0, 4), np.float32),
                         octaves=s["left_octave"], other_octaves=np.zeros(0, np.int32))
    assert (m == -1).all()
    # a minimum length nobody reaches: every gate closed
    tm = TwoFrameLineMatcher(gpu_ctx, 2.0, K=s["K"], b=s["b"], minLineLength=100000)
    m, d, gate = tm.MatchLines(s["desc_left"], s["desc_right"], lines=s["left"], other_lines=s["right"], octaves=s["left_octave"],
                               other_octaves=s["right_octave"], want_gate=True)
    assert (m == -1).all() and not gate.any()


@pytest.mark.parametrize("dim", [1, 3, 7, 31, 33, 71, 73, 127, 128])
def test_l2_odd_descriptor_lengths(gpu_ctx, oracle, dim):
    """Descriptor lengths that are not multiples of the vector load width (LBD variants: 8 x bands + ...)."""
    rng = np.random.default_rng(dim)
    q = rng.normal(size=(70, dim)).astype(np.float32); t = rng.normal(size=(45, dim)).astype(np.float32)
    t[7] = q[3]; t[9] = q[3]                                # an exact tie: the lower index wins
    _same(TwoFrameLineMatcher(gpu_ctx, 2.0).BestTwo(q, t), oracle.match_l2f32(q, t))
    m, d = TwoFrameLineMatcher(gpu_ctx, 1e9).MatchLines(q, t)
    me, de = oracle.line_match_greedy(q, t, None, 1e9)
    np.testing.assert_array_equal(m, me); np.testing.assert_array_equal(d, de)


def test_l2_rejects_descriptors_longer_than_128(gpu_ctx):
    q = np.zeros((4, 129), np.float32)
    with pytest.raises(RuntimeError):
        TwoFrameLineMatcher(gpu_ctx, 2.0).BestTwo(q, q)
