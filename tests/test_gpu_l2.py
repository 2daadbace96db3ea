"""GPU parity tests of float-descriptor (L2 / squared L2) matching: the fp32 MFMA
candidate pass + sequential re-score through the C ABI, against the golden
vectors and the CPU oracle.

Bar: indices bit-exact; distances bit-exact as well, because the re-score pass
recomputes them in the oracle's operation order (sequential float32 sum, FP
contraction off on both sides).
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from sfm_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def _tuples(m):
    return [(int(a["queryIdx"]), int(a["trainIdx"]), float(a["distance"])) for a in m]


def test_l2_integer_valued_golden(ctx):
    from sfm_amd import capi

    g = load_golden("l2.npz")
    idx, dist = ctx.knn2(capi.NORM_L2SQR, g["int_q"], g["int_t"])
    np.testing.assert_array_equal(idx, g["int_idx"])
    np.testing.assert_array_equal(dist, g["int_sqr"])
    idx, dist = ctx.knn2(capi.NORM_L2, g["int_q"], g["int_t"])
    np.testing.assert_array_equal(idx, g["int_idx_l2"])
    np.testing.assert_array_equal(dist, g["int_l2"])


def test_l2_real_valued_golden(ctx):
    from sfm_amd import capi

    g = load_golden("l2.npz")
    idx, dist = ctx.knn2(capi.NORM_L2SQR, g["real_q"], g["real_t"])
    d64 = g["real_sqr64"]
    clear = (d64[:, 1] - d64[:, 0] > 1e-5) & (d64[:, 2] - d64[:, 1] > 1e-5)
    np.testing.assert_array_equal(idx[clear], g["real_idx"][clear])
    np.testing.assert_allclose(dist, d64[:, :2], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("nq,nt,dim,integer", [
    (2000, 2500, 128, True), (2000, 2500, 128, False), (1000, 900, 64, True),
    (700, 1300, 64, False), (300, 400, 100, False), (257, 129, 200, False),
    (64, 5, 128, False), (50, 2, 64, True), (5000, 4000, 128, False), (10, 3000, 36, True),
])
@pytest.mark.parametrize("squared", [True, False])
def test_l2_knn2_vs_oracle(ctx, oracle, nq, nt, dim, integer, squared):
    from sfm_amd import capi, synth

    q, t = synth.float_descriptors(nq, nt, dim=dim, seed=nq + nt + dim, integer=integer)
    if nt > 20:
        t[nt // 3] = t[2]   # duplicate train rows: exact distance ties
        t[nt // 3 + 1] = t[2]
        q[1] = t[4]         # zero distance
    norm = capi.NORM_L2SQR if squared else capi.NORM_L2
    idx, dist = ctx.knn2(norm, q, t)
    oidx, odist = oracle.knn2_l2(q, t, squared=squared)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)


def test_l2_tiny_and_empty(ctx, oracle):
    from sfm_amd import capi

    rng = np.random.default_rng(3)
    q = rng.standard_normal((9, 64)).astype(np.float32)
    t = rng.standard_normal((1, 64)).astype(np.float32)
    idx, dist = ctx.knn2(capi.NORM_L2, q, t)
    oidx, odist = oracle.knn2_l2(q, t, squared=False)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)
    idx, dist = ctx.knn2(capi.NORM_L2, q, np.zeros((0, 64), np.float32))
    assert (idx == -1).all()
    assert len(ctx.ratio_matches(capi.NORM_L2, q, t, 0.9)) == 0
    with pytest.raises(capi.SfmError):
        ctx.knn2(capi.NORM_L2, np.zeros((4, 300), np.float32), np.zeros((4, 300), np.float32))


@pytest.mark.parametrize("mode", ["f16", "bf16"])
@pytest.mark.parametrize("nt", [10277, 12288, 4100])
def test_l2_neighbours_planted_across_tiles(ctx, oracle, monkeypatch, nt, mode):
    """The narrow candidate passes keep the three best keys of a stretch of four 64-row tiles
    and join the running lists once per stretch.  Every query gets its three nearest train rows
    planted a chosen number of rows apart -- in one tile, in neighbouring tiles of a stretch, in
    different stretches, across the ranges of two workgroups, at the end of the set -- and the
    result must be the oracle's with (nearly) no query sent to the exhaustive pass."""
    from sfm_amd import capi

    monkeypatch.setenv("SFM_L2_MFMA", mode)
    rng = np.random.default_rng(nt)
    nq, dim = 600, 128
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.standard_normal((nt, dim)).astype(np.float32)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    gaps = [1, 3, 4, 8, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 256, 257, 448, 640, 1500]
    used = set()
    for i in range(nq):
        gap = gaps[i % len(gaps)]
        for _ in range(1000):
            base = int(rng.integers(0, nt - 2 * gap))
            if i in (0, 7, 14):        # three of them at the very end of the set
                base = nt - 1 - 2 * gap - {0: 0, 7: 5, 14: 11}[i]
            order = rng.permutation(3)     # which of the three rows is nearest
            rows = [base + int(o) * gap for o in order]
            if not used.intersection(rows):
                break
        used.update(rows)
        for rank, row in enumerate(rows):
            noise = rng.standard_normal(dim).astype(np.float32)
            t[row] = q[i] + np.float32(0.02 * (rank + 1)) * noise / np.linalg.norm(noise)
    idx, dist = ctx.knn2(capi.NORM_L2SQR, q, t)
    oidx, odist = oracle.knn2_l2(q, t, squared=True)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)
    assert ctx.l2_exhaustive_count() <= 3
    # the queries whose two nearest rows sit in one stretch of one lane's rows had that stretch
    # scored row by row (SFM_L2_COUNT_STRETCHES=1 counts them with the exhaustive ones)
    monkeypatch.setenv("SFM_L2_COUNT_STRETCHES", "1")
    idx, dist = ctx.knn2(capi.NORM_L2SQR, q, t)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)
    assert 10 <= ctx.l2_exhaustive_count() <= nq


@pytest.mark.parametrize("norm_name,ratio", [("L2", 0.7), ("L2SQR", 0.5)])
@pytest.mark.parametrize("integer", [True, False])
def test_l2_robust_matches_vs_oracle(ctx, oracle, norm_name, ratio, integer):
    from sfm_amd import capi, synth

    a, b = synth.float_descriptors(3000, 2600, dim=128, seed=99, integer=integer)
    norm = capi.NORM_L2 if norm_name == "L2" else capi.NORM_L2SQR
    onorm = oracle.NORM_L2 if norm_name == "L2" else oracle.NORM_L2SQR
    got = ctx.robust_matches(norm, a, b, ratio)
    want = oracle.robust_matches(onorm, a, b, ratio)
    assert len(want) > 200
    assert _tuples(got) == _tuples(want)
    f = ctx.ratio_matches(norm, a, b, ratio)
    oi, od = oracle.knn2_l2(a, b, squared=(norm_name == "L2SQR"))
    assert _tuples(f) == _tuples(oracle.ratio_filter(oi, od, len(b), ratio))


def test_matcher_mirror_class(ctx, oracle):
    """DescriptorMatcher keeps the reference's constructor mapping and outputs."""
    import sfm_amd
    from sfm_amd import capi, synth

    assert sfm_amd.DescriptorMatcher("bruteforce_hamming", 0.5, ctx=ctx).norm == capi.NORM_HAMMING
    assert sfm_amd.DescriptorMatcher("BruteForce_SL2", 0.5, ctx=ctx).norm == capi.NORM_L2SQR
    assert sfm_amd.DescriptorMatcher("anything else", 0.5, ctx=ctx).norm == capi.NORM_L2
    with pytest.raises(capi.SfmError):
        sfm_amd.DescriptorMatcher("FLANNBASED", 0.5, ctx=ctx)
    a, b = synth.float_descriptors(800, 900, dim=64, seed=5, integer=True)
    kp_a = np.random.default_rng(1).uniform(0, 640, (800, 2)).astype(np.float32)
    kp_b = np.random.default_rng(2).uniform(0, 640, (900, 2)).astype(np.float32)
    m = sfm_amd.DescriptorMatcher("BRUTEFORCE", 0.8, ctx=ctx)
    pa, pb, matches, ia, ib = m.findRobustMatches(kp_a, kp_b, a, b, useEpipolarFilter=False)
    want = oracle.robust_matches(oracle.NORM_L2, a, b, 0.8)
    assert _tuples(matches) == _tuples(want)
    np.testing.assert_array_equal(pa, kp_a[want["queryIdx"]])
    np.testing.assert_array_equal(pb, kp_b[want["trainIdx"]])
    np.testing.assert_array_equal(ia, want["queryIdx"])
    np.testing.assert_array_equal(ib, want["trainIdx"])
    # with the epipolar filter: the oracle's findFundamentalMat mask, applied in order
    pa, pb, matches, ia, ib = m.findRobustMatches(kp_a, kp_b, a, b, useEpipolarFilter=True)
    _, _, mask, _ = oracle.find_fundamental(kp_a[want["queryIdx"]], kp_b[want["trainIdx"]])
    assert _tuples(matches) == _tuples(want[mask.astype(bool)])


def _near_tie_sets(dim, nq, nt, seed, unit):
    """Train sets built to defeat candidate selection by an approximate key: for every query,
    24 train rows that equal its true second neighbour up to +-1 ulp in a few components, placed
    at indices that share one candidate list (multiples of 8), plus exact duplicates of it."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    t = rng.standard_normal((nt, dim)).astype(np.float32)
    if unit:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        t /= np.linalg.norm(t, axis=1, keepdims=True)
    else:
        q, t = np.round(q * 40), np.round(t * 40)     # integer-valued, SIFT-like magnitudes
    q, t = q.astype(np.float32), t.astype(np.float32)
    span = nt // nq                                   # query i's rows sit in [i*span, (i+1)*span):
    assert span >= 26 * 8                             # one train split, one lane half (index % 8 == 0)
    slots = np.stack([i * span + 8 * rng.permutation(span // 8)[:26] for i in range(nq)])
    for i in range(nq):
        near = q[i] + (0.05 if unit else 3.0) * rng.standard_normal(dim).astype(np.float32)
        if not unit:
            near = np.round(near)
        near = near.astype(np.float32)
        t[slots[i, 0]] = q[i] + (0.01 if unit else 1.0) * np.sign(rng.standard_normal(dim)).astype(np.float32)
        for k in range(1, 25):
            row = near.copy()
            j = rng.integers(0, dim, size=3)
            row[j] = np.nextafter(row[j], np.float32(np.inf) * np.sign(rng.standard_normal(3)).astype(np.float32))
            t[slots[i, k]] = row
        t[slots[i, 25]] = t[slots[i, 7]]              # an exact duplicate among the near ties
    return q, t


@pytest.mark.parametrize("dim", [64, 128])
@pytest.mark.parametrize("squared", [True, False])
@pytest.mark.parametrize("unit", [True, False])
def test_l2_near_ties_in_one_candidate_list(ctx, oracle, dim, squared, unit):
    """Two dozen train rows within an ulp of the second neighbour, all in one candidate list:
    more than the matrix-core pass keeps or re-scores.  Indices and distances must still be the
    oracle's (sequential sum, lower index first), through the exhaustive pass where needed."""
    from sfm_amd import capi

    q, t = _near_tie_sets(dim, 40, 16000, seed=dim + 2 * squared + unit, unit=unit)
    norm = capi.NORM_L2SQR if squared else capi.NORM_L2
    idx, dist = ctx.knn2(norm, q, t)
    n_exhaustive = ctx.l2_exhaustive_count()
    oidx, odist = oracle.knn2_l2(q, t, squared=squared)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)
    assert n_exhaustive >= 30          # the proof fails for (nearly) every query here


def test_l2_exhaustive_pass_alone(ctx, oracle, monkeypatch):
    """SFM_L2_EXHAUSTIVE=1 sends every query through the exhaustive pass; on ordinary data the
    proof succeeds for all but a few queries and the pass stays idle."""
    from sfm_amd import capi, synth

    q, t = synth.float_descriptors(700, 3001, dim=128, seed=17, integer=False)
    t[100] = t[7]
    oidx, odist = oracle.knn2_l2(q, t, squared=False)
    idx, dist = ctx.knn2(capi.NORM_L2, q, t)
    assert ctx.l2_exhaustive_count() <= 7
    np.testing.assert_array_equal(idx, oidx)
    monkeypatch.setenv("SFM_L2_EXHAUSTIVE", "1")
    idx, dist = ctx.knn2(capi.NORM_L2, q, t)
    assert ctx.l2_exhaustive_count() == 700
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)
    a, b = synth.float_descriptors(600, 500, dim=64, seed=3, integer=True)
    got = ctx.robust_matches(capi.NORM_L2SQR, a, b, 0.6)
    assert ctx.l2_exhaustive_count() == 1100
    want = oracle.robust_matches(oracle.NORM_L2SQR, a, b, 0.6)
    assert _tuples(got) == _tuples(want)


@pytest.mark.parametrize("scale", [1.0, 1e-20, 1e18])
def test_l2_zero_rows_and_extreme_magnitudes(ctx, oracle, scale):
    """All-zero train rows and an all-zero query give ranking keys of exactly zero, tiny data
    gives keys in the denormal range, huge data keys near the top of the float range: the
    candidate pass must keep telling rows apart (its keys carry a position tag in their low
    bits) and the result stays the oracle's, ties in index order."""
    from sfm_amd import capi, synth

    q, t = synth.float_descriptors(300, 2500, dim=128, seed=31, integer=False)
    q = (q * np.float32(scale)).astype(np.float32)
    t = (t * np.float32(scale)).astype(np.float32)
    t[[5, 64, 65, 700, 1999, 2400]] = 0.0
    q[[0, 17]] = 0.0
    q[40] = t[123]
    oidx, odist = oracle.knn2_l2(q, t, squared=True)
    idx, dist = ctx.knn2(capi.NORM_L2SQR, q, t)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)
    assert list(idx[0]) == [5, 64]            # the zero query: the first two zero rows


@pytest.mark.parametrize("dim", [32, 64, 128, 100])
@pytest.mark.parametrize("integer", [True, False])
def test_l2_every_matrix_core_pass(ctx, oracle, monkeypatch, dim, integer):
    """The candidate pass runs on the fp16 matrix cores by default (fp16 copies of the rows, the
    rounding measured per row and added to the proof's bound), on the bfloat16 ones under
    SFM_L2_MFMA=bf16 (every float split in a high and a low half) and on the fp32 ones under
    SFM_L2_MFMA=f32.  Whichever it is, the answer is the oracle's, and the wider error bounds of
    the narrow forms send only a few more queries to the exhaustive pass."""
    from sfm_amd import capi, synth

    q, t = synth.float_descriptors(900, 5003, dim=dim, seed=5 + dim, integer=integer)
    if integer:
        q *= 37.0                                       # beyond the integers fp16 holds exactly
        t *= 37.0
    t[4000] = t[11]
    oidx, odist = oracle.knn2_l2(q, t, squared=True)
    counts = {}
    for mode in ("f16", "bf16", "f32"):
        monkeypatch.setenv("SFM_L2_MFMA", mode)
        idx, dist = ctx.knn2(capi.NORM_L2SQR, q, t)
        counts[mode] = ctx.l2_exhaustive_count()
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_array_equal(dist, odist)
    assert counts["f32"] <= counts["bf16"] <= counts["f32"] + 45
    assert counts["f32"] <= counts["f16"] <= counts["f32"] + 90


@pytest.mark.parametrize("dim,rows,counts", [(128, 900, [900, 257, 640, 2, 513]), (64, 700, [700, 65, 300, 699, 128]),
                                             (200, 500, [500, 130, 1, 300, 499])])
@pytest.mark.parametrize("mode", ["f16", "bf16", "f32"])
def test_l2_match_pairs_ragged(ctx, oracle, monkeypatch, dim, rows, counts, mode):
    """Every ordered pair of five float-descriptor frames with very different row counts in one
    call (all frames packed once, each serving as query and as train set), against the
    oracle's cross-checked ratio matching, through each candidate pass."""
    import torch

    from sfm_amd import capi

    monkeypatch.setenv("SFM_L2_MFMA", mode)
    rng = np.random.default_rng(7 + dim)
    n_frames = len(counts)
    base = rng.standard_normal((rows, dim)).astype(np.float32)
    frames = []
    for f in range(n_frames):
        fr = base + 0.15 * rng.standard_normal((rows, dim)).astype(np.float32)
        fresh = rng.random(rows) < 0.3
        fr[fresh] = rng.standard_normal((int(fresh.sum()), dim)).astype(np.float32)
        frames.append(fr[rng.permutation(rows)] * np.float32(3.0 ** f))   # frames of different scale
    stack = np.stack(frames)
    counts = np.array(counts, np.int32)
    pairs = np.array([(i, j) for i in range(n_frames) for j in range(n_frames) if i != j], np.int32)
    dev = torch.device("cuda:0")
    d_counts, d_pairs = torch.from_numpy(counts).to(dev), torch.from_numpy(pairs).to(dev)
    cap = rows
    exhaustive = []
    for junk in (False, True):      # rows past a frame's count: zeros, or whatever a buffer may hold
        for f in range(n_frames):
            rest = stack[f, counts[f]:]
            rest[:] = 0.0
            if junk:
                rest[0::3] = 1e30
                rest[1::3] = np.nan
                rest[2::3] = np.inf
        d_desc = torch.from_numpy(stack).to(dev)
        d_matches = torch.zeros((len(pairs), cap, 4), dtype=torch.int32, device=dev)
        d_mc = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.match_pairs_dev(capi.NORM_L2, d_desc, n_frames, rows * dim * 4, rows, d_counts, dim, d_pairs,
                            len(pairs), 0.8, d_matches, cap, d_mc)
        ctx.synchronize()
        exhaustive.append(ctx.l2_exhaustive_count())
        mc = d_mc.cpu().numpy()
        m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(len(pairs), cap)
        total = 0
        for p, (i, j) in enumerate(pairs):
            want = oracle.robust_matches(oracle.NORM_L2, stack[i][: counts[i]], stack[j][: counts[j]], 0.8)
            assert mc[p] == len(want), (i, j)
            assert _tuples(m[p][: mc[p]]) == _tuples(want), (i, j)
            total += len(want)
        assert total > 100
    # the junk rows set neither the scale of the fp16 rows nor the error bound of the proof
    assert exhaustive[0] == exhaustive[1]


@pytest.mark.parametrize("mode", ["f16", "f32"])
def test_l2_match_pairs_tiny_frames(ctx, oracle, monkeypatch, mode):
    """Frames much smaller than the 32 rows a packing workgroup takes: one workgroup then holds
    rows of several frames (and frames of very different magnitude), and every frame's largest
    norm comes from rows that are not in a workgroup of its own."""
    import torch

    from sfm_amd import capi

    monkeypatch.setenv("SFM_L2_MFMA", mode)
    rng = np.random.default_rng(5)
    n_frames, rows, dim = 13, 7, 64
    counts = np.array([7, 3, 0, 5, 7, 1, 7, 6, 2, 7, 7, 4, 7], np.int32)
    base = rng.standard_normal((rows, dim)).astype(np.float32)
    stack = np.stack([(base[rng.permutation(rows)] + 0.1 * rng.standard_normal((rows, dim)).astype(np.float32))
                      * np.float32(2.0 ** (f % 5)) for f in range(n_frames)])
    for f in range(n_frames):
        stack[f, counts[f]:] = np.nan      # rows past a frame's count may hold anything
    pairs = np.array([(i, (i + k) % n_frames) for i in range(n_frames) for k in (1, 2, 5)], np.int32)
    dev = torch.device("cuda:0")
    d_desc = torch.from_numpy(stack).to(dev)
    d_counts, d_pairs = torch.from_numpy(counts).to(dev), torch.from_numpy(pairs).to(dev)
    d_matches = torch.zeros((len(pairs), rows, 4), dtype=torch.int32, device=dev)
    d_mc = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    ctx.match_pairs_dev(capi.NORM_L2, d_desc, n_frames, rows * dim * 4, rows, d_counts, dim, d_pairs,
                        len(pairs), 0.9, d_matches, rows, d_mc)
    ctx.synchronize()
    mc = d_mc.cpu().numpy()
    m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(len(pairs), rows)
    total = 0
    for p, (i, j) in enumerate(pairs):
        want = oracle.robust_matches(oracle.NORM_L2, stack[i][: counts[i]], stack[j][: counts[j]], 0.9)
        assert mc[p] == len(want), (i, j)
        assert _tuples(m[p][: mc[p]]) == _tuples(want), (i, j)
        total += len(want)
    assert total > 20


def test_l2_match_pairs_many_rows(ctx, oracle):
    """More packed rows (44 frames of 3001) than a workgroup of the narrowing launch scans for
    the largest norm by itself: the separate reduction sets the scale.  The frame with the largest
    rows is not in any pair."""
    import torch

    from sfm_amd import capi

    rng = np.random.default_rng(11)
    n_frames, rows, dim = 44, 3001, 64
    base = rng.standard_normal((rows, dim)).astype(np.float32)
    stack = np.empty((n_frames, rows, dim), np.float32)
    for f in range(n_frames):
        stack[f] = base[rng.permutation(rows)] + 0.2 * rng.standard_normal((rows, dim)).astype(np.float32)
    stack[37] *= np.float32(40.0)
    pairs = np.array([(0, 1), (43, 20)], np.int32)
    dev = torch.device("cuda:0")
    d_desc, d_pairs = torch.from_numpy(stack).to(dev), torch.from_numpy(pairs).to(dev)
    d_matches = torch.zeros((len(pairs), rows, 4), dtype=torch.int32, device=dev)
    d_mc = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    ctx.match_pairs_dev(capi.NORM_L2, d_desc, n_frames, rows * dim * 4, rows, None, dim, d_pairs,
                        len(pairs), 0.8, d_matches, rows, d_mc)
    ctx.synchronize()
    mc = d_mc.cpu().numpy()
    m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(len(pairs), rows)
    for p, (i, j) in enumerate(pairs):
        want = oracle.robust_matches(oracle.NORM_L2, stack[i], stack[j], 0.8)
        assert len(want) > 500 and mc[p] == len(want), (i, j)
        assert _tuples(m[p][: mc[p]]) == _tuples(want), (i, j)


@pytest.mark.parametrize("exhaustive", [False, True])
def test_l2_match_pairs_in_pieces(ctx, oracle, monkeypatch, exhaustive):
    """The pair list cut into pieces of three (SFM_L2_PAIR_CHUNK): the frames are packed once, the
    passes run once per piece; with SFM_L2_EXHAUSTIVE every query of every piece also goes through
    the exhaustive pass, whose counters are handed from piece to piece."""
    import torch

    from sfm_amd import capi

    monkeypatch.setenv("SFM_L2_PAIR_CHUNK", "3")
    if exhaustive:
        monkeypatch.setenv("SFM_L2_EXHAUSTIVE", "1")
    rng = np.random.default_rng(23)
    n_frames, rows, dim = 5, 300, 128
    counts = np.array([300, 257, 64, 299, 120], np.int32)
    base = rng.standard_normal((rows, dim)).astype(np.float32)
    stack = np.stack([base[rng.permutation(rows)] + 0.2 * rng.standard_normal((rows, dim)).astype(np.float32)
                      for _ in range(n_frames)])
    pairs = np.array([(i, j) for i in range(n_frames) for j in range(n_frames) if i != j][:11], np.int32)
    dev = torch.device("cuda:0")
    d_desc = torch.from_numpy(stack).to(dev)
    d_counts, d_pairs = torch.from_numpy(counts).to(dev), torch.from_numpy(pairs).to(dev)
    d_matches = torch.zeros((len(pairs), rows, 4), dtype=torch.int32, device=dev)
    d_mc = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    for _ in range(2):   # a second call finds the counters as the first one left them
        ctx.match_pairs_dev(capi.NORM_L2, d_desc, n_frames, rows * dim * 4, rows, d_counts, dim, d_pairs,
                            len(pairs), 0.8, d_matches, rows, d_mc)
        ctx.synchronize()
        mc = d_mc.cpu().numpy()
        m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(len(pairs), rows)
        for p, (i, j) in enumerate(pairs):
            want = oracle.robust_matches(oracle.NORM_L2, stack[i][: counts[i]], stack[j][: counts[j]], 0.8)
            assert mc[p] == len(want), (i, j)
            assert _tuples(m[p][: mc[p]]) == _tuples(want), (i, j)


@pytest.mark.parametrize("integer", [True, False])
def test_l2_full_size_20000x20000x128(ctx, oracle, integer):
    """The float-matching configuration at the size it is quoted on (20 000 x 20 000 x 128, both
    variants): 500 query rows spread over the set are checked bit for bit against the oracle run
    on those rows against the full train set; all rows are checked for the properties of a 2-NN
    result (valid distinct indices, ordered distances, the reported distances recomputed)."""
    import torch

    from sfm_amd import capi, synth

    n, dim = 20000, 128
    q, t = synth.float_descriptors(n, n, dim=dim, seed=99, integer=integer)
    dev = torch.device("cuda", 0)
    d_q, d_t = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    d_idx = torch.empty((n, 2), dtype=torch.int32, device=dev)
    d_dist = torch.empty((n, 2), dtype=torch.float32, device=dev)
    ctx.knn2_dev(capi.NORM_L2, d_q, n, dim * 4, d_t, n, dim * 4, dim, d_idx, d_dist)
    ctx.synchronize()
    idx, dist = d_idx.cpu().numpy(), d_dist.cpu().numpy()
    assert ctx.l2_exhaustive_count() < n // 100
    assert (idx >= 0).all() and (idx < n).all() and (idx[:, 0] != idx[:, 1]).all()
    assert (dist[:, 0] <= dist[:, 1]).all() and np.isfinite(dist).all()
    ties = dist[:, 0] == dist[:, 1]
    assert (idx[ties, 0] < idx[ties, 1]).all()
    # the reported distances are the distances to the reported rows (float64 check of all rows)
    for k in range(2):
        d64 = np.sqrt(((q.astype(np.float64) - t[idx[:, k]].astype(np.float64)) ** 2).sum(1))
        np.testing.assert_allclose(dist[:, k], d64, rtol=2e-6, atol=1e-6)
    rows = np.linspace(0, n - 1, 500).astype(np.int64)
    oidx, odist = oracle.knn2_l2(q[rows], t, squared=False)
    np.testing.assert_array_equal(idx[rows], oidx)
    np.testing.assert_array_equal(dist[rows], odist)
