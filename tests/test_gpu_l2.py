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
