"""GPU parity tests of the epipolar filter (cv::findFundamentalMat as called at
Source Files/feature_processing.cpp:188) against the CPU oracle.

Parity is unpinned for this stage as for the others: no OpenCV is available, the oracle
restates the published algorithm.  GPU and oracle draw the same subsets (integer work,
compared exactly) and run the same elimination with the same operation order; only the cubic
solver's acos / cos / pow and the log / pow of the iteration update come from different
math libraries, so candidate matrices are compared to 1e-9 and masks are expected -- and
asserted -- to be identical on these seeded inputs.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from sfm_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def two_view(n, seed, outliers=0.2, noise=0.5):
    """Projections of random 3D points into two cameras, with pixel noise and gross outliers;
    returns (p1, p2, is_inlier)."""
    from sfm_amd import synth

    s = synth.triangulation_set(n, seed=seed)
    rng = np.random.default_rng(seed + 1)
    p1, p2 = s["p1"].copy(), s["p2"].copy()
    bad = rng.random(n) < outliers
    p2[bad] = np.stack([rng.uniform(0, 1280, bad.sum()), rng.uniform(0, 720, bad.sum())], 1)
    if noise != 0.5:
        p2 += rng.normal(0, noise, p2.shape).astype(np.float32)
    return p1, p2.astype(np.float32), ~bad


@pytest.mark.parametrize("n,seed,outliers", [(5734, 1, 0.2), (300, 2, 0.4), (15, 3, 0.0), (2000, 4, 0.6),
                                             (64, 5, 0.1)])
def test_ransac_mask_matches_oracle(ctx, oracle, n, seed, outliers):
    p1, p2, truth = two_view(n, seed, outliers)
    found, F, mask = ctx.fundamental(p1, p2)
    ofound, oF, omask, iters = oracle.find_fundamental(p1, p2)
    assert found == ofound and found
    np.testing.assert_array_equal(mask, omask)
    np.testing.assert_allclose(F, oF, rtol=1e-9, atol=1e-12)
    # the filter does its job: it keeps mostly true correspondences
    if outliers < 0.5:
        kept = mask.astype(bool)
        assert kept.sum() > 0.5 * truth.sum()
        assert truth[kept].mean() > 0.9


@pytest.mark.parametrize("n", [0, 3, 6, 7, 8, 11, 14])
def test_small_point_sets(ctx, oracle, n):
    """Fewer than 7 points: nothing found, mask all zero.  Exactly 7: direct solve, mask all
    one.  8 to 14: least median of squares."""
    p1, p2, _ = two_view(40, 11, outliers=0.0)
    p1, p2 = p1[:n], p2[:n]
    found, F, mask = ctx.fundamental(p1, p2)
    omask0 = np.zeros(max(n, 1), np.uint8)[:n]
    ofound, oF, omask, _ = oracle.find_fundamental(p1, p2) if n else (False, None, omask0, 0)
    assert found == ofound
    if n < 7:
        assert not found and (mask == 0).all()
    else:
        np.testing.assert_array_equal(mask, omask)
        np.testing.assert_allclose(F, oF, rtol=1e-9, atol=1e-12)
    if n == 7:
        assert (mask == 1).all()


def test_seven_point_property(ctx):
    """Every matrix of a direct 7-point solve satisfies x2' F x1 = 0 on its points and is
    singular (needs no oracle)."""
    p1, p2, _ = two_view(7, 21, outliers=0.0)
    found, F, mask = ctx.fundamental(p1, p2)
    assert found
    h1 = np.c_[p1.astype(np.float64), np.ones(7)]
    h2 = np.c_[p2.astype(np.float64), np.ones(7)]
    res = np.abs(np.einsum("ni,ij,nj->n", h2, F, h1))
    assert res.max() < 1e-8 * np.abs(F).max() * 1e6
    assert abs(np.linalg.det(F / np.abs(F).max())) < 1e-10


def test_batched_problems_with_device_counts(ctx, oracle):
    import torch

    sets = [two_view(900, 31, 0.3), two_view(700, 32, 0.1), two_view(12, 33, 0.0), two_view(5, 34, 0.0),
            two_view(7, 35, 0.0), two_view(899, 36, 0.5)]
    stride = 900
    P = len(sets)
    pts1 = np.zeros((P, stride, 2), np.float32)
    pts2 = np.zeros((P, stride, 2), np.float32)
    counts = np.array([len(s[0]) for s in sets], np.int32)
    for k, (a, b, _) in enumerate(sets):
        pts1[k, : len(a)] = a
        pts2[k, : len(b)] = b
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(pts1).to(dev), torch.from_numpy(pts2).to(dev)
    dc = torch.from_numpy(counts).to(dev)
    dF = torch.zeros((P, 9), dtype=torch.float64, device=dev)
    dm = torch.full((P, stride), 9, dtype=torch.uint8, device=dev)
    df = torch.zeros(P, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.fundamental_dev(d1, d2, stride, dc, 0, P, 3.0, 0.99, 1000, dF, dm, df)
    ctx.synchronize()
    masks, founds = dm.cpu().numpy(), df.cpu().numpy()
    for k, (a, b, _) in enumerate(sets):
        n = len(a)
        if n < 7:
            assert founds[k] == 0 and (masks[k, :n] == 0).all()
        else:
            ofound, oF, omask, _ = oracle.find_fundamental(a, b)
            assert bool(founds[k]) == ofound
            np.testing.assert_array_equal(masks[k, :n], omask)
        assert (masks[k, n:] == 9).all()  # nothing written past a problem's count


def test_epipolar_filter_pairs_in_place(ctx, oracle):
    """sfm_match_pairs_dev followed by sfm_epipolar_filter_pairs_dev on the device, against
    the reference's order of operations done with the oracle: robust matches, key point
    gather, findFundamentalMat, keep the inliers in order
    (Source Files/feature_processing.cpp:148-204, useEpipolarFilter == true)."""
    import torch

    from sfm_amd import capi, synth

    n, n_frames = 1500, 3
    s = synth.triangulation_set(n, seed=41)
    rng = np.random.default_rng(42)
    base = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    descs, kps = [], []
    views = [s["p1"], s["p2"], s["p2"] + np.float32(3.0)]
    for f in range(n_frames):
        perm = rng.permutation(n)
        d = base[perm].copy()
        d ^= np.packbits(rng.random(size=(n, 256)) < 0.03, axis=1, bitorder="little")
        fresh = rng.random(n) < 0.3
        d[fresh] = rng.integers(0, 256, size=(int(fresh.sum()), 32), dtype=np.uint8)
        kp = np.zeros((n, 7), np.float32)       # 28-byte cv::KeyPoint-like records
        kp[:, :2] = views[f][perm]
        kp[fresh, :2] = np.stack([rng.uniform(0, 1280, fresh.sum()), rng.uniform(0, 720, fresh.sum())], 1)
        descs.append(d)
        kps.append(kp)
    counts = np.array([n, n - 200, 5], np.int32)     # the last frame yields (almost) no matches
    pairs = np.array([[0, 1], [1, 0], [0, 2], [1, 2]], np.int32)
    dev = torch.device("cuda:0")
    d_desc = torch.from_numpy(np.stack(descs)).to(dev)
    d_kp = torch.from_numpy(np.stack(kps)).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_pairs = torch.from_numpy(pairs).to(dev)
    P, cap = len(pairs), n
    d_matches = torch.zeros((P, cap, 4), dtype=torch.int32, device=dev)
    d_mc = torch.zeros(P, dtype=torch.int32, device=dev)
    d_mask = torch.zeros((P, cap), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, n_frames, n * 32, n, d_counts, 32, d_pairs, P, 0.8,
                        d_matches, cap, d_mc)
    ctx.epipolar_filter_pairs_dev(d_kp, n * 28, 28, d_pairs, P, d_matches, cap, d_mc, d_mask=d_mask)
    ctx.synchronize()
    mc = d_mc.cpu().numpy()
    m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(P, cap)
    kept_total = 0
    for p, (a, b) in enumerate(pairs):
        want = oracle.robust_matches(oracle.NORM_HAMMING, descs[a][: counts[a]], descs[b][: counts[b]], 0.8)
        p1 = kps[a][want["queryIdx"], :2].copy()
        p2 = kps[b][want["trainIdx"], :2].copy()
        if len(want) >= 7:
            _, _, omask, _ = oracle.find_fundamental(p1, p2)
        else:
            omask = np.zeros(len(want), np.uint8)   # no matrix: the zero mask drops every match
        want = want[omask.astype(bool)]
        assert mc[p] == len(want), (a, b)
        assert m[p][: mc[p]].tobytes() == want.tobytes(), (a, b)
        kept_total += len(want)
    assert kept_total > 500 and mc[2] == 0


def test_python_wrapper_epipolar_filter(ctx, oracle):
    """DescriptorMatcher.findRobustMatches(useEpipolarFilter=True) end to end."""
    from sfm_amd import DescriptorMatcher, synth

    n = 1200
    s = synth.triangulation_set(n, seed=51)
    rng = np.random.default_rng(52)
    d0 = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    perm = rng.permutation(n)
    d1 = d0[perm].copy()
    d1 ^= np.packbits(rng.random(size=(n, 256)) < 0.04, axis=1, bitorder="little")
    kp0, kp1 = s["p1"], s["p2"][perm].copy()
    wrong = rng.random(n) < 0.15
    kp1[wrong] = np.stack([rng.uniform(0, 1280, wrong.sum()), rng.uniform(0, 720, wrong.sum())], 1)
    m = DescriptorMatcher("BruteForce_Hamming", 0.8, ctx=ctx)
    pp, cp, matches, pi, ci = m.findRobustMatches(kp0, kp1, d0, d1, useEpipolarFilter=True)
    want = oracle.robust_matches(oracle.NORM_HAMMING, d0, d1, 0.8)
    _, _, omask, _ = oracle.find_fundamental(kp0[want["queryIdx"]], kp1[want["trainIdx"]])
    want = want[omask.astype(bool)]
    assert len(want) > 300 and matches.tobytes() == want.tobytes()
    assert np.array_equal(pi, want["queryIdx"]) and np.array_equal(ci, want["trainIdx"])
    assert np.array_equal(pp, kp0[want["queryIdx"]]) and np.array_equal(cp, kp1[want["trainIdx"]])
