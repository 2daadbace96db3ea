"""GPU parity tests of Shi-Tomasi corner detection (cv::goodFeaturesToTrack as called at
Source Files/feature_processing.cpp:93) against the CPU oracle: the corner-strength map is
compared bit for bit, the corner lists element for element.

Parity is unpinned for this stage as for the others (no OpenCV here): the oracle restates the
published algorithm and fixes the one order OpenCV leaves to its build (the box filter's
running sums), see oracle/corners_oracle.c.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from sfm_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def frame(h, w, seed):
    from sfm_amd import synth

    return synth.to_u8(synth.texture(h, w, seed=seed))


@pytest.mark.parametrize("h,w,seed", [(720, 1280, 42), (3, 3, 1), (16, 64, 2), (17, 65, 3), (5, 200, 4),
                                      (131, 7, 5), (480, 640, 6)])
def test_min_eigen_val_bit_exact(ctx, oracle, h, w, seed):
    import torch

    rng = np.random.default_rng(seed)
    img = frame(h, w, seed) if min(h, w) >= 16 else rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    d_img = torch.from_numpy(img).cuda()
    d_eig = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.min_eigen_val_dev(d_img, w, h, w, d_eig)
    ctx.synchronize()
    assert np.array_equal(d_eig.cpu().numpy(), oracle.min_eigen_val(img))


@pytest.mark.parametrize("name", ["a", "a_all", "a_frac", "a_nodist", "b", "c", "d"])
def test_good_features_golden_cases(ctx, name):
    """The committed NumPy vectors (ties, fractional distance, no distance filter, constant
    image), straight against the GPU."""
    g = np.load(os.path.join(GOLDEN, "corners.npz"))
    mc, q, md = g[f"args_{name}"]
    pts = ctx.good_features(g[f"img_{name}"], int(mc), q, md)
    assert np.array_equal(pts, g[f"pts_{name}"])


@pytest.mark.parametrize("h,w,mc,q,md", [(720, 1280, 2000, 0.01, 10.0), (720, 1280, 0, 0.01, 10.0),
                                         (720, 1280, 5000, 0.001, 3.0), (720, 1280, 300, 0.05, 30.0),
                                         (480, 640, 2000, 0.01, 0.0), (480, 640, 0, 0.02, 7.5),
                                         (97, 211, 100, 0.01, 1.0)])
def test_good_features_matches_oracle(ctx, oracle, h, w, mc, q, md):
    img = frame(h, w, 11 + h)
    pts = ctx.good_features(img, mc, q, md)
    want = oracle.good_features(img, mc, q, md)
    assert len(want) > 10
    assert np.array_equal(pts, want)
    if md >= 1:  # the property the filter promises (cells of round(md): exact for integers)
        d = pts[:, None, :] - pts[None, :, :]
        d2 = (d * d).sum(-1) + np.eye(len(pts)) * 1e9
        if float(md).is_integer() and len(pts) < 4000:
            assert d2.min() >= md * md


def test_strided_rows_and_roi(ctx, oracle):
    """A cv::Mat region of interest: rows further apart than the width."""
    big = frame(300, 400, 5)
    roi = big[20:220, 31:333]
    assert not roi.flags["C_CONTIGUOUS"]
    pts = ctx.good_features(roi, 500, 0.01, 8.0)
    assert np.array_equal(pts, oracle.good_features(np.ascontiguousarray(roi), 500, 0.01, 8.0))


def test_plateaus_and_candidate_overflow(ctx, oracle):
    """Blocks of constant colour give plateaus of exactly equal strength: every pixel of a
    plateau is a candidate.  The device entry reports the overflow of a small candidate
    capacity; the host entry repeats with room for every pixel."""
    import torch

    rng = np.random.default_rng(3)
    img = np.kron(rng.integers(0, 2, size=(24, 32)).astype(np.uint8) * 200, np.ones((6, 6), np.uint8))
    h, w = img.shape
    want, ncand = oracle.good_features(img, 0, 0.01, 4.0, return_candidates=True)
    pts = ctx.good_features(img, 0, 0.01, 4.0)
    assert len(want) > 50 and np.array_equal(pts, want)
    d_img = torch.from_numpy(img).cuda()
    d_pts = torch.zeros((w * h, 2), dtype=torch.float32, device="cuda")
    d_info = torch.zeros(3, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.good_features_dev(d_img, w, h, w, 0, 0.01, 4.0, d_pts, w * h, d_info, max_candidates=ncand // 2)
    ctx.synchronize()
    info = d_info.cpu().numpy()
    assert info[1] == ncand and info[2] == 1
    ctx.good_features_dev(d_img, w, h, w, 0, 0.01, 4.0, d_pts, w * h, d_info, max_candidates=ncand)
    ctx.synchronize()
    info = d_info.cpu().numpy()
    assert info[2] == 0 and info[0] == len(want)
    assert np.array_equal(d_pts.cpu().numpy()[: info[0]], want)


def test_thousands_of_equal_strengths(ctx, oracle):
    """A large image of constant blocks: a handful of distinct strengths, thousands of
    candidates each.  Candidates of one strength fall into one value bucket, far more than the
    in-LDS sort of a bucket takes, so the radix path orders them -- by descending address, as
    OpenCV's pointer comparison does."""
    rng = np.random.default_rng(17)
    img = np.kron(rng.integers(0, 2, size=(120, 160)).astype(np.uint8) * 180 + 20, np.ones((4, 4), np.uint8))
    want, ncand = oracle.good_features(img, 3000, 0.01, 3.0, return_candidates=True)
    assert ncand > 15000 and len(want) > 500
    pts = ctx.good_features(img, 3000, 0.01, 3.0)
    assert np.array_equal(pts, want)


def test_dependency_chain_longer_than_the_rounds(ctx, oracle):
    """A ramp of strictly decreasing strengths along a line, spaced closer than minDistance:
    candidate k depends on k-1, so the parallel rounds cannot finish and the sequential pass
    has to; the answer is still the sequential one."""
    h, w = 40, 900
    img = np.full((h, w), 20, np.uint8)
    for k in range(140):  # isolated bright dots of decreasing contrast, 6 px apart
        img[20, 10 + 6 * k] = 250 - k
    want = oracle.good_features(img, 0, 0.0001, 8.0)
    pts = ctx.good_features(img, 0, 0.0001, 8.0)
    assert len(want) > 40 and np.array_equal(pts, want)


def test_generate_flow_features_appends(ctx, oracle):
    """FeatureDetector::generateFlowFeatures keeps what the vector holds and appends."""
    from sfm_amd import FeatureDetector

    img = frame(240, 320, 9)
    det = FeatureDetector(ctx=ctx)
    have = np.array([[1.5, 2.5], [3.0, 4.0]], np.float32)
    out = det.generateFlowFeatures(img, have, 200, 0.01, 10.0)
    want = oracle.good_features(img, 200, 0.01, 10.0)
    assert np.array_equal(out[:2], have) and np.array_equal(out[2:], want)


def test_bad_arguments_raise(ctx):
    from sfm_amd import capi

    img = frame(32, 32, 1)
    with pytest.raises(capi.SfmError):
        ctx.good_features(img, 10, 0.0, 5.0)      # OpenCV asserts qualityLevel > 0
    with pytest.raises(capi.SfmError):
        ctx.good_features(img, 10, 0.01, -1.0)    # ... and minDistance >= 0
    with pytest.raises(capi.SfmError):
        ctx.good_features(img[:2], 10, 0.01, 1.0)  # smaller than the 3x3 neighbourhood


def test_batch_of_images_equals_single_calls(ctx, oracle):
    """Many frames per call: one set of launches, one sort with the image number in the key."""
    import torch

    h, w, n_img, cap = 240, 320, 5, 400
    imgs = np.stack([frame(h, w, 70 + i) for i in range(n_img)])
    imgs[3] = 50                                  # a constant frame in the middle: no corners
    d_imgs = torch.from_numpy(imgs).cuda()
    d_pts = torch.zeros((n_img, cap, 2), dtype=torch.float32, device="cuda")
    d_info = torch.zeros((n_img, 3), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.good_features_batch_dev(d_imgs, n_img, w * h, w, h, w, cap, 0.01, 7.0, d_pts, cap, d_info)
    ctx.synchronize()
    info, pts = d_info.cpu().numpy(), d_pts.cpu().numpy()
    for i in range(n_img):
        want, ncand = oracle.good_features(imgs[i], cap, 0.01, 7.0, return_candidates=True)
        assert info[i, 0] == len(want) and info[i, 1] == ncand and info[i, 2] == 0
        assert np.array_equal(pts[i, : len(want)], want)
    assert info[3, 0] == 0 and info[0, 0] == cap
