"""The C oracle (oracle/*.c) against the independently generated NumPy vectors
in tests/golden/ (made by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden


def _as_tuples(m):
    return [(int(a["queryIdx"]), int(a["trainIdx"]), float(a["distance"])) for a in m]


@pytest.mark.parametrize("d", [32, 61, 64])
def test_hamming_knn2_golden(oracle, d):
    g = load_golden("hamming.npz")
    for threads in (1, 0):
        idx, dist = oracle.knn2_hamming(g[f"q{d}"], g[f"t{d}"], threads=threads)
        np.testing.assert_array_equal(idx, g[f"idx{d}"])
        np.testing.assert_array_equal(dist, g[f"dist{d}"])


@pytest.mark.parametrize("nt", [1, 2, 3])
def test_hamming_tiny_train(oracle, nt):
    g = load_golden("hamming.npz")
    idx, dist = oracle.knn2_hamming(g[f"tiny_q_nt{nt}"], g[f"tiny_t_nt{nt}"])
    np.testing.assert_array_equal(idx, g[f"tiny_idx_nt{nt}"])
    np.testing.assert_array_equal(dist, g[f"tiny_dist_nt{nt}"])


@pytest.mark.parametrize("d", [32, 61, 64])
@pytest.mark.parametrize("tag,ratio", [("r50", 0.5), ("r80", 0.8)])
def test_ratio_and_cross_golden(oracle, d, tag, ratio):
    g = load_golden("hamming.npz")
    q, t = g[f"q{d}"], g[f"t{d}"]
    idx, dist = oracle.knn2_hamming(q, t)
    f = oracle.ratio_filter(idx, dist, len(t), ratio)
    idxb, distb = oracle.knn2_hamming(t, q)
    b = oracle.ratio_filter(idxb, distb, len(q), ratio)
    assert _as_tuples(f) == [tuple(r) for r in g[f"fwd{d}_{tag}"].tolist()]
    assert _as_tuples(b) == [tuple(r) for r in g[f"bwd{d}_{tag}"].tolist()]
    x = oracle.cross_check(f, b)
    want = [tuple(r) for r in g[f"cross{d}_{tag}"].tolist()]
    assert _as_tuples(x) == want
    # the O(n) restatement used by the timed baseline gives the same list
    x2 = oracle.robust_matches(oracle.NORM_HAMMING, q, t, ratio)
    assert _as_tuples(x2) == want
    if tag == "r80":
        assert len(want) > 10  # the fixture is not vacuous


def test_ratio_needs_two_train_rows(oracle):
    g = load_golden("hamming.npz")
    idx, dist = oracle.knn2_hamming(g["tiny_q_nt1"], g["tiny_t_nt1"])
    assert len(oracle.ratio_filter(idx, dist, 1, 0.9)) == 0
    assert len(oracle.robust_matches(oracle.NORM_HAMMING, g["tiny_q_nt1"], g["tiny_t_nt1"], 0.9)) == 0


def test_l2_integer_valued_golden(oracle):
    g = load_golden("l2.npz")
    idx, dist = oracle.knn2_l2(g["int_q"], g["int_t"], squared=True)
    np.testing.assert_array_equal(idx, g["int_idx"])
    np.testing.assert_array_equal(dist, g["int_sqr"])
    idx, dist = oracle.knn2_l2(g["int_q"], g["int_t"], squared=False)
    np.testing.assert_array_equal(idx, g["int_idx_l2"])
    np.testing.assert_array_equal(dist, g["int_l2"])


def test_l2_real_valued_golden(oracle):
    g = load_golden("l2.npz")
    idx, dist = oracle.knn2_l2(g["real_q"], g["real_t"], squared=True)
    d64 = g["real_sqr64"]
    # float32 accumulation error for unit-norm rows is ~1e-6; require index
    # agreement where the float64 gaps are clear of that
    clear = (d64[:, 1] - d64[:, 0] > 1e-5) & (d64[:, 2] - d64[:, 1] > 1e-5)
    assert clear.sum() > 50
    np.testing.assert_array_equal(idx[clear], g["real_idx"][clear])
    np.testing.assert_allclose(dist, d64[:, :2], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_pyr_down_golden(oracle, name):
    g = load_golden("lk.npz")
    np.testing.assert_array_equal(oracle.pyr_down(g[f"pyr_src_{name}"]), g[f"pyr_dst_{name}"])


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_scharr_golden(oracle, name):
    g = load_golden("lk.npz")
    np.testing.assert_array_equal(oracle.scharr_deriv(g[f"pyr_src_{name}"]), g[f"scharr_{name}"])


@pytest.mark.parametrize("exact", [0, 1])
def test_pyrlk_golden(oracle, exact):
    g = load_golden("lk.npz")
    assert oracle.lk_num_levels(64, 48, 9, 9, 3) == int(g["lk_levels"][0]) == 2
    for threads in (1, 0):
        nxt, st, err = oracle.pyrlk(g["lk_prev"], g["lk_curr"], g["lk_pts"], win=(9, 9),
                                    max_level=3, crit_type=3, max_count=30, epsilon=0.01,
                                    acc_exact=bool(exact), threads=threads)
        np.testing.assert_array_equal(st, g[f"lk_status_{exact}"])
        np.testing.assert_array_equal(nxt, g[f"lk_next_{exact}"])
        np.testing.assert_array_equal(err, g[f"lk_err_{exact}"])
    # known answer: tracked points moved by the image shift
    ok = st.astype(bool)
    assert ok.sum() >= 20
    d = (nxt - g["lk_pts"])[ok]
    assert np.abs(np.median(d, axis=0) - g["lk_shift"]).max() < 0.1


def test_pyrlk_second_config_golden(oracle):
    g = load_golden("lk.npz")
    nxt, st, err = oracle.pyrlk(g["lk_prev"], g["lk_curr"], g["lk_pts"], win=(7, 5), max_level=1,
                                crit_type=3, max_count=10, epsilon=0.0, acc_exact=True)
    np.testing.assert_array_equal(st, g["lk2_status"])
    np.testing.assert_array_equal(nxt, g["lk2_next"])
    np.testing.assert_array_equal(err, g["lk2_err"])


def test_pyrlk_identical_images(oracle):
    g = load_golden("lk.npz")
    pts = g["lk_pts"][:30]
    nxt, st, err = oracle.pyrlk(g["lk_prev"], g["lk_prev"], pts, win=(9, 9), acc_exact=True)
    assert st.all()
    # (x - halfWin) + halfWin may round by an ulp; the flow itself is exactly zero
    np.testing.assert_allclose(nxt, pts, rtol=0, atol=4e-6)
    np.testing.assert_array_equal(err, np.zeros(len(pts), np.float32))


def test_float_and_exact_accumulators_agree(oracle):
    g = load_golden("lk.npz")
    a = g["lk_next_0"]
    b = g["lk_next_1"]
    ok = g["lk_status_0"].astype(bool) & g["lk_status_1"].astype(bool)
    assert np.abs(a[ok] - b[ok]).max() < 1e-3


def test_svd4_golden(oracle):
    g = load_golden("svd4.npz")
    for A, w, v in zip(g["A"], g["w"], g["v"]):
        ww, vv = oracle.svd4_null(A)
        np.testing.assert_allclose(ww, w, rtol=1e-12)
        s = np.sign(np.dot(vv, v))
        np.testing.assert_allclose(vv * s, v, atol=1e-9)


def test_triangulate_golden(oracle):
    g = load_golden("triangulate.npz")
    img = np.random.default_rng(int(g["img_seed"][0]))
    # the generator drew the image after the point data; regenerate the same stream
    X, rgb, mask, err = oracle.triangulate(g["K"], g["prev_pose"], g["curr_pose"], g["R"], g["t"],
                                           g["p1"], g["p2"])
    sv = g["sv"]
    well = sv[:, 3] / sv[:, 2] < 0.5  # smallest singular value separated from the next
    assert well.sum() > 380
    rel = np.linalg.norm(X - g["X"], axis=1) / np.linalg.norm(g["X"], axis=1)
    assert rel[well].max() < 1e-8
    np.testing.assert_allclose(err[well], g["err"][well], rtol=1e-6, atol=1e-7)
    # mask agrees wherever no threshold is within a hair of the value
    band = (np.abs(g["err"] - 3.0) < 1e-6) | (np.abs(g["depth"] - 250.0) < 1e-6)
    np.testing.assert_array_equal(mask[well & ~band], g["mask"][well & ~band])
    assert 0 < mask.sum() < len(mask)
    # known answer: noise-free projections of the generating points come back
    del img


def test_triangulate_exact_projections(oracle):
    g = load_golden("triangulate.npz")
    K, Rm, t = g["K"], g["R"], g["t"]
    Xw = g["truth"][:100]
    P1 = K @ g["prev_pose"]
    P2 = K @ g["curr_pose"]
    h = np.hstack([Xw, np.ones((len(Xw), 1))])
    p1 = (P1 @ h.T).T
    p2 = (P2 @ h.T).T
    p1 = (p1[:, :2] / p1[:, 2:]).astype(np.float32)
    p2 = (p2[:, :2] / p2[:, 2:]).astype(np.float32)
    X, _, mask, err = oracle.triangulate(K, g["prev_pose"], g["curr_pose"], Rm, t, p1, p2)
    # float32 pixel coordinates limit the accuracy (~1e-4 px at 1000 px)
    rel = np.linalg.norm(X - Xw, axis=1) / np.linalg.norm(Xw, axis=1)
    assert rel.max() < 5e-3
    assert mask.all()
    assert err.max() < 1e-2


def test_triangulate_colour_lookup(oracle):
    g = load_golden("triangulate.npz")
    rng = np.random.default_rng(123)
    img = rng.integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)
    _, rgb, _, _ = oracle.triangulate(g["K"], g["prev_pose"], g["curr_pose"], g["R"], g["t"],
                                      g["p1"], g["p2"], bgr=img)
    px = np.clip(np.rint(g["p2"][:, 0].astype(np.float64)).astype(int), 0, 1279)
    py = np.clip(np.rint(g["p2"][:, 1].astype(np.float64)).astype(int), 0, 719)
    np.testing.assert_array_equal(rgb, img[py, px])


def test_oracle_pieces_known_answers(oracle):
    """The oracle itself (no GPU work): generator and iteration-count formula against values
    computed by hand from their definitions, the 7-point solver against its defining
    property."""
    # cv::RNG: state = (uint32)state * 4164903690 + (state >> 32), seeded with 2^64 - 1
    state = (1 << 64) - 1
    want = []
    for _ in range(21):
        state = (state & 0xFFFFFFFF) * 4164903690 + (state >> 32)
        want.append((state & 0xFFFFFFFF) % 1000)
    p = np.stack([np.arange(1000) * 1.37 % 640, np.arange(1000) * 7.91 % 480], 1).astype(np.float32)
    subs = oracle.fm_subsets(p, p[::-1].copy(), 3)
    flat = [w for w in want]
    # no repeated index among these draws, so the first subsets are the raw draws
    assert len(set(flat[:7])) == 7
    assert subs[0].tolist() == flat[:7]
    # RANSACUpdateNumIters(0.99, 0.3, 7, 1000) = round(log(0.01) / log(1 - 0.7^7))
    assert oracle.ransac_update_num_iters(0.99, 0.3, 7, 1000) == round(np.log(0.01) / np.log(1 - 0.7 ** 7))
    assert oracle.ransac_update_num_iters(0.99, 0.9, 7, 1000) == 1000


def test_oracle_fundamental_recovers_geometry(oracle):
    """End to end on synthetic two-view data with known outliers: the mask keeps true
    correspondences and rejects gross outliers; the 7-point matrices satisfy their defining
    equations."""
    from sfm_amd import synth

    s = synth.triangulation_set(2000, seed=8)
    rng = np.random.default_rng(9)
    p1, p2 = s["p1"].copy(), s["p2"].copy()
    bad = rng.random(2000) < 0.3
    p2[bad] = np.stack([rng.uniform(0, 1280, bad.sum()), rng.uniform(0, 720, bad.sum())], 1)
    found, F, mask, iters = oracle.find_fundamental(p1, p2)
    assert found and 1 <= iters <= 1000
    kept = mask.astype(bool)
    assert (~bad)[kept].mean() > 0.95 and kept.sum() > 0.5 * (~bad).sum()
    for Fk in oracle.fm_7point(p1[:7], p2[:7]):
        h1 = np.c_[p1[:7].astype(np.float64), np.ones(7)]
        h2 = np.c_[p2[:7].astype(np.float64), np.ones(7)]
        assert np.abs(np.einsum("ni,ij,nj->n", h2, Fk, h1)).max() < 1e-9 * max(1.0, np.abs(Fk).max()) * 1e3
        assert abs(np.linalg.det(Fk / np.abs(Fk).max())) < 1e-10
    # fewer than 7 points: nothing found, mask untouched
    found, _, mask, _ = oracle.find_fundamental(p1[:6], p2[:6])
    assert not found and (mask == 0).all()


def test_min_eigen_val_golden(oracle):
    """cornerMinEigenVal restated in C against the independent whole-array NumPy version."""
    g = load_golden("corners.npz")
    for name in ("a", "b"):
        assert np.array_equal(oracle.min_eigen_val(g[f"img_{name}"]), g[f"eig_{name}"])


@pytest.mark.parametrize("name", ["a", "a_all", "a_frac", "a_nodist", "b", "c", "d"])
def test_good_features_golden(oracle, name):
    """goodFeaturesToTrack: threshold, local maxima, order (ties by address), grid-limited
    minimum distance incl. a fractional one, maxCorners, no distance filter, constant image."""
    g = load_golden("corners.npz")
    mc, q, md = g[f"args_{name}"]
    pts, ncand = oracle.good_features(g[f"img_{name}"], int(mc), q, md, return_candidates=True)
    assert ncand == int(g[f"ncand_{name}"][0])
    assert np.array_equal(pts, g[f"pts_{name}"])


def _exact_two_view(n, seed):
    """Noise-free projections (float64) of random points into two known cameras."""
    from sfm_amd import synth

    s = synth.triangulation_set(n, seed=seed, sigma=0.0, outliers=0.0)
    t = s["t"]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ s["R"]
    return s, E / np.linalg.norm(E)


def test_oracle_five_point_known_answers(oracle):
    """The five-point restatement against its defining properties and a known geometry:
    every solution satisfies the five epipolar equations, is singular with two equal
    singular values, and one of them is the matrix the points were generated with."""
    s, E_true = _exact_two_view(40, 3)
    Kinv = np.linalg.inv(s["K"])
    for start in (0, 5, 10, 20):
        h1 = np.c_[s["p1"][start:start + 5].astype(np.float64), np.ones(5)] @ Kinv.T
        h2 = np.c_[s["p2"][start:start + 5].astype(np.float64), np.ones(5)] @ Kinv.T
        Es = oracle.em_5point(h1[:, :2], h2[:, :2])
        assert 1 <= len(Es) <= 10
        for E in Es:
            assert np.abs(np.einsum("ni,ij,nj->n", h2, E, h1)).max() < 1e-12
            sv = np.linalg.svd(E, compute_uv=False)
            assert sv[2] < 1e-8 and abs(sv[0] - sv[1]) < 1e-6
            assert abs(np.linalg.norm(E) - 1) < 1e-12
        # float32 pixel coordinates limit how exactly the true matrix satisfies the equations
        assert min(min(np.abs(E - E_true).max(), np.abs(E + E_true).max()) for E in Es) < 1e-3


def test_oracle_essential_subsets_and_pose(oracle):
    """Draws (the generator's recurrence evaluated here), RANSAC on contaminated data, and
    recoverPose returning the generating rotation and translation direction."""
    state = (1 << 64) - 1
    want = []
    for _ in range(10):
        state = (state & 0xFFFFFFFF) * 4164903690 + (state >> 32)
        want.append((state & 0xFFFFFFFF) % 1000)
    subs = oracle.em_subsets(1000, 2)
    assert len(set(want[:5])) == 5 and subs[0].tolist() == want[:5]

    s, E_true = _exact_two_view(1500, 4)
    rng = np.random.default_rng(5)
    p2 = s["p2"].copy()
    bad = rng.random(1500) < 0.3
    p2[bad] = np.stack([rng.uniform(0, 1280, bad.sum()), rng.uniform(0, 720, bad.sum())], 1)
    E, mask, iters = oracle.find_essential(s["p1"], p2, s["K"], 0, 0.99, 0.5)
    assert len(E) == 1 and 1 <= iters <= 1000
    kept = mask.astype(bool)
    assert (~bad)[kept].mean() > 0.99 and kept.sum() > 0.9 * (~bad).sum()
    assert min(np.abs(E[0] - E_true).max(), np.abs(E[0] + E_true).max()) < 1e-3
    cnt, R, t, mask2 = oracle.recover_pose(E[0], s["p1"], p2, s["K"], mask)
    assert cnt > 0.9 * kept.sum() and not (mask2 & ~mask).any()
    np.testing.assert_allclose(R, s["R"], atol=1e-3)
    assert t @ (s["t"] / np.linalg.norm(s["t"])) > 0.9999
    # decomposeEssentialMat: proper rotations, unit translation, E = [t]x R up to sign
    R1, R2, tt = oracle.decompose_essential(E_true)
    for Rk in (R1, R2):
        assert abs(np.linalg.det(Rk) - 1) < 1e-12 and np.abs(Rk @ Rk.T - np.eye(3)).max() < 1e-12
    assert abs(np.linalg.norm(tt) - 1) < 1e-12
    assert min(np.abs(Rk - s["R"]).max() for Rk in (R1, R2)) < 1e-6
    # fewer than five points: nothing
    assert len(oracle.find_essential(s["p1"][:4], p2[:4], s["K"])[0]) == 0


def test_lk_accumulator_modes_differ_only_in_the_band(oracle):
    """The oracle's two accumulator modes (exact integers, as the GPU; float, as OpenCV) on one
    shape: points that run the same iterations on every level agree within 1e-3 px; the rest are
    below 1 % and within the termination epsilon."""
    from sfm_amd import synth

    h, w, n = 240, 320, 1500
    prev, curr = synth.lk_pair(h, w, seed=63, shift=(1.7, -1.1), angle_deg=0.3)
    pts = synth.grid_points(h, w, n, seed=63, margin=4)
    e = oracle.pyrlk(prev, curr, pts, acc_exact=True, return_level_iters=True)
    f = oracle.pyrlk(prev, curr, pts, acc_exact=False, return_level_iters=True)
    assert e[3].shape == (4, n) and e[3].sum() > 4 * n
    same = (e[3] == f[3]).all(axis=0)
    np.testing.assert_array_equal(e[1][same], f[1][same])
    t = same & e[1].astype(bool)
    assert np.abs(e[0][t] - f[0][t]).max() < 1e-3
    assert (~same).mean() <= 0.01
    b = ~same & e[1].astype(bool) & f[1].astype(bool)
    if b.any():
        assert np.abs(e[0][b] - f[0][b]).max() <= 0.01
