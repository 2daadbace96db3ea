"""GPU parity tests of the matching stage: HIP kernels through the C ABI against
the CPU oracle and the committed golden vectors.  Integer work: bit-exact."""
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
    return [(int(a["queryIdx"]), int(a["trainIdx"]), int(a["imgIdx"]), float(a["distance"]))
            for a in m]


def make_frames(rng, n, d, flip_p=0.05, reuse=0.7):
    a = rng.integers(0, 256, size=(n, d), dtype=np.uint8)
    perm = rng.permutation(n)
    b = a[perm].copy()
    nre = int(reuse * n)
    flips = rng.random((nre, d * 8)) < flip_p
    b[:nre] ^= np.packbits(flips, axis=1)
    b[nre:] = rng.integers(0, 256, size=(n - nre, d), dtype=np.uint8)
    return a, b


@pytest.mark.parametrize("d", [32, 61, 64])
def test_hamming_knn2_golden(ctx, d):
    from sfm_amd import capi

    g = load_golden("hamming.npz")
    idx, dist = ctx.knn2(capi.NORM_HAMMING, g[f"q{d}"], g[f"t{d}"])
    np.testing.assert_array_equal(idx, g[f"idx{d}"])
    np.testing.assert_array_equal(dist, g[f"dist{d}"])


@pytest.mark.parametrize("nt", [1, 2, 3])
def test_hamming_tiny_train(ctx, nt):
    from sfm_amd import capi

    g = load_golden("hamming.npz")
    idx, dist = ctx.knn2(capi.NORM_HAMMING, g[f"tiny_q_nt{nt}"], g[f"tiny_t_nt{nt}"])
    np.testing.assert_array_equal(idx, g[f"tiny_idx_nt{nt}"])
    np.testing.assert_array_equal(dist, g[f"tiny_dist_nt{nt}"])


def test_hamming_empty_inputs(ctx):
    from sfm_amd import capi

    q = np.zeros((5, 32), np.uint8)
    idx, dist = ctx.knn2(capi.NORM_HAMMING, q, np.zeros((0, 32), np.uint8))
    assert (idx == -1).all() and (dist == np.finfo(np.float32).max).all()
    idx, dist = ctx.knn2(capi.NORM_HAMMING, np.zeros((0, 32), np.uint8), q)
    assert idx.shape == (0, 2)
    assert len(ctx.robust_matches(capi.NORM_HAMMING, q, np.zeros((0, 32), np.uint8), 0.8)) == 0
    assert len(ctx.ratio_matches(capi.NORM_HAMMING, q, q[:1], 0.8)) == 0


@pytest.mark.parametrize("path", ["", "0"])
@pytest.mark.parametrize("nq,nt,d", [(2000, 2000, 32), (1500, 2300, 32), (777, 1031, 61),
                                     (1024, 512, 64), (300, 5000, 16), (64, 257, 128),
                                     (100, 90, 256), (513, 700, 7), (700, 9000, 32),
                                     (129, 128, 64), (600, 8193, 16)])
def test_hamming_knn2_vs_oracle(ctx, oracle, monkeypatch, nq, nt, d, path):
    """Single-direction 2-NN, ties included, through the kernel the library picks (the
    matrix-core kernel for 16/32/64-byte rows and at least 128 rows a side; above 8192 train
    rows it works in more than one train range) and through the scalar kernel
    (SFM_PAIR_KERNEL=0)."""
    from sfm_amd import capi

    monkeypatch.setenv("SFM_PAIR_KERNEL", path)
    rng = np.random.default_rng(nq * 31 + nt)
    q = rng.integers(0, 256, size=(nq, d), dtype=np.uint8)
    t = rng.integers(0, 256, size=(nt, d), dtype=np.uint8)
    # few distinct values per byte -> many exact distance ties
    if d <= 16:
        q &= 0x03
        t &= 0x03
    t[nt // 2] = t[3]
    q[5] = t[7]
    idx, dist = ctx.knn2(capi.NORM_HAMMING, q, t)
    oidx, odist = oracle.knn2_hamming(q, t)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)


def test_hamming_strided_rows(ctx, oracle):
    from sfm_amd import capi

    rng = np.random.default_rng(5)
    big_q = rng.integers(0, 256, size=(400, 48), dtype=np.uint8)
    big_t = rng.integers(0, 256, size=(600, 40), dtype=np.uint8)
    q, t = big_q[:, :32], big_t[:, :32]  # row stride larger than the row
    idx = np.empty((400, 2), np.int32)
    dist = np.empty((400, 2), np.float32)
    import ctypes as C
    ctx.check(capi.lib().sfm_knn2(ctx._h, C.c_int(0), capi._ptr(big_q), C.c_int(400),
                                  C.c_size_t(48), capi._ptr(big_t), C.c_int(600), C.c_size_t(40),
                                  C.c_int(32), capi._ptr(idx), capi._ptr(dist)))
    oidx, odist = oracle.knn2_hamming(q, t)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)


@pytest.mark.parametrize("d", [32, 61, 64])
@pytest.mark.parametrize("tag,ratio", [("r50", 0.5), ("r80", 0.8)])
def test_ratio_and_cross_golden(ctx, d, tag, ratio):
    from sfm_amd import capi

    g = load_golden("hamming.npz")
    q, t = g[f"q{d}"], g[f"t{d}"]
    f = ctx.ratio_matches(capi.NORM_HAMMING, q, t, ratio)
    assert [(a, b, 0, c) for a, b, c in g[f"fwd{d}_{tag}"].tolist()] == _tuples(f)
    b = ctx.ratio_matches(capi.NORM_HAMMING, t, q, ratio)
    assert [(a, b_, 0, c) for a, b_, c in g[f"bwd{d}_{tag}"].tolist()] == _tuples(b)
    x = ctx.robust_matches(capi.NORM_HAMMING, q, t, ratio)
    assert [(a, b_, 0, c) for a, b_, c in g[f"cross{d}_{tag}"].tolist()] == _tuples(x)


@pytest.mark.parametrize("n_prev,n_curr,d,ratio", [(2000, 2000, 32, 0.5), (2000, 1700, 32, 0.8),
                                                   (900, 1300, 61, 0.7), (8192, 8192, 32, 0.8)])
def test_robust_matches_vs_oracle(ctx, oracle, n_prev, n_curr, d, ratio):
    from sfm_amd import capi

    rng = np.random.default_rng(1234 + n_prev + d)
    n = max(n_prev, n_curr)
    a, b = make_frames(rng, n, d)
    a, b = a[:n_prev], b[:n_curr]
    got = ctx.robust_matches(capi.NORM_HAMMING, a, b, ratio)
    want = oracle.robust_matches(oracle.NORM_HAMMING, a, b, ratio)
    assert len(want) > 100
    assert _tuples(got) == _tuples(want)
    # size-independent properties: sorted by curr index, one-to-one, symmetric
    assert (np.diff(got["trainIdx"]) > 0).all()
    assert len(set(got["queryIdx"].tolist())) == len(got)
    swapped = ctx.robust_matches(capi.NORM_HAMMING, b, a, ratio)
    assert sorted(zip(swapped["trainIdx"].tolist(), swapped["queryIdx"].tolist())) == \
        sorted(zip(got["queryIdx"].tolist(), got["trainIdx"].tolist()))


def test_self_match_distance_zero(ctx):
    from sfm_amd import capi

    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, size=(3000, 32), dtype=np.uint8)
    idx, dist = ctx.knn2(capi.NORM_HAMMING, a, a)
    np.testing.assert_array_equal(idx[:, 0], np.arange(3000))
    assert (dist[:, 0] == 0).all()


def test_match_pairs_batched(ctx, oracle):
    """All pairs of a short chained sequence in one call, ragged frame sizes."""
    import torch

    from sfm_amd import capi

    rng = np.random.default_rng(77)
    n_frames, rows, d = 6, 700, 32
    frames = [rng.integers(0, 256, size=(rows, d), dtype=np.uint8)]
    for _ in range(n_frames - 1):
        frames.append(make_frames(rng, rows, d)[1] ^ 0)
        frames[-1][: rows // 2] = frames[-2][rng.permutation(rows)[: rows // 2]]
    counts = np.array([700, 650, 700, 512, 699, 1], np.int32)
    pairs = np.array([(i, j) for i in range(n_frames) for j in range(i + 1, n_frames)], np.int32)
    stack = np.stack(frames)
    dev = torch.device("cuda:0")
    d_desc = torch.from_numpy(stack).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_pairs = torch.from_numpy(pairs).to(dev)
    cap = rows
    d_matches = torch.zeros((len(pairs), cap, 4), dtype=torch.int32, device=dev)
    d_mc = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, n_frames, rows * d, rows, d_counts, d, d_pairs,
                        len(pairs), 0.8, d_matches, cap, d_mc)
    ctx.synchronize()
    mc = d_mc.cpu().numpy()
    m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(len(pairs), cap)
    total = 0
    for p, (i, j) in enumerate(pairs):
        want = oracle.robust_matches(oracle.NORM_HAMMING, stack[i][: counts[i]],
                                     stack[j][: counts[j]], 0.8)
        assert mc[p] == len(want)
        assert _tuples(m[p][: mc[p]]) == _tuples(want)
        total += len(want)
    assert total > 500


def test_robust_matches_strided_rows(ctx, oracle):
    """Descriptor rows that are views into wider host arrays go through the gather path."""
    from sfm_amd import capi, synth

    a, b = synth.descriptor_chain(2, 900, 32, seed=77)
    wide_a = np.zeros((900, 50), np.uint8)
    wide_b = np.zeros((900, 64), np.uint8)
    wide_a[:, 3:35] = a
    wide_b[:, 16:48] = b
    for _ in range(2):
        got = ctx.robust_matches(capi.NORM_HAMMING, wide_a[:, 3:35], wide_b[:, 16:48], 0.7)
        want = oracle.robust_matches(capi.NORM_HAMMING, a, b, 0.7)
        assert len(want) > 100 and got.tobytes() == want.tobytes()
        fwd = ctx.ratio_matches(capi.NORM_HAMMING, wide_a[:, 3:35], wide_b[:, 16:48], 0.7)
        idx, dist = oracle.knn2_hamming(a, b)
        assert fwd.tobytes() == oracle.ratio_filter(idx, dist, 900, 0.7).tobytes()


@pytest.mark.parametrize("d,rows,counts", [
    (32, 2600, [2600, 1025, 1024, 257, 2, 2049]),   # workgroup / batch boundaries of the pair kernel
    (16, 1500, [1500, 1023, 300, 1, 1280, 64]),     # 16-byte descriptors: the 4-word instantiation
    (13, 1300, [1300, 511, 257, 90, 1299, 3]),      # padded to 16 bytes first
    (64, 1300, [1300, 300, 700, 2, 1025, 65]),      # 64-byte descriptors: two rows of prev per lane
])
@pytest.mark.parametrize("path", ["", "0", "1", "2"])
def test_match_pairs_ragged(ctx, oracle, monkeypatch, d, rows, counts, path):
    """Every pair of six frames with very different row counts, against the oracle, through
    each of the three Hamming kernels (SFM_PAIR_KERNEL: 0 single-direction, 1 one-pass pair
    kernel, 2 matrix-core kernel) and through the one the library picks by size."""
    import torch

    from sfm_amd import capi

    monkeypatch.setenv("SFM_PAIR_KERNEL", path)
    rng = np.random.default_rng(1000 + d)
    n_frames = len(counts)
    base = rng.integers(0, 256, size=(rows, d), dtype=np.uint8)
    frames = []
    for f in range(n_frames):
        fr = base.copy()
        flip = rng.random(size=(rows, d * 8)) < 0.04
        fr ^= np.packbits(flip, axis=1, bitorder="little")
        fresh = rng.random(rows) < 0.3
        fr[fresh] = rng.integers(0, 256, size=(int(fresh.sum()), d), dtype=np.uint8)
        frames.append(fr[rng.permutation(rows)])
    stack = np.stack(frames)
    counts = np.array(counts, np.int32)
    pairs = np.array([(i, j) for i in range(n_frames) for j in range(n_frames) if i != j], np.int32)
    dev = torch.device("cuda:0")
    d_desc = torch.from_numpy(stack).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_pairs = torch.from_numpy(pairs).to(dev)
    cap = rows
    d_matches = torch.zeros((len(pairs), cap, 4), dtype=torch.int32, device=dev)
    d_mc = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, n_frames, rows * d, rows, d_counts, d, d_pairs,
                        len(pairs), 0.75, d_matches, cap, d_mc)
    ctx.synchronize()
    mc = d_mc.cpu().numpy()
    m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(len(pairs), cap)
    total = 0
    for p, (i, j) in enumerate(pairs):
        want = oracle.robust_matches(oracle.NORM_HAMMING, stack[i][: counts[i]],
                                     stack[j][: counts[j]], 0.75)
        assert mc[p] == len(want), (i, j)
        assert _tuples(m[p][: mc[p]]) == _tuples(want), (i, j)
        total += len(want)
    assert total > 200


@pytest.mark.parametrize("d,kernel", [(61, "hamming_mfma_kernel"), (32, "hamming_mfma_kernel"),
                                      (128, None)])
def test_row_width_picks_the_kernel(ctx, oracle, d, kernel):
    """AKAZE's 61-byte rows are zero-padded to 64 and go through the matrix-core kernel like 32-
    and 64-byte rows; widths it has no shape for (128) must not.  The per-kernel launch
    counters of the context say which kernel ran; the result is checked as well."""
    from sfm_amd import capi

    rng = np.random.default_rng(61 + d)
    a, b = make_frames(rng, 700, d)
    b = b[:650]
    ctx.set_profiling(True)
    before = ctx.profile()
    got = ctx.robust_matches(capi.NORM_HAMMING, a, b, 0.8)
    after = ctx.profile()
    ctx.set_profiling(False)
    ran = {k for k, (_, n) in after.items() if n > before.get(k, (0, 0))[1]}
    if kernel:
        assert kernel in ran, ran
    else:
        assert "hamming_mfma_kernel" not in ran, ran
    want = oracle.robust_matches(oracle.NORM_HAMMING, a, b, 0.8)
    assert len(want) > 100 and _tuples(got) == _tuples(want)


def test_two_direction_path_agrees(oracle):
    """SFM_PAIR_KERNEL=0 routes pair matching through the single-direction kernel twice; it
    must give what the default path gives.  The library reads the switch on every call; the
    child process is only there so that the variable is set before the library loads and
    nothing else in this test process sees it."""
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from sfm_amd import capi, synth
from oracle import pyoracle
a, b = synth.descriptor_chain(2, 3000, 32, seed=21)
b = b[:2500]
got = capi.Context(0).robust_matches(capi.NORM_HAMMING, a, b, 0.7)
want = pyoracle.robust_matches(pyoracle.NORM_HAMMING, a, b, 0.7)
assert len(want) > 300 and got.tobytes() == want.tobytes()
print("two-direction ok", len(got))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SFM_PAIR_KERNEL="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "two-direction ok" in out.stdout


def test_match_pairs_many_problems(ctx, oracle):
    """480 problems of up to 1100 rows in one launch: enough workgroups for the matrix-core
    kernel's larger shape (four query tiles per wave, 1024 queries per workgroup), ragged
    row counts around its tile and workgroup boundaries."""
    import torch

    from sfm_amd import capi

    rng = np.random.default_rng(4242)
    n_frames, rows, d = 16, 1100, 32
    base = rng.integers(0, 256, size=(rows, d), dtype=np.uint8)
    frames = []
    for f in range(n_frames):
        fr = base.copy()
        flip = rng.random(size=(rows, d * 8)) < 0.05
        fr ^= np.packbits(flip, axis=1, bitorder="little")
        fresh = rng.random(rows) < 0.3
        fr[fresh] = rng.integers(0, 256, size=(int(fresh.sum()), d), dtype=np.uint8)
        frames.append(fr[rng.permutation(rows)])
    stack = np.stack(frames)
    counts = np.array([1100, 1025, 1024, 1023, 513, 512, 129, 128, 127, 65, 64, 33, 2, 1, 1099, 700],
                      np.int32)
    pairs = np.array([(i, j) for i in range(n_frames) for j in range(n_frames) if i != j], np.int32)
    dev = torch.device("cuda:0")
    d_desc = torch.from_numpy(stack).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_pairs = torch.from_numpy(pairs).to(dev)
    cap = rows
    d_matches = torch.zeros((len(pairs), cap, 4), dtype=torch.int32, device=dev)
    d_mc = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, n_frames, rows * d, rows, d_counts, d, d_pairs,
                        len(pairs), 0.8, d_matches, cap, d_mc)
    ctx.synchronize()
    mc = d_mc.cpu().numpy()
    m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(len(pairs), cap)
    total = 0
    for p, (i, j) in enumerate(pairs):
        want = oracle.robust_matches(oracle.NORM_HAMMING, stack[i][: counts[i]],
                                     stack[j][: counts[j]], 0.8)
        assert mc[p] == len(want), (i, j)
        assert _tuples(m[p][: mc[p]]) == _tuples(want), (i, j)
        total += len(want)
    assert total > 2000
