"""GPU tests of windowed / all-pairs matching as the sharded configuration runs it: the compact
(u16 q, u16 t, u16 distance) match records of sfm_match_pairs_compact_dev, HipPairMatcher +
match_all_pairs, the reassembly into pair order and the RCCL collectives with a one-rank group.
Everything is compared with the CPU oracle's findRobustMatches pair by pair, bit for bit.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KPS, D = 8192, 32


def _oracle_records(oracle, desc, counts, pairs, ratio):
    from sfm_amd import allpairs

    out, cs = [], []
    for a, b in pairs:
        m = oracle.robust_matches(oracle.NORM_HAMMING, desc[a][:counts[a]], desc[b][:counts[b]], ratio, threads=0)
        r = np.zeros(len(m), allpairs.MATCH16_DTYPE)
        r["queryIdx"], r["trainIdx"], r["distance"] = m["queryIdx"], m["trainIdx"], m["distance"]
        out.append(r)
        cs.append(len(m))
    return np.array(cs, np.int64), np.concatenate(out) if out else np.zeros(0, allpairs.MATCH16_DTYPE)


def _records(t):
    from sfm_amd import allpairs

    return t.cpu().numpy().view(np.uint16).reshape(-1, 3).copy().view(allpairs.MATCH16_DTYPE).reshape(-1)


@pytest.fixture(scope="module")
def env():
    import torch

    from sfm_amd import capi

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield ctx, dev
    ctx.close()


def test_window_of_12_frames_equals_oracle(env, oracle):
    """12 frames x 8192 x 32 B, all 66 pairs, one rank: counts, offsets and every record."""
    import torch

    from sfm_amd import allpairs, capi, synth

    ctx, dev = env
    F = 12
    desc = synth.descriptor_chain(F, KPS, D, seed=1234)
    pairs = allpairs.window_pairs(F)
    d_desc = torch.from_numpy(desc).to(dev)
    # a small first guess of the output size forces the grow-and-repeat path
    matcher = allpairs.HipPairMatcher(ctx, capi.NORM_HAMMING, d_desc, KPS, D, 0.8, matches_per_pair=64)
    c, off, m = allpairs.match_all_pairs(pairs, matcher, 0, 1)
    want_c, want = _oracle_records(oracle, desc, [KPS] * F, pairs, 0.8)
    np.testing.assert_array_equal(c, want_c)
    np.testing.assert_array_equal(off, np.concatenate([[0], np.cumsum(want_c)]))
    assert want_c.sum() > 20000 and _records(m).tobytes() == want.tobytes()
    # the same through the cv::DMatch entry point: identical lists
    d_pairs = torch.from_numpy(pairs).to(dev)
    d_m = torch.zeros((len(pairs), KPS, 4), dtype=torch.int32, device=dev)
    d_c = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, F, KPS * D, KPS, None, D, d_pairs, len(pairs), 0.8, d_m,
                        KPS, d_c)
    ctx.synchronize()
    np.testing.assert_array_equal(d_c.cpu().numpy(), want_c)
    full = d_m.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(len(pairs), KPS)
    k = int(np.argmax(want_c))
    np.testing.assert_array_equal(full[k, :want_c[k]]["queryIdx"], want[off[k]:off[k + 1]]["queryIdx"])


def test_ragged_frames_two_shards_reassembled(env, oracle):
    """Frames of different sizes (one with a single row, one empty); the two shards of a
    two-rank run are computed one after the other on this GPU and put back into pair order by
    the same reassembly (device segment copy) the ranks use after their all-gather."""
    import torch

    from sfm_amd import allpairs, capi, synth

    ctx, dev = env
    F, rows = 9, 1500
    desc = synth.descriptor_chain(F, rows, D, seed=77)
    counts = np.array([1500, 1400, 1, 1500, 0, 777, 1500, 128, 1499], np.int32)
    pairs = allpairs.window_pairs(F)
    d_desc = torch.from_numpy(desc).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    matcher = allpairs.HipPairMatcher(ctx, capi.NORM_HAMMING, d_desc, rows, D, 0.85, d_counts=d_counts)
    want_c, want = _oracle_records(oracle, desc, counts, pairs, 0.85)
    assert want_c.sum() > 2000 and (want_c == 0).any()
    world = 2
    per_rank = (len(pairs) + world - 1) // world
    shard_c, shard_p = [], []
    for r in range(world):
        c, p = matcher(pairs[allpairs.shard_indices(len(pairs), r, world)])
        shard_c.append(c.cpu().numpy().astype(np.int64))
        shard_p.append(p.clone())
    width = max(int(p.shape[0]) for p in shard_p)
    all_c = np.full((world, per_rank), -1, np.int64)
    all_p = torch.zeros((world * width, 3), dtype=torch.int16, device=dev)
    for r in range(world):
        all_c[r, :len(shard_c[r])] = shard_c[r]
        all_p[r * width: r * width + shard_p[r].shape[0]] = shard_p[r]
    c, off, m = allpairs.reassemble(all_c, all_p, width, len(pairs), world, matcher.gather_segments)
    np.testing.assert_array_equal(c, want_c)
    assert _records(m).tobytes() == want.tobytes()


def test_collectives_with_a_one_rank_rccl_group(env, oracle):
    """match_all_pairs' count / payload all-gathers and PaddedGather over RCCL (world size 1)."""
    import torch
    import torch.distributed as dist

    from sfm_amd import allpairs, capi, synth

    ctx, dev = env
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        F, rows = 5, 2048
        desc = synth.descriptor_chain(F, rows, D, seed=5)
        pairs = allpairs.window_pairs(F)
        d_desc = torch.from_numpy(desc).to(dev)
        matcher = allpairs.HipPairMatcher(ctx, capi.NORM_HAMMING, d_desc, rows, D, 0.8)
        c, off, m = allpairs.match_all_pairs(pairs, matcher, 0, 1, always_exchange=True)
        want_c, want = _oracle_records(oracle, desc, [rows] * F, pairs, 0.8)
        np.testing.assert_array_equal(c, want_c)
        assert _records(m).tobytes() == want.tobytes()
        # fixed-capacity gather of the per-frame-pair bench
        P, cap = 4, rows
        d_pairs = torch.from_numpy(pairs[:P]).to(dev)
        d_m = torch.zeros((P, cap, 4), dtype=torch.int32, device=dev)
        d_c = torch.zeros(P, dtype=torch.int32, device=dev)
        ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, F, rows * D, rows, None, D, d_pairs, P, 0.8, d_m, cap, d_c)
        g = allpairs.PaddedGather(P, cap, 1, dev)
        g.start(d_m, d_c)
        g.wait()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(g.counts[0].cpu().numpy(), want_c[:P])
        assert torch.equal(g.matches[0], d_m)
        # the same with the compact records written back to back (what the bench exchanges)
        d_k = torch.zeros((P * cap, 3), dtype=torch.int16, device=dev)
        d_off = torch.zeros(P + 1, dtype=torch.int64, device=dev)
        ctx.match_pairs_compact_dev(capi.NORM_HAMMING, d_desc, F, rows * D, rows, None, D, d_pairs, P, 0.8, d_k,
                                    P * cap, d_c, d_off)
        gk = allpairs.PaddedGather(P, cap, 1, dev, compact=True)
        gk.start(d_k, d_c)
        gk.wait()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(gk.counts[0].cpu().numpy(), want_c[:P])
        total = int(want_c[:P].sum())
        assert _records(gk.matches[0][:total]).tobytes() == want[:total].tobytes()
    finally:
        dist.destroy_process_group()


def test_one_pair_in_query_slabs_equals_oracle(env, oracle):
    """Fewer pairs than GPUs: one 8192 x 8192 pair split by query rows.  The slabs of three
    ranks computed one after the other on this GPU reassemble to the oracle's findRobustMatches;
    so does match_all_pairs' slab route on a one-rank group."""
    import torch

    from sfm_amd import allpairs, capi, synth

    ctx, dev = env
    desc = synth.descriptor_chain(3, KPS, D, seed=77)
    d_desc = torch.from_numpy(desc).to(dev)
    slab = allpairs.HipSlabMatcher(ctx, capi.NORM_HAMMING, d_desc, 0.8)
    want = oracle.robust_matches(oracle.NORM_HAMMING, desc[2], desc[1], 0.8, threads=0)
    assert len(want) > 1000
    world = 3
    parts = {k: [] for k in ("if", "df", "ib", "db")}
    for r in range(world):
        lo, hi = allpairs.slab_bounds(KPS, r, world)
        i, d_ = slab.knn_slab(2, 1, lo, hi)
        parts["if"].append(i), parts["df"].append(d_)
        i, d_ = slab.knn_slab(1, 2, lo, hi)
        parts["ib"].append(i), parts["db"].append(d_)
    n, rec = slab.cross(*(torch.cat(parts[k]) for k in ("if", "df", "ib", "db")))
    got = _records(rec)
    assert n == len(want)
    assert np.array_equal(got["queryIdx"], want["queryIdx"]) and np.array_equal(got["trainIdx"], want["trainIdx"])
    assert np.array_equal(got["distance"], want["distance"].astype(np.uint16))
    # the same through match_all_pairs (world 1: the slab is the whole frame)
    c, off, m = allpairs.match_all_pairs(np.array([[2, 1]]), None, 0, 1, shard="slab", slab=slab)
    assert c.tolist() == [len(want)] and _records(m).tobytes() == got.tobytes()
    # the raw entry point on ragged sizes: forward 2-NN of 1000 rows against 37, backward of 37 against 1000
    a, b = desc[0][:1000], desc[1][:37]
    d_a, d_b = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    fi, fd = torch.empty((1000, 2), dtype=torch.int32, device=dev), torch.empty((1000, 2), dtype=torch.float32, device=dev)
    bi, bd = torch.empty((37, 2), dtype=torch.int32, device=dev), torch.empty((37, 2), dtype=torch.float32, device=dev)
    ctx.knn2_dev(capi.NORM_HAMMING, d_a, 1000, D, d_b, 37, D, D, fi, fd)
    ctx.knn2_dev(capi.NORM_HAMMING, d_b, 37, D, d_a, 1000, D, D, bi, bd)
    out = torch.zeros((37, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.ratio_cross_dev(fi, fd, 1000, bi, bd, 37, 0.9, out, 37, cnt)
    ctx.synchronize()
    w2 = oracle.robust_matches(oracle.NORM_HAMMING, a, b, 0.9)
    assert int(cnt.item()) == len(w2)
    assert out[: len(w2)].cpu().numpy().view(capi.DMATCH_DTYPE).reshape(-1).tobytes() == w2.tobytes()


def test_full_window_512_frames(env, oracle):
    """The sharded configuration at the size it is quoted on: 512 frames x 8192 x 32 B, all
    130 816 pairs through HipPairMatcher on one rank.  Every pair's count is in range and the
    offsets are their running sum; one pair in a thousand is compared record by record with the
    oracle's findRobustMatches."""
    import torch

    from sfm_amd import allpairs, capi, synth

    ctx, dev = env
    F = 512
    desc = synth.descriptor_chain(F, KPS, D, seed=1234)
    pairs = allpairs.window_pairs(F)
    assert len(pairs) == 130816
    d_desc = torch.from_numpy(desc).to(dev)
    matcher = allpairs.HipPairMatcher(ctx, capi.NORM_HAMMING, d_desc, KPS, D, 0.5)
    c, off, m = allpairs.match_all_pairs(pairs, matcher, 0, 1)
    assert c.shape == (len(pairs),) and (c >= 0).all() and (c <= KPS).all()
    np.testing.assert_array_equal(off, np.concatenate([[0], np.cumsum(c)]))
    assert m.shape[0] == off[-1]
    # neighbouring frames share most of their rows; distant ones next to nothing
    near = c[pairs[:, 1] - pairs[:, 0] == 1]
    assert near.min() > 1000 and c[pairs[:, 1] - pairs[:, 0] > 40].max() < near.min()
    rec = _records(m)
    for i in range(17, len(pairs), 1000):
        a, b = pairs[i]
        want = oracle.robust_matches(oracle.NORM_HAMMING, desc[a], desc[b], 0.5, threads=0)
        got = rec[off[i]: off[i + 1]]
        assert len(got) == len(want), (a, b)
        assert np.array_equal(got["queryIdx"], want["queryIdx"]) and np.array_equal(got["trainIdx"], want["trainIdx"])
        assert np.array_equal(got["distance"], want["distance"].astype(np.uint16))


def test_compact_entry_point_edge_cases(env):
    """Empty frames, no pairs, an output buffer that is too small, unsupported norms."""
    import torch

    from sfm_amd import capi

    ctx, dev = env
    d_desc = torch.zeros((3, 64, D), dtype=torch.uint8, device=dev)
    d_pairs = torch.tensor([[0, 1], [1, 2]], dtype=torch.int32, device=dev)
    d_out = torch.zeros((16, 3), dtype=torch.int16, device=dev)
    d_c = torch.full((2,), 7, dtype=torch.int32, device=dev)
    d_off = torch.full((3,), 7, dtype=torch.int64, device=dev)
    # frames without rows (the Hamming plan once divided by zero here) through both entry points
    ctx.match_pairs_compact_dev(capi.NORM_HAMMING, d_desc, 3, 0, 0, None, D, d_pairs, 2, 0.5, d_out, 16, d_c, d_off)
    ctx.synchronize()
    assert d_c.tolist() == [0, 0] and d_off.tolist() == [0, 0, 0]
    d_c.fill_(7)
    ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, 3, 0, 0, None, D, d_pairs, 2, 0.5, None, 0, d_c)
    ctx.match_pairs_dev(capi.NORM_L2, d_desc, 3, 0, 0, None, D, d_pairs, 2, 0.5, None, 0, d_c)
    ctx.synchronize()
    assert d_c.tolist() == [0, 0]
    # all-zero descriptors: every distance ties at 0, no ratio test passes
    ctx.match_pairs_compact_dev(capi.NORM_HAMMING, d_desc, 3, 64 * D, 64, None, D, d_pairs, 2, 0.5, d_out, 16, d_c,
                                d_off)
    ctx.synchronize()
    assert d_off.tolist() == [0, 0, 0]
    # identical distinct rows in both frames: every row matches itself; 4 slots hold the first 4
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, size=(64, D), dtype=np.uint8)
    d_desc = torch.from_numpy(np.stack([frame, frame, frame])).to(dev)
    d_out.fill_(-1)
    ctx.match_pairs_compact_dev(capi.NORM_HAMMING, d_desc, 3, 64 * D, 64, None, D, d_pairs, 2, 0.5, d_out, 4, d_c,
                                d_off)
    ctx.synchronize()
    assert d_c.tolist() == [64, 64] and d_off.tolist() == [0, 64, 128]
    assert d_out[:4].tolist() == [[i, i, 0] for i in range(4)] and (d_out[4:] == -1).all()
    with pytest.raises(capi.SfmError):
        ctx.match_pairs_compact_dev(capi.NORM_L2, d_desc, 3, 64 * D, 64, None, D, d_pairs, 2, 0.5, d_out, 16, d_c,
                                    d_off)
