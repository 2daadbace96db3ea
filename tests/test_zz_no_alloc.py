"""GPU test of sfm_triangulate_noalloc_dev, the entry point that allocates nothing: its output
must equal sfm_triangulate_dev's bit for bit (same kernels, the deferred-point lists live in a
fixed device array instead of the context's workspace), on two calls in a row, the second with
a smaller input so that stale list contents of the first call would show; and on a fresh
context the call must not change the device's free memory.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(ctx, fn, s, n, dev, with_image):
    import torch

    d_p1 = torch.from_numpy(s["p1"][:n].copy()).to(dev)
    d_p2 = torch.from_numpy(s["p2"][:n].copy()).to(dev)
    d_X = torch.full((n, 3), -7.0, dtype=torch.float64, device=dev)
    d_mask = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    d_err = torch.full((n,), -7.0, dtype=torch.float64, device=dev)
    d_rgb = torch.full((n, 3), 9, dtype=torch.uint8, device=dev)
    if with_image:
        img = np.random.default_rng(5).integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)
        d_img = torch.from_numpy(img).to(dev)
        image = (d_img, 1280, 720, 1280 * 3)
    else:
        image = (None, 0, 0, 0)
    fn(s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], d_p1, d_p2, n, *image, 1e-4, 250.0,
       3.0, d_X, d_rgb if with_image else None, d_mask, d_err)
    ctx.synchronize()
    return d_X.cpu().numpy(), d_mask.cpu().numpy(), d_err.cpu().numpy(), d_rgb.cpu().numpy()


def test_noalloc_entry_point_allocates_nothing():
    """On a context that has never run anything (no workspace exists yet) the call must leave
    the device's free memory as it found it: a call that took the workspace route, or any other
    hipMalloc, would show as a drop.  Its output is compared with the allocating entry point
    afterwards."""
    import torch

    from sfm_amd import capi, synth

    dev = torch.device("cuda", 0)
    n = 1 << 20      # the largest accepted size: the allocating entry point needs 4 MiB of lists for it
    s = synth.triangulation_set(n, seed=31, outliers=0.3)
    d_p1, d_p2 = torch.from_numpy(s["p1"]).to(dev), torch.from_numpy(s["p2"]).to(dev)
    d_X = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_mask = torch.empty(n, dtype=torch.uint8, device=dev)
    d_X2, d_mask2 = torch.empty_like(d_X), torch.empty_like(d_mask)
    ctx = capi.Context(0)
    try:
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(dev)[0]
        ctx.triangulate_noalloc_dev(s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], d_p1, d_p2, n,
                                    None, 0, 0, 0, 1e-4, 250.0, 3.0, d_X, None, d_mask, None)
        ctx.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free_before
        # the allocating entry point on the same fresh context does take memory (its lists)
        ctx.triangulate_dev(s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], d_p1, d_p2, n,
                            None, 0, 0, 0, 1e-4, 250.0, 3.0, d_X2, None, d_mask2, None)
        ctx.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] < free_before
        assert torch.equal(d_X.view(torch.int64), d_X2.view(torch.int64)) and torch.equal(d_mask, d_mask2)
    finally:
        ctx.close()


def test_noalloc_triangulation_equals_allocating_entry_point():
    import torch

    from sfm_amd import capi, synth

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    try:
        # 30 % gross outliers: plenty of deferred points, so the lists are really used
        big = synth.triangulation_set(300_001, seed=21, outliers=0.3)
        small = synth.triangulation_set(40_000, seed=22, outliers=0.3)
        for s, n, with_image in ((big, 300_001, True), (small, 40_000, False), (small, 777, True)):
            want = _run(ctx, ctx.triangulate_dev, s, n, dev, with_image)
            got = _run(ctx, ctx.triangulate_noalloc_dev, s, n, dev, with_image)
            for g, w in zip(got, want):
                assert g.tobytes() == w.tobytes()
            assert np.isfinite(got[0]).all() and 0.3 < got[1].mean() < 0.9
        # the stated limit is enforced with a clear error and nothing is launched
        limit = capi.lib().sfm_triangulate_noalloc_max_points()
        assert limit == 1 << 20
        d = torch.zeros(8, device=dev)
        with pytest.raises(capi.SfmError, match="exceed"):
            ctx.triangulate_noalloc_dev(big["K"], big["prev_pose"], big["curr_pose"], big["R"], big["t"],
                                        d, d, limit + 1, None, 0, 0, 0, 1e-4, 250.0, 3.0, d, None, d, None)
        # the largest accepted size runs
        full = synth.triangulation_set(limit, seed=23)
        want = _run(ctx, ctx.triangulate_dev, full, limit, dev, False)
        got = _run(ctx, ctx.triangulate_noalloc_dev, full, limit, dev, False)
        for g, w in zip(got, want):
            assert g.tobytes() == w.tobytes()
    finally:
        ctx.close()
