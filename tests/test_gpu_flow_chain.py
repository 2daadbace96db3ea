"""The tracking front end of the reference's main loop (AppSolver::findGoodImages, Source Files/
app_solver.cpp:40-79) as one device-resident chain, without a host round trip between stages:
generateFlowFeatures -> computeFlow -> filterPointsByStatusMask / ByBoundary -> findCameraPose.
Every stage is compared with the CPU oracle fed with the oracle's own previous stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, MAXC = 640, 480, 1500
WIN, LEVELS = (21, 21), 3
BOUND = (8, 8, W - 16, H - 16)
K = np.array([[700.0, 0, W / 2], [0, 700.0, H / 2], [0, 0, 1]])


def frames(P):
    from sfm_amd import synth

    prevs, currs = [], []
    for p in range(P):
        tex = synth.texture(H, W, seed=300 + p)
        prevs.append(synth.to_u8(tex))
        currs.append(synth.to_u8(synth.affine_warp(tex, (2.1 - p, -1.3 + 0.7 * p), 0.25 * (p + 1))))
    return np.stack(prevs), np.stack(currs)


def test_corners_to_pose_on_device(oracle):
    import torch

    from sfm_amd import capi

    P = 3
    prevs, currs = frames(P)
    ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    d_prev, d_curr = torch.from_numpy(prevs).cuda(), torch.from_numpy(currs).cuda()
    d_corners = torch.zeros((P, MAXC, 2), dtype=torch.float32, device="cuda")
    d_info = torch.zeros((P, 3), dtype=torch.int32, device="cuda")
    d_next = torch.zeros((P, MAXC, 2), dtype=torch.float32, device="cuda")
    d_status = torch.zeros((P, MAXC), dtype=torch.uint8, device="cuda")
    d_err = torch.zeros((P, MAXC), dtype=torch.float32, device="cuda")
    d_p1 = torch.zeros((P, MAXC, 2), dtype=torch.float32, device="cuda")
    d_p2 = torch.zeros((P, MAXC, 2), dtype=torch.float32, device="cuda")
    d_counts = torch.zeros(P, dtype=torch.int32, device="cuda")
    d_E = torch.zeros((P, 10, 9), dtype=torch.float64, device="cuda")
    d_R = torch.zeros((P, 9), dtype=torch.float64, device="cuda")
    d_t = torch.zeros((P, 3), dtype=torch.float64, device="cuda")
    d_mask = torch.zeros((P, MAXC), dtype=torch.uint8, device="cuda")
    d_found = torch.zeros(P, dtype=torch.int32, device="cuda")
    d_inl = torch.zeros(P, dtype=torch.int32, device="cuda")

    # the chain: nothing below reads device memory from the host until the end
    for p in range(P):
        ctx.good_features_dev(d_prev[p], W, H, W, MAXC, 0.01, 10.0, d_corners[p], MAXC, d_info[p])
    ctx.pyrlk_batch_dev(d_prev, d_curr, P, W * H, W, H, W, d_corners, MAXC, WIN, LEVELS, 3, 30, 0.01,
                        1e-4, 10.0, d_next, d_status, d_err)
    d_valid = d_info[:, 0].contiguous()
    ctx.filter_tracked_dev(d_corners, d_next, d_status, d_valid, P, MAXC, BOUND, d_p1, d_p2, d_counts)
    ctx.find_camera_pose_dev(d_p1, d_p2, MAXC, d_counts, 0, P, K, 0, 0.99, 0.5, d_E, d_R, d_t, d_mask, d_found, d_inl)
    ctx.synchronize()

    info, counts, inl = d_info.cpu().numpy(), d_counts.cpu().numpy(), d_inl.cpu().numpy()
    corners, p1s, p2s = d_corners.cpu().numpy(), d_p1.cpu().numpy(), d_p2.cpu().numpy()
    masks, Rs = d_mask.cpu().numpy(), d_R.cpu().numpy()
    for p in range(P):
        want_c = oracle.good_features(prevs[p], MAXC, 0.01, 10.0)
        assert info[p, 0] == len(want_c) > 200 and np.array_equal(corners[p, : len(want_c)], want_c)
        nxt, st, err = oracle.pyrlk(prevs[p], currs[p], want_c, win=WIN, max_level=LEVELS, crit_type=3,
                                    max_count=30, epsilon=0.01, max_error=10.0, acc_exact=True)
        keep = st.astype(bool)
        keep &= (BOUND[0] <= nxt[:, 0]) & (nxt[:, 0] < BOUND[0] + BOUND[2])
        keep &= (BOUND[1] <= nxt[:, 1]) & (nxt[:, 1] < BOUND[1] + BOUND[3])
        a, b = want_c[keep], nxt[keep]
        assert counts[p] == len(a) > 100
        assert np.array_equal(p1s[p, : len(a)], a) and np.array_equal(p2s[p, : len(a)], b)
        E, emask, _ = oracle.find_essential(a, b, K, 0, 0.99, 0.5)
        if len(E) != 1 or oracle.essential_residual(E[0], a[0], b[0], K) > 1e-3:
            assert inl[p] == 0
            continue
        cnt, R, t, pmask = oracle.recover_pose(E[0], a, b, K, emask)
        assert inl[p] == cnt
        assert np.array_equal(masks[p, : len(a)], pmask)
        np.testing.assert_allclose(Rs[p].reshape(3, 3), R, atol=1e-7)
    ctx.close()
