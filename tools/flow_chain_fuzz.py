"""The device-resident front end on random frames against the CPU oracle, stage by stage
(developer tool; run on a GPU box): sfm_good_features_batch_dev -> sfm_pyrlk_batch_dev ->
sfm_filter_tracked_dev -> sfm_find_camera_pose_dev (AppSolver::findGoodImages, Source Files/
app_solver.cpp:40-79) for 1-4 frame pairs per call with random sizes, motions, windows, corner
settings, boundaries and pose methods.  Corners, tracked positions, the filtered lists, inlier
counts and masks must equal the oracle's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from sfm_amd import capi, synth  # noqa: E402


def main(argv=None):
    import torch

    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 30
    rng = np.random.default_rng(31337)
    dev = torch.device("cuda:0")
    ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    bad = 0
    for case in range(n_cases):
        H, W = int(rng.integers(120, 400)), int(rng.integers(160, 520))
        P = int(rng.integers(1, 5))
        maxc = int(rng.choice([50, 400, 1500]))
        win = int(rng.choice([9, 15, 21, 21, 31]))
        levels = int(rng.integers(0, 4))
        qual, mind = float(rng.choice([0.01, 0.05])), float(rng.choice([3.0, 8.0, 15.0]))
        method = int(rng.integers(0, 2))
        thr = float(rng.choice([0.5, 1.0]))
        bound = (int(rng.integers(0, 20)), int(rng.integers(0, 20)), W - int(rng.integers(20, 60)), H - int(rng.integers(20, 60)))
        if rng.integers(0, 4) == 0:
            bound = None
        f = float(rng.uniform(300, 900))
        K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
        prevs, currs = [], []
        for p in range(P):
            tex = synth.texture(H, W, seed=int(rng.integers(0, 10 ** 6)))
            if rng.integers(0, 5) == 0:
                tex[: H // 2] = 0.4                     # half the frame flat: few corners there
            prevs.append(synth.to_u8(tex))
            currs.append(synth.to_u8(synth.affine_warp(tex, (float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6))),
                                                       float(rng.uniform(-1.5, 1.5)))))
        prevs, currs = np.stack(prevs), np.stack(currs)
        d_prev, d_curr = torch.from_numpy(prevs).to(dev), torch.from_numpy(currs).to(dev)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)   # noqa: E731
        d_corners, d_info = z((P, maxc, 2), torch.float32), z((P, 3), torch.int32)
        d_next, d_status, d_err = z((P, maxc, 2), torch.float32), z((P, maxc), torch.uint8), z((P, maxc), torch.float32)
        d_p1, d_p2, d_counts = z((P, maxc, 2), torch.float32), z((P, maxc, 2), torch.float32), z((P,), torch.int32)
        d_E, d_R, d_t = z((P, 10, 9), torch.float64), z((P, 9), torch.float64), z((P, 3), torch.float64)
        d_mask, d_found, d_inl = z((P, maxc), torch.uint8), z((P,), torch.int32), z((P,), torch.int32)
        ctx.good_features_batch_dev(d_prev, P, W * H, W, H, W, maxc, qual, mind, d_corners, maxc, d_info)
        ctx.pyrlk_batch_dev(d_prev, d_curr, P, W * H, W, H, W, d_corners, maxc, (win, win), levels, 3, 30, 0.01,
                            1e-4, 10.0, d_next, d_status, d_err)
        d_valid = d_info[:, 0].contiguous()
        ctx.filter_tracked_dev(d_corners, d_next, d_status, d_valid, P, maxc, bound, d_p1, d_p2, d_counts)
        ctx.find_camera_pose_dev(d_p1, d_p2, maxc, d_counts, 0, P, K, method, 0.99, thr, d_E, d_R, d_t, d_mask,
                                 d_found, d_inl)
        ctx.synchronize()
        info, counts, inl = d_info.cpu().numpy(), d_counts.cpu().numpy(), d_inl.cpu().numpy()
        corners, p1s, p2s = d_corners.cpu().numpy(), d_p1.cpu().numpy(), d_p2.cpu().numpy()
        masks, Rs = d_mask.cpu().numpy(), d_R.cpu().numpy()
        why = ""
        for p in range(P):
            want_c = pyoracle.good_features(prevs[p], maxc, qual, mind)
            if info[p, 0] != len(want_c) or not np.array_equal(corners[p, : len(want_c)], want_c):
                why = f"pair {p}: corners {info[p, 0]} / {len(want_c)}"
                break
            if len(want_c) == 0:
                if counts[p] != 0 or inl[p] != 0:
                    why = f"pair {p}: no corners but counts {counts[p]} inliers {inl[p]}"
                    break
                continue
            nxt, st, err = pyoracle.pyrlk(prevs[p], currs[p], want_c, win=(win, win), max_level=levels, crit_type=3,
                                          max_count=30, epsilon=0.01, max_error=10.0, acc_exact=True)
            keep = st.astype(bool)
            if bound is not None:
                keep &= (bound[0] <= nxt[:, 0]) & (nxt[:, 0] < bound[0] + bound[2])
                keep &= (bound[1] <= nxt[:, 1]) & (nxt[:, 1] < bound[1] + bound[3])
            a, b = want_c[keep], nxt[keep]
            if counts[p] != len(a) or not (np.array_equal(p1s[p, : len(a)], a) and np.array_equal(p2s[p, : len(a)], b)):
                why = f"pair {p}: filtered lists {counts[p]} / {len(a)}"
                break
            if len(a) <= 5:
                ok = inl[p] == 0
            else:
                E, emask, _ = pyoracle.find_essential(a, b, K, method, 0.99, thr)
                if len(E) != 1 or pyoracle.essential_residual(E[0], a[0], b[0], K) > 1e-3:
                    ok = inl[p] == 0
                else:
                    cnt, R, t, pmask = pyoracle.recover_pose(E[0], a, b, K, emask)
                    ok = inl[p] == cnt and np.array_equal(masks[p, : len(a)], pmask)
                    if ok and cnt > 0:
                        ok = np.allclose(Rs[p].reshape(3, 3), R, atol=1e-7)
            if not ok:
                why = f"pair {p}: pose of {len(a)} correspondences (inliers {inl[p]})"
                break
        if why:
            bad += 1
            print(f"MISMATCH case {case}: {W}x{H} pairs={P} maxc={maxc} win={win} levels={levels} q={qual} d={mind} "
                  f"method={method} thr={thr} bound={bound}: {why}", flush=True)
        elif case % 5 == 0:
            print(f"case {case} ok ({W}x{H} pairs={P} corners={info[:, 0].tolist()} kept={counts.tolist()} "
                  f"inliers={inl.tolist()})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
