"""Random image sizes, windows, pyramid depths, termination criteria and point sets through
sfm_pyrlk against the CPU oracle with exact accumulators (developer tool; run on a GPU box).
Positions, status and error must agree bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from sfm_amd import capi, synth  # noqa: E402


def main(argv=None):
    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 120
    fixed_win = int(argv[1]) if len(argv) > 1 else 0   # e.g. 21: every case with a 21 x 21 window
    batch = len(argv) > 2 and argv[2] == "batch"       # 2-6 image pairs per call (sfm_pyrlk_batch_dev)
    rng = np.random.default_rng(4242)
    ctx = capi.Context(0)
    bad = 0
    for case in range(n_cases):
        h = int(rng.integers(24, 260))
        w = int(rng.integers(24, 330))
        big = len(argv) > 2 and argv[2] == "big"          # frames up to full HD, thousands of points
        if big:
            h, w = int(rng.integers(400, 1100)), int(rng.integers(500, 2000))
        prev, curr = synth.lk_pair(h, w, seed=int(rng.integers(0, 10 ** 6)))
        if rng.integers(0, 4) == 0:       # a flat region: singular normal matrices
            prev[: h // 3, : w // 2] = 90
            curr[: h // 3, : w // 2] = 90
        win = (int(rng.integers(3, 34)), int(rng.integers(3, 34)))
        if fixed_win:
            win = (fixed_win, fixed_win)
        max_level = int(rng.integers(0, 5))
        crit = int(rng.choice([1, 2, 3]))
        max_count = int(rng.choice([1, 5, 30, 100]))
        eps = float(rng.choice([0.0, 0.001, 0.01, 0.3]))
        thr = float(rng.choice([1e-4, 1e-3, 0.0]))
        max_err = float(rng.choice([10.0, 3.0, 1e9]))
        n = int(rng.integers(1, 400)) if not big else int(rng.integers(500, 4000))
        pts = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)
        if rng.integers(0, 2):            # points on whole pixels
            pts[: n // 2] = np.round(pts[: n // 2])
        args = dict(win=win, max_level=max_level, crit_type=crit, max_count=max_count, epsilon=eps,
                    min_eig_threshold=thr, max_error=max_err)
        n_pairs = int(rng.integers(2, 7)) if batch else 1
        if batch:
            # several image pairs per call through the device entry point: every pair has its own
            # images and points; rows of the images may be further apart than the width
            import torch

            stride = w + int(rng.choice([0, 1, 13]))
            dev = torch.device("cuda:0")
            prevs, currs, ptss = [prev], [curr], [pts]
            for _ in range(n_pairs - 1):
                a, b = synth.lk_pair(h, w, seed=int(rng.integers(0, 10 ** 6)))
                prevs.append(a)
                currs.append(b)
                ptss.append(np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32))
            def pack(imgs):
                buf = np.zeros((n_pairs, h, stride), np.uint8)
                for k, im in enumerate(imgs):
                    buf[k, :, :w] = im
                return torch.from_numpy(buf).to(dev)
            d_prev, d_curr = pack(prevs), pack(currs)
            d_pts = torch.from_numpy(np.stack(ptss)).to(dev)
            d_nxt = torch.empty_like(d_pts)
            d_st = torch.empty((n_pairs, n), dtype=torch.uint8, device=dev)
            d_err = torch.empty((n_pairs, n), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()   # the uploads ran on torch's stream, the library has its own
            ctx.pyrlk_batch_dev(d_prev, d_curr, n_pairs, h * stride, w, h, stride, d_pts, n, win, max_level, crit,
                                max_count, eps, thr, max_err, d_nxt, d_st, d_err)
            ctx.synchronize()
            nxt, st, err = d_nxt.cpu().numpy().reshape(-1, 2), d_st.cpu().numpy().ravel(), d_err.cpu().numpy().ravel()
            res = [pyoracle.pyrlk(a, b, q, acc_exact=True, **args) for a, b, q in zip(prevs, currs, ptss)]
            onxt = np.concatenate([r[0] for r in res])
            ost = np.concatenate([r[1] for r in res])
            oerr = np.concatenate([r[2] for r in res])
        else:
            nxt, st, err = ctx.pyrlk(prev, curr, pts, **args)
            onxt, ost, oerr = pyoracle.pyrlk(prev, curr, pts, acc_exact=True, **args)
        ok = (np.array_equal(st, ost) and np.array_equal(nxt.view(np.uint32), onxt.view(np.uint32)) and
              np.array_equal(err.view(np.uint32), oerr.view(np.uint32)))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: {w}x{h} win={win} levels={max_level} crit={crit} count={max_count} "
                  f"eps={eps} thr={thr} max_err={max_err} n={n} pairs={n_pairs}: status {int((st != ost).sum())}, "
                  f"positions {int((nxt != onxt).any(1).sum())}, errors {int((err != oerr).sum())}", flush=True)
        elif case % 20 == 0:
            print(f"case {case} ok ({w}x{h} win={win} levels={max_level} n={n} tracked={int(st.sum())})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
