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


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    fixed_win = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # e.g. 21: every case with a 21 x 21 window
    rng = np.random.default_rng(4242)
    ctx = capi.Context(0)
    bad = 0
    for case in range(n_cases):
        h = int(rng.integers(24, 260))
        w = int(rng.integers(24, 330))
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
        n = int(rng.integers(1, 400))
        pts = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)
        if rng.integers(0, 2):            # points on whole pixels
            pts[: n // 2] = np.round(pts[: n // 2])
        args = dict(win=win, max_level=max_level, crit_type=crit, max_count=max_count, epsilon=eps,
                    min_eig_threshold=thr, max_error=max_err)
        nxt, st, err = ctx.pyrlk(prev, curr, pts, **args)
        onxt, ost, oerr = pyoracle.pyrlk(prev, curr, pts, acc_exact=True, **args)
        ok = (np.array_equal(st, ost) and np.array_equal(nxt.view(np.uint32), onxt.view(np.uint32)) and
              np.array_equal(err.view(np.uint32), oerr.view(np.uint32)))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: {w}x{h} win={win} levels={max_level} crit={crit} count={max_count} "
                  f"eps={eps} thr={thr} max_err={max_err} n={n}: status {int((st != ost).sum())}, "
                  f"positions {int((nxt != onxt).any(1).sum())}, errors {int((err != oerr).sum())}", flush=True)
        elif case % 20 == 0:
            print(f"case {case} ok ({w}x{h} win={win} levels={max_level} n={n} tracked={int(st.sum())})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    return 1 if bad else 0


sys.exit(main())
