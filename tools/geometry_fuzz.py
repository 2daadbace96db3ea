"""Random two-view point sets through the epipolar filter and the essential-matrix RANSAC against
the CPU oracle (developer tool; run on a GPU box).  Reports how many inlier masks are identical:
both sides draw the same samples and run the same algebra; only a few transcendental functions
come from different math libraries."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from sfm_amd import capi, synth  # noqa: E402


def main(argv=None):
    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 100
    rng = np.random.default_rng(5)
    ctx = capi.Context(0)
    same_f = same_e = 0
    strict_bad = 0      # differing masks from 15 points on (RANSAC, or an LMedS median outside the sample)
    for case in range(n_cases):
        n = int(rng.choice([8, 15, 40, 200, 1000, 5000]))
        seed = int(rng.integers(0, 10 ** 6))
        s = synth.triangulation_set(n, seed=seed)
        p1, p2 = s["p1"].copy(), s["p2"].copy()
        bad = rng.random(n) < float(rng.choice([0.0, 0.1, 0.3, 0.5]))
        p2[bad] = np.stack([rng.uniform(0, 1280, bad.sum()), rng.uniform(0, 720, bad.sum())], 1)
        p2 = (p2 + rng.normal(0, float(rng.choice([0.0, 0.3, 1.0])), p2.shape)).astype(np.float32)
        found, F, mask = ctx.fundamental(p1, p2)
        ofound, oF, omask, _ = pyoracle.find_fundamental(p1, p2)
        ok_f = found == ofound and np.array_equal(mask, omask)
        same_f += ok_f
        method = int(rng.integers(0, 2))
        thresh = float(rng.choice([0.5, 1.0, 3.0]))
        E, emask = ctx.find_essential(p1, p2, s["K"], method, 0.99, thresh)
        oE, oemask, _ = pyoracle.find_essential(p1, p2, s["K"], method, 0.99, thresh)
        ok_e = np.array_equal(emask, oemask)
        same_e += ok_e
        if not (ok_f and ok_e):
            # 8-point sets run LMedS on 7- / 5-point samples: the median of eight residuals is the
            # residual of a sample point, a rounding residue, so the winner is decided below 1e-15
            strict_bad += n >= 15
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", f"geom_case{case}.npz"), p1=p1, p2=p2, K=s["K"], F=F, oF=oF,
                     mask=mask, omask=omask, E=E, oE=oE, emask=emask, oemask=oemask, method=method, thresh=thresh)
            with np.printoptions(precision=6, linewidth=200):
                if not ok_f:
                    print("  F gpu", F.ravel(), "\n  F cpu", oF.ravel(), "\n  masks", mask, omask)
                if not ok_e:
                    print("  E gpu", E.ravel(), "\n  E cpu", oE.ravel(), "\n  masks", emask, oemask)
            print(f"case {case}: n={n} seed={seed} method={method} thresh={thresh}: fundamental mask "
                  f"{'same' if ok_f else f'differs in {int((mask != omask).sum())}'}, essential mask "
                  f"{'same' if ok_e else f'differs in {int((emask != oemask).sum())}'}", flush=True)
    print(f"fundamental: {same_f} of {n_cases} masks identical; essential: {same_e} of {n_cases}; "
          f"{strict_bad} differing cases with 15 or more points")
    ctx.close()
    return 1 if strict_bad else 0


if __name__ == "__main__":
    sys.exit(main())
