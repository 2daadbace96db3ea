"""Random two-view geometries through sfm_recover_pose against the CPU oracle (developer tool; run
on a GPU box).  The essential matrix is the true one of the generated pose (scaled and, in a
third of the cases, perturbed), so the run isolates cv::recoverPose: decomposition, the four
cheirality counts and the pick.  A case agrees when the inlier count and mask are identical and
R, t agree to 1e-9."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from sfm_amd import capi  # noqa: E402
from tools.triangulate_fuzz import rotation  # noqa: E402


def main(argv=None):
    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 200
    rng = np.random.default_rng(777)
    ctx = capi.Context(0)
    bad = 0
    for case in range(n_cases):
        n = int(rng.choice([5, 6, 8, 63, 64, 65, 500, 5000]))
        f = rng.uniform(300, 3000)
        K = np.array([[f, 0, rng.uniform(200, 1000)], [0, f * rng.uniform(0.9, 1.1), rng.uniform(150, 600)],
                      [0, 0, 1]])
        R = rotation(rng, 25)
        t = rng.standard_normal(3)
        t *= 10 ** rng.uniform(-2, 0.7) / np.linalg.norm(t)
        zlo = 10 ** rng.uniform(-0.3, 1.5)
        zhi = zlo * 10 ** rng.uniform(0.1, 1.5)
        X = np.empty((n, 3))
        X[:, 2] = rng.uniform(zlo, zhi, n)
        X[:, 0] = rng.uniform(-0.4, 0.4, n) * X[:, 2]
        X[:, 1] = rng.uniform(-0.3, 0.3, n) * X[:, 2]
        sigma = float(rng.choice([0.0, 0.1, 0.5, 2.0]))
        p1 = X @ K.T
        p1 = p1[:, :2] / p1[:, 2:] + rng.normal(0, sigma, (n, 2))
        p2 = (X @ R.T + t) @ K.T
        p2 = p2[:, :2] / p2[:, 2:] + rng.normal(0, sigma, (n, 2))
        out = rng.random(n) < float(rng.choice([0.0, 0.02, 0.3]))
        p2[out] += rng.uniform(-80, 80, (int(out.sum()), 2))
        p1, p2 = p1.astype(np.float32), p2.astype(np.float32)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = tx @ R * 10 ** rng.uniform(-2, 2) * rng.choice([-1.0, 1.0])
        if rng.integers(0, 3) == 0:
            E = E + rng.standard_normal((3, 3)) * 1e-3 * np.abs(E).max()
        mask_in = None
        if rng.integers(0, 2):
            mask_in = (rng.random(n) < 0.8).astype(np.uint8)
        cnt, Rg, tg, mg = ctx.recover_pose(E, p1, p2, K, mask=mask_in)
        ocnt, Ro, to, mo = pyoracle.recover_pose(E, p1, p2, K, mask=mask_in)
        same = (int(cnt) == int(ocnt) and np.array_equal(np.asarray(mg).ravel(), np.asarray(mo).ravel())
                and np.allclose(Rg, Ro, atol=1e-9) and np.allclose(np.ravel(tg), np.ravel(to), atol=1e-9))
        # No point in front of both cameras for any of the four candidates: the pick falls on the
        # decomposition's first candidate, whose translation sign is the sign the 3x3 SVD happened
        # to give.  Zero inliers either way, R equal, t equal up to sign.
        tie = (int(cnt) == 0 and int(ocnt) == 0 and np.allclose(Rg, Ro, atol=1e-9) and
               np.allclose(np.abs(np.ravel(tg)), np.abs(np.ravel(to)), atol=1e-9))
        if not same and not tie:
            bad += 1
            print(f"MISMATCH case {case}: n={n} sigma={sigma} |t|={np.linalg.norm(t):.3f} z=[{zlo:.2f},{zhi:.2f}] "
                  f"count {int(cnt)} / {int(ocnt)}, mask differs in "
                  f"{int((np.asarray(mg).ravel() != np.asarray(mo).ravel()).sum())}, "
                  f"dR {np.abs(Rg - Ro).max():.2e} dt {np.abs(np.ravel(tg) - np.ravel(to)).max():.2e}", flush=True)
        elif case % 20 == 0:
            print(f"case {case} ok (n={n} sigma={sigma} inliers {int(cnt)})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
