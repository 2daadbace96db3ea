"""Random two-view geometries through sfm_triangulate against the CPU oracle (developer tool; run
on a GPU box).  Intrinsics, both poses, baseline, depth range, pixel noise, outlier share and
the number of correspondences vary per case.  A case agrees when every point whose system has a
clear null vector (fourth singular value below half the third) and a position within 1000 units
(the reference's distance filter is 250) lies within 1e-6 relative of
the oracle's (a hundredth of the 1e-4 bound; cases above 1e-9 are counted: the library works on
A^T A, whose rounding grows with the square of the system's condition number, i.e. for baselines
that are short against the depth), reprojection errors within 1e-5 relative (a point close to the
second camera's focal plane turns 1e-10 of position into 1e-6 of reprojection error), and masks are equal
wherever no threshold lies within 1e-6 of the tested value."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from sfm_amd import capi  # noqa: E402


def rotation(rng, max_deg):
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(-max_deg, max_deg))
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def main(argv=None):
    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 200
    only = int(argv[1]) if len(argv) > 1 else -1   # one case, with its worst points printed
    rng = np.random.default_rng(31337)
    ctx = capi.Context(0)
    bad = 0
    above = 0
    worst = 0.0
    for case in range(n_cases):
        n = int(rng.choice([1, 2, 63, 64, 65, 255, 1000, 5003, 20000]))
        f = rng.uniform(300, 3000)
        K = np.array([[f, 0, rng.uniform(200, 1000)], [0, f * rng.uniform(0.9, 1.1), rng.uniform(150, 600)],
                      [0, 0, 1]])
        R0 = rotation(rng, 20) if rng.integers(0, 2) else np.eye(3)
        t0 = rng.uniform(-2, 2, 3) if rng.integers(0, 2) else np.zeros(3)
        Rr = rotation(rng, 25)
        base = rng.standard_normal(3)
        base *= 10 ** rng.uniform(-2, 0.7) / np.linalg.norm(base)
        R1, t1 = Rr @ R0, Rr @ t0 + base
        prev_pose = np.hstack([R0, t0[:, None]])
        curr_pose = np.hstack([R1, t1[:, None]])
        zlo = 10 ** rng.uniform(-0.3, 1.5)
        zhi = zlo * 10 ** rng.uniform(0.1, 1.5)
        Xc = np.empty((n, 3))
        Xc[:, 2] = rng.uniform(zlo, zhi, n)
        Xc[:, 0] = rng.uniform(-0.4, 0.4, n) * Xc[:, 2]
        Xc[:, 1] = rng.uniform(-0.3, 0.3, n) * Xc[:, 2]
        Xw = (Xc - t0) @ R0            # camera 0 frame -> world (R0^T (Xc - t0))
        def project(R, t):
            x = (Xw @ R.T + t) @ K.T
            return x[:, :2] / x[:, 2:]
        sigma = float(rng.choice([0.0, 0.1, 0.5, 2.0]))
        p1 = project(R0, t0) + rng.normal(0, sigma, (n, 2))
        p2 = project(R1, t1) + rng.normal(0, sigma, (n, 2))
        out = rng.random(n) < float(rng.choice([0.0, 0.02, 0.3]))
        p2[out] += rng.uniform(-80, 80, (int(out.sum()), 2))
        p1, p2 = p1.astype(np.float32), p2.astype(np.float32)
        max_err = float(rng.choice([0.5, 3.0, 10.0]))
        args = (K, prev_pose, curr_pose, R1, t1, p1, p2)
        if only >= 0 and case != only:
            continue
        X, _, mask, err = ctx.triangulate(*args, max_proj_err=max_err)
        oX, _, omask, oerr = pyoracle.triangulate(*args, max_proj_err=max_err)
        # conditioning of each system from the oracle's side: singular values of A
        P1, P2 = K @ prev_pose, K @ curr_pose
        A = np.stack([p1[:, :1] * P1[2] - P1[0], p1[:, 1:] * P1[2] - P1[1],
                      p2[:, :1] * P2[2] - P2[0], p2[:, 1:] * P2[2] - P2[1]], axis=1)
        sv = np.linalg.svd(A, compute_uv=False)
        well = sv[:, 3] < 0.5 * sv[:, 2]
        # points the distance filter keeps; one that lands near infinity (w -> 0) has no digits to compare
        finite = np.isfinite(oX).all(axis=1) & well & (np.linalg.norm(oX, axis=1) < 1e3)
        rel = np.linalg.norm(X - oX, axis=1) / np.maximum(np.linalg.norm(oX, axis=1), 1e-300)
        r = float(rel[finite].max()) if finite.any() else 0.0
        # reprojection errors of exact data are rounding residue (1e-10 .. 1e-7 px): a millionth of a pixel.
        # A point a thousandth of its distance in front of (or behind) a camera's focal plane is left
        # out: there the projection turns 1e-12 of position into 1e-5 of the error, in both codes alike.
        Xh = np.concatenate([oX, np.ones((n, 1))], axis=1)
        dist = np.maximum(np.linalg.norm(oX, axis=1), 1e-300)
        steady = (np.abs(Xh @ P1[2]) > 1e-3 * dist) & (np.abs(Xh @ P2[2]) > 1e-3 * dist)
        e_ok = np.allclose(err[finite & steady], oerr[finite & steady], rtol=1e-5, atol=1e-6)
        band = np.abs(oerr - max_err) < 1e-6 * max(1.0, max_err)
        m_ok = np.array_equal(mask[finite & ~band], omask[finite & ~band])
        worst = max(worst, r)
        if case == only:
            _, _, Vt = np.linalg.svd(A)
            nX = Vt[:, 3, :3] / Vt[:, 3, 3:]
            rel_n = np.linalg.norm(X - nX, axis=1) / np.linalg.norm(nX, axis=1)
            rel_o = np.linalg.norm(oX - nX, axis=1) / np.linalg.norm(nX, axis=1)
            print("GPU vs numpy SVD: max", rel_n[finite].max(), " oracle vs numpy SVD: max", rel_o[finite].max())
            ediff = np.where(finite, np.abs(err - oerr) / np.maximum(np.abs(oerr), 1e-2), 0)
            for i in np.argsort(ediff)[-3:]:
                print(f" err point {i}: gpu {err[i]!r} oracle {oerr[i]!r} X gpu {X[i]} oracle {oX[i]} sv {sv[i]}")
            for i in np.argsort(np.where(finite, rel, 0))[-5:]:
                print(f" point {i}: sv {sv[i]}\n   gpu {X[i]}\n   oracle {oX[i]}\n   numpy {nX[i]}\n   err gpu {err[i]} oracle {oerr[i]}"
                      f" outlier {bool(out[i])} p1 {p1[i]} p2 {p2[i]}")
        above += r > 1e-9
        if r > 1e-6 or not e_ok or not m_ok:
            bad += 1
            print(f"MISMATCH case {case}: n={n} f={f:.0f} base={np.linalg.norm(base):.3f} z=[{zlo:.2f},{zhi:.2f}] "
                  f"sigma={sigma} rel={r:.3e} err_ok={e_ok} mask_ok={m_ok}", flush=True)
        elif case % 20 == 0:
            print(f"case {case} ok (n={n} base={np.linalg.norm(base):.3f} sigma={sigma} rel max {r:.2e}, "
                  f"{int(well.sum())} of {n} well conditioned)", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree; largest relative difference {worst:.3e}; "
          f"{above} cases above 1e-9")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
