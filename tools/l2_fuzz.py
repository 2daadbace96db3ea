"""Random shapes through the L2 2-NN entry points against the CPU oracle (developer tool; run on
a GPU box).  Every case must agree bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from sfm_amd import capi  # noqa: E402


def main(argv=None):
    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 150
    big = len(argv) > 1 and argv[1] == "big"   # train sets around and above 8192 rows
    rng = np.random.default_rng(2026)
    ctx = capi.Context(0)
    bad = 0
    for case in range(n_cases):
        dim = int(rng.choice([1, 3, 16, 32, 61, 64, 65, 100, 128, 129, 200, 256]))
        nq = int(rng.choice([1, 2, 31, 33, 64, 255, 257, 700, 1500]))
        nt = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 129, 500, 1250, 3000]))
        if big:
            nq = int(rng.choice([500, 4097, 9000]))
            nt = int(rng.choice([8191, 8192, 8193, 12000, 16385, 20000]))
            dim = int(rng.choice([32, 64, 128]))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            q = rng.standard_normal((nq, dim)).astype(np.float32)
            t = rng.standard_normal((nt, dim)).astype(np.float32)
        elif kind == 1:     # integers with many exact ties
            q = rng.integers(0, 4, (nq, dim)).astype(np.float32)
            t = rng.integers(0, 4, (nt, dim)).astype(np.float32)
        elif kind == 2:     # offset cluster
            q = (50 + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
            t = (50 + 0.1 * rng.standard_normal((nt, dim))).astype(np.float32)
        elif kind == 3:     # duplicates of a few rows
            base = rng.standard_normal((5, dim)).astype(np.float32)
            q = base[rng.integers(0, 5, nq)].copy()
            t = base[rng.integers(0, 5, nt)].copy()
        else:               # wide magnitudes
            q = (rng.standard_normal((nq, dim)) * 10.0 ** rng.uniform(-6, 6, (nq, 1))).astype(np.float32)
            t = (rng.standard_normal((nt, dim)) * 10.0 ** rng.uniform(-6, 6, (nt, 1))).astype(np.float32)
        strided = bool(rng.integers(0, 2)) and dim > 1
        if strided:         # rows further apart than their width
            qq = np.zeros((nq, dim + 3), np.float32)
            qq[:, :dim] = q
            q = qq[:, :dim]
        squared = bool(rng.integers(0, 2))
        mode = str(rng.choice(["f16", "f16", "bf16", "f32"]))
        os.environ["SFM_L2_MFMA"] = mode
        norm = capi.NORM_L2SQR if squared else capi.NORM_L2
        idx, dist = ctx.knn2(norm, q, t)
        oidx, odist = pyoracle.knn2_l2(np.ascontiguousarray(q), t, squared=squared)
        ok = np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: dim={dim} nq={nq} nt={nt} kind={kind} strided={strided} "
                  f"squared={squared} mode={mode}: {int((idx != oidx).sum())} indices, "
                  f"{int((dist != odist).sum())} distances differ", flush=True)
        elif case % 25 == 0:
            print(f"case {case} ok (dim={dim} nq={nq} nt={nt} kind={kind} mode={mode} "
                  f"exhaustive={ctx.l2_exhaustive_count()})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
