"""Random shapes through the Hamming matching entry points against the CPU oracle (developer
tool; run on a GPU box).  Every case must agree bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from sfm_amd import capi  # noqa: E402


def main(argv=None):
    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 200
    big = len(argv) > 1 and argv[1] == "big"   # row counts around and above one 8192-row train range
    rng = np.random.default_rng(77)
    ctx = capi.Context(0)
    bad = 0
    for case in range(n_cases):
        d = int(rng.choice([1, 7, 13, 16, 31, 32, 33, 61, 64]))
        n_prev = int(rng.choice([1, 2, 3, 63, 64, 65, 500, 1023, 1025, 2049, 4000]))
        n_curr = int(rng.choice([1, 2, 3, 63, 64, 65, 500, 1023, 1025, 2049, 4000]))
        if big:
            n_prev = int(rng.choice([500, 8191, 8192, 8193, 12000, 16385, 20000]))
            n_curr = int(rng.choice([500, 8191, 8192, 8193, 12000, 16385, 20000]))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            a = rng.integers(0, 256, (n_prev, d), dtype=np.uint8)
            b = rng.integers(0, 256, (n_curr, d), dtype=np.uint8)
        elif kind == 1:     # b is a noisy shuffle of a: many true matches
            a = rng.integers(0, 256, (n_prev, d), dtype=np.uint8)
            b = a[rng.integers(0, n_prev, n_curr)].copy()
            flip = rng.random((n_curr, d * 8)) < 0.03
            b ^= np.packbits(flip, axis=1, bitorder="little")
        else:               # few distinct rows: ties everywhere
            base = rng.integers(0, 256, (4, d), dtype=np.uint8)
            a = base[rng.integers(0, 4, n_prev)].copy()
            b = base[rng.integers(0, 4, n_curr)].copy()
        path = str(rng.choice(["", "0", "1", "2"]))
        os.environ["SFM_PAIR_KERNEL"] = path
        ratio = float(rng.choice([0.5, 0.75, 0.9]))
        got = ctx.robust_matches(capi.NORM_HAMMING, a, b, ratio)
        want = pyoracle.robust_matches(pyoracle.NORM_HAMMING, a, b, ratio)
        idx, dist = ctx.knn2(capi.NORM_HAMMING, a, b)
        oidx, odist = pyoracle.knn2_hamming(a, b)
        ok = got.tobytes() == want.tobytes() and np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: d={d} n_prev={n_prev} n_curr={n_curr} kind={kind} path={path!r} "
                  f"ratio={ratio}: matches {len(got)} vs {len(want)}", flush=True)
        elif case % 25 == 0:
            print(f"case {case} ok (d={d} {n_prev}x{n_curr} kind={kind} path={path!r} matches={len(want)})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
