"""Random images and parameters through sfm_good_features against the CPU oracle (developer
tool; run on a GPU box).  The corner lists must be identical."""
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
    rng = np.random.default_rng(99)
    ctx = capi.Context(0)
    bad = 0
    for case in range(n_cases):
        h = int(rng.integers(3, 300))
        w = int(rng.integers(3, 400))
        if len(argv) > 1 and argv[1] == "big":   # up to full-HD frames
            h, w = int(rng.integers(300, 1100)), int(rng.integers(400, 2000))
        kind = int(rng.integers(0, 4))
        if kind == 0 and min(h, w) >= 16:
            img = synth.to_u8(synth.texture(h, w, seed=int(rng.integers(0, 10 ** 6))))
        elif kind == 1:     # blocks of constant colour: plateaus of equal strength
            b = int(rng.integers(2, 9))
            img = np.kron(rng.integers(0, 2, ((h + b - 1) // b, (w + b - 1) // b)).astype(np.uint8) * 180 + 20,
                          np.ones((b, b), np.uint8))[:h, :w]
        elif kind == 2:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        else:
            img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
            for _ in range(int(rng.integers(0, 40))):
                img[rng.integers(0, h), rng.integers(0, w)] = rng.integers(0, 256)
        img = np.ascontiguousarray(img)
        mc = int(rng.choice([0, 1, 10, 200, 2000]))
        q = float(rng.choice([0.5, 0.1, 0.01, 0.001]))
        md = float(rng.choice([0.0, 0.5, 1.0, 2.5, 7.0, 10.0, 30.0]))
        got = ctx.good_features(img, mc, q, md)
        want = pyoracle.good_features(img, mc, q, md)
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH case {case}: {w}x{h} kind={kind} maxCorners={mc} quality={q} minDistance={md}: "
                  f"{len(got)} vs {len(want)} corners", flush=True)
        elif case % 20 == 0:
            print(f"case {case} ok ({w}x{h} kind={kind} maxCorners={mc} q={q} d={md}: {len(want)} corners)", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
