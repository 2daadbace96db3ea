"""How many queries each L2 candidate pass sends to the exhaustive pass, per data shape
(developer tool; run on a GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth  # noqa: E402


def sets():
    rng = np.random.default_rng(3)
    n, d = 8000, 128
    q, t = synth.float_descriptors(n, n, dim=d, seed=1, integer=False)
    yield "unit-norm chain", q, t
    q, t = synth.float_descriptors(n, n, dim=d, seed=2, integer=True)
    yield "integer chain", q, t
    g = rng.standard_normal((n, d)).astype(np.float32)
    yield "gaussian", g[: n // 2].copy(), g[n // 2:].copy()
    centers = rng.standard_normal((40, d)).astype(np.float32)
    c = centers[rng.integers(0, 40, n)] + 0.05 * rng.standard_normal((n, d)).astype(np.float32)
    yield "40 tight clusters", c[: n // 2].copy(), c[n // 2:].copy()
    c = centers[rng.integers(0, 40, n)] + 0.002 * rng.standard_normal((n, d)).astype(np.float32)
    yield "40 very tight clusters", c[: n // 2].copy(), c[n // 2:].copy()
    u = np.abs(rng.standard_normal((n, d))).astype(np.float32) ** 3
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    yield "heavy-tailed unit", u[: n // 2].copy(), u[n // 2:].copy()
    s = np.floor(np.clip(512 * u / 0.2, 0, 255)).astype(np.float32)
    yield "sift-like bytes", s[: n // 2].copy(), s[n // 2:].copy()
    m = rng.standard_normal((n, d)).astype(np.float32) * (10.0 ** rng.uniform(-3, 3, (n, 1))).astype(np.float32)
    yield "rows of mixed magnitude", m[: n // 2].copy(), m[n // 2:].copy()


def main():
    ctx = capi.Context(0)
    for name, q, t in sets():
        line = f"{name:26s} nq={len(q)} nt={len(t)}:"
        for mode in ("f32", "bf16", "f16"):
            os.environ["SFM_L2_MFMA"] = mode
            ctx.knn2(capi.NORM_L2SQR, q, t)
            t0 = time.perf_counter()
            ctx.knn2(capi.NORM_L2SQR, q, t)
            dt = time.perf_counter() - t0
            line += f"  {mode}: {ctx.l2_exhaustive_count():5d} ({1e3 * dt:.1f} ms)"
        print(line, flush=True)


main()
