"""What the exhaustive pass costs when a few queries of the 20 000 x 20 000 x 128 set fail the
proof: n queries get three train rows within rounding of each other (developer tool; run on a
GPU box).  Prints the per-kernel event times of the call."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sfm_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
q, t0 = synth.float_descriptors(20000, 20000, dim=128, seed=99, integer=False)
for n_fail in (0, 1, 2, 8, 64, 512):
    t = t0.copy()
    for i in range(n_fail):
        t[3 * i + 100] = t[3 * i + 101] = t[3 * i + 102] = q[7 * i]
    ctx.knn2(capi.NORM_L2SQR, q, t)           # (warm: workspace, clocks)
    ctx.set_profiling(True)
    for _ in range(5):
        ctx.knn2(capi.NORM_L2SQR, q, t)
    prof = ctx.profile()
    ctx.set_profiling(False)
    print(n_fail, "planted; listed:", ctx.l2_exhaustive_count(),
          {k: round(1e3 * ms / n, 1) for k, (ms, n) in prof.items()}, flush=True)
