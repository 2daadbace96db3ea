"""Tracking time per frame pair for a few tracker settings -- the configuration the bench is
quoted on (21 x 21, 3 levels, 30 iterations or 0.01) and the reference application's own defaults
(Source Files/_app.cpp:26-29: window 30, 5 levels, 250 -> 100 iterations, epsilon 0) among them
(developer tool; run on a GPU box): 16 pairs of 1280 x 720 frames, 5000 points each."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sfm_amd import capi, synth  # noqa: E402

W, H, N, PAIRS = 1280, 720, 5000, 16
dev = torch.device("cuda:0")
ctx = capi.Context(0)
prevs, currs = [], []
for k in range(PAIRS):
    p, c = synth.lk_pair(H, W, seed=42 + k)
    prevs.append(p)
    currs.append(c)
d_prev = torch.from_numpy(np.stack(prevs)).to(dev)
d_curr = torch.from_numpy(np.stack(currs)).to(dev)
pts = synth.grid_points(H, W, N, seed=42, margin=35)
d_pts = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).to(dev)
d_out = torch.empty((PAIRS, N, 2), dtype=torch.float32, device=dev)
d_status = torch.empty((PAIRS, N), dtype=torch.uint8, device=dev)
d_err = torch.empty((PAIRS, N), dtype=torch.float32, device=dev)
crit = capi.TERM_COUNT | capi.TERM_EPS
for name, win, levels, count, eps in (("bench (21, 3, 30, 0.01)", (21, 21), 3, 30, 0.01),
                                      ("15 x 15, 3 levels", (15, 15), 3, 30, 0.01),
                                      ("31 x 31, 3 levels", (31, 31), 3, 30, 0.01),
                                      ("reference defaults (30, 5, 100, 0)", (30, 30), 5, 100, 0.0)):
    def step():
        ctx.pyrlk_batch_dev(d_prev, d_curr, PAIRS, W * H, W, H, W, d_pts, N, win, levels, crit, count, eps,
                            1e-4, 10.0, d_out, d_status, d_err)
    for _ in range(3):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {1e3 * dt / PAIRS:.4f} ms per frame pair, tracked {int(d_status.sum().item())} of {PAIRS * N}",
          flush=True)
