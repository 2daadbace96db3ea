"""Times the HOST-pointer entry points (inputs copied over PCIe, results copied back, call
returns when they are in place) on the bench's per-pair workload.  Developer tool; the numbers
it prints are the PCIe-inclusive rates quoted beside the device-resident bench value."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sfm_amd import capi, synth  # noqa: E402


def med(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


ctx = capi.Context(0)
a, b = synth.descriptor_chain(2, 8192, 32, seed=1234)
t_match = med(lambda: ctx.robust_matches(capi.NORM_HAMMING, a, b, 0.5))
a2, b2 = synth.descriptor_chain(2, 2000, 32, seed=1234)
t_match2k = med(lambda: ctx.robust_matches(capi.NORM_HAMMING, a2, b2, 0.5))
prev, curr = synth.lk_pair(720, 1280, seed=42)
pts = synth.grid_points(720, 1280, 5000, seed=42, margin=15)
t_lk = med(lambda: ctx.pyrlk(prev, curr, pts))
s = synth.triangulation_set(10_000_000, seed=7)
args = (s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], s["p1"], s["p2"])
t_tri = med(lambda: ctx.triangulate(*args), reps=3, warm=1)
q, t = synth.float_descriptors(20000, 20000, dim=128, seed=99, integer=False)
t_l2 = med(lambda: ctx.knn2(capi.NORM_L2, q, t), reps=5, warm=1)
print(f"robust_matches host 8192x8192x32B: {t_match * 1e3:.3f} ms")
print(f"robust_matches host 2000x2000x32B: {t_match2k * 1e3:.3f} ms")
print(f"pyrlk host 1280x720 5000 pts: {t_lk * 1e3:.3f} ms")
print(f"frame pair host (match + lk): {(t_match + t_lk) * 1e3:.3f} ms -> "
      f"{(2 * 8192 + 5000) / (t_match + t_lk) / 1e6:.2f} M features/s")
print(f"triangulate host 10M: {t_tri * 1e3:.1f} ms -> {10 / t_tri:.1f} M correspondences/s")
print(f"knn2 L2 host 20000x20000x128: {t_l2 * 1e3:.2f} ms")
