"""Times the three Hamming kernels (SFM_PAIR_KERNEL=0/1/2) on launches of different shapes.

Usage (GPU box): python tools/hamming_paths.py
Prints microseconds per sfm_match_pairs_dev call (median of 20) for n_pairs x rows x 32 B.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import capi, synth  # noqa: E402


def time_case(ctx, n_pairs, rows, d=32):
    dev = torch.device("cuda:0")
    desc = synth.descriptor_chain(n_pairs + 1, rows, d, seed=5)
    d_desc = torch.from_numpy(desc).to(dev)
    pairs = np.array([(i, i + 1) for i in range(n_pairs)], np.int32)
    d_pairs = torch.from_numpy(pairs).to(dev)
    d_m = torch.zeros((n_pairs, rows, 4), dtype=torch.int32, device=dev)
    d_c = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    out = {}
    for mode in ("0", "1", "2", ""):
        os.environ["SFM_PAIR_KERNEL"] = mode
        times = []
        for it in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, n_pairs + 1, rows * d, rows, None, d, d_pairs,
                                n_pairs, 0.5, d_m, rows, d_c)
            e1.record()
            torch.cuda.synchronize()
            if it >= 5:
                times.append(e0.elapsed_time(e1) * 1e3)
        out[mode or "auto"] = (float(np.median(times)), int(d_c.sum().item()))
    return out


def main():
    ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    for n_pairs, rows in ((1, 500), (1, 1000), (1, 2000), (1, 4096), (1, 8192), (4, 2000), (8, 1000),
                          (32, 500), (8, 8192), (32, 8192)):
        r = time_case(ctx, n_pairs, rows)
        print(f"{n_pairs:3d} x {rows:5d}: " + "  ".join(f"{k}={v[0]:8.1f}us" for k, v in r.items()) +
              f"  matches={r['auto'][1]}", flush=True)
        assert len({v[1] for v in r.values()}) == 1, r


if __name__ == "__main__":
    main()
