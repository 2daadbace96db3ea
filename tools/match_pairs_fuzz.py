"""Random frame sets through the batched matcher entry points (sfm_match_pairs_dev and
sfm_match_pairs_compact_dev) against the CPU oracle (developer tool; run on a GPU box): 2-7 frames
with ragged row counts (0 and 1 included), descriptor widths and norms, random pair lists with
repeats and both orders, frame strides wider than the rows, each Hamming kernel.  Every pair's
cross-checked matches must equal the oracle's robust_matches on the same two frames bit for bit,
in both output formats."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from sfm_amd import capi  # noqa: E402


def main(argv=None):
    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 150
    rng = np.random.default_rng(9090)
    dev = torch.device("cuda:0")
    ctx = capi.Context(0)
    bad = 0
    for case in range(n_cases):
        l2 = rng.integers(0, 4) == 0
        n_frames = int(rng.integers(2, 8))
        rows = int(rng.choice([1, 2, 63, 64, 65, 129, 500, 1025, 2049] if l2 else
                              [1, 2, 63, 64, 65, 129, 500, 1025, 2049, 4100, 8192, 9000]))
        if l2:
            d = int(rng.choice([3, 32, 64, 100, 128]))
            base = rng.standard_normal((rows, d)).astype(np.float32)
            frames = []
            for _ in range(n_frames):
                fr = base + 0.1 * rng.standard_normal((rows, d)).astype(np.float32)
                fresh = rng.random(rows) < 0.3
                fr[fresh] = rng.standard_normal((int(fresh.sum()), d)).astype(np.float32)
                frames.append(fr[rng.permutation(rows)])
            norm, onorm = (capi.NORM_L2, pyoracle.NORM_L2) if rng.integers(0, 2) else (capi.NORM_L2SQR, pyoracle.NORM_L2SQR)
            elem = 4
        else:
            d = int(rng.choice([7, 16, 32, 61, 64]))
            base = rng.integers(0, 256, (rows, d), dtype=np.uint8)
            frames = []
            for _ in range(n_frames):
                fr = base.copy()
                fr ^= np.packbits(rng.random((rows, d * 8)) < 0.04, axis=1, bitorder="little")
                fresh = rng.random(rows) < 0.3
                fr[fresh] = rng.integers(0, 256, (int(fresh.sum()), d), dtype=np.uint8)
                frames.append(fr[rng.permutation(rows)])
            norm, onorm = capi.NORM_HAMMING, pyoracle.NORM_HAMMING
            elem = 1
        path = "" if l2 else str(rng.choice(["", "0", "1", "2"]))
        os.environ["SFM_PAIR_KERNEL"] = path
        stack = np.stack(frames)
        counts = np.array([int(rng.choice([0, 1, rows // 3, rows - 1, rows, rows])) for _ in range(n_frames)], np.int32)
        counts = np.clip(counts, 0, rows)
        uniform = rng.integers(0, 3) == 0          # no per-frame counts: every frame is full
        n_pairs = int(rng.integers(1, 13))
        pairs = rng.integers(0, n_frames, (n_pairs, 2)).astype(np.int32)
        pairs[pairs[:, 0] == pairs[:, 1], 1] = (pairs[pairs[:, 0] == pairs[:, 1], 0] + 1) % n_frames
        ratio = float(rng.choice([0.5, 0.75, 0.9]))
        pad = int(rng.choice([0, 0, 1, 5])) * (4 if l2 else 1)   # frames further apart than rows * d
        frame_elems = rows * d + pad
        flat = np.zeros((n_frames, frame_elems), stack.dtype)
        flat[:, : rows * d] = stack.reshape(n_frames, -1)
        d_desc = torch.from_numpy(flat).to(dev)
        d_counts = None if uniform else torch.from_numpy(counts).to(dev)
        d_pairs = torch.from_numpy(pairs).to(dev)
        cap = max(rows, 1)
        d_matches = torch.zeros((n_pairs, cap, 4), dtype=torch.int32, device=dev)
        d_mc = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
        out_cap = n_pairs * cap
        d_out = torch.zeros((out_cap, 3), dtype=torch.int16, device=dev)
        d_pc = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
        d_off = torch.zeros(n_pairs + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ctx.match_pairs_dev(norm, d_desc, n_frames, frame_elems * elem, rows, d_counts, d, d_pairs, n_pairs, ratio,
                            d_matches, cap, d_mc)
        compact = not l2
        if compact:
            ctx.match_pairs_compact_dev(norm, d_desc, n_frames, frame_elems * elem, rows, d_counts, d, d_pairs,
                                        n_pairs, ratio, d_out, out_cap, d_pc, d_off)
        ctx.synchronize()
        mc = d_mc.cpu().numpy()
        m = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(n_pairs, cap)
        pc, off = d_pc.cpu().numpy(), d_off.cpu().numpy()
        rec = d_out.cpu().numpy().view(np.uint16)
        why = ""
        for p, (i, j) in enumerate(pairs):
            ci, cj = (rows, rows) if uniform else (counts[i], counts[j])
            want = pyoracle.robust_matches(onorm, stack[i][:ci], stack[j][:cj], ratio)
            if mc[p] != len(want) or m[p][: mc[p]].tobytes() != want.tobytes():
                why = f"pair {p} ({i},{j}) counts ({ci},{cj}): {mc[p]} matches vs {len(want)}"
                break
            if compact:
                r = rec[off[p]: off[p] + pc[p]]
                if (pc[p] != len(want) or not np.array_equal(r[:, 0], want["queryIdx"]) or
                        not np.array_equal(r[:, 1], want["trainIdx"]) or
                        not np.array_equal(r[:, 2], want["distance"].astype(np.uint16))):
                    why = f"compact records of pair {p} ({i},{j}) counts ({ci},{cj})"
                    break
        if why:
            bad += 1
            print(f"MISMATCH case {case}: norm={norm} d={d} rows={rows} frames={n_frames} pairs={n_pairs} "
                  f"path={path!r} uniform={uniform} pad={pad}: {why}", flush=True)
        elif case % 10 == 0:
            print(f"case {case} ok (norm={norm} d={d} rows={rows} frames={n_frames} pairs={n_pairs} path={path!r} "
                  f"matches={int(mc.sum())})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
