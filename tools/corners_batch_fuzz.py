"""Random batches through sfm_good_features_batch_dev against the CPU oracle (developer tool; run
on a GPU box): 1-9 frames per call whose content differs from frame to frame (texture, blocks,
noise, constant, a nearly constant frame with one blob), rows further apart than the width,
every combination of maxCorners, quality and minDistance.  Corner lists (order included) and
candidate counts must be identical per frame."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from sfm_amd import capi, synth  # noqa: E402


def content(rng, h, w):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return synth.to_u8(synth.texture(h, w, seed=int(rng.integers(0, 10 ** 6))))
    if kind == 1:      # blocks: plateaus of equal strength
        b = int(rng.choice([4, 8, 16]))
        coarse = rng.integers(0, 256, (h // b + 1, w // b + 1), dtype=np.uint8)
        return np.kron(coarse, np.ones((b, b), np.uint8))[:h, :w].copy()
    if kind == 2:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 3:
        return np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    img = np.full((h, w), 120, np.uint8)
    y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
    img[max(0, y - 3): y + 3, max(0, x - 3): x + 3] = 250
    return img


def main(argv=None):
    import torch

    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 60
    rng = np.random.default_rng(909)
    ctx = capi.Context(0)
    dev = torch.device("cuda:0")
    bad = 0
    for case in range(n_cases):
        h, w = int(rng.integers(8, 260)), int(rng.integers(8, 340))
        n_img = int(rng.integers(1, 10))
        stride = w + int(rng.choice([0, 3, 16]))
        imgs = np.zeros((n_img, h, stride), np.uint8)
        for i in range(n_img):
            imgs[i, :, :w] = content(rng, h, w)
        mc = int(rng.choice([0, 1, 10, 200, 2000]))
        q = float(rng.choice([0.001, 0.01, 0.1, 0.5]))
        md = float(rng.choice([0.0, 1.0, 3.5, 10.0, 30.0]))
        cap = mc if mc else h * w
        d_imgs = torch.from_numpy(imgs).to(dev)
        d_pts = torch.zeros((n_img, cap, 2), dtype=torch.float32, device=dev)
        d_info = torch.zeros((n_img, 3), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.good_features_batch_dev(d_imgs, n_img, h * stride, w, h, stride, mc, q, md, d_pts, cap, d_info,
                                    max_candidates=h * w)
        ctx.synchronize()
        info, pts = d_info.cpu().numpy(), d_pts.cpu().numpy()
        why = ""
        for i in range(n_img):
            want, ncand = pyoracle.good_features(imgs[i, :, :w], mc, q, md, return_candidates=True)
            if info[i, 0] != len(want) or info[i, 1] != ncand or info[i, 2] != 0:
                why = f"frame {i}: corners {info[i, 0]} / {len(want)}, candidates {info[i, 1]} / {ncand}, overflow {info[i, 2]}"
                break
            if not np.array_equal(pts[i, : len(want)], want):
                why = f"frame {i}: {int((pts[i, : len(want)] != want).any(1).sum())} corners differ"
                break
        if why:
            bad += 1
            print(f"MISMATCH case {case}: {w}x{h} stride={stride} frames={n_img} maxCorners={mc} q={q} minDist={md}: {why}",
                  flush=True)
        elif case % 10 == 0:
            print(f"case {case} ok ({w}x{h} frames={n_img} maxCorners={mc} q={q} minDist={md} corners={info[:, 0].tolist()})",
                  flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
