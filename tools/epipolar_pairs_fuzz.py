"""Random batches through sfm_epipolar_filter_pairs_dev against the CPU oracle (developer tool;
run on a GPU box): 1-7 frame pairs per call, every pair with its own match list (0 to several
thousand matches, so that "none", the direct 7-point solve, LMedS and RANSAC all occur in one
batch), key point records of 8 or 28 bytes, random outlier shares and thresholds.  Kept matches,
counts and masks must equal the reference's order of operations done with the oracle
(Source Files/feature_processing.cpp:187-204)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from sfm_amd import capi, synth  # noqa: E402


def main(argv=None):
    import torch

    argv = sys.argv[1:] if argv is None else [str(a) for a in argv]
    n_cases = int(argv[0]) if len(argv) > 0 else 60
    rng = np.random.default_rng(2024)
    ctx = capi.Context(0)
    dev = torch.device("cuda:0")
    bad = 0
    for case in range(n_cases):
        n_frames = int(rng.integers(2, 5))
        n_kp = int(rng.choice([40, 300, 2500]))
        rec = int(rng.choice([8, 28]))                     # plain points or cv::KeyPoint records
        # every frame is its own view of one set of 3D points (a frame matched against itself or
        # against a copy shifted in the image plane has no unique epipolar geometry: which model
        # wins is then a matter of rounding)
        s = synth.triangulation_set(n_kp, seed=int(rng.integers(0, 10 ** 6)))
        sigma = float(rng.choice([0.0, 0.3, 1.0]))
        kps = np.zeros((n_frames, n_kp, rec // 4), np.float32)
        for f in range(n_frames):
            R = synth.rot_y(float(rng.uniform(-6, 6)))
            t = np.array([f * 0.8 + rng.uniform(0.3, 0.6), rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3)])
            x = (s["X"] @ R.T + t) @ s["K"].T
            kps[f, :, :2] = x[:, :2] / x[:, 2:] + rng.normal(0, sigma, (n_kp, 2))
        n_pairs = int(rng.integers(1, 8))
        a_of = rng.integers(0, n_frames, n_pairs)
        b_of = (a_of + rng.integers(1, n_frames, n_pairs)) % n_frames      # never the frame itself
        pairs = np.stack([a_of, b_of], 1).astype(np.int32)
        cap = n_kp
        counts = np.array([int(rng.choice([0, 3, 6, 7, 8, 11, 14, 15, 16, 40, n_kp // 2, n_kp])) for _ in range(n_pairs)],
                          np.int32)
        counts = np.minimum(counts, cap)
        matches = np.zeros((n_pairs, cap), capi.DMATCH_DTYPE)
        outl = float(rng.choice([0.0, 0.1, 0.4]))
        for p in range(n_pairs):
            c = counts[p]
            q = rng.permutation(n_kp)[:c]
            t = q.copy()
            wrong = rng.random(c) < outl                      # mismatched correspondences
            t[wrong] = rng.integers(0, n_kp, int(wrong.sum()))
            matches[p, :c]["queryIdx"] = q
            matches[p, :c]["trainIdx"] = t
            matches[p, :c]["distance"] = rng.integers(0, 64, c)
        thresh = float(rng.choice([3.0, 1.0, 0.5]))
        d_kp = torch.from_numpy(kps).to(dev)
        d_pairs = torch.from_numpy(pairs).to(dev)
        d_matches = torch.from_numpy(matches.view(np.int32).reshape(n_pairs, cap, 4)).to(dev)
        d_mc = torch.from_numpy(counts).to(dev)
        d_mask = torch.zeros((n_pairs, cap), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.epipolar_filter_pairs_dev(d_kp, n_kp * rec, rec, d_pairs, n_pairs, d_matches, cap, d_mc, thresh=thresh,
                                      d_mask=d_mask)
        ctx.synchronize()
        mc = d_mc.cpu().numpy()
        got = d_matches.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(n_pairs, cap)
        gmask = d_mask.cpu().numpy()
        why = ""
        for p, (a, b) in enumerate(pairs):
            c = counts[p]
            m = matches[p, :c]
            p1 = kps[a][m["queryIdx"], :2].copy()
            p2 = kps[b][m["trainIdx"], :2].copy()
            if 8 <= c <= 13:
                # Least median of squares on 7-point samples: the median of 8..13 residuals is the
                # residual of a sample point, a rounding residue, so which model wins (and which
                # points pass a threshold derived from it) is decided below 1e-15 by either
                # implementation, as by OpenCV itself.  Checked here: the kept list is the list
                # the call's own mask selects, in order.
                own = m[gmask[p, :c].astype(bool)]
                if mc[p] != len(own) or got[p][: mc[p]].tobytes() != own.tobytes():
                    why = f"pair {p} ({a},{b}) of {c} matches: kept list and mask disagree"
                    break
                continue
            if c >= 7:
                _, _, omask, _ = pyoracle.find_fundamental(p1, p2, thresh=thresh)
            else:
                omask = np.zeros(c, np.uint8)
            want = m[omask.astype(bool)]
            if mc[p] != len(want) or got[p][: mc[p]].tobytes() != want.tobytes():
                why = f"pair {p} ({a},{b}) of {c} matches: kept {mc[p]} / {len(want)}"
                break
            if not np.array_equal(gmask[p, :c], omask):
                why = f"pair {p}: mask differs in {int((gmask[p, :c] != omask).sum())}"
                break
        if why:
            bad += 1
            print(f"MISMATCH case {case}: frames={n_frames} kp={n_kp} rec={rec} pairs={n_pairs} counts={counts.tolist()} "
                  f"thresh={thresh}: {why}", flush=True)
        elif case % 10 == 0:
            print(f"case {case} ok (kp={n_kp} pairs={n_pairs} counts={counts.tolist()} kept={mc.tolist()})", flush=True)
    print(f"{n_cases - bad} of {n_cases} cases agree")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
