import sys, numpy as np
sys.path.insert(0, ".")
from sfm_amd import capi, synth
from oracle import pyoracle
ctx = capi.Context(0)
for n, seed in ((100003, 5), (1000, 9)):
    s = synth.triangulation_set(n, seed=seed)
    args = (s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], s["p1"], s["p2"])
    X, _, mask, err = ctx.triangulate(*args)
    oX, _, omask, oerr = pyoracle.triangulate(*args)
    rel = np.linalg.norm(X - oX, axis=1) / np.linalg.norm(oX, axis=1)
    print(n, "rel max", rel.max(), "p99.9", np.quantile(rel, 0.999), "median", np.median(rel),
          "mask mismatches", int((mask != omask).sum()), "err maxdiff", np.abs(err-oerr).max())
    worst = np.argsort(rel)[-3:]
    print(" worst", rel[worst], oX[worst])
import torch, time
n = 10_000_000
s = synth.triangulation_set(n, seed=7)
dev = torch.device("cuda:0")
d1 = torch.from_numpy(s["p1"]).to(dev); d2 = torch.from_numpy(s["p2"]).to(dev)
dX = torch.empty((n,3), dtype=torch.float64, device=dev); dm = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.set_profiling(True)
for _ in range(3):
    ctx.triangulate_dev(s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], d1, d2, n, None, 0,0,0, 1e-4, 250.0, 3.0, dX, None, dm, None)
print(ctx.profile())
