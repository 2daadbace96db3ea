import sys, os, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np
if len(sys.argv) > 1:
    from sfm_amd import capi, synth
    from oracle import pyoracle
    n = 100003
    s = synth.triangulation_set(n, seed=7 + n)
    args = (s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], s["p1"], s["p2"])
    ctx = capi.Context(0)
    X, _, mask, err = ctx.triangulate(*args)
    oX, _, omask, oerr = pyoracle.triangulate(*args)
    rel = np.linalg.norm(X - oX, axis=1) / np.maximum(np.linalg.norm(oX, axis=1), 1e-300)
    P1, P2 = s["K"] @ s["prev_pose"], s["K"] @ s["curr_pose"]
    p1, p2 = s["p1"].astype(np.float64), s["p2"].astype(np.float64)
    A = np.stack([p1[:, :1] * P1[2] - P1[0], p1[:, 1:] * P1[2] - P1[1], p2[:, :1] * P2[2] - P2[0], p2[:, 1:] * P2[2] - P2[1]], axis=1)
    idx = np.argsort(rel)[-6:]
    sv = np.linalg.svd(A[idx], compute_uv=False)
    print(sys.argv[1], "max", rel.max(), "count>1e-9", int((rel > 1e-9).sum()))
    for k, i in enumerate(idx):
        print("  pt", i, "rel", rel[i], "sv", sv[k], "ratio43", sv[k][3] / sv[k][2], "X", oX[i], "mask", mask[i], omask[i])
else:
    for env in ({}, {"SFM_TRIANGULATE_NORMAL_EQ": "1"}, {"SFM_TRIANGULATE_JACOBI": "1"}):
        subprocess.run([sys.executable, __file__, json.dumps(env)], env=dict(os.environ, **env))
