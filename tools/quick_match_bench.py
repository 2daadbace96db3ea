"""Times sfm_match_pairs_dev on a chained synthetic window (developer tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sfm_amd import capi

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
d = 32
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(1)
desc = torch.randint(0, 256, (n_frames, rows, d), dtype=torch.uint8, generator=g).to(dev)
pairs = np.array([(i, j) for i in range(n_frames) for j in range(i + 1, n_frames)], np.int32)
d_pairs = torch.from_numpy(pairs).to(dev)
P = len(pairs)
cap = rows
d_m = torch.zeros((P, cap, 4), dtype=torch.int32, device=dev)
d_c = torch.zeros(P, dtype=torch.int32, device=dev)
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
def run():
    ctx.match_pairs_dev(0, desc, n_frames, rows * d, rows, None, d, d_pairs, P, 0.8, d_m, cap, d_c)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    ops = 2.0 * P * rows * rows * 16
    print(f"pairs={P} rows={rows} {ms:.3f} ms  {ms/P*1000:.1f} us/pair  {ops/ms/1e9:.2f} T lane-ops/s "
          f"({ops/ms/1e9/78.6*100:.1f}% of 78.6T)")
