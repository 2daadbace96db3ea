"""CPU-only tests: the C-ABI library loads and exports what include/sfm_hip.h
declares, the product path fails loudly without a GPU, host-side logic, and the
multi-rank gather over gloo (world size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sfm_amd import capi

    lib = capi.lib()
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sfm_hip.h but not exported"
    assert sorted(capi.EXPORTS) == names
    assert b"gfx950" in lib.sfm_version()


def test_dmatch_layout_matches_cv_dmatch():
    from sfm_amd import capi

    assert capi.DMATCH_DTYPE.itemsize == 16
    assert [capi.DMATCH_DTYPE.fields[k][1] for k in ("queryIdx", "trainIdx", "imgIdx", "distance")] \
        == [0, 4, 8, 12]


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a machine without a GPU")
def test_no_gpu_fails_loudly():
    """There is no CPU fallback: creating a context without a device raises."""
    from sfm_amd import capi

    with pytest.raises(capi.SfmError) as e:
        capi.Context(0)
    assert "device" in str(e.value).lower() or "hip" in str(e.value).lower()
    import sfm_amd
    m = sfm_amd.DescriptorMatcher("BRUTEFORCE_HAMMING", 0.5)
    with pytest.raises(capi.SfmError):
        m.ratioMaches(np.zeros((4, 32), np.uint8), np.zeros((4, 32), np.uint8))


def test_package_does_not_import_the_oracle():
    """The product path must not route through oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sfm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "liboracle" not in text and "orc_" not in text, f


def test_matcher_constructor_mapping():
    import sfm_amd
    from sfm_amd import capi

    assert sfm_amd.DescriptorMatcher("bruteforce_hamming", 0.5).norm == capi.NORM_HAMMING
    assert sfm_amd.DescriptorMatcher("BRUTEFORCE_SL2", 0.5).norm == capi.NORM_L2SQR
    assert sfm_amd.DescriptorMatcher("BRUTEFORCE", 0.5).norm == capi.NORM_L2
    assert sfm_amd.DescriptorMatcher("akaze?", 0.5).norm == capi.NORM_L2
    with pytest.raises(capi.SfmError):
        sfm_amd.DescriptorMatcher("flannbased", 0.5)
    of = sfm_amd.OptFlow((3, 250, 0.0), 30, 5, 10.0, 300, 0.01, 20.0, 100)
    assert of.winSize == (30, 30) and of.additionalSettings.maxError == 10.0
    cam = sfm_amd.CameraParameters(np.diag([2000.0, 2000.0, 2.0]), None, 2.0)
    assert cam.K33d[2, 2] == 1.0 and cam.K33d[0, 0] == 1000.0


def test_window_pairs_and_sharding():
    from sfm_amd import allpairs

    p = allpairs.window_pairs(512)
    assert len(p) == 130816 and tuple(p[0]) == (0, 1) and tuple(p[-1]) == (510, 511)
    assert len(allpairs.window_pairs(10, window=2)) == 9 + 8
    for world in (1, 2, 3, 8):
        got = np.sort(np.concatenate([allpairs.shard_indices(len(p), r, world)
                                      for r in range(world)]))
        np.testing.assert_array_equal(got, np.arange(len(p)))
        sizes = [len(allpairs.shard_indices(len(p), r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_synthetic_inputs_are_seeded():
    from sfm_amd import synth

    a = synth.descriptor_chain(3, 64, 32, seed=5)
    b = synth.descriptor_chain(3, 64, 32, seed=5)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (3, 64, 32) and a.dtype == np.uint8
    pts = synth.grid_points(720, 1280, 5000)
    assert pts.shape == (5000, 2) and pts[:, 0].min() >= 15 and pts[:, 1].max() <= 720 - 16
    s = synth.triangulation_set(100)
    assert s["p1"].dtype == np.float32 and s["p1"].shape == (100, 2)


_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from sfm_amd import allpairs, synth
from oracle import pyoracle   # the checker stands in for the GPU matcher in this CPU test

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
desc = synth.descriptor_chain(7, 300, 32, seed=11)
counts = np.array([300, 280, 300, 1, 300, 255, 300])
pairs = allpairs.window_pairs(7)

def match_fn(sub):
    cs, ms = [], []
    for a, b in sub:
        m = pyoracle.robust_matches(pyoracle.NORM_HAMMING, desc[a][:counts[a]], desc[b][:counts[b]], 0.8)
        cs.append(len(m))
        ms.append(np.stack([m["queryIdx"], m["trainIdx"], m["distance"].view(np.int32)], 1).reshape(-1, 3))
    c = torch.tensor(cs, dtype=torch.int32)
    p = torch.from_numpy(np.concatenate(ms).astype(np.int32)) if ms else torch.empty((0, 3), dtype=torch.int32)
    return c, p

c, off, m = allpairs.match_all_pairs(pairs, match_fn, rank, world)
np.savez(os.path.join({out!r}, f"rank{{rank}}.npz"), c=c, off=off, m=m.numpy())
dist.barrier()
dist.destroy_process_group()
"""


def test_all_pairs_gather_gloo_world2(tmp_path, oracle):
    from sfm_amd import allpairs, synth

    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2",
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    # single-process truth
    desc = synth.descriptor_chain(7, 300, 32, seed=11)
    counts = np.array([300, 280, 300, 1, 300, 255, 300])
    pairs = allpairs.window_pairs(7)
    want_c, want_m = [], []
    for a, b in pairs:
        m = oracle.robust_matches(oracle.NORM_HAMMING, desc[a][:counts[a]], desc[b][:counts[b]], 0.8)
        want_c.append(len(m))
        want_m.append(np.stack([m["queryIdx"], m["trainIdx"], m["distance"].view(np.int32)], 1))
    want_m = np.concatenate(want_m).astype(np.int32)
    assert sum(want_c) > 300 and 0 in want_c
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_array_equal(got["c"], np.array(want_c))
        np.testing.assert_array_equal(got["off"], np.concatenate([[0], np.cumsum(want_c)]))
        np.testing.assert_array_equal(got["m"], want_m)


_PADDED_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from sfm_amd import allpairs

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
P, cap = 3, 5
g = allpairs.PaddedGather(P, cap, world, torch.device("cpu"))
for step in range(2):
    m = (torch.arange(P * cap * 4, dtype=torch.int32).reshape(P, cap, 4) + 1000 * rank + 7 * step)
    c = torch.tensor([rank + 1, 0, 5 - step], dtype=torch.int32)
    g.start(m, c)
    g.wait()
    for r in range(world):
        want = torch.arange(P * cap * 4, dtype=torch.int32).reshape(P, cap, 4) + 1000 * r + 7 * step
        assert torch.equal(g.matches[r], want)
        assert g.counts[r].tolist() == [r + 1, 0, 5 - step]
dist.barrier()
dist.destroy_process_group()
print("padded gather ok")
"""


def test_padded_gather_gloo_world2(tmp_path):
    """The sync-free all-gather of fixed-capacity match buffers the per-frame-pair bench uses."""
    script = tmp_path / "padded_worker.py"
    script.write_text(_PADDED_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29618", WORLD_SIZE="2",
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0


_SLAB_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from sfm_amd import allpairs, synth
from oracle import pyoracle   # the checker stands in for the GPU kernels in this CPU test

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
F, rows = 5, 301
desc = synth.descriptor_chain(F, rows, 32, seed=23)

# 1. the descriptor stack from per-rank shares (frame f lives on rank f % world)
per = (F + world - 1) // world
local = np.zeros((per, rows, 32), np.uint8)
for s in range(per):
    if s * world + rank < F:
        local[s] = desc[s * world + rank]
stack, nbytes = allpairs.replicate_frames(torch.from_numpy(local), F, rank, world)
assert stack.shape == (F, rows, 32) and np.array_equal(stack.numpy(), desc)
assert nbytes == (world - 1) * local.nbytes

# 2. fewer pairs than ranks: every pair split by query rows
class Slab:
    ratio = np.float32(0.8)
    def n_rows(self, f):
        return rows
    def knn_slab(self, q, t, lo, hi):
        idx, dst = pyoracle.knn2_hamming(stack[q].numpy()[lo:hi], stack[t].numpy())
        return torch.from_numpy(idx.reshape(-1, 2)), torch.from_numpy(dst.reshape(-1, 2))
    def cross(self, idx_f, dist_f, idx_b, dist_b):
        i_f, d_f, i_b, d_b = (x.numpy() for x in (idx_f, dist_f, idx_b, dist_b))
        best_f = np.where((i_f[:, 1] >= 0) & (d_f[:, 0] < self.ratio * d_f[:, 1]), i_f[:, 0], -1)
        best_b = np.where((i_b[:, 1] >= 0) & (d_b[:, 0] < self.ratio * d_b[:, 1]), i_b[:, 0], -1)
        out = [(p, c, int(d_f[p, 0])) for c, p in enumerate(best_b) if p >= 0 and best_f[p] == c]
        return len(out), torch.tensor(out, dtype=torch.int32).reshape(-1, 3)

pairs = np.array([[0, 1]] if world > 1 else [[0, 1], [1, 3]], np.int32)
def no_pair_matcher(sub):
    raise AssertionError("the pair partition must not run")
c, off, m = allpairs.match_all_pairs(pairs, no_pair_matcher, rank, world, slab=Slab())
np.savez(os.path.join({out!r}, f"slab{{rank}}.npz"), c=c, off=off, m=m.numpy())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,port", [(2, "29621"), (3, "29622")])
def test_slab_partition_and_replication_gloo(tmp_path, oracle, world, port):
    """One pair, more ranks than pairs: match_all_pairs splits it by query rows; every rank ends
    with the single-process result.  Also the descriptor stack all-gathered from per-rank shares."""
    from sfm_amd import synth

    script = tmp_path / "slab_worker.py"
    script.write_text(_SLAB_WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    desc = synth.descriptor_chain(5, 301, 32, seed=23)
    want = oracle.robust_matches(oracle.NORM_HAMMING, desc[0], desc[1], 0.8)
    want_m = np.stack([want["queryIdx"], want["trainIdx"], want["distance"].astype(np.int32)], 1)
    assert len(want) > 50
    for r in range(world):
        got = np.load(tmp_path / f"slab{r}.npz")
        np.testing.assert_array_equal(got["c"], [len(want)])
        np.testing.assert_array_equal(got["off"], [0, len(want)])
        np.testing.assert_array_equal(got["m"], want_m)


def test_bench_roofline_entries():
    """The roofline entries bench.py derives from per-kernel times: the contraction count of
    the matrix-core kernel, the lane-op count of the scalar kernels and the tracker's
    multiply-add count, each divided by the launch time it was given."""
    sys.path.insert(0, ROOT)
    import bench

    rows, d = 8192, 32
    prof = {"hamming_mfma_kernel": (8.0, 20), "lk_track_kernel": (12.0, 20)}
    r = bench.hamming_roofline(prof, 64, rows, d, "2 dirs x 32 pairs")
    assert r["kernel"] == "hamming_mfma_kernel" and r["bound"] == "mfma"
    ops = 64 * rows * rows * 256 * 2
    assert abs(r["achieved"] - ops / 400e-6 / 1e12) < 1.0
    assert abs(r["frac"] - r["achieved"] / bench.FP4_MFMA_PEAK_TOPS) < 1e-3
    lane_ops = 64 * rows * rows * 16
    assert abs(r["valu_equivalent"]["achieved"] - lane_ops / 400e-6 / 1e12) < 0.1
    r = bench.hamming_roofline({"hamming_pair_kernel": (25.0, 20)}, 64, rows, d, "x")
    assert r["bound"] == "valu" and abs(r["achieved"] - lane_ops / 1250e-6 / 1e12) < 0.01
    assert bench.hamming_roofline({}, 64, rows, d, "x")["kernel"] is None
    lk = bench.lk_roofline(prof, 32, 12.5)
    macs = 32 * bench.LK_POINTS * (4 * 6615 + 12.5 * 2646)
    assert lk["kernel"] == "lk_track_kernel" and abs(lk["achieved"] - macs / 600e-6 / 1e12) < 0.01
    assert lk["iterations_source"].startswith("cpu_baseline")
    assert bench.lk_roofline(prof, 32, None)["iterations_per_point"] == bench.LK_ITERS_PER_POINT


def test_bench_gpus_n_starts_n_ranks():
    """`bench.py --gpus 2` without a launcher starts two ranks of its own (fresh child
    processes with distinct LOCAL_RANKs) and forwards rank 0's single JSON line."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ranks"],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1] and out["local_ranks"] == [0, 1]
    assert out["distinct_processes"] == 2


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a machine without a GPU")
def test_bench_failed_rank_gives_nonzero_exit():
    """A rank that fails (no GPU here) makes the launching bench.py exit non-zero, with no
    result line on stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c5",
                        "--frames", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_adapter_opencv_branch_compiles():
    """The branch of include/sfm_hip_adapter.hpp that uses OpenCV's own types (and the Seam-2
    cv::DescriptorMatcher / cv::SparsePyrLKOpticalFlow subclasses) compiles against a
    declarations-only mirror of OpenCV's public names; every reference call site is named in
    tests/cpp/adapter_opencv_compile.cpp.  Compile only: no OpenCV exists here to link or run."""
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only",
                        "-I" + os.path.join(ROOT, "tests", "cpp", "opencv_decl"),
                        "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "adapter_opencv_compile.cpp")],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-4000:]
    # and the stand-in branch (what tests/cpp/adapter_check.cpp runs on the GPU box)
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-DSFM_ADAPTER_NO_OPENCV",
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_check.cpp")],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-4000:]


def test_hamming_fold_stays_behind_its_mfma():
    """gfx950 does not interlock a vector read of an MFMA result, and the compiler adds no wait
    states for a reader inside an asm statement: in the emitted ISA of the Hamming kernel every
    such read must lie at least the 11 wait states of an 8-pass MFMA behind its writer."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazards.py"), "--min", "11"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("hamming_mfma_kernel") >= 4, out.stdout
