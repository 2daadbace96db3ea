"""The C++ adapter (include/sfm_hip_adapter.hpp) driven the way the reference's main loop
drives DescriptorMatcher / OptFlow / Reconstruction (Source Files/app_solver.cpp:105-118,
:182, :404, :435), checked against the CPU oracle.  The program is tests/cpp/adapter_check,
built by `make -C sfm_amd/csrc`; it is compiled without OpenCV, against the adapter's stand-in
types."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sfm_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "tests", "cpp", "adapter_check")


def run(mode, payload, tmp_path):
    assert os.path.exists(CHECK), "tests/cpp/adapter_check missing: run make -C sfm_amd/csrc"
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(payload)
    subprocess.run([CHECK, mode, str(fin), str(fout)], check=True, timeout=120)
    return fout.read_bytes()


class Cursor:
    def __init__(self, buf):
        self.buf, self.pos = buf, 0

    def i32(self):
        v = struct.unpack_from("<i", self.buf, self.pos)[0]
        self.pos += 4
        return v

    def arr(self, dtype, count):
        a = np.frombuffer(self.buf, dtype=dtype, count=count, offset=self.pos)
        self.pos += a.nbytes
        return a


DMATCH = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


@pytest.mark.parametrize("norm,use_filter", [(0, 0), (0, 1), (1, 0), (2, 0)])
def test_find_robust_matches(oracle, tmp_path, norm, use_filter):
    if norm == 0:
        prev, curr = synth.descriptor_chain(2, 1500, 32, seed=5)
        curr = curr[:1300]
    else:
        prev, curr = synth.float_descriptors(700, 650, dim=64, seed=17, integer=True)
    ratio = 0.6 if norm == 0 else 0.8
    payload = struct.pack("<iiiifi", norm, len(prev), len(curr), prev.shape[1], ratio, use_filter)
    out = Cursor(run("match", payload + prev.tobytes() + curr.tobytes(), tmp_path))

    if norm == 0:
        idx, dist = oracle.knn2_hamming(prev, curr)
    else:
        idx, dist = oracle.knn2_l2(prev, curr, squared=(norm == 2))
    want_fwd = oracle.ratio_filter(idx, dist, len(curr), ratio)
    want = oracle.robust_matches(norm, prev, curr, ratio)
    if use_filter:  # the caller's filter of the check program drops matches 0, 3, 6, ...
        want = want[np.arange(len(want)) % 3 != 0]

    n = out.i32()
    fwd = out.arr(DMATCH, n)
    assert fwd.tobytes() == want_fwd.tobytes()
    n = out.i32()
    matches = out.arr(DMATCH, n)
    assert len(want) > 50 and matches.tobytes() == want.tobytes()
    n = out.i32()
    assert n == len(want)
    prev_pts = out.arr("<f4", 2 * n).reshape(n, 2)
    curr_pts = out.arr("<f4", 2 * n).reshape(n, 2)
    q, t = want["queryIdx"].astype(np.float32), want["trainIdx"].astype(np.float32)
    assert np.array_equal(prev_pts, np.stack([q + 0.25, 2 * q], 1))
    assert np.array_equal(curr_pts, np.stack([3 * t, t + 0.5], 1))
    n = out.i32()
    assert n == len(want)
    assert np.array_equal(out.arr("<i4", n), want["queryIdx"])
    assert np.array_equal(out.arr("<i4", n), want["trainIdx"])


def test_compute_flow(oracle, tmp_path):
    prev, curr = synth.lk_pair(200, 264, seed=3)
    pts = synth.grid_points(200, 264, 300, seed=3, margin=12)
    win, max_level, max_error = 15, 2, 4.0
    payload = struct.pack("<iiiiiiidf", 264, 200, len(pts), win, max_level, 3, 20, 0.02, max_error)
    out = Cursor(run("flow", payload + prev.tobytes() + curr.tobytes() + pts.tobytes(), tmp_path))
    n = out.i32()
    assert n == len(pts)
    nxt = out.arr("<f4", 2 * n).reshape(n, 2)
    status = out.arr("u1", n)
    assert out.i32() == max_level and out.i32() == win
    onxt, ost, _ = oracle.pyrlk(prev, curr, pts, win=(win, win), max_level=max_level, crit_type=3,
                                max_count=20, epsilon=0.02, max_error=max_error, acc_exact=True)
    assert np.array_equal(status, ost) and status.sum() > 0.8 * n
    assert np.array_equal(nxt, onxt)


def test_triangulate_cloud(oracle, tmp_path):
    s = synth.triangulation_set(3000, seed=11)
    h, w = 720, 1280
    rng = np.random.default_rng(2)
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    payload = struct.pack("<iii", len(s["p1"]), w, h)
    for key in ("K", "prev_pose", "curr_pose", "R", "t"):
        payload += np.ascontiguousarray(s[key], dtype="<f8").tobytes()
    payload += struct.pack("<fff", 1e-4, 250.0, 3.0)
    payload += s["p1"].tobytes() + s["p2"].tobytes() + bgr.tobytes()
    out = Cursor(run("tri", payload, tmp_path))
    n = out.i32()
    assert n == len(s["p1"])
    X = out.arr("<f8", 3 * n).reshape(n, 3)
    rgb = out.arr("u1", 3 * n).reshape(n, 3)
    assert out.i32() == n + 1  # the mask list is appended to
    mask = out.arr("u1", n + 1)
    oX, orgb, omask, oerr = oracle.triangulate(s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"],
                                               s["p1"], s["p2"], bgr=bgr)
    rel = np.linalg.norm(X - oX, axis=1) / np.linalg.norm(oX, axis=1)
    assert rel.max() < 1e-9  # fp64 on both sides; the contract is 1e-4 relative
    assert np.array_equal(rgb, orgb)
    assert mask[0] == 1
    clear = np.abs(oerr - 3.0) > 1e-6
    assert np.array_equal(mask[1:][clear], omask[clear])


def test_misuse_raises():
    assert os.path.exists(CHECK)
    res = subprocess.run([CHECK, "errors"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr


def test_find_robust_matches_default_epipolar_filter(oracle, tmp_path):
    """useEpipolarFilter = true with the adapter's default filter (sfm_fundamental): the
    matches kept are those the oracle's findFundamentalMat marks, in order."""
    n = 1400
    s = synth.triangulation_set(n, seed=61)
    rng = np.random.default_rng(62)
    prev = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    perm = rng.permutation(n)
    curr = prev[perm].copy()
    curr ^= np.packbits(rng.random(size=(n, 256)) < 0.04, axis=1, bitorder="little")
    kp0, kp1 = s["p1"].astype(np.float32), s["p2"][perm].astype(np.float32)
    wrong = rng.random(n) < 0.2
    kp1[wrong] = np.stack([rng.uniform(0, 1280, wrong.sum()), rng.uniform(0, 720, wrong.sum())], 1).astype(np.float32)
    payload = struct.pack("<iiiifi", 0, n, n, 32, 0.8, 2)
    out = Cursor(run("match", payload + prev.tobytes() + curr.tobytes() + kp0.tobytes() + kp1.tobytes(), tmp_path))

    want = oracle.robust_matches(0, prev, curr, 0.8)
    _, _, mask, _ = oracle.find_fundamental(kp0[want["queryIdx"]], kp1[want["trainIdx"]])
    want = want[mask.astype(bool)]
    out.arr(DMATCH, out.i32())  # forward ratio matches, covered above
    matches = out.arr(DMATCH, out.i32())
    assert 300 < len(want) and matches.tobytes() == want.tobytes()
    k = out.i32()
    assert k == len(want)
    assert np.array_equal(out.arr("<f4", 2 * k).reshape(k, 2), kp0[want["queryIdx"]])
    assert np.array_equal(out.arr("<f4", 2 * k).reshape(k, 2), kp1[want["trainIdx"]])
    assert out.i32() == k
    assert np.array_equal(out.arr("<i4", k), want["queryIdx"])
    assert np.array_equal(out.arr("<i4", k), want["trainIdx"])


def test_generate_flow_features(oracle, tmp_path):
    """sfm_hip::FeatureDetector::generateFlowFeatures appends what goodFeaturesToTrack finds."""
    img = synth.to_u8(synth.texture(240, 320, seed=9))
    payload = struct.pack("<iiidd", 320, 240, 150, 0.01, 8.0)
    out = Cursor(run("corners", payload + img.tobytes(), tmp_path))
    n = out.i32()
    pts = out.arr("<f4", 2 * n).reshape(n, 2)
    want = oracle.good_features(img, 150, 0.01, 8.0)
    assert len(want) == 150
    assert np.array_equal(pts[0], [1.5, 2.5]) and np.array_equal(pts[1:], want)


def test_find_camera_pose(oracle, tmp_path):
    """sfm_hip::Tracking::findCameraPose against the same steps done with the oracle."""
    n = 1800
    s = synth.triangulation_set(n, seed=71, sigma=0.2, outliers=0.0)
    rng = np.random.default_rng(72)
    p1, p2 = s["p1"], s["p2"].copy()
    bad = rng.random(n) < 0.25
    p2[bad] = np.stack([rng.uniform(0, 1280, bad.sum()), rng.uniform(0, 720, bad.sum())], 1).astype(np.float32)
    K = np.ascontiguousarray(s["K"], dtype=np.float64)
    payload = struct.pack("<iidd", n, 10, 0.99, 0.5) + K.tobytes() + p1.tobytes() + p2.tobytes()
    out = Cursor(run("pose", payload, tmp_path))
    ok, n_in = out.i32(), out.i32()
    R = out.arr("<f8", 9).reshape(3, 3)
    t = out.arr("<f8", 3)
    rows = out.i32()
    mask = out.arr("u1", rows)
    E, omask, _ = oracle.find_essential(p1, p2, K, 0, 0.99, 0.5)
    assert len(E) == 1 and rows == n
    if oracle.essential_residual(E[0], p1[0], p2[0], K) > 1e-3:
        assert ok == 0 and np.array_equal(mask, omask)
        return
    cnt, oR, ot, omask2 = oracle.recover_pose(E[0], p1, p2, K, omask)
    assert ok == int(cnt > 10) and n_in == cnt
    assert np.array_equal(mask, omask2)
    np.testing.assert_allclose(R, oR, atol=1e-8)
    np.testing.assert_allclose(t, ot, atol=1e-8)
