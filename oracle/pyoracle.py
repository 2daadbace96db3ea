"""ctypes binding of oracle/liboracle.so, the CPU restatement of the hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg; never by the sfm_amd package.  Parity with OpenCV
itself is unpinned (see oracle/oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

NORM_HAMMING, NORM_L2, NORM_L2SQR = 0, 1, 2

DMATCH_DTYPE = np.dtype(
    [("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]
)


def build():
    """Compile liboracle.so with the recipe in oracle/Makefile."""
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_ratio_filter.restype = C.c_int
        _lib.orc_cross_check.restype = C.c_int
        _lib.orc_robust_matches.restype = C.c_int
        _lib.orc_lk_num_levels.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def knn2_hamming(q, t, threads=0):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    nq, d = q.shape
    nt = t.shape[0]
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.float32)
    lib().orc_knn2_hamming(_p(q), C.c_int(nq), C.c_size_t(d), _p(t), C.c_int(nt), C.c_size_t(d),
                           C.c_int(d), _p(idx), _p(dist), C.c_int(threads))
    return idx, dist


def knn2_l2(q, t, squared=False, threads=0):
    q = np.ascontiguousarray(q, dtype=np.float32)
    t = np.ascontiguousarray(t, dtype=np.float32)
    nq, d = q.shape
    nt = t.shape[0]
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.float32)
    lib().orc_knn2_l2(_p(q), C.c_int(nq), C.c_size_t(d), _p(t), C.c_int(nt), C.c_size_t(d),
                      C.c_int(d), C.c_int(1 if squared else 0), _p(idx), _p(dist),
                      C.c_int(threads))
    return idx, dist


def ratio_filter(idx, dist, nt, ratio):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    nq = idx.shape[0]
    out = np.zeros(max(nq, 1), DMATCH_DTYPE)
    n = lib().orc_ratio_filter(_p(idx), _p(dist), C.c_int(nq), C.c_int(nt), C.c_float(ratio),
                               _p(out))
    return out[:n].copy()


def cross_check(f, b):
    f = np.ascontiguousarray(f, dtype=DMATCH_DTYPE)
    b = np.ascontiguousarray(b, dtype=DMATCH_DTYPE)
    out = np.zeros(max(len(b), 1), DMATCH_DTYPE)
    n = lib().orc_cross_check(_p(f), C.c_int(len(f)), _p(b), C.c_int(len(b)), _p(out))
    return out[:n].copy()


def robust_matches(norm, prev, curr, ratio, threads=0):
    dt = np.uint8 if norm == NORM_HAMMING else np.float32
    prev = np.ascontiguousarray(prev, dtype=dt)
    curr = np.ascontiguousarray(curr, dtype=dt)
    d = prev.shape[1] if prev.ndim == 2 else curr.shape[1]
    out = np.zeros(max(len(curr), 1), DMATCH_DTYPE)
    n = lib().orc_robust_matches(C.c_int(norm), _p(prev), C.c_int(len(prev)), _p(curr),
                                 C.c_int(len(curr)), C.c_int(d), C.c_float(ratio), _p(out),
                                 C.c_int(threads))
    return out[:n].copy()


def pyr_down(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down(_p(img), C.c_int(w), C.c_int(h), C.c_size_t(img.strides[0]), _p(out),
                       C.c_size_t(out.strides[0]))
    return out


def scharr_deriv(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().orc_scharr_deriv(_p(img), C.c_int(w), C.c_int(h), C.c_size_t(img.strides[0]), _p(out),
                           C.c_size_t(2 * w))
    return out


def lk_num_levels(w, h, win_w, win_h, max_level):
    return lib().orc_lk_num_levels(C.c_int(w), C.c_int(h), C.c_int(win_w), C.c_int(win_h),
                                   C.c_int(max_level))


def pyrlk(prev, curr, prev_pts, win=(21, 21), max_level=3, crit_type=3, max_count=30,
          epsilon=0.01, min_eig_threshold=1e-4, max_error=10.0, acc_exact=False, threads=0,
          return_iters=False, return_level_iters=False):
    """return_level_iters: also returns an int32 array [levels + 1][n] with the iterations every
    point ran on every pyramid level (0 where a level was skipped)."""
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    curr = np.ascontiguousarray(curr, dtype=np.uint8)
    assert prev.shape == curr.shape and prev.ndim == 2
    h, w = prev.shape
    pts = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
    n = pts.shape[0]
    out = np.zeros((n, 2), np.float32)
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    iters = C.c_int64(0)
    level_iters = None
    if return_level_iters:
        levels = lk_num_levels(w, h, win[0], win[1], max_level)
        level_iters = np.zeros((levels + 1, max(n, 1)), np.int32)
        lib().orc_lk_set_iter_buffer(_p(level_iters))
    try:
        _orc_pyrlk(prev, curr, w, h, pts, n, win, max_level, crit_type, max_count, epsilon, min_eig_threshold,
                   max_error, out, status, err, acc_exact, threads, iters)
    finally:
        if return_level_iters:
            lib().orc_lk_set_iter_buffer(None)
    res = (out, status, err)
    if return_iters:
        res += (iters.value,)
    if return_level_iters:
        res += (level_iters[:, :n],)
    return res


def _orc_pyrlk(prev, curr, w, h, pts, n, win, max_level, crit_type, max_count, epsilon, min_eig_threshold,
               max_error, out, status, err, acc_exact, threads, iters):
    lib().orc_pyrlk(_p(prev), _p(curr), C.c_int(w), C.c_int(h), C.c_size_t(prev.strides[0]),
                    _p(pts), C.c_int(n), C.c_int(win[0]), C.c_int(win[1]), C.c_int(max_level),
                    C.c_int(crit_type), C.c_int(max_count), C.c_double(epsilon),
                    C.c_double(min_eig_threshold), C.c_float(max_error), _p(out), _p(status),
                    _p(err), C.c_int(1 if acc_exact else 0), C.c_int(threads), C.byref(iters))


def triangulate(K, prev_pose, curr_pose, R, t, prev_pts, curr_pts, bgr=None, min_dist=1e-4,
                max_dist=250.0, max_proj_err=3.0, threads=0):
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
    prev_pose = np.ascontiguousarray(prev_pose, dtype=np.float64).reshape(3, 4)
    curr_pose = np.ascontiguousarray(curr_pose, dtype=np.float64).reshape(3, 4)
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(3, 3)
    t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
    p1 = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(curr_pts, dtype=np.float32).reshape(-1, 2)
    n = p1.shape[0]
    X = np.zeros((n, 3), np.float64)
    rgb = np.zeros((n, 3), np.uint8)
    mask = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float64)
    if bgr is not None:
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        ih, iw = bgr.shape[:2]
        bp, bs = _p(bgr), bgr.strides[0]
    else:
        ih = iw = 0
        bp, bs = None, 0
    lib().orc_triangulate(_p(K), _p(prev_pose), _p(curr_pose), _p(R), _p(t), _p(p1), _p(p2),
                          C.c_int(n), bp, C.c_int(iw), C.c_int(ih), C.c_size_t(bs),
                          C.c_float(min_dist), C.c_float(max_dist), C.c_float(max_proj_err),
                          _p(X), _p(rgb), _p(mask), _p(err), C.c_int(threads))
    return X, rgb, mask, err


def find_fundamental(p1, p2, thresh=3.0, confidence=0.99, max_iters=1000):
    """Returns (found, F [3,3], mask [n] uint8, iterations run)."""
    p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(-1, 2)
    n = p1.shape[0]
    F = np.zeros(9)
    mask = np.zeros(max(n, 1), np.uint8)
    iters = C.c_int(0)
    f = lib().orc_find_fundamental
    f.restype = C.c_int
    found = f(_p(p1), _p(p2), C.c_int(n), C.c_double(thresh), C.c_double(confidence),
              C.c_int(max_iters), _p(F), _p(mask), C.byref(iters))
    return bool(found), F.reshape(3, 3), mask[:n], iters.value


def fm_7point(p1, p2):
    """run7Point on seven correspondences: [k, 3, 3] matrices, k in 0..3."""
    p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(7, 2)
    p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(7, 2)
    F = np.zeros(27)
    f = lib().orc_fm_7point
    f.restype = C.c_int
    k = f(_p(p1), _p(p2), _p(F))
    return F.reshape(3, 3, 3)[:max(k, 0)]


def fm_subsets(p1, p2, max_iters=1000, max_attempts=10000):
    p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(-1, 2)
    idx = np.zeros((max_iters, 7), np.int32)
    f = lib().orc_fm_subsets
    f.restype = C.c_int
    k = f(_p(p1), _p(p2), C.c_int(p1.shape[0]), C.c_int(max_iters), C.c_int(max_attempts), _p(idx))
    return idx[:k]


def ransac_update_num_iters(p, ep, model_points, max_iters):
    f = lib().orc_ransac_update_num_iters
    f.restype = C.c_int
    return f(C.c_double(p), C.c_double(ep), C.c_int(model_points), C.c_int(max_iters))


def svd4_null(A):
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(4, 4)
    w = np.zeros(4)
    v = np.zeros(4)
    lib().orc_svd4_null(_p(A), _p(w), _p(v))
    return w, v


def min_eigen_val(img):
    """cv::cornerMinEigenVal(img, 3, 3): float32 [h, w]."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    eig = np.zeros((h, w), np.float32)
    lib().orc_min_eigen_val(_p(img), C.c_int(w), C.c_int(h), C.c_size_t(w), _p(eig))
    return eig


def good_features(img, max_corners, quality, min_distance, return_candidates=False):
    """cv::goodFeaturesToTrack(img, maxCorners, qualityLevel, minDistance): float32 [n, 2]."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = w * h if max_corners <= 0 else max_corners
    out = np.zeros((max(cap, 1), 2), np.float32)
    ncand = C.c_int(0)
    f = lib().orc_good_features
    f.restype = C.c_int
    n = f(_p(img), C.c_int(w), C.c_int(h), C.c_size_t(w), C.c_int(max_corners), C.c_double(quality),
          C.c_double(min_distance), _p(out), C.c_int(cap), C.byref(ncand))
    if return_candidates:
        return out[:n].copy(), ncand.value
    return out[:n].copy()


def em_5point(m1, m2):
    """Five normalised correspondences -> list of essential matrices [k, 3, 3]."""
    m1 = np.ascontiguousarray(m1, dtype=np.float64).reshape(5, 2)
    m2 = np.ascontiguousarray(m2, dtype=np.float64).reshape(5, 2)
    out = np.zeros((10, 9))
    f = lib().orc_em_5point
    f.restype = C.c_int
    n = f(_p(m1), _p(m2), _p(out))
    return out[:n].reshape(n, 3, 3).copy()


def em_subsets(count, max_iters=1000):
    out = np.zeros((max_iters, 5), np.int32)
    f = lib().orc_em_subsets
    f.restype = C.c_int
    n = f(C.c_int(count), C.c_int(max_iters), _p(out))
    return out[:n]


def find_essential(p1, p2, K, method=0, prob=0.99, threshold=1.0, max_iters=1000):
    """cv::findEssentialMat: returns (E [k, 3, 3] (k = 1 normally), mask [n] uint8, iterations)."""
    p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
    n = p1.shape[0]
    E = np.zeros((10, 9))
    mask = np.zeros(max(n, 1), np.uint8)
    iters = C.c_int(0)
    f = lib().orc_find_essential
    f.restype = C.c_int
    k = f(_p(p1), _p(p2), C.c_int(n), _p(K), C.c_int(method), C.c_double(prob), C.c_double(threshold),
          C.c_int(max_iters), _p(E), _p(mask), C.byref(iters))
    return E[:k].reshape(k, 3, 3).copy(), mask[:n], iters.value


def decompose_essential(E):
    E = np.ascontiguousarray(E, dtype=np.float64).reshape(3, 3)
    R1, R2, t = np.zeros(9), np.zeros(9), np.zeros(3)
    lib().orc_decompose_essential(_p(E), _p(R1), _p(R2), _p(t))
    return R1.reshape(3, 3), R2.reshape(3, 3), t


def recover_pose(E, p1, p2, K, mask=None):
    """cv::recoverPose: returns (count, R, t, mask)."""
    E = np.ascontiguousarray(E, dtype=np.float64).reshape(3, 3)
    p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
    n = p1.shape[0]
    R, t = np.zeros(9), np.zeros(3)
    m = np.ones(max(n, 1), np.uint8) if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).copy()
    f = lib().orc_recover_pose
    f.restype = C.c_int
    cnt = f(_p(E), _p(p1), _p(p2), C.c_int(n), _p(K), _p(R), _p(t), _p(m))
    return cnt, R.reshape(3, 3), t, m[:n]


def essential_residual(E, p0, p1, K):
    E = np.ascontiguousarray(E, dtype=np.float64).reshape(3, 3)
    p0 = np.ascontiguousarray(p0, dtype=np.float32).reshape(2)
    p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(2)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
    f = lib().orc_essential_residual
    f.restype = C.c_double
    return float(f(_p(E), _p(p0), _p(p1), _p(K)))
