"""ctypes binding of the C ABI in include/sfm_hip.h (sfm_amd/lib/libsfm_hip.so).

torch is imported before the library is loaded so that both share one HIP
runtime in the process (torch's bundled libamdhip64 has the same soname as the
system one the library was linked against).
"""
import ctypes as C
import os

import numpy as np

try:  # the library itself does not need torch; sharing its HIP runtime does
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsfm_hip.so")

POSE_RANSAC, POSE_LMEDS = 0, 1  # SFM_POSE_*
NORM_HAMMING, NORM_L2, NORM_L2SQR = 0, 1, 2
TERM_COUNT, TERM_EPS = 1, 2

DMATCH_DTYPE = np.dtype(
    [("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]
)

# every symbol include/sfm_hip.h declares
EXPORTS = [
    "sfm_ctx_create", "sfm_ctx_create_on_stream", "sfm_ctx_destroy", "sfm_ctx_synchronize",
    "sfm_ctx_reserve", "sfm_last_error", "sfm_version",
    "sfm_ctx_set_profiling", "sfm_profile_kernel_count", "sfm_profile_kernel_name",
    "sfm_ctx_profile_read",
    "sfm_l2_exhaustive_count", "sfm_knn2", "sfm_knn2_dev", "sfm_ratio_matches", "sfm_robust_matches",
    "sfm_match_pairs_dev", "sfm_match_pairs_workspace",
    "sfm_match_pairs_compact_dev", "sfm_match_pairs_compact_workspace", "sfm_gather_segments_dev",
    "sfm_ratio_cross_dev",
    "sfm_pyrlk", "sfm_pyrlk_batch_dev", "sfm_pyrlk_workspace",
    "sfm_lk_pyramid_bytes", "sfm_lk_build_pyramids_dev", "sfm_pyrlk_pyramids_dev", "sfm_pyrlk_pyramids_workspace",
    "sfm_triangulate", "sfm_triangulate_dev", "sfm_triangulate_matches_dev",
    "sfm_triangulate_noalloc_dev", "sfm_triangulate_noalloc_max_points",
    "sfm_fundamental", "sfm_fundamental_dev", "sfm_fundamental_workspace",
    "sfm_epipolar_filter_pairs_dev", "sfm_epipolar_filter_pairs_workspace",
    "sfm_good_features", "sfm_good_features_dev", "sfm_good_features_workspace",
    "sfm_good_features_batch_dev", "sfm_good_features_batch_workspace",
    "sfm_min_eigen_val_dev",
    "sfm_find_essential", "sfm_find_essential_dev", "sfm_find_essential_workspace",
    "sfm_recover_pose", "sfm_recover_pose_dev", "sfm_recover_pose_workspace",
    "sfm_find_camera_pose_dev", "sfm_find_camera_pose_workspace",
    "sfm_filter_tracked_dev",
]


class SfmError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libsfm_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfmError(
                f"{LIB_PATH} is missing: build it with `make -C sfm_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        l.sfm_last_error.restype = C.c_char_p
        l.sfm_last_error.argtypes = [C.c_void_p]
        l.sfm_version.restype = C.c_char_p
        l.sfm_profile_kernel_name.restype = C.c_char_p
        l.sfm_ctx_destroy.restype = None
        l.sfm_ctx_destroy.argtypes = [C.c_void_p]
        for name in ("sfm_match_pairs_workspace", "sfm_match_pairs_compact_workspace", "sfm_pyrlk_workspace",
                     "sfm_lk_pyramid_bytes", "sfm_pyrlk_pyramids_workspace", "sfm_fundamental_workspace",
                     "sfm_epipolar_filter_pairs_workspace", "sfm_good_features_workspace", "sfm_good_features_batch_workspace", "sfm_find_essential_workspace", "sfm_recover_pose_workspace",
                     "sfm_find_camera_pose_workspace"):
            if hasattr(l, name):
                getattr(l, name).restype = C.c_size_t
        _lib = l
    return _lib


def _ptr(a):
    """void* of a numpy array, a torch tensor, an int address or None."""
    if a is None:
        return C.c_void_p(None)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    return C.c_void_p(a.data_ptr())


class Context:
    """An sfm_ctx: a device, a stream and the library's device workspace."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p(None)
        l = lib()
        if stream is None:
            rc = l.sfm_ctx_create(C.c_int(device), C.byref(self._h))
        else:
            rc = l.sfm_ctx_create_on_stream(C.c_int(device), C.c_void_p(int(stream)),
                                            C.byref(self._h))
        if rc != 0:
            raise SfmError(l.sfm_last_error(None).decode())
        self.device = device
        self.stream = None if stream is None else int(stream)   # None: a stream of the context's own

    def close(self):
        if self._h:
            lib().sfm_ctx_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise SfmError(f"[{rc}] " + lib().sfm_last_error(self._h).decode())

    def synchronize(self):
        self.check(lib().sfm_ctx_synchronize(self._h))

    def reserve(self, nbytes):
        self.check(lib().sfm_ctx_reserve(self._h, C.c_size_t(nbytes)))

    # ---- per-kernel event timing -------------------------------------------
    def set_profiling(self, enabled):
        self.check(lib().sfm_ctx_set_profiling(self._h, C.c_int(1 if enabled else 0)))

    def profile(self):
        """{kernel name: (total ms, launches)} since profiling was switched on."""
        out = {}
        for k in range(lib().sfm_profile_kernel_count()):
            ms, n = C.c_double(0), C.c_int(0)
            self.check(lib().sfm_ctx_profile_read(self._h, C.c_int(k), C.byref(ms), C.byref(n)))
            if n.value:
                out[lib().sfm_profile_kernel_name(C.c_int(k)).decode()] = (ms.value, n.value)
        return out

    # ---- matching ---------------------------------------------------------
    @staticmethod
    def _desc(norm, a):
        dt = np.uint8 if norm == NORM_HAMMING else np.float32
        a = np.asarray(a, dtype=dt)
        if a.ndim != 2:
            raise ValueError("descriptors must be a 2-D array (rows x length)")
        # rows may sit in a wider array (row pitch > row length); anything else is made dense
        if not (a.shape[0] == 0 or (a.strides[1] == a.itemsize and a.strides[0] >= a.shape[1] * a.itemsize)):
            a = np.ascontiguousarray(a)
        return a

    def knn2(self, norm, q, t):
        q, t = self._desc(norm, q), self._desc(norm, t)
        nq, d = q.shape
        nt = t.shape[0]
        if nt and t.shape[1] != d:
            raise ValueError("query and train descriptors differ in length")
        idx = np.full((nq, 2), -1, np.int32)
        dist = np.full((nq, 2), np.finfo(np.float32).max, np.float32)
        self.check(lib().sfm_knn2(self._h, C.c_int(norm), _ptr(q), C.c_int(nq),
                                  C.c_size_t(q.strides[0]), _ptr(t), C.c_int(nt),
                                  C.c_size_t(t.strides[0] if nt else d * q.itemsize), C.c_int(d),
                                  _ptr(idx), _ptr(dist)))
        return idx, dist

    def l2_exhaustive_count(self):
        """Queries of the last L2 call that were recomputed against every train row."""
        n = C.c_int(0)
        self.check(lib().sfm_l2_exhaustive_count(self._h, C.byref(n)))
        return n.value

    def knn2_dev(self, norm, d_q, nq, q_stride, d_t, nt, t_stride, d, d_idx, d_dist):
        self.check(lib().sfm_knn2_dev(self._h, C.c_int(norm), _ptr(d_q), C.c_int(nq),
                                      C.c_size_t(q_stride), _ptr(d_t), C.c_int(nt),
                                      C.c_size_t(t_stride), C.c_int(d), _ptr(d_idx),
                                      _ptr(d_dist)))

    def ratio_cross_dev(self, d_fwd_idx, d_fwd_dist, n_prev, d_bwd_idx, d_bwd_dist, n_curr, ratio, d_matches, cap,
                        d_match_count):
        self.check(lib().sfm_ratio_cross_dev(
            self._h, _ptr(d_fwd_idx), _ptr(d_fwd_dist), C.c_int(n_prev), _ptr(d_bwd_idx), _ptr(d_bwd_dist),
            C.c_int(n_curr), C.c_float(ratio), _ptr(d_matches), C.c_int(cap), _ptr(d_match_count)))

    def ratio_matches(self, norm, q, t, ratio):
        q, t = self._desc(norm, q), self._desc(norm, t)
        nq, d = q.shape
        nt = t.shape[0]
        out = np.zeros(max(nq, 1), DMATCH_DTYPE)
        cnt = C.c_int(0)
        self.check(lib().sfm_ratio_matches(self._h, C.c_int(norm), _ptr(q), C.c_int(nq),
                                           C.c_size_t(q.strides[0]), _ptr(t), C.c_int(nt),
                                           C.c_size_t(t.strides[0] if nt else d * q.itemsize),
                                           C.c_int(d), C.c_float(ratio), _ptr(out),
                                           C.byref(cnt)))
        return out[:cnt.value].copy()

    def robust_matches(self, norm, prev, curr, ratio):
        prev, curr = self._desc(norm, prev), self._desc(norm, curr)
        n_prev, n_curr = prev.shape[0], curr.shape[0]
        d = prev.shape[1] if n_prev else curr.shape[1]
        out = np.zeros(max(min(n_prev, n_curr), 1), DMATCH_DTYPE)
        cnt = C.c_int(0)
        self.check(lib().sfm_robust_matches(
            self._h, C.c_int(norm), _ptr(prev), C.c_int(n_prev),
            C.c_size_t(prev.strides[0] if n_prev else d * prev.itemsize), _ptr(curr),
            C.c_int(n_curr), C.c_size_t(curr.strides[0] if n_curr else d * curr.itemsize),
            C.c_int(d), C.c_float(ratio), _ptr(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def match_pairs_dev(self, norm, d_desc, n_frames, frame_stride, rows_per_frame, d_counts, d,
                        d_pairs, n_pairs, ratio, d_matches, cap, d_match_counts):
        self.check(lib().sfm_match_pairs_dev(
            self._h, C.c_int(norm), _ptr(d_desc), C.c_int(n_frames), C.c_size_t(frame_stride),
            C.c_int(rows_per_frame), _ptr(d_counts), C.c_int(d), _ptr(d_pairs), C.c_int(n_pairs),
            C.c_float(ratio), _ptr(d_matches), C.c_int(cap), _ptr(d_match_counts)))

    def match_pairs_compact_dev(self, norm, d_desc, n_frames, frame_stride, rows_per_frame, d_counts, d,
                                d_pairs, n_pairs, ratio, d_out, out_cap, d_pair_counts, d_offsets):
        """All pairs' cross-checked matches back to back as 6-byte (q, t, distance) records."""
        self.check(lib().sfm_match_pairs_compact_dev(
            self._h, C.c_int(norm), _ptr(d_desc), C.c_int(n_frames), C.c_size_t(frame_stride),
            C.c_int(rows_per_frame), _ptr(d_counts), C.c_int(d), _ptr(d_pairs), C.c_int(n_pairs),
            C.c_float(ratio), _ptr(d_out), C.c_size_t(out_cap), _ptr(d_pair_counts), _ptr(d_offsets)))

    @staticmethod
    def match_pairs_compact_workspace(norm, n_frames, rows_per_frame, d, n_pairs):
        return int(lib().sfm_match_pairs_compact_workspace(C.c_int(norm), C.c_int(n_frames),
                                                           C.c_int(rows_per_frame), C.c_int(d),
                                                           C.c_int(n_pairs)))

    def gather_segments_dev(self, d_src, d_src_off, d_dst_off, d_len, n_segments, elem_bytes, d_dst):
        self.check(lib().sfm_gather_segments_dev(self._h, _ptr(d_src), _ptr(d_src_off), _ptr(d_dst_off),
                                                 _ptr(d_len), C.c_int(n_segments), C.c_int(elem_bytes),
                                                 _ptr(d_dst)))

    @staticmethod
    def match_pairs_workspace(norm, n_frames, rows_per_frame, d, n_pairs):
        return int(lib().sfm_match_pairs_workspace(C.c_int(norm), C.c_int(n_frames),
                                                   C.c_int(rows_per_frame), C.c_int(d),
                                                   C.c_int(n_pairs)))

    # ---- Lucas-Kanade -----------------------------------------------------
    def pyrlk(self, prev, curr, prev_pts, win=(21, 21), max_level=3,
              crit_type=TERM_COUNT | TERM_EPS, max_count=30, epsilon=0.01,
              min_eig_threshold=1e-4, max_error=10.0):
        prev, curr = np.asarray(prev, dtype=np.uint8), np.asarray(curr, dtype=np.uint8)
        if prev.ndim != 2 or prev.shape != curr.shape:
            raise ValueError("images must be equal-size 8-bit single-channel arrays")
        # rows may be a view into a wider image (a cv::Mat region of interest), each image with
        # its own row pitch; anything else is made dense first
        if not (prev.strides[1] == 1 and prev.strides[0] >= prev.shape[1]):
            prev = np.ascontiguousarray(prev)
        if not (curr.strides[1] == 1 and curr.strides[0] >= curr.shape[1]):
            curr = np.ascontiguousarray(curr)
        h, w = prev.shape
        pts = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
        n = pts.shape[0]
        out = np.zeros((n, 2), np.float32)
        status = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self.check(lib().sfm_pyrlk(
            self._h, _ptr(prev), _ptr(curr), C.c_int(w), C.c_int(h), C.c_size_t(prev.strides[0]),
            C.c_size_t(curr.strides[0]), _ptr(pts), C.c_int(n), C.c_int(win[0]), C.c_int(win[1]), C.c_int(max_level),
            C.c_int(crit_type), C.c_int(max_count), C.c_double(epsilon),
            C.c_double(min_eig_threshold), C.c_float(max_error), _ptr(out), _ptr(status),
            _ptr(err)))
        return out, status, err

    def pyrlk_batch_dev(self, d_prev, d_curr, n_pairs, image_stride, width, height, stride,
                        d_prev_pts, n, win, max_level, crit_type, max_count, epsilon,
                        min_eig_threshold, max_error, d_curr_pts, d_status, d_err):
        self.check(lib().sfm_pyrlk_batch_dev(
            self._h, _ptr(d_prev), _ptr(d_curr), C.c_int(n_pairs), C.c_size_t(image_stride),
            C.c_int(width), C.c_int(height), C.c_size_t(stride), _ptr(d_prev_pts), C.c_int(n),
            C.c_int(win[0]), C.c_int(win[1]), C.c_int(max_level), C.c_int(crit_type),
            C.c_int(max_count), C.c_double(epsilon), C.c_double(min_eig_threshold),
            C.c_float(max_error), _ptr(d_curr_pts), _ptr(d_status), _ptr(d_err)))

    @staticmethod
    def lk_pyramid_bytes(width, height, win, max_level):
        return int(lib().sfm_lk_pyramid_bytes(C.c_int(width), C.c_int(height), C.c_int(win[0]), C.c_int(win[1]),
                                              C.c_int(max_level)))

    def lk_build_pyramids_dev(self, d_frames, n_frames, image_stride, width, height, stride, win, max_level,
                              d_pyramids):
        self.check(lib().sfm_lk_build_pyramids_dev(
            self._h, _ptr(d_frames), C.c_int(n_frames), C.c_size_t(image_stride), C.c_int(width), C.c_int(height),
            C.c_size_t(stride), C.c_int(win[0]), C.c_int(win[1]), C.c_int(max_level), _ptr(d_pyramids)))

    def pyrlk_pyramids_dev(self, d_pyramids, n_pyramids, d_pairs, n_pairs, width, height, d_prev_pts, n, win,
                           max_level, crit_type, max_count, epsilon, min_eig_threshold, max_error, d_curr_pts,
                           d_status, d_err):
        self.check(lib().sfm_pyrlk_pyramids_dev(
            self._h, _ptr(d_pyramids), C.c_int(n_pyramids), _ptr(d_pairs), C.c_int(n_pairs), C.c_int(width),
            C.c_int(height), _ptr(d_prev_pts), C.c_int(n), C.c_int(win[0]), C.c_int(win[1]), C.c_int(max_level),
            C.c_int(crit_type), C.c_int(max_count), C.c_double(epsilon), C.c_double(min_eig_threshold),
            C.c_float(max_error), _ptr(d_curr_pts), _ptr(d_status), _ptr(d_err)))

    @staticmethod
    def pyrlk_pyramids_workspace(n_pairs, width, height, win, max_level):
        return int(lib().sfm_pyrlk_pyramids_workspace(C.c_int(n_pairs), C.c_int(width), C.c_int(height),
                                                      C.c_int(win[0]), C.c_int(win[1]), C.c_int(max_level)))

    @staticmethod
    def pyrlk_workspace(n_pairs, width, height, win, max_level):
        return int(lib().sfm_pyrlk_workspace(C.c_int(n_pairs), C.c_int(width), C.c_int(height),
                                             C.c_int(win[0]), C.c_int(win[1]),
                                             C.c_int(max_level)))

    # ---- triangulation ----------------------------------------------------
    @staticmethod
    def _mats(K, prev_pose, curr_pose, R, t):
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        prev_pose = np.ascontiguousarray(prev_pose, dtype=np.float64).reshape(3, 4)
        curr_pose = np.ascontiguousarray(curr_pose, dtype=np.float64).reshape(3, 4)
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(3, 3)
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
        return K, prev_pose, curr_pose, R, t

    def triangulate(self, K, prev_pose, curr_pose, R, t, prev_pts, curr_pts, bgr=None,
                    min_dist=1e-4, max_dist=250.0, max_proj_err=3.0):
        K, prev_pose, curr_pose, R, t = self._mats(K, prev_pose, curr_pose, R, t)
        p1 = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(curr_pts, dtype=np.float32).reshape(-1, 2)
        if p1.shape != p2.shape:
            raise ValueError("prev and curr point lists differ in length")
        n = p1.shape[0]
        X = np.zeros((n, 3), np.float64)
        rgb = np.zeros((n, 3), np.uint8)
        mask = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float64)
        if bgr is not None:
            bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
            ih, iw = bgr.shape[:2]
            bs = bgr.strides[0]
        else:
            ih = iw = bs = 0
        self.check(lib().sfm_triangulate(
            self._h, _ptr(K), _ptr(prev_pose), _ptr(curr_pose), _ptr(R), _ptr(t), _ptr(p1),
            _ptr(p2), C.c_int(n), _ptr(bgr), C.c_int(iw), C.c_int(ih), C.c_size_t(bs),
            C.c_float(min_dist), C.c_float(max_dist), C.c_float(max_proj_err), _ptr(X), _ptr(rgb),
            _ptr(mask), _ptr(err)))
        return X, rgb, mask, err

    def triangulate_dev(self, K, prev_pose, curr_pose, R, t, d_prev_pts, d_curr_pts, n, d_bgr,
                        img_w, img_h, img_stride, min_dist, max_dist, max_proj_err, d_X, d_rgb,
                        d_mask, d_err=None):
        K, prev_pose, curr_pose, R, t = self._mats(K, prev_pose, curr_pose, R, t)
        self.check(lib().sfm_triangulate_dev(
            self._h, _ptr(K), _ptr(prev_pose), _ptr(curr_pose), _ptr(R), _ptr(t),
            _ptr(d_prev_pts), _ptr(d_curr_pts), C.c_int(n), _ptr(d_bgr), C.c_int(img_w),
            C.c_int(img_h), C.c_size_t(img_stride), C.c_float(min_dist), C.c_float(max_dist),
            C.c_float(max_proj_err), _ptr(d_X), _ptr(d_rgb), _ptr(d_mask), _ptr(d_err)))

    def triangulate_noalloc_dev(self, K, prev_pose, curr_pose, R, t, d_prev_pts, d_curr_pts, n, d_bgr,
                                img_w, img_h, img_stride, min_dist, max_dist, max_proj_err, d_X, d_rgb,
                                d_mask, d_err=None):
        """triangulate_dev without any device allocation (fixed scratch, n <= 2^20)."""
        K, prev_pose, curr_pose, R, t = self._mats(K, prev_pose, curr_pose, R, t)
        self.check(lib().sfm_triangulate_noalloc_dev(
            self._h, _ptr(K), _ptr(prev_pose), _ptr(curr_pose), _ptr(R), _ptr(t),
            _ptr(d_prev_pts), _ptr(d_curr_pts), C.c_int(n), _ptr(d_bgr), C.c_int(img_w),
            C.c_int(img_h), C.c_size_t(img_stride), C.c_float(min_dist), C.c_float(max_dist),
            C.c_float(max_proj_err), _ptr(d_X), _ptr(d_rgb), _ptr(d_mask), _ptr(d_err)))

    # ---- epipolar filter --------------------------------------------------
    def fundamental(self, pts1, pts2, thresh=3.0, confidence=0.99, max_iters=1000):
        """cv::findFundamentalMat(pts1, pts2, mask): returns (found, F [3,3], mask [n] uint8)."""
        p1 = np.ascontiguousarray(pts1, dtype=np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(pts2, dtype=np.float32).reshape(-1, 2)
        if p1.shape != p2.shape:
            raise ValueError("point lists differ in length")
        n = p1.shape[0]
        F = np.zeros(9)
        mask = np.zeros(max(n, 1), np.uint8)
        found = C.c_int(0)
        self.check(lib().sfm_fundamental(self._h, _ptr(p1), _ptr(p2), C.c_int(n), C.c_double(thresh),
                                         C.c_double(confidence), C.c_int(max_iters), _ptr(F),
                                         _ptr(mask), C.byref(found)))
        return bool(found.value), F.reshape(3, 3), mask[:n]

    def fundamental_dev(self, d_pts1, d_pts2, problem_stride, d_counts, count_uniform, n_problems,
                        thresh, confidence, max_iters, d_F, d_mask, d_found):
        self.check(lib().sfm_fundamental_dev(
            self._h, _ptr(d_pts1), _ptr(d_pts2), C.c_size_t(problem_stride), _ptr(d_counts),
            C.c_int(count_uniform), C.c_int(n_problems), C.c_double(thresh), C.c_double(confidence),
            C.c_int(max_iters), _ptr(d_F), _ptr(d_mask), _ptr(d_found)))

    def epipolar_filter_pairs_dev(self, d_keypts, frame_stride, keypt_stride, d_pairs, n_pairs,
                                  d_matches, cap, d_match_counts, thresh=3.0, confidence=0.99,
                                  max_iters=1000, d_mask=None):
        self.check(lib().sfm_epipolar_filter_pairs_dev(
            self._h, _ptr(d_keypts), C.c_size_t(frame_stride), C.c_size_t(keypt_stride), _ptr(d_pairs),
            C.c_int(n_pairs), _ptr(d_matches), C.c_int(cap), _ptr(d_match_counts), C.c_double(thresh),
            C.c_double(confidence), C.c_int(max_iters), _ptr(d_mask)))

    @staticmethod
    def epipolar_filter_pairs_workspace(n_pairs, cap, max_iters=1000):
        return int(lib().sfm_epipolar_filter_pairs_workspace(C.c_int(n_pairs), C.c_int(cap),
                                                             C.c_int(max_iters)))

    def find_essential(self, pts1, pts2, K, method=0, prob=0.99, threshold=1.0, max_iters=1000):
        """cv::findEssentialMat(pts1, pts2, K, method, prob, threshold, mask): returns
        (E [k, 3, 3], mask [n] uint8); k is 1 for a found model, 0 for none, and the number of
        solutions of the direct solve for exactly five points."""
        p1 = np.ascontiguousarray(pts1, dtype=np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(pts2, dtype=np.float32).reshape(-1, 2)
        if p1.shape != p2.shape:
            raise ValueError("point lists differ in length")
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        n = p1.shape[0]
        E = np.zeros((10, 9))
        mask = np.zeros(max(n, 1), np.uint8)
        k = C.c_int(0)
        self.check(lib().sfm_find_essential(self._h, _ptr(p1), _ptr(p2), C.c_int(n), _ptr(K), C.c_int(method),
                                            C.c_double(prob), C.c_double(threshold), C.c_int(max_iters),
                                            _ptr(E), _ptr(mask), C.byref(k)))
        return E[:k.value].reshape(k.value, 3, 3).copy(), mask[:n]

    def find_essential_dev(self, d_pts1, d_pts2, problem_stride, d_counts, count_uniform, n_problems, K,
                           method, prob, threshold, max_iters, d_E, d_mask, d_found):
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        self.check(lib().sfm_find_essential_dev(
            self._h, _ptr(d_pts1), _ptr(d_pts2), C.c_size_t(problem_stride), _ptr(d_counts),
            C.c_int(count_uniform), C.c_int(n_problems), _ptr(K), C.c_int(method), C.c_double(prob),
            C.c_double(threshold), C.c_int(max_iters), _ptr(d_E), _ptr(d_mask), _ptr(d_found)))

    @staticmethod
    def find_essential_workspace(n_problems, problem_stride, max_iters=1000):
        return int(lib().sfm_find_essential_workspace(C.c_int(n_problems), C.c_size_t(problem_stride),
                                                      C.c_int(max_iters)))

    def filter_tracked_dev(self, d_prev, d_curr, d_status, d_valid, n_pairs, n, boundary, d_prev_out, d_curr_out,
                           d_counts):
        """Status + boundary filter of tracked points; boundary = (x, y, w, h) or None."""
        bx, by, bw, bh = boundary if boundary is not None else (0, 0, 0, 0)
        self.check(lib().sfm_filter_tracked_dev(
            self._h, _ptr(d_prev), _ptr(d_curr), _ptr(d_status), _ptr(d_valid), C.c_int(n_pairs), C.c_int(n),
            C.c_int(bx), C.c_int(by), C.c_int(bw), C.c_int(bh), _ptr(d_prev_out), _ptr(d_curr_out), _ptr(d_counts)))

    def recover_pose(self, E, pts1, pts2, K, mask=None):
        """cv::recoverPose(E, pts1, pts2, K, R, t, mask): returns (count, R [3,3], t [3], mask)."""
        E = np.ascontiguousarray(E, dtype=np.float64).reshape(3, 3)
        p1 = np.ascontiguousarray(pts1, dtype=np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(pts2, dtype=np.float32).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        n = p1.shape[0]
        m = np.ones(max(n, 1), np.uint8) if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).copy()
        R, t = np.zeros(9), np.zeros(3)
        good = C.c_int(0)
        self.check(lib().sfm_recover_pose(self._h, _ptr(E), _ptr(p1), _ptr(p2), C.c_int(n), _ptr(K), _ptr(R),
                                          _ptr(t), _ptr(m), C.byref(good)))
        return good.value, R.reshape(3, 3), t, m[:n]

    def find_camera_pose_dev(self, d_pts1, d_pts2, problem_stride, d_counts, count_uniform, n_problems, K,
                             method, prob, threshold, d_E, d_R, d_t, d_mask, d_found, d_inliers):
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        self.check(lib().sfm_find_camera_pose_dev(
            self._h, _ptr(d_pts1), _ptr(d_pts2), C.c_size_t(problem_stride), _ptr(d_counts),
            C.c_int(count_uniform), C.c_int(n_problems), _ptr(K), C.c_int(method), C.c_double(prob),
            C.c_double(threshold), _ptr(d_E), _ptr(d_R), _ptr(d_t), _ptr(d_mask), _ptr(d_found),
            _ptr(d_inliers)))

    @staticmethod
    def find_camera_pose_workspace(n_problems, problem_stride):
        return int(lib().sfm_find_camera_pose_workspace(C.c_int(n_problems), C.c_size_t(problem_stride)))

    def good_features(self, img, max_corners, quality, min_distance):
        """cv::goodFeaturesToTrack(img, maxCorners, qualityLevel, minDistance): float32 [n, 2],
        strongest first (host image in, host corners out)."""
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
            raise ValueError("image must be a 2-D uint8 array with contiguous rows")
        h, w = img.shape
        cap = int(max_corners) if max_corners > 0 else w * h
        out = np.zeros((max(cap, 1), 2), np.float32)
        count = C.c_int(0)
        self.check(lib().sfm_good_features(self._h, _ptr(img), C.c_int(w), C.c_int(h),
                                           C.c_size_t(img.strides[0]), C.c_int(max_corners),
                                           C.c_double(quality), C.c_double(min_distance), _ptr(out),
                                           C.c_int(cap), C.byref(count)))
        return out[:count.value].copy()

    def good_features_dev(self, d_img, w, h, stride, max_corners, quality, min_distance, d_corners, cap,
                          d_info, max_candidates=0):
        self.check(lib().sfm_good_features_dev(
            self._h, _ptr(d_img), C.c_int(w), C.c_int(h), C.c_size_t(stride), C.c_int(max_corners),
            C.c_double(quality), C.c_double(min_distance), C.c_int(max_candidates), _ptr(d_corners),
            C.c_int(cap), _ptr(d_info)))

    def good_features_batch_dev(self, d_imgs, n_images, image_stride, w, h, stride, max_corners, quality,
                                min_distance, d_corners, cap, d_info, max_candidates=0):
        self.check(lib().sfm_good_features_batch_dev(
            self._h, _ptr(d_imgs), C.c_int(n_images), C.c_size_t(image_stride), C.c_int(w), C.c_int(h),
            C.c_size_t(stride), C.c_int(max_corners), C.c_double(quality), C.c_double(min_distance),
            C.c_int(max_candidates), _ptr(d_corners), C.c_int(cap), _ptr(d_info)))

    @staticmethod
    def good_features_batch_workspace(n_images, w, h, max_candidates=0):
        return int(lib().sfm_good_features_batch_workspace(C.c_int(n_images), C.c_int(w), C.c_int(h),
                                                           C.c_int(max_candidates)))

    @staticmethod
    def good_features_workspace(w, h, max_candidates=0):
        return int(lib().sfm_good_features_workspace(C.c_int(w), C.c_int(h), C.c_int(max_candidates)))

    def min_eigen_val_dev(self, d_img, w, h, stride, d_eig):
        self.check(lib().sfm_min_eigen_val_dev(self._h, _ptr(d_img), C.c_int(w), C.c_int(h),
                                               C.c_size_t(stride), _ptr(d_eig)))

    @staticmethod
    def fundamental_workspace(n_problems, max_iters):
        return int(lib().sfm_fundamental_workspace(C.c_int(n_problems), C.c_int(max_iters)))

    def triangulate_matches_dev(self, K, prev_pose, curr_pose, R, t, d_matches, d_count, cap,
                                d_prev_keypts, prev_stride, d_curr_keypts, curr_stride, d_bgr,
                                img_w, img_h, img_stride, min_dist, max_dist, max_proj_err, d_X,
                                d_rgb, d_mask, d_err=None):
        """Triangulates the correspondences a match list (device, as sfm_match_pairs_dev wrote
        it for one pair) selects from two device key point arrays."""
        K, prev_pose, curr_pose, R, t = self._mats(K, prev_pose, curr_pose, R, t)
        self.check(lib().sfm_triangulate_matches_dev(
            self._h, _ptr(K), _ptr(prev_pose), _ptr(curr_pose), _ptr(R), _ptr(t),
            _ptr(d_matches), _ptr(d_count), C.c_int(cap), _ptr(d_prev_keypts),
            C.c_size_t(prev_stride), _ptr(d_curr_keypts), C.c_size_t(curr_stride), _ptr(d_bgr),
            C.c_int(img_w), C.c_int(img_h), C.c_size_t(img_stride), C.c_float(min_dist),
            C.c_float(max_dist), C.c_float(max_proj_err), _ptr(d_X), _ptr(d_rgb), _ptr(d_mask),
            _ptr(d_err)))
