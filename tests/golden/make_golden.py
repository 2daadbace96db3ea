#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/.

Every expected output here comes from NumPy code written independently of both
the C oracle (oracle/*.c) and the HIP kernels: exhaustive distance matrices with
a stable sort, numpy.linalg.svd, and a straight NumPy transcription of the
published OpenCV pyramidal Lucas-Kanade algorithm.  The reference application
(Lowcash/SfM) ships no tests or fixtures and its arithmetic lives in OpenCV,
which is not installed, so these vectors pin the restatement, not OpenCV itself
("parity unpinned").

Run from the repository root:  python tests/golden/make_golden.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F32 = np.float32


# --------------------------------------------------------------------------
# matching
# --------------------------------------------------------------------------
def hamming_matrix(q, t):
    x = q[:, None, :] ^ t[None, :, :]
    return np.unpackbits(x, axis=2).sum(axis=2).astype(np.int32)


def top2_stable(dm):
    """Two smallest per row; equal distances keep the lower column first."""
    order = np.argsort(dm, axis=1, kind="stable")[:, :2]
    nq = dm.shape[0]
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), np.finfo(np.float32).max, np.float32)
    k = order.shape[1]
    idx[:, :k] = order
    dist[:, :k] = np.take_along_axis(dm, order, axis=1).astype(np.float32)
    return idx, dist


def ratio_list(idx, dist, ratio, nt):
    out = []
    if nt < 2:
        return out
    for i in range(idx.shape[0]):
        if dist[i, 0] < F32(ratio) * dist[i, 1]:
            out.append((i, int(idx[i, 0]), float(dist[i, 0])))
    return out


def cross_list(f, b):
    out = []
    for bq, bt, _ in b:
        for fq, ft, fd in f:
            if bq == ft and bt == fq:
                out.append((fq, ft, fd))
                break
    return out


def make_frames(rng, n, d, flip_p=0.05, reuse=0.7):
    a = rng.integers(0, 256, size=(n, d), dtype=np.uint8)
    perm = rng.permutation(n)
    b = a[perm].copy()
    nre = int(reuse * n)
    flips = rng.random((nre, d * 8)) < flip_p
    b[:nre] ^= np.packbits(flips, axis=1)
    b[nre:] = rng.integers(0, 256, size=(n - nre, d), dtype=np.uint8)
    return a, b


def gen_hamming():
    rng = np.random.default_rng(1234)
    out = {}
    for d in (32, 61, 64):
        a, b = make_frames(rng, 96, d)
        # constructed exact ties: duplicate train rows and duplicate query rows
        b[5] = b[40]
        b[17] = b[3]
        b[60] = b[3]
        a[10] = a[11]
        # a query equal to a train row: distance 0
        a[20] = b[33]
        dm = hamming_matrix(a, b)
        idx, dist = top2_stable(dm)
        out[f"q{d}"] = a
        out[f"t{d}"] = b
        out[f"idx{d}"] = idx
        out[f"dist{d}"] = dist
        dmb = hamming_matrix(b, a)
        idxb, distb = top2_stable(dmb)
        for ratio, tag in ((0.5, "r50"), (0.8, "r80")):
            f = ratio_list(idx, dist, ratio, len(b))
            bm = ratio_list(idxb, distb, ratio, len(a))
            x = cross_list(f, bm)
            out[f"fwd{d}_{tag}"] = np.array(f, np.float64).reshape(-1, 3)
            out[f"bwd{d}_{tag}"] = np.array(bm, np.float64).reshape(-1, 3)
            out[f"cross{d}_{tag}"] = np.array(x, np.float64).reshape(-1, 3)
    # tiny train sets
    a = rng.integers(0, 256, size=(7, 32), dtype=np.uint8)
    for nt in (1, 2, 3):
        t = rng.integers(0, 256, size=(nt, 32), dtype=np.uint8)
        idx, dist = top2_stable(hamming_matrix(a, t))
        out[f"tiny_q_nt{nt}"] = a
        out[f"tiny_t_nt{nt}"] = t
        out[f"tiny_idx_nt{nt}"] = idx
        out[f"tiny_dist_nt{nt}"] = dist
    np.savez_compressed(os.path.join(HERE, "hamming.npz"), **out)


def gen_l2():
    rng = np.random.default_rng(99)
    out = {}
    # integer-valued SIFT-like rows: every partial sum is exact in float32
    q = rng.integers(0, 256, size=(80, 128)).astype(np.float32)
    t = rng.integers(0, 256, size=(112, 128)).astype(np.float32)
    t[7] = t[50]
    t[90] = q[12]
    t[91] = q[12]
    d2 = ((q[:, None, :].astype(np.float64) - t[None, :, :]) ** 2).sum(axis=2)
    assert d2.max() < 2 ** 24
    idx, dist = top2_stable(d2)
    out["int_q"], out["int_t"] = q, t
    out["int_idx"], out["int_sqr"] = idx, dist
    # NORM_L2: float sqrt of the (exact) float32 squared distance
    d1 = np.sqrt(d2.astype(np.float32)).astype(np.float32)
    idx1, dist1 = top2_stable(d1)
    out["int_idx_l2"], out["int_l2"] = idx1, dist1
    # real-valued unit-norm rows, dim 64: float64 truth (index agreement is
    # checked where the float64 gap between neighbours is not a near tie)
    q = rng.standard_normal((64, 64))
    t = rng.standard_normal((96, 64))
    t[:40] = q[:40] + 0.05 * rng.standard_normal((40, 64))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    q = q.astype(np.float32)
    t = t.astype(np.float32)
    d2 = ((q[:, None, :].astype(np.float64) - t[None, :, :].astype(np.float64)) ** 2).sum(axis=2)
    order = np.argsort(d2, axis=1, kind="stable")
    out["real_q"], out["real_t"] = q, t
    out["real_idx"] = order[:, :2].astype(np.int32)
    out["real_sqr64"] = np.take_along_axis(d2, order[:, :3], axis=1)
    np.savez_compressed(os.path.join(HERE, "l2.npz"), **out)


# --------------------------------------------------------------------------
# triangulation
# --------------------------------------------------------------------------
def rot_y(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def gen_triangulate():
    rng = np.random.default_rng(7)
    K = np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]])
    R = rot_y(5.0)
    t = np.array([-1.0, 0.02, 0.05])
    prev_pose = np.hstack([np.eye(3), np.zeros((3, 1))])
    curr_pose = np.hstack([R, t[:, None]])
    n = 400
    Xw = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(5, 50, n)], 1)

    def proj(P, X):
        h = (P @ np.hstack([X, np.ones((len(X), 1))]).T).T
        return h[:, :2] / h[:, 2:3]

    P1, P2 = K @ prev_pose, K @ curr_pose
    p1, p2 = proj(P1, Xw), proj(P2, Xw)
    noise = rng.normal(0, 0.5, (n, 2))
    p2n = p2 + noise
    bad = rng.random(n) < 0.05
    p2n[bad] += rng.uniform(-80, 80, (int(bad.sum()), 2))
    # a few points behind the camera / beyond max distance
    Xfar = np.array([[0.0, 0.0, 400.0], [1.0, 1.0, 600.0]])
    p1 = np.vstack([p1, proj(P1, Xfar)])
    p2n = np.vstack([p2n, proj(P2, Xfar)])
    p1 = p1.astype(np.float32)
    p2n = p2n.astype(np.float32)
    n = len(p1)

    X = np.zeros((n, 3))
    err = np.zeros(n)
    depth = np.zeros(n)
    sv = np.zeros((n, 4))
    for i in range(n):
        x1, y1 = np.float64(p1[i, 0]), np.float64(p1[i, 1])
        x2, y2 = np.float64(p2n[i, 0]), np.float64(p2n[i, 1])
        A = np.stack([x1 * P1[2] - P1[0], y1 * P1[2] - P1[1], x2 * P2[2] - P2[0],
                      y2 * P2[2] - P2[1]])
        _, s, vt = np.linalg.svd(A)
        v = vt[3]
        X[i] = v[:3] / v[3] if v[3] != 0 else v[:3]
        sv[i] = s
        xc = R @ X[i] + t
        u = K[0, 0] * xc[0] / xc[2] + K[0, 2]
        w = K[1, 1] * xc[1] / xc[2] + K[1, 2]
        err[i] = np.hypot(u - x2, w - y2)
        depth[i] = xc[2]
    min_d, max_d, max_e = np.float32(1e-4), np.float32(250.0), np.float32(3.0)
    mask = (err < max_e) & (depth > min_d) & (depth < max_d)
    img = rng.integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)
    px = np.clip(np.rint(p2n[:, 0].astype(np.float64)).astype(int), 0, 1279)
    py = np.clip(np.rint(p2n[:, 1].astype(np.float64)).astype(int), 0, 719)
    rgb = img[py, px]
    # the colour image is regenerated from its seed by the tests (kept out of the file)
    np.savez_compressed(os.path.join(HERE, "triangulate.npz"), K=K, R=R, t=t,
                        prev_pose=prev_pose, curr_pose=curr_pose, p1=p1, p2=p2n, X=X, err=err,
                        depth=depth, mask=mask.astype(np.uint8), sv=sv, rgb=rgb,
                        img_seed=np.array([7]), truth=Xw)

    # standalone 4x4 SVD cases
    A = rng.standard_normal((32, 4, 4)) * np.array([1000.0, 1000.0, 600.0, 800.0])
    w = np.zeros((32, 4))
    v = np.zeros((32, 4))
    for i in range(32):
        _, s, vt = np.linalg.svd(A[i])
        w[i], v[i] = s, vt[3]
    np.savez_compressed(os.path.join(HERE, "svd4.npz"), A=A, w=w, v=v)


# --------------------------------------------------------------------------
# pyramid, Scharr, Lucas-Kanade
# --------------------------------------------------------------------------
def np_pyr_down(img):
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    p = np.pad(img.astype(np.int64), ((2, 3), (2, 3)), mode="reflect")
    k = np.array([1, 4, 6, 4, 1], np.int64)
    hor = sum(k[i] * p[:, i:i + 2 * dw:2] for i in range(5))
    ver = sum(k[i] * hor[i:i + 2 * dh:2, :] for i in range(5))
    return ((ver + 128) >> 8).astype(np.uint8)


def np_scharr(img):
    p = np.pad(img.astype(np.int32), 1, mode="reflect")
    sm_v = 3 * (p[:-2, :] + p[2:, :]) + 10 * p[1:-1, :]
    df_v = p[2:, :] - p[:-2, :]
    ix = sm_v[:, 2:] - sm_v[:, :-2]
    iy = 3 * (df_v[:, 2:] + df_v[:, :-2]) + 10 * df_v[:, 1:-1]
    return np.stack([ix, iy], axis=2).astype(np.int16)


def seq_sum_f32(x):
    x = np.asarray(x).astype(np.float32).ravel()
    return np.cumsum(x, dtype=np.float32)[-1] if x.size else F32(0)


def weights(a, b):
    one = F32(1)
    s = F32(1 << 14)
    w00 = int(np.rint((one - a) * (one - b) * s))
    w01 = int(np.rint(a * (one - b) * s))
    w10 = int(np.rint((one - a) * b * s))
    return w00, w01, w10, (1 << 14) - w00 - w01 - w10


def interp(img, ix, iy, ww, wh, wts, shift):
    w00, w01, w10, w11 = wts
    p = img[iy:iy + wh + 1, ix:ix + ww + 1].astype(np.int64)
    v = p[:-1, :-1] * w00 + p[:-1, 1:] * w01 + p[1:, :-1] * w10 + p[1:, 1:] * w11
    return (v + (1 << (shift - 1))) >> shift


def np_pyrlk(prev, curr, pts, win, max_level, max_count, eps, min_eig, max_err, exact):
    ww, wh = win
    h, w = prev.shape
    sizes = [(w, h)]
    levels = max_level
    sw, sh = w, h
    for lv in range(max_level + 1):
        sw, sh = (sw + 1) // 2, (sh + 1) // 2
        if sw <= ww or sh <= wh:
            levels = lv
            break
    pI, pJ = [prev], [curr]
    for lv in range(1, levels + 1):
        pI.append(np_pyr_down(pI[-1]))
        pJ.append(np_pyr_down(pJ[-1]))
    pad = ((wh, wh), (ww, ww))
    padI = [np.pad(x, pad, mode="reflect") for x in pI]
    padJ = [np.pad(x, pad, mode="reflect") for x in pJ]
    padD = [np.pad(np_scharr(x), pad + ((0, 0),), mode="constant") for x in pI]
    n = len(pts)
    nxt = np.zeros((n, 2), np.float32)
    status = np.ones(n, np.uint8)
    err = np.zeros(n, np.float32)
    eps2 = float(eps) * float(eps)
    scale20 = F32(1.0 / (1 << 20))
    half = np.array([(ww - 1) * 0.5, (wh - 1) * 0.5], np.float32)
    patch0 = None
    for lv in range(levels, -1, -1):
        lw, lh = pI[lv].shape[1], pI[lv].shape[0]
        for i in range(n):
            pp = pts[i] * F32(1.0 / (1 << lv))
            np_ = pp.copy() if lv == levels else nxt[i] * F32(2)
            nxt[i] = np_
            pp = pp - half
            ix, iy = int(np.floor(pp[0])), int(np.floor(pp[1]))
            if ix < -ww or ix >= lw or iy < -wh or iy >= lh:
                if lv == 0:
                    status[i] = 0
                    err[i] = 0
                continue
            wts = weights(pp[0] - F32(ix), pp[1] - F32(iy))
            Iw = interp(padI[lv], ix + ww, iy + wh, ww, wh, wts, 9)
            dx = interp(padD[lv][:, :, 0], ix + ww, iy + wh, ww, wh, wts, 14)
            dy = interp(padD[lv][:, :, 1], ix + ww, iy + wh, ww, wh, wts, 14)
            if lv == 0 and i == 0:
                patch0 = np.stack([Iw, dx, dy]).astype(np.int16)
            if exact:
                A11, A12, A22 = F32(int((dx * dx).sum())), F32(int((dx * dy).sum())), F32(
                    int((dy * dy).sum()))
            else:
                A11, A12, A22 = seq_sum_f32(dx * dx), seq_sum_f32(dx * dy), seq_sum_f32(dy * dy)
            A11, A12, A22 = A11 * scale20, A12 * scale20, A22 * scale20
            D = A11 * A22 - A12 * A12
            me = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + F32(4) * A12 * A12)) / F32(
                2 * ww * wh)
            if me < F32(min_eig) or D < np.finfo(np.float32).eps:
                if lv == 0:
                    status[i] = 0
                continue
            D = F32(1) / D
            np_ = np_ - half
            pd = np.zeros(2, np.float32)
            for j in range(max_count):
                jx, jy = int(np.floor(np_[0])), int(np.floor(np_[1]))
                if jx < -ww or jx >= lw or jy < -wh or jy >= lh:
                    if lv == 0:
                        status[i] = 0
                    break
                wts = weights(np_[0] - F32(jx), np_[1] - F32(jy))
                diff = interp(padJ[lv], jx + ww, jy + wh, ww, wh, wts, 9) - Iw
                if exact:
                    b1, b2 = F32(int((diff * dx).sum())), F32(int((diff * dy).sum()))
                else:
                    b1, b2 = seq_sum_f32(diff * dx), seq_sum_f32(diff * dy)
                b1, b2 = b1 * scale20, b2 * scale20
                d = np.array([(A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D], np.float32)
                np_ = np_ + d
                nxt[i] = np_ + half
                if float(d[0]) * float(d[0]) + float(d[1]) * float(d[1]) <= eps2:
                    break
                if j > 0 and abs(float(d[0] + pd[0])) < 0.01 and abs(float(d[1] + pd[1])) < 0.01:
                    nxt[i] = nxt[i] - d * F32(0.5)
                    break
                pd = d
            if status[i] and lv == 0:
                e = nxt[i] - half
                ex, ey = int(np.floor(e[0])), int(np.floor(e[1]))
                if ex < -ww or ex >= lw or ey < -wh or ey >= lh:
                    status[i] = 0
                    continue
                wts = weights(e[0] - F32(ex), e[1] - F32(ey))
                diff = interp(padJ[lv], ex + ww, ey + wh, ww, wh, wts, 9) - Iw
                err[i] = seq_sum_f32(np.abs(diff)) * F32(1) / F32(32 * ww * wh)
    status[err > F32(max_err)] = 0
    return nxt, status, err, levels, patch0


def texture(rng, h, w, octaves=((3.0, 1.0), (1.5, 0.6), (6.0, 0.8))):
    """Band-limited noise: white noise blurred by Gaussians in the Fourier domain."""
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.fftfreq(w)[None, :]
    acc = np.zeros((h, w))
    for sigma, gain in octaves:
        nz = rng.standard_normal((h, w))
        g = np.exp(-2 * (np.pi ** 2) * (sigma ** 2) * (fx ** 2 + fy ** 2))
        f = np.real(np.fft.ifft2(np.fft.fft2(nz) * g))
        acc += gain * f / f.std()
    return acc


def shift_image(img, dx, dy):
    """Fourier shift: out(x, y) = img(x - dx, y - dy)."""
    h, w = img.shape
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.fftfreq(w)[None, :]
    ph = np.exp(-2j * np.pi * (fx * dx + fy * dy))
    return np.real(np.fft.ifft2(np.fft.fft2(img) * ph))


def to_u8(x):
    x = (x - x.mean()) / x.std()
    return np.clip(np.rint(128 + 45 * x), 0, 255).astype(np.uint8)


def gen_lk():
    rng = np.random.default_rng(42)
    out = {}
    # byte-exact pyramid / Scharr cases incl. odd sizes and tiny images
    for name, (h, w) in {"a": (48, 64), "b": (37, 51), "c": (5, 7), "d": (2, 3), "e": (1, 9)}.items():
        img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        out[f"pyr_src_{name}"] = img
        out[f"pyr_dst_{name}"] = np_pyr_down(img) if min(h, w) >= 3 else np.zeros((0, 0), np.uint8)
        if min(h, w) >= 2:
            out[f"scharr_{name}"] = np_scharr(img)
    # LK: translated texture, truth = the shift
    tex = texture(rng, 48, 64)
    shift = (1.3, -0.7)
    prev = to_u8(tex)
    curr = to_u8(shift_image(tex, *shift))
    gx, gy = np.meshgrid(np.arange(8, 60, 9.5), np.arange(8, 44, 8.25))
    pts = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    pts += rng.uniform(-0.5, 0.5, pts.shape).astype(np.float32)
    # points on and outside the border exercise the out-of-range branches
    extra = np.array([[0.2, 0.4], [63.5, 47.2], [-30.0, 10.0], [70.0, 60.0], [31.5, 0.0]],
                     np.float32)
    pts = np.vstack([pts, extra]).astype(np.float32)
    out["lk_prev"], out["lk_curr"], out["lk_pts"] = prev, curr, pts
    out["lk_shift"] = np.array(shift)
    for exact in (0, 1):
        nxt, st, er, levels, patch0 = np_pyrlk(prev, curr, pts, (9, 9), 3, 30, 0.01, 1e-4, 10.0,
                                               bool(exact))
        out[f"lk_next_{exact}"], out[f"lk_status_{exact}"], out[f"lk_err_{exact}"] = nxt, st, er
        out["lk_levels"] = np.array([levels])
        out["lk_patch0"] = patch0
    # a second configuration: 7x5 window, 2 levels requested, COUNT only
    nxt, st, er, levels, _ = np_pyrlk(prev, curr, pts, (7, 5), 1, 10, 0.0, 1e-4, 10.0, True)
    out["lk2_next"], out["lk2_status"], out["lk2_err"] = nxt, st, er
    np.savez_compressed(os.path.join(HERE, "lk.npz"), **out)


# --------------------------------------------------------------------------
# Shi-Tomasi corners (goodFeaturesToTrack)
# --------------------------------------------------------------------------
def np_min_eigen_val(img):
    """cornerMinEigenVal(img, 3, 3) in float32, whole-array NumPy: Sobel with the scale
    1 / (4 * 3 * 255) folded into the smoothing taps, products, 3x3 box sum (rows first),
    smaller eigenvalue.  Borders: reflect-101 == numpy 'reflect'."""
    h, w = img.shape
    k1 = F32(1.0 / (4.0 * 3.0 * 255.0))
    k0 = F32(2.0 / (4.0 * 3.0 * 255.0))
    p = np.pad(img.astype(np.int32), 1, mode="reflect")
    up, mid, dn = p[:-2], p[1:-1], p[2:]
    # Dx: horizontal difference (exact integers), vertical [1 2 1] * scale
    diff = lambda r: (r[:, 2:] - r[:, :-2]).astype(F32)
    dx = (diff(up) + diff(dn)) * k1 + diff(mid) * k0
    # Dy: horizontal [1 2 1] * scale, vertical difference
    smooth = lambda r: (r[:, 2:] + r[:, :-2]).astype(F32) * k1 + r[:, 1:-1].astype(F32) * k0
    dy = smooth(dn) - smooth(up)
    out = []
    for prod in (dx * dx, dx * dy, dy * dy):
        q = np.pad(prod.astype(F32), 1, mode="reflect")
        rows = (q[:, :-2] + q[:, 1:-1]) + q[:, 2:]
        out.append((rows[:-2] + rows[1:-1]) + rows[2:])
    a, b, c = out[0] * F32(0.5), out[1], out[2] * F32(0.5)
    return ((a + c) - np.sqrt((a - c) * (a - c) + b * b)).astype(F32)


def np_good_features(img, max_corners, quality, min_distance):
    h, w = img.shape
    eig = np_min_eigen_val(img)
    thresh = F32(np.float64(eig.max()) * quality)
    eig = np.where(eig > thresh, eig, F32(0))
    inner = eig[1:-1, 1:-1]
    dil = np.max([eig[1 + j:h - 1 + j, 1 + i:w - 1 + i] for j in (-1, 0, 1) for i in (-1, 0, 1)], axis=0)
    ys, xs = np.nonzero((inner != 0) & (inner == dil))
    ys, xs = ys + 1, xs + 1
    vals = eig[ys, xs]
    pos = ys * w + xs
    order = np.lexsort((-pos, -vals.astype(np.float64)))   # value descending, then address descending
    ys, xs = ys[order], xs[order]
    kept = []
    if min_distance >= 1:
        cell = int(np.rint(min_distance))      # cvRound: half to even, like rint
        md2 = min_distance * min_distance
        for x, y in zip(xs, ys):
            ok = True
            for (ax, ay) in kept:
                if abs(ax // cell - x // cell) <= 1 and abs(ay // cell - y // cell) <= 1:
                    ddx, ddy = F32(x) - F32(ax), F32(y) - F32(ay)
                    if np.float64(F32(ddx * ddx + ddy * ddy)) < md2:
                        ok = False
                        break
            if ok:
                kept.append((int(x), int(y)))
                if max_corners > 0 and len(kept) == max_corners:
                    break
    else:
        kept = [(int(x), int(y)) for x, y in zip(xs, ys)]
        if max_corners > 0:
            kept = kept[:max_corners]
    return np.array(kept, F32).reshape(-1, 2), len(order)


def gen_corners():
    rng = np.random.default_rng(77)
    out = {}
    a = to_u8(texture(rng, 60, 80))
    b = rng.integers(0, 256, size=(33, 47), dtype=np.uint8)
    c = np.zeros((24, 40), np.uint8)          # two identical blobs: equal strengths, tie order
    c[5:9, 6:10] = 200
    c[5:9, 26:30] = 200
    c[15:19, 16:20] = 90
    d = np.full((9, 11), 37, np.uint8)        # constant image: no corner at all
    cases = {"a": (a, 50, 0.01, 5.0), "a_all": (a, 0, 0.05, 3.0), "a_frac": (a, 0, 0.01, 2.5),
             "a_nodist": (a, 25, 0.2, 0.0), "b": (b, 30, 0.1, 4.0), "c": (c, 0, 0.01, 6.0),
             "d": (d, 10, 0.01, 3.0)}
    for name, (img, mc, q, md) in cases.items():
        pts, ncand = np_good_features(img, mc, q, md)
        out[f"img_{name}"] = img
        out[f"args_{name}"] = np.array([mc, q, md], np.float64)
        out[f"pts_{name}"] = pts
        out[f"ncand_{name}"] = np.array([ncand])
    out["eig_a"] = np_min_eigen_val(a)
    out["eig_b"] = np_min_eigen_val(b)
    np.savez_compressed(os.path.join(HERE, "corners.npz"), **out)


if __name__ == "__main__":
    import sys

    todo = sys.argv[1:] or ["hamming", "l2", "triangulate", "lk", "corners"]
    for name in todo:
        globals()["gen_" + name]()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
