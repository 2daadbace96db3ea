"""Seeded synthetic inputs for the tests and the benchmark (no datasets exist
for the reference; shapes follow BASELINE.json's configs)."""
import numpy as np


def perturb_descriptors(rng, a, flip_p=0.05, reuse=0.7):
    """Next frame of a binary-descriptor chain: `reuse` of the rows are re-used in
    permuted order with each bit flipped with probability flip_p, the rest are fresh."""
    n, d = a.shape
    perm = rng.permutation(n)
    b = a[perm].copy()
    nre = int(reuse * n)
    flips = rng.random((nre, d * 8)) < flip_p
    b[:nre] ^= np.packbits(flips, axis=1)
    b[nre:] = rng.integers(0, 256, size=(n - nre, d), dtype=np.uint8)
    return b


def descriptor_chain(n_frames, rows, d=32, seed=1234):
    """[n_frames, rows, d] uint8: each frame perturbs its predecessor."""
    rng = np.random.default_rng(seed)
    frames = [rng.integers(0, 256, size=(rows, d), dtype=np.uint8)]
    for _ in range(n_frames - 1):
        frames.append(perturb_descriptors(rng, frames[-1]))
    return np.stack(frames)


def float_descriptors(nq, nt, dim=128, seed=99, integer=True):
    """SIFT-like float descriptors: 70 % of the train rows are noisy copies of query rows."""
    rng = np.random.default_rng(seed)
    if integer:
        q = rng.gamma(0.8, 30.0, size=(nq, dim))
        q = np.minimum(np.floor(q), 255.0)
        t = np.minimum(np.floor(rng.gamma(0.8, 30.0, size=(nt, dim))), 255.0)
        nre = min(int(0.7 * nt), nq)
        idx = rng.permutation(nq)[:nre]
        noise = np.rint(rng.normal(0, 6.0, size=(nre, dim)))
        t[:nre] = np.clip(q[idx] + noise, 0, 255)
    else:
        q = rng.standard_normal((nq, dim))
        t = rng.standard_normal((nt, dim))
        nre = min(int(0.7 * nt), nq)
        idx = rng.permutation(nq)[:nre]
        t[:nre] = q[idx] + 0.15 * rng.standard_normal((nre, dim))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        t /= np.linalg.norm(t, axis=1, keepdims=True)
    return q.astype(np.float32), t.astype(np.float32)


def texture(h, w, seed=42, octaves=((2.0, 1.0), (5.0, 0.8), (11.0, 0.6))):
    """Band-limited noise: white noise low-passed by Gaussians in the Fourier domain."""
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.rfftfreq(w)[None, :]
    acc = np.zeros((h, w))
    for sigma, gain in octaves:
        nz = rng.standard_normal((h, w))
        gsn = np.exp(-2 * (np.pi ** 2) * (sigma ** 2) * (fx ** 2 + fy ** 2))
        f = np.fft.irfft2(np.fft.rfft2(nz) * gsn, s=(h, w))
        acc += gain * f / f.std()
    return acc


def to_u8(x):
    x = (x - x.mean()) / x.std()
    return np.clip(np.rint(128 + 48 * x), 0, 255).astype(np.uint8)


def affine_warp(img, shift=(1.7, -1.1), angle_deg=0.3):
    """out(p) = img(A^-1 (p - c) + c - shift) with bilinear sampling (float image)."""
    h, w = img.shape
    a = np.deg2rad(angle_deg)
    ca, sa = np.cos(a), np.sin(a)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    dx, dy = xs - cx - shift[0], ys - cy - shift[1]
    sx = ca * dx + sa * dy + cx
    sy = -sa * dx + ca * dy + cy
    x0 = np.clip(np.floor(sx).astype(int), 0, w - 2)
    y0 = np.clip(np.floor(sy).astype(int), 0, h - 2)
    fx = np.clip(sx - x0, 0, 1)
    fy = np.clip(sy - y0, 0, 1)
    return (img[y0, x0] * (1 - fx) * (1 - fy) + img[y0, x0 + 1] * fx * (1 - fy) +
            img[y0 + 1, x0] * (1 - fx) * fy + img[y0 + 1, x0 + 1] * fx * fy)


def warp_points(pts, shape, shift=(1.7, -1.1), angle_deg=0.3):
    """Where affine_warp moves image points (x, y)."""
    h, w = shape
    a = np.deg2rad(angle_deg)
    ca, sa = np.cos(a), np.sin(a)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    dx, dy = pts[:, 0] - cx, pts[:, 1] - cy
    return np.stack([ca * dx - sa * dy + cx + shift[0], sa * dx + ca * dy + cy + shift[1]], 1)


def lk_pair(h=720, w=1280, seed=42, shift=(1.7, -1.1), angle_deg=0.3):
    tex = texture(h, w, seed)
    return to_u8(tex), to_u8(affine_warp(tex, shift, angle_deg))


def grid_points(h, w, n, seed=42, margin=15):
    """n points on a jittered grid at least `margin` pixels from the border."""
    rng = np.random.default_rng(seed)
    aspect = (w - 2 * margin) / (h - 2 * margin)
    ny = max(1, int(np.floor(np.sqrt(n / aspect))))
    nx = int(np.ceil(n / ny))
    gx = np.linspace(margin, w - 1 - margin, nx)
    gy = np.linspace(margin, h - 1 - margin, ny)
    xs, ys = np.meshgrid(gx, gy)
    pts = np.stack([xs.ravel(), ys.ravel()], 1)[:n]
    jit = 0.45 * min(gx[1] - gx[0] if nx > 1 else 1.0, gy[1] - gy[0] if ny > 1 else 1.0)
    pts += rng.uniform(-jit, jit, pts.shape)
    pts[:, 0] = np.clip(pts[:, 0], margin, w - 1 - margin)
    pts[:, 1] = np.clip(pts[:, 1], margin, h - 1 - margin)
    return pts.astype(np.float32)


def rot_y(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def triangulation_set(n, seed=7, sigma=0.5, outliers=0.02):
    """Two-view correspondences of points in a 10x10 box at depth 5-50 (config C3)."""
    rng = np.random.default_rng(seed)
    K = np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]])
    R = rot_y(5.0)
    t = np.array([-1.0, 0.0, 0.0])
    prev_pose = np.hstack([np.eye(3), np.zeros((3, 1))])
    curr_pose = np.hstack([R, t[:, None]])
    X = np.empty((n, 3))
    X[:, 0] = rng.uniform(-5, 5, n)
    X[:, 1] = rng.uniform(-5, 5, n)
    X[:, 2] = rng.uniform(5, 50, n)
    p1 = X @ K.T
    p1 = p1[:, :2] / p1[:, 2:]
    xc = X @ R.T + t
    p2 = xc @ K.T
    p2 = p2[:, :2] / p2[:, 2:]
    p1 += rng.normal(0, sigma, p1.shape)
    p2 += rng.normal(0, sigma, p2.shape)
    bad = rng.random(n) < outliers
    p2[bad] += rng.uniform(-60, 60, (int(bad.sum()), 2))
    return dict(K=K, R=R, t=t, prev_pose=prev_pose, curr_pose=curr_pose,
                p1=p1.astype(np.float32), p2=p2.astype(np.float32), X=X)
