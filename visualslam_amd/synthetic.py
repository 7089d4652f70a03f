"""Seeded synthetic inputs for the point-landmark update and the corner front end.

Scene(L, seed): a scenario-2 style filter state (reference src/SLAM.cpp:96-139:
sigma = 7 px, kappa = 0.5, initial pose (0,-5,-1.8 | -pi/2, pi, 0)) with L
landmarks that all project inside the reference's landmark-initialisation
margins (src/SLAM.cpp:119-120: 250 px in x, 100 px in y), noisy pixel
measurements, Bernoulli(0.9) visibility and an upper-triangular prior
square-root covariance.  Frame(seed): a 1920x1080 u8 image of rectangles and
discs on a smooth gradient plus Gaussian noise.

Pure numpy; used by tests and bench.py to build inputs (never by the kernels).
"""
import os
import re

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
CAMERA_XML = os.path.join(DATA_DIR, "camera.xml")


def read_camera_xml(path=CAMERA_XML):
    """Minimal reader of the cv::FileStorage layout written by
    CameraParameters::write (reference src/cameraModel.cpp:60-69).
    Returns dict(K=3x3, dist=12, flag, width, height)."""
    txt = open(path).read()

    def block(tag):
        m = re.search(r"<%s[^>]*>(.*?)</%s>" % (tag, tag), txt, re.S)
        if not m:
            raise ValueError("camera xml: missing <%s>" % tag)
        return m.group(1)

    def mat(tag):
        b = block(tag)
        rows = int(re.search(r"<rows>(\d+)</rows>", b).group(1))
        cols = int(re.search(r"<cols>(\d+)</cols>", b).group(1))
        data = [float(t) for t in re.search(r"<data>(.*?)</data>", b, re.S).group(1).split()]
        return np.array(data).reshape(rows, cols)

    K = mat("camera_matrix")
    dist = mat("distortion_coefficients").ravel()
    flag = int(block("flag").strip())
    w, h = [int(t) for t in block("imageSize").split()]
    return dict(K=K, dist=dist, flag=flag, width=w, height=h)


def _coeffs(cam):
    """Coefficient gating of reference src/cameraModel.hpp:114-148."""
    d = np.zeros(12)
    d[:len(cam["dist"])] = cam["dist"]
    rational = (cam["flag"] & 0x4000) == 0x4000
    thin = (cam["flag"] & 0x8000) == 0x8000
    k1, k2, p1, p2, k3 = d[0], d[1], d[2], d[3], d[4]
    if rational:
        s = d[8:12] if thin else d[5:9]
    else:
        s = np.zeros(4)
    k456 = d[5:8] if thin else np.zeros(3)
    return k1, k2, k3, k456[0], k456[1], k456[2], p1, p2, s[0], s[1], s[2], s[3]


def distort(cam, u, v):
    k1, k2, k3, k4, k5, k6, p1, p2, s1, s2, s3, s4 = _coeffs(cam)
    r2 = u * u + v * v
    r4 = r2 * r2
    r6 = r4 * r2
    c = (1 + k1 * r2 + k2 * r4 + k3 * r6) / (1 + k4 * r2 + k5 * r4 + k6 * r6)
    up = c * u + 2 * p1 * u * v + p2 * (r2 + 2 * u * u) + s1 * r2 + s2 * r4
    vp = c * v + 2 * p2 * u * v + p1 * (r2 + 2 * v * v) + s3 * r2 + s4 * r4
    return up, vp


def rpy2rot(th):
    cp, sp = np.cos(th[0]), np.sin(th[0])
    ct, st = np.cos(th[1]), np.sin(th[1])
    cs, ss = np.cos(th[2]), np.sin(th[2])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Ry = np.array([[ct, 0, st], [0, 1, 0], [-st, 0, ct]])
    Rz = np.array([[cs, -ss, 0], [ss, cs, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def project(cam, eta, r):
    """Pixels (N,2) of world points r (N,3); no culling."""
    R = rpy2rot(eta[3:6])
    q = (r - eta[:3]) @ R
    u, v = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    up, vp = distort(cam, u, v)
    K = cam["K"]
    return np.stack([K[0, 0] * up + K[0, 2], K[1, 1] * vp + K[1, 2]], axis=1)


def undistort_normalized(cam, xd, yd, iters=60):
    """Invert the distortion map by damped Newton with a finite-difference Jacobian."""
    u, v = xd.copy(), yd.copy()
    h = 1e-7
    for _ in range(iters):
        fu, fv = distort(cam, u, v)
        eu, ev = fu - xd, fv - yd
        fu_u, fv_u = distort(cam, u + h, v)
        fu_v, fv_v = distort(cam, u, v + h)
        a, b = (fu_u - fu) / h, (fu_v - fu) / h
        c, d = (fv_u - fv) / h, (fv_v - fv) / h
        det = a * d - b * c
        du = (d * eu - b * ev) / det
        dv = (-c * eu + a * ev) / det
        u -= du
        v -= dv
        if np.max(np.abs(du)) + np.max(np.abs(dv)) < 1e-14:
            break
    return u, v


class Scene:
    """One filter instance: x (n), y (2L), visible (L), S (n x n upper), sigma."""

    def __init__(self, L, seed, cam=None, sigma=7.0, dense_S=True):
        cam = read_camera_xml() if cam is None else cam
        rng = np.random.default_rng(seed)
        n = 12 + 3 * L
        x = np.zeros(n)
        x[6:12] = np.array([0.0, -5.0, -1.8, -3.14159265359 / 2, 3.14159265359, 0.0]) + 0.01 * rng.standard_normal(6)
        px = np.stack([rng.uniform(250, cam["width"] - 250, L), rng.uniform(100, cam["height"] - 100, L)], axis=1)
        depth = rng.uniform(2.0, 12.0, L)
        K = cam["K"]
        u, v = undistort_normalized(cam, (px[:, 0] - K[0, 2]) / K[0, 0], (px[:, 1] - K[1, 2]) / K[1, 1])
        q = np.stack([u, v, np.ones(L)], axis=1) * depth[:, None]
        R = rpy2rot(x[9:12])
        r = x[6:9] + q @ R.T
        x[12:] = r.ravel()
        h = project(cam, x[6:12], r)
        self.pixel_true = h
        self.y = (h + sigma * rng.standard_normal((L, 2))).ravel()
        self.visible = (rng.random(L) < 0.9).astype(np.uint8)
        diag = np.concatenate([np.full(6, 0.1), np.full(6, 0.001), np.full(3 * L, 0.5)])
        if dense_S:
            S = np.zeros((n, n), order="F")
            mask = np.triu(rng.random((n, n)) < 0.01, 1)
            vals = 0.05 * np.sqrt(np.outer(diag, diag)) * rng.standard_normal((n, n))
            S[mask] = vals[mask]
            S[np.arange(n), np.arange(n)] = diag
            self.S = S
        else:
            self.S = None
        self.diag = diag
        self.x, self.L, self.n, self.sigma, self.cam = x, L, n, sigma, cam


def frame(seed, w=1920, h=1080, n_rect=400, n_disc=200):
    """Synthetic u8 frame: smooth gradient + rectangles (half of them rotated)
    + discs of random intensity + N(0, 2^2) noise, clipped to [0, 255]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = 60.0 + 80.0 * xx / w + 50.0 * yy / h
    for i in range(n_rect):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        hw, hh = rng.uniform(8, 90), rng.uniform(8, 90)
        val = rng.uniform(0, 255)
        ang = rng.uniform(0, np.pi) if i % 2 else 0.0
        rad = int(np.ceil(np.hypot(hw, hh))) + 1
        x0, x1 = max(0, int(cx) - rad), min(w, int(cx) + rad + 1)
        y0, y1 = max(0, int(cy) - rad), min(h, int(cy) + rad + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        sx, sy = xx[y0:y1, x0:x1] - cx, yy[y0:y1, x0:x1] - cy
        ca, sa = np.cos(ang), np.sin(ang)
        m = (np.abs(ca * sx + sa * sy) <= hw) & (np.abs(-sa * sx + ca * sy) <= hh)
        img[y0:y1, x0:x1][m] = val
    for _ in range(n_disc):
        cx, cy, rad = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(5, 60)
        val = rng.uniform(0, 255)
        ir = int(np.ceil(rad)) + 1
        x0, x1 = max(0, int(cx) - ir), min(w, int(cx) + ir + 1)
        y0, y1 = max(0, int(cy) - ir), min(h, int(cy) + ir + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        m = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= rad * rad
        img[y0:y1, x0:x1][m] = val
    img += 2.0 * rng.standard_normal((h, w)).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
