"""ctypes binding of the CPU oracle (oracle/libvslam_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never by the visualslam_amd package.
All matrices are column-major float64 (numpy order='F'), like Eigen::MatrixXd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvslam_oracle.so")


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("d", C.c_double * 12), ("fov_deg", C.c_double),
                ("flag", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("reserved_", C.c_int32)]


class CornerParams(C.Structure):
    _fields_ = [("block_size", C.c_int32), ("border", C.c_int32), ("max_features", C.c_int32),
                ("harris_k", C.c_float), ("harris_thresh", C.c_float), ("shi_thresh", C.c_float)]


FEATURE_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("score", np.float32)])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


_lib = _load()
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _b(a):
    return a.ctypes.data_as(_bp) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _F(a):
    return np.asfortranarray(a, dtype=np.float64)


_lib.orc_world_to_pixel.restype = C.c_int
_lib.orc_single_measure.restype = C.c_int
_lib.orc_point_loglik_value.restype = C.c_double
_lib.orc_point_loglik_autodiff.restype = C.c_long
_lib.orc_log_gaussian.restype = C.c_double
_lib.orc_trs.restype = C.c_int
_lib.orc_measurement_update_iekf.restype = C.c_int
_lib.orc_corner_detect.restype = C.c_int
_lib.orc_normcdf.restype = C.c_double
_lib.orc_normcdf.argtypes = [C.c_double]
_lib.orc_chi2inv.restype = C.c_double
_lib.orc_chi2inv.argtypes = [C.c_double, C.c_double]
_lib.orc_individual_compat.restype = C.c_int
_lib.orc_joint_compat.restype = C.c_int
_lib.orc_joint_compat.argtypes = [_ip, C.c_int, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp]
_lib.orc_snn_density.restype = C.c_double
_lib.orc_snn.restype = C.c_double


def make_camera(fx, fy, cx, cy, dist, flag, width, height, fov_deg=120.0):
    cam = Camera()
    cam.fx, cam.fy, cam.cx, cam.cy = fx, fy, cx, cy
    for i in range(12):
        cam.d[i] = float(dist[i]) if i < len(dist) else 0.0
    cam.fov_deg, cam.flag, cam.width, cam.height = fov_deg, flag, width, height
    return cam


def rot(kind, a):
    R = np.zeros((3, 3), order="F")
    getattr(_lib, "orc_rot" + kind)(C.c_double(a), _d(R))
    return R


def rpy2rot(theta):
    R = np.zeros((3, 3), order="F")
    _lib.orc_rpy2rot(_d(_f64(theta)), _d(R))
    return R


def rot2rpy(R):
    th = np.zeros(3)
    _lib.orc_rot2rpy(_d(_F(R)), _d(th))
    return th


def world_to_pixel(r, eta, cam):
    px = np.full(2, np.nan)
    code = _lib.orc_world_to_pixel(_d(_f64(r)), _d(_f64(eta)), C.byref(cam), _d(px))
    return code, px


def single_measure(j, x, cam, layout=0, want_J=True):
    x = _f64(x)
    n = x.size
    h = np.full(2, np.nan)
    SR = np.zeros((2, 2), order="F")
    J = np.zeros((2, n), order="F") if want_J else None
    code = _lib.orc_single_measure(j, _d(x), n, C.byref(cam), layout, _d(h), _d(SR), _d(J))
    return code, h, SR, J


def bundle_measure(x, cam, layout=0, want_SR=True, want_C=True):
    x = _f64(x)
    n = x.size
    L = (n - 12) // 3
    h = np.zeros(2 * L)
    SR = np.zeros((2 * L, 2 * L), order="F") if want_SR else None
    Cm = np.zeros((2 * L, n), order="F") if want_C else None
    code = np.zeros(L, dtype=np.int32)
    _lib.orc_bundle_measure(_d(x), n, C.byref(cam), layout, _d(h), _d(SR), _d(Cm), _i(code))
    return h, SR, Cm, code


def point_loglik_value(x, y, visible, cam, sigma):
    x, y = _f64(x), _f64(y)
    vis = np.ascontiguousarray(visible, dtype=np.uint8)
    code = np.zeros(vis.size, dtype=np.int32)
    v = _lib.orc_point_loglik_value(_d(x), x.size, _d(y), _b(vis), C.byref(cam), C.c_double(sigma), _i(code))
    return v, code


def _lik(fn, x, y, visible, cam, sigma, with_code=True):
    x, y = _f64(x), _f64(y)
    n = x.size
    vis = np.ascontiguousarray(visible, dtype=np.uint8)
    val = C.c_double(0.0)
    g = np.zeros(n)
    H = np.zeros((n, n), order="F")
    code = np.zeros(vis.size, dtype=np.int32)
    args = [_d(x), n, _d(y), _b(vis), C.byref(cam), C.c_double(sigma), C.byref(val), _d(g), _d(H)]
    if with_code:
        args.append(_i(code))
    fn(*args)
    return val.value, g, H, code


def point_loglik_dual_sparse(x, y, visible, cam, sigma):
    return _lik(_lib.orc_point_loglik_dual_sparse, x, y, visible, cam, sigma)


def point_loglik_analytic(x, y, visible, cam, sigma):
    return _lik(_lib.orc_point_loglik_analytic, x, y, visible, cam, sigma)


def point_loglik_autodiff(x, y, visible, cam, sigma, row_begin=0, row_end=None):
    """Reference-faithful n(n+1)/2-pass evaluation; returns (value, g, H, passes)."""
    x, y = _f64(x), _f64(y)
    n = x.size
    vis = np.ascontiguousarray(visible, dtype=np.uint8)
    val = C.c_double(0.0)
    g = np.zeros(n)
    H = np.zeros((n, n), order="F")
    row_end = n if row_end is None else row_end
    passes = _lib.orc_point_loglik_autodiff(_d(x), n, _d(y), _b(vis), C.byref(cam), C.c_double(sigma),
                                            row_begin, row_end, C.byref(val), _d(g), _d(H))
    return val.value, g, H, passes


def log_gaussian(x, mu, S, want_g=True, want_H=True):
    x, mu, S = _f64(x), _f64(mu), _F(S)
    n = x.size
    g = np.zeros(n) if want_g else None
    H = np.zeros((n, n), order="F") if want_H else None
    v = _lib.orc_log_gaussian(_d(x), _d(mu), _d(S), n, n, _d(g), _d(H))
    return v, g, H


def qr_upper(A):
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    _lib.orc_qr_upper(_d(A), A.shape[0], A.shape[1], A.shape[0])
    return A


def affine_sqrt(S, Cm, SR):
    S, Cm, SR = _F(S), _F(Cm), _F(SR)
    n, m = S.shape[0], Cm.shape[0]
    Syy = np.zeros((m, m), order="F")
    _lib.orc_affine_sqrt(_d(S), n, n, _d(Cm), m, m, _d(SR), m, _d(Syy), m)
    return Syy


def condition_on_marginal(muyx, Syx, y):
    muyx, Syx, y = _f64(muyx), _F(Syx), _f64(y)
    ny = y.size
    nx = muyx.size - ny
    mu = np.zeros(nx)
    Sx = np.zeros((nx, nx), order="F")
    _lib.orc_condition_on_marginal(_d(muyx), _d(Syx), ny, nx, ny + nx, _d(y), _d(mu), _d(Sx), nx)
    return mu, Sx


def iekf_post_sqrt(Q, v):
    Q, v = _F(Q), _f64(v)
    n = v.size
    S = np.zeros((n, n), order="F")
    _lib.orc_iekf_post_sqrt(_d(Q), _d(v), n, _d(S))
    return S


def sym_eig(H):
    H = _F(H)
    n = H.shape[0]
    Q = np.zeros((n, n), order="F")
    v = np.zeros(n)
    _lib.orc_sym_eig(_d(H), n, _d(Q), _d(v))
    return Q, v


def trs(Q, v, g, D):
    Q, v, g = _F(Q), _f64(v), _f64(g)
    p = np.zeros(v.size)
    ret = _lib.orc_trs(_d(Q), _d(v), _d(g), v.size, C.c_double(D), _d(p))
    return ret, p


def measurement_update_iekf(mu, S, y, visible, cam, sigma):
    mu, S, y = _f64(mu), _F(S), _f64(y)
    n = mu.size
    vis = np.ascontiguousarray(visible, dtype=np.uint8)
    mup = np.zeros(n)
    Sp = np.zeros((n, n), order="F")
    it, ev = C.c_int(0), C.c_int(0)
    ret = _lib.orc_measurement_update_iekf(_d(mu), _d(S), n, _d(y), _b(vis), C.byref(cam), C.c_double(sigma),
                                           _d(mup), _d(Sp), C.byref(it), C.byref(ev))
    return ret, mup, Sp, it.value, ev.value


def process_model(x, pos_tune, ori_tune, want_SQ=True, want_J=True):
    x, pt, ot = _f64(x), _F(pos_tune), _F(ori_tune)
    n = x.size
    f = np.zeros(n)
    SQ = np.zeros((n, n), order="F") if want_SQ else None
    J = np.zeros((n, n), order="F") if want_J else None
    _lib.orc_process_model(_d(x), n, _d(pt), _d(ot), _d(f), _d(SQ), _d(J))
    return f, SQ, J


def rk4_sde(xdw, pos_tune, ori_tune, dt, want_J=True):
    xdw, pt, ot = _f64(xdw), _F(pos_tune), _F(ori_tune)
    n = xdw.size // 2
    xn = np.zeros(n)
    J = np.zeros((n, 2 * n), order="F") if want_J else None
    _lib.orc_rk4_sde(_d(xdw), n, _d(pt), _d(ot), C.c_double(dt), _d(xn), _d(J))
    return xn, J


def time_update(mu, S, pos_tune, ori_tune, dt):
    mu, S, pt, ot = _f64(mu), _F(S), _F(pos_tune), _F(ori_tune)
    n = mu.size
    mup = np.zeros(n)
    Sp = np.zeros((n, n), order="F")
    _lib.orc_time_update(_d(mu), _d(S), n, n, _d(pt), _d(ot), C.c_double(dt), _d(mup), _d(Sp), n)
    return mup, Sp


def remove_landmark(mu, S, j):
    mu, S = _f64(mu), _F(S)
    n = mu.size
    mo = np.zeros(n - 3)
    So = np.zeros((n - 3, n - 3), order="F")
    _lib.orc_remove_landmark(_d(mu), _d(S), n, n, int(j), _d(mo), _d(So), n - 3)
    return mo, So


def match_hamming(query, train):
    query = np.ascontiguousarray(query, dtype=np.uint8)
    train = np.ascontiguousarray(train, dtype=np.uint8).reshape(-1, query.shape[1])
    nq, nb = query.shape
    idx = np.zeros(nq, np.int32)
    dist = np.zeros(nq, np.int32)
    _lib.orc_match_hamming(_b(query), nq, _b(train), train.shape[0], nb, _i(idx), _i(dist))
    return idx, dist


def corner_params(block_size=7, border=None, max_features=10000, harris_k=0.04,
                  harris_thresh=1e-4, shi_thresh=1e-3):
    p = CornerParams()
    p.block_size = block_size
    p.border = (block_size // 2 + 2) if border is None else border
    p.max_features = max_features
    p.harris_k, p.harris_thresh, p.shi_thresh = harris_k, harris_thresh, shi_thresh
    return p


def corner_response(gray, params):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    har = np.zeros((h, w), dtype=np.float32)
    shi = np.zeros((h, w), dtype=np.float32)
    _lib.orc_corner_response(_b(gray), w, h, w, C.byref(params),
                             har.ctypes.data_as(C.POINTER(C.c_float)), shi.ctypes.data_as(C.POINTER(C.c_float)))
    return har, shi


def corner_detect(gray, params, which, cap=None):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    cap = params.max_features if cap is None else cap
    out = np.zeros(cap, dtype=FEATURE_DTYPE)
    n = _lib.orc_corner_detect(_b(gray), w, h, w, C.byref(params), which, out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].copy()


def optical_ray(cam, u, v, length=8.0):
    """generateOpticalRay (reference src/SLAM.cpp:876-903): ray of the pixel in camera coordinates, times length."""
    r = np.zeros(3)
    _lib.orc_optical_ray.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
    _lib.orc_optical_ray.restype = None
    _lib.orc_optical_ray(C.byref(cam), float(u), float(v), float(length), r.ctypes.data_as(C.c_void_p))
    return r


def initialize_landmark(mu, S, cam, u, v, length=8.0, kappa=0.5):
    """initializeNewLandmark (reference src/SLAM.cpp:905-941): (mu, S) with the landmark of pixel (u, v) appended."""
    mu, S = _f64(mu), _F(S)
    n = mu.size
    mo, So = np.zeros(n + 3), np.zeros((n + 3, n + 3), order="F")
    _lib.orc_initialize_landmark.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                             C.c_double, C.c_void_p, C.c_void_p]
    _lib.orc_initialize_landmark.restype = None
    _lib.orc_initialize_landmark(mu.ctypes.data_as(C.c_void_p), S.ctypes.data_as(C.c_void_p), n, n, C.byref(cam), float(u), float(v),
                                 float(length), float(kappa), mo.ctypes.data_as(C.c_void_p), So.ctypes.data_as(C.c_void_p))
    return mo, So


def pixel_distance(landmark_xy, x, y, thresh):
    """pixelDistance (reference src/SLAM.cpp:746-760): is any landmark pixel closer than thresh?"""
    lm = np.ascontiguousarray(landmark_xy, dtype=np.float32).reshape(-1, 2)
    _lib.orc_pixel_distance.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_double]
    return bool(_lib.orc_pixel_distance(lm.ctypes.data_as(C.c_void_p), lm.shape[0], float(x), float(y), float(thresh)))


def select_new_landmarks(feats, landmark_xy, feature_thresh, width_thresh, height_thresh, dist_thresh, image_w, image_h, max_new):
    """The selection loop of new landmarks (reference src/SLAM.cpp:449-472): indices into feats, in order."""
    feats = np.ascontiguousarray(feats, dtype=FEATURE_DTYPE)
    lm = np.ascontiguousarray(landmark_xy, dtype=np.float32).reshape(-1, 2)
    sel = np.zeros(max(max_new, 1), dtype=np.int32)
    _lib.orc_select_new_landmarks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                              C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = _lib.orc_select_new_landmarks(feats.ctypes.data_as(C.c_void_p), feats.size, lm.ctypes.data_as(C.c_void_p), lm.shape[0],
                                      float(feature_thresh), float(width_thresh), float(height_thresh), float(dist_thresh),
                                      int(image_w), int(image_h), int(max_new), sel.ctypes.data_as(C.c_void_p), sel.size)
    return sel[:min(n, sel.size)].copy()


def fast9_strength(gray, x, y):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    return int(_lib.orc_fast9_strength(_b(gray), w, h, w, int(x), int(y)))


def fast9_detect(gray, threshold=20, nonmax=True, max_features=16384, cap=None):
    """FAST-9/16 keypoints (first stage of cv::ORB::detect, reference src/imagefeatures.cpp:73-87), sorted by score."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    cap = max_features if cap is None else cap
    out = np.zeros(cap, dtype=FEATURE_DTYPE)
    n = _lib.orc_fast9_detect(_b(gray), w, h, w, int(threshold), int(bool(nonmax)), int(max_features),
                              out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].copy()


def orb_detect_level(gray, nfeatures, fast_threshold=20, edge=75, block=7, harris_k=0.04):
    """Keypoints of one ORB level (cv::ORB::detect, reference src/imagefeatures.cpp:73-87; parity unpinned), level coordinates."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    cap = max(4 * nfeatures + 1024, 1024)
    out = np.zeros(cap, dtype=FEATURE_DTYPE)
    n = _lib.orc_orb_detect_level(_b(gray), w, h, w, int(nfeatures), int(fast_threshold), int(edge), int(block),
                                  C.c_float(harris_k), out.ctypes.data_as(C.c_void_p), cap)
    assert n >= 0
    return out[:n].copy()


def resize_bilinear(gray, dw, dh):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    sh, sw = gray.shape
    dst = np.zeros((dh, dw), dtype=np.uint8)
    _lib.orc_resize_bilinear(_b(gray), sw, sh, sw, _b(dst), dw, dh, dw)
    return dst


def orb_level_plan(w, h, nfeatures=10000, scale_factor=1.2, nlevels=8):
    lw, lh, lf = (np.zeros(nlevels, dtype=np.int32) for _ in range(3))
    ls = np.zeros(nlevels, dtype=np.float32)
    _lib.orc_orb_level_plan(int(nfeatures), C.c_float(scale_factor), int(nlevels), int(w), int(h),
                            lw.ctypes.data_as(C.c_void_p), lh.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p),
                            ls.ctypes.data_as(C.c_void_p))
    return lw, lh, lf, ls


def orb_detect(gray, nfeatures=10000, scale_factor=1.2, nlevels=8, edge=75, fast_threshold=20, block=7, harris_k=0.04):
    """All levels: list over levels of (x float32, y float32, response float32) arrays in base-image pixels."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    lw, lh, lf, ls = orb_level_plan(w, h, nfeatures, scale_factor, nlevels)
    out, img = [], gray
    for l in range(nlevels):
        if l > 0:
            img = resize_bilinear(img, int(lw[l]), int(lh[l]))
        if lw[l] < 2 * edge + 1 or lh[l] < 2 * edge + 1:
            out.append(np.zeros(0, dtype=FEATURE_DTYPE))
            continue
        out.append(orb_detect_level(img, int(lf[l]), fast_threshold, edge, block, harris_k))
    return out, ls


_lib.orc_fast_atan2.restype = C.c_float
_lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
_lib.orc_orb_ic_angle.restype = C.c_float


def gaussian_blur7(gray):
    """7 x 7 Gaussian, sigma 2, reflect-101 borders, in the fixed point this library defines (see vslam_oracle.c)."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    dst = np.zeros((h, w), dtype=np.uint8)
    _lib.orc_gaussian_blur7(_b(gray), w, h, w, _b(dst), w)
    return dst


def fast_atan2(y, x):
    return float(_lib.orc_fast_atan2(C.c_float(y), C.c_float(x)))


def orb_default_pattern():
    pat = np.zeros(1024, dtype=np.int8)
    _lib.orc_orb_default_pattern(pat.ctypes.data_as(C.c_void_p))
    return pat


def orb_describe_level(gray, kps, pattern=None, blur=None):
    """Orientation (degrees) and 32-byte descriptors of keypoints `kps` (FEATURE_DTYPE, level coordinates) on one
    level image: cv::ORB::compute as the reference calls it (src/imagefeatures.cpp:88-96), restated; parity unpinned."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    blur = gaussian_blur7(gray) if blur is None else np.ascontiguousarray(blur, dtype=np.uint8)
    pattern = orb_default_pattern() if pattern is None else np.ascontiguousarray(pattern, dtype=np.int8)
    n = len(kps)
    angle = np.zeros(n, dtype=np.float32)
    desc = np.zeros((n, 32), dtype=np.uint8)
    for i in range(n):
        x, y = int(kps["x"][i]), int(kps["y"][i])
        angle[i] = _lib.orc_orb_ic_angle(_b(gray), w, x, y, None, None)
        _lib.orc_orb_describe(_b(blur), w, x, y, C.c_float(angle[i]), pattern.ctypes.data_as(C.c_void_p),
                              desc[i].ctypes.data_as(C.c_void_p))
    return angle, desc


def orb_detect_and_compute(gray, nfeatures=10000, scale_factor=1.2, nlevels=8, edge=75, fast_threshold=20, block=7, harris_k=0.04,
                           pattern=None):
    """orb_detect plus orientation and descriptors: lists over levels of keypoints, angles, descriptors; scales."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    lw, lh, lf, ls = orb_level_plan(w, h, nfeatures, scale_factor, nlevels)
    kps, angles, descs, img = [], [], [], gray
    for l in range(nlevels):
        if l > 0:
            img = resize_bilinear(img, int(lw[l]), int(lh[l]))
        if lw[l] < 2 * edge + 1 or lh[l] < 2 * edge + 1:
            kps.append(np.zeros(0, dtype=FEATURE_DTYPE)); angles.append(np.zeros(0, dtype=np.float32))
            descs.append(np.zeros((0, 32), dtype=np.uint8))
            continue
        k = orb_detect_level(img, int(lf[l]), fast_threshold, edge, block, harris_k)
        a, d = orb_describe_level(img, k, pattern)
        kps.append(k); angles.append(a); descs.append(d)
    return kps, angles, descs, ls


def normcdf(z):
    return float(_lib.orc_normcdf(float(z)))


def chi2inv(p, nu):
    return float(_lib.orc_chi2inv(float(p), float(nu)))


def individual_compat(i, j, y, muY, SYY, chi2LUT):
    """y: 2 x m (features in columns)."""
    y, muY, SYY, lut = _F(y), _f64(muY), _F(SYY), _f64(chi2LUT)
    return bool(_lib.orc_individual_compat(int(i), int(j), y.shape[0], _d(y), _d(muY), _d(SYY), SYY.shape[0], _d(lut)))


def joint_compat(idx, sU, y, muY, SYY, chi2LUT):
    y, muY, SYY, lut = _F(y), _f64(muY), _F(SYY), _f64(chi2LUT)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    s = C.c_double(0.0)
    ok = _lib.orc_joint_compat(_i(idx), idx.size, float(sU), y.shape[0], _d(y), _d(muY), _d(SYY), _d(lut), C.byref(s))
    return bool(ok), s.value


def snn_density(muY, SYY, y, width, height, enforce_jc=False):
    y, muY, SYY = _F(y), _f64(muY), _F(SYY)
    n = muY.size // 2
    idx = np.zeros(n, dtype=np.int32)
    s = _lib.orc_snn_density(_d(muY), _d(SYY), n, _d(y), y.shape[1], int(width), int(height), int(enforce_jc), _i(idx))
    return float(s), idx


def snn(mux, Sxx, y, cam, layout=0, enforce_jc=False):
    mux, Sxx, y = _f64(mux), _F(Sxx), _F(y)
    n = (mux.size - 12) // 3
    idx = np.zeros(n, dtype=np.int32)
    s = _lib.orc_snn(_d(mux), _d(Sxx), mux.size, _d(y), y.shape[1], C.byref(cam), int(layout), int(enforce_jc), _i(idx))
    return float(s), idx
