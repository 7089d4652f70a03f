"""Thin Python surface over the C ABI: numpy arrays go through the host-pointer
entry points, torch CUDA tensors through the device-pointer (_dev) ones.

Matrices are column-major float64 (numpy order='F'), the layout of
Eigen::MatrixXd in the reference.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Camera, CornerParams, OrbParams, SelectParams, VslamError, check  # noqa: F401

FEATURE_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("score", np.float32)])
LAYOUT_COMPAT, LAYOUT_STRICT = 0, 1
HARRIS, SHI_TOMASI = 0, 1


TRS_EIG, TRS_CHOLESKY = 0, 1


class Context:
    """One device context (one HIP stream).  stream: a raw hipStream_t handle
    (e.g. torch.cuda.current_stream().cuda_stream) to borrow, or None to own one."""

    def __init__(self, device=0, stream=None):
        self._L = _lib.lib()
        self.ptr = C.c_void_p()
        if stream is None:
            st = self._L.vslam_ctx_create(device, C.byref(self.ptr))
        else:
            st = self._L.vslam_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(self.ptr))
        if st != 0:
            raise VslamError("cannot create a context on device %d: %s (status %d); libvslam_hip needs a gfx950 GPU"
                             % (device, self._L.vslam_status_string(st).decode(), st))
        self.device = device

    def set_trs_solver(self, solver):
        """TRS_CHOLESKY (default: device route, eigenvalue fallback) or TRS_EIG (the reference's host eigenvalue route)."""
        check(self._L.vslam_ctx_set_trs_solver(self.ptr, int(solver)), self.ptr)

    def last_trs_route(self):
        return int(self._L.vslam_ctx_last_trs_route(self.ptr))

    def last_update_profile(self):
        """Wall-clock split of the last device-route measurement update (dict, ms and counts)."""
        out = np.zeros(8)
        check(self._L.vslam_ctx_last_update_profile(self.ptr, out.ctypes.data), self.ptr)
        return {"cost_evaluations_ms": out[0], "trust_region_ms": out[1], "posterior_factor_ms": out[2], "device_part_ms": out[3],
                "cost_evaluations": int(out[4]), "factorisations": int(out[5]), "iterations": int(out[6])}

    def synchronize(self):
        check(self._L.vslam_ctx_synchronize(self.ptr), self.ptr)

    def set_kernel_timing(self, on):
        """Measurement aid: events around the two kernels of the batched point log-likelihood."""
        check(self._L.vslam_ctx_set_kernel_timing(self.ptr, int(bool(on))), self.ptr)

    def last_point_loglik_times(self):
        """(terms_ms, assemble_ms) of the last timed two-kernel point_loglik_batch_dev call; waits for it."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        check(self._L.vslam_ctx_last_point_loglik_times(self.ptr, C.byref(a), C.byref(b)), self.ptr)
        return a.value, b.value

    def close(self):
        if self.ptr:
            self._L.vslam_ctx_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_from_xml(path):
    cam = Camera()
    check(_lib.lib().vslam_camera_from_xml(str(path).encode(), C.byref(cam)))
    return cam


def read_pgm(path):
    """A binary PGM (P5, 8-bit) as an (h, w) uint8 array, through the library's reader."""
    L = _lib.lib()
    dims = np.zeros(3, dtype=np.int32)
    check(L.vslam_pgm_info(str(path).encode(), dims.ctypes.data, dims.ctypes.data + 4, dims.ctypes.data + 8))
    gray = np.zeros((int(dims[1]), int(dims[0])), dtype=np.uint8)
    check(L.vslam_pgm_read(str(path).encode(), gray.ctypes.data, gray.size, dims.ctypes.data, dims.ctypes.data + 4))
    return gray


def write_pgm(path, gray):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    check(_lib.lib().vslam_pgm_write(str(path).encode(), gray.ctypes.data, gray.shape[1], gray.shape[0], gray.shape[1]))


def corner_params(block_size=7, border=None, max_features=10000, harris_k=0.04,
                  harris_thresh=1e-4, shi_thresh=1e-3):
    p = CornerParams()
    p.block_size = block_size
    p.border = (block_size // 2 + 2) if border is None else border
    p.max_features = max_features
    p.harris_k, p.harris_thresh, p.shi_thresh = harris_k, harris_thresh, shi_thresh
    return p


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _F(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _tp(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _tpc(t):
    """Device pointer of a tensor that the C ABI addresses as one dense block (no stride arguments)."""
    if t is not None and not t.is_contiguous():
        raise ValueError("libvslam_hip needs a contiguous tensor here (got strides %s for shape %s)" % (tuple(t.stride()), tuple(t.shape)))
    return _tp(t)


# ------------------------------------------------------------------ host-pointer calls
def project(ctx, cam, x, j0=0, L=None):
    x = _f64(x)
    n = x.size
    L = (n - 12) // 3 - j0 if L is None else L
    px = np.zeros(2 * L)
    code = np.zeros(L, dtype=np.int32)
    check(ctx._L.vslam_project(ctx.ptr, C.byref(cam), _p(x), n, j0, L, _p(px), _p(code)), ctx.ptr)
    return px, code


def bundle_measure(ctx, cam, x, layout=LAYOUT_COMPAT, want_SR=True, want_C=True):
    x = _f64(x)
    n = x.size
    L = (n - 12) // 3
    h = np.zeros(2 * L)
    SR = np.zeros((2 * L, 2 * L), order="F") if want_SR else None
    Cm = np.zeros((2 * L, n), order="F") if want_C else None
    code = np.zeros(L, dtype=np.int32)
    check(ctx._L.vslam_bundle_measure(ctx.ptr, C.byref(cam), _p(x), n, layout, _p(h), _p(SR), _p(Cm), _p(code)), ctx.ptr)
    return h, SR, Cm, code


def bundle_measure_dev(ctx, cam, x, h, SR=None, Cm=None, code=None, layout=LAYOUT_COMPAT):
    """Device-resident bundle evaluation: x (n), h (2L), SR (2L x 2L, column-major), Cm (2L x n, column-major) and
    code (L, int32) are torch tensors on the context's device; SR, Cm and code may be None."""
    n = x.numel()
    check(ctx._L.vslam_bundle_measure_dev(ctx.ptr, C.byref(cam), _tp(x), n, layout, _tp(h), _tp(SR), _tp(Cm), _tp(code)), ctx.ptr)


def single_measure(ctx, cam, j, x, layout=LAYOUT_COMPAT, want_SR=True, want_J=True):
    x = _f64(x)
    n = x.size
    h = np.zeros(2)
    SR = np.zeros((2, 2), order="F") if want_SR else None
    J = np.zeros((2, n), order="F") if want_J else None
    code = np.zeros(1, dtype=np.int32)
    check(ctx._L.vslam_single_measure(ctx.ptr, C.byref(cam), j, _p(x), n, layout, _p(h), _p(SR), _p(J), _p(code)), ctx.ptr)
    return int(code[0]), h, SR, J


def point_loglik(ctx, cam, x, y, visible, sigma, want_g=True, want_H=True, ldH=None):
    x, y = _f64(x), _f64(y)
    n = x.size
    vis = np.ascontiguousarray(visible, dtype=np.uint8)
    val = np.zeros(1)
    g = np.zeros(n) if (want_g or want_H) else None
    ld = n if ldH is None else ldH
    H = np.zeros((ld, n), order="F") if want_H else None
    code = np.zeros((n - 12) // 3, dtype=np.int32)
    check(ctx._L.vslam_point_loglik(ctx.ptr, C.byref(cam), _p(x), n, _p(y), _p(vis), float(sigma),
                                    _p(val), _p(g), _p(H), ld, _p(code)), ctx.ptr)
    return float(val[0]), g, H, code


def log_gaussian(ctx, x, mu, S, want_g=True, want_H=True):
    x, mu, S = _f64(x), _f64(mu), _F(S)
    n = x.size
    val = np.zeros(1)
    g = np.zeros(n) if want_g or want_H else None
    H = np.zeros((n, n), order="F") if want_H else None
    check(ctx._L.vslam_log_gaussian(ctx.ptr, _p(x), _p(mu), _p(S), n, S.shape[0], _p(val), _p(g), _p(H)), ctx.ptr)
    return float(val[0]), g, H


def qr_upper(ctx, A):
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    check(ctx._L.vslam_qr_upper(ctx.ptr, _p(A), A.shape[0], A.shape[1], A.shape[0]), ctx.ptr)
    return A


def affine_sqrt(ctx, S, Cm, SR):
    S, Cm, SR = _F(S), _F(Cm), _F(SR)
    n, m = S.shape[0], Cm.shape[0]
    Syy = np.zeros((m, m), order="F")
    check(ctx._L.vslam_affine_sqrt(ctx.ptr, _p(S), n, n, _p(Cm), m, m, _p(SR), m, _p(Syy), m), ctx.ptr)
    return Syy


def affine_bundle(ctx, cam, mux, Sxx, layout=LAYOUT_COMPAT):
    mux, Sxx = _f64(mux), _F(Sxx)
    n = mux.size
    L = (n - 12) // 3
    muy = np.zeros(2 * L)
    Syy = np.zeros((2 * L, 2 * L), order="F")
    code = np.zeros(L, dtype=np.int32)
    check(ctx._L.vslam_affine_bundle(ctx.ptr, C.byref(cam), _p(mux), _p(Sxx), n, n, layout,
                                     _p(muy), _p(Syy), 2 * L, _p(code)), ctx.ptr)
    return muy, Syy, code


def iekf_post_sqrt(ctx, Q, v):
    Q, v = _F(Q), _f64(v)
    n = v.size
    S = np.zeros((n, n), order="F")
    check(ctx._L.vslam_iekf_post_sqrt(ctx.ptr, _p(Q), _p(v), n, _p(S)), ctx.ptr)
    return S


def condition_on_marginal(ctx, muyx, Syx, y):
    muyx, Syx, y = _f64(muyx), _F(Syx), _f64(y)
    ny = y.size
    nx = muyx.size - ny
    mu = np.zeros(nx)
    Sx = np.zeros((nx, nx), order="F")
    check(ctx._L.vslam_condition_on_marginal(ctx.ptr, _p(muyx), _p(Syx), ny, nx, ny + nx, _p(y), _p(mu), _p(Sx), nx), ctx.ptr)
    return mu, Sx


def time_update(ctx, mu, S, pos_tune, ori_tune, dt):
    """timeUpdateContinuous with SlamProcessModel (reference src/gaussian.hpp:337-383): returns (mu+, S+)."""
    mu, S, pt, ot = _f64(mu), _F(S), _F(pos_tune), _F(ori_tune)
    n = mu.size
    mup = np.zeros(n)
    Sp = np.zeros((n, n), order="F")
    check(ctx._L.vslam_time_update(ctx.ptr, _p(mu), _p(S), n, n, _p(pt), _p(ot), float(dt), _p(mup), _p(Sp), n), ctx.ptr)
    return mup, Sp


def remove_landmark(ctx, mu, S, j):
    """removeBadLandmarks (reference src/SLAM.cpp:796-820): returns (mu, S) without landmark j."""
    mu, S = _f64(mu), _F(S)
    n = mu.size
    mo = np.zeros(max(n - 3, 0))
    So = np.zeros((max(n - 3, 0), max(n - 3, 0)), order="F")
    check(ctx._L.vslam_remove_landmark(ctx.ptr, _p(mu), _p(S), n, n, int(j), _p(mo), _p(So), max(n - 3, 1)), ctx.ptr)
    return mo, So


def match_hamming(ctx, query, train):
    """cv::BFMatcher(NORM_HAMMING).match (reference src/SLAM.cpp:355): (train_idx, distance) per query row."""
    query = np.ascontiguousarray(query, dtype=np.uint8)
    nq, nb = query.shape
    train = np.ascontiguousarray(train, dtype=np.uint8).reshape(-1, nb)
    idx = np.zeros(nq, np.int32)
    dist = np.zeros(nq, np.int32)
    check(ctx._L.vslam_match_hamming(ctx.ptr, _p(query), nq, _p(train), train.shape[0], nb, _p(idx), _p(dist)), ctx.ptr)
    return idx, dist


def cost_joint_density(ctx, cam, x, mu, S, y, visible, sigma):
    x, mu, S, y = _f64(x), _f64(mu), _F(S), _f64(y)
    n = x.size
    vis = np.ascontiguousarray(visible, dtype=np.uint8)
    val = np.zeros(1)
    g = np.zeros(n)
    H = np.zeros((n, n), order="F")
    check(ctx._L.vslam_cost_joint_density(ctx.ptr, C.byref(cam), _p(x), _p(mu), _p(S), n, n, _p(y), _p(vis),
                                          float(sigma), _p(val), _p(g), _p(H)), ctx.ptr)
    return float(val[0]), g, H


def measurement_update_iekf(ctx, cam, mu, S, y, visible, sigma):
    mu, S, y = _f64(mu), _F(S), _f64(y)
    n = mu.size
    vis = np.ascontiguousarray(visible, dtype=np.uint8)
    mup = np.zeros(n)
    Sp = np.zeros((n, n), order="F")
    info = np.zeros(3, dtype=np.int32)
    check(ctx._L.vslam_measurement_update_iekf(ctx.ptr, C.byref(cam), _p(mu), _p(S), n, n, _p(y), _p(vis), float(sigma),
                                               _p(mup), _p(Sp), n, C.c_void_p(info.ctypes.data),
                                               C.c_void_p(info.ctypes.data + 4), C.c_void_p(info.ctypes.data + 8)), ctx.ptr)
    return int(info[0]), mup, Sp, int(info[1]), int(info[2])


def measurement_update_iekf_batch_dev(ctx, cam, mu, S, y, visible, sigma, mu_post, S_post):
    """Bank of filters on the GPU (torch tensors): mu (B, n) f64, S (B, n, n) holding each column-major
    factor, y (B, 2L) f64, visible (B, L) u8; mu_post / S_post alike.  Returns (fmin_code, iters, status)
    int32 arrays of B entries; status 1 marks a filter to redo with measurement_update_iekf."""
    B, n = mu.shape
    code = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    status = np.zeros(B, dtype=np.int32)
    check(ctx._L.vslam_measurement_update_iekf_batch_dev(ctx.ptr, C.byref(cam), _tpc(mu), _tpc(S), n, B, _tpc(y), _tpc(visible),
                                                         float(sigma), _tpc(mu_post), _tpc(S_post), _p(code), _p(iters),
                                                         _p(status)), ctx.ptr)
    return code, iters, status


def normcdf(z):
    return float(_lib.lib().vslam_normcdf(float(z)))


def chi2inv(p, nu):
    return float(_lib.lib().vslam_chi2inv(float(p), float(nu)))


def individual_compat(ctx, y, muY, SYY, chi2LUT):
    """All (feature i, landmark j) pairs: bool array (m, n).  y is 2 x m (features in columns)."""
    y, muY, SYY, lut = _F(y), _f64(muY), _F(SYY), _f64(chi2LUT)
    m, n = y.shape[1], muY.size // 2
    out = np.zeros((m, n), dtype=np.uint8)
    check(ctx._L.vslam_individual_compat(ctx.ptr, _p(y), m, _p(muY), _p(SYY), n, SYY.shape[0], _p(lut), _p(out)), ctx.ptr)
    return out.astype(bool)


def joint_compat(ctx, idx, sU, y, muY, SYY, chi2LUT):
    y, muY, SYY, lut = _F(y), _f64(muY), _F(SYY), _f64(chi2LUT)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    s = np.zeros(1)
    jc = np.zeros(1, dtype=np.int32)
    check(ctx._L.vslam_joint_compat(ctx.ptr, _p(idx), idx.size, float(sU), _p(y), y.shape[1], _p(muY), _p(SYY), SYY.shape[0],
                                    _p(lut), _p(s), _p(jc)), ctx.ptr)
    return bool(jc[0]), float(s[0])


def snn_density(ctx, muY, SYY, y, width, height, enforce_jc=False):
    y, muY, SYY = _F(y), _f64(muY), _F(SYY)
    n = muY.size // 2
    idx = np.zeros(n, dtype=np.int32)
    s = np.zeros(1)
    check(ctx._L.vslam_snn_density(ctx.ptr, _p(muY), _p(SYY), n, SYY.shape[0], _p(y), y.shape[1], int(width), int(height),
                                   int(enforce_jc), _p(idx), _p(s)), ctx.ptr)
    return float(s[0]), idx


def snn(ctx, cam, mux, Sxx, y, layout=LAYOUT_COMPAT, enforce_jc=False):
    mux, Sxx, y = _f64(mux), _F(Sxx), _F(y)
    n = (mux.size - 12) // 3
    idx = np.zeros(n, dtype=np.int32)
    s = np.zeros(1)
    check(ctx._L.vslam_snn(ctx.ptr, C.byref(cam), _p(mux), _p(Sxx), mux.size, mux.size, layout, _p(y), y.shape[1],
                           int(enforce_jc), _p(idx), _p(s)), ctx.ptr)
    return float(s[0]), idx


def corner_detect(ctx, gray, params, detector=HARRIS, cap=None):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    cap = params.max_features if cap is None else cap
    out = np.zeros(cap, dtype=FEATURE_DTYPE)
    cnt = np.zeros(1, dtype=np.int32)
    check(ctx._L.vslam_corner_detect(ctx.ptr, _p(gray), w, h, w, C.byref(params), detector, _p(out), cap, _p(cnt)), ctx.ptr)
    return out[:int(cnt[0])].copy()


def select_params(feature_thresh=1e-4, width_thresh=250.0, height_thresh=100.0, pixel_distance_thresh=150.0,
                  image_width=1920, image_height=1080, max_new=50):
    """Defaults: the reference's scenario-2 tuning (src/SLAM.cpp:113-120)."""
    p = SelectParams()
    p.feature_thresh, p.width_thresh, p.height_thresh = feature_thresh, width_thresh, height_thresh
    p.pixel_distance_thresh = pixel_distance_thresh
    p.image_width, p.image_height, p.max_new = image_width, image_height, max_new
    return p


def pixel_distance(landmark_xy, x, y, thresh):
    """pixelDistance (reference src/SLAM.cpp:746-760), host function."""
    lm = np.ascontiguousarray(landmark_xy, dtype=np.float32).reshape(-1, 2)
    return bool(_lib.lib().vslam_pixel_distance(_p(lm) if lm.size else None, lm.shape[0], float(x), float(y), float(thresh)))


def select_new_landmarks(ctx, feats, landmark_xy, params, cap=None):
    """The selection loop of new landmarks (reference src/SLAM.cpp:449-472): indices into feats, in order."""
    feats = np.ascontiguousarray(feats, dtype=FEATURE_DTYPE)
    lm = np.ascontiguousarray(landmark_xy, dtype=np.float32).reshape(-1, 2)
    cap = max(int(params.max_new), 1) if cap is None else cap
    sel = np.zeros(cap, dtype=np.int32)
    cnt = np.zeros(1, dtype=np.int32)
    check(ctx._L.vslam_select_new_landmarks(ctx.ptr, _p(feats) if feats.size else None, feats.size, _p(lm) if lm.size else None,
                                            lm.shape[0], C.byref(params), _p(sel), cap, _p(cnt)), ctx.ptr)
    return sel[:min(int(cnt[0]), cap)].copy()


def select_new_landmarks_batch_dev(ctx, feats, k_each, K, landmark_xy, l_each, L, params, selected, count):
    """torch tensors on the GPU: feats (B, K, 3) int32 records, k_each (B,) int32 or None, landmark_xy (B, L, 2) f32,
    l_each (B,) int32 or None, selected (B, cap) int32, count (B,) int32."""
    B, cap = selected.shape
    check(ctx._L.vslam_select_new_landmarks_batch_dev(ctx.ptr, _tpc(feats), K, _tpc(k_each), K, _tpc(landmark_xy), 2 * L, _tpc(l_each), L,
                                                      C.byref(params), B, _tpc(selected), cap, _tpc(count)), ctx.ptr)


def optical_ray(cam, u, v, ray_length=8.0):
    """generateOpticalRay (reference src/SLAM.cpp:876-903), host function: ray_length * K^-1 [u' v' 1]^T."""
    r = np.zeros(3)
    if _lib.lib().vslam_optical_ray(C.byref(cam), float(u), float(v), float(ray_length), _p(r)) != 0:
        raise _lib.VslamError("libvslam_hip: optical_ray: camera constants or ray length are not usable")
    return r


def initialize_landmarks(ctx, cam, mu, S, pixels, ray_length=8.0, kappa=0.5):
    """initializeNewLandmark (reference src/SLAM.cpp:905-941) for the keypoints pixels (M, 2): (mu, S) grown by 3M states.
    Defaults: optical_ray_length and kappa of the reference's scenario 2 (src/SLAM.cpp:122-123)."""
    mu, S = _f64(mu), _F(S)
    px = np.ascontiguousarray(pixels, dtype=np.float64).reshape(-1, 2)
    n, M = mu.size, px.shape[0]
    m = n + 3 * M
    mo, So = np.zeros(m), np.zeros((m, m), order="F")
    check(ctx._L.vslam_initialize_landmarks(ctx.ptr, C.byref(cam), _p(mu), _p(S), n, S.shape[0], _p(px), M, float(ray_length), float(kappa),
                                            _p(mo), _p(So), m), ctx.ptr)
    return mo, So


def initialize_landmarks_dev(ctx, cam, mu, S, pixels, mu_out, S_out, ray_length=8.0, kappa=0.5):
    """torch tensors on the GPU: mu (n,), S (n x n, column-major: element (r, c) at c * n + r), pixels (M, 2) f64,
    mu_out (n + 3M,), S_out ((n + 3M) x (n + 3M), column-major)."""
    n, M = mu.numel(), pixels.shape[0]
    m = n + 3 * M
    check(ctx._L.vslam_initialize_landmarks_dev(ctx.ptr, C.byref(cam), _tpc(mu), _tpc(S), n, n, _tpc(pixels), M, float(ray_length),
                                                float(kappa), _tpc(mu_out), _tpc(S_out), m), ctx.ptr)


def initialize_landmarks_batch_dev(ctx, cam, mu, S, pixels, mu_out, S_out, ray_length=8.0, kappa=0.5):
    """A bank of filters, torch tensors on the GPU: mu (B, n), S (B, n, n) each column-major, pixels (B, M, 2) f64,
    mu_out (B, n + 3M), S_out (B, n + 3M, n + 3M) each column-major."""
    B, n = mu.shape
    M = pixels.shape[1]
    check(ctx._L.vslam_initialize_landmarks_batch_dev(ctx.ptr, C.byref(cam), _tpc(mu), _tpc(S), n, B, _tpc(pixels), M, float(ray_length),
                                                      float(kappa), _tpc(mu_out), _tpc(S_out)), ctx.ptr)


def fast9_detect(ctx, gray, threshold=20, nonmax=True, max_features=16384, cap=None, stride=None):
    """FAST-9/16 keypoints (first stage of cv::ORB::detect, reference src/imagefeatures.cpp:73-87), sorted by score."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    if stride is not None:                      # image is a window of `w_used` columns inside rows of `stride` bytes
        w_used, row_bytes = stride
        w, stride = w_used, row_bytes
    else:
        stride = w
    cap = max_features if cap is None else cap
    out = np.zeros(cap, dtype=FEATURE_DTYPE)
    cnt = np.zeros(1, dtype=np.int32)
    check(ctx._L.vslam_fast9_detect(ctx.ptr, _p(gray), w, h, stride, int(threshold), int(bool(nonmax)), int(max_features),
                                    _p(out), cap, _p(cnt)), ctx.ptr)
    return out[:int(cnt[0])].copy()


ORB_KEYPOINT_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("response", np.float32), ("octave", np.int32)])


def orb_params(nfeatures=10000, scale_factor=1.2, nlevels=8, edge_threshold=75, fast_threshold=20, harris_block=7, harris_k=0.04):
    """cv::ORB::create(nfeatures, 1.2f, 8, 75, 0, 2, HARRIS_SCORE, 75, 20) as at reference src/imagefeatures.cpp:73-82."""
    return OrbParams(int(nfeatures), float(scale_factor), int(nlevels), int(edge_threshold), int(fast_threshold), int(harris_block),
                     float(harris_k), 0)


def orb_level_plan(params, w, h):
    nl = params.nlevels
    lw, lh, lf = (np.zeros(nl, dtype=np.int32) for _ in range(3))
    ls = np.zeros(nl, dtype=np.float32)
    check(_lib.lib().vslam_orb_level_plan(C.byref(params), int(w), int(h), _p(lw), _p(lh), _p(lf), _p(ls)))
    return lw, lh, lf, ls


def orb_detect(ctx, gray, params=None, cap=32768):
    """ORB keypoints of all pyramid levels (no orientation, no descriptors): (keypoints, per-level counts)."""
    params = orb_params() if params is None else params
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    out = np.zeros(cap, dtype=ORB_KEYPOINT_DTYPE)
    cnt = np.zeros(1, dtype=np.int32)
    lev = np.zeros(params.nlevels, dtype=np.int32)
    check(ctx._L.vslam_orb_detect(ctx.ptr, _p(gray), w, h, w, C.byref(params), _p(out), cap, _p(cnt), _p(lev)), ctx.ptr)
    return out[:int(cnt[0])].copy(), lev


def orb_default_pattern():
    """The library's own 256 x (x0, y0, x1, y1) test pattern (int8[1024]); cv::ORB's learned table is a caller's input."""
    pat = np.zeros(1024, dtype=np.int8)
    check(_lib.lib().vslam_orb_default_pattern(_p(pat)))
    return pat


def orb_detect_and_compute(ctx, gray, params=None, pattern=None, cap=32768):
    """ORB keypoints of all pyramid levels with orientation (degrees) and 32-byte descriptors:
    (keypoints, angles, descriptors, per-level counts)."""
    params = orb_params() if params is None else params
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    out = np.zeros(cap, dtype=ORB_KEYPOINT_DTYPE)
    angle = np.zeros(cap, dtype=np.float32)
    desc = np.zeros((cap, 32), dtype=np.uint8)
    cnt = np.zeros(1, dtype=np.int32)
    lev = np.zeros(params.nlevels, dtype=np.int32)
    pat = None if pattern is None else np.ascontiguousarray(pattern, dtype=np.int8)
    check(ctx._L.vslam_orb_detect_and_compute(ctx.ptr, _p(gray), w, h, w, C.byref(params), _p(pat), _p(out), _p(angle), _p(desc), cap,
                                              _p(cnt), _p(lev)), ctx.ptr)
    k = int(cnt[0])
    return out[:k].copy(), angle[:k].copy(), desc[:k].copy(), lev


def orb_detect_and_compute_batch_dev(ctx, frames, out, count, params=None, pattern=None, angle=None, desc=None, level_counts=None):
    """The whole ORB front end for device-resident frames: frames (F, h, w) uint8; out (F, cap, 4) int32 holding the
    ORB_KEYPOINT records (x, y, response as float bits, octave); count (F,) int32; optional angle (F, cap) float32 with
    desc (F, cap, 32) uint8, level_counts (F, nlevels) int32, pattern int8[1024] on the device.  Queued, not synchronised."""
    params = orb_params() if params is None else params
    F, h, w = frames.shape
    check(ctx._L.vslam_orb_detect_and_compute_batch_dev(ctx.ptr, _tp(frames), w, h, frames.stride(1), frames.stride(0), F, C.byref(params),
                                                        _tp(pattern), _tp(out), _tp(angle), _tp(desc), out.shape[1], _tp(count),
                                                        _tp(level_counts)), ctx.ptr)


def gaussian_blur7_dev(ctx, src, dst):
    """src and dst (h, w) uint8 torch tensors on the GPU, rows contiguous."""
    check(ctx._L.vslam_gaussian_blur7_dev(ctx.ptr, _tp(src), src.shape[1], src.shape[0], src.stride(0), _tp(dst), dst.stride(0)), ctx.ptr)


def orb_describe_level_dev(ctx, gray, blurred, keypoints, count, max_count, pattern, angle, desc):
    """Orientation and descriptors of one level's keypoints; every argument a torch tensor on the GPU (gray, blurred
    (h, w) uint8; keypoints a FEATURE record buffer; count int32[1]; pattern int8[1024]; angle float32; desc uint8 (., 32))."""
    check(ctx._L.vslam_orb_describe_level_dev(ctx.ptr, _tp(gray), gray.stride(0), _tp(blurred), blurred.stride(0), gray.shape[1], gray.shape[0],
                                              _tp(keypoints), _tp(count), int(max_count), _tp(pattern), _tp(angle), _tp(desc)), ctx.ptr)


def orb_detect_level_batch_dev(ctx, gray, w, h, stride, stride_frame, frames, nfeatures, out, cap, count,
                               fast_threshold=20, edge_threshold=75, harris_block=7, harris_k=0.04):
    check(ctx._L.vslam_orb_detect_level_batch_dev(ctx.ptr, _tp(gray), w, h, stride, stride_frame, frames, int(nfeatures), int(fast_threshold),
                                                  int(edge_threshold), int(harris_block), float(harris_k), _tp(out), cap, _tp(count)), ctx.ptr)


def resize_bilinear_dev(ctx, src, dst):
    """src (sh, sw) and dst (dh, dw) uint8 torch tensors on the GPU, rows contiguous."""
    check(ctx._L.vslam_resize_bilinear_dev(ctx.ptr, _tp(src), src.shape[1], src.shape[0], src.stride(0), _tp(dst), dst.shape[1],
                                           dst.shape[0], dst.stride(0)), ctx.ptr)


def corner_response(ctx, gray, params):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    har = np.zeros((h, w), dtype=np.float32)
    shi = np.zeros((h, w), dtype=np.float32)
    check(ctx._L.vslam_corner_response(ctx.ptr, _p(gray), w, h, w, C.byref(params), _p(har), _p(shi)), ctx.ptr)
    return har, shi


# ------------------------------------------------------------------ device-pointer calls (torch tensors)
def point_loglik_batch_dev(ctx, cam, n, x, y, visible, sigma, value, g=None, H=None, code=None, ldH=None, noalloc=False):
    """x (B,n) f64, y (B,2L) f64, visible (B,L) u8, value (B,) f64, g (B,n) f64,
    H (B,n,ldH) f64 holding each instance column-major, code (B,L) i32: torch
    CUDA tensors, contiguous.  Queued on the context's stream, not synchronised.
    noalloc=True takes the entry point that allocates nothing (static device scratch,
    at most 81920 landmarks per call)."""
    B = x.shape[0]
    ld = n if ldH is None else ldH
    fn = ctx._L.vslam_point_loglik_batch_dev_noalloc if noalloc else ctx._L.vslam_point_loglik_batch_dev
    check(fn(
        ctx.ptr, C.byref(cam), B, n, _tp(x), x.stride(0), _tp(y), y.stride(0), _tp(visible), visible.stride(0),
        float(sigma), _tp(value), _tp(g), 0 if g is None else g.stride(0),
        _tp(H), ld, 0 if H is None else H.stride(0), _tp(code)), ctx.ptr)


def qr_upper_batch_dev(ctx, A, rows, cols, lda, batch, stride):
    check(ctx._L.vslam_qr_upper_batch_dev(ctx.ptr, _tp(A), rows, cols, lda, stride, batch), ctx.ptr)


def snn_density_batch_dev(ctx, muY, SYY, y, n, m, width, height, idx, surprisal, enforce_jc=False):
    """torch tensors on the GPU: muY (B, 2n) f64, SYY (B, 2n, 2n) column-major per instance (i.e. the transpose
    of the row-major view), y (B, m, 2) = column-major 2 x m, idx (B, n) int32, surprisal (B,) f64."""
    B, N = muY.shape[0], 2 * n
    check(ctx._L.vslam_snn_density_batch_dev(ctx.ptr, B, _tp(muY), N, _tp(SYY), N, N * N, n, _tp(y), 2 * m, m, int(width),
                                             int(height), int(enforce_jc), _tp(idx), _tp(surprisal)), ctx.ptr)


def time_update_batch_dev(ctx, mu, S, n, pos_tune, ori_tune, dt, mu_out, S_out, s_is_upper=True):
    """torch tensors on the GPU: mu (B, n) f64, S (B, n, n) column-major per instance; outputs alike (may alias).
    s_is_upper: every S is upper triangular (structured kernels); False takes the dense QR."""
    pt, ot = _F(pos_tune), _F(ori_tune)
    check(ctx._L.vslam_time_update_batch_dev(ctx.ptr, _tp(mu), _tp(S), n, mu.shape[0], _p(pt), _p(ot), float(dt),
                                             _tp(mu_out), _tp(S_out), int(bool(s_is_upper))), ctx.ptr)


def remove_landmark_dev(ctx, mu, S, j, mu_out, S_out, s_is_upper=True):
    """torch f64 tensors on the GPU: mu (n), S column-major n x n (i.e. the tensor holds S^T), mu_out (n-3),
    S_out column-major (n-3) x (n-3); outputs must not overlap the inputs."""
    n = mu.shape[0]
    check(ctx._L.vslam_remove_landmark_dev(ctx.ptr, _tpc(mu), _tpc(S), n, n, int(j), _tpc(mu_out), _tpc(S_out), max(n - 3, 1),
                                           int(bool(s_is_upper))), ctx.ptr)


def affine_bundle_batch_dev(ctx, cam, mux, Sxx, muy, Syy, code=None, layout=LAYOUT_COMPAT):
    """A bank of filters, torch tensors on the GPU: mux (B, n), Sxx (B, n, n) each column-major, muy (B, 2L),
    Syy (B, 2L, 2L) each column-major, code (B, L) int32 or None."""
    B, n = mux.shape
    check(ctx._L.vslam_affine_bundle_batch_dev(ctx.ptr, C.byref(cam), _tpc(mux), _tpc(Sxx), n, B, layout, _tpc(muy), _tpc(Syy), _tpc(code)),
          ctx.ptr)


def remove_landmark_batch_dev(ctx, mu, S, j_each, mu_out, S_out, status=None):
    """A bank of filters with upper-triangular factors, torch tensors on the GPU: mu (B, n), S (B, n, n) each column-major,
    j_each (B,) int32, mu_out (B, n-3), S_out (B, n-3, n-3) each column-major, status (B,) int32 or None."""
    B, n = mu.shape
    check(ctx._L.vslam_remove_landmark_batch_dev(ctx.ptr, _tpc(mu), _tpc(S), n, B, _tpc(j_each), _tpc(mu_out), _tpc(S_out), _tpc(status)),
          ctx.ptr)


def match_hamming_batch_dev(ctx, query, train, idx, dist=None, nt_each=None):
    """torch u8 tensors on the GPU: query (B, nq, bytes), train (B, nt, bytes); idx/dist (B, nq) int32."""
    B, nq, nb = query.shape
    nt = train.shape[1]
    check(ctx._L.vslam_match_hamming_batch_dev(ctx.ptr, _tp(query), nq * nb, nq, _tp(train), nt * nb, nt,
                                               _tp(nt_each) if nt_each is not None else None, nb, B, _tp(idx),
                                               _tp(dist) if dist is not None else None), ctx.ptr)


def fast9_detect_batch_dev(ctx, gray, w, h, stride, stride_frame, frames, threshold, nonmax, max_features, out, cap, count):
    check(ctx._L.vslam_fast9_detect_batch_dev(ctx.ptr, _tp(gray), w, h, stride, stride_frame, frames, int(threshold),
                                              int(bool(nonmax)), int(max_features), _tp(out), cap, _tp(count)), ctx.ptr)


def corner_detect_batch_dev(ctx, gray, w, h, stride, stride_frame, frames, params, detector, out, cap, count):
    check(ctx._L.vslam_corner_detect_batch_dev(ctx.ptr, _tp(gray), w, h, stride, stride_frame, frames,
                                               C.byref(params), detector, _tp(out), cap, _tp(count)), ctx.ptr)
