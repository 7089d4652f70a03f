"""ctypes loader for libvslam_hip.so (the C ABI of include/vslam_hip.h).

There is no CPU fallback: if the shared library is missing, or a context cannot
be created on a gfx950 device, the error is raised to the caller.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSLAM_HIP_LIB: another build of the same library (diagnostic builds, A/B runs)
LIB_PATH = os.environ.get("VSLAM_HIP_LIB") or os.path.join(_HERE, "libvslam_hip.so")


class VslamError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("d", C.c_double * 12), ("fov_deg", C.c_double),
                ("flag", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("reserved_", C.c_int32)]


class CornerParams(C.Structure):
    _fields_ = [("block_size", C.c_int32), ("border", C.c_int32), ("max_features", C.c_int32),
                ("harris_k", C.c_float), ("harris_thresh", C.c_float), ("shi_thresh", C.c_float)]


class SelectParams(C.Structure):
    _fields_ = [("feature_thresh", C.c_double), ("width_thresh", C.c_double), ("height_thresh", C.c_double),
                ("pixel_distance_thresh", C.c_double), ("image_width", C.c_int32), ("image_height", C.c_int32),
                ("max_new", C.c_int32), ("reserved", C.c_int32)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("edge_threshold", C.c_int32),
                ("fast_threshold", C.c_int32), ("harris_block", C.c_int32), ("harris_k", C.c_float), ("reserved_", C.c_int32)]


class Feature(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("score", C.c_float)]


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  A process must
    use ONE HIP runtime: if this library pulled in /opt/rocm's copy first, a
    later `import torch` would load a second runtime and find no GPU.  So when
    torch is installed, its runtime is loaded first (by path, without
    importing torch); libvslam_hip then binds to it by soname.  Standalone C++
    consumers are unaffected (RUNPATH -> /opt/rocm).  VSLAM_HIP_RUNTIME=system
    skips this."""
    if os.environ.get("VSLAM_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded library; raises VslamError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VslamError("libvslam_hip.so not found at %s -- build it with "
                             "`make -C visualslam_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
        _preload_torch_hip_runtime()
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


_vp = C.c_void_p
_i = C.c_int
_d = C.c_double
_sz = C.c_size_t


def _declare(L):
    L.vslam_status_string.restype = C.c_char_p
    L.vslam_status_string.argtypes = [C.c_int32]
    L.vslam_last_error.restype = C.c_char_p
    L.vslam_last_error.argtypes = [_vp]
    sig = {
        "vslam_ctx_create": [_i, C.POINTER(_vp)],
        "vslam_ctx_create_on_stream": [_i, _vp, C.POINTER(_vp)],
        "vslam_ctx_destroy": [_vp],
        "vslam_ctx_synchronize": [_vp],
        "vslam_ctx_set_trs_solver": [_vp, C.c_int],
        "vslam_ctx_last_trs_route": [_vp],
        "vslam_ctx_last_update_profile": [_vp, _vp],
        "vslam_ctx_set_kernel_timing": [_vp, C.c_int],
        "vslam_ctx_last_point_loglik_times": [_vp, _vp, _vp],
        "vslam_camera_from_xml": [C.c_char_p, C.POINTER(Camera)],
        "vslam_pgm_info": [C.c_char_p, _vp, _vp, _vp],
        "vslam_pgm_read": [C.c_char_p, _vp, _sz, _vp, _vp],
        "vslam_pgm_write": [C.c_char_p, _vp, _i, _i, _i],
        "vslam_project": [_vp, C.POINTER(Camera), _vp, _i, _i, _i, _vp, _vp],
        "vslam_bundle_measure": [_vp, C.POINTER(Camera), _vp, _i, _i, _vp, _vp, _vp, _vp],
        "vslam_bundle_measure_dev": [_vp, C.POINTER(Camera), _vp, _i, _i, _vp, _vp, _vp, _vp],
        "vslam_single_measure": [_vp, C.POINTER(Camera), _i, _vp, _i, _i, _vp, _vp, _vp, _vp],
        "vslam_point_loglik": [_vp, C.POINTER(Camera), _vp, _i, _vp, _vp, _d, _vp, _vp, _vp, _i, _vp],
        "vslam_point_loglik_batch_dev": [_vp, C.POINTER(Camera), _i, _i, _vp, _sz, _vp, _sz, _vp, _sz, _d,
                                         _vp, _vp, _sz, _vp, _i, _sz, _vp],
        "vslam_point_loglik_batch_dev_noalloc": [_vp, C.POINTER(Camera), _i, _i, _vp, _sz, _vp, _sz, _vp, _sz, _d,
                                                 _vp, _vp, _sz, _vp, _i, _sz, _vp],
        "vslam_log_gaussian": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
        "vslam_affine_sqrt": [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _i],
        "vslam_affine_bundle": [_vp, C.POINTER(Camera), _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp],
        "vslam_affine_bundle_batch_dev": [_vp, C.POINTER(Camera), _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
        "vslam_qr_upper": [_vp, _vp, _i, _i, _i],
        "vslam_qr_upper_batch_dev": [_vp, _vp, _i, _i, _i, _sz, _i],
        "vslam_iekf_post_sqrt": [_vp, _vp, _vp, _i, _vp],
        "vslam_condition_on_marginal": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i],
        "vslam_cost_joint_density": [_vp, C.POINTER(Camera), _vp, _vp, _vp, _i, _i, _vp, _vp, _d, _vp, _vp, _vp],
        "vslam_measurement_update_iekf_batch_dev": [_vp, C.POINTER(Camera), _vp, _vp, _i, _i, _vp, _vp, _d, _vp, _vp, _vp, _vp, _vp],
        "vslam_measurement_update_iekf": [_vp, C.POINTER(Camera), _vp, _vp, _i, _i, _vp, _vp, _d,
                                          _vp, _vp, _i, _vp, _vp, _vp],
        "vslam_individual_compat": [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp],
        "vslam_joint_compat": [_vp, _vp, _i, _d, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp],
        "vslam_snn_density": [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp],
        "vslam_snn_density_batch_dev": [_vp, _i, _vp, _sz, _vp, _i, _sz, _i, _vp, _sz, _i, _i, _i, _i, _vp, _vp],
        "vslam_snn": [_vp, C.POINTER(Camera), _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp],
        "vslam_time_update": [_vp, _vp, _vp, _i, _i, _vp, _vp, _d, _vp, _vp, _i],
        "vslam_time_update_batch_dev": [_vp, _vp, _vp, _i, _i, _vp, _vp, _d, _vp, _vp, _i],
        "vslam_remove_landmark": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i],
        "vslam_remove_landmark_dev": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i],
        "vslam_remove_landmark_batch_dev": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
        "vslam_match_hamming": [_vp, _vp, _i, _vp, _i, _i, _vp, _vp],
        "vslam_match_hamming_batch_dev": [_vp, _vp, _sz, _i, _vp, _sz, _i, _vp, _i, _i, _vp, _vp],
        "vslam_corner_detect": [_vp, _vp, _i, _i, _i, C.POINTER(CornerParams), _i, _vp, _i, _vp],
        "vslam_corner_response": [_vp, _vp, _i, _i, _i, C.POINTER(CornerParams), _vp, _vp],
        "vslam_select_new_landmarks": [_vp, _vp, _i, _vp, _i, C.POINTER(SelectParams), _vp, _i, _vp],
        "vslam_select_new_landmarks_batch_dev": [_vp, _vp, _sz, _vp, _i, _vp, _sz, _vp, _i, C.POINTER(SelectParams), _i, _vp, _i, _vp],
        "vslam_initialize_landmarks": [_vp, C.POINTER(Camera), _vp, _vp, _i, _i, _vp, _i, _d, _d, _vp, _vp, _i],
        "vslam_initialize_landmarks_dev": [_vp, C.POINTER(Camera), _vp, _vp, _i, _i, _vp, _i, _d, _d, _vp, _vp, _i],
        "vslam_initialize_landmarks_batch_dev": [_vp, C.POINTER(Camera), _vp, _vp, _i, _i, _vp, _i, _d, _d, _vp, _vp],
        "vslam_fast9_detect": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
        "vslam_fast9_detect_batch_dev": [_vp, _vp, _i, _i, _i, _sz, _i, _i, _i, _i, _vp, _i, _vp],
        "vslam_orb_level_plan": [C.POINTER(OrbParams), _i, _i, _vp, _vp, _vp, _vp],
        "vslam_orb_detect_level_batch_dev": [_vp, _vp, _i, _i, _i, _sz, _i, _i, _i, _i, _i, C.c_float, _vp, _i, _vp],
        "vslam_resize_bilinear_dev": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i],
        "vslam_orb_detect": [_vp, _vp, _i, _i, _i, C.POINTER(OrbParams), _vp, _i, _vp, _vp],
        "vslam_orb_default_pattern": [_vp],
        "vslam_gaussian_blur7_dev": [_vp, _vp, _i, _i, _i, _vp, _i],
        "vslam_orb_describe_level_dev": [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp],
        "vslam_orb_detect_and_compute": [_vp, _vp, _i, _i, _i, C.POINTER(OrbParams), _vp, _vp, _vp, _vp, _i, _vp, _vp],
        "vslam_orb_detect_and_compute_batch_dev": [_vp, _vp, _i, _i, _i, _sz, _i, C.POINTER(OrbParams), _vp, _vp, _vp, _vp, _i, _vp, _vp],
        "vslam_corner_detect_batch_dev": [_vp, _vp, _i, _i, _i, _sz, _i, C.POINTER(CornerParams), _i, _vp, _i, _vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name, None)
        if fn is None:
            continue          # declared in the header but not built: surfaced by tests/test_abi.py
        fn.restype = C.c_int32
        fn.argtypes = args
    L.vslam_has_device_code.restype = C.c_int
    L.vslam_normcdf.restype = C.c_double
    L.vslam_normcdf.argtypes = [_d]
    L.vslam_chi2inv.restype = C.c_double
    L.vslam_chi2inv.argtypes = [_d, _d]
    L.vslam_pixel_distance.restype = C.c_int
    L.vslam_pixel_distance.argtypes = [_vp, _i, C.c_float, C.c_float, _d]
    L.vslam_optical_ray.restype = C.c_int
    L.vslam_optical_ray.argtypes = [C.POINTER(Camera), _d, _d, _d, _vp]


def check(status, ctx=None):
    if status != 0:
        L = lib()
        msg = L.vslam_status_string(status).decode()
        if ctx:
            detail = L.vslam_last_error(ctx).decode()
            if detail:
                msg += ": " + detail
        raise VslamError("libvslam_hip: %s (status %d)" % (msg, status))
