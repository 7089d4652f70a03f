"""The C-ABI library loads without a GPU and exports every symbol that
include/vslam_hip.h declares; host-only entry points work; device entry points
fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vslam_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vslam_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    from visualslam_amd import _lib
    lib = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/vslam_hip.h but not exported: %s" % missing


def test_camera_xml_reader(cam_dict):
    from visualslam_amd import api, synthetic as syn
    cam = api.camera_from_xml(syn.CAMERA_XML)
    assert cam.fx == 9.0500769512576187e+02 and cam.fy == 9.1029976093126402e+02
    assert cam.cx == 9.6719531371665323e+02 and cam.cy == 5.9638049919715968e+02
    assert list(cam.d) == list(cam_dict["dist"]) and cam.flag == 49152
    assert (cam.width, cam.height, cam.fov_deg) == (1920, 1080, 120.0)
    with pytest.raises(api.VslamError):
        api.camera_from_xml("/nonexistent/camera.xml")


def test_struct_layouts_match_header():
    from visualslam_amd import _lib
    assert ctypes.sizeof(_lib.Camera) == 152
    assert ctypes.sizeof(_lib.Feature) == 12
    assert ctypes.sizeof(_lib.CornerParams) == 24


def test_no_cpu_fallback():
    """Without a gfx950 device a context cannot be created and nothing computes."""
    import torch
    from visualslam_amd import api
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.VslamError):
        api.Context(0)


def test_status_strings():
    from visualslam_amd import _lib
    lib = _lib.lib()
    assert lib.vslam_status_string(0) == b"ok"
    assert lib.vslam_status_string(-2) == b"shape mismatch"
    assert lib.vslam_has_device_code() == 1
