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


def _camera_xml_text():
    from visualslam_amd import synthetic as syn
    with open(syn.CAMERA_XML) as fh:
        return fh.read()


@pytest.mark.parametrize("case", ["truncated", "matrix_short", "not_a_number", "no_distortion", "too_many_coefficients",
                                  "no_flag", "zero_width", "negative_focal", "nan_focal", "empty"])
def test_camera_xml_reader_rejects_malformed_files(tmp_path, case):
    """Wire-format hardening of the calibration reader (the reference reads the file with cv::FileStorage,
    src/cameraModel.cpp:93-100): a damaged file is an I/O error, never a half-filled camera."""
    import re
    from visualslam_amd import api
    txt = _camera_xml_text()
    data = re.compile(r"(<camera_matrix.*?<data>)(.*?)(</data>)", re.S)
    if case == "truncated":
        txt = txt[: len(txt) // 2]
    elif case == "matrix_short":
        txt = data.sub(lambda m: m.group(1) + " ".join(m.group(2).split()[:8]) + m.group(3), txt, count=1)
    elif case == "not_a_number":
        txt = data.sub(lambda m: m.group(1) + m.group(2).replace(m.group(2).split()[0], "9.05e+02x", 1) + m.group(3), txt, count=1)
    elif case == "no_distortion":
        txt = re.sub(r"<distortion_coefficients.*?</distortion_coefficients>", "", txt, flags=re.S)
    elif case == "too_many_coefficients":
        txt = re.sub(r"(<distortion_coefficients.*?<data>)(.*?)(</data>)", lambda m: m.group(1) + " ".join(["0.1"] * 13) + m.group(3), txt, count=1, flags=re.S)
    elif case == "no_flag":
        txt = re.sub(r"<flag>.*?</flag>", "", txt, flags=re.S)
    elif case == "zero_width":
        txt = re.sub(r"<imageSize>.*?</imageSize>", "<imageSize>0 1080</imageSize>", txt, flags=re.S)
    elif case == "negative_focal":
        txt = data.sub(lambda m: m.group(1) + "-" + m.group(2).lstrip() + m.group(3), txt, count=1)
    elif case == "nan_focal":
        txt = data.sub(lambda m: m.group(1) + " ".join(["nan"] + m.group(2).split()[1:]) + m.group(3), txt, count=1)
    elif case == "empty":
        txt = ""
    path = tmp_path / "camera.xml"
    path.write_text(txt)
    with pytest.raises(api.VslamError):
        api.camera_from_xml(path)


def test_camera_xml_reader_accepts_reformatted_file(tmp_path):
    """Whitespace and line breaks inside the data blocks do not matter."""
    import re
    from visualslam_amd import api, synthetic as syn
    txt = re.sub(r"[ \t]+", "  ", _camera_xml_text()).replace("\n", "\n\n")
    path = tmp_path / "camera.xml"
    path.write_text(txt)
    a, b = api.camera_from_xml(path), api.camera_from_xml(syn.CAMERA_XML)
    assert bytes(a) == bytes(b)


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
