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


def test_pgm_round_trip_and_header_forms(tmp_path):
    """Binary PGM (P5) reader / writer: the wire format of gray frames at this library's boundary (the reference reads
    frames through OpenCV, src/SLAM.cpp:236).  Comments and any white space between the header tokens are legal."""
    import numpy as np
    from visualslam_amd import api
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    p = tmp_path / "a.pgm"
    api.write_pgm(p, img)
    assert p.read_bytes() == b"P5\n53 37\n255\n" + img.tobytes()
    assert np.array_equal(api.read_pgm(p), img)
    q = tmp_path / "b.pgm"
    q.write_bytes(b"P5 # made by a test\n# second comment\r\n53\t37 # width height\n255\n" + img.tobytes())
    assert np.array_equal(api.read_pgm(q), img)
    q.write_bytes(b"P5\n1 1\n7 " + bytes([3]))                     # smallest image, maxval below 255
    assert api.read_pgm(q).tolist() == [[3]]
    # the committed 1080p frame, converted, survives the trip
    from PIL import Image
    from conftest import GOLDEN
    import os
    frame = np.array(Image.open(os.path.join(GOLDEN, "frame_1023_gray.png")))
    api.write_pgm(p, frame)
    assert np.array_equal(api.read_pgm(p), frame)


@pytest.mark.parametrize("case", ["missing", "empty", "p2_ascii", "p6", "no_maxval", "zero_width", "huge_width", "sixteen_bit",
                                  "truncated", "no_separator", "letters", "ten_digits"])
def test_pgm_reader_rejects_malformed_files(tmp_path, case):
    from visualslam_amd import api
    body = bytes(range(12))
    data = {
        "empty": b"", "p2_ascii": b"P2\n4 3\n255\n" + b" ".join(b"%d" % v for v in body), "p6": b"P6\n2 2\n255\n" + body,
        "no_maxval": b"P5\n4 3\n", "zero_width": b"P5\n0 3\n255\n", "huge_width": b"P5\n70000 1\n255\n" + body,
        "sixteen_bit": b"P5\n2 3\n65535\n" + body, "truncated": b"P5\n4 3\n255\n" + body[:11],
        "no_separator": b"P5\n4 3\n255" + body, "letters": b"P5\n4 x\n255\n" + body, "ten_digits": b"P5\n0000000004 3\n255\n" + body,
    }
    p = tmp_path / "bad.pgm"
    if case != "missing":
        p.write_bytes(data[case])
    with pytest.raises(api.VslamError) as e:
        api.read_pgm(p)
    if case == "sixteen_bit":
        assert "unsupported" in str(e.value)


def test_pgm_reader_reports_a_small_buffer(tmp_path):
    import numpy as np
    from visualslam_amd import _lib, api
    p = tmp_path / "a.pgm"
    api.write_pgm(p, np.zeros((3, 4), np.uint8))
    buf, dims = np.zeros(11, np.uint8), np.zeros(2, np.int32)
    st = _lib.lib().vslam_pgm_read(str(p).encode(), buf.ctypes.data, buf.size, dims.ctypes.data, dims.ctypes.data + 4)
    assert st == -6 and dims.tolist() == [4, 3]


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
