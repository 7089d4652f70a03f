import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def cam_dict():
    from visualslam_amd import synthetic as syn
    return syn.read_camera_xml()


@pytest.fixture(scope="session")
def ocam(cam_dict):
    """Camera struct for the CPU oracle."""
    from oracle import pyoracle as orc
    c = cam_dict
    return orc.make_camera(c["K"][0, 0], c["K"][1, 1], c["K"][0, 2], c["K"][1, 2], c["dist"], c["flag"],
                           c["width"], c["height"])


@pytest.fixture(scope="session")
def gcam():
    """Camera struct for libvslam_hip, parsed by the library's own camera.xml reader."""
    from visualslam_amd import api, synthetic as syn
    return api.camera_from_xml(syn.CAMERA_XML)


@pytest.fixture(scope="session")
def ctx():
    """The library context shared by the GPU tests.  It runs on torch's current stream, so that tensors the tests
    build with torch (fills, copies) are ordered before the library's kernels without a device-wide synchronisation;
    tests that want a stream of the library's own create their context themselves."""
    import torch
    from visualslam_amd import api
    c = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def rel_err(a, b, floor=None):
    """max |a-b| / max(|b|, floor): relative to the largest reference entry
    (cancellation-prone entries get the block's scale as an absolute floor)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b)) if b.size else 1.0
    if floor is not None:
        scale = max(scale, floor)
    if scale == 0:
        scale = 1.0
    return float(np.max(np.abs(a - b)) / scale) if a.size else 0.0
