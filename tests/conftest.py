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


_CONTEXTS = {}


def _context(kind):
    import torch
    from visualslam_amd import api
    if kind not in _CONTEXTS:
        _CONTEXTS[kind] = api.Context(0, stream=torch.cuda.current_stream().cuda_stream) if kind == "borrowed" else api.Context(0)
    return _CONTEXTS[kind]


@pytest.fixture(scope="session", autouse=True)
def _close_contexts():
    yield
    for c in _CONTEXTS.values():
        c.close()
    _CONTEXTS.clear()


@pytest.fixture
def ctx(request):
    """The library context shared by the GPU tests.  "borrowed" (the default) runs on torch's current stream, so that
    tensors the tests build with torch (fills, copies) are ordered before the library's kernels without a device-wide
    synchronisation.  "owned" is what Context(device) gives a caller: a non-blocking stream of the library's own;
    modules that set BOTH_CONTEXTS = True run every test that takes `ctx` under both (a test that hands torch tensors
    to an owned-stream context synchronises the device first)."""
    return _context(getattr(request, "param", "borrowed"))


def pytest_generate_tests(metafunc):
    if "ctx" in metafunc.fixturenames and getattr(metafunc.module, "BOTH_CONTEXTS", False):
        metafunc.parametrize("ctx", ["borrowed", "owned"], indirect=True)


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


# ---- per-landmark comparison (the parity statement: "per-landmark log-likelihood / gradient / Hessian within 1e-9
# relative") ----
# A block is one landmark's piece of an output: its value term, its 3-gradient, its 3 x 3 diagonal block, one of its
# two 3 x 3 cross blocks with the pose, its contribution to a 3-piece of the pose gradient or to a 3 x 3 piece of the
# 6 x 6 pose block.  The error of a block is taken relative to THAT block's largest reference entry.  Absolute floor:
# a block whose largest entry is below BLOCK_FLOOR times the largest block of its kind is measured against that floor
# instead (its entries are differences of terms of the larger size and carry their rounding); a block that is exactly
# zero in the reference (culled or invisible landmark) must be exactly zero.
BLOCK_FLOOR = 1e-6


def block_errs(got, ref, floor_frac=BLOCK_FLOOR):
    """got, ref: (L, ...) arrays of per-landmark blocks.  Returns (worst relative error, index of the worst block);
    asserts exact zeros where the reference block is exactly zero."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    L = ref.shape[0]
    g2, r2 = got.reshape(L, -1), ref.reshape(L, -1)
    scale = np.max(np.abs(r2), axis=1)
    zero = scale == 0.0
    assert not g2[zero].any(), "a block that is exactly zero in the reference must be exactly zero"
    top = float(scale.max()) if L else 0.0
    if top == 0.0:
        return 0.0, -1
    err = np.max(np.abs(g2 - r2), axis=1) / np.maximum(scale, floor_frac * top)
    j = int(np.argmax(err))
    return float(err[j]), j


def oracle_per_landmark(ocam, x, y, visible, sigma, flavour="dual"):
    """Each landmark's term on its own from the CPU oracle: the term of landmark j depends on the pose eta = x[6:12]
    and r_j only, so it is evaluated on the 15-state [0 (6), eta, r_j].  Returns ell (L), g9 (L, 9), H9 (L, 9, 9),
    code (L); the 9 variables are (rCNn, Theta, r_j)."""
    from oracle import pyoracle as orc
    fn = orc.point_loglik_dual_sparse if flavour == "dual" else orc.point_loglik_analytic
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    L = (x.size - 12) // 3
    ell, g9, H9, code = np.zeros(L), np.zeros((L, 9)), np.zeros((L, 9, 9)), np.zeros(L, dtype=np.int32)
    xj = np.zeros(15)
    xj[6:12] = x[6:12]
    for j in range(L):
        xj[12:15] = x[12 + 3 * j:15 + 3 * j]
        v, g, H, c = fn(xj, y[2 * j:2 * j + 2], [visible[j]], ocam, sigma)
        ell[j], g9[j], H9[j], code[j] = v, g[6:], H[6:, 6:], c[0]
    return ell, g9, H9, code
