"""GPU parity of projection and the measurement model (h, SR, Jacobian) in both
column layouts, against the CPU oracle and the symbolic golden vectors.
The reference has no test for these outputs (its matcher/measurement tests
are vacuous or disabled), so the oracle + golden vectors define the check.
Tolerance 1e-9 relative (fp64)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-9


def test_project_golden(ctx, gcam):
    from visualslam_amd import api
    G = json.load(open(os.path.join(GOLDEN, "world_to_pixel.json")))
    for k in G["cases"]:
        x = np.zeros(15)
        x[6:12], x[12:15] = k["eta"], k["r"]
        px, code = api.project(ctx, gcam, x)
        assert code[0] == k["code"], k["tag"]
        if k["code"] == 1:
            assert np.isnan(px).all()
        else:
            assert rel_err(px, k["px"], floor=1.0) <= TOL


def test_project_range(ctx, gcam, ocam):
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(40, 3, dense_S=False)
    px, code = api.project(ctx, gcam, sc.x, j0=7, L=20)
    for j in range(20):
        c0, p0 = orc.world_to_pixel(sc.x[12 + 3 * (7 + j):15 + 3 * (7 + j)], sc.x[6:12], ocam)
        assert code[j] == c0 and rel_err(px[2 * j:2 * j + 2], p0) <= TOL
    with pytest.raises(api.VslamError):
        api.project(ctx, gcam, sc.x, j0=30, L=20)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("L,seed", [(1, 2), (4, 21), (50, 1023), (150, 3), (700, 6)])
def test_bundle_vs_oracle(ctx, gcam, ocam, layout, L, seed):
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(L, seed, dense_S=False)
    x = sc.x.copy()
    x[3:6] = [0.3, -0.2, 0.5]                  # the COMPAT quirk reads these
    if L >= 4:
        x[12 + 3 * 2:15 + 3 * 2] = x[6:9] - (x[12 + 3 * 2:15 + 3 * 2] - x[6:9])    # behind the camera
    h, SR, C, code = api.bundle_measure(ctx, gcam, x, layout)
    h0, SR0, C0, code0 = orc.bundle_measure(x, ocam, layout)
    assert np.array_equal(code, code0)
    ok = code0 != 1
    assert rel_err(h.reshape(-1, 2)[ok], h0.reshape(-1, 2)[ok]) <= TOL
    assert np.isnan(h.reshape(-1, 2)[~ok]).all()
    assert np.array_equal(SR, SR0)
    assert rel_err(C, C0) <= TOL
    assert np.array_equal(C != 0, C0 != 0), "same sparsity pattern"


@pytest.mark.parametrize("L,shift", [(3, 0), (150, 0), (150, 1), (2100, 0), (4500, 1)])
def test_bundle_device_entry_fills_every_element(ctx, gcam, ocam, L, shift):
    """vslam_bundle_measure_dev into NaN-filled device arrays (every element of the dense C and SR must be written,
    zeros included), also at an 8-byte-aligned base address, against the host-pointer entry bit for bit; more than
    2048 landmarks puts several landmarks into one workgroup."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(L, 77 + L, dense_S=False)
    n, m = 12 + 3 * L, 2 * L
    dev = torch.device("cuda:0")
    x = torch.tensor(sc.x, device=dev)
    h = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
    Cbuf = torch.full((m * n + 2,), float("nan"), dtype=torch.float64, device=dev)
    Sbuf = torch.full((m * m + 2,), float("nan"), dtype=torch.float64, device=dev)
    code = torch.full((L,), -1, dtype=torch.int32, device=dev)
    Cd, Sd = Cbuf[shift:shift + m * n], Sbuf[shift:shift + m * m]
    torch.cuda.synchronize()                    # the fills run on torch's stream, the library on the context's
    api.bundle_measure_dev(ctx, gcam, x, h, Sd, Cd, code, layout=1)
    ctx.synchronize()
    torch.cuda.synchronize()
    Cg = Cd.cpu().numpy().reshape(n, m).T
    Sg = Sd.cpu().numpy().reshape(m, m).T
    assert np.isnan(Cbuf[shift + m * n:].cpu().numpy()).all() and (shift == 0 or np.isnan(Cbuf[0].item()))
    assert np.isnan(Sbuf[shift + m * m:].cpu().numpy()).all() and (shift == 0 or np.isnan(Sbuf[0].item()))
    if L <= 2100:
        h0, SR0, C0, code0 = api.bundle_measure(ctx, gcam, sc.x, 1)
        assert np.array_equal(Cg, C0) and np.array_equal(Sg, SR0)
        assert np.array_equal(h.cpu().numpy(), h0, equal_nan=True) and np.array_equal(code.cpu().numpy(), code0)
    if L <= 150:
        _, _, C1, _ = orc.bundle_measure(sc.x, ocam, 1)
        assert rel_err(Cg, C1) <= TOL
    assert not np.isnan(Cg).any() and np.array_equal(Sg, 0.01 * np.eye(m))
    nz = np.count_nonzero(Cg, axis=0)
    assert (nz[12:] <= 2).all()                 # a landmark column holds its own two rows only


def test_bundle_golden(ctx, gcam):
    from visualslam_amd import api
    B = json.load(open(os.path.join(GOLDEN, "bundle.json")))
    for layout, key in ((0, "C_compat"), (1, "C_strict")):
        h, SR, C, code = api.bundle_measure(ctx, gcam, B["x"], layout)
        assert rel_err(h, B["h"]) <= TOL
        assert rel_err(C, np.array(B[key])) <= TOL
        assert np.array_equal(SR, 0.01 * np.eye(2 * B["L"]))


def test_bundle_arities(ctx, gcam, ocam):
    """1-, 2- and 3-output arities of the Bundle functor (measurementPointLandmark.cpp:13,31,44)."""
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(9, 4, dense_S=False)
    h1, SR1, C1, _ = api.bundle_measure(ctx, gcam, sc.x, 0, want_SR=False, want_C=False)
    h2, SR2, C2, _ = api.bundle_measure(ctx, gcam, sc.x, 0, want_SR=True, want_C=False)
    h3, SR3, C3, _ = api.bundle_measure(ctx, gcam, sc.x, 0)
    assert SR1 is None and C1 is None and C2 is None
    assert np.array_equal(h1, h2) and np.array_equal(h2, h3) and np.array_equal(SR2, SR3)


@pytest.mark.parametrize("layout", [0, 1])
def test_single_vs_oracle(ctx, gcam, ocam, layout):
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(6, 17, dense_S=False)
    x = sc.x.copy()
    x[3:6] = [0.1, 0.2, -0.4]
    for j in range(6):
        code, h, SR, J = api.single_measure(ctx, gcam, j, x, layout)
        c0, h0, SR0, J0 = orc.single_measure(j, x, ocam, layout)
        assert code == c0 and rel_err(h, h0) <= TOL and np.array_equal(SR, SR0) and rel_err(J, J0) <= TOL
    with pytest.raises(api.VslamError):
        api.single_measure(ctx, gcam, 6, x, layout)
