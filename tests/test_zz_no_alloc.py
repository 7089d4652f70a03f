"""The entry point that allocates nothing (vslam_point_loglik_batch_dev_noalloc) against the
existing batched device entry point (vslam_point_loglik_batch_dev), through the C ABI.

Both run the same two kernels; the no-allocation form keeps the per-landmark records and the
per-block pose sums in a static device array of the library instead of a buffer from hipMalloc.
Bar: bit-identical value / gradient / Hessian / cull codes, on two calls in a row, the second
with a smaller input (the array keeps the first call's contents), plus the capacity error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(tctx, gcam, scs, noalloc):
    import torch
    from visualslam_amd import api
    B, L = len(scs), scs[0].L
    n = 12 + 3 * L
    dev = torch.device("cuda:0")
    x = torch.tensor(np.stack([s.x for s in scs]), device=dev)
    y = torch.tensor(np.stack([s.y for s in scs]), device=dev)
    vis = torch.tensor(np.stack([s.visible for s in scs]), device=dev)
    val = torch.full((B,), float("nan"), dtype=torch.float64, device=dev)
    g = torch.full((B, n), float("nan"), dtype=torch.float64, device=dev)
    H = torch.full((B, n, n), float("nan"), dtype=torch.float64, device=dev)
    code = torch.full((B, L), -9, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    api.point_loglik_batch_dev(tctx, gcam, n, x, y, vis, 7.0, val, g, H, code, noalloc=noalloc)
    torch.cuda.synchronize()
    return val.cpu().numpy(), g.cpu().numpy(), H.cpu().numpy(), code.cpu().numpy()


def test_noalloc_matches_batch_dev_twice(gcam, ocam):
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        # first call: 40 instances of 300 landmarks (two 256-landmark blocks each);
        # second call: smaller in both the batch and the landmark count
        for B, L, seed in ((40, 300, 700), (3, 50, 900)):
            scs = [syn.Scene(L, seed + i, dense_S=False) for i in range(B)]
            want = _run(tctx, gcam, scs, noalloc=False)
            got = _run(tctx, gcam, scs, noalloc=True)
            for a, b in zip(got, want):
                assert not np.isnan(a).any()
                assert np.array_equal(a, b)
            # and the first instance against the CPU oracle, to the parity tolerance of the path
            v0, g0, H0, c0 = orc.point_loglik_analytic(scs[0].x, scs[0].y, scs[0].visible, ocam, 7.0)
            assert abs(got[0][0] - v0) <= 1e-9 * abs(v0)
            assert np.max(np.abs(got[1][0] - g0)) <= 1e-9 * np.max(np.abs(g0))
            assert np.max(np.abs(got[2][0].T - H0)) <= 1e-9 * np.max(np.abs(H0))
            assert np.array_equal(got[3][0], c0)
    finally:
        tctx.close()


def test_noalloc_rejects_sizes_beyond_the_static_scratch(gcam):
    import torch
    from visualslam_amd import api
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        B, L = 600, 150                                   # 90 000 landmarks > 81 920
        n = 12 + 3 * L
        dev = torch.device("cuda:0")
        x = torch.zeros((B, n), dtype=torch.float64, device=dev)
        y = torch.zeros((B, 2 * L), dtype=torch.float64, device=dev)
        vis = torch.ones((B, L), dtype=torch.uint8, device=dev)
        val = torch.zeros((B,), dtype=torch.float64, device=dev)
        g = torch.zeros((B, n), dtype=torch.float64, device=dev)
        H = torch.zeros((B, n, n), dtype=torch.float64, device=dev)
        with pytest.raises(api.VslamError, match="static scratch"):
            api.point_loglik_batch_dev(tctx, gcam, n, x, y, vis, 7.0, val, g, H, None, noalloc=True)
        with pytest.raises(api.VslamError, match="ldH"):
            api.point_loglik_batch_dev(tctx, gcam, n, x[:2], y[:2], vis[:2], 7.0, val[:2], g[:2], H[:2], None, ldH=n + 4,
                                       noalloc=True)
    finally:
        tctx.close()
