"""The measurement aid of the batched point log-likelihood (events around its two kernels): results are unchanged
with timing on, durations are positive, and asking without a timed call is an error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_timed_call_gives_the_same_results_and_two_durations(gcam):
    import torch
    from visualslam_amd import api, synthetic as syn
    B, L = 24, 150                                   # 24 * 462^2 * 8 bytes > 32 MiB: the two-kernel form
    n = 12 + 3 * L
    dev = torch.device("cuda:0")
    scs = [syn.Scene(L, 7000 + i, dense_S=False) for i in range(4)]
    rep = B // len(scs)
    x = torch.tensor(np.tile(np.stack([s.x for s in scs]), (rep, 1)), device=dev)
    y = torch.tensor(np.tile(np.stack([s.y for s in scs]), (rep, 1)), device=dev)
    vis = torch.tensor(np.tile(np.stack([s.visible for s in scs]), (rep, 1)), device=dev)
    out = []
    c = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        with pytest.raises(api.VslamError):
            c.last_point_loglik_times()              # timing is off
        for timed in (False, True):
            c.set_kernel_timing(timed)
            val = torch.empty(B, dtype=torch.float64, device=dev)
            g = torch.empty((B, n), dtype=torch.float64, device=dev)
            H = torch.full((B, n, n), float("nan"), dtype=torch.float64, device=dev)
            code = torch.empty((B, L), dtype=torch.int32, device=dev)
            if timed:
                with pytest.raises(api.VslamError):
                    c.last_point_loglik_times()      # nothing timed yet
            api.point_loglik_batch_dev(c, gcam, n, x, y, vis, 7.0, val, g, H, code)
            if timed:
                t_terms, t_asm = c.last_point_loglik_times()
                assert 0.0 < t_terms < 50.0 and 0.0 < t_asm < 50.0
            torch.cuda.synchronize()
            out.append((val.cpu().numpy(), g.cpu().numpy(), H.cpu().numpy()))
        for a, b in zip(out[0], out[1]):
            np.testing.assert_array_equal(a, b)
        c.set_kernel_timing(False)
    finally:
        c.close()
