"""GPU parity of the time update (vslam_time_update*) against the CPU oracle's dense restatement of
timeUpdateContinuous (reference src/gaussian.hpp:337-383).  The device path works on the 12-state
RK4 block and an (n + 6) x n QR -- structured O(n^2) kernels for an upper-triangular factor, the
blocked dense QR otherwise; the oracle does the reference's full n x (2n + 1) recursion and the
3n x n QR.  Tolerance: 1e-9 relative (fp64).  Parity with the reference itself is unpinned: it has no
test with numbers for this step."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

PT = np.diag([0.5, 0.5, 0.175])
OT = np.diag([0.05, 0.05, 0.05])
TOL = 1e-9


def _filter(L, seed):
    from visualslam_amd import synthetic as syn
    sc = syn.Scene(L, seed)
    rng = np.random.default_rng(seed + 77)
    x = sc.x.copy()
    x[:6] = 0.2 * rng.standard_normal(6)               # moving camera
    return x, sc.S


@pytest.mark.parametrize("L,seed,dt", [(1, 5, 0.02), (2, 6, 1 / 30), (50, 1023, 1 / 30), (150, 3, 0.05)])
def test_time_update_matches_oracle(ctx, L, seed, dt):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    x, S = _filter(L, seed)
    mu0, S0 = orc.time_update(x, S, PT, OT, dt)
    mu1, S1 = api.time_update(ctx, x, S, PT, OT, dt)
    assert rel_err(mu1, mu0) < TOL
    assert rel_err(S1, S0) < TOL
    assert not np.tril(S1, -1).any()


def test_no_landmarks_and_full_tuning_matrices(ctx):
    """n = 12 (no landmarks) and non-diagonal tuning matrices."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(9)
    x = 0.3 * rng.standard_normal(12)
    S = np.triu(0.1 * rng.standard_normal((12, 12))) + 0.3 * np.eye(12)
    pt, ot = 0.2 * rng.standard_normal((3, 3)), 0.1 * rng.standard_normal((3, 3))
    mu0, S0 = orc.time_update(x, S, pt, ot, 0.04)
    mu1, S1 = api.time_update(ctx, x, S, pt, ot, 0.04)
    assert rel_err(mu1, mu0) < TOL and rel_err(S1, S0) < TOL


def test_general_square_factor(ctx):
    """The reference does not require S to be triangular; neither does the device path."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(10)
    n = 27
    x = 0.3 * rng.standard_normal(n)
    S = 0.1 * rng.standard_normal((n, n)) + 0.3 * np.eye(n)
    mu0, S0 = orc.time_update(x, S, PT, OT, 0.03)
    mu1, S1 = api.time_update(ctx, x, S, PT, OT, 0.03)
    assert rel_err(mu1, mu0) < TOL and rel_err(S1, S0) < TOL


def test_zero_timestep_keeps_the_covariance(ctx):
    """dt = 0: the mean does not move and S'S is unchanged (R of qr(S) has the rows of S up to sign)."""
    from visualslam_amd import api
    x, S = _filter(20, 11)
    mu1, S1 = api.time_update(ctx, x, S, PT, OT, 0.0)
    assert np.array_equal(mu1, x)
    assert rel_err(S1.T @ S1, S.T @ S) < 1e-12


def test_errors(ctx):
    from visualslam_amd import api
    with pytest.raises(api.VslamError):
        api.time_update(ctx, np.zeros(11), np.eye(11), PT, OT, 0.1)
    with pytest.raises(api.VslamError):
        api.time_update(ctx, np.zeros(15), np.eye(15), PT, OT, -1.0)


def test_covariance_against_mpmath_golden(ctx):
    """tests/golden/time_update.json (50-digit mpmath evaluation of the cited formulas): propagated mean and
    covariance of a 15-state filter, through the host-pointer entry (structured path: S is upper triangular)."""
    import json
    import os
    from conftest import GOLDEN
    from visualslam_amd import api
    with open(os.path.join(GOLDEN, "time_update.json")) as fh:
        case = [c for c in json.load(fh)["cases"] if "P_next_colmajor" in c][0]
    n, dt = case["n"], case["dt"]
    S = np.array(case["S_colmajor"]).reshape(n, n).T
    mu = np.array(case["xdw"][:n])
    mup, Sp = api.time_update(ctx, mu, S, np.diag(case["pos_tune_diag"]), np.diag(case["ori_tune_diag"]), dt)
    P = np.array(case["P_next_colmajor"]).reshape(n, n).T
    assert rel_err(Sp.T @ Sp, P) <= 1e-12
    assert np.abs(mup - np.array(case["xnext"])).max() <= 1e-13
    # a general (non-triangular) factor of the same covariance takes the dense path and gives the same covariance
    rng = np.random.default_rng(1)
    Qr, _ = np.linalg.qr(rng.standard_normal((n, n)))
    mup2, Sp2 = api.time_update(ctx, mu, Qr @ S, np.diag(case["pos_tune_diag"]), np.diag(case["ori_tune_diag"]), dt)
    assert rel_err(Sp2.T @ Sp2, P) <= 1e-12


def test_structured_and_dense_paths_agree(ctx):
    """The same upper-triangular factors through both device paths (s_is_upper on / off), out of place."""
    import torch
    from visualslam_amd import api
    B, L = 3, 60
    n = 12 + 3 * L
    xs, Ss = zip(*[_filter(L, 300 + b) for b in range(B)])
    dev = torch.device("cuda:0")
    mu = torch.tensor(np.stack(xs), device=dev)
    S = torch.tensor(np.stack([s.T for s in Ss]).copy(), device=dev)
    outs = []
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    for upper in (True, False):
        mo = torch.full_like(mu, float("nan"))
        So = torch.full_like(S, float("nan"))                     # the lower triangle must be written too
        api.time_update_batch_dev(tctx, mu, S, n, PT, OT, 0.04, mo, So, s_is_upper=upper)
        torch.cuda.synchronize()
        outs.append((mo.cpu().numpy(), So.cpu().numpy()))
    tctx.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert rel_err(outs[0][1], outs[1][1]) < 1e-12
    assert not np.isnan(outs[0][1]).any()


def test_structured_path_large_state(ctx):
    """n = 1212 takes the per-panel launches of the structured path; checked through the covariance identity
    P+ = Jx P Jx' + dt Jw Q Jw' restricted to the pose block and a few landmark entries, and against the dense path."""
    import torch
    from visualslam_amd import api
    L = 400
    n = 12 + 3 * L
    x, S = _filter(L, 17)
    dev = torch.device("cuda:0")
    mu = torch.tensor(x[None], device=dev)
    St = torch.tensor(np.ascontiguousarray(S.T)[None], device=dev)
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    res = []
    for upper in (True, False):
        mo, So = torch.empty_like(mu), torch.empty_like(St)
        api.time_update_batch_dev(tctx, mu, St, n, PT, OT, 1 / 30, mo, So, s_is_upper=upper)
        torch.cuda.synchronize()
        res.append(So[0].cpu().numpy().T)
    tctx.close()
    assert rel_err(res[0], res[1]) < 1e-11
    assert not np.tril(res[0], -1).any()


def test_batch_dev_in_place(ctx):
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api
    B, L = 12, 10
    n = 12 + 3 * L
    xs, Ss = zip(*[_filter(L, 200 + b) for b in range(B)])
    dev = torch.device("cuda:0")
    mu = torch.tensor(np.stack(xs), device=dev)
    S = torch.tensor(np.stack([s.T for s in Ss]).copy(), device=dev)          # column-major per instance
    tctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    api.time_update_batch_dev(tctx, mu, S, n, PT, OT, 0.04, mu, S)
    torch.cuda.synchronize()
    mu_h, S_h = mu.cpu().numpy(), S.cpu().numpy()
    for b in range(B):
        mu0, S0 = orc.time_update(xs[b], Ss[b], PT, OT, 0.04)
        assert rel_err(mu_h[b], mu0) < TOL
        assert rel_err(S_h[b].T, S0) < TOL
    tctx.close()
