"""Bank of filters through vslam_measurement_update_iekf_batch_dev against the single-filter entry
(Cholesky route) on the same inputs: posterior mean and covariance to 1e-9 where the iteration counts agree
(relative to the posterior spread where rounding noise made them part), same fmin code."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _bank(L, B, seed0, perturb=True):
    from visualslam_amd import synthetic as syn
    scs = [syn.Scene(L, seed0 + i, dense_S=True) for i in range(B)]
    rng = np.random.default_rng(seed0)
    mus = []
    for sc in scs:
        d = np.concatenate([np.zeros(6), 1e-4 * rng.standard_normal(6), 1e-2 * rng.standard_normal(3 * L)])
        mus.append(sc.x + (d if perturb else 0.0))
    return scs, np.stack(mus)


@pytest.mark.parametrize("L,B", [(5, 3), (50, 6), (150, 4), (500, 2)])
def test_bank_matches_single_filter_updates(ctx, gcam, L, B):
    import torch
    from visualslam_amd import api
    scs, mus = _bank(L, B, 700 + L)
    n = 12 + 3 * L
    dev = torch.device("cuda:0")
    mu = torch.tensor(mus, device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs]), device=dev)
    y = torch.tensor(np.stack([sc.y for sc in scs]), device=dev)
    vis = torch.tensor(np.stack([sc.visible for sc in scs]).astype(np.uint8), device=dev)
    mup = torch.empty_like(mu)
    Sp = torch.empty_like(S)
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, scs[0].sigma, mup, Sp)
    torch.cuda.synchronize()
    assert not status.any()                    # every filter is finished on the device, indefinite Hessians included
    ctx.set_trs_solver(api.TRS_CHOLESKY)
    try:
        for b, sc in enumerate(scs):
            c0, m0, S0, it0, _ = api.measurement_update_iekf(ctx, gcam, mus[b], sc.S, sc.y, sc.visible, sc.sigma)
            assert ctx.last_trs_route() == api.TRS_CHOLESKY
            m1 = mup[b].cpu().numpy()
            S1 = Sp[b].cpu().numpy().T
            assert code[b] == c0
            assert abs(int(iters[b]) - it0) <= 4
            if int(iters[b]) == it0:
                assert np.abs(m1 - m0).max() <= 1e-9 * max(1.0, np.abs(m0).max())
                assert rel_err(S1.T @ S1, S0.T @ S0) <= 1e-9
            else:
                # the two entries use different triangular-solve kernels; in the last iterations, where rho is rounding
                # noise, their accept/reject decisions may part: the means then differ by steps that are tiny against
                # the posterior spread
                assert np.linalg.norm(np.linalg.solve(S0.T, m1 - m0)) <= 1e-6
                assert rel_err(S1.T @ S1, S0.T @ S0) <= 1e-6
            assert not np.tril(S1, -1).any()
    finally:
        ctx.set_trs_solver(api.TRS_CHOLESKY)


@pytest.mark.parametrize("L,B,perturb", [(5, 4, False), (20, 3, False), (5, 4, True), (20, 3, True), (150, 3, True)])
def test_bank_matches_the_cpu_restatement(ctx, gcam, ocam, L, B, perturb):
    """The bank against the CPU restatement of measurementUpdateIEKF (eigenvalue trust-region route, as the
    reference runs it), filter by filter: same return code, posterior mean to 1e-9 of its largest entry and
    S^T S to 1e-9 relative where the iteration counts agree (they must for the five-landmark filters started at
    the scene's own mean), up to the 150-landmark configuration (state dimension 462).  The counts never differ by
    more than four.  Elsewhere the last iterations are decided by rounding (|p.g| < 2 eps, rho of steps
    the cost no longer resolves): the counts may differ and the results are compared against the posterior spread."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api
    scs, mus = _bank(L, B, 4100 + L, perturb)
    dev = torch.device("cuda:0")
    mu = torch.tensor(mus, device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs]), device=dev)
    y = torch.tensor(np.stack([sc.y for sc in scs]), device=dev)
    vis = torch.tensor(np.stack([sc.visible for sc in scs]).astype(np.uint8), device=dev)
    mup, Sp = torch.empty_like(mu), torch.empty_like(S)
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, scs[0].sigma, mup, Sp)
    torch.cuda.synchronize()
    assert not status.any()
    same_count = 0
    for b, sc in enumerate(scs):
        ret, m0, S0, it0, _ = orc.measurement_update_iekf(mus[b], sc.S, sc.y, sc.visible, ocam, sc.sigma)
        m1 = mup[b].cpu().numpy()
        S1 = Sp[b].cpu().numpy().T
        assert code[b] == ret
        assert abs(int(iters[b]) - it0) <= 4
        if not perturb and L == 5:
            assert int(iters[b]) == it0
        if int(iters[b]) == it0:
            same_count += 1
            assert np.abs(m1 - m0).max() <= 1e-9 * max(1.0, np.abs(m0).max())
            assert rel_err(S1.T @ S1, S0.T @ S0) <= 1e-9
        else:
            # Newton steps more or less at the point where |p.g| < 2 eps: the means differ by such a step, which
            # is tiny against the posterior spread (|S^-T dm| = sqrt(p.H p) ~ sqrt(eps))
            z = np.linalg.solve(S0.T, m1 - m0)
            assert np.linalg.norm(z) <= 1e-6
            assert rel_err(S1.T @ S1, S0.T @ S0) <= 1e-6
    assert same_count == B or perturb or L != 5


def test_bank_with_collapsed_trust_region_matches_eigenvalue_route(ctx, gcam):
    """Filters whose last iterations are rejected steps with a shrinking radius (the cost no longer resolves the
    step, rho is rounding noise): there the bank's multiplier iteration stops on |ff| * Delta < 1e-10, before the
    reference's |ff| < sqrt(eps).  The posterior must still agree with the eigenvalue route of the single-filter
    entry: to 1e-9 where the iteration counts agree; the counts may differ by accept/reject decisions taken on
    rounding noise, and the results are then compared against the posterior spread."""
    import torch
    from visualslam_amd import api
    from visualslam_amd import synthetic as syn
    L, B = 150, 3
    scs = [syn.Scene(L, 9000 + i, dense_S=True) for i in range(B)]
    rng = np.random.default_rng(1)
    mus = np.stack([sc.x + np.concatenate([np.zeros(6), 1e-4 * rng.standard_normal(6), 1e-2 * rng.standard_normal(3 * L)])
                    for sc in scs])
    dev = torch.device("cuda:0")
    mu = torch.tensor(mus, device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs]), device=dev)
    y = torch.tensor(np.stack([sc.y for sc in scs]), device=dev)
    vis = torch.tensor(np.stack([sc.visible for sc in scs]).astype(np.uint8), device=dev)
    mup, Sp = torch.empty_like(mu), torch.empty_like(S)
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, scs[0].sigma, mup, Sp)
    torch.cuda.synchronize()
    assert status[0] == 0 and iters[0] >= 7      # this filter goes through several rejected steps
    ctx.set_trs_solver(api.TRS_EIG)
    try:
        for b, sc in enumerate(scs):
            if status[b]:                         # an indefinite Hessian on the way: reported for the caller to redo
                continue
            c0, m0, S0, it0, _ = api.measurement_update_iekf(ctx, gcam, mus[b], sc.S, sc.y, sc.visible, sc.sigma)
            assert ctx.last_trs_route() == api.TRS_EIG
            m1 = mup[b].cpu().numpy()
            S1 = Sp[b].cpu().numpy().T
            assert code[b] == c0
            assert abs(int(iters[b]) - it0) <= 4
            if int(iters[b]) == it0:
                assert np.abs(m1 - m0).max() <= 1e-9 * max(1.0, np.abs(m0).max())
                assert rel_err(S1.T @ S1, S0.T @ S0) <= 1e-9
            else:
                # one route accepted a step of rounding-noise size that the other rejected: the means differ by
                # such a step, tiny against the posterior spread
                assert np.linalg.norm(np.linalg.solve(S0.T, m1 - m0)) <= 1e-6
                assert rel_err(S1.T @ S1, S0.T @ S0) <= 1e-6
    finally:
        ctx.set_trs_solver(api.TRS_CHOLESKY)


def test_bank_rejects_bad_arguments(ctx, gcam):
    import torch
    from visualslam_amd import api
    dev = torch.device("cuda:0")
    mu = torch.zeros((2, 16), dtype=torch.float64, device=dev)          # 16 is not 12 + 3L
    S = torch.zeros((2, 16, 16), dtype=torch.float64, device=dev)
    y = torch.zeros((2, 2), dtype=torch.float64, device=dev)
    vis = torch.ones((2, 1), dtype=torch.uint8, device=dev)
    with pytest.raises(api.VslamError):
        api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, 7.0, mu.clone(), S.clone())
    mu = torch.zeros((2, 15), dtype=torch.float64, device=dev)
    S = torch.eye(15, dtype=torch.float64, device=dev).repeat(2, 1, 1)
    with pytest.raises(api.VslamError):
        api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, 0.0, mu.clone(), S.clone())


def test_bank_of_one_and_repeat_call(ctx, gcam):
    """A bank of one filter, called twice on the same context (workspace reuse), gives the same answer twice."""
    import torch
    from visualslam_amd import api
    scs, mus = _bank(50, 1, 31)
    sc = scs[0]
    dev = torch.device("cuda:0")
    mu = torch.tensor(mus, device=dev)
    S = torch.tensor(np.ascontiguousarray(sc.S.T)[None], device=dev)
    y = torch.tensor(sc.y[None], device=dev)
    vis = torch.tensor(sc.visible.astype(np.uint8)[None], device=dev)
    outs = []
    for _ in range(2):
        mup, Sp = torch.empty_like(mu), torch.empty_like(S)
        code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, sc.sigma, mup, Sp)
        torch.cuda.synchronize()
        outs.append((code.copy(), status.copy(), mup.cpu().numpy(), Sp.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])


def test_bank_larger_than_one_pass(ctx, gcam):
    """More filters than one pass holds (16 384): the second pass must pick up the right instances."""
    import torch
    from visualslam_amd import api
    L, B, uniq = 1, 16384 + 40, 7
    scs, mus = _bank(L, uniq, 900)
    idx = np.arange(B) % uniq
    dev = torch.device("cuda:0")
    mu = torch.tensor(mus[idx], device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs])[idx], device=dev)
    y = torch.tensor(np.stack([sc.y for sc in scs])[idx], device=dev)
    vis = torch.tensor(np.stack([sc.visible for sc in scs]).astype(np.uint8)[idx], device=dev)
    mup, Sp = torch.empty_like(mu), torch.empty_like(S)
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, scs[0].sigma, mup, Sp)
    torch.cuda.synchronize()
    m, P = mup.cpu().numpy(), Sp.cpu().numpy()
    for b in (0, 5, 16383, 16384, 16390, B - 1):       # same inputs => same outputs as the first occurrence, bit for bit
        first = int(idx[b])
        assert status[b] == status[first] and code[b] == code[first] and iters[b] == iters[first]
        if status[b] == 0:
            assert np.array_equal(m[b], m[first]) and np.array_equal(P[b], P[first])
    ctx.set_trs_solver(api.TRS_CHOLESKY)
    try:
        for b in (16384, B - 1):
            sc = scs[int(idx[b])]
            c0, m0, S0, _, _ = api.measurement_update_iekf(ctx, gcam, mus[int(idx[b])], sc.S, sc.y, sc.visible, sc.sigma)
            if status[b] == 0 and ctx.last_trs_route() == api.TRS_CHOLESKY:
                assert np.abs(m[b] - m0).max() <= 1e-9 * max(1.0, np.abs(m0).max())
                assert rel_err(P[b] @ P[b].T, S0.T @ S0) <= 1e-9
    finally:
        ctx.set_trs_solver(api.TRS_CHOLESKY)


def test_subproblems_solved_ahead_change_nothing_that_matters(ctx, gcam, monkeypatch):
    """The bank with and without the subproblems of rejected steps solved ahead (VSLAM_BANK_SOLVE_AHEAD=0): a radius
    solved ahead is 0.25^k of the filter's radius instead of 0.25 |p| step by step, equal to 1e-9 relative where it is
    used, so the results agree to 1e-9 where the iteration counts do and against the posterior spread elsewhere; the
    scenes are the ones whose last iterations are rejected steps."""
    import torch
    from visualslam_amd import api
    from visualslam_amd import synthetic as syn
    L, B = 150, 6
    scs = [syn.Scene(L, 9000 + i, dense_S=True) for i in range(B)]
    rng = np.random.default_rng(1)
    mus = np.stack([sc.x + np.concatenate([np.zeros(6), 1e-4 * rng.standard_normal(6), 1e-2 * rng.standard_normal(3 * L)])
                    for sc in scs])
    dev = torch.device("cuda:0")
    mu = torch.tensor(mus, device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs]), device=dev)
    y = torch.tensor(np.stack([sc.y for sc in scs]), device=dev)
    vis = torch.tensor(np.stack([sc.visible for sc in scs]).astype(np.uint8), device=dev)
    out = []
    for ahead in ("1", "0"):
        monkeypatch.setenv("VSLAM_BANK_SOLVE_AHEAD", ahead)
        mup, Sp = torch.empty_like(mu), torch.empty_like(S)
        code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, scs[0].sigma, mup, Sp)
        torch.cuda.synchronize()
        out.append((np.array(code), np.array(iters), np.array(status), mup.cpu().numpy(), Sp.cpu().numpy()))
    (c1, i1, s1, m1, S1), (c0, i0, s0, m0, S0) = out
    assert np.array_equal(s1, s0) and np.array_equal(c1[s0 == 0], c0[s0 == 0])
    assert (i0[s0 == 0] >= 7).any()                       # rejected steps did occur
    for b in range(B):
        if s0[b]:
            continue
        A0, A1 = S0[b].T, S1[b].T
        if i1[b] == i0[b]:
            assert np.abs(m1[b] - m0[b]).max() <= 1e-9 * max(1.0, np.abs(m0[b]).max())
            assert rel_err(A1.T @ A1, A0.T @ A0) <= 1e-9
        else:
            assert np.linalg.norm(np.linalg.solve(A0.T, m1[b] - m0[b])) <= 1e-6
            assert rel_err(A1.T @ A1, A0.T @ A0) <= 1e-6


def test_subproblems_solved_ahead_are_consumed_and_counts_stay_near_the_oracle(ctx, gcam, ocam, monkeypatch, capfd):
    """A bank in which the subproblems solved ahead are certainly used: fourteen filters that finish early beside two whose
    last iterations are rejected steps (spare slots are only taken while filters and subproblems together stay within
    half the bank).  The library's per-round trace must show spare slots in use and rounds with more than one evaluation
    pass (a solution solved ahead was evaluated); every filter's iteration count stays within four of the CPU
    restatement's and its posterior agrees with it."""
    import re
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    L = 150
    rng = np.random.default_rng(1)
    tail = [syn.Scene(L, 9000 + i, dense_S=True) for i in range(2)]                 # several rejected steps at the end (checked below)
    quick = syn.Scene(L, 9001, dense_S=True)
    scs = tail + [quick] * 14
    mus = [sc.x + np.concatenate([np.zeros(6), 1e-4 * rng.standard_normal(6), 1e-2 * rng.standard_normal(3 * L)]) for sc in tail]
    mus += [quick.x.copy() for _ in range(14)]
    mus = np.stack(mus)
    dev = torch.device("cuda:0")
    mu = torch.tensor(mus, device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs]), device=dev)
    y = torch.tensor(np.stack([sc.y for sc in scs]), device=dev)
    vis = torch.tensor(np.stack([sc.visible for sc in scs]).astype(np.uint8), device=dev)
    mup, Sp = torch.empty_like(mu), torch.empty_like(S)
    torch.cuda.synchronize()
    monkeypatch.setenv("VSLAM_BANK_TRACE", "1")
    capfd.readouterr()
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, 7.0, mup, Sp)
    torch.cuda.synchronize()
    trace = capfd.readouterr().err
    rounds = re.findall(r"bank round \d+: (\d+) filters \+ (\d+) subproblems solved ahead, (\d+) multiplier steps, (\d+) evaluation passes", trace)
    assert rounds, trace
    assert any(int(ahead) > 0 and int(passes) > 1 for _, ahead, _, passes in rounds), trace
    assert not status.any()
    for b in (0, 1, 2):                                                              # the two long filters and one of the quick ones
        ret0, m0, S0, it0, _ = orc.measurement_update_iekf(mus[b], scs[b].S, scs[b].y, scs[b].visible, ocam, 7.0)
        m1, S1 = mup[b].cpu().numpy(), Sp[b].cpu().numpy().T
        assert code[b] == ret0 and abs(int(iters[b]) - it0) <= 4
        if int(iters[b]) == it0:
            assert np.abs(m1 - m0).max() <= 1e-9 * max(1.0, np.abs(m0).max())
            assert rel_err(S1.T @ S1, S0.T @ S0) <= 1e-9
        else:
            assert np.linalg.norm(np.linalg.solve(S0.T, m1 - m0)) <= 1e-6
            assert rel_err(S1.T @ S1, S0.T @ S0) <= 1e-6
    assert max(int(iters[0]), int(iters[1])) >= 7
    for b in range(3, 16):                                                           # identical filters, identical answers
        assert iters[b] == iters[2] and torch.equal(mup[b], mup[2]) and torch.equal(Sp[b], Sp[2])


@pytest.mark.gpu
def test_bank_entry_with_large_states_stays_on_the_device_and_equals_the_single_filter_entry(ctx, gcam):
    """States too large for the lock-step bank's kernels (more than 1 024 unknowns) take the single-filter device route one
    filter after the other, inputs and results in device memory: same code as the host-pointer entry, so the same numbers."""
    import torch
    from visualslam_amd import api, synthetic as syn
    L = 400
    scs = [syn.Scene(L, 60 + b, dense_S=True) for b in range(2)]
    dev = torch.device("cuda:0")
    mu = torch.tensor(np.stack([sc.x for sc in scs]), device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs]), device=dev)
    y = torch.tensor(np.stack([sc.y for sc in scs]), device=dev)
    vis = torch.tensor(np.stack([sc.visible for sc in scs]).astype(np.uint8), device=dev)
    mup, Sp = torch.full_like(mu, float("nan")), torch.full_like(S, float("nan"))
    torch.cuda.synchronize()
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, 7.0, mup, Sp)
    torch.cuda.synchronize()
    assert not status.any()
    for b, sc in enumerate(scs):
        ret0, m0, S0, it0, _ = api.measurement_update_iekf(ctx, gcam, sc.x, sc.S, sc.y, sc.visible, 7.0)
        assert code[b] == ret0 and iters[b] == it0
        assert np.abs(mup[b].cpu().numpy() - m0).max() <= 1e-12 * max(1.0, np.abs(m0).max())
        assert rel_err(Sp[b].cpu().numpy().T, S0) <= 1e-12
