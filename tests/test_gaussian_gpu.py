"""GPU parity of the square-root Gaussian algebra (logGaussian, affineTransform,
conditioning, joint cost) and of the full iterated-EKF measurement update
against the CPU oracle and the reference's known answers.  fp64, 1e-9 relative
unless a test states otherwise."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
BOTH_CONTEXTS = True      # every test that takes `ctx` runs on torch's stream and on a stream of the library's own
TOL = 1e-9


def test_log_gaussian_reference_kats(ctx):
    """test/src/logGaussian.cpp:10-155"""
    from visualslam_amd import api
    v, g, H = api.log_gaussian(ctx, [1., 2, 3], [2., 4, 6], np.diag([1., 2, 3]))
    assert v == pytest.approx(-6.04857506884207, rel=1e-13)
    assert np.allclose(g, [1.0, 0.5, 1 / 3.0], rtol=1e-13)
    assert np.allclose(H, np.diag([-1.0, -0.25, -1 / 9.0]), rtol=1e-13, atol=1e-10)
    mu = np.array([np.sqrt(350 * np.log(10) / np.pi)])
    v, _, _ = api.log_gaussian(ctx, np.zeros(1), mu, np.array([[1.0 / np.sqrt(2 * np.pi)]]), False, False)
    assert v == pytest.approx(-805.904782547916, rel=1e-13)
    for a in (1e-4, 1e4):                               # determinant under/overflow
        n = 100
        v, _, _ = api.log_gaussian(ctx, np.zeros(n), np.zeros(n), a * np.eye(n), False, False)
        assert v == pytest.approx(-n * np.log(a) - n / 2.0 * np.log(2 * np.pi), rel=1e-13)


@pytest.mark.parametrize("n", [1, 5, 33, 162, 462, 700])
def test_log_gaussian_vs_oracle(ctx, n):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(n)
    S = np.triu(rng.standard_normal((n, n))) * 0.05 + np.diag(rng.uniform(0.2, 1.5, n) * rng.choice([-1, 1], n))
    x, mu = rng.standard_normal(n), rng.standard_normal(n)
    v, g, H = api.log_gaussian(ctx, x, mu, S)
    v0, g0, H0 = orc.log_gaussian(x, mu, S)
    assert abs(v - v0) <= TOL * abs(v0) and rel_err(g, g0) <= TOL and rel_err(H, H0) <= TOL
    v1, g1, H1 = api.log_gaussian(ctx, x, mu, S, want_g=True, want_H=False)
    assert H1 is None and v1 == v and np.array_equal(g1, g)


@pytest.mark.parametrize("n,m", [(2, 2), (12, 5), (162, 100), (462, 300)])
def test_affine_sqrt_vs_oracle(ctx, n, m):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(n * 7 + m)
    S = np.triu(rng.standard_normal((n, n))) * 0.02 + np.diag(rng.uniform(0.01, 0.5, n))
    C = rng.standard_normal((m, n))
    SR = np.diag(rng.uniform(0.01, 0.1, m))
    assert rel_err(api.affine_sqrt(ctx, S, C, SR), orc.affine_sqrt(S, C, SR)) <= TOL


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("L,seed", [(3, 12), (50, 1023), (150, 3)])
def test_affine_bundle_fused_vs_oracle(ctx, gcam, ocam, layout, L, seed):
    """affineTransform(mu, S, MeasurementPointLandmarkBundle) as at SLAM.cpp:371-387:
    the fused block-sparse S*C^T path must equal the dense reference formula."""
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(L, seed)
    x = sc.x.copy()
    x[3:6] = [0.02, -0.01, 0.03]
    muy, Syy, code = api.affine_bundle(ctx, gcam, x, sc.S, layout)
    h0, SR0, C0, code0 = orc.bundle_measure(x, ocam, layout)
    Syy0 = orc.affine_sqrt(sc.S, C0, SR0)
    assert np.array_equal(code, code0) and rel_err(muy, h0) <= TOL
    assert rel_err(Syy, Syy0) <= TOL
    # and through the unfused device path: bundle kernel -> dense GEMM -> QR
    h, SR, C, _ = api.bundle_measure(ctx, gcam, x, layout)
    assert rel_err(api.affine_sqrt(ctx, sc.S, C, SR), Syy0) <= TOL


@pytest.mark.parametrize("layout", [0, 1])
def test_affine_bundle_bank_vs_oracle(ctx, gcam, ocam, layout):
    """The same transform for a bank of device-resident filters: every filter against the oracle."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    B, L = 5, 40
    n, m = 12 + 3 * L, 2 * L
    scs = [syn.Scene(L, 300 + b) for b in range(B)]
    xs = np.stack([sc.x for sc in scs])
    xs[:, 3:6] = [0.02, -0.01, 0.03]
    dev = torch.device("cuda:0")
    dx = torch.tensor(xs, device=dev)
    dS = torch.tensor(np.stack([np.ascontiguousarray(sc.S.T) for sc in scs]), device=dev)
    dmu = torch.full((B, m), float("nan"), dtype=torch.float64, device=dev)
    dSy = torch.full((B, m, m), float("nan"), dtype=torch.float64, device=dev)
    dc = torch.full((B, L), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()          # the inputs were made on torch's stream; the context may run on one of its own
    api.affine_bundle_batch_dev(ctx, gcam, dx, dS, dmu, dSy, dc, layout)
    ctx.synchronize()
    torch.cuda.synchronize()
    muy, Syy, code = dmu.cpu().numpy(), dSy.cpu().numpy(), dc.cpu().numpy()
    for b in range(B):
        h0, SR0, C0, code0 = orc.bundle_measure(xs[b], ocam, layout)
        Syy0 = orc.affine_sqrt(scs[b].S, C0, SR0)
        assert np.array_equal(code[b], code0) and rel_err(muy[b], h0) <= TOL
        assert rel_err(Syy[b].T, Syy0) <= TOL
        assert not np.tril(Syy[b].T, -1).any()


def test_affine_bundle_large_bank_and_tall_single_vs_oracle(ctx, gcam, ocam):
    """The two forms of the QR that stop at the band under which the stacked diagonal noise factor leaves zeros: a bank of
    more than eight filters (one workgroup per matrix, panels of band + 32 rows) and one tall transform (blocks of 128
    columns resident in LDS, band + 128 rows per launch), each against the dense formula of the oracle."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    B, L = 12, 150
    n, m = 12 + 3 * L, 2 * L
    scs = [syn.Scene(L, 700 + b) for b in range(3)]
    xs = np.stack([scs[b % 3].x for b in range(B)])
    dev = torch.device("cuda:0")
    dx = torch.tensor(xs, device=dev)
    dS = torch.tensor(np.stack([np.ascontiguousarray(scs[b % 3].S.T) for b in range(B)]), device=dev)
    dmu = torch.full((B, m), float("nan"), dtype=torch.float64, device=dev)
    dSy = torch.full((B, m, m), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    api.affine_bundle_batch_dev(ctx, gcam, dx, dS, dmu, dSy, None, 0)
    ctx.synchronize()
    torch.cuda.synchronize()
    Syy = dSy.cpu().numpy()
    for b in (0, 1, 2, 11):
        h0, SR0, C0, _ = orc.bundle_measure(xs[b], ocam, 0)
        assert rel_err(Syy[b].T, orc.affine_sqrt(scs[b % 3].S, C0, SR0)) <= TOL
        assert not np.tril(Syy[b].T, -1).any()
    sc = syn.Scene(400, 5)                                     # 2012 x 800: six resident blocks and a tail of panels
    muy, Syy1, _ = api.affine_bundle(ctx, gcam, sc.x, sc.S, 0)
    h0, SR0, C0, _ = orc.bundle_measure(sc.x, ocam, 0)
    assert rel_err(muy, h0) <= TOL and rel_err(Syy1, orc.affine_sqrt(sc.S, C0, SR0)) <= TOL


def test_condition_on_marginal(ctx):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(3)
    for ny, nx in ((3, 4), (40, 90), (1, 1)):
        S = np.triu(rng.standard_normal((ny + nx, ny + nx))) + 3 * np.eye(ny + nx)
        mu, y = rng.standard_normal(ny + nx), rng.standard_normal(ny)
        m, Sx = api.condition_on_marginal(ctx, mu, S, y)
        m0, Sx0 = orc.condition_on_marginal(mu, S, y)
        assert rel_err(m, m0) <= TOL and np.array_equal(Sx, Sx0)


def test_iekf_post_sqrt_vs_oracle(ctx):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    rng = np.random.default_rng(8)
    for n in (3, 40, 162):
        A = rng.standard_normal((n, n))
        Hs = A @ A.T + n * np.eye(n)
        Q, v = orc.sym_eig(Hs)
        S = api.iekf_post_sqrt(ctx, Q, v)
        assert rel_err(S, orc.iekf_post_sqrt(Q, v)) <= TOL
        assert rel_err(S.T @ S, np.linalg.inv(Hs)) <= 1e-8


@pytest.mark.parametrize("L,seed", [(2, 5), (50, 1023)])
def test_cost_joint_density(ctx, gcam, ocam, L, seed):
    """CostJointDensity (gaussian.hpp:455-466): V = -(logprior + loglik), g, H."""
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(L, seed)
    rng = np.random.default_rng(1)
    x = sc.x + 1e-3 * rng.standard_normal(sc.n)
    V, g, H = api.cost_joint_density(ctx, gcam, x, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    lp, gp, Hp = orc.log_gaussian(x, sc.x, sc.S)
    ll, gl, Hl, _ = orc.point_loglik_dual_sparse(x, sc.y, sc.visible, ocam, sc.sigma)
    assert abs(V + (lp + ll)) <= TOL * abs(lp + ll)
    assert rel_err(g, -(gp + gl)) <= TOL and rel_err(H, -(Hp + Hl)) <= TOL


_ORACLE_UPDATES = {}


def _oracle_update(ocam, L, seed):
    """The CPU restatement's update of Scene(L, seed), computed once per session (5.7 s at 150 landmarks)."""
    from oracle import pyoracle as orc
    from visualslam_amd import synthetic as syn
    if (L, seed) not in _ORACLE_UPDATES:
        sc = syn.Scene(L, seed)
        _ORACLE_UPDATES[(L, seed)] = orc.measurement_update_iekf(sc.x, sc.S, sc.y, sc.visible, ocam, sc.sigma)
    return _ORACLE_UPDATES[(L, seed)]


# Scenes whose last iterations are steps of rounding-noise size (|p| ~ 1e-8: rho = (f - fn) / dm is then a quotient of
# rounding errors of the cost function, and the accept / reject decisions follow whichever arithmetic evaluated it).
# Measured for Scene(150, 3): the CPU restatement runs 12 iterations and ends at |g| = 7.7e-5, both routes of the
# library run 5 and end at |g| = 2.9e-10 with the same cost to 15 digits; the means differ by 7e-7 of the posterior spread.
ROUNDING_DECIDED_TAIL = {(150, 3)}


@pytest.mark.parametrize("solver", ["eig", "cholesky"])
@pytest.mark.parametrize("L,seed", [(2, 41), (5, 41), (20, 7), (50, 1023), (150, 9001), (150, 3)])
def test_measurement_update_iekf_vs_oracle(ctx, gcam, ocam, L, seed, solver):
    """measurementUpdateIEKF (gaussian.hpp:469-532) as called at SLAM.cpp:492, up to the 150-landmark
    configuration (state dimension 462).  The posterior mean must match the oracle's; the posterior
    square-root factor is compared through S^T S and |diag S| because its row signs follow the
    eigenvector signs of the host eigensolver (the reference has the same dependence on Eigen's
    SelfAdjointEigenSolver).  Both routes are checked: "eig" restates the reference's host loop, "cholesky"
    (the default) runs the same iteration on the device from Cholesky factors.  Iteration and evaluation
    counts must equal the oracle's on both routes, except for the scenes listed in ROUNDING_DECIDED_TAIL,
    whose results are compared against the posterior spread instead."""
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(L, seed)
    ctx.set_trs_solver(api.TRS_EIG if solver == "eig" else api.TRS_CHOLESKY)
    try:
        ret, mup, Sp, iters, evals = api.measurement_update_iekf(ctx, gcam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    finally:
        ctx.set_trs_solver(api.TRS_CHOLESKY)        # the library default
    assert ctx.last_trs_route() == (api.TRS_EIG if solver == "eig" else api.TRS_CHOLESKY)
    ret0, mup0, Sp0, iters0, evals0 = _oracle_update(ocam, L, seed)
    assert ret == 0 and ret0 == 0
    if (L, seed) not in ROUNDING_DECIDED_TAIL:
        assert (iters, evals) == (iters0, evals0)
    assert not np.tril(Sp, -1).any()
    if (iters, evals) == (iters0, evals0):
        assert np.max(np.abs(mup - mup0) / np.maximum(np.abs(mup0), sc.diag)) <= 1e-7
        assert rel_err(Sp.T @ Sp, Sp0.T @ Sp0) <= 1e-7
        assert rel_err(np.abs(np.diag(Sp)), np.abs(np.diag(Sp0))) <= 1e-7
    else:
        assert np.linalg.norm(np.linalg.solve(Sp0.T, mup - mup0)) <= 2e-6
        assert rel_err(Sp.T @ Sp, Sp0.T @ Sp0) <= 1e-6
        V_lib, g_lib, _ = api.cost_joint_density(ctx, gcam, mup, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
        V_orc, g_orc, _ = api.cost_joint_density(ctx, gcam, mup0, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
        assert V_lib <= V_orc + 1e-12 * abs(V_orc) and np.linalg.norm(g_lib) <= max(np.linalg.norm(g_orc), 1e-8)   # no worse a minimiser
    # the update must not increase the cost and the posterior information matches the Hessian at the optimum
    V0, _, _ = api.cost_joint_density(ctx, gcam, sc.x, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    V1, g1, H1 = api.cost_joint_density(ctx, gcam, mup, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    assert V1 <= V0
    assert rel_err(np.linalg.inv(Sp.T @ Sp), H1) <= 1e-6


def _indefinite_start(seed=2):
    """A start far from the optimum with a loose prior: the exact likelihood Hessian makes the joint Hessian indefinite
    at the first iterates."""
    from visualslam_amd import synthetic as syn
    sc = syn.Scene(6, 3)
    S = np.diag(np.concatenate([np.full(6, 0.1), np.full(6, 0.5), np.full(18, 5.0)]))
    rng = np.random.default_rng(seed)
    mu = sc.x + np.concatenate([np.zeros(6), 0.05 * rng.standard_normal(6), 0.8 * rng.standard_normal(18)])
    return sc, mu, S


def test_measurement_update_iekf_indefinite_stays_on_the_device(ctx, gcam, ocam):
    """An indefinite Hessian (src/fmin.cpp:59-69 starts the multiplier above -l1 in the eigenbasis): the device route
    meets it as a failed Cholesky factorisation, shifts by the failed pivot until the factorisation succeeds and carries
    on with the same secular equation; the update stays on the device and matches the oracle's eigenvalue route."""
    from oracle import pyoracle as orc
    from visualslam_amd import api
    sc, mu, S = _indefinite_start()
    H0 = api.cost_joint_density(ctx, gcam, mu, mu, S, sc.y, sc.visible, sc.sigma)[2]
    assert np.linalg.eigvalsh(H0)[0] < 0.0                          # the case is what it claims to be
    ret, mup, Sp, iters, evals = api.measurement_update_iekf(ctx, gcam, mu, S, sc.y, sc.visible, sc.sigma)
    ret0, mup0, Sp0, iters0, evals0 = orc.measurement_update_iekf(mu, S, sc.y, sc.visible, ocam, sc.sigma)
    assert ctx.last_trs_route() == api.TRS_CHOLESKY
    assert ret == ret0 and abs(iters - iters0) <= 2 and abs(evals - evals0) <= 2
    assert np.max(np.abs(mup - mup0) / np.maximum(np.abs(mup0), 1e-3)) <= 1e-6
    assert rel_err(Sp.T @ Sp, Sp0.T @ Sp0) <= 1e-6
    assert not np.tril(Sp, -1).any()


def test_bank_with_indefinite_filters_finishes_them(ctx, gcam, ocam):
    """The same in a bank: filters with indefinite Hessians beside well-behaved ones all come back finished
    (status 0) and match the oracle filter by filter."""
    import torch
    from oracle import pyoracle as orc
    from visualslam_amd import api, synthetic as syn
    cases = [_indefinite_start(2), _indefinite_start(5), _indefinite_start(9)]
    sc0 = syn.Scene(6, 3)
    cases.append((sc0, sc0.x.copy(), np.diag(sc0.diag)))            # a positive definite filter among them
    dev = torch.device("cuda:0")
    mu = torch.tensor(np.stack([c[1] for c in cases]), device=dev)
    S = torch.tensor(np.stack([np.ascontiguousarray(c[2].T) for c in cases]), device=dev)
    y = torch.tensor(np.stack([c[0].y for c in cases]), device=dev)
    vis = torch.tensor(np.stack([c[0].visible for c in cases]).astype(np.uint8), device=dev)
    mup, Sp = torch.empty_like(mu), torch.empty_like(S)
    torch.cuda.synchronize()
    code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, mu, S, y, vis, 7.0, mup, Sp)
    torch.cuda.synchronize()
    assert not status.any()
    indefinite = 0
    for b, (sc, m_, S_) in enumerate(cases):
        H0 = api.cost_joint_density(ctx, gcam, m_, m_, S_, sc.y, sc.visible, sc.sigma)[2]
        indefinite += np.linalg.eigvalsh(H0)[0] < 0.0
        ret0, mup0, Sp0, it0, _ = orc.measurement_update_iekf(m_, S_, sc.y, sc.visible, ocam, sc.sigma)
        m1, S1 = mup[b].cpu().numpy(), Sp[b].cpu().numpy().T
        # these updates run 30 and more iterations through indefinite Hessians; the counts may part by a few rejected steps
        assert code[b] == ret0 and abs(int(iters[b]) - it0) <= 8
        assert np.max(np.abs(m1 - mup0) / np.maximum(np.abs(mup0), 1e-3)) <= 1e-6
        assert rel_err(S1.T @ S1, Sp0.T @ Sp0) <= 1e-6
    assert indefinite >= 2


def test_measurement_update_iekf_2000_landmarks(ctx, gcam):
    """The 2,000-landmark update (state dimension 6012) on the opt-in device route.  Checked through
    properties that do not need a CPU run of the same size: the posterior mean is a stationary point of
    the joint cost, the cost went down, and S^T S is the inverse of the Hessian there."""
    from visualslam_amd import api, synthetic as syn
    sc = syn.Scene(2000, 42, dense_S=True)
    rng = np.random.default_rng(5)
    mu = sc.x + np.concatenate([np.zeros(6), 1e-4 * rng.standard_normal(6), 1e-2 * rng.standard_normal(3 * sc.L)])
    ctx.set_trs_solver(api.TRS_CHOLESKY)
    try:
        ret, mup, Sp, iters, evals = api.measurement_update_iekf(ctx, gcam, mu, sc.S, sc.y, sc.visible, sc.sigma)
    finally:
        ctx.set_trs_solver(api.TRS_CHOLESKY)        # the library default
    assert ret == 0 and ctx.last_trs_route() == api.TRS_CHOLESKY
    assert not np.tril(Sp, -1).any()
    V0, g0, _ = api.cost_joint_density(ctx, gcam, mu, mu, sc.S, sc.y, sc.visible, sc.sigma)
    V1, g1, H1 = api.cost_joint_density(ctx, gcam, mup, mu, sc.S, sc.y, sc.visible, sc.sigma)
    assert V1 < V0
    assert np.linalg.norm(g1) <= 1e-7 * np.linalg.norm(g0)
    probe = rng.standard_normal((sc.n, 4))
    back = Sp.T @ (Sp @ (H1 @ probe))                       # (S^T S) H = I on random probes
    assert rel_err(back, probe) <= 1e-8


@pytest.mark.parametrize("case", ["interior", "boundary", "indefinite"])
def test_trs_driver_vs_dense_numpy_solve_on_the_gpu_box(case):
    """The host driver of the eigenvalue route (trs of src/fmin.cpp:40-131 as restated in the library) against an
    independent dense numpy solution, run with the GPU tests so that the library that is loaded here is the one checked."""
    import ctypes as C
    from visualslam_amd import _lib
    import test_host_logic_cpu as H
    lib = _lib.lib()
    lib.vslam_debug_sym_eig.restype = None
    lib.vslam_debug_sym_eig.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.vslam_debug_trs.restype = C.c_int
    lib.vslam_debug_trs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    H.test_trs_host_vs_dense_numpy_solve(lib, case, 162)


def test_structural_shortcuts_change_nothing_beyond_rounding(ctx, gcam, monkeypatch):
    """The forms that skip work on known zeros -- QR panels that stop at the band of the stacked noise factor, the bank's
    Cholesky that updates the trailing matrix two block rows at a time -- against the forms that do not (the library's
    switches are read at every call): the predicted measurement densities and the updated filters of a bank agree to
    rounding."""
    import torch
    from visualslam_amd import api, synthetic as syn
    B, L = 20, 60
    n, m = 12 + 3 * L, 2 * L
    scs = [syn.Scene(L, 900 + b, dense_S=True) for b in range(4)]
    dev = torch.device("cuda:0")
    dx = torch.tensor(np.stack([scs[b % 4].x for b in range(B)]), device=dev)
    dS = torch.tensor(np.stack([np.ascontiguousarray(scs[b % 4].S.T) for b in range(B)]), device=dev)
    dy = torch.tensor(np.stack([scs[b % 4].y for b in range(B)]), device=dev)
    dv = torch.tensor(np.stack([scs[b % 4].visible for b in range(B)]).astype(np.uint8), device=dev)
    torch.cuda.synchronize()

    def density():
        mu = torch.full((B, m), float("nan"), dtype=torch.float64, device=dev)
        Sy = torch.full((B, m, m), float("nan"), dtype=torch.float64, device=dev)
        api.affine_bundle_batch_dev(ctx, gcam, dx, dS, mu, Sy, None, 0)
        ctx.synchronize(); torch.cuda.synchronize()
        return Sy.cpu().numpy()

    def update():
        mup, Sp = torch.empty_like(dx), torch.empty_like(dS)
        code, iters, status = api.measurement_update_iekf_batch_dev(ctx, gcam, dx, dS, dy, dv, 7.0, mup, Sp)
        torch.cuda.synchronize()
        assert not status.any()
        return mup.cpu().numpy(), Sp.cpu().numpy(), np.asarray(iters)

    Sy1 = density()
    mu1, Sp1, it1 = update()
    monkeypatch.setenv("VSLAM_QR_NO_BAND", "1")
    monkeypatch.setenv("VSLAM_POTRF_ONE_LEVEL", "1")
    Sy0 = density()
    mu0, Sp0, it0 = update()
    for b in range(B):
        assert rel_err(Sy1[b], Sy0[b]) <= 1e-12
        assert np.abs(mu1[b] - mu0[b]).max() <= 1e-9 * max(1.0, np.abs(mu0[b]).max())
        assert rel_err(Sp1[b].T @ Sp1[b], Sp0[b].T @ Sp0[b]) <= 1e-9
    assert np.abs(it1 - it0).max() <= 4
