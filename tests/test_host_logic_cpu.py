"""Host-side logic of libvslam_hip that needs no GPU: the symmetric eigensolver and the trust-region
subproblem of the eigenvalue route (src/fmin.hpp:42-44, src/fmin.cpp:40-131), chi2inv / normcdf
(src/utility.cpp:13-24).  Checked against numpy and against the CPU oracle's restatement."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from visualslam_amd import _lib
    lib = _lib.lib()
    lib.vslam_debug_sym_eig.restype = None
    lib.vslam_debug_sym_eig.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.vslam_debug_trs.restype = C.c_int
    lib.vslam_debug_trs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    return lib


def _eig(L, H):
    n = H.shape[0]
    H = np.asfortranarray(H)
    Q, v = np.zeros((n, n), order="F"), np.zeros(n)
    L.vslam_debug_sym_eig(H.ctypes.data, n, Q.ctypes.data, v.ctypes.data)
    return Q, v


@pytest.mark.parametrize("n", [1, 2, 3, 18, 162, 300])
def test_sym_eig_host(L, n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    H = A + A.T + np.diag(rng.uniform(-2, 6, n))              # indefinite, like a Hessian far from the optimum
    Q, v = _eig(L, np.tril(H) + np.triu(rng.standard_normal((n, n)), 1))      # only the lower triangle may be read
    w = np.linalg.eigvalsh(H)
    assert np.all(np.diff(v) >= 0)                             # ascending, like Eigen's SelfAdjointEigenSolver
    assert np.max(np.abs(v - w)) <= 1e-12 * max(1.0, np.max(np.abs(w)))
    assert np.max(np.abs(Q.T @ Q - np.eye(n))) <= 1e-12
    assert np.max(np.abs(H @ Q - Q * v)) <= 1e-11 * max(1.0, np.max(np.abs(w)))


def test_sym_eig_host_baseline_variant_agrees(L):
    """The solver is built for the baseline x86-64 instruction set and for AVX2+FMA; the dispatcher picks one
    per CPU.  Both must give the same decomposition up to rounding (and eigenvector signs)."""
    n = 90
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, n))
    H = np.asfortranarray(A + A.T)
    Q1, v1 = _eig(L, H)
    Q0, v0 = np.zeros((n, n), order="F"), np.zeros(n)
    L.vslam_debug_sym_eig_base.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.vslam_debug_sym_eig_base.restype = None
    L.vslam_debug_sym_eig_base(H.ctypes.data, n, Q0.ctypes.data, v0.ctypes.data)
    assert np.max(np.abs(v0 - v1)) <= 1e-12 * np.max(np.abs(v1))
    assert np.max(np.abs(np.abs(Q0.T @ Q1) - np.eye(n))) <= 1e-9
    assert np.max(np.abs(H @ Q0 - Q0 * v0)) <= 1e-11 * np.max(np.abs(v0))


def test_sym_eig_host_repeated_and_diagonal(L):
    Q, v = _eig(L, np.diag([3.0, 1.0, 1.0, 2.0]))
    assert v.tolist() == [1.0, 1.0, 2.0, 3.0]
    assert np.max(np.abs(np.abs(Q) @ np.ones(4) - 1.0)) <= 1e-15      # a permutation up to signs


@pytest.mark.parametrize("case", ["interior", "boundary", "indefinite", "hard"])
def test_trs_host_vs_oracle(L, case):
    """Same step as the oracle's restatement of trs, and the optimality conditions of the subproblem."""
    from oracle import pyoracle as orc
    rng = np.random.default_rng(7)
    n = 12
    A = rng.standard_normal((n, n))
    H = A @ A.T + 0.5 * np.eye(n)
    g = rng.standard_normal(n)
    D = 10.0
    if case == "boundary":
        D = 0.05
    if case in ("indefinite", "hard"):
        H = H - 6.0 * np.eye(n)
    Q, v = _eig(L, H)
    if case == "hard":
        g = 0.01 * (g - Q[:, 0] * (Q[:, 0] @ g))               # no component along the lowest eigenvector: |q0'g| < eps
    p = np.zeros(n)
    ret = L.vslam_debug_trs(Q.ctypes.data, v.ctypes.data, g.ctypes.data, n, D, p.ctypes.data)
    ret0, p0 = orc.trs(Q, v, g, D)
    assert ret == ret0
    if case != "hard":
        assert np.max(np.abs(p - p0)) <= 1e-10 * max(1.0, np.max(np.abs(p0)))
    assert np.linalg.norm(p) <= D * (1 + 1e-6)
    if case == "interior":
        assert np.max(np.abs(H @ p + g)) <= 1e-9
    else:
        assert abs(np.linalg.norm(p) - D) <= 1e-6 * D          # the constraint is active
    model = lambda q: 0.5 * q @ H @ q + g @ q
    for _ in range(50):                                        # no feasible point does better
        q = rng.standard_normal(n)
        q *= D * rng.uniform(0, 1) / np.linalg.norm(q)
        assert model(p) <= model(q) + 1e-9 * (1 + abs(model(q)))


def _trs_dense(H, g, D):
    """The trust-region subproblem min g'p + p'Hp/2, |p| <= D solved without the eigenbasis formulas of the driver:
    the multiplier by bisection on |(H + lam I)^-1 g| = D with dense solves (numpy only, nothing shared with the
    library or the oracle)."""
    n = g.size
    w = np.linalg.eigvalsh(H)
    if w[0] > 0:
        p = -np.linalg.solve(H, g)
        if np.linalg.norm(p) <= D:
            return p
    lo = max(0.0, -w[0]) + 1e-14 * max(1.0, abs(w[0]))
    hi = lo + 1.0
    norm_at = lambda lam: np.linalg.norm(np.linalg.solve(H + lam * np.eye(n), g))
    while norm_at(hi) > D:
        hi = lo + 2.0 * (hi - lo)
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if norm_at(mid) > D:
            lo = mid
        else:
            hi = mid
    return -np.linalg.solve(H + hi * np.eye(n), g)


@pytest.mark.parametrize("n", [12, 162])
@pytest.mark.parametrize("case", ["interior", "boundary", "indefinite"])
def test_trs_host_vs_dense_numpy_solve(L, case, n):
    """The step of the driver's trs against an independent dense solution of the same subproblem."""
    rng = np.random.default_rng(11 + n)
    A = rng.standard_normal((n, n))
    H = A @ A.T / n + 0.5 * np.eye(n)
    g = rng.standard_normal(n)
    D = 100.0
    if case == "boundary":
        D = 0.05
    if case == "indefinite":
        H = H - 3.0 * np.eye(n)
        D = 0.7
    Q, v = _eig(L, H)
    p = np.zeros(n)
    L.vslam_debug_trs(Q.ctypes.data, v.ctypes.data, g.ctypes.data, n, D, p.ctypes.data)
    p0 = _trs_dense(H, g, D)
    assert np.max(np.abs(p - p0)) <= 1e-6 * max(1.0, np.max(np.abs(p0)))      # the driver stops its multiplier iteration at sqrt(eps)
    assert abs(np.linalg.norm(p) - (np.linalg.norm(p0))) <= 1e-6 * D


def test_chi2inv_normcdf_host(L):
    from oracle import pyoracle as orc
    from visualslam_amd import api
    for p in (0.01, 0.5, 0.9973002039367398, 0.999999):
        for nu in (1, 2, 3, 10, 300, 4000):
            a, b = api.chi2inv(p, nu), orc.chi2inv(p, nu)
            assert abs(a - b) <= 1e-12 * b
    for z in (-8.0, -1.0, 0.0, 0.3, 3.0, 8.0):
        assert api.normcdf(z) == orc.normcdf(z)
