"""CPU checks of the oracle's time update (reference src/gaussian.hpp:140-239,337-383, src/model.cpp:9-136).

The reference holds no numbers for this step (test/src/kalman.cpp and processModel.cpp are commented
out; the former only checked output dimensions): parity with the reference is UNPINNED.  The
restatement is checked against what the cited code must satisfy: the analytic process Jacobian equals
finite differences of f, the propagated RK4 Jacobian equals finite differences of the RK4 map, the
factor reproduces the propagated covariance, and the dimensions of test/src/kalman.cpp's case."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import pyoracle as orc

PT = np.diag([0.5, 0.5, 0.175])        # scenario 2, src/SLAM.cpp:99-106
OT = np.diag([0.05, 0.05, 0.05])


def _state(n, seed):
    rng = np.random.default_rng(seed)
    x = 0.3 * rng.standard_normal(n)
    x[10] = 0.4 + 0.1 * rng.standard_normal()          # away from cos(theta) = 0
    return x, rng


def test_process_model_structure_and_jacobian():
    x, _ = _state(21, 0)
    f, SQ, J = orc.process_model(x, PT, OT)
    assert np.all(f[:6] == 0) and np.all(f[12:] == 0)
    R = orc.rpy2rot(x[9:12])
    assert np.allclose(f[6:9], R @ x[0:3], rtol=0, atol=1e-15)
    assert np.array_equal(SQ[:3, :3], PT) and np.array_equal(SQ[3:6, 3:6], OT)
    SQ[:3, :3] = 0; SQ[3:6, 3:6] = 0
    assert not SQ.any()
    Jfd = np.zeros_like(J)
    for k in range(x.size):
        e = np.zeros(x.size); e[k] = 1e-6
        Jfd[:, k] = (orc.process_model(x + e, PT, OT, False, False)[0] - orc.process_model(x - e, PT, OT, False, False)[0]) / 2e-6
    assert np.abs(J - Jfd).max() < 1e-9
    assert not J[:6].any() and not J[12:].any() and not J[:, 6:9].any() and not J[:, 12:].any()


def test_rk4_jacobian_matches_finite_differences():
    n = 18
    x, rng = _state(n, 1)
    xdw = np.concatenate([x, 0.01 * rng.standard_normal(n)])
    dt = 0.05
    xn, J = orc.rk4_sde(xdw, PT, OT, dt)
    Jfd = np.zeros_like(J)
    for k in range(2 * n):
        e = np.zeros(2 * n); e[k] = 1e-6
        Jfd[:, k] = (orc.rk4_sde(xdw + e, PT, OT, dt, False)[0] - orc.rk4_sde(xdw - e, PT, OT, dt, False)[0]) / 2e-6
    assert np.abs(J - Jfd).max() < 1e-9
    # landmark rows and the velocity rows are [e_i, e_i]
    for i in list(range(6)) + list(range(12, n)):
        want = np.zeros(2 * n); want[i] = 1.0; want[n + i] = 1.0
        assert np.array_equal(J[i], want)
    # first-order check of the mean: x + f dt + dw
    f = orc.process_model(x, PT, OT, False, False)[0]
    assert np.abs(xn - (x + f * dt + xdw[n:])).max() < 5e-3


@pytest.mark.parametrize("n,seed", [(12, 2), (18, 3), (42, 4)])
def test_time_update_factor_reproduces_the_covariance(n, seed):
    x, rng = _state(n, seed)
    S = np.triu(0.05 * rng.standard_normal((n, n))) + 0.2 * np.eye(n)
    dt = 1.0 / 30.0
    mup, Sp = orc.time_update(x, S, PT, OT, dt)
    xn, J = orc.rk4_sde(np.concatenate([x, np.zeros(n)]), PT, OT, dt)
    assert np.array_equal(mup, xn)
    SQ = orc.process_model(x, PT, OT)[1]
    P = J[:, :n] @ (S.T @ S) @ J[:, :n].T + dt * J[:, n:] @ (SQ.T @ SQ) @ J[:, n:].T
    assert np.abs(Sp.T @ Sp - P).max() < 1e-13 * max(1.0, np.abs(P).max())
    assert not np.tril(Sp, -1).any()
    # Householder sign convention: the same R as LAPACK's on the same 3n x n matrix
    Sxdw = np.zeros((2 * n, 2 * n)); Sxdw[:n, :n] = S; Sxdw[n:, n:] = SQ * np.sqrt(dt)
    A = np.vstack([Sxdw @ J.T, np.zeros((n, n))])
    R = np.linalg.qr(A, mode="r")
    assert np.abs(Sp - R).max() < 1e-12


def test_kalman_case_dimensions():
    """test/src/kalman.cpp:11-72 (commented out upstream): 2 landmarks, diagonal S, timestep 0.01; the only
    assertion is that the output shapes equal the input shapes."""
    mu = np.arange(1.0, 19.0)
    S = np.diag(np.arange(1.0, 19.0))
    mup, Sp = orc.time_update(mu, S, np.zeros((3, 3)), np.zeros((3, 3)), 0.01)
    assert mup.shape == mu.shape and Sp.shape == S.shape
    assert np.array_equal(mup[12:], mu[12:]) and np.array_equal(mup[:6], mu[:6])
    assert np.all(np.isfinite(Sp))


def test_remove_landmark_keeps_the_marginal():
    """removeBadLandmarks (src/SLAM.cpp:796-820): mean rows / factor columns 12+3j.. leave; R'R is the marginal."""
    rng = np.random.default_rng(5)
    n, j = 24, 2
    mu = rng.standard_normal(n)
    S = np.triu(rng.standard_normal((n, n)))
    mo, So = orc.remove_landmark(mu, S, j)
    keep = [i for i in range(n) if not 18 <= i < 21]
    assert np.array_equal(mo, mu[keep])
    assert np.abs(So.T @ So - (S.T @ S)[np.ix_(keep, keep)]).max() < 1e-12
    assert np.abs(So - np.linalg.qr(S[:, keep], mode="r")).max() < 1e-12       # LAPACK's signs


def test_match_hamming_against_numpy():
    rng = np.random.default_rng(6)
    q = rng.integers(0, 256, (17, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (301, 32), dtype=np.uint8)
    t[200] = q[3]; t[250] = q[3]                     # two exact copies: the first wins
    idx, dist = orc.match_hamming(q, t)
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(idx, d.argmin(axis=1)) and np.array_equal(dist, d.min(axis=1))
    assert idx[3] == 200 and dist[3] == 0
    idx, dist = orc.match_hamming(q, np.zeros((0, 32), np.uint8))
    assert np.all(idx == -1)


def _golden_cases():
    with open(os.path.join(GOLDEN, "time_update.json")) as fh:
        return json.load(fh)["cases"]


def test_rk4_and_jacobian_against_mpmath_golden():
    """tests/golden/time_update.json: the RK4 step and its Jacobian from an independent 50-digit mpmath
    evaluation of the cited formulas (tests/golden/make_golden_time_update.py)."""
    for case in _golden_cases():
        n, dt = case["n"], case["dt"]
        xn, J = orc.rk4_sde(np.array(case["xdw"]), PT, OT, dt)
        want_x = np.array(case["xnext"])
        want_J = np.array(case["J_colmajor"]).reshape(2 * n, n).T
        assert np.abs(xn - want_x).max() <= 1e-14 * max(1.0, np.abs(want_x).max())
        assert np.abs(J - want_J).max() <= 1e-13


def test_time_update_covariance_against_mpmath_golden():
    case = [c for c in _golden_cases() if "P_next_colmajor" in c][0]
    n, dt = case["n"], case["dt"]
    S = np.array(case["S_colmajor"]).reshape(n, n).T
    mu = np.array(case["xdw"][:n])
    mup, Sp = orc.time_update(mu, S, np.diag(case["pos_tune_diag"]), np.diag(case["ori_tune_diag"]), dt)
    P = np.array(case["P_next_colmajor"]).reshape(n, n).T
    assert np.abs(Sp.T @ Sp - P).max() <= 1e-13 * np.abs(P).max()
    assert np.abs(mup - np.array(case["xnext"])).max() <= 1e-14
