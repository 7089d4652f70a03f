#!/usr/bin/env python3
"""Golden vectors for the time update, independent of the C oracle: the process model of the reference
(src/model.cpp:9-43), one RK4 step of the augmented state (src/gaussian.hpp:171-200) and the Jacobian of
that step with respect to [x; dw], all in 50-digit mpmath arithmetic, the Jacobian by central differences
with a 1e-20 step (truncation error 1e-40, far below fp64).  Also the propagated covariance
P+ = J blkdiag(P, Q dt) J' for a 15-state filter.  Writes tests/golden/time_update.json.

    python tests/golden/make_golden_time_update.py
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50


def f_model(x):
    """SlamProcessModel mean function: x = [nu(6); eta(6); landmarks...] -> dx/dt."""
    phi, theta, psi = x[9], x[10], x[11]
    sph, cph, sth, cth, sps, cps = mp.sin(phi), mp.cos(phi), mp.sin(theta), mp.cos(theta), mp.sin(psi), mp.cos(psi)
    R = [[cps * cth, cps * sth * sph - sps * cph, cps * sth * cph + sps * sph],
         [sps * cth, sps * sth * sph + cps * cph, sps * sth * cph - cps * sph],
         [-sth, cth * sph, cth * cph]]
    tth = sth / cth
    T = [[1, sph * tth, cph * tth], [0, cph, -sph], [0, sph / cth, cph / cth]]
    out = [mp.mpf(0)] * len(x)
    for r in range(3):
        out[6 + r] = sum(R[r][k] * x[k] for k in range(3))
        out[9 + r] = sum(T[r][k] * x[3 + k] for k in range(3))
    return out


def rk4(xdw, dt):
    n = len(xdw) // 2
    x, dw = xdw[:n], xdw[n:]
    add = lambda a, b: [p + q for p, q in zip(a, b)]
    f1 = f_model(x)
    f2 = f_model(add(x, [(f * dt + w) / 2 for f, w in zip(f1, dw)]))
    f3 = f_model(add(x, [(f * dt + w) / 2 for f, w in zip(f2, dw)]))
    f4 = f_model(add(x, [f * dt + w for f, w in zip(f3, dw)]))
    return [x[i] + (f1[i] + 2 * f2[i] + 2 * f3[i] + f4[i]) * dt / 6 + dw[i] for i in range(n)]


def jacobian(xdw, dt):
    h = mp.mpf(10) ** -20
    cols = []
    for k in range(len(xdw)):
        up, dn = list(xdw), list(xdw)
        up[k] += h
        dn[k] -= h
        a, b = rk4(up, dt), rk4(dn, dt)
        cols.append([(p - q) / (2 * h) for p, q in zip(a, b)])
    return cols                                                   # cols[k][i] = d xnext_i / d xdw_k


def main():
    rng = np.random.default_rng(2024)
    cases = []
    for ci in range(4):
        n = 12 if ci < 2 else 15
        x = 0.3 * rng.standard_normal(n)
        x[10] = 0.4 + 0.1 * rng.standard_normal()
        dw = 0.01 * rng.standard_normal(n) if ci % 2 == 0 else np.zeros(n)
        dt = [0.05, 1.0 / 30.0, 0.02, 0.1][ci]
        xdw = [mp.mpf(float(v)) for v in np.concatenate([x, dw])]
        xn = rk4(xdw, mp.mpf(dt))
        J = jacobian(xdw, mp.mpf(dt))
        case = {"n": n, "dt": dt, "xdw": [float(v) for v in xdw], "xnext": [float(v) for v in xn],
                "J_colmajor": [float(v) for col in J for v in col]}
        if n == 15 and ci == 3:
            # covariance propagation for an upper-triangular factor S and the scenario-2 tuning (src/SLAM.cpp:99-106)
            S = np.triu(0.05 * rng.standard_normal((n, n))) + 0.2 * np.eye(n)
            pt, ot = [0.5, 0.5, 0.175], [0.05, 0.05, 0.05]
            Sm = mp.matrix(S.tolist())
            P = Sm.T * Sm
            Jm = mp.matrix(n, 2 * n)
            for k in range(2 * n):
                for i in range(n):
                    Jm[i, k] = J[k][i]
            Qd = mp.zeros(n, n)
            for i in range(3):
                Qd[i, i] = mp.mpf(pt[i]) ** 2 * mp.mpf(dt)
                Qd[3 + i, 3 + i] = mp.mpf(ot[i]) ** 2 * mp.mpf(dt)
            Jx, Jw = Jm[:, :n], Jm[:, n:]
            Pp = Jx * P * Jx.T + Jw * Qd * Jw.T
            case["S_colmajor"] = [float(v) for v in S.T.ravel()]
            case["pos_tune_diag"], case["ori_tune_diag"] = pt, ot
            case["P_next_colmajor"] = [float(Pp[i, j]) for j in range(n) for i in range(n)]
        cases.append(case)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_update.json")
    with open(out, "w") as fh:
        json.dump({"generator": "tests/golden/make_golden_time_update.py (mpmath, 50 digits)", "cases": cases}, fh)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
