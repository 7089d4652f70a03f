#!/usr/bin/env python3
"""Times the device pieces of one large filter update (L landmarks, n = 12 + 3L states):
cost evaluation, square-root QRs, fused affine transform.  Host-pointer entry points, so
wall times include PCIe; run under rocprofv3 --kernel-trace --stats for kernel times.

    python tools/run_c4.py [L] [--check]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import api, synthetic as syn  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
    check = "--check" in sys.argv
    sc = syn.Scene(L, 42, dense_S=True)
    n = sc.n
    ctx = api.Context(0)
    cam = api.camera_from_xml(syn.CAMERA_XML)
    rng = np.random.default_rng(5)
    mu = sc.x + np.concatenate([np.zeros(6), 1e-4 * rng.standard_normal(6), 1e-2 * rng.standard_normal(3 * L)])

    def timed(name, f, reps=3):
        f()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = f()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-28s %10.3f ms" % (name, dt * 1e3), flush=True)
        return out

    print("L=%d n=%d" % (L, n), flush=True)
    v, g, H = timed("point_loglik (v,g,H)", lambda: api.point_loglik(ctx, cam, sc.x, sc.y, sc.visible, sc.sigma)[:3])
    timed("log_gaussian (v,g,H)", lambda: api.log_gaussian(ctx, sc.x, mu, sc.S))
    V, G, HH = timed("cost_joint_density", lambda: api.cost_joint_density(ctx, cam, sc.x, mu, sc.S, sc.y, sc.visible, sc.sigma))
    A = np.asfortranarray(rng.standard_normal((n, n)))
    R = timed("qr_upper n x n", lambda: api.qr_upper(ctx, A))
    muy, Syy, code = timed("affine_bundle", lambda: api.affine_bundle(ctx, cam, sc.x, sc.S))
    Q = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, n)))[0]) if n <= 3100 else None
    if Q is not None:
        vv = rng.uniform(0.5, 50.0, n)
        Sp = timed("iekf_post_sqrt", lambda: api.iekf_post_sqrt(ctx, Q, vv))
    if check:
        t0 = time.perf_counter()
        R0 = np.linalg.qr(A, mode="r")
        print("numpy qr %.1f ms" % ((time.perf_counter() - t0) * 1e3))
        sgn = np.sign(np.diag(R)) * np.sign(np.diag(R0))
        err = np.abs(R - sgn[:, None] * R0).max() / np.abs(R0).max()
        print("qr_upper vs LAPACK (row-sign aligned): %.2e, min sign agreement %d/%d" % (err, int((sgn > 0).sum()), n))
        err2 = np.abs(R.T @ R - A.T @ A).max() / np.abs(A.T @ A).max()
        print("R'R vs A'A: %.2e" % err2)
        print("H symmetric: %.2e" % (np.abs(H - H.T).max() / np.abs(H).max()))
    ctx.close()


if __name__ == "__main__":
    main()
