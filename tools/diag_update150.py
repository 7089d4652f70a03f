"""Diagnostic: the 150-landmark update on both routes against the CPU oracle -- iteration counts, cost at the
result, gradient norm there, distance between the means in units of the posterior spread."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as orc
from visualslam_amd import api, synthetic as syn

L, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = syn.read_camera_xml()
ocam = orc.make_camera(c["K"][0, 0], c["K"][1, 1], c["K"][0, 2], c["K"][1, 2], c["dist"], c["flag"], c["width"], c["height"])
gcam = api.camera_from_xml(syn.CAMERA_XML)
ctx = api.Context(0)
sc = syn.Scene(L, seed)
ret0, m0, S0, it0, ev0 = orc.measurement_update_iekf(sc.x, sc.S, sc.y, sc.visible, ocam, sc.sigma)
V0, g0, _ = api.cost_joint_density(ctx, gcam, m0, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
print("oracle   ret %d iters %d evals %d  V %.15g |g| %.3g" % (ret0, it0, ev0, V0, np.linalg.norm(g0)))
for name, solver in (("eig", api.TRS_EIG), ("cholesky", api.TRS_CHOLESKY)):
    ctx.set_trs_solver(solver)
    ret, m, S, it, ev = api.measurement_update_iekf(ctx, gcam, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    V, g, _ = api.cost_joint_density(ctx, gcam, m, sc.x, sc.S, sc.y, sc.visible, sc.sigma)
    z = np.linalg.solve(S0.T, m - m0)
    print("%-8s ret %d iters %d evals %d  V %.15g |g| %.3g  |S^-T dm| %.3g  max|dm|/diag %.3g  SS rel %.3g" % (
        name, ret, it, ev, V, np.linalg.norm(g), np.linalg.norm(z), np.max(np.abs(m - m0) / np.maximum(np.abs(m0), sc.diag)),
        np.max(np.abs(S.T @ S - S0.T @ S0)) / np.max(np.abs(S0.T @ S0))))
