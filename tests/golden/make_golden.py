"""Generate high-precision golden vectors for the point-landmark path.

Run once in the build container (needs sympy + mpmath; ~2-3 minutes):

    python tests/golden/make_golden.py

Everything is derived symbolically from the formulas of the reference
(src/cameraModel.hpp:32-216 projection, src/rotation.hpp:22-63 rotations,
src/model.cpp:369-406 + src/gaussian.hpp:30-46 log-likelihood term,
src/measurementPointLandmark.hpp:204-296 Jacobian blocks) and evaluated with
40-digit mpmath arithmetic, then rounded to float64.  The reference itself
cannot be built here (its third-party dependencies are absent), so these
vectors -- not reference outputs -- pin the oracle for the quantities the
reference's own tests leave uncovered.

Outputs (tests/golden/): world_to_pixel.json, landmark_derivs.json,
scenes_small.json, bundle.json.
"""
import json
import os
import sys

import mpmath as mp
import numpy as np
import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from visualslam_amd import synthetic as syn  # noqa: E402

mp.mp.dps = 40

CAM = syn.read_camera_xml()
K1, K2, K3, K4, K5, K6, P1, P2, S1, S2, S3, S4 = syn._coeffs(CAM)
FX, FY, CX, CY = CAM["K"][0, 0], CAM["K"][1, 1], CAM["K"][0, 2], CAM["K"][1, 2]
W, H = CAM["width"], CAM["height"]

# ---------------- symbolic model ----------------
c1, c2, c3, ph, th, ps, r1, r2_, r3 = sp.symbols("c1 c2 c3 ph th ps r1 r2 r3", real=True)
y1, y2, sig = sp.symbols("y1 y2 sig", real=True)
fx, fy, cx, cy = sp.symbols("fx fy cx cy", real=True)
k1, k2, k3, k4, k5, k6, p1, p2, s1, s2, s3, s4 = sp.symbols("k1 k2 k3 k4 k5 k6 p1 p2 s1 s2 s3 s4", real=True)
Z = [c1, c2, c3, ph, th, ps, r1, r2_, r3]
CONSTS = [fx, fy, cx, cy, k1, k2, k3, k4, k5, k6, p1, p2, s1, s2, s3, s4]
CONST_VALS = [FX, FY, CX, CY, K1, K2, K3, K4, K5, K6, P1, P2, S1, S2, S3, S4]

Rx = sp.Matrix([[1, 0, 0], [0, sp.cos(ph), -sp.sin(ph)], [0, sp.sin(ph), sp.cos(ph)]])
Ry = sp.Matrix([[sp.cos(th), 0, sp.sin(th)], [0, 1, 0], [-sp.sin(th), 0, sp.cos(th)]])
Rz = sp.Matrix([[sp.cos(ps), -sp.sin(ps), 0], [sp.sin(ps), sp.cos(ps), 0], [0, 0, 1]])
Rnc = Rz * Ry * Rx
q = Rnc.T * sp.Matrix([r1 - c1, r2_ - c2, r3 - c3])
u, v = q[0] / q[2], q[1] / q[2]
rr2 = u * u + v * v
rr4 = rr2 * rr2
rr6 = rr4 * rr2
cc = (1 + k1 * rr2 + k2 * rr4 + k3 * rr6) / (1 + k4 * rr2 + k5 * rr4 + k6 * rr6)
up = cc * u + p1 * 2 * u * v + p2 * (rr2 + 2 * u * u) + s1 * rr2 + s2 * rr4
vp = cc * v + p2 * 2 * u * v + p1 * (rr2 + 2 * v * v) + s3 * rr2 + s4 * rr4
hx, hy = fx * up + cx, fy * vp + cy
ell = -((y1 - hx) ** 2 + (y2 - hy) ** 2) / (2 * sig ** 2) - sp.log(2 * sp.pi) - 2 * sp.log(sig)


def mpf(x):
    return mp.mpf(float(x))


def build_functions():
    print("differentiating ...", flush=True)
    grad = [sp.diff(ell, z) for z in Z]
    hess = [[sp.diff(grad[i], Z[j]) for j in range(i, 9)] for i in range(9)]
    flat = [ell, hx, hy, q[0], q[1], q[2]] + grad + [e for row in hess for e in row]
    jac_h = [sp.diff(hx, z) for z in Z] + [sp.diff(hy, z) for z in Z]
    args = Z + [y1, y2, sig] + CONSTS
    print("lambdify ...", flush=True)
    f_all = sp.lambdify(args, flat, modules="mpmath", cse=True)
    f_jac = sp.lambdify(args, jac_h, modules="mpmath", cse=True)
    return f_all, f_jac


def eval_landmark(f_all, z, y, sigma):
    a = [mpf(t) for t in z] + [mpf(y[0]), mpf(y[1]), mpf(sigma)] + [mpf(t) for t in CONST_VALS]
    out = f_all(*a)
    val, px, py, qx, qy, qz = out[:6]
    g = out[6:15]
    Hm = [[None] * 9 for _ in range(9)]
    k = 15
    for i in range(9):
        for j in range(i, 9):
            Hm[i][j] = out[k]
            Hm[j][i] = out[k]
            k += 1
    return val, (px, py), (qx, qy, qz), g, Hm


def cull_code(px, qv):
    """Branch structure of src/cameraModel.hpp:63-75,179-181,204-213 in exact arithmetic."""
    qx, qy, qz = qv
    nrm = mp.sqrt(qx * qx + qy * qy + qz * qz)
    if qz / nrm < mp.cos(mp.mpf(120) * ((mp.pi / 180) / 2)):
        return 1
    if qz <= 0:
        return 1
    if not (0 <= px[0] <= W - 1 and 0 <= px[1] <= H - 1):
        return 2
    return 0


def fl(x):
    return float(x)


def main():
    f_all, f_jac = build_functions()
    rng = np.random.default_rng(20240611)

    # ---- 1. world_to_pixel cases ----
    cases = []
    base_eta = np.array([0.0, -5.0, -1.8, -np.pi / 2, np.pi, 0.0])

    def add_case(eta, r, tag):
        z = list(eta) + list(r)
        val, px, qv, g, Hm = eval_landmark(f_all, z, (0.0, 0.0), 1.0)
        code = cull_code(px, qv)
        cases.append(dict(tag=tag, eta=[fl(t) for t in eta], r=[fl(t) for t in r],
                          px=[fl(px[0]), fl(px[1])], code=code))

    sc = syn.Scene(120, 77, dense_S=False)
    for j in range(120):                                   # in view, scenario-2 pose
        add_case(sc.x[6:12], sc.x[12 + 3 * j:15 + 3 * j], "inview")
    for _ in range(60):                                    # random poses, wide spread: all cull classes
        eta = np.concatenate([rng.normal(0, 2, 3), rng.uniform(-np.pi, np.pi, 3)])
        R = syn.rpy2rot(eta[3:])
        qq = np.array([rng.normal(0, 2.0), rng.normal(0, 2.0), rng.uniform(-3, 8)])
        add_case(eta, eta[:3] + R @ qq, "random")
    R0 = syn.rpy2rot(base_eta[3:])
    add_case(base_eta, base_eta[:3] + R0 @ np.array([0.0, 0.0, 4.0]), "optical_axis")
    add_case(base_eta, base_eta[:3] + R0 @ np.array([0.0, 0.0, -4.0]), "behind")
    add_case(base_eta, base_eta[:3] + R0 @ np.array([3.0, 0.0, 1.0]), "outside_cone")
    for t in np.linspace(0.9, 1.6, 15):                    # sweep across the image edge
        add_case(base_eta, base_eta[:3] + R0 @ np.array([t * 4.0, 0.3, 4.0]), "edge_sweep_x")
    for t in np.linspace(0.4, 1.0, 15):
        add_case(base_eta, base_eta[:3] + R0 @ np.array([0.2, t * 4.0, 4.0]), "edge_sweep_y")
    json.dump(dict(camera="visualslam_amd/data/camera.xml", fov_deg=120.0, cases=cases),
              open(os.path.join(HERE, "world_to_pixel.json"), "w"), indent=0)
    print("world_to_pixel:", len(cases), "cases; codes",
          {c: sum(1 for k in cases if k["code"] == c) for c in (0, 1, 2)}, flush=True)

    # ---- 2. per-landmark value / gradient / Hessian ----
    derivs = []
    sc = syn.Scene(10, 5, dense_S=False)
    for sigma in (2.0, 7.0):
        for j in range(10):
            z = list(sc.x[6:12]) + list(sc.x[12 + 3 * j:15 + 3 * j])
            y = sc.y[2 * j:2 * j + 2]
            val, px, qv, g, Hm = eval_landmark(f_all, z, y, sigma)
            derivs.append(dict(z=[fl(t) for t in z], y=[fl(t) for t in y], sigma=sigma,
                               value=fl(val), px=[fl(px[0]), fl(px[1])],
                               grad=[fl(t) for t in g], hess=[[fl(t) for t in row] for row in Hm]))
    # the optical axis (u = v = 0): derivatives stay smooth there
    z = list(base_eta) + list(base_eta[:3] + R0 @ np.array([0.0, 0.0, 5.0]))
    val, px, qv, g, Hm = eval_landmark(f_all, z, (CX + 3.0, CY - 2.0), 7.0)
    derivs.append(dict(z=[fl(t) for t in z], y=[CX + 3.0, CY - 2.0], sigma=7.0, value=fl(val),
                       px=[fl(px[0]), fl(px[1])], grad=[fl(t) for t in g],
                       hess=[[fl(t) for t in row] for row in Hm]))
    json.dump(dict(order="rCNn(3), Theta=(phi,theta,psi), r_j(3)", cases=derivs),
              open(os.path.join(HERE, "landmark_derivs.json"), "w"), indent=0)
    print("landmark_derivs:", len(derivs), flush=True)

    # ---- 3. assembled small scenes ----
    scenes = []
    for L, seed in ((1, 11), (3, 12), (8, 13)):
        sc = syn.Scene(L, seed, dense_S=False)
        if L == 8:
            sc.visible[:] = 1
            sc.visible[2] = 0                                # one invisible landmark
            sc.x[12 + 3 * 5:15 + 3 * 5] = sc.x[6:9] - (sc.x[12 + 3 * 5:15 + 3 * 5] - sc.x[6:9])  # one behind the camera
        n = sc.n
        g = [mp.mpf(0)] * n
        Hs = [[mp.mpf(0)] * n for _ in range(n)]
        val = mp.mpf(0)
        codes = []
        for j in range(L):
            z = list(sc.x[6:12]) + list(sc.x[12 + 3 * j:15 + 3 * j])
            v_, px, qv, gj, Hj = eval_landmark(f_all, z, sc.y[2 * j:2 * j + 2], sc.sigma)
            code = cull_code(px, qv)
            codes.append(code)
            if not sc.visible[j] or code != 0:
                continue
            idx = list(range(6, 12)) + list(range(12 + 3 * j, 15 + 3 * j))
            val += v_
            for a in range(9):
                g[idx[a]] += gj[a]
                for b in range(9):
                    Hs[idx[a]][idx[b]] += Hj[a][b]
        scenes.append(dict(L=L, seed=seed, sigma=sc.sigma, x=[fl(t) for t in sc.x], y=[fl(t) for t in sc.y],
                           visible=[int(t) for t in sc.visible], code=codes, value=fl(val),
                           g=[fl(t) for t in g], H=[[fl(t) for t in row] for row in Hs]))
    json.dump(dict(scenes=scenes), open(os.path.join(HERE, "scenes_small.json"), "w"), indent=0)
    print("scenes_small:", [s["L"] for s in scenes], flush=True)

    # ---- 4. bundle h / C in both layouts ----
    sc = syn.Scene(4, 21, dense_S=False)
    sc.x[3:6] = np.array([0.3, -0.2, 0.5])                   # non-zero angular rates: exercises the COMPAT quirk
    n, L = sc.n, sc.L
    hvec, Cs, Cc = [], [[0.0] * n for _ in range(2 * L)], [[0.0] * n for _ in range(2 * L)]
    # dR/dTheta evaluated at arbitrary angles, for the COMPAT rotation block
    dRs = [sp.lambdify([ph, th, ps], Rz * Ry * Rx.diff(ph), "mpmath"),
           sp.lambdify([ph, th, ps], Rz * Ry.diff(th) * Rx, "mpmath"),
           sp.lambdify([ph, th, ps], Rz.diff(ps) * Ry * Rx, "mpmath")]
    f_q = sp.lambdify(Z, list(q), "mpmath")
    qs = sp.symbols("qx qy qz", real=True)
    uu, vv = qs[0] / qs[2], qs[1] / qs[2]
    t2 = uu * uu + vv * vv
    t4, t6 = t2 * t2, t2 * t2 * t2
    ccq = (1 + k1 * t2 + k2 * t4 + k3 * t6) / (1 + k4 * t2 + k5 * t4 + k6 * t6)
    hq = sp.Matrix([fx * (ccq * uu + p1 * 2 * uu * vv + p2 * (t2 + 2 * uu * uu) + s1 * t2 + s2 * t4) + cx,
                    fy * (ccq * vv + p2 * 2 * uu * vv + p1 * (t2 + 2 * vv * vv) + s3 * t2 + s4 * t4) + cy])
    f_Jq = sp.lambdify(list(qs) + CONSTS, hq.jacobian(list(qs)), "mpmath")
    Rfun = sp.lambdify([ph, th, ps], Rnc, "mpmath")
    for j in range(L):
        z = [mpf(t) for t in list(sc.x[6:12]) + list(sc.x[12 + 3 * j:15 + 3 * j])]
        a = z + [mpf(0), mpf(0), mpf(1)] + [mpf(t) for t in CONST_VALS]
        out = f_all(*a)
        hvec += [fl(out[1]), fl(out[2])]
        jac = f_jac(*a)                                      # STRICT = true Jacobian
        for k in range(2):
            for a_ in range(6):
                Cs[2 * j + k][6 + a_] = fl(jac[9 * k + a_])
            for a_ in range(3):
                Cs[2 * j + k][12 + 3 * j + a_] = fl(jac[9 * k + 6 + a_])
        # COMPAT: src/measurementPointLandmark.hpp:249-296
        qv = f_q(*z)
        JrPCc = f_Jq(*(list(qv) + [mpf(t) for t in CONST_VALS]))
        Rm = Rfun(*z[3:6])
        JrPNn = JrPCc * Rm.T
        d = mp.matrix([z[6] - z[0], z[7] - z[1], z[8] - z[2]])
        w = [mpf(t) for t in sc.x[3:6]]
        for k in range(2):
            for a_ in range(3):
                Cc[2 * j + k][a_] = fl(-JrPNn[k, a_])
                Cc[2 * j + k][6 + 3 * j + a_] = fl(JrPNn[k, a_])
                col = JrPCc * (dRs[a_](*w).T * d)
                Cc[2 * j + k][3 + a_] = fl(col[k])
    json.dump(dict(L=L, x=[fl(t) for t in sc.x], h=hvec, C_strict=Cs, C_compat=Cc, SR_diag=0.01),
              open(os.path.join(HERE, "bundle.json"), "w"), indent=0)
    print("bundle done", flush=True)


if __name__ == "__main__":
    main()
