// Per-landmark fp64 math of the point-landmark measurement model: projection,
// Jacobian blocks and the exact log-likelihood derivatives in a compact block
// form.  Device code (gfx950); every function is forced inline so one lane
// evaluates one landmark entirely in registers.
//
// Model (reference src/cameraModel.hpp:32-216, src/rotation.hpp:22-63):
//   R = Rz(psi) Ry(theta) Rx(phi),  q = R^T (r - c),  (u,v) = (qx,qy)/qz,
//   D(u,v) rational + tangential + thin-prism distortion,  h = K D + c0.
// Log-likelihood term (reference src/model.cpp:392-404, src/gaussian.hpp:30-46):
//   l = -|y - h|^2 / (2 s^2) - log(2 pi) - 2 log s.
//
// Derivatives w.r.t. z = (c, Theta, r) use the chain h = g(q(z)):
//   w = Jg^T e / s^2,   M = (-Jg^T Jg + sum_k e_k Hess g_k) / s^2   (3x3, q-space)
//   grad_r = R w, grad_c = -R w, grad_Theta_k = A_k . w
//   H_rr = H_cc = N = R M R^T, H_cr = -N
//   H_rTheta = E = R M A + [D_k],  H_cTheta = -E
//   H_ThetaTheta = F = A^T M A + [B_kl . w]
// with A_k = (dR/dTheta_k)^T d, B_kl = (d2R/dTheta_k dTheta_l)^T d,
// D_k = (dR/dTheta_k) w, d = r - c.  The axis rotations' derivatives are the
// same plane rotation advanced by 90 degrees per order, so every product is a
// chain of three 2-D rotations.
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"

namespace vslam {

#define VSLAM_DEV __device__ __forceinline__

struct V3 { double x, y, z; };

struct Pose {
    double c[3];              // camera position rCNn
    double sp, cp, st, ct, ss, cs;   // sin/cos of phi, theta, psi
};

VSLAM_DEV Pose make_pose(const double *eta)
{
    Pose p;
    p.c[0] = eta[0]; p.c[1] = eta[1]; p.c[2] = eta[2];
    sincos(eta[3], &p.sp, &p.cp);
    sincos(eta[4], &p.st, &p.ct);
    sincos(eta[5], &p.ss, &p.cs);
    return p;
}

// Same, with the three sincos evaluated by lanes 0..2 of the wavefront in parallel
// and broadcast: a third of the dependent-instruction chain.  All 64 lanes must call.
VSLAM_DEV Pose make_pose_wave(const double *eta)
{
    const int lane = threadIdx.x & 63;
    double s = 0.0, c = 0.0;
    if (lane < 3) sincos(eta[3 + lane], &s, &c);
    Pose p;
    p.c[0] = eta[0]; p.c[1] = eta[1]; p.c[2] = eta[2];
    p.sp = __shfl(s, 0, 64); p.cp = __shfl(c, 0, 64);
    p.st = __shfl(s, 1, 64); p.ct = __shfl(c, 1, 64);
    p.ss = __shfl(s, 2, 64); p.cs = __shfl(c, 2, 64);
    return p;
}

// cos/sin of (angle + K*90deg): the K-th derivative of a plane rotation.
template <int K> VSLAM_DEV void ord(double c, double s, double &ck, double &sk)
{
    if (K == 0) { ck = c; sk = s; }
    else if (K == 1) { ck = -s; sk = c; }
    else { ck = -c; sk = -s; }
}

// Transposed applications: (d^K R_axis / d angle^K)^T t
template <int K> VSLAM_DEV V3 RzT(const Pose &p, V3 t)
{
    double ck, sk; ord<K>(p.cs, p.ss, ck, sk);
    V3 o; o.x = ck * t.x + sk * t.y; o.y = -sk * t.x + ck * t.y; o.z = (K == 0) ? t.z : 0.0; return o;
}
template <int K> VSLAM_DEV V3 RyT(const Pose &p, V3 t)
{
    double ck, sk; ord<K>(p.ct, p.st, ck, sk);
    V3 o; o.z = ck * t.z + sk * t.x; o.x = -sk * t.z + ck * t.x; o.y = (K == 0) ? t.y : 0.0; return o;
}
template <int K> VSLAM_DEV V3 RxT(const Pose &p, V3 t)
{
    double ck, sk; ord<K>(p.cp, p.sp, ck, sk);
    V3 o; o.y = ck * t.y + sk * t.z; o.z = -sk * t.y + ck * t.z; o.x = (K == 0) ? t.x : 0.0; return o;
}
// Forward applications: (d^K R_axis / d angle^K) t
template <int K> VSLAM_DEV V3 Rzf(const Pose &p, V3 t)
{
    double ck, sk; ord<K>(p.cs, p.ss, ck, sk);
    V3 o; o.x = ck * t.x - sk * t.y; o.y = sk * t.x + ck * t.y; o.z = (K == 0) ? t.z : 0.0; return o;
}
template <int K> VSLAM_DEV V3 Ryf(const Pose &p, V3 t)
{
    double ck, sk; ord<K>(p.ct, p.st, ck, sk);
    V3 o; o.z = ck * t.z - sk * t.x; o.x = sk * t.z + ck * t.x; o.y = (K == 0) ? t.y : 0.0; return o;
}
template <int K> VSLAM_DEV V3 Rxf(const Pose &p, V3 t)
{
    double ck, sk; ord<K>(p.cp, p.sp, ck, sk);
    V3 o; o.y = ck * t.y - sk * t.z; o.z = sk * t.y + ck * t.z; o.x = (K == 0) ? t.x : 0.0; return o;
}

VSLAM_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VSLAM_DEV V3 Rfwd(const Pose &p, V3 t) { return Rzf<0>(p, Ryf<0>(p, Rxf<0>(p, t))); }

// Distortion map and (optionally) its first/second partials w.r.t. (u,v),
// written through r2 = u^2 + v^2 so they stay smooth on the optical axis.
struct Distort {
    double up, vp;             // D(u,v)
    double j00, j01, j10, j11; // dD/d(u,v)
    double huu, huv, hvv;      // Hessian of up
    double guu, guv, gvv;      // Hessian of vp
};
template <int ORDER> VSLAM_DEV void distort(const CamK &k, double u, double v, Distort &o)
{
    double u2 = u * u, v2 = v * v, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2;
    double N = 1.0 + k.k1 * r2 + k.k2 * r4 + k.k3 * r6;
    double Dn = 1.0 + k.k4 * r2 + k.k5 * r4 + k.k6 * r6;
    double c = N / Dn;
    double uv = u * v;
    o.up = c * u + k.p1 * 2 * uv + k.p2 * (r2 + 2 * u2) + k.s1 * r2 + k.s2 * r4;
    o.vp = c * v + k.p2 * 2 * uv + k.p1 * (r2 + 2 * v2) + k.s3 * r2 + k.s4 * r4;
    if (ORDER >= 1) {
        double N1 = k.k1 + 2 * k.k2 * r2 + 3 * k.k3 * r4, D1 = k.k4 + 2 * k.k5 * r2 + 3 * k.k6 * r4;
        double c1 = (N1 - c * D1) / Dn;
        o.j00 = c + 2 * u2 * c1 + 2 * k.p1 * v + 6 * k.p2 * u + 2 * k.s1 * u + 4 * k.s2 * r2 * u;
        o.j01 = 2 * uv * c1 + 2 * k.p1 * u + 2 * k.p2 * v + 2 * k.s1 * v + 4 * k.s2 * r2 * v;
        o.j10 = 2 * uv * c1 + 2 * k.p2 * v + 2 * k.p1 * u + 2 * k.s3 * u + 4 * k.s4 * r2 * u;
        o.j11 = c + 2 * v2 * c1 + 2 * k.p2 * u + 6 * k.p1 * v + 2 * k.s3 * v + 4 * k.s4 * r2 * v;
        if (ORDER >= 2) {
            double N2 = 2 * k.k2 + 6 * k.k3 * r2, D2 = 2 * k.k5 + 6 * k.k6 * r2;
            double c2 = (N2 - 2 * c1 * D1 - c * D2) / Dn;
            o.huu = 6 * u * c1 + 4 * u2 * u * c2 + 6 * k.p2 + 2 * k.s1 + 4 * k.s2 * r2 + 8 * k.s2 * u2;
            o.huv = 2 * v * c1 + 4 * u2 * v * c2 + 2 * k.p1 + 8 * k.s2 * uv;
            o.hvv = 2 * u * c1 + 4 * u * v2 * c2 + 2 * k.p2 + 2 * k.s1 + 4 * k.s2 * r2 + 8 * k.s2 * v2;
            o.guu = 2 * v * c1 + 4 * v * u2 * c2 + 2 * k.p1 + 2 * k.s3 + 4 * k.s4 * r2 + 8 * k.s4 * u2;
            o.guv = 2 * u * c1 + 4 * v2 * u * c2 + 2 * k.p2 + 8 * k.s4 * uv;
            o.gvv = 6 * v * c1 + 4 * v2 * v * c2 + 6 * k.p1 + 2 * k.s3 + 4 * k.s4 * r2 + 8 * k.s4 * v2;
        }
    }
}

// Cull tests of worldToPixel (reference src/cameraModel.hpp:63-75,179-181,204-213).
// Returns the code; px is written for codes 0 and 2 and NaN for code 1.
VSLAM_DEV int project_q(const CamK &k, V3 q, double *px, Distort *dist_out, double *u_out, double *v_out)
{
    double nrm = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    px[0] = px[1] = __builtin_nan("");
    if (q.z / nrm < k.cos_half_fov) return 1;
    if (q.z <= 0.0) return 1;
    double u = q.x / q.z, v = q.y / q.z;
    Distort d;
    distort<2>(k, u, v, d);
    px[0] = k.fx * d.up + k.cx;
    px[1] = k.fy * d.vp + k.cy;
    if (dist_out) *dist_out = d;
    if (u_out) { *u_out = u; *v_out = v; }
    bool inW = 0.0 <= px[0] && px[0] <= k.wmax;
    bool inH = 0.0 <= px[1] && px[1] <= k.hmax;
    return (inW && inH) ? 0 : 2;
}

VSLAM_DEV V3 landmark_offset(const Pose &p, const double *r)
{
    V3 d; d.x = r[0] - p.c[0]; d.y = r[1] - p.c[1]; d.z = r[2] - p.c[2]; return d;
}

VSLAM_DEV int project_landmark(const CamK &k, const Pose &p, const double *r, double *px)
{
    V3 q = RxT<0>(p, RyT<0>(p, RzT<0>(p, landmark_offset(p, r))));
    return project_q(k, q, px, nullptr, nullptr, nullptr);
}

// Compact per-landmark record: 28 doubles.
struct Terms {
    double ell;
    double gr[3];   // d l / d r        (d l / d c = -gr)
    double gT[3];   // d l / d Theta
    double N[6];    // xx xy xz yy yz zz
    double E[9];    // E[i*3+k] = d2 l / d r_i d Theta_k
    double F[6];    // pp pt ps tt ts ss  (phi, theta, psi)
    int code;
};

VSLAM_DEV void terms_zero(Terms &t)
{
    t.ell = 0.0;
    for (int i = 0; i < 3; ++i) { t.gr[i] = 0.0; t.gT[i] = 0.0; }
    for (int i = 0; i < 6; ++i) { t.N[i] = 0.0; t.F[i] = 0.0; }
    for (int i = 0; i < 9; ++i) t.E[i] = 0.0;
}

// ORDER 0: value only, 1: + gradient, 2: + Hessian.
template <int ORDER>
VSLAM_DEV void landmark_terms(const CamK &k, const Pose &p, const double *r, const double *y, bool visible,
                              double is2, double lconst, Terms &t)
{
    terms_zero(t);
    V3 d = landmark_offset(p, r);
    V3 Z0 = RzT<0>(p, d);
    V3 Y00 = RyT<0>(p, Z0);
    V3 q = RxT<0>(p, Y00);
    double px[2], u, v;
    Distort D;
    t.code = project_q(k, q, px, &D, &u, &v);
    if (t.code != 0 || !visible) return;
    double e0 = y[0] - px[0], e1 = y[1] - px[1];
    t.ell = -0.5 * (e0 * e0 + e1 * e1) * is2 + lconst;
    if (ORDER == 0) return;

    // eps = Jp^T e / s^2 (2-vector), w = Ja^T eps (3-vector in q-space)
    double a00 = k.fx * D.j00, a01 = k.fx * D.j01, a10 = k.fy * D.j10, a11 = k.fy * D.j11;
    double se0 = e0 * is2, se1 = e1 * is2;
    double eu = a00 * se0 + a10 * se1, ev = a01 * se0 + a11 * se1;
    double iz = 1.0 / q.z;
    V3 w; w.x = iz * eu; w.y = iz * ev; w.z = -iz * (u * eu + v * ev);

    V3 Z1 = RzT<1>(p, d);
    V3 Y01 = RyT<1>(p, Z0), Y10 = RyT<0>(p, Z1);
    V3 A[3];
    A[0] = RxT<1>(p, Y00);   // phi
    A[1] = RxT<0>(p, Y01);   // theta
    A[2] = RxT<0>(p, Y10);   // psi
    V3 gr = Rfwd(p, w);
    t.gr[0] = gr.x; t.gr[1] = gr.y; t.gr[2] = gr.z;
    for (int a = 0; a < 3; ++a) t.gT[a] = dot(A[a], w);
    if (ORDER == 1) return;

    // M2 = (-Jp^T Jp + e0 fx Hess(up) + e1 fy Hess(vp)) / s^2   (2x2 sym)
    double m11 = -(a00 * a00 + a10 * a10) * is2 + se0 * k.fx * D.huu + se1 * k.fy * D.guu;
    double m12 = -(a00 * a01 + a10 * a11) * is2 + se0 * k.fx * D.huv + se1 * k.fy * D.guv;
    double m22 = -(a01 * a01 + a11 * a11) * is2 + se0 * k.fx * D.hvv + se1 * k.fy * D.gvv;
    // M = Ja^T M2 Ja + eu Hess(u) + ev Hess(v)   (3x3 sym)
    double iz2 = iz * iz;
    double Mxx = iz2 * m11, Mxy = iz2 * m12, Myy = iz2 * m22;
    double Mxz = -iz2 * (m11 * u + m12 * v + eu);
    double Myz = -iz2 * (m12 * u + m22 * v + ev);
    double Mzz = iz2 * (m11 * u * u + 2 * m12 * u * v + m22 * v * v + 2 * (eu * u + ev * v));

    // G = R M (columns of M rotated), N = G R^T = R (rows of G)
    V3 m0 = {Mxx, Mxy, Mxz}, m1 = {Mxy, Myy, Myz}, m2 = {Mxz, Myz, Mzz};
    V3 g0 = Rfwd(p, m0), g1 = Rfwd(p, m1), g2 = Rfwd(p, m2);   // G = [g0 g1 g2]
    V3 r0 = {g0.x, g1.x, g2.x}, r1 = {g0.y, g1.y, g2.y}, r2 = {g0.z, g1.z, g2.z};   // rows of G
    V3 n0 = Rfwd(p, r0), n1 = Rfwd(p, r1), n2 = Rfwd(p, r2);   // N columns (N symmetric)
    t.N[0] = n0.x; t.N[1] = n0.y; t.N[2] = n0.z; t.N[3] = n1.y; t.N[4] = n1.z; t.N[5] = n2.z;

    // MA_k = M A_k ;  E[:,k] = R MA_k + (dR/dTheta_k) w ;  F_kl = A_k . MA_l + B_kl . w
    V3 MA[3];
    for (int a = 0; a < 3; ++a) {
        MA[a].x = Mxx * A[a].x + Mxy * A[a].y + Mxz * A[a].z;
        MA[a].y = Mxy * A[a].x + Myy * A[a].y + Myz * A[a].z;
        MA[a].z = Mxz * A[a].x + Myz * A[a].y + Mzz * A[a].z;
    }
    V3 Dk[3];
    Dk[0] = Rzf<0>(p, Ryf<0>(p, Rxf<1>(p, w)));
    Dk[1] = Rzf<0>(p, Ryf<1>(p, Rxf<0>(p, w)));
    Dk[2] = Rzf<1>(p, Ryf<0>(p, Rxf<0>(p, w)));
    for (int a = 0; a < 3; ++a) {
        V3 c = Rfwd(p, MA[a]);
        t.E[0 * 3 + a] = c.x + Dk[a].x;
        t.E[1 * 3 + a] = c.y + Dk[a].y;
        t.E[2 * 3 + a] = c.z + Dk[a].z;
    }
    V3 Z2 = RzT<2>(p, d);
    V3 Y02 = RyT<2>(p, Z0), Y11 = RyT<1>(p, Z1), Y20 = RyT<0>(p, Z2);
    V3 Bpp = RxT<2>(p, Y00), Bpt = RxT<1>(p, Y01), Bps = RxT<1>(p, Y10);
    V3 Btt = RxT<0>(p, Y02), Bts = RxT<0>(p, Y11), Bss = RxT<0>(p, Y20);
    t.F[0] = dot(A[0], MA[0]) + dot(Bpp, w);
    t.F[1] = dot(A[0], MA[1]) + dot(Bpt, w);
    t.F[2] = dot(A[0], MA[2]) + dot(Bps, w);
    t.F[3] = dot(A[1], MA[1]) + dot(Btt, w);
    t.F[4] = dot(A[1], MA[2]) + dot(Bts, w);
    t.F[5] = dot(A[2], MA[2]) + dot(Bss, w);
}

// N as a full symmetric 3x3 lookup
VSLAM_DEV double Nij(const Terms &t, int i, int j)
{
    const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return t.N[idx[i][j]];
}

// Wave-wide (64-lane) sum, result in every lane; fixed butterfly order.
VSLAM_DEV double wave_sum(double v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace vslam
