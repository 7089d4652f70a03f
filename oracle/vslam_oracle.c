/*
 * CPU oracle for the VisualSLAM point-landmark hot path -- see vslam_oracle.h.
 * TEST INFRASTRUCTURE ONLY; never linked into the shipped library.
 *
 * Plain C, scalar, single-threaded (the reference has no threading).  Every
 * function cites the reference lines it restates (paths relative to the
 * reference checkout).  No reference source text is reproduced here.
 */
#include "vslam_oracle.h"

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327950288
#endif

#define ORC_RATIONAL 0x4000   /* cv::CALIB_RATIONAL_MODEL */
#define ORC_THINPRISM 0x8000  /* cv::CALIB_THIN_PRISM_MODEL */

/* ------------------------------------------------------------------ */
/* small 3x3 helpers (column-major, M[r + 3*c])                        */
/* ------------------------------------------------------------------ */
static void mat3_mul(const double *A, const double *B, double *C)
{
    double T[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += A[r + 3 * k] * B[k + 3 * c];
            T[r + 3 * c] = s;
        }
    memcpy(C, T, sizeof T);
}
static void mat3_tvec(const double *A, const double *v, double *out) /* out = A^T v */
{
    for (int c = 0; c < 3; ++c)
        out[c] = A[0 + 3 * c] * v[0] + A[1 + 3 * c] * v[1] + A[2 + 3 * c] * v[2];
}
static void __attribute__((unused)) mat3_vec(const double *A, const double *v, double *out) /* out = A v */
{
    for (int r = 0; r < 3; ++r)
        out[r] = A[r + 0] * v[0] + A[r + 3] * v[1] + A[r + 6] * v[2];
}

/* src/rotation.hpp:22-30 */
void orc_rotx(double a, double *R)
{
    double c = cos(a), s = sin(a);
    R[0] = 1; R[3] = 0; R[6] = 0;
    R[1] = 0; R[4] = c; R[7] = -s;
    R[2] = 0; R[5] = s; R[8] = c;
}
/* src/rotation.hpp:32-40 */
void orc_roty(double a, double *R)
{
    double c = cos(a), s = sin(a);
    R[0] = c;  R[3] = 0; R[6] = s;
    R[1] = 0;  R[4] = 1; R[7] = 0;
    R[2] = -s; R[5] = 0; R[8] = c;
}
/* src/rotation.hpp:42-50 */
void orc_rotz(double a, double *R)
{
    double c = cos(a), s = sin(a);
    R[0] = c; R[3] = -s; R[6] = 0;
    R[1] = s; R[4] = c;  R[7] = 0;
    R[2] = 0; R[5] = 0;  R[8] = 1;
}
/* src/rotation.hpp:52-63: Rnc = Rz(psi) Ry(theta) Rx(phi), theta = (phi,theta,psi) */
void orc_rpy2rot(const double *th, double *R)
{
    double Rx[9], Ry[9], Rz[9], T[9];
    orc_rotz(th[2], Rz);
    orc_roty(th[1], Ry);
    orc_rotx(th[0], Rx);
    mat3_mul(Rz, Ry, T);
    mat3_mul(T, Rx, R);
}
/* src/rotation.hpp:13-20 */
void orc_rot2rpy(const double *R, double *th)
{
    th[0] = atan2(R[2 + 3 * 1], R[2 + 3 * 2]);
    th[1] = atan2(-R[2 + 3 * 0], sqrt(pow(R[2 + 3 * 1], 2) + pow(R[2 + 3 * 2], 2)));
    th[2] = atan2(R[1 + 3 * 0], R[0 + 3 * 0]);
}

/* Distortion-coefficient selection, src/cameraModel.hpp:114-148 (the gating
 * is swapped relative to OpenCV's naming; reproduced as written). */
typedef struct { double fx, fy, cx, cy, k1, k2, k3, k4, k5, k6, p1, p2, s1, s2, s3, s4; } coeffs;
static void select_coeffs(const orc_camera *cam, coeffs *k)
{
    int rational = (cam->flag & ORC_RATIONAL) == ORC_RATIONAL;
    int thin = (cam->flag & ORC_THINPRISM) == ORC_THINPRISM;
    assert((cam->flag & ~(ORC_RATIONAL | ORC_THINPRISM)) == 0);
    k->fx = cam->fx; k->fy = cam->fy; k->cx = cam->cx; k->cy = cam->cy;
    k->k1 = cam->d[0]; k->k2 = cam->d[1]; k->p1 = cam->d[2]; k->p2 = cam->d[3]; k->k3 = cam->d[4];
    if (rational) {
        if (thin) { k->s1 = cam->d[8]; k->s2 = cam->d[9]; k->s3 = cam->d[10]; k->s4 = cam->d[11]; }
        else      { k->s1 = cam->d[5]; k->s2 = cam->d[6]; k->s3 = cam->d[7];  k->s4 = cam->d[8]; }
    } else { k->s1 = k->s2 = k->s3 = k->s4 = 0.0; }
    if (thin) { k->k4 = cam->d[5]; k->k5 = cam->d[6]; k->k6 = cam->d[7]; }
    else      { k->k4 = k->k5 = k->k6 = 0.0; }
}

/* src/cameraModel.hpp:32-216 */
int orc_world_to_pixel(const double *rPNn, const double *eta, const orc_camera *cam, double *px)
{
    double Rnc[9], d[3], q[3];
    orc_rpy2rot(eta + 3, Rnc);
    for (int i = 0; i < 3; ++i) d[i] = rPNn[i] - eta[i];
    mat3_tvec(Rnc, d, q);                                   /* :61 */
    double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    double uz = q[2] / nrm;                                 /* :63 */
    double cAngle = cos(cam->fov_deg * ((M_PI / 180.0) / 2.)); /* :67-68 */
    if (uz < cAngle) return 1;                              /* :70-75 */
    coeffs k; select_coeffs(cam, &k);
    if (q[2] <= 0) return 1;                                /* :179-181 */
    double u = q[0] / q[2], v = q[1] / q[2];
    double u2 = u * u, v2 = v * v, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2;
    double alpha = k.k1 * r2 + k.k2 * r4 + k.k3 * r6;
    double beta = k.k4 * r2 + k.k5 * r4 + k.k6 * r6;
    double c = (1.0 + alpha) / (1.0 + beta);
    double up = c * u + k.p1 * 2 * u * v + k.p2 * (r2 + 2 * u2) + k.s1 * r2 + k.s2 * r4;
    double vp = c * v + k.p2 * 2 * u * v + k.p1 * (r2 + 2 * v2) + k.s3 * r2 + k.s4 * r4;
    px[0] = k.fx * up + k.cx;                               /* :202 */
    px[1] = k.fy * vp + k.cy;
    int inW = 0 <= px[0] && px[0] <= cam->width - 1;        /* :204-213 */
    int inH = 0 <= px[1] && px[1] <= cam->height - 1;
    if (!(inW && inH)) return 2;
    return 0;
}

/* src/measurementPointLandmark.hpp:10-304 */
int orc_single_measure(int j, const double *x, int n, const orc_camera *cam, int layout,
                       double *h, double *SR, double *J)
{
    const double *eta = x + 6;
    const double *rP = x + 12 + 3 * j;
    int res = orc_world_to_pixel(rP, eta, cam, h);          /* :26-28 */
    if (SR) { SR[0] = 0.01; SR[1] = 0.0; SR[2] = 0.0; SR[3] = 0.01; } /* :35-41 */
    if (!J) return res;
    memset(J, 0, sizeof(double) * 2 * (size_t)n);           /* :56-57 */

    double Rnc[9], d[3], q[3];
    orc_rpy2rot(eta + 3, Rnc);                              /* :72 */
    for (int i = 0; i < 3; ++i) d[i] = rP[i] - eta[i];
    mat3_tvec(Rnc, d, q);                                   /* :74 */
    coeffs k; select_coeffs(cam, &k);
    double X = q[0], Y = q[1], Z = q[2], u, v;
    if (Z <= 0) { u = 0; v = 0; } else { u = X / Z; v = Y / Z; }   /* :175-181 */
    double u2 = u * u, v2 = v * v, r2 = u2 + v2, r = sqrt(r2), r3 = r2 * r;
    double r4 = r2 * r2, r5 = r4 * r, r6 = r4 * r2;
    double alpha = k.k1 * r2 + k.k2 * r4 + k.k3 * r6;
    double beta = k.k4 * r2 + k.k5 * r4 + k.k6 * r6;
    double c = (1.0 + alpha) / (1.0 + beta);
    double dudr[3], dvdr[3];
    if (Z <= 0) { dudr[0] = dudr[1] = dudr[2] = 0; dvdr[0] = dvdr[1] = dvdr[2] = 0; } /* :208-214 */
    else {
        dudr[0] = 1 / Z; dudr[1] = 0; dudr[2] = -X / (Z * Z);
        dvdr[0] = 0; dvdr[1] = 1 / Z; dvdr[2] = -Y / (Z * Z);
    }
    double drdu = (u != 0) ? pow(u2 + v2, -0.5) * u : 0;    /* :216-227 */
    double drdv = (v != 0) ? pow(u2 + v2, -0.5) * v : 0;
    double dalphadr = 2 * k.k1 * r + 4 * k.k2 * r3 + 6 * k.k3 * r5;  /* :229-232 */
    double dbetadr = 2 * k.k4 * r + 4 * k.k5 * r3 + 6 * k.k6 * r5;
    double dcdr = (dalphadr * (1 + beta) - (1 + alpha) * dbetadr) / ((1 + beta) * (1 + beta));
    double duddu = dcdr * drdu * u + c + 2 * k.p1 * v + k.p2 * (2 * r * drdu + 4 * u) + 2 * k.s1 * r * drdu + 4 * k.s2 * r3 * drdu;
    double duddv = dcdr * drdv * u + 2 * k.p1 * u + k.p2 * (2 * r * drdv) + 2 * k.s1 * r * drdv + 4 * k.s2 * r3 * drdv;
    double dvddu = dcdr * drdu * v + 2 * k.p2 * v + k.p1 * (2 * r * drdu) + 2 * k.s3 * r * drdu + 4 * k.s4 * r3 * drdu;
    double dvddv = dcdr * drdv * v + c + 2 * k.p2 * u + k.p1 * (2 * r * drdv + 4 * v) + 2 * k.s3 * r * drdv + 4 * k.s4 * r3 * drdv;
    double JrPCc[6]; /* 2x3 col-major */                    /* :249-254 */
    for (int i = 0; i < 3; ++i) {
        JrPCc[0 + 2 * i] = k.fx * (duddu * dudr[i] + duddv * dvdr[i]);
        JrPCc[1 + 2 * i] = k.fy * (dvddu * dudr[i] + dvddv * dvdr[i]);
    }
    double JrPNn[6];                                        /* :256: JrPCc * Rnc^T */
    for (int r_ = 0; r_ < 2; ++r_)
        for (int c_ = 0; c_ < 3; ++c_) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += JrPCc[r_ + 2 * t] * Rnc[c_ + 3 * t];
            JrPNn[r_ + 2 * c_] = s;
        }
    /* rotation block, :260-292.  COMPAT builds RX,RY,RZ from x(3),x(4),x(5). */
    const double *ang = (layout == 0) ? (x + 3) : (x + 9);
    double RX[9], RY[9], RZ[9], T[9], dPhi[9], dTheta[9], dPsi[9];
    static const double S1[9] = {0, 0, 0, 0, 0, 1, 0, -1, 0};   /* rows: 0 0 0 / 0 0 -1 / 0 1 0 */
    static const double S2[9] = {0, 0, -1, 0, 0, 0, 1, 0, 0};   /* rows: 0 0 1 / 0 0 0 / -1 0 0 */
    static const double S3[9] = {0, 1, 0, -1, 0, 0, 0, 0, 0};   /* rows: 0 -1 0 / 1 0 0 / 0 0 0 */
    orc_rotx(ang[0], RX); orc_roty(ang[1], RY); orc_rotz(ang[2], RZ);
    mat3_mul(RZ, S3, T); mat3_mul(T, RY, T); mat3_mul(T, RX, dPsi);     /* :286 */
    mat3_mul(RZ, RY, T); mat3_mul(T, S2, T); mat3_mul(T, RX, dTheta);   /* :287 */
    mat3_mul(RZ, RY, T); mat3_mul(T, RX, T); mat3_mul(T, S1, dPhi);     /* :288 */
    double JRnc[6], t3[3];
    const double *dR[3] = {dPhi, dTheta, dPsi};
    for (int a = 0; a < 3; ++a) {                           /* :290-292 */
        mat3_tvec(dR[a], d, t3);
        for (int r_ = 0; r_ < 2; ++r_)
            JRnc[r_ + 2 * a] = JrPCc[r_ + 0] * t3[0] + JrPCc[r_ + 2] * t3[1] + JrPCc[r_ + 4] * t3[2];
    }
    int c0 = (layout == 0) ? 0 : 6, c1 = (layout == 0) ? 3 : 9;
    int c2 = (layout == 0) ? 6 + 3 * j : 12 + 3 * j;        /* :294-296 */
    for (int a = 0; a < 3; ++a)
        for (int r_ = 0; r_ < 2; ++r_) {
            J[r_ + 2 * (c0 + a)] = -JrPNn[r_ + 2 * a];
            J[r_ + 2 * (c1 + a)] = JRnc[r_ + 2 * a];
            J[r_ + 2 * (c2 + a)] = JrPNn[r_ + 2 * a];
        }
    return res;
}

/* src/measurementPointLandmark.cpp:13-70 */
void orc_bundle_measure(const double *x, int n, const orc_camera *cam, int layout,
                        double *h, double *SR, double *C, int32_t *code)
{
    int L = (n - 12) / 3, m = 2 * L;
    double *Jt = C ? (double *)malloc(sizeof(double) * 2 * (size_t)n) : NULL;
    if (C) memset(C, 0, sizeof(double) * (size_t)m * n);
    for (int j = 0; j < L; ++j) {
        double hj[2] = {NAN, NAN};
        int res = orc_single_measure(j, x, n, cam, layout, hj, NULL, Jt);
        h[2 * j] = hj[0]; h[2 * j + 1] = hj[1];
        if (code) code[j] = res;
        if (C)
            for (int c = 0; c < n; ++c) {                   /* :64 */
                C[(2 * j) + (size_t)m * c] = Jt[0 + 2 * c];
                C[(2 * j + 1) + (size_t)m * c] = Jt[1 + 2 * c];
            }
    }
    if (SR) {                                               /* :67-69 */
        memset(SR, 0, sizeof(double) * (size_t)m * m);
        for (int i = 0; i < m; ++i) SR[i + (size_t)m * i] = 0.01;
    }
    free(Jt);
}

/* src/model.cpp:369-406 with logGaussian (src/gaussian.hpp:30-46) on S = sigma*I2 */
double orc_point_loglik_value(const double *x, int n, const double *y, const uint8_t *visible,
                              const orc_camera *cam, double sigma, int32_t *code)
{
    int L = (n - 12) / 3;
    double cost = 0.0;
    for (int j = 0; j < L; ++j) {
        double px[2];
        int flag = orc_world_to_pixel(x + 12 + 3 * j, x + 6, cam, px);
        if (code) code[j] = flag;
        if (!visible[j] || flag != 0) continue;
        double z0 = (y[2 * j] - px[0]) / sigma, z1 = (y[2 * j + 1] - px[1]) / sigma;
        double log_sum = log(fabs(sigma)) + log(fabs(sigma));
        cost += -0.5 * (z0 * z0 + z1 * z1) - 2.0 / 2 * log(2 * M_PI) - log_sum;
    }
    return cost;
}

/* ------------------------------------------------------------------ */
/* second-order dual numbers: f + a e1 + b e2 + ab e1 e2               */
/* (what autodiff::dual2nd carries: 4 doubles)                          */
/* ------------------------------------------------------------------ */
typedef struct { double v, a, b, ab; } d2;
static inline d2 d2k(double c) { d2 r = {c, 0, 0, 0}; return r; }
static inline d2 d2add(d2 x, d2 y) { d2 r = {x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; return r; }
static inline d2 d2sub(d2 x, d2 y) { d2 r = {x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; return r; }
static inline d2 d2neg(d2 x) { d2 r = {-x.v, -x.a, -x.b, -x.ab}; return r; }
static inline d2 d2mul(d2 x, d2 y)
{
    d2 r = {x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b,
            x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
    return r;
}
static inline d2 d2scale(double s, d2 x) { d2 r = {s * x.v, s * x.a, s * x.b, s * x.ab}; return r; }
static inline d2 d2fun(d2 x, double f, double f1, double f2)
{
    d2 r = {f, f1 * x.a, f1 * x.b, f1 * x.ab + f2 * x.a * x.b};
    return r;
}
static inline d2 d2div(d2 x, d2 y)
{
    double iv = 1.0 / y.v;
    return d2mul(x, d2fun(y, iv, -iv * iv, 2 * iv * iv * iv));
}
static inline d2 d2sin(d2 x) { double s = sin(x.v), c = cos(x.v); return d2fun(x, s, c, -s); }
static inline d2 d2cos(d2 x) { double s = sin(x.v), c = cos(x.v); return d2fun(x, c, -s, -c); }
static inline d2 d2sqrt(d2 x) { double s = sqrt(x.v); return d2fun(x, s, 0.5 / s, -0.25 / (s * x.v)); }

static void d2mat3_mul(const d2 *A, const d2 *B, d2 *C)
{
    d2 T[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            d2 s = d2mul(A[r], B[3 * c]);
            s = d2add(s, d2mul(A[r + 3], B[1 + 3 * c]));
            s = d2add(s, d2mul(A[r + 6], B[2 + 3 * c]));
            T[r + 3 * c] = s;
        }
    memcpy(C, T, sizeof T);
}

/* worldToPixel<dual2nd>, src/cameraModel.hpp:32-216, including the dense
 * 3x3 products of rpy2rot and the unused r = sqrt(r2). */
static int d2_world_to_pixel(const d2 *rP, const d2 *eta, const orc_camera *cam, const coeffs *k, d2 *px)
{
    d2 Rx[9], Ry[9], Rz[9], R[9];
    d2 cph = d2cos(eta[3]), sph = d2sin(eta[3]);
    d2 cth = d2cos(eta[4]), sth = d2sin(eta[4]);
    d2 cps = d2cos(eta[5]), sps = d2sin(eta[5]);
    d2 one = d2k(1), zero = d2k(0);
    Rx[0] = one;  Rx[3] = zero; Rx[6] = zero;
    Rx[1] = zero; Rx[4] = cph;  Rx[7] = d2neg(sph);
    Rx[2] = zero; Rx[5] = sph;  Rx[8] = cph;
    Ry[0] = cth;  Ry[3] = zero; Ry[6] = sth;
    Ry[1] = zero; Ry[4] = one;  Ry[7] = zero;
    Ry[2] = d2neg(sth); Ry[5] = zero; Ry[8] = cth;
    Rz[0] = cps;  Rz[3] = d2neg(sps); Rz[6] = zero;
    Rz[1] = sps;  Rz[4] = cps;  Rz[7] = zero;
    Rz[2] = zero; Rz[5] = zero; Rz[8] = one;
    d2mat3_mul(Rz, Ry, R);
    d2mat3_mul(R, Rx, R);
    d2 d[3], q[3];
    for (int i = 0; i < 3; ++i) d[i] = d2sub(rP[i], eta[i]);
    for (int c = 0; c < 3; ++c) {
        d2 s = d2mul(R[3 * c], d[0]);
        s = d2add(s, d2mul(R[1 + 3 * c], d[1]));
        s = d2add(s, d2mul(R[2 + 3 * c], d[2]));
        q[c] = s;
    }
    d2 nrm = d2sqrt(d2add(d2add(d2mul(q[0], q[0]), d2mul(q[1], q[1])), d2mul(q[2], q[2])));
    d2 uq2 = d2div(q[2], nrm);
    double cAngle = cos(cam->fov_deg * ((M_PI / 180.0) / 2.));
    if (uq2.v < cAngle) return 1;
    if (q[2].v <= 0) return 1;
    d2 u = d2div(q[0], q[2]), v = d2div(q[1], q[2]);
    d2 u2 = d2mul(u, u), v2 = d2mul(v, v), r2 = d2add(u2, v2);
    volatile double r_unused = sqrt(r2.v); (void)r_unused;   /* :192, value only matters for cost */
    d2 r4 = d2mul(r2, r2), r6 = d2mul(r4, r2);
    d2 alpha = d2add(d2add(d2scale(k->k1, r2), d2scale(k->k2, r4)), d2scale(k->k3, r6));
    d2 beta = d2add(d2add(d2scale(k->k4, r2), d2scale(k->k5, r4)), d2scale(k->k6, r6));
    d2 c = d2div(d2add(one, alpha), d2add(one, beta));
    d2 uv = d2mul(u, v);
    d2 up = d2add(d2add(d2add(d2add(d2mul(c, u), d2scale(k->p1 * 2, uv)),
                              d2scale(k->p2, d2add(r2, d2scale(2, u2)))),
                        d2scale(k->s1, r2)), d2scale(k->s2, r4));
    d2 vp = d2add(d2add(d2add(d2add(d2mul(c, v), d2scale(k->p2 * 2, uv)),
                              d2scale(k->p1, d2add(r2, d2scale(2, v2)))),
                        d2scale(k->s3, r2)), d2scale(k->s4, r4));
    px[0] = d2add(d2scale(k->fx, up), d2k(k->cx));
    px[1] = d2add(d2scale(k->fy, vp), d2k(k->cy));
    int inW = 0 <= px[0].v && px[0].v <= cam->width - 1;
    int inH = 0 <= px[1].v && px[1].v <= cam->height - 1;
    if (!(inW && inH)) return 2;
    return 0;
}

/* one landmark term of src/model.cpp:392-404 on dual numbers */
static int d2_landmark_term(const d2 *eta, const d2 *rP, const double *yj, const orc_camera *cam,
                            const coeffs *k, double sigma, d2 *term)
{
    d2 px[2];
    int flag = d2_world_to_pixel(rP, eta, cam, k, px);
    if (flag != 0) return flag;
    d2 z0 = d2scale(1.0 / sigma, d2sub(d2k(yj[0]), px[0]));
    d2 z1 = d2scale(1.0 / sigma, d2sub(d2k(yj[1]), px[1]));
    d2 sq = d2add(d2mul(z0, z0), d2mul(z1, z1));
    double cst = -2.0 / 2 * log(2 * M_PI) - (log(fabs(sigma)) + log(fabs(sigma)));
    *term = d2add(d2scale(-0.5, sq), d2k(cst));
    return 0;
}

/* autodiff::hessian pass structure on src/model.cpp:427-437: one full
 * likelihood evaluation per seed pair (i, j>=i). */
long orc_point_loglik_autodiff(const double *x, int n, const double *y, const uint8_t *visible,
                               const orc_camera *cam, double sigma, int row_begin, int row_end,
                               double *value, double *g, double *H)
{
    int L = (n - 12) / 3;
    coeffs k; select_coeffs(cam, &k);
    d2 *xd = (d2 *)malloc(sizeof(d2) * (size_t)n);
    long passes = 0;
    for (int i = 0; i < n; ++i) xd[i] = d2k(x[i]);
    for (int i = row_begin; i < row_end; ++i) {
        for (int j = i; j < n; ++j) {
            xd[i].a = 1.0;
            xd[j].b = 1.0;
            d2 cost = d2k(0);
            for (int l = 0; l < L; ++l) {
                if (!visible[l]) continue;
                d2 term;
                if (d2_landmark_term(xd + 6, xd + 12 + 3 * l, y + 2 * l, cam, &k, sigma, &term) == 0)
                    cost = d2add(cost, term);
            }
            xd[i].a = 0.0;
            xd[j].b = 0.0;
            ++passes;
            if (H) { H[i + (size_t)n * j] = cost.ab; H[j + (size_t)n * i] = cost.ab; }
            if (g) g[i] = cost.a;
            if (value) *value = cost.v;
        }
    }
    free(xd);
    return passes;
}

void orc_point_loglik_dual_sparse(const double *x, int n, const double *y, const uint8_t *visible,
                                  const orc_camera *cam, double sigma,
                                  double *value, double *g, double *H, int32_t *code)
{
    int L = (n - 12) / 3;
    coeffs k; select_coeffs(cam, &k);
    if (g) memset(g, 0, sizeof(double) * (size_t)n);
    if (H) memset(H, 0, sizeof(double) * (size_t)n * n);
    double val = 0.0;
    for (int l = 0; l < L; ++l) {
        int idx[9];
        for (int a = 0; a < 6; ++a) idx[a] = 6 + a;
        for (int a = 0; a < 3; ++a) idx[6 + a] = 12 + 3 * l + a;
        d2 z[9];
        for (int a = 0; a < 9; ++a) z[a] = d2k(x[idx[a]]);
        d2 term = d2k(0.0);
        int flag = d2_landmark_term(z, z + 6, y + 2 * l, cam, &k, sigma, &term);
        if (code) code[l] = flag;
        if (!visible[l] || flag != 0) continue;
        val += term.v;
        for (int a = 0; a < 9; ++a)
            for (int b = a; b < 9; ++b) {
                z[a].a = 1.0; z[b].b = 1.0;
                d2_landmark_term(z, z + 6, y + 2 * l, cam, &k, sigma, &term);
                z[a].a = 0.0; z[b].b = 0.0;
                if (H) {
                    H[idx[a] + (size_t)n * idx[b]] += term.ab;
                    if (a != b) H[idx[b] + (size_t)n * idx[a]] += term.ab;
                }
                if (g && b == a) g[idx[a]] += term.a;
            }
    }
    if (value) *value = val;
}

/* ------------------------------------------------------------------ */
/* closed-form gradient and exact Hessian                               */
/* ------------------------------------------------------------------ */
/* Distortion map D(u,v) of src/cameraModel.hpp:189-201 with first and second
 * partials w.r.t. (u,v), written through r2 = u^2+v^2 (smooth at the axis). */
static void distort_d2(const coeffs *k, double u, double v, double *D, double *JD /*2x2 row-major*/,
                       double *HDu /*uu,uv,vv*/, double *HDv)
{
    double u2 = u * u, v2 = v * v, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2;
    double N = 1 + k->k1 * r2 + k->k2 * r4 + k->k3 * r6;
    double N1 = k->k1 + 2 * k->k2 * r2 + 3 * k->k3 * r4, N2 = 2 * k->k2 + 6 * k->k3 * r2;
    double Dn = 1 + k->k4 * r2 + k->k5 * r4 + k->k6 * r6;
    double D1 = k->k4 + 2 * k->k5 * r2 + 3 * k->k6 * r4, D2 = 2 * k->k5 + 6 * k->k6 * r2;
    double c = N / Dn;
    double c1 = (N1 - c * D1) / Dn;                 /* dc/dr2 */
    double c2 = (N2 - 2 * c1 * D1 - c * D2) / Dn;   /* d2c/dr2^2 */
    D[0] = c * u + 2 * k->p1 * u * v + k->p2 * (r2 + 2 * u2) + k->s1 * r2 + k->s2 * r4;
    D[1] = c * v + 2 * k->p2 * u * v + k->p1 * (r2 + 2 * v2) + k->s3 * r2 + k->s4 * r4;
    JD[0] = c + 2 * u2 * c1 + 2 * k->p1 * v + 6 * k->p2 * u + 2 * k->s1 * u + 4 * k->s2 * r2 * u;
    JD[1] = 2 * u * v * c1 + 2 * k->p1 * u + 2 * k->p2 * v + 2 * k->s1 * v + 4 * k->s2 * r2 * v;
    JD[2] = 2 * u * v * c1 + 2 * k->p2 * v + 2 * k->p1 * u + 2 * k->s3 * u + 4 * k->s4 * r2 * u;
    JD[3] = c + 2 * v2 * c1 + 2 * k->p2 * u + 6 * k->p1 * v + 2 * k->s3 * v + 4 * k->s4 * r2 * v;
    HDu[0] = 6 * u * c1 + 4 * u2 * u * c2 + 6 * k->p2 + 2 * k->s1 + 4 * k->s2 * r2 + 8 * k->s2 * u2;
    HDu[1] = 2 * v * c1 + 4 * u2 * v * c2 + 2 * k->p1 + 8 * k->s2 * u * v;
    HDu[2] = 2 * u * c1 + 4 * u * v2 * c2 + 2 * k->p2 + 2 * k->s1 + 4 * k->s2 * r2 + 8 * k->s2 * v2;
    HDv[0] = 2 * v * c1 + 4 * v * u2 * c2 + 2 * k->p1 + 2 * k->s3 + 4 * k->s4 * r2 + 8 * k->s4 * u2;
    HDv[1] = 2 * u * c1 + 4 * v2 * u * c2 + 2 * k->p2 + 8 * k->s4 * u * v;
    HDv[2] = 6 * v * c1 + 4 * v2 * v * c2 + 6 * k->p1 + 2 * k->s3 + 4 * k->s4 * r2 + 8 * k->s4 * v2;
}

/* Gradient (9) and Hessian (9x9, col-major) of one landmark term w.r.t.
 * z = (rCNn, Theta, r_j), by the dense chain rule
 *   h = K D(pi(q)),  q = R(Theta)^T (r - rCNn),
 *   grad l = J^T e / s^2,  hess l = (-J^T J + sum_k e_k hess h_k) / s^2. */
static int analytic_landmark(const double *eta, const double *rP, const double *yj,
                             const orc_camera *cam, const coeffs *k, double sigma,
                             double *val, double *g9, double *H81)
{
    double px[2];
    int flag = orc_world_to_pixel(rP, eta, cam, px);
    if (flag != 0) return flag;
    double Rx[9], Ry[9], Rz[9], dRx[9], dRy[9], dRz[9], ddRx[9], ddRy[9], ddRz[9];
    double sp = sin(eta[3]), cp = cos(eta[3]), st = sin(eta[4]), ct = cos(eta[4]);
    double ss = sin(eta[5]), cs = cos(eta[5]);
    orc_rotx(eta[3], Rx); orc_roty(eta[4], Ry); orc_rotz(eta[5], Rz);
    memset(dRx, 0, sizeof dRx); memset(dRy, 0, sizeof dRy); memset(dRz, 0, sizeof dRz);
    dRx[4] = -sp; dRx[7] = -cp; dRx[5] = cp; dRx[8] = -sp;
    dRy[0] = -st; dRy[6] = ct; dRy[2] = -ct; dRy[8] = -st;
    dRz[0] = -ss; dRz[3] = -cs; dRz[1] = cs; dRz[4] = -ss;
    memset(ddRx, 0, sizeof ddRx); memset(ddRy, 0, sizeof ddRy); memset(ddRz, 0, sizeof ddRz);
    ddRx[4] = -cp; ddRx[7] = sp; ddRx[5] = -sp; ddRx[8] = -cp;
    ddRy[0] = -ct; ddRy[6] = -st; ddRy[2] = st; ddRy[8] = -ct;
    ddRz[0] = -cs; ddRz[3] = ss; ddRz[1] = -ss; ddRz[4] = -cs;
    const double *Xs[3] = {Rx, dRx, ddRx}, *Ys[3] = {Ry, dRy, ddRy}, *Zs[3] = {Rz, dRz, ddRz};
    /* Rd[a][b][c] = Z^(c) Y^(b) X^(a) : derivative orders (phi, theta, psi) */
    double R[9], dR[3][9], ddR[3][3][9], T[9];
    mat3_mul(Zs[0], Ys[0], T); mat3_mul(T, Xs[0], R);
    for (int a = 0; a < 3; ++a) {
        int o[3] = {0, 0, 0}; o[a] = 1;
        mat3_mul(Zs[o[2]], Ys[o[1]], T); mat3_mul(T, Xs[o[0]], dR[a]);
        for (int b = 0; b < 3; ++b) {
            int o2[3] = {0, 0, 0}; o2[a] += 1; o2[b] += 1;
            mat3_mul(Zs[o2[2]], Ys[o2[1]], T); mat3_mul(T, Xs[o2[0]], ddR[a][b]);
        }
    }
    double d[3], q[3];
    for (int i = 0; i < 3; ++i) d[i] = rP[i] - eta[i];
    mat3_tvec(R, d, q);
    /* Jq: 3x9, columns (c, Theta, r) */
    double Jq[3][9];
    for (int m = 0; m < 3; ++m)
        for (int i = 0; i < 3; ++i) { Jq[m][i] = -R[i + 3 * m]; Jq[m][6 + i] = R[i + 3 * m]; }
    for (int a = 0; a < 3; ++a) {
        double t[3]; mat3_tvec(dR[a], d, t);
        for (int m = 0; m < 3; ++m) Jq[m][3 + a] = t[m];
    }
    /* Hq[m]: 9x9 second derivatives of q_m */
    double Hq[3][9][9];
    memset(Hq, 0, sizeof Hq);
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) {
            double t[3]; mat3_tvec(ddR[a][b], d, t);
            for (int m = 0; m < 3; ++m) Hq[m][3 + a][3 + b] = t[m];
        }
        for (int i = 0; i < 3; ++i)
            for (int m = 0; m < 3; ++m) {
                double e = dR[a][i + 3 * m];           /* d q_m / d r_i d Theta_a */
                Hq[m][3 + a][6 + i] = e;  Hq[m][6 + i][3 + a] = e;
                Hq[m][3 + a][i] = -e;     Hq[m][i][3 + a] = -e;
            }
    }
    /* (u,v) of q */
    double z = q[2], u = q[0] / z, v = q[1] / z;
    double Ja[2][3] = {{1 / z, 0, -u / z}, {0, 1 / z, -v / z}};
    double Ha[2][3][3];
    memset(Ha, 0, sizeof Ha);
    Ha[0][0][2] = Ha[0][2][0] = -1 / (z * z); Ha[0][2][2] = 2 * u / (z * z);
    Ha[1][1][2] = Ha[1][2][1] = -1 / (z * z); Ha[1][2][2] = 2 * v / (z * z);
    double Dv[2], JD[4], HDu[3], HDv[3];
    distort_d2(k, u, v, Dv, JD, HDu, HDv);
    double f[2] = {k->fx, k->fy};
    double Jp[2][2] = {{f[0] * JD[0], f[0] * JD[1]}, {f[1] * JD[2], f[1] * JD[3]}};
    double Hp[2][2][2] = {{{f[0] * HDu[0], f[0] * HDu[1]}, {f[0] * HDu[1], f[0] * HDu[2]}},
                          {{f[1] * HDv[0], f[1] * HDv[1]}, {f[1] * HDv[1], f[1] * HDv[2]}}};
    /* g_k(q): Jg 2x3, Hg[k] 3x3 */
    double Jg[2][3], Hg[2][3][3];
    for (int kk = 0; kk < 2; ++kk)
        for (int i = 0; i < 3; ++i) {
            Jg[kk][i] = Jp[kk][0] * Ja[0][i] + Jp[kk][1] * Ja[1][i];
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) s += Ja[a][i] * Hp[kk][a][b] * Ja[b][j];
                s += Jp[kk][0] * Ha[0][i][j] + Jp[kk][1] * Ha[1][i][j];
                Hg[kk][i][j] = s;
            }
        }
    /* h_k(z): J 2x9, Hh[k] 9x9 */
    double J[2][9];
    double Hh[2][9][9];
    for (int kk = 0; kk < 2; ++kk)
        for (int i = 0; i < 9; ++i) {
            J[kk][i] = Jg[kk][0] * Jq[0][i] + Jg[kk][1] * Jq[1][i] + Jg[kk][2] * Jq[2][i];
            for (int j = 0; j < 9; ++j) {
                double s = 0;
                for (int a = 0; a < 3; ++a) {
                    for (int b = 0; b < 3; ++b) s += Jq[a][i] * Hg[kk][a][b] * Jq[b][j];
                    s += Jg[kk][a] * Hq[a][i][j];
                }
                Hh[kk][i][j] = s;
            }
        }
    double e[2] = {yj[0] - px[0], yj[1] - px[1]}, is2 = 1.0 / (sigma * sigma);
    *val = -0.5 * (e[0] * e[0] + e[1] * e[1]) * is2 - log(2 * M_PI) - 2 * log(fabs(sigma));
    for (int i = 0; i < 9; ++i) {
        g9[i] = (J[0][i] * e[0] + J[1][i] * e[1]) * is2;
        for (int j = 0; j < 9; ++j)
            H81[i + 9 * j] = (-(J[0][i] * J[0][j] + J[1][i] * J[1][j]) + e[0] * Hh[0][i][j] + e[1] * Hh[1][i][j]) * is2;
    }
    return 0;
}

void orc_point_loglik_analytic(const double *x, int n, const double *y, const uint8_t *visible,
                               const orc_camera *cam, double sigma,
                               double *value, double *g, double *H, int32_t *code)
{
    int L = (n - 12) / 3;
    coeffs k; select_coeffs(cam, &k);
    if (g) memset(g, 0, sizeof(double) * (size_t)n);
    if (H) memset(H, 0, sizeof(double) * (size_t)n * n);
    double val = 0.0;
    for (int l = 0; l < L; ++l) {
        double v = 0.0, g9[9], H81[81];
        int idx[9];
        for (int a = 0; a < 6; ++a) idx[a] = 6 + a;
        for (int a = 0; a < 3; ++a) idx[6 + a] = 12 + 3 * l + a;
        int flag;
        if (!visible[l]) { double px[2]; flag = orc_world_to_pixel(x + 12 + 3 * l, x + 6, cam, px); if (code) code[l] = flag; continue; }
        flag = analytic_landmark(x + 6, x + 12 + 3 * l, y + 2 * l, cam, &k, sigma, &v, g9, H81);
        if (code) code[l] = flag;
        if (flag != 0) continue;
        val += v;
        for (int a = 0; a < 9; ++a) {
            if (g) g[idx[a]] += g9[a];
            if (H) for (int b = 0; b < 9; ++b) H[idx[a] + (size_t)n * idx[b]] += H81[a + 9 * b];
        }
    }
    if (value) *value = val;
}

/* ------------------------------------------------------------------ */
/* Gaussian algebra                                                     */
/* ------------------------------------------------------------------ */
/* solve S^T z = b (S upper) in place: forward substitution */
static void trsv_ut(const double *S, int n, int ld, double *b)
{
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= S[k + (size_t)ld * i] * b[k];
        b[i] = s / S[i + (size_t)ld * i];
    }
}
/* solve S z = b (S upper) in place: back substitution */
static void trsv_u(const double *S, int n, int ld, double *b)
{
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= S[i + (size_t)ld * k] * b[k];
        b[i] = s / S[i + (size_t)ld * i];
    }
}

/* src/gaussian.hpp:30-74 */
double orc_log_gaussian(const double *x, const double *mu, const double *S, int n, int ldS,
                        double *g, double *H)
{
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
    double log_sum = 0.0, sq = 0.0;
    for (int i = 0; i < n; ++i) { log_sum += log(fabs(S[i + (size_t)ldS * i])); z[i] = x[i] - mu[i]; }
    trsv_ut(S, n, ldS, z);                                  /* :43 */
    for (int i = 0; i < n; ++i) sq += z[i] * z[i];
    double val = -0.5 * sq - (double)n / 2 * log(2 * M_PI) - log_sum;  /* :45 */
    if (g) {                                                /* :55-56 */
        for (int i = 0; i < n; ++i) g[i] = z[i];
        trsv_u(S, n, ldS, g);
        for (int i = 0; i < n; ++i) g[i] = -g[i];
    }
    if (H) {                                                /* :69-72 */
        for (int c = 0; c < n; ++c) {
            double *col = H + (size_t)n * c;
            for (int i = 0; i < n; ++i) col[i] = (i == c) ? 1.0 : 0.0;
            trsv_ut(S, n, ldS, col);
            trsv_u(S, n, ldS, col);
            for (int i = 0; i < n; ++i) col[i] = -col[i];
        }
    }
    free(z);
    return val;
}

/* Unblocked Householder QR, reflector convention of Eigen's makeHouseholder /
 * LAPACK dlarfg: beta = -sign(alpha) * norm (alpha >= 0 -> negative beta);
 * a column whose tail is exactly zero is left untouched (tau = 0). */
void orc_qr_upper(double *A, int rows, int cols, int lda)
{
    int steps = rows < cols ? rows : cols;
    double *v = (double *)malloc(sizeof(double) * (size_t)rows);
    for (int k = 0; k < steps; ++k) {
        double *col = A + (size_t)lda * k;
        double tail = 0.0;
        for (int i = k + 1; i < rows; ++i) tail += col[i] * col[i];
        double c0 = col[k], beta, tau;
        if (tail <= 2.2250738585072014e-308) { beta = c0; tau = 0.0; }
        else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
            for (int i = k + 1; i < rows; ++i) v[i] = col[i] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        if (tau != 0.0) {
            for (int j = k + 1; j < cols; ++j) {
                double *cj = A + (size_t)lda * j;
                double w = cj[k];
                for (int i = k + 1; i < rows; ++i) w += v[i] * cj[i];
                w *= tau;
                cj[k] -= w;
                for (int i = k + 1; i < rows; ++i) cj[i] -= v[i] * w;
            }
        }
        col[k] = beta;
        for (int i = k + 1; i < rows; ++i) col[i] = 0.0;
    }
    for (int j = 0; j < cols; ++j)
        for (int i = j + 1; i < rows; ++i) A[i + (size_t)lda * j] = 0.0;
    free(v);
}

/* src/gaussian.hpp:117-127 */
void orc_affine_sqrt(const double *S, int n, int ldS, const double *C, int m, int ldC,
                     const double *SR, int ldSR, double *Syy, int ldSyy)
{
    int rows = n + m;
    double *A = (double *)calloc((size_t)rows * m, sizeof(double));
    for (int j = 0; j < m; ++j) {
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += S[i + (size_t)ldS * k] * C[j + (size_t)ldC * k];
            A[i + (size_t)rows * j] = s;
        }
        for (int i = 0; i < m; ++i) A[n + i + (size_t)rows * j] = SR[i + (size_t)ldSR * j];
    }
    orc_qr_upper(A, rows, m, rows);
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) Syy[i + (size_t)ldSyy * j] = A[i + (size_t)rows * j];
    free(A);
}

/* ================= time update ================= */

/* SlamProcessModel, src/model.cpp:9-136.  f (n); SQ (n x n, or NULL): position_tune /
 * orientation_tune (3x3 column-major) on the diagonal blocks 0-2 and 3-5; dfdx (n x n, or NULL). */
void orc_process_model(const double *x, int n, const double *pos_tune, const double *ori_tune,
                       double *f, double *SQ, double *dfdx)
{
    double Rnc[9], T[9];                                     /* column-major 3x3 */
    const double phi = x[9], theta = x[10], psi = x[11];
    orc_rpy2rot(x + 9, Rnc);
    T[0] = 1; T[3] = sin(phi) * tan(theta); T[6] = cos(phi) * tan(theta);
    T[1] = 0; T[4] = cos(phi);              T[7] = -sin(phi);
    T[2] = 0; T[5] = sin(phi) / cos(theta); T[8] = cos(phi) / cos(theta);
    for (int i = 0; i < n; ++i) f[i] = 0.0;
    for (int i = 0; i < 3; ++i) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 3; ++k) { a += Rnc[i + 3 * k] * x[k]; b += T[i + 3 * k] * x[3 + k]; }
        f[6 + i] = a; f[9 + i] = b;
    }
    if (SQ) {
        for (size_t i = 0; i < (size_t)n * n; ++i) SQ[i] = 0.0;
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) {
                SQ[r + (size_t)n * c] = pos_tune[r + 3 * c];
                SQ[3 + r + (size_t)n * (3 + c)] = ori_tune[r + 3 * c];
            }
    }
    if (!dfdx) return;
    for (size_t i = 0; i < (size_t)n * n; ++i) dfdx[i] = 0.0;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            dfdx[6 + r + (size_t)n * c] = Rnc[r + 3 * c];
            dfdx[9 + r + (size_t)n * (3 + c)] = T[r + 3 * c];
        }
    const double x1dot = x[0], x2dot = x[1], x3dot = x[2], thetadot = x[4], psidot = x[5];
    double d[6][3];                                          /* dfdnu(:, 3..5), src/model.cpp:117-132 */
    d[0][0] = x2dot * (sin(phi) * sin(psi) + cos(phi) * cos(psi) * sin(theta)) + x3dot * (cos(phi) * sin(psi) - cos(psi) * sin(phi) * sin(theta));
    d[1][0] = -x2dot * (cos(psi) * sin(phi) - cos(phi) * sin(psi) * sin(theta)) - x3dot * (cos(phi) * cos(psi) + sin(phi) * sin(psi) * sin(theta));
    d[2][0] = x2dot * cos(phi) * cos(theta) - x3dot * cos(theta) * sin(phi);
    d[3][0] = thetadot * cos(phi) * tan(theta) - psidot * sin(phi) * tan(theta);
    d[4][0] = -psidot * cos(phi) - thetadot * sin(phi);
    d[5][0] = (thetadot * cos(phi)) / cos(theta) - (psidot * sin(phi)) / cos(theta);
    d[0][1] = x3dot * cos(phi) * cos(psi) * cos(theta) - x1dot * cos(psi) * sin(theta) + x2dot * cos(psi) * cos(theta) * sin(phi);
    d[1][1] = x3dot * cos(phi) * cos(theta) * sin(psi) - x1dot * sin(psi) * sin(theta) + x2dot * cos(theta) * sin(phi) * sin(psi);
    d[2][1] = -x1dot * cos(theta) - x3dot * cos(phi) * sin(theta) - x2dot * sin(phi) * sin(theta);
    d[3][1] = psidot * cos(phi) * (pow(tan(theta), 2) + 1) + thetadot * sin(phi) * (pow(tan(theta), 2) + 1);
    d[4][1] = 0;
    d[5][1] = (psidot * cos(phi) * sin(theta)) / pow(cos(theta), 2) + (thetadot * sin(phi) * sin(theta)) / pow(cos(theta), 2);
    d[0][2] = x3dot * (cos(psi) * sin(phi) - cos(phi) * sin(psi) * sin(theta)) - x2dot * (cos(phi) * cos(psi) + sin(phi) * sin(psi) * sin(theta)) - x1dot * cos(theta) * sin(psi);
    d[1][2] = x3dot * (sin(phi) * sin(psi) + cos(phi) * cos(psi) * sin(theta)) - x2dot * (cos(phi) * sin(psi) - cos(psi) * sin(phi) * sin(theta)) + x1dot * cos(psi) * cos(theta);
    d[2][2] = d[3][2] = d[4][2] = d[5][2] = 0;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 6; ++r) dfdx[6 + r + (size_t)n * (9 + c)] = d[r][c];
}

/* augmentGradients, src/gaussian.hpp:140-161: dX = [f(X(:,0)), Jx * X(:,1:)] for X n x (2n+1). */
static void augment_gradients(const double *X, int n, const double *pos_tune, const double *ori_tune,
                              double *dX, double *Jx)
{
    const int nc = 2 * n + 1;
    orc_process_model(X, n, pos_tune, ori_tune, dX, NULL, Jx);
    for (int c = 1; c < nc; ++c) {
        double *out = dX + (size_t)n * c;
        const double *xc = X + (size_t)n * c;
        for (int i = 0; i < n; ++i) out[i] = 0.0;
        for (int k = 0; k < n; ++k) {
            const double xk = xc[k];
            if (xk == 0.0) continue;                          /* a dense product; zero terms add nothing */
            const double *jc = Jx + (size_t)n * k;
            for (int i = 0; i < n; ++i) out[i] += jc[i] * xk;
        }
    }
}

/* RK4SDEHelper, src/gaussian.hpp:171-239.  xdw = [x; dw] (2n).  J (n x 2n column-major) may be NULL. */
void orc_rk4_sde(const double *xdw, int n, const double *pos_tune, const double *ori_tune, double dt,
                 double *xnext, double *J)
{
    const int nc = 2 * n + 1;
    const size_t sz = (size_t)n * nc;
    double *X = (double *)calloc(sz, sizeof(double)), *dW = (double *)calloc(sz, sizeof(double));
    double *Xs = (double *)malloc(sz * sizeof(double)), *Jx = (double *)malloc((size_t)n * n * sizeof(double));
    double *F[4];
    for (int s = 0; s < 4; ++s) F[s] = (double *)malloc(sz * sizeof(double));
    for (int i = 0; i < n; ++i) {
        X[i] = xdw[i]; dW[i] = xdw[n + i];
        X[i + (size_t)n * (1 + i)] = 1.0;
        dW[i + (size_t)n * (1 + n + i)] = 1.0;
    }
    augment_gradients(X, n, pos_tune, ori_tune, F[0], Jx);
    for (size_t i = 0; i < sz; ++i) Xs[i] = X[i] + (F[0][i] * dt + dW[i]) / 2;
    augment_gradients(Xs, n, pos_tune, ori_tune, F[1], Jx);
    for (size_t i = 0; i < sz; ++i) Xs[i] = X[i] + (F[1][i] * dt + dW[i]) / 2;
    augment_gradients(Xs, n, pos_tune, ori_tune, F[2], Jx);
    for (size_t i = 0; i < sz; ++i) Xs[i] = X[i] + F[2][i] * dt + dW[i];
    augment_gradients(Xs, n, pos_tune, ori_tune, F[3], Jx);
    for (size_t i = 0; i < sz; ++i) Xs[i] = X[i] + (F[0][i] + 2 * F[1][i] + 2 * F[2][i] + F[3][i]) * dt / 6 + dW[i];
    for (int i = 0; i < n; ++i) xnext[i] = Xs[i];
    if (J) for (size_t i = 0; i < (size_t)n * 2 * n; ++i) J[i] = Xs[n + i];
    for (int s = 0; s < 4; ++s) free(F[s]);
    free(X); free(dW); free(Xs); free(Jx);
}

/* timeUpdateContinuous, src/gaussian.hpp:337-383, with SlamProcessModel: the augmented state
 * [x; dw] ~ N([mu; 0], blkdiag(S, SQ sqrt(dt))) goes through affineTransform with the RK4 map. */
void orc_time_update(const double *mu, const double *S, int n, int ldS, const double *pos_tune,
                     const double *ori_tune, double dt, double *mu_out, double *S_out, int ldo)
{
    const int n2 = 2 * n, rows = 3 * n;
    double *xdw = (double *)calloc((size_t)n2, sizeof(double));
    double *f = (double *)malloc(sizeof(double) * (size_t)n), *SQ = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *Sxdw = (double *)calloc((size_t)n2 * n2, sizeof(double));
    double *J = (double *)malloc(sizeof(double) * (size_t)n * n2);
    double *A = (double *)calloc((size_t)rows * n, sizeof(double));           /* [Sxdw * J^T ; SR = 0] */
    for (int i = 0; i < n; ++i) xdw[i] = mu[i];
    orc_process_model(mu, n, pos_tune, ori_tune, f, SQ, NULL);
    const double sq = sqrt(dt);
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) {
            Sxdw[r + (size_t)n2 * c] = S[r + (size_t)ldS * c];
            Sxdw[n + r + (size_t)n2 * (n + c)] = SQ[r + (size_t)n * c] * sq;
        }
    orc_rk4_sde(xdw, n, pos_tune, ori_tune, dt, mu_out, J);
    for (int c = 0; c < n; ++c) {                                             /* column c of Sxdw * J^T */
        double *out = A + (size_t)rows * c;
        for (int k = 0; k < n2; ++k) {
            const double jk = J[c + (size_t)n * k];
            if (jk == 0.0) continue;
            const double *sc = Sxdw + (size_t)n2 * k;
            for (int i = 0; i < n2; ++i) out[i] += sc[i] * jk;
        }
    }
    orc_qr_upper(A, rows, n, rows);
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) S_out[r + (size_t)ldo * c] = A[r + (size_t)rows * c];
    free(xdw); free(f); free(SQ); free(Sxdw); free(J); free(A);
}

/* removeBadLandmarks, src/SLAM.cpp:796-820 (removeRow / removeColumn :848-868). */
void orc_remove_landmark(const double *mu, const double *S, int n, int ldS, int j, double *mu_out,
                         double *S_out, int ldo)
{
    const int c0 = 12 + 3 * j, m = n - 3;
    double *A = (double *)malloc(sizeof(double) * (size_t)n * m);
    for (int i = 0; i < m; ++i) mu_out[i] = mu[i < c0 ? i : i + 3];
    for (int c = 0; c < m; ++c)
        for (int r = 0; r < n; ++r) A[r + (size_t)n * c] = S[r + (size_t)ldS * (c < c0 ? c : c + 3)];
    orc_qr_upper(A, n, m, n);
    for (int c = 0; c < m; ++c)
        for (int r = 0; r < m; ++r) S_out[r + (size_t)ldo * c] = A[r + (size_t)n * c];
    free(A);
}

/* cv::BFMatcher(NORM_HAMMING).match(query, train), src/SLAM.cpp:222,355: nearest train descriptor per
 * query, first index among equal distances; -1 when there is no train descriptor.  OpenCV's matcher
 * is not in this environment: this restates its documented behaviour (parity unpinned). */
void orc_match_hamming(const uint8_t *query, int nq, const uint8_t *train, int nt, int desc_bytes,
                       int32_t *train_idx, int32_t *distance)
{
    for (int q = 0; q < nq; ++q) {
        int best = -1, bi = -1;
        for (int t = 0; t < nt; ++t) {
            int d = 0;
            for (int b = 0; b < desc_bytes; ++b) d += __builtin_popcount((unsigned)(query[(size_t)q * desc_bytes + b] ^ train[(size_t)t * desc_bytes + b]));
            if (bi < 0 || d < best) { best = d; bi = t; }
        }
        train_idx[q] = bi;
        if (distance) distance[q] = best;
    }
}

/* ================= data association ================= */

/* src/utility.cpp:20-24 */
double orc_normcdf(double z) { return 0.5 * erfc(-z / sqrt(2.0)); }

/* Regularised lower incomplete gamma P(a, x): series for x < a + 1, Lentz continued fraction for
 * Q = 1 - P otherwise. */
static double gamma_p(double a, double x)
{
    if (x <= 0.0) return 0.0;
    const double lg = lgamma(a);
    if (x < a + 1.0) {
        double ap = a, sum = 1.0 / a, del = sum;
        for (int it = 0; it < 10000; ++it) {
            ap += 1.0; del *= x / ap; sum += del;
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        return sum * exp(-x + a * log(x) - lg);
    }
    double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;
    for (int i = 1; i < 10000; ++i) {
        double an = -i * (i - a);
        b += 2.0;
        d = an * d + b; if (fabs(d) < 1e-300) d = 1e-300;
        c = b + an / c; if (fabs(c) < 1e-300) c = 1e-300;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-17) break;
    }
    return 1.0 - exp(-x + a * log(x) - lg) * h;
}

/* src/utility.cpp:13-18 */
double orc_chi2inv(double p, double nu)
{
    const double a = nu / 2.0;
    if (p <= 0.0) return 0.0;
    double lo = 0.0, hi = a > 1.0 ? a : 1.0;
    while (gamma_p(a, hi) < p) { lo = hi; hi *= 2.0; }
    double x = 0.5 * (lo + hi);
    for (int it = 0; it < 200; ++it) {
        const double f = gamma_p(a, x) - p;
        if (f > 0.0) hi = x; else lo = x;
        const double dens = exp(-x + (a - 1.0) * log(x) - lgamma(a));     /* dP/dx */
        double xn = x - f / dens;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (fabs(xn - x) <= 4e-16 * fabs(xn)) { x = xn; break; }
        x = xn;
    }
    return 2.0 * x;
}

/* src/dataAssociation.cpp:154-177 */
int orc_individual_compat(int i, int j, int ny, const double *y, const double *muY,
                          const double *SYY, int N, const double *chi2LUT)
{
    double *A = (double *)malloc(sizeof(double) * (size_t)N * ny);
    for (int c = 0; c < ny; ++c)
        for (int r = 0; r < N; ++r) A[r + (size_t)N * c] = SYY[r + (size_t)N * (j * ny + c)];   /* :165 */
    orc_qr_upper(A, N, ny, N);                                                                  /* :167-171 */
    double t[16];
    for (int r = 0; r < ny; ++r) t[r] = y[r + (size_t)ny * i] - muY[j * ny + r];
    trsv_ut(A, ny, N, t);                                                                       /* :172 */
    double z = 0.0;
    for (int r = 0; r < ny; ++r) z += t[r] * t[r];
    free(A);
    return z < chi2LUT[j];                                                                      /* :175 */
}

/* src/dataAssociation.cpp:179-285 */
int orc_joint_compat(const int *idx, int n, double sU, int ny, const double *y, const double *muY,
                     const double *SYY, const double *chi2LUT, double *surprisal)
{
    const int N = n * ny;
    int nA = 0;
    for (int j = 0; j < n; ++j) nA += idx[j] != -1;
    const int k = ny * nA;
    double *SS = (double *)malloc(sizeof(double) * (size_t)N * (k > 0 ? k : 1));
    double *t = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
    int c = 0;
    for (int j = 0; j < n; ++j) {
        if (idx[j] == -1) continue;
        for (int q = 0; q < ny; ++q, ++c) {
            for (int r = 0; r < N; ++r) SS[r + (size_t)N * c] = SYY[r + (size_t)N * (ny * j + q)];  /* :209-215 */
            t[c] = y[q + (size_t)ny * idx[j]] - muY[ny * j + q];                                    /* :223-230 */
        }
    }
    double dM2A = 0.0, s2 = 0.0;
    if (k > 0) {
        orc_qr_upper(SS, N, k, N);                                                                  /* :217-218 */
        trsv_ut(SS, k, N, t);                                                                       /* :230 */
        for (int r = 0; r < k; ++r) { dM2A += t[r] * t[r]; s2 += log(fabs(SS[r + (size_t)N * r])); }
    }
    const double halflog2pi = log(M_PI * 2) / 2;
    const double sA = ny * nA * halflog2pi + s2 + 0.5 * dM2A;                                       /* :233-237 */
    *surprisal = sA + (n - nA) * sU;                                                                /* :240-243 */
    free(SS); free(t);
    return nA > 0 ? dM2A <= chi2LUT[nA - 1] : 1;                                                    /* :276-284 */
}

/* std::set_difference as the reference calls it (src/dataAssociation.cpp:106) on a second range that is
 * not sorted: the merge walk is restated literally so that the same features are offered. */
static int set_difference_walk(const int *a, int na, const int *b, int nb, int *out)
{
    int i = 0, j = 0, k = 0;
    while (i < na && j < nb) {
        if (a[i] < b[j]) out[k++] = a[i++];
        else if (b[j] < a[i]) ++j;
        else { ++i; ++j; }
    }
    while (i < na) out[k++] = a[i++];
    return k;
}

/* src/dataAssociation.cpp:63-149 */
double orc_snn_density(const double *muY, const double *SYY, int n, const double *y, int m,
                       int width, int height, int enforce_jc, int *idx)
{
    const int ny = 2, N = ny * n;
    const double c = 2 * orc_normcdf(3.0) - 1;                                       /* :27-29 */
    double *lut = (double *)malloc(sizeof(double) * (size_t)n);
    for (int jj = 0; jj < n; ++jj) lut[jj] = orc_chi2inv(c, ny * (jj + 1));          /* :64-69 */
    int *midx = (int *)malloc(sizeof(int) * (size_t)m), *diff = (int *)malloc(sizeof(int) * (size_t)m);
    int *cur = (int *)malloc(sizeof(int) * (size_t)n), *next = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < m; ++i) midx[i] = i;
    for (int j = 0; j < n; ++j) idx[j] = -1;
    const double sU = log((double)width) + log((double)height);                      /* :84 */
    double s = n * sU;
    (void)N;
    for (int j = 0; j < n; ++j) {
        double dsmin = INFINITY, scur = s, snext = 0.0;
        memcpy(cur, idx, sizeof(int) * (size_t)n);
        const int nd = set_difference_walk(midx, m, cur, j, diff);                   /* :106 */
        for (int q = 0; q < nd; ++q) {
            const int i = diff[q];
            if (!orc_individual_compat(i, j, ny, y, muY, SYY, ny * n, lut)) continue;   /* :113 */
            memcpy(next, cur, sizeof(int) * (size_t)n);
            next[j] = i;
            const int jc = orc_joint_compat(next, n, sU, ny, y, muY, SYY, lut, &snext);  /* :120 */
            if (enforce_jc && !jc) continue;
            const double ds = snext - scur;
            if (ds < dsmin) { memcpy(idx, next, sizeof(int) * (size_t)n); dsmin = ds; s = snext; }
        }
        orc_joint_compat(cur, n, sU, ny, y, muY, SYY, lut, &snext);                  /* :134-135 */
        const double ds = snext - scur;
        if (ds < dsmin) { memcpy(idx, cur, sizeof(int) * (size_t)n); s = snext; }    /* :138-142 */
    }
    free(lut); free(midx); free(diff); free(cur); free(next);
    return s;
}

/* src/dataAssociation.cpp:25-150 */
double orc_snn(const double *mux, const double *Sxx, int nx_all, const double *y, int m,
               const orc_camera *cam, int layout, int enforce_jc, int *idx)
{
    const int n = (nx_all - 12) / 3, N = 2 * n;
    double *h = (double *)malloc(sizeof(double) * (size_t)N);
    double *SR = (double *)calloc((size_t)N * N, sizeof(double));
    double *C = (double *)calloc((size_t)N * nx_all, sizeof(double));
    double *SYY = (double *)calloc((size_t)N * N, sizeof(double));
    orc_bundle_measure(mux, nx_all, cam, layout, h, SR, C, NULL);
    orc_affine_sqrt(Sxx, nx_all, nx_all, C, N, N, SR, N, SYY, N);                    /* :61 */
    const double s = orc_snn_density(h, SYY, n, y, m, cam->width, cam->height, enforce_jc, idx);
    free(h); free(SR); free(C); free(SYY);
    return s;
}

/* src/gaussian.cpp:19-36 */
void orc_condition_on_marginal(const double *muyx, const double *Syx, int ny, int nx, int ld,
                               const double *y, double *mux_out, double *Sx_out, int ldo)
{
    double *t = (double *)malloc(sizeof(double) * (size_t)ny);
    for (int i = 0; i < ny; ++i) t[i] = y[i] - muyx[i];
    trsv_ut(Syx, ny, ld, t);                                /* S1^T \ (y - muy) */
    for (int i = 0; i < nx; ++i) {
        double s = muyx[ny + i];
        for (int k = 0; k < ny; ++k) s += Syx[k + (size_t)ld * (ny + i)] * t[k];  /* S2^T t */
        mux_out[i] = s;
    }
    for (int j = 0; j < nx; ++j)
        for (int i = 0; i < nx; ++i) Sx_out[i + (size_t)ldo * j] = Syx[(ny + i) + (size_t)ld * (ny + j)];
    free(t);
}

/* src/gaussian.hpp:513-516 */
void orc_iekf_post_sqrt(const double *Q, const double *v, int n, double *S_out)
{
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) S_out[i + (size_t)n * j] = (1.0 / sqrt(v[i])) * Q[j + (size_t)n * i];
    orc_qr_upper(S_out, n, n, n);
}

/* Cyclic Jacobi; ascending eigenvalues; uses the lower triangle of H. */
void orc_sym_eig(const double *Hin, int n, double *Q, double *v)
{
    double *A = (double *)malloc(sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i) {
            A[i + (size_t)n * j] = Hin[i + (size_t)n * j];
            A[j + (size_t)n * i] = Hin[i + (size_t)n * j];
        }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Q[i + (size_t)n * j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int j = 0; j < n; ++j) {
            diag += A[j + (size_t)n * j] * A[j + (size_t)n * j];
            for (int i = j + 1; i < n; ++i) off += A[i + (size_t)n * j] * A[i + (size_t)n * j];
        }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p + (size_t)n * q];
                if (apq == 0.0) continue;
                double app = A[p + (size_t)n * p], aqq = A[q + (size_t)n * q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int kk = 0; kk < n; ++kk) {
                    double akp = A[kk + (size_t)n * p], akq = A[kk + (size_t)n * q];
                    A[kk + (size_t)n * p] = c * akp - s * akq;
                    A[kk + (size_t)n * q] = s * akp + c * akq;
                }
                for (int kk = 0; kk < n; ++kk) {
                    double apk = A[p + (size_t)n * kk], aqk = A[q + (size_t)n * kk];
                    A[p + (size_t)n * kk] = c * apk - s * aqk;
                    A[q + (size_t)n * kk] = s * apk + c * aqk;
                }
                for (int kk = 0; kk < n; ++kk) {
                    double qkp = Q[kk + (size_t)n * p], qkq = Q[kk + (size_t)n * q];
                    Q[kk + (size_t)n * p] = c * qkp - s * qkq;
                    Q[kk + (size_t)n * q] = s * qkp + c * qkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) v[i] = A[i + (size_t)n * i];
    /* selection sort ascending */
    for (int i = 0; i < n - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j) if (v[j] < v[m]) m = j;
        if (m != i) {
            double t = v[i]; v[i] = v[m]; v[m] = t;
            for (int kk = 0; kk < n; ++kk) {
                double tq = Q[kk + (size_t)n * i]; Q[kk + (size_t)n * i] = Q[kk + (size_t)n * m]; Q[kk + (size_t)n * m] = tq;
            }
        }
    }
    free(A);
}

static double vnorm(const double *a, int n) { double s = 0; for (int i = 0; i < n; ++i) s += a[i] * a[i]; return sqrt(s); }

/* src/fmin.cpp:40-131 */
int orc_trs(const double *Q, const double *v, const double *g, int n, double D, double *p)
{
    const double eps = 2.220446049250313e-16, sqrteps = sqrt(eps);
    const int maxIterations = 20;
    double l1 = v[0];
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    double *vlam = (double *)malloc(sizeof(double) * (size_t)n);
    double *t = (double *)malloc(sizeof(double) * (size_t)n);
    int ret = 0;
    for (int j = 0; j < n; ++j) { double s = 0; for (int i = 0; i < n; ++i) s += Q[i + (size_t)n * j] * g[i]; a[j] = s; }
    double lam = (l1 < 0) ? 1.01 * fabs(l1) : 0;
    for (int i = 0; i < n; ++i) vlam[i] = v[i] + lam;
#define ORC_P_FROM(vl) do { for (int i_ = 0; i_ < n; ++i_) t[i_] = a[i_] / (vl)[i_]; \
        for (int r_ = 0; r_ < n; ++r_) { double s_ = 0; for (int c_ = 0; c_ < n; ++c_) s_ += Q[r_ + (size_t)n * c_] * t[c_]; p[r_] = -s_; } } while (0)
    ORC_P_FROM(vlam);
    int hard = fabs(a[0]) < eps && l1 < 0;
    double pn = vnorm(p, n);
    if (l1 < 0 || pn > D || fabs(lam * (pn - D)) > sqrteps) {
        if (hard) {
            double *pvec = (double *)calloc((size_t)n, sizeof(double));
            double ssq = 0;
            for (int i = 0; i < n; ++i)
                if (fabs(v[i] - l1) > sqrteps) {
                    double sc = a[i] / (v[i] - l1);
                    ssq += sc * sc;
                    for (int r = 0; r < n; ++r) pvec[r] += Q[r + (size_t)n * i] * sc;
                }
            double tt = sqrt(D * D - ssq);
            for (int r = 0; r < n; ++r) p[r] = tt * Q[r] - pvec[r];
            double pg = 0, quad = 0;
            for (int r = 0; r < n; ++r) pg += p[r] * g[r];
            for (int j = 0; j < n; ++j) { double s = 0; for (int i = 0; i < n; ++i) s += Q[i + (size_t)n * j] * p[i]; quad += s * v[j] * s; }
            if (pg + 0.5 * quad > 0.0)
                for (int r = 0; r < n; ++r) p[r] = -tt * Q[r] - pvec[r];
            free(pvec);
        } else {
            int k;
            for (k = 0; k < maxIterations; ++k) {
                double ppn2 = 0, dpp = 0;
                for (int i = 0; i < n; ++i) {
                    double pp = -a[i] / vlam[i], dp = a[i] / (vlam[i] * vlam[i]);
                    ppn2 += pp * pp; dpp += dp * pp;
                }
                double ppn = sqrt(ppn2);
                double ff = 1 / D - 1 / ppn;
                double gg = dpp / (ppn * ppn * ppn);
                double lo = fmax(0.0, -l1) + sqrteps * fmax(0.0, -l1);
                lam = fmax(lo, lam - ff / gg);
                for (int i = 0; i < n; ++i) vlam[i] = v[i] + lam;
                if (fabs(ff) < sqrteps) break;
            }
            ORC_P_FROM(vlam);
            if (k >= maxIterations) ret = 1;
        }
    }
#undef ORC_P_FROM
    free(a); free(vlam); free(t);
    return ret;
}

/* CostJointDensity, src/gaussian.hpp:455-466, with the prior Hessian cached
 * (it does not depend on x; the reference recomputes it, :459). */
typedef struct {
    int n; const double *mu, *S, *y; const uint8_t *vis; const orc_camera *cam; double sigma;
    double *Hprior; int evals;
} cost_ctx;
static double cost_eval(cost_ctx *c, const double *x, double *g, double *H)
{
    int n = c->n;
    double *gp = (double *)malloc(sizeof(double) * (size_t)n);
    double lp;
    if (!c->Hprior) {
        c->Hprior = (double *)malloc(sizeof(double) * (size_t)n * n);
        lp = orc_log_gaussian(x, c->mu, c->S, n, n, gp, c->Hprior);
    } else lp = orc_log_gaussian(x, c->mu, c->S, n, n, gp, NULL);
    double ll;
    orc_point_loglik_analytic(x, n, c->y, c->vis, c->cam, c->sigma, &ll, g, H, NULL);
    for (int i = 0; i < n; ++i) g[i] = -(gp[i] + g[i]);
    for (size_t i = 0; i < (size_t)n * n; ++i) H[i] = -(c->Hprior[i] + H[i]);
    free(gp);
    c->evals++;
    return -(lp + ll);
}

/* src/gaussian.hpp:469-532 driving src/fmin.hpp:19-107 */
int orc_measurement_update_iekf(const double *mu, const double *S, int n,
                                const double *y, const uint8_t *visible,
                                const orc_camera *cam, double sigma,
                                double *mu_post, double *S_post, int *iters, int *cost_evals)
{
    size_t nn = (size_t)n * n;
    cost_ctx c = {n, mu, S, y, visible, cam, sigma, NULL, 0};
    double *x = mu_post;
    memcpy(x, mu, sizeof(double) * (size_t)n);              /* gaussian.hpp:494 */
    double *g = (double *)malloc(sizeof(double) * n), *gn = (double *)malloc(sizeof(double) * n);
    double *xn = (double *)malloc(sizeof(double) * n), *p = (double *)malloc(sizeof(double) * n);
    double *H = (double *)malloc(sizeof(double) * nn), *Hn = (double *)malloc(sizeof(double) * nn);
    double *Q = (double *)malloc(sizeof(double) * nn), *v = (double *)malloc(sizeof(double) * n);
    double f = cost_eval(&c, x, g, H);                      /* fmin.hpp:39 */
    orc_sym_eig(H, n, Q, v);
    double Delta = 1e0;
    int ret = 1, i;
    for (i = 0; i < 50; ++i) {
        orc_trs(Q, v, g, n, Delta, p);
        double pg = 0; for (int k = 0; k < n; ++k) pg += p[k] * g[k];
        double LambdaSq = -pg;
        if (fabs(LambdaSq) < 2 * 2.220446049250313e-16) { ret = 0; break; }
        for (int k = 0; k < n; ++k) xn[k] = x[k] + p[k];
        double fn = cost_eval(&c, xn, gn, Hn);
        double pHp = 0;
        for (int r = 0; r < n; ++r) { double s = 0; for (int cc = 0; cc < n; ++cc) s += H[r + (size_t)n * cc] * p[cc]; pHp += p[r] * s; }
        double dm = -pg - 0.5 * pHp;
        double rho = (f - fn) / dm;
        double pn = vnorm(p, n);
        if (rho < 0.1) Delta = 0.25 * pn;
        else if (rho > 0.75 && pn > 0.8 * Delta) Delta = 2.0 * Delta;
        if (rho >= 0.001) {
            memcpy(x, xn, sizeof(double) * n); f = fn;
            memcpy(g, gn, sizeof(double) * n); memcpy(H, Hn, sizeof(double) * nn);
            orc_sym_eig(H, n, Q, v);
        }
    }
    orc_iekf_post_sqrt(Q, v, n, S_post);
    if (iters) *iters = i;
    if (cost_evals) *cost_evals = c.evals;
    free(g); free(gn); free(xn); free(p); free(H); free(Hn); free(Q); free(v); free(c.Hprior);
    return ret;
}

/* ------------------------------------------------------------------ */
/* image front end                                                      */
/* ------------------------------------------------------------------ */
/*
 * The reference calls cv::ORB with HARRIS_SCORE (src/imagefeatures.cpp:71-96);
 * the corner arithmetic itself lives in OpenCV, which is not available, and no
 * reference test covers it: keypoint parity is UNPINNED by the reference.
 * This oracle defines the contract the GPU must match bit for bit:
 *
 *   Ix, Iy   3x3 Sobel on the u8 image, exact int32
 *   a,b,c    sums of Ix^2, Iy^2, IxIy over a block_size^2 window, exact int32
 *   scale    s = 1/(4*block_size*255), s2 = s*s, s4 = s2*s2 (float32)
 *   Harris   ((fa*fb - fc*fc) - k*((fa+fb)*(fa+fb))) * s4      (float32, each
 *            operation rounded once, no fused multiply-add)
 *   Shi-T.   (((fa+fb)*0.5f) - sqrtf(d*d + fc*fc)) * s2, d = (fa-fb)*0.5f
 *   valid    pixels at least 1+block_size/2 from every edge; 0 elsewhere
 *   NMS      3x3: strictly greater than the 4 neighbours that precede in
 *            raster order, greater-or-equal than the 4 that follow
 *   keep     response > threshold, border <= x < w-border, same for y
 *   order    score descending, then y, then x ascending; first max_features
 */
static void structure_tensor(const uint8_t *g, int w, int h, int stride, int bs,
                             int32_t *A, int32_t *B, int32_t *C)
{
    size_t npx = (size_t)w * h;
    int16_t *Ix = (int16_t *)calloc(npx, sizeof(int16_t)), *Iy = (int16_t *)calloc(npx, sizeof(int16_t));
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const uint8_t *r0 = g + (size_t)stride * (y - 1), *r1 = g + (size_t)stride * y, *r2 = g + (size_t)stride * (y + 1);
            Ix[x + (size_t)w * y] = (int16_t)((r0[x + 1] - r0[x - 1]) + 2 * (r1[x + 1] - r1[x - 1]) + (r2[x + 1] - r2[x - 1]));
            Iy[x + (size_t)w * y] = (int16_t)((r2[x - 1] - r0[x - 1]) + 2 * (r2[x] - r0[x]) + (r2[x + 1] - r0[x + 1]));
        }
    int hb = bs / 2, R = 1 + hb;
    memset(A, 0, npx * sizeof(int32_t)); memset(B, 0, npx * sizeof(int32_t)); memset(C, 0, npx * sizeof(int32_t));
    for (int y = R; y < h - R; ++y)
        for (int x = R; x < w - R; ++x) {
            int32_t a = 0, b = 0, c = 0;
            for (int dy = -hb; dy <= hb; ++dy)
                for (int dx = -hb; dx <= hb; ++dx) {
                    int32_t ix = Ix[(x + dx) + (size_t)w * (y + dy)], iy = Iy[(x + dx) + (size_t)w * (y + dy)];
                    a += ix * ix; b += iy * iy; c += ix * iy;
                }
            A[x + (size_t)w * y] = a; B[x + (size_t)w * y] = b; C[x + (size_t)w * y] = c;
        }
    free(Ix); free(Iy);
}

void orc_corner_response(const uint8_t *gray, int w, int h, int stride,
                         const orc_corner_params *p, float *harris, float *shi)
{
    size_t npx = (size_t)w * h;
    int32_t *A = (int32_t *)malloc(npx * 4), *B = (int32_t *)malloc(npx * 4), *C = (int32_t *)malloc(npx * 4);
    structure_tensor(gray, w, h, stride, p->block_size, A, B, C);
    volatile float s = 1.0f / (4.0f * (float)p->block_size * 255.0f);
    volatile float s2 = s * s;
    volatile float s4 = s2 * s2;
    int R = 1 + p->block_size / 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t i = x + (size_t)w * y;
            if (x < R || y < R || x >= w - R || y >= h - R) {
                if (harris) harris[i] = 0.0f;
                if (shi) shi[i] = 0.0f;
                continue;
            }
            volatile float fa = (float)A[i], fb = (float)B[i], fc = (float)C[i];
            volatile float t1 = fa * fb, t2 = fc * fc, t3 = t1 - t2;
            volatile float t4 = fa + fb, t5 = t4 * t4, t6 = p->harris_k * t5, t7 = t3 - t6;
            if (harris) harris[i] = t7 * s4;
            volatile float m = t4 * 0.5f, d = (fa - fb) * 0.5f, dd = d * d, disc = dd + t2;
            volatile float root = sqrtf(disc), lam = m - root;
            if (shi) shi[i] = lam * s2;
        }
    free(A); free(B); free(C);
}

static int feat_cmp(const void *pa, const void *pb)
{
    const orc_feature *a = (const orc_feature *)pa, *b = (const orc_feature *)pb;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    return a->x < b->x ? -1 : (a->x > b->x);
}

int orc_corner_detect(const uint8_t *gray, int w, int h, int stride,
                      const orc_corner_params *p, int which, orc_feature *out, int cap)
{
    size_t npx = (size_t)w * h;
    float *resp = (float *)malloc(npx * sizeof(float));
    orc_corner_response(gray, w, h, stride, p, which == 0 ? resp : NULL, which == 0 ? NULL : resp);
    float thr = which == 0 ? p->harris_thresh : p->shi_thresh;
    int bd = p->border;
    size_t cnt = 0, capc = 1024;
    orc_feature *cand = (orc_feature *)malloc(capc * sizeof(orc_feature));
    for (int y = bd; y < h - bd; ++y)
        for (int x = bd; x < w - bd; ++x) {
            float r = resp[x + (size_t)w * y];
            if (!(r > thr)) continue;
            int ok = 1;
            for (int dy = -1; dy <= 1 && ok; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!dx && !dy) continue;
                    float nb = resp[(x + dx) + (size_t)w * (y + dy)];
                    int before = dy < 0 || (dy == 0 && dx < 0);
                    if (before ? !(r > nb) : !(r >= nb)) { ok = 0; break; }
                }
            if (!ok) continue;
            if (cnt == capc) { capc *= 2; cand = (orc_feature *)realloc(cand, capc * sizeof(orc_feature)); }
            cand[cnt].x = x; cand[cnt].y = y; cand[cnt].score = r; ++cnt;
        }
    qsort(cand, cnt, sizeof(orc_feature), feat_cmp);
    int lim = p->max_features < cap ? p->max_features : cap;
    int nout = (int)(cnt < (size_t)lim ? cnt : (size_t)lim);
    memcpy(out, cand, (size_t)nout * sizeof(orc_feature));
    free(cand); free(resp);
    return nout;
}


/* ---- FAST-9 segment test (first stage of cv::ORB::detect as configured at src/imagefeatures.cpp:73-87:
 * fastThreshold 20, FAST 9/16 with non-maximum suppression).  OpenCV is not in this build: the contract
 * below restates the published detector (Rosten & Drummond) with OpenCV's conventions and is unpinned by
 * the reference.
 *   ring: the 16 pixels of the radius-3 Bresenham circle, clockwise from (0, +3);
 *   strength s(p) = max over the 16 arcs of 9 contiguous ring pixels of min over the arc of (I_q - I_p),
 *                   and the same with (I_p - I_q); the larger of the two;
 *   p is a corner iff s(p) > threshold (9 contiguous pixels all brighter than I_p + t or all darker than
 *   I_p - t); its score is s(p) - 1, the largest threshold at which it is still a corner;
 *   pixels closer than 3 to the image border are never corners;
 *   nonmax: a corner is kept iff its strength exceeds that of every 8-neighbour that is a corner;
 *   output sorted by score descending, ties by y then x ascending (feat_cmp). */
static const int fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int orc_fast9_strength(const uint8_t *gray, int w, int h, int stride, int x, int y)
{
    if (x < 3 || y < 3 || x >= w - 3 || y >= h - 3) return 0;
    int v = gray[x + (size_t)stride * y], d[16], best = 0;
    for (int i = 0; i < 16; ++i) d[i] = (int)gray[(x + fast_dx[i]) + (size_t)stride * (y + fast_dy[i])] - v;
    for (int pol = 0; pol < 2; ++pol)
        for (int a = 0; a < 16; ++a) {
            int m = 1 << 30;
            for (int k = 0; k < 9; ++k) {
                int q = pol ? -d[(a + k) & 15] : d[(a + k) & 15];
                if (q < m) m = q;
            }
            if (m > best) best = m;
        }
    return best;
}

int orc_fast9_detect(const uint8_t *gray, int w, int h, int stride, int threshold, int nonmax, int max_features,
                     orc_feature *out, int cap)
{
    size_t npx = (size_t)w * h;
    int *st = (int *)calloc(npx, sizeof(int));
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int s = orc_fast9_strength(gray, w, h, stride, x, y);
            st[x + (size_t)w * y] = s > threshold ? s : 0;
        }
    size_t cnt = 0, capc = 1024;
    orc_feature *cand = (orc_feature *)malloc(capc * sizeof(orc_feature));
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int s = st[x + (size_t)w * y];
            if (!s) continue;
            int ok = 1;
            if (nonmax)
                for (int dy = -1; dy <= 1 && ok; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                        if ((dx || dy) && !(s > st[(x + dx) + (size_t)w * (y + dy)])) { ok = 0; break; }
            if (!ok) continue;
            if (cnt == capc) { capc *= 2; cand = (orc_feature *)realloc(cand, capc * sizeof(orc_feature)); }
            cand[cnt].x = x; cand[cnt].y = y; cand[cnt].score = (float)(s - 1); ++cnt;
        }
    qsort(cand, cnt, sizeof(orc_feature), feat_cmp);
    int lim = max_features < cap ? max_features : cap;
    int nout = (int)(cnt < (size_t)lim ? cnt : (size_t)lim);
    memcpy(out, cand, (size_t)nout * sizeof(orc_feature));
    free(cand); free(st);
    return nout;
}


/* ---- ORB keypoints of one pyramid level: cv::ORB::detect as configured at src/imagefeatures.cpp:73-87
 * (HARRIS_SCORE, edgeThreshold 75, fastThreshold 20).  PARITY UNPINNED: OpenCV is not in the reference tree, no
 * reference test covers it (test/src/matcher.cpp is commented out); this restates the published algorithm of
 * OpenCV 4.5's ORB detector and is the contract the GPU path is held to:
 *   1. FAST-9/16 at fast_threshold with non-maximum suppression (orc_fast9_detect);
 *   2. KeyPointsFilter::runByImageBorder: keep edge <= x < w - edge, edge <= y < h - edge;
 *   3. KeyPointsFilter::retainBest(2 n) on the FAST score: the 2 n strongest and every tie of the last of them;
 *   4. HarrisResponses at those corners: over the block x block window centred on the corner,
 *        Ix = 2 (p[+1] - p[-1]) + (p[-s+1] - p[-s-1]) + (p[s+1] - p[s-1]),  Iy alike with rows and columns swapped,
 *        a = sum Ix^2, b = sum Iy^2, c = sum Ix Iy in int,
 *        response = ((float)a b - (float)c c - k ((float)a + b) ((float)a + b)) * scale^4, scale = 1 / (4 block 255),
 *      every float operation rounded once, scale^4 formed as ((scale scale) scale) scale;
 *   5. retainBest(n) on the response, ties included.
 * Output sorted by response descending, ties by y then x.  Returns the count, or -1 when cap is too small. */
static int retain_best_sorted(orc_feature *f, int cnt, int n)
{
    if (n <= 0) return 0;
    if (cnt <= n) return cnt;
    float cut = f[n - 1].score;
    int k = n;
    while (k < cnt && f[k].score == cut) ++k;
    return k;
}

int orc_orb_detect_level(const uint8_t *gray, int w, int h, int stride, int nfeatures, int fast_threshold, int edge,
                         int block, float harris_k, orc_feature *out, int cap)
{
    int capc = w * h / 4 + 16;
    orc_feature *cand = (orc_feature *)malloc((size_t)capc * sizeof(orc_feature));
    int cnt = orc_fast9_detect(gray, w, h, stride, fast_threshold, 1, capc, cand, capc);
    int m = 0;
    for (int i = 0; i < cnt; ++i)
        if (cand[i].x >= edge && cand[i].x < w - edge && cand[i].y >= edge && cand[i].y < h - edge) cand[m++] = cand[i];
    m = retain_best_sorted(cand, m, 2 * nfeatures);
    volatile float scale = 1.0f / (4.0f * (float)block * 255.0f);
    volatile float s2 = scale * scale;
    volatile float s3 = s2 * scale;
    volatile float s4 = s3 * scale;
    const int r = block / 2;
    for (int i = 0; i < m; ++i) {
        int a = 0, b = 0, c = 0;
        for (int dy = -r; dy < block - r; ++dy)
            for (int dx = -r; dx < block - r; ++dx) {
                const uint8_t *p = gray + (size_t)(cand[i].y + dy) * stride + cand[i].x + dx;
                int Ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-stride + 1] - (int)p[-stride - 1]) + ((int)p[stride + 1] - (int)p[stride - 1]);
                int Iy = ((int)p[stride] - (int)p[-stride]) * 2 + ((int)p[stride - 1] - (int)p[-stride - 1]) + ((int)p[stride + 1] - (int)p[-stride + 1]);
                a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
            }
        volatile float fa = (float)a, fb = (float)b, fc = (float)c;
        volatile float ab = fa * fb, cc = fc * fc, tr = fa + fb;
        volatile float ktr = harris_k * tr;
        volatile float ktr2 = ktr * tr;
        volatile float d1 = ab - cc;
        volatile float d2 = d1 - ktr2;
        volatile float resp = d2 * s4;
        cand[i].score = resp;
    }
    qsort(cand, (size_t)m, sizeof(orc_feature), feat_cmp);
    m = retain_best_sorted(cand, m, nfeatures);
    int ret = m;
    if (m > cap) ret = -1;
    else memcpy(out, cand, (size_t)m * sizeof(orc_feature));
    free(cand);
    return ret;
}

/* The level-to-level resize of the pyramid as this build defines it (OpenCV's INTER_LINEAR_EXACT arithmetic is not in
 * the reference tree: PARITY UNPINNED): bilinear at source position (x + 1/2) sw / dw - 1/2, held in units of 1/2048
 * pixel by integer arithmetic, clamped to the image; weights in 1/2048, result rounded to nearest. */
void orc_resize_bilinear(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride)
{
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            long long X = ((long long)(2 * x + 1) * sw * 1024) / dw - 1024, Y = ((long long)(2 * y + 1) * sh * 1024) / dh - 1024;
            if (X < 0) X = 0;
            if (X > (long long)(sw - 1) * 2048) X = (long long)(sw - 1) * 2048;
            if (Y < 0) Y = 0;
            if (Y > (long long)(sh - 1) * 2048) Y = (long long)(sh - 1) * 2048;
            int x0 = (int)(X >> 11), y0 = (int)(Y >> 11), ax = (int)(X & 2047), ay = (int)(Y & 2047);
            int x1 = x0 + 1 < sw ? x0 + 1 : sw - 1, y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
            unsigned p00 = src[(size_t)y0 * sstride + x0], p01 = src[(size_t)y0 * sstride + x1];
            unsigned p10 = src[(size_t)y1 * sstride + x0], p11 = src[(size_t)y1 * sstride + x1];
            unsigned v = p00 * (2048 - ax) * (2048 - ay) + p01 * ax * (2048 - ay) + p10 * (2048 - ax) * ay + p11 * ax * ay;
            dst[(size_t)y * dstride + x] = (uint8_t)((v + (1u << 21)) >> 22);
        }
}

/* Level sizes, feature budgets and scales of cv::ORB's pyramid (firstLevel 0): scale_l = (float)pow(scaleFactor, l),
 * size = cvRound(cols / scale_l) x cvRound(rows / scale_l), budgets by the geometric series with factor 1 / scaleFactor,
 * the last level taking the remainder.  PARITY UNPINNED (restated from the published OpenCV 4.5 algorithm). */
void orc_orb_level_plan(int nfeatures, float scale_factor, int nlevels, int w, int h, int *lw, int *lh, int *lf, float *ls)
{
    float factor = (float)(1.0 / (double)scale_factor);
    float desired = (float)nfeatures * (1.0f - factor) / (1.0f - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels; ++l) {
        float sc = (float)pow((double)scale_factor, (double)l);
        ls[l] = sc;
        lw[l] = (int)nearbyint((double)((float)w / sc));
        lh[l] = (int)nearbyint((double)((float)h / sc));
        if (l < nlevels - 1) { lf[l] = (int)nearbyint((double)desired); sum += lf[l]; desired *= factor; }
        else lf[l] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    }
}


/* ---- caller-side selection of new landmarks, src/SLAM.cpp:449-472 with pixelDistance, src/SLAM.cpp:746-760
 * (pinned by test/src/pixelDistance.cpp).  The loop index of pixelDistance is uninitialised in the reference
 * (`for(int i; ...)`); it is defined as 0 here.  Arithmetic as there: cv::Point2f difference, float products and
 * sum, float square root, comparison with the double threshold. */
int orc_pixel_distance(const float *landmark_xy, int L, float x, float y, double thresh)
{
    int within = 0;
    for (int i = 0; i < L; ++i) {
        volatile float dx = landmark_xy[2 * i] - x, dy = landmark_xy[2 * i + 1] - y;
        volatile float xx = dx * dx, yy = dy * dy;
        volatile float d2 = xx + yy;
        float dist = sqrtf(d2);
        if ((double)dist < thresh) within = 1;
    }
    return within;
}

/* feats: K records sorted by score descending (src/SLAM.cpp:457).  landmark_xy: 2 x L pixel positions of the
 * landmarks already in the map.  A feature is taken iff capacity is left, no landmark (old or just taken) is
 * within the radius, it lies inside the margins and its score exceeds the threshold (src/SLAM.cpp:459-472).
 * Returns the number taken; selected[] receives their indices into feats, in order. */
int orc_select_new_landmarks(const orc_feature *feats, int K, const float *landmark_xy, int L, double feature_thresh,
                             double width_thresh, double height_thresh, double dist_thresh, int image_w, int image_h,
                             int max_new, int32_t *selected, int cap)
{
    float *lm = (float *)malloc(sizeof(float) * 2 * (size_t)(L + (max_new > 0 ? max_new : 0) + 1));
    memcpy(lm, landmark_xy, sizeof(float) * 2 * (size_t)L);
    int nl = L, taken = 0;
    for (int i = 0; i < K; ++i) {
        int room = max_new - taken;
        float fx = (float)feats[i].x, fy = (float)feats[i].y;
        int score_ok = (double)feats[i].score > feature_thresh;
        int in_w = width_thresh <= (double)fx && (double)fx <= (double)image_w - width_thresh;
        int in_h = height_thresh <= (double)fy && (double)fy <= (double)image_h - height_thresh;
        int within = nl > 0 ? orc_pixel_distance(lm, nl, fx, fy, dist_thresh) : 0;
        if (room > 0 && !within && in_w && in_h && score_ok) {
            lm[2 * nl] = fx; lm[2 * nl + 1] = fy; ++nl;
            if (taken < cap) selected[taken] = i;
            ++taken;
        }
    }
    free(lm);
    return taken;
}


/* ---- initialisation of a new landmark, src/SLAM.cpp:905-941 with generateOpticalRay, src/SLAM.cpp:876-903.
 * generateOpticalRay calls cv::undistortPoints(pixel, K, distCoeffs, R = I, P = [K | 0]); OpenCV is not in this
 * build, so its documented iteration is restated (unpinned): normalise with K, five fixed-point steps
 *   x <- (x0 - dx(x, y)) * icdist(x, y),  icdist = (1 + k4 r2 + k5 r2^2 + k6 r2^3) / (1 + k1 r2 + k2 r2^2 + k3 r2^3),
 *   dx = 2 p1 x y + p2 (r2 + 2 x^2) + s1 r2 + s2 r2^2,  dy = p1 (r2 + 2 y^2) + 2 p2 x y + s3 r2 + s4 r2^2
 * with the coefficients in OpenCV's order (whatever the flag says), the result back to pixels with P and
 * stored as cv::Point2f (float).  The ray is K^-1 [u v 1]^T times the ray length.
 * gcc 11.4 -O3 pairs the x and y updates of the loop into SSE lanes and the result then differs from the
 * scalar evaluation (and from clang's, and from -O2) in the seventh digit: pairing is switched off here. */
__attribute__((optimize("no-tree-slp-vectorize")))
void orc_undistort_point(const orc_camera *cam, double u, double v, float *out_uv)
{
    const double *d = cam->d;
    const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4], k4 = d[5], k5 = d[6], k6 = d[7];
    const double s1 = d[8], s2 = d[9], s3 = d[10], s4 = d[11];
    /* the reference passes the pixel through a cv::Point2f */
    double x0 = ((double)(float)u - cam->cx) / cam->fx, y0 = ((double)(float)v - cam->cy) / cam->fy;
    double x = x0, y = y0;
    for (int it = 0; it < 5; ++it) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + s1 * r2 + s2 * r2 * r2;
        double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + s3 * r2 + s4 * r2 * r2;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    out_uv[0] = (float)(x * cam->fx + cam->cx);
    out_uv[1] = (float)(y * cam->fy + cam->cy);
}

void orc_optical_ray(const orc_camera *cam, double u, double v, double length, double *r)
{
    float uv[2];
    orc_undistort_point(cam, u, v, uv);
    /* K^-1 [u v 1]^T for K = [fx 0 cx; 0 fy cy; 0 0 1] */
    r[0] = length * (((double)uv[0] - cam->cx) / cam->fx);
    r[1] = length * (((double)uv[1] - cam->cy) / cam->fy);
    r[2] = length;
}

/* mu_out (n+3), S_out ((n+3) x (n+3), ld n+3): state and factor with the landmark appended (src/SLAM.cpp:906-929). */
void orc_initialize_landmark(const double *mu, const double *S, int n, int ldS, const orc_camera *cam, double u, double v,
                             double length, double kappa, double *mu_out, double *S_out)
{
    int m = n + 3;
    memset(S_out, 0, sizeof(double) * (size_t)m * m);
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) S_out[r + (size_t)m * c] = S[r + (size_t)ldS * c];
    for (int k = 0; k < 3; ++k) S_out[(n + k) + (size_t)m * (n + k)] = kappa;
    memcpy(mu_out, mu, sizeof(double) * (size_t)n);
    double ray[3], R[9];
    orc_optical_ray(cam, u, v, length, ray);
    orc_rpy2rot(mu + 9, R);
    for (int i = 0; i < 3; ++i)
        mu_out[n + i] = mu[6 + i] + (R[i + 3 * 0] * ray[0] + R[i + 3 * 1] * ray[1] + R[i + 3 * 2] * ray[2]);
}

/* ------------------------------------------------------------------ */
/* ORB orientation and rBRIEF descriptors: the compute() half of cv::ORB as detectAndDrawORB calls it
 * (reference src/imagefeatures.cpp:88-96; the descriptors feed BFMatcher(NORM_HAMMING), src/SLAM.cpp:222,355).
 * OpenCV is a dependency that is not part of the reference tree, so what follows restates the published ORB
 * algorithm (Rublee et al. 2011) with OpenCV's conventions as documented; parity unpinned.
 *   orientation: intensity centroid of the circular patch of radius 15 around the keypoint on the level image,
 *                angle = atan2(m01, m10) in degrees by the 7th-order odd polynomial of cv::fastAtan2;
 *   blur:        7 x 7 Gaussian, sigma 2, reflect-101 borders, on the level image.  Fixed point: the weights are
 *                round(256 g_i) with the centre weight taking the remainder {18, 34, 49, 54, 49, 34, 18}, the row
 *                pass is exact, the column pass rounds (sum + 2^15) >> 16.  This library's own definition (OpenCV's
 *                8-bit GaussianBlur uses its own fixed-point kernel and is not reproduced).
 *   descriptor:  256 binary tests I(p0) < I(p1) on the blurred level image, the test points rotated by the angle:
 *                x' = round(x a - y b), y' = round(x b + y a), a = (float)cos, b = (float)sin of the angle in
 *                radians (float), products and sums in float, rounding half to even; bit k of byte i = test 8 i + k.
 *   pattern:     256 x (x0, y0, x1, y1), |coordinate| <= 13.  cv::ORB uses a learned table that exists only in
 *                OpenCV's sources; a caller who wants cv::ORB's bits passes that table.  The default below is a fixed
 *                pseudo-random table of this library. */
static const int orc_gauss7_w[7] = {18, 34, 49, 54, 49, 34, 18};
static int orc_reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i; return i; }

void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            unsigned acc = 0;
            for (int j = -3; j <= 3; ++j) {
                const uint8_t *row = src + (size_t)orc_reflect101(y + j, h) * sstride;
                unsigned hsum = 0;
                for (int i = -3; i <= 3; ++i) hsum += (unsigned)orc_gauss7_w[i + 3] * row[orc_reflect101(x + i, w)];
                acc += (unsigned)orc_gauss7_w[j + 3] * hsum;
            }
            dst[(size_t)y * dstride + x] = (uint8_t)((acc + 32768u) >> 16);
        }
}

/* half-widths of the rows of the circular patch of radius 15: u_max[v], v = 0..15 */
void orc_orb_umax(int *umax)
{
    const int hp = 15;
    int v, v0, vmax = (int)floor(hp * sqrt(2.0) / 2 + 1), vmin = (int)ceil(hp * sqrt(2.0) / 2);
    for (v = 0; v <= hp + 1; ++v) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = (int)nearbyint(sqrt((double)hp * hp - (double)v * v));
    for (v = hp, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* angle in degrees in [0, 360): the polynomial approximation of cv::fastAtan2 (maximum error about 0.3 degrees) */
float orc_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.220446049250313e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.220446049250313e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

float orc_orb_ic_angle(const uint8_t *gray, int stride, int x, int y, int *m01_out, int *m10_out)
{
    int umax[17];
    orc_orb_umax(umax);
    int m01 = 0, m10 = 0;
    for (int v = -15; v <= 15; ++v) {
        const int d = umax[v < 0 ? -v : v];
        const uint8_t *row = gray + (size_t)(y + v) * stride + x;
        for (int u = -d; u <= d; ++u) { m10 += u * row[u]; m01 += v * row[u]; }
    }
    if (m01_out) *m01_out = m01;
    if (m10_out) *m10_out = m10;
    return orc_fast_atan2((float)m01, (float)m10);
}

void orc_orb_default_pattern(int8_t *pattern)
{
    uint32_t st = 0x34985739u;
    for (int i = 0; i < 256; ++i) {
        int c[4];
        do {
            for (int k = 0; k < 4; ++k) { st = st * 1664525u + 1013904223u; c[k] = (int)((st >> 16) % 27u) - 13; }
        } while (c[0] == c[2] && c[1] == c[3]);
        for (int k = 0; k < 4; ++k) pattern[4 * i + k] = (int8_t)c[k];
    }
}

void orc_orb_describe(const uint8_t *blur, int stride, int x, int y, float angle_deg, const int8_t *pattern, uint8_t *desc)
{
    const float angle = angle_deg * (float)(3.14159265358979323846 / 180.0);
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t *center = blur + (size_t)y * stride + x;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            const int8_t *t = pattern + 4 * (8 * i + k);
            int v[2];
            for (int q = 0; q < 2; ++q) {
                const float px = (float)t[2 * q], py = (float)t[2 * q + 1];
                const float fx = px * a - py * b, fy = px * b + py * a;
                const int ix = (int)nearbyintf(fx), iy = (int)nearbyintf(fy);
                v[q] = center[(long)iy * stride + ix];
            }
            val |= (v[0] < v[1]) << k;
        }
        desc[i] = (uint8_t)val;
    }
}
