// Time update of the square-root filter: timeUpdateContinuous (reference src/gaussian.hpp:337-383)
// with SlamProcessModel (src/model.cpp:9-136) and the RK4 step with Jacobian propagation
// (RK4SDEHelper, src/gaussian.hpp:171-239).
//
// The reference pushes the augmented state [x; dw] (2n) through the RK4 map, multiplies the
// 2n x 2n factor blkdiag(S, SQ sqrt(dt)) by the n x 2n Jacobian and takes the R of a dense
// 3n x n Householder QR.  The process model only moves the camera pose from the body velocities
// and the attitude, so
//   * f and its Jacobian live in a 12-state (nu, eta); every landmark row of the RK4 Jacobian is
//     [e_i, e_i]: the RK4 recursion is done on a 12 x 25 array by one small workgroup;
//   * S Jx^T differs from S only in columns 6..11 (eta), each a combination of S's first 12 columns;
//   * SQ has six non-zero rows, so of the lower 2n rows of the reference's matrix six survive.
// The QR therefore runs on an (n + 6) x n matrix with the same R (rows of zeros change neither the
// reflectors nor their signs) through the blocked Householder kernels of qr.hip.
#include "linalg.hpp"

namespace vslam {

struct TuNoise {
    double sq6[36];      // blkdiag(position_tune, orientation_tune) * sqrt(dt), 6 x 6 column-major
    double dt;
};

constexpr int TU_COLS = 25;      // [state, d/dx(0..11), d/ddw(0..11)]

// f(6..11) and the rows 6..11 of df/dx (6 x 12, row-major in Jm) at the 12-state x.
__device__ void process_model12(const double *x, double *f6, double (*Jm)[12])
{
    const double phi = x[9], theta = x[10], psi = x[11];
    const double sph = sin(phi), cph = cos(phi), sth = sin(theta), cth = cos(theta), sps = sin(psi), cps = cos(psi);
    const double tth = tan(theta);
    // Rnc = Rz(psi) Ry(theta) Rx(phi)  (src/rotation.hpp:22-63)
    const double R[3][3] = {{cps * cth, cps * sth * sph - sps * cph, cps * sth * cph + sps * sph},
                            {sps * cth, sps * sth * sph + cps * cph, sps * sth * cph - cps * sph},
                            {-sth, cth * sph, cth * cph}};
    const double T[3][3] = {{1.0, sph * tth, cph * tth}, {0.0, cph, -sph}, {0.0, sph / cth, cph / cth}};
    for (int r = 0; r < 6; ++r)
        for (int k = 0; k < 12; ++k) Jm[r][k] = 0.0;
    for (int r = 0; r < 3; ++r) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 3; ++k) {
            a += R[r][k] * x[k];
            b += T[r][k] * x[3 + k];
            Jm[r][k] = R[r][k];
            Jm[3 + r][3 + k] = T[r][k];
        }
        f6[r] = a; f6[3 + r] = b;
    }
    const double x1dot = x[0], x2dot = x[1], x3dot = x[2], thetadot = x[4], psidot = x[5];
    // d f / d (phi, theta, psi), src/model.cpp:117-132
    Jm[0][9] = x2dot * (sph * sps + cph * cps * sth) + x3dot * (cph * sps - cps * sph * sth);
    Jm[1][9] = -x2dot * (cps * sph - cph * sps * sth) - x3dot * (cph * cps + sph * sps * sth);
    Jm[2][9] = x2dot * cph * cth - x3dot * cth * sph;
    Jm[3][9] = thetadot * cph * tth - psidot * sph * tth;
    Jm[4][9] = -psidot * cph - thetadot * sph;
    Jm[5][9] = (thetadot * cph) / cth - (psidot * sph) / cth;
    Jm[0][10] = x3dot * cph * cps * cth - x1dot * cps * sth + x2dot * cps * cth * sph;
    Jm[1][10] = x3dot * cph * cth * sps - x1dot * sps * sth + x2dot * cth * sph * sps;
    Jm[2][10] = -x1dot * cth - x3dot * cph * sth - x2dot * sph * sth;
    Jm[3][10] = psidot * cph * (tth * tth + 1.0) + thetadot * sph * (tth * tth + 1.0);
    Jm[4][10] = 0.0;
    Jm[5][10] = (psidot * cph * sth) / (cth * cth) + (thetadot * sph * sth) / (cth * cth);
    Jm[0][11] = x3dot * (cps * sph - cph * sps * sth) - x2dot * (cph * cps + sph * sps * sth) - x1dot * cth * sps;
    Jm[1][11] = x3dot * (sph * sps + cph * cps * sth) - x2dot * (cph * sps - cps * sph * sth) + x1dot * cps * cth;
}

// One workgroup per filter instance; lane c < 25 carries column c of the 12 x 25 RK4 arrays
// (X = [x, I, 0], dW = [0, 0, I]).  Writes the propagated mean and the 12 x 24 Jacobian block
// [d/dx(0..11), d/ddw(0..11)] (column-major, 12 rows) to jac.
__global__ __launch_bounds__(64) void tu_rk4_kernel(const double *mu, size_t stride_mu, int n, double dt,
                                                    double *mu_out, size_t stride_out, double *jac)
{
    __shared__ double xs[12];
    const int c = threadIdx.x, inst = blockIdx.x;
    const double *m = mu + (size_t)inst * stride_mu;
    double *mo = mu_out + (size_t)inst * stride_out;
    for (int i = 12 + c; i < n; i += 64) mo[i] = m[i];                 // landmarks do not move
    double X[12], dW[12], Xs[12], F[4][12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        X[i] = (c == 0) ? m[i] : ((c == i + 1) ? 1.0 : 0.0);
        dW[i] = (c == i + 13) ? 1.0 : 0.0;
        Xs[i] = X[i];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        __syncthreads();
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) xs[i] = Xs[i];
        }
        __syncthreads();
        double xl[12], f6[6], Jm[6][12];
#pragma unroll
        for (int i = 0; i < 12; ++i) xl[i] = xs[i];
        process_model12(xl, f6, Jm);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            F[s][r] = 0.0;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 12; ++k) a += Jm[r][k] * Xs[k];
            F[s][6 + r] = (c == 0) ? f6[r] : a;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (s < 2) Xs[i] = X[i] + (F[s][i] * dt + dW[i]) / 2;
            else if (s == 2) Xs[i] = X[i] + F[s][i] * dt + dW[i];
        }
    }
    if (c >= TU_COLS) return;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const double v = X[i] + (F[0][i] + 2 * F[1][i] + 2 * F[2][i] + F[3][i]) * dt / 6 + dW[i];
        if (c == 0) mo[i] = v;
        else jac[(size_t)inst * 288 + (size_t)(c - 1) * 12 + i] = v;
    }
}

// A = [S Jx^T ; sqrt(dt) SQ6 Jw^T] ((n + 6) x n, leading dimension lda) from S and the 12 x 24 block.
__global__ __launch_bounds__(256) void tu_build_kernel(const double *S, int ldS, size_t strideS, int n, const double *jac,
                                                       TuNoise nz, double *A, int lda, size_t strideA)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, inst = blockIdx.z;
    if (r >= n + 6) return;
    const double *Si = S + (size_t)inst * strideS;
    const double *J = jac + (size_t)inst * 288;
    double v;
    if (r < n) {
        if (c < 6 || c >= 12) v = Si[(size_t)c * ldS + r];
        else {
            v = 0.0;
            for (int k = 0; k < 12; ++k) v += Si[(size_t)k * ldS + r] * J[k * 12 + c];
        }
    } else {
        v = 0.0;
        if (c < 12) {
            const int rr = r - n;
            for (int k = 0; k < 6; ++k) v += nz.sq6[rr + 6 * k] * J[144 + k * 12 + c];
        }
    }
    A[(size_t)inst * strideA + (size_t)c * lda + r] = v;
}

__global__ void tu_copy_out_kernel(const double *A, int lda, size_t strideA, int n, double *S_out, int ldo, size_t strideO)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, inst = blockIdx.z;
    if (r < n) S_out[(size_t)inst * strideO + (size_t)c * ldo + r] = A[(size_t)inst * strideA + (size_t)c * lda + r];
}

static int tu_lda(int n) { return (n + 6 + 7) & ~7; }

size_t time_update_ws_bytes(int n, int chunk)
{
    return Arena::pad((size_t)chunk * 288 * 8) + Arena::pad((size_t)chunk * tu_lda(n) * n * 8) + qr_ws_bytes(n + 6, n, chunk);
}

// Instances per pass: the (n + 6) x n work matrices of one pass stay under 1 GiB.
int time_update_chunk(int n, int batch)
{
    const size_t per = (size_t)tu_lda(n) * n * 8;
    size_t c = ((size_t)1 << 30) / per;
    if (c < 1) c = 1;
    if (c > 32768) c = 32768;                       // grid and batch limits of the kernels below
    return (size_t)batch < c ? batch : (int)c;
}

// All pointers device.  mu_out may alias mu and S_out may alias S.
vslam_status time_update_dev(vslam_ctx *ctx, const double *mu, size_t stride_mu, const double *S, int ldS, size_t strideS, int n,
                             int batch, const TuNoise &nz, double *mu_out, size_t stride_muo, double *S_out, int ldo, size_t strideO,
                             void *ws)
{
    hipStream_t s = ctx->stream;
    const int chunk = time_update_chunk(n, batch), lda = tu_lda(n);
    const size_t strideA = (size_t)lda * n;
    Arena ar(ws);
    double *jac = ar.take<double>((size_t)chunk * 288);
    double *A = ar.take<double>((size_t)chunk * strideA);
    void *qws = ar.take<char>(qr_ws_bytes(n + 6, n, chunk));
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = (batch - b0) < chunk ? (batch - b0) : chunk;
        hipLaunchKernelGGL(tu_rk4_kernel, dim3(nb), dim3(64), 0, s, mu + (size_t)b0 * stride_mu, stride_mu, n, nz.dt,
                           mu_out + (size_t)b0 * stride_muo, stride_muo, jac);
        hipLaunchKernelGGL(tu_build_kernel, dim3((n + 6 + 255) / 256, n, nb), dim3(256), 0, s, S + (size_t)b0 * strideS, ldS, strideS, n,
                           jac, nz, A, lda, strideA);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(qr_upper_dev(ctx, A, n + 6, n, lda, strideA, nb, qws));
        hipLaunchKernelGGL(tu_copy_out_kernel, dim3((n + 255) / 256, n, nb), dim3(256), 0, s, A, lda, strideA, n,
                           S_out + (size_t)b0 * strideO, ldo, strideO);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
    }
    return VSLAM_OK;
}

static vslam_status make_noise(vslam_ctx *ctx, const double *pos_tune, const double *ori_tune, double dt, TuNoise *nz)
{
    if (!pos_tune || !ori_tune) return fail(ctx, VSLAM_ERR_ARG, "time_update: null tuning matrix");
    if (!(dt >= 0.0)) return fail(ctx, VSLAM_ERR_ARG, "time_update: timestep %g", dt);
    const double sq = sqrt(dt);
    for (int i = 0; i < 36; ++i) nz->sq6[i] = 0.0;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            nz->sq6[r + 6 * c] = pos_tune[r + 3 * c] * sq;
            nz->sq6[3 + r + 6 * (3 + c)] = ori_tune[r + 3 * c] * sq;
        }
    nz->dt = dt;
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_time_update(vslam_ctx *ctx, const double *mu, const double *S, int n, int ldS, const double *pos_tune,
                                          const double *ori_tune, double dt, double *mu_out, double *S_out, int ldSout)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "time_update: null pointer");
    if (n < 12 || ldS < n || ldSout < n) return fail(ctx, VSLAM_ERR_SHAPE, "time_update: n=%d ldS=%d ldSout=%d", n, ldS, ldSout);
    TuNoise nz;
    VSLAM_TRY(make_noise(ctx, pos_tune, ori_tune, dt, &nz));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(ws_reserve(ctx, 2 * Arena::pad(n * 8) + 2 * Arena::pad(nn * 8) + time_update_ws_bytes(n, 1)));
    Arena ar(ctx->ws);
    double *dmu = ar.take<double>(n), *dmo = ar.take<double>(n), *dS = ar.take<double>(nn), *dSo = ar.take<double>(nn);
    void *tws = ar.take<char>(time_update_ws_bytes(n, 1));
    VSLAM_TRY(h2d(ctx, dmu, mu, n * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    VSLAM_TRY(time_update_dev(ctx, dmu, n, dS, n, nn, n, 1, nz, dmo, n, dSo, n, nn, tws));
    VSLAM_TRY(d2h(ctx, mu_out, dmo, n * 8));
    VSLAM_TRY(d2h_2d(ctx, S_out, (size_t)ldSout * 8, dSo, (size_t)n * 8, (size_t)n * 8, n));
    return stream_finish(ctx);
}

extern "C" vslam_status vslam_time_update_batch_dev(vslam_ctx *ctx, const double *mu, const double *S, int n, int batch,
                                                    const double *pos_tune, const double *ori_tune, double dt, double *mu_out,
                                                    double *S_out)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "time_update: null pointer");
    if (n < 12 || batch < 1) return fail(ctx, VSLAM_ERR_SHAPE, "time_update: n=%d batch=%d", n, batch);
    TuNoise nz;
    VSLAM_TRY(make_noise(ctx, pos_tune, ori_tune, dt, &nz));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int chunk = time_update_chunk(n, batch);
    VSLAM_TRY(ws_reserve(ctx, time_update_ws_bytes(n, chunk)));
    const size_t nn = (size_t)n * n;
    return time_update_dev(ctx, mu, n, S, n, nn, n, batch, nz, mu_out, n, S_out, n, nn, ctx->ws);
}
