// Iterated-EKF measurement update (reference src/gaussian.hpp:469-532 driving src/fmin.hpp:19-107 and
// src/fmin.cpp:40-131) for a bank of independent filters in lock step.
//
// One filter's update is a chain of a few hundred small dependent kernels (newton.hip): a single
// instance leaves the chip idle and several host threads with their own contexts do not scale.  Here
// every kernel of the chain takes the whole bank -- instance = a grid dimension -- and the host keeps
// the per-instance scalars of the trust-region iteration (radius, multiplier, cost, state of the
// loop), sends a mask of the instances that take part in the next kernels and reads back a few
// doubles per instance per step.  The arithmetic per instance is that of the device route of
// newton.hip (Cholesky factors instead of the eigendecomposition; identical sequence of multipliers);
// an instance whose Hessian is not positive definite is reported (status 1) for the caller to redo
// with vslam_measurement_update_iekf.
#include "landmark_math.hpp"
#include "linalg.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace vslam {

namespace {

constexpr int BK_T = 256;

// Triangular solve with the upper-triangular S of every active instance, one workgroup each:
// trans = 1: S^T z = rhs (blocks first to last), trans = 0: S z = rhs (last to first); out = scale * z.
// The right-hand side lives in LDS; per 64-wide diagonal block the block is staged in LDS, wave 0
// substitutes (lane k owns unknown k, finished unknowns are broadcast with v_readlane) and the whole
// workgroup removes the block's contribution from the unknowns still open.
__global__ __launch_bounds__(BK_T) void bk_trsv_kernel(const double *S, int ldS, size_t strideS, int n, int trans, const double *rhs,
                                                       double *out, double scale, const int *active)
{
    extern __shared__ double bk_lds[];
    double *z = bk_lds;                         // [n] (padded to even)
    double *Ud = z + ((n + 1) & ~1);            // [64][65]
    double *xs = Ud + 64 * 65;                  // [64]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (active && !active[b]) return;
    const double *Sb = S + (size_t)b * strideS;
    for (int i = tid; i < n; i += BK_T) z[i] = rhs[(size_t)b * n + i];
    const int nblk = (n + 63) / 64;
    for (int step = 0; step < nblk; ++step) {
        const int blk = trans ? step : nblk - 1 - step, i0 = blk * 64, ib = min(64, n - i0);
        __syncthreads();
        for (int idx = tid; idx < 64 * 64; idx += BK_T) {
            const int r = idx & 63, c = idx >> 6;
            double v = (r == c) ? 1.0 : 0.0;
            if (r < ib && c < ib) v = (r <= c) ? Sb[(size_t)(i0 + c) * ldS + i0 + r] : 0.0;
            Ud[r * 65 + c] = v;
        }
        __syncthreads();
        if (tid < 64) {
            const int k = tid;
            double acc = (k < ib) ? z[i0 + k] : 0.0, xk = 0.0;
            const double rd = 1.0 / Ud[k * 65 + k];
            double u[64];                               // this lane's coefficients: column k (trans) or row k of the block
#pragma unroll
            for (int l = 0; l < 64; ++l) u[l] = trans ? Ud[l * 65 + k] : Ud[k * 65 + l];
            if (trans) {
#pragma unroll
                for (int l = 0; l < 64; ++l) {
                    const double xl = readlane_f64(acc, l) * readlane_f64(rd, l);
                    if (k == l) xk = xl;
                    acc -= ((k > l) ? u[l] : 0.0) * xl;
                }
            } else {
#pragma unroll
                for (int l = 63; l >= 0; --l) {
                    const double xl = readlane_f64(acc, l) * readlane_f64(rd, l);
                    if (k == l) xk = xl;
                    acc -= ((k < l) ? u[l] : 0.0) * xl;
                }
            }
            xs[k] = xk;
            if (k < ib) z[i0 + k] = xk;
        }
        __syncthreads();
        if (trans) {
            // later columns j: z[j] -= S[i0.., j] . x.  Eight lanes share a column, each holding the rows
            // 16 q + 2 sub + {0, 1}, so that one load instruction of the wave covers whole cache lines of
            // eight columns (a thread per column would touch 64 lines per instruction).  Only full blocks
            // have later columns, so all 64 rows exist.
            const int sub = tid & 7, grp = tid >> 3;
            double xr[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xr[2 * q] = xs[16 * q + 2 * sub]; xr[2 * q + 1] = xs[16 * q + 2 * sub + 1]; }
            for (int j0 = i0 + ib; j0 < n; j0 += BK_T / 8) {
                const int j = j0 + grp;
                double sum = 0.0;
                if (j < n) {
                    const double *col = Sb + (size_t)j * ldS + i0 + 2 * sub;
                    double v[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[2 * q] = col[16 * q]; v[2 * q + 1] = col[16 * q + 1]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum += v[q] * xr[q];
                }
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                sum += __shfl_xor(sum, 4, 64);
                if (sub == 0 && j < n) z[j] -= sum;
            }
        } else {
            for (int i = tid; i < i0; i += BK_T) {
                double s0 = 0.0, s1 = 0.0;
                int l = 0;
                for (; l + 2 <= ib; l += 2) {
                    s0 += Sb[(size_t)(i0 + l) * ldS + i] * xs[l];
                    s1 += Sb[(size_t)(i0 + l + 1) * ldS + i] * xs[l + 1];
                }
                if (l < ib) s0 += Sb[(size_t)(i0 + l) * ldS + i] * xs[l];
                z[i] -= s0 + s1;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += BK_T) out[(size_t)b * n + i] = scale * z[i];
}

// out[b][q] = a_q[b] . b_q[b], q < 3 (null pairs give 0); one workgroup per active instance.
struct BkDotArgs { const double *a[3]; const double *b[3]; };
__global__ __launch_bounds__(BK_T) void bk_dots_kernel(BkDotArgs d, int n, double *out, const int *active)
{
    __shared__ double red[3][BK_T / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (active && !active[b]) return;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = tid; i < n; i += BK_T)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (d.a[q]) acc[q] += d.a[q][(size_t)b * n + i] * d.b[q][(size_t)b * n + i];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double v = wave_sum(acc[q]);
        if ((tid & 63) == 0) red[q][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 3) out[(size_t)b * 4 + tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
}

// out[b] = a * x[b] + c * y[b] over vectors of length n (y may be null), active instances only
__global__ void bk_axpby_kernel(double a, const double *x, double c, const double *y, double *out, int n, const int *active)
{
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if ((active && !active[b]) || i >= n) return;
    const size_t e = (size_t)b * n + i;
    out[e] = a * x[e] + (y ? c * y[e] : 0.0);
}

// dst[b] = src[b] (count doubles per instance), active instances only
__global__ void bk_copy_kernel(double *dst, const double *src, size_t count, const int *active)
{
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t base = (size_t)b * count, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[base + i] = src[base + i];
}

// upper triangle of U[b] = that of H[b] + lam[b] I  (n x n, ld n), active instances only; the factorisation and the
// solves that follow never touch the strict lower triangle.  One column per blockIdx.x step.
__global__ void bk_shifted_copy_kernel(double *U, const double *H, int n, const double *lam, const int *active)
{
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t base = (size_t)b * n * n;
    const double l = lam[b];
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        const size_t col = base + (size_t)c * n;
        for (int r = threadIdx.x; r <= c; r += blockDim.x) U[col + r] = H[col + r] + (r == c ? l : 0.0);
    }
}

// out[b] = a[b] - c[b] / out[b] = -(a[b] + c[b]) over `count` doubles per instance, active instances only
__global__ void bk_sub_kernel(const double *a, const double *c, double *out, size_t count, const int *active)
{
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t base = (size_t)b * count, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[base + i] = a[base + i] - c[base + i];
}
__global__ void bk_neg_sum_kernel(const double *a, const double *c, double *out, size_t count, const int *active)
{
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t base = (size_t)b * count, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[base + i] = -(a[base + i] + c[base + i]);
}

// logGaussian value (src/gaussian.hpp:41-45) of every instance from z = S^-T (x - mu); g = -g in place.
__global__ __launch_bounds__(BK_T) void bk_log_gaussian_value_kernel(const double *z, const double *S, size_t strideS, int n, double *value, double *g,
                                                                     const int *active)
{
    __shared__ double red[2][BK_T / 64];
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const double *Sb = S + (size_t)b * strideS;
    double sq = 0.0, ls = 0.0;
    for (int i = threadIdx.x; i < n; i += BK_T) {
        const double zi = z[(size_t)b * n + i];
        sq += zi * zi;
        ls += log(fabs(Sb[(size_t)i * n + i]));
        g[(size_t)b * n + i] = -g[(size_t)b * n + i];
    }
    sq = wave_sum(sq); ls = wave_sum(ls);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = ls; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        const double c = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        value[b] = -0.5 * a - (double)n / 2 * log(2 * 3.14159265358979323846264338327950288) - c;
    }
}

// B(i, j) = A(n-1-i, n-1-j), or with `transposed` B(i, j) = A(n-1-j, n-1-i), per instance
__global__ __launch_bounds__(256) void bk_reverse_kernel(const double *A, int n, double *B, int transposed)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= n) return;
    const size_t o = (size_t)blockIdx.z * n * n;
    const int ri = n - 1 - i, rj = n - 1 - j;
    B[o + (size_t)j * n + i] = transposed ? A[o + (size_t)ri * n + rj] : A[o + (size_t)rj * n + ri];
}

struct BankWs {
    double *Hp, *Hl, *H, *W, *U, *Hcur;                                     // [B][n*n]
    double *x, *tmp, *gp, *gl, *g, *xcur, *gcur, *p, *t, *q, *xn, *Hpv;     // [B][n]
    double *vp, *vl, *V, *dots, *lam;                                       // [B] ([B][4] for dots)
    int *info, *active;                                                     // [B]
    void *tws;
};

size_t bank_ws_bytes(int n, int B)
{
    const size_t nn = (size_t)n * n;
    return 6 * Arena::pad((size_t)B * nn * 8) + 12 * Arena::pad((size_t)B * n * 8) + 8 * Arena::pad((size_t)B * 4 * 8) + trtri_batch_ws_bytes(n, B);
}

void bank_carve(void *ws, int n, int B, BankWs *w)
{
    Arena ar(ws);
    const size_t nn = (size_t)n * n;
    w->Hp = ar.take<double>(B * nn); w->Hl = ar.take<double>(B * nn); w->H = ar.take<double>(B * nn);
    w->W = ar.take<double>(B * nn); w->U = ar.take<double>(B * nn); w->Hcur = ar.take<double>(B * nn);
    double **v[12] = {&w->x, &w->tmp, &w->gp, &w->gl, &w->g, &w->xcur, &w->gcur, &w->p, &w->t, &w->q, &w->xn, &w->Hpv};
    for (double **q : v) *q = ar.take<double>((size_t)B * n);
    w->vp = ar.take<double>(4 * B); w->vl = ar.take<double>(4 * B); w->V = ar.take<double>(4 * B);
    w->dots = ar.take<double>(4 * B); w->lam = ar.take<double>(4 * B);
    w->info = reinterpret_cast<int *>(ar.take<double>(4 * B)); w->active = reinterpret_cast<int *>(ar.take<double>(4 * B));
    ar.take<double>(4 * B);
    w->tws = ar.take<char>(trtri_batch_ws_bytes(n, B));
}

int grid_for(size_t count) { return (int)std::min<size_t>((count + 1023) / 1024, 8192); }

}  // namespace

// One chunk of B filters, all pointers device except the per-instance result arrays.
static vslam_status iekf_bank_chunk(vslam_ctx *ctx, const CamK &k, double sigma, int n, int B, const double *mu, const double *S,
                                    const double *y, const uint8_t *vis, double *mu_post, double *S_post, int32_t *fmin_code,
                                    int32_t *iters, int32_t *status)
{
    const size_t nn = (size_t)n * n;
    const int L = (n - 12) / 3, nb = (n + 255) / 256;
    hipStream_t s = ctx->stream;
    VSLAM_TRY(ws_reserve(ctx, bank_ws_bytes(n, B)));
    BankWs w;
    bank_carve(ctx->ws, n, B, &w);
    const size_t trsv_lds = (size_t)(((n + 1) & ~1) + 64 * 65 + 64) * 8;
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&bk_trsv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trsv_lds));
    const double eps = std::numeric_limits<double>::epsilon(), sqrteps = std::sqrt(eps);

    std::vector<int> act(B, 1), m(B, 0), acc(B, 0), hinfo(B, 0), ret(B, 1), its(B, 50), stat(B, 0), done(B, 0);
    std::vector<double> f(B, 0.0), fn(B, 0.0), Delta(B, 1.0), lam(B, 0.0), hd((size_t)4 * B, 0.0), pn(B, 0.0);

    auto trsv = [&](const double *M, bool trans, const double *rhs, double *out, double scale, const int *mask) {
        hipLaunchKernelGGL(bk_trsv_kernel, dim3(B), dim3(BK_T), trsv_lds, s, M, n, nn, n, trans ? 1 : 0, rhs, out, scale, mask);
    };
    auto send_mask = [&](const std::vector<int> &h) -> vslam_status { return h2d(ctx, w.active, h.data(), (size_t)B * sizeof(int)); };
    auto any = [&](const std::vector<int> &h) { return std::any_of(h.begin(), h.end(), [](int v) { return v != 0; }); };
    // results of bk_dots_kernel and the factorisation flags of all instances
    auto dots = [&](const double *a0, const double *b0, const double *a1, const double *b1, const double *a2, const double *b2) -> vslam_status {
        BkDotArgs d;
        d.a[0] = a0; d.b[0] = b0; d.a[1] = a1; d.b[1] = b1; d.a[2] = a2; d.b[2] = b2;
        hipLaunchKernelGGL(bk_dots_kernel, dim3(B), dim3(BK_T), 0, s, d, n, w.dots, w.active);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(d2h(ctx, hd.data(), w.dots, (size_t)4 * B * 8));
        VSLAM_TRY(d2h(ctx, hinfo.data(), w.info, (size_t)B * sizeof(int)));
        return stream_sync(ctx);
    };
    // for the instances of the mask on the device: U^T U = Hcur + lam I, p = -(Hcur + lam I)^-1 gcur
    auto factor_and_step = [&]() -> vslam_status {
        VSLAM_TRY(h2d(ctx, w.lam, lam.data(), (size_t)B * 8));
        hipLaunchKernelGGL(bk_shifted_copy_kernel, dim3(std::min(n, 64), B), dim3(256), 0, s, w.U, w.Hcur, n, w.lam, w.active);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(bk_copy_kernel, dim3(1, B), dim3(256), 0, s, w.t, w.gcur, (size_t)n, w.active);
        VSLAM_TRY(potrf_upper_batch_dev(ctx, w.U, n, n, nn, B, w.info, w.active, /*zero_lower=*/false, w.t, (size_t)n));   // t = U^-T g comes with the factor
        trsv(w.U, false, w.t, w.p, -1.0, w.active);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        return VSLAM_OK;
    };
    bool prior_ready = false;
    // V, g, H of CostJointDensity (src/gaussian.hpp:435-467) at w.x for every instance; V to the host
    auto eval = [&](std::vector<double> &V) -> vslam_status {
        const int *a = w.active;
        hipLaunchKernelGGL(bk_sub_kernel, dim3(1, B), dim3(256), 0, s, w.x, mu, w.tmp, (size_t)n, a);
        trsv(S, true, w.tmp, w.tmp, 1.0, a);                              // z = S^-T (x - mu)
        trsv(S, false, w.tmp, w.gp, 1.0, a);                              // S^-1 z (negated below)
        hipLaunchKernelGGL(bk_log_gaussian_value_kernel, dim3(B), dim3(BK_T), 0, s, w.tmp, S, nn, n, w.vp, w.gp, a);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        if (!prior_ready) {                                               // the prior Hessian does not depend on x
            VSLAM_TRY(trtri_upper_batch_dev(ctx, S, n, n, nn, w.W, n, nn, B, w.tws));
            GemmDesc d;                                                   // Hp = -W W^T
            d.M = n; d.N = n; d.K = n; d.transA = false; d.transB = true; d.amode = A_UPPER; d.b_upper = true; d.alpha = -1.0;
            d.A = w.W; d.lda = n; d.strideA = nn; d.B = w.W; d.ldb = n; d.strideB = nn; d.C = w.Hp; d.ldc = n; d.strideC = nn; d.batch = B;
            VSLAM_TRY(gemm_f64(ctx, d));
            prior_ready = true;
        }
        VSLAM_TRY(launch_point_loglik(ctx, k, B, n, w.x, n, y, 2 * (size_t)L, vis, L, sigma, w.vl, w.gl, n, w.Hl, n, nn, nullptr, a));
        hipLaunchKernelGGL(bk_neg_sum_kernel, dim3(1, B), dim3(64), 0, s, w.vp, w.vl, w.V, (size_t)1, a);
        hipLaunchKernelGGL(bk_neg_sum_kernel, dim3(1, B), dim3(256), 0, s, w.gp, w.gl, w.g, (size_t)n, a);
        hipLaunchKernelGGL(bk_neg_sum_kernel, dim3(grid_for(nn), B), dim3(256), 0, s, w.Hp, w.Hl, w.H, nn, a);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(d2h(ctx, V.data(), w.V, (size_t)B * 8));
        return stream_sync(ctx);
    };
    // current iterate := the point just evaluated, for the instances of the mask on the device
    auto accept = [&]() {
        hipLaunchKernelGGL(bk_copy_kernel, dim3(1, B), dim3(256), 0, s, w.xcur, w.x, (size_t)n, w.active);
        hipLaunchKernelGGL(bk_copy_kernel, dim3(1, B), dim3(256), 0, s, w.gcur, w.g, (size_t)n, w.active);
        hipLaunchKernelGGL(bk_copy_kernel, dim3(grid_for(nn), B), dim3(256), 0, s, w.Hcur, w.H, nn, w.active);
    };
    auto drop_failed = [&](std::vector<int> &mask) {           // a non-positive pivot: the instance leaves the bank
        for (int b = 0; b < B; ++b)
            if (hinfo[b] != 0 && !done[b]) { stat[b] = 1; done[b] = 1; act[b] = 0; mask[b] = 0; }
    };

    VSLAM_HIP_TRY(ctx, hipMemsetAsync(w.info, 0, (size_t)B * sizeof(int), s));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(w.x, mu, (size_t)B * n * 8, hipMemcpyDeviceToDevice, s));
    VSLAM_TRY(send_mask(act));
    VSLAM_TRY(eval(f));                                                   // src/fmin.hpp:39
    accept();
    const bool trace = getenv("VSLAM_BANK_TRACE") != nullptr;      // diagnostic: per-iteration wall time, filters taking part, multiplier steps
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now_ms();
    for (int it = 0; it < 50 && any(act); ++it) {
        int trace_steps = 0, trace_active = (int)std::count_if(act.begin(), act.end(), [](int v) { return v != 0; });
        // ---- trust-region subproblem, src/fmin.cpp:40-131 with l1 > 0 ----
        std::fill(lam.begin(), lam.end(), 0.0);
        VSLAM_TRY(send_mask(act));
        VSLAM_TRY(factor_and_step());
        VSLAM_TRY(dots(w.p, w.p, nullptr, nullptr, nullptr, nullptr));
        drop_failed(act);
        for (int b = 0; b < B; ++b) m[b] = act[b] && std::sqrt(hd[(size_t)4 * b]) > Delta[b];
        for (int kk = 0; kk < 20 && any(m); ++kk) {
            VSLAM_TRY(send_mask(m));
            trsv(w.U, true, w.p, w.q, 1.0, w.active);
            VSLAM_TRY(dots(w.p, w.p, w.q, w.q, nullptr, nullptr));
            drop_failed(m);
            std::vector<int> stop(B, 0);
            for (int b = 0; b < B; ++b)
                if (m[b]) {
                    const double ppn = std::sqrt(hd[(size_t)4 * b]), dpp = hd[(size_t)4 * b + 1];
                    const double ff = 1 / Delta[b] - 1 / ppn, gg = -dpp / (ppn * ppn * ppn);
                    const double lam_new = std::max(0.0, lam[b] - ff / gg);
                    // see trs_multiplier_settled() in linalg.hpp
                    stop[b] = std::fabs(ff) < sqrteps || trs_multiplier_settled(ff, Delta[b]);
                    if (lam_new == lam[b]) { stop[b] = 1; m[b] = 0; }       // same multiplier: nothing to refactor
                    lam[b] = lam_new;
                }
            if (!any(m)) break;
            ++trace_steps;
            VSLAM_TRY(send_mask(m));
            VSLAM_TRY(factor_and_step());
            for (int b = 0; b < B; ++b)
                if (stop[b]) m[b] = 0;
        }
        // ---- src/fmin.hpp:54-105 ----
        VSLAM_TRY(send_mask(act));
        GemmDesc gv;                                   // Hpv = Hcur * p
        gv.M = n; gv.N = 1; gv.K = n; gv.transA = false; gv.transB = false;
        gv.A = w.Hcur; gv.lda = n; gv.strideA = nn; gv.B = w.p; gv.ldb = n; gv.strideB = n; gv.C = w.Hpv; gv.ldc = n; gv.strideC = n; gv.batch = B; gv.active = w.active;
        VSLAM_TRY(gemm_f64(ctx, gv));
        hipLaunchKernelGGL(bk_axpby_kernel, dim3(nb, B), dim3(256), 0, s, 1.0, w.xcur, 1.0, w.p, w.xn, n, w.active);
        VSLAM_TRY(dots(w.p, w.gcur, w.p, w.Hpv, w.p, w.p));
        drop_failed(act);
        for (int b = 0; b < B; ++b)
            if (act[b] && std::fabs(-hd[(size_t)4 * b]) < 2 * eps) { ret[b] = 0; its[b] = it; done[b] = 1; act[b] = 0; }   // src/fmin.hpp:61-67
        if (!any(act)) break;
        VSLAM_TRY(send_mask(act));
        hipLaunchKernelGGL(bk_copy_kernel, dim3(1, B), dim3(256), 0, s, w.x, w.xn, (size_t)n, w.active);
        VSLAM_TRY(eval(fn));
        for (int b = 0; b < B; ++b) {
            acc[b] = 0;
            if (!act[b]) continue;
            const double pg = hd[(size_t)4 * b], pHp = hd[(size_t)4 * b + 1], pnb = std::sqrt(hd[(size_t)4 * b + 2]);
            const double dm = -pg - 0.5 * pHp, rho = (f[b] - fn[b]) / dm;
            if (rho < 0.1) Delta[b] = 0.25 * pnb;
            else if (rho > 0.75 && pnb > 0.8 * Delta[b]) Delta[b] = 2.0 * Delta[b];
            if (rho >= 0.001) { f[b] = fn[b]; acc[b] = 1; }
        }
        if (any(acc)) {
            VSLAM_TRY(send_mask(acc));
            accept();
        }
        if (trace) {
            const double t = now_ms();
            std::fprintf(stderr, "bank iteration %d: %d filters, %d multiplier steps, %d accepted, %.2f ms\n", it, trace_active, trace_steps,
                         (int)std::count_if(acc.begin(), acc.end(), [](int v) { return v != 0; }), t - t_prev);
            t_prev = t;
        }
    }
    // ---- posterior square roots: S = M^-1 with M M^T = H at the final iterate (see newton.hip) ----
    std::vector<int> ok(B);
    for (int b = 0; b < B; ++b) ok[b] = stat[b] == 0;
    VSLAM_TRY(send_mask(ok));
    hipLaunchKernelGGL(bk_reverse_kernel, dim3(nb, n, B), dim3(256), 0, s, w.Hcur, n, w.U, 0);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(potrf_upper_batch_dev(ctx, w.U, n, n, nn, B, w.info, w.active));
    hipLaunchKernelGGL(bk_reverse_kernel, dim3(nb, n, B), dim3(256), 0, s, w.U, n, w.Hl, 1);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(trtri_upper_batch_dev(ctx, w.Hl, n, n, nn, S_post, n, nn, B, w.tws));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(mu_post, w.xcur, (size_t)B * n * 8, hipMemcpyDeviceToDevice, s));
    VSLAM_TRY(d2h(ctx, hinfo.data(), w.info, (size_t)B * sizeof(int)));
    VSLAM_TRY(stream_finish(ctx));
    for (int b = 0; b < B; ++b) {
        if (hinfo[b] != 0) stat[b] = 1;
        if (status) status[b] = stat[b];
        if (fmin_code) fmin_code[b] = ret[b];
        if (iters) iters[b] = its[b];
    }
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_measurement_update_iekf_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mu,
                                                                const double *S, int n, int batch, const double *y,
                                                                const uint8_t *visible, double sigma, double *mu_post,
                                                                double *S_post, int32_t *fmin_code, int32_t *iters, int32_t *status)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !y || !visible || !mu_post || !S_post) return fail(ctx, VSLAM_ERR_ARG, "measurement_update_iekf_batch: null pointer");
    if (n < 15 || (n - 12) % 3 != 0 || batch < 1) return fail(ctx, VSLAM_ERR_SHAPE, "measurement_update_iekf_batch: n=%d is not 12+3L or batch=%d", n, batch);
    if (!(sigma > 0.0)) return fail(ctx, VSLAM_ERR_ARG, "measurement_update_iekf_batch: sigma must be positive");
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    const int L = (n - 12) / 3;
    // instances per pass: the work matrices of one pass stay under 16 GiB
    size_t chunk = ((size_t)16 << 30) / (7 * nn * 8);
    chunk = std::max<size_t>(1, std::min<size_t>(chunk, 16384));
    for (int b0 = 0; b0 < batch; b0 += (int)chunk) {
        const int B = std::min<int>((int)chunk, batch - b0);
        VSLAM_TRY(iekf_bank_chunk(ctx, k, sigma, n, B, mu + (size_t)b0 * n, S + (size_t)b0 * nn, y + (size_t)b0 * 2 * L,
                                  visible + (size_t)b0 * L, mu_post + (size_t)b0 * n, S_post + (size_t)b0 * nn,
                                  fmin_code ? fmin_code + b0 : nullptr, iters ? iters + b0 : nullptr, status ? status + b0 : nullptr));
    }
    return VSLAM_OK;
}
