// Landmark bookkeeping on the square-root factor: removeBadLandmarks (reference src/SLAM.cpp:796-820)
// drops the three columns of a landmark from S and re-triangularises the n x (n-3) matrix by
// Householder QR (same reflector signs as the reference's Eigen::HouseholderQR).
//
// An upper-triangular S (every factor the filter produces) takes the structured kernels below: without
// columns c0..c0+2 the matrix is upper triangular up to column c0 and has three sub-diagonals from there
// on, so column j's reflector only touches rows j..j+3 -- the same reflectors the dense QR computes,
// O(n^2) work instead of O(n^3).  Any other S goes through the dense QR of qr.hip.
#include "linalg.hpp"

#include <cstdlib>

namespace vslam {

constexpr int RM_W = 16;             // panel width
constexpr int RM_H = RM_W + 3;       // active rows of a panel: its diagonal block and three rows below
constexpr int RM_T = 512;

// A (n x (n-3), ld lda) = upper triangle of S without columns c0..c0+2, zeros below the three sub-diagonals.
__global__ void drop_columns_upper_kernel(const double *S, int ldS, int n, int c0, double *A, int lda)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    const int cs = c < c0 ? c : c + 3;
    if (r < n) A[(size_t)c * lda + r] = (r <= cs) ? S[(size_t)cs * ldS + r] : 0.0;
}

// Number of non-zero entries strictly below the diagonal of S.
__global__ __launch_bounds__(256) void count_lower_kernel(const double *S, int ldS, int n, int *count)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < n && r > c && S[(size_t)c * ldS + r] != 0.0) atomicAdd(count, 1);
}

// One wave factors a panel: lane c < w holds rows j0..j0+RM_H-1 of column j0+c in registers; at step k
// lane k's tail norm, pivot and reflector entries are broadcast with v_readlane and every later lane
// updates its own column.  LAPACK / Eigen reflector convention.  Afterwards lane c holds R on and above
// the diagonal and its reflector (three entries) below it.
__device__ __forceinline__ void rm_factor_wave(double (&pc)[RM_H], int lane, int w, double &tau_own)
{
    tau_own = 0.0;
#pragma unroll
    for (int k = 0; k < RM_W; ++k) {
        if (k < w) {
            const double t2 = pc[k + 1] * pc[k + 1] + pc[k + 2] * pc[k + 2] + pc[k + 3] * pc[k + 3];
            const double tail = readlane_f64(t2, k), alpha = readlane_f64(pc[k], k);
            double beta, tau, scale;
            if (tail <= 2.2250738585072014e-308) { beta = alpha; tau = 0.0; scale = 0.0; }
            else {
                beta = sqrt(alpha * alpha + tail);
                if (alpha >= 0.0) beta = -beta;
                scale = 1.0 / (alpha - beta);
                tau = (beta - alpha) / beta;
            }
            const double v1 = readlane_f64(pc[k + 1], k) * scale, v2 = readlane_f64(pc[k + 2], k) * scale,
                         v3 = readlane_f64(pc[k + 3], k) * scale;
            const double d = tau * (pc[k] + v1 * pc[k + 1] + v2 * pc[k + 2] + v3 * pc[k + 3]);
            const bool later = lane > k, own = lane == k;
            pc[k] = own ? beta : (later ? pc[k] - d : pc[k]);
            pc[k + 1] = own ? v1 : (later ? pc[k + 1] - d * v1 : pc[k + 1]);
            pc[k + 2] = own ? v2 : (later ? pc[k + 2] - d * v2 : pc[k + 2]);
            pc[k + 3] = own ? v3 : (later ? pc[k + 3] - d * v3 : pc[k + 3]);
            if (own) tau_own = tau;
        }
    }
}

// Panel columns in (lanes >= w and rows past the matrix read zeros) ...
__device__ __forceinline__ void rm_load_panel_column(const double *A, int lda, int n, int j0, int w, int lane, double (&pc)[RM_H])
{
    const bool in = lane < w;
    const double *col = A + (size_t)(j0 + (in ? lane : 0)) * lda + j0;
#pragma unroll
    for (int i = 0; i < RM_H; ++i) pc[i] = (in && j0 + i < n) ? col[i] : 0.0;
}

// ... and out: R back to the matrix with zeros below the diagonal, reflectors (Pm[d][k] = entry d+1 of
// reflector k) and taus to LDS.
__device__ __forceinline__ void rm_store_panel_column(const double (&pc)[RM_H], double tau_own, int lane, int w, double *A, int lda,
                                                      int n, int j0, double (*Pm)[RM_W], double *taus)
{
    if (lane < RM_W) {
        double v[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < RM_W; ++k)
            if (k == lane && lane < w) { v[0] = pc[k + 1]; v[1] = pc[k + 2]; v[2] = pc[k + 3]; }
        Pm[0][lane] = v[0]; Pm[1][lane] = v[1]; Pm[2][lane] = v[2];
        taus[lane] = lane < w ? tau_own : 0.0;
    }
    if (lane < w) {
        double *col = A + (size_t)(j0 + lane) * lda + j0;
#pragma unroll
        for (int i = 0; i < RM_H; ++i)
            if (j0 + i < n) col[i] = (i <= lane) ? pc[i] : 0.0;
    }
}

// The panel's reflectors, in order, on columns [c_begin, m), one column per thread.
__device__ __forceinline__ void rm_apply(const double (*Pm)[RM_W], const double *taus, double *A, int lda, int n, int m, int j0,
                                         int c_begin, int slot, int nslots)
{
    for (int c = c_begin + slot; c < m; c += nslots) {
        double *col = A + (size_t)c * lda + j0;
        double a[RM_H];
#pragma unroll
        for (int i = 0; i < RM_H; ++i) a[i] = (j0 + i < n) ? col[i] : 0.0;
#pragma unroll
        for (int k = 0; k < RM_W; ++k) {
            const double v1 = Pm[0][k], v2 = Pm[1][k], v3 = Pm[2][k];
            const double d = taus[k] * (a[k] + v1 * a[k + 1] + v2 * a[k + 2] + v3 * a[k + 3]);
            a[k] -= d; a[k + 1] -= d * v1; a[k + 2] -= d * v2; a[k + 3] -= d * v3;
        }
#pragma unroll
        for (int i = 0; i < RM_H; ++i)
            if (j0 + i < n) col[i] = a[i];
    }
}

// Whole factorisation by one workgroup (n up to about a thousand: the later columns fit a few per thread).
__device__ __forceinline__ void rm_qr_workgroup(double *A, int lda, int n, int m, int c0)
{
    __shared__ double P[3][RM_W];
    __shared__ double taus[RM_W];
    const int tid = threadIdx.x;
    for (int j0 = c0; j0 < m; j0 += RM_W) {
        const int w = min(RM_W, m - j0);
        if (tid < 64) {
            double pc[RM_H], tau_own;
            rm_load_panel_column(A, lda, n, j0, w, tid, pc);
            rm_factor_wave(pc, tid, w, tau_own);
            rm_store_panel_column(pc, tau_own, tid, w, A, lda, n, j0, P, taus);
        }
        __syncthreads();
        rm_apply(P, taus, A, lda, n, m, j0, j0 + w, tid, RM_T);
        __syncthreads();
    }
}

__global__ __launch_bounds__(RM_T) void rm_qr_kernel(double *A, int lda, int n, int m, int c0) { rm_qr_workgroup(A, lda, n, m, c0); }

// ---- a bank of filters: instance b drops landmark j_each[b]; an index outside 0..L-1 leaves the instance's outputs
// untouched and is reported in status[b] ----
__device__ __forceinline__ bool rm_bank_index(const int *j_each, int b, int n, int &c0)
{
    const int j = j_each[b];
    c0 = 12 + 3 * j;
    return j >= 0 && j < (n - 12) / 3;
}

__global__ void rm_bank_rows_kernel(const double *mu, int n, const int *j_each, double *out, int *status)
{
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    int c0;
    const bool ok = rm_bank_index(j_each, b, n, c0);
    if (i == 0 && status) status[b] = ok ? 0 : 1;
    if (ok && i < n - 3) out[(size_t)b * (n - 3) + i] = mu[(size_t)b * n + (i < c0 ? i : i + 3)];
}

__global__ void rm_bank_columns_kernel(const double *S, int n, const int *j_each, double *A)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    int c0;
    if (!rm_bank_index(j_each, b, n, c0)) return;
    const int cs = c < c0 ? c : c + 3;
    const size_t o = (size_t)b * n * n;
    if (r < n) A[o + (size_t)c * n + r] = (r <= cs) ? S[o + (size_t)cs * n + r] : 0.0;
}

__global__ __launch_bounds__(RM_T) void rm_bank_qr_kernel(double *A, int n, const int *j_each)
{
    const int b = blockIdx.x;
    int c0;
    if (!rm_bank_index(j_each, b, n, c0)) return;
    rm_qr_workgroup(A + (size_t)b * n * n, n, n, n - 3, c0);
}

__global__ void rm_bank_copy_out_kernel(const double *A, int n, const int *j_each, double *S_out)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z, m = n - 3;
    int c0;
    if (!rm_bank_index(j_each, b, n, c0)) return;
    if (r < m) S_out[(size_t)b * m * m + (size_t)c * m + r] = (r <= c) ? A[(size_t)b * n * n + (size_t)c * n + r] : 0.0;
}

// Large n: one launch pair per panel (reflectors through Vg: 3 * RM_W entries and RM_W taus).
__global__ __launch_bounds__(64) void rm_qr_factor_kernel(double *A, int lda, int n, int m, int j0, double *Vg)
{
    __shared__ double P[3][RM_W];
    __shared__ double taus[RM_W];
    const int tid = threadIdx.x, w = min(RM_W, m - j0);
    double pc[RM_H], tau_own;
    rm_load_panel_column(A, lda, n, j0, w, tid, pc);
    rm_factor_wave(pc, tid, w, tau_own);
    rm_store_panel_column(pc, tau_own, tid, w, A, lda, n, j0, P, taus);
    __syncthreads();
    if (tid < 3 * RM_W) Vg[tid] = (&P[0][0])[tid];
    if (tid < RM_W) Vg[3 * RM_W + tid] = taus[tid];
}

__global__ __launch_bounds__(256) void rm_qr_apply_kernel(double *A, int lda, int n, int m, int j0, const double *Vg)
{
    __shared__ double P[3][RM_W];
    __shared__ double taus[RM_W];
    const int tid = threadIdx.x;
    if (tid < 3 * RM_W) (&P[0][0])[tid] = Vg[tid];
    if (tid < RM_W) taus[tid] = Vg[3 * RM_W + tid];
    __syncthreads();
    rm_apply(P, taus, A, lda, n, m, j0, j0 + min(RM_W, m - j0), blockIdx.x * 256 + tid, gridDim.x * 256);
}

// A (n x m, m = n - 3, from drop_columns_upper_kernel) -> R in its leading m x m block.  Vg: 4 * RM_W doubles.
static vslam_status remove_qr_upper_dev(vslam_ctx *ctx, double *A, int lda, int n, int c0, double *Vg)
{
    const int m = n - 3;
    hipStream_t s = ctx->stream;
    if (n <= 1024) hipLaunchKernelGGL(rm_qr_kernel, dim3(1), dim3(RM_T), 0, s, A, lda, n, m, c0);
    else
        for (int j0 = c0; j0 < m; j0 += RM_W) {
            hipLaunchKernelGGL(rm_qr_factor_kernel, dim3(1), dim3(64), 0, s, A, lda, n, m, j0, Vg);
            const int later = m - j0 - RM_W;
            if (later > 0) hipLaunchKernelGGL(rm_qr_apply_kernel, dim3((later + 255) / 256), dim3(256), 0, s, A, lda, n, m, j0, Vg);
        }
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

// A (n x (n-3), ld lda) = S without columns c0..c0+2.
__global__ void drop_columns_kernel(const double *S, int ldS, int n, int c0, double *A, int lda)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < n) A[(size_t)c * lda + r] = S[(size_t)(c < c0 ? c : c + 3) * ldS + r];
}

__global__ void drop_rows_kernel(const double *mu, int n, int r0, double *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n - 3) out[i] = mu[i < r0 ? i : i + 3];
}

// leading m x m block of A (upper triangle, zeros below) to S_out
__global__ void rm_copy_out_kernel(const double *A, int lda, int m, double *S_out, int ldo)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < m) S_out[(size_t)c * ldo + r] = (r <= c) ? A[(size_t)c * lda + r] : 0.0;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_remove_landmark_dev(vslam_ctx *ctx, const double *mu, const double *S, int n, int ldS, int j,
                                                  double *mu_out, double *S_out, int ldSout, int s_is_upper)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "remove_landmark: null pointer");
    if (n < 15 || (n - 12) % 3 != 0 || ldS < n) return fail(ctx, VSLAM_ERR_SHAPE, "remove_landmark: n=%d is not 12+3L (L >= 1) or ldS < n", n);
    const int L = (n - 12) / 3, m = n - 3, c0 = 12 + 3 * j;
    if (j < 0 || j >= L) return fail(ctx, VSLAM_ERR_ARG, "remove_landmark: landmark %d of %d", j, L);
    if (ldSout < m) return fail(ctx, VSLAM_ERR_SHAPE, "remove_landmark: ldSout=%d < %d", ldSout, m);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(nn * 8) + Arena::pad(4 * RM_W * 8) + (s_is_upper ? 0 : qr_ws_bytes(n, m, 1))));
    Arena ar(ctx->ws);
    double *dA = ar.take<double>(nn), *Vg = ar.take<double>(4 * RM_W);
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(drop_rows_kernel, dim3((m + 255) / 256), dim3(256), 0, s, mu, n, c0, mu_out);
    if (s_is_upper) {
        hipLaunchKernelGGL(drop_columns_upper_kernel, dim3((n + 255) / 256, m), dim3(256), 0, s, S, ldS, n, c0, dA, n);
        VSLAM_TRY(remove_qr_upper_dev(ctx, dA, n, n, c0, Vg));
    } else {
        void *qws = ar.take<char>(qr_ws_bytes(n, m, 1));
        hipLaunchKernelGGL(drop_columns_kernel, dim3((n + 255) / 256, m), dim3(256), 0, s, S, ldS, n, c0, dA, n);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(qr_upper_dev(ctx, dA, n, m, n, 0, 1, qws));
    }
    hipLaunchKernelGGL(rm_copy_out_kernel, dim3((m + 255) / 256, m), dim3(256), 0, s, dA, n, m, S_out, ldSout);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

extern "C" vslam_status vslam_remove_landmark_batch_dev(vslam_ctx *ctx, const double *mu, const double *S, int n, int batch, const int32_t *j_each,
                                                        double *mu_out, double *S_out, int32_t *status)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !j_each || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "remove_landmark: null pointer");
    if (n < 15 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "remove_landmark: n=%d is not 12+3L (L >= 1)", n);
    if (batch < 1 || batch > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "remove_landmark: batch=%d out of range", batch);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int m = n - 3;
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(nn * 8 * (size_t)batch)));
    Arena ar(ctx->ws);
    double *dA = ar.take<double>(nn * (size_t)batch);
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(rm_bank_rows_kernel, dim3((m + 255) / 256, batch), dim3(256), 0, s, mu, n, j_each, mu_out, status);
    hipLaunchKernelGGL(rm_bank_columns_kernel, dim3((n + 255) / 256, m, batch), dim3(256), 0, s, S, n, j_each, dA);
    hipLaunchKernelGGL(rm_bank_qr_kernel, dim3(batch), dim3(RM_T), 0, s, dA, n, j_each);
    hipLaunchKernelGGL(rm_bank_copy_out_kernel, dim3((m + 255) / 256, m, batch), dim3(256), 0, s, dA, n, j_each, S_out);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

extern "C" vslam_status vslam_remove_landmark(vslam_ctx *ctx, const double *mu, const double *S, int n, int ldS, int j,
                                              double *mu_out, double *S_out, int ldSout)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "remove_landmark: null pointer");
    if (n < 15 || (n - 12) % 3 != 0 || ldS < n) return fail(ctx, VSLAM_ERR_SHAPE, "remove_landmark: n=%d is not 12+3L (L >= 1) or ldS < n", n);
    const int L = (n - 12) / 3, m = n - 3;
    if (j < 0 || j >= L) return fail(ctx, VSLAM_ERR_ARG, "remove_landmark: landmark %d of %d", j, L);
    if (ldSout < m) return fail(ctx, VSLAM_ERR_SHAPE, "remove_landmark: ldSout=%d < %d", ldSout, m);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(ws_reserve(ctx, 2 * Arena::pad(n * 8) + 2 * Arena::pad(nn * 8) + Arena::pad(64) + Arena::pad(4 * RM_W * 8) + qr_ws_bytes(n, m, 1)));
    Arena ar(ctx->ws);
    double *dmu = ar.take<double>(n), *dmo = ar.take<double>(n), *dS = ar.take<double>(nn), *dA = ar.take<double>(nn);
    int *dcount = ar.take<int>(1);
    double *Vg = ar.take<double>(4 * RM_W);
    void *qws = ar.take<char>(qr_ws_bytes(n, m, 1));
    hipStream_t s = ctx->stream;
    const int c0 = 12 + 3 * j;
    VSLAM_TRY(h2d(ctx, dmu, mu, n * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    // an upper-triangular factor (what the filter produces) takes the structured kernels
    int lower = 0;
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(dcount, 0, sizeof(int), s));
    hipLaunchKernelGGL(count_lower_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, dS, n, n, dcount);
    hipLaunchKernelGGL(drop_rows_kernel, dim3((m + 255) / 256), dim3(256), 0, s, dmu, n, c0, dmo);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(d2h(ctx, &lower, dcount, sizeof(int)));
    VSLAM_TRY(stream_sync(ctx));
    if (lower == 0 && !getenv("VSLAM_REMOVE_DENSE")) {
        hipLaunchKernelGGL(drop_columns_upper_kernel, dim3((n + 255) / 256, m), dim3(256), 0, s, dS, n, n, c0, dA, n);
        VSLAM_TRY(remove_qr_upper_dev(ctx, dA, n, n, c0, Vg));
    } else {
        hipLaunchKernelGGL(drop_columns_kernel, dim3((n + 255) / 256, m), dim3(256), 0, s, dS, n, n, c0, dA, n);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(qr_upper_dev(ctx, dA, n, m, n, 0, 1, qws));
    }
    VSLAM_TRY(d2h(ctx, mu_out, dmo, (size_t)m * 8));
    VSLAM_TRY(d2h_2d(ctx, S_out, (size_t)ldSout * 8, dA, (size_t)n * 8, (size_t)m * 8, m));
    return stream_finish(ctx);
}
