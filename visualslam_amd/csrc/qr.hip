// Blocked Householder QR (R only) for the square-root-covariance
// re-triangularisations: Eigen::HouseholderQR + triangularView<Upper> at
// reference src/gaussian.hpp:121-124 (affineTransform), :515-516
// (measurementUpdateIEKF) and src/SLAM.cpp:810-819 (landmark removal).
//
// Reflector convention = Eigen makeHouseholder / LAPACK dlarfg:
//   beta = -sign(alpha) * ||x||  (alpha >= 0 -> beta < 0),  v = x / (alpha - beta),
//   tau = (beta - alpha) / beta;  a column whose tail is exactly zero keeps its
//   value (tau = 0).  So R carries the same signs as the reference's.
//
// Per 32-column panel:
//   qr_panel_kernel   one workgroup per matrix factors the m x 32 panel
//                     left-looking: column c receives the previous reflectors
//                     in compact-WY form (two passes over the panel's finished
//                     columns, one 32-value workgroup reduction each), then its
//                     own reflector is formed; T (32x32) is built on the fly.
//   gemm (MFMA)       W = V^T A2, split over K so the skinny product fills the chip
//   qr_apply_t_kernel W2 = T^T * sum(W slabs)
//   gemm (MFMA)       A2 -= V W2
// V is read straight out of the factored panel (unit diagonal and zeros above
// it are synthesised by the GEMM's operand loader).
#include "linalg.hpp"

namespace vslam {

constexpr int NB = 32;
constexpr int PT = 512;          // panel kernel threads (8 waves): 256 VGPRs per lane available

__device__ __forceinline__ double wave_sum_f64(double v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sums acc[0..NB) over the workgroup; result in out[0..NB) (LDS), valid after the call.
__device__ __forceinline__ void block_reduce_nb(double (&acc)[NB + 1], int count, double *red /*[PT/64][NB+1]*/, double *out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < NB + 1; ++k)
        if (k < count) acc[k] = wave_sum_f64(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NB + 1; ++k)
            if (k < count) red[wave * (NB + 1) + k] = acc[k];
    }
    __syncthreads();
    if (tid < count) {
        double s = 0.0;
        for (int w = 0; w < PT / 64; ++w) s += red[w * (NB + 1) + tid];
        out[tid] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(PT) void qr_panel_kernel(double *Abase, int lda, size_t strideA, int mp, int jb,
                                                      double *Tbase /*[batch][NB*NB]*/)
{
    double *A = Abase + (size_t)blockIdx.x * strideA;       // panel origin (row j0, column j0)
    double *Tg = Tbase + (size_t)blockIdx.x * NB * NB;
    __shared__ double T[NB][NB + 1];
    __shared__ double red[(PT / 64) * (NB + 1)];
    __shared__ double z[NB + 1], y[NB], u[NB + 1];
    __shared__ double sc[4];                                  // tau, scale, beta
    const int tid = threadIdx.x;

    for (int i = tid; i < NB * (NB + 1); i += PT) (&T[0][0])[i] = 0.0;
    __syncthreads();

    for (int c = 0; c < jb; ++c) {
        double *col = A + (size_t)c * lda;
        double acc[NB + 1];
        // ---- pass A: z_k = v_k^T a, k < c ----
        if (c > 0) {
#pragma unroll
            for (int k = 0; k < NB + 1; ++k) acc[k] = 0.0;
            for (int i = tid; i < mp; i += PT) {
                const double ai = col[i];
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    if (k < c) {
                        double coef = (i > k) ? A[(size_t)k * lda + i] : ((i == k) ? 1.0 : 0.0);
                        acc[k] += coef * ai;
                    }
            }
            block_reduce_nb(acc, c, red, z);
            // y = T^T z  (T upper triangular c x c)
            if (tid < c) {
                double s = 0.0;
                for (int l = 0; l <= tid; ++l) s += T[l][tid] * z[l];
                y[tid] = s;
            }
            __syncthreads();
        }
        // ---- pass C: a' = a - V y; tail norm and u_k = sum_{i>c} V[i,k] a'[i] ----
#pragma unroll
        for (int k = 0; k < NB + 1; ++k) acc[k] = 0.0;
        for (int i = tid; i < mp; i += PT) {
            double ai = col[i];
            double vrow[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k < c) {
                    vrow[k] = (i > k) ? A[(size_t)k * lda + i] : ((i == k) ? 1.0 : 0.0);
                    ai -= vrow[k] * y[k];
                }
            if (c > 0) col[i] = ai;
            if (i > c) {
                acc[NB] += ai * ai;
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    if (k < c) acc[k] += vrow[k] * ai;
            }
        }
        // reduce u_0..u_{c-1} and the tail norm (kept in slot c of the reduced vector)
        {
            double tmp = acc[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k == c) acc[k] = tmp;
        }
        block_reduce_nb(acc, c + 1, red, u);
        // ---- reflector ----
        if (tid == 0) {
            const double alpha = col[c], tail = u[c];
            double beta, tau, scale;
            if (tail <= 2.2250738585072014e-308) { beta = alpha; tau = 0.0; scale = 0.0; }
            else {
                beta = sqrt(alpha * alpha + tail);
                if (alpha >= 0.0) beta = -beta;
                scale = 1.0 / (alpha - beta);
                tau = (beta - alpha) / beta;
            }
            sc[0] = tau; sc[1] = scale; sc[2] = beta;
        }
        __syncthreads();
        const double tau = sc[0], scale = sc[1], beta = sc[2];
        // T[0:c, c] = -tau * T[0:c,0:c] * (V^T v_c),  (V^T v_c)_k = V[c,k] + scale * u_k
        if (tid < c) z[tid] = A[(size_t)tid * lda + c] + scale * u[tid];
        __syncthreads();
        if (tid < c) {
            double s = 0.0;
            for (int k = tid; k < c; ++k) s += T[tid][k] * z[k];
            T[tid][c] = -tau * s;
        }
        if (tid == 0) { T[c][c] = tau; col[c] = beta; }
        for (int i = c + 1 + tid; i < mp; i += PT) col[i] *= scale;     // v tail (zeros when tau == 0)
        __syncthreads();
    }
    for (int i = tid; i < NB * NB; i += PT) Tg[i] = T[i % NB][i / NB];   // column-major, ld NB
}

// W2[0:jb, n] = T^T * sum_s Wpart[s][0:jb, n]
__global__ __launch_bounds__(256) void qr_apply_t_kernel(const double *Tbase, const double *Wpart, size_t strideSplit,
                                                         size_t strideBatchW, int ksplit, int jb, int n2, int ldw,
                                                         double *W2, size_t strideBatchW2)
{
    __shared__ double T[NB * NB];
    const int b = blockIdx.y;
    const double *Tg = Tbase + (size_t)b * NB * NB;
    for (int i = threadIdx.x; i < NB * NB; i += 256) T[i] = Tg[i];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n2) return;
    double w[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) w[k] = 0.0;
    for (int s = 0; s < ksplit; ++s) {
        const double *src = Wpart + (size_t)b * strideBatchW + (size_t)s * strideSplit + (size_t)n * ldw;
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (k < jb) w[k] += src[k];
    }
    double *dst = W2 + (size_t)b * strideBatchW2 + (size_t)n * ldw;
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (k < jb) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < NB; ++l)
                if (l <= k) s += T[l + k * NB] * w[l];       // (T^T)[k][l] = T[l][k]
            dst[k] = s;
        }
}

__global__ void zero_strict_lower_kernel(double *Abase, int rows, int cols, int lda, size_t strideA)
{
    double *A = Abase + (size_t)blockIdx.z * strideA;
    const int c = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows && r > c) A[(size_t)c * lda + r] = 0.0;
}

static int pick_ksplit(int mp, int n2, int batch)
{
    // skinny product (NB x n2, K = mp): spread K so that ~512 workgroups are in flight
    const int tilesN = (n2 + 63) / 64;
    int want = 512 / (tilesN * batch > 0 ? tilesN * batch : 1);
    int maxs = (mp + 255) / 256;                       // at least 256 rows of K per slab
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    return want;
}

static int max_ksplit(int rows, int batch)
{
    int a = 512 / batch, b = (rows + 255) / 256;
    int m = a < b ? a : b;
    return m < 1 ? 1 : (m > 64 ? 64 : m);
}

size_t qr_ws_bytes(int rows, int cols, int batch)
{
    size_t t = Arena::pad((size_t)batch * NB * NB * 8);
    size_t wpart = Arena::pad((size_t)batch * max_ksplit(rows, batch) * NB * (size_t)cols * 8);
    size_t w2 = Arena::pad((size_t)batch * NB * (size_t)cols * 8);
    return t + wpart + w2;
}

vslam_status qr_upper_dev(vslam_ctx *ctx, double *A, int rows, int cols, int lda, size_t strideA, int batch, void *ws)
{
    if (rows < 1 || cols < 1 || lda < rows) return fail(ctx, VSLAM_ERR_SHAPE, "qr: bad shape %dx%d ld %d", rows, cols, lda);
    if (batch < 1 || batch > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "qr: batch=%d out of range", batch);
    Arena ar(ws);
    double *T = ar.take<double>((size_t)batch * NB * NB);
    double *Wpart = ar.take<double>((size_t)batch * max_ksplit(rows, batch) * NB * (size_t)cols);
    double *W2 = ar.take<double>((size_t)batch * NB * (size_t)cols);
    const int kmax = rows < cols ? rows : cols;
    hipStream_t s = ctx->stream;
    for (int j0 = 0; j0 < kmax; j0 += NB) {
        const int jb = (kmax - j0) < NB ? (kmax - j0) : NB;
        const int mp = rows - j0;
        double *P = A + (size_t)j0 * lda + j0;
        hipLaunchKernelGGL(qr_panel_kernel, dim3(batch), dim3(PT), 0, s, P, lda, strideA, mp, jb, T);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        const int n2 = cols - j0 - jb;
        if (n2 <= 0) continue;
        double *A2 = P + (size_t)jb * lda;
        const int ksplit = pick_ksplit(mp, n2, batch);
        const size_t slab = (size_t)NB * n2;
        GemmDesc g1;                                   // Wpart[s] = V^T A2 over K slab s
        g1.M = jb; g1.N = n2; g1.K = mp; g1.transA = true; g1.transB = false; g1.amode = A_UNIT_LOWER;
        g1.A = P; g1.lda = lda; g1.strideA = strideA;
        g1.B = A2; g1.ldb = lda; g1.strideB = strideA;
        g1.C = Wpart; g1.ldc = NB; g1.strideC = slab * ksplit; g1.batch = batch;
        g1.ksplit = ksplit; g1.strideSplit = slab;
        if (ksplit == 1) { g1.C = W2; g1.strideC = slab; }   // no slabs: result goes straight to W2's buffer
        VSLAM_TRY(gemm_f64(ctx, g1));
        // W2 = T^T * sum slabs (ksplit == 1: in place on W2's buffer)
        const double *src = (ksplit == 1) ? W2 : Wpart;
        hipLaunchKernelGGL(qr_apply_t_kernel, dim3((n2 + 255) / 256, batch), dim3(256), 0, s, T, src, slab,
                           (ksplit == 1) ? slab : slab * ksplit, ksplit, jb, n2, NB, W2, slab);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        GemmDesc g3;                                   // A2 -= V W2
        g3.M = mp; g3.N = n2; g3.K = jb; g3.transA = false; g3.transB = false; g3.amode = A_UNIT_LOWER;
        g3.alpha = -1.0; g3.beta = 1.0;
        g3.A = P; g3.lda = lda; g3.strideA = strideA;
        g3.B = W2; g3.ldb = NB; g3.strideB = slab;
        g3.C = A2; g3.ldc = lda; g3.strideC = strideA; g3.batch = batch;
        VSLAM_TRY(gemm_f64(ctx, g3));
    }
    dim3 grid((rows + 255) / 256, cols, batch);
    hipLaunchKernelGGL(zero_strict_lower_kernel, grid, dim3(256), 0, s, A, rows, cols, lda, strideA);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_qr_upper_batch_dev(vslam_ctx *ctx, double *A, int rows, int cols, int lda,
                                                 size_t stride_A, int batch)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!A) return fail(ctx, VSLAM_ERR_ARG, "qr_upper: null matrix");
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (rows < 1 || cols < 1 || lda < rows) return fail(ctx, VSLAM_ERR_SHAPE, "qr_upper: bad shape %dx%d ld %d", rows, cols, lda);
    VSLAM_TRY(ws_reserve(ctx, qr_ws_bytes(rows, cols, batch)));
    return qr_upper_dev(ctx, A, rows, cols, lda, stride_A, batch, ctx->ws);
}

extern "C" vslam_status vslam_qr_upper(vslam_ctx *ctx, double *A, int rows, int cols, int lda)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!A) return fail(ctx, VSLAM_ERR_ARG, "qr_upper: null matrix");
    if (rows < 1 || cols < 1 || lda < rows) return fail(ctx, VSLAM_ERR_SHAPE, "qr_upper: bad shape %dx%d ld %d", rows, cols, lda);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t mat = Arena::pad((size_t)rows * cols * 8);
    VSLAM_TRY(ws_reserve(ctx, mat + qr_ws_bytes(rows, cols, 1)));
    Arena ar(ctx->ws);
    double *dA = ar.take<double>((size_t)rows * cols);
    void *qws = ar.take<char>(qr_ws_bytes(rows, cols, 1));
    hipStream_t s = ctx->stream;
    VSLAM_HIP_TRY(ctx, hipMemcpy2DAsync(dA, (size_t)rows * 8, A, (size_t)lda * 8, (size_t)rows * 8, cols, hipMemcpyHostToDevice, s));
    VSLAM_TRY(qr_upper_dev(ctx, dA, rows, cols, rows, 0, 1, qws));
    VSLAM_HIP_TRY(ctx, hipMemcpy2DAsync(A, (size_t)lda * 8, dA, (size_t)rows * 8, (size_t)rows * 8, cols, hipMemcpyDeviceToHost, s));
    VSLAM_HIP_TRY(ctx, hipStreamSynchronize(s));
    return VSLAM_OK;
}

// Test hook (not part of the public header): C = alpha*op(A)*op(B) + beta*C with host pointers.
extern "C" vslam_status vslam_debug_gemm(vslam_ctx *ctx, int M, int N, int K, int transA, int transB, int amode, int b_upper,
                                         double alpha, const double *A, int lda, int rowsA, int colsA,
                                         const double *B, int ldb, int rowsB, int colsB, double beta, double *C, int ldc, int ksplit)
{
    if (!ctx || !A || !B || !C) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t ba = (size_t)lda * colsA * 8, bb = (size_t)ldb * colsB * 8, bc = (size_t)ldc * N * 8;
    const int ks = ksplit < 1 ? 1 : ksplit;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(ba) + Arena::pad(bb) + Arena::pad(bc) * (ks + 1)));
    Arena ar(ctx->ws);
    double *dA = ar.take<double>((size_t)lda * colsA), *dB = ar.take<double>((size_t)ldb * colsB), *dC = ar.take<double>((size_t)ldc * N);
    double *dS = ar.take<double>((size_t)ldc * N * ks);
    hipStream_t s = ctx->stream;
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dA, A, ba, hipMemcpyHostToDevice, s));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dB, B, bb, hipMemcpyHostToDevice, s));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dC, C, bc, hipMemcpyHostToDevice, s));
    GemmDesc g;
    g.M = M; g.N = N; g.K = K; g.transA = transA != 0; g.transB = transB != 0; g.amode = (AMode)amode; g.b_upper = b_upper != 0;
    g.alpha = alpha; g.beta = beta; g.A = dA; g.lda = lda; g.B = dB; g.ldb = ldb; g.C = dC; g.ldc = ldc;
    if (ks > 1) { g.C = dS; g.ksplit = ks; g.strideSplit = (size_t)ldc * N; }
    VSLAM_TRY(gemm_f64(ctx, g));
    if (ks > 1) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(C, dS, bc * ks, hipMemcpyDeviceToHost, s));   // caller sums the slabs
    else VSLAM_HIP_TRY(ctx, hipMemcpyAsync(C, dC, bc, hipMemcpyDeviceToHost, s));
    VSLAM_HIP_TRY(ctx, hipStreamSynchronize(s));
    return VSLAM_OK;
}
