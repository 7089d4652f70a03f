// Triangular kernels for the square-root Gaussian algebra: blocked inverse of
// an upper-triangular factor and single-vector triangular solves.
//
// logGaussian (reference src/gaussian.hpp:30-74) needs z = S^-T (x - mu),
// g = -S^-1 z and H = -S^-1 S^-T; conditionGaussianOnMarginal
// (src/gaussian.cpp:34) needs one S1^-T solve.  The O(n^3) part, the prior
// Hessian, is formed as W = S^-1 (blocked, MFMA GEMMs) followed by -W W^T
// (one MFMA GEMM) instead of the reference's two n-column triangular solves;
// the O(n^2) vector solves run as blocked substitution in one workgroup.
#include "linalg.hpp"

namespace vslam {

constexpr int TBK = 64;          // diagonal block of the triangular inverse
constexpr int SB = 32;           // block of the vector solves
constexpr int ST = 512;          // threads of the vector-solve workgroup

// D = inverse of the upper-triangular jb x jb block at S (ld ldS); D written with ld TBK, zeros below.
__global__ __launch_bounds__(64) void trtri_diag_kernel(const double *S, int ldS, int jb, double *D)
{
    __shared__ double U[TBK][TBK + 1];
    const int c = threadIdx.x;
    for (int r = 0; r < jb; ++r) U[r][c] = (c < jb && r <= c) ? S[(size_t)c * ldS + r] : 0.0;
    __syncthreads();
    // column c of the inverse by back substitution on e_c: x_r = (delta_rc - sum_{k>r} U[r][k] x_k) / U[r][r]
    double x[TBK];
#pragma unroll
    for (int r = 0; r < TBK; ++r) x[r] = 0.0;
    if (c < jb) {
#pragma unroll
        for (int r = TBK - 1; r >= 0; --r) {
            if (r <= c && r < jb) {
                double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < TBK; ++k)
                    if (k > r && k <= c) s -= U[r][k] * x[k];
                x[r] = s / U[r][r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TBK; ++r)
        if (c < TBK) D[(size_t)c * TBK + r] = (c < jb && r <= c) ? x[r] : 0.0;
}

__global__ void copy_block_kernel(const double *src, int lds, double *dst, int ldd, int rows, int cols, double scale)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (r < rows && c < cols) dst[(size_t)c * ldd + r] = scale * src[(size_t)c * lds + r];
}

__global__ void zero_lower_block_kernel(double *W, int ldW, int n)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (r < n && r > c) W[(size_t)c * ldW + r] = 0.0;
}

size_t trtri_ws_bytes(int n) { return Arena::pad((size_t)TBK * TBK * 8) + Arena::pad((size_t)n * TBK * 8); }

vslam_status trtri_upper_dev(vslam_ctx *ctx, const double *S, int n, int ldS, double *W, int ldW, void *ws)
{
    Arena ar(ws);
    double *D = ar.take<double>((size_t)TBK * TBK);
    double *Tmp = ar.take<double>((size_t)n * TBK);
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(zero_lower_block_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, W, ldW, n);
    for (int j0 = 0; j0 < n; j0 += TBK) {
        const int jb = (n - j0) < TBK ? (n - j0) : TBK;
        hipLaunchKernelGGL(trtri_diag_kernel, dim3(1), dim3(64), 0, s, S + (size_t)j0 * ldS + j0, ldS, jb, D);
        hipLaunchKernelGGL(copy_block_kernel, dim3(1, jb), dim3(64), 0, s, D, TBK, W + (size_t)j0 * ldW + j0, ldW, jb, jb, 1.0);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        if (j0 == 0) continue;
        GemmDesc g1;                       // Tmp = W[0:j0,0:j0] * S[0:j0, J]
        g1.M = j0; g1.N = jb; g1.K = j0; g1.transA = false; g1.transB = false; g1.amode = A_UPPER;
        g1.A = W; g1.lda = ldW; g1.B = S + (size_t)j0 * ldS; g1.ldb = ldS; g1.C = Tmp; g1.ldc = n;
        VSLAM_TRY(gemm_f64(ctx, g1));
        GemmDesc g2;                       // W[0:j0, J] = -Tmp * D
        g2.M = j0; g2.N = jb; g2.K = jb; g2.transA = false; g2.transB = false; g2.b_upper = true; g2.alpha = -1.0;
        g2.A = Tmp; g2.lda = n; g2.B = D; g2.ldb = TBK; g2.C = W + (size_t)j0 * ldW; g2.ldc = ldW;
        VSLAM_TRY(gemm_f64(ctx, g2));
    }
    return VSLAM_OK;
}

// Blocked substitution, one workgroup.  z may alias b.
__global__ __launch_bounds__(ST) void trsv_upper_kernel(const double *S, int n, int ldS, int trans, const double *b, double *z)
{
    __shared__ double red[(ST / 64) * SB];
    __shared__ double rhs[SB], sol[SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += ST) z[i] = b[i];
    __syncthreads();
    if (trans) {
        // S^T z = b: forward.  Block I: r = b_I - S[0:i0, I]^T z[0:i0], then the 32x32 lower solve.
        for (int i0 = 0; i0 < n; i0 += SB) {
            const int ib = min(SB, n - i0);
            double acc[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) acc[j] = 0.0;
            for (int k = tid; k < i0; k += ST) {
                const double zk = z[k];
#pragma unroll
                for (int j = 0; j < SB; ++j)
                    if (j < ib) acc[j] += S[(size_t)(i0 + j) * ldS + k] * zk;
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                double v = acc[j];
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane == 0) red[wave * SB + j] = v;
            }
            __syncthreads();
            if (tid < ib) {
                double sum = 0.0;
                for (int w = 0; w < ST / 64; ++w) sum += red[w * SB + tid];
                rhs[tid] = z[i0 + tid] - sum;
            }
            __syncthreads();
            if (tid == 0) {
                for (int j = 0; j < ib; ++j) {
                    double v = rhs[j];
                    for (int l = 0; l < j; ++l) v -= S[(size_t)(i0 + j) * ldS + i0 + l] * sol[l];
                    sol[j] = v / S[(size_t)(i0 + j) * ldS + i0 + j];
                }
            }
            __syncthreads();
            if (tid < ib) z[i0 + tid] = sol[tid];
            __syncthreads();
        }
    } else {
        // S z = b: backward.  Solve the diagonal block, then z[0:i0] -= S[0:i0, I] z_I.
        const int nblk = (n + SB - 1) / SB;
        for (int blk = nblk - 1; blk >= 0; --blk) {
            const int i0 = blk * SB, ib = min(SB, n - i0);
            if (tid == 0) {
                for (int j = ib - 1; j >= 0; --j) {
                    double v = z[i0 + j];
                    for (int l = j + 1; l < ib; ++l) v -= S[(size_t)(i0 + l) * ldS + i0 + j] * sol[l];
                    sol[j] = v / S[(size_t)(i0 + j) * ldS + i0 + j];
                }
            }
            __syncthreads();
            if (tid < ib) z[i0 + tid] = sol[tid];
            for (int r = tid; r < i0; r += ST) {
                double v = z[r];
#pragma unroll
                for (int j = 0; j < SB; ++j)
                    if (j < ib) v -= S[(size_t)(i0 + j) * ldS + r] * sol[j];
                z[r] = v;
            }
            __syncthreads();
        }
    }
}

vslam_status trsv_upper_dev(vslam_ctx *ctx, const double *S, int n, int ldS, bool trans, const double *b, double *z)
{
    hipLaunchKernelGGL(trsv_upper_kernel, dim3(1), dim3(ST), 0, ctx->stream, S, n, ldS, trans ? 1 : 0, b, z);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam
