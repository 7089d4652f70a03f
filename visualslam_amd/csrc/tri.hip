// Triangular kernels for the square-root Gaussian algebra: blocked inverse of
// an upper-triangular factor and single-vector triangular solves.
//
// logGaussian (reference src/gaussian.hpp:30-74) needs z = S^-T (x - mu),
// g = -S^-1 z and H = -S^-1 S^-T; conditionGaussianOnMarginal
// (src/gaussian.cpp:34) needs one S1^-T solve.  The O(n^3) part, the prior
// Hessian, is formed as W = S^-1 (blocked, MFMA GEMMs) followed by -W W^T
// (one MFMA GEMM) instead of the reference's two n-column triangular solves;
// the O(n^2) vector solves run as blocked substitution, one launch per diagonal block.
#include "linalg.hpp"

#include <algorithm>

namespace vslam {

constexpr int TBK = 64;          // diagonal block of the triangular inverse

// Workgroup b (one wave) inverts the upper-triangular diagonal block b (64 x 64, the last one
// smaller) of S into the same block of W (zeros below the diagonal).  Lane c owns column
// c of the inverse and back-substitutes e_c:  x_r = (delta_rc - sum_{k>r} U[r][k] x_k) / U[r][r];
// both matrices live in LDS (U read as broadcasts, the inverse lane-contiguous).
__global__ __launch_bounds__(64) void trtri_diag_kernel(const double *S, int ldS, int n, double *W, int ldW, size_t strideS, size_t strideW)
{
    S += (size_t)blockIdx.y * strideS;                 // instance blockIdx.y of a batch
    W += (size_t)blockIdx.y * strideW;
    extern __shared__ double trtri_lds[];
    double *U = trtri_lds;                        // [TBK][TBK + 1]
    double *X = trtri_lds + TBK * (TBK + 1);      // [TBK][TBK], X[r][c]
    const int c = threadIdx.x, j0 = blockIdx.x * TBK, jb = min(TBK, n - j0);
    const double *Sb = S + (size_t)j0 * ldS + j0;
    double *D = W + (size_t)j0 * ldW + j0;
    for (int cb = 0; cb < TBK; cb += 16) {        // lane = row: coalesced column reads, 16 in flight
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cc = cb + j;
            v[j] = (c <= cc && cc < jb) ? Sb[(size_t)cc * ldS + c] : ((c == cc) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) U[c * (TBK + 1) + cb + j] = v[j];
    }
    __syncthreads();
    for (int r = TBK - 1; r >= 0; --r) {
        const double num = lds_dot_sub((r == c) ? 1.0 : 0.0, U + r * (TBK + 1), 1, X + c, TBK, r + 1, TBK);
        X[r * TBK + c] = (r <= c) ? num / U[r * (TBK + 1) + r] : 0.0;
    }
    __syncthreads();
    // lane = row of the inverse: a column of W is written as one contiguous run
    for (int cc = 0; cc < jb; ++cc)
        if (c < jb) D[(size_t)cc * ldW + c] = X[c * TBK + cc];
}

__global__ void zero_lower_block_kernel(double *W, int ldW, int n, size_t strideW)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (r < n && r > c) W[(size_t)blockIdx.z * strideW + (size_t)c * ldW + r] = 0.0;
}

// Scratch of the triangular inverse: the products S12 W22 of one level (see below).
static size_t trtri_tmp_doubles(int n, int batch)
{
    size_t need = 1;
    for (int b = TBK; b < n; b *= 2) {
        const int full = n / (2 * b), i0 = full * 2 * b;
        const int r = (i0 + b < n) ? n - i0 - b : 0;                     // the ragged pair behind the full ones
        const size_t lvl = batch == 1 ? std::max((size_t)full * b * b, (size_t)b * r) : (size_t)batch * b * (full > 0 ? b : r);
        need = std::max(need, lvl);
    }
    return need;
}
size_t trtri_ws_bytes(int n) { return Arena::pad(trtri_tmp_doubles(n, 1) * 8); }
size_t trtri_batch_ws_bytes(int n, int batch) { return Arena::pad(trtri_tmp_doubles(n, batch) * 8); }

// W = S^-1 for `batch` upper-triangular matrices (element strides strideS / strideW), bottom up: every 64 x 64
// diagonal block is inverted by one launch, then neighbouring blocks of width b = 64, 128, 256, ... are joined,
//     [S11 S12; 0 S22]^-1 = [W11  -W11 S12 W22; 0  W22],
// all pairs of a level in two GEMM launches (one matrix: the pairs are the batch dimension of the GEMM; a bank of
// matrices: one pair at a time with the matrices as the batch dimension).  The products are large and few instead
// of a thin one per block column: 2 log2(n / 64) launches on the matrix cores.
vslam_status trtri_upper_batch_dev(vslam_ctx *ctx, const double *S, int n, int ldS, size_t strideS, double *W, int ldW, size_t strideW,
                                   int batch, void *ws)
{
    Arena ar(ws);
    const int nblk = (n + TBK - 1) / TBK;
    double *Tmp = ar.take<double>(trtri_tmp_doubles(n, batch));
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(zero_lower_block_kernel, dim3((n + 255) / 256, n, batch), dim3(256), 0, s, W, ldW, n, strideW);
    const size_t dlds = (size_t)(TBK * (TBK + 1) + TBK * TBK) * 8;
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&trtri_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
    hipLaunchKernelGGL(trtri_diag_kernel, dim3(nblk, batch), dim3(64), dlds, s, S, ldS, n, W, ldW, strideS, strideW);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    // pairs [i0, i0+b) and [i0+b, i0+b+r) of one level: `count` of them `pair_stride` blocks apart (or the bank)
    auto join = [&](int i0, int b, int r, int count) -> vslam_status {
        const bool bank = batch > 1;
        const size_t ps = (size_t)2 * b;
        GemmDesc g1;                       // Tmp = S12 W22
        g1.M = b; g1.N = r; g1.K = r; g1.transA = false; g1.transB = false; g1.b_upper = true;
        g1.A = S + (size_t)(i0 + b) * ldS + i0; g1.lda = ldS; g1.strideA = bank ? strideS : ps * ldS + ps;
        g1.B = W + (size_t)(i0 + b) * ldW + i0 + b; g1.ldb = ldW; g1.strideB = bank ? strideW : ps * ldW + ps;
        g1.C = Tmp; g1.ldc = b; g1.strideC = (size_t)b * r; g1.batch = bank ? batch : count;
        VSLAM_TRY(gemm_f64(ctx, g1));
        GemmDesc g2;                       // W12 = -W11 Tmp
        g2.M = b; g2.N = r; g2.K = b; g2.transA = false; g2.transB = false; g2.amode = A_UPPER; g2.alpha = -1.0;
        g2.A = W + (size_t)i0 * ldW + i0; g2.lda = ldW; g2.strideA = bank ? strideW : ps * ldW + ps;
        g2.B = Tmp; g2.ldb = b; g2.strideB = (size_t)b * r;
        g2.C = W + (size_t)(i0 + b) * ldW + i0; g2.ldc = ldW; g2.strideC = bank ? strideW : ps * ldW + ps; g2.batch = bank ? batch : count;
        return gemm_f64(ctx, g2);
    };
    for (int b = TBK; b < n; b *= 2) {
        const int full = n / (2 * b), i0r = full * 2 * b;
        if (batch == 1) {
            if (full > 0) VSLAM_TRY(join(0, b, b, full));
        } else {
            for (int p = 0; p < full; ++p) VSLAM_TRY(join(p * 2 * b, b, b, 1));
        }
        if (i0r + b < n) VSLAM_TRY(join(i0r, b, n - i0r - b, 1));
    }
    return VSLAM_OK;
}

vslam_status trtri_upper_dev(vslam_ctx *ctx, const double *S, int n, int ldS, double *W, int ldW, void *ws)
{
    return trtri_upper_batch_dev(ctx, S, n, ldS, 0, W, ldW, 0, 1, ws);
}

// Vector solves with the upper-triangular S, one launch per 64-wide diagonal block: every
// workgroup solves the block redundantly (one wave, the running right-hand side passed between
// lanes), workgroup 0 stores that part of the solution, and each workgroup then removes the block's
// contribution from its own 256 entries of the remaining right-hand side.  z holds the right-hand
// side on entry and the solution on return.
//   trans = 0:  S z = b,   blocks from the last to the first, remaining entries are rows above;
//   trans = 1:  S^T z = b, blocks from the first to the last, remaining entries are later columns.
__global__ __launch_bounds__(256) void trsv_step_kernel(const double *S, int n, int ldS, int trans, int i0, int ib, double *z)
{
    __shared__ double U[TBK][TBK + 1];
    __shared__ double sol[TBK];
    const int tid = threadIdx.x;
    {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int idx = tid + 256 * j, r = idx % TBK, c = idx / TBK;
            v[j] = (r <= c && c < ib) ? S[(size_t)(i0 + c) * ldS + i0 + r] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int idx = tid + 256 * j;
            U[idx % TBK][idx / TBK] = v[j];
        }
    }
    __syncthreads();
    if (tid < 64) {
        double v = (tid < ib) ? z[i0 + tid] : 0.0;
        // substitution with the running right-hand side in the lanes; the pivot lane's value and the
        // reciprocal of its diagonal entry are broadcast with v_readlane (the chain per step is
        // readlane -> multiply -> multiply-add)
        const double rinv = 1.0 / U[tid][tid];                      // rows past ib hold a zero diagonal: unused
        if (trans) {
            for (int k = 0; k < ib; ++k) {
                const double xk = readlane_f64(v, k) * readlane_f64(rinv, k);
                if (tid > k) v -= U[k][tid] * xk;
                else if (tid == k) v = xk;
            }
        } else {
            for (int k = ib - 1; k >= 0; --k) {
                const double xk = readlane_f64(v, k) * readlane_f64(rinv, k);
                if (tid < k) v -= U[tid][k] * xk;
                else if (tid == k) v = xk;
            }
        }
        sol[tid] = v;
        if (blockIdx.x == 0 && tid < ib) z[i0 + tid] = v;
    }
    __syncthreads();
    if (trans) {
        const int c = i0 + ib + blockIdx.x * 256 + tid;             // later entry: z_c -= S[I, c]^T sol
        if (c < n) {
            const double *col = S + (size_t)c * ldS + i0;
            double acc = 0.0;
            for (int j = 0; j < ib; ++j) acc += col[j] * sol[j];
            z[c] -= acc;
        }
    } else {
        const int r = blockIdx.x * 256 + tid;                        // earlier entry: z_r -= S[r, I] sol
        if (r < i0) {
            double acc = 0.0;
            for (int j = 0; j < ib; ++j) acc += S[(size_t)(i0 + j) * ldS + r] * sol[j];
            z[r] -= acc;
        }
    }
}

// The same solve in ONE launch for up to TC_MAX diagonal blocks: workgroup i owns the i-th block in
// dependency order (trans: first to last; otherwise last to first).  It adds up the contributions of the
// blocks before it as their solutions appear (coefficients are fetched ahead of the wait: they do not
// depend on it), solves its own 64 x 64 block in one wave and publishes the 64 entries.  Only the chain of
// diagonal solves is sequential; the off-diagonal work of all workgroups overlaps it.
// A solution entry is its own ready mark: `sol` is filled with a NaN of all-ones bits before the launch and a
// reader polls the entries it needs until none of them carries that pattern any more, so a link of the chain costs
// one trip through memory (the entries themselves), not a flag and then the data.  (A solution that is itself this
// NaN -- only a right-hand side holding the same pattern produces one -- ends in the bounded wait below.)
// All workgroups are resident together (TC_MAX <= 240 of 256 compute units), a wait that exceeds its bound raises
// *err and lets every wave run to the end.
constexpr int TC_MAX = 240;
static_assert(TC_MAX * TBK <= VSLAM_CHAIN_SOL_DOUBLES, "solution buffer of the context");
__global__ __launch_bounds__(256) void trsv_chain_kernel(const double *S, int n, int ldS, int trans, double *z, double *sol, int *err)
{
    __shared__ double U[TBK][TBK + 1];
    __shared__ double part[4][TBK];
    const int tid = threadIdx.x, c = tid & 63, p = tid >> 6;
    const int nblk = (n + TBK - 1) / TBK;
    const int w = trans ? (int)blockIdx.x : nblk - 1 - (int)blockIdx.x;
    const int i0 = w * TBK, ib = min(TBK, n - i0);
    {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int idx = tid + 256 * j, r = idx % TBK, cc = idx / TBK;
            v[j] = (r <= cc && cc < ib) ? S[(size_t)(i0 + cc) * ldS + i0 + r] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int idx = tid + 256 * j;
            U[idx % TBK][idx / TBK] = v[j];
        }
    }
    const double rhs = (tid < ib) ? z[i0 + tid] : 0.0;          // own right-hand side: fetched before any waiting
    // thread (c, p): own entry c, entries 16p..16p+15 of the other block
    double acc = 0.0;
    bool dead = false;
    for (int s = 0; s < (int)blockIdx.x; ++s) {
        const int v = trans ? s : nblk - 1 - s;
        const int v0 = v * TBK;                                   // the last block may be short: entries j >= n are skipped
        double coef[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = v0 + 16 * p + e;
            // trans: U(j, i0 + c)  (rows of the earlier block, own column);  otherwise U(i0 + c, j)
            coef[e] = (c < ib && j < n) ? (trans ? S[(size_t)(i0 + c) * ldS + j] : S[(size_t)j * ldS + i0 + c]) : 0.0;
        }
        double zv[16];
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = min(v0 + 16 * p + e, n - 1);
                zv[e] = __hip_atomic_load(&sol[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (__double_as_longlong(zv[e]) != -1ll);
            }
            if (ok || dead) break;
            if ((++spins & 255u) == 0) {
                if (spins > (1u << 22)) atomicExch(err, 1);
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) dead = true;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += coef[e] * zv[e];      // coefficients beyond the matrix are zero
    }
    part[p][c] = acc;
    __syncthreads();
    if (tid < 64) {
        double v = (tid < ib) ? rhs - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]) : 0.0;
        // substitution with the running right-hand side in the lanes; the pivot lane's value and the
        // reciprocal of its diagonal entry are broadcast with v_readlane (the chain per step is
        // readlane -> multiply -> multiply-add)
        const double rinv = 1.0 / U[tid][tid];                      // rows past ib hold a zero diagonal: unused
        if (trans) {
            for (int k = 0; k < ib; ++k) {
                const double xk = readlane_f64(v, k) * readlane_f64(rinv, k);
                if (tid > k) v -= U[k][tid] * xk;
                else if (tid == k) v = xk;
            }
        } else {
            for (int k = ib - 1; k >= 0; --k) {
                const double xk = readlane_f64(v, k) * readlane_f64(rinv, k);
                if (tid < k) v -= U[tid][k] * xk;
                else if (tid == k) v = xk;
            }
        }
        if (tid < ib) {
            __hip_atomic_store(&sol[i0 + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z[i0 + tid] = v;                                         // nobody else reads or writes this block of z
        }
    }
}

vslam_status trsv_upper_dev(vslam_ctx *ctx, const double *S, int n, int ldS, bool trans, const double *b, double *z)
{
    hipStream_t s = ctx->stream;
    if (z != b) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(z, b, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    const int nblk = (n + TBK - 1) / TBK;
    if (nblk > 1 && nblk <= TC_MAX) {
        VSLAM_HIP_TRY(ctx, hipMemsetAsync(ctx->chain_sol, 0xff, (size_t)n * 8, s));
        hipLaunchKernelGGL(trsv_chain_kernel, dim3(nblk), dim3(256), 0, s, S, n, ldS, trans ? 1 : 0, z, ctx->chain_sol, ctx->flags);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        return VSLAM_OK;
    }
    for (int step = 0; step < nblk; ++step) {
        const int blk = trans ? step : nblk - 1 - step;
        const int i0 = blk * TBK, ib = (n - i0) < TBK ? (n - i0) : TBK;
        const int rest = trans ? n - i0 - ib : i0;
        const int grid = rest > 0 ? (rest + 255) / 256 : 1;
        hipLaunchKernelGGL(trsv_step_kernel, dim3(grid), dim3(256), 0, s, S, n, ldS, trans ? 1 : 0, i0, ib, z);
    }
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam

// Test hook (not part of the public header): W = S^-1 for `batch` upper-triangular n x n matrices held one after
// the other on the host (column-major, ld n).
extern "C" vslam_status vslam_debug_trtri(vslam_ctx *ctx, const double *S, int n, int batch, double *W)
{
    using namespace vslam;
    if (!ctx || !S || !W || n < 1 || batch < 1) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n, bytes = nn * batch * 8;
    const size_t tws = batch == 1 ? trtri_ws_bytes(n) : trtri_batch_ws_bytes(n, batch);
    VSLAM_TRY(ws_reserve(ctx, 2 * Arena::pad(bytes) + tws));
    Arena ar(ctx->ws);
    double *dS = ar.take<double>(nn * batch), *dW = ar.take<double>(nn * batch);
    void *w = ar.take<char>(tws);
    VSLAM_TRY(h2d(ctx, dS, S, bytes));
    if (batch == 1) VSLAM_TRY(trtri_upper_dev(ctx, dS, n, n, dW, n, w));
    else VSLAM_TRY(trtri_upper_batch_dev(ctx, dS, n, n, nn, dW, n, nn, batch, w));
    VSLAM_TRY(d2h(ctx, W, dW, bytes));
    return stream_finish(ctx);
}
