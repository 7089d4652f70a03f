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
// smaller) of S into the same block of W (zeros below the diagonal).  Lane c owns column c of the inverse -- in
// registers -- and back-substitutes e_c:  x_r = (delta_rc - sum_{k>r} U[r][k] x_k) / U[r][r], the rows of U as
// broadcast reads from LDS, four partial sums per row, the pivots' reciprocals taken once; the loops are unrolled
// completely, so x stays in registers and a row costs its multiply-adds.  The result goes through the same LDS
// block transposed, so that a column of W is written as one contiguous run.
constexpr int TDS = TBK + 2;     // row stride of the block in LDS (even: rows are read as pairs)
__global__ __launch_bounds__(64) void trtri_diag_kernel(const double *S, int ldS, int n, double *W, int ldW, size_t strideS, size_t strideW)
{
    S += (size_t)blockIdx.y * strideS;                 // instance blockIdx.y of a batch
    W += (size_t)blockIdx.y * strideW;
    __shared__ __align__(16) double U[TBK * TDS];
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
        for (int j = 0; j < 16; ++j) U[c * TDS + cb + j] = v[j];
    }
    __syncthreads();
    const double rinv = 1.0 / U[c * TDS + c];          // lane r: reciprocal of pivot r
    double x[TBK];
#pragma unroll
    for (int r = TBK - 1; r >= 0; --r) {
        double s[4] = {(r == c) ? 1.0 : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = r + 1; k < TBK; ++k) s[k & 3] = fma(-U[r * TDS + k], x[k], s[k & 3]);
        x[r] = (r <= c) ? ((s[0] + s[1]) + (s[2] + s[3])) * readlane_f64(rinv, r) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TBK; ++r) U[r * TDS + c] = x[r];        // U is spent: it now holds the inverse, X[r][c]
    __syncthreads();
    // lane = row of the inverse: a column of W is written as one contiguous run
    for (int cc = 0; cc < jb; ++cc)
        if (c < jb) D[(size_t)cc * ldW + c] = U[c * TDS + cc];
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
                                   int batch, void *ws, bool zero_lower)
{
    Arena ar(ws);
    const int nblk = (n + TBK - 1) / TBK;
    double *Tmp = ar.take<double>(trtri_tmp_doubles(n, batch));
    hipStream_t s = ctx->stream;
    if (zero_lower) hipLaunchKernelGGL(zero_lower_block_kernel, dim3((n + 255) / 256, n, batch), dim3(256), 0, s, W, ldW, n, strideW);
    hipLaunchKernelGGL(trtri_diag_kernel, dim3(nblk, batch), dim3(64), 0, s, S, ldS, n, W, ldW, strideS, strideW);
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

vslam_status trtri_upper_dev(vslam_ctx *ctx, const double *S, int n, int ldS, double *W, int ldW, void *ws, bool zero_lower)
{
    return trtri_upper_batch_dev(ctx, S, n, ldS, 0, W, ldW, 0, 1, ws, zero_lower);
}

__global__ void scale_kernel(double *z, int n, double scale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] *= scale;
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

// The same solve in ONE launch for up to TC_MAX diagonal blocks.  Blocks are numbered in dependency order
// (position 0 first: the first block when trans, the last otherwise) and inside a block the entries are numbered
// the same way (slot 0 first), so both solves read as "entry k feeds every later entry".
//
// Workgroup 0 walks the whole chain and never waits for another workgroup on its critical path:
//   * its last wave substitutes through the 64 x 64 diagonal block held in LDS, already scaled by the reciprocal
//     pivots, so that a step of the chain is one lane broadcast and one multiply-add; the same step feeds the
//     coupling block towards the next position (a second, independent multiply-add), which takes the newest
//     solution block off the critical path altogether;
//   * its eight other waves meanwhile prepare the next position: the contributions of the positions 2 ... TC_LAG-1
//     back (coefficients fetched one link ahead, solution blocks from LDS), the scaled diagonal and coupling
//     blocks, and the sum over everything further back, which a helper workgroup supplies.
// Helper workgroup b accumulates, for position TC_LAG-1+b, the contributions of all positions at least TC_LAG
// back as their solutions appear and publishes the 64 sums.  Those travel through memory (two trips and a little
// arithmetic), which the TC_LAG-1 links in between cover.
// Published values are their own ready marks: `sol` (solutions, then sums: a stretch of the context's ring that no
// earlier launch has used since the ring was last filled) holds a NaN of all-ones bits and a reader polls until the
// pattern is gone (a value that is itself this NaN is published as the canonical quiet NaN, chain_publishable).  All workgroups are resident together (at most TC_MAX of 256 compute
// units, one workgroup each; a launch of more than 16 takes the GPU's turn, see trsv_upper_dev); a wait that
// exceeds its bound raises *err and lets every wave run to the end.
// rhs and z may be the same vector (an entry is read one link before it is written); z = scale * solution.
constexpr int TC_MAX = 240;
constexpr int TC_LAG = 3;
constexpr int TC_BANK_BLOCKS = 16;                            // a bank matrix has at most this many blocks (its solution stays in LDS)
constexpr int TC_T = 576;                                     // eight preparing waves and the chain wave
constexpr int TCS = TBK + 1;                                  // row stride of the staged blocks in LDS
constexpr int TC_LDS_DOUBLES = 4 * TBK * TCS + 2 * TBK + 16 * TBK + 4 * TBK + TC_LAG * TBK;
static_assert((size_t)2 * TC_MAX * TBK <= VSLAM_CHAIN_SOL_DOUBLES, "solution buffer of the context");

struct ChainGeo {
    const double *S; int n, ldS, trans, nblk;
    // global index of slot `slot` of position `pos`, or -1 outside the matrix
    __device__ __forceinline__ int at(int pos, int slot) const
    {
        if (pos < 0 || pos >= nblk) return -1;
        const int idx = trans ? pos * TBK + slot : (nblk - 1 - pos) * TBK + (TBK - 1 - slot);
        return idx < n ? idx : -1;
    }
    // The coefficient with which source slot k of position ps enters the equation of slot l of position pd, for a
    // thread whose lane runs along the memory rows (trans: lane = k and l = lu; otherwise lane = l and k = lu, lu
    // the same for the whole wave).  raw() fetches without branches from indices clamped into the matrix (so that
    // many fetches stay in flight together); live() tells whether the coefficient exists at all.
    __device__ __forceinline__ void where(int ps, int pd, int lane, int lu, int &row, int &col) const
    {
        const int pr = trans ? ps : pd, pc = trans ? pd : ps;            // positions of the memory row and column
        row = trans ? pr * TBK + lane : (nblk - 1 - pr) * TBK + (TBK - 1 - lane);
        col = trans ? pc * TBK + lu : (nblk - 1 - pc) * TBK + (TBK - 1 - lu);
    }
    template <bool CLAMP = true>
    __device__ __forceinline__ double raw(int ps, int pd, int lane, int lu) const
    {
        int row, col;
        where(ps, pd, lane, lu, row, col);
        if (CLAMP) { row = min(max(row, 0), n - 1); col = min(max(col, 0), n - 1); }
        return S[(size_t)col * ldS + row];
    }
    // True when the positions lo ... hi all exist and hold 64 entries each: nothing to clamp, nothing to mask.
    __device__ __forceinline__ bool whole(int lo, int hi) const
    {
        const bool ragged = (n % TBK) != 0;                              // the last block of the matrix is short
        return lo >= 0 && hi < nblk && !(ragged && (trans ? hi == nblk - 1 : lo == 0));
    }
    __device__ __forceinline__ bool live(int ps, int pd, int lane, int lu) const
    {
        int row, col;
        where(ps, pd, lane, lu, row, col);
        return ps >= 0 && pd < nblk && row < n && col < n;
    }
};

// A value as it is published: the one bit pattern that means "not ready" (a NaN of all-ones bits, which a right-hand
// side or a matrix holding that very NaN would propagate) is replaced by the canonical quiet NaN, so a reader sees a
// NaN arrive -- and the solve returns NaNs as the one-launch-per-block path does -- instead of polling to its bound.
__device__ __forceinline__ double chain_publishable(double v)
{
    return __double_as_longlong(v) == -1ll ? __longlong_as_double(0x7ff8000000000000ll) : v;
}

// Polls one published value; false once the launch has been declared dead.
__device__ __forceinline__ double chain_wait(const double *p, int *err, bool &dead)
{
    unsigned spins = 0;
    for (;;) {
        const double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__double_as_longlong(v) != -1ll || dead) return v;
        if ((++spins & 255u) == 0) {
            if (spins > (1u << 22)) atomicExch(err, 1);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) dead = true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Sum over the 64 lanes of each of v[0..7]; on return every lane holds the total of v[j] for
// j = 4 (lane bit 5) + 2 (lane bit 4) + (lane bit 3).  Ten exchanges instead of forty-eight.
__device__ __forceinline__ double chain_reduce8(const double (&v)[8], int lane)
{
    double a[4], b[2];
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = (h5 ? v[m + 4] : v[m]) + __shfl_xor(h5 ? v[m] : v[m + 4], 32);
#pragma unroll
    for (int m = 0; m < 2; ++m) b[m] = (h4 ? a[m + 2] : a[m]) + __shfl_xor(h4 ? a[m] : a[m + 2], 16);
    double c = (h3 ? b[1] : b[0]) + __shfl_xor(h3 ? b[0] : b[1], 8);
    c += __shfl_xor(c, 4);
    c += __shfl_xor(c, 2);
    c += __shfl_xor(c, 1);
    return c;
}

// Thread (lane, w, j) of the preparing waves and of the helpers handles the coefficient of source slot k in the
// equation of slot l, with the lanes along whichever of the two is contiguous in memory:
//   trans:  k = lane, l = w + 8 j        otherwise:  l = lane, k = w + 8 j
// (interleaved over the waves: the triangular diagonal block then gives every wave the same share).
//
// BANK: one workgroup per matrix of a bank (blockIdx.y; matrices strideS apart, right-hand sides and results n apart,
// `active` masks matrices out) and no helpers: all solution blocks stay in LDS, and what a helper would supply --
// the contributions of the positions three or more back -- the preparing waves add up themselves one link early
// (the blocks it needs are final by then) and carry, unreduced, in registers into the link that uses it.
template <bool trans, bool BANK>
__global__ __launch_bounds__(TC_T) void trsv_chain_kernel(const double *S, int n, int ldS, const double *rhs, double scale, double *z, double *sol, int *err,
                                                          size_t strideS, const int *active, int listed)
{
    extern __shared__ double tc_lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, w = wave & 7;
    if (BANK) {
        const int m = instance_of(active, listed, blockIdx.y);
        if (m < 0) return;
        S += (size_t)m * strideS; rhs += (size_t)m * n; z += (size_t)m * n;
    }
    ChainGeo g{S, n, ldS, trans, (n + TBK - 1) / TBK};
    double *sums = sol + (size_t)g.nblk * TBK;                   // helper sums, one block of 64 per position
    bool dead = false;
    const int jl = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);     // whose total chain_reduce8 leaves here

    if (!BANK && blockIdx.x > 0) {
        // helper: everything at least TC_LAG positions back, for position pos
        const int pos = TC_LAG - 1 + (int)blockIdx.x;
        double *part = tc_lds;                                   // [8][64]
        if (wave < 8) {
            double acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.0;
            for (int s = 0; s + TC_LAG <= pos; ++s) {
                double cf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) cf[j] = g.raw(s, pos, lane, w + 8 * j);
#pragma unroll
                for (int j = 0; j < 8; ++j) cf[j] = g.live(s, pos, lane, w + 8 * j) ? cf[j] : 0.0;
                if (trans) {
                    const double x = chain_wait(sol + (size_t)s * TBK + lane, err, dead);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += cf[j] * x;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += cf[j] * chain_wait(sol + (size_t)s * TBK + w + 8 * j, err, dead);
                }
            }
            if (trans) {
                const double tot = chain_reduce8(acc, lane);
                if ((lane & 7) == 0)
                    __hip_atomic_store(&sums[(size_t)pos * TBK + w + 8 * jl], chain_publishable(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                part[w * TBK + lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            }
        }
        if (!trans) {
            __syncthreads();
            if (tid < TBK) {
                double h = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) h += part[q * TBK + tid];
                __hip_atomic_store(&sums[(size_t)pos * TBK + tid], chain_publishable(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }

    double *Cs = tc_lds;                          // [2][64][TCS] diagonal block, scaled, row = source slot
    double *Bs = Cs + 2 * TBK * TCS;              // [2][64][TCS] coupling towards the next position, scaled
    double *rb = Bs + 2 * TBK * TCS;              // [2][64] trans: right-hand side of a position without its newest contribution
    double *part = rb + 2 * TBK;                  // [2][8][64] otherwise: the same in eight parts, negated
    double *rinvs = part + 2 * 8 * TBK;           // [4][64] reciprocal pivots
    double *xring = rinvs + 4 * TBK;              // [TC_LAG][64] the latest solution blocks (BANK: [nblk][64], all of them)
    auto xslot = [&](int pos) { return BANK ? pos : pos % TC_LAG; };

    double cst[8], bst[8], tcf[TC_LAG - 2][8], zr = 0.0, dgl = 1.0;
    const bool zlane = trans ? (lane & 7) == 0 : w == 0;          // lanes that finish a right-hand side entry
    const int zslot = trans ? w + 8 * jl : lane;
    auto diag_of = [&](int pos) {
        const int a = g.at(pos, lane);
        return (a >= 0) ? S[(size_t)a * ldS + a] : 1.0;
    };
    // Whole blocks (the rule; only the ends of the chain and a short last block differ) take a short way: one
    // address per block and thread, the eight coefficients a fixed stride apart, no clamping and no masks.
    const ptrdiff_t sj = trans ? 8 * (ptrdiff_t)ldS : -8 * (ptrdiff_t)ldS;       // from one of the thread's eight slots to the next
    auto block_ptr = [&](int ps, int pd) {
        int row, col;
        g.where(ps, pd, lane, w, row, col);
        return S + (size_t)col * ldS + row;
    };
    auto load_stage = [&](int pos) {
        if (g.whole(pos, pos + 1)) {
            const double *pc = block_ptr(pos, pos), *pb = block_ptr(pos, pos + 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) { cst[j] = pc[j * sj]; bst[j] = pb[j * sj]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                cst[j] = g.raw(pos, pos, lane, w + 8 * j);
                bst[j] = g.raw(pos, pos + 1, lane, w + 8 * j);
            }
        }
    };
    auto load_terms = [&](int pos) {
        if (g.whole(pos - (TC_LAG - 1), pos)) {
#pragma unroll
            for (int d = 2; d < TC_LAG; ++d) {
                const double *pt = block_ptr(pos - d, pos);
#pragma unroll
                for (int j = 0; j < 8; ++j) tcf[d - 2][j] = pt[j * sj];
            }
        } else {
#pragma unroll
            for (int d = 2; d < TC_LAG; ++d)
#pragma unroll
                for (int j = 0; j < 8; ++j) tcf[d - 2][j] = g.raw(pos - d, pos, lane, w + 8 * j);
        }
        const int c = g.at(pos, zslot);
        zr = (zlane && c >= 0) ? rhs[c] : 0.0;
    };
    auto do_stage = [&](int pos) {
        const int par = pos & 1;
        const double *ri = rinvs + (pos & 3) * TBK;
        if (g.whole(pos, pos + 1)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int lu = w + 8 * j, k = trans ? lane : lu, l = trans ? lu : lane;
                const double rinv = ri[k], cv = cst[j] * rinv;
                Cs[(par * TBK + k) * TCS + l] = (k < l) ? cv : 0.0;          // source slot before destination slot
                Bs[(par * TBK + k) * TCS + l] = bst[j] * rinv;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int lu = w + 8 * j, k = trans ? lane : lu, l = trans ? lu : lane;
                const double rinv = ri[k], cv = cst[j] * rinv, bv = bst[j] * rinv;
                Cs[(par * TBK + k) * TCS + l] = (k < l && g.live(pos, pos, lane, lu)) ? cv : 0.0;
                Bs[(par * TBK + k) * TCS + l] = g.live(pos, pos + 1, lane, lu) ? bv : 0.0;
            }
        }
    };
    double carry[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};    // BANK: the far contributions to the position of the next link
    auto do_terms = [&](int pos) -> double {
        double pr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pr[j] = BANK ? carry[j] : 0.0;
        if (g.whole(pos - (TC_LAG - 1), pos)) {
#pragma unroll
            for (int d = 2; d < TC_LAG; ++d) {
                const double *x = xring + xslot(pos - d) * TBK;
#pragma unroll
                for (int j = 0; j < 8; ++j) pr[j] = fma(tcf[d - 2][j], x[trans ? lane : w + 8 * j], pr[j]);
            }
        } else {
#pragma unroll
            for (int d = 2; d < TC_LAG; ++d) {
                const int src = pos - d;
                if (src >= 0) {
                    const double *x = xring + xslot(src) * TBK;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const double cf = g.live(src, pos, lane, w + 8 * j) ? tcf[d - 2][j] : 0.0;
                        pr[j] = fma(cf, x[trans ? lane : w + 8 * j], pr[j]);
                    }
                }
            }
        }
        return trans ? chain_reduce8(pr, lane) : ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7]));
    };
    // The helper's sum for a position is asked for when a link begins and looked at when it ends (the answer takes
    // a trip through memory); only if it has not been published by then does the wave wait.
    auto ask_sum = [&](int pos) {
        return (!BANK && zlane && pos >= TC_LAG) ? __hip_atomic_load(sums + (size_t)pos * TBK + zslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    };
    auto finish_terms = [&](int pos, double acc, double h, double zv) {
        if (!BANK && zlane && pos >= TC_LAG && __double_as_longlong(h) == -1ll) h = chain_wait(sums + (size_t)pos * TBK + zslot, err, dead);
        if (trans) {
            if (zlane) rb[(pos & 1) * TBK + zslot] = zv - h - acc;
        } else {
            if (zlane) acc -= zv - h;
            part[((pos & 1) * 8 + w) * TBK + lane] = acc;
        }
    };
    // A link whose blocks are all whole: the staged coefficients of position pos go to LDS and the registers are
    // refilled for position pos + 1 one by one, so that the fetches stream while the arithmetic runs.
    auto stage_and_load = [&](int pos) {
        const int par = pos & 1;
        const double *ri = rinvs + (pos & 3) * TBK;
        const double *pc = block_ptr(pos + 1, pos + 1), *pb = block_ptr(pos + 1, pos + 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lu = w + 8 * j, k = trans ? lane : lu, l = trans ? lu : lane;
            const double rinv = ri[k], cv = cst[j] * rinv;
            Cs[(par * TBK + k) * TCS + l] = (k < l) ? cv : 0.0;              // source slot before destination slot
            Bs[(par * TBK + k) * TCS + l] = bst[j] * rinv;
            cst[j] = (k < l) ? pc[j * sj] : 0.0;
            bst[j] = pb[j * sj];
        }
    };
    // BANK: carry <- contributions of the positions 0 ... pos - TC_LAG to position pos (all final when this is called)
    auto far_terms = [&](int pos) {
#pragma unroll
        for (int j = 0; j < 8; ++j) carry[j] = 0.0;
        for (int src = 0; src + TC_LAG <= pos; ++src) {
            const double *x = xring + src * TBK;
            double cf[8];
            if (g.whole(src, src) && g.whole(pos, pos)) {
                const double *pt = block_ptr(src, pos);
#pragma unroll
                for (int j = 0; j < 8; ++j) cf[j] = pt[j * sj];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) cf[j] = g.raw(src, pos, lane, w + 8 * j);
#pragma unroll
                for (int j = 0; j < 8; ++j) cf[j] = g.live(src, pos, lane, w + 8 * j) ? cf[j] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) carry[j] = fma(cf[j], x[trans ? lane : w + 8 * j], carry[j]);
        }
    };
    // 1 / d from the hardware estimate and two correction steps (a full division is five times the instructions)
    auto recip = [](double d) {
        double y = __builtin_amdgcn_rcp(d);
        y = fma(y, fma(-d, y, 1.0), y);
        return fma(y, fma(-d, y, 1.0), y);
    };
    // The preparing waves keep their fetches for the next link in flight across the barrier: it orders LDS only.
    auto lds_barrier = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };

    if (wave < 8) {
        if (wave == 7) {
            rinvs[0 * TBK + lane] = 1.0 / diag_of(0);
            rinvs[1 * TBK + lane] = 1.0 / diag_of(1);
            dgl = diag_of(2);
        }
        load_stage(0);
        load_terms(0);
    }
    lds_barrier();
    if (wave < 8) {
        finish_terms(0, do_terms(0), 0.0, zr);
        do_stage(0);
        load_stage(1);
        load_terms(1);
    }
    lds_barrier();
    double nacc = 0.0;                                            // chain wave: what the block just solved adds to the next one
    // Two loops, one per role, meeting at the same barriers: neither role carries the other's registers.
    if (wave == 8) {
        for (int i = 0; i < g.nblk; ++i) {
            const int par = i & 1;
            const double *C = Cs + par * TBK * TCS + lane, *B = Bs + par * TBK * TCS + lane;
            double v;
            if (trans) {
                v = rb[par * TBK + lane] - nacc;
            } else {
                const double *q = part + par * 8 * TBK + lane;
                v = -(((q[0] + q[TBK]) + (q[2 * TBK] + q[3 * TBK])) + ((q[4 * TBK] + q[5 * TBK]) + (q[6 * TBK] + q[7 * TBK]))) - nacc;
            }
            // the coefficients of a step are fetched from LDS eight steps ahead, into the registers the step just freed;
            // the coupling sums run in four independent parts beside the chain and are complete when it ends
            double c8[8], b8[8], na[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 8; ++q) { c8[q] = C[q * TCS]; b8[q] = B[q * TCS]; }
#pragma unroll
            for (int k = 0; k < TBK; ++k) {
                const double yk = readlane_f64(v, k);
                v = fma(-c8[k & 7], yk, v);
                na[k & 3] = fma(b8[k & 7], yk, na[k & 3]);
                if (k + 8 < TBK) { c8[k & 7] = C[(k + 8) * TCS]; b8[k & 7] = B[(k + 8) * TCS]; }
            }
            nacc = (na[0] + na[1]) + (na[2] + na[3]);
            asm volatile("" : "+v"(nacc));                        // finished here, not at its use behind the barrier
            const int c = g.at(i, lane);
            const double x = (c >= 0) ? v * rinvs[(i & 3) * TBK + lane] : 0.0;
            xring[xslot(i) * TBK + lane] = x;
            if (!BANK) __hip_atomic_store(&sol[(size_t)i * TBK + lane], chain_publishable(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c >= 0) z[c] = scale * x;
            lds_barrier();
        }
        return;
    }
    for (int i = 0; i < g.nblk; ++i) {
        if (i + 1 < g.nblk) {
            const bool whole = g.whole(i + 2 - TC_LAG, i + 3);
            const double h = ask_sum(i + 1), zv = zr;             // zr: right-hand side entry of position i + 1, fetched a link ago
            if (wave == 7) rinvs[((i + 2) & 3) * TBK + lane] = recip(dgl);       // before this link's fetches queue up behind it
            const double acc = do_terms(i + 1);
            if (whole) {
                load_terms(i + 2);
                stage_and_load(i + 1);
            } else {
                do_stage(i + 1);
            }
            if (wave == 7) dgl = diag_of(i + 3);
            if (!whole) {
                load_stage(i + 2);
                load_terms(i + 2);
            }
            finish_terms(i + 1, acc, h, zv);
            if (BANK && i + 2 < g.nblk) far_terms(i + 2);
        }
        lds_barrier();
    }
}

vslam_status trsv_upper_dev(vslam_ctx *ctx, const double *S, int n, int ldS, bool trans, const double *b, double *z, double scale)
{
    hipStream_t s = ctx->stream;
    const int nblk = (n + TBK - 1) / TBK;
    const int helpers_wanted = nblk > TC_LAG ? nblk - TC_LAG : 0;
    // the chain's workgroups wait for one another: only when all of them can be resident together (else one launch per block, below)
    if (nblk > 1 && nblk <= TC_MAX && 1 + helpers_wanted <= resident_workgroup_cap()) {
        const size_t lds = (size_t)TC_LDS_DOUBLES * 8, need = (size_t)2 * nblk * TBK;
        VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&trsv_chain_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&trsv_chain_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (ctx->chain_used + need > VSLAM_CHAIN_SOL_DOUBLES) {          // ring used up: every entry back to "not ready", behind the calls that used it
            VSLAM_HIP_TRY(ctx, hipMemsetAsync(ctx->chain_sol, 0xff, VSLAM_CHAIN_SOL_DOUBLES * 8, s));
            ctx->chain_used = 0;
        }
        double *sol = ctx->chain_sol + ctx->chain_used;
        ctx->chain_used += need;
        const int helpers = nblk > TC_LAG ? nblk - TC_LAG : 0;
        // Take the GPU's turn, so that this launch and a blocked factorisation or another chain from another context never
        // wait for members that found no room: always when the launch is larger than what the factorisation's launcher
        // leaves free (16 compute units), and for any multi-workgroup chain once a second context lives on the device --
        // two small chains beside a full-size factorisation add up to more than the device holds.  (A single context
        // queues all such launches on its one stream; a second PROCESS on the GPU is outside what this can order.)
        if (1 + helpers > 16 || (helpers > 0 && contexts_on_device(ctx->device) > 1)) {
            DeviceTurnGuard turn(ctx);
            if (turn.status() != VSLAM_OK) return turn.status();
            if (trans) hipLaunchKernelGGL((trsv_chain_kernel<true, false>), dim3(1 + helpers), dim3(TC_T), lds, s, S, n, ldS, b, scale, z, sol, ctx->flags, (size_t)0, (const int *)nullptr, 0);
            else hipLaunchKernelGGL((trsv_chain_kernel<false, false>), dim3(1 + helpers), dim3(TC_T), lds, s, S, n, ldS, b, scale, z, sol, ctx->flags, (size_t)0, (const int *)nullptr, 0);
            VSLAM_HIP_TRY(ctx, hipGetLastError());
            return VSLAM_OK;
        }
        if (trans) hipLaunchKernelGGL((trsv_chain_kernel<true, false>), dim3(1 + helpers), dim3(TC_T), lds, s, S, n, ldS, b, scale, z, sol, ctx->flags, (size_t)0, (const int *)nullptr, 0);
        else hipLaunchKernelGGL((trsv_chain_kernel<false, false>), dim3(1 + helpers), dim3(TC_T), lds, s, S, n, ldS, b, scale, z, sol, ctx->flags, (size_t)0, (const int *)nullptr, 0);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        return VSLAM_OK;
    }
    if (z != b) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(z, b, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    for (int step = 0; step < nblk; ++step) {
        const int blk = trans ? step : nblk - 1 - step;
        const int i0 = blk * TBK, ib = (n - i0) < TBK ? (n - i0) : TBK;
        const int rest = trans ? n - i0 - ib : i0;
        const int grid = rest > 0 ? (rest + 255) / 256 : 1;
        hipLaunchKernelGGL(trsv_step_kernel, dim3(grid), dim3(256), 0, s, S, n, ldS, trans ? 1 : 0, i0, ib, z);
    }
    if (scale != 1.0) hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, z, n, scale);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

// Strict lower triangle := transpose of the strict upper triangle, for `batch` n x n matrices strideA apart: what a
// symmetric product computed on its upper tiles only needs before something reads the whole matrix.  One workgroup per
// 64 x 64 tile on or above the diagonal, the transposition through LDS so that reads and writes both run along columns.
__global__ __launch_bounds__(256) void mirror_upper_kernel(double *A, int n, int lda, size_t strideA, int T)
{
    __shared__ double tile[64][65];
    // linear id -> (ti <= tj), column by column
    int l = blockIdx.x, tj = (int)((sqrtf(8.0f * (float)l + 1.0f) - 1.0f) * 0.5f);
    while ((tj + 1) * (tj + 2) / 2 <= l) ++tj;
    while (tj * (tj + 1) / 2 > l) --tj;
    const int ti = l - tj * (tj + 1) / 2, i0 = ti * 64, j0 = tj * 64;
    double *Ab = A + (size_t)blockIdx.y * strideA;
    const int a = threadIdx.x & 63, q = threadIdx.x >> 6;
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int r = i0 + a, c = j0 + 4 * p + q;
        tile[4 * p + q][a] = (r < n && c < n) ? Ab[(size_t)c * lda + r] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int row = j0 + a, col = i0 + 4 * p + q;             // element (row, col) of the lower triangle = upper (col, row)
        if (row < n && col < n && row > col) Ab[(size_t)col * lda + row] = tile[a][4 * p + q];
    }
}

vslam_status mirror_upper_dev(vslam_ctx *ctx, double *A, int n, int lda, size_t strideA, int batch)
{
    const int T = (n + 63) / 64;
    hipLaunchKernelGGL(mirror_upper_kernel, dim3(T * (T + 1) / 2, batch), dim3(256), 0, ctx->stream, A, n, lda, strideA, T);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

// Index reversal of symmetric / triangular n x n matrices (ld n, `batch` of them stride apart), upper triangle of the
// result only (64 x 64 tiles that touch it):  B(i, j) = A(n-1-i, n-1-j), or with `transposed` B(i, j) = A(n-1-j, n-1-i)
// -- the upper-triangular J U^T J of an upper-triangular U.  The transposed form goes through an LDS tile so that both
// the reads of A and the writes of B run along columns.  transposed == 2: A is symmetric and only its upper triangle is
// read (an element below the diagonal of B's diagonal tiles comes from its mirror image).
__global__ __launch_bounds__(256) void reverse_upper_kernel(const double *A, int n, double *B, int transposed, size_t stride)
{
    __shared__ double tile[64][65];
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    if (i0 > j0 + 63) return;                                     // wholly below the diagonal
    const double *Ab = A + (size_t)blockIdx.z * stride;
    double *Bb = B + (size_t)blockIdx.z * stride;
    const int a = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (!transposed) {
#pragma unroll 4
        for (int p = 0; p < 16; ++p) {
            const int i = i0 + a, j = j0 + 4 * p + q;
            if (i < n && j < n) Bb[(size_t)j * n + i] = Ab[(size_t)(n - 1 - j) * n + (n - 1 - i)];
        }
        return;
    }
    double v[16];                                                 // all of a thread's reads in flight together
#pragma unroll
    for (int p = 0; p < 16; ++p) {                                // tile[ii][jj] = B(i0 + ii, j0 + jj) = A(n-1-j, n-1-i): rows of A along the lanes
        const int i = i0 + 4 * p + q, j = j0 + a;
        const bool flip = transposed == 2 && i > j;
        const int ci = flip ? j : i, rj = flip ? i : j;
        v[p] = (i < n && j < n) ? Ab[(size_t)(n - 1 - ci) * n + (n - 1 - rj)] : 0.0;
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) tile[4 * p + q][a] = v[p];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int jj = 4 * p + q, i = i0 + a, j = j0 + jj;
        if (i < n && j < n) Bb[(size_t)j * n + i] = tile[a][jj];
    }
}

vslam_status reverse_upper_dev(vslam_ctx *ctx, const double *A, int n, double *B, bool transposed, size_t stride, int batch, bool sym_upper)
{
    const int T = (n + 63) / 64;
    hipLaunchKernelGGL(reverse_upper_kernel, dim3(T, T, batch), dim3(256), 0, ctx->stream, A, n, B, sym_upper ? 2 : transposed ? 1 : 0, stride);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

// The same solves for a bank: matrix m at S + m * strideS, right-hand side and result at rhs / z + m * n (they may be
// the same vectors), matrices with active[m] == 0 skipped (active may be null; listed: active holds `count` matrix
// indices instead).  False when the size is outside what
// the one-workgroup-per-matrix kernel holds in LDS (the caller then takes its own route).
bool trsv_upper_bank_fits(int n) { return n > TBK && (n + TBK - 1) / TBK <= TC_BANK_BLOCKS; }
vslam_status trsv_upper_bank_dev(vslam_ctx *ctx, const double *S, int n, int ldS, size_t strideS, bool trans, const double *rhs, double *z,
                                 double scale, const int *active, int count, bool listed)
{
    if (!trsv_upper_bank_fits(n)) return fail(ctx, VSLAM_ERR_SHAPE, "bank triangular solve: n = %d is outside 65 ... %d", n, TC_BANK_BLOCKS * TBK);
    const int nblk = (n + TBK - 1) / TBK;
    const size_t lds = (size_t)(TC_LDS_DOUBLES - TC_LAG * TBK + nblk * TBK) * 8;
    hipStream_t s = ctx->stream;
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&trsv_chain_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&trsv_chain_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int m0 = 0; m0 < count; m0 += 65535) {
        const int cnt = count - m0 < 65535 ? count - m0 : 65535;
        // a list carries the matrix indices itself; a mask (or no selection) counts from the first matrix of the launch
        const size_t first = listed ? 0 : (size_t)m0;
        const double *Sm = S + first * strideS, *rm = rhs + first * n;
        double *zm = z + first * n;
        const int *am = active ? active + m0 : nullptr;
        if (trans) hipLaunchKernelGGL((trsv_chain_kernel<true, true>), dim3(1, cnt), dim3(TC_T), lds, s, Sm, n, ldS, rm, scale, zm, (double *)nullptr, ctx->flags, strideS, am, listed ? 1 : 0);
        else hipLaunchKernelGGL((trsv_chain_kernel<false, true>), dim3(1, cnt), dim3(TC_T), lds, s, Sm, n, ldS, rm, scale, zm, (double *)nullptr, ctx->flags, strideS, am, listed ? 1 : 0);
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

// Test hook (not part of the public header): z = S^-1 b (trans = 0) or S^-T b (trans = 1) for one upper-triangular
// n x n matrix on the host (column-major, ld n).
extern "C" vslam_status vslam_debug_trsv_reps(vslam_ctx *ctx, const double *S, int n, int trans, const double *b, double *z, int reps);
extern "C" vslam_status vslam_debug_trsv(vslam_ctx *ctx, const double *S, int n, int trans, const double *b, double *z)
{
    return vslam_debug_trsv_reps(ctx, S, n, trans, b, z, 1);
}
// The same solve queued `reps` times back to back (the matrix goes to the device once).
extern "C" vslam_status vslam_debug_trsv_reps(vslam_ctx *ctx, const double *S, int n, int trans, const double *b, double *z, int reps)
{
    using namespace vslam;
    if (!ctx || !S || !b || !z || n < 1) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(nn * 8) + 2 * Arena::pad((size_t)n * 8)));
    Arena ar(ctx->ws);
    double *dS = ar.take<double>(nn), *db = ar.take<double>(n), *dz = ar.take<double>(n);
    VSLAM_TRY(h2d(ctx, dS, S, nn * 8));
    VSLAM_TRY(h2d(ctx, db, b, (size_t)n * 8));
    for (int rep = 0; rep < (reps < 1 ? 1 : reps); ++rep) VSLAM_TRY(trsv_upper_dev(ctx, dS, n, n, trans != 0, db, dz));
    VSLAM_TRY(d2h(ctx, z, dz, (size_t)n * 8));
    return stream_finish(ctx);
}

// Test hook (not part of the public header): the same for `batch` matrices held one after the other ([batch][n][n]
// column-major, right-hand sides and results [batch][n]); active (host, may be null) masks matrices out, whose results
// are left as they were.
extern "C" vslam_status vslam_debug_trsv_bank(vslam_ctx *ctx, const double *S, int n, int batch, int trans, const double *b, double *z, double scale,
                                              const int *active)
{
    using namespace vslam;
    if (!ctx || !S || !b || !z || n < 1 || batch < 1) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(nn * batch * 8) + 2 * Arena::pad((size_t)n * batch * 8) + Arena::pad((size_t)batch * sizeof(int))));
    Arena ar(ctx->ws);
    double *dS = ar.take<double>(nn * batch), *db = ar.take<double>((size_t)n * batch), *dz = ar.take<double>((size_t)n * batch);
    int *da = ar.take<int>(batch);
    VSLAM_TRY(h2d(ctx, dS, S, nn * batch * 8));
    VSLAM_TRY(h2d(ctx, db, b, (size_t)n * batch * 8));
    VSLAM_TRY(h2d(ctx, dz, z, (size_t)n * batch * 8));
    if (active) VSLAM_TRY(h2d(ctx, da, active, (size_t)batch * sizeof(int)));
    VSLAM_TRY(trsv_upper_bank_dev(ctx, dS, n, n, nn, trans != 0, db, dz, scale, active ? da : nullptr, batch));
    VSLAM_TRY(d2h(ctx, z, dz, (size_t)n * batch * 8));
    return stream_finish(ctx);
}
