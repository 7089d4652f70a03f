// Device-resident trust-region Newton iteration of the iterated-EKF measurement update.
//
// The reference's fminNewtonTrustEig (src/fmin.hpp:19-107) takes an eigendecomposition of the
// Hessian at every accepted iterate and solves the trust-region subproblem in the eigenbasis
// (src/fmin.cpp:40-131): p(lam) = -(H + lam I)^-1 g, with Newton's method on
// 1/Delta - 1/|p(lam)| when the full step leaves the region.  When H is positive definite the
// same quantities follow from Cholesky factors H + lam I = U^T U:
//     p = -U^-1 U^-T g,      p^T (H + lam I)^-1 p = |U^-T p|^2,
// so the iteration below produces the reference's sequence of lam (start 0, lower clamp 0, same
// update; it also stops once |p| matches Delta to 1e-10 relative, see trs_multiplier_settled in
// linalg.hpp) without an eigensolver; everything stays in HBM and only a handful of
// scalars cross PCIe per iteration.  An iterate whose Hessian is not positive definite is handed
// back to the caller, which takes the eigenvalue route for it.
//
// The posterior square-root covariance (src/gaussian.hpp:513-516: S = triu(qr(diag(v^-1/2) Q^T)),
// i.e. the upper-triangular S with S^T S = H^-1) is formed as the inverse of the upper-triangular M
// with M M^T = H (a Cholesky factorisation of the index-reversed H).  It equals the reference's S up
// to the signs of its rows, which there depend on the eigensolver's choice of eigenvector signs.
#include "potf.hpp"
#include "trs_search.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <cmath>
#include <limits>

namespace vslam {

typedef double double4_t __attribute__((ext_vector_type(4)));

// Upper Cholesky of the jb x jb diagonal block at (j0, j0): the block goes to LDS, one wave factors it
// (potf2_lds), the four waves invert its 16 x 16 diagonal sub-blocks for the row-panel solve and write the
// factor back.  Only the upper triangle is read and written.  Used for the first block of a factorisation and for
// trailing matrices too small for the GEMM epilogue; every other diagonal block is factored by the workgroup of
// the trailing update that produced it (gemm.hip).
__global__ __launch_bounds__(256) void potf2_upper_kernel(double *A, int lda, int j0, int jb, int *info, size_t strideA, const int *active, int listed,
                                                           double *dinv, size_t stride_dinv)
{
    __shared__ __align__(16) double U[POTF_LDS_DOUBLES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = instance_of(active, listed, blockIdx.y);        // instance of a batch (stride strideA, one info word each)
    if (b < 0) return;
    A += (size_t)b * strideA;
    for (int cb = 16 * wave; cb < 16 * wave + 16; cb += 16) {      // lane = row: coalesced column reads, 16 in flight
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            v[j] = (lane <= cb + j && cb + j < jb) ? A[(size_t)(j0 + cb + j) * lda + j0 + lane] : ((lane == cb + j) ? 1.0 : 0.0);
#pragma unroll
        for (int j = 0; j < 16; ++j) U[lane * PLD + cb + j] = v[j];
    }
    __syncthreads();
    potf2_lds4(U, U + CB * PLD, jb, j0, info + b);
    potf_dinv16(U, U + CB * PLD, dinv + (size_t)blockIdx.y * stride_dinv);
    for (int c = wave; c < jb; c += 4)
        if (lane <= c) A[(size_t)(j0 + c) * lda + j0 + lane] = U[lane * PLD + c];
}

// Row panel of the factor: X = U_JJ^-T * A[J, rest] in place, on the matrix cores.  A workgroup owns 64 columns,
// a wave 16 of them; the 64 x 16 right-hand side of a wave lives in registers as four 16 x 16 blocks in the
// accumulator layout of v_mfma_f64_16x16x4_f64, which is also the layout of its B operand, so the blocked forward
// substitution
//     X_i = (U_ii)^-T R_i,     R_j -= (U_ij)^T X_i   (j > i)
// is a chain of MFMA groups with no shuffles in between.  (U_ii)^-T comes from the factorisation of the diagonal
// block (potf_dinv16), U_ij from the factored block itself.
constexpr int TLD = 17;          // row stride of the 16 x 16 inverses in LDS
// A right-hand side vector (R, one per matrix) rides along as column n: when the factorisation ends it holds U^-T r.
__global__ __launch_bounds__(256) void potrf_trsm_kernel(double *A, int lda, int n, int j0, size_t strideA, const int *active, int listed,
                                                          const double *dinv, size_t stride_dinv, double *R, size_t strideR)
{
    __shared__ __align__(16) double U[CB * PLD];                   // U[l][j] at l * PLD + j (upper triangle meaningful)
    __shared__ double Yt[4 * 16 * TLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kr = lane >> 4;
    int bx, by;
    bank_grid_xcd(&bx, &by);                                       // a matrix's column blocks on one XCD: the update finds the panel in that L2
    const int b = instance_of(active, listed, by);
    if (b < 0) return;
    A += (size_t)b * strideA;
    dinv += (size_t)by * stride_dinv;                              // scratch of the grid row, as the kernels before and after use it
    const int jb = min(CB, n - j0);                                // the last block row may be short (then only R is to its right)
    const int c0 = j0 + jb + bx * CB;
    const int col = c0 + 16 * wave + l15;
    const bool cok = col < n + (R ? 1 : 0);
    // right-hand side: block bi, register q <-> row 16 bi + 4 q + kr of this lane's column
    double4_t Rb[4];
    double *colp = (cok && col == n) ? R + (size_t)b * strideR + j0 : A + (size_t)(cok ? col : min(c0, n - 1)) * lda + j0;
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int q = 0; q < 4; ++q) Rb[bi][q] = (cok && 16 * bi + 4 * q + kr < jb) ? colp[16 * bi + 4 * q + kr] : 0.0;
    {
        double v[16];                                              // lane = row: coalesced column reads, 16 in flight
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = (lane <= 16 * wave + j && 16 * wave + j < jb) ? A[(size_t)(j0 + 16 * wave + j) * lda + j0 + lane] : 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) U[lane * PLD + 16 * wave + j] = v[j];
        for (int e = tid; e < 1024; e += 256) Yt[(e >> 4) * TLD + (e & 15)] = dinv[e];
    }
    __syncthreads();
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
        double4_t X = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4)
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(Yt[(16 * bi + l15) * TLD + k0 + kr], Rb[bi][k0 >> 2], X, 0, 0, 0);
        Rb[bi] = X;
#pragma unroll
        for (int bj = bi + 1; bj < 4; ++bj)
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4)
                Rb[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(16 * bi + k0 + kr) * PLD + 16 * bj + l15], X[k0 >> 2], Rb[bj], 0, 0, 0);
    }
    if (cok) {
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (16 * bi + 4 * q + kr < jb) colp[16 * bi + 4 * q + kr] = Rb[bi][q];
    }
}

// Trailing update of the factorisation: C (N x N, its origin on the diagonal of the matrix; the tiles that touch
// the upper triangle) -= P^T P, P = the K finished rows above C, which lie contiguously in every column.  One
// workgroup per 64 x 64 tile of the upper triangle (no workgroups for the others).  At K = 64 the launch is bound by
// the read-modify-write of C and by the matrix cores in about equal parts, so the kernel is kept minimal: the old
// tile is requested first, P goes through LDS sixteen rows at a time with plain indexing (row stride 18 doubles:
// the 32 lanes of a read phase cover all banks), four workgroups per compute unit cover one another's waits.
// The workgroup of tile (0, 0) then factors its tile -- the next diagonal block (potf.hpp) -- while the rest of the
// launch is still updating.
struct CholUpdArgs {
    const double *P;               // element (first finished row, first column of C)
    double *C;                     // element (0, 0) of the trailing block
    int N, K, lda;
    size_t strideA;
    const int *active; int listed;
    double *dinv; size_t stride_dinv; int *info; int j0;
    double *R; size_t strideR;     // optional right-hand side riding along as column N of the trailing block:
    int k0g, r0g;                  //   its entries beside P start at R[k0g], those beside C at R[r0g]
    const double *Csrc;            // optional: the old values of the trailing block come from here (same layout, strideSrc apart)
    size_t strideSrc;              //   plus a shift of the diagonal -- the first update of a factorisation that does not
    const double *lamv; double lam; //   start from a copy of the matrix (lamv: one shift per matrix, else lam)
    int strip;                     // 1: only the first block row of C (tiles (0, 0 .. )), the launch has one workgroup per tile column
};
constexpr int UKC = 16, ULD = UKC + 2;
// linear id -> tile (tm <= tn) of the upper triangle, column by column
__device__ __forceinline__ void upper_tile(int l, int *tm, int *tn)
{
    int n = (int)((sqrtf(8.0f * (float)l + 1.0f) - 1.0f) * 0.5f);
    while ((n + 1) * (n + 2) / 2 <= l) ++n;
    while (n * (n + 1) / 2 > l) --n;
    *tn = n; *tm = l - n * (n + 1) / 2;
}
__global__ __launch_bounds__(256, 4) void chol_update_kernel(CholUpdArgs g)
{
    __shared__ __align__(16) double Ps[POTF_LDS_DOUBLES];             // operand chunks; afterwards the diagonal block of tile (0, 0)
    static_assert(2 * CB * ULD <= POTF_LDS_DOUBLES, "operand chunks fit");
    double *const Pm = Ps, *const Pn = Ps + CB * ULD;                 // [column of the tile][row of the chunk]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kr = lane >> 4;
    int bx, by;
    bank_grid_xcd(&bx, &by);                                       // the tiles of a matrix on one XCD: its row panel is fetched into one L2
    const int b = instance_of(g.active, g.listed, by);
    if (b < 0) return;
    const int T = (g.N + CB - 1) / CB, tri = T * (T + 1) / 2;
    int tile_m, tile_n;
    if (g.strip) { tile_m = 0; tile_n = bx; }
    else if (bx < tri) upper_tile(bx, &tile_m, &tile_n);
    else { tile_n = T; tile_m = bx - tri; }           // the right-hand side in a tile column of its own
    const int m0 = tile_m * CB, n0 = tile_n * CB;
    const double *P = g.P + (size_t)b * g.strideA;
    double *C = g.C + (size_t)b * g.strideA;
    double *R = g.R ? g.R + (size_t)b * g.strideR : nullptr;
    const int NC = g.N + (R ? 1 : 0);                              // columns: the trailing block and the right-hand side
    auto pcol = [&](int c) -> const double * { return c == g.N ? R + g.k0g : P + (size_t)c * g.lda; };
    auto ccol = [&](int c) -> double * { return c == g.N ? R + g.r0g : C + (size_t)c * g.lda; };
    const double *Cold = g.Csrc ? g.Csrc + (size_t)b * g.strideSrc : C;
    auto ocol = [&](int c) -> const double * { return c == g.N ? R + g.r0g : Cold + (size_t)c * g.lda; };
    // the shift of a source's diagonal is added where the old values are used, not where they are requested: the
    // requests stay in flight under the products
    const double shift = (g.Csrc && m0 == n0) ? (g.lamv ? g.lamv[b] : g.lam) : 0.0;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const bool diag = m0 == n0 && (!R || n0 + CB <= g.N);      // one operand block serves both sides (not when the right-hand side is among the tile's columns)
    // old values of the tile, in the accumulator layout: [i][j][r] <-> row m0+wm+16i+l15, column n0+wn+16j+kr+4r
    double cold[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(m0 + wm + 16 * i + l15, g.N - 1), n = min(n0 + wn + 16 * j + kr + 4 * r, NC - 1);
                cold[i][j][r] = ocol(n)[m];
            }
    double4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const double *Bn = diag ? Pm : Pn;
    if (bx == 0 && g.K <= CB && diag) {
        // tile (0, 0) heads the chain of diagonal blocks: its K x 64 operand in one trip through memory
        // (lane = row, a wave fetches whole columns), then the products straight through
        double ra[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) ra[e] = lane < g.K ? P[(size_t)min(wave + 4 * e, g.N - 1) * g.lda + lane] : 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) Ps[(wave + 4 * e) * PLD + lane] = ra[e];
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < CB; kk += 4) {
            double am[2], bn[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) am[i] = Ps[(wm + 16 * i + l15) * PLD + kk + kr];
#pragma unroll
            for (int j = 0; j < 2; ++j) bn[j] = Ps[(wn + 16 * j + l15) * PLD + kk + kr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[j], am[i], acc[i][j], 0, 0, 0);
        }
    } else
    for (int k0 = 0; k0 < g.K; k0 += UKC) {
        double ra[4], rb[4];                                        // thread: row tid % 16 of the chunk, columns tid / 16 + 16 e
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (tid >> 4) + 16 * e, kk = tid & 15;
            ra[e] = P[(size_t)min(m0 + c, g.N - 1) * g.lda + k0 + kk];
            if (!diag) rb[e] = pcol(min(n0 + c, NC - 1))[k0 + kk];
        }
        if (k0 > 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (tid >> 4) + 16 * e, kk = tid & 15;
            Pm[c * ULD + kk] = ra[e];
            if (!diag) Pn[c * ULD + kk] = rb[e];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < UKC; kk += 4) {
            double am[2], bn[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) am[i] = Pm[(wm + 16 * i + l15) * ULD + kk + kr];
#pragma unroll
            for (int j = 0; j < 2; ++j) bn[j] = Bn[(wn + 16 * j + l15) * ULD + kk + kr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[j], am[i], acc[i][j], 0, 0, 0);
        }
    }
    if (bx == 0) {                                                  // tile (0, 0): the next diagonal block
        __syncthreads();
        double *U = Ps, *invd = Ps + CB * PLD;
        const int jb = min(CB, g.N);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = wm + 16 * i + l15, n = wn + 16 * j + kr + 4 * r;
                    U[m * PLD + n] = (m <= n && n < jb) ? cold[i][j][r] + (m == n ? shift : 0.0) - acc[i][j][r] : (m == n ? 1.0 : 0.0);
                    if (R && n == g.N && m < g.N) R[g.r0g + m] = cold[i][j][r] - acc[i][j][r];    // a trailing block narrower than the tile
                }
        __syncthreads();
        potf2_lds4(U, invd, jb, g.j0, g.info + b);
        potf_dinv16(U, invd, g.dinv + (size_t)by * g.stride_dinv);       // scratch of the grid row (the row panel kernel reads it there)
        for (int c = wave; c < jb; c += 4)
            if (lane <= c) C[(size_t)c * g.lda + lane] = U[lane * PLD + c];
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + l15, n = n0 + wn + 16 * j + kr + 4 * r;
                if (m < g.N && n < NC) ccol(n)[m] = cold[i][j][r] + (m == n ? shift : 0.0) - acc[i][j][r];
            }
}

__global__ void add_diag_kernel(double *A, int lda, int n, double lam)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(size_t)i * lda + i] += lam;
}

// Rows 0 ... jb-1 of the upper triangle of A := those of Src, the diagonal shifted by lam: what the first diagonal block
// and the first row panel of a factorisation read when the matrix is not copied as a whole (the first trailing update
// takes the rest from Src itself).  A wave per column: its lanes are the (at most 64) rows.
__global__ void potrf_first_rows_kernel(double *A, int lda, size_t strideA, const double *Src, size_t strideSrc, int n, int jb,
                                        const double *lamv, double lam, const int *active, int listed)
{
    const int b = instance_of(active, listed, blockIdx.y);
    if (b < 0) return;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), r = threadIdx.x & 63;
    if (c >= n || r >= jb || r > c) return;
    const double shift = lamv ? lamv[b] : lam;
    A[(size_t)b * strideA + (size_t)c * lda + r] = Src[(size_t)b * strideSrc + (size_t)c * lda + r] + (r == c ? shift : 0.0);
}

// strict lower triangle := 0 for matrix blockIdx.y of a batch (grid-stride over the n*n elements)
__global__ void zero_below_diag_kernel(double *A, int lda, int n, size_t strideA, const int *active, int listed)
{
    const int b = instance_of(active, listed, blockIdx.y);
    if (b < 0) return;
    double *Ab = A + (size_t)b * strideA;
    const size_t nn = (size_t)n * n, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nn; idx += stride) {
        const int c = (int)(idx / n), r = (int)(idx - (size_t)c * n);
        if (r > c) Ab[(size_t)c * lda + r] = 0.0;
    }
}

// In-place upper Cholesky A = U^T U of `batch` symmetric n x n matrices (element stride strideA; both
// triangles are overwritten: U above, zeros below).  info: one device word per matrix, 0 on entry.
// active (device, may be null): matrices whose entry is 0 are left alone (except for the zeroing of
// their strict lower triangle).  zero_lower = false leaves the strict lower triangle as it is (a caller
// that only ever reads the upper triangle of the factor saves the pass).  listed: `active` is a list of `batch`
// matrix indices instead of a mask over `batch` matrices (info, rhs and A are still indexed by matrix).
// from: factor from->H + shift * I (same ld, matrices from->stride apart) into A without copying it first; only the upper
// triangle of A is then written.
vslam_status potrf_upper_batch_dev(vslam_ctx *ctx, double *A, int n, int lda, size_t strideA, int batch, int *info, const int *active,
                                   bool zero_lower, double *rhs, size_t stride_rhs, bool listed, const PotrfFrom *from)
{
    hipStream_t s = ctx->stream;
    if (from)
        hipLaunchKernelGGL(potrf_first_rows_kernel, dim3((n + 3) / 4, batch), dim3(256), 0, s, A, lda, strideA, from->H, from->stride, n, n < CB ? n : CB,
                           from->lamv, from->lam, active, listed ? 1 : 0);
    VSLAM_TRY(potf_dinv_reserve(ctx, batch));
    double *dinv = ctx->potf_dinv;
    const int extra = rhs ? 1 : 0;                    // the right-hand side rides along as column n
    bool factored = false;                            // the diagonal block of this step was factored by the previous update
    // Banks take the trailing update two block rows at a time: after the row panel of block row k only block row k + 1 is
    // updated (a strip of tiles, whose first one factors the next diagonal block), its row panel is solved, and the rest of
    // the trailing matrix then receives both block rows in one pass with K = 128 -- the read-modify-write of C, which is
    // what a bank's update launches spend their time on, happens once per 128 columns.  A large matrix gains a little too
    // (n = 6012: 88.6 -> 86.2 ms per update); a small single one gains nothing (its launches are as long as the chain of
    // diagonal blocks) and keeps the one-level form, bit for bit.
    const bool two_level = (batch >= 16 || n >= 2048) && !getenv("VSLAM_POTRF_ONE_LEVEL");
    auto launch_trsm = [&](int j0, int rest) {
        hipLaunchKernelGGL(potrf_trsm_kernel, dim3((rest + extra + CB - 1) / CB, batch), dim3(256), 0, s, A, lda, n, j0, strideA, active, listed ? 1 : 0, dinv, (size_t)1024,
                           rhs, stride_rhs);
    };
    // C = the trailing block whose origin is (c0, c0), N columns; P = the K finished rows from row p0 on above it
    auto launch_update = [&](int p0, int K, int c0, int N, bool strip) {
        CholUpdArgs g;                                // A[rest, rest] -= X^T X (upper triangle; the rest is zeroed at the end)
        g.P = A + (size_t)c0 * lda + p0; g.C = A + (size_t)c0 * lda + c0; g.N = N; g.K = K; g.lda = lda;
        g.strideA = strideA; g.active = active; g.listed = listed ? 1 : 0; g.dinv = dinv; g.stride_dinv = 1024; g.info = info; g.j0 = c0;
        g.R = rhs; g.strideR = stride_rhs; g.k0g = p0; g.r0g = c0;
        g.Csrc = nullptr; g.strideSrc = 0; g.lamv = nullptr; g.lam = 0.0; g.strip = strip ? 1 : 0;
        if (from && p0 == 0) { g.Csrc = from->H + (size_t)c0 * lda + c0; g.strideSrc = from->stride; g.lamv = from->lamv; g.lam = from->lam; }
        const int T = (N + CB - 1) / CB, TC = (N + extra + CB - 1) / CB;
        const int tiles = strip ? TC : T * (T + 1) / 2 + (TC > T ? T : 0);
        hipLaunchKernelGGL(chol_update_kernel, dim3(tiles, batch), dim3(256), 0, s, g);
    };
    int group = two_level ? 2 : 1;                    // block rows per pass over the trailing matrix (3 and 4 measured no better: 87.3, 89.8 ms against 86.7)
    if (const char *e = getenv("VSLAM_POTRF_GROUP")) { const int v = atoi(e); if (v >= 1 && v <= 8) group = v; }      // A/B runs
    for (int j0 = 0; j0 < n;) {
        const int jb = (n - j0) < CB ? (n - j0) : CB;
        if (!factored) hipLaunchKernelGGL(potf2_upper_kernel, dim3(1, batch), dim3(256), 0, s, A, lda, j0, jb, info, strideA, active, listed ? 1 : 0, dinv, (size_t)1024);
        const int rest = n - j0 - jb;
        if (rest + extra <= 0) break;
        int g = group;                                // block rows of this pass: all full, and something left beyond them
        while (g > 1 && n - j0 - g * CB <= 0) --g;
        if (g == 1) {
            launch_trsm(j0, rest);
            VSLAM_HIP_TRY(ctx, hipGetLastError());
            if (rest <= 0) break;
            factored = true;                          // the update's workgroup for the next diagonal tile factors it as well
            launch_update(j0, jb, j0 + jb, rest, false);
            VSLAM_HIP_TRY(ctx, hipGetLastError());
            j0 += CB;
            continue;
        }
        for (int i = 0; i < g; ++i) {
            const int jj = j0 + i * CB;
            launch_trsm(jj, n - jj - CB);
            // block row i + 1 of the group receives the block rows before it (a strip of tiles; its first one factors the
            // next diagonal block); the last one is followed by the pass over everything beyond the group
            if (i + 1 < g) launch_update(j0, (i + 1) * CB, jj + CB, n - jj - CB, true);
        }
        launch_update(j0, g * CB, j0 + g * CB, n - j0 - g * CB, false);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        factored = true;
        j0 += g * CB;
    }
    if (!zero_lower) return VSLAM_OK;
    const size_t zblocks = ((size_t)n * n + 4095) / 4096;
    hipLaunchKernelGGL(zero_below_diag_kernel, dim3((unsigned)(zblocks < 1024 ? zblocks : 1024), batch), dim3(256), 0, s, A, lda, n, strideA, active, listed ? 1 : 0);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

vslam_status potrf_upper_dev(vslam_ctx *ctx, double *A, int n, int lda, int *info)
{
    return potrf_upper_batch_dev(ctx, A, n, lda, 0, 1, info, nullptr);
}

// out[i] = a_i . b_i for up to six vector pairs of length n (null pairs give 0).  One workgroup.  out[6] carries
// *info (the factorisation's verdict travels to the host with the dot products).
struct DotArgs { const double *a[6]; const double *b[6]; };
__global__ __launch_bounds__(1024) void dots_kernel(DotArgs d, int n, double *out, const int *info)
{
    __shared__ double red[6][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < n; i += 1024)
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (d.a[q]) acc[q] += d.a[q][i] * d.b[q][i];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        double v = acc[q];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[q][wave] = v;
    }
    __syncthreads();
    if (tid < 6) {
        double sum = 0.0;
        for (int w = 0; w < 16; ++w) sum += red[tid][w];
        out[tid] = sum;
        if (tid == 0) out[6] = info ? (double)*info : 0.0;
    }
}

__global__ void axpby_kernel(double a, const double *x, double b, const double *y, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.0);
}

// B(i, j) = A(n-1-i, n-1-j), or with `transposed` B(i, j) = A(n-1-j, n-1-i): both index reversals of a square matrix.
// y = A x for a dense column-major n x n matrix, read once at memory speed: workgroup (rb, slab) takes rows
// 64 rb ... 64 rb + 63 and every MV_SLABS-th group of four columns; a wave reads 64 consecutive rows of a column
// (one 512-byte stretch), the four waves hold different columns.  part[slab][row] receives the slab's sum;
// matvec_sum_kernel adds the slabs in a fixed order.
constexpr int MV_SLABS = 16;
__global__ __launch_bounds__(256) void matvec_slab_kernel(const double *A, int lda, int n, const double *x, double *part)
{
    __shared__ double red[4][64];
    const int r = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6, slab = blockIdx.y;
    const int rr = min(r, n - 1);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int c = 4 * slab + q;
    for (; c + 12 * MV_SLABS < n; c += 16 * MV_SLABS) {
        const double v0 = A[(size_t)c * lda + rr], v1 = A[(size_t)(c + 4 * MV_SLABS) * lda + rr];
        const double v2 = A[(size_t)(c + 8 * MV_SLABS) * lda + rr], v3 = A[(size_t)(c + 12 * MV_SLABS) * lda + rr];
        a0 = fma(v0, x[c], a0); a1 = fma(v1, x[c + 4 * MV_SLABS], a1);
        a2 = fma(v2, x[c + 8 * MV_SLABS], a2); a3 = fma(v3, x[c + 12 * MV_SLABS], a3);
    }
    for (; c < n; c += 4 * MV_SLABS) a0 = fma(A[(size_t)c * lda + rr], x[c], a0);
    red[q][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 64 && r < n)
        part[(size_t)slab * n + r] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void matvec_sum_kernel(const double *part, int n, double *y)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double acc = 0.0;
#pragma unroll
    for (int sl = 0; sl < MV_SLABS; ++sl) acc += part[(size_t)sl * n + r];
    y[r] = acc;
}

size_t newton_ws_bytes(int n)
{
    const size_t nn = (size_t)n * n;
    return 2 * Arena::pad(nn * 8) + 8 * Arena::pad((size_t)n * 8) + Arena::pad(64) + Arena::pad(64);
}

namespace {
struct NewtonDevice {
    double *U, *Hcur, *gcur, *xcur, *p, *t, *q, *xn, *Hp, *spare, *dots;
    int *info;
};
void newton_carve(void *ws, int n, NewtonDevice *w)
{
    Arena ar(ws);
    const size_t nn = (size_t)n * n;
    w->U = ar.take<double>(nn); w->Hcur = ar.take<double>(nn);
    w->gcur = ar.take<double>(n); w->xcur = ar.take<double>(n); w->p = ar.take<double>(n); w->t = ar.take<double>(n);
    w->q = ar.take<double>(n); w->xn = ar.take<double>(n); w->Hp = ar.take<double>(n); w->spare = ar.take<double>(n);
    w->dots = ar.take<double>(8);
    w->info = ar.take<int>(16);
}
}  // namespace

// Runs the whole update on the device.  On return *status_out is
//   0  finished: mu_post (host) and S_post (host, ld ldSp) hold the result, fmin_code/iters/evals set;
//   1  an iterate's Hessian was not positive definite: the outputs are meaningless and the caller
//      should take the eigenvalue route from the start.
vslam_status iekf_newton_device(vslam_ctx *ctx, const CamK &k, double sigma, CostDevice *c, void *ws,
                                double *mu_post, double *S_post, int ldSp, int *fmin_code, int *iters, int *evals_out,
                                int *status_out, bool device_out)
{
    const int n = c->n;
    const size_t nn = (size_t)n * n;
    hipStream_t s = ctx->stream;
    NewtonDevice w;
    newton_carve(ws, n, &w);
    const double eps = std::numeric_limits<double>::epsilon();
    const int nb = (n + 255) / 256;
    int evals = 0, factorisations = 0;
    double hd[8];
    int hinfo = 0;
    *status_out = 0;
    // Wall-clock split of the call for the bench's breakdown; every phase below ends at a stream synchronisation
    // (the cost value, the dot products and the final copies are read on the host), so the split costs nothing.
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    const clk::time_point t_call = clk::now();
    clk::time_point t_phase = t_call;
    double ms_cost = 0.0, ms_trs = 0.0;
    auto phase_to = [&](double &acc) { acc += ms_since(t_phase); t_phase = clk::now(); };

    auto dots = [&](const double *a0, const double *b0, const double *a1, const double *b1, const double *a2, const double *b2,
                    const double *a3 = nullptr, const double *b3 = nullptr, const double *a4 = nullptr, const double *b4 = nullptr,
                    const double *a5 = nullptr, const double *b5 = nullptr) -> vslam_status {
        DotArgs d;
        d.a[0] = a0; d.b[0] = b0; d.a[1] = a1; d.b[1] = b1; d.a[2] = a2; d.b[2] = b2; d.a[3] = a3; d.b[3] = b3;
        d.a[4] = a4; d.b[4] = b4; d.a[5] = a5; d.b[5] = b5;
        hipLaunchKernelGGL(dots_kernel, dim3(1), dim3(1024), 0, s, d, n, w.dots, (const int *)w.info);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(d2h(ctx, hd, w.dots, 7 * 8));
        VSLAM_TRY(stream_sync(ctx));
        hinfo = (int)hd[6];
        return VSLAM_OK;
    };
    // moments of the multiplier search (trs_search.hpp) at the current factor: |p|^2, |U^-T p|^2, |U^-1 U^-T p|^2, |U^-T U^-1 U^-T p|^2
    auto moments = [&]() -> vslam_status {
        VSLAM_TRY(trsv_upper_dev(ctx, w.U, n, n, true, w.p, w.q));
        VSLAM_TRY(trsv_upper_dev(ctx, w.U, n, n, false, w.q, w.t));
        VSLAM_TRY(trsv_upper_dev(ctx, w.U, n, n, true, w.t, w.spare));
        return dots(w.p, w.p, w.q, w.q, w.t, w.t, w.spare, w.spare, w.gcur, w.gcur, w.gcur, w.p);
    };
    // U^T U = Hcur + lam I;  p = -(Hcur + lam I)^-1 gcur
    auto factor_and_step = [&](double lam) -> vslam_status {
        VSLAM_HIP_TRY(ctx, hipMemcpyAsync(w.t, w.gcur, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        // t = U^-T g comes with the factor; Hcur is read where it lies (no copy), the strict lower triangle of U keeps
        // stale values, which the solves never read
        PotrfFrom from{w.Hcur, 0, nullptr, lam};
        VSLAM_TRY(potrf_upper_batch_dev(ctx, w.U, n, n, 0, 1, w.info, nullptr, /*zero_lower=*/false, w.t, 0, false, &from));
        ++factorisations;
        return trsv_upper_dev(ctx, w.U, n, n, false, w.t, w.p, -1.0);
    };
    auto eval_at = [&](const double *x, double *f) -> vslam_status {
        if (x != c->x) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(c->x, x, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        VSLAM_TRY(cost_eval_dev(ctx, k, sigma, c));
        VSLAM_TRY(d2h(ctx, f, c->vals + 2, 8));
        VSLAM_TRY(stream_sync(ctx));
        ++evals;
        phase_to(ms_cost);
        return VSLAM_OK;
    };
    auto accept = [&]() -> vslam_status {                    // current iterate := the point just evaluated
        VSLAM_HIP_TRY(ctx, hipMemcpyAsync(w.xcur, c->x, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        VSLAM_HIP_TRY(ctx, hipMemcpyAsync(w.gcur, c->g, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        VSLAM_HIP_TRY(ctx, hipMemcpyAsync(w.Hcur, c->H, nn * 8, hipMemcpyDeviceToDevice, s));
        return VSLAM_OK;
    };

    VSLAM_HIP_TRY(ctx, hipMemsetAsync(w.info, 0, 16 * sizeof(int), s));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(c->x, c->mu, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    double f = 0.0, fn = 0.0, Delta = 1e0;
    VSLAM_TRY(eval_at(c->x, &f));                                        // src/fmin.hpp:39
    VSLAM_TRY(accept());
    int ret = 1, it;
    const bool trace = getenv("VSLAM_TRS_TRACE") != nullptr;       // diagnostic: the multipliers tried, per iteration
    double warm_lam = 0.0;                                         // where the next search starts after an indefinite Hessian
    for (it = 0; it < 50; ++it) {
        // ---- trust-region subproblem, src/fmin.cpp:40-131, by the multiplier search of trs_search.hpp ----
        TrsSlot ts;
        ts.Delta = Delta;
        if (warm_lam > 0.0) trs_warm_start(ts, warm_lam);
        for (bool first = true;; first = false) {
            if (trace) std::fprintf(stderr, "  iteration %d: Delta %.6g, factorisation at multiplier %.10g\n", it, Delta, ts.lam);
            VSLAM_TRY(factor_and_step(ts.lam));
            if (first) VSLAM_TRY(dots(w.p, w.p, nullptr, nullptr, nullptr, nullptr));
            else VSLAM_TRY(moments());
            if (hinfo != 0) {
                // H + lam I is not positive definite: its first non-positive pivot sits on the factor's diagonal
                double pivot = 0.0;
                VSLAM_TRY(d2h(ctx, &pivot, w.U + (size_t)(hinfo - 1) * n + (hinfo - 1), 8));
                VSLAM_HIP_TRY(ctx, hipMemsetAsync(w.info, 0, 16 * sizeof(int), s));
                VSLAM_TRY(stream_sync(ctx));
                if (trs_after_failure(ts, pivot) == TRS_GIVE_UP) { *status_out = 1; return VSLAM_OK; }
                continue;
            }
            if (first) {
                if (trs_interior(ts, hd[0])) break;                 // the Newton step lies inside the region
                VSLAM_TRY(moments());
            }
            const TrsNext next = trs_after_solve(ts, hd[0], hd[1], hd[2], hd[3], hd[4], -hd[5]);
            if (next == TRS_GIVE_UP) { *status_out = 1; return VSLAM_OK; }
            if (next == TRS_DONE) break;
            if (next == TRS_REFACTOR_LAST) { VSLAM_TRY(factor_and_step(ts.lam)); break; }      // its verdict arrives with the dot products below
        }
        warm_lam = ts.indefinite ? ts.lam : 0.0;
        // ---- src/fmin.hpp:54-105 ----
        if (n >= MV_SLABS) {                           // Hp = Hcur * p; the factor in w.U is spent, its memory takes the slab sums
            hipLaunchKernelGGL(matvec_slab_kernel, dim3((n + 63) / 64, MV_SLABS), dim3(256), 0, s, w.Hcur, n, n, w.p, w.U);
            hipLaunchKernelGGL(matvec_sum_kernel, dim3(nb), dim3(256), 0, s, w.U, n, w.Hp);
            VSLAM_HIP_TRY(ctx, hipGetLastError());
        } else {
            GemmDesc gv;
            gv.M = n; gv.N = 1; gv.K = n; gv.transA = false; gv.transB = false;
            gv.A = w.Hcur; gv.lda = n; gv.B = w.p; gv.ldb = n; gv.C = w.Hp; gv.ldc = n;
            VSLAM_TRY(gemm_f64(ctx, gv));
        }
        hipLaunchKernelGGL(axpby_kernel, dim3(nb), dim3(256), 0, s, 1.0, w.xcur, 1.0, w.p, w.xn, n);
        VSLAM_TRY(dots(w.p, w.gcur, w.p, w.Hp, w.p, w.p));
        if (hinfo != 0) { *status_out = 1; return VSLAM_OK; }
        const double pg = hd[0], pHp = hd[1], pn = std::sqrt(hd[2]);
        phase_to(ms_trs);
        if (std::fabs(-pg) < 2 * eps) { ret = 0; break; }
        VSLAM_TRY(eval_at(w.xn, &fn));
        const double dm = -pg - 0.5 * pHp, rho = (f - fn) / dm;
        if (rho < 0.1) Delta = 0.25 * pn;
        else if (rho > 0.75 && pn > 0.8 * Delta) Delta = 2.0 * Delta;
        if (rho >= 0.001) { f = fn; VSLAM_TRY(accept()); }
    }
    // ---- posterior: the upper-triangular S with S^T S = H^-1 at the final iterate.  With J the index reversal,
    // J H J = U'^T U' (Cholesky) gives H = M M^T for the upper-triangular M = J U'^T J, hence S = M^-1
    // (S^T S = M^-T M^-1 = H^-1): one factorisation and one triangular inverse, no QR. ----
    VSLAM_TRY(reverse_upper_dev(ctx, w.Hcur, n, w.U, false, 0, 1));
    VSLAM_TRY(potrf_upper_batch_dev(ctx, w.U, n, n, 0, 1, w.info, nullptr, /*zero_lower=*/false));      // the reversal reads the upper tiles only
    VSLAM_TRY(reverse_upper_dev(ctx, w.U, n, c->Hl, true, 0, 1));                                  // Hl, W: scratch from here on
    const bool piped = !device_out && n >= PIPELINED_COPY_MIN_N;       // the factor goes home as its upper triangle, the zeros below are written on the host
    VSLAM_TRY(trtri_upper_dev(ctx, c->Hl, n, n, c->W, n, c->tws, /*zero_lower=*/!piped));
    VSLAM_TRY(d2h(ctx, &hinfo, w.info, sizeof(int)));
    if (device_out) {
        VSLAM_HIP_TRY(ctx, hipMemcpyAsync(mu_post, w.xcur, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        VSLAM_HIP_TRY(ctx, hipMemcpy2DAsync(S_post, (size_t)ldSp * 8, c->W, (size_t)n * 8, (size_t)n * 8, (size_t)n, hipMemcpyDeviceToDevice, s));
    } else {
        VSLAM_TRY(d2h(ctx, mu_post, w.xcur, (size_t)n * 8));
        if (piped) VSLAM_TRY(d2h_upper_pipelined(ctx, S_post, (size_t)ldSp, c->W, (size_t)n, n));
        else VSLAM_TRY(d2h_2d(ctx, S_post, (size_t)ldSp * 8, c->W, (size_t)n * 8, (size_t)n * 8, n));
    }
    VSLAM_TRY(stream_finish(ctx));
    if (hinfo != 0) { *status_out = 1; return VSLAM_OK; }
    *fmin_code = ret; *iters = it; *evals_out = evals;
    ctx->prof[0] = ms_cost; ctx->prof[1] = ms_trs; ctx->prof[2] = ms_since(t_phase); ctx->prof[3] = ms_since(t_call);
    ctx->prof[4] = evals; ctx->prof[5] = factorisations; ctx->prof[6] = it;
    return VSLAM_OK;
}

}  // namespace vslam

// Test hook (not part of the public header): in-place upper Cholesky factors of `batch` symmetric n x n matrices held
// one after the other on the host (column-major, ld n; only the upper triangles are read); info[b] = 0 or the
// 1-based column of the first non-positive pivot.  rhs (optional, [batch][n]): replaced by U^-T rhs.
extern "C" vslam_status vslam_debug_potrf(vslam_ctx *ctx, double *A, int n, int batch, int *info, double *rhs)
{
    using namespace vslam;
    if (!ctx || !A || !info || n < 1 || batch < 1) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n, bytes = nn * batch * 8, rbytes = (size_t)n * batch * 8;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(bytes) + Arena::pad((size_t)batch * sizeof(int)) + Arena::pad(rbytes)));
    Arena ar(ctx->ws);
    double *dA = ar.take<double>(nn * batch);
    int *dinfo = ar.take<int>(batch);
    double *dr = ar.take<double>((size_t)n * batch);
    VSLAM_TRY(h2d(ctx, dA, A, bytes));
    if (rhs) VSLAM_TRY(h2d(ctx, dr, rhs, rbytes));
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(dinfo, 0, (size_t)batch * sizeof(int), ctx->stream));
    VSLAM_TRY(potrf_upper_batch_dev(ctx, dA, n, n, nn, batch, dinfo, nullptr, true, rhs ? dr : nullptr, (size_t)n));
    VSLAM_TRY(d2h(ctx, A, dA, bytes));
    VSLAM_TRY(d2h(ctx, info, dinfo, (size_t)batch * sizeof(int)));
    if (rhs) VSLAM_TRY(d2h(ctx, rhs, dr, rbytes));
    return stream_finish(ctx);
}

// Test hook (not part of the public header): upper Cholesky factors of H[b] + shift[b] I for `batch` symmetric n x n
// matrices on the host, factored out of place (U, [batch][n][n] column-major, receives the upper triangles; its strict
// lower triangles are left as the caller filled them).  sel (optional): a list of `nsel` matrix indices to factor.
extern "C" vslam_status vslam_debug_potrf_from(vslam_ctx *ctx, const double *H, int n, int batch, const double *shift, int *info, double *U,
                                               const int *sel, int nsel)
{
    using namespace vslam;
    if (!ctx || !H || !info || !U || !shift || n < 1 || batch < 1) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n, bytes = nn * batch * 8;
    VSLAM_TRY(ws_reserve(ctx, 2 * Arena::pad(bytes) + 3 * Arena::pad((size_t)batch * 8)));
    Arena ar(ctx->ws);
    double *dH = ar.take<double>(nn * batch), *dU = ar.take<double>(nn * batch), *dl = ar.take<double>(batch);
    int *dinfo = ar.take<int>(batch), *dsel = ar.take<int>(batch);
    VSLAM_TRY(h2d(ctx, dH, H, bytes));
    VSLAM_TRY(h2d(ctx, dU, U, bytes));
    VSLAM_TRY(h2d(ctx, dl, shift, (size_t)batch * 8));
    if (sel) VSLAM_TRY(h2d(ctx, dsel, sel, (size_t)nsel * sizeof(int)));
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(dinfo, 0, (size_t)batch * sizeof(int), ctx->stream));
    PotrfFrom from{dH, nn, dl, 0.0};
    VSLAM_TRY(potrf_upper_batch_dev(ctx, dU, n, n, nn, sel ? nsel : batch, dinfo, sel ? dsel : nullptr, false, nullptr, 0, sel != nullptr, &from));
    VSLAM_TRY(d2h(ctx, U, dU, bytes));
    VSLAM_TRY(d2h(ctx, info, dinfo, (size_t)batch * sizeof(int)));
    return stream_finish(ctx);
}
