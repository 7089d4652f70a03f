// Panel / column-block factorisation of the blocked Householder QR (qr.hip) for the square-root-covariance
// re-triangularisations of the reference (Eigen::HouseholderQR at src/gaussian.hpp:121-124, :515-516,
// src/SLAM.cpp:810-819).
//
// A 32-column panel of a tall matrix is a chain of 32 dependent reflections, and with the rows spread over
// many workgroups every link of that chain costs a cross-workgroup exchange.  This kernel factors a panel
// with TWO exchanges instead of 32 and still returns the Householder reflectors of Eigen / LAPACK:
//
//   1. Cholesky QR, twice.  G = P^T P (MFMA on each workgroup's rows, all-reduced), R1 = chol(G),
//      Q1 = P R1^-1 (a triangular solve per row, local), then the same on Q1: G2 = Q1^T Q1, R2 = chol(G2).
//      G2 is the measured orthogonality of Q1: the pass is only trusted when |G2 - I| <= 1e-3 entrywise
//      (so ||G2 - I|| <= 0.032), which bounds cond(Q1) and makes Q = Q1 R2^-1 orthonormal to rounding.  P = Q (R2 R1).
//   2. Householder reconstruction (Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik 2015): the thin Q of
//      a Householder QR satisfies Q - [S; 0] = Y U with Y unit lower trapezoidal (the reflectors), U upper
//      triangular and S = diag(+-1) chosen during the elimination, S_k = -sign(pivot), so that every pivot has
//      magnitude >= 1; then T = -U S Y1^-T and R = S (R2 R1).  Here the elimination runs on
//      Q1_top - S R2 = Y1 (U R2) directly, which needs only the panel's top 32 rows (published by their owner
//      with the first exchange, so every workgroup repeats the 32 x 32 work and no third exchange is needed)
//      and gives Y2 = Q1 (U R2)^-1 by one more local triangular solve.
//      The pivot of step k is alpha_k / |x_k| of the k-th Householder step: when it is within 1e-7 of 0 (sign
//      of alpha not trustworthy) or within 1e-12 of +-1 (column tail zero or negligible: LAPACK keeps such a
//      column, tau = 0) the panel is handed to the column-by-column path below, which reproduces dlarfg
//      literally.  So do panels whose Cholesky factor breaks down or whose G2 fails the test.
//   3. Column-by-column path (rare): right-looking Householder with one all-reduce per column.
//
// Several panels can stay resident: a workgroup keeps its rows of up to 128 columns in LDS, factors the first
// panel, applies its reflectors to the resident columns to the right (W = V^T A2 by MFMA, one all-reduce,
// A2 -= V (T^T W) by MFMA), factors the next panel, and so on -- no kernel boundary and no pass over HBM
// between the panels of a column block.
//
// Exchanges are two-hop all-reduces (reduce-scatter, all-gather) so that every workgroup adds the partial sums
// of a slice in the same order and all of them continue from bit-identical data.  Payload words are stored and
// loaded at agent scope (sc1); each hop is signalled by one flag word per workgroup holding a sequence number
// that only grows during one factorisation; waits are bounded and raise the context's error word.
#include "qr_block.hpp"
#include <cstdlib>

namespace vslam {

typedef double double4_t __attribute__((ext_vector_type(4)));

// The workgroup's dynamic LDS.  The 32 x 32 routines below are real calls (inlined, their unrolled bodies
// cost the kernel its register allocation); they address LDS through this array by offset (in doubles), so
// the accesses stay DS instructions.
extern __shared__ __attribute__((aligned(16))) double qb_lds[];
#define QB_NOINLINE __attribute__((noinline))
// Keeps a value where the program computes it: without it the scheduler sinks the trailing updates of the
// factorisations below to their first use, which turns them into long dependent chains in front of every pivot
// and keeps every broadcast value alive until then.
#define QB_PIN(x) asm volatile("" : "+v"(x))
// Diagnostic build only (-DQB_STAMPS): thread 0 of workgroup 0 leaves shader-clock readings at the phase
// boundaries of a panel in the 32 spare doubles behind the panel's T (read back with vslam_debug_ws_read).
#ifdef QB_STAMPS
#define QB_STAMP(n) do { if (g == 0 && tid == 0) Tg[1024 + (n)] = (double)(long long)__builtin_readcyclecounter(); } while (0)
#else
#define QB_STAMP(n) do { } while (0)
#endif

constexpr int QB_LDM = 33;                  // row stride of the 32 x 32 work matrices in LDS
constexpr unsigned QB_SPIN_LIMIT = 1u << 21;

struct QbArgs {
    double *A; int lda; size_t strideA;     // element (j0, j0): first row / column of the block, matrix 0
    int mp;                                  // rows from j0 to the end of the matrix
    int bwc;                                 // columns of the block held in LDS
    int nfac;                                // columns to factor (<= bwc, <= mp)
    int rows_per, G, ldr, bw;                // rows per workgroup, workgroups per matrix, LDS column stride, LDS columns allocated
    unsigned seq0;                           // sequence numbers seq0+1 .. of this launch
    double *T; size_t strideT;               // compact-WY T of panel p of matrix b: T + b*strideT + p*QR_MSTRIDE (column-major, ld 32)
    double *V; int ldv;                      // optional: explicit reflectors (unit diagonal, zeros above), element (row j0, block column 0)
    int roff;                                // rows of this launch above the block's diagonal origin (they belong to the previous block)
    const double *Vprev; int ldvp;           // roff > 0: explicit reflectors of the previous block, element (first row of this launch, its column 0)
    const double *Tprev; int prev_np;        //           and its panels' T (QR_MSTRIDE apart); applied to the resident columns first
    int vprev_rows;                          //           rows of Vprev counted from the launch's first row (zero beyond: a banded matrix)
    double *xbuf; size_t xstride;            // exchange area per matrix: part [G][cmax], red [cmax], top [1024]
    int cmax;
    int *err;
};


// An exchanged value is its own ready mark: a double travels as two 64-bit words, each carrying 32 bits of the value and
// the 32-bit sequence number of the exchange it belongs to (stored and loaded as relaxed 64-bit atomics at agent scope,
// so a word is never torn).  A reader polls the two words until both carry the number it expects; no separate flag, no
// drain of the stores before a flag, one trip through memory per hop.  The buffers are zeroed before the first launch of
// a factorisation and the numbers only grow during it (the first one is 1), so a stale word never matches.
typedef unsigned long long qb_word;
struct Xchg {
    qb_word *part;                           // [G][cmax] values: what every workgroup contributes to the current sum
    qb_word *red;                            // [cmax] values: the sums, each written by the workgroup that owns its slice
    qb_word *top;                            // [1024] values: the panel's top rows, published by their owner
    int *err;
    int G, g, cmax;
    unsigned seq;                            // number of the last exchange entered
    int *dead;                               // LDS word: a wait gave up; every later wait returns at once
    double *scratch;                         // LDS, 256 doubles
};

__device__ __forceinline__ void qb_ll_store(qb_word *item, double v, unsigned seq)
{
    const qb_word bits = (qb_word)__double_as_longlong(v), tag = (qb_word)seq << 32;
    __hip_atomic_store(item, (bits & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(item + 1, (bits >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool qb_ll_try(const qb_word *item, unsigned seq, double *v)
{
    const qb_word w0 = __hip_atomic_load(item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const qb_word w1 = __hip_atomic_load(item + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *v = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
    return (unsigned)(w0 >> 32) == seq && (unsigned)(w1 >> 32) == seq;
}
// Polls one value until it belongs to exchange seq.  Bounded: a wait that gives up raises the context's error word
// and marks the workgroup dead, after which every wait returns at once (with garbage; the call fails as a whole).
__device__ __forceinline__ double qb_ll_wait(const Xchg &x, const qb_word *item, unsigned seq)
{
    double v;
    if (qb_ll_try(item, seq, &v)) return v;
    unsigned spins = 0;
    while (*reinterpret_cast<volatile int *>(x.dead) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (qb_ll_try(item, seq, &v)) return v;
        if ((++spins & 63u) == 0) {
            if (spins > QB_SPIN_LIMIT) atomicExch(x.err, 1);
            if (__hip_atomic_load(x.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) *x.dead = 1;
        }
    }
    return v;
}

// This workgroup's contribution idx to the NEXT exchange (the one the following qb_allreduce enters).  Every workgroup
// must contribute every index below the count it passes to qb_allreduce.
__device__ __forceinline__ void qb_put(const Xchg &x, int idx, double v)
{
    qb_word *dst = (x.G == 1) ? x.red + 2 * (size_t)idx : x.part + 2 * ((size_t)x.g * x.cmax + idx);
    qb_ll_store(dst, v, x.seq + 1);
}
// Entry idx of the panel's top rows, for the next exchange (owner only) / as published with the last exchange entered.
__device__ __forceinline__ void qb_put_top(const Xchg &x, int idx, double v) { qb_ll_store(x.top + 2 * (size_t)idx, v, x.seq + 1); }
__device__ __forceinline__ double qb_top(const Xchg &x, int idx) { return qb_ll_wait(x, x.top + 2 * (size_t)idx, x.seq); }
// Sum idx of the last exchange entered.  Every workgroup reads every index below the count of that exchange before it
// contributes to the next one (that is what allows the buffers to be reused without a second set).
__device__ __forceinline__ double qb_red(const Xchg &x, int idx) { return qb_ll_wait(x, x.red + 2 * (size_t)idx, x.seq); }

// Enters the next exchange: sums over the workgroups of one matrix the `cnt` values each of them has contributed with
// qb_put; afterwards qb_red(x, idx) returns the sums.  Two hops (reduce-scatter, all-gather): this workgroup adds up its
// slice of the values in a fixed order, so every workgroup reads identical bits, and publishes it; the gather is the
// readers' polling.  All threads of the workgroup call this.
__device__ __forceinline__ void qb_allreduce(Xchg &x, int cnt)
{
    const int tid = threadIdx.x;
    ++x.seq;
    if (x.G == 1) return;                                        // the contributions went straight to the sums
    const int per = (((cnt + x.G - 1) / x.G) + 1) & ~1;
    const int e0 = x.g * per, e1 = min(cnt, e0 + per), len = e1 - e0;
    __syncthreads();                                             // scratch is free again
    if (per <= 128) {
        // several threads per element, each adding a fixed subset of the workgroups in order; the subsets are
        // then added in order: every load of the slice is in flight at once
        const int ngrp = min(8, QB_T / per), el = tid % per, grp = tid / per;
        if (grp < ngrp && el < len) {
            double s = 0.0;
            int q = grp;
            for (; q + 7 * ngrp < x.G; q += 8 * ngrp) {
                double v[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ok[u] = qb_ll_try(x.part + 2 * ((size_t)(q + u * ngrp) * x.cmax + e0 + el), x.seq, &v[u]);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (!ok[u]) v[u] = qb_ll_wait(x, x.part + 2 * ((size_t)(q + u * ngrp) * x.cmax + e0 + el), x.seq);
                    s += v[u];
                }
            }
            for (; q < x.G; q += ngrp) s += qb_ll_wait(x, x.part + 2 * ((size_t)q * x.cmax + e0 + el), x.seq);
            x.scratch[grp * per + el] = s;
        }
        __syncthreads();
        if (tid < len) {
            double s = 0.0;
            for (int grp2 = 0; grp2 < ngrp; ++grp2) s += x.scratch[grp2 * per + tid];
            qb_ll_store(x.red + 2 * (size_t)(e0 + tid), s, x.seq);
        }
    } else {
        for (int e = e0 + tid; e < e1; e += QB_T) {
            double s = 0.0;
            for (int q = 0; q < x.G; ++q) s += qb_ll_wait(x, x.part + 2 * ((size_t)q * x.cmax + e), x.seq);
            qb_ll_store(x.red + 2 * (size_t)e, s, x.seq);
        }
    }
}

__device__ __forceinline__ double rsqrt_f64(double d)
{
    double y = __builtin_amdgcn_rsq(d);                    // about 2^-26 relative; two Newton steps would be one more than fp64 resolves
    const double h = 0.5 * y, e = fma(-d * y, y, 1.0);     // y <- y + y e/2 (1 + 3e/4): third order, one pass
    y = fma(h * e, fma(0.75, e, 1.0), y);
    return y;
}
__device__ __forceinline__ double rcp_f64(double p)
{
    double y = __builtin_amdgcn_rcp(p);
    const double e = fma(-p, y, 1.0);                      // y <- y (1 + e + e^2): third order, one pass
    y = fma(y * e, 1.0 + e, y);
    return y;
}

// The 32 x 32 routines below work on full 32 x 32 matrices: a narrower panel is padded by the caller (identity
// in the triangular factors and Gram matrices, 0.5 on the diagonal of the top rows), so their loops carry no
// bounds and unroll into straight-line code.
//
// One wave: upper Cholesky factor of M (LDS, row-major, ld QB_LDM), in place, strict lower part cleared;
// invd[k] = 1 / R_kk.  Lane j owns column j in registers.  Row k goes to LDS as soon as it is final and comes
// back as broadcast reads for the updates it drives; those are applied one step late, under the next pivot's
// reciprocal square root, except for the entry the next pivot row needs (one v_readlane pair).
// False when a pivot is not safely positive (then M holds garbage).
__device__ QB_NOINLINE bool qb_chol32(int oM, int oinvd)
{
    double *M = qb_lds + oM, *invd = qb_lds + oinvd;
    const int lane = threadIdx.x & 63, j = lane & 31;
    double r[32], bc[32];                                       // bc: row k-1 as broadcast values, in flight during the pivot work
#pragma unroll
    for (int i = 0; i < 32; ++i) { r[i] = M[i * QB_LDM + j]; bc[i] = 0.0; }
    const double dj0 = M[j * QB_LDM + j];
    bool ok = true;
    double prev = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k >= 1) {
#pragma unroll
            for (int i = k + 1; i < 32; ++i) bc[i] = M[(k - 1) * QB_LDM + i];
        }
        asm volatile("" ::: "memory");
        const double d = readlane_f64(r[k], k), d0 = readlane_f64(dj0, k);
        ok = ok && (d > d0 * 1e-13) && (d0 > 0.0);
        const double inv = rsqrt_f64(d);
        const double rk = (j >= k) ? r[k] * inv : 0.0;          // R_kj
        if (lane < 32) M[k * QB_LDM + j] = rk;
        if (lane == k) invd[k] = inv;
        if (k >= 1) {
#pragma unroll
            for (int i = k + 1; i < 32; ++i) { r[i] -= bc[i] * prev; QB_PIN(r[i]); }
        }
        if (k + 1 < 32) { r[k + 1] -= readlane_f64(rk, k + 1) * rk; QB_PIN(r[k + 1]); }
        prev = rk;
        asm volatile("" ::: "memory");
    }
    return ok;
}

// x (1 x 32) <- x R^-1, R upper triangular in LDS (row-major, ld QB_LDM), invd = reciprocals of its diagonal.
// Row k+1 of R is fetched (broadcast reads) while row k is applied.
__device__ __forceinline__ void qb_trsm_row(double (&x)[32], int oR, int oinvd)
{
    const double *R = qb_lds + oR, *invd = qb_lds + oinvd;
    double cur[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) cur[j] = R[j];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        x[k] *= invd[k];
        // eight columns at a time: apply row k, then refill those registers with row k+1
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
            if (8 * c8 + 7 > k) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (8 * c8 + u > k) { x[8 * c8 + u] -= x[k] * cur[8 * c8 + u]; QB_PIN(x[8 * c8 + u]); }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (8 * c8 + u > k + 1) cur[8 * c8 + u] = R[((k + 1) & 31) * QB_LDM + 8 * c8 + u];
                asm volatile("" ::: "memory");
            }
        }
    }
}

// Neighbouring lanes 2q, 2q+1 work as a pair: value of the pair's lane ODD for both / the partner's value (DPP quad_perm).
template <int ODD>
__device__ __forceinline__ double qb_pair_bcast(double v)
{
    constexpr int ctrl = ODD ? 0xF5 : 0xA0;                   // quad_perm [1,1,3,3] / [0,0,2,2]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double qb_pair_swap(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// The same row solve shared by a pair of lanes: lane h of the pair holds the entries j = 2 t + h.  Every entry goes
// through the operations of qb_trsm_row in the same order (same bits); the pair halves the multiply-adds and the
// broadcast reads per lane, the finished entry x_k crosses the pair by DPP.
__device__ __forceinline__ void qb_trsm_row2(double (&x)[16], int h, int oR, int oinvd)
{
    const double *R = qb_lds + oR + h, *invd = qb_lds + oinvd;
    double cur[16];                                             // row k of R at this lane's columns; refilled with row k+1 as it is used up
#pragma unroll
    for (int t = 0; t < 16; ++t) cur[t] = R[2 * t];
    double inv = invd[0];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int tk = k >> 1;
        const double own = x[tk] * inv;
        const double xk = (k & 1) ? qb_pair_bcast<1>(own) : qb_pair_bcast<0>(own);
        x[tk] = (h == (k & 1)) ? xk : x[tk];
        if (k + 1 < 32) inv = invd[k + 1];
        // four of this lane's columns at a time: apply row k, then refill those registers with row k+1
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            if (4 * c4 + 3 >= tk) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = 4 * c4 + u;
                    if (t > tk) { x[t] -= cur[t] * xk; QB_PIN(x[t]); }
                    else if (t == tk && (k & 1) == 0) { x[t] -= (h ? cur[t] : 0.0) * xk; QB_PIN(x[t]); }     // j = k + 1 belongs to the odd lane
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = 4 * c4 + u;
                    if (k + 1 < 32 && t >= ((k + 1) >> 1)) cur[t] = R[(k + 1) * QB_LDM + 2 * t];
                }
                asm volatile("" ::: "memory");
            }
        }
    }
}

// Rows first_row + (t >> 1) + (nthreads / 2) * pass < row_end (except those in [skip_lo, skip_hi)) of the nb-column matrix
// at qb_lds[oB + i * rs + j * cs]:  row <- row R^-1, two lanes per row; t = this thread's number among the nthreads
// (a multiple of 64) that share the work.
__device__ QB_NOINLINE void qb_trsm_rows2(int oB, int rs, int cs, int first_row, int row_end, int nthreads, int t, int skip_lo, int skip_hi,
                                          int oR, int oinvd, int nb)
{
    const int h = t & 1;
    for (int i = first_row + (t >> 1); i < row_end; i += nthreads >> 1) {
        if (i >= skip_lo && i < skip_hi) continue;
        double *row = qb_lds + oB + (size_t)i * rs;
        double xr[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) xr[q] = (2 * q + h < nb) ? row[(size_t)(2 * q + h) * cs] : 0.0;
        qb_trsm_row2(xr, h, oR, oinvd);
#pragma unroll
        for (int q = 0; q < 16; ++q) if (2 * q + h < nb) row[(size_t)(2 * q + h) * cs] = xr[q];
    }
}

// value of position R in this lane's row of 16 lanes, for every lane of the row (DPP row_share): a broadcast that
// stays in vector registers
template <int R>
__device__ __forceinline__ double qb_rowshare(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + R, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + R, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Step K of the signed elimination of a 16 x 16 block held by columns (lane c of every row of 16 lanes owns column
// c: a[r] = row r, r2c[r] = R2(r, c) or 0 below the diagonal).
// Lanes left of the pivot column keep their entries (their multiplicand is zero); the pivot lane's own entries
// below the diagonal stay unscaled until the end (qb_lu16 multiplies them by the lane's pivot reciprocal, myinv).
template <int K>
__device__ __forceinline__ void qb_lu16_step(double (&a)[16], const double (&r2c)[16], int c, int lane, double *invw_o, double *sgn_o, bool &ok, double &myinv)
{
    const double e = qb_rowshare<K>(a[K]), rkk = qb_rowshare<K>(r2c[K]);
    const double s = (e >= 0.0) ? -1.0 : 1.0, ae = fabs(e);
    ok = ok && (ae > 1e-7 * rkk) && (rkk - ae > 1e-12 * rkk);
    a[K] -= s * r2c[K];                                         // row K of Q - S R2 (nothing left of the diagonal: R2 is upper triangular)
    const double inv = rcp_f64(e - s * rkk);
    if (lane == K) { invw_o[K] = inv; sgn_o[K] = s; }
    myinv = (c == K) ? inv : myinv;
    if constexpr (K < 15) {
        const double akm = (c > K) ? a[K] : 0.0;
#define QB_LU16_ROW(M) if constexpr ((M) > K) { a[M] -= (qb_rowshare<K>(a[M]) * inv) * akm; QB_PIN(a[M]); }
        QB_LU16_ROW(1) QB_LU16_ROW(2) QB_LU16_ROW(3) QB_LU16_ROW(4) QB_LU16_ROW(5) QB_LU16_ROW(6) QB_LU16_ROW(7) QB_LU16_ROW(8)
        QB_LU16_ROW(9) QB_LU16_ROW(10) QB_LU16_ROW(11) QB_LU16_ROW(12) QB_LU16_ROW(13) QB_LU16_ROW(14) QB_LU16_ROW(15)
#undef QB_LU16_ROW
    }
}

// One wave: the signed elimination of the 16 x 16 block at (o, o) of Q (LDS) against the same block of R2; multipliers
// below the diagonal and W on and above it go back to Q, invw / sgn for the 16 pivots.
__device__ __forceinline__ bool qb_lu16(double *Q, const double *R2, int o, double *invw, double *sgn)
{
    const int lane = threadIdx.x & 63, c = lane & 15;
    double a[16], r2c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a[r] = Q[(o + r) * QB_LDM + o + c]; r2c[r] = (r <= c) ? R2[(o + r) * QB_LDM + o + c] : 0.0; }
    bool ok = true;
    double myinv = 0.0;
    qb_lu16_step<0>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv); qb_lu16_step<1>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv);
    qb_lu16_step<2>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv); qb_lu16_step<3>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv);
    qb_lu16_step<4>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv); qb_lu16_step<5>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv);
    qb_lu16_step<6>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv); qb_lu16_step<7>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv);
    qb_lu16_step<8>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv); qb_lu16_step<9>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv);
    qb_lu16_step<10>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv); qb_lu16_step<11>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv);
    qb_lu16_step<12>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv); qb_lu16_step<13>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv);
    qb_lu16_step<14>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv); qb_lu16_step<15>(a, r2c, c, lane, invw + o, sgn + o, ok, myinv);
    if (lane < 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Q[(o + r) * QB_LDM + o + c] = (r > c) ? a[r] * myinv : a[r];
    }
    return ok;
}

// The signed elimination of the reconstruction by the whole workgroup, in two steps of 16 pivots.
// In: Q (Q1_top, LDS), R2 (upper, LDS).  Finds S = diag(+-1) and Q - S R2 = Y1 W with Y1 unit lower, W upper (= U R2);
// Q is overwritten with Y1 (strictly lower part) and W (upper part), invw[k] = 1 / W_kk, sgn[k] = S_k.
//   wave 0: the leading 16 x 16 block in registers, the pivot row and the multipliers handed round by DPP row_share;
//   wave 1: W12 = L11^-1 (Q12 - S1 R2_12), a lane per column;   wave 2: L21 = Q21 W11^-1, a lane per row;
//   wave 0: Q22 -= L21 W12 on the matrix cores, then the trailing block like the leading one.
// Every wave of the workgroup calls this (three barriers inside).  *verdict (LDS) = 0 when a pivot ratio
// alpha_k / |x_k| is too close to 0 or to +-1 to be decided here.
__device__ QB_NOINLINE void qb_lu_signed_wg(int oQ, int oR2, int oinvw, int osgn, int *verdict)
{
    double *Q = qb_lds + oQ, *invw = qb_lds + oinvw, *sgn = qb_lds + osgn;
    const double *R2 = qb_lds + oR2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kr = lane >> 4;
    bool ok = true;
    if (wave == 0) ok = qb_lu16(Q, R2, 0, invw, sgn);
    __syncthreads();
    if (wave == 1) {
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = Q[r * QB_LDM + 16 + c] - sgn[r] * R2[r * QB_LDM + 16 + c];
#pragma unroll
        for (int k = 0; k < 15; ++k) {
#pragma unroll
            for (int m2 = k + 1; m2 < 16; ++m2) { x[m2] -= Q[m2 * QB_LDM + k] * x[k]; QB_PIN(x[m2]); }
            asm volatile("" ::: "memory");
        }
        if (lane < 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Q[r * QB_LDM + 16 + c] = x[r];
        }
    } else if (wave == 2) {
        double yv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) yv[k] = Q[(16 + c) * QB_LDM + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            yv[k] *= invw[k];
#pragma unroll
            for (int j = k + 1; j < 16; ++j) { yv[j] -= Q[k * QB_LDM + j] * yv[k]; QB_PIN(yv[j]); }
            asm volatile("" ::: "memory");
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) Q[(16 + c) * QB_LDM + k] = yv[k];
        }
    }
    __syncthreads();
    if (wave == 0) {
        double4_t t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = Q[(16 + 4 * r + kr) * QB_LDM + 16 + c];
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4)
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(-Q[(16 + c) * QB_LDM + k0 + kr], Q[(k0 + kr) * QB_LDM + 16 + c], t, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Q[(16 + 4 * r + kr) * QB_LDM + 16 + c] = t[r];
        const bool ok2 = qb_lu16(Q, R2, 16, invw, sgn);
        if (lane == 0) *verdict = (ok && ok2) ? 1 : 0;
    }
    __syncthreads();
}


// Rows i = row_begin, row_begin + row_step, ... < row_end (except those in [skip_lo, skip_hi)) of the nb-column
// matrix at qb_lds[oB + i * rs + j * cs]:  row <- row R^-1.
__device__ QB_NOINLINE void qb_trsm_rows(int oB, int rs, int cs, int row_begin, int row_end, int row_step, int skip_lo, int skip_hi,
                                         int oR, int oinvd, int nb)
{
    for (int i = row_begin; i < row_end; i += row_step) {
        if (i >= skip_lo && i < skip_hi) continue;
        double *row = qb_lds + oB + (size_t)i * rs;
        double xr[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) xr[j] = (j < nb) ? row[(size_t)j * cs] : 0.0;
        qb_trsm_row(xr, oR, oinvd);
#pragma unroll
        for (int j = 0; j < 32; ++j) if (j < nb) row[(size_t)j * cs] = xr[j];
    }
}

// One wave: the upper triangle of R2 R1 (both upper, zero below the diagonal) to qb_lds[oDst + j * cs + i] for
// i <= j < nb, on the matrix cores: the three 16 x 16 tiles on and above the diagonal, K = 32 each.
__device__ QB_NOINLINE void qb_rr_upper(int oR2, int oR1, int oDst, int cs, int nb)
{
    const int lane = threadIdx.x & 63, l15 = lane & 15, kr = lane >> 4;
    const double *R2 = qb_lds + oR2, *R1 = qb_lds + oR1;
    double4_t t[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k0 = 0; k0 < 32; k0 += 4) {
        const double a0 = R2[l15 * QB_LDM + k0 + kr], a1 = R2[(16 + l15) * QB_LDM + k0 + kr];
        const double b0 = R1[(k0 + kr) * QB_LDM + l15], b1 = R1[(k0 + kr) * QB_LDM + 16 + l15];
        t[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, t[0], 0, 0, 0);        // tile (0, 0)
        t[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, t[1], 0, 0, 0);        // tile (0, 1)
        t[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, t[2], 0, 0, 0);        // tile (1, 1)
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ta = q == 2, tb = q >= 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ta + 4 * r + kr, j = 16 * tb + l15;
            if (i <= j && j < nb) qb_lds[oDst + (size_t)j * cs + i] = t[q][r];
        }
    }
}


// One wave: T^T = -Y1^-1 S U^T with U = W R2^-1 (W = U R2 and Y1 as left by qb_lu_signed in oQ), written to oT
// as a full 32 x 32 lower-triangular matrix (row-major, zeros elsewhere and beyond nb).  The lane pair (2 j, 2 j + 1)
// owns column j of T^T, lane h of the pair the rows i = 2 t + h.
__device__ QB_NOINLINE void qb_make_tt(int oQ, int oR2, int oinvd2, int osgn, int oT, int nb)
{
    const int lane = threadIdx.x & 63, j = lane >> 1, h = lane & 1;
    const double *Q = qb_lds + oQ, *sgn = qb_lds + osgn;
    double tt[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) tt[t] = (2 * t + h >= j) ? Q[j * QB_LDM + 2 * t + h] : 0.0;          // row j of W
    qb_trsm_row2(tt, h, oR2, oinvd2);                                                                // row j of U
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int i = 2 * t + h;
        tt[t] = (i >= j && i < nb && j < nb) ? -sgn[i] * tt[t] : 0.0;                                 // -(S U^T)(i, j)
    }
    // forward substitution with the unit lower Y1: each lane sums over its own rows, the pair adds up
#pragma unroll
    for (int i = 1; i < 32; ++i) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int t = 0; 2 * t < i; ++t) {
            const bool in = 2 * t + h < i;                                                           // only the last term of the odd lane can fall out
            const double yv = in ? Q[i * QB_LDM + min(2 * t + h, 31)] : 0.0;
            if (t & 1) s1 += yv * tt[t];
            else s0 += yv * tt[t];
        }
        const double part = s0 + s1, sum = part + qb_pair_swap(part);
        if (h == (i & 1)) tt[i >> 1] -= sum;
        QB_PIN(tt[i >> 1]);
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) qb_lds[oT + (2 * t + h) * QB_LDM + j] = tt[t];
}

// Value of the explicit reflector matrix at local row i, panel column a (block column c0 + a): zero above the
// unit diagonal, the stored tail below.
__device__ __forceinline__ double qb_yv(const double *Bk, int ldr, int r0, int c0, int nb, int i, int a)
{
    const int d = r0 + i - (c0 + a);
    if (a >= nb || d < 0) return 0.0;
    return d == 0 ? 1.0 : Bk[(size_t)(c0 + a) * ldr + i];
}

// Wave `wave` adds to the exchange buffer the 16 x 16 tile (wave >> 1, wave & 1) of  X^T Z  over the local rows
// [lo, rows_pad): X = panel columns c0 .. c0+nb (UNIT: as explicit reflectors), Z = columns z0 .. z0+zn.
template <bool UNIT>
__device__ __forceinline__ void qb_xtz_tile(const double *Bk, int ldr, int r0, int lo, int rows_pad, int c0, int nb, int z0, int zn,
                                            const Xchg &x, int dst /* first index of the [32][32] row-major block in the exchange */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ta = wave >> 1, tb = wave & 1;
    const int a = 16 * ta + (lane & 15), zc = 16 * tb + (lane & 15), kr = lane >> 4;
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    if (16 * ta < nb && 16 * tb < zn) {
        double4_t ac[4] = {acc, acc, acc, acc};                 // four chains: a dependent fp64 MFMA waits for the one before it
        // Branch-free operand reads: every LDS address is valid (columns clamped; rows up to the next multiple of
        // 16, zero beyond the matrix), masks are applied to the values.  Sixteen rows per trip: the eight reads
        // of a trip are issued before its four MFMAs.
        const double *xp = Bk + (size_t)(c0 + min(a, nb - 1)) * ldr, *zp = Bk + (size_t)(z0 + min(zc, zn - 1)) * ldr;
        const bool xok = a < nb, zok = zc < zn;
        const int dgl = c0 + a - r0;                            // local row of this reflector column's unit diagonal (r0 counts from the block's diagonal origin)
        const int rows16 = (rows_pad + 15) & ~15;
        for (int k0 = lo & ~15; k0 < rows16; k0 += 16) {
            double xv[4], zv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { xv[u] = xp[k0 + 4 * u + kr]; zv[u] = zp[k0 + 4 * u + kr]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = k0 + 4 * u + kr;
                if (UNIT) xv[u] = (!xok || i < dgl) ? 0.0 : (i == dgl ? 1.0 : xv[u]);
                else xv[u] = (xok && i >= lo) ? xv[u] : 0.0;
                zv[u] = (zok && i >= lo) ? zv[u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ac[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u], zv[u], ac[u], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (ac[0][r] + ac[1][r]) + (ac[2][r] + ac[3][r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) qb_put(x, dst + (16 * ta + 4 * r + kr) * 32 + zc, acc[r]);
}

__global__ __launch_bounds__(QB_T) void qr_block_kernel(QbArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, b = blockIdx.y, ldr = a.ldr;
    double *Bk = qb_lds;                                   // [bw][ldr]: column-major rows of this workgroup
    double *M0 = Bk + (size_t)a.bw * ldr;                  // G -> R1, later T^T
    double *M1 = M0 + 32 * QB_LDM;                         // G2 -> R2, later W / T^T W
    double *M2 = M1 + 32 * QB_LDM;                         // top rows of the panel -> Q1_top -> Y1 \ U R2
    double *invd1 = M2 + 32 * QB_LDM, *invd2 = invd1 + 32, *invw = invd2 + 32, *sgn = invw + 32, *taus = sgn + 32;
    double *red8 = taus + 32;                              // [8][32]
    double *tot = red8 + 256, *rc = tot + 32, *wv = rc + 32, *sc = wv + 32;
    int *iflag = reinterpret_cast<int *>(sc + 8);          // [0] dead, [1] verdict of a step
    const int oM0 = (int)(M0 - qb_lds), oM1 = (int)(M1 - qb_lds), oM2 = (int)(M2 - qb_lds);
    const int oinvd1 = (int)(invd1 - qb_lds), oinvd2 = (int)(invd2 - qb_lds), oinvw = (int)(invw - qb_lds), osgn = (int)(sgn - qb_lds);

    double *A = a.A + (size_t)b * a.strideA;
    const int r0 = g * a.rows_per;                         // first row of this workgroup, from the launch's first row
    const int rb = r0 - a.roff;                            // the same from the block's diagonal origin (negative: rows of the previous block)
    const int nr = min(a.rows_per, a.mp - r0);
    const int rows_pad = (nr + 3) & ~3;
    Xchg x;
    {
        qb_word *base = reinterpret_cast<qb_word *>(a.xbuf + (size_t)b * a.xstride);
        x.part = base;
        x.red = base + 2 * (size_t)a.G * a.cmax;
        x.top = base + 2 * (size_t)(a.G + 1) * a.cmax;
        x.err = a.err; x.G = a.G; x.g = g; x.cmax = a.cmax; x.seq = a.seq0; x.dead = iflag; x.scratch = red8;
    }
    if (tid == 0) { iflag[0] = 0; iflag[1] = 0; }
    for (int c = wave; c < a.bwc; c += QB_T / 64)
        for (int i = lane; i < a.rows_per; i += 64) Bk[(size_t)c * ldr + i] = (i < nr) ? A[(size_t)c * a.lda + r0 + i] : 0.0;
    __syncthreads();

    // ---- the previous block's reflectors, panel by panel, on every resident column (all rows of the launch) ----
    if (a.roff > 0) {
        const int kr = lane >> 4, l15 = lane & 15;
        const int nchunk = (a.bwc + 31) / 32;
        for (int pp = 0; pp < a.prev_np; ++pp) {
            const double *Vp = a.Vprev + (size_t)(32 * pp) * a.ldvp + r0;      // this workgroup's rows of the panel's reflectors
            // W = Vp^T C: 32 x bwc, wave w owns column tiles 2w and 2w+1 of each 64-column half
            for (int half = 0; half < (a.bwc + 63) / 64; ++half) {
                const int zc0 = 64 * half + 32 * (wave >> 1) + 16 * (wave & 1);       // 16 columns
                double4_t acc0 = (double4_t){0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
                if (zc0 < a.bwc) {
                    for (int k0 = 0; k0 < rows_pad; k0 += 8) {
                        double xv[2][2], zv[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int i = k0 + 4 * u + kr;
                            const bool in = i < nr, inv = in && r0 + i < a.vprev_rows;
                            xv[u][0] = inv ? Vp[(size_t)l15 * a.ldvp + i] : 0.0;
                            xv[u][1] = inv ? Vp[(size_t)(16 + l15) * a.ldvp + i] : 0.0;
                            zv[u] = (in && zc0 + l15 < a.bwc) ? Bk[(size_t)(zc0 + l15) * ldr + i] : 0.0;
                        }
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[0][0], zv[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[0][1], zv[0], acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[1][0], zv[1], acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[1][1], zv[1], acc3, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { acc0[r] += acc2[r]; acc1[r] += acc3[r]; }
                }
                // chunk ch = zc0 / 32 holds W[a][c] at ch * 1024 + a * 32 + (c - 32 ch)
                const int dst = 1024 * (zc0 >> 5) + (zc0 & 31) + l15;
                if ((zc0 >> 5) < nchunk)                             // every index of the exchange is contributed (zeros beyond the block)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qb_put(x, dst + (4 * r + kr) * 32, acc0[r]);
                    qb_put(x, dst + (16 + 4 * r + kr) * 32, acc1[r]);
                }
            }
            qb_allreduce(x, 1024 * nchunk);
            for (int e = tid; e < 1024; e += QB_T) M0[(e >> 5) * QB_LDM + (e & 31)] = a.Tprev[(size_t)pp * QR_MSTRIDE + e];   // T^T: [a][b] = T(b, a)
            const int ta = wave >> 1, tb = wave & 1;
            for (int ch = 0; ch < nchunk; ++ch) {
                const int z0 = 32 * ch, zn = min(32, a.bwc - z0);
                __syncthreads();
                for (int e = tid; e < 1024; e += QB_T) M1[(e >> 5) * QB_LDM + (e & 31)] = qb_red(x, 1024 * ch + e);
                __syncthreads();
                double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k0 = 0; k0 < 32; k0 += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(M0[(16 * ta + l15) * QB_LDM + k0 + kr], M1[(k0 + kr) * QB_LDM + 16 * tb + l15], acc, 0, 0, 0);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) M1[(16 * ta + 4 * r + kr) * QB_LDM + 16 * tb + l15] = acc[r];
                __syncthreads();
                for (int rt = wave; rt < (nr + 15) >> 4; rt += QB_T / 64) {
                    const int i = 16 * rt + l15;
                    double vq[8];
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) vq[q8] = (i < nr && r0 + i < a.vprev_rows) ? Vp[(size_t)(4 * q8 + kr) * a.ldvp + i] : 0.0;
                    double4_t u[2][2];
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) { u[cb][0] = (double4_t){0.0, 0.0, 0.0, 0.0}; u[cb][1] = u[cb][0]; }
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
                            u[cb][q8 & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(M1[(4 * q8 + kr) * QB_LDM + 16 * cb + l15], vq[q8], u[cb][q8 & 1], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        if (16 * cb >= zn) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int zc = 16 * cb + 4 * r + kr;
                            if (zc < zn && i < nr) Bk[(size_t)(z0 + zc) * ldr + i] -= u[cb][0][r] + u[cb][1][r];
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    const int npan = (a.nfac + 31) / 32;
    for (int p = 0; p < npan; ++p) {
        const int c0 = 32 * p, nb = min(32, a.nfac - c0);
        const int lo = max(c0 - rb, 0);                    // first local row inside the panel (>= nr: none)
        const int gtop = (c0 + a.roff) / a.rows_per, toploc = c0 + a.roff - gtop * a.rows_per;
        const bool owner = (g == gtop);
        double *Tg = a.T + (size_t)b * a.strideT + (size_t)p * QR_MSTRIDE;

        // the panel as it stands goes back to memory: the column-by-column path restarts from it
        if (p > 0 || a.roff > 0) {
            for (int c = wave; c < nb; c += QB_T / 64)
                for (int i = lane; i < nr; i += 64) A[(size_t)(c0 + c) * a.lda + r0 + i] = Bk[(size_t)(c0 + c) * ldr + i];
        }

        bool fast = true;
        QB_STAMP(0);
        // ---- Gram matrix of the panel, the panel's top rows ----
        qb_xtz_tile<false>(Bk, ldr, rb, lo, rows_pad, c0, nb, c0, nb, x, 0);
        if (owner)
            for (int e = tid; e < 1024; e += QB_T) {
                const int i = e >> 5, j = e & 31;
                qb_put_top(x, e, (i < nb && j < nb) ? Bk[(size_t)(c0 + j) * ldr + toploc + i] : 0.0);
            }
        QB_STAMP(1);
        qb_allreduce(x, 1024);
        QB_STAMP(2);
        for (int e = tid; e < 1024; e += QB_T) {
            const int i = e >> 5, j = e & 31;
            const bool in = i < nb && j < nb;                    // a narrower panel: padded for the 32 x 32 routines
            const double gv = qb_red(x, e), tv = qb_top(x, e);
            M0[i * QB_LDM + j] = in ? gv : (i == j ? 1.0 : 0.0);
            M2[i * QB_LDM + j] = in ? tv : (i == j ? 0.5 : 0.0);
        }
        __syncthreads();
        if (wave == 0) { const bool ok = qb_chol32(oM0, oinvd1); if (lane == 0) iflag[1] = ok ? 1 : 0; }
        __syncthreads();
        fast = iflag[1] != 0;
        QB_STAMP(3);
        if (fast) {
            // ---- Q1 = P R1^-1 on the local rows and on the copy of the top rows ----
            // few rows per workgroup: two lanes per row on three waves, the copy of the top rows on the fourth
            if (nr - lo <= 96) { if (wave < 3) qb_trsm_rows2(c0 * ldr, 1, ldr, lo, nr, 192, tid, 0, 0, oM0, oinvd1, nb); }
            else qb_trsm_rows(c0 * ldr, 1, ldr, lo + tid, nr, QB_T, 0, 0, oM0, oinvd1, nb);
            if (wave == 3) qb_trsm_rows2(oM2, QB_LDM, 1, 0, 32, 64, lane, 0, 0, oM0, oinvd1, 32);
            __syncthreads();
            QB_STAMP(4);
            // ---- G2 = Q1^T Q1: how orthonormal Q1 came out ----
            qb_xtz_tile<false>(Bk, ldr, rb, lo, rows_pad, c0, nb, c0, nb, x, 0);
            QB_STAMP(5);
            qb_allreduce(x, 1024);
            QB_STAMP(6);
            if (tid == 0) iflag[1] = 1;
            __syncthreads();
            {
                bool bad = false;
                for (int e = tid; e < 1024; e += QB_T) {
                    const int i = e >> 5, j = e & 31;
                    const double v = qb_red(x, e);
                    const bool in = i < nb && j < nb;
                    M1[i * QB_LDM + j] = in ? v : (i == j ? 1.0 : 0.0);
                    if (in && !(fabs(v - (i == j ? 1.0 : 0.0)) <= 1e-3)) bad = true;
                }
                if (bad) iflag[1] = 0;
            }
            __syncthreads();
            fast = iflag[1] != 0;
        }
        if (fast) {
            __syncthreads();
            QB_STAMP(7);
            if (wave == 0) { const bool ok = qb_chol32(oM1, oinvd2); if (lane == 0) iflag[1] = ok ? 1 : 0; }
            __syncthreads();
            fast = iflag[1] != 0;
            QB_STAMP(8);
        }
        if (fast) {
            // the signed elimination (waves 0 to 2); wave 3 of the owner first: R2 R1 into the top rows (their Q1 is
            // no longer needed: M2 holds the copy), signs applied afterwards
            __syncthreads();
            if (wave == 3 && owner) qb_rr_upper(oM1, oM0, c0 * ldr + toploc, ldr, nb);
            qb_lu_signed_wg(oM2, oM1, oinvw, osgn, &iflag[1]);
            fast = iflag[1] != 0;
            QB_STAMP(9);
        }
        if (fast) {
            // ---- reflectors below the top rows: Y2 = Q1 (U R2)^-1; wave 2: T^T; wave 3 of the owner: finish the top rows ----
            if (nr - lo <= 64) { if (wave < 2) qb_trsm_rows2(c0 * ldr, 1, ldr, lo, nr, 128, tid, c0 - rb, c0 + nb - rb, oM2, oinvw, nb); }
            else qb_trsm_rows(c0 * ldr, 1, ldr, lo + tid, nr, QB_T, c0 - rb, c0 + nb - rb, oM2, oinvw, nb);
            if (wave == 2) qb_make_tt(oM2, oM1, oinvd2, osgn, oM0, nb);
            if (wave == 3 && owner)
                for (int e = lane; e < 1024; e += 64) {
                    const int i = e >> 5, j = e & 31;
                    if (i < nb && j < nb) {
                        double *dst = Bk + (size_t)(c0 + j) * ldr + toploc + i;
                        *dst = (i <= j) ? sgn[i] * (*dst) : M2[i * QB_LDM + j];
                    }
                }
            __syncthreads();
            QB_STAMP(10);
        } else {
            // ---- column by column (dlarfg / dlarf literally), one all-reduce per column ----
            __syncthreads();
            for (int c = wave; c < nb; c += QB_T / 64)
                for (int i = lane; i < nr; i += 64) Bk[(size_t)(c0 + c) * ldr + i] = A[(size_t)(c0 + c) * a.lda + r0 + i];
            for (int e = tid; e < 32 * QB_LDM; e += QB_T) M1[e] = 0.0;       // M1[k][c] = v_k^T v_c (k < c)
            if (tid < 32) taus[tid] = 0.0;
            __syncthreads();
            const int k = tid & 31, grp = tid >> 5;
            for (int c = 0; c < nb; ++c) {
                // dot products of the tail of column c (rows below its diagonal) with every column
                {
                    double s = 0.0;
                    const int lo_c = max(c0 + c + 1 - rb, 0);
                    if (k < nb)
                        for (int i = lo_c + grp; i < nr; i += QB_T / 32) s += Bk[(size_t)(c0 + k) * ldr + i] * Bk[(size_t)(c0 + c) * ldr + i];
                    red8[grp * 32 + k] = s;
                }
                __syncthreads();
                if (tid < 32) {
                    double s = 0.0;
#pragma unroll
                    for (int q = 0; q < QB_T / 32; ++q) s += red8[q * 32 + tid];
                    qb_put(x, tid, s);
                    const int li = c0 + c - rb;                               // row c of the panel, if it is here
                    qb_put(x, 32 + tid, (li >= 0 && li < nr && tid < nb) ? Bk[(size_t)(c0 + tid) * ldr + li] : 0.0);
                }
                qb_allreduce(x, 64);
                if (tid < 32) { tot[tid] = qb_red(x, tid); rc[tid] = qb_red(x, 32 + tid); }
                __syncthreads();
                if (tid < 32) {
                    const double alpha = rc[c], tail = tot[c];
                    double beta, tau, scale;
                    if (tail <= 2.2250738585072014e-308) { beta = alpha; tau = 0.0; scale = 0.0; }
                    else {
                        beta = sqrt(alpha * alpha + tail);
                        if (alpha >= 0.0) beta = -beta;
                        scale = 1.0 / (alpha - beta);
                        tau = (beta - alpha) / beta;
                    }
                    const double vta = rc[tid] + scale * tot[tid];
                    if (tid < c) M1[tid * QB_LDM + c] = vta;
                    if (tid == c) taus[c] = tau;
                    wv[tid] = (tid > c && tid < nb) ? tau * vta : 0.0;
                    if (tid == 0) { sc[0] = tau; sc[1] = scale; sc[2] = beta; }
                }
                __syncthreads();
                {
                    const double scale = sc[1], beta = sc[2], wk = wv[k];
                    const int lo_c = max(c0 + c - rb, 0);
                    if (k >= c && k < nb)
                        for (int i = lo_c + grp; i < nr; i += QB_T / 32) {
                            const bool diag = (rb + i == c0 + c);
                            const double vi = diag ? 1.0 : Bk[(size_t)(c0 + c) * ldr + i] * scale;
                            const double xk = Bk[(size_t)(c0 + k) * ldr + i];
                            Bk[(size_t)(c0 + k) * ldr + i] = (k > c) ? xk - vi * wk : (diag ? beta : vi);
                        }
                }
                __syncthreads();
            }
            // T = (striu(V^T V) + diag(1/tau))^-1, stored transposed: M0[j][i] = T(i, j)
            for (int e = tid; e < 32 * QB_LDM; e += QB_T) M0[e] = 0.0;
            __syncthreads();
            if (tid < 32) {
                const int j = tid;
                double t[32];
#pragma unroll
                for (int i = 31; i >= 0; --i) {
                    const double tau_i = taus[i];
                    double sum = 0.0;
#pragma unroll
                    for (int l = i + 1; l < 32; ++l) sum += M1[i * QB_LDM + l] * t[l];
                    t[i] = (i == j) ? tau_i : ((i < j) ? -tau_i * sum : 0.0);
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) M0[j * QB_LDM + i] = t[i];
            }
            __syncthreads();
        }
        // ---- this panel's T for the updates outside the block ----
        if (g == 0)
            for (int e = tid; e < 1024; e += QB_T) Tg[e] = M0[(e >> 5) * QB_LDM + (e & 31)];
        QB_STAMP(11);
#ifdef QB_STAMPS
        if (g == 0 && tid == 0) Tg[1024 + 31] = fast ? 1.0 : 0.0;    // which path the panel took
#endif

        // ---- apply the panel's reflectors to the resident columns on its right ----
        const int nrest = a.bwc - (c0 + nb);
        if (nrest > 0) {
            const int nch = (nrest + 31) / 32;
            for (int ch = 0; ch < nch; ++ch) {
                const int z0 = c0 + nb + 32 * ch, zn = min(32, a.bwc - z0);
                qb_xtz_tile<true>(Bk, ldr, rb, lo, rows_pad, c0, nb, z0, zn, x, 1024 * ch);
            }
            qb_allreduce(x, 1024 * nch);
            const int ta = wave >> 1, tb = wave & 1, kr = lane >> 4, l15 = lane & 15;
            for (int ch = 0; ch < nch; ++ch) {
                const int z0 = c0 + nb + 32 * ch, zn = min(32, a.bwc - z0);
                for (int e = tid; e < 1024; e += QB_T) M1[(e >> 5) * QB_LDM + (e & 31)] = qb_red(x, 1024 * ch + e);
                __syncthreads();
                // W <- T^T W
                double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k0 = 0; k0 < 32; k0 += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(M0[(16 * ta + l15) * QB_LDM + k0 + kr], M1[(k0 + kr) * QB_LDM + 16 * tb + l15], acc, 0, 0, 0);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) M1[(16 * ta + 4 * r + kr) * QB_LDM + 16 * tb + l15] = acc[r];
                __syncthreads();
                // A2 -= V W on the local rows, 16 rows x 16 columns per MFMA tile
                const int t_lo = lo >> 4, t_hi = (nr + 15) >> 4;
                for (int rt = t_lo + wave; rt < t_hi; rt += QB_T / 64) {
                    const int i = 16 * rt + l15;
                    double yv[8];
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) yv[q8] = qb_yv(Bk, ldr, rb, c0, nb, i, 4 * q8 + kr);
                    double4_t u[2][2];
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) { u[cb][0] = (double4_t){0.0, 0.0, 0.0, 0.0}; u[cb][1] = u[cb][0]; }
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
                            u[cb][q8 & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(M1[(4 * q8 + kr) * QB_LDM + 16 * cb + l15], yv[q8], u[cb][q8 & 1], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        if (16 * cb >= zn) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int zc = 16 * cb + 4 * r + kr;
                            if (zc < zn && i < nr) Bk[(size_t)(z0 + zc) * ldr + i] -= u[cb][0][r] + u[cb][1][r];
                        }
                    }
                }
                __syncthreads();
            }
        }
    }

    // ---- write the block back; explicit reflectors for the updates outside the block ----
    for (int c = wave; c < a.bwc; c += QB_T / 64)
        for (int i = lane; i < nr; i += 64) A[(size_t)c * a.lda + r0 + i] = Bk[(size_t)c * ldr + i];
    if (a.V) {
        for (int c = wave; c < a.nfac; c += QB_T / 64)
            for (int i = lane; i < nr; i += 64) {
                const int d = rb + i - c;
                if (rb + i >= 0) a.V[(size_t)c * a.ldv + rb + i] = d < 0 ? 0.0 : (d == 0 ? 1.0 : Bk[(size_t)c * ldr + i]);
            }
    }
}

bool qr_block_plan(int rows, int batch, int bw_want, QrBlockPlan *p)
{
    if (rows < 1 || batch < 1) return false;
    const int cap = resident_workgroup_cap();              // 240 on a whole MI355X
    const int gcap = batch <= 8 ? cap / batch : cap;       // workgroups per matrix that are certainly resident together
    if (gcap < 1) return false;
    int bw = bw_want < 32 ? 32 : (bw_want > QB_BW_MAX ? QB_BW_MAX : (bw_want & ~31));
    if (batch > 8) bw = 32;                                // banks: one panel per launch, as few workgroups per matrix as LDS allows
    for (; bw >= 32; bw -= 32) {
        // rows per workgroup: a multiple of 32, as small as keeps the workgroup count within 64 (or the cap) ...
        // 64 rows per workgroup measured best for one matrix (6012 x 6012: 27.3 ms against 29.0 ms at 128 rows and
        // 28.4 ms at 32): the per-row work of a launch halves while the 32 x 32 chains and the exchanges stay
        int gmax = gcap;
        int rp = 64;
        if (const char *e = getenv("VSLAM_QR_ROWS_PER")) rp = atoi(e) < 32 ? 32 : (atoi(e) & ~31);      // A/B runs
        if (batch > 8) {
            const int g = (rows + 511) / 512;              // 512 rows of a 32-column panel fill a workgroup's LDS
            rp = ((rows + g - 1) / g + 31) & ~31;
            if (rp < 128) rp = 128;
        } else if ((rows + rp - 1) / rp > gmax) rp = ((rows + gmax - 1) / gmax + 31) & ~31;
        size_t lds = ((size_t)bw * (rp + 2) + QB_SMALL_DOUBLES) * 8;
        if (lds > QB_LDS_MAX && bw == 32 && gcap > 64) {
            // ... very tall matrices: up to the cap, one panel at a time
            gmax = gcap;
            rp = ((rows + gmax - 1) / gmax + 31) & ~31;
            if (rp < 128) rp = 128;
            lds = ((size_t)bw * (rp + 2) + QB_SMALL_DOUBLES) * 8;
        }
        if (lds > QB_LDS_MAX) continue;
        p->bw = bw; p->rows_per = rp; p->ldr = rp + 2; p->G = (rows + rp - 1) / rp;
        if (p->G > gcap) continue;
        p->cmax = bw > 32 ? 32 * bw : 1024;                    // W of one panel of reflectors against every resident column
        p->lds_bytes = lds < ((size_t)96 << 10) ? ((size_t)96 << 10) : lds;     // one workgroup per compute unit
        p->xbuf_doubles = 2 * ((size_t)(p->G + 1) * p->cmax + 1024);      // two words per value: contributions, sums, top rows
        return true;
    }
    return false;
}

vslam_status qr_block_launch(vslam_ctx *ctx, hipStream_t stream, const QrBlockPlan &pl, double *A, int lda, size_t strideA, int batch,
                             int mp, int bwc, int nfac, unsigned seq0, double *T, size_t strideT, double *V, int ldv,
                             double *xbuf, int roff, const double *Vprev, int ldvp, const double *Tprev, int prev_np, int vprev_rows)
{
    if (bwc > pl.bw || nfac > bwc || nfac > mp || nfac < 1) return fail(ctx, VSLAM_ERR_SHAPE, "qr block: %d columns (%d to factor) in a block of %d", bwc, nfac, pl.bw);
    const int G = (mp + pl.rows_per - 1) / pl.rows_per;
    const int cap = resident_workgroup_cap();
    if (G > pl.G || G > cap) return fail(ctx, VSLAM_ERR_SHAPE, "qr block: %d workgroups per matrix exceed the plan (%d)", G, pl.G);
    if (roff < 0 || roff % 32 != 0 || (roff > 0 && (!Vprev || !Tprev || batch != 1 || nfac > mp - roff)))
        return fail(ctx, VSLAM_ERR_SHAPE, "qr block: inconsistent previous-block arguments");
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&qr_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
    // The workgroups of a matrix wait for one another, so all workgroups of a launch must be resident together: a bank
    // is cut into launches of at most resident_workgroup_cap() workgroups (they follow one another on the stream).
    int per_launch = batch;
    if (G > 1) {
        int per_cu = 0, cus = 0;
        VSLAM_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, qr_block_kernel, QB_T, pl.lds_bytes));
        VSLAM_HIP_TRY(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
        int room = per_cu * cus;
        if (room > cap) room = cap;
        if (per_cu < 1 || G > room)
            return fail(ctx, VSLAM_ERR_CAPACITY, "qr block: %d workgroups that wait for one another, but only %d x %d fit on the device", G, per_cu, cus);
        per_launch = room / G;
    }
    if (per_launch > 65535) per_launch = 65535;
    for (int b0 = 0; b0 < batch; b0 += per_launch) {
        const int nb_ = batch - b0 < per_launch ? batch - b0 : per_launch;
        QbArgs a;
        a.A = A + (size_t)b0 * strideA; a.lda = lda; a.strideA = strideA; a.mp = mp; a.bwc = bwc; a.nfac = nfac;
        a.rows_per = pl.rows_per; a.G = G; a.ldr = pl.ldr; a.bw = pl.bw; a.seq0 = seq0;
        a.T = T + (size_t)b0 * strideT; a.strideT = strideT; a.V = V; a.ldv = ldv;
        a.roff = roff; a.Vprev = Vprev; a.ldvp = ldvp; a.Tprev = Tprev; a.prev_np = prev_np; a.vprev_rows = vprev_rows < 0 ? mp : vprev_rows;
        a.xbuf = xbuf + (size_t)b0 * pl.xbuf_doubles; a.xstride = pl.xbuf_doubles;
        a.cmax = pl.cmax;
        a.err = ctx->flags;
        hipLaunchKernelGGL(qr_block_kernel, dim3(G, nb_), dim3(QB_T), pl.lds_bytes, stream, a);
    }
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam
