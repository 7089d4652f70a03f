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
//   panel factorisation   qr_panel_coop_kernel: the panel lives in the LDS of one workgroup (<= 512
//                         rows, any batch) or of G co-resident workgroups that exchange one vector
//                         per column (taller panels, batch <= 8); right-looking, T from V^T V at
//                         the end.  qr_panel_kernel (one workgroup per matrix, left-looking from
//                         global memory) remains for tall panels in large batches.
//   gemm (MFMA)           W = V^T A2, split over K so the skinny product fills the chip
//   qr_block_solve_kernel W2 = T^T * sum(W slabs)
//   gemm (MFMA)           A2 -= V W2
// V is read straight out of the factored panel (unit diagonal and zeros above it are synthesised by
// the GEMM's operand loader).  One wide matrix additionally groups four panels into a 128-column
// block whose far updates run on a side stream (see qr_outer_block).
#include "qr_block.hpp"
#include <cstdlib>
#include <mutex>

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
                                                      double *Tbase /*[batch][NB*NB + NB], T column-major ld NB*/)
{
    double *A = Abase + (size_t)blockIdx.x * strideA;       // panel origin (row j0, column j0)
    double *Tg = Tbase + (size_t)blockIdx.x * (NB * NB + NB);
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

// ---- tall panels: many workgroups per panel --------------------------------------
// A panel of several thousand rows keeps one workgroup busy for milliseconds, so a
// single large matrix is factored by G co-resident workgroups per panel instead.
// Workgroup g holds rows [g*rows_per, (g+1)*rows_per) of the m x 32 panel in LDS for
// the whole factorisation (right-looking, column by column, the order of LAPACK's
// dgeqr2).  Column c needs ONE exchange: every workgroup publishes
//     p_k = sum_{i > c, i in its rows} P[i][k] * P[i][c],   k = 0..31
// (k = c: squared norm of the tail; k > c: tail dot products for the update;
//  k < c: V_k^T x_c for the compact-WY T) and, if it owns row c, that row; every
// workgroup then adds the G partial vectors in the same order, so all of them derive
// bit-identical beta / tau / w and update their own rows.  One LDS pass applies
// reflector c and accumulates the dot products of column c+1.
//
// The exchange has no counter: a slot holds a reserved NaN bit pattern until its data
// arrives, and readers poll the data words themselves (one round trip).  Slots rotate
// through three buffers indexed by a column count that keeps running across the panel
// launches of one factorisation: publishing column t first marks the slot of t+1 empty
// (nobody reads that buffer any more: t-2 was consumed before anyone published t-1),
// then release-stores the data of t.  A genuine input NaN with the reserved payload
// would stall the poll; it ends at the iteration bound, raises *err, and every wave
// still runs to the end.  The launcher keeps G * batch <= 240 so all workgroups are
// resident together.
#ifndef VSLAM_CP_T
#define VSLAM_CP_T 1024
#endif
constexpr int CP_T = VSLAM_CP_T;     // threads per workgroup
constexpr int CP_NG = CP_T / 32;     // row groups (each 32 lanes = the panel's 32 columns)
constexpr int CP_RMAX = 512;         // most rows a workgroup holds
constexpr int CP_LD = NB + 1;        // LDS row stride (doubles)
constexpr int CP_GMAX = 64;          // most workgroups per panel (8 polled slots per thread)
constexpr int CP_SLOT = 2 * NB;      // doubles per slot: partial sums, then row c
constexpr unsigned CP_EMPTY32 = 0x7FF8DEADu;
constexpr unsigned long long CP_EMPTY = 0x7FF8DEAD7FF8DEADull;
constexpr unsigned CP_SPIN_LIMIT = 1u << 22;
constexpr int MSTRIDE = NB * NB + NB;   // doubles reserved per panel for T (column-major, ld NB)

struct CoopPanelArgs {
    double *A;            // panel origin (row j0, column j0) of matrix 0
    int lda;
    size_t strideA;
    int mp, jb, rows_per, G;
    unsigned tbase;       // running column count at the first column of this launch
    double *M;            // [batch][MSTRIDE]: the panel's compact-WY T, column-major ld NB
    double *slots;        // [batch][3][CP_GMAX][CP_SLOT]
    int *err;
    double *V;            // optional: explicit reflectors (unit diagonal, zeros above), element (panel row 0, panel column 0)
    int ldv, vzero;       // leading dimension; rows above the panel to clear in these columns
};

static size_t coop_lds_bytes(int rows_per)
{
    return ((size_t)rows_per * CP_LD + CP_NG * NB + 3 * NB + NB * (NB + 1) + 4) * 8;
}

__device__ __forceinline__ double coop_poll(const double *p, int *err, bool &dead)
{
    unsigned spins = 0;
    for (;;) {
        const double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__double_as_longlong(v) != (long long)CP_EMPTY || dead) return v;
        if ((++spins & 255u) == 0) {
            if (spins > CP_SPIN_LIMIT) atomicExch(err, 1);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { dead = true; return 0.0; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ __launch_bounds__(CP_T) void qr_panel_coop_kernel(CoopPanelArgs a)
{
    extern __shared__ double lds[];
    double *P = lds;                                   // [rows_per][CP_LD]
    double *red = P + (size_t)a.rows_per * CP_LD;      // [CP_NG][NB]
    double *tot = red + CP_NG * NB;                        // [NB]
    double *rc = tot + NB;                             // [NB] row c of the panel
    double *w = rc + NB;                               // [NB]
    double *Ms = w + NB;                               // [NB][NB+1]: Ms[k][r] = v_k^T v_r, k < r; row NB-ish tail: taus
    double *sc = Ms + NB * (NB + 1);                   // tau, scale, beta
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
    const int k = tid & (NB - 1), grp = tid >> 5;      // CP_NG row groups x 32 columns
    double *A = a.A + (size_t)b * a.strideA;
    double *slots = a.slots + (size_t)b * 3 * CP_GMAX * CP_SLOT;
    const int r0 = g * a.rows_per;
    const int nr = min(a.rows_per, a.mp - r0);         // >= 1 by construction of G
    const int jb = a.jb, G = a.G;
    const bool solo = (G == 1);                        // the whole panel is here: no exchange, sums stay in LDS
    bool dead = false;

    for (int idx = tid; idx < nr * NB; idx += CP_T) {
        const int li = idx % nr, kk = idx / nr;
        P[li * CP_LD + kk] = (kk < jb) ? A[(size_t)kk * a.lda + r0 + li] : 0.0;
    }
    for (int i = tid; i < NB * (NB + 1); i += CP_T) Ms[i] = 0.0;
    __syncthreads();

    // red[grp][k] holds this workgroup's partial sums for column `col`; publish them (and row `col`).
    // Thread k < 32 owns words k and 32+k of the slot: it empties both in the next buffer
    // before its release-store of partial word k, the word every reader waits for.
    auto publish = [&](int col) {
        if (tid < NB) {
            const unsigned t = a.tbase + (unsigned)col;
            double *next = slots + ((size_t)((t + 1) % 3) * CP_GMAX + g) * CP_SLOT;
            double *cur = slots + ((size_t)(t % 3) * CP_GMAX + g) * CP_SLOT;
            const double empty = __longlong_as_double((long long)CP_EMPTY);
            __hip_atomic_store(&next[tid], empty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&next[NB + tid], empty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (col >= r0 && col < r0 + nr)
                __hip_atomic_store(&cur[NB + tid], P[(col - r0) * CP_LD + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < CP_NG / 2; ++q) v += red[q * NB + tid];
            __hip_atomic_store(&cur[tid], v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // dot products of the tail of column 0 with every column
    {
        double s = 0.0;
        const int lo = max(1 - r0, 0);
        for (int li = lo + grp; li < nr; li += CP_NG) s += P[li * CP_LD + k] * P[li * CP_LD];
        s += __shfl_xor(s, 32, 64);
            if ((tid & 32) == 0) red[(tid >> 6) * NB + k] = s;
    }
    __syncthreads();
    if (!solo) publish(0);

    for (int c = 0; c < jb; ++c) {
        const unsigned t = a.tbase + (unsigned)c;
        const double *cur = slots + (size_t)(t % 3) * CP_GMAX * CP_SLOT;
        if (solo) {
            if (tid < NB) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < CP_NG / 2; ++q) s += red[q * NB + tid];
                tot[tid] = s;
                rc[tid] = P[c * CP_LD + tid];
            }
            // tot / rc are written and next read by the same half-wave: no workgroup barrier needed
        } else {
        // every workgroup adds the partials in the same order
        {
            double v[(CP_GMAX + CP_NG - 1) / CP_NG];
#pragma unroll
            for (int j = 0; j < (CP_GMAX + CP_NG - 1) / CP_NG; ++j) {
                const int q = grp + CP_NG * j;
                v[j] = (q < G) ? __hip_atomic_load(&cur[(size_t)q * CP_SLOT + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < (CP_GMAX + CP_NG - 1) / CP_NG; ++j) {
                const int q = grp + CP_NG * j;
                if (q < G && __double_as_longlong(v[j]) == (long long)CP_EMPTY) v[j] = coop_poll(&cur[(size_t)q * CP_SLOT + k], a.err, dead);
                s += v[j];
            }
            __syncthreads();                           // red is free: publish() of this column has read it
            s += __shfl_xor(s, 32, 64);
            if ((tid & 32) == 0) red[(tid >> 6) * NB + k] = s;
            if (tid < NB) rc[tid] = coop_poll(&cur[(size_t)(c / a.rows_per) * CP_SLOT + NB + tid], a.err, dead);
        }
        __syncthreads();
        if (tid < NB) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < CP_NG / 2; ++q) s += red[q * NB + tid];
            tot[tid] = s;
        }
        }
        // tot / rc were written by lanes 0..31 of wave 0, the only readers until the next barrier
        // lanes 0..31 of wave 0: reflector scalars (each lane the same), (V^T v_c)_k for k < c into the
        // inverse-T form, update weights w_k = tau * v_c^T a_k for k > c
        if (tid < NB) {
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
            if (tid < c) Ms[tid * (NB + 1) + c] = vta;
            if (tid == c) Ms[c * (NB + 1) + NB] = tau;      // taus live in the pad column
            w[tid] = (tid > c && tid < jb) ? tau * vta : 0.0;
            if (tid == 0) { sc[0] = tau; sc[1] = scale; sc[2] = beta; }
        }
        __syncthreads();
        const double scale = sc[1], beta = sc[2];
        const int c1 = (c + 1 < NB) ? c + 1 : c;       // column whose dots are accumulated (unused when c+1 == jb)
        const double wk = w[k], wc1 = w[c1];
        // one pass over this workgroup's rows >= c: apply reflector c, accumulate the
        // dot products of the tail (rows > c+1) of the updated column c+1 with every column
        double s = 0.0;
        {
            const int lo = max(c - r0, 0);
            for (int li0 = lo + grp; li0 < nr; li0 += 4 * CP_NG) {
                double xc[4], x1[4], xk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double *row = P + min(li0 + CP_NG * u, nr - 1) * CP_LD;
                    xc[u] = row[c]; x1[u] = row[c1]; xk[u] = row[k];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int li = li0 + CP_NG * u, i = r0 + li;
                    if (li < nr) {
                        const double vi = (i == c) ? 1.0 : xc[u] * scale;
                        const double nk = (k > c) ? xk[u] - vi * wk : ((k == c) ? ((i == c) ? beta : vi) : xk[u]);
                        const double n1 = x1[u] - vi * wc1;
                        if (k >= c) P[li * CP_LD + k] = nk;
                        if (i > c + 1) s += nk * n1;
                    }
                }
            }
        }
        __syncthreads();                               // T / red readers of this column are done; rows are updated
        if (c + 1 < jb) {
            s += __shfl_xor(s, 32, 64);
            if ((tid & 32) == 0) red[(tid >> 6) * NB + k] = s;
            __syncthreads();
            if (!solo) publish(c + 1);
        }
    }
    __syncthreads();
    // waves 1.. write the panel back while wave 0 (of workgroup 0) builds T below
    if (tid >= 64) {
        for (int idx = tid - 64; idx < nr * jb; idx += CP_T - 64) {
            const int li = idx % nr, kk = idx / nr;
            A[(size_t)kk * a.lda + r0 + li] = P[li * CP_LD + kk];
        }
        if (a.V) {
            for (int idx = tid - 64; idx < nr * jb; idx += CP_T - 64) {
                const int li = idx % nr, kk = idx / nr, i = r0 + li;
                a.V[(size_t)kk * a.ldv + i] = (i < kk) ? 0.0 : ((i == kk) ? 1.0 : P[li * CP_LD + kk]);
            }
            if (g == 0)
                for (int idx = tid - 64; idx < a.vzero * jb; idx += CP_T - 64) {
                    const int z = idx % a.vzero, kk = idx / a.vzero;
                    a.V[(ptrdiff_t)kk * a.ldv - 1 - z] = 0.0;
                }
        }
    }
    // T = (striu(V^T V) + diag(1/tau))^-1: lane j back-substitutes column j (a reflector with
    // tau = 0 gives a zero row and column).
    if (g == 0 && tid < NB) {
        double *Mg = a.M + (size_t)b * MSTRIDE;
        const int j = tid;
        double t[NB];
#pragma unroll
        for (int i = NB - 1; i >= 0; --i) {
            const double tau_i = Ms[i * (NB + 1) + NB];
            double sum = 0.0;
#pragma unroll
            for (int l = i + 1; l < NB; ++l) sum += Ms[i * (NB + 1) + l] * t[l];
            t[i] = (i == j) ? tau_i : ((i < j) ? -tau_i * sum : 0.0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) Mg[j * NB + i] = t[i];
    }
}

// W = T^T Y for a block of `ob` <= 128 reflectors made of panels of 32 whose own T_p are known
// (Tsm[p], column-major ld NB): the block T is never formed; with G = V^T V,
//   w_p = T_p^T ( y_p - G[p-rows, earlier rows] w_{<p} ),  p = 0, 1, ...
// Y arrives as `ks` K-slabs that are summed on load.  A workgroup owns 16 columns; lane r of a
// 32-lane group owns row 32p + r of one column.
constexpr int SOLVE_CT = 16;
__global__ __launch_bounds__(256) void qr_block_solve_kernel(const double *Ypart, size_t strideSplit, int ks, int ldy,
                                                             size_t strideBatchY, int ob, int n2, const double *Gbig,
                                                             const double *Tsm, size_t strideBatchT, double *Wout, int ldw,
                                                             size_t strideBatchW)
{
    constexpr int OBM = 128;
    __shared__ double Ys[OBM][SOLVE_CT + 1];
    __shared__ double Gp[OBM][NB + 1];
    const int tid = threadIdx.x, r = tid & 31, cg = tid >> 5, b = blockIdx.y;
    const int c0 = blockIdx.x * SOLVE_CT;
    const double *Y = Ypart + (size_t)b * strideBatchY;
    const double *Tb = Tsm + (size_t)b * strideBatchT;
    double *W = Wout + (size_t)b * strideBatchW;
    for (int idx = tid; idx < ob * SOLVE_CT; idx += 256) {
        const int row = idx % ob, c = idx / ob, col = c0 + c;
        double v = 0.0;
        if (col < n2) {
#pragma unroll 8
            for (int sidx = 0; sidx < ks; ++sidx) v += Y[(size_t)sidx * strideSplit + (size_t)col * ldy + row];
        }
        Ys[row][c] = v;
    }
    const int np = (ob + NB - 1) / NB;
    for (int p = 0; p < np; ++p) {
        const int h = p * NB, wq = min(NB, ob - h);
        const double *Tp = Tb + (size_t)p * MSTRIDE;
        __syncthreads();
        for (int idx = tid; idx < h * NB; idx += 256) {              // Gp[l][r] = G(h + r, l), l < h
            const int l = idx / NB, rr = idx % NB;
            Gp[l][rr] = (rr < wq) ? Gbig[(size_t)l * OBM + h + rr] : 0.0;
        }
        for (int idx = tid; idx < NB * NB; idx += 256) {             // Gp[h + l][r] = T_p(l, r)
            const int rr = idx / NB, l = idx % NB;
            Gp[h + l][rr] = (l <= rr && rr < wq) ? Tp[rr * NB + l] : 0.0;
        }
        __syncthreads();
        double y[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cc = cg + 8 * half;
            y[half] = (r < wq) ? Ys[h + r][cc] : 0.0;
            if (p > 0) {
                double sum = 0.0;
                for (int l = 0; l < h; ++l) sum += Gp[l][r] * Ys[l][cc];
                y[half] -= sum;
            }
        }
        double w0 = 0.0, w1 = 0.0;
#pragma unroll
        for (int l = 0; l < NB; ++l) {
            const double t = Gp[h + l][r];
            w0 += t * __shfl(y[0], l, 32);
            w1 += t * __shfl(y[1], l, 32);
        }
        if (r < wq) { Ys[h + r][cg] = w0; Ys[h + r][cg + 8] = w1; }
    }
    __syncthreads();
    for (int idx = tid; idx < ob * SOLVE_CT; idx += 256) {
        const int row = idx % ob, c = idx / ob, col = c0 + c;
        if (col < n2) W[(size_t)col * ldw + row] = Ys[row][c];
    }
}

// Rows per workgroup for the many-workgroup panel kernel on an m-row matrix (0: use the single-workgroup kernel).
static int coop_rows_per(int rows, int batch)
{
    if (rows < 128 || batch > 8) return 0;
    const int cap = resident_workgroup_cap();
    if (cap / batch < 2) return 0;
    const int gmax = (cap / batch) < CP_GMAX ? (cap / batch) : CP_GMAX;
    int rp = rows >= 2048 ? 128 : 64;                  // measured: fewer, fatter workgroups lose to the LDS pass
    if ((rows + rp - 1) / rp > gmax) rp = (((rows + gmax - 1) / gmax) + 7) & ~7;
    return rp > CP_RMAX ? 0 : rp;
}

constexpr int OB = 128;          // outer block: four panels share one trailing update

// out[i] = sum over slabs of part[s * strideSplit + i]
__global__ __launch_bounds__(256) void sum_slabs_kernel(const double *part, size_t strideSplit, int ksplit, size_t count, double *out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int k = 0; k < ksplit; ++k) s += part[(size_t)k * strideSplit + i];
    out[i] = s;
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
    int maxs = (mp + 127) / 128;                       // at least 128 rows of K per slab
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    return want;
}

static int max_ksplit(int rows, int batch)
{
    int a = 512 / batch, b = (rows + 127) / 128;
    int m = a < b ? a : b;
    return m < 1 ? 1 : (m > 64 ? 64 : m);
}

// Two-level blocking applies to one tall, wide matrix: panels of 32 columns are factored and
// applied inside an outer block of 128 columns, whose 128 reflectors then update the rest of
// the matrix with K = 128 products (4x less trailing-matrix traffic than per-panel updates).
static bool use_outer_blocks(int rows, int cols, int batch)
{
    return batch == 1 && coop_rows_per(rows, batch) > 0 && cols >= 2 * OB;
}

// Columns qr_block_kernel keeps resident per launch.
static bool use_outer_blocks(int rows, int cols, int batch);
static int block_width(int rows, int cols, int batch)
{
    return use_outer_blocks(rows, cols, batch) ? QB_BW_MAX : QR_NB;
}

struct QrWs {
    double *T, *Wpart, *W2, *slots;
    double *xbuf;                         // exchange area of qr_block_kernel
    double *Vbuf[2], *G[2], *Mo[2], *Gpart, *Wn, *Wnpart, *Wb, *Wbpart;   // outer-block path only (double-buffered per block parity)
};

static void qr_carve(void *ws, int rows, int cols, int batch, QrWs *w)
{
    Arena ar(ws);
    w->T = ar.take<double>((size_t)(batch > 4 ? batch : 4) * MSTRIDE);
    w->Wpart = ar.take<double>((size_t)batch * max_ksplit(rows, batch) * NB * (size_t)cols);
    w->W2 = ar.take<double>((size_t)batch * NB * (size_t)cols);
    w->slots = ar.take<double>((size_t)(batch > 8 ? 1 : batch) * 3 * CP_GMAX * CP_SLOT);
    {
        QrBlockPlan pl;
        const bool ok = qr_block_plan(rows, batch, block_width(rows, cols, batch), &pl);
        w->xbuf = ar.take<double>(ok ? (size_t)batch * pl.xbuf_doubles : 1);
    }
    if (use_outer_blocks(rows, cols, batch)) {
        for (int i = 0; i < 2; ++i) {
            w->Vbuf[i] = ar.take<double>((size_t)rows * OB);
            w->G[i] = ar.take<double>((size_t)OB * OB);
            w->Mo[i] = ar.take<double>((size_t)(OB / NB) * MSTRIDE);
        }
        w->Gpart = ar.take<double>((size_t)64 * OB * OB);
        w->Wn = ar.take<double>((size_t)OB * OB);
        w->Wnpart = ar.take<double>((size_t)8 * OB * OB);
        w->Wb = ar.take<double>((size_t)OB * cols);
        w->Wbpart = ar.take<double>((size_t)8 * OB * cols);
    }
}

size_t qr_ws_bytes(int rows, int cols, int batch)
{
    size_t t = Arena::pad((size_t)(batch > 4 ? batch : 4) * MSTRIDE * 8);
    size_t wpart = Arena::pad((size_t)batch * max_ksplit(rows, batch) * NB * (size_t)cols * 8);
    size_t w2 = Arena::pad((size_t)batch * NB * (size_t)cols * 8);
    size_t coop = Arena::pad((size_t)(batch > 8 ? 1 : batch) * 3 * CP_GMAX * CP_SLOT * 8);
    {
        QrBlockPlan pl;
        const bool ok = qr_block_plan(rows, batch, block_width(rows, cols, batch), &pl);
        coop += Arena::pad((ok ? (size_t)batch * pl.xbuf_doubles : 1) * 8);
    }
    size_t big = 0;
    if (use_outer_blocks(rows, cols, batch))
        big = 2 * (Arena::pad((size_t)rows * OB * 8) + Arena::pad((size_t)OB * OB * 8) + Arena::pad((size_t)(OB / NB) * MSTRIDE * 8)) +
              Arena::pad((size_t)64 * OB * OB * 8) + Arena::pad((size_t)OB * OB * 8) + Arena::pad((size_t)8 * OB * OB * 8) +
              Arena::pad((size_t)OB * cols * 8) + Arena::pad((size_t)8 * OB * cols * 8);
    return t + wpart + w2 + coop + big;
}

struct QrJob {
    vslam_ctx *ctx;
    double *A;
    int rows, cols, lda;
    size_t strideA;
    int batch, rows_per;
    unsigned tbase;
    bool use_block;          // panels go through qr_block_kernel
    QrBlockPlan pl;
    unsigned launches;       // launches of qr_block_kernel so far in this factorisation
    int ob;                  // reflectors per outer block (128, or what qr_block_kernel keeps resident)
    int band;                // >= 0: A(i, c) == 0 for i > c + band on entry (a diagonal block under a dense one); -1: dense
    QrWs w;
};

// Rows of the panel at (j, j) with jb columns that can hold a non-zero: below a band the reflectors of the panel are zero
// and so is everything they would change, so neither the factorisation nor the update of the later columns goes there.
static inline int panel_rows(const QrJob &q, int j, int jb)
{
    const int mp = q.rows - j;
    return (q.band >= 0 && q.band + jb < mp) ? q.band + jb : mp;
}

// Factors the jb-column panel whose origin is (j, j); T goes to Tdst.  Vout (optional) receives the
// explicit reflectors, vzero rows above the panel cleared.
static vslam_status qr_factor_panel(QrJob &q, int j, int jb, double *Tdst, double *Vout, int ldv, int vzero)
{
    const int mp = panel_rows(q, j, jb);
    double *P = q.A + (size_t)j * q.lda + j;
    hipStream_t s = q.ctx->stream;
    if (q.use_block) {
        const unsigned seq0 = 256u * q.launches++;
        if (Vout && vzero > 0)
            VSLAM_HIP_TRY(q.ctx, hipMemset2DAsync(Vout - vzero, (size_t)ldv * 8, 0, (size_t)vzero * 8, (size_t)jb, s));
        return qr_block_launch(q.ctx, s, q.pl, P, q.lda, q.strideA, q.batch, mp, jb, jb, seq0, Tdst, (size_t)MSTRIDE, Vout, ldv,
                               q.w.xbuf);
    }
    if (q.rows_per > 0 || mp <= CP_RMAX) {
        // A panel of at most CP_RMAX rows fits one workgroup's LDS: no exchange, any batch.
        const int rows_per = (mp <= CP_RMAX) ? ((mp + 63) / 64) * 64 : q.rows_per;
        CoopPanelArgs ca;
        ca.A = P; ca.lda = q.lda; ca.strideA = q.strideA; ca.mp = mp; ca.jb = jb; ca.rows_per = rows_per;
        ca.G = (mp + rows_per - 1) / rows_per; ca.tbase = q.tbase;
        ca.M = Tdst; ca.slots = q.w.slots; ca.err = q.ctx->flags;
        ca.V = Vout; ca.ldv = ldv; ca.vzero = vzero;
        const size_t lds = coop_lds_bytes(rows_per);
        hipLaunchKernelGGL(qr_panel_coop_kernel, dim3(ca.G, q.batch), dim3(CP_T), lds, s, ca);
        q.tbase += (unsigned)jb;
    } else {
        if (Vout) return fail(q.ctx, VSLAM_ERR_SHAPE, "qr: explicit reflectors need a tall panel");
        hipLaunchKernelGGL(qr_panel_kernel, dim3(q.batch), dim3(PT), 0, s, P, q.lda, q.strideA, mp, jb, Tdst);
    }
    VSLAM_HIP_TRY(q.ctx, hipGetLastError());
    return VSLAM_OK;
}

// Applies the block reflector of the panel at (j, j) (V below the diagonal of the panel, T at Tsrc)
// to the n2 columns that follow it.
static vslam_status qr_apply_panel(QrJob &q, int j, int jb, const double *Tsrc, int n2)
{
    if (n2 <= 0) return VSLAM_OK;
    vslam_ctx *ctx = q.ctx;
    hipStream_t s = ctx->stream;
    const int mp = panel_rows(q, j, jb), batch = q.batch;
    double *P = q.A + (size_t)j * q.lda + j;
    double *A2 = P + (size_t)jb * q.lda;
    const int ksplit = pick_ksplit(mp, n2, batch);
    const size_t slab = (size_t)NB * n2;
    GemmDesc g1;                                   // Wpart[s] = V^T A2 over K slab s
    g1.M = jb; g1.N = n2; g1.K = mp; g1.transA = true; g1.transB = false; g1.amode = A_UNIT_LOWER;
    g1.A = P; g1.lda = q.lda; g1.strideA = q.strideA;
    g1.B = A2; g1.ldb = q.lda; g1.strideB = q.strideA;
    g1.C = q.w.Wpart; g1.ldc = NB; g1.strideC = slab * ksplit; g1.batch = batch;
    g1.ksplit = ksplit; g1.strideSplit = slab;
    if (ksplit == 1) { g1.C = q.w.W2; g1.strideC = slab; }   // no slabs: result goes straight to W2's buffer
    VSLAM_TRY(gemm_f64(ctx, g1));
    // W2 = T^T * sum slabs (ksplit == 1: in place on W2's buffer)
    const double *src = (ksplit == 1) ? q.w.W2 : q.w.Wpart;
    hipLaunchKernelGGL(qr_block_solve_kernel, dim3((n2 + SOLVE_CT - 1) / SOLVE_CT, batch), dim3(256), 0, s, src, slab, ksplit, NB,
                       (ksplit == 1) ? slab : slab * ksplit, jb, n2, (const double *)nullptr, Tsrc, (size_t)MSTRIDE,
                       q.w.W2, NB, slab);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    GemmDesc g3;                                   // A2 -= V W2
    g3.M = mp; g3.N = n2; g3.K = jb; g3.transA = false; g3.transB = false; g3.amode = A_UNIT_LOWER;
    g3.alpha = -1.0; g3.beta = 1.0;
    g3.A = P; g3.lda = q.lda; g3.strideA = q.strideA;
    g3.B = q.w.W2; g3.ldb = NB; g3.strideB = slab;
    g3.C = A2; g3.ldc = q.lda; g3.strideC = q.strideA; g3.batch = batch;
    return gemm_f64(ctx, g3);
}

// The two large products of a far update, for a full block of OB = 128 reflectors held explicitly (V: rows x 128,
// unit diagonal, zeros above it).  Both are bound by how fast operands reach the matrix cores, so they are kept
// minimal like the trailing update of the Cholesky factorisation (newton.hip): plain indexing, sixteen rows or
// reflectors at a time through LDS, several workgroups per compute unit.
typedef double qr_double4 __attribute__((ext_vector_type(4)));
constexpr int FKC = 16, FLD = FKC + 2;

// Y slab s = V[rows of the slab]^T C[rows of the slab, 64 columns]: a workgroup owns all 128 reflectors against 64
// columns (32 MFMAs per wave and 16-row chunk against 12 loads per thread), rows split over gridDim.y slabs.
__global__ __launch_bounds__(256, 3) void qr_far_vtc_kernel(const double *V, int ldv, const double *C, int ldc, int rows, int nc, int kchunk,
                                                            double *Y /* [slab][nc][OB] */, size_t slab_stride)
{
    __shared__ __align__(16) double Vs[OB * FLD], Cs[64 * FLD];        // [reflector / column][row of the chunk]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kr = lane >> 4;
    const int n0 = blockIdx.x * 64, k_begin = blockIdx.y * kchunk, k_end = min(rows, k_begin + kchunk);
    qr_double4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (qr_double4){0.0, 0.0, 0.0, 0.0};
    const int kk = tid & 15, cq = tid >> 4;                           // thread: row kk of the chunk, columns cq + 16 e
    for (int k0 = k_begin; k0 < k_end; k0 += FKC) {
        double rv[8], rc[4];
        const bool in = k0 + kk < k_end;
#pragma unroll
        for (int e = 0; e < 8; ++e) rv[e] = in ? V[(size_t)(cq + 16 * e) * ldv + k0 + kk] : 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) rc[e] = in ? C[(size_t)min(n0 + cq + 16 * e, nc - 1) * ldc + k0 + kk] : 0.0;
        if (k0 > k_begin) __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) Vs[(cq + 16 * e) * FLD + kk] = rv[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) Cs[(cq + 16 * e) * FLD + kk] = rc[e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < FKC; q += 4) {
            double am[2], bn[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) am[i] = Vs[(32 * wave + 16 * i + l15) * FLD + q + kr];
#pragma unroll
            for (int j = 0; j < 4; ++j) bn[j] = Cs[(16 * j + l15) * FLD + q + kr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[j], am[i], acc[i][j], 0, 0, 0);
        }
    }
    // acc[i][j][r] <-> reflector 32 wave + 16 i + l15, column n0 + 16 j + kr + 4 r
    double *Ys = Y + (size_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * j + kr + 4 * r;
                if (n < nc) Ys[(size_t)n * OB + 32 * wave + 16 * i + l15] = acc[i][j][r];
            }
}

// C -= V W for the 128 reflectors: one workgroup per 64 x 64 tile of C, the old tile requested first.
__global__ __launch_bounds__(256, 4) void qr_far_update_kernel(const double *V, int ldv, const double *W /* [nc][OB] */, double *C, int ldc,
                                                               int rows, int nc, int tilesM)
{
    constexpr int ALD = 64 + 16;                                      // V chunk [reflector][row]: row stride 16 mod 32 doubles
    __shared__ __align__(16) double Vs[FKC * ALD], Ws[64 * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kr = lane >> 4;
    const int m0 = (blockIdx.x % tilesM) * 64, n0 = (blockIdx.x / tilesM) * 64;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    double cold[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(m0 + wm + 16 * i + l15, rows - 1), n = min(n0 + wn + 16 * j + kr + 4 * r, nc - 1);
                cold[i][j][r] = C[(size_t)n * ldc + m];
            }
    qr_double4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (qr_double4){0.0, 0.0, 0.0, 0.0};
    const int mrow = min(m0 + lane, rows - 1);
    const int kk = tid & 15, cq = tid >> 4;
    for (int k0 = 0; k0 < OB; k0 += FKC) {
        double rv[4], rw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) rv[e] = V[(size_t)(k0 + wave + 4 * e) * ldv + mrow];            // lane = row: 512 contiguous bytes per reflector
#pragma unroll
        for (int e = 0; e < 4; ++e) rw[e] = W[(size_t)min(n0 + cq + 16 * e, nc - 1) * OB + k0 + kk];
        if (k0 > 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) Vs[(wave + 4 * e) * ALD + lane] = rv[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) Ws[(cq + 16 * e) * FLD + kk] = rw[e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < FKC; q += 4) {
            double am[2], bn[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) am[i] = Vs[(q + kr) * ALD + wm + 16 * i + l15];
#pragma unroll
            for (int j = 0; j < 2; ++j) bn[j] = Ws[(wn + 16 * j + l15) * FLD + q + kr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[j], am[i], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + l15, n = n0 + wn + 16 * j + kr + 4 * r;
                if (m < rows && n < nc) C[(size_t)n * ldc + m] = cold[i][j][r] - acc[i][j][r];
            }
}

// Applies the 128 reflectors of the block held in Vbuf/G/Mo[par] to `nc` columns starting at C
// (rows j0.. of the matrix) on `stream`:  Y = V^T C (K-slabs) -> W = T^T Y -> C -= V W.
static vslam_status qr_apply_outer(QrJob &q, int par, int mp0, double *C, int nc, double *W, double *Wpart, hipStream_t stream)
{
    if (nc <= 0) return VSLAM_OK;
    vslam_ctx *ctx = q.ctx;
    const int tiles = 2 * ((nc + 63) / 64);
    int kw = 512 / tiles;
    if (kw > mp0 / 512) kw = mp0 / 512;
    if (kw > 8) kw = 8;
    if (kw < 1) kw = 1;
    const size_t count = (size_t)OB * nc;
    const int ob = q.ob;
    const bool full = ob == OB && !getenv("VSLAM_QR_FAR_GEMM");      // the dedicated kernels take a full block of reflectors
    if (full) {
        // 128 x 64 tiles: half as many workgroups per slab as the GEMM's tiling, so more slabs (at most 8: the slab buffer)
        kw = 768 / ((nc + 63) / 64);
        if (kw > mp0 / 256) kw = mp0 / 256;
        if (kw > 8) kw = 8;
        if (kw < 1) kw = 1;
        int kchunk = (mp0 + kw - 1) / kw;
        kchunk = ((kchunk + FKC - 1) / FKC) * FKC;
        hipLaunchKernelGGL(qr_far_vtc_kernel, dim3((nc + 63) / 64, kw), dim3(256), 0, stream, q.w.Vbuf[par], mp0, C, q.lda, mp0, nc, kchunk,
                           (kw == 1) ? W : Wpart, count);
    } else {
        GemmDesc gw;                                   // Y = V^T C (K-slabs)
        gw.M = ob; gw.N = nc; gw.K = mp0; gw.transA = true; gw.transB = false;
        gw.A = q.w.Vbuf[par]; gw.lda = mp0; gw.B = C; gw.ldb = q.lda;
        gw.C = (kw == 1) ? W : Wpart; gw.ldc = OB; gw.ksplit = kw; gw.strideSplit = count;
        VSLAM_TRY(gemm_f64_on(ctx, gw, stream));
    }
    hipLaunchKernelGGL(qr_block_solve_kernel, dim3((nc + SOLVE_CT - 1) / SOLVE_CT, 1), dim3(256), 0, stream,
                       (kw == 1) ? W : Wpart, count, kw, OB, (size_t)0, ob, nc, q.w.G[par], q.w.Mo[par], (size_t)0, W, OB, (size_t)0);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    if (full) {
        const int tilesM = (mp0 + 63) / 64;
        hipLaunchKernelGGL(qr_far_update_kernel, dim3(tilesM * ((nc + 63) / 64)), dim3(256), 0, stream, q.w.Vbuf[par], mp0, W, C, q.lda, mp0, nc, tilesM);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        return VSLAM_OK;
    }
    GemmDesc gu;                                   // C -= V W
    gu.M = mp0; gu.N = nc; gu.K = ob; gu.transA = false; gu.transB = false; gu.alpha = -1.0; gu.beta = 1.0;
    gu.A = q.w.Vbuf[par]; gu.lda = mp0; gu.B = W; gu.ldb = OB; gu.C = C; gu.ldc = q.lda;
    return gemm_f64_on(ctx, gu, stream);
}

// One outer block of OB columns at (j0, j0), block number `blk`, parity par = blk & 1.
// Main stream: the block's four panels; each panel's reflectors are applied at once to the columns
// the next panels need -- the rest of this block and the whole next block.
// Side stream: the block's 128 reflectors as one unit (Gram matrix + the panels' T) update
// everything beyond the next block: first the 128 columns of block blk+2 ("near"), then the rest
// ("far"), overlapping the next block's panels.
// Events (ctx->ev): [par] reflectors of the block complete (main -> side); [2+par] near update
// done, [4+par] far update done (side -> main).  The next block's panel applies reach into block
// blk+2's columns, so they wait for this block's near update; the reflector buffers of a parity
// are rewritten two blocks later, after that block's far update.
static vslam_status qr_outer_block(QrJob &q, int j0, int blk)
{
    vslam_ctx *ctx = q.ctx;
    hipStream_t s = ctx->stream;
    const int mp0 = q.rows - j0, par = blk & 1;
    const int reach = (j0 + 2 * OB < q.cols) ? j0 + 2 * OB : q.cols;         // panel applies stop here
    if (blk >= 2) VSLAM_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev[4 + par], 0));
    for (int p = 0; p < OB / NB; ++p) {
        const int j = j0 + p * NB;
        double *Tp = q.w.Mo[par] + (size_t)p * MSTRIDE;
        VSLAM_TRY(qr_factor_panel(q, j, NB, Tp, q.w.Vbuf[par] + (size_t)p * NB * mp0 + p * NB, mp0, p * NB));
        if (p == 0 && blk >= 1) VSLAM_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev[2 + (par ^ 1)], 0));
        VSLAM_TRY(qr_apply_panel(q, j, NB, Tp, reach - (j + NB)));
    }
    const int nside = q.cols - reach;
    if (nside <= 0) return VSLAM_OK;
    hipStream_t t = ctx->side;
    VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->ev[par], s));
    VSLAM_HIP_TRY(ctx, hipStreamWaitEvent(t, ctx->ev[par], 0));
    int ks = (mp0 + 255) / 256;
    if (ks > 64) ks = 64;
    GemmDesc gg;                                   // Gram slabs
    gg.M = OB; gg.N = OB; gg.K = mp0; gg.transA = true; gg.transB = false;
    gg.A = q.w.Vbuf[par]; gg.lda = mp0; gg.B = q.w.Vbuf[par]; gg.ldb = mp0;
    gg.C = q.w.Gpart; gg.ldc = OB; gg.ksplit = ks; gg.strideSplit = (size_t)OB * OB;
    if (ks == 1) gg.C = q.w.G[par];
    VSLAM_TRY(gemm_f64_on(ctx, gg, t));
    if (ks > 1) {
        hipLaunchKernelGGL(sum_slabs_kernel, dim3(OB * OB / 256), dim3(256), 0, t, q.w.Gpart, (size_t)OB * OB, ks, (size_t)OB * OB, q.w.G[par]);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
    }
    double *C = q.A + (size_t)reach * q.lda + j0;
    const int nnear = nside < OB ? nside : OB;
    VSLAM_TRY(qr_apply_outer(q, par, mp0, C, nnear, q.w.Wn, q.w.Wnpart, t));
    VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->ev[2 + par], t));
    VSLAM_TRY(qr_apply_outer(q, par, mp0, C + (size_t)nnear * q.lda, nside - nnear, q.w.Wb, q.w.Wbpart, t));
    VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->ev[4 + par], t));
    return VSLAM_OK;
}

// One outer block of `ob` columns at (j0, j0) factored by ONE launch of qr_block_kernel: its panels and the
// updates between them stay in LDS, and the launch first applies the previous block's reflectors to its own
// columns (it starts `ob` rows higher to cover the rows of R they touch), so consecutive launches follow each
// other on the main stream with no small kernels between them.
// Side stream: the block's reflectors as one unit (Gram matrix + the panels' T) update everything beyond the
// next block, overlapping the next launch; when no further resident block follows, everything beyond this one.
// Events: [par] block factored (main -> side); [4+par] its update done (side -> main: the launch two blocks
// later reads those columns and rewrites this parity's reflector buffers).
static vslam_status qr_outer_block_resident(QrJob &q, int j0, int blk, bool next_resident)
{
    vslam_ctx *ctx = q.ctx;
    hipStream_t s = ctx->stream, t = ctx->side;
    // rows of a block from its diagonal origin on: all of them, or as far as the band lets its last column reach
    const int ob = q.pl.bw, par = blk & 1;
    auto block_rows = [&](int j) { const int mp = q.rows - j; return (q.band >= 0 && q.band + ob < mp) ? q.band + ob : mp; };
    const int mp0 = block_rows(j0), mpp = blk > 0 ? block_rows(j0 - ob) : 0;       // this block's rows, the previous block's
    if (blk >= 2) VSLAM_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev[4 + par], 0));
    const unsigned seq0 = 256u * q.launches++;
    if (blk == 0)
        VSLAM_TRY(qr_block_launch(ctx, s, q.pl, q.A + (size_t)j0 * q.lda + j0, q.lda, q.strideA, 1, mp0, ob, ob, seq0,
                                  q.w.Mo[par], 0, q.w.Vbuf[par], mp0, q.w.xbuf));
    else
        VSLAM_TRY(qr_block_launch(ctx, s, q.pl, q.A + (size_t)j0 * q.lda + (j0 - ob), q.lda, q.strideA, 1, mp0 + ob, ob, ob, seq0,
                                  q.w.Mo[par], 0, q.w.Vbuf[par], mp0, q.w.xbuf,
                                  ob, q.w.Vbuf[par ^ 1], mpp, q.w.Mo[par ^ 1], ob / QR_NB, mpp));
    const int first = next_resident ? j0 + 2 * ob : j0 + ob;     // first column the side stream updates
    const int nside = q.cols - first;
    if (nside <= 0) return VSLAM_OK;
    VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->ev[par], s));
    VSLAM_HIP_TRY(ctx, hipStreamWaitEvent(t, ctx->ev[par], 0));
    int ks = (mp0 + 255) / 256;
    if (ks > 64) ks = 64;
    GemmDesc gg;                                   // Gram slabs
    gg.M = ob; gg.N = ob; gg.K = mp0; gg.transA = true; gg.transB = false;
    gg.A = q.w.Vbuf[par]; gg.lda = mp0; gg.B = q.w.Vbuf[par]; gg.ldb = mp0;
    gg.C = q.w.Gpart; gg.ldc = OB; gg.ksplit = ks; gg.strideSplit = (size_t)OB * OB;
    if (ks == 1) gg.C = q.w.G[par];
    VSLAM_TRY(gemm_f64_on(ctx, gg, t));
    if (ks > 1) {
        hipLaunchKernelGGL(sum_slabs_kernel, dim3(OB * OB / 256), dim3(256), 0, t, q.w.Gpart, (size_t)OB * OB, ks, (size_t)OB * OB, q.w.G[par]);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
    }
    VSLAM_TRY(qr_apply_outer(q, par, mp0, q.A + (size_t)first * q.lda + j0, nside, q.w.Wb, q.w.Wbpart, t));
    VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->ev[4 + par], t));
    return VSLAM_OK;
}

// Factorisations whose workgroups wait for one another (qr_block_kernel with more than one workgroup per matrix) take the
// GPU's turn (DeviceTurnGuard).
static vslam_status qr_upper_dev_inner(vslam_ctx *ctx, double *A, int rows, int cols, int lda, size_t strideA, int batch, void *ws,
                                       int band, bool zero_lower);

// Rows of the tallest launch: with a band a panel has band + NB rows; the form that keeps 128-column blocks resident plans
// for the whole height (its launches then simply use fewer workgroups than planned).
static int tallest_panel(int rows, int cols, int batch, int band)
{
    if (band < 0 || use_outer_blocks(rows, cols, batch) || getenv("VSLAM_QR_NO_BAND")) return rows;
    return band + NB < rows ? band + NB : rows;
}

vslam_status qr_upper_dev(vslam_ctx *ctx, double *A, int rows, int cols, int lda, size_t strideA, int batch, void *ws, int band,
                          bool zero_lower)
{
    QrBlockPlan pl;
    const int tall = (rows >= 1 && cols >= 1 && batch >= 1) ? tallest_panel(rows, cols, batch, band) : rows;
    const bool waits = rows >= 1 && cols >= 1 && batch >= 1 &&
                       ((qr_block_plan(tall, batch, block_width(rows, cols, batch), &pl) && pl.G > 1) || coop_rows_per(rows, batch) > 0);
    if (!waits) return qr_upper_dev_inner(ctx, A, rows, cols, lda, strideA, batch, ws, band, zero_lower);
    DeviceTurnGuard turn(ctx);                        // queued behind the previous such factorisation or solve on this GPU
    if (turn.status() != VSLAM_OK) return turn.status();
    return qr_upper_dev_inner(ctx, A, rows, cols, lda, strideA, batch, ws, band, zero_lower);
}

static vslam_status qr_upper_dev_inner(vslam_ctx *ctx, double *A, int rows, int cols, int lda, size_t strideA, int batch, void *ws,
                                       int band, bool zero_lower)
{
    if (rows < 1 || cols < 1 || lda < rows) return fail(ctx, VSLAM_ERR_SHAPE, "qr: bad shape %dx%d ld %d", rows, cols, lda);
    if (batch < 1 || batch > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "qr: batch=%d out of range", batch);
    QrJob q;
    q.ctx = ctx; q.A = A; q.rows = rows; q.cols = cols; q.lda = lda; q.strideA = strideA; q.batch = batch;
    q.tbase = 0; q.rows_per = coop_rows_per(rows, batch);
    q.launches = 0; q.ob = OB;
    const int tall = tallest_panel(rows, cols, batch, band);
    q.band = (band >= 0 && band < rows && !getenv("VSLAM_QR_NO_BAND")) ? band : -1;
    q.use_block = qr_block_plan(tall, batch, block_width(rows, cols, batch), &q.pl) && !getenv("VSLAM_QR_COLUMNWISE");
    qr_carve(ws, rows, cols, batch, &q.w);
    const int kmax = rows < cols ? rows : cols;
    hipStream_t s = ctx->stream;
    // the exchange area of qr_block_kernel: its words carry sequence numbers that start at 1 in every factorisation
    if (q.use_block) VSLAM_HIP_TRY(ctx, hipMemsetAsync(q.w.xbuf, 0, (size_t)batch * q.pl.xbuf_doubles * 8, s));
    if (q.rows_per > 0) {
        const size_t words = (size_t)batch * 3 * CP_GMAX * CP_SLOT * 2;
        VSLAM_HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(q.w.slots), (int)CP_EMPTY32, words, s));
    }
    {
        const int solo_rows = ((((rows < CP_RMAX) ? rows : CP_RMAX) + 63) / 64) * 64;   // largest single-workgroup panel
        const size_t lds = coop_lds_bytes(solo_rows > q.rows_per ? solo_rows : q.rows_per);
        if (lds > 48 * 1024)
            VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&qr_panel_coop_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    int j0 = 0;
    if (use_outer_blocks(rows, cols, batch)) {
        VSLAM_TRY(side_stream_init(ctx));
        int blk = 0;
        const bool resident = q.use_block && q.pl.bw > QR_NB && !getenv("VSLAM_QR_PANELWISE");
        if (!resident) q.band = -1;                  // the panel-wise outer blocks keep whole columns of reflectors: no band there
        if (resident) {
            q.ob = q.pl.bw;
            for (; j0 + q.ob <= kmax; j0 += q.ob, ++blk)
                VSLAM_TRY(qr_outer_block_resident(q, j0, blk, j0 + 2 * q.ob <= kmax));
        } else {
            for (; j0 + OB <= kmax; j0 += OB, ++blk)
                VSLAM_TRY(qr_outer_block(q, j0, blk));
        }
        if (blk > 0) {                                                           // join the side stream
            VSLAM_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev[4], 0));
            VSLAM_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev[5], 0));
        }
    }
    for (; j0 < kmax; j0 += NB) {
        const int jb = (kmax - j0) < NB ? (kmax - j0) : NB;
        VSLAM_TRY(qr_factor_panel(q, j0, jb, q.w.T, nullptr, 0, 0));
        VSLAM_TRY(qr_apply_panel(q, j0, jb, q.w.T, cols - j0 - jb));
    }
    if (zero_lower) {
        dim3 grid((rows + 255) / 256, cols, batch);
        hipLaunchKernelGGL(zero_strict_lower_kernel, grid, dim3(256), 0, s, A, rows, cols, lda, strideA);
    }
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
    VSLAM_TRY(h2d_2d(ctx, dA, (size_t)rows * 8, A, (size_t)lda * 8, (size_t)rows * 8, cols));
    VSLAM_TRY(qr_upper_dev(ctx, dA, rows, cols, rows, 0, 1, qws));
    VSLAM_TRY(d2h_2d(ctx, A, (size_t)lda * 8, dA, (size_t)rows * 8, (size_t)rows * 8, cols));
    return stream_finish(ctx);
}

// Test hook (not part of the public header): copies `count` doubles from the start of the context's workspace
// (where the last factorisation left the compact-WY T of its last panel) to the host.
extern "C" vslam_status vslam_debug_ws_read(vslam_ctx *ctx, double *out, int count)
{
    if (!ctx || !out || count < 0 || (size_t)count * 8 > ctx->ws_bytes) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    VSLAM_HIP_TRY(ctx, hipMemcpy(out, ctx->ws, (size_t)count * 8, hipMemcpyDeviceToHost));
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
    VSLAM_TRY(h2d(ctx, dA, A, ba));
    VSLAM_TRY(h2d(ctx, dB, B, bb));
    VSLAM_TRY(h2d(ctx, dC, C, bc));
    GemmDesc g;
    g.M = M; g.N = N; g.K = K; g.transA = transA != 0; g.transB = transB != 0; g.amode = (AMode)amode; g.b_upper = b_upper != 0;
    g.alpha = alpha; g.beta = beta; g.A = dA; g.lda = lda; g.B = dB; g.ldb = ldb; g.C = dC; g.ldc = ldc;
    if (ks > 1) { g.C = dS; g.ksplit = ks; g.strideSplit = (size_t)ldc * N; }
    VSLAM_TRY(gemm_f64(ctx, g));
    if (ks > 1) VSLAM_TRY(d2h(ctx, C, dS, bc * ks));   // caller sums the slabs
    else VSLAM_TRY(d2h(ctx, C, dC, bc));
    VSLAM_TRY(stream_sync(ctx));
    return VSLAM_OK;
}
