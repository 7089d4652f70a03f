// Data association on the predicted-measurement density (muY, SYY): individualCompatibility,
// jointCompatibility and the greedy surprisal nearest-neighbour search snn of the reference
// (src/dataAssociation.cpp:25-285), one workgroup per filter instance.
//
// The reference re-factors SYY(:, columns of the associated landmarks) by Householder QR for every
// (landmark j, candidate feature i) pair.  All candidates of landmark j share that column set, and the
// set only ever grows at its right end (landmarks are visited in increasing order), so the factor is
// kept incrementally here, right-looking: when a landmark is associated its two reflectors are applied
// at once to the columns of every later landmark in a working copy of SYY, so that when landmark j's
// turn comes its two columns already hold r12 (coupling to the earlier columns, rows < k) and the
// part the two new reflectors act on; for candidate i only the last two entries of
// z = SA^-T (yA - muA) change:
//     z_last(i) = r22^-T (y_i - muY_j - r12^T z_prev),   dM2(i) = |z_prev|^2 + |z_last(i)|^2.
// Same arithmetic as a from-scratch Householder factorisation in LAPACK's column order (the signs of
// the rows of SA do not enter |z| or log|diag|), O(N (n - j)) per landmark instead of O(m N k^2).
//
// The reference builds the list of still-unassigned features with std::set_difference on a range
// that is not sorted (src/dataAssociation.cpp:106); the walk is restated literally (a feature is
// withheld only if its index is not below every index withheld before it), so the same candidates
// are offered.
#include "linalg.hpp"

#include <cmath>
#include <mutex>
#include <vector>

namespace vslam {

constexpr int AT = 1024;                 // threads per workgroup (16 waves)

struct AssocArgs {
    const double *muY; size_t s_mu;      // [2n]
    const double *SYY; int ldS; size_t s_S;
    const double *y; size_t s_y; int m;  // 2 x m, features in columns
    int n;                               // landmarks; N = 2n rows
    const double *lut;                   // chi2inv(c, 2(k+1)), k = 0..n-1 (shared by all instances)
    double sU;
    int enforce, mode;                   // mode 0: snn; 1: jointCompatibility of the idx given in idx
    int32_t *idx; size_t s_idx;          // [n] out (mode 0) / in (mode 1)
    double *s_out;                       // [batch] surprisal
    int32_t *jc_out;                     // [batch] (mode 1) joint compatibility
    double *ws; size_t s_ws;             // per instance: W[N*N] z[N] a[2N] ind[3n] (2N reserved)
    uint8_t *avail; size_t s_av;         // [m]
    int vec_lds;                         // work vectors in LDS instead of ws (set by assoc_launch)
};

__device__ __forceinline__ double wg_sum(double v, double *red)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < AT / 64; ++w) s += red[w];
    return s;
}

// Householder reflector of x[k0..N) (x in global memory): returns beta; v = [1; x_tail * scale] is left in
// x[k0+1..N) (x[k0] is left untouched), tau and scale via out parameters.  LAPACK / Eigen sign convention.
__device__ double make_reflector(double *x, int k0, int N, double *red, double *tau_out)
{
    double part = 0.0;
    for (int r = k0 + 1 + threadIdx.x; r < N; r += AT) part += x[r] * x[r];
    const double tail = wg_sum(part, red);
    const double alpha = x[k0];
    double beta, tau, scale;
    if (tail <= 2.2250738585072014e-308) { beta = alpha; tau = 0.0; scale = 0.0; }
    else {
        beta = sqrt(alpha * alpha + tail);
        if (alpha >= 0.0) beta = -beta;
        scale = 1.0 / (alpha - beta);
        tau = (beta - alpha) / beta;
    }
    __syncthreads();
    for (int r = k0 + 1 + threadIdx.x; r < N; r += AT) x[r] *= scale;
    __syncthreads();
    *tau_out = tau;
    return beta;
}

// y -= tau * (v^T y) v with v = [1 at k0; x[k0+1..N)]
__device__ void apply_reflector(const double *x, double tau, int k0, int N, double *yv, double *red)
{
    double part = (threadIdx.x == 0) ? yv[k0] : 0.0;
    for (int r = k0 + 1 + threadIdx.x; r < N; r += AT) part += x[r] * yv[r];
    const double d = tau * wg_sum(part, red);
    __syncthreads();
    if (threadIdx.x == 0) yv[k0] -= d;
    for (int r = k0 + 1 + threadIdx.x; r < N; r += AT) yv[r] -= d * x[r];
    __syncthreads();
}



// R factor (r00, r01, r11) of the Householder QR of the two columns c0, c1 (N rows, N >= 2), by one wave;
// LAPACK / Eigen reflector convention.  Nothing is stored besides the three numbers.
__device__ __forceinline__ void wave_qr2(const double *c0, const double *c1, int N, double *r00, double *r01, double *r11)
{
    const int lane = threadIdx.x & 63;
    double s00 = 0.0, s01 = 0.0;
    for (int r = 1 + lane; r < N; r += 64) { const double x = c0[r]; s00 += x * x; s01 += x * c1[r]; }
    for (int off = 32; off >= 1; off >>= 1) { s00 += __shfl_xor(s00, off, 64); s01 += __shfl_xor(s01, off, 64); }
    const double alpha = c0[0];
    double beta, tau, scale;
    if (s00 <= 2.2250738585072014e-308) { beta = alpha; tau = 0.0; scale = 0.0; }
    else {
        beta = sqrt(alpha * alpha + s00);
        if (alpha >= 0.0) beta = -beta;
        scale = 1.0 / (alpha - beta);
        tau = (beta - alpha) / beta;
    }
    const double d = tau * (c1[0] + scale * s01);                  // tau * v^T c1
    double t11 = 0.0, a11 = 0.0;
    for (int r = 1 + lane; r < N; r += 64) {
        const double y = c1[r] - d * (scale * c0[r]);
        if (r == 1) a11 = y; else t11 += y * y;
    }
    for (int off = 32; off >= 1; off >>= 1) { t11 += __shfl_xor(t11, off, 64); a11 += __shfl_xor(a11, off, 64); }
    double b1;
    if (t11 <= 2.2250738585072014e-308) b1 = a11;
    else { b1 = sqrt(a11 * a11 + t11); if (a11 >= 0.0) b1 = -b1; }
    *r00 = beta; *r01 = c1[0] - d; *r11 = b1;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wave, work vectors in LDS: the two new reflectors of a landmark's columns c0 = ca[0..N), c1 = ca[N..2N) on rows
// k.. (make_reflector, apply_reflector, make_reflector with the same conventions), the coupling of the two columns
// with z_prev and v1^T v0 for the commit -- the whole serial part of a landmark without a workgroup barrier (a
// workgroup-wide sum costs two barriers of sixteen waves for vectors a single wave holds in a few elements per lane).
// out: beta0, tau0, r01, beta1, tau1, c0 . z, c1 . z, v1^T v0, and the number of leading rows outside which both
// reflectors are zero.  On return ca holds the scaled reflector tails as the workgroup-wide routines leave them.
__device__ void wave_panel(double *ca, const double *z, int k, int N, int rl, double *out)
{
    // rl: both columns are zero from row rl on (rl = N when nothing is known)
    const int lane = threadIdx.x & 63;
    double *c0 = ca, *c1 = ca + N;
    // coupling with z_prev (rows < k are not touched below)
    double p0 = 0.0, p1 = 0.0;
    for (int r = lane; r < k; r += 64) { const double zr = z[r]; p0 += c0[r] * zr; p1 += c1[r] * zr; }
    p0 = wave_sum(p0); p1 = wave_sum(p1);
    // reflector 0 and what it needs of column 1 in one pass
    double t0 = 0.0, d01 = 0.0;
    for (int r = k + 1 + lane; r < rl; r += 64) { const double x = c0[r]; t0 += x * x; d01 += x * c1[r]; }
    t0 = wave_sum(t0); d01 = wave_sum(d01);
    const double alpha0 = c0[k];
    double beta0, tau0, scale0;
    if (t0 <= 2.2250738585072014e-308) { beta0 = alpha0; tau0 = 0.0; scale0 = 0.0; }
    else {
        beta0 = sqrt(alpha0 * alpha0 + t0);
        if (alpha0 >= 0.0) beta0 = -beta0;
        scale0 = 1.0 / (alpha0 - beta0);
        tau0 = (beta0 - alpha0) / beta0;
    }
    const double d = tau0 * (c1[k] + scale0 * d01);            // tau0 * v0^T c1
    const double r01 = c1[k] - d;
    // scale the tail of column 0, apply the reflector to column 1, and gather reflector 1's sums on the way
    double t1 = 0.0;
    int last = k + 1;                                            // last row in which a reflector can be non-zero
    for (int r = k + 1 + lane; r < rl; r += 64) {
        const double v = c0[r] * scale0, y = c1[r] - d * v;
        c0[r] = v; c1[r] = y;
        if (r > k + 1) t1 += y * y;
        if (v != 0.0 || y != 0.0) last = r;
    }
    for (int off = 32; off >= 1; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
    if (lane == 0) c1[k] = r01;
    t1 = wave_sum(t1);
    double beta1, tau1 = 0.0, scale1 = 0.0;
    const double alpha1 = (k + 1 < N) ? c1[k + 1] : 0.0;
    if (k + 1 >= N || t1 <= 2.2250738585072014e-308) beta1 = alpha1;
    else {
        beta1 = sqrt(alpha1 * alpha1 + t1);
        if (alpha1 >= 0.0) beta1 = -beta1;
        scale1 = 1.0 / (alpha1 - beta1);
        tau1 = (beta1 - alpha1) / beta1;
    }
    double ps = 0.0;                                             // v1^T v0 = v0(k+1) + sum_{r >= k+2} v0(r) v1(r)
    for (int r = k + 2 + lane; r < rl; r += 64) { const double v = c1[r] * scale1; c1[r] = v; ps += c0[r] * v; }
    ps = wave_sum(ps) + ((k + 1 < N) ? c0[k + 1] : 0.0);
    if (lane == 0) { out[0] = beta0; out[1] = tau0; out[2] = r01; out[3] = beta1; out[4] = tau1; out[5] = p0; out[6] = p1; out[7] = ps; out[8] = (double)(min(last, N - 1) + 1); }
}

// Before the walk, with the whole chip: the working copy W of SYY, and (mode 0) the individual factors
// (src/dataAssociation.cpp:165-171), which do not depend on the search.  One workgroup per landmark and instance: its
// two columns copied, their 2 x 2 R factor by the first wavefront.  (Left to the walking workgroup, the copy of a
// 4 000 x 4 000 factor and its 2 000 column-pair factors went through one compute unit: most of the call.)
__global__ __launch_bounds__(256) void assoc_prepare_kernel(AssocArgs a)
{
    const int j = blockIdx.x, inst = blockIdx.y, tid = threadIdx.x, N = 2 * a.n;
    const double *SYY = a.SYY + inst * a.s_S;
    double *W = a.ws + inst * a.s_ws;
    const double *c0 = SYY + (size_t)(2 * j) * a.ldS, *c1 = c0 + a.ldS;
    __shared__ int ext_s;
    if (tid == 0) ext_s = 0;
    __syncthreads();
    int last = 0;                                            // 1 + the last row in which either column is non-zero
    for (int r = tid; r < N; r += 256) {
        const double v0 = c0[r], v1 = c1[r];
        W[(size_t)(2 * j) * N + r] = v0; W[(size_t)(2 * j + 1) * N + r] = v1;
        if (v0 != 0.0 || v1 != 0.0) last = r + 1;
    }
    atomicMax(&ext_s, last);
    __syncthreads();
    if (tid == 0) (W + (size_t)N * N + 3 * (size_t)N + 3 * (size_t)a.n)[j] = (double)ext_s;      // ext[j], behind ind
    if (a.mode == 0 && tid < 64) {
        double r00, r01, r11;
        wave_qr2(c0, c1, N, &r00, &r01, &r11);
        double *ind = W + (size_t)N * N + 3 * (size_t)N;
        if (tid == 0) { ind[3 * j] = r00; ind[3 * j + 1] = r01; ind[3 * j + 2] = r11; }
    }
}

__global__ __launch_bounds__(AT) void assoc_kernel(AssocArgs a)
{
    __shared__ double red[AT / 64];
    __shared__ double pn[9];                 // results of wave_panel
    __shared__ double cand_z0[AT / 64], cand_z1[AT / 64];
    __shared__ double cand_v[AT / 64];
    __shared__ int cand_i[AT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, inst = blockIdx.x;
    const int n = a.n, N = 2 * n, m = a.m;
    const double *muY = a.muY + inst * a.s_mu, *y = a.y + inst * a.s_y;
    int32_t *idx = a.idx + inst * a.s_idx;
    extern __shared__ double vecs[];         // the 3 N doubles of work vectors when they fit (a.vec_lds)
    double *W = a.ws + inst * a.s_ws;        // working copy of SYY (N x N, ld N): accepted reflectors already applied
    double *z = a.vec_lds ? vecs : W + (size_t)N * N;
    // ind[3j..3j+2]: R factor of landmark j's two columns on their own (assoc_prepare_kernel), in global memory: it is
    // read once per landmark
    double *ca = z + N;
    const double *ind = W + (size_t)N * N + 3 * (size_t)N, *ext = ind + 3 * (size_t)n;       // ext[j]: rows of landmark j's columns that can be non-zero
    uint8_t *avail = a.avail + inst * a.s_av;
    const double halflog2pi = log(3.14159265358979323846264338327950288 * 2) / 2;

    int k = 0, nA = 0;                       // accepted reflectors / associated landmarks
    int rlim = 0;                            // no column met so far (nor any reflector made from them) reaches row rlim or beyond
    int cursor = 0;                          // state of the offer-list walk (mode 0)
    if (a.mode == 0)
        for (int i = tid; i < m; i += AT) avail[i] = 1;
    __syncthreads();
    double zz = 0.0, logdet = 0.0, s = n * a.sU;
    for (int j = 0; j < n; ++j) {
        const int given = a.mode == 1 ? idx[j] : -2;
        if (a.mode == 1 && given == -1) continue;
        // ---- landmark j's two columns from the working copy (the accepted reflectors are in) ----
        // an upper-triangular factor (what affineTransform returns) has nothing below row 2 j + 1 in these columns, and the
        // reflectors of earlier landmarks, made from shorter columns, put nothing there either
        rlim = a.vec_lds ? min(N, max(rlim, max((int)ext[j], k + 2))) : N;      // the pivots' rows k, k + 1 are always inside
        for (int r = tid; r < rlim; r += AT) { ca[r] = W[(size_t)(2 * j) * N + r]; ca[N + r] = W[(size_t)(2 * j + 1) * N + r]; }
        __syncthreads();
        // ---- two new reflectors on rows k.. (joint factor) ----
        double tau0, tau1, beta0, beta1, r01, c0, c1, s01 = 0.0;
        int rmax = N;                        // rows k .. rmax - 1 are all the new reflectors touch
        if (a.vec_lds) {
            // the vectors live in LDS: one wave does the serial part, everybody else meets it at one barrier
            if (wave == 0) wave_panel(ca, z, k, N, rlim, pn);
            __syncthreads();
            beta0 = pn[0]; tau0 = pn[1]; r01 = pn[2]; beta1 = pn[3]; tau1 = pn[4]; c0 = pn[5]; c1 = pn[6]; s01 = pn[7]; rmax = min((int)pn[8], rlim);
        } else {
            beta0 = make_reflector(ca, k, N, red, &tau0);
            apply_reflector(ca, tau0, k, N, ca + N, red);
            r01 = ca[N + k];
            beta1 = (k + 1 < N) ? make_reflector(ca + N, k + 1, N, red, &tau1) : (tau1 = 0.0, ca[N + k + 1]);
            // coupling of the new columns with z_prev: c_q = r12(:, q)^T z_prev
            double p0 = 0.0, p1 = 0.0;
            for (int r = tid; r < k; r += AT) { p0 += ca[r] * z[r]; p1 += ca[N + r] * z[r]; }
            c0 = wg_sum(p0, red); c1 = wg_sum(p1, red);
        }
        double best_ds = INFINITY, best_z0 = 0.0, best_z1 = 0.0;
        int best_i = -1;
        const double scur = s;
        if (a.mode == 0) {
            const double ib0 = ind[3 * j], i01 = ind[3 * j + 1], ib1 = ind[3 * j + 2];   // individual factor
            // ---- features still on offer: avail[] holds the std::set_difference walk over idx[0..j), kept up
            //      to date as associations are made (see the end of the loop body) ----
            // ---- candidates ----
            double my_ds = INFINITY, my_z0 = 0.0, my_z1 = 0.0;
            int my_i = 0x7fffffff;
            const double mu0 = muY[2 * j], mu1 = muY[2 * j + 1];
            for (int i = tid; i < m; i += AT) {
                if (!avail[i]) continue;
                const double e0 = y[2 * i] - mu0, e1 = y[2 * i + 1] - mu1;
                const double q0 = e0 / ib0, q1 = (e1 - i01 * q0) / ib1;                // SYYj^T \ e  (:172)
                const double zind = q0 * q0 + q1 * q1;
                if (!(zind < a.lut[j])) continue;                                      // :175
                const double z0 = (e0 - c0) / beta0, z1 = (e1 - c1 - r01 * z0) / beta1;
                const double dM2 = zz + (z0 * z0 + z1 * z1);
                if (a.enforce && !(dM2 <= a.lut[nA])) continue;                        // :120-123, :278
                const double s1 = 2 * (nA + 1) * halflog2pi, s2 = logdet + log(fabs(beta0)) + log(fabs(beta1));
                const double snext = ((s1 + s2) + 0.5 * dM2) + (n - nA - 1) * a.sU;    // :233-243
                const double ds = snext - scur;
                if (ds < my_ds) { my_ds = ds; my_i = i; my_z0 = z0; my_z1 = z1; }      // first minimum in index order
            }
            // workgroup arg-min with the lowest index winning ties
            for (int off = 32; off >= 1; off >>= 1) {
                const double ods = __shfl_xor(my_ds, off, 64), oz0 = __shfl_xor(my_z0, off, 64), oz1 = __shfl_xor(my_z1, off, 64);
                const int oi = __shfl_xor(my_i, off, 64);
                if (ods < my_ds || (ods == my_ds && oi < my_i)) { my_ds = ods; my_i = oi; my_z0 = oz0; my_z1 = oz1; }
            }
            // every wave's best to LDS, then every thread picks the winner itself (the arrays are written again only
            // behind the barrier that ends this landmark)
            if (lane == 0) { cand_v[wave] = my_ds; cand_i[wave] = my_i; cand_z0[wave] = my_z0; cand_z1[wave] = my_z1; }
            __syncthreads();
            {
                int bw = 0;
#pragma unroll
                for (int q = 1; q < AT / 64; ++q)
                    if (cand_v[q] < cand_v[bw] || (cand_v[q] == cand_v[bw] && cand_i[q] < cand_i[bw])) bw = q;
                if (cand_v[bw] < INFINITY) { best_ds = cand_v[bw]; best_z0 = cand_z0[bw]; best_z1 = cand_z1[bw]; best_i = cand_i[bw]; }
            }
            // landmark j left unassociated (:133-142)
            const double sA = nA > 0 ? (2 * nA * halflog2pi + logdet) + 0.5 * zz : 0.0;
            const double su = sA + (n - nA) * a.sU;
            if (su - scur < best_ds) { best_i = -1; s = su; }
            else s = scur + best_ds;     // exact: snext is recomputed below when the association is committed
        } else {
            const double e0 = y[2 * given] - muY[2 * j], e1 = y[2 * given + 1] - muY[2 * j + 1];
            best_i = given;
            best_z0 = (e0 - c0) / beta0;
            best_z1 = (e1 - c1 - r01 * best_z0) / beta1;
        }
        if (a.mode == 0) {
            // The reference rebuilds the offer list for every landmark by walking idx[0..j) with one cursor; the
            // walk over a prefix never changes, so one step per landmark continues it: idx[j] is withheld from
            // later landmarks only if it is not below the cursor.
            if (best_i >= cursor && best_i < m) {
                if (tid == 0) avail[best_i] = 0;
                cursor = best_i + 1;
            }
            if (tid == 0) idx[j] = best_i;
        }
        if (best_i < 0) { __syncthreads(); continue; }
        // ---- commit: the two reflectors go into every later landmark's columns; z and the running sums grow by two.
        // v0 = [1 at k; ca(k+1..)], v1 = [1 at k+1; ca(N + k+2..)].  With s01 = v1^T v0 both act in one read pass
        // and one update pass per column:  a -= tau0 d0 v0 + tau1 (e1 - tau0 d0 s01) v1,  d0 = v0^T a, e1 = v1^T a.
        // A factor that arrives upper triangular (what affineTransform returns) keeps the work small: while no landmark
        // has been left out both reflectors are the identity (tau = 0: nothing to apply), and after s omissions they are
        // zero below row 2 j + 1, so the passes stop there. ----
        if (tau0 != 0.0 || tau1 != 0.0) {
            if (!a.vec_lds) {
                double ps = (tid == 0 && k + 1 < N) ? ca[k + 1] : 0.0;                   // v1(k+1) = 1 times v0(k+1)
                for (int r = k + 2 + tid; r < N; r += AT) ps += ca[r] * ca[N + r];
                s01 = wg_sum(ps, red);
            }
            constexpr int CG = 4;                                                        // columns a wave handles at once
            for (int c0 = 2 * (j + 1) + wave * CG; c0 < N; c0 += (AT / 64) * CG) {
                double d0[CG], e1[CG];
#pragma unroll
                for (int q = 0; q < CG; ++q) { d0[q] = 0.0; e1[q] = 0.0; }
                for (int r = k + lane; r < rmax; r += 64) {
                    const double v0 = (r == k) ? 1.0 : ca[r];
                    const double v1 = (r == k) ? 0.0 : ((r == k + 1) ? 1.0 : ca[N + r]);
                    double av[CG];
#pragma unroll
                    for (int q = 0; q < CG; ++q) av[q] = (c0 + q < N) ? W[(size_t)(c0 + q) * N + r] : 0.0;
#pragma unroll
                    for (int q = 0; q < CG; ++q) { d0[q] += v0 * av[q]; e1[q] += v1 * av[q]; }
                }
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    for (int off = 32; off >= 1; off >>= 1) { d0[q] += __shfl_xor(d0[q], off, 64); e1[q] += __shfl_xor(e1[q], off, 64); }
                    d0[q] *= tau0;
                    e1[q] = tau1 * (e1[q] - d0[q] * s01);
                }
                for (int r = k + lane; r < rmax; r += 64) {
                    const double v0 = (r == k) ? 1.0 : ca[r];
                    const double v1 = (r == k) ? 0.0 : ((r == k + 1) ? 1.0 : ca[N + r]);
                    double av[CG];
#pragma unroll
                    for (int q = 0; q < CG; ++q) av[q] = (c0 + q < N) ? W[(size_t)(c0 + q) * N + r] : 0.0;
#pragma unroll
                    for (int q = 0; q < CG; ++q)
                        if (c0 + q < N) W[(size_t)(c0 + q) * N + r] = av[q] - (d0[q] * v0 + e1[q] * v1);
                }
            }
        }
        if (tid == 0) { z[k] = best_z0; z[k + 1] = best_z1; }
        zz += best_z0 * best_z0 + best_z1 * best_z1;
        logdet += log(fabs(beta0)) + log(fabs(beta1));
        k += 2; nA += 1;
        if (a.mode == 0) s = ((2 * nA * halflog2pi + logdet) + 0.5 * zz) + (n - nA) * a.sU;
        __syncthreads();
    }
    if (a.mode == 1) {
        const double sA = nA > 0 ? (2 * nA * halflog2pi + logdet) + 0.5 * zz : 0.0;
        s = sA + (n - nA) * a.sU;
        if (tid == 0) a.jc_out[inst] = nA > 0 ? (zz <= a.lut[nA - 1]) : 1;
    }
    if (tid == 0) a.s_out[inst] = s;
}

// individualCompatibility for every (feature i, landmark j): out[i * n + j].  One workgroup per landmark.
__global__ __launch_bounds__(AT) void individual_compat_kernel(const double *y, int m, const double *muY, const double *SYY,
                                                               int ldS, int n, const double *lut, double *cols, uint8_t *out)
{
    __shared__ double red[AT / 64];
    const int j = blockIdx.x, N = 2 * n, tid = threadIdx.x;
    double *cb = cols + (size_t)j * 2 * N;
    for (int r = tid; r < 2 * N; r += AT) cb[r] = SYY[(size_t)(2 * j + r / N) * ldS + (r % N)];
    __syncthreads();
    double t0, t1;
    const double b0 = make_reflector(cb, 0, N, red, &t0);
    apply_reflector(cb, t0, 0, N, cb + N, red);
    const double r01 = cb[N];
    const double b1 = make_reflector(cb + N, 1, N, red, &t1);
    for (int i = tid; i < m; i += AT) {
        const double e0 = y[2 * i] - muY[2 * j], e1 = y[2 * i + 1] - muY[2 * j + 1];
        const double q0 = e0 / b0, q1 = (e1 - r01 * q0) / b1;
        out[(size_t)i * n + j] = (q0 * q0 + q1 * q1) < lut[j];
    }
}

// ---- host helpers: normcdf / chi2inv (src/utility.cpp:13-24) ----
static double gamma_p_host(double a, double x)
{
    if (x <= 0.0) return 0.0;
    const double lg = std::lgamma(a);
    if (x < a + 1.0) {                                  // series
        double ap = a, term = 1.0 / a, sum = term;
        for (int it = 0; it < 100000; ++it) {
            ap += 1.0; term *= x / ap; sum += term;
            if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
        }
        return sum * std::exp(-x + a * std::log(x) - lg);
    }
    const double tiny = 1e-300;                         // modified Lentz for Q(a, x)
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    for (int i = 1; i < 100000; ++i) {
        const double an = -i * (i - a);
        b += 2.0;
        d = an * d + b; if (std::fabs(d) < tiny) d = tiny;
        c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < 1e-17) break;
    }
    return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}

double chi2inv_host(double p, double nu)
{
    const double a = nu / 2.0;
    if (!(p > 0.0)) return 0.0;
    double lo = 0.0, hi = a > 1.0 ? a : 1.0;
    while (gamma_p_host(a, hi) < p) { lo = hi; hi *= 2.0; }
    double x = 0.5 * (lo + hi);
    for (int it = 0; it < 200; ++it) {
        const double f = gamma_p_host(a, x) - p;
        if (f > 0.0) hi = x; else lo = x;
        const double dens = std::exp(-x + (a - 1.0) * std::log(x) - std::lgamma(a));
        double xn = x - f / dens;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        const bool done = std::fabs(xn - x) <= 4e-16 * std::fabs(xn);
        x = xn;
        if (done) break;
    }
    return 2.0 * x;
}

double normcdf_host(double z) { return 0.5 * std::erfc(-z / std::sqrt(2.0)); }

size_t assoc_ws_doubles(int n) { const size_t N = 2 * (size_t)n; return N * N + 5 * N + 64; }

// Queues the association kernel for `batch` instances; every pointer is device memory.
vslam_status assoc_launch(vslam_ctx *ctx, AssocArgs a, int batch)
{
    // The work vectors (z and one column pair: 3 N doubles) live in LDS when they fit: every phase
    // boundary is then an LDS round trip instead of one through L2.
    const size_t vec_bytes = (size_t)3 * 2 * a.n * sizeof(double);
    a.vec_lds = vec_bytes <= 144 * 1024;
    const size_t lds = a.vec_lds ? vec_bytes : 0;
    if (lds > 48 * 1024)
        VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&assoc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(assoc_prepare_kernel, dim3(a.n, batch), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(assoc_kernel, dim3(batch), dim3(AT), lds, ctx->stream, a);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" double vslam_normcdf(double z) { return normcdf_host(z); }
extern "C" double vslam_chi2inv(double p, double nu) { return chi2inv_host(p, nu); }

namespace {
// chi2 LUT of snn (src/dataAssociation.cpp:27-29, 63-69) on the device.
vslam_status upload_lut(vslam_ctx *ctx, int n, double *dlut)
{
    // The gates chi2inv(c, 2 (k + 1)) depend on nothing but k: computed once per process (each is an iteration on the
    // incomplete gamma function; at 2 000 landmarks they took longer than the kernel) and kept at a fixed address, so
    // the copy below may still be in flight when this function returns.
    constexpr int kLutMax = 32768;
    static double table[kLutMax];
    static int filled = 0;
    static std::mutex mtx;
    if (n > kLutMax) {
        std::vector<double> lut((size_t)n);
        const double c = 2 * normcdf_host(3.0) - 1;
        for (int k = 0; k < n; ++k) lut[(size_t)k] = chi2inv_host(c, 2.0 * (k + 1));
        VSLAM_TRY(h2d(ctx, dlut, lut.data(), (size_t)n * 8));
        return stream_sync(ctx);                  // lut goes out of scope
    }
    {
        std::lock_guard<std::mutex> lock(mtx);
        if (filled < n) {
            const double c = 2 * normcdf_host(3.0) - 1;
            for (int k = filled; k < n; ++k) table[k] = chi2inv_host(c, 2.0 * (k + 1));
            filled = n;
        }
    }
    return h2d(ctx, dlut, table, (size_t)n * 8);
}
}  // namespace

extern "C" vslam_status vslam_individual_compat(vslam_ctx *ctx, const double *y, int m, const double *muY, const double *SYY,
                                                int n, int ldS, const double *chi2LUT, uint8_t *compat)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!y || !muY || !SYY || !chi2LUT || !compat) return fail(ctx, VSLAM_ERR_ARG, "individual_compat: null pointer");
    if (n < 1 || m < 1 || ldS < 2 * n) return fail(ctx, VSLAM_ERR_SHAPE, "individual_compat: n=%d m=%d ldS=%d", n, m, ldS);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t N = 2 * (size_t)n;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(2 * m * 8) + Arena::pad(N * 8) + Arena::pad(N * N * 8) + Arena::pad(n * 8) +
                                  Arena::pad(2 * N * n * 8) + Arena::pad((size_t)m * n)));
    Arena ar(ctx->ws);
    double *dy = ar.take<double>(2 * (size_t)m), *dmu = ar.take<double>(N), *dS = ar.take<double>(N * N), *dl = ar.take<double>(n);
    double *cols = ar.take<double>(2 * N * n);
    uint8_t *dout = ar.take<uint8_t>((size_t)m * n);
    VSLAM_TRY(h2d(ctx, dy, y, 2 * (size_t)m * 8));
    VSLAM_TRY(h2d(ctx, dmu, muY, N * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, N * 8, SYY, (size_t)ldS * 8, N * 8, N));
    VSLAM_TRY(h2d(ctx, dl, chi2LUT, (size_t)n * 8));
    hipLaunchKernelGGL(individual_compat_kernel, dim3(n), dim3(AT), 0, ctx->stream, dy, m, dmu, dS, (int)N, n, dl, cols, dout);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(d2h(ctx, compat, dout, (size_t)m * n));
    return stream_finish(ctx);
}

namespace {
// Shared host-pointer driver of jointCompatibility (mode 1) and snn on a density (mode 0).
vslam_status assoc_host(vslam_ctx *ctx, int mode, const double *muY, const double *SYY, int n, int ldS, const double *y, int m,
                        const double *chi2LUT, double sU, int enforce, int32_t *idx, double *surprisal, int32_t *jc)
{
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t N = 2 * (size_t)n;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(2 * (size_t)m * 8) + Arena::pad(N * 8) + Arena::pad(N * N * 8) + Arena::pad((size_t)n * 8) +
                                  Arena::pad((size_t)n * 4) + Arena::pad(64) + Arena::pad(assoc_ws_doubles(n) * 8) + Arena::pad((size_t)m)));
    Arena ar(ctx->ws);
    AssocArgs a;
    double *dy = ar.take<double>(2 * (size_t)m), *dmu = ar.take<double>(N), *dS = ar.take<double>(N * N), *dl = ar.take<double>(n);
    int32_t *didx = ar.take<int32_t>(n);
    double *dout = ar.take<double>(4);
    a.ws = ar.take<double>(assoc_ws_doubles(n)); a.s_ws = 0;
    a.avail = ar.take<uint8_t>(m); a.s_av = 0;
    VSLAM_TRY(h2d(ctx, dy, y, 2 * (size_t)m * 8));
    VSLAM_TRY(h2d(ctx, dmu, muY, N * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, N * 8, SYY, (size_t)ldS * 8, N * 8, N));
    if (chi2LUT) VSLAM_TRY(h2d(ctx, dl, chi2LUT, (size_t)n * 8));
    else VSLAM_TRY(upload_lut(ctx, n, dl));
    if (mode == 1) VSLAM_TRY(h2d(ctx, didx, idx, (size_t)n * 4));
    a.muY = dmu; a.s_mu = 0; a.SYY = dS; a.ldS = (int)N; a.s_S = 0; a.y = dy; a.s_y = 0; a.m = m; a.n = n; a.lut = dl; a.sU = sU;
    a.enforce = enforce; a.mode = mode; a.idx = didx; a.s_idx = 0; a.s_out = dout; a.jc_out = reinterpret_cast<int32_t *>(dout + 2);
    VSLAM_TRY(assoc_launch(ctx, a, 1));
    if (mode == 0) VSLAM_TRY(d2h(ctx, idx, didx, (size_t)n * 4));
    VSLAM_TRY(d2h(ctx, surprisal, dout, 8));
    if (jc) VSLAM_TRY(d2h(ctx, jc, dout + 2, 4));
    return stream_finish(ctx);
}
}  // namespace

extern "C" vslam_status vslam_joint_compat(vslam_ctx *ctx, const int32_t *idx, int n, double sU, const double *y, int m,
                                           const double *muY, const double *SYY, int ldS, const double *chi2LUT,
                                           double *surprisal, int32_t *compatible)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!idx || !y || !muY || !SYY || !chi2LUT || !surprisal || !compatible) return fail(ctx, VSLAM_ERR_ARG, "joint_compat: null pointer");
    if (n < 1 || m < 1 || ldS < 2 * n) return fail(ctx, VSLAM_ERR_SHAPE, "joint_compat: n=%d m=%d ldS=%d", n, m, ldS);
    for (int j = 0; j < n; ++j)
        if (idx[j] < -1 || idx[j] >= m) return fail(ctx, VSLAM_ERR_ARG, "joint_compat: idx[%d]=%d outside [-1, %d)", j, idx[j], m);
    return assoc_host(ctx, 1, muY, SYY, n, ldS, y, m, chi2LUT, sU, 0, const_cast<int32_t *>(idx), surprisal, compatible);
}

extern "C" vslam_status vslam_snn_density(vslam_ctx *ctx, const double *muY, const double *SYY, int n, int ldS, const double *y, int m,
                                          int width, int height, int enforce_jc, int32_t *idx, double *surprisal)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!y || !muY || !SYY || !idx || !surprisal) return fail(ctx, VSLAM_ERR_ARG, "snn_density: null pointer");
    if (n < 1 || m < 1 || ldS < 2 * n || width < 1 || height < 1) return fail(ctx, VSLAM_ERR_SHAPE, "snn_density: n=%d m=%d ldS=%d", n, m, ldS);
    const double sU = std::log((double)width) + std::log((double)height);      // src/dataAssociation.cpp:84
    return assoc_host(ctx, 0, muY, SYY, n, ldS, y, m, nullptr, sU, enforce_jc, idx, surprisal, nullptr);
}

extern "C" vslam_status vslam_snn_density_batch_dev(vslam_ctx *ctx, int batch, const double *muY, size_t stride_mu,
                                                    const double *SYY, int ldS, size_t stride_S, int n,
                                                    const double *y, size_t stride_y, int m, int width, int height,
                                                    int enforce_jc, int32_t *idx, double *surprisal)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!y || !muY || !SYY || !idx || !surprisal) return fail(ctx, VSLAM_ERR_ARG, "snn_density_batch: null pointer");
    if (batch < 1 || n < 1 || m < 1 || ldS < 2 * n || width < 1 || height < 1)
        return fail(ctx, VSLAM_ERR_SHAPE, "snn_density_batch: batch=%d n=%d m=%d ldS=%d", batch, n, m, ldS);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t wsd = (assoc_ws_doubles(n) + 31) & ~size_t(31), av = ((size_t)m + 255) & ~size_t(255);
    VSLAM_TRY(ws_reserve(ctx, Arena::pad((size_t)n * 8) + Arena::pad((size_t)batch * wsd * 8) + Arena::pad((size_t)batch * av)));
    Arena ar(ctx->ws);
    AssocArgs a;
    double *dl = ar.take<double>(n);
    a.ws = ar.take<double>((size_t)batch * wsd); a.s_ws = wsd;
    a.avail = ar.take<uint8_t>((size_t)batch * av); a.s_av = av;
    VSLAM_TRY(upload_lut(ctx, n, dl));
    a.muY = muY; a.s_mu = stride_mu; a.SYY = SYY; a.ldS = ldS; a.s_S = stride_S; a.y = y; a.s_y = stride_y; a.m = m; a.n = n;
    a.lut = dl; a.sU = std::log((double)width) + std::log((double)height);
    a.enforce = enforce_jc; a.mode = 0; a.idx = idx; a.s_idx = (size_t)n; a.s_out = surprisal; a.jc_out = nullptr;
    return assoc_launch(ctx, a, batch);
}

extern "C" vslam_status vslam_affine_bundle(vslam_ctx *ctx, const vslam_camera *cam, const double *mux,
                                            const double *Sxx, int n, int ldS, int layout,
                                            double *muy, double *Syy, int ldSyy, int32_t *code);

extern "C" vslam_status vslam_snn(vslam_ctx *ctx, const vslam_camera *cam, const double *mux, const double *Sxx, int nx_all, int ldS,
                                  int layout, const double *y, int m, int enforce_jc, int32_t *idx, double *surprisal)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!cam || !mux || !Sxx || !y || !idx || !surprisal) return fail(ctx, VSLAM_ERR_ARG, "snn: null pointer");
    if (nx_all < 15 || (nx_all - 12) % 3 != 0 || ldS < nx_all || m < 1) return fail(ctx, VSLAM_ERR_SHAPE, "snn: nx=%d m=%d", nx_all, m);
    const int n = (nx_all - 12) / 3, N = 2 * n;
    std::vector<double> muY((size_t)N), SYY((size_t)N * N);
    VSLAM_TRY(vslam_affine_bundle(ctx, cam, mux, Sxx, nx_all, ldS, layout, muY.data(), SYY.data(), N, nullptr));   // :50-61
    return vslam_snn_density(ctx, muY.data(), SYY.data(), n, N, y, m, cam->width, cam->height, enforce_jc, idx, surprisal);
}
