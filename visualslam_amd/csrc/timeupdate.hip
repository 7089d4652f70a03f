// Time update of the square-root filter: timeUpdateContinuous (reference src/gaussian.hpp:337-383)
// with SlamProcessModel (src/model.cpp:9-136) and the RK4 step with Jacobian propagation
// (RK4SDEHelper, src/gaussian.hpp:171-239).
//
// The reference pushes the augmented state [x; dw] (2n) through the RK4 map, multiplies the
// 2n x 2n factor blkdiag(S, SQ sqrt(dt)) by the n x 2n Jacobian and takes the R of a dense
// 3n x n Householder QR.  The process model only moves the camera pose from the body velocities
// and the attitude, so
//   * f and its Jacobian live in a 12-state (nu, eta); every landmark row of the RK4 Jacobian is
//     [e_i, e_i]: the RK4 recursion is done on a 12 x 25 array by one small workgroup;
//   * S Jx^T differs from S only in columns 6..11 (eta), each a combination of S's first 12 columns;
//   * SQ has six non-zero rows, so of the lower 2n rows of the reference's matrix six survive.
// The QR therefore runs on an (n + 6) x n matrix with the same R (rows of zeros change neither the
// reflectors nor their signs).  A general square S takes the blocked Householder kernels of qr.hip.
// An upper-triangular S -- every factor the filter itself produces -- leaves that matrix triangular
// apart from its leading 12 x 12 block and the six noise rows, so column j's reflector only involves
// row j and the noise rows: tu_qr_kernel walks the columns in panels of 16 with one workgroup per
// filter, the same Householder reflectors in the same order at O(n^2) instead of O(n^3) work.
#include "linalg.hpp"

namespace vslam {

struct TuNoise {
    double sq6[36];      // blkdiag(position_tune, orientation_tune) * sqrt(dt), 6 x 6 column-major
    double dt;
};

constexpr int TU_COLS = 25;      // [state, d/dx(0..11), d/ddw(0..11)]

// f(6..11) and the rows 6..11 of df/dx (6 x 12, row-major in Jm) at the 12-state x.
__device__ void process_model12(const double *x, double *f6, double (*Jm)[12])
{
    const double phi = x[9], theta = x[10], psi = x[11];
    const double sph = sin(phi), cph = cos(phi), sth = sin(theta), cth = cos(theta), sps = sin(psi), cps = cos(psi);
    const double tth = tan(theta);
    // Rnc = Rz(psi) Ry(theta) Rx(phi)  (src/rotation.hpp:22-63)
    const double R[3][3] = {{cps * cth, cps * sth * sph - sps * cph, cps * sth * cph + sps * sph},
                            {sps * cth, sps * sth * sph + cps * cph, sps * sth * cph - cps * sph},
                            {-sth, cth * sph, cth * cph}};
    const double T[3][3] = {{1.0, sph * tth, cph * tth}, {0.0, cph, -sph}, {0.0, sph / cth, cph / cth}};
    for (int r = 0; r < 6; ++r)
        for (int k = 0; k < 12; ++k) Jm[r][k] = 0.0;
    for (int r = 0; r < 3; ++r) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 3; ++k) {
            a += R[r][k] * x[k];
            b += T[r][k] * x[3 + k];
            Jm[r][k] = R[r][k];
            Jm[3 + r][3 + k] = T[r][k];
        }
        f6[r] = a; f6[3 + r] = b;
    }
    const double x1dot = x[0], x2dot = x[1], x3dot = x[2], thetadot = x[4], psidot = x[5];
    // d f / d (phi, theta, psi), src/model.cpp:117-132
    Jm[0][9] = x2dot * (sph * sps + cph * cps * sth) + x3dot * (cph * sps - cps * sph * sth);
    Jm[1][9] = -x2dot * (cps * sph - cph * sps * sth) - x3dot * (cph * cps + sph * sps * sth);
    Jm[2][9] = x2dot * cph * cth - x3dot * cth * sph;
    Jm[3][9] = thetadot * cph * tth - psidot * sph * tth;
    Jm[4][9] = -psidot * cph - thetadot * sph;
    Jm[5][9] = (thetadot * cph) / cth - (psidot * sph) / cth;
    Jm[0][10] = x3dot * cph * cps * cth - x1dot * cps * sth + x2dot * cps * cth * sph;
    Jm[1][10] = x3dot * cph * cth * sps - x1dot * sps * sth + x2dot * cth * sph * sps;
    Jm[2][10] = -x1dot * cth - x3dot * cph * sth - x2dot * sph * sth;
    Jm[3][10] = psidot * cph * (tth * tth + 1.0) + thetadot * sph * (tth * tth + 1.0);
    Jm[4][10] = 0.0;
    Jm[5][10] = (psidot * cph * sth) / (cth * cth) + (thetadot * sph * sth) / (cth * cth);
    Jm[0][11] = x3dot * (cps * sph - cph * sps * sth) - x2dot * (cph * cps + sph * sps * sth) - x1dot * cth * sps;
    Jm[1][11] = x3dot * (sph * sps + cph * cps * sth) - x2dot * (cph * sps - cps * sph * sth) + x1dot * cps * cth;
}

// One workgroup per filter instance; lane c < 25 carries column c of the 12 x 25 RK4 arrays
// (X = [x, I, 0], dW = [0, 0, I]).  Writes the propagated mean and the 12 x 24 Jacobian block
// [d/dx(0..11), d/ddw(0..11)] (column-major, 12 rows) to jac.
__global__ __launch_bounds__(64) void tu_rk4_kernel(const double *mu, size_t stride_mu, int n, double dt,
                                                    double *mu_out, size_t stride_out, double *jac)
{
    __shared__ double xs[12];
    const int c = threadIdx.x, inst = blockIdx.x;
    const double *m = mu + (size_t)inst * stride_mu;
    double *mo = mu_out + (size_t)inst * stride_out;
    for (int i = 12 + c; i < n; i += 64) mo[i] = m[i];                 // landmarks do not move
    double X[12], dW[12], Xs[12], F[4][12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        X[i] = (c == 0) ? m[i] : ((c == i + 1) ? 1.0 : 0.0);
        dW[i] = (c == i + 13) ? 1.0 : 0.0;
        Xs[i] = X[i];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        __syncthreads();
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) xs[i] = Xs[i];
        }
        __syncthreads();
        double xl[12], f6[6], Jm[6][12];
#pragma unroll
        for (int i = 0; i < 12; ++i) xl[i] = xs[i];
        process_model12(xl, f6, Jm);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            F[s][r] = 0.0;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 12; ++k) a += Jm[r][k] * Xs[k];
            F[s][6 + r] = (c == 0) ? f6[r] : a;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (s < 2) Xs[i] = X[i] + (F[s][i] * dt + dW[i]) / 2;
            else if (s == 2) Xs[i] = X[i] + F[s][i] * dt + dW[i];
        }
    }
    if (c >= TU_COLS) return;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const double v = X[i] + (F[0][i] + 2 * F[1][i] + 2 * F[2][i] + F[3][i]) * dt / 6 + dW[i];
        if (c == 0) mo[i] = v;
        else jac[(size_t)inst * 288 + (size_t)(c - 1) * 12 + i] = v;
    }
}

// A = [S Jx^T ; sqrt(dt) SQ6 Jw^T] ((n + 6) x n, leading dimension lda) from S and the 12 x 24 block.
__global__ __launch_bounds__(256) void tu_build_kernel(const double *S, int ldS, size_t strideS, int n, const double *jac,
                                                       TuNoise nz, double *A, int lda, size_t strideA)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, inst = blockIdx.z;
    if (r >= n + 6) return;
    const double *Si = S + (size_t)inst * strideS;
    const double *J = jac + (size_t)inst * 288;
    double v;
    if (r < n) {
        if (c < 6 || c >= 12) v = Si[(size_t)c * ldS + r];
        else {
            v = 0.0;
            for (int k = 0; k < 12; ++k) v += Si[(size_t)k * ldS + r] * J[k * 12 + c];
        }
    } else {
        v = 0.0;
        if (c < 12) {
            const int rr = r - n;
            for (int k = 0; k < 6; ++k) v += nz.sq6[rr + 6 * k] * J[144 + k * 12 + c];
        }
    }
    A[(size_t)inst * strideA + (size_t)c * lda + r] = v;
}

__global__ void tu_copy_out_kernel(const double *A, int lda, size_t strideA, int n, double *S_out, int ldo, size_t strideO)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, inst = blockIdx.z;
    if (r < n) S_out[(size_t)inst * strideO + (size_t)c * ldo + r] = A[(size_t)inst * strideA + (size_t)c * lda + r];
}

// ---- upper-triangular S: structured path ----
constexpr int TQ_W = 16;             // panel width
constexpr int TQ_H = TQ_W + 6;       // active rows of a panel: its diagonal block and the six noise rows
constexpr int TQ_T = 512;

// Row r of the new columns 6..11 of S' = S Jx^T into S_out (a thread reads its row before it writes it, so
// S_out may be S), and the noise rows N = sqrt(dt) SQ6 Jw^T (6 x n column-major, zero beyond column 11).
__global__ __launch_bounds__(256) void tu_build_upper_kernel(const double *S, int ldS, size_t strideS, int n, const double *jac,
                                                             TuNoise nz, double *S_out, int ldo, size_t strideO, double *Nws)
{
    const int r = blockIdx.x * 256 + threadIdx.x, inst = blockIdx.y;
    if (r >= n) return;
    const double *Si = S + (size_t)inst * strideS;
    double *So = S_out + (size_t)inst * strideO;
    const double *J = jac + (size_t)inst * 288;
    double row[12], out[6];
#pragma unroll
    for (int k = 0; k < 12; ++k) row[k] = (r <= k) ? Si[(size_t)k * ldS + r] : 0.0;      // upper triangle only
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) v += row[k] * J[k * 12 + 6 + c];
        out[c] = v;
    }
    if (r < 12) {
#pragma unroll
        for (int c = 0; c < 6; ++c) So[(size_t)(6 + c) * ldo + r] = out[c];
    }
    double *Nn = Nws + (size_t)inst * 6 * n;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
        double v = 0.0;
        if (r < 12)
            for (int k = 0; k < 6; ++k) v += nz.sq6[rr + 6 * k] * J[144 + k * 12 + r];
        Nn[(size_t)6 * r + rr] = v;                                 // here r is the column of N
    }
}

// S_out = upper triangle of S for every column except 6..11 of the first 12 rows' block (written by the
// build kernel); zeros below the diagonal outside the leading 12 x 12 block.  Skipped when S_out is S.
__global__ __launch_bounds__(256) void tu_copy_upper_kernel(const double *S, int ldS, size_t strideS, int n, double *S_out, int ldo,
                                                            size_t strideO)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, inst = blockIdx.z;
    if (r >= n) return;
    if (c >= 6 && c < 12 && r < 12) return;
    S_out[(size_t)inst * strideO + (size_t)c * ldo + r] = (r <= c) ? S[(size_t)inst * strideS + (size_t)c * ldS + r] : 0.0;
}

// ---- Householder QR of [S'; N] in place on S_out (R) with N in Nws, in panels of TQ_W columns ----
// A panel's active rows are its 16 x 16 diagonal block and the six noise rows (22 x 16).
// Factorisation: one wave, lane c < 16 holding column c in registers; at step k lane k's tail norm and
// pivot are broadcast with v_readlane, the reflector entries likewise, and every later lane updates its
// own column -- no LDS, no barrier.  Afterwards lane c holds R above and the reflector below the diagonal.
// Application: every later column of the matrix (its 16 rows of the block row and its six noise
// entries, in registers, one column per thread) receives the sixteen reflectors in order, the
// reflector entries read from LDS as broadcasts.  LAPACK / Eigen reflector convention throughout.
__device__ __forceinline__ void tq_factor_wave(double (&pc)[TQ_H], int lane, int w, double &tau_own)
{
    tau_own = 0.0;
#pragma unroll
    for (int k = 0; k < TQ_W; ++k) {
        if (k < w) {
        double t2 = 0.0;
#pragma unroll
        for (int i = k + 1; i < TQ_H; ++i) t2 += pc[i] * pc[i];
        const double tail = readlane_f64(t2, k), alpha = readlane_f64(pc[k], k);
        double beta, tau, scale;
        if (tail <= 2.2250738585072014e-308) { beta = alpha; tau = 0.0; scale = 0.0; }
        else {
            beta = sqrt(alpha * alpha + tail);
            if (alpha >= 0.0) beta = -beta;
            scale = 1.0 / (alpha - beta);
            tau = (beta - alpha) / beta;
        }
        double v[TQ_H];
#pragma unroll
        for (int i = k + 1; i < TQ_H; ++i) v[i] = readlane_f64(pc[i], k) * scale;
        double dsum = pc[k];
#pragma unroll
        for (int i = k + 1; i < TQ_H; ++i) dsum += v[i] * pc[i];
        dsum *= tau;
        const bool later = lane > k, own = lane == k;
        pc[k] = own ? beta : (later ? pc[k] - dsum : pc[k]);
#pragma unroll
        for (int i = k + 1; i < TQ_H; ++i) pc[i] = own ? v[i] : (later ? pc[i] - dsum * v[i] : pc[i]);
        if (own) tau_own = tau;
        }
    }
}

// Loads column j0 + lane of the panel (lanes >= w and rows past the matrix read zeros).
__device__ __forceinline__ void tq_load_panel_column(const double *A, int ldo, const double *Nn, int j0, int w, int lane, double (&pc)[TQ_H])
{
    const bool in = lane < w;
    const double *col = A + (size_t)(j0 + (in ? lane : 0)) * ldo + j0;
    const double *nc = Nn + (size_t)6 * (j0 + (in ? lane : 0));
#pragma unroll
    for (int i = 0; i < TQ_W; ++i) pc[i] = (in && i < w) ? col[i] : 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) pc[TQ_W + r] = in ? nc[r] : 0.0;
}

// Columns [c_begin, n), one per thread, thread index `slot` of `nslots`; Pm = reflectors (below the
// diagonal of each column k; [TQ_H][TQ_W + 1]) and taus in LDS.
__device__ __forceinline__ void tq_apply(const double (*Pm)[TQ_W + 1], const double *taus, double *A, int ldo, double *Nn, int n,
                                         int j0, int w, int c_begin, int slot, int nslots)
{
    for (int c = c_begin + slot; c < n; c += nslots) {
        double *col = A + (size_t)c * ldo + j0, *nc = Nn + (size_t)6 * c;
        double a[TQ_H];
#pragma unroll
        for (int i = 0; i < TQ_W; ++i) a[i] = (i < w) ? col[i] : 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) a[TQ_W + r] = nc[r];
#pragma unroll
        for (int k = 0; k < TQ_W; ++k) {
            double d = a[k];
#pragma unroll
            for (int i = k + 1; i < TQ_H; ++i) d += Pm[i][k] * a[i];
            d *= taus[k];
            a[k] -= d;
#pragma unroll
            for (int i = k + 1; i < TQ_H; ++i) a[i] -= d * Pm[i][k];
            asm volatile("" ::: "memory");                      // keep the next reflector's LDS reads from being hoisted (registers)
        }
#pragma unroll
        for (int i = 0; i < TQ_W; ++i)
            if (i < w) col[i] = a[i];
#pragma unroll
        for (int r = 0; r < 6; ++r) nc[r] = a[TQ_W + r];
    }
}

// Lane `lane` < w of the factoring wave: R column back to the matrix (zeros below the diagonal inside the
// block), reflector and tau to Pm / taus.
__device__ __forceinline__ void tq_store_panel_column(const double (&pc)[TQ_H], double tau_own, int lane, int w, double *A, int ldo,
                                                      int j0, double (*Pm)[TQ_W + 1], double *taus)
{
    if (lane < TQ_W) {
#pragma unroll
        for (int i = 0; i < TQ_H; ++i) Pm[i][lane] = (lane < w && i > lane) ? pc[i] : 0.0;
        taus[lane] = lane < w ? tau_own : 0.0;
    }
    if (lane < w) {
        double *col = A + (size_t)(j0 + lane) * ldo + j0;
#pragma unroll
        for (int i = 0; i < TQ_W; ++i)
            if (i < w) col[i] = (i <= lane) ? pc[i] : 0.0;
    }
}

// Whole factorisation by one workgroup per filter (n up to about a thousand: the later columns fit a few per thread).
__global__ __launch_bounds__(TQ_T) void tu_qr_kernel(double *S_out, int ldo, size_t strideO, int n, double *Nws)
{
    __shared__ double P[TQ_H][TQ_W + 1];
    __shared__ double taus[TQ_W];
    const int tid = threadIdx.x, inst = blockIdx.x;
    double *A = S_out + (size_t)inst * strideO;
    double *Nn = Nws + (size_t)inst * 6 * n;
    for (int j0 = 0; j0 < n; j0 += TQ_W) {
        const int w = min(TQ_W, n - j0);
        if (tid < 64) {
            double pc[TQ_H], tau_own;
            tq_load_panel_column(A, ldo, Nn, j0, w, tid, pc);
            tq_factor_wave(pc, tid, w, tau_own);
            tq_store_panel_column(pc, tau_own, tid, w, A, ldo, j0, P, taus);
        }
        __syncthreads();
        tq_apply(P, taus, A, ldo, Nn, n, j0, w, j0 + w, tid, TQ_T);
        __syncthreads();
    }
}

// Large n: one launch pair per panel.  Factor: one wave per filter, reflectors and taus to Vg
// ([batch][TQ_H * (TQ_W + 1) + TQ_W]).  Apply: grid over the later columns.
__global__ __launch_bounds__(64) void tu_qr_factor_kernel(double *S_out, int ldo, size_t strideO, int n, double *Nws, int j0, double *Vg)
{
    __shared__ double P[TQ_H][TQ_W + 1];
    __shared__ double taus[TQ_W];
    const int tid = threadIdx.x, inst = blockIdx.x;
    double *A = S_out + (size_t)inst * strideO;
    const int w = min(TQ_W, n - j0);
    double pc[TQ_H], tau_own;
    tq_load_panel_column(A, ldo, Nws + (size_t)inst * 6 * n, j0, w, tid, pc);
    tq_factor_wave(pc, tid, w, tau_own);
    tq_store_panel_column(pc, tau_own, tid, w, A, ldo, j0, P, taus);
    __syncthreads();
    double *out = Vg + (size_t)inst * (TQ_H * (TQ_W + 1) + TQ_W);
    for (int i = tid; i < TQ_H * (TQ_W + 1); i += 64) out[i] = (&P[0][0])[i];
    if (tid < TQ_W) out[TQ_H * (TQ_W + 1) + tid] = taus[tid];
}

__global__ __launch_bounds__(256) void tu_qr_apply_kernel(double *S_out, int ldo, size_t strideO, int n, double *Nws, int j0, const double *Vg)
{
    __shared__ double P[TQ_H][TQ_W + 1];
    __shared__ double taus[TQ_W];
    const int tid = threadIdx.x, inst = blockIdx.y;
    const double *in = Vg + (size_t)inst * (TQ_H * (TQ_W + 1) + TQ_W);
    for (int i = tid; i < TQ_H * (TQ_W + 1); i += 256) (&P[0][0])[i] = in[i];
    if (tid < TQ_W) taus[tid] = in[TQ_H * (TQ_W + 1) + tid];
    __syncthreads();
    const int w = min(TQ_W, n - j0);
    tq_apply(P, taus, S_out + (size_t)inst * strideO, ldo, Nws + (size_t)inst * 6 * n, n, j0, w, j0 + w,
             blockIdx.x * 256 + tid, gridDim.x * 256);
}

// Number of non-zero entries strictly below the diagonal of S (0: the structured path applies).
__global__ __launch_bounds__(256) void tu_count_lower_kernel(const double *S, int ldS, int n, int *count)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < n && r > c && S[(size_t)c * ldS + r] != 0.0) atomicAdd(count, 1);
}

static int tu_lda(int n) { return (n + 6 + 7) & ~7; }


// Instances per pass: the (n + 6) x n work matrices of one pass stay under 1 GiB.
int time_update_chunk(int n, int batch)
{
    const size_t per = (size_t)tu_lda(n) * n * 8;
    size_t c = ((size_t)1 << 30) / per;
    if (c < 1) c = 1;
    if (c > 32768) c = 32768;                       // grid and batch limits of the kernels below
    return (size_t)batch < c ? batch : (int)c;
}

size_t time_update_ws_bytes(int n, int chunk)
{
    return Arena::pad((size_t)chunk * 288 * 8) + Arena::pad((size_t)chunk * tu_lda(n) * n * 8) + qr_ws_bytes(n + 6, n, chunk);
}
size_t time_update_upper_ws_bytes(int n, int batch)
{
    return Arena::pad((size_t)batch * 288 * 8) + Arena::pad((size_t)batch * 6 * n * 8) +
           Arena::pad((size_t)(batch < 32768 ? batch : 32768) * (22 * 17 + 16) * 8);
}

// All pointers device.  mu_out may alias mu and S_out may alias S.  upper: S is upper triangular (its
// strict lower part is not read) and the structured kernels run; otherwise the dense QR.
vslam_status time_update_dev(vslam_ctx *ctx, const double *mu, size_t stride_mu, const double *S, int ldS, size_t strideS, int n,
                             int batch, const TuNoise &nz, double *mu_out, size_t stride_muo, double *S_out, int ldo, size_t strideO,
                             bool upper, void *ws)
{
    hipStream_t s = ctx->stream;
    if (upper) {
        Arena ar(ws);
        double *jac = ar.take<double>((size_t)batch * 288);
        double *Nws = ar.take<double>((size_t)batch * 6 * n);
        double *Vg = ar.take<double>((size_t)(batch < 32768 ? batch : 32768) * (TQ_H * (TQ_W + 1) + TQ_W));
        for (int b0 = 0; b0 < batch; b0 += 32768) {                 // grid limits
            const int nb = (batch - b0) < 32768 ? (batch - b0) : 32768;
            const double *Sb = S + (size_t)b0 * strideS;
            double *So = S_out + (size_t)b0 * strideO;
            hipLaunchKernelGGL(tu_rk4_kernel, dim3(nb), dim3(64), 0, s, mu + (size_t)b0 * stride_mu, stride_mu, n, nz.dt,
                               mu_out + (size_t)b0 * stride_muo, stride_muo, jac + (size_t)b0 * 288);
            hipLaunchKernelGGL(tu_build_upper_kernel, dim3((n + 255) / 256, nb), dim3(256), 0, s, Sb, ldS, strideS, n,
                               jac + (size_t)b0 * 288, nz, So, ldo, strideO, Nws + (size_t)b0 * 6 * n);
            if (So != Sb)
                hipLaunchKernelGGL(tu_copy_upper_kernel, dim3((n + 255) / 256, n, nb), dim3(256), 0, s, Sb, ldS, strideS, n, So, ldo, strideO);
            double *Nb = Nws + (size_t)b0 * 6 * n;
            if (n <= 1024) hipLaunchKernelGGL(tu_qr_kernel, dim3(nb), dim3(TQ_T), 0, s, So, ldo, strideO, n, Nb);
            else
                for (int j0 = 0; j0 < n; j0 += TQ_W) {
                    hipLaunchKernelGGL(tu_qr_factor_kernel, dim3(nb), dim3(64), 0, s, So, ldo, strideO, n, Nb, j0, Vg);
                    const int later = n - j0 - TQ_W;
                    if (later > 0)
                        hipLaunchKernelGGL(tu_qr_apply_kernel, dim3((later + 255) / 256, nb), dim3(256), 0, s, So, ldo, strideO, n, Nb, j0, Vg);
                }
            VSLAM_HIP_TRY(ctx, hipGetLastError());
        }
        return VSLAM_OK;
    }
    const int chunk = time_update_chunk(n, batch), lda = tu_lda(n);
    const size_t strideA = (size_t)lda * n;
    Arena ar(ws);
    double *jac = ar.take<double>((size_t)chunk * 288);
    double *A = ar.take<double>((size_t)chunk * strideA);
    void *qws = ar.take<char>(qr_ws_bytes(n + 6, n, chunk));
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = (batch - b0) < chunk ? (batch - b0) : chunk;
        hipLaunchKernelGGL(tu_rk4_kernel, dim3(nb), dim3(64), 0, s, mu + (size_t)b0 * stride_mu, stride_mu, n, nz.dt,
                           mu_out + (size_t)b0 * stride_muo, stride_muo, jac);
        hipLaunchKernelGGL(tu_build_kernel, dim3((n + 6 + 255) / 256, n, nb), dim3(256), 0, s, S + (size_t)b0 * strideS, ldS, strideS, n,
                           jac, nz, A, lda, strideA);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(qr_upper_dev(ctx, A, n + 6, n, lda, strideA, nb, qws));
        hipLaunchKernelGGL(tu_copy_out_kernel, dim3((n + 255) / 256, n, nb), dim3(256), 0, s, A, lda, strideA, n,
                           S_out + (size_t)b0 * strideO, ldo, strideO);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
    }
    return VSLAM_OK;
}

static vslam_status make_noise(vslam_ctx *ctx, const double *pos_tune, const double *ori_tune, double dt, TuNoise *nz)
{
    if (!pos_tune || !ori_tune) return fail(ctx, VSLAM_ERR_ARG, "time_update: null tuning matrix");
    if (!(dt >= 0.0)) return fail(ctx, VSLAM_ERR_ARG, "time_update: timestep %g", dt);
    const double sq = sqrt(dt);
    for (int i = 0; i < 36; ++i) nz->sq6[i] = 0.0;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            nz->sq6[r + 6 * c] = pos_tune[r + 3 * c] * sq;
            nz->sq6[3 + r + 6 * (3 + c)] = ori_tune[r + 3 * c] * sq;
        }
    nz->dt = dt;
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_time_update(vslam_ctx *ctx, const double *mu, const double *S, int n, int ldS, const double *pos_tune,
                                          const double *ori_tune, double dt, double *mu_out, double *S_out, int ldSout)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "time_update: null pointer");
    if (n < 12 || ldS < n || ldSout < n) return fail(ctx, VSLAM_ERR_SHAPE, "time_update: n=%d ldS=%d ldSout=%d", n, ldS, ldSout);
    TuNoise nz;
    VSLAM_TRY(make_noise(ctx, pos_tune, ori_tune, dt, &nz));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    const size_t tws_bytes = time_update_ws_bytes(n, 1) > time_update_upper_ws_bytes(n, 1) ? time_update_ws_bytes(n, 1) : time_update_upper_ws_bytes(n, 1);
    VSLAM_TRY(ws_reserve(ctx, 2 * Arena::pad(n * 8) + 2 * Arena::pad(nn * 8) + Arena::pad(64) + tws_bytes));
    Arena ar(ctx->ws);
    double *dmu = ar.take<double>(n), *dmo = ar.take<double>(n), *dS = ar.take<double>(nn), *dSo = ar.take<double>(nn);
    int *dcount = ar.take<int>(1);
    void *tws = ar.take<char>(tws_bytes);
    VSLAM_TRY(h2d(ctx, dmu, mu, n * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    // an upper-triangular factor (what the filter produces) takes the structured kernels
    int lower = 0;
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(dcount, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(tu_count_lower_kernel, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, dS, n, n, dcount);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(d2h(ctx, &lower, dcount, sizeof(int)));
    VSLAM_TRY(stream_sync(ctx));
    VSLAM_TRY(time_update_dev(ctx, dmu, n, dS, n, nn, n, 1, nz, dmo, n, dSo, n, nn, lower == 0, tws));
    VSLAM_TRY(d2h(ctx, mu_out, dmo, n * 8));
    VSLAM_TRY(d2h_2d(ctx, S_out, (size_t)ldSout * 8, dSo, (size_t)n * 8, (size_t)n * 8, n));
    return stream_finish(ctx);
}

extern "C" vslam_status vslam_time_update_batch_dev(vslam_ctx *ctx, const double *mu, const double *S, int n, int batch,
                                                    const double *pos_tune, const double *ori_tune, double dt, double *mu_out,
                                                    double *S_out, int s_is_upper)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "time_update: null pointer");
    if (n < 12 || batch < 1) return fail(ctx, VSLAM_ERR_SHAPE, "time_update: n=%d batch=%d", n, batch);
    TuNoise nz;
    VSLAM_TRY(make_noise(ctx, pos_tune, ori_tune, dt, &nz));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    if (s_is_upper) {
        VSLAM_TRY(ws_reserve(ctx, time_update_upper_ws_bytes(n, batch)));
        return time_update_dev(ctx, mu, n, S, n, nn, n, batch, nz, mu_out, n, S_out, n, nn, true, ctx->ws);
    }
    const int chunk = time_update_chunk(n, batch);
    VSLAM_TRY(ws_reserve(ctx, time_update_ws_bytes(n, chunk)));
    return time_update_dev(ctx, mu, n, S, n, nn, n, batch, nz, mu_out, n, S_out, n, nn, false, ctx->ws);
}
