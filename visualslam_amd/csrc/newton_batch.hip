// Iterated-EKF measurement update (reference src/gaussian.hpp:469-532 driving src/fmin.hpp:19-107 and
// src/fmin.cpp:40-131) for a bank of independent filters in lock step.
//
// One filter's update is a chain of a few hundred small dependent kernels (newton.hip): a single
// instance leaves the chip idle and several host threads with their own contexts do not scale.  Here
// every kernel of the chain takes the whole bank -- instance = a grid dimension -- and the host keeps
// the per-instance scalars of the trust-region iteration (radius, multiplier, cost, state of the
// loop), sends a mask of the instances that take part in the next kernels and reads back a few
// doubles per instance per step.  The arithmetic per instance is that of the device route of
// newton.hip (Cholesky factors instead of the eigendecomposition; identical sequence of multipliers);
// an instance whose Hessian is not positive definite is reported (status 1) for the caller to redo
// with vslam_measurement_update_iekf.
#include "landmark_math.hpp"
#include "linalg.hpp"
#include "trs_search.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace vslam {

namespace {

constexpr int BK_T = 256;

// Triangular solve with the upper-triangular S of every active instance, one workgroup each:
// trans = 1: S^T z = rhs (blocks first to last), trans = 0: S z = rhs (last to first); out = scale * z.
// The right-hand side lives in LDS; per 64-wide diagonal block the block is staged in LDS, wave 0
// substitutes (lane k owns unknown k, finished unknowns are broadcast with v_readlane) and the whole
// workgroup removes the block's contribution from the unknowns still open.
__global__ __launch_bounds__(BK_T) void bk_trsv_kernel(const double *S, int ldS, size_t strideS, int n, int trans, const double *rhs,
                                                       double *out, double scale, const int *active)
{
    extern __shared__ double bk_lds[];
    double *z = bk_lds;                         // [n] (padded to even)
    double *Ud = z + ((n + 1) & ~1);            // [64][65]
    double *xs = Ud + 64 * 65;                  // [64]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (active && !active[b]) return;
    const double *Sb = S + (size_t)b * strideS;
    for (int i = tid; i < n; i += BK_T) z[i] = rhs[(size_t)b * n + i];
    const int nblk = (n + 63) / 64;
    for (int step = 0; step < nblk; ++step) {
        const int blk = trans ? step : nblk - 1 - step, i0 = blk * 64, ib = min(64, n - i0);
        __syncthreads();
        for (int idx = tid; idx < 64 * 64; idx += BK_T) {
            const int r = idx & 63, c = idx >> 6;
            double v = (r == c) ? 1.0 : 0.0;
            if (r < ib && c < ib) v = (r <= c) ? Sb[(size_t)(i0 + c) * ldS + i0 + r] : 0.0;
            Ud[r * 65 + c] = v;
        }
        __syncthreads();
        if (tid < 64) {
            const int k = tid;
            double acc = (k < ib) ? z[i0 + k] : 0.0, xk = 0.0;
            const double rd = 1.0 / Ud[k * 65 + k];
            double u[64];                               // this lane's coefficients: column k (trans) or row k of the block
#pragma unroll
            for (int l = 0; l < 64; ++l) u[l] = trans ? Ud[l * 65 + k] : Ud[k * 65 + l];
            if (trans) {
#pragma unroll
                for (int l = 0; l < 64; ++l) {
                    const double xl = readlane_f64(acc, l) * readlane_f64(rd, l);
                    if (k == l) xk = xl;
                    acc -= ((k > l) ? u[l] : 0.0) * xl;
                }
            } else {
#pragma unroll
                for (int l = 63; l >= 0; --l) {
                    const double xl = readlane_f64(acc, l) * readlane_f64(rd, l);
                    if (k == l) xk = xl;
                    acc -= ((k < l) ? u[l] : 0.0) * xl;
                }
            }
            xs[k] = xk;
            if (k < ib) z[i0 + k] = xk;
        }
        __syncthreads();
        if (trans) {
            // later columns j: z[j] -= S[i0.., j] . x.  Eight lanes share a column, each holding the rows
            // 16 q + 2 sub + {0, 1}, so that one load instruction of the wave covers whole cache lines of
            // eight columns (a thread per column would touch 64 lines per instruction).  Only full blocks
            // have later columns, so all 64 rows exist.
            const int sub = tid & 7, grp = tid >> 3;
            double xr[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xr[2 * q] = xs[16 * q + 2 * sub]; xr[2 * q + 1] = xs[16 * q + 2 * sub + 1]; }
            for (int j0 = i0 + ib; j0 < n; j0 += BK_T / 8) {
                const int j = j0 + grp;
                double sum = 0.0;
                if (j < n) {
                    const double *col = Sb + (size_t)j * ldS + i0 + 2 * sub;
                    double v[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[2 * q] = col[16 * q]; v[2 * q + 1] = col[16 * q + 1]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum += v[q] * xr[q];
                }
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                sum += __shfl_xor(sum, 4, 64);
                if (sub == 0 && j < n) z[j] -= sum;
            }
        } else {
            for (int i = tid; i < i0; i += BK_T) {
                double s0 = 0.0, s1 = 0.0;
                int l = 0;
                for (; l + 2 <= ib; l += 2) {
                    s0 += Sb[(size_t)(i0 + l) * ldS + i] * xs[l];
                    s1 += Sb[(size_t)(i0 + l + 1) * ldS + i] * xs[l + 1];
                }
                if (l < ib) s0 += Sb[(size_t)(i0 + l) * ldS + i] * xs[l];
                z[i] -= s0 + s1;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += BK_T) out[(size_t)b * n + i] = scale * z[i];
}

// out[b][q] = a_q[b] . b_q[b], q < 6 (null pairs give 0); one workgroup per active instance.  out[b][6] = info[b].
constexpr int BK_DOTS = 8, BK_NDOT = 6;          // doubles per slot in the result; dot products among them
struct BkDotArgs { const double *a[BK_NDOT]; const double *b[BK_NDOT]; };
__global__ __launch_bounds__(BK_T) void bk_dots_kernel(BkDotArgs d, int n, double *out, const int *active, const int *info)
{
    __shared__ double red[BK_NDOT][BK_T / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (info && tid == 0) out[(size_t)b * BK_DOTS + 6] = (double)info[b];      // every slot's factorisation flag travels with the dot products
    if (active && !active[b]) return;
    double acc[BK_NDOT] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < n; i += BK_T)
#pragma unroll
        for (int q = 0; q < BK_NDOT; ++q)
            if (d.a[q]) acc[q] += d.a[q][(size_t)b * n + i] * d.b[q][(size_t)b * n + i];
#pragma unroll
    for (int q = 0; q < BK_NDOT; ++q) {
        if (!d.a[q]) continue;
        const double v = wave_sum(acc[q]);
        if ((tid & 63) == 0) red[q][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < BK_NDOT) out[(size_t)b * BK_DOTS + tid] = d.a[tid] ? ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3] : 0.0;
}

// out[b] = a * x[b] + c * y[b] over vectors of length n (y may be null), active instances only
__global__ void bk_axpby_kernel(double a, const double *x, double c, const double *y, double *out, int n, const int *active)
{
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if ((active && !active[b]) || i >= n) return;
    const size_t e = (size_t)b * n + i;
    out[e] = a * x[e] + (y ? c * y[e] : 0.0);
}

// y[b] = H[b] x[b] for the symmetric n x n column-major matrices of the selected instances (ld n, matrices n * n apart,
// vectors n apart), reading only the tiles of 64 x 64 on and above the diagonal: workgroup (rb, row of the grid) takes rows
// r0 = 64 rb ... r0 + 63.  The columns from r0 on are read as rows of the matrix (its four waves every fourth column, a
// wave reads 64 consecutive rows of a column); the columns before r0 are taken from the transposed side, H[0 .. r0-1][r]
// dot x[0 .. r0-1], a wave per column.  Fixed summation order.
__global__ __launch_bounds__(256) void bk_matvec_kernel(const double *H, int n, const double *x, double *y, const int *active, int listed)
{
    __shared__ double red[4][64], left[64];
    const int b = instance_of(active, listed, blockIdx.y);
    if (b < 0) return;
    const double *Hb = H + (size_t)b * n * n, *xb = x + (size_t)b * n;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, r0 = blockIdx.x * 64;
    const int r = r0 + lane, rr = min(r, n - 1);
    double a0 = 0.0, a1 = 0.0;
    int c = r0 + q;
#pragma unroll 4
    for (; c + 4 < n; c += 8) {
        const double v0 = Hb[(size_t)c * n + rr], v1 = Hb[(size_t)(c + 4) * n + rr];
        a0 = fma(v0, xb[c], a0); a1 = fma(v1, xb[c + 4], a1);
    }
    if (c < n) a0 = fma(Hb[(size_t)c * n + rr], xb[c], a0);
    red[q][lane] = a0 + a1;
    for (int j = q; j < 64; j += 16) {                            // four of the wave's columns together
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = lane; i < r0; i += 64) {
            const double xi = xb[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = fma(Hb[(size_t)min(r0 + j + 4 * e, n - 1) * n + i], xi, t[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double v = wave_sum(t[e]);
            if (lane == 0) left[j + 4 * e] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && r < n)
        y[(size_t)b * n + r] = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) + left[lane];
}

// dst[b] = src[b] (count doubles per instance), active instances only
__global__ void bk_copy_kernel(double *dst, const double *src, size_t count, const int *active, int listed = 0)
{
    const int b = instance_of(active, listed, blockIdx.y);
    if (b < 0) return;
    const size_t base = (size_t)b * count, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[base + i] = src[base + i];
}

// dst[s] = src[src_of[s]] (count doubles per slot) for every slot s with src_of[s] >= 0
__global__ void bk_gather_kernel(double *dst, const double *src, size_t count, const int *src_of)
{
    const int s = blockIdx.y, from = src_of[s];
    if (from < 0) return;
    const size_t to = (size_t)s * count, base = (size_t)from * count, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[to + i] = src[base + i];
}

// the same for the tiles of 64 x 64 on and above the diagonal of n x n column-major matrices (ld n, n * n apart); columns
// are dealt as in bk_neg_sum_upper_kernel
__global__ __launch_bounds__(256) void bk_gather_upper_kernel(double *dst, const double *src, int n, const int *src_of)
{
    const int s = blockIdx.y, from = src_of[s];
    if (from < 0) return;
    const size_t to = (size_t)s * n * n, base = (size_t)from * n * n;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, j0 = blockIdx.x * 16;
    const int rows = min(n, (j0 / 64 + 1) * 64);
#pragma unroll 2
    for (int i = lane; i < rows; i += 64) {
        double v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = src[base + (size_t)min(j0 + q + 4 * e, n - 1) * n + i];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j0 + q + 4 * e < n) dst[to + (size_t)(j0 + q + 4 * e) * n + i] = v[e];
    }
}

// out[b] = a[b] - c[b] / out[b] = -(a[b] + c[b]) over `count` doubles per instance, active instances only
__global__ void bk_sub_kernel(const double *a, const double *c, double *out, size_t count, const int *active)
{
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t base = (size_t)b * count, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[base + i] = a[base + i] - c[base + i];
}
__global__ void bk_neg_sum_kernel(const double *a, const double *c, double *out, size_t count, const int *active)
{
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t base = (size_t)b * count, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[base + i] = -(a[base + i] + c[base + i]);
}

// out[b] = -(a[b] + c[b]) on the tiles of 64 x 64 on and above the diagonal of n x n column-major matrices (ld n, n * n
// apart), active instances only: column j down to the end of its diagonal tile.  A workgroup takes 16 columns, a wave
// every fourth of them.
__global__ __launch_bounds__(256) void bk_neg_sum_upper_kernel(const double *a, const double *c, double *out, int n, const int *active)
{
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const size_t base = (size_t)b * n * n;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, j0 = blockIdx.x * 16;
    const int rows = min(n, (j0 / 64 + 1) * 64);                 // the 16 columns lie in one tile column
#pragma unroll 2
    for (int i = lane; i < rows; i += 64) {
        double v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {                             // a wave's four columns together: eight reads in flight
            const size_t o = base + (size_t)min(j0 + q + 4 * e, n - 1) * n + i;
            v[e] = -(a[o] + c[o]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j0 + q + 4 * e < n) out[base + (size_t)(j0 + q + 4 * e) * n + i] = v[e];
    }
}

// logGaussian value (src/gaussian.hpp:41-45) of every instance from z = S^-T (x - mu); g = -g in place.
__global__ __launch_bounds__(BK_T) void bk_log_gaussian_value_kernel(const double *z, const double *S, size_t strideS, int n, double *value, double *g,
                                                                     const int *active)
{
    __shared__ double red[2][BK_T / 64];
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const double *Sb = S + (size_t)b * strideS;
    double sq = 0.0, ls = 0.0;
    for (int i = threadIdx.x; i < n; i += BK_T) {
        const double zi = z[(size_t)b * n + i];
        sq += zi * zi;
        ls += log(fabs(Sb[(size_t)i * n + i]));
        g[(size_t)b * n + i] = -g[(size_t)b * n + i];
    }
    sq = wave_sum(sq); ls = wave_sum(ls);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = ls; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        const double c = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        value[b] = -0.5 * a - (double)n / 2 * log(2 * 3.14159265358979323846264338327950288) - c;
    }
}

// B filters and Bt - B further slots in which trust-region subproblems of the same filters are solved ahead
// (iekf_bank_chunk): the slots beyond B own a factor, a current Hessian and the vectors, nothing else.
struct BankWs {
    double *Hp, *Hl, *W;                                                    // [B][n*n]
    double *U, *Hcur;                                                       // [Bt][n*n]
    double *x, *tmp, *gp, *gl, *g, *xcur, *gcur, *p, *t, *q, *xn, *Hpv;     // [Bt][n]
    double *vp, *vl, *V, *dots, *lam;                                       // [Bt] ([Bt][BK_DOTS] for dots)
    int *info, *active, *map;                                               // [Bt]
    void *tws;
};

size_t bank_ws_bytes(int n, int B, int Bt)
{
    const size_t nn = (size_t)n * n;
    return 3 * Arena::pad((size_t)B * nn * 8) + 2 * Arena::pad((size_t)Bt * nn * 8) + 12 * Arena::pad((size_t)Bt * n * 8) +
           9 * Arena::pad((size_t)Bt * 4 * 8) + Arena::pad((size_t)Bt * BK_DOTS * 8) + trtri_batch_ws_bytes(n, B);
}

void bank_carve(void *ws, int n, int B, int Bt, BankWs *w)
{
    Arena ar(ws);
    const size_t nn = (size_t)n * n;
    w->Hp = ar.take<double>(B * nn); w->Hl = ar.take<double>(B * nn);
    w->W = ar.take<double>(B * nn); w->U = ar.take<double>(Bt * nn); w->Hcur = ar.take<double>(Bt * nn);
    double **v[12] = {&w->x, &w->tmp, &w->gp, &w->gl, &w->g, &w->xcur, &w->gcur, &w->p, &w->t, &w->q, &w->xn, &w->Hpv};
    for (double **q : v) *q = ar.take<double>((size_t)Bt * n);
    w->vp = ar.take<double>(4 * Bt); w->vl = ar.take<double>(4 * Bt); w->V = ar.take<double>(4 * Bt);
    w->dots = ar.take<double>((size_t)BK_DOTS * Bt); ar.take<double>(4 * Bt); w->lam = ar.take<double>(4 * Bt);
    w->info = reinterpret_cast<int *>(ar.take<double>(4 * Bt));
    ar.take<double>(4 * Bt);
    w->active = reinterpret_cast<int *>(w->lam + Bt);      // mask and list sit behind the multipliers: one upload carries all three
    w->map = reinterpret_cast<int *>(ar.take<double>(4 * Bt));
    ar.take<double>(4 * Bt);
    w->tws = ar.take<char>(trtri_batch_ws_bytes(n, B));
}

int grid_for(size_t count) { return (int)std::min<size_t>((count + 1023) / 1024, 8192); }

}  // namespace

// Slots beyond the B filters of a chunk, for subproblems solved ahead (see iekf_bank_chunk).
static int bank_extra_slots(int B)
{
    if (const char *e = getenv("VSLAM_BANK_SOLVE_AHEAD")) if (atoi(e) == 0) return 0;     // A/B runs
    return std::min(B, 1024);
}

// One chunk of B filters, all pointers device except the per-instance result arrays.
//
// The filters run in lock step; the host keeps each filter's radius, multiplier, cost and loop state.  Towards
// the end of an update most filters are finished and the rest go through rejected steps: the cost no longer
// resolves the step, the radius is quartered (src/fmin.hpp:88-96) and the subproblem is solved again on an
// unchanged Hessian and gradient -- six to eight dependent factorisations whose kernels leave the chip idle.
// Those subproblems do not depend on the cost evaluations in between, so they are solved AHEAD: a filter whose
// step was rejected gets up to SOLVE_AHEAD extra slots beyond the B filters, each with a copy of its Hessian,
// gradient and iterate and the radius quartered once more, and all of them go through the same launches.  When the
// filter's next step is rejected as well and its new radius 0.25 |p| equals the radius solved for (to 1e-9
// relative: it does whenever the solution lay on the boundary, to the 1e-10 of the multiplier iteration), the next
// subproblem's solution is already there and only the cost evaluation remains.  An accepted step, or a radius that
// differs, discards what was solved ahead.
static vslam_status iekf_bank_chunk(vslam_ctx *ctx, const CamK &k, double sigma, int n, int B, const double *mu, const double *S,
                                    const double *y, const uint8_t *vis, double *mu_post, double *S_post, int32_t *fmin_code,
                                    int32_t *iters, int32_t *status)
{
    constexpr int SOLVE_AHEAD = 6, MAX_ITERATIONS = 50;
    const size_t nn = (size_t)n * n;
    const int L = (n - 12) / 3, nb = (n + 255) / 256;
    const int Bt = B + bank_extra_slots(B);
    hipStream_t s = ctx->stream;
    VSLAM_TRY(ws_reserve(ctx, bank_ws_bytes(n, B, Bt)));
    BankWs w;
    bank_carve(ctx->ws, n, B, Bt, &w);
    const size_t trsv_lds = (size_t)(((n + 1) & ~1) + 64 * 65 + 64) * 8;
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&bk_trsv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trsv_lds));
    const double eps = std::numeric_limits<double>::epsilon();

    // per slot (filters 0 .. B-1, then the slots solved ahead)
    std::vector<int> trs(Bt, 0), m(Bt, 0), hinfo(Bt, 0), map(Bt, -1);
    std::vector<double> Delta(Bt, 1.0), lam(Bt, 0.0), hd((size_t)BK_DOTS * Bt, 0.0);
    // per filter
    std::vector<int> act(B, 1), acc(B, 0), ret(B, 1), its(B, MAX_ITERATIONS), stat(B, 0), done(B, 0), itb(B, 0), rej(B, 0), cur(B, 0);
    std::vector<double> f(B, 0.0), fn(B, 0.0), warm(B, 0.0);      // warm: where a filter's next search starts after an indefinite Hessian
    // A filter whose step was rejected faces the same Hessian and gradient with a smaller radius: its factor U and step p
    // of the last search are still in its slot, |p| exceeds the new radius (0.25 |p| at most), so the new search needs no
    // factorisation at zero -- it continues from the moments at that factor (and seeds the subproblems solved ahead).
    std::vector<TrsSlot> ts_last(B);
    std::vector<int> have_factor(B, 0);
    const bool continue_rejected = !getenv("VSLAM_BANK_NO_CONTINUE");
    std::vector<std::vector<int>> ahead(B);                     // slots holding the filter's next subproblems, nearest first

    // One workgroup per matrix either way: the chain form of tri.hip where its blocks fit LDS (VSLAM_BANK_TRSV_OLD=1 keeps
    // the older kernel for A/B runs), bk_trsv_kernel otherwise.
    const bool chain_trsv = trsv_upper_bank_fits(n) && !getenv("VSLAM_BANK_TRSV_OLD");
    auto trsv = [&](const double *M, bool trans, const double *rhs, double *out, double scale, const int *mask, int count) -> vslam_status {
        if (chain_trsv) return trsv_upper_bank_dev(ctx, M, n, n, nn, trans, rhs, out, scale, mask, count);
        hipLaunchKernelGGL(bk_trsv_kernel, dim3(count), dim3(BK_T), trsv_lds, s, M, n, nn, n, trans ? 1 : 0, rhs, out, scale, mask);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        return VSLAM_OK;
    };
    // A mask over the filters, or over all slots; the device array always holds Bt mask words and, behind them, the
    // slots of the mask as a list (nlisted of them): the factorisations and their solves launch over the list, so a
    // step in which a handful of slots take part does not dispatch workgroups for all of them.
    std::vector<int> sel((size_t)2 * Bt, 0);
    int nlisted = 0;
    int *const listv = w.active + Bt;
    std::vector<double> up((size_t)2 * Bt, 0.0);                  // staging: Bt multipliers, then the 2 Bt words of mask and list
    auto send_mask = [&](const std::vector<int> &h, bool with_lam = false) -> vslam_status {
        std::fill(sel.begin(), sel.end(), 0);
        nlisted = 0;
        for (size_t i = 0; i < h.size(); ++i)
            if (h[i]) { sel[i] = 1; sel[(size_t)Bt + nlisted++] = (int)i; }
        if (!with_lam) return h2d(ctx, w.active, sel.data(), sel.size() * sizeof(int));
        std::memcpy(up.data(), lam.data(), (size_t)Bt * 8);
        std::memcpy(up.data() + Bt, sel.data(), sel.size() * sizeof(int));
        return h2d(ctx, w.lam, up.data(), up.size() * 8);
    };
    auto any = [&](const std::vector<int> &h) { return std::any_of(h.begin(), h.end(), [](int v) { return v != 0; }); };
    // results of bk_dots_kernel and the factorisation flags of all slots
    auto dots = [&](const double *a0, const double *b0, const double *a1, const double *b1, const double *a2, const double *b2,
                    const double *a3 = nullptr, const double *b3 = nullptr, const double *a4 = nullptr, const double *b4 = nullptr,
                    const double *a5 = nullptr, const double *b5 = nullptr) -> vslam_status {
        BkDotArgs d;
        d.a[0] = a0; d.b[0] = b0; d.a[1] = a1; d.b[1] = b1; d.a[2] = a2; d.b[2] = b2; d.a[3] = a3; d.b[3] = b3;
        d.a[4] = a4; d.b[4] = b4; d.a[5] = a5; d.b[5] = b5;
        hipLaunchKernelGGL(bk_dots_kernel, dim3(Bt), dim3(BK_T), 0, s, d, n, w.dots, w.active, (const int *)w.info);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(d2h(ctx, hd.data(), w.dots, (size_t)BK_DOTS * Bt * 8));
        VSLAM_TRY(stream_sync(ctx));
        for (int sl = 0; sl < Bt; ++sl) hinfo[sl] = (int)hd[(size_t)BK_DOTS * sl + 6];
        return VSLAM_OK;
    };
    // for the slots of the mask on the device: U^T U = Hcur + lam I, p = -(Hcur + lam I)^-1 gcur
    // (the multipliers travel with the mask: send_mask(mask, true) right before)
    auto factor_and_step = [&]() -> vslam_status {
        if (nlisted == 0) return VSLAM_OK;
        const bool by_list = chain_trsv && nlisted < Bt;            // launches over the list of slots (the older solve kernel takes masks only)
        const int *selp = by_list ? listv : w.active;
        const int rows = by_list ? nlisted : Bt, lf = by_list ? 1 : 0;
        hipLaunchKernelGGL(bk_copy_kernel, dim3(1, rows), dim3(256), 0, s, w.t, w.gcur, (size_t)n, selp, lf);
        // t = U^-T g comes with the factor; Hcur + lam I is read where it lies (no shifted copy into U first)
        PotrfFrom from{w.Hcur, nn, w.lam, 0.0};
        VSLAM_TRY(potrf_upper_batch_dev(ctx, w.U, n, n, nn, rows, w.info, selp, /*zero_lower=*/false, w.t, (size_t)n, by_list, &from));
        if (by_list) VSLAM_TRY(trsv_upper_bank_dev(ctx, w.U, n, n, nn, false, w.t, w.p, -1.0, selp, rows, true));
        else VSLAM_TRY(trsv(w.U, false, w.t, w.p, -1.0, w.active, Bt));
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        return VSLAM_OK;
    };
    bool prior_ready = false;
    // V, g, H of CostJointDensity (src/gaussian.hpp:435-467) at w.x for every filter of the mask; V to the host
    auto eval = [&](std::vector<double> &V) -> vslam_status {
        const int *a = w.active;
        hipLaunchKernelGGL(bk_sub_kernel, dim3(1, B), dim3(256), 0, s, w.x, mu, w.tmp, (size_t)n, a);
        VSLAM_TRY(trsv(S, true, w.tmp, w.tmp, 1.0, a, B));                // z = S^-T (x - mu)
        VSLAM_TRY(trsv(S, false, w.tmp, w.gp, 1.0, a, B));                // S^-1 z (negated below)
        hipLaunchKernelGGL(bk_log_gaussian_value_kernel, dim3(B), dim3(BK_T), 0, s, w.tmp, S, nn, n, w.vp, w.gp, a);
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        if (!prior_ready) {                                               // the prior Hessian does not depend on x
            VSLAM_TRY(trtri_upper_batch_dev(ctx, S, n, n, nn, w.W, n, nn, B, w.tws, /*zero_lower=*/false));   // W enters the product below as a triangular operand
            GemmDesc d;                                                   // Hp = -W W^T
            d.M = n; d.N = n; d.K = n; d.transA = false; d.transB = true; d.amode = A_UPPER; d.b_upper = true; d.alpha = -1.0;
            d.A = w.W; d.lda = n; d.strideA = nn; d.B = w.W; d.ldb = n; d.strideB = nn; d.C = w.Hp; d.ldc = n; d.strideC = nn; d.batch = B;
            d.c_upper = true;                                             // symmetric: the upper tiles are all accept() reads
            VSLAM_TRY(gemm_f64(ctx, d));
            prior_ready = true;
        }
        VSLAM_TRY(launch_point_loglik(ctx, k, B, n, w.x, n, y, 2 * (size_t)L, vis, L, sigma, w.vl, w.gl, n, w.Hl, n, nn, nullptr, a));
        hipLaunchKernelGGL(bk_neg_sum_kernel, dim3(1, B), dim3(64), 0, s, w.vp, w.vl, w.V, (size_t)1, a);
        hipLaunchKernelGGL(bk_neg_sum_kernel, dim3(1, B), dim3(256), 0, s, w.gp, w.gl, w.g, (size_t)n, a);
        // the Hessian -(Hp + Hl) of a point is only ever wanted when the point is accepted: accept() forms it there
        VSLAM_HIP_TRY(ctx, hipGetLastError());
        VSLAM_TRY(d2h(ctx, V.data(), w.V, (size_t)B * 8));
        return stream_sync(ctx);
    };
    // current iterate := the point just evaluated, for the filters of the mask on the device
    auto accept = [&]() {
        hipLaunchKernelGGL(bk_copy_kernel, dim3(1, B), dim3(256), 0, s, w.xcur, w.x, (size_t)n, w.active);
        hipLaunchKernelGGL(bk_copy_kernel, dim3(1, B), dim3(256), 0, s, w.gcur, w.g, (size_t)n, w.active);
        // Hcur = -(Hp + Hl) of the point just evaluated; every reader of Hcur (the factorisation, the product with the
        // step, the posterior's reversal) takes the tiles on and above the diagonal only
        hipLaunchKernelGGL(bk_neg_sum_upper_kernel, dim3((n + 15) / 16, B), dim3(256), 0, s, w.Hp, w.Hl, w.Hcur, n, w.active);
    };
    // dst[slot] = src[map[slot]] for the slots whose map entry is not negative
    auto gather = [&](double *dst, const double *src, size_t count, int slots) {
        hipLaunchKernelGGL(bk_gather_kernel, dim3(grid_for(count), slots), dim3(256), 0, s, dst, src, count, w.map);
    };
    auto send_map = [&]() -> vslam_status { return h2d(ctx, w.map, map.data(), (size_t)Bt * sizeof(int)); };
    // a slot whose subproblem cannot be finished here (the hard case of an indefinite Hessian, a Hessian that is not
    // finite): a filter leaves the bank (status 1); a slot solved ahead just drops out with everything behind it
    auto drop_failed = [&](std::vector<int> &mask, int only = -1) {
        for (int sl = 0; sl < Bt; ++sl) {
            if (only >= 0 ? sl != only : hinfo[sl] == 0) continue;
            if (!trs[sl]) continue;
            trs[sl] = 0; mask[sl] = 0; m[sl] = 0;
            if (sl < B) { stat[sl] = 1; done[sl] = 1; act[sl] = 0; ahead[sl].clear(); }
            else
                for (int fb = 0; fb < B; ++fb) {
                    auto &ch = ahead[fb];
                    const auto itp = std::find(ch.begin(), ch.end(), sl);
                    if (itp != ch.end()) { for (auto q = itp; q != ch.end(); ++q) { trs[*q] = 0; mask[*q] = 0; m[*q] = 0; } ch.erase(itp, ch.end()); break; }
                }
        }
    };

    VSLAM_HIP_TRY(ctx, hipMemsetAsync(w.info, 0, (size_t)Bt * sizeof(int), s));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(w.x, mu, (size_t)B * n * 8, hipMemcpyDeviceToDevice, s));
    VSLAM_TRY(send_mask(act));
    VSLAM_TRY(eval(f));                                                   // src/fmin.hpp:39
    accept();
    const int ahead_pct = getenv("VSLAM_BANK_AHEAD_PCT") ? atoi(getenv("VSLAM_BANK_AHEAD_PCT")) : 50;   // tuning aid
    const bool trace = getenv("VSLAM_BANK_TRACE") != nullptr;      // diagnostic: per-round wall time, filters taking part, multiplier steps
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now_ms();
    for (int round = 0; round < MAX_ITERATIONS && any(act); ++round) {
        int trace_steps = 0, trace_levels = 0, trace_ahead = 0;
        const int trace_active = (int)std::count(act.begin(), act.end(), 1);
        // ---- subproblems to solve ahead ----
        std::fill(trs.begin(), trs.end(), 0);
        std::fill(map.begin(), map.end(), -1);
        for (int b = 0; b < B; ++b) { trs[b] = act[b]; ahead[b].clear(); }
        if (Bt > B) {
            int nrej = 0, nact = 0;
            for (int b = 0; b < B; ++b) { nrej += act[b] && rej[b]; nact += act[b]; }
            // only while the bank is latency bound: filters and subproblems together stay within a fraction of the
            // bank (beyond it every launch takes longer in proportion and solving ahead costs what it saves)
            const int room = std::min(Bt - B, B * ahead_pct / 100 - nact);
            const int levels = (nrej && room > 0) ? std::min(SOLVE_AHEAD, room / nrej) : 0;
            if (levels >= 1) {
                int v = B;
                for (int b = 0; b < B; ++b) {
                    if (!(act[b] && rej[b])) continue;
                    double d = Delta[b];
                    for (int j = 0; j < levels && itb[b] + 1 + j < MAX_ITERATIONS; ++j, ++v) {
                        d *= 0.25;
                        Delta[v] = d; trs[v] = 1; map[v] = b; ahead[b].push_back(v);
                        ++trace_ahead;
                    }
                }
                VSLAM_TRY(send_map());
                VSLAM_HIP_TRY(ctx, hipMemsetAsync(w.info + B, 0, (size_t)(Bt - B) * sizeof(int), s));
                hipLaunchKernelGGL(bk_gather_upper_kernel, dim3((n + 15) / 16, Bt), dim3(256), 0, s, w.Hcur, w.Hcur, n, w.map);   // Hcur holds its upper tiles only
                gather(w.gcur, w.gcur, (size_t)n, Bt);
                gather(w.xcur, w.xcur, (size_t)n, Bt);
                VSLAM_HIP_TRY(ctx, hipGetLastError());
            }
        }
        // ---- trust-region subproblems, src/fmin.cpp:40-131: every slot runs the multiplier search of trs_search.hpp ----
        std::fill(lam.begin(), lam.end(), 0.0);
        std::vector<TrsSlot> ts(Bt);
        std::vector<int> Q(Bt, 0), R(Bt, 0), last(Bt, 0), failed(Bt, 0), scratch(Bt, 0);
        std::vector<double> piv(Bt, 0.0);
        bool info_dirty = false;
        std::vector<int> first(Bt, 0), seed(Bt, 0), waiting(Bt, 0);
        for (int sl = 0; sl < Bt; ++sl) {
            ts[sl].Delta = Delta[sl];
            if (!trs[sl]) continue;
            const int fb = sl < B ? sl : map[sl];                                       // the filter a slot solved ahead belongs to
            if (continue_rejected && fb >= 0 && rej[fb] && have_factor[fb]) {
                if (sl < B) {
                    ts[sl] = ts_last[fb];
                    ts[sl].Delta = Delta[sl]; ts[sl].left = ts[sl].lam; ts[sl].hi = std::numeric_limits<double>::infinity();
                    ts[sl].steps = 0; ts[sl].fails = 0; ts[sl].bump = 0.0;
                    lam[sl] = ts[sl].lam; Q[sl] = 1; seed[sl] = 1;
                } else waiting[sl] = 1;                                                  // seeded when its filter's moments arrive
                continue;
            }
            first[sl] = 1;
            if (fb >= 0 && warm[fb] > 0.0) { trs_warm_start(ts[sl], warm[fb]); lam[sl] = ts[sl].lam; }
        }
        if (any(first)) {
            VSLAM_TRY(send_mask(first, true));
            VSLAM_TRY(factor_and_step());
            VSLAM_TRY(dots(w.p, w.p, nullptr, nullptr, nullptr, nullptr));
            for (int sl = 0; sl < Bt; ++sl) {
                if (!first[sl]) continue;
                if (hinfo[sl]) failed[sl] = 1;                                           // H itself is not positive definite
                else if (!trs_interior(ts[sl], hd[(size_t)BK_DOTS * sl])) Q[sl] = 1;      // the Newton step leaves the region
            }
        }
        for (int kk = 0; kk < 80; ++kk) {
            if (any(failed)) {
                // the first non-positive pivot of each failed factorisation sits on the factor's diagonal
                for (int sl = 0; sl < Bt; ++sl)
                    if (failed[sl]) VSLAM_TRY(d2h(ctx, &piv[sl], w.U + (size_t)sl * nn + (size_t)(hinfo[sl] - 1) * n + (hinfo[sl] - 1), 8));
                VSLAM_TRY(stream_sync(ctx));
                for (int sl = 0; sl < Bt; ++sl) {
                    if (!failed[sl]) continue;
                    failed[sl] = 0;
                    if (trs_after_failure(ts[sl], piv[sl]) == TRS_GIVE_UP) { drop_failed(scratch, sl); continue; }
                    lam[sl] = ts[sl].lam; R[sl] = 1; last[sl] = 0;
                }
                info_dirty = true;
            }
            if (any(R)) {
                ++trace_steps;
                if (info_dirty) { VSLAM_HIP_TRY(ctx, hipMemsetAsync(w.info, 0, (size_t)Bt * sizeof(int), s)); info_dirty = false; }
                VSLAM_TRY(send_mask(R, true));
                VSLAM_TRY(factor_and_step());
                for (int sl = 0; sl < Bt; ++sl)
                    if (R[sl]) { R[sl] = 0; if (!last[sl]) Q[sl] = 1; }                     // a last factorisation's verdict arrives with the dot products further down
            }
            if (!any(Q)) break;
            // moments of the search at every fresh factor: |p|^2, |U^-T p|^2, |U^-1 U^-T p|^2, |U^-T U^-1 U^-T p|^2
            VSLAM_TRY(send_mask(Q));
            if (chain_trsv && nlisted < Bt) {
                VSLAM_TRY(trsv_upper_bank_dev(ctx, w.U, n, n, nn, true, w.p, w.q, 1.0, listv, nlisted, true));
                VSLAM_TRY(trsv_upper_bank_dev(ctx, w.U, n, n, nn, false, w.q, w.t, 1.0, listv, nlisted, true));
                VSLAM_TRY(trsv_upper_bank_dev(ctx, w.U, n, n, nn, true, w.t, w.Hpv, 1.0, listv, nlisted, true));
            } else {
                VSLAM_TRY(trsv(w.U, true, w.p, w.q, 1.0, w.active, Bt));
                VSLAM_TRY(trsv(w.U, false, w.q, w.t, 1.0, w.active, Bt));
                VSLAM_TRY(trsv(w.U, true, w.t, w.Hpv, 1.0, w.active, Bt));
            }
            VSLAM_TRY(dots(w.p, w.p, w.q, w.q, w.t, w.t, w.Hpv, w.Hpv, w.gcur, w.gcur, w.gcur, w.p));
            for (int sl = 0; sl < Bt; ++sl) {
                if (!Q[sl]) continue;
                Q[sl] = 0;
                if (!trs[sl]) continue;
                if (hinfo[sl]) { failed[sl] = 1; continue; }
                const double *h4 = &hd[(size_t)BK_DOTS * sl];
                if (seed[sl]) {
                    // the same moments start the filter's subproblems solved ahead, each at the model's root for its own radius
                    seed[sl] = 0;
                    for (int v : ahead[sl]) {
                        if (!waiting[v] || !trs[v]) continue;
                        waiting[v] = 0;
                        ts[v] = ts[sl];
                        ts[v].Delta = Delta[v];
                        const TrsNext nv = trs_after_solve(ts[v], h4[0], h4[1], h4[2], h4[3], h4[4], -h4[5]);
                        if (nv == TRS_GIVE_UP || nv == TRS_DONE) { ts[v] = TrsSlot(); ts[v].Delta = Delta[v]; lam[v] = 0.0; R[v] = 1; last[v] = 0; continue; }   // back to a search from zero
                        lam[v] = ts[v].lam; R[v] = 1; last[v] = nv == TRS_REFACTOR_LAST;
                    }
                }
                const TrsNext next = trs_after_solve(ts[sl], h4[0], h4[1], h4[2], h4[3], h4[4], -h4[5]);
                if (next == TRS_GIVE_UP) { drop_failed(scratch, sl); continue; }
                if (next == TRS_DONE) continue;
                lam[sl] = ts[sl].lam; R[sl] = 1; last[sl] = next == TRS_REFACTOR_LAST;
            }
        }
        for (int sl = B; sl < Bt; ++sl)
            if (waiting[sl]) drop_failed(scratch, sl);              // never seeded (its filter left the bank on the way)
        for (int b = 0; b < B; ++b)
            if (trs[b]) { warm[b] = ts[b].indefinite ? ts[b].lam : 0.0; ts_last[b] = ts[b]; have_factor[b] = 1; }
        // ---- src/fmin.hpp:54-105: predicted decrease and the trial point of every slot ----
        VSLAM_TRY(send_mask(trs));
        if (nlisted > 0) {                             // Hpv = Hcur * p
            hipLaunchKernelGGL(bk_matvec_kernel, dim3((n + 63) / 64, nlisted), dim3(256), 0, s, w.Hcur, n, w.p, w.Hpv, listv, 1);
            VSLAM_HIP_TRY(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(bk_axpby_kernel, dim3(nb, Bt), dim3(256), 0, s, 1.0, w.xcur, 1.0, w.p, w.xn, n, w.active);
        VSLAM_TRY(dots(w.p, w.gcur, w.p, w.Hpv, w.p, w.p));
        drop_failed(scratch);                                 // a last factorisation that failed after all: the filter is handed back
        // ---- the filters' iterations: first on their own subproblem, then, rejected step after rejected step, on
        //      those solved ahead ----
        std::vector<int> pend(act);
        for (int b = 0; b < B; ++b) cur[b] = b;
        while (any(pend)) {
            ++trace_levels;
            for (int b = 0; b < B; ++b)
                if (pend[b] && std::fabs(-hd[(size_t)BK_DOTS * cur[b]]) < 2 * eps) {                       // src/fmin.hpp:61-67
                    ret[b] = 0; its[b] = itb[b]; done[b] = 1; act[b] = 0; pend[b] = 0; ahead[b].clear();
                }
            if (!any(pend)) break;
            std::fill(map.begin(), map.end(), -1);
            for (int b = 0; b < B; ++b)
                if (pend[b]) map[b] = cur[b];
            VSLAM_TRY(send_map());
            gather(w.x, w.xn, (size_t)n, B);
            VSLAM_TRY(send_mask(pend));
            VSLAM_TRY(eval(fn));
            std::vector<int> next(B, 0);
            for (int b = 0; b < B; ++b) {
                acc[b] = 0;
                if (!pend[b]) continue;
                const int sl = cur[b];
                const double pg = hd[(size_t)BK_DOTS * sl], pHp = hd[(size_t)BK_DOTS * sl + 1], pnb = std::sqrt(hd[(size_t)BK_DOTS * sl + 2]);
                const double dm = -pg - 0.5 * pHp, rho = (f[b] - fn[b]) / dm;
                double Dn = Delta[sl];
                if (rho < 0.1) Dn = 0.25 * pnb;
                else if (rho > 0.75 && pnb > 0.8 * Delta[sl]) Dn = 2.0 * Delta[sl];
                Delta[b] = Dn;
                ++itb[b];
                if (rho >= 0.001) { f[b] = fn[b]; acc[b] = 1; rej[b] = 0; ahead[b].clear(); }
                else {
                    rej[b] = 1;
                    if (!ahead[b].empty() && itb[b] < MAX_ITERATIONS && std::fabs(Delta[ahead[b].front()] - Dn) <= 1e-9 * Dn) {
                        cur[b] = ahead[b].front();
                        ahead[b].erase(ahead[b].begin());
                        next[b] = 1;
                    } else ahead[b].clear();
                }
            }
            if (any(acc)) {
                VSLAM_TRY(send_mask(acc));
                accept();
            }
            pend.swap(next);
        }
        for (int b = 0; b < B; ++b)
            if (act[b] && itb[b] >= MAX_ITERATIONS) { act[b] = 0; done[b] = 1; }                    // ret stays 1, its stays at the limit
        if (trace) {
            const double t = now_ms();
            std::fprintf(stderr, "bank round %d: %d filters + %d subproblems solved ahead, %d multiplier steps, %d evaluation passes, %.2f ms\n", round,
                         trace_active, trace_ahead, trace_steps, trace_levels, t - t_prev);
            t_prev = t;
        }
    }
    // ---- posterior square roots: S = M^-1 with M M^T = H at the final iterate (see newton.hip) ----
    std::vector<int> ok(B);
    for (int b = 0; b < B; ++b) ok[b] = stat[b] == 0;
    VSLAM_TRY(send_mask(ok));
    VSLAM_TRY(reverse_upper_dev(ctx, w.Hcur, n, w.U, false, nn, B, /*sym_upper=*/true));   // Hcur holds its upper tiles only
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(w.info, 0, (size_t)B * sizeof(int), s));
    VSLAM_TRY(potrf_upper_batch_dev(ctx, w.U, n, n, nn, B, w.info, w.active, /*zero_lower=*/false));     // the reversal reads the upper tiles only
    VSLAM_TRY(reverse_upper_dev(ctx, w.U, n, w.Hl, true, nn, B));
    VSLAM_TRY(trtri_upper_batch_dev(ctx, w.Hl, n, n, nn, S_post, n, nn, B, w.tws));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(mu_post, w.xcur, (size_t)B * n * 8, hipMemcpyDeviceToDevice, s));
    VSLAM_TRY(d2h(ctx, hinfo.data(), w.info, (size_t)B * sizeof(int)));
    VSLAM_TRY(stream_finish(ctx));
    for (int b = 0; b < B; ++b) {
        if (hinfo[b] != 0) stat[b] = 1;
        if (status) status[b] = stat[b];
        if (fmin_code) fmin_code[b] = ret[b];
        if (iters) iters[b] = its[b];
    }
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_measurement_update_iekf_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mu,
                                                                const double *S, int n, int batch, const double *y,
                                                                const uint8_t *visible, double sigma, double *mu_post,
                                                                double *S_post, int32_t *fmin_code, int32_t *iters, int32_t *status)
{
    if (!ctx) return VSLAM_ERR_ARG;
    // n = 12: filters whose maps hold no landmark yet (empty likelihood, src/model.cpp:392; the update returns the prior)
    if (n < 12 || (n - 12) % 3 != 0 || batch < 1) return fail(ctx, VSLAM_ERR_SHAPE, "measurement_update_iekf_batch: n=%d is not 12+3L or batch=%d", n, batch);
    if (!mu || !S || !mu_post || !S_post || (n > 12 && (!y || !visible))) return fail(ctx, VSLAM_ERR_ARG, "measurement_update_iekf_batch: null pointer");
    if (!(sigma > 0.0)) return fail(ctx, VSLAM_ERR_ARG, "measurement_update_iekf_batch: sigma must be positive");
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    const int L = (n - 12) / 3;
    if (!trsv_upper_bank_fits(n) && !getenv("VSLAM_BANK_ANY_SIZE")) {
        // Large states (the bank's one-workgroup triangular solve no longer holds a solution in LDS): one such filter fills the
        // chip by itself, and the kernels the lock-step bank has left for these sizes are the older, slower ones.  The filters go through the single-filter device route one after the
        // other, inputs and results staying on the GPU (a 2 000-landmark update: 0.09 s against 0.5 s).
        VSLAM_TRY(ws_reserve(ctx, cost_ws_bytes(n) + newton_ws_bytes(n)));
        for (int b = 0; b < batch; ++b) {
            CostDevice c;
            cost_carve(ctx->ws, n, &c);
            hipStream_t s = ctx->stream;
            VSLAM_HIP_TRY(ctx, hipMemcpyAsync(c.mu, mu + (size_t)b * n, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
            VSLAM_HIP_TRY(ctx, hipMemcpyAsync(c.S, S + (size_t)b * nn, nn * 8, hipMemcpyDeviceToDevice, s));
            VSLAM_HIP_TRY(ctx, hipMemcpyAsync(c.y, y + (size_t)b * 2 * L, (size_t)2 * L * 8, hipMemcpyDeviceToDevice, s));
            VSLAM_HIP_TRY(ctx, hipMemcpyAsync(c.vis, visible + (size_t)b * L, (size_t)L, hipMemcpyDeviceToDevice, s));
            int code = 1, it = 0, ev = 0, st = 0;
            VSLAM_TRY(iekf_newton_device(ctx, k, sigma, &c, static_cast<char *>(ctx->ws) + cost_ws_bytes(n), mu_post + (size_t)b * n,
                                         S_post + (size_t)b * nn, n, &code, &it, &ev, &st, /*device_out=*/true));
            if (fmin_code) fmin_code[b] = code;
            if (iters) iters[b] = it;
            if (status) status[b] = st;
        }
        return VSLAM_OK;
    }
    // instances per pass: the work matrices of one pass stay under 64 GiB of the 288 GB (VSLAM_BANK_PASS_GIB: another limit).
    // Larger passes amortise the rounds in which few filters are still busy: 4 096 filters of 150 landmarks take 279 ms in
    // one pass (14 700 updates/s), 2 048 filters 153 ms in one pass against 165 ms in two.
    size_t pass_gib = 64;
    if (const char *e = getenv("VSLAM_BANK_PASS_GIB")) { const long v = atol(e); if (v >= 1 && v <= 200) pass_gib = (size_t)v; }
    size_t chunk = (pass_gib << 30) / (8 * nn * 8);
    chunk = std::max<size_t>(1, std::min<size_t>(chunk, 16384));
    for (int b0 = 0; b0 < batch; b0 += (int)chunk) {
        const int B = std::min<int>((int)chunk, batch - b0);
        VSLAM_TRY(iekf_bank_chunk(ctx, k, sigma, n, B, mu + (size_t)b0 * n, S + (size_t)b0 * nn, y + (size_t)b0 * 2 * L,
                                  visible + (size_t)b0 * L, mu_post + (size_t)b0 * n, S_post + (size_t)b0 * nn,
                                  fmin_code ? fmin_code + b0 : nullptr, iters ? iters + b0 : nullptr, status ? status + b0 : nullptr));
    }
    return VSLAM_OK;
}
