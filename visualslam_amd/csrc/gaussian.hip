// Square-root Gaussian algebra entry points: logGaussian, affineTransform's QR
// step (dense and fused with the landmark bundle), the IEKF posterior
// square-root, conditioning, and the joint cost CostJointDensity.
// Reference: src/gaussian.hpp:30-131,435-532, src/gaussian.cpp:19-36.
#include "landmark_math.hpp"
#include "linalg.hpp"

namespace vslam {

__global__ void vec_sub_kernel(const double *a, const double *b, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}

// value = -0.5 |z|^2 - n/2 log(2 pi) - sum log|S_ii|   (src/gaussian.hpp:41-45); g = -g_in in place
__global__ __launch_bounds__(256) void log_gaussian_value_kernel(const double *z, const double *S, int n, int ldS,
                                                                 double *value, double *g)
{
    __shared__ double red[2][4];
    double sq = 0.0, ls = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        sq += z[i] * z[i];
        ls += log(fabs(S[(size_t)i * ldS + i]));
        if (g) g[i] = -g[i];
    }
    sq = wave_sum(sq); ls = wave_sum(ls);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = ls; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        double b = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        value[0] = -0.5 * a - (double)n / 2 * log(2 * 3.14159265358979323846264338327950288) - b;
    }
}

// A[i,j] = Q[j,i] / sqrt(v[i])   (diag(v^-1/2) Q^T, src/gaussian.hpp:513), tiled transpose through LDS
__global__ __launch_bounds__(256) void scaled_transpose_kernel(const double *Q, const double *v, int n, double *A)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;      // A rows bx.., A cols by..
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        // read Q[row = by + tx, col = bx + r]  (contiguous in tx)
        int qr = by + tx, qc = bx + r;
        tile[r][tx] = (qr < n && qc < n) ? Q[(size_t)qc * n + qr] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int ar = bx + tx, ac = by + r;                         // write A[ar, ac] = Q[ac, ar] / sqrt(v[ar])
        if (ar < n && ac < n) A[(size_t)ac * n + ar] = tile[tx][r] * (1.0 / sqrt(v[ar]));
    }
}

// out = -(a + b) elementwise over an n x n (ld n) matrix or a vector
__global__ void neg_sum_kernel(const double *a, const double *b, double *out, size_t count)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) out[i] = -(a[i] + b[i]);
}
__global__ void neg_sum_scalar_kernel(const double *a, const double *b, double *out) { out[0] = -(a[0] + b[0]); }

// Compact per-landmark measurement Jacobian for the fused affine transform.
struct JacRec { double Jc[2][6]; double Jl[2][3]; };

__device__ __forceinline__ Pose rot_pose2(const double *x, int layout, const Pose &pose)
{
    if (layout == VSLAM_LAYOUT_STRICT) return pose;
    Pose r = pose;
    sincos(x[3], &r.sp, &r.cp); sincos(x[4], &r.st, &r.ct); sincos(x[5], &r.ss, &r.cs);
    return r;
}

// (blockIdx.y: instance of a bank; x, h, code and rec are 12+3L, 2L, L and L entries apart)
__global__ void bundle_jac_compact_kernel(CamK k, const double *x, int L, int layout, double *h, int32_t *code, JacRec *rec)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L) return;
    x += (size_t)blockIdx.y * (12 + 3 * L); h += (size_t)blockIdx.y * 2 * L; rec += (size_t)blockIdx.y * L;
    if (code) code += (size_t)blockIdx.y * L;
    const Pose p = make_pose(x + 6);
    const Pose pr = rot_pose2(x, layout, p);
    const double *r = x + 12 + 3 * j;
    V3 d = landmark_offset(p, r);
    V3 q = RxT<0>(p, RyT<0>(p, RzT<0>(p, d)));
    double px[2];
    int c = project_q(k, q, px, nullptr, nullptr, nullptr);
    h[2 * j] = px[0]; h[2 * j + 1] = px[1];
    if (code) code[j] = c;
    JacRec o;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int a = 0; a < 6; ++a) o.Jc[i][a] = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) o.Jl[i][a] = 0.0;
    }
    if (q.z > 0.0) {
        double iz = 1.0 / q.z, u = q.x * iz, v = q.y * iz;
        Distort D;
        distort<1>(k, u, v, D);
        double a00 = k.fx * D.j00, a01 = k.fx * D.j01, a10 = k.fy * D.j10, a11 = k.fy * D.j11;
        V3 jg0 = {a00 * iz, a01 * iz, -iz * (a00 * u + a01 * v)};
        V3 jg1 = {a10 * iz, a11 * iz, -iz * (a10 * u + a11 * v)};
        V3 jp0 = Rfwd(p, jg0), jp1 = Rfwd(p, jg1);
        V3 Z0 = RzT<0>(pr, d), Z1 = RzT<1>(pr, d);
        V3 Y00 = RyT<0>(pr, Z0), Y01 = RyT<1>(pr, Z0), Y10 = RyT<0>(pr, Z1);
        V3 A0 = RxT<1>(pr, Y00), A1 = RxT<0>(pr, Y01), A2 = RxT<0>(pr, Y10);
        o.Jl[0][0] = jp0.x; o.Jl[0][1] = jp0.y; o.Jl[0][2] = jp0.z;
        o.Jl[1][0] = jp1.x; o.Jl[1][1] = jp1.y; o.Jl[1][2] = jp1.z;
#pragma unroll
        for (int a = 0; a < 3; ++a) { o.Jc[0][a] = -o.Jl[0][a]; o.Jc[1][a] = -o.Jl[1][a]; }
        o.Jc[0][3] = dot(jg0, A0); o.Jc[0][4] = dot(jg0, A1); o.Jc[0][5] = dot(jg0, A2);
        o.Jc[1][3] = dot(jg1, A0); o.Jc[1][4] = dot(jg1, A1); o.Jc[1][5] = dot(jg1, A2);
    }
    rec[j] = o;
}

// Top block of affineTransform's QR input without materialising C:
//   A[i, 2j+k] = sum_a S[i, cam0+a] Jc[k][a] + sum_a S[i, lm0+3j+a] Jl[k][a]      (i < n)
// and the bottom block SR = 0.01 I (rows n..n+2L).  One thread per (row, landmark).
// (blockIdx.z: instance of a bank; S, rec and A are ldS*n, L and ldA*2L entries apart)
__global__ __launch_bounds__(256) void s_times_ct_kernel(const double *S, int n, int ldS, const JacRec *rec, int L, int layout,
                                                         double *A, int ldA)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    const int m = 2 * L;
    if (i >= n + m) return;
    S += (size_t)blockIdx.z * ldS * n; rec += (size_t)blockIdx.z * L; A += (size_t)blockIdx.z * ldA * m;
    double *c0 = A + (size_t)(2 * j) * ldA, *c1 = c0 + ldA;
    if (i >= n) {
        const int r = i - n;
        c0[i] = (r == 2 * j) ? 0.01 : 0.0;
        c1[i] = (r == 2 * j + 1) ? 0.01 : 0.0;
        return;
    }
    const int cam0 = layout == VSLAM_LAYOUT_STRICT ? 6 : 0, lm0 = (layout == VSLAM_LAYOUT_STRICT ? 12 : 6) + 3 * j;
    const JacRec &J = rec[j];
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double s = S[(size_t)(cam0 + a) * ldS + i];
        s0 += s * J.Jc[0][a]; s1 += s * J.Jc[1][a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double s = S[(size_t)(lm0 + a) * ldS + i];
        s0 += s * J.Jl[0][a]; s1 += s * J.Jl[1][a];
    }
    c0[i] = s0; c1[i] = s1;
}

// dst[b] (m x m, ld m) = leading m x m block of src[b] (ld lds, stride_src entries apart)
// dst = the upper triangle of the leading m x m block of src, zeros below the diagonal (src still holds reflectors there)
__global__ void copy_block_batch_kernel(const double *src, int lds, size_t stride_src, double *dst, int m)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < m) dst[(size_t)blockIdx.z * m * m + (size_t)c * m + r] = r <= c ? src[(size_t)blockIdx.z * stride_src + (size_t)c * lds + r] : 0.0;
}

// mux_out[i] = mu[ny+i] + sum_k S2[k,i] t[k]   (src/gaussian.cpp:34)
__global__ void condition_mean_kernel(const double *muyx, const double *Syx, int ny, int nx, int ld, const double *t, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx) return;
    const double *col = Syx + (size_t)(ny + i) * ld;
    double s = muyx[ny + i];
    for (int k = 0; k < ny; ++k) s += col[k] * t[k];
    out[i] = s;
}

__global__ void copy2d_kernel(const double *src, int lds, double *dst, int ldd, int rows, int cols)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (r < rows && c < cols) dst[(size_t)c * ldd + r] = src[(size_t)c * lds + r];
}

// ---- device-side compositions (all pointers device, queued on ctx->stream) ----

// value/g/H of logGaussian; g and H may be null.  tmp: n doubles; Wbuf: n*n doubles when H; tws: trtri workspace.
vslam_status log_gaussian_dev(vslam_ctx *ctx, const double *x, const double *mu, const double *S, int n, int ldS,
                              double *value, double *g, double *H, double *tmp, double *Wbuf, void *tws)
{
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(vec_sub_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, mu, tmp, n);
    VSLAM_TRY(trsv_upper_dev(ctx, S, n, ldS, true, tmp, tmp));            // z = S^-T (x - mu)
    if (g) VSLAM_TRY(trsv_upper_dev(ctx, S, n, ldS, false, tmp, g));      // S^-1 z (negated below)
    hipLaunchKernelGGL(log_gaussian_value_kernel, dim3(1), dim3(256), 0, s, tmp, S, n, ldS, value, g);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    if (H) {
        VSLAM_TRY(trtri_upper_dev(ctx, S, n, ldS, Wbuf, n, tws, /*zero_lower=*/false));      // W enters the product below as a triangular operand
        GemmDesc d;                                                       // H = -W W^T
        d.M = n; d.N = n; d.K = n; d.transA = false; d.transB = true; d.amode = A_UPPER; d.b_upper = true; d.alpha = -1.0;
        d.A = Wbuf; d.lda = n; d.B = Wbuf; d.ldb = n; d.C = H; d.ldc = n;
        d.c_upper = true;                                                 // the product is symmetric: upper tiles, then mirrored
        VSLAM_TRY(gemm_f64(ctx, d));
        VSLAM_TRY(mirror_upper_dev(ctx, H, n, n, 0, 1));
    }
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_log_gaussian(vslam_ctx *ctx, const double *x, const double *mu, const double *S,
                                           int n, int ldS, double *value, double *g, double *H)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!x || !mu || !S || !value) return fail(ctx, VSLAM_ERR_ARG, "log_gaussian: null pointer");
    if (n < 1 || ldS < n) return fail(ctx, VSLAM_ERR_SHAPE, "log_gaussian: n=%d ldS=%d", n, ldS);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    size_t bytes = 4 * Arena::pad(n * 8) + Arena::pad(nn * 8) + Arena::pad(64) + (H ? 2 * Arena::pad(nn * 8) + trtri_ws_bytes(n) : 0);
    VSLAM_TRY(ws_reserve(ctx, bytes));
    Arena ar(ctx->ws);
    double *dx = ar.take<double>(n), *dmu = ar.take<double>(n), *dtmp = ar.take<double>(n), *dg = ar.take<double>(n);
    double *dS = ar.take<double>(nn), *dval = ar.take<double>(1);
    double *dH = H ? ar.take<double>(nn) : nullptr, *dW = H ? ar.take<double>(nn) : nullptr;
    void *tws = H ? ar.take<char>(trtri_ws_bytes(n)) : nullptr;
    VSLAM_TRY(h2d(ctx, dx, x, n * 8));
    VSLAM_TRY(h2d(ctx, dmu, mu, n * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    VSLAM_TRY(log_gaussian_dev(ctx, dx, dmu, dS, n, n, dval, (g || H) ? dg : nullptr, dH, dtmp, dW, tws));
    VSLAM_TRY(d2h(ctx, value, dval, 8));
    if (g) VSLAM_TRY(d2h(ctx, g, dg, n * 8));
    if (H) VSLAM_TRY(d2h(ctx, H, dH, nn * 8));
    return stream_finish(ctx);
}

extern "C" vslam_status vslam_affine_sqrt(vslam_ctx *ctx, const double *S, int n, int ldS, const double *C, int m, int ldC,
                                          const double *SR, int ldSR, double *Syy, int ldSyy)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!S || !C || !SR || !Syy) return fail(ctx, VSLAM_ERR_ARG, "affine_sqrt: null pointer");
    if (n < 1 || m < 1 || ldS < n || ldC < m || ldSR < m || ldSyy < m) return fail(ctx, VSLAM_ERR_SHAPE, "affine_sqrt: bad dimensions");
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rows = n + m;
    const size_t bS = Arena::pad((size_t)n * n * 8), bC = Arena::pad((size_t)m * n * 8), bA = Arena::pad((size_t)rows * m * 8);
    VSLAM_TRY(ws_reserve(ctx, bS + bC + bA + qr_ws_bytes(rows, m, 1)));
    Arena ar(ctx->ws);
    double *dS = ar.take<double>((size_t)n * n), *dC = ar.take<double>((size_t)m * n), *dA = ar.take<double>((size_t)rows * m);
    void *qws = ar.take<char>(qr_ws_bytes(rows, m, 1));
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    VSLAM_TRY(h2d_2d(ctx, dC, (size_t)m * 8, C, (size_t)ldC * 8, (size_t)m * 8, n));
    VSLAM_TRY(h2d_2d(ctx, dA + n, (size_t)rows * 8, SR, (size_t)ldSR * 8, (size_t)m * 8, m));
    GemmDesc d;                                          // top block: S * C^T   (src/gaussian.hpp:119)
    d.M = n; d.N = m; d.K = n; d.transA = false; d.transB = true;
    d.A = dS; d.lda = n; d.B = dC; d.ldb = m; d.C = dA; d.ldc = rows;
    VSLAM_TRY(gemm_f64(ctx, d));
    VSLAM_TRY(qr_upper_dev(ctx, dA, rows, m, rows, 0, 1, qws));
    VSLAM_TRY(d2h_2d(ctx, Syy, (size_t)ldSyy * 8, dA, (size_t)rows * 8, (size_t)m * 8, m));
    return stream_finish(ctx);
}

extern "C" vslam_status vslam_affine_bundle(vslam_ctx *ctx, const vslam_camera *cam, const double *mux, const double *Sxx,
                                            int n, int ldS, int layout, double *muy, double *Syy, int ldSyy, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (n < 12 || (n - 12) % 3 != 0 || ldS < n) return fail(ctx, VSLAM_ERR_SHAPE, "affine_bundle: n=%d is not 12+3L or ldS < n", n);
    if (layout != VSLAM_LAYOUT_COMPAT && layout != VSLAM_LAYOUT_STRICT) return fail(ctx, VSLAM_ERR_ARG, "affine_bundle: bad layout %d", layout);
    if (!mux || !Sxx) return fail(ctx, VSLAM_ERR_ARG, "affine_bundle: null pointer");
    if (n == 12) return VSLAM_OK;        // a map without landmarks: the predicted measurement has no entries (muy, Syy are 0-dimensional)
    if (!muy || !Syy) return fail(ctx, VSLAM_ERR_ARG, "affine_bundle: null pointer");
    const int L = (n - 12) / 3, m = 2 * L, rows = n + m;
    if (ldSyy < m) return fail(ctx, VSLAM_ERR_SHAPE, "affine_bundle: ldSyy=%d < %d", ldSyy, m);
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t bytes = Arena::pad(n * 8) + Arena::pad((size_t)n * n * 8) + Arena::pad(m * 8) + Arena::pad(L * 4) +
                   Arena::pad((size_t)L * sizeof(JacRec)) + Arena::pad((size_t)rows * m * 8) + qr_ws_bytes(rows, m, 1);
    VSLAM_TRY(ws_reserve(ctx, bytes));
    Arena ar(ctx->ws);
    double *dx = ar.take<double>(n), *dS = ar.take<double>((size_t)n * n), *dh = ar.take<double>(m);
    int32_t *dc = ar.take<int32_t>(L);
    JacRec *rec = ar.take<JacRec>(L);
    double *dA = ar.take<double>((size_t)rows * m);
    void *qws = ar.take<char>(qr_ws_bytes(rows, m, 1));
    hipStream_t s = ctx->stream;
    VSLAM_TRY(h2d(ctx, dx, mux, n * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)n * 8, Sxx, (size_t)ldS * 8, (size_t)n * 8, n));
    hipLaunchKernelGGL(bundle_jac_compact_kernel, dim3((L + 63) / 64), dim3(64), 0, s, k, dx, L, layout, dh, dc, rec);
    hipLaunchKernelGGL(s_times_ct_kernel, dim3((rows + 255) / 256, L), dim3(256), 0, s, dS, n, n, rec, L, layout, dA, rows);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(qr_upper_dev(ctx, dA, rows, m, rows, 0, 1, qws, n));      // the noise rows are diagonal: column c is zero below row n + c
    VSLAM_TRY(d2h(ctx, muy, dh, m * 8));
    if (code) VSLAM_TRY(d2h(ctx, code, dc, L * 4));
    VSLAM_TRY(d2h_2d(ctx, Syy, (size_t)ldSyy * 8, dA, (size_t)rows * 8, (size_t)m * 8, m));
    return stream_finish(ctx);
}

extern "C" vslam_status vslam_affine_bundle_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mux, const double *Sxx, int n,
                                                      int batch, int layout, double *muy, double *Syy, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (n < 12 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "affine_bundle: n=%d is not 12+3L", n);
    if (layout != VSLAM_LAYOUT_COMPAT && layout != VSLAM_LAYOUT_STRICT) return fail(ctx, VSLAM_ERR_ARG, "affine_bundle: bad layout %d", layout);
    if (batch < 1 || batch > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "affine_bundle: batch=%d out of range", batch);
    if (!mux || !Sxx) return fail(ctx, VSLAM_ERR_ARG, "affine_bundle: null pointer");
    if (n == 12) return VSLAM_OK;        // maps without landmarks: nothing to predict
    if (!muy || !Syy) return fail(ctx, VSLAM_ERR_ARG, "affine_bundle: null pointer");
    const int L = (n - 12) / 3, m = 2 * L, rows = n + m;
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t per_A = (size_t)rows * m;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad((size_t)batch * L * sizeof(JacRec)) + Arena::pad((size_t)batch * per_A * 8) + qr_ws_bytes(rows, m, batch)));
    Arena ar(ctx->ws);
    JacRec *rec = ar.take<JacRec>((size_t)batch * L);
    double *dA = ar.take<double>((size_t)batch * per_A);
    void *qws = ar.take<char>(qr_ws_bytes(rows, m, batch));
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(bundle_jac_compact_kernel, dim3((L + 63) / 64, batch), dim3(64), 0, s, k, mux, L, layout, muy, code, rec);
    hipLaunchKernelGGL(s_times_ct_kernel, dim3((rows + 255) / 256, L, batch), dim3(256), 0, s, Sxx, n, n, rec, L, layout, dA, rows);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    // rows n .. n+m-1 hold the diagonal noise factor: column c is zero below row n + c, and stays so under the reflectors of
    // the columns before it
    VSLAM_TRY(qr_upper_dev(ctx, dA, rows, m, rows, per_A, batch, qws, n, false));
    hipLaunchKernelGGL(copy_block_batch_kernel, dim3((m + 255) / 256, m, batch), dim3(256), 0, s, dA, rows, per_A, Syy, m);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

extern "C" vslam_status vslam_iekf_post_sqrt(vslam_ctx *ctx, const double *Q, const double *v, int n, double *S_out)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!Q || !v || !S_out) return fail(ctx, VSLAM_ERR_ARG, "iekf_post_sqrt: null pointer");
    if (n < 1) return fail(ctx, VSLAM_ERR_SHAPE, "iekf_post_sqrt: n=%d", n);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(ws_reserve(ctx, 2 * Arena::pad(nn * 8) + Arena::pad(n * 8) + qr_ws_bytes(n, n, 1)));
    Arena ar(ctx->ws);
    double *dQ = ar.take<double>(nn), *dA = ar.take<double>(nn), *dv = ar.take<double>(n);
    void *qws = ar.take<char>(qr_ws_bytes(n, n, 1));
    hipStream_t s = ctx->stream;
    VSLAM_TRY(h2d(ctx, dQ, Q, nn * 8));
    VSLAM_TRY(h2d(ctx, dv, v, n * 8));
    hipLaunchKernelGGL(scaled_transpose_kernel, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, s, dQ, dv, n, dA);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(qr_upper_dev(ctx, dA, n, n, n, 0, 1, qws));
    VSLAM_TRY(d2h(ctx, S_out, dA, nn * 8));
    return stream_finish(ctx);
}

extern "C" vslam_status vslam_condition_on_marginal(vslam_ctx *ctx, const double *muyx, const double *Syx, int ny, int nx, int ld,
                                                    const double *y, double *mux_out, double *Sx_out, int ldo)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!muyx || !Syx || !y || !mux_out || !Sx_out) return fail(ctx, VSLAM_ERR_ARG, "condition_on_marginal: null pointer");
    const int nt = ny + nx;
    if (ny < 1 || nx < 1 || ld < nt || ldo < nx) return fail(ctx, VSLAM_ERR_SHAPE, "condition_on_marginal: bad dimensions");
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ws_reserve(ctx, Arena::pad((size_t)nt * nt * 8) + 4 * Arena::pad(nt * 8) + Arena::pad((size_t)nx * nx * 8)));
    Arena ar(ctx->ws);
    double *dS = ar.take<double>((size_t)nt * nt), *dmu = ar.take<double>(nt), *dy = ar.take<double>(nt), *dt = ar.take<double>(nt),
           *dout = ar.take<double>(nt), *dSx = ar.take<double>((size_t)nx * nx);
    hipStream_t s = ctx->stream;
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)nt * 8, Syx, (size_t)ld * 8, (size_t)nt * 8, nt));
    VSLAM_TRY(h2d(ctx, dmu, muyx, nt * 8));
    VSLAM_TRY(h2d(ctx, dy, y, ny * 8));
    hipLaunchKernelGGL(vec_sub_kernel, dim3((ny + 255) / 256), dim3(256), 0, s, dy, dmu, dt, ny);
    VSLAM_TRY(trsv_upper_dev(ctx, dS, ny, nt, true, dt, dt));
    hipLaunchKernelGGL(condition_mean_kernel, dim3((nx + 63) / 64), dim3(64), 0, s, dmu, dS, ny, nx, nt, dt, dout);
    hipLaunchKernelGGL(copy2d_kernel, dim3((nx + 255) / 256, nx), dim3(256), 0, s, dS + (size_t)ny * nt + ny, nt, dSx, nx, nx, nx);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(d2h(ctx, mux_out, dout, nx * 8));
    VSLAM_TRY(d2h_2d(ctx, Sx_out, (size_t)ldo * 8, dSx, (size_t)nx * 8, (size_t)nx * 8, nx));
    return stream_finish(ctx);
}

// ---- cost state shared with the IEKF driver (iekf.cpp) ----
namespace vslam {

size_t cost_ws_bytes(int n)
{
    const size_t nn = (size_t)n * n;
    const int L = (n - 12) / 3;
    return 8 * Arena::pad(n * 8) + Arena::pad(2 * L * 8) + Arena::pad(L) + 5 * Arena::pad(nn * 8) + Arena::pad(64) + trtri_ws_bytes(n);
}

void cost_carve(void *ws, int n, CostDevice *c)
{
    Arena ar(ws);
    const size_t nn = (size_t)n * n;
    c->n = n; c->L = (n - 12) / 3;
    c->x = ar.take<double>(n); c->mu = ar.take<double>(n); c->tmp = ar.take<double>(n);
    c->gp = ar.take<double>(n); c->gl = ar.take<double>(n); c->g = ar.take<double>(n);
    ar.take<double>(n); ar.take<double>(n);
    c->y = ar.take<double>(2 * c->L);
    c->vis = ar.take<uint8_t>(c->L);
    c->S = ar.take<double>(nn); c->Hp = ar.take<double>(nn); c->Hl = ar.take<double>(nn); c->H = ar.take<double>(nn); c->W = ar.take<double>(nn);
    c->vals = ar.take<double>(8);
    c->tws = ar.take<char>(trtri_ws_bytes(n));
    c->prior_ready = false;
}

// V, g, H of CostJointDensity at the x held in c->x.  The prior Hessian does not
// depend on x, so it is formed on the first call only (the reference recomputes
// it every call, src/gaussian.hpp:459; the value is identical).
vslam_status cost_eval_dev(vslam_ctx *ctx, const CamK &k, double sigma, CostDevice *c)
{
    const int n = c->n;
    hipStream_t s = ctx->stream;
    VSLAM_TRY(log_gaussian_dev(ctx, c->x, c->mu, c->S, n, n, c->vals + 0, c->gp, c->prior_ready ? nullptr : c->Hp, c->tmp, c->W, c->tws));
    c->prior_ready = true;
    VSLAM_TRY(launch_point_loglik(ctx, k, 1, n, c->x, 0, c->y, 0, c->vis, 0, sigma, c->vals + 1, c->gl, 0, c->Hl, n, 0, nullptr, nullptr));
    hipLaunchKernelGGL(neg_sum_scalar_kernel, dim3(1), dim3(1), 0, s, c->vals + 0, c->vals + 1, c->vals + 2);
    hipLaunchKernelGGL(neg_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, s, c->gp, c->gl, c->g, (size_t)n);
    const size_t nn = (size_t)n * n;
    int blocks = (int)((nn + 1023) / 1024);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(neg_sum_kernel, dim3(blocks), dim3(256), 0, s, c->Hp, c->Hl, c->H, nn);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam

extern "C" vslam_status vslam_cost_joint_density(vslam_ctx *ctx, const vslam_camera *cam, const double *x,
                                                 const double *mu, const double *S, int n, int ldS,
                                                 const double *y, const uint8_t *visible, double sigma,
                                                 double *value, double *g, double *H)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (n < 12 || (n - 12) % 3 != 0 || ldS < n) return fail(ctx, VSLAM_ERR_SHAPE, "cost_joint_density: n=%d is not 12+3L or ldS < n", n);
    // n = 12: no landmark yet, the likelihood is the empty sum (src/model.cpp:392) and y / visible have no entries
    if (!x || !mu || !S || !value || !g || !H || (n > 12 && (!y || !visible))) return fail(ctx, VSLAM_ERR_ARG, "cost_joint_density: null pointer");
    if (!(sigma > 0.0)) return fail(ctx, VSLAM_ERR_ARG, "cost_joint_density: sigma must be positive");
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ws_reserve(ctx, cost_ws_bytes(n)));
    CostDevice c;
    cost_carve(ctx->ws, n, &c);
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(h2d(ctx, c.x, x, n * 8));
    VSLAM_TRY(h2d(ctx, c.mu, mu, n * 8));
    VSLAM_TRY(h2d_2d(ctx, c.S, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    VSLAM_TRY(h2d(ctx, c.y, y, 2 * c.L * 8));
    VSLAM_TRY(h2d(ctx, c.vis, visible, c.L));
    VSLAM_TRY(cost_eval_dev(ctx, k, sigma, &c));
    VSLAM_TRY(d2h(ctx, value, c.vals + 2, 8));
    VSLAM_TRY(d2h(ctx, g, c.g, n * 8));
    VSLAM_TRY(d2h(ctx, H, c.H, nn * 8));
    return stream_finish(ctx);
}
