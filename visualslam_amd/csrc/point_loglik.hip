// Point-landmark log-likelihood: value, gradient and exact Hessian, assembled
// into the dense state gradient / Hessian by one kernel launch.
//
// Replaces PointLogLikelihood::operator() (reference src/model.cpp:408-437),
// whose gradient/Hessian come from n and n(n+1)/2 forward-mode passes; here
// every landmark's 9x9 block is evaluated in closed form by one lane.
//
// Work decomposition (grid.x = 1 + panels, grid.y = batch instance):
//   block 0  "camera block": loops over all L landmarks, 64 per wavefront.
//            Each lane evaluates one landmark, stores its rows of the six dense
//            pose columns (rows 12+3j.. of columns 6..11, contiguous across
//            lanes), its gradient entries and cull code, and accumulates the
//            28 pose sums (6x6 block, 6-gradient, value) which are reduced by
//            a wavefront butterfly + a fixed-order LDS pass.  It also writes
//            the all-zero velocity columns 0..5.
//   block p  "landmark panel": P consecutive landmarks = 3P columns of H.
//            Streams zeros over its column range with 16-byte stores (H is
//            >97 % zeros and must be fully written: the reference's callee
//            resizes and fills the output), then one lane per landmark
//            re-evaluates that landmark with the same instruction sequence
//            (so H comes out exactly symmetric) and drops the 6x3 cross block
//            and the 3x3 diagonal block into place.
// The kernel is HBM-write bound: 8 n^2 bytes of H per instance.
#include "landmark_math.hpp"

namespace vslam {

struct LoglikArgs {
    const double *x; size_t sx;
    const double *y; size_t sy;
    const uint8_t *vis; size_t sv;
    double *value;
    double *g; size_t sg;
    double *H; size_t sH;
    int32_t *code;
    int ldH, n, L, P;
    double is2, lconst;
};

constexpr int kThreads = 256;

// Cooperative zero fill of p[0..count) by the whole block; p is 8-byte aligned.
__device__ __forceinline__ void block_zero(double *p, size_t count)
{
    if (count == 0) return;
    const int tid = threadIdx.x;
    size_t head = (reinterpret_cast<uintptr_t>(p) & 8) ? 1 : 0;
    if (head && tid == 0) p[0] = 0.0;
    double2 *p2 = reinterpret_cast<double2 *>(p + head);
    size_t n2 = (count - head) >> 1;
    const double2 z = make_double2(0.0, 0.0);
    size_t i = tid;
    for (; i + 3 * kThreads < n2; i += 4 * kThreads) {
        p2[i] = z; p2[i + kThreads] = z; p2[i + 2 * kThreads] = z; p2[i + 3 * kThreads] = z;
    }
    for (; i < n2; i += kThreads) p2[i] = z;
    if (((count - head) & 1) && tid == 0) p[count - 1] = 0.0;
}

__device__ __forceinline__ void block_zero_cols(double *H, int ldH, int n, int c0, int c1)
{
    if (c1 <= c0) return;
    if (ldH == n) block_zero(H + (size_t)c0 * ldH, (size_t)(c1 - c0) * n);
    else
        for (int c = c0; c < c1; ++c) block_zero(H + (size_t)c * ldH, n);
}

template <int ORDER>
__global__ __launch_bounds__(kThreads) void point_loglik_kernel(CamK cam, LoglikArgs a)
{
    const int b = blockIdx.y, tid = threadIdx.x;
    const double *x = a.x + (size_t)b * a.sx;
    const double *y = a.y + (size_t)b * a.sy;
    const uint8_t *vis = a.vis + (size_t)b * a.sv;
    double *g = (ORDER >= 1) ? a.g + (size_t)b * a.sg : nullptr;
    double *H = (ORDER >= 2) ? a.H + (size_t)b * a.sH : nullptr;
    const int n = a.n, L = a.L, ldH = a.ldH;
    const bool is_cam = blockIdx.x == 0;

    int jb, je;
    if (is_cam) { jb = 0; je = L; }
    else { jb = (blockIdx.x - 1) * a.P; je = min(L, jb + a.P); }

    if (ORDER >= 2) {
        if (is_cam) block_zero_cols(H, ldH, n, 0, 6);
        else {
            block_zero_cols(H, ldH, n, 12 + 3 * jb, 12 + 3 * je);
            __syncthreads();   // zeros are ordered before this block's value stores
        }
    }

    double acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0.0;

    if (jb + tid < je) {
        const Pose pose = make_pose(x + 6);
        for (int j = jb + tid; j < je; j += kThreads) {
            Terms t;
            landmark_terms<ORDER>(cam, pose, x + 12 + 3 * j, y + 2 * j, vis[j] != 0, a.is2, a.lconst, t);
            const double Nf[3][3] = {{t.N[0], t.N[1], t.N[2]}, {t.N[1], t.N[3], t.N[4]}, {t.N[2], t.N[4], t.N[5]}};
            if (is_cam) {
                acc[0] += t.ell;
                if (a.code) a.code[(size_t)b * L + j] = t.code;
                if (ORDER >= 1) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) { acc[1 + i] += t.gr[i]; acc[4 + i] += t.gT[i]; g[12 + 3 * j + i] = t.gr[i]; }
                }
                if (ORDER >= 2) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) { acc[7 + i] += t.N[i]; acc[22 + i] += t.F[i]; }
#pragma unroll
                    for (int i = 0; i < 9; ++i) acc[13 + i] += t.E[i];
                    // rows 12+3j.. of the pose columns: H[r_i][c_a] = -N[i][a], H[r_i][Theta_k] = E[i][k]
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        double *col = H + (size_t)(6 + c) * ldH + 12 + 3 * j;
                        double *colT = H + (size_t)(9 + c) * ldH + 12 + 3 * j;
#pragma unroll
                        for (int i = 0; i < 3; ++i) { col[i] = -Nf[i][c]; colT[i] = t.E[i * 3 + c]; }
                    }
                }
            } else if (ORDER >= 2) {
                // landmark columns 12+3j+i: rows 6..8 = -N[:,i], rows 9..11 = E[i][:], rows 12+3j.. = N[:,i]
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double *col = H + (size_t)(12 + 3 * j + i) * ldH;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        col[6 + r] = -Nf[r][i];
                        col[9 + r] = t.E[i * 3 + r];
                        col[12 + 3 * j + r] = Nf[r][i];
                    }
                }
            }
        }
    }
    if (!is_cam) return;

    // ---- reduce the 28 pose sums: wavefront butterfly, then 4 partials in fixed order ----
    __shared__ double red[kThreads / 64][28];
    constexpr int NRED = (ORDER == 0) ? 1 : (ORDER == 1) ? 7 : 28;
#pragma unroll
    for (int i = 0; i < NRED; ++i) acc[i] = wave_sum(acc[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NRED; ++i) red[tid >> 6][i] = acc[i];
    }
    __syncthreads();
    __shared__ double tot[28];
    if (tid < NRED) tot[tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    __syncthreads();
    if (tid == 0) a.value[b] = tot[0];
    if (ORDER >= 1 && tid < 12) {
        double v = 0.0;
        if (tid >= 6 && tid < 9) v = -tot[1 + (tid - 6)];
        else if (tid >= 9) v = tot[4 + (tid - 9)];
        g[tid] = v;
    }
    if (ORDER >= 2 && tid < 72) {
        // head rows 0..11 of columns 6..11
        const int c = tid / 12, r = tid % 12;          // column 6+c, row r
        const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
        double v = 0.0;
        if (r >= 6) {
            const int rr = r - 6;
            if (c < 3 && rr < 3) v = tot[7 + sym[rr][c]];              // sum N
            else if (c >= 3 && rr >= 3) v = tot[22 + sym[rr - 3][c - 3]];  // sum F
            else if (c < 3) v = -tot[13 + c * 3 + (rr - 3)];           // H[Theta_k][c_i] = -E[i][k]
            else v = -tot[13 + rr * 3 + (c - 3)];                      // H[c_i][Theta_k] = -E[i][k]
        }
        H[(size_t)(6 + c) * ldH + r] = v;
    }
}

static int pick_panel(int L, int batch)
{
    // aim for a few thousand workgroups so all 256 CUs stream stores
    long total = (long)L * batch;
    int P = (int)((total + 4095) / 4096);
    if (P < 1) P = 1;
    if (P > kThreads) P = kThreads;
    return P;
}

vslam_status launch_point_loglik(vslam_ctx *ctx, const CamK &cam, int batch, int n,
                                 const double *x, size_t sx, const double *y, size_t sy,
                                 const uint8_t *vis, size_t sv, double sigma,
                                 double *value, double *g, size_t sg, double *H, int ldH, size_t sH,
                                 int32_t *code)
{
    if (!x || !y || !vis || !value) return fail(ctx, VSLAM_ERR_ARG, "point_loglik: null input/value pointer");
    if (n < 15 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: n=%d is not 12+3L with L>=1", n);
    if (batch < 1 || batch > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: batch=%d out of range", batch);
    if (H && !g) return fail(ctx, VSLAM_ERR_ARG, "point_loglik: H requested without g");
    if (H && ldH < n) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: ldH=%d < n=%d", ldH, n);
    if (!(sigma > 0.0)) return fail(ctx, VSLAM_ERR_ARG, "point_loglik: sigma must be positive");
    LoglikArgs a;
    a.x = x; a.sx = sx; a.y = y; a.sy = sy; a.vis = vis; a.sv = sv;
    a.value = value; a.g = g; a.sg = sg; a.H = H; a.sH = sH; a.code = code;
    a.ldH = ldH; a.n = n; a.L = (n - 12) / 3;
    a.P = pick_panel(a.L, batch);
    a.is2 = 1.0 / (sigma * sigma);
    a.lconst = -std::log(2.0 * 3.14159265358979323846264338327950288) - 2.0 * std::log(sigma);
    const int panels = (a.L + a.P - 1) / a.P;
    dim3 block(kThreads);
    if (H) {
        dim3 grid(1 + panels, batch);
        hipLaunchKernelGGL(point_loglik_kernel<2>, grid, block, 0, ctx->stream, cam, a);
    } else if (g) {
        hipLaunchKernelGGL(point_loglik_kernel<1>, dim3(1, batch), block, 0, ctx->stream, cam, a);
    } else {
        hipLaunchKernelGGL(point_loglik_kernel<0>, dim3(1, batch), block, 0, ctx->stream, cam, a);
    }
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_point_loglik_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, int batch, int n,
                                                     const double *x, size_t stride_x, const double *y, size_t stride_y,
                                                     const uint8_t *visible, size_t stride_vis, double sigma,
                                                     double *value, double *g, size_t stride_g,
                                                     double *H, int ldH, size_t stride_H, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_point_loglik(ctx, k, batch, n, x, stride_x, y, stride_y, visible, stride_vis, sigma, value, g,
                               stride_g, H, ldH, stride_H, code);
}

extern "C" vslam_status vslam_point_loglik(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n,
                                           const double *y, const uint8_t *visible, double sigma,
                                           double *value, double *g, double *H, int ldH, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!x || !y || !visible || !value) return fail(ctx, VSLAM_ERR_ARG, "point_loglik: null pointer");
    if (n < 15 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: n=%d is not 12+3L with L>=1", n);
    if (H && ldH < n) return fail(ctx, VSLAM_ERR_SHAPE, "point_loglik: ldH=%d < n=%d", ldH, n);
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int L = (n - 12) / 3;
    const size_t nn = (size_t)n * n;
    size_t bytes = Arena::pad(n * 8) * 2 + Arena::pad(2 * L * 8) + Arena::pad(L) + Arena::pad(L * 4) + Arena::pad(8) +
                   (H ? Arena::pad(nn * 8) : 0);
    VSLAM_TRY(ws_reserve(ctx, bytes));
    Arena ar(ctx->ws);
    double *dx = ar.take<double>(n), *dg = ar.take<double>(n), *dy = ar.take<double>(2 * L);
    uint8_t *dv = ar.take<uint8_t>(L);
    int32_t *dc = ar.take<int32_t>(L);
    double *dval = ar.take<double>(1);
    double *dH = H ? ar.take<double>(nn) : nullptr;
    hipStream_t s = ctx->stream;
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dx, x, n * 8, hipMemcpyHostToDevice, s));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dy, y, 2 * L * 8, hipMemcpyHostToDevice, s));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dv, visible, L, hipMemcpyHostToDevice, s));
    VSLAM_TRY(launch_point_loglik(ctx, k, 1, n, dx, 0, dy, 0, dv, 0, sigma, dval, g ? dg : (H ? dg : nullptr), 0, dH, n, 0, dc));
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(value, dval, 8, hipMemcpyDeviceToHost, s));
    if (g) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(g, dg, n * 8, hipMemcpyDeviceToHost, s));
    if (code) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(code, dc, L * 4, hipMemcpyDeviceToHost, s));
    if (H) VSLAM_HIP_TRY(ctx, hipMemcpy2DAsync(H, (size_t)ldH * 8, dH, (size_t)n * 8, (size_t)n * 8, n, hipMemcpyDeviceToHost, s));
    VSLAM_HIP_TRY(ctx, hipStreamSynchronize(s));
    return VSLAM_OK;
}
