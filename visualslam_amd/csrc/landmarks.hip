// Landmark bookkeeping on the square-root factor: removeBadLandmarks (reference src/SLAM.cpp:796-820)
// drops the three columns of a landmark from S and re-triangularises the n x (n-3) matrix with the
// Householder QR of qr.hip (same reflector signs as the reference's Eigen::HouseholderQR).
#include "linalg.hpp"

namespace vslam {

// A (n x (n-3), ld lda) = S without columns c0..c0+2.
__global__ void drop_columns_kernel(const double *S, int ldS, int n, int c0, double *A, int lda)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < n) A[(size_t)c * lda + r] = S[(size_t)(c < c0 ? c : c + 3) * ldS + r];
}

__global__ void drop_rows_kernel(const double *mu, int n, int r0, double *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n - 3) out[i] = mu[i < r0 ? i : i + 3];
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_remove_landmark(vslam_ctx *ctx, const double *mu, const double *S, int n, int ldS, int j,
                                              double *mu_out, double *S_out, int ldSout)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!mu || !S || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "remove_landmark: null pointer");
    if (n < 15 || (n - 12) % 3 != 0 || ldS < n) return fail(ctx, VSLAM_ERR_SHAPE, "remove_landmark: n=%d is not 12+3L (L >= 1) or ldS < n", n);
    const int L = (n - 12) / 3, m = n - 3;
    if (j < 0 || j >= L) return fail(ctx, VSLAM_ERR_ARG, "remove_landmark: landmark %d of %d", j, L);
    if (ldSout < m) return fail(ctx, VSLAM_ERR_SHAPE, "remove_landmark: ldSout=%d < %d", ldSout, m);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(ws_reserve(ctx, 2 * Arena::pad(n * 8) + 2 * Arena::pad(nn * 8) + qr_ws_bytes(n, m, 1)));
    Arena ar(ctx->ws);
    double *dmu = ar.take<double>(n), *dmo = ar.take<double>(n), *dS = ar.take<double>(nn), *dA = ar.take<double>(nn);
    void *qws = ar.take<char>(qr_ws_bytes(n, m, 1));
    hipStream_t s = ctx->stream;
    const int c0 = 12 + 3 * j;
    VSLAM_TRY(h2d(ctx, dmu, mu, n * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    hipLaunchKernelGGL(drop_rows_kernel, dim3((m + 255) / 256), dim3(256), 0, s, dmu, n, c0, dmo);
    hipLaunchKernelGGL(drop_columns_kernel, dim3((n + 255) / 256, m), dim3(256), 0, s, dS, n, n, c0, dA, n);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(qr_upper_dev(ctx, dA, n, m, n, 0, 1, qws));
    VSLAM_TRY(d2h(ctx, mu_out, dmo, (size_t)m * 8));
    VSLAM_TRY(d2h_2d(ctx, S_out, (size_t)ldSout * 8, dA, (size_t)n * 8, (size_t)m * 8, m));
    return stream_finish(ctx);
}
