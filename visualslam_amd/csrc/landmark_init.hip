// Initialisation of new landmarks: reference initializeNewLandmark (src/SLAM.cpp:905-941) with
// generateOpticalRay (src/SLAM.cpp:876-903).  The state grows by three rows per landmark, the factor by a
// diagonal block kappa * I, and the landmark starts on the optical ray through its keypoint,
//   r = rCNn + Rnc(Theta) * ray_length * K^-1 [u' v' 1]^T,
// (u', v') = the keypoint with the lens distortion removed.  The reference appends one landmark per call
// from the same predicted pose, so M landmarks appended at once give the same state as M calls.
//
// The reference removes the distortion with cv::undistortPoints(pixel, K, dist, R = I, P = [K | 0]); that
// routine is restated here from its documentation: normalise with K, five fixed-point steps
//   (x, y) <- ((x0, y0) - tangential/prism(x, y)) * (1 + k4 r2 + k5 r2^2 + k6 r2^3) / (1 + k1 r2 + k2 r2^2 + k3 r2^3),
// back to pixels with K, through cv::Point2f (float) on the way in and out.  The coefficients are read in
// OpenCV's order k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 whatever the calibration flag says, as OpenCV reads them.
// Compiled without FMA contraction so that host and device agree on the float roundings.
#include "landmark_math.hpp"

#include <cmath>

namespace vslam {

struct RayK {
    double fx, fy, cx, cy, d[12], length;
};

__host__ __device__ inline void undistorted_ray(const RayK &k, double u, double v, double *r)
{
    const double k1 = k.d[0], k2 = k.d[1], p1 = k.d[2], p2 = k.d[3], k3 = k.d[4], k4 = k.d[5], k5 = k.d[6], k6 = k.d[7];
    const double s1 = k.d[8], s2 = k.d[9], s3 = k.d[10], s4 = k.d[11];
    const double x0 = ((double)(float)u - k.cx) / k.fx, y0 = ((double)(float)v - k.cy) / k.fy;
    double x = x0, y = y0;
    for (int it = 0; it < 5; ++it) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + s1 * r2 + s2 * r2 * r2;
        const double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + s3 * r2 + s4 * r2 * r2;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    const float uu = (float)(x * k.fx + k.cx), vv = (float)(y * k.fy + k.cy);
    r[0] = k.length * (((double)uu - k.cx) / k.fx);
    r[1] = k.length * (((double)vv - k.cy) / k.fy);
    r[2] = k.length;
}

// mu_out[0..n) = mu, mu_out[n + 3 t ..] = landmark t on its optical ray
// (instance blockIdx.y of a bank: inputs n, n*n, 2*M and outputs n+3M, (n+3M)^2 doubles apart when batched)
__global__ void init_mu_kernel(const double *mu, int n, const double *pixels, int M, RayK k, double *mu_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    mu += (size_t)blockIdx.y * n; pixels += (size_t)blockIdx.y * 2 * M; mu_out += (size_t)blockIdx.y * (n + 3 * M);
    if (i < n) { mu_out[i] = mu[i]; return; }
    const int t = i - n;
    if (t >= M) return;
    double r[3];
    undistorted_ray(k, pixels[2 * t], pixels[2 * t + 1], r);
    const Pose p = make_pose(mu + 6);
    V3 a; a.x = r[0]; a.y = r[1]; a.z = r[2];
    const V3 w = Rfwd(p, a);
    mu_out[n + 3 * t] = p.c[0] + w.x;
    mu_out[n + 3 * t + 1] = p.c[1] + w.y;
    mu_out[n + 3 * t + 2] = p.c[2] + w.z;
}

// S_out = [S 0; 0 kappa I], m = n + 3 M rows and columns
__global__ void init_S_kernel(const double *S, int ldS, int n, int m, double kappa, double *S_out, int ldo)
{
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    S += (size_t)blockIdx.z * ldS * n; S_out += (size_t)blockIdx.z * ldo * m;
    if (r >= m) return;
    double v = (r == c) ? kappa : 0.0;
    if (r < n && c < n) v = S[(size_t)c * ldS + r];
    S_out[(size_t)c * ldo + r] = v;
}

static bool ray_constants(const vslam_camera *cam, double length, RayK *k)
{
    if (!cam || !(cam->fx > 0.0 && cam->fy > 0.0) || !std::isfinite(cam->fx) || !std::isfinite(cam->fy) || !std::isfinite(cam->cx) ||
        !std::isfinite(cam->cy) || !std::isfinite(length))
        return false;
    k->fx = cam->fx; k->fy = cam->fy; k->cx = cam->cx; k->cy = cam->cy; k->length = length;
    for (int i = 0; i < 12; ++i) {
        if (!std::isfinite(cam->d[i])) return false;
        k->d[i] = cam->d[i];
    }
    return true;
}

static vslam_status check_init(vslam_ctx *ctx, const vslam_camera *cam, const void *mu, const void *S, int n, int ldS, const void *pixels, int M,
                               double length, double kappa, const void *mu_out, const void *S_out, int ldSout, RayK *k)
{
    if (!mu || !S || !pixels || !mu_out || !S_out) return fail(ctx, VSLAM_ERR_ARG, "initialize_landmarks: null pointer");
    if (n < 12 || (n - 12) % 3 != 0 || ldS < n) return fail(ctx, VSLAM_ERR_SHAPE, "initialize_landmarks: n=%d is not 12+3L or ldS < n", n);
    if (M < 1 || M > (1 << 20)) return fail(ctx, VSLAM_ERR_SHAPE, "initialize_landmarks: %d new landmarks", M);
    if (ldSout < n + 3 * M) return fail(ctx, VSLAM_ERR_SHAPE, "initialize_landmarks: ldSout=%d < %d", ldSout, n + 3 * M);
    if (n + 3 * M > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "initialize_landmarks: %d states exceed 65535", n + 3 * M);
    if (!std::isfinite(kappa)) return fail(ctx, VSLAM_ERR_ARG, "initialize_landmarks: kappa is not finite");
    if (!ray_constants(cam, length, k)) return fail(ctx, VSLAM_ERR_ARG, "initialize_landmarks: camera constants or ray length are not usable");
    return VSLAM_OK;
}

static vslam_status launch_init(vslam_ctx *ctx, const RayK &k, const double *mu, const double *S, int n, int ldS, const double *pixels, int M,
                                double kappa, double *mu_out, double *S_out, int ldSout, int batch = 1)
{
    const int m = n + 3 * M;
    hipLaunchKernelGGL(init_mu_kernel, dim3((n + M + 255) / 256, batch), dim3(256), 0, ctx->stream, mu, n, pixels, M, k, mu_out);
    hipLaunchKernelGGL(init_S_kernel, dim3((m + 255) / 256, m, batch), dim3(256), 0, ctx->stream, S, ldS, n, m, kappa, S_out, ldSout);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" int vslam_optical_ray(const vslam_camera *cam, double u, double v, double ray_length, double *ray)
{
    RayK k;
    if (!ray || !ray_constants(cam, ray_length, &k)) return 1;
    undistorted_ray(k, u, v, ray);
    return 0;
}

extern "C" vslam_status vslam_initialize_landmarks_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mu, const double *S, int n, int ldS,
                                                       const double *pixels, int M, double ray_length, double kappa, double *mu_out,
                                                       double *S_out, int ldSout)
{
    if (!ctx) return VSLAM_ERR_ARG;
    RayK k;
    VSLAM_TRY(check_init(ctx, cam, mu, S, n, ldS, pixels, M, ray_length, kappa, mu_out, S_out, ldSout, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_init(ctx, k, mu, S, n, ldS, pixels, M, kappa, mu_out, S_out, ldSout);
}

extern "C" vslam_status vslam_initialize_landmarks_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mu, const double *S, int n,
                                                             int batch, const double *pixels, int M, double ray_length, double kappa,
                                                             double *mu_out, double *S_out)
{
    if (!ctx) return VSLAM_ERR_ARG;
    RayK k;
    VSLAM_TRY(check_init(ctx, cam, mu, S, n, n, pixels, M, ray_length, kappa, mu_out, S_out, n + 3 * M, &k));
    if (batch < 1 || batch > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "initialize_landmarks: batch=%d out of range", batch);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_init(ctx, k, mu, S, n, n, pixels, M, kappa, mu_out, S_out, n + 3 * M, batch);
}

extern "C" vslam_status vslam_initialize_landmarks(vslam_ctx *ctx, const vslam_camera *cam, const double *mu, const double *S, int n, int ldS,
                                                   const double *pixels, int M, double ray_length, double kappa, double *mu_out,
                                                   double *S_out, int ldSout)
{
    if (!ctx) return VSLAM_ERR_ARG;
    RayK k;
    VSLAM_TRY(check_init(ctx, cam, mu, S, n, ldS, pixels, M, ray_length, kappa, mu_out, S_out, ldSout, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int m = n + 3 * M;
    const size_t nn = (size_t)n * n, mm = (size_t)m * m;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(n * 8) + Arena::pad(nn * 8) + Arena::pad((size_t)M * 16) + Arena::pad((size_t)m * 8) + Arena::pad(mm * 8)));
    Arena ar(ctx->ws);
    double *dmu = ar.take<double>(n), *dS = ar.take<double>(nn), *dpx = ar.take<double>(2 * (size_t)M);
    double *dmo = ar.take<double>(m), *dSo = ar.take<double>(mm);
    VSLAM_TRY(h2d(ctx, dmu, mu, (size_t)n * 8));
    VSLAM_TRY(h2d_2d(ctx, dS, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    VSLAM_TRY(h2d(ctx, dpx, pixels, (size_t)M * 16));
    VSLAM_TRY(launch_init(ctx, k, dmu, dS, n, n, dpx, M, kappa, dmo, dSo, m));
    VSLAM_TRY(d2h(ctx, mu_out, dmo, (size_t)m * 8));
    VSLAM_TRY(d2h_2d(ctx, S_out, (size_t)ldSout * 8, dSo, (size_t)m * 8, (size_t)m * 8, m));
    return stream_finish(ctx);
}
