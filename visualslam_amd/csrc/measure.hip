// Point-landmark measurement model: pixels, fixed measurement noise and the
// analytic 2 x n Jacobian, for one landmark or the whole bundle.
//
// Replaces worldToPixel<double> (reference src/cameraModel.hpp:32-216),
// MeasurementPointLandmarkSingle::operator() (src/measurementPointLandmark.hpp:10-304)
// and MeasurementPointLandmarkBundle::operator() (src/measurementPointLandmark.cpp:13-70).
//
// Bundle kernel decomposition (same shape as the log-likelihood kernel):
//   block 0   all L landmarks, one per lane: pixel, cull code and the six dense
//             pose columns of C (rows 2j,2j+1 are 16 contiguous bytes per lane,
//             contiguous across lanes); zero-fills the six all-zero columns.
//   block p   P landmarks: one streaming pass over their 3P columns of C and their
//             2P columns of SR, each 16-byte row pair written once (the 2x3 landmark
//             block and the 0.01 diagonal of SR where they fall, zeros elsewhere).
// C (2L x n) and SR (2L x 2L) are >97 % zeros but are dense outputs of the
// reference interface, so the kernel is HBM-write bound.
//
// Column layouts: STRICT puts d h/d rCNn at 6..8, d h/d Theta at 9..11 and
// d h/d r_j at 12+3j.  COMPAT reproduces the reference as written
// (src/measurementPointLandmark.hpp:282-296): columns 0..2, 3..5 and 6+3j, with
// the rotation-derivative matrices built from x(3), x(4), x(5).
#include "landmark_math.hpp"

namespace vslam {

constexpr int kThreads = 256;

__device__ __forceinline__ void block_zero_f64(double *p, size_t count)
{
    if (count == 0) return;
    const int tid = threadIdx.x;
    size_t head = (reinterpret_cast<uintptr_t>(p) & 8) ? 1 : 0;
    if (head && tid == 0) p[0] = 0.0;
    double2 *p2 = reinterpret_cast<double2 *>(p + head);
    size_t n2 = (count - head) >> 1;
    const double2 z = make_double2(0.0, 0.0);
    for (size_t i = tid; i < n2; i += kThreads) p2[i] = z;
    if (((count - head) & 1) && tid == 0) p[count - 1] = 0.0;
}
__device__ __forceinline__ void block_zero_cols_f64(double *M, int ld, int rows, int c0, int c1)
{
    if (c1 <= c0) return;
    if (ld == rows) block_zero_f64(M + (size_t)c0 * ld, (size_t)(c1 - c0) * rows);
    else
        for (int c = c0; c < c1; ++c) block_zero_f64(M + (size_t)c * ld, rows);
}

// Jacobian blocks of one landmark: JP (2x3) = d h / d r_j, JT (2x3) = d h / d Theta.
// pose_rot supplies the angles used for the rotation derivative (the true pose in
// STRICT, x(3..5) in COMPAT).  Mirrors the reference's guard: q_z <= 0 gives a
// zero Jacobian (src/measurementPointLandmark.hpp:175-181,208-214).
struct Jac { double JP[2][3], JT[2][3]; double px[2]; int code; };

__device__ __forceinline__ void landmark_jacobian(const CamK &k, const Pose &p, const Pose &pose_rot, const double *r, Jac &o)
{
    V3 d = landmark_offset(p, r);
    V3 q = RxT<0>(p, RyT<0>(p, RzT<0>(p, d)));
    o.code = project_q(k, q, o.px, nullptr, nullptr, nullptr);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) { o.JP[i][c] = 0.0; o.JT[i][c] = 0.0; }
    if (q.z <= 0.0) return;
    double iz = 1.0 / q.z, u = q.x * iz, v = q.y * iz;
    Distort D;
    distort<1>(k, u, v, D);
    double a00 = k.fx * D.j00, a01 = k.fx * D.j01, a10 = k.fy * D.j10, a11 = k.fy * D.j11;
    V3 jg0 = {a00 * iz, a01 * iz, -iz * (a00 * u + a01 * v)};
    V3 jg1 = {a10 * iz, a11 * iz, -iz * (a10 * u + a11 * v)};
    V3 jp0 = Rfwd(p, jg0), jp1 = Rfwd(p, jg1);          // rows of JrPCc * Rnc^T
    o.JP[0][0] = jp0.x; o.JP[0][1] = jp0.y; o.JP[0][2] = jp0.z;
    o.JP[1][0] = jp1.x; o.JP[1][1] = jp1.y; o.JP[1][2] = jp1.z;
    V3 Z0 = RzT<0>(pose_rot, d), Z1 = RzT<1>(pose_rot, d);
    V3 Y00 = RyT<0>(pose_rot, Z0), Y01 = RyT<1>(pose_rot, Z0), Y10 = RyT<0>(pose_rot, Z1);
    V3 A0 = RxT<1>(pose_rot, Y00), A1 = RxT<0>(pose_rot, Y01), A2 = RxT<0>(pose_rot, Y10);
    o.JT[0][0] = dot(jg0, A0); o.JT[0][1] = dot(jg0, A1); o.JT[0][2] = dot(jg0, A2);
    o.JT[1][0] = dot(jg1, A0); o.JT[1][1] = dot(jg1, A1); o.JT[1][2] = dot(jg1, A2);
}

__device__ __forceinline__ Pose rot_pose(const double *x, int layout, const Pose &pose)
{
    if (layout == VSLAM_LAYOUT_STRICT) return pose;
    Pose r = pose;
    sincos(x[3], &r.sp, &r.cp);
    sincos(x[4], &r.st, &r.ct);
    sincos(x[5], &r.ss, &r.cs);
    return r;
}

struct BundleArgs {
    const double *x;
    double *h, *SR, *C;
    int32_t *code;
    int n, L, P, layout;
};

__global__ __launch_bounds__(kThreads) void bundle_measure_kernel(CamK cam, BundleArgs a)
{
    const int tid = threadIdx.x, n = a.n, L = a.L, m = 2 * L;
    const bool is_cam = blockIdx.x == 0;
    const bool strict = a.layout == VSLAM_LAYOUT_STRICT;
    const int cam_c0 = strict ? 6 : 0, lm_c0 = strict ? 12 : 6;
    int jb, je;
    if (is_cam) { jb = 0; je = L; }
    else { jb = (blockIdx.x - 1) * a.P; je = min(L, jb + a.P); }

    if (!is_cam) {
        // One pass over this panel's columns, every 16-byte pair of rows written exactly once: the pair
        // (2j, 2j+1) of a column that belongs to landmark j carries the Jacobian entries (or 0.01 on the
        // diagonal of SR), every other pair is zero.  Consecutive lanes write consecutive pairs, so a wave
        // emits whole cache lines and no line is touched twice.
        __shared__ double JPs[kThreads][6];
        const int np = je - jb;
        if (tid < np) {
            const Pose pose = make_pose(a.x + 6);
            const Pose prot = rot_pose(a.x, a.layout, pose);
            Jac J;
            landmark_jacobian(cam, pose, prot, a.x + 12 + 3 * (jb + tid), J);
#pragma unroll
            for (int c = 0; c < 3; ++c) { JPs[tid][2 * c] = J.JP[0][c]; JPs[tid][2 * c + 1] = J.JP[1][c]; }
        }
        __syncthreads();
        if (a.C) {
            double *base = a.C + (size_t)(lm_c0 + 3 * jb) * m;
            const int pairs = 3 * np * L;                          // m / 2 = L pairs per column
            const bool aligned = (reinterpret_cast<uintptr_t>(base) & 15) == 0;
            for (int i = tid; i < pairs; i += kThreads) {
                const int col = i / L, rr = i - col * L, jl = col / 3, c = col - 3 * jl;
                double2 v = make_double2(0.0, 0.0);
                if (rr == jb + jl) v = make_double2(JPs[jl][2 * c], JPs[jl][2 * c + 1]);
                if (aligned) reinterpret_cast<double2 *>(base)[i] = v;
                else { base[2 * (size_t)i] = v.x; base[2 * (size_t)i + 1] = v.y; }
            }
        }
        if (a.SR) {
            double *base = a.SR + (size_t)(2 * jb) * m;
            const int pairs = 2 * np * L;
            const bool aligned = (reinterpret_cast<uintptr_t>(base) & 15) == 0;
            for (int i = tid; i < pairs; i += kThreads) {
                const int col = i / L, rr = i - col * L;
                double2 v = make_double2(0.0, 0.0);
                if (rr == jb + (col >> 1)) v = (col & 1) ? make_double2(0.0, 0.01) : make_double2(0.01, 0.0);
                if (aligned) reinterpret_cast<double2 *>(base)[i] = v;
                else { base[2 * (size_t)i] = v.x; base[2 * (size_t)i + 1] = v.y; }
            }
        }
        return;
    }
    if (a.C) block_zero_cols_f64(a.C, m, m, strict ? 0 : n - 6, strict ? 6 : n);
    if (jb + tid >= je) return;
    const Pose pose = make_pose(a.x + 6);
    const Pose prot = rot_pose(a.x, a.layout, pose);
    for (int j = jb + tid; j < je; j += kThreads) {
        Jac J;
        landmark_jacobian(cam, pose, prot, a.x + 12 + 3 * j, J);
        a.h[2 * j] = J.px[0]; a.h[2 * j + 1] = J.px[1];
        if (a.code) a.code[j] = J.code;
        if (a.C) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double2 *colP = reinterpret_cast<double2 *>(a.C + (size_t)(cam_c0 + c) * m + 2 * j);
                double2 *colT = reinterpret_cast<double2 *>(a.C + (size_t)(cam_c0 + 3 + c) * m + 2 * j);
                *colP = make_double2(-J.JP[0][c], -J.JP[1][c]);
                *colT = make_double2(J.JT[0][c], J.JT[1][c]);
            }
        }
    }
}

__global__ void project_kernel(CamK cam, const double *x, int j0, int L, double *px, int32_t *code)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L) return;
    const Pose pose = make_pose(x + 6);
    double p[2];
    int c = project_landmark(cam, pose, x + 12 + 3 * (j0 + j), p);
    px[2 * j] = p[0]; px[2 * j + 1] = p[1];
    code[j] = c;
}

// One landmark, 2 x n Jacobian (single workgroup).
__global__ __launch_bounds__(kThreads) void single_measure_kernel(CamK cam, const double *x, int n, int j, int layout,
                                                                  double *h, double *SR, double *J, int32_t *code)
{
    if (J) block_zero_f64(J, (size_t)2 * n);
    __syncthreads();
    if (threadIdx.x != 0) return;
    const Pose pose = make_pose(x + 6);
    const Pose prot = rot_pose(x, layout, pose);
    Jac o;
    landmark_jacobian(cam, pose, prot, x + 12 + 3 * j, o);
    h[0] = o.px[0]; h[1] = o.px[1];
    code[0] = o.code;
    if (SR) { SR[0] = 0.01; SR[1] = 0.0; SR[2] = 0.0; SR[3] = 0.01; }
    if (J) {
        const bool strict = layout == VSLAM_LAYOUT_STRICT;
        const int c0 = strict ? 6 : 0, c2 = (strict ? 12 : 6) + 3 * j;
        for (int c = 0; c < 3; ++c)
            for (int i = 0; i < 2; ++i) {
                J[i + 2 * (c0 + c)] = -o.JP[i][c];
                J[i + 2 * (c0 + 3 + c)] = o.JT[i][c];
                J[i + 2 * (c2 + c)] = o.JP[i][c];
            }
    }
}

// Device-side launcher shared with the fused affine transform (gaussian.hip).
vslam_status launch_bundle_measure(vslam_ctx *ctx, const CamK &cam, const double *x, int n, int layout,
                                   double *h, double *SR, double *C, int32_t *code)
{
    if (n < 12 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "bundle_measure: n=%d is not 12+3L", n);
    if (layout != VSLAM_LAYOUT_COMPAT && layout != VSLAM_LAYOUT_STRICT) return fail(ctx, VSLAM_ERR_ARG, "bundle_measure: bad layout %d", layout);
    if (n == 12) return VSLAM_OK;        // a map without landmarks: the bundle's loop (src/measurementPointLandmark.cpp:61) is empty, h / SR / C have no entries
    BundleArgs a;
    a.x = x; a.h = h; a.SR = SR; a.C = C; a.code = code; a.n = n; a.L = (n - 12) / 3; a.layout = layout;
    int P = (a.L + 2047) / 2048;
    a.P = P < 1 ? 1 : (P > kThreads ? kThreads : P);
    int panels = (C || SR) ? (a.L + a.P - 1) / a.P : 0;
    hipLaunchKernelGGL(bundle_measure_kernel, dim3(1 + panels), dim3(kThreads), 0, ctx->stream, cam, a);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_project(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n,
                                      int j0, int L, double *px, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!x || !px || !code) return fail(ctx, VSLAM_ERR_ARG, "project: null pointer");
    if (n < 15 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "project: n=%d is not 12+3L with L>=1", n);
    if (j0 < 0 || L < 1 || j0 + L > (n - 12) / 3) return fail(ctx, VSLAM_ERR_SHAPE, "project: landmark range [%d,%d) outside the state", j0, j0 + L);
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(n * 8) + Arena::pad(2 * L * 8) + Arena::pad(L * 4)));
    Arena ar(ctx->ws);
    double *dx = ar.take<double>(n), *dpx = ar.take<double>(2 * L);
    int32_t *dc = ar.take<int32_t>(L);
    hipStream_t s = ctx->stream;
    VSLAM_TRY(h2d(ctx, dx, x, n * 8));
    hipLaunchKernelGGL(project_kernel, dim3((L + 63) / 64), dim3(64), 0, s, k, dx, j0, L, dpx, dc);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(d2h(ctx, px, dpx, 2 * L * 8));
    VSLAM_TRY(d2h(ctx, code, dc, L * 4));
    VSLAM_TRY(stream_sync(ctx));
    return VSLAM_OK;
}

extern "C" vslam_status vslam_bundle_measure(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n,
                                             int layout, double *h, double *SR, double *C, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (n < 12 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "bundle_measure: n=%d is not 12+3L", n);
    if (!x || (n > 12 && !h)) return fail(ctx, VSLAM_ERR_ARG, "bundle_measure: null pointer");
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int L = (n - 12) / 3, m = 2 * L;
    size_t bytes = Arena::pad(n * 8) + Arena::pad(m * 8) + Arena::pad(L * 4) + (C ? Arena::pad((size_t)m * n * 8) : 0) +
                   (SR ? Arena::pad((size_t)m * m * 8) : 0);
    VSLAM_TRY(ws_reserve(ctx, bytes));
    Arena ar(ctx->ws);
    double *dx = ar.take<double>(n), *dh = ar.take<double>(m);
    int32_t *dc = ar.take<int32_t>(L);
    double *dC = C ? ar.take<double>((size_t)m * n) : nullptr;
    double *dSR = SR ? ar.take<double>((size_t)m * m) : nullptr;
    VSLAM_TRY(h2d(ctx, dx, x, n * 8));
    VSLAM_TRY(launch_bundle_measure(ctx, k, dx, n, layout, dh, dSR, dC, dc));
    VSLAM_TRY(d2h(ctx, h, dh, m * 8));
    if (code) VSLAM_TRY(d2h(ctx, code, dc, L * 4));
    if (C) VSLAM_TRY(d2h(ctx, C, dC, (size_t)m * n * 8));
    if (SR) VSLAM_TRY(d2h(ctx, SR, dSR, (size_t)m * m * 8));
    VSLAM_TRY(stream_sync(ctx));
    return VSLAM_OK;
}

extern "C" vslam_status vslam_bundle_measure_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n, int layout,
                                                 double *h, double *SR, double *C, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!x || (n > 12 && !h)) return fail(ctx, VSLAM_ERR_ARG, "bundle_measure_dev: null pointer");
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_bundle_measure(ctx, k, x, n, layout, h, SR, C, code);
}

extern "C" vslam_status vslam_single_measure(vslam_ctx *ctx, const vslam_camera *cam, int j, const double *x, int n,
                                             int layout, double *h, double *SR, double *J, int32_t *code)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!x || !h || !code) return fail(ctx, VSLAM_ERR_ARG, "single_measure: null pointer");
    if (n < 15 || (n - 12) % 3 != 0) return fail(ctx, VSLAM_ERR_SHAPE, "single_measure: n=%d is not 12+3L with L>=1", n);
    if (j < 0 || j >= (n - 12) / 3) return fail(ctx, VSLAM_ERR_SHAPE, "single_measure: landmark %d outside the state", j);
    if (layout != VSLAM_LAYOUT_COMPAT && layout != VSLAM_LAYOUT_STRICT) return fail(ctx, VSLAM_ERR_ARG, "single_measure: bad layout %d", layout);
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(n * 8) + Arena::pad(2 * n * 8) + 3 * Arena::pad(64)));
    Arena ar(ctx->ws);
    double *dx = ar.take<double>(n), *dJ = ar.take<double>(2 * n), *dh = ar.take<double>(2), *dSR = ar.take<double>(4);
    int32_t *dc = ar.take<int32_t>(1);
    hipStream_t s = ctx->stream;
    VSLAM_TRY(h2d(ctx, dx, x, n * 8));
    hipLaunchKernelGGL(single_measure_kernel, dim3(1), dim3(kThreads), 0, s, k, dx, n, j, layout, dh, SR ? dSR : nullptr,
                       J ? dJ : nullptr, dc);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(d2h(ctx, h, dh, 16));
    VSLAM_TRY(d2h(ctx, code, dc, 4));
    if (SR) VSLAM_TRY(d2h(ctx, SR, dSR, 32));
    if (J) VSLAM_TRY(d2h(ctx, J, dJ, 2 * n * 8));
    VSLAM_TRY(stream_sync(ctx));
    return VSLAM_OK;
}
