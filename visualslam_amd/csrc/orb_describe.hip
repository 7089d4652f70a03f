// Orientation and rBRIEF descriptors of ORB keypoints: the compute() half of cv::ORB as detectAndDrawORB calls
// it (reference src/imagefeatures.cpp:88-96); the descriptors are what BFMatcher(NORM_HAMMING) consumes at
// src/SLAM.cpp:222,355.  OpenCV's sources are not part of the reference, so the published algorithm (Rublee et
// al., "ORB: an efficient alternative to SIFT or SURF", 2011) is restated with OpenCV's documented conventions and
// the CPU oracle defines the contract; parity unpinned.
//
//   blur         7 x 7 Gaussian (sigma 2, reflect-101 borders) of a level image in 8.8 fixed point: weights
//                {18, 34, 49, 54, 49, 34, 18} / 256 per axis, the row pass exact, the column pass (sum + 2^15) >> 16.
//   orientation  intensity centroid over the circular patch of radius 15 on the level image (integer moments),
//                angle in degrees by the odd 7th-order polynomial of cv::fastAtan2.
//   descriptor   256 tests I(p0) < I(p1) on the blurred level image with the pattern rotated by the angle;
//                coordinates rounded half to even; bit k of byte i is test 8 i + k.
//
// One wavefront per keypoint: two lanes per patch row accumulate the moments, a wave reduction gives the angle in
// every lane, then lane l evaluates tests 4 l .. 4 l + 3 and pairs of lanes assemble the bytes.  The work is a few
// hundred scattered byte reads per keypoint out of an image that sits in L2; 10,000 keypoints fill the chip.
#include "common.hpp"
#include "orb.hpp"

#include <cmath>

namespace vslam {

namespace {

constexpr int kReach = 23;      // pattern coordinates are limited to |c| <= 15: rotated and rounded they stay within 22

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// 64 x 16 output pixels per workgroup: the 70 x 22 source pixels (reflected at the image sides) go to LDS once, the row
// pass writes 22 x 64 sixteen-bit sums (at most 255 * 256) back to LDS, the column pass reads seven of them per output.
constexpr int G7_W = 64, G7_H = 16;
__global__ __launch_bounds__(256) void gauss7_kernel(const uint8_t *src, int w, int h, int sstride, size_t sframe, uint8_t *dst, int dstride,
                                                     size_t dframe)
{
    __shared__ uint8_t px[(G7_H + 6) * (G7_W + 8)];
    __shared__ uint16_t hs[(G7_H + 6) * G7_W];
    const int tid = threadIdx.x, x0 = blockIdx.x * G7_W, y0 = blockIdx.y * G7_H;
    src += (size_t)blockIdx.z * sframe;
    dst += (size_t)blockIdx.z * dframe;
    for (int i = tid; i < (G7_H + 6) * (G7_W + 6); i += 256) {
        const int r = i / (G7_W + 6), c = i - r * (G7_W + 6);
        px[r * (G7_W + 8) + c] = src[(size_t)reflect101(min(y0 + r - 3, h + 2), h) * sstride + reflect101(min(x0 + c - 3, w + 2), w)];
    }
    __syncthreads();
    const unsigned wt[7] = {18u, 34u, 49u, 54u, 49u, 34u, 18u};
    for (int i = tid; i < (G7_H + 6) * G7_W; i += 256) {
        const int r = i / G7_W, c = i - r * G7_W;
        const uint8_t *p = px + r * (G7_W + 8) + c;
        unsigned sum = 0;
#pragma unroll
        for (int t = 0; t < 7; ++t) sum += wt[t] * p[t];
        hs[i] = (uint16_t)sum;
    }
    __syncthreads();
    for (int i = tid; i < G7_H * G7_W; i += 256) {
        const int r = i / G7_W, c = i - r * G7_W, x = x0 + c, y = y0 + r;
        if (x >= w || y >= h) continue;
        unsigned acc = 0;
#pragma unroll
        for (int t = 0; t < 7; ++t) acc += wt[t] * hs[(r + t) * G7_W + c];
        dst[(size_t)y * dstride + x] = (uint8_t)((acc + 32768u) >> 16);
    }
}

// The same blur for images whose rows start on 4-byte boundaries (source and destination): whole words in and out, four
// adjacent pixels per thread.  128 x 32 output pixels per workgroup; the row pass takes the four sums of a pixel quad
// from three LDS words with byte-aligning shifts and two 4-way byte dot products per pixel (weights packed as bytes),
// the column pass reads seven 16-byte rows of sums per quad.  Integer arithmetic: the same values as the kernel above.
constexpr int G7W_W = 128, G7W_H = 32, G7W_PW = G7W_W / 4 + 2;        // words per staged source row (columns x0-4 .. x0+131)
__global__ __launch_bounds__(256) void gauss7_words_kernel(const uint8_t *src, int w, int h, int sstride, size_t sframe, uint8_t *dst, int dstride,
                                                           size_t dframe)
{
    __shared__ uint32_t px[(G7W_H + 6) * (G7W_PW + 1)];
    __shared__ __align__(16) uint32_t hs[(G7W_H + 6) * G7W_W];
    const int tid = threadIdx.x, x0 = blockIdx.x * G7W_W, y0 = blockIdx.y * G7W_H;
    src += (size_t)blockIdx.z * sframe;
    dst += (size_t)blockIdx.z * dframe;
    for (int i = tid; i < (G7W_H + 6) * G7W_PW; i += 256) {
        const int r = i / G7W_PW, k = i - r * G7W_PW, gx = x0 - 4 + 4 * k;
        const uint8_t *row = src + (size_t)reflect101(min(y0 + r - 3, h + 2), h) * sstride;
        uint32_t v;
        if (gx >= 0 && gx + 4 <= w) v = *reinterpret_cast<const uint32_t *>(row + gx);
        else {
            v = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) v |= (uint32_t)row[reflect101(min(gx + q, w + 2), w)] << (8 * q);
        }
        px[r * (G7W_PW + 1) + k] = v;
    }
    __syncthreads();
    constexpr uint32_t W_LO = 18u | (34u << 8) | (49u << 16) | (54u << 24), W_HI = 49u | (34u << 8) | (18u << 16);
    for (int i = tid; i < (G7W_H + 6) * (G7W_W / 4); i += 256) {
        const int r = i / (G7W_W / 4), j = i - r * (G7W_W / 4);
        const uint32_t *p = px + r * (G7W_PW + 1) + j;
        const uint32_t a = p[0], b = p[1], c = p[2];          // columns x0+4j-4 .. x0+4j+7
        uint4 o;
        o.x = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 1), W_LO, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 1), W_HI, 0u, false), false);
        o.y = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 2), W_LO, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 2), W_HI, 0u, false), false);
        o.z = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 3), W_LO, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, b, 3), W_HI, 0u, false), false);
        o.w = __builtin_amdgcn_udot4(b, W_LO, __builtin_amdgcn_udot4(c, W_HI, 0u, false), false);
        *reinterpret_cast<uint4 *>(hs + r * G7W_W + 4 * j) = o;
    }
    __syncthreads();
    const unsigned wt[7] = {18u, 34u, 49u, 54u, 49u, 34u, 18u};
    for (int i = tid; i < G7W_H * (G7W_W / 4); i += 256) {
        const int r = i / (G7W_W / 4), j = i - r * (G7W_W / 4), x = x0 + 4 * j, y = y0 + r;
        if (x >= w || y >= h) continue;
        unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const uint4 v = *reinterpret_cast<const uint4 *>(hs + (r + t) * G7W_W + 4 * j);
            a0 += wt[t] * v.x; a1 += wt[t] * v.y; a2 += wt[t] * v.z; a3 += wt[t] * v.w;
        }
        const uint32_t o = ((a0 + 32768u) >> 16) | (((a1 + 32768u) >> 16) << 8) | (((a2 + 32768u) >> 16) << 16) | (((a3 + 32768u) >> 16) << 24);
        uint8_t *d = dst + (size_t)y * dstride + x;
        if (x + 4 <= w) *reinterpret_cast<uint32_t *>(d) = o;
        else for (int q = 0; q < w - x; ++q) d[q] = (uint8_t)(o >> (8 * q));
    }
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    const bool flat = ax >= ay;
    const float c = flat ? __fdiv_rn(ay, __fadd_rn(ax, 2.220446049250313e-16f)) : __fdiv_rn(ax, __fadd_rn(ay, 2.220446049250313e-16f));
    const float c2 = __fmul_rn(c, c);
    float r = __fadd_rn(__fmul_rn(p7, c2), p5);
    r = __fadd_rn(__fmul_rn(r, c2), p3);
    r = __fadd_rn(__fmul_rn(r, c2), p1);
    r = __fmul_rn(r, c);
    float a = flat ? r : __fsub_rn(90.f, r);
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

struct DescArgs {
    const uint8_t *gray, *blur;
    size_t gframe, bframe;                    // frame f of a batch: gray + f * gframe, blur + f * bframe
    int w, h, stride, bstride;
    const vslam_feature *kp;
    const int32_t *count;
    int max_count;
    const int8_t *pattern;
    float *angle;
    uint8_t *desc;
    unsigned long long umax_lo, umax_hi;      // half-widths of patch rows 0..7 and 8..15, a byte each
};

__global__ __launch_bounds__(256) void orb_describe_kernel(DescArgs a)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6), f = blockIdx.y;
    const int n = min(a.count[f], a.max_count);
    if (i >= n) return;
    a.kp += (size_t)f * a.max_count; a.angle += (size_t)f * a.max_count; a.desc += (size_t)f * a.max_count * 32;
    a.gray += (size_t)f * a.gframe; a.blur += (size_t)f * a.bframe;
    const int x = a.kp[i].x, y = a.kp[i].y;
    if (x < kReach || y < kReach || x >= a.w - kReach || y >= a.h - kReach) {      // patch would leave the image: no orientation, zero descriptor
        if (lane == 0) a.angle[i] = -1.0f;
        if (lane < 32) a.desc[(size_t)i * 32 + lane] = 0;
        return;
    }
    int m10 = 0, m01 = 0;
    if (lane < 62) {
        const int v = (lane >> 1) - 15, av = v < 0 ? -v : v;
        const int d = (int)(((av < 8 ? a.umax_lo : a.umax_hi) >> (8 * (av & 7))) & 0xffull);
        const uint8_t *row = a.gray + (size_t)(y + v) * a.stride + x;
        const int u0 = (lane & 1) ? 0 : -d, u1 = (lane & 1) ? d : -1;
        int s0 = 0, s1 = 0;
        for (int u = u0; u <= u1; ++u) { const int p = row[u]; s0 += p; s1 += u * p; }
        m10 = s1; m01 = v * s0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o, 64); m01 += __shfl_xor(m01, o, 64); }
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    if (lane == 0) a.angle[i] = angle;
    const float rad = __fmul_rn(angle, (float)(3.14159265358979323846 / 180.0));
    const float ca = (float)cos((double)rad), sa = (float)sin((double)rad);
    const uint8_t *center = a.blur + (size_t)y * a.bstride + x;
    const int4 raw = reinterpret_cast<const int4 *>(a.pattern)[lane];      // tests 4 lane .. 4 lane + 3: 16 signed bytes
    const int words[4] = {raw.x, raw.y, raw.z, raw.w};
    int nib = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int val[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float px = (float)(int)(signed char)(words[t] >> (16 * q)), py = (float)(int)(signed char)(words[t] >> (16 * q + 8));
            const float fx = __fsub_rn(__fmul_rn(px, ca), __fmul_rn(py, sa)), fy = __fadd_rn(__fmul_rn(px, sa), __fmul_rn(py, ca));
            const int ix = min(max((int)rintf(fx), -(kReach - 1)), kReach - 1), iy = min(max((int)rintf(fy), -(kReach - 1)), kReach - 1);
            val[q] = center[(ptrdiff_t)iy * a.bstride + ix];
        }
        nib |= (val[0] < val[1]) << t;
    }
    const int hi = __shfl_down(nib, 1, 64);
    if (!(lane & 1)) a.desc[(size_t)i * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
}

// The same for sixteen keypoints per workgroup, one wavefront each.  The wavefront-per-keypoint kernel above spends half of
// its instructions on the arctangent and the double-precision sine and cosine, which every wavefront evaluates for its one
// keypoint; here the sixteen wavefronts leave their moments in LDS, sixteen lanes of the first wavefront do that
// arithmetic for all of them at once, and every wavefront fetches its angle, cosine and sine back.  Same operations per
// keypoint in the same order, so the same bits.  On images whose rows start on 4-byte boundaries the moments of a patch
// row come from at most five aligned words per lane (bytes outside the row's chord masked off, sums by v_dot4_u32_u8)
// instead of up to sixteen byte reads.  (Staging the 45 x 45 neighbourhood of the blurred image in LDS for the tests
// was measured as well: 118 us per level against 107 us with the bytes read in place.)
constexpr int kDescKp = 16;
__global__ __launch_bounds__(64 * kDescKp) void orb_describe16_kernel(DescArgs a, int words)
{
    __shared__ int mom[kDescKp][2];
    __shared__ float trig[kDescKp][3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = blockIdx.x * kDescKp + wv, f = blockIdx.y;
    const int n = min(a.count[f], a.max_count);
    a.kp += (size_t)f * a.max_count; a.angle += (size_t)f * a.max_count; a.desc += (size_t)f * a.max_count * 32;
    a.gray += (size_t)f * a.gframe; a.blur += (size_t)f * a.bframe;
    const bool live = i < n;
    int x = 0, y = 0;
    if (live) { x = a.kp[i].x; y = a.kp[i].y; }
    const bool inside = live && !(x < kReach || y < kReach || x >= a.w - kReach || y >= a.h - kReach);
    int m10 = 0, m01 = 0;
    if (inside && lane < 62) {
        const int v = (lane >> 1) - 15, av = v < 0 ? -v : v;
        const int d = (int)(((av < 8 ? a.umax_lo : a.umax_hi) >> (8 * (av & 7))) & 0xffull);
        const uint8_t *row = a.gray + (size_t)(y + v) * a.stride;
        const int u0 = (lane & 1) ? 0 : -d, u1 = (lane & 1) ? d : -1;
        int s0 = 0, s1 = 0;
        if (words) {
            const int c0 = x + u0, c1 = x + u1, A = c0 & ~3;                 // columns c0 .. c1 of the row (none when c1 < c0)
            const uint32_t *wp = reinterpret_cast<const uint32_t *>(row + A);
            uint32_t w[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) w[q] = (A + 4 * q <= c1) ? wp[q] : 0u;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int first = A + 4 * q, lo = max(c0 - first, 0), hi = min(c1 - first, 3);
                const uint32_t mask = hi >= lo ? (0xffffffffu << (8 * lo)) & (0xffffffffu >> (8 * (3 - hi))) : 0u;
                const uint32_t wq = w[q] & mask;
                const int sp = (int)__builtin_amdgcn_udot4(wq, 0x01010101u, 0u, false);
                s0 += sp;
                s1 += (first - x) * sp + (int)__builtin_amdgcn_udot4(wq, 0x03020100u, 0u, false);
            }
        } else {
            for (int u = u0; u <= u1; ++u) { const int p = row[x + u]; s0 += p; s1 += u * p; }
        }
        m10 = s1; m01 = v * s0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o, 64); m01 += __shfl_xor(m01, o, 64); }
    if (lane == 0) { mom[wv][0] = m10; mom[wv][1] = m01; }
    __syncthreads();
    if (threadIdx.x < kDescKp) {
        const float angle = fast_atan2_deg((float)mom[threadIdx.x][1], (float)mom[threadIdx.x][0]);
        const float rad = __fmul_rn(angle, (float)(3.14159265358979323846 / 180.0));
        trig[threadIdx.x][0] = angle; trig[threadIdx.x][1] = (float)cos((double)rad); trig[threadIdx.x][2] = (float)sin((double)rad);
    }
    __syncthreads();
    if (!live) return;
    if (!inside) {                                             // patch would leave the image: no orientation, zero descriptor
        if (lane == 0) a.angle[i] = -1.0f;
        if (lane < 32) a.desc[(size_t)i * 32 + lane] = 0;
        return;
    }
    const float ca = trig[wv][1], sa = trig[wv][2];
    if (lane == 0) a.angle[i] = trig[wv][0];
    const uint8_t *center = a.blur + (size_t)y * a.bstride + x;
    const int4 raw = reinterpret_cast<const int4 *>(a.pattern)[lane];      // tests 4 lane .. 4 lane + 3: 16 signed bytes
    const int wds[4] = {raw.x, raw.y, raw.z, raw.w};
    int nib = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int val[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float px = (float)(int)(signed char)(wds[t] >> (16 * q)), py = (float)(int)(signed char)(wds[t] >> (16 * q + 8));
            const float fx = __fsub_rn(__fmul_rn(px, ca), __fmul_rn(py, sa)), fy = __fadd_rn(__fmul_rn(px, sa), __fmul_rn(py, ca));
            const int ix = min(max((int)rintf(fx), -(kReach - 1)), kReach - 1), iy = min(max((int)rintf(fy), -(kReach - 1)), kReach - 1);
            val[q] = center[(ptrdiff_t)iy * a.bstride + ix];
        }
        nib |= (val[0] < val[1]) << t;
    }
    const int hi = __shfl_down(nib, 1, 64);
    if (!(lane & 1)) a.desc[(size_t)i * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
}

// u_max of cv::ORB for a patch of 31: half-width of patch row v, v = 0..15
void patch_half_widths(unsigned long long *lo, unsigned long long *hi)
{
    const int hp = 15;
    int umax[17] = {0};
    const int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = (int)std::nearbyint(std::sqrt((double)hp * hp - (double)v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
    *lo = 0; *hi = 0;
    for (int v = 0; v < 8; ++v) { *lo |= (unsigned long long)umax[v] << (8 * v); *hi |= (unsigned long long)umax[8 + v] << (8 * v); }
}

}  // namespace

vslam_status launch_gauss7(vslam_ctx *ctx, const uint8_t *src, int w, int h, int sstride, size_t sframe, uint8_t *dst, int dstride,
                           size_t dframe, int frames)
{
    const bool words = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)sstride | (uintptr_t)dstride |
                         (uintptr_t)sframe | (uintptr_t)dframe) & 3) == 0 && w >= 4;
    if (words)
        hipLaunchKernelGGL(gauss7_words_kernel, dim3((w + G7W_W - 1) / G7W_W, (h + G7W_H - 1) / G7W_H, frames), dim3(256), 0, ctx->stream, src, w, h,
                           sstride, sframe, dst, dstride, dframe);
    else
        hipLaunchKernelGGL(gauss7_kernel, dim3((w + G7_W - 1) / G7_W, (h + G7_H - 1) / G7_H, frames), dim3(256), 0, ctx->stream, src, w, h, sstride,
                           sframe, dst, dstride, dframe);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

vslam_status launch_orb_describe(vslam_ctx *ctx, const uint8_t *gray, int stride, size_t gframe, const uint8_t *blur, int bstride,
                                 size_t bframe, int w, int h, int frames, const vslam_feature *kp, const int32_t *count, int max_count,
                                 const int8_t *pattern, float *angle, uint8_t *desc)
{
    if (max_count < 1 || frames < 1) return VSLAM_OK;
    DescArgs a;
    a.gray = gray; a.blur = blur; a.gframe = gframe; a.bframe = bframe; a.w = w; a.h = h; a.stride = stride; a.bstride = bstride;
    a.kp = kp; a.count = count; a.max_count = max_count; a.pattern = pattern; a.angle = angle; a.desc = desc;
    patch_half_widths(&a.umax_lo, &a.umax_hi);
    if (getenv("VSLAM_ORB_DESCRIBE_WAVE")) {                   // A/B runs: a wavefront per keypoint does everything itself
        hipLaunchKernelGGL(orb_describe_kernel, dim3((max_count + 3) / 4, frames), dim3(256), 0, ctx->stream, a);
    } else {
        const int words = ((reinterpret_cast<uintptr_t>(gray) | (uintptr_t)stride | (uintptr_t)gframe) & 3) == 0;
        hipLaunchKernelGGL(orb_describe16_kernel, dim3((max_count + kDescKp - 1) / kDescKp, frames), dim3(64 * kDescKp), 0, ctx->stream, a, words);
    }
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

void orb_default_pattern(int8_t *pattern)
{
    uint32_t st = 0x34985739u;
    for (int i = 0; i < 256; ++i) {
        int c[4];
        do {
            for (int k = 0; k < 4; ++k) { st = st * 1664525u + 1013904223u; c[k] = (int)((st >> 16) % 27u) - 13; }
        } while (c[0] == c[2] && c[1] == c[3]);
        for (int k = 0; k < 4; ++k) pattern[4 * i + k] = (int8_t)c[k];
    }
}

bool orb_pattern_ok(const int8_t *pattern)
{
    for (int i = 0; i < VSLAM_ORB_PATTERN_BYTES; ++i)
        if (pattern[i] < -15 || pattern[i] > 15) return false;
    return true;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_orb_default_pattern(int8_t *pattern)
{
    if (!pattern) return VSLAM_ERR_ARG;
    orb_default_pattern(pattern);
    return VSLAM_OK;
}

extern "C" vslam_status vslam_gaussian_blur7_dev(vslam_ctx *ctx, const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!src || !dst) return fail(ctx, VSLAM_ERR_ARG, "gaussian_blur7: null pointer");
    if (w < 1 || h < 1 || sstride < w || dstride < w || w > 65535 || h > 65535)
        return fail(ctx, VSLAM_ERR_SHAPE, "gaussian_blur7: %dx%d (strides %d, %d)", w, h, sstride, dstride);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_gauss7(ctx, src, w, h, sstride, 0, dst, dstride, 0, 1);
}

extern "C" vslam_status vslam_orb_describe_level_dev(vslam_ctx *ctx, const uint8_t *gray, int stride, const uint8_t *blurred, int bstride,
                                                     int w, int h, const vslam_feature *keypoints, const int32_t *count, int max_count,
                                                     const int8_t *pattern, float *angle, uint8_t *desc)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !blurred || !keypoints || !count || !pattern || !angle || !desc) return fail(ctx, VSLAM_ERR_ARG, "orb_describe: null pointer");
    if (w < 1 || h < 1 || stride < w || bstride < w || max_count < 0) return fail(ctx, VSLAM_ERR_SHAPE, "orb_describe: %dx%d (strides %d, %d), %d keypoints", w, h, stride, bstride, max_count);
    if (reinterpret_cast<uintptr_t>(pattern) & 15) return fail(ctx, VSLAM_ERR_ARG, "orb_describe: the pattern must be 16-byte aligned");
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_orb_describe(ctx, gray, stride, 0, blurred, bstride, 0, w, h, 1, keypoints, count, max_count, pattern, angle, desc);
}
