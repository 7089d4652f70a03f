// Corner front end: Harris / Shi-Tomasi response, 3x3 non-maximum suppression,
// thresholding and score-sorted keypoint output, as LDS-tiled stencil kernels.
//
// Stands in for the feature stage of detectAndDrawORB (reference
// src/imagefeatures.cpp:71-96, cv::ORB with HARRIS_SCORE) as consumed by
// src/SLAM.cpp:449-472 (sort by response, threshold, pick).  The arithmetic
// contract (the reference delegates it to OpenCV):
//   Ix, Iy   3x3 Sobel on the u8 image, exact int32
//   a, b, c  block_size^2 window sums of Ix^2, Iy^2, Ix*Iy, exact int32
//   s = 1/(4*block_size*255), s2 = s*s, s4 = s2*s2                (float32)
//   Harris   ((fa*fb - fc*fc) - k*((fa+fb)*(fa+fb))) * s4         (float32, one
//            rounding per operation, no fused multiply-add)
//   Shi-T.   ((fa+fb)*0.5f - sqrtf(d*d + fc*fc)) * s2, d=(fa-fb)*0.5f
//   valid    pixels >= 1+block_size/2 from every edge, 0 elsewhere
//   NMS      strictly greater than the 4 raster-preceding neighbours,
//            greater-or-equal than the 4 following ones
//   output   response > threshold, inside the border; sorted by score
//            descending, then y, then x ascending; first max_features.
//
// Kernel 1 (corner_tile_kernel): one workgroup per 64x16 output tile.  The u8
// tile plus a (block/2+2)-pixel halo is read once from HBM with aligned 4-byte
// row loads into LDS; gradients -> products -> separable running window sums
// -> response -> NMS all stay in LDS.  Survivors are staged in LDS and appended
// with one atomic reservation per workgroup to a per-frame candidate list of
// 64-bit sort keys that encode score, y and x.
// Kernel 2 (sort_chunk_kernel): bitonic sort of parts of 2048 keys in LDS, one
// 256-thread workgroup per part; merge_parts_kernel joins the parts of a chunk
// of 16384 keys in LDS; merge_chunks_kernel merges the chunks of a longer list
// keeping the top 16384 (enough for any max_features <= 16384).  The last kernel
// unpacks the leading keys into Feature records.
#include "common.hpp"
#include <type_traits>
#include "orb.hpp"

#include <cmath>
#include <vector>
#include <cstdlib>

namespace vslam {

constexpr int TW = 64, TH = 16;            // output tile: 40 KB of LDS, four workgroups per CU (measured: 64x32 -40%, 64x8 -23%)
constexpr int MAXHB = 7;                   // block_size <= 15
constexpr int SORT_S = 16384;              // keys per LDS sort
constexpr int CAND_CHUNKS = 4;             // candidate capacity = 4 * SORT_S per frame and detector
constexpr int CAND_CAP = CAND_CHUNKS * SORT_S;

__device__ __forceinline__ uint64_t make_key(float score, int x, int y)
{
    uint32_t u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // order-preserving float -> uint
    uint32_t lo = ~(((uint32_t)y << 16) | (uint32_t)x);      // smaller (y,x) sorts first
    return ((uint64_t)u << 32) | lo;
}
__device__ __forceinline__ void unpack_key(uint64_t key, vslam_feature *f)
{
    uint32_t u = (uint32_t)(key >> 32), lo = ~(uint32_t)key;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    f->score = __uint_as_float(u);
    f->x = (int32_t)(lo & 0xffffu);
    f->y = (int32_t)(lo >> 16);
}

struct CornerArgs {
    const uint8_t *gray; size_t stride_frame;
    int w, h, stride;
    int bs, border;
    float k, thr_h, thr_s;
    float s2, s4;
    uint64_t *cand;        // [frame][det][CAND_CAP]
    int32_t *cand_count;   // [frame][det]
    float *map_h, *map_s;  // dense maps (MAPS mode), single frame
    int tiles_x, frames;
};

// DET: bit0 Harris, bit1 Shi-Tomasi.  MAPS: write dense response maps instead of keypoints.
// grid.x = tiles * frames with the frame index fastest, so the workgroups in
// flight at any moment belong to many frames and their (one per workgroup and
// detector) candidate-count atomics spread over many counters.
// BS: window size fixed at compile time (7, cv::ORB's Harris block) so that every thread's LDS reads of a stage are
// issued together and the short loops are unrolled; BS = 0 takes the window size from the arguments.
template <int DET, bool MAPS, int BS>
__global__ __launch_bounds__(256) void corner_tile_kernel(CornerArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int lcnt[2], gbase[2];
    const int wbs = BS > 0 ? BS : a.bs;
    const int hb = wbs >> 1, R = hb + 1;           // response needs image +-R
    const int halo = R + 1;                        // +1 for the NMS ring
    const int IW = TW + 2 * halo, IH = TH + 2 * halo;      // image tile
    const int GW = TW + 2 * (hb + 1), GH = TH + 2 * (hb + 1);  // gradient-product tile (response ring +-hb)
    const int RW = TW + 2, RH = TH + 2;            // response tile
    const int IP = IW | 1, GP = GW | 1, RP = RW | 1;   // odd LDS row pitches: rows land in different banks
    // LDS carve-up (int32 everywhere: exact integer structure tensor)
    int32_t *pa = reinterpret_cast<int32_t *>(smem);               // GW*GH   Ix^2   (later: Harris map)
    int32_t *pb = pa + GP * GH;                                    //         Iy^2   (later: Shi-Tomasi map)
    int32_t *pc = pb + GP * GH;                                    //         IxIy   (later: staged candidates)
    int32_t *ha = pc + GP * GH;                                    // RW*GH   horizontal sums
    int32_t *hbb = ha + RP * GH;
    int32_t *hc = hbb + RP * GH;
    int32_t *img = ha;                                             // IW*IH unpacked pixels, dead before ha is written

    const int tid = threadIdx.x;
    const int frame = blockIdx.x % a.frames, tile = blockIdx.x / a.frames;
    const int x0 = (tile % a.tiles_x) * TW, y0 = (tile / a.tiles_x) * TH;
    const uint8_t *gray = a.gray + (size_t)frame * a.stride_frame;
    const int w = a.w, h = a.h;
    if (tid < 2) lcnt[tid] = 0;

    // ---- 1. image tile: aligned 4-byte loads, unpacked to int32 (clamped at the image edge;
    //         clamped pixels only feed outputs that are marked invalid) ----
    {
        const int gx0 = x0 - halo, gy0 = y0 - halo;
        const int ax0 = gx0 & ~3;
        // x0 is a multiple of 4, so the number of aligned words per row does not depend on the tile
        const int ND = BS > 0 ? (((TW + BS / 2 + 2 - 1) >> 2) - ((-(BS / 2 + 2)) >> 2) + 1) : ((gx0 + IW - 1) >> 2) - (ax0 >> 2) + 1;
        const bool fast_ok = ((a.stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0);
        // every thread's row words are requested before any is unpacked (at most MAXIT rounds of 256 items)
        constexpr int MAXIT = ((TW + 2 * (MAXHB + 2) + 6) / 4 + 1) * (TH + 2 * (MAXHB + 2)) / 256 + 1;
        uint32_t words[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int item = tid + it * 256;
            words[it] = 0;
            if (item < ND * IH) {
                const int r = item / ND, d = item - r * ND;
                const int gy = gy0 + r, gx = ax0 + 4 * d;
                const int cy = min(max(gy, 0), h - 1);
                if (fast_ok && gx >= 0 && gx + 3 < w) words[it] = *reinterpret_cast<const uint32_t *>(gray + (size_t)cy * a.stride + gx);
                else
                    for (int b = 0; b < 4; ++b) words[it] |= (uint32_t)gray[(size_t)cy * a.stride + min(max(gx + b, 0), w - 1)] << (8 * b);
            }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int item = tid + it * 256;
            if (item < ND * IH) {
                const int r = item / ND, d = item - r * ND;
                const int gx = ax0 + 4 * d;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int lx = gx + b - gx0;
                    if (lx >= 0 && lx < IW) img[__mul24(r, IP) + lx] = (int32_t)((words[it] >> (8 * b)) & 255u);
                }
            }
        }
    }
    __syncthreads();
    // ---- 2. Sobel gradients and their products ----
    if (BS == 7) {
        constexpr int NSEG = 9, SEG = 8;           // GW = 72 = 9 segments of 8 columns
        for (int item = tid; item < GH * NSEG; item += 256) {
            const int ly = item / NSEG, xb = (item - ly * NSEG) * SEG;
            const int32_t *r0 = img + __mul24(ly, IP) + xb;
            int v0[SEG + 2], v1[SEG + 2], v2[SEG + 2];
#pragma unroll
            for (int k = 0; k < SEG + 2; ++k) { v0[k] = r0[k]; v1[k] = r0[IP + k]; v2[k] = r0[2 * IP + k]; }
#pragma unroll
            for (int k = 0; k < SEG; ++k) {
                const int ix = (v0[k + 2] - v0[k]) + 2 * (v1[k + 2] - v1[k]) + (v2[k + 2] - v2[k]);
                const int iy = (v2[k] - v0[k]) + 2 * (v2[k + 1] - v0[k + 1]) + (v2[k + 2] - v0[k + 2]);
                const int p = __mul24(ly, GP) + xb + k;
                pa[p] = __mul24(ix, ix); pb[p] = __mul24(iy, iy); pc[p] = __mul24(ix, iy);   // |ix|, |iy| <= 1020
            }
        }
    } else {
        const int nseg = max(1, 256 / GH), seglen = (GW + nseg - 1) / nseg;   // ~one (row, segment) item per thread
        for (int item = tid; item < GH * nseg; item += 256) {
            const int ly = item / nseg, sgm = item - ly * nseg;
            const int xb = sgm * seglen, xe = min(GW, xb + seglen);
            if (xb >= xe) continue;
            const int32_t *r0 = img + ly * IP + xb, *r1 = r0 + IP, *r2 = r1 + IP;
            int a0 = r0[0], a1 = r0[1], b0 = r1[0], b1 = r1[1], c0 = r2[0], c1 = r2[1];
            for (int lx = xb; lx < xe; ++lx) {
                const int o = lx - xb + 2;
                const int a2 = r0[o], b2 = r1[o], c2 = r2[o];
                const int ix = (a2 - a0) + 2 * (b2 - b0) + (c2 - c0);
                const int iy = (c0 - a0) + 2 * (c1 - a1) + (c2 - a2);
                const int p = ly * GP + lx;
                pa[p] = ix * ix; pb[p] = iy * iy; pc[p] = ix * iy;
                a0 = a1; a1 = a2; b0 = b1; b1 = b2; c0 = c1; c1 = c2;
            }
        }
    }
    __syncthreads();
    // ---- 3. horizontal window sums (running sums; integers, so exact): RW x GH ----
    if (BS == 7) {
        constexpr int NSEG = 9, SEG = 8;           // RW = 66 <= 72
        for (int item = tid; item < GH * NSEG; item += 256) {
            const int ly = item / NSEG, xb = (item - ly * NSEG) * SEG;
            const int rowoff = __mul24(ly, GP);
            const int32_t *qa = pa + rowoff, *qb = pb + rowoff, *qc = pc + rowoff;
            int va[SEG + 6], vb[SEG + 6], vc[SEG + 6];
#pragma unroll
            for (int k = 0; k < SEG + 6; ++k) {
                const int x = min(xb + k, GW - 1);
                va[k] = qa[x]; vb[k] = qb[x]; vc[k] = qc[x];
            }
            int sa = 0, sb = 0, sc = 0;
#pragma unroll
            for (int t = 0; t < 7; ++t) { sa += va[t]; sb += vb[t]; sc += vc[t]; }
#pragma unroll
            for (int k = 0; k < SEG; ++k) {
                if (k > 0) { sa += va[k + 6] - va[k - 1]; sb += vb[k + 6] - vb[k - 1]; sc += vc[k + 6] - vc[k - 1]; }
                if (xb + k < RW) { const int o = __mul24(ly, RP) + xb + k; ha[o] = sa; hbb[o] = sb; hc[o] = sc; }
            }
        }
    } else {
        const int nseg = max(1, 256 / GH), seglen = (RW + nseg - 1) / nseg, bs = a.bs;
        for (int item = tid; item < GH * nseg; item += 256) {
            const int ly = item / nseg, sgm = item - ly * nseg;
            const int xb = sgm * seglen, xe = min(RW, xb + seglen);
            if (xb >= xe) continue;
            const int32_t *qa = pa + ly * GP + xb, *qb = pb + ly * GP + xb, *qc = pc + ly * GP + xb;
            int sa = 0, sb = 0, sc = 0;
            for (int t = 0; t < bs; ++t) { sa += qa[t]; sb += qb[t]; sc += qc[t]; }
            int o = ly * RP + xb;
            ha[o] = sa; hbb[o] = sb; hc[o] = sc;
            for (int lx = xb + 1; lx < xe; ++lx) {
                const int in = lx - xb + bs - 1, out = lx - xb - 1;
                sa += qa[in] - qa[out]; sb += qb[in] - qb[out]; sc += qc[in] - qc[out];
                ++o;
                ha[o] = sa; hbb[o] = sb; hc[o] = sc;
            }
        }
    }
    __syncthreads();
    // ---- 4. vertical running sums + response: RW x RH, stored over the product buffers ----
    float *resp_h = reinterpret_cast<float *>(pa), *resp_s = reinterpret_cast<float *>(pb);
    {
        const int vlen = (RH + 2) / 3, bs = wbs;
        for (int item = tid; item < RW * 3; item += 256) {
            const int sgm = item / RW, lx = item - sgm * RW;
            const int yb = sgm * vlen, ye = min(RH, yb + vlen);
            constexpr int VROWS = (TH + 2 + 2) / 3 + 6;            // rows one thread reads when BS == 7
            int ca[VROWS], cb[VROWS], cc[VROWS];
            int sa = 0, sb = 0, sc = 0;
            if (BS == 7) {
#pragma unroll
                for (int t = 0; t < VROWS; ++t) {
                    const int q = __mul24(min(yb + t, GH - 1), RP) + lx;
                    ca[t] = ha[q]; cb[t] = hbb[q]; cc[t] = hc[q];
                }
#pragma unroll
                for (int t = 0; t < 7; ++t) { sa += ca[t]; sb += cb[t]; sc += cc[t]; }
            } else {
                for (int t = 0; t < bs; ++t) { const int q = (yb + t) * RP + lx; sa += ha[q]; sb += hbb[q]; sc += hc[q]; }
            }
            const int gx = x0 - 1 + lx;
#pragma unroll 8
            for (int j = 0; j < (BS == 7 ? VROWS - 6 : 1 << 20); ++j) {
                const int ly = yb + j;
                if (ly >= ye) break;
                if (j > 0) {
                    if (BS == 7) { sa += ca[j + 6] - ca[j - 1]; sb += cb[j + 6] - cb[j - 1]; sc += cc[j + 6] - cc[j - 1]; }
                    else {
                        const int in = (ly + bs - 1) * RP + lx, out = (ly - 1) * RP + lx;
                        sa += ha[in] - ha[out]; sb += hbb[in] - hbb[out]; sc += hc[in] - hc[out];
                    }
                }
                const int gy = y0 - 1 + ly;
                const bool valid = gx >= R && gy >= R && gx < w - R && gy < h - R;
                const float fa = (float)sa, fb = (float)sb, fc = (float)sc;
                const float t2 = __fmul_rn(fc, fc), t4 = __fadd_rn(fa, fb);
                const int i = __mul24(ly, RP) + lx;
                if (DET & 1) {
                    const float t3 = __fsub_rn(__fmul_rn(fa, fb), t2);
                    const float t6 = __fmul_rn(a.k, __fmul_rn(t4, t4));
                    resp_h[i] = valid ? __fmul_rn(__fsub_rn(t3, t6), a.s4) : 0.0f;
                }
                if (DET & 2) {
                    const float m = __fmul_rn(t4, 0.5f), d = __fmul_rn(__fsub_rn(fa, fb), 0.5f);
                    // sqrtf is the correctly rounded square root under this build's flags (no fast-math); the
                    // exhaustive comparison with (float)sqrt((double)x) is vslam_debug_sqrtf_mismatches below
                    const float root = sqrtf(__fadd_rn(__fmul_rn(d, d), t2));
                    resp_s[i] = valid ? __fmul_rn(__fsub_rn(m, root), a.s2) : 0.0f;
                }
            }
        }
    }
    __syncthreads();
    // ---- 5. maps, or NMS + threshold; survivors staged in LDS ----
    uint64_t *stage = reinterpret_cast<uint64_t *>(pc);            // [2][TW*TH/4]: 3x3 NMS keeps <= 1 pixel per 2x2
    constexpr int STAGE_CAP = TW * TH / 4;
    for (int i = tid; i < TW * TH; i += 256) {
        const int ly = i >> 6, lx = i & 63;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= w || gy >= h) continue;
        const int c = __mul24(ly + 1, RP) + (lx + 1);
        if (MAPS) {
            if ((DET & 1) && a.map_h) a.map_h[(size_t)gy * w + gx] = resp_h[c];
            if ((DET & 2) && a.map_s) a.map_s[(size_t)gy * w + gx] = resp_s[c];
            continue;
        }
        if (gx < a.border || gy < a.border || gx >= w - a.border || gy >= h - a.border) continue;
#pragma unroll
        for (int det = 0; det < 2; ++det) {
            if (!(DET & (1 << det))) continue;
            const float *rp = det == 0 ? resp_h : resp_s;
            const float thr = det == 0 ? a.thr_h : a.thr_s;
            const float r = rp[c];
            if (!(r > thr)) continue;
            const bool ok = r > rp[c - RP - 1] && r > rp[c - RP] && r > rp[c - RP + 1] && r > rp[c - 1] &&
                            r >= rp[c + 1] && r >= rp[c + RP - 1] && r >= rp[c + RP] && r >= rp[c + RP + 1];
            if (!ok) continue;
            const int slot = atomicAdd(&lcnt[det], 1);
            if (slot < STAGE_CAP) stage[det * STAGE_CAP + slot] = make_key(r, gx, gy);
        }
    }
    if (MAPS) return;
    __syncthreads();
    // ---- 6. one global reservation per workgroup and detector, then a coalesced copy ----
    if (tid < 2 && (DET & (1 << tid)) && lcnt[tid] > 0)
        gbase[tid] = atomicAdd(a.cand_count + ((size_t)frame * 2 + tid), min(lcnt[tid], STAGE_CAP));
    __syncthreads();
#pragma unroll
    for (int det = 0; det < 2; ++det) {
        if (!(DET & (1 << det))) continue;
        const int cnt = min(lcnt[det], STAGE_CAP);
        uint64_t *dst = a.cand + ((size_t)frame * 2 + det) * CAND_CAP;
        for (int i = tid; i < cnt; i += 256) {
            const int pos = gbase[det] + i;
            if (pos < CAND_CAP) dst[pos] = stage[det * STAGE_CAP + i];
        }
    }
}

// ---- sliding-window detector (7x7 window, keypoints only) -------------------------------------
// One wavefront owns a 128-column strip (116 output columns + a 6-pixel halo either side) of a band
// of rows and walks down it one image row per step; lane l holds the column pair xs + 2l (E) and
// xs + 2l + 1 (O).  Nothing is recomputed vertically: the three newest pixel rows, a seven-row ring of
// horizontal window sums, the running vertical sums and the last response rows live in registers.
// Horizontal neighbours are the lane's other column or come from cross-lane reads (ds_bpermute) of
// neighbouring lanes -- a 7-wide window of a column pair is  odd(l-2) + pair(l-1) + pair(l) +
// pair(l+1) [+ even(l+2)] -- so there is no LDS tile and no barrier.  Step t takes pixel row t:
//   gradient row t-1 -> products -> horizontal 7-sums -> ring -> vertical sums = window of row t-4
//   -> responses of row t-4 -> non-maximum suppression of row t-5.
// Same integer and float operations per pixel as corner_tile_kernel, so the keypoints are identical.
constexpr int SW_HALO = 6, SW_OUT = 128 - 2 * SW_HALO;
typedef short sw_short2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_read_i(int v, int addr) { return __builtin_amdgcn_ds_bpermute(addr, v); }
__device__ __forceinline__ float lane_read_f(float v, int addr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }

// Maxima without the quieting of signalling NaNs that fmaxf asks of the compiler (a v_max_f32 x, x, x in front of every
// value that arrives from another lane): the responses compared here are finite.
__device__ __forceinline__ float max3_raw(float x, float y, float z)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ float max2_raw(float x, float y)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// Correctly rounded square root of x = 0 or a normal number far from both ends of the range (the sums of squares of the
// Shi-Tomasi response): the hardware estimate (within one unit in the last place) and the step to the neighbour whose
// residual says so, as the compiler's sqrtf does it -- without that routine's rescaling of tiny arguments and its test
// for zero and infinity (for x = 0 both residual tests fail and 0 stays).  Checked against sqrtf over every such x by
// vslam_debug_sqrtf_mismatches.
__device__ __forceinline__ float sqrt_rn_normal(float x)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const float yd = __int_as_float(__float_as_int(y) - 1), yu = __int_as_float(__float_as_int(y) + 1);
    const float rd = __builtin_fmaf(-yd, y, x), ru = __builtin_fmaf(-yu, y, x);
    float r = (rd <= 0.0f) ? yd : y;
    r = (ru > 0.0f) ? yu : r;
    return r;
}

template <int DET>
__global__ __launch_bounds__(64) void corner_strip_kernel(CornerArgs a, int nstrips, int nseg, int seg_h)
{
    __shared__ uint64_t stage[2][128];
    const int lane = threadIdx.x;
    // Workgroup ids are dealt round-robin to the eight XCDs, each with an L2 of its own.  Renumbered so that an XCD
    // works through one contiguous run of (frame, band, strip): neighbouring strips and bands, which share their halo
    // columns and rows, then meet in the same L2 instead of fetching them from HBM once per XCD.
    int bid;
    {
        const int nb = (int)gridDim.x, x = (int)blockIdx.x & 7, q = nb >> 3, r = nb & 7;
        bid = x * q + min(x, r) + ((int)blockIdx.x >> 3);
    }
    const int strip = bid % nstrips; bid /= nstrips;
    const int seg = bid % nseg, frame = bid / nseg;
    const int w = a.w, h = a.h;
    const int ya = seg * seg_h, yb = min(h, ya + seg_h);
    if (ya >= yb) return;
    const int gx0 = strip * SW_OUT - SW_HALO + 2 * lane;      // column E; O is gx0 + 1
    const uint8_t *gray = a.gray + (size_t)frame * a.stride_frame;
    // A lane fetches its column pair with one two-byte load (gx0 is even; the address need not be).  Pairs that stick out
    // of the image are pulled back inside: the columns outside then hold pixels of other columns, which only ever reach
    // responses that col_valid discards (a valid response reads the columns gx - 4 ... gx + 4, all inside) -- except that
    // with an odd width the last column, w - 1, is the E of a pair pulled back by one and sits in the second byte.
    const int cx = min(max(gx0, 0), w - 2);
    const unsigned pair_off = (unsigned)cx;
    const uint32_t pair_sel = (gx0 - cx == 1) ? 0x0c010c01u : 0x0c010c00u;     // v_perm: bytes -> {E, O} as two 16-bit values
    // byte addresses of the neighbouring lanes for ds_bpermute (wrapping; wrapped values only reach halo columns)
    const int am1 = ((lane - 1) & 63) << 2, ap1 = ((lane + 1) & 63) << 2, am2 = ((lane - 2) & 63) << 2, ap2 = ((lane + 2) & 63) << 2;
    constexpr int R = 4;                                      // response needs the image +-R (7x7 window of 3x3 gradients)
    bool col_valid[2], col_out[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int gx = gx0 + e, off = 2 * lane + e;
        col_valid[e] = gx >= R && gx < w - R;
        col_out[e] = off >= SW_HALO && off < 128 - SW_HALO && gx >= a.border && gx < w - a.border;
    }
    // per column threshold: +inf where the column is halo or outside the border, so it never yields a keypoint
    float thr_col[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) { thr_col[0][e] = col_out[e] ? a.thr_h : INFINITY; thr_col[1][e] = col_out[e] ? a.thr_s : INFINITY; }
    int cnt[2] = {0, 0};                                      // staged keys per detector (wave-uniform)

    uint32_t r0 = 0, r1 = 0, r2 = 0;                          // pixel rows t-2, t-1, t: {E, O} as two 16-bit values
    int ring[7][2][3];
#pragma unroll
    for (int u = 0; u < 7; ++u)
#pragma unroll
        for (int e = 0; e < 2; ++e) { ring[u][e][0] = 0; ring[u][e][1] = 0; ring[u][e][2] = 0; }
    int vs[2][3] = {{0, 0, 0}, {0, 0, 0}};                    // vertical window sums of a, b, c
    // per detector and column: centre-row value, 3-maximum of the row above the centre and of the centre row;
    // cl: left neighbour of E (O of lane l-1), cr: right neighbour of O (E of lane l+1), both of the centre row
    float cv[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, m_up[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, m_c[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float cl[2] = {0.f, 0.f}, cr[2] = {0.f, 0.f};

    // appends the keys of the lanes with `hit`; a full wave of staged keys goes out with one reservation
    auto emit = [&](int det, bool hit, float score, int x, int y) {
        const unsigned long long mask = __ballot(hit);
        if (!mask) return;
        const int pos = cnt[det] + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit) stage[det][pos] = make_key(score, x, y);
        cnt[det] += __popcll(mask);
        if (cnt[det] >= 64) {
            int base = 0;
            if (lane == 0) base = atomicAdd(a.cand_count + ((size_t)frame * 2 + det), 64);
            base = __shfl(base, 0, 64);
            uint64_t *dst = a.cand + ((size_t)frame * 2 + det) * CAND_CAP;
            const uint64_t key = stage[det][lane], rest = stage[det][64 + lane];
            if (base + lane < CAND_CAP) dst[base + lane] = key;
            stage[det][lane] = rest;
            cnt[det] -= 64;
        }
    };

    const int tstart = ya - 6, tend = yb + 5;                 // steps [tstart, tend), run in blocks of 7 (the ring size)
    auto fetch_pair = [&](int row) -> uint32_t {            // the row's address is wave-uniform, the column offset a 32-bit register
        const size_t ro = (size_t)min(max(row, 0), h - 1) * a.stride;
        const uint8_t *rp = gray + (((size_t)__builtin_amdgcn_readfirstlane((int)(ro >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)ro));
        uint16_t v;
        __builtin_memcpy(&v, rp + pair_off, 2);
        return v;
    };
    uint32_t nxt[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) nxt[u] = fetch_pair(tstart + u);
    for (int t0 = tstart; t0 < tend; t0 += 7) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {                          // steps past tend only touch rows no band emits
            const int t = t0 + u;
            const uint32_t pix = __builtin_amdgcn_perm(nxt[u], nxt[u], pair_sel);
            nxt[u] = fetch_pair(t + 7);                        // row t + 7 takes this slot: seven rows stay in flight
            // Sobel of row t-1, separable (smooth and difference), both columns of the pair in one register: the sums stay
            // below 2^16 (4 * 255), the differences are 16-bit two's complement
            r0 = r1; r1 = r2; r2 = pix;
            const uint32_t smp = r0 + 2 * r1 + r2;
            const uint32_t dfp = __builtin_bit_cast(uint32_t, __builtin_bit_cast(sw_short2, r2) - __builtin_bit_cast(sw_short2, r0));
            // to the right-hand neighbour: {sum, difference} of O; to the left-hand one: those of E
            const int pk_l = lane_read_i((int)__builtin_amdgcn_perm(smp, dfp, 0x07060302u), am1);   // O of lane l-1
            const int pk_r = lane_read_i((int)__builtin_amdgcn_perm(smp, dfp, 0x05040100u), ap1);   // E of lane l+1
            const int ix[2] = {(int)(smp >> 16) - (int)((unsigned)pk_l >> 16), (int)((unsigned)pk_r >> 16) - (int)(smp & 0xffffu)};
            // {df_l + 2 dfE + dfO, dfE + 2 dfO + df_r}
            const sw_short2 dfs = __builtin_bit_cast(sw_short2, dfp);
            const sw_short2 iyp = dfs + dfs + __builtin_bit_cast(sw_short2, __builtin_amdgcn_alignbit(dfp, dfp, 16)) +
                                  __builtin_bit_cast(sw_short2, __builtin_amdgcn_perm((uint32_t)pk_r, (uint32_t)pk_l, 0x05040100u));
            const int iy[2] = {(int)iyp.x, (int)iyp.y};
            int hs[2][3];                                      // horizontal 7-sums of the products
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int pe = ch == 0 ? __mul24(ix[0], ix[0]) : (ch == 1 ? __mul24(iy[0], iy[0]) : __mul24(ix[0], iy[0]));
                const int po = ch == 0 ? __mul24(ix[1], ix[1]) : (ch == 1 ? __mul24(iy[1], iy[1]) : __mul24(ix[1], iy[1]));
                const int pr = pe + po;
                const int mid = lane_read_i(pr, am1) + pr + lane_read_i(pr, ap1);
                hs[0][ch] = lane_read_i(po, am2) + mid;
                hs[1][ch] = mid + lane_read_i(pe, ap2);
            }
            const int yr = t - 4;                              // responses of row yr
            const bool row_valid = yr >= R && yr < h - R;
            float resp[2][2];                                  // [detector][column]
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) { vs[e][ch] += hs[e][ch] - ring[u][e][ch]; ring[u][e][ch] = hs[e][ch]; }
                const bool valid = col_valid[e] && row_valid;
                const float fa = (float)vs[e][0], fb = (float)vs[e][1], fc = (float)vs[e][2];
                const float t2 = __fmul_rn(fc, fc), t4 = __fadd_rn(fa, fb);
                resp[0][e] = 0.f; resp[1][e] = 0.f;
                if (DET & 1) {
                    const float t3 = __fsub_rn(__fmul_rn(fa, fb), t2);
                    const float t6 = __fmul_rn(a.k, __fmul_rn(t4, t4));
                    resp[0][e] = valid ? __fmul_rn(__fsub_rn(t3, t6), a.s4) : 0.0f;
                }
                if (DET & 2) {
                    const float m = __fmul_rn(t4, 0.5f), d = __fmul_rn(__fsub_rn(fa, fb), 0.5f);
                    const float root = sqrt_rn_normal(__fadd_rn(__fmul_rn(d, d), t2));     // 0 or >= 1/4: squares of halves of integers
                    resp[1][e] = valid ? __fmul_rn(__fsub_rn(m, root), a.s2) : 0.0f;
                }
            }
            // non-maximum suppression of row yc = t - 5 with the rows above (m_up) and below (this one)
            const int yc = t - 5;
            const bool row_out = yc >= ya && yc < yb && yc >= a.border && yc < h - a.border;
#pragma unroll
            for (int det = 0; det < 2; ++det) {
                if (!(DET & (1 << det))) continue;
                const float re = resp[det][0], ro = resp[det][1];
                const float l = lane_read_f(ro, am1), rr = lane_read_f(re, ap1);       // left of E, right of O
                const float m_dn[2] = {max3_raw(l, re, ro), max3_raw(re, ro, rr)};
                const float ce = cv[det][0], co = cv[det][1];
                // strictly above the threshold, the row above and the left neighbour; at least the right neighbour and the row below
                const bool hit_e = (int)(ce > max3_raw(thr_col[det][0], m_up[det][0], cl[det])) & (int)(ce >= max2_raw(co, m_dn[0]));
                const bool hit_o = (int)(co > max3_raw(thr_col[det][1], m_up[det][1], ce)) & (int)(co >= max2_raw(cr[det], m_dn[1]));
                if (row_out && __builtin_amdgcn_ballot_w64((int)hit_e | (int)hit_o)) {
                    emit(det, hit_e, ce, gx0, yc);
                    emit(det, hit_o, co, gx0 + 1, yc);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) { m_up[det][e] = m_c[det][e]; m_c[det][e] = m_dn[e]; }
                cv[det][0] = re; cv[det][1] = ro; cl[det] = l; cr[det] = rr;
            }
        }
    }
#pragma unroll
    for (int det = 0; det < 2; ++det) {
        if (!(DET & (1 << det)) || cnt[det] == 0) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(a.cand_count + ((size_t)frame * 2 + det), cnt[det]);
        base = __shfl(base, 0, 64);
        uint64_t *dst = a.cand + ((size_t)frame * 2 + det) * CAND_CAP;
        if (lane < cnt[det] && base + lane < CAND_CAP) dst[base + lane] = stage[det][lane];
    }
}

// Bitonic sort (descending) of one chunk of CHUNK candidate keys in LDS.
// grid = (CAND_CAP / CHUNK, frames, 2 detectors); chunks beyond the candidate count exit.  THREADS: a power of two, at
// most CHUNK / 2 (a pass of the network is a barrier of the whole workgroup: few wavefronts make it cheap).
template <int CHUNK, int THREADS>
__global__ __launch_bounds__(THREADS) void sort_chunk_kernel(uint64_t *cand, const int32_t *cand_count, int det_mask)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    const int det = blockIdx.y;
    if (!(det_mask & (1 << det))) return;
    const size_t list = (size_t)blockIdx.x * 2 + det;
    const int total = min(cand_count[list], CAND_CAP);
    const int base = blockIdx.z * CHUNK;
    if (base >= total) return;
    const int nloc = min(CHUNK, total - base);
    int S = THREADS;
    while (S < nloc) S <<= 1;
    uint64_t *src = cand + list * CAND_CAP + base;
    const int tid = threadIdx.x;
    for (int i = tid; i < S; i += THREADS) keys[i] = (i < nloc) ? src[i] : 0ull;
    __syncthreads();
    // The passes with partner distance below LB of a stage stay inside aligned blocks of LB keys: a thread
    // takes such a block into registers and runs them (for the stages up to LB: all of their passes) without a
    // barrier in between.  S >= THREADS >= 2 * LB, so every later stage has its direction constant over a block.
    constexpr int LB = 8;            // 16 measured slower: half the threads idle in the local phase
    auto local_passes = [&](int k_first, int k_last) {
        for (int blk = tid; blk < S / LB; blk += THREADS) {
            uint64_t r[LB];
#pragma unroll
            for (int q = 0; q < LB; ++q) r[q] = keys[LB * blk + q];
            for (int k = k_first; k <= k_last; k <<= 1) {
#pragma unroll
                for (int j = LB / 2; j > 0; j >>= 1) {
                    if (j < k) {
#pragma unroll
                        for (int q = 0; q < LB; ++q) {
                            if ((q & j) == 0) {
                                const bool desc = ((LB * blk + q) & k) == 0;
                                const uint64_t a = r[q], b = r[q | j];
                                const bool sw = (a < b) == desc;
                                r[q] = sw ? b : a; r[q | j] = sw ? a : b;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < LB; ++q) keys[LB * blk + q] = r[q];
        }
        __syncthreads();
    };
    auto strided_passes = [&](int k, int jhi, auto NP) {
        constexpr int np = decltype(NP)::value, nk = 1 << np;
        const int jlo = jhi >> (np - 1);
        for (int t = tid; t < S / nk; t += THREADS) {
            const int low = t & (jlo - 1), base = (t - low) * nk + low;
            const bool desc = (base & k) == 0;
            uint64_t r[nk];
#pragma unroll
            for (int q = 0; q < nk; ++q) r[q] = keys[base + q * jlo];
#pragma unroll
            for (int jj = nk / 2; jj > 0; jj >>= 1) {
#pragma unroll
                for (int q = 0; q < nk; ++q) {
                    if ((q & jj) == 0) {
                        const uint64_t a = r[q], b = r[q | jj];
                        const bool sw = (a < b) == desc;
                        r[q] = sw ? b : a; r[q | jj] = sw ? a : b;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < nk; ++q) keys[base + q * jlo] = r[q];
        }
        __syncthreads();
    };
    local_passes(2, LB);
    for (int k = 2 * LB; k <= S; k <<= 1) {
        // The passes with partner distance >= LB of a stage, up to three at a time: the passes jhi, jhi/2, jhi/4 only ever
        // pair keys whose indices differ in those three bits, so a thread takes the eight keys base + q * (jhi/4) into
        // registers and runs the three passes without a barrier in between (their direction is that of bit k of base,
        // which the eight indices share: 2 * jhi <= k).
        int c = 0;
        for (int j = k >> 1; j >= LB; j >>= 1) ++c;
        for (int jhi = k >> 1; c > 0;) {
            const int np = c < 3 ? c : 3;
            if (np == 3) strided_passes(k, jhi, std::integral_constant<int, 3>());
            else if (np == 2) strided_passes(k, jhi, std::integral_constant<int, 2>());
            else strided_passes(k, jhi, std::integral_constant<int, 1>());
            jhi >>= np; c -= np;
        }
        local_passes(k, k);
    }
    // write back the sorted chunk, padded to CHUNK with zeros so merges can read blindly
    for (int i = tid; i < CHUNK; i += THREADS) src[i] = (i < S) ? keys[i] : 0ull;
}

// The sorted parts (SORT_P keys each, descending, zero padded: sort_chunk_kernel<SORT_P>) of one SORT_S-chunk of a list
// merged in LDS into one sorted chunk, zero padded to SORT_S.  Neighbouring sorted runs of m keys become runs of 2m:
// one pass pairs key i of the first run with key m-1-i of the second (after it every key of the first half is at least
// every key of the second and both halves are bitonic), then the passes with partner distance m/2 ... 1 sort the halves;
// all comparisons point the same way, so the zeros at the end never move.  A list of n candidates thus has its first
// sort spread over ceil(n / SORT_P) workgroups instead of one.
// grid = (frames, 2 detectors, CAND_CHUNKS).
constexpr int SORT_P = 2048;
__global__ __launch_bounds__(1024) void merge_parts_kernel(uint64_t *cand, const int32_t *cand_count, int det_mask)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    const int det = blockIdx.y;
    if (!(det_mask & (1 << det))) return;
    const size_t list = (size_t)blockIdx.x * 2 + det;
    const int total = min(cand_count[list], CAND_CAP);
    const int base = blockIdx.z * SORT_S;
    if (base >= total) return;
    const int nparts = (min(SORT_S, total - base) + SORT_P - 1) / SORT_P;
    uint64_t *src = cand + list * CAND_CAP + base;
    const int tid = threadIdx.x;
    if (nparts == 1) {                                      // sorted already; the later merges read whole chunks
        if (total > SORT_S)
            for (int i = SORT_P + tid; i < SORT_S; i += 1024) src[i] = 0ull;
        return;
    }
    int M = 2 * SORT_P;
    while (M < nparts * SORT_P) M <<= 1;
    for (int i = tid; i < M; i += 1024) keys[i] = (i < nparts * SORT_P) ? src[i] : 0ull;
    __syncthreads();
    constexpr int LB = 8;
    auto cmpx = [](uint64_t &a, uint64_t &b) { const uint64_t x = a, y = b; const bool sw = x < y; a = sw ? y : x; b = sw ? x : y; };
    auto strided_passes = [&](int jhi, auto NP) {          // the passes jhi, jhi/2, ... (np of them) on 2^np keys per thread
        constexpr int np = decltype(NP)::value, nk = 1 << np;
        const int jlo = jhi >> (np - 1);
        for (int t = tid; t < M / nk; t += 1024) {
            const int low = t & (jlo - 1), b0 = (t - low) * nk + low;
            uint64_t r[nk];
#pragma unroll
            for (int q = 0; q < nk; ++q) r[q] = keys[b0 + q * jlo];
#pragma unroll
            for (int jj = nk / 2; jj > 0; jj >>= 1) {
#pragma unroll
                for (int q = 0; q < nk; ++q)
                    if ((q & jj) == 0) cmpx(r[q], r[q | jj]);
            }
#pragma unroll
            for (int q = 0; q < nk; ++q) keys[b0 + q * jlo] = r[q];
        }
        __syncthreads();
    };
    for (int m = SORT_P; m < M; m <<= 1) {
        for (int t = tid; t < (M >> 1); t += 1024) {
            const int off = t & (m - 1), b0 = (t - off) * 2;
            uint64_t a = keys[b0 + off], b = keys[b0 + 2 * m - 1 - off];
            if (a < b) { keys[b0 + off] = b; keys[b0 + 2 * m - 1 - off] = a; }
        }
        __syncthreads();
        int c = 0;
        for (int j = m >> 1; j >= LB; j >>= 1) ++c;
        for (int jhi = m >> 1; c > 0;) {
            const int np = c < 3 ? c : 3;
            if (np == 3) strided_passes(jhi, std::integral_constant<int, 3>());
            else if (np == 2) strided_passes(jhi, std::integral_constant<int, 2>());
            else strided_passes(jhi, std::integral_constant<int, 1>());
            jhi >>= np; c -= np;
        }
        strided_passes(LB / 2, std::integral_constant<int, 3>());     // partner distances 4, 2, 1: eight neighbouring keys
    }
    for (int i = tid; i < SORT_S; i += 1024) src[i] = (i < M) ? keys[i] : 0ull;
}

// Merge sorted chunks c and c+step (both descending, zero padded) keeping the top SORT_S in chunk c.
// grid = (frames, 2 detectors, pairs of chunks); lists that fit one chunk -- the rule -- leave at once.
__global__ __launch_bounds__(1024) void merge_chunks_kernel(uint64_t *cand, const int32_t *cand_count, int step, int det_mask)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    const int det = blockIdx.y;
    if (!(det_mask & (1 << det))) return;
    const size_t list = (size_t)blockIdx.x * 2 + det;
    const int total = min(cand_count[list], CAND_CAP);
    const int ca = blockIdx.z * 2 * step, cb = ca + step;
    if (cb * SORT_S >= total) return;                   // partner chunk is empty
    uint64_t *A = cand + list * CAND_CAP + (size_t)ca * SORT_S;
    const uint64_t *B = cand + list * CAND_CAP + (size_t)cb * SORT_S;
    const int tid = threadIdx.x;
    // max(A[i], B[S-1-i]) is the top-S set as a bitonic sequence
    for (int i = tid; i < SORT_S; i += 1024) {
        uint64_t x = A[i], y = B[SORT_S - 1 - i];
        keys[i] = x > y ? x : y;
    }
    __syncthreads();
    for (int j = SORT_S >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (SORT_S >> 1); t += 1024) {
            int i = 2 * t - (t & (j - 1)), p = i + j;
            uint64_t a = keys[i], b = keys[p];
            if (a < b) { keys[i] = b; keys[p] = a; }
        }
        __syncthreads();
    }
    for (int i = tid; i < SORT_S; i += 1024) A[i] = keys[i];
}

// Every candidate list of a batch sorted by descending key, the leading SORT_S keys of each in place: parts of SORT_P keys
// sorted side by side, merged per SORT_S-chunk in LDS, then the chunks of a long list merged pairwise keeping the top SORT_S.
// one_chunk: the caller knows that no list holds more than SORT_S keys (the chunk merges are then not launched).
static void launch_sort_lists(hipStream_t s, uint64_t *cand, const int32_t *cand_count, int frames, int det_mask, bool one_chunk = false)
{
    static const bool whole = getenv("VSLAM_SORT_WHOLE") != nullptr;        // A/B runs: one workgroup per SORT_S-chunk
    if (whole) {
        hipLaunchKernelGGL((sort_chunk_kernel<SORT_S, 1024>), dim3(frames, 2, CAND_CHUNKS), dim3(1024), SORT_S * 8, s, cand, cand_count, det_mask);
    } else {
        hipLaunchKernelGGL((sort_chunk_kernel<SORT_P, 256>), dim3(frames, 2, CAND_CAP / SORT_P), dim3(256), SORT_P * 8, s, cand, cand_count, det_mask);
        hipLaunchKernelGGL(merge_parts_kernel, dim3(frames, 2, CAND_CHUNKS), dim3(1024), SORT_S * 8, s, cand, cand_count, det_mask);
    }
    if (one_chunk) return;
    for (int step = 1; step < CAND_CHUNKS; step <<= 1)
        hipLaunchKernelGGL(merge_chunks_kernel, dim3(frames, 2, CAND_CHUNKS / (2 * step)), dim3(1024), SORT_S * 8, s, cand, cand_count, step, det_mask);
}

// Unpacks the leading keys of each list into Feature records.
__global__ void emit_features_kernel(const uint64_t *cand, const int32_t *cand_count, int det_first, int ndet,
                                     int cap, int max_features, vslam_feature *out, int32_t *count)
{
    const int f = blockIdx.y, d = blockIdx.z;                  // d indexes the requested detectors
    const int det = det_first + d;
    const size_t list = (size_t)f * 2 + det;
    const int raw = cand_count[list];
    const int total = min(raw, CAND_CAP);
    int nout = min(min(total, cap), min(max_features, SORT_S));
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) count[(size_t)f * ndet + d] = raw > CAND_CAP ? -1 : nout;   // -1: candidate list overflowed
    if (i >= nout) return;
    vslam_feature ft;
    unpack_key(cand[list * CAND_CAP + i], &ft);
    out[((size_t)f * ndet + d) * cap + i] = ft;
}

// ---- FAST-9/16 segment test (first stage of cv::ORB::detect as configured at reference
// src/imagefeatures.cpp:73-87: fastThreshold 20, non-maximum suppression) -------------------------
// strength s(p) = max over the 16 arcs of 9 contiguous pixels of the radius-3 circle of the arc's
// minimum of (I_q - I_p), and the same for (I_p - I_q); corner iff s > threshold, score s - 1; with
// nonmax a corner survives iff it is stronger than every neighbouring corner.  Integer work, equal to
// the CPU oracle bit for bit.  One workgroup per 64 x 16 tile: image tile with a 4-pixel halo in LDS,
// strengths of the tile and a 1-pixel ring in LDS, then suppression and one reservation per workgroup
// in the frame's candidate list, which the sort / merge / emit kernels above finish.
constexpr int FT_T = 256;          // threads per tile (128 and 512 measured 10 % slower)
constexpr int FT_W = 64, FT_H = 32, FT_IW = FT_W + 8, FT_IH = FT_H + 8, FT_SW = FT_W + 2, FT_SH = FT_H + 2;

struct FastArgs {
    const uint8_t *gray; size_t stride_frame;
    int w, h, stride, threshold, nonmax, tiles_x, frames;
    int border = 0;        // corners closer than this to an image side are dropped after suppression (ORB's edge threshold)
    uint64_t *cand;        // [frame][2][CAND_CAP], list 0 used
    int32_t *cand_count;   // [frame][2]
};

__device__ __forceinline__ int fast9_strength(const int (&d)[16])
{
    int lo2[16], hi2[16], lo4[16], hi4[16], best = 0, worst = 255;
#pragma unroll
    for (int i = 0; i < 16; ++i) { lo2[i] = min(d[i], d[(i + 1) & 15]); hi2[i] = max(d[i], d[(i + 1) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; ++i) { lo4[i] = min(lo2[i], lo2[(i + 2) & 15]); hi4[i] = max(hi2[i], hi2[(i + 2) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int lo9 = min(min(lo4[i], lo4[(i + 4) & 15]), d[(i + 8) & 15]);      // min over ring pixels i .. i+8
        const int hi9 = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
        best = max(best, lo9);                                                    // brighter arc
        worst = min(worst, hi9);                                                  // darker arc: its minimum of -d is -hi9
    }
    return max(best, -worst);
}

typedef short ft_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ft_short2 ft_even(uint32_t x) { return __builtin_bit_cast(ft_short2, __builtin_amdgcn_perm(x, x, 0x0c020c00u)); }   // bytes 0, 2 as two 16-bit values
__device__ __forceinline__ ft_short2 ft_odd(uint32_t x) { return __builtin_bit_cast(ft_short2, __builtin_amdgcn_perm(x, x, 0x0c030c01u)); }    // bytes 1, 3
__device__ __forceinline__ uint32_t ft_bits(ft_short2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ int ft_rank(unsigned long long m)        // active lanes below this one in the ballot m
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__global__ __launch_bounds__(FT_T) void fast9_tile_kernel(FastArgs a)
{
    constexpr int FT_WPR = (FT_IW + 4) / 4;            // words per row of the tile image
    constexpr int FT_GROUPS = (FT_SW + 3 + 3) / 4;     // aligned groups of four columns that cover the columns with a strength
    __shared__ __align__(4) uint8_t img[FT_IH][FT_IW + 4];
    __shared__ __align__(4) uint8_t st[FT_SH][FT_SW + 2];              // strengths (<= 255), 0 = no corner
    constexpr int FT_STAGE = 512;                      // staged candidates per tile; the rare tile with more appends the rest one by one
    __shared__ uint64_t stage[FT_STAGE];
    __shared__ uint16_t queue[FT_SH * FT_SW];          // positions (lx | ly << 8) that passed both quick tests
    __shared__ uint16_t corners[FT_SH * FT_SW];        // ... of those, the ones with an arc of nine
    __shared__ int lcnt, gbase, qn, cn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int frame = blockIdx.x % a.frames, tile = blockIdx.x / a.frames;
    const int x0 = (tile % a.tiles_x) * FT_W, y0 = (tile / a.tiles_x) * FT_H;
    const uint8_t *gray = a.gray + (size_t)frame * a.stride_frame;
    const int w = a.w, h = a.h, t = a.threshold;
    if (tid == 0) { lcnt = 0; qn = 0; cn = 0; }
    for (int i = tid; i < (int)(sizeof(st) / 4); i += FT_T) reinterpret_cast<uint32_t *>(&st[0][0])[i] = 0u;
    const bool words = ((reinterpret_cast<uintptr_t>(gray) | (uintptr_t)a.stride) & 3) == 0;
    // the tile with its halo and the ring of strengths lie wholly inside the image, at least 3 pixels from its sides
    const bool inner = x0 >= 4 && y0 >= 4 && x0 + FT_W + 4 <= w && y0 + FT_H + 4 <= h;
    if (words && inner) {
        for (int idx = tid; idx < FT_IH * FT_WPR; idx += FT_T) {
            const int ly = idx / FT_WPR, lw = idx - ly * FT_WPR;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(gray + (size_t)(y0 - 4 + ly) * a.stride + (x0 - 4));
            *reinterpret_cast<uint32_t *>(&img[ly][4 * lw]) = lw < FT_IW / 4 ? src[lw] : 0u;
        }
    } else if (words) {
        // rows start on 4-byte boundaries and x0 - 4 is a multiple of 4: whole words, bytes past the image zeroed
        for (int idx = tid; idx < FT_IH * FT_WPR; idx += FT_T) {
            const int ly = idx / FT_WPR, lw = idx - ly * FT_WPR, gx = x0 - 4 + 4 * lw, gy = y0 - 4 + ly;
            uint32_t v = 0;
            if (lw < FT_IW / 4 && gy >= 0 && gy < h && gx >= 0 && gx < w) {
                if (gx + 4 <= a.stride) v = *reinterpret_cast<const uint32_t *>(gray + (size_t)gy * a.stride + gx);
                else for (int q = 0; q < 4 && gx + q < a.stride; ++q) v |= (uint32_t)gray[(size_t)gy * a.stride + gx + q] << (8 * q);
                if (gx + 4 > w) v &= 0xffffffffu >> (8 * (gx + 4 - w));
            }
            *reinterpret_cast<uint32_t *>(&img[ly][4 * lw]) = v;
        }
    } else {
        for (int idx = tid; idx < FT_IH * (FT_IW + 4); idx += FT_T) {
            const int ly = idx / (FT_IW + 4), lx = idx - ly * (FT_IW + 4), gx = x0 - 4 + lx, gy = y0 - 4 + ly;
            img[ly][lx] = (lx < FT_IW && gx >= 0 && gy >= 0 && gx < w && gy < h) ? gray[(size_t)gy * a.stride + gx] : (uint8_t)0;
        }
    }
    __syncthreads();
    // Strengths of the tile and its 1-pixel ring (FT_SW x FT_SH), in three passes.
    // Pass 1 -- every position, four to a lane: an arc of 9 of the 16 ring pixels contains one pixel of every opposite
    // pair, so the vertical pair rejects most pixels and the horizontal one most others.  A lane takes an aligned word of
    // four centre pixels with the words three rows above and below and the two words shifted by three columns, and does
    // the comparisons on pairs of 16-bit values (c + t < max(pair) or min(pair) < c - t, read off the sign bits of packed
    // differences).  Positions that pass are appended to a queue in LDS.
    // Pass 2 -- the queue, densely packed: sixteen ring pixels into two 16-bit masks (brighter than c + t, darker than
    // c - t); a corner has nine contiguous bits in one of them.  Corners go to a second queue.
    // Pass 3 -- the corners: the exact strength (best arc minimum), which only they need.
    // (Evaluated in place, a wavefront pays for the full test whenever ONE of its 64 pixels needs it, which on textured
    // frames is most wavefronts.)
    const uint32_t *imgw = reinterpret_cast<const uint32_t *>(&img[0][0]);
    for (int idx = tid; idx < FT_SH * FT_GROUPS; idx += FT_T) {
        const int ly = idx / FT_GROUPS, k = idx - ly * FT_GROUPS;
        const uint32_t *row = imgw + (ly + 3) * FT_WPR + k;
        const uint32_t c = row[0], wl = k > 0 ? row[-1] : 0u, wr = row[1], up = row[-3 * FT_WPR], dn = row[3 * FT_WPR];
        const uint32_t l3 = __builtin_amdgcn_alignbyte(c, wl, 1), r3 = __builtin_amdgcn_alignbyte(wr, c, 3);
        const ft_short2 tt = {(short)t, (short)t};
        uint32_t pe, po;
        {
            const ft_short2 cc = ft_even(c), ct = cc + tt, cm = cc - tt;
            const ft_short2 u = ft_even(up), d = ft_even(dn), l = ft_even(l3), r = ft_even(r3);
            const ft_short2 xu = ct - __builtin_elementwise_max(u, d), xd = __builtin_elementwise_min(u, d) - cm;
            const ft_short2 yu = ct - __builtin_elementwise_max(l, r), yd = __builtin_elementwise_min(l, r) - cm;
            pe = (ft_bits(xu) & ft_bits(yu)) | (ft_bits(xd) & ft_bits(yd));
        }
        {
            const ft_short2 cc = ft_odd(c), ct = cc + tt, cm = cc - tt;
            const ft_short2 u = ft_odd(up), d = ft_odd(dn), l = ft_odd(l3), r = ft_odd(r3);
            const ft_short2 xu = ct - __builtin_elementwise_max(u, d), xd = __builtin_elementwise_min(u, d) - cm;
            const ft_short2 yu = ct - __builtin_elementwise_max(l, r), yd = __builtin_elementwise_min(l, r) - cm;
            po = (ft_bits(xu) & ft_bits(yu)) | (ft_bits(xd) & ft_bits(yd));
        }
        // pixel j of the group: column lx = 4 k + j - 3 of the strength array, image column x0 - 4 + 4 k + j
        uint32_t m4 = ((pe >> 15) & 0x10001u) | ((po >> 14) & 0x20002u);
        m4 = (m4 | (m4 >> 14)) & 0xfu;                                          // bit j: pixel j passed
        if (inner) m4 &= k == 0 ? 0x8u : (k == FT_GROUPS - 1 ? 0x1u : 0xfu);  // columns 0 .. FT_SW - 1 of the strength array
        else {
            const int py = y0 - 1 + ly, gx = x0 - 4 + 4 * k;
            const int jlo = max(max(3 - 4 * k, 3 - gx), 0), jhi = min(min(FT_SW + 2 - 4 * k, w - 4 - gx), 3);
            m4 &= (py >= 3 && py < h - 3 && jhi >= jlo) ? ((2u << jhi) - 1u) & ~((1u << jlo) - 1u) : 0u;
        }
        if (m4) {
            // one LDS atomic per lane with a survivor (about one lane in five), written out: the compiler would turn a plain
            // atomicAdd of a lane-dependent value into a loop over the active lanes
            int at;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(at) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)&qn), "v"(__popc(m4)) : "memory");
            const int pos = (4 * k - 3) + (ly << 8);
            if (m4 & 1u) queue[at++] = (uint16_t)pos;
            if (m4 & 2u) queue[at++] = (uint16_t)(pos + 1);
            if (m4 & 4u) queue[at++] = (uint16_t)(pos + 2);
            if (m4 & 8u) queue[at] = (uint16_t)(pos + 3);
        }
    }
    __syncthreads();
    const int nq = qn;
    for (int i0 = 0; i0 < nq; i0 += FT_T) {
        const int i = i0 + tid;
        bool corner = false;
        int pos = 0;
        if (i < nq) {
            pos = queue[i];
            const int lx = pos & 255, ly = pos >> 8;
            const uint8_t *c = &img[ly + 3][lx + 3];
            constexpr int R = FT_IW + 4;
            const int v = c[0], tv = v + t, vm = v - t;
            uint32_t mb = 0, md = 0;
            auto ring = [&](int off) {
                const int I = c[off];
                mb = __builtin_amdgcn_alignbit(mb, (uint32_t)(tv - I), 31);        // shift in "I > v + t"
                md = __builtin_amdgcn_alignbit(md, (uint32_t)(I - vm), 31);        // shift in "I < v - t"
            };
            ring(3 * R);      ring(3 * R + 1);  ring(2 * R + 2);  ring(R + 3);
            ring(3);          ring(-R + 3);     ring(-2 * R + 2); ring(-3 * R + 1);
            ring(-3 * R);     ring(-3 * R - 1); ring(-2 * R - 2); ring(-R - 3);
            ring(-3);         ring(R - 3);      ring(2 * R - 2);  ring(3 * R - 1);
            auto nine = [](uint32_t m) {                                            // nine contiguous bits of the circular 16-bit mask
                const uint32_t x = m | (m << 16);
                const uint32_t r2 = x & (x >> 1), r4 = r2 & (r2 >> 2), r8 = r4 & (r4 >> 4);
                return (r8 & (x >> 8)) & 0xffffu;
            };
            corner = (nine(mb) | nine(md)) != 0u;
        }
        const unsigned long long m = __ballot(corner);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&cn, __popcll(m));
            base = __shfl(base, leader, 64);
            if (corner) corners[base + ft_rank(m)] = (uint16_t)pos;
        }
    }
    __syncthreads();
    const int nc = cn;
    for (int i = tid; i < nc; i += FT_T) {
        const int lx = corners[i] & 255, ly = corners[i] >> 8;
        const int cy = ly + 3, cx = lx + 3, v = img[cy][cx];
        int d[16];
        d[0] = img[cy + 3][cx] - v;       d[1] = img[cy + 3][cx + 1] - v;  d[2] = img[cy + 2][cx + 2] - v;  d[3] = img[cy + 1][cx + 3] - v;
        d[4] = img[cy][cx + 3] - v;       d[5] = img[cy - 1][cx + 3] - v;  d[6] = img[cy - 2][cx + 2] - v;  d[7] = img[cy - 3][cx + 1] - v;
        d[8] = img[cy - 3][cx] - v;       d[9] = img[cy - 3][cx - 1] - v;  d[10] = img[cy - 2][cx - 2] - v; d[11] = img[cy - 1][cx - 3] - v;
        d[12] = img[cy][cx - 3] - v;      d[13] = img[cy + 1][cx - 3] - v; d[14] = img[cy + 2][cx - 2] - v; d[15] = img[cy + 3][cx - 1] - v;
        const int sv = fast9_strength(d);
        if (sv > t) st[ly][lx] = (uint8_t)sv;
    }
    __syncthreads();
    // suppression and output: the corners of the tile proper (the ring only serves as neighbours)
    for (int i = tid; i < nc; i += FT_T) {
        const int lx = (corners[i] & 255) - 1, ly = (corners[i] >> 8) - 1;
        if (lx < 0 || ly < 0 || lx >= FT_W || ly >= FT_H) continue;
        const int px = x0 + lx, py = y0 + ly;
        if (px >= w || py >= h) continue;
        if (px < a.border || py < a.border || px >= w - a.border || py >= h - a.border) continue;
        const int s = st[ly + 1][lx + 1];
        if (s == 0) continue;
        bool keep = true;
        if (a.nonmax) {
            const int nb = max(max(max(st[ly][lx], st[ly][lx + 1]), max(st[ly][lx + 2], st[ly + 1][lx])),
                               max(max(st[ly + 1][lx + 2], st[ly + 2][lx]), max(st[ly + 2][lx + 1], st[ly + 2][lx + 2])));
            keep = s > nb;
        }
        if (keep) {
            const uint64_t key = make_key((float)(s - 1), px, py);
            const int slot = atomicAdd(&lcnt, 1);
            if (slot < FT_STAGE) stage[slot] = key;
            else {
                const int gslot = atomicAdd(a.cand_count + (size_t)frame * 2, 1);
                if (gslot < CAND_CAP) a.cand[(size_t)frame * 2 * CAND_CAP + gslot] = key;
            }
        }
    }
    __syncthreads();
    const int nstaged = min(lcnt, FT_STAGE);
    if (tid == 0) gbase = nstaged ? atomicAdd(a.cand_count + (size_t)frame * 2, nstaged) : 0;
    __syncthreads();
    uint64_t *dst = a.cand + (size_t)frame * 2 * CAND_CAP;
    for (int i = tid; i < nstaged; i += FT_T)
        if (gbase + i < CAND_CAP) dst[gbase + i] = stage[i];
}

static size_t detect_ws_bytes(int frames);

// ---- keypoints of one ORB pyramid level, cv::ORB::detect as the reference configures it (src/imagefeatures.cpp:73-87:
// HARRIS_SCORE, edgeThreshold 75, fastThreshold 20).  OpenCV's sources are not part of the reference; the chain restated
// here (and in the CPU oracle, which defines the contract) is the published one: FAST-9/16 with non-maximum suppression,
// corners closer than the edge threshold to a side dropped, the 2 n strongest by FAST score kept (with every tie of the
// last one), the Harris measure of a 7 x 7 block of Sobel-like integer gradients evaluated AT those corners only
// (k = 0.04, scaled by (1 / (4 * 7 * 255))^4), the n best by that measure kept (again with ties).
// Candidate list 0 of the frame holds the FAST keys, list 1 the Harris keys; both are sorted by the kernels above.

// One workgroup per frame: how many leading keys of the sorted list survive "retain the n best, ties included".
// kept[frame] = that count, or -1 when the ties run past the part of the list that is sorted (SORT_S keys).
__global__ __launch_bounds__(1024) void orb_cut_kernel(const uint64_t *cand, const int32_t *cand_count, int list_of_frame, int n_keep, int32_t *kept)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    const size_t list = (size_t)f * 2 + list_of_frame;
    const int raw = cand_count[list];
    const uint64_t *keys = cand + list * CAND_CAP;
    const int avail = min(raw, SORT_S);                       // sorted prefix
    if (raw <= n_keep) { if (tid == 0) kept[f] = raw > CAND_CAP ? -1 : raw; return; }
    if (n_keep <= 0) { if (tid == 0) kept[f] = 0; return; }
    if (n_keep > avail) { if (tid == 0) kept[f] = -1; return; }
    const uint32_t cut = (uint32_t)(keys[n_keep - 1] >> 32);  // score bits of the n-th best
    for (int i = n_keep - 1 + tid; i < avail; i += 1024) {
        const bool in = (uint32_t)(keys[i] >> 32) == cut;
        const bool next_in = (i + 1 < avail) && (uint32_t)(keys[i + 1] >> 32) == cut;
        if (in && !next_in) kept[f] = (i + 1 == avail && raw > avail) ? -1 : i + 1;     // exactly one thread finds the end of the ties
    }
}

struct OrbHarrisArgs {
    const uint8_t *gray; size_t stride_frame;
    int w, h, stride, block;
    float k, scale4;
    uint64_t *cand; int32_t *cand_count; const int32_t *kept;
};

// Thread i of a frame: Harris measure at the i-th kept FAST corner -> key i of list 1.
__global__ __launch_bounds__(256) void orb_harris_kernel(OrbHarrisArgs a)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int kept = a.kept[f];
    if (i == 0) a.cand_count[(size_t)f * 2 + 1] = kept < 0 ? 0 : kept;
    if (i >= kept) return;
    vslam_feature ft;
    unpack_key(a.cand[(size_t)f * 2 * CAND_CAP + i], &ft);
    const uint8_t *gray = a.gray + (size_t)f * a.stride_frame;
    const int r = a.block / 2, st = a.stride;
    int sa = 0, sb = 0, sc = 0;
    for (int dy = -r; dy < a.block - r; ++dy) {
        const uint8_t *p = gray + (size_t)(ft.y + dy) * st + ft.x - r;
        for (int dx = 0; dx < a.block; ++dx, ++p) {
            const int Ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-st + 1] - (int)p[-st - 1]) + ((int)p[st + 1] - (int)p[st - 1]);
            const int Iy = ((int)p[st] - (int)p[-st]) * 2 + ((int)p[st - 1] - (int)p[-st - 1]) + ((int)p[st + 1] - (int)p[-st + 1]);
            sa += Ix * Ix; sb += Iy * Iy; sc += Ix * Iy;
        }
    }
    const float fa = (float)sa, fb = (float)sb, fc = (float)sc;
    const float tr = __fadd_rn(fa, fb);
    const float resp = __fmul_rn(__fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(a.k, tr), tr)), a.scale4);
    a.cand[((size_t)f * 2 + 1) * CAND_CAP + i] = make_key(resp, ft.x, ft.y);
}

// The same for the 7 x 7 block on an image whose rows start on 4-byte boundaries: the 9 x 9 patch of a corner arrives as
// nine 12-byte loads (the aligned words that cover columns x-4 .. x+4) instead of ~400 scattered byte reads -- the
// kernel above spends its time on cache-line transactions, 64 per load instruction of a wavefront -- and the gradients
// come from shared differences: Ix = (1 2 1)^T applied to the horizontal central differences, Iy = (1 2 1) applied to
// the vertical ones.  Integer sums: the same values whatever the order.
__global__ __launch_bounds__(256) void orb_harris7_kernel(OrbHarrisArgs a)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int kept = a.kept[f];
    if (i == 0) a.cand_count[(size_t)f * 2 + 1] = kept < 0 ? 0 : kept;
    if (i >= kept) return;
    vslam_feature ft;
    unpack_key(a.cand[(size_t)f * 2 * CAND_CAP + i], &ft);
    const uint8_t *gray = a.gray + (size_t)f * a.stride_frame;
    const int xa = (ft.x - 4) & ~3, sh = (ft.x - 4) - xa;
    int p[9][9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(gray + (size_t)(ft.y - 4 + r) * a.stride + xa);
        const uint32_t w0 = row[0], w1 = row[1], w2 = row[2];
        const uint32_t b0 = __builtin_amdgcn_alignbyte(w1, w0, sh), b1 = __builtin_amdgcn_alignbyte(w2, w1, sh), b2 = w2 >> (8 * sh);
#pragma unroll
        for (int c = 0; c < 4; ++c) { p[r][c] = (b0 >> (8 * c)) & 255; p[r][4 + c] = (b1 >> (8 * c)) & 255; }
        p[r][8] = b2 & 255;
    }
    int D[9][7], V[7][9];                                  // D[r][c] = p[r][c+2] - p[r][c],  V[r][c] = p[r+2][c] - p[r][c]
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int c = 0; c < 7; ++c) D[r][c] = p[r][c + 2] - p[r][c];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int c = 0; c < 9; ++c) V[r][c] = p[r + 2][c] - p[r][c];
    int sa = 0, sb = 0, sc = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const int Ix = D[r][c] + 2 * D[r + 1][c] + D[r + 2][c], Iy = V[r][c] + 2 * V[r][c + 1] + V[r][c + 2];
            sa += Ix * Ix; sb += Iy * Iy; sc += Ix * Iy;
        }
    const float fa = (float)sa, fb = (float)sb, fc = (float)sc;
    const float tr = __fadd_rn(fa, fb);
    const float resp = __fmul_rn(__fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(a.k, tr), tr)), a.scale4);
    a.cand[((size_t)f * 2 + 1) * CAND_CAP + i] = make_key(resp, ft.x, ft.y);
}

__global__ void orb_emit_kernel(const uint64_t *cand, const int32_t *kept, int cap, vslam_feature *out, int32_t *count)
{
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = kept[f];
    if (i == 0) count[f] = k;
    if (i >= k || i >= cap) return;
    vslam_feature ft;
    unpack_key(cand[((size_t)f * 2 + 1) * CAND_CAP + i], &ft);
    out[(size_t)f * cap + i] = ft;
}

// dst(x, y) of the bilinear resize used between pyramid levels: source position (x + 1/2) sw / dw - 1/2 in units of
// 1/2048 pixel (integer arithmetic throughout, so the CPU oracle and this kernel agree bit for bit), clamped to the image.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const uint8_t *src, int sw, int sh, int sstride, size_t sframe, uint8_t *dst, int dw,
                                                              int dh, int dstride, size_t dframe)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    src += (size_t)blockIdx.z * sframe;                       // frame blockIdx.z of a batch
    dst += (size_t)blockIdx.z * dframe;
    long long X = ((long long)(2 * x + 1) * sw * 1024) / dw - 1024, Y = ((long long)(2 * y + 1) * sh * 1024) / dh - 1024;
    X = X < 0 ? 0 : (X > (long long)(sw - 1) * 2048 ? (long long)(sw - 1) * 2048 : X);
    Y = Y < 0 ? 0 : (Y > (long long)(sh - 1) * 2048 ? (long long)(sh - 1) * 2048 : Y);
    const int x0 = (int)(X >> 11), y0 = (int)(Y >> 11), ax = (int)(X & 2047), ay = (int)(Y & 2047);
    const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
    const unsigned p00 = src[(size_t)y0 * sstride + x0], p01 = src[(size_t)y0 * sstride + x1];
    const unsigned p10 = src[(size_t)y1 * sstride + x0], p11 = src[(size_t)y1 * sstride + x1];
    const unsigned v = p00 * (2048 - ax) * (2048 - ay) + p01 * ax * (2048 - ay) + p10 * (2048 - ax) * ay + p11 * ax * ay;
    dst[(size_t)y * dstride + x] = (uint8_t)((v + (1u << 21)) >> 22);
}

// The same resize, four adjacent output pixels per thread: one 64-bit division per thread for the first pixel's source
// position and one for the row's, then the exact quotient and remainder are carried from pixel to pixel (the numerator
// grows by 2 * sw * 1024 = qs * dw + rs per pixel), and the four bytes leave as one word when the destination's rows
// start on 4-byte boundaries.  The kernel above spends its time on two 64-bit divisions per pixel.
__global__ __launch_bounds__(256) void resize_bilinear4_kernel(const uint8_t *src, int sw, int sh, int sstride, size_t sframe, uint8_t *dst, int dw,
                                                               int dh, int dstride, size_t dframe, int qs, int rs, int word_stores)
{
    const int x = 4 * (blockIdx.x * 256 + threadIdx.x), y = blockIdx.y;
    if (x >= dw) return;
    src += (size_t)blockIdx.z * sframe;
    dst += (size_t)blockIdx.z * dframe;
    long long Y = ((long long)(2 * y + 1) * sh * 1024) / dh - 1024;
    Y = Y < 0 ? 0 : (Y > (long long)(sh - 1) * 2048 ? (long long)(sh - 1) * 2048 : Y);
    const int y0 = (int)(Y >> 11), ay = (int)(Y & 2047), y1 = min(y0 + 1, sh - 1);
    const uint8_t *r0 = src + (size_t)y0 * sstride, *r1 = src + (size_t)y1 * sstride;
    const long long N = (long long)(2 * x + 1) * sw * 1024;
    long long q = N / dw;
    int r = (int)(N - q * dw);
    const long long xmax = (long long)(sw - 1) * 2048;
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long long X = q - 1024;
        X = X < 0 ? 0 : (X > xmax ? xmax : X);
        const int x0 = (int)(X >> 11), ax = (int)(X & 2047), x1 = min(x0 + 1, sw - 1);
        const unsigned p00 = r0[x0], p01 = r0[x1], p10 = r1[x0], p11 = r1[x1];
        const unsigned v = p00 * (2048 - ax) * (2048 - ay) + p01 * ax * (2048 - ay) + p10 * (2048 - ax) * ay + p11 * ax * ay;
        o |= ((v + (1u << 21)) >> 22) << (8 * i);
        q += qs; r += rs;
        if (r >= dw) { r -= dw; ++q; }
    }
    uint8_t *d = dst + (size_t)y * dstride + x;
    if (word_stores && x + 4 <= dw) *reinterpret_cast<uint32_t *>(d) = o;
    else for (int i = 0; i < 4 && x + i < dw; ++i) d[i] = (uint8_t)(o >> (8 * i));
}

// The same for sources whose rows start on 4-byte boundaries and shrink factors up to 2: the four output pixels of a
// thread take their source pixels from at most 11 adjacent columns, which arrive as three aligned words per source row
// (six loads instead of sixteen byte reads -- the byte form is bound by the number of memory instructions, not by their
// bytes or by arithmetic), and every division is a 32-bit one: the block's first numerator is uniform, a thread's is
// k = 4 * threadIdx.x steps of (qs, rs) further on, and rb + k * rs stays below 2^31.
__global__ __launch_bounds__(256) void resize_bilinear4w_kernel(const uint8_t *src, int sw, int sh, int sstride, size_t sframe, uint8_t *dst, int dw,
                                                                int dh, int dstride, size_t dframe, int qs, int rs, int word_stores)
{
    const int xb = 1024 * blockIdx.x, k = 4 * threadIdx.x, x = xb + k, y = blockIdx.y;
    if (x >= dw) return;
    src += (size_t)blockIdx.z * sframe;
    dst += (size_t)blockIdx.z * dframe;
    long long Y = ((long long)(2 * y + 1) * sh * 1024) / dh - 1024;
    Y = Y < 0 ? 0 : (Y > (long long)(sh - 1) * 2048 ? (long long)(sh - 1) * 2048 : Y);
    const int y0 = (int)(Y >> 11), ay = (int)(Y & 2047), y1 = min(y0 + 1, sh - 1);
    const uint8_t *r0 = src + (size_t)y0 * sstride, *r1 = src + (size_t)y1 * sstride;
    const long long Nb = (long long)(2 * xb + 1) * sw * 1024;              // uniform over the workgroup
    const long long qb = Nb / dw;
    const unsigned rb = (unsigned)(Nb - qb * dw);
    const unsigned rk = rb + (unsigned)k * (unsigned)rs, carry = rk / (unsigned)dw;
    int q = (int)qb + k * qs + (int)carry, r = (int)(rk - carry * (unsigned)dw);
    const int xmax = (sw - 1) * 2048;
    int X[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        X[i] = min(max(q - 1024, 0), xmax);
        q += qs; r += rs;
        if (r >= dw) { r -= dw; ++q; }
    }
    const int xa = (X[0] >> 11) & ~3;
    uint32_t o = 0;
    if (xa + 12 <= sstride && (X[3] >> 11) + 1 - xa <= 11) {
        const uint32_t *w0p = reinterpret_cast<const uint32_t *>(r0 + xa), *w1p = reinterpret_cast<const uint32_t *>(r1 + xa);
        const uint32_t a0 = w0p[0], a1 = w0p[1], a2 = w0p[2], b0 = w1p[0], b1 = w1p[1], b2 = w1p[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x0 = X[i] >> 11, ax = X[i] & 2047, off = x0 - xa;
            const uint32_t alo = off < 4 ? a0 : (off < 8 ? a1 : a2), ahi = off < 4 ? a1 : (off < 8 ? a2 : 0u);
            const uint32_t blo = off < 4 ? b0 : (off < 8 ? b1 : b2), bhi = off < 4 ? b1 : (off < 8 ? b2 : 0u);
            const uint32_t pa = __builtin_amdgcn_alignbyte(ahi, alo, off & 3), pb = __builtin_amdgcn_alignbyte(bhi, blo, off & 3);
            const bool last = x0 >= sw - 1;                                  // the right neighbour is the pixel itself
            const unsigned p00 = pa & 255u, p01 = last ? p00 : (pa >> 8) & 255u, p10 = pb & 255u, p11 = last ? p10 : (pb >> 8) & 255u;
            // (p00 (1 - ax) + p01 ax) (1 - ay) + (p10 (1 - ax) + p11 ax) ay: the same integer as the four-term sum, fewer operations
            const unsigned v = (p00 * (2048 - ax) + p01 * ax) * (2048 - ay) + (p10 * (2048 - ax) + p11 * ax) * ay;
            o |= ((v + (1u << 21)) >> 22) << (8 * i);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x0 = X[i] >> 11, ax = X[i] & 2047, x1 = min(x0 + 1, sw - 1);
            const unsigned p00 = r0[x0], p01 = r0[x1], p10 = r1[x0], p11 = r1[x1];
            // (p00 (1 - ax) + p01 ax) (1 - ay) + (p10 (1 - ax) + p11 ax) ay: the same integer as the four-term sum, fewer operations
            const unsigned v = (p00 * (2048 - ax) + p01 * ax) * (2048 - ay) + (p10 * (2048 - ax) + p11 * ax) * ay;
            o |= ((v + (1u << 21)) >> 22) << (8 * i);
        }
    }
    uint8_t *d = dst + (size_t)y * dstride + x;
    if (word_stores && x + 4 <= dw) *reinterpret_cast<uint32_t *>(d) = o;
    else for (int i = 0; i < 4 && x + i < dw; ++i) d[i] = (uint8_t)(o >> (8 * i));
}

static vslam_status launch_resize(vslam_ctx *ctx, const uint8_t *src, int sw, int sh, int sstride, size_t sframe, uint8_t *dst, int dw, int dh,
                                  int dstride, size_t dframe, int frames)
{
    const long long step = 2ll * sw * 1024;                  // numerator step of the source position per output pixel
    const int words = ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)dstride | (uintptr_t)dframe) & 3) == 0;
    const bool src_words = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)sstride | (uintptr_t)sframe) & 3) == 0;
    // word loads need 32-bit positions: (sw - 1) * 2048 and the carried remainders (1 020 * rs + dw) must fit
    const bool small = sw <= 65536 && dw <= 65536 && step / dw < (1ll << 20);
    if (src_words && small && !getenv("VSLAM_RESIZE_BYTES"))
        hipLaunchKernelGGL(resize_bilinear4w_kernel, dim3((dw + 1023) / 1024, dh, frames), dim3(256), 0, ctx->stream, src, sw, sh, sstride, sframe, dst, dw, dh,
                           dstride, dframe, (int)(step / dw), (int)(step % dw), words);
    else
        hipLaunchKernelGGL(resize_bilinear4_kernel, dim3((dw + 1023) / 1024, dh, frames), dim3(256), 0, ctx->stream, src, sw, sh, sstride, sframe, dst, dw, dh,
                           dstride, dframe, (int)(step / dw), (int)(step % dw), words);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

static size_t orb_ws_bytes(int frames) { return detect_ws_bytes(frames) + Arena::pad((size_t)frames * 2 * 4); }

static vslam_status launch_orb_level(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, size_t stride_frame, int frames,
                                     int nfeatures, int fast_threshold, int edge, int block, float harris_k,
                                     vslam_feature *out, int cap, int32_t *count, void *ws)
{
    Arena ar(ws);
    FastArgs a;
    a.gray = gray; a.stride_frame = stride_frame; a.w = w; a.h = h; a.stride = stride; a.threshold = fast_threshold; a.nonmax = 1;
    a.border = edge;
    a.cand = ar.take<uint64_t>((size_t)frames * 2 * CAND_CAP);
    a.cand_count = ar.take<int32_t>((size_t)frames * 2);
    int32_t *kept = ar.take<int32_t>((size_t)frames * 2);           // [0..frames): after the FAST cut, [frames..): after the Harris cut
    hipStream_t s = ctx->stream;
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(a.cand_count, 0, (size_t)frames * 2 * 4, s));
    a.tiles_x = (w + FT_W - 1) / FT_W; a.frames = frames;
    const size_t tiles = (size_t)a.tiles_x * ((h + FT_H - 1) / FT_H) * frames;
    if (tiles > 0x7fffffffu) return fail(ctx, VSLAM_ERR_CAPACITY, "orb_detect: too many tiles for one launch");
    hipLaunchKernelGGL(fast9_tile_kernel, dim3((unsigned)tiles), dim3(FT_T), 0, s, a);
    launch_sort_lists(s, a.cand, a.cand_count, frames, 1);
    hipLaunchKernelGGL(orb_cut_kernel, dim3(frames), dim3(1024), 0, s, a.cand, a.cand_count, 0, 2 * nfeatures, kept);
    OrbHarrisArgs ha;
    ha.gray = gray; ha.stride_frame = stride_frame; ha.w = w; ha.h = h; ha.stride = stride; ha.block = block; ha.k = harris_k;
    {
        volatile float sc = 1.0f / (4.0f * (float)block * 255.0f);   // scale * scale * scale * scale, left to right, one rounding each
        volatile float s2 = sc * sc;
        volatile float s3 = s2 * sc;
        volatile float s4 = s3 * sc;
        ha.scale4 = s4;
    }
    ha.cand = a.cand; ha.cand_count = a.cand_count; ha.kept = kept;
    const bool words = block == 7 && ((reinterpret_cast<uintptr_t>(gray) | (uintptr_t)stride | (uintptr_t)stride_frame) & 3) == 0;
    if (words) hipLaunchKernelGGL(orb_harris7_kernel, dim3(SORT_S / 256, frames), dim3(256), 0, s, ha);
    else hipLaunchKernelGGL(orb_harris_kernel, dim3(SORT_S / 256, frames), dim3(256), 0, s, ha);
    launch_sort_lists(s, a.cand, a.cand_count, frames, 2, /*one_chunk=*/true);      // at most 2 * nfeatures <= SORT_S keys
    hipLaunchKernelGGL(orb_cut_kernel, dim3(frames), dim3(1024), 0, s, a.cand, a.cand_count, 1, nfeatures, kept + frames);
    const int lim = cap < SORT_S ? cap : SORT_S;
    hipLaunchKernelGGL(orb_emit_kernel, dim3((lim + 255) / 256, frames), dim3(256), 0, s, a.cand, kept + frames, cap, out, count);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

static vslam_status check_orb_level(vslam_ctx *ctx, int w, int h, int stride, int nfeatures, int fast_threshold, int edge, int block)
{
    if (block < 3 || block > 15 || !(block & 1)) return fail(ctx, VSLAM_ERR_ARG, "orb_detect: Harris block %d must be odd in [3,15]", block);
    if (edge < block / 2 + 2) return fail(ctx, VSLAM_ERR_ARG, "orb_detect: edge threshold %d below the Harris block's reach %d", edge, block / 2 + 2);
    if (w < 2 * edge + 1 || h < 2 * edge + 1 || stride < w) return fail(ctx, VSLAM_ERR_SHAPE, "orb_detect: image %dx%d (stride %d) inside an edge of %d is empty", w, h, stride, edge);
    if (w > 65535 || h > 65535) return fail(ctx, VSLAM_ERR_CAPACITY, "orb_detect: image side exceeds 65535");
    if (fast_threshold < 0 || fast_threshold > 254) return fail(ctx, VSLAM_ERR_ARG, "orb_detect: FAST threshold %d outside [0,254]", fast_threshold);
    if (nfeatures < 0 || 2 * nfeatures > SORT_S) return fail(ctx, VSLAM_ERR_CAPACITY, "orb_detect: %d features per level exceed %d", nfeatures, SORT_S / 2);
    return VSLAM_OK;
}

static size_t tile_lds_bytes(int bs)
{
    const int hb = bs / 2, halo = hb + 2;
    const int IW = TW + 2 * halo, IH = TH + 2 * halo, GW = TW + 2 * (hb + 1), GH = TH + 2 * (hb + 1), RW = TW + 2;
    const int IP = IW | 1, GP = GW | 1, RP = RW | 1;                       // odd row pitches, as in the kernel
    size_t sums = (size_t)RP * GH * 4 * 3, imgb = (size_t)IP * IH * 4;     // the image tile overlays the sums
    size_t b = (size_t)GP * GH * 4 * 3 + (sums > imgb ? sums : imgb);
    return (b + 15) & ~size_t(15);
}

static vslam_status check_params(vslam_ctx *ctx, const vslam_corner_params *p, int w, int h, int stride, int detector)
{
    if (!p) return fail(ctx, VSLAM_ERR_ARG, "corner: null params");
    if (p->block_size < 3 || p->block_size > 2 * MAXHB + 1 || !(p->block_size & 1))
        return fail(ctx, VSLAM_ERR_ARG, "corner: block_size %d must be odd in [3,15]", p->block_size);
    if (p->border < p->block_size / 2 + 2) return fail(ctx, VSLAM_ERR_ARG, "corner: border %d < block_size/2+2", p->border);
    if (w < 2 * p->border + 1 || h < 2 * p->border + 1 || stride < w) return fail(ctx, VSLAM_ERR_SHAPE, "corner: image %dx%d (stride %d) too small", w, h, stride);
    if (w > 65535 || h > 65535) return fail(ctx, VSLAM_ERR_CAPACITY, "corner: image side exceeds 65535");
    if (detector < 0 || detector > 2) return fail(ctx, VSLAM_ERR_ARG, "corner: bad detector %d", detector);
    if (p->max_features < 1 || p->max_features > SORT_S) return fail(ctx, VSLAM_ERR_CAPACITY, "corner: max_features %d outside [1,%d]", p->max_features, SORT_S);
    return VSLAM_OK;
}

static CornerArgs make_args(const uint8_t *gray, int w, int h, int stride, size_t stride_frame, const vslam_corner_params *p)
{
    CornerArgs a;
    a.gray = gray; a.stride_frame = stride_frame; a.w = w; a.h = h; a.stride = stride;
    a.bs = p->block_size; a.border = p->border; a.k = p->harris_k; a.thr_h = p->harris_thresh; a.thr_s = p->shi_thresh;
    volatile float s = 1.0f / (4.0f * (float)p->block_size * 255.0f);
    volatile float s2 = s * s;
    volatile float s4 = s2 * s2;
    a.s2 = s2; a.s4 = s4;
    a.cand = nullptr; a.cand_count = nullptr; a.map_h = nullptr; a.map_s = nullptr;
    return a;
}

// Workspace per frame: 2 candidate lists + 2 counters.
static size_t detect_ws_bytes(int frames) { return Arena::pad((size_t)frames * 2 * CAND_CAP * 8) + Arena::pad((size_t)frames * 2 * 4); }

static vslam_status launch_detect(vslam_ctx *ctx, CornerArgs a, int frames, int detector, const vslam_corner_params *p,
                                  vslam_feature *out, int cap, int32_t *count, void *ws)
{
    Arena ar(ws);
    a.cand = ar.take<uint64_t>((size_t)frames * 2 * CAND_CAP);
    a.cand_count = ar.take<int32_t>((size_t)frames * 2);
    hipStream_t s = ctx->stream;
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(a.cand_count, 0, (size_t)frames * 2 * 4, s));
    a.tiles_x = (a.w + TW - 1) / TW; a.frames = frames;
    dim3 grid((unsigned)((size_t)a.tiles_x * ((a.h + TH - 1) / TH) * frames));
    size_t lds = tile_lds_bytes(a.bs);
    const int det_mask = detector == 2 ? 3 : (1 << detector);
    if (a.bs == 7 && !getenv("VSLAM_CORNER_TILE")) {
        const int nstrips = (a.w + SW_OUT - 1) / SW_OUT;
        // bands of rows: enough wavefronts to fill the chip, at least 64 rows each (11 extra steps per band)
        int target_waves = 8192;
        if (const char *e = getenv("VSLAM_STRIP_WAVES")) { const int v = atoi(e); if (v >= 64 && v <= (1 << 20)) target_waves = v; }      // A/B runs
        int nseg = (target_waves + nstrips * frames - 1) / (nstrips * frames);
        const int max_seg = (a.h + 63) / 64;
        if (nseg > max_seg) nseg = max_seg;
        if (nseg < 1) nseg = 1;
        const int seg_h = (a.h + nseg - 1) / nseg;
        dim3 sgrid((unsigned)((size_t)nstrips * nseg * frames));
        if (det_mask == 1) hipLaunchKernelGGL((corner_strip_kernel<1>), sgrid, dim3(64), 0, s, a, nstrips, nseg, seg_h);
        else if (det_mask == 2) hipLaunchKernelGGL((corner_strip_kernel<2>), sgrid, dim3(64), 0, s, a, nstrips, nseg, seg_h);
        else hipLaunchKernelGGL((corner_strip_kernel<3>), sgrid, dim3(64), 0, s, a, nstrips, nseg, seg_h);
    } else if (a.bs == 7) {
        if (det_mask == 1) hipLaunchKernelGGL((corner_tile_kernel<1, false, 7>), grid, dim3(256), lds, s, a);
        else if (det_mask == 2) hipLaunchKernelGGL((corner_tile_kernel<2, false, 7>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((corner_tile_kernel<3, false, 7>), grid, dim3(256), lds, s, a);
    } else {
        if (det_mask == 1) hipLaunchKernelGGL((corner_tile_kernel<1, false, 0>), grid, dim3(256), lds, s, a);
        else if (det_mask == 2) hipLaunchKernelGGL((corner_tile_kernel<2, false, 0>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((corner_tile_kernel<3, false, 0>), grid, dim3(256), lds, s, a);
    }
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    launch_sort_lists(s, a.cand, a.cand_count, frames, det_mask);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    const int ndet = detector == 2 ? 2 : 1, det_first = detector == 2 ? 0 : detector;
    const int lim = cap < p->max_features ? cap : p->max_features;
    hipLaunchKernelGGL(emit_features_kernel, dim3((lim + 255) / 256, frames, ndet), dim3(256), 0, s, a.cand, a.cand_count,
                       det_first, ndet, cap, p->max_features, out, count);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

static vslam_status launch_fast9(vslam_ctx *ctx, FastArgs a, int frames, int max_features, vslam_feature *out, int cap, int32_t *count, void *ws)
{
    Arena ar(ws);
    a.cand = ar.take<uint64_t>((size_t)frames * 2 * CAND_CAP);
    a.cand_count = ar.take<int32_t>((size_t)frames * 2);
    hipStream_t s = ctx->stream;
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(a.cand_count, 0, (size_t)frames * 2 * 4, s));
    a.tiles_x = (a.w + FT_W - 1) / FT_W; a.frames = frames;
    const size_t tiles = (size_t)a.tiles_x * ((a.h + FT_H - 1) / FT_H) * frames;
    if (tiles > 0x7fffffffu) return fail(ctx, VSLAM_ERR_CAPACITY, "fast9_detect: too many tiles for one launch");
    hipLaunchKernelGGL(fast9_tile_kernel, dim3((unsigned)tiles), dim3(FT_T), 0, s, a);
    launch_sort_lists(s, a.cand, a.cand_count, frames, 1);
    const int lim = cap < max_features ? cap : max_features;
    hipLaunchKernelGGL(emit_features_kernel, dim3((lim + 255) / 256, frames, 1), dim3(256), 0, s, a.cand, a.cand_count,
                       0, 1, cap, max_features, out, count);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

static vslam_status check_fast9(vslam_ctx *ctx, int w, int h, int stride, int threshold, int max_features)
{
    if (w < 7 || h < 7 || stride < w) return fail(ctx, VSLAM_ERR_SHAPE, "fast9_detect: image %dx%d (stride %d) too small", w, h, stride);
    if (w > 65535 || h > 65535) return fail(ctx, VSLAM_ERR_CAPACITY, "fast9_detect: image side exceeds 65535");
    if (threshold < 0 || threshold > 254) return fail(ctx, VSLAM_ERR_ARG, "fast9_detect: threshold %d outside [0,254]", threshold);
    if (max_features < 1 || max_features > SORT_S) return fail(ctx, VSLAM_ERR_CAPACITY, "fast9_detect: max_features %d outside [1,%d]", max_features, SORT_S);
    return VSLAM_OK;
}

static vslam_status ensure_lds_attr(vslam_ctx *ctx)
{
    // The attribute belongs to the function on the current device: one flag per context (a context is bound
    // to one device and is not used from two threads at once).
    if (ctx->corner_lds_attr_set) return VSLAM_OK;
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sort_chunk_kernel<SORT_S, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, SORT_S * 8));
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_parts_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SORT_S * 8));
    VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_chunks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SORT_S * 8));
    const int big = (int)tile_lds_bytes(2 * MAXHB + 1);
    const void *tile_kernels[] = {
        reinterpret_cast<const void *>(&corner_tile_kernel<1, false, 0>), reinterpret_cast<const void *>(&corner_tile_kernel<2, false, 0>),
        reinterpret_cast<const void *>(&corner_tile_kernel<3, false, 0>), reinterpret_cast<const void *>(&corner_tile_kernel<3, true, 0>),
        reinterpret_cast<const void *>(&corner_tile_kernel<1, false, 7>), reinterpret_cast<const void *>(&corner_tile_kernel<2, false, 7>),
        reinterpret_cast<const void *>(&corner_tile_kernel<3, false, 7>), reinterpret_cast<const void *>(&corner_tile_kernel<3, true, 7>)};
    for (const void *kfn : tile_kernels) VSLAM_HIP_TRY(ctx, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ctx->corner_lds_attr_set = true;
    return VSLAM_OK;
}

// Counts the non-negative finite floats x for which sqrtf(x) differs from the correctly rounded
// square root (float)sqrt((double)x) (exact: 53 >= 2*24 + 2 bits), plus those among x = 0 and x >= 2^-64 for which
// sqrt_rn_normal(x), the form the sliding-window detector uses on its sums of squares, does.
__global__ void sqrtf_check_kernel(unsigned long long *bad)
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine = 0;
    for (unsigned long long u = tid; u < 0x7f800000ull; u += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)u);
        const unsigned exact = __float_as_uint((float)sqrt((double)x));
        mine += __float_as_uint(sqrtf(x)) != exact;
        if (u == 0 || u >= 0x1f800000ull) mine += __float_as_uint(sqrt_rn_normal(x)) != exact;
    }
    if (mine) atomicAdd(bad, mine);
}

}  // namespace vslam

using namespace vslam;

// Test hook (not part of the public header): the Shi-Tomasi response relies on sqrtf being correctly rounded.
extern "C" vslam_status vslam_debug_sqrtf_mismatches(vslam_ctx *ctx, unsigned long long *count)
{
    if (!ctx || !count) return VSLAM_ERR_ARG;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ws_reserve(ctx, 256));
    unsigned long long *d = static_cast<unsigned long long *>(ctx->ws);
    VSLAM_HIP_TRY(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    hipLaunchKernelGGL(sqrtf_check_kernel, dim3(4096), dim3(256), 0, ctx->stream, d);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    VSLAM_TRY(d2h(ctx, count, d, 8));
    return stream_finish(ctx);
}

extern "C" vslam_status vslam_corner_detect_batch_dev(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                                                      size_t stride_frame, int frames, const vslam_corner_params *params,
                                                      int detector, vslam_feature *out, int cap, int32_t *count)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !out || !count || cap < 1) return fail(ctx, VSLAM_ERR_ARG, "corner_detect: null pointer or cap < 1");
    if (frames < 1 || frames > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "corner_detect: frames=%d out of range", frames);
    VSLAM_TRY(check_params(ctx, params, w, h, stride, detector));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ensure_lds_attr(ctx));
    VSLAM_TRY(ws_reserve(ctx, detect_ws_bytes(frames)));
    CornerArgs a = make_args(gray, w, h, stride, stride_frame, params);
    return launch_detect(ctx, a, frames, detector, params, out, cap, count, ctx->ws);
}

extern "C" vslam_status vslam_corner_detect(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                                            const vslam_corner_params *params, int detector,
                                            vslam_feature *out, int cap, int32_t *count)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !out || !count || cap < 1) return fail(ctx, VSLAM_ERR_ARG, "corner_detect: null pointer or cap < 1");
    VSLAM_TRY(check_params(ctx, params, w, h, stride, detector));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ensure_lds_attr(ctx));
    const int ndet = detector == 2 ? 2 : 1;
    const size_t img_bytes = (size_t)stride * h, out_bytes = (size_t)ndet * cap * sizeof(vslam_feature);
    VSLAM_TRY(ws_reserve(ctx, detect_ws_bytes(1) + Arena::pad(img_bytes) + Arena::pad(out_bytes) + Arena::pad(8)));
    Arena ar(ctx->ws);
    void *dws = ar.take<char>(detect_ws_bytes(1));
    uint8_t *dimg = ar.take<uint8_t>(img_bytes);
    vslam_feature *dout = ar.take<vslam_feature>((size_t)ndet * cap);
    int32_t *dcount = ar.take<int32_t>(2);
    VSLAM_TRY(h2d(ctx, dimg, gray, img_bytes));
    CornerArgs a = make_args(dimg, w, h, stride, 0, params);
    VSLAM_TRY(launch_detect(ctx, a, 1, detector, params, dout, cap, dcount, dws));
    int32_t hc[2] = {0, 0};
    VSLAM_TRY(d2h(ctx, hc, dcount, ndet * 4));
    VSLAM_TRY(stream_sync(ctx));
    for (int d = 0; d < ndet; ++d) {
        if (hc[d] < 0) return fail(ctx, VSLAM_ERR_CAPACITY, "corner_detect: more than %d candidates survive NMS; raise the threshold", CAND_CAP);
        count[d] = hc[d];
        if (hc[d] > 0)
            VSLAM_HIP_TRY(ctx, hipMemcpy(out + (size_t)d * cap, dout + (size_t)d * cap, (size_t)hc[d] * sizeof(vslam_feature), hipMemcpyDeviceToHost));
    }
    return VSLAM_OK;
}

extern "C" vslam_status vslam_fast9_detect_batch_dev(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, size_t stride_frame,
                                                     int frames, int threshold, int nonmax, int max_features,
                                                     vslam_feature *out, int cap, int32_t *count)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !out || !count || cap < 1) return fail(ctx, VSLAM_ERR_ARG, "fast9_detect: null pointer or cap < 1");
    if (frames < 1 || frames > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "fast9_detect: frames=%d out of range", frames);
    VSLAM_TRY(check_fast9(ctx, w, h, stride, threshold, max_features));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ensure_lds_attr(ctx));
    VSLAM_TRY(ws_reserve(ctx, detect_ws_bytes(frames)));
    FastArgs a;
    a.gray = gray; a.stride_frame = stride_frame; a.w = w; a.h = h; a.stride = stride; a.threshold = threshold; a.nonmax = nonmax ? 1 : 0;
    return launch_fast9(ctx, a, frames, max_features, out, cap, count, ctx->ws);
}

extern "C" vslam_status vslam_fast9_detect(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, int threshold, int nonmax,
                                           int max_features, vslam_feature *out, int cap, int32_t *count)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !out || !count || cap < 1) return fail(ctx, VSLAM_ERR_ARG, "fast9_detect: null pointer or cap < 1");
    VSLAM_TRY(check_fast9(ctx, w, h, stride, threshold, max_features));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ensure_lds_attr(ctx));
    const size_t img_bytes = (size_t)stride * h, out_bytes = (size_t)cap * sizeof(vslam_feature);
    VSLAM_TRY(ws_reserve(ctx, detect_ws_bytes(1) + Arena::pad(img_bytes) + Arena::pad(out_bytes) + Arena::pad(8)));
    Arena ar(ctx->ws);
    void *dws = ar.take<char>(detect_ws_bytes(1));
    uint8_t *dimg = ar.take<uint8_t>(img_bytes);
    vslam_feature *dout = ar.take<vslam_feature>((size_t)cap);
    int32_t *dcount = ar.take<int32_t>(2);
    VSLAM_TRY(h2d(ctx, dimg, gray, img_bytes));
    FastArgs a;
    a.gray = dimg; a.stride_frame = 0; a.w = w; a.h = h; a.stride = stride; a.threshold = threshold; a.nonmax = nonmax ? 1 : 0;
    VSLAM_TRY(launch_fast9(ctx, a, 1, max_features, dout, cap, dcount, dws));
    int32_t hc = 0;
    VSLAM_TRY(d2h(ctx, &hc, dcount, 4));
    VSLAM_TRY(stream_sync(ctx));
    if (hc < 0) return fail(ctx, VSLAM_ERR_CAPACITY, "fast9_detect: more than %d corners; raise the threshold", CAND_CAP);
    *count = hc;
    if (hc > 0) VSLAM_HIP_TRY(ctx, hipMemcpy(out, dout, (size_t)hc * sizeof(vslam_feature), hipMemcpyDeviceToHost));
    return VSLAM_OK;
}

// ---- ORB keypoint detection (no orientation, no descriptors) ----
extern "C" vslam_status vslam_orb_detect_level_batch_dev(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, size_t stride_frame,
                                                         int frames, int nfeatures, int fast_threshold, int edge_threshold, int harris_block,
                                                         float harris_k, vslam_feature *out, int cap, int32_t *count)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !out || !count || cap < 1) return fail(ctx, VSLAM_ERR_ARG, "orb_detect: null pointer or cap < 1");
    if (frames < 1 || frames > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "orb_detect: frames=%d out of range", frames);
    VSLAM_TRY(check_orb_level(ctx, w, h, stride, nfeatures, fast_threshold, edge_threshold, harris_block));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ensure_lds_attr(ctx));
    VSLAM_TRY(ws_reserve(ctx, orb_ws_bytes(frames)));
    return launch_orb_level(ctx, gray, w, h, stride, stride_frame, frames, nfeatures, fast_threshold, edge_threshold, harris_block, harris_k,
                            out, cap, count, ctx->ws);
}

extern "C" vslam_status vslam_resize_bilinear_dev(vslam_ctx *ctx, const uint8_t *src, int sw, int sh, int sstride,
                                                  uint8_t *dst, int dw, int dh, int dstride)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!src || !dst) return fail(ctx, VSLAM_ERR_ARG, "resize: null pointer");
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1 || sstride < sw || dstride < dw || sw > 65535 || sh > 65535 || dw > 65535 || dh > 65535)
        return fail(ctx, VSLAM_ERR_SHAPE, "resize: %dx%d (stride %d) -> %dx%d (stride %d)", sw, sh, sstride, dw, dh, dstride);
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(launch_resize(ctx, src, sw, sh, sstride, (size_t)0, dst, dw, dh, dstride, (size_t)0, 1));
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

// cvRound: to nearest, ties to even
static int round_half_even(double v) { return (int)std::nearbyint(v); }

extern "C" vslam_status vslam_orb_level_plan(const vslam_orb_params *p, int w, int h, int32_t *level_w, int32_t *level_h,
                                             int32_t *level_features, float *level_scale)
{
    if (!p || !level_w || !level_h || !level_features || !level_scale) return VSLAM_ERR_ARG;
    if (p->nlevels < 1 || p->nlevels > VSLAM_ORB_MAX_LEVELS || !(p->scale_factor > 1.0f) || p->nfeatures < 0 || w < 1 || h < 1) return VSLAM_ERR_ARG;
    const float factor = (float)(1.0 / (double)p->scale_factor);
    float desired = (float)p->nfeatures * (1.0f - factor) / (1.0f - (float)std::pow((double)factor, (double)p->nlevels));
    int sum = 0;
    for (int l = 0; l < p->nlevels; ++l) {
        const float sc = (float)std::pow((double)p->scale_factor, (double)l);
        level_scale[l] = sc;
        level_w[l] = round_half_even((double)((float)w / sc));
        level_h[l] = round_half_even((double)((float)h / sc));
        if (l < p->nlevels - 1) {
            level_features[l] = round_half_even((double)desired);
            sum += level_features[l];
            desired *= factor;
        } else {
            level_features[l] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
        }
    }
    return VSLAM_OK;
}

// All levels of one host image.  With desc != nullptr the keypoints of every level are also oriented and described
// on that level (pattern: host table or nullptr for the library's own).
static vslam_status orb_run(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, const vslam_orb_params *p, const int8_t *pattern,
                            vslam_orb_keypoint *out, float *angle, uint8_t *desc, int cap, int32_t *count, int32_t *level_counts)
{
    const bool describe = desc != nullptr;
    int32_t lw[VSLAM_ORB_MAX_LEVELS], lh[VSLAM_ORB_MAX_LEVELS], lf[VSLAM_ORB_MAX_LEVELS];
    float ls[VSLAM_ORB_MAX_LEVELS];
    if (vslam_orb_level_plan(p, w, h, lw, lh, lf, ls) != VSLAM_OK) return fail(ctx, VSLAM_ERR_ARG, "orb_detect: bad pyramid parameters");
    if (stride < w) return fail(ctx, VSLAM_ERR_SHAPE, "orb_detect: stride %d < width %d", stride, w);
    int8_t own_pattern[VSLAM_ORB_PATTERN_BYTES];
    if (describe) {
        if (!pattern) { orb_default_pattern(own_pattern); pattern = own_pattern; }
        else if (!orb_pattern_ok(pattern)) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute: pattern coordinates must lie in [-15, 15]");
        if (p->edge_threshold < 23) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute: edge_threshold %d < 23 (reach of the rotated pattern)", p->edge_threshold);
    }
    int nl = p->nlevels;
    while (nl > 0 && (lw[nl - 1] < 2 * p->edge_threshold + 1 || lh[nl - 1] < 2 * p->edge_threshold + 1)) --nl;   // levels with nothing inside the edge
    for (int l = 0; l < nl; ++l) VSLAM_TRY(check_orb_level(ctx, lw[l], lh[l], lw[l], lf[l], p->fast_threshold, p->edge_threshold, p->harris_block));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ensure_lds_attr(ctx));
    // rows of the library's own level images and of the blurred copy start on 16-byte boundaries (the kernels then
    // move whole words); the bytes between a row's end and the next row's start are never used
    int lsd[VSLAM_ORB_MAX_LEVELS];
    for (int l = 0; l < nl; ++l) lsd[l] = (lw[l] + 15) & ~15;
    const int bsd = (w + 15) & ~15;
    size_t pyr = 0;
    for (int l = 0; l < nl; ++l) pyr += Arena::pad((size_t)lsd[l] * lh[l]);
    const size_t feat_bytes = (size_t)SORT_S * sizeof(vslam_feature);
    const size_t desc_bytes = describe ? (size_t)nl * (Arena::pad((size_t)SORT_S * 4) + Arena::pad((size_t)SORT_S * 32)) + Arena::pad((size_t)bsd * h) + Arena::pad(VSLAM_ORB_PATTERN_BYTES) : 0;
    VSLAM_TRY(ws_reserve(ctx, orb_ws_bytes(1) + Arena::pad((size_t)stride * h) + pyr + (size_t)nl * Arena::pad(feat_bytes) + Arena::pad(64) + desc_bytes));
    Arena ar(ctx->ws);
    void *dws = ar.take<char>(orb_ws_bytes(1));
    uint8_t *dbase = ar.take<uint8_t>((size_t)stride * h);
    uint8_t *dlev[VSLAM_ORB_MAX_LEVELS];
    vslam_feature *dfeat[VSLAM_ORB_MAX_LEVELS];
    float *dangle[VSLAM_ORB_MAX_LEVELS];
    uint8_t *ddesc[VSLAM_ORB_MAX_LEVELS];
    for (int l = 0; l < nl; ++l) dlev[l] = ar.take<uint8_t>((size_t)lsd[l] * lh[l]);
    for (int l = 0; l < nl; ++l) dfeat[l] = ar.take<vslam_feature>(SORT_S);
    int32_t *dcount = ar.take<int32_t>(16);
    uint8_t *dblur = nullptr;
    int8_t *dpattern = nullptr;
    if (describe) {
        for (int l = 0; l < nl; ++l) { dangle[l] = ar.take<float>(SORT_S); ddesc[l] = ar.take<uint8_t>((size_t)SORT_S * 32); }
        dblur = ar.take<uint8_t>((size_t)bsd * h);
        dpattern = ar.take<int8_t>(VSLAM_ORB_PATTERN_BYTES);
        VSLAM_TRY(h2d(ctx, dpattern, pattern, VSLAM_ORB_PATTERN_BYTES));
    }
    VSLAM_TRY(h2d(ctx, dbase, gray, (size_t)stride * h));
    for (int l = 0; l < nl; ++l) {
        // level l is resized from level l-1 (level 0 is the image itself)
        const uint8_t *img = dbase;
        int iw = w, ih = h, is = stride;
        if (l > 0) {
            const uint8_t *prev = l == 1 ? dbase : dlev[l - 1];
            const int pw = l == 1 ? w : lw[l - 1], ph = l == 1 ? h : lh[l - 1], ps = l == 1 ? stride : lsd[l - 1];
            VSLAM_TRY(launch_resize(ctx, prev, pw, ph, ps, (size_t)0, dlev[l], lw[l], lh[l], lsd[l], (size_t)0, 1));
            img = dlev[l]; iw = lw[l]; ih = lh[l]; is = lsd[l];
        }
        VSLAM_TRY(launch_orb_level(ctx, img, iw, ih, is, 0, 1, lf[l], p->fast_threshold, p->edge_threshold, p->harris_block, p->harris_k,
                                   dfeat[l], SORT_S, dcount + l, dws));
        if (describe) {                                 // the blurred copy is reused level after level (stream order)
            const int bs_l = (iw + 15) & ~15;
            VSLAM_TRY(launch_gauss7(ctx, img, iw, ih, is, 0, dblur, bs_l, 0, 1));
            VSLAM_TRY(launch_orb_describe(ctx, img, is, 0, dblur, bs_l, 0, iw, ih, 1, dfeat[l], dcount + l, SORT_S, dpattern, dangle[l], ddesc[l]));
        }
    }
    int32_t hc[16] = {0};
    VSLAM_TRY(d2h(ctx, hc, dcount, sizeof(int32_t) * 16));
    VSLAM_TRY(stream_sync(ctx));
    int total = 0;
    std::vector<vslam_feature> tmp;
    for (int l = 0; l < p->nlevels; ++l) {
        const int k = l < nl ? hc[l] : 0;
        if (k < 0) return fail(ctx, VSLAM_ERR_CAPACITY, "orb_detect: level %d has more candidates (or ties at the cut) than %d", l, SORT_S);
        if (level_counts) level_counts[l] = k;
        if (k == 0) continue;
        if (total + k > cap) return fail(ctx, VSLAM_ERR_CAPACITY, "orb_detect: more than %d keypoints", cap);
        tmp.resize((size_t)k);
        VSLAM_HIP_TRY(ctx, hipMemcpy(tmp.data(), dfeat[l], (size_t)k * sizeof(vslam_feature), hipMemcpyDeviceToHost));
        if (describe) {
            VSLAM_HIP_TRY(ctx, hipMemcpy(angle + total, dangle[l], (size_t)k * sizeof(float), hipMemcpyDeviceToHost));
            VSLAM_HIP_TRY(ctx, hipMemcpy(desc + (size_t)total * 32, ddesc[l], (size_t)k * 32, hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < k; ++i) {
            vslam_orb_keypoint kp;
            kp.x = l == 0 ? (float)tmp[(size_t)i].x : (float)tmp[(size_t)i].x * ls[l];      // pt *= scale, float
            kp.y = l == 0 ? (float)tmp[(size_t)i].y : (float)tmp[(size_t)i].y * ls[l];
            kp.response = tmp[(size_t)i].score; kp.octave = l;
            out[total++] = kp;
        }
    }
    *count = total;
    return VSLAM_OK;
}

extern "C" vslam_status vslam_orb_detect(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, const vslam_orb_params *p,
                                         vslam_orb_keypoint *out, int cap, int32_t *count, int32_t *level_counts)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !p || !out || !count || cap < 1) return fail(ctx, VSLAM_ERR_ARG, "orb_detect: null pointer or cap < 1");
    return orb_run(ctx, gray, w, h, stride, p, nullptr, out, nullptr, nullptr, cap, count, level_counts);
}

extern "C" vslam_status vslam_orb_detect_and_compute(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, const vslam_orb_params *p,
                                                     const int8_t *pattern, vslam_orb_keypoint *out, float *angle, uint8_t *desc, int cap,
                                                     int32_t *count, int32_t *level_counts)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !p || !out || !angle || !desc || !count || cap < 1) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute: null pointer or cap < 1");
    return orb_run(ctx, gray, w, h, stride, p, pattern, out, angle, desc, cap, count, level_counts);
}

// ---- the whole ORB front end for a batch of device-resident frames ----
struct OrbPackArgs {
    const vslam_feature *feat[VSLAM_ORB_MAX_LEVELS];      // [frames][cap_l] per level
    const float *angle[VSLAM_ORB_MAX_LEVELS];
    const uint8_t *desc[VSLAM_ORB_MAX_LEVELS];
    int cap_l[VSLAM_ORB_MAX_LEVELS];
    float scale[VSLAM_ORB_MAX_LEVELS];
    const int32_t *level_count;                           // [level][frames]
    int nl, nlevels_out, frames, cap;
    vslam_orb_keypoint *out; float *out_angle; uint8_t *out_desc;
    int32_t *count, *level_counts;                        // [frames], optional [frames][nlevels_out]
};

// The levels' keypoint lists of a frame one after the other (level 0 first), positions scaled to the base image as
// cv::ORB reports them (pt *= scale, in float).  A level that overflowed, or more keypoints than `cap`, gives count -1
// for the frame.  Grid (frame, level, part): every workgroup works out the frame's offsets for itself and copies one
// part of one level (a workgroup per frame left three quarters of the chip idle).
constexpr int ORB_PACK_PARTS = 4;
__global__ __launch_bounds__(256) void orb_pack_kernel(OrbPackArgs a)
{
    __shared__ int off[VSLAM_ORB_MAX_LEVELS + 1];
    const int f = blockIdx.x, lv = blockIdx.y, part = blockIdx.z, tid = threadIdx.x;
    if (tid == 0) {
        int total = 0;
        bool bad = false;
        for (int l = 0; l < a.nl; ++l) {
            const int k = a.level_count[(size_t)l * a.frames + f];
            off[l] = total;
            if (k < 0 || k > a.cap_l[l]) bad = true; else total += k;
        }
        off[a.nl] = total;
        if (total > a.cap) bad = true;
        if (lv == 0 && part == 0) {
            a.count[f] = bad ? -1 : total;
            if (a.level_counts)
                for (int l = 0; l < a.nlevels_out; ++l) a.level_counts[(size_t)f * a.nlevels_out + l] = l < a.nl ? a.level_count[(size_t)l * a.frames + f] : 0;
        }
        if (bad) off[a.nl] = -1;
    }
    __syncthreads();
    if (off[a.nl] < 0 || lv >= a.nl) return;
    {
        const int l = lv;
        const int k = off[l + 1] - off[l];
        const int per = (k + ORB_PACK_PARTS - 1) / ORB_PACK_PARTS, i0 = part * per, i1 = min(k, i0 + per);
        const vslam_feature *src = a.feat[l] + (size_t)f * a.cap_l[l];
        vslam_orb_keypoint *dst = a.out + (size_t)f * a.cap + off[l];
        for (int i = i0 + tid; i < i1; i += 256) {
            vslam_orb_keypoint kp;
            kp.x = l == 0 ? (float)src[i].x : __fmul_rn((float)src[i].x, a.scale[l]);
            kp.y = l == 0 ? (float)src[i].y : __fmul_rn((float)src[i].y, a.scale[l]);
            kp.response = src[i].score; kp.octave = l;
            dst[i] = kp;
        }
        if (a.out_desc) {
            const float *sa = a.angle[l] + (size_t)f * a.cap_l[l];
            float *da = a.out_angle + (size_t)f * a.cap + off[l];
            for (int i = i0 + tid; i < i1; i += 256) da[i] = sa[i];
            const uint32_t *sd = reinterpret_cast<const uint32_t *>(a.desc[l] + (size_t)f * a.cap_l[l] * 32);
            uint32_t *dd = reinterpret_cast<uint32_t *>(a.out_desc + ((size_t)f * a.cap + off[l]) * 32);
            for (int i = 8 * i0 + tid; i < 8 * i1; i += 256) dd[i] = sd[i];
        }
    }
}

extern "C" vslam_status vslam_orb_detect_and_compute_batch_dev(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, size_t stride_frame,
                                                               int frames, const vslam_orb_params *p, const int8_t *pattern,
                                                               vslam_orb_keypoint *out, float *angle, uint8_t *desc, int cap, int32_t *count,
                                                               int32_t *level_counts)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray || !p || !out || !count || cap < 1) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute_batch: null pointer or cap < 1");
    if ((angle == nullptr) != (desc == nullptr)) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute_batch: angle and desc go together");
    if (frames < 1 || frames > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "orb_detect_and_compute_batch: frames=%d out of range", frames);
    const bool describe = desc != nullptr;
    if (describe && (reinterpret_cast<uintptr_t>(desc) & 3)) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute_batch: desc must be 4-byte aligned");
    if (pattern && (reinterpret_cast<uintptr_t>(pattern) & 15)) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute_batch: the pattern must be 16-byte aligned");
    int32_t lw[VSLAM_ORB_MAX_LEVELS], lh[VSLAM_ORB_MAX_LEVELS], lf[VSLAM_ORB_MAX_LEVELS];
    float ls[VSLAM_ORB_MAX_LEVELS];
    if (vslam_orb_level_plan(p, w, h, lw, lh, lf, ls) != VSLAM_OK) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute_batch: bad pyramid parameters");
    if (stride < w || (frames > 1 && stride_frame < (size_t)stride * (h - 1) + w))
        return fail(ctx, VSLAM_ERR_SHAPE, "orb_detect_and_compute_batch: stride %d < width %d or frames overlap (frame stride %zu)", stride, w, stride_frame);
    if (describe && p->edge_threshold < 23) return fail(ctx, VSLAM_ERR_ARG, "orb_detect_and_compute_batch: edge_threshold %d < 23 (reach of the rotated pattern)", p->edge_threshold);
    int nl = p->nlevels;
    while (nl > 0 && (lw[nl - 1] < 2 * p->edge_threshold + 1 || lh[nl - 1] < 2 * p->edge_threshold + 1)) --nl;
    for (int l = 0; l < nl; ++l) VSLAM_TRY(check_orb_level(ctx, lw[l], lh[l], lw[l], lf[l], p->fast_threshold, p->edge_threshold, p->harris_block));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ensure_lds_attr(ctx));
    // per level room for the budget plus ties at the cut
    OrbPackArgs pa;
    // The levels are independent once their images exist: each level's chain of kernels (FAST, sorts, cuts, Harris, blur,
    // descriptors -- most of them far too small to fill the chip) runs on one of three streams beside the others, each with
    // candidate lists and a blurred image of its own, while the main stream goes on down the pyramid.
    const bool beside = nl > 1 && frames * (size_t)w * h <= ((size_t)1 << 28) && !getenv("VSLAM_ORB_SERIAL");
    const int nws = beside ? nl : 1;
    size_t bytes = nws * orb_ws_bytes(frames) + Arena::pad((size_t)VSLAM_ORB_MAX_LEVELS * frames * 4) + Arena::pad(VSLAM_ORB_PATTERN_BYTES);
    int lsd[VSLAM_ORB_MAX_LEVELS];                       // row strides of the library's own level images: multiples of 16 bytes
    for (int l = 0; l < nl; ++l) lsd[l] = (lw[l] + 15) & ~15;
    const int bsd = (w + 15) & ~15;
    for (int l = 0; l < nl; ++l) {
        pa.cap_l[l] = std::min(SORT_S, lf[l] + 1024);
        if (l > 0) bytes += Arena::pad((size_t)frames * lsd[l] * lh[l]);
        bytes += Arena::pad((size_t)frames * pa.cap_l[l] * sizeof(vslam_feature));
        if (describe) bytes += Arena::pad((size_t)frames * pa.cap_l[l] * 4) + Arena::pad((size_t)frames * pa.cap_l[l] * 32);
    }
    if (describe) bytes += Arena::pad((size_t)frames * bsd * h);
    if (describe && beside) for (int l = 1; l < nl; ++l) bytes += Arena::pad((size_t)frames * lsd[l] * lh[l]);
    VSLAM_TRY(ws_reserve(ctx, bytes));
    Arena ar(ctx->ws);
    void *dws[VSLAM_ORB_MAX_LEVELS];
    for (int l = 0; l < nws; ++l) dws[l] = ar.take<char>(orb_ws_bytes(frames));
    int32_t *dcount = ar.take<int32_t>((size_t)VSLAM_ORB_MAX_LEVELS * frames);
    int8_t *dpattern = ar.take<int8_t>(VSLAM_ORB_PATTERN_BYTES);
    uint8_t *dlev[VSLAM_ORB_MAX_LEVELS] = {nullptr};
    vslam_feature *dfeat[VSLAM_ORB_MAX_LEVELS];
    float *dangle[VSLAM_ORB_MAX_LEVELS] = {nullptr};
    uint8_t *ddesc[VSLAM_ORB_MAX_LEVELS] = {nullptr};
    for (int l = 0; l < nl; ++l) {
        if (l > 0) dlev[l] = ar.take<uint8_t>((size_t)frames * lsd[l] * lh[l]);
        dfeat[l] = ar.take<vslam_feature>((size_t)frames * pa.cap_l[l]);
        if (describe) { dangle[l] = ar.take<float>((size_t)frames * pa.cap_l[l]); ddesc[l] = ar.take<uint8_t>((size_t)frames * pa.cap_l[l] * 32); }
    }
    uint8_t *dblur[VSLAM_ORB_MAX_LEVELS] = {nullptr};
    if (describe) {
        dblur[0] = ar.take<uint8_t>((size_t)frames * bsd * h);
        for (int l = 1; l < nl; ++l) dblur[l] = beside ? ar.take<uint8_t>((size_t)frames * lsd[l] * lh[l]) : dblur[0];
    }
    hipStream_t s = ctx->stream;
    if (beside) VSLAM_TRY(level_streams_init(ctx));
    if (describe && !pattern) {
        int8_t own[VSLAM_ORB_PATTERN_BYTES];
        orb_default_pattern(own);
        VSLAM_TRY(h2d(ctx, dpattern, own, VSLAM_ORB_PATTERN_BYTES));
        pattern = dpattern;
    }
    int queued = 0;                                           // chains queued on the level streams
    vslam_status level_status = VSLAM_OK;
    for (int l = 0; l < nl; ++l) {
        const uint8_t *img = gray;
        int iw = w, ih = h, is = stride;
        size_t ifr = stride_frame;
        if (l > 0) {
            const uint8_t *prev = l == 1 ? gray : dlev[l - 1];
            const int pw = l == 1 ? w : lw[l - 1], ph = l == 1 ? h : lh[l - 1], ps = l == 1 ? stride : lsd[l - 1];
            const size_t pf = l == 1 ? stride_frame : (size_t)lsd[l - 1] * lh[l - 1];
            ifr = (size_t)lsd[l] * lh[l];
            const vslam_status rs = launch_resize(ctx, prev, pw, ph, ps, pf, dlev[l], lw[l], lh[l], lsd[l], ifr, frames);
            if (rs != VSLAM_OK) { level_status = rs; break; }
            img = dlev[l]; iw = lw[l]; ih = lh[l]; is = lsd[l];
        }
        // the level's chain: on the main stream, or on a stream of its own behind "image ready" and in front of "level done"
        hipStream_t chain = s;
        if (beside) {
            chain = ctx->lvl_stream[l % 3];
            VSLAM_HIP_TRY(ctx, hipEventRecord(ctx->lvl_ev[2 * l], s));
            VSLAM_HIP_TRY(ctx, hipStreamWaitEvent(chain, ctx->lvl_ev[2 * l], 0));
        }
        ctx->stream = chain;                                  // the launchers below queue on the context's stream
        vslam_status st = launch_orb_level(ctx, img, iw, ih, is, ifr, frames, lf[l], p->fast_threshold, p->edge_threshold, p->harris_block, p->harris_k,
                                           dfeat[l], pa.cap_l[l], dcount + (size_t)l * frames, dws[beside ? l : 0]);
        if (st == VSLAM_OK && describe) {
            const int bs_l = (iw + 15) & ~15;
            st = launch_gauss7(ctx, img, iw, ih, is, ifr, dblur[l], bs_l, (size_t)bs_l * ih, frames);
            if (st == VSLAM_OK)
                st = launch_orb_describe(ctx, img, is, ifr, dblur[l], bs_l, (size_t)bs_l * ih, iw, ih, frames, dfeat[l], dcount + (size_t)l * frames,
                                         pa.cap_l[l], pattern, dangle[l], ddesc[l]);
        }
        ctx->stream = s;
        if (beside) { (void)hipEventRecord(ctx->lvl_ev[2 * l + 1], chain); ++queued; }
        if (st != VSLAM_OK) { level_status = st; break; }
        pa.feat[l] = dfeat[l]; pa.angle[l] = dangle[l]; pa.desc[l] = ddesc[l]; pa.scale[l] = ls[l];
    }
    for (int l = 0; l < queued; ++l) (void)hipStreamWaitEvent(s, ctx->lvl_ev[2 * l + 1], 0);      // the main stream joins every chain, also after an error
    VSLAM_TRY(level_status);
    pa.level_count = dcount; pa.nl = nl; pa.nlevels_out = p->nlevels; pa.frames = frames; pa.cap = cap;
    pa.out = out; pa.out_angle = angle; pa.out_desc = desc; pa.count = count; pa.level_counts = level_counts;
    hipLaunchKernelGGL(orb_pack_kernel, dim3(frames, pa.nl > 0 ? pa.nl : 1, ORB_PACK_PARTS), dim3(256), 0, s, pa);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

extern "C" vslam_status vslam_corner_response(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                                              const vslam_corner_params *params, float *harris, float *shi)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!gray) return fail(ctx, VSLAM_ERR_ARG, "corner_response: null image");
    VSLAM_TRY(check_params(ctx, params, w, h, stride, 2));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_TRY(ensure_lds_attr(ctx));
    const size_t img_bytes = (size_t)stride * h, map_bytes = (size_t)w * h * 4;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(img_bytes) + 2 * Arena::pad(map_bytes)));
    Arena ar(ctx->ws);
    uint8_t *dimg = ar.take<uint8_t>(img_bytes);
    float *dh = ar.take<float>((size_t)w * h), *ds = ar.take<float>((size_t)w * h);
    hipStream_t s = ctx->stream;
    VSLAM_TRY(h2d(ctx, dimg, gray, img_bytes));
    CornerArgs a = make_args(dimg, w, h, stride, 0, params);
    a.map_h = dh; a.map_s = ds;
    a.tiles_x = (w + TW - 1) / TW; a.frames = 1;
    dim3 grid((unsigned)(a.tiles_x * ((h + TH - 1) / TH)));
    if (a.bs == 7) hipLaunchKernelGGL((corner_tile_kernel<3, true, 7>), grid, dim3(256), tile_lds_bytes(a.bs), s, a);
    else hipLaunchKernelGGL((corner_tile_kernel<3, true, 0>), grid, dim3(256), tile_lds_bytes(a.bs), s, a);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    if (harris) VSLAM_TRY(d2h(ctx, harris, dh, map_bytes));
    if (shi) VSLAM_TRY(d2h(ctx, shi, ds, map_bytes));
    VSLAM_TRY(stream_sync(ctx));
    return VSLAM_OK;
}
