// Caller-side selection of new landmarks: the loop of reference src/SLAM.cpp:449-472 over the features
// sorted by score, with pixelDistance (src/SLAM.cpp:746-760; tests: test/src/pixelDistance.cpp).
// A feature becomes a landmark iff capacity is left, its score exceeds the threshold, it lies inside the
// image margins and no landmark -- already in the map or taken earlier in the same loop -- is closer than
// the pixel radius.  The loop is greedy and order dependent; here one workgroup per frame walks the
// features in chunks of 256: every thread tests its feature against the map and against the landmarks
// taken so far (independent work), then the chunk is resolved in order -- the first surviving feature is
// taken, the later ones of the chunk are tested against it, and so on -- which needs a few barriers per
// taken landmark instead of one step per feature.  The map is staged in LDS and tested in full by every
// thread (no early exit: a loop without a carried dependency is several times faster than one that stops at
// the first landmark in range).
// Distances are formed as the reference forms them: float differences, float products and sum (this file
// is compiled without FMA contraction), correctly rounded float square root, comparison with the double
// threshold.  The loop index that pixelDistance leaves uninitialised is taken as 0.
#include "common.hpp"

#include <climits>
#include <cmath>

namespace vslam {

constexpr int SEL_T = 256;
constexpr int SEL_MAX_NEW = 1024;
constexpr int SEL_LM_TILE = 2048;

struct SelArgs {
    const vslam_feature *feats; size_t stride_feats; const int32_t *k_each; int K;
    const float *lm; size_t stride_lm; const int32_t *l_each; int L;
    double fthr, wthr, hthr, dthr;
    int iw, ih, max_new;
    int32_t *selected; int cap; int32_t *count;
};

__host__ __device__ inline bool sel_within(float lx, float ly, float x, float y, double thr)
{
    const float dx = lx - x, dy = ly - y;
    const float xx = dx * dx, yy = dy * dy;
    const float d2 = xx + yy;
    return (double)sqrtf(d2) < thr;
}

__global__ __launch_bounds__(SEL_T) void select_kernel(SelArgs a)
{
    __shared__ float ax[SEL_MAX_NEW], ay[SEL_MAX_NEW];
    __shared__ float mx[SEL_LM_TILE], my[SEL_LM_TILE];           // landmarks of the map, a tile at a time
    __shared__ int n_acc, first;
    const int b = blockIdx.x, tid = threadIdx.x;
    const vslam_feature *feats = a.feats + (size_t)b * a.stride_feats;
    const float *lm = a.lm + (size_t)b * a.stride_lm;
    int K = a.K, L = a.L;
    if (a.k_each) K = max(0, min(K, a.k_each[b]));
    if (a.l_each) L = max(0, min(L, a.l_each[b]));
    int32_t *sel = a.selected + (size_t)b * a.cap;
    if (tid == 0) n_acc = 0;
    const bool map_in_lds = L <= SEL_LM_TILE;                     // the usual case: staged once for all chunks
    if (map_in_lds)
        for (int l = tid; l < L; l += SEL_T) { mx[l] = lm[2 * l]; my[l] = lm[2 * l + 1]; }
    __syncthreads();
    for (int c0 = 0; c0 < K; c0 += SEL_T) {
        if (n_acc >= a.max_new) break;                            // uniform: n_acc only changes between barriers
        const int i = c0 + tid, seen = n_acc;
        bool cand = false;
        float fx = 0.0f, fy = 0.0f;
        if (i < K) {
            const vslam_feature f = feats[i];
            fx = (float)f.x; fy = (float)f.y;
            cand = (double)f.score > a.fthr && a.wthr <= (double)fx && (double)fx <= (double)a.iw - a.wthr &&
                   a.hthr <= (double)fy && (double)fy <= (double)a.ih - a.hthr;
        }
        // against the map: every landmark is tested (no early exit, so that the loop carries no dependency)
        for (int t0 = 0; t0 < L; t0 += SEL_LM_TILE) {
            const int tl = min(SEL_LM_TILE, L - t0);
            if (!map_in_lds) {
                __syncthreads();
                for (int l = tid; l < tl; l += SEL_T) { mx[l] = lm[2 * (t0 + l)]; my[l] = lm[2 * (t0 + l) + 1]; }
                __syncthreads();
            }
            bool within = false;
            int l = 0;
            for (; l + 4 <= tl; l += 4) {
                const bool w0 = sel_within(mx[l], my[l], fx, fy, a.dthr), w1 = sel_within(mx[l + 1], my[l + 1], fx, fy, a.dthr);
                const bool w2 = sel_within(mx[l + 2], my[l + 2], fx, fy, a.dthr), w3 = sel_within(mx[l + 3], my[l + 3], fx, fy, a.dthr);
                within |= (w0 | w1) | (w2 | w3);
            }
            for (; l < tl; ++l) within |= sel_within(mx[l], my[l], fx, fy, a.dthr);
            cand = cand && !within;
        }
        {
            bool within = false;
            for (int l = 0; l < seen; ++l) within |= sel_within(ax[l], ay[l], fx, fy, a.dthr);
            cand = cand && !within;
        }
        for (;;) {                                                // resolve the chunk in feature order
            __syncthreads();
            const int cnt = n_acc;                                // read where nobody writes it: the winner's store comes two barriers later
            if (tid == 0) first = INT_MAX;
            __syncthreads();
            if (cand) atomicMin(&first, tid);
            __syncthreads();
            const int win = first;
            if (win == INT_MAX || cnt >= a.max_new) break;        // uniform: both values were read between barriers
            if (tid == win) {
                ax[cnt] = fx; ay[cnt] = fy;
                if (cnt < a.cap) sel[cnt] = i;
                n_acc = cnt + 1;
                cand = false;
            }
            __syncthreads();
            if (cand && tid > win) cand = !sel_within(ax[cnt], ay[cnt], fx, fy, a.dthr);
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) a.count[b] = n_acc;
}

static vslam_status check_select(vslam_ctx *ctx, const vslam_select_params *p, int K, int L)
{
    if (!p) return fail(ctx, VSLAM_ERR_ARG, "select_new_landmarks: null params");
    if (K < 0 || L < 0) return fail(ctx, VSLAM_ERR_SHAPE, "select_new_landmarks: K=%d L=%d", K, L);
    if (p->max_new > SEL_MAX_NEW) return fail(ctx, VSLAM_ERR_CAPACITY, "select_new_landmarks: max_new %d exceeds %d", p->max_new, SEL_MAX_NEW);
    if (!(p->pixel_distance_thresh >= 0.0) || p->image_width < 1 || p->image_height < 1)
        return fail(ctx, VSLAM_ERR_ARG, "select_new_landmarks: radius %g, image %d x %d", p->pixel_distance_thresh, p->image_width, p->image_height);
    return VSLAM_OK;
}

static SelArgs make_sel(const vslam_select_params *p)
{
    SelArgs a{};
    a.fthr = p->feature_thresh; a.wthr = p->width_thresh; a.hthr = p->height_thresh; a.dthr = p->pixel_distance_thresh;
    a.iw = p->image_width; a.ih = p->image_height; a.max_new = p->max_new;
    return a;
}

}  // namespace vslam

using namespace vslam;

extern "C" int vslam_pixel_distance(const float *landmark_xy, int L, float x, float y, double thresh)
{
    int within = 0;
    for (int i = 0; landmark_xy && i < L; ++i)
        if (sel_within(landmark_xy[2 * i], landmark_xy[2 * i + 1], x, y, thresh)) within = 1;
    return within;
}

extern "C" vslam_status vslam_select_new_landmarks_batch_dev(vslam_ctx *ctx, const vslam_feature *feats, size_t stride_feats,
                                                             const int32_t *k_each, int K, const float *landmark_xy, size_t stride_lm,
                                                             const int32_t *l_each, int L, const vslam_select_params *params, int batch,
                                                             int32_t *selected, int cap, int32_t *count)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!selected || !count || cap < 1 || (K > 0 && !feats) || (L > 0 && !landmark_xy)) return fail(ctx, VSLAM_ERR_ARG, "select_new_landmarks: null pointer or cap < 1");
    if (batch < 1) return fail(ctx, VSLAM_ERR_SHAPE, "select_new_landmarks: batch=%d", batch);
    VSLAM_TRY(check_select(ctx, params, K, L));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    SelArgs a = make_sel(params);
    a.feats = feats; a.stride_feats = stride_feats; a.k_each = k_each; a.K = K;
    a.lm = landmark_xy; a.stride_lm = stride_lm; a.l_each = l_each; a.L = L;
    a.selected = selected; a.cap = cap; a.count = count;
    hipLaunchKernelGGL(select_kernel, dim3(batch), dim3(SEL_T), 0, ctx->stream, a);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

extern "C" vslam_status vslam_select_new_landmarks(vslam_ctx *ctx, const vslam_feature *feats, int K, const float *landmark_xy, int L,
                                                   const vslam_select_params *params, int32_t *selected, int cap, int32_t *count)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (!selected || !count || cap < 1 || (K > 0 && !feats) || (L > 0 && !landmark_xy)) return fail(ctx, VSLAM_ERR_ARG, "select_new_landmarks: null pointer or cap < 1");
    VSLAM_TRY(check_select(ctx, params, K, L));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t fb = (size_t)(K > 0 ? K : 1) * sizeof(vslam_feature), lb = (size_t)(L > 0 ? L : 1) * 8, sb = (size_t)cap * 4;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(fb) + Arena::pad(lb) + Arena::pad(sb) + Arena::pad(8)));
    Arena ar(ctx->ws);
    vslam_feature *df = ar.take<vslam_feature>(K > 0 ? K : 1);
    float *dl = ar.take<float>(2 * (size_t)(L > 0 ? L : 1));
    int32_t *ds = ar.take<int32_t>(cap), *dc = ar.take<int32_t>(2);
    if (K > 0) VSLAM_TRY(h2d(ctx, df, feats, (size_t)K * sizeof(vslam_feature)));
    if (L > 0) VSLAM_TRY(h2d(ctx, dl, landmark_xy, (size_t)L * 8));
    SelArgs a = make_sel(params);
    a.feats = df; a.stride_feats = 0; a.k_each = nullptr; a.K = K;
    a.lm = dl; a.stride_lm = 0; a.l_each = nullptr; a.L = L;
    a.selected = ds; a.cap = cap; a.count = dc;
    hipLaunchKernelGGL(select_kernel, dim3(1), dim3(SEL_T), 0, ctx->stream, a);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    int32_t n = 0;
    VSLAM_TRY(d2h(ctx, &n, dc, 4));
    VSLAM_TRY(stream_sync(ctx));
    *count = n;
    const int nout = n < cap ? n : cap;
    if (nout > 0) {
        VSLAM_TRY(d2h(ctx, selected, ds, (size_t)nout * 4));
        VSLAM_TRY(stream_sync(ctx));
    }
    return VSLAM_OK;
}
