// Shared host-side plumbing of libvslam_hip: context, error reporting,
// device workspace.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/vslam_hip.h"

// A device-to-host copy staged in pinned memory, finished (pinned -> caller's buffer) after the stream drains.
struct vslam_pending_copy { void *dst; const void *src; size_t width, height, dpitch; };

constexpr size_t VSLAM_CHAIN_SOL_DOUBLES = (size_t)1 << 20;   // ring of solution blocks and helper sums of the chained triangular solves (one call: at most 2 * 240 * 64)

struct vslam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool counted = false;        // entered in the per-device count of live contexts
    void *ws = nullptr;          // device workspace of the entry points, grown on demand
    size_t ws_bytes = 0;
    void *scratch = nullptr;     // kernel-internal scratch (partial results), separate so that an
    size_t scratch_bytes = 0;    // entry point's carve-up of `ws` stays valid across launches
    void *pinned = nullptr;      // pinned host staging arena of the host-pointer entry points
    size_t pinned_bytes = 0, pinned_off = 0;
    std::vector<vslam_pending_copy> pending;
    bool corner_lds_attr_set = false;   // corner.hip: large dynamic LDS sizes registered for this context's device
    bool failed = false;         // an entry point returned an error: its staged copies are dropped, not completed
    hipStream_t side = nullptr;  // second stream for work that overlaps the main one (created on first use)
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // streams on which independent chains of small kernels run beside one another (the pyramid levels of the ORB front end),
    // with a pair of events per chain: its input is ready (main -> chain), it is done (chain -> main); created on first use
    hipStream_t lvl_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t lvl_ev[32] = {nullptr};        // two per pyramid level (at most 16 levels)
    int *flags = nullptr;        // device: [0] set by a multi-workgroup kernel whose wait gave up
    double *chain_sol = nullptr; // device: solution blocks of the chained triangular solve as its workgroups publish them.  A ring: every call takes
                                 // a fresh stretch (filled with the not-ready pattern), the whole ring is refilled when it runs out
    size_t chain_used = VSLAM_CHAIN_SOL_DOUBLES;   // "used up": the first call fills the ring, on the context's stream
    double *potf_dinv = nullptr; // device: per matrix of a blocked Cholesky, the inverses of the current diagonal block's 16 x 16 sub-blocks
    int potf_dinv_batch = 0;
    int *flags_host = nullptr;   // pinned mirror read by stream_finish
    int trs_solver = 1;          // VSLAM_TRS_CHOLESKY (default: device route, eigenvalue route when the Hessian is indefinite) / VSLAM_TRS_EIG
    double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // last measurement update on the device route: ms in cost evaluations, trust-region
                                 // factorisations + solves, posterior factor, whole call; counts of evaluations and factorisations
    int trs_last = -1;           // route the last measurement update took (-1: none yet)
    void *ring = nullptr;        // pinned ring of the pipelined triangular transfers
    size_t ring_bytes = 0;
    hipEvent_t ring_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool time_kernels = false;   // measurement aid: events around the two kernels of the batched log-likelihood
    hipEvent_t tev[3] = {nullptr, nullptr, nullptr};
    bool tev_recorded = false;
    char err[512] = {0};
};

namespace vslam {

inline vslam_status fail(vslam_ctx *ctx, vslam_status s, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
        // device-to-host copies the failed call staged must not reach its host pointers later; a call that staged none
        // (an argument error, a diagnostic query) leaves the context as it was
        if (!ctx->pending.empty()) ctx->failed = true;
    }
    return s;
}

#define VSLAM_HIP_TRY(ctx, expr)                                                             \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return vslam::fail((ctx), VSLAM_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,      \
                               hipGetErrorString(e_), __FILE__, __LINE__);                   \
    } while (0)

#define VSLAM_TRY(expr)                         \
    do {                                        \
        vslam_status s_ = (expr);               \
        if (s_ != VSLAM_OK) return s_;          \
    } while (0)

// Ensures the context workspace holds at least `bytes`; contents are not kept
// across a growth.  Synchronises the stream before freeing the old block.
vslam_status ws_reserve(vslam_ctx *ctx, size_t bytes);
vslam_status scratch_reserve(vslam_ctx *ctx, size_t bytes);
vslam_status potf_dinv_reserve(vslam_ctx *ctx, int batch);   // ctx->potf_dinv holds 1024 doubles per matrix
// Synchronises the context's stream and turns a raised device flag into an error.
vslam_status stream_finish(vslam_ctx *ctx);
// Synchronises the stream and completes the staged device-to-host copies.
vslam_status stream_sync(vslam_ctx *ctx);
// Host <-> device copies of the host-pointer entry points, staged through pinned memory (pageable
// hipMemcpyAsync runs at a fraction of the PCIe rate).  Pitches and widths in bytes; the host side
// may be strided.  d2h results are valid after stream_sync / stream_finish.
vslam_status h2d(vslam_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
vslam_status d2h(vslam_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
vslam_status h2d_2d(vslam_ctx *ctx, void *dst_dev, size_t dpitch, const void *src_host, size_t spitch, size_t width, size_t height);
vslam_status d2h_2d(vslam_ctx *ctx, void *dst_host, size_t dpitch, const void *src_dev, size_t spitch, size_t width, size_t height);
// Large upper-triangular factors between a caller's pageable matrix and device memory: only the columns' upper parts
// cross PCIe, in column blocks through a ring of pinned buffers, while a few host threads move the blocks between the
// ring and the caller's matrix (one contiguous pageable copy runs at about 11 GB/s, most of it the single-threaded
// pass over the caller's pages).  d2h: the strict lower triangle of dst is set to zero.  h2d: the strict lower
// triangle of the device matrix is left as it is (no kernel of the library reads it).  Both return when complete.
constexpr int PIPELINED_COPY_MIN_N = 2048;      // below this one staged copy is as fast
vslam_status d2h_upper_pipelined(vslam_ctx *ctx, double *dst_host, size_t ldd, const double *src_dev, size_t lds, int n);
vslam_status h2d_upper_pipelined(vslam_ctx *ctx, double *dst_dev, size_t ldd, const double *src_host, size_t lds, int n);
// Looks up, once per GPU, the device address of the static scratch array of the entry point that
// allocates nothing (point_loglik.hip); called at context creation with the device current.
vslam_status loglik_static_scratch_init(vslam_ctx *ctx);
// Creates ctx->lvl_stream[] and ctx->lvl_ev[] on first use.
vslam_status level_streams_init(vslam_ctx *ctx);
// Creates ctx->side and ctx->ev[] on first use.
vslam_status side_stream_init(vslam_ctx *ctx);

// Carves 256-byte aligned sub-buffers out of the workspace.
struct Arena {
    char *base;
    size_t off = 0;
    explicit Arena(void *p) : base(static_cast<char *>(p)) {}
    static size_t pad(size_t b) { return (b + 255) & ~size_t(255); }
    template <class T> T *take(size_t count)
    {
        T *p = reinterpret_cast<T *>(base + off);
        off += pad(count * sizeof(T));
        return p;
    }
};

// Camera constants resolved on the host (coefficient gating of the reference's
// src/cameraModel.hpp:114-148) and passed to kernels by value.
struct CamK {
    double fx, fy, cx, cy;
    double k1, k2, k3, k4, k5, k6, p1, p2, s1, s2, s3, s4;
    double cos_half_fov, wmax, hmax;
};
vslam_status resolve_camera(vslam_ctx *ctx, const vslam_camera *cam, CamK *out);

// Most workgroups of ONE launch whose members wait for one another (blocked QR, chained triangular solve): such a launch
// needs all its members resident at once, one per compute unit (each takes most of a unit's LDS).  Derived from the
// compute-unit count the device reports when a context is created on it (16 units are left for launches that do not take
// the GPU's turn below): 240 on a whole MI355X, less on a partitioned or masked one; the smallest over all devices this
// process has contexts on.  Callers that would exceed it take their multi-launch paths instead.
int resident_workgroup_cap();
// Number of live contexts on the device: with more than one, launches of different contexts can overlap on the GPU.
int contexts_on_device(int device);

// Launches whose workgroups wait for one another (the blocked QR's exchange, the chained triangular solve with helper
// workgroups) are queued one after the other on each GPU, whatever context or stream they come from: while the guard
// lives, the stream has been made to wait for the event the previous such call recorded at its end, and the guard's end
// records this call's.  Workgroups of two of them can then never hold compute units while waiting for members that found
// none free.  Nothing blocks on the host beyond the mutex around the enqueue.
class DeviceTurnGuard {
public:
    explicit DeviceTurnGuard(vslam_ctx *ctx);
    ~DeviceTurnGuard();
    vslam_status status() const { return st_; }
    DeviceTurnGuard(const DeviceTurnGuard &) = delete;
    DeviceTurnGuard &operator=(const DeviceTurnGuard &) = delete;
private:
    vslam_ctx *ctx_;
    vslam_status st_;
    bool held_;
};

// A grid of (items, rows) workgroups over a bank, rows being its matrices: the hardware deals workgroup ids round-robin
// to the eight XCDs, which would spread the items of one matrix over all eight L2 caches.  This renumbering gives
// every XCD whole matrices instead (all items of row r on XCD r mod 8), so that what the items of a matrix share -- the
// row panel of a factorisation step -- is fetched into one L2 once.  The last rows mod 8 keep the plain numbering.
#if defined(__HIPCC__)
__device__ __forceinline__ void bank_grid_xcd(int *item, int *row)
{
    const int T = (int)gridDim.x, R = (int)gridDim.y, full = R & ~7;
    const int L = (int)blockIdx.y * T + (int)blockIdx.x;
    if (L < full * T) {
        const int xcd = L & 7, idx = L >> 3;
        *row = xcd + 8 * (idx / T);
        *item = idx % T;
    } else {
        *row = (int)blockIdx.y;
        *item = (int)blockIdx.x;
    }
}
#endif

// Which instance of a bank row y of a grid works on: every row its own (sel null), the rows whose mask word is
// non-zero (listed == 0: the grid covers all instances, the others leave at once) or the y-th entry of a list of
// instances (listed != 0: the grid covers the list only -- a launch over a thousand instances of which a handful
// take part otherwise spends its time dispatching workgroups that leave).  -1: nothing to do.
#if defined(__HIPCC__)
__device__ __forceinline__ int instance_of(const int *sel, int listed, int y)
{
    if (!sel) return y;
    if (listed) return sel[y];
    return sel[y] ? y : -1;
}
#endif

}  // namespace vslam
