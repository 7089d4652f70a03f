// Context lifecycle, status strings, camera.xml reader.
#include "common.hpp"
#include <atomic>
#include <mutex>
#include <thread>
#include <sched.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace vslam {

vslam_status ws_reserve(vslam_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->ws_bytes) return VSLAM_OK;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->ws) {
        VSLAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        VSLAM_HIP_TRY(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    size_t want = bytes + (bytes >> 2) + (1u << 20);
    hipError_t e = hipMalloc(&ctx->ws, want);
    if (e != hipSuccess) return fail(ctx, VSLAM_ERR_ALLOC, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    ctx->ws_bytes = want;
    return VSLAM_OK;
}

vslam_status scratch_reserve(vslam_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return VSLAM_OK;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->scratch) {
        VSLAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        VSLAM_HIP_TRY(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    size_t want = bytes + (bytes >> 2) + (1u << 16);
    hipError_t e = hipMalloc(&ctx->scratch, want);
    if (e != hipSuccess) return fail(ctx, VSLAM_ERR_ALLOC, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    ctx->scratch_bytes = want;
    return VSLAM_OK;
}

vslam_status potf_dinv_reserve(vslam_ctx *ctx, int batch)
{
    if (batch <= ctx->potf_dinv_batch) return VSLAM_OK;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->potf_dinv) {
        VSLAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        VSLAM_HIP_TRY(ctx, hipFree(ctx->potf_dinv));
        ctx->potf_dinv = nullptr;
        ctx->potf_dinv_batch = 0;
    }
    const int want = batch < 8 ? 8 : batch;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->potf_dinv), (size_t)want * 1024 * 8);
    if (e != hipSuccess) return fail(ctx, VSLAM_ERR_ALLOC, "hipMalloc(%zu) failed: %s", (size_t)want * 1024 * 8, hipGetErrorString(e));
    ctx->potf_dinv_batch = want;
    return VSLAM_OK;
}

static vslam_status pinned_take(vslam_ctx *ctx, size_t bytes, char **out)
{
    const size_t need = (bytes + 255) & ~size_t(255);
    if (ctx->failed) VSLAM_TRY(stream_sync(ctx));        // leftovers of a call that returned an error: drained and dropped
    if (ctx->pinned_off + need > ctx->pinned_bytes) {
        VSLAM_TRY(stream_sync(ctx));                     // everything staged so far is consumed; arena is empty again
        if (need > ctx->pinned_bytes) {
            if (ctx->pinned) { VSLAM_HIP_TRY(ctx, hipHostFree(ctx->pinned)); ctx->pinned = nullptr; ctx->pinned_bytes = 0; }
            size_t want = need + (need >> 2);
            if (want < (size_t(4) << 20)) want = size_t(4) << 20;
            hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
            if (e != hipSuccess) return fail(ctx, VSLAM_ERR_ALLOC, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            ctx->pinned_bytes = want;
        }
    }
    *out = static_cast<char *>(ctx->pinned) + ctx->pinned_off;
    ctx->pinned_off += need;
    return VSLAM_OK;
}

vslam_status stream_sync(vslam_ctx *ctx)
{
    VSLAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->failed) {
        // Copies staged by an entry point that then failed: their host destinations may be gone
        // (temporaries of the caller, stack variables of the entry point).
        ctx->failed = false;
        ctx->pending.clear();
        ctx->pinned_off = 0;
        return VSLAM_OK;
    }
    for (const vslam_pending_copy &c : ctx->pending) {
        const char *src = static_cast<const char *>(c.src);
        char *dst = static_cast<char *>(c.dst);
        if (c.dpitch == c.width || c.height == 1) std::memcpy(dst, src, c.width * c.height);
        else for (size_t r = 0; r < c.height; ++r) std::memcpy(dst + r * c.dpitch, src + r * c.width, c.width);
    }
    ctx->pending.clear();
    ctx->pinned_off = 0;
    return VSLAM_OK;
}

// Transfers of this size and more go between device memory and the caller's buffer directly: the runtime moves
// pageable memory in pipelined pieces, which beats one pass through the staging arena plus one large host copy
// (and keeps the arena small).  The call returns when the transfer is complete.
constexpr size_t DIRECT_COPY_BYTES = (size_t)8 << 20;

vslam_status h2d_2d(vslam_ctx *ctx, void *dst_dev, size_t dpitch, const void *src_host, size_t spitch, size_t width, size_t height)
{
    if (width == 0 || height == 0) return VSLAM_OK;
    if (width * height >= DIRECT_COPY_BYTES) {
        if ((spitch == width && dpitch == width) || height == 1) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, width * height, hipMemcpyHostToDevice, ctx->stream));
        else VSLAM_HIP_TRY(ctx, hipMemcpy2DAsync(dst_dev, dpitch, src_host, spitch, width, height, hipMemcpyHostToDevice, ctx->stream));
        VSLAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // the caller's buffer may change as soon as this returns
        return VSLAM_OK;
    }
    char *stage;
    VSLAM_TRY(pinned_take(ctx, width * height, &stage));
    const char *src = static_cast<const char *>(src_host);
    if (spitch == width || height == 1) std::memcpy(stage, src, width * height);
    else for (size_t r = 0; r < height; ++r) std::memcpy(stage + r * width, src + r * spitch, width);
    if (dpitch == width || height == 1) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dst_dev, stage, width * height, hipMemcpyHostToDevice, ctx->stream));
    else VSLAM_HIP_TRY(ctx, hipMemcpy2DAsync(dst_dev, dpitch, stage, width, width, height, hipMemcpyHostToDevice, ctx->stream));
    return VSLAM_OK;
}

vslam_status d2h_2d(vslam_ctx *ctx, void *dst_host, size_t dpitch, const void *src_dev, size_t spitch, size_t width, size_t height)
{
    if (width == 0 || height == 0) return VSLAM_OK;
    if (width * height >= DIRECT_COPY_BYTES) {
        // ordered behind everything queued so far; complete on return (a failed call later on cannot leave it pending)
        if ((spitch == width && dpitch == width) || height == 1) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, width * height, hipMemcpyDeviceToHost, ctx->stream));
        else VSLAM_HIP_TRY(ctx, hipMemcpy2DAsync(dst_host, dpitch, src_dev, spitch, width, height, hipMemcpyDeviceToHost, ctx->stream));
        VSLAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return VSLAM_OK;
    }
    char *stage;
    VSLAM_TRY(pinned_take(ctx, width * height, &stage));
    if (spitch == width || height == 1) VSLAM_HIP_TRY(ctx, hipMemcpyAsync(stage, src_dev, width * height, hipMemcpyDeviceToHost, ctx->stream));
    else VSLAM_HIP_TRY(ctx, hipMemcpy2DAsync(stage, width, src_dev, spitch, width, height, hipMemcpyDeviceToHost, ctx->stream));
    ctx->pending.push_back({dst_host, stage, width, height, dpitch});
    return VSLAM_OK;
}

vslam_status h2d(vslam_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) { return h2d_2d(ctx, dst_dev, bytes, src_host, bytes, bytes, 1); }
vslam_status d2h(vslam_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) { return d2h_2d(ctx, dst_host, bytes, src_dev, bytes, bytes, 1); }

namespace {
constexpr int RING_SLOTS = 8, RING_COLS = 64;

vslam_status ring_reserve(vslam_ctx *ctx, int n)
{
    const size_t slot = (size_t)RING_COLS * n * 8, want = slot * RING_SLOTS;
    if (ctx->ring_bytes < want) {
        if (ctx->ring) { VSLAM_HIP_TRY(ctx, hipHostFree(ctx->ring)); ctx->ring = nullptr; ctx->ring_bytes = 0; }
        hipError_t e = hipHostMalloc(&ctx->ring, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(ctx, VSLAM_ERR_ALLOC, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        ctx->ring_bytes = want;
    }
    for (int i = 0; i < RING_SLOTS; ++i)
        if (!ctx->ring_ev[i]) VSLAM_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ring_ev[i], hipEventDisableTiming));
    return VSLAM_OK;
}

int transfer_threads()
{
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) hw = (unsigned)c; }
    int t = (int)(hw > 1 ? hw - 1 : 1);
    return t > RING_SLOTS ? RING_SLOTS : t;
}
}  // namespace

// Column block c holds columns [c * RING_COLS, ...) and, of each, rows 0 .. (end of the block): a rectangle in the ring.
vslam_status d2h_upper_pipelined(vslam_ctx *ctx, double *dst, size_t ldd, const double *src, size_t lds, int n)
{
    VSLAM_TRY(ring_reserve(ctx, n));
    const int nchunk = (n + RING_COLS - 1) / RING_COLS, W = transfer_threads();
    const size_t slot_doubles = (size_t)RING_COLS * n;
    std::vector<std::atomic<int>> recorded(nchunk), consumed(nchunk);
    for (int c = 0; c < nchunk; ++c) { recorded[c].store(0); consumed[c].store(0); }
    std::atomic<int> failed{0};
    const int device = ctx->device;
    auto worker = [&](int t) {
        (void)hipSetDevice(device);
        for (int c = t; c < nchunk; c += W) {
            while (!recorded[c].load(std::memory_order_acquire)) { if (failed.load()) return; std::this_thread::yield(); }
            if (hipEventSynchronize(ctx->ring_ev[c % RING_SLOTS]) != hipSuccess) { failed.store(1); consumed[c].store(1, std::memory_order_release); return; }
            const int j0 = c * RING_COLS, j1 = std::min(n, j0 + RING_COLS), rows = j1;
            const double *buf = static_cast<const double *>(ctx->ring) + (size_t)(c % RING_SLOTS) * slot_doubles;
            for (int j = j0; j < j1; ++j) {
                double *col = dst + (size_t)j * ldd;
                std::memcpy(col, buf + (size_t)(j - j0) * rows, (size_t)(j + 1) * 8);
                std::memset(col + j + 1, 0, (size_t)(n - j - 1) * 8);
            }
            consumed[c].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < W; ++t) pool.emplace_back(worker, t);
    vslam_status st = VSLAM_OK;
    for (int c = 0; c < nchunk && st == VSLAM_OK; ++c) {
        if (c >= RING_SLOTS) while (!consumed[c - RING_SLOTS].load(std::memory_order_acquire)) std::this_thread::yield();
        const int j0 = c * RING_COLS, j1 = std::min(n, j0 + RING_COLS), rows = j1;
        double *buf = static_cast<double *>(ctx->ring) + (size_t)(c % RING_SLOTS) * slot_doubles;
        hipError_t e = hipMemcpy2DAsync(buf, (size_t)rows * 8, src + (size_t)j0 * lds, lds * 8, (size_t)rows * 8, (size_t)(j1 - j0), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ring_ev[c % RING_SLOTS], ctx->stream);
        if (e != hipSuccess) { failed.store(1); st = fail(ctx, VSLAM_ERR_DEVICE, "pipelined copy to the host: %s", hipGetErrorString(e)); break; }
        recorded[c].store(1, std::memory_order_release);
    }
    for (std::thread &th : pool) th.join();
    if (st == VSLAM_OK && failed.load()) st = fail(ctx, VSLAM_ERR_DEVICE, "pipelined copy to the host: waiting for a block failed");
    return st;
}

vslam_status h2d_upper_pipelined(vslam_ctx *ctx, double *dst, size_t ldd, const double *src, size_t lds, int n)
{
    VSLAM_TRY(ring_reserve(ctx, n));
    const int nchunk = (n + RING_COLS - 1) / RING_COLS, W = transfer_threads();
    const size_t slot_doubles = (size_t)RING_COLS * n;
    std::vector<std::atomic<int>> packed(nchunk), sent(nchunk);
    for (int c = 0; c < nchunk; ++c) { packed[c].store(0); sent[c].store(0); }
    std::atomic<int> failed{0};
    auto worker = [&](int t) {
        for (int c = t; c < nchunk; c += W) {
            // slot c % RING_SLOTS is free once the copy of chunk c - RING_SLOTS has left it
            if (c >= RING_SLOTS) while (!sent[c - RING_SLOTS].load(std::memory_order_acquire)) { if (failed.load()) return; std::this_thread::yield(); }
            const int j0 = c * RING_COLS, j1 = std::min(n, j0 + RING_COLS), rows = j1;
            double *buf = static_cast<double *>(ctx->ring) + (size_t)(c % RING_SLOTS) * slot_doubles;
            for (int j = j0; j < j1; ++j) {
                std::memcpy(buf + (size_t)(j - j0) * rows, src + (size_t)j * lds, (size_t)(j + 1) * 8);
                if (j + 1 < rows) std::memset(buf + (size_t)(j - j0) * rows + j + 1, 0, (size_t)(rows - j - 1) * 8);    // the block's own diagonal tile gets clean zeros
            }
            packed[c].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < W; ++t) pool.emplace_back(worker, t);
    vslam_status st = VSLAM_OK;
    for (int c = 0; c < nchunk && st == VSLAM_OK; ++c) {
        while (!packed[c].load(std::memory_order_acquire)) std::this_thread::yield();
        const int j0 = c * RING_COLS, j1 = std::min(n, j0 + RING_COLS), rows = j1;
        const double *buf = static_cast<const double *>(ctx->ring) + (size_t)(c % RING_SLOTS) * slot_doubles;
        hipError_t e = hipMemcpy2DAsync(dst + (size_t)j0 * ldd, ldd * 8, buf, (size_t)rows * 8, (size_t)rows * 8, (size_t)(j1 - j0), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ring_ev[c % RING_SLOTS], ctx->stream);
        // the slot may be refilled once this copy has read it: the sender waits for the copy RING_SLOTS - 2 chunks back
        if (e == hipSuccess && c >= RING_SLOTS - 2) {
            const int d = c - (RING_SLOTS - 2);
            // chunk d's event slot is still its own: chunk d + RING_SLOTS has not been enqueued yet
            e = hipEventSynchronize(ctx->ring_ev[d % RING_SLOTS]);
            if (e == hipSuccess) sent[d].store(1, std::memory_order_release);
        }
        if (e != hipSuccess) { failed.store(1); st = fail(ctx, VSLAM_ERR_DEVICE, "pipelined copy to the device: %s", hipGetErrorString(e)); break; }
    }
    if (st == VSLAM_OK) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = fail(ctx, VSLAM_ERR_DEVICE, "pipelined copy to the device: %s", hipGetErrorString(e));
    }
    for (int c = 0; c < nchunk; ++c) sent[c].store(1, std::memory_order_release);
    failed.store(st != VSLAM_OK);
    for (std::thread &th : pool) th.join();
    return st;
}

vslam_status side_stream_init(vslam_ctx *ctx)
{
    if (ctx->side) return VSLAM_OK;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    VSLAM_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    for (int i = 0; i < 6; ++i) VSLAM_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev[i], hipEventDisableTiming));
    return VSLAM_OK;
}

vslam_status level_streams_init(vslam_ctx *ctx)
{
    if (ctx->lvl_stream[0]) return VSLAM_OK;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < 32; ++i) VSLAM_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->lvl_ev[i], hipEventDisableTiming));
    for (int i = 2; i >= 0; --i) VSLAM_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->lvl_stream[i], hipStreamNonBlocking));
    return VSLAM_OK;
}

namespace {
struct DeviceTurn { std::mutex m; hipEvent_t last = nullptr; int contexts = 0; };
DeviceTurn g_turn[64];
std::mutex g_cap_mutex;
int g_resident_cap = 240;
}

int resident_workgroup_cap()
{
    std::lock_guard<std::mutex> lock(g_cap_mutex);
    return g_resident_cap;
}

int contexts_on_device(int device)
{
    if (device < 0 || device >= 64) return 2;                   // no bookkeeping for such a device: assume company
    std::lock_guard<std::mutex> lock(g_turn[device].m);
    return g_turn[device].contexts;
}

static void note_context(int device, int compute_units, int delta)
{
    if (compute_units > 0) {
        std::lock_guard<std::mutex> lock(g_cap_mutex);
        const int cap = compute_units > 32 ? compute_units - 16 : (compute_units > 1 ? compute_units / 2 : 1);
        if (cap < g_resident_cap) g_resident_cap = cap;
    }
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lock(g_turn[device].m);
        g_turn[device].contexts += delta;
    }
}

DeviceTurnGuard::DeviceTurnGuard(vslam_ctx *ctx) : ctx_(ctx), st_(VSLAM_OK), held_(false)
{
    if (ctx->device < 0 || ctx->device >= 64) return;           // no queue for such a device: the call runs unguarded
    DeviceTurn &turn = g_turn[ctx->device];
    turn.m.lock();
    held_ = true;
    hipError_t e = hipSuccess;
    if (turn.last) e = hipStreamWaitEvent(ctx->stream, turn.last, 0);
    else e = hipEventCreateWithFlags(&turn.last, hipEventDisableTiming);
    if (e != hipSuccess) st_ = fail(ctx, VSLAM_ERR_DEVICE, "device turn: %s", hipGetErrorString(e));
}

DeviceTurnGuard::~DeviceTurnGuard()
{
    if (!held_) return;
    DeviceTurn &turn = g_turn[ctx_->device];
    if (turn.last) (void)hipEventRecord(turn.last, ctx_->stream);    // also after a failed enqueue: what was queued still runs
    turn.m.unlock();
}

vslam_status stream_finish(vslam_ctx *ctx)
{
    VSLAM_HIP_TRY(ctx, hipMemcpyAsync(ctx->flags_host, ctx->flags, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VSLAM_TRY(stream_sync(ctx));
    if (ctx->flags_host[0] != 0) {
        ctx->flags_host[0] = 0;
        VSLAM_HIP_TRY(ctx, hipMemsetAsync(ctx->flags, 0, sizeof(int), ctx->stream));
        return fail(ctx, VSLAM_ERR_DEVICE, "a multi-workgroup kernel gave up waiting at its barrier; results are invalid");
    }
    return VSLAM_OK;
}

vslam_status resolve_camera(vslam_ctx *ctx, const vslam_camera *cam, CamK *k)
{
    const int RATIONAL = 0x4000, THIN = 0x8000;
    if (!cam) return fail(ctx, VSLAM_ERR_ARG, "camera is null");
    if (!(cam->fx > 0.0 && cam->fy > 0.0 && std::isfinite(cam->fx) && std::isfinite(cam->fy) && std::isfinite(cam->cx) && std::isfinite(cam->cy)))
        return fail(ctx, VSLAM_ERR_ARG, "camera: focal lengths must be positive and finite (fx=%g fy=%g cx=%g cy=%g)", cam->fx, cam->fy, cam->cx, cam->cy);
    if (cam->width < 1 || cam->height < 1 || !(cam->fov_deg > 0.0 && cam->fov_deg < 360.0))
        return fail(ctx, VSLAM_ERR_ARG, "camera: image size %d x %d, field of view %g deg", cam->width, cam->height, cam->fov_deg);
    if (cam->flag & ~(RATIONAL | THIN))
        return fail(ctx, VSLAM_ERR_UNSUPPORTED, "calibration flag %d: only RATIONAL|THIN_PRISM are supported", cam->flag);
    bool rational = (cam->flag & RATIONAL) == RATIONAL, thin = (cam->flag & THIN) == THIN;
    k->fx = cam->fx; k->fy = cam->fy; k->cx = cam->cx; k->cy = cam->cy;
    k->k1 = cam->d[0]; k->k2 = cam->d[1]; k->p1 = cam->d[2]; k->p2 = cam->d[3]; k->k3 = cam->d[4];
    if (rational) {
        const double *s = thin ? cam->d + 8 : cam->d + 5;
        k->s1 = s[0]; k->s2 = s[1]; k->s3 = s[2]; k->s4 = s[3];
    } else {
        k->s1 = k->s2 = k->s3 = k->s4 = 0.0;
    }
    if (thin) { k->k4 = cam->d[5]; k->k5 = cam->d[6]; k->k6 = cam->d[7]; }
    else      { k->k4 = k->k5 = k->k6 = 0.0; }
    const double pi = 3.14159265358979323846264338327950288;
    k->cos_half_fov = std::cos(cam->fov_deg * ((pi / 180.0) / 2.));
    k->wmax = double(cam->width - 1);
    k->hmax = double(cam->height - 1);
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" {

int vslam_has_device_code(void) { return 1; }

const char *vslam_status_string(vslam_status s)
{
    switch (s) {
    case VSLAM_OK: return "ok";
    case VSLAM_ERR_ARG: return "invalid argument";
    case VSLAM_ERR_SHAPE: return "shape mismatch";
    case VSLAM_ERR_DEVICE: return "HIP runtime error";
    case VSLAM_ERR_ALLOC: return "allocation failed";
    case VSLAM_ERR_IO: return "I/O error";
    case VSLAM_ERR_CAPACITY: return "capacity exceeded";
    case VSLAM_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
    }
}

const char *vslam_last_error(const vslam_ctx *ctx) { return ctx ? ctx->err : "null context"; }

static vslam_status ctx_make(int device, void *stream, bool borrow, vslam_ctx **out)
{
    if (!out) return VSLAM_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return VSLAM_ERR_DEVICE;
    if (device < 0 || device >= count) return VSLAM_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return VSLAM_ERR_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return VSLAM_ERR_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "libvslam_hip: device %d is %s; this library carries gfx950 code only\n", device,
                     prop.gcnArchName);
        return VSLAM_ERR_UNSUPPORTED;
    }
    vslam_ctx *c = new vslam_ctx();
    c->device = device;
    c->counted = true;
    vslam::note_context(device, prop.multiProcessorCount, +1);
    if (borrow) {
        c->stream = static_cast<hipStream_t>(stream);
        c->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return VSLAM_ERR_DEVICE; }
        c->own_stream = true;
    }
    if (hipMalloc(reinterpret_cast<void **>(&c->flags), 320 * sizeof(int)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->chain_sol), (size_t)VSLAM_CHAIN_SOL_DOUBLES * 8) != hipSuccess ||
        hipMemset(c->flags, 0, 320 * sizeof(int)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->flags_host), 64 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        vslam_ctx_destroy(c);
        return VSLAM_ERR_ALLOC;
    }
    c->flags_host[0] = 0;
    if (loglik_static_scratch_init(c) != VSLAM_OK) {
        std::fprintf(stderr, "libvslam_hip: %s\n", c->err);
        vslam_ctx_destroy(c);
        return VSLAM_ERR_DEVICE;
    }
    *out = c;
    return VSLAM_OK;
}

vslam_status vslam_ctx_create(int device, vslam_ctx **out) { return ctx_make(device, nullptr, false, out); }
vslam_status vslam_ctx_create_on_stream(int device, void *hip_stream, vslam_ctx **out)
{
    return ctx_make(device, hip_stream, true, out);
}

vslam_status vslam_ctx_destroy(vslam_ctx *ctx)
{
    if (!ctx) return VSLAM_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->counted) vslam::note_context(ctx->device, 0, -1);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->potf_dinv) (void)hipFree(ctx->potf_dinv);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ring) (void)hipHostFree(ctx->ring);
    for (int i = 0; i < 8; ++i) if (ctx->ring_ev[i]) (void)hipEventDestroy(ctx->ring_ev[i]);
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    for (int i = 0; i < 3; ++i) if (ctx->lvl_stream[i]) { (void)hipStreamSynchronize(ctx->lvl_stream[i]); (void)hipStreamDestroy(ctx->lvl_stream[i]); }
    for (int i = 0; i < 32; ++i) if (ctx->lvl_ev[i]) (void)hipEventDestroy(ctx->lvl_ev[i]);
    for (int i = 0; i < 6; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 3; ++i) if (ctx->tev[i]) (void)hipEventDestroy(ctx->tev[i]);
    if (ctx->flags) (void)hipFree(ctx->flags);
    if (ctx->chain_sol) (void)hipFree(ctx->chain_sol);
    if (ctx->flags_host) (void)hipHostFree(ctx->flags_host);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VSLAM_OK;
}

vslam_status vslam_ctx_synchronize(vslam_ctx *ctx)
{
    if (!ctx) return VSLAM_ERR_ARG;
    return stream_finish(ctx);
}

// Reads the numbers of <tag> ... <data> n n n </data> ... </tag>.
static bool xml_block(const std::string &txt, const char *tag, std::string *body)
{
    std::string open = std::string("<") + tag;
    size_t a = txt.find(open);
    if (a == std::string::npos) return false;
    a = txt.find('>', a);
    if (a == std::string::npos) return false;
    size_t b = txt.find(std::string("</") + tag + ">", a);
    if (b == std::string::npos) return false;
    *body = txt.substr(a + 1, b - a - 1);
    return true;
}
static bool xml_numbers(const std::string &body, std::vector<double> *v)
{
    std::istringstream is(body);
    std::string tok;
    v->clear();
    while (is >> tok) {
        char *end = nullptr;
        double d = std::strtod(tok.c_str(), &end);
        if (end == tok.c_str() || *end != '\0') return false;
        v->push_back(d);
    }
    return true;
}

vslam_status vslam_camera_from_xml(const char *path, vslam_camera *cam)
{
    if (!path || !cam) return VSLAM_ERR_ARG;
    std::ifstream f(path);
    if (!f) return VSLAM_ERR_IO;
    std::stringstream ss;
    ss << f.rdbuf();
    std::string txt = ss.str(), K, D, body, data;
    std::vector<double> v;
    std::memset(cam, 0, sizeof *cam);
    cam->fov_deg = 120.0;
    if (!xml_block(txt, "camera_matrix", &K) || !xml_block(K, "data", &data) || !xml_numbers(data, &v) || v.size() != 9)
        return VSLAM_ERR_IO;
    cam->fx = v[0]; cam->cx = v[2]; cam->fy = v[4]; cam->cy = v[5];
    if (!xml_block(txt, "distortion_coefficients", &D) || !xml_block(D, "data", &data) || !xml_numbers(data, &v) ||
        v.empty() || v.size() > 12)
        return VSLAM_ERR_IO;
    for (size_t i = 0; i < v.size(); ++i) cam->d[i] = v[i];
    if (!xml_block(txt, "flag", &body) || !xml_numbers(body, &v) || v.size() != 1) return VSLAM_ERR_IO;
    cam->flag = int32_t(v[0]);
    if (!xml_block(txt, "imageSize", &body) || !xml_numbers(body, &v) || v.size() != 2) return VSLAM_ERR_IO;
    if (!(v[0] >= 1.0 && v[0] <= 65536.0 && v[1] >= 1.0 && v[1] <= 65536.0)) return VSLAM_ERR_IO;
    cam->width = int32_t(v[0]);
    cam->height = int32_t(v[1]);
    // a calibration that cannot be one: non-finite entries or a non-positive focal length
    if (!(std::isfinite(cam->fx) && std::isfinite(cam->fy) && std::isfinite(cam->cx) && std::isfinite(cam->cy)) || cam->fx <= 0.0 || cam->fy <= 0.0)
        return VSLAM_ERR_IO;
    for (int i = 0; i < 12; ++i)
        if (!std::isfinite(cam->d[i])) return VSLAM_ERR_IO;
    return VSLAM_OK;
}

// ---- binary PGM (P5) ----
namespace {
// Next header token of a PGM: skips white space and '#' comments (to the end of the line).  False at end of file
// or when the token is not a decimal number of at most 9 digits.
bool pgm_token(std::FILE *f, long *out)
{
    int c = std::fgetc(f);
    for (;;) {
        while (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f') c = std::fgetc(f);
        if (c != '#') break;
        while (c != '\n' && c != '\r' && c != EOF) c = std::fgetc(f);
    }
    if (c < '0' || c > '9') return false;
    long v = 0;
    int digits = 0;
    while (c >= '0' && c <= '9') {
        if (++digits > 9) return false;
        v = v * 10 + (c - '0');
        c = std::fgetc(f);
    }
    if (c != EOF) std::ungetc(c, f);
    *out = v;
    return true;
}
// Reads the header; on success the stream stands on the first sample.
vslam_status pgm_header(std::FILE *f, int32_t *w, int32_t *h, int32_t *maxval)
{
    if (std::fgetc(f) != 'P' || std::fgetc(f) != '5') return VSLAM_ERR_IO;
    long W, H, M;
    if (!pgm_token(f, &W) || !pgm_token(f, &H) || !pgm_token(f, &M)) return VSLAM_ERR_IO;
    const int c = std::fgetc(f);                           // exactly one white-space byte ends the header
    if (!(c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f')) return VSLAM_ERR_IO;
    if (W < 1 || H < 1 || W > 65536 || H > 65536 || M < 1 || M > 65535) return VSLAM_ERR_IO;
    *w = (int32_t)W; *h = (int32_t)H; *maxval = (int32_t)M;
    return VSLAM_OK;
}
}  // namespace

vslam_status vslam_pgm_info(const char *path, int32_t *w, int32_t *h, int32_t *maxval)
{
    if (!path || !w || !h) return VSLAM_ERR_ARG;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return VSLAM_ERR_IO;
    int32_t m = 0;
    const vslam_status st = pgm_header(f, w, h, &m);
    std::fclose(f);
    if (st == VSLAM_OK && maxval) *maxval = m;
    return st;
}

vslam_status vslam_pgm_read(const char *path, uint8_t *gray, size_t cap, int32_t *w, int32_t *h)
{
    if (!path || !gray || !w || !h) return VSLAM_ERR_ARG;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return VSLAM_ERR_IO;
    int32_t m = 0;
    vslam_status st = pgm_header(f, w, h, &m);
    if (st == VSLAM_OK && m > 255) st = VSLAM_ERR_UNSUPPORTED;          // two bytes per sample
    const size_t need = st == VSLAM_OK ? (size_t)*w * (size_t)*h : 0;
    if (st == VSLAM_OK && need > cap) st = VSLAM_ERR_CAPACITY;
    if (st == VSLAM_OK && std::fread(gray, 1, need, f) != need) st = VSLAM_ERR_IO;      // truncated file
    std::fclose(f);
    return st;
}

vslam_status vslam_pgm_write(const char *path, const uint8_t *gray, int w, int h, int stride)
{
    if (!path || !gray) return VSLAM_ERR_ARG;
    if (w < 1 || h < 1 || w > 65536 || h > 65536 || stride < w) return VSLAM_ERR_SHAPE;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return VSLAM_ERR_IO;
    bool ok = std::fprintf(f, "P5\n%d %d\n255\n", w, h) > 0;
    for (int r = 0; ok && r < h; ++r) ok = std::fwrite(gray + (size_t)r * stride, 1, (size_t)w, f) == (size_t)w;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? VSLAM_OK : VSLAM_ERR_IO;
}

}  // extern "C"
