// Brute-force Hamming matcher: cv::BFMatcher(cv::NORM_HAMMING).match(landmark_descriptors,
// descriptors_found, matches) at reference src/SLAM.cpp:222,355 -- for every query descriptor the
// train descriptor at the smallest Hamming distance, the lowest train index among equals.
// Byte/bit work, HBM/L2 bound: every workgroup keeps MQ query descriptors in registers and streams the
// train set once, 16 bytes per load.
#include "common.hpp"

namespace vslam {

constexpr int MQ = 8;            // queries per workgroup
constexpr int MT = 256;          // threads per workgroup

// key = distance << 32 | train index: the minimum key is the first closest descriptor.
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// words = descriptor length in 32-bit words (8 for ORB's 32 bytes); descriptors are rows of `words` words.
template <int WORDS>
__global__ __launch_bounds__(MT) void match_hamming_kernel(const uint32_t *query, size_t stride_q, int nq, const uint32_t *train,
                                                           size_t stride_t, int nt, const int32_t *nt_each, int32_t *idx,
                                                           int32_t *dist)
{
    const int inst = blockIdx.y, q0 = blockIdx.x * MQ, tid = threadIdx.x;
    const uint32_t *Q = query + (size_t)inst * stride_q;
    const uint32_t *T = train + (size_t)inst * stride_t;
    if (nt_each) { const int c = nt_each[inst]; nt = c < nt ? (c < 0 ? 0 : c) : nt; }
    uint32_t qv[MQ][WORDS];
#pragma unroll
    for (int q = 0; q < MQ; ++q)
#pragma unroll
        for (int w = 0; w < WORDS; ++w) qv[q][w] = (q0 + q < nq) ? Q[(size_t)(q0 + q) * WORDS + w] : 0u;
    unsigned long long best[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) best[q] = ~0ull;
    for (int j = tid; j < nt; j += MT) {
        uint32_t tv[WORDS];
        if constexpr (WORDS % 4 == 0) {
            const uint4 *p = reinterpret_cast<const uint4 *>(T + (size_t)j * WORDS);
#pragma unroll
            for (int w = 0; w < WORDS / 4; ++w) {
                const uint4 v = p[w];
                tv[4 * w] = v.x; tv[4 * w + 1] = v.y; tv[4 * w + 2] = v.z; tv[4 * w + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int w = 0; w < WORDS; ++w) tv[w] = T[(size_t)j * WORDS + w];
        }
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            unsigned d = 0;
#pragma unroll
            for (int w = 0; w < WORDS; ++w) d += __popc(qv[q][w] ^ tv[w]);
            const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)j;
            best[q] = key < best[q] ? key : best[q];
        }
    }
    __shared__ unsigned long long red[MT / 64][MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const unsigned long long m = wave_min_u64(best[q]);
        if ((tid & 63) == 0) red[tid >> 6][q] = m;
    }
    __syncthreads();
    if (tid < MQ && q0 + tid < nq) {
        unsigned long long m = red[0][tid];
        for (int w = 1; w < MT / 64; ++w) m = red[w][tid] < m ? red[w][tid] : m;
        const size_t o = (size_t)inst * nq + q0 + tid;
        idx[o] = (m == ~0ull) ? -1 : (int32_t)(m & 0xffffffffu);
        if (dist) dist[o] = (m == ~0ull) ? -1 : (int32_t)(m >> 32);
    }
}

// Generic descriptor length (any multiple of 4 bytes).
__global__ __launch_bounds__(MT) void match_hamming_any_kernel(const uint32_t *query, size_t stride_q, int nq, const uint32_t *train,
                                                               size_t stride_t, int nt, const int32_t *nt_each, int words,
                                                               int32_t *idx, int32_t *dist)
{
    const int inst = blockIdx.y, q = blockIdx.x, tid = threadIdx.x;
    const uint32_t *Q = query + (size_t)inst * stride_q + (size_t)q * words;
    const uint32_t *T = train + (size_t)inst * stride_t;
    if (nt_each) { const int c = nt_each[inst]; nt = c < nt ? (c < 0 ? 0 : c) : nt; }
    unsigned long long best = ~0ull;
    for (int j = tid; j < nt; j += MT) {
        unsigned d = 0;
        for (int w = 0; w < words; ++w) d += __popc(Q[w] ^ T[(size_t)j * words + w]);
        const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)j;
        best = key < best ? key : best;
    }
    __shared__ unsigned long long red[MT / 64];
    const unsigned long long m0 = wave_min_u64(best);
    if ((tid & 63) == 0) red[tid >> 6] = m0;
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = red[0];
        for (int w = 1; w < MT / 64; ++w) m = red[w] < m ? red[w] : m;
        const size_t o = (size_t)inst * nq + q;
        idx[o] = (m == ~0ull) ? -1 : (int32_t)(m & 0xffffffffu);
        if (dist) dist[o] = (m == ~0ull) ? -1 : (int32_t)(m >> 32);
    }
}

vslam_status match_hamming_dev(vslam_ctx *ctx, const uint8_t *query, size_t stride_q, int nq, const uint8_t *train, size_t stride_t,
                               int nt, const int32_t *nt_each, int desc_bytes, int batch, int32_t *idx, int32_t *dist)
{
    if (nq == 0) return VSLAM_OK;
    const int words = desc_bytes / 4;
    hipStream_t s = ctx->stream;
    const uint32_t *Q = reinterpret_cast<const uint32_t *>(query), *T = reinterpret_cast<const uint32_t *>(train);
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(train) | (stride_t & 15)) & 15) == 0;
    if (words == 8 && aligned16)
        hipLaunchKernelGGL(match_hamming_kernel<8>, dim3((nq + MQ - 1) / MQ, batch), dim3(MT), 0, s, Q, stride_q / 4, nq, T, stride_t / 4, nt,
                           nt_each, idx, dist);
    else
        hipLaunchKernelGGL(match_hamming_any_kernel, dim3(nq, batch), dim3(MT), 0, s, Q, stride_q / 4, nq, T, stride_t / 4, nt, nt_each,
                           words, idx, dist);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

static vslam_status check_match_args(vslam_ctx *ctx, const void *query, int nq, const void *train, int nt, int desc_bytes, const void *idx)
{
    if (nq < 0 || nt < 0) return fail(ctx, VSLAM_ERR_SHAPE, "match_hamming: nq=%d nt=%d", nq, nt);
    if (desc_bytes < 4 || desc_bytes % 4 != 0 || desc_bytes > 1024)
        return fail(ctx, VSLAM_ERR_SHAPE, "match_hamming: descriptor length %d is not a multiple of 4 bytes in [4, 1024]", desc_bytes);
    if ((nq > 0 && (!query || !idx)) || (nt > 0 && !train)) return fail(ctx, VSLAM_ERR_ARG, "match_hamming: null pointer");
    return VSLAM_OK;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_match_hamming(vslam_ctx *ctx, const uint8_t *query, int nq, const uint8_t *train, int nt, int desc_bytes,
                                            int32_t *train_idx, int32_t *distance)
{
    if (!ctx) return VSLAM_ERR_ARG;
    VSLAM_TRY(check_match_args(ctx, query, nq, train, nt, desc_bytes, train_idx));
    if (nq == 0) return VSLAM_OK;
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bq = (size_t)nq * desc_bytes, bt = (size_t)nt * desc_bytes;
    VSLAM_TRY(ws_reserve(ctx, Arena::pad(bq) + Arena::pad(bt + 16) + 2 * Arena::pad((size_t)nq * 4)));
    Arena ar(ctx->ws);
    uint8_t *dq = ar.take<uint8_t>(bq), *dt = ar.take<uint8_t>(bt + 16);
    int32_t *di = ar.take<int32_t>(nq), *dd = ar.take<int32_t>(nq);
    VSLAM_TRY(h2d(ctx, dq, query, bq));
    if (nt > 0) VSLAM_TRY(h2d(ctx, dt, train, bt));
    VSLAM_TRY(match_hamming_dev(ctx, dq, bq, nq, dt, bt, nt, nullptr, desc_bytes, 1, di, dd));
    VSLAM_TRY(d2h(ctx, train_idx, di, (size_t)nq * 4));
    if (distance) VSLAM_TRY(d2h(ctx, distance, dd, (size_t)nq * 4));
    return stream_finish(ctx);
}

extern "C" vslam_status vslam_match_hamming_batch_dev(vslam_ctx *ctx, const uint8_t *query, size_t stride_query, int nq,
                                                      const uint8_t *train, size_t stride_train, int nt, const int32_t *nt_each,
                                                      int desc_bytes, int batch, int32_t *train_idx, int32_t *distance)
{
    if (!ctx) return VSLAM_ERR_ARG;
    VSLAM_TRY(check_match_args(ctx, query, nq, train, nt, desc_bytes, train_idx));
    if (batch < 1 || batch > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "match_hamming: batch=%d", batch);
    if ((stride_query | stride_train) % 4 != 0 || ((reinterpret_cast<uintptr_t>(query) | reinterpret_cast<uintptr_t>(train)) & 3))
        return fail(ctx, VSLAM_ERR_ARG, "match_hamming: descriptors must be 4-byte aligned");
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return match_hamming_dev(ctx, query, stride_query, nq, train, stride_train, nt, nt_each, desc_bytes, batch, train_idx, distance);
}
