// Where the time of a rank-K update of a trailing matrix goes: C(upper tiles) -= P^T P with P the K rows above C
// (column-major, leading dimension n), 64 x 64 tiles, one workgroup per tile.  Each phase of the kernel can be
// switched off: old C read + write, operand loads into LDS, products.
//   hipcc -O3 --offload-arch=gfx950 syrk_probe.hip -o syrk_probe && ./syrk_probe [n] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int CB = 64;

// linear id -> tile of the upper triangle, column by column
__device__ __forceinline__ void tri_tile(int l, int *tm, int *tn)
{
    int n = (int)((sqrtf(8.0f * (float)l + 1.0f) - 1.0f) * 0.5f);
    while ((n + 1) * (n + 2) / 2 <= l) ++n;
    while (n * (n + 1) / 2 > l) --n;
    *tn = n; *tm = l - n * (n + 1) / 2;
}

template <int KC, bool DO_C, bool DO_LOAD, bool DO_MFMA, int WGS>
__global__ __launch_bounds__(256, WGS) void syrk_probe(const double *P, double *C, int N, int K, int lda, int swz)
{
    constexpr int SLD = KC + 2;
    __shared__ __align__(16) double Ps[2 * CB * SLD];
    double *const Pm = Ps, *const Pn = Ps + CB * SLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kr = lane >> 4;
    int l = blockIdx.x;
    if (swz) { const int nb = gridDim.x, x = l & 7, q = nb >> 3, r = nb & 7; l = x * q + min(x, r) + (l >> 3); }
    int tile_m, tile_n;
    tri_tile(l, &tile_m, &tile_n);
    const int m0 = tile_m * CB, n0 = tile_n * CB;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const bool diag = m0 == n0;
    double cold[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(m0 + wm + 16 * i + l15, N - 1), n = min(n0 + wn + 16 * j + kr + 4 * r, N - 1);
                cold[i][j][r] = DO_C ? C[(size_t)n * lda + m] : 0.0;
            }
    double4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    constexpr int EPT = CB * KC / 256;            // elements per thread and block
    constexpr int TPC = KC;                       // threads per column (lane = row of P)
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (DO_LOAD) {
            double ra[EPT], rb[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int idx = tid + 256 * e, c = idx / TPC, kk = idx % TPC;
                ra[e] = P[(size_t)min(m0 + c, N - 1) * lda + k0 + kk];
                if (!diag) rb[e] = P[(size_t)min(n0 + c, N - 1) * lda + k0 + kk];
            }
            if (k0 > 0) __syncthreads();
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int idx = tid + 256 * e, c = idx / TPC, kk = idx % TPC;
                Pm[c * SLD + kk] = ra[e];
                if (!diag) Pn[c * SLD + kk] = rb[e];
            }
        } else if (k0 == 0) {
            for (int e = tid; e < 2 * CB * SLD; e += 256) Ps[e] = 1e-3 * e;
        }
        __syncthreads();
        if (DO_MFMA) {
            const double *Bn = diag ? Pm : Pn;
#pragma unroll 4
            for (int kk = 0; kk < KC; kk += 4) {
                double am[2], bn[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) am[i] = Pm[(wm + 16 * i + l15) * SLD + kk + kr];
#pragma unroll
                for (int j = 0; j < 2; ++j) bn[j] = Bn[(wn + 16 * j + l15) * SLD + kk + kr];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[j], am[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    if (!DO_MFMA) acc[0][0][0] = Ps[tid];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + l15, n = n0 + wn + 16 * j + kr + 4 * r;
                if (m < N && n < N && (DO_C || (acc[i][j][r] == 123.456))) C[(size_t)n * lda + m] = cold[i][j][r] - acc[i][j][r];
            }
}

// as syrk_probe<16, true, true, true> with the next chunk's loads issued before the current chunk's products
template <int WGS>
__global__ __launch_bounds__(256, WGS) void syrk_probe_prefetch(const double *P, double *C, int N, int K, int lda)
{
    constexpr int KC = 16, SLD = KC + 2;
    __shared__ __align__(16) double Ps[2][2 * CB * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kr = lane >> 4;
    int tile_m, tile_n;
    tri_tile(blockIdx.x, &tile_m, &tile_n);
    const int m0 = tile_m * CB, n0 = tile_n * CB;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const bool diag = m0 == n0;
    double cold[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(m0 + wm + 16 * i + l15, N - 1), n = min(n0 + wn + 16 * j + kr + 4 * r, N - 1);
                cold[i][j][r] = C[(size_t)n * lda + m];
            }
    double4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (tid >> 4) + 16 * e, kk = tid & 15;
            ra[e] = P[(size_t)min(m0 + c, N - 1) * lda + k0 + kk];
            if (!diag) rb[e] = P[(size_t)min(n0 + c, N - 1) * lda + k0 + kk];
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (tid >> 4) + 16 * e, kk = tid & 15;
            Ps[buf][c * SLD + kk] = ra[e];
            if (!diag) Ps[buf][CB * SLD + c * SLD + kk] = rb[e];
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += KC, buf ^= 1) {
        if (k0 + KC < K) fetch(k0 + KC);
        const double *Pm = Ps[buf], *Bn = diag ? Ps[buf] : Ps[buf] + CB * SLD;
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            double am[2], bn[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) am[i] = Pm[(wm + 16 * i + l15) * SLD + kk + kr];
#pragma unroll
            for (int j = 0; j < 2; ++j) bn[j] = Bn[(wn + 16 * j + l15) * SLD + kk + kr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[j], am[i], acc[i][j], 0, 0, 0);
        }
        if (k0 + KC < K) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + l15, n = n0 + wn + 16 * j + kr + 4 * r;
                if (m < N && n < N) C[(size_t)n * lda + m] = cold[i][j][r] - acc[i][j][r];
            }
}

template <typename F>
static float time_it(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.0f / reps;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 6012, K = argc > 2 ? atoi(argv[2]) : 64;
    const int N = n - K, lda = n;                 // trailing block: rows/columns K .. n
    double *A;
    CK(hipMalloc(&A, (size_t)n * n * 8));
    std::vector<double> h((size_t)n * n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
    CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    const double *P = A + (size_t)K * lda;        // element (0, K): rows 0..K of the columns of the trailing block
    double *C = A + (size_t)K * lda + K;
    const int T = (N + CB - 1) / CB, tiles = T * (T + 1) / 2;
    const double flops = 2.0 * tiles * CB * CB * (double)K;
    printf("n=%d K=%d trailing %d tiles %d (%.2f GFLOP, C %.0f MB r+w)\n", n, K, N, tiles, flops * 1e-9, 2.0 * tiles * CB * CB * 8 * 1e-6);
#define RUN(name, KC, c, l, m, w, swz) do { \
        float us = time_it([&] { hipLaunchKernelGGL((syrk_probe<KC, c, l, m, w>), dim3(tiles), dim3(256), 0, 0, P, C, N, K, lda, swz); }, 20); \
        CK(hipGetLastError()); \
        printf("%-44s %8.1f us  %6.1f TFLOP/s\n", name, us, flops / us * 1e-6); } while (0)
    RUN("KC=64 all, 2 WG/CU", 64, true, true, true, 2, 0);
    RUN("KC=64 all, 2 WG/CU, xcd runs", 64, true, true, true, 2, 1);
    RUN("KC=64 no C", 64, false, true, true, 2, 0);
    RUN("KC=64 no loads (LDS preset)", 64, true, false, true, 2, 0);
    RUN("KC=64 no products", 64, true, true, false, 2, 0);
    RUN("KC=64 C only", 64, true, false, false, 2, 0);
    RUN("KC=64 products only", 64, false, false, true, 2, 0);
    RUN("KC=64 loads only", 64, false, true, false, 2, 0);
    RUN("KC=32 all, 4 WG/CU", 32, true, true, true, 4, 0);
    RUN("KC=32 all, 4 WG/CU, xcd runs", 32, true, true, true, 4, 1);
    RUN("KC=32 products only", 32, false, false, true, 4, 0);
    RUN("KC=16 all, 4 WG/CU", 16, true, true, true, 4, 0);
    RUN("KC=16 all, 5 WG/CU", 16, true, true, true, 5, 0);
    RUN("KC=16 all, 6 WG/CU", 16, true, true, true, 6, 0);
#define RUNP(name, w) do { \
        float us = time_it([&] { hipLaunchKernelGGL((syrk_probe_prefetch<w>), dim3(tiles), dim3(256), 0, 0, P, C, N, K, lda); }, 20); \
        CK(hipGetLastError()); \
        printf("%-44s %8.1f us  %6.1f TFLOP/s\n", name, us, flops / us * 1e-6); } while (0)
    RUNP("KC=16 prefetch + two LDS buffers, 3 WG/CU", 3);
    RUNP("KC=16 prefetch + two LDS buffers, 4 WG/CU", 4);
    RUNP("KC=16 prefetch + two LDS buffers, 5 WG/CU", 5);
    return 0;
}
