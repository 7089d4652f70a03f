// fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64), column-major.
//
// Serves every dense contraction on the hot path: the compact-WY trailing
// update of the square-root-covariance QR (reference src/gaussian.hpp:121-124,
// 515-516), S*C^T of affineTransform (src/gaussian.hpp:119), and the prior
// Hessian -S^-1 S^-T of logGaussian (src/gaussian.hpp:69-72).
//
// Tiling for 64-lane wavefronts: a 256-thread workgroup (4 waves as 2x2) owns
// a BM x BN tile of C; each wave owns (BM/2) x (BN/2) as 16x16 MFMA tiles,
// 4 f64 accumulators per lane per tile.  K advances in steps of 16 through LDS
// (A and B staged k-major with a 16-double row pad so the two 16-lane groups
// of a 32-lane LDS phase hit disjoint banks); the next K-step's global loads
// are issued into registers before the current step's MFMAs.
//
// The MFMA is issued with the operands swapped (rows of the instruction's
// result index n, columns index m) so that the 16 lanes sharing an accumulator
// register hold 16 consecutive rows of one C column: stores are 128-byte
// contiguous segments of the column-major C.
#include "linalg.hpp"

namespace vslam {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BK = 16;

struct GemmArgs {
    int M, N, K, lda, ldb, ldc, amode, b_upper, kchunk, tilesM, tilesN, c_upper;
    double alpha, beta;
    const double *A, *B;
    double *C;
    size_t sA, sB, sC, sSplit;
    const int *active;
};

// Tiles (tm, tn) of a BM x BN tiling that touch the upper triangle (tm BM <= tn BN + BN - 1), enumerated in bands of
// eight tile rows, each band column by column from its first such column on.  count_only: the number of tiles.
__host__ __device__ inline int upper_tile_walk(int l, int tilesM, int tilesN, int BM, int BN, int *tm_out, int *tn_out)
{
    int total = 0;
    for (int r0 = 0; r0 < tilesM; r0 += 8) {
        const int rows = (tilesM - r0) < 8 ? (tilesM - r0) : 8;
        int tn = (r0 * BM) / BN;                                   // first column whose tile (r0, tn) touches the triangle
        for (; tn < tilesN; ++tn) {                                // head of the band: columns that hold fewer than `rows` tiles
            const int tmax = (tn * BN + BN - 1) / BM;
            const int c = (tmax - r0 + 1) < rows ? (tmax - r0 + 1) : rows;
            if (c == rows) break;
            if (l >= 0 && l < c) { *tm_out = r0 + l; *tn_out = tn; return -1; }
            if (l >= 0) l -= c;
            total += c;
        }
        const int full = (tilesN - tn) * rows;                     // the rest of the band: `rows` tiles per column
        if (l >= 0 && l < full) { *tm_out = r0 + l % rows; *tn_out = tn + l / rows; return -1; }
        if (l >= 0) l -= full;
        total += full;
    }
    return total;
}
__device__ __forceinline__ void upper_tile_of(int l, int tilesM, int tilesN, int BM, int BN, int *tm, int *tn)
{
    *tm = 0; *tn = 0;
    upper_tile_walk(l, tilesM, tilesN, BM, BN, tm, tn);
}

template <bool TA>
__device__ __forceinline__ double fetch_a(const GemmArgs &g, const double *A, int m, int k)
{
    if (m >= g.M || k >= g.K) return 0.0;
    const int r = TA ? k : m, c = TA ? m : k;        // stored coordinates
    if (g.amode == A_UNIT_LOWER) {
        if (r < c) return 0.0;
        if (r == c) return 1.0;
    } else if (g.amode == A_UPPER) {
        if (r > c) return 0.0;
    }
    return A[(size_t)c * g.lda + r];
}
template <bool TB>
__device__ __forceinline__ double fetch_b(const GemmArgs &g, const double *B, int k, int n)
{
    if (k >= g.K || n >= g.N) return 0.0;
    const int r = TB ? n : k, c = TB ? k : n;
    if (g.b_upper && r > c) return 0.0;
    return B[(size_t)c * g.ldb + r];
}

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgs g)
{
    constexpr int LDA_S = BM + 16, LDB_S = BN + 16;       // (stride mod 32 doubles) == 16
    constexpr int TMw = BM / 32, TNw = BN / 32;           // 16x16 MFMA tiles per wave and dimension
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256; // staged elements per thread
    __shared__ __attribute__((aligned(16))) double smem[2 * BK * LDA_S + 2 * BK * LDB_S];   // two K-steps resident: one barrier per step
    double *const As2 = smem, *const Bs2 = smem + 2 * BK * LDA_S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Tile order.  Workgroup ids are dealt round-robin to the eight XCDs (an L2 each): renumbered so that an XCD works
    // through one contiguous run of tiles, and the tiles ordered in bands of eight tile rows swept along N, so that
    // the tiles an XCD has in flight share a few row blocks of A and column blocks of B that stay in its L2.
    // A triangular update launches workgroups for the tiles that touch the upper triangle only, in the same order.
    int tile_m, tile_n;
    {
        const int nb = (int)gridDim.x, x = (int)blockIdx.x & 7, q = nb >> 3, r = nb & 7;
        // (A triangular operand makes the tiles' K ranges run from a whole K down to one block along the order: a contiguous
        // run per XCD would give one XCD all the long tiles, so there the ids stay as dealt, every XCD a mix.)
        const bool ragged_k = g.amode == A_UPPER || g.b_upper;
        const int l = ragged_k ? (int)blockIdx.x : x * q + min(x, r) + ((int)blockIdx.x >> 3);
        if (g.c_upper) upper_tile_of(l, g.tilesM, g.tilesN, BM, BN, &tile_m, &tile_n);
        else {
            const int per_group = 8 * g.tilesN, group = l / per_group, first_m = group * 8;
            const int gsz = min(8, g.tilesM - first_m), in_group = l - group * per_group;
            tile_m = first_m + in_group % gsz;
            tile_n = in_group / gsz;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y, b = blockIdx.z;
    if (g.active && !g.active[b]) return;
    const double *A = g.A + (size_t)b * g.sA;
    const double *B = g.B + (size_t)b * g.sB;
    double *C = g.C + (size_t)b * g.sC + (size_t)split * g.sSplit;
    int k_begin = split * g.kchunk, k_end = min(g.K, k_begin + g.kchunk);
    // Triangular operands: the stretch of K over which an operand of this tile is identically zero is skipped
    // (products of triangular factors and inverses then cost the flops of their non-zero parts only).
    if (g.amode == A_UPPER) {
        if (TA) k_end = min(k_end, m0 + BM);                 // stored (k, m): zero for k > m
        else k_begin = max(k_begin, m0);                     // stored (m, k): zero for k < m
    }
    if (g.b_upper) {
        if (TB) k_begin = max(k_begin, n0);                  // stored (n, k): zero for k < n
        else k_end = min(k_end, n0 + BN);                    // stored (k, n): zero for k > n
    }

    const int wm = (wave & 1) * (BM / 2), wn = (wave >> 1) * (BN / 2);
    double4_t acc[TMw][TNw];
#pragma unroll
    for (int i = 0; i < TMw; ++i)
#pragma unroll
        for (int j = 0; j < TNw; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // C = alpha A B + beta C: the old values of this tile are requested before the products are formed, so that the
    // update at the end does not wait for them (a tile of a rank-64 update has only four K-steps to hide a load behind)
    const bool rmw = gridDim.y == 1 && g.beta != 0.0;
    double cold[TMw][TNw][4];
    if (rmw) {
#pragma unroll
        for (int i = 0; i < TMw; ++i)
#pragma unroll
            for (int j = 0; j < TNw; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = min(m0 + wm + 16 * i + (lane & 15), g.M - 1), n = min(n0 + wn + 16 * j + (lane >> 4) + 4 * r, g.N - 1);
                    cold[i][j][r] = C[(size_t)n * g.ldc + m];
                }
    }

    double ra[EA], rb[EB];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + e * 256;
            const int m = TA ? idx / BK : idx % BM, k = TA ? idx % BK : idx / BM;
            ra[e] = (k0 + k < k_end) ? fetch_a<TA>(g, A, m0 + m, k0 + k) : 0.0;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int idx = tid + e * 256;
            const int n = TB ? idx % BN : idx / BK, k = TB ? idx / BN : idx % BK;
            rb[e] = (k0 + k < k_end) ? fetch_b<TB>(g, B, k0 + k, n0 + n) : 0.0;
        }
    };
    auto store_tiles = [&](int buf) {
        double *As = As2 + buf * BK * LDA_S, *Bs = Bs2 + buf * BK * LDB_S;
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + e * 256;
            const int m = TA ? idx / BK : idx % BM, k = TA ? idx % BK : idx / BM;
            As[k * LDA_S + m] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int idx = tid + e * 256;
            const int n = TB ? idx % BN : idx / BK, k = TB ? idx / BN : idx % BK;
            Bs[k * LDB_S + n] = rb[e];
        }
    };

    if (k_begin < k_end) {
        load_tiles(k_begin);
        store_tiles(0);
        __syncthreads();
        if (k_begin + BK < k_end) load_tiles(k_begin + BK);   // in flight during the MFMAs below
        int buf = 0;
        for (int k0 = k_begin; k0 < k_end; k0 += BK, buf ^= 1) {
            const double *As = As2 + buf * BK * LDA_S, *Bs = Bs2 + buf * BK * LDB_S;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                const int krow = kk + (lane >> 4);
                double am[TMw], bn[TNw];
#pragma unroll
                for (int i = 0; i < TMw; ++i) am[i] = As[krow * LDA_S + wm + i * 16 + (lane & 15)];
#pragma unroll
                for (int j = 0; j < TNw; ++j) bn[j] = Bs[krow * LDB_S + wn + j * 16 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < TMw; ++i)
#pragma unroll
                    for (int j = 0; j < TNw; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[j], am[i], acc[i][j], 0, 0, 0);
            }
            if (k0 + BK < k_end) store_tiles(buf ^ 1);        // nobody reads that buffer during this step
            __syncthreads();
            if (k0 + 2 * BK < k_end) load_tiles(k0 + 2 * BK);
        }
    }
    // acc[i][j][r] = C[m0+wm+16i+(lane&15)][n0+wn+16j+(lane>>4)+4r]
    const bool split_mode = gridDim.y > 1;
#pragma unroll
    for (int i = 0; i < TMw; ++i)
#pragma unroll
        for (int j = 0; j < TNw; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + (lane & 15), n = n0 + wn + 16 * j + (lane >> 4) + 4 * r;
                if (m < g.M && n < g.N) {
                    double *p = C + (size_t)n * g.ldc + m;
                    if (split_mode) *p = acc[i][j][r];
                    else *p = rmw ? g.alpha * acc[i][j][r] + g.beta * cold[i][j][r] : g.alpha * acc[i][j][r];
                }
            }
}

template <int BM, int BN>
static void launch(const GemmArgs &a, dim3 grid, hipStream_t s, bool ta, bool tb)
{
    if (!ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, false>), grid, dim3(256), 0, s, a);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, false>), grid, dim3(256), 0, s, a);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, true, true>), grid, dim3(256), 0, s, a);
}

vslam_status gemm_f64(vslam_ctx *ctx, const GemmDesc &d) { return gemm_f64_on(ctx, d, ctx->stream); }

vslam_status gemm_f64_on(vslam_ctx *ctx, const GemmDesc &d, hipStream_t stream)
{
    if (d.M <= 0 || d.N <= 0 || d.batch <= 0) return VSLAM_OK;
    if (!d.A || !d.B || !d.C) return fail(ctx, VSLAM_ERR_ARG, "gemm: null operand");
    if (d.ksplit < 1 || d.batch > 65535 || d.ksplit > 65535) return fail(ctx, VSLAM_ERR_SHAPE, "gemm: bad split/batch");
    GemmArgs a;
    a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc;
    a.amode = (int)d.amode; a.b_upper = d.b_upper ? 1 : 0; a.c_upper = d.c_upper ? 1 : 0;
    a.alpha = d.alpha; a.beta = d.beta; a.A = d.A; a.B = d.B; a.C = d.C;
    a.sA = d.strideA; a.sB = d.strideB; a.sC = d.strideC; a.sSplit = d.strideSplit; a.active = d.active;
    int kchunk = (d.K + d.ksplit - 1) / d.ksplit;
    kchunk = ((kchunk + BK - 1) / BK) * BK;
    a.kchunk = kchunk > 0 ? kchunk : BK;
    const bool smallM = d.M <= 32;
    const int bm = smallM ? 32 : 64, bn = 64;
    a.tilesM = (d.M + bm - 1) / bm;
    a.tilesN = (d.N + bn - 1) / bn;
    int tmp_m = 0, tmp_n = 0;
    const int tiles = d.c_upper ? upper_tile_walk(-1, a.tilesM, a.tilesN, bm, bn, &tmp_m, &tmp_n) : a.tilesM * a.tilesN;
    dim3 grid(tiles, d.ksplit, d.batch);
    if (smallM) launch<32, 64>(a, grid, stream, d.transA, d.transB);
    else launch<64, 64>(a, grid, stream, d.transA, d.transB);
    VSLAM_HIP_TRY(ctx, hipGetLastError());
    return VSLAM_OK;
}

}  // namespace vslam
