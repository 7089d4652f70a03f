// How fast can one workgroup factor a 64 x 64 diagonal block (upper Cholesky, fp64, block in LDS)?  Variants of the
// routine in visualslam_amd/csrc/potf.hpp timed in isolation: shader-clock cycles per call, one workgroup, block
// restored from a copy before every call.
//   hipcc -O3 --offload-arch=gfx950 potf2_probe.hip -o potf2_probe && ./potf2_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int CB = 64, PLD = CB + 2, PR = 8;

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * y, e = fma(-d * y, y, 1.0);
    return fma(h * e, fma(0.75, e, 1.0), y);
}

// V0: the library's routine: one wave, left-looking in groups of eight rows
__device__ __attribute__((noinline)) void potf2_v0(double *U, double *invd)
{
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x & 63;
    for (int kb = 0; kb < CB; kb += PR) {
        double acc[PR];
#pragma unroll
        for (int r = 0; r < PR; ++r) acc[r] = U[(kb + r) * PLD + lane];
#pragma unroll 4
        for (int l = 0; l < kb; ++l) {
            const double own = U[l * PLD + lane];
            const double2 *bc = reinterpret_cast<const double2 *>(U + l * PLD + kb);
            const double2 b0 = bc[0], b1 = bc[1], b2 = bc[2], b3 = bc[3];
            acc[0] -= b0.x * own; acc[1] -= b0.y * own; acc[2] -= b1.x * own; acc[3] -= b1.y * own;
            acc[4] -= b2.x * own; acc[5] -= b2.y * own; acc[6] -= b3.x * own; acc[7] -= b3.y * own;
        }
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const int row = kb + r;
            const double dkk = readlane_f64(acc[r], row);
            const double inv = rsqrt(dkk);
            const double u = (lane == row) ? dkk * inv : acc[r] * inv;
            U[row * PLD + lane] = u;
            if (lane == row) invd[row] = inv;
#pragma unroll
            for (int r2 = r + 1; r2 < PR; ++r2) acc[r2] -= readlane_f64(u, kb + r2) * u;
        }
    }
}

// V1: four waves.  Per group of 16 rows: the strip (16 rows x the columns from the group on) first receives the
// finished rows above it on the matrix cores (wave w owns the 16 columns 16 w ..), then wave 0 factors the 16 x 16
// diagonal sub-block in registers and every wave substitutes its own 16 columns of the strip.
__device__ __attribute__((noinline)) void potf2_v1(double *U, double *invd)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, kr = lane >> 4;
#pragma unroll 1
    for (int o = 0; o < CB; o += 16) {
        const int wo = o >> 4;
        // (a) strip rows o..o+15, columns 16 wave ..: -= U[0:o, rows]^T U[0:o, cols]   (waves left of the diagonal idle)
        if (wave >= wo && o > 0) {
            double4_t t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = U[(o + 4 * r + kr) * PLD + 16 * wave + c];
            for (int k0 = 0; k0 < o; k0 += 4)
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(k0 + kr) * PLD + o + c], U[(k0 + kr) * PLD + 16 * wave + c], t, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) U[(o + 4 * r + kr) * PLD + 16 * wave + c] = t[r];
        }
        __syncthreads();
        // (b) wave wo: the diagonal sub-block
        if (wave == wo) {
            double a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = U[(o + r) * PLD + o + c];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double d = readlane_f64(a[k], k);
                const double inv = fast_rsqrt(d);
                const double u = (c == k) ? d * inv : a[k] * inv;
                a[k] = u;
#pragma unroll
                for (int r = k + 1; r < 16; ++r) a[r] -= readlane_f64(u, r) * u;
                if (lane == k) invd[o + k] = inv;
            }
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r <= c) U[(o + r) * PLD + o + c] = a[r];
            }
        }
        __syncthreads();
        // (c) waves right of the diagonal: their 16 columns of the strip, X = U_ii^-T R, a lane per column (four copies)
        if (wave > wo) {
            double x[16];
            const int col = 16 * wave + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = U[(o + r) * PLD + col];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                x[k] *= invd[o + k];
#pragma unroll
                for (int r = k + 1; r < 16; ++r) x[r] -= U[(o + k) * PLD + o + r] * x[k];
            }
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) U[(o + r) * PLD + col] = x[r];
            }
        }
        __syncthreads();
    }
}

// V2: as V1, but the strip's substitution uses the inverse of the diagonal sub-block on the matrix cores
// (the inverse is wanted by the row-panel solve anyway): wave wo back-substitutes the 16 unit vectors after its
// factorisation, the others multiply.
__device__ __attribute__((noinline)) void potf2_v2(double *U, double *invd, double *Yt /* [64][17] */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, kr = lane >> 4;
#pragma unroll 1
    for (int o = 0; o < CB; o += 16) {
        const int wo = o >> 4;
        if (wave >= wo && o > 0) {
            double4_t t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = U[(o + 4 * r + kr) * PLD + 16 * wave + c];
            for (int k0 = 0; k0 < o; k0 += 4)
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(k0 + kr) * PLD + o + c], U[(k0 + kr) * PLD + 16 * wave + c], t, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) U[(o + 4 * r + kr) * PLD + 16 * wave + c] = t[r];
        }
        __syncthreads();
        if (wave == wo) {
            double a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = U[(o + r) * PLD + o + c];
            double myinv = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double d = readlane_f64(a[k], k);
                const double inv = fast_rsqrt(d);
                const double u = (c == k) ? d * inv : a[k] * inv;
                a[k] = u;
#pragma unroll
                for (int r = k + 1; r < 16; ++r) a[r] -= readlane_f64(u, r) * u;
                if (lane == k) { invd[o + k] = inv; }
                if (c == k) myinv = inv;
            }
            // inverse: lane c owns column c of V = U_ii^-1; row r of U_ii as lane broadcasts
            double v[16];
#pragma unroll
            for (int r = 15; r >= 0; --r) {
                double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
                for (int k = r + 1; k < 16; ++k) s -= readlane_f64(a[r], k) * v[k];
                v[r] = (r <= c) ? s * readlane_f64(myinv, r) : 0.0;
            }
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r <= c) U[(o + r) * PLD + o + c] = a[r];
                    Yt[(o + c) * 17 + r] = v[r];                 // Yt[i][k] = V[k][i]
                }
            }
        }
        __syncthreads();
        if (wave > wo) {
            double4_t x = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4)
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(Yt[(o + c) * 17 + k0 + kr], U[(o + k0 + kr) * PLD + 16 * wave + c], x, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) U[(o + 4 * r + kr) * PLD + 16 * wave + c] = x[r];
        }
        __syncthreads();
    }
}

// value of lane (row of 16 lanes, position R) for every lane of that row: DPP row_share, stays in vector registers
template <int R>
__device__ __forceinline__ double rowshare_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + R, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + R, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ void chol16_step(double (&a)[16], int c, double *invd_o, int lane)
{
    const double d = rowshare_f64<K>(a[K]);
    const double inv = fast_rsqrt(d);
    const double u = (c == K) ? d * inv : a[K] * inv;
    a[K] = u;
    if (lane == K) invd_o[K] = inv;
    if constexpr (K + 1 < 16) a[K + 1] -= rowshare_f64<K + 1 < 16 ? K + 1 : 15>(u) * u;
    if constexpr (K + 2 < 16) a[K + 2] -= rowshare_f64<K + 2 < 16 ? K + 2 : 15>(u) * u;
    if constexpr (K + 3 < 16) a[K + 3] -= rowshare_f64<K + 3 < 16 ? K + 3 : 15>(u) * u;
    if constexpr (K + 4 < 16) a[K + 4] -= rowshare_f64<K + 4 < 16 ? K + 4 : 15>(u) * u;
    if constexpr (K + 5 < 16) a[K + 5] -= rowshare_f64<K + 5 < 16 ? K + 5 : 15>(u) * u;
    if constexpr (K + 6 < 16) a[K + 6] -= rowshare_f64<K + 6 < 16 ? K + 6 : 15>(u) * u;
    if constexpr (K + 7 < 16) a[K + 7] -= rowshare_f64<K + 7 < 16 ? K + 7 : 15>(u) * u;
    if constexpr (K + 8 < 16) a[K + 8] -= rowshare_f64<K + 8 < 16 ? K + 8 : 15>(u) * u;
    if constexpr (K + 9 < 16) a[K + 9] -= rowshare_f64<K + 9 < 16 ? K + 9 : 15>(u) * u;
    if constexpr (K + 10 < 16) a[K + 10] -= rowshare_f64<K + 10 < 16 ? K + 10 : 15>(u) * u;
    if constexpr (K + 11 < 16) a[K + 11] -= rowshare_f64<K + 11 < 16 ? K + 11 : 15>(u) * u;
    if constexpr (K + 12 < 16) a[K + 12] -= rowshare_f64<K + 12 < 16 ? K + 12 : 15>(u) * u;
    if constexpr (K + 13 < 16) a[K + 13] -= rowshare_f64<K + 13 < 16 ? K + 13 : 15>(u) * u;
    if constexpr (K + 14 < 16) a[K + 14] -= rowshare_f64<K + 14 < 16 ? K + 14 : 15>(u) * u;
    if constexpr (K + 15 < 16) a[K + 15] -= rowshare_f64<15>(u) * u;
}
__device__ __forceinline__ void chol16_regs(double (&a)[16], int c, double *invd_o, int lane)
{
    chol16_step<0>(a, c, invd_o, lane); chol16_step<1>(a, c, invd_o, lane); chol16_step<2>(a, c, invd_o, lane); chol16_step<3>(a, c, invd_o, lane);
    chol16_step<4>(a, c, invd_o, lane); chol16_step<5>(a, c, invd_o, lane); chol16_step<6>(a, c, invd_o, lane); chol16_step<7>(a, c, invd_o, lane);
    chol16_step<8>(a, c, invd_o, lane); chol16_step<9>(a, c, invd_o, lane); chol16_step<10>(a, c, invd_o, lane); chol16_step<11>(a, c, invd_o, lane);
    chol16_step<12>(a, c, invd_o, lane); chol16_step<13>(a, c, invd_o, lane); chol16_step<14>(a, c, invd_o, lane); chol16_step<15>(a, c, invd_o, lane);
}

// V3: V1 with the pivot row of the diagonal sub-block handed round by DPP row_share (vector registers only)
__device__ long long g_phase[4];
#define STAMP(i) do { if (tid == 0) { const long long now_ = __builtin_readcyclecounter(); g_phase[i] += now_ - tprev; tprev = now_; } } while (0)
__device__ __attribute__((noinline)) void potf2_v3(double *U, double *invd)
{
    long long tprev = __builtin_readcyclecounter();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, kr = lane >> 4;
#pragma unroll 1
    for (int o = 0; o < CB; o += 16) {
        const int wo = o >> 4;
        if (wave >= wo && o > 0) {
            double4_t t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = U[(o + 4 * r + kr) * PLD + 16 * wave + c];
            for (int k0 = 0; k0 < o; k0 += 4)
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(k0 + kr) * PLD + o + c], U[(k0 + kr) * PLD + 16 * wave + c], t, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) U[(o + 4 * r + kr) * PLD + 16 * wave + c] = t[r];
        }
        __syncthreads();
        STAMP(0);
        if (wave == wo) {
            double a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = U[(o + r) * PLD + o + c];
            chol16_regs(a, c, invd + o, lane);
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r <= c) U[(o + r) * PLD + o + c] = a[r];
            }
        }
        __syncthreads();
        STAMP(1);
        if (wave > wo) {
            double x[16];
            const int col = 16 * wave + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = U[(o + r) * PLD + col];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                x[k] *= invd[o + k];
#pragma unroll
                for (int r = k + 1; r < 16; ++r) x[r] -= U[(o + k) * PLD + o + r] * x[k];
            }
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) U[(o + r) * PLD + col] = x[r];
            }
        }
        __syncthreads();
        STAMP(2);
    }
}

// V4: the library's routine (potf2_lds4 of visualslam_amd/csrc/potf.hpp, without the pivot check): wave 0 factors eight rows
// at a time and applies only the last eight finished rows to them; the other waves apply those to the rows further down
__device__ __forceinline__ void potf2_v4(double *U, double *invd)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int kb = 0; kb < CB; kb += PR) {
        if (wave == 0) {
            double acc[PR];
#pragma unroll
            for (int r = 0; r < PR; ++r) acc[r] = U[(kb + r) * PLD + lane];
            if (kb > 0) {
#pragma unroll
                for (int l = kb - PR; l < kb; ++l) {
                    const double own = U[l * PLD + lane];
                    const double2 *bc = reinterpret_cast<const double2 *>(U + l * PLD + kb);
                    const double2 b0 = bc[0], b1 = bc[1], b2 = bc[2], b3 = bc[3];
                    acc[0] -= b0.x * own; acc[1] -= b0.y * own; acc[2] -= b1.x * own; acc[3] -= b1.y * own;
                    acc[4] -= b2.x * own; acc[5] -= b2.y * own; acc[6] -= b3.x * own; acc[7] -= b3.y * own;
                }
            }
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                const int row = kb + r;
                const double dkk = readlane_f64(acc[r], row);
                const double inv = rsqrt(dkk);
                const double u = (lane == row) ? dkk * inv : acc[r] * inv;
                U[row * PLD + lane] = u;
                if (lane == row) invd[row] = inv;
#pragma unroll
                for (int r2 = r + 1; r2 < PR; ++r2) acc[r2] -= readlane_f64(u, kb + r2) * u;
            }
        } else if (kb > 0 && kb + PR < CB) {
            // rows kb + 8 and beyond receive the rows kb - 8 ... kb - 1 (finished one barrier ago)
            double own[PR];
#pragma unroll
            for (int q = 0; q < PR; ++q) own[q] = U[(kb - PR + q) * PLD + lane];
            for (int r = kb + PR + (wave - 1); r < CB; r += 3) {
                const double2 *bc = reinterpret_cast<const double2 *>(U + (kb - PR) * PLD + (r & ~1));
                double a = U[r * PLD + lane];
#pragma unroll
                for (int q = 0; q < PR; ++q) {
                    const double2 b = bc[q * (PLD / 2)];
                    a -= ((r & 1) ? b.y : b.x) * own[q];
                }
                U[r * PLD + lane] = a;
            }
        }
        __syncthreads();
    }
}

__device__ long long g_phase5[4];
#define STAMP5(i) do { if (tid == 0) { const long long now_ = __builtin_readcyclecounter(); g_phase5[i] += now_ - tprev; tprev = now_; } } while (0)
// V5: V4 with the short reciprocal square root and phase counters: wave 0 factors eight rows
// at a time and applies only the last eight finished rows to them; the other waves apply those to the rows further down
__device__ __forceinline__ void potf2_v5(double *U, double *invd)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    long long tprev = __builtin_readcyclecounter();
    for (int kb = 0; kb < CB; kb += PR) {
        if (wave == 0) {
            double acc[PR];
#pragma unroll
            for (int r = 0; r < PR; ++r) acc[r] = U[(kb + r) * PLD + lane];
            if (kb > 0) {
#pragma unroll
                for (int l = kb - PR; l < kb; ++l) {
                    const double own = U[l * PLD + lane];
                    const double2 *bc = reinterpret_cast<const double2 *>(U + l * PLD + kb);
                    const double2 b0 = bc[0], b1 = bc[1], b2 = bc[2], b3 = bc[3];
                    acc[0] -= b0.x * own; acc[1] -= b0.y * own; acc[2] -= b1.x * own; acc[3] -= b1.y * own;
                    acc[4] -= b2.x * own; acc[5] -= b2.y * own; acc[6] -= b3.x * own; acc[7] -= b3.y * own;
                }
            }
            asm volatile("" : "+v"(acc[0]), "+v"(acc[7]));
            STAMP5(0);
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                const int row = kb + r;
                const double dkk = readlane_f64(acc[r], row);
                const double inv = fast_rsqrt(dkk);
                const double u = (lane == row) ? dkk * inv : acc[r] * inv;
                U[row * PLD + lane] = u;
                if (lane == row) invd[row] = inv;
#pragma unroll
                for (int r2 = r + 1; r2 < PR; ++r2) acc[r2] -= readlane_f64(u, kb + r2) * u;
            }
            STAMP5(1);
        } else if (kb > 0 && kb + PR < CB) {
            // rows kb + 8 and beyond receive the rows kb - 8 ... kb - 1 (finished one barrier ago)
            double own[PR];
#pragma unroll
            for (int q = 0; q < PR; ++q) own[q] = U[(kb - PR + q) * PLD + lane];
            for (int r = kb + PR + (wave - 1); r < CB; r += 3) {
                const double2 *bc = reinterpret_cast<const double2 *>(U + (kb - PR) * PLD + (r & ~1));
                double a = U[r * PLD + lane];
#pragma unroll
                for (int q = 0; q < PR; ++q) {
                    const double2 b = bc[q * (PLD / 2)];
                    a -= ((r & 1) ? b.y : b.x) * own[q];
                }
                U[r * PLD + lane] = a;
            }
        }
        __syncthreads();
        STAMP5(2);
    }
}

// V6: as V4, but wave 0 keeps two groups of eight rows in registers: the one it factors and the next one, which
// receives every new pivot row at once (beside the pivot chain) and, row by row in the same steps, the eight rows finished
// one group earlier (from LDS); the other waves apply a finished group to the rows two groups further down and beyond.
__device__ __forceinline__ void potf2_v6(double *U, double *invd)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double cur[PR], nxt[PR];
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < PR; ++r) { cur[r] = U[r * PLD + lane]; nxt[r] = U[(PR + r) * PLD + lane]; }
    }
    for (int kb = 0; kb < CB; kb += PR) {
        if (wave == 0) {
            const bool has_next = kb + PR < CB, has_prev = kb > 0;
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                const int row = kb + r;
                const double dkk = readlane_f64(cur[r], row);
                const double inv = fast_rsqrt(dkk);
                const double u = (lane == row) ? dkk * inv : cur[r] * inv;
                U[row * PLD + lane] = u;
                if (lane == row) invd[row] = inv;
#pragma unroll
                for (int r2 = r + 1; r2 < PR; ++r2) cur[r2] -= readlane_f64(u, kb + r2) * u;
                if (has_next) {
#pragma unroll
                    for (int q = 0; q < PR; ++q) nxt[q] -= readlane_f64(u, kb + PR + q) * u;
                    if (has_prev) {                                  // row r of the group finished before this one
                        const int l = kb - PR + r;
                        const double own = U[l * PLD + lane];
                        const double2 *bc = reinterpret_cast<const double2 *>(U + l * PLD + kb + PR);
                        const double2 b0 = bc[0], b1 = bc[1], b2 = bc[2], b3 = bc[3];
                        nxt[0] -= b0.x * own; nxt[1] -= b0.y * own; nxt[2] -= b1.x * own; nxt[3] -= b1.y * own;
                        nxt[4] -= b2.x * own; nxt[5] -= b2.y * own; nxt[6] -= b3.x * own; nxt[7] -= b3.y * own;
                    }
                }
            }
        } else if (kb > 0 && kb + 2 * PR < CB) {
            // rows kb + 16 and beyond receive the rows kb - 8 ... kb - 1 (finished one barrier ago)
            double own[PR];
#pragma unroll
            for (int q = 0; q < PR; ++q) own[q] = U[(kb - PR + q) * PLD + lane];
            for (int r = kb + 2 * PR + (wave - 1); r < CB; r += 3) {
                const double2 *bc = reinterpret_cast<const double2 *>(U + (kb - PR) * PLD + (r & ~1));
                double a = U[r * PLD + lane];
#pragma unroll
                for (int q = 0; q < PR; ++q) {
                    const double2 b = bc[q * (PLD / 2)];
                    a -= ((r & 1) ? b.y : b.x) * own[q];
                }
                U[r * PLD + lane] = a;
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < PR; ++r) cur[r] = nxt[r];
            if (kb + 2 * PR < CB) {
#pragma unroll
                for (int r = 0; r < PR; ++r) nxt[r] = U[(kb + 2 * PR + r) * PLD + lane];
            }
        }
    }
}

// V7: as V6 with the new pivot rows reaching the next group one step late, as broadcast reads of the stored row;
// V6: as V4, but wave 0 keeps two groups of eight rows in registers: the one it factors and the next one, which
// receives every new pivot row at once (beside the pivot chain) and, row by row in the same steps, the eight rows finished
// one group earlier (from LDS); the other waves apply a finished group to the rows two groups further down and beyond.
__device__ __forceinline__ void potf2_v7(double *U, double *invd)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double cur[PR], nxt[PR];
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < PR; ++r) { cur[r] = U[r * PLD + lane]; nxt[r] = U[(PR + r) * PLD + lane]; }
    }
    for (int kb = 0; kb < CB; kb += PR) {
        if (wave == 0) {
            const bool has_next = kb + PR < CB, has_prev = kb > 0;
            double uprev = 0.0;
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                const int row = kb + r;
                const double dkk = readlane_f64(cur[r], row);
                const double inv = fast_rsqrt(dkk);
                const double u = (lane == row) ? dkk * inv : cur[r] * inv;
                U[row * PLD + lane] = u;
                if (lane == row) invd[row] = inv;
#pragma unroll
                for (int r2 = r + 1; r2 < PR; ++r2) cur[r2] -= readlane_f64(u, kb + r2) * u;
                if (has_next) {
                    if (r > 0) {                                     // the pivot row of the step before: stored by now
                        const double2 *pc = reinterpret_cast<const double2 *>(U + (row - 1) * PLD + kb + PR);
                        const double2 p0 = pc[0], p1 = pc[1], p2 = pc[2], p3 = pc[3];
                        nxt[0] -= p0.x * uprev; nxt[1] -= p0.y * uprev; nxt[2] -= p1.x * uprev; nxt[3] -= p1.y * uprev;
                        nxt[4] -= p2.x * uprev; nxt[5] -= p2.y * uprev; nxt[6] -= p3.x * uprev; nxt[7] -= p3.y * uprev;
                    }
                    uprev = u;
                    if (has_prev) {                                  // row r of the group finished before this one
                        const int l = kb - PR + r;
                        const double own = U[l * PLD + lane];
                        const double2 *bc = reinterpret_cast<const double2 *>(U + l * PLD + kb + PR);
                        const double2 b0 = bc[0], b1 = bc[1], b2 = bc[2], b3 = bc[3];
                        nxt[0] -= b0.x * own; nxt[1] -= b0.y * own; nxt[2] -= b1.x * own; nxt[3] -= b1.y * own;
                        nxt[4] -= b2.x * own; nxt[5] -= b2.y * own; nxt[6] -= b3.x * own; nxt[7] -= b3.y * own;
                    }
                }
            }
            if (has_next) {                                          // the group's last pivot row
                const double2 *pc = reinterpret_cast<const double2 *>(U + (kb + PR - 1) * PLD + kb + PR);
                const double2 p0 = pc[0], p1 = pc[1], p2 = pc[2], p3 = pc[3];
                nxt[0] -= p0.x * uprev; nxt[1] -= p0.y * uprev; nxt[2] -= p1.x * uprev; nxt[3] -= p1.y * uprev;
                nxt[4] -= p2.x * uprev; nxt[5] -= p2.y * uprev; nxt[6] -= p3.x * uprev; nxt[7] -= p3.y * uprev;
            }
        } else if (kb > 0 && kb + 2 * PR < CB) {
            // rows kb + 16 and beyond receive the rows kb - 8 ... kb - 1 (finished one barrier ago)
            double own[PR];
#pragma unroll
            for (int q = 0; q < PR; ++q) own[q] = U[(kb - PR + q) * PLD + lane];
            for (int r = kb + 2 * PR + (wave - 1); r < CB; r += 3) {
                const double2 *bc = reinterpret_cast<const double2 *>(U + (kb - PR) * PLD + (r & ~1));
                double a = U[r * PLD + lane];
#pragma unroll
                for (int q = 0; q < PR; ++q) {
                    const double2 b = bc[q * (PLD / 2)];
                    a -= ((r & 1) ? b.y : b.x) * own[q];
                }
                U[r * PLD + lane] = a;
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < PR; ++r) cur[r] = nxt[r];
            if (kb + 2 * PR < CB) {
#pragma unroll
                for (int r = 0; r < PR; ++r) nxt[r] = U[(kb + 2 * PR + r) * PLD + lane];
            }
        }
    }
}

// V8: wave 0 only factors (eight pivots per group, rows complete in LDS when it loads them); after each group all four
// waves apply it to the next eight rows (two rows a wave, second barrier), then waves 1-3 apply it to the rows beyond
// while wave 0 is already at the next group's pivots.
__device__ __forceinline__ void potf2_v8(double *U, double *invd)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto apply = [&](int kb, int r) {               // row r -= the contribution of the finished rows kb ... kb + 7
        const double2 *bc = reinterpret_cast<const double2 *>(U + kb * PLD + (r & ~1));
        double a = U[r * PLD + lane];
#pragma unroll
        for (int q = 0; q < PR; ++q) {
            const double2 b = bc[q * (PLD / 2)];
            a -= ((r & 1) ? b.y : b.x) * U[(kb + q) * PLD + lane];
        }
        U[r * PLD + lane] = a;
    };
    for (int kb = 0; kb < CB; kb += PR) {
        if (wave == 0) {
            double acc[PR];
#pragma unroll
            for (int r = 0; r < PR; ++r) acc[r] = U[(kb + r) * PLD + lane];
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                const int row = kb + r;
                const double dkk = readlane_f64(acc[r], row);
                const double inv = fast_rsqrt(dkk);
                const double u = (lane == row) ? dkk * inv : acc[r] * inv;
                U[row * PLD + lane] = u;
                if (lane == row) invd[row] = inv;
#pragma unroll
                for (int r2 = r + 1; r2 < PR; ++r2) acc[r2] -= readlane_f64(u, kb + r2) * u;
            }
        }
        __syncthreads();
        if (kb + PR < CB) {
            apply(kb, kb + PR + 2 * wave);
            apply(kb, kb + PR + 2 * wave + 1);
            __syncthreads();
            if (wave > 0)
                for (int r = kb + 2 * PR + (wave - 1); r < CB; r += 3) apply(kb, r);
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void probe(const double *A, double *out, long long *cycles, int reps)
{
    __shared__ __align__(16) double U[CB * PLD + CB];
    __shared__ double Yt[64 * 17];
    const int tid = threadIdx.x;
    long long total = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = tid; e < CB * CB; e += 256) { const int r = e & 63, c = e >> 6; U[r * PLD + c] = (r <= c) ? A[(size_t)c * CB + r] : 0.0; }
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        if (V == 0) potf2_v0(U, U + CB * PLD);
        if (V == 1) potf2_v1(U, U + CB * PLD);
        if (V == 2) potf2_v2(U, U + CB * PLD, Yt);
        if (V == 3) potf2_v3(U, U + CB * PLD);
        if (V == 4) potf2_v4(U, U + CB * PLD);
        if (V == 5) potf2_v5(U, U + CB * PLD);
        if (V == 6) potf2_v6(U, U + CB * PLD);
        if (V == 7) potf2_v7(U, U + CB * PLD);
        if (V == 8) potf2_v8(U, U + CB * PLD);
        __syncthreads();
        total += __builtin_readcyclecounter() - t0;
    }
    if (tid == 0) cycles[0] = total / reps;
    for (int e = tid; e < CB * CB; e += 256) { const int r = e & 63, c = e >> 6; out[(size_t)c * CB + r] = (r <= c) ? U[r * PLD + c] : 0.0; }
}

int main()
{
    const int n = CB;
    std::vector<double> G((size_t)n * n), H((size_t)n * n), R((size_t)n * n);
    srand(1);
    for (auto &g : G) g = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = 0; k < n; ++k) s += G[(size_t)k * n + i] * G[(size_t)k * n + j] / n;
            H[(size_t)j * n + i] = s;
        }
    // host reference
    R = H;
    for (int k = 0; k < n; ++k) {
        for (int l = 0; l < k; ++l)
            for (int j = k; j < n; ++j) R[(size_t)j * n + k] -= R[(size_t)k * n + l] * R[(size_t)j * n + l];
        const double d = std::sqrt(R[(size_t)k * n + k]);
        for (int j = k; j < n; ++j) R[(size_t)j * n + k] = (j == k) ? d : R[(size_t)j * n + k] / d;
    }
    double *dA, *dO;
    long long *dC;
    CK(hipMalloc(&dA, H.size() * 8)); CK(hipMalloc(&dO, H.size() * 8)); CK(hipMalloc(&dC, 8));
    CK(hipMemcpy(dA, H.data(), H.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> O(H.size());
    auto run = [&](const char *name, void (*k)(const double *, double *, long long *, int)) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dA, dO, dC, 50);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dA, dO, dC, 200);
        CK(hipDeviceSynchronize());
        long long cyc = 0;
        CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost));
        double err = 0.0;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i <= j; ++i) err = std::fmax(err, std::fabs(O[(size_t)j * n + i] - R[(size_t)j * n + i]));   // host: U(i, j) = L(j, i) at [j n + i]
        printf("%-60s %8lld shader cycles per call, max error %.2e\n", name, cyc, err);
    };
    run("V0 one wave, left-looking in groups of eight rows", probe<0>);
    run("V1 four waves, 16-row strips: MFMA + registers + lanes", probe<1>);
    run("V2 as V1 with the sub-block inverse and MFMA strips", probe<2>);
    run("V3 as V1, pivot rows by DPP row_share", probe<3>);
    run("V4 wave 0 factors, the other waves update the rows below", probe<4>);
    run("V5 as V4, short reciprocal square root, phase counters", probe<5>);
    run("V6 wave 0 holds two groups, updates beside the pivot chain", probe<6>);
    run("V7 as V6, new pivot rows reach the next group through LDS", probe<7>);
    run("V8 wave 0 only factors; all waves update the next eight rows", probe<8>);
    {
        long long ph[4];
        CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase5), sizeof ph));
        printf("V5 phases over 250 calls (wave 0): applying the last eight rows %lld, eight pivots %lld, barrier %lld cycles per call\n", ph[0] / 250, ph[1] / 250, ph[2] / 250);
    }
    {
        long long ph[4];
        CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof ph));
        printf("V3 phases over 250 calls: strip update %lld, diagonal sub-block %lld, strip substitution %lld cycles per call\n", ph[0] / 250, ph[1] / 250, ph[2] / 250);
    }
    return 0;
}
