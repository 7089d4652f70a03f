// Diagonal-block routines of the blocked upper Cholesky factorisation (newton.hip): shared by the stand-alone
// kernel for the first block and by the epilogue of the trailing-update GEMM (gemm.hip), whose workgroup for the
// next diagonal tile factors it as soon as it is updated -- under the rest of the update instead of after it.
#pragma once

#include "linalg.hpp"

namespace vslam {

constexpr int CB = 64;           // Cholesky block
constexpr int PLD = CB + 2;      // row stride of the block in LDS; even: the eight broadcast values of a row are two aligned 16-byte pairs apart
constexpr int PR = 8;
constexpr int POTF_LDS_DOUBLES = CB * PLD + CB;       // the block, then 1 / U_kk

// 1 / sqrt(d) from the hardware estimate (about 2^-26 relative) and one third-order correction
// y <- y + y e/2 (1 + 3e/4), e = 1 - d y^2: full double precision in five operations on the pivot chain.
__device__ __forceinline__ double potf_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * y, e = fma(-d * y, y, 1.0);
    return fma(h * e, fma(0.75, e, 1.0), y);
}

// One wave: upper Cholesky factor of the 64 x 64 block U (LDS, U[l][j] at l * PLD + j, identity beyond the
// matrix), in place, lane j owning column j.  Rows are finished eight at a time: the eight rows first receive all
// finished rows above them (left-looking; per finished row one own-column read and eight broadcast reads feed
// eight multiply-adds held in registers), then are factored among themselves with lane broadcasts.  Entries below
// the diagonal are neither meaningful on entry nor on return.  invd[k] = 1 / U_kk.  A non-positive pivot in the
// first jb rows stores its 1-based global column (j0 + row + 1) in *info (first one wins) and stays, unscaled, on the
// diagonal of U (the value the trust-region search shifts by, trs_search.hpp); the arithmetic carries on with NaNs.
__device__ __attribute__((noinline)) void potf2_lds(double *U, double *invd, int jb, int j0, int *info)
{
    const int lane = threadIdx.x & 63;
    bool bad = false;
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
            if (!(dkk > 0.0) && row < jb && !bad) { bad = true; if (lane == row) atomicCAS(info, 0, j0 + row + 1); }
            const double inv = rsqrt(dkk);                       // one reciprocal square root per row instead of a square root and a division
            // row `row` of U (meaningful for lanes >= row); a non-positive pivot is left on the diagonal as it is, for the caller to read
            const double u = (lane == row) ? (dkk > 0.0 ? dkk * inv : dkk) : acc[r] * inv;
            U[row * PLD + lane] = u;
            if (lane == row) invd[row] = inv;
#pragma unroll
            for (int r2 = r + 1; r2 < PR; ++r2) acc[r2] -= readlane_f64(u, kb + r2) * u;
        }
    }
}

// The same factorisation by the whole workgroup of four waves (every thread calls it; it holds its own
// barriers).  The rows are still finished eight at a time by wave 0, but of the finished rows only the last eight
// are applied by it (to the eight rows it factors next, in registers); the other three waves meanwhile apply the
// same eight finished rows to all rows further down, in LDS.  Wave 0 then finds its next eight rows up to date
// except for the group just finished -- its part of the work per group no longer grows with the group's position.
__device__ __forceinline__ void potf2_lds4(double *U, double *invd, int jb, int j0, int *info)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool bad = false;
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
                if (!(dkk > 0.0) && row < jb && !bad) { bad = true; if (lane == row) atomicCAS(info, 0, j0 + row + 1); }
                const double inv = rsqrt(dkk);
                const double u = (lane == row) ? (dkk > 0.0 ? dkk * inv : dkk) : acc[r] * inv;     // a non-positive pivot stays on the diagonal as it is
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

// Wave w of four: Yt = (U_ww)^-T for the w-th 16 x 16 diagonal sub-block of the factored block, written dense
// (Yt[i][k] at 256 w + 16 i + k, lower triangular) to dinv.  Lane c < 16 owns column c of U_ww^-1 and
// back-substitutes e_c with broadcast reads of U.  The row-panel solve multiplies by these instead of
// substituting through all 64 rows.
__device__ __forceinline__ void potf_dinv16(const double *U, const double *invd, double *dinv)
{
    const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3, c = lane & 15;
    const double *Ub = U + (16 * w) * PLD + 16 * w;
    double v[16];
#pragma unroll
    for (int r = 15; r >= 0; --r) {
        double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = r + 1; k < 16; ++k) s -= Ub[r * PLD + k] * v[k];
        v[r] = (r <= c) ? s * invd[16 * w + r] : 0.0;
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) dinv[256 * w + 16 * c + k] = v[k];
    }
}

}  // namespace vslam
