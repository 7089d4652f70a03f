// Column-block factorisation kernel of the blocked Householder QR (qr_block.hip), as used by qr.hip.
#pragma once

#include "linalg.hpp"

namespace vslam {

constexpr int QR_NB = 32;                             // panel width
constexpr int QR_MSTRIDE = QR_NB * QR_NB + QR_NB;     // doubles reserved per panel for its compact-WY T (column-major, ld 32)
constexpr int QB_T = 256;                             // threads per workgroup (one wave per SIMD)
constexpr int QB_BW_MAX = 128;                        // most columns a workgroup keeps in LDS
constexpr int QB_SMALL_DOUBLES = 3712;                // 32 x 32 work matrices and vectors beside the block
constexpr size_t QB_LDS_MAX = 163840;

// How a matrix of `rows` rows (the tallest panel) is spread over workgroups.
struct QrBlockPlan {
    int bw;                 // columns resident per launch (multiple of 32)
    int rows_per, ldr, G;   // rows per workgroup, LDS column stride, workgroups for the tallest panel
    int cmax;               // doubles per workgroup in the exchange buffer
    size_t lds_bytes;
    size_t xbuf_doubles;    // exchange buffer per matrix
};
// False when no arrangement fits (batch > 8, or more rows than resident_workgroup_cap() workgroups can hold).
bool qr_block_plan(int rows, int batch, int bw_want, QrBlockPlan *p);

// Factors the first `nfac` columns of the block whose first element is A (rows mp), keeping `bwc` >= nfac columns
// resident and updated.  seq0: sequence numbers seq0+1 ... seq0+64*panels are used (they must grow from launch to
// launch of one factorisation).  T receives one compact-WY factor per panel; V (optional, batch 1) the explicit
// reflectors.  roff > 0: the launch starts roff rows above the block's diagonal origin (A, mp and Vprev count from
// there) and first applies the previous block's reflectors Vprev / Tprev (prev_np panels) to the resident columns;
// vprev_rows >= 0: Vprev has that many rows (a banded matrix: the rest of the launch's rows see zeros).
vslam_status qr_block_launch(vslam_ctx *ctx, hipStream_t stream, const QrBlockPlan &pl, double *A, int lda, size_t strideA, int batch,
                             int mp, int bwc, int nfac, unsigned seq0, double *T, size_t strideT, double *V, int ldv,
                             double *xbuf, int roff = 0, const double *Vprev = nullptr, int ldvp = 0,
                             const double *Tprev = nullptr, int prev_np = 0, int vprev_rows = -1);

}  // namespace vslam
