// Device-side dense fp64 building blocks (column-major, device pointers, all
// queued on ctx->stream): MFMA GEMM, blocked Householder QR, triangular
// inverse / solves.  Declarations only; kernels live in gemm.hip / qr.hip /
// tri.hip.
#pragma once

#include "common.hpp"

#include <cmath>

namespace vslam {

// v of lane l (l wave-uniform) as a wave-uniform value: two v_readlane, no LDS round trip.
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

#ifdef __HIPCC__
// acc - sum_{l in [l0, l1)} a[l * sa] * b[l * sb] with the loads of eight terms in flight at a time
// (the callers run one wave per workgroup, so nothing else hides the LDS latency).
__device__ __forceinline__ double lds_dot_sub(double acc, const double *a, int sa, const double *b, int sb, int l0, int l1)
{
    double s0 = acc, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int l = l0;
    for (; l + 8 <= l1; l += 8) {
        double u[8], x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { u[j] = a[(l + j) * sa]; x[j] = b[(l + j) * sb]; }
        s0 -= u[0] * x[0]; s1 -= u[1] * x[1]; s2 -= u[2] * x[2]; s3 -= u[3] * x[3];
        s0 -= u[4] * x[4]; s1 -= u[5] * x[5]; s2 -= u[6] * x[6]; s3 -= u[7] * x[7];
    }
    for (; l < l1; ++l) s0 -= a[l * sa] * b[l * sb];
    return (s0 + s1) + (s2 + s3);
}
#endif

// How the A operand of a GEMM is read from memory.
enum AMode {
    A_PLAIN = 0,
    A_UNIT_LOWER = 1,   // Householder vectors stored below the diagonal of a panel:
                        // element (r,c) reads 0 for r<c, 1 for r==c, memory for r>c
    A_UPPER = 2,        // upper-triangular matrix: element (r,c) reads 0 for r>c
};

struct GemmDesc {
    int M, N, K;
    bool transA, transB;     // op(A) is M x K, op(B) is K x N
    AMode amode = A_PLAIN;   // applies to the stored A matrix coordinates
    bool b_upper = false;    // stored B is upper triangular (reads 0 below the diagonal)
    bool c_upper = false;    // only the tiles of C that touch its upper triangle are computed and written
    double alpha = 1.0, beta = 0.0;
    const double *A; int lda; size_t strideA = 0;
    const double *B; int ldb; size_t strideB = 0;
    double *C; int ldc; size_t strideC = 0;
    int batch = 1;
    const int *active = nullptr;   // device, optional: batch members whose entry is 0 are skipped
    int ksplit = 1;          // >1: split K; slab s of batch b is written (beta ignored) at
    size_t strideSplit = 0;  //     C + b*strideC + s*strideSplit
};

// C = alpha * op(A) * op(B) + beta * C on v_mfma_f64_16x16x4_f64.
vslam_status gemm_f64(vslam_ctx *ctx, const GemmDesc &d);
vslam_status gemm_f64_on(vslam_ctx *ctx, const GemmDesc &d, hipStream_t stream);   // same, on an explicit stream

// In-place Householder QR of `batch` matrices (rows x cols, leading dimension
// lda, element stride strideA); on return the upper triangle holds R (LAPACK /
// Eigen sign convention) and the strict lower triangle is zero.  `ws` must
// hold qr_ws_bytes(rows, cols, batch) bytes.
size_t qr_ws_bytes(int rows, int cols, int batch);
// band >= 0 promises A(i, c) == 0 for i > c + band (e.g. a diagonal block stacked under a dense one of `band` rows): panels
// then stop at the band.  zero_lower = false leaves the reflectors below the diagonal for a caller who copies R out itself.
vslam_status qr_upper_dev(vslam_ctx *ctx, double *A, int rows, int cols, int lda, size_t strideA, int batch, void *ws, int band = -1,
                          bool zero_lower = true);

// W = inverse of the upper-triangular S (n x n); strict lower part of W zeroed (zero_lower = false: only inside the
// diagonal 64 x 64 blocks -- for a caller that reads W as a triangular operand and saves the pass).
size_t trtri_ws_bytes(int n);
vslam_status trtri_upper_dev(vslam_ctx *ctx, const double *S, int n, int ldS, double *W, int ldW, void *ws, bool zero_lower = true);
size_t trtri_batch_ws_bytes(int n, int batch);
vslam_status trtri_upper_batch_dev(vslam_ctx *ctx, const double *S, int n, int ldS, size_t strideS, double *W, int ldW, size_t strideW,
                                   int batch, void *ws, bool zero_lower = true);

// Upper triangle of B(i, j) = A(n-1-i, n-1-j) (transposed: A(n-1-j, n-1-i)) for `batch` n x n matrices (ld n) `stride` apart;
// what lies below the diagonal tiles of B is not written.  sym_upper: A is symmetric and only its upper triangle is read.
vslam_status reverse_upper_dev(vslam_ctx *ctx, const double *A, int n, double *B, bool transposed, size_t stride, int batch, bool sym_upper = false);
// Strict lower triangle := transpose of the strict upper triangle (batch matrices strideA apart).
vslam_status mirror_upper_dev(vslam_ctx *ctx, double *A, int n, int lda, size_t strideA, int batch);

// Solves S^T x = b (trans = true) or S x = b (trans = false) for one vector, S upper triangular (the strict lower
// triangle is never read); z = scale * x.  b and z may be the same vector.
vslam_status trsv_upper_dev(vslam_ctx *ctx, const double *S, int n, int ldS, bool trans, const double *b, double *z, double scale = 1.0);
// The same for a bank of matrices strideS apart (vectors n apart; active[m] == 0 skips matrix m; active may be null),
// one workgroup per matrix; only for sizes trsv_upper_bank_fits accepts.
bool trsv_upper_bank_fits(int n);
vslam_status trsv_upper_bank_dev(vslam_ctx *ctx, const double *S, int n, int ldS, size_t strideS, bool trans, const double *rhs, double *z,
                                 double scale, const int *active, int count, bool listed = false);

// Batched point-landmark log-likelihood (point_loglik.hip), all pointers device, strides in elements;
// active (optional, device): instances whose entry is 0 are skipped.
vslam_status launch_point_loglik(vslam_ctx *ctx, const CamK &cam, int batch, int n,
                                 const double *x, size_t sx, const double *y, size_t sy,
                                 const uint8_t *vis, size_t sv, double sigma,
                                 double *value, double *g, size_t sg, double *H, int ldH, size_t sH, int32_t *code,
                                 const int *active);

// Device buffers of one CostJointDensity problem (reference src/gaussian.hpp:435-467),
// carved out of the context workspace; shared by the cost entry point and the
// IEKF driver.
struct CostDevice {
    int n, L;
    double *x, *mu, *S, *y, *tmp, *gp, *gl, *g, *Hp, *Hl, *H, *vals;   // vals: [logprior, loglik, V]
    uint8_t *vis;
    double *W; void *tws;
    bool prior_ready;
};
size_t cost_ws_bytes(int n);
void cost_carve(void *ws, int n, CostDevice *c);
vslam_status cost_eval_dev(vslam_ctx *ctx, const CamK &k, double sigma, CostDevice *c);


// Device-resident trust-region Newton iteration + posterior square root of the IEKF update
// (newton.hip).  `ws` holds newton_ws_bytes(n) bytes beyond the CostDevice carve-up; *status_out
// is 0 when the results were produced, 1 when an iterate's Hessian was not positive definite
// (outputs are then meaningless and the caller takes the eigenvalue route).
// Stopping rule of the multiplier iteration on the Cholesky routes, next to the reference's |ff| < sqrt(eps)
// with ff = 1/Delta - 1/|p| (src/fmin.cpp:118-126).  Once the trust region has shrunk (Delta << 1) that
// absolute test asks for more digits than ff has, and the reference then simply runs its 20 Newton steps,
// which cost nothing in the eigenbasis; here each step is a factorisation.  When |p| matches Delta to
// 1e-10 relative, the Newton step taken from that ff lands within rounding of the limit (quadratic
// convergence), which is where the reference's remaining steps stay, flipping the last bits.
inline bool trs_multiplier_settled(double ff, double Delta) { return std::fabs(ff) * Delta < 1e-10; }

size_t newton_ws_bytes(int n);
vslam_status potrf_upper_dev(vslam_ctx *ctx, double *A, int n, int lda, int *info);
// rhs (optional, one n-vector per matrix, stride_rhs apart): on return U^-T rhs -- the forward substitution rides
// along with the factorisation as one more column of the row-panel solves and trailing updates.
struct PotrfFrom { const double *H; size_t stride; const double *lamv; double lam; };      // factor H + shift I (lamv: device, one shift per matrix; else lam)
vslam_status potrf_upper_batch_dev(vslam_ctx *ctx, double *A, int n, int lda, size_t strideA, int batch, int *info, const int *active,
                                   bool zero_lower = true, double *rhs = nullptr, size_t stride_rhs = 0, bool listed = false,
                                   const PotrfFrom *from = nullptr);
// from: the matrices to factor are from->H + shift I (same ld); A receives the factors (upper triangles only) without a copy first.
// listed: `active` holds `batch` matrix indices (the launches cover those matrices only) instead of a mask over `batch` matrices.
// device_out: mu_post and S_post are device memory (the results stay on the GPU) instead of the caller's host memory.
vslam_status iekf_newton_device(vslam_ctx *ctx, const CamK &k, double sigma, CostDevice *c, void *ws,
                                double *mu_post, double *S_post, int ldSp, int *fmin_code, int *iters, int *evals_out,
                                int *status_out, bool device_out = false);

}  // namespace vslam
