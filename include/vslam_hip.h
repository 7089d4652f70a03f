/*
 * libvslam_hip -- C ABI of the MI355X (gfx950) implementation of the
 * VisualSLAM point-landmark hot path and corner front end.
 *
 * Every entry point is extern "C", takes plain pointers and sizes, and
 * returns a vslam_status (0 = ok, negative = error) instead of the
 * reference's assert() aborts.  All matrices are column-major doubles, the
 * layout of Eigen::MatrixXd, so a caller holding Eigen objects passes
 * M.data(), M.rows(), M.outerStride() straight through.
 *
 * Two families:
 *   - host-pointer entry points (no suffix): drop-in for the reference's call
 *     sites; the library stages inputs to the device, runs the kernels on the
 *     context's stream, copies the results back and synchronises.
 *   - device-pointer entry points (_dev suffix): every pointer is device
 *     memory; work is queued on the context's stream and NOT synchronised.
 *     Batched variants take a batch count and element strides.
 *
 * Each declaration cites the reference interface it replaces
 * (paths relative to the reference checkout).
 */
#ifndef VSLAM_HIP_H
#define VSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t vslam_status;
enum {
    VSLAM_OK = 0,
    VSLAM_ERR_ARG = -1,          /* null pointer / bad enum */
    VSLAM_ERR_SHAPE = -2,        /* dimensions violate the reference's asserts */
    VSLAM_ERR_DEVICE = -3,       /* HIP runtime error (see vslam_last_error) */
    VSLAM_ERR_ALLOC = -4,
    VSLAM_ERR_IO = -5,
    VSLAM_ERR_CAPACITY = -6,     /* problem larger than a kernel's static limit */
    VSLAM_ERR_UNSUPPORTED = -7   /* calibration flag outside RATIONAL|THIN_PRISM */
};

typedef struct vslam_ctx vslam_ctx;

/* CameraParameters, src/cameraModel.hpp:12-23.  d[] is OpenCV's order
 * k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4; flag is the OpenCV calibration flag
 * (CALIB_RATIONAL_MODEL 0x4000, CALIB_THIN_PRISM_MODEL 0x8000). */
typedef struct {
    double fx, fy, cx, cy;
    double d[12];
    double fov_deg;              /* fieldOfView, 120 by default (not serialised) */
    int32_t flag;
    int32_t width, height;
    int32_t reserved_;
} vslam_camera;

/* Feature, src/imagefeatures.h:15-19 */
typedef struct { int32_t x, y; float score; } vslam_feature;

typedef struct {
    int32_t block_size;          /* structure-tensor window (odd, 3..15), 7 = cv::ORB's Harris */
    int32_t border;              /* keypoints satisfy border <= x < w-border (>= block_size/2+2) */
    int32_t max_features;        /* cap per detector (reference: 10000, src/SLAM.cpp:114) */
    float harris_k;              /* 0.04 */
    float harris_thresh;         /* keep if response > thresh (reference: 1e-4, src/SLAM.cpp:116) */
    float shi_thresh;
} vslam_corner_params;

enum { VSLAM_LAYOUT_COMPAT = 0, VSLAM_LAYOUT_STRICT = 1 };
/* BOTH runs the two detectors off one structure tensor: out holds 2*cap records
 * (Harris first, then Shi-Tomasi) and count two entries per frame. */
enum { VSLAM_DETECTOR_HARRIS = 0, VSLAM_DETECTOR_SHI_TOMASI = 1, VSLAM_DETECTOR_BOTH = 2 };

/* ---- context ---- */
/* Creates a context on `device` with its own non-blocking stream. */
vslam_status vslam_ctx_create(int device, vslam_ctx **out);
/* Same, but queues all work on the caller's hipStream_t (e.g. torch's current
 * stream); the stream is borrowed, not destroyed. */
vslam_status vslam_ctx_create_on_stream(int device, void *hip_stream, vslam_ctx **out);
vslam_status vslam_ctx_destroy(vslam_ctx *ctx);
vslam_status vslam_ctx_synchronize(vslam_ctx *ctx);
const char *vslam_status_string(vslam_status s);
const char *vslam_last_error(const vslam_ctx *ctx);
/* 1 when the library was built with its gfx950 kernels (always, for this build). */
int vslam_has_device_code(void);

/* ---- camera.xml ----
 * CameraParameters::read + importCalibrationData, src/cameraModel.cpp:71-77,93-100.
 * Host only; parses the cv::FileStorage XML (<camera_matrix>,
 * <distortion_coefficients>, <flag>, <imageSize>). */
vslam_status vslam_camera_from_xml(const char *path, vslam_camera *cam);

/* ---- gray frames on disk (PGM) ----
 * The reference takes its frames from cv::VideoCapture / cv::imread (src/SLAM.cpp:236, src/main.cpp:60-70) and hands
 * detectAndDrawORB a cv::Mat (src/SLAM.cpp:349); OpenCV is not part of this library, so a frame crosses the
 * boundary as plain 8-bit gray rows and its file format is binary PGM ("P5": magic, width, height, maxval <= 255
 * as decimal text separated by white space or '#' comments, one white-space byte, then width * height bytes).
 * Host only.  vslam_pgm_info reads the header; vslam_pgm_read fills gray[0 .. w*h) row by row (cap = its size in
 * bytes; VSLAM_ERR_CAPACITY when too small).  A file that is not P5, has maxval > 255 (16-bit samples:
 * VSLAM_ERR_UNSUPPORTED), a zero / oversized (> 65536) dimension or fewer bytes than the header promises is
 * VSLAM_ERR_IO and leaves gray untouched beyond what was read. */
vslam_status vslam_pgm_info(const char *path, int32_t *w, int32_t *h, int32_t *maxval);
vslam_status vslam_pgm_read(const char *path, uint8_t *gray, size_t cap, int32_t *w, int32_t *h);
/* Writes w x h gray rows (row stride `stride` bytes) as P5 with maxval 255. */
vslam_status vslam_pgm_write(const char *path, const uint8_t *gray, int w, int h, int stride);

/* ---- projection ----
 * worldToPixel<double>, src/cameraModel.hpp:32-216, for landmarks j0..j0+L-1
 * of state x (n = 12 + 3*nLandmarks).  px[2L]; code[L] in {0,1,2}. Culled
 * entries: code 2 keeps the computed pixel, code 1 writes NaN. */
vslam_status vslam_project(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n,
                           int j0, int L, double *px, int32_t *code);

/* ---- measurement model ----
 * MeasurementPointLandmarkBundle::operator(), src/measurementPointLandmark.cpp:13-70
 * (which loops MeasurementPointLandmarkSingle, src/measurementPointLandmark.hpp:10-304).
 * h[2L]; SR (2L x 2L, ld 2L) = 0.01*I or NULL; C (2L x n, ld 2L) or NULL;
 * code[L] or NULL.  layout COMPAT reproduces the reference's column placement
 * (0-2, 3-5, 6+3j) and its use of x(3..5) in the rotation block; STRICT places
 * the blocks at 6-8, 9-11, 12+3j with the true pose angles. */
vslam_status vslam_bundle_measure(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n,
                                  int layout, double *h, double *SR, double *C, int32_t *code);
/* The same evaluation with every array resident on the context's device (x, h, SR, C, code are device
 * pointers; SR, C and code may be NULL).  Enqueued on the context's stream; returns without waiting. */
vslam_status vslam_bundle_measure_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n,
                                      int layout, double *h, double *SR, double *C, int32_t *code);
/* MeasurementPointLandmarkSingle::operator(), src/measurementPointLandmark.hpp:10,33,46.
 * h[2], SR[4] or NULL, J (2 x n) or NULL; returns the projection code through *code. */
vslam_status vslam_single_measure(vslam_ctx *ctx, const vslam_camera *cam, int j, const double *x, int n,
                                  int layout, double *h, double *SR, double *J, int32_t *code);

/* ---- point log-likelihood ----
 * PointLogLikelihood::operator() x3, src/model.cpp:408-437 (value / gradient /
 * exact Hessian of pointLogLikelihood, src/model.cpp:369-406).
 * y[2L] = landmarks[j].pixel_measurement, visible[L] = landmarks[j].isVisible,
 * sigma = measurement_noise.  g (n) and H (n x n, leading dimension ldH) may
 * be NULL, selecting the 1-, 2- or 3-output arity.  code[L] may be NULL. */
vslam_status vslam_point_loglik(vslam_ctx *ctx, const vslam_camera *cam, const double *x, int n,
                                const double *y, const uint8_t *visible, double sigma,
                                double *value, double *g, double *H, int ldH, int32_t *code);
/* Device-pointer, strided-batch form: instance b reads x + b*stride_x etc.
 * Strides are in elements; value has stride 1, code stride L. */
vslam_status vslam_point_loglik_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, int batch, int n,
                                          const double *x, size_t stride_x,
                                          const double *y, size_t stride_y,
                                          const uint8_t *visible, size_t stride_vis, double sigma,
                                          double *value, double *g, size_t stride_g,
                                          double *H, int ldH, size_t stride_H, int32_t *code);
/* The same call for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc); it has no counterpart in the reference.  It calls no hipMalloc / hipFree: the
 * per-landmark records and per-block pose sums between its two kernels live in a static device array of the
 * library (about 13.5 MB per GPU, held until the process ends) whose address is looked up at context creation.
 * Limits: batch * L <= 81920 landmarks and batch * ceil(L / 256) <= 4096, else VSLAM_ERR_CAPACITY; when H is
 * given, ldH must equal n.  Results are those of vslam_point_loglik_batch_dev bit for bit (same kernels).
 * Every call on one GPU shares that GPU's array: calls that overlap on one GPU (different streams or contexts)
 * race; calls on one stream are ordered and safe. */
vslam_status vslam_point_loglik_batch_dev_noalloc(vslam_ctx *ctx, const vslam_camera *cam, int batch, int n,
                                                  const double *x, size_t stride_x,
                                                  const double *y, size_t stride_y,
                                                  const uint8_t *visible, size_t stride_vis, double sigma,
                                                  double *value, double *g, size_t stride_g,
                                                  double *H, int ldH, size_t stride_H, int32_t *code);
/* Measurement aid without a counterpart in the reference: with timing on, the batched call above records events
 * around its two kernels (per-landmark terms; assembly of the dense Hessian) whenever it takes its two-kernel form
 * (batch * n * n * 8 >= 32 MiB), and the second function waits for them and returns the two durations of the last
 * such call in milliseconds.  Off by default; the events cost a few microseconds per call. */
vslam_status vslam_ctx_set_kernel_timing(vslam_ctx *ctx, int on);
vslam_status vslam_ctx_last_point_loglik_times(vslam_ctx *ctx, double *terms_ms, double *assemble_ms);

/* ---- Gaussian algebra ----
 * logGaussian x3, src/gaussian.hpp:30-74.  S upper triangular sqrt-covariance. */
vslam_status vslam_log_gaussian(vslam_ctx *ctx, const double *x, const double *mu, const double *S,
                                int n, int ldS, double *value, double *g, double *H);
/* The QR step of affineTransform, src/gaussian.hpp:117-127:
 * Syy = triu(qr([S*C^T ; SR]))[0:m,0:m]; S n x n, C m x n, SR m x m. */
vslam_status vslam_affine_sqrt(vslam_ctx *ctx, const double *S, int n, int ldS,
                               const double *C, int m, int ldC, const double *SR, int ldSR,
                               double *Syy, int ldSyy);
/* affineTransform(mux, Sxx, MeasurementPointLandmarkBundle, muy, Syy) as called at
 * src/SLAM.cpp:371-387 and src/dataAssociation.cpp:53-61, fused on the device. */
vslam_status vslam_affine_bundle(vslam_ctx *ctx, const vslam_camera *cam, const double *mux,
                                 const double *Sxx, int n, int ldS, int layout,
                                 double *muy, double *Syy, int ldSyy, int32_t *code);
/* Device-pointer batch of independent filters (the same transform per filter, src/gaussian.hpp:85-131 with the
 * bundle of src/measurementPointLandmark.cpp:13-70): filter b reads mux + b*n, Sxx + b*n*n (ld n) and writes
 * muy + b*2L, Syy + b*(2L)^2 (ld 2L, upper triangular), code + b*L (may be NULL).  Queued, not synchronised. */
vslam_status vslam_affine_bundle_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mux, const double *Sxx, int n,
                                           int batch, int layout, double *muy, double *Syy, int32_t *code);
/* Eigen::HouseholderQR + triangularView<Upper>, src/gaussian.hpp:121-124,515-516,
 * src/SLAM.cpp:810-819: in place; R in the upper triangle, zeros below. */
vslam_status vslam_qr_upper(vslam_ctx *ctx, double *A, int rows, int cols, int lda);
vslam_status vslam_qr_upper_batch_dev(vslam_ctx *ctx, double *A, int rows, int cols, int lda,
                                      size_t stride_A, int batch);
/* src/gaussian.hpp:513-516: S = triu(qr(diag(1/sqrt(v)) * Q^T)). */
vslam_status vslam_iekf_post_sqrt(vslam_ctx *ctx, const double *Q, const double *v, int n, double *S_out);
/* conditionGaussianOnMarginal, src/gaussian.cpp:19-36 (joint dim ny+nx, ld). */
vslam_status vslam_condition_on_marginal(vslam_ctx *ctx, const double *muyx, const double *Syx,
                                         int ny, int nx, int ld, const double *y,
                                         double *mux_out, double *Sx_out, int ldo);

/* ---- cost + measurement update ----
 * CostJointDensity::operator() (g,H arity), src/gaussian.hpp:455-466, with
 * PointLogLikelihood as the likelihood: V = -(logprior + loglik), g, H. */
vslam_status vslam_cost_joint_density(vslam_ctx *ctx, const vslam_camera *cam, const double *x,
                                      const double *mu, const double *S, int n, int ldS,
                                      const double *y, const uint8_t *visible, double sigma,
                                      double *value, double *g, double *H);
/* measurementUpdateIEKF(mux, Sxx, u, y, PointLogLikelihood, param, muxGy, SxxGy),
 * src/gaussian.hpp:469-532 as called at src/SLAM.cpp:492: trust-region Newton
 * (src/fmin.hpp:19-107, src/fmin.cpp:40-131) on the host driving the device
 * cost kernels, then the posterior square-root QR.  fmin_code: 0 converged,
 * 1 iteration limit; iters / cost_evals optional. */
vslam_status vslam_measurement_update_iekf(vslam_ctx *ctx, const vslam_camera *cam,
                                           const double *mu, const double *S, int n, int ldS,
                                           const double *y, const uint8_t *visible, double sigma,
                                           double *mu_post, double *S_post, int ldSp,
                                           int32_t *fmin_code, int32_t *iters, int32_t *cost_evals);

/* The same update (src/gaussian.hpp:469-532 as called at src/SLAM.cpp:492) for a bank of `batch`
 * independent filters held on the device: mu (batch x n), S (batch matrices n x n, column-major,
 * ld n, upper triangular), y (batch x 2L), visible (batch x L), mu_post and S_post like mu and S are
 * device pointers; fmin_code / iters / status are host arrays of `batch` entries (each may be NULL).
 * Every step of the trust-region iteration runs over the whole bank in one set of kernel launches,
 * on the Cholesky route described below.  Per filter the iterates follow the reference's to about
 * 1e-10 relative, not bit for bit: the multiplier iteration of the trust-region subproblem also stops once
 * |p| matches the radius to 1e-10 relative (with a collapsed trust region the reference's absolute test
 * |1/Delta - 1/|p|| < sqrt(eps) asks for more digits than the quantity has), and the accept / radius tests
 * on rho are discontinuous, so a last-bit difference can in principle change a decision.  The same holds for
 * the subproblems the bank solves ahead: a filter whose step was rejected has its next subproblems (radius
 * quartered again and again, Hessian and gradient unchanged) solved in spare slots beside the running ones,
 * and a solution is used when the radius the reference would set, 0.25 |p|, equals the one solved for to 1e-9
 * relative (VSLAM_BANK_SOLVE_AHEAD=0 in the environment switches this off; the work space grows by up to two
 * n x n matrices per filter for the spare slots).
 * States of more than 1 024 unknowns do not run in lock step (one such filter fills the GPU): they are updated one after
 * the other by the single-filter device route, inputs and results staying in device memory.
 * A Hessian that is not positive definite at some iterate is handled within the bank (see CHOLESKY below).
 * status[b]: 0 = mu_post / S_post of filter b are valid; 1 = a subproblem of filter b could not be finished from
 * Cholesky factors (the hard case of the trust-region subproblem, a non-finite Hessian, or a posterior Hessian
 * that is not positive definite): its outputs are meaningless and the caller redoes that filter with
 * vslam_measurement_update_iekf, which has the eigenvalue route to fall back on.  Returns after the results are
 * complete. */
vslam_status vslam_measurement_update_iekf_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mu,
                                                     const double *S, int n, int batch, const double *y,
                                                     const uint8_t *visible, double sigma, double *mu_post,
                                                     double *S_post, int32_t *fmin_code, int32_t *iters, int32_t *status);

/* How vslam_measurement_update_iekf solves the trust-region subproblem of src/fmin.cpp:40-131.
 * CHOLESKY (default; no counterpart in the reference): from Cholesky factors of H + lambda I on the device, no
 *      eigensolver.  The subproblem's solution is the reference's (the search for the multiplier ends with the
 *      reference's test and its final step; steps agree to about 1e-10 relative: see the stopping rule noted at
 *      vslam_measurement_update_iekf_batch_dev), but the multipliers tried on the way are chosen from a rational
 *      model of |p(lambda)| built from a few triangular solves, not by the reference's Newton iteration, so fewer
 *      factorisations are needed.  An indefinite Hessian shows as a failed factorisation whose first non-positive
 *      pivot gives the next shift; the search then brackets the multiplier above -lambda_min as More & Sorensen's
 *      safeguards do.  Only the hard case (no boundary solution above -lambda_min: the bracket collapses) and
 *      non-finite input send the update to the EIG route, so a caller who changes nothing gets this route with
 *      that fallback.  The posterior square root is the inverse of a triangular factor of the Hessian (equal to
 *      the reference's up to row signs).  Iteration counts equal the reference's except where a decision of the
 *      outer loop (accept / shrink on rho) is taken on rounding noise, i.e. in the last iterations of an update
 *      whose steps have fallen below the resolution of the cost; the EIG route reproduces those counts.
 * EIG (opt-in, the reference's arithmetic step for step): from a symmetric eigendecomposition of the Hessian
 *      taken on the host at every accepted iterate (O(n^3) scalar work per iterate; 110 ms per update at 150
 *      landmarks against 4.6 ms, and not usable at 2,000). */
enum { VSLAM_TRS_EIG = 0, VSLAM_TRS_CHOLESKY = 1 };
vslam_status vslam_ctx_set_trs_solver(vslam_ctx *ctx, int solver);
/* Route the context's last measurement update actually took (VSLAM_TRS_CHOLESKY = device route,
 * VSLAM_TRS_EIG = host eigenvalue route), -1 before the first update.  Diagnostic. */
int vslam_ctx_last_trs_route(const vslam_ctx *ctx);
/* Measurement aid without a counterpart in the reference: wall-clock split of the context's last measurement
 * update when it ran on the device route.  out8 = { ms in cost evaluations (a-8: prior + likelihood, g, H),
 * ms in the trust-region subproblems (Cholesky factorisations, triangular solves, dot products), ms in the
 * posterior factor, ms of the whole device part, cost evaluations, factorisations, iterations, 0 }. */
vslam_status vslam_ctx_last_update_profile(vslam_ctx *ctx, double *out8);

/* ---- data association ----
 * normcdf and chi2inv, src/utility.cpp:13-24 (host). */
double vslam_normcdf(double z);
double vslam_chi2inv(double p, double nu);
/* individualCompatibility, src/dataAssociation.cpp:154-177, for every pair at once:
 * compat[i*n + j] = feature i (column i of the 2 x m matrix y) is compatible with landmark j.
 * muY[2n], SYY (2n x 2n upper, ld ldS), chi2LUT[n]. */
vslam_status vslam_individual_compat(vslam_ctx *ctx, const double *y, int m, const double *muY, const double *SYY,
                                     int n, int ldS, const double *chi2LUT, uint8_t *compat);
/* jointCompatibility, src/dataAssociation.cpp:179-285: idx[n] (feature of landmark j or -1). */
vslam_status vslam_joint_compat(vslam_ctx *ctx, const int32_t *idx, int n, double sU, const double *y, int m,
                                const double *muY, const double *SYY, int ldS, const double *chi2LUT,
                                double *surprisal, int32_t *compatible);
/* The search of snn, src/dataAssociation.cpp:63-149, on a given predicted-measurement density;
 * the chi2 table and the per-landmark surprisal log(width)+log(height) are formed as there. */
vslam_status vslam_snn_density(vslam_ctx *ctx, const double *muY, const double *SYY, int n, int ldS, const double *y, int m,
                               int width, int height, int enforce_jc, int32_t *idx, double *surprisal);
/* Device-pointer batch of independent instances (one workgroup each): instance b reads muY + b*stride_mu,
 * SYY + b*stride_S, y + b*stride_y (2 x m), writes idx + b*n and surprisal[b].  Queued, not synchronised. */
vslam_status vslam_snn_density_batch_dev(vslam_ctx *ctx, int batch, const double *muY, size_t stride_mu,
                                         const double *SYY, int ldS, size_t stride_S, int n,
                                         const double *y, size_t stride_y, int m, int width, int height,
                                         int enforce_jc, int32_t *idx, double *surprisal);
/* snn(mux, Sxx, y, param, idx, enforceJointCompatibility), src/dataAssociation.cpp:25-150 as called at
 * src/SLAM.cpp:396,533: affineTransform through the landmark bundle, then the search.  Returns the
 * surprisal through *surprisal; idx[(nx_all-12)/3]. */
vslam_status vslam_snn(vslam_ctx *ctx, const vslam_camera *cam, const double *mux, const double *Sxx, int nx_all, int ldS,
                       int layout, const double *y, int m, int enforce_jc, int32_t *idx, double *surprisal);

/* ---- time update ----
 * timeUpdateContinuous(mukm1, Skm1, u, SlamProcessModel, param, timestep, muk, Sk), src/gaussian.hpp:337-383
 * as called at src/SLAM.cpp:250,337,507: the RK4 step with Jacobian propagation (RK4SDEHelper,
 * src/gaussian.hpp:171-239) of SlamProcessModel (src/model.cpp:9-136) pushed through affineTransform.
 * pos_tune / ori_tune are SlamParameters::position_tune / orientation_tune (3 x 3 column-major,
 * src/model.hpp:17-18); the input u is unused by the reference's process model and has no argument
 * here.  S and S_out are n x n column-major (upper-triangular square-root covariance; S_out may be S,
 * mu_out may be mu).  The reference's 3n x n Householder QR keeps its R: the rows that are
 * identically zero are left out. */
vslam_status vslam_time_update(vslam_ctx *ctx, const double *mu, const double *S, int n, int ldS,
                               const double *pos_tune, const double *ori_tune, double dt,
                               double *mu_out, double *S_out, int ldSout);
/* Device-pointer batch of independent filters: mu + b*n, S + b*n*n (ld n), same for the outputs.
 * pos_tune / ori_tune are host pointers.  s_is_upper != 0: the caller guarantees that every S is upper
 * triangular, as every factor this library or the reference's QR returns is; its strict lower part is
 * then not read and the structured O(n^2) kernels run (vslam_time_update checks this itself).
 * Queued on the context's stream, not synchronised. */
vslam_status vslam_time_update_batch_dev(vslam_ctx *ctx, const double *mu, const double *S, int n, int batch,
                                         const double *pos_tune, const double *ori_tune, double dt,
                                         double *mu_out, double *S_out, int s_is_upper);

/* ---- landmark removal ----
 * removeBadLandmarks(mup, Sp, landmarks, j), src/SLAM.cpp:796-820: the three states of landmark j leave
 * mu (rows 12+3j..14+3j) and S (the same columns, src/SLAM.cpp:848-868); the n x (n-3) factor is
 * re-triangularised by Householder QR and its leading (n-3) x (n-3) block returned. */
vslam_status vslam_remove_landmark(vslam_ctx *ctx, const double *mu, const double *S, int n, int ldS, int j,
                                   double *mu_out, double *S_out, int ldSout);
/* The same on device-resident state (src/SLAM.cpp:796-820): mu (n), S (n x n, ldS), mu_out (n-3) and
 * S_out ((n-3) x (n-3), ldSout) are device pointers; the outputs must not overlap the inputs.
 * s_is_upper != 0: the caller guarantees an upper-triangular S (every factor this library or the
 * reference's QR returns); its strict lower part is then not read and the reflectors, which only span
 * four rows each, are applied in O(n^2) work.  0: dense QR.  Queued on the context's stream, not
 * synchronised (vslam_remove_landmark checks the triangularity itself). */
vslam_status vslam_remove_landmark_dev(vslam_ctx *ctx, const double *mu, const double *S, int n, int ldS, int j,
                                       double *mu_out, double *S_out, int ldSout, int s_is_upper);
/* Device-pointer batch of independent filters with upper-triangular factors (src/SLAM.cpp:796-820 per filter):
 * filter b drops landmark j_each[b] (device array) from mu + b*n, S + b*n*n (ld n) into mu_out + b*(n-3),
 * S_out + b*(n-3)^2 (ld n-3); one workgroup re-triangularises one filter.  status (device, batch entries, may
 * be NULL): 0, or 1 for an index outside 0..L-1, whose outputs are left untouched.  The strict lower part of S
 * is not read.  Queued on the context's stream, not synchronised. */
vslam_status vslam_remove_landmark_batch_dev(vslam_ctx *ctx, const double *mu, const double *S, int n, int batch, const int32_t *j_each,
                                             double *mu_out, double *S_out, int32_t *status);

/* ---- descriptor matching ----
 * cv::BFMatcher(cv::NORM_HAMMING).match(query, train, matches), src/SLAM.cpp:222,355: for each of the
 * nq query descriptors (rows of desc_bytes bytes; 32 for ORB) the index of the train descriptor at the
 * smallest Hamming distance -- the lowest index among equals -- and that distance (may be NULL).
 * nt == 0 gives index and distance -1 (OpenCV returns no match). */
vslam_status vslam_match_hamming(vslam_ctx *ctx, const uint8_t *query, int nq, const uint8_t *train, int nt,
                                 int desc_bytes, int32_t *train_idx, int32_t *distance);
/* Device-pointer batch: instance b reads query + b*stride_query, train + b*stride_train (byte strides,
 * multiples of 4; 16-byte aligned train sets take the fast path), writes train_idx + b*nq.  nt_each
 * (device, may be NULL) gives the number of valid train descriptors of each instance (<= nt). */
vslam_status vslam_match_hamming_batch_dev(vslam_ctx *ctx, const uint8_t *query, size_t stride_query, int nq,
                                           const uint8_t *train, size_t stride_train, int nt, const int32_t *nt_each,
                                           int desc_bytes, int batch, int32_t *train_idx, int32_t *distance);

/* ---- corner front end ----
 * The feature stage of detectAndDrawORB, src/imagefeatures.cpp:71-96, reduced
 * to what the caller consumes (src/SLAM.cpp:449-472): keypoints with a
 * Harris (or Shi-Tomasi) response, sorted by response.  gray is w x h u8 with
 * row stride `stride` bytes.  out[cap] receives min(count, cap, max_features)
 * records sorted by score descending (ties: y, then x ascending); *count is
 * the number written.  At most 65536 candidates may survive NMS per detector
 * (VSLAM_ERR_CAPACITY otherwise; the batch form reports count = -1 for such a
 * frame) and max_features <= 16384. */
vslam_status vslam_corner_detect(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                                 const vslam_corner_params *params, int detector,
                                 vslam_feature *out, int cap, int32_t *count);
/* Dense response maps (float32, w*h, row-major); either may be NULL. */
vslam_status vslam_corner_response(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                                   const vslam_corner_params *params, float *harris, float *shi);
/* Device-pointer batch: frames at gray + f*stride_frame; out + f*cap; count[f]. */
vslam_status vslam_corner_detect_batch_dev(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                                           size_t stride_frame, int frames,
                                           const vslam_corner_params *params, int detector,
                                           vslam_feature *out, int cap, int32_t *count);

/* ---- selection of new landmarks ----
 * The loop of src/SLAM.cpp:449-472 over the features sorted by score (src/SLAM.cpp:457): feature i becomes a
 * landmark iff fewer than max_new were taken so far, feats[i].score > feature_thresh, width_thresh <= x <=
 * image_width - width_thresh, height_thresh <= y <= image_height - height_thresh and pixelDistance
 * (src/SLAM.cpp:746-760) finds no landmark -- of the map (landmark_xy: L pairs x, y) or taken earlier in this
 * loop -- closer than pixel_distance_thresh.  selected[0..min(*count, cap)) are the indices taken, in order;
 * max_new = max_landmarks - landmarks in the map, at most 1024. */
typedef struct {
    double feature_thresh, width_thresh, height_thresh, pixel_distance_thresh;
    int32_t image_width, image_height, max_new, reserved;
} vslam_select_params;
/* pixelDistance(landmarks, keypoint, thresh), src/SLAM.cpp:746-760 (host function; its loop index, which the
 * reference leaves uninitialised, starts at 0): 1 iff some landmark pixel is closer than thresh. */
int vslam_pixel_distance(const float *landmark_xy, int L, float x, float y, double thresh);
vslam_status vslam_select_new_landmarks(vslam_ctx *ctx, const vslam_feature *feats, int K, const float *landmark_xy, int L,
                                        const vslam_select_params *params, int32_t *selected, int cap, int32_t *count);
/* Device-pointer batch, one frame per workgroup: frame b reads feats + b*stride_feats (k_each[b] of them if
 * k_each is given, else K; negative counts read as 0), landmark_xy + b*stride_lm (floats; l_each[b] pairs if
 * given, else L) and writes selected + b*cap, count[b].  Queued, not synchronised. */
vslam_status vslam_select_new_landmarks_batch_dev(vslam_ctx *ctx, const vslam_feature *feats, size_t stride_feats,
                                                  const int32_t *k_each, int K, const float *landmark_xy, size_t stride_lm,
                                                  const int32_t *l_each, int L, const vslam_select_params *params, int batch,
                                                  int32_t *selected, int cap, int32_t *count);

/* ---- initialisation of new landmarks ----
 * generateOpticalRay(param, pixel, p), src/SLAM.cpp:876-903 (host function): ray[3] = ray_length * K^-1 [u' v' 1]^T
 * with (u', v') the pixel freed of lens distortion.  The reference obtains (u', v') from
 * cv::undistortPoints(pixel, K, dist, I, [K | 0]); OpenCV is not part of the reference, so the contract is that
 * routine's documented iteration (five fixed-point steps, coefficients cam->d[0..11] in OpenCV's order whatever
 * the flag, pixel and result stored as float).  Returns 0, or 1 on a null pointer or non-finite constants. */
int vslam_optical_ray(const vslam_camera *cam, double u, double v, double ray_length, double *ray);
/* initializeNewLandmark(...), src/SLAM.cpp:905-941, for M keypoints at once (the reference appends them one
 * per call from the same predicted pose, which gives the same state): mu_out (n + 3M) = [mu; r_1 .. r_M] with
 * r_t = mu[6..8] + Rnc(mu[9..11]) * ray(pixels[2t], pixels[2t+1]) (src/SLAM.cpp:918-929), S_out
 * ((n+3M) x (n+3M), ldSout) = [S 0; 0 kappa I] (src/SLAM.cpp:908-915).  pixels: M pairs u, v. */
vslam_status vslam_initialize_landmarks(vslam_ctx *ctx, const vslam_camera *cam, const double *mu, const double *S, int n, int ldS,
                                        const double *pixels, int M, double ray_length, double kappa,
                                        double *mu_out, double *S_out, int ldSout);
/* The same on device-resident state: every pointer but cam is a device pointer, the outputs must not overlap
 * the inputs.  Queued on the context's stream, not synchronised. */
vslam_status vslam_initialize_landmarks_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mu, const double *S, int n, int ldS,
                                            const double *pixels, int M, double ray_length, double kappa,
                                            double *mu_out, double *S_out, int ldSout);
/* Device-pointer batch of independent filters that take in the same number of keypoints: filter b reads mu + b*n,
 * S + b*n*n (ld n), pixels + b*2*M and writes mu_out + b*(n+3M), S_out + b*(n+3M)^2 (ld n+3M).  Queued, not
 * synchronised. */
vslam_status vslam_initialize_landmarks_batch_dev(vslam_ctx *ctx, const vslam_camera *cam, const double *mu, const double *S, int n,
                                                  int batch, const double *pixels, int M, double ray_length, double kappa,
                                                  double *mu_out, double *S_out);

/* FAST-9/16 with non-maximum suppression, the first stage of cv::ORB::detect as the reference configures
 * it (src/imagefeatures.cpp:73-87, fastThreshold 20).  OpenCV's code is not part of the reference, so the
 * contract is the published segment test with OpenCV's conventions: a pixel is a corner iff 9 contiguous
 * pixels of the 16-pixel radius-3 circle are all brighter than I_p + threshold or all darker than
 * I_p - threshold; score = the largest threshold at which it still is; pixels within 3 of the border
 * never are; nonmax != 0 keeps a corner only if it is stronger than every neighbouring corner.  Output
 * as for vslam_corner_detect (sorted by score, ties by y then x; same capacity limits). */
vslam_status vslam_fast9_detect(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, int threshold, int nonmax,
                                int max_features, vslam_feature *out, int cap, int32_t *count);
/* Device-pointer batch: frame f at gray + f*stride_frame, out + f*cap, count[f] (-1: more than 65536 corners). */
vslam_status vslam_fast9_detect_batch_dev(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, size_t stride_frame,
                                          int frames, int threshold, int nonmax, int max_features,
                                          vslam_feature *out, int cap, int32_t *count);

/* ---- ORB keypoints: cv::ORB::create(nfeatures, 1.2f, 8, 75, 0, 2, HARRIS_SCORE, 75, 20)->detect as called by
 * detectAndDrawORB, src/imagefeatures.cpp:73-87 (orientation and descriptors follow further down).  OpenCV's sources are not
 * part of the reference: the chain below restates the published algorithm and the CPU oracle defines the contract
 * ("parity unpinned").  Per pyramid level: FAST-9/16 at fast_threshold with non-maximum suppression; corners closer
 * than edge_threshold to a side dropped; the 2 n strongest by FAST score kept, every tie of the last one included;
 * at those corners only, the Harris measure of a harris_block^2 window of Sobel-like integer gradients,
 * (a b - c^2 - k (a + b)^2) (1 / (4 block 255))^4 in float; the n best kept, ties included.  Levels: level l has
 * scale s_l = scale_factor^l, size round(w / s_l) x round(h / s_l), resized from level l-1 (bilinear, integer
 * arithmetic in 1/2048 pixel: this library's own definition, not OpenCV's INTER_LINEAR_EXACT); the feature budget
 * n_l follows the geometric series of cv::ORB. */
#define VSLAM_ORB_MAX_LEVELS 16
typedef struct {
    int32_t nfeatures;        /* 10000 at src/SLAM.cpp:349 */
    float scale_factor;       /* 1.2 */
    int32_t nlevels;          /* 8 */
    int32_t edge_threshold;   /* 75 */
    int32_t fast_threshold;   /* 20 */
    int32_t harris_block;     /* 7 */
    float harris_k;           /* 0.04 */
    int32_t reserved_;
} vslam_orb_params;
/* cv::KeyPoint as far as detection fills it: position in base-image pixels (level position times the level's
 * scale, in float), response, octave. */
typedef struct { float x, y, response; int32_t octave; } vslam_orb_keypoint;
/* Sizes, feature budgets and scales of the pyramid levels (host; arrays of params->nlevels entries). */
vslam_status vslam_orb_level_plan(const vslam_orb_params *params, int w, int h, int32_t *level_w, int32_t *level_h,
                                  int32_t *level_features, float *level_scale);
/* One level for a batch of device-resident frames: keypoints in level coordinates, sorted by Harris response
 * (ties by y then x); count[f] = -1 when a frame has more than 65536 corners or ties at a cut beyond 16384. */
vslam_status vslam_orb_detect_level_batch_dev(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, size_t stride_frame,
                                              int frames, int nfeatures, int fast_threshold, int edge_threshold, int harris_block,
                                              float harris_k, vslam_feature *out, int cap, int32_t *count);
/* The level-to-level resize on its own (device pointers). */
vslam_status vslam_resize_bilinear_dev(vslam_ctx *ctx, const uint8_t *src, int sw, int sh, int sstride,
                                       uint8_t *dst, int dw, int dh, int dstride);
/* All levels of one host image: keypoints level by level (level 0 first), within a level sorted by response;
 * level_counts (optional) receives the number per level. */
vslam_status vslam_orb_detect(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, const vslam_orb_params *params,
                              vslam_orb_keypoint *out, int cap, int32_t *count, int32_t *level_counts);

/* ---- ORB orientation and descriptors: the compute() half of detectAndDrawORB's cv::ORB
 * (src/imagefeatures.cpp:88-96), whose 32-byte descriptors BFMatcher(NORM_HAMMING) consumes (src/SLAM.cpp:222,355;
 * vslam_match_hamming here).  Restated from the published algorithm, the CPU oracle defines the contract
 * ("parity unpinned"):
 *   orientation  intensity centroid of the radius-15 circular patch on the keypoint's level image, degrees in
 *                [0, 360) by the polynomial arctangent of cv::fastAtan2;
 *   blur         7 x 7 Gaussian, sigma 2, reflect-101 borders, 8.8 fixed point with weights
 *                {18, 34, 49, 54, 49, 34, 18} / 256 (this library's definition, not OpenCV's fixed-point kernel);
 *   descriptor   256 tests I(p0) < I(p1) on the blurred level image, pattern points rotated by the angle and rounded
 *                half to even; bit k of byte i is test 8 i + k.
 * The pattern is 256 x (x0, y0, x1, y1) signed bytes, every coordinate in [-15, 15].  cv::ORB's learned table lives
 * only in OpenCV's sources: a caller who needs cv::ORB's bits passes that table; NULL selects the fixed
 * pseudo-random table returned by vslam_orb_default_pattern (descriptors then match each other, not OpenCV's). */
#define VSLAM_ORB_PATTERN_BYTES 1024
vslam_status vslam_orb_default_pattern(int8_t *pattern);
/* The blur on its own (device pointers). */
vslam_status vslam_gaussian_blur7_dev(vslam_ctx *ctx, const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
/* Orientation and descriptors of the keypoints of one level (all pointers on the device): keypoints[0 ..
 * min(*count, max_count)) in level coordinates, gray = the level image, blurred = its blur, pattern 16-byte
 * aligned; angle[max_count], desc[max_count][32].  A keypoint closer than 23 pixels to a side gets angle -1 and a
 * zero descriptor. */
vslam_status vslam_orb_describe_level_dev(vslam_ctx *ctx, const uint8_t *gray, int stride, const uint8_t *blurred, int bstride,
                                          int w, int h, const vslam_feature *keypoints, const int32_t *count, int max_count,
                                          const int8_t *pattern, float *angle, uint8_t *desc);
/* vslam_orb_detect plus angle[cap] and desc[cap][32] for every keypoint (pattern: host table or NULL). */
vslam_status vslam_orb_detect_and_compute(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, const vslam_orb_params *params,
                                          const int8_t *pattern, vslam_orb_keypoint *out, float *angle, uint8_t *desc, int cap,
                                          int32_t *count, int32_t *level_counts);
/* The same for a batch of device-resident frames (frame f at gray + f*stride_frame), everything on the device and
 * queued on the context's stream: out + f*cap, angle + f*cap, desc + f*cap*32 (angle and desc both NULL: keypoints
 * only), count[f] (-1: a level overflowed or the frame has more than cap keypoints), level_counts[f][nlevels]
 * (optional).  pattern: device table, 16-byte aligned, or NULL for the library's own. */
vslam_status vslam_orb_detect_and_compute_batch_dev(vslam_ctx *ctx, const uint8_t *gray, int w, int h, int stride, size_t stride_frame,
                                                    int frames, const vslam_orb_params *params, const int8_t *pattern,
                                                    vslam_orb_keypoint *out, float *angle, uint8_t *desc, int cap, int32_t *count,
                                                    int32_t *level_counts);

#ifdef __cplusplus
}
#endif
#endif
