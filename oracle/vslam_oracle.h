/*
 * CPU oracle for the VisualSLAM point-landmark hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar, single-threaded restatement of
 * the reference algorithm (Tom0Brien/VisualSLAM) used to check the HIP
 * kernels.  Nothing in the shipped library (libvslam_hip.so) links, loads or
 * calls it; only tests/, bench.py's cpu_baseline leg and
 * __graft_entry__.smoke() may.
 *
 * Pinning: the reference cannot be built in this environment (Eigen, OpenCV,
 * autodiff, Boost, Catch2, VTK are absent), so the oracle is pinned by
 *   - the reference's own known-answer tests where they exist
 *     (test/src/logGaussian.cpp, test/src/affineTransform.cpp,
 *      test/src/rpy2rot.cpp, test/src/rot2rpy.cpp), and
 *   - 40-digit sympy/mpmath vectors generated from the cited formulas
 *     (tests/golden/make_golden.py).
 * worldToPixel pixel values, MeasurementPointLandmark outputs, the point
 * log-likelihood derivatives, measurementUpdateIEKF and every image-feature
 * output have no reference test: for those, parity is UNPINNED by the
 * reference and rests on the restatement + the symbolic vectors.
 *
 * All matrices are column-major doubles, like Eigen::MatrixXd.
 */
#ifndef VSLAM_ORACLE_H
#define VSLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as vslam_camera in include/vslam_hip.h (152 bytes). */
typedef struct {
    double fx, fy, cx, cy;
    double d[12];      /* OpenCV order: k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 */
    double fov_deg;    /* CameraParameters::fieldOfView, default 120 */
    int32_t flag;      /* CALIB_RATIONAL_MODEL 0x4000 | CALIB_THIN_PRISM_MODEL 0x8000 */
    int32_t width, height;
    int32_t reserved_;
} orc_camera;

/* src/rotation.hpp:13-63 */
void orc_rotx(double a, double *R);
void orc_roty(double a, double *R);
void orc_rotz(double a, double *R);
void orc_rpy2rot(const double *theta, double *R);
void orc_rot2rpy(const double *R, double *theta);

/* src/cameraModel.hpp:32-216.  Returns 0 in view, 1 outside cone / behind
 * camera, 2 outside image.  px is written exactly when the reference assigns
 * it (codes 0 and 2); on code 1 it is left untouched. */
int orc_world_to_pixel(const double *rPNn, const double *eta, const orc_camera *cam, double *px);

/* src/measurementPointLandmark.hpp:10-304.  layout 0 = COMPAT (reference
 * column placement and x(3..5) rotation-derivative quirk), 1 = STRICT.
 * h[2], SR[2*2] (may be NULL), J[2*n] col-major (may be NULL). */
int orc_single_measure(int j, const double *x, int n, const orc_camera *cam, int layout,
                       double *h, double *SR, double *J);

/* src/measurementPointLandmark.cpp:13-70.  h[2L], SR[2L*2L] or NULL,
 * C[2L*n] or NULL, code[L].  Culled rows of h: code 2 keeps the computed
 * pixel, code 1 is NaN (the reference leaves it stale). */
void orc_bundle_measure(const double *x, int n, const orc_camera *cam, int layout,
                        double *h, double *SR, double *C, int32_t *code);

/* src/model.cpp:369-406: value of the point log-likelihood. */
double orc_point_loglik_value(const double *x, int n, const double *y, const uint8_t *visible,
                              const orc_camera *cam, double sigma, int32_t *code);

/* src/model.cpp:427-437 as autodiff::hessian evaluates it: n(n+1)/2 passes of
 * the whole likelihood on second-order dual numbers.  row_begin/row_end bound
 * the seed rows i (for bounded timing samples); pass 0,n for everything.
 * Returns the number of full likelihood passes done. */
long orc_point_loglik_autodiff(const double *x, int n, const double *y, const uint8_t *visible,
                               const orc_camera *cam, double sigma,
                               int row_begin, int row_end,
                               double *value, double *g, double *H);

/* Same derivatives, but dual passes only over the 45 seed pairs inside each
 * landmark's 9 active variables (every other second derivative is exactly 0). */
void orc_point_loglik_dual_sparse(const double *x, int n, const double *y, const uint8_t *visible,
                                  const orc_camera *cam, double sigma,
                                  double *value, double *g, double *H, int32_t *code);

/* Closed-form gradient and exact Hessian (chain rule), scalar fp64. */
void orc_point_loglik_analytic(const double *x, int n, const double *y, const uint8_t *visible,
                               const orc_camera *cam, double sigma,
                               double *value, double *g, double *H, int32_t *code);

/* src/gaussian.hpp:30-74.  S upper triangular (ld = ldS).  g, H may be NULL. */
double orc_log_gaussian(const double *x, const double *mu, const double *S, int n, int ldS,
                        double *g, double *H);

/* Householder QR in place, LAPACK/Eigen sign convention; on return the upper
 * triangle of A holds R (strict lower part zeroed).  Eigen::HouseholderQR as
 * used at src/gaussian.hpp:121-124,515-516. */
void orc_qr_upper(double *A, int rows, int cols, int lda);

/* src/gaussian.hpp:117-127: Syy = triu(qr([S*C^T ; SR]))[0:m,0:m]. */
void orc_affine_sqrt(const double *S, int n, int ldS, const double *C, int m, int ldC,
                     const double *SR, int ldSR, double *Syy, int ldSyy);

/* src/gaussian.cpp:19-36 */
void orc_condition_on_marginal(const double *muyx, const double *Syx, int ny, int nx, int ld,
                               const double *y, double *mux_out, double *Sx_out, int ldo);

/* src/gaussian.hpp:513-516: S = triu(qr(diag(1/sqrt(v)) Q^T)). */
void orc_iekf_post_sqrt(const double *Q, const double *v, int n, double *S_out);

/* Symmetric eigendecomposition (cyclic Jacobi), ascending eigenvalues, reads
 * the lower triangle like Eigen::SelfAdjointEigenSolver (src/fmin.hpp:42-44). */
void orc_sym_eig(const double *H, int n, double *Q, double *v);

/* src/fmin.cpp:40-131 */
int orc_trs(const double *Q, const double *v, const double *g, int n, double D, double *p);

/* src/gaussian.hpp:469-532 + src/fmin.hpp:19-107 with the point likelihood
 * (analytic derivatives).  Returns fmin's code (0 converged, 1 max iter);
 * iters/cost_evals are optional outputs. */
int orc_measurement_update_iekf(const double *mu, const double *S, int n,
                                const double *y, const uint8_t *visible,
                                const orc_camera *cam, double sigma,
                                double *mu_post, double *S_post,
                                int *iters, int *cost_evals);

/* ---- time update ---- */
/* SlamProcessModel, src/model.cpp:9-136: f (n), SQ (n x n or NULL), dfdx (n x n or NULL);
 * pos_tune / ori_tune are SlamParameters::position_tune / orientation_tune (3x3 column-major). */
void orc_process_model(const double *x, int n, const double *pos_tune, const double *ori_tune,
                       double *f, double *SQ, double *dfdx);
/* RK4SDEHelper with Jacobian propagation, src/gaussian.hpp:140-239.  xdw = [x; dw] (2n);
 * xnext (n); J = d xnext / d [x; dw] (n x 2n) or NULL. */
void orc_rk4_sde(const double *xdw, int n, const double *pos_tune, const double *ori_tune, double dt,
                 double *xnext, double *J);
/* timeUpdateContinuous, src/gaussian.hpp:337-383 (dense, the 3n x n Householder QR included).
 * No reference test holds numbers for it (test/src/kalman.cpp is commented out): parity unpinned. */
void orc_time_update(const double *mu, const double *S, int n, int ldS, const double *pos_tune,
                     const double *ori_tune, double dt, double *mu_out, double *S_out, int ldo);

/* removeBadLandmarks, src/SLAM.cpp:796-820: mu_out (n-3), S_out ((n-3) x (n-3), ld ldo). */
void orc_remove_landmark(const double *mu, const double *S, int n, int ldS, int j, double *mu_out,
                         double *S_out, int ldo);
/* cv::BFMatcher(NORM_HAMMING).match, src/SLAM.cpp:222,355 (first nearest train descriptor per query). */
void orc_match_hamming(const uint8_t *query, int nq, const uint8_t *train, int nt, int desc_bytes,
                       int32_t *train_idx, int32_t *distance);

/* ---- data association (src/dataAssociation.cpp, src/utility.cpp) ---- */
/* normcdf, src/utility.cpp:20-24. */
double orc_normcdf(double z);
/* chi2inv(p, nu) = 2 * gamma_p_inv(nu/2, p), src/utility.cpp:13-18 (Boost's inverse of the regularised
 * lower incomplete gamma function restated as a bracketed Newton iteration on its series /
 * continued-fraction evaluation). */
double orc_chi2inv(double p, double nu);
/* individualCompatibility, src/dataAssociation.cpp:154-177.  y is ny x m column-major, SYY N x N. */
int orc_individual_compat(int i, int j, int ny, const double *y, const double *muY,
                          const double *SYY, int N, const double *chi2LUT);
/* jointCompatibility, src/dataAssociation.cpp:179-285 (idx has n entries, -1 = unassociated). */
int orc_joint_compat(const int *idx, int n, double sU, int ny, const double *y, const double *muY,
                     const double *SYY, const double *chi2LUT, double *surprisal);
/* snn, src/dataAssociation.cpp:25-150, on a precomputed predicted-measurement density (muY, SYY):
 * greedy surprisal nearest neighbour over landmarks j = 0..n-1 and features 0..m-1.  idx[n] out. */
double orc_snn_density(const double *muY, const double *SYY, int n, const double *y, int m,
                       int width, int height, int enforce_jc, int *idx);
/* The full snn: affineTransform through the landmark bundle first (src/dataAssociation.cpp:50-61). */
double orc_snn(const double *mux, const double *Sxx, int nx_all, const double *y, int m,
               const orc_camera *cam, int layout, int enforce_jc, int *idx);

/* ---- image front end (contract defined here; see the .c file) ---- */
typedef struct {
    int32_t block_size;      /* structure-tensor window, odd, default 7 */
    int32_t border;          /* keypoints need border <= x < w-border; >= block/2+2 */
    int32_t max_features;    /* cap per detector */
    float   harris_k;        /* 0.04 */
    float   harris_thresh;   /* keep if response > thresh */
    float   shi_thresh;
} orc_corner_params;

typedef struct { int32_t x, y; float score; } orc_feature;  /* src/imagefeatures.h:15-19 */

void orc_corner_response(const uint8_t *gray, int w, int h, int stride,
                         const orc_corner_params *p, float *harris, float *shi);
/* which: 0 = Harris, 1 = Shi-Tomasi.  Returns keypoints sorted by score
 * descending (ties: y then x ascending), at most min(cap, max_features). */
int orc_corner_detect(const uint8_t *gray, int w, int h, int stride,
                      const orc_corner_params *p, int which, orc_feature *out, int cap);
/* generateOpticalRay (src/SLAM.cpp:876-903) and initializeNewLandmark (src/SLAM.cpp:905-941). */
void orc_undistort_point(const orc_camera *cam, double u, double v, float *out_uv);
void orc_optical_ray(const orc_camera *cam, double u, double v, double length, double *r);
void orc_initialize_landmark(const double *mu, const double *S, int n, int ldS, const orc_camera *cam, double u, double v,
                             double length, double kappa, double *mu_out, double *S_out);
/* pixelDistance (src/SLAM.cpp:746-760) and the selection loop of new landmarks (src/SLAM.cpp:449-472). */
int orc_pixel_distance(const float *landmark_xy, int L, float x, float y, double thresh);
int orc_select_new_landmarks(const orc_feature *feats, int K, const float *landmark_xy, int L, double feature_thresh,
                             double width_thresh, double height_thresh, double dist_thresh, int image_w, int image_h,
                             int max_new, int32_t *selected, int cap);
/* FAST-9/16 (first stage of cv::ORB::detect, src/imagefeatures.cpp:73-87): strength of one pixel and the detector. */
int orc_fast9_strength(const uint8_t *gray, int w, int h, int stride, int x, int y);
int orc_fast9_detect(const uint8_t *gray, int w, int h, int stride, int threshold, int nonmax, int max_features,
                     orc_feature *out, int cap);
int orc_orb_detect_level(const uint8_t *gray, int w, int h, int stride, int nfeatures, int fast_threshold, int edge,
                         int block, float harris_k, orc_feature *out, int cap);
void orc_resize_bilinear(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride);
void orc_orb_level_plan(int nfeatures, float scale_factor, int nlevels, int w, int h, int *lw, int *lh, int *lf, float *ls);
/* orientation and rBRIEF descriptors of cv::ORB::compute (reference src/imagefeatures.cpp:88-96), restated; parity unpinned */
void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
void orc_orb_umax(int *umax);
float orc_fast_atan2(float y, float x);
float orc_orb_ic_angle(const uint8_t *gray, int stride, int x, int y, int *m01_out, int *m10_out);
void orc_orb_default_pattern(int8_t *pattern);
void orc_orb_describe(const uint8_t *blur, int stride, int x, int y, float angle_deg, const int8_t *pattern, uint8_t *desc);

#ifdef __cplusplus
}
#endif
#endif
