// Host-side mirror of the reference's functor / free-function interface for the
// point-landmark path, implemented over the C ABI of include/vslam_hip.h.
//
// Same names, argument order, output-resizing behaviour and return codes as the
// reference (paths relative to the reference checkout):
//   CameraParameters, importCalibrationData      src/cameraModel.hpp:12-30
//   MeasurementPointLandmarkSingle / Bundle      src/measurementPointLandmark.hpp:9-311
//   SlamParameters, Landmark, PointLogLikelihood src/model.hpp:12-56, src/imagefeatures.h:21-27
//   logGaussian, affineTransform,
//   measurementUpdateIEKF                         src/gaussian.hpp:30-131,469-532
//   conditionGaussianOnMarginal                   src/gaussian.cpp:19-36
//   SlamProcessModel, timeUpdateContinuous        src/model.hpp:36-41, src/gaussian.hpp:337-383
//   removeBadLandmarks, matchHamming (BFMatcher)  src/SLAM.cpp:796-820, 222,355
//   Feature, detectCorners, detectAndComputeORB   src/imagefeatures.h:15-19,37, src/imagefeatures.cpp:73-96
//   chi2inv, normcdf                              src/utility.cpp:13-24
//   individualCompatibility, jointCompatibility,
//   snn                                           src/dataAssociation.h:10-12
// The reference passes Eigen objects; Eigen is not a dependency here, so the
// mirror uses two tiny column-major value types (Vector, Matrix) with the same
// resize / data / rows / cols surface.  With Eigen available the functors accept
// Eigen::VectorXd / MatrixXd unchanged: only .data(),
// .rows(), .cols(), .size() and .resize() are used.
//
// Contract violations that the reference asserts on throw std::runtime_error
// carrying the library's message.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vslam_hip.h"

namespace vslam_host {

struct Vector {
    std::vector<double> v;
    Vector() = default;
    explicit Vector(int n) : v((size_t)n, 0.0) {}
    Vector(std::initializer_list<double> l) : v(l) {}
    int rows() const { return (int)v.size(); }
    int cols() const { return 1; }
    int size() const { return (int)v.size(); }
    void resize(int n, int = 1) { v.resize((size_t)n); }
    double *data() { return v.data(); }
    const double *data() const { return v.data(); }
    double &operator()(int i) { return v[(size_t)i]; }
    double operator()(int i) const { return v[(size_t)i]; }
};

struct Matrix {                       // column-major, leading dimension == rows
    std::vector<double> v;
    int r = 0, c = 0;
    Matrix() = default;
    Matrix(int rows, int cols) : v((size_t)rows * cols, 0.0), r(rows), c(cols) {}
    int rows() const { return r; }
    int cols() const { return c; }
    int size() const { return r * c; }
    void resize(int rows, int cols) { r = rows; c = cols; v.assign((size_t)rows * cols, 0.0); }
    double *data() { return v.data(); }
    const double *data() const { return v.data(); }
    double &operator()(int i, int j) { return v[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return v[(size_t)j * r + i]; }
};

// One process-wide device context (device 0 unless VSLAM_DEVICE says otherwise).
inline vslam_ctx *context()
{
    static vslam_ctx *ctx = [] {
        vslam_ctx *c = nullptr;
        int dev = 0;
        if (const char *e = std::getenv("VSLAM_DEVICE")) dev = std::atoi(e);
        vslam_status s = vslam_ctx_create(dev, &c);
        if (s != VSLAM_OK) throw std::runtime_error(std::string("vslam: cannot create a context: ") + vslam_status_string(s));
        return c;
    }();
    return ctx;
}

inline void check(vslam_status s)
{
    if (s != VSLAM_OK)
        throw std::runtime_error(std::string("vslam: ") + vslam_status_string(s) + ": " + vslam_last_error(context()));
}

// ---- camera -------------------------------------------------------------
struct CameraParameters {
    vslam_camera cam{};
    CameraParameters() { cam.fov_deg = 120.0; }
};
inline void importCalibrationData(const std::string &calibrationFilePath, CameraParameters &param)
{
    vslam_status s = vslam_camera_from_xml(calibrationFilePath.c_str(), &param.cam);
    if (s != VSLAM_OK) throw std::runtime_error("vslam: cannot read " + calibrationFilePath);
}

// ---- measurement model ----------------------------------------------------
// layout: VSLAM_LAYOUT_COMPAT reproduces the reference's Jacobian column placement.
struct MeasurementPointLandmarkSingle {
    int layout = VSLAM_LAYOUT_COMPAT;
    template <class Vec> int operator()(const int &j, const Vec &x, const CameraParameters &param, Vec &rQOi) const
    {
        int32_t code = 0;
        rQOi.resize(2, 1);
        check(vslam_single_measure(context(), &param.cam, j, x.data(), (int)x.rows(), layout, rQOi.data(), nullptr, nullptr, &code));
        return code;
    }
    template <class Vec, class Mat> int operator()(const int &j, const Vec &x, const CameraParameters &param, Vec &rQOi, Mat &SR) const
    {
        int32_t code = 0;
        rQOi.resize(2, 1); SR.resize(2, 2);
        check(vslam_single_measure(context(), &param.cam, j, x.data(), (int)x.rows(), layout, rQOi.data(), SR.data(), nullptr, &code));
        return code;
    }
    template <class Vec, class Mat> int operator()(const int &j, const Vec &x, const CameraParameters &param, Vec &rQOi, Mat &SR, Mat &J) const
    {
        int32_t code = 0;
        rQOi.resize(2, 1); SR.resize(2, 2); J.resize(2, (int)x.rows());
        check(vslam_single_measure(context(), &param.cam, j, x.data(), (int)x.rows(), layout, rQOi.data(), SR.data(), J.data(), &code));
        return code;
    }
};

struct MeasurementPointLandmarkBundle {
    int layout = VSLAM_LAYOUT_COMPAT;
    template <class Vec> void operator()(const Vec &x, const CameraParameters &param, Vec &rQOi) const
    {
        const int L = ((int)x.rows() - 12) / 3;
        rQOi.resize(2 * L, 1);
        check(vslam_bundle_measure(context(), &param.cam, x.data(), (int)x.rows(), layout, rQOi.data(), nullptr, nullptr, nullptr));
    }
    template <class Vec, class Mat> void operator()(const Vec &x, const CameraParameters &param, Vec &rQOi, Mat &SR) const
    {
        const int L = ((int)x.rows() - 12) / 3;
        rQOi.resize(2 * L, 1); SR.resize(2 * L, 2 * L);
        check(vslam_bundle_measure(context(), &param.cam, x.data(), (int)x.rows(), layout, rQOi.data(), SR.data(), nullptr, nullptr));
    }
    template <class Vec, class Mat> void operator()(const Vec &x, const CameraParameters &param, Vec &rQOi, Mat &SR, Mat &C) const
    {
        const int L = ((int)x.rows() - 12) / 3;
        rQOi.resize(2 * L, 1); SR.resize(2 * L, 2 * L); C.resize(2 * L, (int)x.rows());
        check(vslam_bundle_measure(context(), &param.cam, x.data(), (int)x.rows(), layout, rQOi.data(), SR.data(), C.data(), nullptr));
    }
};

// ---- log-likelihood ---------------------------------------------------------
struct Landmark {                      // the fields of the reference's Landmark that the likelihood reads
    bool isVisible = false;
    double pixel_measurement[2] = {0.0, 0.0};
};
struct SlamParameters {
    CameraParameters camera_param;
    double measurement_noise = 7.0;
    std::vector<Landmark> landmarks;
    Matrix position_tune{3, 3};        // process noise factors (src/model.hpp:17-18)
    Matrix orientation_tune{3, 3};
};

namespace detail {
inline void pack(const SlamParameters &p, std::vector<double> &y, std::vector<uint8_t> &vis)
{
    y.resize(2 * p.landmarks.size());
    vis.resize(p.landmarks.size());
    for (size_t j = 0; j < p.landmarks.size(); ++j) {
        y[2 * j] = p.landmarks[j].pixel_measurement[0];
        y[2 * j + 1] = p.landmarks[j].pixel_measurement[1];
        vis[j] = p.landmarks[j].isVisible ? 1 : 0;
    }
}
}  // namespace detail

struct PointLogLikelihood {
    // y and u are ignored, as in the reference (measurements travel in param.landmarks).
    template <class Vec> double operator()(const Vec &, const Vec &x, const Vec &, const SlamParameters &param) const
    {
        std::vector<double> y; std::vector<uint8_t> vis; detail::pack(param, y, vis);
        double val = 0.0;
        check(vslam_point_loglik(context(), &param.camera_param.cam, x.data(), (int)x.rows(), y.data(), vis.data(),
                                 param.measurement_noise, &val, nullptr, nullptr, 0, nullptr));
        return val;
    }
    template <class Vec> double operator()(const Vec &, const Vec &x, const Vec &, const SlamParameters &param, Vec &g) const
    {
        std::vector<double> y; std::vector<uint8_t> vis; detail::pack(param, y, vis);
        double val = 0.0;
        g.resize((int)x.rows(), 1);
        check(vslam_point_loglik(context(), &param.camera_param.cam, x.data(), (int)x.rows(), y.data(), vis.data(),
                                 param.measurement_noise, &val, g.data(), nullptr, 0, nullptr));
        return val;
    }
    template <class Vec, class Mat> double operator()(const Vec &, const Vec &x, const Vec &, const SlamParameters &param, Vec &g, Mat &H) const
    {
        std::vector<double> y; std::vector<uint8_t> vis; detail::pack(param, y, vis);
        double val = 0.0;
        const int n = (int)x.rows();
        g.resize(n, 1); H.resize(n, n);
        check(vslam_point_loglik(context(), &param.camera_param.cam, x.data(), n, y.data(), vis.data(),
                                 param.measurement_noise, &val, g.data(), H.data(), n, nullptr));
        return val;
    }
};

// ---- Gaussian algebra ---------------------------------------------------------
template <class Vec, class Mat> double logGaussian(const Vec &x, const Vec &mu, const Mat &S)
{
    double v = 0.0;
    check(vslam_log_gaussian(context(), x.data(), mu.data(), S.data(), (int)x.rows(), (int)S.rows(), &v, nullptr, nullptr));
    return v;
}
template <class Vec, class Mat> double logGaussian(const Vec &x, const Vec &mu, const Mat &S, Vec &g)
{
    double v = 0.0;
    g.resize((int)x.rows(), 1);
    check(vslam_log_gaussian(context(), x.data(), mu.data(), S.data(), (int)x.rows(), (int)S.rows(), &v, g.data(), nullptr));
    return v;
}
template <class Vec, class Mat> double logGaussian(const Vec &x, const Vec &mu, const Mat &S, Vec &g, Mat &H)
{
    double v = 0.0;
    g.resize((int)x.rows(), 1); H.resize((int)x.rows(), (int)x.rows());
    check(vslam_log_gaussian(context(), x.data(), mu.data(), S.data(), (int)x.rows(), (int)S.rows(), &v, g.data(), H.data()));
    return v;
}

// affineTransform(mux, Sxx, h, muy, Syy): h is any functor h(x, y, SR, C) (src/gaussian.hpp:85-131).
template <class Func, class Vec, class Mat> void affineTransform(const Vec &mux, const Mat &Sxx, Func h, Vec &muy, Mat &Syy)
{
    if (mux.size() <= 0 || Sxx.cols() != Sxx.rows() || mux.rows() != Sxx.rows()) throw std::runtime_error("affineTransform: bad input shapes");
    Mat SR, C;
    h(mux, muy, SR, C);
    const int m = (int)muy.size(), n = (int)mux.size();
    if (m <= 0 || SR.rows() != m || SR.cols() != m || C.rows() != m || C.cols() != n) throw std::runtime_error("affineTransform: bad model output shapes");
    Syy.resize(m, m);
    check(vslam_affine_sqrt(context(), Sxx.data(), n, (int)Sxx.rows(), C.data(), m, (int)C.rows(), SR.data(), (int)SR.rows(), Syy.data(), m));
}
// The call of src/SLAM.cpp:371-387 (h = MeasurementPointLandmarkBundle bound to the camera), fused on the device.
template <class Vec, class Mat>
void affineTransform(const Vec &mux, const Mat &Sxx, const MeasurementPointLandmarkBundle &h, const CameraParameters &param, Vec &muy, Mat &Syy)
{
    const int n = (int)mux.size(), m = 2 * ((n - 12) / 3);
    muy.resize(m, 1); Syy.resize(m, m);
    check(vslam_affine_bundle(context(), &param.cam, mux.data(), Sxx.data(), n, (int)Sxx.rows(), h.layout, muy.data(), Syy.data(), m, nullptr));
}

template <class Vec, class Mat>
void conditionGaussianOnMarginal(const Vec &muyxjoint, const Mat &Syxjoint, const Vec &y, Vec &muxcond, Mat &Sxcond)
{
    const int ny = (int)y.rows(), nx = (int)Syxjoint.cols() - ny;
    muxcond.resize(nx, 1); Sxcond.resize(nx, nx);
    check(vslam_condition_on_marginal(context(), muyxjoint.data(), Syxjoint.data(), ny, nx, (int)Syxjoint.rows(), y.data(),
                                      muxcond.data(), Sxcond.data(), nx));
}

// measurementUpdateIEKF(mux, Sxx, u, y, PointLogLikelihood, param, muxGy, SxxGy), src/gaussian.hpp:469-532
// as called at src/SLAM.cpp:492.  Returns fmin's code (0 converged, 1 iteration limit).
template <class Vec, class Mat>
int measurementUpdateIEKF(const Vec &mux, const Mat &Sxx, const Vec &, const Vec &, PointLogLikelihood, const SlamParameters &param,
                          Vec &muxGy, Mat &SxxGy)
{
    std::vector<double> y; std::vector<uint8_t> vis; detail::pack(param, y, vis);
    const int n = (int)mux.size();
    muxGy.resize(n, 1); SxxGy.resize(n, n);
    int32_t code = 0;
    check(vslam_measurement_update_iekf(context(), &param.camera_param.cam, mux.data(), Sxx.data(), n, (int)Sxx.rows(), y.data(), vis.data(),
                                        param.measurement_noise, muxGy.data(), SxxGy.data(), n, &code, nullptr, nullptr));
    return code;
}

// timeUpdateContinuous(mukm1, Skm1, u, SlamProcessModel, param, timestep, muk, Sk), src/gaussian.hpp:337-383
// as called at src/SLAM.cpp:250,337,507.  u is unused, as in the reference's process model.
struct SlamProcessModel {};
template <class Vec, class Mat>
void timeUpdateContinuous(const Vec &mukm1, const Mat &Skm1, const Vec &, SlamProcessModel, const SlamParameters &param, double timestep,
                          Vec &muk, Mat &Sk)
{
    const int n = (int)mukm1.size();
    if (n <= 0 || Skm1.rows() != n || Skm1.cols() != n) throw std::runtime_error("timeUpdateContinuous: bad input shapes");
    muk.resize(n, 1); Sk.resize(n, n);
    check(vslam_time_update(context(), mukm1.data(), Skm1.data(), n, n, param.position_tune.data(), param.orientation_tune.data(),
                            timestep, muk.data(), Sk.data(), n));
}

// removeBadLandmarks(mup, Sp, landmarks, j), src/SLAM.cpp:796-820: landmark j leaves the list, the mean and the factor.
template <class Vec, class Mat, class LandmarkT>
void removeBadLandmarks(Vec &mup, Mat &Sp, std::vector<LandmarkT> &landmarks, int j)
{
    const int n = (int)mup.size();
    Vec mu_in = mup;
    Mat S_in = Sp;
    if (j < 0 || j >= (int)landmarks.size()) throw std::runtime_error("removeBadLandmarks: bad landmark index");
    landmarks.erase(landmarks.begin() + j);
    mup.resize(n - 3, 1); Sp.resize(n - 3, n - 3);
    check(vslam_remove_landmark(context(), mu_in.data(), S_in.data(), n, (int)S_in.rows(), j, mup.data(), Sp.data(), n - 3));
}

// cv::BFMatcher(cv::NORM_HAMMING).match(query, train, matches), src/SLAM.cpp:222,355, on row-major
// descriptor arrays: the train index (and distance) of the nearest descriptor for every query row.
// ---- selection of new landmarks (src/SLAM.cpp:449-472) and pixelDistance (src/SLAM.cpp:746-760) ----
struct KeyPoint { float x = 0.0f, y = 0.0f, response = 0.0f; };   // the fields of cv::KeyPoint the loop reads (pt.x, pt.y, response)
inline bool pixelDistance(const std::vector<KeyPoint> &landmark_keypoints, const KeyPoint &keypoint, double pixel_distance_thresh)
{
    std::vector<float> lm(2 * landmark_keypoints.size());
    for (size_t i = 0; i < landmark_keypoints.size(); ++i) { lm[2 * i] = landmark_keypoints[i].x; lm[2 * i + 1] = landmark_keypoints[i].y; }
    return vslam_pixel_distance(lm.data(), (int)landmark_keypoints.size(), keypoint.x, keypoint.y, pixel_distance_thresh) != 0;
}
// sorted: keypoints sorted by response, descending (sortByScore, src/imagefeatures.cpp:20-22) with integer pixel
// positions, as the detectors of this library return them.  Returns the indices that become landmarks, in order.
inline std::vector<int32_t> selectNewLandmarks(const std::vector<vslam_feature> &sorted, const std::vector<KeyPoint> &landmark_keypoints,
                                               double feature_thresh, double initial_width_thresh, double initial_height_thresh,
                                               double initial_pixel_distance_thresh, int image_width, int image_height, int max_new_landmarks)
{
    std::vector<float> lm(2 * landmark_keypoints.size());
    for (size_t i = 0; i < landmark_keypoints.size(); ++i) { lm[2 * i] = landmark_keypoints[i].x; lm[2 * i + 1] = landmark_keypoints[i].y; }
    vslam_select_params sp{feature_thresh, initial_width_thresh, initial_height_thresh, initial_pixel_distance_thresh,
                           image_width, image_height, max_new_landmarks, 0};
    std::vector<int32_t> sel((size_t)(max_new_landmarks > 0 ? max_new_landmarks : 1));
    int32_t count = 0;
    check(vslam_select_new_landmarks(context(), sorted.empty() ? nullptr : sorted.data(), (int)sorted.size(), lm.empty() ? nullptr : lm.data(),
                                     (int)landmark_keypoints.size(), &sp, sel.data(), (int)sel.size(), &count));
    sel.resize((size_t)std::min<int32_t>(count, (int32_t)sel.size()));
    return sel;
}

// generateOpticalRay(param, pixel, p), src/SLAM.cpp:876-903: the ray of a pixel in camera coordinates, optical_ray_length long in z.
inline Vector generateOpticalRay(const CameraParameters &param, double u, double v, double optical_ray_length)
{
    Vector r(3);
    if (vslam_optical_ray(&param.cam, u, v, optical_ray_length, r.data()) != 0) throw std::runtime_error("vslam: generateOpticalRay: unusable camera constants");
    return r;
}
// initializeNewLandmark(...), src/SLAM.cpp:905-941, for all keypoints selected in a frame at once: the predicted state
// (mup, Sp) and the filtered state (muEKF, SEKF) grow by three rows per keypoint; Sp and SEKF get kappa on the new
// diagonal, mup the points on the optical rays, muEKF zeros (src/SLAM.cpp:908-915,929).  The Landmark bookkeeping
// (src/SLAM.cpp:931-937) stays with the caller.
template <class Vec, class Mat>
void initializeNewLandmarks(Vec &mup, Mat &Sp, Vec &muEKF, Mat &SEKF, const std::vector<KeyPoint> &keypoints, const CameraParameters &param,
                            double optical_ray_length, double kappa)
{
    const int n = (int)mup.size(), M = (int)keypoints.size(), m = n + 3 * M;
    if (M == 0) return;
    std::vector<double> px(2 * (size_t)M);
    for (int t = 0; t < M; ++t) { px[2 * t] = keypoints[(size_t)t].x; px[2 * t + 1] = keypoints[(size_t)t].y; }
    Vec mu_in = mup, muE_in = muEKF;
    Mat S_in = Sp, SE_in = SEKF;
    mup.resize(m, 1); Sp.resize(m, m); muEKF.resize(m, 1); SEKF.resize(m, m);
    check(vslam_initialize_landmarks(context(), &param.cam, mu_in.data(), S_in.data(), n, (int)S_in.rows(), px.data(), M, optical_ray_length,
                                     kappa, mup.data(), Sp.data(), m));
    check(vslam_initialize_landmarks(context(), &param.cam, muE_in.data(), SE_in.data(), n, (int)SE_in.rows(), px.data(), M,
                                     optical_ray_length, kappa, muEKF.data(), SEKF.data(), m));
    for (int i = n; i < m; ++i) muEKF(i) = 0.0;
}

struct Match { int queryIdx, trainIdx; float distance; };       // the fields of cv::DMatch the caller reads
inline void matchHamming(const uint8_t *query, int nq, const uint8_t *train, int nt, int desc_bytes, std::vector<Match> &matches)
{
    std::vector<int32_t> idx((size_t)nq), dist((size_t)nq);
    check(vslam_match_hamming(context(), query, nq, train, nt, desc_bytes, idx.data(), dist.data()));
    matches.clear();
    for (int q = 0; q < nq; ++q)
        if (idx[(size_t)q] >= 0) matches.push_back(Match{q, idx[(size_t)q], (float)dist[(size_t)q]});
}

// measurementUpdateIEKF solves its trust-region subproblems on the device from Cholesky factors by default
// (falling back to the reference's host eigendecomposition when a Hessian is not positive definite; see
// vslam_hip.h).  useDeviceTrustRegion(false) selects the eigendecomposition route outright.
inline void useDeviceTrustRegion(bool on) { check(vslam_ctx_set_trs_solver(context(), on ? VSLAM_TRS_CHOLESKY : VSLAM_TRS_EIG)); }

// ---- data association -------------------------------------------------------
inline double chi2inv(double p, double nu) { return vslam_chi2inv(p, nu); }
inline double normcdf(double z) { return vslam_normcdf(z); }

// y is 2 x m (features in columns), SYY upper triangular (2n x 2n).
template <class Mat, class Vec>
bool individualCompatibility(const int &i, const int &j, const int &ny, const Mat &y, const Vec &muY, const Mat &SYY,
                             const std::vector<double> &chi2LUT)
{
    if (ny != 2 || (int)y.rows() != ny || i < 0 || j < 0 || i >= (int)y.cols()) throw std::runtime_error("individualCompatibility: bad arguments");
    const int n = (int)muY.rows() / 2, m = (int)y.cols();
    std::vector<uint8_t> all((size_t)m * n);
    check(vslam_individual_compat(context(), y.data(), m, muY.data(), SYY.data(), n, (int)SYY.rows(), chi2LUT.data(), all.data()));
    return all[(size_t)i * n + j] != 0;
}

template <class Mat, class Vec>
bool jointCompatibility(const std::vector<int> &idx, const double &sU, const int &ny, const Mat &y, const Vec &muY, const Mat &SYY,
                        const std::vector<double> &chi2LUT, double &surprisal)
{
    if (ny != 2 || (int)y.rows() != ny || (int)idx.size() * ny != (int)muY.rows()) throw std::runtime_error("jointCompatibility: bad arguments");
    std::vector<int32_t> id(idx.begin(), idx.end());
    int32_t jc = 0;
    check(vslam_joint_compat(context(), id.data(), (int)id.size(), sU, y.data(), (int)y.cols(), muY.data(), SYY.data(), (int)SYY.rows(),
                             chi2LUT.data(), &surprisal, &jc));
    return jc != 0;
}

// snn(mux, Sxx, y, param, idx, enforceJointCompatibility), src/dataAssociation.cpp:25-150; returns the surprisal.
template <class Vec, class Mat>
double snn(const Vec &mux, const Mat &Sxx, const Mat &y, const CameraParameters &param, std::vector<int> &idx,
           bool enforceJointCompatibility = false, int layout = VSLAM_LAYOUT_COMPAT)
{
    const int n = ((int)mux.rows() - 12) / 3;
    std::vector<int32_t> id((size_t)(n > 0 ? n : 0));
    double s = 0.0;
    check(vslam_snn(context(), &param.cam, mux.data(), Sxx.data(), (int)mux.rows(), (int)Sxx.rows(), layout, y.data(), (int)y.cols(),
                    enforceJointCompatibility ? 1 : 0, id.data(), &s));
    idx.assign(id.begin(), id.end());
    return s;
}

// ---- corner front end ---------------------------------------------------------
using Feature = vslam_feature;         // {int x; int y; float score;}, src/imagefeatures.h:15-19
// Keypoints of one 8-bit grayscale frame sorted by score (sortByScore, src/imagefeatures.cpp:20-22).
inline int detectCorners(const uint8_t *gray, int width, int height, int stride, int maxNumFeatures, float threshold,
                         int detector, std::vector<Feature> &keypoints)
{
    vslam_corner_params p{};
    p.block_size = 7; p.border = 5; p.max_features = maxNumFeatures; p.harris_k = 0.04f;
    p.harris_thresh = threshold; p.shi_thresh = threshold;
    keypoints.resize((size_t)maxNumFeatures);
    int32_t count = 0;
    check(vslam_corner_detect(context(), gray, width, height, stride, &p, detector, keypoints.data(), maxNumFeatures, &count));
    keypoints.resize((size_t)count);
    return 0;
}

// ---- ORB front end ---------------------------------------------------------------
// detectAndDrawORB's orb->detect + orb->compute (src/imagefeatures.cpp:73-96) without the drawing: keypoints of all
// pyramid levels (level 0 first, by response within a level), their angles in degrees and 32-byte descriptors
// (row i of `descriptors` belongs to keypoints[i]; the layout of a continuous CV_8U cv::Mat with 32 columns).
// `pattern` is cv::ORB's 256 x 4 test table as signed bytes, or nullptr for the library's own table.
using OrbKeypoint = vslam_orb_keypoint;        // {x, y, response, octave}
inline int detectAndComputeORB(const uint8_t *gray, int width, int height, int stride, int maxNumFeatures,
                               std::vector<OrbKeypoint> &keypoints, std::vector<float> &angles, std::vector<uint8_t> &descriptors,
                               const int8_t *pattern = nullptr)
{
    vslam_orb_params p{};
    p.nfeatures = maxNumFeatures; p.scale_factor = 1.2f; p.nlevels = 8; p.edge_threshold = 75; p.fast_threshold = 20;
    p.harris_block = 7; p.harris_k = 0.04f;
    const int cap = 2 * maxNumFeatures + 1024;             // ties at a level's cut are kept, as cv::ORB keeps them
    keypoints.resize((size_t)cap); angles.resize((size_t)cap); descriptors.resize((size_t)cap * 32);
    int32_t count = 0;
    check(vslam_orb_detect_and_compute(context(), gray, width, height, stride, &p, pattern, keypoints.data(), angles.data(),
                                       descriptors.data(), cap, &count, nullptr));
    keypoints.resize((size_t)count); angles.resize((size_t)count); descriptors.resize((size_t)count * 32);
    return 0;
}

}  // namespace vslam_host
