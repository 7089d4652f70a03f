// Exercises the C++ host mirror (visualslam_amd/host/vslam_host.hpp) the way the
// reference's Catch2 suite exercises its own functors: the known-answer cases of
// test/src/logGaussian.cpp:10-155 and test/src/affineTransform.cpp:7-136, the
// arity consistency checks of test/src/measurementModel.cpp, and a full
// measurementUpdateIEKF call as at src/SLAM.cpp:492.  Needs a gfx950 GPU.
//
// usage: test_host_api <camera.xml>     exit code = number of failed checks
#include <cmath>
#include <cstdio>
#include <random>

#include "../../visualslam_amd/host/vslam_host.hpp"

using namespace vslam_host;

static int failures = 0, checks = 0;
#define CHECK_NEAR(a, b, tol)                                                                          \
    do {                                                                                               \
        ++checks;                                                                                      \
        double a_ = (a), b_ = (b), t_ = (tol);                                                         \
        if (!(std::fabs(a_ - b_) <= t_ * (1.0 + std::fabs(b_)))) {                                     \
            ++failures;                                                                                \
            std::printf("FAIL %s:%d  %s = %.15g, expected %.15g\n", __FILE__, __LINE__, #a, a_, b_);   \
        }                                                                                              \
    } while (0)
#define CHECK_TRUE(c)                                                            \
    do {                                                                         \
        ++checks;                                                                \
        if (!(c)) { ++failures; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); } \
    } while (0)

// Polar-coordinate measurement functor with the published answers of the reference's affineTransform test.
struct Polar {
    void operator()(const Vector &x, Vector &f, Matrix &SR, Matrix &J) const
    {
        const double pi = 3.14159265358979323846;
        double r2 = x(0) * x(0) + x(1) * x(1), r = std::sqrt(r2);
        f.resize(2);
        f(0) = std::atan2(x(0), x(1));
        f(1) = r;
        SR.resize(2, 2);
        SR(0, 0) = pi / 180.0;
        SR(1, 1) = 0.1;
        J.resize(2, 2);
        J(0, 0) = x(1) / r2;  J(0, 1) = -x(0) / r2;
        J(1, 0) = x(0) / r;   J(1, 1) = x(1) / r;
    }
};

static void test_log_gaussian()
{
    Vector x{1, 2, 3}, mu{2, 4, 6}, g;
    Matrix S(3, 3), H;
    S(0, 0) = 1; S(1, 1) = 2; S(2, 2) = 3;
    CHECK_NEAR(logGaussian(x, mu, S), -6.04857506884207, 1e-12);
    CHECK_NEAR(logGaussian(x, mu, S, g), -6.04857506884207, 1e-12);
    CHECK_TRUE(g.rows() == 3 && g.cols() == 1);
    CHECK_NEAR(g(0), 1.0, 1e-12); CHECK_NEAR(g(1), 0.5, 1e-12); CHECK_NEAR(g(2), 1.0 / 3.0, 1e-12);
    CHECK_NEAR(logGaussian(x, mu, S, g, H), -6.04857506884207, 1e-12);
    CHECK_TRUE(H.rows() == 3 && H.cols() == 3);
    CHECK_NEAR(H(0, 0), -1.0, 1e-12); CHECK_NEAR(H(1, 1), -0.25, 1e-12); CHECK_NEAR(H(2, 2), -1.0 / 9.0, 1e-12);
    CHECK_NEAR(H(0, 1), 0.0, 1e-12); CHECK_NEAR(H(2, 0), 0.0, 1e-12);
}

static void test_affine_transform()
{
    {
        Vector mux{1, 3}, muy; Matrix Sxx(2, 2), Syy;
        Sxx(0, 0) = 10; Sxx(1, 1) = 1;
        affineTransform(mux, Sxx, Polar(), muy, Syy);
        CHECK_TRUE(muy.rows() == 2 && Syy.rows() == 2 && Syy.cols() == 2);
        CHECK_NEAR(muy(0), 0.321750554397, 1e-10); CHECK_NEAR(muy(1), 3.16227766017, 1e-10);
        CHECK_NEAR(Syy(0, 0), -3.00171694492, 1e-10); CHECK_NEAR(Syy(1, 0), 0.0, 1e-12);
        CHECK_NEAR(Syy(0, 1), -3.12886418774, 1e-10); CHECK_NEAR(Syy(1, 1), -1.05839921328, 1e-10);
    }
    {
        Vector mux{1, 1}, muy; Matrix Sxx(2, 2), Syy;
        Sxx(0, 0) = 1.5; Sxx(1, 1) = 1.8;
        affineTransform(mux, Sxx, Polar(), muy, Syy);
        CHECK_NEAR(muy(0), 0.785398163397, 1e-10); CHECK_NEAR(muy(1), 1.41421356237, 1e-10);
        CHECK_NEAR(Syy(0, 0), -1.17166745172, 1e-10); CHECK_NEAR(Syy(1, 0), 0.0, 1e-12);
        CHECK_NEAR(Syy(0, 1), 0.298734812658, 1e-10); CHECK_NEAR(Syy(1, 1), -1.63271476741, 1e-10);
    }
}

// A camera 2 m behind a fronto-parallel cloud of landmarks; measurements = projections + noise.
struct Scene {
    SlamParameters param;
    Vector x, mu;
    Matrix S;
    Scene(const std::string &xml, int L, unsigned seed)
    {
        importCalibrationData(xml, param.camera_param);
        param.measurement_noise = 7.0;
        const int n = 12 + 3 * L;
        std::mt19937 rng(seed);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        std::normal_distribution<double> N01(0.0, 1.0);
        x.resize(n);
        for (int i = 0; i < 6; ++i) x(i) = 0.01 * U(rng);
        x(6) = 0.05 * U(rng); x(7) = 0.05 * U(rng); x(8) = 0.05 * U(rng);
        x(9) = 0.02 * U(rng); x(10) = 0.02 * U(rng); x(11) = 0.02 * U(rng);
        // Camera z axis is the optical axis; with near-zero angles it is the world z axis.
        for (int j = 0; j < L; ++j) {
            x(12 + 3 * j) = 0.8 * U(rng);
            x(13 + 3 * j) = 0.5 * U(rng);
            x(14 + 3 * j) = 2.0 + 0.5 * U(rng);
        }
        MeasurementPointLandmarkBundle h;
        Vector y;
        h(x, param.camera_param, y);
        param.landmarks.resize((size_t)L);
        for (int j = 0; j < L; ++j) {
            param.landmarks[(size_t)j].isVisible = (j % 5) != 4;
            param.landmarks[(size_t)j].pixel_measurement[0] = y(2 * j) + 3.0 * N01(rng);
            param.landmarks[(size_t)j].pixel_measurement[1] = y(2 * j + 1) + 3.0 * N01(rng);
        }
        mu = x;
        for (int i = 6; i < n; ++i) mu(i) += 0.01 * N01(rng);
        S.resize(n, n);
        for (int i = 0; i < n; ++i) {
            S(i, i) = i < 12 ? 0.05 : 0.2;
            for (int j = i + 1; j < n; ++j) S(i, j) = 0.002 * U(rng);
        }
    }
};

static void test_measurement_model(const std::string &xml)
{
    Scene sc(xml, 6, 3);
    const int n = sc.x.rows();
    MeasurementPointLandmarkBundle hb;
    MeasurementPointLandmarkSingle hs;
    Vector yb1, yb2, yb3, y1, y2, y3;
    Matrix SRb2, SRb3, C, SR2, SR3, J;
    hb(sc.x, sc.param.camera_param, yb1);
    hb(sc.x, sc.param.camera_param, yb2, SRb2);
    hb(sc.x, sc.param.camera_param, yb3, SRb3, C);
    CHECK_TRUE(yb1.rows() == 12 && SRb2.rows() == 12 && SRb2.cols() == 12 && C.rows() == 12 && C.cols() == n);
    for (int j = 0; j < 6; ++j) {
        int c1 = hs(j, sc.x, sc.param.camera_param, y1);
        int c2 = hs(j, sc.x, sc.param.camera_param, y2, SR2);
        int c3 = hs(j, sc.x, sc.param.camera_param, y3, SR3, J);
        CHECK_TRUE(c1 == 0 && c2 == 0 && c3 == 0);
        CHECK_TRUE(J.rows() == 2 && J.cols() == n);
        for (int r = 0; r < 2; ++r) {
            CHECK_NEAR(y1(r), yb1(2 * j + r), 1e-13);
            CHECK_NEAR(y2(r), yb2(2 * j + r), 1e-13);
            CHECK_NEAR(y3(r), yb3(2 * j + r), 1e-13);
            CHECK_NEAR(SR3(r, r), 0.01, 1e-15);
            CHECK_NEAR(SRb3(2 * j + r, 2 * j + r), 0.01, 1e-15);
            for (int c = 0; c < n; ++c) CHECK_NEAR(J(r, c), C(2 * j + r, c), 1e-12);
        }
    }
    // Fused device form of affineTransform(mux, Sxx, bundle, ...) against the generic one.
    Vector muy_a, muy_b;
    Matrix Syy_a, Syy_b;
    affineTransform(sc.x, sc.S, hb, sc.param.camera_param, muy_a, Syy_a);
    auto bound = [&](const Vector &xx, Vector &f, Matrix &SR, Matrix &CC) { hb(xx, sc.param.camera_param, f, SR, CC); };
    affineTransform(sc.x, sc.S, bound, muy_b, Syy_b);
    for (int i = 0; i < 12; ++i) {
        CHECK_NEAR(muy_a(i), muy_b(i), 1e-12);
        for (int j = 0; j < 12; ++j) CHECK_NEAR(Syy_a(i, j), Syy_b(i, j), 1e-9);
    }
}

static void test_point_log_likelihood(const std::string &xml)
{
    Scene sc(xml, 8, 11);
    const int n = sc.x.rows();
    PointLogLikelihood ll;
    Vector y, u, g2, g3;
    Matrix H;
    double v1 = ll(y, sc.x, u, sc.param);
    double v2 = ll(y, sc.x, u, sc.param, g2);
    double v3 = ll(y, sc.x, u, sc.param, g3, H);
    CHECK_TRUE(g2.rows() == n && H.rows() == n && H.cols() == n);
    CHECK_NEAR(v2, v1, 1e-13); CHECK_NEAR(v3, v1, 1e-13);
    for (int i = 0; i < n; ++i) CHECK_NEAR(g3(i), g2(i), 1e-13);
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) CHECK_NEAR(H(i, j), H(j, i), 1e-13);
    for (int i = 0; i < 6; ++i) CHECK_NEAR(g2(i), 0.0, 0.0);   // the velocity states do not enter the likelihood
    // Central differences of the value against the gradient, and of the gradient against the Hessian.
    const double h = 1e-6;
    for (int i = 6; i < n; i += 3) {
        Vector xp = sc.x, xm = sc.x, gp, gm;
        xp(i) += h; xm(i) -= h;
        double fp = ll(y, xp, u, sc.param, gp), fm = ll(y, xm, u, sc.param, gm);
        CHECK_NEAR((fp - fm) / (2 * h), g2(i), 1e-6);
        for (int k = 6; k < n; k += 5) CHECK_NEAR((gp(k) - gm(k)) / (2 * h), H(k, i), 1e-5);
    }
}

static void test_measurement_update(const std::string &xml)
{
    Scene sc(xml, 10, 5);
    const int n = sc.x.rows();
    Vector u, y, muxGy, g;
    Matrix SxxGy;
    int code = measurementUpdateIEKF(sc.mu, sc.S, u, y, PointLogLikelihood(), sc.param, muxGy, SxxGy);
    CHECK_TRUE(code == 0);
    CHECK_TRUE(muxGy.rows() == n && SxxGy.rows() == n && SxxGy.cols() == n);
    for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) CHECK_TRUE(SxxGy(i, j) == 0.0);
    // The posterior mean is a stationary point of -(logprior + loglik).
    Vector gl, gp;
    PointLogLikelihood ll;
    ll(y, muxGy, u, sc.param, gl);
    logGaussian(muxGy, sc.mu, sc.S, gp);
    double gn = 0.0, scale = 0.0;
    for (int i = 0; i < n; ++i) { gn += (gl(i) + gp(i)) * (gl(i) + gp(i)); scale += gl(i) * gl(i) + gp(i) * gp(i); }
    CHECK_TRUE(std::sqrt(gn) <= 1e-6 * (1.0 + std::sqrt(scale)));
    // The opt-in device route reaches the same posterior (S up to row signs: compare S^T S on the diagonal).
    Vector mu2; Matrix S2;
    useDeviceTrustRegion(true);
    int code2 = measurementUpdateIEKF(sc.mu, sc.S, u, y, PointLogLikelihood(), sc.param, mu2, S2);
    useDeviceTrustRegion(false);
    CHECK_TRUE(code2 == 0);
    for (int j = 0; j < n; ++j) {
        CHECK_NEAR(mu2(j), muxGy(j), 1e-7);
        double a = 0.0, b = 0.0;
        for (int i = 0; i <= j; ++i) { a += S2(i, j) * S2(i, j); b += SxxGy(i, j) * SxxGy(i, j); }
        CHECK_NEAR(a, b, 1e-7);
    }
    // The update does not inflate any marginal variance: diag(Spost' Spost) <= diag(S' S).
    for (int j = 0; j < n; ++j) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i <= j; ++i) { a += SxxGy(i, j) * SxxGy(i, j); b += sc.S(i, j) * sc.S(i, j); }
        CHECK_TRUE(a <= b * (1.0 + 1e-9));
    }
}

static void test_condition_on_marginal()
{
    // Joint of [y; x] with an uncorrelated square-root factor: conditioning on y leaves x's marginal untouched.
    Vector mu{1, 2, 3, 4}, yv{1.5, 2.5}, mc;
    Matrix S(4, 4), Sc;
    S(0, 0) = 2; S(1, 1) = 3; S(2, 2) = 0.5; S(3, 3) = 0.25; S(2, 3) = 0.1;
    conditionGaussianOnMarginal(mu, S, yv, mc, Sc);
    CHECK_TRUE(mc.rows() == 2 && Sc.rows() == 2 && Sc.cols() == 2);
    CHECK_NEAR(mc(0), 3.0, 1e-13); CHECK_NEAR(mc(1), 4.0, 1e-13);
    CHECK_NEAR(Sc(0, 0), 0.5, 1e-13); CHECK_NEAR(Sc(0, 1), 0.1, 1e-13); CHECK_NEAR(Sc(1, 1), 0.25, 1e-13);
    // With correlation S12: mu_x|y = mu_x + S12' * S11^-T (y - mu_y).
    S(0, 2) = 0.4; S(1, 3) = -0.6;
    conditionGaussianOnMarginal(mu, S, yv, mc, Sc);
    CHECK_NEAR(mc(0), 3.0 + 0.4 * (0.5 / 2.0), 1e-13);
    CHECK_NEAR(mc(1), 4.0 - 0.6 * (0.5 / 3.0), 1e-13);
}

// test/src/individualCompatibility.cpp:11-60, test/src/jointCompatibility.cpp:10-82,85-129, test/src/chi2inv.cpp
static void test_data_association(const std::string &xml)
{
    CHECK_NEAR(chi2inv(0.5, 1), 0.454936423119573, 1e-12);
    CHECK_NEAR(chi2inv(0.75, 3), 4.108344935632316, 1e-12);
    CHECK_NEAR(normcdf(1), 0.841344746068543, 1e-13);
    Vector muY{1658.780527, 963.3626597};
    Matrix SYY(2, 2), y(2, 3);
    SYY(0, 0) = -53.41061755; SYY(0, 1) = 9.901075665; SYY(1, 1) = -37.3526274;
    const double ys[2][3] = {{1754.156364, 1714.701131, 1921.012306}, {765.1206969, 934.6000308, 849.6044461}};
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) y(r, c) = ys[r][c];
    std::vector<double> lut{11.829158081900795};
    CHECK_TRUE(individualCompatibility(0, 0, 2, y, muY, SYY, lut) == false);
    CHECK_TRUE(individualCompatibility(1, 0, 2, y, muY, SYY, lut) == true);
    CHECK_TRUE(individualCompatibility(2, 0, 2, y, muY, SYY, lut) == false);
    double surprisal = 0.0;
    const double sU = 14.832478857591873;
    CHECK_TRUE(jointCompatibility(std::vector<int>{1}, sU, 2, y, muY, SYY, lut, surprisal) == true);
    CHECK_NEAR(surprisal, 10.105667572052072, 1e-7);
    CHECK_TRUE(jointCompatibility(std::vector<int>{-1}, sU, 2, y, muY, SYY, lut, surprisal) == true);
    CHECK_NEAR(surprisal, 14.832478857591873, 1e-12);
    // snn on a scene whose features are the noisy projections of its landmarks in reverse order
    Scene sc(xml, 5, 9);
    Vector h;
    MeasurementPointLandmarkBundle()(sc.x, sc.param.camera_param, h);
    Matrix feats(2, 5);
    for (int j = 0; j < 5; ++j) { feats(0, 4 - j) = h(2 * j) + 0.5; feats(1, 4 - j) = h(2 * j + 1) - 0.5; }
    std::vector<int> idx;
    const double s = snn(sc.x, sc.S, feats, sc.param.camera_param, idx);
    CHECK_TRUE(idx.size() == 5);
    for (int j = 0; j < 5; ++j) CHECK_TRUE(idx[(size_t)j] == 4 - j);
    CHECK_TRUE(s < 5 * (std::log(sc.param.camera_param.cam.width) + std::log(sc.param.camera_param.cam.height)));
}

// The case of test/src/kalman.cpp:11-72 (two landmarks, diagonal factor, timestep 0.01): output shapes
// equal input shapes; in addition the velocity and landmark means do not move and the first
// column's norm is the propagated variance of the first state.
static void test_time_update()
{
    const int n = 18;
    Vector mu(n), u, mup;
    Matrix S(n, n), Sp;
    for (int i = 0; i < n; ++i) { mu(i) = 0.1 * (i + 1); S(i, i) = i + 1; }
    SlamParameters p;
    p.position_tune(0, 0) = 0.5; p.position_tune(1, 1) = 0.5; p.position_tune(2, 2) = 0.175;
    p.orientation_tune(0, 0) = p.orientation_tune(1, 1) = p.orientation_tune(2, 2) = 0.05;
    SlamProcessModel pm;
    const double timestep = 0.01;
    timeUpdateContinuous(mu, S, u, pm, p, timestep, mup, Sp);
    CHECK_TRUE(mup.rows() == mu.rows() && mup.cols() == mu.cols());
    CHECK_TRUE(Sp.rows() == S.rows() && Sp.cols() == S.cols());
    for (int i = 0; i < 6; ++i) CHECK_NEAR(mup(i), mu(i), 0.0);
    for (int i = 12; i < n; ++i) CHECK_NEAR(mup(i), mu(i), 0.0);
    CHECK_TRUE(std::fabs(mup(6) - mu(6)) > 1e-6);                       // the camera moved
    CHECK_NEAR(Sp(0, 0) * Sp(0, 0), 1.0 + timestep * 0.25, 1e-13);
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) CHECK_TRUE(Sp(i, j) == 0.0);
    for (int j = 12; j < n; ++j) CHECK_NEAR(std::fabs(Sp(j, j)), j + 1.0, 1e-13);   // landmarks: no process noise
}

// removeBadLandmarks (src/SLAM.cpp:796-820) and the descriptor matcher (src/SLAM.cpp:355).
// test/src/pixelDistance.cpp: landmarks at (0,0) and (100,0)
static void test_pixel_distance_and_selection()
{
    std::vector<KeyPoint> lms(2);
    lms[1].x = 100.0f;
    KeyPoint kp;
    CHECK_TRUE(pixelDistance(lms, kp, 10.0));                 // :9-39
    CHECK_TRUE(pixelDistance(lms, kp, 200.0));                // :41-71
    kp.x = 500.0f; kp.y = 500.0f;
    CHECK_TRUE(!pixelDistance(lms, kp, 200.0));               // :73-105
    // the loop of src/SLAM.cpp:449-472 on five sorted keypoints, empty map, the reference's thresholds
    std::vector<vslam_feature> f = {{300, 300, 5.0f}, {310, 305, 4.0f}, {900, 500, 3.0f}, {960, 540, 2.0f}, {1700, 900, 1.0f}};
    std::vector<int32_t> sel = selectNewLandmarks(f, {}, 1e-4, 250.0, 100.0, 150.0, 1920, 1080, 10);
    CHECK_TRUE(sel.size() == 2 && sel[0] == 0 && sel[1] == 2);   // 1 and 3 are within 150 px of a taken one, 4 is outside the margin
    sel = selectNewLandmarks(f, {}, 1e-4, 250.0, 100.0, 150.0, 1920, 1080, 1);
    CHECK_TRUE(sel.size() == 1 && sel[0] == 0);
}

// initializeNewLandmark / generateOpticalRay (src/SLAM.cpp:876-941): both states grow, the new landmarks are seen at
// their keypoints.
static void test_landmark_initialisation(const std::string &xml)
{
    Scene sc(xml, 2, 21);
    const int n = sc.mu.rows();
    Vector mup = sc.mu, muEKF = sc.mu;
    Matrix Sp = sc.S, SEKF = sc.S;
    std::vector<KeyPoint> kps(3);
    kps[0].x = 400.0f; kps[0].y = 300.0f; kps[1].x = 1500.25f; kps[1].y = 800.5f; kps[2].x = 960.0f; kps[2].y = 540.0f;
    initializeNewLandmarks(mup, Sp, muEKF, SEKF, kps, sc.param.camera_param, 8.0, 0.5);
    CHECK_TRUE(mup.rows() == n + 9 && muEKF.rows() == n + 9 && Sp.rows() == n + 9 && Sp.cols() == n + 9 && SEKF.rows() == n + 9);
    for (int i = 0; i < n; ++i) {
        CHECK_NEAR(mup(i), sc.mu(i), 0.0);
        CHECK_NEAR(muEKF(i), sc.mu(i), 0.0);
        for (int j = 0; j < n; ++j) CHECK_NEAR(Sp(i, j), sc.S(i, j), 0.0);
    }
    for (int i = n; i < n + 9; ++i) {
        CHECK_NEAR(muEKF(i), 0.0, 0.0);
        for (int j = 0; j < n + 9; ++j) {
            CHECK_NEAR(Sp(i, j), i == j ? 0.5 : 0.0, 0.0);
            CHECK_NEAR(SEKF(j, i), i == j ? 0.5 : 0.0, 0.0);
        }
    }
    MeasurementPointLandmarkSingle hs;
    Vector y;
    for (int t = 0; t < 3; ++t) {
        CHECK_TRUE(hs(2 + t, mup, sc.param.camera_param, y) == 0);
        CHECK_NEAR(y(0), kps[(size_t)t].x, 0.01);
        CHECK_NEAR(y(1), kps[(size_t)t].y, 0.01);
    }
    Vector ray = generateOpticalRay(sc.param.camera_param, 960.0, 540.0, 3.0);
    CHECK_NEAR(ray(2), 3.0, 0.0);
}

static void test_landmark_bookkeeping()
{
    const int n = 21;
    Vector mu(n);
    Matrix S(n, n);
    std::mt19937 gen(11);
    std::normal_distribution<double> nd;
    for (int j = 0; j < n; ++j) { mu(j) = j; for (int i = 0; i <= j; ++i) S(i, j) = nd(gen); }
    std::vector<Landmark> lms(3);
    Vector mu2 = mu; Matrix S2 = S;
    removeBadLandmarks(mu2, S2, lms, 1);
    CHECK_TRUE(lms.size() == 2 && mu2.rows() == n - 3 && S2.rows() == n - 3 && S2.cols() == n - 3);
    for (int i = 0; i < n - 3; ++i) CHECK_NEAR(mu2(i), i < 15 ? i : i + 3, 0.0);
    // S2' S2 equals the marginal of S' S
    for (int a = 0; a < n - 3; a += 4)
        for (int b = 0; b < n - 3; b += 3) {
            const int ia = a < 15 ? a : a + 3, ib = b < 15 ? b : b + 3;
            double got = 0.0, want = 0.0;
            for (int k = 0; k < n - 3; ++k) got += S2(k, a) * S2(k, b);
            for (int k = 0; k < n; ++k) want += S(k, ia) * S(k, ib);
            CHECK_NEAR(got, want, 1e-12);
        }
    std::vector<uint8_t> train(64 * 32), query(2 * 32);
    for (size_t i = 0; i < train.size(); ++i) train[i] = (uint8_t)(gen() & 0xFF);
    for (int b = 0; b < 32; ++b) { query[(size_t)b] = train[(size_t)(40 * 32 + b)]; query[(size_t)(32 + b)] = train[(size_t)(7 * 32 + b)]; }
    query[0] ^= 0x11;                                  // two bits away from train row 40
    std::vector<Match> matches;
    matchHamming(query.data(), 2, train.data(), 64, 32, matches);
    CHECK_TRUE(matches.size() == 2);
    CHECK_TRUE(matches[0].trainIdx == 40 && matches[0].distance == 2.0f && matches[0].queryIdx == 0);
    CHECK_TRUE(matches[1].trainIdx == 7 && matches[1].distance == 0.0f);
}

static void test_detect_corners()
{
    const int w = 320, h = 240;
    std::vector<uint8_t> img((size_t)w * h, 30);
    for (int y = 60; y < 180; ++y) for (int x = 80; x < 240; ++x) img[(size_t)y * w + x] = 220;
    std::vector<Feature> kp;
    detectCorners(img.data(), w, h, w, 100, 1e-4f, VSLAM_DETECTOR_HARRIS, kp);
    CHECK_TRUE(kp.size() >= 4 && kp.size() <= 100);
    for (size_t i = 1; i < kp.size(); ++i) CHECK_TRUE(kp[i - 1].score >= kp[i].score);
    int near = 0;
    const int cx[4] = {80, 239, 80, 239}, cy[4] = {60, 60, 179, 179};
    for (int c = 0; c < 4; ++c)
        for (const Feature &f : kp)
            if (std::abs(f.x - cx[c]) <= 4 && std::abs(f.y - cy[c]) <= 4) { ++near; break; }
    CHECK_TRUE(near == 4);
}

static void test_detect_and_compute_orb()
{
    // a field of bright blocks on a dark ground: corners well inside the 75-pixel edge
    const int w = 640, h = 480;
    std::vector<uint8_t> img((size_t)w * h, 40);
    for (int by = 0; by < 5; ++by)
        for (int bx = 0; bx < 7; ++bx)
            for (int y = 0; y < 30; ++y)
                for (int x = 0; x < 30; ++x) img[(size_t)(100 + 60 * by + y) * w + 100 + 60 * bx + x] = (uint8_t)(150 + 10 * ((bx + by) % 5));
    // a little noise: on a noiseless pattern neighbouring corner pixels tie in FAST score and suppress one another
    uint32_t st = 12345u;
    for (uint8_t &px : img) { st = st * 1664525u + 1013904223u; px = (uint8_t)(px + (int)((st >> 24) % 7u) - 3); }
    std::vector<OrbKeypoint> kp;
    std::vector<float> ang;
    std::vector<uint8_t> desc;
    detectAndComputeORB(img.data(), w, h, w, 500, kp, ang, desc);
    CHECK_TRUE(kp.size() >= 100 && ang.size() == kp.size() && desc.size() == 32 * kp.size());
    int level0 = 0, any_bits = 0;
    for (size_t i = 0; i < kp.size(); ++i) {
        CHECK_TRUE(ang[i] >= 0.0f && ang[i] <= 360.0f);
        CHECK_TRUE(kp[i].octave >= 0 && kp[i].octave < 8 && kp[i].x >= 75.0f && kp[i].x < (float)(w - 75));
        if (i > 0 && kp[i].octave == kp[i - 1].octave) CHECK_TRUE(kp[i - 1].response >= kp[i].response);
        level0 += kp[i].octave == 0;
        for (int b = 0; b < 32; ++b) any_bits |= desc[32 * i + b];
    }
    if (!(level0 >= 10 && any_bits != 0)) std::printf("orb: %zu keypoints, %d on level 0, descriptor bits or-ed %d\n", kp.size(), level0, any_bits);
    CHECK_TRUE(level0 >= 10 && any_bits != 0);
    // a descriptor matches itself at distance 0 and the matcher finds it
    std::vector<Match> matches;
    matchHamming(desc.data(), (int)kp.size(), desc.data(), (int)kp.size(), 32, matches);
    CHECK_TRUE(matches.size() == kp.size());
    for (const Match &mt : matches) CHECK_TRUE(mt.distance == 0.0f);
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <camera.xml>\n", argv[0]); return 2; }
    try {
        test_log_gaussian();
        test_affine_transform();
        test_condition_on_marginal();
        test_measurement_model(argv[1]);
        test_point_log_likelihood(argv[1]);
        test_measurement_update(argv[1]);
        test_detect_corners();
        test_detect_and_compute_orb();
        test_data_association(argv[1]);
        test_time_update();
        test_landmark_bookkeeping();
        test_pixel_distance_and_selection();
        test_landmark_initialisation(argv[1]);
        // Contract violations surface as exceptions carrying the library's message, not aborts.
        bool threw = false;
        try {
            Vector mu{1, 2, 3}, y{1, 2}, mc;
            Matrix bad(2, 3), Sc;          // a joint factor must have at least ny + nx rows
            conditionGaussianOnMarginal(mu, bad, y, mc, Sc);
        }
        catch (const std::runtime_error &) { threw = true; }
        CHECK_TRUE(threw);
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 99;
    }
    std::printf("%d checks, %d failures\n", checks, failures);
    return failures > 98 ? 98 : failures;
}
