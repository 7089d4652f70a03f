// Driver of the iterated-EKF measurement update (src/gaussian.hpp:469-532).
// Default route (VSLAM_TRS_EIG): the trust-region Newton loop and the Moré-Sorensen
// subproblem run on the host exactly as in the reference (src/fmin.hpp:19-107,
// src/fmin.cpp:40-131) and drive the device kernels for the cost, gradient, Hessian
// and the posterior square-root QR.  Opt-in route (VSLAM_TRS_CHOLESKY): the whole
// iteration runs on the device from Cholesky factors (newton.hip); an update that meets
// an indefinite Hessian is redone on the default route.
//
// The symmetric eigendecomposition the loop needs (Eigen::SelfAdjointEigenSolver
// in the reference, src/fmin.hpp:42-44,99-101) is done here by Householder
// tridiagonalisation + implicit-shift QL on the host; eigenvalues ascending,
// lower triangle read, like Eigen.
#include "linalg.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>
#include <cstdlib>

namespace vslam {


// ---- host symmetric eigensolver (the reference's Eigen::SelfAdjointEigenSolver, src/fmin.hpp:42-44) ----
// Householder tridiagonalisation (the symmetric matrix-vector product of each step as one pass over
// the contiguous rows of the lower triangle), accumulation of the transformation (row-wise, so every
// inner loop runs over contiguous memory) and implicit-shift QL.  The QL iteration itself only touches the two
// diagonals; its plane rotations are recorded and then applied to the vectors, which are kept
// transposed (row i = vector i) so that a rotation runs over two contiguous rows.
// (Tried on the 16 cores a GPU box gives one process: a lock-step thread team for the first two phases
// loses to one thread because of the barrier per Householder step, and splitting only the rotation
// pass by columns gains 15 %: not worth threads inside the library.)
namespace {
struct Rot { int i; double c, s; };

// The solver is compiled twice from sym_eig.inc: for the baseline x86-64 instruction set and for AVX2 + FMA
// (its loops are long contiguous axpy / dot / rotation passes over doubles); the variant is chosen once
// from the CPU the process runs on.
#define EIG_FN(name) name##_base
#define EIG_ATTR
#include "sym_eig.inc"
#undef EIG_FN
#undef EIG_ATTR
#if defined(__x86_64__)
#define EIG_FN(name) name##_avx2
#define EIG_ATTR __attribute__((target("avx2,fma")))
#include "sym_eig.inc"
#undef EIG_FN
#undef EIG_ATTR
#endif
}  // namespace

// Eigen-decomposition of the symmetric matrix whose lower triangle is in H
// (column-major, n x n).  Q column-major eigenvectors, v ascending eigenvalues.
void sym_eig_host(const double *H, int n, double *Q, double *v)
{
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (wide) { sym_eig_avx2(H, n, Q, v); return; }
#endif
    sym_eig_base(H, n, Q, v);
}

static double norm2(const std::vector<double> &a) { double s = 0; for (double t : a) s += t * t; return std::sqrt(s); }

// minimise 0.5 p'Hp + g'p subject to |p| <= D with H = Q diag(v) Q'   (src/fmin.cpp:40-131)
int trs_host(const double *Q, const double *v, const double *g, int n, double D, std::vector<double> &p)
{
    const double eps = std::numeric_limits<double>::epsilon(), sqrteps = std::sqrt(eps);
    const int maxIterations = 20;
    p.assign(n, 0.0);
    const double l1 = v[0];
    std::vector<double> a(n), vlam(n), t(n);
    for (int j = 0; j < n; ++j) { double s = 0; const double *q = Q + (size_t)j * n; for (int i = 0; i < n; ++i) s += q[i] * g[i]; a[j] = s; }
    double lam = l1 < 0 ? 1.01 * std::fabs(l1) : 0.0;
    for (int i = 0; i < n; ++i) vlam[i] = v[i] + lam;
    auto p_from = [&]() {
        std::fill(p.begin(), p.end(), 0.0);
        for (int c = 0; c < n; ++c) { const double w = a[c] / vlam[c]; const double *q = Q + (size_t)c * n; for (int r = 0; r < n; ++r) p[r] -= q[r] * w; }
    };
    p_from();
    const bool hard = std::fabs(a[0]) < eps && l1 < 0;
    const double pn = norm2(p);
    if (l1 < 0 || pn > D || std::fabs(lam * (pn - D)) > sqrteps) {
        if (hard) {
            std::vector<double> pvec(n, 0.0);
            double ssq = 0.0;
            for (int i = 0; i < n; ++i)
                if (std::fabs(v[i] - l1) > sqrteps) {
                    const double scv = a[i] / (v[i] - l1);
                    ssq += scv * scv;
                    const double *q = Q + (size_t)i * n;
                    for (int r = 0; r < n; ++r) pvec[r] += q[r] * scv;
                }
            const double tt = std::sqrt(D * D - ssq);
            for (int r = 0; r < n; ++r) p[r] = tt * Q[r] - pvec[r];
            double pg = 0, quad = 0;
            for (int r = 0; r < n; ++r) pg += p[r] * g[r];
            for (int j = 0; j < n; ++j) { double s = 0; const double *q = Q + (size_t)j * n; for (int i = 0; i < n; ++i) s += q[i] * p[i]; quad += s * v[j] * s; }
            if (pg + 0.5 * quad > 0.0)
                for (int r = 0; r < n; ++r) p[r] = -tt * Q[r] - pvec[r];
        } else {
            int k;
            for (k = 0; k < maxIterations; ++k) {
                double ppn2 = 0, dpp = 0;
                for (int i = 0; i < n; ++i) { const double pp = -a[i] / vlam[i], dp = a[i] / (vlam[i] * vlam[i]); ppn2 += pp * pp; dpp += dp * pp; }
                const double ppn = std::sqrt(ppn2), ff = 1 / D - 1 / ppn, gg = dpp / (ppn * ppn * ppn);
                const double lo = std::max(0.0, -l1) + sqrteps * std::max(0.0, -l1);
                lam = std::max(lo, lam - ff / gg);
                for (int i = 0; i < n; ++i) vlam[i] = v[i] + lam;
                if (std::fabs(ff) < sqrteps) break;
            }
            p_from();
            if (k >= maxIterations) return 1;
        }
    }
    return 0;
}

}  // namespace vslam

using namespace vslam;

extern "C" vslam_status vslam_ctx_set_trs_solver(vslam_ctx *ctx, int solver)
{
    if (!ctx) return VSLAM_ERR_ARG;
    if (solver != VSLAM_TRS_CHOLESKY && solver != VSLAM_TRS_EIG) return fail(ctx, VSLAM_ERR_ARG, "set_trs_solver: unknown solver %d", solver);
    ctx->trs_solver = solver;
    return VSLAM_OK;
}

// Test hook (not part of the public header): the host eigensolver of the eigenvalue route.
extern "C" void vslam_debug_sym_eig(const double *H, int n, double *Q, double *v) { sym_eig_host(H, n, Q, v); }
// Test hook: the baseline-instruction-set variant, whatever the CPU supports.
extern "C" void vslam_debug_sym_eig_base(const double *H, int n, double *Q, double *v) { sym_eig_base(H, n, Q, v); }
// Test hook: the trust-region subproblem in the eigenbasis (src/fmin.cpp:40-131); returns trs's code.
extern "C" int vslam_debug_trs(const double *Q, const double *v, const double *g, int n, double D, double *p)
{
    std::vector<double> pv;
    const int ret = trs_host(Q, v, g, n, D, pv);
    for (int i = 0; i < n; ++i) p[i] = pv[(size_t)i];
    return ret;
}

extern "C" int vslam_ctx_last_trs_route(const vslam_ctx *ctx) { return ctx ? ctx->trs_last : -1; }

extern "C" vslam_status vslam_ctx_last_update_profile(vslam_ctx *ctx, double *out8)
{
    if (!ctx || !out8) return VSLAM_ERR_ARG;
    if (ctx->trs_last != VSLAM_TRS_CHOLESKY) return fail(ctx, VSLAM_ERR_ARG, "last_update_profile: the last measurement update did not run on the device route");
    for (int i = 0; i < 8; ++i) out8[i] = ctx->prof[i];
    return VSLAM_OK;
}

extern "C" vslam_status vslam_iekf_post_sqrt(vslam_ctx *ctx, const double *Q, const double *v, int n, double *S_out);

extern "C" vslam_status vslam_measurement_update_iekf(vslam_ctx *ctx, const vslam_camera *cam,
                                                      const double *mu, const double *S, int n, int ldS,
                                                      const double *y, const uint8_t *visible, double sigma,
                                                      double *mu_post, double *S_post, int ldSp,
                                                      int32_t *fmin_code, int32_t *iters, int32_t *cost_evals)
{
    if (!ctx) return VSLAM_ERR_ARG;
    // n = 12: the frames before the first landmark; the reference runs the update on every frame (src/SLAM.cpp:489-495)
    // with an empty likelihood (src/model.cpp:392), which returns the prior
    if (n < 12 || (n - 12) % 3 != 0 || ldS < n || ldSp < n) return fail(ctx, VSLAM_ERR_SHAPE, "measurement_update_iekf: n=%d is not 12+3L or ld < n", n);
    if (!mu || !S || !mu_post || !S_post || (n > 12 && (!y || !visible))) return fail(ctx, VSLAM_ERR_ARG, "measurement_update_iekf: null pointer");
    if (!(sigma > 0.0)) return fail(ctx, VSLAM_ERR_ARG, "measurement_update_iekf: sigma must be positive");
    CamK k;
    VSLAM_TRY(resolve_camera(ctx, cam, &k));
    VSLAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool try_device = ctx->trs_solver == VSLAM_TRS_CHOLESKY;
    VSLAM_TRY(ws_reserve(ctx, cost_ws_bytes(n) + (try_device ? newton_ws_bytes(n) : 0)));
    CostDevice c;
    cost_carve(ctx->ws, n, &c);
    const size_t nn = (size_t)n * n;
    VSLAM_TRY(h2d(ctx, c.mu, mu, n * 8));
    // a large factor crosses PCIe as its upper triangle only, in column blocks packed by several host threads
    if (n >= PIPELINED_COPY_MIN_N) VSLAM_TRY(h2d_upper_pipelined(ctx, c.S, (size_t)n, S, (size_t)ldS, n));
    else VSLAM_TRY(h2d_2d(ctx, c.S, (size_t)n * 8, S, (size_t)ldS * 8, (size_t)n * 8, n));
    VSLAM_TRY(h2d(ctx, c.y, y, 2 * c.L * 8));
    VSLAM_TRY(h2d(ctx, c.vis, visible, c.L));

    if (try_device) {
        int code = 1, it = 0, ev = 0, status = 0;
        VSLAM_TRY(iekf_newton_device(ctx, k, sigma, &c, static_cast<char *>(ctx->ws) + cost_ws_bytes(n), mu_post, S_post, ldSp,
                                     &code, &it, &ev, &status));
        if (status == 0) {
            ctx->trs_last = VSLAM_TRS_CHOLESKY;
            if (fmin_code) *fmin_code = code;
            if (iters) *iters = it;
            if (cost_evals) *cost_evals = ev;
            return VSLAM_OK;
        }
        c.prior_ready = false;      // the device route used the prior-Hessian buffers as scratch
    }

    ctx->trs_last = VSLAM_TRS_EIG;
    int evals = 0;
    auto cost = [&](const std::vector<double> &x, std::vector<double> &g, std::vector<double> &H, double *f) -> vslam_status {
        VSLAM_TRY(h2d(ctx, c.x, x.data(), n * 8));
        VSLAM_TRY(cost_eval_dev(ctx, k, sigma, &c));
        VSLAM_TRY(d2h(ctx, f, c.vals + 2, 8));
        VSLAM_TRY(d2h(ctx, g.data(), c.g, n * 8));
        VSLAM_TRY(d2h(ctx, H.data(), c.H, nn * 8));
        VSLAM_TRY(stream_sync(ctx));
        ++evals;
        return VSLAM_OK;
    };

    // fminNewtonTrustEig, src/fmin.hpp:19-107
    std::vector<double> x(mu, mu + n), g(n), gn(n), xn(n), p, H(nn), Hn(nn), Q(nn), v(n);
    double f = 0.0, fn = 0.0;
    VSLAM_TRY(cost(x, g, H, &f));                                       // :39
    sym_eig_host(H.data(), n, Q.data(), v.data());                      // :42-44
    double Delta = 1e0;
    int ret = 1, it;
    for (it = 0; it < 50; ++it) {
        trs_host(Q.data(), v.data(), g.data(), n, Delta, p);            // :52
        double pg = 0;
        for (int i = 0; i < n; ++i) pg += p[i] * g[i];
        const double LambdaSq = -pg;
        if (std::fabs(LambdaSq) < 2 * std::numeric_limits<double>::epsilon()) { ret = 0; break; }   // :61-67
        for (int i = 0; i < n; ++i) xn[i] = x[i] + p[i];
        VSLAM_TRY(cost(xn, gn, Hn, &fn));                               // :72
        double pHp = 0;
        for (int cidx = 0; cidx < n; ++cidx) {
            const double *col = H.data() + (size_t)cidx * n;
            double sdot = 0;
            for (int r = 0; r < n; ++r) sdot += p[r] * col[r];
            pHp += sdot * p[cidx];
        }
        const double dm = -pg - 0.5 * pHp, rho = (f - fn) / dm, pn = norm2(p);
        if (rho < 0.1) Delta = 0.25 * pn;
        else if (rho > 0.75 && pn > 0.8 * Delta) Delta = 2.0 * Delta;
        if (rho >= 0.001) {
            x = xn; f = fn; g = gn; H.swap(Hn);
            sym_eig_host(H.data(), n, Q.data(), v.data());
        }
    }
    for (int i = 0; i < n; ++i) mu_post[i] = x[i];
    if (fmin_code) *fmin_code = ret;
    if (iters) *iters = it;
    if (cost_evals) *cost_evals = evals;
    // posterior square-root covariance, src/gaussian.hpp:513-516
    std::vector<double> Sp(nn);
    VSLAM_TRY(vslam_iekf_post_sqrt(ctx, Q.data(), v.data(), n, Sp.data()));
    for (int cidx = 0; cidx < n; ++cidx)
        for (int r = 0; r < n; ++r) S_post[(size_t)cidx * ldSp + r] = Sp[(size_t)cidx * n + r];
    return VSLAM_OK;
}
