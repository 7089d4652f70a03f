// Host-side search for the multiplier of one trust-region subproblem
//     minimise g'p + p'Hp/2  subject to |p| <= Delta            (src/fmin.cpp:40-131)
// when all that is available per trial multiplier lam is a Cholesky factorisation H + lam I = U'U on the device (or
// its failure) and a few triangular solves.  The reference works in the eigenbasis of H: it starts at lam = 0 (or at
// 1.01 |l1| when the smallest eigenvalue l1 is negative), takes Newton steps on 1/Delta - 1/|p(lam)| with the lower
// clamp max(0, -l1), stops once |1/Delta - 1/|p|| < sqrt(eps) and evaluates p at the multiplier one step further.
// The subproblem has one solution on the boundary (outside the "hard case"), so any safeguarded iteration that ends
// with the same test and the same final step arrives at the reference's p to rounding; what differs is the number
// of factorisations on the way.  Two things keep that number small:
//   * with p = -(H + lam I)^-1 g the moments m_k = p'(H + lam I)^-(k-2) p, k = 2..5, are the squared norms of
//     p, U^-T p, U^-1 U^-T p, U^-T U^-1 U^-T p (one triangular solve each).  They are the first four moments of the
//     measure that writes |p(lam + d)|^2 as an integral of 1 / (1 + d t)^2, so a two-node Gauss rule reproduces
//     |p(lam + d)|^2 far beyond the tangent the Newton step uses; the next multiplier is the root of that model
//     (three to four factorisations per subproblem where the Newton iteration takes six to eight);
//   * an indefinite H shows up as a failed factorisation.  The first non-positive pivot d of H + lam I bounds the
//     shift that repairs the leading block (every later Schur complement grows by at least the shift), so the next
//     trial is lam - d (plus a margin); trials that fail raise `lo`, trials that succeed with |p| < Delta lower `hi`,
//     and model steps that leave (lo, hi) are replaced by its midpoint (More & Sorensen's safeguards).  The hard case
//     (no root above -l1) ends with the bracket collapsing: the caller falls back to the eigenvalue route.
#pragma once

#include <cmath>
#include <limits>

namespace vslam {

struct TrsSlot {
    double Delta = 1.0;
    double lam = 0.0;          // multiplier of the factorisation last requested
    double lo = 0.0;           // largest multiplier at which the factorisation failed (0: none failed; lam = 0 then counts as failing only if `indefinite`)
    double hi = std::numeric_limits<double>::infinity();     // smallest multiplier that factored with |p| <= Delta
    double left = -1.0;        // largest multiplier that factored with |p| > Delta (the root lies above it); -1: none yet
    double bump = 0.0;         // extra shift after repeated failures
    bool indefinite = false;   // some factorisation failed
    int steps = 0, fails = 0;
};

enum TrsNext { TRS_DONE = 0, TRS_REFACTOR = 1, TRS_REFACTOR_LAST = 2, TRS_GIVE_UP = 3 };

// Root d of the two-node Gauss model of |p(lam + d)|^2 = Delta^2; false when the moments do not give a proper rule.
inline bool trs_model_step(double m2, double m3, double m4, double m5, double Delta, double *d_out)
{
    if (!(m2 > 0.0 && m3 > 0.0 && m4 > 0.0 && m5 > 0.0) || !std::isfinite(m2 + m3 + m4 + m5)) return false;
    const double s = m3 / m2;                                  // scale of t = 1 / (eigenvalue + lam)
    const double u2 = m4 / (m2 * s * s), u3 = m5 / (m2 * s * s * s);      // scaled moments u0 = u1 = 1
    const double det = 1.0 - u2;                               // <= 0 (Cauchy-Schwarz); 0 for a one-point measure
    if (!(det < -1e-10)) return false;
    const double c1 = (u3 - u2) / det, c0 = (u2 * u2 - u3) / det;          // t^2 + c1 t + c0 orthogonal to 1 and t
    const double disc = c1 * c1 - 4.0 * c0;
    if (!(disc > 0.0)) return false;
    const double rt = std::sqrt(disc), t1 = 0.5 * (-c1 - rt), t2 = 0.5 * (-c1 + rt);
    if (!(t1 > 0.0 && t2 > t1)) return false;
    const double w2 = (1.0 - t1) / (t2 - t1), w1 = 1.0 - w2;
    if (!(w1 > 0.0 && w2 > 0.0)) return false;
    const double target = Delta * Delta / m2;
    auto f = [&](double x) { const double a = 1.0 + x * t1, b = 1.0 + x * t2; return w1 / (a * a) + w2 / (b * b); };
    double xl, xh;                                             // f decreases from +inf at -1/t2 to 0
    if (target < 1.0) { xl = 0.0; xh = 1.0; int guard = 0; while (f(xh) > target && ++guard < 2000) xh *= 4.0; if (guard >= 2000) return false; }
    else { xl = -(1.0 - 1e-12) / t2; xh = 0.0; }
    for (int it = 0; it < 200; ++it) {
        const double xm = 0.5 * (xl + xh);
        if (xm == xl || xm == xh) break;
        if (f(xm) > target) xl = xm; else xh = xm;
    }
    *d_out = 0.5 * (xl + xh) / s;
    return std::isfinite(*d_out);
}

// The same with three nodes from six moments m_0 .. m_5, m_0 = |g|^2 and m_1 = -g'p being two dot products more: the low
// moments carry the part of the gradient that lies along large eigenvalues, which the higher ones hardly see but which
// decides |p| once the multiplier is large (collapsed trust regions): one factorisation fewer per subproblem there.
// |p(lam + d)|^2 = sum_i w_i t_i^2 / (1 + d t_i)^2 over the rule's nodes t_i and weights w_i.
inline bool trs_model_step3(const double (&m)[6], double Delta, double *d_out)
{
    for (double v : m) if (!(v > 0.0) || !std::isfinite(v)) return false;
    const double s = m[1] / m[0];
    double u[6], sp = 1.0;
    for (int j = 0; j < 6; ++j) { u[j] = m[j] / (m[0] * sp); sp *= s; }
    // Cholesky of the 3 x 3 Hankel matrix H[i][j] = u[i+j]: H = L L^T
    double L[3][3] = {{0}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j <= i; ++j) {
            double acc = u[i + j];
            for (int k = 0; k < j; ++k) acc -= L[i][k] * L[j][k];
            if (i == j) { if (!(acc > 1e-14)) return false; L[i][i] = std::sqrt(acc); }
            else L[i][j] = acc / L[j][j];
        }
    // J = L^-1 H1 L^-T with H1[i][j] = u[i+j+1]: symmetric, its eigenvalues are the nodes
    double A[3][3], B[3][3];
    for (int j = 0; j < 3; ++j)                       // A = L^-1 H1 (forward substitution, column by column)
        for (int i = 0; i < 3; ++i) {
            double acc = u[i + j + 1];
            for (int k = 0; k < i; ++k) acc -= L[i][k] * A[k][j];
            A[i][j] = acc / L[i][i];
        }
    for (int i = 0; i < 3; ++i)                       // B = A L^-T: rows of A solved against L
        for (int j = 0; j < 3; ++j) {
            double acc = A[i][j];
            for (int k = 0; k < j; ++k) acc -= B[i][k] * L[j][k];
            B[i][j] = acc / L[j][j];
        }
    double J[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[i][j] = 0.5 * (B[i][j] + B[j][i]);
    for (int sweep = 0; sweep < 30; ++sweep) {        // cyclic Jacobi on the 3 x 3 matrix
        const double off = std::fabs(J[0][1]) + std::fabs(J[0][2]) + std::fabs(J[1][2]);
        if (off < 1e-18 * (std::fabs(J[0][0]) + std::fabs(J[1][1]) + std::fabs(J[2][2]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (J[p][q] == 0.0) continue;
                const double th = (J[q][q] - J[p][p]) / (2.0 * J[p][q]);
                const double tt = (th >= 0.0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double c = 1.0 / std::sqrt(tt * tt + 1.0), sn = tt * c;
                for (int k = 0; k < 3; ++k) { const double a = J[k][p], b = J[k][q]; J[k][p] = c * a - sn * b; J[k][q] = sn * a + c * b; }
                for (int k = 0; k < 3; ++k) { const double a = J[p][k], b = J[q][k]; J[p][k] = c * a - sn * b; J[q][k] = sn * a + c * b; }
            }
    }
    double t[3] = {J[0][0], J[1][1], J[2][2]};
    if (!(t[0] > 0.0 && t[1] > 0.0 && t[2] > 0.0)) return false;
    // weights from the first three moments: sum w = u0, sum w t = u1, sum w t^2 = u2 (Cramer)
    const double d01 = t[1] - t[0], d02 = t[2] - t[0], d12 = t[2] - t[1];
    if (!(std::fabs(d01 * d02 * d12) > 1e-14)) return false;
    double w[3];
    w[0] = (u[2] - (t[1] + t[2]) * u[1] + t[1] * t[2] * u[0]) / (d01 * d02);
    w[1] = -(u[2] - (t[0] + t[2]) * u[1] + t[0] * t[2] * u[0]) / (d01 * d12);
    w[2] = (u[2] - (t[0] + t[1]) * u[1] + t[0] * t[1] * u[0]) / (d02 * d12);
    if (!(w[0] > 0.0 && w[1] > 0.0 && w[2] > 0.0)) return false;
    // nodes and weights are those of the scaled variable t / s with unit total weight: |p(lam + d)|^2 = m0 s^2 f(d s)
    const double tmax = std::fmax(t[0], std::fmax(t[1], t[2]));
    const double target = Delta * Delta / (m[0] * s * s);
    auto f = [&](double x) { double acc = 0.0; for (int i = 0; i < 3; ++i) { const double a = 1.0 + x * t[i]; acc += w[i] * t[i] * t[i] / (a * a); } return acc; };
    double xl, xh;
    if (f(0.0) > target) { xl = 0.0; xh = 1.0; int guard = 0; while (f(xh) > target && ++guard < 2000) xh *= 4.0; if (guard >= 2000) return false; }
    else { xl = -(1.0 - 1e-12) / tmax; xh = 0.0; }
    for (int it = 0; it < 200; ++it) {
        const double xm = 0.5 * (xl + xh);
        if (xm == xl || xm == xh) break;
        if (f(xm) > target) xl = xm; else xh = xm;
    }
    *d_out = 0.5 * (xl + xh) / s;
    return std::isfinite(*d_out);
}

// Stopping rule next to the reference's |ff| < sqrt(eps), ff = 1/Delta - 1/|p| (src/fmin.cpp:118-126): once the trust
// region has shrunk (Delta << 1) that absolute test asks for more digits than ff has, and the reference then simply runs
// its 20 Newton steps, which cost nothing in the eigenbasis; here each step is a factorisation.  When |p| matches Delta
// to 1e-10 relative the step taken from that ff moves p by less than that: the search stops without it.
inline bool trs_settled(double ff, double Delta) { return std::fabs(ff) * Delta < 1e-10; }

// A filter whose last subproblem met an indefinite Hessian starts its next one at the multiplier that one ended with
// instead of at zero (where the factorisation would fail first and be repaired pivot by pivot); if the Hessian has
// become positive definite meanwhile the search comes back down to zero by itself.
inline void trs_warm_start(TrsSlot &s, double lam_prev) { if (lam_prev > 0.0 && std::isfinite(lam_prev)) s.lam = lam_prev; }

// The factorisation at s.lam failed; `pivot` is its first non-positive pivot.
inline TrsNext trs_after_failure(TrsSlot &s, double pivot)
{
    s.indefinite = true;
    if (!(pivot <= 0.0) || ++s.fails > 40) return TRS_GIVE_UP;          // NaN pivot (the matrix itself is not finite), or no end in sight
    if (s.lam > s.lo) s.lo = s.lam;
    const double floor_ = 1e-10 * (1.0 + s.lam);
    s.bump = s.bump > 0.0 ? 4.0 * s.bump : floor_;
    double next = s.lam + std::fmax(-1.01 * pivot, s.bump);
    const double top = std::isfinite(s.hi) ? s.hi : (s.left >= 0.0 ? s.left : std::numeric_limits<double>::infinity());
    if (std::isfinite(top)) {
        if (!(top - s.lo > 1e-13 * top)) return TRS_GIVE_UP;           // bracket collapsed: hard case
        if (next >= top) next = 0.5 * (s.lo + top);
    }
    s.lam = next;
    return TRS_REFACTOR;
}

// The factorisation at s.lam succeeded and gave |p|^2 = m2; interior = this was the plain Newton step (lam = 0, no
// failure before): it is the solution when it lies inside the region.
inline bool trs_interior(const TrsSlot &s, double m2) { return !s.indefinite && s.lam == 0.0 && std::sqrt(m2) <= s.Delta; }

// ... and the three further moments: decides whether the search is over and where the next factorisation goes.
// m0 = |g|^2 and m1 = -g'p refine the model (trs_model_step3); pass 0 to leave them out.
inline TrsNext trs_after_solve(TrsSlot &s, double m2, double m3, double m4, double m5, double m0 = 0.0, double m1 = 0.0)
{
    const double sqrteps = std::sqrt(std::numeric_limits<double>::epsilon());
    const double pn = std::sqrt(m2), D = s.Delta;
    if (!(pn > 0.0) || !std::isfinite(pn)) return TRS_GIVE_UP;
    const double ff = 1.0 / D - 1.0 / pn;
    if (trs_settled(ff, D)) return TRS_DONE;
    if (s.lam == 0.0 && !s.indefinite && pn <= D) return TRS_DONE;      // a search started above zero (trs_warm_start) came back to the interior Newton step
    const bool conv = std::fabs(ff) < sqrteps;
    if (pn > D) { if (s.lam > s.left) s.left = s.lam; }
    else if (s.lam < s.hi) s.hi = s.lam;
    s.bump = 0.0;
    double d = 0.0, next;
    const double m6[6] = {m0, m1, m2, m3, m4, m5};
    if (m0 > 0.0 && m1 > 0.0 && trs_model_step3(m6, D, &d)) next = s.lam + d;
    else if (trs_model_step(m2, m3, m4, m5, D, &d)) next = s.lam + d;
    else next = s.lam - ff / (-m3 / (pn * pn * pn));                    // the reference's Newton step (src/fmin.cpp:121-123)
    // a search that started above zero and finds the root at or below zero has the plain Newton step left to try
    if (!s.indefinite && s.left < 0.0 && s.lo == 0.0 && !(next > 0.0)) { ++s.steps; s.lam = 0.0; return TRS_REFACTOR; }
    // the root lies in (max(left, lo), hi)
    const double lower = std::fmax(s.left >= 0.0 ? s.left : 0.0, s.lo);
    const bool lower_open = s.left >= 0.0 || s.indefinite;              // the bound itself is not a candidate
    if (std::isfinite(s.hi) && !(next < s.hi)) next = 0.5 * (lower + s.hi);
    if (lower_open ? !(next > lower) : !(next >= lower)) next = std::isfinite(s.hi) ? 0.5 * (lower + s.hi) : (lower > 0.0 ? 2.0 * lower : 1.0);
    if (std::isfinite(s.hi) && !(s.hi - lower > 1e-13 * s.hi) && !conv) return s.indefinite ? TRS_GIVE_UP : TRS_DONE;
    if (next == s.lam) return TRS_DONE;
    if (++s.steps > 24) return TRS_DONE;                                // the reference gives up after 20 steps and uses what it has
    s.lam = next;
    return conv ? TRS_REFACTOR_LAST : TRS_REFACTOR;
}

}  // namespace vslam
