// lbfgsb.hpp -- a small, dependency-free box-constrained limited-memory quasi-Newton minimiser for the host-side learners
// of this repo (moihgp_online.hpp, moihgp_regression.hpp).
//
// The reference drives its learners with LBFGS++ (reference moihgp/include/LBFGSpp/, a third-party Eigen library vendored
// there; BASELINE.json north_star: "LBFGS++ ... stay on host").  LBFGS++ needs Eigen3, which this image does not have, so the
// Eigen-free learners here carry their own optimiser with the same call surface:
//     LBFGSBParam   { m, epsilon, epsilon_rel, past, delta, max_iterations, max_submin, max_linesearch, min_step, max_step,
//                     ftol, wolfe }                                   (field names and defaults of LBFGSpp/Param.h:228-345)
//     LBFGSBSolver  { int minimize(f, x, fx, lb, ub);  BFGSMat getBFGSMat(); }       (LBFGSpp/LBFGSB.h:115, :243)
//     BFGSMat       { get_m(); apply_Hv(v, a, res) }  two-loop recursion  a H v      (LBFGSpp/BFGSMat.h:151-178, :491)
// It is NOT a restatement of LBFGS++'s algorithm (generalised Cauchy point + subspace minimisation + More-Thuente): it is a
// projected L-BFGS method -- gradient projection to fix the active set, an L-BFGS direction on the free variables, and a
// projected backtracking (Armijo) search capped at `max_step` -- which solves the same problem with different iterates.
// Users who have Eigen can keep LBFGS++ itself: include/moihgp_cxx/compat/ lets the reference's own learner headers compile
// against this library's moihgp::MOIHGP<SS>.  Pure host code: no GPU, no libmoihgp.so dependency.
#ifndef MOIHGP_CXX_LBFGSB_HPP_
#define MOIHGP_CXX_LBFGSB_HPP_

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <stdexcept>
#include <vector>

namespace moihgp {
namespace opt {

typedef std::vector<double> Vector;

inline double dot(const Vector& a, const Vector& b) {
    double s = 0.0;
    for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
    return s;
}
inline double norm(const Vector& a) { return std::sqrt(dot(a, a)); }

struct LBFGSBParam {            // LBFGSpp/Param.h:228-345 (same names, same defaults)
    int m = 6;
    double epsilon = 1e-5;
    double epsilon_rel = 1e-5;
    int past = 1;
    double delta = 1e-10;
    int max_iterations = 0;
    int max_submin = 10;        // unused here (no subspace minimisation); kept so that reference-side settings carry over
    int max_linesearch = 20;
    double min_step = 1e-20;
    double max_step = 1e+20;
    double ftol = 1e-4;
    double wolfe = 0.9;         // unused by the Armijo search; kept for the same reason
    void check_param() const {
        if (m <= 0) throw std::invalid_argument("'m' must be positive");
        if (epsilon < 0 || epsilon_rel < 0) throw std::invalid_argument("'epsilon' must be non-negative");
        if (past < 0 || delta < 0) throw std::invalid_argument("'past' / 'delta' must be non-negative");
        if (max_iterations < 0) throw std::invalid_argument("'max_iterations' must be non-negative");
        if (max_linesearch <= 0) throw std::invalid_argument("'max_linesearch' must be positive");
        if (min_step < 0 || max_step < min_step) throw std::invalid_argument("bad step bounds");
        if (ftol <= 0 || ftol >= 0.5) throw std::invalid_argument("'ftol' must satisfy 0 < ftol < 0.5");
    }
};

// Limited-memory BFGS approximation from the last m correction pairs, cyclic storage as LBFGSpp/BFGSMat.h:37-50.
class BFGSMat {
public:
    BFGSMat() : m_m(0), m_theta(1.0), m_ncorr(0), m_ptr(0) {}
    void reset(int /*n*/, int m) {
        m_m = m; m_theta = 1.0; m_ncorr = 0; m_ptr = m;
        m_s.assign(m, Vector()); m_y.assign(m, Vector()); m_ys.assign(m, 0.0); m_alpha.assign(m, 0.0);
    }
    void add_correction(const Vector& s, const Vector& y) {           // BFGSMat.h:81-99
        const int loc = m_ptr % m_m;
        m_s[loc] = s; m_y[loc] = y;
        const double ys = dot(s, y);
        m_ys[loc] = ys;
        m_theta = dot(y, y) / ys;
        if (m_ncorr < m_m) m_ncorr++;
        m_ptr = loc + 1;
    }
    // res = a * H * v, H the inverse-Hessian approximation with H0 = I / theta (two-loop recursion, BFGSMat.h:151-178)
    void apply_Hv(const Vector& v, const double& a, Vector& res) {
        res.resize(v.size());
        for (size_t i = 0; i < v.size(); i++) res[i] = a * v[i];
        int j = m_ptr % m_m;
        for (int i = 0; i < m_ncorr; i++) {
            j = (j + m_m - 1) % m_m;
            m_alpha[j] = dot(m_s[j], res) / m_ys[j];
            for (size_t q = 0; q < res.size(); q++) res[q] -= m_alpha[j] * m_y[j][q];
        }
        for (size_t q = 0; q < res.size(); q++) res[q] /= m_theta;
        for (int i = 0; i < m_ncorr; i++) {
            const double beta = dot(m_y[j], res) / m_ys[j];
            for (size_t q = 0; q < res.size(); q++) res[q] += (m_alpha[j] - beta) * m_s[j][q];
            j = (j + 1) % m_m;
        }
    }
    // the same recursion with every pair restricted to the coordinates flagged in `free_var` (the direction on the face)
    void apply_Hv_free(const Vector& v, const std::vector<char>& free_var, Vector& res) {
        res = v;
        for (size_t q = 0; q < res.size(); q++) if (!free_var[q]) res[q] = 0.0;
        std::vector<double> alpha(m_m, 0.0), ysf(m_m, 0.0);
        std::vector<char> use(m_m, 0);
        double theta = 1.0;
        bool have_theta = false;
        int j = m_ptr % m_m;
        for (int i = 0; i < m_ncorr; i++) {
            j = (j + m_m - 1) % m_m;
            double ys = 0.0, yy = 0.0;
            for (size_t q = 0; q < res.size(); q++) if (free_var[q]) { ys += m_s[j][q] * m_y[j][q]; yy += m_y[j][q] * m_y[j][q]; }
            ysf[j] = ys;
            use[j] = ys > std::numeric_limits<double>::epsilon() * yy;
            if (!use[j]) continue;
            if (!have_theta) { theta = yy / ys; have_theta = true; }  // most recent usable pair
            double a = 0.0;
            for (size_t q = 0; q < res.size(); q++) if (free_var[q]) a += m_s[j][q] * res[q];
            alpha[j] = a / ys;
            for (size_t q = 0; q < res.size(); q++) if (free_var[q]) res[q] -= alpha[j] * m_y[j][q];
        }
        for (size_t q = 0; q < res.size(); q++) res[q] /= theta;
        for (int i = 0; i < m_ncorr; i++) {
            if (use[j]) {
                double b = 0.0;
                for (size_t q = 0; q < res.size(); q++) if (free_var[q]) b += m_y[j][q] * res[q];
                b /= ysf[j];
                for (size_t q = 0; q < res.size(); q++) if (free_var[q]) res[q] += (alpha[j] - b) * m_s[j][q];
            }
            j = (j + 1) % m_m;
        }
    }
    int get_m() { return m_m; }                                        // BFGSMat.h:491
    int num_corrections() const { return m_ncorr; }
    double theta() const { return m_theta; }

private:
    int m_m;
    double m_theta;
    std::vector<Vector> m_s, m_y;
    Vector m_ys, m_alpha;
    int m_ncorr, m_ptr;
};

class LBFGSBSolver {
public:
    explicit LBFGSBSolver(const LBFGSBParam& param) : m_param(param) { m_param.check_param(); }

    // ||P(x - g) - x||_inf, the projected-gradient norm of LBFGSB.h:64-67
    static double proj_grad_norm(const Vector& x, const Vector& g, const Vector& lb, const Vector& ub) {
        double r = 0.0;
        for (size_t i = 0; i < x.size(); i++) r = std::max(r, std::fabs(std::min(std::max(x[i] - g[i], lb[i]), ub[i]) - x[i]));
        return r;
    }

    // Minimise f over lb <= x <= ub.  `f(x, grad)` returns the value and overwrites grad.  Returns the iteration count.
    // Same stopping rules as LBFGSB.h:133-205: projected gradient (epsilon, epsilon_rel), `past`/`delta`, max_iterations.
    template <typename Foo>
    int minimize(Foo& f, Vector& x, double& fx, const Vector& lb, const Vector& ub) {
        const size_t n = x.size();
        if (lb.size() != n || ub.size() != n) throw std::invalid_argument("'lb' and 'ub' must have the same size as 'x'");
        for (size_t i = 0; i < n; i++) x[i] = std::min(std::max(x[i], lb[i]), ub[i]);     // force_bounds, LBFGSB.h:56-59
        m_bfgs.reset((int)n, m_param.m);
        Vector grad(n), xp(n), gradp(n), drt(n), xt(n), s(n), y(n);
        std::vector<char> free_var(n);
        std::vector<double> fpast(m_param.past > 0 ? m_param.past : 0);
        fx = f(x, grad);
        double pg = proj_grad_norm(x, grad, lb, ub);
        if (m_param.past > 0) fpast[0] = fx;
        if (pg <= m_param.epsilon || pg <= m_param.epsilon_rel * norm(x)) return 1;
        const double eps = std::numeric_limits<double>::epsilon();
        int k = 1;
        for (;;) {
            xp = x; gradp = grad;
            // active set: at a bound with the gradient pointing out of the box
            for (size_t i = 0; i < n; i++)
                free_var[i] = !((x[i] <= lb[i] && grad[i] > 0.0) || (x[i] >= ub[i] && grad[i] < 0.0) || lb[i] == ub[i]);
            m_bfgs.apply_Hv_free(grad, free_var, drt);
            double slope = 0.0;
            for (size_t i = 0; i < n; i++) { drt[i] = -drt[i]; slope += grad[i] * drt[i]; }
            if (!(slope < 0.0)) {                                      // not a descent direction: steepest descent on the face
                slope = 0.0;
                for (size_t i = 0; i < n; i++) { drt[i] = free_var[i] ? -grad[i] : 0.0; slope += grad[i] * drt[i]; }
            }
            const double dn = norm(drt);
            if (dn == 0.0) return k;
            // projected backtracking search: x(t) = P(xp + t drt), t <= max_step / |drt| (max_step bounds the move, as the
            // reference's learner uses it: moihgp_online.h:156), Armijo on the actual displacement
            double step = std::min(1.0, m_param.max_step / dn);
            if (k == 1 && m_bfgs.num_corrections() == 0) step = std::min(step, 1.0 / dn);
            bool ok = false;
            double ft = fx;
            for (int ls = 0; ls < m_param.max_linesearch; ls++) {
                double dec = 0.0;
                for (size_t i = 0; i < n; i++) { xt[i] = std::min(std::max(xp[i] + step * drt[i], lb[i]), ub[i]); dec += gradp[i] * (xt[i] - xp[i]); }
                ft = f(xt, grad);
                if (std::isfinite(ft) && ft <= fx + m_param.ftol * dec) { ok = true; break; }
                step *= 0.5;
                if (step < m_param.min_step) break;
            }
            if (!ok) { fx = f(xp, grad); x = xp; return k; }          // no acceptable point: stay (and leave f's state at x)
            x = xt; fx = ft;
            pg = proj_grad_norm(x, grad, lb, ub);
            if (pg <= m_param.epsilon || pg <= m_param.epsilon_rel * norm(x)) return k;
            if (m_param.past > 0) {
                const double fxd = fpast[k % m_param.past];
                if (k >= m_param.past && std::fabs(fxd - fx) <= m_param.delta * std::max(std::max(std::fabs(fx), std::fabs(fxd)), 1.0)) return k;
                fpast[k % m_param.past] = fx;
            }
            if (m_param.max_iterations != 0 && k >= m_param.max_iterations) return k;
            for (size_t i = 0; i < n; i++) { s[i] = x[i] - xp[i]; y[i] = grad[i] - gradp[i]; }
            if (dot(s, y) > eps * dot(y, y)) m_bfgs.add_correction(s, y);        // LBFGSB.h:212-213
            k++;
        }
    }

    BFGSMat getBFGSMat() { return m_bfgs; }                           // LBFGSB.h:243

private:
    LBFGSBParam m_param;
    BFGSMat m_bfgs;
};

}  // namespace opt
}  // namespace moihgp

#endif
