// Drives moihgp::opt::LBFGSBSolver (include/moihgp_cxx/lbfgsb.hpp) on small bound-constrained problems; the Python test
// compares the minimisers with SciPy's L-BFGS-B.  Pure host code.
//   stdin:  problem n  then lb[n] ub[n] x0[n]  (problem 0: Rosenbrock, 1: convex quadratic with a fixed tridiagonal matrix)
//   stdout: iterations, f, x[n]
#include <cstdio>
#include <vector>
#include "moihgp_cxx/lbfgsb.hpp"

using moihgp::opt::Vector;

struct Rosenbrock {
    int calls = 0;
    double operator()(const Vector& x, Vector& g) {
        calls++;
        const size_t n = x.size();
        double f = 0.0;
        for (size_t i = 0; i < n; i++) g[i] = 0.0;
        for (size_t i = 0; i + 1 < n; i++) {
            const double a = x[i + 1] - x[i] * x[i], b = 1.0 - x[i];
            f += 100.0 * a * a + b * b;
            g[i] += -400.0 * a * x[i] - 2.0 * b;
            g[i + 1] += 200.0 * a;
        }
        return f;
    }
};
struct Quadratic {            // f = 1/2 x'Ax - b'x,  A = tridiag(-1, 2.5, -1),  b_i = sin(i + 1) * 3
    double operator()(const Vector& x, Vector& g) {
        const size_t n = x.size();
        double f = 0.0;
        for (size_t i = 0; i < n; i++) {
            double ax = 2.5 * x[i] - (i > 0 ? x[i - 1] : 0.0) - (i + 1 < n ? x[i + 1] : 0.0);
            const double b = 3.0 * std::sin((double)(i + 1));
            g[i] = ax - b;
            f += 0.5 * x[i] * ax - b * x[i];
        }
        return f;
    }
};

int main() {
    int prob; size_t n;
    if (scanf("%d %zu", &prob, &n) != 2) return 2;
    Vector lb(n), ub(n), x(n);
    for (auto& v : lb) if (scanf("%lf", &v) != 1) return 2;
    for (auto& v : ub) if (scanf("%lf", &v) != 1) return 2;
    for (auto& v : x) if (scanf("%lf", &v) != 1) return 2;
    moihgp::opt::LBFGSBParam prm;
    prm.m = 10; prm.max_iterations = 2000; prm.epsilon = 1e-9; prm.epsilon_rel = 1e-9; prm.past = 0; prm.max_linesearch = 40;
    moihgp::opt::LBFGSBSolver solver(prm);
    double fx = 0.0;
    int it;
    if (prob == 0) { Rosenbrock f; it = solver.minimize(f, x, fx, lb, ub); }
    else { Quadratic f; it = solver.minimize(f, x, fx, lb, ub); }
    printf("%d\n%.17g\n", it, fx);
    for (double v : x) printf("%.17g ", v);
    printf("\n");
    // the proximal term of the learners reads the solver's matrix afterwards: a H v must be finite and H positive
    moihgp::opt::BFGSMat B = solver.getBFGSMat();
    Vector v(n, 1.0), hv;
    B.apply_Hv(v, 0.5, hv);
    printf("%.17g\n", moihgp::opt::dot(v, hv));
    return 0;
}
