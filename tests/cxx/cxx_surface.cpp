// Exercises moihgp::MOIHGP<SS> (include/moihgp_cxx/moihgp.hpp) the way the reference's C++ callers use the class
// (moihgp_online.h:61-70: update -> step(x,y,dx,xnew,dxnew) -> negLogLikelihood(x,y,dx,g) -> x = xnew).
// Reads a case from stdin, prints results to stdout; tests/test_cxx_surface.py compares them with the oracle.
#include <cstdio>
#include <vector>
#include "moihgp_cxx/moihgp.hpp"

using Vec = std::vector<double>;
int main() {
    size_t M, L; int kern; double dt;
    if (scanf("%zu %zu %d %lf", &M, &L, &kern, &dt) != 4) return 2;
    auto run = [&](auto& gp) {
        size_t d = gp.getIGPDim(), P = gp.getNumIGPParam(), np = gp.getNumParam();
        Vec params(np), y(M), grad;
        for (auto& v : params) if (scanf("%lf", &v) != 1) return 2;
        std::vector<Vec> x(L, Vec(d)), xnew;
        std::vector<std::vector<Vec>> dx(L, std::vector<Vec>(P, Vec(d))), dxnew;
        for (auto& xl : x) for (auto& v : xl) if (scanf("%lf", &v) != 1) return 2;
        for (auto& a : dx) for (auto& b : a) for (auto& v : b) if (scanf("%lf", &v) != 1) return 2;
        int nticks; if (scanf("%d", &nticks) != 1) return 2;
        gp.update(params);
        Vec p = gp.getParams();                               // converts from MOIHGP::Params
        printf("%zu %zu %zu\n", d, P, np);
        for (double v : p) printf("%.17g ", v); printf("\n");
        for (int t = 0; t < nticks; t++) {
            for (auto& v : y) if (scanf("%lf", &v) != 1) return 2;
            double loss = gp.negLogLikelihood(x, y, dx, grad);      // pre-step state (moihgp_online.h:66)
            double loss2 = gp.negLogLikelihood(x, y);
            Vec yhat;
            gp.step(x, y, dx, xnew, yhat, dxnew);
            printf("%.17g %.17g\n", loss, loss2);
            for (double v : grad) printf("%.17g ", v); printf("\n");
            for (double v : yhat) printf("%.17g ", v); printf("\n");
            x = xnew; dx = dxnew;
        }
        Vec yh; gp.step(x, xnew, yh);                                 // prediction-only overload
        for (auto& xl : xnew) for (double v : xl) printf("%.17g ", v); printf("\n");
        return 0;
    };
    try {
        if (kern == 0) { moihgp::MOIHGP<moihgp::Matern32StateSpace> gp(dt, M, L, false); return run(gp); }
        moihgp::MOIHGP<moihgp::Matern52StateSpace> gp(dt, M, L, false); return run(gp);
    } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 3; }
}
