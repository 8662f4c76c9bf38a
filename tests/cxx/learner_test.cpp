// Drives the C++ learners of include/moihgp_cxx/ on the GPU; tests/test_cxx_learner.py feeds the cases and compares with the
// oracle.  Modes (first token on stdin):
//   objective  M L kern dt gamma W nticks | params0[np] | probe[np] | Y[nticks][M]
//              -> OnlineObjective: push_back every tick, then loss and gradient at `probe` (empty BFGS matrix)   (moihgp_online.h:40-93)
//   online     M L kern dt gamma W nticks seed | params0[np] | Y[nticks][M]
//              -> MOIHGPOnlineLearning::step per tick: yhat, then after the last tick the parameters, f(old), f(new)
//   regression M L kern dt nticks seed | params0[np] | Y[nticks][M]
//              -> RegressionObjective at params0 (apply_params), then predict()
//   online_dev M L kern dt gamma W nticks | params0[np] | Y[nticks][M]
//              -> the host learner and the device-vector learner (lbfgsb_dev.hpp) side by side from the same parameters: per tick both yhat,
//                 then both parameter vectors, iteration counts and final objective values
//   eigenlike  M L kern dt | params0[np] | y[M] ma[M]
//              -> the reference learner's call pattern with an Eigen-like vector library (tests/cxx/eigen_like.hpp)
#include <cstdio>
#include <cstring>
#include <vector>
#include "moihgp_cxx/moihgp_online.hpp"
#include "moihgp_cxx/moihgp_regression.hpp"
#include "moihgp_cxx/lbfgsb_dev.hpp"
#include "eigen_like.hpp"

using Vec = std::vector<double>;
static bool rd(Vec& v) { for (auto& e : v) if (scanf("%lf", &e) != 1) return false; return true; }
static void pr(const Vec& v) { for (double e : v) printf("%.17g ", e); printf("\n"); }

template <class SS> int objective_mode(size_t M, size_t L, double dt) {
    double gamma; size_t W; int nt;
    if (scanf("%lf %zu %d", &gamma, &W, &nt) != 3) return 2;
    moihgp::MOIHGP<SS> gp(dt, M, L, false);
    Vec p0(gp.getNumParam()), probe(gp.getNumParam()), y(M), grad;
    if (!rd(p0) || !rd(probe)) return 2;
    gp.update(p0);
    moihgp::OnlineObjective<SS> obj(&gp, gamma, W);
    Vec got = gp.getParams();
    pr(got);
    for (int t = 0; t < nt; t++) { if (!rd(y)) return 2; obj.push_back(y); }
    double loss = obj(probe, grad);
    printf("%.17g\n", loss); pr(grad); pr(obj.ma);
    printf("%zu\n", obj.Y.size());
    return 0;
}
template <class SS> int online_mode(size_t M, size_t L, double dt) {
    double gamma; size_t W; int nt; unsigned long long seed;
    if (scanf("%lf %zu %d %llu", &gamma, &W, &nt, &seed) != 4) return 2;
    moihgp::MOIHGPOnlineLearning<SS> learner(dt, M, L, gamma, W, false);
    printf("%zu %zu %zu %zu %zu %zu\n", learner.getNumParam(), learner.getNumOutput(), learner.getNumLatent(), learner.getNumIGPParam(),
           learner.getIGPDim(), learner.getWindowsize());
    Vec first = learner.getParams();
    pr(first);                                                     // ctor state: random near-identity U, S = 1, sigma = 1e-2, (1, 1, 0.1)
    Vec y(M);
    for (int t = 0; t < nt; t++) {
        if (!rd(y)) return 2;
        Vec yhat = learner.step(y);
        pr(yhat);
    }
    Vec pnew = learner.getParams(), g;
    pr(pnew);
    Vec pold = learner.objective().oldparams;
    double fold = learner.objective()(pold, g), fnew = learner.objective()(pnew, g);
    printf("%.17g %.17g\n", fold, fnew);
    return 0;
}
template <class SS> int online_dev_mode(size_t M, size_t L, double dt) {
    double gamma; size_t W; int nt;
    if (scanf("%lf %zu %d", &gamma, &W, &nt) != 3) return 2;
    moihgp::MOIHGPOnlineLearning<SS> host(dt, M, L, gamma, W, false);
    moihgp::MOIHGPOnlineLearningDev<SS> dev(dt, M, L, gamma, W, false);
    Vec p0(host.getNumParam()), y(M);
    if (!rd(p0)) return 2;
    host.setParams(p0); dev.setParams(p0);
    for (int t = 0; t < nt; t++) {
        if (!rd(y)) return 2;
        Vec a = host.step(y), b = dev.step(y);
        pr(a); pr(b);
    }
    Vec ph = host.getParams(), pd = dev.getParams();
    pr(ph); pr(pd);
    printf("%d %.17g\n", dev.last_iterations, dev.last_fx);
    return 0;
}
template <class SS> int regression_mode(size_t M, size_t L, double dt) {
    int nt;
    if (scanf("%d", &nt) != 1) return 2;
    moihgp::MOIHGPRegression<SS> reg(dt, M, L, (size_t)nt, false);
    Vec p0(reg.getNumParam()), g;
    if (!rd(p0)) return 2;
    std::vector<Vec> Y((size_t)nt, Vec(M));
    for (auto& y : Y) if (!rd(y)) return 2;
    reg.objective().set_data(Y);
    reg.objective().apply_params = true;
    double loss = reg.objective()(p0, g);
    printf("%.17g\n", loss); pr(g);
    std::vector<Vec> Yhat = reg.predict(Y);
    for (auto& yh : Yhat) pr(yh);
    int iters = reg.fit(Y);                                          // apply_params = true: a real fit
    Vec pfit = reg.getParams();
    double lfit = reg.objective()(pfit, g);
    printf("%d %.17g\n", iters, lfit);
    return 0;
}
template <class SS> int eigenlike_mode(size_t M, size_t L, double dt) {
    using eigen_like::VectorXd;
    moihgp::MOIHGP<SS>* gp = new moihgp::MOIHGP<SS>(dt, M, L, false);
    const size_t d = gp->getIGPDim(), P = gp->getNumIGPParam(), np = gp->getNumParam();
    VectorXd params = gp->getParams();                             // moihgp_online.h:31 `oldparams = _gp->getParams()`
    Vec p0(np);
    if (!rd(p0)) return 2;
    for (size_t i = 0; i < np; i++) params[(long)i] = p0[i];
    gp->update(params);                                             // :43
    VectorXd again;
    again = gp->getParams();                                        // assignment form (:152)
    std::vector<VectorXd> x(L, VectorXd((long)d).setZero()), xnew(L, VectorXd((long)d).setZero());
    std::vector<std::vector<VectorXd>> dx(L, std::vector<VectorXd>(P, VectorXd((long)d).setZero())), dxnew = dx;
    VectorXd y((long)M), ma((long)M), yhat, g((long)np);
    for (size_t m = 0; m < M; m++) if (scanf("%lf", &y[(long)m]) != 1) return 2;
    for (size_t m = 0; m < M; m++) if (scanf("%lf", &ma[(long)m]) != 1) return 2;
    gp->step(x, y - ma, xnew, yhat);                                // :178, y - ma is a lazy expression, yhat unsized
    yhat += ma;                                                     // :179
    x = xnew;
    gp->step(x, y - ma, dx, xnew, dxnew);                           // :64 / :89
    VectorXd yc = y - ma;                                           // :63
    double loss = gp->negLogLikelihood(x, yc, dx, g);               // :66
    printf("%.17g\n", loss);
    for (long i = 0; i < yhat.size(); i++) printf("%.17g ", yhat[i]); printf("\n");
    for (long i = 0; i < g.size(); i++) printf("%.17g ", g[i]); printf("\n");
    for (long i = 0; i < again.size(); i++) printf("%.17g ", again[i]); printf("\n");
    delete gp;
    return 0;
}

int main() {
    char mode[32]; size_t M, L; int kern; double dt;
    if (scanf("%31s %zu %zu %d %lf", mode, &M, &L, &kern, &dt) != 5) return 2;
    try {
#define DISPATCH(fn) (kern == 0 ? fn<moihgp::Matern32StateSpace>(M, L, dt) : fn<moihgp::Matern52StateSpace>(M, L, dt))
        if (!strcmp(mode, "objective")) return DISPATCH(objective_mode);
        if (!strcmp(mode, "online")) return DISPATCH(online_mode);
        if (!strcmp(mode, "online_dev")) return DISPATCH(online_dev_mode);
        if (!strcmp(mode, "regression")) return DISPATCH(regression_mode);
        if (!strcmp(mode, "eigenlike")) return DISPATCH(eigenlike_mode);
    } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 3; }
    return 2;
}
