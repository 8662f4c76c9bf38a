// BASELINE.json configs[2] as a LOOP: "M=4096 outputs, T=10000, Matern-5/2, fp32, online-learning L-BFGS outer loop" -- ticks of the online learner
// (moihgp_online.h:173-187: filter the new observation, then re-fit on the window with <= 5 L-BFGS iterations) with the parameter vector,
// its gradient and the optimiser's correction pairs on the device (include/moihgp_cxx/lbfgsb_dev.hpp).  Prints one JSON object.
//   learner_bench M L W nticks [kern] [threading]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "moihgp_cxx/lbfgsb_dev.hpp"

template <class SS> int run(size_t M, size_t L, size_t W, int nt, bool threading) {
    using clk = std::chrono::steady_clock;
    moihgp::MOIHGPOnlineLearningDev<SS> learner(0.1, M, L, 0.9, W, threading);
    std::vector<double> y(M);
    unsigned long long s = 20260101ULL;
    auto rnd = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return ((s >> 11) * (1.0 / 9007199254740992.0)) - 0.5; };
    double total = 0.0, worst = 0.0;
    int evals_like = 0;
    size_t evals0 = 0, skipped0 = 0;
    for (int t = 0; t < nt + 1; t++) {
        for (size_t m = 0; m < M; m++) y[m] = std::sin(0.05 * t * (1 + m % 7)) + 0.2 * rnd();
        if (t == 1 && getenv("LEARNER_BENCH_PHASES")) moihgp::opt::phases().enabled = true;            // (phase table on stderr; tick 0 excluded)
        const auto t0 = clk::now();
        std::vector<double> yhat = learner.step(y);
        const double sec = std::chrono::duration<double>(clk::now() - t0).count();
        if (t > 0) { total += sec; worst = sec > worst ? sec : worst; evals_like += learner.last_iterations; }       // (tick 0 warms up: allocations, first polar factor)
        else { evals0 = learner.objective().evaluations; skipped0 = learner.objective().updates_skipped; }
        if (!std::isfinite(yhat[0])) { fprintf(stderr, "non-finite prediction at tick %d\n", t); return 3; }
    }
    if (moihgp::opt::phases().enabled) {
        fprintf(stderr, "phases over %d ticks (%.1f ms per tick in all):\n", nt, total / nt * 1e3);
        for (const auto& kv : moihgp::opt::phases().seconds)
            fprintf(stderr, "  %-48s %8.2f ms per tick  (%ld calls)\n", kv.first.c_str(), kv.second / nt * 1e3, moihgp::opt::phases().calls[kv.first]);
    }
    printf("{\"outputs\": %zu, \"latents\": %zu, \"window\": %zu, \"ticks\": %d, \"seconds_per_tick\": %.6f, \"worst_tick_seconds\": %.6f, "
           "\"lbfgs_iterations_per_tick\": %.2f, \"objective_evaluations_per_tick\": %.2f, \"updates_skipped_per_tick\": %.2f, \"threading\": %s, \"num_param\": %zu, \"final_objective\": %.10g}\n",
           M, L, W, nt, total / nt, worst, (double)evals_like / nt, (double)(learner.objective().evaluations - evals0) / nt, (double)(learner.objective().updates_skipped - skipped0) / nt, threading ? "true" : "false",
           learner.getNumParam(), learner.last_fx);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: learner_bench M L W nticks [kern: 0 Matern-3/2, 1 Matern-5/2]\n"); return 2; }
    const size_t M = strtoull(argv[1], 0, 10), L = strtoull(argv[2], 0, 10), W = strtoull(argv[3], 0, 10);
    const int nt = atoi(argv[4]), kern = argc > 5 ? atoi(argv[5]) : 1;
    const bool threading = argc > 6 && atoi(argv[6]) != 0;
    try { return kern == 0 ? run<moihgp::Matern32StateSpace>(M, L, W, nt, threading) : run<moihgp::Matern52StateSpace>(M, L, W, nt, threading); }
    catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 3; }
}
