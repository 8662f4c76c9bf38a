// stationary_x.hip -- IHGP::update (filter-mode part, ihgp.h:117-133) for STACKED models: one lane per latent, fp64.
//
// A stacked model is the sum of J Matern components observed through one output (include/moihgp.h, MOIHGP_STACK): not a
// class of the reference, but exactly what its IHGP<StateSpace> template computes for a StateSpace whose F and Pinf are
// block-diagonal in the reference's own component models (matern32ss.h:40-64 / matern52ss.h:38-75) and H = [H_1 .. H_J]:
//     A = expm(dt F); Q = sym(Pinf - A Pinf A^T); PP = DARE(A, H^T, Q, R); S; K; HA; AKHA        (ihgp.h:120-130)
// with the literal solver of utils/dare.h:10-33 on the full D x D system (D = J * d_base <= 12).
// The matrix exponential is taken block by block (expm of a block-diagonal matrix is block-diagonal); a literal evaluation
// exponentiates the full matrix as ihgp.h:120 does: the two agree to rounding (checked in tests/).
// Hyper-parameter sensitivities (ihgp.h:136-200) are not built for stacked models yet: they are filter-mode latents.
#include "common.h"

#pragma clang fp contract(off)
#include "stationary_common.h"

namespace moihgp {
namespace {

// F and Pinf of one component (the reference's model code, restated in stationary.hip's ss_build as well)
template <int DB>
__device__ void component(double magnitude, double lengthscale, double* F, double* Pinf) {
    for (int i = 0; i < DB * DB; i++) { F[i] = 0.0; Pinf[i] = 0.0; }
    if constexpr (DB == 2) {                                  // matern32ss.h:40-52
        double lam = sqrt(3.0) / lengthscale, lam2 = lam * lam;
        F[1] = 1.0; F[2] = -lam2; F[3] = -2.0 * lam;
        Pinf[0] = magnitude; Pinf[3] = magnitude * lam2;
    } else {                                                  // matern52ss.h:38-58, `lam = sqrt(3)/l` as there
        double lam = sqrt(3.0) / lengthscale, lam2 = lam * lam, len2 = lengthscale * lengthscale, len4 = len2 * len2;
        double kappa = 5.0 / 3.0 * magnitude / len2;
        F[1] = 1.0; F[5] = 1.0; F[6] = -lam2 * lam; F[7] = -3.0 * lam2; F[8] = -3.0 * lam;
        Pinf[0] = magnitude; Pinf[8] = 25.0 * magnitude / len4; Pinf[4] = kappa; Pinf[6] = -kappa; Pinf[2] = -kappa;
    }
}

template <int DB, int J>
__global__ void __launch_bounds__(64) stack_update_kernel(double dt, const double* __restrict__ params, size_t n,
                                                          double* __restrict__ cb64, float* __restrict__ cb32,
                                                          int* __restrict__ n_unstable) {
    constexpr int D = DB * J, NN = D * D, P = 2 * J + 1;
    using L = XC<D>;
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    const double* prm = params + l * P;
    const double R = prm[2 * J];
    double A[NN], Pinf[NN], H[D], Q[NN], PP[NN], T1[NN], T2[NN];
    for (int i = 0; i < NN; i++) { A[i] = 0.0; Pinf[i] = 0.0; }
    for (int i = 0; i < D; i++) H[i] = 0.0;
    for (int j = 0; j < J; j++) {
        double F[DB * DB], Pj[DB * DB], E[DB * DB];
        component<DB>(prm[2 * j], prm[2 * j + 1], F, Pj);
        for (int i = 0; i < DB * DB; i++) F[i] *= dt;
        expm<DB>(F, E);                                                  // ihgp.h:120, block j
        for (int a = 0; a < DB; a++)
            for (int b = 0; b < DB; b++) {
                A[(j * DB + a) * D + j * DB + b] = E[a * DB + b];
                Pinf[(j * DB + a) * D + j * DB + b] = Pj[a * DB + b];
            }
        H[j * DB] = 1.0;
    }
    mt<D>(A, T2);
    mm<D>(A, Pinf, T1); mm<D>(T1, T2, T2);
    for (int i = 0; i < NN; i++) T1[i] = Pinf[i] - T2[i];              // ihgp.h:121
    for (int i = 0; i < D; i++)
        for (int j = 0; j < D; j++) Q[i * D + j] = (T1[i * D + j] + T1[j * D + i]) / 2.0;   // ihgp.h:122
    int dare_iters = dare<D>(A, H, Q, R, PP);                          // ihgp.h:125
    double K[D], HA[D], AKHA[NN];
    mv<D>(PP, H, K);
    double S = R;
    for (int i = 0; i < D; i++) S += H[i] * K[i];                       // ihgp.h:126
    for (int i = 0; i < D; i++) K[i] = K[i] / S;                        // ihgp.h:127
    for (int j = 0; j < D; j++) { double t = 0.0; for (int i = 0; i < D; i++) t += H[i] * A[i * D + j]; HA[j] = t; }   // ihgp.h:129
    for (int i = 0; i < D; i++)
        for (int j = 0; j < D; j++) AKHA[i * D + j] = A[i * D + j] - K[i] * HA[j];   // ihgp.h:130

    double* o64 = cb64 + l * L::SIZE;
    float* o32 = cb32 + l * L::SIZE;
    auto put = [&](int off, double v) { o64[off] = v; o32[off] = (float)v; };
    for (int i = L::AB; i < L::SIZE; i++) put(i, 0.0);                   // padding of the slab tables
    for (int i = 0; i < NN; i++) { put(L::AKHA + i, AKHA[i]); put(L::A + i, A[i]); }
    for (int i = 0; i < D; i++) { put(L::K + i, K[i]); put(L::HA + i, HA[i]); put(L::K16 + i, K[i]); put(L::HA16 + i, HA[i]); }
    put(L::S, S); put(L::LOGS, log(S)); put(L::ITERS, (double)dare_iters);
    for (int j = 0; j < J; j++)
        for (int a = 0; a < DB; a++)
            for (int b = 0; b < DB; b++) put(L::AB + j * DB * DB + a * DB + b, A[(j * DB + a) * D + j * DB + b]);

    // tables of the segment solve (recursion_x.hip): g_k = AKHA^(CK-1-k) K, and M^(2^lv) with M = AKHA^CK
    bool ok = true;
    double g[D];
    for (int i = 0; i < D; i++) g[i] = K[i];
    for (int k = kChunkX - 1; k >= 0; k--) {
        for (int i = 0; i < D; i++) { put(L::G + k * D + i, g[i]); ok = ok && (fabs(g[i]) < 1e18); }
        mv<D>(AKHA, g, g);
    }
    for (int i = 0; i < NN; i++) T1[i] = AKHA[i];
    for (int q = 1; q < kChunkX; q <<= 1) mm<D>(T1, T1, T1);            // M = AKHA^CK
    int nlev64 = 1, nlev32 = 1;
    for (int lv = 0; lv < 6; lv++) {
        double big = 0.0;
        for (int i = 0; i < NN; i++) { put(L::SP + lv * L::LS + i, T1[i]); ok = ok && (fabs(T1[i]) < 1e18); big = fmax(big, fabs(T1[i])); }   // false for NaN too
        if (big * D >= 1e-20) nlev64 = lv + 1;
        if (big * D >= 1e-10) nlev32 = lv + 1;
        mm<D>(T1, T1, T1);
    }
    o64[L::NLEV] = (double)nlev64;
    o32[L::NLEV] = (float)nlev32;
    put(L::SCANOK, ok ? 1.0 : 0.0);
    if (!ok) { atomicAdd(&n_unstable[0], 1); atomicAdd(&n_unstable[1], 1); }
}

template <int DB, int J>
void launch_t(double dt, const double* params, size_t n, double* cb64, float* cb32, int* n_unstable, hipStream_t s) {
    hipLaunchKernelGGL((stack_update_kernel<DB, J>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, dt, params, n, cb64, cb32, n_unstable);
}

}  // namespace

void launch_stack_update(int kernel, double dt, const double* params_dev, size_t n, double* cb64, float* cb32, int* n_unstable,
                         hipStream_t stream) {
    if (n == 0) return;
    MOIHGP_HIP_FATAL(hipMemsetAsync(n_unstable, 0, 2 * sizeof(int), stream));
    const int base = kernel_base(kernel), J = kernel_stack(kernel);
    if (base == 0) {
        if (J == 2) launch_t<2, 2>(dt, params_dev, n, cb64, cb32, n_unstable, stream);
        else if (J == 3) launch_t<2, 3>(dt, params_dev, n, cb64, cb32, n_unstable, stream);
        else launch_t<2, 4>(dt, params_dev, n, cb64, cb32, n_unstable, stream);
    } else {
        if (J == 2) launch_t<3, 2>(dt, params_dev, n, cb64, cb32, n_unstable, stream);
        else if (J == 3) launch_t<3, 3>(dt, params_dev, n, cb64, cb32, n_unstable, stream);
        else launch_t<3, 4>(dt, params_dev, n, cb64, cb32, n_unstable, stream);
    }
    MOIHGP_HIP_FATAL(hipGetLastError());
}

}  // namespace moihgp
