// stationary_x.hip -- IHGP::update (filter-mode part, ihgp.h:117-133) for STACKED models: one wavefront per latent, fp64.
//
// A stacked model is the sum of J Matern components observed through one output (include/moihgp.h, MOIHGP_STACK): not a
// class of the reference, but exactly what its IHGP<StateSpace> template computes for a StateSpace whose F and Pinf are
// block-diagonal in the reference's own component models (matern32ss.h:40-64 / matern52ss.h:38-75) and H = [H_1 .. H_J]:
//     A = expm(dt F); Q = sym(Pinf - A Pinf A^T); PP = DARE(A, H^T, Q, R); S; K; HA; AKHA        (ihgp.h:120-130)
// with the literal solver of utils/dare.h:10-33 on the full D x D system (D = J * d_base <= 12).
// The matrix exponential is taken block by block (expm of a block-diagonal matrix is block-diagonal); a literal evaluation
// exponentiates the full matrix as ihgp.h:120 does: the two agree to rounding (checked in tests/).
// The hyper-parameter sensitivities (ihgp.h:136-200) follow, one parameter after the other, into a separate fp64 block (XD).
#include "common.h"

#pragma clang fp contract(off)
#include "stationary_common.h"

namespace moihgp {
namespace {

// F and Pinf of one component (the reference's model code, restated in stationary.hip's ss_build as well)
template <int DB>
__device__ void component(double magnitude, double lengthscale, double* F, double* Pinf, double* dF_len = nullptr,
                          double* dPinf_mag = nullptr, double* dPinf_len = nullptr) {
    for (int i = 0; i < DB * DB; i++) {
        F[i] = 0.0; Pinf[i] = 0.0;
        if (dF_len) { dF_len[i] = 0.0; dPinf_mag[i] = 0.0; dPinf_len[i] = 0.0; }
    }
    if constexpr (DB == 2) {                                  // matern32ss.h:40-64
        double lam = sqrt(3.0) / lengthscale, lam2 = lam * lam;
        double len3 = 6.0 / (lengthscale * lengthscale * lengthscale);
        F[1] = 1.0; F[2] = -lam2; F[3] = -2.0 * lam;
        Pinf[0] = magnitude; Pinf[3] = magnitude * lam2;
        if (dF_len) {
            dF_len[2] = len3; dF_len[3] = 2.0 * lam / lengthscale;
            dPinf_mag[0] = 1.0; dPinf_mag[3] = lam2;
            dPinf_len[3] = -magnitude * len3;
        }
    } else {                                                  // matern52ss.h:38-75, `lam = sqrt(3)/l` as there
        double lam = sqrt(3.0) / lengthscale, lam2 = lam * lam, len2 = lengthscale * lengthscale, len3 = len2 * lengthscale, len4 = len2 * len2;
        double kappa = 5.0 / 3.0 * magnitude / len2, kappa2 = -2.0 * kappa / lengthscale, sq5 = sqrt(5.0);
        F[1] = 1.0; F[5] = 1.0; F[6] = -lam2 * lam; F[7] = -3.0 * lam2; F[8] = -3.0 * lam;
        Pinf[0] = magnitude; Pinf[8] = 25.0 * magnitude / len4; Pinf[4] = kappa; Pinf[6] = -kappa; Pinf[2] = -kappa;
        if (dF_len) {
            dF_len[6] = 15.0 * sq5 / len4; dF_len[7] = 30.0 / len3; dF_len[8] = sq5 * lam2;
            for (int i = 0; i < 9; i++) dPinf_mag[i] = Pinf[i] / magnitude;
            dPinf_len[4] = kappa2; dPinf_len[6] = -kappa2; dPinf_len[2] = -kappa2;
            dPinf_len[8] = -100.0 * magnitude / len2 / len3;
        }
    }
}

// ---- wave-cooperative dense helpers: a D x D matrix lives in (wave-private) LDS, lane e owns entries e, e+64, e+128.
// Every entry is accumulated in the same k order as a plain sequential loop (and FP contraction is off), so results -- and
// with them the DARE iteration count -- are those of a one-thread evaluation.
__device__ inline void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int D>
__device__ inline void wmm(const double* A, const double* B, double* C, int lane) {          // C = A B (C may alias A or B)
    double r[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int e = lane + 64 * q;
        double sum = 0.0;
        if (e < D * D) {
            const int i = e / D, j = e % D;
            for (int k = 0; k < D; k++) sum += A[i * D + k] * B[k * D + j];
        }
        r[q] = sum;
    }
    lds_sync();
#pragma unroll
    for (int q = 0; q < 3; q++) if (lane + 64 * q < D * D) C[lane + 64 * q] = r[q];
    lds_sync();
}
template <int D>
__device__ inline void wmm_bt(const double* A, const double* B, double* C, int lane) {       // C = A B^T (C may alias A or B)
    double r[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int e = lane + 64 * q;
        double sum = 0.0;
        if (e < D * D) {
            const int i = e / D, j = e % D;
            for (int k = 0; k < D; k++) sum += A[i * D + k] * B[j * D + k];
        }
        r[q] = sum;
    }
    lds_sync();
#pragma unroll
    for (int q = 0; q < 3; q++) if (lane + 64 * q < D * D) C[lane + 64 * q] = r[q];
    lds_sync();
}
template <int D>
__device__ inline void wmt(const double* A, double* At, int lane) {                           // At = A^T (no aliasing)
#pragma unroll
    for (int q = 0; q < 3; q++) { const int e = lane + 64 * q; if (e < D * D) At[(e % D) * D + e / D] = A[e]; }
    lds_sync();
}
template <int D>
__device__ inline void wmv(const double* A, const double* x, double* y, int lane) {            // y = A x (y may alias x)
    double sum = 0.0;
    if (lane < D) for (int k = 0; k < D; k++) sum += A[lane * D + k] * x[k];
    lds_sync();
    if (lane < D) y[lane] = sum;
    lds_sync();
}

template <int DB, int J>
__global__ void __launch_bounds__(64) stack_update_kernel(double dt, const double* __restrict__ params, size_t n,
                                                          double* __restrict__ cb64, float* __restrict__ cb32,
                                                          double* __restrict__ cbd64, int* __restrict__ n_unstable,
                                                          int* __restrict__ rescue_idx /* or NULL: latents whose fp32 tables are unusable but whose fp64 ones are (listed,
                                                                                          counted in n_unstable[2], their fp32 SCANOK = -1: capi.cpp sweeps them in fp64) */) {
    constexpr int D = DB * J, NN = D * D, P = 2 * J + 1, BB = DB * DB;
    using L = XC<D>;
    __shared__ double sA[NN], sAT[NN], sQ[NN], sP[NN], sT1[NN], sT2[NN], sAKHA[NN], sH[D], sV1[D], sV2[D], sV3[D], sK[D], sHA[D];
    // sensitivities (ihgp.h:136-200)
    __shared__ double sPinf[NN], sAAKH[NN], sdA[NN], sdPi[NN], sQL[NN], sT3[NN], sAK[D], sPHt[D], sHPP[D];
    __shared__ double sdAb[J][BB], sdPm[J][BB], sdPl[J][BB];
    const size_t l = blockIdx.x;                                   // one wavefront per latent
    const int lane = threadIdx.x;
    if (l >= n) return;
    const double* prm = params + l * P;
    const double R = prm[2 * J];
    for (int e = lane; e < NN; e += 64) { sA[e] = 0.0; sT1[e] = 0.0; }     // sT1 = Pinf for now
    if (lane < D) sH[lane] = (lane % DB == 0) ? 1.0 : 0.0;
    lds_sync();
    if (lane < J) {                                                   // ihgp.h:120, block by block (one lane per block)
        double F[DB * DB], Pj[DB * DB], E[DB * DB];
        component<DB>(prm[2 * lane], prm[2 * lane + 1], F, Pj);
        for (int i = 0; i < DB * DB; i++) F[i] *= dt;
        expm<DB>(F, E);
        for (int a = 0; a < DB; a++)
            for (int b = 0; b < DB; b++) {
                sA[(lane * DB + a) * D + lane * DB + b] = E[a * DB + b];
                sT1[(lane * DB + a) * D + lane * DB + b] = Pj[a * DB + b];
            }
    } else if (cbd64 && lane >= 16 && lane < 16 + J) {
        // meanwhile, for the lengthscale of component j: the 2 DB x 2 DB block exponential of ihgp.h:163-167 (block j of dA)
        const int j = lane - 16;
        double F[BB], Pj[BB], dFl[BB], dPm[BB], dPl[BB];
        component<DB>(prm[2 * j], prm[2 * j + 1], F, Pj, dFl, dPm, dPl);
        // evaluated on its blocks (expm_blt: [[F, 0], [dF, F]] is block lower triangular); dA_j is the lower left block
        double X[BB], Y[BB], EX[BB], EY[BB];
        for (int i = 0; i < BB; i++) { X[i] = dt * F[i]; Y[i] = dt * dFl[i]; }
        expm_blt<DB>(X, Y, EX, EY);
        for (int a = 0; a < DB; a++)
            for (int b = 0; b < DB; b++) {
                sdAb[j][a * DB + b] = EY[a * DB + b];
                sdPm[j][a * DB + b] = dPm[a * DB + b];
                sdPl[j][a * DB + b] = dPl[a * DB + b];
            }
    }
    lds_sync();
    for (int e = lane; e < NN; e += 64) sPinf[e] = sT1[e];
    lds_sync();
    wmt<D>(sA, sAT, lane);
    wmm<D>(sA, sT1, sT2, lane);                                       // A Pinf
    wmm<D>(sT2, sAT, sT2, lane);                                      // A Pinf A^T
    for (int e = lane; e < NN; e += 64) sP[e] = sT1[e] - sT2[e];      // ihgp.h:121 (sP as scratch)
    lds_sync();
    for (int e = lane; e < NN; e += 64) sQ[e] = (sP[e] + sP[(e % D) * D + e / D]) / 2.0;     // ihgp.h:122
    lds_sync();

    // ---- DARE, utils/dare.h:10-33 with Ad = A, Bd = H^T ----
    for (int e = lane; e < NN; e += 64) sP[e] = sQ[e];
    lds_sync();
    int dare_iters = kDareMaxIter;
    for (int it = 0; it < kDareMaxIter; it++) {
        wmm<D>(sAT, sP, sT1, lane);                                   // AdT P
        wmm<D>(sT1, sA, sT2, lane);                                   // AdT P Ad
        wmv<D>(sP, sH, sV1, lane);                                    // PB = P Bd
        double g = R;
        for (int i = 0; i < D; i++) g += sH[i] * sV1[i];              // R + BdT P Bd (every lane, same order)
        wmv<D>(sAT, sV1, sV2, lane);                                  // APB = AdT P Bd
        {   // BtP[j] = sum_i Bd[i] P[i][j];  BPA[j] = sum_k BtP[k] Ad[k][j]
            double sum = 0.0;
            if (lane < D) for (int i = 0; i < D; i++) sum += sH[i] * sP[i * D + lane];
            lds_sync();
            if (lane < D) sV3[lane] = sum;
            lds_sync();
            sum = 0.0;
            if (lane < D) for (int k = 0; k < D; k++) sum += sV3[k] * sA[k * D + lane];
            lds_sync();
            if (lane < D) sV3[lane] = sum;                            // BPA
            lds_sync();
        }
        const double ginv = 1.0 / g;
        double diff = -INFINITY;
        for (int e = lane; e < NN; e += 64) {
            const int i = e / D, j = e % D;
            const double v = sT2[e] - sV2[i] * ginv * sV3[j] + sQ[e];   // dare.h:23
            sT1[e] = v;
            const double dlt = v - sP[e];
            if (dlt > diff) diff = dlt;                               // maxCoeff, dare.h:25
        }
        for (int o = 32; o >= 1; o >>= 1) { const double other = __shfl_xor(diff, o, 64); if (other > diff) diff = other; }
        diff = fabs(diff);
        lds_sync();
        for (int e = lane; e < NN; e += 64) sP[e] = (sT1[e] + sT1[(e % D) * D + e / D]) / 2.0;   // dare.h:26
        lds_sync();
        if (diff < kDareTol) { dare_iters = it + 1; break; }
    }
    wmv<D>(sP, sH, sV1, lane);                                        // PP H^T
    double S = R;
    for (int i = 0; i < D; i++) S += sH[i] * sV1[i];                  // ihgp.h:126
    if (lane < D) sK[lane] = sV1[lane] / S;                           // ihgp.h:127
    {
        double t = 0.0;
        if (lane < D) for (int i = 0; i < D; i++) t += sH[i] * sA[i * D + lane];
        if (lane < D) sHA[lane] = t;                                  // ihgp.h:129
    }
    lds_sync();
    for (int e = lane; e < NN; e += 64) sAKHA[e] = sA[e] - sK[e / D] * sHA[e % D];             // ihgp.h:130
    lds_sync();

    double* o64 = cb64 + l * L::SIZE;
    float* o32 = cb32 + l * L::SIZE;
    auto put = [&](int off, double v) { o64[off] = v; o32[off] = (float)v; };
    // every entry of the block is written exactly once (the slab tables with their zero padding)
    for (int e = lane; e < NN; e += 64) { put(L::AKHA + e, sAKHA[e]); put(L::A + e, sA[e]); }
    if (lane < D) { put(L::K + lane, sK[lane]); put(L::HA + lane, sHA[lane]); }
    if (lane < 16) { put(L::K16 + lane, lane < D ? sK[lane] : 0.0); put(L::HA16 + lane, lane < D ? sHA[lane] : 0.0); }
    if (lane == 0) { put(L::S, S); put(L::LOGS, log(S)); put(L::ITERS, (double)dare_iters); }
    for (int e = lane; e < L::HA16 - L::AB; e += 64) {
        const int j = e / (DB * DB), a = (e / DB) % DB, b = e % DB;
        put(L::AB + e, e < J * DB * DB ? sA[(j * DB + a) * D + j * DB + b] : 0.0);
    }
    for (int e = kChunkX * D + lane; e < L::GN; e += 64) put(L::G + e, 0.0);
    for (int e = lane; e < L::SIZE - L::PK; e += 64) {              // coefficient pairs of the packed fp32 replay (and the block's zero padding)
        constexpr int PB = DB * DB + 2 * DB, NPAIR = (J + 1) / 2;
        const int pr = e / 2, half = e % 2, p = pr / PB, i = pr % PB, j = 2 * p + half;      // block j supplies this half of pair i of block pair p
        double v = 0.0;
        if (p < NPAIR && j < J) {
            if (i < DB * DB) v = sA[(j * DB + i / DB) * D + j * DB + i % DB];
            else if (i < DB * DB + DB) v = sHA[j * DB + (i - DB * DB)];
            else v = sK[j * DB + (i - DB * DB - DB)];
        }
        put(L::PK + e, v);
    }

    // tables of the segment solve (recursion_x.hip): g_k = AKHA^(CK-1-k) K, and M^(2^lv) with M = AKHA^CK
    // ("tame" per precision: a mildly unstable latent -- the literal DARE of dare.h:23 does return such gains -- still scans as
    // long as the largest power stays far inside the format's range: 1e150 in fp64, 1e18 in fp32)
    bool ok = true, ok32 = true;
    if (lane < D) sV1[lane] = sK[lane];
    lds_sync();
    for (int k = kChunkX - 1; k >= 0; k--) {
        if (lane < D) { put(L::G + k * D + lane, sV1[lane]); ok = ok && (fabs(sV1[lane]) < 1e150); ok32 = ok32 && (fabs(sV1[lane]) < 1e18); }
        wmv<D>(sAKHA, sV1, sV1, lane);
    }
    for (int e = lane; e < NN; e += 64) sT1[e] = sAKHA[e];
    lds_sync();
    for (int q = 1; q < kChunkX; q <<= 1) wmm<D>(sT1, sT1, sT1, lane);  // M = AKHA^CK
    int nlev64 = 1, nlev32 = 1;
    for (int lv = 0; lv < 6; lv++) {
        double big = 0.0;
        for (int e = lane; e < L::LS; e += 64) {
            const double v = e < NN ? sT1[e] : 0.0;
            put(L::SP + lv * L::LS + e, v); ok = ok && (fabs(v) < 1e150); ok32 = ok32 && (fabs(v) < 1e18); big = fmax(big, fabs(v));   // false for NaN too
        }
        for (int o = 32; o >= 1; o >>= 1) big = fmax(big, __shfl_xor(big, o, 64));
        if (big * D >= 1e-20) nlev64 = lv + 1;
        if (big * D >= 1e-10) nlev32 = lv + 1;
        wmm<D>(sT1, sT1, sT1, lane);
    }
    ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
    ok32 = __builtin_amdgcn_ballot_w64(!ok32) == 0;
    if (lane == 0) {
        o64[L::NLEV] = (double)nlev64;
        o32[L::NLEV] = (float)nlev32;
        o64[L::SCANOK] = ok ? 1.0 : 0.0;
        const bool rescue = rescue_idx && ok && !ok32;
        o32[L::SCANOK] = ok32 ? 1.0f : (rescue ? -1.0f : 0.0f);
        if (!ok) atomicAdd(&n_unstable[0], 1);
        if (!ok32) atomicAdd(&n_unstable[1], 1);
        if (rescue) rescue_idx[atomicAdd(&n_unstable[2], 1)] = (int)l;
        if (nlev64 > 5) atomicAdd(&n_unstable[3], 1);                  // a filter that remembers more than ~1000 ticks (capi.cpp: imputation of missing ticks)
    }
    if (!cbd64) return;

    // ---- sensitivities, ihgp.h:136-200, one hyper-parameter after the other: p = 2j (magnitude of component j),
    //      2j + 1 (its lengthscale), 2J (noise).  dF != 0 only for lengthscales; dPinf == 0 only for the noise; dR != 0 only for it.
    using X = XD<D, P>;
    double* od = cbd64 + l * X::SIZE;
    wmv<D>(sP, sH, sPHt, lane);                                       // PP H^T
    if (lane < D) { double t = 0.0; for (int i = 0; i < D; i++) t += sH[i] * sP[i * D + lane]; sHPP[lane] = t; }   // H PP
    wmv<D>(sA, sK, sAK, lane);                                        // ihgp.h:132
    for (int e = lane; e < NN; e += 64) sAAKH[e] = sA[e] - sAK[e / D] * sH[e % D];                // ihgp.h:133
    lds_sync();
    wmt<D>(sAAKH, sT3, lane);                                         // Ad^T of the Lyapunov iteration
    for (int p = 0; p < P; p++) {
        const int j = p / 2, q = p % 2;
        const bool noise = (p == 2 * J);
        const bool dF_zero = noise || q == 0, dR_zero = !noise;
        const double dR = noise ? 1.0 : 0.0;
        for (int e = lane; e < NN; e += 64) {
            const int r = e / D, c = e % D;
            const bool inb = !noise && r / DB == j && c / DB == j;
            const int be = (r % DB) * DB + (c % DB);
            sdA[e] = (inb && q == 1) ? sdAb[j][be] : 0.0;
            sdPi[e] = inb ? (q == 0 ? sdPm[j][be] : sdPl[j][be]) : 0.0;
        }
        lds_sync();
        if (dF_zero) {                                                // ihgp.h:141
            if (noise) {
                for (int e = lane; e < NN; e += 64) sQ[e] = 0.0;      // ihgp.h:146
            } else {
                wmm<D>(sA, sdPi, sT1, lane); wmm_bt<D>(sT1, sA, sT2, lane);
                for (int e = lane; e < NN; e += 64) sQ[e] = sdPi[e] - sT2[e];          // ihgp.h:150
            }
            lds_sync();
            for (int e = lane; e < NN; e += 64)                       // ihgp.h:154 / :158 (as AK dR AK^T)
                sQL[e] = dR_zero ? sQ[e] : sAK[e / D] * dR * sAK[e % D] + sQ[e];
            lds_sync();
        } else {
            wmm<D>(sdA, sPinf, sT1, lane); wmm_bt<D>(sT1, sA, sT2, lane);               // dA Pinf A^T
            wmm<D>(sA, sPinf, sT1, lane); wmm_bt<D>(sT1, sdA, sQL, lane);               // A Pinf dA^T
            wmm<D>(sA, sdPi, sT1, lane); wmm_bt<D>(sT1, sA, sQ, lane);                  // A dPinf A^T
            for (int e = lane; e < NN; e += 64) sQ[e] = sdPi[e] - sT2[e] - sQ[e] - sQL[e];   // ihgp.h:175
            lds_sync();
            wmm<D>(sdA, sP, sT1, lane); wmm_bt<D>(sT1, sA, sT2, lane);                  // dA PP A^T
            wmm<D>(sA, sP, sT1, lane); wmm_bt<D>(sT1, sdA, sQL, lane);                  // A PP dA^T
            wmv<D>(sdA, sPHt, sV1, lane);                                                // dA PP H^T
            wmv<D>(sdA, sHPP, sV2, lane);                                                // (H PP dA^T)^T
            for (int e = lane; e < NN; e += 64) {                     // ihgp.h:179
                const double v = sT2[e] + sQL[e] - sV1[e / D] * sAK[e % D] - sAK[e / D] * sV2[e % D];
                sT1[e] = v + sQ[e];
            }
            lds_sync();
            for (int e = lane; e < NN; e += 64) sQL[e] = sT1[e];
            lds_sync();
        }
        // dPP = DLyap(A - A K H, QLyap), utils/dare.h:36-58 (literal `AdT P Ad - P + Q`); P lives in sQ
        for (int e = lane; e < NN; e += 64) sQ[e] = sQL[e];
        lds_sync();
        int its = kDareMaxIter;
        for (int it = 0; it < kDareMaxIter; it++) {
            wmm<D>(sT3, sQ, sT1, lane);
            wmm<D>(sT1, sAAKH, sT2, lane);
            double diff = -INFINITY;
            for (int e = lane; e < NN; e += 64) {
                const double v = sT2[e] - sQ[e] + sQL[e];             // dare.h:48 (sic)
                sT1[e] = v;
                const double dlt = v - sQ[e];
                if (dlt > diff) diff = dlt;
            }
            for (int o = 32; o >= 1; o >>= 1) { const double other = __shfl_xor(diff, o, 64); if (other > diff) diff = other; }
            diff = fabs(diff);
            lds_sync();
            for (int e = lane; e < NN; e += 64) sQ[e] = (sT1[e] + sT1[(e % D) * D + e / D]) / 2.0;
            lds_sync();
            if (diff < kDareTol) { its = it + 1; break; }
        }
        double dS = dR;
        for (int i = 0; i < D; i++)
            for (int c = 0; c < D; c++) dS += sH[i] * sQ[i * D + c] * sH[c];             // ihgp.h:188
        if (lane < D) {                                               // ihgp.h:189
            double t = 0.0;
            for (int c = 0; c < D; c++) t += (sQ[lane * D + c] - sP[lane * D + c] * dS / S) * sH[c];
            sV1[lane] = t / S;                                        // dK
        }
        if (lane < D) { double t = 0.0; for (int i = 0; i < D; i++) t += sH[i] * sdA[i * D + lane]; sV2[lane] = t; }   // H dA
        lds_sync();
        if (lane == 0) { od[X::DS + p] = dS; od[X::ITERS + p] = (double)its; }
        if (lane < D) { od[X::DK + p * D + lane] = sV1[lane]; od[X::HDA + p * D + lane] = dF_zero ? 0.0 : sV2[lane]; }
        for (int e = lane; e < NN; e += 64) {
            od[X::DA + p * NN + e] = sdA[e];
            od[X::DAKHA + p * NN + e] = dF_zero ? -sV1[e / D] * sHA[e % D]                               // ihgp.h:192
                                                : sdA[e] - sV1[e / D] * sHA[e % D] - sK[e / D] * sV2[e % D];   // ihgp.h:197
        }
        lds_sync();
    }
}

// The few-latents team kernel of recursion_x.hip for the reference's OWN models (Matern-3/2, d = 2; Matern-5/2, d = 3: one component, J = 1):
// its tables in the stacked layout, from the matrices IHGP::update has already put into the latent's CB block (stationary.hip) -- the same
// AKHA, K, A, HA, S, log S to the last bit, so the two paths filter with one model; added here: the 32-row response table, the diagonal
// block, the padded slabs, the scan powers of 32-tick chunks with their level counts, and the coefficient pairs of the packed fp32 replay.
template <int D>
__global__ void __launch_bounds__(64) xc_from_cb_kernel(const double* __restrict__ cb, size_t n, double* __restrict__ xc64, float* __restrict__ xc32) {
    constexpr int NN = D * D, DB = D, J = 1;
    using C = CB<D>;
    using L = XC<D>;
    __shared__ double sAKHA[NN], sA[NN], sK[D], sHA[D], sV1[D], sG[kChunkX * D];
    const size_t l = blockIdx.x;
    const int lane = threadIdx.x;
    if (l >= n) return;
    const double* __restrict__ c = cb + l * C::SIZE;
    for (int e = lane; e < NN; e += 64) { sAKHA[e] = c[C::AKHA + e]; sA[e] = c[C::A + e]; }
    if (lane < D) { sK[lane] = c[C::K + lane]; sHA[lane] = c[C::HA + lane]; sV1[lane] = c[C::K + lane]; }
    lds_sync();
    bool ok = true, ok32 = true;
    for (int k = kChunkX - 1; k >= 0; k--) {                           // g_k = AKHA^(31-k) K, as stack_update_kernel forms it
        if (lane < D) { sG[k * D + lane] = sV1[lane]; ok = ok && (fabs(sV1[lane]) < 1e150); ok32 = ok32 && (fabs(sV1[lane]) < 1e18); }
        wmv<D>(sAKHA, sV1, sV1, lane);
    }
    // scan powers M^(2^k), M = AKHA^32, and the levels that matter per precision, as stack_update_kernel derives them
    __shared__ double sT1[NN], sSP[6 * L::LS];
    for (int e = lane; e < NN; e += 64) sT1[e] = sAKHA[e];
    lds_sync();
    for (int q = 1; q < kChunkX; q <<= 1) wmm<D>(sT1, sT1, sT1, lane);
    int nlev64 = 1, nlev32 = 1;
    for (int lv = 0; lv < 6; lv++) {
        double big = 0.0;
        for (int e = lane; e < L::LS; e += 64) {
            const double v = e < NN ? sT1[e] : 0.0;
            sSP[lv * L::LS + e] = v; ok = ok && (fabs(v) < 1e150); ok32 = ok32 && (fabs(v) < 1e18); big = fmax(big, fabs(v));
        }
        for (int o = 32; o >= 1; o >>= 1) big = fmax(big, __shfl_xor(big, o, 64));
        if (big * D >= 1e-20) nlev64 = lv + 1;
        if (big * D >= 1e-10) nlev32 = lv + 1;
        wmm<D>(sT1, sT1, sT1, lane);
    }
    ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
    ok32 = __builtin_amdgcn_ballot_w64(!ok32) == 0;
    double* o64 = xc64 + l * L::SIZE;
    float* o32 = xc32 + l * L::SIZE;
    for (int e = lane; e < L::SIZE; e += 64) {                         // every entry of the block exactly once
        double v = 0.0, v32 = 0.0;
        bool split = false;
        if (e < L::K) v = sAKHA[e - L::AKHA];
        else if (e < L::A) v = sK[e - L::K];
        else if (e < L::HA) v = sA[e - L::A];
        else if (e < L::S) v = sHA[e - L::HA];
        else if (e == L::S) v = c[C::S];
        else if (e == L::LOGS) v = c[C::LOGS];
        else if (e == L::SCANOK) { v = ok ? 1.0 : 0.0; v32 = ok32 ? 1.0 : 0.0; split = true; }
        else if (e == L::NLEV) { v = (double)nlev64; v32 = (double)nlev32; split = true; }
        else if (e >= L::SP && e < L::PK) v = sSP[e - L::SP];
        else if (e >= L::AB && e < L::HA16) { const int i = e - L::AB; v = i < DB * DB ? sA[i] : 0.0; }
        else if (e >= L::HA16 && e < L::K16) { const int i = e - L::HA16; v = i < D ? sHA[i] : 0.0; }
        else if (e >= L::K16 && e < L::G) { const int i = e - L::K16; v = i < D ? sK[i] : 0.0; }
        else if (e >= L::G && e < L::SP) { const int i = e - L::G; v = i < kChunkX * D ? sG[i] : 0.0; }
        else if (e >= L::PK) {                                         // coefficient pairs [A_rq | HA_q | K_r] of the one block, paired with zeros
            constexpr int PB = DB * DB + 2 * DB;
            const int ee = e - L::PK, pr = ee / 2, half = ee % 2;
            if (half == 0 && pr < PB) v = pr < DB * DB ? sA[pr] : (pr < DB * DB + DB ? sHA[pr - DB * DB] : sK[pr - DB * DB - DB]);
        }
        o64[e] = v;
        o32[e] = (float)(split ? v32 : v);
    }
}

template <int DB, int J>
void launch_t(double dt, const double* params, size_t n, double* cb64, float* cb32, double* cbd64, int* n_unstable, int* rescue_idx, hipStream_t s) {
    hipLaunchKernelGGL((stack_update_kernel<DB, J>), dim3((unsigned)n), dim3(64), 0, s, dt, params, n, cb64, cb32, cbd64, n_unstable, rescue_idx);
}

}  // namespace

void launch_xc_from_cb(int d, const double* cb64, size_t n, double* xc64, float* xc32, hipStream_t stream) {
    if (n == 0) return;
    if (d == 2) hipLaunchKernelGGL(xc_from_cb_kernel<2>, dim3((unsigned)n), dim3(64), 0, stream, cb64, n, xc64, xc32);
    else hipLaunchKernelGGL(xc_from_cb_kernel<3>, dim3((unsigned)n), dim3(64), 0, stream, cb64, n, xc64, xc32);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

void launch_stack_update(int kernel, double dt, const double* params_dev, size_t n, double* cb64, float* cb32, double* cbd64,
                         int* n_unstable /* int[4] */, int* rescue_idx /* int[n] or NULL */, hipStream_t stream) {
    if (n == 0) return;
    MOIHGP_HIP_FATAL(hipMemsetAsync(n_unstable, 0, 4 * sizeof(int), stream));
    const int base = kernel_base(kernel), J = kernel_stack(kernel);
    if (base == 0) {
        if (J == 2) launch_t<2, 2>(dt, params_dev, n, cb64, cb32, cbd64, n_unstable, rescue_idx, stream);
        else if (J == 3) launch_t<2, 3>(dt, params_dev, n, cb64, cb32, cbd64, n_unstable, rescue_idx, stream);
        else launch_t<2, 4>(dt, params_dev, n, cb64, cb32, cbd64, n_unstable, rescue_idx, stream);
    } else {
        if (J == 2) launch_t<3, 2>(dt, params_dev, n, cb64, cb32, cbd64, n_unstable, rescue_idx, stream);
        else if (J == 3) launch_t<3, 3>(dt, params_dev, n, cb64, cb32, cbd64, n_unstable, rescue_idx, stream);
        else launch_t<3, 4>(dt, params_dev, n, cb64, cb32, cbd64, n_unstable, rescue_idx, stream);
    }
    MOIHGP_HIP_FATAL(hipGetLastError());
}

}  // namespace moihgp
