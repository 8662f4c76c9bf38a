// stationary.hip -- batched IHGP::update on the device: 16 latents per 64-lane workgroup, fp64 always.
//
// Restates, for every latent independently, reference include/moihgp/ihgp.h:117-201:
//   A = expm(dt F); Q = sym(Pinf - A Pinf A^T); PP = DARE(A, H^T, Q, R); S, K, HA, AKHA;
//   per hyper-parameter p: dA[p] (2d x 2d block expm), dQ, QLyap (four cases on exact zeros),
//   dPP = DLyap(A - A K H, QLyap), dS, dK, dAKHA, HdA.
// with the state-space models of matern32ss.h:40-64 / matern52ss.h:38-75 and the fixed-point
// solvers of include/utils/dare.h:10-58 (tolerance 1e-8, at most 100 iterations, literal forms).
// The matrix exponential follows the published algorithm behind Eigen's MatrixBase::exp()
// (Higham 2005: Pade [3/3]..[13/13] chosen on the 1-norm, scaling and squaring).
//
// This kernel is launch-latency sized work (n latents x ~3e4 fp64 flops); it is written for
// fidelity to the reference's operation order, not for speed: FP contraction is off so that the
// iteration counts of DARE/DLyap agree with a plain-C evaluation.
#include "common.h"

#pragma clang fp contract(off)
#include "stationary_common.h"

namespace moihgp {
namespace {

template <int D>
struct SS {
    double F[D * D], Pinf[D * D], H[D], R;
    double dF[3][D * D], dPinf[3][D * D], dR[3];
};

// matern32ss.h:40-64 (D == 2) and matern52ss.h:38-75 (D == 3), including `lam = sqrt(3)/l` there.
template <int D>
__device__ void ss_build(int kernel, const double* params, SS<D>& s) {
    for (int i = 0; i < D * D; i++) {
        s.F[i] = 0.0; s.Pinf[i] = 0.0;
        for (int p = 0; p < 3; p++) { s.dF[p][i] = 0.0; s.dPinf[p][i] = 0.0; }
    }
    for (int i = 0; i < D; i++) s.H[i] = 0.0;
    s.H[0] = 1.0;
    double magnitude = params[0], lengthscale = params[1];
    s.R = params[2];
    s.dR[0] = 0.0; s.dR[1] = 0.0; s.dR[2] = 1.0;
    if constexpr (D == 2) {
        double lam = sqrt(3.0) / lengthscale, lam2 = lam * lam;
        double len3 = 6.0 / (lengthscale * lengthscale * lengthscale);
        s.F[1] = 1.0;
        s.F[2] = -lam2;
        s.F[3] = -2.0 * lam;
        s.Pinf[0] = magnitude;
        s.Pinf[3] = magnitude * lam2;
        s.dF[1][2] = len3;
        s.dF[1][3] = 2.0 * lam / lengthscale;
        s.dPinf[0][0] = 1.0;
        s.dPinf[0][3] = lam2;
        s.dPinf[1][3] = -magnitude * len3;
    } else {
        double lam = sqrt(3.0) / lengthscale;
        double lam2 = lam * lam, len2 = lengthscale * lengthscale, len3 = len2 * lengthscale, len4 = len2 * len2;
        double kappa = 5.0 / 3.0 * magnitude / len2, kappa2 = -2.0 * kappa / lengthscale, sq5 = sqrt(5.0);
        s.F[1] = 1.0; s.F[5] = 1.0;
        s.F[6] = -lam2 * lam; s.F[7] = -3.0 * lam2; s.F[8] = -3.0 * lam;
        s.Pinf[0] = magnitude; s.Pinf[8] = 25.0 * magnitude / len4; s.Pinf[4] = kappa;
        s.Pinf[6] = -kappa; s.Pinf[2] = -kappa;
        s.dF[1][6] = 15.0 * sq5 / len4; s.dF[1][7] = 30.0 / len3; s.dF[1][8] = sq5 * lam2;
        for (int i = 0; i < 9; i++) s.dPinf[0][i] = s.Pinf[i] / magnitude;
        s.dPinf[1][4] = kappa2; s.dPinf[1][6] = -kappa2; s.dPinf[1][2] = -kappa2;
        s.dPinf[1][8] = -100.0 * magnitude / len2 / len3;
    }
    (void)kernel;
}

// One workgroup of four wavefronts updates kLPB = 16 latents (lane = latent) in three phases, so that the independent fixed-point
// chains of a latent run side by side instead of one after the other (the 100-iteration loops are latency chains).  The roles
// are split by WAVEFRONT, not by lane: lanes of one wavefront that take different branches run them one after the other, which is
// what the first version did (16 latents per 64-lane block, roles by lane group: 0.27 ms for any latent count up to 4096).
//   phase 1   wave 0: A, Q, DARE, S, K, HA, AKHA (ihgp.h:120-133)     wave 1: the 2d x 2d block expm of the one
//             hyper-parameter with dF != 0, which does not depend on the DARE (ihgp.h:163-167)
//   phase 2   waves 0..2: one hyper-parameter each: dQ, QLyap, DLyap, dS, dK, dAKHA, HdA (ihgp.h:136-200)
//   phase 3   waves 0..1: the scan tables of recursion.hip, one stream precision each
// Every quantity is computed by exactly the expressions of the one-lane evaluation, in the same order.
// kLPB latents per workgroup (lanes 0..kLPB-1 of the role waves): 16 for up to a few thousand latents (one workgroup per CU, phase 3
// in two rounds), 64 beyond (four times fewer workgroups: 65536 latents 0.89 instead of 2.05 ms, 4096 latents 0.213 instead of 0.205).

template <int D>
struct Shared {               // per latent, in LDS
    double A[D * D], PP[D * D], AAKH[D * D], AKHA[D * D], dA1[D * D];
    double K[D], HA[D], AK[D], PPHt[D], HPP[D];
    double S;
};

template <int D, int kLPB>
__global__ void __launch_bounds__(256) ihgp_update_kernel(int kernel, double dt, const double* __restrict__ params,
                                                         size_t n, double* __restrict__ cb64, float* __restrict__ cb32,
                                                         int* __restrict__ n_unstable) {
    using L = CB<D>;
    constexpr int P = kNumIgpParam, NN = D * D;
    __shared__ Shared<D> sh[kLPB];
    const int role = threadIdx.x >> 6, lane_l = threadIdx.x & 63;      // role: wave-uniform; lane_l: the latent within the block
    const size_t l0 = (size_t)blockIdx.x * kLPB;

    // ---- phase 1 ----
    if (role < 2 && lane_l < kLPB) {
        const int li = lane_l;
        const size_t l = l0 + li;
        if (l < n) {
            double prm[3] = {params[l * 3 + 0], params[l * 3 + 1], params[l * 3 + 2]};
            SS<D> s;
            ss_build<D>(kernel, prm, s);
            Shared<D>& q = sh[li];
            if (role == 0) {
                double* o64 = cb64 + l * L::SIZE;
                float* o32 = cb32 + l * L::SIZE;
                auto put = [&](int off, double v) { o64[off] = v; o32[off] = (float)v; };
                double A[NN], AT[NN], T1[NN], T2[NN], Q[NN], PP[NN];
                for (int i = 0; i < NN; i++) T1[i] = dt * s.F[i];
                expm<D>(T1, A);                                                    // ihgp.h:120
                mt<D>(A, AT);
                mm<D>(A, s.Pinf, T1); mm<D>(T1, AT, T2);
                for (int i = 0; i < NN; i++) T1[i] = s.Pinf[i] - T2[i];             // ihgp.h:121
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) Q[i * D + j] = (T1[i * D + j] + T1[j * D + i]) / 2.0;   // ihgp.h:122
                int dare_iters = dare<D>(A, s.H, Q, s.R, PP);                       // ihgp.h:125
                double PPHt[D], HPP[D], K[D], HA[D], AK[D];
                mv<D>(PP, s.H, PPHt);
                double S = s.R;
                for (int i = 0; i < D; i++) S += s.H[i] * PPHt[i];                  // ihgp.h:126
                for (int i = 0; i < D; i++) K[i] = PPHt[i] / S;                     // ihgp.h:127
                for (int j = 0; j < D; j++) { double t = 0.0; for (int i = 0; i < D; i++) t += s.H[i] * PP[i * D + j]; HPP[j] = t; }
                for (int j = 0; j < D; j++) { double t = 0.0; for (int i = 0; i < D; i++) t += s.H[i] * A[i * D + j]; HA[j] = t; }   // ihgp.h:129
                mv<D>(A, K, AK);                                                    // ihgp.h:132
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) {
                        q.AKHA[i * D + j] = A[i * D + j] - K[i] * HA[j];            // ihgp.h:130
                        q.AAKH[i * D + j] = A[i * D + j] - AK[i] * s.H[j];          // ihgp.h:133
                        q.A[i * D + j] = A[i * D + j];
                        q.PP[i * D + j] = PP[i * D + j];
                    }
                for (int i = 0; i < D; i++) { q.K[i] = K[i]; q.HA[i] = HA[i]; q.AK[i] = AK[i]; q.PPHt[i] = PPHt[i]; q.HPP[i] = HPP[i]; }
                q.S = S;
                for (int i = 0; i < NN; i++) { put(L::AKHA + i, q.AKHA[i]); put(L::A + i, A[i]); }
                for (int i = 0; i < D; i++) { put(L::K + i, K[i]); put(L::HA + i, HA[i]); }
                put(L::S, S);
                put(L::LOGS, log(S));
                put(L::ITERS, (double)dare_iters);
                for (int p = 0; p < P; p++) put(L::PARAMS + p, prm[p]);
            } else {
                // the lengthscale is the only hyper-parameter with dF != 0 in both models (matern32ss.h:57-58, matern52ss.h:62-64)
                // exp of dt [[F, 0], [dF, F]] (ihgp.h:163-166), evaluated on its blocks; dA is the lower left one (ihgp.h:167)
                double X[NN], Y[NN], EX[NN], EY[NN];
                for (int i = 0; i < NN; i++) { X[i] = dt * s.F[i]; Y[i] = dt * s.dF[1][i]; }
                expm_blt<D>(X, Y, EX, EY);
                for (int i = 0; i < NN; i++) q.dA1[i] = EY[i];
            }
        }
    }
    __syncthreads();

    // ---- phase 2 ----
    if (role < P && lane_l < kLPB) {
        const int li = lane_l, p = role;
        const size_t l = l0 + li;
        if (l < n) {
            double prm[3] = {params[l * 3 + 0], params[l * 3 + 1], params[l * 3 + 2]};
            SS<D> s;
            ss_build<D>(kernel, prm, s);
            const Shared<D>& q = sh[li];
            double* o64 = cb64 + l * L::SIZE;
            float* o32 = cb32 + l * L::SIZE;
            auto put = [&](int off, double v) { o64[off] = v; o32[off] = (float)v; };
            double A[NN], AT[NN], PP[NN], AAKH[NN], T1[NN], T2[NN];
            for (int i = 0; i < NN; i++) { A[i] = q.A[i]; PP[i] = q.PP[i]; AAKH[i] = q.AAKH[i]; }
            mt<D>(A, AT);
            const double S = q.S;
            double dA[NN], dAT[NN], dQ[NN], QL[NN], dPP[NN];
            const bool dF_zero = all_zero<D>(s.dF[p]), dPinf_zero = all_zero<D>(s.dPinf[p]), dR_zero = (s.dR[p] == 0.0);
            if (dF_zero) {                                                  // ihgp.h:141
                for (int i = 0; i < NN; i++) dA[i] = 0.0;
                if (dPinf_zero) {
                    for (int i = 0; i < NN; i++) dQ[i] = 0.0;
                } else {
                    mm<D>(A, s.dPinf[p], T1); mm<D>(T1, AT, T2);
                    for (int i = 0; i < NN; i++) dQ[i] = s.dPinf[p][i] - T2[i];   // ihgp.h:150
                }
                if (dR_zero) {
                    for (int i = 0; i < NN; i++) QL[i] = dQ[i];             // ihgp.h:154
                } else {
                    // ihgp.h:158 is `AK * AK^T * dR` ((d x d) * (1 x 1), an invalid Eigen product);
                    // evaluated with its evident meaning AK dR AK^T, as ihgp.h:183 writes it.
                    for (int i = 0; i < D; i++)
                        for (int j = 0; j < D; j++) QL[i * D + j] = q.AK[i] * s.dR[p] * q.AK[j] + dQ[i * D + j];
                }
            } else {
                for (int i = 0; i < NN; i++) dA[i] = q.dA1[i];              // ihgp.h:167 (phase 1; dF != 0 only for p == 1)
                mt<D>(dA, dAT);
                double dAPAt[NN], APdAt[NN];
                mm<D>(dA, s.Pinf, T1); mm<D>(T1, AT, dAPAt);
                mm<D>(A, s.Pinf, T1); mm<D>(T1, dAT, APdAt);
                if (dPinf_zero) {
                    for (int i = 0; i < NN; i++) dQ[i] = -dAPAt[i] - APdAt[i];   // ihgp.h:171
                } else {
                    mm<D>(A, s.dPinf[p], T1); mm<D>(T1, AT, T2);
                    for (int i = 0; i < NN; i++) dQ[i] = s.dPinf[p][i] - dAPAt[i] - T2[i] - APdAt[i];   // ihgp.h:175
                }
                double t1[NN], t2[NN], dAPPHt[D], HPPdAT[D];
                mm<D>(dA, PP, T1); mm<D>(T1, AT, t1);
                mm<D>(A, PP, T1); mm<D>(T1, dAT, t2);
                double PPHt[D];
                for (int i = 0; i < D; i++) PPHt[i] = q.PPHt[i];
                mv<D>(dA, PPHt, dAPPHt);
                for (int j = 0; j < D; j++) { double t = 0.0; for (int k = 0; k < D; k++) t += q.HPP[k] * dAT[k * D + j]; HPPdAT[j] = t; }
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) {                           // ihgp.h:179 / :183
                        double v = t1[i * D + j] + t2[i * D + j] - dAPPHt[i] * q.AK[j] - q.AK[i] * HPPdAT[j];
                        if (!dR_zero) v += q.AK[i] * s.dR[p] * q.AK[j];
                        QL[i * D + j] = v + dQ[i * D + j];
                    }
            }
            int its = dlyap<D>(AAKH, QL, dPP);                               // ihgp.h:187
            double dS = s.dR[p];
            for (int i = 0; i < D; i++)
                for (int j = 0; j < D; j++) dS += s.H[i] * dPP[i * D + j] * s.H[j];   // ihgp.h:188
            double dK[D];
            for (int i = 0; i < D; i++) {                                    // ihgp.h:189
                double t = 0.0;
                for (int j = 0; j < D; j++) t += (dPP[i * D + j] - PP[i * D + j] * dS / S) * s.H[j];
                dK[i] = t / S;
            }
            put(L::DS + p, dS);
            put(L::ITERS + 1 + p, (double)its);
            for (int i = 0; i < D; i++) put(L::DK + p * D + i, dK[i]);
            for (int i = 0; i < NN; i++) put(L::DA + p * NN + i, dA[i]);
            if (dF_zero) {                                                   // ihgp.h:192-193
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) put(L::DAKHA + p * NN + i * D + j, -dK[i] * q.HA[j]);
                for (int i = 0; i < D; i++) put(L::HDA + p * D + i, 0.0);
            } else {                                                         // ihgp.h:197-198
                double HdA[D];
                for (int j = 0; j < D; j++) { double t = 0.0; for (int i = 0; i < D; i++) t += s.H[i] * dA[i * D + j]; HdA[j] = t; }
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) put(L::DAKHA + p * NN + i * D + j, dA[i * D + j] - dK[i] * q.HA[j] - q.K[i] * HdA[j]);
                for (int i = 0; i < D; i++) put(L::HDA + p * D + i, HdA[i]);
            }
        }
    }

    // ---- phase 3: tables of the segment solve (recursion.hip), one set per stream dtype because the chunk length differs.
    // Sixteen lanes per (latent, precision): lane r builds PJ[r] = M^(r+1), row r of the response table g_r = AKHA^(ck-1-r) K and,
    // for r < 4, SP[r] = M^(2^r), each from the binary powers AKHA^(1,2,4,8) and M^(1,2,4,8,16) (M = AKHA^ck) it forms itself;
    // the sixteen lanes write one contiguous run of the block.  (One lane per pair walking all 16 powers and storing 250 scalars
    // 2.9 KB apart from its neighbours took 76 of the kernel's 220 us.)  These tables are internal to this library: no order of
    // evaluation to keep.
    for (int item = threadIdx.x; item < kLPB * 2 * 16; item += 256) {
        const int r = item & 15, pass = (item >> 4) & 1, li = item >> 5;
        const size_t l = l0 + li;
        if (l >= n) continue;                                        // (whole 16-lane groups leave together)
        const Shared<D>& q = sh[li];
        const int ck = pass == 0 ? kChunk64 : kChunk32;
        double ak[4][NN], mp[5][NN];
        for (int i = 0; i < NN; i++) ak[0][i] = q.AKHA[i];
        for (int b = 1; b < 4; b++) mm<D>(ak[b - 1], ak[b - 1], ak[b]);          // AKHA^(2,4,8)
        if (ck == 8) { for (int i = 0; i < NN; i++) mp[0][i] = ak[3][i]; }
        else mm<D>(ak[3], ak[3], mp[0]);                                         // M = AKHA^ck, ck in {8, 16}
        for (int b = 1; b < 5; b++) mm<D>(mp[b - 1], mp[b - 1], mp[b]);          // M^(2,4,8,16)
        double pj[NN], g[D];
        for (int i = 0; i < NN; i++) pj[i] = (i % (D + 1) == 0) ? 1.0 : 0.0;
        for (int b = 0; b < 5; b++) {
            double t[NN];
            mm<D>(mp[b], pj, t);
            const bool take = ((r + 1) >> b) & 1;
            for (int i = 0; i < NN; i++) pj[i] = take ? t[i] : pj[i];
        }
        const int e = ck - 1 - r;                                    // exponent of this lane's row of the response table (< 0: no row)
        for (int i = 0; i < D; i++) g[i] = q.K[i];
        for (int b = 0; b < 4; b++) {
            double t[D];
            mv<D>(ak[b], g, t);
            const bool take = e >= 0 && ((e >> b) & 1);
            for (int i = 0; i < D; i++) g[i] = take ? t[i] : g[i];
        }
        if (e < 0) for (int i = 0; i < D; i++) g[i] = 0.0;
        // scan usable?  every entry finite and far from the overflow threshold of the block's precision
        const double lim = pass == 0 ? 1e150 : 1e18;
        bool ok = true;
        for (int i = 0; i < NN; i++) ok = ok && (fabs(pj[i]) < lim);                 // false for NaN too
        for (int i = 0; i < D; i++) ok = ok && (fabs(g[i]) < lim);
        double spv[NN];                                              // this lane's scan power M^(2^r), r < 4 (selected without indexing by r)
        for (int i = 0; i < NN; i++) spv[i] = r == 0 ? mp[0][i] : (r == 1 ? mp[1][i] : (r == 2 ? mp[2][i] : mp[3][i]));
        if (r < 4) for (int i = 0; i < NN; i++) ok = ok && (fabs(spv[i]) < lim);
        const unsigned long long bal = __ballot(ok);
        const int grp = (threadIdx.x & 63) >> 4;
        const bool all_ok = ((bal >> (16 * grp)) & 0xFFFFull) == 0xFFFFull;
        double* o64 = cb64 + l * L::SIZE;
        float* o32 = cb32 + l * L::SIZE;
        auto put = [&](int off, double v) { if (pass == 0) o64[off] = v; else o32[off] = (float)v; };
        for (int i = 0; i < D; i++) put(L::G + r * D + i, g[i]);
        for (int i = 0; i < NN; i++) put(L::PJ + r * NN + i, pj[i]);
        if (r < 4) for (int i = 0; i < NN; i++) put(L::SP + r * NN + i, spv[i]);
        if (r == 0) {
            for (int i = L::SCANOK; i < L::SIZE; i++) put(i, 0.0);
            put(L::SCANOK, all_ok ? 1.0 : 0.0);
            if (!all_ok) atomicAdd(&n_unstable[pass], 1);
        }
    }
}

}  // namespace

void launch_ihgp_update(int kernel, int d, double dt, const double* params_dev, size_t n, double* cb64, float* cb32,
                        int* n_unstable, hipStream_t stream) {
    if (n == 0) return;
    MOIHGP_HIP_FATAL(hipMemsetAsync(n_unstable, 0, 2 * sizeof(int), stream));
    const bool many = n > 8192;
    dim3 block(256), grid((unsigned)((n + (many ? 64 : 16) - 1) / (many ? 64 : 16)));
    if (d == 2) {
        if (many) hipLaunchKernelGGL((ihgp_update_kernel<2, 64>), grid, block, 0, stream, kernel, dt, params_dev, n, cb64, cb32, n_unstable);
        else hipLaunchKernelGGL((ihgp_update_kernel<2, 16>), grid, block, 0, stream, kernel, dt, params_dev, n, cb64, cb32, n_unstable);
    } else {
        if (many) hipLaunchKernelGGL((ihgp_update_kernel<3, 64>), grid, block, 0, stream, kernel, dt, params_dev, n, cb64, cb32, n_unstable);
        else hipLaunchKernelGGL((ihgp_update_kernel<3, 16>), grid, block, 0, stream, kernel, dt, params_dev, n, cb64, cb32, n_unstable);
    }
    MOIHGP_HIP_FATAL(hipGetLastError());
}

}  // namespace moihgp
