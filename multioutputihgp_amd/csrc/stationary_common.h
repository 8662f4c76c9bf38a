// stationary_common.h -- small dense helpers shared by stationary.hip and stationary_x.hip (one lane per latent, fp64):
// matrix products, Eigen-style Pade matrix exponential, and the literal fixed-point solvers of the reference's utils/dare.h.
// Include AFTER `#pragma clang fp contract(off)`: the iteration counts of DARE/DLyap must agree with a plain-C evaluation.
#pragma once
#include "common.h"

namespace moihgp {
namespace {

template <int N>
__device__ inline void mm(const double* A, const double* B, double* C) {
    double T[N * N];
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            double s = 0.0;
            for (int k = 0; k < N; k++) s += A[i * N + k] * B[k * N + j];
            T[i * N + j] = s;
        }
    for (int i = 0; i < N * N; i++) C[i] = T[i];
}
template <int N>
__device__ inline void mt(const double* A, double* At) {
    double T[N * N];
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) T[j * N + i] = A[i * N + j];
    for (int i = 0; i < N * N; i++) At[i] = T[i];
}
template <int N>
__device__ inline void mv(const double* A, const double* x, double* y) {
    double T[N];
    for (int i = 0; i < N; i++) {
        double s = 0.0;
        for (int k = 0; k < N; k++) s += A[i * N + k] * x[k];
        T[i] = s;
    }
    for (int i = 0; i < N; i++) y[i] = T[i];
}
template <int N>
__device__ inline bool all_zero(const double* A) {
    for (int i = 0; i < N * N; i++)
        if (A[i] != 0.0) return false;
    return true;
}

// X = Den^-1 Num, partial pivoting (Eigen partialPivLu().solve)
template <int N>
__device__ void lu_solve(const double* Ain, const double* Bin, double* X) {
    double A[N * N], B[N * N];
    for (int i = 0; i < N * N; i++) { A[i] = Ain[i]; B[i] = Bin[i]; }
    for (int k = 0; k < N; k++) {
        int p = k;
        double best = fabs(A[k * N + k]);
        for (int i = k + 1; i < N; i++)
            if (fabs(A[i * N + k]) > best) { best = fabs(A[i * N + k]); p = i; }
        if (p != k)
            for (int j = 0; j < N; j++) {
                double t = A[k * N + j]; A[k * N + j] = A[p * N + j]; A[p * N + j] = t;
                t = B[k * N + j]; B[k * N + j] = B[p * N + j]; B[p * N + j] = t;
            }
        for (int i = k + 1; i < N; i++) {
            double f = A[i * N + k] / A[k * N + k];
            for (int j = k; j < N; j++) A[i * N + j] -= f * A[k * N + j];
            for (int j = 0; j < N; j++) B[i * N + j] -= f * B[k * N + j];
        }
    }
    for (int j = 0; j < N; j++)
        for (int i = N - 1; i >= 0; i--) {
            double s = B[i * N + j];
            for (int k = i + 1; k < N; k++) s -= A[i * N + k] * X[k * N + j];
            X[i * N + j] = s / A[i * N + i];
        }
}

// E = exp(Ain): Pade approximant of degree 3/5/7/9/13 by 1-norm, scaling & squaring for the last.
template <int N>
__device__ void expm(const double* Ain, double* E) {
    constexpr int NN = N * N;
    double A[NN], A2[NN], A4[NN], A6[NN], U[NN], V[NN], T[NN];
    double l1 = 0.0;
    for (int j = 0; j < N; j++) {
        double s = 0.0;
        for (int i = 0; i < N; i++) s += fabs(Ain[i * N + j]);
        if (s > l1) l1 = s;
    }
    for (int i = 0; i < NN; i++) A[i] = Ain[i];
    int squarings = 0;
    if (l1 < 1.495585217958292e-002) {
        const double b[] = {120., 60., 12., 1.};
        mm<N>(A, A, A2);
        for (int i = 0; i < NN; i++) T[i] = b[3] * A2[i];
        for (int i = 0; i < N; i++) T[i * N + i] += b[1];
        mm<N>(A, T, U);
        for (int i = 0; i < NN; i++) V[i] = b[2] * A2[i];
        for (int i = 0; i < N; i++) V[i * N + i] += b[0];
    } else if (l1 < 2.539398330063230e-001) {
        const double b[] = {30240., 15120., 3360., 420., 30., 1.};
        mm<N>(A, A, A2); mm<N>(A2, A2, A4);
        for (int i = 0; i < NN; i++) T[i] = b[5] * A4[i] + b[3] * A2[i];
        for (int i = 0; i < N; i++) T[i * N + i] += b[1];
        mm<N>(A, T, U);
        for (int i = 0; i < NN; i++) V[i] = b[4] * A4[i] + b[2] * A2[i];
        for (int i = 0; i < N; i++) V[i * N + i] += b[0];
    } else if (l1 < 9.504178996162932e-001) {
        const double b[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
        mm<N>(A, A, A2); mm<N>(A2, A2, A4); mm<N>(A4, A2, A6);
        for (int i = 0; i < NN; i++) T[i] = b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
        for (int i = 0; i < N; i++) T[i * N + i] += b[1];
        mm<N>(A, T, U);
        for (int i = 0; i < NN; i++) V[i] = b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
        for (int i = 0; i < N; i++) V[i * N + i] += b[0];
    } else if (l1 < 2.097847961257068e+000) {
        const double b[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.};
        double A8[NN];
        mm<N>(A, A, A2); mm<N>(A2, A2, A4); mm<N>(A4, A2, A6); mm<N>(A6, A2, A8);
        for (int i = 0; i < NN; i++) T[i] = b[9] * A8[i] + b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
        for (int i = 0; i < N; i++) T[i * N + i] += b[1];
        mm<N>(A, T, U);
        for (int i = 0; i < NN; i++) V[i] = b[8] * A8[i] + b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
        for (int i = 0; i < N; i++) V[i * N + i] += b[0];
    } else {
        const double b[] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                            129060195264000., 10559470521600., 670442572800., 33522128640., 1323241920.,
                            40840800., 960960., 16380., 182., 1.};
        const double maxnorm = 5.371920351148152;
        frexp(l1 / maxnorm, &squarings);
        if (squarings < 0) squarings = 0;
        double sc = ldexp(1.0, -squarings);
        for (int i = 0; i < NN; i++) A[i] *= sc;
        mm<N>(A, A, A2); mm<N>(A2, A2, A4); mm<N>(A4, A2, A6);
        for (int i = 0; i < NN; i++) V[i] = b[13] * A6[i] + b[11] * A4[i] + b[9] * A2[i];
        mm<N>(A6, V, T);
        for (int i = 0; i < NN; i++) T[i] += b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
        for (int i = 0; i < N; i++) T[i * N + i] += b[1];
        mm<N>(A, T, U);
        for (int i = 0; i < NN; i++) T[i] = b[12] * A6[i] + b[10] * A4[i] + b[8] * A2[i];
        mm<N>(A6, T, V);
        for (int i = 0; i < NN; i++) V[i] += b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
        for (int i = 0; i < N; i++) V[i * N + i] += b[0];
    }
    for (int i = 0; i < NN; i++) { double u = U[i], v = V[i]; A2[i] = u + v; A4[i] = -u + v; }
    lu_solve<N>(A4, A2, E);
    for (int s = 0; s < squarings; s++) mm<N>(E, E, E);
}

// exp of the block lower triangular matrix [[X, 0], [Y, X]] (2N x 2N), returned as its blocks [[EX, 0], [EY, EX]]: the matrix
// ihgp.h:163-167 exponentiates to get dA = EY.  Same algorithm as expm<2N> above (degree by the 1-norm of the whole matrix, the
// same Pade coefficients, scaling and squaring), carried out on the blocks: a product of two such matrices is three N x N
// products instead of eight, and the linear solve is two N x N solves.  Equal to the dense evaluation up to rounding (the dense
// LU may pivot across the blocks); a third of its instructions, which is what this straight-line code is bound by.
template <int N>
struct BltPair { double x[N * N], y[N * N]; };
template <int N>
__device__ inline void blt_mul(const BltPair<N>& a, const BltPair<N>& b, BltPair<N>& c) {
    double t1[N * N], t2[N * N], tx[N * N];
    mm<N>(a.x, b.x, tx);
    mm<N>(a.y, b.x, t1);
    mm<N>(a.x, b.y, t2);
    for (int i = 0; i < N * N; i++) { c.x[i] = tx[i]; c.y[i] = t1[i] + t2[i]; }
}
template <int N>
__device__ void expm_blt(const double* X, const double* Y, double* EX, double* EY) {
    constexpr int NN = N * N;
    BltPair<N> A, A2, A4, A6, U, V, T;
    double l1 = 0.0;
    for (int j = 0; j < N; j++) {                      // columns j < N carry both blocks: they bound the others
        double s = 0.0;
        for (int i = 0; i < N; i++) s += fabs(X[i * N + j]);
        for (int i = 0; i < N; i++) s += fabs(Y[i * N + j]);
        if (s > l1) l1 = s;
    }
    for (int i = 0; i < NN; i++) { A.x[i] = X[i]; A.y[i] = Y[i]; }
    int squarings = 0;
    // T = sum_k c_k P_k (+ d I), on both blocks
    auto comb = [&](BltPair<N>& out, double c3, const BltPair<N>* p3, double c2, const BltPair<N>* p2, double c1, const BltPair<N>* p1, double diag, bool accumulate) {
        for (int i = 0; i < NN; i++) {
            double vx = accumulate ? out.x[i] : 0.0, vy = accumulate ? out.y[i] : 0.0;
            if (p3) { vx += c3 * p3->x[i]; vy += c3 * p3->y[i]; }
            if (p2) { vx += c2 * p2->x[i]; vy += c2 * p2->y[i]; }
            if (p1) { vx += c1 * p1->x[i]; vy += c1 * p1->y[i]; }
            out.x[i] = vx; out.y[i] = vy;
        }
        for (int i = 0; i < N; i++) out.x[i * N + i] += diag;
    };
    if (l1 < 1.495585217958292e-002) {
        const double b[] = {120., 60., 12., 1.};
        blt_mul<N>(A, A, A2);
        comb(T, 0, nullptr, 0, nullptr, b[3], &A2, b[1], false);
        blt_mul<N>(A, T, U);
        comb(V, 0, nullptr, 0, nullptr, b[2], &A2, b[0], false);
    } else if (l1 < 2.539398330063230e-001) {
        const double b[] = {30240., 15120., 3360., 420., 30., 1.};
        blt_mul<N>(A, A, A2); blt_mul<N>(A2, A2, A4);
        comb(T, 0, nullptr, b[5], &A4, b[3], &A2, b[1], false);
        blt_mul<N>(A, T, U);
        comb(V, 0, nullptr, b[4], &A4, b[2], &A2, b[0], false);
    } else if (l1 < 9.504178996162932e-001) {
        const double b[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
        blt_mul<N>(A, A, A2); blt_mul<N>(A2, A2, A4); blt_mul<N>(A4, A2, A6);
        comb(T, b[7], &A6, b[5], &A4, b[3], &A2, b[1], false);
        blt_mul<N>(A, T, U);
        comb(V, b[6], &A6, b[4], &A4, b[2], &A2, b[0], false);
    } else if (l1 < 2.097847961257068e+000) {
        const double b[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.};
        BltPair<N> A8;
        blt_mul<N>(A, A, A2); blt_mul<N>(A2, A2, A4); blt_mul<N>(A4, A2, A6); blt_mul<N>(A6, A2, A8);
        comb(T, b[7], &A6, b[5], &A4, b[3], &A2, 0.0, false);
        comb(T, 0, nullptr, 0, nullptr, b[9], &A8, b[1], true);
        blt_mul<N>(A, T, U);
        comb(V, b[6], &A6, b[4], &A4, b[2], &A2, 0.0, false);
        comb(V, 0, nullptr, 0, nullptr, b[8], &A8, b[0], true);
    } else {
        const double b[] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                            129060195264000., 10559470521600., 670442572800., 33522128640., 1323241920.,
                            40840800., 960960., 16380., 182., 1.};
        const double maxnorm = 5.371920351148152;
        frexp(l1 / maxnorm, &squarings);
        if (squarings < 0) squarings = 0;
        double sc = ldexp(1.0, -squarings);
        for (int i = 0; i < NN; i++) { A.x[i] *= sc; A.y[i] *= sc; }
        blt_mul<N>(A, A, A2); blt_mul<N>(A2, A2, A4); blt_mul<N>(A4, A2, A6);
        comb(V, b[13], &A6, b[11], &A4, b[9], &A2, 0.0, false);
        blt_mul<N>(A6, V, T);
        comb(T, b[7], &A6, b[5], &A4, b[3], &A2, b[1], true);
        blt_mul<N>(A, T, U);
        comb(T, b[12], &A6, b[10], &A4, b[8], &A2, 0.0, false);
        blt_mul<N>(A6, T, V);
        comb(V, b[6], &A6, b[4], &A4, b[2], &A2, b[0], true);
    }
    // (V - U) R = V + U on the blocks: P Rx = Nx ; P Ry = Ny - S Rx, with P, S the blocks of V - U
    double Px[NN], Sy[NN], Nx[NN], Ny[NN], Rx[NN], Ry[NN], t[NN];
    for (int i = 0; i < NN; i++) { Px[i] = -U.x[i] + V.x[i]; Sy[i] = -U.y[i] + V.y[i]; Nx[i] = U.x[i] + V.x[i]; Ny[i] = U.y[i] + V.y[i]; }
    lu_solve<N>(Px, Nx, Rx);
    mm<N>(Sy, Rx, t);
    for (int i = 0; i < NN; i++) Ny[i] -= t[i];
    lu_solve<N>(Px, Ny, Ry);
    BltPair<N> E;
    for (int i = 0; i < NN; i++) { E.x[i] = Rx[i]; E.y[i] = Ry[i]; }
    for (int s = 0; s < squarings; s++) blt_mul<N>(E, E, E);
    for (int i = 0; i < NN; i++) { EX[i] = E.x[i]; EY[i] = E.y[i]; }
}

constexpr double kDareTol = 1e-8;   // utils/dare.h:7
constexpr int kDareMaxIter = 100;   // utils/dare.h:8

// utils/dare.h:10-33, Bd = H^T
template <int D>
__device__ int dare(const double* Ad, const double* Bd, const double* Q, double R, double* P) {
    double AdT[D * D], Pn[D * D], T1[D * D], PB[D], APB[D], BtP[D], BPA[D];
    mt<D>(Ad, AdT);
    for (int i = 0; i < D * D; i++) P[i] = Q[i];
    for (int it = 0; it < kDareMaxIter; it++) {
        mm<D>(AdT, P, T1); mm<D>(T1, Ad, Pn);
        mv<D>(P, Bd, PB);
        double g = R;
        for (int i = 0; i < D; i++) g += Bd[i] * PB[i];
        mv<D>(AdT, PB, APB);
        for (int j = 0; j < D; j++) { double s = 0.0; for (int i = 0; i < D; i++) s += Bd[i] * P[i * D + j]; BtP[j] = s; }
        for (int j = 0; j < D; j++) { double s = 0.0; for (int k = 0; k < D; k++) s += BtP[k] * Ad[k * D + j]; BPA[j] = s; }
        double ginv = 1.0 / g;
        double diff = -INFINITY;
        for (int i = 0; i < D; i++)
            for (int j = 0; j < D; j++) {
                double v = Pn[i * D + j] - APB[i] * ginv * BPA[j] + Q[i * D + j];   // dare.h:23
                Pn[i * D + j] = v;
                double dlt = v - P[i * D + j];
                if (dlt > diff) diff = dlt;                                       // maxCoeff, dare.h:25
            }
        diff = fabs(diff);
        for (int i = 0; i < D; i++)
            for (int j = 0; j < D; j++) P[i * D + j] = (Pn[i * D + j] + Pn[j * D + i]) / 2.0;   // dare.h:26
        if (diff < kDareTol) return it + 1;
    }
    return kDareMaxIter;
}

// utils/dare.h:36-58 (literal `AdT P Ad - P + Q`)
template <int D>
__device__ int dlyap(const double* Ad, const double* Q, double* P) {
    double AdT[D * D], Pn[D * D], T1[D * D];
    mt<D>(Ad, AdT);
    for (int i = 0; i < D * D; i++) P[i] = Q[i];
    for (int it = 0; it < kDareMaxIter; it++) {
        mm<D>(AdT, P, T1); mm<D>(T1, Ad, Pn);
        double diff = -INFINITY;
        for (int i = 0; i < D * D; i++) {
            double v = Pn[i] - P[i] + Q[i];
            Pn[i] = v;
            double dlt = v - P[i];
            if (dlt > diff) diff = dlt;
        }
        diff = fabs(diff);
        for (int i = 0; i < D; i++)
            for (int j = 0; j < D; j++) P[i * D + j] = (Pn[i * D + j] + Pn[j * D + i]) / 2.0;
        if (diff < kDareTol) return it + 1;
    }
    return kDareMaxIter;
}

}  // namespace
}  // namespace moihgp
