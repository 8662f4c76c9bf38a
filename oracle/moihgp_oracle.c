/* moihgp_oracle.c -- plain-C CPU restatement of the MOIHGP hot path.
 * TEST INFRASTRUCTURE ONLY (see moihgp_oracle.h).  PARITY UNPINNED (oracle/README.md).
 *
 * Every function cites the reference lines it follows; citations are into
 * /root/reference/moihgp/include/ unless a path is given.
 *
 * Third-party arithmetic restated from its published algorithm (dependency absent
 * from /root/reference and from this image): Eigen3 >= 3.3 (CMakeLists.txt:13, no pin)
 *   - MatrixBase::exp(): Higham (2005) scaling & squaring with [3/3]..[13/13] Pade
 *     approximants chosen on the matrix 1-norm, as in unsupported/Eigen/MatrixFunctions.
 *   - thin SVD -> only U V^T (polar factor) and singular values are consumed:
 *     one-sided Jacobi (Hestenes).
 *   - ldlt().solve on U0^T U0 -> Cholesky-free symmetric Gaussian elimination.
 */
#include "moihgp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define D ORC_DMAX
#define NP ORC_PMAX
#define NMAX (2 * ORC_DMAX)

/* ---------------------------------------------------------------- small dense helpers (row-major, n x n) */
static void mm(int n, const double* A, const double* B, double* C) {          /* C = A B */
    double T[NMAX * NMAX];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int k = 0; k < n; k++) s += A[i * n + k] * B[k * n + j];
            T[i * n + j] = s;
        }
    memcpy(C, T, sizeof(double) * n * n);
}
static void mt(int n, const double* A, double* At) {
    double T[NMAX * NMAX];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) T[j * n + i] = A[i * n + j];
    memcpy(At, T, sizeof(double) * n * n);
}
static void mv(int n, const double* A, const double* x, double* y) {
    double T[NMAX];
    for (int i = 0; i < n; i++) { double s = 0.0; for (int k = 0; k < n; k++) s += A[i * n + k] * x[k]; T[i] = s; }
    memcpy(y, T, sizeof(double) * n);
}
static int all_zero(int n, const double* A) { for (int i = 0; i < n * n; i++) if (A[i] != 0.0) return 0; return 1; }

/* solve LU X = B for square n, partial pivoting (Eigen partialPivLu().solve) */
static void lu_solve(int n, const double* Ain, const double* Bin, double* X) {
    double A[NMAX * NMAX], B[NMAX * NMAX];
    memcpy(A, Ain, sizeof(double) * n * n); memcpy(B, Bin, sizeof(double) * n * n);
    for (int k = 0; k < n; k++) {
        int p = k; double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++) if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
        if (p != k) for (int j = 0; j < n; j++) {
            double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t;
            t = B[k * n + j]; B[k * n + j] = B[p * n + j]; B[p * n + j] = t;
        }
        for (int i = k + 1; i < n; i++) {
            double f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            for (int j = 0; j < n; j++) B[i * n + j] -= f * B[k * n + j];
        }
    }
    for (int j = 0; j < n; j++)
        for (int i = n - 1; i >= 0; i--) {
            double s = B[i * n + j];
            for (int k = i + 1; k < n; k++) s -= A[i * n + k] * X[k * n + j];
            X[i * n + j] = s / A[i * n + i];
        }
}

/* ---------------------------------------------------------------- expm (Eigen MatrixExponential restated; ihgp.h:120,167) */
void orc_expm(int n, const double* Ain, double* E) {
    double A[NMAX * NMAX], A2[NMAX * NMAX], A4[NMAX * NMAX], A6[NMAX * NMAX], A8[NMAX * NMAX];
    double U[NMAX * NMAX], V[NMAX * NMAX], T[NMAX * NMAX];
    int nn = n * n, squarings = 0;
    double l1 = 0.0;
    for (int j = 0; j < n; j++) { double s = 0.0; for (int i = 0; i < n; i++) s += fabs(Ain[i * n + j]); if (s > l1) l1 = s; }
    memcpy(A, Ain, sizeof(double) * nn);
#define AXPYI(dst, c0) for (int i_ = 0; i_ < n; i_++) dst[i_ * n + i_] += (c0)
    if (l1 < 1.495585217958292e-002) {
        static const double b[] = {120., 60., 12., 1.};
        mm(n, A, A, A2);
        for (int i = 0; i < nn; i++) T[i] = b[3] * A2[i];
        AXPYI(T, b[1]); mm(n, A, T, U);
        for (int i = 0; i < nn; i++) V[i] = b[2] * A2[i];
        AXPYI(V, b[0]);
    } else if (l1 < 2.539398330063230e-001) {
        static const double b[] = {30240., 15120., 3360., 420., 30., 1.};
        mm(n, A, A, A2); mm(n, A2, A2, A4);
        for (int i = 0; i < nn; i++) T[i] = b[5] * A4[i] + b[3] * A2[i];
        AXPYI(T, b[1]); mm(n, A, T, U);
        for (int i = 0; i < nn; i++) V[i] = b[4] * A4[i] + b[2] * A2[i];
        AXPYI(V, b[0]);
    } else if (l1 < 9.504178996162932e-001) {
        static const double b[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
        mm(n, A, A, A2); mm(n, A2, A2, A4); mm(n, A4, A2, A6);
        for (int i = 0; i < nn; i++) T[i] = b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
        AXPYI(T, b[1]); mm(n, A, T, U);
        for (int i = 0; i < nn; i++) V[i] = b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
        AXPYI(V, b[0]);
    } else if (l1 < 2.097847961257068e+000) {
        static const double b[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.};
        mm(n, A, A, A2); mm(n, A2, A2, A4); mm(n, A4, A2, A6); mm(n, A6, A2, A8);
        for (int i = 0; i < nn; i++) T[i] = b[9] * A8[i] + b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
        AXPYI(T, b[1]); mm(n, A, T, U);
        for (int i = 0; i < nn; i++) V[i] = b[8] * A8[i] + b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
        AXPYI(V, b[0]);
    } else {
        static const double b[] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                                   129060195264000., 10559470521600., 670442572800., 33522128640., 1323241920.,
                                   40840800., 960960., 16380., 182., 1.};
        const double maxnorm = 5.371920351148152;
        frexp(l1 / maxnorm, &squarings);
        if (squarings < 0) squarings = 0;
        double sc = ldexp(1.0, -squarings);
        for (int i = 0; i < nn; i++) A[i] *= sc;
        mm(n, A, A, A2); mm(n, A2, A2, A4); mm(n, A4, A2, A6);
        for (int i = 0; i < nn; i++) V[i] = b[13] * A6[i] + b[11] * A4[i] + b[9] * A2[i];
        mm(n, A6, V, T);
        for (int i = 0; i < nn; i++) T[i] += b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
        AXPYI(T, b[1]); mm(n, A, T, U);
        for (int i = 0; i < nn; i++) T[i] = b[12] * A6[i] + b[10] * A4[i] + b[8] * A2[i];
        mm(n, A6, T, V);
        for (int i = 0; i < nn; i++) V[i] += b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
        AXPYI(V, b[0]);
    }
#undef AXPYI
    double Num[NMAX * NMAX], Den[NMAX * NMAX];
    for (int i = 0; i < nn; i++) { Num[i] = U[i] + V[i]; Den[i] = -U[i] + V[i]; }
    lu_solve(n, Den, Num, E);
    for (int s = 0; s < squarings; s++) mm(n, E, E, E);
}

/* ---------------------------------------------------------------- state-space models */
typedef struct {
    int d, P;
    double F[D * D], Pinf[D * D], H[D], R;
    double dF[NP][D * D], dPinf[NP][D * D], dR[NP];
} ss_t;

static void ss_build_stacked(int kernel, const double* params, ss_t* s);

int orc_dmax(void) { return ORC_DMAX; }
int orc_pmax(void) { return ORC_PMAX; }

static void ss_build(int kernel, const double* params, ss_t* s) {
    if (kernel >> 4) { ss_build_stacked(kernel, params, s); return; }
    memset(s, 0, sizeof(*s));
    double magnitude = params[0], lengthscale = params[1];
    s->P = 3;
    s->R = params[2];
    s->dR[2] = 1.0;                                                   /* matern32ss.h:30-33 */
    s->H[0] = 1.0;
    if (kernel == ORC_MATERN32) {                                     /* matern32ss.h:40-64 */
        int d = s->d = 2;
        double lam = sqrt(3.0) / lengthscale, lam2 = lam * lam;
        double len3 = 6.0 / (lengthscale * lengthscale * lengthscale);
        s->F[0 * d + 1] = 1.0;
        s->F[1 * d + 0] = -lam2;
        s->F[1 * d + 1] = -2.0 * lam;
        s->Pinf[0] = magnitude;
        s->Pinf[1 * d + 1] = magnitude * lam2;
        s->dF[1][1 * d + 0] = len3;
        s->dF[1][1 * d + 1] = 2.0 * lam / lengthscale;
        s->dPinf[0][0] = 1.0;                                          /* setIdentity, :27 */
        s->dPinf[0][1 * d + 1] = lam2;
        s->dPinf[1][1 * d + 1] = -magnitude * len3;
    } else {                                                          /* matern52ss.h:38-75 */
        int d = s->d = 3;
        double lam = sqrt(3.0) / lengthscale;                         /* :42 (sic) */
        double lam2 = lam * lam, len2 = lengthscale * lengthscale, len3 = len2 * lengthscale, len4 = len2 * len2;
        double kappa = 5.0 / 3.0 * magnitude / len2, kappa2 = -2.0 * kappa / lengthscale, sq5 = sqrt(5.0);
        s->F[0 * d + 1] = 1.0; s->F[1 * d + 2] = 1.0;
        s->F[2 * d + 0] = -lam2 * lam; s->F[2 * d + 1] = -3.0 * lam2; s->F[2 * d + 2] = -3.0 * lam;
        s->Pinf[0] = magnitude; s->Pinf[2 * d + 2] = 25.0 * magnitude / len4; s->Pinf[1 * d + 1] = kappa;
        s->Pinf[2 * d + 0] = -kappa; s->Pinf[0 * d + 2] = -kappa;
        s->dF[1][2 * d + 0] = 15.0 * sq5 / len4; s->dF[1][2 * d + 1] = 30.0 / len3; s->dF[1][2 * d + 2] = sq5 * lam2;
        for (int i = 0; i < d * d; i++) s->dPinf[0][i] = s->Pinf[i] / magnitude;     /* :66 */
        s->dPinf[1][1 * d + 1] = kappa2; s->dPinf[1][2 * d + 0] = -kappa2; s->dPinf[1][0 * d + 2] = -kappa2;
        s->dPinf[1][2 * d + 2] = -100.0 * magnitude / len2 / len3;
    }
}

/* Stacked state (header note): each component is built by the reference's own model code above and placed in its block. */
static void ss_build_stacked(int kernel, const double* params, ss_t* s) {
    int base = kernel & 15, J = kernel >> 4;
    int db = base == ORC_MATERN32 ? 2 : 3, d = db * J, P = 2 * J + 1;
    memset(s, 0, sizeof(*s));
    if (d > D || P > NP) { s->d = 0; s->P = 0; return; }               /* needs the wide build; orc_ihgp_update reports it */
    s->d = d; s->P = P;
    s->R = params[2 * J];
    s->dR[P - 1] = 1.0;
    for (int j = 0; j < J; j++) {
        ss_t* c = (ss_t*)malloc(sizeof(ss_t));
        double cp[3] = {params[2 * j], params[2 * j + 1], params[2 * J]};
        ss_build(base, cp, c);
        for (int a = 0; a < db; a++) {
            s->H[j * db + a] = c->H[a];
            for (int b = 0; b < db; b++) {
                int dst = (j * db + a) * d + (j * db + b), src = a * db + b;
                s->F[dst] = c->F[src];
                s->Pinf[dst] = c->Pinf[src];
                for (int q = 0; q < 2; q++) { s->dF[2 * j + q][dst] = c->dF[q][src]; s->dPinf[2 * j + q][dst] = c->dPinf[q][src]; }
            }
        }
        free(c);
    }
}

/* ---------------------------------------------------------------- utils/dare.h */
#define DARE_TOL 1e-8
#define DARE_MAXITER 100

/* utils/dare.h:10-33 with Bd = H^T = e_0 folded in only through explicit products */
static int dare(int n, const double* Ad, const double* Bd /* n */, const double* Q, double R, double* P) {
    double AdT[D * D], Pn[D * D], T1[D * D], PB[D], APB[D], BPA[D];
    mt(n, Ad, AdT);
    memcpy(P, Q, sizeof(double) * n * n);
    for (int it = 0; it < DARE_MAXITER; it++) {
        /* AdT P Ad */
        mm(n, AdT, P, T1); mm(n, T1, Ad, Pn);
        /* AdT P Bd (n) ; BdT P Ad (n) ; g = R + BdT P Bd */
        mv(n, P, Bd, PB);
        double g = R; for (int i = 0; i < n; i++) g += Bd[i] * PB[i];
        mv(n, AdT, PB, APB);                                          /* AdT * P * Bd */
        double BtP[D];
        for (int j = 0; j < n; j++) { double s = 0.0; for (int i = 0; i < n; i++) s += Bd[i] * P[i * n + j]; BtP[j] = s; }
        for (int j = 0; j < n; j++) { double s = 0.0; for (int k = 0; k < n; k++) s += BtP[k] * Ad[k * n + j]; BPA[j] = s; }
        double ginv = 1.0 / g;                                        /* (R + BdT P Bd).inverse() of a 1x1 */
        double diff = -INFINITY;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double v = Pn[i * n + j] - APB[i] * ginv * BPA[j] + Q[i * n + j];   /* :23 */
                Pn[i * n + j] = v;
                double dlt = v - P[i * n + j];
                if (dlt > diff) diff = dlt;                           /* maxCoeff, :25 */
            }
        diff = fabs(diff);
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) P[i * n + j] = (Pn[i * n + j] + Pn[j * n + i]) / 2.0;   /* :26 */
        if (diff < DARE_TOL) return it + 1;
    }
    return DARE_MAXITER;
}

/* utils/dare.h:36-58 */
static int dlyap(int n, const double* Ad, const double* Q, double* P) {
    double AdT[D * D], Pn[D * D], T1[D * D];
    mt(n, Ad, AdT);
    memcpy(P, Q, sizeof(double) * n * n);
    for (int it = 0; it < DARE_MAXITER; it++) {
        mm(n, AdT, P, T1); mm(n, T1, Ad, Pn);
        double diff = -INFINITY;
        for (int i = 0; i < n * n; i++) {
            double v = Pn[i] - P[i] + Q[i];                           /* :48 (sic) */
            Pn[i] = v;
            double dlt = v - P[i];
            if (dlt > diff) diff = dlt;
        }
        diff = fabs(diff);
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) P[i * n + j] = (Pn[i * n + j] + Pn[j * n + i]) / 2.0;
        if (diff < DARE_TOL) return it + 1;
    }
    return DARE_MAXITER;
}

/* ---------------------------------------------------------------- IHGP::update, ihgp.h:117-201 */
static int ihgp_update_core(orc_ihgp* g, const ss_t* sp, int kernel, double dt, const double* params);
int orc_ihgp_update(orc_ihgp* g, int kernel, double dt, const double* params) {
    ss_t s;
    ss_build(kernel, params, &s);
    return ihgp_update_core(g, &s, kernel, dt, params);
}
/* The same IHGP::update on a caller-supplied StateSpace (the template parameter of ihgp.h:17: any F, Pinf, H, R and their
 * derivatives), row-major, d <= ORC_DMAX, P <= ORC_PMAX.  For the branch audit (oracle/README.md): the reference's own two
 * models reach only three of the QLyap cases of ihgp.h:141-185; a synthetic StateSpace reaches all of them. */
int orc_ihgp_update_ss(orc_ihgp* g, int d, int P, double dt, const double* F, const double* Pinf, const double* H, double R,
                       const double* dF, const double* dPinf, const double* dR) {
    ss_t s;
    memset(&s, 0, sizeof(s));
    if (d < 1 || d > D || P < 0 || P > NP) { memset(g, 0, sizeof(*g)); return -1; }
    s.d = d; s.P = P; s.R = R;
    memcpy(s.F, F, sizeof(double) * d * d); memcpy(s.Pinf, Pinf, sizeof(double) * d * d); memcpy(s.H, H, sizeof(double) * d);
    for (int p = 0; p < P; p++) {
        memcpy(s.dF[p], dF + p * d * d, sizeof(double) * d * d);
        memcpy(s.dPinf[p], dPinf + p * d * d, sizeof(double) * d * d);
        s.dR[p] = dR[p];
    }
    return ihgp_update_core(g, &s, -1, dt, NULL);
}
static int ihgp_update_core(orc_ihgp* g, const ss_t* sp, int kernel, double dt, const double* params) {
    ss_t s = *sp;
    int n = s.d, nn = n * n;
    memset(g, 0, sizeof(*g));
    if (n == 0) return -1;                                            /* stacked model beyond this build's capacity */
    g->kernel = kernel; g->d = n; g->P = s.P; g->dt = dt;
    if (params) for (int p = 0; p < s.P; p++) g->params[p] = params[p];
    double T1[D * D], T2[D * D], AT[D * D];
    for (int i = 0; i < nn; i++) T1[i] = dt * s.F[i];
    orc_expm(n, T1, g->A);                                            /* :120 */
    const double* A = g->A;
    mt(n, A, AT);
    mm(n, A, s.Pinf, T1); mm(n, T1, AT, T2);
    for (int i = 0; i < nn; i++) T1[i] = s.Pinf[i] - T2[i];           /* :121 */
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) g->Q[i * n + j] = (T1[i * n + j] + T1[j * n + i]) / 2.0;   /* :122 */
    g->dare_iters = dare(n, A, s.H, g->Q, s.R, g->PP);                /* :125 (Bd = H^T) */
    const double* PP = g->PP;
    double PPHt[D]; mv(n, PP, s.H, PPHt);
    double S = s.R; for (int i = 0; i < n; i++) S += s.H[i] * PPHt[i];   /* :126 */
    g->S = S;
    for (int i = 0; i < n; i++) g->K[i] = PPHt[i] / S;                /* :127 */
    double HPP[D];
    for (int j = 0; j < n; j++) { double t = 0.0; for (int i = 0; i < n; i++) t += s.H[i] * PP[i * n + j]; HPP[j] = t; }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) g->PF[i * n + j] = PP[i * n + j] - g->K[i] * HPP[j];   /* :128 */
    for (int j = 0; j < n; j++) { double t = 0.0; for (int i = 0; i < n; i++) t += s.H[i] * A[i * n + j]; g->HA[j] = t; }   /* :129 */
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) g->AKHA[i * n + j] = A[i * n + j] - g->K[i] * g->HA[j];   /* :130 */
    double AK[D]; mv(n, A, g->K, AK);                                  /* :132 */
    double AAKH[D * D];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) AAKH[i * n + j] = A[i * n + j] - AK[i] * s.H[j];   /* :133 */

    for (int p = 0; p < s.P; p++) {                                   /* :136 */
        double dAT[D * D], dQ[D * D], QL[D * D], dPP[D * D];
        double* dA = g->dA[p];
        int dF_zero = all_zero(n, s.dF[p]), dPinf_zero = all_zero(n, s.dPinf[p]), dR_zero = (s.dR[p] == 0.0);
        if (dF_zero) {                                                /* :141 */
            memset(dA, 0, sizeof(double) * nn);                       /* :143 */
            if (dPinf_zero) memset(dQ, 0, sizeof(double) * nn);       /* :146 */
            else { mm(n, A, s.dPinf[p], T1); mm(n, T1, AT, T2); for (int i = 0; i < nn; i++) dQ[i] = s.dPinf[p][i] - T2[i]; }   /* :150 */
            if (dR_zero) memcpy(QL, dQ, sizeof(double) * nn);         /* :154 */
            else for (int i = 0; i < n; i++) for (int j = 0; j < n; j++)
                QL[i * n + j] = AK[i] * s.dR[p] * AK[j] + dQ[i * n + j];   /* :158 restated as AK dR AK^T (header note) */
        } else {
            double FF[NMAX * NMAX], EF[NMAX * NMAX];
            int m = 2 * n;
            memset(FF, 0, sizeof(FF));
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {  /* :163-166 */
                FF[i * m + j] = dt * s.F[i * n + j];
                FF[(n + i) * m + (n + j)] = dt * s.F[i * n + j];
                FF[(n + i) * m + j] = dt * s.dF[p][i * n + j];
            }
            orc_expm(m, FF, EF);
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) dA[i * n + j] = EF[(n + i) * m + j];   /* :167 */
            mt(n, dA, dAT);
            double dAPAt[D * D], APdAt[D * D];
            mm(n, dA, s.Pinf, T1); mm(n, T1, AT, dAPAt);              /* dA Pinf AT */
            mm(n, A, s.Pinf, T1); mm(n, T1, dAT, APdAt);              /* A Pinf dAT */
            if (dPinf_zero) for (int i = 0; i < nn; i++) dQ[i] = -dAPAt[i] - APdAt[i];   /* :171 */
            else { mm(n, A, s.dPinf[p], T1); mm(n, T1, AT, T2);
                   for (int i = 0; i < nn; i++) dQ[i] = s.dPinf[p][i] - dAPAt[i] - T2[i] - APdAt[i]; }   /* :175 */
            /* :179/:183  dA PP AT + A PP dAT - dA PP HT AK^T - AK H PP dAT [+ AK dR AK^T] + dQ */
            double t1[D * D], t2[D * D], dAPPHt[D], HPPdAT[D];
            mm(n, dA, PP, T1); mm(n, T1, AT, t1);
            mm(n, A, PP, T1); mm(n, T1, dAT, t2);
            mv(n, dA, PPHt, dAPPHt);                                  /* dA PP HT */
            for (int j = 0; j < n; j++) { double t = 0.0; for (int k = 0; k < n; k++) t += HPP[k] * dAT[k * n + j]; HPPdAT[j] = t; }
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
                double v = t1[i * n + j] + t2[i * n + j] - dAPPHt[i] * AK[j] - AK[i] * HPPdAT[j];
                if (!dR_zero) v += AK[i] * s.dR[p] * AK[j];
                QL[i * n + j] = v + dQ[i * n + j];
            }
        }
        g->dlyap_iters[p] = dlyap(n, AAKH, QL, dPP);                  /* :187 */
        double dS = s.dR[p];
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) dS += s.H[i] * dPP[i * n + j] * s.H[j];   /* :188 */
        g->dS[p] = dS;
        for (int i = 0; i < n; i++) {                                  /* :189 */
            double t = 0.0;
            for (int j = 0; j < n; j++) t += (dPP[i * n + j] - PP[i * n + j] * dS / S) * s.H[j];
            g->dK[p][i] = t / S;
        }
        if (dF_zero) {                                                /* :192-193 */
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) g->dAKHA[p][i * n + j] = -g->dK[p][i] * g->HA[j];
            for (int i = 0; i < n; i++) g->HdA[p][i] = 0.0;
        } else {                                                      /* :197-198 */
            double HdA[D];
            for (int j = 0; j < n; j++) { double t = 0.0; for (int i = 0; i < n; i++) t += s.H[i] * dA[i * n + j]; HdA[j] = t; }
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++)
                g->dAKHA[p][i * n + j] = dA[i * n + j] - g->dK[p][i] * g->HA[j] - g->K[i] * HdA[j];
            for (int i = 0; i < n; i++) g->HdA[p][i] = HdA[i];
        }
    }
    return g->dare_iters;
}

/* ---------------------------------------------------------------- IHGP::step x4, ihgp.h:37-100 */
void orc_ihgp_step(const orc_ihgp* g, const double* x, int has_y, double y, const double* dx,
                   double* xnew, double* yhat, double* dxnew) {
    int n = g->d;
    double xn[D];
    if (!has_y || isnan(y)) {                                         /* :39-47 / :96-100 */
        mv(n, g->A, x, xn);
        if (dx && dxnew) for (int p = 0; p < g->P; p++) {
            double a[D], b[D];
            mv(n, g->dA[p], x, a); mv(n, g->A, dx + p * n, b);
            for (int i = 0; i < n; i++) dxnew[p * n + i] = a[i] + b[i];   /* :45 */
        }
    } else {
        mv(n, g->AKHA, x, xn);
        for (int i = 0; i < n; i++) xn[i] = xn[i] + g->K[i] * y;      /* :50 */
        if (dx && dxnew) for (int p = 0; p < g->P; p++) {
            double a[D], b[D];
            mv(n, g->dAKHA[p], x, a); mv(n, g->AKHA, dx + p * n, b);
            for (int i = 0; i < n; i++) dxnew[p * n + i] = a[i] + b[i] + g->dK[p][i] * y;   /* :54 */
        }
    }
    for (int i = 0; i < n; i++) xnew[i] = xn[i];
    if (yhat) *yhat = xn[0];
}

/* ihgp.h:204-222 */
double orc_ihgp_nll(const orc_ihgp* g, const double* x, double y, const double* dx, double* grad) {
    int n = g->d;
    double hx = 0.0; for (int i = 0; i < n; i++) hx += g->HA[i] * x[i];
    double v = y - hx;
    double loss = 0.5 * (v * v / g->S + log(g->S));
    if (dx && grad) for (int p = 0; p < g->P; p++) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < n; i++) { a += g->HdA[p][i] * x[i]; b += g->HA[i] * dx[p * n + i]; }
        double dv = -a - b;                                           /* :218 restated (header note) */
        grad[p] = (v * dv - 0.5 * (v * v / g->S - 1) * g->dS[p]) / g->S;   /* :219 */
    }
    return loss;
}

/* ---------------------------------------------------------------- polar factor via one-sided Jacobi SVD */
int orc_polar(size_t M, size_t L, const double* Ain, double* Up, double* sv) {
    if (M < L) return -1;
    double* G = (double*)malloc(sizeof(double) * M * L);              /* columns get orthogonalised: G = A V */
    double* V = (double*)calloc(L * L, sizeof(double));
    memcpy(G, Ain, sizeof(double) * M * L);
    for (size_t i = 0; i < L; i++) V[i * L + i] = 1.0;
    const double eps = 2.220446049250313e-16;
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (size_t p = 0; p + 1 < L; p++)
            for (size_t q = p + 1; q < L; q++) {
                double a = 0, b = 0, c = 0;
                for (size_t i = 0; i < M; i++) { double gp = G[i * L + p], gq = G[i * L + q]; a += gp * gp; b += gq * gq; c += gp * gq; }
                if (fabs(c) <= eps * sqrt(a * b) || c == 0.0) continue;
                rotated = 1;
                double zeta = (b - a) / (2.0 * c);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (size_t i = 0; i < M; i++) { double gp = G[i * L + p], gq = G[i * L + q]; G[i * L + p] = cs * gp - sn * gq; G[i * L + q] = sn * gp + cs * gq; }
                for (size_t i = 0; i < L; i++) { double vp = V[i * L + p], vq = V[i * L + q]; V[i * L + p] = cs * vp - sn * vq; V[i * L + q] = sn * vp + cs * vq; }
            }
        if (!rotated) break;
    }
    for (size_t j = 0; j < L; j++) {
        double s = 0; for (size_t i = 0; i < M; i++) s += G[i * L + j] * G[i * L + j];
        s = sqrt(s);
        if (sv) sv[j] = s;
        for (size_t i = 0; i < M; i++) G[i * L + j] /= s;             /* now G = svdU */
    }
    for (size_t i = 0; i < M; i++)
        for (size_t j = 0; j < L; j++) {
            double s = 0; for (size_t k = 0; k < L; k++) s += G[i * L + k] * V[j * L + k];
            Up[i * L + j] = s;                                        /* svdU * svdV^T, moihgp.h:439 */
        }
    free(G); free(V);
    return 0;
}

/* ---------------------------------------------------------------- MOIHGP, moihgp.h:76-757 */
struct orc_gp {
    int kernel; double dt; size_t M, L; int d, P; size_t num_param;
    double* U;      /* [M][L] */
    double* S;      /* [L] */
    double sigma;
    orc_ihgp* igp;  /* [L] */
    int literal_ugrad;
    int threading;  /* moihgp.h:747 _threading: observable through the value lik1 returns (:590 vs :597-607) */
};

orc_gp* orc_gp_new(int kernel, double dt, size_t M, size_t L) {
    orc_gp* gp = (orc_gp*)calloc(1, sizeof(orc_gp));
    gp->kernel = kernel; gp->dt = dt; gp->M = M; gp->L = L;
    gp->igp = (orc_ihgp*)calloc(L, sizeof(orc_ihgp));
    double def[ORC_PMAX > 3 ? ORC_PMAX : 3] = {1.0, 1.0, 0.1};        /* matern32ss.h:34-36 */
    if (kernel >> 4) {                                                /* stacked: (magnitude 1, lengthscale j + 1) per component, noise 0.1 */
        int J = kernel >> 4;
        for (int j = 0; j < J && 2 * j + 1 < (int)(sizeof(def) / sizeof(def[0])); j++) { def[2 * j] = 1.0; def[2 * j + 1] = (double)(j + 1); }
        if (2 * J < (int)(sizeof(def) / sizeof(def[0]))) def[2 * J] = 0.1;
    }
    for (size_t l = 0; l < L; l++) orc_ihgp_update(&gp->igp[l], kernel, dt, def);   /* moihgp.h:86-90 */
    gp->d = gp->igp[0].d; gp->P = gp->igp[0].P;
    gp->num_param = M * L + L + 1 + L * gp->P;                        /* :93 */
    gp->U = (double*)calloc(M * L, sizeof(double));
    for (size_t i = 0; i < M && i < L; i++) gp->U[i * L + i] = 1.0;   /* deterministic stand-in for :103-125 (random) */
    gp->S = (double*)malloc(sizeof(double) * L);
    for (size_t l = 0; l < L; l++) gp->S[l] = 1.0;                    /* :126 */
    gp->sigma = 1e-2;                                                 /* :127 */
    gp->literal_ugrad = 1;
    gp->threading = 0;                                                /* the reference's default everywhere (pywrapper.py:12) */
    return gp;
}
/* moihgp.h:81 constructor argument `threading`, with the override of :128-135 (fewer than two latents: always off) */
void orc_gp_set_threading(orc_gp* gp, int threading) { gp->threading = (gp->L < 2) ? 0 : (threading != 0); }
int orc_gp_get_threading(orc_gp* gp) { return gp->threading; }
orc_gp* orc_gp_new_t(int kernel, double dt, size_t M, size_t L, int threading) {
    orc_gp* gp = orc_gp_new(kernel, dt, M, L);
    orc_gp_set_threading(gp, threading);
    return gp;
}
void orc_gp_del(orc_gp* gp) { if (!gp) return; free(gp->U); free(gp->S); free(gp->igp); free(gp); }
size_t orc_gp_igp_dim(orc_gp* gp) { return (size_t)gp->d; }
size_t orc_gp_num_param(orc_gp* gp) { return gp->num_param; }
size_t orc_gp_num_igp_param(orc_gp* gp) { return (size_t)gp->P; }
const orc_ihgp* orc_gp_latent(orc_gp* gp, size_t l) { return &gp->igp[l]; }
void orc_gp_get_U(orc_gp* gp, double* U) { memcpy(U, gp->U, sizeof(double) * gp->M * gp->L); }
void orc_gp_set_literal_ugrad(orc_gp* gp, int literal) { gp->literal_ugrad = literal; }

void orc_gp_update(orc_gp* gp, const double* params) {                /* :431-457 */
    size_t M = gp->M, L = gp->L, sizeU = M * L;
    orc_polar(M, L, params, gp->U, NULL);                             /* :436-446, params head is U row-major */
    memcpy(gp->S, params + sizeU, sizeof(double) * L);                /* :448 */
    gp->sigma = params[sizeU + L];                                    /* :449 */
    const double* igp = params + sizeU + L + 1;                       /* :450-456: column l of a P x L col-major map */
    for (size_t l = 0; l < L; l++) orc_ihgp_update(&gp->igp[l], gp->kernel, gp->dt, igp + l * gp->P);
}

void orc_gp_get_params(orc_gp* gp, double* params) {                  /* :721-738 */
    size_t M = gp->M, L = gp->L, sizeU = M * L;
    memcpy(params, gp->U, sizeof(double) * sizeU);
    memcpy(params + sizeU, gp->S, sizeof(double) * L);
    params[sizeU + L] = gp->sigma;
    for (size_t l = 0; l < L; l++) for (int p = 0; p < gp->P; p++) params[sizeU + L + 1 + l * gp->P + p] = gp->igp[l].params[p];
}

/* symmetric positive definite solve (stand-in for Eigen ldlt().solve, moihgp.h:177) */
static void spd_solve(size_t n, double* A, double* b) {
    for (size_t k = 0; k < n; k++) {
        for (size_t i = k + 1; i < n; i++) {
            double f = A[i * n + k] / A[k * n + k];
            for (size_t j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (size_t ii = n; ii-- > 0;) {
        double s = b[ii];
        for (size_t k = ii + 1; k < n; k++) s -= A[ii * n + k] * b[k];
        b[ii] = s / A[ii * n + ii];
    }
}

void orc_gp_project(orc_gp* gp, const double* y, double* Ty) {        /* :150-182 */
    size_t M = gp->M, L = gp->L, nobs = 0;
    for (size_t i = 0; i < M; i++) if (!isnan(y[i])) nobs++;
    if (nobs != M) {                                                  /* :167-178 */
        double* N = (double*)calloc(L * L, sizeof(double));
        double* r = (double*)calloc(L, sizeof(double));
        for (size_t i = 0; i < M; i++) {
            if (isnan(y[i])) continue;
            for (size_t a = 0; a < L; a++) {
                r[a] += gp->U[i * L + a] * y[i];
                for (size_t b = 0; b < L; b++) N[a * L + b] += gp->U[i * L + a] * gp->U[i * L + b];
            }
        }
        spd_solve(L, N, r);
        for (size_t l = 0; l < L; l++) Ty[l] = (1.0 / sqrt(gp->S[l])) * r[l];
        free(N); free(r);
    } else {
        for (size_t l = 0; l < L; l++) {                              /* :181 */
            double s = 0.0;
            for (size_t i = 0; i < M; i++) s += gp->U[i * L + l] * y[i];
            Ty[l] = (1.0 / sqrt(gp->S[l])) * s;
        }
    }
}

static void unproject(orc_gp* gp, const double* Tyhat, double* yhat) {   /* :222-225 */
    for (size_t i = 0; i < gp->M; i++) {
        double s = 0.0;
        for (size_t l = 0; l < gp->L; l++) s += gp->U[i * gp->L + l] * (sqrt(gp->S[l]) * Tyhat[l]);
        yhat[i] = s;
    }
}

static void gp_step(orc_gp* gp, const double* x, const double* y, const double* dx, double* xnew, double* yhat, double* dxnew) {
    size_t L = gp->L; int d = gp->d, P = gp->P;
    double* Ty = (double*)malloc(sizeof(double) * L);
    double* Tyhat = (double*)malloc(sizeof(double) * L);
    if (y) orc_gp_project(gp, y, Ty);
    for (size_t l = 0; l < L; l++)                                    /* :217-221 */
        orc_ihgp_step(&gp->igp[l], x + l * d, y != NULL, y ? Ty[l] : 0.0, dx ? dx + l * P * d : NULL,
                      xnew + l * d, &Tyhat[l], dxnew ? dxnew + l * P * d : NULL);
    if (yhat) unproject(gp, Tyhat, yhat);
    free(Ty); free(Tyhat);
}
void orc_gp_step1(orc_gp* gp, const double* x, const double* y, const double* dx, double* xnew, double* yhat, double* dxnew) { gp_step(gp, x, y, dx, xnew, yhat, dxnew); }
void orc_gp_step2(orc_gp* gp, const double* x, const double* y, const double* dx, double* xnew, double* dxnew) { gp_step(gp, x, y, dx, xnew, NULL, dxnew); }
void orc_gp_step3(orc_gp* gp, const double* x, const double* y, double* xnew, double* yhat) { gp_step(gp, x, y, NULL, xnew, yhat, NULL); }
void orc_gp_step4(orc_gp* gp, const double* x, double* xnew, double* yhat) { gp_step(gp, x, NULL, NULL, xnew, yhat, NULL); }

/* global NLL terms, moihgp.h:649-653 / :499-503 */
static double global_terms(orc_gp* gp, const double* y, double* y_UUTy_out, double* m_n_out, double* Uty) {
    size_t M = gp->M, L = gp->L;
    for (size_t l = 0; l < L; l++) { double s = 0.0; for (size_t i = 0; i < M; i++) s += gp->U[i * L + l] * y[i]; Uty[l] = s; }
    double nrm = 0.0;
    for (size_t i = 0; i < M; i++) {                                  /* ((I - U U^T) y).norm() */
        double s = 0.0;
        for (size_t l = 0; l < L; l++) s += gp->U[i * L + l] * Uty[l];
        double r = y[i] - s;
        nrm += r * r;
    }
    nrm = sqrt(nrm);
    double m_n = (double)M - (double)L; if (m_n < 0.0) m_n = 0.0;     /* :502 */
    double Ssum = 0.0; for (size_t l = 0; l < L; l++) Ssum += gp->S[l];
    *y_UUTy_out = nrm; *m_n_out = m_n;
    return 0.5 * log(Ssum) + 0.5 * m_n * log(gp->sigma) + 0.5 * nrm / gp->sigma;   /* :503 (sic) */
}

double orc_gp_lik2(orc_gp* gp, const double* x, const double* y) {   /* :614-688 */
    size_t L = gp->L; int d = gp->d;
    double* Ty = (double*)malloc(sizeof(double) * L);
    double* Uty = (double*)malloc(sizeof(double) * L);
    orc_gp_project(gp, y, Ty);
    double nrm, m_n;
    double loss = global_terms(gp, y, &nrm, &m_n, Uty);
    for (size_t l = 0; l < L; l++) loss += orc_ihgp_nll(&gp->igp[l], x + l * d, Ty[l], NULL, NULL);   /* :684 */
    free(Ty); free(Uty);
    return loss;
}

double orc_gp_lik1(orc_gp* gp, const double* x, const double* y, const double* dx, double* grad) {   /* :460-611 */
    size_t M = gp->M, L = gp->L, sizeU = M * L; int d = gp->d, P = gp->P;
    double* Ty = (double*)malloc(sizeof(double) * L);
    double* Uty = (double*)malloc(sizeof(double) * L);
    double* pv = (double*)malloc(sizeof(double) * L);
    orc_gp_project(gp, y, Ty);
    double nrm, m_n;
    double loss = global_terms(gp, y, &nrm, &m_n, Uty);
    double sigma = gp->sigma;
    for (size_t l = 0; l < L; l++) {                                  /* :505-512 */
        const orc_ihgp* g = &gp->igp[l];
        double hax = 0.0, hak = 0.0;
        for (int i = 0; i < d; i++) { hax += g->HA[i] * x[l * d + i]; hak += g->HA[i] * g->K[i]; }
        double vi = y[l] - hax;                                       /* raw y(idx), sic :510 */
        pv[l] = vi * (1 - hak) / g->S;
    }
    memset(grad, 0, sizeof(double) * gp->num_param);                  /* :537 */
    if (gp->literal_ugrad) {                                          /* :513-552 */
        /* SVD of U: singular values sv, Lmat = I + su (invS - I) su^T, Rmat = I + sv (invS - I) sv^T.
         * With U = G diag(1/s) V^T... we get su, sv from the same Jacobi routine applied to U. */
        double* G = (double*)malloc(sizeof(double) * M * L);
        double* V = (double*)calloc(L * L, sizeof(double));
        double* s = (double*)malloc(sizeof(double) * L);
        /* one-sided Jacobi inline (same as orc_polar but keeping su, sv) */
        memcpy(G, gp->U, sizeof(double) * M * L);
        for (size_t i = 0; i < L; i++) V[i * L + i] = 1.0;
        const double eps = 2.220446049250313e-16;
        for (int sweep = 0; sweep < 60; sweep++) {
            int rotated = 0;
            for (size_t p = 0; p + 1 < L; p++) for (size_t q = p + 1; q < L; q++) {
                double a = 0, b = 0, c = 0;
                for (size_t i = 0; i < M; i++) { double gp_ = G[i * L + p], gq = G[i * L + q]; a += gp_ * gp_; b += gq * gq; c += gp_ * gq; }
                if (fabs(c) <= eps * sqrt(a * b) || c == 0.0) continue;
                rotated = 1;
                double zeta = (b - a) / (2.0 * c);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (size_t i = 0; i < M; i++) { double gp_ = G[i * L + p], gq = G[i * L + q]; G[i * L + p] = cs * gp_ - sn * gq; G[i * L + q] = sn * gp_ + cs * gq; }
                for (size_t i = 0; i < L; i++) { double vp = V[i * L + p], vq = V[i * L + q]; V[i * L + p] = cs * vp - sn * vq; V[i * L + q] = sn * vp + cs * vq; }
            }
            if (!rotated) break;
        }
        for (size_t j = 0; j < L; j++) {
            double t = 0; for (size_t i = 0; i < M; i++) t += G[i * L + j] * G[i * L + j];
            s[j] = sqrt(t);
            for (size_t i = 0; i < M; i++) G[i * L + j] /= s[j];
        }
        double* Lm = (double*)calloc(M * M, sizeof(double));
        double* Rm = (double*)calloc(L * L, sizeof(double));
        for (size_t i = 0; i < M; i++) for (size_t j = 0; j < M; j++) {
            double t = (i == j) ? 1.0 : 0.0;
            for (size_t k = 0; k < L; k++) t += G[i * L + k] * (1.0 / s[k] - 1.0) * G[j * L + k];
            Lm[i * M + j] = t;
        }
        for (size_t i = 0; i < L; i++) for (size_t j = 0; j < L; j++) {
            double t = (i == j) ? 1.0 : 0.0;
            for (size_t k = 0; k < L; k++) t += V[i * L + k] * (1.0 / s[k] - 1.0) * V[j * L + k];
            Rm[i * L + j] = t;
        }
        /* yU = y^T U  (= Uty) */
        for (size_t r = 0; r < M; r++) for (size_t c = 0; c < L; c++) {
            /* dU = Lm[:, r] (outer) Rm[c, :]  =>  dU^T y = Rm[c,:]^T * (Lm[:,r] . y) */
            double ly = 0.0; for (size_t i = 0; i < M; i++) ly += Lm[i * M + r] * y[i];
            double val = 0.0, acc = 0.0;
            for (size_t k = 0; k < L; k++) {
                double dUTy_k = Rm[c * L + k] * ly;
                acc += Uty[k] * dUTy_k;                               /* y^T U dU^T y */
                val += pv[k] * (1.0 / sqrt(gp->S[k])) * dUTy_k;       /* :547-551 */
            }
            grad[r * L + c] = -acc / sigma + val;                     /* :546 */
        }
        free(G); free(V); free(s); free(Lm); free(Rm);
    } else {
        for (size_t r = 0; r < M; r++) for (size_t c = 0; c < L; c++)
            grad[r * L + c] = y[r] * (pv[c] / sqrt(gp->S[c]) - Uty[c] / sigma);
    }
    for (size_t l = 0; l < L; l++) {                                  /* :553-562 */
        double sq = sqrt(gp->S[l]);
        grad[sizeU + l] = 0.5 / gp->S[l] + pv[l] * (-0.5 * (1.0 / sq / sq / sq) * Uty[l]);
    }
    grad[sizeU + L] = 0.5 * (m_n - nrm / sigma) / sigma;              /* :563 */
    for (size_t l = 0; l < L; l++) {                                  /* :564-607 */
        double g[NP];
        double ll = orc_ihgp_nll(&gp->igp[l], x + l * d, Ty[l], dx + l * P * d, g);
        /* The two branches of the reference differ: the threaded one adds the per-latent loss (:590 `loss += args[idx].loss`),
         * the serial one calls IHGP::negLogLikelihood for its gradient and DROPS the value it returns (:600, no `loss +=`).
         * With threading off (the default, and forced for L < 2) lik1 therefore returns the three global terms of :503 only. */
        if (gp->threading) loss += ll;
        double dn = g[P - 1];
        grad[sizeU + l] -= dn * sigma / gp->S[l] / gp->S[l];
        grad[sizeU + L] += dn / gp->S[l];
        for (int p = 0; p < P; p++) grad[sizeU + L + 1 + l * P + p] = g[p];   /* :608-609 */
    }
    free(Ty); free(Uty); free(pv);
    return loss;
}

/* ---------------------------------------------------------------- batched sweeps (the timed hot loop) */
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#define STREAM_AT(Ty, l, t) ((layout) == 0 ? (Ty)[(l) * ld + (t)] : (Ty)[(t) * ld + (l)])

double orc_filter_stream(const orc_ihgp* g, size_t L, size_t T, const double* Ty, size_t ld, int layout,
                         double* x, double* yhat, double* nll_per_latent, int nthreads) {
    double total = 0.0;
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : total) schedule(static)
    for (long l = 0; l < (long)L; l++) {
        const orc_ihgp* gl = &g[l];
        int n = gl->d;
        double xs[D], logS = log(gl->S), acc = 0.0;
        for (int i = 0; i < n; i++) xs[i] = x[l * n + i];
        for (size_t t = 0; t < T; t++) {
            double y = STREAM_AT(Ty, l, t), xn[D];
            if (isnan(y)) {
                for (int i = 0; i < n; i++) { double s = 0.0; for (int k = 0; k < n; k++) s += gl->A[i * n + k] * xs[k]; xn[i] = s; }
            } else {
                double hx = 0.0; for (int i = 0; i < n; i++) hx += gl->HA[i] * xs[i];
                double v = y - hx;
                acc += 0.5 * (v * v / gl->S + logS);                  /* ihgp.h:206-207, pre-step x */
                for (int i = 0; i < n; i++) { double s = 0.0; for (int k = 0; k < n; k++) s += gl->AKHA[i * n + k] * xs[k]; xn[i] = s + gl->K[i] * y; }   /* ihgp.h:90 */
            }
            for (int i = 0; i < n; i++) xs[i] = xn[i];
            if (yhat) { if (layout == 0) yhat[l * ld + t] = xn[0]; else yhat[t * ld + l] = xn[0]; }
        }
        for (int i = 0; i < n; i++) x[l * n + i] = xs[i];
        if (nll_per_latent) nll_per_latent[l] = acc;
        total += acc;
    }
    return total;
}

double orc_filter_stream_f32(const orc_ihgp* g, size_t L, size_t T, const float* Ty, size_t ld, int layout,
                             float* x, float* yhat, double* nll_per_latent, int nthreads) {
    double total = 0.0;
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : total) schedule(static)
    for (long l = 0; l < (long)L; l++) {
        const orc_ihgp* gl = &g[l];
        int n = gl->d;
        float xs[D], A[D * D], AKHA[D * D], K[D], HA[D];
        for (int i = 0; i < n * n; i++) { A[i] = (float)gl->A[i]; AKHA[i] = (float)gl->AKHA[i]; }
        for (int i = 0; i < n; i++) { K[i] = (float)gl->K[i]; HA[i] = (float)gl->HA[i]; xs[i] = x[l * n + i]; }
        double logS = log(gl->S), acc = 0.0, Sinv = 1.0 / gl->S;
        for (size_t t = 0; t < T; t++) {
            float y = STREAM_AT(Ty, l, t), xn[D];
            if (isnan(y)) {
                for (int i = 0; i < n; i++) { float s = 0.0f; for (int k = 0; k < n; k++) s += A[i * n + k] * xs[k]; xn[i] = s; }
            } else {
                float hx = 0.0f; for (int i = 0; i < n; i++) hx += HA[i] * xs[i];
                double v = (double)(y - hx);
                acc += 0.5 * (v * v * Sinv + logS);
                for (int i = 0; i < n; i++) { float s = 0.0f; for (int k = 0; k < n; k++) s += AKHA[i * n + k] * xs[k]; xn[i] = s + K[i] * y; }
            }
            for (int i = 0; i < n; i++) xs[i] = xn[i];
            if (yhat) { if (layout == 0) yhat[l * ld + t] = xn[0]; else yhat[t * ld + l] = xn[0]; }
        }
        for (int i = 0; i < n; i++) x[l * n + i] = xs[i];
        if (nll_per_latent) nll_per_latent[l] = acc;
        total += acc;
    }
    return total;
}

/* ---------------------------------------------------------------- fair-optimised CPU baseline (BASELINE.md variant (ii), round 4)
 * The same filter sweep as orc_filter_stream (ihgp.h:81-93 + :204-209 per tick and latent) written the way a CPU wants it, so that the
 * number bench.py prints next to the GPU's is a fair one: state dimension fixed at compile time (one instantiation per d), the layout
 * branch hoisted out of the tick loop, no division and no isnan branch per tick, and FB latents side by side in the lanes of a SIMD
 * register (struct-of-arrays constants and states per block; the tick loop is `omp simd` over the block).  Series-major streams are
 * transposed tile by tile (FT ticks x FB latents) so that the recursion reads and writes contiguous vectors.  Innovation form
 * (v = y - HA x, x' = A x + K v; a missing tick is v = 0: ihgp.h:83-87), Sum v^2 accumulated per tile in stream precision and in fp64 across
 * tiles: results equal orc_filter_stream's to rounding (tests/test_oracle.py), not bit for bit.  Timed by bench.py only. */
#define FB64 8
#define FB32 16
#define FT 64
/* Series-major streams reach the tick-major tile either row by row (scalar copies: FT contiguous loads per latent) or tick by tick (one
 * gather / scatter per tick).  Measured with the vectorised tick loop on 16 threads of the GPU boxes' EPYC 9575F (bench.py cpu_baseline, C3
 * shape): 18.8e9 steps/s with gathers (4.3 cycles per step and thread), 14.2e9 without; a 2.1 GHz AVX-512 Xeon: 2.7 against 3.3 ns per step
 * and thread.  On unless asked otherwise (make native FASTFLAGS=-DORC_FAST_GATHER=0). */
#ifndef ORC_FAST_GATHER
#define ORC_FAST_GATHER 1
#endif

#define ORC_DEF_FAST(N, REAL, FB, SUF, UINT)                                                                                            \
static void fast_block_##N##SUF(const orc_ihgp* g, size_t nb, size_t T, const REAL* Ty, size_t ld, int layout, size_t l0,         \
                                REAL* x, REAL* yhat, double* nllp) {                                                             \
    REAL A[N * N][FB], K[N][FB], HA[N][FB], xs[N][FB], yb[FT][FB], ob[FT][FB], part[FB], pcnt[FB];                                \
    double Sinv[FB], logS[FB], acc[FB], cnt[FB];                                                                                  \
    for (size_t b = 0; b < FB; b++) {                                                                                             \
        const orc_ihgp* gl = &g[b < nb ? b : 0];                                                                                  \
        for (int i = 0; i < N * N; i++) A[i][b] = (REAL)gl->A[i];                                                                 \
        for (int i = 0; i < N; i++) { K[i][b] = (REAL)gl->K[i]; HA[i][b] = (REAL)gl->HA[i]; xs[i][b] = b < nb ? x[(l0 + b) * N + i] : (REAL)0; } \
        Sinv[b] = 1.0 / gl->S; logS[b] = log(gl->S); acc[b] = 0.0; cnt[b] = 0.0;                                                   \
    }                                                                                                                             \
    for (size_t t0 = 0; t0 < T; t0 += FT) {                                                                                       \
        const size_t tt = T - t0 < FT ? T - t0 : FT;                                                                              \
        if (ORC_FAST_GATHER && layout == 0 && nb == FB) {   /* a full block: the FB strided loads of a tick as one gather */      \
            const REAL* r0 = Ty + l0 * ld + t0;                                                                                   \
            for (size_t t = 0; t < tt; t++) {                                                                                     \
                _Pragma("omp simd")                                                                                               \
                for (size_t b = 0; b < FB; b++) yb[t][b] = r0[b * ld + t];                                                        \
            }                                                                                                                     \
        } else if (layout == 0) {                                                                                                 \
            for (size_t b = 0; b < FB; b++) {                                                                                     \
                if (b < nb) { const REAL* r = Ty + (l0 + b) * ld + t0; for (size_t t = 0; t < tt; t++) yb[t][b] = r[t]; }         \
                else for (size_t t = 0; t < tt; t++) yb[t][b] = (REAL)NAN;                                                        \
            }                                                                                                                     \
        } else {                                                                                                                  \
            for (size_t t = 0; t < tt; t++) { const REAL* r = Ty + (t0 + t) * ld + l0; for (size_t b = 0; b < FB; b++) yb[t][b] = b < nb ? r[b] : (REAL)NAN; } \
        }                                                                                                                         \
        for (size_t b = 0; b < FB; b++) { part[b] = (REAL)0; pcnt[b] = (REAL)0; }     /* (one scalar type inside the simd loop) */ \
        for (size_t t = 0; t < tt; t++) {                                                                                         \
            _Pragma("omp simd")                                                                                                   \
            for (size_t b = 0; b < FB; b++) {                                                                                     \
                const REAL y = yb[t][b];                                                                                          \
                const int obs = (y == y);                                                                                         \
                REAL hx = (REAL)0, xn[N];                                                                                         \
                for (int i = 0; i < N; i++) hx += HA[i][b] * xs[i][b];                                                            \
                /* v = obs ? y - hx : 0 as a bit mask: gcc leaves the select on a NaN test unconverted ("control flow in loop") */   \
                /* and the whole latent-parallel loop scalar -- 13 cycles per step where the vector form takes 2 */                \
                const REAL dv = y - hx;                                                                                           \
                UINT bits; memcpy(&bits, &dv, sizeof bits); bits &= (UINT)0 - (UINT)obs;                                          \
                REAL v; memcpy(&v, &bits, sizeof v);                                                                              \
                part[b] += v * v;                                                                                                 \
                pcnt[b] += (REAL)obs;                                                                                             \
                for (int i = 0; i < N; i++) { REAL s_ = K[i][b] * v; for (int k = 0; k < N; k++) s_ += A[i * N + k][b] * xs[k][b]; xn[i] = s_; } \
                for (int i = 0; i < N; i++) xs[i][b] = xn[i];                                                                     \
                ob[t][b] = xn[0];                                                                                                 \
            }                                                                                                                     \
        }                                                                                                                         \
        for (size_t b = 0; b < FB; b++) { acc[b] += (double)part[b]; cnt[b] += (double)pcnt[b]; }                                 \
        if (yhat) {                                                                                                               \
            if (ORC_FAST_GATHER && layout == 0 && nb == FB) {                                                                     \
                REAL* r0 = yhat + l0 * ld + t0;                                                                                   \
                for (size_t t = 0; t < tt; t++) {                                                                                 \
                    _Pragma("omp simd")                                                                                           \
                    for (size_t b = 0; b < FB; b++) r0[b * ld + t] = ob[t][b];                                                    \
                }                                                                                                                 \
            }                                                                                                                     \
            else if (layout == 0) { for (size_t b = 0; b < nb; b++) { REAL* r = yhat + (l0 + b) * ld + t0; for (size_t t = 0; t < tt; t++) r[t] = ob[t][b]; } } \
            else { for (size_t t = 0; t < tt; t++) { REAL* r = yhat + (t0 + t) * ld + l0; for (size_t b = 0; b < nb; b++) r[b] = ob[t][b]; } } \
        }                                                                                                                         \
    }                                                                                                                             \
    for (size_t b = 0; b < nb; b++) {                                                                                             \
        for (int i = 0; i < N; i++) x[(l0 + b) * N + i] = xs[i][b];                                                               \
        nllp[b] = 0.5 * (acc[b] * Sinv[b] + cnt[b] * logS[b]);                                                                    \
    }                                                                                                                             \
}

ORC_DEF_FAST(2, double, FB64, d, uint64_t) ORC_DEF_FAST(3, double, FB64, d, uint64_t) ORC_DEF_FAST(2, float, FB32, f, uint32_t) ORC_DEF_FAST(3, float, FB32, f, uint32_t)
#if ORC_DMAX >= 12
ORC_DEF_FAST(4, double, FB64, d, uint64_t) ORC_DEF_FAST(6, double, FB64, d, uint64_t) ORC_DEF_FAST(8, double, FB64, d, uint64_t) ORC_DEF_FAST(9, double, FB64, d, uint64_t) ORC_DEF_FAST(12, double, FB64, d, uint64_t)
ORC_DEF_FAST(4, float, FB32, f, uint32_t) ORC_DEF_FAST(6, float, FB32, f, uint32_t) ORC_DEF_FAST(8, float, FB32, f, uint32_t) ORC_DEF_FAST(9, float, FB32, f, uint32_t) ORC_DEF_FAST(12, float, FB32, f, uint32_t)
#endif

/* is_f32: Ty / x / yhat are float arrays (the fp32 configs), else double.  Returns the total NLL, or NAN for a state dimension without an
 * instantiation (or latents of mixed dimension). */
double orc_filter_stream_fast(const orc_ihgp* g, size_t L, size_t T, const void* Ty, size_t ld, int layout, void* x, void* yhat,
                              double* nll_per_latent, int nthreads, int is_f32) {
    if (L == 0) return 0.0;
    const int n = g[0].d;
    for (size_t l = 1; l < L; l++) if (g[l].d != n) return NAN;
    const size_t fb = is_f32 ? FB32 : FB64, nblk = (L + fb - 1) / fb;
    double total = 0.0;
    int bad = 0;
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : total) schedule(static)
    for (long bi = 0; bi < (long)nblk; bi++) {
        const size_t l0 = (size_t)bi * fb, nb = L - l0 < fb ? L - l0 : fb;
        double nl[FB32];
#define ORC_FAST_CASE(N)                                                                                                          \
        case N: { if (is_f32) fast_block_##N##f(g + l0, nb, T, (const float*)Ty, ld, layout, l0, (float*)x, (float*)yhat, nl);    \
                  else fast_block_##N##d(g + l0, nb, T, (const double*)Ty, ld, layout, l0, (double*)x, (double*)yhat, nl); } break;
        switch (n) {
            ORC_FAST_CASE(2) ORC_FAST_CASE(3)
#if ORC_DMAX >= 12
            ORC_FAST_CASE(4) ORC_FAST_CASE(6) ORC_FAST_CASE(8) ORC_FAST_CASE(9) ORC_FAST_CASE(12)
#endif
            default: bad = 1; for (size_t b = 0; b < nb; b++) nl[b] = NAN;
        }
#undef ORC_FAST_CASE
        for (size_t b = 0; b < nb; b++) { if (nll_per_latent) nll_per_latent[l0 + b] = nl[b]; total += nl[b]; }
    }
    return bad ? NAN : total;
}

double orc_grad_stream(const orc_ihgp* g, size_t L, size_t T, const double* Ty, size_t ld, int layout,
                       double* x, double* dx, double* yhat, double* nll_per_latent, double* grad, int nthreads) {
    double total = 0.0;
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : total) schedule(static)
    for (long l = 0; l < (long)L; l++) {
        const orc_ihgp* gl = &g[l];
        int n = gl->d, P = gl->P;
        double xs[D], dxs[NP * D], xn[D], dxn[NP * D], gacc[NP] = {0, 0, 0}, gt[NP], acc = 0.0;
        for (int i = 0; i < n; i++) xs[i] = x[l * n + i];
        for (int i = 0; i < P * n; i++) dxs[i] = dx[l * P * n + i];
        for (size_t t = 0; t < T; t++) {
            double y = STREAM_AT(Ty, l, t), yh;
            orc_ihgp_step(gl, xs, 1, y, dxs, xn, &yh, dxn);           /* moihgp_online.h:64 */
            if (!isnan(y)) {
                acc += orc_ihgp_nll(gl, xs, y, dxs, gt);              /* moihgp_online.h:66 (pre-step x, dx) */
                for (int p = 0; p < P; p++) gacc[p] += gt[p];
            }
            for (int i = 0; i < n; i++) xs[i] = xn[i];
            for (int i = 0; i < P * n; i++) dxs[i] = dxn[i];
            if (yhat) { if (layout == 0) yhat[l * ld + t] = yh; else yhat[t * ld + l] = yh; }
        }
        for (int i = 0; i < n; i++) x[l * n + i] = xs[i];
        for (int i = 0; i < P * n; i++) dx[l * P * n + i] = dxs[i];
        for (int p = 0; p < P; p++) grad[l * P + p] = gacc[p];
        if (nll_per_latent) nll_per_latent[l] = acc;
        total += acc;
    }
    return total;
}

/* BASELINE.md variant (i): reference-shaped cost profile.  One call per (tick, latent) on
 * heap-allocated dynamic vectors, temporaries malloc'd per product like Eigen MatrixXd
 * expressions `AKHA * x + K * y` (ihgp.h:90) evaluate; serial loop of moihgp.h:367-373. */
typedef struct { int n; double* v; } dynvec;
static dynvec dv_new(int n) { dynvec r; r.n = n; r.v = (double*)malloc(sizeof(double) * n); return r; }
static void __attribute__((noinline)) refshaped_step(const orc_ihgp* g, const dynvec* x, double y, dynvec* xnew, double* yhat) {
    int n = g->d;
    if (isnan(y)) {
        dynvec t = dv_new(n);
        for (int i = 0; i < n; i++) { double s = 0.0; for (int k = 0; k < n; k++) s += g->A[i * n + k] * x->v[k]; t.v[i] = s; }
        memcpy(xnew->v, t.v, sizeof(double) * n); free(t.v);
    } else {
        dynvec t1 = dv_new(n), t2 = dv_new(n);
        for (int i = 0; i < n; i++) { double s = 0.0; for (int k = 0; k < n; k++) s += g->AKHA[i * n + k] * x->v[k]; t1.v[i] = s; }
        for (int i = 0; i < n; i++) t2.v[i] = g->K[i] * y;
        for (int i = 0; i < n; i++) xnew->v[i] = t1.v[i] + t2.v[i];
        free(t1.v); free(t2.v);
    }
    *yhat = xnew->v[0];
}
double orc_filter_stream_refshaped(const orc_ihgp* g, size_t L, size_t T, const double* Ty, size_t ld, int layout,
                                   double* x, double* yhat) {
    int n = g[0].d;
    dynvec* xs = (dynvec*)malloc(sizeof(dynvec) * L);
    dynvec* xn = (dynvec*)malloc(sizeof(dynvec) * L);
    for (size_t l = 0; l < L; l++) { xs[l] = dv_new(n); xn[l] = dv_new(n); memcpy(xs[l].v, x + l * n, sizeof(double) * n); }
    double chk = 0.0;
    for (size_t t = 0; t < T; t++) {
        for (size_t l = 0; l < L; l++) {
            double yh;
            refshaped_step(&g[l], &xs[l], STREAM_AT(Ty, l, t), &xn[l], &yh);
            if (yhat) { if (layout == 0) yhat[l * ld + t] = yh; else yhat[t * ld + l] = yh; }
            chk += yh;
        }
        for (size_t l = 0; l < L; l++) { dynvec tmp = xs[l]; xs[l] = xn[l]; xn[l] = tmp; }
    }
    for (size_t l = 0; l < L; l++) { memcpy(x + l * n, xs[l].v, sizeof(double) * n); free(xs[l].v); free(xn[l].v); }
    free(xs); free(xn);
    return chk;
}
