/* moihgp_oracle.h -- CPU restatement of the MOIHGP hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (multioutputihgp_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference has no tests/fixtures and cannot be built here
 * (Eigen3 >= 3.3 absent).  See oracle/README.md.
 *
 * Citations `file:line` are into /root/reference/moihgp/include/.
 */
#ifndef MOIHGP_ORACLE_H_
#define MOIHGP_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Capacity of the fixed-size arrays below.  The default build (libmoihgp_oracle.so) covers the reference's two models;
 * `make wide` builds the same source with -DORC_DMAX=12 -DORC_PMAX=9 (libmoihgp_oracle_x.so) for the stacked models. */
#ifndef ORC_DMAX
#define ORC_DMAX 3          /* max per-latent state dim (Matern-5/2) */
#endif
#ifndef ORC_PMAX
#define ORC_PMAX 3          /* per-latent hyper-parameters (magnitude, lengthscale, noise) */
#endif

enum { ORC_MATERN32 = 0, ORC_MATERN52 = 1 };
/* Stacked state ("sum of J Matern components observed through one output", BASELINE.json's d=6 / d=12 configs): NOT a model
 * of the reference, but the plain composition its IHGP<StateSpace> template (ihgp.h:17-35) admits -- F, Pinf block-diagonal
 * from the reference's own component models, H = [H_1 .. H_J], params = [mag_1, len_1, .., mag_J, len_J, noise], P = 2J+1.
 * Kernel id = base | (J << 4), J >= 2; needs the wide build. */
#define ORC_STACK(base, J) ((base) | ((J) << 4))
int orc_dmax(void);
int orc_pmax(void);

/* Stationary matrices of one latent IHGP (moihgp/ihgp.h:243-254), row-major. */
typedef struct {
    int kernel, d, P;
    double dt;
    double params[ORC_PMAX];
    double A[ORC_DMAX * ORC_DMAX];
    double Q[ORC_DMAX * ORC_DMAX];
    double PP[ORC_DMAX * ORC_DMAX];
    double PF[ORC_DMAX * ORC_DMAX];
    double K[ORC_DMAX];
    double S;
    double HA[ORC_DMAX];
    double AKHA[ORC_DMAX * ORC_DMAX];
    double dA[ORC_PMAX][ORC_DMAX * ORC_DMAX];
    double dS[ORC_PMAX];
    double dK[ORC_PMAX][ORC_DMAX];
    double dAKHA[ORC_PMAX][ORC_DMAX * ORC_DMAX];
    double HdA[ORC_PMAX][ORC_DMAX];
    int dare_iters;
    int dlyap_iters[ORC_PMAX];
} orc_ihgp;

/* ---- L0/L1 ------------------------------------------------------------- */
void orc_expm(int n, const double* A, double* E);            /* Eigen MatrixBase::exp() restated */
int  orc_ihgp_update(orc_ihgp* g, int kernel, double dt, const double* params);   /* ihgp.h:117-201 */
/* IHGP<StateSpace>::update for an arbitrary StateSpace (ihgp.h:17-35 template argument): F, Pinf [d][d], H [d], R, and per
 * hyper-parameter dF, dPinf [P][d][d], dR [P], all row-major.  Used by the branch audit to reach every QLyap case of :141-185. */
int  orc_ihgp_update_ss(orc_ihgp* g, int d, int P, double dt, const double* F, const double* Pinf, const double* H, double R,
                        const double* dF, const double* dPinf, const double* dR);
/* ihgp.h:37-100; has_y=0 -> predict-only overload; dx/dxnew may be NULL */
void orc_ihgp_step(const orc_ihgp* g, const double* x, int has_y, double y, const double* dx,
                   double* xnew, double* yhat, double* dxnew);
/* ihgp.h:204-222; dx/grad may be NULL */
double orc_ihgp_nll(const orc_ihgp* g, const double* x, double y, const double* dx, double* grad);

/* ---- L2: MOIHGP mirror of the reference C ABI (src/wrapper.cpp:31-326) --- */
typedef struct orc_gp orc_gp;
orc_gp* orc_gp_new(int kernel, double dt, size_t num_output, size_t num_latent);   /* threading off (the reference's default) */
/* moihgp.h:81 with its `threading` argument; :128-135 forces it off for num_latent < 2.  The flag is observable: the
 * gradient overload of negLogLikelihood adds the per-latent losses only in its threaded branch (:590), not in the serial
 * one (:597-607), so lik1 returns the global terms of :503 alone when threading is off.  lik2 (:614-688) adds them in both. */
orc_gp* orc_gp_new_t(int kernel, double dt, size_t num_output, size_t num_latent, int threading);
void    orc_gp_set_threading(orc_gp* gp, int threading);
int     orc_gp_get_threading(orc_gp* gp);
void    orc_gp_del(orc_gp* gp);
void    orc_gp_step1(orc_gp* gp, const double* x, const double* y, const double* dx, double* xnew, double* yhat, double* dxnew);
void    orc_gp_step2(orc_gp* gp, const double* x, const double* y, const double* dx, double* xnew, double* dxnew);
void    orc_gp_step3(orc_gp* gp, const double* x, const double* y, double* xnew, double* yhat);
void    orc_gp_step4(orc_gp* gp, const double* x, double* xnew, double* yhat);
void    orc_gp_update(orc_gp* gp, const double* params);
double  orc_gp_lik1(orc_gp* gp, const double* x, const double* y, const double* dx, double* grad);
double  orc_gp_lik2(orc_gp* gp, const double* x, const double* y);
void    orc_gp_get_params(orc_gp* gp, double* params);
size_t  orc_gp_igp_dim(orc_gp* gp);
size_t  orc_gp_num_param(orc_gp* gp);
size_t  orc_gp_num_igp_param(orc_gp* gp);
/* introspection for tests */
const orc_ihgp* orc_gp_latent(orc_gp* gp, size_t l);
void    orc_gp_get_U(orc_gp* gp, double* U /* [M][L] */);
void    orc_gp_set_literal_ugrad(orc_gp* gp, int literal);   /* 1: moihgp.h:538-552 loop, 0: rank-1 closed form */
void    orc_gp_project(orc_gp* gp, const double* y, double* Ty);   /* moihgp.h:150-182 */

/* polar factor svdU * svdV^T of an M x L (M >= L) row-major matrix (moihgp.h:438-446);
 * sv (may be NULL) receives the singular values (unsorted). */
int orc_polar(size_t M, size_t L, const double* A, double* Upolar, double* sv);

/* ---- batched sweeps over pre-projected streams (the timed hot loop) ------ */
/* Ty is series-major [L][ld] (layout 0) or tick-major [T][ld] (layout 1).
 * For each latent: for t: v = y - HA x (pre-step); nll += .5(v^2/S + log S);
 * x <- AKHA x + K y; yhat_t = x[0].  NaN ticks: x <- A x, no NLL term.
 * x [L][d] in/out.  yhat may be NULL.  nll_per_latent [L] may be NULL.  Returns sum NLL. */
double orc_filter_stream(const orc_ihgp* g, size_t L, size_t T, const double* Ty, size_t ld, int layout,
                         double* x, double* yhat, double* nll_per_latent, int nthreads);
/* same with sensitivities: dx [L][P][d] in/out, grad [L][P] (sum over ticks of ihgp.h:219). */
double orc_grad_stream(const orc_ihgp* g, size_t L, size_t T, const double* Ty, size_t ld, int layout,
                       double* x, double* dx, double* yhat, double* nll_per_latent, double* grad, int nthreads);
/* fp32 state/stream variant of orc_filter_stream (fp64 NLL accumulation), for the fp32 configs */
double orc_filter_stream_f32(const orc_ihgp* g, size_t L, size_t T, const float* Ty, size_t ld, int layout,
                             float* x, float* yhat, double* nll_per_latent, int nthreads);

/* "reference-shaped" timing loop (BASELINE.md variant (i)): one call per tick per latent with
 * heap-allocated dynamic vectors/matrices, single thread, as moihgp.h:367-373 + ihgp.h:81-93 cost. */
double orc_filter_stream_refshaped(const orc_ihgp* g, size_t L, size_t T, const double* Ty, size_t ld, int layout,
                                   double* x, double* yhat);

/* fair-optimised CPU baseline (round 4): d-specialised, layout hoisted, FB latents per SIMD register; equal to orc_filter_stream to rounding */
double orc_filter_stream_fast(const orc_ihgp* g, size_t L, size_t T, const void* Ty, size_t ld, int layout, void* x, void* yhat,
                              double* nll_per_latent, int nthreads, int is_f32);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
