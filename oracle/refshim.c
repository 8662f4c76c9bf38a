/* refshim.c -- TEST INFRASTRUCTURE ONLY.  Exposes the C oracle (moihgp_oracle.c) under the reference's own C ABI
 * names (gp32_* / gp52_*, reference moihgp/src/wrapper.cpp:31-624) so that the reference's UNMODIFIED Python files
 * (pywrapper.py, online_learning.py), copied to a temporary directory at generation time, can run in this container and
 * produce golden vectors for the learner logic (oracle/gen_golden_learner.py).  Never part of the product. */
#include <stdbool.h>
#include <stddef.h>
#include "moihgp_oracle.h"

#define SHIM(PFX)                                                                                                   \
    void* PFX##_new(double dt, size_t M, size_t L, bool threading) { return orc_gp_new_t(ORC_MATERN32, dt, M, L, threading); } \
    void PFX##_del(void* g) { orc_gp_del((orc_gp*)g); }                                                              \
    void PFX##_step1(void* g, double* x, double* y, double* dx, double* xn, double* yh, double* dxn) { orc_gp_step1((orc_gp*)g, x, y, dx, xn, yh, dxn); } \
    void PFX##_step2(void* g, double* x, double* y, double* dx, double* xn, double* dxn) { orc_gp_step2((orc_gp*)g, x, y, dx, xn, dxn); } \
    void PFX##_step3(void* g, double* x, double* y, double* xn, double* yh) { orc_gp_step3((orc_gp*)g, x, y, xn, yh); } \
    void PFX##_step4(void* g, double* x, double* xn, double* yh) { orc_gp_step4((orc_gp*)g, x, xn, yh); }            \
    void PFX##_update(void* g, double* p) { orc_gp_update((orc_gp*)g, p); }                                          \
    double PFX##_lik1(void* g, double* x, double* y, double* dx, double* grad) { return orc_gp_lik1((orc_gp*)g, x, y, dx, grad); } \
    double PFX##_lik2(void* g, double* x, double* y) { return orc_gp_lik2((orc_gp*)g, x, y); }                       \
    void PFX##_get_params(void* g, double* p) { orc_gp_get_params((orc_gp*)g, p); }                                  \
    size_t PFX##_igp_dim(void* g) { return orc_gp_igp_dim((orc_gp*)g); }                                             \
    size_t PFX##_num_param(void* g) { return orc_gp_num_param((orc_gp*)g); }                                         \
    size_t PFX##_num_igp_param(void* g) { return orc_gp_num_igp_param((orc_gp*)g); }

SHIM(gp32)
SHIM(gp52) /* wrapper.cpp:22: GP52 is the Matern-3/2 model */
