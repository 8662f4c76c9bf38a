// stack_dispatch.hip -- what the stacked filter's translation units share: the dispatcher over the per-model units of recursion_x.hip
// (one unit per (DB, J), built in parallel by the Makefile) and the update-time table of the few-latents team kernel.
#include "x_common.h"

namespace moihgp {
namespace {

// The scan powers of the chunk-templated team kernel, per latent and chunk length: M^(2^k), M = AKHA^CK, k = 0..6 (fp64 products through
// LDS, one or a few entries per lane), and what update() derives from such powers for 32-tick chunks (stationary_x.hip, same bounds): the
// number of levels that matter in each precision, whether M^64 still does ("decays"), whether every power stays inside the format
// ("tame": 1e150 / 1e18).  Run once per IHGP::update for banks small enough for the team kernel (a dozen dependent D x D products:
// 3.5 us when the sweep kernel formed them itself, per launch -- more than the stream's load latency hides).
template <int D>
__global__ void __launch_bounds__(64) team_powers_kernel(const double* __restrict__ cb64, double* __restrict__ tp64, float* __restrict__ tp32) {
    using Lay = XC<D>;
    constexpr int NN = D * D, TPL = team_powers_len<D>();
    __shared__ double pwr[4 * Lay::LS];
    const int lane = threadIdx.x, cki = blockIdx.y, CK = 16 + 4 * cki;
    const size_t l = blockIdx.x;
    const double* __restrict__ c64 = cb64 + l * Lay::SIZE;
    double* o64 = tp64 + (l * kTeamNck + cki) * TPL;
    float* o32 = tp32 + (l * kTeamNck + cki) * TPL;
    auto mm = [&](double* dst, const double* a, const double* b) {                      // dst = a b   (dst distinct from both)
        for (int e = lane; e < NN; e += 64) {
            const int i = e / D, j = e % D;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) acc = fma(a[i * D + k], b[k * D + j], acc);
            dst[e] = acc;
        }
        wave_lds_fence();
    };
    double *sq = pwr, *sq2 = pwr + Lay::LS, *m = pwr + 2 * Lay::LS, *m2 = pwr + 3 * Lay::LS;
    for (int e = lane; e < NN; e += 64) sq[e] = c64[Lay::AKHA + e];
    wave_lds_fence();
    bool have = false;
    for (int k = 0; (CK >> k) != 0; k++) {                                              // AKHA^CK by squaring and multiplying
        if ((CK >> k) & 1) {
            if (!have) { for (int e = lane; e < NN; e += 64) m[e] = sq[e]; wave_lds_fence(); have = true; }
            else { mm(m2, m, sq); double* t = m; m = m2; m2 = t; }
        }
        if ((CK >> (k + 1)) != 0) { mm(sq2, sq, sq); double* t = sq; sq = sq2; sq2 = t; }
    }
    bool tame64 = true, tame32 = true;
    int nlv64 = 1, nlv32 = 1, dec64 = 1, dec32 = 1;
    for (int lv = 0; lv < 7; lv++) {
        double big = 0.0;
        for (int e = lane; e < Lay::LS; e += 64) {
            const double v = e < NN ? m[e] : 0.0;
            o64[lv * Lay::LS + e] = v;
            o32[lv * Lay::LS + e] = (float)v;
            tame64 = tame64 && (fabs(v) < 1e150);                                       // (false for NaN too)
            tame32 = tame32 && (fabs(v) < 1e18);
            big = fmax(big, fabs(v));
        }
        for (int o = 32; o >= 1; o >>= 1) big = fmax(big, __shfl_xor(big, o, 64));
        if (big * D >= 1e-20) { if (lv < 6) nlv64 = lv + 1; else dec64 = 0; }
        if (big * D >= 1e-10) { if (lv < 6) nlv32 = lv + 1; else dec32 = 0; }
        if (lv < 6) { mm(m2, m, m); double* t = m; m = m2; m2 = t; }
    }
    tame64 = __builtin_amdgcn_ballot_w64(!tame64) == 0;
    tame32 = __builtin_amdgcn_ballot_w64(!tame32) == 0;
    if (lane < 16) {
        o64[7 * Lay::LS + lane] = lane == 0 ? (double)nlv64 : lane == 1 ? (double)dec64 : lane == 2 ? (tame64 ? 1.0 : 0.0) : 0.0;
        o32[7 * Lay::LS + lane] = lane == 0 ? (float)nlv32 : lane == 1 ? (float)dec32 : lane == 2 ? (tame32 ? 1.0f : 0.0f) : 0.0f;
    }
}

}  // namespace

// the per-model units (recursion_x.hip compiled with -DMOIHGP_X_TU=DBJ)
#define MOIHGP_X_DECL(DBJ)                                                                                                                           \
    int launch_filter_x_##DBJ(int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const float* cb32, const void* xin, void* x, \
                              void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, double* scratch, size_t scratch_len,     \
                              int force_slices, size_t ldo, int* link_flags, double* link_state, double* total, int max_links, int team_mode,       \
                              const double* tp64, const float* tp32)
MOIHGP_X_DECL(21); MOIHGP_X_DECL(31); MOIHGP_X_DECL(22); MOIHGP_X_DECL(23); MOIHGP_X_DECL(24); MOIHGP_X_DECL(32); MOIHGP_X_DECL(33); MOIHGP_X_DECL(34);
#undef MOIHGP_X_DECL

void launch_team_powers(int kernel, const double* cb64, size_t L, double* tp64, float* tp32, hipStream_t stream) {
    if (L == 0) return;
    const int d = (kernel_base(kernel) == 0 ? 2 : 3) * kernel_stack(kernel);
    dim3 grid((unsigned)L, kTeamNck);
    switch (d) {
        case 2: hipLaunchKernelGGL(team_powers_kernel<2>, grid, dim3(64), 0, stream, cb64, tp64, tp32); break;
        case 3: hipLaunchKernelGGL(team_powers_kernel<3>, grid, dim3(64), 0, stream, cb64, tp64, tp32); break;
        case 4: hipLaunchKernelGGL(team_powers_kernel<4>, grid, dim3(64), 0, stream, cb64, tp64, tp32); break;
        case 6: hipLaunchKernelGGL(team_powers_kernel<6>, grid, dim3(64), 0, stream, cb64, tp64, tp32); break;
        case 8: hipLaunchKernelGGL(team_powers_kernel<8>, grid, dim3(64), 0, stream, cb64, tp64, tp32); break;
        case 9: hipLaunchKernelGGL(team_powers_kernel<9>, grid, dim3(64), 0, stream, cb64, tp64, tp32); break;
        case 12: hipLaunchKernelGGL(team_powers_kernel<12>, grid, dim3(64), 0, stream, cb64, tp64, tp32); break;
        default: break;
    }
    MOIHGP_HIP_FATAL(hipGetLastError());
}

int launch_filter_teamc_plain(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* xc64, const float* xc32, const double* tp64, const float* tp32,
                              const void* xin, void* x, void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, size_t ldo, double* total, int team_mode) {
    if (!xc64 || !tp64) return -1;
    return d == 2 ? launch_filter_x_21(dtype, Ty, T, ld, L, xc64, xc32, xin, x, yhat, nll, stream, ev0, ev1, nullptr, 0, -1, ldo, nullptr, nullptr, total, -1, team_mode, tp64, tp32)
                  : launch_filter_x_31(dtype, Ty, T, ld, L, xc64, xc32, xin, x, yhat, nll, stream, ev0, ev1, nullptr, 0, -1, ldo, nullptr, nullptr, total, -1, team_mode, tp64, tp32);
}

int launch_filter_stream_x(int kernel, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const float* cb32,
                           const void* xin, void* x, void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                           double* scratch, size_t scratch_len, int force_slices, size_t ldo, int* link_flags, double* link_state, double* total, int max_links, int team_mode,
                           const double* tp64, const float* tp32) {
    if (L == 0) return 0;
    if (ldo == 0) ldo = ld;
    const int base = kernel_base(kernel), J = kernel_stack(kernel);
#define MOIHGP_X_CASE(DBB, JJ)                                                                                        \
    if (base == (DBB == 2 ? 0 : 1) && J == JJ)                                                                        \
        return launch_filter_x_##DBB##JJ(dtype, Ty, T, ld, L, cb64, cb32, xin, x, yhat, nll, stream, ev0, ev1, scratch, scratch_len, force_slices, ldo, link_flags, link_state, total, max_links, team_mode, tp64, tp32)
    MOIHGP_X_CASE(2, 1); MOIHGP_X_CASE(3, 1);            // (the reference's own models in the stacked layout: launch_xc_from_cb)
    MOIHGP_X_CASE(2, 2); MOIHGP_X_CASE(2, 3); MOIHGP_X_CASE(2, 4);
    MOIHGP_X_CASE(3, 2); MOIHGP_X_CASE(3, 3); MOIHGP_X_CASE(3, 4);
#undef MOIHGP_X_CASE
    set_last_error("stacked kernel id %d is not built", kernel);
    return 1;
}

}  // namespace moihgp
