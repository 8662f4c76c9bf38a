// grad_x.hip -- sensitivity / gradient sweep (ihgp.h:37-57 with :212-222, A2 + A5) for STACKED models, state dim D <= 12 and
// P = 2J + 1 <= 9 hyper-parameters.
//
// A first, plain mapping (the filter of these models is in recursion_x.hip; this sweep is what learning their
// hyper-parameters needs): one wavefront per latent walks the stream tick by tick; the (P + 1) D state entries -- x and the
// P sensitivity vectors dx_p -- are spread over the lanes, two per lane at most, and every entry is one row of
//     x'    = AKHA x + K y                                   (ihgp.h:50; A x for a missing tick, :41)
//     dx_p' = dAKHA_p x + AKHA dx_p + dK_p y                 (ihgp.h:54; dA_p x + A dx_p, :45)
// evaluated from the latent's matrices, which sit in LDS for the whole sweep together with a double-buffered copy of the
// state; the stream is staged through LDS 64 ticks at a time.  The NLL and its gradient use the PRE-step state (ihgp.h:206,
// :216-219); lanes 0 .. P-1 carry one gradient entry each.  All arithmetic is fp64 whatever the stream's type.
// Latency-bound by construction (about 0.25 us per tick at D = 12): fine for the learners' windows, slow for long streams.
#include "kernels_common.h"
#include <cstdlib>

namespace moihgp {
namespace {

template <typename T, int DB, int J>
__global__ void __launch_bounds__(64)
grad_x_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const double* __restrict__ cb64, const double* __restrict__ cbd64,
              T* __restrict__ x, T* __restrict__ dx, T* __restrict__ yhat, double* __restrict__ nll, double* __restrict__ grad, int out_mode,
              const int* __restrict__ only /* NULL: every latent whole */, size_t t_cont) {
    constexpr int D = DB * J, NN = D * D, P = 2 * J + 1, NE = (P + 1) * D;
    using Lc = XC<D>;
    using Ld = XD<D, P>;
    extern __shared__ double sm[];
    double* sAKHA = sm;                 // [NN]
    double* sA = sAKHA + NN;            // [NN]
    double* sdAKHA = sA + NN;           // [P][NN]
    double* sdA = sdAKHA + P * NN;      // [P][NN]
    double* sK = sdA + P * NN;          // [D]
    double* sHA = sK + D;               // [D]
    double* sdK = sHA + D;              // [P][D]
    double* sHdA = sdK + P * D;         // [P][D]
    double* sdS = sHdA + P * D;         // [P]
    double* sst = sdS + P;              // [2][NE]   state: x (D) then dx (P*D), double buffered
    double* sy = sst + 2 * NE;          // [64]
    double* syh = sy + 64;              // [64]
    const int lane = threadIdx.x;
    const size_t l = blockIdx.x;
    if (l >= L) return;
    // after grad_scan_x_kernel: a flagged latent is swept whole; the others continue at tick t_cont from the carried (x, dx) and
    // ADD to the sums that kernel left in nll / grad
    const bool cont = only && !only[l];
    if (cont && t_cont >= Tlen) return;
    const size_t t_first = cont ? t_cont : 0;
    const double* c = cb64 + l * Lc::SIZE;
    const double* cd = cbd64 + l * Ld::SIZE;
    for (int e = lane; e < NN; e += 64) { sAKHA[e] = c[Lc::AKHA + e]; sA[e] = c[Lc::A + e]; }
    for (int e = lane; e < P * NN; e += 64) { sdAKHA[e] = cd[Ld::DAKHA + e]; sdA[e] = cd[Ld::DA + e]; }
    for (int e = lane; e < D; e += 64) { sK[e] = c[Lc::K + e]; sHA[e] = c[Lc::HA + e]; }
    for (int e = lane; e < P * D; e += 64) { sdK[e] = cd[Ld::DK + e]; sHdA[e] = cd[Ld::HDA + e]; }
    for (int e = lane; e < P; e += 64) sdS[e] = cd[Ld::DS + e];
    for (int e = lane; e < NE; e += 64) sst[e] = e < D ? (double)x[l * D + e] : (double)dx[l * P * D + (e - D)];
    const double S = c[Lc::S], logS = c[Lc::LOGS];
    wave_lds_fence();

    double acc = 0.0, g = 0.0;          // lane 0: sum of NLL terms; lanes 0..P-1: gradient entry p
    int cur = 0;
    const T* row = Ty + l * ld;
    T* orow = yhat ? yhat + l * ld : nullptr;
    for (size_t t0 = t_first; t0 < Tlen; t0 += 64) {
        const int nt = (int)(Tlen - t0 < 64 ? Tlen - t0 : 64);
        if (lane < nt) sy[lane] = (double)row[t0 + lane];
        wave_lds_fence();
        for (int k = 0; k < nt; k++) {
            const double y = sy[k];
            const bool miss = (y != y);
            const double* st = sst + cur * NE;
            double* sn = sst + (cur ^ 1) * NE;
            if (!miss && lane < P) {                                 // ihgp.h:206-207, :216-219 on the pre-step state
                double hx = 0.0;
                for (int j = 0; j < D; j++) hx += sHA[j] * st[j];
                const double v = y - hx;
                double a = 0.0, b = 0.0;
                for (int j = 0; j < D; j++) { a += sHdA[lane * D + j] * st[j]; b += sHA[j] * st[D + lane * D + j]; }
                const double dv = -a - b;
                g += (v * dv - 0.5 * (v * v / S - 1) * sdS[lane]) / S;
                if (lane == 0) acc += 0.5 * (v * v / S + logS);
            }
            if (out_mode == 2 && lane == 0) {                          // predicted mean HA x_t of the pre-step state (window objective: pv of moihgp.h:510)
                double hx = 0.0;
                for (int j = 0; j < D; j++) hx += sHA[j] * st[j];
                syh[k] = hx;
            }
            const double* Mx = miss ? sA : sAKHA;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int e = lane + 64 * q;
                if (e < NE) {
                    double v;
                    if (e < D) {
                        double s = 0.0;
                        for (int j = 0; j < D; j++) s += Mx[e * D + j] * st[j];
                        v = miss ? s : s + sK[e] * y;                // ihgp.h:41 / :50
                    } else {
                        const int p = (e - D) / D, i = (e - D) % D;
                        const double* dM = (miss ? sdA : sdAKHA) + p * NN;
                        double a = 0.0, b = 0.0;
                        for (int j = 0; j < D; j++) { a += dM[i * D + j] * st[j]; b += Mx[i * D + j] * st[D + p * D + j]; }
                        v = a + b;
                        if (!miss) v += sdK[p * D + i] * y;          // ihgp.h:45 / :54
                    }
                    sn[e] = v;
                    if (e == 0 && out_mode != 2) syh[k] = v;         // ihgp.h:51 `yhat = xnew(0, 0)`
                }
            }
            wave_lds_fence();
            cur ^= 1;
        }
        if (orow && lane < nt) orow[t0 + lane] = (T)syh[lane];
        wave_lds_fence();
    }
    const double* st = sst + cur * NE;
    for (int e = lane; e < NE; e += 64) {
        if (e < D) x[l * D + e] = (T)st[e];
        else dx[l * P * D + (e - D)] = (T)st[e];
    }
    if (lane == 0 && nll) nll[l] = cont ? nll[l] + acc : acc;
    if (lane < P) grad[l * P + lane] = cont ? grad[l * P + lane] + g : g;
}

template <typename T, int DB, int J>
int launch_gx(const void* Ty, size_t Tlen, size_t ld, size_t L, const double* cb64, const double* cbd64, void* x, void* dx, void* yhat,
              double* nll, double* grad, hipStream_t stream, int out_mode, const int* only, size_t t_cont) {
    constexpr int D = DB * J, NN = D * D, P = 2 * J + 1, NE = (P + 1) * D;
    const size_t smem = (size_t)(2 * NN + 2 * P * NN + 2 * D + 2 * P * D + P + 2 * NE + 128) * sizeof(double);
    hipLaunchKernelGGL((grad_x_kernel<T, DB, J>), dim3((unsigned)L), dim3(64), smem, stream, (const T*)Ty, Tlen, ld, L, cb64, cbd64,
                       (T*)x, (T*)dx, (T*)yhat, nll, grad, out_mode, only, t_cont);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("grad_x_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

int launch_grad_stream_x(int kernel, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const double* cbd64,
                         void* x, void* dx, void* yhat, double* nll, double* grad, hipStream_t stream, int out_mode, int* flags, double* hp, int hp_build) {
    if (L == 0) return 0;
    // long streams: the time-parallel sweep (grad_scan_x.hip) takes the whole 32-tick chunks of every latent it can; what it leaves
    // (flags[l] = 1: missing ticks, unusable scan tables; the last T mod 32 ticks of the others) is walked tick by tick here
    static const size_t scan_from = [] { const char* e = std::getenv("MOIHGP_GRADX_SCAN_FROM"); return e ? (size_t)std::atoll(e) : (size_t)512; }();
    const int* only = nullptr;
    size_t t_cont = 0;
    if (flags && hp && T >= scan_from) {
        t_cont = T / kChunkX * kChunkX;
        if (int rc = launch_grad_scan_x(kernel, dtype, Ty, t_cont, ld, L, cb64, cbd64, x, dx, yhat, nll, grad, flags, hp, stream, out_mode, hp_build)) return rc;
        only = flags;
    }
    const int base = kernel_base(kernel), J = kernel_stack(kernel);
#define MOIHGP_GX_CASE(DBB, JJ)                                                                                          \
    if (base == (DBB == 2 ? 0 : 1) && J == JJ)                                                                           \
        return dtype == 0 ? launch_gx<double, DBB, JJ>(Ty, T, ld, L, cb64, cbd64, x, dx, yhat, nll, grad, stream, out_mode, only, t_cont) \
                          : launch_gx<float, DBB, JJ>(Ty, T, ld, L, cb64, cbd64, x, dx, yhat, nll, grad, stream, out_mode, only, t_cont)
    MOIHGP_GX_CASE(2, 2); MOIHGP_GX_CASE(2, 3); MOIHGP_GX_CASE(2, 4);
    MOIHGP_GX_CASE(3, 2); MOIHGP_GX_CASE(3, 3); MOIHGP_GX_CASE(3, 4);
#undef MOIHGP_GX_CASE
    set_last_error("stacked kernel id %d is not built", kernel);
    return 1;
}

}  // namespace moihgp
