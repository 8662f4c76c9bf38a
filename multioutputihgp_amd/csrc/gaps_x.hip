// gaps_x.hip -- what the imputation sweep of recursion_x.hip (filter_x_gaps_a / _b_kernel: missing ticks of the stacked models' many-latent sweep, the
// method is described there) needs from outside: its scratch (the lists of gaps, one row per latent) and the filters' scalar impulse responses
// s_k = HA AKHA^k K -- the ordinary sweep run over a unit observation at tick 0 from a zero state, writing predicted observations -- once per
// parameter update.
#include "kernels_common.h"

namespace moihgp {
namespace {

template <typename T>
__global__ void __launch_bounds__(256) gap_impulse_kernel(T* __restrict__ imp_in, size_t L) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l < L) imp_in[l * kGapSMax] = T(1);
}

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// ---- fp32 banks: latents whose fp32 scan tables are unusable (a mildly unstable filter: its largest table entry leaves 1e18) while the fp64 ones
// are fine -- swept in fp64 on the side (capi.cpp) instead of tick by tick in the fp32 kernel, where one of them holds the whole launch (4096 x
// 10^4 Matern32x2: 7 such latents, 302 us against 80) ----
__global__ void __launch_bounds__(256) rescue_gather_kernel(const float* __restrict__ Ty, size_t T, size_t ld, const int* __restrict__ idx, const double* __restrict__ cb64, int xcs,
                                                            const float* __restrict__ xin, int d, double* __restrict__ rows, double* __restrict__ cbc, double* __restrict__ xc) {
    const size_t c = blockIdx.x, l = (size_t)idx[c];
    if (Ty) {                                                           // per sweep: the stream rows and start states
        for (size_t t = (size_t)blockIdx.y * 256 + threadIdx.x; t < T; t += (size_t)gridDim.y * 256) rows[c * ld + t] = (double)Ty[l * ld + t];
        if (blockIdx.y == 0 && (int)threadIdx.x < d) xc[c * d + threadIdx.x] = (double)xin[l * d + threadIdx.x];
    } else if (blockIdx.y == 0) {                                       // per update: the constant blocks
        for (int e = threadIdx.x; e < xcs; e += 256) cbc[c * xcs + e] = cb64[l * xcs + e];
    }
}
__global__ void __launch_bounds__(256) rescue_scatter_kernel(const int* __restrict__ idx, const double* __restrict__ rows, size_t T, size_t ld, const double* __restrict__ xc, int d,
                                                             const double* __restrict__ nllc, float* __restrict__ yhat, size_t ldo, float* __restrict__ x, double* __restrict__ nll) {
    const size_t c = blockIdx.x, l = (size_t)idx[c];
    if (yhat)
        for (size_t t = (size_t)blockIdx.y * 256 + threadIdx.x; t < T; t += (size_t)gridDim.y * 256) yhat[l * ldo + t] = (float)rows[c * ld + t];
    if (blockIdx.y == 0) {
        if ((int)threadIdx.x < d) x[l * d + threadIdx.x] = (float)xc[c * d + threadIdx.x];
        if (threadIdx.x == 0 && nll) nll[l] = nllc[c];
    }
}

}  // namespace

void launch_rescue_gather(const float* Ty, size_t T, size_t ld, const int* idx, size_t n, const double* cb64, int xcs, const float* xin, int d,
                          double* rows, double* cbc, double* xc, hipStream_t s) {
    const unsigned parts = Ty ? (unsigned)((T + 4095) / 4096 ? (T + 4095) / 4096 : 1) : 1u;
    hipLaunchKernelGGL(rescue_gather_kernel, dim3((unsigned)n, parts), dim3(256), 0, s, Ty, T, ld, idx, cb64, xcs, xin, d, rows, cbc, xc);
}
void launch_rescue_scatter(const int* idx, size_t n, const double* rows, size_t T, size_t ld, const double* xc, int d, const double* nllc,
                           float* yhat, size_t ldo, float* x, double* nll, hipStream_t s) {
    const unsigned parts = (unsigned)((T + 4095) / 4096 ? (T + 4095) / 4096 : 1);
    hipLaunchKernelGGL(rescue_scatter_kernel, dim3((unsigned)n, parts), dim3(256), 0, s, idx, rows, T, ld, xc, d, nllc, yhat, ldo, x, nll);
}

size_t gap_bank_bytes(int d, int dtype, size_t L, size_t T) {
    const size_t es = dtype == 0 ? 8 : 4;
    const size_t gcap = (T + 63) / 64 * 64;
    return 2 * up256(L * (size_t)kGapSMax * es) + 2 * up256(L * (size_t)d * es) + up256(L * sizeof(int)) + up256(L * gcap * sizeof(int)) + 2 * up256(L * gcap * es);
}

GapBank gap_bank_carve(void* base, int d, int dtype, size_t L, size_t T) {
    const size_t es = dtype == 0 ? 8 : 4;
    unsigned char* p = static_cast<unsigned char*>(base);
    GapBank b;
    b.gcap = (T + 63) / 64 * 64;
    b.imp_in = p; p += up256(L * (size_t)kGapSMax * es);               // (the constant parts first: they stay put when only T changes)
    b.imp_out = p; p += up256(L * (size_t)kGapSMax * es);
    b.xz = p; p += up256(L * (size_t)d * es);
    b.x1 = p; p += up256(L * (size_t)d * es);
    b.gstat = reinterpret_cast<int*>(p); p += up256(L * sizeof(int));
    b.gpos = reinterpret_cast<int*>(p); p += up256(L * b.gcap * sizeof(int));
    b.gval = p; p += up256(L * b.gcap * es);
    b.gw = p;
    return b;
}

// the constant parts of a freshly carved bank: unit impulses, a zero start state.  Asynchronous on `s`.
int gap_bank_init(const GapBank& b, int d, int dtype, size_t L, hipStream_t s) {
    const size_t es = dtype == 0 ? 8 : 4;
    if (hipMemsetAsync(b.imp_in, 0, L * (size_t)kGapSMax * es, s) != hipSuccess || hipMemsetAsync(b.xz, 0, L * (size_t)d * es, s) != hipSuccess) {
        set_last_error("gap bank: memset failed"); return 2;
    }
    if (dtype == 0) hipLaunchKernelGGL(gap_impulse_kernel<double>, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, (double*)b.imp_in, L);
    else hipLaunchKernelGGL(gap_impulse_kernel<float>, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, (float*)b.imp_in, L);
    return 0;
}

// The impulse responses of all L filters into b.imp_out: b.imp_out[l][k + 1] = s_k of latent l.  Asynchronous on `s`.
int launch_gap_impulse(const GapBank& b, int kernel, int dtype, size_t L, const double* cb64, const float* cb32, hipStream_t s) {
    return launch_filter_stream_x(kernel, dtype, b.imp_in, kGapSMax, kGapSMax, L, cb64, cb32, b.xz, b.x1, b.imp_out, nullptr, s, nullptr, nullptr, nullptr, 0,
                                  -6, kGapSMax, nullptr, nullptr, nullptr, -1, 0, nullptr, nullptr);
}

}  // namespace moihgp
