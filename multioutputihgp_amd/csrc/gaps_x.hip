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

}  // namespace

size_t gap_bank_bytes(int d, int dtype, size_t L, size_t T) {
    const size_t es = dtype == 0 ? 8 : 4;
    const size_t gcap = (T + 63) / 64 * 64;
    return 2 * up256(L * (size_t)kGapSMax * es) + 2 * up256(L * (size_t)d * es) + up256(L * sizeof(int)) + up256(L * gcap * sizeof(int)) + up256(L * gcap * es);
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
    b.gval = p;
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
