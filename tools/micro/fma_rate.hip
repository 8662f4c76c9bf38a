// Issue rate of fp64 / fp32 multiply-adds in the forms the stacked kernels use: plain (VGPR operands), with a scalar operand, and with
// a DPP row broadcast operand (v_fmac_f64_dpp row_newbcast).  One wave per SIMD..8 waves per SIMD, 16 independent accumulators.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/fma_rate.hip -o /tmp/fma_rate 2>/dev/null && /tmp/fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NACC = 16, ITERS = 4096;
template <typename T, int MODE>   // 0 plain, 1 scalar operand, 2 DPP broadcast operand
__global__ void __launch_bounds__(64) k(T* out, const T* in, T s) {
    T acc[NACC], slab = in[threadIdx.x & 15], x = in[16 + (threadIdx.x & 63)];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = in[i];
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (MODE == 0) acc[i] = __builtin_fma(slab, x, acc[i]);
            else if (MODE == 1) acc[i] = __builtin_fma(s, x, acc[i]);
            else if (sizeof(T) == 8) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(slab), "v"(x));
            else asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(slab), "v"(x));
        }
    }
    T r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <typename T, int MODE>
void run(const char* name, int waves_per_simd) {
    T *out, *in;
    const int nblk = 256 * 4 * waves_per_simd;
    (void)hipMalloc(&out, sizeof(T) * 64 * nblk); (void)hipMalloc(&in, sizeof(T) * 128); (void)hipMemset(in, 0, sizeof(T) * 128);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, MODE>), dim3(nblk), dim3(64), 0, 0, out, in, (T)1.0000001);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<T, MODE>), dim3(nblk), dim3(64), 0, 0, out, in, (T)1.0000001);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = (double)ITERS * NACC * waves_per_simd;
    printf("%-28s %d waves/SIMD: %.2f cycles per wave-instruction (at 2.4 GHz), %.1f TFLOP/s\n", name, waves_per_simd,
           ms * 1e-3 * 2.4e9 / insts_per_simd, 2.0 * 64 * ITERS * NACC * nblk / (ms * 1e-3) / 1e12);
    (void)hipFree(out); (void)hipFree(in);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<double, 0>("fp64 fma plain", w); run<double, 1>("fp64 fma scalar operand", w); run<double, 2>("fp64 fmac dpp row_newbcast", w);
        run<float, 0>("fp32 fma plain", w); run<float, 1>("fp32 fma scalar operand", w); run<float, 2>("fp32 fmac dpp row_newbcast", w);
    }
    return 0;
}
