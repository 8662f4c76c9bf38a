// Raw copy bandwidth of this box with hand-written kernels (the ceiling any streaming kernel can be measured against).
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/copy_bench.hip -o /tmp/copy_bench && /tmp/copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float nt4 __attribute__((ext_vector_type(4)));
template <bool NT, int UNROLL>
__global__ void __launch_bounds__(256) copy_kernel(const nt4* __restrict__ in, nt4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    nt4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (i + (size_t)u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(in + i + (size_t)u * 256) : in[i + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (i + (size_t)u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], out + i + (size_t)u * 256); else out[i + (size_t)u * 256] = v[u]; }
}
template <bool NT, int UNROLL>
double run(const std::vector<nt4*>& in, const std::vector<nt4*>& out, size_t n, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned grid = (unsigned)((n + 256 * UNROLL - 1) / (256 * UNROLL));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((copy_kernel<NT, UNROLL>), dim3(grid), dim3(256), 0, 0, in[r % in.size()], out[r % out.size()], n);
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((copy_kernel<NT, UNROLL>), dim3(grid), dim3(256), 0, 0, in[r % in.size()], out[r % out.size()], n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 2.0 * n * 16 * reps / (ms * 1e-3) / 1e12;
}
int main() {
    const size_t bytes = 164u << 20;                       // one C3 stream (fp32): 164 MB in, 164 MB out
    const size_t n = bytes / 16;
    for (int nbuf : {1, 6}) {                              // 1: the same pair every time (Infinity-Cache assisted); 6: rotating pairs (cold)
        std::vector<nt4*> in(nbuf), out(nbuf);
        for (int i = 0; i < nbuf; i++) { hipMalloc(&in[i], bytes); hipMalloc(&out[i], bytes); hipMemset(in[i], 1, bytes); }
        printf("%d buffer pair(s) of %zu MB:\n", nbuf, bytes >> 20);
        printf("  plain  unroll 1: %.2f TB/s   unroll 4: %.2f   unroll 8: %.2f\n", run<false, 1>(in, out, n, 60), run<false, 4>(in, out, n, 60), run<false, 8>(in, out, n, 60));
        printf("  nontmp unroll 1: %.2f TB/s   unroll 4: %.2f   unroll 8: %.2f\n", run<true, 1>(in, out, n, 60), run<true, 4>(in, out, n, 60), run<true, 8>(in, out, n, 60));
        for (int i = 0; i < nbuf; i++) { hipFree(in[i]); hipFree(out[i]); }
    }
    return 0;
}
