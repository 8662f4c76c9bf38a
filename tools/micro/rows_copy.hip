// Does the ACCESS PATTERN of the many-latent sweep cap its cold-stream rate?  A copy with the sweep's pattern and none of its arithmetic:
// 4096 wavefronts, each walking its own row of a [L][ld] array (40 KB apart) in pieces of G x 1 KB per step (the sweep: G = 4), loads of
// step k+1 in flight while step k is stored -- against (b) the same bytes laid out segment-major [T / seg][L][seg], where the wavefronts
// of a launch walk ONE contiguous front like a plain copy, and (c) the plain linear copy.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/rows_copy.hip -o /tmp/rows_copy && /tmp/rows_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// one wave per row; G vectors of 16 B per lane per step (G KB per wave-step); row-major: step s of row l at l*ld + s*G*256 floats;
// segment-major: at (s*L + l)*G*256 floats
template <int G, bool SEGMAJOR>
__global__ void __launch_bounds__(256, 4) rows_copy(const float* __restrict__ in, float* __restrict__ out, size_t L, size_t ld, int nstep) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t l = (size_t)blockIdx.x * 4 + wave;
    if (l >= L) return;
    auto src = [&](int s) { return SEGMAJOR ? in + ((size_t)s * L + l) * (G * 256) : in + l * ld + (size_t)s * (G * 256); };
    auto dst = [&](int s) { return SEGMAJOR ? out + ((size_t)s * L + l) * (G * 256) : out + l * ld + (size_t)s * (G * 256); };
    f4 cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; i++) cur[i] = *reinterpret_cast<const f4*>(src(0) + (i * 64 + lane) * 4);
    for (int s = 0; s < nstep; s++) {
        if (s + 1 < nstep) {
#pragma unroll
            for (int i = 0; i < G; i++) nxt[i] = *reinterpret_cast<const f4*>(src(s + 1) + (i * 64 + lane) * 4);
        }
#pragma unroll
        for (int i = 0; i < G; i++) { f4 v = cur[i]; v.x += 1.0f; __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst(s) + (i * 64 + lane) * 4)); }
#pragma unroll
        for (int i = 0; i < G; i++) cur[i] = nxt[i];
    }
}

template <int G, bool SM>
double run(const std::vector<float*>& in, const std::vector<float*>& out, size_t L, size_t T, int reps) {
    const int nstep = (int)(T / (G * 256));
    const size_t ld = T;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((rows_copy<G, SM>), dim3((unsigned)(L / 4)), dim3(256), 0, 0, in[r % in.size()], out[r % out.size()], L, ld, nstep);
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((rows_copy<G, SM>), dim3((unsigned)(L / 4)), dim3(256), 0, 0, in[r % in.size()], out[r % out.size()], L, ld, nstep);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 2.0 * L * (double)nstep * G * 1024 * reps / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t L = 4096, T = 10240;                      // 10 sweep segments of 1024 floats per row: 168 MB in, 168 MB out
    const size_t bytes = L * T * 4;
    for (int nbuf : {1, 6}) {
        std::vector<float*> in(nbuf), out(nbuf);
        for (int i = 0; i < nbuf; i++) { hipMalloc(&in[i], bytes); hipMalloc(&out[i], bytes); hipMemset(in[i], 0, bytes); }
        printf("%d buffer pair(s) of %zu MB (%s):\n", nbuf, bytes >> 20, nbuf == 1 ? "resident" : "rotating = cold");
        printf("  row-major [L][T], one wave per row:      1 KB/step %.2f   4 KB/step %.2f   8 KB/step %.2f TB/s\n", run<1, false>(in, out, L, T, 40), run<4, false>(in, out, L, T, 40), run<8, false>(in, out, L, T, 40));
        printf("  segment-major [T/seg][L][seg]:           1 KB/step %.2f   4 KB/step %.2f   8 KB/step %.2f TB/s\n", run<1, true>(in, out, L, T, 40), run<4, true>(in, out, L, T, 40), run<8, true>(in, out, L, T, 40));
        for (int i = 0; i < nbuf; i++) { hipFree(in[i]); hipFree(out[i]); }
    }
    return 0;
}
