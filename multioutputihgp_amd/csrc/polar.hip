// polar.hip -- polar factor  svdU * svdV^T  of the mixing matrix on the device (reference moihgp.h:433-447,
// where Eigen's BDCSVD/JacobiSVD is used only to form that product).  The factor is unique for full column
// rank, so any convergent method reproduces it to rounding; here Newton-Schulz:
//     X_0 = A / s  (s >= sigma_max),   X_{k+1} = X_k (3/2 I - 1/2 X_k^T X_k)
// two MFMA GEMMs per step (gemm_mfma.hip), quadratic convergence once ||X^T X - I|| < 1; for the near-orthonormal
// matrices L-BFGS hands to update() that is 4-6 steps.  s^2 = min(||G||_inf, trace G) with G = A^T A is a
// rigorous upper bound of sigma_max^2.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace moihgp {
namespace {

// per row i of G (one wave per row, coalesced): rows[4i..] = { sum_j |G_ij|, G_ii, max_j |G_ij - delta_ij|, sum_j (G_ij - delta_ij)^2 }
__global__ void __launch_bounds__(256) gram_row_stats_kernel(const double* __restrict__ G, size_t L, double* __restrict__ rows) {
    const int lane = threadIdx.x & 63;
    const size_t i = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= L) return;
    double s = 0.0, dev = 0.0, sq = 0.0;
    for (size_t j = lane; j < L; j += 64) {
        const double g = G[i * L + j], e = g - (i == j ? 1.0 : 0.0);
        s += fabs(g);
        dev = fmax(dev, fabs(e));
        sq = fma(e, e, sq);
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); dev = fmax(dev, __shfl_xor(dev, o)); sq += __shfl_xor(sq, o); }
    if (lane == 0) { rows[4 * i] = s; rows[4 * i + 1] = G[i * L + i]; rows[4 * i + 2] = dev; rows[4 * i + 3] = sq; }
}

// out[0] = max_i sum_j |G_ij| ; out[1] = trace(G) ; out[2] = max_ij |G_ij - delta_ij| ; out[5] = ||G - I||_inf = max_i sum_j |G_ij - delta_ij| ;
// out[6] = ||G - I||_F^2     (one workgroup over the row records)
__global__ void __launch_bounds__(256) gram_stats_kernel(const double* __restrict__ rows, size_t L, double* __restrict__ out) {
    __shared__ double r0[256], r1[256], r2[256], r3[256], r4[256];
    const int tid = threadIdx.x;
    double mx = 0.0, tr = 0.0, dev = 0.0, einf = 0.0, fro = 0.0;
    for (size_t i = tid; i < L; i += 256) {
        const double s = rows[4 * i], g = rows[4 * i + 1];
        mx = fmax(mx, s); tr += g; dev = fmax(dev, rows[4 * i + 2]); fro += rows[4 * i + 3];
        einf = fmax(einf, s - fabs(g) + fabs(g - 1.0));               // row sum of |G - I|
    }
    r0[tid] = mx; r1[tid] = tr; r2[tid] = dev; r3[tid] = einf; r4[tid] = fro;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { r0[tid] = fmax(r0[tid], r0[tid + o]); r1[tid] += r1[tid + o]; r2[tid] = fmax(r2[tid], r2[tid + o]); r3[tid] = fmax(r3[tid], r3[tid + o]); r4[tid] += r4[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { out[0] = r0[0]; out[1] = r1[0]; out[2] = r2[0]; out[5] = r3[0]; out[6] = r4[0]; }
}

// y = G v (one wave per row), used by the power iteration for lambda_max(G)
__global__ void __launch_bounds__(256) symv_kernel(const double* __restrict__ G, size_t L, const double* __restrict__ v, double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const size_t i = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= L) return;
    double s = 0.0;
    for (size_t j = lane; j < L; j += 64) s += G[i * L + j] * v[j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[i] = s;
}

// v <- y / ||y||, stats[3] = ||y|| (the Rayleigh-type estimate of lambda_max when ||v|| = 1)      (one workgroup)
__global__ void __launch_bounds__(256) normalize_kernel(const double* __restrict__ y, size_t L, double* __restrict__ v, double* __restrict__ stats) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (size_t i = tid; i < L; i += 256) s += y[i] * y[i];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double nrm = sqrt(red[0]);
    for (size_t i = tid; i < L; i += 256) v[i] = y[i] / nrm;
    if (tid == 0) stats[3] = nrm;
}

__global__ void fill_kernel(double* __restrict__ v, size_t n, double val) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = val;
}

// X *= 1/sqrt(s2), G *= 1/s2 with s2 = stats[4]
__global__ void scale2_kernel(double* __restrict__ X, size_t nX, double* __restrict__ G, size_t nG, const double* __restrict__ stats) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double s2 = stats[4];
    if (i < nX) X[i] *= 1.0 / sqrt(s2);
    if (i < nG) G[i] *= 1.0 / s2;
}

__global__ void scale_kernel(double* __restrict__ X, size_t n, const double* __restrict__ stats) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s2 = fmin(stats[0], stats[1]);
    X[i] *= 1.0 / sqrt(s2);
}

// W = 3/2 I - 1/2 G
__global__ void ns_weight_kernel(const double* __restrict__ G, double* __restrict__ W, size_t L) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L * L) return;
    const size_t i = idx / L, j = idx % L;
    W[idx] = (i == j ? 1.5 : 0.0) - 0.5 * G[idx];
}

// Small matrices (M L <= 4096, L <= 32: the 8 x 4 demo model, a 64 x 16 mixing ...): the same Newton-Schulz iteration as ONE
// workgroup with everything in LDS -- the Gram matrix, the scale s^2 = min(||G||_inf, trace G), the weight 3/2 I - 1/2 G, the
// update, the stopping test -- so that update() of a small model is a single launch instead of a dozen with host round trips.
// status[0] = number of steps, or -1 if the iteration does not reach orthonormality (rank-deficient input).
__global__ void __launch_bounds__(256) polar_small_kernel(double* __restrict__ A, int M, int L, int* __restrict__ status) {
    extern __shared__ double sm[];
    double* X = sm;                    // [M][L]
    double* Xn = X + M * L;            // [M][L]
    double* G = Xn + M * L;            // [L][L]
    double* red = G + L * L;           // [256]
    __shared__ double s_err, s_prev, s_scale;
    __shared__ int s_done;
    const int tid = threadIdx.x, ML = M * L, LL = L * L;
    for (int e = tid; e < ML; e += 256) X[e] = A[e];
    if (tid == 0) { s_prev = 1e300; s_done = -2; }
    __syncthreads();
    auto gram = [&]() {
        for (int e = tid; e < LL; e += 256) {
            const int i = e / L, j = e % L;
            double acc = 0.0;
            for (int m = 0; m < M; m++) acc += X[m * L + i] * X[m * L + j];
            G[e] = acc;
        }
        __syncthreads();
    };
    auto block_max = [&](double v) {
        red[tid] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmax(red[tid], red[tid + o]); __syncthreads(); }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    auto block_add = [&](double v) {
        red[tid] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    gram();
    {   // scale: s^2 = min(max row sum of |G|, trace G) >= sigma_max^2
        double rs = 0.0, tr = 0.0;
        for (int i = tid; i < L; i += 256) {
            double a = 0.0;
            for (int j = 0; j < L; j++) a += fabs(G[i * L + j]);
            rs = fmax(rs, a);
            tr += G[i * L + i];
        }
        const double mx = block_max(rs), trace = block_add(tr);
        const double s2 = fmin(mx, trace);
        if (tid == 0) s_scale = (s2 > 0.0 && s2 == s2) ? 1.0 / sqrt(s2) : nan("");
        __syncthreads();
        const double sc = s_scale;
        for (int e = tid; e < ML; e += 256) X[e] *= sc;
        __syncthreads();
    }
    for (int it = 1; it <= 200; it++) {
        gram();
        double dev = 0.0;
        for (int e = tid; e < LL; e += 256) dev = fmax(dev, fabs(G[e] - ((e / L) == (e % L) ? 1.0 : 0.0)));
        bool bad = false;
        for (int e = tid; e < LL; e += 256) bad = bad || (G[e] != G[e]);
        const double err = block_max(bad ? 1e300 : dev);
        if (tid == 0) {
            if (!(err < 1e299)) s_done = -1;                                             // NaN somewhere
            else if (err < 1e-14 || (err < 1e-11 && err >= 0.5 * s_prev)) s_done = it - 1;   // orthonormal to rounding
            else if (it > 80 && err >= s_prev) s_done = -1;                              // stagnating: rank deficient
            s_prev = err;
        }
        __syncthreads();
        if (s_done != -2) break;
        for (int e = tid; e < ML; e += 256) {                                            // X (3/2 I - 1/2 G)
            const int m = e / L, j = e % L;
            double acc = 0.0;
            for (int k = 0; k < L; k++) acc += X[m * L + k] * ((k == j ? 1.5 : 0.0) - 0.5 * G[k * L + j]);
            Xn[e] = acc;
        }
        __syncthreads();
        for (int e = tid; e < ML; e += 256) X[e] = Xn[e];
        __syncthreads();
    }
    const int res = s_done == -2 ? -1 : s_done;
    for (int e = tid; e < ML; e += 256) A[e] = res < 0 ? nan("") : X[e];     // a rank-deficient input leaves NaN, as update() reports it
    if (tid == 0) { __threadfence_system(); status[0] = res; }
}

}  // namespace

bool polar_small_fits(size_t M, size_t L) { return M * L <= 4096 && L <= 32; }

// A_dev (M x L, device) <- its polar factor, one launch; *status_dev (device int) receives the step count or -1
void launch_polar_small(double* A_dev, size_t M, size_t L, int* status_dev, hipStream_t s) {
    const size_t smem = (2 * M * L + L * L + 256) * sizeof(double);
    hipLaunchKernelGGL(polar_small_kernel, dim3(1), dim3(256), smem, s, A_dev, (int)M, (int)L, status_dev);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

// MOIHGP_POLAR_TRACE=1: one line per Gram matrix on stderr; MOIHGP_POLAR_DUMP=<file> + MOIHGP_POLAR_DUMP_CALL=<n>: the first Gram matrix of
// the n-th call, raw doubles (tools/polar_spectrum.py turns it into singular values).  Development aids, read once.
static int polar_trace_level() { static const int v = [] { const char* e = std::getenv("MOIHGP_POLAR_TRACE"); return e ? std::atoi(e) : 0; }(); return v; }
static void polar_dump_gram(const double* G, size_t L, hipStream_t s) {
    static const char* path = std::getenv("MOIHGP_POLAR_DUMP");
    static const int want = [] { const char* e = std::getenv("MOIHGP_POLAR_DUMP_CALL"); return e ? std::atoi(e) : 0; }();
    static int call = 0;
    if (!path || call++ != want) return;
    std::vector<double> h(L * L);
    MOIHGP_HIP_FATAL(hipMemcpyAsync(h.data(), G, sizeof(double) * L * L, hipMemcpyDeviceToHost, s));
    MOIHGP_HIP_FATAL(hipStreamSynchronize(s));
    if (FILE* f = std::fopen(path, "wb")) { std::fwrite(h.data(), sizeof(double), L * L, f); std::fclose(f); }
}

size_t polar_work_doubles(size_t M, size_t L) { return M * L + 2 * L * L + 8 + 4 * L + polar_deflate_work_doubles(M, L); }

int polar_factor_device(double* X, size_t M, size_t L, double* work, hipStream_t s, int* deflate_warm) {
    double* G = work;                 // L*L
    double* W = G + L * L;            // L*L
    double* Xn = W + L * L;           // M*L
    double* stats = Xn + M * L;       // 8
    double* rows = stats + 8;         // 4*L  (also the two power-iteration vectors)
    double* dwork = rows + 4 * L;     // polar_deflate_work_doubles(M, L)
    double h[8];
    const unsigned nbLL = (unsigned)((L * L + 255) / 256), nbML = (unsigned)((M * L + 255) / 256), nbRow = (unsigned)((L + 3) / 4);
    static const bool deflate_on = [] { const char* e = std::getenv("MOIHGP_POLAR_DEFLATE"); return !(e && e[0] == '0'); }();
    auto gram_and_stats = [&](bool form = true) -> int {
        if (form && launch_gram(X, M, L, G, s)) return -1;
        hipLaunchKernelGGL(gram_row_stats_kernel, dim3(nbRow), dim3(256), 0, s, G, L, rows);
        hipLaunchKernelGGL(gram_stats_kernel, dim3(1), dim3(256), 0, s, rows, L, stats);
        MOIHGP_HIP_FATAL(hipMemcpyAsync(h, stats, sizeof(double) * 7, hipMemcpyDeviceToHost, s));
        MOIHGP_HIP_FATAL(hipStreamSynchronize(s));
        return 0;
    };
    if (gram_and_stats()) return -1;
    double bound = fmin(h[0], h[1]);
    if (!(bound > 0.0) || h[0] != h[0]) return -1;
    polar_dump_gram(G, L, s);
    // ---- a few outlying singular values (the online learner's iterate: polar_deflate.hip) are taken out exactly before anything else:
    // worth an attempt whenever the input is not orthonormal to 1e-3 already (two tall-skinny passes over G decide whether it pays)
    if (!(deflate_on && h[2] > 1e-3) && deflate_warm) *deflate_warm = 0;
    if (deflate_on && h[2] > 1e-3) {
        int pairs = 0;
        if (polar_deflate(X, M, L, G, h[6], dwork, s, &pairs, polar_trace_level(), deflate_warm) != 0) return -1;
        if (pairs > 0) {
            if (gram_and_stats(false)) return -1;          // (the deflation has updated G along with X)
            bound = fmin(h[0], h[1]);
            if (!(bound > 0.0) || h[0] != h[0]) return -1;
        }
    }
    // ---- scale: Newton-Schulz converges for sigma / s in (0, sqrt 3).  s^2 comes from an estimate of lambda_max(G) (power iteration: a
    // Rayleigh value, i.e. a LOWER bound that can be far too low when the start vector misses the dominant direction) held against the
    // rigorous upper bound min(||G||_inf, trace G): s^2 >= bound / 2.9 whatever the estimate says, so sigma^2 / s^2 <= 2.9 < 3 always.
    if (bound > 2.0) {                    // (an input this close to orthonormal is not scaled at all, below: no estimate needed)
        double* v = rows;
        double* y = rows + L;
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, v, L, 1.0 / sqrt((double)L));
        for (int it = 0; it < 16; it++) {
            hipLaunchKernelGGL(symv_kernel, dim3(nbRow), dim3(256), 0, s, (const double*)G, L, (const double*)v, y);
            hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, s, (const double*)y, L, v, stats);
        }
        MOIHGP_HIP_FATAL(hipMemcpyAsync(h, stats, sizeof(double) * 4, hipMemcpyDeviceToHost, s));
        MOIHGP_HIP_FATAL(hipStreamSynchronize(s));
    }
    // sigma_max / s = 1.2 when the estimate is exact (a singular value somewhat ABOVE 1 comes back below it in one step, 1.2 -> 0.94, while
    // the bulk starts 23 % closer to 1: one step fewer on every wide spectrum simulated, never one more)
    double s2 = fmax(h[3] / 1.44, bound / 2.9);
    if (h[3] >= 0.8 && h[3] <= 1.44 && bound <= 2.9) s2 = 1.0;     // already there: scaling would only move the bulk
    if (!(s2 > 0.0) || s2 > bound) s2 = bound;
    bool used_bound = (s2 == bound);
    // An input that is orthonormal already to within the reach of the iteration -- the learner's case: the previous polar factor plus
    // an L-BFGS step of norm <= 0.1 (moihgp_online.h:156) -- needs no scaling at all: the RIGOROUS bound on lambda_max is below 2, so
    // every singular value lies inside (0, sqrt 3) as it stands, and dividing by 1.05 lambda_max would only push them 2.5 % away from 1
    // (error 0.05 instead of ~0.005: one more step, 3.4 ms at 4096^2).
    const bool unscaled = bound <= 2.0;
    if (unscaled) { s2 = 1.0; used_bound = true; }
    h[4] = s2;
    if (!unscaled && s2 != 1.0) {
        MOIHGP_HIP_FATAL(hipMemcpyAsync(stats + 4, &h[4], sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(scale2_kernel, dim3(nbML > nbLL ? nbML : nbLL), dim3(256), 0, s, X, M * L, G, L * L, (const double*)stats);   // G is now X^T X
    }
    double prev = 1e300;
    double* cur = X;
    double* nxt = Xn;
    for (int it = 1; it <= 200; it++) {
        if (it > 1 && launch_gram(cur, M, L, G, s)) return -1;
        hipLaunchKernelGGL(gram_row_stats_kernel, dim3(nbRow), dim3(256), 0, s, G, L, rows);
        hipLaunchKernelGGL(gram_stats_kernel, dim3(1), dim3(256), 0, s, rows, L, stats);
        double hh[6];
        MOIHGP_HIP_FATAL(hipMemcpyAsync(hh, stats, sizeof(double) * 6, hipMemcpyDeviceToHost, s));
        MOIHGP_HIP_FATAL(hipStreamSynchronize(s));
        h[0] = hh[0]; h[1] = hh[1]; h[2] = hh[2];
        const double err = h[2], einf = hh[5];
        if (polar_trace_level()) std::fprintf(stderr, "polar: gram %d  max|G-I| %.3e  ||G-I||inf %.3e  ||G||inf %.4f  tr/L %.6f  s2 %.4f lam_est %.4f\n", it, err, einf, hh[0], hh[1] / (double)L, h[4], h[3]);
        if (err != err) return -1;
        // converged: orthonormal to rounding (the error floor of an fp64 Gram matrix is ~ K eps), or no longer improving
        if (err < 1e-14 || (err < 1e-11 && err >= 0.5 * prev)) {
            if (cur != X) MOIHGP_HIP_FATAL(hipMemcpyAsync(X, cur, sizeof(double) * M * L, hipMemcpyDeviceToDevice, s));
            return it - 1;
        }
        if (it > 3 && err > 4.0 * prev && !used_bound) {
            // the spectral estimate was too small after all (error growing): the current iterate still has the polar factor
            // of A, so rescale it by the rigorous bound of ITS Gram matrix (h[0], h[1] above) and carry on
            used_bound = true;
            h[4] = fmin(h[0], h[1]);
            MOIHGP_HIP_FATAL(hipMemcpyAsync(stats + 4, &h[4], sizeof(double), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(scale2_kernel, dim3(nbML > nbLL ? nbML : nbLL), dim3(256), 0, s, cur, M * L, G, L * L, (const double*)stats);
            MOIHGP_HIP_FATAL(hipStreamSynchronize(s));
            prev = 1e300;
            continue;
        }
        if (it > 80 && err >= prev) break;       // stagnating far from orthonormal: rank deficient
        prev = err;
        hipLaunchKernelGGL(ns_weight_kernel, dim3(nbLL), dim3(256), 0, s, G, W, L);
        if (launch_matmul_nn(cur, M, L, W, nxt, s)) return -1;
        double* t = cur; cur = nxt; nxt = t;
        // E = I - X^T X obeys E' = 3/4 E^2 + 1/4 E^3 under this step, and ||E||_2 <= ||E||_inf for a symmetric E: once that norm
        // is below 1e-7 the step just taken has brought every entry of the next Gram matrix below 1e-14 -- no need to form it
        if (einf < 1e-7) {
            if (cur != X) MOIHGP_HIP_FATAL(hipMemcpyAsync(X, cur, sizeof(double) * M * L, hipMemcpyDeviceToDevice, s));
            return it;
        }
    }
    return -1;
}

}  // namespace moihgp
