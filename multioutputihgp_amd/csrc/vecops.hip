// vecops.hip -- the handful of vector kernels a bound-constrained L-BFGS needs when its vectors live on the device
// (include/moihgp.h "device vectors"): theta, the gradient and the m correction pairs of the learners' optimiser are
// 8 (M L + ..)-byte vectors -- 134 MB each at M = L = 4096 -- and the host form of the loop (moihgp_online.h:40-72 under
// LBFGS++ / include/moihgp_cxx/lbfgsb.hpp) streams ~40 of them through the CPU per iteration.  With moihgp_update_dev /
// moihgp_window_eval_dev the objective never leaves the device; these kernels keep the optimiser's own arithmetic there too.
// All fp64.  Reductions are two-stage and deterministic: a fixed grid of kRedBlocks workgroups accumulates thread-strided partial sums
// (fixed order per thread, a tree per workgroup), a second one-workgroup kernel adds the partials in index order; the scalar goes to
// a page-locked host word.  HBM-bound by construction (one or two streams per element, no reuse).
#include "common.h"
#include <cstdlib>
#include "../../include/moihgp.h"

#include <map>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

namespace moihgp {
namespace {

constexpr int kRedBlocks = 1024, kRedThreads = 256;

__device__ inline double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) for (int w = 0; w < kRedThreads / 64; w++) r += red[w];
    return r;                                            // valid in thread 0
}
__device__ inline double block_max(double v, double* red) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double u = __shfl_xor(v, o); v = (u > v || u != u) ? u : v; }      // (NaN sticks)
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) for (int w = 0; w < kRedThreads / 64; w++) r = (red[w] > r || red[w] != red[w]) ? red[w] : r;
    return r;
}

__global__ void __launch_bounds__(kRedThreads) dot_kernel(size_t n, const double* __restrict__ a, const double* __restrict__ b, const unsigned char* __restrict__ mask,
                                                          double* __restrict__ partial) {
    __shared__ double red[kRedThreads / 64];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += (size_t)kRedBlocks * kRedThreads)
        if (!mask || mask[i]) acc = fma(a[i], b[i], acc);
    const double r = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// xt = clamp(xp + step drt, lb, ub);  partial sums of gradp (xt - xp)   (the projected search point and the Armijo decrease, lbfgsb.hpp)
__global__ void __launch_bounds__(kRedThreads) proj_step_kernel(size_t n, const double* __restrict__ xp, const double* __restrict__ drt, double step,
                                                                const double* __restrict__ lb, const double* __restrict__ ub, const double* __restrict__ gradp,
                                                                double* __restrict__ xt, double* __restrict__ partial) {
    __shared__ double red[kRedThreads / 64];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += (size_t)kRedBlocks * kRedThreads) {
        const double v = fmin(fmax(fma(step, drt[i], xp[i]), lb[i]), ub[i]);
        xt[i] = v;
        acc = fma(gradp[i], v - xp[i], acc);
    }
    const double r = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// max_i |clamp(x_i - g_i, lb_i, ub_i) - x_i|   (projected-gradient norm, LBFGSB.h:64-67)
__global__ void __launch_bounds__(kRedThreads) proj_grad_kernel(size_t n, const double* __restrict__ x, const double* __restrict__ g, const double* __restrict__ lb,
                                                                const double* __restrict__ ub, double* __restrict__ partial) {
    __shared__ double red[kRedThreads / 64];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += (size_t)kRedBlocks * kRedThreads) {
        const double d = fabs(fmin(fmax(x[i] - g[i], lb[i]), ub[i]) - x[i]);
        acc = (d > acc || d != d) ? d : acc;
    }
    const double r = block_max(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ void __launch_bounds__(kRedThreads) finish_kernel(const double* __restrict__ partial, int is_max, double* __restrict__ out_mapped) {
    __shared__ double red[kRedThreads / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < kRedBlocks; i += kRedThreads) {
        const double v = partial[i];
        if (is_max) acc = (v > acc || v != v) ? v : acc; else acc += v;
    }
    const double r = is_max ? block_max(acc, red) : block_sum(acc, red);
    if (threadIdx.x == 0) *out_mapped = r;
}

__global__ void __launch_bounds__(256) axpy_kernel(size_t n, double alpha, const double* __restrict__ x, double* __restrict__ y, const unsigned char* __restrict__ mask) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (!mask || mask[i]) y[i] = fma(alpha, x[i], y[i]);
}
// y = alpha x, entries with mask == 0 set to zero
__global__ void __launch_bounds__(256) scale_kernel(size_t n, double alpha, const double* __restrict__ x, double* __restrict__ y, const unsigned char* __restrict__ mask) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = (!mask || mask[i]) ? alpha * x[i] : 0.0;
}
// out = a - b
__global__ void __launch_bounds__(256) sub_kernel(size_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a[i] - b[i];
}
// x = clamp(x, lb, ub)
__global__ void __launch_bounds__(256) clamp_kernel(size_t n, double* __restrict__ x, const double* __restrict__ lb, const double* __restrict__ ub) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = fmin(fmax(x[i], lb[i]), ub[i]);
}
// free_i = not (at a bound with the gradient pointing out of the box, or a fixed variable)
__global__ void __launch_bounds__(256) active_kernel(size_t n, const double* __restrict__ x, const double* __restrict__ g, const double* __restrict__ lb,
                                                     const double* __restrict__ ub, unsigned char* __restrict__ free_var) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        free_var[i] = !((x[i] <= lb[i] && g[i] > 0.0) || (x[i] >= ub[i] && g[i] < 0.0) || lb[i] == ub[i]) ? 1 : 0;
}

inline unsigned grid_for(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b > 4096 ? 4096 : (b ? b : 1)); }

}  // namespace
}  // namespace moihgp

using namespace moihgp;

struct moihgp_dvec_ctx {
    hipStream_t stream = nullptr;
    double* partial = nullptr;          // [kRedBlocks]
    double* result = nullptr;           // page-locked, device-mapped scalar
};

template <typename F>
static int dv_guard(F&& body) {
    try { return body(); }
    catch (const HipFailure& f) { set_last_error("HIP error %d (%s) at %s:%d: %s", (int)f.err, hipGetErrorString(f.err), f.file, f.line, f.what); return 2; }
    catch (const std::exception& e) { set_last_error("%s", e.what()); return 4; }
}

static double dv_finish(moihgp_dvec_ctx* c, int is_max) {
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kRedThreads), 0, c->stream, c->partial, is_max, c->result);
    MOIHGP_HIP_FATAL(hipGetLastError());
    MOIHGP_HIP_FATAL(hipStreamSynchronize(c->stream));
    return *c->result;
}

// cache of freed device vectors (see moihgp_dvec_alloc below)
static std::mutex g_pool_mutex;
static std::map<size_t, std::vector<void*>> g_pool;            // bytes -> free blocks of exactly that size
static std::unordered_map<void*, size_t> g_live;               // every block this file handed out and has not returned to the driver
static size_t g_pool_bytes = 0;
constexpr size_t kPoolMinBytes = 1 << 20;                      // smaller blocks go straight back to the driver
// cached at most: 8 GB unless told otherwise (MOIHGP_DVEC_CACHE_GB at first use, moihgp_dvec_cache_limit() at any time; 0 = keep nothing).
// The online learner at M = L = 4096 frees and re-allocates 2.7 GB of correction pairs per solve; a library that shares the GPU with a
// framework's own allocator should not sit on much more than its caller's working set.
static size_t g_pool_max = [] { const char* e = std::getenv("MOIHGP_DVEC_CACHE_GB"); return (size_t)((e ? std::atof(e) : 8.0) * (double)((size_t)1 << 30)); }();

extern "C" {

moihgp_dvec_ctx* moihgp_dvec_ctx_new(void) {
    moihgp_dvec_ctx* c = new (std::nothrow) moihgp_dvec_ctx();
    if (!c) return nullptr;
    const int rc = dv_guard([&] {
        MOIHGP_HIP_FATAL(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        MOIHGP_HIP_FATAL(hipMalloc((void**)&c->partial, kRedBlocks * sizeof(double)));
        MOIHGP_HIP_FATAL(hipHostMalloc((void**)&c->result, 64, hipHostMallocMapped));
        return 0;
    });
    if (rc) { moihgp_dvec_ctx_del(c); return nullptr; }
    return c;
}
void* moihgp_dvec_ctx_stream(moihgp_dvec_ctx* c) { return c ? (void*)c->stream : nullptr; }
void moihgp_dvec_ctx_del(moihgp_dvec_ctx* c) {
    if (!c) return;
    if (c->partial) (void)hipFree(c->partial);
    if (c->result) (void)hipHostFree(c->result);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
// Device vectors come from a small cache of freed blocks keyed by size: an optimiser over 10^7 parameters frees and allocates its 2 m
// correction pairs at every solve (the learner hands last solve's matrix to the next objective, moihgp_online.h:182), and hipMalloc /
// hipFree of 134 MB blocks cost 1-2 ms each -- 80 ms of a 220 ms learner tick at M = L = 4096.  A cached free keeps hipFree's meaning for
// the caller (everything queued on the device has finished when it returns); moihgp_dvec_trim() hands the cache back to the driver.
static void* dv_alloc_bytes(size_t bytes, const char* what) {
    if (bytes == 0) bytes = 1;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        auto it = g_pool.find(bytes);
        if (it != g_pool.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            g_pool_bytes -= bytes;
            return p;
        }
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        moihgp_dvec_trim();                                          // the cache may hold what is missing
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); set_last_error("%s: out of device memory (%zu bytes)", what, bytes); return nullptr; }
    }
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        g_live[p] = bytes;
    }
    return p;
}
double* moihgp_dvec_alloc(size_t n) { return static_cast<double*>(dv_alloc_bytes(n * sizeof(double), "dvec_alloc")); }
unsigned char* moihgp_dvec_alloc_mask(size_t n) { return static_cast<unsigned char*>(dv_alloc_bytes(n, "dvec_alloc_mask")); }
void moihgp_dvec_free(void* p) {
    if (!p) return;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        auto it = g_live.find(p);
        if (it != g_live.end()) bytes = it->second;
    }
    if (bytes != 0 && bytes >= kPoolMinBytes) {
        (void)hipDeviceSynchronize();                                // (what hipFree would have waited for)
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (g_pool_bytes + bytes <= g_pool_max) { g_pool[bytes].push_back(p); g_pool_bytes += bytes; return; }
    }
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        g_live.erase(p);
    }
    (void)hipFree(p);
}
void moihgp_dvec_trim(void) {
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (auto& kv : g_pool) for (void* p : kv.second) { blocks.push_back(p); g_live.erase(p); }
        g_pool.clear();
        g_pool_bytes = 0;
    }
    for (void* p : blocks) (void)hipFree(p);
}
void moihgp_dvec_cache_limit(size_t bytes) {
    { std::lock_guard<std::mutex> lock(g_pool_mutex); g_pool_max = bytes; }
    bool over;
    { std::lock_guard<std::mutex> lock(g_pool_mutex); over = g_pool_bytes > bytes; }
    if (over) moihgp_dvec_trim();
}
int moihgp_dvec_upload(moihgp_dvec_ctx* c, double* dst_dev, const double* src_host, size_t n) {
    return dv_guard([&] { MOIHGP_HIP_FATAL(hipMemcpyAsync(dst_dev, src_host, n * sizeof(double), hipMemcpyHostToDevice, c->stream)); MOIHGP_HIP_FATAL(hipStreamSynchronize(c->stream)); return 0; });
}
int moihgp_dvec_download(moihgp_dvec_ctx* c, double* dst_host, const double* src_dev, size_t n) {
    return dv_guard([&] { MOIHGP_HIP_FATAL(hipMemcpyAsync(dst_host, src_dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream)); MOIHGP_HIP_FATAL(hipStreamSynchronize(c->stream)); return 0; });
}
int moihgp_dvec_copy(moihgp_dvec_ctx* c, double* dst_dev, const double* src_dev, size_t n) {
    return dv_guard([&] { MOIHGP_HIP_FATAL(hipMemcpyAsync(dst_dev, src_dev, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream)); return 0; });
}
int moihgp_dvec_sync(moihgp_dvec_ctx* c) { return dv_guard([&] { MOIHGP_HIP_FATAL(hipStreamSynchronize(c->stream)); return 0; }); }

int moihgp_dvec_dot(moihgp_dvec_ctx* c, size_t n, const double* a, const double* b, const unsigned char* mask, double* result) {
    return dv_guard([&] {
        hipLaunchKernelGGL(dot_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, c->stream, n, a, b, mask, c->partial);
        *result = dv_finish(c, 0);
        return 0;
    });
}
int moihgp_dvec_proj_step(moihgp_dvec_ctx* c, size_t n, const double* xp, const double* drt, double step, const double* lb, const double* ub, const double* gradp,
                          double* xt, double* dec) {
    return dv_guard([&] {
        hipLaunchKernelGGL(proj_step_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, c->stream, n, xp, drt, step, lb, ub, gradp, xt, c->partial);
        *dec = dv_finish(c, 0);
        return 0;
    });
}
int moihgp_dvec_proj_grad_norm(moihgp_dvec_ctx* c, size_t n, const double* x, const double* g, const double* lb, const double* ub, double* result) {
    return dv_guard([&] {
        hipLaunchKernelGGL(proj_grad_kernel, dim3(kRedBlocks), dim3(kRedThreads), 0, c->stream, n, x, g, lb, ub, c->partial);
        *result = dv_finish(c, 1);
        return 0;
    });
}
int moihgp_dvec_axpy(moihgp_dvec_ctx* c, size_t n, double alpha, const double* x, double* y, const unsigned char* mask) {
    return dv_guard([&] { hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, n, alpha, x, y, mask); MOIHGP_HIP_FATAL(hipGetLastError()); return 0; });
}
int moihgp_dvec_scale(moihgp_dvec_ctx* c, size_t n, double alpha, const double* x, double* y, const unsigned char* mask) {
    return dv_guard([&] { hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, n, alpha, x, y, mask); MOIHGP_HIP_FATAL(hipGetLastError()); return 0; });
}
int moihgp_dvec_sub(moihgp_dvec_ctx* c, size_t n, const double* a, const double* b, double* out) {
    return dv_guard([&] { hipLaunchKernelGGL(sub_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, n, a, b, out); MOIHGP_HIP_FATAL(hipGetLastError()); return 0; });
}
int moihgp_dvec_clamp(moihgp_dvec_ctx* c, size_t n, double* x, const double* lb, const double* ub) {
    return dv_guard([&] { hipLaunchKernelGGL(clamp_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, n, x, lb, ub); MOIHGP_HIP_FATAL(hipGetLastError()); return 0; });
}
int moihgp_dvec_active_set(moihgp_dvec_ctx* c, size_t n, const double* x, const double* g, const double* lb, const double* ub, unsigned char* free_var) {
    return dv_guard([&] { hipLaunchKernelGGL(active_kernel, dim3(grid_for(n)), dim3(256), 0, c->stream, n, x, g, lb, ub, free_var); MOIHGP_HIP_FATAL(hipGetLastError()); return 0; });
}

}  // extern "C"
