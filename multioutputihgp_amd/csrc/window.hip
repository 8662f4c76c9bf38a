// window.hip -- one evaluation of the learners' windowed objective on the device.
//
// Reference loop (moihgp_online.h:61-70, moihgp_regression.h:42-50, online_learning.py:83-89), W ticks:
//     step(x, y_t, dx, xnew, dxnew);  loss += negLogLikelihood(x, y_t, dx, g);  grad += g;  x = xnew;  dx = dxnew
// i.e. per tick one OILMM projection (moihgp.h:181), L sensitivity steps (ihgp.h:37-57) and the full NLL gradient
// (moihgp.h:460-611).  Summed over the window this is
//     Ty  = S^-1/2 U^T Y^T                                  one GEMM            (gemm_mfma.hip)
//     per-latent sweep with sensitivities over Ty            grad_scan_kernel    (grad.hip)  -> nll_l, g_l[P], HA x_t
//     pv_{l,t} = (y_t(l) - HA x_{l,t}) (1 - HA K) / S_l      (raw y(l), sic moihgp.h:510)
//     grad_U   = Y^T Z,  Z_{t,l} = pv/sqrt(S_l) - (U^T y_t)_l / sigma          one GEMM   (closed form of moihgp.h:538-552)
//     r_t      = || y_t - U U^T y_t ||                         one GEMM + row norms
//     loss     = W (1/2 log sum S + 1/2 m_n log sigma) + 1/2 sum_t r_t / sigma [+ sum_l nll_l]          (moihgp.h:503; the bracket only
//                with threading on: the serial branch of the gradient overload drops the per-latent losses, moihgp.h:590 vs :597-607)
//     grad_S_l = W/(2 S_l) - 1/2 S_l^-3/2 sum_t pv (U^T y) - sigma/S_l^2 g_l[noise]                     (moihgp.h:555-561, :604)
//     grad_sigma = sum_t 1/2 (m_n - r_t/sigma)/sigma + sum_l g_l[noise]/S_l                             (moihgp.h:563, :605)
// all fp64.  A tick with missing outputs (NaN in y_t; tmiss[t] = 1): its column of Ty is re-projected by least squares over the observed
// rows (moihgp.h:485-494; tick.hip ls_project_kernel, the stream path's kernel), which is all the per-latent terms see (:565-607).  The
// other terms are, in the reference, dense products with a y_t that holds NaN -- U^T y_t (all NaN), (I - U U^T) y_t, pv with raw y(l)
// -- so for such a tick (U^T y_t)_l is NaN here too and everything downstream of it follows by arithmetic: loss, grad_U, grad_S,
// grad_sigma NaN; per-latent gradient and carried state finite.
#include "common.h"

namespace moihgp {
namespace {

constexpr int P = kNumIgpParam;

// one wave per latent: Z row, spu[l] = sum_t pv * Uty
template <int D>
__global__ void __launch_bounds__(256) window_z_kernel(size_t M, size_t L, size_t W, size_t ldw, const double* __restrict__ cb,
                                                       const double* __restrict__ S, const double* __restrict__ sigma_p,
                                                       const double* __restrict__ Y, const double* __restrict__ Ty,
                                                       const double* __restrict__ hx, double* __restrict__ Z, double* __restrict__ spu,
                                                       const int* __restrict__ tmiss) {
    using Lay = CB<D>;
    const int lane = threadIdx.x & 63;
    const size_t l = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (l >= L) return;
    const double* c = cb + l * Lay::SIZE;
    double hak = 0.0;
    for (int i = 0; i < D; i++) hak += c[Lay::HA + i] * c[Lay::K + i];
    const double Sl = S[l], sq = sqrt(Sl), sigma = *sigma_p, f = (1 - hak) / c[Lay::S];
    double acc = 0.0;
    for (size_t t = lane; t < W; t += 64) {
        const double pv = (Y[t * M + l] - hx[l * ldw + t]) * f;          // moihgp.h:510-511
        const double uty = (tmiss && tmiss[t]) ? __builtin_nan("") : Ty[l * ldw + t] * sq;   // (U^T y_t)_l: a dense product with y_t (moihgp.h:544)
        Z[l * ldw + t] = pv * (1.0 / sq) - uty / sigma;
        acc += pv * uty;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) spu[l] = acc;
}

// the same for a stacked kernel (run-time state dim; HA, K, S from the XC block)
__global__ void __launch_bounds__(256) window_z_x_kernel(size_t M, size_t L, size_t W, size_t ldw, int d, const double* __restrict__ cb,
                                                         const double* __restrict__ S, const double* __restrict__ sigma_p,
                                                         const double* __restrict__ Y, const double* __restrict__ Ty,
                                                         const double* __restrict__ hx, double* __restrict__ Z, double* __restrict__ spu,
                                                         const int* __restrict__ tmiss) {
    const int lane = threadIdx.x & 63;
    const size_t l = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (l >= L) return;
    const double* c = cb + l * (size_t)xc_size(d);
    const double* Kp = c + d * d;
    const double* HAp = Kp + d + d * d;
    double hak = 0.0;
    for (int i = 0; i < d; i++) hak += HAp[i] * Kp[i];
    const double Sl = S[l], sq = sqrt(Sl), sigma = *sigma_p, f = (1 - hak) / HAp[d];
    double acc = 0.0;
    for (size_t t = lane; t < W; t += 64) {
        const double pv = (Y[t * M + l] - hx[l * ldw + t]) * f;          // moihgp.h:510-511
        const double uty = (tmiss && tmiss[t]) ? __builtin_nan("") : Ty[l * ldw + t] * sq;
        Z[l * ldw + t] = pv * (1.0 / sq) - uty / sigma;
        acc += pv * uty;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) spu[l] = acc;
}

// one wave per tick: r_t = || y_t - U U^T y_t ||   (moihgp.h:501, un-squared norm)
__global__ void __launch_bounds__(256) window_resid_kernel(size_t M, size_t W, const double* __restrict__ Y, const double* __restrict__ UU,
                                                           double* __restrict__ rt) {
    const int lane = threadIdx.x & 63;
    const size_t t = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= W) return;
    double s = 0.0;
    for (size_t m = lane; m < M; m += 64) { const double r = Y[t * M + m] - UU[t * M + m]; s += r * r; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) rt[t] = sqrt(s);
}

__device__ double wg_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) window_finalize_kernel(size_t M, size_t L, size_t W, const double* __restrict__ S,
                                                              const double* __restrict__ sigma_p, const double* __restrict__ rt,
                                                              const double* __restrict__ spu, const double* __restrict__ nll,
                                                              const double* __restrict__ gl, double* __restrict__ loss,
                                                              double* __restrict__ grad, int add_latent_loss, int Pn) {
    __shared__ double red[256];
    const int tid = threadIdx.x, nt = blockDim.x;
    const double sigma = *sigma_p;
    double a = 0.0, b = 0.0, c = 0.0, gs = 0.0;
    for (size_t l = tid; l < L; l += nt) { a += S[l]; c += nll[l]; gs += gl[l * Pn + (Pn - 1)] / S[l]; }
    for (size_t t = tid; t < W; t += nt) b += rt[t];
    const double Ssum = wg_sum(a, red), rsum = wg_sum(b, red), lsum = wg_sum(c, red), gsum = wg_sum(gs, red);
    double m_n = (double)M - (double)L;
    if (m_n < 0.0) m_n = 0.0;
    const double Wd = (double)W;
    if (tid == 0) {
        // moihgp.h:503 summed over ticks; the per-latent losses only on the threaded branch (:590), the serial one (:597-607) drops them
        *loss = Wd * (0.5 * log(Ssum) + 0.5 * m_n * log(sigma)) + 0.5 * rsum / sigma + (add_latent_loss ? lsum : 0.0);
        grad[M * L + L] = 0.5 * (Wd * m_n - rsum / sigma) / sigma + gsum;                                     // moihgp.h:563, :605
    }
    for (size_t l = tid; l < L; l += nt) {
        const double Sl = S[l], sq = sqrt(Sl);
        grad[M * L + l] = Wd * 0.5 / Sl - 0.5 * (1.0 / sq / sq / sq) * spu[l] - gl[l * Pn + (Pn - 1)] * sigma / Sl / Sl;   // moihgp.h:555-561, :604
        for (int p = 0; p < Pn; p++) grad[M * L + L + 1 + l * Pn + p] = gl[l * Pn + p];                          // moihgp.h:608-609
    }
}

}  // namespace

int launch_window_objective(const TickArgs& a, const double* cb64, const float* cb32, const WindowBufs& w, int* fallback, double* loss,
                            double* grad, hipStream_t s, int kernel) {
    int rc;
    if ((rc = launch_project_stream(0, w.Y, w.W, a.M, a.L, a.U, nullptr, a.invsqrtS, w.Ty, w.ldw, s))) return rc;
    if (w.tmiss && (rc = launch_project_stream_missing(0, w.Y, w.W, a.M, a.L, a.U, a.invsqrtS, w.Ty, w.ldw, s))) return rc;   // moihgp.h:485-494
    if (a.d > 3) { if ((rc = launch_grad_stream_x(kernel, 0, w.Ty, w.W, w.ldw, a.L, cb64, a.cbd64, w.x, w.dx, w.hx, w.nll, w.gl, s, /*out_mode=*/2))) return rc; }
    else if ((rc = launch_grad_stream(a.d, 0, w.Ty, w.W, w.ldw, a.L, cb64, cb32, w.x, w.dx, w.hx, w.nll, w.gl, fallback, s, /*out_mode=*/2))) return rc;
    dim3 b256(256);
    if (a.d > 3)
        hipLaunchKernelGGL(window_z_x_kernel, dim3((unsigned)((a.L + 3) / 4)), b256, 0, s, a.M, a.L, w.W, w.ldw, a.d, cb64, a.S, a.sigma, w.Y, w.Ty, w.hx, w.Z, w.spu, w.tmiss);
    else if (a.d == 2)
        hipLaunchKernelGGL(window_z_kernel<2>, dim3((unsigned)((a.L + 3) / 4)), b256, 0, s, a.M, a.L, w.W, w.ldw, cb64, a.S, a.sigma, w.Y, w.Ty, w.hx, w.Z, w.spu, w.tmiss);
    else
        hipLaunchKernelGGL(window_z_kernel<3>, dim3((unsigned)((a.L + 3) / 4)), b256, 0, s, a.M, a.L, w.W, w.ldw, cb64, a.S, a.sigma, w.Y, w.Ty, w.hx, w.Z, w.spu, w.tmiss);
    if ((rc = launch_ugrad_gemm(w.Y, w.W, a.M, w.Z, w.ldw, a.L, grad, s))) return rc;                         // grad[0 .. M*L) = U-gradient
    if ((rc = launch_unproject_stream(0, w.Ty, w.W, w.ldw, a.M, a.L, a.U, nullptr, a.sqrtS, w.UU, s))) return rc;           // U (U^T y_t)
    hipLaunchKernelGGL(window_resid_kernel, dim3((unsigned)((w.W + 3) / 4)), b256, 0, s, a.M, w.W, w.Y, w.UU, w.rt);
    hipLaunchKernelGGL(window_finalize_kernel, dim3(1), b256, 0, s, a.M, a.L, w.W, a.S, a.sigma, w.rt, w.spu, w.nll, w.gl, loss, grad, a.lik1_latent_loss, a.P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("window objective launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace moihgp
