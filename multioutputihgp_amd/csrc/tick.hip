// tick.hip -- single-tick kernels behind the reference's per-tick ABI (gp32_step1..4, gp32_lik1/2).
// All fp64, device pointers.  These calls are launch-latency bound by construction (the reference
// ABI hands over one observation vector per FFI crossing); the throughput path is recursion.hip.
//
// Reference: include/moihgp/moihgp.h:148-428 (step x4: project -> L x IHGP::step -> unproject),
// :460-688 (negLogLikelihood x2), include/moihgp/ihgp.h:37-100, :204-222.
#include "common.h"

namespace moihgp {
namespace {

constexpr int P = kNumIgpParam;

// Ty = S^-1/2 U^T y (moihgp.h:181); Uty = U^T y kept for the NLL terms.
__global__ void project_tick_kernel(size_t M, size_t L, const double* __restrict__ U, const double* __restrict__ S,
                                    const double* __restrict__ y, double* __restrict__ Ty, double* __restrict__ Uty) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double s = 0.0;
    for (size_t m = 0; m < M; m++) s += U[m * L + l] * y[m];
    if (Uty) Uty[l] = s;
    if (Ty) Ty[l] = (1.0 / sqrt(S[l])) * s;
}

// Same projection for larger M: the rows are cut into chunks handled by different workgroups (4 waves each take a quarter of
// the chunk; lanes run over 64 consecutive latents so that every load is a coalesced 512-byte row segment); a second tiny
// kernel adds the chunk partials in a fixed order (deterministic, no atomics).
__global__ void __launch_bounds__(256) project_partial_kernel(size_t M, size_t L, size_t rows_per_chunk, const double* __restrict__ U,
                                                              const double* __restrict__ y, double* __restrict__ part) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t l = (size_t)blockIdx.x * 64 + lane, c = blockIdx.y;
    const size_t m0 = c * rows_per_chunk, m1 = (m0 + rows_per_chunk < M) ? m0 + rows_per_chunk : M;
    double s = 0.0;
    if (l < L)
        for (size_t m = m0 + w; m < m1; m += 4) s += U[m * L + l] * y[m];
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && l < L) part[c * L + l] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

__global__ void project_finish_kernel(size_t L, size_t nchunk, const double* __restrict__ part, const double* __restrict__ S,
                                      double* __restrict__ Ty, double* __restrict__ Uty) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double s = 0.0;
    for (size_t c = 0; c < nchunk; c++) s += part[c * L + l];
    if (Uty) Uty[l] = s;
    if (Ty) Ty[l] = (1.0 / sqrt(S[l])) * s;
}

// Normal equations over observed rows (moihgp.h:167-177): N = U0^T U0, r = U0^T y0.
__global__ void normal_eq_kernel(size_t M, size_t L, const double* __restrict__ U, const double* __restrict__ y,
                                 double* __restrict__ N, double* __restrict__ r) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L * L) return;
    size_t a = idx / L, b = idx % L;
    double s = 0.0, rr = 0.0;
    for (size_t m = 0; m < M; m++) {
        double ym = y[m];
        if (ym != ym) continue;
        double ua = U[m * L + a];
        s += ua * U[m * L + b];
        if (b == 0) rr += ua * ym;
    }
    N[idx] = s;
    if (b == 0) r[a] = rr;
}

// Solve N a = r for SPD N in place (stand-in for Eigen ldlt().solve, moihgp.h:177), one workgroup.
// Then Ty = S^-1/2 a.  Edge path (only ticks with missing outputs); O(L^3) on one CU.
__global__ void __launch_bounds__(256) spd_solve_kernel(size_t L, double* __restrict__ N, double* __restrict__ r,
                                                        const double* __restrict__ S, double* __restrict__ Ty) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (size_t k = 0; k < L; k++) {
        const double pivot = N[k * L + k];
        for (size_t i = k + 1 + tid; i < L; i += nt) {
            double f = N[i * L + k] / pivot;
            for (size_t j = k; j < L; j++) N[i * L + j] -= f * N[k * L + j];
            r[i] -= f * r[k];
        }
        __syncthreads();
    }
    __shared__ double red[256];
    for (size_t ii = L; ii-- > 0;) {
        double s = 0.0;
        for (size_t k = ii + 1 + tid; k < L; k += nt) s += N[ii * L + k] * r[k];
        red[tid] = s;
        __syncthreads();
        for (int o = nt / 2; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) r[ii] = (r[ii] - red[0]) / N[ii * L + ii];
        __syncthreads();
    }
    for (size_t l = tid; l < L; l += nt) Ty[l] = (1.0 / sqrt(S[l])) * r[l];
}

// L x IHGP::step (ihgp.h:37-100), one lane per latent.
template <int D>
__global__ void step_tick_kernel(size_t L, const double* __restrict__ cb, const double* __restrict__ x,
                                 const double* __restrict__ Ty, const double* __restrict__ dx, double* __restrict__ xnew,
                                 double* __restrict__ Tyhat, double* __restrict__ dxnew) {
    using Lay = CB<D>;
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const double* c = cb + l * Lay::SIZE;
    double xs[D], xn[D];
    for (int i = 0; i < D; i++) xs[i] = x[l * D + i];
    const bool has_y = (Ty != nullptr);
    const double y = has_y ? Ty[l] : 0.0;
    const bool miss = !has_y || (y != y);                                // ihgp.h:39 / :96
    const double* Mx = c + (miss ? Lay::A : Lay::AKHA);
    for (int i = 0; i < D; i++) {
        double s = 0.0;
        for (int k = 0; k < D; k++) s += Mx[i * D + k] * xs[k];
        xn[i] = miss ? s : s + c[Lay::K + i] * y;                        // ihgp.h:41 / :50
    }
    for (int i = 0; i < D; i++) xnew[l * D + i] = xn[i];
    if (Tyhat) Tyhat[l] = xn[0];                                         // ihgp.h:42 / :51
    if (dx && dxnew) {
        for (int p = 0; p < P; p++) {
            const double* dM = c + (miss ? Lay::DA : Lay::DAKHA) + p * D * D;
            for (int i = 0; i < D; i++) {
                double a = 0.0, b = 0.0;
                for (int k = 0; k < D; k++) { a += dM[i * D + k] * xs[k]; b += Mx[i * D + k] * dx[(l * P + p) * D + k]; }
                double v = a + b;
                if (!miss) v += c[Lay::DK + p * D + i] * y;              // ihgp.h:45 / :54
                dxnew[(l * P + p) * D + i] = v;
            }
        }
    }
}

// yhat = U S^1/2 Tyhat (moihgp.h:222-225): one wave per output row.
__global__ void __launch_bounds__(256) unproject_tick_kernel(size_t M, size_t L, const double* __restrict__ U,
                                                             const double* __restrict__ S, const double* __restrict__ Tyhat,
                                                             double* __restrict__ yhat) {
    const int lane = threadIdx.x & 63;
    size_t m = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    double s = 0.0;
    for (size_t l = lane; l < L; l += 64) s += U[m * L + l] * (sqrt(S[l]) * Tyhat[l]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) yhat[m] = s;
}

// Per-latent NLL terms: loss_l (ihgp.h:204-209), grad_l (ihgp.h:212-222), pv_l (moihgp.h:505-512).
template <int D>
__device__ inline void igp_nll_one(const double* __restrict__ c, const double* __restrict__ xl, double yraw, double y,
                                   const double* __restrict__ dxl /* [P][D] or NULL */, double* lossv, double* pv, double* igrad /* [P] */) {
    using Lay = CB<D>;
    const double S = c[Lay::S];
    double hx = 0.0, hak = 0.0;
    for (int i = 0; i < D; i++) { hx += c[Lay::HA + i] * xl[i]; hak += c[Lay::HA + i] * c[Lay::K + i]; }
    const double v = y - hx;
    *lossv = 0.5 * (v * v / S + log(S));                                 // ihgp.h:207
    if (dxl) {
        *pv = (yraw - hx) * (1 - hak) / S;                               // moihgp.h:510-511 (raw y(idx), sic)
        for (int p = 0; p < P; p++) {
            double a = 0.0, b = 0.0;
            for (int i = 0; i < D; i++) { a += c[Lay::HDA + p * D + i] * xl[i]; b += c[Lay::HA + i] * dxl[p * D + i]; }
            double dv = -a - b;                                          // ihgp.h:218
            igrad[p] = (v * dv - 0.5 * (v * v / S - 1) * c[Lay::DS + p]) / S;   // ihgp.h:219
        }
    }
}

template <int D>
__global__ void igp_nll_kernel(size_t L, const double* __restrict__ cb, const double* __restrict__ x,
                               const double* __restrict__ yraw, const double* __restrict__ Ty, const double* __restrict__ dx,
                               double* __restrict__ lossv, double* __restrict__ pv, double* __restrict__ igrad) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    igp_nll_one<D>(cb + l * CB<D>::SIZE, x + l * D, yraw[l], Ty[l], dx ? dx + l * P * D : nullptr, lossv + l, pv + l, igrad + l * P);
}

// ---- stacked kernels (MOIHGP_STACK: d = 4 .. 12, P = 2J + 1 = 5 .. 9) behind the per-tick ABI ---------------------------------------------
// Same one-tick arithmetic as the kernels above (ihgp.h:37-100, :204-222) with the state dimension and the parameter count as run-time
// values and the matrices read from the stacked layout: XC block (AKHA, K, A, HA, S) + XD sensitivity block (dAKHA, dK, dA, HdA, dS).
// These calls are launch-latency bound (one FFI crossing per tick); nothing here is worth unrolling.
constexpr int kGenD = kMaxStackDim, kGenP = 9;
struct XBlk {
    const double *AKHA, *K, *A, *HA, *DAKHA, *DK, *DA, *HDA, *DS;
    double S;
};
__device__ inline XBlk xblk(const double* cb64, const double* cbd64, size_t l, int d, int P) {
    const double* c = cb64 + l * (size_t)xc_size(d);
    XBlk b;
    b.AKHA = c; b.K = c + d * d; b.A = b.K + d; b.HA = b.A + d * d; b.S = b.HA[d];                 // XC<D>: AKHA, K, A, HA, S
    if (cbd64) {
        const double* x = cbd64 + l * (size_t)xd_size(d, P);
        b.DAKHA = x; b.DK = x + P * d * d; b.DA = b.DK + P * d; b.HDA = b.DA + P * d * d; b.DS = b.HDA + P * d;   // XD<D, P>
    } else { b.DAKHA = b.DK = b.DA = b.HDA = b.DS = nullptr; }
    return b;
}

__global__ void step_tick_x_kernel(size_t L, int d, int P, const double* __restrict__ cb64, const double* __restrict__ cbd64,
                                   const double* __restrict__ x, const double* __restrict__ Ty, const double* __restrict__ dx,
                                   double* __restrict__ xnew, double* __restrict__ Tyhat, double* __restrict__ dxnew) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const XBlk b = xblk(cb64, cbd64, l, d, P);
    double xs[kGenD], xn[kGenD];
    for (int i = 0; i < d; i++) xs[i] = x[l * d + i];
    const bool has_y = (Ty != nullptr);
    const double y = has_y ? Ty[l] : 0.0;
    const bool miss = !has_y || (y != y);                                // ihgp.h:39 / :96
    const double* Mx = miss ? b.A : b.AKHA;
    for (int i = 0; i < d; i++) {
        double sacc = 0.0;
        for (int k = 0; k < d; k++) sacc += Mx[i * d + k] * xs[k];
        xn[i] = miss ? sacc : sacc + b.K[i] * y;                         // ihgp.h:41 / :50
    }
    for (int i = 0; i < d; i++) xnew[l * d + i] = xn[i];
    if (Tyhat) Tyhat[l] = xn[0];                                         // ihgp.h:42 / :51
    if (dx && dxnew) {
        for (int p = 0; p < P; p++) {
            const double* dM = (miss ? b.DA : b.DAKHA) + p * d * d;
            for (int i = 0; i < d; i++) {
                double a = 0.0, c2 = 0.0;
                for (int k = 0; k < d; k++) { a += dM[i * d + k] * xs[k]; c2 += Mx[i * d + k] * dx[(l * P + p) * d + k]; }
                double v = a + c2;
                if (!miss) v += b.DK[p * d + i] * y;                     // ihgp.h:45 / :54
                dxnew[(l * P + p) * d + i] = v;
            }
        }
    }
}

__global__ void igp_nll_x_kernel(size_t L, int d, int P, const double* __restrict__ cb64, const double* __restrict__ cbd64,
                                 const double* __restrict__ x, const double* __restrict__ yraw, const double* __restrict__ Ty,
                                 const double* __restrict__ dx, double* __restrict__ lossv, double* __restrict__ pv, double* __restrict__ igrad) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const XBlk b = xblk(cb64, dx ? cbd64 : nullptr, l, d, P);
    const double* xl = x + l * d;
    double hx = 0.0, hak = 0.0;
    for (int i = 0; i < d; i++) { hx += b.HA[i] * xl[i]; hak += b.HA[i] * b.K[i]; }
    const double v = Ty[l] - hx;
    lossv[l] = 0.5 * (v * v / b.S + log(b.S));                           // ihgp.h:207
    if (dx) {
        pv[l] = (yraw[l] - hx) * (1 - hak) / b.S;                        // moihgp.h:510-511 (raw y(idx), sic)
        for (int p = 0; p < P; p++) {
            double a = 0.0, c2 = 0.0;
            for (int i = 0; i < d; i++) { a += b.HDA[p * d + i] * xl[i]; c2 += b.HA[i] * dx[(l * P + p) * d + i]; }
            const double dv = -a - c2;                                   // ihgp.h:218
            igrad[l * P + p] = (v * dv - 0.5 * (v * v / b.S - 1) * b.DS[p]) / b.S;   // ihgp.h:219
        }
    } else pv[l] = 0.0;
}

// resid2[m] = (y - U U^T y)_m^2  (moihgp.h:501 / :651): one wave per output row.
__global__ void __launch_bounds__(256) resid_kernel(size_t M, size_t L, const double* __restrict__ U,
                                                    const double* __restrict__ y, const double* __restrict__ Uty,
                                                    double* __restrict__ resid2) {
    const int lane = threadIdx.x & 63;
    size_t m = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    double s = 0.0;
    for (size_t l = lane; l < L; l += 64) s += U[m * L + l] * Uty[l];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) { double r = y[m] - s; resid2[m] = r * r; }
}

__device__ double block_sum(double v, double* red) {
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

// loss and the S / sigma / per-latent gradient entries (moihgp.h:503, :553-563, :598-609): one workgroup.
__device__ inline void nll_finalize_body(size_t M, size_t L, const double* __restrict__ S, double sigma, const double* Uty,
                                         const double* lossv, const double* pv, const double* igrad, const double* resid2,
                                         double* __restrict__ loss, double* __restrict__ grad, double* red, bool add_latent_loss, int Pn = kNumIgpParam) {
    const int tid = threadIdx.x, nt = blockDim.x;
    double a = 0.0, b = 0.0, c = 0.0;
    for (size_t l = tid; l < L; l += nt) { a += S[l]; c += lossv[l]; }
    for (size_t m = tid; m < M; m += nt) b += resid2[m];
    const double Ssum = block_sum(a, red);
    const double nrm = sqrt(block_sum(b, red));
    const double lsum = block_sum(c, red);
    double m_n = (double)M - (double)L;
    if (m_n < 0.0) m_n = 0.0;                                            // moihgp.h:502
    // moihgp.h:503 (sic) + the per-latent terms: always in the overload without gradient (:684); in the gradient overload only on
    // its threaded branch (:590) -- the serial branch (:597-607) computes them and drops them
    if (tid == 0) *loss = 0.5 * log(Ssum) + 0.5 * m_n * log(sigma) + 0.5 * nrm / sigma + (add_latent_loss ? lsum : 0.0);
    if (!grad) return;
    const size_t sizeU = M * L;
    double gs = 0.0;
    for (size_t l = tid; l < L; l += nt) {
        const double Sl = S[l], sq = sqrt(Sl);
        const double dn = igrad[l * Pn + (Pn - 1)];
        double g = 0.5 / Sl + pv[l] * (-0.5 * (1.0 / sq / sq / sq) * Uty[l]);   // moihgp.h:555-561
        g -= dn * sigma / Sl / Sl;                                       // moihgp.h:604
        grad[sizeU + l] = g;
        gs += dn / Sl;                                                   // moihgp.h:605
        for (int p = 0; p < Pn; p++) grad[sizeU + L + 1 + l * Pn + p] = igrad[l * Pn + p];   // moihgp.h:608-609
    }
    const double gsum = block_sum(gs, red);
    if (tid == 0) grad[sizeU + L] = 0.5 * (m_n - nrm / sigma) / sigma + gsum;   // moihgp.h:563
}

__global__ void __launch_bounds__(256) nll_finalize_kernel(size_t M, size_t L, const double* __restrict__ S,
                                                           const double* __restrict__ sigma_p, const double* __restrict__ Uty,
                                                           const double* __restrict__ lossv, const double* __restrict__ pv,
                                                           const double* __restrict__ igrad, const double* __restrict__ resid2,
                                                           double* __restrict__ loss, double* __restrict__ grad, int add_latent_loss, int Pn) {
    __shared__ double red[256];
    nll_finalize_body(M, L, S, *sigma_p, Uty, lossv, pv, igrad, resid2, loss, grad, red, add_latent_loss != 0, Pn);
}

// U-gradient (moihgp.h:538-552).  U is a polar factor (moihgp.h:438-446) so its singular values are
// 1 and `dU` (moihgp.h:545) reduces to the one-hot basis matrix dA[idx1]; the M*L-iteration loop of
// dense products then collapses to  grad_U[r][c] = y_r (pv_c / sqrt(S_c) - (U^T y)_c / sigma).
__global__ void ugrad_kernel(size_t M, size_t L, const double* __restrict__ S, const double* __restrict__ sigma_p,
                             const double* __restrict__ y, const double* __restrict__ Uty, const double* __restrict__ pv,
                             double* __restrict__ grad) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * L) return;
    size_t r = idx / L, c = idx % L;
    grad[idx] = y[r] * (pv[c] * (1.0 / sqrt(S[c])) - Uty[c] / *sigma_p);
}


// ---- whole streams: ticks with missing outputs (moihgp.h:167-178) ----------------------------------------------------------------------
// The stream projection (gemm_mfma.hip) computes S^-1/2 U^T y_t for every tick; a tick whose observation vector holds NaN comes out
// as a NaN column.  This kernel, queued behind it, re-projects exactly those ticks the way the reference does, by least squares over
// the observed rows:  a = (U0^T U0)^-1 U0^T y_obs,  Ty_t = S^-1/2 a.  U is a polar factor (moihgp.h:438-446): U^T U = I, so with
// U_miss the k rows of the missing outputs  U0^T U0 = I - U_miss^T U_miss  and (Woodbury)
//     a = r + U_miss^T (I_k - U_miss U_miss^T)^-1 U_miss r,      r = U0^T y_obs = U^T (y with its NaNs set to 0):
// a k x k solve (k = number of missing outputs of the tick) instead of the reference's L x L LDLT, the same vector to rounding.
// One workgroup per tick; a tick without NaN costs one pass over its M observations and leaves at once.  fp64 inside whatever the
// stream type.  k > kLsMaxMissing or fewer observed outputs than latents: the column stays NaN (the recursion then treats the tick as
// missing; the reference would factor a singular matrix there).

template <typename T>
__global__ void __launch_bounds__(256) ls_project_kernel(const T* __restrict__ Y, size_t Tn, size_t M, size_t L, const double* __restrict__ U,
                                                         const double* __restrict__ invsqrtS, T* __restrict__ Ty, size_t ld) {
    extern __shared__ double lsm[];
    double* r = lsm;                                   // [L]
    double* G = r + L;                                 // [k][k+1] augmented system (I - U_miss U_miss^T | U_miss r)
    __shared__ int miss[kLsMaxMissing];
    __shared__ int kcount;
    const size_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* y = Y + t * M;
    if (tid == 0) kcount = 0;
    __syncthreads();
    for (size_t m = tid; m < M; m += 256) {
        const T v = y[m];
        if (v != v) { const int slot = atomicAdd(&kcount, 1); if (slot < kLsMaxMissing) miss[slot] = (int)m; }
    }
    __syncthreads();
    const int k = kcount;
    if (k == 0) return;                                // fully observed tick: the GEMM's column stands
    if (k > kLsMaxMissing || M - (size_t)k < L) return;   // not solvable here: the NaN column stands (tick treated as missing)
    if (tid == 0) {                                    // deterministic order of the missing rows (the atomics arrive in any order)
        for (int i = 1; i < k; i++) { const int v = miss[i]; int j = i - 1; while (j >= 0 && miss[j] > v) { miss[j + 1] = miss[j]; j--; } miss[j + 1] = v; }
    }
    // r = U^T y0: lanes over latents (coalesced rows of U), NaN observations skipped (moihgp.h:171-176)
    for (size_t l = tid; l < L; l += 256) {
        double sacc = 0.0;
        for (size_t m = 0; m < M; m++) { const double v = (double)y[m]; if (v == v) sacc = fma(U[m * L + l], v, sacc); }
        r[l] = sacc;
    }
    __syncthreads();
    // augmented k x (k+1) system: entry (i, j) = delta_ij - u_i . u_j, entry (i, k) = u_i . r; one wave per entry, butterfly sum
    const int kk = k + 1;
    for (int e = wave; e < k * kk; e += 4) {
        const int i = e / kk, j = e % kk;
        const double* ui = U + (size_t)miss[i] * L;
        double sacc = 0.0;
        if (j < k) { const double* uj = U + (size_t)miss[j] * L; for (size_t l = lane; l < L; l += 64) sacc = fma(ui[l], uj[l], sacc); }
        else for (size_t l = lane; l < L; l += 64) sacc = fma(ui[l], r[l], sacc);
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
        if (lane == 0) G[i * kk + j] = (j < k) ? ((i == j ? 1.0 : 0.0) - sacc) : sacc;
    }
    __syncthreads();
    if (wave == 0) {                                   // (I - G) c = w: symmetric positive definite, elimination without pivoting, lane = row
        for (int p = 0; p < k; p++) {
            const double piv = G[p * kk + p];
            if (lane > p && lane < k) {
                const double f = G[lane * kk + p] / piv;
                for (int j = p; j < kk; j++) G[lane * kk + j] -= f * G[p * kk + j];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
            for (int i = k - 1; i >= 0; i--) {
                double sacc = G[i * kk + k];
                for (int j = i + 1; j < k; j++) sacc -= G[i * kk + j] * G[j * kk + k];
                G[i * kk + k] = sacc / G[i * kk + i];
            }
        }
    }
    __syncthreads();
    // a = r + U_miss^T c;  Ty[l][t] = S_l^-1/2 a_l  (moihgp.h:177)
    for (size_t l = tid; l < L; l += 256) {
        double a = r[l];
        for (int i = 0; i < k; i++) a = fma(U[(size_t)miss[i] * L + l], G[i * kk + k], a);
        Ty[l * ld + t] = (T)(invsqrtS[l] * a);
    }
}

// ---- the same projection with the LATENTS SPLIT OVER RANKS (sharded.py): a pair of kernels around one all-reduce -------------------------
// (U0^T U0)^-1 couples all latents, but with U^T U = I the Woodbury form needs, from the other shards, only sums over the latent
// columns:  G = U_miss U_miss^T (k x k) and b = U_miss r (k), r = U^T (y, NaN -> 0).  Kernel 1 (one workgroup per affected tick) forms this
// rank's part of [b | G] from ITS columns of U and its projected column Ty[:, t] (r_l = sqrt(S_l) Ty[l][t]) into a packed record of
// kmax + kmax^2 doubles (zero padded); the caller all-reduces the records of all affected ticks at once; kernel 2 solves
// (I - G) w = b (every rank the same small system) and corrects this rank's rows:  Ty[l][t] += S_l^-1/2 sum_i U[m_i][l] w_i.
// The missing outputs of a tick are found from the tick's observation vector, in ascending order (the same order on every rank).
template <typename T>
__global__ void __launch_bounds__(256) ls_shard_gram_kernel(const T* __restrict__ Y, size_t M, size_t Lr, const int* __restrict__ ticks, int kmax,
                                                            const double* __restrict__ U /* [M][Lr] this rank's columns */, const double* __restrict__ sqrtS,
                                                            const T* __restrict__ Ty, size_t ld, double* __restrict__ packed /* [n][kmax + kmax^2] */) {
    __shared__ int miss[kLsMaxMissing];
    __shared__ int kcount;
    const size_t t = (size_t)ticks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* y = Y + t * M;
    double* rec = packed + (size_t)blockIdx.x * (size_t)(kmax + kmax * kmax);
    if (tid == 0) kcount = 0;
    for (int e = tid; e < kmax + kmax * kmax; e += 256) rec[e] = 0.0;
    __syncthreads();
    for (size_t m = tid; m < M; m += 256) {
        const T v = y[m];
        if (v != v) { const int slot = atomicAdd(&kcount, 1); if (slot < kLsMaxMissing) miss[slot] = (int)m; }
    }
    __syncthreads();
    const int k = kcount < kmax ? kcount : kmax;
    if (tid == 0) { for (int i = 1; i < k; i++) { const int v = miss[i]; int j = i - 1; while (j >= 0 && miss[j] > v) { miss[j + 1] = miss[j]; j--; } miss[j + 1] = v; } }
    __syncthreads();
    // entry (i, j < k) = u_i . u_j, entry (i, k) -> b_i = u_i . r over this rank's columns: one wave per entry, butterfly sum
    for (int e = wave; e < k * (k + 1); e += 4) {
        const int i = e / (k + 1), j = e % (k + 1);
        const double* ui = U + (size_t)miss[i] * Lr;
        double sacc = 0.0;
        if (j < k) { const double* uj = U + (size_t)miss[j] * Lr; for (size_t l = lane; l < Lr; l += 64) sacc = fma(ui[l], uj[l], sacc); }
        else for (size_t l = lane; l < Lr; l += 64) sacc = fma(ui[l], sqrtS[l] * (double)Ty[l * ld + t], sacc);
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
        if (lane == 0) { if (j < k) rec[kmax + i * kmax + j] = sacc; else rec[i] = sacc; }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) ls_shard_apply_kernel(const T* __restrict__ Y, size_t M, size_t Lr, const int* __restrict__ ticks, int kmax,
                                                             const double* __restrict__ U, const double* __restrict__ invsqrtS,
                                                             const double* __restrict__ packed /* reduced over the ranks */, T* __restrict__ Ty, size_t ld) {
    __shared__ int miss[kLsMaxMissing];
    __shared__ int kcount;
    __shared__ double Gs[kLsMaxMissing * (kLsMaxMissing + 1)];
    const size_t t = (size_t)ticks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* y = Y + t * M;
    const double* rec = packed + (size_t)blockIdx.x * (size_t)(kmax + kmax * kmax);
    if (tid == 0) kcount = 0;
    __syncthreads();
    for (size_t m = tid; m < M; m += 256) {
        const T v = y[m];
        if (v != v) { const int slot = atomicAdd(&kcount, 1); if (slot < kLsMaxMissing) miss[slot] = (int)m; }
    }
    __syncthreads();
    const int k = kcount < kmax ? kcount : kmax, kk = k + 1;
    if (k == 0) return;
    if (tid == 0) { for (int i = 1; i < k; i++) { const int v = miss[i]; int j = i - 1; while (j >= 0 && miss[j] > v) { miss[j + 1] = miss[j]; j--; } miss[j + 1] = v; } }
    for (int e = tid; e < k * kk; e += 256) {            // augmented system (I - G | b)
        const int i = e / kk, j = e % kk;
        Gs[e] = j < k ? ((i == j ? 1.0 : 0.0) - rec[kmax + i * kmax + j]) : rec[i];
    }
    __syncthreads();
    if (wave == 0) {                                       // symmetric positive definite: elimination without pivoting, lane = row (as ls_project_kernel)
        for (int p = 0; p < k; p++) {
            const double piv = Gs[p * kk + p];
            if (lane > p && lane < k) {
                const double f = Gs[lane * kk + p] / piv;
                for (int j = p; j < kk; j++) Gs[lane * kk + j] -= f * Gs[p * kk + j];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
            for (int i = k - 1; i >= 0; i--) {
                double sacc = Gs[i * kk + k];
                for (int j = i + 1; j < k; j++) sacc -= Gs[i * kk + j] * Gs[j * kk + k];
                Gs[i * kk + k] = sacc / Gs[i * kk + i];
            }
        }
    }
    __syncthreads();
    for (size_t l = tid; l < Lr; l += 256) {
        double a = 0.0;
        for (int i = 0; i < k; i++) a = fma(U[(size_t)miss[i] * Lr + l], Gs[i * kk + k], a);
        Ty[l * ld + t] = (T)((double)Ty[l * ld + t] + invsqrtS[l] * a);
    }
}
}  // namespace

static inline unsigned nblk(size_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

void launch_project_tick(const TickArgs& a, const double* y, double* Ty, double* Uty, double* part /* [32][L] scratch */, hipStream_t s) {
    if (a.M <= 128 || !part) {
        hipLaunchKernelGGL(project_tick_kernel, dim3(nblk(a.L, 128)), dim3(128), 0, s, a.M, a.L, a.U, a.S, y, Ty, Uty);
    } else {
        size_t nchunk = (a.M + 63) / 64;
        if (nchunk > 32) nchunk = 32;
        const size_t rpc = (a.M + nchunk - 1) / nchunk;
        nchunk = (a.M + rpc - 1) / rpc;
        hipLaunchKernelGGL(project_partial_kernel, dim3(nblk(a.L, 64), (unsigned)nchunk), dim3(256), 0, s, a.M, a.L, rpc, a.U, y, part);
        hipLaunchKernelGGL(project_finish_kernel, dim3(nblk(a.L, 256)), dim3(256), 0, s, a.L, nchunk, part, a.S, Ty, Uty);
    }
    MOIHGP_HIP_FATAL(hipGetLastError());
}

void launch_project_tick_missing(const TickArgs& a, const double* y, double* Ty, double* work, hipStream_t s) {
    double* N = work;
    double* r = work + a.L * a.L;
    hipLaunchKernelGGL(normal_eq_kernel, dim3(nblk(a.L * a.L, 128)), dim3(128), 0, s, a.M, a.L, a.U, y, N, r);
    hipLaunchKernelGGL(spd_solve_kernel, dim3(1), dim3(256), 0, s, a.L, N, r, a.S, Ty);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

// max |G - I| over an L x L matrix (G = U^T U from launch_gram): how far the mixing is from orthonormal columns
__global__ void __launch_bounds__(1024) ortho_defect_kernel(const double* __restrict__ G, size_t L, double* __restrict__ out) {
    __shared__ double red[16];
    double mx = 0.0;
    for (size_t i = threadIdx.x; i < L * L; i += 1024) {
        const double d = fabs(G[i] - ((i / L == i % L) ? 1.0 : 0.0));
        mx = (d > mx || d != d) ? d : mx;               // (NaN sticks)
    }
    for (int o = 32; o > 0; o >>= 1) { const double v = __shfl_xor(mx, o); mx = (v > mx || v != v) ? v : mx; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m2 = 0.0;
        for (int w = 0; w < 16; w++) m2 = (red[w] > m2 || red[w] != red[w]) ? red[w] : m2;
        *out = m2;
    }
}
void launch_ortho_defect(const double* G, size_t L, double* out, hipStream_t s) {
    hipLaunchKernelGGL(ortho_defect_kernel, dim3(1), dim3(1024), 0, s, G, L, out);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

int launch_project_stream_missing(int dtype, const void* Y, size_t T, size_t M, size_t L, const double* U, const double* invsqrtS, void* Ty, size_t ld,
                                   hipStream_t s) {
    if (T == 0 || L == 0) return 0;
    if (!ls_project_fits(L)) return 0;                 // (the NaN columns stand: include/moihgp.h)
    const size_t smem = ls_project_lds_bytes(L);
    if (dtype == 0) {
        if (smem > 48 * 1024) MOIHGP_HIP_FATAL(hipFuncSetAttribute(reinterpret_cast<const void*>(ls_project_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(ls_project_kernel<double>, dim3((unsigned)T), dim3(256), smem, s, (const double*)Y, T, M, L, U, invsqrtS, (double*)Ty, ld);
    } else {
        if (smem > 48 * 1024) MOIHGP_HIP_FATAL(hipFuncSetAttribute(reinterpret_cast<const void*>(ls_project_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(ls_project_kernel<float>, dim3((unsigned)T), dim3(256), smem, s, (const float*)Y, T, M, L, U, invsqrtS, (float*)Ty, ld);
    }
    MOIHGP_HIP_FATAL(hipGetLastError());
    return 0;
}

int launch_ls_shard(int phase, int dtype, const void* Y, size_t M, size_t Lr, const int* ticks, size_t n, int kmax, const double* U, const double* sqrtS,
                    const double* invsqrtS, double* packed, void* Ty, size_t ld, hipStream_t s) {
    if (n == 0) return 0;
    if (kmax < 1 || kmax > kLsMaxMissing) { set_last_error("ls_shard: kmax must be 1 .. %d", kLsMaxMissing); return 1; }
    if (phase == 0) {
        if (dtype == 0) hipLaunchKernelGGL(ls_shard_gram_kernel<double>, dim3((unsigned)n), dim3(256), 0, s, (const double*)Y, M, Lr, ticks, kmax, U, sqrtS, (const double*)Ty, ld, packed);
        else hipLaunchKernelGGL(ls_shard_gram_kernel<float>, dim3((unsigned)n), dim3(256), 0, s, (const float*)Y, M, Lr, ticks, kmax, U, sqrtS, (const float*)Ty, ld, packed);
    } else {
        if (dtype == 0) hipLaunchKernelGGL(ls_shard_apply_kernel<double>, dim3((unsigned)n), dim3(256), 0, s, (const double*)Y, M, Lr, ticks, kmax, U, invsqrtS, packed, (double*)Ty, ld);
        else hipLaunchKernelGGL(ls_shard_apply_kernel<float>, dim3((unsigned)n), dim3(256), 0, s, (const float*)Y, M, Lr, ticks, kmax, U, invsqrtS, packed, (float*)Ty, ld);
    }
    MOIHGP_HIP_FATAL(hipGetLastError());
    return 0;
}

void launch_step_tick(const TickArgs& a, const double* x, const double* Ty, const double* dx, double* xnew,
                      double* Tyhat, double* dxnew, hipStream_t s) {
    dim3 grid(nblk(a.L, 64)), block(64);
    if (a.d > 3)
        hipLaunchKernelGGL(step_tick_x_kernel, grid, block, 0, s, a.L, a.d, a.P, a.cb64, a.cbd64, x, Ty, dx, xnew, Tyhat, dxnew);
    else if (a.d == 2)
        hipLaunchKernelGGL(step_tick_kernel<2>, grid, block, 0, s, a.L, a.cb64, x, Ty, dx, xnew, Tyhat, dxnew);
    else
        hipLaunchKernelGGL(step_tick_kernel<3>, grid, block, 0, s, a.L, a.cb64, x, Ty, dx, xnew, Tyhat, dxnew);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

// ---- small models: the whole of MOIHGP::step (moihgp.h:148-428, no missing outputs) as ONE workgroup -------------------------
// project -> L filter steps -> unproject with workgroup barriers in between, inputs read straight from page-locked mapped host
// memory, outputs and finally a sequence number written straight back to it: a call is one launch and a spin on that number
// (8-10 us) instead of a copy, three or four launches and a stream synchronisation (20-24 us).
template <int D>
__global__ void __launch_bounds__(256) fused_step_kernel(size_t M, size_t L, const double* __restrict__ cb, const double* __restrict__ U,
                                                         const double* __restrict__ S, const double* __restrict__ x,
                                                         const double* __restrict__ y, const double* __restrict__ dx,
                                                         double* __restrict__ xnew, double* __restrict__ yhat, double* __restrict__ dxnew,
                                                         volatile unsigned long long* flag, unsigned long long seq) {
    using Lay = CB<D>;
    extern __shared__ double sm[];
    double* sy = sm;            // [M]
    double* sTy = sm + M;       // [L]
    double* sTyh = sTy + L;     // [L]
    const int tid = threadIdx.x;
    if (y) {
        for (size_t m = tid; m < M; m += 256) sy[m] = y[m];
        __syncthreads();
        for (size_t l = tid; l < L; l += 256) {                          // moihgp.h:181
            double s = 0.0;
            for (size_t m = 0; m < M; m++) s += U[m * L + l] * sy[m];
            sTy[l] = (1.0 / sqrt(S[l])) * s;
        }
        __syncthreads();
    }
    for (size_t l = tid; l < L; l += 256) {                              // ihgp.h:37-100, as step_tick_kernel
        const double* c = cb + l * Lay::SIZE;
        double xs[D], xn[D];
        for (int i = 0; i < D; i++) xs[i] = x[l * D + i];
        const bool miss = (y == nullptr);
        const double yl = miss ? 0.0 : sTy[l];
        const double* Mx = c + (miss ? Lay::A : Lay::AKHA);
        for (int i = 0; i < D; i++) {
            double s = 0.0;
            for (int k = 0; k < D; k++) s += Mx[i * D + k] * xs[k];
            xn[i] = miss ? s : s + c[Lay::K + i] * yl;
        }
        for (int i = 0; i < D; i++) xnew[l * D + i] = xn[i];
        sTyh[l] = xn[0];
        if (dx && dxnew) {
            for (int p = 0; p < P; p++) {
                const double* dM = c + (miss ? Lay::DA : Lay::DAKHA) + p * D * D;
                for (int i = 0; i < D; i++) {
                    double a = 0.0, b = 0.0;
                    for (int k = 0; k < D; k++) { a += dM[i * D + k] * xs[k]; b += Mx[i * D + k] * dx[(l * P + p) * D + k]; }
                    double v = a + b;
                    if (!miss) v += c[Lay::DK + p * D + i] * yl;
                    dxnew[(l * P + p) * D + i] = v;
                }
            }
        }
    }
    __syncthreads();
    if (yhat) {                                                          // moihgp.h:222-225, one wave per output row
        const int lane = tid & 63;
        for (size_t m = tid >> 6; m < M; m += 4) {
            double s = 0.0;
            for (size_t l = lane; l < L; l += 64) s += U[m * L + l] * (sqrt(S[l]) * sTyh[l]);
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) yhat[m] = s;
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) *flag = seq;
}

// (measured: one workgroup wins up to a few thousand mixing entries -- 8x4: 24 -> 13 us per call; at 256x256 it loses, 77 vs 28 us)
bool fused_step_fits(size_t M, size_t L) { return M * L <= 8192 && (M + 2 * L) * sizeof(double) <= 48 * 1024; }

void launch_fused_step(const TickArgs& a, const double* x, const double* y, const double* dx, double* xnew, double* yhat, double* dxnew,
                       unsigned long long* flag, unsigned long long seq, hipStream_t s) {
    const size_t smem = (a.M + 2 * a.L) * sizeof(double);
    if (a.d == 2)
        hipLaunchKernelGGL(fused_step_kernel<2>, dim3(1), dim3(256), smem, s, a.M, a.L, a.cb64, a.U, a.S, x, y, dx, xnew, yhat, dxnew, flag, seq);
    else
        hipLaunchKernelGGL(fused_step_kernel<3>, dim3(1), dim3(256), smem, s, a.M, a.L, a.cb64, a.U, a.S, x, y, dx, xnew, yhat, dxnew, flag, seq);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

// The same for MOIHGP::negLogLikelihood (moihgp.h:460-688, every output observed): projection, per-latent terms, residual,
// loss, and with dx the whole gradient vector, one workgroup, results and the sequence number straight to mapped host memory.
template <int D>
__global__ void __launch_bounds__(256) fused_lik_kernel(size_t M, size_t L, const double* __restrict__ cb, const double* __restrict__ U,
                                                        const double* __restrict__ S, const double* __restrict__ sigma_p,
                                                        const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ dx,
                                                        double* __restrict__ loss, double* __restrict__ grad,
                                                        volatile unsigned long long* flag, unsigned long long seq, int lik1_latent_loss) {
    extern __shared__ double sm[];
    double* sy = sm;                 // [M]
    double* sres = sy + M;           // [M]
    double* sUty = sres + M;         // [L]
    double* sTy = sUty + L;          // [L]
    double* slos = sTy + L;          // [L]
    double* spv = slos + L;          // [L]
    double* sig = spv + L;           // [L][P]
    double* red = sig + L * P;       // [256]
    const int tid = threadIdx.x, lane = tid & 63;
    for (size_t m = tid; m < M; m += 256) sy[m] = y[m];
    __syncthreads();
    for (size_t l = tid; l < L; l += 256) {                              // moihgp.h:181
        double s = 0.0;
        for (size_t m = 0; m < M; m++) s += U[m * L + l] * sy[m];
        sUty[l] = s;
        sTy[l] = (1.0 / sqrt(S[l])) * s;
    }
    __syncthreads();
    for (size_t l = tid; l < L; l += 256) {
        spv[l] = 0.0;
        igp_nll_one<D>(cb + l * CB<D>::SIZE, x + l * D, sy[l], sTy[l], dx ? dx + l * P * D : nullptr, slos + l, spv + l, sig + l * P);
    }
    for (size_t m = tid >> 6; m < M; m += 4) {                           // moihgp.h:501 / :651, one wave per row
        double s = 0.0;
        for (size_t l = lane; l < L; l += 64) s += U[m * L + l] * sUty[l];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) { const double r = sy[m] - s; sres[m] = r * r; }
    }
    __syncthreads();
    const double sigma = *sigma_p;
    nll_finalize_body(M, L, S, sigma, sUty, slos, spv, sig, sres, loss, dx ? grad : nullptr, red, !dx || lik1_latent_loss != 0);
    if (dx && grad)
        for (size_t idx = tid; idx < M * L; idx += 256) {                // moihgp.h:538-552 in its rank-1 form (ugrad_kernel)
            const size_t r = idx / L, c = idx % L;
            grad[idx] = sy[r] * (spv[c] * (1.0 / sqrt(S[c])) - sUty[c] / sigma);
        }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) *flag = seq;
}

static size_t fused_lik_smem(size_t M, size_t L) { return (2 * M + (4 + P) * L + 256) * sizeof(double); }
bool fused_lik_fits(size_t M, size_t L) { return M * L <= 8192 && M >= L && fused_lik_smem(M, L) <= 48 * 1024; }

void launch_fused_lik(const TickArgs& a, const double* x, const double* y, const double* dx, double* loss, double* grad,
                      unsigned long long* flag, unsigned long long seq, hipStream_t s) {
    const size_t smem = fused_lik_smem(a.M, a.L);
    if (a.d == 2)
        hipLaunchKernelGGL(fused_lik_kernel<2>, dim3(1), dim3(256), smem, s, a.M, a.L, a.cb64, a.U, a.S, a.sigma, x, y, dx, loss, grad, flag, seq, a.lik1_latent_loss);
    else
        hipLaunchKernelGGL(fused_lik_kernel<3>, dim3(1), dim3(256), smem, s, a.M, a.L, a.cb64, a.U, a.S, a.sigma, x, y, dx, loss, grad, flag, seq, a.lik1_latent_loss);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

void launch_unproject_tick(const TickArgs& a, const double* Tyhat, double* yhat, hipStream_t s) {
    hipLaunchKernelGGL(unproject_tick_kernel, dim3(nblk(a.M, 4)), dim3(256), 0, s, a.M, a.L, a.U, a.S, Tyhat, yhat);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

void launch_nll_tick(const TickArgs& a, const double* x, const double* y, const double* Ty, const double* Uty,
                     const double* dx, double* loss, double* grad, double* scratch, hipStream_t s) {
    double* lossv = scratch;                 // [L]
    double* pv = lossv + a.L;                // [L]
    double* igrad = pv + a.L;                // [L][P]
    double* resid2 = igrad + a.L * a.P;      // [M]
    dim3 grid(nblk(a.L, 64)), block(64);
    if (a.d > 3)
        hipLaunchKernelGGL(igp_nll_x_kernel, grid, block, 0, s, a.L, a.d, a.P, a.cb64, a.cbd64, x, y, Ty, dx, lossv, pv, igrad);
    else if (a.d == 2)
        hipLaunchKernelGGL(igp_nll_kernel<2>, grid, block, 0, s, a.L, a.cb64, x, y, Ty, dx, lossv, pv, igrad);
    else
        hipLaunchKernelGGL(igp_nll_kernel<3>, grid, block, 0, s, a.L, a.cb64, x, y, Ty, dx, lossv, pv, igrad);
    hipLaunchKernelGGL(resid_kernel, dim3(nblk(a.M, 4)), dim3(256), 0, s, a.M, a.L, a.U, y, Uty, resid2);
    hipLaunchKernelGGL(nll_finalize_kernel, dim3(1), dim3(256), 0, s, a.M, a.L, a.S, a.sigma, Uty, lossv, pv, igrad,
                       resid2, loss, dx ? grad : nullptr, (!dx || a.lik1_latent_loss) ? 1 : 0, a.P);
    if (dx && grad)
        hipLaunchKernelGGL(ugrad_kernel, dim3(nblk(a.M * a.L, 256)), dim3(256), 0, s, a.M, a.L, a.S, a.sigma, y, Uty, pv, grad);
    MOIHGP_HIP_FATAL(hipGetLastError());
}

}  // namespace moihgp
