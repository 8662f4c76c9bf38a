// recursion.hip -- the hot path: per-latent steady-state Kalman recursion + NLL over whole
// time streams (reference include/moihgp/ihgp.h:81-100 step, :204-222 negLogLikelihood, driven
// tick by tick by moihgp.h:367-373 / moihgp_online.h:61-70 / moihgp_regression.h:42-50).
//
// Mapping (MI355X, wave64): ONE WAVEFRONT OWNS ONE LATENT.  The latent's stream is series-major
// (contiguous in time), so the wave reads it with fully coalesced 16-byte-per-lane loads, 64*CK
// ticks ("segment") at a time.  Inside a segment the 64 lanes are 64 consecutive time chunks of
// CK ticks.  The recursion x <- AKHA x + K y is linear time-invariant, so the segment is solved
// exactly (up to rounding) in three steps without any serial walk over 64*CK ticks:
//   1. every lane runs its CK ticks from a zero state (lane 0 from the carried-in state) -> z_j
//   2. a 6-step Kogge-Stone scan over lanes composes the chunk maps: s_j = M s_{j-1} + z_j with
//      M = AKHA^CK and the uniform powers M^(2^k) (per-latent constants, computed once per wave)
//   3. every lane re-runs its CK ticks from its true start state, emitting yhat / NLL terms.
// The per-latent matrices are wave-uniform (scalar registers); the y chunk lives in VGPRs between
// pass 1 and pass 3, so HBM sees each stream element exactly once in and once out.
// The coalesced <-> chunk-per-lane re-layout goes through a wave-private, padded LDS tile
// (no workgroup barrier: the four waves of a block never communicate).
//
// Missing data (NaN y, ihgp.h:83-87: x <- A x) and the ragged tail segment make chunk maps
// lane-dependent; those segments take the generic path that scans (M_j, z_j) pairs.
#include "common.h"

namespace moihgp {
namespace {

template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = float4; };
template <> struct VecOf<double> { using type = double2; };

template <typename T> __device__ inline void unpack(const typename VecOf<T>::type& v, T* out);
template <> __device__ inline void unpack<float>(const float4& v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ inline void unpack<double>(const double2& v, double* o) { o[0] = v.x; o[1] = v.y; }
template <typename T> __device__ inline typename VecOf<T>::type pack(const T* in);
template <> __device__ inline float4 pack<float>(const float* i) { return make_float4(i[0], i[1], i[2], i[3]); }
template <> __device__ inline double2 pack<double>(const double* i) { return make_double2(i[0], i[1]); }

__device__ inline void wave_lds_fence() {
    // LDS operations of one wave execute in program order; this only stops the compiler from
    // moving LDS accesses across the hand-over between lanes of the same wave.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int D>
__device__ inline void matvec_acc(const T* m, const T* v, T* out /* out = m v + out */) {
#pragma unroll
    for (int i = 0; i < D; i++) {
        T s = out[i];
#pragma unroll
        for (int j = 0; j < D; j++) s = fma(m[i * D + j], v[j], s);
        out[i] = s;
    }
}

template <typename T, int D>
__device__ inline void matmul(const T* a, const T* b, T* c) {
    T t[D * D];
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = 0; j < D; j++) {
            T s = 0;
#pragma unroll
            for (int k = 0; k < D; k++) s = fma(a[i * D + k], b[k * D + j], s);
            t[i * D + j] = s;
        }
#pragma unroll
    for (int i = 0; i < D * D; i++) c[i] = t[i];
}

constexpr int kWavesPerBlock = 4;

// ---------------------------------------------------------------------------------------------
template <typename T, int D, int CK, bool WRITE, bool NLL>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
filter_scan_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT,
                   const double* __restrict__ cb64, T* __restrict__ x, T* __restrict__ yhat, double* __restrict__ nll) {
    using V = typename VecOf<T>::type;
    using Lay = CB<D>;
    constexpr int EPV = 16 / sizeof(T);        // elements per 16-byte vector
    constexpr int VPL = CK / EPV;              // vectors per lane per segment
    constexpr int SEG = 64 * CK;               // ticks per segment
    constexpr int NVP = 64 * (VPL + 1);        // padded vectors per wave tile (one pad vector per lane row)
    static_assert(CK % EPV == 0 && (CK & (CK - 1)) == 0, "CK must be a power of two multiple of 16 bytes");
    __shared__ V lds_all[kWavesPerBlock][NVP];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = (size_t)blockIdx.x * kWavesPerBlock + wave;
    if (l >= L) return;
    V* lds = lds_all[wave];

    // ---- wave-uniform per-latent constants -------------------------------------------------
    const T* c = cbT + l * Lay::SIZE;
    T akha[D * D], kk[D], aa[D * D], ha[D];
#pragma unroll
    for (int i = 0; i < D * D; i++) { akha[i] = c[Lay::AKHA + i]; aa[i] = c[Lay::A + i]; }
#pragma unroll
    for (int i = 0; i < D; i++) { kk[i] = c[Lay::K + i]; ha[i] = c[Lay::HA + i]; }
    // scan powers M^(2^s), M = AKHA^CK, from the fp64 master copy
    T mp[6][D * D];
    {
        const double* c64 = cb64 + l * Lay::SIZE;
        double m[D * D];
#pragma unroll
        for (int i = 0; i < D * D; i++) m[i] = c64[Lay::AKHA + i];
#pragma unroll
        for (int s = 1; s < CK; s <<= 1) matmul<double, D>(m, m, m);
#pragma unroll
        for (int s = 0; s < 6; s++) {
#pragma unroll
            for (int i = 0; i < D * D; i++) mp[s][i] = (T)m[i];
            if (s < 5) matmul<double, D>(m, m, m);
        }
    }

    const T* row = Ty + l * ld;
    T* orow = WRITE ? yhat + l * ld : nullptr;
    T xin[D];
#pragma unroll
    for (int i = 0; i < D; i++) xin[i] = x[l * D + i];

    double acc = 0.0;          // per-lane sum of v^2 over observed ticks
    unsigned nobs = 0;         // per-lane count of observed ticks handled by the generic path
    size_t nobs_uniform = 0;   // observed ticks handled by the fast path (same for all lanes: count per wave)

    const size_t nfull = Tlen / SEG;
    const size_t nseg = (Tlen + SEG - 1) / SEG;

    V r[VPL];
    auto load_seg = [&](size_t seg) {
        const size_t base = seg * SEG;
        if (seg < nfull) {
#pragma unroll
            for (int i = 0; i < VPL; i++) r[i] = *reinterpret_cast<const V*>(row + base + (size_t)(i * 64 + lane) * EPV);
        } else {
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                size_t t0 = base + (size_t)(i * 64 + lane) * EPV;
                T zero[EPV] = {};
                r[i] = (t0 < Tlen) ? *reinterpret_cast<const V*>(row + t0) : pack<T>(zero);   // ld >= roundup(T, EPV)
            }
        }
    };
    if (nseg > 0) load_seg(0);

    for (size_t seg = 0; seg < nseg; seg++) {
        const size_t tbase = seg * SEG;
        // ---- coalesced registers -> LDS -> chunk-per-lane registers ---------------------------
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            int q = i * 64 + lane;
            lds[q + q / VPL] = r[i];
        }
        wave_lds_fence();
        T y[CK];
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            V v = lds[lane * (VPL + 1) + k];
            unpack<T>(v, &y[k * EPV]);
        }
        wave_lds_fence();
        if (seg + 1 < nseg) load_seg(seg + 1);   // prefetch: in flight during the arithmetic below

        bool generic = (seg >= nfull);            // ragged tail
        T z[D];
        if (!generic) {
            // ---- pass 1: chunk response from zero state (lane 0: from the carried-in state) ----
#pragma unroll
            for (int i = 0; i < D; i++) z[i] = (lane == 0) ? xin[i] : T(0);
#pragma unroll
            for (int k = 0; k < CK; k++) {
                T zn[D];
#pragma unroll
                for (int i = 0; i < D; i++) zn[i] = kk[i] * y[k];
                matvec_acc<T, D>(akha, z, zn);
#pragma unroll
                for (int i = 0; i < D; i++) z[i] = zn[i];
            }
            // a NaN anywhere in the chunk poisons z: route the whole segment to the generic path
            bool bad = false;
#pragma unroll
            for (int i = 0; i < D; i++) bad |= (z[i] != z[i]);
            generic = __any(bad);
        }
        if (!generic) {
            // ---- Kogge-Stone scan with uniform chunk map powers ------------------------------
#pragma unroll
            for (int s = 0; s < 6; s++) {
                const int o = 1 << s;
                T t[D];
#pragma unroll
                for (int i = 0; i < D; i++) t[i] = __shfl_up(z[i], o);
                if (lane >= o) matvec_acc<T, D>(mp[s], t, z);
            }
            T xs[D];
#pragma unroll
            for (int i = 0; i < D; i++) {
                T up = __shfl_up(z[i], 1);
                xs[i] = (lane == 0) ? xin[i] : up;
                xin[i] = __shfl(z[i], 63);
            }
            // ---- pass 2: replay the chunk from its true start state ---------------------------
#pragma unroll
            for (int k = 0; k < CK; k++) {
                if (NLL) {
                    T hx = 0;
#pragma unroll
                    for (int i = 0; i < D; i++) hx = fma(ha[i], xs[i], hx);
                    double v = (double)(y[k] - hx);
                    acc = fma(v, v, acc);
                }
                T xn[D];
#pragma unroll
                for (int i = 0; i < D; i++) xn[i] = kk[i] * y[k];
                matvec_acc<T, D>(akha, xs, xn);
#pragma unroll
                for (int i = 0; i < D; i++) xs[i] = xn[i];
                y[k] = xs[0];
            }
            nobs_uniform += SEG;
        } else {
            // ---- generic path: lane-dependent chunk maps (missing ticks, ragged tail) ----------
            const size_t t0 = tbase + (size_t)lane * CK;
            T mj[D * D];
#pragma unroll
            for (int i = 0; i < D * D; i++) mj[i] = (i % (D + 1) == 0) ? T(1) : T(0);
#pragma unroll
            for (int i = 0; i < D; i++) z[i] = (lane == 0) ? xin[i] : T(0);
#pragma unroll
            for (int k = 0; k < CK; k++) {
                const bool valid = (t0 + k) < Tlen;
                const T yk = y[k];
                const bool miss = (yk != yk);
                if (valid) {
                    T b[D * D], zn[D];
#pragma unroll
                    for (int i = 0; i < D * D; i++) b[i] = miss ? aa[i] : akha[i];
#pragma unroll
                    for (int i = 0; i < D; i++) zn[i] = miss ? T(0) : kk[i] * yk;
                    matvec_acc<T, D>(b, z, zn);
#pragma unroll
                    for (int i = 0; i < D; i++) z[i] = zn[i];
                    matmul<T, D>(b, mj, mj);
                }
            }
#pragma unroll
            for (int s = 0; s < 6; s++) {
                const int o = 1 << s;
                T zp[D], mq[D * D];
#pragma unroll
                for (int i = 0; i < D; i++) zp[i] = __shfl_up(z[i], o);
#pragma unroll
                for (int i = 0; i < D * D; i++) mq[i] = __shfl_up(mj[i], o);
                if (lane >= o) {
                    matvec_acc<T, D>(mj, zp, z);
                    matmul<T, D>(mj, mq, mj);
                }
            }
            T xs[D];
#pragma unroll
            for (int i = 0; i < D; i++) {
                T up = __shfl_up(z[i], 1);
                xs[i] = (lane == 0) ? xin[i] : up;
                xin[i] = __shfl(z[i], 63);
            }
#pragma unroll
            for (int k = 0; k < CK; k++) {
                const bool valid = (t0 + k) < Tlen;
                const T yk = y[k];
                const bool miss = (yk != yk);
                if (valid) {
                    if (NLL && !miss) {
                        T hx = 0;
#pragma unroll
                        for (int i = 0; i < D; i++) hx = fma(ha[i], xs[i], hx);
                        double v = (double)(yk - hx);
                        acc = fma(v, v, acc);
                        nobs++;
                    }
                    T b[D * D], xn[D];
#pragma unroll
                    for (int i = 0; i < D * D; i++) b[i] = miss ? aa[i] : akha[i];
#pragma unroll
                    for (int i = 0; i < D; i++) xn[i] = miss ? T(0) : kk[i] * yk;
                    matvec_acc<T, D>(b, xs, xn);
#pragma unroll
                    for (int i = 0; i < D; i++) xs[i] = xn[i];
                    y[k] = xs[0];
                }
            }
        }

        // ---- chunk-per-lane registers -> LDS -> coalesced stores ------------------------------
        if (WRITE) {
#pragma unroll
            for (int k = 0; k < VPL; k++) lds[lane * (VPL + 1) + k] = pack<T>(&y[k * EPV]);
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                int q = i * 64 + lane;
                V o = lds[q + q / VPL];
                size_t tq = tbase + (size_t)q * EPV;
                if (seg < nfull || tq < Tlen) *reinterpret_cast<V*>(orow + tq) = o;
            }
            wave_lds_fence();
        }
    }

    // ---- epilogue: carried-out state and the latent's NLL ---------------------------------------
    if (NLL) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            acc += __shfl_xor(acc, o);
            nobs += __shfl_xor(nobs, o);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < D; i++) x[l * D + i] = xin[i];
        if (NLL) {
            const double* c64 = cb64 + l * Lay::SIZE;
            double n = (double)nobs_uniform + (double)nobs;
            nll[l] = 0.5 * (acc / c64[Lay::S] + n * c64[Lay::LOGS]);   // sum of ihgp.h:207 terms
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Sensitivity sweep (ihgp.h:37-57 step with dx, :212-222 NLL gradient), one lane per latent,
// sequential in time.  First correct version for the learning sweeps (short windows W <= 128,
// moihgp_online.h:61-70); the long-stream form reuses the scan structure above in a later round.
template <typename T, int D>
__global__ void __launch_bounds__(64)
grad_seq_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT,
                const double* __restrict__ cb64, T* __restrict__ x, T* __restrict__ dx, T* __restrict__ yhat,
                double* __restrict__ nll, double* __restrict__ grad) {
    using Lay = CB<D>;
    constexpr int P = kNumIgpParam;
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const T* c = cbT + l * Lay::SIZE;
    const double* c64 = cb64 + l * Lay::SIZE;
    T akha[D * D], aa[D * D], kk[D], ha[D], dakha[P][D * D], da[P][D * D], dk[P][D], hda[P][D];
    for (int i = 0; i < D * D; i++) { akha[i] = c[Lay::AKHA + i]; aa[i] = c[Lay::A + i]; }
    for (int i = 0; i < D; i++) { kk[i] = c[Lay::K + i]; ha[i] = c[Lay::HA + i]; }
    for (int p = 0; p < P; p++) {
        for (int i = 0; i < D * D; i++) { dakha[p][i] = c[Lay::DAKHA + p * D * D + i]; da[p][i] = c[Lay::DA + p * D * D + i]; }
        for (int i = 0; i < D; i++) { dk[p][i] = c[Lay::DK + p * D + i]; hda[p][i] = c[Lay::HDA + p * D + i]; }
    }
    const double S = c64[Lay::S], logS = c64[Lay::LOGS];
    double dS[P];
    for (int p = 0; p < P; p++) dS[p] = c64[Lay::DS + p];
    T xs[D], dxs[P][D];
    for (int i = 0; i < D; i++) xs[i] = x[l * D + i];
    for (int p = 0; p < P; p++)
        for (int i = 0; i < D; i++) dxs[p][i] = dx[(l * P + p) * D + i];
    double acc = 0.0, g[P] = {0.0, 0.0, 0.0};
    const T* row = Ty + l * ld;
    for (size_t t = 0; t < Tlen; t++) {
        const T y = row[t];
        const bool miss = (y != y);
        T xn[D], dxn[P][D];
        if (!miss) {
            T hx = 0;
            for (int i = 0; i < D; i++) hx = fma(ha[i], xs[i], hx);
            double v = (double)(y - hx);
            acc += 0.5 * (v * v / S + logS);                            // ihgp.h:215
            for (int p = 0; p < P; p++) {
                T a = 0, b = 0;
                for (int i = 0; i < D; i++) { a = fma(hda[p][i], xs[i], a); b = fma(ha[i], dxs[p][i], b); }
                double dv = (double)(-a - b);                           // ihgp.h:218
                g[p] += (v * dv - 0.5 * (v * v / S - 1.0) * dS[p]) / S;   // ihgp.h:219
            }
            for (int i = 0; i < D; i++) xn[i] = kk[i] * y;
            matvec_acc<T, D>(akha, xs, xn);                             // ihgp.h:50
            for (int p = 0; p < P; p++) {
                for (int i = 0; i < D; i++) dxn[p][i] = dk[p][i] * y;
                matvec_acc<T, D>(dakha[p], xs, dxn[p]);
                matvec_acc<T, D>(akha, dxs[p], dxn[p]);                 // ihgp.h:54
            }
        } else {
            for (int i = 0; i < D; i++) xn[i] = 0;
            matvec_acc<T, D>(aa, xs, xn);                               // ihgp.h:41
            for (int p = 0; p < P; p++) {
                for (int i = 0; i < D; i++) dxn[p][i] = 0;
                matvec_acc<T, D>(da[p], xs, dxn[p]);
                matvec_acc<T, D>(aa, dxs[p], dxn[p]);                   // ihgp.h:45
            }
        }
        for (int i = 0; i < D; i++) xs[i] = xn[i];
        for (int p = 0; p < P; p++)
            for (int i = 0; i < D; i++) dxs[p][i] = dxn[p][i];
        if (yhat) yhat[l * ld + t] = xn[0];
    }
    for (int i = 0; i < D; i++) x[l * D + i] = xs[i];
    for (int p = 0; p < P; p++)
        for (int i = 0; i < D; i++) dx[(l * P + p) * D + i] = dxs[p][i];
    if (nll) nll[l] = acc;
    for (int p = 0; p < P; p++) grad[l * P + p] = g[p];
}

template <typename T, int D, int CK>
int launch_filter_t(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, void* x,
                    void* yhat, double* nll, hipStream_t stream) {
    dim3 block(64 * kWavesPerBlock), grid((unsigned)((L + kWavesPerBlock - 1) / kWavesPerBlock));
    const T* ty = static_cast<const T*>(Ty);
    T* xs = static_cast<T*>(x);
    T* yh = static_cast<T*>(yhat);
    if (yhat && nll)
        hipLaunchKernelGGL((filter_scan_kernel<T, D, CK, true, true>), grid, block, 0, stream, ty, Tlen, ld, L, cbT, cb64, xs, yh, nll);
    else if (yhat)
        hipLaunchKernelGGL((filter_scan_kernel<T, D, CK, true, false>), grid, block, 0, stream, ty, Tlen, ld, L, cbT, cb64, xs, yh, nll);
    else if (nll)
        hipLaunchKernelGGL((filter_scan_kernel<T, D, CK, false, true>), grid, block, 0, stream, ty, Tlen, ld, L, cbT, cb64, xs, yh, nll);
    else
        hipLaunchKernelGGL((filter_scan_kernel<T, D, CK, false, false>), grid, block, 0, stream, ty, Tlen, ld, L, cbT, cb64, xs, yh, nll);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("filter_scan_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

int launch_filter_stream(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64,
                         const float* cb32, void* x, void* yhat, double* nll, hipStream_t stream) {
    if (L == 0) return 0;
    if (dtype == 0) {
        if (d == 2) return launch_filter_t<double, 2, 8>(Ty, T, ld, L, cb64, cb64, x, yhat, nll, stream);
        return launch_filter_t<double, 3, 8>(Ty, T, ld, L, cb64, cb64, x, yhat, nll, stream);
    }
    if (d == 2) return launch_filter_t<float, 2, 16>(Ty, T, ld, L, cb32, cb64, x, yhat, nll, stream);
    return launch_filter_t<float, 3, 16>(Ty, T, ld, L, cb32, cb64, x, yhat, nll, stream);
}

int launch_grad_stream(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64,
                       const float* cb32, void* x, void* dx, void* yhat, double* nll, double* grad, hipStream_t stream) {
    if (L == 0) return 0;
    dim3 block(64), grid((unsigned)((L + 63) / 64));
    if (dtype == 0) {
        if (d == 2)
            hipLaunchKernelGGL((grad_seq_kernel<double, 2>), grid, block, 0, stream, (const double*)Ty, T, ld, L, cb64, cb64, (double*)x, (double*)dx, (double*)yhat, nll, grad);
        else
            hipLaunchKernelGGL((grad_seq_kernel<double, 3>), grid, block, 0, stream, (const double*)Ty, T, ld, L, cb64, cb64, (double*)x, (double*)dx, (double*)yhat, nll, grad);
    } else {
        if (d == 2)
            hipLaunchKernelGGL((grad_seq_kernel<float, 2>), grid, block, 0, stream, (const float*)Ty, T, ld, L, cb32, cb64, (float*)x, (float*)dx, (float*)yhat, nll, grad);
        else
            hipLaunchKernelGGL((grad_seq_kernel<float, 3>), grid, block, 0, stream, (const float*)Ty, T, ld, L, cb32, cb64, (float*)x, (float*)dx, (float*)yhat, nll, grad);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("grad_seq_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace moihgp
