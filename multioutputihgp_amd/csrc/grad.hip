// grad.hip -- sensitivity sweep over whole streams: the step with hyper-parameter sensitivities
// (reference include/moihgp/ihgp.h:37-57) and the per-latent NLL gradient (ihgp.h:212-222), accumulated over
// ticks in the order of the learners' loops (moihgp_online.h:61-70, moihgp_regression.h:42-50):
//     v = y - HA x ;  nll += 1/2 (v^2/S + log S) ;  dv_p = -HdA_p x - HA dx_p ;
//     grad_p += (v dv_p - 1/2 (v^2/S - 1) dS_p) / S                                   (pre-step x, dx)
//     x' = AKHA x + K y ;  dx_p' = dAKHA_p x + AKHA dx_p + dK_p y
//
// grad_scan_kernel: one wavefront owns one latent, lanes are consecutive time chunks of CK ticks, as in
// recursion.hip.  The mean x is solved per segment by the chunk response + DPP scan of the filter.  Each
// sensitivity dx_p obeys the SAME linear time-invariant recursion (transition AKHA) driven by the input
// u_t = dAKHA_p x_t + dK_p y_t, which is known once the true x trajectory of the chunk is replayed.  One replay per
// segment does everything: it walks the true x, accumulates the zero-state response dz_p of every sensitivity, and sums
// the gradient terms in split form -- by linearity dx_p(k) = dz_p(k) + AKHA^k dx_p(start), so
//     sum_k v_k dv_pk = sum_k v_k (-HdA_p x_k - HA dz_p(k))  -  (sum_k v_k HA AKHA^k) dx_p(start)
// with the row vector w = sum_k v_k HA AKHA^k accumulated beside it; the three dz_p are then scanned with the very same
// powers of M = AKHA^CK, which yields every chunk's dx_p(start) (to close the sums) and the segment's end state.
// Sums over ticks:   grad_p = (sum v dv_p)/S - 1/2 (sum v^2 / S - n) dS_p / S
// VALU-bound (about 115 vector ops per tick at d = 3, P = 3 against 4-8 bytes of stream): its roofline is the
// vector ALU, not HBM.
//
// Streams with missing ticks (NaN; ihgp.h:39-47 swaps AKHA, dAKHA, dK for A, dA, 0) break the uniform chunk maps:
// the scan kernel flags such a latent and grad_seq_kernel (one lane per latent, sequential, loads prefetched in
// 16-byte vectors) recomputes it.  The reference never feeds a NaN to IHGP through MOIHGP (SURVEY 8a notes).
#include "kernels_common.h"
#include <type_traits>

#ifndef MOIHGP_GRAD_SPREG
#define MOIHGP_GRAD_SPREG 1
#endif
#ifndef MOIHGP_GRAD_MINW
#define MOIHGP_GRAD_MINW 1
#endif

namespace moihgp {
namespace {

constexpr int P = kNumIgpParam;
// Parameters whose dF is identically zero: magnitude and noise; only the lengthscale moves F (matern32ss.h:54-55,
// matern52ss.h:61-63).  For those the reference itself stores dA = 0, HdA = 0 and dAKHA = -dK HA (ihgp.h:141-143, :189-193), so
// dAKHA_p x + dK_p y = dK_p (y - HA x) = dK_p v: the scan kernel uses that form (12 instead of 21 multiply-adds per tick).
__device__ constexpr bool kDFzero[P] = {true, false, true};

// Two sensitivities side by side: magnitude and noise (the dF = 0 parameters) obey the same recursion with the same matrices and
// differ only in their dK, so they travel as one 2-vector per state entry; in fp32 every multiply-add on the pair is ONE packed
// v_pk_fma_f32 with the matrix entry broadcast.  fp64 has no packed form and keeps the one-parameter-at-a-time code.
template <typename T> struct PairOf { typedef T type __attribute__((ext_vector_type(2))); };
template <typename T, int D>
__device__ inline void matvec_acc2(const T* m, const typename PairOf<T>::type* v, typename PairOf<T>::type* out) {
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = 0; j < D; j++) out[i] = m[i * D + j] * v[j] + out[i];
}
template <int CTRL, int ROW_MASK, typename T2>
__device__ inline T2 dpp0_2(T2 v) { T2 r; r.x = dpp0<CTRL, ROW_MASK>(v.x); r.y = dpp0<CTRL, ROW_MASK>(v.y); return r; }
template <typename T, int D>
__device__ inline void dpp_scan2(typename PairOf<T>::type* z, const T* sp, const T* pj) {
    typename PairOf<T>::type t[D];
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0_2<DPP_ROW_SHR + 1, 0xF>(z[i]);
    matvec_acc2<T, D>(sp + 0 * D * D, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0_2<DPP_ROW_SHR + 2, 0xF>(z[i]);
    matvec_acc2<T, D>(sp + 1 * D * D, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0_2<DPP_ROW_SHR + 4, 0xF>(z[i]);
    matvec_acc2<T, D>(sp + 2 * D * D, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0_2<DPP_ROW_SHR + 8, 0xF>(z[i]);
    matvec_acc2<T, D>(sp + 3 * D * D, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0_2<DPP_ROW_BCAST15, 0x2>(z[i]);
    matvec_acc2<T, D>(pj, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0_2<DPP_ROW_BCAST15, 0x4>(z[i]);
    matvec_acc2<T, D>(pj, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0_2<DPP_ROW_BCAST15, 0x8>(z[i]);
    matvec_acc2<T, D>(pj, t, z);
}
static_assert(P == 3 && kDFzero[0] && !kDFzero[1] && kDFzero[2], "the scan kernel pairs parameters 0 and 2 and treats 1 in full");

template <typename T, int D>
struct GradConst {
    T a[D * D], k[D], akha[D * D], dakha[P][D * D], dk[P][D], hda[P][D];
};

// The per-latent constants are wave-uniform and would all be scalar registers: 66 of them at d = 3, more than the scalar
// file can hold next to the scan tables, and the overflow comes back as v_readlane / v_writelane traffic in the tick loop.
// The sensitivity blocks are therefore pinned in vector registers (there is room), the mean's A, K, AKHA stay scalar.
#ifndef MOIHGP_GRAD_PINV
#define MOIHGP_GRAD_PINV 1
#endif
template <typename T>
__device__ inline T pin_vgpr(T v) {
#if MOIHGP_GRAD_PINV
    asm("" : "+v"(v));
#endif
    return v;
}

template <typename T, int D>
__device__ inline void load_grad_const(GradConst<T, D>& c, const T* cb) {
    using Lay = CB<D>;
#pragma unroll
    for (int i = 0; i < D * D; i++) { c.a[i] = cb[Lay::A + i]; c.akha[i] = cb[Lay::AKHA + i]; }
#pragma unroll
    for (int i = 0; i < D; i++) c.k[i] = cb[Lay::K + i];
#pragma unroll
    for (int p = 0; p < P; p++) {
#pragma unroll
        for (int i = 0; i < D * D; i++) c.dakha[p][i] = pin_vgpr(cb[Lay::DAKHA + p * D * D + i]);
#pragma unroll
        for (int i = 0; i < D; i++) { c.dk[p][i] = pin_vgpr(cb[Lay::DK + p * D + i]); c.hda[p][i] = pin_vgpr(cb[Lay::HDA + p * D + i]); }
    }
}

// one tick of the mean in innovation form (H = e0^T: HA is row 0 of A); returns v, writes yhat
template <typename T, int D>
__device__ inline T tick_mean(const GradConst<T, D>& c, T* xs, T y, T& hx_out) {
    T hx = 0;
#pragma unroll
    for (int j = 0; j < D; j++) hx = fma(c.a[j], xs[j], hx);
    const T v = y - hx;
    T xn[D];
    xn[0] = fma(c.k[0], v, hx);
#pragma unroll
    for (int i = 1; i < D; i++) {
        T s = c.k[i] * v;
#pragma unroll
        for (int j = 0; j < D; j++) s = fma(c.a[i * D + j], xs[j], s);
        xn[i] = s;
    }
#pragma unroll
    for (int i = 0; i < D; i++) xs[i] = xn[i];
    hx_out = hx;
    return v;
}

// WRITE: 0 no stream output, 1 filtered means yhat_t = x_{t+1}[0] (ihgp.h:51), 2 predicted means hx_t = HA x_t (pre-step; what
// MOIHGP::negLogLikelihood needs for `pv`, moihgp.h:505-512)
template <typename T, int D, int CK, int WRITE>
__global__ void __launch_bounds__(64 * kWavesPerBlock, MOIHGP_GRAD_MINW)
grad_scan_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT,
                 const double* __restrict__ cb64, T* __restrict__ x, T* __restrict__ dx, T* __restrict__ yhat,
                 double* __restrict__ nll, double* __restrict__ grad, int* __restrict__ fallback) {
    using V = typename VecOf<T>::type;
    using Lay = CB<D>;
    constexpr int EPV = 16 / sizeof(T), VPL = CK / EPV, SEG = 64 * CK, NVP = 64 * (VPL + 1);
    constexpr int NTAB = 2 * CK * D + 4 * D * D;                 // g table + scan powers + HA AKHA^k rows, per wave, in LDS
    constexpr int HP = CK * D + 4 * D * D;                       // offset of the HA AKHA^k rows
    static_assert(CK % EPV == 0, "CK must be a multiple of 16 bytes");
    __shared__ V lds_all[kWavesPerBlock][NVP];
    __shared__ T tab_all[kWavesPerBlock][NTAB];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = (size_t)blockIdx.x * kWavesPerBlock + wave;
    if (l >= L) return;
    V* lds = lds_all[wave];
    T* tab = tab_all[wave];
    const T* cb = cbT + l * Lay::SIZE;
    const double* c64 = cb64 + l * Lay::SIZE;

    // ---- tables of the segment solve for THIS chunk length, from the fp64 master copy: g_k = AKHA^(CK-1-k) K,
    //      sp = M^(1,2,4,8) with M = AKHA^CK (parked in LDS, re-read per segment), pj = M^(lane%16+1) (registers)
    T pj[D * D];
    {
        double ak[D * D], g[D], m[D * D], pw[D * D];
#pragma unroll
        for (int i = 0; i < D * D; i++) ak[i] = c64[Lay::AKHA + i];
#pragma unroll
        for (int i = 0; i < D; i++) g[i] = c64[Lay::K + i];
#pragma unroll
        for (int k = CK - 1; k >= 0; k--) {
#pragma unroll
            for (int i = 0; i < D; i++) if (lane == i) tab[k * D + i] = (T)g[i];
            double gn[D];
#pragma unroll
            for (int i = 0; i < D; i++) { double s = 0; for (int j = 0; j < D; j++) s = fma(ak[i * D + j], g[j], s); gn[i] = s; }
#pragma unroll
            for (int i = 0; i < D; i++) g[i] = gn[i];
        }
#pragma unroll
        for (int i = 0; i < D * D; i++) m[i] = ak[i];
#pragma unroll
        for (int q = 1; q < CK; q <<= 1) matmul<double, D>(m, m, m);          // CK is a power of two
#pragma unroll
        for (int i = 0; i < D * D; i++) pw[i] = m[i];
#pragma unroll
        for (int lv = 0; lv < 4; lv++) {
#pragma unroll
            for (int i = 0; i < D * D; i++) if (lane == i) tab[CK * D + lv * D * D + i] = (T)pw[i];
            matmul<double, D>(pw, pw, pw);
        }
        {   // hp_k = HA AKHA^k, k < CK: what a start sensitivity contributes to HA dx at tick k of its chunk
            double h[D];
#pragma unroll
            for (int i = 0; i < D; i++) h[i] = c64[Lay::HA + i];
#pragma unroll
            for (int k = 0; k < CK; k++) {
#pragma unroll
                for (int i = 0; i < D; i++) if (lane == i) tab[HP + k * D + i] = (T)h[i];
                double hn[D];
#pragma unroll
                for (int j = 0; j < D; j++) { double t = 0; for (int i = 0; i < D; i++) t = fma(h[i], ak[i * D + j], t); hn[j] = t; }
#pragma unroll
                for (int j = 0; j < D; j++) h[j] = hn[j];
            }
        }
        // pj = M^(r+1), r = lane & 15, by binary powering on the bits of r
        double acc[D * D], sq[D * D];
#pragma unroll
        for (int i = 0; i < D * D; i++) { acc[i] = m[i]; sq[i] = m[i]; }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            double t[D * D];
            matmul<double, D>(acc, sq, t);
            const bool take = ((lane & 15) >> b) & 1;
#pragma unroll
            for (int i = 0; i < D * D; i++) acc[i] = take ? t[i] : acc[i];
            matmul<double, D>(sq, sq, sq);
        }
#pragma unroll
        for (int i = 0; i < D * D; i++) pj[i] = (T)acc[i];
        wave_lds_fence();
        // unstable latent (rho(AKHA) > 1): the largest power (lane 15: M^16) has left the range where the scan is trustworthy
        // in this precision -> leave the latent to the sequential kernel
        const double lim = sizeof(T) == 4 ? 1e18 : 1e150;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < D * D; i++) bad |= !(fabs(acc[i]) < lim);
        if (__any(bad)) {
            if (lane == 0) fallback[l] = 2;
            return;
        }
    }
    static_assert((CK & (CK - 1)) == 0, "CK must be a power of two");

    GradConst<T, D> c;
    load_grad_const<T, D>(c, cb);

    const T* row = Ty + l * ld;
    T* orow = WRITE ? yhat + l * ld : nullptr;
    T xin[D], dxin[P][D];
#pragma unroll
    for (int i = 0; i < D; i++) xin[i] = x[l * D + i];
#pragma unroll
    for (int p = 0; p < P; p++)
#pragma unroll
        for (int i = 0; i < D; i++) dxin[p][i] = dx[(l * P + p) * D + i];

    double sv2 = 0.0, svdv[P] = {0.0, 0.0, 0.0};     // per-lane sums over ticks
    unsigned nobs = 0;
    bool has_nan = false;

    const size_t nfull = Tlen / SEG, nseg = (Tlen + SEG - 1) / SEG;
    // one segment; the ragged last one is its own instantiation so that full segments carry no masking
    auto segment = [&](const size_t seg, auto tail_c) {
        constexpr bool tail = decltype(tail_c)::value;
        const size_t tbase = seg * SEG, t0 = tbase + (size_t)lane * CK;
        // ---- coalesced loads -> LDS -> chunk-per-lane registers (padding past Tlen reads as zero) --------
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int q = i * 64 + lane;
            const size_t tq = tbase + (size_t)q * EPV;
            T e[EPV] = {};
            if (!tail || tq < Tlen) {
                unpack<T>(*reinterpret_cast<const V*>(row + tq), e);
                if (tail) {
#pragma unroll
                    for (int k = 0; k < EPV; k++) if (tq + k >= Tlen) e[k] = T(0);
                }
            }
            lds[q + q / VPL] = pack<T>(e);
        }
        wave_lds_fence();
        // Full segments hold the chunk in registers and unroll the tick loops.  The ragged last segment runs once per latent
        // but would set the kernel's register allocation (masks keep old and new values alive): it keeps its chunk in LDS and
        // walks it with rolled loops instead.
        T y[CK];
        T* yl = reinterpret_cast<T*>(lds + lane * (VPL + 1));
        if constexpr (!tail) {
#pragma unroll
            for (int k = 0; k < VPL; k++) unpack<T>(lds[lane * (VPL + 1) + k], &y[k * EPV]);
        }

        // ---- (a) mean: chunk response + scan -------------------------------------------------------------
#if MOIHGP_GRAD_SPREG
        T sp[4 * D * D];                     // scan powers of this segment in registers
#pragma unroll
        for (int i = 0; i < 4 * D * D; i++) sp[i] = tab[CK * D + i];
#else
        const T* sp = tab + CK * D;          // scan powers stay in LDS: broadcast reads at every use
#endif
        T z[D];
#pragma unroll
        for (int i = 0; i < D; i++) z[i] = T(0);
        if constexpr (tail) {
#pragma unroll 1
            for (int k = 0; k < CK; k++) {
                const T yk = yl[k];
#pragma unroll
                for (int i = 0; i < D; i++) z[i] = fma(tab[k * D + i], yk, z[i]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < CK; k++)
#pragma unroll
                for (int i = 0; i < D; i++) z[i] = fma(tab[k * D + i], y[k], z[i]);
        }
        bool bad = false;
#pragma unroll
        for (int i = 0; i < D; i++) bad |= (z[i] != z[i]);
        if (__any(bad)) { has_nan = true; return; }                // missing ticks: leave the latent to grad_gen_kernel (second pass)
        {
            T x0[D];
#pragma unroll
            for (int i = 0; i < D; i++) x0[i] = (lane == 0) ? xin[i] : T(0);
            matvec_acc<T, D>(sp, x0, z);
        }
        dpp_scan<T, D>(z, sp, pj);
        T xs[D];
#pragma unroll
        for (int i = 0; i < D; i++) xs[i] = wave_shr1(z[i], xin[i]);

        // ---- (b) the one replay: true mean trajectory, zero-state response dz_p of every sensitivity over the chunk, and the
        //      gradient sums in split form.  With dx_p(k) = dz_p(k) + AKHA^k dx_p(start) (linearity) the tick term is
        //      v dv_p = v (-HdA_p x - HA dz_p) - v (HA AKHA^k) dx_p(start): the first part is summed here, the second as the row
        //      vector w = sum_k v_k hp_k, contracted with the start state once the scan below has produced it.
        T pv2 = 0, pvdv[P];
        int jl = 63, nl = CK;                                       // lane and tick count of the last chunk with data
        if (tail) { jl = (int)((Tlen - 1 - tbase) / CK); nl = (int)(Tlen - tbase - (size_t)jl * CK); }
        if constexpr (sizeof(T) == 4) {
            using T2 = typename PairOf<T>::type;
            T dz1[D], w[D];
            T2 dz02[D], dk02[D];
            T pv1 = 0;
            T2 pv02 = {T(0), T(0)};
#pragma unroll
            for (int i = 0; i < D; i++) { dk02[i].x = c.dk[0][i]; dk02[i].y = c.dk[2][i]; }
            {
#pragma unroll
                for (int i = 0; i < D; i++) { dz1[i] = T(0); dz02[i] = T2{T(0), T(0)}; w[i] = T(0); }
                auto tick = [&](const int k, const T yk) -> T {
                    const bool valid = !tail || (t0 + k) < Tlen;
                    T xo[D], hx;
#pragma unroll
                    for (int i = 0; i < D; i++) xo[i] = xs[i];
                    const T vr = tick_mean<T, D>(c, xs, yk, hx);        // xs <- AKHA x + K y (ihgp.h:50), v = y - HA x
                    T v = vr;
                    if (tail) {
                        v = valid ? v : T(0);
                        nobs += valid ? 1u : 0u;
#pragma unroll
                        for (int i = 0; i < D; i++) xs[i] = valid ? xs[i] : xo[i];
                    }
                    // lengthscale (dF != 0): -dv without the start-state part (ihgp.h:218), u = dAKHA x + dK y + AKHA dz (ihgp.h:54)
                    T a1 = 0, u1[D];
#pragma unroll
                    for (int i = 0; i < D; i++) a1 = fma(c.a[i], dz1[i], a1);
#pragma unroll
                    for (int i = 0; i < D; i++) a1 = fma(c.hda[1][i], xo[i], a1);
#pragma unroll
                    for (int i = 0; i < D; i++) u1[i] = c.dk[1][i] * yk;
                    matvec_acc<T, D>(c.dakha[1], xo, u1);
                    matvec_acc<T, D>(c.akha, dz1, u1);
                    // magnitude and noise as a pair: dAKHA_p x + dK_p y = dK_p v
                    T2 a02 = {T(0), T(0)}, u02[D];
#pragma unroll
                    for (int i = 0; i < D; i++) a02 = c.a[i] * dz02[i] + a02;
#pragma unroll
                    for (int i = 0; i < D; i++) u02[i] = dk02[i] * vr;
                    matvec_acc2<T, D>(c.akha, dz02, u02);
#pragma unroll
                    for (int i = 0; i < D; i++) { dz1[i] = valid ? u1[i] : dz1[i]; dz02[i] = valid ? u02[i] : dz02[i]; }
                    pv2 = fma(v, v, pv2);
                    pv1 = fma(-v, a1, pv1);
                    pv02 = (-v) * a02 + pv02;
#pragma unroll
                    for (int i = 0; i < D; i++) w[i] = fma(v, tab[HP + k * D + i], w[i]);
                    return (WRITE == 2) ? hx : xs[0];
                };
                if constexpr (tail) {
#pragma unroll 1
                    for (int k = 0; k < CK; k++) {
                        const T o = tick(k, yl[k]);
                        if (WRITE) yl[k] = o;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < CK; k++) y[k] = tick(k, y[k]);
                }
            }
            if (!tail) nobs += CK;
            // ---- (c) scan the sensitivities with the same powers; lane 0 carries dx_in; close the gradient sums --------
#pragma unroll
            for (int i = 0; i < D; i++) xin[i] = read_lane(xs[i], jl);  // state after the last valid tick
            {   // the lengthscale
                T d0[D], dzl[D], ds[D];
#pragma unroll
                for (int i = 0; i < D; i++) { d0[i] = (lane == 0) ? dxin[1][i] : T(0); dzl[i] = dz1[i]; }
                matvec_acc<T, D>(sp, d0, dz1);
                dpp_scan<T, D>(dz1, sp, pj);
#pragma unroll
                for (int i = 0; i < D; i++) ds[i] = wave_shr1(dz1[i], dxin[1][i]);         // dx_p at the start of this lane's chunk
                T sw = 0;
#pragma unroll
                for (int i = 0; i < D; i++) sw = fma(w[i], ds[i], sw);
                pvdv[1] = pv1 - sw;
                if (!tail) {
#pragma unroll
                    for (int i = 0; i < D; i++) dxin[1][i] = read_lane(dz1[i], 63);        // inclusive scan value = end of the segment
                } else {
                    // the last chunk holds nl <= CK ticks: its end state is dz (frozen there) + AKHA^nl dx(start)
                    for (int k = 0; k < nl; k++) {
                        T t[D];
#pragma unroll
                        for (int i = 0; i < D; i++) t[i] = T(0);
                        matvec_acc<T, D>(c.akha, ds, t);
#pragma unroll
                        for (int i = 0; i < D; i++) ds[i] = t[i];
                    }
#pragma unroll
                    for (int i = 0; i < D; i++) dxin[1][i] = read_lane(dzl[i] + ds[i], jl);
                }
            }
            {   // magnitude and noise
                T2 d0[D], dzl[D], ds[D];
#pragma unroll
                for (int i = 0; i < D; i++) {
                    d0[i].x = (lane == 0) ? dxin[0][i] : T(0); d0[i].y = (lane == 0) ? dxin[2][i] : T(0);
                    dzl[i] = dz02[i];
                }
                matvec_acc2<T, D>(sp, d0, dz02);
                dpp_scan2<T, D>(dz02, sp, pj);
#pragma unroll
                for (int i = 0; i < D; i++) { ds[i].x = wave_shr1(dz02[i].x, dxin[0][i]); ds[i].y = wave_shr1(dz02[i].y, dxin[2][i]); }
                T2 sw = {T(0), T(0)};
#pragma unroll
                for (int i = 0; i < D; i++) sw = w[i] * ds[i] + sw;
                pvdv[0] = pv02.x - sw.x; pvdv[2] = pv02.y - sw.y;
                if (!tail) {
#pragma unroll
                    for (int i = 0; i < D; i++) { dxin[0][i] = read_lane(dz02[i].x, 63); dxin[2][i] = read_lane(dz02[i].y, 63); }
                } else {
                    for (int k = 0; k < nl; k++) {
                        T2 t[D];
#pragma unroll
                        for (int i = 0; i < D; i++) t[i] = T2{T(0), T(0)};
                        matvec_acc2<T, D>(c.akha, ds, t);
#pragma unroll
                        for (int i = 0; i < D; i++) ds[i] = t[i];
                    }
#pragma unroll
                    for (int i = 0; i < D; i++) { dxin[0][i] = read_lane(dzl[i].x + ds[i].x, jl); dxin[2][i] = read_lane(dzl[i].y + ds[i].y, jl); }
                }
            }
        } else {
            // fp64: no packed form; one parameter after the other keeps fewer values alive (240 against 284 VGPRs)
            T dz[P][D], w[D];
#pragma unroll
            for (int p = 0; p < P; p++) pvdv[p] = T(0);
            {
#pragma unroll
                for (int p = 0; p < P; p++)
#pragma unroll
                    for (int i = 0; i < D; i++) dz[p][i] = T(0);
#pragma unroll
                for (int i = 0; i < D; i++) w[i] = T(0);
                auto tick = [&](const int k, const T yk) -> T {
                    const bool valid = !tail || (t0 + k) < Tlen;
                    T xo[D], hx;
#pragma unroll
                    for (int i = 0; i < D; i++) xo[i] = xs[i];
                    const T vr = tick_mean<T, D>(c, xs, yk, hx);      // xs <- AKHA x + K y (ihgp.h:50), v = y - HA x
                    T v = vr;
                    if (tail) {
                        v = valid ? v : T(0);
                        nobs += valid ? 1u : 0u;
#pragma unroll
                        for (int i = 0; i < D; i++) xs[i] = valid ? xs[i] : xo[i];
                    }
                    T dv[P];
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        T a = 0, u[D];
#pragma unroll
                        for (int i = 0; i < D; i++) a = fma(c.a[i], dz[p][i], a);
                        if (kDFzero[p]) {
#pragma unroll
                            for (int i = 0; i < D; i++) u[i] = c.dk[p][i] * vr;               // dAKHA_p x + dK_p y = dK_p v
                        } else {
#pragma unroll
                            for (int i = 0; i < D; i++) a = fma(c.hda[p][i], xo[i], a);
#pragma unroll
                            for (int i = 0; i < D; i++) u[i] = c.dk[p][i] * yk;
                            matvec_acc<T, D>(c.dakha[p], xo, u);        // u = dAKHA_p x + dK_p y   (pre-step x, ihgp.h:54)
                        }
                        dv[p] = a;                                      // -dv_p without the start-state part (ihgp.h:218)
                        matvec_acc<T, D>(c.akha, dz[p], u);             // + AKHA dz
#pragma unroll
                        for (int i = 0; i < D; i++) dz[p][i] = valid ? u[i] : dz[p][i];
                    }
                    pv2 = fma(v, v, pv2);
#pragma unroll
                    for (int p = 0; p < P; p++) pvdv[p] = fma(-v, dv[p], pvdv[p]);
#pragma unroll
                    for (int i = 0; i < D; i++) w[i] = fma(v, tab[HP + k * D + i], w[i]);
                    return (WRITE == 2) ? hx : xs[0];
                };
                if constexpr (tail) {
#pragma unroll 1
                    for (int k = 0; k < CK; k++) {
                        const T o = tick(k, yl[k]);
                        if (WRITE) yl[k] = o;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < CK; k++) y[k] = tick(k, y[k]);
                }
            }
            if (!tail) nobs += CK;
            // ---- (c) scan the sensitivities with the same powers; lane 0 carries dx_in; close the gradient sums --------
#pragma unroll
            for (int i = 0; i < D; i++) xin[i] = read_lane(xs[i], jl);  // state after the last valid tick
#pragma unroll
            for (int p = 0; p < P; p++) {
                T d0[D], dzl[D], ds[D];
#pragma unroll
                for (int i = 0; i < D; i++) { d0[i] = (lane == 0) ? dxin[p][i] : T(0); dzl[i] = dz[p][i]; }
                matvec_acc<T, D>(sp, d0, dz[p]);
                dpp_scan<T, D>(dz[p], sp, pj);
#pragma unroll
                for (int i = 0; i < D; i++) ds[i] = wave_shr1(dz[p][i], dxin[p][i]);       // dx_p at the start of this lane's chunk
                T s = 0;
#pragma unroll
                for (int i = 0; i < D; i++) s = fma(w[i], ds[i], s);
                pvdv[p] -= s;
                if (!tail) {
#pragma unroll
                    for (int i = 0; i < D; i++) dxin[p][i] = read_lane(dz[p][i], 63);      // inclusive scan value = end of the segment
                } else {
                    // the last chunk holds nl <= CK ticks: its end state is dz (frozen there) + AKHA^nl dx(start)
                    for (int k = 0; k < nl; k++) {
                        T t[D];
#pragma unroll
                        for (int i = 0; i < D; i++) t[i] = T(0);
                        matvec_acc<T, D>(c.akha, ds, t);
#pragma unroll
                        for (int i = 0; i < D; i++) ds[i] = t[i];
                    }
#pragma unroll
                    for (int i = 0; i < D; i++) dxin[p][i] = read_lane(dzl[i] + ds[i], jl);
                }
            }
        }
        sv2 += (double)pv2;
#pragma unroll
        for (int p = 0; p < P; p++) svdv[p] += (double)pvdv[p];

        if (WRITE) {
            if constexpr (!tail) {
#pragma unroll
                for (int k = 0; k < VPL; k++) lds[lane * (VPL + 1) + k] = pack<T>(&y[k * EPV]);
            }
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                const int q = i * 64 + lane;
                const size_t tq = tbase + (size_t)q * EPV;
                if (!tail || tq < Tlen) nt_store(lds[q + q / VPL], reinterpret_cast<V*>(orow + tq));
            }
        }
        wave_lds_fence();
    };
    for (size_t seg = 0; seg < nfull && !has_nan; seg++) segment(seg, std::false_type{});
    if (nseg > nfull && !has_nan) segment(nfull, std::true_type{});

    if (has_nan) {                                                  // nothing has been written for this latent's state yet
        if (lane == 0) {                                            // flag + a place in the compact list grad_gen_kernel works from
            fallback[l] = 1;
            int* list = fallback + L;
            list[atomicAdd(list + L, 1)] = (int)l;
        }
        return;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sv2 += __shfl_xor(sv2, o);
        nobs += __shfl_xor(nobs, o);
#pragma unroll
        for (int p = 0; p < P; p++) svdv[p] += __shfl_xor(svdv[p], o);
    }
    if (lane == 0) {
        fallback[l] = 0;
#pragma unroll
        for (int i = 0; i < D; i++) x[l * D + i] = xin[i];
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int i = 0; i < D; i++) dx[(l * P + p) * D + i] = dxin[p][i];
        const double S = c64[Lay::S], n = (double)nobs;
        if (nll) nll[l] = 0.5 * (sv2 / S + n * c64[Lay::LOGS]);
#pragma unroll
        for (int p = 0; p < P; p++) grad[l * P + p] = svdv[p] / S - 0.5 * (sv2 / S - n) * c64[Lay::DS + p] / S;
    }
}

// ---------------------------------------------------------------------------------------------
// One lane per latent, sequential in time (ihgp.h:37-57, :212-222 literally, including the missing-data branch).
// Runs only the latents flagged in `only` (NULL: all).  y is fetched one 16-byte vector at a time.
template <typename T, int D>
__global__ void __launch_bounds__(64)
grad_seq_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT,
                const double* __restrict__ cb64, T* __restrict__ x, T* __restrict__ dx, T* __restrict__ yhat,
                double* __restrict__ nll, double* __restrict__ grad, const int* __restrict__ only, int write_mode) {
    using V = typename VecOf<T>::type;
    using Lay = CB<D>;
    constexpr int EPV = 16 / sizeof(T);
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    if (only && !only[l]) return;
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
    for (size_t tb = 0; tb < Tlen; tb += EPV) {
        T yv[EPV];
        unpack<T>(*reinterpret_cast<const V*>(row + tb), yv);       // ld >= roundup(T, EPV)
        for (int e = 0; e < EPV && tb + e < Tlen; e++) {
            const T y = yv[e];
            const bool miss = (y != y);
            T xn[D], dxn[P][D];
            T hx = 0;
            for (int i = 0; i < D; i++) hx = fma(ha[i], xs[i], hx);
            if (!miss) {
                double v = (double)(y - hx);
                acc += 0.5 * (v * v / S + logS);                        // ihgp.h:215
                for (int p = 0; p < P; p++) {
                    T a = 0, b = 0;
                    for (int i = 0; i < D; i++) { a = fma(hda[p][i], xs[i], a); b = fma(ha[i], dxs[p][i], b); }
                    double dv = (double)(-a - b);                       // ihgp.h:218
                    g[p] += (v * dv - 0.5 * (v * v / S - 1.0) * dS[p]) / S;   // ihgp.h:219
                }
                for (int i = 0; i < D; i++) xn[i] = kk[i] * y;
                matvec_acc<T, D>(akha, xs, xn);                         // ihgp.h:50
                for (int p = 0; p < P; p++) {
                    for (int i = 0; i < D; i++) dxn[p][i] = dk[p][i] * y;
                    matvec_acc<T, D>(dakha[p], xs, dxn[p]);
                    matvec_acc<T, D>(akha, dxs[p], dxn[p]);             // ihgp.h:54
                }
            } else {
                for (int i = 0; i < D; i++) xn[i] = 0;
                matvec_acc<T, D>(aa, xs, xn);                           // ihgp.h:41
                for (int p = 0; p < P; p++) {
                    for (int i = 0; i < D; i++) dxn[p][i] = 0;
                    matvec_acc<T, D>(da[p], xs, dxn[p]);
                    matvec_acc<T, D>(aa, dxs[p], dxn[p]);               // ihgp.h:45
                }
            }
            for (int i = 0; i < D; i++) xs[i] = xn[i];
            for (int p = 0; p < P; p++)
                for (int i = 0; i < D; i++) dxs[p][i] = dxn[p][i];
            if (yhat) yhat[l * ld + tb + e] = (write_mode == 2) ? hx : xn[0];
        }
    }
    for (int i = 0; i < D; i++) x[l * D + i] = xs[i];
    for (int p = 0; p < P; p++)
        for (int i = 0; i < D; i++) dx[(l * P + p) * D + i] = dxs[p][i];
    if (nll) nll[l] = acc;
    for (int p = 0; p < P; p++) grad[l * P + p] = g[p];
}

template <typename T, int D, int CK>
int launch_grad_t(const T* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, T* x, T* dx, T* yhat,
                  double* nll, double* grad, int* fallback, int out_mode, hipStream_t stream) {
    dim3 block(64 * kWavesPerBlock), grid((unsigned)((L + kWavesPerBlock - 1) / kWavesPerBlock));
    (void)hipMemsetAsync(fallback + 2 * L, 0, sizeof(int), stream);     // number of latents left for the second pass
    if (yhat && out_mode == 2)
        hipLaunchKernelGGL((grad_scan_kernel<T, D, CK, 2>), grid, block, 0, stream, Ty, Tlen, ld, L, cbT, cb64, x, dx, yhat, nll, grad, fallback);
    else if (yhat)
        hipLaunchKernelGGL((grad_scan_kernel<T, D, CK, 1>), grid, block, 0, stream, Ty, Tlen, ld, L, cbT, cb64, x, dx, yhat, nll, grad, fallback);
    else
        hipLaunchKernelGGL((grad_scan_kernel<T, D, CK, 0>), grid, block, 0, stream, Ty, Tlen, ld, L, cbT, cb64, x, dx, yhat, nll, grad, fallback);
    // latents with missing ticks (flag 1): scan over the chunks' affine maps (grad_gen.hip); it clears the flags it has dealt with
    if (int rc = launch_grad_gen(D, sizeof(T) == 8 ? 0 : 1, Ty, Tlen, ld, L, cb64, sizeof(T) == 8 ? nullptr : reinterpret_cast<const float*>(cbT),
                                 x, dx, yhat, nll, grad, fallback, stream, out_mode)) return rc;
    // what is still flagged (unstable latents) is redone sequentially (x / dx were left untouched)
    hipLaunchKernelGGL((grad_seq_kernel<T, D>), dim3((unsigned)((L + 63) / 64)), dim3(64), 0, stream, Ty, Tlen, ld, L, cbT, cb64, x, dx, yhat, nll,
                       grad, (const int*)fallback, out_mode);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("grad kernels launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

// fallback: device int[2 L + 1] scratch: flags of the latents left to the later passes [L], the compact list of those with missing ticks
// [L] and its length
int launch_grad_stream(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64,
                       const float* cb32, void* x, void* dx, void* yhat, double* nll, double* grad, int* fallback,
                       hipStream_t stream, int out_mode) {
    if (L == 0) return 0;
    // short windows (the online learner's W <= 128 ticks, moihgp_online.h:61-70) use 16-byte chunks so that a
    // window still spreads over the lanes of the wave; long streams use 16-tick (fp32) / 8-tick (fp64) chunks (tuning hook: MOIHGP_GRAD_CK)
    const bool shortw = T <= (dtype == 0 ? 256 : 512);
    static const int ck_override = [] { const char* e = std::getenv("MOIHGP_GRAD_CK"); return e ? std::atoi(e) : 0; }();
#define MOIHGP_GRAD_CASE(TT, DD, CKK, CB) \
    return launch_grad_t<TT, DD, CKK>((const TT*)Ty, T, ld, L, CB, cb64, (TT*)x, (TT*)dx, (TT*)yhat, nll, grad, fallback, out_mode, stream)
    if (dtype == 0) {
        if (d == 2) { if (shortw) MOIHGP_GRAD_CASE(double, 2, 2, cb64); MOIHGP_GRAD_CASE(double, 2, 8, cb64); }
        if (shortw || ck_override == 2) MOIHGP_GRAD_CASE(double, 3, 2, cb64);
        if (ck_override == 4) MOIHGP_GRAD_CASE(double, 3, 4, cb64);
        MOIHGP_GRAD_CASE(double, 3, 8, cb64);
    }
    if (d == 2) { if (shortw) MOIHGP_GRAD_CASE(float, 2, 4, cb32); MOIHGP_GRAD_CASE(float, 2, 16, cb32); }
    if (shortw || ck_override == 4) MOIHGP_GRAD_CASE(float, 3, 4, cb32);
    if (ck_override == 8) MOIHGP_GRAD_CASE(float, 3, 8, cb32);
    MOIHGP_GRAD_CASE(float, 3, 16, cb32);           // measured: 16-tick chunks (236 VGPRs) 0.178 ms, 8-tick chunks (172) 0.228 ms at 4096 x 1e4
#undef MOIHGP_GRAD_CASE
}

}  // namespace moihgp
