// grad_gen.hip -- the sensitivity / gradient sweep (ihgp.h:37-57 with :212-222, A2 + A5) for streams WITH MISSING TICKS, parallel
// in time.  Second pass behind grad_scan_kernel (grad.hip), which handles the latents whose stream has no gap and flags the others.
//
// A missing tick (NaN y, ihgp.h:39-47) advances x <- A x, dx_p <- dA_p x + A dx_p: the chunk maps are no longer powers of one
// matrix, so the segment solve of grad.hip (chunk response from a table, scan with the uniform powers of AKHA^CK) does not
// apply.  What still holds: over a chunk the augmented state s = (x, dx_1 .. dx_P) moves by an AFFINE map
//         s_end = Phi_j s_start + zeta_j,      Phi_j = [[M, 0], [N_p, M]]   (block lower triangular: dx never feeds x)
// whatever the pattern of gaps.  So, as recursion.hip's generic_segment does for the filter:
//   pass 1   every lane walks its chunk once with FOUR vectors: the response from a zero state (driven by y) -> zeta_j = (z, dz_p),
//            and the D unit start states (no input) -> the columns of M and N_p;
//   scan     of the pairs (Phi_j, zeta_j) over the 64 lanes: in-row Kogge-Stone with row_shr (lanes without a source compose with
//            the identity), then three row_bcast:15 rounds in which a lane folds the finished prefix of the previous row through
//            its own in-row Phi;
//   pass 2   replay of the chunk from its true start state: NLL / gradient sums (ihgp.h:215-219 on the pre-step state), outputs.
// One tick of any of these vectors is the INNOVATION form (w = 1 observed, 0 missing; v = w (y - HA x)):
//         dv_p = -w ((H dA_p) x + HA dx_p)        x' = A x + K v        dx_p' = A dx_p + dA_p x + dK_p v + K dv_p
// (== ihgp.h:50,54 for w = 1 and :41,45 for w = 0), which needs only A, K, HA, dK_p and the dA_p that are not zero: for the
// reference's models that is the lengthscale alone (matern32ss.h:54-55, matern52ss.h:61-63), so 36 wave-uniform scalars in all.
// A latent with more than one non-zero dA_p (no such model exists) is left flagged for grad_seq_kernel.
//
// Cost: about 4 x 81 multiply-adds per tick in pass 1, 81 in pass 2, ~2300 per lane and segment in the scan: ~5 x the arithmetic of
// the gap-free path, all 64 lanes busy -- against the tick-by-tick walk of a segment on 16 lanes that round 2's first version used.
#include "kernels_common.h"

namespace moihgp {
namespace {

constexpr int P = kNumIgpParam;

// wave-uniform constants of one latent (scalar registers)
template <typename T, int D>
struct GenConst {
    T a[D * D], k[D], ha[D], dk[P][D], dal[D * D], hdal[D];
    int pl;                                // the parameter with dA_p != 0 (-1: none)
};

// One tick of one vector (x, dx_p) in innovation form.  yin: observation (0 for the homogeneous vectors); w: tick observed.
// v / dv are returned for the sums of pass 2.
template <typename T, int D>
__device__ inline void gen_tick(const GenConst<T, D>& c, T (&x)[D], T (&dx)[P][D], T yin, bool w, T& v_out, T (&dv_out)[P], T& hx_out) {
    T hx = 0;
#pragma unroll
    for (int j = 0; j < D; j++) hx = fma(c.ha[j], x[j], hx);
    const T v = w ? yin - hx : T(0);
    T dv[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        T s = 0;
#pragma unroll
        for (int j = 0; j < D; j++) s = fma(c.ha[j], dx[p][j], s);
        if (p == c.pl) {
#pragma unroll
            for (int j = 0; j < D; j++) s = fma(c.hdal[j], x[j], s);
        }
        dv[p] = w ? -s : T(0);
    }
    T xn[D], dxn[P][D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        T s = c.k[i] * v;
#pragma unroll
        for (int j = 0; j < D; j++) s = fma(c.a[i * D + j], x[j], s);
        xn[i] = s;
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
#pragma unroll
        for (int i = 0; i < D; i++) {
            T s = c.dk[p][i] * v;
            s = fma(c.k[i], dv[p], s);
#pragma unroll
            for (int j = 0; j < D; j++) s = fma(c.a[i * D + j], dx[p][j], s);
            dxn[p][i] = s;
        }
        if (p == c.pl) {
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = 0; j < D; j++) dxn[p][i] = fma(c.dal[i * D + j], x[j], dxn[p][i]);
        }
    }
#pragma unroll
    for (int i = 0; i < D; i++) x[i] = xn[i];
#pragma unroll
    for (int p = 0; p < P; p++)
#pragma unroll
        for (int i = 0; i < D; i++) dx[p][i] = dxn[p][i];
    v_out = v; hx_out = hx;
#pragma unroll
    for (int p = 0; p < P; p++) dv_out[p] = dv[p];
}

// Affine map of a chunk: M (row-major), N_p, z, dz_p.
template <typename T, int D>
struct GenMap {
    T m[D * D], n[P][D * D], z[D], dz[P][D];
};

// cur <- cur o prev, prev taken from the lane O places down the row (identity where there is none)
template <int O, typename T, int D>
__device__ inline void gen_scan_level(GenMap<T, D>& g) {
    T z1[D], m1[D * D];
#pragma unroll
    for (int i = 0; i < D; i++) z1[i] = dpp0<DPP_ROW_SHR + O, 0xF>(g.z[i]);
#pragma unroll
    for (int i = 0; i < D * D; i++) m1[i] = dpp_fill<DPP_ROW_SHR + O, 0xF>((i % (D + 1) == 0) ? T(1) : T(0), g.m[i]);
    // dz_p = N2p z1 + M2 dz1p + dz2p ;  N_p = N2p M1 + M2 N1p   (own M2 still unchanged)
#pragma unroll
    for (int p = 0; p < P; p++) {
        T dz1[D], n1[D * D];
#pragma unroll
        for (int i = 0; i < D; i++) dz1[i] = dpp0<DPP_ROW_SHR + O, 0xF>(g.dz[p][i]);
#pragma unroll
        for (int i = 0; i < D * D; i++) n1[i] = dpp0<DPP_ROW_SHR + O, 0xF>(g.n[p][i]);
        matvec_acc<T, D>(g.n[p], z1, g.dz[p]);
        matvec_acc<T, D>(g.m, dz1, g.dz[p]);
        T t[D * D], u[D * D];
        matmul<T, D>(g.n[p], m1, t);
        matmul<T, D>(g.m, n1, u);
#pragma unroll
        for (int i = 0; i < D * D; i++) g.n[p][i] = t[i] + u[i];
    }
    matvec_acc<T, D>(g.m, z1, g.z);          // z = M2 z1 + z2
    matmul<T, D>(g.m, m1, g.m);              // M = M2 M1
}

// row_bcast:15 hands lane 15 of row r - 1 to every lane of row r; MASK selects the one row that takes part in this round (the others
// read zeros and add nothing): zeta <- Phi_inrow zeta_prev + zeta
template <int MASK, typename T, int D>
__device__ inline void gen_cross_round(GenMap<T, D>& g) {
    T zp[D];
#pragma unroll
    for (int i = 0; i < D; i++) zp[i] = dpp0<DPP_ROW_BCAST15, MASK>(g.z[i]);
#pragma unroll
    for (int p = 0; p < P; p++) {
        T dzp[D];
#pragma unroll
        for (int i = 0; i < D; i++) dzp[i] = dpp0<DPP_ROW_BCAST15, MASK>(g.dz[p][i]);
        matvec_acc<T, D>(g.n[p], zp, g.dz[p]);
        matvec_acc<T, D>(g.m, dzp, g.dz[p]);
    }
    matvec_acc<T, D>(g.m, zp, g.z);
}

// WRITE: 0 no stream output, 1 filtered means (ihgp.h:51), 2 predicted means HA x_t
template <typename T, int D, int CK, int WRITE>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
grad_gen_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64,
                T* __restrict__ x, T* __restrict__ dx, T* __restrict__ yhat, double* __restrict__ nll, double* __restrict__ grad,
                int* __restrict__ fallback) {
    using V = typename VecOf<T>::type;
    using Lay = CB<D>;
    constexpr int EPV = 16 / sizeof(T), VPL = CK / EPV, SEG = 64 * CK, NVP = 64 * (VPL + 1);
    static_assert(CK % EPV == 0, "CK must be a multiple of 16 bytes");
    __shared__ V lds_all[kWavesPerBlock][NVP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // only the latents the first pass left for missing ticks, from its compact list (so that the wavefronts with work fill whole
    // workgroups and the rest of the grid exits at once: the sweep's time follows the number of such latents)
    const size_t widx = (size_t)blockIdx.x * kWavesPerBlock + wave;
    const int* list = fallback + L;
    if (widx >= L || (int)widx >= list[L]) return;
    const size_t l = (size_t)list[widx];
    V* lds = lds_all[wave];
    const T* cb = cbT + l * Lay::SIZE;
    const double* c64 = cb64 + l * Lay::SIZE;

    GenConst<T, D> c;
    {
        int pl = -1, nnz = 0;
#pragma unroll
        for (int p = 0; p < P; p++) {
            bool nz = false;
#pragma unroll
            for (int i = 0; i < D * D; i++) nz |= (cb[Lay::DA + p * D * D + i] != T(0));
#pragma unroll
            for (int i = 0; i < D; i++) nz |= (cb[Lay::HDA + p * D + i] != T(0));
            if (nz) { pl = p; nnz++; }
        }
        if (nnz > 1) return;                                        // (stays flagged: grad_seq_kernel)
        c.pl = pl;
        const int ps = pl < 0 ? 0 : pl;
#pragma unroll
        for (int i = 0; i < D * D; i++) { c.a[i] = cb[Lay::A + i]; c.dal[i] = pl < 0 ? T(0) : cb[Lay::DA + ps * D * D + i]; }
#pragma unroll
        for (int i = 0; i < D; i++) { c.k[i] = cb[Lay::K + i]; c.ha[i] = cb[Lay::HA + i]; c.hdal[i] = pl < 0 ? T(0) : cb[Lay::HDA + ps * D + i]; }
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int i = 0; i < D; i++) c.dk[p][i] = cb[Lay::DK + p * D + i];
    }

    const T* row = Ty + l * ld;
    T* orow = WRITE ? yhat + l * ld : nullptr;
    T xin[D], dxin[P][D];                                           // wave-uniform: state before the segment
#pragma unroll
    for (int i = 0; i < D; i++) xin[i] = x[l * D + i];
#pragma unroll
    for (int p = 0; p < P; p++)
#pragma unroll
        for (int i = 0; i < D; i++) dxin[p][i] = dx[(l * P + p) * D + i];
    double sv2 = 0.0, svdv[P] = {0.0, 0.0, 0.0};                    // per-lane sums over ticks
    unsigned nobs = 0;

    const size_t nseg = (Tlen + SEG - 1) / SEG;
    for (size_t seg = 0; seg < nseg; seg++) {
        const size_t tbase = seg * SEG, t0 = tbase + (size_t)lane * CK;
        // ---- coalesced loads -> LDS tile, chunk per lane row (zero past the end; those ticks are handled as missing below) ----
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int q = i * 64 + lane;
            const size_t tq = tbase + (size_t)q * EPV;
            T e[EPV] = {};
            if (tq < Tlen) unpack<T>(*reinterpret_cast<const V*>(row + tq), e);
            lds[q + q / VPL] = pack<T>(e);
        }
        wave_lds_fence();
        T* yl = reinterpret_cast<T*>(lds + lane * (VPL + 1));

        // ---- pass 1: the chunk's affine map ----
        GenMap<T, D> g;
        {
            T bx[D][D], bdx[D][P][D];                               // the D unit start states
#pragma unroll
            for (int cidx = 0; cidx < D; cidx++) {
#pragma unroll
                for (int i = 0; i < D; i++) bx[cidx][i] = (i == cidx) ? T(1) : T(0);
#pragma unroll
                for (int p = 0; p < P; p++)
#pragma unroll
                    for (int i = 0; i < D; i++) bdx[cidx][p][i] = T(0);
            }
#pragma unroll
            for (int i = 0; i < D; i++) g.z[i] = T(0);
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int i = 0; i < D; i++) g.dz[p][i] = T(0);
#pragma unroll 2
            for (int k = 0; k < CK; k++) {
                const T yk = yl[k];
                const bool w = !(yk != yk) && (t0 + k) < Tlen;
                T v, dv[P], hx;
                gen_tick<T, D>(c, g.z, g.dz, yk, w, v, dv, hx);
#pragma unroll
                for (int cidx = 0; cidx < D; cidx++) gen_tick<T, D>(c, bx[cidx], bdx[cidx], T(0), w, v, dv, hx);
            }
#pragma unroll
            for (int cidx = 0; cidx < D; cidx++)
#pragma unroll
                for (int i = 0; i < D; i++) {
                    g.m[i * D + cidx] = bx[cidx][i];
#pragma unroll
                    for (int p = 0; p < P; p++) g.n[p][i * D + cidx] = bdx[cidx][p][i];
                }
        }
        // lane 0 starts from the carried state: zeta_0 <- Phi_0 s_in + zeta_0
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < P; p++) {
                matvec_acc<T, D>(g.n[p], xin, g.dz[p]);
                matvec_acc<T, D>(g.m, dxin[p], g.dz[p]);
            }
            matvec_acc<T, D>(g.m, xin, g.z);
        }
        // ---- scan of the (Phi_j, zeta_j) pairs ----
        gen_scan_level<1, T, D>(g);
        gen_scan_level<2, T, D>(g);
        gen_scan_level<4, T, D>(g);
        gen_scan_level<8, T, D>(g);
        // cross-row: rows 1, 2, 3 in turn fold the finished prefix of the row before through their own in-row map
        gen_cross_round<0x2, T, D>(g);
        gen_cross_round<0x4, T, D>(g);
        gen_cross_round<0x8, T, D>(g);
        // start state of the lane's chunk = inclusive result of the lane before it (lane 0: the carried state)
        T xs[D], dxs[P][D];
#pragma unroll
        for (int i = 0; i < D; i++) xs[i] = wave_shr1(g.z[i], xin[i]);
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int i = 0; i < D; i++) dxs[p][i] = wave_shr1(g.dz[p][i], dxin[p][i]);

        // ---- pass 2: replay from the true start state ----
        // (sums in fp64 also for fp32 streams: a latent whose trajectory leaves fp32's range for v^2 -- the mildly unstable ones of the
        //  literal DARE that the first pass mistook for streams with gaps -- keeps the finite sums the tick-by-tick kernel gave it)
        double part2 = 0.0, partdv[P] = {0.0, 0.0, 0.0};
#pragma unroll 2
        for (int k = 0; k < CK; k++) {
            const T yk = yl[k];
            const bool valid = (t0 + k) < Tlen;
            const bool w = !(yk != yk) && valid;
            T xo[D], dxo[P][D];
#pragma unroll
            for (int i = 0; i < D; i++) xo[i] = xs[i];
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int i = 0; i < D; i++) dxo[p][i] = dxs[p][i];
            T v, dv[P], hx;
            gen_tick<T, D>(c, xs, dxs, yk, w, v, dv, hx);
            const double vd = (double)v;
            part2 = fma(vd, vd, part2);
#pragma unroll
            for (int p = 0; p < P; p++) partdv[p] = fma(vd, (double)dv[p], partdv[p]);
            nobs += w ? 1u : 0u;
            if (!valid) {                                           // past the end of the stream: the state stays
#pragma unroll
                for (int i = 0; i < D; i++) xs[i] = xo[i];
#pragma unroll
                for (int p = 0; p < P; p++)
#pragma unroll
                    for (int i = 0; i < D; i++) dxs[p][i] = dxo[p][i];
            }
            if (WRITE) yl[k] = (WRITE == 2) ? hx : xs[0];
        }
        sv2 += part2;
#pragma unroll
        for (int p = 0; p < P; p++) svdv[p] += partdv[p];
        // carried state = state of the lane that owns the last valid tick of the segment
        int jl = 63;
        if (tbase + SEG > Tlen) jl = (int)((Tlen - 1 - tbase) / CK);
#pragma unroll
        for (int i = 0; i < D; i++) xin[i] = read_lane(xs[i], jl);
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int i = 0; i < D; i++) dxin[p][i] = read_lane(dxs[p][i], jl);
        if (WRITE) {
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                const int q = i * 64 + lane;
                const size_t tq = tbase + (size_t)q * EPV;
                if (tq < Tlen) nt_store(lds[q + q / VPL], reinterpret_cast<V*>(orow + tq));
            }
        }
        wave_lds_fence();
    }

#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
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

template <typename T, int D, int CK>
int launch_gen_t(const T* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, T* x, T* dx, T* yhat, double* nll, double* grad,
                 int* fallback, int out_mode, hipStream_t stream) {
    dim3 block(64 * kWavesPerBlock), grid((unsigned)((L + kWavesPerBlock - 1) / kWavesPerBlock));
    if (yhat && out_mode == 2)
        hipLaunchKernelGGL((grad_gen_kernel<T, D, CK, 2>), grid, block, 0, stream, Ty, Tlen, ld, L, cbT, cb64, x, dx, yhat, nll, grad, fallback);
    else if (yhat)
        hipLaunchKernelGGL((grad_gen_kernel<T, D, CK, 1>), grid, block, 0, stream, Ty, Tlen, ld, L, cbT, cb64, x, dx, yhat, nll, grad, fallback);
    else
        hipLaunchKernelGGL((grad_gen_kernel<T, D, CK, 0>), grid, block, 0, stream, Ty, Tlen, ld, L, cbT, cb64, x, dx, yhat, nll, grad, fallback);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("grad_gen_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

// Latents flagged 1 in fallback[] (missing ticks; set by grad_scan_kernel) are swept whole; the flag is cleared where that succeeded.
int launch_grad_gen(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const float* cb32, void* x, void* dx,
                    void* yhat, double* nll, double* grad, int* fallback, hipStream_t stream, int out_mode) {
    if (L == 0) return 0;
    if (dtype == 0) {
        if (d == 2) return launch_gen_t<double, 2, 8>((const double*)Ty, T, ld, L, cb64, cb64, (double*)x, (double*)dx, (double*)yhat, nll, grad, fallback, out_mode, stream);
        return launch_gen_t<double, 3, 8>((const double*)Ty, T, ld, L, cb64, cb64, (double*)x, (double*)dx, (double*)yhat, nll, grad, fallback, out_mode, stream);
    }
    if (d == 2) return launch_gen_t<float, 2, 16>((const float*)Ty, T, ld, L, cb32, cb64, (float*)x, (float*)dx, (float*)yhat, nll, grad, fallback, out_mode, stream);
    return launch_gen_t<float, 3, 16>((const float*)Ty, T, ld, L, cb32, cb64, (float*)x, (float*)dx, (float*)yhat, nll, grad, fallback, out_mode, stream);
}

}  // namespace moihgp
