// recursion_x.hip -- the filter + NLL sweep (ihgp.h:81-93, :204-209) for STACKED models, state dim D in {4, 6, 8, 9, 12}.
//
// Same contract as recursion.hip (one wavefront owns one latent; series-major stream; NaN = missing), different
// machine mapping, because a D x D matrix no longer fits the register file next to the state:
//
//   * the per-latent matrices are WAVE-UNIFORM, so they are never held per lane: every product reads its matrix
//     through scalar loads (s_load from the latent's constant block, K$-resident) and feeds the FMAs as SGPR operands;
//     the vector registers hold only state-sized vectors;
//   * a segment is 64 lanes x 32 ticks.  z_j = sum_k g_k y_k (chunk response), then a 6-level Kogge-Stone scan
//     x_j = M x_{j-1} + z_j with the uniform powers M^(1,2,4,8,16,32) (lane shifts by ds_bpermute), then every lane
//     replays its 32 ticks from its start state in innovation form,  v = y - HA x;  x <- A x + K v  (== AKHA x + K y,
//     ihgp.h:90), which only touches the J diagonal blocks of A: those stay in SGPRs for the whole replay, HA and K in
//     VGPRs, so the replay loop has no memory access besides one LDS read and write per tick;
//   * a segment that holds a NaN (or a latent whose scan tables overflowed, rho(AKHA) > 1) is run tick by tick with the
//     rows of AKHA spread over the lanes (lane i owns row i, lane D owns HA; the state is gathered by v_readlane).
//
// Roofline: VALU-bound for D >= 9 in fp64 (per tick about D*DB + 3D FMA replay + 7 D^2 / 32 scan + D response
// against 16 B of traffic), near the HBM / VALU balance point for D = 6.  DESIGN.md 3.7.
#include "kernels_common.h"
#include <hip/hip_ext.h>

namespace moihgp {
namespace {

template <typename T> __device__ inline T bperm(int addr, T v);
template <> __device__ inline float bperm<float>(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}
template <> __device__ inline double bperm<double>(int addr, double v) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)u);
    unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// out += m v for a WAVE-UNIFORM matrix m (scalar loads, SGPR operands).  The scalar register file holds ~100 values, so the
// rows are fetched in batches of about 72 dwords; the scheduling barrier keeps the compiler from hoisting every s_load of a
// 12 x 12 product to the top (288 dwords: it would spill SGPRs into VGPR lanes).
// A wave-uniform view of a constant block for use INSIDE loops: the asm hides the pointer's provenance, so the loads cannot
// be hoisted out of the enclosing loop (the blocks are loop-invariant, and LICM would otherwise pull thousands of scalar
// loads in front of the segment loop and spill them); readfirstlane + the constant address space make every access through
// the result a scalar load (s_load, SGPR operand of the FMA).  The blocks are written by the update kernel only.
template <typename T> using uptr = const __attribute__((address_space(4))) T*;
template <typename T>
__device__ inline uptr<T> launder(const T* p) {
    unsigned long long u = reinterpret_cast<unsigned long long>(p);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return (uptr<T>)(((unsigned long long)hi << 32) | lo);
}

template <typename T, int D>
__device__ inline void matvec_u(const T* __restrict__ m0, const T (&v)[D], T (&out)[D]) {
    const uptr<T> m = launder(m0);
    constexpr int DW = D * (int)(sizeof(T) / 4), RB = 72 / DW < 1 ? 1 : 72 / DW;
#pragma unroll
    for (int i = 0; i < D; i++) {
        T s = out[i];
#pragma unroll
        for (int j = 0; j < D; j++) s = fma(m[i * D + j], v[j], s);
        out[i] = s;
        if ((i + 1) % RB == 0 && i + 1 < D) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int DB, int J, bool WRITE, bool NLL, bool TAIL>
__device__ inline void replay(const T* __restrict__ c, T* tile_lane, int first_tick, int n, const T (&hav)[DB * J], const T (&kv)[DB * J],
                              T (&xs)[DB * J], double& acc, unsigned& nobs) {
    constexpr int D = DB * J;
    using Lay = XC<D>;
    const uptr<T> ab = launder(c + Lay::AB);       // once per segment: J*DB*DB uniform scalars, SGPR-resident across the loop
    T a[J * DB * DB];
#pragma unroll
    for (int i = 0; i < J * DB * DB; i++) a[i] = ab[i];
#pragma unroll 1
    for (int k = 0; k < kChunkX; k++) {
        const T y = tile_lane[k];
        T hx = 0;
#pragma unroll
        for (int i = 0; i < D; i++) hx = fma(hav[i], xs[i], hx);
        const T v = y - hx;
        const bool valid = !TAIL || (first_tick + k < n);
        if (NLL && valid) {
            const double vd = (double)v;
            acc = fma(vd, vd, acc);                                 // ihgp.h:206-207, pre-step state
            nobs++;
        }
        T xn[D];
#pragma unroll
        for (int j = 0; j < J; j++)
#pragma unroll
            for (int r = 0; r < DB; r++) {
                T sum = kv[j * DB + r] * v;
#pragma unroll
                for (int q = 0; q < DB; q++) sum = fma(a[j * DB * DB + r * DB + q], xs[j * DB + q], sum);
                xn[j * DB + r] = sum;                               // ihgp.h:90 as A x + K (y - HA x)
            }
#pragma unroll
        for (int i = 0; i < D; i++) xs[i] = valid ? xn[i] : xs[i];
        if (WRITE) tile_lane[k] = xn[0];                            // ihgp.h:91 `yhat = xnew(0, 0)`, literally
    }
}

// n ticks of the tile, one after the other: lane i < D owns row i of AKHA (and of A for missing ticks), lane D owns HA.
template <typename T, int D, bool WRITE, bool NLL>
__device__ inline void sequential(const T* __restrict__ c, T* tile, int stride, int n, int lane, T (&xc)[D], double& acc, unsigned& nobs) {
    using Lay = XC<D>;
    T rowF[D], rowP[D], kk = 0, xv = 0;
    const int r = lane < D ? lane : 0;
#pragma unroll
    for (int j = 0; j < D; j++) {
        rowF[j] = lane < D ? c[Lay::AKHA + r * D + j] : (lane == D ? c[Lay::HA + j] : T(0));
        rowP[j] = lane < D ? c[Lay::A + r * D + j] : T(0);
    }
    if (lane < D) kk = c[Lay::K + r];
#pragma unroll
    for (int i = 0; i < D; i++) if (lane == i) xv = xc[i];
#pragma unroll 1
    for (int t = 0; t < n; t++) {
        T* slot = tile + (t / kChunkX) * stride + (t % kChunkX);
        const T y = *slot;                                           // same address in every lane: one broadcast read
        const bool miss = (y != y);
        T s = 0;
        if (miss) {                                                  // ihgp.h:83-87: x <- A x, no likelihood term
#pragma unroll
            for (int j = 0; j < D; j++) s = fma(rowP[j], read_lane(xv, j), s);
        } else {
#pragma unroll
            for (int j = 0; j < D; j++) s = fma(rowF[j], read_lane(xv, j), s);
            s = fma(kk, y, s);
            if (NLL) {
                const double v = (double)(y - read_lane(s, D));      // lane D computed HA x
                if (lane == 0) { acc = fma(v, v, acc); nobs++; }
            }
        }
        xv = s;
        if (WRITE) { const T yh = read_lane(s, 0); if (lane == 0) *slot = yh; }
    }
#pragma unroll
    for (int i = 0; i < D; i++) xc[i] = read_lane(xv, i);
}

template <typename T, int DB, int J, bool WRITE, bool NLL, int WPB>
__global__ void __launch_bounds__(64 * WPB)
filter_x_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64,
                T* __restrict__ x, T* __restrict__ yhat, double* __restrict__ nll) {
    constexpr int D = DB * J;
    using V = typename VecOf<T>::type;
    using Lay = XC<D>;
    constexpr int CK = kChunkX, EPV = 16 / sizeof(T), STRIDE = CK + EPV, SEG = 64 * CK;
    __shared__ __attribute__((aligned(16))) T tiles[WPB][64 * STRIDE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = (size_t)blockIdx.x * WPB + wave;
    if (l >= L) return;                                              // no workgroup barrier below
    const T* __restrict__ c = cbT + l * Lay::SIZE;
    T* tile = tiles[wave];
    T* tile_lane = tile + lane * STRIDE;
    const T* row = Ty + l * ld;
    T* orow = WRITE ? yhat + l * ld : nullptr;
    T xc[D];
#pragma unroll
    for (int i = 0; i < D; i++) xc[i] = x[l * D + i];
    double acc = 0.0;
    unsigned nobs = 0;
    const bool scan_ok = c[Lay::SCANOK] != T(0);
    // HA and K of this latent, replicated in vector registers for the replay loop (the asm pins them there: as uniform
    // values the compiler would otherwise park them in scalar registers and run out)
    T hav[D], kv[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        hav[i] = c[Lay::HA + i];
        kv[i] = c[Lay::K + i];
        asm volatile("" : "+v"(hav[i]), "+v"(kv[i]));
    }

    for (size_t t0 = 0; t0 < Tlen; t0 += SEG) {
        const int n = (int)(Tlen - t0 < (size_t)SEG ? Tlen - t0 : (size_t)SEG);
        // ---- stage in: coalesced 16-byte loads, chunk-major into the padded tile ----
#pragma unroll
        for (int r = 0; r < CK / EPV; r++) {
            const int e = (r * 64 + lane) * EPV;
            T vals[EPV];
            if (e < n) unpack<T>(nt_load(reinterpret_cast<const V*>(row + t0 + e)), vals);
#pragma unroll
            for (int q = 0; q < EPV; q++) if (e + q >= n) vals[q] = T(0);        // beyond the stream: inert zeros
            *reinterpret_cast<V*>(tile + (e / CK) * STRIDE + (e % CK)) = pack<T>(vals);
        }
        wave_lds_fence();
        // ---- chunk response z = sum_k g_k y_k, and the missing-data test ----
        T z[D];
#pragma unroll
        for (int i = 0; i < D; i++) z[i] = T(0);
        bool bad = false;
#pragma unroll 1
        for (int kv = 0; kv < CK / EPV; kv++) {
            T yv[EPV];
            unpack<T>(*reinterpret_cast<const V*>(tile_lane + kv * EPV), yv);
            const uptr<T> g = launder(c + Lay::G + kv * EPV * D);
#pragma unroll
            for (int q = 0; q < EPV; q++) {
                bad = bad || (yv[q] != yv[q]);
#pragma unroll
                for (int i = 0; i < D; i++) z[i] = fma(g[q * D + i], yv[q], z[i]);
            }
        }
        if (!scan_ok || __builtin_amdgcn_ballot_w64(bad) != 0) {
            sequential<T, D, WRITE, NLL>(c, tile, STRIDE, n, lane, xc, acc, nobs);
        } else {
            // ---- carry-in on lane 0, then the 64-lane Kogge-Stone scan with uniform powers ----
            T t[D];
#pragma unroll
            for (int i = 0; i < D; i++) t[i] = T(0);
            matvec_u<T, D>(c + Lay::SP, xc, t);
#pragma unroll
            for (int i = 0; i < D; i++) z[i] += (lane == 0) ? t[i] : T(0);
#pragma unroll
            for (int lv = 0; lv < 6; lv++) {
                const int s = 1 << lv, addr = ((lane - s) & 63) * 4;
#pragma unroll
                for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, z[i]); t[i] = lane >= s ? m : T(0); }
                matvec_u<T, D>(c + Lay::SP + lv * D * D, t, z);
            }
            // ---- start state of every lane = end state of the lane before it ----
            T xs[D];
            {
                const int addr = ((lane - 1) & 63) * 4;
#pragma unroll
                for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, z[i]); xs[i] = lane >= 1 ? m : xc[i]; }
            }
            if (n == SEG) replay<T, DB, J, WRITE, NLL, false>(c, tile_lane, lane * CK, n, hav, kv, xs, acc, nobs);
            else replay<T, DB, J, WRITE, NLL, true>(c, tile_lane, lane * CK, n, hav, kv, xs, acc, nobs);
            const int jl = (n - 1) / CK;                             // the lane that holds the last tick
#pragma unroll
            for (int i = 0; i < D; i++) xc[i] = read_lane(xs[i], jl);
        }
        // ---- stage out ----
        if (WRITE) {
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < CK / EPV; r++) {
                const int e = (r * 64 + lane) * EPV;
                if (e < n) nt_store(*reinterpret_cast<const V*>(tile + (e / CK) * STRIDE + (e % CK)), reinterpret_cast<V*>(orow + t0 + e));
            }
        }
        wave_lds_fence();
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < D; i++) x[l * D + i] = xc[i];
    }
    if (NLL) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { acc += __shfl_xor(acc, o, 64); nobs += __shfl_xor(nobs, o, 64); }
        if (lane == 0) {
            const double* c64 = cb64 + l * Lay::SIZE;
            nll[l] = 0.5 * (acc / c64[Lay::S] + (double)nobs * c64[Lay::LOGS]);
        }
    }
}

template <typename T, int DB, int J, int WPB>
int launch_x(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, void* x, void* yhat, double* nll,
             hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    dim3 block(64 * WPB), grid((unsigned)((L + WPB - 1) / WPB));
    const T* ty = static_cast<const T*>(Ty);
    T* xs = static_cast<T*>(x);
    T* yh = static_cast<T*>(yhat);
    if (yhat && nll) hipExtLaunchKernelGGL((filter_x_kernel<T, DB, J, true, true, WPB>), grid, block, 0, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xs, yh, nll);
    else if (yhat) hipExtLaunchKernelGGL((filter_x_kernel<T, DB, J, true, false, WPB>), grid, block, 0, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xs, yh, nll);
    else if (nll) hipExtLaunchKernelGGL((filter_x_kernel<T, DB, J, false, true, WPB>), grid, block, 0, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xs, yh, nll);
    else hipExtLaunchKernelGGL((filter_x_kernel<T, DB, J, false, false, WPB>), grid, block, 0, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xs, yh, nll);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("filter_x_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

template <typename T, int DB, int J>
int launch_xd(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, void* x, void* yhat, double* nll,
              hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    // few latents: one wavefront per workgroup, so that they spread over the compute units
    if (L < 1024) return launch_x<T, DB, J, 1>(Ty, Tlen, ld, L, cbT, cb64, x, yhat, nll, stream, ev0, ev1);
    return launch_x<T, DB, J, 4>(Ty, Tlen, ld, L, cbT, cb64, x, yhat, nll, stream, ev0, ev1);
}

}  // namespace

int launch_filter_stream_x(int kernel, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const float* cb32,
                           void* x, void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (L == 0) return 0;
    const int base = kernel_base(kernel), J = kernel_stack(kernel);
#define MOIHGP_X_CASE(DBB, JJ)                                                                                        \
    if (base == (DBB == 2 ? 0 : 1) && J == JJ)                                                                        \
        return dtype == 0 ? launch_xd<double, DBB, JJ>(Ty, T, ld, L, cb64, cb64, x, yhat, nll, stream, ev0, ev1)      \
                          : launch_xd<float, DBB, JJ>(Ty, T, ld, L, cb32, cb64, x, yhat, nll, stream, ev0, ev1)
    MOIHGP_X_CASE(2, 2); MOIHGP_X_CASE(2, 3); MOIHGP_X_CASE(2, 4);
    MOIHGP_X_CASE(3, 2); MOIHGP_X_CASE(3, 3); MOIHGP_X_CASE(3, 4);
#undef MOIHGP_X_CASE
    set_last_error("stacked kernel id %d is not built", kernel);
    return 1;
}

}  // namespace moihgp
