// gaps_x.hip -- missing ticks in the many-latent sweep of the stacked models (state dim 4 .. 12) by EXACT IMPUTATION (round 4).
// Reference semantics: ihgp.h:83-87 -- a NaN observation advances the state by x <- A x (no correction), i.e. the innovation form
// x' = A x + K v with v = 0 -- and ihgp.h:204-209 adds no likelihood term for it.
//
// Why.  The sweep of recursion_x.hip solves a segment of 64 x 32 ticks in parallel because every chunk moves the state by the same matrix
// AKHA^32.  A chunk with a gap moves it by a matrix of its own; for d <= 6 that matrix fits a lane and the chunks' maps are scanned, beyond it
// does not, and the second pass of that file treats gaps as broken links of the scan (one scan + one replay per chunk with a gap) or walks the
// segment tick by tick: 12-16 x the gap-free sweep once most chunks hold a gap (1 % of the ticks missing: 3.8 ms against 0.31 ms at d = 12,
// open since round 2).
//
// What.  A missing tick is an observation that happens to equal its own prediction: with y_p := w_p = HA x_p (the predicted mean at the gap) the
// ordinary recursion gives v_p = 0 and x_{p+1} = A x_p -- the reference's branch.  The w_p are not known in advance, but they obey a SCALAR
// triangular system.  Sweep the stream with the gaps set to zero (state x'); then e = x - x' moves by AKHA between gaps and is kicked by
// K w_p at each gap, so
//         w_p = HA x'_p + sum over gaps q < p of  s_(p-q-1) w_q,        s_k = HA AKHA^k K   (the filter's scalar impulse response),
// and HA x'_p is what that first sweep writes in place of its filtered means (the PRED instantiation of recursion_x.hip: y - v, exactly HA x'
// where y' = 0; the filtered mean xnew(0, 0) would not do -- for the stacked models H sums over the blocks).  The sum runs over the few gaps inside
// the decay of s.  Filling the gaps with w_p and sweeping ONCE MORE gives the true filtered means, states and sum of v^2 (the gaps contribute
// v = 0 to it); only the count of observed ticks needs correcting: nll -= n_gaps log(S) / 2.
//   two gap-free sweeps of the latents that hold gaps + a scalar recursion over their gaps, whatever the density of the gaps.
// The first pass of recursion_x.hip hands such latents over (flags); they are gathered into a compact bank whose size only the device knows (the
// sweeps run over L slots and leave the empty ones at once), the gather listing every gap's tick on the way; the second sweep writes straight into
// the caller's arrays.  A latent with fewer than `min_gaps` gaps, or one this cannot take -- an impulse response that has not decayed within kSMax
// ticks, more gaps inside its decay than the window holds -- keeps its flag and takes the second pass of recursion_x.hip as before.
#include "kernels_common.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace moihgp {
namespace {

constexpr int kSMax = 1024;        // impulse-response table per latent (ticks); beyond it the response must be negligible
constexpr int kRing = 256;         // gaps inside the decay window a wave keeps (4 per lane, in registers)
constexpr int kPart = 2048;        // ticks one workgroup of the gather moves; the gaps are listed per part
constexpr int kCtl = 16;           // control block (ints): [0] slots in use, [kCtl + c] slot c is live, [kCtl + L + c] the latent it holds

constexpr int DPP_ROW_BCAST31 = 0x143;
__device__ inline double wave_sum(double v) {                          // all in DPP; the total comes back uniform
    v += dpp0<DPP_ROW_SHR + 1, 0xF>(v);
    v += dpp0<DPP_ROW_SHR + 2, 0xF>(v);
    v += dpp0<DPP_ROW_SHR + 4, 0xF>(v);
    v += dpp0<DPP_ROW_SHR + 8, 0xF>(v);
    v += dpp0<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp0<DPP_ROW_BCAST31, 0xC>(v);
    return read_lane(v, 63);
}

// ---- flagged latents -> compact list (order preserving), one workgroup ---------------------------------------------------------------
__global__ void __launch_bounds__(1024) gap_compact_kernel(const int* __restrict__ flags, size_t L, int* __restrict__ ctl) {
    __shared__ int wsum[16];
    __shared__ int base;
    int* idx = ctl + kCtl + L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (size_t l0 = 0; l0 < L; l0 += 1024) {
        const size_t l = l0 + tid;
        const int f = (l < L && flags[l] != 0) ? 1 : 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(f != 0);
        const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __builtin_popcountll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++) off += wsum[w];
        if (f) idx[off + before] = (int)l;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; w++) t += wsum[w]; base += t; }
        __syncthreads();
    }
    if (tid == 0) ctl[0] = base;
}

// ---- gather: stream rows (NaN -> 0) of the listed latents into the compact bank; the gaps' ticks, in order, per part --------------------------------
// E scalars per load (16 bytes when the rows allow it)
template <typename T, int E>
__global__ void __launch_bounds__(256) gap_gather_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, const int* __restrict__ ctl, size_t L,
                                                         T* __restrict__ cTy, int* __restrict__ glist, int* __restrict__ cntp) {
    constexpr int NJ = kPart / (256 * E);
    typedef T Vec __attribute__((ext_vector_type(E)));
    __shared__ int wc[NJ][4];
    const int c = blockIdx.x;
    if (c >= ctl[0]) return;
    const size_t l = (size_t)ctl[kCtl + L + c];
    const size_t parts = gridDim.y, t0 = (size_t)blockIdx.y * kPart;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Vec y[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const size_t t = t0 + ((size_t)j * 256 + tid) * E;
        if (E > 1 && t + E <= Tlen) y[j] = __builtin_nontemporal_load(reinterpret_cast<const Vec*>(Ty + l * ld + t));
        else {
#pragma unroll
            for (int e = 0; e < E; e++) y[j][e] = t + e < Tlen ? Ty[l * ld + t + e] : T(0);
        }
    }
    // the gaps of this part, in tick order: (j, wave, lane, e)
    int before[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        int b = 0, n = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(y[j][e] != y[j][e]);
            b += __builtin_popcountll(m & ((1ull << lane) - 1ull));
            n += __builtin_popcountll(m);
        }
        before[j] = b;
        if (lane == 0) wc[j][wave] = n;
    }
    __syncthreads();
    int run = 0;
    int* list = glist + ((size_t)c * parts + blockIdx.y) * kPart;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        int off = run;
#pragma unroll
        for (int w = 0; w < 4; w++) { off += w < wave ? wc[j][w] : 0; run += wc[j][w]; }
        off += before[j];
        const size_t t = t0 + ((size_t)j * 256 + tid) * E;
        Vec o;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const bool miss = y[j][e] != y[j][e];
            if (miss) list[off++] = (int)(t + e);
            o[e] = miss ? T(0) : y[j][e];
        }
        if (E > 1 && t + E <= Tlen) *reinterpret_cast<Vec*>(cTy + (size_t)c * ld + t) = o;
        else {
#pragma unroll
            for (int e = 0; e < E; e++) if (t + e < Tlen) cTy[(size_t)c * ld + t + e] = o[e];
        }
    }
    if (tid == 0) cntp[(size_t)c * parts + blockIdx.y] = run;
}

// ---- the scalar recursion over one latent's gaps: one wavefront per compact slot ------------------------------------------------------
// imp: per latent of the full bank, the filter's response to a unit observation at tick 0 from a zero state, as the PRED sweep writes it:
// imp[k + 1] = HA AKHA^k K = s_k.  On return the gaps of compact stream c are filled with their w_p and ctl[kCtl + c] = 1; a slot this cannot
// solve gets 0 there (left to the second pass).  Everything that steers the loops is kept in scalar registers.
template <typename T>
__global__ void __launch_bounds__(256) gap_solve_kernel(size_t Tlen, size_t ld, size_t L, int* __restrict__ ctl, const T* __restrict__ imp, const T* __restrict__ cyh,
                                                        T* __restrict__ cTy, const int* __restrict__ glist, const int* __restrict__ cntp, int parts, int min_gaps) {
    __shared__ double stab[4][kSMax];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 4 + wave;
    if (c >= ctl[0]) return;                                            // (no workgroup barrier below: waves are independent)
    const int* cnts = cntp + (size_t)c * parts;
    int gaps = 0;
    for (int p0 = 0; p0 < parts; p0 += 64) gaps += p0 + lane < parts ? cnts[p0 + lane] : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) gaps += __shfl_xor(gaps, o, 64);
    gaps = __builtin_amdgcn_readfirstlane(gaps);
    if (gaps < min_gaps) return;
    double* st = stab[wave];
    // ---- impulse response table, and where it has died out ----
    const T* si = imp + (size_t)ctl[kCtl + L + c] * kSMax;
    double smax = 0.0;
    bool bad = false;
#pragma unroll 4
    for (int k = lane; k < kSMax; k += 64) {
        const double sv = k + 1 < kSMax ? (double)si[k + 1] : 0.0;
        st[k] = sv;
        bad |= !(fabs(sv) < 1e300);
        smax = fmax(smax, fabs(sv));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) smax = fmax(smax, __shfl_xor(smax, o));
    const double tol = sizeof(T) == 8 ? 1e-17 : 1e-9;                  // (the sweep itself drops scan levels below 1e-20 / 1e-10)
    int kd = 0;                                                        // one past the last k whose |s_k| still matters
    for (int k = lane; k < kSMax; k += 64)
        if (fabs(st[k]) > tol * smax) kd = k + 1;                      // (each lane reads back what it wrote)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kd = max(kd, __shfl_xor(kd, o));
    const int kdec = __builtin_amdgcn_readfirstlane(kd);
    bool ok = __builtin_amdgcn_ballot_w64(bad) == 0 && kdec <= kSMax - 64 && Tlen < (1u << 30);
    wave_lds_fence();
    // ---- the gaps one after the other, a part's first 64 fetched while the part before is worked on ----
    const T* yh = cyh + (size_t)c * ld;                                // first sweep's predicted observations HA x' (gaps swept as zeros)
    T* fill = cTy + (size_t)c * ld;
    const int* lists = glist + (size_t)c * parts * kPart;
    int rp[4];                                                         // ring of the last kRing gaps: entry e in lane e % 64, register e / 64
    double rw[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { rp[r] = -(1 << 30); rw[r] = 0.0; }
    int total = 0;
    const int tmax = (int)Tlen - 1;
    auto clampt = [&](int t) { return t < 0 ? 0 : (t > tmax ? tmax : t); };       // (a list entry past the part's count is whatever the memory held)
    // stage A (two parts ahead): count and the first 64 ticks; stage B (one part ahead): HA x' at those ticks
    int nA = cnts[0], posA = lists[lane];
    int nB = nA, posB = posA;
    T hvB = ok ? yh[clampt(posB)] : T(0);
    nA = parts > 1 ? cnts[1] : 0; posA = parts > 1 ? lists[kPart + lane] : 0;
    for (int part = 0; part < parts && ok; part++) {
        const int n = __builtin_amdgcn_readfirstlane(nB);
        int pos = posB;
        double hv = (double)hvB;
        nB = nA; posB = posA;
        hvB = part + 1 < parts ? yh[clampt(posB)] : T(0);
        nA = part + 2 < parts ? cnts[part + 2] : 0;
        posA = part + 2 < parts ? lists[(size_t)(part + 2) * kPart + lane] : 0;
        for (int g0 = 0; g0 < n && ok; g0 += 64) {
            if (g0 > 0) {                                               // more than 64 gaps in this part: fetched as they come
                pos = g0 + lane < n ? lists[(size_t)part * kPart + g0 + lane] : 0;
                hv = (double)yh[clampt(pos)];
            }
            const int m = n - g0 < 64 ? n - g0 : 64;
            for (int e = 0; e < m; e++) {
                const int tp = __builtin_amdgcn_readlane(pos, e);
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r == 0 || total > 64 * r) {                    // (scalar: register r holds entries only after 64 r gaps)
                        const int k = tp - rp[r] - 1;                  // >= 0: every entry is an earlier tick (an empty one: huge)
                        const double sv = st[k < kdec ? k : kSMax - 1];      // (the table's last entry is zero)
                        acc = fma(sv, rw[r], acc);
                    }
                }
                const double w = read_lane(hv, e) + wave_sum(acc);
                const int slot = total & (kRing - 1), sr = slot >> 6, sl = slot & 63;
                const int oldp = __builtin_amdgcn_readlane(sr == 0 ? rp[0] : sr == 1 ? rp[1] : sr == 2 ? rp[2] : rp[3], sl);
                if (tp - oldp - 1 < kdec || tp > tmax || tp < 0) { ok = false; break; }     // more gaps inside the decay than the window holds
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool mine = r == sr && lane == sl;
                    rp[r] = mine ? tp : rp[r];
                    rw[r] = mine ? w : rw[r];
                }
                if (lane == 0) fill[tp] = (T)w;
                total++;
            }
        }
    }
    if (lane == 0) ctl[kCtl + c] = ok ? 1 : 0;
}

template <typename T>
__global__ void __launch_bounds__(256) gap_impulse_kernel(T* __restrict__ imp_in, size_t L) {
    const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (l < L) imp_in[l * kSMax] = T(1);
}

template <typename T>
int run_typed(const GapBank& b, int kernel, int d, const void* Ty, size_t T_, size_t ld, size_t L, const double* cb64, const void* cbT, const void* xin, void* x,
              void* yhat, double* nll, size_t ldo, int* flags, int min_gaps, hipStream_t s) {
    const int dtype = sizeof(T) == 8 ? 0 : 1;
    const unsigned parts = (unsigned)((T_ + kPart - 1) / kPart);
    T* cTy = static_cast<T*>(b.stream_in);
    T* cyh = static_cast<T*>(b.stream_out);
    T* cx = static_cast<T*>(b.x1);
    const float* cb32 = dtype == 0 ? nullptr : (const float*)cbT;
    constexpr int E = 16 / (int)sizeof(T);
    const bool vec = ld % E == 0 && (reinterpret_cast<uintptr_t>(Ty) & 15) == 0;
    hipLaunchKernelGGL(gap_compact_kernel, dim3(1), dim3(1024), 0, s, (const int*)flags, L, b.ctl);
#define MOIHGP_GATHER(E_) hipLaunchKernelGGL((gap_gather_kernel<T, E_>), dim3((unsigned)L, parts), dim3(256), 0, s, (const T*)Ty, T_, ld, (const int*)b.ctl, L, cTy, b.glist, b.cntp)
    if (vec) MOIHGP_GATHER(E); else MOIHGP_GATHER(1);
#undef MOIHGP_GATHER
    // The sweeps below read the stream of slot c from the compact bank, its constants and start state from the full arrays (the map in b.ctl), and
    // leave a slot with fewer than min_gaps gaps alone (force_slices = -5 / -4: recursion_x.hip's compact modes).
    // first sweep: gaps as zeros, predicted observations out
    if (int rc = launch_filter_stream_x(kernel, dtype, cTy, T_, ld, L, cb64, cb32, xin, cx, cyh, nullptr, s, nullptr, nullptr, reinterpret_cast<double*>(b.cntp), parts,
                                        -5, ld, b.ctl, nullptr, nullptr, min_gaps, 0, nullptr, nullptr)) return rc;
    hipLaunchKernelGGL((gap_solve_kernel<T>), dim3((unsigned)((L + 3) / 4)), dim3(256), 0, s, T_, ld, L, b.ctl, (const T*)b.imp_out, (const T*)cyh, cTy, (const int*)b.glist,
                       (const int*)b.cntp, (int)parts, min_gaps);
    // second sweep: gaps filled with their own predictions; its means, end states and NLLs (less the gaps' terms) go straight to the caller's arrays, and
    // the flag of every latent it sweeps is cleared
    if (int rc = launch_filter_stream_x(kernel, dtype, cTy, T_, ld, L, cb64, cb32, xin, x, yhat, nll, s, nullptr, nullptr, reinterpret_cast<double*>(b.cntp), parts,
                                        -4, ldo, b.ctl, reinterpret_cast<double*>(flags), nullptr, min_gaps, 0, nullptr, nullptr)) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("gap imputation launch: %s", hipGetErrorString(e)); return 2; }
    if (const char* tr = getenv("MOIHGP_GAP_TRACE"); tr && tr[0] == '1') {       // diagnostics (the tests read it): synchronises the stream
        int n = 0;
        (void)hipMemcpyAsync(&n, b.ctl, sizeof(int), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        std::vector<int> st((size_t)(n > 0 ? n : 0)), ng(st.size() * parts);
        if (n > 0) { (void)hipMemcpy(st.data(), b.ctl + kCtl, sizeof(int) * n, hipMemcpyDeviceToHost); (void)hipMemcpy(ng.data(), b.cntp, sizeof(int) * ng.size(), hipMemcpyDeviceToHost); }
        long taken = 0, solved = 0, gaps = 0;
        for (int i = 0; i < n; i++) {
            long g = 0;
            for (unsigned p = 0; p < parts; p++) g += ng[(size_t)i * parts + p];
            if (g >= min_gaps) { taken++; if (st[i]) { solved++; gaps += g; } }
        }
        fprintf(stderr, "moihgp gap imputation: %ld latents handed over, %ld solved, %ld gaps filled\n", taken, solved, gaps);
    }
    return 0;
}

}  // namespace

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

size_t gap_bank_bytes(int d, int dtype, size_t L, size_t ld, size_t T) {
    const size_t es = dtype == 0 ? 8 : 4;
    const size_t parts = (T + kPart - 1) / kPart;
    return up256((kCtl + 2 * L) * sizeof(int)) + 2 * up256(L * ld * es) + 2 * up256(L * (size_t)d * es) +
           2 * up256(L * (size_t)kSMax * es) + up256(L * parts * kPart * sizeof(int)) + up256(L * parts * sizeof(int));
}

GapBank gap_bank_carve(void* base, int d, int dtype, size_t L, size_t ld, size_t T) {
    const size_t es = dtype == 0 ? 8 : 4;
    const size_t parts = (T + kPart - 1) / kPart;
    unsigned char* p = static_cast<unsigned char*>(base);
    GapBank b;
    b.ctl = reinterpret_cast<int*>(p); p += up256((kCtl + 2 * L) * sizeof(int));
    b.imp_in = p; p += up256(L * (size_t)kSMax * es);                  // (the constant parts first: they stay put when only T or ld changes)
    b.imp_out = p; p += up256(L * (size_t)kSMax * es);
    b.xz = p; p += up256(L * (size_t)d * es);
    b.stream_in = p; p += up256(L * ld * es);
    b.stream_out = p; p += up256(L * ld * es);
    b.x1 = p; p += up256(L * (size_t)d * es);
    b.glist = reinterpret_cast<int*>(p); p += up256(L * parts * kPart * sizeof(int));
    b.cntp = reinterpret_cast<int*>(p);
    return b;
}

// the constant parts of a freshly carved bank: unit impulses, a zero start state.  Asynchronous on `s`.
int gap_bank_init(const GapBank& b, int d, int dtype, size_t L, hipStream_t s) {
    const size_t es = dtype == 0 ? 8 : 4;
    if (hipMemsetAsync(b.imp_in, 0, L * (size_t)kSMax * es, s) != hipSuccess || hipMemsetAsync(b.xz, 0, L * (size_t)d * es, s) != hipSuccess) {
        set_last_error("gap bank: memset failed"); return 2;
    }
    if (dtype == 0) hipLaunchKernelGGL(gap_impulse_kernel<double>, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, (double*)b.imp_in, L);
    else hipLaunchKernelGGL(gap_impulse_kernel<float>, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, (float*)b.imp_in, L);
    return 0;
}

// The impulse responses of all L filters into b.imp_out (the PRED sweep over a unit observation at tick 0, the full bank): once per parameter update.
int launch_gap_impulse(const GapBank& b, int kernel, int dtype, size_t L, const double* cb64, const float* cb32, hipStream_t s) {
    return launch_filter_stream_x(kernel, dtype, b.imp_in, kSMax, kSMax, L, cb64, cb32, b.xz, b.x1, b.imp_out, nullptr, s, nullptr, nullptr, nullptr, 0,
                                  -6, kSMax, nullptr, nullptr, nullptr, -1, 0, nullptr, nullptr);
}

// The latents flagged in `flags` (by the first pass of the stacked sweep: force_slices = -2) whose stream holds min_gaps missing ticks or more are
// swept by imputation; the flag of every latent this solved is cleared, the others keep theirs for the second pass.  Asynchronous on `s`, no host
// synchronisation.
int launch_gap_imputation(const GapBank& b, int kernel, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const float* cb32, const void* xin, void* x,
                          void* yhat, double* nll, size_t ldo, int* flags, int min_gaps, hipStream_t s) {
    const int base = kernel_base(kernel), J = kernel_stack(kernel);
    const int d = (base == 0 ? 2 : 3) * (J ? J : 1);
    if (d != 4 && d != 6 && d != 8 && d != 9 && d != 12) { set_last_error("gap imputation: state dimension %d is not built", d); return 1; }
    if (dtype == 0) return run_typed<double>(b, kernel, d, Ty, T, ld, L, cb64, cb64, xin, x, yhat, nll, ldo, flags, min_gaps, s);
    return run_typed<float>(b, kernel, d, Ty, T, ld, L, cb64, cb32, xin, x, yhat, nll, ldo, flags, min_gaps, s);
}

}  // namespace moihgp
