// recursion.hip -- the hot path: per-latent steady-state Kalman recursion + NLL over whole
// time streams (reference include/moihgp/ihgp.h:81-100 step, :204-209 negLogLikelihood, driven
// tick by tick by moihgp.h:367-373 / moihgp_online.h:61-70 / moihgp_regression.h:42-50).
//
// Mapping (MI355X, wave64): ONE WAVEFRONT OWNS ONE LATENT.  The latent's stream is series-major
// (contiguous in time), so the wave reads it with fully coalesced 16-byte-per-lane loads, 64*CK
// ticks ("segment") at a time.  Inside a segment the 64 lanes are 64 consecutive time chunks of
// CK ticks.  The recursion x <- AKHA x + K y is linear time-invariant, so a segment is solved
// exactly (up to rounding) without any serial walk over its 64*CK ticks:
//   1. chunk response from a zero state as a CK-tap dot product z_j = sum_k g_k y_k,
//      g_k = AKHA^(CK-1-k) K (no dependency chain); lane 0 adds M x_in, M = AKHA^CK
//   2. inclusive scan s_j = M s_{j-1} + z_j over the 64 lanes entirely in DPP (kernels_common.h dpp_scan)
//   3. replay of the chunk from its true start state in innovation form, emitting yhat / v^2.
// g, M^(1,2,4,8), M^(lane%16+1) are per-latent tables written by IHGP::update (stationary.hip); the y
// chunk lives in VGPRs between steps 1 and 3, so HBM sees each stream element exactly once in and
// once out (filtered means leave through streaming stores).  The coalesced <-> chunk-per-lane
// re-layout goes through a wave-private, padded LDS tile (no workgroup barrier: the four waves of a
// block never communicate).
//
// Missing data (NaN y, ihgp.h:83-87: x <- A x) makes chunk maps lane-dependent: generic_segment()
// scans (M_j, z_j) pairs.  Ragged tails run the fast path with a masked replay.  With few latents
// (L < 1024) the slices of one latent are spread over the waves of a workgroup (time split, below).
#include "kernels_common.h"
#include <hip/hip_ext.h>

#ifndef MOIHGP_GENERIC_UNROLL
#define MOIHGP_GENERIC_UNROLL 4
#endif

namespace moihgp {
namespace {

// Per-latent constants of the fast path.  sp / g / a / k are wave-uniform; pj is per lane.
template <typename T, int D, int CK>
struct FastConst {
    T a[D * D];        // A = expm(dt F): x <- A x + K v is the innovation form of ihgp.h:90 (v = y - HA x)
    T k[D];            // K
    T g[CK * D];       // g_k = AKHA^(CK-1-k) K
    const T* sp;       // M^(1,2,4,8), M = AKHA^CK: 4*D*D scalars parked in LDS (wave-private), re-read at every scan level
    T pj[D * D];       // M^(lane%16 + 1)
};

// Step 3 of the segment solve: a lane replays its chunk from its true start state xs in innovation form (H = e0^T, so HA is row 0 of A):
//     hx = A0.x ; v = y - hx ; yhat = hx + K0 v ; x_i <- A_i.x + K_i v
// y[] holds the chunk on entry and the filtered means on exit, xs the state after the lane's last valid tick.  TAIL: ragged last segment
// (zero padded), the replay masked per tick.
template <typename T, int D, int CK, bool NLL, bool TAIL>
__device__ inline void replay_chunk(T* y, T* xs, const FastConst<T, D, CK>& c, size_t t0, size_t Tlen, double& acc, unsigned& nobs) {
    T part = 0;                              // sum of v^2 over this chunk (<= 16 terms) in stream precision
#pragma unroll
    for (int k = 0; k < CK; k++) {
        const bool valid = !TAIL || (t0 + k) < Tlen;
        T hx = 0;
#pragma unroll
        for (int j = 0; j < D; j++) hx = fma(c.a[j], xs[j], hx);
        T v = y[k] - hx;
        if (TAIL) v = valid ? v : T(0);
        if (NLL) {
            part = fma(v, v, part);
            if (TAIL) nobs += valid ? 1u : 0u;
        }
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
        for (int i = 0; i < D; i++) xs[i] = (TAIL && !valid) ? xs[i] : xn[i];
        y[k] = xs[0];
    }
    if (NLL) acc += (double)part;
}

// Fast path for one segment (64 lanes x CK ticks) of one latent without missing ticks.  y[] holds the
// lane's chunk on entry and the filtered means on exit.  xin (wave-uniform) is the state before the
// segment's first tick and is replaced by the state after its last valid tick.
//   1. chunk response from a zero state as a CK-tap dot product  z_j = sum_k g_k y_k  (no serial chain);
//      lane 0 adds M xin
//   2. inclusive scan s_j = M s_{j-1} + z_j over the 64 lanes, all in DPP: four in-row Kogge-Stone
//      levels (row_shr 1,2,4,8 with the uniform M^(1,2,4,8)), then three row_bcast:15 rounds that hand
//      the finished prefix of row r-1 to row r through the per-lane power M^(lane%16+1)
//   3. replay of the chunk from its true start state in innovation form (H = e0^T, so HA is row 0 of A):
//      hx = A0.x ; v = y - hx ; yhat = hx + K0 v ; x_i <- A_i.x + K_i v
//   TAIL = true: ragged last segment (zero padded): lanes past the end never feed a valid lane; the
//   replay is masked per tick.
// Returns false (wave-uniform) without touching xin / acc when a NaN was met: the caller then runs
// generic_segment() on the chunk still parked in LDS.
template <typename T, int D, int CK, bool NLL, bool TAIL, bool ENDONLY = false>
__device__ inline bool fast_segment(T* y, T* xin, const FastConst<T, D, CK>& c, int lane, size_t t0, size_t Tlen,
                                    double& acc, unsigned& nobs) {
    // ---- 1. chunk response ----------------------------------------------------------------------
    T z[D];
#pragma unroll
    for (int i = 0; i < D; i++) z[i] = T(0);
#pragma unroll
    for (int k = 0; k < CK; k++)
#pragma unroll
        for (int i = 0; i < D; i++) z[i] = fma(c.g[k * D + i], y[k], z[i]);
    bool bad = false;                       // a NaN y poisons z: one vote per segment detects missing data
#pragma unroll
    for (int i = 0; i < D; i++) bad |= (z[i] != z[i]);
    if (__any(bad)) return false;
    {
        T x0[D];
#pragma unroll
        for (int i = 0; i < D; i++) x0[i] = (lane == 0) ? xin[i] : T(0);
        matvec_acc<T, D>(c.sp, x0, z);      // lane 0: + M xin
    }
    // ---- 2. scan --------------------------------------------------------------------------------
    dpp_scan<T, D>(z, c.sp, c.pj);
    if (ENDONLY) {                           // only the state after the (full) segment is wanted: no replay
#pragma unroll
        for (int i = 0; i < D; i++) xin[i] = read_lane(z[i], 63);
        return true;
    }
    // exclusive state: lane j starts from the inclusive result of lane j-1, lane 0 from xin
    T xs[D];
#pragma unroll
    for (int i = 0; i < D; i++) xs[i] = wave_shr1(z[i], xin[i]);
    // ---- 3. replay ------------------------------------------------------------------------------
    replay_chunk<T, D, CK, NLL, TAIL>(y, xs, c, t0, Tlen, acc, nobs);
    // state after the last valid tick = final replay state of the lane that owns it
    int jl = 63;
    if (TAIL) {
        size_t last = Tlen - 1 - (t0 - (size_t)lane * CK);
        jl = (int)(last / CK);
    }
#pragma unroll
    for (int i = 0; i < D; i++) xin[i] = read_lane(xs[i], jl);
    return true;
}

// Missing-data path (ihgp.h:83-87: a NaN observation advances x <- A x, i.e. the innovation form with v = 0).  Chunk
// maps become lane-dependent, so every lane accumulates the transition matrix M_j of its chunk next to its response
// z_j, and the pairs are scanned: in-row Kogge-Stone with row_shr (lanes without a source compose with the identity),
// then the three row_bcast:15 rounds, where a lane folds the finished prefix of the previous row through its own in-row
// matrix.  Works on the chunk still parked in LDS (ch = this lane's CK elements) and leaves the filtered means there.
// Ticks past the end of the stream are treated as missing (they only influence lanes whose results are discarded) and
// frozen in the replay.  mseg (optional) receives the transition matrix of the whole segment (time split).
template <int O, typename T, int D>
__device__ inline void generic_scan_level(T* z, T* m) {
    T zp[D], mp[D * D];
#pragma unroll
    for (int i = 0; i < D; i++) zp[i] = dpp0<DPP_ROW_SHR + O, 0xF>(z[i]);
#pragma unroll
    for (int i = 0; i < D * D; i++) mp[i] = dpp_fill<DPP_ROW_SHR + O, 0xF>((i % (D + 1) == 0) ? T(1) : T(0), m[i]);
    matvec_acc<T, D>(m, zp, z);          // z = M zp + z
    matmul<T, D>(m, mp, m);              // M = M Mp
}

template <typename T, int D, int CK, bool NLL>
__device__ inline void generic_segment(T* ch, T* xin, const T* cb /* this latent's constant block */,
                                       int lane, size_t t0, size_t Tlen, double& acc, unsigned& nobs,
                                       T* mseg = nullptr /* out: transition matrix of the whole segment */) {
    using Lay = CB<D>;
    // The tick loops are unrolled by 4 only and read the chunk from LDS tick by tick: fully unrolled with the chunk in registers
    // this path set the whole kernel's register allocation and spilled scalars into the NaN-free loop (measured with 4 instead of
    // CK: C3 fp32 60.4 -> 58.3 us, 5 % missing ticks 137 -> 129 us; fp64 with missing ticks 232 -> 245 us).
    T a[D * D], kk[D];
#pragma unroll
    for (int i = 0; i < D * D; i++) a[i] = cb[Lay::A + i];
#pragma unroll
    for (int i = 0; i < D; i++) kk[i] = cb[Lay::K + i];
    // one masked tick in innovation form: hx = A0.x ; v = miss ? 0 : y - hx ; x <- A x + K v
    auto tick = [&](T* xs, T yk, bool miss, T& hx_out) -> T {
        T hx = 0;
#pragma unroll
        for (int j = 0; j < D; j++) hx = fma(a[j], xs[j], hx);
        const T v = miss ? T(0) : yk - hx;
        T xn[D];
        xn[0] = fma(kk[0], v, hx);
#pragma unroll
        for (int i = 1; i < D; i++) {
            T s = kk[i] * v;
#pragma unroll
            for (int j = 0; j < D; j++) s = fma(a[i * D + j], xs[j], s);
            xn[i] = s;
        }
#pragma unroll
        for (int i = 0; i < D; i++) xs[i] = xn[i];
        hx_out = hx;
        return v;
    };
    // ---- pass 1: chunk response z (lane 0 from the carried-in state) and chunk matrix M ------------------------------
    T z[D], m[D * D];
#pragma unroll
    for (int i = 0; i < D; i++) z[i] = (lane == 0) ? xin[i] : T(0);
#pragma unroll
    for (int i = 0; i < D * D; i++) m[i] = (i % (D + 1) == 0) ? T(1) : T(0);
#pragma unroll MOIHGP_GENERIC_UNROLL
    for (int k = 0; k < CK; k++) {
        const T yk = ch[k];
        const bool miss = (yk != yk) || (t0 + k) >= Tlen;
        T hx;
        tick(z, yk, miss, hx);
        // M <- (A - w K HA) M with w = !miss; HA M is row 0 of A M
        T am[D * D];
        matmul<T, D>(a, m, am);
#pragma unroll
        for (int i = 0; i < D; i++) {
            const T wk = miss ? T(0) : kk[i];
#pragma unroll
            for (int j = 0; j < D; j++) m[i * D + j] = fma(-wk, am[j], am[i * D + j]);
        }
    }
    // ---- scan of the (M_j, z_j) pairs ------------------------------------------------------------------------------------
    generic_scan_level<1, T, D>(z, m);
    generic_scan_level<2, T, D>(z, m);
    generic_scan_level<4, T, D>(z, m);
    generic_scan_level<8, T, D>(z, m);
    if (mseg) {                          // segment matrix = row3 * row2 * row1 * row0 (in-row totals sit in lanes 15, 31, 47, 63)
        T r[D * D];
#pragma unroll
        for (int i = 0; i < D * D; i++) mseg[i] = read_lane(m[i], 15);
#pragma unroll
        for (int q = 1; q < 4; q++) {
#pragma unroll
            for (int i = 0; i < D * D; i++) r[i] = read_lane(m[i], 16 * q + 15);
            matmul<T, D>(r, mseg, mseg);
        }
    }
    {
        T t[D];
#pragma unroll
        for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_BCAST15, 0x2>(z[i]);
        matvec_acc<T, D>(m, t, z);
#pragma unroll
        for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_BCAST15, 0x4>(z[i]);
        matvec_acc<T, D>(m, t, z);
#pragma unroll
        for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_BCAST15, 0x8>(z[i]);
        matvec_acc<T, D>(m, t, z);
    }
    T xs[D];
#pragma unroll
    for (int i = 0; i < D; i++) xs[i] = wave_shr1(z[i], xin[i]);
    // ---- pass 2: replay from the true start state -----------------------------------------------------------------------
    T part = 0;
#pragma unroll MOIHGP_GENERIC_UNROLL
    for (int k = 0; k < CK; k++) {
        const T yk = ch[k];
        const bool valid = (t0 + k) < Tlen;
        const bool miss = (yk != yk);
        T xo[D], hx;
#pragma unroll
        for (int i = 0; i < D; i++) xo[i] = xs[i];
        const T v = tick(xs, yk, miss || !valid, hx);
        if (NLL) {
            part = fma(v, v, part);              // v = 0 on missing / past-the-end ticks
            nobs += (valid && !miss) ? 1u : 0u;
        }
#pragma unroll
        for (int i = 0; i < D; i++) xs[i] = valid ? xs[i] : xo[i];
        ch[k] = xs[0];
    }
    if (NLL) acc += (double)part;
    const size_t last = Tlen - 1 - (t0 - (size_t)lane * CK);
    const int jl = (last / CK) > 63 ? 63 : (int)(last / CK);
#pragma unroll
    for (int i = 0; i < D; i++) xin[i] = read_lane(xs[i], jl);
}

// One sweep over `Tlen` ticks of one stream (`row`) by one wavefront, segment by segment.
//   SLICEMAP = false: the real sweep (filtered means to `orow` if WRITE, sum of v^2 in acc/nobs if NLL)
//   SLICEMAP = true : no replay and no outputs; xin ends as the state reached from the given start and msl as
//                     the transition matrix of the whole sweep (x_end = msl * x_start + [x_end from zero]),
//                     exact also across missing ticks.  Used by the time split below.
#ifndef MOIHGP_FILTER_PREFETCH
#define MOIHGP_FILTER_PREFETCH 1
#endif
constexpr int kPrefetch = MOIHGP_FILTER_PREFETCH;

template <typename T, int D, int CK, bool WRITE, bool NLL, bool SLICEMAP, int DBG>
__device__ inline void sweep(const T* __restrict__ row, T* __restrict__ orow, size_t Tlen, T* xin, const FastConst<T, D, CK>& c,
                             const T* __restrict__ cb, typename VecOf<T>::type* lds, int lane, double& acc, unsigned& nobs,
                             size_t& nobs_uniform, T* msl) {
    using V = typename VecOf<T>::type;
    constexpr int EPV = 16 / sizeof(T);        // elements per 16-byte vector
    constexpr int VPL = CK / EPV;              // vectors per lane per segment
    constexpr int SEG = 64 * CK;               // ticks per segment
    T* chunk = reinterpret_cast<T*>(&lds[lane * (VPL + 1)]);   // this lane's CK elements inside the tile
    const size_t nfull = Tlen / SEG;
    const size_t nseg = (Tlen + SEG - 1) / SEG;
    T mfull[D * D];                            // SLICEMAP: transition of one full NaN-free segment, AKHA^SEG = M^64
    if (SLICEMAP) {
#pragma unroll
        for (int i = 0; i < D * D; i++) { msl[i] = (i % (D + 1) == 0) ? T(1) : T(0); mfull[i] = c.sp[3 * D * D + i]; }
        matmul<T, D>(mfull, mfull, mfull);   // M^16
        matmul<T, D>(mfull, mfull, mfull);   // M^32
        matmul<T, D>(mfull, mfull, mfull);   // M^64
    }

    // Segments in flight ahead of the one being solved: kPrefetch register sets, filled round-robin.  With one set a wave has at most
    // 4 KB on its way while it computes; a cold stream (nothing of it in the Infinity Cache) then leaves the memory system short of
    // requests: 16 waves x 4 KB per CU against the ~50 KB per CU that 6.4 TB/s x ~2 us of latency ask for (a plain copy kernel with
    // one 16-byte element per thread reaches 6.4 TB/s cold on this box, tools/micro/copy_bench.hip).
    V rs[kPrefetch][VPL];
    auto load_full = [&](size_t seg, V (&r)[VPL]) {
        const T* p = row + seg * SEG + (size_t)lane * EPV;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            if (DBG & 4) r[i] = nt_load(reinterpret_cast<const V*>(p + (size_t)i * 64 * EPV));
            else r[i] = *reinterpret_cast<const V*>(p + (size_t)i * 64 * EPV);
        }
    };
    auto load_tail = [&](size_t seg, V (&r)[VPL]) {
        const size_t base = seg * SEG;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            size_t tq = base + (size_t)(i * 64 + lane) * EPV;
            T e[EPV] = {};
            if (tq < Tlen) {                                        // ld >= roundup(T, EPV): the vector is in bounds
                unpack<T>(*reinterpret_cast<const V*>(row + tq), e);
#pragma unroll
                for (int k = 0; k < EPV; k++) if (tq + k >= Tlen) e[k] = T(0);   // row padding may hold anything
            }
            r[i] = pack<T>(e);
        }
    };
    auto load_any = [&](size_t seg, V (&r)[VPL]) {
        if (seg < nfull) load_full(seg, r); else if (seg < nseg) load_tail(seg, r);
    };
#pragma unroll
    for (int q = 0; q < kPrefetch; q++) load_any((size_t)q, rs[q]);

    // one segment: its data is in r (on its way since kPrefetch segments ago); r is refilled with segment seg + kPrefetch
    auto one_segment = [&](const size_t seg, V (&r)[VPL]) {
        const size_t tbase = seg * SEG;
        const size_t t0 = tbase + (size_t)lane * CK;
        // ---- coalesced registers -> LDS -> chunk-per-lane registers ---------------------------
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            int q = i * 64 + lane;
            lds[q + q / VPL] = r[i];
        }
        wave_lds_fence();
        bool done;
        {
            T y[CK];
#pragma unroll
            for (int k = 0; k < VPL; k++) {
                V v = lds[lane * (VPL + 1) + k];
                unpack<T>(v, &y[k * EPV]);
            }
            // prefetch: in flight during the arithmetic below (and that of the kPrefetch - 1 segments after this one)
            load_any(seg + kPrefetch, r);
            if (DBG & 1) {                       // tuning probe: staging path only, no arithmetic
                done = true;
#pragma unroll
                for (int k = 0; k < CK; k++) y[k] = y[k] + xin[0];
            } else if (seg < nfull) {
                done = fast_segment<T, D, CK, NLL, false, SLICEMAP>(y, xin, c, lane, t0, Tlen, acc, nobs);
                if (done) nobs_uniform += SEG;
                if (done && SLICEMAP) matmul<T, D>(mfull, msl, msl);
            } else if (SLICEMAP) {
                done = false;                    // ragged tail: the generic path also yields its map
            } else {
                done = fast_segment<T, D, CK, NLL, true>(y, xin, c, lane, t0, Tlen, acc, nobs);
            }
            if (done && WRITE) {
#pragma unroll
                for (int k = 0; k < VPL; k++) lds[lane * (VPL + 1) + k] = pack<T>(&y[k * EPV]);
            }
        }
        if (!done) {
            if (SLICEMAP) {
                T mseg[D * D];
                generic_segment<T, D, CK, NLL>(chunk, xin, cb, lane, t0, Tlen, acc, nobs, mseg);
                matmul<T, D>(mseg, msl, msl);
            } else {
                generic_segment<T, D, CK, NLL>(chunk, xin, cb, lane, t0, Tlen, acc, nobs);
            }
        }

        // ---- LDS -> coalesced stores ----------------------------------------------------------
        if (WRITE) {
            wave_lds_fence();
            T* po = orow + tbase + (size_t)lane * EPV;
            if (seg < nfull) {
#pragma unroll
                for (int i = 0; i < VPL; i++) {
                    int q = i * 64 + lane;
                    // the filtered means are written once and not re-read by this kernel: stream them past the caches so that
                    // the input stream keeps its place in L2 / Infinity Cache (DBG bit 2: plain stores, for A/B)
                    if (DBG & 2) *reinterpret_cast<V*>(po + (size_t)i * 64 * EPV) = lds[q + q / VPL];
                    else nt_store(lds[q + q / VPL], reinterpret_cast<V*>(po + (size_t)i * 64 * EPV));
                }
            } else {
#pragma unroll
                for (int i = 0; i < VPL; i++) {
                    int q = i * 64 + lane;
                    if (tbase + (size_t)q * EPV < Tlen) nt_store(lds[q + q / VPL], reinterpret_cast<V*>(po + (size_t)i * 64 * EPV));
                }
            }
        }
        wave_lds_fence();
    };
    for (size_t seg = 0; seg < nseg; seg += kPrefetch) {
#pragma unroll
        for (int q = 0; q < kPrefetch; q++)
            if (seg + q < nseg) one_segment(seg + q, rs[q]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// TEAM path of the time split (few latents; tried first, inside the SPLIT kernel below): a wavefront keeps its slice -- up to kTeamSeg
// segments -- IN REGISTERS between the two halves of the solve, the stream is read once and nothing is computed twice:
//   A. per own segment: load, chunk responses, scan from a ZERO state -> z_j (state after chunk j had the segment started from zero); the
//      chunk y and z stay in registers; lane 63's z, the segment's zero-start end state e0, goes to LDS.            __syncthreads()
//   B. the state entering the wave's first segment by the serial recurrence over ALL segments before it,  c <- M^64 c + e0  (3 x 3 products,
//      fp64, every wave for itself: <= 32 steps); a start state enters a segment linearly, so the true state after chunk j is
//      z_j + M^(j+1) c = z_j + pj (M^16)^(j / 16) c  with the per-lane power pj = M^(j % 16 + 1) of the scan's own table; then the replay
//      from the true start states, the stores, and the next own segment starts from this one's end state.
// Against the two-pass split (slice maps, then a second sweep that re-reads the slice from L2 and repeats response and scan) this saves
// the second load / LDS transposition / response / scan of every segment: ~200 of ~700 instructions per segment.
// A missing tick anywhere in the latent's stream (chunk maps are then no powers of one matrix) sends the whole workgroup to the two-pass
// split, which is exact across gaps: returns false, workgroup-uniform, nothing written.
constexpr int kTeamSeg = 4;
constexpr int kTeamMaxSegs = 32;            // segments per stream: kMaxSplit waves x kTeamSeg

template <typename T, int D, int CK, bool WRITE, bool NLL>
__device__ inline bool team_slice(const T* __restrict__ row, T* __restrict__ orow, size_t Tlen /* this wave's slice */, size_t toff /* its first tick */,
                                  const FastConst<T, D, CK>& c, typename VecOf<T>::type* lds, int lane, double (*e0all)[D], int* dirty,
                                  const T* xstart /* the latent's start state */, T* xout /* state after the slice */,
                                  double& acc, unsigned& nobs, size_t& nobs_uniform) {
    using V = typename VecOf<T>::type;
    constexpr int EPV = 16 / sizeof(T), VPL = CK / EPV, SEG = 64 * CK;
    const int nsg = (int)((Tlen + SEG - 1) / SEG), nfull = (int)(Tlen / SEG);          // own segments (<= kTeamSeg), the full ones among them
    const int g0 = (int)(toff / SEG);                                                  // index of the first own segment in the stream
    T y[kTeamSeg][CK], z[kTeamSeg][D];
    bool bad = false;
    // ---- A ----
#pragma unroll
    for (int sg = 0; sg < kTeamSeg; sg++) {
        if (sg < nsg) {                                                                // wave-uniform
            V r[VPL];
            if (sg < nfull) {
                const T* p = row + (size_t)sg * SEG + (size_t)lane * EPV;
#pragma unroll
                for (int i = 0; i < VPL; i++) r[i] = *reinterpret_cast<const V*>(p + (size_t)i * 64 * EPV);
            } else {
#pragma unroll
                for (int i = 0; i < VPL; i++) {
                    const size_t tq = (size_t)sg * SEG + (size_t)(i * 64 + lane) * EPV;
                    T e[EPV] = {};
                    if (tq < Tlen) {                                                   // ld >= roundup(T, EPV): the vector is in bounds
                        unpack<T>(*reinterpret_cast<const V*>(row + tq), e);
#pragma unroll
                        for (int k = 0; k < EPV; k++) if (tq + k >= Tlen) e[k] = T(0);
                    }
                    r[i] = pack<T>(e);
                }
            }
#pragma unroll
            for (int i = 0; i < VPL; i++) { const int q = i * 64 + lane; lds[q + q / VPL] = r[i]; }
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < VPL; k++) unpack<T>(lds[lane * (VPL + 1) + k], &y[sg][k * EPV]);
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < D; i++) z[sg][i] = T(0);
#pragma unroll
            for (int k = 0; k < CK; k++)
#pragma unroll
                for (int i = 0; i < D; i++) z[sg][i] = fma(c.g[k * D + i], y[sg][k], z[sg][i]);
#pragma unroll
            for (int i = 0; i < D; i++) bad |= (z[sg][i] != z[sg][i]);             // a NaN y poisons z
            dpp_scan<T, D>(z[sg], c.sp, c.pj);
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < D; i++) e0all[g0 + sg][i] = (double)z[sg][i];
            }
        }
    }
    if (__any(bad) && lane == 0) *dirty = 1;
    __syncthreads();
    if (*dirty) return false;
    // ---- B ----
    double cc[D];
    {
        double ms[D * D], t[D * D];                                                    // M^64 = ((M^8)^2)^2)^2, the transition of a whole segment
#pragma unroll
        for (int i = 0; i < D * D; i++) ms[i] = (double)c.sp[3 * D * D + i];
        matmul<double, D>(ms, ms, t); matmul<double, D>(t, t, ms); matmul<double, D>(ms, ms, t);
#pragma unroll
        for (int i = 0; i < D; i++) cc[i] = (double)xstart[i];
        for (int g = 0; g < g0; g++) {
            double cn[D];
#pragma unroll
            for (int i = 0; i < D; i++) {
                double a = e0all[g][i];
#pragma unroll
                for (int j = 0; j < D; j++) a = fma(t[i * D + j], cc[j], a);
                cn[i] = a;
            }
#pragma unroll
            for (int i = 0; i < D; i++) cc[i] = cn[i];
        }
    }
    T m16[D * D];                                                                      // M^16: lane 15's entry of the per-lane table
#pragma unroll
    for (int i = 0; i < D * D; i++) m16[i] = read_lane(c.pj[i], 15);
    const int rowi = lane >> 4;
#pragma unroll
    for (int sg = 0; sg < kTeamSeg; sg++) {
        if (sg < nsg) {
            T cin[D], u[D], t1[D];
#pragma unroll
            for (int i = 0; i < D; i++) { cin[i] = (T)cc[i]; u[i] = cin[i]; }
#pragma unroll
            for (int rr = 1; rr < 4; rr++) {                                           // u = (M^16)^row cin
#pragma unroll
                for (int i = 0; i < D; i++) t1[i] = T(0);
                matvec_acc<T, D>(m16, u, t1);
#pragma unroll
                for (int i = 0; i < D; i++) u[i] = rowi >= rr ? t1[i] : u[i];
            }
            matvec_acc<T, D>(c.pj, u, z[sg]);                                          // true state after chunk j: z_j + M^(j+1) cin
            T xs[D];
#pragma unroll
            for (int i = 0; i < D; i++) xs[i] = wave_shr1(z[sg][i], cin[i]);
            const size_t t0 = (size_t)sg * SEG + (size_t)lane * CK;
            int jl = 63;
            if (sg < nfull) {
                replay_chunk<T, D, CK, NLL, false>(y[sg], xs, c, t0, Tlen, acc, nobs);
                nobs_uniform += SEG;
            } else {
                replay_chunk<T, D, CK, NLL, true>(y[sg], xs, c, t0, Tlen, acc, nobs);
                jl = (int)((Tlen - 1 - (size_t)sg * SEG) / CK);
            }
#pragma unroll
            for (int i = 0; i < D; i++) cc[i] = (double)read_lane(xs[i], jl);
            if (WRITE) {
#pragma unroll
                for (int k = 0; k < VPL; k++) lds[lane * (VPL + 1) + k] = pack<T>(&y[sg][k * EPV]);
                wave_lds_fence();
                T* po = orow + (size_t)sg * SEG + (size_t)lane * EPV;
#pragma unroll
                for (int i = 0; i < VPL; i++) {
                    const int q = i * 64 + lane;
                    if (sg < nfull || (size_t)sg * SEG + (size_t)q * EPV < Tlen) nt_store(lds[q + q / VPL], reinterpret_cast<V*>(po + (size_t)i * 64 * EPV));
                }
                wave_lds_fence();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; i++) xout[i] = (T)cc[i];
    return true;
}

constexpr int kMaxSplit = 8;    // slices per latent = waves per workgroup in split mode (512 threads: 256-VGPR budget)

// ---------------------------------------------------------------------------------------------
// SPLIT = false: 4 waves per workgroup, wave w of block b owns latent 4b + w and sweeps all T ticks.
// SPLIT = true (few latents): workgroup b owns latent b; its nsplit (<= 16) waves own consecutive time slices
//   of Tslice ticks (a whole number of segments).  Every wave first computes the affine map of its slice
//   (SLICEMAP sweep, stream read #1), parks it in LDS, the workgroup synchronises once, every wave folds the
//   maps of the slices before it into its start state, and then runs the real sweep of its slice (stream
//   read #2 comes from L2: the slice was touched microseconds ago by the same CU).  One launch, no global
//   synchronisation, no workspace.
template <typename T, int D, int CK, bool WRITE, bool NLL, int MINW, bool SPLIT, int DBG = 0>
__global__ void __launch_bounds__(SPLIT ? 64 * kMaxSplit : 64 * kWavesPerBlock, SPLIT ? 1 : MINW)
filter_scan_kernel(const T* __restrict__ Ty, size_t Ttot, size_t ld, size_t L, const T* __restrict__ cbT,
                   const double* __restrict__ cb64, const T* xin0 /* start state */, T* x /* end state; may be the same buffer */,
                   T* __restrict__ yhat, double* __restrict__ nll, int nsplit, size_t Tslice,
                   int nbig /* split: slices [0, nbig) hold Tslice ticks, the later ones one segment less (nbig == nsplit: all alike) */,
                   size_t ldo /* row stride of yhat (the stream's own is ld) */, int team /* split: try the team path first (slices of <= kTeamSeg segments) */) {
    using V = typename VecOf<T>::type;
    using Lay = CB<D>;
    constexpr int EPV = 16 / sizeof(T);
    constexpr int VPL = CK / EPV;
    constexpr int NVP = 64 * (VPL + 1);        // padded vectors per wave tile (one pad vector per lane row)
    static_assert(CK % EPV == 0 && CK <= 16, "CK must be a multiple of 16 bytes and fit the G table");
    // LDS: one padded tile per wave (dynamic: 4 or nsplit waves), then in split mode the per-slice carry records
    // [map (D*D), end-from-zero (D), sum v^2, n_obs] as doubles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V (*lds_all)[NVP] = reinterpret_cast<V (*)[NVP]>(smem);
    constexpr int CR = D * D + D + 2;
    double (*carry)[CR] = reinterpret_cast<double (*)[CR]>(smem + (size_t)(SPLIT ? nsplit : kWavesPerBlock) * NVP * sizeof(V));

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = SPLIT ? (size_t)blockIdx.x : (size_t)blockIdx.x * kWavesPerBlock + wave;
    if (!SPLIT && l >= L) return;
    // split: the slice lengths differ by one segment at most, the longer ones first, so that with 8 waves (two per SIMD, wave w on
    // SIMD w % 4) every SIMD carries the same number of segments: both passes are issue-bound once two waves share a SIMD
    const size_t Tsl = SPLIT ? (wave < nbig ? Tslice : Tslice - 64 * CK) : 0;
    const size_t toff = SPLIT ? (wave < nbig ? (size_t)wave * Tslice : (size_t)nbig * Tslice + (size_t)(wave - nbig) * (Tslice - 64 * CK)) : 0;
    const size_t Tlen = SPLIT ? (toff >= Ttot ? 0 : ((Ttot - toff) < Tsl ? (Ttot - toff) : Tsl)) : Ttot;
    V* lds = lds_all[wave];

    // ---- per-latent constants: wave-uniform scalar loads, plus this lane's cross-row power ----------
    const T* cb = cbT + l * Lay::SIZE;
    FastConst<T, D, CK> c;
#pragma unroll
    for (int i = 0; i < D * D; i++) c.a[i] = cb[Lay::A + i];
#pragma unroll
    for (int i = 0; i < D; i++) c.k[i] = cb[Lay::K + i];
#pragma unroll
    for (int i = 0; i < CK * D; i++) c.g[i] = cb[Lay::G + i];
    {   // park the scan powers in LDS: as registers they would be 36-72 more uniform values than the SGPR file holds
        T* spw = reinterpret_cast<T*>(smem + (size_t)(SPLIT ? nsplit : kWavesPerBlock) * NVP * sizeof(V) +
                                      (SPLIT ? (size_t)nsplit * CR * sizeof(double) : 0)) + (size_t)wave * (4 * D * D);
        if (lane < 4 * D * D) spw[lane] = cb[Lay::SP + lane];
        c.sp = spw;
        wave_lds_fence();
    }
#pragma unroll
    for (int i = 0; i < D * D; i++) c.pj[i] = cb[Lay::PJ + (lane & 15) * D * D + i];

    // An unstable latent (rho(AKHA) > 1: scan tables overflowed, flagged by IHGP::update) is left to filter_seq_kernel.
    // Uniform over the wave, and over the workgroup in split mode (one latent per workgroup), and ahead of any barrier.
    if (cb[Lay::SCANOK] == T(0)) return;
    const T* row = Ty + l * ld + toff;
    T* orow = WRITE ? yhat + l * ldo + toff : nullptr;
    T xin[D];
    double acc = 0.0;          // per-lane sum of v^2 over observed ticks
    unsigned nobs = 0;         // per-lane count of observed ticks (tail / generic segments)
    size_t nobs_uniform = 0;   // observed ticks of full fast-path segments (every lane contributes CK)

    bool team_done = false;
    if (SPLIT) {
        if (team) {
            // (behind the tiles, the carry records and the scan powers: the segments' zero-start end states and the "stream has a gap" flag)
            unsigned char* tb = smem + (size_t)nsplit * NVP * sizeof(V) + (size_t)nsplit * CR * sizeof(double) + (((size_t)nsplit * 4 * D * D * sizeof(T) + 15) & ~(size_t)15);
            double (*e0all)[D] = reinterpret_cast<double (*)[D]>(tb);
            int* dirty = reinterpret_cast<int*>(tb + (size_t)kTeamMaxSegs * D * sizeof(double));
            if (threadIdx.x == 0) *dirty = 0;
            __syncthreads();
            team_done = team_slice<T, D, CK, WRITE, NLL>(row, orow, Tlen, toff, c, lds, lane, e0all, dirty, xin0 + l * D, xin, acc, nobs, nobs_uniform);
        }
    }
    if (SPLIT && !team_done) {
        // ---- pass 1: affine map of this slice, from a zero start.  Nobody consumes the map of the LAST
        // slice, so it skips this pass (it is also the only slice that can be ragged: Tslice is a whole
        // number of segments, hence pass 1 never meets a tail).
        if (wave < nsplit - 1) {
            T msl[D * D];
#pragma unroll
            for (int i = 0; i < D; i++) xin[i] = T(0);
            sweep<T, D, CK, false, false, true, 0>(row, nullptr, Tlen, xin, c, cb, lds, lane, acc, nobs, nobs_uniform, msl);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < D * D; i++) carry[wave][i] = (double)msl[i];
#pragma unroll
                for (int i = 0; i < D; i++) carry[wave][D * D + i] = (double)xin[i];
            }
        }
        __syncthreads();
        // ---- fold the slices before this one into its start state (uniform, <= 7 small mat-vecs) ----
        double xc[D];
#pragma unroll
        for (int i = 0; i < D; i++) xc[i] = (double)xin0[l * D + i];
        for (int s = 0; s < wave; s++) {
            double xn[D];
#pragma unroll
            for (int i = 0; i < D; i++) {
                double a = carry[s][D * D + i];
#pragma unroll
                for (int j = 0; j < D; j++) a = fma(carry[s][i * D + j], xc[j], a);
                xn[i] = a;
            }
#pragma unroll
            for (int i = 0; i < D; i++) xc[i] = xn[i];
        }
#pragma unroll
        for (int i = 0; i < D; i++) xin[i] = (T)xc[i];
        acc = 0.0; nobs = 0; nobs_uniform = 0;
        __syncthreads();       // everyone has read x[l] and the maps before anything is overwritten
    } else if (!SPLIT) {
#pragma unroll
        for (int i = 0; i < D; i++) xin[i] = xin0[l * D + i];
    }

    // ---- the real sweep (split + state only: just the last slice needs it) -------------------------
    if (!team_done && (WRITE || NLL || !SPLIT || wave == nsplit - 1)) {
        T unused[D * D];
        sweep<T, D, CK, WRITE, NLL, false, DBG>(row, orow, Tlen, xin, c, cb, lds, lane, acc, nobs, nobs_uniform, unused);
    }

    // ---- epilogue: carried-out state and the latent's NLL ---------------------------------------
    if (NLL) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            acc += __shfl_xor(acc, o);
            nobs += __shfl_xor(nobs, o);
        }
    }
    const double* c64 = cb64 + l * Lay::SIZE;
    if (SPLIT) {
        if (NLL) {
            if (lane == 0) { carry[wave][D * D + D] = acc; carry[wave][D * D + D + 1] = (double)nobs_uniform + (double)nobs; }
            __syncthreads();
        }
        if (lane == 0 && wave == nsplit - 1) {
#pragma unroll
            for (int i = 0; i < D; i++) x[l * D + i] = xin[i];
        }
        if (NLL && lane == 0 && wave == 0) {
            double a = 0.0, n = 0.0;
            for (int s = 0; s < nsplit; s++) { a += carry[s][D * D + D]; n += carry[s][D * D + D + 1]; }
            nll[l] = 0.5 * (a / c64[Lay::S] + n * c64[Lay::LOGS]);
        }
    } else if (lane == 0) {
#pragma unroll
        for (int i = 0; i < D; i++) x[l * D + i] = xin[i];
        if (NLL) {
            double n = (double)nobs_uniform + (double)nobs;
            nll[l] = 0.5 * (acc / c64[Lay::S] + n * c64[Lay::LOGS]);   // sum of ihgp.h:207 terms
        }
    }
}

// The scalar the optimiser consumes (moihgp.h:684 `loss += ...`): sum of nll[] in a fixed order (thread-strided, a butterfly per
// wavefront, then the 16 wavefront sums in order), one workgroup, launched right behind the sweep.  (Fusing it into the sweep as a last-arrival reduction was tried: the 4096
// device-scope atomics on one counter, and any agent-scope fence -- a whole-L2 write-back per wavefront, since the XCDs do not
// share an L2 -- cost far more than this launch: 136-205 us against 60.)
__global__ void __launch_bounds__(1024) nll_total_kernel(const double* __restrict__ nll, size_t L, double* __restrict__ total) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (size_t i = tid; i < L; i += 1024) s += nll[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; w++) t += red[w];
        *total = t;
    }
}

// Exact sequential filter for the latents flagged unstable (SCANOK == 0): one lane per latent, tick by tick in innovation
// form (identical in meaning to ihgp.h:81-93 / :204-209, missing ticks included), stream fetched in 16-byte vectors.
// Launched only when IHGP::update reported such latents; every other lane exits at once.
template <typename T, int D, bool TILED = false>
__global__ void __launch_bounds__(64)
filter_seq_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64,
                  const T* xin0, T* x, T* __restrict__ yhat, double* __restrict__ nll, size_t ldo) {
    using V = typename VecOf<T>::type;
    using Lay = CB<D>;
    constexpr int EPV = 16 / sizeof(T);
    constexpr size_t SEGT = 4096 / sizeof(T);                    // ticks per tile of the segment-major layout
    // element t of latent l: series-major l ld + t; segment-major ((t / SEGT) L + l) SEGT + t % SEGT
    auto at = [&](size_t l_, size_t t_, size_t ld_) { return TILED ? ((t_ / SEGT) * L + l_) * SEGT + t_ % SEGT : l_ * ld_ + t_; };
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const T* cb = cbT + l * Lay::SIZE;
    if (cb[Lay::SCANOK] != T(0)) return;
    T a[D * D], kk[D], xs[D];
    for (int i = 0; i < D * D; i++) a[i] = cb[Lay::A + i];
    for (int i = 0; i < D; i++) { kk[i] = cb[Lay::K + i]; xs[i] = xin0[l * D + i]; }
    double acc = 0.0, n = 0.0;
    for (size_t tb = 0; tb < Tlen; tb += EPV) {
        T yv[EPV];
        unpack<T>(*reinterpret_cast<const V*>(Ty + at(l, tb, ld)), yv);
        for (int e = 0; e < EPV && tb + e < Tlen; e++) {
            const T yk = yv[e];
            const bool miss = (yk != yk);
            T hx = 0;
            for (int q = 0; q < D; q++) hx = fma(a[q], xs[q], hx);
            const T v = miss ? T(0) : yk - hx;
            if (!miss) { acc = fma((double)v, (double)v, acc); n += 1.0; }
            T xn[D];
            xn[0] = fma(kk[0], v, hx);
            for (int i = 1; i < D; i++) {
                T s = kk[i] * v;
                for (int q = 0; q < D; q++) s = fma(a[i * D + q], xs[q], s);
                xn[i] = s;
            }
            for (int i = 0; i < D; i++) xs[i] = xn[i];
            if (yhat) yhat[at(l, tb + e, ldo)] = xs[0];
        }
    }
    for (int i = 0; i < D; i++) x[l * D + i] = xs[i];
    if (nll) { const double* c64 = cb64 + l * Lay::SIZE; nll[l] = 0.5 * (acc / c64[Lay::S] + n * c64[Lay::LOGS]); }
}

// ===========================================================================================================================
// LDS-DMA form of the many-latent sweep (round 4).  Same mapping and the same segment solve as above (one wavefront = one latent,
// fast_segment / generic_segment unchanged); what changes is how the stream reaches the lanes:
//   * the stream is fetched by `global_load_lds_dwordx4` (gfx950 LDS-DMA): no destination VGPRs, so the bytes a wave keeps in flight
//     are bounded by its LDS ring, not by the 128-register budget of four waves per SIMD (the register-staged kernel above has room
//     for ONE 4 KB segment in flight per wave);
//   * a wave owns a ring of NP pieces of 1 KB (one piece = one DMA instruction = the 16 chunks of one DPP row of a segment; a
//     segment is 4 pieces in both precisions).  Segment s lives in slots (4 s + i) mod NP; while it is solved the NP - 4 pieces behind
//     it are on their way, and its own four slots are refilled with pieces 4 s + NP + i as soon as its results have left them;
//   * an LDS-DMA instruction writes LDS linearly (wave-uniform base + 16 lane), so the chunk-per-lane layout is put on the SOURCE
//     address: lane ln of a piece fetches vector (ln & 3) ^ sw(ln >> 2) of chunk ln >> 2, sw(jj) = (jj ^ (jj >> 2)) & 3 -- every piece
//     still reads 1 KB of contiguous stream, permuted inside its 64-byte chunks -- and lane j finds vector k of its chunk at
//     16 (k ^ sw(j & 15)) inside its 64 bytes: conflict-free for ds_read_b128 (each of its four 16-lane groups covers all 64 banks)
//     and for ds_write_b128 (eight-lane groups, 32 banks).  No padding: 4 KB per segment instead of 5;
//   * results go back through the segment's own slots (ds_write_b128 swizzled, ds_read_b128 linear) and leave by streaming stores
//     with the same per-lane permutation, 1 KB contiguous per instruction;
//   * every LDS access of the hot path is inline asm: the compiler does not track which DMA a ds_read depends on and would drain ALL
//     DMAs in flight (s_waitcnt vmcnt(0)) before any LDS read it can see.  The waits are counted by hand: vector-memory operations of a
//     wave complete in issue order, and behind the last piece of segment s there are always exactly NP - 4 younger DMAs and (from
//     segment 1 on) the 4 stores of segment s - 1.  Pieces past the end of the stream are still issued -- every lane re-reads the last
//     valid vector, one cache line per instruction -- so that the count holds to the last segment; the ring slots they fill are never read.
#ifndef MOIHGP_FILTER_DMA
#define MOIHGP_FILTER_DMA 1
#endif
constexpr int kDmaRing32 = 8, kDmaWaves32 = 4;     // fp32: 8 KB ring, 4 waves per SIMD (16 x 8.2 KB = 131 KB of the 160 KB per CU)
constexpr int kDmaRing64 = 8, kDmaWaves64 = 3;     // fp64: 132 VGPRs, 3 waves per SIMD (12 x 8.3 KB)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;
__device__ inline unsigned lds_addr_of(const void* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; }
template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T> struct AsmVec;
template <> struct AsmVec<float> { using type = nt_f4; };
template <> struct AsmVec<double> { using type = nt_d2; };
__device__ inline void av_unpack(const nt_f4& v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ inline void av_unpack(const nt_d2& v, double* o) { o[0] = v.x; o[1] = v.y; }
__device__ inline nt_f4 av_pack(const float* i) { nt_f4 v = {i[0], i[1], i[2], i[3]}; return v; }
__device__ inline nt_d2 av_pack(const double* i) { nt_d2 v = {i[0], i[1]}; return v; }

// four 16-byte LDS reads / writes at four per-lane addresses, complete on return
template <typename AV>
__device__ inline void lds_read4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, AV (&r)[4]) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
template <typename AV>
__device__ inline void lds_write4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, const AV (&r)[4]) {
    asm volatile("ds_write_b128 %4, %0\n\tds_write_b128 %5, %1\n\tds_write_b128 %6, %2\n\tds_write_b128 %7, %3"
                 :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
// N scalars of a wave-uniform table row (16-byte aligned, padded to whole vectors) from LDS: every lane reads the same address (broadcast)
template <typename T, int N>
__device__ inline void lds_read_uniform(unsigned addr, T* out) {
    using AV = typename AsmVec<T>::type;
    constexpr int EPV = 16 / sizeof(T), NV = (N + EPV - 1) / EPV;
    static_assert(NV >= 1 && NV <= 5, "table row of 1..5 vectors");
    AV v[5];
    if constexpr (NV == 1) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v[0]) : "v"(addr) : "memory");
    else if constexpr (NV == 2) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v[0]), "=&v"(v[1]) : "v"(addr) : "memory");
    else if constexpr (NV == 3) asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(addr) : "memory");
    else if constexpr (NV == 4) asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr) : "memory");
    else asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\tds_read_b128 %3, %5 offset:48\n\tds_read_b128 %4, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
                      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]) : "v"(addr) : "memory");
    T e[5 * EPV];
#pragma unroll
    for (int i = 0; i < NV; i++) av_unpack(v[i], &e[i * EPV]);
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = e[i];
}

// dpp_scan (kernels_common.h) with the in-row powers M^(1,2,4,8) fetched from the wave's LDS table by the reads above
template <typename T, int D, int SPL /* bytes per level */>
__device__ inline void dpp_scan_dma(T* z, unsigned sp_addr, const T* pj) {
    T t[D], m[D * D];
    lds_read_uniform<T, D * D>(sp_addr + 0 * SPL, m);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_SHR + 1, 0xF>(z[i]);
    matvec_acc<T, D>(m, t, z);
    lds_read_uniform<T, D * D>(sp_addr + 1 * SPL, m);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_SHR + 2, 0xF>(z[i]);
    matvec_acc<T, D>(m, t, z);
    lds_read_uniform<T, D * D>(sp_addr + 2 * SPL, m);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_SHR + 4, 0xF>(z[i]);
    matvec_acc<T, D>(m, t, z);
    lds_read_uniform<T, D * D>(sp_addr + 3 * SPL, m);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_SHR + 8, 0xF>(z[i]);
    matvec_acc<T, D>(m, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_BCAST15, 0x2>(z[i]);
    matvec_acc<T, D>(pj, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_BCAST15, 0x4>(z[i]);
    matvec_acc<T, D>(pj, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_BCAST15, 0x8>(z[i]);
    matvec_acc<T, D>(pj, t, z);
}

// Per-latent constants of the DMA kernel: as FastConst, the scan powers by LDS address
template <typename T, int D, int CK>
struct DmaConst {
    T a[D * D], k[D], g[CK * D];
    unsigned sp_addr;
    T pj[D * D];
};

// the segment solve of fast_segment() on top of dpp_scan_dma.  TAIL: 0 = full segment; 1 = ragged, the stream ends on a chunk boundary (whole lanes
// are valid or not: no per-tick mask, lanes past the end hold zeros and are simply not counted); 2 = ragged, any length (replay masked per tick)
template <typename T, int D, int CK, int SPL, bool NLL, int TAIL>
__device__ inline bool dma_segment(T* y, T* xin, const DmaConst<T, D, CK>& c, int lane, size_t t0, size_t Tlen, double& acc, unsigned& nobs) {
    T z[D];
#pragma unroll
    for (int i = 0; i < D; i++) z[i] = T(0);
#pragma unroll
    for (int k = 0; k < CK; k++)
#pragma unroll
        for (int i = 0; i < D; i++) z[i] = fma(c.g[k * D + i], y[k], z[i]);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < D; i++) bad |= (z[i] != z[i]);
    if (__any(bad)) return false;
    {
        T x0[D], m[D * D];
        lds_read_uniform<T, D * D>(c.sp_addr, m);      // M = M^1 (level 0 of the table)
#pragma unroll
        for (int i = 0; i < D; i++) x0[i] = (lane == 0) ? xin[i] : T(0);
        matvec_acc<T, D>(m, x0, z);
    }
    dpp_scan_dma<T, D, SPL>(z, c.sp_addr, c.pj);
    T xs[D];
#pragma unroll
    for (int i = 0; i < D; i++) xs[i] = wave_shr1(z[i], xin[i]);
    FastConst<T, D, CK> fc;                            // replay_chunk reads a, k only (references: no copies survive inlining)
#pragma unroll
    for (int i = 0; i < D * D; i++) fc.a[i] = c.a[i];
#pragma unroll
    for (int i = 0; i < D; i++) fc.k[i] = c.k[i];
    int jl = 63;
    if (TAIL == 1) {
        // whole lanes: lane j is valid iff t0 < Tlen; invalid lanes replay zeros from whatever state the scan gave them -- nobody reads them
        double part = 0.0; unsigned none = 0;
        replay_chunk<T, D, CK, NLL, false>(y, xs, fc, t0, Tlen, part, none);
        const bool valid = t0 < Tlen;
        if (NLL) { acc += valid ? part : 0.0; nobs += valid ? (unsigned)CK : 0u; }
        jl = (int)((Tlen - 1 - (t0 - (size_t)lane * CK)) / CK);
    } else if (TAIL == 2) {
        replay_chunk<T, D, CK, NLL, true>(y, xs, fc, t0, Tlen, acc, nobs);
        jl = (int)((Tlen - 1 - (t0 - (size_t)lane * CK)) / CK);
    } else {
        replay_chunk<T, D, CK, NLL, false>(y, xs, fc, t0, Tlen, acc, nobs);
    }
#pragma unroll
    for (int i = 0; i < D; i++) xin[i] = read_lane(xs[i], jl);
    return true;
}

// TILED: the streams are laid out SEGMENT-MAJOR, [ceil(T / SEG)][L][SEG] with SEG = 4 KB of ticks (1024 fp32 / 512 fp64) -- segment s of every
// latent side by side -- instead of series-major [L][ld].  The wavefronts of a launch move through the segments roughly together, so with this
// layout the chip reads and writes ONE contiguous front, like a plain copy, instead of 4096 row streams 40 KB apart: measured with a copy
// kernel of the sweep's access pattern (tools/micro/rows_copy.hip, profiles/r04/rows_copy_access_pattern.log) a cold stream moves at
// 5.7-6.05 TB/s segment-major against 5.0-5.25 TB/s series-major.  `ld` / `ldo` are ignored; the last tile is allocated whole.
template <typename T, int D, int CK, bool WRITE, bool NLL, int NP, int MINW, bool TILED = false>
__global__ void __launch_bounds__(64 * kWavesPerBlock, MINW)
filter_dma_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64,
                  const T* xin0, T* x, T* __restrict__ yhat, double* __restrict__ nll, size_t ldo) {
    using Lay = CB<D>;
    using AV = typename AsmVec<T>::type;
    constexpr int EPV = 16 / sizeof(T);
    constexpr int VPL = CK / EPV;
    static_assert(VPL == 4, "a chunk is four 16-byte vectors in both precisions");
    constexpr int SEG = 64 * CK;                               // ticks per segment
    constexpr int PT = 16 * CK;                                // ticks per piece (1 KB)
    constexpr int SPL = ((D * D * (int)sizeof(T)) + 15) / 16 * 16;   // bytes per level of the LDS power table
    constexpr int WAVE_LDS = NP * 1024 + 4 * SPL;
    static_assert(NP >= 8 && NP <= 40, "ring of 8 .. 40 pieces");
    constexpr int WSTEADY = NP - 4 + (WRITE ? 4 : 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = (size_t)blockIdx.x * kWavesPerBlock + wave;
    if (l >= L) return;
    unsigned char* ring = smem + (size_t)wave * WAVE_LDS;
    const unsigned ring_addr = lds_addr_of(ring);
    const unsigned sp_addr = ring_addr + NP * 1024;
    const T* cb = cbT + l * Lay::SIZE;

    if (Tlen == 0) {                                           // nothing to sweep: the state passes through, no NLL term
        if (lane == 0 && cb[Lay::SCANOK] != T(0)) {
#pragma unroll
            for (int i = 0; i < D; i++) x[l * D + i] = xin0[l * D + i];
            if (NLL) nll[l] = 0.0;
        }
        return;
    }
    // ---- the stream first: prologue, the whole ring -----------------------------------------------------------------------------------
    const size_t nseg = (Tlen + SEG - 1) / SEG;
    const size_t nfullp = Tlen / PT;                           // pieces that lie entirely inside the stream
    const unsigned jj = lane & 15, rr = lane >> 4;
    const unsigned sw = (jj ^ (jj >> 2)) & 3;
    // byte offset of this lane's vector inside a piece, as fetched / stored (the swizzle lives on the global side)
    const unsigned goff = (unsigned)(((lane & ~3) | ((lane & 3) ^ (((lane >> 2) ^ (lane >> 4)) & 3))) * 16);
    // series-major: this latent's row; segment-major: this latent's tile of segment 0, the tiles of later segments L * 4 KB apart
    const unsigned char* rowb = TILED ? reinterpret_cast<const unsigned char*>(Ty) + l * 4096 : reinterpret_cast<const unsigned char*>(Ty + l * ld);
    const size_t seg_stride = TILED ? L * 4096 : 4096;           // bytes from one segment of this latent to the next
    // last vector of the row that starts inside the stream (ld >= roundup(T, EPV): it is in bounds); pieces past the end re-read it
    const size_t lastv = ((Tlen - 1) / EPV) * 16;
    auto issue_piece = [&](size_t p, unsigned slot, bool inside) {
        lds_void_t* dst = (lds_void_t*)(ring + (size_t)slot * 1024);
        if (TILED) {
            // (the last tile is allocated whole: no clamping inside the stream's segments; pieces past the last segment re-read its last piece)
            const size_t pp = p < 4 * nseg ? p : 4 * nseg - 1;
            __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(rowb + (pp >> 2) * seg_stride + (pp & 3) * 1024 + goff), dst, 16, 0, 0);
        } else if (inside) {
            __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(rowb + p * 1024 + goff), dst, 16, 0, 0);
        } else {
            size_t o = p * 1024 + goff;
            o = o < lastv ? o : lastv;
            __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(rowb + o), dst, 16, 0, 0);
        }
    };
    if (nfullp >= (size_t)NP) {
#pragma unroll
        for (int i = 0; i < NP; i++) issue_piece((size_t)i, (unsigned)i, true);
    } else {
#pragma unroll
        for (int i = 0; i < NP; i++) issue_piece((size_t)i, (unsigned)i, false);
    }
    // ---- per-lane constants: plain loads, issued behind the ring.  The compiler waits for them with vmcnt(0) at their first use (it does
    // not count past LDS-DMA operations): once per sweep, inside segment 0, which needs the head of the ring anyway ------------------------
    DmaConst<T, D, CK> c;
#pragma unroll
    for (int i = 0; i < D * D; i++) c.pj[i] = cb[Lay::PJ + (lane & 15) * D * D + i];
    const T spv = cb[Lay::SP + (lane < 4 * D * D ? lane : 0)];   // lane e < 4 D D: entry e of the power table, on its way to LDS

    // ---- wave-uniform constants (scalar loads: their own counter) -------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < D * D; i++) c.a[i] = cb[Lay::A + i];
#pragma unroll
    for (int i = 0; i < D; i++) c.k[i] = cb[Lay::K + i];
#pragma unroll
    for (int i = 0; i < CK * D; i++) c.g[i] = cb[Lay::G + i];
    c.sp_addr = sp_addr;
    const T scanok = cb[Lay::SCANOK];
    T xin[D];
#pragma unroll
    for (int i = 0; i < D; i++) xin[i] = xin0[l * D + i];
    double acc = 0.0;
    unsigned nobs = 0;
    size_t nobs_uniform = 0;
    unsigned char* orowb = WRITE ? (TILED ? reinterpret_cast<unsigned char*>(yhat) + l * 4096 : reinterpret_cast<unsigned char*>(yhat + l * ldo)) : nullptr;

    unsigned slot0 = 0;
    for (size_t seg = 0; seg < nseg; seg++) {
        // ---- wait for this segment's four pieces ------------------------------------------------------------------------------------
        if (seg == 0) {
            wait_vmcnt<NP - 4>();
            // power table -> LDS, level lv at sp_addr + lv SPL
            if (lane < 4 * D * D) {
                const unsigned a = sp_addr + (unsigned)(lane / (D * D)) * SPL + (unsigned)(lane % (D * D)) * (unsigned)sizeof(T);
                if constexpr (sizeof(T) == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(spv) : "memory");
                else asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(spv) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // An unstable latent (rho(AKHA) > 1: scan tables overflowed, flagged by IHGP::update) is left to filter_seq_kernel
            if (scanok == T(0)) break;
        } else {
            wait_vmcnt<WSTEADY>();
        }
        const size_t tbase = seg * SEG;
        const size_t t0 = tbase + (size_t)lane * CK;
        // this lane's 64 bytes: row rr of the segment = piece 4 seg + rr, in slot (slot0 + rr) mod NP
        unsigned sl = slot0 + rr; sl = sl >= (unsigned)NP ? sl - NP : sl;
        const unsigned cbase = ring_addr + sl * 1024 + jj * 64;
        const unsigned ca0 = cbase + (sw << 4), ca1 = cbase + ((sw ^ 1u) << 4), ca2 = cbase + ((sw ^ 2u) << 4), ca3 = cbase + ((sw ^ 3u) << 4);   // vector k at 16 (k ^ sw)
        bool done;
        {
            AV v[4];
            lds_read4<AV>(ca0, ca1, ca2, ca3, v);
            T y[CK];
#pragma unroll
            for (int k = 0; k < 4; k++) av_unpack(v[k], &y[k * EPV]);
            const bool full = tbase + SEG <= Tlen;
            if (full) {
                done = dma_segment<T, D, CK, SPL, NLL, 0>(y, xin, c, lane, t0, Tlen, acc, nobs);
                if (done) nobs_uniform += SEG;
            } else {
                // ticks past the end: clamped pieces hold copies of the last vector, the row's padding anything
#pragma unroll
                for (int k = 0; k < CK; k++) y[k] = (t0 + k) < Tlen ? y[k] : T(0);
                if (Tlen % CK == 0) done = dma_segment<T, D, CK, SPL, NLL, 1>(y, xin, c, lane, t0, Tlen, acc, nobs);
                else done = dma_segment<T, D, CK, SPL, NLL, 2>(y, xin, c, lane, t0, Tlen, acc, nobs);
            }
            if (!done) {
                // missing data: the chunk goes back to the lane's 64 bytes in natural order for generic_segment (whose LDS accesses the
                // compiler sees and orders behind every DMA in flight: correct, and this path is bound by its arithmetic)
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = av_pack(&y[k * EPV]);
                lds_write4<AV>(cbase, cbase + 16, cbase + 32, cbase + 48, v);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wave_lds_fence();
                T* chunk = reinterpret_cast<T*>(ring + (size_t)sl * 1024 + jj * 64);
                generic_segment<T, D, CK, NLL>(chunk, xin, cb, lane, t0, Tlen, acc, nobs);
                wave_lds_fence();
                if (WRITE) {
                    lds_read4<AV>(cbase, cbase + 16, cbase + 32, cbase + 48, v);
#pragma unroll
                    for (int k = 0; k < 4; k++) av_unpack(v[k], &y[k * EPV]);
                }
            }
            if (WRITE) {
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = av_pack(&y[k * EPV]);
                lds_write4<AV>(ca0, ca1, ca2, ca3, v);
            }
        }
        // ---- results: LDS (linear per piece) -> streaming stores, 1 KB contiguous per instruction -----------------------------------
        unsigned s1 = slot0 + 1, s2 = slot0 + 2, s3 = slot0 + 3;
        s1 = s1 >= (unsigned)NP ? s1 - NP : s1; s2 = s2 >= (unsigned)NP ? s2 - NP : s2; s3 = s3 >= (unsigned)NP ? s3 - NP : s3;
        if (WRITE) {
            AV o[4];
            const unsigned la = ring_addr + (unsigned)lane * 16;
            lds_read4<AV>(la + slot0 * 1024, la + s1 * 1024, la + s2 * 1024, la + s3 * 1024, o);
            unsigned char* po = orowb + seg * seg_stride + goff;
            if (TILED || tbase + SEG <= Tlen) {                  // (a tile is stored whole: its padding belongs to the stream)
#pragma unroll
                for (int i = 0; i < 4; i++) __builtin_nontemporal_store(o[i], reinterpret_cast<AV*>(po + i * 1024));
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (tbase * sizeof(T) + i * 1024 + goff < Tlen * sizeof(T)) __builtin_nontemporal_store(o[i], reinterpret_cast<AV*>(po + i * 1024));
            }
        }
        // ---- refill the four slots with the pieces NP ahead (only while another segment will wait on the count) ---------------------
        if (seg + 1 < nseg) {
            const size_t pn = 4 * seg + NP;
            if (pn + 4 <= nfullp) { issue_piece(pn + 0, slot0, true); issue_piece(pn + 1, s1, true); issue_piece(pn + 2, s2, true); issue_piece(pn + 3, s3, true); }
            else { issue_piece(pn + 0, slot0, false); issue_piece(pn + 1, s1, false); issue_piece(pn + 2, s2, false); issue_piece(pn + 3, s3, false); }
        }
        slot0 += 4; slot0 = slot0 >= (unsigned)NP ? slot0 - NP : slot0;
    }
    wait_vmcnt<0>();                                           // no DMA may land in LDS that the next workgroup already owns

    if (scanok == T(0)) return;
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
            const double n = (double)nobs_uniform + (double)nobs;
            nll[l] = 0.5 * (acc / c64[Lay::S] + n * c64[Lay::LOGS]);   // sum of ihgp.h:207 terms
        }
    }
}

template <typename T, int D, int CK, int NP, int MINW, bool TILED = false>
int launch_filter_dma_t(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, const void* xin, void* x,
                        void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, int n_unstable, double* total, size_t ldo) {
    constexpr int SPL = ((D * D * (int)sizeof(T)) + 15) / 16 * 16;
    constexpr size_t smem = (size_t)kWavesPerBlock * (NP * 1024 + 4 * SPL);
    dim3 block(64 * kWavesPerBlock), grid((unsigned)((L + kWavesPerBlock - 1) / kWavesPerBlock));
    const T* ty = static_cast<const T*>(Ty);
    const T* xi = static_cast<const T*>(xin);
    T* xs = static_cast<T*>(x);
    T* yh = static_cast<T*>(yhat);
#define MOIHGP_DMA_LAUNCH(W_, N_)                                                                                                     \
    do {                                                                                                                              \
        auto kfn = filter_dma_kernel<T, D, CK, W_, N_, NP, MINW, TILED>;                                                              \
        if (smem > 65536) {                                                                                                           \
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (attr != hipSuccess) { set_last_error("filter_dma_kernel: %zu bytes of LDS refused: %s", smem, hipGetErrorString(attr)); return 2; } \
        }                                                                                                                             \
        hipExtLaunchKernelGGL(kfn, grid, block, smem, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, ldo);         \
    } while (0)
    if (yhat && nll) MOIHGP_DMA_LAUNCH(true, true);
    else if (yhat) MOIHGP_DMA_LAUNCH(true, false);
    else if (nll) MOIHGP_DMA_LAUNCH(false, true);
    else MOIHGP_DMA_LAUNCH(false, false);
#undef MOIHGP_DMA_LAUNCH
    if (n_unstable > 0)
        hipLaunchKernelGGL((filter_seq_kernel<T, D, TILED>), dim3((unsigned)((L + 63) / 64)), dim3(64), 0, stream, ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, ldo);
    if (total && nll) hipLaunchKernelGGL(nll_total_kernel, dim3(1), dim3(1024), 0, stream, nll, L, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("filter_dma_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

template <typename T, int D, int CK, int MINW, bool SPLIT>
int launch_filter_t(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, const void* xin, void* x,
                    void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, int nsplit, size_t Tslice, int n_unstable,
                    double* total, int nbig, size_t ldo, int team_ok = 1) {
    dim3 block(SPLIT ? 64 * nsplit : 64 * kWavesPerBlock);
    dim3 grid(SPLIT ? (unsigned)L : (unsigned)((L + kWavesPerBlock - 1) / kWavesPerBlock));
    constexpr size_t tile = 64 * (CK / (16 / sizeof(T)) + 1) * 16;                    // padded LDS tile per wave
    const size_t smem = (SPLIT ? (size_t)nsplit * (tile + (D * D + D + 2) * sizeof(double)) : (size_t)kWavesPerBlock * tile) +
                        (size_t)(SPLIT ? nsplit : kWavesPerBlock) * 4 * D * D * sizeof(T) +
                        (SPLIT ? (size_t)kTeamMaxSegs * D * sizeof(double) + 32 : 0);             // (team path: zero-start end states of the segments, flag)
    // team path of the split: every slice fits kTeamSeg segments kept in registers
    const int team = (SPLIT && team_ok && Tslice <= (size_t)kTeamSeg * 64 * CK && (Tlen + 64 * CK - 1) / (64 * CK) <= (size_t)kTeamMaxSegs) ? 1 : 0;
    const T* ty = static_cast<const T*>(Ty);
    const T* xi = static_cast<const T*>(xin);
    T* xs = static_cast<T*>(x);
    T* yh = static_cast<T*>(yhat);
    // hipExtLaunchKernelGGL attaches the (optional) events to the dispatch itself: kernel-exact timing
    if (yhat && nll)
        hipExtLaunchKernelGGL((filter_scan_kernel<T, D, CK, true, true, MINW, SPLIT>), grid, block, smem, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, nsplit, Tslice, nbig, ldo, team);
    else if (yhat)
        hipExtLaunchKernelGGL((filter_scan_kernel<T, D, CK, true, false, MINW, SPLIT>), grid, block, smem, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, nsplit, Tslice, nbig, ldo, team);
    else if (nll)
        hipExtLaunchKernelGGL((filter_scan_kernel<T, D, CK, false, true, MINW, SPLIT>), grid, block, smem, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, nsplit, Tslice, nbig, ldo, team);
    else
        hipExtLaunchKernelGGL((filter_scan_kernel<T, D, CK, false, false, MINW, SPLIT>), grid, block, smem, stream, ev0, ev1, 0, ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, nsplit, Tslice, nbig, ldo, team);
    if (n_unstable > 0)
        hipLaunchKernelGGL((filter_seq_kernel<T, D>), dim3((unsigned)((L + 63) / 64)), dim3(64), 0, stream, ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, ldo);
    if (total && nll) hipLaunchKernelGGL(nll_total_kernel, dim3(1), dim3(1024), 0, stream, nll, L, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("filter_scan_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

void launch_nll_total(const double* nll, size_t L, double* total, hipStream_t stream) {
    hipLaunchKernelGGL(nll_total_kernel, dim3(1), dim3(1024), 0, stream, nll, L, total);
}

// Slices per latent for the time split: enough wavefronts to occupy the chip when L is small, each slice a whole
// number of segments so that only a latent's last slice is ragged; at most kMaxSplit slices (one workgroup).
void filter_split_plan(int dtype, size_t T, size_t L, int* nsplit, size_t* Tslice, int* nbig) {
    const size_t seg = 64 * (size_t)(dtype == 0 ? kChunk64 : kChunk32);
    *nsplit = 1; *Tslice = T; *nbig = 1;
    // One workgroup per latent, `want` wavefronts each, and all of them resident at once: the fp64 kernel (171 VGPRs) fits 8 wavefronts
    // per CU, i.e. one workgroup of 8 (up to 256 latents) or two of 4 (up to 512).  Beyond that the workgroups come in two rounds, and
    // with 2 or 3 slices the two passes of the split cost as much as the one pass without it.  Measured (tools/micro/split_threshold.py,
    // T = 10^4, fp64 / fp32 kernel time in us; the plan before this rule in brackets):
    //    L = 384: 24.0 / 15.7 with 4 slices  (6 slices: 32.9 / 15.9;  no split: 30.1 / 17.7)
    //    L = 768: 32.4 / 19.1 without split  (3 slices: 42.6 / 20.8)        L = 1023: 33.8 / 19.3  (2 slices: 53.0 / 22.0)
    if (L == 0 || L > 512 || T < 2 * seg) return;
    size_t want = (2048 + L - 1) / L;                      // aim for >= 2048 wavefronts
    if (want > (size_t)kMaxSplit) want = kMaxSplit;
    if (L > 256 && want > 4) want = 4;
    const size_t segs = (T + seg - 1) / seg;
    if (want > segs) want = segs;
    if (want < 2) return;
    // `want` slices of per or per - 1 segments, the longer ones first (20 segments over 8 waves: 3 3 3 3 2 2 2 2, i.e. 5 segments on
    // each of the four SIMDs; equal slices of 3 would put 6 on three of them)
    const size_t per = (segs + want - 1) / want;
    const size_t big = segs - want * (per - 1);              // slices that hold `per` segments; 1 <= big <= want
    if (per == 1 && big < want) { *nsplit = (int)big; *Tslice = seg; *nbig = (int)big; return; }
    *nsplit = (int)want; *Tslice = per * seg; *nbig = (int)big;
}

// series-major <-> segment-major copies (callers that hold one layout and want the other; the projection GEMM writes either directly)
template <typename T>
__global__ void __launch_bounds__(256) retile_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t L, size_t Tlen, size_t ld, int to_tiled) {
    constexpr size_t SEGT = 4096 / sizeof(T);
    const size_t nseg = (Tlen + SEGT - 1) / SEGT;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;           // element of the tiled array
    if (e >= nseg * L * SEGT) return;
    const size_t sgi = e / (L * SEGT), l = (e / SEGT) % L, t = sgi * SEGT + e % SEGT;
    if (to_tiled) dst[e] = t < Tlen ? src[l * ld + t] : T(0);
    else if (t < Tlen) dst[l * ld + t] = src[e];
}
int launch_stream_retile(int dtype, const void* src, void* dst, size_t L, size_t T, size_t ld, int to_tiled, hipStream_t stream) {
    if (L == 0 || T == 0) return 0;
    const size_t segt = dtype == 0 ? 512 : 1024, n = (T + segt - 1) / segt * L * segt;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(retile_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double*)src, (double*)dst, L, T, ld, to_tiled);
    else hipLaunchKernelGGL(retile_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)src, (float*)dst, L, T, ld, to_tiled);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("retile_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

// The sweep over SEGMENT-MAJOR streams ([ceil(T / SEG)][L][SEG], SEG = 4096 / sizeof(scalar) ticks; filter_dma_kernel TILED): the reference's
// own models (d = 2, 3), any number of latents (one wavefront per latent: meant for the many-latent shapes).
int launch_filter_stream_tiled(int d, int dtype, const void* Ty, size_t T, size_t L, const double* cb64, const float* cb32, const void* xin, void* x,
                               void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, int n_unstable, double* total, int variant) {
    if (L == 0) return 0;
#ifdef MOIHGP_TUNING
    if (variant >= 20 && variant < 30 && d == 3) {        // ring length / waves per SIMD probes, as for the series-major sweep
#define MOIHGP_DMA_PROBE_T(V_, NP_, MW_)                                                                                                \
        if (variant == V_) return dtype == 0 ? launch_filter_dma_t<double, 3, kChunk64, NP_, MW_, true>(Ty, T, 0, L, cb64, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, 0) \
                                             : launch_filter_dma_t<float, 3, kChunk32, NP_, MW_, true>(Ty, T, 0, L, cb32, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, 0)
        MOIHGP_DMA_PROBE_T(21, 9, 4); MOIHGP_DMA_PROBE_T(22, 12, 3); MOIHGP_DMA_PROBE_T(23, 16, 2); MOIHGP_DMA_PROBE_T(24, 8, 3); MOIHGP_DMA_PROBE_T(25, 12, 2);
#undef MOIHGP_DMA_PROBE_T
    }
#else
    (void)variant;
#endif
    if (dtype == 0) {
        if (d == 2) return launch_filter_dma_t<double, 2, kChunk64, kDmaRing64, kDmaWaves64, true>(Ty, T, 0, L, cb64, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, 0);
        return launch_filter_dma_t<double, 3, kChunk64, kDmaRing64, kDmaWaves64, true>(Ty, T, 0, L, cb64, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, 0);
    }
    if (d == 2) return launch_filter_dma_t<float, 2, kChunk32, kDmaRing32, kDmaWaves32, true>(Ty, T, 0, L, cb32, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, 0);
    return launch_filter_dma_t<float, 3, kChunk32, kDmaRing32, kDmaWaves32, true>(Ty, T, 0, L, cb32, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, 0);
}

int launch_filter_stream(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64,
                         const float* cb32, const void* xin, void* x, void* yhat, double* nll, hipStream_t stream, int variant,
                         hipEvent_t ev0, hipEvent_t ev1, int nsplit, size_t Tslice, int n_unstable, double* total, int nbig, size_t ldo) {
    if (L == 0) return 0;
    if (ldo == 0) ldo = ld;
    if (nsplit > kMaxSplit) { set_last_error("nsplit > %d", kMaxSplit); return 1; }
    if (nbig <= 0 || nbig > nsplit) nbig = nsplit;
#define MOIHGP_FILTER_CASE(TT, DD, CKK, MW, CB)                                                                                   \
    do {                                                                                                                          \
        if (nsplit > 1) return launch_filter_t<TT, DD, CKK, 1, true>(Ty, T, ld, L, CB, cb64, xin, x, yhat, nll, stream, ev0, ev1, nsplit, Tslice, n_unstable, total, nbig, ldo); \
        return launch_filter_t<TT, DD, CKK, MW, false>(Ty, T, ld, L, CB, cb64, xin, x, yhat, nll, stream, ev0, ev1, 1, T, n_unstable, total, 1, ldo);        \
    } while (0)
    // register caps: fp32 <= 128 VGPRs (4 waves/SIMD: all 4096 wavefronts of a 4096-latent shard resident),
    // fp64 uncapped (188 VGPRs, 2 waves/SIMD: capping it to 168 spills and is 35 % slower)
#ifdef MOIHGP_TUNING
    // Tuning probes (make TUNING=1 -> lib/libmoihgp_tuning.so; tools/kbench.py): other register caps (1), plain stores (2), nontemporal
    // loads (4), both (6), and the staging-only kernel (9), which does NO arithmetic.  None of them is part of the shipped library.
    if (variant == 1 && dtype == 0 && d == 3) MOIHGP_FILTER_CASE(double, 3, kChunk64, 3, cb64);
    if (dtype == 1 && d == 3 && (variant == 2 || variant == 4 || variant == 6 || variant == 9)) {
        dim3 block(64 * kWavesPerBlock), grid((unsigned)((L + kWavesPerBlock - 1) / kWavesPerBlock));
        const size_t sm = (size_t)kWavesPerBlock * (64 * 5 * 16 + 36 * 4);
#define MOIHGP_PROBE(MW, DBG_) hipExtLaunchKernelGGL((filter_scan_kernel<float, 3, kChunk32, true, true, MW, false, DBG_>), grid, block, sm, stream, ev0, ev1, 0, \
                                                     (const float*)Ty, T, ld, L, cb32, cb64, (const float*)xin, (float*)x, (float*)yhat, nll, 1, T, 1, ldo, 0)
        if (variant == 2) MOIHGP_PROBE(4, 2);
        else if (variant == 4) MOIHGP_PROBE(4, 4);
        else if (variant == 6) MOIHGP_PROBE(4, 6);
        else MOIHGP_PROBE(1, 1);
#undef MOIHGP_PROBE
        return 0;
    }
    if (variant == 1 && dtype == 1 && d == 3) MOIHGP_FILTER_CASE(float, 3, kChunk32, 1, cb32);
    // LDS-DMA kernel, ring length / waves per SIMD probes (d = 3): 20 + k
    if (variant >= 20 && variant < 30 && d == 3 && nsplit == 1) {
#define MOIHGP_DMA_PROBE(V_, NP_, MW_)                                                                                                  \
        if (variant == V_) return dtype == 0 ? launch_filter_dma_t<double, 3, kChunk64, NP_, MW_>(Ty, T, ld, L, cb64, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, ldo) \
                                             : launch_filter_dma_t<float, 3, kChunk32, NP_, MW_>(Ty, T, ld, L, cb32, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, ldo)
        MOIHGP_DMA_PROBE(20, 8, 4); MOIHGP_DMA_PROBE(21, 9, 4); MOIHGP_DMA_PROBE(22, 12, 3); MOIHGP_DMA_PROBE(23, 16, 2); MOIHGP_DMA_PROBE(24, 8, 3);
        MOIHGP_DMA_PROBE(25, 12, 2); MOIHGP_DMA_PROBE(26, 8, 2); MOIHGP_DMA_PROBE(27, 20, 2);
#undef MOIHGP_DMA_PROBE
    }
    if (variant == 10) variant = 0;      // 10: the register-staged kernel whatever the default
    else if (variant == 0) variant = -1; // 0: the shipped default (below)
#else
    if (variant != 0) { set_last_error("filter variant %d: tuning probes are compiled only with -DMOIHGP_TUNING", variant); return 1; }
    variant = -1;
#endif
    // many latents (no time split): the LDS-DMA kernel
    if (variant == -1 && nsplit == 1 && MOIHGP_FILTER_DMA) {
        if (dtype == 0) {
            if (d == 2) return launch_filter_dma_t<double, 2, kChunk64, kDmaRing64, kDmaWaves64>(Ty, T, ld, L, cb64, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, ldo);
            return launch_filter_dma_t<double, 3, kChunk64, kDmaRing64, kDmaWaves64>(Ty, T, ld, L, cb64, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, ldo);
        }
        if (d == 2) return launch_filter_dma_t<float, 2, kChunk32, kDmaRing32, kDmaWaves32>(Ty, T, ld, L, cb32, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, ldo);
        return launch_filter_dma_t<float, 3, kChunk32, kDmaRing32, kDmaWaves32>(Ty, T, ld, L, cb32, cb64, xin, x, yhat, nll, stream, ev0, ev1, n_unstable, total, ldo);
    }
    if (dtype == 0) {
        if (d == 2) MOIHGP_FILTER_CASE(double, 2, kChunk64, 1, cb64);
        MOIHGP_FILTER_CASE(double, 3, kChunk64, 1, cb64);
    }
    if (d == 2) MOIHGP_FILTER_CASE(float, 2, kChunk32, 4, cb32);
    MOIHGP_FILTER_CASE(float, 3, kChunk32, 4, cb32);
#undef MOIHGP_FILTER_CASE
}

}  // namespace moihgp
