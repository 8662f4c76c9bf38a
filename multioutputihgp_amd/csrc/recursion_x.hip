// recursion_x.hip -- the filter + NLL sweep (ihgp.h:81-93, :204-209) for STACKED models, state dim D in {4, 6, 8, 9, 12}.
//
// Same contract as recursion.hip (one wavefront owns one latent; series-major stream; NaN = missing), different
// machine mapping, because a D x D matrix no longer fits the register file next to the state:
//
//   * the per-latent matrices are WAVE-UNIFORM, so they are never held one scalar per lane.  The scan and response tables
//     live in "slabs": one register holds 16 table entries, the same 16 in each of the wave's four 16-lane rows, and the
//     FMA reads entry e as a DPP broadcast operand (v_fmac_f64_dpp / v_fmac_f32_dpp row_newbcast:e) -- a uniform operand
//     at no instruction cost, without going through the scalar register file.  A 12 x 12 matrix is 9 such registers; the scan
//     powers of a segment are fetched at its start (coalesced, L2-resident), so the scan phase itself issues no load;
//   * a segment is 64 lanes x 32 ticks.  z_j = sum_k g_k y_k (chunk response, g streamed in per segment), then a 6-level
//     Kogge-Stone scan x_j = M x_{j-1} + z_j with the uniform powers M^(1,2,4,8,16,32) (lane shifts by ds_bpermute), then
//     every lane replays its 32 ticks from its start state in innovation form,  v = y - HA x;  x <- A x + K v
//     (== AKHA x + K y, ihgp.h:90), which only touches the J diagonal blocks of A: those sit in SGPRs for the whole replay,
//     HA and K in two slabs, so the replay loop has no memory access besides one LDS read and write per tick;
//   * a segment that holds a NaN (or a latent whose scan tables overflowed, rho(AKHA) > 1) is run tick by tick with the
//     rows of AKHA spread over the lanes (lane i owns row i, lane D owns HA; the state is gathered by v_readlane).
//
// Roofline: VALU-bound for D >= 9 in fp64 (per tick about D*DB + 3D FMA replay + 7 D^2 / 32 scan + D response
// against 16 B of traffic), near the HBM / VALU balance point for D = 6.  DESIGN.md 3.7.
//
// One translation unit per model: the Makefile compiles this file eight times, -DMOIHGP_X_TU=22, 23, 24, 32, 33, 34 (= DB J; 21, 31: the
// reference's own models, team kernel only), in
// parallel (one unit with all six took eight minutes); stack_dispatch.hip holds the dispatcher over the units' entries.
#ifndef MOIHGP_X_TU
#error "compile with -DMOIHGP_X_TU=<DB J> (21, 31, 22, 23, 24, 32, 33, 34): csrc/Makefile"
#endif
#include "x_common.h"
#include <atomic>
#include <hip/hip_ext.h>
#include <cstdlib>

namespace moihgp {
namespace {

// Tuning probes (-DMOIHGP_TUNING -DMOIHGP_X_SKIP=bits, experiment libraries only): leave out the chunk response (1), the scan levels (2)
// or the replay (4) of the clean path, to attribute the kernel's time to its phases.  Results are wrong by construction.
#if defined(MOIHGP_TUNING) && defined(MOIHGP_X_SKIP)
constexpr int kXSkip = MOIHGP_X_SKIP;
#else
constexpr int kXSkip = 0;
#endif

// ---- the replay: every lane walks its chunk of kChunkX ticks from its true start state, in innovation form
//          v = y - HA x;   x <- A x + K v   (== AKHA x + K y, ihgp.h:90);   yhat = x(0) (ihgp.h:91);   sum of v^2 for ihgp.h:207
// A is block diagonal (J blocks of DB x DB).  Three forms of the wave-uniform operands, by what the register files hold:
//   PK  (fp32): the state travels as PAIRS across two components, X_q = (x_q of block 2p, x_q of block 2p + 1), and every
//        multiply-add on a pair is ONE v_pk_fma_f32 with the coefficient pair in an SGPR pair (table XC::PK, written by the update kernel):
//        18 vector instructions per tick at d = 6 where the scalar form takes 38 (an odd J pairs its last block with zeros);
//   SOP (fp64, d <= 8): HA, K and the blocks of A as plain scalar operands: no zero-initialised partial sums, no DPP issue cost
//        (a lone wave issues a DPP multiply-add every ~9 cycles, a scalar-operand one every ~4.5: tools/micro/fma_rate.hip);
//   DPP (fp64, d = 9, 12): HA and K as slabs read through row_newbcast, the blocks of A as scalar operands (all three would be 90+
//        SGPRs at d = 9).
// The chunk is read and written 16 bytes at a time (2 / 4 ticks), and the next group's read is issued before this group's arithmetic:
// a tick no longer waits for its own LDS read.
// One instantiation serves full, ragged and warm-up segments alike (a second, masked one took part in the kernel's register
// allocation and cost the full segments 10 %), and the tick loop carries no per-lane masking: every lane walks its whole chunk,
// lanes past the end compute on zero padding nobody reads.  What a ragged end needs is taken at the one tick where the stream
// ends -- tick klast of lane jl, a wave-uniform test: that lane parks its state in LDS (the carry-out) and adds its running sum
// of squared innovations to the total there.  head (warm-up ticks of a time slice, not counted) is a multiple of the chunk length, so a
// lane's chunk counts as a whole or not at all.
template <typename T, int DB, int J>
struct ReplayConst {
    static constexpr int D = DB * J;
    static constexpr bool PK = sizeof(T) == 4, SOP = sizeof(T) == 8 && D <= 8;
    static constexpr int NPAIR = (J + 1) / 2, NPK = NPAIR * (DB * DB + 2 * DB);
    typedef T T2 __attribute__((ext_vector_type(2)));
    T a[PK ? 1 : J * DB * DB];                     // diagonal blocks of A
    T has[SOP ? D : 1], ks[SOP ? D : 1];           // HA, K as scalars
    T ha, kk;                                      // HA, K as slabs (DPP form)
    T2 pk[PK ? NPK : 1];                           // per block pair: A (row-major), HA, K as coefficient pairs
};

template <typename T, int DB, int J, typename P>
__device__ inline void load_replay_const(P cu /* uniform view of the latent's block */, const T* __restrict__ c, int lane, ReplayConst<T, DB, J>& rc) {
    using RC = ReplayConst<T, DB, J>;
    using Lay = XC<DB * J>;
    if constexpr (RC::PK) {
#pragma unroll
        for (int i = 0; i < RC::NPK; i++) { rc.pk[i].x = cu[Lay::PK + 2 * i]; rc.pk[i].y = cu[Lay::PK + 2 * i + 1]; }
    } else {
#pragma unroll
        for (int i = 0; i < J * DB * DB; i++) rc.a[i] = cu[Lay::AB + i];
        if constexpr (RC::SOP) {
#pragma unroll
            for (int i = 0; i < DB * J; i++) { rc.has[i] = cu[Lay::HA + i]; rc.ks[i] = cu[Lay::K + i]; }
        } else {
            rc.ha = c[Lay::HA16 + (lane & 15)];
            rc.kk = c[Lay::K16 + (lane & 15)];
        }
    }
}

// DPP form (fp64, d = 9 and 12): tick by tick, HA and K through slabs; the d = 12 instantiation sits at 256 registers exactly (two waves
// per SIMD), which the grouped form below would overflow.
// PRED (the imputation's first sweep and the filters' impulse responses: GAPS = 1 / 3 of filter_x_body): the tile receives the PREDICTED observation HA x (pre-step) instead of the filtered mean xnew(0, 0) -- as
// y - v, which at a tick whose y is zero is HA x exactly
template <typename T, int DB, int J, bool WRITE, bool NLL, bool PRED = false>
__device__ inline void replay_dpp(const T (&a)[J * DB * DB], const T& ha, const T& kk, T* tile_lane, T* carry, int lane, int n, int head,
                              T (&xs)[DB * J], T (&xc)[DB * J], double& acc, unsigned& nobs) {
    constexpr int D = DB * J;
    const int jl = (n - 1) / kChunkX, klast = (n - 1) % kChunkX;      // lane and tick of the last tick of the segment
    double part = 0.0;
    const bool counted = lane * kChunkX >= head;
    // (unrolled by two: the state ping-pongs between two register sets instead of being copied back every tick -- 12 of the ~85
    //  instructions of a tick at d = 12)
#pragma unroll 2
    for (int k = 0; k < kChunkX; k++) {
        const T y = tile_lane[k];
        T h0 = 0, h1 = 0, h2 = 0;                                    // three partial sums of HA x: shorter dependent chains
        static_for<D>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            fmac_bc<i>(i % 3 == 0 ? h0 : (i % 3 == 1 ? h1 : h2), ha, xs[i]);
        });
        const T v = y - ((h0 + h1) + h2);
        if (NLL) {
            const double vd = (double)v;
            part = fma(vd, vd, part);                               // ihgp.h:206-207, pre-step state
        }
        T xn[D];
#pragma unroll
        for (int j = 0; j < J; j++)
#pragma unroll
            for (int r = 0; r < DB; r++) {
                T sum = a[j * DB * DB + r * DB] * xs[j * DB];
#pragma unroll
                for (int q = 1; q < DB; q++) sum = fma(a[j * DB * DB + r * DB + q], xs[j * DB + q], sum);
                xn[j * DB + r] = sum;
            }
        static_for<D>([&](auto ii) { fmac_bc<decltype(ii)::value>(xn[decltype(ii)::value], kk, v); });   // ihgp.h:90 as A x + K (y - HA x)
#pragma unroll
        for (int i = 0; i < D; i++) xs[i] = xn[i];
        if (WRITE) tile_lane[k] = PRED ? (y - v) : xn[0];           // ihgp.h:91 `yhat = xnew(0, 0)`, literally
        if (k == klast) {                                           // wave-uniform
            if (lane == jl) {
                if (NLL && counted) acc += part;
#pragma unroll
                for (int i = 0; i < D; i++) carry[i] = xn[i];
            }
        }
    }
    if (NLL) {
        acc += (counted && lane < jl) ? part : 0.0;
        nobs += counted ? (lane < jl ? (unsigned)kChunkX : (lane == jl ? (unsigned)(klast + 1) : 0u)) : 0u;
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < D; i++) xc[i] = carry[i];                    // one broadcast read per entry
    wave_lds_fence();
}

// FULLFAST: a full segment without warm-up (n == 64 kChunkX, head == 0: a wave-uniform test) takes a copy of the loop without the per-tick
// "is this the stream's last tick" test (five scalar instructions and a branch per tick: a lone wave issues one instruction of any kind
// per four cycles, so they cost as much as vector ones); its carry-out is lane 63's final state.  The team kernel asks for it; the
// many-latent kernel keeps the single loop (a second copy took part in its register allocation).
template <typename T, int DB, int J, bool WRITE, bool NLL, bool FULLFAST = false, int CKR = kChunkX /* ticks per lane */, bool PRED = false>
__device__ inline void replay(const ReplayConst<T, DB, J>& rc, T* tile_lane, T* carry, int lane, int n, int head,
                              T (&xs)[DB * J], T (&xc)[DB * J], double& acc, unsigned& nobs) {
    using RC = ReplayConst<T, DB, J>;
    using V = typename VecOf<T>::type;
    constexpr int D = DB * J, EPV = 16 / sizeof(T), NG = CKR / EPV;
    if constexpr (!RC::PK && !RC::SOP) {
        static_assert(RC::PK || RC::SOP || CKR == kChunkX, "the slab form is built for kChunkX ticks per lane");
        replay_dpp<T, DB, J, WRITE, NLL, PRED>(rc.a, rc.ha, rc.kk, tile_lane, carry, lane, n, head, xs, xc, acc, nobs);
        return;
    }
    const int jl = (n - 1) / CKR, klast = (n - 1) % CKR;      // lane and tick of the last tick of the segment
    const bool counted = lane * CKR >= head;
    // the state: scalars, or pairs across two components (PK)
    typedef typename RC::T2 T2;
    T2 X[RC::PK ? RC::NPAIR * DB : 1];
    if constexpr (RC::PK) {
#pragma unroll
        for (int p = 0; p < RC::NPAIR; p++)
#pragma unroll
            for (int q = 0; q < DB; q++) { X[p * DB + q].x = xs[(2 * p) * DB + q]; X[p * DB + q].y = (2 * p + 1 < J) ? xs[(2 * p + 1 < J ? 2 * p + 1 : 0) * DB + q] : T(0); }
    }
    T partf = 0;                    // PK: sum of v^2 over the chunk in stream precision (32 terms), added to the fp64 total once
    double part = 0.0;
    V cur = *reinterpret_cast<const V*>(tile_lane);
    auto walk = [&](auto check_tag) {
    constexpr bool CHECK = decltype(check_tag)::value;
#pragma unroll 2
    for (int g = 0; g < NG; g++) {
        const V nxt = *reinterpret_cast<const V*>(tile_lane + (g + 1 < NG ? g + 1 : g) * EPV);    // in flight during this group's ticks
        T y[EPV], o[EPV];
        unpack<T>(cur, y);
#pragma unroll
        for (int e = 0; e < EPV; e++) {
            const int k = g * EPV + e;
            if constexpr (RC::PK) {
                constexpr int PB = DB * DB + 2 * DB;                   // coefficient pairs per block pair: A, then HA, then K
                T2 H = rc.pk[DB * DB] * X[0];
#pragma unroll
                for (int p = 0; p < RC::NPAIR; p++)
#pragma unroll
                    for (int q = 0; q < DB; q++) if (p + q > 0) H = rc.pk[p * PB + DB * DB + q] * X[p * DB + q] + H;
                const T v = y[e] - (H.x + H.y);
                if (NLL) partf = fma(v, v, partf);
                const T2 vv = {v, v};
                T2 XN[RC::NPAIR * DB];
#pragma unroll
                for (int p = 0; p < RC::NPAIR; p++)
#pragma unroll
                    for (int r = 0; r < DB; r++) {
                        T2 sacc = rc.pk[p * PB + r * DB] * X[p * DB];
#pragma unroll
                        for (int q = 1; q < DB; q++) sacc = rc.pk[p * PB + r * DB + q] * X[p * DB + q] + sacc;
                        XN[p * DB + r] = rc.pk[p * PB + DB * DB + DB + r] * vv + sacc;
                    }
#pragma unroll
                for (int i = 0; i < RC::NPAIR * DB; i++) X[i] = XN[i];
                o[e] = PRED ? (y[e] - v) : XN[0].x;                    // ihgp.h:91 `yhat = xnew(0, 0)`, literally
                if (CHECK && k == klast) {                             // wave-uniform
                    if (lane == jl) {
                        if (NLL && counted) acc += (double)partf;
#pragma unroll
                        for (int p = 0; p < RC::NPAIR; p++)
#pragma unroll
                            for (int q = 0; q < DB; q++) {
                                carry[(2 * p) * DB + q] = XN[p * DB + q].x;
                                if (2 * p + 1 < J) carry[(2 * p + 1 < J ? 2 * p + 1 : 0) * DB + q] = XN[p * DB + q].y;
                            }
                    }
                }
            } else if constexpr (RC::SOP) {
                T v;
                {
                    T h0 = fma(rc.has[0], xs[0], -y[e]), h1 = rc.has[1] * xs[1];
#pragma unroll
                    for (int i = 2; i < D; i++) { if (i % 2 == 0) h0 = fma(rc.has[i], xs[i], h0); else h1 = fma(rc.has[i], xs[i], h1); }
                    v = -(h0 + h1);                                    // (the negation folds into the uses below as a source modifier)
                }
                if (NLL) {
                    const double vd = (double)v;
                    part = fma(vd, vd, part);                          // ihgp.h:206-207, pre-step state
                }
                T xn[D];
#pragma unroll
                for (int j = 0; j < J; j++)
#pragma unroll
                    for (int r = 0; r < DB; r++) {
                        T sum = rc.a[j * DB * DB + r * DB] * xs[j * DB];
#pragma unroll
                        for (int q = 1; q < DB; q++) sum = fma(rc.a[j * DB * DB + r * DB + q], xs[j * DB + q], sum);
                        xn[j * DB + r] = sum;
                    }
#pragma unroll
                for (int i = 0; i < D; i++) xn[i] = fma(rc.ks[i], v, xn[i]);                        // ihgp.h:90 as A x + K (y - HA x)
#pragma unroll
                for (int i = 0; i < D; i++) xs[i] = xn[i];
                o[e] = PRED ? (y[e] - v) : xn[0];                      // ihgp.h:91 `yhat = xnew(0, 0)`, literally
                if (CHECK && k == klast) {                             // wave-uniform
                    if (lane == jl) {
                        if (NLL && counted) acc += part;
#pragma unroll
                        for (int i = 0; i < D; i++) carry[i] = xn[i];
                    }
                }
            }
        }
        if (WRITE) *reinterpret_cast<V*>(tile_lane + g * EPV) = pack<T>(o);
        cur = nxt;
    }
    };
    if (FULLFAST && n == 64 * CKR && head == 0) {
        walk(std::false_type{});
        if (lane == 63) {                                              // the segment's last tick is lane 63's last
            if constexpr (RC::PK) {
#pragma unroll
                for (int p = 0; p < RC::NPAIR; p++)
#pragma unroll
                    for (int q = 0; q < DB; q++) {
                        carry[(2 * p) * DB + q] = X[p * DB + q].x;
                        if (2 * p + 1 < J) carry[(2 * p + 1 < J ? 2 * p + 1 : 0) * DB + q] = X[p * DB + q].y;
                    }
                if (NLL) acc += (double)partf;
            } else {
#pragma unroll
                for (int i = 0; i < D; i++) carry[i] = xs[i];
                if (NLL) acc += part;
            }
        }
    } else {
        walk(std::true_type{});
    }
    if (NLL) {
        if constexpr (RC::PK) part = (double)partf;
        acc += (counted && lane < jl) ? part : 0.0;
        nobs += counted ? (lane < jl ? (unsigned)CKR : (lane == jl ? (unsigned)(klast + 1) : 0u)) : 0u;
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < D; i++) xc[i] = carry[i];                    // one broadcast read per entry
    wave_lds_fence();
}

// n ticks of the tile, one after the other, in innovation form (an observed tick is ihgp.h:90 as A x + K (y - HA x); a missing one
// ihgp.h:83-87, x <- A x).  One state entry per lane, a QUAD of lanes per component: lane 4 j + r holds entry r of component j and
// row r of its diagonal block of A.  Per tick a lane forms (A x)_i from its quad (DPP quad_perm broadcasts: DB multiply-adds);
// H = [H_1 .. H_J] reads the first state of every component, so HA x is the sum over the quads of their lane-0 value of A x -- two
// row_ror additions inside the wave's first 16-lane row -- and an observed tick adds K (y - HA x).  About a dozen dependent
// instructions per tick (the first version gathered the whole state with v_readlane against full rows of A: 2 D + ... per tick,
// 175 ns; this one 40-60 ns).  One entry per lane: this path must not set the kernel's registers.
template <typename T, int DB, int J, bool WRITE, bool NLL, int CKS = kChunkX /* ticks per tile row */>
__device__ inline void sequential(const T* __restrict__ c, T* tile, int stride, int n, int head, int lane, T (&xc)[DB * J], double& acc, unsigned& nobs) {
    constexpr int D = DB * J;
    using Lay = XC<D>;
    static_assert(J <= 4 && DB <= 3, "one 16-lane row holds the quads");
    const int jb = (lane >> 2) < J ? (lane >> 2) : 0, r = (lane & 3) < DB ? (lane & 3) : 0;
    const bool live = (lane >> 2) < J && (lane & 3) < DB;
    T arow[DB], kk = live ? c[Lay::K + jb * DB + r] : T(0), xv = 0;
#pragma unroll
    for (int q = 0; q < DB; q++) arow[q] = live ? c[Lay::AB + jb * DB * DB + r * DB + q] : T(0);
#pragma unroll
    for (int i = 0; i < D; i++) if (live && jb * DB + r == i) xv = xc[i];
    const bool head_lane = live && r == 0;
    T ynext = tile[0];
#pragma unroll 1
    for (int t = 0; t < n; t++) {
        T* slot = tile + (t / CKS) * stride + (t % CKS);
        const T y = ynext;                                           // same address in every lane: one broadcast read, one tick ahead
        const int tn = t + 1 < n ? t + 1 : t;
        ynext = tile[(tn / CKS) * stride + (tn % CKS)];
        T s = arow[0] * dpp0<0x00, 0xF>(xv);
        s = fma(arow[1], dpp0<0x55, 0xF>(xv), s);
        if (DB > 2) s = fma(arow[DB - 1], dpp0<0xAA, 0xF>(xv), s);
        if (!(y != y)) {
            T hs = head_lane ? s : T(0);
            hs += dpp0<0x124, 0xF>(hs);                              // row_ror:4, row_ror:8: the four quads' lane-0 values summed
            hs += dpp0<0x128, 0xF>(hs);
            const T v = y - dpp0<0x00, 0xF>(hs);                     // (every lane of a quad reads its lane 0)
            s = fma(kk, v, s);
            if (NLL && lane == 0 && t >= head) { const double vd = (double)v; acc = fma(vd, vd, acc); nobs++; }
        }
        xv = s;
        if (WRITE && lane == 0) *slot = s;                           // ihgp.h:91 `yhat = xnew(0, 0)`
    }
#pragma unroll
    for (int i = 0; i < D; i++) xc[i] = read_lane(xv, 4 * (i / DB) + i % DB);
}

// SPLIT (few latents, WPB == 1): workgroup (l, s) of the 2-D grid handles time slice s of latent l.  A slice after the first
// starts from a ZERO state a warm-up of CK * 2^nlev ticks before its first tick: the true state there would only enter
// through AKHA^(CK 2^nlev) = M^(2^nlev), which the update kernel has flagged as below 1e-20 (1e-10 in fp32) -- the criterion
// by which the scan already drops its upper levels.  Warm-up ticks are neither counted nor written.  Latents that do not
// decay that fast (nlev = 6, or unstable) are run whole by slice 0.  Per-slice NLL partials go to nll_part [L][nslice].

// LINKS (second pass, many latents only): the first pass (LINKS = false, given link_flags) stops a latent at the first segment whose
// gaps can be handled as broken links (below: at most max_links chunks with a gap), parks the carried state, the segment's start
// and the per-lane NLL sums in link_state and flags it; the LINKS = true instantiation resumes exactly those latents there and
// finishes them (broken links where a segment allows it, the walk where not).  Two instantiations because the broken-link stages
// keep a second response vector alive next to the scan's: built into the one kernel they took d = 12 fp64 from 256 VGPRs to
// 256 + 72 AGPRs (one wave per SIMD), for every stream, gaps or not.
// The first pass hands a latent over at a segment with at most max_links chunks with a gap (launch argument, carried by segs_per_slice:
// 32, or 3 for fp64 with d > 9, whose second pass runs one wave per SIMD and loses to the walk from ~15 such chunks on); once there, the second pass takes every segment with up to
// kLinksSecondPass such chunks as broken links: a stage is one scan + one replay (~10 us), the walk ~400 us per segment and, in the
// second pass, without a second wave to overlap with.  Measured with tools/filternan.py (profiles/r02/filternan_links.log).
constexpr int kLinksSecondPass = 32;
constexpr int kPairMaxDim = 6;       // state dims up to which the second pass can scan the chunks' own maps (D x D per lane in registers)
constexpr int kPairFrom = 2;         // ... and does so for windows with more than this many chunks with a gap
constexpr int kLinkState = 144;      // doubles per latent in link_state: x [D], segment start; from 16 on the per-lane sums of v^2 [64], n_obs [64]

// Waves per SIMD the register allocation must allow (second argument of __launch_bounds__ under HIP).
#ifndef MOIHGP_X_MINW_F32
#define MOIHGP_X_MINW_F32 1
#endif
#ifndef MOIHGP_X_MINW_F64
#define MOIHGP_X_MINW_F64 1
#endif
template <typename T, int D, bool SPLIT, bool LINKS>
constexpr int x_min_waves() {
    if (SPLIT || LINKS) return 1;
    // fp32 up to d = 9: 128 registers, i.e. 4 waves per SIMD = all 4096 wavefronts of a 4096-latent bank resident at once (at 137 registers
    // three fit, and the fourth thousand of wavefronts ran as a second round at one wave per SIMD: d = 6 104.8 -> 94.7 us, d = 9 138 -> 127;
    // d = 12 spills under the cap and lost: 160 -> 166 us)
    if (sizeof(T) == 4 && D <= 9 && MOIHGP_X_MINW_F32 == 1) return 4;
    return sizeof(T) == 4 ? MOIHGP_X_MINW_F32 : MOIHGP_X_MINW_F64;
}

// The sweep of ONE wavefront over (a slice of) one latent's stream: the body of filter_x_kernel, and the fallback of the team kernel
// below (one latent per workgroup), which hands a latent it cannot take to this code.  `tile` is the wave's padded LDS tile
// (64 x (kChunkX + 16 bytes)), `carry` its D-entry carry-out slot.
// Missing ticks by exact imputation (the fused kernel below; the method is described there).  GAPS: 0 = none of it;
//   1 = first sweep: resumes at the pass-1 hand-over, the gaps swept as zeros, the tile receives the PREDICTED observations HA x, of which those
//       at the gaps are listed (tick, HA x') in gio->pos / gio->val, in tick order; nothing else is written (no means, no end state);
//   2 = second sweep: resumes likewise, the gaps filled from gio->val (in the same order), and everything is written as usual;
//   3 = the sweep writes predicted observations instead of filtered means (the filters' impulse responses).
template <typename T>
struct GapIO {
    int* pos; T* val;            // this latent's lists
    int* lds;                    // 128 ints of the wave's own LDS
    const double* resume;        // the latent's hand-over record (kLinkState doubles)
    int count;                   // gaps listed (out, GAPS = 1)
    bool patch;                  // GAPS = 2: the lists hold the fill values
};
template <typename T, int DB, int J, bool WRITE, bool NLL, bool SPLIT, bool LINKS, int GAPS = 0>
__device__ __forceinline__ void filter_x_body(const T* __restrict__ Ty, size_t Tlen, size_t ld, const T* __restrict__ cbT, const double* __restrict__ cb64,
                const T* xin0 /* start state */, T* x /* end state; may be the same buffer */, T* __restrict__ yhat, double* __restrict__ nll,
                int nslice, int segs_per_slice, double* __restrict__ nll_part, size_t ldo /* row stride of yhat */,
                int* __restrict__ link_flags /* [L] or NULL */, double* __restrict__ link_state /* [L][kLinkState] */,
                const size_t l, const int slice, const int lane, T* __restrict__ tile, T* __restrict__ carry, GapIO<T>* gio = nullptr) {
    constexpr int D = DB * J;
    constexpr bool PRED = GAPS == 1 || GAPS == 3;
    static_assert(GAPS == 0 || (!SPLIT && !LINKS), "the imputation sweeps are whole-stream, first-pass sweeps");
    using V = typename VecOf<T>::type;
    using Lay = XC<D>;
    constexpr int CK = kChunkX, EPV = 16 / sizeof(T), STRIDE = CK + EPV, SEG = 64 * CK;
    const T* __restrict__ c = cbT + l * Lay::SIZE;
    T* tile_lane = tile + lane * STRIDE;
    const T* row = Ty + l * ld;
    T* orow = WRITE ? yhat + l * ldo : nullptr;
    T xc[D];
#pragma unroll
    for (int i = 0; i < D; i++) xc[i] = xin0[l * D + i];
    double acc = 0.0;
    unsigned nobs = 0;
    size_t t_resume = 0;
    if (LINKS) {                                                     // second pass: resume where the first one stopped this latent
        if (!link_flags[l]) return;
        if (lane == 0) link_flags[l] = 0;                              // (consumed: the flags are all zero again when the pass ends -- no memset per sweep)
        const double* st = link_state + l * kLinkState;
#pragma unroll
        for (int i = 0; i < D; i++) xc[i] = (T)st[i];
        t_resume = (size_t)st[D];
        if (NLL) { acc = st[16 + lane]; nobs = (unsigned)st[80 + lane]; }          // (per-lane partial sums, added up at the end)
    }
    int gbase = 0;                                                   // gaps listed so far (scalar)
    if constexpr (GAPS == 1 || GAPS == 2) {                          // resume where the first pass stopped this latent (its flag is the caller's business)
        const double* st = gio->resume;
#pragma unroll
        for (int i = 0; i < D; i++) xc[i] = (T)st[i];
        t_resume = (size_t)st[D];
        if (NLL) { acc = st[16 + lane]; nobs = (unsigned)st[80 + lane]; }
    }
    const bool scan_ok = __builtin_amdgcn_readfirstlane((int)(c[Lay::SCANOK] > T(0))) != 0;
    // SCANOK < 0 (fp32 blocks of a bank of 1024 latents and more, stationary_x.hip): tables unusable in fp32, fine in fp64 -- the caller sweeps this
    // latent in fp64 on the side (capi.cpp), the many-latent kernel leaves it alone
    if constexpr (sizeof(T) == 4 && !SPLIT && !LINKS && (GAPS == 0 || GAPS == 3)) {
        if (__builtin_amdgcn_readfirstlane((int)(c[Lay::SCANOK] < T(0))) != 0) return;
    }
    // The scan powers and the response table are streamed in at the start of every segment (L2-resident, coalesced, issued ahead
    // of the phases that use them) rather than held for the whole sweep: that leaves registers to fetch the NEXT segment of the
    // stream during the replay, so a wave does not sit waiting for HBM at a segment boundary.  fp64 with 12 states has room for
    // three quarters of a segment only (half in the split kernels); the rest is fetched at the segment start.
    constexpr int NPF = (sizeof(T) == 8 && D > 9) ? (kChunkX / (16 / (int)sizeof(T))) * (SPLIT ? 2 : 3) / 4 : kChunkX / (16 / (int)sizeof(T));   // prefetched vectors per lane
    // (the register budget must not overflow into AGPRs: a reload placed in front of a DPP FMA is a hazard the compiler does not
    // see -- tools/check_dpp_hazard.py checks the built code)
    constexpr int NSL = Lay::LS / 16, NSG = Lay::GN / 16, NRES = 4;
    T ha = c[Lay::HA16 + (lane & 15)], kk = c[Lay::K16 + (lane & 15)];
    const int nlev = __builtin_amdgcn_readfirstlane((int)c[Lay::NLEV]);
    // the ticks this wave sweeps: [t_start, t_end), of which [t_begin, t_end) count (everything, unless SPLIT)
    size_t t_begin = 0, t_end = Tlen, t_start = (LINKS || GAPS == 1 || GAPS == 2) ? t_resume : 0;
    bool last = true;
    if (SPLIT) {
        const bool split_ok = scan_ok && nlev <= 5;                  // M^(2^nlev) is in the table and negligible
        if (split_ok) {
            // Slice 0 owns its segs_per_slice whole segments; a later slice owns that many ticks MINUS its warm-up, so that warm-up + own
            // ticks are exactly segs_per_slice segment passes (a slice that owned whole segments and started its warm-up in the segment
            // before them paid one more pass, 64 lanes wide, for a few hundred warm-up ticks: two passes instead of one at one segment
            // per slice).  The warm-up length is this latent's own (nlev), so the slices of different latents differ; the grid holds
            // enough slices for the longest warm-up, the surplus ones of a fast-decaying latent find nothing to do.
            const size_t span = (size_t)segs_per_slice * SEG, own = span - ((size_t)CK << nlev);
            t_begin = slice == 0 ? 0 : span + (size_t)(slice - 1) * own;
            const size_t stop = slice == 0 ? span : t_begin + own;
            t_end = stop < Tlen ? stop : Tlen;
        }
        if ((!split_ok && slice > 0) || t_begin >= t_end) {          // nothing to do for this slice
            if (NLL && lane == 0) nll_part[l * nslice + slice] = 0.0;
            if (Tlen > 0 || slice > 0) return;
        }
        if (split_ok && slice > 0) {
            t_start = t_begin - ((size_t)CK << nlev);                // t_begin >= SEG = 64 CK >= CK 2^nlev
#pragma unroll
            for (int i = 0; i < D; i++) xc[i] = T(0);
        }
        last = (t_end == Tlen);
    }
    T sp[NRES][NSL];
    V pre[NPF];
    auto fetch = [&](size_t t0, int n) {
        int lo = lane;
        asm volatile("" : "+v"(lo));       // recompute the lane addresses here: hoisted out of the loop they would be 32 live registers
#pragma unroll
        for (int r = 0; r < NPF; r++) {
            const int e = (r * 64 + lane) * EPV;
            V v{};                                                   // always (re)defined: dead between its use and the next fetch
            // (uniform base + 32-bit lane offset: 64-bit per-lane addresses for all the loads would cost 32 registers)
            if (e < n) v = nt_load(reinterpret_cast<const V*>(row + t0) + (r * 64 + lo));
            pre[r] = v;
        }
    };
    if (t_end > t_start) fetch(t_start, (int)(t_end - t_start < (size_t)SEG ? t_end - t_start : (size_t)SEG));

    for (size_t t0 = t_start; t0 < t_end; t0 += SEG) {
        const int n = (int)(t_end - t0 < (size_t)SEG ? t_end - t0 : (size_t)SEG);
        const int head = SPLIT && t_begin > t0 ? (int)(t_begin - t0) : 0;      // warm-up ticks at the front of this segment
        // The imputation sweeps stage the tile FIRST and deal with the gaps there, before the tables below fill the registers (at that point the
        // d = 12 fp64 sweep sits at 256 registers exactly; the gap bookkeeping on top of it cost 50 more and a wave per SIMD).
        int seg_gaps = 0;
        if constexpr (GAPS == 1 || GAPS == 2) {
#pragma unroll
            for (int r = 0; r < CK / EPV; r++) {
                const int e = (r * 64 + lane) * EPV;
                T vals[EPV];
                if (r < NPF) unpack<T>(pre[r < NPF ? r : 0], vals);
                else if (e < n) unpack<T>(nt_load(reinterpret_cast<const V*>(row + t0 + e)), vals);
#pragma unroll
                for (int q = 0; q < EPV; q++) if (e + q >= n) vals[q] = T(0);        // beyond the stream: inert zeros
                *reinterpret_cast<V*>(tile + (e / CK) * STRIDE + (e % CK)) = pack<T>(vals);
            }
            wave_lds_fence();
            if (GAPS == 1 || gio->patch) {
                // this lane's chunk: which of its ticks are missing; their places in the latent's lists (tick order = lane order, then k)
                unsigned mask = 0;
#pragma unroll
                for (int kv = 0; kv < CK / EPV; kv++) {
                    T yv[EPV];
                    unpack<T>(*reinterpret_cast<const V*>(tile_lane + kv * EPV), yv);
#pragma unroll
                    for (int q = 0; q < EPV; q++) mask |= (yv[q] != yv[q]) ? (1u << (kv * EPV + q)) : 0u;
                }
                const int cnt = __builtin_popcount(mask);
                int inc = cnt;                                       // inclusive scan over the lanes, in DPP
                inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_SHR + 1, 0xF, 0xF, false);
                inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_SHR + 2, 0xF, 0xF, false);
                inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_SHR + 4, 0xF, 0xF, false);
                inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_SHR + 8, 0xF, 0xF, false);
                inc += __builtin_amdgcn_update_dpp(0, inc, DPP_ROW_BCAST15, 0xA, 0xF, false);
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
                seg_gaps = __builtin_amdgcn_readlane(inc, 63);
                if (seg_gaps) {
                    const int at = gbase + inc - cnt;
                    if (GAPS == 1) { gio->lds[lane] = (int)mask; gio->lds[64 + lane] = at; }
                    int j = 0;
                    for (unsigned mm = mask; mm; mm &= mm - 1) {
                        const int k = __builtin_ctz(mm);
                        tile_lane[k] = GAPS == 1 ? T(0) : gio->val[at + j];
                        j++;
                    }
                    wave_lds_fence();
                }
                if (GAPS == 2) gbase += seg_gaps;
            }
            __builtin_amdgcn_sched_barrier(0);                       // (none of the loads below before this is done)
        }
        // ---- issue this segment's table traffic first: the response slabs (used once per segment, so streamed rather than
        // kept) and the diagonal blocks of A for the replay (scalar loads; the SGPRs are idle until then) ----
        const uptr<T> cu = launder(c);
        constexpr int NSG0 = NSG / 2, NSG1 = NSG - NSG0;           // two halves: the second is fetched while the first is used
        T g0[NSG0], g1[NSG1];
        load_slabs<T, NSG0>(cu + Lay::G, lane, g0);
        load_slabs<T, NSG1>(cu + Lay::G + NSG0 * 16, lane, g1);
        T ablk[J * DB * DB];                                        // (the paths with gaps walk with these; the clean path's replay has its own set, rc)
#pragma unroll
        for (int i = 0; i < J * DB * DB; i++) ablk[i] = cu[Lay::AB + i];
        ReplayConst<T, DB, J> rc;
        load_replay_const<T, DB, J>(cu, c, lane, rc);
        // ---- stage in: coalesced 16-byte loads, chunk-major into the padded tile ----
        if constexpr (GAPS != 1 && GAPS != 2) {
#pragma unroll
            for (int r = 0; r < CK / EPV; r++) {
                const int e = (r * 64 + lane) * EPV;
                T vals[EPV];
                if (r < NPF) unpack<T>(pre[r < NPF ? r : 0], vals);
                else if (e < n) unpack<T>(nt_load(reinterpret_cast<const V*>(row + t0 + e)), vals);
#pragma unroll
                for (int q = 0; q < EPV; q++) if (e + q >= n) vals[q] = T(0);        // beyond the stream: inert zeros
                *reinterpret_cast<V*>(tile + (e / CK) * STRIDE + (e % CK)) = pack<T>(vals);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                           // (the prefetch registers are free from here on)
#pragma unroll
        for (int lv = 0; lv < NRES; lv++) load_slabs<T, NSL>(cu + Lay::SP + lv * Lay::LS, lane, sp[lv]);   // (M^16, M^32: on demand below)
        wave_lds_fence();
        // ---- chunk response z = sum_k g_k y_k, and the missing-data test ----
        T z[D];
#pragma unroll
        for (int i = 0; i < D; i++) z[i] = T(0);
        bool bad = false;
        static_for<CK / EPV>([&](auto kvv) {
            constexpr int kv = decltype(kvv)::value;
            T yv[EPV];
            unpack<T>(*reinterpret_cast<const V*>(tile_lane + kv * EPV), yv);
            static_for<EPV>([&](auto qq) {
                constexpr int k = kv * EPV + decltype(qq)::value;
                bad = bad || (yv[decltype(qq)::value] != yv[decltype(qq)::value]);
                if constexpr (kXSkip & 1) { z[k % D] += yv[decltype(qq)::value]; return; }
                static_for<D>([&](auto ii) {
                    constexpr int e = k * D + decltype(ii)::value, sl = e / 16;
                    if constexpr (sl < NSG0) fmac_bc<e % 16>(z[decltype(ii)::value], g0[sl], yv[decltype(qq)::value]);
                    else fmac_bc<e % 16>(z[decltype(ii)::value], g1[sl - NSG0], yv[decltype(qq)::value]);
                });
            });
            // keep the scheduler from hoisting every LDS read of the chunk to the top (that alone would be 64 registers)
            if constexpr (kv % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        });
        const size_t left = t0 + SEG < t_end ? t_end - (t0 + SEG) : 0;
        const int nnext = (int)(left < (size_t)SEG ? left : (size_t)SEG);
        // Chunks (lanes) that hold a missing tick.  Few of them, and a latent the scan can be trusted with: the second pass (LINKS) treats
        // them as broken links of the chain (below); otherwise (time slices, too: their warm-up bookkeeping assumes whole passes) the
        // segment is walked tick by tick.  (The two forms of the test below are deliberate: the d = 12 fp64 instantiations sit at
        // 256 VGPRs exactly, and each of them stays there with one form and tips into AGPRs -- one wave per SIMD -- with the other.)
        unsigned long long dirty = 0;
        bool links = false;
        if constexpr (!SPLIT) {
            dirty = __builtin_amdgcn_ballot_w64(bad);
            links = LINKS && scan_ok && dirty != 0 && (D <= kPairMaxDim || __builtin_popcountll(dirty) <= kLinksSecondPass);
        }
        if (SPLIT ? (!scan_ok || __builtin_amdgcn_ballot_w64(bad) != 0) : (!scan_ok || (dirty != 0 && !links))) {
            fetch(t0 + SEG, nnext);
            __builtin_amdgcn_sched_barrier(0);
            // An unstable latent whose state has left the format (inf / NaN) has nothing finite ahead of it: the rest of its
            // stream is filled with NaN instead of being walked tick by tick (that walk is 10 x slower than the segment solve, and
            // one such latent would hold the whole launch back; the reference's own values there are inf / NaN as well).
            bool any_lost = false;
#pragma unroll
            for (int i = 0; i < D; i++) any_lost = any_lost || !((xc[i] - xc[i]) == T(0));
            if (!scan_ok && any_lost) {
                const T qnan = __builtin_nan("");
                if (WRITE) {
#pragma unroll 4
                    for (int k = 0; k < CK; k++) tile_lane[k] = qnan;
                }
#pragma unroll
                for (int i = 0; i < D; i++) xc[i] = qnan;
                if (NLL) acc = __builtin_nan("");
            } else {
                if (!LINKS && !SPLIT && link_flags && scan_ok && __builtin_popcountll(dirty) <= segs_per_slice) {
                    // first pass: hand the latent over to the second one, from this segment on
                    double* st = link_state + l * kLinkState;
                    if (NLL) { st[16 + lane] = acc; st[80 + lane] = (double)nobs; }
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < D; i++) st[i] = (double)xc[i];
                        st[D] = (double)t0;
                        link_flags[l] = 1;
                    }
                    return;
                }
                sequential<T, DB, J, WRITE, NLL>(c, tile, STRIDE, n, head, lane, xc, acc, nobs);
            }
        } else if (LINKS && links) {
            // ---- Broken links.  The chunks WITHOUT a gap still move the state by the uniform M = AKHA^32 and their responses z are
            // the table sums above; a chunk with a gap moves it by a matrix of its own, which nobody forms: the chain is cut there.
            // Stage by stage: scan the gap-free run [lo, hi) that starts from the carried state, which gives the start state of every
            // chunk up to and including the one with the gap (hi); replay exactly those chunks, gap-aware (a missing tick is
            // x <- A x: v = 0, ihgp.h:83-87); the end state of chunk hi is the carried state of the next run.  One scan + one replay
            // per chunk with a gap, against ~200 ns per tick for the walk: sparse gaps cost little more than none.
            // (The next window's stream is fetched at the END of this branch: its registers are needed here, and an allocation that
            // overflows into AGPRs puts v_accvgpr_read in front of the DPP multiply-adds -- the hazard the build checks for.)
            const int jl = (n - 1) / CK, klast = (n - 1) % CK;       // lane and tick of the segment's last tick
            T cin[D];
#pragma unroll
            for (int i = 0; i < D; i++) cin[i] = xc[i];
            double part = 0.0;
            unsigned cnt = 0;
            int lo = 0;
            // ---- Small states (D <= 6) with more than a couple of such chunks: the chunk maps themselves fit a lane.  Every lane
            // walks its chunk once with D + 1 vectors -- the response from a zero state and the D unit start states, gap-aware --
            // which gives its affine map (M_j, z_j); the pairs are scanned over the lanes (Kogge-Stone, lane shifts by ds_bpermute,
            // identity where a lane has no source), and ONE gap-aware replay from the true start states finishes the window.
            // ~25 us for a window however many gaps it holds, against one scan + replay (~10 us) per chunk with a gap above.
            bool use_pairs = false;
            T xs_pairs[D];
            if constexpr (D <= kPairMaxDim) {
                use_pairs = __builtin_popcountll(dirty) > kPairFrom;
                if (use_pairs) {
                    T mcol[D][D], zr[D];                             // mcol[c] = M e_c (column c of the chunk's transition matrix)
#pragma unroll
                    for (int cidx = 0; cidx < D; cidx++)
#pragma unroll
                        for (int i = 0; i < D; i++) mcol[cidx][i] = (i == cidx) ? T(1) : T(0);
#pragma unroll
                    for (int i = 0; i < D; i++) zr[i] = T(0);
                    auto tick_vec = [&](T (&xv)[D], T yin, bool miss) {
                        T h0 = 0, h1 = 0;
                        static_for<D>([&](auto ii) { constexpr int i = decltype(ii)::value; fmac_bc<i>(i % 2 == 0 ? h0 : h1, ha, xv[i]); });
                        const T v = miss ? T(0) : yin - (h0 + h1);
                        T xn[D];
#pragma unroll
                        for (int j = 0; j < J; j++)
#pragma unroll
                            for (int r = 0; r < DB; r++) {
                                T sum = ablk[j * DB * DB + r * DB] * xv[j * DB];
#pragma unroll
                                for (int q = 1; q < DB; q++) sum = fma(ablk[j * DB * DB + r * DB + q], xv[j * DB + q], sum);
                                xn[j * DB + r] = sum;
                            }
                        static_for<D>([&](auto ii) { fmac_bc<decltype(ii)::value>(xn[decltype(ii)::value], kk, v); });
#pragma unroll
                        for (int i = 0; i < D; i++) xv[i] = xn[i];
                    };
#pragma unroll 1
                    for (int k = 0; k < CK; k++) {
                        const T y = tile_lane[k];
                        const bool miss = (y != y);
                        tick_vec(zr, miss ? T(0) : y, miss);
#pragma unroll
                        for (int cidx = 0; cidx < D; cidx++) tick_vec(mcol[cidx], T(0), miss);
                    }
                    if (lane == 0) {                                 // the carried state enters through lane 0's map
#pragma unroll
                        for (int cidx = 0; cidx < D; cidx++)
#pragma unroll
                            for (int i = 0; i < D; i++) zr[i] = fma(mcol[cidx][i], xc[cidx], zr[i]);
                    }
#pragma unroll 1
                    for (int lv = 0; lv < 6; lv++) {
                        const int sh = 1 << lv, addr = ((lane - sh) & 63) * 4;
                        const bool has = lane >= sh;
                        T zp[D], ncol[D][D];
#pragma unroll
                        for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, zr[i]); zp[i] = has ? m : T(0); }
#pragma unroll
                        for (int cidx = 0; cidx < D; cidx++) {      // new column c = M (partner's column c)
                            T pc[D];
#pragma unroll
                            for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, mcol[cidx][i]); pc[i] = has ? m : ((i == cidx) ? T(1) : T(0)); }
#pragma unroll
                            for (int i = 0; i < D; i++) {
                                T sacc = 0;
#pragma unroll
                                for (int q = 0; q < D; q++) sacc = fma(mcol[q][i], pc[q], sacc);
                                ncol[cidx][i] = sacc;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < D; q++)
#pragma unroll
                            for (int i = 0; i < D; i++) zr[i] = fma(mcol[q][i], zp[q], zr[i]);     // z = M z_p + z (old M)
#pragma unroll
                        for (int cidx = 0; cidx < D; cidx++)
#pragma unroll
                            for (int i = 0; i < D; i++) mcol[cidx][i] = ncol[cidx][i];
                    }
                    const int addr1 = ((lane - 1) & 63) * 4;
#pragma unroll
                    for (int i = 0; i < D; i++) { const T m = bperm<T>(addr1, zr[i]); xs_pairs[i] = lane >= 1 ? m : xc[i]; }
                }
            }
#pragma unroll 1
            for (;;) {
                const unsigned long long rem = (dirty >> lo) << lo;
                int hi = (rem && !use_pairs) ? (int)__builtin_ctzll(rem) : jl;
                if (hi > jl) hi = jl;
                const bool in_run = lane >= lo && lane < hi;
                T zz[D], t[D];
#pragma unroll
                for (int i = 0; i < D; i++) { zz[i] = in_run ? z[i] : T(0); t[i] = T(0); }   // (z of a chunk with a gap is NaN: never read)
                if (lo < hi && !use_pairs) {
                    matvec_bc<T, D, NSL>(sp[0], cin, t);
#pragma unroll
                    for (int i = 0; i < D; i++) zz[i] += (lane == lo) ? t[i] : T(0);
#pragma unroll
                    for (int lv = 0; lv < 6; lv++) {
                        if (lv >= nlev) break;
                        const int sh = 1 << lv, addr = ((lane - sh) & 63) * 4;
#pragma unroll
                        for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, zz[i]); t[i] = lane >= sh ? m : T(0); }
                        if (lv < NRES) matvec_bc<T, D, NSL>(sp[lv < NRES ? lv : 0], t, zz);
                        else {
                            T hp[NSL];
                            load_slabs<T, NSL>(cu + Lay::SP + lv * Lay::LS, lane, hp);
                            matvec_bc<T, D, NSL>(hp, t, zz);
                        }
                    }
                }
                T xs[D];
                {
                    const int addr = ((lane - 1) & 63) * 4;
#pragma unroll
                    for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, zz[i]); xs[i] = lane == lo ? cin[i] : m; }
                }
                if constexpr (D <= kPairMaxDim) {
                    if (use_pairs) {
#pragma unroll
                        for (int i = 0; i < D; i++) xs[i] = xs_pairs[i];
                    }
                }
                const bool act = lane >= lo && lane <= hi;
#pragma unroll 1
                for (int k = 0; k < CK; k++) {
                    const T y = tile_lane[k];
                    const bool miss = (y != y);
                    T h0 = 0, h1 = 0, h2 = 0;
                    static_for<D>([&](auto ii) {
                        constexpr int i = decltype(ii)::value;
                        fmac_bc<i>(i % 3 == 0 ? h0 : (i % 3 == 1 ? h1 : h2), ha, xs[i]);
                    });
                    const T v = miss ? T(0) : y - ((h0 + h1) + h2);
                    if (NLL) {
                        const bool counted = act && !miss && (lane < jl || k <= klast);
                        const double vd = counted ? (double)v : 0.0;
                        part = fma(vd, vd, part);
                        cnt += counted ? 1u : 0u;
                    }
                    T xn[D];
#pragma unroll
                    for (int j = 0; j < J; j++)
#pragma unroll
                        for (int r = 0; r < DB; r++) {
                            T sum = ablk[j * DB * DB + r * DB] * xs[j * DB];
#pragma unroll
                            for (int q = 1; q < DB; q++) sum = fma(ablk[j * DB * DB + r * DB + q], xs[j * DB + q], sum);
                            xn[j * DB + r] = sum;
                        }
                    static_for<D>([&](auto ii) { fmac_bc<decltype(ii)::value>(xn[decltype(ii)::value], kk, v); });
#pragma unroll
                    for (int i = 0; i < D; i++) xs[i] = xn[i];
                    if (WRITE && act) tile_lane[k] = xn[0];
                    if (k == klast && lane == jl && act) {             // the state after the segment's last tick
#pragma unroll
                        for (int i = 0; i < D; i++) carry[i] = xn[i];
                    }
                }
                if (hi >= jl) break;
#pragma unroll
                for (int i = 0; i < D; i++) cin[i] = read_lane(xs[i], hi);
                lo = hi + 1;
            }
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < D; i++) xc[i] = carry[i];
            wave_lds_fence();
            if (NLL) { acc += part; nobs += cnt; }
            __builtin_amdgcn_sched_barrier(0);
            fetch(t0 + SEG, nnext);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // ---- carry-in on lane 0, then the 64-lane Kogge-Stone scan with uniform powers ----
            T t[D];
#pragma unroll
            for (int i = 0; i < D; i++) t[i] = T(0);
            matvec_bc<T, D, NSL>(sp[0], xc, t);
#pragma unroll
            for (int i = 0; i < D; i++) z[i] += (lane == 0) ? t[i] : T(0);
#pragma unroll
            for (int lv = 0; lv < 6; lv++) {
                if (lv >= nlev || (kXSkip & 2)) break;               // uniform: the remaining powers are negligible
                const int s = 1 << lv, addr = ((lane - s) & 63) * 4;
#pragma unroll
                for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, z[i]); t[i] = lane >= s ? m : T(0); }
                if (lv < NRES) matvec_bc<T, D, NSL>(sp[lv < NRES ? lv : 0], t, z);
                else {
                    T hi[NSL];
                    load_slabs<T, NSL>(cu + Lay::SP + lv * Lay::LS, lane, hi);
                    matvec_bc<T, D, NSL>(hi, t, z);
                }
            }
            // ---- start state of every lane = end state of the lane before it ----
            T xs[D];
            {
                const int addr = ((lane - 1) & 63) * 4;
#pragma unroll
                for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, z[i]); xs[i] = lane >= 1 ? m : xc[i]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch(t0 + SEG, nnext);                                  // next segment's stream, in flight during the replay
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kXSkip & 4) {
#pragma unroll
                for (int i = 0; i < D; i++) xc[i] = read_lane(xs[i], 63);
            } else
            replay<T, DB, J, WRITE, NLL, false, kChunkX, PRED>(rc, tile_lane, carry, lane, n, head, xs, xc, acc, nobs);
        }
        if constexpr (GAPS == 1) {                                   // the predictions at the gaps, listed; nothing else leaves the tile
            if (seg_gaps) {
                wave_lds_fence();
                const int at = gio->lds[64 + lane];
                int j = 0;
                for (unsigned mm = (unsigned)gio->lds[lane]; mm; mm &= mm - 1) {
                    const int k = __builtin_ctz(mm);
                    gio->pos[at + j] = (int)(t0 + (size_t)lane * CK + k);
                    gio->val[at + j] = tile_lane[k];
                    j++;
                }
                gbase += seg_gaps;
            }
        }
        // ---- stage out ----
        if (WRITE && GAPS != 1) {
            wave_lds_fence();
            int lo = lane;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int r = 0; r < CK / EPV; r++) {
                const int e = (r * 64 + lane) * EPV;
                if (e < n && e >= head) {                            // (head is a multiple of CK: whole vectors)
                    const V out = *reinterpret_cast<const V*>(tile + (e / CK) * STRIDE + (e % CK));
                    nt_store(out, reinterpret_cast<V*>(orow + t0) + (r * 64 + lo));
                }
            }
        }
        wave_lds_fence();
    }
    if constexpr (GAPS == 1) gio->count = gbase;
    if (lane == 0 && last && GAPS != 1) {
#pragma unroll
        for (int i = 0; i < D; i++) x[l * D + i] = xc[i];
    }
    if (NLL) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { acc += __shfl_xor(acc, o, 64); nobs += __shfl_xor(nobs, o, 64); }
        if (lane == 0) {
            const double* c64 = cb64 + l * Lay::SIZE;
            const double v = 0.5 * (acc / c64[Lay::S] + (double)nobs * c64[Lay::LOGS]);
            if (SPLIT) nll_part[l * nslice + slice] = v; else nll[l] = v;
        }
    }
}

template <typename T, int DB, int J, bool WRITE, bool NLL, int WPB, bool SPLIT, bool LINKS, bool PREDOUT = false>
__global__ void __launch_bounds__(64 * WPB, (x_min_waves<T, DB * J, SPLIT, LINKS>()))
filter_x_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64,
                const T* xin0, T* x, T* __restrict__ yhat, double* __restrict__ nll, int nslice, int segs_per_slice, double* __restrict__ nll_part, size_t ldo,
                int* __restrict__ link_flags, double* __restrict__ link_state) {
    constexpr int D = DB * J, STRIDE = kChunkX + 16 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) T tiles[WPB][64 * STRIDE];
    __shared__ T carries[WPB][D];                                    // carry-out of a segment (written by the lane that holds its last tick)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = SPLIT ? (size_t)blockIdx.x : (size_t)blockIdx.x * WPB + wave;   // SPLIT: grid = (latents, slices)
    const int slice = SPLIT ? (int)blockIdx.y : 0;
    if (l >= L) return;                                              // no workgroup barrier below
    filter_x_body<T, DB, J, WRITE, NLL, SPLIT, LINKS, PREDOUT ? 3 : 0>(Ty, Tlen, ld, cbT, cb64, xin0, x, yhat, nll, nslice, segs_per_slice, nll_part, ldo, link_flags,
                                                                     link_state, l, slice, lane, tiles[wave], carries[wave]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// MISSING TICKS BY EXACT IMPUTATION (round 4): the latents the first pass handed over (a state too wide for per-chunk maps: d >= 8 by default).
// Reference semantics: ihgp.h:83-87 -- a NaN observation advances the state by x <- A x (no correction), i.e. the innovation form
// x' = A x + K v with v = 0 -- and ihgp.h:204-209 adds no likelihood term for it.
//
// Why.  The sweep above solves a segment of 64 x 32 ticks in parallel because every chunk moves the state by the same matrix AKHA^32.  A chunk
// with a gap moves it by a matrix of its own; up to d = 6 that matrix fits a lane and the chunks' maps are scanned, beyond it does not, and the
// second pass treats gaps as broken links of the scan (one scan + one replay per chunk with a gap) or walks the segment tick by tick: 12-16 x the
// gap-free sweep once most chunks hold a gap (1 % of the ticks missing: 3.8 ms against 0.31 ms at d = 12, open since round 2).
//
// What.  A missing tick is an observation that happens to equal its own prediction: with y_p := w_p = HA x_p (the predicted observation at the
// gap) the ordinary recursion gives v_p = 0 and x_(p+1) = A x_p -- the reference's branch.  The w_p are not known in advance, but they obey a
// SCALAR triangular system.  Sweep the stream with the gaps set to zero (state x'); then e = x - x' moves by AKHA between gaps and is kicked
// by K w_p at each gap, so
//         w_p = HA x'_p + sum over gaps q < p of  s_(p-q-1) w_q,        s_k = HA AKHA^k K   (the filter's scalar impulse response),
// where HA x'_p is what the first sweep leaves in its tile in place of the filtered mean (y - v: exactly HA x' where y' = 0; the filtered mean
// xnew(0, 0) would not do -- for the stacked models H sums over the blocks), and the sum runs over the few gaps inside the decay of s.  Filling the
// gaps with w_p and sweeping ONCE MORE gives the true filtered means, states and sum of v^2 (the gaps contribute v = 0 to it); only the count
// of observed ticks needs correcting: nll -= n_gaps log(S) / 2.
//   Two gap-free sweeps + a scalar recursion over the gaps, whatever their density -- and all of it per latent: one wavefront takes a flagged
// latent through the first sweep and the recursion (filter_x_gaps_a_kernel: the lists of gaps live in the latent's scratch row, the recursion's
// table in the wave's tile), one through the second sweep (filter_x_gaps_b_kernel); a bank without gaps leaves both at once.  (One kernel for
// all three stages was tried: two inlined sweeps in one kernel cost 315 registers at d = 12 fp64 -- one wave per SIMD -- and ran 1.4 x slower.)  The impulse responses come from the same sweep run over a unit observation (gaps_x.hip, once per parameter
// update).  A latent the table form cannot take -- a response that has not decayed within kGapSMax ticks, more gaps inside its decay than the
// ring holds -- is solved by the state form of the same system (gap_solve_states); only a latent whose response is not finite or GROWS (rho(AKHA) > 1: the
// zero-filled sweep then departs from the true one exponentially and the sum x' + e cancels) keeps its flag and takes the second pass (LINKS) as before.
constexpr int kGapRing = 256;         // gaps inside the decay window the recursion keeps (LDS)

// The scalar recursion over one latent's n gaps (ticks pos[], predictions val[], both in tick order): on success wout[g] = w_g.  imp: the filter's
// response to a unit observation at tick 0 from a zero state, as the GAPS = 3 sweep writes it: imp[k + 1] = s_k.  lds: the wave's tile.
// 64 gaps at a time, one per lane, by forward substitution in column order: the finished w of the earlier gaps -- the last kGapRing of the blocks
// before (newest first, until one lies outside the decay), then lane by lane inside the block -- are broadcast, and every later lane adds
// s_(p-q-1) w_q to its own sum.  No reduction across lanes; about a dozen instructions per gap and broadcast.
template <typename T>
__device__ __forceinline__ int gap_solve_wave(const T* __restrict__ imp, const int* pos, const T* val, T* wout, const int n, const size_t Tlen, const int lane, unsigned char* lds) {
    T* st = reinterpret_cast<T*>(lds);
    int* ringp = reinterpret_cast<int*>(lds + kGapSMax * sizeof(T));
    double* ringw = reinterpret_cast<double*>(lds + kGapSMax * sizeof(T) + kGapRing * sizeof(int));
    constexpr int kZero = kGapSMax - 1;                                // (the table's last entry is zero)
    double smax = 0.0;
    bool bad = false;
#pragma unroll 4
    for (int k = lane; k < kGapSMax; k += 64) {
        const T sv = k + 1 < kGapSMax ? imp[k + 1] : T(0);
        st[k] = sv;
        bad |= !(fabs((double)sv) < 1e300);
        smax = fmax(smax, fabs((double)sv));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) smax = fmax(smax, __shfl_xor(smax, o));
    const double tol = sizeof(T) == 8 ? 1e-17 : 1e-9;                  // (the sweep itself drops scan levels below 1e-20 / 1e-10)
    int kd = 0;                                                        // one past the last k whose |s_k| still matters
    for (int k = lane; k < kGapSMax; k += 64)
        if (fabs((double)st[k]) > tol * smax) kd = k + 1;              // (each lane reads back what it wrote)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kd = max(kd, __shfl_xor(kd, o));
    const int kdec = __builtin_amdgcn_readfirstlane(kd);
    // A response that GROWS (rho(AKHA) > 1: the literal DARE returns such gains) rules imputation out altogether: the sweep with zeros at the gaps
    // then departs from the true one like rho^t, and x = x' + e cancels that many digits.  Growth over the table: its last quarter against its
    // first, extrapolated to the length of the stream; beyond 1e4 (fp64; 1e2 in fp32) the latent is left to the second pass.
    double head = 0.0, tail = 0.0;
    for (int k = lane; k < kGapSMax / 4; k += 64) { head = fmax(head, fabs((double)st[k])); tail = fmax(tail, fabs((double)st[3 * kGapSMax / 4 + k])); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { head = fmax(head, __shfl_xor(head, o)); tail = fmax(tail, __shfl_xor(tail, o)); }
    const bool grows = tail > head && log(tail / head) * ((double)Tlen / (0.75 * kGapSMax)) > (sizeof(T) == 8 ? 9.2 : 4.6);
    // (the return value: 0 = solved, otherwise why not -- 1: a table that is not finite or grows, 2: a response that outlives the table, 3: more gaps
    // inside the decay than the ring holds, 4: a stream too long for 32-bit ticks, 5: a fill value that is not finite)
    int why = (__builtin_amdgcn_ballot_w64(bad) != 0 || __builtin_amdgcn_readfirstlane((int)grows)) ? 1 : kdec > kGapSMax - 64 ? 2 : Tlen >= (1u << 30) ? 4 : 0;
    bool wild = false;
    wave_lds_fence();
    int nring = 0;                                                     // gaps solved so far; gap g sits in ring slot g % kGapRing
    int posN = lane < n ? pos[lane] : 0;
    T hvN = lane < n ? val[lane] : T(0);
    for (int g0 = 0; g0 < n && why == 0; g0 += 64) {
        const int p = posN;
        double w = (double)hvN;
        if (g0 + 64 < n) {                                             // the next 64, in flight during these
            posN = g0 + 64 + lane < n ? pos[g0 + 64 + lane] : 0;
            hvN = g0 + 64 + lane < n ? val[g0 + 64 + lane] : T(0);
        }
        const int m = n - g0 < 64 ? n - g0 : 64;
        const int pfirst = __builtin_amdgcn_readlane(p, 0);
        if (nring >= kGapRing) {                                       // the ring is full: the gap it dropped last must be outside the decay
            const int dropped = __builtin_amdgcn_readfirstlane(ringp[nring & (kGapRing - 1)]);
            if (pfirst - dropped - 1 < kdec) { why = 3; break; }
        }
        const int nh = nring < kGapRing ? nring : kGapRing;
        for (int h = 1; h <= nh; h++) {                                // the blocks before, newest gap first
            const int slot = (nring - h) & (kGapRing - 1);
            const int pq = __builtin_amdgcn_readfirstlane(ringp[slot]);
            if (pfirst - pq - 1 >= kdec) break;
            const double wq = ringw[slot];
            const int k = p - pq - 1;                                  // (idle lanes: negative)
            w = fma((double)st[(unsigned)k < (unsigned)kdec ? k : kZero], wq, w);
        }
        for (int q = 0; q + 1 < m; q++) {                              // inside the block: lane q is final when its turn comes
            const double wq = read_lane(w, q);
            const int k = p - __builtin_amdgcn_readlane(p, q) - 1;     // (lanes up to q, idle lanes: negative)
            w = fma((double)st[(unsigned)k < (unsigned)kdec ? k : kZero], wq, w);
        }
        if (lane < m) {
            const int slot = (nring + lane) & (kGapRing - 1);
            ringp[slot] = p;
            ringw[slot] = w;
            wout[g0 + lane] = (T)w;
            wild |= !(fabs(w) < (sizeof(T) == 8 ? 1e300 : 1e37));
        }
        nring += m;
        wave_lds_fence();
    }
    if (why == 0 && __builtin_amdgcn_ballot_w64(wild) != 0) why = 5;
    return why;
}

__device__ inline double wave_sum_dpp(double v) {                      // all in DPP; the total comes back uniform
    v += dpp0<DPP_ROW_SHR + 1, 0xF>(v);
    v += dpp0<DPP_ROW_SHR + 2, 0xF>(v);
    v += dpp0<DPP_ROW_SHR + 4, 0xF>(v);
    v += dpp0<DPP_ROW_SHR + 8, 0xF>(v);
    v += dpp0<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp0<0x143 /* row_bcast:31 */, 0xC>(v);
    return read_lane(v, 63);
}

// The same system without the table, for the latents the table form cannot take (a filter that remembers more than kGapSMax ticks, more gaps inside
// its memory than the ring holds): the error state e = x - x' itself, D numbers carried from gap to gap --
//     w_p = HA x'_p + HA e_p ;   e <- AKHA e + K w_p  through the gap ;   e <- AKHA^g e  across the g observed ticks to the next one
// with AKHA^g from binary powers: AKHA^(1, 2, 4, 8, 16) formed here (LDS), AKHA^(32 .. 1024) the latent's own scan levels (fp64 block).  Any
// memory length, any density, and exact -- but a few D x D matrix-vector products on D lanes per gap (~230 instructions where the table form
// takes a dozen).  All of it in fp64.
template <typename T, int D>
__device__ __forceinline__ bool gap_solve_states(const double* __restrict__ c64, const int* pos, const T* val, T* wout, const int n, const int lane, unsigned char* lds) {
    using Lay = XC<D>;
    constexpr int NN = D * D;
    double* pw = reinterpret_cast<double*>(lds);                       // [5][D * D], row-major
    for (int e = lane; e < NN; e += 64) pw[e] = c64[Lay::AKHA + e];
    wave_lds_fence();
    for (int lv = 1; lv < 5; lv++) {
        const double* src = pw + (lv - 1) * NN;
        for (int e = lane; e < NN; e += 64) {
            const int i = e / D, j = e % D;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) acc = fma(src[i * D + k], src[k * D + j], acc);
            pw[lv * NN + e] = acc;
        }
        wave_lds_fence();
    }
    const int li = lane < D ? lane : 0;                                // (idle lanes shadow lane 0; nobody reads them)
    const double ha = c64[Lay::HA + li], kk = c64[Lay::K + li];
    double err = 0.0;                                                  // lane i < D: entry i of e
    auto advance = [&](const double* m) {                              // e <- m e (m row-major, LDS or global)
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < D; j++) acc = fma(m[li * D + j], read_lane(err, j), acc);
        err = acc;
    };
    int tprev = -1;
    bool wild = false;
    int posN = lane < n ? pos[lane] : 0;
    T hvN = lane < n ? val[lane] : T(0);
    for (int g0 = 0; g0 < n; g0 += 64) {
        const int p = posN;
        const double hv = (double)hvN;
        if (g0 + 64 < n) {
            posN = g0 + 64 + lane < n ? pos[g0 + 64 + lane] : 0;
            hvN = g0 + 64 + lane < n ? val[g0 + 64 + lane] : T(0);
        }
        const int m = n - g0 < 64 ? n - g0 : 64;
        double wv = 0.0;
        for (int q = 0; q < m; q++) {
            const int tp = __builtin_amdgcn_readlane(p, q);
            if (tprev >= 0) {                                          // across the observed ticks since the last gap
                int g = tp - tprev - 1;
#pragma unroll
                for (int b = 0; b < 5; b++) if (g & (1 << b)) advance(pw + b * NN);
                g >>= 5;
#pragma unroll
                for (int lv = 0; lv < 6; lv++) if (g & (1 << lv)) advance(c64 + Lay::SP + lv * Lay::LS);
                for (g >>= 6; g > 0; g--) { advance(c64 + Lay::SP + 5 * Lay::LS); advance(c64 + Lay::SP + 5 * Lay::LS); }       // 2048 ticks at a time
            }
            const double w = read_lane(hv, q) + wave_sum_dpp(lane < D ? ha * err : 0.0);
            advance(pw);                                               // through the gap: e <- AKHA e + K w
            err = fma(kk, w, err);
            wv = lane == q ? w : wv;
            tprev = tp;
        }
        if (lane < m) wout[g0 + lane] = (T)wv;
        wild |= !(fabs(wv) < (sizeof(T) == 8 ? 1e300 : 1e37));
    }
    return __builtin_amdgcn_ballot_w64(wild) == 0;                     // every fill value finite
}

// Is imputation safe for this latent over a stream of Tlen ticks?  Not if its filter is unstable (rho(AKHA) > 1: the literal DARE of dare.h:23 returns
// such gains for part of the parameter box): the sweep with zeros at the gaps then departs from the true one like rho^t and x = x' + e cancels
// that many digits -- 1e11 of them for rho = 1.0026 over 16384 ticks (tools/fuzz_campaign.py, seed 401: a latent whose growing mode is so
// weakly observed that its impulse response still DECAYS over the 1024 ticks of the table, which is why the table alone is not asked).  Growth is
// read off the matrix: n2 = max|AKHA^2048|, n4 = max|AKHA^4096| (the latent's scan level AKHA^1024, squared twice here).  What a rounding error
// can grow to over the stream is bounded by n4 up to 4096 ticks and, where the powers still grow (n4 > n2: by then only a growing mode is
// left), by n4 (n4 / n2)^((Tlen - 4096) / 2048) beyond; more than 1e4 (fp64 streams) / 1e2 (fp32) and the latent is left to the second pass.
// lds: D * D doubles x 2.
template <typename T, int D>
__device__ __forceinline__ bool gap_filter_grows(const double* __restrict__ c64, const size_t Tlen, const int lane, unsigned char* lds) {
    using Lay = XC<D>;
    constexpr int NN = D * D;
    double* a = reinterpret_cast<double*>(lds);
    double* b = a + NN;
    for (int e = lane; e < NN; e += 64) a[e] = c64[Lay::SP + 5 * Lay::LS + e];      // AKHA^1024
    wave_lds_fence();
    double big[2];
    bool nan = false;
#pragma unroll
    for (int sq = 0; sq < 2; sq++) {
        const double* src = sq ? b : a;
        double* dst = sq ? a : b;
        double m = 0.0;
        for (int e = lane; e < NN; e += 64) {
            const int i = e / D, j = e % D;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) acc = fma(src[i * D + k], src[k * D + j], acc);
            dst[e] = acc;
            m = fmax(m, fabs(acc));
            nan |= !(acc == acc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        big[sq] = m;
        wave_lds_fence();
    }
    const double limit = sizeof(T) == 8 ? 9.2 : 4.6;                    // log of 1e4 / 1e2
    double lg = big[1] > 0.0 ? log(big[1]) : -700.0;
    if (big[1] > big[0] && big[0] > 0.0 && Tlen > 4096) lg += log(big[1] / big[0]) * ((double)(Tlen - 4096) / 2048.0);
    const bool grows = __builtin_amdgcn_ballot_w64(nan) != 0 || lg > limit;
    return __builtin_amdgcn_readfirstlane((int)grows) != 0;
}

// Register caps of the two kernels below: what the plain sweep of the same model happens to fit (fp64: 256 = two waves per SIMD; fp32: three at
// d = 12, four below) -- the gap bookkeeping costs 4 .. 16 registers more, and a few spilled values are cheaper than the wave they would cost.
template <typename T, int D>
constexpr int x_gaps_min_waves() { return sizeof(T) == 8 ? 2 : (D <= 9 ? 4 : 3); }

// first sweep + recursion: the latents the first pass flagged (the flag stays: filter_x_gaps_b_kernel clears it)
template <typename T, int DB, int J, int WPB>
__global__ void __launch_bounds__(64 * WPB, (x_gaps_min_waves<T, DB * J>()))
filter_x_gaps_a_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64, const T* xin0,
                       const int* __restrict__ flags, const double* __restrict__ link_state /* the first pass's hand-over records */,
                       const T* __restrict__ imp /* [L][kGapSMax] impulse responses */, int* gpos, T* gval /* [L][gcap] scratch: the gaps' ticks and predictions */,
                       T* gw /* [L][gcap] out: their fill values */, size_t gcap,
                       int* __restrict__ gstat /* [L] out: 2 * gaps + 1 if solved (bit 30: by the state form), else 2 * the reason why not */) {
    constexpr int D = DB * J, STRIDE = kChunkX + 16 / (int)sizeof(T);
    static_assert(sizeof(T) * 64 * STRIDE >= sizeof(T) * kGapSMax + (sizeof(int) + sizeof(double)) * kGapRing, "the recursion's table and ring borrow the wave's tile");
    static_assert(sizeof(T) * 64 * STRIDE >= sizeof(double) * 5 * D * D, "so do the state form's powers");
    __shared__ __attribute__((aligned(16))) T tiles[WPB][64 * STRIDE];
    __shared__ T carries[WPB][D];
    __shared__ int glds[WPB][128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = (size_t)blockIdx.x * WPB + wave;
    if (l >= L || flags[l] == 0) return;                             // no workgroup barrier below
    unsigned char* lds = reinterpret_cast<unsigned char*>(tiles[wave]);
    if (gap_filter_grows<T, D>(cb64 + l * XC<D>::SIZE, Tlen, lane, lds)) {      // an unstable filter: not for imputation (its flag stays: second pass)
        if (lane == 0) gstat[l] = 2 * 1;
        return;
    }
    GapIO<T> gio;
    gio.pos = gpos + l * gcap; gio.val = gval + l * gcap; gio.lds = glds[wave]; gio.resume = link_state + l * kLinkState; gio.count = 0; gio.patch = false;
    filter_x_body<T, DB, J, true, false, false, false, 1>(Ty, Tlen, ld, cbT, cb64, xin0, nullptr, nullptr, nullptr, 1, 0, nullptr, 0, nullptr, nullptr,
                                                          l, 0, lane, tiles[wave], carries[wave], &gio);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");               // the lists: written above, read below, by this wave
    wave_lds_fence();
    const int n_gaps = __builtin_amdgcn_readfirstlane(gio.count);
    int why = gap_solve_wave<T>(imp + l * kGapSMax, gio.pos, gio.val, gw + l * gcap, n_gaps, Tlen, lane, lds);
    int form = 0;
    if (why == 2 || why == 3) {                                      // a long memory, or many gaps inside it: the state form
        wave_lds_fence();
        why = gap_solve_states<T, D>(cb64 + l * XC<D>::SIZE, gio.pos, gio.val, gw + l * gcap, n_gaps, lane, lds) ? 0 : 5;
        form = 1 << 30;
    }
    if (lane == 0) gstat[l] = why == 0 ? (2 * n_gaps + 1) | form : 2 * why;
}

// second sweep: the gaps filled with their own predictions (a latent the recursion gave up keeps its flag: the second pass takes it)
template <typename T, int DB, int J, bool WRITE, bool NLL, int WPB>
__global__ void __launch_bounds__(64 * WPB, (x_gaps_min_waves<T, DB * J>()))
filter_x_gaps_b_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64,
                       const T* xin0, T* x, T* __restrict__ yhat, double* __restrict__ nll, size_t ldo,
                       int* __restrict__ flags /* cleared here */, const double* __restrict__ link_state,
                       const int* __restrict__ gpos, const T* __restrict__ gw, size_t gcap, const int* __restrict__ gstat) {
    constexpr int D = DB * J, STRIDE = kChunkX + 16 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) T tiles[WPB][64 * STRIDE];
    __shared__ T carries[WPB][D];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = (size_t)blockIdx.x * WPB + wave;
    if (l >= L || flags[l] == 0) return;                             // no workgroup barrier below
    const int stat = gstat[l];
    if (!(stat & 1)) return;
    GapIO<T> gio;
    gio.pos = const_cast<int*>(gpos) + l * gcap; gio.val = const_cast<T*>(gw) + l * gcap; gio.lds = nullptr; gio.resume = link_state + l * kLinkState;
    gio.count = 0; gio.patch = true;
    filter_x_body<T, DB, J, WRITE, NLL, false, false, 2>(Ty, Tlen, ld, cbT, cb64, xin0, x, yhat, nll, 1, 0, nullptr, ldo, nullptr, nullptr,
                                                         l, 0, lane, tiles[wave], carries[wave], &gio);
    if (lane == 0) {
        // the gaps were swept as observations that equal their predictions: v = 0, but counted (ihgp.h:204-209 does not).  Nothing observed: exactly 0
        const int n_gaps = (stat & 0x3FFFFFFF) >> 1;
        if (NLL) nll[l] = (size_t)n_gaps == Tlen ? 0.0 : nll[l] - 0.5 * (double)n_gaps * cb64[l * XC<D>::SIZE + XC<D>::LOGS];
        flags[l] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// TEAM: few latents (BASELINE.json configs[1]: 256 latents x 10^4 ticks).  One WORKGROUP per latent, one wavefront per segment of
// 64 x 32 ticks, every segment of the stream at once (the stream holds at most kTeamWaves segments), the stream read ONCE and no tick
// replayed twice (the time slices of the SPLIT kernel above start from a zero state a warm-up before their first tick: for a
// 10^4-tick stream that is 9 one-segment passes per latent, half of each a warm-up, where this kernel does 5).
//   A. every wave: its segment into its tile, chunk responses z_j, Kogge-Stone scan from a ZERO state -> e0_j, the state after chunk j
//      had the segment started from zero; lane 63's is the segment's zero-start end state, parked in LDS.           __syncthreads()
//   B. the true state entering segment s is  c_s = e0_63(s-1) + M^64 c_(s-1),  M^64 = AKHA^2048 -- the power the update kernel has
//      flagged as below 1e-20 (1e-10 in fp32) whenever nlev <= 5, the criterion the scan itself drops its upper levels by and the
//      SPLIT kernel's warm-up rests on -- so  c_s = e0_63(s-1)  to rounding (c_0 = the caller's start state).  It enters the lanes'
//      start states linearly:  start_j = e0_(j-1) + M^j c_s,  and  q_j = M^j c_s  is the same scan applied to (c_s, 0, 0, ..).
//      Then the replay, the stores, and the latent's NLL from the waves' partial sums (LDS, second barrier).
// A latent this cannot take -- a missing tick anywhere in its stream (chunk maps are no longer powers of one matrix), scan tables that
// do not decay that fast or are unusable -- is handed, inside the same launch, to the one-wavefront sweep above (wave 0 runs
// filter_x_body over the whole stream, the other waves leave): exact, slower, as the SPLIT kernel treats such latents.
constexpr int kTeamWaves = 8;          // 512 threads: the one-wavefront fallback keeps its 256-register budget
template <int D> constexpr int team_table_len() { return XC<D>::GN + 6 * XC<D>::LS; }             // response table + the six scan powers (contiguous in XC)
template <typename T, int D> constexpr size_t team_part_offset(int nw) { return (((size_t)nw * (64 * (kChunkX + 16 / sizeof(T)) + 2 * D) + team_table_len<D>()) * sizeof(T) + 15) / 16 * 16; }
template <typename T, int D> constexpr size_t team_smem_bytes(int nw) { return team_part_offset<T, D>(nw) + (size_t)nw * 2 * sizeof(double); }

template <typename T, int DB, int J, bool WRITE, bool NLL>
__global__ void __launch_bounds__(64 * kTeamWaves)
filter_x_team_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64,
                     const T* xin0, T* x, T* __restrict__ yhat, double* __restrict__ nll, size_t ldo, int nw /* wavefronts = segments of the stream */) {
    constexpr int D = DB * J;
    using V = typename VecOf<T>::type;
    using Lay = XC<D>;
    constexpr int CK = kChunkX, EPV = 16 / sizeof(T), STRIDE = CK + EPV, SEG = 64 * CK;
    constexpr int NSL = Lay::LS / 16, NSG = Lay::GN / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char team_smem[];
    T* tiles = reinterpret_cast<T*>(team_smem);                                       // [nw][64 * STRIDE]
    T* e0end = tiles + (size_t)nw * 64 * STRIDE;                                       // [nw][D]  zero-start end state of every segment
    T* carries = e0end + (size_t)nw * D;                                               // [nw][D]  carry-out slots of the replays (and of the fallback sweep)
    T* tab = carries + (size_t)nw * D;                                                 // [GN + 6 LS]  this latent's response table and scan powers
    double* part = reinterpret_cast<double*>(team_smem + team_part_offset<T, D>(nw));  // [nw][2]  per-wave sum of v^2, observed ticks
    __shared__ int dirty_any;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = blockIdx.x;
    if (l >= L) return;
    const T* __restrict__ c = cbT + l * Lay::SIZE;
    static_assert(Lay::SP == Lay::G + Lay::GN, "the team kernel copies G and SP as one range");
    const bool scan_ok = __builtin_amdgcn_readfirstlane((int)(c[Lay::SCANOK] > T(0))) != 0;
    const int nlev = __builtin_amdgcn_readfirstlane((int)c[Lay::NLEV]);
    const bool team_ok = scan_ok && nlev <= 5;                                         // uniform over the workgroup
    T* tile = tiles + (size_t)wave * 64 * STRIDE;
    T* tile_lane = tile + lane * STRIDE;
    const T* row = Ty + l * ld;
    const size_t t0 = (size_t)wave * SEG;
    const int n = (int)(Tlen - t0 < (size_t)SEG ? Tlen - t0 : (size_t)SEG);            // (nw = ceil(Tlen / SEG): every wave owns >= 1 tick)
    // A wave's life here is one dependent chain (little else runs on its SIMD to hide a wait), so every global load of the sweep is
    // issued up front: the wave's segment, the latent's tables (ONE copy per workgroup, into LDS: all waves sweep the same latent, and an
    // LDS read later costs ~100 cycles where a fetch from L2 / HBM costs 1-2 us), the replay's blocks of A (scalar loads)
    V raw[CK / EPV];
    ReplayConst<T, DB, J> rc;
    if (team_ok) {
#pragma unroll
        for (int r = 0; r < CK / EPV; r++) {
            const int e = (r * 64 + lane) * EPV;
            raw[r] = V{};
            if (e < n) raw[r] = nt_load(reinterpret_cast<const V*>(row + t0 + e));
        }
        for (int e = threadIdx.x; e < team_table_len<D>(); e += blockDim.x) tab[e] = c[Lay::G + e];
        load_replay_const<T, DB, J>(launder(c), c, lane, rc);
    }
    if (threadIdx.x == 0) dirty_any = 0;
    __syncthreads();
    T z[D];
    bool bad = false;
    if (team_ok) {
        // ---- A. stage in (chunk-major into the padded tile), chunk response, scan from a zero state ----
#pragma unroll
        for (int r = 0; r < CK / EPV; r++) {
            const int e = (r * 64 + lane) * EPV;
            T vals[EPV];
            unpack<T>(raw[r], vals);
#pragma unroll
            for (int q = 0; q < EPV; q++) if (e + q >= n) vals[q] = T(0);           // beyond the stream: inert zeros
            *reinterpret_cast<V*>(tile + (e / CK) * STRIDE + (e % CK)) = pack<T>(vals);
        }
        T g[NSG];
        load_slabs<T, NSG>(tab, lane, g);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < D; i++) z[i] = T(0);
        static_for<CK / EPV>([&](auto kvv) {
            constexpr int kv = decltype(kvv)::value;
            T yv[EPV];
            unpack<T>(*reinterpret_cast<const V*>(tile_lane + kv * EPV), yv);
            static_for<EPV>([&](auto qq) {
                constexpr int k = kv * EPV + decltype(qq)::value;
                bad = bad || (yv[decltype(qq)::value] != yv[decltype(qq)::value]);
                if constexpr (kXSkip & 1) { z[k % D] += yv[decltype(qq)::value]; return; }
                static_for<D>([&](auto ii) {
                    constexpr int e = k * D + decltype(ii)::value;
                    fmac_bc<e % 16>(z[decltype(ii)::value], g[e / 16], yv[decltype(qq)::value]);
                });
            });
            if constexpr (kv % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        });
        if (__builtin_amdgcn_ballot_w64(bad) != 0) { if (lane == 0) dirty_any = 1; }
        else {
            T t[D], sp[2][NSL];
            if (nlev > 0) load_slabs<T, NSL>(tab + Lay::GN, lane, sp[0]);
#pragma unroll
            for (int lv = 0; lv < 6; lv++) {
                if (lv >= nlev || (kXSkip & 2)) break;
                const int sh = 1 << lv, addr = ((lane - sh) & 63) * 4;
#pragma unroll
                for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, z[i]); t[i] = lane >= sh ? m : T(0); }
                if (lv + 1 < nlev) load_slabs<T, NSL>(tab + Lay::GN + (lv + 1) * Lay::LS, lane, sp[(lv + 1) & 1]);   // (next power: in flight during this level)
                matvec_bc<T, D, NSL>(sp[lv & 1], t, z);
            }
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < D; i++) e0end[wave * D + i] = z[i];
            }
        }
    }
    __syncthreads();
    if (!team_ok || dirty_any) {
        // ---- fallback: the whole stream by one wavefront (the SPLIT kernel's own treatment of such latents) ----
        if (wave != 0) return;
        filter_x_body<T, DB, J, WRITE, NLL, true, false>(Ty, Tlen, ld, cbT, cb64, xin0, x, yhat, nll, /*nslice=*/1, /*segs_per_slice=*/(int)((Tlen + SEG - 1) / SEG),
                                                         /*nll_part=*/nll, ldo, nullptr, nullptr, l, /*slice=*/0, lane, tile, carries);
        return;
    }
    // ---- B. the state entering this segment, its way into the lanes' start states, replay ----
    // q_j = M^j c_s: c_s is the same vector in every lane, so lane j applies the binary powers M^(2^k) of the set bits of j (they commute):
    // per level one uniform matrix-vector product and a select, and no lane exchange (a scan of (c_s, 0, 0, ..) gives the same vectors
    // through five ds_bpermute round trips, on a wave whose critical path this is)
    T xs[D];
    {
        T q[D];
#pragma unroll
        for (int i = 0; i < D; i++) q[i] = wave == 0 ? xin0[l * D + i] : e0end[(wave - 1) * D + i];     // (uniform address: one broadcast read)
        T sp[NSL];
        if (nlev > 0) load_slabs<T, NSL>(tab + Lay::GN, lane, sp);
#pragma unroll
        for (int lv = 0; lv < 6; lv++) {
            if (lv >= nlev || (kXSkip & 2)) break;
            T t[D];
#pragma unroll
            for (int i = 0; i < D; i++) t[i] = T(0);
            matvec_bc<T, D, NSL>(sp, q, t);                            // M^(2^lv) q
            if (lv + 1 < nlev) load_slabs<T, NSL>(tab + Lay::GN + (lv + 1) * Lay::LS, lane, sp);     // (the next power, in flight during the selects)
            const bool bit = (lane >> lv) & 1;
#pragma unroll
            for (int i = 0; i < D; i++) q[i] = bit ? t[i] : q[i];
        }
        // (a lane whose index has a bit at or above nlev set would need M^(2^nlev) or more: below 1e-20 / 1e-10 by the table's own criterion)
        const bool far = (lane >> nlev) != 0;
        const int addr1 = ((lane - 1) & 63) * 4;
#pragma unroll
        for (int i = 0; i < D; i++) { const T m = bperm<T>(addr1, z[i]); xs[i] = (lane >= 1 ? m : T(0)) + (far ? T(0) : q[i]); }
    }
    T xc[D];
    double acc = 0.0;
    unsigned nobs = 0;
    if constexpr (kXSkip & 4) {
#pragma unroll
        for (int i = 0; i < D; i++) xc[i] = read_lane(xs[i], 63);
    } else
    replay<T, DB, J, WRITE, NLL, true>(rc, tile_lane, carries + wave * D, lane, n, 0, xs, xc, acc, nobs);
    // ---- stage out ----
    if (WRITE) {
        wave_lds_fence();
        T* orow = yhat + l * ldo;
#pragma unroll
        for (int r = 0; r < CK / EPV; r++) {
            const int e = (r * 64 + lane) * EPV;
            if (e < n) nt_store(*reinterpret_cast<const V*>(tile + (e / CK) * STRIDE + (e % CK)), reinterpret_cast<V*>(orow + t0 + e));
        }
    }
    if (wave == nw - 1 && lane == 0) {
#pragma unroll
        for (int i = 0; i < D; i++) x[l * D + i] = xc[i];
    }
    if (NLL) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { acc += __shfl_xor(acc, o, 64); nobs += __shfl_xor(nobs, o, 64); }
        if (lane == 0) { part[2 * wave] = acc; part[2 * wave + 1] = (double)nobs; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0, nn = 0.0;
            for (int w = 0; w < nw; w++) { a += part[2 * w]; nn += part[2 * w + 1]; }      // in segment order (deterministic)
            const double* c64 = cb64 + l * Lay::SIZE;
            nll[l] = 0.5 * (a / c64[Lay::S] + nn * c64[Lay::LOGS]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// A segment WITH GAPS as a scan of the chunks' own affine maps (state dims up to kPairMaxDim; the second pass of the many-latent
// kernel does the same inline): a missing tick is x <- A x, so a chunk that holds one moves the state by a matrix of its own.  Every lane
// walks its chunk once with D + 1 vectors, gap-aware -- the response from a zero state, zr, and the D unit start states, mcol[c] = M e_c --
// and the pairs (M_j, z_j) are scanned over the lanes (Kogge-Stone, lane shifts by ds_bpermute, identity where a lane has no source).
// On return lane j holds the PREFIX map: the state after chunk j is  mcol x + zr  for a state x entering the segment -- which the
// caller may not know yet (the team kernel scans while its carry-in is still on its way).
template <typename T, int DB, int J, int CKC>
__device__ inline void chunk_maps_scan(uptr<T> cu, const T* __restrict__ c, const T* tile_lane, int lane, T (&mcol)[DB * J][DB * J], T (&zr)[DB * J]) {
    constexpr int D = DB * J;
    using Lay = XC<D>;
    const T ha = c[Lay::HA16 + (lane & 15)], kk = c[Lay::K16 + (lane & 15)];
    T ablk[J * DB * DB];
#pragma unroll
    for (int i = 0; i < J * DB * DB; i++) ablk[i] = cu[Lay::AB + i];
#pragma unroll
    for (int cidx = 0; cidx < D; cidx++)
#pragma unroll
        for (int i = 0; i < D; i++) mcol[cidx][i] = (i == cidx) ? T(1) : T(0);
#pragma unroll
    for (int i = 0; i < D; i++) zr[i] = T(0);
    auto tick_vec = [&](T (&xv)[D], T yin, bool miss) {
        T h0 = 0, h1 = 0;
        static_for<D>([&](auto ii) { constexpr int i = decltype(ii)::value; fmac_bc<i>(i % 2 == 0 ? h0 : h1, ha, xv[i]); });
        const T v = miss ? T(0) : yin - (h0 + h1);
        T xn[D];
#pragma unroll
        for (int j = 0; j < J; j++)
#pragma unroll
            for (int r = 0; r < DB; r++) {
                T sum = ablk[j * DB * DB + r * DB] * xv[j * DB];
#pragma unroll
                for (int q = 1; q < DB; q++) sum = fma(ablk[j * DB * DB + r * DB + q], xv[j * DB + q], sum);
                xn[j * DB + r] = sum;
            }
        static_for<D>([&](auto ii) { fmac_bc<decltype(ii)::value>(xn[decltype(ii)::value], kk, v); });
#pragma unroll
        for (int i = 0; i < D; i++) xv[i] = xn[i];
    };
#pragma unroll 1
    for (int k = 0; k < CKC; k++) {
        const T y = tile_lane[k];
        const bool miss = (y != y);
        tick_vec(zr, miss ? T(0) : y, miss);
#pragma unroll
        for (int cidx = 0; cidx < D; cidx++) tick_vec(mcol[cidx], T(0), miss);
    }
#pragma unroll 1
    for (int lv = 0; lv < 6; lv++) {
        const int sh = 1 << lv, addr = ((lane - sh) & 63) * 4;
        const bool has = lane >= sh;
        T zp[D], ncol[D][D];
#pragma unroll
        for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, zr[i]); zp[i] = has ? m : T(0); }
#pragma unroll
        for (int cidx = 0; cidx < D; cidx++) {      // new column c = M (partner's column c)
            T pc[D];
#pragma unroll
            for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, mcol[cidx][i]); pc[i] = has ? m : ((i == cidx) ? T(1) : T(0)); }
#pragma unroll
            for (int i = 0; i < D; i++) {
                T sacc = 0;
#pragma unroll
                for (int q = 0; q < D; q++) sacc = fma(mcol[q][i], pc[q], sacc);
                ncol[cidx][i] = sacc;
            }
        }
#pragma unroll
        for (int q = 0; q < D; q++)
#pragma unroll
            for (int i = 0; i < D; i++) zr[i] = fma(mcol[q][i], zp[q], zr[i]);     // z = M z_p + z (old M)
#pragma unroll
        for (int cidx = 0; cidx < D; cidx++)
#pragma unroll
            for (int i = 0; i < D; i++) mcol[cidx][i] = ncol[cidx][i];
    }
}

// The lane's chunk replayed from its true start state xs, gap-aware (ihgp.h:83-87: a missing tick is x <- A x, nothing counted); ticks
// from n on (beyond the stream) are neither written nor counted.  The filtered means go to the tile, the innovations to acc / nobs.
template <typename T, int DB, int J, int CKC, bool WRITE, bool NLL>
__device__ inline void replay_gaps(uptr<T> cu, const T* __restrict__ c, T* tile_lane, int lane, int n, T (&xs)[DB * J], double& acc, unsigned& nobs) {
    constexpr int D = DB * J;
    using Lay = XC<D>;
    const T ha = c[Lay::HA16 + (lane & 15)], kk = c[Lay::K16 + (lane & 15)];
    T ablk[J * DB * DB];
#pragma unroll
    for (int i = 0; i < J * DB * DB; i++) ablk[i] = cu[Lay::AB + i];
    double part = 0.0;
    unsigned cnt = 0;
#pragma unroll 1
    for (int k = 0; k < CKC; k++) {
        const bool live = lane * CKC + k < n;
        const T y = tile_lane[k];
        const bool miss = (y != y);
        T h0 = 0, h1 = 0, h2 = 0;
        static_for<D>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            fmac_bc<i>(i % 3 == 0 ? h0 : (i % 3 == 1 ? h1 : h2), ha, xs[i]);
        });
        const T v = miss ? T(0) : y - ((h0 + h1) + h2);
        if (NLL) {
            const bool counted = live && !miss;
            const double vd = counted ? (double)v : 0.0;
            part = fma(vd, vd, part);
            cnt += counted ? 1u : 0u;
        }
        T xn[D];
#pragma unroll
        for (int j = 0; j < J; j++)
#pragma unroll
            for (int r = 0; r < DB; r++) {
                T sum = ablk[j * DB * DB + r * DB] * xs[j * DB];
#pragma unroll
                for (int q = 1; q < DB; q++) sum = fma(ablk[j * DB * DB + r * DB + q], xs[j * DB + q], sum);
                xn[j * DB + r] = sum;
            }
        static_for<D>([&](auto ii) { fmac_bc<decltype(ii)::value>(xn[decltype(ii)::value], kk, v); });
#pragma unroll
        for (int i = 0; i < D; i++) xs[i] = live ? xn[i] : xs[i];                  // (beyond the stream the state stays: lane jl ends on the stream's last state)
        if (WRITE && live) tile_lane[k] = xn[0];
    }
    if (NLL) { acc += part; nobs += cnt; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// TEAM, CHUNK LENGTH AS A PARAMETER: kTeamCWaves = 8 wavefronts per latent -- two on every SIMD of the compute unit that holds the
// latent -- and the chunk length chosen so that eight segments of 64 chunks cover the stream (10^4 ticks: CK = 20, 1280-tick segments).
// Stage stamps of the first versions (tools/team_stamps.py, profiles/r03/team_stamps_*.log) showed the few-latents sweep bound by the
// ISSUE SLOTS OF ONE COMPUTE UNIT, not by latency: five 2048-tick wavefronts sit 2-1-1-1 on the four SIMDs, ten 1024-tick ones 3-3-2-2,
// and in both the youngest wavefront of the fullest SIMD finishes last, having waited for its elders at every stage.  Eight wavefronts
// balance the SIMDs, and the longest chunk that still gives eight segments keeps the scans' share (two per segment, whatever its length) lowest.
// The tables come from the latent's block as far as they do not depend on CK (the response table of a CK-tick chunk is the last CK rows of
// G; the replay's coefficients); the scan powers M^(2^k), M = AKHA^CK, k = 0..6, and the facts derived from them come from
// team_powers_kernel's table (below; filled at IHGP::update for banks of fewer than 1024 latents).
// No whole-stream fallback lives in this kernel; what the 32-tick team kernel hands back is handled here, exactly and in place, through a
// chain of LDS flags:
//   * every wave publishes the TRUE state after its segment, eend[w], and then done[w]; wave w takes its carry-in from eend[w-1] once
//     done[w-1] is set (wave 0 from the caller's start state: the chain always terminates; all waves of a workgroup are resident);
//   * a clean segment of a latent that decays publishes its zero-start end state right after its scan -- nobody waits;  one that does
//     not decay publishes e0 + M^64 c_w once its own carry-in is there (a serial chain of one matrix-vector product per segment);
//   * a segment with a missing tick -- or every segment of a latent that is not tame -- is walked tick by tick from its true carry-in
//     (sequential(), the many-latent kernel's own walk) and publishes the walk's end state: only such segments wait for each other.
// For fp32 (packed replay) at every state dimension and fp64 (scalar-operand replay) up to d = 8; fp64 d = 9, 12 keep the 32-tick team kernel.
#ifdef MOIHGP_TUNING
// tuning builds only: shader-clock stamps of workgroup 0's wavefronts at the stage boundaries (tools/team_stamps.py)
__device__ unsigned long long g_team_stamps[16][16];
#define MOIHGP_STAMP(i_) do { if ((i_) == 9) __builtin_amdgcn_s_waitcnt(0); if (blockIdx.x == 0 && lane == 0) { g_team_stamps[wave][i_] = __builtin_readcyclecounter(); if ((i_) == 0 || (i_) == 9) g_team_stamps[wave][(i_) == 0 ? 14 : 15] = wall_clock64(); } } while (0)
#else
#define MOIHGP_STAMP(i_) do { } while (0)
#endif
constexpr int kTeamCWaves = 8;
// tile row of one lane: CK ticks + padding such that the rows' 16-byte groups fall on distinct LDS banks (an odd number of groups per row)
template <typename T, int CK> constexpr int teamc_stride() { return ((CK + 16 / (int)sizeof(T)) / (16 / (int)sizeof(T))) % 2 == 1 ? CK + 16 / (int)sizeof(T) : CK + 32 / (int)sizeof(T); }
template <typename T, int D, int CK> constexpr int teamc_table_len() { return (CK * D + 15) / 16 * 16 + 7 * XC<D>::LS; }      // last CK rows of G | M^(1,2,..,64)
template <typename T, int D, int CK> constexpr size_t teamc_aux_offset(int nw) {    // tiles | e0 [nw][D] | eend [nw][D] | carries [nw][D] | tables
    return (((size_t)nw * (64 * teamc_stride<T, CK>() + 3 * D) + teamc_table_len<T, D, CK>()) * sizeof(T) + 15) / 16 * 16;
}
template <typename T, int D, int CK> constexpr size_t teamc_smem_bytes(int nw) {      // + sums [nw][2] | flags [nw]
    return teamc_aux_offset<T, D, CK>(nw) + 2 * (size_t)nw * sizeof(double) + (size_t)nw * sizeof(int);
}

template <typename T, int DB, int J, bool WRITE, bool NLL, int CK>
__global__ void __launch_bounds__(64 * kTeamCWaves)
filter_x_teamc_kernel(const T* __restrict__ Ty, size_t Tlen, size_t ld, size_t L, const T* __restrict__ cbT, const double* __restrict__ cb64,
                      const T* __restrict__ tpT /* team_powers_kernel's table in T */, const T* xin0, T* x, T* __restrict__ yhat, double* __restrict__ nll,
                      size_t ldo, int nw /* wavefronts = segments of 64 CK ticks */) {
    constexpr int D = DB * J, NN = D * D;
    using V = typename VecOf<T>::type;
    using Lay = XC<D>;
    constexpr int EPV = 16 / sizeof(T), STRIDE = teamc_stride<T, CK>(), SEG = 64 * CK;
    constexpr int NSL = Lay::LS / 16, GNC = (CK * D + 15) / 16 * 16, NSG = GNC / 16;
    static_assert(CK % EPV == 0 && CK <= kChunkX, "whole 16-byte groups; the response table is a tail of the 32-tick one");
    static_assert(ReplayConst<T, DB, J>::PK || ReplayConst<T, DB, J>::SOP || CK == kChunkX, "the DPP replay form is written for 32-tick chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char team_smem[];
    T* tiles = reinterpret_cast<T*>(team_smem);                                       // [nw][64 * STRIDE]
    T* e0s = tiles + (size_t)nw * 64 * STRIDE;                                         // [nw][D]  zero-start end state of a clean segment
    T* eend = e0s + (size_t)nw * D;                                                    // [nw][D]  TRUE state after the segment
    T* carries = eend + (size_t)nw * D;                                                // [nw][D]  carry-out slots of the replays
    T* tab = carries + (size_t)nw * D;                                                 // [GNC | 7 LS]
    double* part = reinterpret_cast<double*>(team_smem + teamc_aux_offset<T, D, CK>(nw));  // [nw][2]  per-wave sum of v^2, observed ticks
    int* done = reinterpret_cast<int*>(part + 2 * (size_t)nw);                         // [nw]     eend[w] is valid
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = blockIdx.x;
    if (l >= L) return;
    const T* __restrict__ c = cbT + l * Lay::SIZE;
    const T* __restrict__ tp = tpT + (l * kTeamNck + (CK - 16) / 4) * team_powers_len<D>();
    MOIHGP_STAMP(0);
    T* tile = tiles + (size_t)wave * 64 * STRIDE;
    T* tile_lane = tile + lane * STRIDE;
    const T* row = Ty + l * ld;
    const size_t t0 = (size_t)wave * SEG;
    const int n = (int)(Tlen - t0 < (size_t)SEG ? Tlen - t0 : (size_t)SEG);            // (nw = ceil(Tlen / SEG): every wave owns >= 1 tick)
    // every global load of the sweep up front: the wave's segment, the latent's tables (one copy per workgroup, into LDS), the replay's operands
    V raw[CK / EPV];
#pragma unroll
    for (int r = 0; r < CK / EPV; r++) {
        const int e = (r * 64 + lane) * EPV;
        raw[r] = V{};
        if (e < n) raw[r] = nt_load(reinterpret_cast<const V*>(row + t0 + e));
    }
    for (int e = threadIdx.x; e < teamc_table_len<T, D, CK>(); e += blockDim.x)
        tab[e] = e < CK * D ? c[Lay::G + (kChunkX - CK) * D + e] : e < GNC ? T(0) : tp[e - GNC];     // g_k of a CK-tick chunk: the table's last CK rows; the powers
    ReplayConst<T, DB, J> rc;
    load_replay_const<T, DB, J>(launder(c), c, lane, rc);
    T cin[D];                                                                          // (x may be xin0's own buffer: read before anyone can have written)
#pragma unroll
    for (int i = 0; i < D; i++) cin[i] = xin0[l * D + i];
    if (lane == 0) done[wave] = 0;
    __syncthreads();
    MOIHGP_STAMP(1);
    const bool scan_ok = __builtin_amdgcn_readfirstlane((int)(c[Lay::SCANOK] > T(0) && tp[7 * Lay::LS + 2] != T(0))) != 0;   // response table and powers tame
    const int nlev = __builtin_amdgcn_readfirstlane((int)tp[7 * Lay::LS]);              // levels of the scan that matter
    const bool decays = __builtin_amdgcn_readfirstlane((int)(tp[7 * Lay::LS + 1] != T(0))) != 0;    // M^64 negligible: segments chain through e0 alone
    const T* pw = tab + GNC;                                                           // pw + lv * LS: M^(2^lv)
    // ---- A. stage in, chunk response, (clean segments) scan from a zero state ----
#pragma unroll
    for (int r = 0; r < CK / EPV; r++) {
        const int e = (r * 64 + lane) * EPV;
        T vals[EPV];
        unpack<T>(raw[r], vals);
#pragma unroll
        for (int q = 0; q < EPV; q++) if (e + q >= n) vals[q] = T(0);               // beyond the stream: inert zeros
        *reinterpret_cast<V*>(tile + (e / CK) * STRIDE + (e % CK)) = pack<T>(vals);
    }
    MOIHGP_STAMP(2);
    T g[NSG];
    load_slabs<T, NSG>(tab, lane, g);
    wave_lds_fence();
    T z[D];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < D; i++) z[i] = T(0);
    static_for<CK / EPV>([&](auto kvv) {
        constexpr int kv = decltype(kvv)::value;
        T yv[EPV];
        unpack<T>(*reinterpret_cast<const V*>(tile_lane + kv * EPV), yv);
        static_for<EPV>([&](auto qq) {
            constexpr int k = kv * EPV + decltype(qq)::value;
            bad = bad || (yv[decltype(qq)::value] != yv[decltype(qq)::value]);
            static_for<D>([&](auto ii) {
                constexpr int e = k * D + decltype(ii)::value;
                fmac_bc<e % 16>(z[decltype(ii)::value], g[e / 16], yv[decltype(qq)::value]);
            });
        });
    });
    MOIHGP_STAMP(3);
    const bool gaps = __builtin_amdgcn_ballot_w64(bad) != 0;                             // wave-uniform
    constexpr bool PAIRS = D <= kPairMaxDim;                                           // a segment with gaps as a scan of its chunks' own maps
    const bool walk = !scan_ok || (gaps && !PAIRS);
    T xc[D];
    double acc = 0.0;
    unsigned nobs = 0;
    bool done_by_pairs = false;
    if constexpr (PAIRS) {
        if (scan_ok && gaps) {
            // the chunk maps and their prefix products need no carry-in: scanned while it is still on its way; then the state after
            // every chunk is one D x D product away, this segment's end state is published at once, and one gap-aware replay finishes it
            T mcol[D][D], zr[D];
            chunk_maps_scan<T, DB, J, CK>(launder(c), c, tile_lane, lane, mcol, zr);
            if (wave != 0) {
                while (__hip_atomic_load(&done[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int i = 0; i < D; i++) cin[i] = eend[(wave - 1) * D + i];
            }
            T yj[D];                                                                   // state after this lane's chunk
#pragma unroll
            for (int i = 0; i < D; i++) yj[i] = zr[i];
#pragma unroll
            for (int q = 0; q < D; q++)
#pragma unroll
                for (int i = 0; i < D; i++) yj[i] = fma(mcol[q][i], cin[q], yj[i]);
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < D; i++) eend[wave * D + i] = yj[i];
            }
            wave_lds_fence();
            if (lane == 0) __hip_atomic_store(&done[wave], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            T xs[D];
            const int addr1 = ((lane - 1) & 63) * 4;
#pragma unroll
            for (int i = 0; i < D; i++) { const T m = bperm<T>(addr1, yj[i]); xs[i] = lane >= 1 ? m : cin[i]; }
            replay_gaps<T, DB, J, CK, WRITE, NLL>(launder(c), c, tile_lane, lane, n, xs, acc, nobs);
            const int jl = __builtin_amdgcn_readfirstlane((n - 1) / CK);               // the lane whose chunk holds the stream's / segment's last tick
#pragma unroll
            for (int i = 0; i < D; i++) xc[i] = read_lane(xs[i], jl);
            done_by_pairs = true;
        }
    }
    if (!done_by_pairs) {
    if (!walk) {
        T t[D], sp[2][NSL];
        load_slabs<T, NSL>(pw, lane, sp[0]);
#pragma unroll
        for (int lv = 0; lv < 6; lv++) {
            if (lv >= nlev) break;
            const int sh = 1 << lv, addr = ((lane - sh) & 63) * 4;
#pragma unroll
            for (int i = 0; i < D; i++) { const T m = bperm<T>(addr, z[i]); t[i] = lane >= sh ? m : T(0); }
            if (lv + 1 < nlev) load_slabs<T, NSL>(pw + (lv + 1) * Lay::LS, lane, sp[(lv + 1) & 1]);   // (next power: in flight during this level)
            matvec_bc<T, D, NSL>(sp[lv & 1], t, z);
        }
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < D; i++) { e0s[wave * D + i] = z[i]; if (decays) eend[wave * D + i] = z[i]; }
        }
        if (decays) {                                                                  // true end state = e0 to rounding: publish, nobody has to wait for us
            wave_lds_fence();
            if (lane == 0) __hip_atomic_store(&done[wave], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    MOIHGP_STAMP(4);
    // ---- B. the state entering this segment ----
    if (wave != 0) {
        while (__hip_atomic_load(&done[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < D; i++) cin[i] = eend[(wave - 1) * D + i];                  // (uniform address: one broadcast read)
    }
    MOIHGP_STAMP(5);
    if (walk) {
        // tick by tick from the true carry-in; outputs into the tile, sum of v^2 and the count in lane 0
#pragma unroll
        for (int i = 0; i < D; i++) xc[i] = cin[i];
        bool any_lost = false;
#pragma unroll
        for (int i = 0; i < D; i++) any_lost = any_lost || !((xc[i] - xc[i]) == T(0));
        if (!scan_ok && any_lost) {                                                    // an unstable latent that has left the format: NaN from here on, as the
            const T qnan = __builtin_nan("");                                          // many-latent kernel does (its walk would give the same, 10 x slower)
            if (WRITE) {
#pragma unroll 4
                for (int k = 0; k < CK; k++) tile_lane[k] = qnan;
            }
#pragma unroll
            for (int i = 0; i < D; i++) xc[i] = qnan;
            if (NLL) acc = __builtin_nan("");
        } else {
            sequential<T, DB, J, WRITE, NLL, CK>(c, tile, STRIDE, n, 0, lane, xc, acc, nobs);
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < D; i++) eend[wave * D + i] = xc[i];
        }
        wave_lds_fence();
        if (lane == 0) __hip_atomic_store(&done[wave], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        if (!decays) {                                                                 // true end state = e0 + M^64 cin: our successor is waiting for it
            T t[D], sp[NSL];
            load_slabs<T, NSL>(pw + 6 * Lay::LS, lane, sp);
#pragma unroll
            for (int i = 0; i < D; i++) t[i] = e0s[wave * D + i];
            matvec_bc<T, D, NSL>(sp, cin, t);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < D; i++) eend[wave * D + i] = t[i];
            }
            wave_lds_fence();
            if (lane == 0) __hip_atomic_store(&done[wave], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // q_j = M^j cin from the binary powers of the lane's own index (cin is uniform: no lane exchange), start states, replay
        T xs[D];
        {
            T q[D], sp[NSL];
#pragma unroll
            for (int i = 0; i < D; i++) q[i] = cin[i];
            load_slabs<T, NSL>(pw, lane, sp);
#pragma unroll
            for (int lv = 0; lv < 6; lv++) {
                if (lv >= nlev) break;
                T t[D];
#pragma unroll
                for (int i = 0; i < D; i++) t[i] = T(0);
                matvec_bc<T, D, NSL>(sp, q, t);
                if (lv + 1 < nlev) load_slabs<T, NSL>(pw + (lv + 1) * Lay::LS, lane, sp);
                const bool bit = (lane >> lv) & 1;
#pragma unroll
                for (int i = 0; i < D; i++) q[i] = bit ? t[i] : q[i];
            }
            const bool far = (lane >> nlev) != 0;                                      // would need M^(2^nlev) or more: below the table's own criterion
            const int addr1 = ((lane - 1) & 63) * 4;
#pragma unroll
            for (int i = 0; i < D; i++) { const T m = bperm<T>(addr1, z[i]); xs[i] = (lane >= 1 ? m : T(0)) + (far ? T(0) : q[i]); }
        }
        MOIHGP_STAMP(6);
        replay<T, DB, J, WRITE, NLL, true, CK>(rc, tile_lane, carries + wave * D, lane, n, 0, xs, xc, acc, nobs);
    }
    }
    MOIHGP_STAMP(7);
    // ---- stage out ----
    if (WRITE) {
        wave_lds_fence();
        T* orow = yhat + l * ldo;
#pragma unroll
        for (int r = 0; r < CK / EPV; r++) {
            const int e = (r * 64 + lane) * EPV;
            if (e < n) nt_store(*reinterpret_cast<const V*>(tile + (e / CK) * STRIDE + (e % CK)), reinterpret_cast<V*>(orow + t0 + e));
        }
    }
    MOIHGP_STAMP(8);
    if (wave == nw - 1 && lane == 0) {
#pragma unroll
        for (int i = 0; i < D; i++) x[l * D + i] = xc[i];
    }
    if (NLL) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { acc += __shfl_xor(acc, o, 64); nobs += __shfl_xor(nobs, o, 64); }
        if (lane == 0) { part[2 * wave] = acc; part[2 * wave + 1] = (double)nobs; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0, nn = 0.0;
            for (int w = 0; w < nw; w++) { a += part[2 * w]; nn += part[2 * w + 1]; }      // in segment order (deterministic)
            const double* c64 = cb64 + l * Lay::SIZE;
            nll[l] = 0.5 * (a / c64[Lay::S] + nn * c64[Lay::LOGS]);
        }
    }
    MOIHGP_STAMP(9);
}

#if defined(MOIHGP_TUNING) && MOIHGP_X_TU == 32        // (tools/team_stamps.py runs the d = 6 model)
extern "C" int moihgp_tuning_team_stamps(unsigned long long* out /* [16][16] */) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_team_stamps), sizeof(unsigned long long) * 256);
}
#endif

// nll[l] = sum over the slices, in slice order (deterministic)
__global__ void __launch_bounds__(256) sum_slices_kernel(const double* __restrict__ part, size_t L, int nslice, double* __restrict__ nll) {
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double s = 0.0;
    for (int k = 0; k < nslice; k++) s += part[l * nslice + k];
    nll[l] = s;
}

// The same plus the scalar the optimiser consumes (moihgp.h:684): per-latent sums in slice order, then their total in a fixed order
// (thread-strided partial sums, a butterfly per wavefront, the wavefront sums in order) -- one workgroup, one launch instead of two.
__global__ void __launch_bounds__(1024) sum_slices_total_kernel(const double* __restrict__ part, size_t L, int nslice, double* __restrict__ nll,
                                                                double* __restrict__ total) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (size_t l = tid; l < L; l += 1024) {
        double v = 0.0;
        for (int k = 0; k < nslice; k++) v += part[l * nslice + k];
        nll[l] = v;
        s += v;
    }
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

template <typename T, int DB, int J, int WPB, bool SPLIT>
int launch_x(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, const void* xin, void* x, void* yhat, double* nll,
             hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, int nslice, int segs_per_slice, double* nll_part, size_t ldo,
             int* link_flags = nullptr, double* link_state = nullptr, double* total = nullptr,
             int pass_mode = 0 /* 0: first pass + second (LINKS) pass; 1: the first only; 2: the second only;
                                    4: the first only, writing predicted observations HA x instead of filtered means (no NLL);
                                    5: the imputation sweeps of the latents the first pass flagged (filter_x_gaps_a / _b_kernel) */,
             const GapArgs* ga = nullptr) {
    dim3 block(64 * WPB), grid(SPLIT ? (unsigned)L : (unsigned)((L + WPB - 1) / WPB), SPLIT ? (unsigned)nslice : 1u);
    const T* ty = static_cast<const T*>(Ty);
    const T* xi = static_cast<const T*>(xin);
    T* xs = static_cast<T*>(x);
    T* yh = static_cast<T*>(yhat);
    if (SPLIT) link_flags = nullptr;
    // (link_flags: zero when the handle allocates them, set by the first pass, cleared again by the second as it takes a latent over)
#define MOIHGP_X_LAUNCH(W_, N_)                                                                                                              \
    do {                                                                                                                                     \
        if (pass_mode != 2)                                                                                                                  \
            hipExtLaunchKernelGGL((filter_x_kernel<T, DB, J, W_, N_, WPB, SPLIT, false>), grid, block, 0, stream, ev0, ev1, 0,                \
                                  ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, nslice, segs_per_slice, nll_part, ldo, link_flags, link_state); \
        if constexpr (!SPLIT) {                                                                                                              \
            if (link_flags && (pass_mode == 0 || pass_mode == 2))  /* second pass: the latents stopped at a segment with gaps (none: the grid exits at once) */ \
                hipLaunchKernelGGL((filter_x_kernel<T, DB, J, W_, N_, WPB, SPLIT, true>), grid, block, 0, stream,                             \
                                   ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, nslice, segs_per_slice, nll_part, ldo, link_flags, link_state); \
        }                                                                                                                                    \
    } while (0)
    if constexpr (!SPLIT) {
        if (pass_mode == 4) {
            hipLaunchKernelGGL((filter_x_kernel<T, DB, J, true, false, WPB, SPLIT, false, true>), grid, block, 0, stream,
                               ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, nslice, segs_per_slice, nll_part, ldo, link_flags, link_state);
            hipError_t e4 = hipGetLastError();
            if (e4 != hipSuccess) { set_last_error("filter_x_kernel launch: %s", hipGetErrorString(e4)); return 2; }
            return 0;
        }
        if constexpr (J >= 2) {
            if (pass_mode == 5) {
                hipLaunchKernelGGL((filter_x_gaps_a_kernel<T, DB, J, WPB>), grid, block, 0, stream, ty, Tlen, ld, L, cbT, cb64, xi, (const int*)link_flags,
                                   (const double*)link_state, static_cast<const T*>(ga->imp), ga->gpos, static_cast<T*>(ga->gval), static_cast<T*>(ga->gw), ga->gcap, ga->gstat);
#define MOIHGP_X_GAPS(W_, N_) hipLaunchKernelGGL((filter_x_gaps_b_kernel<T, DB, J, W_, N_, WPB>), grid, block, 0, stream, ty, Tlen, ld, L, cbT, cb64, xi, xs, yh, nll, ldo, \
                                                 link_flags, (const double*)link_state, (const int*)ga->gpos, static_cast<const T*>(ga->gw), ga->gcap, (const int*)ga->gstat)
                if (yhat && nll) MOIHGP_X_GAPS(true, true);
                else if (yhat) MOIHGP_X_GAPS(true, false);
                else if (nll) MOIHGP_X_GAPS(false, true);
                else { set_last_error("the imputation sweep needs an output"); return 1; }
#undef MOIHGP_X_GAPS
                hipError_t e5 = hipGetLastError();
                if (e5 != hipSuccess) { set_last_error("filter_x_gaps kernels launch: %s", hipGetErrorString(e5)); return 2; }
                return 0;
            }
        }
    }
    if (yhat && nll) MOIHGP_X_LAUNCH(true, true);
    else if (yhat) MOIHGP_X_LAUNCH(true, false);
    else if (nll) MOIHGP_X_LAUNCH(false, true);
    else MOIHGP_X_LAUNCH(false, false);
#undef MOIHGP_X_LAUNCH
    if (SPLIT && nll && total) hipLaunchKernelGGL(sum_slices_total_kernel, dim3(1), dim3(1024), 0, stream, nll_part, L, nslice, nll, total);
    else if (SPLIT && nll) hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, stream, nll_part, L, nslice, nll);
    else if (nll && total) launch_nll_total(nll, L, total, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("filter_x_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

template <typename T, int DB, int J>
int launch_x_team(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, const void* xin, void* x, void* yhat, double* nll,
                  hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, size_t ldo, double* total, int nw) {
    const size_t smem = team_smem_bytes<T, DB * J>(nw);
    dim3 block(64 * nw), grid((unsigned)L);
#define MOIHGP_TEAM_LAUNCH(W_, N_)                                                                                                                   \
    do {                                                                                                                                             \
        auto kfn = filter_x_team_kernel<T, DB, J, W_, N_>;                                                                                           \
        static size_t attr_set = 48 * 1024;  /* (more dynamic LDS than the default limit needs the attribute: raised on demand, per instantiation) */    \
        if (smem > attr_set) { MOIHGP_HIP_FATAL(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr_set = smem; } \
        hipExtLaunchKernelGGL(kfn, grid, block, smem, stream, ev0, ev1, 0, (const T*)Ty, Tlen, ld, L, cbT, cb64, (const T*)xin, (T*)x, (T*)yhat, nll, ldo, nw); \
    } while (0)
    if (yhat && nll) MOIHGP_TEAM_LAUNCH(true, true);
    else if (yhat) MOIHGP_TEAM_LAUNCH(true, false);
    else if (nll) MOIHGP_TEAM_LAUNCH(false, true);
    else MOIHGP_TEAM_LAUNCH(false, false);
#undef MOIHGP_TEAM_LAUNCH
    if (nll && total) launch_nll_total(nll, L, total, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("filter_x_team_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

template <typename T, int DB, int J, int CK>
int launch_x_teamc(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, const T* tpT, const void* xin, void* x, void* yhat, double* nll,
                   hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, size_t ldo, double* total, int nw) {
    const size_t smem = teamc_smem_bytes<T, DB * J, CK>(nw);
    dim3 block(64 * nw), grid((unsigned)L);
#define MOIHGP_TEAMC_LAUNCH(W_, N_)                                                                                                                  \
    do {                                                                                                                                             \
        auto kfn = filter_x_teamc_kernel<T, DB, J, W_, N_, CK>;                                                                                      \
        static size_t attr_set = 48 * 1024;                                                                                                          \
        if (smem > attr_set) { MOIHGP_HIP_FATAL(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr_set = smem; } \
        hipExtLaunchKernelGGL(kfn, grid, block, smem, stream, ev0, ev1, 0, (const T*)Ty, Tlen, ld, L, cbT, cb64, tpT, (const T*)xin, (T*)x, (T*)yhat, nll, ldo, nw); \
    } while (0)
    if (yhat && nll) MOIHGP_TEAMC_LAUNCH(true, true);
    else if (yhat) MOIHGP_TEAMC_LAUNCH(true, false);
    else if (nll) MOIHGP_TEAMC_LAUNCH(false, true);
    else MOIHGP_TEAMC_LAUNCH(false, false);
#undef MOIHGP_TEAMC_LAUNCH
    if (nll && total) launch_nll_total(nll, L, total, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("filter_x_teamc_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

// workgroups of the chunk-templated team kernel a compute unit holds at once (registers -- the gap path keeps a D x D map per lane --, LDS, waves),
// as the runtime computes it; asked once per instantiation and wavefront count
template <typename T, int DB, int J, int CK>
int teamc_blocks_per_cu(int nw, size_t smem, bool w, bool n) {
    static std::atomic<int> cache[4][kTeamCWaves + 1] = {};          // (sweeps may be issued from several host threads: the answer is the same, the write atomic)
    std::atomic<int>& slot = cache[(w ? 2 : 0) + (n ? 1 : 0)][nw];
    if (slot.load(std::memory_order_relaxed) == 0) {
        const void* fn = w ? (n ? reinterpret_cast<const void*>(filter_x_teamc_kernel<T, DB, J, true, true, CK>) : reinterpret_cast<const void*>(filter_x_teamc_kernel<T, DB, J, true, false, CK>))
                           : (n ? reinterpret_cast<const void*>(filter_x_teamc_kernel<T, DB, J, false, true, CK>) : reinterpret_cast<const void*>(filter_x_teamc_kernel<T, DB, J, false, false, CK>));
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * nw, smem) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
        slot.store(nb, std::memory_order_relaxed);
    }
    return slot.load(std::memory_order_relaxed);
}

// the chunk-templated team kernel for a stream of Tlen ticks, if one of its chunk lengths gives 2 .. kTeamCWaves segments that fit a compute unit
// (returns -1 if none does: the caller goes on to the other forms)
template <typename T, int DB, int J>
int try_x_teamc(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, const T* tpT, const void* xin, void* x, void* yhat, double* nll,
                hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, size_t ldo, double* total, int team_mode) {
    if (!tpT) return -1;
    if constexpr (ReplayConst<T, DB, J>::PK || ReplayConst<T, DB, J>::SOP) {
        int rc = -1;
        auto attempt = [&](auto ckk) {
            constexpr int CK = decltype(ckk)::value;
            if (rc != -1 || Tlen > 64 * (size_t)CK * kTeamCWaves) return;               // (ascending CK: the shortest chunk that covers the stream in eight segments)
            const size_t nw = (Tlen + 64 * (size_t)CK - 1) / (64 * (size_t)CK);
            if (nw < 2) return;
            const size_t smem = teamc_smem_bytes<T, DB * J, CK>((int)nw);
            if (smem > 150 * 1024) { rc = -2; return; }                                  // (a longer chunk needs more LDS still)
            const size_t per_cu = (size_t)teamc_blocks_per_cu<T, DB, J, CK>((int)nw, smem, yhat != nullptr, nll != nullptr);
            // up to TWO rounds of workgroups over the 256 compute units (measured at 10^4 ticks, profiles/r03/midL_team_vs_split.log: d = 3 fp64 at 320 /
            // 384 / 512 latents 17.7 / 18.4 / 20.4 us against 22.7 / 24.0 / 27.0 with the time split, d = 6 fp32 at 512 19.0 against 22.5; three rounds are
            // level, four lose)
            if (!(team_mode == 1 || L <= 2 * 256 * per_cu)) { rc = -2; return; }
            rc = launch_x_teamc<T, DB, J, CK>(Ty, Tlen, ld, L, cbT, cb64, tpT, xin, x, yhat, nll, stream, ev0, ev1, ldo, total, (int)nw);
        };
        attempt(std::integral_constant<int, 16>{});
        attempt(std::integral_constant<int, 20>{});
        attempt(std::integral_constant<int, 24>{});
        attempt(std::integral_constant<int, 28>{});
        attempt(std::integral_constant<int, 32>{});
        return rc == -2 ? -1 : rc;
    } else {
        return -1;
    }
}

template <typename T, int DB, int J>
int launch_xd(const void* Ty, size_t Tlen, size_t ld, size_t L, const T* cbT, const double* cb64, const void* xin, void* x, void* yhat, double* nll,
              hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, double* scratch, size_t scratch_len, int force_slices, size_t ldo,
              int* link_flags, double* link_state, double* total, int env_links, int team_mode, const T* tpT) {
    constexpr size_t SEG = 64 * (size_t)kChunkX;
    if (L >= 1024) {
        // chunks with a gap per segment up to which the broken-link stages of the second pass beat the tick-by-tick walk (measured,
        // tools/filternan.py: a stage costs one scan + one replay, the second pass of the fp64 d = 12 kernel runs one wave per SIMD)
        // force_slices < -1 (missing ticks by imputation, filter_x_gaps_a / _b_kernel): -2 = the first pass alone, handing over EVERY latent that holds a
        // gap; -3 = the second pass alone; -6 = the first pass alone, writing predicted observations HA x instead of filtered means (the filters'
        // impulse responses); -7 = the imputation sweep of the latents the first pass flagged (scratch then points at a host-side GapArgs)
        const int pass_mode = force_slices == -2 ? 1 : force_slices == -3 ? 2 : force_slices == -6 ? 4 : force_slices == -7 ? 5 : 0;
        if (pass_mode >= 4)
            return launch_x<T, DB, J, 4, false>(Ty, Tlen, ld, L, cbT, cb64, xin, x, yhat, nll, stream, ev0, ev1, 1, 0, nullptr, ldo,
                                                pass_mode == 5 ? link_flags : nullptr, link_state, nullptr, pass_mode, reinterpret_cast<const GapArgs*>(scratch));
        const int max_links = pass_mode == 1 ? 64 : (env_links >= 0 ? env_links : (DB * J <= kPairMaxDim ? 64 : ((sizeof(T) == 8 && DB * J > 9) ? 3 : 32)));
        return launch_x<T, DB, J, 4, false>(Ty, Tlen, ld, L, cbT, cb64, xin, x, yhat, nll, stream, ev0, ev1, 1, max_links, nullptr, ldo,
                                            (max_links > 0 && link_state) ? link_flags : nullptr, link_state, total, pass_mode);
    }
    const size_t nseg = (Tlen + SEG - 1) / SEG;
    // few latents, a stream of 2 .. kTeamWaves segments: one workgroup per latent, one wavefront per segment (filter_x_team_kernel), as long as
    // all workgroups are resident at once (LDS: a wave's tile is 9 / 17 KB)
    // Left to itself (team_mode -1) fp32 at d >= 8 keeps the time split below: measured at 256 latents x 10^4 ticks (tools/smallnan.py,
    // profiles/r03/smallnan_*.log) it is as fast on streams without gaps (d = 12: 18.4 against 17.8-18.4 us, d = 9: 16.9 against 17.1) and
    // four times faster on streams with gaps (350 us against 1300-1700: the split's slices work on a gappy latent side by side, a team
    // kernel walks its segments one after the other, and the chunk-map scan that fixes this up to d = 6 needs a D x D map per lane).
    if (team_mode == -1 && DB * J > kPairMaxDim && sizeof(T) == 4) team_mode = 0;
    // ... eight wavefronts and the chunk length to match, where the replay takes the chunk length (fp32; fp64 up to d = 8)
    // (team_mode 2 = the 32-tick team kernel only)
    if (team_mode != 0 && team_mode != 2 && force_slices == 0) {
        const int rc = try_x_teamc<T, DB, J>(Ty, Tlen, ld, L, cbT, cb64, tpT, xin, x, yhat, nll, stream, ev0, ev1, ldo, total, team_mode);
        if (rc != -1) return rc;
    }
    if (team_mode != 0 && force_slices == 0 && nseg >= 2 && nseg <= (size_t)kTeamWaves) {
        const size_t smem = team_smem_bytes<T, DB * J>((int)nseg);
        size_t per_cu = (160 * 1024) / smem;
        if (per_cu > 2048 / (64 * nseg)) per_cu = 2048 / (64 * nseg);
        if (smem <= 150 * 1024 && (team_mode == 1 || team_mode == 2 || L <= 256 * per_cu))
            return launch_x_team<T, DB, J>(Ty, Tlen, ld, L, cbT, cb64, xin, x, yhat, nll, stream, ev0, ev1, ldo, total, (int)nseg);
    }
    // otherwise: one wavefront per workgroup, and the stream cut into time slices (one wavefront each) while that adds
    // wavefronts the chip can still use
    size_t want = force_slices > 0 ? (size_t)force_slices : (2048 + L - 1) / L;
    if (want > nseg) want = nseg;
    if (want * L > scratch_len) want = scratch_len / L;
    if (want < 1) want = 1;                                            // (one slice = the whole stream: same kernel)
    const size_t per = nseg ? (nseg + want - 1) / want : 1;
    // slices after the first own per * SEG ticks minus their warm-up (at most CK * 32 ticks: filter_x_kernel): enough of them for the
    // longest warm-up, within the scratch the caller provided
    size_t n = 1;
    if (Tlen > per * SEG) {
        const size_t own_min = per * SEG - (size_t)kChunkX * 32;
        n = 1 + (Tlen - per * SEG + own_min - 1) / own_min;
    }
    if (n * L > scratch_len) return launch_x<T, DB, J, 1, true>(Ty, Tlen, ld, L, cbT, cb64, xin, x, yhat, nll, stream, ev0, ev1, 1, (int)nseg, scratch, ldo, nullptr, nullptr, total);
    return launch_x<T, DB, J, 1, true>(Ty, Tlen, ld, L, cbT, cb64, xin, x, yhat, nll, stream, ev0, ev1, (int)n, (int)per, scratch, ldo, nullptr, nullptr, total);
}

}  // namespace

// this translation unit's model (DB, J) = (MOIHGP_X_TU / 10, MOIHGP_X_TU % 10): both precisions behind one entry (stack_dispatch.hip picks the unit)
#define MOIHGP_X_CAT2(a_, b_) a_##b_
#define MOIHGP_X_CAT(a_, b_) MOIHGP_X_CAT2(a_, b_)
int MOIHGP_X_CAT(launch_filter_x_, MOIHGP_X_TU)(int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const float* cb32,
                           const void* xin, void* x, void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                           double* scratch, size_t scratch_len, int force_slices, size_t ldo, int* link_flags, double* link_state, double* total, int max_links, int team_mode,
                           const double* tp64, const float* tp32) {
    constexpr int DBB = MOIHGP_X_TU / 10, JJ = MOIHGP_X_TU % 10;
    if constexpr (JJ == 1) {
        // the reference's own models (one component); force_slices -1 = the few-latents team kernel or nothing (the caller carries on with recursion.hip)
        if (force_slices == -1) {
            if (L == 0 || team_mode == 0) return -1;
            return dtype == 0 ? try_x_teamc<double, DBB, 1>(Ty, T, ld, L, cb64, cb64, tp64, xin, x, yhat, nll, stream, ev0, ev1, ldo ? ldo : ld, total, team_mode)
                              : try_x_teamc<float, DBB, 1>(Ty, T, ld, L, cb32, cb64, tp32, xin, x, yhat, nll, stream, ev0, ev1, ldo ? ldo : ld, total, team_mode);
        }
    }
    return dtype == 0 ? launch_xd<double, DBB, JJ>(Ty, T, ld, L, cb64, cb64, xin, x, yhat, nll, stream, ev0, ev1, scratch, scratch_len, force_slices, ldo, link_flags, link_state, total, max_links, team_mode, tp64)
                      : launch_xd<float, DBB, JJ>(Ty, T, ld, L, cb32, cb64, xin, x, yhat, nll, stream, ev0, ev1, scratch, scratch_len, force_slices, ldo, link_flags, link_state, total, max_links, team_mode, tp32);
}

}  // namespace moihgp
