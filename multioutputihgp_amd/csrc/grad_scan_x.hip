// grad_scan_x.hip -- the sensitivity / gradient sweep (ihgp.h:37-57 with :212-222, A2 + A5) for STACKED models over LONG
// streams, parallel in time.  (grad_x.hip walks a stream tick by tick, every matrix operand through LDS: right for the learners'
// windows, 96 ms for 4096 latents x 10^4 ticks at D = 12.)
//
// Mapping as in recursion_x.hip: one wavefront owns one latent, a segment is 64 lanes x 32 ticks, lane j owns chunk j;
// wave-uniform tables are "slabs" read through DPP row broadcasts, the diagonal blocks of A sit in SGPRs.  What makes the sweep
// affordable is the INNOVATION FORM of the sensitivity recursion.  With v = y - HA x (ihgp.h:206) the step ihgp.h:50 is
// x' = A x + K v, and differentiating that instead of x' = AKHA x + K y gives
//     dv_p  = -(H dA_p) x - HA dx_p                                    (ihgp.h:218)
//     dx_p' = A dx_p + dA_p x + dK_p v + K dv_p                        (== dAKHA_p x + AKHA dx_p + dK_p y, ihgp.h:54)
// in which A is block diagonal (J blocks of DB x DB: a stacked model is a sum of independent components) and dA_p is ONE such
// block (p a lengthscale) or zero: about 8 D + 2 D DB multiply-adds per tick and parameter instead of the 2 D^2 + 3 D of the
// dense form (D = 12: 170 against 324).
//
// Per segment (all arithmetic fp64, whatever the stream's type):
//   1. z_j = sum_k g_k y_k, 64-lane scan with the powers of M = AKHA^32  ->  start state x_j of every chunk       (as the filter)
//   2. replay x over the chunk: v_k, sum v^2; v replaces y in the LDS tile (everything below needs v, not y)
//   3. the ADJOINT of the chunk, one backward walk over the innovations, lam_(k-1) = A^T lam_k + HA^T (v_k - K . lam_k) from lam = 0: it gives
//      r_j = lam_(-1) = sum_k v_k (HA AKHA^k)^T, what a START sensitivity of the chunk contributes to sum_k v_k HA dx(k), per unit, and
//      w_j = sum_i v_i lam_i, which turns a forcing dK_p v into sum_k v_k HA dz_p(k) = w . dK_p
//   4a. for every parameter p that does not move A (dA_p = 0: the magnitudes, the noise variance): no replay at all -- the chunk-end dz_p is
//      the response of the innovations to the table gp_p[k] = AKHA^(CK-1-k) dK_p (gp_table_kernel), and s = -w . dK_p
//   4b. for every other parameter p (a lengthscale): replay (x, dz) from dz = 0 -- dz is the chunk-LOCAL sensitivity, forced by dA_p x + dK_p v and fed
//      back through K dv -- collecting s = sum_k v_k dv_k(local); the chunk-end dz are scanned with the SAME powers of M (the
//      homogeneous part of the recursion is AKHA, whatever p) to the true start sensitivity dx_j of every chunk, and by
//      linearity  sum_k v_k dv_k = s - r_j . dx_j.  No table of d(AKHA^n)/dp is needed anywhere.
//   5. (streams of means wanted) one more replay of x writes them into the tile.
// The kernel takes the stream's whole chunks; the last T mod 32 ticks, and whole latents whose stream holds missing ticks or
// whose scan tables are unusable (SCANOK == 0), are left to grad_x_kernel (flags[l] = 1), which continues from the carried
// (x, dx) and adds to nll / grad.
//
// Cost: about (60 + 170 P) D/12 multiply-adds per tick in the replays, 10 scans of <= 6 levels x D^2 per 32 ticks, 2 D per tick in
// the two response sums: VALU-bound, 16 B/tick of traffic at most.  DESIGN.md 3.8.
#include "x_common.h"

namespace moihgp {
namespace {

constexpr int kFewGaps = 3;                                          // chunks with a gap per window up to which the window is cut at them (else walked whole)
constexpr int kGxWaves = 2;                                           // wavefronts per workgroup (LDS: 18.3 KB each)
constexpr int kGxTables = 9;                                          // slab tables per latent in the table buffer: gp_p, p < P <= 9

template <int D> struct GxLds {
    static constexpr int CK = kChunkX, STRIDE = CK + 2;
    static constexpr int HPN = (CK * D + 15) / 16 * 16;
    double tile[64 * STRIDE];        // the segment, chunk per lane row, padded
    double carry[10 * D];            // carried x and sensitivities dx_p between segments ([0]: x, [p + 1]: dx_p)
    double gacc[9];                  // sum over the stream of v dv_p
};

// gp_p[k] = AKHA^(CK-1-k) dK_p, k < CK, per latent and hyper-parameter, as slab tables [P][k][i]: for a parameter that
// does not move A (dA_p = 0: the magnitudes and the noise variance, matern52ss.h:61-63) the chunk-local sensitivity obeys
// dz' = AKHA dz + dK_p v, so its value at the chunk's end is the chunk response of the innovations to this table -- exactly as the
// state's is the response of the observations to g_k = AKHA^(CK-1-k) K.  Lane j < D holds row j of AKHA and entry j of the running vector.
template <int D, int P>
__global__ void __launch_bounds__(64) gp_table_kernel(const double* __restrict__ cb64, const double* __restrict__ cbd64, double* __restrict__ hp, size_t L) {
    using Lc = XC<D>;
    using Ld = XD<D, P>;
    constexpr int HPN = GxLds<D>::HPN;
    const int lane = threadIdx.x;
    const size_t l = blockIdx.x;
    if (l >= L) return;
    const double* c = cb64 + l * Lc::SIZE;
    const double* cd = cbd64 + l * Ld::SIZE;
    double arow[D];
    const int jr = lane < D ? lane : 0;
#pragma unroll
    for (int i = 0; i < D; i++) arow[i] = c[Lc::AKHA + jr * D + i];
    for (int p = 0; p < P; p++) {
        double* out = hp + (l * (size_t)(kGxTables) + p) * HPN;
        for (int e = kChunkX * D + lane; e < HPN; e += 64) out[e] = 0.0;
        double u = cd[Ld::DK + p * D + jr];
#pragma unroll 1
        for (int k = kChunkX - 1; k >= 0; k--) {
            if (lane < D) out[k * D + lane] = u;
            double un = 0.0;
#pragma unroll
            for (int i = 0; i < D; i++) un = fma(arow[i], read_lane(u, i), un);
            u = un;
        }
    }
}

// z += sum_k tab[k][.] tile_lane[k]: the chunk response to a slab table [CK][D] (16-aligned, global or LDS)
template <int D, typename Ptr>
__device__ inline void chunk_response(Ptr tab, const double* tile_lane, int lane, double (&z)[D], bool& bad) {
    constexpr int CK = kChunkX, NSG = (CK * D + 15) / 16, NSG0 = NSG / 2, NSG1 = NSG - NSG0;
    double g0[NSG0], g1[NSG1];
    load_slabs<double, NSG0>(tab, lane, g0);
    load_slabs<double, NSG1>(tab + NSG0 * 16, lane, g1);
    static_for<CK / 2>([&](auto kvv) {
        constexpr int kv = decltype(kvv)::value;
        double yv[2];
        unpack<double>(*reinterpret_cast<const double2*>(tile_lane + kv * 2), yv);
        static_for<2>([&](auto qq) {
            constexpr int k = kv * 2 + decltype(qq)::value;
            bad = bad || (yv[decltype(qq)::value] != yv[decltype(qq)::value]);
            static_for<D>([&](auto ii) {
                constexpr int e = k * D + decltype(ii)::value, sl = e / 16;
                if constexpr (sl < NSG0) fmac_bc<e % 16>(z[decltype(ii)::value], g0[sl], yv[decltype(qq)::value]);
                else fmac_bc<e % 16>(z[decltype(ii)::value], g1[sl - NSG0], yv[decltype(qq)::value]);
            });
        });
        if constexpr (kv % 8 == 7) __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of the chunk from piling up in registers
    });
}

// A segment that holds missing ticks, walked tick by tick (ihgp.h:37-57 incl. the missing-data branch, and :215-219) with all
// P + 1 vectors of the state in registers.  Layout: a 16-lane row holds ONE vector, a quad of lanes per component (lane 4 j + e of
// the row = entry e of component j); x is replicated in all four rows, the P sensitivities sit in rows and "sets" (registers):
// dx_p lives in row (p + 1) % 4 of set (p + 1) / 4.  One tick, innovation form with w = 1 observed / 0 missing:
//     t = A x (the lane's block row against its quad: quad_perm broadcasts);  HA x = sum over the quads of their lane-0 entry of t
//     (H reads the first state of every component; two row_ror additions);  v = w (y - HA x);  x' = t + K v
//     u_p = A dx_p + dA_p x;  H u_p = HA dx_p + (H dA_p) x the same way;  dv_p = -w H u_p;  dx_p' = u_p + dK_p v + K dv_p
// -- no matrix operand leaves the registers.  About 100 instructions per tick for all ten vectors at d = 12 (the tick-by-tick
// kernel of grad_x.hip fetches ~1200 operands from LDS for the same tick).  Results go where the segment solve puts them: carried
// vectors in sm.carry, sum v dv_p in sm.gacc, sum v^2 into acc (lane 0), outputs into the tile; nmiss counts the missing ticks.
template <int DB, int J, int WRITE>
__device__ inline void walk_segment_x(const double* __restrict__ c, const double* __restrict__ cd, GxLds<DB * J>& sm, int n, int lane,
                                      double& acc, unsigned& nmiss) {
    constexpr int D = DB * J, P = 2 * J + 1, CK = kChunkX, STRIDE = GxLds<D>::STRIDE, NSET = (P + 1 + 3) / 4;
    using Lc = XC<D>;
    using Ld = XD<D, P>;
    static_assert(J <= 4 && DB <= 3, "one 16-lane row holds a vector");
    const int rowi = lane >> 4, jb = ((lane >> 2) & 3) < J ? ((lane >> 2) & 3) : 0, e = (lane & 3) < DB ? (lane & 3) : 0;
    const bool live = ((lane >> 2) & 3) < J && (lane & 3) < DB;
    const bool head = live && e == 0;
    double arow[DB], kx = live ? c[Lc::K + jb * DB + e] : 0.0;
#pragma unroll
    for (int q = 0; q < DB; q++) arow[q] = live ? c[Lc::AB + jb * DB * DB + e * DB + q] : 0.0;
    const double x0 = live ? sm.carry[jb * DB + e] : 0.0;
    wave_lds_fence();
    // One set of sensitivities per pass over the segment (x is walked again in every pass: cheap, and three sets at once -- their rows
    // of dA_p, dK_p, the vectors and the sums -- took the whole kernel from 253 VGPRs into AGPRs, the gap-free path included).
    // Outputs, sum v^2 and the carried x come from the first pass.
#pragma unroll 1
    for (int sidx = 0; sidx < NSET; sidx++) {
        const int q = 4 * sidx + rowi, p = q - 1;                    // vector q of the state: 0 = x (kept apart), q >= 1: dx_{q-1}
        const bool has = live && q >= 1 && q <= P;
        const int ps = has ? p : 0;
        double darow[DB];
#pragma unroll
        for (int qq = 0; qq < DB; qq++) darow[qq] = has ? cd[Ld::DA + ps * D * D + (jb * DB + e) * D + jb * DB + qq] : 0.0;
        const double dk = has ? cd[Ld::DK + ps * D + jb * DB + e] : 0.0;
        double dxv = has ? sm.carry[(ps + 1) * D + jb * DB + e] : 0.0, g = 0.0, xv = x0, sv2 = 0.0;
        unsigned miss_cnt = 0;
        const bool first = sidx == 0;
        double ynext = sm.tile[0];
#pragma unroll 1
        for (int t = 0; t < n; t++) {
            double* slot = sm.tile + (t / CK) * STRIDE + (t % CK);
            const double y = ynext;
            const int tn = t + 1 < n ? t + 1 : t;
            ynext = sm.tile[(tn / CK) * STRIDE + (tn % CK)];
            const bool w = !(y != y);
            double xb[DB];
            xb[0] = dpp0<0x00, 0xF>(xv);
            xb[1] = dpp0<0x55, 0xF>(xv);
            if (DB > 2) xb[DB - 1] = dpp0<0xAA, 0xF>(xv);
            double tx = 0.0;
#pragma unroll
            for (int qq = 0; qq < DB; qq++) tx = fma(arow[qq], xb[qq], tx);
            double hs = head ? tx : 0.0;
            hs += dpp0<0x124, 0xF>(hs);                              // row_ror:4, row_ror:8: the row's quads summed
            hs += dpp0<0x128, 0xF>(hs);
            const double hx = dpp0<0x00, 0xF>(hs);                   // (every lane of a quad reads its lane 0: HA x)
            const double v = w ? y - hx : 0.0;
            const double xn = fma(kx, v, tx);
            double u = 0.0;
            u = fma(arow[0], dpp0<0x00, 0xF>(dxv), u);
            u = fma(arow[1], dpp0<0x55, 0xF>(dxv), u);
            if (DB > 2) u = fma(arow[DB - 1], dpp0<0xAA, 0xF>(dxv), u);
#pragma unroll
            for (int qq = 0; qq < DB; qq++) u = fma(darow[qq], xb[qq], u);
            double ds = (head && has) ? u : 0.0;
            ds += dpp0<0x124, 0xF>(ds);
            ds += dpp0<0x128, 0xF>(ds);
            const double dv = w ? -dpp0<0x00, 0xF>(ds) : 0.0;       // -(HA dx_p + (H dA_p) x), ihgp.h:218
            dxv = has ? fma(kx, dv, fma(dk, v, u)) : 0.0;
            g = fma(v, dv, g);
            sv2 = fma(v, v, sv2);
            miss_cnt += w ? 0u : 1u;
            xv = live ? xn : 0.0;
            // (the observation must stay in the tile for the next pass: the outputs go out in the LAST one)
            if (WRITE && sidx == NSET - 1 && lane == 0) *slot = (WRITE == 2) ? hx : xn;   // HA x_t, or ihgp.h:51 `yhat = xnew(0, 0)`
        }
        // ---- back to where the segment solve keeps things (the carried x only once every pass has started from the old one) ----
        if (has) sm.carry[q * D + jb * DB + e] = dxv;
        if (has && jb == 0 && e == 0) sm.gacc[q - 1] += g;
        if (first && lane == 0) { acc += sv2; nmiss += miss_cnt; }
        if (sidx == NSET - 1 && live && rowi == 0) sm.carry[jb * DB + e] = xv;
        wave_lds_fence();
    }
}

// TS: the stream's type.  WRITE: 0 none, 1 filtered means (ihgp.h:51), 2 predicted means HA x_t.
template <typename TS, int DB, int J, int WRITE>
__global__ void __launch_bounds__(64 * kGxWaves)
grad_scan_x_kernel(const TS* __restrict__ Ty, size_t Tpar /* whole chunks */, size_t ld, size_t L, const double* __restrict__ cb64,
                   const double* __restrict__ cbd64, TS* __restrict__ x, TS* __restrict__ dx, TS* __restrict__ yhat, double* __restrict__ nll,
                   double* __restrict__ grad, int* __restrict__ flags, const double* __restrict__ hpg /* [L][kGxTables][HPN]: gp_table_kernel */) {
    constexpr int D = DB * J, P = 2 * J + 1, CK = kChunkX, SEG = 64 * CK, STRIDE = GxLds<D>::STRIDE;
    constexpr int EPV = 16 / (int)sizeof(TS);
    using VS = typename VecOf<TS>::type;
    using Lc = XC<D>;
    using Ld = XD<D, P>;
    constexpr int NSL = Lc::LS / 16, NAB = J * DB * DB, NSA = (NAB + 15) / 16;
    static_assert(P <= 9 && D <= 16, "table sizes");
    __shared__ __attribute__((aligned(16))) GxLds<D> lds_all[kGxWaves];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t l = (size_t)blockIdx.x * kGxWaves + wave;
    if (l >= L) return;                                              // no workgroup barrier below
    GxLds<D>& sm = lds_all[wave];
    const double* __restrict__ c = cb64 + l * Lc::SIZE;
    const double* __restrict__ cd = cbd64 + l * Ld::SIZE;
    double* tile_lane = sm.tile + lane * STRIDE;
    const TS* row = Ty + l * ld;
    TS* orow = WRITE ? yhat + l * ld : nullptr;
    const bool scan_ok = __builtin_amdgcn_readfirstlane((int)(c[Lc::SCANOK] != 0.0)) != 0;
    if (!scan_ok) {
        if (lane == 0) flags[l] = 1;
        return;
    }
    const int nlev = __builtin_amdgcn_readfirstlane((int)c[Lc::NLEV]);

    for (int e = lane; e < (P + 1) * D; e += 64) sm.carry[e] = e < D ? (double)x[l * D + e] : (double)dx[l * P * D + (e - D)];
    if (lane < P) sm.gacc[lane] = 0.0;
    wave_lds_fence();
    const double* __restrict__ hpl = hpg + l * (size_t)kGxTables * GxLds<D>::HPN;        // [p]: gp_p

    double acc = 0.0;                                                // per lane: sum of v^2
    unsigned nmiss = 0;                                              // lane 0: missing ticks met (segments walked tick by tick)
    const double ha = c[Lc::HA16 + (lane & 15)], kk = c[Lc::K16 + (lane & 15)];

    // A window of up to 64 chunks per turn; it starts wherever the last one ended (a multiple of 32 ticks), which is what lets a gap cut
    // it short: see step 1.
    size_t t0 = 0;
    while (t0 < Tpar) {
        int n = (int)(Tpar - t0 < (size_t)SEG ? Tpar - t0 : (size_t)SEG);          // a multiple of CK
        int nc = n / CK;                                             // chunks (= lanes) in use
        bool mine = lane < nc;
        const uptr<double> cu = launder(c);
        // ---- stage in: coalesced 16-byte loads, chunk-major into the padded tile, zeros past the end ----
        // (eight 16-byte loads per batch, and the lane's addresses recomputed here: the sixteen loads of an fp64 stream hoisted to the top, or
        //  their loop-invariant offsets and LDS addresses kept live across the segment loop, sat at the kernel's point of highest pressure,
        //  and its allocation paid for them in AGPR copies inside the replays)
#pragma unroll 1
        for (int rb = 0; rb < CK / EPV; rb += 8) {
            VS raw[8];
            int lo = lane;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int e = ((rb + r) * 64 + lo) * EPV;
                raw[r] = VS{};
                if (e < n) raw[r] = nt_load(reinterpret_cast<const VS*>(row + t0 + e));
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int e = ((rb + r) * 64 + lo) * EPV;
                TS vals[EPV];
                unpack<TS>(raw[r], vals);
                double* dst = sm.tile + (e / CK) * STRIDE + (e % CK);
#pragma unroll
                for (int q = 0; q < EPV; q += 2) *reinterpret_cast<double2*>(dst + q) = make_double2((double)vals[q], (double)vals[q + 1]);
            }
        }
        double ablk[NAB];
#pragma unroll
        for (int i = 0; i < NAB; i++) ablk[i] = cu[Lc::AB + i];
        wave_lds_fence();

        double xs[D], rr[D], ww[D];                                   // start state of the lane's chunk; r and w of step 3
        bool walked = false;
        // it = 0: the state itself (steps 1-3); it = p + 1: parameter p (step 4).  One loop so that the scan is instantiated once.
#pragma unroll 1
        for (int it = 0; it <= P; it++) {
            const int p = it > 0 ? it - 1 : 0;
            double z[D], s = 0.0;
#pragma unroll
            for (int i = 0; i < D; i++) z[i] = 0.0;
            if (it == 0) {
                bool bad = false;
                chunk_response<D>(cu + Lc::G, tile_lane, lane, z, bad);
                // Missing ticks in this window.  A few chunks with a gap: solve the gap-free chunks in front of the first of them as a
                // (shorter) window of their own, walk that one chunk tick by tick in the next turn, and start the turn after it right
                // behind it -- a gap then costs one more turn and 32 walked ticks instead of a walk of 2048.  Many: walk the window.
                const unsigned long long dirty = __builtin_amdgcn_ballot_w64(bad && mine);
                if (dirty != 0) {
                    const int first = (int)__builtin_ctzll(dirty);
                    if (__builtin_popcountll(dirty) <= kFewGaps && first > 0) {
                        nc = first; n = first * CK; mine = lane < nc;      // the prefix; lanes behind it are masked like lanes past the end
                    } else {
                        if (__builtin_popcountll(dirty) <= kFewGaps) n = CK;   // first == 0: just the chunk with the gap
                        walk_segment_x<DB, J, WRITE>(c, cd, sm, n, lane, acc, nmiss);
                        walked = true;
                        break;
                    }
                }
            } else {
                // ---- step 4: replay (x, dz_p) over the chunk from dz = 0 ----
                double dab[NSA], dkp, hdap;
                {
                    // the diagonal blocks of dA_p, packed like AB: entry e = (j, r, q) -> DA_p[(j DB + r) D + j DB + q]
#pragma unroll
                    for (int sidx = 0; sidx < NSA; sidx++) {
                        const int e = sidx * 16 + (lane & 15);
                        const int jb = e / (DB * DB), rq = e % (DB * DB);
                        const int src = e < NAB ? (jb * DB + rq / DB) * D + jb * DB + rq % DB : 0;
                        dab[sidx] = cd[Ld::DA + p * D * D + src];      // (entries past NAB are never broadcast; slabs must come
                    }                                                  //  straight from a load: no VALU write ahead of a DPP read)
                    const int i16 = (lane & 15) < D ? (lane & 15) : 0;
                    dkp = cd[Ld::DK + p * D + i16];
                    hdap = cd[Ld::HDA + p * D + i16];
                }
                // Which blocks of dA_p are there at all: none (p a magnitude or the noise variance: dA_p = 0, H dA_p = 0), one (p the
                // lengthscale of component j), or -- not produced by the update kernel, kept for safety -- several.  Read off the
                // data, so nothing here depends on the order of the parameters.
                unsigned nzmask = 0;
#pragma unroll
                for (int sidx = 0; sidx < NSA; sidx++) {
                    const int e = sidx * 16 + (lane & 15);
                    if (e < NAB && dab[sidx] != 0.0) nzmask |= 1u << (e / (DB * DB));
                }
                if ((lane & 15) < D && hdap != 0.0) nzmask |= 1u << J;
                unsigned blocks = 0;
#pragma unroll
                for (int j = 0; j <= J; j++) blocks |= (__builtin_amdgcn_ballot_w64((nzmask >> j) & 1u) != 0 ? 1u : 0u) << j;
                const int nblk = __builtin_popcount(blocks & ((1u << J) - 1u));
                // MODE: -1 no dA_p, j = only block j, J = every block
                const int mode = nblk == 0 ? ((blocks >> J) ? J : -1) : (nblk == 1 ? __builtin_ctz(blocks) : J);
                double xr[D];
#pragma unroll
                for (int i = 0; i < D; i++) xr[i] = xs[i];
                auto replay_p = [&](auto mode_c) {
                    constexpr int MODE = decltype(mode_c)::value - 1;   // (integral_constant of MODE + 1)
                    // (four ticks per turn, written out: within a turn the state is renamed, not copied; the loop-carried copy of
                    //  x and dz -- 24 of ~180 instructions of a tick -- is paid once per four ticks)
                    auto one_tick = [&](const int k) {
                        const double v = tile_lane[k];
                        double d0 = 0.0, d1 = 0.0, d2 = 0.0;           // dv = -(H dA_p) x - HA dz, three partial sums
                        static_for<D>([&](auto ii) {
                            constexpr int i = decltype(ii)::value;
                            if constexpr (MODE == J || (MODE >= 0 && i / DB == MODE)) fmac_bc<i>(i % 3 == 0 ? d0 : (i % 3 == 1 ? d1 : d2), hdap, xr[i]);
                        });
                        static_for<D>([&](auto ii) {
                            constexpr int i = decltype(ii)::value;
                            fmac_bc<i>(i % 3 == 0 ? d0 : (i % 3 == 1 ? d1 : d2), ha, z[i]);
                        });
                        const double dv = -((d0 + d1) + d2);
                        s = fma(v, dv, s);                            // (v is zero in lanes past the end)
                        double xn[D], zn[D];
#pragma unroll
                        for (int j = 0; j < J; j++)
#pragma unroll
                            for (int r = 0; r < DB; r++) {
                                double sx = ablk[j * DB * DB + r * DB] * xr[j * DB], sz = ablk[j * DB * DB + r * DB] * z[j * DB];
#pragma unroll
                                for (int q = 1; q < DB; q++) {
                                    sx = fma(ablk[j * DB * DB + r * DB + q], xr[j * DB + q], sx);
                                    sz = fma(ablk[j * DB * DB + r * DB + q], z[j * DB + q], sz);
                                }
                                xn[j * DB + r] = sx;
                                zn[j * DB + r] = sz;
                            }
                        static_for<NAB>([&](auto ee) {                // + dA_p x (block diagonal)
                            constexpr int e = decltype(ee)::value, jb = e / (DB * DB), r = (e % (DB * DB)) / DB, q = e % DB;
                            if constexpr (MODE == J || jb == MODE) fmac_bc<e % 16>(zn[jb * DB + r], dab[e / 16], xr[jb * DB + q]);
                        });
                        static_for<D>([&](auto ii) {
                            constexpr int i = decltype(ii)::value;
                            fmac_bc<i>(xn[i], kk, v);                 // x' = A x + K v
                            fmac_bc<i>(zn[i], dkp, v);                // + dK_p v
                            fmac_bc<i>(zn[i], kk, dv);                // + K dv
                        });
#pragma unroll
                        for (int i = 0; i < D; i++) { xr[i] = xn[i]; z[i] = zn[i]; }
                    };
#pragma unroll 1
                    for (int k = 0; k < CK; k += 4) { one_tick(k); one_tick(k + 1); one_tick(k + 2); one_tick(k + 3); }
                };
                if (mode < 0) {
                    // dA_p = 0, H dA_p = 0: dz' = AKHA dz + dK_p v.  The chunk-end dz is the response of the innovations (in the tile since step 2)
                    // to gp_p[k] = AKHA^(CK-1-k) dK_p, and sum_k v_k dv_k(local) = -w . dK_p with the adjoint sum w of step 3: no replay.
                    bool dummy = false;
                    chunk_response<D>(launder(hpl + (size_t)__builtin_amdgcn_readfirstlane(p) * GxLds<D>::HPN), tile_lane, lane, z, dummy);
                    double s0 = 0.0, s1 = 0.0;
                    static_for<D>([&](auto ii) { constexpr int i = decltype(ii)::value; fmac_bc<i>(i % 2 == 0 ? s0 : s1, dkp, ww[i]); });
                    s = -(s0 + s1);
                }
                else if (mode >= J) {
                    // dA_p with more than one block: not produced by the update kernel (a lengthscale moves one component).  An instantiation of
                    // the replay for it set this kernel's register allocation without ever running; such a latent goes to the tick-by-tick
                    // kernel whole instead (nothing of it has been written yet: the carried state lives in LDS until the end).
                    if (lane == 0) flags[l] = 1;
                    return;
                }
                else static_for<J>([&](auto jj) { if (mode == decltype(jj)::value) replay_p(std::integral_constant<int, decltype(jj)::value + 1>{}); });
            }
            // ---- carry-in on lane 0, then the 64-lane Kogge-Stone scan with the uniform powers of M (as the filter's) ----
            double t[D], cin[D];
#pragma unroll
            for (int i = 0; i < D; i++) cin[i] = sm.carry[it * D + i];
            {
                double m0[NSL];
                load_slabs<double, NSL>(cu + Lc::SP, lane, m0);
#pragma unroll
                for (int i = 0; i < D; i++) { t[i] = 0.0; z[i] = mine ? z[i] : 0.0; }
                matvec_bc<double, D, NSL>(m0, cin, t);
#pragma unroll
                for (int i = 0; i < D; i++) z[i] += (lane == 0) ? t[i] : 0.0;
#pragma unroll 1
                for (int lv = 0; lv < nlev; lv++) {
                    double m[NSL];
                    load_slabs<double, NSL>(cu + Lc::SP + lv * Lc::LS, lane, m);
                    const int sh = 1 << lv, addr = ((lane - sh) & 63) * 4;
#pragma unroll
                    for (int i = 0; i < D; i++) { const double mv = bperm<double>(addr, z[i]); t[i] = lane >= sh ? mv : 0.0; }
                    matvec_bc<double, D, NSL>(m, t, z);
                }
            }
            // start of every lane's chunk = end of the chunk before it; the carry-out is the end of the last chunk in use
            double st[D], cout[D];
            {
                const int addr = ((lane - 1) & 63) * 4;
#pragma unroll
                for (int i = 0; i < D; i++) {
                    const double mv = bperm<double>(addr, z[i]);
                    st[i] = lane >= 1 ? mv : cin[i];
                    cout[i] = read_lane(z[i], nc - 1);
                }
            }
            if (it == 0) {
#pragma unroll
                for (int i = 0; i < D; i++) xs[i] = st[i];
                // ---- step 2: replay x; v replaces y in the tile (zero in lanes past the end) ----
                double xr[D];
#pragma unroll
                for (int i = 0; i < D; i++) xr[i] = xs[i];
                double part = 0.0;
                auto x_tick = [&](const int k) {
                    const double y = tile_lane[k];
                    double h0 = 0.0, h1 = 0.0, h2 = 0.0;
                    static_for<D>([&](auto ii) {
                        constexpr int i = decltype(ii)::value;
                        fmac_bc<i>(i % 3 == 0 ? h0 : (i % 3 == 1 ? h1 : h2), ha, xr[i]);
                    });
                    const double v = mine ? y - ((h0 + h1) + h2) : 0.0;
                    part = fma(v, v, part);                           // ihgp.h:206-207, pre-step state
                    double xn[D];
#pragma unroll
                    for (int j = 0; j < J; j++)
#pragma unroll
                        for (int r = 0; r < DB; r++) {
                            double sx = ablk[j * DB * DB + r * DB] * xr[j * DB];
#pragma unroll
                            for (int q = 1; q < DB; q++) sx = fma(ablk[j * DB * DB + r * DB + q], xr[j * DB + q], sx);
                            xn[j * DB + r] = sx;
                        }
                    static_for<D>([&](auto ii) { fmac_bc<decltype(ii)::value>(xn[decltype(ii)::value], kk, v); });
#pragma unroll
                    for (int i = 0; i < D; i++) xr[i] = xn[i];
                    tile_lane[k] = v;
                };
#pragma unroll 1
                for (int k = 0; k < CK; k += 2) { x_tick(k); x_tick(k + 1); }
                acc += part;
                // ---- step 3: the adjoint of the chunk, walked BACKWARD over the innovations:
                //     lam_(k-1) = AKHA^T lam_k + HA^T v_k = A^T lam_k + HA^T (v_k - K . lam_k),   lam_(CK-1) = 0,
                // i.e. lam_i = sum_(k>i) v_k (HA AKHA^(k-1-i))^T: what a unit forcing at tick i does to sum_k v_k HA dz(k).  Two sums come out of it:
                //     r = lam_(-1) = sum_k v_k (HA AKHA^k)^T   what a START sensitivity of the chunk contributes (was a response to the hp table), and
                //     w = sum_i v_i lam_i                      so that a forcing dK_p v (every parameter with dA_p = 0) gives sum_k v_k HA dz(k) = w . dK_p.
                // A^T is block diagonal like A: 36 + 3 x 12 multiply-adds per tick at d = 12, once for all such parameters.
                {
                    double lam[D];
#pragma unroll
                    for (int i = 0; i < D; i++) { lam[i] = 0.0; ww[i] = 0.0; }
                    auto back_tick = [&](const int k) {
                        const double v = tile_lane[k];
                        double k0 = 0.0, k1 = 0.0, k2 = 0.0;
                        static_for<D>([&](auto ii) {
                            constexpr int i = decltype(ii)::value;
                            fmac_bc<i>(i % 3 == 0 ? k0 : (i % 3 == 1 ? k1 : k2), kk, lam[i]);
                        });
#pragma unroll
                        for (int i = 0; i < D; i++) ww[i] = fma(v, lam[i], ww[i]);
                        const double mu = v - ((k0 + k1) + k2);
                        double ln[D];
#pragma unroll
                        for (int j = 0; j < J; j++)
#pragma unroll
                            for (int q = 0; q < DB; q++) {                             // (A^T lam)_q of block j = sum_r A[r][q] lam_r
                                double sl = ablk[j * DB * DB + q] * lam[j * DB];
#pragma unroll
                                for (int r = 1; r < DB; r++) sl = fma(ablk[j * DB * DB + r * DB + q], lam[j * DB + r], sl);
                                ln[j * DB + q] = sl;
                            }
                        static_for<D>([&](auto ii) { fmac_bc<decltype(ii)::value>(ln[decltype(ii)::value], ha, mu); });
#pragma unroll
                        for (int i = 0; i < D; i++) lam[i] = ln[i];
                    };
#pragma unroll 1
                    for (int k = CK - 1; k >= 0; k -= 2) { back_tick(k); back_tick(k - 1); }
#pragma unroll
                    for (int i = 0; i < D; i++) rr[i] = lam[i];
                }
            } else {
                // sum_k v_k dv_k over the chunk = s - r . dx_start; all chunks of the segment into the stream's total
                double tot = s;
#pragma unroll
                for (int i = 0; i < D; i++) tot = fma(-rr[i], st[i], tot);
                tot = mine ? tot : 0.0;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
                if (lane == 0) sm.gacc[p] += tot;
            }
            if (lane < D) {
                double cv = 0.0;
#pragma unroll
                for (int i = 0; i < D; i++) cv = lane == i ? cout[i] : cv;
                sm.carry[it * D + lane] = cv;
            }
            wave_lds_fence();
        }
        // ---- step 5: the stream of means (a walked segment has written them already) ----
        if (WRITE) {
          if (!walked) {
            double xr[D];
#pragma unroll
            for (int i = 0; i < D; i++) xr[i] = xs[i];
#pragma unroll 1
            for (int k = 0; k < CK; k++) {
                const double v = tile_lane[k];
                double xn[D], h0 = 0.0, h1 = 0.0, h2 = 0.0;
                if (WRITE == 2) {
                    static_for<D>([&](auto ii) {
                        constexpr int i = decltype(ii)::value;
                        fmac_bc<i>(i % 3 == 0 ? h0 : (i % 3 == 1 ? h1 : h2), ha, xr[i]);
                    });
                }
#pragma unroll
                for (int j = 0; j < J; j++)
#pragma unroll
                    for (int r = 0; r < DB; r++) {
                        double sx = ablk[j * DB * DB + r * DB] * xr[j * DB];
#pragma unroll
                        for (int q = 1; q < DB; q++) sx = fma(ablk[j * DB * DB + r * DB + q], xr[j * DB + q], sx);
                        xn[j * DB + r] = sx;
                    }
                static_for<D>([&](auto ii) { fmac_bc<decltype(ii)::value>(xn[decltype(ii)::value], kk, v); });
#pragma unroll
                for (int i = 0; i < D; i++) xr[i] = xn[i];
                tile_lane[k] = WRITE == 2 ? (h0 + h1) + h2 : xn[0];   // HA x_t, or ihgp.h:51 `yhat = xnew(0, 0)`
            }
          }
            wave_lds_fence();
#pragma unroll 1
            for (int rb = 0; rb < CK / EPV; rb += 8) {
                int lo = lane;
                asm volatile("" : "+v"(lo));                         // (addresses recomputed here, as in the stage-in)
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int e = ((rb + r) * 64 + lo) * EPV;
                    if (e < n) {
                        const double* src = sm.tile + (e / CK) * STRIDE + (e % CK);
                        TS vals[EPV];
#pragma unroll
                        for (int q = 0; q < EPV; q++) vals[q] = (TS)src[q];
                        nt_store(pack<TS>(vals), reinterpret_cast<VS*>(orow + t0 + e));
                    }
                }
            }
        }
        wave_lds_fence();
        t0 += (size_t)n;
    }

    // ---- results: carried (x, dx), NLL and gradient of the ticks swept (grad_x_kernel adds the stream's last T mod 32 ticks) ----
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        flags[l] = 0;
    }
    for (int e = lane; e < (P + 1) * D; e += 64) {
        if (e < D) x[l * D + e] = (TS)sm.carry[e];
        else dx[l * P * D + (e - D)] = (TS)sm.carry[e];
    }
    const double S = c[Lc::S], nobs = (double)Tpar - (double)__shfl(nmiss, 0, 64);
    if (lane == 0 && nll) nll[l] = 0.5 * (acc / S + nobs * c[Lc::LOGS]);
    if (lane < P) grad[l * P + lane] = sm.gacc[lane] / S - 0.5 * (acc / S - nobs) * cd[Ld::DS + lane] / S;   // ihgp.h:219 summed over the ticks
}

template <typename TS, int DB, int J>
int launch_gsx(const void* Ty, size_t Tpar, size_t ld, size_t L, const double* cb64, const double* cbd64, void* x, void* dx, void* yhat,
               double* nll, double* grad, int* flags, double* hp, hipStream_t stream, int out_mode, int hp_build) {
    dim3 block(64 * kGxWaves), grid((unsigned)((L + kGxWaves - 1) / kGxWaves));
    if (hp_build) hipLaunchKernelGGL((gp_table_kernel<DB * J, 2 * J + 1>), dim3((unsigned)L), dim3(64), 0, stream, cb64, cbd64, hp, L);
#define MOIHGP_GSX_LAUNCH(W_) hipLaunchKernelGGL((grad_scan_x_kernel<TS, DB, J, W_>), grid, block, 0, stream, (const TS*)Ty, Tpar, ld, L, cb64, cbd64, \
                                                 (TS*)x, (TS*)dx, (TS*)yhat, nll, grad, flags, (const double*)hp)
    if (yhat && out_mode == 2) MOIHGP_GSX_LAUNCH(2);
    else if (yhat) MOIHGP_GSX_LAUNCH(1);
    else MOIHGP_GSX_LAUNCH(0);
#undef MOIHGP_GSX_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("grad_scan_x_kernel launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

// The per-latent tables of the sweep (gp_p[k] = AKHA^(CK-1-k) dK_p and the HA AKHA^k rows) alone: they change with the hyper-parameters only, so
// the handle builds them on its own stream whenever it rewrites the sensitivity blocks (capi.cpp run_ihgp_update / grad_stream_impl) and the
// sweeps, whatever stream they run on, only ever read them.
int launch_gp_table_x(int kernel, const double* cb64, const double* cbd64, double* hp, size_t L, hipStream_t stream) {
    if (L == 0) return 0;
    const int base = kernel_base(kernel), J = kernel_stack(kernel);
#define MOIHGP_GPT_CASE(DBB, JJ)                                                                                                        \
    if (base == (DBB == 2 ? 0 : 1) && J == JJ) {                                                                                        \
        hipLaunchKernelGGL((gp_table_kernel<DBB * JJ, 2 * JJ + 1>), dim3((unsigned)L), dim3(64), 0, stream, cb64, cbd64, hp, L);        \
        hipError_t e = hipGetLastError();                                                                                               \
        if (e != hipSuccess) { set_last_error("gp_table_kernel launch: %s", hipGetErrorString(e)); return 2; }                          \
        return 0;                                                                                                                       \
    }
    MOIHGP_GPT_CASE(2, 2); MOIHGP_GPT_CASE(2, 3); MOIHGP_GPT_CASE(2, 4);
    MOIHGP_GPT_CASE(3, 2); MOIHGP_GPT_CASE(3, 3); MOIHGP_GPT_CASE(3, 4);
#undef MOIHGP_GPT_CASE
    set_last_error("stacked kernel id %d is not built", kernel);
    return 1;
}

// The whole chunks [0, Tpar) of every latent's stream; flags[l] = 1 where the latent was left untouched (missing ticks, unusable scan
// tables), 0 where (x, dx, nll, grad) now hold the state after / the sums over those Tpar ticks.
int launch_grad_scan_x(int kernel, int dtype, const void* Ty, size_t Tpar, size_t ld, size_t L, const double* cb64, const double* cbd64,
                       void* x, void* dx, void* yhat, double* nll, double* grad, int* flags, double* hp, hipStream_t stream, int out_mode, int hp_build) {
    if (L == 0) return 0;
    const int base = kernel_base(kernel), J = kernel_stack(kernel);
#define MOIHGP_GSX_CASE(DBB, JJ)                                                                                                        \
    if (base == (DBB == 2 ? 0 : 1) && J == JJ)                                                                                          \
        return dtype == 0 ? launch_gsx<double, DBB, JJ>(Ty, Tpar, ld, L, cb64, cbd64, x, dx, yhat, nll, grad, flags, hp, stream, out_mode, hp_build)  \
                          : launch_gsx<float, DBB, JJ>(Ty, Tpar, ld, L, cb64, cbd64, x, dx, yhat, nll, grad, flags, hp, stream, out_mode, hp_build)
#ifndef MOIHGP_GSX_ONLY_D12       // (development: the d = 12 kernels alone, for quick resource checks)
    MOIHGP_GSX_CASE(2, 2); MOIHGP_GSX_CASE(2, 3); MOIHGP_GSX_CASE(2, 4);
    MOIHGP_GSX_CASE(3, 2); MOIHGP_GSX_CASE(3, 3);
#endif
    MOIHGP_GSX_CASE(3, 4);
#undef MOIHGP_GSX_CASE
    set_last_error("stacked kernel id %d is not built", kernel);
    return 1;
}

}  // namespace moihgp
