// common.h -- shared definitions for libmoihgp.so (gfx950 HIP).
//
// Per-latent "constant block": the stationary matrices of one IHGP (reference
// include/moihgp/ihgp.h:243-254), computed once per hyper-parameter update by
// stationary.hip and read by every recursion kernel.  One block per latent, array-of-
// structs, so that a wavefront that owns one latent fetches its block with scalar loads.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace moihgp {

constexpr int kNumIgpParam = 3;   // magnitude, lengthscale, noise (matern32ss.h:34-36)

// Offsets (in scalars) inside a constant block for state dim D and P hyper-parameters.
template <int D, int P = kNumIgpParam>
struct CB {
    static constexpr int AKHA  = 0;                 // [D*D]  A - K H A            ihgp.h:130
    static constexpr int K     = AKHA + D * D;      // [D]    PP H^T / S           ihgp.h:127
    static constexpr int A     = K + D;             // [D*D]  expm(dt F)           ihgp.h:120
    static constexpr int HA    = A + D * D;         // [D]    H A                  ihgp.h:129
    static constexpr int S     = HA + D;            // [1]    H PP H^T + R         ihgp.h:126
    static constexpr int LOGS  = S + 1;             // [1]    log(S)
    static constexpr int DAKHA = LOGS + 1;          // [P][D*D]                    ihgp.h:192,197
    static constexpr int DK    = DAKHA + P * D * D; // [P][D]                      ihgp.h:189
    static constexpr int DA    = DK + P * D;        // [P][D*D]                    ihgp.h:143,167
    static constexpr int HDA   = DA + P * D * D;    // [P][D]   (H dA)^T           ihgp.h:193,198
    static constexpr int DS    = HDA + P * D;       // [P]                         ihgp.h:188
    static constexpr int PARAMS = DS + P;           // [P]    hyper-parameters of this latent
    static constexpr int ITERS = PARAMS + P;        // [1+P]  DARE / DLyap iteration counts (as scalars)
    // ---- tables for the time-parallel segment solve of recursion.hip (derived from AKHA, K) ----
    // CK = ticks per lane per segment: kChunk32 in the fp32 copy of the block, kChunk64 in the fp64 one.
    static constexpr int G     = ITERS + 1 + P;     // [16][D]   g_k = AKHA^(CK-1-k) K, chunk response z = sum_k g_k y_k
    static constexpr int SP    = G + 16 * D;        // [4][D*D]  M^(1,2,4,8), M = AKHA^CK (in-row scan levels)
    static constexpr int PJ    = SP + 4 * D * D;    // [16][D*D] M^(r+1), r = lane % 16 (cross-row fix-up)
    static constexpr int SCANOK = PJ + 16 * D * D;  // [1]  1 if every table entry is finite and tame in this block's precision, else 0:
                                                    //      an unstable latent (rho(AKHA) > 1, possible with the reference's literal DARE) is
                                                    //      then filtered by the exact sequential path instead of the scan
    static constexpr int RAW   = SCANOK + 1;
    static constexpr int SIZE  = (RAW + 3) / 4 * 4; // padded to 16/32 bytes
};

constexpr int kChunk32 = 16;   // fp32: 16 ticks = 64 B per lane per segment (1024-tick segments)
constexpr int kChunk64 = 8;    // fp64:  8 ticks = 64 B per lane per segment (512-tick segments)

constexpr int cb_size(int d) { return d == 2 ? CB<2>::SIZE : CB<3>::SIZE; }

// ---- stacked models (sum of J Matern components, state dim D = J * d_base up to 12; include/moihgp.h MOIHGP_STACK) ----------
// Filter-mode block (no hyper-parameter sensitivities yet): the matrices of ihgp.h:120-130 plus the tables of the
// time-parallel segment solve of recursion_x.hip.  Both the fp64 and the fp32 copy use kChunkX ticks per lane.
constexpr int kChunkX = 32;          // ticks per lane per segment (2048-tick segments)
constexpr int kMaxStackDim = 12;
constexpr int kPkMax = 2 * 2 * (3 * 3 + 2 * 3);     // (two block pairs of DB = 3: the largest PK table of any stacked model)
template <int D>
struct XC {
    static constexpr int AKHA   = 0;                // [D*D]
    static constexpr int K      = AKHA + D * D;     // [D]
    static constexpr int A      = K + D;            // [D*D]
    static constexpr int HA     = A + D * D;        // [D]
    static constexpr int S      = HA + D;           // [1]
    static constexpr int LOGS   = S + 1;            // [1]
    static constexpr int ITERS  = LOGS + 1;         // [1]  DARE iteration count
    static constexpr int SCANOK = ITERS + 1;        // [1]  as CB::SCANOK
    static constexpr int NLEV   = SCANOK + 1;       // [1]  scan levels that matter in this block's precision: M^(2^k) for k >= NLEV is
                                                    //      below D * |entry| < 1e-20 (fp64 block) / 1e-10 (fp32 block) and is skipped
    static constexpr int AB     = (NLEV + 1 + 3) / 4 * 4;     // [J][DB*DB] the diagonal blocks of A, packed (<= 3*D scalars)
    // ---- "slab" tables of recursion_x.hip: 16-element groups, each fetched as one register whose 16-lane rows all hold the
    // same 16 scalars, and consumed by v_fmac_*_dpp row_newbcast (a broadcast operand at no instruction cost).  16-aligned.
    static constexpr int LS     = (D * D + 15) / 16 * 16;      // one matrix, padded
    static constexpr int GN     = (kChunkX * D + 15) / 16 * 16;
    static constexpr int HA16   = (AB + 3 * D + 15) / 16 * 16; // [16]  HA, zero padded
    static constexpr int K16    = HA16 + 16;                   // [16]  K, zero padded
    static constexpr int G      = K16 + 16;         // [GN]  g_k = AKHA^(CK-1-k) K, row-major [k][i]
    static constexpr int SP     = G + GN;           // [6][LS]  M^(1,2,4,8,16,32), M = AKHA^CK: levels of a 64-lane Kogge-Stone scan
    // coefficient pairs of the packed fp32 replay (recursion_x.hip): per pair p of diagonal blocks (2p, 2p + 1; an odd count pairs its last
    // block with zeros), DB*DB + 2 DB pairs [A_rq | HA_q | K_r], each pair = (value of block 2p, value of block 2p + 1)
    static constexpr int PK     = SP + 6 * LS;
    static constexpr int SIZE   = (PK + kPkMax + 15) / 16 * 16;   // a multiple of 16: every latent's tables stay 16-aligned
};
// Sensitivity block of a stacked latent (fp64 only; gradient sweeps read it): ihgp.h:136-200 for P = 2J + 1 hyper-parameters.
template <int D, int P>
struct XD {
    static constexpr int DAKHA = 0;                    // [P][D*D]
    static constexpr int DK    = DAKHA + P * D * D;    // [P][D]
    static constexpr int DA    = DK + P * D;           // [P][D*D]
    static constexpr int HDA   = DA + P * D * D;       // [P][D]   (H dA)^T
    static constexpr int DS    = HDA + P * D;          // [P]
    static constexpr int ITERS = DS + P;               // [P]      DLyap iteration counts
    static constexpr int SIZE  = (ITERS + P + 3) / 4 * 4;
};
constexpr int xd_size(int d, int P) { return (P * (2 * d * d + 2 * d) + 2 * P + 3) / 4 * 4; }
constexpr int xc_size(int d) {
    return d == 2 ? XC<2>::SIZE : d == 3 ? XC<3>::SIZE : d == 4 ? XC<4>::SIZE : d == 6 ? XC<6>::SIZE : d == 8 ? XC<8>::SIZE : d == 9 ? XC<9>::SIZE : d == 12 ? XC<12>::SIZE : 0;
}
// kernel id = base | (J << 4) (include/moihgp.h); J == 0 for the reference's two models
inline int kernel_base(int kernel) { return kernel & 15; }
inline int kernel_stack(int kernel) { return kernel >> 4; }

// ---- error handling -------------------------------------------------------------------------
// A failing HIP call anywhere below the C ABI raises HipFailure; it never crosses the ABI: every extern "C" entry catches it
// (capi.cpp).  The additive entries (part 2 of include/moihgp.h, which have return codes) turn it into rc = 2 + moihgp_last_error();
// the reference entries (gpXX_*: all void / value returns, wrapper.cpp:31-326, no status channel) print it and abort.
struct HipFailure {
    hipError_t err;
    const char* what;
    const char* file;
    int line;
};
void set_last_error(const char* fmt, ...);
[[noreturn]] void fatal_hip(hipError_t e, const char* what, const char* file, int line);

#define MOIHGP_HIP_FATAL(expr)                                                      \
    do {                                                                            \
        hipError_t e__ = (expr);                                                    \
        if (e__ != hipSuccess) ::moihgp::fatal_hip(e__, #expr, __FILE__, __LINE__); \
    } while (0)

// ---- kernel launchers (implemented in the .hip files) ---------------------------------------
// stationary.hip: IHGP::update for n latents.  params_dev [n][3] fp64 (device).  Writes the fp64
// constant blocks and their fp32 copies.
void launch_ihgp_update(int kernel, int d, double dt, const double* params_dev, size_t n,
                        double* cb64, float* cb32, int* n_unstable /* device int[2]: fp64 count, fp32 count */, hipStream_t stream);

// stationary_x.hip / recursion_x.hip: the same two stages for stacked models (state dim d in {4, 6, 8, 9, 12}).
// params_dev [n][2J+1] = (magnitude_j, lengthscale_j) x J, noise.
void launch_stack_update(int kernel, double dt, const double* params_dev, size_t n, double* cb64, float* cb32,
                         double* cbd64 /* [n] sensitivity blocks XD, or NULL to skip them */, int* n_unstable /* int[4]: unusable tables fp64 / fp32, fp32 only (listed in rescue_idx), slow filters */,
                         int* rescue_idx /* int[n] or NULL: see stack_update_kernel */, hipStream_t stream);
// gaps_x.hip: the latents listed in idx (n of them: fp32 tables unusable, fp64 ones fine) of an fp32 bank, swept in fp64: rows and start states
// (Ty != NULL: per sweep) or constant blocks (Ty == NULL: per update) out into a compact fp64 bank, results back
void launch_rescue_gather(const float* Ty, size_t T, size_t ld, const int* idx, size_t n, const double* cb64, int xcs, const float* xin, int d,
                          double* rows, double* cbc, double* xc, hipStream_t s);
void launch_rescue_scatter(const int* idx, size_t n, const double* rows, size_t T, size_t ld, const double* xc, int d, const double* nllc,
                           float* yhat, size_t ldo, float* x, double* nll, hipStream_t s);
// stationary_x.hip: the reference's own models (d = 2, 3) in the stacked layout, for the few-latents team kernel: [n][xc_size(d)] from the CB blocks
void launch_xc_from_cb(int d, const double* cb64, size_t n, double* xc64, float* xc32, hipStream_t stream);
// stack_dispatch.hip: the team kernel for those models, if the stream and the bank suit it (returns -1 if not: the caller carries on)
int launch_filter_teamc_plain(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* xc64, const float* xc32, const double* tp64, const float* tp32,
                              const void* xin, void* x, void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, size_t ldo, double* total, int team_mode);
// recursion_x.hip: scan powers of the chunk-templated team kernel (few latents), [L][5][team_powers_elems(d)] in both precisions; after every launch_stack_update
constexpr size_t team_powers_elems(int d) { return 5 * (size_t)(7 * ((d * d + 15) / 16 * 16) + 16); }
void launch_team_powers(int kernel, const double* cb64, size_t L, double* tp64, float* tp32, hipStream_t stream);
int launch_filter_stream_x(int kernel, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const float* cb32,
                           const void* xin, void* x, void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                           double* scratch = nullptr /* [scratch_len] per-slice NLL partials of the time split */, size_t scratch_len = 0,
                           int force_slices = 0 /* tuning / test hook: 1 = no split, n > 1 = n slices */, size_t ld_out = 0 /* row stride of yhat; 0 = ld */,
                           int* link_flags = nullptr /* [L] */, double* link_state = nullptr /* [L][144] */, double* total = nullptr /* sum of nll[] */,
                           int max_links = -1 /* -1: automatic */, int team_mode = -1 /* few latents: -1 automatic, 0 never, 1 always use a team kernel, 2 the 32-tick one */,
                           const double* tp64 = nullptr, const float* tp32 = nullptr /* launch_team_powers' tables, or NULL: no chunk-templated team kernel */
                           /* scratch; with both (L >= 1024): segments with few gaps are handled as broken links by a second pass instead of
                              being walked tick by tick */);

// grad_gen.hip: gradient sweep of the latents flagged 1 in fallback[] (missing ticks), scan over the chunks' affine maps; clears the flag
int launch_grad_gen(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const float* cb32, void* x, void* dx,
                    void* yhat, double* nll, double* grad, int* fallback, hipStream_t stream, int out_mode);
// grad_x.hip: sensitivity / gradient sweep of the stacked models (needs the XD blocks).
int launch_grad_stream_x(int kernel, int dtype, const void* Ty, size_t T, size_t ld, size_t L, const double* cb64, const double* cbd64,
                         void* x, void* dx, void* yhat, double* nll, double* grad, hipStream_t stream,
                         int out_mode = 1 /* 1: yhat holds filtered means, 2: predicted means HA x_t (pre-step state) */,
                         int* flags = nullptr /* device [L] */, double* hp = nullptr /* device [L][gradx_hp_len(d)] */
                         /* with both scratch areas, streams of >= 512 ticks take the time-parallel sweep */,
                         int hp_build = 1 /* 0: hp already holds the table of the current hyper-parameters */);
constexpr size_t gradx_hp_len(int d) { return 9 * ((size_t)(kChunkX * d + 15) / 16 * 16); }   // (grad_scan_x.hip kGxTables slab tables per latent)
// grad_scan_x.hip: the same sweep parallel in time over the stream's whole 32-tick chunks [0, Tpar); flags[l] = 1 marks latents left
// untouched (missing ticks, unusable scan tables), for the others (x, dx, nll, grad) hold the state after / sums over those ticks.
int launch_gp_table_x(int kernel, const double* cb64, const double* cbd64, double* hp, size_t L, hipStream_t stream);
int launch_grad_scan_x(int kernel, int dtype, const void* Ty, size_t Tpar, size_t ld, size_t L, const double* cb64, const double* cbd64,
                       void* x, void* dx, void* yhat, double* nll, double* grad, int* flags, double* hp, hipStream_t stream, int out_mode, int hp_build = 1);

// Missing ticks of the stacked models' many-latent sweep by exact imputation (recursion_x.hip: filter_x_gaps_a / _b_kernel -- two gap-free sweeps of
// a latent that holds gaps around a scalar recursion over its gaps).  gaps_x.hip: the scratch it works in and the filters' impulse
// responses it needs.
constexpr int kGapSMax = 1024;        // impulse-response table per latent (ticks); beyond it the response must be negligible
struct GapArgs {                      // what launch_filter_stream_x(force_slices = -7) finds behind its `scratch` argument (host memory)
    const void* imp;                  // [L][kGapSMax] impulse responses, the stream's scalar type
    int* gpos; void* gval; void* gw;  // [L][gcap] the gaps' ticks / their predictions / their fill values
    size_t gcap;
    int* gstat;                       // [L] per latent swept by imputation: 2 * gaps + 1 if solved (bit 30: by the state form), else 2 * the reason why not
};
struct GapBank {
    void* imp_in = nullptr; void* imp_out = nullptr;            // [L][kGapSMax]: unit impulses (constant: gap_bank_init) and the filters' responses to them
    void* xz = nullptr;                                         // zeros [L][d]: the start state of the impulse sweep
    void* x1 = nullptr;                                         // its end states [L][d] (unused)
    int* gpos = nullptr; void* gval = nullptr; void* gw = nullptr;   // [L][gcap]
    size_t gcap = 0;
    int* gstat = nullptr;                                       // [L]
};
size_t gap_bank_bytes(int d, int dtype, size_t L, size_t T);
GapBank gap_bank_carve(void* base, int d, int dtype, size_t L, size_t T);
int gap_bank_init(const GapBank& b, int d, int dtype, size_t L, hipStream_t s);
int launch_gap_impulse(const GapBank& b, int kernel, int dtype, size_t L, const double* cb64, const float* cb32, hipStream_t s);
// recursion.hip: the same sweep over segment-major streams [ceil(T / SEG)][L][SEG], SEG = 4096 / sizeof(scalar) ticks (d = 2, 3)
int launch_filter_stream_tiled(int d, int dtype, const void* Ty, size_t T, size_t L, const double* cb64, const float* cb32, const void* xin, void* x,
                               void* yhat, double* nll, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, int n_unstable, double* total, int variant = 0);
// series-major [L][ld] <-> segment-major [ceil(T / SEG)][L][SEG] (to_tiled != 0: src is series-major; ticks past T are written as zeros)
int launch_stream_retile(int dtype, const void* src, void* dst, size_t L, size_t T, size_t ld, int to_tiled, hipStream_t stream);
// recursion.hip: batched sweeps over series-major streams.
int launch_filter_stream(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L,
                         const double* cb64, const float* cb32, const void* xin /* start state */, void* x /* end state, may alias xin */,
                         void* yhat, double* nll,
                         hipStream_t stream, int variant = 0, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                         int nsplit = 1, size_t Tslice = 0, int n_unstable = 0 /* latents with SCANOK == 0 in this dtype's blocks */,
                         double* total = nullptr /* device scalar: sum of nll[] (optional) */, int nbig = 0 /* see filter_split_plan; 0 = all slices alike */,
                         size_t ld_out = 0 /* row stride of yhat; 0 = ld */);
void launch_nll_total(const double* nll, size_t L, double* total, hipStream_t stream);
// Time split for small L (slices of one latent = wavefronts of one workgroup): nsplit == 1 means none.
void filter_split_plan(int dtype, size_t T, size_t L, int* nsplit, size_t* Tslice, int* nbig /* leading slices of Tslice ticks; the rest hold one segment less */);
int launch_grad_stream(int d, int dtype, const void* Ty, size_t T, size_t ld, size_t L,
                       const double* cb64, const float* cb32, void* x, void* dx, void* yhat,
                       double* nll, double* grad, int* fallback /* int[2 L + 1] scratch */, hipStream_t stream,
                       int out_mode = 1 /* 1: yhat holds filtered means, 2: predicted means HA x_t */);

// tick.hip: one-tick kernels behind the reference ABI (all fp64, device pointers).
struct TickArgs {
    int d; size_t M, L;
    const double* cb64;      // [L] constant blocks
    const double* U;         // [M][L]
    const double* S;         // [L]
    const double* sqrtS;     // [L] S^1/2
    const double* invsqrtS;  // [L] S^-1/2
    const double* sigma;     // [1]
    // moihgp.h:565-607: the gradient overload of negLogLikelihood adds the per-latent losses in its threaded branch (:590) and drops
    // them in the serial one (:597-607); 1 = add them (threading on, or MOIHGP_LIK1_FULL_LOSS=1), 0 = the serial branch's value.
    int lik1_latent_loss;
    // stacked kernels (MOIHGP_STACK: state dim 4..12, P = 2J + 1 hyper-parameters): the per-latent matrices live in an XC block (cb64)
    // and an XD sensitivity block (cbd64) instead of one CB block; P = 3 and cbd64 = NULL for the reference's two models
    int P = kNumIgpParam;
    const double* cbd64 = nullptr;
};
void launch_project_tick(const TickArgs& a, const double* y, double* Ty, double* Uty, double* part /* [32][L] scratch or NULL */, hipStream_t s);
void launch_project_tick_missing(const TickArgs& a, const double* y, double* Ty, double* work /*L*L+L*/, hipStream_t s);
// whole streams: re-project the ticks whose observation vector holds NaN by least squares over the observed rows (moihgp.h:167-178),
// behind launch_project_stream; needs U^T U = I (a polar factor).  Y [T][M], Ty [L][ld] of `dtype`.
constexpr int kLsMaxMissing = 64;          // missing outputs per tick the k x k system is built for (one wavefront eliminates it, lane = row)
// the least-squares kernel keeps r = U^T y0 [L] and the augmented k x (k + 1) system in the workgroup's LDS: L <= 15040
constexpr size_t ls_project_lds_bytes(size_t L) { return (L + (size_t)kLsMaxMissing * (kLsMaxMissing + 1)) * sizeof(double); }
constexpr bool ls_project_fits(size_t L) { return ls_project_lds_bytes(L) <= 150 * 1024; }
int launch_project_stream_missing(int dtype, const void* Y, size_t T, size_t M, size_t L, const double* U, const double* invsqrtS, void* Ty, size_t ld,
                                   hipStream_t s);
// the same with the latents split over ranks: phase 0 = this rank's part of [U_miss r | U_miss U_miss^T] per affected tick (packed [n][kmax + kmax^2]),
// phase 1 = solve the (all-reduced) k x k systems and correct this rank's rows of Ty.  ticks: device int32 [n], the affected ticks.
int launch_ls_shard(int phase, int dtype, const void* Y, size_t M, size_t Lr, const int* ticks, size_t n, int kmax, const double* U, const double* sqrtS,
                    const double* invsqrtS, double* packed, void* Ty, size_t ld, hipStream_t s);
void launch_ortho_defect(const double* G /* L x L */, size_t L, double* out /* device scalar: max |G - I| */, hipStream_t s);
void launch_step_tick(const TickArgs& a, const double* x, const double* Ty /*NULL: predict only*/, const double* dx,
                      double* xnew, double* Tyhat, double* dxnew, hipStream_t s);
void launch_unproject_tick(const TickArgs& a, const double* Tyhat, double* yhat, hipStream_t s);
// small models: MOIHGP::step as one workgroup, host-mapped inputs / outputs, completion signalled through *flag = seq
bool fused_step_fits(size_t M, size_t L);
bool fused_lik_fits(size_t M, size_t L);
void launch_fused_lik(const TickArgs& a, const double* x, const double* y, const double* dx, double* loss, double* grad,
                      unsigned long long* flag, unsigned long long seq, hipStream_t s);
void launch_fused_step(const TickArgs& a, const double* x, const double* y, const double* dx, double* xnew, double* yhat, double* dxnew,
                       unsigned long long* flag, unsigned long long seq, hipStream_t s);
// NLL of one tick: loss (device scalar) and, if grad != NULL, the full gradient vector
// [M*L + L + 1 + L*P] (moihgp.h:460-611).  scratch: >= 4*L + 8 doubles.
void launch_nll_tick(const TickArgs& a, const double* x, const double* y, const double* Ty, const double* Uty,
                     const double* dx, double* loss, double* grad, double* scratch, hipStream_t s);

// gemm_mfma.hip: MFMA GEMMs (whole-stream projection, Gram / update products of the polar factor).
// invsqrtS / sqrtS: the device vectors S^-1/2 and S^1/2 (launch_scales)
void launch_scales(const double* S, size_t L, double* sqrtS, double* invsqrtS, hipStream_t s);
void launch_narrow(const double* src, size_t n, float* dst, hipStream_t s);       // dst = (float) src
int launch_project_stream(int dtype, const void* Y, size_t T, size_t M, size_t L, const double* U, const float* U32 /* optional fp32 image of U */, const double* invsqrtS,
                          void* Ty, size_t ld, hipStream_t s);
int launch_unproject_stream(int dtype, const void* Tyhat, size_t T, size_t ld, size_t M, size_t L, const double* U, const float* U32,
                            const double* sqrtS, void* Yhat, hipStream_t s);
// gradU[r][c] = sum_t Y[t][r] Z[c][t]   (Y tick-major [W][M], Z series-major [L][ldz])
int launch_ugrad_gemm(const double* Y, size_t W, size_t M, const double* Z, size_t ldz, size_t L, double* gradU, hipStream_t s);
int launch_gram(const double* X, size_t M, size_t L, double* G, hipStream_t s);                       // G = X^T X
int launch_matmul_nn(const double* X, size_t M, size_t L, const double* W, double* C, hipStream_t s);   // C = X W

// polar.hip: polar factor of an M x L matrix on the device (Newton-Schulz), moihgp.h:433-447.
// A_dev is overwritten with the factor; work needs polar_work_doubles(M, L) doubles.  Returns the iteration count, or
// -1 if it did not converge (rank-deficient input).
int polar_factor_device(double* A_dev, size_t M, size_t L, double* work, hipStream_t s, int* deflate_warm = nullptr /* in / out: see polar_deflate */);
size_t polar_work_doubles(size_t M, size_t L);
// polar_deflate.hip: exact deflation of up to 32 outlying singular values ahead of the iteration (X: M x L, G = X^T X, frob2 = ||G - I||_F^2);
// *n_pairs > 0: X (polar factor unchanged) and G (= the new X^T X) were updated in place.  Returns 0 or an error code.
size_t polar_deflate_work_doubles(size_t M, size_t L);
int polar_deflate(double* X, size_t M, size_t L, double* G, double frob2, double* work, hipStream_t s, int* n_pairs, int trace, int* warm = nullptr);
// small matrices: the same iteration as one workgroup in LDS (one launch); status_dev: steps taken or -1
bool polar_small_fits(size_t M, size_t L);
void launch_polar_small(double* A_dev, size_t M, size_t L, int* status_dev, hipStream_t s);

// window.hip: the learners' windowed objective (moihgp_online.h:61-70) as one device pipeline.
struct WindowBufs {
    size_t W, ldw;            // ticks in the window, row stride of the [L][ldw] streams
    double* Y;                // [W][M]  observations (tick-major)
    double* Ty;               // [L][ldw] projected stream
    double* hx;               // [L][ldw] predicted means HA x_t
    double* Z;                // [L][ldw] pv/sqrt(S) - U^T y / sigma
    double* UU;               // [W][M]  U U^T y_t
    double* rt;               // [W]     ||y_t - U U^T y_t||
    double* spu;              // [L]     sum_t pv * (U^T y)
    double* nll;              // [L]
    double* gl;               // [L][P]
    double* x;                // [L][d]
    double* dx;               // [L][P][d]
    const int* tmiss = nullptr;   // [W] 1 where y_t holds NaN (missing outputs), or NULL when the window holds none
};
int launch_window_objective(const TickArgs& a, const double* cb64, const float* cb32, const WindowBufs& w, int* fallback,
                            double* loss /* device scalar */, double* grad /* device [M*L+L+1+L*P] */, hipStream_t s,
                            int kernel = 0 /* kernel id: a stacked one takes the stacked sweep */);

}  // namespace moihgp
