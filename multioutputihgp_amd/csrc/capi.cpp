// capi.cpp -- the C ABI of libmoihgp.so: the reference's 26 gp32_*/gp52_* symbols
// (reference moihgp/src/wrapper.cpp:31-624) plus the additive batched entries of include/moihgp.h.
//
// Host side = parameter bookkeeping + kernel orchestration only.  Every arithmetic step of the reference path runs in a HIP
// kernel (stationary*.hip, tick.hip, recursion*.hip, grad*.hip, gemm_mfma.hip, polar.hip, window.hip), including the polar
// factor of the mixing matrix in update() (moihgp.h:433-447); the host draws the constructor's random matrix and moves bytes.
#include "../../include/moihgp.h"
#include "common.h"

#include <cmath>
#include <cstdarg>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <random>
#include <string>
#include <vector>

namespace moihgp {

static thread_local char g_last_error[512] = "";

void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

[[noreturn]] void fatal_hip(hipError_t e, const char* what, const char* file, int line) {
    (void)hipGetLastError();                   // clear the sticky error: the next call of an entry with a return code starts clean
    throw HipFailure{e, what, file, line};
}

template <typename T>
static T* dev_alloc(size_t n) {
    void* p = nullptr;
    MOIHGP_HIP_FATAL(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    return static_cast<T*>(p);
}

}  // namespace moihgp

using namespace moihgp;

struct moihgp_gp {
    int kernel = 0;
    double dt = 0;
    size_t M = 0, L = 0;
    int d = 2, P = kNumIgpParam;
    size_t num_param = 0;
    bool latents_only = false;
    // moihgp.h:747 `_threading` (ctor argument, forced off for L < 2 by :128-135).  There is no pthread fan-out here, but the flag
    // is observable in the reference: the gradient overload of negLogLikelihood returns the per-latent losses only on its threaded
    // branch (:590), the serial branch (:597-607) drops them.  lik1_full (MOIHGP_LIK1_FULL_LOSS=1) asks for the summed form always.
    bool threading = false, lik1_full = false;
    // host mirrors of the parameters (moihgp.h:741-744 + per-latent matern*ss.h:_params)
    std::vector<double> U, S, igp;
    double sigma = 1e-2;
    // device state
    hipStream_t stream = nullptr;
    float* dU32 = nullptr;     // fp32 image of dU for the fp32 stream products, rebuilt on first use after U changed
    bool u32_valid = false;
    double *dU = nullptr, *dS = nullptr, *dsqrtS = nullptr, *dinvsqrtS = nullptr, *dsigma = nullptr, *dparams = nullptr, *cb64 = nullptr;
    float* cb32 = nullptr;
    // per-tick staging
    double *dx = nullptr, *dy = nullptr, *ddx = nullptr;       // one device block [x | y | dx]
    int* dunstable = nullptr;                                  // device int[4]: latents flagged unstable (fp64, fp32 blocks); [2]: fp32 only, swept in fp64;
    int n_unstable[4] = {0, 0, 0, 0};                          // [3]: stacked models, filters that remember more than ~1000 ticks
    int* drescue_idx = nullptr;                                // stacked models, 1024 latents and more: the latents counted in n_unstable[2]
    void* drescue = nullptr; size_t rescue_cap = 0;            // their compact fp64 bank (filter_stream, fp32 streams)
    void* drescue_const = nullptr; size_t rescue_const_cap = 0; // ... its constant blocks and team-kernel powers (per update)
    double* dpart = nullptr;                                   // [32][L] chunk partials of the per-tick projection
    double *dTy = nullptr, *dUty = nullptr, *dTyhat = nullptr, *dloss = nullptr, *dgrad = nullptr, *dscratch = nullptr;
    double* dwork = nullptr;   // L*L + L, lazily (missing-output projection)
    double* dpolar = nullptr;  // polar_work_doubles(M, L), lazily (device polar factor)
    int* dfallback = nullptr;  // [2 L + 1] flags of the latents the gradient sweep leaves to its later passes, their compact list, its length
    double* dlink = nullptr;   // [L][144] stacked filter: where the second (broken-link) pass resumes a latent (on first use)
    bool hp_valid = false;     // dhp matches the current tables (cleared by every IHGP::update)
    // options (moihgp_set_option; defaults from the environment, read once in gp_create)
    int opt_filter_split = 0, opt_filter_variant = 0, opt_filter_maxlinks = -1, opt_filter_team = -1, opt_filter_plain_x = -1;
    int* dwinmiss = nullptr;   // [W] window objective: 1 where the tick's observation vector holds NaN (on first use)
    size_t winmiss_cap = 0;
    bool win_has_nan = false;
    int polar_its = 0;         // Newton-Schulz steps of the last device polar factor (0: single-workgroup kernel / none yet)
    int polar_warm = 0;        // dpolar holds the outlying subspace of the previous polar factor (polar_deflate.hip warm start)
    int opt_filter_impute = -1; // option "filter_impute": missing ticks of the stacked many-latent sweep by imputation (filter_x_gaps_a / _b_kernel): -1 = for d >= 8,
                                // 0 never, 1 always
    void* dgap = nullptr;       // its scratch: impulse responses and the lists of gaps (gap_bank_bytes), on first use
    size_t gap_cap = 0;
    unsigned long long cb_version = 1;        // bumped whenever the constant blocks are rebuilt
    unsigned long long gap_imp_version = 0;   // cb_version the bank's impulse responses were swept for
    unsigned long long gap_sig = 0;   // scalar type the bank's constant parts (at its head) were written for
    int opt_polar_warm = 0;    // option "polar_warm_start": use it (off by default: update() is then a function of its argument alone, bit for bit)
    double* dhp = nullptr;     // [L][gradx_hp_len(d)] HA AKHA^k rows of the stacked models' time-parallel gradient sweep (on first use)
    double* dxscratch = nullptr; // stacked kernels, few latents: per-slice NLL partials
    int* dlinkflags = nullptr;   // stacked filter, 1024 latents and more: hand-over flags of its second pass (zero between sweeps: the second pass clears what it takes)
    double* dtp64 = nullptr;     // fewer than 1024 latents: scan powers of the chunk-templated team kernel (launch_team_powers), per update
    float* dtp32 = nullptr;
    double* dxc64 = nullptr;     // the reference's own models, fewer than 1024 latents: their tables in the stacked layout (launch_xc_from_cb), per update
    float* dxc32 = nullptr;
    double* cbd64 = nullptr;     // stacked kernels: sensitivity blocks (XD), fp64; filled once somebody asks for gradients
    bool sens_wanted = false, sens_valid = false;
    bool U_host_stale = false; // the device holds a newer U than the host mirror (fetched on getParams)
    bool mix_ortho = true;     // U^T U = I to 1e-9 (always after update() / the constructor: a polar factor; checked after moihgp_set_mixing)
    double *hin = nullptr, *hout = nullptr, *hgrad = nullptr;   // page-locked, device-mapped per-tick staging
    unsigned long long* hflag = nullptr;                        // mapped completion word of the fused small-model step
    unsigned long long seq = 0;
    bool fused_ok = false;
    bool polar_pending = false;                                 // a small-matrix polar factor whose verdict has not been read yet
    std::vector<void*> pinned; // caller buffers page-locked through moihgp_pin_host_buffer
    // caller streams that carried batched work of this handle since the last rewrite of its tables / mixing, and the event used to
    // make the handle's own stream wait for them before the next rewrite (order_after_sweeps)
    std::vector<hipStream_t> user_streams;
    hipEvent_t order_ev = nullptr;
    // window objective (moihgp_window_set / moihgp_window_eval)
    WindowBufs win{};
    double* dwin = nullptr;
    size_t win_cap = 0;
    // optional kernel-exact timing of filter launches (moihgp_profile_enable)
    unsigned prof_stride = 1, prof_seen = 0;     // every prof_stride-th launch carries an event pair (moihgp_profile_stride)
    std::vector<hipEvent_t> prof_ev;
    int prof_n = 0;

    TickArgs tick() const { return TickArgs{d, M, L, cb64, dU, dS, dsqrtS, dinvsqrtS, dsigma, (threading || lik1_full) ? 1 : 0, P, cbd64}; }
};

static void gp_free(moihgp_gp* g) {
    if (!g) return;
    void* ptrs[] = {g->dU, g->dS, g->dsqrtS, g->dinvsqrtS, g->dsigma, g->dparams, g->cb64, g->cb32, g->dx, g->dpart, g->dTy, g->dUty, g->dTyhat, g->dloss, g->dgrad, g->dscratch, g->dwork, g->dpolar, g->dfallback, g->dwin, g->dunstable, g->dxscratch, g->cbd64, g->dU32, g->dhp, g->dlink, g->dwinmiss, g->dtp64, g->dtp32, g->dxc64, g->dxc32, g->dlinkflags, g->dgap, g->drescue_idx, g->drescue, g->drescue_const};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (void* p : g->pinned) (void)hipHostUnregister(p);
    if (g->hin) (void)hipHostFree(g->hin);
    if (g->hout) (void)hipHostFree(g->hout);
    if (g->hgrad) (void)hipHostFree(g->hgrad);
    if (g->hflag) (void)hipHostFree(g->hflag);
    for (hipEvent_t e : g->prof_ev) (void)hipEventDestroy(e);
    if (g->order_ev) (void)hipEventDestroy(g->order_ev);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

// The batched entries (moihgp_filter_stream*, moihgp_grad_stream, moihgp_project/unproject_stream) are asynchronous on the CALLER's
// stream and read the constant blocks and the mixing; update / update_latents / set_mixing / reseed_U rewrite those on the handle's
// own (non-blocking) stream.  note_user_stream remembers which streams carried such work; order_after_sweeps, called before every
// rewrite, makes the handle's stream wait for everything those streams hold at that moment, so a rewrite issued while sweeps are still
// in flight (pipelined objective evaluations followed by an update) cannot overwrite tables under them.  Nothing is recorded per
// launch: the hot path pays nothing.
static void note_user_stream(moihgp_gp* g, hipStream_t s) {
    for (hipStream_t u : g->user_streams) if (u == s) return;
    g->user_streams.push_back(s);
}
static void order_after_sweeps(moihgp_gp* g) {
    if (g->user_streams.empty()) return;
    if (!g->order_ev) MOIHGP_HIP_FATAL(hipEventCreateWithFlags(&g->order_ev, hipEventDisableTiming));
    for (hipStream_t u : g->user_streams) {
        if (hipEventRecord(g->order_ev, u) != hipSuccess) { (void)hipGetLastError(); continue; }   // a stream the caller has destroyed since
        MOIHGP_HIP_FATAL(hipStreamWaitEvent(g->stream, g->order_ev, 0));
    }
    g->user_streams.clear();
}

// `_on` entries (moihgp_update_dev_on, moihgp_window_eval_dev_on): the handle's stream takes its place behind whatever the caller has queued
// on `s` so far (the operands need not be complete on the host side), and `s` waits for what the handle has queued -- events only, no host
// synchronisation.  s == the handle's own stream: nothing to do.
static void wait_for_caller(moihgp_gp* g, hipStream_t s) {
    if (s == g->stream) return;
    if (!g->order_ev) MOIHGP_HIP_FATAL(hipEventCreateWithFlags(&g->order_ev, hipEventDisableTiming));
    MOIHGP_HIP_FATAL(hipEventRecord(g->order_ev, s));
    MOIHGP_HIP_FATAL(hipStreamWaitEvent(g->stream, g->order_ev, 0));
}
static void caller_waits(moihgp_gp* g, hipStream_t s) {
    if (s == g->stream) return;
    if (!g->order_ev) MOIHGP_HIP_FATAL(hipEventCreateWithFlags(&g->order_ev, hipEventDisableTiming));
    MOIHGP_HIP_FATAL(hipEventRecord(g->order_ev, g->stream));
    MOIHGP_HIP_FATAL(hipStreamWaitEvent(s, g->order_ev, 0));
}

static void upload_mixing(moihgp_gp* g) {
    if (g->latents_only) return;
    order_after_sweeps(g);
    g->u32_valid = false;
    if (!g->U_host_stale)    // otherwise the device copy is the current one (device polar factor)
        MOIHGP_HIP_FATAL(hipMemcpyAsync(g->dU, g->U.data(), sizeof(double) * g->M * g->L, hipMemcpyHostToDevice, g->stream));
    MOIHGP_HIP_FATAL(hipMemcpyAsync(g->dS, g->S.data(), sizeof(double) * g->L, hipMemcpyHostToDevice, g->stream));
    launch_scales(g->dS, g->L, g->dsqrtS, g->dinvsqrtS, g->stream);
    MOIHGP_HIP_FATAL(hipMemcpyAsync(g->dsigma, &g->sigma, sizeof(double), hipMemcpyHostToDevice, g->stream));
    // fp32 image of the mixing for the fp32 stream products, once somebody has asked for it: rebuilt here, on the handle's stream
    // (every caller of upload_mixing synchronises it), so that the batched entries only ever READ it, whatever stream they run on
    if (g->dU32) { launch_narrow(g->dU, g->M * g->L, g->dU32, g->stream); g->u32_valid = true; }
}

static void run_ihgp_update(moihgp_gp* g) {
    order_after_sweeps(g);
    g->hp_valid = false;
    g->cb_version++;
    MOIHGP_HIP_FATAL(hipMemcpyAsync(g->dparams, g->igp.data(), sizeof(double) * g->L * g->P, hipMemcpyHostToDevice, g->stream));
    if (kernel_stack(g->kernel)) {
        // the sensitivities cost nine more 100-iteration Lyapunov solves per latent at d = 12: only for handles that use them
        if (g->L >= 1024 && !g->drescue_idx) g->drescue_idx = dev_alloc<int>(g->L);
        launch_stack_update(g->kernel, g->dt, g->dparams, g->L, g->cb64, g->cb32, g->sens_wanted ? g->cbd64 : nullptr, g->dunstable, g->drescue_idx, g->stream);
        g->sens_valid = g->sens_wanted;
        // the time-parallel gradient sweep's per-latent tables follow the sensitivity blocks: rebuilt HERE, on the handle's stream (which this
        // function synchronises), so that hp_valid never claims a table nobody built and sweeps on any caller stream find it complete
        if (g->dhp && g->sens_valid) {
            if (launch_gp_table_x(g->kernel, g->cb64, g->cbd64, g->dhp, g->L, g->stream)) throw HipFailure{hipErrorLaunchFailure, "gp_table_kernel", __FILE__, __LINE__};
            g->hp_valid = true;
        }
        if (g->L < 1024) {
            if (!g->dtp64) { g->dtp64 = dev_alloc<double>(g->L * team_powers_elems(g->d)); g->dtp32 = dev_alloc<float>(g->L * team_powers_elems(g->d)); }
            launch_team_powers(g->kernel, g->cb64, g->L, g->dtp64, g->dtp32, g->stream);
        }
    }
    else {
        launch_ihgp_update(g->kernel, g->d, g->dt, g->dparams, g->L, g->cb64, g->cb32, g->dunstable, g->stream);
        // the stacked filter's kernels serve these models too (one component): their tables in its layout, from the CB blocks just written
        if (!g->dxc64) { g->dxc64 = dev_alloc<double>(g->L * (size_t)xc_size(g->d)); g->dxc32 = dev_alloc<float>(g->L * (size_t)xc_size(g->d)); }
        launch_xc_from_cb(g->d, g->cb64, g->L, g->dxc64, g->dxc32, g->stream);
        if (g->L < 1024) {
            if (!g->dtp64) { g->dtp64 = dev_alloc<double>(g->L * team_powers_elems(g->d)); g->dtp32 = dev_alloc<float>(g->L * team_powers_elems(g->d)); }
            launch_team_powers(g->kernel | (1 << 4), g->dxc64, g->L, g->dtp64, g->dtp32, g->stream);
        }
    }
    MOIHGP_HIP_FATAL(hipMemcpyAsync(g->n_unstable, g->dunstable, 4 * sizeof(int), hipMemcpyDeviceToHost, g->stream));
    MOIHGP_HIP_FATAL(hipStreamSynchronize(g->stream));
    if (kernel_stack(g->kernel) && g->drescue_idx && g->n_unstable[2] > 0) {
        // fp32 sweeps of this bank take n_unstable[2] latents in fp64 on the side (filter_stream_impl): their constant blocks and the few-latents team
        // kernel's scan powers, compact, once per update
        const size_t n = (size_t)g->n_unstable[2], xcs = (size_t)xc_size(g->d), tpe = team_powers_elems(g->d);
        const size_t need = n * (xcs + tpe) * sizeof(double) + n * tpe * sizeof(float);
        if (g->rescue_const_cap < need) {
            if (g->drescue_const) { MOIHGP_HIP_FATAL(hipDeviceSynchronize()); MOIHGP_HIP_FATAL(hipFree(g->drescue_const)); g->drescue_const = nullptr; g->rescue_const_cap = 0; }
            MOIHGP_HIP_FATAL(hipMalloc(&g->drescue_const, need));
            g->rescue_const_cap = need;
        }
        double* cbc = static_cast<double*>(g->drescue_const);
        double* tpd = cbc + n * xcs;
        float* tpf = reinterpret_cast<float*>(tpd + n * tpe);
        launch_rescue_gather(nullptr, 0, 0, g->drescue_idx, n, g->cb64, (int)xcs, nullptr, g->d, nullptr, cbc, nullptr, g->stream);
        launch_team_powers(g->kernel, cbc, n, tpd, tpf, g->stream);
        MOIHGP_HIP_FATAL(hipStreamSynchronize(g->stream));
    }
}

static void ensure_sensitivities(moihgp_gp* g) {
    if (kernel_stack(g->kernel) && !g->sens_valid) { g->sens_wanted = true; run_ihgp_update(g); }
}

static bool compute_polar_fwd(moihgp_gp* g, const double* Uparam);
static void draw_U(moihgp_gp* g, unsigned long long seed, bool use_seed) {
    // moihgp.h:103-125: U = polar(I + N(0, 1e-3))
    std::mt19937 gen;
    if (use_seed) gen.seed((unsigned)(seed ^ (seed >> 32)));
    else { std::random_device rd; gen.seed(rd()); }
    std::normal_distribution<> distr(0.0, 1e-3);
    std::vector<double> I(g->M * g->L, 0.0);
    for (size_t r = 0; r < g->M; r++)
        for (size_t c = 0; c < g->L; c++) I[r * g->L + c] = (r == c ? 1.0 : 0.0) + distr(gen);
    compute_polar_fwd(g, I.data());
}

static moihgp_gp* gp_create(int kernel, double dt, size_t M, size_t L, bool latents_only, const double* params_LP, bool threading = false) {
    g_last_error[0] = 0;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_last_error("libmoihgp: no usable HIP device (%s); this library has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        std::fprintf(stderr, "%s\n", g_last_error);
        return nullptr;
    }
    const int kbase = kernel_base(kernel), kstack = kernel_stack(kernel);
    if ((kbase != MOIHGP_MATERN32 && kbase != MOIHGP_MATERN52) || kstack == 1 || kstack > 4) { set_last_error("unknown kernel id %d", kernel); return nullptr; }
    if (L == 0) { set_last_error("num_latent must be >= 1"); return nullptr; }
    if (!latents_only && M < L) {
        // moihgp.h:510 indexes y(idx) for idx < num_latent and the thin SVD needs full column rank
        set_last_error("num_output (%zu) must be >= num_latent (%zu)", M, L);
        return nullptr;
    }
    moihgp_gp* g = new moihgp_gp();
    try {
    g->kernel = kernel; g->dt = dt; g->M = M; g->L = L; g->latents_only = latents_only;
    g->threading = (L < 2) ? false : threading;                          // moihgp.h:128-135
    { const char* fl = std::getenv("MOIHGP_LIK1_FULL_LOSS"); g->lik1_full = fl && fl[0] == '1'; }
    // tuning / test hooks: the environment is consulted here, once; moihgp_set_option changes them per handle afterwards
    if (const char* e = std::getenv("MOIHGP_FILTER_SPLIT")) g->opt_filter_split = std::atoi(e);
    if (const char* e = std::getenv("MOIHGP_FILTER_MAXLINKS")) g->opt_filter_maxlinks = std::atoi(e);
    if (const char* e = std::getenv("MOIHGP_FILTER_TEAM")) g->opt_filter_team = std::atoi(e);
#ifdef MOIHGP_TUNING
    if (const char* e = std::getenv("MOIHGP_FILTER_VARIANT")) g->opt_filter_variant = std::atoi(e);
#endif
    g->d = (kbase == MOIHGP_MATERN32) ? 2 : 3;                           // matern32ss.h:95, matern52ss.h:106
    if (kstack) { g->d *= kstack; g->P = 2 * kstack + 1; }               // (magnitude_j, lengthscale_j) x J, noise
    g->num_param = M * L + L + 1 + L * g->P;                             // moihgp.h:93
    MOIHGP_HIP_FATAL(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    const size_t cbs = kstack ? (size_t)xc_size(g->d) : (size_t)cb_size(g->d);
    g->dparams = dev_alloc<double>(L * g->P);
    g->cb64 = dev_alloc<double>(L * cbs);
    g->cb32 = dev_alloc<float>(L * cbs);
    if (kstack) g->cbd64 = dev_alloc<double>(L * (size_t)xd_size(g->d, g->P));
    g->dfallback = dev_alloc<int>(2 * L + 1);
    g->dunstable = dev_alloc<int>(4);
    MOIHGP_HIP_FATAL(hipMemset(g->dunstable, 0, 4 * sizeof(int)));
    g->igp.resize(L * g->P);
    for (size_t l = 0; l < L; l++) {
        if (params_LP) for (int p = 0; p < g->P; p++) g->igp[l * g->P + p] = params_LP[l * g->P + p];
        else if (kstack) { for (int j = 0; j < kstack; j++) { g->igp[l * g->P + 2 * j] = 1.0; g->igp[l * g->P + 2 * j + 1] = (double)(j + 1); } g->igp[l * g->P + 2 * kstack] = 0.1; }
        else { g->igp[l * g->P + 0] = 1.0; g->igp[l * g->P + 1] = 1.0; g->igp[l * g->P + 2] = 0.1; }   // matern32ss.h:34-36
    }
    if (!latents_only) {
        g->U.assign(M * L, 0.0);
        g->S.assign(L, 1.0);                                             // moihgp.h:126
        g->sigma = 1e-2;                                                 // moihgp.h:127
        g->dU = dev_alloc<double>(M * L);
        MOIHGP_HIP_FATAL(hipHostMalloc((void**)&g->hflag, 64, hipHostMallocMapped));   // [0] completion word of the fused kernels, [1] polar verdict
        std::memset(g->hflag, 0, 64);
        draw_U(g, 0, false);
        g->dS = dev_alloc<double>(L);
        g->dsqrtS = dev_alloc<double>(L);
        g->dinvsqrtS = dev_alloc<double>(L);
        g->dsigma = dev_alloc<double>(1);
        {   // inputs of the per-tick ABI as ONE device block [x | y | dx] (a single packed copy per call)
            const size_t nin = L * g->d + M + L * g->P * g->d;
            g->dx = dev_alloc<double>(nin);
            g->dy = g->dx + L * g->d;
            g->ddx = g->dy + M;
            MOIHGP_HIP_FATAL(hipHostMalloc((void**)&g->hin, sizeof(double) * nin, hipHostMallocMapped));
            const char* fe = std::getenv("MOIHGP_TICK_FUSED");            // 0: always take the multi-kernel path
            g->fused_ok = !kstack && fused_step_fits(M, L) && !(fe && fe[0] == '0');     // (the one-workgroup kernels are built for the reference's two models)
            MOIHGP_HIP_FATAL(hipHostMalloc((void**)&g->hout, sizeof(double) * (nin + 8), hipHostMallocMapped));
            if (g->num_param * sizeof(double) <= (size_t)1 << 20)
                MOIHGP_HIP_FATAL(hipHostMalloc((void**)&g->hgrad, sizeof(double) * g->num_param, hipHostMallocMapped));
        }
        g->dTy = dev_alloc<double>(L);
        g->dpart = dev_alloc<double>(32 * L);
        g->dUty = dev_alloc<double>(L);
        g->dTyhat = dev_alloc<double>(L);
        g->dloss = dev_alloc<double>(1);
        g->dgrad = dev_alloc<double>(g->num_param);
        g->dscratch = dev_alloc<double>(2 * L + L * g->P + M + 8);
        upload_mixing(g);
    }
    run_ihgp_update(g);                                                  // moihgp.h:86-90 -> ihgp.h:33
    } catch (...) { gp_free(g); throw; }                                 // (an allocation or launch failed: release what exists, report upstream)
    return g;
}

// ---- per-tick paths (moihgp.h:148-428) ----------------------------------------------------------
static bool has_nan(const double* y, size_t n) {                       // moihgp.h:150-158
    for (size_t i = 0; i < n; i++)
        if (y[i] != y[i]) return true;
    return false;
}

// ---- per-tick staging: ONE packed host->device copy from a page-locked block [x | y | dx]; the kernels write their
// results (xnew, yhat, dxnew, loss, and the gradient when it is small) straight into a page-locked, device-mapped host
// block, so a call costs one async copy + the kernel launches + one stream synchronisation.
static void stage_inputs(moihgp_gp* g, const double* x, const double* y, const double* dx) {
    const size_t L = g->L, d = g->d, P = g->P, M = g->M;
    double* h = g->hin;
    std::memcpy(h, x, sizeof(double) * L * d);
    if (y) std::memcpy(h + L * d, y, sizeof(double) * M);
    if (dx) std::memcpy(h + L * d + M, dx, sizeof(double) * L * P * d);
    const size_t n = L * d + M + (dx ? L * P * d : 0);           // dx sits last: skipped when absent
    MOIHGP_HIP_FATAL(hipMemcpyAsync(g->dx, h, sizeof(double) * n, hipMemcpyHostToDevice, g->stream));
}

static void do_project(moihgp_gp* g, const double* y_host) {
    TickArgs a = g->tick();
    launch_project_tick(a, g->dy, g->dTy, g->dUty, g->dpart, g->stream);   // moihgp.h:181
    if (has_nan(y_host, g->M)) {                                         // moihgp.h:167-178
        if (!g->dwork) g->dwork = dev_alloc<double>(g->L * g->L + g->L);
        launch_project_tick_missing(a, g->dy, g->dTy, g->dwork, g->stream);
    }
}

// completion of a fused small-model kernel: spin on the sequence number it writes last into mapped host memory
static void wait_flag(moihgp_gp* g, unsigned long long seq) {
    volatile unsigned long long* f = g->hflag;
    for (long spins = 0; *f != seq; spins++) {
        __builtin_ia32_pause();
        if (spins > 4000000) { MOIHGP_HIP_FATAL(hipStreamSynchronize(g->stream)); break; }   // far beyond any healthy call: let the runtime report
    }
    std::atomic_thread_fence(std::memory_order_acquire);
}

static void do_step(moihgp_gp* g, const double* x, const double* y, const double* dx, double* xnew, double* yhat, double* dxnew) {
    if (g->latents_only) { set_last_error("per-tick ABI needs a full MOIHGP object"); std::fprintf(stderr, "%s\n", g_last_error); std::abort(); }
    const size_t L = g->L, d = g->d, P = g->P, M = g->M;
    order_after_sweeps(g);                       // handle-owned scratch (flags, partial sums) is shared with sweeps still in flight on caller streams
    if (dx) ensure_sensitivities(g);             // (stacked kernels compute dAKHA, dK, .. from the first call that needs them on)
    double* o_x = g->hout;                       // mapped host block: [xnew | yhat | dxnew | loss]
    double* o_y = o_x + L * d;
    double* o_dx = o_y + M;
    if (g->fused_ok && !(y && has_nan(y, M))) {
        // small model, every output observed: one workgroup does project -> step -> unproject on the mapped blocks
        double* h = g->hin;
        std::memcpy(h, x, sizeof(double) * L * d);
        if (y) std::memcpy(h + L * d, y, sizeof(double) * M);
        if (dx) std::memcpy(h + L * d + M, dx, sizeof(double) * L * P * d);
        const unsigned long long seq = ++g->seq;
        launch_fused_step(g->tick(), h, y ? h + L * d : nullptr, dx ? h + L * d + M : nullptr, o_x, yhat ? o_y : nullptr, dx ? o_dx : nullptr,
                          g->hflag, seq, g->stream);
        wait_flag(g, seq);
        std::memcpy(xnew, o_x, sizeof(double) * L * d);
        if (yhat) std::memcpy(yhat, o_y, sizeof(double) * M);
        if (dx && dxnew) std::memcpy(dxnew, o_dx, sizeof(double) * L * P * d);
        return;
    }
    stage_inputs(g, x, y, dx);
    if (y) do_project(g, y);
    TickArgs a = g->tick();
    launch_step_tick(a, g->dx, y ? g->dTy : nullptr, dx ? g->ddx : nullptr, o_x, g->dTyhat, dx ? o_dx : nullptr, g->stream);
    if (yhat) launch_unproject_tick(a, g->dTyhat, o_y, g->stream);
    MOIHGP_HIP_FATAL(hipStreamSynchronize(g->stream));
    std::memcpy(xnew, o_x, sizeof(double) * L * d);
    if (yhat) std::memcpy(yhat, o_y, sizeof(double) * M);
    if (dx && dxnew) std::memcpy(dxnew, o_dx, sizeof(double) * L * P * d);
}

static double do_lik(moihgp_gp* g, const double* x, const double* y, const double* dx, double* grad) {
    if (g->latents_only) { set_last_error("per-tick ABI needs a full MOIHGP object"); std::fprintf(stderr, "%s\n", g_last_error); std::abort(); }
    const size_t L = g->L, d = g->d, P = g->P, M = g->M;
    order_after_sweeps(g);
    if (dx) ensure_sensitivities(g);
    double* o_loss = g->hout + L * d + M + L * P * d;
    const bool small_grad = g->hgrad != nullptr;                         // gradient written straight to mapped host memory
    if (g->fused_ok && small_grad && fused_lik_fits(M, L) && !has_nan(y, M)) {
        double* h = g->hin;
        std::memcpy(h, x, sizeof(double) * L * d);
        std::memcpy(h + L * d, y, sizeof(double) * M);
        if (dx) std::memcpy(h + L * d + M, dx, sizeof(double) * L * P * d);
        const unsigned long long seq = ++g->seq;
        launch_fused_lik(g->tick(), h, h + L * d, dx ? h + L * d + M : nullptr, o_loss, g->hgrad, g->hflag, seq, g->stream);
        wait_flag(g, seq);
        if (dx && grad) std::memcpy(grad, g->hgrad, sizeof(double) * g->num_param);
        return *o_loss;
    }
    stage_inputs(g, x, y, dx);
    do_project(g, y);
    TickArgs a = g->tick();
    launch_nll_tick(a, g->dx, g->dy, g->dTy, g->dUty, dx ? g->ddx : nullptr, o_loss, small_grad ? g->hgrad : g->dgrad, g->dscratch, g->stream);
    if (dx && grad && !small_grad) MOIHGP_HIP_FATAL(hipMemcpyAsync(grad, g->dgrad, sizeof(double) * g->num_param, hipMemcpyDeviceToHost, g->stream));
    MOIHGP_HIP_FATAL(hipStreamSynchronize(g->stream));
    if (dx && grad && small_grad) std::memcpy(grad, g->hgrad, sizeof(double) * g->num_param);
    return *o_loss;
}

// U = polar(Uparam) (moihgp.h:433-447), always on the device: Newton-Schulz on the MFMA GEMMs (polar.hip), or, for small
// matrices (the 8 x 4 demo of example.py, a 64 x 16 mixing), the same iteration as a single workgroup in LDS -- one launch.
// MOIHGP_POLAR=gemm forces the multi-kernel path.  The host mirror of U is refreshed lazily (getParams).
static bool compute_polar(moihgp_gp* g, const double* Uparam, bool from_device = false);
static bool compute_polar_fwd(moihgp_gp* g, const double* Uparam) { return compute_polar(g, Uparam); }
static bool compute_polar(moihgp_gp* g, const double* Uparam, bool from_device) {
    const size_t M = g->M, L = g->L;
    bool small = polar_small_fits(M, L);
    if (const char* e = std::getenv("MOIHGP_POLAR")) { if (e[0] == 'g') small = false; }
    order_after_sweeps(g);
    g->u32_valid = false;
    MOIHGP_HIP_FATAL(hipMemcpyAsync(g->dU, Uparam, sizeof(double) * M * L, from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g->stream));
    int its = 0;
    if (small) {
        // asynchronous: the kernel leaves its verdict in mapped host memory (and NaN in U if the input is rank deficient);
        // do_update reads it after the synchronisation that ends the update anyway
        int* verdict = reinterpret_cast<int*>(g->hflag + 1);
        *verdict = 0;
        launch_polar_small(g->dU, M, L, verdict, g->stream);
        g->polar_pending = true;
    } else {
        if (!g->dpolar) g->dpolar = dev_alloc<double>(polar_work_doubles(M, L));
        its = polar_factor_device(g->dU, M, L, g->dpolar, g->stream, g->opt_polar_warm ? &g->polar_warm : nullptr);
    }
    g->polar_its = its > 0 ? its : 0;
    if (its < 0) return false;
    g->U_host_stale = true;                      // 8*M*L bytes over PCIe only when somebody asks (getParams)
    g->mix_ortho = true;                         // a polar factor
    return true;
}

static void do_update(moihgp_gp* g, const double* params, bool from_device = false) {          // moihgp.h:431-457
    const size_t M = g->M, L = g->L, sizeU = M * L;
    std::vector<double> tail;
    if (from_device) {
        // device-resident parameters (moihgp_update_dev): the mixing part goes device -> device into the polar factor; the small tail
        // [S | sigma | per-latent values] is mirrored to the host (getParams, the per-tick bookkeeping) -- 8 (L + 1 + L P) bytes
        tail.resize(L + 1 + L * (size_t)g->P);
        order_after_sweeps(g);
        MOIHGP_HIP_FATAL(hipMemcpyAsync(tail.data(), params + sizeU, sizeof(double) * tail.size(), hipMemcpyDeviceToHost, g->stream));
    }
    const bool polar_ok = compute_polar(g, params, from_device);
    if (from_device) MOIHGP_HIP_FATAL(hipStreamSynchronize(g->stream));   // (the tail has arrived; the multi-kernel polar factor has synchronised already)
    const double* tp = from_device ? tail.data() : params + sizeU;       // [S | sigma | per-latent values]
    if (!polar_ok) {
        set_last_error("update: mixing matrix is rank deficient");
        std::fprintf(stderr, "libmoihgp: %s\n", g_last_error);
        for (auto& u : g->U) u = std::nan("");
        g->U_host_stale = false;
    }
    for (size_t l = 0; l < L; l++) g->S[l] = tp[l];                      // moihgp.h:448
    g->sigma = tp[L];                                                    // moihgp.h:449
    for (size_t i = 0; i < L * (size_t)g->P; i++) g->igp[i] = tp[L + 1 + i];   // moihgp.h:450-456
    upload_mixing(g);
    run_ihgp_update(g);                                                  // ends with a stream synchronisation
    if (g->polar_pending) {
        g->polar_pending = false;
        if (*reinterpret_cast<volatile int*>(g->hflag + 1) < 0) {
            set_last_error("update: mixing matrix is rank deficient");
            std::fprintf(stderr, "libmoihgp: %s\n", g_last_error);
        }
    }
}

static void do_get_params(moihgp_gp* g, double* params) {            // moihgp.h:721-738
    const size_t M = g->M, L = g->L, sizeU = M * L;
    if (g->U_host_stale) {
        MOIHGP_HIP_FATAL(hipMemcpyAsync(g->U.data(), g->dU, sizeof(double) * sizeU, hipMemcpyDeviceToHost, g->stream));
        MOIHGP_HIP_FATAL(hipStreamSynchronize(g->stream));
        g->U_host_stale = false;
    }
    std::memcpy(params, g->U.data(), sizeof(double) * sizeU);
    std::memcpy(params + sizeU, g->S.data(), sizeof(double) * L);
    params[sizeU + L] = g->sigma;
    std::memcpy(params + sizeU + L + 1, g->igp.data(), sizeof(double) * L * g->P);
}

static int gp52_kernel() {
    // wrapper.cpp:22 typedefs GP52 to the Matern-3/2 model; keep that unless told otherwise.
    const char* e = std::getenv("MOIHGP_GP52_MATERN52");
    return (e && e[0] == '1') ? MOIHGP_MATERN52 : MOIHGP_MATERN32;
}

// The reference ABI has no status channel (all void / value returns, wrapper.cpp:31-326), so a device failure cannot be reported to
// the caller of a gpXX_* entry: say what happened and stop.
[[noreturn]] static void abort_on(const HipFailure& f) {
    std::fprintf(stderr, "libmoihgp: HIP error %d (%s) at %s:%d: %s\n", (int)f.err, hipGetErrorString(f.err), f.file, f.line, f.what);
    std::abort();
}
// body of a reference-ABI entry: nothing may unwind through extern "C"
template <typename F>
static auto guard_abort(F&& body) -> decltype(body()) {
    try { return body(); }
    catch (const HipFailure& f) { abort_on(f); }
    catch (const std::exception& e) { std::fprintf(stderr, "libmoihgp: %s\n", e.what()); std::abort(); }
}
// body of an additive entry with an int return code: failures become rc 2 (HIP) / 4 (host memory) + moihgp_last_error()
template <typename F>
static int guard_rc(F&& body) {
    try { return body(); }
    catch (const HipFailure& f) {
        set_last_error("HIP error %d (%s) at %s:%d: %s", (int)f.err, hipGetErrorString(f.err), f.file, f.line, f.what);
        return 2;
    }
    catch (const std::exception& e) { set_last_error("%s", e.what()); return 4; }
}
// constructors: NULL + moihgp_last_error() (the reference's ctor cannot fail short of std::bad_alloc)
template <typename F>
static moihgp_gp* guard_new(F&& body) {
    try { return body(); }
    catch (const HipFailure& f) {
        set_last_error("HIP error %d (%s) at %s:%d: %s", (int)f.err, hipGetErrorString(f.err), f.file, f.line, f.what);
        std::fprintf(stderr, "libmoihgp: %s\n", g_last_error);
        return nullptr;
    }
    catch (const std::exception& e) { set_last_error("%s", e.what()); return nullptr; }
}

extern "C" {

#define MOIHGP_DEFINE_REFERENCE_ABI(PFX, KERNEL_EXPR)                                                                   \
    moihgp_gp* PFX##_new(double dt, size_t num_output, size_t num_latent, bool threading) {                             \
        return guard_new([&] { return gp_create((KERNEL_EXPR), dt, num_output, num_latent, false, nullptr, threading); }); \
    }                                                                                                                    \
    void PFX##_del(moihgp_gp* gp) { gp_free(gp); }                                                                       \
    void PFX##_step1(moihgp_gp* gp, double* x, double* y, double* dx, double* xnew, double* yhat, double* dxnew) {       \
        guard_abort([&] { do_step(gp, x, y, dx, xnew, yhat, dxnew); });                                                  \
    }                                                                                                                    \
    void PFX##_step2(moihgp_gp* gp, double* x, double* y, double* dx, double* xnew, double* dxnew) {                     \
        guard_abort([&] { do_step(gp, x, y, dx, xnew, nullptr, dxnew); });                                               \
    }                                                                                                                    \
    void PFX##_step3(moihgp_gp* gp, double* x, double* y, double* xnew, double* yhat) {                                  \
        guard_abort([&] { do_step(gp, x, y, nullptr, xnew, yhat, nullptr); });                                           \
    }                                                                                                                    \
    void PFX##_step4(moihgp_gp* gp, double* x, double* xnew, double* yhat) {                                             \
        guard_abort([&] { do_step(gp, x, nullptr, nullptr, xnew, yhat, nullptr); });                                     \
    }                                                                                                                    \
    void PFX##_update(moihgp_gp* gp, double* params) { guard_abort([&] { do_update(gp, params); }); }                    \
    double PFX##_lik1(moihgp_gp* gp, double* x, double* y, double* dx, double* grad) { return guard_abort([&] { return do_lik(gp, x, y, dx, grad); }); } \
    double PFX##_lik2(moihgp_gp* gp, double* x, double* y) { return guard_abort([&] { return do_lik(gp, x, y, nullptr, nullptr); }); } \
    void PFX##_get_params(moihgp_gp* gp, double* params) { guard_abort([&] { do_get_params(gp, params); }); }            \
    size_t PFX##_igp_dim(moihgp_gp* gp) { return (size_t)gp->d; }                                                        \
    size_t PFX##_num_param(moihgp_gp* gp) { return gp->num_param; }                                                      \
    size_t PFX##_num_igp_param(moihgp_gp* gp) { return (size_t)gp->P; }

MOIHGP_DEFINE_REFERENCE_ABI(gp32, MOIHGP_MATERN32)
MOIHGP_DEFINE_REFERENCE_ABI(gp52, gp52_kernel())

const char* moihgp_last_error(void) { return g_last_error; }

int moihgp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int moihgp_version(void) { return 100; }

moihgp_gp* moihgp_new(int kernel, double dt, size_t num_output, size_t num_latent) {
    return guard_new([&] { return gp_create(kernel, dt, num_output, num_latent, false, nullptr); });
}
void moihgp_del(moihgp_gp* gp) { gp_free(gp); }
size_t moihgp_num_output(moihgp_gp* gp) { return gp->M; }
size_t moihgp_num_latent(moihgp_gp* gp) { return gp->L; }
void moihgp_set_threading(moihgp_gp* gp, int threading) { if (gp) gp->threading = (gp->L < 2) ? false : (threading != 0); }   // moihgp.h:128-135
int moihgp_get_threading(moihgp_gp* gp) { return gp && gp->threading ? 1 : 0; }
int moihgp_polar_iterations(moihgp_gp* gp) { return gp ? gp->polar_its : -1; }

void moihgp_reseed_U(moihgp_gp* gp, unsigned long long seed) {           // (void like the constructor it re-runs: a device failure aborts)
    if (!gp || gp->latents_only) return;
    guard_abort([&] {
        draw_U(gp, seed, true);
        upload_mixing(gp);
        MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream));
    });
}

moihgp_gp* moihgp_new_latents(int kernel, double dt, size_t nl, const double* params_LP) {
    return guard_new([&] { return gp_create(kernel, dt, 0, nl, true, params_LP); });
}

static int set_mixing_impl(moihgp_gp* gp, const double* U, const double* S, double sigma) {
    if (!gp || gp->latents_only || !U || !S) { set_last_error("set_mixing: needs a full MOIHGP object and non-null U, S"); return 1; }
    std::memcpy(gp->U.data(), U, sizeof(double) * gp->M * gp->L);
    std::memcpy(gp->S.data(), S, sizeof(double) * gp->L);
    gp->sigma = sigma;
    gp->U_host_stale = false;
    upload_mixing(gp);
    {   // are the columns orthonormal (a column slice of a polar factor is)?  moihgp_project_stream's least-squares path for partially
        // observed ticks relies on U^T U = I; measured on the device: Gram matrix (MFMA), then max |G - I|
        const size_t L = gp->L;
        if (!gp->dwork) gp->dwork = dev_alloc<double>(L * L + L);
        if (int rc = launch_gram(gp->dU, gp->M, L, gp->dwork, gp->stream)) return rc;
        launch_ortho_defect(gp->dwork, L, gp->dloss, gp->stream);
        double defect = 0.0;
        MOIHGP_HIP_FATAL(hipMemcpyAsync(&defect, gp->dloss, sizeof(double), hipMemcpyDeviceToHost, gp->stream));
        MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream));
        gp->mix_ortho = defect < 1e-9;
    }
    return 0;
}

int moihgp_set_mixing(moihgp_gp* gp, const double* U, const double* S, double sigma) {
    return guard_rc([&] { return set_mixing_impl(gp, U, S, sigma); });
}

static int update_latents_impl(moihgp_gp* gp, const double* params_LP) {
    if (!gp || !params_LP) { set_last_error("update_latents: null argument"); return 1; }
    for (size_t i = 0; i < gp->L * (size_t)gp->P; i++) gp->igp[i] = params_LP[i];
    run_ihgp_update(gp);
    return 0;
}

int moihgp_update_latents(moihgp_gp* gp, const double* params_LP) {
    return guard_rc([&] { return update_latents_impl(gp, params_LP); });
}

static int get_latent_impl(moihgp_gp* gp, size_t l, double* A, double* K, double* S, double* HA, double* AKHA, double* dA,
                      double* dS, double* dK, double* dAKHA, double* HdA, int* iters) {
    if (!gp || l >= gp->L) { set_last_error("get_latent: bad latent index"); return 1; }
    const int d = gp->d, P = gp->P;
    if (kernel_stack(gp->kernel)) {
        if (dA || dS || dK || dAKHA || HdA || iters) {
            ensure_sensitivities(gp);
            const int xd = xd_size(d, P);
            std::vector<double> bd(xd);
            MOIHGP_HIP_FATAL(hipMemcpy(bd.data(), gp->cbd64 + l * xd, sizeof(double) * xd, hipMemcpyDeviceToHost));
            // offsets of XD<D, P>: DAKHA, DK, DA, HDA, DS, ITERS in this order
            const int oDK = P * d * d, oDA = oDK + P * d, oHDA = oDA + P * d * d, oDS = oHDA + P * d, oIT = oDS + P;
            auto cpd = [&](double* dst, int off, int n) { if (dst) std::memcpy(dst, bd.data() + off, sizeof(double) * n); };
            cpd(dAKHA, 0, P * d * d); cpd(dK, oDK, P * d); cpd(dA, oDA, P * d * d); cpd(HdA, oHDA, P * d); cpd(dS, oDS, P);
            if (iters) for (int p = 0; p < P; p++) iters[1 + p] = (int)bd[oIT + p];
        }
        const int xs = xc_size(d);
        std::vector<double> bx(xs);
        MOIHGP_HIP_FATAL(hipMemcpy(bx.data(), gp->cb64 + l * xs, sizeof(double) * xs, hipMemcpyDeviceToHost));
        // the offsets of XC<D> depend on D only through d: AKHA, K, A, HA, S, LOGS, ITERS in this order
        const int oK = d * d, oA = oK + d, oHA = oA + d * d, oS = oHA + d, oIT = oS + 2;
        auto cp = [&](double* dst, int off, int n) { if (dst) std::memcpy(dst, bx.data() + off, sizeof(double) * n); };
        cp(AKHA, 0, d * d); cp(K, oK, d); cp(A, oA, d * d); cp(HA, oHA, d); cp(S, oS, 1);
        if (iters) iters[0] = (int)bx[oIT];
        return 0;
    }
    const int cbs = cb_size(d);
    std::vector<double> b(cbs);
    MOIHGP_HIP_FATAL(hipMemcpy(b.data(), gp->cb64 + l * cbs, sizeof(double) * cbs, hipMemcpyDeviceToHost));
    auto copy = [&](double* dst, int off, int n) { if (dst) std::memcpy(dst, b.data() + off, sizeof(double) * n); };
    if (d == 2) {
        using Ly = CB<2>;
        copy(A, Ly::A, 4); copy(K, Ly::K, 2); copy(S, Ly::S, 1); copy(HA, Ly::HA, 2); copy(AKHA, Ly::AKHA, 4);
        copy(dA, Ly::DA, P * 4); copy(dS, Ly::DS, P); copy(dK, Ly::DK, P * 2); copy(dAKHA, Ly::DAKHA, P * 4); copy(HdA, Ly::HDA, P * 2);
        if (iters) for (int i = 0; i < 1 + P; i++) iters[i] = (int)b[Ly::ITERS + i];
    } else {
        using Ly = CB<3>;
        copy(A, Ly::A, 9); copy(K, Ly::K, 3); copy(S, Ly::S, 1); copy(HA, Ly::HA, 3); copy(AKHA, Ly::AKHA, 9);
        copy(dA, Ly::DA, P * 9); copy(dS, Ly::DS, P); copy(dK, Ly::DK, P * 3); copy(dAKHA, Ly::DAKHA, P * 9); copy(HdA, Ly::HDA, P * 3);
        if (iters) for (int i = 0; i < 1 + P; i++) iters[i] = (int)b[Ly::ITERS + i];
    }
    return 0;
}

int moihgp_get_latent(moihgp_gp* gp, size_t l, double* A, double* K, double* S, double* HA, double* AKHA, double* dA,
                      double* dS, double* dK, double* dAKHA, double* HdA, int* iters) {
    return guard_rc([&] { return get_latent_impl(gp, l, A, K, S, HA, AKHA, dA, dS, dK, dAKHA, HdA, iters); });
}

static int check_stream_args(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld, const void* x) {
    if (!gp) { set_last_error("null handle"); return 1; }
    if (dtype != MOIHGP_F64 && dtype != MOIHGP_F32) { set_last_error("dtype must be MOIHGP_F64 or MOIHGP_F32"); return 1; }
    const size_t es = dtype == MOIHGP_F64 ? 8 : 4, epv = 16 / es;
    if (!x || (T > 0 && !Ty)) { set_last_error("null stream/state pointer"); return 1; }
    if (((uintptr_t)Ty & 15) != 0) { set_last_error("stream base must be 16-byte aligned"); return 1; }
    if (ld % epv != 0 || ld < (T + epv - 1) / epv * epv) { set_last_error("ld (%zu) must be a multiple of %zu and >= T rounded up to it", ld, epv); return 1; }
    return 0;
}

int moihgp_filter_stream(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld, void* x, void* yhat, double* nll, void* stream) {
    return moihgp_filter_stream_io(gp, dtype, Ty, T, ld, x, x, yhat, nll, nullptr, stream);
}

// Which sweeps of the reference's own models go through the stacked filter's kernels when nobody says (option filter_plain_x = -1).
static bool plain_x_by_default(int d, int dtype, size_t L, size_t T) {
    (void)d; (void)dtype; (void)L; (void)T;
    return false;
}

static int filter_stream_io_impl(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld, const void* x_in, void* x, void* yhat, double* nll,
                            double* nll_total, void* stream, size_t ld_out = 0) {
    if (int rc = check_stream_args(gp, dtype, Ty, T, ld, x)) return rc;
    if (!x_in) { set_last_error("null start state"); return 1; }
    note_user_stream(gp, (hipStream_t)stream);
    if (nll_total && !nll) { set_last_error("nll_total needs the per-latent nll buffer"); return 1; }
    if (nll_total && T == 0) MOIHGP_HIP_FATAL(hipMemsetAsync(nll_total, 0, sizeof(double), (hipStream_t)stream));
    if (yhat && ((uintptr_t)yhat & 15) != 0) { set_last_error("yhat base must be 16-byte aligned"); return 1; }
    if (ld_out == 0) ld_out = ld;
    {
        const size_t epv = dtype == MOIHGP_F64 ? 2 : 4;
        if (yhat && (ld_out % epv != 0 || ld_out < (T + epv - 1) / epv * epv)) {
            set_last_error("ld_out (%zu) must be a multiple of %zu and >= T rounded up to it", ld_out, epv);
            return 1;
        }
    }
    const int variant = gp->opt_filter_variant;              // tuning probes: only a -DMOIHGP_TUNING build accepts a non-zero value
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!gp->prof_ev.empty() && 2 * (size_t)(gp->prof_n + 1) <= gp->prof_ev.size() && (gp->prof_seen++ % gp->prof_stride) == 0) {
        e0 = gp->prof_ev[2 * gp->prof_n];
        e1 = gp->prof_ev[2 * gp->prof_n + 1];
        gp->prof_n++;
    }
    // the reference's own models through the stacked filter's kernels (one component; tables from launch_xc_from_cb)
    const bool plain_x = !kernel_stack(gp->kernel) && gp->dxc64 && variant == 0 && gp->opt_filter_split == 0 &&
                         (gp->opt_filter_plain_x == 1 || (gp->opt_filter_plain_x == -1 && plain_x_by_default(gp->d, dtype, gp->L, T)));
    if (kernel_stack(gp->kernel) || plain_x) {
        const int kid = plain_x ? (gp->kernel | (1 << 4)) : gp->kernel;
        const double* xb64 = plain_x ? gp->dxc64 : gp->cb64;
        const float* xb32 = plain_x ? gp->dxc32 : gp->cb32;
        const size_t slen = gp->L < 1024 ? gp->L * 16 : 0;              // per-slice NLL partials of the time split (few latents only)
        if (slen && !gp->dxscratch) gp->dxscratch = dev_alloc<double>(slen);
        if (gp->L >= 1024 && !gp->dlink) {                                               // hand-over records and flags of the second (broken-link) pass
            gp->dlink = dev_alloc<double>(gp->L * 144);
            gp->dlinkflags = dev_alloc<int>(gp->L);
            MOIHGP_HIP_FATAL(hipMemsetAsync(gp->dlinkflags, 0, gp->L * sizeof(int), (hipStream_t)stream));
        }
        // fp32 streams, many latents: the latents whose fp32 scan tables are unusable while the fp64 ones are fine (a mildly unstable filter; listed at
        // update(), skipped by the many-latent kernel) are swept in fp64 on the handle's own stream, beside the sweep -- tick by tick in the fp32
        // kernel one of them holds the whole launch (4096 x 10^4 Matern32x2: 7 such latents, 302 us against 80)
        const size_t n_res = (!plain_x && dtype == MOIHGP_F32 && gp->L >= 1024 && T > 0) ? (size_t)gp->n_unstable[2] : 0;
        if (n_res) {
            const size_t xcs = (size_t)xc_size(gp->d), dd = (size_t)gp->d;
            const size_t tpe = team_powers_elems(gp->d);
            const size_t need = (2 * n_res * ld + 2 * n_res * dd + n_res + n_res * 16) * sizeof(double);
            if (gp->rescue_cap < need) {
                if (gp->drescue) { MOIHGP_HIP_FATAL(hipDeviceSynchronize()); MOIHGP_HIP_FATAL(hipFree(gp->drescue)); gp->drescue = nullptr; gp->rescue_cap = 0; }
                MOIHGP_HIP_FATAL(hipMalloc(&gp->drescue, need));
                gp->rescue_cap = need;
            }
            double* rin = static_cast<double*>(gp->drescue);
            double* rout = rin + n_res * ld;
            double* xi = rout + n_res * ld;
            double* xo = xi + n_res * dd;
            double* nc = xo + n_res * dd;
            double* sc = nc + n_res;
            double* cbc = static_cast<double*>(gp->drescue_const);        // (run_ihgp_update)
            double* tpd = cbc + n_res * xcs;                              // the few-latents team kernel's scan powers: segments side by side, a fifth of the latency
            float* tpf = reinterpret_cast<float*>(tpd + n_res * tpe);
            if (const char* tr = getenv("MOIHGP_GAP_TRACE"); tr && tr[0] == '1') fprintf(stderr, "moihgp side sweep: %zu latents of the fp32 bank in fp64\n", n_res);
            wait_for_caller(gp, (hipStream_t)stream);                    // (behind the caller's queue so far: the stream and the start states are there)
            launch_rescue_gather(static_cast<const float*>(Ty), T, ld, gp->drescue_idx, n_res, nullptr, (int)xcs, static_cast<const float*>(x_in), gp->d, rin, nullptr, xi, gp->stream);
            if (int rc = launch_filter_stream_x(kid, MOIHGP_F64, rin, T, ld, n_res, cbc, nullptr, xi, xo, yhat ? rout : nullptr, nll ? nc : nullptr, gp->stream, nullptr, nullptr,
                                                sc, n_res * 16, 0, ld, nullptr, nullptr, nullptr, gp->opt_filter_maxlinks, -1, tpd, tpf)) return rc;
            launch_rescue_scatter(gp->drescue_idx, n_res, rout, T, ld, xo, gp->d, nc, static_cast<float*>(yhat), ld_out, static_cast<float*>(x), nll, gp->stream);
        }
        auto rescued = [&]() {                                          // the caller's stream waits for the side sweep; the total over all latents
            if (!n_res) return;
            caller_waits(gp, (hipStream_t)stream);
            if (nll && nll_total) launch_nll_total(nll, gp->L, nll_total, (hipStream_t)stream);
        };
        // many latents, a state too wide for per-chunk maps: latents whose stream holds missing ticks are swept by imputation (recursion_x.hip:
        // filter_x_gaps_a / _b_kernel) between the first pass, which hands them over, and the second, which takes what the imputation could not.
        // Left to itself: at d >= 8 always (the second pass alone is 4-6 x slower there), below it -- where the second pass scans the chunks' own
        // maps -- only for a bank without slow filters, whose latents would take both (measured, tools/filternan.py, 4096 x 10^4 at 1 % missing:
        // 2 x Matern-5/2 0.81 -> 0.45 ms; 2 x Matern-3/2 at the bench's draw, 13 % of them slow, 0.37 -> 0.63)
        const bool impute = !plain_x && gp->L >= 1024 && T > 0 && gp->opt_filter_split == 0 && (yhat || nll) &&
                            (gp->opt_filter_impute == 1 || (gp->opt_filter_impute == -1 && (gp->d >= 8 || gp->n_unstable[3] == 0)));
        // (its scratch is 12 or 20 bytes per tick and latent -- the lists of gaps, sized for the worst case: beyond 16 GB (MOIHGP_GAP_BANK_GB) the stream
        // is too long for one call's worth of it and the second pass alone takes the gaps, as before round 4; slabs of a long stream stay below)
        static const size_t gap_bank_limit = []() { const char* e = getenv("MOIHGP_GAP_BANK_GB"); const double gb = e ? atof(e) : 16.0; return (size_t)(gb * 1073741824.0); }();
        if (impute && gap_bank_bytes(gp->d, dtype, gp->L, T) <= gap_bank_limit) {
            const size_t need = gap_bank_bytes(gp->d, dtype, gp->L, T);
            if (gp->gap_cap < need) {
                if (gp->dgap) { MOIHGP_HIP_FATAL(hipDeviceSynchronize()); MOIHGP_HIP_FATAL(hipFree(gp->dgap)); gp->dgap = nullptr; gp->gap_cap = 0; }
                void* p = nullptr;
                MOIHGP_HIP_FATAL(hipMalloc(&p, need));
                gp->dgap = p; gp->gap_cap = need; gp->gap_sig = 0;
            }
            const GapBank bank = gap_bank_carve(gp->dgap, gp->d, dtype, gp->L, T);
            const unsigned long long sig = (unsigned long long)(dtype + 1);
            if (gp->gap_sig != sig) {                                    // a fresh bank, or another scalar type: unit impulses and the zero state again
                if (int rc = gap_bank_init(bank, gp->d, dtype, gp->L, (hipStream_t)stream)) return rc;
                gp->gap_sig = sig; gp->gap_imp_version = 0;
            }
            if (gp->gap_imp_version != gp->cb_version) {                 // the filters' impulse responses: once per parameter update
                if (int rc = launch_gap_impulse(bank, kid, dtype, gp->L, xb64, xb32, (hipStream_t)stream)) return rc;
                gp->gap_imp_version = gp->cb_version;
            }
            if (int rc = launch_filter_stream_x(kid, dtype, Ty, T, ld, gp->L, xb64, xb32, x_in, x, yhat, nll, (hipStream_t)stream, e0, e1, gp->dxscratch, slen, -2, ld_out,
                                                gp->dlinkflags, gp->dlink, nullptr, gp->opt_filter_maxlinks, gp->opt_filter_team, gp->dtp64, gp->dtp32)) return rc;
            const GapArgs ga{bank.imp_out, bank.gpos, bank.gval, bank.gw, bank.gcap, bank.gstat};
            const char* trace = getenv("MOIHGP_GAP_TRACE");               // diagnostics (the tests read it): synchronises the stream
            const bool tracing = trace && trace[0] == '1';
            if (tracing) MOIHGP_HIP_FATAL(hipMemsetAsync(bank.gstat, 0xFF, gp->L * sizeof(int), (hipStream_t)stream));
            if (int rc = launch_filter_stream_x(kid, dtype, Ty, T, ld, gp->L, xb64, xb32, x_in, x, yhat, nll, (hipStream_t)stream, nullptr, nullptr,
                                                reinterpret_cast<double*>(const_cast<GapArgs*>(&ga)), 0, -7, ld_out, gp->dlinkflags, gp->dlink, nullptr, -1, 0, nullptr, nullptr)) return rc;
            // what the recursion could not take (a filter with a memory longer than its table): the second pass, as without imputation
            if (int rc = launch_filter_stream_x(kid, dtype, Ty, T, ld, gp->L, xb64, xb32, x_in, x, yhat, nll, (hipStream_t)stream, nullptr, nullptr, gp->dxscratch, slen, -3, ld_out,
                                                gp->dlinkflags, gp->dlink, nullptr, gp->opt_filter_maxlinks, gp->opt_filter_team, gp->dtp64, gp->dtp32)) return rc;
            if (n_res) rescued();
            else if (nll && nll_total) launch_nll_total(nll, gp->L, nll_total, (hipStream_t)stream);
            if (tracing) {
                std::vector<int> st(gp->L);
                MOIHGP_HIP_FATAL(hipMemcpyAsync(st.data(), bank.gstat, gp->L * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
                MOIHGP_HIP_FATAL(hipStreamSynchronize((hipStream_t)stream));
                long taken = 0, solved = 0, gaps = 0, states = 0, why[6] = {0, 0, 0, 0, 0, 0};
                for (int v : st) if (v >= 0) {
                    taken++;
                    if (v & 1) { solved++; gaps += (v & 0x3FFFFFFF) >> 1; states += (v >> 30) & 1; } else why[(v >> 1) < 6 ? (v >> 1) : 0]++;
                }
                fprintf(stderr, "moihgp gap imputation: %ld latents handed over, %ld solved, %ld gaps filled (%ld of the latents by the state form: memory beyond the table or "
                        "the ring; not solved: %ld response not finite or growing, %ld fill value not finite, %ld stream too long)\n", taken, solved, gaps, states, why[1], why[5], why[4]);
            }
            return 0;
        }
        int rc = launch_filter_stream_x(kid, dtype, Ty, T, ld, gp->L, xb64, xb32, x_in, x, yhat, nll, (hipStream_t)stream, e0, e1,
                                        gp->dxscratch, slen, gp->opt_filter_split /* test hook: 1 = off, n = slices */, ld_out, gp->L >= 1024 ? gp->dlinkflags : nullptr, gp->dlink,
                                        (nll && !n_res) ? nll_total : nullptr, gp->opt_filter_maxlinks, gp->opt_filter_team, gp->dtp64, gp->dtp32);
        if (rc == 0) rescued();
        return rc;
    }
    // few latents, streams of 2 .. 8 segments: one workgroup per latent, eight wavefronts (the stacked filter's team kernel, recursion_x.hip)
    // (left to itself only for Matern-5/2: at d = 2 recursion.hip's own split is 5-8 % faster on streams without gaps -- 8.8 against 9.6 us at
    // 256 x 10^4 fp64 -- and 25 % slower on streams with them; at d = 3 the team kernel wins both, 10.0 against 12.1-12.9 us and 21-23 against 25-34)
    if (gp->dxc64 && gp->opt_filter_plain_x != 0 && gp->opt_filter_team != 0 && (gp->d == 3 || gp->opt_filter_team == 1) && gp->opt_filter_split == 0 && variant == 0) {
        const int rc = launch_filter_teamc_plain(gp->d, dtype, Ty, T, ld, gp->L, gp->dxc64, gp->dxc32, gp->dtp64, gp->dtp32, x_in, x, yhat, nll, (hipStream_t)stream, e0, e1,
                                                 ld_out, nll ? nll_total : nullptr, gp->opt_filter_team);
        if (rc != -1) return rc;
    }
    // time split across the wavefronts of a workgroup when there are too few latents to fill the chip
    int nsplit = 1, nbig = 1; size_t Tslice = T;
    filter_split_plan(dtype, T, gp->L, &nsplit, &Tslice, &nbig);
    if (gp->opt_filter_split != 0) {                                    // test / tuning hook: force the slice count (1 = off)
        int n = gp->opt_filter_split;
        const size_t seg = 64 * (size_t)(dtype == 0 ? kChunk64 : kChunk32);
        if (n <= 1 || T == 0) { nsplit = 1; Tslice = T; nbig = 1; }
        else {
            if (n > 8) n = 8;
            size_t per = ((T + seg - 1) / seg + n - 1) / n; if (per < 1) per = 1;
            Tslice = per * seg; nsplit = (int)((T + Tslice - 1) / Tslice); nbig = nsplit;      // forced count: equal slices
        }
    }
    return launch_filter_stream(gp->d, dtype, Ty, T, ld, gp->L, gp->cb64, gp->cb32, x_in, x, yhat, nll, (hipStream_t)stream, variant, e0, e1,
                                nsplit, Tslice, gp->n_unstable[dtype == MOIHGP_F64 ? 0 : 1], nll_total, nbig, ld_out);
}

// segment-major streams (include/moihgp.h: moihgp_filter_stream_tiled)
static int filter_stream_tiled_impl(moihgp_gp* gp, int dtype, const void* Ty, size_t T, const void* x_in, void* x, void* yhat, double* nll, double* nll_total,
                                    void* stream) {
    if (!gp) { set_last_error("null handle"); return 1; }
    if (dtype != MOIHGP_F64 && dtype != MOIHGP_F32) { set_last_error("dtype must be MOIHGP_F64 or MOIHGP_F32"); return 1; }
    if (!x || !x_in || (T > 0 && !Ty)) { set_last_error("null stream/state pointer"); return 1; }
    if (((uintptr_t)Ty & 15) != 0 || (yhat && ((uintptr_t)yhat & 15) != 0)) { set_last_error("stream base must be 16-byte aligned"); return 1; }
    if (kernel_stack(gp->kernel)) { set_last_error("segment-major streams: the reference's own models only (d = 2, 3); stacked models take series-major streams"); return 3; }
    if (nll_total && !nll) { set_last_error("nll_total needs the per-latent nll buffer"); return 1; }
    note_user_stream(gp, (hipStream_t)stream);
    if (nll_total && T == 0) MOIHGP_HIP_FATAL(hipMemsetAsync(nll_total, 0, sizeof(double), (hipStream_t)stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!gp->prof_ev.empty() && 2 * (size_t)(gp->prof_n + 1) <= gp->prof_ev.size() && (gp->prof_seen++ % gp->prof_stride) == 0) {
        e0 = gp->prof_ev[2 * gp->prof_n];
        e1 = gp->prof_ev[2 * gp->prof_n + 1];
        gp->prof_n++;
    }
    return launch_filter_stream_tiled(gp->d, dtype, Ty, T, gp->L, gp->cb64, gp->cb32, x_in, x, yhat, nll, (hipStream_t)stream, e0, e1,
                                      gp->n_unstable[dtype == MOIHGP_F64 ? 0 : 1], nll_total, gp->opt_filter_variant);
}

int moihgp_filter_stream_tiled(moihgp_gp* gp, int dtype, const void* Ty, size_t T, const void* x_in, void* x, void* yhat, double* nll, double* nll_total,
                               void* stream) {
    return guard_rc([&] { return filter_stream_tiled_impl(gp, dtype, Ty, T, x_in, x, yhat, nll, nll_total, stream); });
}

int moihgp_stream_retile(int dtype, const void* src, void* dst, size_t L, size_t T, size_t ld, int to_tiled, void* stream) {
    return guard_rc([&] {
        if (dtype != MOIHGP_F64 && dtype != MOIHGP_F32) { set_last_error("dtype must be MOIHGP_F64 or MOIHGP_F32"); return 1; }
        if (!src || !dst) { set_last_error("null stream pointer"); return 1; }
        return launch_stream_retile(dtype, src, dst, L, T, ld, to_tiled, (hipStream_t)stream);
    });
}

int moihgp_filter_stream_io(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld, const void* x_in, void* x, void* yhat, double* nll,
                            double* nll_total, void* stream) {
    return guard_rc([&] { return filter_stream_io_impl(gp, dtype, Ty, T, ld, x_in, x, yhat, nll, nll_total, stream); });
}

static int profile_enable_impl(moihgp_gp* gp, int max_launches) {
    if (!gp) { set_last_error("null handle"); return 1; }
    for (hipEvent_t e : gp->prof_ev) (void)hipEventDestroy(e);
    gp->prof_ev.clear();
    gp->prof_n = 0;
    gp->prof_seen = 0;
    for (int i = 0; i < 2 * max_launches; i++) {
        hipEvent_t e;
        MOIHGP_HIP_FATAL(hipEventCreate(&e));
        gp->prof_ev.push_back(e);
    }
    return 0;
}

int moihgp_profile_enable(moihgp_gp* gp, int max_launches) {
    return guard_rc([&] { return profile_enable_impl(gp, max_launches); });
}

static int profile_stride_impl(moihgp_gp* gp, int stride) {
    if (!gp || stride < 1) { set_last_error("profile_stride: null handle or stride < 1"); return 1; }
    gp->prof_stride = (unsigned)stride;
    gp->prof_seen = 0;
    return 0;
}

int moihgp_profile_stride(moihgp_gp* gp, int stride) {
    return guard_rc([&] { return profile_stride_impl(gp, stride); });
}

int moihgp_profile_read(moihgp_gp* gp, float* ms, int n) {
    if (!gp) { set_last_error("null handle"); return -1; }
    int cnt = gp->prof_n < n ? gp->prof_n : n;
    const int rc = guard_rc([&] {
        for (int i = 0; i < cnt; i++) {
            MOIHGP_HIP_FATAL(hipEventSynchronize(gp->prof_ev[2 * i + 1]));
            MOIHGP_HIP_FATAL(hipEventElapsedTime(&ms[i], gp->prof_ev[2 * i], gp->prof_ev[2 * i + 1]));
        }
        return 0;
    });
    gp->prof_n = 0;
    gp->prof_seen = 0;
    return rc ? -1 : cnt;
}

int moihgp_filter_stream_v2(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld_in, const void* x_in, void* x, void* yhat, size_t ld_out,
                            double* nll, double* nll_total, void* stream) {
    return guard_rc([&] { return filter_stream_io_impl(gp, dtype, Ty, T, ld_in, x_in, x, yhat, nll, nll_total, stream, ld_out); });
}

static int grad_stream_impl(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld, void* x, void* dx, void* yhat, double* nll,
                       double* grad, void* stream) {
    if (int rc = check_stream_args(gp, dtype, Ty, T, ld, x)) return rc;
    if (!dx || !grad) { set_last_error("grad_stream: dx and grad are required"); return 1; }
    ensure_sensitivities(gp);
    note_user_stream(gp, (hipStream_t)stream);
    if (kernel_stack(gp->kernel))
        {
        // tables of the time-parallel sweep (grad_scan_x.hip): they change with the hyper-parameters only and every IHGP::update rebuilds them
        // (run_ihgp_update) once they exist; the first gradient sweep of a handle builds them here, on the handle's stream, and waits -- a sweep
        // never builds them in its own stream order (a short sweep would skip the build, a sweep on another stream could read them half-written)
        if (!gp->dhp) gp->dhp = dev_alloc<double>(gp->L * gradx_hp_len(gp->d));
        if (!gp->hp_valid) {
            if (int rc = launch_gp_table_x(gp->kernel, gp->cb64, gp->cbd64, gp->dhp, gp->L, gp->stream)) return rc;
            MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream));
            gp->hp_valid = true;
        }
        return launch_grad_stream_x(gp->kernel, dtype, Ty, T, ld, gp->L, gp->cb64, gp->cbd64, x, dx, yhat, nll, grad, (hipStream_t)stream, 1, gp->dfallback, gp->dhp, 0);
    }
    return launch_grad_stream(gp->d, dtype, Ty, T, ld, gp->L, gp->cb64, gp->cb32, x, dx, yhat, nll, grad, gp->dfallback, (hipStream_t)stream);
}

int moihgp_grad_stream(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld, void* x, void* dx, void* yhat, double* nll,
                       double* grad, void* stream) {
    return guard_rc([&] { return grad_stream_impl(gp, dtype, Ty, T, ld, x, dx, yhat, nll, grad, stream); });
}

// fp32 image of the mixing matrix: allocated and built (on the handle's stream, synchronised) the first time an fp32 stream product
// asks for it, kept current by upload_mixing from then on -- the batched entries only read it
static const float* mixing_f32(moihgp_gp* gp) {
    if (!gp->dU32) gp->dU32 = dev_alloc<float>(gp->M * gp->L);
    if (!gp->u32_valid) {
        launch_narrow(gp->dU, gp->M * gp->L, gp->dU32, gp->stream);
        MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream));
        gp->u32_valid = true;
    }
    return gp->dU32;
}

static int project_stream_impl(moihgp_gp* gp, int dtype, const void* Y, size_t T, void* Ty, size_t ld, void* stream) {
    if (!gp || gp->latents_only) { set_last_error("project_stream needs a full MOIHGP object"); return 1; }
    if (ld < T) { set_last_error("ld < T"); return 1; }
    note_user_stream(gp, (hipStream_t)stream);
    if (int rc = launch_project_stream(dtype, Y, T, gp->M, gp->L, gp->dU, dtype == MOIHGP_F64 ? nullptr : mixing_f32(gp), gp->dinvsqrtS, Ty, ld,
                                       (hipStream_t)stream)) return rc;
    // partially observed ticks: least squares over the observed rows (moihgp.h:167-178), one workgroup per tick behind the GEMM
    // (beyond ls_project_fits the NaN column stands, as include/moihgp.h documents: the recursion treats the tick as missing)
    if (gp->mix_ortho && ls_project_fits(gp->L)) return launch_project_stream_missing(dtype, Y, T, gp->M, gp->L, gp->dU, gp->dinvsqrtS, Ty, ld, (hipStream_t)stream);
    return 0;
}

int moihgp_project_stream(moihgp_gp* gp, int dtype, const void* Y, size_t T, void* Ty, size_t ld, void* stream) {
    return guard_rc([&] { return project_stream_impl(gp, dtype, Y, T, Ty, ld, stream); });
}

static int unproject_stream_impl(moihgp_gp* gp, int dtype, const void* Tyhat, size_t T, size_t ld, void* Yhat, void* stream) {
    if (!gp || gp->latents_only) { set_last_error("unproject_stream needs a full MOIHGP object"); return 1; }
    if (ld < T) { set_last_error("ld < T"); return 1; }
    note_user_stream(gp, (hipStream_t)stream);
    return launch_unproject_stream(dtype, Tyhat, T, ld, gp->M, gp->L, gp->dU, dtype == MOIHGP_F64 ? nullptr : mixing_f32(gp), gp->dsqrtS, Yhat,
                                   (hipStream_t)stream);
}

// Latent shards: the two halves of the least-squares projection of partially observed ticks around the caller's all-reduce (tick.hip)
int moihgp_ls_shard_gram(moihgp_gp* gp, int dtype, const void* Y, const int* ticks, size_t n, int kmax, const void* Ty, size_t ld, double* packed, void* stream) {
    return guard_rc([&] {
        if (!gp || gp->latents_only || (n && (!Y || !ticks || !Ty || !packed))) { set_last_error("ls_shard_gram: needs a full (shard) object and non-null buffers"); return 1; }
        note_user_stream(gp, (hipStream_t)stream);
        return launch_ls_shard(0, dtype, Y, gp->M, gp->L, ticks, n, kmax, gp->dU, gp->dsqrtS, gp->dinvsqrtS, packed, const_cast<void*>(Ty), ld, (hipStream_t)stream);
    });
}
int moihgp_ls_shard_apply(moihgp_gp* gp, int dtype, const void* Y, const int* ticks, size_t n, int kmax, const double* packed, void* Ty, size_t ld, void* stream) {
    return guard_rc([&] {
        if (!gp || gp->latents_only || (n && (!Y || !ticks || !Ty || !packed))) { set_last_error("ls_shard_apply: needs a full (shard) object and non-null buffers"); return 1; }
        note_user_stream(gp, (hipStream_t)stream);
        return launch_ls_shard(1, dtype, Y, gp->M, gp->L, ticks, n, kmax, gp->dU, gp->dsqrtS, gp->dinvsqrtS, const_cast<double*>(packed), Ty, ld, (hipStream_t)stream);
    });
}

int moihgp_unproject_stream(moihgp_gp* gp, int dtype, const void* Tyhat, size_t T, size_t ld, void* Yhat, void* stream) {
    return guard_rc([&] { return unproject_stream_impl(gp, dtype, Tyhat, T, ld, Yhat, stream); });
}

static int window_set_impl(moihgp_gp* gp, const double* Y, size_t W) {
    if (!gp || gp->latents_only) { set_last_error("window_set needs a full MOIHGP object"); return 1; }
    if (!Y || W == 0) { set_last_error("window_set: empty window"); return 1; }
    const size_t M = gp->M, L = gp->L, d = gp->d, P = gp->P, ldw = (W + 1) / 2 * 2;
    // ticks with missing outputs (moihgp.h:462-494): projected by least squares over the observed rows with the stream path's k x k
    // kernel; its limits are checked here, on the host, so that a window it cannot take is refused (rc 3) instead of evaluated wrongly
    std::vector<int> tmiss(W, 0);
    bool any = false;
    for (size_t t = 0; t < W; t++) {
        size_t k = 0;
        for (size_t m = 0; m < M; m++) k += (Y[t * M + m] != Y[t * M + m]) ? 1 : 0;
        if (k == 0) continue;
        any = true; tmiss[t] = 1;
        if (k > (size_t)kLsMaxMissing || M - k < L || !gp->mix_ortho || !ls_project_fits(L)) {
            set_last_error("window_set: tick %zu has %zu of %zu outputs missing (the batched objective takes at most %d per tick, at least %zu observed, "
                           "orthonormal mixing, L <= 15040); use the per-tick ABI", t, k, M, kLsMaxMissing, L);
            return 3;
        }
    }
    const size_t need = 2 * W * M + 3 * L * ldw + W + 2 * L + L * P + L * d + L * P * d + 16;
    order_after_sweeps(gp);
    if (gp->win_cap < need) {
        if (gp->dwin) { MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream)); MOIHGP_HIP_FATAL(hipFree(gp->dwin)); gp->dwin = nullptr; gp->win_cap = 0; gp->win.W = 0; }
        gp->dwin = dev_alloc<double>(need);
        gp->win_cap = need;
    }
    double* p = gp->dwin;
    WindowBufs& w = gp->win;
    w.W = W; w.ldw = ldw;
    w.Y = p; p += W * M;  w.UU = p; p += W * M;
    w.Ty = p; p += L * ldw;  w.hx = p; p += L * ldw;  w.Z = p; p += L * ldw;
    w.rt = p; p += W;  w.spu = p; p += L;  w.nll = p; p += L;  w.gl = p; p += L * P;
    w.x = p; p += L * d;  w.dx = p; p += L * P * d;
    w.tmiss = nullptr;
    gp->win_has_nan = any;
    if (any) {
        if (gp->winmiss_cap < W) {
            if (gp->dwinmiss) { MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream)); MOIHGP_HIP_FATAL(hipFree(gp->dwinmiss)); gp->dwinmiss = nullptr; gp->winmiss_cap = 0; }
            gp->dwinmiss = dev_alloc<int>(W);
            gp->winmiss_cap = W;
        }
        MOIHGP_HIP_FATAL(hipMemcpyAsync(gp->dwinmiss, tmiss.data(), sizeof(int) * W, hipMemcpyHostToDevice, gp->stream));
        w.tmiss = gp->dwinmiss;
    }
    MOIHGP_HIP_FATAL(hipMemcpyAsync(w.Y, Y, sizeof(double) * W * M, hipMemcpyHostToDevice, gp->stream));
    MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream));
    return 0;
}

int moihgp_window_set(moihgp_gp* gp, const double* Y, size_t W) {
    return guard_rc([&] { return window_set_impl(gp, Y, W); });
}

// dev == false: the reference-shaped form (host vectors in and out); dev == true: every pointer is a device pointer (moihgp_window_eval_dev)
// on != nullptr (device form only): *on is the caller's stream -- operands are ordered behind it, results in front of its later work, and the
// host is NOT synchronised (moihgp_window_eval_dev_on)
static int window_eval_impl(moihgp_gp* gp, const double* x, const double* dx, double* loss, double* grad, double* xnew, double* dxnew, bool dev,
                            const hipStream_t* on = nullptr) {
    if (!gp || gp->latents_only || gp->win.W == 0) { set_last_error("window_eval: call moihgp_window_set first"); return 1; }
    if (!x || !dx || !loss || !grad) { set_last_error("window_eval: null argument"); return 1; }
    const size_t L = gp->L, d = gp->d, P = gp->P;
    WindowBufs& w = gp->win;
    order_after_sweeps(gp);        // the sweep below shares the handle's flag / list scratch with sweeps that may still be in flight on caller streams
    if (on) wait_for_caller(gp, *on);
    const hipMemcpyKind in = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, out = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    MOIHGP_HIP_FATAL(hipMemcpyAsync(w.x, x, sizeof(double) * L * d, in, gp->stream));
    MOIHGP_HIP_FATAL(hipMemcpyAsync(w.dx, dx, sizeof(double) * L * P * d, in, gp->stream));
    ensure_sensitivities(gp);
    // device form: the kernels write the loss and the gradient straight into the caller's arrays
    if (int rc = launch_window_objective(gp->tick(), gp->cb64, gp->cb32, w, gp->dfallback, dev ? loss : gp->dloss, dev ? grad : gp->dgrad, gp->stream, gp->kernel)) return rc;
    if (!dev) {
        MOIHGP_HIP_FATAL(hipMemcpyAsync(loss, gp->dloss, sizeof(double), hipMemcpyDeviceToHost, gp->stream));
        MOIHGP_HIP_FATAL(hipMemcpyAsync(grad, gp->dgrad, sizeof(double) * gp->num_param, hipMemcpyDeviceToHost, gp->stream));
    }
    if (xnew) MOIHGP_HIP_FATAL(hipMemcpyAsync(xnew, w.x, sizeof(double) * L * d, out, gp->stream));
    if (dxnew) MOIHGP_HIP_FATAL(hipMemcpyAsync(dxnew, w.dx, sizeof(double) * L * P * d, out, gp->stream));
    if (on) caller_waits(gp, *on);
    else MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream));
    return 0;
}

int moihgp_window_eval(moihgp_gp* gp, const double* x, const double* dx, double* loss, double* grad, double* xnew, double* dxnew) {
    return guard_rc([&] { return window_eval_impl(gp, x, dx, loss, grad, xnew, dxnew, false); });
}

int moihgp_window_eval_dev(moihgp_gp* gp, const double* x_dev, const double* dx_dev, double* loss_dev, double* grad_dev, double* xnew_dev, double* dxnew_dev) {
    return guard_rc([&] { return window_eval_impl(gp, x_dev, dx_dev, loss_dev, grad_dev, xnew_dev, dxnew_dev, true); });
}

int moihgp_update_dev(moihgp_gp* gp, const double* params_dev) {
    return guard_rc([&] {
        if (!gp || gp->latents_only || !params_dev) { set_last_error("update_dev: needs a full MOIHGP object and a device parameter vector"); return 1; }
        do_update(gp, params_dev, true);
        return 0;
    });
}

int moihgp_window_eval_dev_on(moihgp_gp* gp, const double* x_dev, const double* dx_dev, double* loss_dev, double* grad_dev, double* xnew_dev, double* dxnew_dev,
                              void* stream) {
    return guard_rc([&] {
        const hipStream_t s = (hipStream_t)stream;
        return window_eval_impl(gp, x_dev, dx_dev, loss_dev, grad_dev, xnew_dev, dxnew_dev, true, &s);
    });
}

int moihgp_update_dev_on(moihgp_gp* gp, const double* params_dev, void* stream) {
    return guard_rc([&] {
        if (!gp || gp->latents_only || !params_dev) { set_last_error("update_dev: needs a full MOIHGP object and a device parameter vector"); return 1; }
        // update() decides the polar factor's steps on the host and mirrors [S | sigma | per-latent values] there: it returns with the new
        // tables complete, as gpXX_update does.  What the stream argument adds is the ordering of the INPUT behind the caller's queue.
        wait_for_caller(gp, (hipStream_t)stream);
        do_update(gp, params_dev, true);
        return 0;
    });
}

int moihgp_get_params_dev(moihgp_gp* gp, double* params_dev) {
    return guard_rc([&] {
        if (!gp || gp->latents_only || !params_dev) { set_last_error("get_params_dev: needs a full MOIHGP object and a device array"); return 1; }
        const size_t M = gp->M, L = gp->L, sizeU = M * L;
        order_after_sweeps(gp);
        if (gp->U_host_stale) MOIHGP_HIP_FATAL(hipMemcpyAsync(params_dev, gp->dU, sizeof(double) * sizeU, hipMemcpyDeviceToDevice, gp->stream));
        else MOIHGP_HIP_FATAL(hipMemcpyAsync(params_dev, gp->U.data(), sizeof(double) * sizeU, hipMemcpyHostToDevice, gp->stream));
        std::vector<double> tail(L + 1 + L * (size_t)gp->P);
        std::memcpy(tail.data(), gp->S.data(), sizeof(double) * L);
        tail[L] = gp->sigma;
        std::memcpy(tail.data() + L + 1, gp->igp.data(), sizeof(double) * L * gp->P);
        MOIHGP_HIP_FATAL(hipMemcpyAsync(params_dev + sizeU, tail.data(), sizeof(double) * tail.size(), hipMemcpyHostToDevice, gp->stream));
        MOIHGP_HIP_FATAL(hipStreamSynchronize(gp->stream));
        return 0;
    });
}

int moihgp_release_stream(moihgp_gp* gp, void* stream) {
    return guard_rc([&] {
        if (!gp) { set_last_error("null handle"); return 1; }
        hipStream_t s = (hipStream_t)stream;
        for (size_t i = 0; i < gp->user_streams.size(); i++) {
            if (gp->user_streams[i] != s) continue;
            if (!gp->order_ev) MOIHGP_HIP_FATAL(hipEventCreateWithFlags(&gp->order_ev, hipEventDisableTiming));
            MOIHGP_HIP_FATAL(hipEventRecord(gp->order_ev, s));
            MOIHGP_HIP_FATAL(hipStreamWaitEvent(gp->stream, gp->order_ev, 0));
            gp->user_streams.erase(gp->user_streams.begin() + (long)i);
            break;
        }
        return 0;
    });
}

int moihgp_set_option(moihgp_gp* gp, const char* name, long value) {
    if (!gp || !name) { set_last_error("set_option: null argument"); return 1; }
    const std::string n(name);
    if (n == "filter_split") { if (value < 0 || value > 64) { set_last_error("filter_split: 0 (automatic), 1 (off) or a slice count"); return 1; } gp->opt_filter_split = (int)value; return 0; }
    if (n == "filter_impute") { if (value < -1 || value > 1) { set_last_error("filter_impute: -1 (automatic: state dimension >= 8), 0 (never), 1 (always, stacked models)"); return 1; } gp->opt_filter_impute = (int)value; return 0; }
    if (n == "polar_warm_start") { if (value < 0 || value > 1) { set_last_error("polar_warm_start: 0 or 1"); return 1; } gp->opt_polar_warm = (int)value; gp->polar_warm = 0; return 0; }
    if (n == "filter_plain_x") { if (value < -1 || value > 1) { set_last_error("filter_plain_x: -1 (automatic), 0 (never), 1 (always: the stacked filter's kernels for Matern-3/2 and -5/2)"); return 1; } gp->opt_filter_plain_x = (int)value; return 0; }
    if (n == "filter_team") { if (value < -1 || value > 2) { set_last_error("filter_team: -1 (automatic), 0 (never), 1 (whenever the stream fits), 2 (the 32-tick-chunk form only)"); return 1; } gp->opt_filter_team = (int)value; return 0; }
    if (n == "filter_maxlinks") { if (value < -1 || value > 64) { set_last_error("filter_maxlinks: -1 (automatic) .. 64"); return 1; } gp->opt_filter_maxlinks = (int)value; return 0; }
    if (n == "filter_variant") {
#ifdef MOIHGP_TUNING
        gp->opt_filter_variant = (int)value; return 0;
#else
        if (value == 0) return 0;
        set_last_error("filter_variant: the tuning probes are compiled only with -DMOIHGP_TUNING (make TUNING=1)"); return 1;
#endif
    }
    set_last_error("set_option: unknown option '%s'", name);
    return 1;
}

static int pin_host_buffer_impl(moihgp_gp* gp, void* ptr, size_t bytes) {
    if (!gp || !ptr || bytes == 0) { set_last_error("pin_host_buffer: bad argument"); return 1; }
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); set_last_error("hipHostRegister: %s", hipGetErrorString(e)); return 2; }
    gp->pinned.push_back(ptr);
    return 0;
}

int moihgp_pin_host_buffer(moihgp_gp* gp, void* ptr, size_t bytes) {
    return guard_rc([&] { return pin_host_buffer_impl(gp, ptr, bytes); });
}

int moihgp_stream_sync(void* stream) {
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { set_last_error("stream sync: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // extern "C"
