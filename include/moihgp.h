/* moihgp.h -- C ABI of libmoihgp.so (MI355X / gfx950 HIP implementation).
 *
 * Part 1 is the drop-in boundary: the exact 26 symbols the reference exports from
 * moihgp/src/wrapper.cpp and that moihgp/pywrapper.py binds through ctypes
 * (pywrapper.py:28-94).  Signatures, buffer layouts and ownership are the reference's:
 * every pointer is a caller-owned host buffer of C-contiguous doubles, valid for the
 * duration of the call only.
 *     x, xnew   [L][d]          (wrapper.cpp:63,84)
 *     dx, dxnew [L][P][d]       (wrapper.cpp:69,90)
 *     y, yhat   [M]
 *     params, grad [M*L + L + 1 + L*P] = U row-major | S | sigma | (magnitude, lengthscale, noise) x L
 *                               (moihgp.h:93, :431-457, :721-738)
 * All arithmetic behind these symbols runs in HIP kernels on the current device; there is no
 * CPU fallback.  If no usable GPU is present `*_new` returns NULL and moihgp_last_error()
 * says why.  A later HIP failure (a launch, copy or allocation that fails) inside one of the
 * reference entries prints the error and aborts -- they are all void / value returns, the
 * reference ABI has no status channel (wrapper.cpp:31-326).  The additive entries of part 2
 * that return int report it instead: rc 1 = invalid argument, 2 = HIP failure, 3 = unsupported
 * input (a window whose missing outputs exceed the limits of moihgp_project_stream), 4 = host memory; moihgp_last_error() holds
 * the text.  Nothing ever unwinds across the ABI.
 *
 * Part 2 is additive: batched entry points over whole time streams (the per-tick ABI costs one
 * FFI crossing + several launches per tick and cannot amortise them), taking DEVICE pointers.
 *
 * Citations `file:line` are into the reference tree /root/reference/moihgp/.
 */
#ifndef MOIHGP_C_API_H_
#define MOIHGP_C_API_H_

#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ Part 1: reference ABI */
/* Opaque handle; the reference returns `GP32*` / `GP52*` (src/wrapper.cpp:21-22).            */
typedef struct moihgp_gp moihgp_gp;

/* replaces src/wrapper.cpp:31-34  (MOIHGP ctor, include/moihgp/moihgp.h:81-136).  `threading` selects the
 * reference's per-call pthread fan-out (moihgp.h:184-214), which the GPU replaces -- but the flag is OBSERVABLE in
 * the reference and is honoured here: the gradient overload of negLogLikelihood adds the per-latent losses only
 * on its threaded branch (moihgp.h:590); the serial branch (:597-607) computes the per-latent gradients and drops
 * the losses.  So with threading == false (the default of pywrapper.py:12, online_learning.py:12, example.py:37;
 * forced for num_latent < 2 by moihgp.h:128-135) gpXX_lik1 returns 1/2 log(sum S) + 1/2 (M-L)+ log sigma +
 * 1/2 ||(I-UU^T)y|| / sigma alone, with threading == true the same plus sum_l 1/2 (v_l^2/S_l + log S_l) (= gpXX_lik2).
 * The gradient is the same in both.  MOIHGP_LIK1_FULL_LOSS=1 (read at construction) selects the summed form
 * regardless of the flag. */
moihgp_gp* gp32_new(double dt, size_t num_output, size_t num_latent, bool threading);
/* replaces src/wrapper.cpp:37-40 (which runs the destructor but leaks the object; we free it) */
void   gp32_del(moihgp_gp* gp);
/* replaces src/wrapper.cpp:43-94   -> MOIHGP::step overload 1, moihgp.h:148-226 */
void   gp32_step1(moihgp_gp* gp, double* x, double* y, double* dx, double* xnew, double* yhat, double* dxnew);
/* replaces src/wrapper.cpp:97-145  -> MOIHGP::step overload 2, moihgp.h:229-301 */
void   gp32_step2(moihgp_gp* gp, double* x, double* y, double* dx, double* xnew, double* dxnew);
/* replaces src/wrapper.cpp:148-184 -> MOIHGP::step overload 3, moihgp.h:304-378 */
void   gp32_step3(moihgp_gp* gp, double* x, double* y, double* xnew, double* yhat);
/* replaces src/wrapper.cpp:187-220 -> MOIHGP::step overload 4, moihgp.h:381-428 */
void   gp32_step4(moihgp_gp* gp, double* x, double* xnew, double* yhat);
/* replaces src/wrapper.cpp:223-228 -> MOIHGP::update, moihgp.h:431-457 (+ IHGP::update, ihgp.h:117-201) */
void   gp32_update(moihgp_gp* gp, double* params);
/* replaces src/wrapper.cpp:231-268 -> MOIHGP::negLogLikelihood(x,y,dx,grad), moihgp.h:460-611 */
double gp32_lik1(moihgp_gp* gp, double* x, double* y, double* dx, double* grad);
/* replaces src/wrapper.cpp:271-296 -> MOIHGP::negLogLikelihood(x,y), moihgp.h:614-688 */
double gp32_lik2(moihgp_gp* gp, double* x, double* y);
/* replaces src/wrapper.cpp:299-307 -> MOIHGP::getParams, moihgp.h:721-738 */
void   gp32_get_params(moihgp_gp* gp, double* params);
/* replace src/wrapper.cpp:310-325 */
size_t gp32_igp_dim(moihgp_gp* gp);
size_t gp32_num_param(moihgp_gp* gp);
size_t gp32_num_igp_param(moihgp_gp* gp);

/* replaces src/wrapper.cpp:329-624.  NOTE: in the reference `GP52` is a typedef of the
 * Matern-3/2 model (typo alias, wrapper.cpp:22), so gp52_* behave exactly like gp32_*.
 * That behaviour is kept by default.  Set the environment variable MOIHGP_GP52_MATERN52=1
 * (read at gp52_new) to get the real Matern-5/2 model of include/moihgp/matern52ss.h, or
 * use moihgp_new(MOIHGP_MATERN52, ...). */
moihgp_gp* gp52_new(double dt, size_t num_output, size_t num_latent, bool threading);
void   gp52_del(moihgp_gp* gp);
void   gp52_step1(moihgp_gp* gp, double* x, double* y, double* dx, double* xnew, double* yhat, double* dxnew);
void   gp52_step2(moihgp_gp* gp, double* x, double* y, double* dx, double* xnew, double* dxnew);
void   gp52_step3(moihgp_gp* gp, double* x, double* y, double* xnew, double* yhat);
void   gp52_step4(moihgp_gp* gp, double* x, double* xnew, double* yhat);
void   gp52_update(moihgp_gp* gp, double* params);
double gp52_lik1(moihgp_gp* gp, double* x, double* y, double* dx, double* grad);
double gp52_lik2(moihgp_gp* gp, double* x, double* y);
void   gp52_get_params(moihgp_gp* gp, double* params);
size_t gp52_igp_dim(moihgp_gp* gp);
size_t gp52_num_param(moihgp_gp* gp);
size_t gp52_num_igp_param(moihgp_gp* gp);

/* ------------------------------------------------------------------ Part 2: additive ABI */
enum { MOIHGP_MATERN32 = 0, MOIHGP_MATERN52 = 1 };
enum { MOIHGP_F64 = 0, MOIHGP_F32 = 1 };
/* Stacked state: the sum of J Matern components observed through one output (state dim J*2 or J*3 up to 12; BASELINE.json's
 * d = 6 / d = 12 shapes).  The reference ships no such model; it is what its IHGP<StateSpace> template (ihgp.h:17-35)
 * computes for a StateSpace with F, Pinf block-diagonal in the reference's component models and H = [H_1 .. H_J].
 * J in {2, 3, 4}.  Hyper-parameters per latent: [magnitude_1, lengthscale_1, .., magnitude_J, lengthscale_J, noise]
 * (P = 2J + 1).  Accepted everywhere a kernel id is: latent banks (moihgp_new_latents / moihgp_update_latents /
 * moihgp_filter_stream / moihgp_grad_stream / moihgp_get_latent) and full objects (moihgp_new: OILMM mixing, the gpXX_* per-tick
 * entries on the handle, moihgp_project_stream / moihgp_unproject_stream, the window objective), with params / grad laid out as
 * [U row-major | S | sigma | P values per latent] (moihgp.h:93 with num_igp_param = 2J + 1).  The sensitivities of IHGP::update
 * (ihgp.h:136-200) are computed from the first call that needs them on. */
#define MOIHGP_STACK(base, J) ((base) | ((J) << 4))

/* Thread-local message of the last failure ("" if none). */
const char* moihgp_last_error(void);
/* Number of usable HIP devices (0 if none / runtime missing). */
int         moihgp_device_count(void);
/* Library/ABI version: major*10000 + minor*100 + patch. */
int         moihgp_version(void);

/* Same object as gp32_new/gp52_new with an explicit kernel (the StateSpace template argument of
 * moihgp::MOIHGP<SS>, include/moihgp/moihgp.h:76).  Returns NULL on failure. */
moihgp_gp*  moihgp_new(int kernel, double dt, size_t num_output, size_t num_latent);
void        moihgp_del(moihgp_gp* gp);
size_t      moihgp_num_output(moihgp_gp* gp);
size_t      moihgp_num_latent(moihgp_gp* gp);
/* The `threading` constructor argument of moihgp.h:81 for objects made by moihgp_new (which start with it off), with the
 * override of moihgp.h:128-135 (num_latent < 2: always off).  It decides the value gpXX_lik1 / moihgp_window_eval return
 * (see gp32_new above). */
void        moihgp_set_threading(moihgp_gp* gp, int threading);
int         moihgp_get_threading(moihgp_gp* gp);
/* Newton-Schulz steps the last update() / construction took for the polar factor of the mixing (moihgp.h:433-447 computes it by SVD;
 * DESIGN.md 3.6): one symmetric Gram product and one M x L x L product each.  0 for small models (single-workgroup kernel). */
int         moihgp_polar_iterations(moihgp_gp* gp);

/* Deterministic counterpart of the ctor's random U (moihgp.h:103-125 uses std::random_device):
 * reseeds and redraws U = polar(I + N(0,1e-3)) from a fixed 64-bit seed. */
void        moihgp_reseed_U(moihgp_gp* gp, unsigned long long seed);

/* Latent-sharded construction: the object owns latents [l0, l0+nl) only and takes only
 * their (magnitude, lengthscale, noise) triples; no mixing matrix.  Used by one rank per GPU.
 * params_LP is a HOST array [nl][P], P = 3 (2J + 1 for a stacked kernel). */
moihgp_gp*  moihgp_new_latents(int kernel, double dt, size_t num_latent_local, const double* params_LP);
/* Re-run IHGP::update (ihgp.h:117-201) for every owned latent from HOST params [nl][P]. */
int         moihgp_update_latents(moihgp_gp* gp, const double* params_LP);

/* Set the mixing of a full object directly, WITHOUT the polar-factor step of update() (moihgp.h:433-447): U [M][L] row-major,
 * S [L], sigma, all HOST.  For latent shards of a bigger model: rank r builds an object with its L_r latents and hands it the
 * columns [lo_r, hi_r) of the global orthonormal factor, so that moihgp_project_stream / moihgp_unproject_stream compute this
 * rank's part of S^-1/2 U^T y and of U S^1/2 Ty (sharded.py).  Combine with moihgp_update_latents for the per-latent parameters. */
int         moihgp_set_mixing(moihgp_gp* gp, const double* U, const double* S, double sigma);

/* Copy the stationary matrices of latent l (ihgp.h:243-254) to HOST buffers (any may be NULL):
 * A[d*d] K[d] S[1] HA[d] AKHA[d*d] dA[P*d*d] dS[P] dK[P*d] dAKHA[P*d*d] HdA[P*d], row-major;
 * iters[1+P] = DARE iteration count followed by the P DLyap counts (utils/dare.h returns are ignored
 * by the reference; exposed here for diagnosis). */
int         moihgp_get_latent(moihgp_gp* gp, size_t l, double* A, double* K, double* S, double* HA, double* AKHA,
                              double* dA, double* dS, double* dK, double* dAKHA, double* HdA, int* iters);

/* ---- ordering contract of the batched entries ---------------------------------------------------
 * moihgp_filter_stream(_io), moihgp_grad_stream, moihgp_project_stream and moihgp_unproject_stream are ASYNCHRONOUS on the
 * stream the caller passes and read the handle's per-latent tables and mixing.  gpXX_update, moihgp_update_latents,
 * moihgp_set_mixing and moihgp_reseed_U rewrite those; they (a) first make their internal stream wait for everything enqueued so
 * far on every stream that carried batched work of this handle, so a rewrite never overtakes a sweep that is still in flight,
 * and (b) return only when the new tables are complete, so batched work enqueued afterwards on any stream sees them.  Buffers
 * the caller owns (streams, states, nll, grad) are ordered by the caller's own stream discipline as usual.  A handle is not
 * thread-safe (as the reference object, moihgp.h:431-457 mutates shared matrices): one host thread at a time.
 * The sweeps also use small scratch areas that belong to the handle (flags and hand-over records of their second passes, per-slice
 * partial sums): batched work of ONE handle must be ordered among itself -- keep it on one stream, or order the streams with events.
 * Different handles are independent. */

/* Hand a stream back before destroying it.  The handle remembers (by value) every stream that carried batched work since its last
 * table rewrite and records an event on each of them at the next rewrite; a stream destroyed in between would be a dangling handle
 * there.  moihgp_release_stream orders the handle's own stream behind everything enqueued on `stream` so far and forgets it.  Streams
 * that live as long as the handle (torch's pool, the default stream) never need this. */
int moihgp_release_stream(moihgp_gp* gp, void* stream);

/* Options of a handle (tuning / test hooks; defaults come from the environment variables of the same meaning, read ONCE when the
 * handle is created -- nothing below consults the environment per launch):
 *   "filter_split"    0 = automatic time split for few latents, 1 = off, n > 1 = n slices      (env MOIHGP_FILTER_SPLIT)
 *   "filter_team"     few latents: -1 = automatic, 0 = never, 1 = always take a one-workgroup-per-latent kernel when the stream fits one
 *                     (2 .. 8 segments of 1024 .. 2048 ticks), 2 = the 2048-tick form only -- stacked models; the reference's own models have
 *                     no such form and take the chunk-templated kernel under 2 as under 1                        (env MOIHGP_FILTER_TEAM)
 *   "filter_plain_x"  Matern-3/2 and -5/2 through the stacked filter's kernels (one component): -1 = automatic, 0 = never, 1 = always
 *   "filter_maxlinks" -1 = automatic; chunks with a gap per segment that the stacked filter's second pass takes as broken links
 *                                                                                               (env MOIHGP_FILTER_MAXLINKS)
 *   "filter_impute"   stacked models, 1024 latents and more: latents whose stream holds missing ticks are swept twice without gaps around a scalar
 *                     recursion over their gaps (exact imputation, csrc/recursion_x.hip: filter_x_gaps_a / _b_kernel) instead of by the second
 *                     pass's broken links / tick-by-tick walk, which keeps the latents whose filter remembers more than 1024 ticks:
 *                     -1 = automatic (state dimension >= 8; below it if no latent's filter is that slow), 0 = never, 1 = always.  (env MOIHGP_GAP_TRACE=1: one line per sweep on stderr with the
 *                     number of latents taken, solved and not, and why not; synchronises the stream)
 *   "polar_warm_start" 0 (default) / 1: update() starts the deflation of outlying singular values (csrc/polar_deflate.hip) from the subspace the
 *                     previous update() of this handle found -- consecutive objective evaluations of a learner differ by one line-search step,
 *                     and one pass over the Gram matrix is saved.  The factor is the same to rounding (<= 1e-12), but no longer a function of
 *                     the argument alone bit for bit: off unless asked (include/moihgp_cxx/lbfgsb_dev.hpp asks).
 *   "filter_variant"  kernel tiling probes; accepted only by a library built with -DMOIHGP_TUNING (make TUNING=1), rc 1 otherwise:
 *                     the probe instantiations (some of which do no arithmetic) are not part of the shipped library.
 * Returns 0, or 1 for an unknown name / a value out of range. */
int moihgp_set_option(moihgp_gp* gp, const char* name, long value);

/* ---- batched recursion over pre-projected streams (DEVICE pointers) -------------------------
 * The stream is SERIES-MAJOR: Ty[l*ld + t], l < L (latents owned by gp), t < T, element type per
 * `dtype`.  Requirements: base pointers 16-byte aligned; ld*sizeof(elem) a multiple of 16 and
 * ld >= T rounded up to a multiple of 16/sizeof(elem).
 *
 * For every latent, sequentially in t (order of moihgp_online.h:61-70 / moihgp_regression.h:45-49):
 *     v = y_t - HA x            (pre-step state; ihgp.h:206)
 *     nll += 1/2 (v^2/S + log S)                       (ihgp.h:207)
 *     x <- AKHA x + K y_t ;  yhat_t = x[0]             (ihgp.h:90-91)
 * A NaN y_t takes the missing-data branch x <- A x (ihgp.h:83-87) and contributes no NLL term.
 *
 *   x        [L][d]  in: state before tick 0; out: state after tick T-1   (dtype)
 *   yhat     [L][ld] or NULL                                              (dtype)
 *   nll      [L] doubles or NULL: per-latent sum of ihgp.h:207 terms (always fp64)
 *   stream   hipStream_t (NULL = default stream).  The call is asynchronous.
 * Returns 0 on success, nonzero on invalid arguments (see moihgp_last_error()).            */
int moihgp_filter_stream(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld,
                         void* x, void* yhat, double* nll, void* stream);
/* The same sweep with the start state read from x_in and the end state written to x (x_in == x is the form above).  A caller
 * that sweeps again and again from one fixed state -- the learners restart every objective evaluation from the same x
 * (moihgp_regression.h:38, moihgp_online.h:57) -- keeps that state in a buffer of its own and saves a reset per sweep.
 * nll_total (DEVICE double, may be NULL; needs nll): the sum over the latents of nll[], the scalar the optimiser consumes
 * (moihgp.h:684), added in a fixed order by a one-wavefront kernel queued right behind the sweep. */
int moihgp_filter_stream_io(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld,
                            const void* x_in, void* x, void* yhat, double* nll, double* nll_total, void* stream);

/* The same sweep with a row stride of its own for yhat (ld_out, same rules as ld: a multiple of 16 bytes, >= T rounded up), so that the
 * filtered means can land in a buffer shaped differently from the stream -- a compact [L][T'] array next to a column slice of a wider
 * slab, say.  The two entries above hand yhat the stream's stride, which the library cannot check against the caller's allocation: an
 * output buffer narrower than the stream's rows is then written out of bounds.  Prefer this entry whenever Ty and yhat are not
 * allocated alike.  yhat may be NULL (ld_out is then ignored). */
int moihgp_filter_stream_v2(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld_in,
                            const void* x_in, void* x, void* yhat, size_t ld_out, double* nll, double* nll_total, void* stream);

/* ---- segment-major streams (round 4) -------------------------------------------------------------------------------------------------
 * The same sweep over streams laid out [ceil(T / SEG)][L][SEG], SEG = 4096 / sizeof(scalar) ticks (1024 fp32, 512 fp64): segment s of every
 * latent side by side, tile (s, l) at ((s L + l) SEG) scalars from the base, the last tile allocated whole (its ticks past T are ignored on
 * input and unspecified on output).  The wavefronts of a launch move through the segments together, so the chip reads and writes one
 * contiguous front instead of L row streams: a cold stream moves ~12 % faster.  For the reference's own models (d = 2, 3);
 * stacked models return 3.  x_in / x / nll / nll_total / stream as moihgp_filter_stream_io; yhat (may be NULL) in the same layout.
 * moihgp_stream_retile copies between the two layouts (to_tiled != 0: src series-major with row stride ld, dst segment-major). */
int moihgp_filter_stream_tiled(moihgp_gp* gp, int dtype, const void* Ty_tiled, size_t T, const void* x_in, void* x, void* yhat_tiled, double* nll,
                               double* nll_total, void* stream);
int moihgp_stream_retile(int dtype, const void* src, void* dst, size_t L, size_t T, size_t ld, int to_tiled, void* stream);

/* As above plus the hyper-parameter sensitivities (ihgp.h:54) and the per-latent NLL gradient
 * (ihgp.h:216-220), summed over ticks:
 *   dx   [L][P][d] in/out (dtype);  grad [L][P] doubles out. */
int moihgp_grad_stream(moihgp_gp* gp, int dtype, const void* Ty, size_t T, size_t ld,
                       void* x, void* dx, void* yhat, double* nll, double* grad, void* stream);

/* ---- OILMM projection over whole streams (DEVICE pointers) ----------------------------------
 * Y is TICK-MAJOR [T][M] (a stream of observation vectors, as the reference's callers hold it,
 * example.py:40-42); the projected stream comes out series-major [L][ld] ready for the recursion:
 *     Ty[l][t] = S_l^-1/2 * sum_m U[m][l] Y[t][m]                 (moihgp.h:181)
 * and back:  Yhat[t][m] = sum_l U[m][l] S_l^1/2 Tyhat[l][t]       (moihgp.h:222-225)
 * A tick whose observation vector holds NaN (missing outputs) is projected as the reference does it, by least squares over the observed
 * rows (moihgp.h:167-178): Ty[:, t] = S^-1/2 (U0^T U0)^-1 U0^T y_obs -- evaluated through the k x k system of the k missing rows
 * (U^T U = I for the polar factor update() installs: U0^T U0 = I - U_miss^T U_miss), the same vector to rounding.  It needs
 * orthonormal columns (checked when the mixing comes from moihgp_set_mixing), at most 64 missing outputs in the tick, at least L
 * observed ones, and L <= 15040 (the tick's U^T y lives in the workgroup's LDS); otherwise the NaN propagates into Ty[:, t], i.e. the
 * recursion treats the whole tick as missing (use the per-tick ABI for such ticks). */
int moihgp_project_stream(moihgp_gp* gp, int dtype, const void* Y, size_t T, void* Ty, size_t ld, void* stream);
/* The same least-squares projection when the latents are split over ranks (one object per rank holding ITS columns of the global orthonormal
 * factor, moihgp_set_mixing): (U0^T U0)^-1 couples all latents, but by Woodbury only sums over the latent columns cross the shards.
 *   1. project the stream with the NaNs replaced by zeros (moihgp_project_stream on the zero-filled Y): Ty = S^-1/2 U_r^T y0, column-local;
 *   2. moihgp_ls_shard_gram:  per affected tick (ticks: DEVICE int32 [n], ascending) this rank's part of [U_miss r | U_miss U_miss^T] into
 *      packed [n][kmax + kmax^2] doubles (zero padded; kmax = the largest number of missing outputs in one tick, <= 64);
 *   3. the caller all-reduces (SUM) `packed` over the ranks -- the path's one extra exchange;
 *   4. moihgp_ls_shard_apply: solves the k x k systems (identical on every rank) and corrects this rank's rows of Ty in place.
 * Y is the ORIGINAL stream (with its NaNs: they name the missing outputs). */
int moihgp_ls_shard_gram(moihgp_gp* gp, int dtype, const void* Y, const int* ticks, size_t n, int kmax, const void* Ty, size_t ld, double* packed, void* stream);
int moihgp_ls_shard_apply(moihgp_gp* gp, int dtype, const void* Y, const int* ticks, size_t n, int kmax, const double* packed, void* Ty, size_t ld, void* stream);
int moihgp_unproject_stream(moihgp_gp* gp, int dtype, const void* Tyhat, size_t T, size_t ld, void* Yhat, void* stream);

/* ---- the learners' windowed objective as one call (HOST pointers, fp64) -----------------------------
 * One evaluation of the loop of moihgp_online.h:61-70 / moihgp_regression.h:42-50 / online_learning.py:83-89:
 *     for t < W:  step(x, y_t, dx, xnew, dxnew);  loss += negLogLikelihood(x, y_t, dx, g);  grad += g;  x = xnew;  dx = dxnew
 * with the object's current parameters (set them with gpXX_update first, moihgp_online.h:43).  `loss` is the sum of what
 * gpXX_lik1 returns per tick, i.e. it follows the object's `threading` flag (moihgp.h:590 vs :597-607, see gp32_new).
 * moihgp_window_set uploads the window Y [W][M] (tick-major, already de-meaned by the caller) once; it stays
 * resident for all evaluations of one L-BFGS solve.
 * Ticks with missing outputs (NaN in y_t) are projected by least squares over the observed rows, as moihgp_project_stream does it
 * (moihgp.h:485-494, same k x k Woodbury form and the same limits: at most 64 missing outputs per tick, at least L observed, orthonormal
 * mixing, L <= 15040; moihgp_window_set returns 3 beyond them: use the per-tick ABI for such windows).  Everything else of such a tick is
 * the reference's own arithmetic on a vector that holds NaN (moihgp.h:499-563: (I - U U^T) y, y^T U dU^T y, pv with raw y(l) are dense
 * products with y): the loss and the U / S / sigma part of the gradient come out NaN, the per-latent part (which only sees the projected
 * stream, moihgp.h:565-607) and the carried state are finite -- exactly what gpXX_lik1 / gpXX_step1 return tick by tick.
 * moihgp_window_eval:  x [L][d], dx [L][P][d] state before the window;  *loss, grad [M*L+L+1+L*P] summed over the
 * W ticks;  xnew / dxnew (may be NULL) state after the window. */
int moihgp_window_set(moihgp_gp* gp, const double* Y, size_t W);
int moihgp_window_eval(moihgp_gp* gp, const double* x, const double* dx, double* loss, double* grad, double* xnew, double* dxnew);

/* ---- the same objective without PCIe in the optimiser's inner loop (DEVICE pointers, fp64) ------------------------------------
 * gpXX_update and moihgp_window_eval move the parameter and gradient vectors (8 (M L + ..) bytes each: 134 MB at M = L = 4096) across
 * PCIe on every objective evaluation.  An optimiser that keeps theta, its gradient and its correction pairs on the device calls these
 * instead: params / grad are DEVICE arrays laid out exactly as the host ones ([U row-major | S | sigma | P values per latent],
 * moihgp.h:93); x, dx, xnew, dxnew, loss are DEVICE arrays / a DEVICE scalar.  moihgp_update_dev == gpXX_update (moihgp.h:431-457; the
 * small tail S | sigma | per-latent values is mirrored to the host for getParams, the mixing is not copied until somebody asks);
 * moihgp_window_eval_dev == moihgp_window_eval on the window moihgp_window_set installed.  Both run on the handle's own stream and
 * return when the results are complete (like their host forms), so the caller's streams need no extra ordering AFTER the call.  The
 * input arrays must be complete when the call is made: synchronise (or otherwise finish) the stream that produced them first -- the
 * handle's stream does not wait for caller streams it has never seen.
 * Same values as the host forms, bit for bit (same kernels). */
int moihgp_update_dev(moihgp_gp* gp, const double* params_dev);
int moihgp_window_eval_dev(moihgp_gp* gp, const double* x_dev, const double* dx_dev, double* loss_dev, double* grad_dev,
                           double* xnew_dev, double* dxnew_dev);
/* The same two entries for callers that produce the operands on a stream of their own (`stream`: a hipStream_t; NULL = the default
 * stream).  The handle's stream is ordered BEHIND everything queued on `stream` at the time of the call (an event, no host
 * synchronisation): operands written by kernels or copies still in flight there are fine.
 *   moihgp_window_eval_dev_on does not synchronise the host at all: `stream` is made to wait for the results, so whatever the caller
 *     queues on it afterwards (an axpy of the gradient, the download of the loss) sees them; other streams need their own ordering.
 *   moihgp_update_dev_on returns, like gpXX_update, when the new tables are complete (the polar factor's step count is decided on the
 *     host, and the tail S | sigma | per-latent values is mirrored there), so only the ordering of its INPUT changes.
 * Values bit-identical to the forms above (tests/test_gpu_configs.py::test_dev_entries_take_the_callers_stream). */
int moihgp_update_dev_on(moihgp_gp* gp, const double* params_dev, void* stream);
int moihgp_window_eval_dev_on(moihgp_gp* gp, const double* x_dev, const double* dx_dev, double* loss_dev, double* grad_dev,
                              double* xnew_dev, double* dxnew_dev, void* stream);
/* getParams (moihgp.h:721-738) into a DEVICE array [num_param]. */
int moihgp_get_params_dev(moihgp_gp* gp, double* params_dev);

/* ---- device vectors: what a bound-constrained L-BFGS needs to keep theta, the gradient and its correction pairs on the device ---------
 * (include/moihgp_cxx/lbfgsb_dev.hpp is such an optimiser; with moihgp_update_dev / moihgp_window_eval_dev one iteration of the learners'
 * loop, moihgp_online.h:40-72 + LBFGSB.h:117-241, then moves no parameter-sized vector across PCIe).  fp64, DEVICE pointers; a context owns
 * a stream, the scratch of the reductions and a page-locked result word.  Element-wise entries are asynchronous on the context's stream;
 * the reductions (dot, proj_step, proj_grad_norm) are deterministic (fixed grid, fixed order) and return their scalar after
 * synchronising it.  mask (unsigned char [n], may be NULL): entries whose byte is 0 are skipped (the free-variable set of the active-set
 * method).  Return codes as for the other additive entries. */
typedef struct moihgp_dvec_ctx moihgp_dvec_ctx;
moihgp_dvec_ctx* moihgp_dvec_ctx_new(void);
void    moihgp_dvec_ctx_del(moihgp_dvec_ctx* c);
void*   moihgp_dvec_ctx_stream(moihgp_dvec_ctx* c);                   /* the context's hipStream_t (for the `_on` entries above) */
double* moihgp_dvec_alloc(size_t n);                                  /* n doubles of device memory (NULL on failure) */
unsigned char* moihgp_dvec_alloc_mask(size_t n);
void    moihgp_dvec_free(void* p);                                    /* as hipFree for the caller; blocks of 1 MB and more are kept for the next alloc of that size */
void    moihgp_dvec_trim(void);                                       /* hand the kept blocks back to the driver */
void    moihgp_dvec_cache_limit(size_t bytes);                        /* most the cache may hold (default 8 GB, or MOIHGP_DVEC_CACHE_GB; 0: keep nothing); trims if over */
int moihgp_dvec_upload(moihgp_dvec_ctx* c, double* dst_dev, const double* src_host, size_t n);     /* synchronous */
int moihgp_dvec_download(moihgp_dvec_ctx* c, double* dst_host, const double* src_dev, size_t n);   /* synchronous */
int moihgp_dvec_copy(moihgp_dvec_ctx* c, double* dst_dev, const double* src_dev, size_t n);
int moihgp_dvec_sync(moihgp_dvec_ctx* c);
int moihgp_dvec_dot(moihgp_dvec_ctx* c, size_t n, const double* a, const double* b, const unsigned char* mask, double* result);
int moihgp_dvec_axpy(moihgp_dvec_ctx* c, size_t n, double alpha, const double* x, double* y, const unsigned char* mask);     /* y += alpha x */
int moihgp_dvec_scale(moihgp_dvec_ctx* c, size_t n, double alpha, const double* x, double* y, const unsigned char* mask);    /* y = alpha x (0 where masked out) */
int moihgp_dvec_sub(moihgp_dvec_ctx* c, size_t n, const double* a, const double* b, double* out);                           /* out = a - b */
int moihgp_dvec_clamp(moihgp_dvec_ctx* c, size_t n, double* x, const double* lb, const double* ub);
/* free_var[i] = 0 where x sits at a bound with the gradient pointing out of the box, or lb == ub; 1 elsewhere */
int moihgp_dvec_active_set(moihgp_dvec_ctx* c, size_t n, const double* x, const double* g, const double* lb, const double* ub, unsigned char* free_var);
/* xt = clamp(xp + step drt, lb, ub);  *dec = sum gradp (xt - xp)   (projected search point and its Armijo decrease) */
int moihgp_dvec_proj_step(moihgp_dvec_ctx* c, size_t n, const double* xp, const double* drt, double step, const double* lb, const double* ub, const double* gradp,
                          double* xt, double* dec);
/* max |clamp(x - g, lb, ub) - x|   (projected-gradient norm, LBFGSB.h:64-67) */
int moihgp_dvec_proj_grad_norm(moihgp_dvec_ctx* c, size_t n, const double* x, const double* g, const double* lb, const double* ub, double* result);

/* Kernel-exact timing of moihgp_filter_stream launches (bench / diagnosis).  After
 * moihgp_profile_enable(gp, n) the next n launches on this handle are bracketed by a HIP event pair
 * attached to the dispatch itself (hipExtLaunchKernel), not to the stream.  The pair brackets the sweep's FIRST (dominant) kernel --
 * the one rocprofv3 lists as filter_scan_kernel / filter_x_kernel; the small kernels queued behind it (the NLL total, the slice sums,
 * and for streams with missing ticks the second pass of the stacked filter) are outside it: time such streams with the wall clock.  moihgp_profile_read waits
 * for them, writes the per-launch durations in milliseconds, rearms the slots and returns the count.
 * moihgp_profile_enable(gp, 0) turns it off.  An event pair makes the dispatch it brackets wait for its predecessor and costs
 * 3-6 us of launch overlap; moihgp_profile_stride(gp, k) attaches pairs to every k-th launch only (default 1), so a timed loop
 * can sample its kernel durations without slowing every pass. */
int moihgp_profile_enable(moihgp_gp* gp, int max_launches);
int moihgp_profile_stride(moihgp_gp* gp, int stride);
int moihgp_profile_read(moihgp_gp* gp, float* ms, int n);

/* Page-lock a caller-owned HOST buffer that is passed again and again to the reference ABI (the params / grad staging arrays
 * of pywrapper.py:146-149 are 8*(M*L+..) bytes: 134 MB at M = L = 4096), so that the copies behind gpXX_update / gpXX_lik1 /
 * moihgp_window_eval run at PCIe rate instead of the pageable rate.  The buffer must outlive the handle (it is unregistered
 * in *_del).  Optional: everything works without it. */
int moihgp_pin_host_buffer(moihgp_gp* gp, void* ptr, size_t bytes);

/* Stream synchronisation helper for callers without a HIP runtime binding. */
int moihgp_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOIHGP_C_API_H_ */
