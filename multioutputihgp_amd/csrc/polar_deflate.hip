// polar_deflate.hip -- exact deflation of a few outlying singular values ahead of the Newton-Schulz iteration of polar.hip
// (reference moihgp.h:433-447: U <- svdU svdV^T of the mixing parameters, once per objective evaluation).
//
// Why.  Newton-Schulz needs ~log2 steps from the WIDTH of the spectrum, whatever its shape: a single singular value at 4.7 costs nine
// steps (two 4096^3 fp64 GEMMs each) even if the other 4095 sit at 1 to eleven digits.  That is exactly the spectrum the online learner
// hands to update(): its iterate is the previous polar factor plus a few L-BFGS steps along window gradients, which are rank one per
// tick (moihgp.h:538-552), and it is never re-orthonormalised (moihgp_online.h:185) -- after ten ticks at M = L = 4096: sigma_1 = 4.68,
// seven more between 1.003 and 1.2, sixteen beyond 1 +- 1e-8, everything else within 3e-11 of 1 (profiles/r04/learner_spectrum.log).
//
// What.  With G = A^T A = I + E, a block of NB = 32 Ritz pairs (lambda_i, w_i) of E is computed by subspace iteration with
// Rayleigh-Ritz -- tall-skinny products only, E is read once per iteration -- and for the pairs that have CONVERGED the exact factor
//     W = I + sum_i ((1 + lambda_i)^(-1/2) - 1) w_i w_i^T            ( = G^(-1/2) on span{w_i}, identity on its complement )
// is applied, X1 = A W = A + (A w_i) d_i w_i^T: a rank-NB update.  W is a function of G up to the residuals of the pairs, so
// polar(X1) = polar(A) up to  max_i |d_i| ||E w_i - lambda_i w_i||, which is what the acceptance test bounds (64 eps max(1, |lambda|max):
// the rounding floor of the residual itself).  Pairs that have not converged are left alone; the iteration in polar.hip then starts from
// X1 and finishes the job -- for the learner's spectrum in ONE step instead of nine.  Nothing is assumed about the input: if E is not
// dominated by a few directions (captured energy below 90 % of ||E||_F^2 after two iterations) the attempt is abandoned and A untouched.
//
// Kernels (fp64): ts_mm (MFMA 16x16x4, (A - I) B for a 32-column B), ts_gram (B^T C, deterministic two-stage sum), ts_rotate (B Q diag(c)),
// jacobi32 (symmetric 32 x 32 eigenproblem, one workgroup, round-robin parallel Jacobi), colstats, rank_update.
#include "common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace moihgp {
namespace {

constexpr int NB = 32;                 // block size of the subspace iteration
constexpr int GR = 128;                // rows per workgroup of the two-stage column reductions
typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- C[R][NB] = (A - (SUBI ? I : 0)) B;  A: R x K row-major (lda), B: K x NB row-major.  One workgroup per 16 rows; its four waves split K.
template <bool SUBI>
__global__ void __launch_bounds__(256) ts_mm_kernel(const double* __restrict__ A, size_t lda, size_t R, size_t K, const double* __restrict__ B,
                                                    double* __restrict__ C) {
    __shared__ double part[4][16][NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const size_t r0 = (size_t)blockIdx.x * 16;
    const size_t row = r0 + c < R ? r0 + c : R - 1;                    // (rows past the end: computed on a valid row, never stored)
    const size_t kq = ((K + 3) / 4 + 15) / 16 * 16;                      // k range per wave, a multiple of 16
    const size_t kb = (size_t)wave * kq, ke = kb + kq < K ? kb + kq : K;
    double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double* arow = A + row * lda;
    const bool vec = (lda % 4 == 0) && ((uintptr_t)A % 32 == 0);
    size_t k0 = kb;
    if (vec) {
        for (; k0 + 16 <= ke; k0 += 16) {
            const size_t k = k0 + 4 * (size_t)g;
            const double4_t av = *reinterpret_cast<const double4_t*>(arow + k);
            const double* bp = B + k * NB + c;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const double b0 = bp[(size_t)e * NB], b1 = bp[(size_t)e * NB + 16];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[e], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[e], b1, acc1, 0, 0, 0);
            }
        }
    }
    for (; k0 < ke; k0 += 4) {                                           // ragged / unaligned remainder: one MFMA depth at a time, guarded
        const size_t k = k0 + (size_t)g;
        const bool in = k < ke;
        const double a = in ? arow[k] : 0.0;
        const double b0 = in ? B[k * NB + c] : 0.0, b1 = in ? B[k * NB + 16 + c] : 0.0;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {                                        // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 reg
        part[wave][g + 4 * r][c] = acc0[r];
        part[wave][g + 4 * r][16 + c] = acc1[r];
    }
    __syncthreads();
    for (int e = tid; e < 16 * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        const size_t gi = r0 + i;
        if (gi >= R) continue;
        double v = ((part[0][i][j] + part[1][i][j]) + part[2][i][j]) + part[3][i][j];
        if (SUBI) v -= B[gi * NB + j];
        C[gi * NB + j] = v;
    }
}

// ---- S = X^T Y (NB x NB) for X, Y: R x NB.  Stage 1: one workgroup per GR rows -> part[blk][NB*NB]; stage 2 sums the blocks in order.
__global__ void __launch_bounds__(256) ts_gram_kernel(const double* __restrict__ X, const double* __restrict__ Y, size_t R, double* __restrict__ part) {
    __shared__ double xs[GR][NB], ys[GR][NB];
    const int tid = threadIdx.x;
    const size_t r0 = (size_t)blockIdx.x * GR;
    for (int e = tid; e < GR * NB; e += 256) {
        const size_t r = r0 + e / NB;
        xs[e / NB][e % NB] = r < R ? X[r * NB + e % NB] : 0.0;
        ys[e / NB][e % NB] = r < R ? Y[r * NB + e % NB] : 0.0;
    }
    __syncthreads();
    const int i = tid >> 3, j0 = (tid & 7) * 4;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int r = 0; r < GR; r++) {
        const double xv = xs[r][i];
        a0 = fma(xv, ys[r][j0], a0); a1 = fma(xv, ys[r][j0 + 1], a1); a2 = fma(xv, ys[r][j0 + 2], a2); a3 = fma(xv, ys[r][j0 + 3], a3);
    }
    double* p = part + (size_t)blockIdx.x * NB * NB + i * NB + j0;
    p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3;
}
// S[i][j] = sum_blk part (fixed order); sym: S <- (S + S^T) / 2
__global__ void __launch_bounds__(1024) ts_gram_reduce_kernel(const double* __restrict__ part, int nblk, double* __restrict__ S, int sym) {
    __shared__ double t[NB * NB];
    const int e = threadIdx.x;
    double s = 0.0;
    for (int b = 0; b < nblk; b++) s += part[(size_t)b * NB * NB + e];
    t[e] = s;
    __syncthreads();
    const int i = e / NB, j = e % NB;
    S[e] = sym ? 0.5 * (t[e] + t[j * NB + i]) : s;
}

// ---- Yout = Yin Q diag(cs)   (R x NB times NB x NB);  scale_mode 0: none; 1: cs[j] = theta[j] > thr ? theta[j]^(-1/2) : 0 (orthonormalisation,
// thr = 1e-20 max theta); 2: cs[j] = 1 / theta[j] where theta[j] > 0 else 0 (column normalisation, Q ignored = identity when Q == nullptr)
__global__ void __launch_bounds__(256) ts_rotate_kernel(const double* __restrict__ Yin, size_t R, const double* __restrict__ Q, const double* __restrict__ theta,
                                                        int scale_mode, double* __restrict__ Yout) {
    __shared__ double q[NB][NB + 1], yr[8][NB], cs[NB];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 256) q[e / NB][e % NB] = Q ? Q[e] : ((e / NB) == (e % NB) ? 1.0 : 0.0);
    if (tid < NB) {
        double sc = 1.0;
        if (scale_mode == 1) {
            double mx = 0.0;
            for (int j = 0; j < NB; j++) mx = fmax(mx, theta[j]);
            sc = (theta[tid] > 1e-20 * mx && theta[tid] > 0.0) ? 1.0 / sqrt(theta[tid]) : 0.0;
        } else if (scale_mode == 2) sc = theta[tid] > 0.0 ? 1.0 / theta[tid] : 0.0;
        cs[tid] = sc;
    }
    const size_t r = (size_t)blockIdx.x * 8 + (tid >> 5);
    const int j = tid & 31;
    yr[tid >> 5][j] = r < R ? Yin[r * NB + j] : 0.0;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NB; k++) s = fma(yr[tid >> 5][k], q[k][j], s);
    if (r < R) Yout[r * NB + j] = s * cs[j];
}

// ---- eigen-decomposition of a symmetric NB x NB matrix: w[NB] eigenvalues, Z[NB][NB] eigenvectors in its COLUMNS.  One workgroup; parallel
// cyclic Jacobi in round-robin order (NB - 1 rounds of NB / 2 disjoint rotations per sweep), matrix and vectors in LDS.
__global__ void __launch_bounds__(256) jacobi32_kernel(const double* __restrict__ S, double* __restrict__ w, double* __restrict__ Z) {
    __shared__ double a[NB][NB + 1], z[NB][NB + 1], cs[NB / 2][2], red[256];
    __shared__ int pq[NB / 2][2];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 256) { a[e / NB][e % NB] = S[e]; z[e / NB][e % NB] = (e / NB) == (e % NB) ? 1.0 : 0.0; }
    __syncthreads();
    bool last = false;
    for (int sweep = 0; sweep < 16; sweep++) {
        // off-diagonal mass against the diagonal's: converged when it is below rounding
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < NB * NB; e += 256) { const double v = a[e / NB][e % NB]; if ((e / NB) == (e % NB)) dia += v * v; else off += v * v; }
        red[tid] = off; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const double offs = red[0]; __syncthreads();
        red[tid] = dia; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const double dias = red[0]; __syncthreads();
        // converged: the off-diagonal mass is at its rounding floor (~ n^2 eps^2 of the diagonal's).  Jacobi converges quadratically at the end,
        // so once the mass is within 100 x of that floor ONE more sweep lands on it: no sweep is spent confirming what cannot improve
        if (last || !(offs > 1e-30 * dias) || !(offs == offs)) break;    // (NaN input: leave at once, the caller checks the values)
        if (offs <= 1e-26 * dias) last = true;
        for (int rnd = 0; rnd < NB - 1; rnd++) {
            if (tid < NB / 2) {
                // round-robin: position 0 holds index 0, positions 1 .. NB-1 rotate; pair k = (pos k, pos NB-1-k)
                auto at = [&](int pos) { return pos == 0 ? 0 : 1 + (pos - 1 + rnd) % (NB - 1); };
                int p = at(tid), q = at(NB - 1 - tid);
                if (p > q) { const int t = p; p = q; q = t; }
                const double apq = a[p][q], app = a[p][p], aqq = a[q][q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * sqrt(fabs(app * aqq)) ) {
                    const double th = (aqq - app) / (2.0 * apq);
                    const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0); s = t * c;
                }
                pq[tid][0] = p; pq[tid][1] = q; cs[tid][0] = c; cs[tid][1] = s;
            }
            __syncthreads();
            for (int e = tid; e < NB * (NB / 2); e += 256) {              // columns of a and of z:  (p, q) <- (c p - s q, s p + c q)
                const int r = e / (NB / 2), k = e % (NB / 2);
                const int p = pq[k][0], q = pq[k][1];
                const double c = cs[k][0], s = cs[k][1];
                const double ap = a[r][p], aq = a[r][q];
                a[r][p] = c * ap - s * aq; a[r][q] = s * ap + c * aq;
                const double zp = z[r][p], zq = z[r][q];
                z[r][p] = c * zp - s * zq; z[r][q] = s * zp + c * zq;
            }
            __syncthreads();
            for (int e = tid; e < NB * (NB / 2); e += 256) {              // rows of a
                const int j = e / (NB / 2), k = e % (NB / 2);
                const int p = pq[k][0], q = pq[k][1];
                const double c = cs[k][0], s = cs[k][1];
                const double ap = a[p][j], aq = a[q][j];
                a[p][j] = c * ap - s * aq; a[q][j] = s * ap + c * aq;
            }
            __syncthreads();
        }
    }
    for (int e = tid; e < NB * NB; e += 256) Z[e] = z[e / NB][e % NB];
    if (tid < NB) w[tid] = a[tid][tid];
}

// ---- per column i: out[i] = sum_r (YZ[r][i] - lam[i] W[r][i])^2 (squared residual of the Ritz pair), out[NB + i] = sum_r YZ[r][i]^2
__global__ void __launch_bounds__(256) colstats_kernel(const double* __restrict__ YZ, const double* __restrict__ Wm, const double* __restrict__ lam, size_t R,
                                                       double* __restrict__ part /* [nblk][2 NB] */) {
    __shared__ double s1[8][NB], s2[8][NB];
    const int tid = threadIdx.x, j = tid & 31, sub = tid >> 5;
    const size_t r0 = (size_t)blockIdx.x * GR;
    const double l = lam[j];
    double a = 0.0, b = 0.0;
    for (int rr = sub; rr < GR; rr += 8) {
        const size_t r = r0 + rr;
        if (r >= R) break;
        const double y = YZ[r * NB + j], d = y - l * Wm[r * NB + j];
        a = fma(d, d, a); b = fma(y, y, b);
    }
    s1[sub][j] = a; s2[sub][j] = b;
    __syncthreads();
    if (tid < NB) {
        double x = 0.0, y = 0.0;
        for (int q = 0; q < 8; q++) { x += s1[q][tid]; y += s2[q][tid]; }
        part[(size_t)blockIdx.x * 2 * NB + tid] = x; part[(size_t)blockIdx.x * 2 * NB + NB + tid] = y;
    }
}
// out[0..NB) = sqrt(residual^2), out[NB..2NB) = sqrt(norm^2), fixed order over the blocks
__global__ void __launch_bounds__(64) colstats_reduce_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    const int e = threadIdx.x;
    double s = 0.0;
    for (int b = 0; b < nblk; b++) s += part[(size_t)b * 2 * NB + e];
    out[e] = sqrt(s);
}

// ---- X[m][l] += sum_j P[m][j] d[j] W[l][j]      (rank-NB update, in place)
__global__ void __launch_bounds__(256) rank_update_kernel(double* __restrict__ X, size_t M, size_t L, const double* __restrict__ P, const double* __restrict__ d,
                                                          const double* __restrict__ Wm) {
    constexpr int ROWS = 32;
    __shared__ double pd[ROWS][NB];
    const int tid = threadIdx.x;
    const size_t l = (size_t)blockIdx.x * 256 + tid, m0 = (size_t)blockIdx.y * ROWS;
    for (int e = tid; e < ROWS * NB; e += 256) {
        const size_t m = m0 + e / NB;
        pd[e / NB][e % NB] = m < M ? P[m * NB + e % NB] * d[e % NB] : 0.0;
    }
    double wv[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) wv[j] = l < L ? Wm[l * NB + j] : 0.0;
    __syncthreads();
    if (l >= L) return;
    for (int i = 0; i < ROWS; i++) {
        const size_t m = m0 + i;
        if (m >= M) break;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NB; j++) s = fma(pd[i][j], wv[j], s);
        X[m * L + l] += s;
    }
}

// ---- the Gram matrix after the deflation, without forming it again: with W = I + V D V^T (V: the Ritz vectors, orthonormal; D = diag d) and
// G V = V + E V,  V^T G V = I + Lambda (Galerkin),   W G W = G + V D (G V)^T + (G V) D V^T + V D (I + Lambda) D V^T  -- three rank-NB terms, one
// pass over G instead of a 2 M L^2 product.  G_ij += sum_c p1_ic B_jc + p2_ic V_jc,  B = V + E V, p1 = V d, p2 = B d + V d^2 (1 + lambda).
__global__ void __launch_bounds__(256) gram_update_kernel(double* __restrict__ G, size_t L, const double* __restrict__ Wm, const double* __restrict__ YZ,
                                                          const double* __restrict__ d, const double* __restrict__ lam) {
    constexpr int ROWS = 32;
    __shared__ double p1[ROWS][NB], p2[ROWS][NB];
    const int tid = threadIdx.x;
    const size_t j = (size_t)blockIdx.x * 256 + tid, i0 = (size_t)blockIdx.y * ROWS;
    for (int e = tid; e < ROWS * NB; e += 256) {
        const size_t i = i0 + e / NB;
        const int c = e % NB;
        double a = 0.0, b = 0.0;
        if (i < L) {
            const double v = Wm[i * NB + c], bb = v + YZ[i * NB + c], dc = d[c], mc = dc * dc * (1.0 + lam[c]);
            a = v * dc; b = bb * dc + v * mc;
        }
        p1[e / NB][c] = a; p2[e / NB][c] = b;
    }
    double vj[NB], bj[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) { vj[c] = j < L ? Wm[j * NB + c] : 0.0; bj[c] = j < L ? vj[c] + YZ[j * NB + c] : 0.0; }
    __syncthreads();
    if (j >= L) return;
    for (int r = 0; r < ROWS; r++) {
        const size_t i = i0 + r;
        if (i >= L) break;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < NB; c++) s = fma(p1[r][c], bj[c], fma(p2[r][c], vj[c], s));
        G[i * L + j] += s;
    }
}

// ---- deterministic start block: +-1 / sqrt(L) from a hash of (row, column)
__global__ void init_block_kernel(double* __restrict__ V, size_t L) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L * NB) return;
    unsigned long long h = (unsigned long long)e * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; h *= 0x94D049BB133111EBull; h ^= h >> 29;
    V[e] = ((h & 1) ? 1.0 : -1.0) / sqrt((double)L);
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("%s launch: %s", what, hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

size_t polar_deflate_work_doubles(size_t M, size_t L) {
    const size_t nblk = (L + GR - 1) / GR;
    return 4 * L * NB + M * NB + nblk * NB * NB + nblk * 2 * NB + 3 * NB * NB + 6 * NB + 64;
}

// X (M x L, device) with Gram matrix G (L x L, device) = X^T X and frob2 = ||G - I||_F^2.  On success *n_pairs pairs were deflated: when > 0,
// X and G are both updated in place (G <- W G W by three rank-NB terms, not by a new product).  Returns 0, or a HIP / launch error code.
// *warm (in / out, may be NULL): non-zero on entry = `work` still holds the Ritz vectors of the previous successful call for the same (M, L) --
// consecutive objective evaluations of a learner differ by one line-search step, and so do their outlying subspaces: the iteration starts from
// them instead of the random block (one pass fewer).  Set on return according to whether Ritz vectors were left behind.
int polar_deflate(double* X, size_t M, size_t L, double* G, double frob2, double* work, hipStream_t s, int* n_pairs, int trace, int* warm) {
    *n_pairs = 0;
    const bool warm_start = warm && *warm;
    if (warm) *warm = 0;
    if (L < 2 * NB || !(frob2 > 0.0)) return 0;
    const int nblk = (int)((L + GR - 1) / GR);
    double* V = work;                     // L x NB: current orthonormal basis
    double* Y = V + L * NB;               // L x NB: E V
    double* Wm = Y + L * NB;              // L x NB: Ritz vectors V Z
    double* YZ = Wm + L * NB;             // L x NB: E W
    double* P = YZ + L * NB;              // M x NB: A W
    double* gpart = P + M * NB;           // nblk x NB^2
    double* cpart = gpart + (size_t)nblk * NB * NB;   // nblk x 2 NB
    double* T = cpart + (size_t)nblk * 2 * NB;        // NB^2
    double* Z = T + NB * NB;              // NB^2
    double* Q = Z + NB * NB;              // NB^2
    double* lam = Q + NB * NB;            // NB
    double* theta = lam + NB;             // NB
    double* cst = theta + NB;             // 2 NB: residual norms, column norms
    double* dvec = cst + 2 * NB;          // NB
    const unsigned rb16 = (unsigned)((L + 15) / 16), rb8 = (unsigned)((L + 7) / 8);
    auto gram = [&](const double* A_, const double* B_, double* S_, int sym) {
        hipLaunchKernelGGL(ts_gram_kernel, dim3(nblk), dim3(256), 0, s, A_, B_, L, gpart);
        hipLaunchKernelGGL(ts_gram_reduce_kernel, dim3(1), dim3(NB * NB), 0, s, (const double*)gpart, nblk, S_, sym);
    };
    auto orthonormalise = [&](double* B_ /* in / out, via Y as scratch is NOT allowed: uses YZ */, double* tmp) {
        gram(B_, B_, T, 1);
        hipLaunchKernelGGL(jacobi32_kernel, dim3(1), dim3(256), 0, s, (const double*)T, theta, Q);
        hipLaunchKernelGGL(ts_rotate_kernel, dim3(rb8), dim3(256), 0, s, (const double*)B_, L, (const double*)Q, (const double*)theta, 1, tmp);
    };
    if (warm_start) {
        // the previous call's Ritz vectors: orthonormal to rounding (V Z with V orthonormal, Z orthogonal)
        MOIHGP_HIP_FATAL(hipMemcpyAsync(V, Wm, sizeof(double) * L * NB, hipMemcpyDeviceToDevice, s));
    } else {
        // start block, orthonormalised twice (the first pass leaves O(cond eps) behind)
        hipLaunchKernelGGL(init_block_kernel, dim3((unsigned)((L * NB + 255) / 256)), dim3(256), 0, s, Y, L);
        orthonormalise(Y, V);
        orthonormalise(V, Y);
        MOIHGP_HIP_FATAL(hipMemcpyAsync(V, Y, sizeof(double) * L * NB, hipMemcpyDeviceToDevice, s));
    }
    if (int rc = check_launch("polar_deflate (start block)")) return rc;

    double h_lam[NB], h_cst[2 * NB], h_d[NB];
    bool accept[NB];
    int n_ok = 0;
    constexpr int kMaxIter = 6;
    for (int it = 0; it < kMaxIter; it++) {
        hipLaunchKernelGGL((ts_mm_kernel<true>), dim3(rb16), dim3(256), 0, s, (const double*)G, L, L, L, (const double*)V, Y);              // Y = E V
        gram(V, Y, T, 1);                                                                                                    // T = V^T E V
        hipLaunchKernelGGL(jacobi32_kernel, dim3(1), dim3(256), 0, s, (const double*)T, lam, Z);
        hipLaunchKernelGGL(ts_rotate_kernel, dim3(rb8), dim3(256), 0, s, (const double*)V, L, (const double*)Z, (const double*)nullptr, 0, Wm);   // Ritz vectors
        hipLaunchKernelGGL(ts_rotate_kernel, dim3(rb8), dim3(256), 0, s, (const double*)Y, L, (const double*)Z, (const double*)nullptr, 0, YZ);   // E W
        hipLaunchKernelGGL(colstats_kernel, dim3(nblk), dim3(256), 0, s, (const double*)YZ, (const double*)Wm, (const double*)lam, L, cpart);
        hipLaunchKernelGGL(colstats_reduce_kernel, dim3(1), dim3(2 * NB), 0, s, (const double*)cpart, nblk, cst);
        if (int rc = check_launch("polar_deflate (iteration)")) return rc;
        MOIHGP_HIP_FATAL(hipMemcpyAsync(h_lam, lam, sizeof(double) * NB, hipMemcpyDeviceToHost, s));
        MOIHGP_HIP_FATAL(hipMemcpyAsync(h_cst, cst, sizeof(double) * 2 * NB, hipMemcpyDeviceToHost, s));
        MOIHGP_HIP_FATAL(hipStreamSynchronize(s));
        double lmax = 0.0, energy = 0.0, worst = 0.0;
        bool finite = true;
        for (int i = 0; i < NB; i++) { finite = finite && std::isfinite(h_lam[i]) && std::isfinite(h_cst[i]); lmax = fmax(lmax, fabs(h_lam[i])); energy += h_lam[i] * h_lam[i]; }
        if (!finite) return 0;
        const double tolc = 64.0 * 2.220446049250313e-16 * fmax(1.0, lmax);
        n_ok = 0;
        for (int i = 0; i < NB; i++) {
            const double g1 = 1.0 + h_lam[i];
            h_d[i] = g1 > 1e-12 ? 1.0 / sqrt(g1) - 1.0 : 0.0;           // (an eigenvalue of G at 0: rank deficient, leave it to the iteration's own verdict)
            // what accepting the pair does to the polar factor is ~ |d_i| res_i; a residual AT its rounding floor (64 eps |lambda|max) is as good
            // as it gets whatever d_i is (a singular value far below 1 has a large d_i -- and a polar factor that is that ill-conditioned anyway)
            const double crit = fmin(fabs(h_d[i]) * h_cst[i], h_cst[i]);
            accept[i] = g1 > 1e-12 && crit <= tolc && fabs(h_d[i]) > 1e-13;
            if (fabs(h_d[i]) > 1e-13) worst = fmax(worst, crit);
            n_ok += accept[i] ? 1 : 0;
        }
        if (trace) std::fprintf(stderr, "polar: deflation pass %d%s  |lambda|max %.4e  captured %.4f of ||E||_F^2  worst |d| res %.2e (tol %.1e)  converged pairs %d\n",
                                it + 1, warm_start ? " (warm start)" : "", lmax, energy / frob2, worst, tolc, n_ok);
        if ((it >= 1 || warm_start) && energy < 0.9 * frob2) {           // E is not a few directions: nothing to gain here
            if (warm_start && it == 0) {                                  // (or the kept basis belongs to another matrix: once more from the random block)
                int cold = 0;
                const int rc = polar_deflate(X, M, L, G, frob2, work, s, n_pairs, trace, &cold);
                if (warm) *warm = cold;
                return rc;
            }
            return 0;
        }
        if (worst <= tolc) break;                                        // every pair that matters has converged
        if (it + 1 == kMaxIter) break;
        // next basis: the columns of E W are nearly orthogonal with norms |lambda_i|: normalise, then orthonormalise
        hipLaunchKernelGGL(ts_rotate_kernel, dim3(rb8), dim3(256), 0, s, (const double*)YZ, L, (const double*)nullptr, (const double*)(cst + NB), 2, Y);
        orthonormalise(Y, V);
        if (it == 0 && !warm_start) {                                    // (from the second Rayleigh-Ritz on the normalised columns are near-orthonormal: one pass)
            orthonormalise(V, Y);
            MOIHGP_HIP_FATAL(hipMemcpyAsync(V, Y, sizeof(double) * L * NB, hipMemcpyDeviceToDevice, s));
        }
    }
    if (n_ok == 0) return 0;
    for (int i = 0; i < NB; i++) if (!accept[i]) h_d[i] = 0.0;
    MOIHGP_HIP_FATAL(hipMemcpyAsync(dvec, h_d, sizeof(double) * NB, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((ts_mm_kernel<false>), dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s, (const double*)X, L, M, L, (const double*)Wm, P);    // P = A W
    hipLaunchKernelGGL(rank_update_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)((M + 31) / 32)), dim3(256), 0, s, X, M, L, (const double*)P, (const double*)dvec,
                       (const double*)Wm);
    // G <- W G W: the Gram matrix of the updated X (polar.hip continues from it)
    hipLaunchKernelGGL(gram_update_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)((L + 31) / 32)), dim3(256), 0, s, G, L, (const double*)Wm, (const double*)YZ,
                       (const double*)dvec, (const double*)lam);
    if (int rc = check_launch("polar_deflate (update)")) return rc;
    MOIHGP_HIP_FATAL(hipStreamSynchronize(s));                            // (h_d is on this frame's stack)
    *n_pairs = n_ok;
    if (warm) *warm = 1;                                                  // Wm holds this matrix's Ritz vectors for the next call
    return 0;
}

}  // namespace moihgp
