// gemm_mfma.hip -- LDS-staged MFMA GEMM for the dense products around the recursion:
//   * OILMM projection / un-projection of whole streams (reference moihgp.h:181, :222-225, per tick there)
//   * the Gram / update products of the Newton-Schulz polar factor in MOIHGP::update (moihgp.h:433-447)
//
//   C(i,j) = rs(i) * sum_k A(i,k) * ks(k) * B(k,j)                i < Mi, j < Nj, k < Kk
//
// Operand layouts are given by which index is contiguous in memory:
//   A_ICONTIG: A(i,k) = A[k*lda + i]   else  A(i,k) = A[i*lda + k]
//   B_KCONTIG: B(k,j) = B[j*ldb + k]   else  B(k,j) = B[k*ldb + j]
//   C(i,j) = C[i*ldc + j]
// so every product needed here runs without a transpose pass.
//
// Tile: 128 x 128 x 16 per 256-thread workgroup; 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 MFMA tiles of
// 16 x 16 (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32: one element of A and of B per lane per
// instruction).  Both operands are staged k-major in LDS (As[k][i], Bs[k][j], row stride 128 + 16 elements so that
// the two 16-lane halves of a ds_read hit disjoint bank halves), the next tile is fetched into registers while
// the current one is multiplied.  Workgroup ids are remapped so that the 8 tiles sharing an XCD's L2 are
// neighbours in the output.
//
// Roofline: MFMA (fp64 and fp32-input MFMA both run at 64 FLOP/clk/SIMD on gfx950).
#include "common.h"

namespace moihgp {
namespace {

constexpr int BM = 128, BN = 128, LDT = BM + 16;
#ifndef MOIHGP_GEMM_BK
#define MOIHGP_GEMM_BK 32
#endif
constexpr int BK = MOIHGP_GEMM_BK, KH = BK / 16;   // k-depth of a tile; staged as KH slabs of 16

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <typename T> struct Acc;
template <> struct Acc<double> {
    using type = double4_t;
    static __device__ inline type mfma(double a, double b, type c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ inline int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Acc<float> {
    using type = float4_t;
    static __device__ inline type mfma(float a, float b, type c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D map of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
    static __device__ inline int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// 8 contiguous source elements (16-byte aligned) -> 8 compute-type registers
template <typename TS, typename TC> __device__ inline void load8(const TS* p, TC* out);
template <> __device__ inline void load8<double, double>(const double* p, double* o) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int v = 0; v < 4; v++) { double2 t = q[v]; o[2 * v] = t.x; o[2 * v + 1] = t.y; }
}
template <> __device__ inline void load8<double, float>(const double* p, float* o) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int v = 0; v < 4; v++) { double2 t = q[v]; o[2 * v] = (float)t.x; o[2 * v + 1] = (float)t.y; }
}
template <> __device__ inline void load8<float, float>(const float* p, float* o) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int v = 0; v < 2; v++) { float4 t = q[v]; o[4 * v] = t.x; o[4 * v + 1] = t.y; o[4 * v + 2] = t.z; o[4 * v + 3] = t.w; }
}

// SYM: the product is symmetric (Gram matrix X^T X): only tiles on or above the diagonal are computed; mirror_upper_kernel
// fills the lower triangle afterwards with coalesced reads and writes.
template <typename TC, typename TA, typename TB, bool A_ICONTIG, bool B_KCONTIG, bool SYM = false>
__global__ void __launch_bounds__(256)
gemm_mfma_kernel(size_t Mi, size_t Nj, size_t Kk, const TA* __restrict__ A, size_t lda, const TB* __restrict__ B, size_t ldb,
                 TC* __restrict__ C, size_t ldc, const double* __restrict__ rs, int rs_mode, const double* __restrict__ ks, int ks_mode,
                 unsigned tiles_m, unsigned tiles_n) {
    // fp32: rows k with bit 3 set keep their columns XOR 32 -- the k-contiguous operands write rows e and 8 + e from the two threads that share a
    // column, which are 8 x 144 words = a multiple of the 64 banks apart (25 % of the LDS cycles were such conflicts, profiles/r04/gemm_variants.log);
    // reads take 16 consecutive columns of one row and do not notice: unproject 70.6 -> 71.7 % of the fp32 MFMA peak, project unchanged (74.4 %).
    // MOIHGP_GEMM_DB=1 (tried, off): two LDS buffers and one barrier per k-tile, the next tile's stash beside this tile's MFMAs -- 74 KB per
    // workgroup, two workgroups per compute unit instead of four: 71.1 / 67.5 %, the four resident workgroups already hide each other's barriers.
#ifndef MOIHGP_GEMM_DB
#define MOIHGP_GEMM_DB 0
#endif
#ifndef MOIHGP_GEMM_SWZ
#define MOIHGP_GEMM_SWZ 1
#endif
    constexpr int NBUF = (sizeof(TC) == 4 && MOIHGP_GEMM_DB) ? 2 : 1;
    constexpr bool SWZ = sizeof(TC) == 4 && MOIHGP_GEMM_SWZ;           // (fp64 rows are 288 words: its reads are laid out for that, its writes conflict two ways still)
    __shared__ TC As[NBUF][BK][LDT];
    __shared__ TC Bs[NBUF][BK][LDT];
    auto swz = [](int k) { return SWZ ? (k & 8) << 2 : 0; };
    // ---- XCD-aware tile order: workgroups b, b+8, b+16.. share an XCD; give each XCD a contiguous run of tiles ----
    const unsigned nwg = SYM ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    unsigned tm, tn;
    if (SYM) {
        // only tiles on or above the diagonal exist: row tm holds tiles_m - tm of them
        tm = 0;
        unsigned rem = bid;
        while (rem >= tiles_m - tm) { rem -= tiles_m - tm; tm++; }
        tn = tm + rem;
    } else {
        // column-panel order with groups of 8 row tiles (keeps a B panel and 8 A panels hot in L2)
        constexpr unsigned GROUP = 8;
        const unsigned per_group = GROUP * tiles_n;
        const unsigned g = bid / per_group, first_m = g * GROUP;
        const unsigned gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
        tm = first_m + (bid % per_group) % gsz;
        tn = (bid % per_group) / gsz;
    }
    const size_t i0 = (size_t)tm * BM, j0 = (size_t)tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    typename Acc<TC>::type acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[a][b][r] = 0;

    // register staging of one 128 x BK tile of each operand as KH slabs of 16 in k: 8 elements per thread and slab, contiguous
    // in memory (along i / j for the *-contiguous layouts, along k otherwise).  Interior tiles take 16-byte vector loads.
    TC ra[KH][8], rb[KH][8];
    const bool a_vec = ((uintptr_t)A % 16 == 0) && (lda % (16 / sizeof(TA)) == 0);
    const bool b_vec = ((uintptr_t)B % 16 == 0) && (ldb % (16 / sizeof(TB)) == 0);
    const bool i_full = i0 + BM <= Mi, j_full = j0 + BN <= Nj;
    auto fetch = [&](size_t kt) {
        const bool k_full = kt + BK <= Kk;
#pragma unroll
        for (int h = 0; h < KH; h++) {
            const size_t k0 = kt + 16 * h;
            if (A_ICONTIG) {      // rows of As are contiguous in memory: thread -> (k = tid / 16, 8 consecutive i)
                const int kk = tid >> 4, ii = (tid & 15) * 8;
                const size_t gk = k0 + kk;
                if (a_vec && i_full && k_full) load8<TA, TC>(A + gk * lda + i0 + ii, ra[h]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const size_t gi = i0 + ii + e;
                        ra[h][e] = (gk < Kk && gi < Mi) ? (TC)A[gk * lda + gi] : TC(0);
                    }
                }
                // the k scale goes onto whichever operand a thread holds for ONE k (one load, 8 multiplies): here that is A
                if (ks_mode == 1 && gk < Kk) {
                    const TC sc = (TC)ks[gk];
#pragma unroll
                    for (int e = 0; e < 8; e++) ra[h][e] *= sc;
                }
            } else {              // A(i,k) = A[i*lda + k]: thread -> (i = tid / 2, 8 consecutive k)
                const int ii = tid >> 1, kk = (tid & 1) * 8;
                const size_t gi = i0 + ii;
                if (a_vec && i_full && k_full) load8<TA, TC>(A + gi * lda + k0 + kk, ra[h]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const size_t gk = k0 + kk + e;
                        ra[h][e] = (gk < Kk && gi < Mi) ? (TC)A[gi * lda + gk] : TC(0);
                    }
                }
            }
            if (B_KCONTIG) {      // B(k,j) = B[j*ldb + k]: thread -> (j = tid / 2, 8 consecutive k)
                const int jj = tid >> 1, kk = (tid & 1) * 8;
                const size_t gj = j0 + jj;
                if (b_vec && j_full && k_full) load8<TB, TC>(B + gj * ldb + k0 + kk, rb[h]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const size_t gk = k0 + kk + e;
                        rb[h][e] = (gk < Kk && gj < Nj) ? (TC)B[gj * ldb + gk] : TC(0);
                    }
                }
                if (ks_mode == 1 && !A_ICONTIG) {       // both operands k-contiguous: eight scales per thread
#pragma unroll
                    for (int e = 0; e < 8; e++) { const size_t gk = k0 + kk + e; if (gk < Kk) rb[h][e] *= (TC)ks[gk]; }
                }
            } else {              // B(k,j) = B[k*ldb + j]: thread -> (k = tid / 16, 8 consecutive j)
                const int kk = tid >> 4, jj = (tid & 15) * 8;
                const size_t gk = k0 + kk;
                if (b_vec && j_full && k_full) load8<TB, TC>(B + gk * ldb + j0 + jj, rb[h]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const size_t gj = j0 + jj + e;
                        rb[h][e] = (gk < Kk && gj < Nj) ? (TC)B[gk * ldb + gj] : TC(0);
                    }
                }
                if (ks_mode == 1 && !A_ICONTIG && gk < Kk) {
                    const TC sc = (TC)ks[gk];
#pragma unroll
                    for (int e = 0; e < 8; e++) rb[h][e] *= sc;
                }
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int h = 0; h < KH; h++) {
            if (A_ICONTIG) {
                const int kk = 16 * h + (tid >> 4), ii = ((tid & 15) * 8) ^ swz(kk);
#pragma unroll
                for (int e = 0; e < 8; e++) As[buf][kk][ii + e] = ra[h][e];
            } else {
                const int ii = tid >> 1, kk = 16 * h + (tid & 1) * 8;
#pragma unroll
                for (int e = 0; e < 8; e++) As[buf][kk + e][ii ^ swz(kk)] = ra[h][e];       // (kk is a multiple of 8: one swizzle for its eight rows)
            }
            if (B_KCONTIG) {
                const int jj = tid >> 1, kk = 16 * h + (tid & 1) * 8;
#pragma unroll
                for (int e = 0; e < 8; e++) Bs[buf][kk + e][jj ^ swz(kk)] = rb[h][e];
            } else {
                const int kk = 16 * h + (tid >> 4), jj = ((tid & 15) * 8) ^ swz(kk);
#pragma unroll
                for (int e = 0; e < 8; e++) Bs[buf][kk][jj + e] = rb[h][e];
            }
        }
    };
    auto multiply = [&](int buf) {
#pragma unroll
        for (int s = 0; s < BK / 4; s++) {
            TC af[4], bf[4];
            const int kr = s * 4 + (lane >> 4), c = lane & 15;
            const int sw = swz(s * 4);                                 // (compile-time: the lane's part of kr is below bit 3)
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = As[buf][kr][wm * 64 + ((a * 16) ^ sw) + c];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = Bs[buf][kr][wn * 64 + ((b * 16) ^ sw) + c];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = Acc<TC>::mfma(af[a], bf[b], acc[a][b]);
        }
    };

    fetch(0);
    if constexpr (NBUF == 2) {
        stash(0);
        __syncthreads();
        if (BK < Kk) fetch(BK);
        int cur = 0;
        for (size_t k0 = 0; k0 < Kk; k0 += BK) {
            multiply(cur);
            if (k0 + BK < Kk) stash(cur ^ 1);        // the tile fetched during the previous multiply
            __syncthreads();                         // the next tile is complete; this one is fully consumed
            if (k0 + 2 * BK < Kk) fetch(k0 + 2 * BK);
            cur ^= 1;
        }
    } else {
        for (size_t k0 = 0; k0 < Kk; k0 += BK) {
            __syncthreads();                 // previous tile fully consumed
            stash(0);
            __syncthreads();
            if (k0 + BK < Kk) fetch(k0 + BK);   // in flight during the MFMAs below
            multiply(0);
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------------------
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t gi = i0 + wm * 64 + a * 16 + Acc<TC>::row(lane, r);
            if (gi >= Mi) continue;
            TC scale = TC(1);
            if (rs_mode == 1) scale = (TC)rs[gi];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const size_t gj = j0 + wn * 64 + b * 16 + (lane & 15);
                if (gj < Nj) {
                    C[gi * ldc + gj] = scale * acc[a][b][r];
                }
            }
        }
}

// G[j][i] = G[i][j] for the 128 x 128 tiles strictly above the diagonal: 32 x 32 sub-tiles transposed through LDS.
__global__ void __launch_bounds__(256) mirror_upper_kernel(double* __restrict__ G, size_t L, unsigned tiles) {
    __shared__ double t[32][33];
    const unsigned tm = blockIdx.y, tn = blockIdx.x;
    if (tn <= tm) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads
    for (int si = 0; si < 4; si++)
        for (int sj = 0; sj < 4; sj++) {
            const size_t i0 = (size_t)tm * BM + si * 32, j0 = (size_t)tn * BN + sj * 32;
            for (int r = ty; r < 32; r += 8) {
                const size_t i = i0 + r, j = j0 + tx;
                t[r][tx] = (i < L && j < L) ? G[i * L + j] : 0.0;
            }
            __syncthreads();
            for (int r = ty; r < 32; r += 8) {
                const size_t j = j0 + r, i = i0 + tx;
                if (i < L && j < L) G[j * L + i] = t[tx][r];
            }
            __syncthreads();
        }
}

template <typename TC, typename TA, typename TB, bool AI, bool BK_, bool SYM = false>
int launch(size_t Mi, size_t Nj, size_t Kk, const TA* A, size_t lda, const TB* B, size_t ldb, TC* C, size_t ldc, const double* rs,
           int rs_mode, const double* ks, int ks_mode, hipStream_t s) {
    const unsigned tm = (unsigned)((Mi + BM - 1) / BM), tn = (unsigned)((Nj + BN - 1) / BN);
    if (tm == 0 || tn == 0) return 0;
    hipLaunchKernelGGL((gemm_mfma_kernel<TC, TA, TB, AI, BK_, SYM>), dim3(SYM ? tm * (tm + 1) / 2 : tm * tn), dim3(256), 0, s, Mi, Nj, Kk, A, lda, B, ldb, C, ldc, rs,
                       rs_mode, ks, ks_mode, tm, tn);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("gemm_mfma launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace

__global__ void scales_kernel(const double* __restrict__ S, size_t L, double* __restrict__ sqrtS, double* __restrict__ invsqrtS) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const double q = sqrt(S[l]);
    sqrtS[l] = q;
    invsqrtS[l] = 1.0 / q;                      // moihgp.h:163 `1 / sqrt(S(idx))`
}
void launch_scales(const double* S, size_t L, double* sqrtS, double* invsqrtS, hipStream_t s) {
    hipLaunchKernelGGL(scales_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, S, L, sqrtS, invsqrtS);
}
__global__ void narrow_kernel(const double* __restrict__ src, size_t n, float* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
// fp32 image of the mixing matrix for the fp32 stream products: the kernels convert U to the compute type anyway, and at
// 128 x 128 tiles the operand traffic (L2) is what bounds them -- 4-byte U elements cut it by a third
void launch_narrow(const double* src, size_t n, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(narrow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, n, dst);
}
// Ty[l][t] = S_l^-1/2 * sum_m U[m][l] Y[t][m]:  i = l, j = t, k = m;  A = U (i-contiguous, lda = L);  B = Y (k-contiguous, ldb = M)
int launch_project_stream(int dtype, const void* Y, size_t T, size_t M, size_t L, const double* U, const float* U32, const double* S /* invsqrtS */,
                          void* Ty, size_t ld, hipStream_t s) {
    if (dtype == 0) return launch<double, double, double, true, true>(L, T, M, U, L, (const double*)Y, M, (double*)Ty, ld, S, 1, nullptr, 0, s);
    if (U32) return launch<float, float, float, true, true>(L, T, M, U32, L, (const float*)Y, M, (float*)Ty, ld, S, 1, nullptr, 0, s);
    return launch<float, double, float, true, true>(L, T, M, U, L, (const float*)Y, M, (float*)Ty, ld, S, 1, nullptr, 0, s);
}

// Yhat[t][m] = sum_l Tyhat[l][t] * sqrt(S_l) * U[m][l]:  i = t, j = m, k = l;  A = Tyhat (i-contiguous, lda = ld);  B = U (k-contiguous, ldb = L)
int launch_unproject_stream(int dtype, const void* Tyhat, size_t T, size_t ld, size_t M, size_t L, const double* U, const float* U32,
                            const double* S /* sqrtS */, void* Yhat, hipStream_t s) {
    if (dtype == 0) return launch<double, double, double, true, true>(T, M, L, (const double*)Tyhat, ld, U, L, (double*)Yhat, M, nullptr, 0, S, 1, s);
    if (U32) return launch<float, float, float, true, true>(T, M, L, (const float*)Tyhat, ld, U32, L, (float*)Yhat, M, nullptr, 0, S, 1, s);
    return launch<float, float, double, true, true>(T, M, L, (const float*)Tyhat, ld, U, L, (float*)Yhat, M, nullptr, 0, S, 1, s);
}

// gradU[r][c] = sum_t Y[t][r] Z[c][t]:  i = r, j = c, k = t;  A = Y (i-contiguous, lda = M);  B = Z (k-contiguous, ldb = ldz)
int launch_ugrad_gemm(const double* Y, size_t W, size_t M, const double* Z, size_t ldz, size_t L, double* gradU, hipStream_t s) {
    return launch<double, double, double, true, true>(M, L, W, Y, M, Z, ldz, gradU, L, nullptr, 0, nullptr, 0, s);
}

// G[a][b] = sum_m X[m][a] X[m][b]  (X is M x L row-major):  A i-contiguous (lda = L), B j-contiguous (ldb = L)
int launch_gram(const double* X, size_t M, size_t L, double* G, hipStream_t s) {
    if (int rc = launch<double, double, double, true, false, true>(L, L, M, X, L, X, L, G, L, nullptr, 0, nullptr, 0, s)) return rc;   // upper tiles only
    const unsigned tiles = (unsigned)((L + BM - 1) / BM);
    if (tiles > 1) hipLaunchKernelGGL(mirror_upper_kernel, dim3(tiles, tiles), dim3(256), 0, s, G, L, tiles);
    return 0;
}

// C[m][b] = sum_a X[m][a] W[a][b]  (X M x L, W L x L, row-major):  A k-contiguous (lda = L), B j-contiguous (ldb = L)
int launch_matmul_nn(const double* X, size_t M, size_t L, const double* W, double* C, hipStream_t s) {
    return launch<double, double, double, false, false>(M, L, L, X, L, W, L, C, L, nullptr, 0, nullptr, 0, s);
}

}  // namespace moihgp
