// oilmm.hip -- OILMM projection / un-projection of whole observation streams.
//
//   project:    Ty[l][t]   = S_l^-1/2 * sum_m U[m][l] * Y[t][m]          (moihgp.h:181, per tick there)
//   unproject:  Yhat[t][m] = sum_l U[m][l] * S_l^1/2 * Tyhat[l][t]       (moihgp.h:222-225)
//
// The reference evaluates these per tick as (LxL diag)(LxM)(M) products, O(L^2 M) per tick
// (moihgp.h:181); here they are two GEMMs over the whole stream, written so that the projected
// stream comes out SERIES-MAJOR [L][ld], the layout recursion.hip streams with coalesced loads.
//
// v1: LDS-tiled 64x64x16 GEMM on the vector ALUs (correctness first).  The MFMA form
// (v_mfma_f64_16x16x4_f64 / v_mfma_f32_32x32x2_f32) is the next step for this file; it is priced
// against the MFMA roofline separately from the recursion.
#include "common.h"

namespace moihgp {
namespace {

constexpr int BM = 64, BN = 64, BK = 16;

// C(i,j) = rs[i] * sum_k A(i,k) * ks[k] * B(k,j)
// A(i,k) = A[i*sa_i + k*sa_k]  (tile loads are contiguous along i: sa_i == 1)
// B(k,j) = B[k*sb_k + j*sb_j]  (tile loads are contiguous along k: sb_k == 1)
// C(i,j) = C[i*sc_i + j]       (stores contiguous along j)
template <typename TA, typename TB, typename TC>
__global__ void __launch_bounds__(256)
gemm_tile_kernel(size_t Mi, size_t Nj, size_t Kk, const TA* __restrict__ A, size_t sa_k, const TB* __restrict__ B,
                 size_t sb_j, TC* __restrict__ C, size_t sc_i, const double* __restrict__ rs, int rs_mode,
                 const double* __restrict__ ks, int ks_mode) {
    __shared__ TC As[BK][BM + 4];
    __shared__ TC Bs[BK][BN + 4];
    const int tid = threadIdx.x;
    const size_t i0 = (size_t)blockIdx.y * BM, j0 = (size_t)blockIdx.x * BN;
    const int ti = tid % 16, tj = tid / 16;
    TC acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0;

    for (size_t k0 = 0; k0 < Kk; k0 += BK) {
        // A tile: 64 (i) x 16 (k), threads run along i
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int ii = tid % 64, kk = tid / 64 + 4 * r;
            size_t gi = i0 + ii, gk = k0 + kk;
            TC v = 0;
            if (gi < Mi && gk < Kk) v = (TC)A[gi + gk * sa_k];
            As[kk][ii] = v;
        }
        // B tile: 16 (k) x 64 (j), threads run along k
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int kk = tid % 16, jj = tid / 16 + 16 * r;
            size_t gk = k0 + kk, gj = j0 + jj;
            TC v = 0;
            if (gk < Kk && gj < Nj) {
                v = (TC)B[gk + gj * sb_j];
                if (ks_mode == 1) v *= (TC)sqrt(ks[gk]);
            }
            Bs[kk][jj] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk++) {
            TC a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { a[q] = As[kk][ti * 4 + q]; b[q] = Bs[kk][tj * 4 + q]; }
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[p][q] = fma(a[p], b[q], acc[p][q]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        size_t gi = i0 + ti * 4 + p;
        if (gi >= Mi) continue;
        TC scale = 1;
        if (rs_mode == 1) scale = (TC)(1.0 / sqrt(rs[gi]));
#pragma unroll
        for (int q = 0; q < 4; q++) {
            size_t gj = j0 + tj * 4 + q;
            if (gj < Nj) C[gi * sc_i + gj] = scale * acc[p][q];
        }
    }
}

}  // namespace

int launch_project_stream(int dtype, const void* Y, size_t T, size_t M, size_t L, const double* U, const double* S,
                          void* Ty, size_t ld, hipStream_t s) {
    if (T == 0 || L == 0) return 0;
    // i = l (L), j = t (T), k = m (M):  A(i,k) = U[k*L + i];  B(k,j) = Y[j*M + k];  C = Ty[i*ld + j]
    dim3 grid((unsigned)((T + BN - 1) / BN), (unsigned)((L + BM - 1) / BM)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL((gemm_tile_kernel<double, double, double>), grid, block, 0, s, L, T, M, U, L, (const double*)Y, M,
                           (double*)Ty, ld, S, 1, (const double*)nullptr, 0);
    else
        hipLaunchKernelGGL((gemm_tile_kernel<double, float, float>), grid, block, 0, s, L, T, M, U, L, (const float*)Y, M,
                           (float*)Ty, ld, S, 1, (const double*)nullptr, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("project_stream launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

int launch_unproject_stream(int dtype, const void* Tyhat, size_t T, size_t ld, size_t M, size_t L, const double* U,
                            const double* S, void* Yhat, hipStream_t s) {
    if (T == 0 || M == 0) return 0;
    // i = t (T), j = m (M), k = l (L):  A(i,k) = Tyhat[k*ld + i];  B(k,j) = sqrt(S_k) U[j*L + k];  C = Yhat[i*M + j]
    dim3 grid((unsigned)((M + BN - 1) / BN), (unsigned)((T + BM - 1) / BM)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL((gemm_tile_kernel<double, double, double>), grid, block, 0, s, T, M, L, (const double*)Tyhat, ld, U, L,
                           (double*)Yhat, M, (const double*)nullptr, 0, S, 1);
    else
        hipLaunchKernelGGL((gemm_tile_kernel<float, double, float>), grid, block, 0, s, T, M, L, (const float*)Tyhat, ld, U, L,
                           (float*)Yhat, M, (const double*)nullptr, 0, S, 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("unproject_stream launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

}  // namespace moihgp
