// kernels_common.h -- device helpers shared by recursion.hip and grad.hip (wave-per-latent segment kernels).
#pragma once
#include "common.h"

namespace moihgp {
namespace {

template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = float4; };
template <> struct VecOf<double> { using type = double2; };

template <typename T> __device__ inline void unpack(const typename VecOf<T>::type& v, T* out);
template <> __device__ inline void unpack<float>(const float4& v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ inline void unpack<double>(const double2& v, double* o) { o[0] = v.x; o[1] = v.y; }
template <typename T> __device__ inline typename VecOf<T>::type pack(const T* in);
template <> __device__ inline float4 pack<float>(const float* i) { return make_float4(i[0], i[1], i[2], i[3]); }
template <> __device__ inline double2 pack<double>(const double* i) { return make_double2(i[0], i[1]); }

// streaming (non-temporal) 16-byte accesses
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef double nt_d2 __attribute__((ext_vector_type(2)));
__device__ inline float4 nt_load(const float4* p) { nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ inline double2 nt_load(const double2* p) { nt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(p)); return make_double2(v.x, v.y); }
__device__ inline void nt_store(float4 v, float4* p) { nt_f4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p)); }
__device__ inline void nt_store(double2 v, double2* p) { nt_d2 t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<nt_d2*>(p)); }

__device__ inline void wave_lds_fence() {
    // LDS operations of one wave execute in program order; this only stops the compiler from
    // moving LDS accesses across the hand-over between lanes of the same wave.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int D>
__device__ inline void matvec_acc(const T* m, const T* v, T* out /* out = m v + out */) {
#pragma unroll
    for (int i = 0; i < D; i++) {
        T s = out[i];
#pragma unroll
        for (int j = 0; j < D; j++) s = fma(m[i * D + j], v[j], s);
        out[i] = s;
    }
}

template <typename T, int D>
__device__ inline void matmul(const T* a, const T* b, T* c) {
    T t[D * D];
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = 0; j < D; j++) {
            T s = 0;
#pragma unroll
            for (int k = 0; k < D; k++) s = fma(a[i * D + k], b[k * D + j], s);
            t[i * D + j] = s;
        }
#pragma unroll
    for (int i = 0; i < D * D; i++) c[i] = t[i];
}

constexpr int kWavesPerBlock = 4;

// ---- DPP lane movement (no LDS, no ds_bpermute) --------------------------------------------------
// v_mov_b32_dpp with `old` = 0: lanes that the control leaves without a source (or that row_mask
// excludes) read 0, which is exactly the "no contribution" value of the scan below.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ inline double dpp0(double v) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xF, false);
    unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// same with an explicit fill value `old` for lanes the control leaves without a source
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_fill(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_fill(double old, double v) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v), f = __builtin_bit_cast(unsigned long long, old);
    unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)u, CTRL, ROW_MASK, 0xF, false);
    unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// wave_shr:1 with lane 0 keeping `first`
__device__ inline float wave_shr1(float v, float first) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ inline double wave_shr1(double v, double first) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v), f = __builtin_bit_cast(unsigned long long, first);
    unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)u, 0x138, 0xF, 0xF, false);
    unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(u >> 32), 0x138, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ inline float read_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ inline double read_lane(double v, int l) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
constexpr int DPP_ROW_SHR = 0x110, DPP_ROW_BCAST15 = 0x142;


// Inclusive scan  s_j = M s_{j-1} + z_j  over the 64 lanes of a wave, all in DPP (no LDS, no ds_bpermute):
// four in-row Kogge-Stone levels (row_shr 1,2,4,8 with the uniform powers sp = M^(1,2,4,8)), then three
// row_bcast:15 rounds that hand the finished prefix of row r-1 to row r through the per-lane power
// pj = M^(lane%16 + 1).
template <typename T, int D>
__device__ inline void dpp_scan(T* z, const T* sp, const T* pj) {
    T t[D];
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_SHR + 1, 0xF>(z[i]);
    matvec_acc<T, D>(sp + 0 * D * D, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_SHR + 2, 0xF>(z[i]);
    matvec_acc<T, D>(sp + 1 * D * D, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_SHR + 4, 0xF>(z[i]);
    matvec_acc<T, D>(sp + 2 * D * D, t, z);
#pragma unroll
    for (int i = 0; i < D; i++) t[i] = dpp0<DPP_ROW_SHR + 8, 0xF>(z[i]);
    matvec_acc<T, D>(sp + 3 * D * D, t, z);
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

}  // namespace
}  // namespace moihgp
