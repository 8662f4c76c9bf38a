// x_common.h -- device helpers shared by the stacked-model kernels (recursion_x.hip, grad_scan_x.hip): wave-uniform tables as
// "slabs" read through DPP row broadcasts, scalar-load views of constant blocks, ds_bpermute lane shifts, compile-time loops.
#pragma once
#include "kernels_common.h"
#include <type_traits>
#include <utility>

namespace moihgp {
namespace {

template <typename T> __device__ inline T bperm(int addr, T v);
template <> __device__ inline float bperm<float>(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}
template <> __device__ inline double bperm<double>(int addr, double v) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)u);
    unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// out += m v for a WAVE-UNIFORM matrix m (scalar loads, SGPR operands).  The scalar register file holds ~100 values, so the
// rows are fetched in batches of about 72 dwords; the scheduling barrier keeps the compiler from hoisting every s_load of a
// 12 x 12 product to the top (288 dwords: it would spill SGPRs into VGPR lanes).
// A wave-uniform view of a constant block for use INSIDE loops: the asm hides the pointer's provenance, so the loads cannot
// be hoisted out of the enclosing loop (the blocks are loop-invariant, and LICM would otherwise pull them all in front of the
// segment loop and run out of registers); readfirstlane + the constant address space make uniform accesses through the
// result scalar loads.  The blocks are written by the update kernel only.
template <typename T> using uptr = const __attribute__((address_space(4))) T*;
template <typename T>
__device__ inline uptr<T> launder(const T* p) {
    unsigned long long u = reinterpret_cast<unsigned long long>(p);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return (uptr<T>)(((unsigned long long)hi << 32) | lo);
}

// compile-time loop: f(std::integral_constant<int, 0>) .. f(std::integral_constant<int, N-1>)
template <typename F, int... I>
__device__ inline void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ inline void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// acc += slab[E] * x, slab[E] = entry E (0..15) of a slab register, read as a DPP row broadcast.  Every lane of the wave must
// be active (the kernel runs whole waves), and `slab` is only ever written by loads (no VALU-write -> DPP-read hazard).
template <int E> __device__ inline void fmac_bc(double& acc, const double& slab, const double& x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(slab), "v"(x), "n"(E));
}
template <int E> __device__ inline void fmac_bc(float& acc, const float& slab, const float& x) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(slab), "v"(x), "n"(E));
}

// slabs of a 16-aligned table: register s holds entries [16 s, 16 s + 16), replicated over the four rows of the wave
template <typename T, int NS, typename P>
__device__ inline void load_slabs(P tab, int lane, T (&slab)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; s++) slab[s] = tab[s * 16 + (lane & 15)];
}

// out += M v, M a row-major D x D matrix held in slabs (j outer, i inner: D independent accumulation chains in flight)
template <typename T, int D, int NS>
__device__ inline void matvec_bc(const T (&m)[NS], const T (&v)[D], T (&out)[D]) {
    static_for<D>([&](auto jj) {
        static_for<D>([&](auto ii) {
            constexpr int e = decltype(ii)::value * D + decltype(jj)::value;
            fmac_bc<e % 16>(out[decltype(ii)::value], m[e / 16], v[decltype(jj)::value]);
        });
    });
}

// scan powers of the chunk-templated team kernel (recursion_x.hip), built by stack_dispatch.hip's team_powers_kernel at every update
constexpr int kTeamNck = 5;                                  // chunk lengths the kernel is built for: 16, 20, .., 32 (index (CK - 16) / 4)
template <int D> constexpr int team_powers_len() { return 7 * XC<D>::LS + 16; }       // M^(1,2,..,64) | levels that matter, decays, tame, 0..

}  // namespace
}  // namespace moihgp
