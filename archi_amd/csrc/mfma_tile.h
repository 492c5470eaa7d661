// mfma_tile.h -- shared device helpers of the MFMA kernels (scan.hip, gemm.hip, gemm_ln.hip, attention.hip):
// LDS-DMA staging from inline asm with hand-counted vmcnt, 32x32x16 bf16/f16 MFMA wrappers.
//
// LDS-DMA (global_load_lds): LDS[M0 + lane*16] <- *g, 16 B per lane. Issued from inline asm so hipcc does not
// serialise it against the ds_reads of the OTHER ring slots (it cannot prove they do not alias and would wait
// vmcnt(0) before every fragment read). Completion is waited for by hand with a COUNTED vmcnt before the step
// barrier. N loads share one statement: M0 walks 1 KiB pieces.
#pragma once
#include <type_traits>

#include "common.h"

namespace ak {
namespace mt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS[M0 + lane*16] <- *g (16 B per lane); N consecutive 1 KiB pieces per statement.
template <int N>
__device__ inline void glds16xN(const char *const (&g)[N], int goff, uint32_t lds_wave_base) {
    static_assert(N == 1 || N == 2 || N == 4 || N == 8, "pieces per wave");
    if constexpr (N == 8) {
        const char *const lo[4] = {g[0], g[1], g[2], g[3]};
        const char *const hi[4] = {g[4], g[5], g[6], g[7]};
        glds16xN<4>(lo, goff, lds_wave_base);
        glds16xN<4>(hi, goff, lds_wave_base + 4096);
    } else if constexpr (N == 1) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(g[0] + goff), "s"(lds_wave_base) : "memory", "m0");
    } else if constexpr (N == 2) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                     :: "v"(g[0] + goff), "v"(g[1] + goff), "s"(lds_wave_base) : "memory", "m0", "scc");
    } else {
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off"
                     :: "v"(g[0] + goff), "v"(g[1] + goff), "v"(g[2] + goff), "v"(g[3] + goff), "s"(lds_wave_base)
                     : "memory", "m0", "scc");
    }
}
__device__ inline void glds4(const void *g, uint32_t lds_wave_base) {   // LDS[M0 + lane*4] <- *g, 4 B per lane
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(lds_wave_base) : "memory", "m0");
}
__device__ inline void keep_live(const f32x16 &v) {   // ablation runs: keeps an accumulator from being optimised away
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#endif
}
template <int N>
__device__ inline void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ inline uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}
__device__ inline f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two floats -> packed bf16, round-to-nearest-even, one instruction (v_cvt_pk_bf16_f32, gfx950)
template <bool IS_BF16>
__device__ inline f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
    if constexpr (IS_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}

}  // namespace mt
}  // namespace ak
