// gelu_table.h -- GELU by LDS table for the bf16 encoder path (ffn.hip: the fused hidden-384 layer kernel; gemm.hip: the
// epilogue of the wide FFN-up tile). Device-side lookup only; the table is built on the host in ffn.hip (gelu_table_host).
#pragma once
#include "common.h"

namespace ak {

typedef float gt_f32x4 __attribute__((ext_vector_type(4)));

// GELU BY TABLE (round 4). The phase-B operand is bf16, so the kernel needs gelu(x) to 8 significant bits only -- but the
// polynomial above is ~19 fp32 VALU operations per value, and VALU work does not hide under the SIMD partner's MFMAs on this
// part (k_ffn384r: 160 packed instructions under a partner's 48 MFMAs took 1.6 k cycles, ~10 per instruction). Here x is
// converted to f16 and rounded to 7 mantissa bits (sign + 5 exponent + 7 mantissa bits = 13 index bits;
// f16's exponent range covers every magnitude that matters: below 2^-14 gelu(x) = x / 2 is < 3e-5), and the index selects one of
// 8192 bf16 entries = the exact erf-GELU of the rounded input (host, double precision) in a 16 KB LDS table:
//   v_cvt_f16_f32 + v_add + v_lshrrev + v_and + ds_read_u16 per value, + half a v_lshl_or to pair two results.
// Error: the input is rounded like a bf16 tensor would round it (relative 2^-9) before an exact GELU -- what a bf16 framework
// computes when the up-projection's output is stored as bf16 -- instead of fp32 in / polynomial (7.8e-6) / bf16 out.
constexpr int GELU_TAB_BYTES = 8192 * 2;
// The table sits at LDS address 0 (the kernel traps if its dynamic LDS does not start there), so the masked bits ARE the address:
// v_cvt_f16_f32 + v_add + v_lshrrev + v_and + ds_read_u16 per value. No clamp: the table covers f16's whole range (for large x the
// entry is the bf16 rounding of x itself), an overflowing input reads the +-2^16 entries, a NaN some entry inside the table.
__device__ inline uint32_t f_gelu_tab1(float x) {
    const _Float16 h = (_Float16)x;
    const uint32_t hb = (uint32_t)__builtin_bit_cast(uint16_t, h);
    const uint32_t addr = ((hb + 4u) >> 2) & 0x3ffeu;              // 2 * (f16 bits rounded to 13 bits)
    return *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)addr;
}
__device__ inline uint2 f_gelu_tab4(gt_f32x4 v) {
    const uint32_t r0 = f_gelu_tab1(v.x), r1 = f_gelu_tab1(v.y), r2 = f_gelu_tab1(v.z), r3 = f_gelu_tab1(v.w);
    return uint2{r0 | (r1 << 16), r2 | (r3 << 16)};
}

// the 16 KB device copy (built once per process by gelu_table_create, ffn.hip)
int gelu_table_create();
const uint16_t *gelu_table_dev();

}  // namespace ak
