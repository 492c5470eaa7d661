// gelu_table.h -- GELU by LDS table for the bf16 encoder path (ffn.hip: the fused hidden-384 layer kernel; gemm.hip: the
// epilogue of the wide FFN-up tile). Device-side lookup only; the table is built on the host in ffn.hip (gelu_table_host).
#pragma once
#include "common.h"

namespace ak {

typedef float gt_f32x4 __attribute__((ext_vector_type(4)));

// GELU BY TABLE (round 4). The phase-B operand is bf16, so the kernel needs gelu(x) to 8 significant bits only -- but the
// polynomial above is ~19 fp32 VALU operations per value, and VALU work does not hide under the SIMD partner's MFMAs on this
// part (k_ffn384r: 160 packed instructions under a partner's 48 MFMAs took 1.6 k cycles, ~10 per instruction; the first table
// lookup, 4.5 VALU instructions per value, 905 cycles per chunk: still ~10 each). The lookup is therefore written for the
// fewest instructions: two pre-activations are converted to one packed f16 pair (v_cvt_pkrtz: round towards zero), both
// halves shifted right by two (v_pk_lshrrev_b16) and masked -- the upper 13 of an f16's 16 bits (sign + 5 exponent + 7
// mantissa bits) times two ARE the byte address of the entry, because the table sits at LDS address 0 (the kernels trap if
// their dynamic LDS does not start there): 2.5 VALU instructions + one ds_read_u16 per value, pairing included.
// Entry i covers the f16 bit patterns [8 i, 8 i + 8) and holds bf16(gelu(the bucket's midpoint, bit pattern 8 i + 4)), exact erf
// GELU in double precision on the host. f16's exponent range covers every magnitude that matters (below 2^-14 gelu(x) = x / 2
// is < 3e-5; a finite x above 65504 converts -- round towards zero -- to the largest finite f16 and reads the last finite bucket,
// bf16(gelu(65408)) = 65280: the function is clamped there, which no bf16 activation of a trained encoder comes near). The exponent-31
// entries keep non-finite inputs non-finite: NaN -> NaN, +inf -> +inf, -inf -> -0.
// Error: truncation + midpoint entry = the input moved by at most half a bucket (relative 2^-8), like a bf16 tensor's rounding
// (2^-9) of the up-projection's output in a bf16 framework, then one bf16 rounding of the exact function.
constexpr int GELU_TAB_BYTES = 8192 * 2;
typedef unsigned short gt_u16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t f_gelu_tab_lds(uint32_t addr) { return *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)addr; }
// two values -> two bf16 results in one register
__device__ inline uint32_t f_gelu_tab2(float x, float y) {
    const gt_u16x2 s = __builtin_bit_cast(gt_u16x2, __builtin_amdgcn_cvt_pkrtz(x, y)) >> (gt_u16x2){2, 2};
    const uint32_t u = __builtin_bit_cast(uint32_t, s), mask = 0x3ffeu;
    const uint32_t a0 = u & mask;
    uint32_t a1, r;
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a1) : "s"(mask), "v"(u));
    const uint32_t r0 = f_gelu_tab_lds(a0), r1 = f_gelu_tab_lds(a1);      // ds_read_u16 zero-extends
    asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r) : "v"(r1), "v"(r0));
    return r;
}
__device__ inline uint32_t f_gelu_tab1(float x) { return f_gelu_tab2(x, x) & 0xffffu; }
__device__ inline uint2 f_gelu_tab4(gt_f32x4 v) { return uint2{f_gelu_tab2(v.x, v.y), f_gelu_tab2(v.z, v.w)}; }

// the 16 KB device copy (built once per process by gelu_table_create, ffn.hip)
int gelu_table_create();
const uint16_t *gelu_table_dev();

}  // namespace ak
