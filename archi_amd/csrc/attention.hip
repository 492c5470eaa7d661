// attention.hip -- bidirectional multi-head self-attention of the BERT encoder (a1/a2):
//   ctx = softmax(q k^T / sqrt(hd) + mask) v      per (sequence, head), hd = 32 or 64, S <= 512
// q is pre-scaled by log2(e)/sqrt(hd) (so the softmax runs on v_exp_f32 = 2^x directly) and v arrives
// transposed ([B][H][S]) from the QKV GEMM epilogue (gemm.hip).
//
// One wave owns 32 queries. The score tile is computed SWAPPED (A = keys, B = queries), so a lane
// owns one query column and 16 keys per 32x32 MFMA tile: the softmax max/sum are lane-local plus
// one exchange with lane^32. P feeds the P.V MFMA straight from registers (K-order of the two
// operands is chosen to match the accumulator layout, no LDS round trip). Keys are processed in
// chunks of 128 with an online-softmax rescale, so S = 512 fits the register file.
#include "mfma_tile.h"
#include "encoder_kernels.h"

namespace ak {
using namespace mt;


// occupancy target: 4 waves per SIMD at hd = 32 (128 registers; without it hipcc parks the score tile in AGPRs, 130
// registers and 240 copy instructions per 128-key chunk), 2 at hd = 64
// NW waves per workgroup (32 queries each) share one staged K / V^T: 16 for S >= 512, 8 for S >= 256, else 4. Fewer,
// larger workgroups stage K/V once instead of 2-4 times, and at hd = 64, S = 512 (142 KB of LDS: one workgroup per
// CU) they put 4 waves on a SIMD instead of 1: 653 -> 209 us per bge-base layer, 86 -> 72 us per MiniLM layer.
template <int HD, int NW, int CB = 4>   // CB: 32-key blocks per online-softmax chunk
__global__ __launch_bounds__(NW * 64, (HD == 32 && NW <= 8 && CB == 2) ? 6 : ((HD == 32 || NW == 16) ? 4 : 2)) void k_attn(AttnArgs a) {
    constexpr int NT = NW * 64;
    constexpr int DB = HD / 32, KSTEPS = HD / 16;
    constexpr int KSTRIDE = HD * 2 + 16;             // padded K row (bytes): conflict-free b128 reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, H = a.H;
    const int VSTRIDE = S * 2 + 16;                  // padded V^T row (bytes)
    char *sK = smem;
    char *sV = sK + S * KSTRIDE;
    float *sM = (float *)(sV + HD * VSTRIDE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * (NW * 32) + wave * 32;

    // ---- stage K [S][HD], V^T [HD][S], mask for this (b, h)
    {
        const uint16_t *kg = a.k + ((int64_t)b * S) * H + h * HD;
        constexpr int KC = HD / 8;                   // 16-B chunks per K row
        for (int i = tid; i < S * KC; i += NT) {
            int s = i / KC, c = i - s * KC;
            *(uint4 *)(sK + s * KSTRIDE + c * 16) = *(const uint4 *)(kg + (int64_t)s * H + c * 8);
        }
        const uint16_t *vg = a.vt + ((int64_t)b * H + h * HD) * S;
        const int VC = S / 8;
        for (int i = tid; i < HD * VC; i += NT) {
            int d = i / VC, c = i - d * VC;
            *(uint4 *)(sV + d * VSTRIDE + c * 16) = *(const uint4 *)(vg + (int64_t)d * S + c * 8);
        }
        for (int i = tid; i < S; i += NT) sM[i] = a.mask[b * S + i] ? 0.f : -__builtin_inff();
    }
    __syncthreads();
    if (q0 >= S) return;

    const int r = lane & 31, kh = lane >> 5;
    int qrow = q0 + r;
    if (qrow >= S) qrow = S - 1;
    uint4 qf[KSTEPS];
#pragma unroll
    for (int st = 0; st < KSTEPS; st++)
        qf[st] = *(const uint4 *)(a.q + ((int64_t)b * S + qrow) * H + h * HD + st * 16 + kh * 8);

    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[d][e] = 0.f;
    float m = -__builtin_inff(), l = 0.f;

    for (int kc0 = 0; kc0 < S; kc0 += 32 * CB) {
        const int nblk = (S - kc0) >= 32 * CB ? CB : (S - kc0) / 32;
        f32x16 sc[CB];
#pragma unroll
        for (int blk = 0; blk < CB; blk++) {
            if (blk >= nblk) {       // ragged tail chunk only: keys past S contribute nothing
#pragma unroll
                for (int e = 0; e < 16; e++) sc[blk][e] = -__builtin_inff();
            } else {
                // the additive mask (0 / -inf per key = per accumulator row) is the MFMA's initial accumulator:
                // no zero fill and no separate add
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 mk = *(const float4 *)&sM[kc0 + blk * 32 + 8 * g + 4 * kh];
                    acc[4 * g + 0] = mk.x; acc[4 * g + 1] = mk.y; acc[4 * g + 2] = mk.z; acc[4 * g + 3] = mk.w;
                }
                const char *kr = sK + (kc0 + blk * 32 + r) * KSTRIDE + kh * 16;
#pragma unroll
                for (int st = 0; st < KSTEPS; st++) acc = mfma_bf16(*(const uint4 *)(kr + st * 32), qf[st], acc);
                sc[blk] = acc;
            }
        }
        float mx = -__builtin_inff();
#pragma unroll
        for (int blk = 0; blk < CB; blk++)
#pragma unroll
            for (int e = 0; e < 16; e++) mx = fmaxf(mx, sc[blk][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        // no unmasked key seen yet: keep everything at zero without branching around the MFMAs
        const float m_use = (m_new == -__builtin_inff()) ? 0.f : m_new;
        const float alpha = (m == -__builtin_inff()) ? 0.f : __builtin_amdgcn_exp2f(m - m_use);
        m = m_new;
        float ls = 0.f;
#pragma unroll
        for (int blk = 0; blk < CB; blk++)
#pragma unroll
            for (int e = 0; e < 16; e++) { float p = __builtin_amdgcn_exp2f(sc[blk][e] - m_use); sc[blk][e] = p; ls += p; }
        l = l * alpha + ls;
#pragma unroll
        for (int d = 0; d < DB; d++)
#pragma unroll
            for (int e = 0; e < 16; e++) o[d][e] *= alpha;
#pragma unroll
        for (int blk = 0; blk < CB; blk++) {
            if (blk >= nblk) break;
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                uint4 pb = {pack_bf16x2(sc[blk][8 * s2 + 0], sc[blk][8 * s2 + 1]), pack_bf16x2(sc[blk][8 * s2 + 2], sc[blk][8 * s2 + 3]),
                            pack_bf16x2(sc[blk][8 * s2 + 4], sc[blk][8 * s2 + 5]), pack_bf16x2(sc[blk][8 * s2 + 6], sc[blk][8 * s2 + 7])};
#pragma unroll
                for (int d = 0; d < DB; d++) {
                    const char *vr = sV + (d * 32 + r) * VSTRIDE + (kc0 + blk * 32 + 16 * s2 + 4 * kh) * 2;
                    uint2 lo = *(const uint2 *)vr, hi = *(const uint2 *)(vr + 16);
                    uint4 va = {lo.x, lo.y, hi.x, hi.y};
                    o[d] = mfma_bf16(va, pb, o[d]);
                }
            }
        }
    }
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (q0 + r < S) {
        uint16_t *dst = a.ctx + ((int64_t)b * S + q0 + r) * H + h * HD;
#pragma unroll
        for (int d = 0; d < DB; d++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint2 ov = {pack_bf16x2(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv),
                            pack_bf16x2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv)};
                *(uint2 *)(dst + d * 32 + 8 * g + 4 * kh) = ov;
            }
    }
}

int launch_attn(const AttnArgs &a, hipStream_t st) {
    const int hd = a.H / a.heads;
    if (hd != 32 && hd != 64) AK_FAIL(-1, "attention: head size must be 32 or 64");
    if (a.S % 32 || a.S > 512) AK_FAIL(-1, "attention: S must be a multiple of 32 and <= 512");
    size_t lds = (size_t)a.S * (hd * 2 + 16) + (size_t)hd * (a.S * 2 + 16) + (size_t)a.S * 4;
    static bool attr = false;
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<32, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<64, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<64, 16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AK_HIP(hipFuncSetAttribute((const void *)k_attn<32, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    static const int force_nw = getenv("AK_ATTN_NW") ? atoi(getenv("AK_ATTN_NW")) : 0;
    const int nw = force_nw ? force_nw : (a.S >= 512 ? 16 : (a.S >= 256 ? 8 : 4));
    dim3 grid((a.S + nw * 32 - 1) / (nw * 32), a.heads, a.B);
    if (hd == 32) {
        if (nw == 16) k_attn<32, 16><<<grid, 1024, lds, st>>>(a);
        else if (nw == 8) k_attn<32, 8, 2><<<grid, 512, lds, st>>>(a);   // 64-key chunks: 78 registers, 3 workgroups per CU (72 vs 76 us)
        else k_attn<32, 4><<<grid, 256, lds, st>>>(a);
    } else {
        if (nw == 16) k_attn<64, 16, 2><<<grid, 1024, lds, st>>>(a);   // 64-key chunks: no spill at 128 registers (208 vs 216 us)
        else if (nw == 8) k_attn<64, 8><<<grid, 512, lds, st>>>(a);
        else k_attn<64, 4><<<grid, 256, lds, st>>>(a);
    }
    AK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ak
