// attn_d.h -- the DMA-staged attention item of attention.hip (k_attn_d) as a device function, with the small lane-exchange / store
// helpers it shares with the other attention kernels. Two callers: k_attn_d itself (attention.hip: one item per workgroup, plain
// loads) and the single-launch query forward (query_forward.hip: the items of a layer as one PHASE of a persistent kernel, where
// q / k / V^T / the mask were written by other workgroups of the same launch and must be read past this CU's vector L1 -- the LD
// policy). Same instructions on the same values in the same order either way: the two are bit-identical by construction.
#pragma once
#include "mfma_tile.h"
#include "encoder_kernels.h"

namespace ak {
using namespace mt;

// LOAD POLICIES for data another workgroup of the SAME launch may have written (query_forward.hip). A CU's vector L1 is never
// refreshed by another CU's stores (MI355X_MICROARCH.md, "Workgroup dispatch ..."): such data is read with device-scope (sc1)
// loads, which are served by the XCD's L2 -- the point where the stores of every CU of that XCD meet. LdPlain: ordinary loads
// (everything a kernel reads was written before its launch).
struct LdPlain {
    static __device__ __forceinline__ uint4 u4(const void *p) { return *(const uint4 *)p; }
    static __device__ __forceinline__ uint2 u2(const void *p) { return *(const uint2 *)p; }
    static __device__ __forceinline__ float4 f4(const void *p) { return *(const float4 *)p; }
    static __device__ __forceinline__ uint32_t u32(const void *p) { return *(const uint32_t *)p; }
    static __device__ __forceinline__ int i32(const void *p) { return *(const int *)p; }
    static __device__ __forceinline__ void glds16(const void *g, uint32_t lds_wave_base) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory", "m0");
    }
};
struct LdL2 {
    static __device__ __forceinline__ uint64_t q8(const void *p) {
        return __hip_atomic_load((const uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // global_load_dwordx2 ... sc1
    }
    static __device__ __forceinline__ uint4 u4(const void *p) {
        const uint64_t a = q8(p), b = q8((const char *)p + 8);
        return uint4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    }
    static __device__ __forceinline__ uint2 u2(const void *p) { const uint64_t a = q8(p); return uint2{(uint32_t)a, (uint32_t)(a >> 32)}; }
    static __device__ __forceinline__ float4 f4(const void *p) {
        const uint4 v = u4(p);
        return float4{__builtin_bit_cast(float, v.x), __builtin_bit_cast(float, v.y), __builtin_bit_cast(float, v.z), __builtin_bit_cast(float, v.w)};
    }
    static __device__ __forceinline__ uint32_t u32(const void *p) {
        return __hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ int i32(const void *p) { return (int)u32(p); }
    static __device__ __forceinline__ void glds16(const void *g, uint32_t lds_wave_base) {                // the LDS-DMA form of the same: sc1
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1"
                     :: "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory", "m0");
    }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));


// occupancy target: 4 waves per SIMD at hd = 32 (128 registers; without it hipcc parks the score tile in AGPRs, 130
// registers and 240 copy instructions per 128-key chunk), 2 at hd = 64
// NW waves per workgroup (32 queries each) share one staged K / V^T: 16 for S >= 512, 8 for S >= 256, else 4. Fewer,
// larger workgroups stage K/V once instead of 2-4 times, and at hd = 64, S = 512 (142 KB of LDS: one workgroup per
// CU) they put 4 waves on a SIMD instead of 1: 653 -> 209 us per bge-base layer, 86 -> 72 us per MiniLM layer.
// max over the two halves of the wave (lane i with lane i ^ 32): v_permlane32_swap leaves {x[0..31], x[0..31]} and
// {x[32..63], x[32..63]} in its two operands
__device__ __forceinline__ float xhalf_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
// max of a lane's 16 scores as seven v_max3_f32 + one v_max_f32 (round 5; the fmaxf chain compiled to 8 v_max + 5 v_max3): the
// softmax loops of the launched kernels are bound by VALU issue, every instruction less counts (same box, kernel-trace averages:
// k_attn_s<64,16> 165.3 against 167.4 us, k_attn_d<32,8> 55.9 against 57.7). Quiet NaNs are skipped as by fmaxf.
// Measured on top of it and not kept: -m as the score MFMA's initial accumulator at hd 32 (sixteen persistent registers, no
// subtraction per block): 80 registers at six waves per SIMD spill inside the block loop -- 64-66 against 56 us.
__device__ __forceinline__ float max3a(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max16(const mt::f32x16 &x) {
    return fmaxf(max3a(max3a(x[0], x[1], x[2]), max3a(x[3], x[4], x[5]), max3a(x[6], x[7], x[8])),
                 max3a(max3a(x[9], x[10], x[11]), max3a(x[12], x[13], x[14]), x[15]));
}

// Context rows of one wave: lane (query r, half kh) holds, per 32-feature tile, features 8g + 4kh + {0..3}, g = 0..3 -- four 8-byte
// runs. One v_permlane32_swap per packed register pair trades runs with the lane of the same query in the other half, so that
// half 0 holds features 0-7 and 16-23 and half 1 features 8-15 and 24-31: two 16-byte stores per lane and tile instead of four
// 8-byte ones, 32 contiguous bytes per row and instruction (the stores of a finishing wave queue behind each other).
// Every lane of the wave must call it (the exchange); `live` masks the stores of rows past the sequence.
template <int DB>
__device__ __forceinline__ void store_ctx_rows(const f32x16 (&o)[DB], float inv, uint16_t *dst, int kh, bool live) {
#pragma unroll
    for (int d = 0; d < DB; d++) {
        uint32_t w[4][2];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            w[g][0] = pack_bf16x2(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
            w[g][1] = pack_bf16x2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        }
#pragma unroll
        for (int gp = 0; gp < 4; gp += 2)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const auto x = __builtin_amdgcn_permlane32_swap(w[gp][j], w[gp + 1][j], false, false);
                w[gp][j] = x[0]; w[gp + 1][j] = x[1];
            }
        if (live) {
            *(uint4 *)(dst + d * 32 + 8 * kh) = uint4{w[0][0], w[0][1], w[1][0], w[1][1]};
            *(uint4 *)(dst + d * 32 + 16 + 8 * kh) = uint4{w[2][0], w[2][1], w[3][0], w[3][1]};
        }
    }
}

__device__ inline void glds16(const void *g, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory", "m0");
}


template <int HD, int NW, class LD>
__device__ __forceinline__ void attn_d_body(const AttnArgs &a, const int it_, char *smem) {
    constexpr int DB = HD / 32, KSTEPS = HD / 16, KROW = HD * 2, CRK = HD / 8, PERKEY = KROW + 2 * HD + 4;
    const int S = a.S, H = a.H, heads = a.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const uint32_t lds0 = lds_addr(smem);
    const int nqb = (S + NW * 32 - 1) / (NW * 32);
    const int it = it_, bh = it / nqb, qb = it - bh * nqb, b = bh / heads, h = bh - b * heads;
    const uint32_t flags_raw = __builtin_amdgcn_readfirstlane(LD::u32(a.blkmask + b));
    const uint32_t flags_all = flags_raw & 0xffffu, full_all = flags_raw >> 16;        // blocks with a real key / of 32 real keys
    const int kx = (kh ^ (HD == 64 ? (r >> 1) & 7 : (r >> 2) & 3)) << 4;
    // tiles: 256-key tiles, then 128 / 64 / 32 (power-of-two rows for the V^T swizzle); tile at key k0 sits at LDS byte k0 * PERKEY
    const char *kg0 = (const char *)(a.k + ((int64_t)b * S) * H + (int64_t)h * a.qk_hs);
    const char *vg0 = (const char *)(a.vt + ((int64_t)b * H + h * HD) * S);
    const int q0 = qb * (NW * 32) + wave * 32;
    int qrow = q0 + r;
    if (qrow >= S) qrow = S - 1;
    uint4 qf[KSTEPS];
#pragma unroll
    for (int st = 0; st < KSTEPS; st++)
        qf[st] = LD::u4(a.q + (int64_t)b * S * H + (int64_t)h * a.qk_hs + (int64_t)qrow * a.qk_ld + st * 16 + kh * 8);
    for (int k0 = 0; k0 < S;) {
        int kt = 256;
        while (kt > S - k0) kt >>= 1;
        const uint32_t sb = lds0 + k0 * PERKEY;
        const int nkp = (kt * KROW) >> 10;
        const int lcr = 31 - __clz(kt >> 3);
        const int vsh = lcr >= 4 ? 0 : 4 - lcr, vmsk = (lcr >= 4 ? 16 : (1 << lcr)) - 1;
        for (int p = wave; p < 2 * nkp + 1; p += NW) {
            if (p < nkp) {
                const int g = p * 64 + lane, row = g / CRK, c = g % CRK;
                const int swz = HD == 64 ? (row >> 1) & 7 : (row >> 2) & 3;
                LD::glds16(kg0 + (int64_t)(k0 + row) * a.qk_ld * 2 + ((c ^ swz) << 4), sb + p * 1024);
            } else if (p < 2 * nkp) {
                const int g = (p - nkp) * 64 + lane, row = g >> lcr, c = g & ((1 << lcr) - 1);
                const int swz = (row >> vsh) & vmsk;
                LD::glds16(vg0 + ((int64_t)row * S + k0) * 2 + ((c ^ swz) << 4), sb + kt * KROW + (p - nkp) * 1024);
            } else if (lane * 4 < kt) {
                LD::glds16(a.maskf + (int64_t)b * S + k0 + lane * 4, sb + kt * KROW + HD * kt * 2);
            }
        }
        k0 += kt;
    }
    wait_vm<0>();
    __syncthreads();
    if (q0 >= S) return;

    f32x16 o[DB];
    float m = 0.f;
    f32x2 l2 = {0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[d][e] = 0.f;
    // One 32-key block: scores (the additive 0 / -inf key mask is the MFMA's initial accumulator), lazy running maximum (see
    // k_attn), exponentials, P . V. The padding mask is per key, so every query of the wave meets its first real key in the
    // same block -- the lowest set bit of the sequence's block bitmap, known before the loop: `first` is wave-uniform, its
    // selects are scalar, and every other block only checks whether a score exceeds the current reference by more than 2^8
    // (one compare + a scalar branch, rarely taken). The two halves of a query's column (lanes r, r + 32) meet in one
    // v_permlane32_swap, not an LDS permute.
    auto block = [&](const char *kr, const float *mrow, const char *vr, int voff, int kt2, int vx, bool first, bool full) {
        f32x16 acc;
        if (full) {                                       // 32 real keys (wave-uniform): zero constant instead of the mask's four LDS reads
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc = mfma_bf16(*(const uint4 *)(kr + kx), qf[0], z);
        } else {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 mk = *(const float4 *)&mrow[8 * g + 4 * kh];
                acc[4 * g + 0] = mk.x; acc[4 * g + 1] = mk.y; acc[4 * g + 2] = mk.z; acc[4 * g + 3] = mk.w;
            }
            acc = mfma_bf16(*(const uint4 *)(kr + kx), qf[0], acc);
        }
#pragma unroll
        for (int st = 1; st < KSTEPS; st++) acc = mfma_bf16(*(const uint4 *)(kr + ((st << 5) ^ kx)), qf[st], acc);
        {
            const f32x2 mm = {m, m};                  // m = 0 until the first live block has set it
#pragma unroll
            for (int e = 0; e < 16; e += 2) {         // subtraction first: its results need no canonicalising v_max
                const f32x2 x = f32x2{acc[e], acc[e + 1]} - mm;
                acc[e] = x[0]; acc[e + 1] = x[1];
            }
#if AK_DBG_KERNELS
            float mx = -__builtin_inff();             // (A/B reference: the fmaxf chain of rounds 2-4)
#pragma unroll
            for (int e = 0; e < 16; e += 2) mx = fmaxf(mx, fmaxf(acc[e], acc[e + 1]));
#else
            float mx = max16(acc);
#endif
            mx = xhalf_max(mx);
            if (first || __any(mx > 8.f)) {           // `first` is wave-uniform: a scalar select, not a per-lane one
                const float delta = first ? (mx > -__builtin_inff() ? mx : 0.f) : fmaxf(mx, 0.f);
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
                m += delta;
                const f32x2 dd = {delta, delta};
                l2 *= alpha;
#pragma unroll
                for (int d = 0; d < DB; d++)
#pragma unroll
                    for (int e = 0; e < 16; e++) o[d][e] *= alpha;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 x = f32x2{acc[e], acc[e + 1]} - dd;
                    acc[e] = x[0]; acc[e + 1] = x[1];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const f32x2 pv = {__builtin_amdgcn_exp2f(acc[e]), __builtin_amdgcn_exp2f(acc[e + 1])};
            acc[e] = pv[0]; acc[e + 1] = pv[1];
            l2 += pv;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const uint4 pb = {pack_bf16x2(acc[8 * s2 + 0], acc[8 * s2 + 1]), pack_bf16x2(acc[8 * s2 + 2], acc[8 * s2 + 3]),
                              pack_bf16x2(acc[8 * s2 + 4], acc[8 * s2 + 5]), pack_bf16x2(acc[8 * s2 + 6], acc[8 * s2 + 7])};
#pragma unroll
            for (int d = 0; d < DB; d++) {
                const uint4 va = *(const uint4 *)(vr + d * 32 * kt2 + ((voff + s2 * 32) ^ vx));
                o[d] = mfma_bf16(va, pb, o[d]);
            }
        }
    };
    const int fb = flags_all ? __builtin_ctz(flags_all) : -1;      // wave-uniform: the first block that holds a real key
    for (int k0 = 0; k0 < S;) {
        int kt = 256;
        while (kt > S - k0) kt >>= 1;
        const char *sb = smem + k0 * PERKEY;
        const char *sV = sb + kt * KROW;
        const float *sM = (const float *)(sV + HD * kt * 2);
        const int lcr = 31 - __clz(kt >> 3);
        const int vsh = lcr >= 4 ? 0 : 4 - lcr, vmsk = (lcr >= 4 ? 16 : (1 << lcr)) - 1;
        const int vx = (kh ^ ((r >> vsh) & vmsk)) << 4;
        const char *krow = sb + r * KROW;
        const char *vrow = sV + r * (kt * 2);
        const int b0 = k0 >> 5;
        const uint32_t flags = flags_all >> b0, fullf = full_all >> b0;
        for (int blk = 0; blk < (kt >> 5); blk++) {
            if (!((flags >> blk) & 1)) continue;        // padding only: exp2(-inf) = 0 in every sum
            block(krow + blk * 32 * KROW, sM + blk * 32, vrow, blk * 64, kt * 2, vx, b0 + blk == fb, (fullf >> blk) & 1);
        }
        k0 += kt;
    }
    float l = l2[0] + l2[1];
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    {
        const int orow = q0 + r < S ? q0 + r : S - 1;
        store_ctx_rows<DB>(o, inv, a.ctx + ((int64_t)b * S + orow) * H + h * HD, kh, q0 + r < S);
    }
}


}  // namespace ak
