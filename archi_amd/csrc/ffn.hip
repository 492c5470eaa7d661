// ffn.hip -- the whole feed-forward block of a BERT layer in ONE kernel, for hidden size 384 (all-MiniLM-L6, the
// reference's default embedder: src/cli/templates/base-config.yaml:145):
//   x <- LayerNorm(x + W2 . GELU(W1 . x + b1) + b2) * gamma + beta          x: [T][384] bf16 (the residual stream, in place)
// Replaces, inside Embeddings.embed_documents (manager.py:373), what ran as two launches (gemm.hip MODE 1 -> gemm_ln.hip):
// their [T][1536] bf16 intermediate was written to HBM by the first and read back by the second -- 2 x 201 MB per layer at
// 65 536 tokens, 2.4 GB of a forward pass whose GEMM K-loops already ran at the vendor rate -- and each launch paid its own
// serial epilogue (137 + 90 us per layer). Here the intermediate never leaves the registers.
//
// Structure (gfx950, wave64, ONE wave per SIMD: 4 waves per workgroup, 512 registers per lane):
//   tile   = 128 tokens per workgroup, 32 per wave. A wave owns its tokens end to end; nothing is exchanged between waves.
//   X      = the wave's 32 token rows live in REGISTERS for the whole tile, already in MFMA B-operand layout
//            (lane = token, k-half; 24 K-steps x 16 B = 96 registers).
//   chunk  = 32 intermediate features at a time (48 chunks of I = 1536):
//            phase A   H^T[32 f x 32 t]  = W1[chunk] . X^T      24 MFMA 32x32x16, A operand from LDS, B = the X registers
//            GELU      bias + exact GELU on the 16 accumulator values a lane holds, rounded to bf16 and packed: in the
//                      accumulator layout a lane owns ONE token and 16 of the 32 features, which is exactly a B operand of
//                      two K = 16 steps once the K order of W2 is permuted to match (done once, at encoder creation)
//            phase B   Y^T[384 x 32 t]  += W2[:, chunk] . H     24 MFMA, A operand from LDS, B = the packed GELU output
//   weights= both matrices are re-laid out once (ak_encoder_create) in FRAGMENT order: per chunk 48 KB, of which every
//            1 KB piece is what one ds_read_b128 of the 64 lanes fetches. Staging is a plain contiguous LDS-DMA copy
//            (global_load_lds, 16 B per lane, full lines), the fragment reads are sequential and bank-conflict free, and
//            there is no swizzle arithmetic anywhere. 3-slot ring of 48 KB chunks: loads run two chunks ahead, one
//            barrier per chunk (48 MFMAs per wave).
//   epilogue= Y + b2 + residual -> LayerNorm over the token's 384 features (lane-local sums + one lane^32 exchange) ->
//            bf16 -> the residual stream, in place.
// MFMA work per wave and tile: 48 x 48 x 32 cycles = 74 k cycles; two tiles per CU at 65 536 tokens.
//
// Measured (round 2, MI355X, 65 536 tokens, AK_FFN_DBG phase counters per wave over its two tiles): wait + barrier 39 k,
// phase A + phase B 357 k (77 cycles per MFMA; 252 k = 55 with the in-loop staging removed: the 12 LDS-DMA pieces a wave
// issues per chunk cost it ~90 cycles each with one wave per SIMD and nothing to overlap them), GELU 9 k, epilogue 36 k:
// 203 us per layer against 227 us for the two launches it replaces; MiniLM forward 2.64 -> 2.56 ms. What was tried on the way:
// per-piece 64-bit VGPR addresses (hipcc put s_waitcnt vmcnt(0) in front of every DMA statement: one memory round trip per
// piece -- the SGPR-base form below has no address register to protect); bias / gamma / beta behind the ring (offsets beyond
// the 16-bit field: ~150 precomputed addresses, spilled, ~100 scratch reloads per tile); the residual re-read from global
// memory block by block (24 exposed round trips per tile -- it now comes out of the X registers with one lane^32 exchange);
// four accumulators round-robin in phase A, two-deep chains in phase B (both slower than what is below).
// The 8-wave variant further down (16 tokens per wave on 16x16x32 MFMAs, 256 registers, two waves per SIMD) is the one
// the encoder launches, with the attention output projection + LayerNorm-1 fused in front of it: MiniLM forward 2.33-2.36 ms
// (AK_FFN_W8=0 selects this 4-wave kernel, AK_FFN_ATT=0 the 8-wave kernel without the fused projection, for A/B).
#include <atomic>
#include "mfma_tile.h"
#include "encoder_kernels.h"
#include "switches.h"
#include "gelu_table.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace ak {
using namespace mt;

constexpr int F_H = 384, F_TOK = 128, F_CH = 32, F_NST = 3;
[[maybe_unused]] constexpr int F_THREADS = 256;     // (the 4-wave generation: dbg library)
[[maybe_unused]] constexpr int F_KS = F_H / 16;                       // 24 K-steps of phase A
[[maybe_unused]] constexpr int F_MO = F_H / 32;                       // 12 output row blocks of phase B
constexpr int F_W1_BYTES = F_CH * F_H * 2;           // 24 KB of W1 per chunk
constexpr int F_SLOT = 2 * F_W1_BYTES;               // + 24 KB of W2
[[maybe_unused]] constexpr int F_PPW = F_SLOT / 1024 / 4;             // 12 one-KB pieces per wave per chunk
constexpr int F_MAXI = 1536;
constexpr int F_PARAM_BYTES = (F_MAXI + 6 * F_H) * 4;     // 15 360: b1 | b2 | gamma | beta | bo | gamma1 | beta1
constexpr int F_LDS = F_PARAM_BYTES + F_NST * F_SLOT;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ inline uint2 f_cvt4(f32x4 v) { return __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4)); }
__device__ inline float f_clamp3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
// exact-GELU on four values: gemm.hip's polynomial erf (degree-9 minimax in u^2, max abs error 7.8e-6), same constants
__device__ inline f32x4 f_gelu4(f32x4 x) {
    f32x4 u = x * 0.70710678118654752f;
    u = {f_clamp3(u.x, -3.2f, 3.2f), f_clamp3(u.y, -3.2f, 3.2f), f_clamp3(u.z, -3.2f, 3.2f), f_clamp3(u.w, -3.2f, 3.2f)};
    const f32x4 t = u * u;
    f32x4 p = __builtin_elementwise_fma(t, (f32x4)(-2.400035948e-09f), (f32x4)(1.419115847e-07f));
    p = __builtin_elementwise_fma(p, t, (f32x4)(-3.739696922e-06f));
    p = __builtin_elementwise_fma(p, t, (f32x4)(5.846631029e-05f));
    p = __builtin_elementwise_fma(p, t, (f32x4)(-6.112857373e-04f));
    p = __builtin_elementwise_fma(p, t, (f32x4)(4.584099166e-03f));
    p = __builtin_elementwise_fma(p, t, (f32x4)(-2.581433021e-02f));
    p = __builtin_elementwise_fma(p, t, (f32x4)(1.118641943e-01f));
    p = __builtin_elementwise_fma(p, t, (f32x4)(-3.757072389e-01f));
    p = __builtin_elementwise_fma(p, t, (f32x4)(1.128325701e+00f));
    f32x4 e = p * u;
    e = {f_clamp3(e.x, -1.f, 1.f), f_clamp3(e.y, -1.f, 1.f), f_clamp3(e.z, -1.f, 1.f), f_clamp3(e.w, -1.f, 1.f)};
    const f32x4 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, e, hx);
}

// GELU BY TABLE: gelu_table.h (the lookup, shared with gemm.hip); the table itself is built below.
// device copy of the table, built once per process (ffn_relayout)
static const uint16_t *g_gelu_tab = nullptr;
const uint16_t *gelu_table_dev() { return g_gelu_tab; }
void gelu_table_host(uint16_t *t) {      // entry i = bf16(gelu(midpoint of the f16 bit patterns [8 i, 8 i + 8))), exact erf GELU in double
    for (int i = 0; i < 8192; i++) {
        const int sign = i >> 12, e = (i >> 7) & 31;
        const double m = (double)(i & 127) + 0.5;                        // bit pattern 8 i + 4: half a bucket above the bucket's start
        double v;
        if (e == 31) {
            // inf / NaN patterns. A pre-activation that is NaN converts to an f16 NaN (the quiet bit is mantissa bit 9: buckets
            // 64-127; anything with a non-zero upper mantissa: buckets 1-127) and must STAY NaN -- the store's suspect-row check
            // looks for it; bucket 0 is +-inf (and signalling NaNs with a 3-bit payload, which no conversion produces):
            // gelu(+inf) = +inf, gelu(-inf) = 0
            t[i] = (i & 127) ? (uint16_t)0x7fc0 : (sign ? (uint16_t)0x8000 : (uint16_t)0x7f80);
            continue;
        }
        if (e == 0) v = ldexp(m / 128.0, -14);                           // f16 subnormals
        else v = ldexp(1.0 + m / 128.0, e - 15);
        if (sign) v = -v;
        const double g = 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
        t[i] = f32_to_bf16((float)g);
    }
}
int gelu_table_create() {
    static std::mutex mu;                      // encoder creation and the first FFN-up launch may come from different threads
    std::lock_guard<std::mutex> lk(mu);
    if (g_gelu_tab) return 0;
    std::vector<uint16_t> t(8192);
    gelu_table_host(t.data());
    uint16_t *d;
    AK_HIP(hipMalloc((void **)&d, GELU_TAB_BYTES));
    AK_HIP(hipMemcpy(d, t.data(), GELU_TAB_BYTES, hipMemcpyHostToDevice));
    g_gelu_tab = d;
    return 0;
}

// ---- one-time weight re-layout (ak_encoder_create) --------------------------------------------------------------
// wf [I/32 chunks][48 KB]: first 24 KB = W1 part [24 K-steps][64 lanes][8 bf16], lane l = (feature row m = l & 31,
// k-half kh = l >> 5): W1[32c + m][16s + 8kh + e]; then 24 KB = W2 part [12 row blocks][2 steps][64 lanes][8 bf16]:
// W2[32mo + m][32c + 16s' + 8(e>>2) + 4kh + (e&3)] -- the K order in which phase A's accumulators hold H.
__global__ void k_ffn_relayout(const uint16_t *__restrict__ w1, const uint16_t *__restrict__ w2, int I, uint16_t *__restrict__ wf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte unit each
    const int64_t units = (int64_t)(I / F_CH) * (F_SLOT / 16);
    if (i >= units) return;
    const int c = (int)(i / (F_SLOT / 16)), u = (int)(i % (F_SLOT / 16));
    uint16_t v[8];
    if (u < F_W1_BYTES / 16) {
        const int s = u / 64, l = u % 64, m = l & 31, kh = l >> 5;
        for (int e = 0; e < 8; e++) v[e] = w1[(int64_t)(F_CH * c + m) * F_H + 16 * s + 8 * kh + e];
    } else {
        const int u2 = u - F_W1_BYTES / 16, mo = u2 / 128, sp = (u2 / 64) & 1, l = u2 % 64, m = l & 31, kh = l >> 5;
        for (int e = 0; e < 8; e++) v[e] = w2[(int64_t)(32 * mo + m) * I + F_CH * c + 16 * sp + 8 * (e >> 2) + 4 * kh + (e & 3)];
    }
    uint4 o;
    o.x = v[0] | ((uint32_t)v[1] << 16); o.y = v[2] | ((uint32_t)v[3] << 16);
    o.z = v[4] | ((uint32_t)v[5] << 16); o.w = v[6] | ((uint32_t)v[7] << 16);
    *(uint4 *)(wf + i * 8) = o;
}

// one fragment = what one ds_read_b128 of the 64 lanes fetches (measured and not kept: two ds_read_b64 over a split layout --
// identical time: the fragment reads are not what paces this kernel, nor is anything else about LDS; with the reads removed
// altogether the two MFMA phases take the same 357 k cycles per wave)
__device__ inline uint4 f_frag(const char *piece_lane16) { return *(const uint4 *)piece_lane16; }

// (the first, 4-wave generation: libarchi_hip_dbg.so only -- AK_FFN_W8=0 there)
#if AK_DBG_KERNELS
__global__ __launch_bounds__(F_THREADS, 1) void k_ffn384(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the small per-feature arrays sit at the FRONT of LDS: every read of them is one base register + an immediate offset
    // (behind the 144 KB ring their offsets exceed the 16-bit field, hipcc kept ~150 precomputed addresses live and spilled
    // them: ~100 scratch reloads per tile in the epilogue, 40 k cycles)
    float *s_b1 = (float *)smem;
    float *s_b2 = s_b1 + F_MAXI, *s_g = s_b2 + F_H, *s_be = s_g + F_H;
    char *ring = smem + F_PARAM_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const int NC = a.I / F_CH;
    const int ntiles = a.T / F_TOK;

    for (int i = tid; i < a.I; i += F_THREADS) s_b1[i] = a.b1[i];
    for (int i = tid; i < F_H; i += F_THREADS) { s_b2[i] = a.b2[i]; s_g[i] = a.gamma[i]; s_be[i] = a.beta[i]; }
    __syncthreads();

    long long t_wait = 0, t_stage = 0, t_a = 0, t_g = 0, t_b = 0, t_e = 0, t_m = 0;
#define FTICK(acc) do { if (a.dbg) { const long long now_ = (long long)__builtin_readcyclecounter(); acc += now_ - t_m; t_m = now_; } } while (0)
    const uint32_t lds0 = lds_addr(ring);
    // staging: wave w copies the 12 consecutive 1 KB pieces [12w, 12w + 12) of a chunk. Inside the chunk loop the twelve
    // instructions are issued ONE per group of four MFMAs (a global_load_lds costs the issuing wave ~50 cycles, and with one
    // wave per SIMD nothing else runs meanwhile: issued in one burst they were 700 cycles per chunk in which the matrix
    // pipe sat idle).
    // SGPR base + constant VGPR offset form: no address VGPR is ever rewritten, so hipcc has no write-after-read reason to
    // put an s_waitcnt vmcnt(0) in front of the statement (with per-piece 64-bit VGPR addresses it did -- every piece then
    // waited for all earlier ones to LAND: the "issue cost" of a burst was really one exposed memory round trip per chunk)
    const uint32_t voff = (uint32_t)lane * 16;
    const char *src_wave = (const char *)a.wf + (wave * F_PPW) * 1024;     // wave-uniform
    auto stage_piece = [&](int c, int i) {
        const char *base = src_wave + (int64_t)c * F_SLOT + i * 1024;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (c % F_NST) * F_SLOT + (wave * F_PPW + i) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
    };
    auto stage = [&](int c) {
#pragma unroll
        for (int i = 0; i < F_PPW; i++) stage_piece(c, i);
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint16_t *xrow = a.x16 + ((int64_t)tile * F_TOK + wave * 32 + r) * F_H;
        // the wave's 32 token rows as MFMA B operands: K-step s holds x[tok][16s + 8kh .. + 8]
        uint4 xb[F_KS];
#pragma unroll
        for (int s = 0; s < F_KS; s++) xb[s] = *(const uint4 *)(xrow + 16 * s + 8 * kh);
        stage(0);
        if (NC > 1) stage(1);

        f32x16 accY[F_MO];
#pragma unroll
        for (int mo = 0; mo < F_MO; mo++)
#pragma unroll
            for (int e = 0; e < 16; e++) accY[mo][e] = 0.f;

        if (a.dbg) t_m = (long long)__builtin_readcyclecounter();
        for (int c = 0; c < NC; c++) {
            // chunk c has landed (this wave's pieces: all but the newest 12 DMA operations; everyone's: after the barrier),
            // and every wave is done with chunk c - 1, whose slot chunk c + 2 overwrites
            if (c + 1 < NC) wait_vm<F_PPW>(); else wait_vm<0>();
            __syncthreads();
            FTICK(t_wait);
            const bool more = c + 2 < NC;            // workgroup-uniform
            const char *slot = ring + (c % F_NST) * F_SLOT + lane * 16;
            // ---- phase A: 24 accumulating MFMAs on ONE accumulator (measured against a round-robin over four accumulators:
            // 112 k vs 174 k cycles per wave for the phase -- the chain rides the pipe's accumulator forwarding).
            // Fragments four at a time, one group ahead (the sched_barrier keeps hipcc from hoisting every ds_read of the
            // phase to its top: 96 registers this kernel does not have).
            f32x16 h0;
#pragma unroll
            for (int e = 0; e < 16; e++) h0[e] = 0.f;
            {
                uint4 fa[2][4];
#pragma unroll
                for (int j = 0; j < 4; j++) fa[0][j] = f_frag(slot + j * 1024);
#pragma unroll
                for (int s0 = 0; s0 < F_KS; s0 += 4) {
                    if (s0 + 4 < F_KS) {
#pragma unroll
                        for (int j = 0; j < 4; j++) fa[((s0 >> 2) + 1) & 1][j] = f_frag(slot + (s0 + 4 + j) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; j++) h0 = mfma_bf16(fa[(s0 >> 2) & 1][j], xb[s0 + j], h0);
                    if (more) stage_piece(c + 2, s0 >> 2);               // pieces 0..5
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            FTICK(t_a);
            // ---- bias + GELU -> bf16, packed straight into phase B's B operands
            uint4 hb[2];
            {
                const float *bb = s_b1 + c * F_CH + 4 * kh;
                uint2 pk[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float4 bi = *(const float4 *)(bb + 8 * g);
                    const f32x4 v = {h0[4 * g + 0] + bi.x, h0[4 * g + 1] + bi.y, h0[4 * g + 2] + bi.z, h0[4 * g + 3] + bi.w};
                    pk[g] = f_cvt4(f_gelu4(v));
                }
                hb[0] = {pk[0].x, pk[0].y, pk[1].x, pk[1].y};
                hb[1] = {pk[2].x, pk[2].y, pk[3].x, pk[3].y};
            }
            FTICK(t_g);
            // ---- phase B: K-step outer, the 12 output blocks inner: consecutive MFMAs write different accumulators.
            const char *w2s = slot + F_W1_BYTES;
            auto frag_off = [](int idx) { const int sp = idx / F_MO, mo = idx % F_MO; return (mo * 2 + sp) * 1024; };   // idx = s' * 12 + mo
            {
                // fragments four at a time, one group ahead
                uint4 fb[2][4];
#pragma unroll
                for (int j = 0; j < 4; j++) fb[0][j] = f_frag(w2s + frag_off(j));
#pragma unroll
                for (int i0 = 0; i0 < 2 * F_MO; i0 += 4) {
                    if (i0 + 4 < 2 * F_MO) {
#pragma unroll
                        for (int j = 0; j < 4; j++) fb[((i0 >> 2) + 1) & 1][j] = f_frag(w2s + frag_off(i0 + 4 + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int idx = i0 + j;
                        accY[idx % F_MO] = mfma_bf16(fb[(i0 >> 2) & 1][j], hb[idx / F_MO], accY[idx % F_MO]);
                    }
                    if (more) stage_piece(c + 2, 6 + (i0 >> 2));         // pieces 6..11
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            FTICK(t_b);
        }
        // ---- epilogue: v = Y + b2 + residual; LayerNorm over the token's 384 features; bf16 back into the stream.
        // lane (token r, half kh) holds features 32mo + 8g + 4kh + j; lane^32 holds the other half of the same token.
        // Memory access as in gemm_ln.hip: in the accumulator layout a wave instruction would touch 32 token rows with 8
        // bytes each, so every 32-feature x 32-token block goes through a wave-private 2 KB LDS scratch (64-byte token rows,
        // 16-byte chunks XOR-swizzled by the row) and moves as 64-byte row segments, 4 lanes per token row.
        FTICK(t_stage);                                    // (measurement: t_stage now = the tile's drain before the epilogue)
        __syncthreads();                                   // every wave is out of the last chunk's fragments: the ring is free
        char *scr = ring + wave * 2048;
        const int tk = lane >> 2, ch = lane & 3;
        const int64_t t0 = (int64_t)tile * F_TOK + wave * 32;
        // The residual IS the X tile the wave still holds as B operands: x[tok][16s + 8kh' + e] sits in lane (tok, kh') of
        // xb[s]. The accumulator layout wants x[tok][32mo + 8g + 4kh + j] = element 4kh + j of K-step s = 2mo + (g >> 1) in the
        // lane with kh' = g & 1: own register for half of the groups, the lane^32 partner's for the other half -- one 8-byte
        // exchange per K-step instead of 24 more global loads per lane (whose latency the epilogue sat through, block by block).
        float sum = 0.f;
#pragma unroll
        for (int mo = 0; mo < F_MO; mo++) {
            f32x16 &v = accY[mo];
#pragma unroll
            for (int gp = 0; gp < 2; gp++) {
                const uint4 own = xb[2 * mo + gp];
                const uint32_t s0_ = kh ? own.x : own.z, s1_ = kh ? own.y : own.w;        // the half the partner needs
                const uint32_t r0_ = __shfl_xor(s0_, 32), r1_ = __shfl_xor(s1_, 32);
                // g = 2gp (needs the kh' = 0 lane's elements 4kh..): kh = 0 own .xy, kh = 1 partner's; g = 2gp + 1 (kh' = 1): kh = 0 partner's, kh = 1 own .zw
                const uint32_t e0 = kh ? r0_ : own.x, e1 = kh ? r1_ : own.y;
                const uint32_t o0 = kh ? own.z : r0_, o1 = kh ? own.w : r1_;
#pragma unroll
                for (int gg_ = 0; gg_ < 2; gg_++) {
                    const int g = 2 * gp + gg_;
                    const uint32_t w0 = gg_ ? o0 : e0, w1 = gg_ ? o1 : e1;
                    const float4 b2 = *(const float4 *)(s_b2 + 32 * mo + 8 * g + 4 * kh);
                    v[4 * g + 0] += b2.x + bf16_to_f32((uint16_t)w0);
                    v[4 * g + 1] += b2.y + bf16_to_f32((uint16_t)(w0 >> 16));
                    v[4 * g + 2] += b2.z + bf16_to_f32((uint16_t)w1);
                    v[4 * g + 3] += b2.w + bf16_to_f32((uint16_t)(w1 >> 16));
                    sum += (v[4 * g + 0] + v[4 * g + 1]) + (v[4 * g + 2] + v[4 * g + 3]);
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float mu = sum * (1.0f / F_H);
        float sq = 0.f;
#pragma unroll
        for (int mo = 0; mo < F_MO; mo++)
#pragma unroll
            for (int e = 0; e < 16; e++) { const float d = accY[mo][e] - mu; sq += d * d; }
        sq += __shfl_xor(sq, 32);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / F_H) + a.eps);
#pragma unroll
        for (int mo = 0; mo < F_MO; mo++) {
            const f32x16 &v = accY[mo];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = 32 * mo + 8 * g + 4 * kh;
                const float4 gg = *(const float4 *)(s_g + n), bt = *(const float4 *)(s_be + n);
                const f32x4 y = {(v[4 * g + 0] - mu) * rstd * gg.x + bt.x, (v[4 * g + 1] - mu) * rstd * gg.y + bt.y,
                                 (v[4 * g + 2] - mu) * rstd * gg.z + bt.z, (v[4 * g + 3] - mu) * rstd * gg.w + bt.w};
                *(uint2 *)(scr + r * 64 + ((g ^ ((r >> 2) & 3)) << 4) + kh * 8) = f_cvt4(y);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int tt = tk + 16 * i;
                const uint4 yo = *(const uint4 *)(scr + tt * 64 + ((ch ^ ((tt >> 2) & 3)) << 4));
                *(uint4 *)(a.x16 + (t0 + tt) * F_H + 32 * mo + ch * 8) = yo;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next tile's first DMA overwrites slots 0 and 1: every wave must be out of the last chunks' fragment reads
        __syncthreads();
        FTICK(t_e);
    }
    if (a.dbg && lane == 0) {
        long long *d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 6;
        d[0] = t_wait; d[1] = t_stage; d[2] = t_a; d[3] = t_g; d[4] = t_b; d[5] = t_e;
    }
#undef FTICK
}
#endif  // AK_DBG_KERNELS


// =====================================================================================================================
// 8-wave variant: the same block, 16 tokens per wave on v_mfma_f32_16x16x32_bf16, ~220 registers, TWO waves per SIMD --
// one wave's staging issue, barrier wait, GELU and epilogue run under its SIMD partner's MFMAs (in the 4-wave kernel
// above nothing does: 1 950 of its 4 400 cycles per chunk the matrix pipe sits idle).
//   lane = (token n = lane & 15, group kg = lane >> 4)
//   X      xb[s] = x[tok][32s + 8kg .. + 8], 12 K-steps (48 registers)
//   phase A  H^T[32 f x 16 t] as two 16-row blocks: 2 x 12 MFMAs; accumulator h[rb] holds H[16rb + 4kg + j][tok]
//   GELU   8 values per lane -> ONE B operand of phase B: logical k (kg, e): e < 4 -> feature 4kg + e, else 16 + 4kg + e - 4
//   phase B  Y^T[384 x 16 t] += W2[:, chunk] . H: 24 MFMAs (one K = 32 step per 16-row output block), 96 accumulator regs
// Weight layout wf16 per chunk: [2 rb][12 s][64 lanes][8 bf16] (W1), then [24 ob][64 lanes][8 bf16] (W2, K permuted).
// Phase counters of the fused-layer launch (AK_FFN_DBG, per wave and ring iteration): barrier wait 1.05 k, phase A 1.05 k,
// GELU + phase B 1.5 k cycles against 1.54 k of matrix-pipe work per SIMD -- and the same 1.54 k of LDS-array time: at 16 tokens
// per wave every 1 KB weight fragment feeds ONE 16-cycle MFMA, 256 B per clock and CU, the LDS peak. Same-box A/B, not kept:
// four accumulator chains in phase A (2.176 vs 2.156 ms per forward), all six ring pieces issued right behind the barrier
// instead of between the MFMA groups (2.12 vs 2.095), s_setprio 1 for waves 4-7 (2.150 vs 2.144).
// =====================================================================================================================
constexpr int G_THREADS8 = 512, G_KS = F_H / 32, G_OB = F_H / 16, G_PPW = F_SLOT / 1024 / 8;   // 12 K-steps, 24 blocks, 6 pieces per wave
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ inline f32x4v mfma16_bf16(uint4 a, uint4 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Attention output projection for the fused-layer kernel: Wo [384][384] as 6 parts of 48 KB, part p = output row blocks
// 4p .. 4p+3: [4 blocks][12 K-steps][64 lanes][8 bf16], lane l = (row m = l & 15, group kg = l >> 4): Wo[16(4p+obl) + m][32s + 8kg + e]
constexpr int G_WO_PARTS = F_H * F_H * 2 / F_SLOT;      // 6
__global__ void k_wo_relayout16(const uint16_t *__restrict__ wo, uint16_t *__restrict__ wof) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte unit each
    if (i >= (int64_t)G_WO_PARTS * (F_SLOT / 16)) return;
    const int p = (int)(i / (F_SLOT / 16)), u = (int)(i % (F_SLOT / 16));
    const int obl = u / (12 * 64), s_ = (u / 64) % 12, l = u % 64, m = l & 15, kg = l >> 4;
    uint4 o;
    const uint16_t *src = wo + (int64_t)(16 * (4 * p + obl) + m) * F_H + 32 * s_ + 8 * kg;
    o.x = src[0] | ((uint32_t)src[1] << 16); o.y = src[2] | ((uint32_t)src[3] << 16);
    o.z = src[4] | ((uint32_t)src[5] << 16); o.w = src[6] | ((uint32_t)src[7] << 16);
    *(uint4 *)(wof + i * 8) = o;
}

__global__ void k_ffn_relayout16(const uint16_t *__restrict__ w1, const uint16_t *__restrict__ w2, int I, uint16_t *__restrict__ wf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte unit each
    const int64_t units = (int64_t)(I / F_CH) * (F_SLOT / 16);
    if (i >= units) return;
    const int c = (int)(i / (F_SLOT / 16)), u = (int)(i % (F_SLOT / 16));
    uint16_t v[8];
    if (u < F_W1_BYTES / 16) {
        const int rb = u / (G_KS * 64), s = (u / 64) % G_KS, l = u % 64, m = l & 15, kg = l >> 4;
        for (int e = 0; e < 8; e++) v[e] = w1[(int64_t)(F_CH * c + 16 * rb + m) * F_H + 32 * s + 8 * kg + e];
    } else {
        const int u2 = u - F_W1_BYTES / 16, ob = u2 / 64, l = u2 % 64, m = l & 15, kg = l >> 4;
        for (int e = 0; e < 8; e++) v[e] = w2[(int64_t)(16 * ob + m) * I + F_CH * c + 16 * (e >> 2) + 4 * kg + (e & 3)];
    }
    uint4 o;
    o.x = v[0] | ((uint32_t)v[1] << 16); o.y = v[2] | ((uint32_t)v[3] << 16);
    o.z = v[4] | ((uint32_t)v[5] << 16); o.w = v[6] | ((uint32_t)v[7] << 16);
    *(uint4 *)(wf + i * 8) = o;
}

// ATT: the attention output projection + residual + LayerNorm-1 run IN FRONT of the feed-forward block, in the same launch
// (a.ctx = attention output, a.x16 = the layer's input = the residual): six more ring slots of Wo fragments, 48 MFMAs each
// like a feed-forward chunk, into the 24 accumulators that later hold Y; LayerNorm-1's output never goes to memory -- it
// is converted from the accumulator layout into the B-operand layout of phase A with a 4-lane exchange (ds_bpermute) and
// stays in the X registers, where the final epilogue also finds its residual. Replaces the gemm_ln.hip launch (42 us per
// layer, 2 x 50 MB of traffic).
// NWV: waves per workgroup = 16-token groups per tile. 8 is the throughput form; 4 (64-token tiles, one wave per SIMD) is launched
// while the 128-token tiles would leave CUs idle: a tile's time is the walk through the weight ring whatever its token count.
template <bool ATT, int NWV = 8>
__global__ __launch_bounds__(NWV * 64, 1) void k_ffn384w8(FfnArgs a) {
    constexpr int NTH = NWV * 64, PPW = F_SLOT / 1024 / NWV, PHALF = PPW / 2, TILE_TOK = 16 * NWV;     // ring pieces per wave and iteration
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_b1 = (float *)smem;
    float *s_b2 = s_b1 + F_MAXI, *s_g = s_b2 + F_H, *s_be = s_g + F_H;
    float *s_bo = s_be + F_H, *s_g1 = s_bo + F_H, *s_be1 = s_g1 + F_H;
    char *ring = smem + F_PARAM_BYTES;
    constexpr int NPRE = ATT ? G_WO_PARTS : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int NC = a.I / F_CH;
    const int ntiles = a.T / TILE_TOK;

    for (int i = tid; i < a.I; i += NTH) s_b1[i] = a.b1[i];
    for (int i = tid; i < F_H; i += NTH) { s_b2[i] = a.b2[i]; s_g[i] = a.gamma[i]; s_be[i] = a.beta[i]; }
    if (ATT)
        for (int i = tid; i < F_H; i += NTH) { s_bo[i] = a.bo[i]; s_g1[i] = a.gamma1[i]; s_be1[i] = a.beta1[i]; }
    __syncthreads();

    long long t_wait = 0, t_a = 0, t_b = 0, t_e = 0, t_m = 0;
#define GTICK(acc) do { if (a.dbg) { const long long now_ = (long long)__builtin_readcyclecounter(); acc += now_ - t_m; t_m = now_; } } while (0)
    const uint32_t lds0 = lds_addr(ring);
    const uint32_t voff = (uint32_t)lane * 16;
    // ring iteration `it` = slot it % 3 = the it-th 48 KB block of [Wo parts (ATT) | feed-forward chunks]
    const char *src_wave = (const char *)(ATT ? a.wof : a.wf) + (wave * PPW) * 1024;     // wave-uniform: this wave's PPW pieces of a block
    auto stage_piece = [&](int it, int i) {
        const char *base = src_wave + (int64_t)it * F_SLOT + i * 1024;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (it % F_NST) * F_SLOT + (wave * PPW + i) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
    };
    const int NT = NPRE + NC;

    // ONE tile per workgroup (grid = tiles; 160 KB of LDS keep it at one workgroup per CU anyway): inside a persistent tile loop
    // hipcc hoisted every tile-invariant address and offset out of the loop -- ~100 registers, spilled to scratch and reloaded
    // in the LayerNorm transition and the epilogue (40 k cycles per tile)
    {
        const int tile = blockIdx.x;
        if (tile >= ntiles) return;
        // wave-uniform row base (SGPRs) + ONE 32-bit lane offset: every global access below is base + lane offset + immediate
        // (with per-lane 64-bit pointers hipcc hoisted ~50 address pairs out of the tile loop and spilled them)
        const int64_t t0 = (int64_t)tile * TILE_TOK + wave * 16;
        const uint16_t *xbase = a.x16 + t0 * F_H;                        // 16 token rows of this wave
        const uint32_t lrow = (uint32_t)(n * F_H);                       // this lane's token row, in elements
        uint4 xb[G_KS];
        if (ATT) {
            const uint16_t *cbase = a.ctx + t0 * F_H;
#pragma unroll
            for (int s = 0; s < G_KS; s++) xb[s] = *(const uint4 *)(cbase + (lrow + 8 * kg) + 32 * s);      // attention output rows, for now
        } else {
#pragma unroll
            for (int s = 0; s < G_KS; s++) xb[s] = *(const uint4 *)(xbase + (lrow + 8 * kg) + 32 * s);
        }
#pragma unroll
        for (int i = 0; i < PPW; i++) stage_piece(0, i);
        if (NT > 1) {
#pragma unroll
            for (int i = 0; i < PPW; i++) stage_piece(1, i);
        }
        f32x4v accY[G_OB];
#pragma unroll
        for (int ob = 0; ob < G_OB; ob++) accY[ob] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        if (a.dbg) t_m = (long long)__builtin_readcyclecounter();

        if (ATT) {
            // ---- attention output projection: Z^T[384 x 16 tok] = Wo . ctx^T, one 48 KB part (4 row blocks x 12 K-steps) per
            // ring iteration, K-step outer / row block inner (consecutive MFMAs on different accumulators)
#pragma unroll
            for (int it = 0; it < NPRE; it++) {
                if (it + 1 < NT) wait_vm<PPW>(); else wait_vm<0>();
                __syncthreads();
                GTICK(t_wait);
                const bool more = it + 2 < NT;
                const char *slot = ring + (it % F_NST) * F_SLOT + lane * 16;
                auto off = [](int i) { return ((i & 3) * G_KS + (i >> 2)) * 1024; };     // i = 4s + obl -> piece obl*12 + s
                uint4 fa[2][4];
#pragma unroll
                for (int j = 0; j < 4; j++) fa[0][j] = f_frag(slot + off(j));
#pragma unroll
                for (int i0 = 0; i0 < 4 * G_KS; i0 += 4) {
                    if (i0 + 4 < 4 * G_KS) {
#pragma unroll
                        for (int j = 0; j < 4; j++) fa[((i0 >> 2) + 1) & 1][j] = f_frag(slot + off(i0 + 4 + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; j++)       // (the part loop is fully unrolled: 4 * it + j is a compile-time register index)
                        accY[4 * it + j] = mfma16_bf16(fa[(i0 >> 2) & 1][j], xb[i0 >> 2], accY[4 * it + j]);
                    if (more && (i0 >> 2) < PPW) stage_piece(it + 2, i0 >> 2);       // all pieces, over the first PPW of the twelve groups
                    __builtin_amdgcn_sched_barrier(0);
                }
                GTICK(t_a);
            }
            // ---- + bo + residual (the layer's input) -> LayerNorm-1 -> bf16 -> the X registers of the feed-forward block.
            // Register budget: the 96 accumulators stay live through this, so the residual comes in four at a time and
            // every pair of packed blocks is exchanged into its X register as soon as it exists.
            float sum = 0.f;
            {
                uint2 rq[2][4];
#pragma unroll
                for (int j = 0; j < 4; j++) rq[0][j] = *(const uint2 *)(xbase + (lrow + 4 * kg) + 16 * j);
#pragma unroll
                for (int o0 = 0; o0 < G_OB; o0 += 4) {
                    if (o0 + 4 < G_OB) {
#pragma unroll
                        for (int j = 0; j < 4; j++) rq[((o0 >> 2) + 1) & 1][j] = *(const uint2 *)(xbase + (lrow + 4 * kg) + 16 * (o0 + 4 + j));
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int ob = o0 + j;
                        const uint2 rr = rq[(o0 >> 2) & 1][j];
                        const float4 bo = *(const float4 *)(s_bo + 16 * ob + 4 * kg);
                        f32x4v &v = accY[ob];
                        v[0] += bo.x + bf16_to_f32((uint16_t)rr.x);
                        v[1] += bo.y + bf16_to_f32((uint16_t)(rr.x >> 16));
                        v[2] += bo.z + bf16_to_f32((uint16_t)rr.y);
                        v[3] += bo.w + bf16_to_f32((uint16_t)(rr.y >> 16));
                        sum += (v[0] + v[1]) + (v[2] + v[3]);
                    }
                }
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mu1 = sum * (1.0f / F_H);
            float sq1 = 0.f;
#pragma unroll
            for (int ob = 0; ob < G_OB; ob++)
#pragma unroll
                for (int e = 0; e < 4; e++) { const float d = accY[ob][e] - mu1; sq1 += d * d; }
            sq1 += __shfl_xor(sq1, 16);
            sq1 += __shfl_xor(sq1, 32);
            const float rstd1 = 1.0f / sqrtf(sq1 * (1.0f / F_H) + a.eps);
            // accumulator layout (lane kg holds features 16ob + 4kg + j) -> B-operand layout (lane kg holds 32s + 8kg + e):
            // features 32s + 8kg + e sit in block 2s + (kg >> 1), lanes kg' = 2(kg & 1) + (e >> 2) of the same token
            const int srcA = (n + 16 * (2 * (kg & 1))) << 2, srcB = srcA + (16 << 2);
            const bool hi = (kg >> 1) != 0;
#pragma unroll
            for (int s_ = 0; s_ < G_KS; s_++) {
                uint2 zp[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int ob = 2 * s_ + u, f = 16 * ob + 4 * kg;
                    const float4 gg = *(const float4 *)(s_g1 + f), bt = *(const float4 *)(s_be1 + f);
                    const f32x4v &v = accY[ob];
                    const f32x4 y = {(v[0] - mu1) * rstd1 * gg.x + bt.x, (v[1] - mu1) * rstd1 * gg.y + bt.y,
                                     (v[2] - mu1) * rstd1 * gg.z + bt.z, (v[3] - mu1) * rstd1 * gg.w + bt.w};
                    zp[u] = f_cvt4(y);
                    accY[ob] = (f32x4v){0.f, 0.f, 0.f, 0.f};
                }
                const uint32_t a0x = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[0].x), a0y = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[0].y);
                const uint32_t b0x = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[0].x), b0y = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[0].y);
                const uint32_t a1x = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[1].x), a1y = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[1].y);
                const uint32_t b1x = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[1].x), b1y = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[1].y);
                xb[s_] = {hi ? a1x : a0x, hi ? a1y : a0y, hi ? b1x : b0x, hi ? b1y : b0y};
            }
        }
        for (int it = NPRE; it < NT; it++) {
            const int c = it - NPRE;
            if (it + 1 < NT) wait_vm<PPW>(); else wait_vm<0>();
            __syncthreads();
            GTICK(t_wait);
            const bool more = it + 2 < NT;
            const char *slot = ring + (it % F_NST) * F_SLOT + lane * 16;
            // ---- phase A: two row blocks x 12 K-steps; the two accumulators alternate. Fragments four at a time, one group ahead.
            f32x4v h[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
            {
                // fragment index i = 2 * s + rb  ->  piece (rb * 12 + s)
                auto off = [](int i) { return ((i & 1) * G_KS + (i >> 1)) * 1024; };
                uint4 fa[2][4];
#pragma unroll
                for (int j = 0; j < 4; j++) fa[0][j] = f_frag(slot + off(j));
#pragma unroll
                for (int i0 = 0; i0 < 2 * G_KS; i0 += 4) {
                    if (i0 + 4 < 2 * G_KS) {
#pragma unroll
                        for (int j = 0; j < 4; j++) fa[((i0 >> 2) + 1) & 1][j] = f_frag(slot + off(i0 + 4 + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int i = i0 + j;
                        h[i & 1] = mfma16_bf16(fa[(i0 >> 2) & 1][j], xb[i >> 1], h[i & 1]);
                    }
                    if (more && (i0 >> 2) < PHALF) stage_piece(it + 2, i0 >> 2);       // first half of the pieces
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            GTICK(t_a);
            // ---- bias + GELU -> the B operand of phase B
            uint4 hb;
            {
                const float4 bi0 = *(const float4 *)(s_b1 + c * F_CH + 4 * kg), bi1 = *(const float4 *)(s_b1 + c * F_CH + 16 + 4 * kg);
                const f32x4 v0 = {h[0][0] + bi0.x, h[0][1] + bi0.y, h[0][2] + bi0.z, h[0][3] + bi0.w};
                const f32x4 v1 = {h[1][0] + bi1.x, h[1][1] + bi1.y, h[1][2] + bi1.z, h[1][3] + bi1.w};
                const uint2 p0 = f_cvt4(f_gelu4(v0)), p1 = f_cvt4(f_gelu4(v1));
                hb = {p0.x, p0.y, p1.x, p1.y};
            }
            // ---- phase B: 24 independent accumulators, one MFMA each
            const char *w2s = slot + F_W1_BYTES;
            {
                uint4 fb[2][4];
#pragma unroll
                for (int j = 0; j < 4; j++) fb[0][j] = f_frag(w2s + j * 1024);
#pragma unroll
                for (int o0 = 0; o0 < G_OB; o0 += 4) {
                    if (o0 + 4 < G_OB) {
#pragma unroll
                        for (int j = 0; j < 4; j++) fb[((o0 >> 2) + 1) & 1][j] = f_frag(w2s + (o0 + 4 + j) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; j++) accY[o0 + j] = mfma16_bf16(fb[(o0 >> 2) & 1][j], hb, accY[o0 + j]);
                    if (more && (o0 >> 2) < PHALF) stage_piece(it + 2, PHALF + (o0 >> 2));     // second half
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            GTICK(t_b);
        }
        // ---- epilogue. lane (token n, group kg) holds Y[16ob + 4kg + j][tok]; the 4 lanes n, n+16, n+32, n+48 hold a token's 384.
        // Residual out of the X registers: x[tok][16ob + 4kg + j] is element 4(kg&1) + j of K-step ob >> 1 in lane group
        // kg' = 2(ob&1) + (kg>>1) of the same token: one 16-byte cross-lane read (4 ds_bpermute) per output block.
        __syncthreads();                                   // the ring is free: per-wave output scratch below
        float sum = 0.f;
        const int src_lo = n + 16 * (kg >> 1);             // source lane for even ob; + 32 for odd ob
#pragma unroll
        for (int ob = 0; ob < G_OB; ob++) {
            const uint4 own = xb[ob >> 1];
            const int src = (src_lo + 32 * (ob & 1)) << 2;  // ds_bpermute takes a byte address
            const uint32_t g0 = __builtin_amdgcn_ds_bpermute(src, (int)own.x), g1 = __builtin_amdgcn_ds_bpermute(src, (int)own.y),
                           g2 = __builtin_amdgcn_ds_bpermute(src, (int)own.z), g3 = __builtin_amdgcn_ds_bpermute(src, (int)own.w);
            const uint32_t w0 = (kg & 1) ? g2 : g0, w1 = (kg & 1) ? g3 : g1;
            const float4 b2 = *(const float4 *)(s_b2 + 16 * ob + 4 * kg);
            f32x4v &v = accY[ob];
            v[0] += b2.x + bf16_to_f32((uint16_t)w0);
            v[1] += b2.y + bf16_to_f32((uint16_t)(w0 >> 16));
            v[2] += b2.z + bf16_to_f32((uint16_t)w1);
            v[3] += b2.w + bf16_to_f32((uint16_t)(w1 >> 16));
            sum += (v[0] + v[1]) + (v[2] + v[3]);
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mu = sum * (1.0f / F_H);
        float sq = 0.f;
#pragma unroll
        for (int ob = 0; ob < G_OB; ob++)
#pragma unroll
            for (int e = 0; e < 4; e++) { const float d = accY[ob][e] - mu; sq += d * d; }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / F_H) + a.eps);
        // output rows through a wave-private scratch: [16 tokens][384 bf16], rows padded to 784 bytes; then 16-byte chunks
        // leave in row order (a wave instruction = 1 KB of consecutive chunks)
        constexpr int ROWP = F_H * 2 + 16;
        char *scr = ring + wave * (16 * ROWP);
#pragma unroll
        for (int ob = 0; ob < G_OB; ob++) {
            const int f = 16 * ob + 4 * kg;
            const float4 gg = *(const float4 *)(s_g + f), bt = *(const float4 *)(s_be + f);
            const f32x4v &v = accY[ob];
            const f32x4 y = {(v[0] - mu) * rstd * gg.x + bt.x, (v[1] - mu) * rstd * gg.y + bt.y,
                             (v[2] - mu) * rstd * gg.z + bt.z, (v[3] - mu) * rstd * gg.w + bt.w};
            *(uint2 *)(scr + n * ROWP + f * 2) = f_cvt4(y);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int idx = i * 64 + lane, tk = idx / 48, ch = idx % 48;
            const uint4 yo = *(const uint4 *)(scr + tk * ROWP + ch * 16);
            *(uint4 *)(xbase + (uint32_t)(tk * F_H + ch * 8)) = yo;
        }
        __syncthreads();                                   // the next tile's staging overwrites the scratch
        GTICK(t_e);
    }
    if (a.dbg && lane == 0) {
        long long *d = a.dbg + ((size_t)blockIdx.x * NWV + wave) * 6;
        d[0] = t_wait; d[1] = 0; d[2] = t_a; d[3] = 0; d[4] = t_b; d[5] = t_e;
    }
#undef GTICK
}

// =====================================================================================================================
// k_ffn384p (round 3) -- the fused layer tail with the feed-forward block split over WAVE PAIRS.
// k_ffn384w8 gives every wave 16 tokens and all features: each 1 KB weight fragment it reads from LDS feeds ONE 16-cycle MFMA
// (ablation: the same kernel reading every other fragment runs the MiniLM forward in 1.89 instead of 2.10 ms). Here waves w and
// w + 4 form a pair that owns 32 tokens. Out-projection + LayerNorm-1 and the final LayerNorm-2 stay token-parallel (each wave
// its own 16 tokens, all 384 features: the code of k_ffn384w8); in between, for the 48 feed-forward chunks, a wave computes
//   phase A: ITS HALF of the chunk's 32 intermediate features (row block rb = role) for all 32 tokens of the pair,
//   phase B: ITS HALF of the 384 output features (out-blocks 12 role .. 12 role + 11) for all 32 tokens,
// so every fragment feeds TWO MFMAs (own tokens, partner's tokens) and the weight layouts are unchanged. What crosses the pair:
//   X   after LayerNorm-1 each wave needs the partner's 16 normalised token rows: 12 KB per wave through LDS, twice per tile half
//   H   per chunk, the GELU output of the partner's feature half: 16 B per lane, double-buffered, read ONE ITERATION LATER
//       (phase B of chunk c-1 runs in iteration c, after the barrier that publishes it: still one barrier per chunk)
//   Y   at the end each wave hands the partner the output features it accumulated for the partner's tokens (12 KB, fp32).
// Ring: two 48 KB slots. Iteration c computes A(c) out of slot c&1 (W1 part) and B(c-1) out of slot (c-1)&1 (W2 part); during
// it waves 0-3 stage W1(c+1) and waves 4-7 stage W2(c) (a wave's six pieces of a chunk are all W1 or all W2), each into a
// region whose previous content was consumed before the iteration's barrier.
// Measured and not kept (same box, MiniLM forward): k_ffn384w8 2.065-2.07 ms, this kernel 2.010, and this kernel with pairs
// (w, w ^ 1) and waves 4-7 one barrier behind waves 0-3 (two barriers per chunk; the GELU of one SIMD partner under the MFMAs
// of the other; bit-exact) 2.04-2.05: both phases are matrix-pipe phases, there is little for a stagger to interleave. The
// two waves of a SIMD taking the phases in opposite order ([B, A] vs [A, B]) needs both orders compiled: the allocator then
// keeps the partner's X rows in scratch (100+ spilled registers).
// Where a tile's ~204 k cycles go (AK_FFN_DBG=1, cycle stamps per section, 65 536 tokens, per wave): out-projection 25 k (its
// MFMAs alone: 10 k), LayerNorm-1 17.7 k, X exchange 4.8 k, the 48 chunks 138 k (2.8 k per chunk against 1.6 k of MFMA time for the two
// waves of a SIMD), Y exchange + LayerNorm-2 15.9 k, stores 2.1 k. With stages compiled out (kernel-trace, 237 us launch): no GELU
// -25 us, no phase-A MFMAs -30, no phase-B MFMAs -22, no out-projection MFMAs -20, fragments read once -7, no per-chunk barrier
// -2, two chunks instead of 48: 84 us -- a third of the launch is the token-parallel head and tail of the tile, fully exposed
// with one workgroup per CU. Tried against that and not kept (all bit-exact, same box, chunk loop / launch):
//   * the iteration as [A(c), B(c-1), GELU(c)] instead of [B(c-1), A(c), GELU(c)]: 174 k / 256 us -- B carries the staging, and
//     issued second its pieces have not landed at the next iteration's wait; with the staging moved into A: 336 k (the
//     per-group `stage_on` tests become vector-condition branches around every group);
//   * on top of the first, GELU(c) in six steps between B's MFMA groups (it does not depend on them): 190 k / 263 us --
//     packed fp32 VALU between the MFMAs of the same wave costs more than the idle matrix pipe it was meant to fill;
//   * the residual rows loaded at kernel start (24 8-byte fragments per lane): LayerNorm-1 17.7 -> 9.6 k, out-projection
//     25 -> 36 k (32-byte segments, four instructions per cache line: it is their L1 request rate, not HBM latency --
//     touching one dword per line at kernel start instead: no change in LayerNorm-1, +5 k in the projection).
// =====================================================================================================================
constexpr int P_RING = 2 * F_SLOT;                                    // 96 KB
constexpr int P_HX = 2 * 8 * 1024;                                    // H exchange: [parity][wave][64 lanes][16 B]
constexpr int P_LDS = F_PARAM_BYTES + P_RING + P_HX;

template <bool DBG>      // DBG: cycle stamps per section (AK_FFN_DBG=1, measurement only)
__global__ __launch_bounds__(G_THREADS8, 1) void k_ffn384p(FfnArgs a) {
    constexpr int NWV = 8, PPW = F_SLOT / 1024 / NWV, TILE_TOK = 16 * NWV, NPRE = G_WO_PARTS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_b1 = (float *)smem;
    float *s_b2 = s_b1 + F_MAXI, *s_g = s_b2 + F_H, *s_be = s_g + F_H;
    float *s_bo = s_be + F_H, *s_g1 = s_bo + F_H, *s_be1 = s_g1 + F_H;
    char *ring = smem + F_PARAM_BYTES;
    char *hx = ring + P_RING;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, partner = wave ^ 4;                    // wave-uniform
    const int vwave = 2 * (wave & 3) + role;                           // token group of the tile: pairs own 32 consecutive tokens
    const int n = lane & 15, kg = lane >> 4;
    const int NC = a.I / F_CH;
    const int ntiles = a.T / TILE_TOK;
    for (int i = tid; i < a.I; i += G_THREADS8) s_b1[i] = a.b1[i];
    for (int i = tid; i < F_H; i += G_THREADS8) {
        s_b2[i] = a.b2[i]; s_g[i] = a.gamma[i]; s_be[i] = a.beta[i];
        s_bo[i] = a.bo[i]; s_g1[i] = a.gamma1[i]; s_be1[i] = a.beta1[i];
    }
    __syncthreads();
    long long tq[7];
    tq[0] = DBG ? (long long)__builtin_readcyclecounter() : 0;
#define PTICK(i) do { if constexpr (DBG) tq[i] = (long long)__builtin_readcyclecounter(); } while (0)
    const uint32_t lds0 = lds_addr(ring);
    const uint32_t voff = (uint32_t)lane * 16;
    // piece i (0..5) of this wave's share of 48 KB block `blk` of [Wo parts | chunks] -> ring slot `slot`
    const char *src_wave = (const char *)a.wof + (wave * PPW) * 1024;
    auto stage_piece = [&](int blk, int slot, int i) {
        const char *base = src_wave + (int64_t)blk * F_SLOT + i * 1024;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + slot * F_SLOT + (wave * PPW + i) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
    };
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int64_t t0 = (int64_t)tile * TILE_TOK + vwave * 16;
    const uint16_t *xbase = a.x16 + t0 * F_H;                          // this wave's own 16 token rows
    const uint32_t lrow = (uint32_t)(n * F_H);
    // xs[0] = own tokens, xs[1] = the partner's tokens (B operands of phase A, 12 K-steps of 32 features)
    uint4 xs[2][G_KS];
    {
        const uint16_t *cbase = a.ctx + t0 * F_H;
#pragma unroll
        for (int s = 0; s < G_KS; s++) xs[0][s] = *(const uint4 *)(cbase + (lrow + 8 * kg) + 32 * s);      // attention output rows, for now
    }
#pragma unroll
    for (int i = 0; i < PPW; i++) stage_piece(0, 0, i);
    f32x4v accY[G_OB];
#pragma unroll
    for (int ob = 0; ob < G_OB; ob++) accY[ob] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    // ---- attention output projection, token-parallel as in k_ffn384w8: six Wo parts through the two slots, one ahead.
    // During the last part waves 0-3 stage W1 of chunk 0 instead (its slot's W1 region is free; W2 of chunk 0 follows in
    // iteration 0 of the chunk loop).
#pragma unroll
    for (int it = 0; it < NPRE; it++) {
        wait_vm<0>();
        __syncthreads();
        const char *slot = ring + (it & 1) * F_SLOT + lane * 16;
        auto off = [](int i) { return ((i & 3) * G_KS + (i >> 2)) * 1024; };     // i = 4s + obl -> piece obl*12 + s
        uint4 fa[2][4];
#pragma unroll
        for (int j = 0; j < 4; j++) fa[0][j] = f_frag(slot + off(j));
#pragma unroll
        for (int i0 = 0; i0 < 4 * G_KS; i0 += 4) {
            if (i0 + 4 < 4 * G_KS) {
#pragma unroll
                for (int j = 0; j < 4; j++) fa[((i0 >> 2) + 1) & 1][j] = f_frag(slot + off(i0 + 4 + j));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++)
                accY[4 * it + j] = mfma16_bf16(fa[(i0 >> 2) & 1][j], xs[0][i0 >> 2], accY[4 * it + j]);
            if ((i0 >> 2) < PPW) {
                if (it + 1 < NPRE) stage_piece(it + 1, (it + 1) & 1, i0 >> 2);
                else if (wave < 4) stage_piece(NPRE, 0, i0 >> 2);                 // W1 of chunk 0 -> slot 0 (chunk c -> slot c & 1)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    PTICK(1);
    // ---- + bo + residual (the layer's input) -> LayerNorm-1 -> bf16 -> xs[0] (k_ffn384w8's code)
    {
        float sum = 0.f;
        {
            uint2 rq[2][4];
#pragma unroll
            for (int j = 0; j < 4; j++) rq[0][j] = *(const uint2 *)(xbase + (lrow + 4 * kg) + 16 * j);
#pragma unroll
            for (int o0 = 0; o0 < G_OB; o0 += 4) {
                if (o0 + 4 < G_OB) {
#pragma unroll
                    for (int j = 0; j < 4; j++) rq[((o0 >> 2) + 1) & 1][j] = *(const uint2 *)(xbase + (lrow + 4 * kg) + 16 * (o0 + 4 + j));
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int ob = o0 + j;
                    const uint2 rr = rq[(o0 >> 2) & 1][j];
                    const float4 bo = *(const float4 *)(s_bo + 16 * ob + 4 * kg);
                    f32x4v &v = accY[ob];
                    v[0] += bo.x + bf16_to_f32((uint16_t)rr.x);
                    v[1] += bo.y + bf16_to_f32((uint16_t)(rr.x >> 16));
                    v[2] += bo.z + bf16_to_f32((uint16_t)rr.y);
                    v[3] += bo.w + bf16_to_f32((uint16_t)(rr.y >> 16));
                    sum += (v[0] + v[1]) + (v[2] + v[3]);
                }
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mu1 = sum * (1.0f / F_H);
        float sq1 = 0.f;
#pragma unroll
        for (int ob = 0; ob < G_OB; ob++)
#pragma unroll
            for (int e = 0; e < 4; e++) { const float d = accY[ob][e] - mu1; sq1 += d * d; }
        sq1 += __shfl_xor(sq1, 16);
        sq1 += __shfl_xor(sq1, 32);
        const float rstd1 = 1.0f / sqrtf(sq1 * (1.0f / F_H) + a.eps);
        const int srcA = (n + 16 * (2 * (kg & 1))) << 2, srcB = srcA + (16 << 2);
        const bool hi = (kg >> 1) != 0;
#pragma unroll
        for (int s_ = 0; s_ < G_KS; s_++) {
            uint2 zp[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int ob = 2 * s_ + u, f = 16 * ob + 4 * kg;
                const float4 gg = *(const float4 *)(s_g1 + f), bt = *(const float4 *)(s_be1 + f);
                const f32x4v &v = accY[ob];
                const f32x4 y = {(v[0] - mu1) * rstd1 * gg.x + bt.x, (v[1] - mu1) * rstd1 * gg.y + bt.y,
                                 (v[2] - mu1) * rstd1 * gg.z + bt.z, (v[3] - mu1) * rstd1 * gg.w + bt.w};
                zp[u] = f_cvt4(y);
            }
            const uint32_t a0x = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[0].x), a0y = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[0].y);
            const uint32_t b0x = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[0].x), b0y = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[0].y);
            const uint32_t a1x = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[1].x), a1y = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[1].y);
            const uint32_t b1x = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[1].x), b1y = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[1].y);
            xs[0][s_] = {hi ? a1x : a0x, hi ? a1y : a0y, hi ? b1x : b0x, hi ? b1y : b0y};
        }
    }
    PTICK(2);
    // ---- X exchange: the partner's normalised rows -> xs[1]. Slot 1 (the last Wo part, consumed) carries six K-steps per round.
    {
        char *mine = ring + F_SLOT + wave * (6 * 1024) + lane * 16;
        const char *theirs = ring + F_SLOT + partner * (6 * 1024) + lane * 16;
#pragma unroll
        for (int rnd = 0; rnd < 2; rnd++) {
            __syncthreads();                              // slot 1: every wave is done with the last Wo part / the previous round
#pragma unroll
            for (int s = 0; s < 6; s++) *(uint4 *)(mine + s * 1024) = xs[0][6 * rnd + s];
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 6; s++) xs[1][6 * rnd + s] = *(const uint4 *)(theirs + s * 1024);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the reads are back before slot 1 is staged over (iteration 0)
    }
    PTICK(3);
    // ---- the 48 chunks. acc[0][j] / acc[1][j]: out-block 12 role + j of the own / the partner's tokens
    f32x4v acc[2][G_KS];
#pragma unroll
    for (int j = 0; j < G_KS; j++) { acc[0][j] = (f32x4v){0.f, 0.f, 0.f, 0.f}; acc[1][j] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
    uint2 mprev[2] = {uint2{0, 0}, uint2{0, 0}};          // this wave's GELU output of the previous chunk (own, partner's tokens)
    const bool r1 = role != 0;
    // phase B of chunk c-1 (24 MFMAs; the iteration's six staging pieces ride between its groups)
    auto phase_b = [&](int c, int sblk, bool stage_on) {
        const uint4 pv = *(const uint4 *)(hx + (((c - 1) & 1) * 8 + partner) * 1024 + lane * 16);     // {their own tokens, their partner's}
        // the B operand holds the chunk's 32 features in the order [row block 0 | row block 1]; pv.zw = the partner's half
        // for MY tokens, pv.xy for ITS tokens
        const uint2 t_own = {pv.z, pv.w}, t_oth = {pv.x, pv.y};
        const uint4 hb0 = r1 ? uint4{t_own.x, t_own.y, mprev[0].x, mprev[0].y} : uint4{mprev[0].x, mprev[0].y, t_own.x, t_own.y};
        const uint4 hb1 = r1 ? uint4{t_oth.x, t_oth.y, mprev[1].x, mprev[1].y} : uint4{mprev[1].x, mprev[1].y, t_oth.x, t_oth.y};
        const char *w2s = ring + ((c - 1) & 1) * F_SLOT + F_W1_BYTES + (12 * role) * 1024 + lane * 16;
        uint4 fb[2][2];                      // two fragments (four MFMAs, 64 pipe cycles) per group, one group ahead
#pragma unroll
        for (int j = 0; j < 2; j++) fb[0][j] = f_frag(w2s + j * 1024);
#pragma unroll
        for (int o0 = 0; o0 < G_KS; o0 += 2) {
            if (o0 + 2 < G_KS) {
#pragma unroll
                for (int j = 0; j < 2; j++) fb[((o0 >> 1) + 1) & 1][j] = f_frag(w2s + (o0 + 2 + j) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                acc[0][o0 + j] = mfma16_bf16(fb[(o0 >> 1) & 1][j], hb0, acc[0][o0 + j]);
                acc[1][o0 + j] = mfma16_bf16(fb[(o0 >> 1) & 1][j], hb1, acc[1][o0 + j]);
            }
            if (stage_on) stage_piece(NPRE + sblk, sblk & 1, o0 >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // phase A of chunk c: row block `role`, twelve K-steps, both token sets; GELU; publish this wave's half of H
    uint2 mnew[2];
    auto phase_a = [&](int c) {
        const char *w1s = ring + (c & 1) * F_SLOT + (role * G_KS) * 1024 + lane * 16;
        f32x4v h[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
        uint4 fa[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++) fa[0][j] = f_frag(w1s + j * 1024);
#pragma unroll
        for (int s0 = 0; s0 < G_KS; s0 += 2) {
            if (s0 + 2 < G_KS) {
#pragma unroll
                for (int j = 0; j < 2; j++) fa[((s0 >> 1) + 1) & 1][j] = f_frag(w1s + (s0 + 2 + j) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                h[0] = mfma16_bf16(fa[(s0 >> 1) & 1][j], xs[0][s0 + j], h[0]);
                h[1] = mfma16_bf16(fa[(s0 >> 1) & 1][j], xs[1][s0 + j], h[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float4 bi = *(const float4 *)(s_b1 + c * F_CH + 16 * role + 4 * kg);
        const f32x4 v0 = {h[0][0] + bi.x, h[0][1] + bi.y, h[0][2] + bi.z, h[0][3] + bi.w};
        const f32x4 v1 = {h[1][0] + bi.x, h[1][1] + bi.y, h[1][2] + bi.z, h[1][3] + bi.w};
        mnew[0] = f_cvt4(f_gelu4(v0));
        mnew[1] = f_cvt4(f_gelu4(v1));
        *(uint4 *)(hx + ((c & 1) * 8 + wave) * 1024 + lane * 16) = uint4{mnew[0].x, mnew[0].y, mnew[1].x, mnew[1].y};
    };
    for (int c = 0; c <= NC; c++) {
        wait_vm<0>();
        __syncthreads();
        // staging of this iteration: waves 0-3 W1(c+1) -> slot (c+1)&1, waves 4-7 W2(c) -> slot c&1
        const int sblk = wave < 4 ? c + 1 : c;
        const bool stage_on = sblk < NC;
        if (c >= 1) phase_b(c, sblk, stage_on);
        else if (stage_on) {
#pragma unroll
            for (int i = 0; i < PPW; i++) stage_piece(NPRE + sblk, sblk & 1, i);
        }
        if (c < NC) phase_a(c);
        mprev[0] = mnew[0]; mprev[1] = mnew[1];
    }
    PTICK(4);
    // ---- Y exchange: acc[1][*] (my features of the partner's tokens) -> the partner; the ring is free behind this barrier
    __syncthreads();
    {
        char *mine = ring + wave * (12 * 1024) + lane * 16;
        const char *theirs = ring + partner * (12 * 1024) + lane * 16;
#pragma unroll
        for (int j = 0; j < G_KS; j++) *(f32x4v *)(mine + j * 1024) = acc[1][j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G_KS; j++) {
            const f32x4v got = *(const f32x4v *)(theirs + j * 1024);     // out-block 12 (1 - role) + j of MY tokens
            accY[j] = r1 ? got : acc[0][j];
            accY[G_KS + j] = r1 ? acc[0][j] : got;
        }
    }
    // ---- epilogue (k_ffn384w8's): + b2 + residual out of the X registers -> LayerNorm-2 -> bf16 rows, in place
    __syncthreads();                                       // the ring is free: per-wave output scratch below
    float sum = 0.f;
    const int src_lo = n + 16 * (kg >> 1);
#pragma unroll
    for (int ob = 0; ob < G_OB; ob++) {
        const uint4 own = xs[0][ob >> 1];
        const int src = (src_lo + 32 * (ob & 1)) << 2;
        const uint32_t g0 = __builtin_amdgcn_ds_bpermute(src, (int)own.x), g1 = __builtin_amdgcn_ds_bpermute(src, (int)own.y),
                       g2 = __builtin_amdgcn_ds_bpermute(src, (int)own.z), g3 = __builtin_amdgcn_ds_bpermute(src, (int)own.w);
        const uint32_t w0 = (kg & 1) ? g2 : g0, w1 = (kg & 1) ? g3 : g1;
        const float4 b2 = *(const float4 *)(s_b2 + 16 * ob + 4 * kg);
        f32x4v &v = accY[ob];
        v[0] += b2.x + bf16_to_f32((uint16_t)w0);
        v[1] += b2.y + bf16_to_f32((uint16_t)(w0 >> 16));
        v[2] += b2.z + bf16_to_f32((uint16_t)w1);
        v[3] += b2.w + bf16_to_f32((uint16_t)(w1 >> 16));
        sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mu = sum * (1.0f / F_H);
    float sq = 0.f;
#pragma unroll
    for (int ob = 0; ob < G_OB; ob++)
#pragma unroll
        for (int e = 0; e < 4; e++) { const float d = accY[ob][e] - mu; sq += d * d; }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / F_H) + a.eps);
    constexpr int ROWP = F_H * 2 + 16;
    char *scr = ring + wave * (16 * ROWP);
#pragma unroll
    for (int ob = 0; ob < G_OB; ob++) {
        const int f = 16 * ob + 4 * kg;
        const float4 gg = *(const float4 *)(s_g + f), bt = *(const float4 *)(s_be + f);
        const f32x4v &v = accY[ob];
        const f32x4 y = {(v[0] - mu) * rstd * gg.x + bt.x, (v[1] - mu) * rstd * gg.y + bt.y,
                         (v[2] - mu) * rstd * gg.z + bt.z, (v[3] - mu) * rstd * gg.w + bt.w};
        *(uint2 *)(scr + n * ROWP + f * 2) = f_cvt4(y);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    PTICK(5);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int idx = i * 64 + lane, tk = idx / 48, ch = idx % 48;
        const uint4 yo = *(const uint4 *)(scr + tk * ROWP + ch * 16);
        *(uint4 *)((uint16_t *)xbase + (uint32_t)(tk * F_H + ch * 8)) = yo;
    }
    if constexpr (DBG) {
        wait_vm<0>();
        tq[6] = (long long)__builtin_readcyclecounter();
        if (a.dbg && lane == 0) {
            long long *d = a.dbg + ((size_t)blockIdx.x * NWV + wave) * 6;
            for (int i = 0; i < 6; i++) d[i] = tq[i + 1] - tq[i];
        }
    }
#undef PTICK
}

// =====================================================================================================================
// k_ffn384r (round 4) -- the fused layer tail with the two waves of every SIMD in DIFFERENT ROLES during the 48 chunks.
// In k_ffn384p / k_ffn384w8 all eight waves run the same stream in step: both waves of a SIMD want the matrix pipe in the same
// interval (phase A, phase B), then both run their GELU while the pipe idles -- 2.8 k cycles per chunk for 1.54 k of pipe work --
// and every wave holds X (96 registers for a pair's 32 tokens) AND 96 output accumulators.
// Here a pair (waves p and p + 4: the two waves of one SIMD) still owns 32 tokens, but
//   wave p     (PRODUCER): phase A for both 16-feature row blocks and all 32 tokens (48 MFMAs) + GELU; holds the 32 token rows
//                          (96 registers) and no output accumulator; writes the chunk's H as two ready B operands (2 x 16 B per
//                          lane) into a double-buffered LDS exchange;
//   wave p + 4 (CONSUMER): phase B for all 384 output features of the 32 tokens (48 MFMAs into 192 accumulator registers) out
//                          of the H the producer published one iteration earlier; holds no X; issues ALL of the iteration's
//                          ring DMA in one burst behind the barrier (it waits for the pipe anyway while the producer's phase A
//                          owns it).
// One barrier per chunk as before. Out-projection + LayerNorm-1 and LayerNorm-2 + stores stay token-parallel over all eight
// waves (the code of k_ffn384w8 / k_ffn384p): the consumer hands its 16 normalised rows to the producer before the loop, and
// after it the producer hands them back while the consumer passes on the output accumulators of the producer's tokens (two
// rounds through the free ring).
// GELU: TAB = true reads it from the LDS table (GELU BY TABLE above; the bias is the accumulators' initial value); TAB = false
// (AK_FFN_GELU=poly) keeps f_gelu4 and adds every product in the order k_ffn384w8 does: bit-identical to it.
// What the measurements said on the way (MI355X, 65 536 tokens, same box unless noted; AK_FFN_DBG=1 stamps per role):
//   * roles alone, polynomial GELU: 223 us against k_ffn384p's 227 -- the producer was the critical path at 1.0 k cycles of
//     phase A + 1.6 k of GELU per chunk: its 160 packed fp32 instructions crawl under the consumer's MFMA stream (~10 cycles
//     each); the same GELU on single-issue VALU instructions (inline asm): slower still (twice the instructions at the same
//     ~10 cycles); the consumer doing the GELU of its own token block: slower (it delays the consumer's MFMAs, 3.3 k per chunk).
//     VALU work of one wave does not hide under its SIMD partner's MFMAs on this part; what helps is having less of it:
//   * GELU by table: producer 1.07 k + 0.96 k per chunk, consumer 0.74 k staging + 1.3 k phase B: 2.2 k per chunk in step;
//   * ring DMA issued between the MFMA groups instead of in a burst: 372 us (consumer) -- each issue stalls the wave ~70 cycles
//     in the middle of its MFMA stream; four of the twelve pieces moved to the producer's phase A: +4 %;
//   * with every MFMA compiled out the loop still took 2.7 k cycles per chunk in workgroups [0, 256) and 1.75 k in the rest:
//     the FIRST round of workgroups streams the layer's weights from HBM (the activations of a layer have pushed them out of
//     L2 and the Infinity Cache by then; a 32 768-token batch does not show it) and one chunk of lead does not cover a miss ->
//     L2 PREFETCH below (-2 % on the forward); scripts/micro/ldsdma_bw.hip: the LDS-DMA path itself moves 48 KB per 830-1100
//     cycles per CU from L2 (45-59 B/clk), whether or not all CUs read the same addresses;
//   * fragment reads three K-steps ahead in the producer (it has the registers), two groups in the consumer: within noise;
//   * the same L2 prefetch for the INPUT rows (attention output + residual, 192 KB) of the workgroup that follows on the XCD,
//     32 lines per chunk iteration: forward 1.90 against 1.87 ms -- the extra HBM stream slows the first round's chunk loop by
//     more than the second round's head gains. Not kept.
// Launch: 196 us against 223 for k_ffn384p in the same rocprofv3 run; MiniLM forward 2.01 -> 1.87 ms.
// =====================================================================================================================
constexpr int R_HX = 2 * 4 * 2048 + 256;                              // H exchange [parity][pair][token block][64 lanes][16 B] + 256 scrap bytes (L2 prefetch target)
constexpr int R_LDS = GELU_TAB_BYTES + F_PARAM_BYTES + P_RING + R_HX;
constexpr int R_PF = 4;                                               // L2 prefetch distance, in chunks

template <bool DBG, bool TAB, int ABL = 0>      // DBG: cycle stamps (AK_FFN_DBG=1); ABL: compile-time ablation mask (FfnArgs::ablate, DBG only)
__global__ __launch_bounds__(G_THREADS8, 1) void k_ffn384r(FfnArgs a) {
    constexpr int NWV = 8, PPW = F_SLOT / 1024 / NWV, TILE_TOK = 16 * NWV, NPRE = G_WO_PARTS;
    constexpr int NCP = 12;                                            // ring pieces per iteration and consumer wave
    constexpr int PD = 3, CD = 2;                                      // fragment groups read ahead: producer / consumer (192 accumulator registers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [GELU table 16 KB, at address 0][per-feature arrays 15 KB][ring 96 KB][H exchange 16 KB]
    float *s_b1 = (float *)(smem + GELU_TAB_BYTES);
    float *s_b2 = s_b1 + F_MAXI, *s_g = s_b2 + F_H, *s_be = s_g + F_H;
    float *s_bo = s_be + F_H, *s_g1 = s_bo + F_H, *s_be1 = s_g1 + F_H;
    char *ring = smem + GELU_TAB_BYTES + F_PARAM_BYTES;
    char *hx = ring + P_RING;
    if (TAB && lds_addr(smem) != 0) __builtin_trap();                  // f_gelu_tab1 addresses the table absolutely
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, pair = wave & 3;                       // wave-uniform; role 0 = producer, 1 = consumer
    const int vwave = 2 * pair + role;                                 // own 16-token group (head and tail are token-parallel)
    const int n = lane & 15, kg = lane >> 4;
    const int NC = a.I / F_CH;
    const int ntiles = a.T / TILE_TOK;
    for (int i = tid; i < a.I; i += G_THREADS8) s_b1[i] = a.b1[i];
    for (int i = tid; i < F_H; i += G_THREADS8) {
        s_b2[i] = a.b2[i]; s_g[i] = a.gamma[i]; s_be[i] = a.beta[i];
        s_bo[i] = a.bo[i]; s_g1[i] = a.gamma1[i]; s_be1[i] = a.beta1[i];
    }
    if constexpr (TAB) {
        for (int i = tid; i < GELU_TAB_BYTES / 16; i += G_THREADS8) *(uint4 *)(smem + i * 16) = ((const uint4 *)a.gelu_tab)[i];
    }
    __syncthreads();
    long long tq[7];
    tq[0] = DBG ? (long long)__builtin_readcyclecounter() : 0;
#define RTICK(i) do { if constexpr (DBG) tq[i] = (long long)__builtin_readcyclecounter(); } while (0)
    const uint32_t lds0 = lds_addr(ring);
    const uint32_t voff = (uint32_t)lane * 16;
    // 1 KB piece `pc` of 48 KB block `blk` of [Wo parts | chunks] -> ring slot `slot` (same piece offset)
    auto stage_piece = [&](int blk, int slot, int pc) {
        const char *base = (const char *)a.wof + (int64_t)blk * F_SLOT + pc * 1024;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + slot * F_SLOT + pc * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
    };
    // L2 PREFETCH. Every workgroup of the first round streams the layer's weights at the same time, and at the bench shape they
    // come from HBM: with the ring one chunk ahead every chunk then waits out a miss. This workgroup touches one 128-byte line
    // in 32 of every 48 KB block, R_PF chunks before the ring asks for the block -- the 32 workgroups that share an XCD's L2
    // (block b runs on XCD b % 8: placement for speed only, nothing depends on it) cover the block between them. The touch is a
    // 4-byte LDS-DMA into a scrap word per lane: no register to protect, and nothing ever waits for it (only the producer of
    // pair 0 prefetches inside the loop, and a producer issues no ring DMA, hence never waits on vmcnt there).
    const int cu_slot = (blockIdx.x >> 3) & 31;
    const uint32_t scrap = __builtin_amdgcn_readfirstlane(lds_addr(hx) + R_HX - 256);
    auto l2_touch = [&](int blk) {             // lines [12 * cu_slot, + 12) of the block's 384
        if (lane < 12) {
            const char *g = (const char *)a.wof + (int64_t)blk * F_SLOT + (12 * cu_slot + lane) * 128;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(scrap) : "memory", "m0");
        }
    };
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int64_t t0 = (int64_t)tile * TILE_TOK + vwave * 16;
    const uint16_t *xbase = a.x16 + t0 * F_H;                          // this wave's own 16 token rows
    const uint32_t lrow = (uint32_t)(n * F_H);
    uint4 xs0[G_KS];                                                   // own tokens: B operands, 12 K-steps of 32 features
    {
        const uint16_t *cbase = a.ctx + t0 * F_H;
#pragma unroll
        for (int s = 0; s < G_KS; s++) xs0[s] = *(const uint4 *)(cbase + (lrow + 8 * kg) + 32 * s);      // attention output rows, for now
    }
    // Wo goes through the ring as TWELVE half-parts of 24 KB (two 16-row output blocks x 12 K-steps; contiguous in the
    // 6 x 48 KB layout) in FOUR 24 KB slots, three half-parts ahead: with 48 KB parts in two slots every part waited out its
    // own DMA (22.6 k cycles for 10 k of MFMA work). Wave w stages pieces [3 w, 3 w + 3) of a half-part.
    constexpr int HP_BYTES = F_SLOT / 2, HP_PPW = HP_BYTES / 1024 / NWV, NHP = 2 * NPRE;
    auto stage_hp = [&](int hp) {
#pragma unroll
        for (int i = 0; i < HP_PPW; i++) {
            const int pc = wave * HP_PPW + i;
            const char *base = (const char *)a.wof + (int64_t)hp * HP_BYTES + pc * 1024;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (hp & 3) * HP_BYTES + pc * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
        }
    };
    stage_hp(0); stage_hp(1); stage_hp(2);
    for (int blk = wave; blk < NPRE + R_PF && blk < NPRE + NC; blk += NWV) l2_touch(blk);      // the Wo parts and the first chunks
    f32x4v accY[G_OB];
#pragma unroll
    for (int ob = 0; ob < G_OB; ob++) accY[ob] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    // ---- attention output projection, token-parallel. Half-part `it` is waited for with a COUNTED vmcnt (the newer half-parts
    // stay in flight across the barrier); behind the barrier every wave is also done with half-part it - 1, whose slot takes
    // half-part it + 3. During half-parts 10 and 11 waves 0-3 stage W1 of chunk 0 (pieces 0..23 of its block) into the first
    // 24 KB of the ring (half-part 8's slot, free by then).
#pragma unroll
    for (int it = 0; it < NHP; it++) {
        // outstanding DMA pieces of this wave that were issued AFTER half-part `it`: (in-flight half-parts) x 3, + the W1 pieces
        if (it + 2 < NHP) wait_vm<2 * HP_PPW>();
        else if (it + 1 < NHP) wait_vm<HP_PPW>();          // it = 10: half-part 11 (W1 of chunk 0 is issued below, after this wait)
        else { if (wave < 4) wait_vm<PPW>(); else wait_vm<0>(); }      // it = 11: waves 0-3 have their six W1 pieces in flight
        __syncthreads();
        if (it + 3 < NHP) stage_hp(it + 3);
        if (it == NHP - 2 && wave < 4) {
#pragma unroll
            for (int i = 0; i < PPW; i++) stage_piece(NPRE, 0, wave * PPW + i);
        }
        const char *slot = ring + (it & 3) * HP_BYTES + lane * 16;
        auto off = [](int i) { return ((i & 1) * G_KS + (i >> 1)) * 1024; };     // i = 2s + j -> piece j*12 + s
        uint4 fa[2][4];
#pragma unroll
        for (int j = 0; j < 4; j++) fa[0][j] = f_frag(slot + off(j));
#pragma unroll
        for (int i0 = 0; i0 < 2 * G_KS; i0 += 4) {
            if (i0 + 4 < 2 * G_KS) {
#pragma unroll
                for (int j = 0; j < 4; j++) fa[((i0 >> 2) + 1) & 1][j] = f_frag(slot + off(i0 + 4 + j));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; j++)
                accY[2 * it + (j & 1)] = mfma16_bf16(fa[(i0 >> 2) & 1][j], xs0[(i0 + j) >> 1], accY[2 * it + (j & 1)]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vm<0>();          // (waves 0-3: W1 of chunk 0 has landed before anything below can depend on it)
    RTICK(1);
    // ---- + bo + residual (the layer's input) -> LayerNorm-1 -> bf16 -> xs0 (k_ffn384w8's code)
    {
        float sum = 0.f;
        {
            // all 24 residual fragments requested at once (48 registers: the attention output rows in xs0 are dead by now).
            // Four at a time, one group ahead, every group sat through its own memory round trip: 18.6 k cycles for this
            // section where the arithmetic needs ~3 k.
            uint2 rq[G_OB];
#pragma unroll
            for (int ob = 0; ob < G_OB; ob++) rq[ob] = *(const uint2 *)(xbase + (lrow + 4 * kg) + 16 * ob);
#pragma unroll
            for (int o0 = 0; o0 < G_OB; o0 += 4) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int ob = o0 + j;
                    const uint2 rr = rq[ob];
                    const float4 bo = *(const float4 *)(s_bo + 16 * ob + 4 * kg);
                    f32x4v &v = accY[ob];
                    v[0] += bo.x + bf16_to_f32((uint16_t)rr.x);
                    v[1] += bo.y + bf16_to_f32((uint16_t)(rr.x >> 16));
                    v[2] += bo.z + bf16_to_f32((uint16_t)rr.y);
                    v[3] += bo.w + bf16_to_f32((uint16_t)(rr.y >> 16));
                    sum += (v[0] + v[1]) + (v[2] + v[3]);
                }
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mu1 = sum * (1.0f / F_H);
        float sq1 = 0.f;
#pragma unroll
        for (int ob = 0; ob < G_OB; ob++)
#pragma unroll
            for (int e = 0; e < 4; e++) { const float d = accY[ob][e] - mu1; sq1 += d * d; }
        sq1 += __shfl_xor(sq1, 16);
        sq1 += __shfl_xor(sq1, 32);
        const float rstd1 = 1.0f / sqrtf(sq1 * (1.0f / F_H) + a.eps);
        const int srcA = (n + 16 * (2 * (kg & 1))) << 2, srcB = srcA + (16 << 2);
        const bool hi = (kg >> 1) != 0;
#pragma unroll
        for (int s_ = 0; s_ < G_KS; s_++) {
            uint2 zp[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int ob = 2 * s_ + u, f = 16 * ob + 4 * kg;
                const float4 gg = *(const float4 *)(s_g1 + f), bt = *(const float4 *)(s_be1 + f);
                const f32x4v &v = accY[ob];
                const f32x4 y = {(v[0] - mu1) * rstd1 * gg.x + bt.x, (v[1] - mu1) * rstd1 * gg.y + bt.y,
                                 (v[2] - mu1) * rstd1 * gg.z + bt.z, (v[3] - mu1) * rstd1 * gg.w + bt.w};
                zp[u] = f_cvt4(y);
            }
            const uint32_t a0x = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[0].x), a0y = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[0].y);
            const uint32_t b0x = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[0].x), b0y = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[0].y);
            const uint32_t a1x = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[1].x), a1y = __builtin_amdgcn_ds_bpermute(srcA, (int)zp[1].y);
            const uint32_t b1x = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[1].x), b1y = __builtin_amdgcn_ds_bpermute(srcB, (int)zp[1].y);
            xs0[s_] = {hi ? a1x : a0x, hi ? a1y : a0y, hi ? b1x : b0x, hi ? b1y : b0y};
        }
    }
    RTICK(2);
    // exchange areas (the ring is free at both points): X rows of a pair's consumer / producer tokens, 12 KB per pair; the
    // consumer's accumulators of the producer's tokens, 12 KB per pair and half
    char *xex = ring + F_SLOT + pair * (12 * 1024) + lane * 16;        // slot 1 before the loop (slot 0 receives W1 of chunk 0)
    char *xback = ring + pair * (12 * 1024) + lane * 16;
    char *yex = ring + F_SLOT + pair * (12 * 1024) + lane * 16;
    long long l0 = 0, l1 = 0, l2 = 0, lt = 0;                          // DBG: per-role loop stamps
#define LTICK(acc) do { if constexpr (DBG) { const long long now_ = (long long)__builtin_readcyclecounter(); acc += now_ - lt; lt = now_; } } while (0)

    if (role == 0) {
        // =============================== PRODUCER ===============================
        uint4 xs1[G_KS];                                               // the consumer's 16 rows
        __syncthreads();                                               // slot 1: every wave is out of the last Wo part
        __syncthreads();                                               // the consumer's rows are in LDS
#pragma unroll
        for (int s = 0; s < G_KS; s++) xs1[s] = *(const uint4 *)(xex + s * 1024);
        __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0): read before iteration 0 stages over slot 1
        RTICK(3);
        if constexpr (DBG) lt = (long long)__builtin_readcyclecounter();      // stamps: phase A, GELU + publish, wait + barrier
        for (int c = 0; c <= NC; c++) {
            __syncthreads();                                           // (no vmcnt wait: a producer issues no ring DMA)
            LTICK(l2);
            if (c == NC) break;
            if (pair == 0 && c + R_PF < NC) l2_touch(NPRE + c + R_PF);
            const char *w1s = ring + (c & 1) * F_SLOT + lane * 16;
            f32x4v h[2][2];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                if constexpr (TAB) {           // table mode: the bias rides in as the accumulators' initial value
                    const float4 bi = *(const float4 *)(s_b1 + c * F_CH + 16 * rb + 4 * kg);
                    h[rb][0] = (f32x4v){bi.x, bi.y, bi.z, bi.w}; h[rb][1] = h[rb][0];
                } else { h[rb][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; h[rb][1] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
            }
            __builtin_amdgcn_s_setprio(1);
            uint4 fa[PD + 1][2];                                       // K-step s: the fragments of row blocks 0 and 1; PD steps ahead
#pragma unroll
            for (int s = 0; s < PD; s++)
#pragma unroll
                for (int rb = 0; rb < 2; rb++) fa[s][rb] = f_frag(w1s + (rb * G_KS + s) * 1024);
#pragma unroll
            for (int s = 0; s < G_KS; s++) {
                if (s + PD < G_KS && !(ABL & 16)) {
#pragma unroll
                    for (int rb = 0; rb < 2; rb++) fa[(s + PD) % (PD + 1)][rb] = f_frag(w1s + (rb * G_KS + s + PD) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    if constexpr ((ABL & 4) != 0) continue;
                    h[rb][0] = mfma16_bf16(fa[s % (PD + 1)][rb], xs0[s], h[rb][0]);
                    h[rb][1] = mfma16_bf16(fa[s % (PD + 1)][rb], xs1[s], h[rb][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            if constexpr (DBG) { asm volatile("" :: "v"(h[0][0]), "v"(h[0][1]), "v"(h[1][0]), "v"(h[1][1])); }
            LTICK(l0);
            // (bias +) GELU -> the two B operands of phase B (token block tb: features [row block 0 | row block 1])
            uint2 g[2][2];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                float4 bi = {0.f, 0.f, 0.f, 0.f};
                if constexpr (!TAB) bi = *(const float4 *)(s_b1 + c * F_CH + 16 * rb + 4 * kg);
#pragma unroll
                for (int tb = 0; tb < 2; tb++) {
                    if constexpr ((ABL & 2) != 0) g[rb][tb] = f_cvt4(h[rb][tb]);
                    else if constexpr (TAB) g[rb][tb] = f_gelu_tab4(h[rb][tb]);
                    else {
                        const f32x4 v = {h[rb][tb][0] + bi.x, h[rb][tb][1] + bi.y, h[rb][tb][2] + bi.z, h[rb][tb][3] + bi.w};
                        g[rb][tb] = f_cvt4(f_gelu4(v));
                    }
                }
            }
            char *hw = hx + ((c & 1) * 4 + pair) * 2048 + lane * 16;
            *(uint4 *)hw = uint4{g[0][0].x, g[0][0].y, g[1][0].x, g[1][0].y};
            *(uint4 *)(hw + 1024) = uint4{g[0][1].x, g[0][1].y, g[1][1].x, g[1][1].y};
            LTICK(l1);
        }
        RTICK(4);
        // ---- hand the consumer its rows back, take the output accumulators of the own tokens (two halves)
        __syncthreads();                                               // every wave is out of the loop: ring and exchange free
#pragma unroll
        for (int s = 0; s < G_KS; s++) *(uint4 *)(xback + s * 1024) = xs1[s];
        __syncthreads();                                               // rows + first half of Y are in LDS
#pragma unroll
        for (int j = 0; j < G_KS; j++) accY[j] = *(const f32x4v *)(yex + j * 1024);
        __syncthreads();                                               // both read: the consumer writes the second half
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G_KS; j++) accY[G_KS + j] = *(const f32x4v *)(yex + j * 1024);
    } else {
        // =============================== CONSUMER ===============================
        __syncthreads();                                               // slot 1: every wave is out of the last Wo part
#pragma unroll
        for (int s = 0; s < G_KS; s++) *(uint4 *)(xex + s * 1024) = xs0[s];
        __syncthreads();
        RTICK(3);
        f32x4v acc[2][G_OB];                                           // [token block: 0 = the producer's, 1 = own][out block]
#pragma unroll
        for (int ob = 0; ob < G_OB; ob++) { acc[0][ob] = (f32x4v){0.f, 0.f, 0.f, 0.f}; acc[1][ob] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
        // iteration c stages 48 pieces: q < 24 -> W1 of chunk c + 1, else W2 of chunk c (chunk x lives in slot x & 1; a piece keeps
        // its offset inside the block); consumer wave j takes pieces [12 j, 12 j + 12): waves 0, 1 all W1, waves 2, 3 all W2
        const int wq0 = NCP * pair, st_ahead = wq0 < 24 ? 1 : 0;
        if constexpr (DBG) lt = (long long)__builtin_readcyclecounter();      // stamps: staging, phase B, wait + barrier
        for (int c = 0; c <= NC; c++) {
            wait_vm<0>();
            __syncthreads();
            LTICK(l2);
            const int sch = c + st_ahead;
            if (sch < NC && !(ABL & 1)) {
#pragma unroll
                for (int i = 0; i < NCP; i++) stage_piece(NPRE + sch, sch & 1, wq0 + i);
            }
            LTICK(l0);
            if (c >= 1) {
                const int cc = c - 1;
                const char *hr = hx + ((cc & 1) * 4 + pair) * 2048 + lane * 16;
                const uint4 hb0 = *(const uint4 *)hr, hb1 = *(const uint4 *)(hr + 1024);
                const char *w2s = ring + (cc & 1) * F_SLOT + F_W1_BYTES + lane * 16;
                uint4 fb[CD + 1][2];                                   // two fragments (four MFMAs) per group, CD groups ahead
#pragma unroll
                for (int g = 0; g < CD; g++)
#pragma unroll
                    for (int j = 0; j < 2; j++) fb[g][j] = f_frag(w2s + (2 * g + j) * 1024);
#pragma unroll
                for (int o0 = 0; o0 < G_OB; o0 += 2) {
                    if (o0 + 2 * CD < G_OB && !(ABL & 16)) {
#pragma unroll
                        for (int j = 0; j < 2; j++) fb[((o0 >> 1) + CD) % (CD + 1)][j] = f_frag(w2s + (o0 + 2 * CD + j) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if constexpr ((ABL & 8) != 0) continue;
                        acc[0][o0 + j] = mfma16_bf16(fb[(o0 >> 1) % (CD + 1)][j], hb0, acc[0][o0 + j]);
                        acc[1][o0 + j] = mfma16_bf16(fb[(o0 >> 1) % (CD + 1)][j], hb1, acc[1][o0 + j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (DBG) { asm volatile("" :: "v"(acc[0][G_OB - 1]), "v"(acc[1][G_OB - 1])); }
            LTICK(l1);
        }
        RTICK(4);
        __syncthreads();                                               // every wave is out of the loop
#pragma unroll
        for (int j = 0; j < G_KS; j++) *(f32x4v *)(yex + j * 1024) = acc[0][j];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < G_KS; s++) xs0[s] = *(const uint4 *)(xback + s * 1024);
        __syncthreads();                                               // the producer has read the first half
#pragma unroll
        for (int j = 0; j < G_KS; j++) *(f32x4v *)(yex + j * 1024) = acc[0][G_KS + j];
        __syncthreads();
#pragma unroll
        for (int ob = 0; ob < G_OB; ob++) accY[ob] = acc[1][ob];
    }
    if constexpr (DBG) { if (a.dbg && lane == 0) { long long *d = a.dbg + ((size_t)(gridDim.x + blockIdx.x) * NWV + wave) * 6; d[0] = l0; d[1] = l1; d[2] = l2; } }
#undef LTICK
    // ---- epilogue (k_ffn384w8's): + b2 + residual out of the X registers -> LayerNorm-2 -> bf16 rows, in place
    __syncthreads();                                       // the exchange areas are read: per-wave output scratch below
    float sum = 0.f;
    const int src_lo = n + 16 * (kg >> 1);
#pragma unroll
    for (int ob = 0; ob < G_OB; ob++) {
        const uint4 own = xs0[ob >> 1];
        const int src = (src_lo + 32 * (ob & 1)) << 2;
        const uint32_t g0 = __builtin_amdgcn_ds_bpermute(src, (int)own.x), g1 = __builtin_amdgcn_ds_bpermute(src, (int)own.y),
                       g2 = __builtin_amdgcn_ds_bpermute(src, (int)own.z), g3 = __builtin_amdgcn_ds_bpermute(src, (int)own.w);
        const uint32_t w0 = (kg & 1) ? g2 : g0, w1 = (kg & 1) ? g3 : g1;
        const float4 b2 = *(const float4 *)(s_b2 + 16 * ob + 4 * kg);
        f32x4v &v = accY[ob];
        v[0] += b2.x + bf16_to_f32((uint16_t)w0);
        v[1] += b2.y + bf16_to_f32((uint16_t)(w0 >> 16));
        v[2] += b2.z + bf16_to_f32((uint16_t)w1);
        v[3] += b2.w + bf16_to_f32((uint16_t)(w1 >> 16));
        sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mu = sum * (1.0f / F_H);
    float sq = 0.f;
#pragma unroll
    for (int ob = 0; ob < G_OB; ob++)
#pragma unroll
        for (int e = 0; e < 4; e++) { const float d = accY[ob][e] - mu; sq += d * d; }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / F_H) + a.eps);
    constexpr int ROWP = F_H * 2 + 16;
    char *scr = ring + wave * (16 * ROWP);
#pragma unroll
    for (int ob = 0; ob < G_OB; ob++) {
        const int f = 16 * ob + 4 * kg;
        const float4 gg = *(const float4 *)(s_g + f), bt = *(const float4 *)(s_be + f);
        const f32x4v &v = accY[ob];
        const f32x4 y = {(v[0] - mu) * rstd * gg.x + bt.x, (v[1] - mu) * rstd * gg.y + bt.y,
                         (v[2] - mu) * rstd * gg.z + bt.z, (v[3] - mu) * rstd * gg.w + bt.w};
        *(uint2 *)(scr + n * ROWP + f * 2) = f_cvt4(y);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    RTICK(5);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int idx = i * 64 + lane, tk = idx / 48, ch = idx % 48;
        const uint4 yo = *(const uint4 *)(scr + tk * ROWP + ch * 16);
        *(uint4 *)((uint16_t *)xbase + (uint32_t)(tk * F_H + ch * 8)) = yo;
    }
    if constexpr (DBG) {
        wait_vm<0>();
        tq[6] = (long long)__builtin_readcyclecounter();
        if (a.dbg && lane == 0) {
            long long *d = a.dbg + ((size_t)blockIdx.x * NWV + wave) * 6;
            for (int i = 0; i < 6; i++) d[i] = tq[i + 1] - tq[i];
        }
    }
#undef RTICK
}

bool ffn_fused_supported(int H, int I, int64_t T) {
    return H == F_H && I % F_CH == 0 && I <= F_MAXI && T % F_TOK == 0 && I / F_CH >= 2;
}
size_t ffn_wo_bytes() { return (size_t)G_WO_PARTS * F_SLOT; }
size_t ffn_weight_bytes(int I) { return ffn_wo_bytes() + (size_t)(I / F_CH) * F_SLOT; }

// AK_FFN_W8=0 selects the 4-wave kernel (A/B); read once, at the first encoder creation
static int ffn_variant() {
    static const int v = dbg_env_int("AK_FFN_W8", 1);        // (the 4-wave generation lives in the dbg library)
    return v;
}

// AK_FFN_ATT=0: keep the attention output projection in its own launch (gemm_ln.hip) -- A/B
bool ffn_fuses_attention_out() {
    static const int v = dbg_env_int("AK_FFN_ATT", 1);       // (the unfused launch sequence: dbg library)
    return v != 0 && ffn_variant() != 0;
}

int ffn_relayout(const uint16_t *wo, const uint16_t *w1, const uint16_t *w2, int I, uint16_t *wbuf, const uint16_t **wf_out, hipStream_t st) {
    if (gelu_table_create()) return -10;
    uint16_t *wf = wbuf + ffn_wo_bytes() / 2;
    const int64_t units = (int64_t)(I / F_CH) * (F_SLOT / 16);
    if (ffn_variant()) k_ffn_relayout16<<<(unsigned)((units + 255) / 256), 256, 0, st>>>(w1, w2, I, wf);
    else k_ffn_relayout<<<(unsigned)((units + 255) / 256), 256, 0, st>>>(w1, w2, I, wf);       // (variant 0: dbg library)
    const int64_t wunits = (int64_t)G_WO_PARTS * (F_SLOT / 16);
    k_wo_relayout16<<<(unsigned)((wunits + 255) / 256), 256, 0, st>>>(wo, wbuf);
    AK_HIP(hipGetLastError());
    *wf_out = wf;
    return 0;
}

// =====================================================================================================================
// QKV projection for hidden 384 in the same mould (replaces the k_gemm<0> launch: 93 us per MiniLM layer, the K = 384 loop
// as long as its split / transpose epilogue): X stays in registers as the B operand (16 tokens per wave), the [1152][384]
// weight matrix streams through the three-slot ring in fragment order, 18 blocks of 64 output features (48 KB, the Wo
// part format), 48 MFMAs per wave and block; every block is finished when its K-steps are -- bias, the softmax scale on
// Q, bf16 -- and is stored AFTER the next block's barrier, in front of that block's DMA issue, so that the counted
// vmcnt wait of the ring never waits for a store just issued.
//   Q / K blocks: A = W rows (features), B = X -> lane (token n, kg) holds features; the rows of a block are permuted at
//     relayout time (qkv_row) so that a lane's 16 values are two runs of 8 consecutive features: 2 x 16-byte stores, a
//     token's 64 features = 2 x 64 contiguous bytes per store instruction.
//   V blocks: operands swapped (A = X, B = W rows) -> lane (feature m, kg) holds 4 consecutive tokens: the transposed
//     [B][H][S] layout is written 8 bytes per lane at the V^T position of its token group (vt_pos).
// Measured and not kept: the projection and the attention of one (sequence, head) in ONE persistent workgroup (x rows of the wave's
// 32 tokens in 96 registers, the head's [96][384] weights through the ring, q / k / v^T never in HBM; bit-for-bit the same
// operand layouts -- a lane's 16 accumulator values ARE its two q fragments, its key's K row chunks, its feature's V^T chunks).
// Parity-green, but 155-175 us per layer against 72 + 63 for the two launches: per item and wave 5.4 k cycles projection (at the
// MFMA rate), 1.2 k pack + LDS writes, 4.2 k ring waits + barriers, and 14-20 k attention -- with 256 registers per wave there are
// two waves per SIMD, and the softmax chain (LDS read -> MFMA -> max -> exchange -> exp -> pack -> MFMA) that six waves per SIMD
// cover in k_attn runs exposed; software-pipelining the score MFMAs one chunk ahead cost more in registers than it hid.
// =====================================================================================================================
constexpr int Q_NB = 3 * F_H / 64;      // 18 blocks of 64 output features
// feature (within its block of 64) held in row m of tile j of a Q / K block
__host__ __device__ inline int qkv_row(int j, int m) { return (j < 2 ? 0 : 32) + 8 * (m >> 2) + 4 * (j & 1) + (m & 3); }

__global__ void k_qkv_relayout16(const uint16_t *__restrict__ wqkv, const float *__restrict__ bqkv, uint16_t *__restrict__ wq16,
                                 float *__restrict__ bq16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte unit each
    if (i < 3 * F_H) {
        const int blk = (int)i / 64, u = (int)i % 64, j = u >> 4, m = u & 15;
        bq16[i] = bqkv[64 * blk + (blk < 2 * F_H / 64 ? qkv_row(j, m) : 16 * j + m)];
    }
    if (i >= (int64_t)Q_NB * (F_SLOT / 16)) return;
    const int blk = (int)(i / (F_SLOT / 16)), u = (int)(i % (F_SLOT / 16));
    const int j = u / (12 * 64), s_ = (u / 64) % 12, l = u % 64, m = l & 15, kg = l >> 4;
    const int row = 64 * blk + (blk < 2 * F_H / 64 ? qkv_row(j, m) : 16 * j + m);
    const uint16_t *src = wqkv + (int64_t)row * F_H + 32 * s_ + 8 * kg;
    uint4 o;
    o.x = src[0] | ((uint32_t)src[1] << 16); o.y = src[2] | ((uint32_t)src[3] << 16);
    o.z = src[4] | ((uint32_t)src[5] << 16); o.w = src[6] | ((uint32_t)src[7] << 16);
    *(uint4 *)(wq16 + i * 8) = o;
}

constexpr int Q_PARAM_BYTES = 3 * F_H * 4 + 512;      // the permuted bias (+ padding to a 1 KiB multiple)
constexpr int Q_LDS = Q_PARAM_BYTES + F_NST * F_SLOT;

template <int TG>      // 16-token groups per wave: every weight fragment read from LDS feeds TG MFMAs
__global__ __launch_bounds__(G_THREADS8, 1) void k_qkv384(QkvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_bias = (float *)smem;
    char *ring = smem + Q_PARAM_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    for (int i = tid; i < 3 * F_H; i += G_THREADS8) s_bias[i] = a.bias[i];
    const uint32_t lds0 = lds_addr(ring);
    const uint32_t voff = (uint32_t)lane * 16;
    const char *src_wave = (const char *)a.w + (wave * G_PPW) * 1024;
    auto stage_piece = [&](int it, int i) {
        const char *base = src_wave + (int64_t)it * F_SLOT + i * 1024;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (it % F_NST) * F_SLOT + (wave * G_PPW + i) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
    };
    const int tile = blockIdx.x;
    const int64_t t0 = (int64_t)tile * (F_TOK * TG) + wave * (16 * TG);
    // Which token lane n of group g holds. TG = 1: token n. TG = 2: the 32 tokens are dealt so that in a V block (lane =
    // feature, 4 tokens per group) a lane's 8 tokens sit at 8 CONSECUTIVE V^T positions 8kg' + 4g + i (kg' = n >> 2 here, the
    // C-row group there): token = vt_pos(8 (n >> 2) + 4 g + (n & 3)), vt_pos being its own inverse -- one 16-byte store
    // per lane and feature row instead of two 8-byte ones.
    auto tok = [&](int g) { return TG == 1 ? n : ((n & 3) | (((n >> 2) & 1) << 2) | (g << 3) | ((n >> 3) << 4)); };
    const uint16_t *xbase = a.x16 + t0 * F_H;
    uint4 xb[TG][G_KS];
#pragma unroll
    for (int g = 0; g < TG; g++)
#pragma unroll
        for (int s = 0; s < G_KS; s++) xb[g][s] = *(const uint4 *)(xbase + (uint32_t)(tok(g) * F_H + 8 * kg) + 32 * s);
#pragma unroll
    for (int i = 0; i < G_PPW; i++) stage_piece(0, i);
#pragma unroll
    for (int i = 0; i < G_PPW; i++) stage_piece(1, i);
    // A block's life: K-steps in its own ring iteration; bias / scale / bf16 UNDER the next block's MFMAs (between the K-step
    // groups); stores behind the barrier after that, in front of that block's DMA issue -- the counted vmcnt wait of the
    // ring then never waits for a store just issued, and the matrix pipe does not idle through an epilogue
    // (AK_QKV_DBG ablations, 256 x 256 tokens: 72 us per launch; without the stores 52; without the ring loads 63; without
    // both 51; without the fragment reads as well 46 = 1.26 PF, the part's sustained MFMA rate: what is left above that is
    // the 150 MB of Q / K / V^T this launch has to write: letting the stores stay in flight across the ring's counted wait
    // (vmcnt(6 + stores)) changes nothing, 82.0 vs 82.2 us on one box -- it is the write path, not the wait.)
    uint4 pend[TG][2];               // packed block waiting for its stores: per token group tiles {0, 1} and {2, 3}
    f32x4v accp[TG][4];              // finished accumulators waiting to be packed
    const int vb = (int)(t0 / a.S), vs0 = (int)(t0 - (int64_t)vb * a.S);      // 16 TG tokens never straddle a sequence (S % 32 == 0)
    const bool vlive = t0 < a.T;                                     // rows past the last real token have no V^T slot
    const int vkg = TG == 1 ? (((kg & 1) << 3) | ((kg & 2) << 1)) : 8 * kg;    // V^T position of this lane's first token

    auto flush = [&](int it) {          // stores of block `it` (wave-uniform kind)
        if (it < 0 || (a.dbg & 1)) return;
        if (it < 2 * F_H / 64) {
            if (a.head_major) {
                // block `it` = heads 2 i and 2 i + 1 (i = it % 6) of this wave's 16 TG tokens: [B][12][S][32], 64 bytes per
                // token and head -> the wave writes 1 KiB TG runs, and the attention kernel stages whole cache lines
                if (!vlive) return;
                uint16_t *dst = (it < F_H / 64 ? a.q : a.k) + (int64_t)vb * a.S * F_H + (int64_t)(2 * (it % (F_H / 64))) * a.S * 32 +
                                (int64_t)vs0 * 32 + 8 * kg;
#pragma unroll
                for (int g = 0; g < TG; g++) {
                    *(uint4 *)(dst + (uint32_t)(tok(g) * 32)) = pend[g][0];
                    *(uint4 *)(dst + (uint32_t)(a.S * 32 + tok(g) * 32)) = pend[g][1];
                }
                return;
            }
            uint16_t *dst = (it < F_H / 64 ? a.q : a.k) + t0 * F_H + 64 * (it % (F_H / 64)) + 8 * kg;
#pragma unroll
            for (int g = 0; g < TG; g++) {
                *(uint4 *)(dst + (uint32_t)(tok(g) * F_H)) = pend[g][0];
                *(uint4 *)(dst + (uint32_t)(tok(g) * F_H) + 32) = pend[g][1];
            }
        } else if (vlive) {
            uint16_t *dst = a.vt + ((int64_t)vb * F_H + 64 * (it - 2 * F_H / 64) + n) * a.S + vs0 + vkg;
            if (TG == 1) {
                *(uint2 *)dst = uint2{pend[0][0].x, pend[0][0].y};
                *(uint2 *)(dst + 16 * (int64_t)a.S) = uint2{pend[0][0].z, pend[0][0].w};
                *(uint2 *)(dst + 32 * (int64_t)a.S) = uint2{pend[0][1].x, pend[0][1].y};
                *(uint2 *)(dst + 48 * (int64_t)a.S) = uint2{pend[0][1].z, pend[0][1].w};
            } else {
                *(uint4 *)dst = uint4{pend[0][0].x, pend[0][0].y, pend[TG - 1][0].x, pend[TG - 1][0].y};
                *(uint4 *)(dst + 16 * (int64_t)a.S) = uint4{pend[0][0].z, pend[0][0].w, pend[TG - 1][0].z, pend[TG - 1][0].w};
                *(uint4 *)(dst + 32 * (int64_t)a.S) = uint4{pend[0][1].x, pend[0][1].y, pend[TG - 1][1].x, pend[TG - 1][1].y};
                *(uint4 *)(dst + 48 * (int64_t)a.S) = uint4{pend[0][1].z, pend[0][1].w, pend[TG - 1][1].z, pend[TG - 1][1].w};
            }
        }
    };
    // bias, scale, bf16 for tile j of token group g of block pb (accumulators in accp) -> two words of pend
    auto pack = [&](int pb, int g, int j, auto vtag) {
        constexpr bool ISV = decltype(vtag)::value;
        uint32_t w0, w1;
        const f32x4v v = accp[g][j];
        if (ISV) {      // lane = feature 16j + n, values = this lane's 4 tokens of group g
            const float bi = s_bias[64 * pb + 16 * j + n];
            w0 = pack_bf16x2(v[0] + bi, v[1] + bi);
            w1 = pack_bf16x2(v[2] + bi, v[3] + bi);
        } else {        // lane = one token, values = rows 4kg .. 4kg+3 of tile j = features qkv_row(j, 4kg ..)
            const float sc = pb < F_H / 64 ? a.qscale : 1.0f;
            const float4 bi = *(const float4 *)(s_bias + 64 * pb + 16 * j + 4 * kg);
            w0 = pack_bf16x2((v[0] + bi.x) * sc, (v[1] + bi.y) * sc);
            w1 = pack_bf16x2((v[2] + bi.z) * sc, (v[3] + bi.w) * sc);
        }
        uint4 &d = pend[g][j >> 1];
        if (j & 1) { d.z = w0; d.w = w1; } else { d.x = w0; d.y = w1; }
    };
    auto block = [&](int it, auto vtag, auto ptag, auto pvtag) {
        constexpr bool ISV = decltype(vtag)::value, HAS_PREV = decltype(ptag)::value;
        if (it + 1 < Q_NB) wait_vm<G_PPW>(); else wait_vm<0>();
        __syncthreads();
        flush(it - 2);
        const bool more = it + 2 < Q_NB;
        const char *slot = ring + (it % F_NST) * F_SLOT + lane * 16;
        auto off = [](int i) { return ((i & 3) * G_KS + (i >> 2)) * 1024; };     // i = 4s + j -> piece j*12 + s
        f32x4v acc[TG][4];
#pragma unroll
        for (int g = 0; g < TG; g++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[g][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        uint4 fa[2][4];
#pragma unroll
        for (int j = 0; j < 4; j++) fa[0][j] = f_frag(slot + off(j));
#pragma unroll
        for (int i0 = 0; i0 < 4 * G_KS; i0 += 4) {
            if (i0 + 4 < 4 * G_KS) {
#pragma unroll
                for (int j = 0; j < 4; j++) fa[((i0 >> 2) + 1) & 1][j] = f_frag(slot + off(i0 + 4 + j));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < TG; g++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[g][j] = ISV ? mfma16_bf16(xb[g][i0 >> 2], fa[(i0 >> 2) & 1][j], acc[g][j])
                                    : mfma16_bf16(fa[(i0 >> 2) & 1][j], xb[g][i0 >> 2], acc[g][j]);
            if (more && (i0 >> 2) < G_PPW && !(a.dbg & 2)) stage_piece(it + 2, i0 >> 2);
            if (HAS_PREV && (i0 >> 2) >= 2 && (i0 >> 2) - 2 < 4 * TG) pack(it - 1, ((i0 >> 2) - 2) >> 2, ((i0 >> 2) - 2) & 3, pvtag);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < TG; g++)
#pragma unroll
            for (int j = 0; j < 4; j++) accp[g][j] = acc[g][j];
    };
    __syncthreads();        // the bias
    constexpr std::false_type F{};
    constexpr std::true_type T{};
    block(0, F, F, F);
    for (int it = 1; it < 2 * F_H / 64; it++) block(it, F, T, F);
    block(2 * F_H / 64, T, T, F);
    for (int it = 2 * F_H / 64 + 1; it < Q_NB; it++) block(it, T, T, T);
    flush(Q_NB - 2);
#pragma unroll
    for (int g = 0; g < TG; g++)
#pragma unroll
        for (int j = 0; j < 4; j++) pack(Q_NB - 1, g, j, T);
    flush(Q_NB - 1);
}

size_t qkv384_weight_bytes() { return (size_t)Q_NB * F_SLOT + 3 * F_H * 4; }
static bool qkv_gemm_forced() { static const bool v = env_get("AK_QKV_GEMM") != nullptr; return v; }
bool qkv384_supported(int H, int64_t T, int S) { return H == F_H && T % F_TOK == 0 && S % 32 == 0 && ffn_variant() != 0 && !qkv_gemm_forced(); }
// wbuf: qkv384_weight_bytes() bytes: [18 blocks of 48 KB | permuted bias]
int qkv384_relayout(const uint16_t *wqkv, const float *bqkv, uint16_t *wbuf, hipStream_t st) {
    const int64_t units = (int64_t)Q_NB * (F_SLOT / 16);
    k_qkv_relayout16<<<(unsigned)((units + 255) / 256), 256, 0, st>>>(wqkv, bqkv, wbuf, (float *)((char *)wbuf + (size_t)Q_NB * F_SLOT));
    AK_HIP(hipGetLastError());
    return 0;
}
int launch_qkv384(const QkvArgs &a0, hipStream_t st) {
    static std::atomic<bool> attr{false};      // (set twice by two first callers at worst: idempotent)
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_qkv384<1>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_qkv384<2>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS));
        attr = true;
    }
    QkvArgs a = a0;
    static const int qdbg = dbg_env_int("AK_QKV_DBG", 0);   // measurement only (wrong results): 1 no stores, 2 no ring loads
    a.dbg = qdbg;
    a.bias = (const float *)((const char *)a.w + (size_t)Q_NB * F_SLOT);
    // 32 tokens per wave (256 per workgroup) when the token count allows: each 1 KB weight fragment read from LDS then feeds
    // two MFMAs -- at 16 tokens per wave the kernel asks LDS for 256 B per clock and CU, its whole bandwidth
    static const int tg_force = env_get("AK_QKV_TG") ? atoi(env_get("AK_QKV_TG")) : 0;
    // ... but only once there are more 128-token tiles than CUs: below that the 16-token form puts twice the workgroups on
    // the chip (forward ms, 32 / 16 tokens per wave: 8192 tokens 0.96 / 0.85, 16 384 1.02 / 0.93, 32 768 1.24 / 1.19, 65 536 2.18 / 2.23)
    const int tg = tg_force ? tg_force : ((a.Tpad % (2 * F_TOK) == 0 && a.Tpad / F_TOK > 256) ? 2 : 1);
    if (tg == 2) k_qkv384<2><<<a.Tpad / (2 * F_TOK), G_THREADS8, Q_LDS, st>>>(a);
    else k_qkv384<1><<<a.Tpad / F_TOK, G_THREADS8, Q_LDS, st>>>(a);
    AK_HIP(hipGetLastError());
    return 0;
}

int launch_ffn384(const FfnArgs &a, hipStream_t st) {
    static std::atomic<bool> attr{false};      // (set twice by two first callers at worst: idempotent)
    if (!attr) {
#if AK_DBG_KERNELS
        AK_HIP(hipFuncSetAttribute((const void *)k_ffn384, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_ffn384w8<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_ffn384w8<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_ffn384w8<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
#endif
        AK_HIP(hipFuncSetAttribute((const void *)k_ffn384w8<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_ffn384p<false>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS));
        if constexpr (DBG_KERNELS) AK_HIP(hipFuncSetAttribute((const void *)k_ffn384p<true>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS));
#define R_ATTR(...) AK_HIP(hipFuncSetAttribute((const void *)k_ffn384r<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS))
        R_ATTR(false, true); R_ATTR(false, false);
        if constexpr (DBG_KERNELS) {
            R_ATTR(true, true); R_ATTR(true, false);
            R_ATTR(true, false, 1); R_ATTR(true, false, 2); R_ATTR(true, false, 4); R_ATTR(true, false, 8); R_ATTR(true, false, 16); R_ATTR(true, false, 3); R_ATTR(true, false, 12); R_ATTR(true, false, 15);
        }
#undef R_ATTR
        attr = true;
    }
    const int ntiles = a.T / F_TOK;
    const int w8 = ffn_variant();
    const int grid = w8 ? ntiles : (ntiles < 256 ? ntiles : 256);      // 8-wave kernel: one tile per workgroup
    // 64-token tiles (4 waves) while they fit the CUs in one round (AK_FFN_NWV=4 / 8 forces). Forward ms, 8 / 4 waves: 6144 tokens
    // 0.83 / 0.71, 8192 0.85 / 0.73, 16 384 0.91 / 0.82, 24 576 1.04 / 1.26, 32 768 1.14 / 1.36, 65 536 2.07 / 2.50
    static const int nwv_force = env_get("AK_FFN_NWV") ? atoi(env_get("AK_FFN_NWV")) : 0;
    const bool half_tiles = w8 && (nwv_force ? nwv_force == 4 : 2 * ntiles <= 256);
    FfnArgs b = a;
    b.gelu_tab = g_gelu_tab;

    static long long *dbg = nullptr;
    static const bool ffn_dbg = env_get("AK_FFN_DBG") != nullptr;
    static const int ffn_ablate = dbg_env_int("AK_FFN_ABLATE", 0);
    if (ffn_dbg) {
        if constexpr (!DBG_KERNELS) AK_FAIL(-1, "AK_FFN_DBG needs libarchi_hip_dbg.so (make -C archi_amd/csrc dbg): the product library carries no instrumented layer kernels");
        if (!dbg) AK_HIP(hipMalloc((void **)&dbg, 4096 * 8 * 6 * 8));
        b.dbg = dbg;
        b.ablate = ffn_ablate;
    } else b.dbg = nullptr;
    static const int pair = dbg_env_int("AK_FFN_PAIR", 1);      // A/B: 0 = k_ffn384w8 on full tiles (dbg library)
    // A/B: AK_FFN_ROLE=0 = the wave-pair kernel k_ffn384p; AK_FFN_GELU=poly = the role kernel with the polynomial GELU (bit-identical to
    // k_ffn384p / k_ffn384w8)
    static const int rolek = env_get("AK_FFN_ROLE") ? atoi(env_get("AK_FFN_ROLE")) : 1;
    static const bool gtab = !(env_get("AK_FFN_GELU") && !strcmp(env_get("AK_FFN_GELU"), "poly"));
    if (a.ctx) {
        if (!w8) AK_FAIL(-1, "launch_ffn384: the fused attention output projection needs the 8-wave kernel");
        if ((const char *)a.wf != (const char *)a.wof + ffn_wo_bytes()) AK_FAIL(-1, "launch_ffn384: wof must sit directly in front of wf");
        if (half_tiles) k_ffn384w8<true, 4><<<2 * ntiles, 256, F_LDS, st>>>(b);
        else if (pair && rolek) {
            if (b.dbg) {
                if constexpr (DBG_KERNELS) {
                    if (b.ablate) {
#define R_AB(m) case m: k_ffn384r<true, false, m><<<grid, G_THREADS8, R_LDS, st>>>(b); break
                        switch (b.ablate) { R_AB(1); R_AB(2); R_AB(4); R_AB(8); R_AB(16); R_AB(3); R_AB(12); R_AB(15);
                            default: AK_FAIL(-1, "launch_ffn384: no instantiation for this AK_FFN_ABLATE"); }
#undef R_AB
                    } else if (gtab) k_ffn384r<true, true><<<grid, G_THREADS8, R_LDS, st>>>(b);
                    else k_ffn384r<true, false><<<grid, G_THREADS8, R_LDS, st>>>(b);
                }
            }
            else if (gtab) k_ffn384r<false, true><<<grid, G_THREADS8, R_LDS, st>>>(b);
            else k_ffn384r<false, false><<<grid, G_THREADS8, R_LDS, st>>>(b);
        }
        else if (pair && b.dbg) { if constexpr (DBG_KERNELS) k_ffn384p<true><<<grid, G_THREADS8, P_LDS, st>>>(b); }
        else if (pair) k_ffn384p<false><<<grid, G_THREADS8, P_LDS, st>>>(b);
        else {
#if AK_DBG_KERNELS
            k_ffn384w8<true><<<grid, G_THREADS8, F_LDS, st>>>(b);
#endif
        }
    } else {
        // the feed-forward block without the fused attention output projection (AK_FFN_ATT=0) and the 4-wave generation:
        // superseded A/B references, instantiated in libarchi_hip_dbg.so only
#if AK_DBG_KERNELS
        if (w8 && half_tiles) k_ffn384w8<false, 4><<<2 * ntiles, 256, F_LDS, st>>>(b);
        else if (w8) k_ffn384w8<false><<<grid, G_THREADS8, F_LDS, st>>>(b);
        else k_ffn384<<<grid, F_THREADS, F_LDS, st>>>(b);
#else
        AK_FAIL(-1, "launch_ffn384: this kernel variant is instantiated in libarchi_hip_dbg.so only");
#endif
    }
    AK_HIP(hipGetLastError());
    if (b.dbg) {    // measurement mode: synchronous read-back and a one-line report per launch
        const int nwv = w8 ? (half_tiles ? 4 : 8) : 4;
        const int ngrid = half_tiles ? 2 * ntiles : grid;
        const bool role_dbg = w8 && !half_tiles && a.ctx && pair && rolek;
        std::vector<long long> h((size_t)ngrid * nwv * 6 * (role_dbg ? 2 : 1));
        AK_HIP(hipStreamSynchronize(st));
        AK_HIP(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double s6[6] = {0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < (size_t)ngrid * nwv * 6; i++) s6[i % 6] += (double)h[i];
        const double nw = (double)ngrid * nwv;
        if (role_dbg) {
            // workgroups [0, 256) start on an idle chip with the layer's weights not in L2 yet; the others follow them
            const size_t base = (size_t)ngrid * nwv * 6;
            for (int half = 0; half < (ngrid > 256 ? 2 : 1); half++) {
                double lp[2][3] = {{0, 0, 0}, {0, 0, 0}};
                const int g0 = half ? 256 : 0, g1 = half ? ngrid : (ngrid > 256 ? 256 : ngrid);
                for (int g = g0; g < g1; g++)
                    for (int w = 0; w < nwv; w++)
                        for (int i = 0; i < 3; i++) lp[w >> 2][i] += (double)h[base + ((size_t)g * nwv + w) * 6 + i];
                const double nr = (double)(g1 - g0) * 4 * (a.I / F_CH);
                fprintf(stderr, "k_ffn384r chunk loop, workgroups [%d, %d), cycles per chunk: producer phase A %.0f, GELU + publish %.0f, wait + barrier %.0f | consumer staging %.0f, phase B %.0f, wait + barrier %.0f\n",
                        g0, g1, lp[0][0] / nr, lp[0][1] / nr, lp[0][2] / nr, lp[1][0] / nr, lp[1][1] / nr, lp[1][2] / nr);
            }
        }
        if (w8 && !half_tiles && a.ctx && pair)
            fprintf(stderr, "k_ffn384%c T=%d: per wave kcycles out-projection %.1f, LayerNorm-1 %.1f, X exchange %.1f, chunk loop %.1f, Y exchange + LayerNorm-2 %.1f, stores %.1f\n",
                    rolek ? 'r' : 'p', a.T, s6[0] / nw / 1e3, s6[1] / nw / 1e3, s6[2] / nw / 1e3, s6[3] / nw / 1e3, s6[4] / nw / 1e3, s6[5] / nw / 1e3);
        else
        fprintf(stderr, "k_ffn384%s T=%d: per wave kcycles wait+barrier %.1f, stage/drain %.1f, phase A %.1f, GELU %.1f, phase B %.1f, epilogue %.1f\n",
                w8 ? "w8" : "", a.T, s6[0] / nw / 1e3, s6[1] / nw / 1e3, s6[2] / nw / 1e3, s6[3] / nw / 1e3, s6[4] / nw / 1e3, s6[5] / nw / 1e3);
    }
    return 0;
}

}  // namespace ak

extern "C" int ak_encoder_gelu_table(uint16_t *out8192) {
    if (!out8192) AK_FAIL(-1, "ak_encoder_gelu_table: out is NULL");
    ak::gelu_table_host(out8192);
    return 0;
}
