// query_forward.hip -- the whole encoder forward pass of a FEW token rows (embed_query: one sequence of <= 64 tokens) in ONE launch.
//
// The reference's read path embeds one query per request thread (Embeddings.embed_query at
// /root/reference/src/data_manager/vectorstore/postgres_vectorstore.py:245,390 inside similarity_search, :227-248). Here that was
// 47 dependent launches -- embed, mask, 6 x (QKV, attention, out-projection, LayerNorm, FFN-up, FFN-down, LayerNorm), pooling -- of
// 4-9 us each for a few microseconds of work between them: 0.22 of embed_query's 0.25 ms was the chain's launch-to-launch latency
// (docs/EXPERIMENTS.md, "The single query"). A grid-wide cooperative kernel was considered in round 5 and not built: the rows that
// travel between phases would cross XCDs, whose L2s are not coherent with each other, so every phase would need an agent-scope
// release + acquire (an L2 write-back and an L1 invalidate, microseconds each).
//
// This kernel stays INSIDE ONE XCD (round-5 review, task 4). 256 workgroups are launched; each reads the XCC id of the CU it landed
// on (s_getreg_b32 HW_REG_XCC_ID); the first to arrive claims its XCD as the launch's home, the workgroups that landed elsewhere
// leave at once, and the P that stayed (32 when the dispatcher deals blocks round-robin -- nothing depends on that, any P >= 1
// works; participation is decided by the register, never by an assumed placement) take dense tickets and run every phase of the
// forward pass as `for (item = ticket; item < items; item += P)`, separated by barriers on a counter in that XCD's L2:
//   * stores of a phase are plain stores; a workgroup waits for its own (s_waitcnt vmcnt(0): they are in the XCD's L2 then -- the
//     vector L1 is write-through) before it arrives at the barrier; the arrival is an atomic add, which executes in the L2;
//   * everything a phase reads that ANOTHER workgroup of this launch wrote is read with device-scope (sc1) loads -- LDS-DMA
//     included --, which bypass this CU's vector L1 and are served by that same L2 (attn_d.h, LdL2). Weights are read normally.
//   No agent-scope release, no L1 invalidate, no L2 write-back: the CUs of one XCD share one coherent L2.
// Every wait is BOUNDED: a workgroup that does not see its barrier complete within ~2 s sets a failure word and leaves, the others
// follow; the host reads the word after the launch (the call synchronises its stream on this path) and re-runs the forward pass
// through the 47-launch path. A hung GPU is not a possible outcome.
//
// The arithmetic of every phase is that of the kernel it replaces -- k_mask_from_lens, k_embed, k_attn_prepare, k_gemm_skinny
// (same K split into 4 / 8 partial tiles, summed in the same order), k_attn_d (the same code: attn_d_body), k_layernorm, k_pool --
// so the rows are BIT-IDENTICAL to the 47-launch path's (tests/test_encoder_gpu.py holds that).
#include <atomic>
#include <mutex>

#include "attn_d.h"
#include "encoder_kernels.h"
#include "switches.h"

namespace ak {
using namespace mt;

namespace {

constexpr int QF_THREADS = 256, QF_GRID = 256;
constexpr unsigned QF_SPIN_MAX = 1u << 21;          // polls (with s_sleep) before a wait gives up: ~2 s
constexpr int QF_LDS = 1024 + 2048 + 8 * 1024 * 4 + 64;   // the largest phase: pooling (red | wgt | part[8][1024]) + the barrier's flag words

__device__ __forceinline__ unsigned qf_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-local barrier on a monotonic counter (never reset inside a launch): generation g completes when the counter reaches g * P.
__device__ __forceinline__ bool qf_barrier(QfCtlSlot *c, unsigned *fail, int P, unsigned &gen, int tid, unsigned *s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this thread's stores are in the L2
    __syncthreads();
    if (tid == 0) {
        gen++;
        const unsigned target = gen * (unsigned)P;
        unsigned ok = 1, spins = 0;
        // the last arriver sees the barrier complete in the value its own add returns; the others poll the L2 word with a short sleep
        // between polls (64 workgroups polling back to back made every ARRIVAL queue behind the polls: 0.30 -> 0.41 ms per forward)
        if (__hip_atomic_fetch_add(&c->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u < target) {
            while (qf_load(&c->count) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > QF_SPIN_MAX || ((spins & 1023u) == 0 && qf_load(fail) != 0)) { ok = 0; break; }
            }
        }
        if (!ok) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

// ---- phase bodies: one virtual block `vb` of the kernel named in the comment, executed by this 256-thread workgroup -------------
// k_mask_from_lens (encoder.hip): contiguous ids + 0 / 1 mask of right-padded rows
__device__ __forceinline__ void qf_mask_from_lens(const QfArgs &a, int vb, int tid) {
    const int64_t i = (int64_t)vb * 256 + tid;
    if (i >= (int64_t)a.B * a.S) return;
    const int b = (int)(i / a.S), t = (int)(i - (int64_t)b * a.S);
    int len = a.lens[(int64_t)b * a.lens_stride];
    len = len < 0 ? 0 : (len > a.S ? a.S : len);
    const bool live = t < len;
    a.oids[i] = live ? a.ids_in[(int64_t)b * a.ld_ids + t] : 0;
    a.omask[i] = live ? 1 : 0;
}

// k_embed<NP> (encoder.hip): one wave per two tokens, LN(word[id] + pos[s] + type[0])
template <int NP>
__device__ __forceinline__ void qf_embed(const QfArgs &a, const int *ids, int vb, int tid) {
    constexpr int H = NP * 128, RW = 2;
    struct __attribute__((packed, aligned(4))) Run { uint32_t w[NP]; };
    struct __attribute__((packed, aligned(8))) RunF { float2 w[NP]; };
    const int T = a.T, S = a.S;
    const int row0 = (vb * 4 + (tid >> 6)) * RW, lane = tid & 63;
    if (row0 >= T) return;
    int id[RW], rowq[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        rowq[q] = row0 + q < T ? row0 + q : T - 1;
        id[q] = LdL2::i32(ids + rowq[q]);
        if (id[q] < 0 || id[q] >= a.vocab) id[q] = 0;
    }
    Run wa[RW], wb[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        wa[q] = *(const Run *)(a.word + (int64_t)id[q] * H + lane * 2 * NP);
        wb[q] = *(const Run *)(a.pos + (int64_t)(rowq[q] % S) * H + lane * 2 * NP);
    }
    const Run c = *(const Run *)(a.type + lane * 2 * NP);
    const RunF gg = *(const RunF *)(a.eg + lane * 2 * NP), bb = *(const RunF *)(a.eb + lane * 2 * NP);
    float2 v[RW][NP];
    float s[RW], sq[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        s[q] = 0.f;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            v[q][j].x = bf16_to_f32((uint16_t)wa[q].w[j]) + bf16_to_f32((uint16_t)wb[q].w[j]) + bf16_to_f32((uint16_t)c.w[j]);
            v[q][j].y = bf16_to_f32((uint16_t)(wa[q].w[j] >> 16)) + bf16_to_f32((uint16_t)(wb[q].w[j] >> 16)) + bf16_to_f32((uint16_t)(c.w[j] >> 16));
            s[q] += v[q][j].x + v[q][j].y;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int q = 0; q < RW; q++) s[q] += __shfl_xor(s[q], off);
#pragma unroll
    for (int q = 0; q < RW; q++) {
        s[q] = s[q] / (float)H;
        sq[q] = 0.f;
#pragma unroll
        for (int j = 0; j < NP; j++) { const float d0 = v[q][j].x - s[q], d1 = v[q][j].y - s[q]; sq[q] += d0 * d0 + d1 * d1; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int q = 0; q < RW; q++) sq[q] += __shfl_xor(sq[q], off);
#pragma unroll
    for (int q = 0; q < RW; q++) {
        if (row0 + q >= T) break;
        const float mu = s[q], rstd = 1.0f / sqrtf(sq[q] / (float)H + a.eps);
        Run o16; RunF o32;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const float2 y = {(v[q][j].x - mu) * rstd * gg.w[j].x + bb.w[j].x, (v[q][j].y - mu) * rstd * gg.w[j].y + bb.w[j].y};
            o32.w[j] = y;
            o16.w[j] = pack_bf16x2(y.x, y.y);
        }
        if (a.x32) *(RunF *)(a.x32 + (int64_t)(row0 + q) * H + lane * 2 * NP) = o32;
        *(Run *)(a.x16 + (int64_t)(row0 + q) * H + lane * 2 * NP) = o16;
    }
}

// k_attn_prepare (attention.hip): additive key mask + the per-sequence bitmap of 32-key blocks (S <= 64 here: one thread per key)
__device__ __forceinline__ void qf_prepare(const QfArgs &a, const int *mask, int b, int tid, uint32_t *s_bits) {
    const int S = a.S, t = tid;
    if (t == 0) *s_bits = 0;
    __syncthreads();
    const int mv = t < S ? LdL2::i32(mask + b * S + t) : 0;
    if (t < S) a.maskf[b * S + t] = mv ? 0.f : -__builtin_inff();
    const uint64_t bal = __ballot(mv != 0);
    const uint32_t lo = (uint32_t)bal, hi = (uint32_t)(bal >> 32);
    if ((t & 63) == 0 && bal)
        atomicOr(s_bits, ((lo ? 1u : 0u) | (hi ? 2u : 0u) | (lo == 0xffffffffu ? 0x10000u : 0u) | (hi == 0xffffffffu ? 0x20000u : 0u)) << (t >> 5));
    __syncthreads();
    if (t == 0) a.blkmask[b] = *s_bits;
    __syncthreads();
}

// k_gemm_skinny<NP, EPI> (gemm_skinny.hip): one 32 x 32 output tile, K split into NP partial tiles (4, or 8 for K >= 1024) that
// meet in LDS and are summed in the order 0 .. NP - 1. Four waves here: wave w computes partials w and w + 4.
struct QfQkv { uint16_t *q, *k, *vt; int H, S, T; float qscale; };
template <int EPI>
__device__ __forceinline__ void qf_gemm_tile(const uint16_t *X, const uint16_t *W, const float *bias, int N, int K, int NP, int m0, int n0,
                                             float *out_f32, uint16_t *out_bf16, int ldo, const QfQkv &qkv, float *part /* [NP][16][64] */, int tid) {
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, kh = lane >> 5;
    const int kw = K / NP;
    const float bv = bias[n0 + (lane & 31)];          // (requested with the operands, as in k_gemm_skinny)
    for (int p = wave; p < NP; p += 4) {
        const uint16_t *xa = X + (int64_t)(m0 + r) * K + p * kw + kh * 8;
        const uint16_t *wb = W + (int64_t)(n0 + r) * K + p * kw + kh * 8;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
        int k = 0;
        for (; k + 96 <= kw; k += 96) {             // 6 K-steps of 16: 12 loads in flight, then 6 MFMAs
            uint4 xa6[6], wb6[6];
#pragma unroll
            for (int j = 0; j < 6; j++) { xa6[j] = LdL2::u4(xa + k + j * 16); wb6[j] = *(const uint4 *)(wb + k + j * 16); }
#pragma unroll
            for (int j = 0; j < 6; j++) acc = mfma_bf16(xa6[j], wb6[j], acc);
        }
        for (; k < kw; k += 16) acc = mfma_bf16(LdL2::u4(xa + k), *(const uint4 *)(wb + k), acc);
#pragma unroll
        for (int i = 0; i < 16; i++) part[(p * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    // accumulator element i of lane l is token row (i&3) + 8*(i>>2) + 4*(l>>5), output column l&31
    for (int idx = tid; idx < 1024; idx += QF_THREADS) {
        const int i = idx >> 6, l = idx & 63;
        float v = 0.f;
        for (int w = 0; w < NP; w++) v += part[(w * 16 + i) * 64 + l];
        const int n = n0 + (l & 31);
        const int64_t m = m0 + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
        v += bv;
        if constexpr (EPI == 1) {
            v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
            out_bf16[m * ldo + n] = (uint16_t)pack_bf16x2(v, 0.f);
        } else if constexpr (EPI == 2) {
            const int H = qkv.H;
            if (n < H) qkv.q[m * H + n] = (uint16_t)pack_bf16x2(v * qkv.qscale, 0.f);
            else if (n < 2 * H) qkv.k[m * H + (n - H)] = (uint16_t)pack_bf16x2(v, 0.f);
            else if (m < qkv.T) {
                const int64_t b = m / qkv.S, sq = m - b * qkv.S;
                qkv.vt[(b * H + (n - 2 * H)) * qkv.S + vt_pos((int)sq)] = (uint16_t)pack_bf16x2(v, 0.f);
            }
        } else {
            out_f32[m * N + n] = v;
        }
    }
    __syncthreads();                                    // `part` is free for the next tile
}

// k_layernorm (encoder.hip), the skinny path's call: y = LN(x + residual) * g + b, residual from res (fp32) or res16 (bf16 stream)
__device__ __forceinline__ void qf_layernorm(const float *x, const float *res, const uint16_t *res16, const float *g, const float *bta, int T, int H,
                                             float eps, float *y32, uint16_t *y16, int vb, int tid) {
    const int row = vb * 4 + (tid >> 6), lane = tid & 63;
    if (row >= T) return;
    const float *xr = x + (int64_t)row * H;
    float4 v[4], gq[4], bq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane * 4 + j * 256;
        if (i < H) { gq[j] = *(const float4 *)(g + i); bq[j] = *(const float4 *)(bta + i); }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane * 4 + j * 256;
        if (i < H) {
            v[j] = LdL2::f4(xr + i);
            if (res) { const float4 rr = LdL2::f4(res + (int64_t)row * H + i); v[j].x += rr.x; v[j].y += rr.y; v[j].z += rr.z; v[j].w += rr.w; }
            else if (res16) {
                const uint2 h = LdL2::u2(res16 + (int64_t)row * H + i);
                v[j].x += bf16_to_f32((uint16_t)h.x); v[j].y += bf16_to_f32((uint16_t)(h.x >> 16));
                v[j].z += bf16_to_f32((uint16_t)h.y); v[j].w += bf16_to_f32((uint16_t)(h.y >> 16));
            }
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mu = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane * 4 + j * 256;
        if (i < H) {
            const float a0 = v[j].x - mu, b0 = v[j].y - mu, c0 = v[j].z - mu, d0 = v[j].w - mu;
            q += (a0 * a0 + b0 * b0) + (c0 * c0 + d0 * d0);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane * 4 + j * 256;
        if (i < H) {
            const float4 gg = gq[j], bb = bq[j];
            const float4 y = {(v[j].x - mu) * rstd * gg.x + bb.x, (v[j].y - mu) * rstd * gg.y + bb.y,
                              (v[j].z - mu) * rstd * gg.z + bb.z, (v[j].w - mu) * rstd * gg.w + bb.w};
            if (y32) *(float4 *)(y32 + (int64_t)row * H + i) = y;
            const uint2 o = {pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w)};
            *(uint2 *)(y16 + (int64_t)row * H + i) = o;
        }
    }
}

// k_pool<IN16> (encoder.hip): masked mean / CLS, then x / max(||x||, 1e-12); one sequence
template <bool IN16>
__device__ __forceinline__ void qf_pool(const float *x, const uint16_t *x16, const int *mask, int S, int H, int pooling, int normalise,
                                        float *out, int b, int tid, char *smem) {
    float *red = (float *)smem;                 // [256]
    float *wgt = red + 256;                     // [512]
    float (*part)[1024] = (float (*)[1024])(wgt + 512);      // [8][1024]
    float c = 0.f;
    for (int s = tid; s < S; s += 256) { const float w = LdL2::i32(mask + b * S + s) ? 1.f : 0.f; wgt[s] = w; c += w; }
    red[tid] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    float cnt = red[0];
    __syncthreads();
    if (cnt < 1e-9f) cnt = 1e-9f;
    const int64_t base = ((int64_t)b * S) * H;
    const int C = H / 8;
    int G = 256 / C; if (G > 8) G = 8;
    const int ch = tid % C, grp = tid / C;
    auto at8 = [&](int s, float (&v)[8]) {
        if constexpr (IN16) {
            const uint4 h = LdL2::u4(x16 + base + (int64_t)s * H + 8 * ch);
            const uint32_t w[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int i = 0; i < 4; i++) { v[2 * i] = bf16_to_f32((uint16_t)w[i]); v[2 * i + 1] = bf16_to_f32((uint16_t)(w[i] >> 16)); }
        } else {
            const float4 f0 = LdL2::f4(x + base + (int64_t)s * H + 8 * ch), f1 = LdL2::f4(x + base + (int64_t)s * H + 8 * ch + 4);
            v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w; v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
        }
    };
    if (grp < G) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (pooling == AK_POOL_CLS) {
            if (grp == 0) at8(0, acc);
        } else {
            int s = grp;
            for (; s + 3 * G < S; s += 4 * G) {
                float v0[8], v1[8], v2[8], v3[8];
                at8(s, v0); at8(s + G, v1); at8(s + 2 * G, v2); at8(s + 3 * G, v3);
                const float w0 = wgt[s], w1 = wgt[s + G], w2 = wgt[s + 2 * G], w3 = wgt[s + 3 * G];
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = fmaf(w3, v3[i], fmaf(w2, v2[i], fmaf(w1, v1[i], fmaf(w0, v0[i], acc[i]))));
            }
            for (; s < S; s += G) {
                float v0[8];
                at8(s, v0);
                const float w0 = wgt[s];
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = fmaf(w0, v0[i], acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) part[grp][8 * ch + i] = acc[i];
    }
    __syncthreads();
    float ss = 0.f;
    float keep[4];
    int nkeep = 0;
    for (int d = tid; d < H; d += 256) {
        float v = part[0][d];
        if (pooling != AK_POOL_CLS) {
            for (int gq = 1; gq < G; gq++) v += part[gq][d];
            v /= cnt;
        }
        keep[nkeep++] = v;
        ss += v * v;
    }
    float nrm = 1.0f;
    if (normalise) {
        red[tid] = ss;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        nrm = sqrtf(red[0]);
        if (nrm < 1e-12f) nrm = 1e-12f;
    }
    nkeep = 0;
    for (int d = tid; d < H; d += 256) out[(int64_t)b * H + d] = keep[nkeep++] / nrm;
    __syncthreads();
}

}  // namespace

// NP = hidden / 128 (k_embed's template parameter), HD = head size
template <int NP, int HD>
__global__ __launch_bounds__(QF_THREADS) void k_query_forward(QfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *s_flag = (unsigned *)(smem + QF_LDS - 64);            // [0] barrier verdict, [1] ticket, [2] P, [3] prepare's bitmap
    const int tid = threadIdx.x;
    QfCtlSlot *c = &a.ctl->slot[a.epoch & 63];
    unsigned *fail = a.fail;                                        // pinned host memory: the host reads it after the launch without a copy
    // ---- who takes part: the workgroups of ONE XCD (the first arriver's) -------------------------------------------------------
    if (tid == 0) {
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;       // hwreg(HW_REG_XCC_ID, 0, 4)
        unsigned expect = 0;
        __hip_atomic_compare_exchange_strong(&c->target, &expect, xcc + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned home = expect == 0 ? xcc + 1 : expect;
        const bool part = home == xcc + 1;
        // ONE word counts arrivals (high half) and hands out tickets (low half): a participant's ticket is the number of participants
        // that arrived before it, and once every workgroup of the grid has arrived the low half is P
        const unsigned long long old = __hip_atomic_fetch_add(&c->arrived_tickets, (1ull << 32) | (part ? 1ull : 0ull), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT);
        int ticket = part ? (int)(old & 0xffffffffull) : -1, P = 0;
        if (part) {
            unsigned spins = 0;
            for (;;) {
                const unsigned long long v = __hip_atomic_load(&c->arrived_tickets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) >= gridDim.x) { P = (int)(v & 0xffffffffull); break; }
                __builtin_amdgcn_s_sleep(1);
                if (++spins > QF_SPIN_MAX) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ticket = -1; break; }
            }
        }
        s_flag[1] = (unsigned)ticket; s_flag[2] = (unsigned)P;
    }
    __syncthreads();
    const int my = (int)s_flag[1], P = (int)s_flag[2];
    if (my < 0 || P <= 0) return;
    if (my == 0 && tid < 8) ((unsigned *)&a.ctl->slot[(a.epoch + 32) & 63])[tid] = 0;        // the slot of launch epoch + 32, long before its turn
    unsigned gen = 0;
#if AK_DBG_KERNELS
    const int skip = a.dbg_skip;        // measurement only (WRONG RESULTS): 1 = no phase bodies (barriers alone), 2 = no barriers
#else
    constexpr int skip = 0;
#endif
#define QF_BARRIER() do { if (!(skip & 2) && !qf_barrier(c, fail, P, gen, tid, s_flag)) return; } while (0)
    constexpr int H = NP * 128;
    const int B = a.B, S = a.S, T = a.T, I = a.I, heads = a.heads;
    const int MB = a.t32 / 32;
    float *part = (float *)smem;
    const int *ids = a.ids, *mask = a.mask;
    // ---- right-padded rows -> ids + mask ------------------------------------------------------------------------------------------
    if (a.lens) {
        for (int vb = (skip & 1) ? 1 << 30 : my; vb < (B * S + 255) / 256; vb += P) qf_mask_from_lens(a, vb, tid);
        ids = a.oids; mask = a.omask;
        QF_BARRIER();
    }
    // ---- embeddings + key mask ----------------------------------------------------------------------------------------------------
    {
        const int ne = (T + 7) / 8;
        for (int vb = (skip & 1) ? 1 << 30 : my; vb < ne + B; vb += P) {
            if (vb < ne) qf_embed<NP>(a, ids, vb, tid);
            else qf_prepare(a, mask, vb - ne, tid, s_flag + 3);
        }
        QF_BARRIER();
    }
    const AttnArgs at{a.q, a.k, a.vt, mask, a.ctx, B, S, H, heads, a.maskf, a.blkmask, H, HD};      // token-major q / k
    const int np_h = H >= 1024 ? 8 : 4, np_i = I >= 1024 ? 8 : 4;
    for (int l = 0; l < a.L; l++) {
        const QfLayer ly = a.layers[l];
        // QKV
        {
            const QfQkv qk{a.q, a.k, a.vt, H, S, T, a.qscale};
            const int nt = 3 * H / 32;
            for (int t = (skip & 1) ? 1 << 30 : my; t < MB * nt; t += P)
                qf_gemm_tile<2>(a.x16, ly.wqkv, ly.bqkv, 3 * H, H, np_h, (t / nt) * 32, (t % nt) * 32, nullptr, nullptr, 0, qk, part, tid);
            QF_BARRIER();
        }
        // attention: one (sequence, head) item = k_attn_d<HD, 4>'s block
        for (int it = (skip & 1) ? 1 << 30 : my; it < B * heads; it += P) {
            attn_d_body<HD, 4, LdL2>(at, it, smem);
            __syncthreads();
        }
        QF_BARRIER();
        // out-projection -> y32, then LayerNorm-1 (+ residual)
        {
            const QfQkv none{};
            const int nt = H / 32;
            for (int t = (skip & 1) ? 1 << 30 : my; t < MB * nt; t += P)
                qf_gemm_tile<0>(a.ctx, ly.wo, ly.bo, H, H, np_h, (t / nt) * 32, (t % nt) * 32, a.y32, nullptr, 0, none, part, tid);
            QF_BARRIER();
            for (int vb = (skip & 1) ? 1 << 30 : my; vb < (T + 3) / 4; vb += P)
                qf_layernorm(a.y32, a.x32, a.x32 ? nullptr : a.x16, ly.ln1g, ly.ln1b, T, H, a.eps, a.x32, a.x16, vb, tid);
            QF_BARRIER();
        }
        // feed-forward: up + GELU -> f (bf16), down -> y32, LayerNorm-2 (+ residual)
        {
            const QfQkv none{};
            int nt = I / 32;
            for (int t = (skip & 1) ? 1 << 30 : my; t < MB * nt; t += P)
                qf_gemm_tile<1>(a.x16, ly.w1, ly.b1, I, H, np_h, (t / nt) * 32, (t % nt) * 32, nullptr, a.f, I, none, part, tid);
            QF_BARRIER();
            nt = H / 32;
            for (int t = (skip & 1) ? 1 << 30 : my; t < MB * nt; t += P)
                qf_gemm_tile<0>(a.f, ly.w2, ly.b2, H, I, np_i, (t / nt) * 32, (t % nt) * 32, a.y32, nullptr, 0, none, part, tid);
            QF_BARRIER();
            for (int vb = (skip & 1) ? 1 << 30 : my; vb < (T + 3) / 4; vb += P)
                qf_layernorm(a.y32, a.x32, a.x32 ? nullptr : a.x16, ly.ln2g, ly.ln2b, T, H, a.eps, a.x32, a.x16, vb, tid);
            QF_BARRIER();
        }
    }
    for (int b = (skip & 1) ? 1 << 30 : my; b < B; b += P) {
        if (a.x32) qf_pool<false>(a.x32, a.x16, mask, S, H, a.pooling, a.normalise, a.out, b, tid, smem);
        else qf_pool<true>(nullptr, a.x16, mask, S, H, a.pooling, a.normalise, a.out, b, tid, smem);
    }
#undef QF_BARRIER
}

bool query_forward_supported(int H, int I, int heads, int64_t T, int S) {
    const int hd = heads > 0 ? H / heads : 0;
    if (switches().query_fused.load(std::memory_order_relaxed) == 0) return false;
    // hidden 384 / head size 32 only: at head size 64 the 47-launch path runs k_attn_s, not k_attn_d -- another kernel, other bits
    return T >= 1 && T <= 64 && S % 32 == 0 && S <= 64 && H == 384 && hd == 32 && I % 32 == 0 &&
           gemm_skinny_supported(H, H) && gemm_skinny_supported(H, I) && gemm_skinny_supported(I, H) && gemm_skinny_supported(3 * H, H);
}

int launch_query_forward(const QfArgs &a, hipStream_t st) {
    static std::atomic<bool> attr{false};
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_query_forward<3, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, QF_LDS));
        attr = true;
    }
    if (a.H != 384) AK_FAIL(-1, "launch_query_forward: hidden size");
    k_query_forward<3, 32><<<QF_GRID, QF_THREADS, QF_LDS, st>>>(a);
    AK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ak
