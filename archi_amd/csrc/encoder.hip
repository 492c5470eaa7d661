// encoder.hip -- chunk-embedding forward pass on the GPU (a1/a2/a3): the C-ABI behind
// Embeddings.embed_documents / embed_query
// (/root/reference/src/data_manager/vectorstore/manager.py:373,
//  src/data_manager/vectorstore/postgres_vectorstore.py:143,245,390).
//   embeddings(LN) -> L x [ QKV GEMM -> attention -> out-proj GEMM + residual + LN ->
//                           FFN-up GEMM(+GELU) -> FFN-down GEMM + residual + LN ] -> pool -> L2 normalise
// (residual + LayerNorm run inside the GEMM for hidden size 384, gemm_ln.hip; as a separate kernel otherwise)
// bf16 MFMA GEMMs with fp32 accumulate; fp32 residual stream, LayerNorm, softmax, pooling.
#include "mfma_tile.h"
#include "encoder_kernels.h"
#include "switches.h"

#include <mutex>
#include <vector>

namespace ak {


// One wave per row: y = LayerNorm(x + res) * g + b. The residual comes from res (fp32; it may alias y32: each lane
// rewrites only what it read) or, when res is NULL, from res16 (the bf16-only residual stream; it may alias y16 the same
// way). Writes bf16 (next GEMM input) and, unless y32 is NULL, fp32. 16 B per lane per access (H % 4 == 0, H <= 1024).
// x16in != NULL: the GEMM output arrives as bf16 rows (bf16-residual mode on the unfused path: halves what the GEMM writes
// and this kernel reads; the sum and the statistics are still fp32).
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ x, const float *res, const uint16_t *res16,
                                                   const float *__restrict__ g, const float *__restrict__ bta, int T, int H,
                                                   float eps, float *y32, uint16_t *y16, const uint16_t *__restrict__ x16in = nullptr) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= T) return;
    const float *xr = x + (int64_t)row * H;
    float4 v[4];
    // gamma / beta of this lane's columns, requested with the row itself: behind the two reductions they were a third dependent
    // round trip of a kernel that is three of them (round 6; same values, same arithmetic)
    float4 gq[4], bq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane * 4 + j * 256;
        if (i < H) { gq[j] = *(const float4 *)(g + i); bq[j] = *(const float4 *)(bta + i); }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int i = lane * 4 + j * 256;
        if (i < H) {
            if (x16in) {
                const uint2 h = *(const uint2 *)(x16in + (int64_t)row * H + i);
                v[j] = {bf16_to_f32((uint16_t)h.x), bf16_to_f32((uint16_t)(h.x >> 16)), bf16_to_f32((uint16_t)h.y),
                        bf16_to_f32((uint16_t)(h.y >> 16))};
            } else v[j] = *(const float4 *)(xr + i);
            if (res) { const float4 rr = *(const float4 *)(res + (int64_t)row * H + i); v[j].x += rr.x; v[j].y += rr.y; v[j].z += rr.z; v[j].w += rr.w; }
            else if (res16) {
                const uint2 h = *(const uint2 *)(res16 + (int64_t)row * H + i);
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
        int i = lane * 4 + j * 256;
        if (i < H) {
            float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, d = v[j].w - mu;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int i = lane * 4 + j * 256;
        if (i < H) {
            const float4 gg = gq[j], bb = bq[j];
            float4 y = {(v[j].x - mu) * rstd * gg.x + bb.x, (v[j].y - mu) * rstd * gg.y + bb.y,
                        (v[j].z - mu) * rstd * gg.z + bb.z, (v[j].w - mu) * rstd * gg.w + bb.w};
            if (y32) *(float4 *)(y32 + (int64_t)row * H + i) = y;
            uint2 o = {mt::pack_bf16x2(y.x, y.y), mt::pack_bf16x2(y.z, y.w)};
            *(uint2 *)(y16 + (int64_t)row * H + i) = o;
        }
    }
}

// bf16 rows in, bf16 rows out, no residual (what follows a MODE 4 GEMM on the hidden != 384 path): a lane owns one contiguous
// run of H / 64 features of TWO rows (two independent load -> reduce -> reduce -> store chains per wave), 4 * NP bytes per
// lane and row and access. NP = H / 128 pairs per lane.
template <int NP>
__global__ __launch_bounds__(256) void k_layernorm16(const uint16_t *__restrict__ x16in, const float *__restrict__ g,
                                                     const float *__restrict__ bta, int T, float eps, uint16_t *__restrict__ y16) {
    constexpr int H = NP * 128, RW = 2;
    struct __attribute__((packed, aligned(4))) Run { uint32_t w[NP]; };
    struct __attribute__((packed, aligned(8))) RunF { float2 w[NP]; };
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (row0 >= T) return;
    Run in[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        const int row = row0 + q < T ? row0 + q : T - 1;
        in[q] = *(const Run *)(x16in + (int64_t)row * H + lane * 2 * NP);
    }
    const RunF gg = *(const RunF *)(g + lane * 2 * NP), bb = *(const RunF *)(bta + lane * 2 * NP);
    float2 v[RW][NP];
    float s[RW], sq[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        s[q] = 0.f;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            v[q][j] = {bf16_to_f32((uint16_t)in[q].w[j]), bf16_to_f32((uint16_t)(in[q].w[j] >> 16))};
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
        const float rstd = 1.0f / sqrtf(sq[q] / (float)H + eps);
        Run o;
#pragma unroll
        for (int j = 0; j < NP; j++)
            o.w[j] = mt::pack_bf16x2((v[q][j].x - s[q]) * rstd * gg.w[j].x + bb.w[j].x, (v[q][j].y - s[q]) * rstd * gg.w[j].y + bb.w[j].y);
        *(Run *)(y16 + (int64_t)(row0 + q) * H + lane * 2 * NP) = o;
    }
}
// lazy LayerNorm, last layer: rows r~ = gamma (.) r with r's (mean, 1 / std) per token -> LN(r) = rstd (r~ - mu gamma) + beta, bf16
__global__ __launch_bounds__(256) void k_ln_apply16(const uint16_t *__restrict__ rt, const float *__restrict__ stats, const float *__restrict__ g,
                                                    const float *__restrict__ bta, int64_t T, int H, uint16_t *__restrict__ y16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 8-feature run each
    const int per = H / 8;
    if (i >= T * per) return;
    const int64_t t = i / per;
    const int f = (int)(i - t * per) * 8;
    const float2 ms = *(const float2 *)(stats + t * 2);
    const uint4 v = *(const uint4 *)(rt + t * H + f);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float a0 = bf16_to_f32((uint16_t)w[q]), a1 = bf16_to_f32((uint16_t)(w[q] >> 16));
        o[q] = mt::pack_bf16x2(fmaf(ms.y, fmaf(-ms.x, g[f + 2 * q], a0), bta[f + 2 * q]), fmaf(ms.y, fmaf(-ms.x, g[f + 2 * q + 1], a1), bta[f + 2 * q + 1]));
    }
    *(uint4 *)(y16 + t * H + f) = uint4{o[0], o[1], o[2], o[3]};
}

template <int NP = 1>
static bool launch_layernorm16(int np, const uint16_t *x16in, const float *g, const float *bta, int T, float eps, uint16_t *y16, hipStream_t st) {
    if constexpr (NP <= 8) {
        if (np == NP) {
            k_layernorm16<NP><<<(unsigned)((T + 7) / 8), 256, 0, st>>>(x16in, g, bta, T, eps, y16);
            return true;
        }
        return launch_layernorm16<NP + 1>(np, x16in, g, bta, T, eps, y16, st);
    } else return false;
}

// One wave per TWO tokens: LN(word[id] + pos[s] + type[0]). A lane owns NP = H / 128 adjacent feature PAIRS (H / 64 features:
// one 4 * NP-byte run of each table row, so a wave instruction covers whole rows -- 12 bytes per lane at H = 384 instead of
// three passes of 4; 44 -> ~20 us per 65 536 tokens); H % 128 == 0, H <= 1024. The kernel is a chain of dependent round trips
// (id -> table row -> two reductions -> store) at whatever the occupancy keeps in flight: two independent tokens per wave
// double that (29 -> ~20 us per 65 536 tokens at hidden 384).
template <int NP>
__global__ __launch_bounds__(256) void k_embed(const int *__restrict__ ids, int T, int S, int vocab,
                                               const uint16_t *__restrict__ word, const uint16_t *__restrict__ pos,
                                               const uint16_t *__restrict__ type, const float *__restrict__ g,
                                               const float *__restrict__ bta, float eps, float *__restrict__ y32 /* nullable */,
                                               uint16_t *__restrict__ y16) {
    constexpr int H = NP * 128, RW = 2;
    struct __attribute__((packed, aligned(4))) Run { uint32_t w[NP]; };
    struct __attribute__((packed, aligned(8))) RunF { float2 w[NP]; };
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW, lane = threadIdx.x & 63;
    if (row0 >= T) return;
    int id[RW], rowq[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        rowq[q] = row0 + q < T ? row0 + q : T - 1;
        id[q] = ids[rowq[q]];
        if (id[q] < 0 || id[q] >= vocab) id[q] = 0;
    }
    Run a[RW], b[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        a[q] = *(const Run *)(word + (int64_t)id[q] * H + lane * 2 * NP);
        b[q] = *(const Run *)(pos + (int64_t)(rowq[q] % S) * H + lane * 2 * NP);
    }
    const Run c = *(const Run *)(type + lane * 2 * NP);
    const RunF gg = *(const RunF *)(g + lane * 2 * NP), bb = *(const RunF *)(bta + lane * 2 * NP);
    float2 v[RW][NP];
    float s[RW], sq[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        s[q] = 0.f;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            v[q][j].x = bf16_to_f32((uint16_t)a[q].w[j]) + bf16_to_f32((uint16_t)b[q].w[j]) + bf16_to_f32((uint16_t)c.w[j]);
            v[q][j].y = bf16_to_f32((uint16_t)(a[q].w[j] >> 16)) + bf16_to_f32((uint16_t)(b[q].w[j] >> 16)) + bf16_to_f32((uint16_t)(c.w[j] >> 16));
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
        const float mu = s[q], rstd = 1.0f / sqrtf(sq[q] / (float)H + eps);
        Run o16; RunF o32;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const float2 y = {(v[q][j].x - mu) * rstd * gg.w[j].x + bb.w[j].x, (v[q][j].y - mu) * rstd * gg.w[j].y + bb.w[j].y};
            o32.w[j] = y;
            o16.w[j] = mt::pack_bf16x2(y.x, y.y);
        }
        if (y32) *(RunF *)(y32 + (int64_t)(row0 + q) * H + lane * 2 * NP) = o32;
        *(Run *)(y16 + (int64_t)(row0 + q) * H + lane * 2 * NP) = o16;
    }
}
template <int NP = 1>
static int launch_embed(int np, const int *ids, int T, int S, int vocab, const uint16_t *word, const uint16_t *pos, const uint16_t *type,
                        const float *g, const float *bta, float eps, float *y32, uint16_t *y16, hipStream_t st) {
    if constexpr (NP <= 8) {
        if (np == NP) {
            k_embed<NP><<<(unsigned)((T + 7) / 8), 256, 0, st>>>(ids, T, S, vocab, word, pos, type, g, bta, eps, y32, y16);
            return 0;
        }
        return launch_embed<NP + 1>(np, ids, T, S, vocab, word, pos, type, g, bta, eps, y32, y16, st);
    } else return -1;
}

// One block per sequence: masked mean (sentence-transformers Pooling) or CLS, then x / max(||x||, 1e-12).
// x: fp32 hidden states, or NULL to read the bf16 stream x16 instead (residual_bf16 mode). Thread t owns the 8 features of
// 16-byte chunk t % (H / 8) for the tokens s = t / (H / 8), + G, + 2G, ... (G = 256 / (H / 8) token groups; four independent
// loads in flight), the groups are then added through LDS in a fixed order (one thread per feature pair looped over all
// S tokens with 4-byte loads before: 47 us per 256 x 256 tokens, 1 TB/s). H % 8 == 0, H / 8 <= 256, H <= 1024.
template <bool IN16>
__global__ __launch_bounds__(256) void k_pool(const float *__restrict__ x, const uint16_t *__restrict__ x16,
                                              const int *__restrict__ mask, int S, int H,
                                              int pooling, int normalise, float *__restrict__ out) {
    __shared__ float red[256];
    __shared__ float wgt[512];
    __shared__ float part[8][1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    float c = 0.f;
    for (int s = tid; s < S; s += 256) { float w = mask[b * S + s] ? 1.f : 0.f; wgt[s] = w; c += w; }
    red[tid] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    float cnt = red[0];
    __syncthreads();
    if (cnt < 1e-9f) cnt = 1e-9f;
    const int64_t base = ((int64_t)b * S) * H;
    const int C = H / 8;                               // 16-byte (bf16) / 32-byte (fp32) chunks per token row
    int G = 256 / C; if (G > 8) G = 8;                 // token groups
    const int ch = tid % C, grp = tid / C;
    auto at8 = [&](int s, float (&v)[8]) {             // features 8 ch .. 8 ch + 7 of token s
        if constexpr (IN16) {
            const uint4 h = *(const uint4 *)(x16 + base + (int64_t)s * H + 8 * ch);
            const uint32_t w[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int i = 0; i < 4; i++) { v[2 * i] = bf16_to_f32((uint16_t)w[i]); v[2 * i + 1] = bf16_to_f32((uint16_t)(w[i] >> 16)); }
        } else {
            const float4 f0 = *(const float4 *)(x + base + (int64_t)s * H + 8 * ch), f1 = *(const float4 *)(x + base + (int64_t)s * H + 8 * ch + 4);
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
    float keep[4];                                     // this thread's (<= 4) features, for the normalisation below
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
}


// =====================================================================================================================
// fp32 PARITY MODE (AkBertConfig.precision == 1): the same forward pass in float32 throughout -- float32 weights, plain
// FMA GEMMs (k ascending), erff GELU, fp32 softmax -- so that the embeddings can be set against the reference's CPU
// embedder (sentence-transformers on torch fp32, manager.py:373) at ~1e-6 instead of the bf16 path's ~1e-3. No MFMA, no
// tuning: a checking tool (the bf16-input MFMA cannot give this, and v_mfma_f32_32x32x2_f32 runs at the f32 vector rate
// anyway). tests/test_encoder_gpu.py holds it to max|diff| <= 1e-5 on unit vectors against transformers.BertModel.
// =====================================================================================================================
__global__ __launch_bounds__(256) void k32_embed(const int *__restrict__ ids, int T, int S, int H, int vocab,
                                                 const float *__restrict__ word, const float *__restrict__ pos,
                                                 const float *__restrict__ type, const float *__restrict__ g,
                                                 const float *__restrict__ bta, float eps, float *__restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= T) return;
    int id = ids[row];
    if (id < 0 || id >= vocab) id = 0;
    const float *w = word + (int64_t)id * H, *p = pos + (int64_t)(row % S) * H;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; v[j] = i < H ? (w[i] + p[i]) + type[i] : 0.f; s += v[j]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mu = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; if (i < H) { const float d = v[j] - mu; q += d * d; } }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; if (i < H) y[(int64_t)row * H + i] = (v[j] - mu) * rstd * g[i] + bta[i]; }
}

// y = LayerNorm(x + r) * g + b (r == NULL: LayerNorm(x)), one wave per row (H <= 1024), in place over x allowed
__global__ __launch_bounds__(256) void k32_add_ln(const float *x, const float *__restrict__ r, int T, int H,
                                                  const float *__restrict__ g, const float *__restrict__ bta, float eps, float *y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= T) return;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; v[j] = i < H ? (r ? x[(int64_t)row * H + i] + r[(int64_t)row * H + i] : x[(int64_t)row * H + i]) : 0.f; s += v[j]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mu = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; if (i < H) { const float d = v[j] - mu; q += d * d; } }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; if (i < H) y[(int64_t)row * H + i] = (v[j] - mu) * rstd * g[i] + bta[i]; }
}

// Y[T][N] = X[T][K] . W[N][K]^T + bias (+ exact GELU): 64 x 64 tile, 16 x 16 threads, 4 x 4 outputs per thread, K in steps of
// 16 through LDS; every output is one fmaf chain over k ascending. N % 64 == 0, K % 16 == 0, T arbitrary.
template <bool GELU>
__global__ __launch_bounds__(256) void k32_gemm(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                                                int T, int N, int K, float *__restrict__ Y) {
    __shared__ float sx[16][65], sw[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int t0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int rr = i >> 4, kk = i & 15;
            sx[kk][rr] = (t0 + rr) < T ? X[(int64_t)(t0 + rr) * K + k0 + kk] : 0.f;
            sw[kk][rr] = W[(int64_t)(n0 + rr) * K + k0 + kk];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { a[i] = sx[kk][ty * 4 + i]; b[i] = sw[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int t = t0 + ty * 4 + i;
        if (t >= T) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + tx * 4 + j;
            float v = acc[i][j] + bias[n];
            if (GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
            Y[(int64_t)t * N + n] = v;
        }
    }
}

// one wave per (sequence, head, query row): scores over the S <= 512 keys (lanes over keys), fp32 softmax, P.V (lanes over
// the head dimension). qkv: [T][3H] (q | k | v); masked keys get -inf like the additive mask of the reference model.
__global__ __launch_bounds__(256) void k32_attn(const float *__restrict__ qkv, const int *__restrict__ mask, int B, int S, int H,
                                                int heads, float *__restrict__ ctx) {
    __shared__ float s_p[4][512];
    __shared__ float s_q[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t gid = (int64_t)blockIdx.x * 4 + wv;          // (b, head, query)
    const int hd = H / heads;
    if (gid >= (int64_t)B * heads * S) return;
    const int qi = (int)(gid % S), hh = (int)((gid / S) % heads), b = (int)(gid / ((int64_t)S * heads));
    const int64_t row = (int64_t)b * S + qi;
    if (lane < hd) s_q[wv][lane] = qkv[row * 3 * H + hh * hd + lane];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const float scale = 1.0f / sqrtf((float)hd);
    float sc[8];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int j = lane + 64 * t;
        sc[t] = -INFINITY;
        if (j < S && mask[b * S + j]) {
            const float *kr = qkv + ((int64_t)b * S + j) * 3 * H + H + hh * hd;
            float d = 0.f;
            for (int e = 0; e < hd; e++) d = fmaf(s_q[wv][e], kr[e], d);
            sc[t] = d * scale;
        }
        mx = fmaxf(mx, sc[t]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int j = lane + 64 * t;
        const float p = (sc[t] == -INFINITY || mx == -INFINITY) ? 0.f : expf(sc[t] - mx);
        if (j < S) s_p[wv][j] = p;
        sum += p;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane < hd) {
        float o = 0.f;
        for (int j = 0; j < S; j++) o = fmaf(s_p[wv][j], qkv[((int64_t)b * S + j) * 3 * H + 2 * H + hh * hd + lane], o);
        ctx[row * H + hh * hd + lane] = sum > 0.f ? o / sum : 0.f;
    }
}

// the whole forward pass in float32; ws (grown here) holds x | y | qkv | ctx | f
static int forward_f32(const AkBertConfig &c, const void *const *w, const uint16_t *const *x3, const int *ids, const int *mask, int B, int S,
                       int pooling, int normalise, float *out, float **ws, size_t *ws_bytes, hipStream_t st);

constexpr int X3_SLOTS = 20;
struct Layer {
    const uint16_t *wqkv; const float *bqkv;
    const uint16_t *wo; const float *bo; const float *ln1g, *ln1b;
    const uint16_t *w1; const float *b1; const uint16_t *w2; const float *b2; const float *ln2g, *ln2b;
    const uint16_t *wf = nullptr;   // hidden 384: W1 and W2 in the fragment order of the fused feed-forward kernel (ffn.hip)
    const uint16_t *wof = nullptr;  // ... and Wo, directly in front of them
    const uint16_t *wq16 = nullptr; // ... and the QKV matrix + permuted bias for k_qkv384
    // lazy LayerNorm (hidden % 256 == 0, not the fused hidden-384 path; gemm.hip): for the matrices whose A operand is a
    // LayerNorm's output, c = W gamma and b' = b + W beta. cqkv uses LayerNorm-2 of the PREVIOUS layer (NULL in layer 0: the
    // embedding LayerNorm is applied by k_embed), c1 this layer's LayerNorm-1.
    const float *cqkv = nullptr, *bqkv_f = nullptr;
    const float *c1 = nullptr, *b1_f = nullptr;
};
struct Encoder {
    AkBertConfig cfg;
    const uint16_t *word, *pos, *type; const float *eg, *eb;
    std::vector<Layer> layers;
    std::vector<void *> owned;      // fused QKV weights/biases
    // activation workspace (grown on demand)
    int64_t cap_tokens = 0; int cap_B = 0;
    std::vector<const void *> raw;  // the caller's weight pointers, in header order (fp32 parity mode reads them directly)
    float *ws32 = nullptr; size_t ws32_bytes = 0;
    std::vector<const uint16_t *> x3;   // precision 2 (split bf16): per layer X3_SLOTS slots -- hi, lo of [wq | wk | wv], their bias, 3 unused, hi, lo of wo, w1, w2,
                                        // then [12]-[15] the same four matrices as [hi | lo] ROWS (gemm.hip MODE 5 / 6), row counts padded with zero
                                        // rows to multiples of 256, and [16]-[18] the biases of q | k | v, wo, w2 padded likewise (float32) (owned)
    // single-launch query forward (query_forward.hip): barrier slots, failure word (pinned host memory), layer table, launch number
    QfCtl *qf_ctl = nullptr; unsigned *qf_fail = nullptr; QfLayer *qf_layers = nullptr; unsigned qf_epoch = 0; bool qf_off = false;
    float *x32 = nullptr, *y32 = nullptr;
    uint16_t *x16 = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *ctx = nullptr, *f = nullptr;
    int *lens_ids = nullptr, *lens_mask = nullptr; int64_t lens_cap = 0;    // ak_encoder_forward_lens: contiguous ids / 0-1 mask of the tile
    float *maskf = nullptr;       // additive key mask [tokens] + per-sequence block bitmap behind it (attention.hip)
    float *st1 = nullptr, *st2 = nullptr;   // lazy LayerNorm: per-token (mean, 1 / std) [tokens][2] behind the two sub-layers
    float *stp = nullptr;                   // ... and the partial sums [H / 128][tokens][2] they are made of
    std::mutex mu;
};

static void free_ws(Encoder &e) {
    void *p[] = {e.x32, e.y32, e.x16, e.q, e.k, e.vt, e.ctx, e.f, e.maskf, e.st1, e.st2, e.stp};
    for (void *x : p) if (x) hipFree(x);
    e.x32 = e.y32 = nullptr; e.x16 = e.q = e.k = e.vt = e.ctx = e.f = nullptr; e.maskf = nullptr; e.st1 = e.st2 = e.stp = nullptr;
    e.cap_tokens = 0;
}
// the V^T buffer [b][H][S] is written for every padded token too (unconditional stores in gemm.hip): a partial
// batch row past the last sequence needs up to S <= 512 more token slots
constexpr int64_t VT_PAD = 512;
static int reserve_ws(Encoder &e, int64_t tpad) {
    if (tpad <= e.cap_tokens) return 0;
    free_ws(e);
    const int H = e.cfg.hidden, I = e.cfg.intermediate;
    AK_HIP(hipMalloc((void **)&e.x32, tpad * H * 4)); AK_HIP(hipMalloc((void **)&e.y32, tpad * H * 4));
    AK_HIP(hipMalloc((void **)&e.x16, tpad * H * 2)); AK_HIP(hipMalloc((void **)&e.q, tpad * H * 2));
    AK_HIP(hipMalloc((void **)&e.k, tpad * H * 2)); AK_HIP(hipMalloc((void **)&e.vt, (tpad + VT_PAD) * H * 2));
    AK_HIP(hipMalloc((void **)&e.ctx, tpad * H * 2)); AK_HIP(hipMalloc((void **)&e.f, tpad * (int64_t)I * 2));
    AK_HIP(hipMalloc((void **)&e.maskf, tpad * 4 + (tpad / 32 + 1) * 4));
    if (!e.layers.empty() && e.layers[0].c1) {
        AK_HIP(hipMalloc((void **)&e.st1, (size_t)tpad * 2 * 4)); AK_HIP(hipMalloc((void **)&e.st2, (size_t)tpad * 2 * 4));
        AK_HIP(hipMalloc((void **)&e.stp, (size_t)(H / 128) * tpad * 2 * 4));
        AK_HIP(hipMemset(e.st1, 0, (size_t)tpad * 2 * 4)); AK_HIP(hipMemset(e.st2, 0, (size_t)tpad * 2 * 4));
        AK_HIP(hipMemset(e.stp, 0, (size_t)(H / 128) * tpad * 2 * 4));
    }
    AK_HIP(hipMemset(e.x32, 0, tpad * H * 4)); AK_HIP(hipMemset(e.y32, 0, tpad * H * 4));
    AK_HIP(hipMemset(e.x16, 0, tpad * H * 2)); AK_HIP(hipMemset(e.q, 0, tpad * H * 2));
    AK_HIP(hipMemset(e.k, 0, tpad * H * 2)); AK_HIP(hipMemset(e.vt, 0, (tpad + VT_PAD) * H * 2));
    AK_HIP(hipMemset(e.ctx, 0, tpad * H * 2)); AK_HIP(hipMemset(e.f, 0, tpad * (int64_t)I * 2));
    e.cap_tokens = tpad;
    return 0;
}


// split mode: from how many tokens on the batch runs on gemm.hip's tiles (below: k3_gemm's 128 x 128 tiles fill the chip better).
// Same box, ms per forward, k3_gemm / tiles (gpurun_out/r6x3c): hidden 384 at 8 192 tokens 1.29 / 1.54, 16 384 2.05 / 2.19,
// 24 576 2.97 / 2.77, 32 768 3.74 / 3.27; hidden 768 at 8 192 6.37 / 6.48, 12 288 9.80 / 10.03, 16 384 12.28 / 11.57, 32 768 24.1 / 20.1
static int64_t x3w_min_tokens(int H) { return H >= 512 ? 16384 : 20480; }
constexpr int X3_PADN_DEFAULT = 1;      // MiniLM 256 x 256, ms per forward: none 6.79-6.81, QKV 6.70, QKV + FFN-down 6.70, all three 6.72 (gpurun_out/r6z2)
// x3: nullptr = exact float32 GEMMs (precision 1); else the layer matrices split into bf16 hi / lo (precision 2: every GEMM as
// hi.hi + lo.hi + hi.lo on the bf16 matrix cores, encoder_f32.hip k3_gemm). Everything else is the same float32 code.
static int forward_f32(const AkBertConfig &c, const void *const *w, const uint16_t *const *x3, const int *ids, const int *mask, int B, int S, int pooling,
                       int normalise, float *out, float **ws, size_t *ws_bytes, hipStream_t st) {
    const int H = c.hidden, I = c.intermediate, L = c.layers;
    const int64_t T = (int64_t)B * S;
    // Split mode on gemm.hip's tiles (x3_tiles below): whole 256-token tiles, so the rows are padded to one (rows past T: zeros in,
    // never read by attention or pooling)
    static const int x3w = env_get("AK_X3_TILES") ? atoi(env_get("AK_X3_TILES")) : 1;      // 0 = off, 2 = at every token count (tests); default: from x3w_min_tokens(H) on
    const int64_t Tp = (T + 255) / 256 * 256;
    const bool x3_tiles = x3 && x3w && (x3w == 2 || T >= x3w_min_tokens(H)) && f32_mfma_supported(H, I, c.heads) && gemm_x3w_supported(Tp, H, H) &&
                          gemm_x3w_supported(Tp, 3 * H, H) && gemm_x3w_supported(Tp, I, H) && gemm_x3w_supported(Tp, H, I);
    const int64_t Ta = x3_tiles ? Tp : T;
    // ... and an output width that is not a multiple of 256 (hidden 384: N = 1152, 384) padded to one where that puts the launch on
    // the wide tile with at least a tile per CU (the matrices carry zero rows for it; the padded columns are written and never read).
    // AK_X3_PADN (A/B): bit 0 QKV, 1 out-projection, 2 FFN-down; bit 3 (tests): at every token count
    static const int padn = env_get("AK_X3_PADN") ? atoi(env_get("AK_X3_PADN")) : X3_PADN_DEFAULT;
    auto padded = [&](int N, int bit) {
        const int Np = (N + 255) / 256 * 256;
        return (N % 256) && (padn >> bit & 1) && ((padn & 8) || (Tp / 256) * (Np / 256) >= 256) ? Np : N;
    };
    const int Qn = x3_tiles ? padded(3 * H, 0) : 3 * H, On = x3_tiles ? padded(H, 1) : H, Dn = x3_tiles ? padded(H, 2) : H;
    const int Yn = On > Dn ? On : Dn;
    const size_t need = (size_t)Ta * (9 * (size_t)H + (size_t)I) * 4;      // x, y, ctx | qkv | q, k, v | f   (tiles: x | y | cs | qkv | xs | fs, Yn + Qn <= 6 H)
    if (need > *ws_bytes) {
        if (*ws) hipFree(*ws);
        *ws = nullptr; *ws_bytes = 0;
        AK_HIP(hipMalloc((void **)ws, need));
        AK_HIP(hipMemsetAsync(*ws, 0, need, st));
        *ws_bytes = need;
    }
    float *x = *ws, *y = x + Ta * H, *ctx = y + Ta * H, *qkv = ctx + Ta * H, *sep = qkv + Ta * 3 * H, *f = sep + Ta * 3 * H;
    const unsigned rows4 = (unsigned)((T + 3) / 4);
    if (!x3_tiles) {
        k32_embed<<<rows4, 256, 0, st>>>(ids, (int)T, S, H, c.vocab_size, (const float *)w[0], (const float *)w[1], (const float *)w[2],
                                         (const float *)w[3], (const float *)w[4], c.ln_eps, x);
        AK_HIP(hipGetLastError());
    }
    // the matrix-core kernels (encoder_f32.hip); the scalar kernels above stay as their cross-check in libarchi_hip_dbg.so
    // (AK_F32_SCALAR=1 there; the GEMM is bit-identical, the attention agrees to float32 rounding)
    static const bool scalar = dbg_env_int("AK_F32_SCALAR", 0) != 0;
    if (x3_tiles) {
        // Activations between the launches as bf16 [hi | lo] rows beside the float32 residual stream: xs (LayerNorm output), cs
        // (attention context), fs (GELU output) -- in the float32 path's ctx / q,k,v / f areas (same bytes per element). Every GEMM
        // is gemm.hip's LDS-DMA tile walking 3 K (MODE 5: float32 out; MODE 6: exact GELU, split out).
        float *y2 = x + Ta * H, *qkv2 = y2 + Ta * Yn;
        uint16_t *cs = (uint16_t *)(qkv2 + Ta * Qn), *xs = cs + Ta * 2 * H, *fs = (uint16_t *)f;
        if ((const char *)(xs + Ta * 2 * H) > (const char *)f) AK_FAIL(-1, "forward (split bf16): workspace layout");
        if (launch_embed_split(ids, T, S, H, c.vocab_size, (const float *)w[0], (const float *)w[1], (const float *)w[2], (const float *)w[3],
                               (const float *)w[4], c.ln_eps, x, xs, st)) return -10;
        for (int l = 0; l < L; l++) {
            const void *const *p = w + 5 + 16 * l;
            const uint16_t *const *s3 = x3 + X3_SLOTS * l;
            auto gemm = [&](int mode, const uint16_t *X, const uint16_t *W2, const float *bias, int N, int K1, float *o32, uint16_t *o16, int nvalid = 0) -> int {
                GemmArgs g{};
                g.X = X; g.W = W2; g.bias = bias; g.T = (int)Tp; g.N = N; g.K = 3 * K1; g.out_f32 = o32; g.out_bf16 = o16; g.ldo = 2 * N; g.nvalid = nvalid;
                return launch_gemm_x3w(mode, g, st);
            };
            if (gemm(5, xs, s3[12], (const float *)s3[16], Qn, H, qkv2, nullptr, 3 * H)) return -10;
            if (launch_attn_x3_split(qkv2, Qn, mask, B, S, H, c.heads, cs, st)) return -10;
            // hidden 384: out-projection / FFN-down with the residual add and the LayerNorm in the epilogue (gemm_ln.hip, X3): the
            // float32 sub-layer output never goes to HBM (AK_X3_GEMMLN=0: MODE 5 + k3_add_ln, as the other widths)
            static const bool x3_gemmln = !(env_get("AK_X3_GEMMLN") && atoi(env_get("AK_X3_GEMMLN")) == 0);
            const bool fuse_ln = x3_gemmln && gemm_ln_supported(H, Tp, H) && gemm_ln_supported(H, Tp, I) && I % 32 == 0;
            auto gemm_ln = [&](const uint16_t *X, const uint16_t *W2, const float *bias, const float *gam, const float *bet, int K1) -> int {
                GemmLnArgs g{};
                g.X = X; g.W = W2; g.bias = bias; g.gamma = gam; g.beta = bet; g.x32 = x; g.x16 = xs; g.T = (int)Tp; g.K = 3 * K1; g.eps = c.ln_eps;
                return launch_gemm_ln_x3(g, st);
            };
            if (fuse_ln) { if (gemm_ln(cs, s3[13], (const float *)p[7], (const float *)p[8], (const float *)p[9], H)) return -10; }
            else {
                if (gemm(5, cs, s3[13], (const float *)s3[17], On, H, y2, nullptr)) return -10;
                if (launch_add_ln_split(y2, On, x, T, H, (const float *)p[8], (const float *)p[9], c.ln_eps, x, xs, st)) return -10;
            }
            // (measured and not kept: the feed-forward pair in 2 / 4 / 8 token chunks so that a chunk's GELU rows are read back from the
            // Infinity Cache -- MiniLM 256 x 256 6.41 -> 6.68 / 7.00 / 8.91 ms, bge-base 128 x 512 37.5 -> 38.5 / 42.5 / 45.6: gpurun_out/r6q5)
            if (gemm(6, xs, s3[14], (const float *)p[11], I, H, nullptr, fs)) return -10;
            if (fuse_ln) { if (gemm_ln(fs, s3[15], (const float *)p[13], (const float *)p[14], (const float *)p[15], I)) return -10; }
            else {
                if (gemm(5, fs, s3[15], (const float *)s3[18], Dn, I, y2, nullptr)) return -10;
                if (launch_add_ln_split(y2, Dn, x, T, H, (const float *)p[14], (const float *)p[15], c.ln_eps, x, xs, st)) return -10;
            }
        }
        k_pool<false><<<B, 256, 0, st>>>(x, nullptr, mask, S, H, pooling, normalise, out);
        AK_HIP(hipGetLastError());
        return 0;
    }
    if ((x3 || !scalar) && f32_mfma_supported(H, I, c.heads)) {
        for (int l = 0; l < L; l++) {
            const void *const *p = w + 5 + 16 * l;
            const uint16_t *const *s3 = x3 ? x3 + X3_SLOTS * l : nullptr;
            // matrix m of the layer (0 wq 1 wk 2 wv 3 wo 4 w1 5 w2; header slot 2 m for q / k / v, then 6, 10, 12). Split mode: the
            // q / k / v halves are ONE [3H][H] array each (s3[0], s3[1]: hi, lo; s3[2] the concatenated bias) -- one launch reads X once
            auto gemm = [&](int epi, const float *X, int m, int wslot, const float *bias, const float *R, int N, int K, float *Y, int ldc, int col0) -> int {
                if (s3) return launch_gemm_x3(epi, X, s3[2 * m], s3[2 * m + 1], bias, R, (int)T, N, K, Y, ldc, col0, st);
                return launch_gemm_f32(epi, X, (const float *)p[wslot], bias, R, (int)T, N, K, Y, ldc, col0, st);
            };
            if (s3) {
                if (launch_gemm_x3(0, x, s3[0], s3[1], (const float *)s3[2], nullptr, (int)T, 3 * H, H, qkv, 3 * H, 0, st)) return -10;
            } else {
                for (int j = 0; j < 3; j++)          // q, k, v straight into qkv[t] = q[t] | k[t] | v[t]
                    if (gemm(0, x, j, 2 * j, (const float *)p[2 * j + 1], nullptr, H, H, qkv, 3 * H, j * H)) return -10;
            }
            static const bool x3_attn_f32 = env_get("AK_X3_ATTN_F32") != nullptr;        // A/B: the float32 attention kernel under the split GEMMs
            if (s3 && !x3_attn_f32) { if (launch_attn_x3(qkv, mask, B, S, H, c.heads, ctx, st)) return -10; }
            else if (launch_attn_f32(qkv, mask, B, S, H, c.heads, ctx, st)) return -10;
            if (gemm(2, ctx, 3, 6, (const float *)p[7], x, H, H, y, H, 0)) return -10;      // + residual
            k32_add_ln<<<rows4, 256, 0, st>>>(y, nullptr, (int)T, H, (const float *)p[8], (const float *)p[9], c.ln_eps, x);
            AK_HIP(hipGetLastError());
            if (gemm(1, x, 4, 10, (const float *)p[11], nullptr, I, H, f, I, 0)) return -10;  // erf GELU
            if (gemm(2, f, 5, 12, (const float *)p[13], x, H, I, y, H, 0)) return -10;
            k32_add_ln<<<rows4, 256, 0, st>>>(y, nullptr, (int)T, H, (const float *)p[14], (const float *)p[15], c.ln_eps, x);
            AK_HIP(hipGetLastError());
        }
        k_pool<false><<<B, 256, 0, st>>>(x, nullptr, mask, S, H, pooling, normalise, out);
        AK_HIP(hipGetLastError());
        return 0;
    }
#if AK_DBG_KERNELS
    const dim3 gt((unsigned)((T + 63) / 64));
    for (int l = 0; l < L; l++) {
        const void *const *p = w + 5 + 16 * l;
        // q, k, v: three GEMMs (the caller's three weight matrices as they lie), then interleaved to qkv[t] = q[t] | k[t] | v[t]
        for (int j = 0; j < 3; j++)
            k32_gemm<false><<<dim3(H / 64, gt.x), 256, 0, st>>>((const float *)x, (const float *)p[2 * j], (const float *)p[2 * j + 1], (int)T, H, H, sep + (int64_t)j * T * H);
        AK_HIP(hipGetLastError());
        for (int j = 0; j < 3; j++)
            AK_HIP(hipMemcpy2DAsync(qkv + (int64_t)j * H, (size_t)3 * H * 4, sep + (int64_t)j * T * H, (size_t)H * 4, (size_t)H * 4, (size_t)T,
                                    hipMemcpyDeviceToDevice, st));
        k32_attn<<<(unsigned)(((int64_t)B * c.heads * S + 3) / 4), 256, 0, st>>>(qkv, mask, B, S, H, c.heads, ctx);
        k32_gemm<false><<<dim3(H / 64, gt.x), 256, 0, st>>>(ctx, (const float *)p[6], (const float *)p[7], (int)T, H, H, y);
        k32_add_ln<<<rows4, 256, 0, st>>>(y, x, (int)T, H, (const float *)p[8], (const float *)p[9], c.ln_eps, x);
        k32_gemm<true><<<dim3(I / 64, gt.x), 256, 0, st>>>(x, (const float *)p[10], (const float *)p[11], (int)T, I, H, f);
        k32_gemm<false><<<dim3(H / 64, gt.x), 256, 0, st>>>(f, (const float *)p[12], (const float *)p[13], (int)T, H, I, y);
        k32_add_ln<<<rows4, 256, 0, st>>>(y, x, (int)T, H, (const float *)p[14], (const float *)p[15], c.ln_eps, x);
        AK_HIP(hipGetLastError());
    }
    k_pool<false><<<B, 256, 0, st>>>(x, nullptr, mask, S, H, pooling, normalise, out);
    AK_HIP(hipGetLastError());
    return 0;
#else
    AK_FAIL(-1, "ak_encoder_forward (precision f32): hidden / intermediate sizes must be multiples of 128, head size 32 or 64");
#endif
}

}  // namespace ak

using namespace ak;

extern "C" int ak_encoder_create(const AkBertConfig *cfg, const void *const *w, int n_weights, ak_encoder_t *out) {
    AK_BIND();
    if (!cfg || !w || !out) AK_FAIL(-1, "ak_encoder_create: NULL argument");
    const int H = cfg->hidden, L = cfg->layers, I = cfg->intermediate;
    if (n_weights != 5 + 16 * L) AK_FAIL(-1, "ak_encoder_create: expected 5 + 16*layers weight pointers");
    // One shape rule for both precisions, enforced HERE: the float32 parity path (f32_mfma_supported) takes exactly the shapes
    // the bf16 path takes, so a config that passes create never fails on shape at its first forward (e.g. 312-d / 12-head
    // models are refused now, not then).
    if (H % 128 || I % 128 || H > 1024) AK_FAIL(-1, "ak_encoder_create: hidden/intermediate must be multiples of 128, hidden <= 1024");
    if (H % cfg->heads || (H / cfg->heads != 32 && H / cfg->heads != 64)) AK_FAIL(-1, "ak_encoder_create: head size must be 32 or 64");
    Encoder *e = new Encoder();
    e->cfg = *cfg;
    e->raw.assign(w, w + n_weights);
    if (cfg->precision == 1) {          // fp32 parity mode: the float32 matrices are used where they lie
        *out = e;
        return 0;
    }
    if (cfg->precision == 2) {          // split-bf16 parity mode: the matrices of every layer as bf16 hi + lo (one pass, here)
        // per layer 12 slots: [0] [1] hi / lo of the CONCATENATED q | k | v matrix [3H][H], [2] its concatenated bias [3H] float32,
        // [3]-[5] unused, then hi / lo of wo (m = 3), w1 (4), w2 (5) at [2 m], [2 m + 1]
        static const int slot[6] = {0, 2, 4, 6, 10, 12};
        auto fail = [&](const char *what) { set_error(what); ak_encoder_destroy(e); return -10; };
        for (int l = 0; l < L; l++) {
            const void *const *p = w + 5 + 16 * l;
            const int64_t hh = (int64_t)H * H;
            uint16_t *qhi = nullptr, *qlo = nullptr; float *qb = nullptr;
            if (hipMalloc((void **)&qhi, (size_t)3 * hh * 2) != hipSuccess) return fail("ak_encoder_create: hipMalloc failed");
            e->owned.push_back(qhi);
            if (hipMalloc((void **)&qlo, (size_t)3 * hh * 2) != hipSuccess) return fail("ak_encoder_create: hipMalloc failed");
            e->owned.push_back(qlo);
            if (hipMalloc((void **)&qb, (size_t)3 * H * 4) != hipSuccess) return fail("ak_encoder_create: hipMalloc failed");
            e->owned.push_back(qb);
            for (int j = 0; j < 3; j++) {
                if (split_hilo((const float *)p[slot[j]], hh, qhi + j * hh, qlo + j * hh, nullptr)) { ak_encoder_destroy(e); return -10; }
                if (hipMemcpy(qb + (size_t)j * H, p[2 * j + 1], (size_t)H * 4, hipMemcpyDeviceToDevice) != hipSuccess) return fail("ak_encoder_create: bias copy failed");
            }
            e->x3.push_back(qhi); e->x3.push_back(qlo); e->x3.push_back((const uint16_t *)qb);
            e->x3.push_back(nullptr); e->x3.push_back(nullptr); e->x3.push_back(nullptr);
            uint16_t *rows2[4] = {nullptr, nullptr, nullptr, nullptr};       // [hi | lo] rows of q | k | v, wo, w1, w2
            for (int m = 3; m < 6; m++) {
                const int64_t n = (int64_t)(m < 4 ? H : I) * H;
                uint16_t *hi = nullptr, *lo = nullptr;
                if (hipMalloc((void **)&hi, (size_t)n * 2) != hipSuccess) return fail("ak_encoder_create: hipMalloc failed");
                e->owned.push_back(hi);
                if (hipMalloc((void **)&lo, (size_t)n * 2) != hipSuccess) return fail("ak_encoder_create: hipMalloc failed");
                e->owned.push_back(lo);
                e->x3.push_back(hi); e->x3.push_back(lo);
                if (split_hilo((const float *)p[slot[m]], n, hi, lo, nullptr)) { ak_encoder_destroy(e); return -10; }
            }
            auto up256 = [](int64_t n) { return (n + 255) / 256 * 256; };
            const int64_t nel[4] = {up256(3 * H) * H, up256(H) * H, up256(I) * H, up256(H) * I};
            for (int m = 0; m < 4; m++) {
                if (hipMalloc((void **)&rows2[m], (size_t)nel[m] * 4) != hipSuccess) return fail("ak_encoder_create: hipMalloc failed");
                e->owned.push_back(rows2[m]);
                e->x3.push_back(rows2[m]);
                if (hipMemset(rows2[m], 0, (size_t)nel[m] * 4) != hipSuccess) return fail("ak_encoder_create: hipMemset failed");
            }
            const int64_t nb[3] = {up256(3 * H), up256(H), up256(H)};
            const void *bsrc[3] = {qb, p[7], p[13]};
            for (int m = 0; m < 3; m++) {
                float *bp = nullptr;
                if (hipMalloc((void **)&bp, (size_t)nb[m] * 4) != hipSuccess) return fail("ak_encoder_create: hipMalloc failed");
                e->owned.push_back(bp);
                e->x3.push_back((const uint16_t *)bp);
                if (hipMemset(bp, 0, (size_t)nb[m] * 4) != hipSuccess ||
                    hipMemcpy(bp, bsrc[m], (size_t)(m == 0 ? 3 * H : H) * 4, hipMemcpyDeviceToDevice) != hipSuccess) return fail("ak_encoder_create: bias copy failed");
            }
            e->x3.push_back(nullptr);
            for (int j = 0; j < 3; j++)
                if (split_rows((const float *)p[slot[j]], H, H, rows2[0] + (size_t)j * hh * 2, nullptr)) { ak_encoder_destroy(e); return -10; }
            if (split_rows((const float *)p[slot[3]], H, H, rows2[1], nullptr) || split_rows((const float *)p[slot[4]], I, H, rows2[2], nullptr) ||
                split_rows((const float *)p[slot[5]], H, I, rows2[3], nullptr)) { ak_encoder_destroy(e); return -10; }
        }
        if (hipDeviceSynchronize() != hipSuccess) { set_error("ak_encoder_create: weight split failed"); ak_encoder_destroy(e); return -10; }
        *out = e;
        return 0;
    }
    if (cfg->precision != 0) { delete e; AK_FAIL(-1, "ak_encoder_create: precision must be 0 (bf16), 1 (fp32 parity mode) or 2 (split-bf16 parity mode)"); }
    e->word = (const uint16_t *)w[0]; e->pos = (const uint16_t *)w[1]; e->type = (const uint16_t *)w[2];
    e->eg = (const float *)w[3]; e->eb = (const float *)w[4];
    for (int l = 0; l < L; l++) {
        const void *const *p = w + 5 + 16 * l;
        uint16_t *wqkv; float *bqkv;
        if (hipMalloc((void **)&wqkv, (size_t)3 * H * H * 2) != hipSuccess || hipMalloc((void **)&bqkv, (size_t)3 * H * 4) != hipSuccess) {
            set_error("ak_encoder_create: hipMalloc failed");
            ak_encoder_destroy(e);
            return -10;
        }
        e->owned.push_back(wqkv); e->owned.push_back(bqkv);
        for (int j = 0; j < 3; j++) {
            hipMemcpy(wqkv + (size_t)j * H * H, p[2 * j], (size_t)H * H * 2, hipMemcpyDeviceToDevice);
            hipMemcpy(bqkv + (size_t)j * H, p[2 * j + 1], (size_t)H * 4, hipMemcpyDeviceToDevice);
        }
        Layer ly;
        ly.wqkv = wqkv; ly.bqkv = bqkv;
        ly.wo = (const uint16_t *)p[6]; ly.bo = (const float *)p[7]; ly.ln1g = (const float *)p[8]; ly.ln1b = (const float *)p[9];
        ly.w1 = (const uint16_t *)p[10]; ly.b1 = (const float *)p[11]; ly.w2 = (const uint16_t *)p[12]; ly.b2 = (const float *)p[13];
        ly.ln2g = (const float *)p[14]; ly.ln2b = (const float *)p[15];
        if (ffn_fused_supported(H, I, 128)) {
            uint16_t *wbuf;
            if (hipMalloc((void **)&wbuf, ffn_weight_bytes(I)) != hipSuccess) {
                set_error("ak_encoder_create: hipMalloc failed");
                ak_encoder_destroy(e);
                return -10;
            }
            e->owned.push_back(wbuf);
            if (ffn_relayout(ly.wo, ly.w1, ly.w2, I, wbuf, &ly.wf, nullptr)) { ak_encoder_destroy(e); return -10; }
            ly.wof = wbuf;
            uint16_t *qbuf;
            if (hipMalloc((void **)&qbuf, qkv384_weight_bytes()) != hipSuccess) {
                set_error("ak_encoder_create: hipMalloc failed");
                ak_encoder_destroy(e);
                return -10;
            }
            e->owned.push_back(qbuf);
            if (qkv384_relayout(wqkv, bqkv, qbuf, nullptr)) { ak_encoder_destroy(e); return -10; }
            ly.wq16 = qbuf;
        } else if (H % 256 == 0 && I % 256 == 0) {
            // lazy LayerNorm: W gamma and b + W beta of the matrices that read a LayerNorm's output (gemm.hip)
            auto dev = [&](size_t bytes) { void *q = nullptr; if (hipMalloc(&q, bytes) != hipSuccess) return (void *)nullptr; e->owned.push_back(q); return q; };
            float *c1 = (float *)dev((size_t)I * 4), *b1f = (float *)dev((size_t)I * 4);
            if (!c1 || !b1f || launch_fold_ln(ly.w1, ly.ln1g, ly.ln1b, ly.b1, I, H, c1, b1f, nullptr)) {
                set_error("ak_encoder_create: lazy LayerNorm terms"); ak_encoder_destroy(e); return -10;
            }
            ly.c1 = c1; ly.b1_f = b1f;
            if (l > 0) {
                const Layer &pv = e->layers[l - 1];
                float *cq = (float *)dev((size_t)3 * H * 4), *bqf = (float *)dev((size_t)3 * H * 4);
                if (!cq || !bqf || launch_fold_ln(wqkv, pv.ln2g, pv.ln2b, bqkv, 3 * H, H, cq, bqf, nullptr)) {
                    set_error("ak_encoder_create: lazy LayerNorm terms"); ak_encoder_destroy(e); return -10;
                }
                ly.cqkv = cq; ly.bqkv_f = bqf;
            }
        }
        e->layers.push_back(ly);
    }
    hipDeviceSynchronize();
    *out = e;
    return 0;
}

extern "C" int ak_encoder_destroy(ak_encoder_t h) {
    AK_BIND();
    if (!h) return 0;
    Encoder *e = (Encoder *)h;
    hipDeviceSynchronize();
    free_ws(*e);
    if (e->lens_ids) hipFree(e->lens_ids);
    if (e->lens_mask) hipFree(e->lens_mask);
    if (e->ws32) hipFree(e->ws32);
    if (e->qf_ctl) hipFree(e->qf_ctl);
    if (e->qf_layers) hipFree(e->qf_layers);
    if (e->qf_fail) hipHostFree(e->qf_fail);
    for (void *p : e->owned) hipFree(p);
    delete e;
    return 0;
}

// ids / mask [B][S] -> out [B][H]; the caller holds e.mu
// right-padded rows given by their lengths (ak_encoder_forward_lens): ids [B] rows `ld` apart, lens `stride` apart
struct LensIn { const int32_t *ids; int ld; const int32_t *lens; int stride; };
static int forward_locked(Encoder &e, const int32_t *ids, const int32_t *mask, int B, int S, int pooling, int normalise, float *out, hipStream_t st,
                          const LensIn *lens_in = nullptr);

extern "C" int ak_encoder_forward(ak_encoder_t h, const int32_t *ids, const int32_t *mask, int B, int S, int pooling,
                                  int normalise, float *out, void *stream) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_encoder_forward: NULL encoder");
    RoctxRange range("ak_encoder_forward");
    Encoder &e = *(Encoder *)h;
    if (B <= 0) return 0;
    if (S % 32 || S > 512 || S > e.cfg.max_position) AK_FAIL(-1, "ak_encoder_forward: S must be a multiple of 32, <= 512 and <= max_position (pad with mask 0)");
    std::lock_guard<std::mutex> lk(e.mu);
    return forward_locked(e, ids, mask, B, S, pooling, normalise, out, (hipStream_t)stream);
}

namespace ak {
// one thread per token slot: the tile's contiguous ids (zero past a row's length) and its 0 / 1 mask
__global__ void k_mask_from_lens(const int *__restrict__ ids, int ld_ids, const int *__restrict__ lens, int lens_stride, int B, int S,
                                 int *__restrict__ oids, int *__restrict__ omask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * S) return;
    const int b = (int)(i / S), t = (int)(i - (int64_t)b * S);
    int len = lens[(int64_t)b * lens_stride];
    len = len < 0 ? 0 : (len > S ? S : len);
    const bool live = t < len;
    oids[i] = live ? ids[(int64_t)b * ld_ids + t] : 0;
    omask[i] = live ? 1 : 0;
}
}  // namespace ak

extern "C" int ak_encoder_forward_lens(ak_encoder_t h, const int32_t *ids, int ld_ids, const int32_t *lens, int lens_stride, int B, int S,
                                       int pooling, int normalise, float *out, void *stream) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_encoder_forward_lens: NULL encoder");
    RoctxRange range("ak_encoder_forward_lens");
    Encoder &e = *(Encoder *)h;
    if (B <= 0) return 0;
    if (!ids || !lens || !out || ld_ids < S || lens_stride < 1) AK_FAIL(-1, "ak_encoder_forward_lens: bad arguments");
    if (S % 32 || S > 512 || S > e.cfg.max_position) AK_FAIL(-1, "ak_encoder_forward_lens: S must be a multiple of 32, <= 512 and <= max_position");
    std::lock_guard<std::mutex> lk(e.mu);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * S;
    if (n > e.lens_cap) {
        if (e.lens_ids) hipFree(e.lens_ids);
        if (e.lens_mask) hipFree(e.lens_mask);
        e.lens_ids = e.lens_mask = nullptr; e.lens_cap = 0;
        const int64_t cap = n < (1 << 16) ? (1 << 16) : n + n / 4;
        AK_HIP(hipMalloc((void **)&e.lens_ids, (size_t)cap * 4));
        AK_HIP(hipMalloc((void **)&e.lens_mask, (size_t)cap * 4));
        e.lens_cap = cap;
    }
    const LensIn li{ids, ld_ids, lens, lens_stride};      // (the ids / mask lay-out launch is forward_locked's: the single-launch query forward does it itself)
    return forward_locked(e, e.lens_ids, e.lens_mask, B, S, pooling, normalise, out, st, &li);
}

// The single-launch forward pass of <= 64 token rows (query_forward.hip). 0 = done (the stream is synchronised); 1 = not taken or
// failed safely (a bounded wait gave up): the caller runs the multi-launch path; < 0 = error.
static int run_query_forward(Encoder &e, const LensIn *lens_in, const int32_t *ids, const int32_t *mask, int B, int S, int pooling, int normalise,
                             float *out, int64_t tpad, hipStream_t st) {
    const int H = e.cfg.hidden, I = e.cfg.intermediate, heads = e.cfg.heads, L = e.cfg.layers;
    const int64_t T = (int64_t)B * S;
    if (!e.qf_ctl) {
        AK_HIP(hipMalloc((void **)&e.qf_ctl, sizeof(QfCtl)));
        AK_HIP(hipMemset(e.qf_ctl, 0, sizeof(QfCtl)));
        AK_HIP(hipHostMalloc((void **)&e.qf_fail, 64));
        *e.qf_fail = 0;
        std::vector<QfLayer> tab;
        for (const Layer &ly : e.layers)
            tab.push_back(QfLayer{ly.wqkv, ly.bqkv, ly.wo, ly.bo, ly.ln1g, ly.ln1b, ly.w1, ly.b1, ly.w2, ly.b2, ly.ln2g, ly.ln2b});
        AK_HIP(hipMalloc((void **)&e.qf_layers, tab.size() * sizeof(QfLayer)));
        AK_HIP(hipMemcpy(e.qf_layers, tab.data(), tab.size() * sizeof(QfLayer), hipMemcpyHostToDevice));
        e.qf_epoch = 0;
    }
    QfArgs a{};
    if (lens_in) { a.ids_in = lens_in->ids; a.ld_ids = lens_in->ld; a.lens = lens_in->lens; a.lens_stride = lens_in->stride; }
    a.ids = ids; a.mask = mask; a.oids = e.lens_ids; a.omask = e.lens_mask;
    a.B = B; a.S = S; a.T = (int)T; a.t32 = (int)((T + 31) / 32 * 32); a.H = H; a.I = I; a.heads = heads; a.L = L; a.vocab = e.cfg.vocab_size;
    a.eps = e.cfg.ln_eps; a.qscale = 1.4426950408889634f / sqrtf((float)(H / heads));
    a.word = e.word; a.pos = e.pos; a.type = e.type; a.eg = e.eg; a.eb = e.eb;
    a.x32 = e.cfg.residual_bf16 ? nullptr : e.x32; a.y32 = e.y32; a.x16 = e.x16; a.q = e.q; a.k = e.k; a.vt = e.vt; a.ctx = e.ctx; a.f = e.f;
    a.maskf = e.maskf; a.blkmask = (uint32_t *)(e.maskf + tpad);
    a.layers = e.qf_layers; a.pooling = pooling; a.normalise = normalise; a.out = out;
    a.ctl = e.qf_ctl; a.fail = e.qf_fail; a.epoch = e.qf_epoch++;
    a.dbg_skip = dbg_env_int("AK_QF_SKIP", 0);
    if (launch_query_forward(a, st)) return -10;
    AK_HIP(hipStreamSynchronize(st));                   // the failure word is read here: this path is synchronous (the caller copies the row out next anyway)
    if (*(volatile unsigned *)e.qf_fail == 0) return 0;
    // a bounded wait gave up (workgroups of the launch did not all arrive in time): nothing is left running. Reset the barrier state,
    // take the multi-launch path for this call, and stop using the single launch on this encoder after three such calls.
    *e.qf_fail = 0;
    AK_HIP(hipMemset(e.qf_ctl, 0, sizeof(QfCtl)));
    e.qf_epoch = 0;
    static std::atomic<int> gave_up{0};
    if (++gave_up >= 3) e.qf_off = true;
    return 1;
}

static int forward_locked(Encoder &e, const int32_t *ids, const int32_t *mask, int B, int S, int pooling, int normalise, float *out, hipStream_t st,
                          const LensIn *lens_in) {
    // lay the ids / 0-1 mask of right-padded rows out (the single-launch query forward below does it inside its one launch instead)
    auto lens_to_mask = [&]() -> int {
        if (!lens_in) return 0;
        const int64_t n = (int64_t)B * S;
        k_mask_from_lens<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(lens_in->ids, lens_in->ld, lens_in->lens, lens_in->stride, B, S, e.lens_ids, e.lens_mask);
        AK_HIP(hipGetLastError());
        lens_in = nullptr;
        return 0;
    };
    if (e.cfg.precision == 1 || e.cfg.precision == 2) {
        if (lens_to_mask()) return -10;
        return forward_f32(e.cfg, e.raw.data(), e.cfg.precision == 2 ? e.x3.data() : nullptr, ids, mask, B, S, pooling, normalise, out,
                           &e.ws32, &e.ws32_bytes, st);
    }
    const int H = e.cfg.hidden, I = e.cfg.intermediate, heads = e.cfg.heads;
    const int64_t T = (int64_t)B * S, tpad = (T + 255) / 256 * 256;
    if (reserve_ws(e, tpad)) return -10;
    // <= 64 token rows (embed_query): the whole forward pass as ONE launch on one XCD (query_forward.hip), bit-identical to the
    // launches below; 1 = not taken / gave up safely
    const int qf_mode = switches().query_fused.load(std::memory_order_relaxed);      // 0 off (default), 1 when it applies, 2 required (tests)
    if (!e.qf_off && query_forward_supported(H, I, heads, T, S) && (lens_in == nullptr || e.lens_ids != nullptr)) {
        const int rc = run_query_forward(e, lens_in, ids, mask, B, S, pooling, normalise, out, tpad, st);
        if (rc <= 0) return rc;
        if (qf_mode == 2) AK_FAIL(-10, "ak_encoder_forward: the single-launch query forward gave up (a bounded wait ran out) and AK_QUERY_FUSED=2 requires it");
    } else if (qf_mode == 2 && T <= 64) {
        AK_FAIL(-1, "ak_encoder_forward: AK_QUERY_FUSED=2 requires the single-launch query forward, which does not take this shape / encoder");
    }
    if (lens_to_mask()) return -10;
    const float eps = e.cfg.ln_eps;
    // H = 384: residual add + LayerNorm run in the epilogue of the GEMM that feeds them (gemm_ln.hip)
    static const bool nofuse = env_get("AK_ENC_NOFUSE") != nullptr;
    const bool fuse = !nofuse && gemm_ln_supported(H, tpad, H) && gemm_ln_supported(H, tpad, I);
    const bool r16 = e.cfg.residual_bf16 != 0;           // bf16-only residual stream: x32 is not used at all
    float *x32 = r16 ? nullptr : e.x32;
    // a few token rows (embed_query: one 32-token tile): output- and K-parallel GEMMs + a LayerNorm kernel instead of
    // the 128-token-tile kernels, whose K walk would be the whole cost (gemm_skinny.hip)
    // Crossover, measured (forward ms, tile path / these kernels): hidden 384 -- the fused-layer path has a floor of ~0.67 ms
    // (a tile walks the whole weight ring whatever its token count; 0.87 ms before the 64-token tiles) -- 1024 tokens 0.66 / 0.36,
    // 2048 0.67 / 0.52, 2560 0.67 / 0.62, 3072 0.67 / 0.70, 4096 0.68 / 0.83; hidden 768: 256 tokens 1.28 / 0.76, 512 3.09 / 1.09,
    // 768 1.30 / 1.43, 1024 1.32 / 1.68. One embed call per file (the reference's manager.py:373) lands exactly here.
    static const int skinny_env = env_get("AK_ENC_SKINNY_MAX") ? atoi(env_get("AK_ENC_SKINNY_MAX")) : -1;
    const int skinny_max = skinny_env >= 0 ? skinny_env : (H == 384 ? 2816 : 640);
    const bool skinny = T <= skinny_max && gemm_skinny_supported(H, H) && gemm_skinny_supported(H, I) &&
                        gemm_skinny_supported(I, H);
    const int t32 = (int)((T + 31) / 32 * 32);
    // unfused GEMM -> LayerNorm path (hidden != 384) in bf16-residual mode: the GEMM output travels as bf16 too
    static const bool y32_forced = env_get("AK_ENC_Y32") != nullptr;
    const bool y16 = r16 && !y32_forced;
    if (launch_embed(H / 128, ids, (int)T, S, e.cfg.vocab_size, e.word, e.pos, e.type, e.eg, e.eb, eps, x32, e.x16, st)) AK_FAIL(-1, "ak_encoder_forward: hidden size");
    AK_HIP(hipGetLastError());
    if (launch_attn_prepare(mask, B, S, e.maskf, (uint32_t *)(e.maskf + tpad), st)) return -10;
    static const bool head_major = !env_get("AK_QK_TOKEN_MAJOR");     // A/B: q / k of the hidden-384 path as [T][384]
    // LAZY LayerNorm (gemm.hip): no LayerNorm launch between the sub-layers. e.q holds the gamma-scaled rows behind the attention
    // block (gamma1 (.) r1, statistics of r1 in st1), e.x16 those behind the feed-forward block (gamma2 (.) r2, st2); in layer 0 e.x16
    // is k_embed's normalised output.
    const bool lazy = y16 && !x32 && !skinny && !fuse && e.st1 && e.layers[0].c1 && gemm_lazy_supported(tpad, H, I);
    if (lazy) {
        const int L = (int)e.layers.size();
        for (int l = 0; l < L; l++) {
            const Layer &ly = e.layers[l];
            GemmArgs g{};
            g.X = e.x16; g.T = (int)tpad; g.N = 3 * H; g.K = H;
            g.q = e.q; g.k = e.k; g.vt = e.vt; g.H = H; g.S = S; g.qscale = 1.4426950408889634f / sqrtf((float)(H / heads));
            g.ldo = (int)T;
            g.nslot = H / 128; g.inv_h = 1.0f / (float)H; g.eps = eps;
            if (l == 0) { g.W = ly.wqkv; g.bias = ly.bqkv; if (launch_gemm(0, g, st)) return -10; }
            else { g.W = ly.wqkv; g.bias = ly.bqkv_f; g.fold_c = ly.cqkv; g.a_stats = e.st2; if (launch_gemm_lazy(0, g, st)) return -10; }
            AttnArgs a{e.q, e.k, e.vt, mask, e.ctx, B, S, H, heads, e.maskf, (const uint32_t *)(e.maskf + tpad), 0, 0};
            if (launch_attn(a, st)) return -10;
            GemmArgs o{};                         // r1 = ctx Wo^T + bo + LN2_prev(r2) -> e.q, st1
            o.X = e.ctx; o.W = ly.wo; o.bias = ly.bo; o.T = (int)tpad; o.N = H; o.K = H; o.out_bf16 = e.q; o.ldo = H; o.res16 = e.x16;
            o.nslot = H / 128; o.inv_h = 1.0f / (float)H; o.eps = eps; o.out_stats = e.stp; o.out_g = ly.ln1g;
            if (l > 0) { o.res_stats = e.st2; o.res_g = e.layers[l - 1].ln2g; o.res_b = e.layers[l - 1].ln2b; }
            if (launch_gemm_lazy(4, o, st)) return -10;
            if (launch_ln_finalize(e.stp, H / 128, tpad, 1.0f / (float)H, eps, e.st1, st)) return -10;
            GemmArgs f1{};                        // f = gelu(LN1(r1) W1^T + b1)
            f1.X = e.q; f1.W = ly.w1; f1.bias = ly.b1_f; f1.fold_c = ly.c1; f1.a_stats = e.st1; f1.T = (int)tpad; f1.N = I; f1.K = H;
            f1.out_bf16 = e.f; f1.ldo = I; f1.nslot = H / 128; f1.inv_h = 1.0f / (float)H; f1.eps = eps;
            if (launch_gemm_lazy(1, f1, st)) return -10;
            GemmArgs f2{};                        // r2 = f W2^T + b2 + LN1(r1) -> e.x16, st2
            f2.X = e.f; f2.W = ly.w2; f2.bias = ly.b2; f2.T = (int)tpad; f2.N = H; f2.K = I; f2.out_bf16 = e.x16; f2.ldo = H; f2.res16 = e.q;
            f2.nslot = H / 128; f2.inv_h = 1.0f / (float)H; f2.eps = eps; f2.out_stats = e.stp; f2.out_g = ly.ln2g;
            f2.res_stats = e.st1; f2.res_g = ly.ln1g; f2.res_b = ly.ln1b;
            if (launch_gemm_lazy(4, f2, st)) return -10;
            if (launch_ln_finalize(e.stp, H / 128, tpad, 1.0f / (float)H, eps, e.st2, st)) return -10;
        }
        // the last LayerNorm is a launch: the pooling kernel reads normalised rows (e.q is free again)
        const Layer &last = e.layers[L - 1];
        k_ln_apply16<<<(unsigned)((T * (H / 8) + 255) / 256), 256, 0, st>>>(e.x16, e.st2, last.ln2g, last.ln2b, T, H, e.q);
        AK_HIP(hipGetLastError());
        k_pool<true><<<B, 256, 0, st>>>(nullptr, e.q, mask, S, H, pooling, normalise, out);
        AK_HIP(hipGetLastError());
        return 0;
    }
    for (const Layer &ly : e.layers) {
        bool qk_head_major = false;
        GemmArgs g{};
        g.X = e.x16; g.W = ly.wqkv; g.bias = ly.bqkv; g.T = (int)tpad; g.N = 3 * H; g.K = H;
        g.q = e.q; g.k = e.k; g.vt = e.vt; g.H = H; g.S = S; g.qscale = 1.4426950408889634f / sqrtf((float)(H / heads));   // log2(e)/sqrt(hd): attention.hip exponentiates with 2^x
        g.ldo = (int)T;   // MODE 0: number of real tokens (rows beyond it have no V^T slot)
        if (skinny && gemm_skinny_supported(3 * H, H)) {
            if (launch_gemm_skinny_qkv(e.x16, ly.wqkv, ly.bqkv, t32, H, H, e.q, e.k, e.vt, S, (int)T, g.qscale, st)) return -10;
        } else if (r16 && ly.wq16 && qkv384_supported(H, tpad, S)) {
            QkvArgs qa{e.x16, ly.wq16, nullptr, e.q, e.k, e.vt, (int)tpad, (int)T, S, g.qscale, 0, head_major ? 1 : 0};
            if (launch_qkv384(qa, st)) return -10;
            qk_head_major = head_major;
        } else if (launch_gemm(0, g, st)) return -10;
        AttnArgs a{e.q, e.k, e.vt, mask, e.ctx, B, S, H, heads, e.maskf, (const uint32_t *)(e.maskf + tpad),
                   qk_head_major ? H / heads : 0, qk_head_major ? S * (H / heads) : 0};
        if (launch_attn(a, st)) return -10;
        static const bool noffn = dbg_env_int("AK_ENC_NOFFN", 0) != 0;
        const bool ffn_fused = !skinny && fuse && r16 && ly.wf && !noffn && ffn_fused_supported(H, I, tpad);
        if (ffn_fused && ffn_fuses_attention_out()) {
            // attention out-projection + residual + LayerNorm-1 + feed-forward block + residual + LayerNorm-2: ONE launch
            FfnArgs fa{e.x16, ly.wf, ly.b1, ly.b2, ly.ln2g, ly.ln2b, e.ctx, ly.wof, ly.bo, ly.ln1g, ly.ln1b, (int)tpad, I, eps, nullptr};
            if (launch_ffn384(fa, st)) return -10;
            continue;
        }
        if (skinny) {
            if (launch_gemm_skinny(e.ctx, ly.wo, ly.bo, t32, H, H, e.y32, nullptr, 0, st)) return -10;
            k_layernorm<<<(unsigned)((T + 3) / 4), 256, 0, st>>>(e.y32, x32, r16 ? e.x16 : nullptr, ly.ln1g, ly.ln1b, (int)T, H, eps, x32, e.x16);
        } else if (fuse) {
            GemmLnArgs o{e.ctx, ly.wo, ly.bo, ly.ln1g, ly.ln1b, x32, e.x16, (int)tpad, H, eps, nullptr};
            if (launch_gemm_ln(o, st)) return -10;
        } else {
            GemmArgs o{};
            o.X = e.ctx; o.W = ly.wo; o.bias = ly.bo; o.T = (int)tpad; o.N = H; o.K = H; o.out_f32 = e.y32; o.res_f32 = e.x32;
            o.out_bf16 = e.q; o.ldo = H;       // y16: the Q buffer is free once attention has run
            o.res16 = e.x16;                   // ... and the residual is added in the GEMM's store pass (MODE 4): the LayerNorm reads one array
            if (launch_gemm(y16 ? 4 : 2, o, st)) return -10;
            if (!(y16 && !x32 && H % 128 == 0 && launch_layernorm16(H / 128, e.q, ly.ln1g, ly.ln1b, (int)T, eps, e.x16, st)))
                k_layernorm<<<(unsigned)((T + 3) / 4), 256, 0, st>>>(e.y32, x32, (r16 && !y16) ? e.x16 : nullptr, ly.ln1g, ly.ln1b, (int)T, H, eps, x32, e.x16, y16 ? e.q : nullptr);
        }
        GemmArgs f1{};
        f1.X = e.x16; f1.W = ly.w1; f1.bias = ly.b1; f1.T = (int)tpad; f1.N = I; f1.K = H; f1.out_bf16 = e.f; f1.ldo = I;
        if (skinny) {
            if (launch_gemm_skinny(e.x16, ly.w1, ly.b1, t32, I, H, nullptr, e.f, I, st)) return -10;
            if (launch_gemm_skinny(e.f, ly.w2, ly.b2, t32, H, I, e.y32, nullptr, 0, st)) return -10;
            k_layernorm<<<(unsigned)((T + 3) / 4), 256, 0, st>>>(e.y32, x32, r16 ? e.x16 : nullptr, ly.ln2g, ly.ln2b, (int)T, H, eps, x32, e.x16);
            AK_HIP(hipGetLastError());
            continue;
        }
        if (ffn_fused) {
            // up-projection + GELU + down-projection + residual + LayerNorm in one launch: the [T][I] intermediate stays in registers
            FfnArgs fa{e.x16, ly.wf, ly.b1, ly.b2, ly.ln2g, ly.ln2b, nullptr, nullptr, nullptr, nullptr, nullptr, (int)tpad, I, eps, nullptr};
            if (launch_ffn384(fa, st)) return -10;
            continue;
        }
        if (launch_gemm(1, f1, st)) return -10;
        if (fuse) {
            GemmLnArgs f2{e.f, ly.w2, ly.b2, ly.ln2g, ly.ln2b, x32, e.x16, (int)tpad, I, eps, nullptr};
            if (launch_gemm_ln(f2, st)) return -10;
        } else {
            GemmArgs f2{};
            f2.X = e.f; f2.W = ly.w2; f2.bias = ly.b2; f2.T = (int)tpad; f2.N = H; f2.K = I; f2.out_f32 = e.y32; f2.res_f32 = e.x32;
            f2.out_bf16 = e.q; f2.ldo = H; f2.res16 = e.x16;
            if (launch_gemm(y16 ? 4 : 2, f2, st)) return -10;
            if (!(y16 && !x32 && H % 128 == 0 && launch_layernorm16(H / 128, e.q, ly.ln2g, ly.ln2b, (int)T, eps, e.x16, st)))
                k_layernorm<<<(unsigned)((T + 3) / 4), 256, 0, st>>>(e.y32, x32, (r16 && !y16) ? e.x16 : nullptr, ly.ln2g, ly.ln2b, (int)T, H, eps, x32, e.x16, y16 ? e.q : nullptr);
        }
        AK_HIP(hipGetLastError());
    }
    if (x32) k_pool<false><<<B, 256, 0, st>>>(x32, e.x16, mask, S, H, pooling, normalise, out);
    else k_pool<true><<<B, 256, 0, st>>>(x32, e.x16, mask, S, H, pooling, normalise, out);
    AK_HIP(hipGetLastError());
    return 0;
}
