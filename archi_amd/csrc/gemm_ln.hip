// gemm_ln.hip -- GEMM with the residual add and LayerNorm fused into its epilogue, for hidden size 384
// (all-MiniLM-L6, the reference's default embedder: src/cli/templates/base-config.yaml:145):
//   y = LayerNorm(X W^T + bias + res) * gamma + beta      X: [T][K] bf16, W: [384][K] bf16
//   res / y32: [T][384] fp32 (in place: the residual stream), y16: [T][384] bf16 (the next GEMM's input)
// Used for the attention output projection (K = 384) and the FFN down-projection (K = 1536).
//
// Why: as two kernels the fp32 GEMM output is written (100 MB at 65 536 tokens) and read back by the LayerNorm,
// 2.4 GB per forward pass of a 10.5 GB total, on a path that is about half HBM-traffic-bound (docs/EXPERIMENTS.md).
// LayerNorm needs every feature of a token, so the tile is ALL 384 features x 128 tokens: 8 waves as
// 4 (96 features = 3 MFMA row blocks) x 2 (64 tokens = 2 column blocks), 96 accumulators per lane; a lane owns
// one token column per column block, so the row statistics are a lane-local sum, one lane^32 exchange and a
// 4-way cross-wave sum through LDS. Ring: 2 slots x (384 + 128) rows x 128 B = 128 KB.
//
// Epilogue memory access. In the accumulator layout a lane owns a token and 4-feature groups, so a wave's load or
// store instruction touches 32 token rows with 16-32 bytes each: every 128-byte line of the residual / output is
// moved in four partial requests, and the first version of this kernel ran its 300 MB epilogue at 3.3 TB/s
// (the stand-alone LayerNorm kernel: 6.2 TB/s). So each 32-feature x 32-token block is transposed through a
// wave-private 4.5 KB LDS scratch (inside the wave's own staging pieces of the ring slot that was just consumed):
// residual rows are read, and output rows written, as full 128-byte lines (8 lanes x 16 B per token row).
// Measured and not kept: a 64-token tile whose freed registers hold the whole residual, requested before the K-loop
// (85.7 us for the K = 384 projection against 85.4 us for this version): the short-K launch is the serial sum of an
// exposed-latency K-loop (2-slot ring, 12 steps per CU) and a ~300 MB epilogue, and neither moved.
#include <atomic>
#include "mfma_tile.h"
#include "encoder_kernels.h"
#include "switches.h"

#include <cstdio>
#include <vector>

namespace ak {
using namespace mt;

// K-step 32 (64-byte LDS rows), 4-slot ring: with 64-deep steps only two 64 KB slots fit and every step waited for a
// load issued one step earlier; 32 KB slots give a ring of four, i.e. loads issued three steps ahead.
constexpr int L_H = 384, L_BT = 128, L_NW = 8, L_THREADS = 512, L_NST = 4, L_LOADS = 4;
constexpr int L_W_BYTES = L_H * 64, L_X_BYTES = L_BT * 64;           // per K-step of 32
constexpr int L_SLOT = L_W_BYTES + L_X_BYTES;                        // 32 KB
static_assert(L_H / 16 / L_NW == 3 && L_BT / 16 / L_NW == 1, "staging: 3 W pieces + 1 X piece (16 rows x 64 B) per wave per K-step");
constexpr int L_LDS = L_NST * L_SLOT + (2 * 4 * L_BT + 3 * L_H) * 4; // + partial sums [2][4][128] + bias/gamma/beta


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// RES16: the residual stream lives in bf16 only (x16 is both the residual read here and the next GEMM's input; x32 is
// not touched): the epilogue moves 196 KB per tile instead of 490 KB. See AkBertConfig.residual_bf16.
// X3 (round 6): the split-bf16 parity mode (precision 2) on this tile, as gemm.hip's MODE 5 / 6: X and W are bf16 rows
// [hi(K') | lo(K')] of float32 values, a.K = 3 K' and the K-loop walks X hi.W hi, X lo.W hi, X hi.W lo (the k-tile index wraps per
// operand); the residual stream is float32 (x32, in place) and x16 receives the LayerNorm's output SPLIT, rows [hi(384) | lo(384)]:
// the next GEMM's operand. Replaces MODE 5 + k3_add_ln for the hidden-384 out-projection and FFN-down (y never goes to HBM).
template <bool RES16, bool X3 = false>
__global__ __launch_bounds__(L_THREADS, 2) void k_gemm_ln(GemmLnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_part = (float *)(smem + L_NST * L_SLOT);      // [2][4][128]: sums, centred squares
    float *s_bias = s_part + 2 * 4 * L_BT, *s_gamma = s_bias + L_H, *s_beta = s_gamma + L_H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;               // 4 (features) x 2 (tokens)
    const int ntiles = a.T / L_BT, KS = a.K / 32;
    const int ldk = X3 ? a.K / 3 * 2 : a.K, KS3 = KS / 3;
    // step kk of the walk = term kk % 3 of k-tile kk / 3: (X hi, W hi), (X lo, W hi), (X hi, W lo) -- the two uses of a half are at most
    // two steps apart, so the second comes from L2 (walking the three terms as three passes over K fetched every hi half twice from
    // the fabric: PMC, bge-base FFN-down 2.3 GB per launch for 0.8 GB of operands)
    auto kt_w = [&](int kk) { if (!X3) return kk; const int q = kk / 3; return q + (kk - 3 * q == 2 ? KS3 : 0); };
    auto kt_x = [&](int kk) { if (!X3) return kk; const int q = kk / 3; return q + (kk - 3 * q == 1 ? KS3 : 0); };
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nsteps = my_tiles * KS;

    for (int i = tid; i < L_H; i += L_THREADS) { s_bias[i] = a.bias[i]; s_gamma[i] = a.gamma[i]; s_beta[i] = a.beta[i]; }

    // LDS rows are 64 B = four 16-byte chunks; chunk c of row q is stored at c ^ ((q >> 2) & 3) (applied on the SOURCE
    // address of the LDS-DMA), which spreads the 16 lanes of a ds_read_b128 group over all 16 bank quads.
    const int r = lane & 31, kh = lane >> 5;
    const int c0 = kh ^ ((r >> 2) & 3);
    const int a_off = (wr * 96 + r) * 64;                  // + mi*32*64
    const int b_off = L_W_BYTES + (wc * 64 + r) * 64;      // + ni*32*64   (X rows follow the W rows in a slot)

    // staging: per K-step a wave copies 3 pieces (16 rows x 64 B) of W and 1 of X; W rows are the same for every tile
    const int st_row = lane >> 2, st_chunk = lane & 3;
    const uint32_t lds0 = lds_addr(smem);
    const char *wptr2[2], *wptr1[1], *xptr[1];
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const int row = (wave * 3 + p) * 16 + st_row;
        const char *g = (const char *)a.W + ((int64_t)row * ldk) * 2 + ((st_chunk ^ ((row >> 2) & 3)) << 4);
        if (p < 2) wptr2[p] = g; else wptr1[0] = g;
    }
    int s_t = 0, s_kk = 0, s_buf = 0, issued = 0;
    auto set_xptr = [&](int ord) {
        int tile = blockIdx.x + ord * gridDim.x;
        if (tile >= ntiles) tile = ntiles - 1;
        const int row = wave * 16 + st_row;
        xptr[0] = (const char *)a.X + ((int64_t)(tile * L_BT + row) * ldk) * 2 + ((st_chunk ^ ((row >> 2) & 3)) << 4);
    };
    set_xptr(0);
    auto stage_next = [&]() {
        const int goff = kt_w(s_kk) * 64;
        const uint32_t base = lds0 + s_buf * L_SLOT;
        glds16xN<2>(wptr2, goff, __builtin_amdgcn_readfirstlane(base + wave * 3 * 1024));
        glds16xN<1>(wptr1, goff, __builtin_amdgcn_readfirstlane(base + (wave * 3 + 2) * 1024));
        glds16xN<1>(xptr, kt_x(s_kk) * 64, __builtin_amdgcn_readfirstlane(base + L_W_BYTES + wave * 1024));
        s_buf = (s_buf + 1) & (L_NST - 1);
        if (++s_kk == KS) { s_kk = 0; s_t++; set_xptr(s_t); }
        issued++;
    };

    f32x16 acc[3][2];
    auto compute = [&](int cur, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const char *buf = smem + cur * L_SLOT;
        uint4 av[2][3], bv[2][2];
        auto load_frags = [&](int k2, uint4 (&a3)[3], uint4 (&b2)[2]) {
            const int coff = (c0 ^ (k2 << 1)) << 4;
#pragma unroll
            for (int ni = 0; ni < 2; ni++) b2[ni] = *(const uint4 *)(buf + b_off + ni * 2048 + coff);
#pragma unroll
            for (int mi = 0; mi < 3; mi++) a3[mi] = *(const uint4 *)(buf + a_off + mi * 2048 + coff);
        };
        load_frags(0, av[0], bv[0]);
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
            if (k2 < 1) load_frags(k2 + 1, av[(k2 + 1) & 1], bv[(k2 + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);   // keep the fragment prefetch above the MFMAs
#pragma unroll
            for (int mi = 0; mi < 3; mi++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++) {
                    if (FIRST && k2 == 0) {
                        f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[mi][ni] = mfma_bf16(av[k2 & 1][mi], bv[k2 & 1][ni], z);
                    } else {
                        acc[mi][ni] = mfma_bf16(av[k2 & 1][mi], bv[k2 & 1][ni], acc[mi][ni]);
                    }
                }
        }
    };

#pragma unroll
    for (int i = 0; i < L_NST - 1; i++)
        if (issued < nsteps) stage_next();
    if (issued >= 3) wait_vm<2 * L_LOADS>(); else if (issued == 2) wait_vm<L_LOADS>(); else wait_vm<0>();
    __syncthreads();

    int cur = 0, step = 0;
    long long t_loop = 0, t_epi = 0, t_mark = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
    for (int ord = 0; ord < my_tiles; ord++) {
        const int tile = blockIdx.x + ord * gridDim.x;
        for (int kk = 0; kk < KS; kk++, step++) {
            if (issued < nsteps) stage_next();
            if (kk == 0) compute(cur, std::true_type{}); else compute(cur, std::false_type{});
            // the NEXT step's slot must have landed; the stages issued after it (up to two) may stay in flight
            const int ahead = issued - (step + 2);
            if (ahead >= 2) wait_vm<2 * L_LOADS>(); else if (ahead == 1) wait_vm<L_LOADS>(); else wait_vm<0>();
            __syncthreads();
            cur = (cur + 1) & (L_NST - 1);
        }
        if (a.dbg) { const long long now = (long long)__builtin_readcyclecounter(); t_loop += now - t_mark; t_mark = now; }
        // ---- epilogue: v = acc + bias + residual; LayerNorm over the 384 features of each token
        // lane (r, kh) of wave (wr, wc) holds, for token column ni: features wr*96 + mi*32 + 8g + 4kh + j.
        // scratch: 32 tokens x 128 B per wave, 16-byte chunks XOR-swizzled by the token row, in the slot consumed by the
        // tile's last step (cur was advanced: slot cur-1). Other waves' next DMA stage targets that slot too, hence the
        // barrier at the end of the epilogue.
        char *scr = smem + ((cur + L_NST - 1) & (L_NST - 1)) * L_SLOT + wave * 4096;
        const int rl_tok = lane >> 3, rl_f4 = (lane & 7) * 4;          // row layout: 8 lanes cover one token row
        float sum[2] = {0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int t0 = tile * L_BT + wc * 64 + ni * 32;
#pragma unroll
            for (int mi = 0; mi < 3; mi++) {
                const int n0 = wr * 96 + mi * 32;
                // residual block, read as full lines, parked in the scratch
                f32x16 &v = acc[mi][ni];
                if constexpr (RES16) {
                    // bf16 rows: 32 tokens x 64 B; 4 lanes cover one token row, chunk ^= (tok>>2)&3
                    const int tk = lane >> 2, ch = lane & 3;
                    uint4 rin[2];
#pragma unroll
                    for (int i = 0; i < 2; i++)
                        rin[i] = *(const uint4 *)(a.x16 + (int64_t)(t0 + tk + 16 * i) * L_H + n0 + ch * 8);
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int tok = tk + 16 * i;
                        *(uint4 *)(scr + tok * 64 + ((ch ^ ((tok >> 2) & 3)) << 4)) = rin[i];
                    }
                } else {
                    float4 rin[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        rin[i] = *(const float4 *)(a.x32 + (int64_t)(t0 + rl_tok + 8 * i) * L_H + n0 + rl_f4);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int tok = rl_tok + 8 * i;
                        *(float4 *)(scr + tok * 128 + (((lane & 7) ^ (tok & 7)) << 4)) = rin[i];
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float4 rr;
                    if constexpr (RES16) {
                        const uint2 h = *(const uint2 *)(scr + r * 64 + ((g ^ ((r >> 2) & 3)) << 4) + kh * 8);
                        rr = {bf16_to_f32((uint16_t)h.x), bf16_to_f32((uint16_t)(h.x >> 16)), bf16_to_f32((uint16_t)h.y),
                              bf16_to_f32((uint16_t)(h.y >> 16))};
                    } else rr = *(const float4 *)(scr + r * 128 + (((2 * g + kh) ^ (r & 7)) << 4));
                    const float4 bb = *(const float4 *)&s_bias[n0 + 8 * g + 4 * kh];
                    v[4 * g + 0] += bb.x + rr.x; v[4 * g + 1] += bb.y + rr.y;
                    v[4 * g + 2] += bb.z + rr.z; v[4 * g + 3] += bb.w + rr.w;
                    sum[ni] += (v[4 * g + 0] + v[4 * g + 1]) + (v[4 * g + 2] + v[4 * g + 3]);
                }
                __builtin_amdgcn_sched_barrier(0);   // one block at a time: the scratch is reused
            }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            sum[ni] += __shfl_xor(sum[ni], 32);
            if (kh == 0) s_part[wr * L_BT + wc * 64 + ni * 32 + r] = sum[ni];
        }
        __syncthreads();
        float mu[2], sq[2] = {0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int tl = wc * 64 + ni * 32 + r;
            mu[ni] = ((s_part[tl] + s_part[L_BT + tl]) + (s_part[2 * L_BT + tl] + s_part[3 * L_BT + tl])) * (1.0f / L_H);
#pragma unroll
            for (int mi = 0; mi < 3; mi++)
#pragma unroll
                for (int e = 0; e < 16; e++) { const float d = acc[mi][ni][e] - mu[ni]; sq[ni] += d * d; }
            sq[ni] += __shfl_xor(sq[ni], 32);
            if (kh == 0) s_part[4 * L_BT + wr * L_BT + tl] = sq[ni];
        }
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int tl = wc * 64 + ni * 32 + r;
            const float *q4 = s_part + 4 * L_BT;
            const float var = ((q4[tl] + q4[L_BT + tl]) + (q4[2 * L_BT + tl] + q4[3 * L_BT + tl])) * (1.0f / L_H);
            const float rstd = 1.0f / sqrtf(var + a.eps);
            const int t0 = tile * L_BT + wc * 64 + ni * 32;
#pragma unroll
            for (int mi = 0; mi < 3; mi++) {
                const int n0 = wr * 96 + mi * 32;
                const f32x16 &v = acc[mi][ni];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int n = n0 + 8 * g + 4 * kh;
                    const float4 gg = *(const float4 *)&s_gamma[n], bt = *(const float4 *)&s_beta[n];
                    const float4 y = {(v[4 * g + 0] - mu[ni]) * rstd * gg.x + bt.x, (v[4 * g + 1] - mu[ni]) * rstd * gg.y + bt.y,
                                      (v[4 * g + 2] - mu[ni]) * rstd * gg.z + bt.z, (v[4 * g + 3] - mu[ni]) * rstd * gg.w + bt.w};
                    if constexpr (RES16) {
                        const f32x4 yv = {y.x, y.y, y.z, y.w};
                        *(uint2 *)(scr + r * 64 + ((g ^ ((r >> 2) & 3)) << 4) + kh * 8) = __builtin_bit_cast(uint2, __builtin_convertvector(yv, bf16x4));
                    } else *(float4 *)(scr + r * 128 + (((2 * g + kh) ^ (r & 7)) << 4)) = y;
                }
                if constexpr (RES16) {
                    // back out as 64-byte row segments of the bf16 stream (in place)
                    const int tk = lane >> 2, ch = lane & 3;
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int tok = tk + 16 * i;
                        const uint4 yo = *(const uint4 *)(scr + tok * 64 + ((ch ^ ((tok >> 2) & 3)) << 4));
                        *(uint4 *)(a.x16 + (int64_t)(t0 + tok) * L_H + n0 + ch * 8) = yo;
                    }
                } else {
                    // back out as full lines: fp32 residual stream (in place) and its bf16 copy
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int tok = rl_tok + 8 * i;
                        const float4 yo = *(const float4 *)(scr + tok * 128 + (((lane & 7) ^ (tok & 7)) << 4));
                        const int64_t off = (int64_t)(t0 + tok) * L_H + n0 + rl_f4;
                        *(float4 *)(a.x32 + off) = yo;
                        const f32x4 yv = {yo.x, yo.y, yo.z, yo.w};
                        const uint2 hi = __builtin_bit_cast(uint2, __builtin_convertvector(yv, bf16x4));
                        if constexpr (X3) {
                            const f32x4 lv = yv - f32x4{__builtin_bit_cast(float, hi.x << 16), __builtin_bit_cast(float, hi.x & 0xffff0000u),
                                                        __builtin_bit_cast(float, hi.y << 16), __builtin_bit_cast(float, hi.y & 0xffff0000u)};
                            uint16_t *o2 = a.x16 + (int64_t)(t0 + tok) * 2 * L_H + n0 + rl_f4;
                            *(uint2 *)o2 = hi;
                            *(uint2 *)(o2 + L_H) = __builtin_bit_cast(uint2, __builtin_convertvector(lv, bf16x4));
                        } else *(uint2 *)(a.x16 + off) = hi;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();   // the scratch slot is the target of every wave's next DMA stage
        if (a.dbg) { const long long now = (long long)__builtin_readcyclecounter(); t_epi += now - t_mark; t_mark = now; }
        // s_part is rewritten by the next tile's epilogue only after its k-loop barriers
    }
    wait_vm<0>();
    if (a.dbg && lane == 0) { a.dbg[((size_t)blockIdx.x * L_NW + wave) * 2] = t_loop; a.dbg[((size_t)blockIdx.x * L_NW + wave) * 2 + 1] = t_epi; }
}

bool gemm_ln_supported(int H, int64_t T, int K) { return H == L_H && T % L_BT == 0 && K % 32 == 0 && K >= 32; }

// the split-bf16 form (X3 above): a.X [T][2 K'] and a.W [384][2 K'] rows [hi | lo], a.K = 3 K' (K' % 32 == 0), a.x32 the float32
// residual stream (in place), a.x16 [T][768] = [hi | lo] of the LayerNorm's output
int launch_gemm_ln_x3(const GemmLnArgs &a, hipStream_t st) {
    static std::atomic<bool> attr{false};
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm_ln<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS));
        attr = true;
    }
    if (a.T % L_BT || a.K % 96 || !a.x32 || !a.x16) AK_FAIL(-1, "gemm_ln (split bf16): T % 128 == 0, K' % 32 == 0, float32 residual stream");
    const int ntiles = a.T / L_BT;
    const int grid = ntiles < 256 ? ntiles : 256;
    GemmLnArgs b = a;
    b.dbg = nullptr;
    k_gemm_ln<false, true><<<grid, L_THREADS, L_LDS, st>>>(b);
    AK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_ln(const GemmLnArgs &a, hipStream_t st) {
    static std::atomic<bool> attr{false};      // (set twice by two first callers at worst: idempotent)
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm_ln<false>, hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm_ln<true>, hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS));
        attr = true;
    }
    const int ntiles = a.T / L_BT;
    const int grid = ntiles < 256 ? ntiles : 256;
    GemmLnArgs b = a;
    static long long *dbg = nullptr;
    static const bool gemmln_dbg = env_get("AK_GEMMLN_DBG") != nullptr;
    if (gemmln_dbg) {
        if (!dbg) AK_HIP(hipMalloc((void **)&dbg, 256 * L_NW * 2 * 8));
        b.dbg = dbg;
    } else b.dbg = nullptr;
    if (a.x32) k_gemm_ln<false><<<grid, L_THREADS, L_LDS, st>>>(b); else k_gemm_ln<true><<<grid, L_THREADS, L_LDS, st>>>(b);
    AK_HIP(hipGetLastError());
    if (b.dbg) {    // measurement mode: synchronous read-back and a one-line report per launch
        std::vector<long long> h((size_t)grid * L_NW * 2);
        AK_HIP(hipStreamSynchronize(st));
        AK_HIP(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double l = 0, e = 0;
        for (size_t i = 0; i < h.size(); i += 2) { l += h[i]; e += h[i + 1]; }
        fprintf(stderr, "k_gemm_ln K=%d: per wave K-loop %.0f kcyc, epilogue %.0f kcyc (%d tiles per workgroup)\n", a.K,
                l / (h.size() / 2) / 1e3, e / (h.size() / 2) / 1e3, (ntiles + grid - 1) / grid);
    }
    return 0;
}

}  // namespace ak
