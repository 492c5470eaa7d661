// gemm_skinny.hip -- Y = X W^T + bias for a FEW token rows (the query side of the read path: embed_query is one
// 32-token tile, postgres_vectorstore.py:245,390).
//
// The batch kernels (gemm.hip, gemm_ln.hip) give one workgroup a 128-token tile and walk K serially; with 32..1024
// tokens that is 1-8 workgroups on a 256-CU part and the walk's latency (12-48 K-steps) is the whole cost: 28 us per
// fused GEMM+LayerNorm call, 12 calls per forward pass, 337 of the 506 us a [1,32] forward spent on the GPU.
// Here the parallelism comes from the OUTPUT and from K instead:
//   grid  = (tokens/32) x (N/32) workgroups; NW waves split K between them (each wave owns K/NW);
//   wave  = one 32x32 accumulator; A (token rows) and B (weight rows) fragments are 16-byte global loads straight
//           into registers -- 8 bf16 of row (lane & 31) at k-offset 8*(lane >> 5) is exactly the MFMA operand layout,
//           and all loads of a wave are issued before its first MFMA (one memory round trip per wave);
//   reduce= the NW partial tiles meet in LDS; the epilogue adds the bias and writes fp32 rows (LayerNorm input) or
//           GELU -> bf16 rows (FFN up-projection) in 128 / 64-byte row segments.
// Weights are read once per 32-token row block (L2-resident: 0.3-2.4 MB per matrix).
#include "encoder_kernels.h"
#include "mfma_tile.h"

namespace ak {
using namespace mt;

// EPI 0: fp32 rows; 1: GELU -> bf16 rows; 2: the QKV split of gemm.hip's MODE 0 (Q pre-scaled and K as [T][H] bf16 rows, V
// transposed to [B][H][S]; q.T real tokens, rows beyond them have no V^T slot)
struct SkinnyQkv { uint16_t *q, *k, *vt; int H, S, T; float qscale; };
template <int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void k_gemm_skinny(const uint16_t *__restrict__ X, const uint16_t *__restrict__ W,
                                                         const float *__restrict__ bias, int N, int K,
                                                         float *__restrict__ out_f32, uint16_t *__restrict__ out_bf16, int ldo,
                                                         SkinnyQkv qkv) {
    __shared__ float part[NW][16][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int kw = K / NW;
    // this thread's bias: every output it finishes below is in column n0 + (lane & 31) (NW * 64 is a multiple of 64). Requested HERE,
    // with the operand loads -- behind the reduction it was one more dependent memory round trip of a kernel that is nothing but one
    // chain of them (round 6; the value and the arithmetic are unchanged)
    const float bv = bias[n0 + (lane & 31)];
    const uint16_t *xa = X + (int64_t)(m0 + r) * K + wave * kw + kh * 8;
    const uint16_t *wb = W + (int64_t)(n0 + r) * K + wave * kw + kh * 8;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    int k = 0;
    for (; k + 96 <= kw; k += 96) {             // 6 K-steps of 16: 12 loads in flight, then 6 MFMAs
        uint4 a[6], b[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { a[j] = *(const uint4 *)(xa + k + j * 16); b[j] = *(const uint4 *)(wb + k + j * 16); }
#pragma unroll
        for (int j = 0; j < 6; j++) acc = mfma_bf16(a[j], b[j], acc);
    }
    for (; k < kw; k += 16) acc = mfma_bf16(*(const uint4 *)(xa + k), *(const uint4 *)(wb + k), acc);
#pragma unroll
    for (int i = 0; i < 16; i++) part[wave][i][lane] = acc[i];
    __syncthreads();
    // accumulator element i of lane l is token row (i&3) + 8*(i>>2) + 4*(l>>5), output column l&31
    for (int idx = threadIdx.x; idx < 1024; idx += NW * 64) {
        const int i = idx >> 6, l = idx & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) v += part[w][i][l];
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
}

bool gemm_skinny_supported(int N, int K) {
    const int nw = K >= 1024 ? 8 : 4;
    return N % 32 == 0 && K % (nw * 16) == 0;
}

// X [rows][K] bf16 (rows a multiple of 32), W [N][K] bf16, bias [N]; out_f32 [rows][N], or GELU -> out_bf16 [rows][ldo]
int launch_gemm_skinny(const uint16_t *X, const uint16_t *W, const float *bias, int rows, int N, int K, float *out_f32,
                       uint16_t *out_bf16, int ldo, hipStream_t st) {
    if (rows % 32 || !gemm_skinny_supported(N, K)) AK_FAIL(-1, "launch_gemm_skinny: unsupported shape");
    const dim3 grid((unsigned)(rows / 32), (unsigned)(N / 32));
    const SkinnyQkv none{};
    if (K >= 1024) {
        if (out_bf16) k_gemm_skinny<8, 1><<<grid, 512, 0, st>>>(X, W, bias, N, K, nullptr, out_bf16, ldo, none);
        else k_gemm_skinny<8, 0><<<grid, 512, 0, st>>>(X, W, bias, N, K, out_f32, nullptr, 0, none);
    } else {
        if (out_bf16) k_gemm_skinny<4, 1><<<grid, 256, 0, st>>>(X, W, bias, N, K, nullptr, out_bf16, ldo, none);
        else k_gemm_skinny<4, 0><<<grid, 256, 0, st>>>(X, W, bias, N, K, out_f32, nullptr, 0, none);
    }
    AK_HIP(hipGetLastError());
    return 0;
}

// the QKV projection for a few token rows: X [rows][H] bf16, W [3H][H], outputs as gemm.hip's MODE 0
int launch_gemm_skinny_qkv(const uint16_t *X, const uint16_t *W, const float *bias, int rows, int H, int K, uint16_t *q,
                           uint16_t *k, uint16_t *vt, int S, int T, float qscale, hipStream_t st) {
    if (rows % 32 || !gemm_skinny_supported(3 * H, K)) AK_FAIL(-1, "launch_gemm_skinny_qkv: unsupported shape");
    const dim3 grid((unsigned)(rows / 32), (unsigned)(3 * H / 32));
    const SkinnyQkv a{q, k, vt, H, S, T, qscale};
    if (K >= 1024) k_gemm_skinny<8, 2><<<grid, 512, 0, st>>>(X, W, bias, 3 * H, K, nullptr, nullptr, 0, a);
    else k_gemm_skinny<4, 2><<<grid, 256, 0, st>>>(X, W, bias, 3 * H, K, nullptr, nullptr, 0, a);
    AK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ak
