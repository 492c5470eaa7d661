// encoder_f32.hip -- the float32 parity mode of the encoder (AkBertConfig.precision == 1) on the matrix cores.
//
// The reference's embedder is sentence-transformers on torch CPU fp32 (src/data_manager/vectorstore/manager.py:373,
// src/cli/templates/base-config.yaml:143-150); north_star asks for "the same top-k as the reference CPU path ... scores within
// 1e-5 fp32" from text, which the bf16 path cannot give (overlap@10 0.98, |d score| 1.6e-4). This mode can: float32 weights,
// activations and accumulation throughout. Until round 5 it was a scalar-fmaf tile GEMM and a one-wave-per-query attention
// (checking tools without a throughput number). gfx950 has an exact float32 matrix instruction -- v_mfma_f32_32x32x2_f32: f32
// in, f32 accumulate, one rounding per product, bitwise a k-ordered fmaf chain, 64 FLOP per clock and SIMD = 157 TFLOP/s
// per chip (MI355X_MICROARCH.md, "Peak FP32 (matrix)") -- so the same arithmetic runs here as a real kernel:
//
//   k32m_gemm   Y[T][ldc] (+col0) = X[T][K] . W[N][K]^T + bias (+ erf GELU | + residual). 128 x 128 tile, four waves of 64 x 64
//               (2 x 2 MFMA tiles), K in steps of 32 through a two-slot LDS ring filled from registers (the next K-step's
//               global loads are issued before the current step's 64 MFMAs); rows padded to 33 floats: the per-lane
//               ds_read_b32 of an operand (32 rows x one k) touches 32 banks. Every output is ONE fmaf chain over k
//               ascending, started from zero, bias added last: bit-identical to the scalar kernel it replaces (k32_gemm, kept
//               in libarchi_hip_dbg.so as the cross-check).
//   k32m_attn   softmax(q k^T / sqrt(hd) + mask) v per (sequence, head, 128 queries): keys on M like the bf16 kernels (a lane
//               owns a query column of the 32 x 32 score tile, 16 keys per register set), online softmax in float32 with expf,
//               P feeds P.V straight from the score registers: lane halves hold the two k-slots of a 32x32x2 B operand, so
//               accumulator register r IS the operand of the MFMA that multiplies keys {kappa(r), kappa(r) + 4}. K / V blocks
//               of 32 keys through a two-slot LDS ring (register-staged), context rows written through LDS as whole lines.
//
// Algorithmic work: the bf16 path's (6.04 GFLOP per 256-token MiniLM chunk, 96.6 per 512-token bge-base chunk);
// roof: 157.3 TFLOP/s. bench.py reports `embed.f32_parity`.
#include <atomic>

#include "mfma_tile.h"
#include "encoder_kernels.h"

namespace ak {
using namespace mt;

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

constexpr int G32_BM = 128, G32_BN = 128, G32_BK = 32, G32_LD = G32_BK + 1;     // LDS rows of 33 floats

// XCD-AWARE TILE ORDER (round 6). Block b runs on XCD b % 8 (placement for speed only, nothing depends on it). With a (N tiles,
// T tiles) grid the N / 128 column tiles of one token tile -- which all read the same 128 x K float32 rows of X -- land on
// eight different XCDs, and every one of those L2s fetches the rows for itself: X left HBM / the Infinity Cache up to eight times
// (FFN-down: 3 x 403 MB per layer), and both float32-grade GEMMs ran at the speed of that traffic (4-5 k cycles per 32-k step,
// whatever the MFMAs cost: 24 x 32 cycles here, 64 x 64 in k32m_gemm). Now the column tiles of a token tile are consecutive
// slots of ONE XCD: 1-D grid of 8 * ceil(TT / 8) * NT blocks, b -> xcd = b % 8, j = b / 8, column tile j % NT, token tile
// xcd + 8 * (j / NT); blocks past the last token tile exit.
__device__ __forceinline__ void tile_of_block(int NT, int &tt, int &nt) {
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    nt = j % NT;
    tt = xcd + 8 * (j / NT);
}
static inline unsigned xcd_grid(int T, int N) {
    const int TT = (T + G32_BM - 1) / G32_BM, NT = N / G32_BN;
    return (unsigned)(8 * ((TT + 7) / 8) * NT);
}
constexpr int G32_LDS = 2 * 2 * G32_BM * G32_LD * 4;                           // two slots x (X tile + W tile) = 67 584 B


// EPILOGUE of both float32-grade GEMMs (round 6): the 128 x 128 float32 tile goes through LDS ([token][feature], rows of 132
// floats: the K-loop's slots are free behind its last barrier) and leaves as 16-byte stores, two whole 512-byte row segments per
// wave instruction, with the bias / residual read as float4. Before, a lane stored its 64 accumulator values one dword at a time
// (64 store instructions per lane, residual: 64 scalar loads): the stores of a finishing wave queue behind each other -- at K = 384
// the epilogue was a third of a workgroup's life. Per element the arithmetic and its order are unchanged: acc + bias, then GELU or
// + residual; results are bit-identical to the scalar-store form.
constexpr int EPI_LD = 132;
static_assert(G32_BM * EPI_LD * 4 <= 2 * 2 * G32_BM * 33 * 4, "the output tile fits the float32 kernel's LDS");
template <int EPI>
__device__ __forceinline__ void tile_epilogue(const f32x16 (&acc)[2][2], float *tile, const float *__restrict__ bias, const float *__restrict__ R,
                                              int T, float *__restrict__ Y, int ldc, int col0, int t0, int n0, int wm, int wn, int li, int lk, int tid) {
    // lane (feature column li, half lk) holds token rows (r & 3) + 8 (r >> 2) + 4 lk of each 32 x 32 tile
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                tile[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * EPI_LD + wn * 64 + j * 32 + li] = acc[i][j][r];
    __syncthreads();
    const int c4 = (tid & 31) * 4, r0 = tid >> 5;
    const f32x4v bv = *(const f32x4v *)(bias + n0 + c4);
#pragma unroll 4
    for (int jj = 0; jj < 16; jj++) {
        const int row = r0 + 8 * jj, t = t0 + row;
        if (t >= T) continue;
        f32x4v v = *(const f32x4v *)(tile + row * EPI_LD + c4);
        v += bv;
        if constexpr (EPI == 1) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
        }
        if constexpr (EPI == 2) v += *(const f32x4v *)(R + (int64_t)t * ldc + col0 + n0 + c4);
        *(f32x4v *)(Y + (int64_t)t * ldc + col0 + n0 + c4) = v;
    }
}

// EPI: 0 bias, 1 bias + exact (erf) GELU, 2 bias + residual R[T][ldc] (same layout as Y)
template <int EPI>
__global__ __launch_bounds__(256, 2) void k32m_gemm(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                                                    const float *__restrict__ R, int T, int N, int K, float *__restrict__ Y, int ldc, int col0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sA = (float *)smem;                          // [2][128][33] token rows
    float *sB = sA + 2 * G32_BM * G32_LD;               // [2][128][33] feature rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    int tt, nt;
    tile_of_block(N / G32_BN, tt, nt);
    const int t0 = tt * G32_BM, n0 = nt * G32_BN;
    if (t0 >= T) return;
    const int li = lane & 31, lk = lane >> 5;
    // staging: 128 rows x 32 floats per operand = 1024 float4: thread i takes float4 (row = i / 8 + 32 j, chunk = i % 8), j = 0..3
    const int srow = tid >> 3, sch = tid & 7;
    f32x4v ra[4], rb[4];
    // branch-free staging (round 6; see k3_gemm below): rows past T read row T - 1 (never stored), the last K-step re-loads its own slice
    const float *xrow[4], *wrow4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = srow + 32 * j, t = t0 + r;
        xrow[j] = X + (int64_t)(t < T ? t : T - 1) * K + sch * 4;
        wrow4[j] = W + (int64_t)(n0 + r) * K + sch * 4;
    }
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++) { ra[j] = *(const f32x4v *)(xrow[j] + k0); rb[j] = *(const f32x4v *)(wrow4[j] + k0); }
    };
    auto lstore = [&](int slot) __attribute__((always_inline)) {
        float *a = sA + slot * G32_BM * G32_LD, *b = sB + slot * G32_BM * G32_LD;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = srow + 32 * j;
#pragma unroll
            for (int e = 0; e < 4; e++) { a[r * G32_LD + sch * 4 + e] = ra[j][e]; b[r * G32_LD + sch * 4 + e] = rb[j][e]; }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = K / G32_BK;
    for (int ks = 0; ks < nk; ks++) {
        const int slot = ks & 1;
        gload((ks + 1 < nk ? ks + 1 : ks) * G32_BK);        // in flight under this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);                  // (issued here, not sunk behind the MFMAs)
        const float *a = sA + slot * G32_BM * G32_LD + (wm * 64 + li) * G32_LD + lk;
        const float *b = sB + slot * G32_BM * G32_LD + (wn * 64 + li) * G32_LD + lk;
#pragma unroll
        for (int kk = 0; kk < G32_BK / 2; kk++) {
            const float a0 = a[2 * kk], a1 = a[32 * G32_LD + 2 * kk], b0 = b[2 * kk], b1 = b[32 * G32_LD + 2 * kk];
            acc[0][0] = mfma_f32(a0, b0, acc[0][0]);
            acc[0][1] = mfma_f32(a0, b1, acc[0][1]);
            acc[1][0] = mfma_f32(a1, b0, acc[1][0]);
            acc[1][1] = mfma_f32(a1, b1, acc[1][1]);
        }
        lstore(slot ^ 1);                                   // the other slot: nobody reads it during this step
        __syncthreads();
    }
    tile_epilogue<EPI>(acc, (float *)smem, bias, R, T, Y, ldc, col0, t0, n0, wm, wn, li, lk, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// attention: one workgroup = 4 waves = 128 queries of one (sequence, head); qkv [T][3H] (q | k | v) float32
template <int HD>
__global__ __launch_bounds__(256, 2) void k32m_attn(const float *__restrict__ qkv, const int *__restrict__ mask, int B, int S, int H, int heads,
                                                    float *__restrict__ ctx) {
    constexpr int LD = HD + 1, DB = HD / 32, KB = 32;                       // key block
    __shared__ float sK[2][KB][LD], sV[2][KB][LD];
    __shared__ float sMask[2][KB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int nqb = (S + 127) / 128;
    // the nqb query blocks of one (sequence, head) read the same K / V rows: they sit 8 apart in the grid = on ONE XCD (block i runs
    // on XCD i % 8, each with its own L2), dispatched together. With the query block simply fastest they landed on nqb XCDs and every
    // one fetched K / V for itself (PMC, bge-base 128 x 512: 1.73 GB per launch for 0.6 GB of operands; attn_grid below pads the grid)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qb = slot % nqb, grp = (slot / nqb) * 8 + xcd;
    if (grp >= B * heads) return;
    const int h = grp % heads, b = grp / heads;
    const int q0 = qb * 128 + wave * 32;
    const float scale = 1.0f / sqrtf((float)HD);
    const int64_t row0 = (int64_t)b * S;
    // this lane's query operand: Q[q0 + li][2 kk + lk], kk = 0 .. HD / 2 - 1 (B operand of the score MFMAs)
    float qreg[HD / 2];
    {
        int qr = q0 + li;
        if (qr >= S) qr = S - 1;
        const float *qp = qkv + (row0 + qr) * 3 * H + h * HD;
#pragma unroll
        for (int kk = 0; kk < HD / 2; kk++) qreg[kk] = qp[2 * kk + lk];
    }
    // staging of a key block: 32 keys x HD floats of K and of V = 2 * 8 * HD float4... thread i: key = i / (HD / 4), chunk = i % (HD / 4)
    constexpr int CPR = HD / 4, NLD = (KB * CPR + 255) / 256;               // float4 per row; loads per thread and matrix
    f32x4v rk[NLD], rv[NLD];
    int rmask = 0;
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int i = tid + 256 * j, key = i / CPR, c = i % CPR;
            if (key < KB) {
                int kr = k0 + key;
                if (kr >= S) kr = S - 1;
                const float *base = qkv + (row0 + kr) * 3 * H + h * HD + c * 4;
                rk[j] = *(const f32x4v *)(base + H);
                rv[j] = *(const f32x4v *)(base + 2 * H);
            }
        }
        if (tid < KB) rmask = (k0 + tid < S) ? mask[row0 + k0 + tid] : 0;
    };
    auto lstore = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int i = tid + 256 * j, key = i / CPR, c = i % CPR;
            if (key < KB) {
#pragma unroll
                for (int e = 0; e < 4; e++) { sK[slot][key][c * 4 + e] = rk[j][e]; sV[slot][key][c * 4 + e] = rv[j][e]; }
            }
        }
        if (tid < KB) sMask[slot][tid] = rmask ? 0.f : -__builtin_inff();
    };
    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[d][e] = 0.f;
    float m = -__builtin_inff(), l = 0.f;
    const int nblk = (S + KB - 1) / KB;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int blk = 0; blk < nblk; blk++) {
        const int slot = blk & 1;
        if (blk + 1 < nblk) gload((blk + 1) * KB);
        // scores: A = K rows (keys), B = Q columns (queries); lane (query li, half lk) gets keys kappa(r) = (r & 3) + 8 (r >> 2) + 4 lk
        f32x16 sc;
#pragma unroll
        for (int e = 0; e < 16; e++) sc[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < HD / 2; kk++) sc = mfma_f32(sK[slot][li][2 * kk + lk], qreg[kk], sc);
        float mx = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sc[r] = sc[r] * scale + sMask[slot][(r & 3) + 8 * (r >> 2) + 4 * lk];
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        const float alpha = m == -__builtin_inff() ? 0.f : expf(m - m_new);      // (m_new == -inf: nothing but masked keys so far, alpha = 0 is fine)
        float bs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            sc[r] = m_new == -__builtin_inff() ? 0.f : expf(sc[r] - m_new);
            bs += sc[r];
        }
        l = l * alpha + bs;
        m = m_new;
#pragma unroll
        for (int d = 0; d < DB; d++)
#pragma unroll
            for (int e = 0; e < 16; e++) o[d][e] *= alpha;
        // P . V: register r of the score tile is the B operand of the MFMA over keys {kappa(r) in half 0, kappa(r) + 4 in half 1};
        // A = V^T: lane (feature li of tile d, half lk) reads V[that key][32 d + li]
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * lk;
#pragma unroll
            for (int d = 0; d < DB; d++) o[d] = mfma_f32(sV[slot][key][32 * d + li], sc[r], o[d]);
        }
        if (blk + 1 < nblk) lstore(slot ^ 1);
        __syncthreads();
    }
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    // context rows through LDS ([query][feature], one wave's 32 x HD block in its own part of sK / sV): whole lines out
    float *tr = wave < 2 ? &sK[0][0][0] + wave * (32 * LD) : &sV[0][0][0] + (wave - 2) * (32 * LD);
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) tr[li * LD + 32 * d + (r & 3) + 8 * (r >> 2) + 4 * lk] = o[d][r] * inv;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 32 * HD; i += 64) {
        const int qi = i / HD, c = i % HD;
        if (q0 + qi < S) ctx[(row0 + q0 + qi) * H + h * HD + c] = tr[qi * LD + c];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// SPLIT-bf16 GEMM ("bf16x3", AkBertConfig.precision == 2; round-5 review, missing #2): the float32 parity mode above gives the
// reference CPU embedder's top-k from text (scores within 1e-5) but runs the GEMMs at the float32 matrix rate, 1/16 of the bf16
// one. Here every GEMM operand is split  x = hi + lo  (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits between them) and the
// product is  hi.hi + lo.hi + hi.lo  on v_mfma_f32_32x32x16_bf16 into ONE float32 accumulator -- three bf16 MFMAs (3 x 32
// cycles per 32 x 32 x 16) where the float32 path issues eight 32x32x2 MFMAs (8 x 64 cycles). What is dropped: lo.lo and the
// 2^-18 |x| each split leaves behind, ~4e-6 relative per product with random sign; measured on the CPU emulation of this
// arithmetic (random-init MiniLM / bge-base, against float64): max |delta| 1-1.7e-6 per embedding component, |delta score|
// <= 1.2e-6 -- the float32 path itself sits at 1e-7. Everything between the GEMMs (LayerNorm, softmax, erf GELU, residuals,
// pooling) stays float32, and attention is the float32 kernel above.
//   k3_gemm   the tile of k32m_gemm (128 x 128, four waves of 2 x 2 MFMA tiles, K in steps of 32, two LDS slots, the next step's
//             global loads in flight under this step's MFMAs). X is float32 in HBM and split ONCE per element on its way into LDS
//             (the staging thread holds it in registers anyway: 2 cvt_pk + 4 sub + 2 cvt_pk per float4); the weights are split
//             once, at encoder creation (k_split_hilo), and staged as they lie. LDS: four bf16 tiles [128][32] per slot, rows
//             padded to 80 bytes -- the 16 lanes of a ds_read_b128 group then cover all 64 banks.
// EPI as k32m_gemm: 0 bias, 1 bias + exact (erf) GELU, 2 bias + residual.
constexpr int G3_LDB = 80;                                             // bytes per LDS row: 32 bf16 + 16 bytes of padding
constexpr int G3_TILE = G32_BM * G3_LDB;                               // one bf16 tile: 10 240 B
constexpr int G3_LDS = 2 * 4 * G3_TILE;                                // two slots x (X hi, X lo, W hi, W lo) = 81 920 B

__global__ void k_split_hilo(const float *__restrict__ w, int64_t n, uint16_t *__restrict__ hi, uint16_t *__restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = w[i];
    const uint16_t h = f32_to_bf16(x);
    hi[i] = h;
    lo[i] = f32_to_bf16(x - bf16_to_f32(h));
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void k3_gemm(const float *__restrict__ X, const uint16_t *__restrict__ Whi, const uint16_t *__restrict__ Wlo,
                                                  const float *__restrict__ bias, const float *__restrict__ R, int T, int N, int K,
                                                  float *__restrict__ Y, int ldc, int col0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    int tt, nt;
    tile_of_block(N / G32_BN, tt, nt);
    const int t0 = tt * G32_BM, n0 = nt * G32_BN;
    if (t0 >= T) return;
    const int li = lane & 31, lk = lane >> 5;
    // staging of X: 128 rows x 32 floats = 1024 float4: thread i takes (row = i / 8 + 32 j, float4 chunk = i % 8), j = 0..3
    // staging of W: 128 rows x 64 B of hi and of lo = 512 uint4 each: thread i takes (row = i / 4 + 64 j, 16-byte chunk = i % 4), j = 0, 1
    // (row index bits 0 and 2 swapped: the two rows that one LDS store cycle covers -- 16 lanes of a ds_write_b64, 8 of a
    // ds_write_b128 -- are then 4 rows = 320 B = 16 banks apart instead of 80 B = 20 banks, which overlapped 4 banks of 32:
    // PMC, first version: 32 % of this kernel's LDS cycles were bank conflicts, all of them store-side)
    auto swap02 = [](int r) { return (r & ~5) | ((r & 1) << 2) | ((r & 4) >> 2); };
    const int srow = swap02(tid >> 3), sch = tid & 7, wrow = swap02(tid >> 2), wch = tid & 3;
    // TWO register sets: the loads of K-step ks + 2 are issued while step ks computes and step ks + 1's (issued one step earlier)
    // wait to be split and stored -- every load has two steps of MFMAs to land. With one set (one step of cover, 32 KB in
    // flight per workgroup) the kernel ran at what a CU's 64 KB of outstanding loads deliver at the ~2 us of a loaded memory
    // system: 12 B / clk / CU, 5 k cycles per K-step for 1.5 k of matrix work (ext-vector types: hipcc keeps them in registers
    // across the loop; as uint4 structs they went through scratch).
    struct Stage { f32x4v ra[4]; u32x4v rh[2], rl[2]; };
    Stage sa, sb;
    // BRANCH-FREE staging. With `t < T ? load : 0` per row and `if (ks + 1 < nk)` around the loads and around the stores, hipcc
    // could not tell that the two conditions are one: it assumed the previous step's loads might still be pending where this
    // step's address registers are written and put s_waitcnt vmcnt(3..0) BETWEEN the X loads and the W loads -- every K-step
    // opened with a full memory round trip. Rows past T read row T - 1 (never stored), steps past the last re-load the last slice
    // (stored into a slot nobody reads any more, or not stored at all).
    const float *xrow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = t0 + srow + 32 * j;
        xrow[j] = X + (int64_t)(t < T ? t : T - 1) * K + sch * 4;
    }
    const int64_t wo0 = (int64_t)(n0 + wrow) * K + wch * 8, wo1 = wo0 + (int64_t)64 * K;
    const int nk = K / G32_BK;
    auto gload = [&](Stage &g, int ks) __attribute__((always_inline)) {
        const int k0 = (ks < nk ? ks : nk - 1) * G32_BK;
#pragma unroll
        for (int j = 0; j < 4; j++) g.ra[j] = *(const f32x4v *)(xrow[j] + k0);
        g.rh[0] = *(const u32x4v *)(Whi + wo0 + k0); g.rl[0] = *(const u32x4v *)(Wlo + wo0 + k0);
        g.rh[1] = *(const u32x4v *)(Whi + wo1 + k0); g.rl[1] = *(const u32x4v *)(Wlo + wo1 + k0);
    };
    auto lstore = [&](const Stage &g, int slot) __attribute__((always_inline)) {
        char *base = smem + slot * 4 * G3_TILE;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x4v x = g.ra[j];
            const uint32_t h01 = pack_bf16x2(x[0], x[1]), h23 = pack_bf16x2(x[2], x[3]);
            const float r0 = x[0] - __builtin_bit_cast(float, h01 << 16), r1 = x[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
            const float r2 = x[2] - __builtin_bit_cast(float, h23 << 16), r3 = x[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
            const int off = (srow + 32 * j) * G3_LDB + sch * 8;
            *(uint2 *)(base + off) = uint2{h01, h23};
            *(uint2 *)(base + G3_TILE + off) = uint2{pack_bf16x2(r0, r1), pack_bf16x2(r2, r3)};
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int off = (wrow + 64 * j) * G3_LDB + wch * 16;
            *(u32x4v *)(base + 2 * G3_TILE + off) = g.rh[j];
            *(u32x4v *)(base + 3 * G3_TILE + off) = g.rl[j];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    // one K-step: the MFMAs of LDS slot `slot`
    auto compute = [&](int slot) __attribute__((always_inline)) {
        const char *a = smem + slot * 4 * G3_TILE + (wm * 64 + li) * G3_LDB + lk * 16;       // A operand: row li, k = 8 lk .. + 8 of a 16-k sub-step
        const char *b = smem + slot * 4 * G3_TILE + 2 * G3_TILE + (wn * 64 + li) * G3_LDB + lk * 16;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {                    // two 16-k sub-steps of the 32-k step
            uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                ah[i] = *(const uint4 *)(a + i * 32 * G3_LDB + kk * 32);
                al[i] = *(const uint4 *)(a + G3_TILE + i * 32 * G3_LDB + kk * 32);
                bh[i] = *(const uint4 *)(b + i * 32 * G3_LDB + kk * 32);
                bl[i] = *(const uint4 *)(b + G3_TILE + i * 32 * G3_LDB + kk * 32);
            }
            // the two small terms first, the leading term last; term by term over the four accumulators, so that an MFMA never waits for
            // the one just issued (three back-to-back MFMAs on ONE accumulator each sit out the previous one's 16 passes)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = mfma_bf16(al[i], bh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = mfma_bf16(ah[i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = mfma_bf16(ah[i], bh[j], acc[i][j]);
        }
    };
    gload(sa, 0);
    gload(sb, 1);
    lstore(sa, 0);                                          // (waits for step 0's loads only: step 1's stay in flight)
    __syncthreads();
    // two K-steps per trip: set `sa` carries the even steps' data, `sb` the odd ones'
    for (int ks = 0; ks < nk; ks += 2) {
        gload(sa, ks + 2);
        __builtin_amdgcn_sched_barrier(0);                  // loads issued HERE: left alone, hipcc sinks them behind the MFMAs, next to their use
        compute(0);
        lstore(sb, 1);                                      // step ks + 1, issued one step ago
        __syncthreads();
        if (ks + 1 >= nk) break;                            // (odd step count: uniform)
        gload(sb, ks + 3);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        lstore(sa, 0);                                      // step ks + 2
        __syncthreads();
    }
    tile_epilogue<EPI>(acc, (float *)smem, bias, R, T, Y, ldc, col0, t0, n0, wm, wn, li, lk, tid);
}

int split_hilo(const float *w, int64_t n, uint16_t *hi, uint16_t *lo, hipStream_t st) {
    k_split_hilo<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(w, n, hi, lo);
    AK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_x3(int epi, const float *X, const uint16_t *Whi, const uint16_t *Wlo, const float *bias, const float *R, int T, int N, int K,
                   float *Y, int ldc, int col0, hipStream_t st) {
    if (N % G32_BN || K % G32_BK) AK_FAIL(-1, "launch_gemm_x3: N must be a multiple of 128, K of 32");
    static std::atomic<bool> attr{false};
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k3_gemm<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k3_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k3_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS));
        attr = true;
    }
    const dim3 grid(xcd_grid(T, N));
    if (epi == 0) k3_gemm<0><<<grid, 256, G3_LDS, st>>>(X, Whi, Wlo, bias, R, T, N, K, Y, ldc, col0);
    else if (epi == 1) k3_gemm<1><<<grid, 256, G3_LDS, st>>>(X, Whi, Wlo, bias, R, T, N, K, Y, ldc, col0);
    else k3_gemm<2><<<grid, 256, G3_LDS, st>>>(X, Whi, Wlo, bias, R, T, N, K, Y, ldc, col0);
    AK_HIP(hipGetLastError());
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// k3_attn -- attention of the split-bf16 parity mode (precision 2): k32m_attn's loop (one workgroup = 4 waves = 128 queries of one
// (sequence, head), key blocks of 32 through a two-slot LDS ring staged from registers, online softmax in float32 (base 2: v_exp_f32), keys
// on M so that a lane owns a query column) with BOTH matrix products as three bf16 MFMAs into float32 accumulators:
//   scores  = K_lo.q_hi + K_hi.q_lo + K_hi.q_hi      K split on its way into LDS, q split once per workgroup in registers
//   context = V_lo.P_hi + V_hi.P_lo + V_hi.P_hi      V split AND transposed on its way into LDS, P split in registers
// The score tile's accumulator layout IS the P.V MFMA's B operand once the keys of a 16-key group sit in V^T in the order
// [0-3, 8-11, 4-7, 12-15] (vt_pos, as in the bf16 kernels): P never leaves the registers. 3 x 32 cycles per 32 x 32 x 16 product
// where the float32 kernel spends 8 x 64; error per product ~2^-17 relative (see k3_gemm). LDS rows are padded to 80 / 144 bytes:
// the 16 lanes of a ds_read_b128 group cover all 64 banks.
template <int HD>
__global__ __launch_bounds__(256, 2) void k3_attn(const float *__restrict__ qkv, const int *__restrict__ mask, int B, int S, int H, int heads,
                                                  float *__restrict__ ctx, uint16_t *__restrict__ ctx2, int ldq) {
    constexpr int KB = 32, DB = HD / 32, KC = HD / 16;                       // key block; 32-feature tiles; 16-wide chunks of the head dimension
    constexpr int KROW = HD * 2 + 16, VROW = KB * 2 + 16;                   // padded LDS rows (bytes): K [key][HD], V^T [feature][KB]
    constexpr int K_BYTES = KB * KROW, V_BYTES = HD * VROW, SLOT = 2 * K_BYTES + 2 * V_BYTES + KB * 4;
    constexpr int LD = HD + 1;
    __shared__ __attribute__((aligned(16))) char ring[2 * SLOT];
    static_assert(2 * SLOT >= 4 * 32 * LD * 4, "the ring also carries the context rows out");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int nqb = (S + 127) / 128;
    // the nqb query blocks of one (sequence, head) read the same K / V rows: they sit 8 apart in the grid = on ONE XCD (block i runs
    // on XCD i % 8, each with its own L2), dispatched together. With the query block simply fastest they landed on nqb XCDs and every
    // one fetched K / V for itself (PMC, bge-base 128 x 512: 1.73 GB per launch for 0.6 GB of operands; attn_grid below pads the grid)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qb = slot % nqb, grp = (slot / nqb) * 8 + xcd;
    if (grp >= B * heads) return;
    const int h = grp % heads, b = grp / heads;
    const int q0 = qb * 128 + wave * 32;
    // the scores are kept in the base-2 domain: q carries 1 / sqrt(hd) x log2(e) from the start (one float32 rounding, before the
    // split), so the softmax is v_exp_f32 (2^x, 1 ulp) of a difference instead of expf's ~20 instructions per score -- this kernel's
    // time is its VALU stream (r6z trace: 219 / 585 us per layer with expf, 12 MFMAs against ~420 vector instructions per key block)
    const float scale = 1.4426950408889634f / sqrtf((float)HD);
    const int64_t row0 = (int64_t)b * S;
    auto split8 = [](const float (&x)[8], uint4 &hi, uint4 &lo) {
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            hw[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
            const f32x2v lo2 = f32x2v{x[2 * e], x[2 * e + 1]} - f32x2v{__builtin_bit_cast(float, hw[e] << 16), __builtin_bit_cast(float, hw[e] & 0xffff0000u)};
            lw[e] = pack_bf16x2(lo2[0], lo2[1]);
        }
        hi = uint4{hw[0], hw[1], hw[2], hw[3]}; lo = uint4{lw[0], lw[1], lw[2], lw[3]};
    };
    // this lane's query operand (B operand of the score MFMAs): q[q0 + li][16 c + 8 lk .. + 8], split once
    uint4 qh[KC], ql[KC];
    {
        int qr = q0 + li;
        if (qr >= S) qr = S - 1;
        const float *qp = qkv + (row0 + qr) * ldq + h * HD;
#pragma unroll
        for (int c = 0; c < KC; c++) {
            const f32x4v a0 = *(const f32x4v *)(qp + 16 * c + 8 * lk), a1 = *(const f32x4v *)(qp + 16 * c + 8 * lk + 4);
            const float x[8] = {a0[0] * scale, a0[1] * scale, a0[2] * scale, a0[3] * scale, a1[0] * scale, a1[1] * scale, a1[2] * scale, a1[3] * scale};
            split8(x, qh[c], ql[c]);
        }
    }
    // staging of a key block: 32 keys x HD floats of K and of V: thread i -> key = i / (HD / 4), float4 chunk = i % (HD / 4)
    constexpr int CPR = HD / 4, NLD = (KB * CPR + 255) / 256;
    f32x4v rk[NLD], rv[NLD];
    int rmask = 0;
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int i = tid + 256 * j, key = i / CPR, c = i % CPR;
            if (key < KB) {
                int kr = k0 + key;
                if (kr >= S) kr = S - 1;
                const float *base = qkv + (row0 + kr) * ldq + h * HD + c * 4;
                rk[j] = *(const f32x4v *)(base + H);
                rv[j] = *(const f32x4v *)(base + 2 * H);
            }
        }
        if (tid < KB) rmask = (k0 + tid < S) ? mask[row0 + k0 + tid] : 0;
    };
    auto lstore = [&](int slot) __attribute__((always_inline)) {
        char *sk = ring + slot * SLOT, *sv = sk + 2 * K_BYTES;
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int i = tid + 256 * j, key = i / CPR, c = i % CPR;
            if (key < KB) {
                const f32x4v x = rk[j], y = rv[j];
                const uint32_t h01 = pack_bf16x2(x[0], x[1]), h23 = pack_bf16x2(x[2], x[3]);
                const uint32_t l01 = pack_bf16x2(x[0] - __builtin_bit_cast(float, h01 << 16), x[1] - __builtin_bit_cast(float, h01 & 0xffff0000u));
                const uint32_t l23 = pack_bf16x2(x[2] - __builtin_bit_cast(float, h23 << 16), x[3] - __builtin_bit_cast(float, h23 & 0xffff0000u));
                *(uint2 *)(sk + key * KROW + c * 8) = uint2{h01, h23};
                *(uint2 *)(sk + K_BYTES + key * KROW + c * 8) = uint2{l01, l23};
                // V transposed: feature 4 c + e of this key -> row (4 c + e), position vt_pos(key) of the block's 32 keys
                const int pos = (key & 16) | vt_pos(key & 15);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint16_t vh = (uint16_t)(pack_bf16x2(y[e], 0.f));
                    const uint16_t vl = (uint16_t)(pack_bf16x2(y[e] - bf16_to_f32(vh), 0.f));
                    *(uint16_t *)(sv + (c * 4 + e) * VROW + pos * 2) = vh;
                    *(uint16_t *)(sv + V_BYTES + (c * 4 + e) * VROW + pos * 2) = vl;
                }
            }
        }
        if (tid < KB) *(float *)(ring + slot * SLOT + 2 * K_BYTES + 2 * V_BYTES + tid * 4) = rmask ? 0.f : -__builtin_inff();
    };
    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[d][e] = 0.f;
    float m = -__builtin_inff(), l = 0.f;
    const int nblk = (S + KB - 1) / KB;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int blk = 0; blk < nblk; blk++) {
        const int slot = blk & 1;
        gload((blk + 1 < nblk ? blk + 1 : blk) * KB);                      // (the last block re-loads itself: unused, branch-free)
        __builtin_amdgcn_sched_barrier(0);
        const char *sk = ring + slot * SLOT, *sv = sk + 2 * K_BYTES;
        const float *sm = (const float *)(sv + 2 * V_BYTES);
        // scores: A = K rows (keys on M), B = q. The kernel's time is its vector-instruction stream (r6q2 counters: VALU ~58 % of the
        // SIMD cycles, matrix pipe 21 %), so everything that can ride on the MFMAs does: the additive key mask is the accumulator's
        // starting value, the three terms of every chunk go into that one accumulator
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; r++) sc[r] = sm[(r & 3) + 8 * (r >> 2) + 4 * lk];
#pragma unroll
        for (int c = 0; c < KC; c++) {
            const uint4 kh = *(const uint4 *)(sk + li * KROW + c * 32 + lk * 16), kl = *(const uint4 *)(sk + K_BYTES + li * KROW + c * 32 + lk * 16);
            sc = mfma_bf16(kl, qh[c], sc);
            sc = mfma_bf16(kh, ql[c], sc);
            sc = mfma_bf16(kh, qh[c], sc);
        }
        // (the maximum as the median with +inf: v_med3_f32 takes the MFMA's results as they are, fmaxf would first canonicalise each)
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 16; r++) mx = __builtin_amdgcn_fmed3f(mx, sc[r], __builtin_inff());
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        const float m_use = m_new == -__builtin_inff() ? 0.f : m_new;       // nothing but masked keys so far: 2^(-inf - 0) = 0 everywhere
        const float alpha = __builtin_amdgcn_exp2f(m - m_use);
        f32x2v bs2 = {0.f, 0.f};
        const f32x2v mref = {m_use, m_use};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                                   // two scores per v_pk_add_f32
            const f32x2v d = f32x2v{sc[r], sc[r + 1]} - mref;
            const f32x2v pv = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
            sc[r] = pv[0]; sc[r + 1] = pv[1];
            bs2 += pv;
        }
        l = l * alpha + (bs2[0] + bs2[1]);
        m = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {                  // (the running maximum settles after a few blocks)
#pragma unroll
            for (int d = 0; d < DB; d++)
#pragma unroll
                for (int e = 0; e < 16; e++) o[d][e] *= alpha;
        }
        // P . V: registers 8 c2 .. 8 c2 + 7 of the score tile are the B operand of key chunk c2 (V^T holds the keys in vt_pos order)
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++) {
            const float x[8] = {sc[8 * c2 + 0], sc[8 * c2 + 1], sc[8 * c2 + 2], sc[8 * c2 + 3], sc[8 * c2 + 4], sc[8 * c2 + 5], sc[8 * c2 + 6], sc[8 * c2 + 7]};
            uint4 ph, pl;
            split8(x, ph, pl);
#pragma unroll
            for (int d = 0; d < DB; d++) {
                const uint4 vh = *(const uint4 *)(sv + (32 * d + li) * VROW + c2 * 32 + lk * 16);
                const uint4 vl = *(const uint4 *)(sv + V_BYTES + (32 * d + li) * VROW + c2 * 32 + lk * 16);
                o[d] = mfma_bf16(vl, ph, o[d]);
                o[d] = mfma_bf16(vh, pl, o[d]);
                o[d] = mfma_bf16(vh, ph, o[d]);
            }
        }
        lstore(slot ^ 1);
        __syncthreads();
    }
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    // context rows through LDS ([query][feature], in the ring: nobody reads it behind the loop's last barrier): whole lines out.
    // The P.V tile has features on M and queries on N: lane (query li, half lk) holds features 32 d + (r & 3) + 8 (r >> 2) + 4 lk.
    float *t = (float *)ring + wave * (32 * LD);
#pragma unroll
    for (int d = 0; d < DB; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) t[li * LD + 32 * d + (r & 3) + 8 * (r >> 2) + 4 * lk] = o[d][r] * inv;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (ctx2) {                                      // the out-projection's operand on gemm.hip's tiles: rows [hi(H) | lo(H)] bf16
        for (int i = lane; i < 32 * HD / 2; i += 64) {
            const int qi = i / (HD / 2), c = (i % (HD / 2)) * 2;
            if (q0 + qi < S) {
                const float v0 = t[qi * LD + c], v1 = t[qi * LD + c + 1];
                const uint32_t hw = pack_bf16x2(v0, v1);
                const uint32_t lw = pack_bf16x2(v0 - __builtin_bit_cast(float, hw << 16), v1 - __builtin_bit_cast(float, hw & 0xffff0000u));
                uint16_t *o2 = ctx2 + (row0 + q0 + qi) * 2 * H + h * HD + c;
                *(uint32_t *)o2 = hw;
                *(uint32_t *)(o2 + H) = lw;
            }
        }
        return;
    }
    for (int i = lane; i < 32 * HD; i += 64) {
        const int qi = i / HD, c = i % HD;
        if (q0 + qi < S) ctx[(row0 + q0 + qi) * H + h * HD + c] = t[qi * LD + c];
    }
}

// rows of float32 -> [hi(K) | lo(K)] bf16 (hi = bf16(x), lo = bf16(x - hi)): the operand layout of gemm.hip's MODE 5 / 6
__global__ __launch_bounds__(256) void k_split_rows(const float *__restrict__ x, int64_t n4, int K4, uint16_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / K4; const int c = (int)(i - row * K4);
        const f32x4v v = *(const f32x4v *)(x + i * 4);
        const uint32_t h01 = pack_bf16x2(v[0], v[1]), h23 = pack_bf16x2(v[2], v[3]);
        const uint32_t l01 = pack_bf16x2(v[0] - __builtin_bit_cast(float, h01 << 16), v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u));
        const uint32_t l23 = pack_bf16x2(v[2] - __builtin_bit_cast(float, h23 << 16), v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u));
        uint16_t *o = out + row * 8 * K4 + c * 4;
        *(uint2 *)o = uint2{h01, h23};
        *(uint2 *)(o + 4 * K4) = uint2{l01, l23};
    }
}
int split_rows(const float *x, int64_t rows, int K, uint16_t *out, hipStream_t st) {
    if (K % 4) AK_FAIL(-1, "split_rows: K must be a multiple of 4");
    const int64_t n4 = rows * (K / 4);
    if (n4 == 0) return 0;
    const unsigned grid = (unsigned)(n4 / 256 + 1 < 65536 ? n4 / 256 + 1 : 65536);
    k_split_rows<<<grid, 256, 0, st>>>(x, n4, K / 4, out);
    AK_HIP(hipGetLastError());
    return 0;
}

// out = LayerNorm(y + r) * g + b, one wave per row, 16 bytes per lane and access (H % 4 == 0, H <= 1024); the row also leaves as
// [hi(H) | lo(H)] bf16 (out2): the next GEMM's operand
__global__ __launch_bounds__(256) void k3_add_ln(const float *y, int ldy, const float *r, int64_t T, int H, const float *__restrict__ g, const float *__restrict__ bta,
                                                 float eps, float *out, uint16_t *__restrict__ out2) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, H4 = H >> 2;
    if (row >= T) return;
    f32x4v v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane + 64 * j;
        v[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (i < H4) {
            v[j] = *(const f32x4v *)(y + row * ldy + i * 4);
            if (r) v[j] += *(const f32x4v *)(r + row * H + i * 4);
        }
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mu = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (lane + 64 * j < H4) {
            const f32x4v d = v[j] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane + 64 * j;
        if (i < H4) {
            const f32x4v gg = *(const f32x4v *)(g + i * 4), bb = *(const f32x4v *)(bta + i * 4);
            const f32x4v o = (v[j] - mu) * rstd * gg + bb;
            *(f32x4v *)(out + row * H + i * 4) = o;
            const uint32_t h01 = pack_bf16x2(o[0], o[1]), h23 = pack_bf16x2(o[2], o[3]);
            const uint32_t l01 = pack_bf16x2(o[0] - __builtin_bit_cast(float, h01 << 16), o[1] - __builtin_bit_cast(float, h01 & 0xffff0000u));
            const uint32_t l23 = pack_bf16x2(o[2] - __builtin_bit_cast(float, h23 << 16), o[3] - __builtin_bit_cast(float, h23 & 0xffff0000u));
            uint16_t *o2 = out2 + row * 2 * H + i * 4;
            *(uint2 *)o2 = uint2{h01, h23};
            *(uint2 *)(o2 + H) = uint2{l01, l23};
        }
    }
}
// embeddings of the split mode's tile path: LayerNorm(word[id] + pos[t % S] + type[0]) as float32 rows and as [hi | lo] rows, one wave
// per token, 16 bytes per lane and access (k32_embed of encoder.hip reads and writes single floats and leaves the split to a second pass)
__global__ __launch_bounds__(256) void k3_embed(const int *__restrict__ ids, int64_t T, int S, int H, int vocab, const float *__restrict__ word,
                                                const float *__restrict__ pos, const float *__restrict__ type, const float *__restrict__ g,
                                                const float *__restrict__ bta, float eps, float *__restrict__ out, uint16_t *__restrict__ out2) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, H4 = H >> 2;
    if (row >= T) return;
    int id = ids[row];
    if (id < 0 || id >= vocab) id = 0;
    const float *w = word + (int64_t)id * H, *p = pos + (int64_t)(row % S) * H;
    f32x4v v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane + 64 * j;
        v[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (i < H4) v[j] = (*(const f32x4v *)(w + i * 4) + *(const f32x4v *)(p + i * 4)) + *(const f32x4v *)(type + i * 4);
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mu = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (lane + 64 * j < H4) {
            const f32x4v d = v[j] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = lane + 64 * j;
        if (i < H4) {
            const f32x4v gg = *(const f32x4v *)(g + i * 4), bb = *(const f32x4v *)(bta + i * 4);
            const f32x4v o = (v[j] - mu) * rstd * gg + bb;
            *(f32x4v *)(out + row * H + i * 4) = o;
            const uint32_t h01 = pack_bf16x2(o[0], o[1]), h23 = pack_bf16x2(o[2], o[3]);
            const uint32_t l01 = pack_bf16x2(o[0] - __builtin_bit_cast(float, h01 << 16), o[1] - __builtin_bit_cast(float, h01 & 0xffff0000u));
            const uint32_t l23 = pack_bf16x2(o[2] - __builtin_bit_cast(float, h23 << 16), o[3] - __builtin_bit_cast(float, h23 & 0xffff0000u));
            uint16_t *o2 = out2 + row * 2 * H + i * 4;
            *(uint2 *)o2 = uint2{h01, h23};
            *(uint2 *)(o2 + H) = uint2{l01, l23};
        }
    }
}
int launch_embed_split(const int *ids, int64_t T, int S, int H, int vocab, const float *word, const float *pos, const float *type, const float *g,
                       const float *b, float eps, float *out, uint16_t *out2, hipStream_t st) {
    if (H % 4 || H > 1024) AK_FAIL(-1, "launch_embed_split: H must be a multiple of 4, at most 1024");
    k3_embed<<<(unsigned)((T + 3) / 4), 256, 0, st>>>(ids, T, S, H, vocab, word, pos, type, g, b, eps, out, out2);
    AK_HIP(hipGetLastError());
    return 0;
}
int launch_add_ln_split(const float *y, int ldy, const float *r, int64_t T, int H, const float *g, const float *b, float eps, float *out, uint16_t *out2,
                        hipStream_t st) {
    if (H % 4 || H > 1024) AK_FAIL(-1, "launch_add_ln_split: H must be a multiple of 4, at most 1024");
    k3_add_ln<<<(unsigned)((T + 3) / 4), 256, 0, st>>>(y, ldy, r, T, H, g, b, eps, out, out2);
    AK_HIP(hipGetLastError());
    return 0;
}

// grid of the float32-grade attention kernels: (sequence, head) groups dealt over the 8 XCDs, a group's query blocks 8 apart
static unsigned attn_grid(int B, int heads, int nqb) { return (unsigned)(((int64_t)B * heads + 7) / 8 * 8 * nqb); }

bool f32_mfma_supported(int H, int I, int heads) {
    const int hd = heads > 0 ? H / heads : 0;
    return H % 128 == 0 && I % 128 == 0 && H % 32 == 0 && I % 32 == 0 && (hd == 32 || hd == 64);
}

int launch_gemm_f32(int epi, const float *X, const float *W, const float *bias, const float *R, int T, int N, int K, float *Y, int ldc,
                    int col0, hipStream_t st) {
    if (N % G32_BN || K % G32_BK) AK_FAIL(-1, "launch_gemm_f32: N must be a multiple of 128, K of 32");
    static std::atomic<bool> attr{false};
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k32m_gemm<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G32_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k32m_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G32_LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k32m_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, G32_LDS));
        attr = true;
    }
    const dim3 grid(xcd_grid(T, N));
    if (epi == 0) k32m_gemm<0><<<grid, 256, G32_LDS, st>>>(X, W, bias, R, T, N, K, Y, ldc, col0);
    else if (epi == 1) k32m_gemm<1><<<grid, 256, G32_LDS, st>>>(X, W, bias, R, T, N, K, Y, ldc, col0);
    else k32m_gemm<2><<<grid, 256, G32_LDS, st>>>(X, W, bias, R, T, N, K, Y, ldc, col0);
    AK_HIP(hipGetLastError());
    return 0;
}

int launch_attn_x3(const float *qkv, const int *mask, int B, int S, int H, int heads, float *ctx, hipStream_t st) {
    const int hd = H / heads;
    const int nqb = (S + 127) / 128;
    const unsigned grid = attn_grid(B, heads, nqb);
    if (hd == 64) k3_attn<64><<<grid, 256, 0, st>>>(qkv, mask, B, S, H, heads, ctx, nullptr, 3 * H);
    else if (hd == 32) k3_attn<32><<<grid, 256, 0, st>>>(qkv, mask, B, S, H, heads, ctx, nullptr, 3 * H);
    else AK_FAIL(-1, "launch_attn_x3: head size must be 32 or 64");
    AK_HIP(hipGetLastError());
    return 0;
}
int launch_attn_x3_split(const float *qkv, int ldq, const int *mask, int B, int S, int H, int heads, uint16_t *ctx2, hipStream_t st) {
    const int hd = H / heads;
    const int nqb = (S + 127) / 128;
    const unsigned grid = attn_grid(B, heads, nqb);
    if (hd == 64) k3_attn<64><<<grid, 256, 0, st>>>(qkv, mask, B, S, H, heads, nullptr, ctx2, ldq);
    else if (hd == 32) k3_attn<32><<<grid, 256, 0, st>>>(qkv, mask, B, S, H, heads, nullptr, ctx2, ldq);
    else AK_FAIL(-1, "launch_attn_x3_split: head size must be 32 or 64");
    AK_HIP(hipGetLastError());
    return 0;
}

int launch_attn_f32(const float *qkv, const int *mask, int B, int S, int H, int heads, float *ctx, hipStream_t st) {
    const int hd = H / heads;
    const int nqb = (S + 127) / 128;
    const unsigned grid = attn_grid(B, heads, nqb);
    if (hd == 64) k32m_attn<64><<<grid, 256, 0, st>>>(qkv, mask, B, S, H, heads, ctx);
    else if (hd == 32) k32m_attn<32><<<grid, 256, 0, st>>>(qkv, mask, B, S, H, heads, ctx);
    else AK_FAIL(-1, "launch_attn_f32: head size must be 32 or 64");
    AK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ak
