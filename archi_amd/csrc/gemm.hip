// gemm.hip -- persistent bf16 MFMA GEMM of the encoder forward pass (a1/a2):
//   OUT[t][n] = sum_k X[t][k] * W[n][k] + bias[n]      X: [T][K] bf16, W: [N][K] bf16 (nn.Linear layout)
// with the epilogues the BERT layer needs fused in:
//   MODE 0  QKV projection: Q (pre-scaled by log2(e)/sqrt(hd)) and K as [T][H] bf16, V TRANSPOSED as
//           [B][H][S] bf16 (the layout the attention kernel's P.V MFMA wants)
//   MODE 1  bf16 output with exact (erf) GELU            (FFN up-projection)
//   MODE 2  fp32 output (the residual is added by the LayerNorm that follows; attention out-proj, FFN down-proj)
//   MODE 3  bf16 output
//   MODE 4  bf16 output + bf16 residual rows
//   MODE 5 / 6  the SPLIT-bf16 parity mode (precision 2, "bf16x3") on this tile: both operands arrive as bf16 hi | lo halves of
//           float32 values, rows [hi(K') | lo(K')], and the K-loop walks 3 K' steps: per k-tile X hi.W hi, X lo.W hi, X hi.W lo (the
//           k-tile index is mapped per operand; a.K = 3 K'). 5: fp32 output (QKV, out-projection, FFN-down); 6: exact erff GELU, output split
//           again into [hi(N) | lo(N)] rows (FFN-up: the next launch's X operand). See launch_gemm_x3w.
//           (Measured and not kept: a MODE 5 in which a workgroup walks whole 256-token row blocks -- all column tiles, one after
//           the other -- and normalises the rows it has just stored, from L2, instead of a LayerNorm launch: bge-base 128 x 512
//           17.1 ms against 15.6 with one row per wave at a time, 20.2 with eight rows in flight -- the tail's registers spill
//           into a kernel that has none to spare, and its loads run with two waves per SIMD to hide them.) (what the LayerNorm of the hidden != 384 path reads: one array instead of two)
// Replaces the torch CPU GEMMs behind SentenceTransformer.encode as called at
// /root/reference/src/data_manager/vectorstore/manager.py:373.
//
// Structure: 8 waves, tile = 128 output features (MFMA A side: weight rows) x 256 tokens (B side),
// K-step 64, 3-slot LDS ring filled by global_load_lds, same source-side XOR swizzle as scan.hip.
// The TOKEN is on the MFMA column axis: a lane owns one token and 4 consecutive output features per
// accumulator group. Storing straight from that layout moves every 128-byte output line in 8 partial
// requests, so each wave's block goes through a 4 KB LDS scratch and leaves as full lines (rows_out,
// rows_out_f32, v_out below).
// Workgroups are persistent over tiles (feature tile fastest, so concurrently running workgroups
// share the X tile through L2) and keep prefetching across tile boundaries.
//
// Epilogue notes (measured by ablation on the 65536-token MiniLM FFN-up launch, 160 us): MFMA loop alone 59 us,
// +30 ring loads, +43 epilogue VALU (64 GELUs per lane per tile), +28 stores. Tried and measured slower or equal,
// so NOT kept: a 128x128 tile with two workgroups per CU (173 us); issuing the stores late with widened vmcnt
// waits (159 us); parking the finished tile in a second accumulator set and running its epilogue between the next
// tile's MFMAs (378 us: 256 VGPRs spill the parked tile to scratch); V tiles with swapped MFMA operands for
// 8-byte transposed stores (138 vs 123 us). What is kept: packed fp32 math (v_pk_fma_f32), a polynomial erf
// without rcp/exp, one-instruction bf16 conversion, and the full-line stores (FFN-up 152 -> 146 us, QKV 117 -> 98 us).
// Also measured and not kept (wide tile, bge-base shapes): starting every other workgroup of an XCD half a tile period
// late so that epilogues of one half overlap K-loops of the other -- the GEMM launches gain 2-3%, the forward pass
// nothing (16.46 vs 16.46-16.52 ms): the epilogue is bound by its own VALU / LDS / store-issue work, not by an HBM burst.
// Also measured and not kept: the 128 x 256 tile on FOUR waves (64 features x 128 tokens each, K-step 32, 3-slot ring of
// 24 KB) so that two independent workgroups share a CU, the second started half a tile period late, one's epilogue under
// the other's K-loop: MiniLM forward 2.585 -> 2.78 ms, bge-base (narrow tile) 18.4 -> 19.4 ms -- twice the barriers per
// MFMA and 16 MFMAs per barrier interval cost more than the overlap returns.
// Round 3, wide tile with the phased K-loop, bge-base 128 x 512 (kernel-trace averages, AK_GEMM_ABLATE): FFN-up 358 us whole, 251
// with the epilogue's global stores compiled out, 173 without the epilogue; out-projection / FFN-down 193 / 143 / 126. The same
// stores aimed at 1.5 MB that stays in L2 (every tile into the same rows): 278 -- it is the 403 MB of NEW lines per launch, not
// the store instructions. Starting the workgroups of an XCD (or alternate XCDs) 7-42 us apart so that their store phases do not
// coincide: no change (355-365). `global_store_dwordx4 ... nt` from inline asm: 347 -> 337 and 187 -> 180, but the embeddings of
// a 6 144-token batch (persistent workgroups, several tiles each) then differed from process to process -- not kept. The FFN-up
// output in K-tile slabs ([token tile][feature / 64][256 tokens][64]: every store instruction one contiguous KiB instead of eight
// 128-byte pieces 6 KB apart; timing only): 347.5 against 346.5 -- not DRAM page locality either.
// Also measured and not kept: K-step 32 with a 5-slot ring (what gained 12% in gemm_ln.hip's 2-slot loop): FFN-up
// 143 -> 151 us, QKV 99 -> 105 us -- a 3-slot ring already hides the load latency, the extra barriers only cost.
#include <atomic>
#include <cmath>
#include <mutex>
#include <vector>
#include "mfma_tile.h"
#include "encoder_kernels.h"
#include "switches.h"
#include "gelu_table.h"

namespace ak {
using namespace mt;

// Two feature-tile widths. BN = 128: 64 x 64 per wave, 3-slot ring of 48 KB. BN = 256 (N % 256 == 0 and enough tiles to
// fill the chip): 128 x 64 per wave (128 accumulators, the scan kernel's wave tile), 2-slot ring of 64 KB -- 128 instead
// of 87 flops per byte staged from L2 and 0.75 instead of 1 KB of fragment reads per MFMA. The bge-base GEMMs (K = 768 /
// 3072) ran at 0.74-0.86 PF on the narrow tile where hipBLASLt's 256x256 macro-tile does 1.05-1.33 PF on the same shapes.
constexpr int G_BT = 256, G_NW = 8, G_THREADS = 512;
constexpr int G_X_BYTES = G_BT * 128, G_X_PW = G_BT / 8 / G_NW;   // 4 pieces of X per wave per K-step
template <int BN> struct GCfg {
    static constexpr int MI = BN / 64;               // 32-feature MFMA row blocks per wave
    static constexpr int WF = BN / 2;                // features per wave row (2 wave rows)
    static constexpr int NSTAGE = BN == 128 ? 3 : 2;
    static constexpr int W_BYTES = BN * 128, W_PW = BN / 8 / G_NW, LOADS = W_PW + G_X_PW;
    static constexpr int LDS = NSTAGE * (W_BYTES + G_X_BYTES) + 2 * BN * 4;   // + bias of the tile, by tile parity
};


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// round-to-nearest-even conversion, two values per instruction (v_cvt_pk_bf16_f32, gfx950)
__device__ inline uint2 cvt_bf16x4(f32x4 v) { return __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4)); }

// exact-GELU 0.5 x (1 + erf(x / sqrt 2)) on four values (two independent v_pk_* chains, so the dependent
// Horner steps of one hide under the other). erf(u) = u Q(u^2) on |u| <= 3.2 with a degree-9 minimax Q fitted
// offline against scipy's erf (max abs error 7.8e-6; 1 - erf(3.2) = 6e-6), u clamped to +-3.2 and the result to
// +-1; odd in u, so no abs/sign handling. No rcp, no exp: the output is rounded to bf16 (2^-9 relative) anyway.
__device__ inline float clamp3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ inline f32x4 gelu_erf4(f32x4 x) {
    f32x4 u = x * 0.70710678118654752f;
    u = {clamp3(u.x, -3.2f, 3.2f), clamp3(u.y, -3.2f, 3.2f), clamp3(u.z, -3.2f, 3.2f), clamp3(u.w, -3.2f, 3.2f)};
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
    f32x4 e = p * u;                                                   // erf(x / sqrt 2)
    e = {clamp3(e.x, -1.f, 1.f), clamp3(e.y, -1.f, 1.f), clamp3(e.z, -1.f, 1.f), clamp3(e.w, -1.f, 1.f)};
    const f32x4 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, e, hx);
}

// GELU OF THE SPLIT MODE (MODE 6). gelu(x) = x Phi(x) must come out at float32 grade there, and erff is ~35 vector instructions per
// value: 4.8 k of them per wave and tile, more SIMD time than the tile's MFMAs at K = 3 x 384 (r6q2 trace: FFN-up at 0.79 PF where
// the other launches of the mode run at 1.1). Phi is smooth and bounded, so it is read from a table of CUBIC PIECES instead: 384
// intervals of 1 / 32 over [-6, 6), four float32 coefficients each (Hermite data from erfc / the density in double precision,
// interpolation error <= h^4 / 384 x max |4th derivative of Phi| = 1.4e-9), 6 KB in LDS behind the tile's ring: index and fraction
// from one fma, one ds_read_b128, three fma and the product with x -- 11 instructions. Outside the range the end pieces hold
// (Phi(-6) = 1e-9). Max |error| against the double-precision function over |x| <= 8: 4.9e-7; the float32 erff formula: 4.5e-7
// (both: the rounding of the product x Phi).
constexpr int PHI_N = 384, PHI_BYTES = PHI_N * 16;
static const float *g_phi_tab = nullptr;
static int phi_table_create() {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_phi_tab) return 0;
    std::vector<float> t(PHI_N * 4);
    const double h = 1.0 / 32.0, r2 = 0.70710678118654752440, rs2pi = 0.39894228040143267794;
    auto Phi = [&](double x) { return 0.5 * erfc(-x * r2); };
    auto phi = [&](double x) { return exp(-0.5 * x * x) * rs2pi; };
    for (int i = 0; i < PHI_N; i++) {
        const double x0 = -6.0 + i * h, x1 = x0 + h, d = Phi(x1) - Phi(x0), f0 = phi(x0), f1 = phi(x1);
        t[4 * i + 0] = (float)Phi(x0); t[4 * i + 1] = (float)(h * f0);
        t[4 * i + 2] = (float)(3.0 * d - h * (2.0 * f0 + f1)); t[4 * i + 3] = (float)(-2.0 * d + h * (f0 + f1));
    }
    float *dv;
    AK_HIP(hipMalloc((void **)&dv, PHI_BYTES));
    AK_HIP(hipMemcpy(dv, t.data(), PHI_BYTES, hipMemcpyHostToDevice));
    g_phi_tab = dv;
    return 0;
}

// LZ: LAZY LayerNorm (GemmArgs). The hidden-768 path used to run GEMM -> k_layernorm16 twice per layer: 2 x 38 us of pure HBM
// traffic per layer (6.5 % of a bge-base forward) for an operation that is two scalars per token. With LZ the sub-layer outputs
// stay un-normalised: a MODE 4 launch adds the (normalised-on-the-fly) residual and writes, per token, the partial sums of its
// output row r and the row itself SCALED COLUMN-WISE by the gamma of the LayerNorm that follows, r~ = gamma (.) r (bf16) -- a
// column scaling needs no row statistics. Whoever reads such rows finishes the LayerNorm where it has the row's (mean, 1/std):
//     residual:  LN(r)_k = rstd (r~_k - mu gamma_k) + beta_k                        (no division by gamma)
//     A operand: LN(r) W^T + b = rstd (r~ W^T - mu c) + b',   c = W gamma,  b' = b + W beta   (a lane owns a token)
// against the UNCHANGED weights. (First version: raw rows and gamma folded into bf16 copies of the weights. A rounded weight is
// wrong the same way for every token: bge-base 1 - cos against the fp32 oracle 5e-5 where the un-folded path has 2e-5, in a
// CPU model of the roundings and on the GPU alike; rounding gamma r instead is an activation rounding like any other.)
template <int MODE, int G_BN, bool PH = false, bool LZ = false>     // PH: the phased K-loop of scan.hip (wide tile only)
__global__ __launch_bounds__(G_THREADS, 2) void k_gemm(GemmArgs a) {
    using C = GCfg<G_BN>;
    constexpr int MI = C::MI, WF = C::WF, G_NSTAGE = C::NSTAGE, G_W_BYTES = C::W_BYTES, G_W_PW = C::W_PW, G_LOADS = C::LOADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // MODE 1 on the wide tile reads its GELU from the LDS table (gelu_table.h), which must sit at LDS address 0: everything else
    // moves up by 16 KB (130 + 16 of the 160 KB; the narrow tile's three-slot ring has no room for it and keeps the polynomial)
    constexpr bool GTAB = MODE == 1 && G_BN == 256;
    char *lbase = smem + (GTAB ? GELU_TAB_BYTES : 0);
    if constexpr (GTAB) {
        if (lds_addr(smem) != 0) __builtin_trap();
        for (int i = threadIdx.x; i < GELU_TAB_BYTES / 16; i += G_THREADS) *(uint4 *)(smem + i * 16) = ((const uint4 *)a.gelu_tab)[i];
        __syncthreads();
    }
    const char *s_phi = lbase + C::LDS;      // MODE 6: the cubic pieces of Phi, behind the ring and the biases
    if constexpr (MODE == 6) {
        for (int i = threadIdx.x; i < PHI_N; i += G_THREADS) *(uint4 *)(lbase + C::LDS + i * 16) = ((const uint4 *)a.phi_tab)[i];
        __syncthreads();
    }
    char *sW = lbase;
    char *sX = lbase + G_NSTAGE * G_W_BYTES;
    float *s_bias = (float *)(lbase + G_NSTAGE * (G_W_BYTES + G_X_BYTES));   // [2][G_BN]
    // LZ, behind the biases: [2][G_BN] fold_c (MODE 0 / 1) or gamma | beta of the residual's LayerNorm (MODE 4: 2 x [2][G_BN]), then
    // the (mean, 1 / std) pairs of the tile's 256 tokens, [2][G_BT] float2 -- all by tile parity, staged one tile ahead like the biases
    constexpr bool AFOLD = LZ && MODE != 4;
    float *s_c = s_bias + 2 * G_BN, *s_g = s_c, *s_b = s_g + 2 * G_BN, *s_g2 = s_b + 2 * G_BN;      // MODE 4: + gamma of the LayerNorm that follows
    float2 *s_st = (float2 *)(s_bias + 2 * G_BN + (AFOLD ? 2 : 6) * G_BN);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;     // 2 (features) x 4 (tokens) waves, WF features x 64 tokens each
    const int ntn = a.N / G_BN, ntt = a.T / G_BT, ntiles = ntn * ntt;
    const int KS = a.K / 64;
    // split operands (MODE 5 / 6): rows of 2 K' elements [hi | lo]; KS3 = k-tiles per half
    constexpr bool X3 = MODE >= 5;
    const int ldk = X3 ? a.K / 3 * 2 : a.K, KS3 = KS / 3;
    // step kk of the walk = term kk % 3 of k-tile kk / 3: (X hi, W hi), (X lo, W hi), (X hi, W lo) -- the two uses of a half are at most
    // two steps apart, so the second comes from L2 (walking the three terms as three passes over K fetched every hi half twice from
    // the fabric: PMC, bge-base FFN-down 2.3 GB per launch for 0.8 GB of operands)
    auto kt_w = [&](int kk) { if (!X3) return kk; const int q = kk / 3; return q + (kk - 3 * q == 2 ? KS3 : 0); };
    auto kt_x = [&](int kk) { if (!X3) return kk; const int q = kk / 3; return q + (kk - 3 * q == 1 ? KS3 : 0); };
    // Tile order. Block b runs on XCD b % 8, and each XCD has its own L2: with the feature tile simply fastest, the ntn column
    // tiles that share a 256-token X tile landed on eight XCDs and X came from HBM once per XCD (PMC, bge-base FFN-up: 943 MB
    // fetched per launch for 105 MB of operands, on top of the 403 MB it writes). XCD x therefore owns the token tiles x, x + 8, ..
    // and its workgroups walk THAT list, feature tile fastest: an X tile is fetched by one XCD only, W stays resident in each L2.
    // (only where the eight lists come out even -- or long enough for a token tile more or less not to matter: with 9 token tiles
    // XCD 0 would walk two of them while the others walk one)
    const bool xcd_order = (gridDim.x & 7) == 0 && ntt >= 8 && ((ntt & 7) == 0 || ntiles >= 4 * (int)gridDim.x) && !(a.flags & 8);
    const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3, xslots = gridDim.x >> 3;
    const int q_total = xcd_order ? ((ntt - xcd + 7) >> 3) * ntn : ntiles;           // tiles of this XCD (or of the launch)
    const int q_first = xcd_order ? xslot : (int)blockIdx.x, q_step = xcd_order ? xslots : (int)gridDim.x;
    const int my_tiles = q_first < q_total ? (q_total - 1 - q_first) / q_step + 1 : 0;
    // Feature blocks (round 5). Walking ALL ntn column tiles of a token tile before the next token tile keeps ntn W tiles
    // (393 KB each at K = 768: 4.7 MB for the FFN-up's 12) live in a 4 MB L2 beside the streaming X tiles: PMC, bge-base FFN-up
    // 736 MB fetched per launch for 105 MB of operands, QKV 477 for 104 (profiles/r05_pmc_fetch_encoder.json) -- W thrashes. With
    // fb > 0 an XCD walks its token tiles once per BLOCK of fb column tiles (column fastest inside the block): fb W tiles stay
    // resident, X is streamed ntn / fb times (from the Infinity Cache after the first). a.fb == 0: one block = the old order.
    const int fb = (xcd_order && a.fb > 0 && a.fb < ntn) ? a.fb : ntn;
    const int ntt_x = xcd_order ? (ntt - xcd + 7) >> 3 : ntt;                       // token tiles in this list
    auto tile_of = [&](int ord, int &tn, int &tt) {      // the ord-th tile of this workgroup (clamped to its last one)
        int q = q_first + ord * q_step;
        if (q >= q_total) q = q_total - 1;
        const int per_block = fb * ntt_x;                 // tiles of one full feature block
        const int f = q / per_block, rem = q - f * per_block;
        const int bw = ntn - f * fb < fb ? ntn - f * fb : fb;        // the last block may be narrower
        const int ti = rem / bw;
        tn = f * fb + (rem - ti * bw);
        tt = xcd_order ? xcd + 8 * ti : ti;
    };
    const int nsteps = my_tiles * KS;

    const int r = lane & 31, kh = lane >> 5;
    const int c0 = kh ^ ((r >> 1) & 7);
    const int a_off = (wr * WF + r) * 128, b_off = (wc * 64 + r) * 128;

    const int st_row = lane >> 3, st_chunk = lane & 7;
    const uint32_t ldsW = lds_addr(sW) + wave * G_W_PW * 1024, ldsX = lds_addr(sX) + wave * G_X_PW * 1024;
    const char *wptr[G_W_PW];
    const char *xptr[G_X_PW];
    int s_t = 0, s_kk = 0, s_buf = 0, issued = 0;
    auto set_ptrs = [&](int ord) {
        int tn, tt;
        tile_of(ord, tn, tt);
#pragma unroll
        for (int p = 0; p < G_W_PW; p++) {
            int row = (wave * G_W_PW + p) * 8 + st_row;
            wptr[p] = (const char *)a.W + ((int64_t)(tn * G_BN + row) * ldk) * 2 + ((st_chunk ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int p = 0; p < G_X_PW; p++) {
            int row = (wave * G_X_PW + p) * 8 + st_row;
            xptr[p] = (const char *)a.X + ((int64_t)(tt * G_BT + row) * ldk) * 2 + ((st_chunk ^ ((row >> 1) & 7)) << 4);
        }
    };
    set_ptrs(0);
    auto stage_next = [&]() {
        glds16xN<G_W_PW>(wptr, kt_w(s_kk) * 128, __builtin_amdgcn_readfirstlane(ldsW + s_buf * G_W_BYTES));
        glds16xN<G_X_PW>(xptr, kt_x(s_kk) * 128, __builtin_amdgcn_readfirstlane(ldsX + s_buf * G_X_BYTES));
        s_buf = (s_buf + 1 == G_NSTAGE) ? 0 : s_buf + 1;
        if (++s_kk == KS) { s_kk = 0; s_t++; set_ptrs(s_t); }
        issued++;
    };

    f32x16 acc[MI][2];
    int p_tn = 0, p_tt = 0, p_par = 0;
    // one store group of the finished tile: P = ni*(MI*4) + mi*4 + g; lane owns token t (column) and 4 consecutive
    // features per accumulator group
    auto piece = [&](auto pc) {
        constexpr int P = decltype(pc)::value, ni = P / (MI * 4), mi = (P >> 2) % MI, g = P & 3;
        const f32x16 &v = acc[mi][ni];
        const int t = p_tt * G_BT + wc * 64 + ni * 32 + r;
        const int n = p_tn * G_BN + wr * WF + mi * 32 + 8 * g + 4 * kh;
        const float4 bi = *(const float4 *)&s_bias[p_par * G_BN + wr * WF + mi * 32 + 8 * g + 4 * kh];
        const f32x4 o = {v[4 * g + 0] + bi.x, v[4 * g + 1] + bi.y, v[4 * g + 2] + bi.z, v[4 * g + 3] + bi.w};
        if constexpr (MODE == 0) {
            if (n < a.H) *(uint2 *)(a.q + (int64_t)t * a.H + n) = cvt_bf16x4(o * a.qscale);
            else if (n < 2 * a.H) *(uint2 *)(a.k + (int64_t)t * a.H + (n - a.H)) = cvt_bf16x4(o);
            else if (t < a.ldo) {
                // V transposed [B][H][S]: 4 features x this lane's token -> 4 two-byte stores, a wave's lanes
                // (consecutive tokens) fill 64 contiguous bytes per feature row. (Computing V tiles with the MFMA
                // operands swapped, so that a lane holds 4 consecutive tokens, measured slower: 138 vs 123 us.)
                const int b = t / a.S, sq = t - b * a.S;
                const uint2 h = cvt_bf16x4(o);
                uint16_t *p = a.vt + ((int64_t)b * a.H + (n - 2 * a.H)) * a.S + vt_pos(sq);
                p[0] = (uint16_t)h.x; p[a.S] = (uint16_t)(h.x >> 16);
                p[2 * (int64_t)a.S] = (uint16_t)h.y; p[3 * (int64_t)a.S] = (uint16_t)(h.y >> 16);
            }
        } else if constexpr (MODE == 1) {
            *(uint2 *)(a.out_bf16 + (int64_t)t * a.ldo + n) = GTAB ? f_gelu_tab4(o) : cvt_bf16x4(gelu_erf4(o));
        } else if constexpr (MODE == 2 || MODE == 5) {
            *(f32x4 *)(a.out_f32 + (int64_t)t * a.N + n) = o;   // the residual is added by the LayerNorm kernel that follows
        } else {
            *(uint2 *)(a.out_bf16 + (int64_t)t * a.ldo + n) = cvt_bf16x4(o);
        }
    };
    auto all_pieces = [&]() {
#define PC(i) if constexpr (i < MI * 8) piece(std::integral_constant<int, i>{});
        PC(0) PC(1) PC(2) PC(3) PC(4) PC(5) PC(6) PC(7) PC(8) PC(9) PC(10) PC(11) PC(12) PC(13) PC(14) PC(15)
        PC(16) PC(17) PC(18) PC(19) PC(20) PC(21) PC(22) PC(23) PC(24) PC(25) PC(26) PC(27) PC(28) PC(29) PC(30) PC(31)
#undef PC
    };
    // bf16 outputs as full lines. In the accumulator layout a wave's store instruction touches 32 token rows with 16
    // bytes each (kh pairs), i.e. every 128-byte line of the output is written in 8 partial requests. Here the wave's
    // 64 features x 32 tokens go through a 4 KB wave-private LDS scratch (its own X staging pieces in the ring slot
    // that was just consumed; 16-byte chunks XOR-swizzled by the token row) and leave as 16 bytes per lane, 8 lanes
    // per token row: one full 128-byte line per row. Used for GELU / plain bf16 outputs and the Q and K tiles.
    // LZ MODE 4 transposes in FLOAT32 (8 KB per wave: 32 token rows x 256 B, 16-byte chunks XOR-swizzled by the row): acc + bias
    // meets the residual unrounded and the row is rounded ONCE, when gamma (.) r is stored (the bf16 scratch rounded the GEMM
    // output before the add: one rounding more per sub-layer, 24 per bge-base forward).
    constexpr bool F32T = LZ && MODE == 4;
    auto rows_out = [&](char *scr, uint16_t *base, int ld, int col0, float scale) {
        const int rl_tok = lane >> 3, rl_c = lane & 7;
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            float a_rs = 1.f, a_nm = 0.f;                // AFOLD: 1 / std and -mean / std of this lane's token (column r of block ni)
            if constexpr (AFOLD) { const float2 ms = s_st[p_par * G_BT + wc * 64 + ni * 32 + r]; a_rs = ms.y; a_nm = -ms.x * ms.y; }
            float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};     // LZ MODE 4: sums of the rows written (tokens rl_tok + 8 i)
            float r_mu[4] = {0.f, 0.f, 0.f, 0.f}, r_rs[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (LZ && MODE == 4) {
                if (a.res_stats) {
#pragma unroll
                    for (int i = 0; i < 4; i++) { const float2 ms = s_st[p_par * G_BT + wc * 64 + ni * 32 + rl_tok + 8 * i]; r_mu[i] = ms.x; r_rs[i] = ms.y; }
                }
            }
#pragma unroll
            for (int hf = 0; hf < MI / 2; hf++) {        // 64 features x 32 tokens per pass through the scratch
                const int fb = wr * WF + hf * 64;
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const f32x16 &v = acc[hf * 2 + mi][ni];
                        const float4 bi = *(const float4 *)&s_bias[p_par * G_BN + fb + mi * 32 + 8 * g + 4 * kh];
                        f32x4 o;
                        if constexpr (AFOLD) {
                            const float4 cc = *(const float4 *)&s_c[p_par * G_BN + fb + mi * 32 + 8 * g + 4 * kh];      // rstd (v - mu c) + b'
                            o = {fmaf(a_rs, v[4 * g + 0], fmaf(a_nm, cc.x, bi.x)), fmaf(a_rs, v[4 * g + 1], fmaf(a_nm, cc.y, bi.y)),
                                 fmaf(a_rs, v[4 * g + 2], fmaf(a_nm, cc.z, bi.z)), fmaf(a_rs, v[4 * g + 3], fmaf(a_nm, cc.w, bi.w))};
                        } else o = {v[4 * g + 0] + bi.x, v[4 * g + 1] + bi.y, v[4 * g + 2] + bi.z, v[4 * g + 3] + bi.w};
                        if constexpr (MODE == 1 && !GTAB) o = gelu_erf4(o);
                        if constexpr (MODE == 0) o = o * scale;
                        if constexpr (F32T) *(f32x4 *)(scr + r * 256 + (((mi * 8 + 2 * g + kh) ^ ((r & 7) << 1)) << 4)) = o;
                        else *(uint2 *)(scr + r * 128 + (((mi * 4 + g) ^ (r & 7)) << 4) + kh * 8) = GTAB ? f_gelu_tab4(o) : cvt_bf16x4(o);
                    }
                uint4 resl[4];
                if constexpr (MODE == 4) {       // the residual rows of this pass: requested before the transposition, added after it
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int t = p_tt * G_BT + wc * 64 + ni * 32 + rl_tok + 8 * i;
                        resl[i] = *(const uint4 *)(a.res16 + (int64_t)t * ld + col0 + fb + rl_c * 8);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int tok = rl_tok + 8 * i;
                    uint4 line;
                    float lf[8];
                    if constexpr (F32T) {
                        const f32x4 l0 = *(const f32x4 *)(scr + tok * 256 + (((2 * rl_c) ^ ((tok & 7) << 1)) << 4));
                        const f32x4 l1 = *(const f32x4 *)(scr + tok * 256 + (((2 * rl_c + 1) ^ ((tok & 7) << 1)) << 4));
                        lf[0] = l0[0]; lf[1] = l0[1]; lf[2] = l0[2]; lf[3] = l0[3]; lf[4] = l1[0]; lf[5] = l1[1]; lf[6] = l1[2]; lf[7] = l1[3];
                    } else line = *(const uint4 *)(scr + tok * 128 + ((rl_c ^ (tok & 7)) << 4));
                    const int t = p_tt * G_BT + wc * 64 + ni * 32 + tok;
                    if constexpr (MODE == 4) {
                        uint32_t lw[4] = {0, 0, 0, 0};
                        if constexpr (!F32T) { lw[0] = line.x; lw[1] = line.y; lw[2] = line.z; lw[3] = line.w; }
                        const uint32_t rw[4] = {resl[i].x, resl[i].y, resl[i].z, resl[i].w};
                        uint32_t ow[4];
                        if constexpr (LZ) {
                            float rv[8];
#pragma unroll
                            for (int q = 0; q < 4; q++) { rv[2 * q] = bf16_to_f32((uint16_t)rw[q]); rv[2 * q + 1] = bf16_to_f32((uint16_t)(rw[q] >> 16)); }
                            if (a.res_stats) {                  // the residual rows hold gamma (.) r: LN(r) = rstd (r~ - mu gamma) + beta
                                const float *sg = &s_g[p_par * G_BN + fb + rl_c * 8], *sb = &s_b[p_par * G_BN + fb + rl_c * 8];
#pragma unroll
                                for (int e = 0; e < 8; e += 4) {
                                    const float4 g4 = *(const float4 *)(sg + e), b4 = *(const float4 *)(sb + e);
                                    rv[e + 0] = fmaf(r_rs[i], fmaf(-r_mu[i], g4.x, rv[e + 0]), b4.x); rv[e + 1] = fmaf(r_rs[i], fmaf(-r_mu[i], g4.y, rv[e + 1]), b4.y);
                                    rv[e + 2] = fmaf(r_rs[i], fmaf(-r_mu[i], g4.z, rv[e + 2]), b4.z); rv[e + 3] = fmaf(r_rs[i], fmaf(-r_mu[i], g4.w, rv[e + 3]), b4.w);
                                }
                            }
                            float g2[8];                        // the column scale of the rows this launch stores (read per token row: 24 registers less across the loop)
                            {
                                const float4 g0 = *(const float4 *)&s_g2[p_par * G_BN + fb + rl_c * 8], g1 = *(const float4 *)&s_g2[p_par * G_BN + fb + rl_c * 8 + 4];
                                g2[0] = g0.x; g2[1] = g0.y; g2[2] = g0.z; g2[3] = g0.w; g2[4] = g1.x; g2[5] = g1.y; g2[6] = g1.z; g2[7] = g1.w;
                            }
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const float w0 = lf[2 * q] + rv[2 * q], w1 = lf[2 * q + 1] + rv[2 * q + 1];                                             // the row r
                                st_s[i] += w0 + w1;
                                st_q[i] = fmaf(w0, w0, fmaf(w1, w1, st_q[i]));
                                ow[q] = pack_bf16x2(w0 * g2[2 * q], w1 * g2[2 * q + 1]);                                                                  // stored: gamma (.) r
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                ow[q] = pack_bf16x2(bf16_to_f32((uint16_t)lw[q]) + bf16_to_f32((uint16_t)rw[q]),
                                                    bf16_to_f32((uint16_t)(lw[q] >> 16)) + bf16_to_f32((uint16_t)(rw[q] >> 16)));
                        }
                        line = uint4{ow[0], ow[1], ow[2], ow[3]};
                    }
                    *(uint4 *)(base + (int64_t)t * ld + col0 + fb + rl_c * 8) = line;
                }
            }
            if constexpr (LZ && MODE == 4) {             // this wave's 128 features of tokens rl_tok + 8 i: 8 lanes (rl_c) per token
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int off = 1; off < 8; off <<= 1) { st_s[i] += __shfl_xor(st_s[i], off); st_q[i] += __shfl_xor(st_q[i], off); }
                    if (rl_c == 0) {
                        const int t = p_tt * G_BT + wc * 64 + ni * 32 + rl_tok + 8 * i;
                        const int slot = (col0 + wr * WF) / 128;
                        *(float2 *)(a.out_stats + ((int64_t)slot * a.T + t) * 2) = float2{st_s[i], st_q[i]};
                    }
                }
            }
        }
    };
    // fp32 output (MODE 2): same idea per 32-feature block (32 tokens x 128 B = the 4 KB scratch)
    auto rows_out_f32 = [&](char *scr) {
        const int rl_tok = lane >> 3, rl_c = lane & 7;
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const f32x16 &v = acc[mi][ni];
                    const float4 bi = *(const float4 *)&s_bias[p_par * G_BN + wr * WF + mi * 32 + 8 * g + 4 * kh];
                    const float4 o = {v[4 * g + 0] + bi.x, v[4 * g + 1] + bi.y, v[4 * g + 2] + bi.z, v[4 * g + 3] + bi.w};
                    *(float4 *)(scr + r * 128 + (((2 * g + kh) ^ (r & 7)) << 4)) = o;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int tok = rl_tok + 8 * i;
                    const float4 line = *(const float4 *)(scr + tok * 128 + ((rl_c ^ (tok & 7)) << 4));
                    const int t = p_tt * G_BT + wc * 64 + ni * 32 + tok;
                    if constexpr (MODE == 5) {      // an output width padded onto the wide tile: the padding columns stay unwritten
                        if (p_tn * G_BN + wr * WF + mi * 32 + rl_c * 4 >= a.nvalid) continue;
                    }
                    *(float4 *)(a.out_f32 + (int64_t)t * a.N + p_tn * G_BN + wr * WF + mi * 32 + rl_c * 4) = line;
                }
            }
    };
    // MODE 6: exact GELU of acc + bias, split into bf16 hi and lo = bf16(v - hi), each half through the same 4 KB scratch as full lines:
    // row t of the output is [hi(N) | lo(N)] (ld = a.ldo = 2 N)
    auto rows_out_split = [&](char *scr, int col0) {
        const int rl_tok = lane >> 3, rl_c = lane & 7;
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int hf = 0; hf < MI / 2; hf++) {
                const int fb = wr * WF + hf * 64;
                f32x4 o[8];
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const f32x16 &v = acc[hf * 2 + mi][ni];
                        const float4 bi = *(const float4 *)&s_bias[p_par * G_BN + fb + mi * 32 + 8 * g + 4 * kh];
                        f32x4 x = {v[4 * g + 0] + bi.x, v[4 * g + 1] + bi.y, v[4 * g + 2] + bi.z, v[4 * g + 3] + bi.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {      // x Phi(x), Phi from its cubic pieces (GELU OF THE SPLIT MODE above)
                            const float t = __builtin_amdgcn_fmed3f(__builtin_fmaf(x[e], 32.0f, 192.0f), 0.0f, 383.99997f);
                            const int iv = (int)t;
                            const float d = t - (float)iv;
                            const float4 c = *(const float4 *)(s_phi + iv * 16);
                            x[e] = x[e] * __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c.w, d, c.z), d, c.y), d, c.x);
                        }
                        o[mi * 4 + g] = x;
                    }
#pragma unroll
                for (int pass = 0; pass < 2; pass++) {
#pragma unroll
                    for (int mi = 0; mi < 2; mi++)
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            f32x4 &x = o[mi * 4 + g];
                            const uint2 h = cvt_bf16x4(x);
                            if (pass == 0)
                                x = x - f32x4{__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                                              __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u)};
                            *(uint2 *)(scr + r * 128 + (((mi * 4 + g) ^ (r & 7)) << 4) + kh * 8) = h;
                        }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int tok = rl_tok + 8 * i;
                        const uint4 line = *(const uint4 *)(scr + tok * 128 + ((rl_c ^ (tok & 7)) << 4));
                        const int t = p_tt * G_BT + wc * 64 + ni * 32 + tok;
                        *(uint4 *)(a.out_bf16 + (int64_t)t * a.ldo + pass * a.N + col0 + fb + rl_c * 8) = line;
                    }
                }
            }
    };
    // V tiles: the same scratch holds the wave's block TRANSPOSED ([64 features][32 tokens] bf16, 64-byte rows), so the
    // transposed output [B][H][S] is written 16 bytes per lane, 4 lanes per feature row (64 contiguous bytes) instead
    // of one 2-byte store per element. S is a multiple of 32 and so is each 32-token block's first token: the batch
    // row and the offset inside the sequence are uniform per block; blocks past the last real token are skipped.
    auto v_out = [&](char *scr) {
        // (the lane's constants of this epilogue are recomputed per tile from an opaque copy of the lane id: hoisted out of the
        // tile loop they were parked in scratch across the K-loop -- the 11 spilled registers of the QKV launch)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int rl_f = ln >> 2, rl_c = ln & 3, r = ln & 31, kh = ln >> 5;
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int t0 = __builtin_amdgcn_readfirstlane(p_tt * G_BT + wc * 64 + ni * 32);
            if (t0 >= a.ldo) continue;
            const int b = t0 / a.S, s0 = t0 - b * a.S;
            float a_rs = 1.f, a_nm = 0.f;
            if constexpr (AFOLD) { const float2 ms = s_st[p_par * G_BT + wc * 64 + ni * 32 + r]; a_rs = ms.y; a_nm = -ms.x * ms.y; }
#pragma unroll
            for (int hf = 0; hf < MI / 2; hf++) {
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const f32x16 &v = acc[hf * 2 + mi][ni];
                        const int f = mi * 32 + 8 * g + 4 * kh;
                        const float4 bi = *(const float4 *)&s_bias[p_par * G_BN + wr * WF + hf * 64 + f];
                        f32x4 o;
                        if constexpr (AFOLD) {
                            const float4 cc = *(const float4 *)&s_c[p_par * G_BN + wr * WF + hf * 64 + f];
                            o = {fmaf(a_rs, v[4 * g + 0], fmaf(a_nm, cc.x, bi.x)), fmaf(a_rs, v[4 * g + 1], fmaf(a_nm, cc.y, bi.y)),
                                 fmaf(a_rs, v[4 * g + 2], fmaf(a_nm, cc.z, bi.z)), fmaf(a_rs, v[4 * g + 3], fmaf(a_nm, cc.w, bi.w))};
                        } else o = {v[4 * g + 0] + bi.x, v[4 * g + 1] + bi.y, v[4 * g + 2] + bi.z, v[4 * g + 3] + bi.w};
                        const uint2 h = cvt_bf16x4(o);
                        uint16_t *p = (uint16_t *)(scr + f * 64) + vt_pos(r);
                        p[0] = (uint16_t)h.x; p[32] = (uint16_t)(h.x >> 16); p[64] = (uint16_t)h.y; p[96] = (uint16_t)(h.y >> 16);
                    }
                const int c0f = p_tn * G_BN - 2 * a.H + wr * WF + hf * 64;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int f = rl_f + 16 * i;
                    const uint4 seg = *(const uint4 *)(scr + f * 64 + rl_c * 16);
                    *(uint4 *)(a.vt + ((int64_t)b * a.H + c0f + f) * a.S + s0 + rl_c * 8) = seg;
                }
            }
        }
    };
    // one K-step (64 deep) out of ring slot `cur`.
    //   FIRST: accumulators start from 0.
    auto compute = [&](int cur, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const char *bufA = sW + cur * G_W_BYTES + a_off;
        const char *bufB = sX + cur * G_X_BYTES + b_off;
        uint4 av[2][MI], bv[2][2];
        auto load_frags = [&](int k2, uint4 (&a)[MI], uint4 (&b)[2]) {
            const int coff = (c0 ^ (k2 << 1)) << 4;
#pragma unroll
            for (int i = 0; i < 2; i++) b[i] = *(const uint4 *)(bufB + i * 4096 + coff);
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = *(const uint4 *)(bufA + i * 4096 + coff);
        };
        load_frags(0, av[0], bv[0]);
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            if (k2 < 3) load_frags(k2 + 1, av[(k2 + 1) & 1], bv[(k2 + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);   // keep the fragment prefetch above the MFMAs
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++) {
                    const uint4 ma = av[k2 & 1][mi], mb = bv[k2 & 1][ni];
                    if (FIRST && k2 == 0) {
                        f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[mi][ni] = mfma_bf16(ma, mb, z);
                    } else {
                        acc[mi][ni] = mfma_bf16(ma, mb, acc[mi][ni]);
                    }
                }
        }
    };
    // the finished tile (p_tn, p_tt, p_par set) through the wave's 4 KB scratch
    auto tile_out = [&](char *scr, int tn) {
        if constexpr (MODE == 1 || MODE == 3 || MODE == 4) rows_out(scr, a.out_bf16, a.ldo, tn * G_BN, 1.0f);
        else if constexpr (MODE == 6) rows_out_split(scr, tn * G_BN);
        else if constexpr (MODE == 0 && G_BN == 256) {     // H % 256 == 0: a tile is all Q, all K or all V
            if (tn * G_BN >= 2 * a.H) v_out(scr);
            else {
                const bool isq = tn * G_BN < a.H;
                rows_out(scr, isq ? a.q : a.k, a.H, isq ? tn * G_BN : tn * G_BN - a.H, isq ? a.qscale : 1.0f);
            }
        } else if constexpr (MODE == 0) {
            if (tn * G_BN + G_BN <= a.H) rows_out(scr, a.q, a.H, tn * G_BN, a.qscale);
            else if (tn * G_BN >= a.H && tn * G_BN + G_BN <= 2 * a.H) rows_out(scr, a.k, a.H, tn * G_BN - a.H, 1.0f);
            else if (tn * G_BN >= 2 * a.H) v_out(scr);
            else if constexpr (G_BN == 128) all_pieces();   // a tile straddling the Q/K/V boundaries (H not a multiple of 128;
                                                            // the wide tile is only launched when H % 256 == 0)
        } else rows_out_f32(scr);
    };
    using T_ = std::true_type; using F_ = std::false_type;

    // LZ: the per-feature vectors and per-token LayerNorm terms of tile (tn, tt) -> LDS by parity, 4-byte LDS-DMA like the biases
    // (issued in front of a tile's pieces: older than them, so the counted waits of the K-loop cover them)
    auto lz_stage = [&](int tn, int tt, int par) {
        if constexpr (LZ) {
            static_assert(G_BN == 256 && G_NW == 8, "lazy LayerNorm: wide tile");
            const int w4 = wave & 3;
            if constexpr (AFOLD) {
                if (wave >= 4) glds4(a.fold_c + tn * G_BN + w4 * 64 + lane, __builtin_amdgcn_readfirstlane(lds_addr(s_c) + (par * G_BN + w4 * 64) * 4));
                glds4(a.a_stats + ((int64_t)tt * G_BT + wave * 32) * 2 + lane, __builtin_amdgcn_readfirstlane(lds_addr(s_st) + (par * G_BT + wave * 32) * 8));
            } else {
                if (wave < 4) glds4(a.out_g + tn * G_BN + w4 * 64 + lane, __builtin_amdgcn_readfirstlane(lds_addr(s_g2) + (par * G_BN + w4 * 64) * 4));
                if (a.res_stats) {
                    if (wave >= 4) glds4(a.res_g + tn * G_BN + w4 * 64 + lane, __builtin_amdgcn_readfirstlane(lds_addr(s_g) + (par * G_BN + w4 * 64) * 4));
                    else glds4(a.res_b + tn * G_BN + w4 * 64 + lane, __builtin_amdgcn_readfirstlane(lds_addr(s_b) + (par * G_BN + w4 * 64) * 4));
                    glds4(a.res_stats + ((int64_t)tt * G_BT + wave * 32) * 2 + lane, __builtin_amdgcn_readfirstlane(lds_addr(s_st) + (par * G_BT + wave * 32) * 8));
                }
            }
        }
    };
    if constexpr (PH) {
        // ------------------------------------------------------------------------------------------------------------
        // Phased K-loop (scan.hip's, see there for the schedule and its hazards): a K-tile (64 deep) lives in LDS as four
        // 16 KiB half-tiles -- Wh0, Xh0, Xh1, Wh1 -- in one of two buffers and is computed in two phases of 16 MFMAs,
        //   X: read Wh0, Xh0, Xh1, quadrants (W0,X0) (W0,X1);   Y: read Wh1, quadrants (W1,X1) (W1,X0),
        // each  [fragment reads + LDS-DMA issue + lgkmcnt(0) + counted vmcnt]  barrier  [MFMA x 16]  barrier, with waves 4-7
        // (the SIMD partners of waves 0-3) one barrier behind: one wave of a SIMD owns the matrix pipe while its partner
        // reads and issues. Different from the scan at the tile end: the epilogue wants the eight waves together and 32 KB
        // of scratch, so the last K-tile of a tile does not issue the K-tile two ahead into its own buffer (that buffer IS
        // the scratch), the halves re-align, the tile leaves, and the skipped half-tiles are issued behind the epilogue.
        // ------------------------------------------------------------------------------------------------------------
        static_assert(G_BN == 256 && MI == 4, "phased loop: 256 x 256 tile");
        if (my_tiles == 0) return;
        // the first tile's biases
        {
            int tn0, tt0;
            tile_of(0, tn0, tt0);
            if (wave < G_BN / 64)
                glds4(a.bias + tn0 * G_BN + wave * 64 + lane, __builtin_amdgcn_readfirstlane(lds_addr(s_bias) + wave * 64 * 4));
            lz_stage(tn0, tt0, 0);
        }
        constexpr int HT = 16384, NHT = 4, S_A0 = 0, S_B0 = 1, S_B1 = 2, S_A1 = 3;
        const bool young = wave >= G_NW / 2;
        uint32_t voA[2][2], voB[2][2];
        int ra[4], rb[4];
        auto lane_consts = [&](int ln) {
            const int srow = ln >> 3, schunk = ln & 7, rr = ln & 31, kq = ln >> 5;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int lr = (wave * 2 + p) * 8 + srow;                                      // row of the half-tile
                    const int frow = ((lr >> 6) * MI + 2 * h + ((lr >> 5) & 1)) * 32 + (lr & 31);  // feature of the tile
                    const int trow = ((lr >> 5) * 2 + h) * 32 + (lr & 31);                         // token of the tile
                    const int gch = (schunk ^ ((lr >> 1) & 7)) << 4;
                    voA[h][p] = (uint32_t)frow * (uint32_t)ldk * 2u + gch;
                    voB[h][p] = (uint32_t)trow * (uint32_t)ldk * 2u + gch;
                }
            const int cc0 = kq ^ ((rr >> 1) & 7);
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
                const int coff = (cc0 ^ (k2 << 1)) << 4;
                ra[k2] = (wr * 64 + rr) * 128 + coff;
                rb[k2] = (wc * 32 + rr) * 128 + coff;
            }
        };
        lane_consts(lane);
        int c_kk = 0, c_ord = 0;                            // staging cursor (K-tile), clamped to this workgroup's last K-tile
        auto c_adv = [&]() {
            if (c_kk + 1 < KS) c_kk++;
            else if (c_ord + 1 < my_tiles) { c_kk = 0; c_ord++; }
        };
        auto sgpr64 = [](const char *ptr) {
            const uint64_t u = (uint64_t)ptr;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
            return ((uint64_t)hi << 32) | lo;
        };
        auto c_pa = [&]() {
            int tn, tt;
            tile_of(c_ord, tn, tt);
            return sgpr64((const char *)a.W + ((int64_t)tn * G_BN * ldk + (int64_t)kt_w(c_kk) * 64) * 2);
        };
        auto c_pb = [&]() {
            int tn, tt;
            tile_of(c_ord, tn, tt);
            return sgpr64((const char *)a.X + ((int64_t)tt * G_BT * ldk + (int64_t)kt_x(c_kk) * 64) * 2);
        };
        const uint32_t lds_w = lds_addr(lbase) + wave * 2048;
        auto issue = [&](uint64_t gbase, uint32_t o0, uint32_t o1, uint32_t dst) {
            if (a.flags & 2) return;
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "v"(o0), "v"(o1), "s"(gbase), "s"(dst) : "memory", "m0", "scc");
        };
        auto issue_slot = [&](auto stag, uint64_t pa, uint64_t pb, int buf) {
            constexpr int S = decltype(stag)::value;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_w + (buf * NHT + S) * HT);
            if constexpr (S == S_A0) issue(pa, voA[0][0], voA[0][1], dst);
            else if constexpr (S == S_B0) issue(pb, voB[0][0], voB[0][1], dst);
            else if constexpr (S == S_B1) issue(pb, voB[1][0], voB[1][1], dst);
            else issue(pa, voA[1][0], voA[1][1], dst);
        };
        using SA0 = std::integral_constant<int, S_A0>; using SB0 = std::integral_constant<int, S_B0>;
        using SB1 = std::integral_constant<int, S_B1>; using SA1 = std::integral_constant<int, S_A1>;
        uint64_t pa1, pb1, pa2, pb2;
        int cur = 0;
        auto kt_advance = [&]() { pa1 = pa2; pb1 = pb2; c_adv(); pa2 = c_pa(); pb2 = c_pb(); cur ^= 1; };
        {
            const uint64_t pa = c_pa(), pb = c_pb();
            issue_slot(SA0{}, pa, pb, 0); issue_slot(SB0{}, pa, pb, 0); issue_slot(SB1{}, pa, pb, 0); issue_slot(SA1{}, pa, pb, 0);
        }
        c_adv();
        pa1 = c_pa(); pb1 = c_pb();
        issue_slot(SA0{}, pa1, pb1, 1); issue_slot(SB0{}, pa1, pb1, 1); issue_slot(SB1{}, pa1, pb1, 1);
        c_adv();
        pa2 = c_pa(); pb2 = c_pb();
        wait_vm<2 * NHT>();
        __syncthreads();

        uint4 fa[2][4], fb0[4], fb1[4];
        [[maybe_unused]] int kt_rel = 0;                    // K-tile of the current tile (dbg: AK_GEMM_ABLATE=16)
        auto BAR = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // LAST: the tile's last K-tile -- phase Y does not issue into its own buffer (the epilogue's scratch)
        auto phase = [&](auto ytag, auto first_tag, auto last_tag) {
            constexpr bool Y = decltype(ytag)::value, FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
            if (young) BAR();
            const char *rbase = lbase + cur * (NHT * HT);
            if constexpr (!Y) {
#pragma unroll
                for (int k2 = 0; k2 < 4; k2++) fb0[k2] = *(const uint4 *)(rbase + S_B0 * HT + rb[k2]);
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int k2 = 0; k2 < 4; k2++) fa[m][k2] = *(const uint4 *)(rbase + S_A0 * HT + m * 32 * 128 + ra[k2]);
#pragma unroll
                for (int k2 = 0; k2 < 4; k2++) fb1[k2] = *(const uint4 *)(rbase + S_B1 * HT + rb[k2]);
                issue_slot(SA1{}, pa1, pb1, cur ^ 1);
            } else {
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int k2 = 0; k2 < 4; k2++) fa[m][k2] = *(const uint4 *)(rbase + S_A1 * HT + m * 32 * 128 + ra[k2]);
                if constexpr (!LAST) { issue_slot(SA0{}, pa2, pb2, cur); issue_slot(SB0{}, pa2, pb2, cur); issue_slot(SB1{}, pa2, pb2, cur); }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): this wave's reads are retired before its barrier
#if AK_DBG_KERNELS
            // AK_GEMM_ABLATE=16 (timing only, WRONG RESULTS possible): the first two K-tiles of a tile do not wait for the previous tile's
            // stores (vmcnt counts loads and stores in one in-order counter: every counted wait behind an epilogue drains its stores
            // first) -- how much of a launch is exposed store drain
            if ((a.flags & 16) && kt_rel < 2 && !(Y && LAST)) wait_vm<2 * NHT + 20>(); else
#endif
            if constexpr (Y && LAST) wait_vm<2>(); else wait_vm<2 * NHT>();
            BAR();
            __builtin_amdgcn_s_setprio(1);
            constexpr int MB = Y ? 2 : 0;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++)
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int nb = Y ? 1 - e : e;       // Y runs (W1,X1) then (W1,X0)
                        const uint4 bf = nb ? fb1[k2] : fb0[k2];
                        if (FIRST && k2 == 0) {
                            f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[MB + m][nb] = mfma_bf16(fa[m][k2], bf, z);
                        } else {
                            acc[MB + m][nb] = mfma_bf16(fa[m][k2], bf, acc[MB + m][nb]);
                        }
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!young) BAR();
        };
        for (int ord = 0; ord < my_tiles; ord++) {
            int tn, tt;
            tile_of(ord, tn, tt);
            const int par = ord & 1;
            {
                int ln = lane;
                asm volatile("" : "+v"(ln));                // opaque: the lane constants are recomputed per tile, not carried
                lane_consts(ln);                            // through the epilogue (carrying them where it costs no spill -- every
            }                                               // mode but the lazy MODE 4 -- measured: bge-base 13.19-13.26 vs 13.23-13.25 ms)
            kt_rel = 0;
            phase(F_{}, T_{}, F_{}); phase(T_{}, T_{}, F_{}); kt_advance();
            for (int kk = 1; kk < KS - 1; kk++) { kt_rel = kk; phase(F_{}, F_{}, F_{}); phase(T_{}, F_{}, F_{}); kt_advance(); }
            kt_rel = KS;
            phase(F_{}, F_{}, T_{}); phase(T_{}, F_{}, T_{});
            // the halves re-align (the older half waits out the younger half's last MFMAs); buffer `cur` is free: every read of it
            // is retired, nothing is in flight into it
            BAR();
            __syncthreads();
            if (!(a.flags & 1)) {
                p_tn = tn; p_tt = tt; p_par = par;
                tile_out(lbase + cur * (NHT * HT) + wave * (LZ && MODE == 4 ? 8192 : 4096), tn);
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int ni = 0; ni < 2; ni++) keep_live(acc[mi][ni]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __syncthreads();                                // every wave is done with the scratch
            int tn1, tt1;
            tile_of(ord + 1, tn1, tt1);
            if (ord + 1 < my_tiles && wave < G_BN / 64)     // the NEXT tile's biases (older than the pieces below: the counted waits hold)
                glds4(a.bias + tn1 * G_BN + wave * 64 + lane,
                      __builtin_amdgcn_readfirstlane(lds_addr(s_bias) + (((ord + 1) & 1) * G_BN + wave * 64) * 4));
            if (ord + 1 < my_tiles) lz_stage(tn1, tt1, (ord + 1) & 1);
            issue_slot(SA0{}, pa2, pb2, cur); issue_slot(SB0{}, pa2, pb2, cur); issue_slot(SB1{}, pa2, pb2, cur);
            kt_advance();
        }
        wait_vm<0>();
        return;
    }

#pragma unroll
    for (int i = 0; i < G_NSTAGE - 1; i++)
        if (issued < nsteps) stage_next();
    if (issued == 2) wait_vm<G_LOADS>(); else wait_vm<0>();
    __syncthreads();

    int cur = 0, step = 0;
    for (int ord = 0; ord < my_tiles; ord++) {
        int tn, tt;
        tile_of(ord, tn, tt);
        const int par = ord & 1;
        if (wave < G_BN / 64)   // this tile's 128 biases -> LDS (invisible to hipcc's vmcnt bookkeeping, like the ring)
            glds4(a.bias + tn * G_BN + wave * 64 + lane,
                  __builtin_amdgcn_readfirstlane(lds_addr(s_bias) + (par * G_BN + wave * 64) * 4));
        for (int kk = 0; kk < KS; kk++, step++) {
            if (issued < nsteps) { if (!(a.flags & 2)) stage_next(); else issued++; }
            if (kk == 0) compute(cur, T_{}); else compute(cur, F_{});
            if (issued >= step + 3) wait_vm<G_LOADS>(); else wait_vm<0>();
            __syncthreads();
            cur = (cur + 1 == G_NSTAGE) ? 0 : cur + 1;
        }
        if (a.flags & 1) {
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++) keep_live(acc[mi][ni]);
            continue;
        }
        p_tn = tn; p_tt = tt; p_par = par;
        const int free_slot = cur == 0 ? G_NSTAGE - 1 : cur - 1;      // consumed by the tile's last K-step
        tile_out(sX + free_slot * G_X_BYTES + wave * G_X_PW * 1024, tn);
    }
    wait_vm<0>();
}

template <int BN, bool PH = false>
static int launch_gemm_bn(int mode, const GemmArgs &a, hipStream_t st) {
    static std::atomic<bool> attr{false};      // (set twice by two first callers at worst: idempotent)
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<0, BN, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<BN>::LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<1, BN, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<BN>::LDS + (BN == 256 ? GELU_TAB_BYTES : 0)));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<2, BN, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<BN>::LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<3, BN, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<BN>::LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<4, BN, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<BN>::LDS));
        attr = true;
    }
    const int ntiles = (a.T / G_BT) * (a.N / BN);
    const int grid = ntiles < 256 ? ntiles : 256;
    switch (mode) {
        case 0: k_gemm<0, BN, PH><<<grid, G_THREADS, GCfg<BN>::LDS, st>>>(a); break;
        case 1: k_gemm<1, BN, PH><<<grid, G_THREADS, GCfg<BN>::LDS + (BN == 256 ? GELU_TAB_BYTES : 0), st>>>(a); break;
        case 2: k_gemm<2, BN, PH><<<grid, G_THREADS, GCfg<BN>::LDS, st>>>(a); break;
        case 4: k_gemm<4, BN, PH><<<grid, G_THREADS, GCfg<BN>::LDS, st>>>(a); break;
        default: k_gemm<3, BN, PH><<<grid, G_THREADS, GCfg<BN>::LDS, st>>>(a); break;
    }
    AK_HIP(hipGetLastError());
    return 0;
}

// ---- lazy LayerNorm ------------------------------------------------------------------------------------------------
static bool gemm_env_default() {      // the A/B switches that move a launch off the wide phased tile switch the lazy path off too
    static const bool ok = !env_get("AK_GEMM_BN") && !(env_get("AK_GEMM_PHASED") && atoi(env_get("AK_GEMM_PHASED")) == 0) && dbg_env_int("AK_GEMM_ABLATE", 0) == 0;
    return ok;
}
static int lazy_mode() {              // AK_ENC_LAZYLN: 0 = off, 2 = at every token count (tests: the suite's batches are small); default 1
    static const int m = env_get("AK_ENC_LAZYLN") ? atoi(env_get("AK_ENC_LAZYLN")) : 1;
    return m;
}
// From how many tiles of the narrowest GEMM (N = H) on: measured on bge-base (128 x 512 per forward, ms, narrow-tile path with its
// LayerNorm launches / lazy path on the wide tile): 8 192 tokens 2.70 / 2.90, 12 288 4.12 / 3.66, 16 384 4.64 / 4.14, 20 480 5.15 / 4.63 --
// half the CUs idle in the out-projection and it is still 10 % ahead (and rounds a sub-layer output once instead of three times).
constexpr int LAZY_MIN_TILES = 128;
bool gemm_lazy_supported(int64_t T, int H, int I) {
    // every GEMM of the layer on the 256 x 256 phased tile: N % 256 == 0, K >= 192
    return lazy_mode() && gemm_env_default() && T % G_BT == 0 && H % 256 == 0 && I % 256 == 0 && H >= 192 &&
           (lazy_mode() == 2 || (T / G_BT) * (H / 256) >= LAZY_MIN_TILES);
}
// AK_GEMM_FB (A/B): column tiles per feature block of the wide tile's order; default by shape (gemm_fb_default)
static int gemm_fb(int ntn) {
    static const int fb_env = env_get("AK_GEMM_FB") ? atoi(env_get("AK_GEMM_FB")) : -1;
    if (fb_env >= 0) return fb_env;
    // bge-base, same box, kernel-trace averages for fb = 0 / 6 / 4 / 3 / 2: FFN-up (12 column tiles) 340 / 335 / 325 / 326 / 336 us, QKV
    // (9) 244 / 247 / 248 / 244 / 252, forward 13.31 / 13.27 / 13.20 / 13.10 / 13.50 ms: four W tiles (1.6 MB) per block for the
    // widest output, one block otherwise (the fetch excess is served by the Infinity Cache: less of a cost than its size suggests)
    return ntn >= 12 ? 4 : 0;
}

int launch_gemm_lazy(int mode, const GemmArgs &a_in, hipStream_t st) {
    GemmArgs a = a_in;
    a.flags = 0;
    a.fb = gemm_fb(a.N / 256);
    if (!gemm_env_default() || a.T % G_BT || a.N % 256 || a.K % 64 || a.K < 192 || (lazy_mode() != 2 && (int64_t)(a.T / G_BT) * (a.N / 256) < LAZY_MIN_TILES) || (mode == 0 && a.H % 256))
        AK_FAIL(-1, "gemm (lazy LayerNorm): shape is not on the wide phased tile");
    if (a.nslot <= 0 || a.inv_h <= 0.f) AK_FAIL(-1, "gemm (lazy LayerNorm): nslot / inv_h not set");
    constexpr int LDS = GCfg<256>::LDS + 6 * 256 * 4 + 2 * G_BT * 8;      // + fold_c | gamma, beta, gamma-out by parity + (mean, 1 / std) of the tokens by parity
    static std::atomic<bool> attr{false};      // (set twice by two first callers at worst: idempotent)
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<0, 256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<1, 256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS + GELU_TAB_BYTES));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<4, 256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    const int ntiles = (a.T / G_BT) * (a.N / 256);
    const int grid = ntiles < 256 ? ntiles : 256;
    if (mode == 0) {
        if (!a.a_stats || !a.fold_c) AK_FAIL(-1, "gemm (lazy LayerNorm): MODE 0 needs a_stats and fold_c");
        k_gemm<0, 256, true, true><<<grid, G_THREADS, LDS, st>>>(a);
    } else if (mode == 1) {
        if (!a.a_stats || !a.fold_c) AK_FAIL(-1, "gemm (lazy LayerNorm): MODE 1 needs a_stats and fold_c");
        if (gelu_table_create()) return -10;
        a.gelu_tab = gelu_table_dev();
        k_gemm<1, 256, true, true><<<grid, G_THREADS, LDS + GELU_TAB_BYTES, st>>>(a);
    } else if (mode == 4) {
        if (!a.out_stats || !a.out_g || (a.res_stats && (!a.res_g || !a.res_b)))
            AK_FAIL(-1, "gemm (lazy LayerNorm): MODE 4 needs out_stats, out_g (and gamma / beta with res_stats)");
        k_gemm<4, 256, true, true><<<grid, G_THREADS, LDS, st>>>(a);
    } else AK_FAIL(-1, "gemm (lazy LayerNorm): mode must be 0, 1 or 4");
    AK_HIP(hipGetLastError());
    return 0;
}

// partial sums [nslot][T][2] of a MODE 4 launch -> (mean, 1 / std) per token [T][2]: what the next launches stage per tile
__global__ __launch_bounds__(256) void k_ln_finalize(const float *__restrict__ part, int nslot, int64_t T, float inv_h, float eps, float *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < nslot; i++) {
        const float2 p = *(const float2 *)(part + ((int64_t)i * T + t) * 2);
        s0 += p.x; s1 += p.y;
    }
    const float mu = s0 * inv_h, var = fmaxf(s1 * inv_h - mu * mu, 0.f);
    *(float2 *)(out + t * 2) = float2{mu, 1.0f / sqrtf(var + eps)};
}
int launch_ln_finalize(const float *part, int nslot, int64_t T, float inv_h, float eps, float *out, hipStream_t st) {
    k_ln_finalize<<<(unsigned)((T + 255) / 256), 256, 0, st>>>(part, nslot, T, inv_h, eps, out);
    AK_HIP(hipGetLastError());
    return 0;
}

// c[n] = sum_k gamma[k] W[n][k], bf[n] = bias[n] + sum_k beta[k] W[n][k]   (W: [N][K] bf16, summed in fp32)
__global__ __launch_bounds__(256) void k_fold_ln(const uint16_t *__restrict__ W, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                 const float *__restrict__ bias, int K, float *__restrict__ c, float *__restrict__ bf) {
    __shared__ float s_c[4], s_d[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float pc = 0.f, pd = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float w = bf16_to_f32(W[(int64_t)n * K + k]);
        pc = fmaf(gamma[k], w, pc);
        pd = fmaf(beta[k], w, pd);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { pc += __shfl_xor(pc, off); pd += __shfl_xor(pd, off); }
    if ((tid & 63) == 0) { s_c[tid >> 6] = pc; s_d[tid >> 6] = pd; }
    __syncthreads();
    if (tid == 0) {
        c[n] = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
        bf[n] = bias[n] + ((s_d[0] + s_d[1]) + (s_d[2] + s_d[3]));
    }
}
int launch_fold_ln(const uint16_t *W, const float *gamma, const float *beta, const float *bias, int N, int K, float *c, float *bf, hipStream_t st) {
    k_fold_ln<<<N, 256, 0, st>>>(W, gamma, beta, bias, K, c, bf);
    AK_HIP(hipGetLastError());
    return 0;
}

// The split-bf16 parity mode on this file's tiles (MODE 5 / 6 above): a.X [T][2 K'] and a.W [N][2 K'] bf16 rows [hi | lo], a.K = 3 K'
// (K' % 64 == 0), bias float32; mode 5 -> a.out_f32 [T][N], mode 6 -> a.out_bf16 [T][2 N] (a.ldo = 2 N) = [hi | lo] of gelu(.)
bool gemm_x3w_supported(int64_t T, int N, int K1) { return T % G_BT == 0 && N % 128 == 0 && K1 % 64 == 0 && K1 >= 64; }
int launch_gemm_x3w(int mode, const GemmArgs &a_in, hipStream_t st) {
    GemmArgs a = a_in;
    a.flags = 0;
    if (a.T % G_BT || a.N % 128 || a.K % 192) AK_FAIL(-1, "gemm (split bf16): shape must be T%256==0, N%128==0, K'%64==0");
    if (mode != 5 && mode != 6) AK_FAIL(-1, "gemm (split bf16): mode must be 5 or 6");
    if (a.nvalid <= 0 || a.nvalid > a.N) a.nvalid = a.N;
    if (mode == 6) {
        if (phi_table_create()) return -10;
        a.phi_tab = g_phi_tab;
    }
    static const int force_bn = env_get("AK_GEMM_BN") ? atoi(env_get("AK_GEMM_BN")) : 0;      // A/B: 128 or 256
    bool wide = a.N % 256 == 0 && (int64_t)(a.T / G_BT) * (a.N / 256) >= 256;
    if (force_bn == 128) wide = false;
    if (force_bn == 256 && a.N % 256 == 0) wide = true;
    static std::atomic<bool> attr{false};
    if (!attr) {
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<5, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<256>::LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<6, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<256>::LDS + PHI_BYTES));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<5, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<128>::LDS));
        AK_HIP(hipFuncSetAttribute((const void *)k_gemm<6, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<128>::LDS + PHI_BYTES));
        attr = true;
    }
    const int bn = wide ? 256 : 128;
    const int ntiles = (a.T / G_BT) * (a.N / bn);
    const int grid = ntiles < 256 ? ntiles : 256;
    if (wide) {
        a.fb = gemm_fb(a.N / 256);
        if (mode == 5) k_gemm<5, 256, true><<<grid, G_THREADS, GCfg<256>::LDS, st>>>(a);
        else k_gemm<6, 256, true><<<grid, G_THREADS, GCfg<256>::LDS + PHI_BYTES, st>>>(a);
    } else {
        if (mode == 5) k_gemm<5, 128><<<grid, G_THREADS, GCfg<128>::LDS, st>>>(a);
        else k_gemm<6, 128><<<grid, G_THREADS, GCfg<128>::LDS + PHI_BYTES, st>>>(a);
    }
    AK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm(int mode, const GemmArgs &a_in, hipStream_t st) {
    GemmArgs a = a_in;
    static const int ablate = dbg_env_int("AK_GEMM_ABLATE", 0);
    static const int force_bn = env_get("AK_GEMM_BN") ? atoi(env_get("AK_GEMM_BN")) : 0;      // A/B: 128 or 256
    a.flags = ablate;
    if (mode == 1) {
        if (gelu_table_create()) return -10;
        a.gelu_tab = gelu_table_dev();
    }
    if (a.T % G_BT || a.N % 128 || a.K % 64) AK_FAIL(-1, "gemm: shape must be T%256==0, N%128==0, K%64==0");
    // the wide tile needs N % 256 == 0 (for the QKV split: H % 256 == 0 too, so no tile straddles Q/K/V) and at least
    // one tile per CU
    bool wide = a.N % 256 == 0 && (int64_t)(a.T / G_BT) * (a.N / 256) >= 256 && (mode != 0 || a.H % 256 == 0);
    if (force_bn == 128) wide = false;
    if (force_bn == 256 && a.N % 256 == 0 && (mode != 0 || a.H % 256 == 0)) wide = true;
    static const int phased = env_get("AK_GEMM_PHASED") ? atoi(env_get("AK_GEMM_PHASED")) : 1;      // A/B: 0 = the in-step loop on the wide tile
    if (wide) a.fb = gemm_fb(a.N / 256);
    if (wide && phased && a.K >= 192) return launch_gemm_bn<256, true>(mode, a, st);
    return wide ? launch_gemm_bn<256>(mode, a, st) : launch_gemm_bn<128>(mode, a, st);
}

}  // namespace ak
