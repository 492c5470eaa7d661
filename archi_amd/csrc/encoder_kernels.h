// encoder_kernels.h -- argument blocks and launchers shared by encoder.hip and its kernels
// (gemm.hip, gemm_ln.hip, attention.hip). One definition, so the caller and the kernel agree on the layout.
#pragma once
#include "common.h"

namespace ak {

struct GemmArgs {
    const uint16_t *X; const uint16_t *W; const float *bias;
    int T, N, K;
    uint16_t *out_bf16; int ldo;
    float *out_f32; const float *res_f32;
    const uint16_t *res16;   // MODE 4: bf16 residual rows (leading dimension ldo) added to the bf16 output: the rows the LayerNorm then reads
    uint16_t *q, *k, *vt; int H, S; float qscale;
    int flags;   // AK_GEMM_ABLATE (measurement only): 1 skip the epilogue, 2 skip the staging loads
    int fb = 0;  // wide tile: column tiles per feature block of the XCD-aware tile order (0 = all of them; gemm.hip "Feature blocks")
    const uint16_t *gelu_tab = nullptr;   // set by launch_gemm (MODE 1): the bf16 GELU table of gelu_table.h
    int nvalid = 0;                       // MODE 5: columns >= nvalid (a width padded to a multiple of 256) are not stored; 0 = all of N
    const float *phi_tab = nullptr;       // set by launch_gemm_x3w (MODE 6): cubic pieces of the normal CDF (gemm.hip, GELU OF THE SPLIT MODE)
    // LAZY LayerNorm (launch_gemm_lazy, gemm.hip): the rows between the sub-layers travel as r~ = gamma (.) r (bf16; r the
    // un-normalised sub-layer output, gamma of the LayerNorm that follows) with r's per-token (mean, 1 / std) [T][2] beside them
    // (launch_ln_finalize over the partial sums [nslot][T][2] = (sum, sum of squares) per 128-feature slice that the producing
    // launch writes); whoever consumes such rows finishes the LayerNorm on the way:
    //   a_stats   (MODE 0 / 1): X holds r~. fold_c[n] = sum_k gamma[k] W[n][k], bias[n] = b[n] + sum_k beta[k] W[n][k], W unchanged:
    //              out = rstd (acc - mu fold_c) + bias  ==  LN(r) W^T + b
    //   res_stats (MODE 4): res16 holds r~; the residual added is rstd (r~ - mu res_g) + res_b. NULL: res16 is added as it is
    //   out_stats, out_g (MODE 4): partial sums of the row this launch computes; it is stored scaled by out_g
    const float *fold_c = nullptr, *a_stats = nullptr, *res_stats = nullptr, *res_g = nullptr, *res_b = nullptr, *out_g = nullptr;
    float *out_stats = nullptr;
    int nslot = 0; float inv_h = 0.f, eps = 0.f;
};
struct AttnArgs {
    const uint16_t *q, *k, *vt;
    const int *mask;
    uint16_t *ctx;
    int B, S, H, heads;
    const float *maskf;          // additive key mask (0 / -inf) [B][S] and, per sequence, the bitmap of 32-key blocks that
    const uint32_t *blkmask;     // hold a real key (launch_attn_prepare, once per forward pass); NULL: the unstreamed kernel
    // layout of q and k inside a sequence's S * H elements: row (token) stride and head stride in elements.
    // 0 / 0 = token-major [S][H] (ld = H, hs = H / heads); head-major [heads][S][hd] (k_qkv384): ld = hd, hs = S * hd
    int qk_ld, qk_hs;
};
struct GemmLnArgs {
    const uint16_t *X; const uint16_t *W; const float *bias; const float *gamma; const float *beta;
    float *x32; uint16_t *x16;     // residual in / LayerNorm out (fp32, in place) and its bf16 copy; x32 == NULL: the
                                   // residual stream is x16 alone (read and rewritten in place)
    int T, K; float eps;
    long long *dbg;   // AK_GEMMLN_DBG (measurement only): per-wave cycles {K-loop, epilogue}, else NULL
};
// fused feed-forward block, hidden size 384, bf16 residual stream (ffn.hip)
struct FfnArgs {
    uint16_t *x16;              // [T][384] bf16: input, residual and output (in place)
    const uint16_t *wf;         // both weight matrices in fragment order (ffn_relayout)
    const float *b1, *b2, *gamma, *beta;
    // fused attention output projection (8-wave kernel only): ctx != NULL -> x16 <- LN2(z + FFN(z)), z = LN1(x16 + ctx . Wo^T + bo);
    // wof = Wo in fragment order, 6 x 48 KB directly in front of wf (one array of ring blocks)
    const uint16_t *ctx; const uint16_t *wof; const float *bo, *gamma1, *beta1;
    int T, I; float eps;
    long long *dbg;             // AK_FFN_DBG (measurement only): per-wave cycles {wait+barrier, stage, phase A, GELU, phase B, epilogue}
    const uint16_t *gelu_tab = nullptr;   // set by launch_ffn384: the 8192-entry bf16 GELU table (ffn.hip, GELU BY TABLE)
    int ablate = 0;             // AK_FFN_ABLATE (instrumented instantiation only; WRONG RESULTS): 1 no ring DMA in the chunk loop, 2 no GELU,
                                // 4 no phase-A MFMAs, 8 no phase-B MFMAs, 16 one fragment read per phase
};
// QKV projection, hidden size 384 (ffn.hip): q (pre-scaled), k [Tpad][384] and v transposed [B][384][S] (vt_pos order)
struct QkvArgs {
    const uint16_t *x16;        // [Tpad][384] bf16
    const uint16_t *w;          // qkv384_relayout output
    const float *bias;          // set by launch_qkv384 (behind the weight blocks)
    uint16_t *q, *k, *vt;
    int Tpad, T, S; float qscale;   // T: real tokens (rows past it have no V^T slot)
    int dbg;                        // AK_QKV_DBG (measurement only)
    int head_major;                 // q / k as [B][heads][S][32] (a head's rows contiguous: whole cache lines for the attention
                                    // kernel's staging) instead of [T][384]; rows past T are then not written
};
size_t qkv384_weight_bytes();
bool qkv384_supported(int H, int64_t T, int S);
int qkv384_relayout(const uint16_t *wqkv, const float *bqkv, uint16_t *wbuf, hipStream_t st);
int launch_qkv384(const QkvArgs &a, hipStream_t st);
bool ffn_fused_supported(int H, int I, int64_t T);
size_t ffn_weight_bytes(int I);          // [Wo fragments (6 x 48 KB) | W1 / W2 chunks]
size_t ffn_wo_bytes();
// wbuf: ffn_weight_bytes(I) bytes; returns in *wf_out the pointer to the feed-forward chunks (wbuf + ffn_wo_bytes())
int ffn_relayout(const uint16_t *wo, const uint16_t *w1, const uint16_t *w2, int I, uint16_t *wbuf, const uint16_t **wf_out, hipStream_t st);
bool ffn_fuses_attention_out();
int launch_ffn384(const FfnArgs &a, hipStream_t st);
int launch_gemm(int mode, const GemmArgs &a, hipStream_t st);
// the lazy-LayerNorm variants (modes 0, 1, 4) of the wide phased tile; gemm_lazy_supported: every GEMM of a layer takes that tile
bool gemm_lazy_supported(int64_t T, int H, int I);
int launch_gemm_lazy(int mode, const GemmArgs &a, hipStream_t st);
int launch_ln_finalize(const float *part, int nslot, int64_t T, float inv_h, float eps, float *out, hipStream_t st);
// c[n] = sum_k gamma[k] W[n][k], bf[n] = bias[n] + sum_k beta[k] W[n][k]   (W: [N][K] bf16)
int launch_fold_ln(const uint16_t *W, const float *gamma, const float *beta, const float *bias, int N, int K, float *c, float *bf, hipStream_t st);
int launch_attn(const AttnArgs &a, hipStream_t st);
int launch_attn_prepare(const int *mask, int B, int S, float *maskf, uint32_t *blkmask, hipStream_t st);
// position of key s inside its V^T row: the keys of a group of 16 are stored [0-3, 8-11, 4-7, 12-15] (attention.hip)
__host__ __device__ inline int vt_pos(int s) { return (s & ~12) | ((s & 4) << 1) | ((s & 8) >> 1); }
bool gemm_ln_supported(int H, int64_t T, int K);
int launch_gemm_ln(const GemmLnArgs &a, hipStream_t st);
int launch_gemm_ln_x3(const GemmLnArgs &a, hipStream_t st);   // split-bf16 operands ([hi | lo] rows, K = 3 K'), float32 residual, LayerNorm output as float32 + [hi | lo] rows
// float32 parity mode on v_mfma_f32_32x32x2_f32 (encoder_f32.hip): Y[T][ldc] (+ col0) = X W^T + bias (epi 0) | gelu (1) | + R (2)
bool f32_mfma_supported(int H, int I, int heads);
int launch_gemm_f32(int epi, const float *X, const float *W, const float *bias, const float *R, int T, int N, int K, float *Y, int ldc,
                    int col0, hipStream_t st);
int launch_attn_f32(const float *qkv, const int *mask, int B, int S, int H, int heads, float *ctx, hipStream_t st);
// split-bf16 parity mode (precision 2): W as bf16 hi + lo (split_hilo, once), X float32 split on its way into LDS; three bf16
// MFMAs per product into one float32 accumulator (encoder_f32.hip k3_gemm); epilogues as launch_gemm_f32
int split_hilo(const float *w, int64_t n, uint16_t *hi, uint16_t *lo, hipStream_t st);
int launch_attn_x3(const float *qkv, const int *mask, int B, int S, int H, int heads, float *ctx, hipStream_t st);   // k3_attn: both products as three bf16 MFMAs
int launch_gemm_x3(int epi, const float *X, const uint16_t *Whi, const uint16_t *Wlo, const float *bias, const float *R, int T, int N, int K,
                   float *Y, int ldc, int col0, hipStream_t st);
// ... and on gemm.hip's LDS-DMA tiles for batches of whole 256-token tiles (MODE 5 / 6 there): the activations travel as bf16
// [hi | lo] rows ([T][2 K], split by whoever produces them), the matrices likewise ([N][2 K], split_rows once at create)
bool gemm_x3w_supported(int64_t T, int N, int K1);
int launch_gemm_x3w(int mode, const GemmArgs &a, hipStream_t st);
int split_rows(const float *x, int64_t rows, int K, uint16_t *out, hipStream_t st);                    // out[r] = [hi(K) | lo(K)]
// out = LayerNorm(y + r) * g + b (float32, in place over r allowed; y rows of ldy floats) and its [hi | lo] rows; r == NULL: no residual
int launch_add_ln_split(const float *y, int ldy, const float *r, int64_t T, int H, const float *g, const float *b, float eps, float *out, uint16_t *out2,
                        hipStream_t st);
int launch_embed_split(const int *ids, int64_t T, int S, int H, int vocab, const float *word, const float *pos, const float *type, const float *g,
                       const float *b, float eps, float *out, uint16_t *out2, hipStream_t st);       // embeddings + LayerNorm: float32 rows and [hi | lo] rows
// k3_attn over qkv rows of ldq floats (q | k | v in the first 3 H), context as [hi | lo] rows
int launch_attn_x3_split(const float *qkv, int ldq, const int *mask, int B, int S, int H, int heads, uint16_t *ctx2, hipStream_t st);
// the whole forward pass of <= 64 token rows in ONE launch confined to one XCD (query_forward.hip)
struct QfCtlSlot { unsigned long long arrived_tickets; unsigned int target; unsigned int count; unsigned int pad[4]; };   // 32 bytes
struct QfCtl { QfCtlSlot slot[64]; };                     // one slot per launch, by launch number mod 64; zeroed 32 launches ahead
struct QfLayer {
    const uint16_t *wqkv; const float *bqkv; const uint16_t *wo; const float *bo, *ln1g, *ln1b;
    const uint16_t *w1; const float *b1; const uint16_t *w2; const float *b2, *ln2g, *ln2b;
};
struct QfArgs {
    const int *ids_in; int ld_ids; const int *lens; int lens_stride;      // right-padded rows + lengths (lens != NULL) ...
    const int *ids, *mask;                                                 // ... or ids / mask [B][S] as they are (lens == NULL)
    int *oids, *omask;                                                     // workspace of the lens form
    int B, S, T, t32, H, I, heads, L, vocab; float eps, qscale;
    const uint16_t *word, *pos, *type; const float *eg, *eb;
    float *x32 /* NULL: bf16 residual stream */, *y32; uint16_t *x16, *q, *k, *vt, *ctx, *f; float *maskf; uint32_t *blkmask;
    const QfLayer *layers;                                                 // device array [L]
    int pooling, normalise; float *out;
    QfCtl *ctl; unsigned *fail; unsigned epoch;
    int dbg_skip;                                                          // dbg library only (AK_QF_SKIP): 1 no phase bodies, 2 no barriers
};
bool query_forward_supported(int H, int I, int heads, int64_t T, int S);
int launch_query_forward(const QfArgs &a, hipStream_t st);
bool gemm_skinny_supported(int N, int K);
int launch_gemm_skinny(const uint16_t *X, const uint16_t *W, const float *bias, int rows, int N, int K, float *out_f32,
                       uint16_t *out_bf16, int ldo, hipStream_t st);
int launch_gemm_skinny_qkv(const uint16_t *X, const uint16_t *W, const float *bias, int rows, int H, int K, uint16_t *q,
                           uint16_t *k, uint16_t *vt, int S, int T, float qscale, hipStream_t st);

}  // namespace ak
