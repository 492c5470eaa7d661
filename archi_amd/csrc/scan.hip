// scan.hip -- the hot kernel: query x corpus similarity as a 16-bit MFMA GEMM
// with LDS-staged tiles and a fused top-k' candidate filter, followed by an
// exact re-rank in the reference arithmetic (exact.hip) and a certificate.
//
// Replaces, for the common case, the sequential scan + top-N heapsort that one
// Postgres backend runs for
//   SELECT ... c.embedding <op> %s::vector AS distance ... ORDER BY distance LIMIT k
// (/root/reference/src/data_manager/vectorstore/postgres_vectorstore.py:317-332).
//
// Structure (gfx950, wave64):
//   grid   = nslices x nqg persistent workgroups of 256 threads (4 waves, 2x2)
//            block b -> XCD b%8; the nqg query-groups of one corpus slice are
//            given consecutive slots on ONE XCD so the slice is read from HBM
//            once and re-read from that XCD's L2.
//   tile   = BM=128 corpus rows x BN=128 queries, K-step 64 (128 B per row),
//            double-buffered in LDS, filled by global_load_lds (16 B/lane, the
//            1 KiB wave piece is 8 rows x 128 B -> full-line coalesced reads).
//            LDS rows are XOR-swizzled on the SOURCE address (chunk ^= (row>>1)&7)
//            so the ds_read_b128 fragment reads are bank-conflict free.
//   mfma   = v_mfma_f32_32x32x16_{bf16,f16}; A = corpus rows, B = queries, so a
//            lane owns ONE query column (lane&31) and 16 corpus rows per tile:
//            the top-k reduction axis is lane-local.
//   top-k' = per (block, query) an append buffer in global memory (L2 resident)
//            with an LDS counter and an LDS threshold: score >= thr -> append.
//            When a counter nears capacity one wave compacts that buffer to its
//            k' best (bitwise binary search over 64-bit keys with ballots) and
//            raises the threshold. Expected appends per (block,query):
//            ~ k' * ln(rows/k'), i.e. a few compactions per scan.
//   output = [nq][nslices][k'] keys (score key << 32 | row slot)
// then: select top-k' per query -> re-rank k' candidates in reference arithmetic
// -> select top-k -> certificate (the k-th exact score beats every non-candidate's
// upper bound), else the caller falls back to the exact path.
#include "index.h"

namespace ak {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;        // corpus rows per tile
constexpr int BN = 128;        // queries per block
constexpr int BK = 64;         // k per LDS stage (128 B per row)
constexpr int THREADS = 256;
constexpr int CAP = 512;       // append-buffer entries per (block, query)
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB

// LDS-DMA, 16 B per lane: LDS[lds_wave_base + lane*16] <- *g. Issued from inline
// asm so hipcc does not serialise it against the ds_reads of the OTHER buffer
// (it cannot prove they do not alias and would wait vmcnt(0) before every
// fragment read). Completion is waited for by hand: wait_glds() before the step
// barrier. M0 carries the LDS base and is saved/restored inside the statement.
__device__ inline void glds16(const void *g, uint32_t lds_wave_base) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_wave_base)
        : "memory");
}
__device__ inline void wait_glds() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ inline uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

template <bool IS_BF16>
__device__ inline f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
    if constexpr (IS_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ inline uint64_t ld_sc1(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2-served, never stale
}

// One wave: reduce the append buffer of one query to its kp best entries.
// Returns the new threshold score key (high half of the kp-th best key).
__device__ inline void compact_wave(uint64_t *buf, int m, int kp, int lane, float *thr_out, int *cnt_out,
                                    uint64_t *final_out /* nullable: write survivors here instead */) {
    constexpr int SLOTS = CAP / 64;
    uint64_t key[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; j++) {
        int idx = j * 64 + lane;
        key[j] = idx < m ? ld_sc1(buf + idx) : KEY_INVALID;
    }
    uint64_t T = KEY_INVALID;
    if (m > kp) {
        // T = kp-th smallest key = min value with count(key <= T) >= kp
        T = 0;
        for (int bit = 63; bit >= 0; bit--) {
            uint64_t test = T | ((1ull << bit) - 1ull);
            int c = 0;
#pragma unroll
            for (int j = 0; j < SLOTS; j++) c += __popcll(__ballot(key[j] <= test));
            if (c < kp) T |= (1ull << bit);
        }
    }
    uint64_t *dst = final_out ? final_out : buf;
    int run = 0;
#pragma unroll
    for (int j = 0; j < SLOTS; j++) {
        bool keep = key[j] != KEY_INVALID && key[j] <= T;
        uint64_t mask = __ballot(keep);
        int pos = run + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep) dst[pos] = key[j];
        run += __popcll(mask);
    }
    if (final_out) {
        for (int i = run + lane; i < kp; i += 64) final_out[i] = KEY_INVALID;
    } else if (lane == 0) {
        *cnt_out = run;
        if (m > kp) *thr_out = key_score((uint32_t)(T >> 32));
    }
}

// rows: [n][D] 16-bit; qs: [nq_pad][D] 16-bit (queries rounded to the scan dtype)
template <bool IS_BF16>
__global__ __launch_bounds__(THREADS, 2) void k_scan(const uint16_t *__restrict__ rows, const float *__restrict__ ea,
                                                     const float *__restrict__ eb, const uint8_t *__restrict__ filter,
                                                     int64_t n, int D, const uint16_t *__restrict__ qs, int nq,
                                                     int nslices, int nqg, int kp, uint64_t *__restrict__ cand,
                                                     uint64_t *__restrict__ out_c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;                          // [2][BM][128 B]
    char *sB = smem + 2 * TILE_BYTES;         // [2][BN][128 B]
    float *s_ea = (float *)(smem + 4 * TILE_BYTES);
    float *s_eb = s_ea + BM;
    float *s_thr = s_eb + BM;
    int *s_cnt = (int *)(s_thr + BN);
    int *s_need = s_cnt + BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware slot mapping: xcd = b % 8; consecutive slots of one XCD walk the
    // query groups of one slice first.
    const int b = blockIdx.x;
    int slice, qg;
    {
        int xcd = b & 7, j = b >> 3;
        int nb8 = (nslices * nqg) >> 3;  // slots per XCD when divisible by 8
        if (((nslices * nqg) & 7) == 0 && (nslices & 7) == 0) {
            (void)nb8;
            qg = j % nqg;
            slice = xcd + 8 * (j / nqg);
        } else {
            qg = b % nqg;
            slice = b / nqg;
        }
    }
    const int64_t ntiles = (n + BM - 1) / BM;
    const int64_t t0 = ntiles * slice / nslices, t1 = ntiles * (slice + 1) / nslices;
    const int KS = D / BK;
    const int q0 = qg * BN;

    if (tid < BN) {
        s_thr[tid] = (q0 + tid < nq) ? -3.4028234663852886e38f : __builtin_inff();  // padded queries never append
        s_cnt[tid] = 0;
    }
    if (tid == 0) *s_need = 0;

    uint64_t *my_cand = cand + ((size_t)blockIdx.x * BN) * CAP;

    // per-lane fragment addressing (see header): row r = lane&31, k-half kh = lane>>5
    const int r = lane & 31, kh = lane >> 5;
    const int sw = (r >> 1) & 7;  // rows differ by multiples of 32 between fragments -> same swizzle
    const int c0 = kh ^ sw;
    const int a_off = (wr * 64 + r) * 128;  // + mi*32*128
    const int b_off = (wc * 64 + r) * 128;  // + ni*32*128

    // staging addressing: wave w stages pieces w*4 .. w*4+3 of each operand tile
    const int st_row = lane >> 3;                        // row within the 8-row piece
    const int st_chunk = lane & 7;                       // LDS chunk position
    const int64_t total_steps = (t1 - t0) * KS;
    const uint32_t ldsA = lds_addr(sA), ldsB = lds_addr(sB);

    auto stage = [&](int64_t step, int buf) {
        int64_t tile = t0 + step / KS;
        int kk = (int)(step % KS);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            int piece = wave * 4 + p;
            int row = piece * 8 + st_row;
            int gchunk = st_chunk ^ ((row >> 1) & 7);
            int64_t grow = tile * BM + row;
            if (grow >= n) grow = n - 1;
            const char *ga = (const char *)rows + (grow * D + kk * BK) * 2 + gchunk * 16;
            glds16(ga, __builtin_amdgcn_readfirstlane(ldsA + buf * TILE_BYTES + piece * 1024));
            const char *gb = (const char *)qs + ((int64_t)(q0 + row) * D + kk * BK) * 2 + gchunk * 16;
            glds16(gb, __builtin_amdgcn_readfirstlane(ldsB + buf * TILE_BYTES + piece * 1024));
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ni][e] = 0.f;

    if (total_steps > 0) stage(0, 0);
    wait_glds();
    __syncthreads();

    for (int64_t step = 0; step < total_steps; step++) {
        const int cur = (int)(step & 1);
        const int kk = (int)(step % KS);
        const int64_t tile = t0 + step / KS;
        if (step + 1 < total_steps) stage(step + 1, cur ^ 1);
        if (kk == 0 && tid < BM) {
            int64_t grow = tile * BM + tid;
            bool ok = grow < n && (!filter || filter[grow]);
            s_ea[tid] = ok ? ea[grow] : 0.f;
            s_eb[tid] = ok ? eb[grow] : -__builtin_inff();
        }
        const char *bufA = sA + cur * TILE_BYTES;
        const char *bufB = sB + cur * TILE_BYTES;
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const int coff = (c0 ^ (k2 << 1)) << 4;
            uint4 a0 = *(const uint4 *)(bufA + a_off + coff);
            uint4 a1 = *(const uint4 *)(bufA + a_off + 32 * 128 + coff);
            uint4 b0 = *(const uint4 *)(bufB + b_off + coff);
            uint4 b1 = *(const uint4 *)(bufB + b_off + 32 * 128 + coff);
            acc[0][0] = mfma32<IS_BF16>(a0, b0, acc[0][0]);
            acc[0][1] = mfma32<IS_BF16>(a0, b1, acc[0][1]);
            acc[1][0] = mfma32<IS_BF16>(a1, b0, acc[1][0]);
            acc[1][1] = mfma32<IS_BF16>(a1, b1, acc[1][1]);
        }
        if (kk == KS - 1) {
            // ---- fused epilogue: score = fma(dot, ea[row], eb[row]); append if >= threshold
            if (KS == 1) __syncthreads();  // s_ea written in this same step
            const uint32_t tile_row0 = (uint32_t)(tile * BM);
#pragma unroll
            for (int ni = 0; ni < 2; ni++) {
                const int qcol = wc * 64 + ni * 32 + r;
                const float thr = s_thr[qcol];
#pragma unroll
                for (int mi = 0; mi < 2; mi++) {
                    float sc[16];
                    float mx = -__builtin_inff();
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        int lrow = wr * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                        sc[e] = fmaf(acc[mi][ni][e], s_ea[lrow], s_eb[lrow]);
                        mx = fmaxf(mx, sc[e]);  // NaN-ignoring
                        acc[mi][ni][e] = 0.f;
                    }
                    if (mx >= thr) {
#pragma unroll
                        for (int e = 0; e < 16; e++) {
                            if (sc[e] >= thr) {
                                int lrow = wr * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                                int pos = atomicAdd(&s_cnt[qcol], 1);
                                my_cand[(size_t)qcol * CAP + pos] =
                                    ((uint64_t)score_key(sc[e]) << 32) | (uint64_t)(tile_row0 + lrow);
                                if (pos >= CAP - BM - 1) *s_need = 1;
                            }
                        }
                    }
                }
            }
            __syncthreads();  // appends visible (vmcnt(0) + barrier), s_need settled
            if (*s_need) {
                for (int q = wave; q < BN; q += 4) {
                    int m = s_cnt[q];
                    if (m > CAP - BM) compact_wave(my_cand + (size_t)q * CAP, m, kp, lane, &s_thr[q], &s_cnt[q], nullptr);
                }
                __syncthreads();
                if (tid == 0) *s_need = 0;
            }
        }
        wait_glds();      // this wave's pieces of the next buffer have landed
        __syncthreads();  // everyone's pieces landed, and the current buffer is free to overwrite
    }

    // final: every query's buffer -> its kp best -> out_c[q][slice][0..kp)
    __syncthreads();
    for (int q = wave; q < BN; q += 4) {
        if (q0 + q >= nq) continue;
        uint64_t *dst = out_c + ((size_t)(q0 + q) * nslices + slice) * kp;
        compact_wave(my_cand + (size_t)q * CAP, s_cnt[q], kp, lane, nullptr, nullptr, dst);
    }
}

// ---------------------------------------------------------------------------
// query preparation for the scan: round to the scan dtype, zero-pad to BN, and
// the per-query terms of the certificate.
//   prep[q] = { eps (score units), scale a, offset b } with
//   s~_units = a * s~' + b  where s~' is the scan's score.
// ---------------------------------------------------------------------------
struct QPrep { double eps, a, b; };

template <bool IS_BF16>
__global__ void k_query_prep(const float *__restrict__ q, const float *__restrict__ nb, int nq, int nq_pad, int D,
                             int metric, float max_na, int corpus_f32_shadow, uint16_t *__restrict__ qs,
                             QPrep *__restrict__ prep) {
    int qi = blockIdx.x;
    int lane = threadIdx.x;  // 64 threads
    uint16_t *dst = qs + (int64_t)qi * D;
    if (qi >= nq) {
        for (int i = lane; i < D; i += 64) dst[i] = 0;
        return;
    }
    const float *v = q + (int64_t)qi * D;
    double err2 = 0.0;
    for (int i = lane; i < D; i += 64) {
        float x = v[i];
        uint16_t h = IS_BF16 ? f32_to_bf16(x) : f32_to_f16(x);
        float y = IS_BF16 ? bf16_to_f32(h) : f16_to_f32(h);
        dst[i] = h;
        double d = (double)x - (double)y;
        err2 += d * d;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) err2 += __shfl_xor(err2, off);
    if (lane == 0) {
        double nq2 = (double)nb[qi];
        double qn = sqrt(nq2);
        double gamma = (double)D * 5.9604644775390625e-08;           // D * 2^-24
        double rho_q = qn > 0 ? sqrt(err2) / qn : 0.0;
        double rho_c = corpus_f32_shadow ? (IS_BF16 ? 0.00390625 : 0.00048828125) : 0.0;  // 2^-8 / 2^-11
        double erel = 4.0 * gamma + rho_q + rho_c + rho_q * rho_c + 1e-6;
        double maxn = sqrt((double)max_na) * (1.0 + gamma);
        QPrep p;
        if (metric == AK_METRIC_COSINE) {
            p.eps = erel + 4.0 * gamma;
            p.a = qn > 0 ? 1.0 / qn : 0.0;
            p.b = 0.0;
        } else if (metric == AK_METRIC_IP) {
            p.eps = erel * qn * maxn;
            p.a = 1.0; p.b = 0.0;
        } else {
            double s = qn + maxn;
            p.eps = (2.0 * erel + 6.0 * gamma) * s * s;
            p.a = 2.0; p.b = -nq2;
        }
        prep[qi] = p;
    }
}

// Certificate + output. One thread per query.
//   top_kp  [nq][kp]  approx keys, ascending (best first)
//   fin_keys/fin_ids [nq][k] exact keys/ids ascending
__global__ void k_certify(const uint64_t *__restrict__ top_kp, const uint64_t *__restrict__ fin_keys,
                          const int64_t *__restrict__ fin_ids, const QPrep *__restrict__ prep,
                          const float *__restrict__ nb, int nq, int k, int kp, int metric,
                          int64_t *__restrict__ out_ids, double *__restrict__ out_dist, int *__restrict__ out_cnt,
                          int *__restrict__ cert, int64_t *__restrict__ stats) {
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int valid_c = 0;
    for (int j = 0; j < kp; j++) valid_c += top_kp[(int64_t)qi * kp + j] != KEY_INVALID;
    int cnt = 0;
    double dk = 0.0;
    for (int j = 0; j < k; j++) {
        uint64_t key = fin_keys[(int64_t)qi * k + j];
        bool valid = key != KEY_INVALID;
        out_ids[(int64_t)qi * k + j] = valid ? fin_ids[(int64_t)qi * k + j] : -1;
        double d = valid ? key_dist(key) : __builtin_nan("");
        out_dist[(int64_t)qi * k + j] = d;
        if (valid) { cnt++; dk = d; }
    }
    if (out_cnt) out_cnt[qi] = cnt;
    int ok = 0;
    float nbq = nb[qi];
    bool qfinite = nbq > 0.f && nbq < __builtin_inff();
    if (cnt == k && qfinite && dk == dk) {
        if (valid_c < kp) ok = 1;  // every finite-score row was a candidate
        else {
            QPrep p = prep[qi];
            float smin = key_score((uint32_t)(top_kp[(int64_t)qi * kp + kp - 1] >> 32));
            double bound = p.a * (double)smin + p.b + p.eps;   // upper bound of any non-candidate's exact score
            double t;
            if (metric == AK_METRIC_COSINE) t = 1.0 - dk;
            else if (metric == AK_METRIC_IP) t = -dk;
            else t = -dk * dk;
            ok = t > bound;
        }
    }
    cert[qi] = ok;
    if (stats) atomicAdd((unsigned long long *)&stats[2], (unsigned long long)valid_c);
}

// ---------------------------------------------------------------------------
bool fast_supported(const Index &ix, int nq, int k) {
    if (ix.dtype == AK_DTYPE_F32) return false;            // f32 corpora: exact path (bf16 shadow scan: TODO)
    if (ix.dim % BK != 0) return false;
    if (ix.n < 4096) return false;                          // tiny index: exact path is cheaper
    if (k > 128) return false;
    return nq > 0;
}

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

FastPlan fast_plan(const Index &ix, int nq, int k) {
    FastPlan p;
    p.kprime = k <= 16 ? 64 : (k <= 48 ? 128 : 256);
    if (p.kprime > CAP - BM) p.kprime = CAP - BM;
    p.qtile = BN;
    p.nqg = (nq + BN - 1) / BN;
    int64_t ntiles = (ix.n + BM - 1) / BM;
    int target = 512;  // 256 CUs x 2 resident workgroups
    int ns = target / p.nqg;
    if (ns < 8) ns = 8;
    ns = (ns / 8) * 8;
    while (ns > 8 && ntiles / ns < 4) ns -= 8;             // keep >= 4 tiles per slice
    if (ns > ntiles) ns = (int)ntiles;
    p.nslices = ns < 1 ? 1 : ns;
    int nq_pad = p.nqg * BN;
    size_t bytes = 0;
    bytes += al((size_t)nq_pad * ix.dim * 2);                               // qs
    bytes += al((size_t)nq * sizeof(QPrep));                                // prep
    bytes += al((size_t)p.nslices * p.nqg * BN * CAP * 8);                  // cand
    bytes += al((size_t)nq * p.nslices * p.kprime * 8);                     // out_c
    bytes += al((size_t)nq * p.kprime * 8) * 2;                             // top_kp keys + ids
    bytes += al((size_t)nq * p.kprime * 8) * 2;                             // rerank keys + ids
    bytes += al((size_t)nq * k * 8) * 2;                                    // final keys + ids
    bytes += al(select_scratch_bytes(nq, (int64_t)p.nslices * p.kprime, p.kprime));
    bytes += 4096;
    p.bytes = bytes;
    return p;
}

int fast_search(Index &ix, const float *queries_dev, const float *nb_dev, int nq, int k, const uint8_t *filter_dev,
                int64_t *out_ids_dev, double *out_dist_dev, int *out_cnt_dev, int *cert_dev, int64_t *stats_dev,
                void *ws, const FastPlan &plan, hipStream_t st) {
    const int kp = plan.kprime, ns = plan.nslices, nqg = plan.nqg, nq_pad = nqg * BN;
    char *p = (char *)ws;
    uint16_t *qs = (uint16_t *)p; p += al((size_t)nq_pad * ix.dim * 2);
    QPrep *prep = (QPrep *)p; p += al((size_t)nq * sizeof(QPrep));
    uint64_t *cand = (uint64_t *)p; p += al((size_t)ns * nqg * BN * CAP * 8);
    uint64_t *out_c = (uint64_t *)p; p += al((size_t)nq * ns * kp * 8);
    uint64_t *top_k = (uint64_t *)p; p += al((size_t)nq * kp * 8);
    int64_t *top_i = (int64_t *)p; p += al((size_t)nq * kp * 8);
    uint64_t *rr_k = (uint64_t *)p; p += al((size_t)nq * kp * 8);
    int64_t *rr_i = (int64_t *)p; p += al((size_t)nq * kp * 8);
    uint64_t *fin_k = (uint64_t *)p; p += al((size_t)nq * k * 8);
    int64_t *fin_i = (int64_t *)p; p += al((size_t)nq * k * 8);
    void *scratch = p;

    const bool bf = ix.dtype == AK_DTYPE_BF16;
    if (bf) k_query_prep<true><<<nq_pad, 64, 0, st>>>(queries_dev, nb_dev, nq, nq_pad, ix.dim, ix.metric, ix.max_na, 0, qs, prep);
    else k_query_prep<false><<<nq_pad, 64, 0, st>>>(queries_dev, nb_dev, nq, nq_pad, ix.dim, ix.metric, ix.max_na, 0, qs, prep);
    AK_HIP(hipGetLastError());

    size_t lds = 4 * TILE_BYTES + (BM * 2 + BN * 2 + 4) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        AK_HIP(hipFuncSetAttribute((const void *)k_scan<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AK_HIP(hipFuncSetAttribute((const void *)k_scan<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    unsigned grid = (unsigned)(ns * nqg);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ix.profile) {
        if (ix.prof_used == ix.prof_events.size()) {
            hipEvent_t a, b;
            AK_HIP(hipEventCreate(&a));
            AK_HIP(hipEventCreate(&b));
            ix.prof_events.emplace_back(a, b);
        }
        ev0 = ix.prof_events[ix.prof_used].first;
        ev1 = ix.prof_events[ix.prof_used].second;
        ix.prof_used++;
        AK_HIP(hipEventRecord(ev0, st));
    }
    if (bf) k_scan<true><<<grid, THREADS, lds, st>>>((const uint16_t *)ix.rows, ix.ea, ix.eb, filter_dev, ix.n, ix.dim, qs, nq, ns, nqg, kp, cand, out_c);
    else k_scan<false><<<grid, THREADS, lds, st>>>((const uint16_t *)ix.rows, ix.ea, ix.eb, filter_dev, ix.n, ix.dim, qs, nq, ns, nqg, kp, cand, out_c);
    AK_HIP(hipGetLastError());
    if (ev1) AK_HIP(hipEventRecord(ev1, st));

    int rc = select_topk(out_c, nullptr, nullptr, nq, (int64_t)ns * kp, kp, top_k, top_i, scratch, st);
    if (rc) return rc;
    rc = rerank(ix, queries_dev, nb_dev, nq, kp, top_k, rr_k, rr_i, st);
    if (rc) return rc;
    rc = select_topk(rr_k, rr_i, nullptr, nq, kp, k, fin_k, fin_i, scratch, st);
    if (rc) return rc;
    k_certify<<<(nq + 63) / 64, 64, 0, st>>>(top_k, fin_k, fin_i, prep, nb_dev, nq, k, kp, ix.metric, out_ids_dev,
                                             out_dist_dev, out_cnt_dev, cert_dev, stats_dev);
    AK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ak
