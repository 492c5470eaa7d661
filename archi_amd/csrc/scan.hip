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

constexpr int BK = 64;         // k per LDS stage (128 B per row)

// LDS-DMA: LDS[lds_wave_base + lane*N] <- *g (N = 16 or 4 bytes per lane). Issued
// from inline asm so hipcc does not serialise it against the ds_reads of the
// OTHER ring slots (it cannot prove they do not alias and would wait vmcnt(0)
// before every fragment read). Completion is waited for by hand with a COUNTED
// vmcnt before the step barrier. M0 carries the LDS base; saved/restored inside.
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
__device__ inline void glds4(const void *g, uint32_t lds_wave_base) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_wave_base)
        : "memory");
}
template <int N>
__device__ inline void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
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
template <int CAP>
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

// Tile configuration: WM x WN waves, each wave 64 corpus rows x (NI*32) queries.
template <int WM_, int WN_, int NI_, int NSTAGE_, int MINW_ = 1>
struct ScanCfg {
    static constexpr int MINW = MINW_;           // min waves per SIMD (launch bounds)
    static constexpr int WM = WM_, WN = WN_, NI = NI_, NSTAGE = NSTAGE_;
    static constexpr int NW = WM * WN;
    static constexpr int THREADS = NW * 64;
    static constexpr int BM = WM * 64;            // corpus rows per tile
    static constexpr int BN = WN * NI * 32;       // queries per block
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    static constexpr int A_PIECES = BM / 8, B_PIECES = BN / 8;     // 1 KiB pieces (8 rows x 128 B)
    static constexpr int A_PW = (A_PIECES + NW - 1) / NW;          // pieces per wave
    static constexpr int B_PW = (B_PIECES + NW - 1) / NW;
    static constexpr int LOADS = A_PW + B_PW;                      // glds per wave per stage
    static constexpr int CAP = BM >= 256 ? 1024 : 512;             // append-buffer entries per (block, query)
    static constexpr int LDS_BYTES = NSTAGE * (A_BYTES + B_BYTES) + (2 * BM + 2 * BN + 4) * 4;
    static_assert(A_PIECES % NW == 0, "A pieces must divide over the waves");
    static_assert(B_PIECES % NW == 0 || B_PIECES < NW, "B pieces layout");
};

// rows: [n][D] 16-bit; qs: [nq_pad][D] 16-bit (queries rounded to the scan dtype)
template <bool IS_BF16, class C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void k_scan(const uint16_t *__restrict__ rows, const float *__restrict__ ea,
                                                     const float *__restrict__ eb, const uint8_t *__restrict__ filter,
                                                     int64_t n, int D, const uint16_t *__restrict__ qs, int nq,
                                                     int nslices, int nqg, int kp, uint64_t *__restrict__ cand,
                                                     uint64_t *__restrict__ out_c) {
    constexpr int BM = C::BM, BN = C::BN, NW = C::NW, NI = C::NI, NSTAGE = C::NSTAGE, CAP = C::CAP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;                                   // [NSTAGE][BM][128 B]
    char *sB = smem + NSTAGE * C::A_BYTES;             // [NSTAGE][BN][128 B]
    float *s_ea = (float *)(smem + NSTAGE * (C::A_BYTES + C::B_BYTES));
    float *s_eb = s_ea + BM;
    float *s_thr = s_eb + BM;
    int *s_cnt = (int *)(s_thr + BN);
    int *s_need = s_cnt + BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / C::WN, wc = wave % C::WN;
    // XCD-aware slot mapping: block b runs on XCD b%8; consecutive slots of one XCD
    // walk the query groups of one slice, so a slice is fetched from HBM once and
    // re-read from that XCD's L2 by the other query groups.
    const int b = blockIdx.x;
    int slice, qg;
    if ((nslices & 7) == 0) {
        int xcd = b & 7, j = b >> 3;
        qg = j % nqg;
        slice = xcd + 8 * (j / nqg);
    } else {
        qg = b % nqg;
        slice = b / nqg;
    }
    const int64_t ntiles = (n + BM - 1) / BM;
    const int64_t t0 = ntiles * slice / nslices, t1 = ntiles * (slice + 1) / nslices;
    const int KS = D / BK;
    const int q0 = qg * BN;

    for (int i = tid; i < BN; i += C::THREADS) {
        s_thr[i] = (q0 + i < nq) ? -3.4028234663852886e38f : __builtin_inff();  // padded queries never append
        s_cnt[i] = 0;
    }
    if (tid == 0) *s_need = 0;

    uint64_t *my_cand = cand + ((size_t)blockIdx.x * BN) * CAP;

    // fragment addressing: row r = lane&31, k-half kh = lane>>5, chunk ^= (row>>1)&7
    const int r = lane & 31, kh = lane >> 5;
    const int c0 = kh ^ ((r >> 1) & 7);
    const int a_off = (wr * 64 + r) * 128;        // + mi*32*128
    const int b_off = (wc * NI * 32 + r) * 128;   // + ni*32*128

    // ---- staging cursor (runs NSTAGE-1 steps ahead of the compute cursor) ----
    const int st_row = lane >> 3, st_chunk = lane & 7;
    const uint32_t ldsA = lds_addr(sA), ldsB = lds_addr(sB), ldsE = lds_addr(s_ea);
    const int64_t nsteps = (t1 - t0) * KS;
    const char *aptr[C::A_PW];
    const char *bptr[C::B_PW > 0 ? C::B_PW : 1];
    int64_t s_tile = t0;
    int s_kk = 0, s_buf = 0;
    auto set_aptr = [&](int64_t tile) {
#pragma unroll
        for (int p = 0; p < C::A_PW; p++) {
            int row = (wave * C::A_PW + p) * 8 + st_row;
            int64_t grow = tile * BM + row;
            if (grow >= n) grow = n - 1;
            int gchunk = st_chunk ^ ((row >> 1) & 7);
            aptr[p] = (const char *)rows + grow * D * 2 + gchunk * 16;
        }
    };
    const bool b_active = wave * C::B_PW < C::B_PIECES || C::B_PIECES >= NW;
#pragma unroll
    for (int p = 0; p < C::B_PW; p++) {
        int piece = (C::B_PIECES >= NW) ? wave * C::B_PW + p : wave;
        int row = piece * 8 + st_row;
        int gchunk = st_chunk ^ ((row >> 1) & 7);
        bptr[p] = (const char *)qs + (int64_t)(q0 + (row < BN ? row : 0)) * D * 2 + gchunk * 16;
    }
    set_aptr(t0);
    auto stage_next = [&]() {   // issue the loads of the staging cursor, then advance it
        const uint32_t la = ldsA + s_buf * C::A_BYTES + wave * C::A_PW * 1024;
        const uint32_t lb = ldsB + s_buf * C::B_BYTES + ((C::B_PIECES >= NW) ? wave * C::B_PW : wave) * 1024;
#pragma unroll
        for (int p = 0; p < C::A_PW; p++) glds16(aptr[p] + s_kk * 128, __builtin_amdgcn_readfirstlane(la + p * 1024));
        if (C::B_PIECES >= NW || wave < C::B_PIECES) {
#pragma unroll
            for (int p = 0; p < C::B_PW; p++) glds16(bptr[p] + s_kk * 128, __builtin_amdgcn_readfirstlane(lb + p * 1024));
        }
        s_buf = (s_buf + 1 == NSTAGE) ? 0 : s_buf + 1;
        if (++s_kk == KS) { s_kk = 0; s_tile++; set_aptr(s_tile); }
    };
    (void)b_active;

    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ni][e] = 0.f;

    // prologue: fill NSTAGE-1 slots, wait for the first
    int64_t issued = 0;
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; i++)
        if (issued < nsteps) { stage_next(); issued++; }
    if (NSTAGE == 3 && issued == 2) wait_vm<C::LOADS>(); else wait_vm<0>();
    __syncthreads();

    int64_t tile = t0;
    int kk = 0, cur = 0;
    for (int64_t step = 0; step < nsteps; step++) {
        if (kk == 0) {
            // per-row epilogue terms of this tile -> LDS (read in the epilogue, >= 1 barrier later)
            if (!filter) {
                if (wave < BM / 64) {
                    int64_t grow = tile * BM + wave * 64 + lane;
                    if (grow >= n) grow = n - 1;
                    glds4(ea + grow, __builtin_amdgcn_readfirstlane(ldsE + wave * 256));
                    glds4(eb + grow, __builtin_amdgcn_readfirstlane(ldsE + BM * 4 + wave * 256));
                }
            } else {
                for (int i = tid; i < BM; i += C::THREADS) {
                    int64_t grow = tile * BM + i;
                    bool ok = grow < n && filter[grow];
                    s_ea[i] = ok ? ea[grow] : 0.f;
                    s_eb[i] = ok ? eb[grow] : -__builtin_inff();
                }
            }
        }
        const bool more = issued < nsteps;
        if (more) { stage_next(); issued++; }
        const char *bufA = sA + cur * C::A_BYTES;
        const char *bufB = sB + cur * C::B_BYTES;
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const int coff = (c0 ^ (k2 << 1)) << 4;
            uint4 av[2], bv[NI];
#pragma unroll
            for (int mi = 0; mi < 2; mi++) av[mi] = *(const uint4 *)(bufA + a_off + mi * 32 * 128 + coff);
#pragma unroll
            for (int ni = 0; ni < NI; ni++) bv[ni] = *(const uint4 *)(bufB + b_off + ni * 32 * 128 + coff);
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++) acc[mi][ni] = mfma32<IS_BF16>(av[mi], bv[ni], acc[mi][ni]);
        }
        if (kk == KS - 1) {
            // ---- fused epilogue: score = fma(dot, ea[row], eb[row]); append if >= threshold
            if (KS == 1) { wait_vm<0>(); __syncthreads(); }   // s_ea was requested in this same step
            const uint32_t tile_row0 = (uint32_t)(tile * BM);
            const bool tail = (tile + 1) * BM > n;             // rows past n alias row n-1: mask them
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int qcol = (wc * NI + ni) * 32 + r;
                const float thr = s_thr[qcol];
#pragma unroll
                for (int mi = 0; mi < 2; mi++) {
                    float sc[16];
                    float mx = -__builtin_inff();
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        int lrow = wr * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                        sc[e] = fmaf(acc[mi][ni][e], s_ea[lrow], s_eb[lrow]);
                        if (tail && (int64_t)tile_row0 + lrow >= n) sc[e] = -__builtin_inff();
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
                for (int q = wave; q < BN; q += NW) {
                    int m = s_cnt[q];
                    if (m > CAP - BM) compact_wave<CAP>(my_cand + (size_t)q * CAP, m, kp, lane, &s_thr[q], &s_cnt[q], nullptr);
                }
                __syncthreads();
                if (tid == 0) *s_need = 0;
            }
        }
        // the NEXT step's slot must have landed; the slot after it may stay in flight
        if (NSTAGE == 3 && more && step + 2 < nsteps) wait_vm<C::LOADS>(); else wait_vm<0>();
        __syncthreads();
        cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        if (++kk == KS) { kk = 0; tile++; }
    }

    // final: every query's buffer -> its kp best -> out_c[q][slice][0..kp)
    __syncthreads();
    for (int q = wave; q < BN; q += NW) {
        if (q0 + q >= nq) continue;
        uint64_t *dst = out_c + ((size_t)(q0 + q) * nslices + slice) * kp;
        compact_wave<CAP>(my_cand + (size_t)q * CAP, s_cnt[q], kp, lane, nullptr, nullptr, dst);
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
// host side: configuration table, plan, launch
// ---------------------------------------------------------------------------
using CfgL = ScanCfg<4, 2, 2, 3, 2>;  // 256 x 128, 8 waves, 3-slot ring : MFMA-bound batches (Q > 64)
using CfgM = ScanCfg<4, 1, 2, 3>;   // 256 x 64 , 4 waves, 3-slot ring : HBM-bound, Q <= 64
using CfgS = ScanCfg<4, 1, 1, 3>;   // 256 x 32 , 4 waves, 3-slot ring : HBM-bound, Q <= 32
using CfgO = ScanCfg<2, 2, 2, 2, 2>;  // 128 x 128, 4 waves, 2-slot ring, 2 blocks/CU (first version; A/B reference)

struct CfgInfo { int bm, bn, cap, threads, lds, blocks_per_cu; };
template <class C> constexpr CfgInfo info_of(int bpc) { return CfgInfo{C::BM, C::BN, C::CAP, C::THREADS, C::LDS_BYTES, bpc}; }
static const CfgInfo g_cfgs[4] = {info_of<CfgL>(1), info_of<CfgM>(1), info_of<CfgS>(1), info_of<CfgO>(2)};
enum { CFG_L = 0, CFG_M = 1, CFG_S = 2, CFG_O = 3 };

static int pick_cfg(int nq) {
    if (const char *e = getenv("AK_SCAN_CFG")) {
        switch (e[0]) { case 'L': return CFG_L; case 'M': return CFG_M; case 'S': return CFG_S; case 'O': return CFG_O; }
    }
    if (nq <= 32) return CFG_S;
    if (nq <= 64) return CFG_M;
    return CFG_L;
}

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
    p.cfg = pick_cfg(nq);
    const CfgInfo &c = g_cfgs[p.cfg];
    p.kprime = k <= 16 ? 64 : (k <= 48 ? 128 : 256);
    if (p.kprime > c.cap - c.bm) p.kprime = c.cap - c.bm;
    p.qtile = c.bn;
    p.nqg = (nq + c.bn - 1) / c.bn;
    int64_t ntiles = (ix.n + c.bm - 1) / c.bm;
    int target = 256 * c.blocks_per_cu;                    // resident workgroups on 256 CUs
    if (const char *e = getenv("AK_SCAN_BLOCKS")) target = atoi(e);
    int ns = target / p.nqg;
    if (ns < 8) ns = 8;
    ns = (ns / 8) * 8;
    while (ns > 8 && ntiles / ns < 2) ns -= 8;             // keep >= 2 tiles per slice
    if (ns > ntiles) ns = (int)ntiles;
    p.nslices = ns < 1 ? 1 : ns;
    int nq_pad = p.nqg * c.bn;
    size_t bytes = 0;
    bytes += al((size_t)nq_pad * ix.dim * 2);                               // qs
    bytes += al((size_t)nq * sizeof(QPrep));                                // prep
    bytes += al((size_t)p.nslices * p.nqg * c.bn * c.cap * 8);              // cand
    bytes += al((size_t)nq * p.nslices * p.kprime * 8);                     // out_c
    bytes += al((size_t)nq * p.kprime * 8) * 2;                             // top_kp keys + ids
    bytes += al((size_t)nq * p.kprime * 8) * 2;                             // rerank keys + ids
    bytes += al((size_t)nq * k * 8) * 2;                                    // final keys + ids
    bytes += al(select_scratch_bytes(nq, (int64_t)p.nslices * p.kprime, p.kprime));
    bytes += 4096;
    p.bytes = bytes;
    return p;
}

template <bool BF, class C>
static int launch_scan(const Index &ix, const uint8_t *filter_dev, const uint16_t *qs, int nq, int ns, int nqg, int kp,
                       uint64_t *cand, uint64_t *out_c, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        AK_HIP(hipFuncSetAttribute((const void *)k_scan<BF, C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    k_scan<BF, C><<<(unsigned)(ns * nqg), C::THREADS, C::LDS_BYTES, st>>>((const uint16_t *)ix.rows, ix.ea, ix.eb, filter_dev,
                                                                         ix.n, ix.dim, qs, nq, ns, nqg, kp, cand, out_c);
    AK_HIP(hipGetLastError());
    return 0;
}

int fast_search(Index &ix, const float *queries_dev, const float *nb_dev, int nq, int k, const uint8_t *filter_dev,
                int64_t *out_ids_dev, double *out_dist_dev, int *out_cnt_dev, int *cert_dev, int64_t *stats_dev,
                void *ws, const FastPlan &plan, hipStream_t st) {
    const CfgInfo &c = g_cfgs[plan.cfg];
    const int kp = plan.kprime, ns = plan.nslices, nqg = plan.nqg, nq_pad = nqg * c.bn;
    char *p = (char *)ws;
    uint16_t *qs = (uint16_t *)p; p += al((size_t)nq_pad * ix.dim * 2);
    QPrep *prep = (QPrep *)p; p += al((size_t)nq * sizeof(QPrep));
    uint64_t *cand = (uint64_t *)p; p += al((size_t)ns * nqg * c.bn * c.cap * 8);
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
    int rc = 0;
#define SCAN(CFG)                                                                                         \
    rc = bf ? launch_scan<true, CFG>(ix, filter_dev, qs, nq, ns, nqg, kp, cand, out_c, st)                \
            : launch_scan<false, CFG>(ix, filter_dev, qs, nq, ns, nqg, kp, cand, out_c, st)
    switch (plan.cfg) {
        case CFG_L: SCAN(CfgL); break;
        case CFG_M: SCAN(CfgM); break;
        case CFG_S: SCAN(CfgS); break;
        default: SCAN(CfgO); break;
    }
#undef SCAN
    if (rc) return rc;
    if (ev1) AK_HIP(hipEventRecord(ev1, st));

    rc = select_topk(out_c, nullptr, nullptr, nq, (int64_t)ns * kp, kp, top_k, top_i, scratch, st);
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
