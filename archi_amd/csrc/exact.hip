// exact.hip -- reference arithmetic for every (query,row) + exact hierarchical
// top-k. This is (1) the first correct HIP path, (2) the fallback for queries
// the MFMA fast path cannot certify, (3) the re-rank arithmetic of the fast
// path (same device function).
//
// Arithmetic restated: pgvector vector.c (VectorCosineSimilarity /
// VectorL2SquaredDistance / VectorInnerProduct) as called by
// /root/reference/src/data_manager/vectorstore/postgres_vectorstore.py:322 --
// float32 accumulators, strictly sequential i = 0..dim-1, one rounding per
// multiply and per add (NO fma contraction), float8 result. See
// oracle/knn_oracle.c for the CPU statement of the same thing.
#include "index.h"
#include "switches.h"

namespace ak {

#pragma clang fp contract(off)

// Stored row elements in chunks of 8 (16 B for 16-bit dtypes, 2 x 16 B for f32): the reference
// arithmetic is a strictly sequential chain, so the only thing to vectorise is the LOAD.
template <int DT>
__device__ inline void load8(const typename Store<DT>::T *row, int i, float (&v)[8]) {
    if constexpr (DT == AK_DTYPE_F32) {
        float4 a = *(const float4 *)(row + i), b = *(const float4 *)(row + i + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        uint4 u = *(const uint4 *)(row + i);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint16_t lo = (uint16_t)(w[j] & 0xffffu), hi = (uint16_t)(w[j] >> 16);
            v[2 * j] = DT == AK_DTYPE_BF16 ? bf16_to_f32(lo) : f16_to_f32(lo);
            v[2 * j + 1] = DT == AK_DTYPE_BF16 ? bf16_to_f32(hi) : f16_to_f32(hi);
        }
    }
}

// One (query,row) distance in the oracle's arithmetic. `na` is the row's
// precomputed pgvector-order sum of squares, `nb` the query's.
template <int DT>
__device__ inline double exact_distance(const typename Store<DT>::T *row, const float *q, int dim,
                                        int metric, float na, float nb) {
    using S = Store<DT>;
    const bool vec = (dim % 8 == 0) && (((uintptr_t)row & 15) == 0);
    float acc = 0.0f;
    if (metric == AK_METRIC_L2) {
        int i = 0;
        if (vec)
            for (; i < dim; i += 8) {
                float v[8];
                load8<DT>(row, i, v);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float diff = __fsub_rn(v[j], q[i + j]);
                    acc = __fadd_rn(acc, __fmul_rn(diff, diff));
                }
            }
        for (; i < dim; i++) {
            float diff = __fsub_rn(S::load(row, i), q[i]);
            acc = __fadd_rn(acc, __fmul_rn(diff, diff));
        }
        return sqrt((double)acc);
    }
    int i = 0;
    if (vec)
        for (; i < dim; i += 8) {
            float v[8];
            load8<DT>(row, i, v);
#pragma unroll
            for (int j = 0; j < 8; j++) acc = __fadd_rn(acc, __fmul_rn(v[j], q[i + j]));
        }
    for (; i < dim; i++) acc = __fadd_rn(acc, __fmul_rn(S::load(row, i), q[i]));
    if (metric == AK_METRIC_IP) return (double)(-acc);
    double sim = (double)acc / sqrt((double)na * (double)nb);
    if (sim > 1.0) sim = 1.0;
    else if (sim < -1.0) sim = -1.0;
    return 1.0 - sim;
}

// Thread per row, QB queries per pass (row element loaded once, reused QB times).
template <int DT, int QB>
__global__ __launch_bounds__(256) void k_exact_dist(const typename Store<DT>::T *__restrict__ rows,
                                                    const float *__restrict__ na,
                                                    const uint8_t *__restrict__ alive,
                                                    const uint8_t *__restrict__ filter, int64_t n, int dim,
                                                    int metric, const float *__restrict__ queries,
                                                    const float *__restrict__ nb, int nq,
                                                    uint64_t *__restrict__ keys) {
    using S = Store<DT>;
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    bool ok = alive[r] && (!filter || filter[r]);
    if (!ok) {
        for (int qi = 0; qi < nq; qi++) keys[(int64_t)qi * n + r] = KEY_INVALID;
        return;
    }
    const typename S::T *row = rows + r * (int64_t)dim;
    float acc[QB];
#pragma unroll
    for (int qi = 0; qi < QB; qi++) acc[qi] = 0.0f;
    if (metric == AK_METRIC_L2) {
        for (int i = 0; i < dim; i++) {
            float a = S::load(row, i);
#pragma unroll
            for (int qi = 0; qi < QB; qi++) {
                float diff = __fsub_rn(a, queries[qi * dim + i]);  // wave-uniform -> scalar loads
                acc[qi] = __fadd_rn(acc[qi], __fmul_rn(diff, diff));
            }
        }
    } else {
        for (int i = 0; i < dim; i++) {
            float a = S::load(row, i);
#pragma unroll
            for (int qi = 0; qi < QB; qi++)
                acc[qi] = __fadd_rn(acc[qi], __fmul_rn(a, queries[qi * dim + i]));
        }
    }
    float nar = na[r];
#pragma unroll
    for (int qi = 0; qi < QB; qi++) {
        if (qi >= nq) break;
        double d;
        if (metric == AK_METRIC_L2) d = sqrt((double)acc[qi]);
        else if (metric == AK_METRIC_IP) d = (double)(-acc[qi]);
        else {
            double sim = (double)acc[qi] / sqrt((double)nar * (double)nb[qi]);
            if (sim > 1.0) sim = 1.0;
            else if (sim < -1.0) sim = -1.0;
            d = 1.0 - sim;
        }
        keys[(int64_t)qi * n + r] = dist_key(d);
    }
}

// nb[q] = sum of squares in float32, element order, one rounding per multiply and per add (the reference's
// accumulator). The chain is sequential, so one wave per query stages the row through LDS with coalesced
// loads and every lane then walks it with broadcast reads (a thread-per-query loop reads 64 rows at once).
__global__ __launch_bounds__(256) void k_query_norms(const float *__restrict__ q, int nq, int dim, float *__restrict__ nb) {
    __shared__ __attribute__((aligned(16))) float s_row[4][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= nq) return;                       // whole wave leaves; no block barrier below
    const float *v = q + (int64_t)i * dim;
    float *row = s_row[wave];
    float s = 0.0f;
    for (int j0 = 0; j0 < dim; j0 += 1024) {
        const int len = dim - j0 < 1024 ? dim - j0 : 1024;
        for (int j = lane; j < len; j += 64) row[j] = v[j0 + j];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): the wave's LDS writes have landed
        int j = 0;
        for (; j + 4 <= len; j += 4) {
            const float4 x = *(const float4 *)(row + j);
            s = __fadd_rn(s, __fmul_rn(x.x, x.x));
            s = __fadd_rn(s, __fmul_rn(x.y, x.y));
            s = __fadd_rn(s, __fmul_rn(x.z, x.z));
            s = __fadd_rn(s, __fmul_rn(x.w, x.w));
        }
        for (; j < len; j++) s = __fadd_rn(s, __fmul_rn(row[j], row[j]));
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) nb[i] = s;
}

int query_norms(const float *queries_dev, int nq, int dim, float *nb_dev, hipStream_t st) {
    if (nq <= 0) return 0;
    k_query_norms<<<(nq + 3) / 4, 256, 0, st>>>(queries_dev, nq, dim, nb_dev);
    AK_HIP(hipGetLastError());
    return 0;
}

// Re-rank kernel used by the fast path: thread per candidate.
// cand [nq][kp] approx keys with the row slot in the low 32 bits.
template <int DT>
__global__ void k_rerank(const typename Store<DT>::T *__restrict__ rows, const float *__restrict__ na,
                         const int64_t *__restrict__ ids, int dim, int metric,
                         const float *__restrict__ queries, const float *__restrict__ nb, int nq, int kp,
                         const uint64_t *__restrict__ cand, uint64_t *__restrict__ okeys,
                         int64_t *__restrict__ oids) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * kp) return;
    int qi = t / kp;
    uint64_t c = cand[t];
    if (c == KEY_INVALID) { okeys[t] = KEY_INVALID; oids[t] = -1; return; }
    int64_t r = (int64_t)(uint32_t)c;
    double d = exact_distance<DT>(rows + r * (int64_t)dim, queries + (int64_t)qi * dim, dim, metric, na[r],
                                  nb[qi]);
    okeys[t] = dist_key(d);
    oids[t] = ids[r];
}

int rerank(Index &ix, const float *queries_dev, const float *nb_dev, int nq, int kp, const uint64_t *cand,
           uint64_t *okeys, int64_t *oids, hipStream_t st) {
    int total = nq * kp;
    int grid = (total + 127) / 128;
#define LAUNCH(DT)                                                                                   \
    k_rerank<DT><<<grid, 128, 0, st>>>((const Store<DT>::T *)ix.rows, ix.na, ix.ids, ix.dim, ix.metric, \
                                       queries_dev, nb_dev, nq, kp, cand, okeys, oids)
    if (ix.dtype == AK_DTYPE_F32) LAUNCH(AK_DTYPE_F32);
    else if (ix.dtype == AK_DTYPE_BF16) LAUNCH(AK_DTYPE_BF16);
    else LAUNCH(AK_DTYPE_F16);
#undef LAUNCH
    AK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Fused tail of the fast path: select + exact re-rank + final top-k + certificate, one 256-thread workgroup per query
// (was k_select_keys + k_rerank + k_finalize: three launches reading / writing [nq][k'] arrays, and a re-rank whose one
// thread per candidate walked its row with dependent 16-byte global loads -- 37-42 us for 768 dimensions at ANY query
// count). Here the candidate rows are fetched by all 256 threads at once (every load of a super-panel in flight together),
// parked in registers, and fed panel by panel through LDS to the one wave that runs the 64 sequential float32 chains.
// ---------------------------------------------------------------------------
constexpr int TL_THREADS = 256, TL_CAP = 2048, TL_PW = 64;    // sort capacity (keys), panel width (elements)

template <int DT>
__global__ __launch_bounds__(TL_THREADS) void k_tail(const typename Store<DT>::T *__restrict__ rows, const float *__restrict__ na,
                                                     const int64_t *__restrict__ ids, int dim, int metric,
                                                     const float *__restrict__ queries, const float *__restrict__ nb,
                                                     const uint64_t *__restrict__ list, int64_t lcap, const int *__restrict__ cnt_g,
                                                     const unsigned int *__restrict__ thr_g, const float *__restrict__ thr0,
                                                     const QPrep *__restrict__ prep, int k, int64_t *__restrict__ out_ids,
                                                     double *__restrict__ out_dist, int *__restrict__ out_cnt,
                                                     int *__restrict__ cert, int64_t *__restrict__ stats, int flags) {
    using S = Store<DT>;
    constexpr int KP = TAIL_KP, ES = (int)sizeof(typename S::T);
    constexpr int ROWB = TL_PW * ES;                 // panel bytes per row: 128 (16-bit) / 256 (f32)
    constexpr int STRIDE = ROWB + 16;                // + one chunk: consecutive rows start 4 banks apart
    constexpr int CPR = ROWB / 16;                   // 16-byte chunks per panel row
    constexpr int CPT = KP * CPR / TL_THREADS;       // chunks per thread per panel: 2 / 4
    constexpr int SPP = 24 / CPT;                    // panels per super-panel: 12 / 6 (24 uint4 = 96 VGPRs parked)
    constexpr int RAW = TL_CAP * 8 > 2 * KP * STRIDE ? TL_CAP * 8 : 2 * KP * STRIDE;
    __shared__ __attribute__((aligned(16))) char s_raw[RAW];             // sort array, then the two panel buffers
    __shared__ __attribute__((aligned(16))) float s_q[TAIL_MAX_DIM];   // the whole query row, staged once
    __shared__ uint64_t s_top[KP];
    __shared__ uint64_t s_rk[KP];
    __shared__ int64_t s_ri[KP];
    __shared__ int s_n;
    uint64_t *s = (uint64_t *)s_raw;
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Filter: a candidate below the main pass's starting threshold theta, or below ANY workgroup's final threshold (each
    // is that workgroup's k-th best - 3 eps, a lower bound of the global k-th best - 3 eps; thr_g holds their maximum),
    // cannot reach the exact top-k. Without a seeding pass every slice exports its own >= k best, 2 500+ keys per query,
    // of which a few dozen clear the maximum.
    float theta = thr0 ? thr0[qi] : -__builtin_inff();
    if (!(theta == theta)) theta = -__builtin_inff();
    const unsigned int tg = thr_g[qi];
    const float tmax = tg ? key_score(~tg) : -__builtin_inff();
    const float cutoff = tmax > theta ? tmax : theta;
    const uint32_t cut_key = score_key(cutoff);       // keep <=> score key <= cut_key
    const int n = cnt_g[qi];
    if (tid == 0) s_n = 0;
    for (int i = tid; i < dim; i += TL_THREADS) s_q[i] = queries[(int64_t)qi * dim + i];
    __syncthreads();
    const uint64_t *lst = list + (int64_t)qi * lcap;
    for (int i = tid; i < n; i += TL_THREADS) {
        const uint64_t key = lst[i];
        if ((uint32_t)(key >> 32) <= cut_key) {
            const int pos = atomicAdd(&s_n, 1);
            if (pos < TL_CAP) s[pos] = key;
        }
    }
    __syncthreads();
    const int total = s_n;
    const bool overflow = total > TL_CAP;            // more survivors than the sort holds: answer, but do not certify
    const int have = overflow ? TL_CAP : total;
    if (tid < KP) s_top[tid] = KEY_INVALID;
    // HISTOGRAM CUT (round 5). More than 512 survivors -- collections without a seeding pass, i.e. up to ~1M chunks, the reference's
    // real size: 650-760 per query on 1M x 384 -- used to go through a 1024- or 2048-key bitonic network (55-66 barrier stages:
    // 17 us of a 37 us tail at Q = 1, 33 of 56 at Q = 256). Only the best k' = 64 keys are wanted: a 256-bin histogram over the
    // survivors' score-key range finds the bin in which the 64th key falls, the keys up to that bin (typically 64-100) are
    // compacted and ranked by counting like the short lists. Exact: everything left out is larger than everything kept. A pile-up
    // in the cut bin (more than 512 keys up to it: near-equal scores) falls back to the network.
    __shared__ __attribute__((aligned(16))) uint64_t s_sel[2 * TL_THREADS + 4];
    __shared__ unsigned int s_hist[TL_THREADS];
    __shared__ unsigned int s_lo, s_hi;
    __shared__ int s_cutbin, s_ncut, s_nsel;
    const uint64_t *arr = s;
    uint64_t *arr_w = s;
    int cntv = have;
    bool by_count = have <= 2 * TL_THREADS;
    if (!by_count && !(flags & 1)) {
        static_assert(TL_THREADS == 256, "histogram cut: one bin per thread");
        if (tid == 0) { s_lo = 0xffffffffu; s_hi = 0u; s_nsel = 0; s_cutbin = 255; s_ncut = have; }
        s_hist[tid] = 0u;
        __syncthreads();
        unsigned int mylo = 0xffffffffu, myhi = 0u;
        for (int i = tid; i < have; i += TL_THREADS) {
            const unsigned int k32 = (unsigned int)(s[i] >> 32);
            mylo = k32 < mylo ? k32 : mylo; myhi = k32 > myhi ? k32 : myhi;
        }
        atomicMin(&s_lo, mylo); atomicMax(&s_hi, myhi);
        __syncthreads();
        const unsigned int lo = s_lo, range = s_hi - lo;
        const int shift = range < 256u ? 0 : (32 - __clz((int)range)) - 8;          // (range >> shift) < 256
        for (int i = tid; i < have; i += TL_THREADS) atomicAdd(&s_hist[((unsigned int)(s[i] >> 32) - lo) >> shift], 1u);
        __syncthreads();
        if (wave == 0) {
            unsigned int c[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { c[j] = s_hist[lane * 4 + j]; sum += c[j]; }
            unsigned int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const unsigned int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
            unsigned int run = incl - sum;
            int mybin = -1; unsigned int mycum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { run += c[j]; if (mybin < 0 && run >= (unsigned int)KP) { mybin = lane * 4 + j; mycum = run; } }
            const uint64_t found = __ballot(mybin >= 0);
            if (found) {
                const int src = __builtin_ctzll(found);
                const int cb = __shfl(mybin, src); const unsigned int cc = __shfl(mycum, src);
                if (lane == 0) { s_cutbin = cb; s_ncut = (int)cc; }
            }
        }
        __syncthreads();
        const int ncut = s_ncut;
        if (ncut <= 2 * TL_THREADS) {
            const unsigned int cutbin = (unsigned int)s_cutbin;
            for (int i = tid; i < have; i += TL_THREADS) {
                const uint64_t key = s[i];
                if ((((unsigned int)(key >> 32) - lo) >> shift) <= cutbin) s_sel[atomicAdd(&s_nsel, 1)] = key;
            }
            arr = s_sel; arr_w = s_sel; cntv = ncut; by_count = true;
        }
    }
    if (by_count && !(flags & 1)) {
        // a few dozen to a few hundred keys: rank by counting (keys are distinct). Every thread walks the list with broadcast LDS
        // reads; one barrier instead of the 21-45 of a bitonic network.
        __syncthreads();
        uint64_t mine[2];
        int rank[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 2; e++) mine[e] = tid + e * TL_THREADS < cntv ? arr[tid + e * TL_THREADS] : KEY_INVALID;
        const int cnt4 = (cntv + 3) & ~3;             // arr[cntv .. cnt4) is padded below; 4 keys per step keep the LDS reads in flight
        for (int j = cntv + tid; j < cnt4; j += TL_THREADS) arr_w[j] = KEY_INVALID;
        __syncthreads();
#pragma unroll 2
        for (int j = 0; j < cnt4; j += 4) {
            const uint4 a = *(const uint4 *)(arr + j), b = *(const uint4 *)(arr + j + 2);
            const uint64_t o0 = ((uint64_t)a.y << 32) | a.x, o1 = ((uint64_t)a.w << 32) | a.z,
                           o2 = ((uint64_t)b.y << 32) | b.x, o3 = ((uint64_t)b.w << 32) | b.z;
            rank[0] += (o0 < mine[0]) + (o1 < mine[0]) + (o2 < mine[0]) + (o3 < mine[0]);
            rank[1] += (o0 < mine[1]) + (o1 < mine[1]) + (o2 < mine[1]) + (o3 < mine[1]);
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
            if (mine[e] != KEY_INVALID && rank[e] < KP) s_top[rank[e]] = mine[e];
        __syncthreads();
    } else {
        int m = KP;
        while (m < have) m <<= 1;
        for (int i = have + tid; i < m; i += TL_THREADS) s[i] = KEY_INVALID;
        __syncthreads();
        if (!(flags & 1))
        for (int size = 2; size <= m; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (m >> 1); t += TL_THREADS) {
                    const int p2 = 2 * t - (t & (stride - 1));
                    const uint64_t a = s[p2], b = s[p2 + stride];
                    const bool up = (p2 & size) == 0;
                    if ((a > b) == up) { s[p2] = b; s[p2 + stride] = a; }
                }
                __syncthreads();
            }
        }
        if (tid < KP) s_top[tid] = s[tid];
        __syncthreads();                              // the sort array is free from here on: panel buffers
    }
    const int valid_c = have < KP ? have : KP;
    const uint64_t last_c = s_top[KP - 1];

    // ---- exact re-rank of the valid_c candidates: chain owner = thread t < KP (wave 0)
    // loader role: chunk c of the panel = row (c / CPR), 16-byte piece (c % CPR); thread handles chunks tid + j*256
    float acc = 0.0f;
    const int npanels = dim / TL_PW;                  // the fast path guarantees dim % 64 == 0
    const char *rbase[CPT];
#pragma unroll
    for (int j = 0; j < CPT; j++) {
        const int c = tid + j * TL_THREADS, r = c / CPR;
        const uint64_t key = s_top[r];
        // rows of missing candidates read row 0 (never used: their chains do not run)
        rbase[j] = (const char *)rows + (int64_t)(uint32_t)(key != KEY_INVALID ? key : 0) * dim * ES + (c % CPR) * 16;
    }
    for (int p0 = 0; p0 < ((flags & 2) ? 0 : npanels); p0 += SPP) {
        const int np = npanels - p0 < SPP ? npanels - p0 : SPP;
        // Every load of the super-panel is issued before anything waits (inline asm: hipcc sinks plain loads to their
        // LDS writes below, behind the panel barriers -- one exposed memory round trip per panel instead of one per
        // super-panel), parked in 24 x 4 registers, then fed to LDS panel by panel.
        uint4 park[SPP][CPT];
#pragma unroll
        for (int pp = 0; pp < SPP; pp++)
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const char *g = rbase[j] + (int64_t)(p0 + (pp < np ? pp : np - 1)) * ROWB;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(park[pp][j]) : "v"(g) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (volatile asm statements keep their order: nothing reads a parked register before the wait)
#pragma unroll
        for (int pp = 0; pp < SPP; pp++)
#pragma unroll
            for (int j = 0; j < CPT; j++)
                asm volatile("" : "+v"(park[pp][j].x), "+v"(park[pp][j].y), "+v"(park[pp][j].z), "+v"(park[pp][j].w));
#pragma unroll
        for (int pp = 0; pp < SPP; pp++) {
            if (pp < np) {                            // block-uniform
                char *buf = s_raw + (pp & 1) * KP * STRIDE;
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    const int c = tid + j * TL_THREADS;
                    *(uint4 *)(buf + (c / CPR) * STRIDE + (c % CPR) * 16) = park[pp][j];
                }
                __syncthreads();                      // one barrier per panel: the buffers alternate
                if (wave == 0 && lane < valid_c && !(flags & 4)) {
                    const char *row = buf + lane * STRIDE;
                    const float *qq = s_q + (p0 + pp) * TL_PW;
#pragma unroll
                    for (int i = 0; i < TL_PW; i += 8) {
                        float v[8];
                        if constexpr (DT == AK_DTYPE_F32) {
                            const float4 a = *(const float4 *)(row + i * 4), b = *(const float4 *)(row + i * 4 + 16);
                            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                        } else {
                            const uint4 u = *(const uint4 *)(row + i * 2);
                            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const uint16_t lo = (uint16_t)(w[e] & 0xffffu), hi = (uint16_t)(w[e] >> 16);
                                v[2 * e] = DT == AK_DTYPE_BF16 ? bf16_to_f32(lo) : f16_to_f32(lo);
                                v[2 * e + 1] = DT == AK_DTYPE_BF16 ? bf16_to_f32(hi) : f16_to_f32(hi);
                            }
                        }
                        if (metric == AK_METRIC_L2) {
#pragma unroll
                            for (int e = 0; e < 8; e++) {
                                const float diff = __fsub_rn(v[e], qq[i + e]);
                                acc = __fadd_rn(acc, __fmul_rn(diff, diff));
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; e++) acc = __fadd_rn(acc, __fmul_rn(v[e], qq[i + e]));
                        }
                    }
                }
            }
        }
        __syncthreads();                              // the next super-panel starts writing buffer 0 again
    }
    if (tid < KP) {
        uint64_t dk = KEY_INVALID;
        int64_t id = INT64_MAX;
        if (tid < valid_c) {
            const int64_t r = (int64_t)(uint32_t)s_top[tid];
            double d;
            if (metric == AK_METRIC_L2) d = sqrt((double)acc);
            else if (metric == AK_METRIC_IP) d = (double)(-acc);
            else {
                double sim = (double)acc / sqrt((double)na[r] * (double)nb[qi]);
                if (sim > 1.0) sim = 1.0;
                else if (sim < -1.0) sim = -1.0;
                d = 1.0 - sim;
            }
            dk = dist_key(d);
            id = ids[r];
        }
        s_rk[tid] = dk;
        s_ri[tid] = id;
    }
    __syncthreads();
    if (wave != 0) return;
    // ---- final top-k by (distance key, id): rank by counting over the 64 re-ranked candidates (pairs are distinct: ids are)
    const uint64_t ek = s_rk[lane];
    const int64_t ei = s_ri[lane];
    int rnk = 0;
    if (!(flags & 8))
        for (int j = 0; j < KP; j++) {
            const uint64_t ok_ = s_rk[j];
            const int64_t oi_ = s_ri[j];
            rnk += (ok_ < ek || (ok_ == ek && oi_ < ei)) ? 1 : 0;
        }
    const bool valid = ek != KEY_INVALID;
    const int cnt = __popcll(__ballot(valid && rnk < k));
    if (valid && rnk < k) {
        out_ids[(int64_t)qi * k + rnk] = ei;
        out_dist[(int64_t)qi * k + rnk] = key_dist(ek);
    }
    for (int r2 = cnt + lane; r2 < k; r2 += 64) {      // fewer than k candidates: pad the tail
        out_ids[(int64_t)qi * k + r2] = -1;
        out_dist[(int64_t)qi * k + r2] = __builtin_nan("");
    }
    // the k-th best distance (rank k-1) for the certificate
    const uint64_t kth_mask = __ballot(valid && rnk == k - 1);
    double dkth = 0.0;
    if (kth_mask) {
        const int src = __builtin_ctzll(kth_mask);
        dkth = key_dist(__shfl(ek, src));
    }
    if (lane != 0) return;
    if (out_cnt) out_cnt[qi] = cnt;
    int ok = 0;
    const float nbq = nb[qi];
    const bool qfinite = nbq > 0.f && nbq < __builtin_inff();
    if (cnt == k && qfinite && dkth == dkth && !overflow) {
        const QPrep p = prep[qi];
        // non-candidates: rows below a workgroup's final threshold (max over the workgroups), candidates dropped here
        // because they scored below the main pass's starting threshold theta (every main-pass workgroup starts from theta
        // and only raises it, so that maximum is >= theta already), and -- when more than k' survived -- rows below the k'-th
        float smin = cutoff > -3.0e38f ? cutoff : -__builtin_inff();
        // k' or more survivors: the k'-th best also bounds what a WORKGROUP dropped when it cut its own export to its k'
        // best (those score no higher than that workgroup's k'-th, which is no higher than the merged k'-th)
        if (total >= KP) smin = fmaxf(smin, key_score((uint32_t)(last_c >> 32)));
        if (smin == -__builtin_inff()) ok = 1;  // every finite-score row was a candidate
        else {
            const double bound = p.a * (double)smin + p.b + p.eps;   // upper bound of any non-candidate's exact score
            double t;
            if (metric == AK_METRIC_COSINE) t = 1.0 - dkth;
            else if (metric == AK_METRIC_IP) t = -dkth;
            else t = -dkth * dkth;
            ok = t > bound;
        }
    }
    cert[qi] = ok;
    if (stats) atomicAdd((unsigned long long *)&stats[2], (unsigned long long)valid_c);
}

int fused_tail(Index &ix, const float *queries_dev, const float *nb_dev, int nq, int k, const uint64_t *list, int64_t lcap,
               const int *cnt, const unsigned int *thr_max, const float *thr0, const QPrep *prep, int64_t *out_ids_dev,
               double *out_dist_dev, int *out_cnt_dev, int *cert_dev, int64_t *stats_dev, hipStream_t st) {
    if (nq <= 0) return 0;
    if (ix.dim % TL_PW != 0 || ix.dim > TAIL_MAX_DIM) AK_FAIL(-1, "fused_tail: dim must be a multiple of 64, at most 4096");
#define LAUNCH(DT)                                                                                                       \
    k_tail<DT><<<nq, TL_THREADS, 0, st>>>((const Store<DT>::T *)ix.rows, ix.na, ix.ids, ix.dim, ix.metric, queries_dev, nb_dev, \
                                          list, lcap, cnt, thr_max, thr0, prep, k, out_ids_dev, out_dist_dev, out_cnt_dev, \
                                          cert_dev, stats_dev, switches().tail_ablate.load(std::memory_order_relaxed))
    if (ix.dtype == AK_DTYPE_F32) LAUNCH(AK_DTYPE_F32);
    else if (ix.dtype == AK_DTYPE_BF16) LAUNCH(AK_DTYPE_BF16);
    else LAUNCH(AK_DTYPE_F16);
#undef LAUNCH
    AK_HIP(hipGetLastError());
    return 0;
}

#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------
// Hierarchical exact selection of the k smallest (key,id) pairs.
// One 256-thread block per (chunk of 4096 entries, query): entries live in
// registers, k rounds of block-wide argmin.
// ---------------------------------------------------------------------------
constexpr int SEL_THREADS = 256;
constexpr int SEL_EPT = 16;
constexpr int SEL_CHUNK = SEL_THREADS * SEL_EPT;

__device__ inline bool pair_less(uint64_t k1, int64_t i1, uint64_t k2, int64_t i2) {
    return k1 < k2 || (k1 == k2 && i1 < i2);
}

__global__ __launch_bounds__(SEL_THREADS) void k_select(const uint64_t *__restrict__ keys,
                                                        const int64_t *__restrict__ ids,
                                                        const int64_t *__restrict__ idmap, int64_t n_in,
                                                        int64_t in_stride, int k, uint64_t *__restrict__ okeys,
                                                        int64_t *__restrict__ oids) {
    __shared__ uint64_t s_k[SEL_THREADS / WAVE];
    __shared__ int64_t s_i[SEL_THREADS / WAVE];
    const int chunk = blockIdx.x, qi = blockIdx.y, nchunks = gridDim.x;
    const int tid = threadIdx.x;
    const uint64_t *kin = keys + (int64_t)qi * in_stride;
    const int64_t *iin = ids ? ids + (int64_t)qi * in_stride : nullptr;
    uint64_t ek[SEL_EPT];
    int64_t ei[SEL_EPT];
#pragma unroll
    for (int e = 0; e < SEL_EPT; e++) {
        int64_t idx = (int64_t)chunk * SEL_CHUNK + e * SEL_THREADS + tid;
        if (idx < n_in) {
            ek[e] = kin[idx];
            ei[e] = iin ? iin[idx] : (idmap ? idmap[idx] : idx);
        } else {
            ek[e] = KEY_INVALID;
            ei[e] = INT64_MAX;
        }
        if (ek[e] == KEY_INVALID) ei[e] = INT64_MAX;
    }
    uint64_t *ok = okeys + ((int64_t)qi * nchunks + chunk) * k;
    int64_t *oi = oids + ((int64_t)qi * nchunks + chunk) * k;
    for (int round = 0; round < k; round++) {
        uint64_t bk = KEY_INVALID;
        int64_t bi = INT64_MAX;
#pragma unroll
        for (int e = 0; e < SEL_EPT; e++)
            if (pair_less(ek[e], ei[e], bk, bi)) { bk = ek[e]; bi = ei[e]; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            uint64_t k2 = __shfl_xor(bk, off);
            int64_t i2 = __shfl_xor(bi, off);
            if (pair_less(k2, i2, bk, bi)) { bk = k2; bi = i2; }
        }
        if ((tid & 63) == 0) { s_k[tid >> 6] = bk; s_i[tid >> 6] = bi; }
        __syncthreads();
        bk = s_k[0]; bi = s_i[0];
#pragma unroll
        for (int w = 1; w < SEL_THREADS / WAVE; w++)
            if (pair_less(s_k[w], s_i[w], bk, bi)) { bk = s_k[w]; bi = s_i[w]; }
        __syncthreads();
        if (tid == 0) { ok[round] = bk; oi[round] = (bk == KEY_INVALID) ? -1 : bi; }
        if (bk == KEY_INVALID) {  // exhausted: pad the tail
            for (int r2 = round + 1 + tid; r2 < k; r2 += SEL_THREADS) { ok[r2] = KEY_INVALID; oi[r2] = -1; }
            break;
        }
#pragma unroll
        for (int e = 0; e < SEL_EPT; e++)
            if (ek[e] == bk && ei[e] == bi) { ek[e] = KEY_INVALID; ei[e] = INT64_MAX; }
    }
}

static inline int64_t nchunks_of(int64_t n) { return n <= 0 ? 1 : (n + SEL_CHUNK - 1) / SEL_CHUNK; }

size_t select_scratch_bytes(int nq, int64_t n_in, int k) {
    // two ping-pong levels, the first being the largest
    int64_t c1 = nchunks_of(n_in);
    if (c1 == 1) return 16;
    int64_t n1 = c1 * k;
    int64_t c2 = nchunks_of(n1);
    int64_t n2 = c2 * k;
    return (size_t)nq * (size_t)(n1 + n2) * 16 + 256;
}

static int select_impl(const uint64_t *keys, const int64_t *ids, const int64_t *idmap, int nq, int64_t n_in,
                       int64_t in_stride, int k, uint64_t *okeys, int64_t *oids, void *scratch, hipStream_t st);

int select_topk(const uint64_t *keys, const int64_t *ids, const int64_t *idmap, int nq, int64_t n_in, int k,
                uint64_t *okeys, int64_t *oids, void *scratch, hipStream_t st) {
    return select_impl(keys, ids, idmap, nq, n_in, n_in, k, okeys, oids, scratch, st);
}

// keys [nq][in_stride], only the first n_in of each row are considered
int select_topk_strided(const uint64_t *keys, int nq, int64_t n_in, int64_t in_stride, int k, uint64_t *okeys,
                        int64_t *oids, void *scratch, hipStream_t st) {
    return select_impl(keys, nullptr, nullptr, nq, n_in, in_stride, k, okeys, oids, scratch, st);
}

static int select_impl(const uint64_t *keys, const int64_t *ids, const int64_t *idmap, int nq, int64_t n_in,
                       int64_t in_stride, int k, uint64_t *okeys, int64_t *oids, void *scratch, hipStream_t st) {
    if (nq <= 0) return 0;
    int64_t c1 = nchunks_of(n_in);
    if (c1 == 1) {
        k_select<<<dim3(1, nq), SEL_THREADS, 0, st>>>(keys, ids, idmap, n_in, in_stride, k, okeys, oids);
        AK_HIP(hipGetLastError());
        return 0;
    }
    int64_t n1 = c1 * k;
    int64_t c2 = nchunks_of(n1);
    int64_t n2 = c2 * k;
    char *p = (char *)scratch;
    uint64_t *ka = (uint64_t *)p; p += (size_t)nq * n1 * 8;
    int64_t *ia = (int64_t *)p; p += (size_t)nq * n1 * 8;
    uint64_t *kb = (uint64_t *)p; p += (size_t)nq * n2 * 8;
    int64_t *ib = (int64_t *)p;
    k_select<<<dim3((unsigned)c1, nq), SEL_THREADS, 0, st>>>(keys, ids, idmap, n_in, in_stride, k, ka, ia);
    AK_HIP(hipGetLastError());
    const uint64_t *ck = ka; const int64_t *ci = ia;
    int64_t cn = n1;
    bool to_b = true;
    for (;;) {
        int64_t c = nchunks_of(cn);
        if (c == 1) {
            k_select<<<dim3(1, nq), SEL_THREADS, 0, st>>>(ck, ci, nullptr, cn, cn, k, okeys, oids);
            AK_HIP(hipGetLastError());
            return 0;
        }
        uint64_t *dk = to_b ? kb : ka;
        int64_t *di = to_b ? ib : ia;
        k_select<<<dim3((unsigned)c, nq), SEL_THREADS, 0, st>>>(ck, ci, nullptr, cn, cn, k, dk, di);
        AK_HIP(hipGetLastError());
        ck = dk; ci = di; cn = c * k; to_b = !to_b;
    }
}

__global__ void k_emit(const uint64_t *__restrict__ keys, const int64_t *__restrict__ ids, int nq, int k,
                       int64_t *__restrict__ out_ids, double *__restrict__ out_dist, int *__restrict__ out_cnt) {
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int cnt = 0;
    for (int j = 0; j < k; j++) {
        uint64_t key = keys[(int64_t)qi * k + j];
        bool valid = key != KEY_INVALID;
        out_ids[(int64_t)qi * k + j] = valid ? ids[(int64_t)qi * k + j] : -1;
        out_dist[(int64_t)qi * k + j] = valid ? key_dist(key) : __builtin_nan("");
        cnt += valid;
    }
    if (out_cnt) out_cnt[qi] = cnt;
}

int emit_results(const uint64_t *keys, const int64_t *ids, int nq, int k, int64_t *out_ids, double *out_dist,
                 int *out_cnt, hipStream_t st) {
    if (nq <= 0) return 0;
    k_emit<<<(nq + 63) / 64, 64, 0, st>>>(keys, ids, nq, k, out_ids, out_dist, out_cnt);
    AK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
__global__ void k_fill_empty(int64_t total, int nq, int64_t *__restrict__ out_ids, double *__restrict__ out_dist,
                             int *__restrict__ out_cnt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { out_ids[i] = -1; out_dist[i] = __builtin_nan(""); }
    if (out_cnt && i < nq) out_cnt[i] = 0;
}

constexpr int EXACT_QB = 8;
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t exact_scratch_bytes(const Index &ix, int k) {
    if (ix.n <= 0) return 256;
    return al256((size_t)EXACT_QB * ix.n * 8) + 2 * al256((size_t)EXACT_QB * k * 8) +
           al256(select_scratch_bytes(EXACT_QB, ix.n, k)) + 256;
}

int exact_search(Index &ix, const float *queries_dev, const float *nb_dev, int nq, int k,
                 const uint8_t *filter_dev, int64_t *out_ids_dev, double *out_dist_dev, int *out_cnt_dev,
                 void *ws, hipStream_t st) {
    constexpr int QB = EXACT_QB;
    const int64_t n = ix.n;
    if (nq <= 0) return 0;
    if (n == 0) {   // nothing stored: every slot invalid
        const int64_t total = (int64_t)nq * k;
        const int64_t span = total > nq ? total : nq;
        k_fill_empty<<<(unsigned)((span + 255) / 256), 256, 0, st>>>(total, nq, out_ids_dev, out_dist_dev, out_cnt_dev);
        AK_HIP(hipGetLastError());
        return 0;
    }
    char *p = (char *)ws;
    uint64_t *keys = (uint64_t *)p; p += al256((size_t)QB * n * 8);
    uint64_t *okeys = (uint64_t *)p; p += al256((size_t)QB * k * 8);
    int64_t *oids = (int64_t *)p; p += al256((size_t)QB * k * 8);
    void *scratch = p;
    int rc = 0;
    unsigned grid = (unsigned)((n + 255) / 256);
    for (int q0 = 0; q0 < nq && rc == 0; q0 += QB) {
        int qc = nq - q0 < QB ? nq - q0 : QB;
        const float *qp = queries_dev + (int64_t)q0 * ix.dim;
        // queries beyond qc inside the QB window would read past the buffer:
        // the kernel only touches queries[qi] for qi < QB via the unrolled loop,
        // so run the tail with a narrower instantiation.
#define LAUNCH(DT, QBN)                                                                               \
    k_exact_dist<DT, QBN><<<grid, 256, 0, st>>>((const Store<DT>::T *)ix.rows, ix.na, ix.alive,       \
                                                filter_dev, n, ix.dim, ix.metric, qp, nb_dev + q0, qc, keys)
#define DISPATCH(DT)                       \
    do {                                   \
        if (qc == 8) LAUNCH(DT, 8);        \
        else if (qc >= 4) { qc = 4; LAUNCH(DT, 4); } \
        else if (qc >= 2) { qc = 2; LAUNCH(DT, 2); } \
        else { qc = 1; LAUNCH(DT, 1); }    \
    } while (0)
        if (ix.dtype == AK_DTYPE_F32) DISPATCH(AK_DTYPE_F32);
        else if (ix.dtype == AK_DTYPE_BF16) DISPATCH(AK_DTYPE_BF16);
        else DISPATCH(AK_DTYPE_F16);
#undef DISPATCH
#undef LAUNCH
        if (hipGetLastError() != hipSuccess) { set_error("k_exact_dist launch failed"); rc = -10; break; }
        rc = select_topk(keys, nullptr, ix.ids, qc, n, k, okeys, oids, scratch, st);
        if (rc) break;
        rc = emit_results(okeys, oids, qc, k, out_ids_dev + (int64_t)q0 * k, out_dist_dev + (int64_t)q0 * k,
                          out_cnt_dev ? out_cnt_dev + q0 : nullptr, st);
        q0 += qc - QB;  // advance by qc (loop adds QB)
    }
    return rc;
}

}  // namespace ak
