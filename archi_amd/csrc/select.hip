// select.hip -- key-only top-k selection for the candidate lists of the scan: the k smallest
// 64-bit keys (score key << 32 | row slot, all distinct) of each query, ascending.
// One 256-thread workgroup bitonic-sorts a chunk of up to 4096 keys in LDS (78 compare-exchange
// stages, 19 us) instead of k rounds of block-wide argmin (k = 64: ~200 us); chunks are merged
// hierarchically like select_topk (exact.hip).
#include "index.h"

namespace ak {

constexpr int SK_THREADS = 256, SK_CHUNK = 4096;

// Candidate lists are mostly padding (KEY_INVALID): the valid keys of the chunk are first compacted to
// the front of the LDS array (block-wide prefix sum), then only the next power of two >= max(valid, k)
// is sorted.
// SK_CHUNK_ x SK_THREADS_: 4096 x 256 (chunked, two levels above 4096 keys) or 8192 x 512 (one launch for the 8192
// keys per query of the usual plan: 128 slots x k' = 64 -- the second level was a launch of its own per search).
template <int SK_CHUNK, int SK_THREADS>
__global__ __launch_bounds__(SK_THREADS) void k_select_keys(const uint64_t *__restrict__ keys, int64_t n_in,
                                                            int64_t in_stride, int k, int sort_n,
                                                            uint64_t *__restrict__ okeys) {
    __shared__ uint64_t s[SK_CHUNK];
    __shared__ int s_wsum[SK_THREADS / WAVE];
    const int chunk = blockIdx.x, qi = blockIdx.y, nchunks = gridDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t *kin = keys + (int64_t)qi * in_stride + (int64_t)chunk * SK_CHUNK;
    const int64_t left = n_in - (int64_t)chunk * SK_CHUNK;
    constexpr int EPT = SK_CHUNK / SK_THREADS;   // 16 keys per thread, thread-contiguous so order is kept
    uint64_t v[EPT];
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = e * SK_THREADS + tid;      // coalesced
        v[e] = (i < sort_n && i < left) ? kin[i] : KEY_INVALID;
        cnt += v[e] != KEY_INVALID;
    }
    // exclusive prefix of cnt over the block
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SK_THREADS / WAVE; w++) { if (w < wv) base += s_wsum[w]; total += s_wsum[w]; }
    int pos = base + incl - cnt;
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (v[e] != KEY_INVALID) s[pos++] = v[e];
    int m = 2;
    const int need = total > k ? total : k;
    while (m < need) m <<= 1;                     // sort size: pow2 >= max(valid, k), <= sort_n
    if (m > SK_CHUNK) m = SK_CHUNK;
    for (int i = total + tid; i < m; i += SK_THREADS) s[i] = KEY_INVALID;
    __syncthreads();
    for (int size = 2; size <= m; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (m >> 1); t += SK_THREADS) {
                const int p2 = 2 * t - (t & (stride - 1));
                const uint64_t a = s[p2], b = s[p2 + stride];
                const bool up = (p2 & size) == 0;
                if ((a > b) == up) { s[p2] = b; s[p2 + stride] = a; }
            }
            __syncthreads();
        }
    }
    uint64_t *ok = okeys + ((int64_t)qi * nchunks + chunk) * k;
    for (int i = tid; i < k; i += SK_THREADS) ok[i] = i < m ? s[i] : KEY_INVALID;
}

static inline int pow2_ge(int64_t n) { int p = 2; while (p < n) p <<= 1; return p; }

size_t select_keys_scratch_bytes(int nq, int64_t n_in, int k) {
    int64_t c1 = (n_in + SK_CHUNK - 1) / SK_CHUNK;
    if (c1 <= 1) return 16;
    return (size_t)nq * (size_t)(c1 * k) * 8 * 2 + 256;
}

// keys [nq][in_stride] (first n_in of each row considered) -> okeys [nq][k] ascending
int select_keys_topk(const uint64_t *keys, int nq, int64_t n_in, int64_t in_stride, int k, uint64_t *okeys,
                     void *scratch, hipStream_t st) {
    if (nq <= 0) return 0;
    if (k > SK_CHUNK) AK_FAIL(-1, "select_keys_topk: k too large");
    const uint64_t *ck = keys;
    int64_t cn = n_in, cs = in_stride;
    uint64_t *bufs[2] = {(uint64_t *)scratch, nullptr};
    int which = 0;
    if (cn > SK_CHUNK && cn <= 2 * SK_CHUNK) {      // one launch of the wide variant
        int sort_n = pow2_ge(cn > k ? cn : k);
        if (sort_n > 2 * SK_CHUNK) sort_n = 2 * SK_CHUNK;
        k_select_keys<2 * SK_CHUNK, 2 * SK_THREADS><<<dim3(1, nq), 2 * SK_THREADS, 0, st>>>(ck, cn, cs, k, sort_n, okeys);
        AK_HIP(hipGetLastError());
        return 0;
    }
    for (;;) {
        int64_t c = cn <= 0 ? 1 : (cn + SK_CHUNK - 1) / SK_CHUNK;
        int sort_n = c == 1 ? pow2_ge(cn > k ? cn : k) : SK_CHUNK;
        if (sort_n > SK_CHUNK) sort_n = SK_CHUNK;
        if (c == 1) {
            k_select_keys<SK_CHUNK, SK_THREADS><<<dim3(1, nq), SK_THREADS, 0, st>>>(ck, cn, cs, k, sort_n, okeys);
            AK_HIP(hipGetLastError());
            return 0;
        }
        if (!bufs[1]) bufs[1] = bufs[0] + (size_t)nq * (size_t)(c * k);
        uint64_t *dst = bufs[which];
        k_select_keys<SK_CHUNK, SK_THREADS><<<dim3((unsigned)c, nq), SK_THREADS, 0, st>>>(ck, cn, cs, k, sort_n, dst);
        AK_HIP(hipGetLastError());
        ck = dst; cn = c * k; cs = cn; which ^= 1;
    }
}

}  // namespace ak
