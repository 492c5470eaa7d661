// index.hip -- C-ABI entry points of the corpus index (ak_index_*), row
// ingestion / synthetic generation kernels, and the search orchestration.
// Replaces: INSERT ... %s::vector (postgres_vectorstore.py:168-180), DELETE
// (:516-529), COUNT (:570-585) and the SELECT ... ORDER BY distance LIMIT k
// query (:317-332) of the reference.
#include "index.h"
#include "switches.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <chrono>
#include <dlfcn.h>
#include <unordered_set>

namespace ak {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

// ---- switches (switches.h): the environment is read once, here ------------------------------------------------------
extern "C" char **environ;
static const std::unordered_map<std::string, std::string> &env_snapshot() {
    static const std::unordered_map<std::string, std::string> *snap = [] {
        auto *m = new std::unordered_map<std::string, std::string>();
        for (char **e = environ; e && *e; e++) {
            if (strncmp(*e, "AK_", 3) != 0) continue;
            const char *eq = strchr(*e, '=');
            if (eq) (*m)[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
        }
        return m;
    }();
    return *snap;
}
const char *env_get(const char *name) {
    const auto &m = env_snapshot();
    const auto it = m.find(name);
    return it == m.end() ? nullptr : it->second.c_str();
}
int env_int(const char *name, int dflt) {
    const char *e = env_get(name);
    return e && *e ? atoi(e) : dflt;
}
namespace {
struct SwitchName { const char *name; std::atomic<int> Switches::*field; int dflt; bool is_flag; bool wrong_results; bool is_char; };
const SwitchName g_switch_names[] = {
    {"AK_SCAN_CFG", &Switches::scan_cfg, 0, false, false, true},
    {"AK_SCAN_BLOCKS", &Switches::scan_blocks, 0, false, false, false},
    {"AK_SCAN_NO192", &Switches::scan_no192, 0, true, false, false},
    {"AK_SEED_RATIO", &Switches::seed_ratio, 32, false, false, false},
    {"AK_SEED_DIV", &Switches::seed_div, 0, false, false, false},
    {"AK_PRE_DIV", &Switches::pre_div, 0, false, false, false},
    {"AK_SCAN_NOSEED", &Switches::scan_noseed, 0, true, false, false},
    {"AK_SCAN_NOPRE", &Switches::scan_nopre, 0, true, false, false},
    {"AK_TAIL_OLD", &Switches::tail_old, 0, true, false, false},
    {"AK_SCAN_DBG", &Switches::scan_dbg, 0, true, false, false},
    {"AK_COALESCE_STATS", &Switches::coalesce_stats, 0, true, false, false},
    {"AK_SHARD_INJECT", &Switches::shard_inject, 0, false, false, false},
    {"AK_QUERY_FUSED", &Switches::query_fused, 0, false, false, false},
#if AK_DBG_KERNELS      // WRONG RESULTS: the product library does not even know the names
    {"AK_SCAN_ABLATE", &Switches::scan_ablate, 0, false, true, false},
    {"AK_TAIL_ABLATE", &Switches::tail_ablate, 0, false, true, false},
#endif
};
int switch_value(const SwitchName &n, const char *v) {
    if (!v || !*v) return n.dflt;
    if (n.is_char) return (int)(unsigned char)v[0];
    if (n.is_flag) return 1;                       // presence switches: any non-empty value
    return atoi(v);
}
}  // namespace
Switches &switches() {
    static Switches sw;
    static const bool once = [] {
        for (const SwitchName &n : g_switch_names) {
            if (n.wrong_results && !DBG_KERNELS) continue;      // the product library does not read them
            (sw.*(n.field)).store(switch_value(n, env_get(n.name)), std::memory_order_relaxed);
        }
        if (const char *e = env_get("AK_SCAN_R192")) sw.scan_r192_pm.store((int)(atof(e) * 1000.0 + 0.5), std::memory_order_relaxed);
        return true;
    }();
    (void)once;
    return sw;
}
// at dlopen (ctypes.CDLL, on the loading thread): the snapshot and the table exist before any entry point can run
__attribute__((constructor)) static void ak_read_environment_at_load() { (void)switches(); }
int switches_set(const char *name, const char *value) {
    if (!name) return -1;
    Switches &sw = switches();
    if (!strcmp(name, "AK_SCAN_R192")) {
        sw.scan_r192_pm.store(value && *value ? (int)(atof(value) * 1000.0 + 0.5) : 850, std::memory_order_relaxed);
        return 0;
    }
    for (const SwitchName &n : g_switch_names)
        if (!strcmp(name, n.name)) {
            if (n.wrong_results && !DBG_KERNELS) return -1;
            // the A/B reference tiles X and O are compiled into the dbg library only: refuse them here rather than accept the
            // switch and run the plan's own tile under a probe that believes it forced another (round-5 advisor finding)
            if (!DBG_KERNELS && n.is_char && value && (*value == 'X' || *value == 'O')) return -1;
            (sw.*(n.field)).store(switch_value(n, value), std::memory_order_relaxed);
            return 0;
        }
    return -1;
}

namespace {
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const RoctxApi &roctx() {
    static const RoctxApi api;
    return api;
}
}  // namespace
RoctxRange::RoctxRange(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
RoctxRange::~RoctxRange() { if (on) roctx().pop(); }

int Workspace::reserve(size_t need) {
    if (need <= bytes) return 0;
    if (buf) hipFree(buf);
    buf = nullptr; bytes = 0;
    AK_HIP(hipMalloc(&buf, need));
    bytes = need;
    return 0;
}
void Workspace::release() {
    if (buf) hipFree(buf);
    buf = nullptr; bytes = 0;
}

// Per host thread: its own stream (concurrent ak_index_search calls do not serialise on one queue) and grow-only
// scratch for the host-buffer search -- device block, scan workspace, pinned staging -- so a Q=1 search does not pay
// hipMalloc + hipFree (a device-wide synchronisation) on every call. Released when the thread ends (request-per-thread
// servers create and drop threads all the time); anything above SCRATCH_KEEP is released at the end of the call.
constexpr size_t SCRATCH_KEEP = 256ull << 20;
struct ThreadCtx {
    hipStream_t stream = nullptr;
    char *dev = nullptr; size_t dev_cap = 0;
    char *ws = nullptr; size_t ws_cap = 0;
    char *pin = nullptr; size_t pin_cap = 0;
    void trim() {
        if (dev_cap > SCRATCH_KEEP) { hipFree(dev); dev = nullptr; dev_cap = 0; }
        if (ws_cap > SCRATCH_KEEP) { hipFree(ws); ws = nullptr; ws_cap = 0; }
        if (pin_cap > SCRATCH_KEEP) { hipHostFree(pin); pin = nullptr; pin_cap = 0; }
    }
    ~ThreadCtx() {
        if (ws) hipFree(ws);
        if (dev) hipFree(dev);
        if (pin) hipHostFree(pin);
        if (stream) hipStreamDestroy(stream);
    }
};
static std::atomic<int> g_device{-1};          // ak_init's device: one process per GPU
static thread_local int t_bound_device = -1;
int bind_thread() {
    const int dev = g_device.load(std::memory_order_relaxed);
    if (dev >= 0 && t_bound_device != dev) {
        AK_HIP(hipSetDevice(dev));
        t_bound_device = dev;
    }
    return 0;
}

// A thread-per-request server drops its thread after every search: building a stream and three scratch blocks per request
// cost ~1 ms against a 0.12 ms search. Finished threads therefore hand their context to a small process-wide pool and new
// threads take one from it; only what does not fit the pool is destroyed.
constexpr size_t CTX_POOL_MAX = 32;
static std::mutex g_ctx_mu;
static std::vector<ThreadCtx *> g_ctx_pool;
struct ThreadCtxHandle {
    ThreadCtx *p = nullptr;
    ThreadCtx &get() {
        if (!p) {
            {
                std::lock_guard<std::mutex> lk(g_ctx_mu);
                if (!g_ctx_pool.empty()) { p = g_ctx_pool.back(); g_ctx_pool.pop_back(); }
            }
            if (!p) p = new ThreadCtx();
        }
        return *p;
    }
    ~ThreadCtxHandle() {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            if (g_ctx_pool.size() < CTX_POOL_MAX) { g_ctx_pool.push_back(p); p = nullptr; }
        }
        delete p;
    }
};
static thread_local ThreadCtxHandle t_ctx_handle;
#define t_ctx (t_ctx_handle.get())
static int thread_stream(hipStream_t *out) {
    if (!t_ctx.stream) AK_HIP(hipStreamCreateWithFlags(&t_ctx.stream, hipStreamNonBlocking));
    *out = t_ctx.stream;
    return 0;
}
static int scratch_reserve(char **p, size_t *cap, size_t need, bool pinned) {
    if (need <= *cap) return 0;
    if (*p) { if (pinned) hipHostFree(*p); else hipFree(*p); *p = nullptr; *cap = 0; }
    need = (need + 4095) & ~(size_t)4095;
    if (pinned) AK_HIP(hipHostMalloc((void **)p, need, hipHostMallocDefault));
    else AK_HIP(hipMalloc((void **)p, need));
    *cap = need;
    return 0;
}

// ---------------------------------------------------------------------------
// ingestion kernels
// ---------------------------------------------------------------------------
// a3: x / max(||x||_2, 1e-12)  (torch.nn.functional.normalize semantics), one wave per row
__global__ __launch_bounds__(256) void k_row_invnorm(const float *__restrict__ x, int64_t n, int dim,
                                                     float *__restrict__ inv) {
    int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (r >= n) return;
    const float *row = x + r * (int64_t)dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s = fmaf(row[i], row[i], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    float nrm = sqrtf(s);
    if (lane == 0) inv[r] = nrm < 1e-12f ? 1e-12f : nrm;  // holds the clamped norm
}

template <int DT>
__global__ __launch_bounds__(256) void k_convert(const float *__restrict__ x, const float *__restrict__ nrm,
                                                 int64_t total, int dim, typename Store<DT>::T *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        float v = x[i];
        if (nrm) v = v / nrm[i / dim];
        out[i] = Store<DT>::cvt(v);
    }
}

// In-place fp32 L2 normalise (ak_l2_normalize_dev)
__global__ __launch_bounds__(256) void k_scale_rows(float *__restrict__ x, const float *__restrict__ nrm,
                                                    int64_t total, int dim) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) x[i] = x[i] / nrm[i / dim];
}

#pragma clang fp contract(off)
// per-row statistics of the STORED values: na in pgvector order (a strictly sequential float32 chain per row), epilogue
// terms. Thread per row for the chain, but the rows are staged through LDS in [256 rows][32 columns] panels so that the
// global reads are contiguous 64/128-byte segments (a plain thread-per-row walk touched 256 different lines per load
// instruction: 79 ms per 10M x 768 rows = 190 GB/s; the bulk-load path of the pgvector bridge runs through here).
template <int DT>
__global__ __launch_bounds__(256) void k_row_stats(const typename Store<DT>::T *__restrict__ rows, int64_t r0,
                                                   int64_t n, int dim, int metric, float *__restrict__ na,
                                                   float *__restrict__ ea, float *__restrict__ eb) {
    __shared__ float s_p[256][33];
    const int t = threadIdx.x;
    const int64_t rb = r0 + (int64_t)blockIdx.x * 256;
    const int64_t left = r0 + n - rb;
    const int nrows = left < 256 ? (int)left : 256;
    float s = 0.0f;
    for (int c0 = 0; c0 < dim; c0 += 32) {
        const int w = dim - c0 < 32 ? dim - c0 : 32;
#pragma unroll 4
        for (int i = 0; i < 32; i++) {
            const int idx = i * 256 + t, rr = idx >> 5, cc = idx & 31;
            if (rr < nrows && cc < w) s_p[rr][cc] = Store<DT>::load(rows + (rb + rr) * (int64_t)dim, c0 + cc);
        }
        __syncthreads();
        if (t < nrows)
            for (int j = 0; j < w; j++) {
                const float a = s_p[t][j];
                s = __fadd_rn(s, __fmul_rn(a, a));
            }
        __syncthreads();
    }
    if (t >= nrows) return;
    const int64_t r = rb + t;
    na[r] = s;
    float a = 1.0f, b = 0.0f;
    if (metric == AK_METRIC_COSINE) {
        if (s > 0.0f && s < INFINITY) a = (float)(1.0 / sqrt((double)s));
        else { a = 0.0f; b = -INFINITY; }
    } else if (metric == AK_METRIC_L2) {
        b = -0.5f * s;
    }
    ea[r] = a;
    eb[r] = b;
}
#pragma clang fp contract(fast)

// Synthetic rows: oracle/knn_oracle.c ako_gen_rows, one wave per row.
template <int DT>
__global__ __launch_bounds__(256) void k_generate(typename Store<DT>::T *__restrict__ rows, int64_t slot0,
                                                  int64_t n, int dim, uint64_t seed, uint32_t stream,
                                                  uint64_t row0, int normalise) {
    int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (i >= n) return;
    uint64_t grow = row0 + (uint64_t)i;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    int npairs = (dim + 1) / 2;
    double nrm = 1.0;
    if (normalise) {
        long long S = 0;
        for (int p = lane; p < npairs; p += 64) {
            uint32_t o[4];
            philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)p, stream, k0, k1, o);
            long long v0 = bytesum(o[0]) + bytesum(o[1]) - 1020;
            S += v0 * v0;
            if (2 * p + 1 < dim) {
                long long v1 = bytesum(o[2]) + bytesum(o[3]) - 1020;
                S += v1 * v1;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) S += __shfl_xor(S, off);
        nrm = S ? sqrt((double)S) : 0.0;
    }
    typename Store<DT>::T *out = rows + (slot0 + i) * (int64_t)dim;
    for (int p = lane; p < npairs; p += 64) {
        uint32_t o[4];
        philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)p, stream, k0, k1, o);
        int v0 = bytesum(o[0]) + bytesum(o[1]) - 1020;
        int v1 = bytesum(o[2]) + bytesum(o[3]) - 1020;
        float x0, x1;
        if (normalise) {
            x0 = nrm != 0.0 ? (float)((double)v0 / nrm) : 0.0f;
            x1 = nrm != 0.0 ? (float)((double)v1 / nrm) : 0.0f;
        } else {
            x0 = (float)v0 * 0.00390625f;
            x1 = (float)v1 * 0.00390625f;
        }
        out[2 * p] = Store<DT>::cvt(x0);
        if (2 * p + 1 < dim) out[2 * p + 1] = Store<DT>::cvt(x1);
    }
}

// Upper-bound terms of the scan's common path: a lane of the 32x32 MFMA tile holds, of every 32-row
// block, either rows {8q+0..3} (lane half 0) or {8q+4..7} (half 1); per block and half the max of ea and
// of eb over those 16 rows. Tombstones only lower ea/eb, so the maxima stay valid upper bounds.
__global__ void k_group_bounds(const float *__restrict__ ea, const float *__restrict__ eb, int64_t n_rows,
                               int64_t blk0, int64_t nblk, float *__restrict__ gb) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblk * 2) return;
    const int64_t blk = blk0 + (t >> 1);
    const int half = (int)(t & 1);
    float me = 0.f, mb = -INFINITY;
    for (int q = 0; q < 4; q++)
        for (int j = 0; j < 4; j++) {
            int64_t row = blk * 32 + 8 * q + 4 * half + j;
            if (row < n_rows) { me = fmaxf(me, ea[row]); mb = fmaxf(mb, eb[row]); }
        }
    gb[blk * 4 + half] = me;
    gb[blk * 4 + 2 + half] = mb;
}

__global__ void k_fill_ids(int64_t *ids, uint8_t *alive, int64_t slot0, int64_t n, int64_t id0) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { ids[slot0 + i] = id0 + i; alive[slot0 + i] = 1; }
}

__global__ void k_kill(const int64_t *__restrict__ slots, int64_t n, uint8_t *alive, float *ea, float *eb) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int64_t s = slots[i];
        alive[s] = 0; ea[s] = 0.0f; eb[s] = -INFINITY;
    }
}

template <int DT>
__global__ void k_fetch(const typename Store<DT>::T *__restrict__ rows, const int64_t *__restrict__ slots,
                        int64_t n, int dim, float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    int64_t r = i / dim; int c = (int)(i % dim);
    out[i] = Store<DT>::load(rows + slots[r] * (int64_t)dim, c);
}

// max over rows of a non-negative float array (float bits of non-negative values order like unsigned integers); infinities
// and NaN are skipped. Grid-stride, one atomicMax per wave (was a single 256-thread block: 16 ms per 10M rows).
__global__ __launch_bounds__(256) void k_max_f32(const float *__restrict__ x, int64_t r0, int64_t n, unsigned int *__restrict__ out_bits) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[r0 + i];
        if (v > m && v < INFINITY) m = v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __float_as_uint(m));
}

// f32 corpora are scanned through a 16-bit shadow: rho = |a - shadow(a)| / |a| per row, maximum over the rows (float bits of
// a non-negative value order like unsigned integers). The certificate's corpus-rounding term used the format's worst case
// (2^-8 for bf16 with a factor 2 of slack); the measured maximum is ~2^-9.6, which keeps dense neighbourhoods certifiable.
__global__ __launch_bounds__(256) void k_shadow_rho(const float *__restrict__ rows, const uint16_t *__restrict__ shadow, int64_t slot0,
                                                    int64_t n, int dim, unsigned int *__restrict__ out_bits) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n) return;
    const float *a = rows + (slot0 + r) * (int64_t)dim;
    const uint16_t *sh = shadow + (slot0 + r) * (int64_t)dim;
    double e2 = 0.0, a2 = 0.0;
    for (int i = lane; i < dim; i += 64) {
        const double x = (double)a[i], d = x - (double)bf16_to_f32(sh[i]);
        e2 += d * d; a2 += x * x;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { e2 += __shfl_xor(e2, off); a2 += __shfl_xor(a2, off); }
    if (lane == 0 && a2 > 0.0 && e2 == e2 && a2 < 1e300) {
        const float rho = (float)(sqrt(e2 / a2) * 1.0001) + 1e-9f;
        atomicMax(out_bits, __float_as_uint(rho));
    }
}

static int finish_rows(Index &ix, int64_t slot0, int64_t n, hipStream_t st) {
    unsigned grid = (unsigned)((n + 255) / 256);
#define LAUNCH(DT) \
    k_row_stats<DT><<<grid, 256, 0, st>>>((const Store<DT>::T *)ix.rows, slot0, n, ix.dim, ix.metric, ix.na, ix.ea, ix.eb)
    if (ix.dtype == AK_DTYPE_F32) LAUNCH(AK_DTYPE_F32);
    else if (ix.dtype == AK_DTYPE_BF16) LAUNCH(AK_DTYPE_BF16);
    else LAUNCH(AK_DTYPE_F16);
#undef LAUNCH
    AK_HIP(hipGetLastError());
    {
        const int64_t blk0 = slot0 / 32, nblk = (slot0 + n + 31) / 32 - blk0;
        k_group_bounds<<<(unsigned)((nblk * 2 + 255) / 256), 256, 0, st>>>(ix.ea, ix.eb, slot0 + n, blk0, nblk, ix.gb);
        AK_HIP(hipGetLastError());
    }
    if (!ix.max_dev) AK_HIP(hipMalloc((void **)&ix.max_dev, 8));
    float *dmax = ix.max_dev;
    AK_HIP(hipMemsetAsync(dmax, 0, 8, st));
    k_max_f32<<<(unsigned)std::min<int64_t>((n + 255) / 256, 1024), 256, 0, st>>>(ix.na, slot0, n, (unsigned int *)dmax);
    if (ix.dtype == AK_DTYPE_F32)
        k_shadow_rho<<<(unsigned)((n + 3) / 4), 256, 0, st>>>((const float *)ix.rows, (const uint16_t *)ix.shadow, slot0, n, ix.dim,
                                                             (unsigned int *)(dmax + 1));
    float hmax[2] = {0.f, 0.f};
    AK_HIP(hipMemcpyAsync(hmax, dmax, 8, hipMemcpyDeviceToHost, st));
    AK_HIP(hipStreamSynchronize(st));
    ix.max_na = std::max(ix.max_na, hmax[0]);
    ix.max_rho = std::max(ix.max_rho, hmax[1]);
    return 0;
}

// ---- compaction / growth ---------------------------------------------------
// dst row i <- src row src[i]: one wave per row, 16-byte pieces (row bytes are a multiple of 2; the tail goes by 2 bytes)
__global__ __launch_bounds__(256) void k_gather_rows(const char *__restrict__ src_rows, const int64_t *__restrict__ src, int64_t m,
                                                     int64_t row_bytes, char *__restrict__ dst_rows) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= m) return;
    const char *s = src_rows + src[i] * row_bytes;
    char *d = dst_rows + i * row_bytes;
    if ((row_bytes & 15) == 0) {
        for (int64_t o = lane * 16; o < row_bytes; o += 64 * 16) *(uint4 *)(d + o) = *(const uint4 *)(s + o);
    } else {
        for (int64_t o = lane * 2; o < row_bytes; o += 64 * 2) *(uint16_t *)(d + o) = *(const uint16_t *)(s + o);
    }
}
__global__ void k_gather_terms(const int64_t *__restrict__ src, int64_t m, const float *__restrict__ na, const float *__restrict__ ea,
                               const float *__restrict__ eb, const int64_t *__restrict__ ids, float *__restrict__ na2,
                               float *__restrict__ ea2, float *__restrict__ eb2, int64_t *__restrict__ ids2, uint8_t *__restrict__ alive2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int64_t s = src[i];
    na2[i] = na[s]; ea2[i] = ea[s]; eb2[i] = eb[s]; ids2[i] = ids[s]; alive2[i] = 1;
}

// ---- re-run of uncertified queries: gather / scatter by query index --------------
__global__ void k_gather_queries(const float *__restrict__ q, const float *__restrict__ nb, const int *__restrict__ idx, int m,
                                 int dim, float *__restrict__ gq, float *__restrict__ gnb) {
    const int j = blockIdx.x;
    if (j >= m) return;
    const int qi = idx[j];
    for (int i = threadIdx.x; i < dim; i += blockDim.x) gq[(int64_t)j * dim + i] = q[(int64_t)qi * dim + i];
    if (threadIdx.x == 0) gnb[j] = nb[qi];
}
// only: NULL or [m] flags -- rows with 0 are left alone
__global__ void k_scatter_results(const int *__restrict__ idx, const int *__restrict__ only, int m, int k,
                                  const int64_t *__restrict__ gi, const double *__restrict__ gd, const int *__restrict__ gc,
                                  int64_t *__restrict__ oi, double *__restrict__ od, int *__restrict__ oc, int *__restrict__ ocert) {
    const int j = blockIdx.x;
    if (j >= m || (only && !only[j])) return;
    const int qi = idx[j];
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        oi[(int64_t)qi * k + i] = gi[(int64_t)j * k + i];
        od[(int64_t)qi * k + i] = gd[(int64_t)j * k + i];
    }
    if (threadIdx.x == 0) {
        if (oc) oc[qi] = gc[j];
        if (ocert) ocert[qi] = 1;
    }
}
__global__ void k_fill_int(int *p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Writers wait (on the host) for the last asynchronous device search: its kernels read rows / ea / eb / gb / ids.
static int writer_fence(Index &ix) {
    std::lock_guard<std::mutex> wl(ix.ws_mu);
    if (ix.ws_pending) {
        AK_HIP(hipEventSynchronize(ix.ws_event));
        ix.ws_pending = false;
    }
    return 0;
}

struct IndexBuffers {
    void *rows = nullptr, *shadow = nullptr;
    float *na = nullptr, *ea = nullptr, *eb = nullptr, *gb = nullptr;
    int64_t *ids = nullptr;
    uint8_t *alive = nullptr;
    void release() {
        if (rows) hipFree(rows);
        if (shadow) hipFree(shadow);
        if (na) hipFree(na);
        if (ea) hipFree(ea);
        if (eb) hipFree(eb);
        if (gb) hipFree(gb);
        if (ids) hipFree(ids);
        if (alive) hipFree(alive);
        *this = IndexBuffers();
    }
};
static hipError_t alloc_buffers(IndexBuffers &b, int64_t capacity, int dim, int dtype) {
    // the 16-bit matrix the MFMA scan reads is allocated in whole 256-row tiles: the phased K-loop (scan.hip) stages the rows
    // of a tail tile past n without clamping (constant per-lane offsets); what it reads there is masked in the epilogue
    const size_t cap_t = ((size_t)capacity + 255) / 256 * 256;
    const size_t rb = (dtype == AK_DTYPE_F32 ? (size_t)capacity : cap_t) * dim * dtype_size(dtype);
    hipError_t e = hipMalloc(&b.rows, rb + 256);
    if (e == hipSuccess && dtype == AK_DTYPE_F32) e = hipMalloc(&b.shadow, cap_t * dim * 2 + 256);
    if (e == hipSuccess) e = hipMalloc((void **)&b.na, capacity * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&b.ea, capacity * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&b.eb, capacity * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&b.gb, ((capacity + 31) / 32) * 16 + 256);
    if (e == hipSuccess) e = hipMalloc((void **)&b.ids, capacity * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&b.alive, capacity);
    if (e != hipSuccess) b.release();
    return e;
}
static void take_buffers(Index &ix, IndexBuffers &b) {   // ix <- b, b <- what ix held
    std::swap(ix.rows, b.rows); std::swap(ix.shadow, b.shadow); std::swap(ix.na, b.na); std::swap(ix.ea, b.ea);
    std::swap(ix.eb, b.eb); std::swap(ix.gb, b.gb); std::swap(ix.ids, b.ids); std::swap(ix.alive, b.alive);
}


// Move the live rows (or, compact == false, all row slots) into buffers of `new_cap` rows. The reference's table has no
// capacity and reclaims dead tuples by itself (autovacuum; VACUUM FULL at reset, manager.py:103-153): re-ingesting a document
// replaces its chunks (ON CONFLICT, postgres_vectorstore.py:168-182 / manager.py:192-211), so a long-running data manager
// would otherwise run an append-only index into "capacity exceeded" with few live rows. Caller holds the unique lock.
static int rebuild(Index &ix, int64_t new_cap, bool compact, hipStream_t st) {
    IndexBuffers nb;
    hipError_t e = alloc_buffers(nb, new_cap, ix.dim, ix.dtype);
    if (e != hipSuccess) AK_FAIL(-10, std::string("index growth / compaction: hipMalloc failed: ") + hipGetErrorString(e));
    const size_t rbytes = (size_t)ix.dim * dtype_size(ix.dtype);
    int rc = 0;
    std::vector<int64_t> src;
    const bool gather = compact && ix.n_alive != ix.n;
    if (!gather) {
        do {
            if (ix.n == 0) break;
            if (hipMemcpyAsync(nb.rows, ix.rows, rbytes * ix.n, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = -10; break; }
            if (ix.shadow && hipMemcpyAsync(nb.shadow, ix.shadow, (size_t)ix.dim * 2 * ix.n, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = -10; break; }
            hipMemcpyAsync(nb.na, ix.na, ix.n * 4, hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(nb.ea, ix.ea, ix.n * 4, hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(nb.eb, ix.eb, ix.n * 4, hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(nb.gb, ix.gb, ((ix.n + 31) / 32) * 16, hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(nb.ids, ix.ids, ix.n * 8, hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(nb.alive, ix.alive, ix.n, hipMemcpyDeviceToDevice, st);
        } while (0);
    } else {
        ix.live_slots(src);
        const int64_t m = (int64_t)src.size();
        int64_t *dsrc = nullptr;
        do {
            if (m == 0) break;
            if (hipMalloc((void **)&dsrc, (size_t)m * 8) != hipSuccess) { rc = -10; break; }
            if (hipMemcpyAsync(dsrc, src.data(), (size_t)m * 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -10; break; }
            k_gather_rows<<<(unsigned)((m + 3) / 4), 256, 0, st>>>((const char *)ix.rows, dsrc, m, (int64_t)rbytes, (char *)nb.rows);
            if (ix.shadow) k_gather_rows<<<(unsigned)((m + 3) / 4), 256, 0, st>>>((const char *)ix.shadow, dsrc, m, (int64_t)ix.dim * 2, (char *)nb.shadow);
            k_gather_terms<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(dsrc, m, ix.na, ix.ea, ix.eb, ix.ids, nb.na, nb.ea, nb.eb, nb.ids, nb.alive);
            const int64_t nblk = (m + 31) / 32;
            k_group_bounds<<<(unsigned)((nblk * 2 + 255) / 256), 256, 0, st>>>(nb.ea, nb.eb, m, 0, nblk, nb.gb);
            if (hipGetLastError() != hipSuccess) { rc = -10; break; }
        } while (0);
        if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = -10;
        if (dsrc) hipFree(dsrc);
    }
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = -10;
    if (rc) { nb.release(); AK_FAIL(-10, "index growth / compaction: device copy failed"); }
    take_buffers(ix, nb);
    nb.release();        // the old buffers
    ix.rebuilt(new_cap, gather ? &src : nullptr);      // the host mirror follows (new slot numbers, layout epoch)
    return 0;
}

// Make room for `add` more rows: reclaim tombstones when that frees a useful share, otherwise (or also) double the buffers.
static int ensure_room(Index &ix, int64_t add, hipStream_t st) {
    IndexBook::RoomPlan plan;
    std::string err;
    if (int rc = ix.plan_room(add, plan, err)) AK_FAIL(rc, err);
    if (plan.what == IndexBook::FITS) return 0;
    if (writer_fence(ix)) return -10;
    return rebuild(ix, plan.new_cap, plan.compact, st);
}

}  // namespace ak

using namespace ak;

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char *ak_last_error(void) { return g_err.c_str(); }
const char *ak_version(void) { return "archi_hip 0.3 (gfx950)"; }
int ak_abi_version(void) { return AK_ABI_VERSION; }
int ak_debug_set(const char *name, const char *value) {
    if (switches_set(name, value)) {
        set_error(std::string("ak_debug_set: unknown switch ") + (name ? name : "(null)") +
                  (DBG_KERNELS ? "" : " (stage-skipping switches exist only in libarchi_hip_dbg.so)"));
        return -1;
    }
    return 0;
}

int ak_init(int device) {
    int cnt = 0;
    AK_HIP(hipGetDeviceCount(&cnt));
    if (cnt <= 0) AK_FAIL(-2, "no HIP device visible: libarchi_hip has no CPU fallback");
    if (device < 0 || device >= cnt) AK_FAIL(-3, "ak_init: device index out of range");
    AK_HIP(hipSetDevice(device));
    hipDeviceProp_t p;
    AK_HIP(hipGetDeviceProperties(&p, device));
    if (std::string(p.gcnArchName).rfind("gfx950", 0) != 0)
        AK_FAIL(-4, std::string("libarchi_hip is built for gfx950 only, found ") + p.gcnArchName);
    g_device.store(device);
    t_bound_device = device;
    return 0;
}

int ak_device_info(char *name_out, int name_cap, int *cu_count, int64_t *hbm_bytes) {
    int dev = 0;
    AK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    AK_HIP(hipGetDeviceProperties(&p, dev));
    if (name_out && name_cap > 0) { strncpy(name_out, p.name, name_cap - 1); name_out[name_cap - 1] = 0; }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return 0;
}

int ak_sync(void *stream) {
    AK_BIND();
    AK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int ak_index_create(int64_t capacity, int dim, int dtype, int metric, ak_index_t *out) {
    AK_BIND();
    if (!out) AK_FAIL(-1, "ak_index_create: out is NULL");
    if (capacity <= 0 || capacity > IndexBook::CAP_MAX) AK_FAIL(-1, "ak_index_create: capacity must be in (0, 2^32)");
    if (dim <= 0 || dim > 65536) AK_FAIL(-1, "ak_index_create: bad dim");
    if (dtype < 0 || dtype > 2) AK_FAIL(-1, "ak_index_create: dtype must be AK_DTYPE_F32/BF16/F16");
    if (metric < 0 || metric > 2) AK_FAIL(-1, "ak_index_create: metric must be AK_METRIC_COSINE/L2/IP");
    Index *ix = new Index();
    ix->dim = dim; ix->dtype = dtype; ix->metric = metric; ix->cap = capacity;
    IndexBuffers b;
    hipError_t e = alloc_buffers(b, capacity, dim, dtype);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ws_event, hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error(std::string("ak_index_create: hipMalloc failed: ") + hipGetErrorString(e));
        b.release();
        delete ix;
        return -10;
    }
    take_buffers(*ix, b);
    *out = ix;
    return 0;
}

int ak_index_destroy(ak_index_t h) {
    AK_BIND();
    if (!h) return 0;
    Index *ix = (Index *)h;
    hipDeviceSynchronize();
    IndexBuffers b;
    take_buffers(*ix, b);
    b.release();
    ix->ws_dev.release();
    ix->ws_fb.release();
    if (ix->ws_event) hipEventDestroy(ix->ws_event);
    if (ix->dbg_dev) hipFree(ix->dbg_dev);
    if (switches().coalesce_stats.load(std::memory_order_relaxed) && ix->co.n_launch)
        fprintf(stderr, "ak_index_search coalescing: %lld requests in %lld launches (%.1f per launch), %lld gather waits\n",
                (long long)ix->co.n_req, (long long)ix->co.n_launch, (double)ix->co.n_req / ix->co.n_launch, (long long)ix->co.n_wait);
    if (ix->max_dev) hipFree(ix->max_dev);
    for (auto &e : ix->prof_events) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    delete ix;
    return 0;
}

static int64_t slot_of(Index &ix, int64_t id) { return ix.slot_of(id); }

int ak_index_add(ak_index_t h, const float *rows, int is_device, int64_t n, const int64_t *ids, int normalise) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_add: NULL index");
    Index &ix = *(Index *)h;
    if (n == 0) return 0;
    if (n < 0 || !rows) AK_FAIL(-1, "ak_index_add: bad arguments");
    std::unique_lock<std::shared_mutex> lk(ix.mu);
    if (ids) {
        std::string err;
        if (int rc = ix.check_new_ids(ids, n, err)) AK_FAIL(rc, "ak_index_add: " + err);
    }
    hipStream_t st;
    if (thread_stream(&st)) return -10;
    if (writer_fence(ix)) return -10;
    if (int rc = ensure_room(ix, n, st)) return rc;
    const int64_t CH = std::max<int64_t>(1, (64ll << 20) / ((int64_t)ix.dim * 4));  // 64 MiB staging
    // staging from the thread's grow-only scratch: per-file ingestion calls this with a few dozen rows at a time, and a
    // hipMalloc + hipFree pair per call (hipFree synchronises the device) would cap ingestion near 2k files/s
    float *stage = nullptr, *nrm = nullptr;
    if (!is_device) {
        if (scratch_reserve(&t_ctx.dev, &t_ctx.dev_cap, (size_t)std::min(CH, n) * ix.dim * 4, false)) return -10;
        stage = (float *)t_ctx.dev;
    }
    if (normalise) {
        if (scratch_reserve(&t_ctx.ws, &t_ctx.ws_cap, (size_t)std::min(CH, n) * 4, false)) return -10;
        nrm = (float *)t_ctx.ws;
    }
    int rc = 0;
    for (int64_t o = 0; o < n && rc == 0; o += CH) {
        int64_t c = std::min(CH, n - o);
        const float *src = rows + o * ix.dim;
        if (!is_device) {
            if (hipMemcpyAsync(stage, src, (size_t)c * ix.dim * 4, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -10; set_error("ak_index_add: H2D failed"); break; }
            src = stage;
        }
        if (normalise) k_row_invnorm<<<(unsigned)((c + 3) / 4), 256, 0, st>>>(src, c, ix.dim, nrm);
        int64_t total = c * ix.dim;
        unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
        size_t off = (size_t)(ix.n + o) * ix.dim;
        if (ix.dtype == AK_DTYPE_F32) {
            k_convert<AK_DTYPE_F32><<<grid, 256, 0, st>>>(src, nrm, total, ix.dim, (float *)ix.rows + off);
            k_convert<AK_DTYPE_BF16><<<grid, 256, 0, st>>>(src, nrm, total, ix.dim, (uint16_t *)ix.shadow + off);
        }
        else if (ix.dtype == AK_DTYPE_BF16) k_convert<AK_DTYPE_BF16><<<grid, 256, 0, st>>>(src, nrm, total, ix.dim, (uint16_t *)ix.rows + off);
        else k_convert<AK_DTYPE_F16><<<grid, 256, 0, st>>>(src, nrm, total, ix.dim, (uint16_t *)ix.rows + off);
        if (hipStreamSynchronize(st) != hipSuccess) { rc = -10; set_error("ak_index_add: convert failed"); }
    }
    if (rc) { t_ctx.trim(); return rc; }
    // ids + alive
    std::vector<int64_t> tmp;
    const int64_t *hid = ids;
    if (!ids) {
        tmp.resize(n);
        for (int64_t i = 0; i < n; i++) tmp[i] = ix.next_id + i;
        hid = tmp.data();
    }
    AK_HIP(hipMemcpyAsync(ix.ids + ix.n, hid, n * 8, hipMemcpyHostToDevice, st));
    AK_HIP(hipMemsetAsync(ix.alive + ix.n, 1, n, st));
    if (finish_rows(ix, ix.n, n, st)) return -10;
    ix.appended(ids, n);
    return 0;
}

int ak_index_generate(ak_index_t h, uint64_t seed, uint32_t stream, uint64_t row0, int64_t n, int normalise,
                      int64_t id0) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_generate: NULL index");
    Index &ix = *(Index *)h;
    if (n <= 0) return 0;
    std::unique_lock<std::shared_mutex> lk(ix.mu);
    if (id0 < 0) AK_FAIL(-1, "ak_index_generate: id0 must be >= 0");
    hipStream_t st;
    if (thread_stream(&st)) return -10;
    if (writer_fence(ix)) return -10;
    if (int rc = ensure_room(ix, n, st)) return rc;
    unsigned grid = (unsigned)((n + 3) / 4);
    if (ix.dtype == AK_DTYPE_F32) {
        k_generate<AK_DTYPE_F32><<<grid, 256, 0, st>>>((float *)ix.rows, ix.n, n, ix.dim, seed, stream, row0, normalise);
        k_generate<AK_DTYPE_BF16><<<grid, 256, 0, st>>>((uint16_t *)ix.shadow, ix.n, n, ix.dim, seed, stream, row0, normalise);
    }
    else if (ix.dtype == AK_DTYPE_BF16) k_generate<AK_DTYPE_BF16><<<grid, 256, 0, st>>>((uint16_t *)ix.rows, ix.n, n, ix.dim, seed, stream, row0, normalise);
    else k_generate<AK_DTYPE_F16><<<grid, 256, 0, st>>>((uint16_t *)ix.rows, ix.n, n, ix.dim, seed, stream, row0, normalise);
    AK_HIP(hipGetLastError());
    k_fill_ids<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(ix.ids, ix.alive, ix.n, n, id0);
    AK_HIP(hipGetLastError());
    if (finish_rows(ix, ix.n, n, st)) return -10;
    ix.appended_generated(id0, n);      // the id map is built lazily for generated rows (10M+ entries): IndexBook::slot_of
    return 0;
}

int ak_index_remove(ak_index_t h, const int64_t *ids, int64_t n, int64_t *n_removed) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_remove: NULL index");
    Index &ix = *(Index *)h;
    if (n_removed) *n_removed = 0;
    if (n <= 0) return 0;
    if (!ids) AK_FAIL(-1, "ak_index_remove: ids is NULL");
    std::unique_lock<std::shared_mutex> lk(ix.mu);
    // pass 1 resolves, nothing is touched: if the fence, the scratch or the kernel fails the host mirror (h_alive, id2slot,
    // n_alive) still agrees with what the device holds
    std::vector<int64_t> slots, live_ids;
    ix.resolve_remove(ids, n, slots, live_ids);
    if (slots.empty()) return 0;
    hipStream_t st;
    if (thread_stream(&st)) return -10;
    if (writer_fence(ix)) return -10;
    if (scratch_reserve(&t_ctx.dev, &t_ctx.dev_cap, slots.size() * 8, false)) return -10;   // thread scratch: no hipMalloc / hipFree per delete
    int64_t *d = (int64_t *)t_ctx.dev;
    AK_HIP(hipMemcpyAsync(d, slots.data(), slots.size() * 8, hipMemcpyHostToDevice, st));
    k_kill<<<(unsigned)((slots.size() + 255) / 256), 256, 0, st>>>(d, (int64_t)slots.size(), ix.alive, ix.ea, ix.eb);
    AK_HIP(hipStreamSynchronize(st));
    ix.removed(slots, live_ids);
    if (n_removed) *n_removed = (int64_t)slots.size();
    t_ctx.trim();
    return 0;
}

int ak_index_count(ak_index_t h, int64_t *out) {
    if (!h || !out) AK_FAIL(-1, "ak_index_count: NULL argument");
    Index &ix = *(Index *)h;
    std::shared_lock<std::shared_mutex> lk(ix.mu);
    *out = ix.n_alive;
    return 0;
}

int ak_index_slots(ak_index_t h, int64_t *out_slots, int64_t *out_capacity, uint64_t *out_epoch) {
    if (!h) AK_FAIL(-1, "ak_index_slots: NULL index");
    Index &ix = *(Index *)h;
    std::shared_lock<std::shared_mutex> lk(ix.mu);
    if (out_slots) *out_slots = ix.n;
    if (out_capacity) *out_capacity = ix.cap;
    if (out_epoch) *out_epoch = ix.epoch;
    return 0;
}

int ak_index_compact(ak_index_t h, int64_t *n_reclaimed) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_compact: NULL index");
    Index &ix = *(Index *)h;
    std::unique_lock<std::shared_mutex> lk(ix.mu);
    const int64_t dead = ix.n - ix.n_alive;
    if (n_reclaimed) *n_reclaimed = dead;
    if (dead == 0) return 0;
    hipStream_t st;
    if (thread_stream(&st)) return -10;
    if (writer_fence(ix)) return -10;
    return rebuild(ix, ix.cap, true, st);
}

int ak_index_lookup(ak_index_t h, const int64_t *ids, int64_t n, int64_t *out_slots) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_lookup: NULL index");
    Index &ix = *(Index *)h;
    std::unique_lock<std::shared_mutex> lk(ix.mu);  // may build the lazy map
    for (int64_t i = 0; i < n; i++) {
        int64_t s = slot_of(ix, ids[i]);
        out_slots[i] = (s >= 0 && ix.h_alive[s]) ? s : -1;
    }
    return 0;
}

int ak_index_fetch(ak_index_t h, const int64_t *row_slots, int64_t n, float *out_host) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_fetch: NULL index");
    Index &ix = *(Index *)h;
    if (n <= 0) return 0;
    std::shared_lock<std::shared_mutex> lk(ix.mu);
    for (int64_t i = 0; i < n; i++)
        if (row_slots[i] < 0 || row_slots[i] >= ix.n) AK_FAIL(-1, "ak_index_fetch: slot out of range");
    hipStream_t st;
    if (thread_stream(&st)) return -10;
    // slots | rows from the thread scratch, in pieces of <= 64 MiB of output
    const int64_t CH = std::max<int64_t>(1, (64ll << 20) / ((int64_t)ix.dim * 4));
    const size_t sb = ((size_t)std::min(CH, n) * 8 + 255) & ~255ull;
    if (scratch_reserve(&t_ctx.dev, &t_ctx.dev_cap, sb + (size_t)std::min(CH, n) * ix.dim * 4, false)) return -10;
    int64_t *ds = (int64_t *)t_ctx.dev;
    float *dout = (float *)(t_ctx.dev + sb);
    for (int64_t o = 0; o < n; o += CH) {
        const int64_t c = std::min(CH, n - o);
        AK_HIP(hipMemcpyAsync(ds, row_slots + o, c * 8, hipMemcpyHostToDevice, st));
        unsigned grid = (unsigned)((c * ix.dim + 255) / 256);
        if (ix.dtype == AK_DTYPE_F32) k_fetch<AK_DTYPE_F32><<<grid, 256, 0, st>>>((const float *)ix.rows, ds, c, ix.dim, dout);
        else if (ix.dtype == AK_DTYPE_BF16) k_fetch<AK_DTYPE_BF16><<<grid, 256, 0, st>>>((const uint16_t *)ix.rows, ds, c, ix.dim, dout);
        else k_fetch<AK_DTYPE_F16><<<grid, 256, 0, st>>>((const uint16_t *)ix.rows, ds, c, ix.dim, dout);
        AK_HIP(hipMemcpyAsync(out_host + o * ix.dim, dout, (size_t)c * ix.dim * 4, hipMemcpyDeviceToHost, st));
        AK_HIP(hipStreamSynchronize(st));
    }
    t_ctx.trim();
    return 0;
}

// Exact distances of one query to listed rows (the semantic leg of the hybrid combine, which needs the
// score of every BM25 hit, not just the top-k): same arithmetic as the search's re-rank.
int ak_index_distances(ak_index_t h, const float *query, const int64_t *ids, int64_t n, double *out_dist,
                       uint8_t *out_found) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_distances: NULL index");
    Index &ix = *(Index *)h;
    if (n <= 0) return 0;
    if (!query || !ids || !out_dist) AK_FAIL(-1, "ak_index_distances: bad arguments");
    if (n > (1 << 30)) AK_FAIL(-1, "ak_index_distances: too many ids");
    std::vector<uint64_t> cand((size_t)n);
    {
        std::unique_lock<std::shared_mutex> lk(ix.mu);  // may build the lazy id map
        for (int64_t i = 0; i < n; i++) {
            int64_t s = slot_of(ix, ids[i]);
            bool ok = s >= 0 && ix.h_alive[s];
            cand[i] = ok ? (uint64_t)s : KEY_INVALID;
            if (out_found) out_found[i] = ok ? 1 : 0;
        }
    }
    std::shared_lock<std::shared_mutex> lk(ix.mu);
    hipStream_t st;
    if (thread_stream(&st)) return -10;
    const size_t qb = ((size_t)ix.dim * 4 + 255) & ~255ull, cb = ((size_t)n * 8 + 255) & ~255ull;
    // query | nb | cand | okeys | oids from the thread scratch (a hipMalloc + hipFree pair per hybrid query synchronised the device)
    if (scratch_reserve(&t_ctx.dev, &t_ctx.dev_cap, qb + 256 + 3 * cb, false)) return -10;
    char *blk = t_ctx.dev;
    float *dq = (float *)blk, *dnb = (float *)(blk + qb);
    uint64_t *dc = (uint64_t *)(blk + qb + 256), *dk = (uint64_t *)(blk + qb + 256 + cb);
    int64_t *di = (int64_t *)(blk + qb + 256 + 2 * cb);
    int rc = 0;
    do {
        if (hipMemcpyAsync(dq, query, (size_t)ix.dim * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(dc, cand.data(), (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -10; break; }
        if ((rc = query_norms(dq, 1, ix.dim, dnb, st))) break;
        if ((rc = rerank(ix, dq, dnb, 1, (int)n, dc, dk, di, st))) break;
        if (hipMemcpyAsync(cand.data(), dk, (size_t)n * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = -10; break; }
    } while (0);
    if (rc) hipStreamSynchronize(st);
    t_ctx.trim();
    if (rc == -10) AK_FAIL(-10, "ak_index_distances: HIP error");
    if (rc) return rc;
    for (int64_t i = 0; i < n; i++) out_dist[i] = cand[i] == KEY_INVALID ? __builtin_nan("") : key_dist(cand[i]);
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------
namespace ak {

struct RerunStats { int64_t second_certified = 0, exact = 0; };

// Queries the first MFMA scan could not certify (or, fast_ran == false, all of them): a second scan with the widest
// candidate lists certifies what a wider list can fix -- more equal scores around the k-th place than k' holds, i.e.
// duplicated chunks -- for the price of one more scan; what is still open goes through the exact path (reference
// arithmetic over every row). Results are scattered into the caller's [nq][k] arrays by query index; cert (nullable)
// gets 1 for every re-run query. Everything is enqueued on st; the only host synchronisation is the read of the second
// scan's certificate flags. reserve(bytes) returns a device buffer that stays valid until its next call.
template <class Reserve>
static int rerun_uncertified(Index &ix, const float *dq, const float *dnb, int nq, int k, const uint8_t *dfl, int64_t *doi,
                             double *dod, int *dct, int *dce, std::vector<int> todo, bool fast_ran, int first_kprime,
                             Reserve reserve, hipStream_t st, RerunStats *rs) {
    if (todo.empty()) return 0;
    int m = (int)todo.size();
    bool second = fast_ran && fast_supported(ix, m, k);
    FastPlan p2;
    size_t sub = exact_scratch_bytes(ix, k);
    if (second) {
        p2 = fast_plan(ix, m, k, true);
        if (p2.kprime <= first_kprime) second = false;
        else sub = std::max(sub, p2.bytes);
    }
    if (!second && m == nq) {      // everything goes through the exact path: no gather
        void *ws = reserve(sub);
        if (!ws) return -10;
        if (int rc = exact_search(ix, dq, dnb, nq, k, dfl, doi, dod, dct, ws, st)) return rc;
        if (dce) k_fill_int<<<(nq + 255) / 256, 256, 0, st>>>(dce, nq, 1);
        AK_HIP(hipGetLastError());
        if (rs) rs->exact += nq;
        return 0;
    }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_idx = 0, o_q = o_idx + al((size_t)m * 4), o_nb = o_q + al((size_t)m * ix.dim * 4), o_i = o_nb + al((size_t)m * 4),
                 o_d = o_i + al((size_t)m * k * 8), o_c = o_d + al((size_t)m * k * 8), o_ce = o_c + al((size_t)m * 4),
                 o_sub = o_ce + al((size_t)m * 4);
    char *base = (char *)reserve(o_sub + sub);
    if (!base) return -10;
    int *didx = (int *)(base + o_idx);
    float *gq = (float *)(base + o_q), *gnb = (float *)(base + o_nb);
    int64_t *gi = (int64_t *)(base + o_i);
    double *gd = (double *)(base + o_d);
    int *gc = (int *)(base + o_c), *gce = (int *)(base + o_ce);
    void *subws = base + o_sub;
    AK_HIP(hipMemcpyAsync(didx, todo.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    k_gather_queries<<<m, 128, 0, st>>>(dq, dnb, didx, m, ix.dim, gq, gnb);
    AK_HIP(hipGetLastError());
    if (second) {
        if (int rc = fast_search(ix, gq, gnb, true, m, k, dfl, gi, gd, gc, gce, nullptr, subws, p2, st)) return rc;
        std::vector<int> c2(m, 0);
        k_scatter_results<<<m, 64, 0, st>>>(didx, gce, m, k, gi, gd, gc, doi, dod, dct, dce);
        AK_HIP(hipGetLastError());
        AK_HIP(hipMemcpyAsync(c2.data(), gce, (size_t)m * 4, hipMemcpyDeviceToHost, st));
        AK_HIP(hipStreamSynchronize(st));
        std::vector<int> still;
        for (int j = 0; j < m; j++) if (!c2[j]) still.push_back(todo[j]);
        if (rs) rs->second_certified += m - (int64_t)still.size();
        if (still.empty()) return 0;
        todo.swap(still);
        m = (int)todo.size();
        // the gather buffers were sized for the longer list: fine
        AK_HIP(hipMemcpyAsync(didx, todo.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
        k_gather_queries<<<m, 128, 0, st>>>(dq, dnb, didx, m, ix.dim, gq, gnb);
        AK_HIP(hipGetLastError());
    }
    if (int rc = exact_search(ix, gq, gnb, m, k, dfl, gi, gd, gc, subws, st)) return rc;
    k_scatter_results<<<m, 64, 0, st>>>(didx, nullptr, m, k, gi, gd, gc, doi, dod, dct, dce);
    AK_HIP(hipGetLastError());
    // `todo` (pageable) was the source of an asynchronous copy: it must outlive it
    AK_HIP(hipStreamSynchronize(st));
    if (rs) rs->exact += m;
    return 0;
}

}  // namespace ak

extern "C" {

// a row_filter is a statement about ONE layout of the index: the caller says which (ak_index_slots), and a mask of another
// layout is refused before a byte of it is read. Caller holds the shared lock.
static int filter_is_current(const Index &ix, const void *row_filter, int64_t filter_len, uint64_t filter_epoch, const char *who) {
    if (!row_filter) return 0;
    if (!ix.filter_matches(filter_len, filter_epoch))
        AK_FAIL(AK_ERR_STALE_FILTER, std::string(who) + ": stale row_filter (built for " + std::to_string(filter_len) + " slots at layout epoch " +
                                         std::to_string(filter_epoch) + ", the index has " + std::to_string(ix.n) + " at epoch " +
                                         std::to_string(ix.epoch) + "): rebuild the mask from ak_index_slots / ak_index_lookup and retry");
    return 0;
}

static int search_host(Index &ix, const float *queries, int nq, int k, int mode, const uint8_t *row_filter, int64_t filter_len,
                       uint64_t filter_epoch, int64_t *out_ids, double *out_dist, int *out_counts, int64_t *out_stats) {
    std::shared_lock<std::shared_mutex> lk(ix.mu);
    if (int frc = filter_is_current(ix, row_filter, filter_len, filter_epoch, "ak_index_search")) return frc;
    hipStream_t st;
    if (thread_stream(&st)) return -10;
    const size_t qb = (size_t)nq * ix.dim * 4, ob = (size_t)nq * k * 8;
    char *blk = nullptr;  // one allocation: queries | nb | out_ids | out_dist | cnt | cert | stats | filter
    size_t off_nb = (qb + 255) & ~255ull, off_oi = off_nb + (((size_t)nq * 4 + 255) & ~255ull),
           off_od = off_oi + ((ob + 255) & ~255ull), off_ct = off_od + ((ob + 255) & ~255ull),
           off_ce = off_ct + (((size_t)nq * 4 + 255) & ~255ull), off_st = off_ce + (((size_t)nq * 4 + 255) & ~255ull),
           off_fl = off_st + 256, total = off_fl + (row_filter ? (size_t)ix.n + 256 : 0);
    if (scratch_reserve(&t_ctx.dev, &t_ctx.dev_cap, total, false)) return -10;
    blk = t_ctx.dev;
    // results come back in ONE device-to-host copy of [ids | dist | counts | certified | stats] into pinned memory
    const size_t out_bytes = off_fl - off_oi, pin_q = qb <= (1u << 20) ? ((qb + 255) & ~255ull) : 0;
    if (scratch_reserve(&t_ctx.pin, &t_ctx.pin_cap, pin_q + out_bytes, true)) return -10;
    char *pin_out = t_ctx.pin + pin_q;
    float *dq = (float *)blk, *dnb = (float *)(blk + off_nb);
    int64_t *doi = (int64_t *)(blk + off_oi);
    double *dod = (double *)(blk + off_od);
    int *dct = (int *)(blk + off_ct), *dce = (int *)(blk + off_ce);
    int64_t *dst = (int64_t *)(blk + off_st);
    uint8_t *dfl = row_filter ? (uint8_t *)(blk + off_fl) : nullptr;
    int rc = 0;
    do {
        const void *qsrc = queries;
        if (pin_q) { memcpy(t_ctx.pin, queries, qb); qsrc = t_ctx.pin; }      // small batches: a truly asynchronous H2D
        if (hipMemcpyAsync(dq, qsrc, qb, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -10; set_error("ak_index_search: H2D failed"); break; }
        if (dfl && ix.n > 0 && hipMemcpyAsync(dfl, row_filter, (size_t)ix.n, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -10; set_error("ak_index_search: filter H2D failed"); break; }
        const bool fast = mode != AK_SEARCH_EXACT && fast_supported(ix, nq, k);
        if (!fast && (rc = query_norms(dq, nq, ix.dim, dnb, st))) break;      // the fast path computes them in its set-up launch
        std::vector<int> todo;
        int first_kprime = 0;
        if (fast) {
            FastPlan plan = fast_plan(ix, nq, k);
            first_kprime = plan.kprime;
            if (scratch_reserve(&t_ctx.ws, &t_ctx.ws_cap, plan.bytes, false)) { rc = -10; break; }
            if ((rc = fast_search(ix, dq, dnb, false, nq, k, dfl, doi, dod, dct, dce, dst, t_ctx.ws, plan, st))) break;
            if (hipMemcpyAsync(pin_out, blk + off_oi, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = -10; break; }
            if (hipStreamSynchronize(st) != hipSuccess) { rc = -10; set_error(std::string("fast_search failed: ") + hipGetErrorString(hipGetLastError())); break; }
            const int *cert = (const int *)(pin_out + (off_ce - off_oi));
            if (out_stats) memcpy(out_stats, pin_out + (off_st - off_oi), 32);
            for (int i = 0; i < nq; i++) if (!cert[i]) todo.push_back(i);
            if (out_stats) { out_stats[0] = nq - (int64_t)todo.size(); out_stats[1] = (int64_t)todo.size(); }
            if (todo.empty() || mode == AK_SEARCH_FAST_ONLY) {        // the common case ends here: one copy, one synchronisation
                memcpy(out_ids, pin_out, ob);
                memcpy(out_dist, pin_out + (off_od - off_oi), ob);
                if (out_counts) memcpy(out_counts, pin_out + (off_ct - off_oi), (size_t)nq * 4);
                break;
            }
        } else {
            for (int i = 0; i < nq; i++) todo.push_back(i);
        }
        // the first scan's workspace is no longer needed (its results sit in `blk`): the re-run takes the same thread scratch
        RerunStats rs;
        auto reserve = [&](size_t bytes) -> void * { return scratch_reserve(&t_ctx.ws, &t_ctx.ws_cap, bytes, false) ? nullptr : t_ctx.ws; };
        if ((rc = rerun_uncertified(ix, dq, dnb, nq, k, dfl, doi, dod, dct, nullptr, todo, fast, first_kprime, reserve, st, &rs))) break;
        if (out_stats && fast) { out_stats[0] += rs.second_certified; out_stats[1] = rs.exact; out_stats[3] = rs.second_certified; }
        hipMemcpyAsync(out_ids, doi, ob, hipMemcpyDeviceToHost, st);
        hipMemcpyAsync(out_dist, dod, ob, hipMemcpyDeviceToHost, st);
        if (out_counts) hipMemcpyAsync(out_counts, dct, (size_t)nq * 4, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) { rc = -10; set_error(std::string("ak_index_search: ") + hipGetErrorString(hipGetLastError())); }
    } while (0);
    if (rc) hipStreamSynchronize(st);      // nothing of a failed call may still be running on the cached buffers
    t_ctx.trim();
    return rc;
}

// Request coalescing on the read path: coalesce.h (host-only; the queueing, leader promotion and grouping live there and run
// under ThreadSanitizer / AddressSanitizer in the CPU suite). Here: what a group's search is. AK_COALESCE=0 turns it off.
namespace ak {
constexpr int COALESCE_MAX_NQ = 16;
static bool coalesce_enabled() {
    static const bool v = !(env_get("AK_COALESCE") && atoi(env_get("AK_COALESCE")) == 0);
    return v;
}
static void run_group(Index &ix, std::vector<SearchReq *> &g) {
    if (g.size() == 1) {
        SearchReq &r = *g[0];
        r.rc = search_host(ix, r.q, r.nq, r.k, r.mode, r.filter, r.flen, r.fepoch, r.out_ids, r.out_dist, r.out_counts, r.out_stats);
        if (r.rc) r.err = g_err;
        return;
    }
    const int k = g[0]->k, dim = ix.dim;
    int total = 0;
    for (auto *r : g) total += r->nq;
    static thread_local std::vector<float> q;
    static thread_local std::vector<int64_t> oi;
    static thread_local std::vector<double> od;
    static thread_local std::vector<int> oc;
    q.resize((size_t)total * dim); oi.resize((size_t)total * k); od.resize((size_t)total * k); oc.resize(total);
    int o = 0;
    for (auto *r : g) { memcpy(q.data() + (size_t)o * dim, r->q, (size_t)r->nq * dim * 4); o += r->nq; }
    int64_t stats[4] = {0, 0, 0, 0};
    const int rc = search_host(ix, q.data(), total, k, g[0]->mode, g[0]->filter, g[0]->flen, g[0]->fepoch, oi.data(), od.data(), oc.data(), stats);
    o = 0;
    for (auto *r : g) {
        r->rc = rc;
        if (rc) r->err = g_err;
        else {
            memcpy(r->out_ids, oi.data() + (size_t)o * k, (size_t)r->nq * k * 8);
            memcpy(r->out_dist, od.data() + (size_t)o * k, (size_t)r->nq * k * 8);
            if (r->out_counts) memcpy(r->out_counts, oc.data() + o, (size_t)r->nq * 4);
            if (r->out_stats) memcpy(r->out_stats, stats, sizeof(stats));       // of the coalesced launch, not of this request alone
        }
        o += r->nq;
    }
}
}  // namespace ak

int ak_index_search(ak_index_t h, const float *queries, int nq, int k, int mode, const uint8_t *row_filter, int64_t filter_len,
                    uint64_t filter_epoch, int64_t *out_ids, double *out_dist, int *out_counts, int64_t *out_stats) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_search: NULL index");
    Index &ix = *(Index *)h;
    if (out_stats) memset(out_stats, 0, 4 * sizeof(int64_t));
    if (nq == 0) return 0;
    if (nq < 0 || k <= 0 || !queries || !out_ids || !out_dist) AK_FAIL(-1, "ak_index_search: bad arguments");
    if (k > 4096) AK_FAIL(-1, "ak_index_search: k > 4096 not supported");
    RoctxRange range("ak_index_search");
    if (nq > COALESCE_MAX_NQ || !coalesce_enabled())
        return search_host(ix, queries, nq, k, mode, row_filter, filter_len, filter_epoch, out_ids, out_dist, out_counts, out_stats);
    SearchReq me{queries, nq, k, mode, row_filter, filter_len, filter_epoch, out_ids, out_dist, out_counts, out_stats};
    // AK_COALESCE_WINDOW_US > 0 makes a promoted leader wait that long for as many callers as the last launch served.
    // Measured (1M x 384 f32, Python request threads): 16 threads 25.2 k q/s without a window, 19.6 k with 100 us; 32 threads
    // 24.1 k / 23.2 k -- the interpreter lock, not the launch count, is the limit there (43 us per request, of which the
    // GPU's share is 20) -- so the default is 0: nobody ever waits for company. The first caller on an idle index never does.
    static const int window_us = env_get("AK_COALESCE_WINDOW_US") ? atoi(env_get("AK_COALESCE_WINDOW_US")) : 0;
    const int rc = ix.co.submit(me, [&](std::vector<SearchReq *> &g) { run_group(ix, g); }, window_us);
    if (rc) set_error(me.err);
    return rc;
}

int ak_index_search_dev(ak_index_t h, const float *queries_dev, int nq, int k, int mode, const uint8_t *row_filter_dev,
                        int64_t filter_len, uint64_t filter_epoch, int64_t *out_ids_dev, double *out_dist_dev, int *out_cert_dev,
                        void *stream) {
    AK_BIND();
    if (!h) AK_FAIL(-1, "ak_index_search_dev: NULL index");
    Index &ix = *(Index *)h;
    if (nq <= 0) return 0;
    if (k <= 0 || k > 4096) AK_FAIL(-1, "ak_index_search_dev: bad k");
    if (!queries_dev || !out_ids_dev || !out_dist_dev) AK_FAIL(-1, "ak_index_search_dev: bad arguments");
    if (mode < AK_SEARCH_AUTO || mode > AK_SEARCH_FAST_ONLY) AK_FAIL(-1, "ak_index_search_dev: bad mode");
    RoctxRange range("ak_index_search_dev");
    std::shared_lock<std::shared_mutex> lk(ix.mu);
    if (int frc = filter_is_current(ix, row_filter_dev, filter_len, filter_epoch, "ak_index_search_dev")) return frc;
    std::lock_guard<std::mutex> wl(ix.ws_mu);
    hipStream_t st = (hipStream_t)stream;
    // the previous call's kernels may still be running out of ws_dev on another stream
    if (ix.ws_pending && ix.ws_stream != st) AK_HIP(hipStreamWaitEvent(st, ix.ws_event, 0));
    const bool fast = mode != AK_SEARCH_EXACT && fast_supported(ix, nq, k);
    FastPlan plan;
    size_t body = exact_scratch_bytes(ix, k);
    if (fast) { plan = fast_plan(ix, nq, k); body = plan.bytes; }
    const size_t a4 = ((size_t)nq * 4 + 255) & ~255ull;
    if (ix.ws_dev.reserve(3 * a4 + 256 + body)) return -10;
    char *p = (char *)ix.ws_dev.buf;
    float *dnb = (float *)p; p += a4;
    int *dct = (int *)p; p += a4;
    int *dce = out_cert_dev ? out_cert_dev : (int *)p; p += a4;
    int64_t *dst = (int64_t *)p; p += 256;
    int rc = 0;
    auto mark = [&]() -> int {
        AK_HIP(hipEventRecord(ix.ws_event, st));
        ix.ws_pending = true; ix.ws_stream = st;
        return 0;
    };
    // whatever path leaves this function -- an error in the middle of AUTO included -- kernels of this call may still be
    // running out of ws_dev / reading rows, ea, eb: writers (add / remove / compaction) must find the fence set
    struct Guard {
        decltype(mark) &m; bool armed = true;
        ~Guard() { if (armed) m(); }
    } guard{mark};
    if (!fast) {
        // shapes the MFMA scan does not take (fewer than 4096 rows, dim % 64 != 0, k > 128) and EXACT mode: reference
        // arithmetic over every row -- exact by construction, so every query counts as certified
        if ((rc = query_norms(queries_dev, nq, ix.dim, dnb, st))) return rc;
        if ((rc = exact_search(ix, queries_dev, dnb, nq, k, row_filter_dev, out_ids_dev, out_dist_dev, dct, p, st))) return rc;
        k_fill_int<<<(nq + 255) / 256, 256, 0, st>>>(dce, nq, 1);
        AK_HIP(hipGetLastError());
        return 0;             // the guard records the fence
    }
    if ((rc = fast_search(ix, queries_dev, dnb, false, nq, k, row_filter_dev, out_ids_dev, out_dist_dev, dct, dce, dst, p, plan, st))) return rc;
    if (mode == AK_SEARCH_FAST_ONLY) return 0;
    // AUTO: read the certificate flags, re-run what is open
    std::vector<int> cert(nq, 0), todo;
    AK_HIP(hipMemcpyAsync(cert.data(), dce, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    AK_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < nq; i++) if (!cert[i]) todo.push_back(i);
    if (!todo.empty()) {
        auto reserve = [&](size_t bytes) -> void * { return ix.ws_fb.reserve(bytes) ? nullptr : ix.ws_fb.buf; };
        if ((rc = rerun_uncertified(ix, queries_dev, dnb, nq, k, row_filter_dev, out_ids_dev, out_dist_dev, dct, dce, todo, true,
                                    plan.kprime, reserve, st, nullptr))) return rc;
        AK_HIP(hipStreamSynchronize(st));
    }
    guard.armed = false;      // the stream was synchronised after the last kernel: nothing of this call is in flight
    ix.ws_pending = false;
    return 0;
}

int ak_index_debug_read(ak_index_t h, int64_t *out, int n) {
    AK_BIND();
    if (!h || !out) AK_FAIL(-1, "ak_index_debug_read: NULL argument");
    Index &ix = *(Index *)h;
    if (!ix.dbg_dev) AK_FAIL(-7, "ak_index_debug_read: run a search with AK_SCAN_DBG=1 first");
    AK_HIP(hipDeviceSynchronize());
    AK_HIP(hipMemcpy(out, ix.dbg_dev, (size_t)(n < 2 * 65536 ? n : 2 * 65536) * 8, hipMemcpyDeviceToHost));
    return 0;
}

int ak_index_scan_plan(ak_index_t h, int nq, int k, int64_t *out8) {
    if (!h || !out8) AK_FAIL(-1, "ak_index_scan_plan: NULL argument");
    Index &ix = *(Index *)h;
    std::shared_lock<std::shared_mutex> lk(ix.mu);
    memset(out8, 0, 8 * sizeof(int64_t));
    if (!fast_supported(ix, nq, k)) return 0;
    FastPlan p = fast_plan(ix, nq, k);
    out8[0] = 1; out8[1] = p.cfg; out8[2] = p.kprime; out8[3] = p.nslices; out8[4] = p.nqg; out8[5] = p.ns_seed;
    out8[6] = p.seed_rows; out8[7] = p.qtile;
    return 0;
}

int ak_index_profile(ak_index_t h, int enable) {
    if (!h) AK_FAIL(-1, "ak_index_profile: NULL index");
    Index &ix = *(Index *)h;
    std::lock_guard<std::mutex> wl(ix.prof_mu);
    ix.profile = enable != 0;
    ix.prof_used = 0;
    return 0;
}

int ak_index_profile_read(ak_index_t h, float *out_ms, int cap, int *n_out) {
    AK_BIND();
    if (!h || !n_out) AK_FAIL(-1, "ak_index_profile_read: NULL argument");
    Index &ix = *(Index *)h;
    std::lock_guard<std::mutex> wl(ix.prof_mu);
    int n = 0;
    for (size_t i = 0; i < ix.prof_used && n < cap; i++) {
        float ms = 0.f;
        AK_HIP(hipEventElapsedTime(&ms, ix.prof_events[i].first, ix.prof_events[i].second));
        out_ms[n++] = ms;
    }
    *n_out = n;
    ix.prof_used = 0;
    return 0;
}

int ak_l2_normalize_dev(float *rows_dev, int64_t n, int dim, void *stream) {
    AK_BIND();
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float *nrm;
    AK_HIP(hipMallocAsync((void **)&nrm, (size_t)n * 4, st));
    k_row_invnorm<<<(unsigned)((n + 3) / 4), 256, 0, st>>>(rows_dev, n, dim, nrm);
    int64_t total = n * dim;
    k_scale_rows<<<(unsigned)std::min<int64_t>((total + 255) / 256, 8192), 256, 0, st>>>(rows_dev, nrm, total, dim);
    AK_HIP(hipGetLastError());
    AK_HIP(hipFreeAsync(nrm, st));
    return 0;
}

}  // extern "C"
