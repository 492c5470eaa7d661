// shardcomm.hip -- the row-sharded search with its exchange step INSIDE the C ABI: per-shard
// MFMA scan -> ONE ncclAllGather (RCCL over xGMI) of the partial top-k + certificate flags -> merge kernel -> flag
// reduction, all enqueued on one HIP stream with no interpreter in between; queries ANY shard left open are re-run on
// EVERY shard through the device AUTO path and exchanged again (all ranks read the same gathered flags, so they take the
// same branch without a further collective). A maintainer binding with ctypes alone reaches the multi-GPU search through
//   ak_comm_unique_id (rank 0) -> [share the 128 bytes by any channel] -> ak_comm_create (every rank) -> ak_index_search_sharded_dev.
// The reference has no counterpart: its scan runs inside one Postgres backend (postgres_vectorstore.py:317-332).
//
// RCCL is looked up at first use (dlopen: the copy already mapped into the process -- PyTorch-ROCm ships one -- else
// librccl.so.1 of the ROCm installation), like the roctx marker library in index.hip: libarchi_hip.so itself loads on a box
// without RCCL, and a process never ends up with two communicator libraries because of this file.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "index.h"
#include "switches.h"

namespace ak {
namespace {

// the slice of rccl.h this file uses (types restated so that the header is not needed at build time)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt64 = 4 };      // ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4
struct RcclApi {
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string where;
    bool ok = false;
    RcclApi() {
        // AK_RCCL_PATH (snapshot at load like every AK_* variable): the communicator library to use instead of the default
        // search -- a site's own RCCL build, or tests/native/fake_rccl.cpp, the shared-memory stand-in that lets several
        // ranks share the ONE GPU of a test box (RCCL itself refuses two ranks on one device)
        void *h = nullptr;
        if (const char *forced = env_get("AK_RCCL_PATH")) {
            if (*forced) {
                h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
                if (!h) return;                                          // named and not loadable: -12, not a silent other library
                where = forced;
            }
        }
        if (!h) { h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD); if (h) where = "librccl.so (already mapped)"; }   // PyTorch-ROCm's copy
        if (!h) { h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL); if (h) where = "librccl.so.1"; }
        if (!h) { h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL); if (h) where = "/opt/rocm/lib/librccl.so.1"; }
        if (!h) { h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL); if (h) where = "librccl.so"; }
        if (!h) return;
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        ok = GetUniqueId && CommInitRank && CommDestroy && AllGather;
    }
};
const RcclApi &rccl() {
    static const RcclApi api;
    return api;
}
#define AK_NCCL(call)                                                                                              \
    do {                                                                                                           \
        const int e__ = (call);                                                                                    \
        if (e__ != ncclSuccess) {                                                                                  \
            ak::set_error(std::string(#call) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(e__) : "?") + \
                          " (rccl error " + std::to_string(e__) + ")");                                            \
            return -12;                                                                                            \
        }                                                                                                          \
    } while (0)

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    std::mutex mu;          // one sharded search at a time per communicator: collectives pair up by issue order
    Workspace ws;
    int *pin = nullptr;     // pinned landing pad of the open-query flags
    size_t pin_cap = 0;
    int64_t *pin64 = nullptr;   // failure path only: the gathered flag + status words of every rank, read on the host
    size_t pin64_cap = 0;
    // Set when THIS rank left a search at a point where the other ranks may still enter a collective it will not join (the HIP
    // runtime refused an allocation, a copy or a launch: the device or the stream is gone). Collectives pair up by issue order,
    // so nothing issued on this communicator afterwards could be trusted: every later call fails fast with AK_ERR_COMM_BROKEN
    // instead of pairing with the wrong collective. Destroy the communicator and create a new one on every rank.
    bool broken = false;
};
#define AK_SHARD_HIP(call)                                                                                         \
    do {                                                                                                           \
        const hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess) {                                                                                   \
            c.broken = true;                                                                                       \
            ak::set_error(std::string(#call) + " failed: " + hipGetErrorString(e__) +                              \
                          " -- the communicator is broken (AK_ERR_COMM_BROKEN on every later call): re-create it on every rank"); \
            return -10;                                                                                            \
        }                                                                                                          \
    } while (0)
#define AK_SHARD_NCCL(call)                                                                                        \
    do {                                                                                                           \
        const int e__ = (call);                                                                                    \
        if (e__ != ncclSuccess) {                                                                                  \
            c.broken = true;                                                                                       \
            ak::set_error(std::string(#call) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(e__) : "?") + \
                          " (rccl error " + std::to_string(e__) + "); the communicator is broken");                \
            return -12;                                                                                            \
        }                                                                                                          \
    } while (0)

// exchange payload of one rank, int64 words: [ids nq * k | float8 bits nq * k | certificate flags, nq int32 padded to whole words |
// STATUS]. The status word (round 5) carries the rank's local return code: a rank whose scan failed (stale filter, workspace
// ...) STILL joins the all-gather, with empty rows and its code there, so every rank reads the same codes after the collective
// and all of them return the same error -- nobody is left waiting in a collective the failing rank never entered. The merge
// kernel addresses ids / bits / flags from the front of a payload and never looks at the tail word.
inline int64_t payload_len(int nq, int k) { return 2 * (int64_t)nq * k + (nq + 1) / 2 + 1; }
__global__ void k_fail_payload(int64_t *__restrict__ pay, int64_t nqk, int64_t nflagwords, int64_t status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nqk) { pay[i] = -1; pay[nqk + i] = 0x7ff8000000000000ll; }            // no row, NaN distance
    if (i < nflagwords) pay[2 * nqk + i] = 0x0000000100000001ll;                   // "certified": the failing rank asks for no re-run
    if (i == 0) pay[2 * nqk + nflagwords] = status;
}
__global__ void k_set_word(int64_t *p, int64_t v) { *p = v; }
__global__ void k_collect_status(const int64_t *__restrict__ gat, int world, int64_t L, int *__restrict__ out) {
    const int r = threadIdx.x;
    if (r < world) out[r] = (int)gat[(int64_t)r * L + L - 1];
}

__global__ void k_gather_rows_f32(const float *__restrict__ q, const int *__restrict__ idx, int m, int dim, float *__restrict__ out) {
    const int j = blockIdx.x;
    if (j >= m) return;
    const float *src = q + (int64_t)idx[j] * dim;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) out[(int64_t)j * dim + c] = src[c];
}
__global__ void k_scatter_rows(const int *__restrict__ idx, int m, int k, const int64_t *__restrict__ si, const double *__restrict__ sd,
                               int64_t *__restrict__ oi, double *__restrict__ od) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * k) return;
    const int j = t / k, c = t % k;
    oi[(int64_t)idx[j] * k + c] = si[t];
    od[(int64_t)idx[j] * k + c] = sd[t];
}
__global__ void k_fill_i32(int *p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace
}  // namespace ak

using namespace ak;

extern "C" {

int ak_comm_unique_id(void *out_id128) {
    if (!out_id128) AK_FAIL(-1, "ak_comm_unique_id: out is NULL");
    if (!rccl().ok) AK_FAIL(-12, "ak_comm_unique_id: RCCL (librccl.so) not found");
    ncclUniqueId id;
    AK_NCCL(rccl().GetUniqueId(&id));
    memcpy(out_id128, id.internal, 128);
    return 0;
}

int ak_comm_create(const void *unique_id128, int rank, int world, ak_comm_t *out) {
    AK_BIND();
    if (!unique_id128 || !out || world < 1 || rank < 0 || rank >= world) AK_FAIL(-1, "ak_comm_create: bad arguments");
    if (!rccl().ok) AK_FAIL(-12, "ak_comm_create: RCCL (librccl.so) not found");
    ncclUniqueId id;
    memcpy(id.internal, unique_id128, 128);
    Comm *c = new Comm();
    c->rank = rank; c->world = world;
    const int e = rccl().CommInitRank(&c->comm, world, id, rank);
    if (e != ncclSuccess) {
        delete c;
        AK_FAIL(-12, std::string("ak_comm_create: ncclCommInitRank failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(e) : "?"));
    }
    *out = c;
    return 0;
}

int ak_comm_destroy(ak_comm_t h) {
    if (!h) return 0;
    Comm *c = (Comm *)h;
    if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
    c->ws.release();
    if (c->pin) hipHostFree(c->pin);
    if (c->pin64) hipHostFree(c->pin64);
    delete c;
    return 0;
}

// ---- the exchange step's small device helpers, exported: archi_amd/sharded.py drives the same exchange over torch.distributed
// (RCCL) and used torch kernels (cat / nonzero / index_select / index_copy_ / fill) for what these launches do (round-5 review) ----
int ak_shard_payload_begin_dev(int64_t *payload_dev, int nq, int k, void *stream) {
    AK_BIND();
    if (!payload_dev || nq <= 0 || k <= 0) AK_FAIL(-1, "ak_shard_payload_begin_dev: bad arguments");
    AK_HIP(hipMemsetAsync(payload_dev + 2 * (int64_t)nq * k, 0, (size_t)((nq + 1) / 2 + 1) * 8, (hipStream_t)stream));
    return 0;
}
int ak_shard_fail_payload_dev(int64_t *payload_dev, int nq, int k, int status, void *stream) {
    AK_BIND();
    if (!payload_dev || nq <= 0 || k <= 0) AK_FAIL(-1, "ak_shard_fail_payload_dev: bad arguments");
    const int64_t nqk = (int64_t)nq * k;
    k_fail_payload<<<(unsigned)((nqk + 255) / 256), 256, 0, (hipStream_t)stream>>>(payload_dev, nqk, (nq + 1) / 2, status);
    AK_HIP(hipGetLastError());
    return 0;
}
int ak_shard_status_dev(int g, const int64_t *gathered_dev, int64_t stride, int *out_status_dev, void *stream) {
    AK_BIND();
    if (g <= 0 || g > 1024 || !gathered_dev || !out_status_dev || stride <= 0) AK_FAIL(-1, "ak_shard_status_dev: bad arguments (at most 1024 ranks)");
    k_collect_status<<<1, 1024, 0, (hipStream_t)stream>>>(gathered_dev, g, stride, out_status_dev);
    AK_HIP(hipGetLastError());
    return 0;
}
int ak_shard_gather_rows_dev(const float *rows_dev, const int *idx_dev, int m, int dim, float *out_dev, void *stream) {
    AK_BIND();
    if (m < 0 || dim <= 0 || (m > 0 && (!rows_dev || !idx_dev || !out_dev))) AK_FAIL(-1, "ak_shard_gather_rows_dev: bad arguments");
    if (m == 0) return 0;
    k_gather_rows_f32<<<m, 128, 0, (hipStream_t)stream>>>(rows_dev, idx_dev, m, dim, out_dev);
    AK_HIP(hipGetLastError());
    return 0;
}
int ak_shard_scatter_topk_dev(const int *idx_dev, int m, int k, const int64_t *sub_ids_dev, const double *sub_dist_dev, int64_t *out_ids_dev,
                              double *out_dist_dev, void *stream) {
    AK_BIND();
    if (m < 0 || k <= 0 || (m > 0 && (!idx_dev || !sub_ids_dev || !sub_dist_dev || !out_ids_dev || !out_dist_dev)))
        AK_FAIL(-1, "ak_shard_scatter_topk_dev: bad arguments");
    if (m == 0) return 0;
    k_scatter_rows<<<(unsigned)(((int64_t)m * k + 255) / 256), 256, 0, (hipStream_t)stream>>>(idx_dev, m, k, sub_ids_dev, sub_dist_dev, out_ids_dev,
                                                                                               out_dist_dev);
    AK_HIP(hipGetLastError());
    return 0;
}

// WHO WAITS FOR WHOM. The contract: a search either returns the same rows on every rank or the same
// error code on every rank, and no rank is left inside a collective another rank never enters. By failure site:
//   * the local scan fails (stale filter, workspace, a launch inside it) ............ travels in the STATUS word of the payload; the
//     rank still enters the all-gather with empty rows; every rank returns that code after the collective.
//   * the first merge fails on one rank (between the first all-gather and a possible second one) ... that rank reads the gathered
//     flag and status words itself (host copy) and knows what the others will do: if some query is open they enter the second
//     all-gather, so it JOINS it with empty rows and its code -- every rank then returns that code; if none is open the others
//     have their result and only this rank returns its error.
//   * the merge after the LAST collective fails .................................... nobody waits any more: only this rank returns it.
//   * the exchange buffers cannot be allocated, or the HIP runtime refuses a memset / copy / launch / synchronise of the
//     exchange itself, or RCCL returns an error ....................................... FATAL TO THE COMMUNICATOR: there is no
//     payload to send, or no working stream to send it on. The rank returns, the communicator is marked broken and every later
//     call on it returns AK_ERR_COMM_BROKEN at once (it would pair with the wrong collective). The other ranks' collective
//     is ended by RCCL's own timeout / abort, as after the loss of a process. They are sized by (nq, k, world) and grown at the
//     first call of a shape, i.e. at warm-up.
// AK_SHARD_INJECT (ak_debug_set; a test hook, errors only, never wrong rows): 1 = this rank's local scan fails with -10,
// 2 = this rank's first merge fails with -10, 3 = this rank's exchange-buffer reservation fails (the fatal class).
int ak_index_search_sharded_dev(ak_index_t h, ak_comm_t ch, const float *queries_dev, int nq, int k, const uint8_t *row_filter_dev,
                                int64_t filter_len, uint64_t filter_epoch, int64_t *out_ids_dev, double *out_dist_dev,
                                int64_t *out_rerun, void *stream) {
    AK_BIND();
    if (out_rerun) *out_rerun = 0;
    if (!h || !ch) AK_FAIL(-1, "ak_index_search_sharded_dev: NULL index or communicator");
    if (nq <= 0) return 0;
    if (k <= 0 || !queries_dev || !out_ids_dev || !out_dist_dev) AK_FAIL(-1, "ak_index_search_sharded_dev: bad arguments");
    Comm &c = *(Comm *)ch;
    Index &ix = *(Index *)h;
    if (c.world > 1024) AK_FAIL(-1, "ak_index_search_sharded_dev: more than 1024 ranks");
    RoctxRange range("ak_index_search_sharded_dev");
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.broken)
        AK_FAIL(AK_ERR_COMM_BROKEN, "ak_index_search_sharded_dev: an earlier search left this communicator at a point the other ranks could not "
                                    "follow (see that call's error); destroy it and create a new one on every rank");
    const int inject = switches().shard_inject.load(std::memory_order_relaxed);
    hipStream_t st = (hipStream_t)stream;
    const int64_t L = payload_len(nq, k);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // [payload L | gathered world * L | open nq + 1, status world] then, for the re-run of m <= nq open queries:
    // [idx nq | sub queries nq * dim | sub payload L | sub gathered world * L | merged ids nq * k | merged dist nq * k | sub open nq + 1, status world]
    const size_t n_open = (size_t)(nq + 1 + c.world);
    const size_t o_pay = 0, o_gat = o_pay + al((size_t)L * 8), o_open = o_gat + al((size_t)c.world * L * 8),
                 o_idx = o_open + al(n_open * 4), o_sq = o_idx + al((size_t)nq * 4), o_sp = o_sq + al((size_t)nq * ix.dim * 4),
                 o_sg = o_sp + al((size_t)L * 8), o_mi = o_sg + al((size_t)c.world * L * 8), o_md = o_mi + al((size_t)nq * k * 8),
                 o_so = o_md + al((size_t)nq * k * 8), total = o_so + al(n_open * 4);
    if (inject == 3 || c.ws.reserve(total)) {
        c.broken = true;
        set_error(std::string("ak_index_search_sharded_dev: the exchange buffers (") + std::to_string(total) + " bytes) could not be reserved" +
                  (inject == 3 ? " [injected]" : "") + " -- the communicator is broken: re-create it on every rank");
        return -10;
    }
    if (c.pin_cap < n_open * 4) {
        if (c.pin) hipHostFree(c.pin);
        c.pin = nullptr; c.pin_cap = 0;
        AK_SHARD_HIP(hipHostMalloc((void **)&c.pin, n_open * 4));
        c.pin_cap = n_open * 4;
    }
    char *w = (char *)c.ws.buf;
    int64_t *pay = (int64_t *)(w + o_pay), *gat = (int64_t *)(w + o_gat);
    int *open = (int *)(w + o_open);
    int64_t *sp = (int64_t *)(w + o_sp), *sg = (int64_t *)(w + o_sg);
    // every rank reads every rank's status after a merge: the lowest failing rank's code is the search's, on all of them
    auto agreed_status = [&](const int *status, int local_rc, const std::string &local_msg) -> int {
        for (int r = 0; r < c.world; r++)
            if (status[r] != 0) {
                if (r == c.rank && local_rc) set_error(local_msg);
                else set_error("ak_index_search_sharded_dev: the local search of shard " + std::to_string(r) + " failed (rc " +
                               std::to_string(status[r]) + (status[r] == AK_ERR_STALE_FILTER ? ", stale row_filter" : "") + "); every rank returns it");
                return status[r];
            }
        return 0;
    };
    // This rank's FIRST merge failed: follow the others through whatever collective they still enter (comment above).
    auto follow_after_failed_merge = [&](int rc_merge) -> int {
        const std::string msg = ak_last_error();
        const int64_t nfw = (nq + 1) / 2, per = nfw + 1;
        if (c.pin64_cap < (size_t)c.world * per * 8) {
            if (c.pin64) hipHostFree(c.pin64);
            c.pin64 = nullptr; c.pin64_cap = 0;
            AK_SHARD_HIP(hipHostMalloc((void **)&c.pin64, (size_t)c.world * per * 8));
            c.pin64_cap = (size_t)c.world * per * 8;
        }
        for (int r = 0; r < c.world; r++)
            AK_SHARD_HIP(hipMemcpyAsync(c.pin64 + (int64_t)r * per, gat + (int64_t)r * L + 2 * (int64_t)nq * k, (size_t)per * 8, hipMemcpyDeviceToHost, st));
        AK_SHARD_HIP(hipStreamSynchronize(st));
        for (int r = 0; r < c.world; r++) {
            const int code = (int)c.pin64[(int64_t)r * per + nfw];
            if (code != 0) {      // the others return this code right after their merge, before any second collective: so does this rank
                set_error("ak_index_search_sharded_dev: the local search of shard " + std::to_string(r) + " failed (rc " + std::to_string(code) +
                          "); every rank returns it (this rank's merge also failed: " + msg + ")");
                return code;
            }
        }
        int m = 0;
        for (int i = 0; i < nq; i++) {
            int o = 0;
            for (int r = 0; r < c.world; r++) o |= ((const int *)(c.pin64 + (int64_t)r * per))[i] == 0;
            m += o;
        }
        if (m > 0) {              // the others re-run m queries and gather payload_len(m, k) words per rank: join with empty rows + the code
            const int64_t Ls = payload_len(m, k), mk = (int64_t)m * k;
            k_fail_payload<<<(unsigned)((mk + 255) / 256), 256, 0, st>>>(sp, mk, (m + 1) / 2, rc_merge);
            AK_SHARD_HIP(hipGetLastError());
            AK_SHARD_NCCL(rccl().AllGather(sp, sg, (size_t)Ls, ncclInt64, c.comm, st));
            AK_SHARD_HIP(hipStreamSynchronize(st));
        }
        set_error(msg);
        return rc_merge;
    };
    // 1. local scan, results written straight into the exchange layout: [ids | float8 bits | flags | status]
    AK_SHARD_HIP(hipMemsetAsync(pay + 2 * (int64_t)nq * k, 0, (size_t)((nq + 1) / 2 + 1) * 8, st));      // (flag padding and status 0 travel too)
    int rc_local;
    if (inject == 1) { set_error("ak_index_search_sharded_dev: injected local failure (AK_SHARD_INJECT=1)"); rc_local = -10; }
    else rc_local = ak_index_search_dev(h, queries_dev, nq, k, AK_SEARCH_FAST_ONLY, row_filter_dev, filter_len, filter_epoch, pay,
                                        (double *)(pay + (int64_t)nq * k), (int *)(pay + 2 * (int64_t)nq * k), stream);
    std::string msg_local;
    if (rc_local) {
        msg_local = ak_last_error();
        const int64_t nqk = (int64_t)nq * k;
        k_fail_payload<<<(unsigned)((nqk + 255) / 256), 256, 0, st>>>(pay, nqk, (nq + 1) / 2, rc_local);
        AK_SHARD_HIP(hipGetLastError());
    }
    // 2. the one collective of a search -- entered by EVERY rank, failed scan or not; 3. merge + flag reduction + status words
    int rc;
    AK_SHARD_NCCL(rccl().AllGather(pay, gat, (size_t)L, ncclInt64, c.comm, st));
    if (inject == 2) { set_error("ak_index_search_sharded_dev: injected merge failure (AK_SHARD_INJECT=2)"); rc = -10; }
    else rc = ak_merge_shards_dev(c.world, nq, k, gat, L, out_ids_dev, out_dist_dev, open, stream);
    if (rc) return follow_after_failed_merge(rc);
    k_collect_status<<<1, 1024, 0, st>>>(gat, c.world, L, open + nq + 1);
    AK_SHARD_HIP(hipGetLastError());
    // 4. which queries did some shard leave open, which shard failed? (the search's one host synchronisation; every rank reads the same words)
    AK_SHARD_HIP(hipMemcpyAsync(c.pin, open, n_open * 4, hipMemcpyDeviceToHost, st));
    AK_SHARD_HIP(hipStreamSynchronize(st));
    if ((rc = agreed_status(c.pin + nq + 1, rc_local, msg_local))) return rc;
    const int m = c.pin[nq];
    if (out_rerun) *out_rerun = m;
    if (m == 0) return 0;
    std::vector<int> idx;
    idx.reserve((size_t)m);
    for (int i = 0; i < nq; i++) if (c.pin[i]) idx.push_back(i);
    if ((int)idx.size() != m) AK_FAIL(-10, "ak_index_search_sharded_dev: open-query flags and their count disagree");   // (same words on every rank: all of them leave here)
    int *didx = (int *)(w + o_idx);
    float *sq = (float *)(w + o_sq);
    int64_t *mi = (int64_t *)(w + o_mi);
    double *md = (double *)(w + o_md);
    int *sopen = (int *)(w + o_so);
    const int64_t Ls = payload_len(m, k);
    AK_SHARD_HIP(hipMemcpyAsync(didx, idx.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    k_gather_rows_f32<<<m, 128, 0, st>>>(queries_dev, didx, m, ix.dim, sq);
    AK_SHARD_HIP(hipGetLastError());
    AK_SHARD_HIP(hipMemsetAsync(sp + 2 * (int64_t)m * k, 0, (size_t)((m + 1) / 2 + 1) * 8, st));
    // AUTO: widest-list scan, then the exact path; every flag is 1 on return (the call synchronises the stream when it re-runs).
    // It validates the filter's epoch again: a writer that moved the local layout since step 1 makes THIS rank's re-run fail --
    // with its code in the second payload, like above.
    rc_local = ak_index_search_dev(h, sq, m, k, AK_SEARCH_AUTO, row_filter_dev, filter_len, filter_epoch, sp, (double *)(sp + (int64_t)m * k),
                                   (int *)(sp + 2 * (int64_t)m * k), stream);
    if (rc_local) {
        msg_local = ak_last_error();
        const int64_t mk = (int64_t)m * k;
        k_fail_payload<<<(unsigned)((mk + 255) / 256), 256, 0, st>>>(sp, mk, (m + 1) / 2, rc_local);
        AK_SHARD_HIP(hipGetLastError());
    }
    AK_SHARD_NCCL(rccl().AllGather(sp, sg, (size_t)Ls, ncclInt64, c.comm, st));
    // (the LAST collective is behind every rank: a failure from here on is this rank's alone and nobody waits for it)
    if ((rc = ak_merge_shards_dev(c.world, m, k, sg, Ls, mi, md, sopen, stream))) return rc;
    k_collect_status<<<1, 1024, 0, st>>>(sg, c.world, Ls, sopen + m + 1);
    AK_HIP(hipGetLastError());
    AK_HIP(hipMemcpyAsync(c.pin, sopen + m, (size_t)(1 + c.world) * 4, hipMemcpyDeviceToHost, st));
    AK_HIP(hipStreamSynchronize(st));          // (idx, pageable, was the source of an asynchronous copy: it outlives it here)
    if ((rc = agreed_status(c.pin + 1, rc_local, msg_local))) return rc;      // (nothing of the failed re-run is scattered into the outputs)
    if (c.pin[0] != 0) AK_FAIL(-10, "ak_index_search_sharded_dev: a query stayed uncertified after the exact re-run");
    k_scatter_rows<<<(m * k + 255) / 256, 256, 0, st>>>(didx, m, k, mi, md, out_ids_dev, out_dist_dev);
    AK_HIP(hipGetLastError());
    AK_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
