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
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);          // PyTorch-ROCm's copy, if the process has it mapped
        if (h) where = "librccl.so (already mapped)";
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
};

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
    delete c;
    return 0;
}

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
    // (a failure to get the exchange buffers themselves is the one local failure that cannot travel through the exchange: they are
    // sized by (nq, k, world) and grown at the first call of a shape, i.e. during warm-up, not under load)
    if (c.ws.reserve(total)) return -10;
    if (c.pin_cap < n_open * 4) {
        if (c.pin) hipHostFree(c.pin);
        c.pin = nullptr; c.pin_cap = 0;
        AK_HIP(hipHostMalloc((void **)&c.pin, n_open * 4));
        c.pin_cap = n_open * 4;
    }
    char *w = (char *)c.ws.buf;
    int64_t *pay = (int64_t *)(w + o_pay), *gat = (int64_t *)(w + o_gat);
    int *open = (int *)(w + o_open);
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
    // 1. local scan, results written straight into the exchange layout: [ids | float8 bits | flags | status]
    AK_HIP(hipMemsetAsync(pay + 2 * (int64_t)nq * k, 0, (size_t)((nq + 1) / 2 + 1) * 8, st));      // (flag padding and status 0 travel too)
    int rc_local = ak_index_search_dev(h, queries_dev, nq, k, AK_SEARCH_FAST_ONLY, row_filter_dev, filter_len, filter_epoch, pay,
                                       (double *)(pay + (int64_t)nq * k), (int *)(pay + 2 * (int64_t)nq * k), stream);
    std::string msg_local;
    if (rc_local) {
        msg_local = ak_last_error();
        const int64_t nqk = (int64_t)nq * k;
        k_fail_payload<<<(unsigned)((nqk + 255) / 256), 256, 0, st>>>(pay, nqk, (nq + 1) / 2, rc_local);
        AK_HIP(hipGetLastError());
    }
    // 2. the one collective of a search -- entered by EVERY rank, failed scan or not; 3. merge + flag reduction + status words
    int rc;
    AK_NCCL(rccl().AllGather(pay, gat, (size_t)L, ncclInt64, c.comm, st));
    if ((rc = ak_merge_shards_dev(c.world, nq, k, gat, L, out_ids_dev, out_dist_dev, open, stream))) return rc;
    k_collect_status<<<1, 1024, 0, st>>>(gat, c.world, L, open + nq + 1);
    AK_HIP(hipGetLastError());
    // 4. which queries did some shard leave open, which shard failed? (the search's one host synchronisation; every rank reads the same words)
    AK_HIP(hipMemcpyAsync(c.pin, open, n_open * 4, hipMemcpyDeviceToHost, st));
    AK_HIP(hipStreamSynchronize(st));
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
    int64_t *sp = (int64_t *)(w + o_sp), *sg = (int64_t *)(w + o_sg), *mi = (int64_t *)(w + o_mi);
    double *md = (double *)(w + o_md);
    int *sopen = (int *)(w + o_so);
    const int64_t Ls = payload_len(m, k);
    AK_HIP(hipMemcpyAsync(didx, idx.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    k_gather_rows_f32<<<m, 128, 0, st>>>(queries_dev, didx, m, ix.dim, sq);
    AK_HIP(hipGetLastError());
    AK_HIP(hipMemsetAsync(sp + 2 * (int64_t)m * k, 0, (size_t)((m + 1) / 2 + 1) * 8, st));
    // AUTO: widest-list scan, then the exact path; every flag is 1 on return (the call synchronises the stream when it re-runs).
    // It validates the filter's epoch again: a writer that moved the local layout since step 1 makes THIS rank's re-run fail --
    // with its code in the second payload, like above.
    rc_local = ak_index_search_dev(h, sq, m, k, AK_SEARCH_AUTO, row_filter_dev, filter_len, filter_epoch, sp, (double *)(sp + (int64_t)m * k),
                                   (int *)(sp + 2 * (int64_t)m * k), stream);
    if (rc_local) {
        msg_local = ak_last_error();
        const int64_t mk = (int64_t)m * k;
        k_fail_payload<<<(unsigned)((mk + 255) / 256), 256, 0, st>>>(sp, mk, (m + 1) / 2, rc_local);
        AK_HIP(hipGetLastError());
    }
    AK_NCCL(rccl().AllGather(sp, sg, (size_t)Ls, ncclInt64, c.comm, st));
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
