// index.h -- the HBM-resident corpus index behind ak_index_* (gfx950 only).
//
// Data layout in HBM (one shard = one process = one GPU):
//   rows  [cap][dim]  storage dtype (f32 / bf16 / f16), row-major, 16-B aligned rows
//   na    [cap] f32   pgvector-order sequential sum a[i]*a[i] of the STORED values
//   ea,eb [cap] f32   per-row epilogue terms of the candidate scan:
//                       cosine: s~ = dot * 1/sqrt(na)          (ea = rsqrt, eb = 0)
//                       l2    : s~ = dot - na/2                (ea = 1, eb = -na/2)
//                       ip    : s~ = dot                       (ea = 1, eb = 0)
//                     dead or zero-norm(cosine) rows: ea = 0, eb = -inf (never a candidate)
//   ids   [cap] i64   document_chunks.id of each row slot
//   alive [cap] u8    0 after ak_index_remove (tombstone)
#pragma once
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <vector>

#include "coalesce.h"      // request coalescing of ak_index_search: host-only, sanitizer-tested on its own
#include "common.h"
#include "index_book.h"    // id <-> slot map, tombstones, growth plan, layout epoch: host-only, sanitizer-tested on its own

namespace ak {

struct Workspace {
    void *buf = nullptr;
    size_t bytes = 0;
    int reserve(size_t need);
    void release();
};

// The host mirror (cap, n, n_alive, epoch, next_id, h_ids, h_alive, id2slot) is the IndexBook base; read and written under `mu`.
struct Index : IndexBook {
    int dim = 0, dtype = 0, metric = 0;
    void *rows = nullptr;
    void *shadow = nullptr;   // f32 corpora only: bf16 copy of the rows that the MFMA candidate scan reads
    float *na = nullptr, *ea = nullptr, *eb = nullptr;
    float *gb = nullptr;      // [ceil(cap/32)][4]: per 32-row block {max ea | rows 8q+0..3, max ea | rows 8q+4..7, max eb .., max eb ..}
    int64_t *ids = nullptr;
    uint8_t *alive = nullptr;
    float max_na = 0.f;  // max over rows of na (for the ip / l2 error bound)
    float max_rho = 0.f; // f32 corpora: max over rows of |a - shadow(a)| / |a| (measured at ingest; certificate term rho_c)
    Coalescer co;
    std::shared_mutex mu;
    Workspace ws_dev;     // workspace of ak_index_search_dev (one call at a time: ws_mu + ws_event order its users)
    Workspace ws_fb;      // ak_index_search_dev, AUTO mode: workspace of the re-run of uncertified queries (rare, grow-only)
    std::mutex ws_mu;
    std::mutex prof_mu;   // profile / debug state below (fast_search is reached under the shared lock only)
    // ak_index_search_dev is asynchronous: its kernels may still be using ws_dev (and reading ea/eb/gb/rows) after the call
    // returned. ws_event is recorded behind the last enqueued kernel; the next device search on ANOTHER stream waits for
    // it on the device, writers (add / remove / compaction) wait for it on the host before they touch the index.
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_pending = false;
    // optional per-launch timing of the scan kernel (ak_index_profile): event pairs
    // recorded on the launch stream, read back after the caller synchronised.
    float *max_dev = nullptr;       // landing pad of finish_rows' maxima {norm^2, shadow error} (allocated once, not per add)
    long long *dbg_dev = nullptr;   // AK_SCAN_DBG: per-wave phase cycle counters of the last two scan launches
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
};

// ---- exact path (exact.hip) -------------------------------------------------
// Reference arithmetic for every (query,row): keys -> hierarchical selection.
// queries_dev [nq][dim] f32, nb_dev [nq] f32 (pgvector-order sum q[i]^2),
// filter_dev NULL or [n] bytes. Outputs on device: [nq][k].
// `ws` holds exact_scratch_bytes(ix, k); nothing is allocated and nothing synchronises inside (asynchronous on st).
size_t exact_scratch_bytes(const Index &ix, int k);
int exact_search(Index &ix, const float *queries_dev, const float *nb_dev, int nq, int k,
                 const uint8_t *filter_dev, int64_t *out_ids_dev, double *out_dist_dev,
                 int *out_cnt_dev, void *ws, hipStream_t st);

// generic selection: smallest k (key,id) pairs per query, hierarchical.
//   keys [nq][n_in]; ids: explicit [nq][n_in] or NULL (then id = idmap ? idmap[i] : i)
//   result: okeys/oids [nq][k] sorted ascending; missing -> KEY_INVALID / -1
// scratch must hold select_scratch_bytes(nq, n_in, k).
size_t select_scratch_bytes(int nq, int64_t n_in, int k);
int select_topk(const uint64_t *keys, const int64_t *ids, const int64_t *idmap, int nq, int64_t n_in,
                int k, uint64_t *okeys, int64_t *oids, void *scratch, hipStream_t st);

int select_topk_strided(const uint64_t *keys, int nq, int64_t n_in, int64_t in_stride, int k, uint64_t *okeys,
                        int64_t *oids, void *scratch, hipStream_t st);

// key-only selection by bitonic sort (select.hip): keys [nq][in_stride] -> okeys [nq][k] ascending
size_t select_keys_scratch_bytes(int nq, int64_t n_in, int k);
int select_keys_topk(const uint64_t *keys, int nq, int64_t n_in, int64_t in_stride, int k, uint64_t *okeys,
                     void *scratch, hipStream_t st);

// exact re-rank of candidates: cand [nq][kp] approx keys (row slot in the low
// 32 bits) -> exact distance keys + global ids, same layout.
int rerank(Index &ix, const float *queries_dev, const float *nb_dev, int nq, int kp, const uint64_t *cand,
           uint64_t *okeys, int64_t *oids, hipStream_t st);

// per-query preparation: nb (pgvector-order sum of squares)
int query_norms(const float *queries_dev, int nq, int dim, float *nb_dev, hipStream_t st);

// keys/ids [nq][k] -> distances + counts
int emit_results(const uint64_t *keys, const int64_t *ids, int nq, int k, int64_t *out_ids,
                 double *out_dist, int *out_cnt, hipStream_t st);

// ---- fast path (scan.hip) ----------------------------------------------------
// per-query certificate terms: s~_units = a * s~' + b where s~' is the scan's score; eps in score units
struct QPrep { double eps, a, b; };

// Fused tail of the fast path (exact.hip), one workgroup per query: the dense candidate list the scans appended to
// (list [nq][lcap], cnt [nq]) -> drop candidates below the main-pass starting threshold thr0 -> best kp = 64 by approximate
// score -> exact re-rank in the reference arithmetic (rows staged through LDS) -> top-k by (distance, id) + certificate.
// thr_max [nq]: atomicMax over the workgroups' final thresholds as ascending-order uints (~score_key).
int fused_tail(Index &ix, const float *queries_dev, const float *nb_dev, int nq, int k, const uint64_t *list, int64_t lcap,
               const int *cnt, const unsigned int *thr_max, const float *thr0, const QPrep *prep, int64_t *out_ids_dev,
               double *out_dist_dev, int *out_cnt_dev, int *cert_dev, int64_t *stats_dev, hipStream_t st);
constexpr int TAIL_KP = 64, TAIL_MAX_DIM = 4096;

struct FastPlan {
    int cfg;        // tile configuration (scan.hip)
    int kprime;     // candidates kept per (slice, query) and re-ranked per query
    int qtile;      // queries per block
    int nslices;    // corpus slices (blocks along the corpus) of the main pass
    int ns_seed;    // slices of the seeding pass (0 = single pass)
    int64_t seed_rows;  // rows [0, seed_rows) are scanned first to seed the thresholds
    int64_t pre_tiles;  // pre-seeding sample: pre_tiles tiles, every pre_stride-th tile of the corpus (0 = none)
    int pre_slices;     // workgroups along the sample
    int pre_stride;
    int nqg;        // query groups
    size_t bytes;   // workspace bytes
};
bool fast_supported(const Index &ix, int nq, int k);
FastPlan fast_plan(const Index &ix, int nq, int k, bool widest = false);
// Candidate scan + select + exact re-rank + certification, all on `st`.
// cert_dev [nq] int32: 1 = top-k proven identical to the exact path.
// nb_dev [nq]: the queries' sums of squares in the reference arithmetic -- an INPUT when nb_ready, otherwise computed here
// (fused into the query set-up launch) and left for the caller. stats_dev (nullable, int64[4]) is zeroed here.
int fast_search(Index &ix, const float *queries_dev, float *nb_dev, bool nb_ready, int nq, int k,
                const uint8_t *filter_dev, int64_t *out_ids_dev, double *out_dist_dev,
                int *out_cnt_dev, int *cert_dev, int64_t *stats_dev, void *ws, const FastPlan &plan,
                hipStream_t st);

}  // namespace ak
