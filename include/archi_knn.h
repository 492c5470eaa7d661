/*
 * archi_knn.h -- C ABI of libarchi_hip.so, the MI355X (gfx950) embedding +
 * retrieval backend that drops in behind archi's embedding-provider /
 * vector-store plugin surface.
 *
 * The reference (archi-physics/archi) is pure Python and has NO FFI for this
 * path: its two engines are reached through
 *   - psycopg2 + SQL (pgvector operators)      src/data_manager/vectorstore/postgres_vectorstore.py:317-335
 *   - LangChain Embeddings.embed_documents     src/data_manager/vectorstore/manager.py:373
 * so every entry point below cites the reference call it REPLACES. The binding
 * a maintainer adds is a ctypes stub; signatures use only
 * plain pointers and sizes.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; ak_last_error() gives a
 *     thread-local message. Nothing falls back to the CPU.
 *   - the caller allocates all outputs; the library never frees caller memory;
 *     pointers are only valid for the duration of the call.
 *   - "dev" pointers are HIP device pointers on the device chosen by ak_init
 *     (one process per GPU); `stream` is a hipStream_t passed as void* (NULL =
 *     the default stream). torch users pass torch.cuda.current_stream().cuda_stream.
 *   - search entry points are re-entrant (Flask request threads call them
 *     concurrently, src/interfaces/chat_app/app.py:1554); add/remove assume a
 *     single writer (src/bin/service_data_manager.py:38,62-73) and are
 *     serialised against searches by a per-index lock.
 */
#ifndef ARCHI_KNN_H
#define ARCHI_KNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* storage dtype of the corpus matrix in HBM */
#define AK_DTYPE_F32 0
#define AK_DTYPE_BF16 1
#define AK_DTYPE_F16 2

/* distance metric == pgvector operator chosen at postgres_vectorstore.py:74-82 */
#define AK_METRIC_COSINE 0 /* "<=>" */
#define AK_METRIC_L2 1     /* "<->" */
#define AK_METRIC_IP 2     /* "<#>" (negative inner product) */

/* search mode */
#define AK_SEARCH_AUTO 0  /* MFMA candidate scan + exact re-rank + certification, exact fallback */
#define AK_SEARCH_EXACT 1 /* exact-arithmetic scan only (slow, reference arithmetic for every row) */
#define AK_SEARCH_FAST_ONLY 2 /* AUTO without the fallback: uncertified queries are reported, not re-run */

/* distinct error code of the search entry points: the row_filter was built for another layout of the index (see
 * ak_index_slots); nothing of it was read; rebuild it and retry */
#define AK_ERR_STALE_FILTER (-11)

/* ak_index_search_sharded_dev on a communicator that an earlier call left at a point the other ranks could not follow (the HIP
 * runtime or RCCL refused a step of the exchange itself): destroy the communicator and create a new one on every rank */
#define AK_ERR_COMM_BROKEN (-13)

/* encoder pooling (sentence-transformers Pooling module [upstream]) */
#define AK_POOL_MEAN 0 /* all-MiniLM-L6-v2 */
#define AK_POOL_CLS 1  /* bge-base-en */

typedef void *ak_index_t;
typedef void *ak_encoder_t;

/* ABI version: bumped whenever a signature in this header changes (3: filter_len / filter_epoch on the search entry points,
 * a fourth out-pointer on ak_index_slots -- round 4; ak_abi_version / ak_debug_set / ak_encoder_forward_lens -- round 5;
 * 4: the sharded exchange's payload carries a status word per rank (wire format of ak_index_search_sharded_dev / ak_merge_shards_dev
 * callers), AK_ERR_COMM_BROKEN and the ak_shard_* helpers, AkBertConfig.precision 2 -- round 6).
 * A binding checks ak_abi_version() == AK_ABI_VERSION right after loading the library (archi_amd/_lib.py does) instead of
 * passing arguments to a function whose parameter list has moved. */
#define AK_ABI_VERSION 4

/* ---- library ---------------------------------------------------------- */
const char *ak_last_error(void);
const char *ak_version(void);
int ak_abi_version(void);
/* Measurement / test hook (no reference counterpart): sets one of the A-B switches of DESIGN.md section 7 for this process, as if
 * the environment variable `name` had held `value` at start-up (NULL or "" = the default). The library reads its switches from
 * the environment ONCE; nothing on the request path calls getenv. Returns -1 for an unknown name -- and, in libarchi_hip.so, for
 * the stage-skipping switches that produce wrong results: those exist only in libarchi_hip_dbg.so. */
int ak_debug_set(const char *name, const char *value);
/* Bind this process to one GPU (one process per GPU). */
int ak_init(int device);
int ak_device_info(char *name_out, int name_cap, int *cu_count, int64_t *hbm_bytes);
int ak_sync(void *stream);

/* ---- index: replaces the document_chunks.embedding column + pgvector --- */
/* src/cli/templates/init.sql:256-274 (vector(D) column), exact branch :290-292 */
int ak_index_create(int64_t capacity, int dim, int dtype, int metric, ak_index_t *out);
int ak_index_destroy(ak_index_t h);

/* INSERT ... %s::vector   (postgres_vectorstore.py:168-180, manager.py:414-422)
 * rows: [n][dim] float32, host (is_device=0) or device (is_device=1) memory.
 * ids : [n] int64 host array (document_chunks.id), unique (-6 on a repeat inside the batch or against a live row);
 *        NULL -> consecutive from one above the largest id ever stored.
 * normalise != 0 applies x / max(||x||, 1e-12) before storing (a3).            */
int ak_index_add(ak_index_t h, const float *rows, int is_device, int64_t n, const int64_t *ids,
                 int normalise);

/* Synthetic corpus generated on device (bench / large parity tests): the
 * counter-based generator specified in oracle/knn_oracle.c (ako_gen_rows).
 * Appends rows [row0, row0+n) of stream `stream`; ids = id0 + i.              */
int ak_index_generate(ak_index_t h, uint64_t seed, uint32_t stream, uint64_t row0, int64_t n,
                      int normalise, int64_t id0);

/* DELETE FROM document_chunks WHERE ... (postgres_vectorstore.py:516-529).
 * Unknown ids are ignored; *n_removed (may be NULL) gets the number deleted. */
int ak_index_remove(ak_index_t h, const int64_t *ids, int64_t n, int64_t *n_removed);

/* SELECT COUNT(*) (postgres_vectorstore.py:570-585): live rows. */
int ak_index_count(ak_index_t h, int64_t *out);
/* Row slots in use (live + tombstones: the length of a row_filter), the current capacity and the LAYOUT EPOCH. The index
 * grows by itself (ak_index_create's capacity is only the first reservation) and reclaims tombstones when an add would
 * otherwise not fit: the reference's table has no capacity and update_vectorstore deletes and re-adds changed files
 * (manager.py:192-211). The epoch changes with every add (the slot count grows) and every reclaim (slot numbers change);
 * a delete alone leaves it (a mask that lets a deleted row pass is harmless). A row_filter is valid for exactly one
 * (slots, epoch) pair, read here in ONE call, and the search entry points take that pair with the mask: the reference
 * evaluates WHERE, distance, ORDER BY and LIMIT in one SQL statement = one snapshot (postgres_vectorstore.py:296-332),
 * and this is how a caller that builds its mask outside the library's lock gets the same guarantee. Any pointer may be NULL. */
int ak_index_slots(ak_index_t h, int64_t *out_slots, int64_t *out_capacity, uint64_t *out_epoch);
/* Reclaim every tombstone now (VACUUM; manager.py:103-153 runs VACUUM FULL at reset). *n_reclaimed may be NULL. */
int ak_index_compact(ak_index_t h, int64_t *n_reclaimed);

/* Copy stored rows back as float32 (values exactly as stored). rows: row slots. */
int ak_index_fetch(ak_index_t h, const int64_t *row_slots, int64_t n, float *out_host);
/* id -> row slot (-1 when absent). */
int ak_index_lookup(ak_index_t h, const int64_t *ids, int64_t n, int64_t *out_slots);
/* Semantic leg of the hybrid query: `1.0 - (c.embedding <op> %s::vector) AS semantic_score` for the rows a
 * BM25 match returns (postgres_vectorstore.py:435-457). Distances of ONE query [dim] float32 (host) to the
 * listed ids, in the search's exact arithmetic; out_dist[i] = NaN and out_found[i] = 0 (out_found may be
 * NULL) for ids that are absent or deleted. */
int ak_index_distances(ak_index_t h, const float *query, const int64_t *ids, int64_t n, double *out_dist,
                       uint8_t *out_found);

/* SELECT ... embedding <op> %s::vector AS distance ... WHERE ... ORDER BY distance
 * ASC LIMIT k   (postgres_vectorstore.py:317-332).
 *   queries   : [nq][dim] float32, host memory
 *   row_filter: NULL, or [filter_len] bytes on the HOST indexed by row slot (see
 *               ak_index_lookup): rows with 0 fail the WHERE clause (:296-310).
 *               filter_len / filter_epoch: the (slots, epoch) pair ak_index_slots returned
 *               when the mask was built; if the index has moved on since (a concurrent
 *               add or reclaim), the call returns AK_ERR_STALE_FILTER without reading the
 *               mask. Both are ignored when row_filter is NULL.
 *   out_ids   : [nq][k] int64; out_dist: [nq][k] float64 (pgvector float8
 *               distance; the caller computes score = 1 - distance for cosine,
 *               :361). Unused tail slots: id -1, distance NaN.
 *   out_counts: [nq] rows returned per query (NULL allowed)
 *   out_stats : NULL or int64[4] = {queries certified by the fast path (either scan),
 *               queries re-run exactly, candidates re-ranked, queries certified
 *               only by the second, widest-candidate-list scan}
 * Re-entrant: the reference runs one such SELECT per request thread
 * (src/interfaces/chat_app/app.py:1554 -> postgres_vectorstore.py:227-248). Calls
 * with nq <= 16 that arrive while another call's search is in flight are COALESCED:
 * they wait for it, then one of them searches for all that carry the same k, mode
 * and row_filter POINTER (+ length and epoch) in one launch (a scan costs the same for 1 query as for 64);
 * each caller gets its own rows back (out_stats then describes the shared launch).
 * Nobody waits when the index is idle. AK_COALESCE=0 turns it off.            */
int ak_index_search(ak_index_t h, const float *queries, int nq, int k, int mode,
                    const uint8_t *row_filter, int64_t filter_len, uint64_t filter_epoch,
                    int64_t *out_ids, double *out_dist, int *out_counts, int64_t *out_stats);

/* Same query, everything resident in HBM (bench path and the row-sharded multi-GPU path: inputs already on the device
 * when the timed region starts; `stream` orders the work).
 *   mode AK_SEARCH_FAST_ONLY: asynchronous, nothing synchronises with the host. out_cert_dev [nq] int32 receives 1 for
 *        every query whose top-k is PROVEN identical to the exact path; the rows of a query with 0 may differ and must be
 *        re-run by the caller (archi_amd/sharded.py reduces the flags over the ranks and re-runs them with AUTO).
 *   mode AK_SEARCH_AUTO: the same, then the flags are read back (one host synchronisation) and open queries are re-run
 *        on the device -- second scan with the widest candidate lists, then the exact path -- so that on return every row
 *        is the reference's ORDER BY distance LIMIT k and every flag is 1.
 *   mode AK_SEARCH_EXACT: reference arithmetic for every row (asynchronous).
 *   Shapes the MFMA scan does not take (fewer than 4096 rows, an empty shard, dim % 64 != 0, k > 128) run the exact path
 *   in every mode and report 1.
 *   row_filter_dev: NULL or [filter_len] bytes on the DEVICE (the WHERE clause, as ak_index_search's row_filter, with the
 *        same (filter_len, filter_epoch) contract and AK_ERR_STALE_FILTER).
 *   out_cert_dev may be NULL (AUTO / EXACT).
 * Workspace comes from the index (grown on first use). Calls on one index are serialised; a call on another stream
 * waits on the device for the previous call's kernels, and ak_index_add / remove / compact wait for them on the host. */
int ak_index_search_dev(ak_index_t h, const float *queries_dev, int nq, int k, int mode,
                        const uint8_t *row_filter_dev, int64_t filter_len, uint64_t filter_epoch,
                        int64_t *out_ids_dev, double *out_dist_dev, int *out_cert_dev, void *stream);

/* How a search of this shape would run: out8 = {fast path usable, tile config id, k', corpus
 * slices, query groups, seed-pass slices, seed-pass rows, queries per workgroup}. The main scan
 * launch covers rows [seed_rows, count).                                                      */
int ak_index_scan_plan(ak_index_t h, int nq, int k, int64_t *out8);

/* Developer aid: per-wave phase cycle counters of the last search's two scan launches (seed pass at
 * [0,65536), main pass at [65536,131072), 8 int64 per wave: k-loop, filter, sync, compaction, final,
 * slow-path entries, compactions, tiles). Filled only when the search ran with AK_SCAN_DBG=1.    */
int ak_index_debug_read(ak_index_t h, int64_t *out, int n);

/* Per-launch timing of the dominant kernel (the MFMA candidate scan): when
 * enabled every search records a HIP event pair around that kernel on the
 * launch stream. Read (after synchronising the stream) returns the durations in
 * ms in launch order and clears the log. Used by bench.py's roofline object.  */
int ak_index_profile(ak_index_t h, int enable);
int ak_index_profile_read(ak_index_t h, float *out_ms, int cap, int *n_out);

/* Cross-shard k-way merge: parts [g][nq][k] on device (after the
 * RCCL all-gather) -> [nq][k], comparator (distance asc, NaN last, id asc).  */
int ak_merge_topk_dev(int g, int nq, int k, const int64_t *part_ids_dev, const double *part_dist_dev,
                      int64_t *out_ids_dev, double *out_dist_dev, void *stream);

/* The row-sharded search's exchange step in one call (archi_amd/sharded.py). payload_dev: the all-gathered buffer, per
 * rank [ids nq*k int64 | float8 distance bits nq*k int64 | certificate flags nq int32, padded to a whole int64] -- what
 * ak_index_search_dev writes when its three outputs point into ONE buffer, so the local search fills the payload in
 * place --, ranks `stride` int64 elements apart (stride >= 2*nq*k + (nq+1)/2). Merges like
 * ak_merge_topk_dev and reduces the flags: out_open_dev [nq + 1] int32 = 1 for every query SOME shard could not certify
 * (the caller re-runs those on every shard with AK_SEARCH_AUTO), their count at [nq].                                */
int ak_merge_shards_dev(int g, int nq, int k, const int64_t *payload_dev, int64_t stride, int64_t *out_ids_dev,
                        double *out_dist_dev, int *out_open_dev, void *stream);

/* ---- the row-sharded search with its exchange step inside the library -------------------------------
 * One process per GPU, every rank holds one row shard in an ak_index_t. The reference has no counterpart (one Postgres
 * backend scans the whole table, postgres_vectorstore.py:317-332); these calls are what a maintainer binds -- with ctypes
 * alone, no torch.distributed -- to run that statement over the 8 GPUs of a node:
 *   rank 0:      ak_comm_unique_id(id)           128 bytes, handed to the other ranks by any channel (file, socket, MPI ...)
 *   every rank:  ak_comm_create(id, rank, world, &comm)     ncclCommInitRank on ak_init's device (RCCL over xGMI)
 *   every rank:  ak_index_search_sharded_dev(shard, comm, ...)   the same queries on every rank, the same result on every rank
 * RCCL is looked up when the first of these is called (the librccl.so already mapped into the process, else ROCm's
 * librccl.so.1); they return -12 where it is missing, the rest of the library does not need it. */
typedef void *ak_comm_t;
#define AK_COMM_ID_BYTES 128
int ak_comm_unique_id(void *out_id128);
int ak_comm_create(const void *unique_id128, int rank, int world, ak_comm_t *out);
int ak_comm_destroy(ak_comm_t c);
/* ak_index_search_dev(FAST_ONLY) on the local shard -> ONE ncclAllGather of [ids | float8 distance bits | certificate flags]
 * (nq * (16 k + 4) bytes per rank) -> ak_merge_shards_dev, all on `stream`; the flags are then read (one host synchronisation)
 * and the queries ANY shard could not certify are re-run on EVERY shard with AK_SEARCH_AUTO, exchanged and merged again --
 * every rank reads the same gathered flags and takes the same branch. On return out_ids_dev / out_dist_dev [nq][k] hold the
 * exact ORDER BY distance LIMIT k over all shards, identical on every rank; *out_rerun (may be NULL) = queries re-run.
 * queries_dev must hold the same rows on every rank; row_filter_dev / filter_len / filter_epoch describe the LOCAL shard (as
 * for ak_index_search_dev). Calls on one communicator are serialised; every rank must issue them in the same order. */
int ak_index_search_sharded_dev(ak_index_t shard, ak_comm_t comm, const float *queries_dev, int nq, int k,
                                const uint8_t *row_filter_dev, int64_t filter_len, uint64_t filter_epoch,
                                int64_t *out_ids_dev, double *out_dist_dev, int64_t *out_rerun, void *stream);

/* Failure contract of ak_index_search_sharded_dev ("one statement succeeds or fails as a whole", postgres_vectorstore.py:317-332):
 *   - a rank whose LOCAL scan fails (stale row_filter, workspace ...) still enters the all-gather, with empty rows and its code in
 *     the payload's status word: every rank returns that code after the collective;
 *   - a rank whose first MERGE fails reads the gathered flag / status words itself and joins the second all-gather whenever the
 *     others enter it (empty rows + its code: every rank returns it); if no query is open only that rank returns the error;
 *   - a failure of the exchange itself -- its buffers, a HIP copy / memset / launch / synchronise on `stream`, an RCCL error --
 *     is FATAL TO THE COMMUNICATOR: the rank returns, and every later call on that communicator returns AK_ERR_COMM_BROKEN at
 *     once instead of pairing with the wrong collective; the other ranks' collective ends by RCCL's own abort / timeout, as after
 *     the loss of a process. Destroy and re-create the communicator on every rank.
 * AK_RCCL_PATH in the environment at load names the communicator library (default: the librccl.so already mapped, else ROCm's).
 *
 * The exchange's small device steps, exported for a caller that drives the same exchange over another transport
 * (archi_amd/sharded.py over torch.distributed): no torch kernel is then needed between the local search and the result.
 *   ak_shard_payload_begin_dev   zero the flag padding and the status word of a payload [2 nq k + (nq+1)/2 + 1] int64 before the
 *                                local ak_index_search_dev writes ids / float8 bits / flags into it
 *   ak_shard_fail_payload_dev    the payload of a rank whose local search failed: no rows (id -1, NaN), every flag "certified"
 *                                (it asks for no re-run), `status` in the last word
 *   ak_shard_status_dev          out_status_dev[r] = (int) last word of rank r's payload, r < g <= 1024, payloads `stride` apart
 *   ak_shard_gather_rows_dev     out[j] = rows[idx[j]] (the open queries of a batch, [m][dim] float32)
 *   ak_shard_scatter_topk_dev    out_ids[idx[j]] = sub_ids[j], out_dist[idx[j]] = sub_dist[j] (rows of k): the re-run's rows back */
int ak_shard_payload_begin_dev(int64_t *payload_dev, int nq, int k, void *stream);
int ak_shard_fail_payload_dev(int64_t *payload_dev, int nq, int k, int status, void *stream);
int ak_shard_status_dev(int g, const int64_t *gathered_dev, int64_t stride, int *out_status_dev, void *stream);
int ak_shard_gather_rows_dev(const float *rows_dev, const int *idx_dev, int m, int dim, float *out_dev, void *stream);
int ak_shard_scatter_topk_dev(const int *idx_dev, int m, int k, const int64_t *sub_ids_dev, const double *sub_dist_dev,
                              int64_t *out_ids_dev, double *out_dist_dev, void *stream);

/* ---- L2 normalise (a3) ------------------------------------------------- */
/* encode_kwargs.normalize_embeddings (src/cli/templates/base-config.yaml:149-150) */
int ak_l2_normalize_dev(float *rows_dev, int64_t n, int dim, void *stream);

/* ---- encoder: replaces Embeddings.embed_documents / embed_query -------- */
/* manager.py:373, postgres_vectorstore.py:143,245,390 */
typedef struct AkBertConfig {
    int vocab_size;     /* 30522 */
    int hidden;         /* 384 (MiniLM-L6) / 768 (bge-base) */
    int layers;         /* 6 / 12 */
    int heads;          /* 12 */
    int intermediate;   /* 1536 / 3072 */
    int max_position;   /* 512 */
    int type_vocab;     /* 2 */
    float ln_eps;       /* 1e-12 */
    int residual_bf16;  /* 0: fp32 residual stream between layers (reference-like); 1: the residual stream is kept
                         * in bf16 only (hidden 384 path): 60% less epilogue traffic, +~1e-6 cosine deviation from
                         * the fp32 reference on top of the bf16 GEMM inputs */
    int precision;      /* 0: bf16 MFMA GEMMs (the measured path). 1: fp32 PARITY MODE -- every matrix in `weights_dev` is then
                         * float32 (same order and shapes) and all arithmetic is float32: GEMMs and attention on
                         * v_mfma_f32_32x32x2_f32 (csrc/encoder_f32.hip; 0.70 / 0.77 of the 157 TFLOP/s float32 matrix roof),
                         * exact erf GELU, fp32 LayerNorm / softmax: the reference's CPU embedder (torch fp32, manager.py:373)
                         * to ~1e-6, at ~1/9 of the bf16 rate. 2: SPLIT-bf16 PARITY MODE ("bf16x3") -- float32 weights as for 1;
                         * every GEMM operand is split x = hi + lo (two bf16 values, lo = bf16(x - hi)) and every product runs
                         * as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with one fp32 accumulator (~2^-16 per
                         * product), everything between the GEMMs in fp32 as for 1: fp32-grade embeddings at ~1/3 of the bf16
                         * rate. Batches of >= 16 384 tokens (hidden >= 512; 20 480 below) run on the bf16 path's GEMM tiles (csrc/gemm.hip MODE 5 / 6: the
                         * activations travel as bf16 [hi | lo] rows, the K-loop walks 3 K; GELU by a cubic table of the normal
                         * CDF, max error 5e-7), smaller ones on csrc/encoder_f32.hip's k3_gemm; both held to 1e-5 by the tests. */
} AkBertConfig;

/* Weight order (all device pointers, bf16 matrices (float32 when cfg->precision == 1) row-major [out][in] exactly
 * as torch.nn.Linear.weight, fp32 vectors):
 *   0 word_emb [vocab][H] bf16, 1 pos_emb [max_pos][H] bf16, 2 type_emb [type][H] bf16,
 *   3 emb_ln_g [H] f32, 4 emb_ln_b [H] f32,
 *   per layer l (base 5 + 16*l):
 *     +0 wq +1 bq +2 wk +3 bk +4 wv +5 bv +6 wo +7 bo +8 ln1_g +9 ln1_b
 *     +10 w1 [I][H] +11 b1 [I] +12 w2 [H][I] +13 b2 [H] +14 ln2_g +15 ln2_b      */
int ak_encoder_create(const AkBertConfig *cfg, const void *const *weights_dev, int n_weights,
                      ak_encoder_t *out);
int ak_encoder_destroy(ak_encoder_t h);
/* ids/mask: [B][S] int32 on device; out: [B][H] float32 on device. Asynchronous on `stream`, with one exception: under
 * AK_QUERY_FUSED=1 (opt-in, hidden 384, B * S <= 64) the forward pass runs as ONE launch confined to one XCD (csrc/query_forward.hip)
 * whose every wait is bounded; the call then synchronises `stream` to read its failure word and, had a wait given up, re-runs the
 * pass through the ordinary launches -- same rows bit for bit either way. Measured slower than the ordinary launches on MI355X
 * (DESIGN.md section 4), hence opt-in. */
int ak_encoder_forward(ak_encoder_t h, const int32_t *ids_dev, const int32_t *mask_dev, int B, int S,
                       int pooling, int normalise, float *out_dev, void *stream);
/* The same forward pass for RIGHT-PADDED rows given by their lengths -- what a tokenizer emits and what the provider's
 * length-sorted tiles hold (replaces the per-tile mask assembly the Python side did with torch kernels; round-4 review):
 *   ids_dev   [B] rows of S token ids, `ld_ids` int32 apart (>= S); whatever lies past a row's length is ignored
 *   lens_dev  one int32 per row, `lens_stride` int32 apart (the tiles carry it as column S of the id rows: ld_ids = S + 1,
 *             lens_dev = ids_dev + S, lens_stride = S + 1); clamped to [0, S]; a row of length 0 embeds to zeros
 *   out_dev   [B][H] float32: the caller passes the address of the tile's first row inside ONE result buffer.
 * The library lays the 0 / 1 mask out itself (one small launch) and runs ak_encoder_forward's kernels on it: results are
 * bit-identical to ak_encoder_forward on the explicit mask. S a multiple of 32, <= 512. */
int ak_encoder_forward_lens(ak_encoder_t h, const int32_t *ids_dev, int ld_ids, const int32_t *lens_dev, int lens_stride, int B, int S,
                            int pooling, int normalise, float *out_dev, void *stream);

/* The 8192-entry bf16 table the fused hidden-384 layer kernel and the wide FFN-up tile read their GELU from (csrc/gelu_table.h):
 * entry i = bf16(gelu(v)), v = the MIDPOINT of the IEEE half bit patterns [8 i, 8 i + 8) (sign, 5 exponent bits, 7 mantissa bits;
 * the lookup truncates, so the midpoint halves its error), exact erf GELU
 * (the activation of the reference's default embedder, all-MiniLM-L6-v2, inside Embeddings.embed_documents, manager.py:373).
 * Host only -- no GPU work; exported so that the CPU suite can hold the table to the exact function. */
int ak_encoder_gelu_table(uint16_t *out8192);

/* ---- host tokenizer: the tokenisation step inside Embeddings.embed_documents -------- */
/* manager.py:373 -> HuggingFaceEmbeddings -> sentence-transformers' BERT WordPiece tokenizer [upstream]. Pure host
 * code (no GPU work): multi-threaded, so that text -> token ids keeps up with ak_encoder_forward at ingestion.
 * vocab_path: the checkpoint's vocab.txt (one token per line, id = line number; needs [CLS] [SEP] [UNK]). */
typedef void *ak_wordpiece_t;
int ak_wordpiece_create(const char *vocab_path, int lowercase, ak_wordpiece_t *out);
int ak_wordpiece_destroy(ak_wordpiece_t h);
/* n texts as one UTF-8 blob, text i = blob[offsets[i] : offsets[i+1]]. out_ids: [n][max_len] int32, zero padded;
 * out_len[i] = ids of text i including [CLS] and [SEP], truncated to max_len (the last kept id is [SEP]), or -1 when
 * the text holds a byte >= 0x80 or a literal special token such as "[SEP]": those need the full Unicode tokenizer
 * and are left to the caller. threads <= 0: all host cores. */
int ak_wordpiece_encode(ak_wordpiece_t h, const char *blob, const int64_t *offsets, int64_t n, int max_len,
                        int threads, int32_t *out_ids, int32_t *out_len);

#ifdef __cplusplus
}
#endif
#endif /* ARCHI_KNN_H */
