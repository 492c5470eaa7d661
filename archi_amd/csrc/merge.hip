// merge.hip -- cross-shard k-way merge of per-shard partial top-k
// (SURVEY.md section 8e): after the RCCL all-gather every rank holds
// parts [g][nq][k]; the merged top-k uses the same (distance asc, NaN last,
// id asc) comparator as every other stage, so the result is identical for any
// shard count.
#include "index.h"

namespace ak {

__global__ void k_merge_prep(int g, int nq, int k, const int64_t *__restrict__ pids,
                             const double *__restrict__ pdist, uint64_t *__restrict__ keys,
                             int64_t *__restrict__ ids) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int total = g * nq * k;
    if (t >= total) return;
    int j = t % k, qi = (t / k) % nq, s = t / (k * nq);
    int64_t id = pids[t];
    int64_t o = ((int64_t)qi * g + s) * k + j;
    keys[o] = id < 0 ? KEY_INVALID : dist_key(pdist[t]);
    ids[o] = id;
}

// The all-gathered payload of the row-sharded search: per rank [ids Q*k int64 | float8 bits Q*k int64 | cert Q int32, padded to
// a whole int64] -- exactly what ak_index_search_dev writes when its three outputs point into one buffer --, ranks `stride`
// int64 elements apart. Keys for the selection + per query "open" = some shard could not certify it.
__global__ void k_merge_prep_payload(int g, int nq, int k, const int64_t *__restrict__ payload, int64_t stride,
                                     uint64_t *__restrict__ keys, int64_t *__restrict__ ids) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int total = g * nq * k;
    if (t >= total) return;
    int j = t % k, qi = (t / k) % nq, s = t / (k * nq);
    const int64_t *part = payload + (int64_t)s * stride;
    int64_t id = part[(int64_t)qi * k + j];
    union { int64_t i; double d; } v; v.i = part[(int64_t)nq * k + (int64_t)qi * k + j];
    int64_t o = ((int64_t)qi * g + s) * k + j;
    keys[o] = id < 0 ? KEY_INVALID : dist_key(v.d);
    ids[o] = id;
}
__global__ void k_open_flags(int g, int nq, int k, const int64_t *__restrict__ payload, int64_t stride, int *__restrict__ open) {
    int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    int o = 0;
    for (int s = 0; s < g; s++) o |= ((const int *)(payload + (int64_t)s * stride + 2 * (int64_t)nq * k))[qi] == 0;   // flags: int32, packed
    open[qi] = o;
    if (o) atomicAdd(&open[nq], 1);
}

}  // namespace ak

using namespace ak;

// Row-sharded search, after the RCCL all-gather (archi_amd/sharded.py): merge the G partial lists and reduce the
// certificate flags. out_open_dev [nq + 1] int32: 1 for every query some shard left uncertified, their count at [nq].
extern "C" int ak_merge_shards_dev(int g, int nq, int k, const int64_t *payload_dev, int64_t stride, int64_t *out_ids_dev,
                                   double *out_dist_dev, int *out_open_dev, void *stream) {
    AK_BIND();
    if (g <= 0 || nq <= 0 || k <= 0 || !payload_dev || !out_ids_dev || !out_dist_dev || !out_open_dev)
        AK_FAIL(-1, "ak_merge_shards_dev: bad arguments");
    if (stride < 2 * (int64_t)nq * k + (nq + 1) / 2) AK_FAIL(-1, "ak_merge_shards_dev: stride shorter than one payload");
    hipStream_t st = (hipStream_t)stream;
    int64_t n_in = (int64_t)g * k;
    size_t kb = (size_t)nq * n_in * 8, ob = (size_t)nq * k * 8, sb = select_scratch_bytes(nq, n_in, k);
    char *blk;
    AK_HIP(hipMallocAsync((void **)&blk, 2 * kb + 2 * ob + sb + 1024, st));
    uint64_t *keys = (uint64_t *)blk;
    int64_t *ids = (int64_t *)(blk + kb);
    uint64_t *ok = (uint64_t *)(blk + 2 * kb);
    int64_t *oi = (int64_t *)(blk + 2 * kb + ob);
    void *scratch = blk + 2 * kb + 2 * ob;
    int total = g * nq * k;
    AK_HIP(hipMemsetAsync(out_open_dev + nq, 0, 4, st));
    k_open_flags<<<(nq + 255) / 256, 256, 0, st>>>(g, nq, k, payload_dev, stride, out_open_dev);
    k_merge_prep_payload<<<(total + 255) / 256, 256, 0, st>>>(g, nq, k, payload_dev, stride, keys, ids);
    AK_HIP(hipGetLastError());
    int rc = select_topk(keys, ids, nullptr, nq, n_in, k, ok, oi, scratch, st);
    if (!rc) rc = emit_results(ok, oi, nq, k, out_ids_dev, out_dist_dev, nullptr, st);
    AK_HIP(hipFreeAsync(blk, st));
    return rc;
}

extern "C" int ak_merge_topk_dev(int g, int nq, int k, const int64_t *part_ids_dev, const double *part_dist_dev,
                                 int64_t *out_ids_dev, double *out_dist_dev, void *stream) {
    AK_BIND();
    if (g <= 0 || nq <= 0 || k <= 0) AK_FAIL(-1, "ak_merge_topk_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int64_t n_in = (int64_t)g * k;
    size_t kb = (size_t)nq * n_in * 8, ob = (size_t)nq * k * 8, sb = select_scratch_bytes(nq, n_in, k);
    char *blk;
    AK_HIP(hipMallocAsync((void **)&blk, 2 * kb + 2 * ob + sb + 1024, st));
    uint64_t *keys = (uint64_t *)blk;
    int64_t *ids = (int64_t *)(blk + kb);
    uint64_t *ok = (uint64_t *)(blk + 2 * kb);
    int64_t *oi = (int64_t *)(blk + 2 * kb + ob);
    void *scratch = blk + 2 * kb + 2 * ob;
    int total = g * nq * k;
    k_merge_prep<<<(total + 255) / 256, 256, 0, st>>>(g, nq, k, part_ids_dev, part_dist_dev, keys, ids);
    AK_HIP(hipGetLastError());
    int rc = select_topk(keys, ids, nullptr, nq, n_in, k, ok, oi, scratch, st);
    if (!rc) rc = emit_results(ok, oi, nq, k, out_ids_dev, out_dist_dev, nullptr, st);
    AK_HIP(hipFreeAsync(blk, st));
    return rc;
}
