"""GPU parity for the hybrid combine: ak_index_distances against the oracle, and
ArchiHipHybridVectorStore.hybrid_search against the reference's formula evaluated over ALL rows
(/root/reference/src/data_manager/vectorstore/postgres_vectorstore.py:435-457) with oracle distances."""
import numpy as np
import pytest

from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("metric", ["cosine", "l2", "inner_product"])
def test_distances_entry_point_matches_oracle(hip, dtype, metric):
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(17)
    n, d = 3000, 200                       # 200: not a multiple of 64 -> exercises the generic row loop
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[5] = 0.0                          # zero vector: NaN cosine distance
    ids = (rng.permutation(10 * n)[:n] + 7).astype(np.int64)
    ix = HipIndex(d, n, dtype=dtype, metric=metric, device=0)
    ix.add(rows, ids=ids)
    ix.remove(ids[10:20])
    stored = ko.round_through(rows, dtype)
    q = rng.standard_normal(d).astype(np.float32)
    ask = np.concatenate([ids[:40], [999_999_999, 3]])          # 10 deleted, 2 unknown
    dist, found = ix.distances(q, ask)
    assert found.tolist() == [True] * 10 + [False] * 10 + [True] * 20 + [False, False]
    for j, i in enumerate(ask):
        if found[j]:
            want = ko.distance(metric, stored[int(np.nonzero(ids == i)[0][0])], q)
            assert (dist[j] == want) or (np.isnan(dist[j]) and np.isnan(want)), (j, dist[j], want)
        else:
            assert np.isnan(dist[j])
    assert ix.distances(q, [])[0].shape == (0,)
    ix.close()


class _TableBm25:
    def __init__(self, hits):
        self.hits = hits

    def scores(self, query, table):
        return dict(self.hits)


class _Emb:
    def __init__(self, dim):
        self.dim = dim

    def embed_documents(self, texts):
        return [[float(x) for x in r] for r in ko.gen_rows(11, 5, 0, len(texts), self.dim, True, "f32")]

    def embed_query(self, text):
        return [float(x) for x in ko.gen_rows(11, 6, len(text), 1, self.dim, True, "f32")[0]]


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_hybrid_search_equals_reference_formula_over_all_rows(hip, sign):
    from archi_amd import vectorstore as vs
    from archi_amd.vectorstore import ArchiHipHybridVectorStore
    vs.reset_collections()
    n, d, k = 20000, 384, 10              # large enough for the certified MFMA scan on the non-hit leg
    emb = _Emb(d)
    rng = np.random.default_rng(3)
    hit_rows = rng.choice(np.arange(1, n + 1), size=1500, replace=False)
    hits = {int(r): float(sign * rng.uniform(0.1, 6.0)) for r in hit_rows}
    store = ArchiHipHybridVectorStore({"hip": {"dtype": "bf16", "capacity": n}}, emb, collection_name="hyb",
                                      distance_metric="cosine", bm25=_TableBm25(hits))
    store.add_texts([f"t{i}" for i in range(n)], [{"page": i % 4} for i in range(n)])
    col = store._collection()
    rids = np.arange(1, n + 1, dtype=np.int64)
    stored = col.index.fetch(col.index.lookup(rids))
    for kwargs in ({}, {"filter": {"page": 2}}):
        for w_s, w_b in ((0.7, 0.3), (0.5, 0.5), (1.0, 0.0)):
            q = "what is the luminosity?"
            got = store.hybrid_search(q, k=k, semantic_weight=w_s, bm25_weight=w_b, **kwargs)
            qv = np.asarray(emb.embed_query(q), np.float32)
            alive = np.ones(n, np.uint8)
            if kwargs:
                alive[(rids - 1) % 4 != 2] = 0
            oi, od, cnt = ko.search(stored, qv, n, "cosine", ids=rids, alive=alive)
            want = sorted((((1.0 - float(od[0, j])) * w_s + hits.get(int(oi[0, j]), 0.0) * w_b, int(oi[0, j]))
                           for j in range(int(cnt[0]))), key=lambda c: (-c[0], c[1]))[:k]
            assert [(doc.page_content, s) for doc, s in got] == [(f"t{rid - 1}", s) for s, rid in want]
    vs.reset_collections()
