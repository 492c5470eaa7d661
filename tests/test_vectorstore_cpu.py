"""CPU suite: ArchiHipVectorStore's host logic against (a) the golden fixture captured by running
the reference's own PostgresVectorStore (tests/golden/make_reference_fixtures.py) and (b) the
behaviours the reference's unit tests pin (tests/unit/test_postgres_vectorstore.py)."""
import json
import os
from unittest.mock import MagicMock

import numpy as np
import pytest

from archi_amd import vectorstore as vs
from archi_amd.vectorstore import ArchiHipVectorStore, Document
from oracle import knn_oracle as ko
from tests.fake_index import OracleIndex

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_wrapper.json")))


def factory(dim, capacity, dtype, metric):
    return OracleIndex(dim, capacity, dtype=dtype, metric=metric)


@pytest.fixture(autouse=True)
def fresh():
    vs.reset_collections()
    yield
    vs.reset_collections()


class FixedEmbeddings:
    def __init__(self, dim, seed):
        self.dim, self.seed = dim, seed

    def embed_documents(self, texts):
        return [[float(x) for x in r] for r in ko.gen_rows(self.seed, 5, 0, len(texts), self.dim, True, "f32")]

    def embed_query(self, text):
        return [float(x) for x in ko.gen_rows(self.seed, 6, len(text), 1, self.dim, True, "f32")[0]]


def _load_golden_db(store, n, dim, seed):
    """Same synthetic table as make_reference_fixtures.build_db, loaded behind the store."""
    vec = ko.gen_rows(seed, 0, 0, n, dim, True, "f32")
    vec[17] = vec[3]
    col = store._collection(dim)
    t = col.table
    ids = []
    for i in range(n):
        md = {"collection": "golden", "source": "web" if i % 3 else "git", "page": i % 7}
        if i % 11 == 0:
            md = {}
        rid = 1000 + i
        t.rows[rid] = {"document_id": rid, "chunk_index": 0, "text": f"chunk {i}", "metadata": md}
        t.register_document(rid, resource_hash=f"h{i // 4}" if i % 5 else None,
                            display_name=f"Doc {i // 4}" if i % 5 else None,
                            source_type="web" if i % 2 else None, url=f"https://x/{i // 4}" if i % 6 == 0 else None,
                            is_deleted=(i % 13 == 0))
        ids.append(rid)
    col.index.add(vec, ids=ids)


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"{c['metric']}-{json.dumps(c['kwargs'])}")
def test_replays_reference_wrapper_fixture(case):
    emb = FixedEmbeddings(case["dim"], case["seed"])
    store = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="golden",
                                distance_metric=case["metric"], index_factory=factory)
    _load_golden_db(store, case["n_rows"], case["dim"], case["seed"])
    res = store.similarity_search_with_score(case["query_text"], k=case["k"], **case["kwargs"])
    got = [{"page_content": d.page_content, "metadata": d.metadata, "score": s} for d, s in res]
    assert got == case["results"]          # contents, merged metadata, order and float scores identical


def test_write_path_matches_reference_fixture():
    w = GOLD["write"]
    emb = FixedEmbeddings(48, 2024)
    store = ArchiHipVectorStore({}, emb, collection_name="golden", index_factory=factory)
    assert store.add_texts([]) == w["empty_add"]
    metas = [{"a": 1}, {"b": 2}]
    ids = store.add_texts(["t0", "t1"], metadatas=metas, ids=["id-0", "id-1"], document_id=42)
    assert ids == w["returned_ids"] and metas == w["metadatas_after"]
    assert store.count() == 2
    assert store.delete() is w["delete_none"]
    assert store.delete(ids=["id-0"]) is w["delete_ids"] and store.count() == 1
    assert store.delete(document_id=42) is w["delete_doc"] and store.count() == 0
    with pytest.raises(ValueError) as e:
        ArchiHipVectorStore({}, emb, distance_metric="manhattan")
    assert str(e.value) == GOLD["bad_metric_error"]


# ---- behaviours pinned by the reference's unit tests -----------------------------------------
@pytest.fixture
def mock_embeddings():
    e = MagicMock()
    e.embed_documents.return_value = [[0.1, 0.2, 0.3] * 128]
    e.embed_query.return_value = [0.1, 0.2, 0.3] * 128
    return e


def test_init_operator_map_and_errors(mock_embeddings):       # reference tests :87-137
    s = ArchiHipVectorStore({}, mock_embeddings)
    assert (s._collection_name, s._distance_metric, s._distance_op) == ("default", "cosine", "<=>")
    assert ArchiHipVectorStore({}, mock_embeddings, distance_metric="l2")._distance_op == "<->"
    assert ArchiHipVectorStore({}, mock_embeddings, distance_metric="inner_product")._distance_op == "<#>"
    with pytest.raises(ValueError, match="distance_metric must be one of"):
        ArchiHipVectorStore({}, mock_embeddings, distance_metric="invalid")
    assert s.embeddings is mock_embeddings
    assert not hasattr(s, "hybrid_search")      # HybridRetriever falls back to the semantic leg


def test_add_and_search_score_is_one_minus_distance(mock_embeddings):   # :187-210, :396-412, :452-474
    s = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, mock_embeddings, collection_name="test_collection",
                            index_factory=factory)
    texts = ["Document about machine learning"]
    ids = s.add_texts(texts, metadatas=[{"source": "web", "page": 5, "custom_field": "custom_value"}])
    mock_embeddings.embed_documents.assert_called_once_with(texts)
    assert len(ids) == 1
    s.table.register_document(7, resource_hash="hash123", display_name="Test Doc")
    s.table.update_row(list(s.table.rows)[0], document_id=7)
    res = s.similarity_search_with_score("query", k=5)
    assert len(res) == 1
    doc, score = res[0]
    assert isinstance(doc, Document) and "machine learning" in doc.page_content
    v = np.array([0.1, 0.2, 0.3] * 128, np.float32)
    assert score == 1.0 - ko.distance("cosine", v, v) and abs(score - 1.0) < 1e-6
    assert doc.metadata["source"] == "web" and doc.metadata["page"] == 5
    assert doc.metadata["resource_hash"] == "hash123" and doc.metadata["display_name"] == "Test Doc"
    assert [d.page_content for d in s.similarity_search("query", k=5)] == [doc.page_content]
    assert s.similarity_search_by_vector([0.1, 0.2, 0.3] * 128, k=1)[0].page_content == doc.page_content


def test_empty_store_and_edge_queries(mock_embeddings):       # :305-312, :593-634
    s = ArchiHipVectorStore({}, mock_embeddings, index_factory=factory)
    assert s.similarity_search("anything", k=5) == [] and s.count() == 0
    s.add_texts(["a"], metadatas=None)
    for q in ("", "What's the \"best\" way? <script>", "机器学习 🚀"):
        assert isinstance(s.similarity_search(q, k=1000), list)


def test_add_documents_upsert_and_from_texts(mock_embeddings):  # :414-429, ON CONFLICT semantics :173-176
    mock_embeddings.embed_documents.side_effect = lambda texts: [[float(i + 1), 0.0, 1.0] for i, _ in enumerate(texts)]
    mock_embeddings.embed_query.return_value = [1.0, 0.0, 1.0]
    s = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, mock_embeddings, index_factory=factory)
    docs = [Document(page_content="First document", metadata={"source": "test1"}),
            Document(page_content="Second document", metadata={"source": "test2"})]
    assert len(s.add_documents(docs, document_id=9)) == 2 and s.count() == 2
    s.add_documents([Document(page_content="First v2", metadata={})], document_id=9)   # same (doc, chunk 0)
    assert s.count() == 2
    assert "First v2" in [d.page_content for d in s.similarity_search("q", k=5)]
    assert "First document" not in [d.page_content for d in s.similarity_search("q", k=5)]
    s2 = ArchiHipVectorStore.from_texts(["x", "y"], mock_embeddings, pg_config={}, collection_name="other",
                                        index_factory=factory)
    assert s2.count() == 2 and s.count() == 2
    with pytest.raises(KeyError):
        ArchiHipVectorStore.from_texts(["x"], mock_embeddings)


def test_store_instances_share_the_process_level_index(mock_embeddings):   # archi.py:61-65
    a = ArchiHipVectorStore({}, mock_embeddings, collection_name="c", index_factory=factory)
    a.add_texts(["one"])
    b = ArchiHipVectorStore({}, mock_embeddings, collection_name="c", index_factory=factory)
    assert b.count() == 1 and len(b.similarity_search("q", k=3)) == 1


# ---- hybrid_search ------------------------------------------------------------
class TableBm25:
    """BM25 leg as a fixed {row id: score} table (the scorer is pluggable; the fixture pins the combine)."""

    def __init__(self, hits):
        self.hits = {int(k): v for k, v in hits.items()}

    def scores(self, query, table):
        return dict(self.hits)


@pytest.mark.parametrize("case", GOLD["hybrid"], ids=lambda c: f"{c['metric']}-{c['semantic_weight']}-{json.dumps(c['kwargs'])}")
def test_hybrid_replays_reference_fixture(case):
    from archi_amd.vectorstore import ArchiHipHybridVectorStore
    emb = FixedEmbeddings(case["dim"], case["seed"])
    store = ArchiHipHybridVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="golden",
                                      distance_metric=case["metric"], index_factory=factory,
                                      bm25=TableBm25(case["bm25_hits"]))
    _load_golden_db(store, case["n_rows"], case["dim"], case["seed"])
    res = store.hybrid_search(case["query_text"], k=case["k"], semantic_weight=case["semantic_weight"],
                              bm25_weight=case["bm25_weight"], **case["kwargs"])
    got = [{"page_content": d.page_content, "metadata": d.metadata, "score": s} for d, s in res]
    assert got == case["results"]          # the reference's own combine over ALL rows, scores bit-equal


def _load_nan_db(store, case):
    _load_golden_db(store, case["n_rows"], case["dim"], case["seed"])
    # the fixture's table has two zero-vector rows: replace them (ON CONFLICT semantics are not what is under test here)
    col = store._collection()
    for i in case["zero_rows"]:
        rid = 1000 + i
        col.index.remove([rid])
        col.index.add(np.zeros((1, case["dim"]), np.float32), ids=[rid])
        col.table.suspects.add(rid)


def test_hybrid_orders_nan_scores_first_like_postgres_desc():
    """Zero-vector rows have a NaN cosine distance -> NaN combined score; PostgreSQL's ORDER BY combined DESC ranks float8
    NaN above every number. Reference run recorded in reference_wrapper.json["hybrid_nan"] (scores None = NaN)."""
    from archi_amd.vectorstore import ArchiHipHybridVectorStore
    case = GOLD["hybrid_nan"]
    emb = FixedEmbeddings(case["dim"], case["seed"])
    store = ArchiHipHybridVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="golden", distance_metric="cosine",
                                      index_factory=factory, bm25=TableBm25(case["bm25_hits"]))
    _load_nan_db(store, case)
    res = store.hybrid_search(case["query_text"], k=case["k"], semantic_weight=case["semantic_weight"], bm25_weight=case["bm25_weight"])
    got = [{"page_content": d.page_content, "metadata": d.metadata, "score": None if s != s else s} for d, s in res]
    assert got == case["results"] and got[0]["score"] is None and got[1]["score"] is None
    # rows inserted through add_texts are flagged by themselves
    vs.reset_collections()
    s2 = ArchiHipHybridVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="z", index_factory=factory, bm25=TableBm25({}))
    s2.add_texts(["a", "zero", "b"], embeddings=np.array([[1, 0, 0], [0, 0, 0], [0, 1, 0]], np.float32))
    assert len(s2.table.suspects) == 1


def test_add_texts_is_one_transaction():
    """a failing index add must leave the previous chunks of the document, the (document, chunk) map and the
    caches untouched (the reference upserts inside one database transaction, postgres_vectorstore.py:168-182)."""
    emb = FixedEmbeddings(8, 5)
    s = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="tx", index_factory=factory)
    s.add_texts(["old0", "old1"], ids=["a", "b"], document_id=1)
    col = s._collection()
    before = (dict(col.table.rows), dict(col.table.by_doc_chunk), s.count())

    def boom(*a, **k):
        raise RuntimeError("ak_index_add failed")
    orig = col.index.add
    col.index.add = boom
    with pytest.raises(RuntimeError):
        s.add_texts(["new0", "new1", "new2"], document_id=1)
    col.index.add = orig
    assert (dict(col.table.rows), dict(col.table.by_doc_chunk), s.count()) == before
    assert sorted(d.page_content for d in s.similarity_search("q", k=5)) == ["old0", "old1"]
    with pytest.raises(ValueError, match="holds 8-d vectors"):            # width checked before anything is touched
        s.add_texts(["x"], embeddings=np.zeros((1, 9), np.float32), document_id=1)
    assert (dict(col.table.rows), dict(col.table.by_doc_chunk), s.count()) == before
    assert s.add_texts(["new0"], ids=["keep-my-id"], document_id=1) == ["keep-my-id"] and s.count() == 2


def test_hybrid_errors_fallback_and_retriever_contract():
    from archi_amd.vectorstore import ArchiHipHybridVectorStore, HostBm25
    emb = FixedEmbeddings(48, 2024)
    assert not hasattr(ArchiHipVectorStore({}, emb, index_factory=factory), "hybrid_search")   # hybrid_retriever.py:55-62
    store = ArchiHipHybridVectorStore({}, emb, collection_name="golden", index_factory=factory)
    with pytest.raises(RuntimeError) as e:                                   # reference test :281-290
        store.hybrid_search("q", k=3)
    assert str(e.value) == GOLD["hybrid_no_index_error"] and "BM25 index" in str(e.value)
    store = ArchiHipHybridVectorStore({"hip": {"bm25": HostBm25()}}, emb, collection_name="golden", index_factory=factory)
    assert store.hybrid_search("q", k=3) == GOLD["hybrid_empty_table"] == []  # empty table -> semantic fallback -> []


def _scalar_bm25(table, query, k1=1.2, b=0.75, sign=1.0):
    """Okapi BM25 from scratch over the live rows, one Python float operation at a time (the formulation HostBm25 vectorises)."""
    import math
    import re
    tok = re.compile(r"\w+")
    post, lens = {}, {}
    for rid in table.live_rids().tolist():
        toks = tok.findall(table.text_at(table.pos(rid)).lower())
        lens[rid] = len(toks)
        for w in toks:
            d = post.setdefault(w, {})
            d[rid] = d.get(rid, 0) + 1
    n = len(lens)
    avg = (sum(lens.values()) / n) if n else 0.0
    out = {}
    for w in dict.fromkeys(tok.findall(query.lower())):
        plist = post.get(w)
        if not plist:
            continue
        idf = math.log(1.0 + (n - len(plist) + 0.5) / (len(plist) + 0.5))
        for rid, tf in plist.items():
            norm = tf + k1 * (1.0 - b + b * lens[rid] / avg)
            out[rid] = out.get(rid, 0.0) + idf * tf * (k1 + 1.0) / norm
    return {rid: sign * v for rid, v in out.items()}


def test_host_bm25_index_follows_appends_deletes_replacements_and_vacuum():
    """HostBm25 indexes only the positions appended since its last call and applies deletes at query time; after every kind of
    table change its scores equal a from-scratch scalar BM25 over the live rows, bit for bit."""
    from archi_amd.chunktable import ChunkTable
    from archi_amd.vectorstore import HostBm25
    rng = np.random.default_rng(3)
    vocab = [f"w{i}" for i in range(40)] + ["the", "The", "muon"]
    def text():
        return " ".join(rng.choice(vocab, size=int(rng.integers(0, 30))))
    t = ChunkTable()
    bm = HostBm25(sign=-1.0)
    assert bm.scores("the muon", t) == {}
    for step in range(12):
        for doc in range(step * 3, step * 3 + 3):
            n = int(rng.integers(1, 6))
            t.append_block(doc, [text() for _ in range(n)], [{} for _ in range(n)])
        if step % 3 == 1:                                  # delete a document, replace another one's chunk
            for rid in t.rids_of_document(step):
                t.kill(rid)
            old = t.find(step + 1, 0)
            if old is not None:
                t.kill(old)
                t.append(t.next_id, step + 1, 0, "the muon " + text(), {})
        if step == 7:
            t.update_row(t.live_rids()[0], text="muon muon rewritten in place")
        if step == 9:
            t.vacuum()
        for q in ("the muon w3", "w1 w1 w39", "absent", ""):
            got, want = bm.scores(q, t), _scalar_bm25(t, q, sign=-1.0)
            assert got == want, (step, q)
            pos, sc = bm.scores_arrays(q, t)
            assert np.all(np.diff(pos) > 0) and t.rids_at(pos).tolist() == sorted(want)
    for rid in t.live_rids().tolist():                     # everything deleted: no live row, no score, no division by zero
        t.kill(rid)
    assert bm.scores("the muon", t) == {}


def test_hybrid_host_bm25_equals_brute_force_over_all_rows():
    from archi_amd.vectorstore import ArchiHipHybridVectorStore, HostBm25
    texts = ["muon detector calibration run", "the muon chamber alignment", "tracker alignment and calibration",
             "calorimeter energy scale", "muon muon muon trigger rates", "software release notes", "grid job submission"] * 6
    texts = [f"{t} #{i}" for i, t in enumerate(texts)]
    emb = FixedEmbeddings(48, 99)
    for sign in (1.0, -1.0):
        vs.reset_collections()
        bm = HostBm25(sign=sign)
        store = ArchiHipHybridVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="c", index_factory=factory, bm25=bm)
        store.add_texts(texts, [{"page": i % 3} for i in range(len(texts))])
        col = store._collection()
        for kwargs in ({}, {"filter": {"page": 1}}):
            q = "muon alignment"
            got = store.hybrid_search(q, k=8, semantic_weight=0.6, bm25_weight=0.4, **kwargs)
            qv = np.asarray(emb.embed_query(q), np.float32)
            hits = bm.scores(q, col.table)
            assert 0 < len(hits) < len(texts)
            want = []
            for rid, r in col.table.rows.items():
                if kwargs and r["metadata"]["page"] != 1:
                    continue
                d = ko.distance("cosine", col.index.rows[col.index.lookup([rid])[0]], qv)
                want.append(((1.0 - d) * 0.6 + hits.get(rid, 0.0) * 0.4, rid))
            want.sort(key=lambda c: (-c[0], c[1]))
            assert [(d.page_content, s) for d, s in got] == [(col.table.rows[rid]["text"], s) for s, rid in want[:8]]
    # idf/tf sanity of the stand-in scorer: the triple-'muon' chunk outranks single mentions
    s = HostBm25().scores("muon", col.table)
    best = max(s, key=s.get)
    assert "muon muon muon" in col.table.rows[best]["text"]


# ---- what the reference's retrievers get (semantic_retriever.py:39, grading_retriever.py:25, hybrid_retriever.py:85-103)
def _load_retriever_db(store, n, dim, seed):
    """make_reference_fixtures.py's retriever table: the golden table with `filename` on every row."""
    _load_golden_db(store, n, dim, seed)
    t = store._collection().table
    for rid, r in t.rows.items():
        i = rid - 1000
        md = r["metadata"] or {"collection": "golden"}
        md["filename"] = f"file{i // 4}.txt"
        t.update_row(rid, metadata=md)


def test_store_answers_the_retrievers_calls_like_the_reference_store():
    """The fixture generator ran the reference's Semantic/Grading/HybridRetriever over the reference store and over
    this build's store and asserted identical output; here the recorded reference-side output is replayed through
    the exact calls those retrievers make."""
    from archi_amd.vectorstore import ArchiHipHybridVectorStore
    R = GOLD["retrievers"]
    assert R["reference_retrievers_over_archi_store_identical"] is True
    emb = FixedEmbeddings(48, 2024)
    q = R["query_text"]

    def dump(res):
        return [{"page_content": d.page_content, "metadata": d.metadata, "score": s} for d, s in res]

    store = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="golden", index_factory=factory)
    _load_retriever_db(store, 300, 48, 2024)
    assert dump(store.similarity_search_with_score(q, k=3)) == R["semantic_k3"]            # SemanticRetriever
    assert [{"page_content": d.page_content, "metadata": d.metadata} for d in store.similarity_search(q, k=3)] == R["grading_k3"]
    assert not hasattr(store, "hybrid_search")
    assert dump(store.similarity_search_with_score(q, k=5)) == R["hybrid_fallback_k5"]     # HybridRetriever, no hybrid_search
    vs.reset_collections()
    hstore = ArchiHipHybridVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="golden", index_factory=factory,
                                       bm25=TableBm25(R["bm25_hits"]))
    _load_retriever_db(hstore, 300, 48, 2024)
    assert dump(hstore.hybrid_search(query=q, k=5, semantic_weight=0.5, bm25_weight=0.5)) == R["hybrid_native_k5"]


def test_where_mask_is_cached_until_rows_or_documents_change():
    """The WHERE clause (:296-310) is a host pass over every row: its mask is reused for repeated filters and dropped
    when a row is added / deleted or a document is soft-deleted."""
    emb = FixedEmbeddings(16, 3)
    s = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="cache", index_factory=factory)
    s.add_texts([f"t{i}" for i in range(6)], metadatas=[{"source": "web" if i % 2 else "git"} for i in range(6)],
                document_id=1)
    s.add_texts(["other"], metadatas=[{"source": "web"}], document_id=2)
    col = s._collection()
    calls = []
    orig = col.table.positions_matching
    col.table.positions_matching = lambda *a, **k: calls.append(1) or orig(*a, **k)
    r1 = s.similarity_search("q", k=10, filter={"source": "web"})
    n1 = len(calls)
    r2 = s.similarity_search("q", k=10, filter={"source": "web"})
    assert n1 > 0 and len(calls) == n1 and [d.page_content for d in r1] == [d.page_content for d in r2]   # second call: cached
    assert len(r1) == 4
    s.table.register_document(2, is_deleted=True)                  # soft delete: documents changed
    r3 = s.similarity_search("q", k=10, filter={"source": "web"})
    assert len(calls) > n1 and len(r3) == 3 and "other" not in [d.page_content for d in r3]
    assert len(s.similarity_search("q", k=10, filter={"source": "web"}, include_deleted=True)) == 4
    s.add_texts(["new web"], metadatas=[{"source": "web"}], document_id=1)                           # rows changed
    assert "new web" in [d.page_content for d in s.similarity_search("q", k=10, filter={"source": "web"})]
    assert len(col.table.where_cache) <= 2


def test_add_texts_batch_equals_per_file_adds_and_rolls_back():
    """The ingestion harness's batched store write: same rows, ids, metadata side effects and ON CONFLICT replacement as
    one add_texts per file; a batch that fails leaves nothing behind."""
    emb = FixedEmbeddings(8, 5)
    files = [([f"a{i}" for i in range(3)], 1), ([f"b{i}" for i in range(2)], 2), (["c0"], None)]
    vec = {n: ko.gen_rows(40 + (n or 0), 5, 0, len(t), 8, True, "f32") for t, n in files}     # distinct rows per file
    s1 = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="one", index_factory=factory)
    for texts, doc in files:
        s1.add_texts(texts, [{"k": j} for j in range(len(texts))], document_id=doc, embeddings=vec[doc])
    s2 = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="two", index_factory=factory)
    metas = [[{"k": j} for j in range(len(t))] for t, _ in files]
    ids = s2.add_texts_batch([(t, m, doc, vec[doc]) for (t, doc), m in zip(files, metas)])
    assert [len(x) for x in ids] == [3, 2, 1] and s2.count() == s1.count() == 6
    assert metas[0][1]["chunk_id"] == ids[0][1] and metas[0][1]["collection"] == "two"
    key = lambda s: sorted((str(r["document_id"]), r["chunk_index"], r["text"], r["metadata"]["k"]) for r in s.table.rows.values())
    assert key(s1) == key(s2)
    q = [float(x) for x in vec[2][1]]
    assert s2.similarity_search_by_vector(q, k=1)[0].page_content == "b1"
    # replacement of document 1's rows, then a failing batch (wrong vector count) that must leave no trace
    s2.add_texts_batch([(["a0 v2"], [{"k": 9}], 1, vec[1][:1])])
    assert s2.count() == 6 and "a0 v2" in [r["text"] for r in s2.table.rows.values()]
    before = (dict(s2.table.rows), dict(s2.table.by_doc_chunk), s2.count())
    with pytest.raises(ValueError):
        s2.add_texts_batch([(["x"], [{}], 7, vec[2][:1]), (["y", "z"], [{}, {}], 8, vec[2][:1])])
    assert (dict(s2.table.rows), dict(s2.table.by_doc_chunk), s2.count()) == before


def test_filtered_search_is_one_snapshot_when_a_writer_moves_the_index_under_it():
    """The reference evaluates WHERE, distance, ORDER BY and LIMIT in one SQL statement = one snapshot
    (postgres_vectorstore.py:296-332). Here the mask is resolved under the table lock and the scan runs outside it; a writer
    that adds rows in between must make the scan REFUSE the mask (layout epoch) and the store rebuild it -- not apply it to
    other rows. The writer is simulated from inside the first search call of the oracle-backed index (no lock is held there,
    exactly the window a concurrent ingestion thread has)."""
    from archi_amd import StaleFilterError
    emb = FixedEmbeddings(16, 9)
    s = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="snap", index_factory=factory)
    s.add_texts([f"web {i}" for i in range(5)], metadatas=[{"source": "web"} for _ in range(5)], document_id=1)
    s.add_texts([f"git {i}" for i in range(5)], metadatas=[{"source": "git"} for _ in range(5)], document_id=2)
    col = s._collection()
    orig = col.index.search
    seen = {"calls": 0, "stale": 0, "locked_retry": False}

    def racing_search(queries, k, **kw):
        seen["calls"] += 1
        if seen["calls"] <= 3 and kw.get("row_filter") is not None:
            # a writer gets in between the mask and the scan, three times in a row
            s.add_texts([f"web late {seen['calls']}"], metadatas=[{"source": "web"}], document_id=3)
        if seen["calls"] == 3:
            seen["locked_retry"] = col.table.lock._is_owned()
        try:
            return orig(queries, k, **kw)
        except StaleFilterError:
            seen["stale"] += 1
            raise
    col.index.search = racing_search
    # two lock-free attempts collide, the third runs under the table lock -- where the simulated writer (same thread: the lock is
    # re-entrant) still gets in, so this call must surface the collision instead of returning rows of a wrong mask
    with pytest.raises(StaleFilterError):
        s.similarity_search("q", k=20, filter={"source": "web"})
    assert seen["stale"] == 3 and seen["locked_retry"]
    # a single collision: the store rebuilds the mask and answers from the new state (all web rows, no git row)
    seen.update(calls=3)
    first = {"done": False}

    def one_collision(queries, k, **kw):
        if not first["done"] and kw.get("row_filter") is not None:
            first["done"] = True
            s.add_texts(["web last"], metadatas=[{"source": "web"}], document_id=4)
        return orig(queries, k, **kw)
    col.index.search = one_collision
    got = [d.page_content for d in s.similarity_search("q", k=20, filter={"source": "web"})]
    # ("web late 1" and "web late 2" were replaced by "web late 3": same document, same chunk index -- ON CONFLICT)
    assert sorted(got) == sorted([f"web {i}" for i in range(5)] + ["web late 3", "web last"])
    # unfiltered searches carry no mask and never collide
    col.index.search = orig
    assert len(s.similarity_search("q", k=50)) == s.count() == 12
