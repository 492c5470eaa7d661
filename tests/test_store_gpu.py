"""GPU suite: the drop-in classes end to end on the real HipIndex -- the golden fixture recorded from the reference's own
PostgresVectorStore / retrievers / hybrid_search (tests/golden/reference_wrapper.json) replayed through
ArchiHipVectorStore with its DEFAULT index factory and DEFAULT dtype (f32 == the reference's vector(D) column), the
pgvector COPY bridge into the GPU index (N2), and the provider constructed through the config plug-in (N4)."""
import io
import json

import numpy as np
import pytest

from archi_amd import vectorstore as vs
from archi_amd.vectorstore import ArchiHipHybridVectorStore, ArchiHipVectorStore
from oracle import knn_oracle as ko
from tests.test_vectorstore_cpu import GOLD, FixedEmbeddings, TableBm25, _load_golden_db, _load_nan_db, _load_retriever_db

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def fresh(hip):
    vs.reset_collections()
    yield
    vs.reset_collections()


def _dump(res):
    return [{"page_content": d.page_content, "metadata": d.metadata, "score": s} for d, s in res]


def test_default_store_is_f32_on_the_real_index():
    s = ArchiHipVectorStore({}, FixedEmbeddings(48, 1), collection_name="dflt")
    s.add_texts(["a", "b"])
    from archi_amd.index import HipIndex
    ix = s._collection().index
    assert isinstance(ix, HipIndex) and ix.dtype == "f32"


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"{c['metric']}-{json.dumps(c['kwargs'])}")
def test_similarity_cases_of_the_reference_fixture(case):
    emb = FixedEmbeddings(case["dim"], case["seed"])
    store = ArchiHipVectorStore({}, emb, collection_name="golden", distance_metric=case["metric"])
    _load_golden_db(store, case["n_rows"], case["dim"], case["seed"])
    assert _dump(store.similarity_search_with_score(case["query_text"], k=case["k"], **case["kwargs"])) == case["results"]


@pytest.mark.parametrize("case", GOLD["hybrid"], ids=lambda c: f"{c['metric']}-{c['semantic_weight']}-{json.dumps(c['kwargs'])}")
def test_hybrid_cases_of_the_reference_fixture(case):
    emb = FixedEmbeddings(case["dim"], case["seed"])
    store = ArchiHipHybridVectorStore({}, emb, collection_name="golden", distance_metric=case["metric"],
                                      bm25=TableBm25(case["bm25_hits"]))
    _load_golden_db(store, case["n_rows"], case["dim"], case["seed"])
    res = store.hybrid_search(case["query_text"], k=case["k"], semantic_weight=case["semantic_weight"],
                              bm25_weight=case["bm25_weight"], **case["kwargs"])
    assert _dump(res) == case["results"]


def test_hybrid_nan_first_on_the_real_index():
    case = GOLD["hybrid_nan"]
    emb = FixedEmbeddings(case["dim"], case["seed"])
    store = ArchiHipHybridVectorStore({}, emb, collection_name="golden", bm25=TableBm25(case["bm25_hits"]))
    _load_nan_db(store, case)
    res = store.hybrid_search(case["query_text"], k=case["k"], semantic_weight=case["semantic_weight"], bm25_weight=case["bm25_weight"])
    assert [{"page_content": d.page_content, "metadata": d.metadata, "score": None if s != s else s} for d, s in res] == case["results"]


def test_retriever_calls_of_the_reference_fixture():
    R = GOLD["retrievers"]
    emb = FixedEmbeddings(48, 2024)
    q = R["query_text"]
    store = ArchiHipVectorStore({}, emb, collection_name="golden")
    _load_retriever_db(store, 300, 48, 2024)
    assert _dump(store.similarity_search_with_score(q, k=3)) == R["semantic_k3"]
    assert [{"page_content": d.page_content, "metadata": d.metadata} for d in store.similarity_search(q, k=3)] == R["grading_k3"]
    assert _dump(store.similarity_search_with_score(q, k=5)) == R["hybrid_fallback_k5"]
    vs.reset_collections()
    hstore = ArchiHipHybridVectorStore({}, emb, collection_name="golden", bm25=TableBm25(R["bm25_hits"]))
    _load_retriever_db(hstore, 300, 48, 2024)
    assert _dump(hstore.hybrid_search(query=q, k=5, semantic_weight=0.5, bm25_weight=0.5)) == R["hybrid_native_k5"]


def test_reingest_cycle_never_hits_a_capacity(hip):
    """update_vectorstore's delete + re-add of changed files (manager.py:192-211) on a store with a tiny first reservation."""
    emb = FixedEmbeddings(32, 7)
    s = ArchiHipVectorStore({"hip": {"capacity": 64}}, emb, collection_name="cycle")
    for rnd in range(30):
        for doc in range(8):
            s.add_texts([f"doc{doc} chunk{i} v{rnd}" for i in range(10)], document_id=doc)       # ON CONFLICT replaces
    assert s.count() == 80
    ix = s._collection().index
    assert ix.slots <= ix.allocated_rows <= 1024
    texts = {d.page_content for d in s.similarity_search("q", k=80)}
    assert texts == {f"doc{doc} chunk{i} v29" for doc in range(8) for i in range(10)}


@pytest.mark.parametrize("id_bytes", [4, 8])
def test_pgcopy_stream_into_the_gpu_index(hip, id_bytes):
    """N2: COPY (SELECT id, embedding FROM document_chunks) TO STDOUT (FORMAT binary) -> load_index_from_pgcopy -> HipIndex
    -> search == oracle on the decoded float32 rows (NULL embeddings skipped, int4 / int8 ids = document_chunks.id)."""
    from archi_amd import pgbridge as pb
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(31)
    n, d = 20000, 384
    vec = rng.standard_normal((n, d)).astype(np.float32)
    null_rows = rng.choice(n, 37, replace=False)
    vec_w = vec.copy()
    vec_w[null_rows] = np.nan
    ids = rng.permutation((10 ** 11 if id_bytes == 8 else 0) + np.arange(n) * 3 + 1).astype(np.int64)
    buf = io.BytesIO()
    pb.write_pgcopy_vectors(buf, ids, vec_w, id_bytes=id_bytes)
    ix = HipIndex(d, 1024, dtype="f32", metric="cosine", device=0)              # grows while loading
    loaded = pb.load_index_from_pgcopy(ix, io.BytesIO(buf.getvalue()), batch=4096)
    keep = np.ones(n, bool); keep[null_rows] = False
    assert loaded == n - 37 == ix.count()
    q = rng.standard_normal((9, d)).astype(np.float32)
    for mode in ("auto", "exact"):
        gi, gd, _ = ix.search(q, 10, mode=mode)
        wi, wd, _ = ko.search(vec[keep], q, 10, "cosine", ids=ids[keep])
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    out = io.BytesIO()                                                           # and back out: the same tuples
    live_ids = ids[keep]
    pb.dump_index_to_pgcopy(ix, ix.lookup(live_ids), live_ids, out)
    ri, rv = pb.read_pgcopy_vectors(io.BytesIO(out.getvalue()))
    assert np.array_equal(ri, live_ids) and np.array_equal(rv, vec[keep])
    ix.close()


def test_provider_built_through_the_config_plugin(hip):
    """N4: the YAML's embedding_class_map entry -> config_plugin.resolve_embedding_classes -> class(**kwargs), as
    VectorStoreManager.__init__ does (manager.py:66-73) -> embed on the GPU."""
    from archi_amd import config_plugin as cp
    cmap = {"ArchiHipEmbeddings": {"class": "ArchiHipEmbeddings", "dimensions": 384,
                                   "kwargs": {"model_name": "sentence-transformers/all-MiniLM-L6-v2",
                                              "model_kwargs": {"device": "cuda", "synthetic_seed": 3},
                                              "encode_kwargs": {"normalize_embeddings": True}}}}
    entry = cp.resolve_embedding_classes(cmap)["ArchiHipEmbeddings"]
    model = entry["class"](**entry["kwargs"])
    tok = np.random.default_rng(0).integers(1000, 30000, size=(3, 16)).astype(np.int32)
    out = np.asarray(model.embed_token_arrays(tok, np.full(3, 16, np.int32)))
    assert out.shape == (3, cp.init_sql_dimensions({"embedding_name": "ArchiHipEmbeddings", "embedding_class_map": cmap}))
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
    model.encoder.close()


def test_dump_to_pgcopy_and_reload_on_the_gpu_index(hip):
    """Durability round trip with the real index: dump (stored float32 rows fetched from HBM) -> reset -> load -> same answers."""
    import io
    from archi_amd import vectorstore as vs2
    from archi_amd.vectorstore import ArchiHipVectorStore
    from oracle import knn_oracle as ko2

    class E:
        def embed_documents(self, texts):
            raise AssertionError

        def embed_query(self, text):
            raise AssertionError
    vs2.reset_collections()
    n, dim = 6000, 128
    vec = ko2.gen_rows(91, 0, 0, n, dim, True, "f32")
    st = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, E(), collection_name="durgpu")
    for doc in range(60):
        st.add_texts([f"d{doc} c{i}" for i in range(100)], [{"page": i % 5, "resource_hash": f"h{doc}"} for i in range(100)],
                     document_id=doc + 1, embeddings=vec[doc * 100:(doc + 1) * 100])
    st.delete(document_id=7)
    st.table.register_document(9, is_deleted=True, display_name="nine")
    q = [float(x) for x in vec[4321]]
    want = [(d.page_content, d.metadata, s) for d, s in st.similarity_search_by_vector_with_score(q, k=20, filter={"page": 1})]
    chunks, docs = io.BytesIO(), io.BytesIO()
    assert st.dump_to_pgcopy(chunks, docs) == n - 100
    vs2.reset_collections()
    st2 = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, E(), collection_name="durgpu")
    assert st2.load_from_pgcopy(io.BytesIO(chunks.getvalue()), io.BytesIO(docs.getvalue())) == n - 100
    got = [(d.page_content, d.metadata, s) for d, s in st2.similarity_search_by_vector_with_score(q, k=20, filter={"page": 1})]
    assert got == want and len(got) == 20
    vs2.reset_collections()


def test_filtered_readers_see_one_snapshot_while_a_writer_moves_the_index():
    """One ingestion writer (add / re-ingest = delete + add / delete / soft-delete a document / compact the index) against 8
    request threads running similarity_search_by_vector_with_score(filter=...) on the REAL index: the reference evaluates
    WHERE, distance, ORDER BY and LIMIT in one statement (postgres_vectorstore.py:296-332), so a reader may see the state
    before or after a write, never a mask applied to the wrong rows. Checked while it runs: no exception; every returned
    chunk satisfies the filter; no chunk of a document that has been soft-deleted before the search started ever comes back;
    scores descend. And afterwards: the store equals the oracle on the final state."""
    import threading
    import time
    from archi_amd import index as hip_index
    rng = np.random.default_rng(5)
    d = 64
    emb = FixedEmbeddings(d, 3)
    s = ArchiHipVectorStore({"hip": {"dtype": "f32", "capacity": 512}}, emb, collection_name="race")      # small: grows and reclaims often

    def unit(n):
        x = rng.standard_normal((n, d)).astype(np.float32)
        return x / np.linalg.norm(x, axis=1, keepdims=True)

    state = {}                      # document id -> (kind, vectors) of the live chunks, writer-side truth
    soft_deleted = set()            # documents soft-deleted so far (monotone: never revived)
    lock = threading.Lock()

    def ingest(doc, kind, n):
        vec = unit(n)
        s.add_texts([f"{kind} {doc} {i}" for i in range(n)], metadatas=[{"source": kind, "doc": doc} for _ in range(n)],
                    document_id=doc, embeddings=vec)
        state[doc] = (kind, vec)

    for doc in range(40):
        ingest(doc, "web" if doc % 2 else "git", 60)
    queries = unit(8)
    stop = threading.Event()
    errors = []
    searches = [0]

    def reader(j):
        try:
            q = [float(x) for x in queries[j]]
            while not stop.is_set():
                with lock:
                    gone_before = set(soft_deleted)
                res = s.similarity_search_by_vector_with_score(q, k=10, filter={"source": "web"})
                scores = [sc for _, sc in res]
                assert scores == sorted(scores, reverse=True), "scores not descending"
                for doc_, _ in res:
                    assert doc_.metadata["source"] == "web", f"filtered-out chunk returned: {doc_.page_content!r}"
                    assert doc_.metadata["doc"] not in gone_before, f"chunk of soft-deleted document {doc_.metadata['doc']} returned"
                assert len(res) == 10
                searches[0] += 1
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=reader, args=(j,)) for j in range(8)]
    for t in threads:
        t.start()
    nxt = 40
    try:
        for cycle in range(60):
            ingest(nxt, "web" if cycle % 3 else "git", 50); nxt += 1                  # plain add (the slot count grows)
            victim = int(rng.choice(sorted(state)))
            kind, _ = state[victim]
            s.delete(document_id=victim)                                               # re-ingest as update_vectorstore does it:
            ingest(victim, kind, 55)                                                   # delete the file's rows, add them again (manager.py:192-211)
            if cycle % 4 == 1:
                gone = int(rng.choice(sorted(state)))
                s.delete(document_id=gone); del state[gone]                            # DELETE
            if cycle % 5 == 2:
                web_docs = [x for x in sorted(state) if state[x][0] == "web" and x not in soft_deleted]
                if len(web_docs) > 6:
                    sd = int(rng.choice(web_docs))
                    s.table.register_document(sd, is_deleted=True)                     # soft delete (documents.is_deleted)
                    with lock:
                        soft_deleted.add(sd)                                           # only now may readers insist on it
            if cycle % 7 == 3:
                with s.table.lock:                                                     # VACUUM: every writer of the index holds the table lock
                    s._collection().index.compact()
            time.sleep(0.002)
    finally:
        stop.set()
        for t in threads:
            t.join()
    assert not errors, errors[:3]
    assert searches[0] > 50
    # final state == oracle: live web chunks of documents that are not soft-deleted
    keep = [(doc, i) for doc in sorted(state) if state[doc][0] == "web" and doc not in soft_deleted for i in range(len(state[doc][1]))]
    rows = np.stack([state[doc][1][i] for doc, i in keep])
    q = queries[:4]
    wi, wd, _ = ko.search(rows, q, 10, "cosine")
    for j in range(4):
        res = s.similarity_search_by_vector_with_score([float(x) for x in q[j]], k=10, filter={"source": "web"})
        want = [f"web {keep[int(i)][0]} {keep[int(i)][1]}" for i in wi[j]]
        assert [doc_.page_content for doc_, _ in res] == want
        assert [sc for _, sc in res] == [1.0 - float(x) for x in wd[j]]


# ---- refresh_from_pgcopy (round-5 review, missing #1): a chat process's GPU collection reconciled with the table the
# data-manager process writes (src/archi/archi.py:61-65, src/data_manager/vectorstore/manager.py:177-214) ---------------------
class _NoEmbed:
    def embed_documents(self, texts):
        raise AssertionError("vectors are handed in")

    def embed_query(self, text):
        x = np.random.default_rng(len(text)).standard_normal(96).astype(np.float32)
        return [float(v) for v in x / np.linalg.norm(x)]


def _mk_shared(cls=ArchiHipHybridVectorStore, **hipcfg):
    kw = {"bm25": vs.HostBm25()} if cls is ArchiHipHybridVectorStore else {}
    return cls({"hip": dict({"dtype": "f32"}, **hipcfg)}, _NoEmbed(), collection_name="shared", **kw)


def test_refresh_from_pgcopy_equals_a_store_loaded_from_scratch_on_the_real_index():
    """Two store processes' worth of state: a writer dump taken before and after add / delete / re-ingest / soft delete / rename /
    an in-place rewrite; the reader collection after refresh_from_pgcopy returns the same similarity_search_with_score and
    hybrid_search results as a store loaded from scratch from the second dump. Then the same listing again: nothing moves."""
    from tests.refresh_scenario import Proc, Table, answers, ingest, unit, writer_moves
    rng = np.random.default_rng(177)
    d = 96
    wp, rp, sp = Proc(), Proc(), Proc()

    def rows_of(ids):
        with wp:
            return table.rows_stream(np.asarray(ids).tolist())
    with wp:
        w = _mk_shared()
        for doc in range(1, 41):
            ingest(w, rng, doc, 100 + doc % 9, d)                    # ~4 100 chunks: the MFMA scan path, not the tiny-index one
        table = Table(w)
        table.commit()
        s0 = (table.rows_stream(), table.documents_stream(), table.ids_stream())
    with rp:
        r = _mk_shared()
        assert r.load_from_pgcopy(s0[0], s0[1], versions_stream=s0[2]) == r.count()
    with wp:
        victim, newvec = writer_moves(w, table, rng, d, 41)
        queries = unit(rng, 6, d)
        queries[0] = newvec
        ids_s, docs_s = table.ids_stream(), table.documents_stream()
        final = (table.rows_stream(), table.documents_stream())
    with rp:
        stats = r.refresh_from_pgcopy(ids_s, rows_of, docs_s)
        assert stats["added"] == 5 * 40 + 35 and stats["updated"] == 1 and stats["removed"] > 0, stats
        got = answers(r, queries, hybrid=True)
    with sp:
        scratch = _mk_shared()
        scratch.load_from_pgcopy(*final)
        want = answers(scratch, queries, hybrid=True)
    assert got == want
    assert any("rewritten in place" in row[0] for row in got[0]) and got[0][0][2] == pytest.approx(1.0, abs=1e-6)
    with wp:
        ids_s, docs_s = table.ids_stream(), table.documents_stream()
    with rp:
        t = r.table
        before = (r._collection().index.layout(), t.version, t.doc_version, t.text_epoch, set(t.where_cache))
        assert r.refresh_from_pgcopy(ids_s, None, docs_s) == {"removed": 0, "added": 0, "updated": 0, "documents_changed": 0, "fetched": 0}
        assert before == (r._collection().index.layout(), t.version, t.doc_version, t.text_epoch, set(t.where_cache))
        assert answers(r, queries, hybrid=True) == want
    for p in (wp, rp, sp):
        p.close()


def test_filtered_readers_never_see_a_dead_row_while_the_collection_is_refreshed():
    """The reader process under load, on the real index (scenario and checks: tests/refresh_scenario.py)."""
    from tests.refresh_scenario import concurrent_refresh_scenario
    searches = concurrent_refresh_scenario(lambda metric, **hipcfg: ArchiHipVectorStore(
        {"hip": dict({"dtype": "f32"}, **hipcfg)}, _NoEmbed(), collection_name="shared", distance_metric=metric))
    assert searches > 50
