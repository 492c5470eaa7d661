"""CPU suite: the columnar host table (archi_amd/chunktable.py), the sync step of the data manager
(/root/reference/src/data_manager/vectorstore/manager.py:177-252) on top of it, and the whole-row COPY loader (N2)."""
import io
import struct
import time

import numpy as np
import pytest

from archi_amd import pgbridge
from archi_amd import vectorstore as vs
from archi_amd.chunktable import ChunkTable
from archi_amd.vectorstore import ArchiHipVectorStore
from oracle import knn_oracle as ko
from tests.fake_index import OracleIndex


def factory(dim, capacity, dtype, metric):
    return OracleIndex(dim, capacity, dtype=dtype, metric=metric)


@pytest.fixture(autouse=True)
def fresh():
    vs.reset_collections()
    yield
    vs.reset_collections()


def test_table_semantics_upsert_delete_filter_vacuum():
    t = ChunkTable()
    for doc in range(1, 5):
        for i in range(3):
            t.append(t.next_id, doc, i, f"d{doc} c{i} ü", {"chunk_id": f"id-{doc}-{i}", "page": i, "resource_hash": f"h{doc}",
                                                             "nested": {"a": [1, 2]}})
    assert len(t) == 12 and t.next_id == 13 and t.find(2, 1) == 5 and t.find(9, 0) is None
    assert t.rows[5] == {"document_id": 2, "chunk_index": 1, "text": "d2 c1 ü",
                         "metadata": {"chunk_id": "id-2-1", "page": 1, "resource_hash": "h2", "nested": {"a": [1, 2]}}}
    assert t.rids_of_document(3) == [7, 8, 9] and t.rids_of_chunk_ids(["id-1-0", "id-4-2", "nope"]) == [1, 12]
    # metadata->>'key' = str(value): numbers compare as their JSON text, nested values as theirs (postgres_vectorstore.py:300-302)
    assert t.rids_at(t.positions_matching({"page": 1})).tolist() == [2, 5, 8, 11]
    assert t.rids_at(t.positions_matching({"page": "1", "resource_hash": "h3"})).tolist() == [8]
    assert t.rids_at(t.positions_matching({"nested": '{"a": [1, 2]}'})).tolist() == list(range(1, 13))
    assert len(t.positions_matching({"missing": "x"})) == 0
    assert t.distinct_values("resource_hash") == {"h1", "h2", "h3", "h4"}
    # ON CONFLICT replacement as the store does it: new row first, the old one dies afterwards
    t.append(t.next_id, 2, 1, "d2 c1 v2", {"chunk_id": "new", "page": 1, "resource_hash": "h2"})
    assert t.find(2, 1) == 13
    t.kill(5)
    assert t.find(2, 1) == 13 and 5 not in t.rows and t.rids_at(t.positions_matching({"page": 1})).tolist() == [2, 8, 11, 13]
    for rid in t.rids_of_document(3):
        t.kill(rid)
    assert t.distinct_values("resource_hash") == {"h1", "h2", "h4"} and len(t) == 9
    before = {rid: t.rows[rid] for rid in t.rows}
    t.vacuum()
    assert {rid: t.rows[rid] for rid in t.rows} == before and t.positions == 9 and t.find(2, 1) == 13
    assert t.rids_at(t.positions_matching({"page": 1})).tolist() == [2, 11, 13]
    t.update_row(13, metadata={"chunk_id": "new", "page": 2, "resource_hash": "hX"})
    assert t.rids_at(t.positions_matching({"page": 1})).tolist() == [2, 11] and "hX" in t.distinct_values("resource_hash")


def test_append_block_leaves_what_row_appends_leave():
    """The ingestion path inserts a file's chunks as one block (columns written as slices): same table as n append() calls --
    rows, (document, chunk) lookups, inverted maps, chunk-id lookups, next id -- including a document_id of None, non-dict
    metadata rows, an empty block, and the fallback when ids were handed in out of order."""
    def fill(t, block):
        docs = [(7, 3), ("doc-b", 1), (None, 2), (7, 2), (9, 0)]
        for doc, n in docs:
            texts = [f"{doc} chunk {i} \u00fc\ud83d\ude00" for i in range(n)]
            mds = [{"chunk_id": f"id-{doc}-{i}-{t.next_id}", "resource_hash": f"h{doc}", "page": i, "deep": {"k": [i]}} for i in range(n)]
            if block:
                first = t.append_block(doc, texts, mds)
                assert first == t.next_id - n
            else:
                for i in range(n):
                    t.append(t.next_id, doc, i, texts[i], mds[i])
        return t
    a, b = fill(ChunkTable(), True), fill(ChunkTable(), False)
    assert len(a) == len(b) == 8 and a.next_id == b.next_id == 9
    assert dict(a.rows) == dict(b.rows)
    for doc in (7, "doc-b", None, 9, "nope"):
        assert a.rids_of_document(doc) == b.rids_of_document(doc)
        assert a.has_document(doc) == (doc in (7, "doc-b"))
    assert a.find(7, 1) == b.find(7, 1) == 8 and a.find(7, 2) == 3        # the later block of document 7 wins where it overlaps
    for flt in ({"resource_hash": "h7"}, {"page": 1}, {"resource_hash": "hNone", "page": 0}):
        assert a.rids_at(a.positions_matching(flt)).tolist() == b.rids_at(b.positions_matching(flt)).tolist()
    cids = [a.rows[r]["metadata"]["chunk_id"] for r in (1, 4, 8)]
    assert a.rids_of_chunk_ids(cids) == b.rids_of_chunk_ids(cids) == [1, 4, 8]
    assert a.distinct_values("resource_hash") == b.distinct_values("resource_hash")
    # a block that cannot be stored (metadata JSON cannot express, a text that is not a str) leaves the table untouched
    before = (len(a), a.next_id, dict(a.rows), a.rids_of_document(7), a.positions)
    with pytest.raises(TypeError):
        a.append_block(7, ["ok", "bad"], [{"chunk_id": "x"}, {"chunk_id": "y", "obj": object()}])
    with pytest.raises(AttributeError):
        a.append_block(7, ["ok", 5], [{}, {}])
    with pytest.raises(TypeError):
        a.append(a.next_id, 7, 0, "t", {"obj": {1, 2}})
    assert (len(a), a.next_id, dict(a.rows), a.rids_of_document(7), a.positions) == before
    assert a.append_block(11, ["after"], [{}]) == before[1] and a.rids_of_document(11) == [before[1]]
    # ids out of order (a row id below the last one): append_block takes the row-by-row path and stays consistent
    c = ChunkTable()
    c.append(10, 1, 0, "x", {"chunk_id": "a"})
    c.append(4, 1, 1, "y", {"chunk_id": "b"})
    first = c.append_block(2, ["p", "q"], [{"chunk_id": "c"}, {"chunk_id": "d"}])
    assert first == 11 and c.rids_of_document(2) == [11, 12] and c.rows[12]["text"] == "q" and c.pos(4) >= 0
    assert c.rids_of_chunk_ids(["b", "d"]) == [4, 12]


def test_table_at_a_million_rows_nothing_is_a_python_pass_over_the_rows():
    """delete(document_id) and a WHERE mask on a 1M-chunk collection used to walk a dict of dicts."""
    t = ChunkTable()
    n_docs, per = 50_000, 20
    t0 = time.perf_counter()
    for doc in range(n_docs):
        md = {"resource_hash": f"h{doc}", "source": "web" if doc % 4 else "git", "collection": "c"}
        for i in range(per):
            t.append(t.next_id, doc, i, "x" * 40, md)
    build_s = time.perf_counter() - t0
    assert len(t) == n_docs * per
    t0 = time.perf_counter()
    rids = t.rids_of_document(31337)
    for rid in rids:
        t.kill(rid)
    dt_delete = time.perf_counter() - t0
    assert len(rids) == per and dt_delete < 0.05
    t0 = time.perf_counter()
    assert t.find(777, 3) == 777 * per + 4 and t.pos(123456) == 123455 and t.row(999_999)["chunk_index"] == 18
    assert time.perf_counter() - t0 < 0.01
    t0 = time.perf_counter()
    pos = t.positions_matching({"resource_hash": "h4242"})
    assert len(pos) == per and time.perf_counter() - t0 < 0.01
    t.positions_matching({"source": "git"})                       # first use of a key: one pass builds its inverted map
    t0 = time.perf_counter()
    assert len(t.positions_matching({"source": "git"})) == n_docs // 4 * per
    assert time.perf_counter() - t0 < 0.5                         # afterwards: list -> array, no row is parsed
    t0 = time.perf_counter()
    assert len(t.distinct_values("resource_hash")) == n_docs - 1
    assert time.perf_counter() - t0 < 2.0
    assert build_s < 120


class HashEmb:
    def embed_documents(self, texts):
        return [[float(x) for x in ko.gen_rows(abs(hash(t)) % 100000, 5, 0, 1, 16, True, "f32")[0]] for t in texts]

    def embed_query(self, text):
        return self.embed_documents([text])[0]


def test_sync_replays_a_catalog_diff_like_update_vectorstore():
    """manager.py:177-214: hashes in the store vs hashes in the catalog -> remove the stale ones, add the missing ones; a
    restarted data manager finds what is already embedded and leaves it alone."""
    from archi_amd.ingest import BatchedIngestor
    store = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, HashEmb(), collection_name="sync", index_factory=factory)
    ing = BatchedIngestor(store, collection="sync")
    catalog = {f"hash{i}": (f"file{i}.txt", f"body of file {i}\n\nsecond paragraph {i} " + "w " * 300) for i in range(6)}

    def add(missing):
        ing.ingest([(h, catalog_now[h][0], catalog_now[h][1]) for h in missing])

    catalog_now = dict(catalog)
    r = store.sync(catalog_now, add)
    assert sorted(r["added"]) == sorted(catalog) and r["removed"] == [] and store.resource_hashes() == set(catalog)
    n0 = store.count()
    assert n0 >= 6
    r = store.sync(catalog_now, add)                              # restart: nothing to do
    assert r["added"] == [] and r["removed"] == [] and store.count() == n0
    del catalog_now["hash1"], catalog_now["hash4"]                # two files left the catalog, one changed (new hash)
    catalog_now["hash2b"] = ("file2.txt", "rewritten file 2")
    del catalog_now["hash2"]
    r = store.sync(catalog_now, add)
    assert r["removed"] == ["hash1", "hash2", "hash4"] and r["added"] == ["hash2b"]
    assert store.resource_hashes() == set(catalog_now)
    texts = [d.page_content for d in store.similarity_search("body", k=50)]
    assert "rewritten file 2" in texts and not any("file 1" in x or "file 4" in x for x in texts)

    def failing(missing):
        raise RuntimeError("embedder down")
    catalog_now["hash9"] = ("file9.txt", "new")
    r = store.sync(catalog_now, failing)                          # manager.py:208-211: logged, the run carries on
    assert r["error"] == "embedder down" and r["added"] == [] and "hash9" not in store.resource_hashes()


# (id, document_id, chunk_index, chunk_text, metadata, embedding) tuple as PostgreSQL's binary COPY writes it, by hand:
# int4 7 | int4 3 | int4 0 | text 'hé' | jsonb (version 1) '{"a": 1}' | vector dim 2 [1.0, -2.5]
CHUNK_TUPLE = (b"\x00\x06" + b"\x00\x00\x00\x04\x00\x00\x00\x07" + b"\x00\x00\x00\x04\x00\x00\x00\x03" +
               b"\x00\x00\x00\x04\x00\x00\x00\x00" + b"\x00\x00\x00\x03h\xc3\xa9" +
               b"\x00\x00\x00\x09\x01{\"a\": 1}" + b"\x00\x00\x00\x0c\x00\x02\x00\x00\x3f\x80\x00\x00\xc0\x20\x00\x00")
NULL_TUPLE = (b"\x00\x06" + b"\x00\x00\x00\x04\x00\x00\x00\x08" + b"\xff\xff\xff\xff" + b"\x00\x00\x00\x04\x00\x00\x00\x01" +
              b"\x00\x00\x00\x00" + b"\xff\xff\xff\xff" + b"\x00\x00\x00\x0c\x00\x02\x00\x00\x00\x00\x00\x00\x3f\x00\x00\x00")


def test_whole_row_copy_stream_known_answer_and_round_trip():
    stream = pgbridge.SIGNATURE + struct.pack(">ii", 0, 0) + CHUNK_TUPLE + NULL_TUPLE + struct.pack(">h", -1)
    (blk,) = list(pgbridge.iter_pgcopy_chunks(io.BytesIO(stream)))
    assert blk["ids"].tolist() == [7, 8] and blk["document_ids"] == [3, None] and blk["chunk_index"].tolist() == [0, 1]
    assert [b.decode("utf-8") for b in blk["text_bytes"]] == ["hé", ""] and blk["metadata"] == [{"a": 1}, None]
    # the stream is parsed through a refillable buffer: a buffer smaller than any field gives the same tuples
    whole = [[None if v is None else bytes(v) for v in row] for row in pgbridge._tuples(io.BytesIO(stream), 6)]
    for chunk in (1, 3, 7, 64):
        assert [[None if v is None else bytes(v) for v in row] for row in pgbridge._tuples(io.BytesIO(stream), 6, chunk=chunk)] == whole
    with pytest.raises(ValueError, match="truncated"):
        list(pgbridge._tuples(io.BytesIO(stream[:-5]), 6, chunk=16))
    assert blk["vectors"].tolist() == [[1.0, -2.5], [0.0, 0.5]]
    out = io.BytesIO()
    pgbridge.write_pgcopy_chunks(out, [(7, 3, 0, "hé", {"a": 1}, np.array([1.0, -2.5], np.float32)),
                                       (8, None, 1, "", None, np.array([0.0, 0.5], np.float32))])
    assert out.getvalue() == stream                              # the writer emits the same bytes
    with pytest.raises(ValueError):
        list(pgbridge.iter_pgcopy_chunks(io.BytesIO(stream[:-9])))
    docs = io.BytesIO()
    pgbridge.write_pgcopy_documents(docs, [{"id": 3, "resource_hash": "h3", "display_name": "Doc 3", "source_type": "web",
                                            "url": None, "is_deleted": True}])
    assert pgbridge.read_pgcopy_documents(io.BytesIO(docs.getvalue())) == [
        {"id": 3, "resource_hash": "h3", "display_name": "Doc 3", "source_type": "web", "url": None, "is_deleted": True}]


def test_store_loaded_from_a_dump_returns_documents_not_bare_ids():
    """VERDICT r2 missing #4: an index loaded through N2 used to answer similarity_search with nothing."""
    n, dim = 500, 24
    vec = ko.gen_rows(31, 0, 0, n, dim, True, "f32")
    rows = []
    for i in range(n):
        md = {"collection": "dump" if i % 7 else "other", "filename": f"f{i // 5}.txt", "chunk_id": f"c{i}", "resource_hash": f"h{i // 5}"}
        if i % 50 == 0:
            md = None
        rows.append((1000 + i, 1 + i // 5, i % 5, f"chunk text {i}", md, vec[i] if i % 31 else None))
    chunks, docs = io.BytesIO(), io.BytesIO()
    pgbridge.write_pgcopy_chunks(chunks, rows)
    pgbridge.write_pgcopy_documents(docs, [{"id": d, "resource_hash": f"h{d - 1}", "display_name": f"Doc {d}", "source_type": "web",
                                            "url": f"https://x/{d}" if d % 2 else None, "is_deleted": d == 3} for d in range(1, 101)])
    store = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, HashEmb(), collection_name="dump", index_factory=factory)
    loaded = store.load_from_pgcopy(io.BytesIO(chunks.getvalue()), io.BytesIO(docs.getvalue()), batch=128)
    keep = [r for r in rows if r[5] is not None and (r[4] or {}).get("collection") in (None, "dump")]
    assert loaded == len(keep) == store.count()
    q = [float(x) for x in vec[17]]
    (doc, score), = store.similarity_search_by_vector_with_score(q, k=1)
    assert doc.page_content == "chunk text 17" and score == 1.0 - ko.distance("cosine", vec[17], vec[17])
    assert doc.metadata["display_name"] == "Doc 4" and doc.metadata["filename"] == "f3.txt"
    got = store.similarity_search_by_vector_with_score(q, k=40)
    assert len(got) == 40 and not any(d.metadata.get("display_name") == "Doc 3" for d, _ in got)   # soft-deleted document
    # the loaded table keeps working like one this backend filled: upsert on (document_id, chunk_index), delete, new ids above
    store.add_texts(["replacement"], [{"filename": "f3.txt"}], document_id=4, embeddings=vec[15:16])     # (4, 0) = row 1015
    q15 = [float(x) for x in vec[15]]
    assert store.count() == loaded and store.similarity_search_by_vector(q15, k=1)[0].page_content == "replacement"
    assert store.table.next_id > 1000 + n
    store.delete(document_id=5)
    assert store.count() == loaded - sum(1 for r in keep if r[1] == 5)
    with pytest.raises(ValueError, match="already in collection"):
        store.load_from_pgcopy(io.BytesIO(chunks.getvalue()))


def test_bulk_load_rows_in_any_order_into_any_table():
    """The COPY loader writes a block's columns as slices when its ids extend the table in order; a stream in another order, ids
    BELOW rows the table already holds (row-by-row fallback), non-ASCII text, metadata with escapes and nested values, NULL
    metadata: the same table as inserting the rows one by one. An id twice in one stream is an error."""
    from archi_amd.chunktable import ChunkTable
    dim = 8
    vec = ko.gen_rows(5, 0, 0, 300, dim, True, "f32")
    def row(i, rid):
        md = None if i % 9 == 0 else {"collection": "c", "resource_hash": f"h{i % 11}", "chunk_id": f"id{rid}", "quote": 'a"b\\c', "deep": {"k": [i, None, 1.5]},
                                      "uni": "\u00fc\u4e2d"}
        return (rid, None if i % 13 == 0 else 1 + i % 17, i % 5, f"text {i} \u00e4\u00f6 \U0001f600", md, vec[i])
    first = [row(i, 500 + i) for i in range(100)]                       # ascending: the slice path
    second = [row(100 + i, 900 - i) for i in range(100)]                # descending stream, above the table: sorted, slice path
    third = [row(200 + i, 100 + i) for i in range(100)]                 # below everything already there: the fallback
    store = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, HashEmb(), collection_name="c", index_factory=factory)
    want = ChunkTable()
    for part in (first, second, third):
        buf = io.BytesIO()
        pgbridge.write_pgcopy_chunks(buf, part)
        buf.seek(0)
        assert store.load_from_pgcopy(buf, batch=37) == 100
    for rid, doc, cidx, text, md, _ in sorted(first, key=lambda r: r[0]) + sorted(second, key=lambda r: r[0]) + third:
        want.append(rid, doc, cidx, text, md if md is not None else {})
    t = store.table
    assert len(t) == 300 and dict(t.rows) == dict(want.rows) and t.next_id == want.next_id == 901
    for doc in (None, 1, 5, 17):
        assert sorted(t.rids_of_document(doc)) == sorted(want.rids_of_document(doc))
    assert sorted(t.rids_at(t.positions_matching({"resource_hash": "h3"})).tolist()) == sorted(want.rids_at(want.positions_matching({"resource_hash": "h3"})).tolist())
    assert t.rids_of_chunk_ids(["id505", "id850", "id150"]) == want.rids_of_chunk_ids(["id505", "id850", "id150"]) and \
        sorted(t.rids_of_chunk_ids(["id505", "id850", "id150"])) == [150, 505, 850]
    q = [float(x) for x in vec[250]]
    assert store.similarity_search_by_vector(q, k=1)[0].page_content == row(250, 150)[3]
    twice = io.BytesIO()
    pgbridge.write_pgcopy_chunks(twice, [row(1, 2000), row(2, 2001), row(3, 2000)])
    twice.seek(0)
    with pytest.raises(ValueError, match="appears twice"):
        store.load_from_pgcopy(twice)
    assert len(store.table) == 300
    # a block whose vectors do not go into the index leaves no rows behind (table rows without vectors made
    # dump_to_pgcopy raise and count() disagree with len(table)); the same stream then loads on a retry
    late = io.BytesIO()
    pgbridge.write_pgcopy_chunks(late, [row(210 + i, 3000 + i) for i in range(20)])
    ix = store._collection().index
    orig_add, fail = ix.add, {"n": 1}

    def flaky_add(*a, **k):
        if fail["n"]:
            fail["n"] -= 1
            raise RuntimeError("out of memory (simulated)")
        return orig_add(*a, **k)
    ix.add = flaky_add
    with pytest.raises(RuntimeError, match="simulated"):
        store.load_from_pgcopy(io.BytesIO(late.getvalue()))
    assert len(store.table) == 300 == store.count() and store.table.pos(3005) < 0
    out = io.BytesIO()
    assert store.dump_to_pgcopy(out) == 300                              # every table row still has its vector
    assert store.load_from_pgcopy(io.BytesIO(late.getvalue())) == 20 and len(store.table) == 320 == store.count()
    # a stream of another width is refused before anything is touched
    wide = io.BytesIO()
    pgbridge.write_pgcopy_chunks(wide, [(5000, 1, 0, "wide", {"collection": "c"}, np.ones(dim + 8, np.float32))])
    wide.seek(0)
    with pytest.raises(ValueError, match="-d"):
        store.load_from_pgcopy(wide)
    assert len(store.table) == 320 == store.count()


def test_dump_and_reload_survives_a_restart():
    """VERDICT r2 missing #4 (durability of the non-vector columns): dump -> (process restart) -> load gives back the same
    collection: texts, metadata, document columns, soft deletes, ids, and the same search answers."""
    emb = HashEmb()
    store = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="dur", index_factory=factory)
    vec = ko.gen_rows(5, 0, 0, 60, 16, True, "f32")
    for doc in range(1, 7):
        store.add_texts([f"doc {doc} chunk {i} é" for i in range(10)], [{"page": i, "resource_hash": f"h{doc}"} for i in range(10)],
                        document_id=doc, embeddings=vec[(doc - 1) * 10: doc * 10])
        store.table.register_document(doc, resource_hash=f"h{doc}", display_name=f"Doc {doc}", source_type="web",
                                      url=None if doc % 2 else f"https://x/{doc}", is_deleted=(doc == 4))
    store.delete(document_id=2)
    q = [float(x) for x in vec[33]]
    before = [(d.page_content, d.metadata, s) for d, s in store.similarity_search_by_vector_with_score(q, k=12, filter={"page": 3})]
    before_all = [(d.page_content, d.metadata, s) for d, s in store.similarity_search_by_vector_with_score(q, k=50, include_deleted=True)]
    chunks, docs = io.BytesIO(), io.BytesIO()
    assert store.dump_to_pgcopy(chunks, docs) == 50
    vs.reset_collections()                                        # "restart"
    again = ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name="dur", index_factory=factory)
    assert again.count() == 0
    assert again.load_from_pgcopy(io.BytesIO(chunks.getvalue()), io.BytesIO(docs.getvalue())) == 50
    assert [(d.page_content, d.metadata, s) for d, s in again.similarity_search_by_vector_with_score(q, k=12, filter={"page": 3})] == before
    assert [(d.page_content, d.metadata, s) for d, s in again.similarity_search_by_vector_with_score(q, k=50, include_deleted=True)] == before_all
    assert again.resource_hashes() == {"h1", "h3", "h4", "h5", "h6"} and again.table.find(5, 7) is not None
