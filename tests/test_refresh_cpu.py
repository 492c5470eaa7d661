"""CPU suite: ArchiHipVectorStore.refresh_from_pgcopy -- a chat process's collection reconciled with the table the data-manager
process writes (round-5 review, missing #1; reference: src/archi/archi.py:61-65, src/data_manager/vectorstore/manager.py:177-214).
Host logic over the oracle-backed index stand-in; the same scenario runs on the real index in tests/test_store_gpu.py."""
import io

import numpy as np
import pytest

from archi_amd import pgbridge
from archi_amd import vectorstore as vs
from archi_amd.vectorstore import ArchiHipHybridVectorStore, ArchiHipVectorStore
from tests.fake_index import OracleIndex
from tests.refresh_scenario import Proc, Table, answers, ingest, unit, writer_moves


def factory(dim, capacity, dtype, metric):
    return OracleIndex(dim, capacity, dtype=dtype, metric=metric)


class NoEmbed:
    def embed_documents(self, texts):
        raise AssertionError("vectors are handed in")

    def embed_query(self, text):
        return [float(x) for x in unit(np.random.default_rng(len(text)), 1, 32)[0]]


@pytest.fixture(autouse=True)
def fresh():
    vs.reset_collections()
    yield
    vs.reset_collections()


def mk(cls=ArchiHipHybridVectorStore):
    kw = {"bm25": vs.HostBm25()} if cls is ArchiHipHybridVectorStore else {}
    return cls({"hip": {"dtype": "f32"}}, NoEmbed(), collection_name="shared", index_factory=factory, **kw)


def test_id_stream_round_trip_and_rejects():
    for idb in (4, 8):
        out = io.BytesIO()
        pgbridge.write_pgcopy_ids(out, [5, 1, 70000], [9, 8, 2 ** 40], id_bytes=idb)
        ids, ver = pgbridge.read_pgcopy_ids(io.BytesIO(out.getvalue()))
        assert ids.tolist() == [5, 1, 70000] and ver.tolist() == [9, 8, 2 ** 40]
    out = io.BytesIO()
    pgbridge.write_pgcopy_ids(out, [3, 4])
    ids, ver = pgbridge.read_pgcopy_ids(io.BytesIO(out.getvalue()))
    assert ids.tolist() == [3, 4] and ver is None
    out = io.BytesIO()
    pgbridge.write_pgcopy_ids(out, [])
    assert pgbridge.read_pgcopy_ids(io.BytesIO(out.getvalue()))[0].tolist() == []
    with pytest.raises(ValueError):
        pgbridge.read_pgcopy_ids(io.BytesIO(out.getvalue()[:-1]))           # truncated
    # a hand-written known answer: header, (id int4 = 258), trailer
    raw = pgbridge.SIGNATURE + b"\x00" * 8 + b"\x00\x01" + b"\x00\x00\x00\x04" + b"\x00\x00\x01\x02" + b"\xff\xff"
    assert pgbridge.read_pgcopy_ids(io.BytesIO(raw))[0].tolist() == [258]


@pytest.mark.parametrize("with_versions", [True, False])
def test_reader_after_refresh_equals_a_store_loaded_from_scratch(with_versions):
    rng = np.random.default_rng(77)
    d = 32
    wp, rp, sp = Proc(), Proc(), Proc()
    with wp:
        w = mk()
        for doc in range(1, 31):
            ingest(w, rng, doc, 20 + doc % 7, d)
        table = Table(w)
        table.commit()
        s0 = (table.rows_stream(), table.documents_stream(), table.ids_stream())
    with rp:
        r = mk()
        n0 = r.load_from_pgcopy(s0[0], s0[1], versions_stream=s0[2] if with_versions else None)
        assert n0 == r.count()
    with wp:
        victim, newvec = writer_moves(w, table, rng, d, 31)
        if not with_versions:
            table.overrides.clear()                 # an id-only listing cannot see an in-place rewrite (documented)
        queries = unit(rng, 5, d)
        if with_versions:
            queries[0] = newvec                      # the rewritten row must answer with its NEW vector and text
        asked = []

        def fetch(ids):                          # (runs inside the reader's refresh: answered from the WRITER's process)
            asked.append(np.asarray(ids).copy())
            return table_rows(wp, table, ids)
        ids_s, docs_s = table.ids_stream(with_versions), table.documents_stream()
        final = (table.rows_stream(), table.documents_stream())
    with rp:
        epoch0 = r._collection().index.layout()[1]
        stats = r.refresh_from_pgcopy(ids_s, fetch, docs_s)
        assert stats["removed"] > 0 and stats["added"] == 5 * 40 + 35 and stats["documents_changed"] >= 3
        assert stats["updated"] == (1 if with_versions else 0)
        assert len(asked) == 1 and len(asked[0]) == stats["fetched"]          # only what the collection lacked travelled
        assert r._collection().index.layout()[1] != epoch0
        got = answers(r, queries, hybrid=True)
        assert r.max_row_id() == max(table.ver) if with_versions else True
    with sp:
        scratch = mk()
        scratch.load_from_pgcopy(*final)
        want = answers(scratch, queries, hybrid=True)
    assert got == want
    if with_versions:
        assert any("rewritten in place" in d_[0] for d_ in got[0])
    # idempotent: the same listing again moves nothing -- no fetch, no epoch change, no cache invalidation
    with wp:
        ids_s, docs_s = table.ids_stream(with_versions), table.documents_stream()
    with rp:
        t = r.table
        r.similarity_search_by_vector_with_score([float(x) for x in queries[1]], k=3, filter={"source": "web"})
        before = (r._collection().index.layout(), t.version, t.doc_version, t.text_epoch, dict(t.where_cache))
        stats = r.refresh_from_pgcopy(ids_s, None, docs_s)
        assert stats == {"removed": 0, "added": 0, "updated": 0, "documents_changed": 0, "fetched": 0}
        after = (r._collection().index.layout(), t.version, t.doc_version, t.text_epoch, dict(t.where_cache))
        assert before[:4] == after[:4] and before[4].keys() == after[4].keys() and len(after[4]) >= 1
    for p in (wp, rp, sp):
        p.close()


def test_tail_refresh_and_first_refresh_of_an_empty_reader():
    rng = np.random.default_rng(3)
    d = 32
    wp, rp = Proc(), Proc()
    with wp:
        w = mk(ArchiHipVectorStore)
        for doc in range(1, 6):
            ingest(w, rng, doc, 10, d)
        table = Table(w)
        table.commit()
        ids_s, docs_s = table.ids_stream(), table.documents_stream()
    with rp:
        r = mk(ArchiHipVectorStore)
        assert r.max_row_id() == 0
        # a reader that holds nothing yet: the full refresh IS the load
        stats = r.refresh_from_pgcopy(ids_s, lambda ids: table_rows(wp, table, ids), docs_s)
        assert stats["added"] == 50 and r.count() == 50
        top = r.max_row_id()
    with wp:
        for doc in range(6, 9):
            ingest(w, rng, doc, 10, d)
        table.commit()
        tail = io.BytesIO()
        live = w.table.live_rids()
        w.dump_to_pgcopy(tail, only_ids=live[live > top].tolist())          # ... WHERE id > %s
        docs_s = table.documents_stream()
    with rp:
        assert r.refresh_tail_from_pgcopy(io.BytesIO(tail.getvalue()), docs_s) == 30 and r.count() == 80
        with pytest.raises(ValueError):
            r.refresh_tail_from_pgcopy(io.BytesIO(tail.getvalue()))          # the same tail again: rows already present
        assert r.count() == 80
        # new or rewritten rows without a way to fetch them: an error, nothing half-applied
    with wp:
        ingest(w, rng, 9, 4, d)
        table.commit()
        ids_s = table.ids_stream()
    with rp:
        with pytest.raises(ValueError):
            r.refresh_from_pgcopy(ids_s, None)
        assert r.count() == 80
    for p in (wp, rp):
        p.close()


def table_rows(wp, table, ids):
    with wp:
        return table.rows_stream(np.asarray(ids).tolist())


def test_versions_survive_vacuum_and_unknown_versions_are_refetched():
    rng = np.random.default_rng(9)
    d = 32
    wp, rp = Proc(), Proc()
    with wp:
        w = mk(ArchiHipVectorStore)
        for doc in range(1, 4):
            ingest(w, rng, doc, 10, d)
        table = Table(w)
        table.commit()
        s0 = (table.rows_stream(), table.documents_stream())
        ids_s = table.ids_stream()
    with rp:
        r = mk(ArchiHipVectorStore)
        r.load_from_pgcopy(*s0)                                      # no versions_stream: every row's version is unknown here
        asked = []
        stats = r.refresh_from_pgcopy(ids_s, lambda ids: (asked.append(len(ids)), table_rows(wp, table, ids))[1])
        assert asked == [30] and stats["updated"] == 30 and stats["added"] == 0          # re-read once ...
        with wp:
            ids_s = table.ids_stream()
        assert r.refresh_from_pgcopy(ids_s, None)["fetched"] == 0                        # ... and vouched for from then on
        t = r.table
        rids = t.live_rids()
        v0 = t.versions_of(rids).copy()
        assert (v0 > 0).all()
        t.kill(int(rids[0])); r._collection().index.remove([int(rids[0])])
        t.vacuum()
        assert np.array_equal(t.versions_of(rids[1:]), v0[1:])
    for p in (wp, rp):
        p.close()


def test_filtered_readers_never_see_a_dead_row_while_the_collection_is_refreshed_host_logic():
    """The store's side of the snapshot protocol under a refresh (table lock, layout epoch, stale-mask retry), over the
    oracle-backed stand-in made thread-safe with one lock; the real index runs the same scenario in tests/test_store_gpu.py."""
    import threading
    from tests.refresh_scenario import concurrent_refresh_scenario

    class LockedOracleIndex(OracleIndex):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self._mu = threading.RLock()

    for name in ("add", "remove", "search", "lookup", "layout", "count", "fetch", "distances"):
        def wrap(fn):
            def locked(self, *a, **kw):
                with self._mu:
                    return fn(self, *a, **kw)
            return locked
        setattr(LockedOracleIndex, name, wrap(getattr(OracleIndex, name)))

    def mk_store(metric, **hipcfg):
        return ArchiHipVectorStore({"hip": dict({"dtype": "f32"}, **hipcfg)}, NoEmbed(), collection_name="shared", distance_metric=metric,
                                   index_factory=lambda dim, cap, dtype, m: LockedOracleIndex(dim, cap, dtype=dtype, metric=m))
    assert concurrent_refresh_scenario(mk_store, cycles=8) > 5


def test_rows_that_vanish_between_the_listing_and_the_fetch_are_left_for_the_next_refresh():
    """The listing and the row fetch are two statements: without a repeatable-read transaction a listed row can be gone when its
    columns are asked for (or come back with a NULL embedding, or belong to another collection by then). Such a row is skipped --
    never invented --, its version stays unknown, and the next refresh asks for it again."""
    rng = np.random.default_rng(21)
    d = 32
    wp, rp = Proc(), Proc()
    with wp:
        w = mk(ArchiHipVectorStore)
        for doc in range(1, 5):
            ingest(w, rng, doc, 8, d)
        table = Table(w)
        table.commit()
        ids_s = table.ids_stream()
        live = sorted(w.table.live_rids().tolist())
    hold_back = set(live[-3:])
    asked = []

    def fetch_without_some(ids):
        asked.append(sorted(np.asarray(ids).tolist()))
        with wp:
            rows = []
            for blk in pgbridge.iter_pgcopy_chunks(table.rows_stream([i for i in np.asarray(ids).tolist()])):
                for i, rid in enumerate(blk["ids"].tolist()):
                    if rid in hold_back and rid != max(hold_back):
                        continue                                           # gone by the time of the fetch
                    emb = None if rid == max(hold_back) else blk["vectors"][i]      # ... or present with a NULL embedding
                    rows.append((rid, blk["document_ids"][i], int(blk["chunk_index"][i]), blk["text_bytes"][i].decode(), blk["metadata"][i], emb))
            out = io.BytesIO()
            pgbridge.write_pgcopy_chunks(out, rows)
            return io.BytesIO(out.getvalue())
    with rp:
        r = mk(ArchiHipVectorStore)
        stats = r.refresh_from_pgcopy(ids_s, fetch_without_some)
        assert stats["added"] == len(live) - 3 and r.count() == len(live) - 3
        with wp:
            ids_s = table.ids_stream()
        stats = r.refresh_from_pgcopy(ids_s, lambda ids: table_rows(wp, table, ids))
        assert asked[0] == live and stats["added"] == 3 and stats["fetched"] == 3 and r.count() == len(live)
        with wp:
            ids_s = table.ids_stream()
        assert r.refresh_from_pgcopy(ids_s, None)["fetched"] == 0
        # a fetch that answers with a row nobody asked for is refused before anything is applied
        with wp:
            ingest(w, rng, 9, 2, d)
            table.commit()
            ids_s = table.ids_stream()
        with pytest.raises(ValueError):
            r.refresh_from_pgcopy(ids_s, lambda ids: table_rows(wp, table, live[:2]))
        assert r.count() == len(live)
    for p in (wp, rp):
        p.close()
