"""TEST INFRASTRUCTURE shared by tests/test_refresh_cpu.py (OracleIndex stand-in) and tests/test_store_gpu.py (real index): two
store PROCESSES' worth of state in one test process. The reference keeps one table in Postgres, written by the data-manager
process (src/data_manager/vectorstore/manager.py:177-214) and read through a fresh PostgresVectorStore per chat request
(src/archi/archi.py:61-65); here each process has its own in-process collection cache, which `Proc` swaps in and out of
archi_amd.vectorstore._collections, and the "table" between them is what the writer dumps as COPY streams."""
import io

import numpy as np

from archi_amd import pgbridge
from archi_amd import vectorstore as vs


class Proc:
    """One process's collection cache: `with proc:` makes it the module's cache for the duration."""

    def __init__(self):
        self.cols = {}

    def __enter__(self):
        self.saved = dict(vs._collections)
        vs._collections.clear()
        vs._collections.update(self.cols)
        return self

    def __exit__(self, *exc):
        self.cols = dict(vs._collections)
        vs._collections.clear()
        vs._collections.update(self.saved)
        return False

    def close(self):
        with self:
            vs.reset_collections()


class Table:
    """What the writer's Postgres table would answer, produced from the writer's store: the id / version listing, rows by id,
    the documents columns. Versions are kept here (an xmin stand-in): a row's version is the transaction counter of the
    statement that last wrote it."""

    def __init__(self, writer_store):
        self.store, self.txid, self.ver = writer_store, 100, {}
        self.overrides = {}          # row id -> (text, metadata, vector): rows "UPDATEd in place" behind the writer store's back

    def commit(self):
        """Stamp every row the writer has (re)written since the last commit with a new transaction id."""
        self.txid += 1
        t = self.store.table
        live = set(t.live_rids().tolist())
        for rid in live:
            self.ver.setdefault(rid, self.txid)
        for rid in [r for r in self.ver if r not in live]:
            del self.ver[rid]

    def update_in_place(self, rid, text, vector):
        self.txid += 1
        t = self.store.table
        md = t.metadata_at(t.pos(rid))
        self.overrides[rid] = (text, md, np.asarray(vector, np.float32))
        self.ver[rid] = self.txid

    def ids_stream(self, with_versions=True):
        rids = sorted(self.store.table.live_rids().tolist())
        out = io.BytesIO()
        pgbridge.write_pgcopy_ids(out, rids, [self.ver[r] for r in rids] if with_versions else None)
        return io.BytesIO(out.getvalue())

    def rows_stream(self, ids=None):
        chunks = io.BytesIO()
        self.store.dump_to_pgcopy(chunks, only_ids=ids)
        if not self.overrides:
            return io.BytesIO(chunks.getvalue())
        rows = []
        for blk in pgbridge.iter_pgcopy_chunks(io.BytesIO(chunks.getvalue())):
            for i, rid in enumerate(blk["ids"].tolist()):
                text, md, vec = blk["text_bytes"][i].decode("utf-8"), blk["metadata"][i], blk["vectors"][i]
                if rid in self.overrides:
                    text, md, vec = self.overrides[rid]
                rows.append((rid, blk["document_ids"][i], int(blk["chunk_index"][i]), text, md, vec))
        out = io.BytesIO()
        pgbridge.write_pgcopy_chunks(out, rows)
        return io.BytesIO(out.getvalue())

    def documents_stream(self):
        out = io.BytesIO()
        pgbridge.write_pgcopy_documents(out, [dict(d, id=k) for k, d in self.store.table.documents.items()])
        return io.BytesIO(out.getvalue())


def answers(store, queries, hybrid):
    """What a chat request sees: plain, filtered and include_deleted similarity searches (+ hybrid on the hybrid class)."""
    out = []
    for q in queries:
        qv = [float(x) for x in q]
        for kw in ({}, {"filter": {"source": "web"}}, {"filter": {"source": "git"}, "include_deleted": True}):
            out.append([(d.page_content, d.metadata, s) for d, s in store.similarity_search_by_vector_with_score(qv, k=10, **kw)])
    if hybrid:
        for text in ("muon trigger", "calorimeter alignment notes", "doc 7"):
            out.append([(d.page_content, d.metadata, s) for d, s in store.hybrid_search(text, k=8)])
    return out


def unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


WORDS = ["muon", "trigger", "calorimeter", "grid", "job", "alignment", "tracker", "release", "notes", "luminosity", "beam", "pixel"]


def ingest(store, rng, doc, n, d, kind=None):
    kind = kind or ("web" if doc % 2 else "git")
    vec = unit(rng, n, d)
    texts = [f"doc {doc} chunk {i} " + " ".join(rng.choice(WORDS, size=4)) for i in range(n)]
    store.add_texts(texts, metadatas=[{"source": kind, "doc": doc, "resource_hash": f"h{doc}"} for _ in range(n)],
                    document_id=doc, embeddings=vec)
    store.table.register_document(doc, resource_hash=f"h{doc}", display_name=f"Doc {doc}", source_type=kind, url=None, is_deleted=False)
    return vec


def writer_moves(w, table, rng, d, first_new_doc):
    """What an ingestion run does between two refreshes: new documents, a hard delete, a re-ingest (delete + add = new row ids),
    a soft delete, a rename, and one row rewritten in place under its id (ON CONFLICT DO UPDATE, postgres_vectorstore.py:168-180)."""
    for doc in range(first_new_doc, first_new_doc + 5):
        ingest(w, rng, doc, 40, d)
    w.delete(document_id=3)
    del w.table.documents[3]
    w.delete(document_id=7)
    ingest(w, rng, 7, 35, d)
    w.table.register_document(9, is_deleted=True)
    w.table.register_document(11, display_name="Doc eleven, renamed")
    table.commit()
    victim = int(w.table.rids_of_document(13)[2])
    newvec = unit(rng, 1, d)[0]
    table.update_in_place(victim, "doc 13 chunk 2 rewritten in place: beam pixel muon", newvec)
    return victim, newvec


def concurrent_refresh_scenario(mk_store, cycles=25):
    """The reader process under load: 8 request threads run filtered searches on the reader's collection while it is refreshed
    again and again from a writer that adds, deletes, re-ingests and soft-deletes documents. A search that STARTS after a
    refresh returned never brings back a chunk of a document that refresh deleted or soft-deleted; every chunk satisfies the
    filter; no exception (stale masks are rebuilt inside the store). Afterwards the reader equals a store loaded from scratch.
    mk_store(metric, **hip_cfg) -> store of collection "shared". The writer is only ever a source of rows, never searched, so
    it lives in the same process under another distance metric (the collection cache is keyed by (name, metric)): no cache
    swapping under running threads. Returns the number of searches the readers completed."""
    import threading
    rng = np.random.default_rng(41)
    d = 64
    w = mk_store("l2", capacity=512)
    for doc in range(1, 41):
        ingest(w, rng, doc, 60, d)
    table = Table(w)
    table.commit()
    r = mk_store("cosine", capacity=512)                 # small first reservation: the refreshes grow and reclaim
    r.load_from_pgcopy(table.rows_stream(), table.documents_stream(), versions_stream=table.ids_stream())
    gone, lock, stop, errors, searches = set(), threading.Lock(), threading.Event(), [], [0]
    queries = unit(rng, 8, d)

    def reader(j):
        try:
            q = [float(x) for x in queries[j]]
            while not stop.is_set():
                with lock:
                    dead_before = set(gone)
                res = r.similarity_search_by_vector_with_score(q, k=10, filter={"source": "web"})
                for doc_, _ in res:
                    assert doc_.metadata["source"] == "web", doc_.page_content
                    assert doc_.metadata["doc"] not in dead_before, f"chunk of dead document {doc_.metadata['doc']} returned"
                assert len(res) == 10
                searches[0] += 1
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=reader, args=(j,)) for j in range(8)]
    for th in threads:
        th.start()
    nxt = 41
    try:
        for cycle in range(cycles):
            ingest(w, rng, nxt, 50, d); nxt += 1
            live_docs = sorted(k for k, v in w.table.documents.items() if not v.get("is_deleted"))
            hard = int(rng.choice(live_docs))
            w.delete(document_id=hard); w.table.documents.pop(hard, None)
            dead_now = {hard}
            if cycle % 2:
                soft = int(rng.choice([k for k in live_docs if k % 2 and k != hard]))
                w.table.register_document(soft, is_deleted=True)
                dead_now.add(soft)
            if cycle % 3 == 0:
                re = int(rng.choice([k for k in live_docs if k not in dead_now]))
                w.delete(document_id=re)
                ingest(w, rng, re, 45, d, kind="web" if re % 2 else "git")
            table.commit()
            r.refresh_from_pgcopy(table.ids_stream(), lambda ids: table.rows_stream(np.asarray(ids).tolist()), table.documents_stream())
            with lock:
                gone.update(dead_now)                                # only now may the readers insist on it
    finally:
        stop.set()
        for th in threads:
            th.join()
    assert not errors, errors[:3]
    got = answers(r, queries[:4], hybrid=False)
    final = (table.rows_stream(), table.documents_stream())
    vs.reset_collections()
    scratch = mk_store("cosine")
    scratch.load_from_pgcopy(*final)
    assert answers(scratch, queries[:4], hybrid=False) == got
    vs.reset_collections()
    return searches[0]
