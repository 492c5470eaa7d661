#!/usr/bin/env python3
"""Generate tests/golden/reference_wrapper_*.json by RUNNING the reference's own
PostgresVectorStore (imported from /root/reference, never copied) against a fake
Postgres connection whose cursor evaluates the SQL semantics with the CPU oracle.

What this pins: the wrapper conventions of
/root/reference/src/data_manager/vectorstore/postgres_vectorstore.py -- embedding ->
text formatting and parameter order (:313-315), result order = row order (:339-364),
score = 1 - distance for cosine / raw distance otherwise (:361), metadata merge
(:342-354), None metadata -> {} , add_texts metadata side effects (:131-157),
delete/from_texts/count behaviour -- composed with the oracle's distance arithmetic.

Run in the build container only (needs /root/reference):
    python tests/golden/make_reference_fixtures.py
"""
import json
import os
import re
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import knn_oracle as ko  # noqa: E402

REF = "/root/reference"


def install_stubs():
    """The reference's own technique (tests/unit/test_vectorstore_manager_batch_commit.py:8-73)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Document:
        def __init__(self, page_content="", metadata=None, **kw):
            self.page_content = page_content
            self.metadata = metadata if metadata is not None else {}

    lc = mod("langchain_core"); lc.__path__ = []
    mod("langchain_core.documents", Document=Document)
    mod("langchain_core.embeddings", Embeddings=object)
    vsm = mod("langchain_core.vectorstores", VectorStore=object); vsm.__path__ = []
    mod("langchain_core.vectorstores.base", VectorStore=object)

    class BaseRetriever:                       # stands in for the pydantic model: keyword fields become attributes
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

    mod("langchain_core.retrievers", BaseRetriever=BaseRetriever)
    cb = mod("langchain_core.callbacks"); cb.__path__ = []
    mod("langchain_core.callbacks.manager", CallbackManagerForRetrieverRun=object)
    pg = mod("psycopg2", OperationalError=Exception, Error=Exception, connect=lambda **kw: None)
    pg.__path__ = []

    def execute_values(cursor, sql, argslist, template=None, **kw):
        cursor.executed_values.append((sql, list(argslist), template))

    pg.extras = mod("psycopg2.extras", RealDictCursor=object, execute_values=execute_values, Json=lambda x: x)
    pg.extensions = mod("psycopg2.extensions", connection=object)
    pg.pool = mod("psycopg2.pool", ThreadedConnectionPool=object, PoolError=Exception)
    pg.sql = mod("psycopg2.sql")
    # the two embedders the reference's resolver knows (config_service.py:479-485): named stand-ins
    mod("langchain_huggingface", HuggingFaceEmbeddings=type("HuggingFaceEmbeddings", (), {}))
    mod("langchain_openai", OpenAIEmbeddings=type("OpenAIEmbeddings", (), {}))
    return Document


class FakeCursor:
    """Evaluates the one SELECT the store issues, with the oracle doing the arithmetic."""

    def __init__(self, db):
        self.db = db
        self.executed = []
        self.executed_values = []
        self.rowcount = 0
        self._rows = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def execute(self, sql, params=None):
        self.executed.append((sql, list(params) if params is not None else None))
        if "COUNT(*)" in sql:
            self._rows = [(len(self.db["rows"]),)]
            return
        if "DELETE" in sql:
            self.rowcount = 1
            return
        if "am.amname = 'bm25'" in sql:          # the BM25-index probe of hybrid_search (:397-418)
            self._rows = [{"relname": "idx_chunks_bm25"}] if self.db.get("bm25_hits") is not None else []
            return
        if "combined_score" in sql:
            self._hybrid(sql, params)
            return
        if "AS distance" not in sql:
            self._rows = []
            return
        op = re.search(r"c\.embedding (<=>|<->|<#>) %s::vector", sql).group(1)
        metric = {"<=>": "cosine", "<->": "l2", "<#>": "inner_product"}[op]
        emb_text, collection = params[0], params[1]
        k = params[-1]
        filters = params[2:-1]
        keys = re.findall(r"c\.metadata->>'(\w+)' = %s", sql)[1:]   # first is 'collection'
        q = np.array([float(t) for t in emb_text.strip("[]").split(",")], dtype=np.float64).astype(np.float32)
        rows = self.db["rows"]
        alive = np.ones(len(rows), np.uint8)
        for i, r in enumerate(rows):
            md = r["metadata"] or {}
            if md.get("collection") not in (None, collection):
                alive[i] = 0
            for kk, vv in zip(keys, filters):
                if str(md.get(kk)) != vv:
                    alive[i] = 0
            if "d.is_deleted = FALSE" in sql and r.get("is_deleted"):
                alive[i] = 0
        ids = np.array([r["id"] for r in rows], dtype=np.int64)
        oi, od, cnt = ko.search(self.db["vectors"], q[None], k, metric, ids=ids, alive=alive)
        by_id = {r["id"]: r for r in rows}
        out = []
        for j in range(int(cnt[0])):
            r = by_id[int(oi[0, j])]
            out.append({"id": r["id"], "chunk_text": r["chunk_text"],
                        "metadata": None if r["metadata"] is None else dict(r["metadata"]),
                        "distance": float(od[0, j]), "resource_hash": r.get("resource_hash"),
                        "display_name": r.get("display_name"), "source_type": r.get("source_type"),
                        "url": r.get("url")})
        self._rows = out

    def _where(self, sql, collection, filters):
        keys = re.findall(r"c\.metadata->>'(\w+)' = %s", sql)[1:]   # first is 'collection'
        rows = self.db["rows"]
        alive = np.ones(len(rows), np.uint8)
        for i, r in enumerate(rows):
            md = r["metadata"] or {}
            if md.get("collection") not in (None, collection):
                alive[i] = 0
            for kk, vv in zip(keys, filters):
                if str(md.get(kk)) != vv:
                    alive[i] = 0
            if "d.is_deleted = FALSE" in sql and r.get("is_deleted"):
                alive[i] = 0
        return alive

    def _hybrid(self, sql, params):
        """The scored CTE + combine of hybrid_search (:435-457): every row passing WHERE gets
        semantic_score = 1.0 - distance (oracle arithmetic) and bm25_score = hit or NULL."""
        op = re.search(r"1\.0 - \(c\.embedding (<=>|<->|<#>) %s::vector\)", sql).group(1)
        metric = {"<=>": "cosine", "<->": "l2", "<#>": "inner_product"}[op]
        emb_text, collection = params[0], params[1]
        query_text, w_s, w_b, k = params[-4], params[-3], params[-2], params[-1]
        filters = params[2:-4]
        q = np.array([float(t) for t in emb_text.strip("[]").split(",")], dtype=np.float64).astype(np.float32)
        rows = self.db["rows"]
        if not rows:
            self._rows = []
            return
        alive = self._where(sql, collection, filters)
        ids = np.array([r["id"] for r in rows], dtype=np.int64)
        oi, od, cnt = ko.search(self.db["vectors"], q[None], len(rows), metric, ids=ids, alive=alive)
        hits = self.db["bm25_hits"](query_text)
        by_id = {r["id"]: r for r in rows}
        scored = []
        for j in range(int(cnt[0])):
            rid = int(oi[0, j])
            sem = 1.0 - float(od[0, j])
            bm = hits.get(rid)
            scored.append((sem * w_s + (bm if bm is not None else 0) * w_b, rid, sem, bm))
        # ORDER BY combined_score DESC: PostgreSQL sorts float8 NaN above every other value, so NaN comes FIRST under DESC
        # (documented float8 semantics); ties by id (the reference leaves them unspecified)
        scored.sort(key=lambda c: (0, 0.0, c[1]) if c[0] != c[0] else (1, -c[0], c[1]))
        out = []
        for comb, rid, sem, bm in scored[:k]:
            r = by_id[rid]
            out.append({"id": rid, "chunk_text": r["chunk_text"],
                        "metadata": None if r["metadata"] is None else dict(r["metadata"]),
                        "semantic_score": sem, "bm25_score": bm, "combined_score": comb,
                        "resource_hash": r.get("resource_hash"), "display_name": r.get("display_name"),
                        "source_type": r.get("source_type"), "url": r.get("url")})
        self._rows = out

    def fetchall(self):
        return self._rows

    def fetchone(self):
        return self._rows[0] if self._rows else None


class FakeConn:
    def __init__(self, db):
        self.db = db
        self.cursors = []
        self.commits = 0

    def cursor(self, cursor_factory=None):
        c = FakeCursor(self.db)
        self.cursors.append(c)
        return c

    def commit(self):
        self.commits += 1

    def close(self):
        pass


class FixedEmbeddings:
    def __init__(self, dim, seed):
        self.dim, self.seed = dim, seed

    def embed_documents(self, texts):
        v = ko.gen_rows(self.seed, 5, 0, len(texts), self.dim, True, "f32")
        return [[float(x) for x in row] for row in v]

    def embed_query(self, text):
        v = ko.gen_rows(self.seed, 6, len(text), 1, self.dim, True, "f32")[0]
        return [float(x) for x in v]


def build_db(n, dim, seed):
    vec = ko.gen_rows(seed, 0, 0, n, dim, True, "f32")
    vec[17] = vec[3]                      # an exact tie
    rows = []
    for i in range(n):
        md = {"collection": "golden", "source": "web" if i % 3 else "git", "page": i % 7}
        if i % 11 == 0:
            md = None                     # NULL metadata (reference test :560-581)
        rows.append({"id": 1000 + i, "chunk_text": f"chunk {i}", "metadata": md,
                     "resource_hash": f"h{i // 4}" if i % 5 else None,
                     "display_name": f"Doc {i // 4}" if i % 5 else None,
                     "source_type": "web" if i % 2 else None, "url": f"https://x/{i // 4}" if i % 6 == 0 else None,
                     "is_deleted": i % 13 == 0})
    return {"rows": rows, "vectors": vec}


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from src.data_manager.vectorstore.postgres_vectorstore import PostgresVectorStore  # noqa: E402

    out = {"generator": "tests/golden/make_reference_fixtures.py", "reference_version": "archi v1.2.4 (/root/reference)",
           "cases": []}
    n, dim, seed = 300, 48, 2024
    db = build_db(n, dim, seed)
    emb = FixedEmbeddings(dim, seed)
    for metric in ("cosine", "l2", "inner_product"):
        for kwargs in ({}, {"filter": {"source": "web"}}, {"filter": {"page": 3}, "include_deleted": True}):
            conn = FakeConn(db)
            store = PostgresVectorStore(pg_config={}, embedding_function=emb, collection_name="golden",
                                        distance_metric=metric, connection=conn)
            res = store.similarity_search_with_score("what is the answer?", k=7, **kwargs)
            sql, params = conn.cursors[-1].executed[-1]
            out["cases"].append({
                "metric": metric, "kwargs": kwargs, "k": 7, "query_text": "what is the answer?",
                "param0_prefix": params[0][:40], "param0_len": len(params[0]), "params_tail": params[1:],
                "n_rows": n, "dim": dim, "seed": seed,
                "results": [{"page_content": d.page_content, "metadata": d.metadata, "score": s} for d, s in res],
            })
    # write path + misc behaviour
    conn = FakeConn({"rows": [], "vectors": np.zeros((0, dim), np.float32)})
    store = PostgresVectorStore(pg_config={}, embedding_function=emb, collection_name="golden", connection=conn)
    metas = [{"a": 1}, {"b": 2}]
    ids = store.add_texts(["t0", "t1"], metadatas=metas, ids=["id-0", "id-1"], document_id=42)
    sql, rows, template = conn.cursors[-1].executed_values[-1]
    out["write"] = {
        "returned_ids": ids, "metadatas_after": metas, "template": template,
        "rows": [[r[0], r[1], r[2], len(r[3]), json.loads(r[4])] for r in rows],
        "commits": conn.commits, "empty_add": store.add_texts([]),
        "delete_none": store.delete(), "delete_ids": store.delete(ids=["id-0"]),
        "delete_doc": store.delete(document_id=42),
    }
    # hybrid_search (:366-491): BM25 hits are a fixed id -> score table (the scorer itself is third-party)
    def hits_for(query_text):
        base = sum(map(ord, query_text))
        return {1000 + (base * 7 + 13 * j) % n: round(0.25 + ((base + j * j) % 17) / 4.0, 2) for j in range(40)}

    out["hybrid"] = []
    for metric, sign, w_s, w_b, kwargs in (("cosine", 1.0, 0.7, 0.3, {}), ("cosine", -1.0, 0.7, 0.3, {}),
                                           ("cosine", 1.0, 0.5, 0.5, {"filter": {"source": "web"}}),
                                           ("l2", 1.0, 0.7, 0.3, {"include_deleted": True}),
                                           ("inner_product", -1.0, 0.2, 0.8, {})):
        db_h = dict(db)
        db_h["bm25_hits"] = lambda qt, sign=sign: {i: sign * v for i, v in hits_for(qt).items()}
        conn = FakeConn(db_h)
        store = PostgresVectorStore(pg_config={}, embedding_function=emb, collection_name="golden",
                                    distance_metric=metric, connection=conn)
        qt = "which detector measures muons?"
        res = store.hybrid_search(qt, k=9, semantic_weight=w_s, bm25_weight=w_b, **kwargs)
        sql, params = conn.cursors[-1].executed[-1]
        out["hybrid"].append({
            "metric": metric, "kwargs": kwargs, "k": 9, "query_text": qt, "semantic_weight": w_s, "bm25_weight": w_b,
            "params_tail": params[1:], "bm25_hits": {str(i): v for i, v in db_h["bm25_hits"](qt).items()},
            "n_rows": n, "dim": dim, "seed": seed,
            "results": [{"page_content": d.page_content, "metadata": d.metadata, "score": s} for d, s in res]})
    conn = FakeConn(db)                                   # no BM25 index -> RuntimeError (:415-418)
    store = PostgresVectorStore(pg_config={}, embedding_function=emb, collection_name="golden", connection=conn)
    try:
        store.hybrid_search("q", k=3)
    except RuntimeError as e:
        out["hybrid_no_index_error"] = str(e)
    db_e = {"rows": [], "vectors": np.zeros((0, dim), np.float32), "bm25_hits": lambda qt: {}}
    store = PostgresVectorStore(pg_config={}, embedding_function=emb, collection_name="golden", connection=FakeConn(db_e))
    out["hybrid_empty_table"] = store.hybrid_search("q", k=3)          # falls back to the semantic query (:467-469)
    # ---- the reference's retrievers (the callers of the store: semantic_retriever.py:39, grading_retriever.py:25,
    # hybrid_retriever.py:85-103), run over the reference store AND over this build's store; identical output
    # is asserted here and the reference-side output is recorded for the CPU suite
    from src.data_manager.vectorstore.retrievers import GradingRetriever, HybridRetriever, SemanticRetriever  # noqa: E402
    from archi_amd import vectorstore as avs  # noqa: E402
    from tests.fake_index import OracleIndex  # noqa: E402

    # the retrievers log metadata['filename'] (semantic_retriever.py:44), which the ingestion always sets
    # (manager.py:316): the retriever table carries it on every row
    db_ret = {"vectors": db["vectors"], "rows": []}
    for i, r in enumerate(db["rows"]):
        r2 = dict(r)
        r2["metadata"] = dict(r["metadata"] or {"collection": "golden"}, filename=f"file{i // 4}.txt")
        db_ret["rows"].append(r2)

    def archi_store(hybrid_hits):
        avs.reset_collections()
        cls = avs.ArchiHipHybridVectorStore if hybrid_hits is not None else avs.ArchiHipVectorStore
        kw = {}
        if hybrid_hits is not None:
            class Tab:
                def scores(self, query, table):
                    return dict(hybrid_hits(query))
            kw["bm25"] = Tab()
        st = cls({"hip": {"dtype": "f32"}}, emb, collection_name="golden", distance_metric="cosine",
                 index_factory=lambda d, cap, dt, m: OracleIndex(d, cap, dtype=dt, metric=m), **kw)
        col = st._collection(dim)
        for r in db_ret["rows"]:
            col.table.rows[r["id"]] = {"document_id": r["id"], "chunk_index": 0, "text": r["chunk_text"],
                                       "metadata": dict(r["metadata"] or {})}
            col.table.register_document(r["id"], resource_hash=r["resource_hash"], display_name=r["display_name"],
                                        source_type=r["source_type"], url=r["url"], is_deleted=r["is_deleted"])
        col.index.add(db_ret["vectors"], ids=[r["id"] for r in db_ret["rows"]])
        return st

    def dump(res):
        return [({"page_content": d.page_content, "metadata": d.metadata, "score": s} if isinstance(t, tuple) else
                 {"page_content": t.page_content, "metadata": t.metadata})
                for t in res for d, s in [t if isinstance(t, tuple) else (t, None)]]

    dm_config = {"embedding_name": "E", "embedding_class_map": {"E": {"kwargs": {"model_name": "all-MiniLM-L6-v2"}}}}
    qtext = "how do I request grid certificates?"
    out["retrievers"] = {"query_text": qtext}
    bm = lambda qt: {i: v for i, v in hits_for(qt).items()}           # noqa: E731
    for name, make, hybrid in (
            ("semantic_k3", lambda st: SemanticRetriever(st, dm_config, k=3), None),
            ("grading_k3", lambda st: GradingRetriever(st, k=3), None),
            ("hybrid_native_k5", lambda st: HybridRetriever(st, k=5, bm25_weight=0.5, semantic_weight=0.5), bm),
            ("hybrid_fallback_k5", lambda st: HybridRetriever(st, k=5), None)):
        db_r = dict(db_ret)
        if hybrid is not None:
            db_r["bm25_hits"] = hybrid
        ref_store = PostgresVectorStore(pg_config={}, embedding_function=emb, collection_name="golden",
                                        connection=FakeConn(db_r))
        if hybrid is None and name.startswith("hybrid"):
            # a store without hybrid_search: what HybridRetriever sees with this build's plain store
            ref_out = dump(HybridRetriever(type("NoHybrid", (), {
                "similarity_search_with_score": ref_store.similarity_search_with_score})(), k=5)._get_relevant_documents(qtext))
        else:
            ref_out = dump(make(ref_store)._get_relevant_documents(qtext))
        mine_out = dump(make(archi_store(hybrid))._get_relevant_documents(qtext))
        assert mine_out == ref_out, (name, mine_out[:1], ref_out[:1])
        out["retrievers"][name] = ref_out
    out["retrievers"]["bm25_hits"] = {str(i): v for i, v in bm(qtext).items()}
    out["retrievers"]["reference_retrievers_over_archi_store_identical"] = True
    avs.reset_collections()
    # ---- hybrid with NaN semantic scores: two zero-vector rows (cosine distance NaN) -> NaN combined score, which
    # PostgreSQL's ORDER BY ... DESC ranks first
    db_n = {"rows": [dict(r) for r in db["rows"]], "vectors": db["vectors"].copy()}
    db_n["vectors"][20] = 0.0
    db_n["vectors"][41] = 0.0
    db_n["bm25_hits"] = lambda qt: dict(hits_for(qt))
    store = PostgresVectorStore(pg_config={}, embedding_function=emb, collection_name="golden", connection=FakeConn(db_n))
    qt = "which detector measures muons?"
    res = store.hybrid_search(qt, k=6, semantic_weight=0.7, bm25_weight=0.3)
    out["hybrid_nan"] = {"metric": "cosine", "k": 6, "query_text": qt, "semantic_weight": 0.7, "bm25_weight": 0.3,
                         "zero_rows": [20, 41], "bm25_hits": {str(i): v for i, v in hits_for(qt).items()},
                         "n_rows": n, "dim": dim, "seed": seed,
                         "results": [{"page_content": d.page_content, "metadata": d.metadata,
                                      "score": None if s != s else s} for d, s in res]}
    # ---- N4: the reference's embedding-class resolver (src/utils/config_service.py:470-496) and its dimension bookkeeping
    # (src/cli/managers/templates_manager.py:393-431 -> vector({{embedding_dimensions}}) in init.sql:266), both RUN here
    from src.utils.config_service import ConfigService  # noqa: E402
    class_map = {
        "HuggingFaceEmbeddings": {"class": "HuggingFaceEmbeddings", "kwargs": {"model_name": "sentence-transformers/all-MiniLM-L6-v2"},
                                  "similarity_score_reference": 10},
        "OpenAIEmbeddings": {"kwargs": {"model": "text-embedding-3-small"}, "dimensions": 1536},
        "ArchiHipEmbeddings": {"class": "ArchiHipEmbeddings", "kwargs": {"model_name": "BAAI/bge-base-en"}, "dimensions": 768},
        "Custom": {"class": "SomethingElse"},
        "NoneCfg": None,
        "AlreadyCallable": {"class": 123},
    }
    resolved = ConfigService._resolve_embedding_classes(class_map)

    def plain(v):
        return {"__class__": v.__name__} if isinstance(v, type) else v

    out["config"] = {"embedding_class_map": class_map,
                     "resolved": {k: {kk: plain(vv) for kk, vv in e.items()} for k, e in resolved.items()},
                     "resolved_empty": ConfigService._resolve_embedding_classes({}),
                     "init_sql_dimensions": []}
    import tempfile
    from pathlib import Path
    from jinja2 import Environment, FileSystemLoader
    from src.cli.managers import templates_manager as tm  # noqa: E402
    env = Environment(loader=FileSystemLoader(os.path.join(REF, "src", "cli", "templates")))
    for name, cmap in (("all-MiniLM-L6-v2", {}), ("text-embedding-3-large", {}), ("SomethingUnknown", {}),
                       ("ArchiHipEmbeddings", class_map), ("OpenAIEmbeddings", class_map), ("HuggingFaceEmbeddings", class_map),
                       (None, class_map)):
        dm = {"embedding_class_map": cmap}
        if name is not None:
            dm["embedding_name"] = name
        with tempfile.TemporaryDirectory() as td:
            ctx = types.SimpleNamespace(
                plan=types.SimpleNamespace(get_service=lambda s: types.SimpleNamespace(enabled=False)),
                secrets_manager=types.SimpleNamespace(get_secret=lambda s: ""),
                config_manager=types.SimpleNamespace(config={"data_manager": dm}), base_dir=Path(td))
            tm.TemplateManager._render_postgres_init(types.SimpleNamespace(env=env), ctx)
            sql = open(os.path.join(td, "init.sql")).read()
        dims = sorted(set(int(x) for x in re.findall(r"vector\((\d+)\)", sql)))
        assert len(dims) == 1, dims
        out["config"]["init_sql_dimensions"].append({"embedding_name": name, "uses_class_map": bool(cmap), "dimensions": dims[0]})
    try:
        PostgresVectorStore(pg_config={}, embedding_function=emb, distance_metric="manhattan")
    except ValueError as e:
        out["bad_metric_error"] = str(e)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_wrapper.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
