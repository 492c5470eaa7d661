"""ArchiHipVectorStore -- drop-in for the reference's PostgresVectorStore
(/root/reference/src/data_manager/vectorstore/postgres_vectorstore.py:25-585):
same constructor, method names, keyword arguments, return types, score
semantics and exceptions, with the `embedding <op> query ORDER BY distance
LIMIT k` scan (:317-332) executed by hand-written HIP kernels on an
HBM-resident index instead of pgvector.

What lives where
  * vectors            -> HipIndex (archi_amd/index.py -> libarchi_hip.so), one per
                          (collection, metric), cached at process level because the
                          reference re-creates the store per chat request
                          (src/archi/archi.py:61-65).
  * chunk text + JSONB -> ChunkTable (archi_amd/chunktable.py, host memory, columnar): the
                          columns of `document_chunks` other than `embedding`
                          (src/cli/templates/init.sql:256-274) and the joined
                          `documents` columns used by the query (:323-326).
`hybrid_search` is deliberately absent from ArchiHipVectorStore: HybridRetriever then falls back to
the semantic leg (src/data_manager/vectorstore/retrievers/hybrid_retriever.py:55-62), which is what a
reference deployment without pg_textsearch does. ArchiHipHybridVectorStore (below) adds it when a BM25
scorer is attached.
"""
from __future__ import annotations

import json
import logging
import os
import math
import re
import threading
import uuid
from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple, Type

import numpy as np

try:  # LangChain is optional: the interfaces are duck-typed
    from langchain_core.documents import Document  # type: ignore
    from langchain_core.vectorstores import VectorStore as _VectorStoreBase  # type: ignore
except Exception:  # pragma: no cover - langchain absent in this image
    class Document:  # minimal stand-in with the two attributes every caller uses
        def __init__(self, page_content: str, metadata: Optional[Dict[str, Any]] = None):
            self.page_content = page_content
            self.metadata = metadata if metadata is not None else {}

        def __repr__(self) -> str:
            return f"Document(page_content={self.page_content!r}, metadata={self.metadata!r})"

    _VectorStoreBase = object

DISTANCE_OPS = {"cosine": "<=>", "l2": "<->", "inner_product": "<#>"}  # :74-78
log = logging.getLogger("archi_amd.vectorstore")     # the reference logs through src/utils/logging.py:23-40 (get_logger(__name__))


from .chunktable import ChunkTable, meta_text  # noqa: E402  (columnar host table)
from ._lib import StaleFilterError  # noqa: E402


class _Collection:
    def __init__(self, index: Any, table: ChunkTable) -> None:
        self.index = index
        self.table = table


_FLAT = (str, int, float, bool, type(None))


def _uuid4_many(n: int) -> List[str]:
    """n random (version 4, RFC 4122 variant) UUID strings -- what `str(uuid.uuid4())` gives (reference :128), formatted
    in bulk: uuid.uuid4() + str() is 3 us a piece, a quarter of the store-side time per chunk at ingestion."""
    if n <= 0:
        return []
    raw = np.frombuffer(os.urandom(16 * n), dtype=np.uint8).reshape(n, 16).copy()
    raw[:, 6] = (raw[:, 6] & 0x0F) | 0x40
    raw[:, 8] = (raw[:, 8] & 0x3F) | 0x80
    h = raw.tobytes().hex()
    return [f"{h[j:j + 8]}-{h[j + 8:j + 12]}-{h[j + 12:j + 16]}-{h[j + 16:j + 20]}-{h[j + 20:j + 32]}" for j in range(0, 32 * n, 32)]


def _jsonb(metadata: Dict[str, Any]) -> Dict[str, Any]:
    """What a JSONB column gives back for this dict (reference :157: Json(metadata)): a deep copy through JSON. Flat
    dicts of str keys and scalar values -- the ingestion path's metadata -- are copied directly (the JSON round trip was
    a fifth of the store-side time per chunk); anything nested, keyed by non-strings, or holding NaN / inf goes through
    json like before."""
    if type(metadata) is dict:
        for k, v in metadata.items():
            if type(k) is not str or type(v) not in _FLAT or (type(v) is float and (v != v or v in (float("inf"), float("-inf")))):
                break
        else:
            return dict(metadata)
    return json.loads(json.dumps(metadata))


def _suspect_rows(vecs: np.ndarray) -> np.ndarray:
    """Rows whose pgvector distance to a finite, non-zero query can come out NaN: a non-finite element, a float32 sum of
    squares that underflows to 0 (cosine: dot / sqrt(0 * nb)) or overflows. A generous superset, decided on the host at
    insert time; hybrid_search asks the GPU for the exact distance of these few rows, because Postgres sorts a NaN
    combined score FIRST under ORDER BY ... DESC (postgres_vectorstore.py:455-457) while the top-k scan ranks NaN last."""
    v = np.asarray(vecs)
    if v.dtype != np.float32 and v.dtype != np.float64:
        v = v.astype(np.float64)
    with np.errstate(all="ignore"):
        n2 = np.einsum("ij,ij->i", v, v, dtype=np.float64)        # float64 accumulation without a float64 copy of the block
        bad = ~np.isfinite(n2) | (n2 < 1e-30) | (np.abs(v).max(axis=1, initial=0.0) > 1e18)
    return bad


class _AllBut:
    """Passing-row set of a WHERE clause that only excludes a few rows (soft deletes): `len()` / indexing as a sorted
    array would need every live id; hybrid_search only asks membership."""

    def __init__(self, denied_sorted: np.ndarray) -> None:
        self.denied = denied_sorted

    def __contains__(self, rid: int) -> bool:
        j = int(np.searchsorted(self.denied, rid))
        return not (j < len(self.denied) and int(self.denied[j]) == rid)


_collections: Dict[Tuple[str, str], _Collection] = {}
_collections_lock = threading.Lock()


class PlanMismatch(ValueError):
    """A sharded upsert was handed vectors embedded for other row ids than the table now assigns (another writer inserted
    between the embedding and the upsert): nothing was stored, the caller embeds again for the new ids."""


def _default_index_factory(dim: int, capacity: int, dtype: str, metric: str, shards: int = 1):
    """shards == 1: one HipIndex on this process's GPU. shards > 1 (pg_config["hip"]["shards"]): this process is one rank
    of a torch.distributed job of that size (one process per GPU) and owns one row shard; every rank makes the same
    store calls and every search is the all-gather + merge of archi_amd/sharded.py."""
    if shards and int(shards) > 1:
        from .sharded import ShardedHipIndex
        return ShardedHipIndex(dim, capacity, dtype=dtype, metric=metric, shards=int(shards))
    from .index import HipIndex  # raises HipBackendError without libarchi_hip.so / a gfx950 GPU
    return HipIndex(dim, capacity, dtype=dtype, metric=metric)


def reset_collections() -> None:
    """Drop every cached collection (tests / `reset_collection: true`, manager.py:103-153)."""
    with _collections_lock:
        for c in _collections.values():
            close = getattr(c.index, "close", None)
            if close:
                close()
        _collections.clear()


class ArchiHipVectorStore(_VectorStoreBase):
    """LangChain-compatible vector store on one MI355X."""

    def __init__(
        self,
        pg_config: Optional[Dict[str, Any]],
        embedding_function: Any,
        collection_name: str = "default",
        distance_metric: str = "cosine",
        *,
        connection: Any = None,
        index_factory: Optional[Callable[..., Any]] = None,
    ):
        """Same signature as the reference (:47-56). `pg_config` is accepted for call-site compatibility; its optional
        "hip" entry tunes the GPU index: {"dtype": "f32"|"bf16"|"f16", "capacity": rows}. `connection` is ignored.

        dtype "f32" (default) stores exactly what the reference's `vector(D)` column stores (float32,
        src/cli/templates/init.sql:266): ids and scores equal the reference CPU path's on the same inputs (the MFMA
        scan reads a bf16 shadow, results come from the exact re-rank on the float32 rows). "bf16" / "f16" are opt-in
        and LOSSY relative to vector(D): half the HBM, scores off by ~1e-3 from the float32 path.
        capacity is only the first reservation: the index grows and reclaims deleted rows by itself."""
        self._pg_config = pg_config or {}
        self._embedding_function = embedding_function
        self._collection_name = collection_name
        self._distance_metric = distance_metric
        self._external_connection = connection
        self._distance_ops = dict(DISTANCE_OPS)
        if distance_metric not in self._distance_ops:
            raise ValueError(f"distance_metric must be one of {list(self._distance_ops.keys())}")
        self._distance_op = self._distance_ops[distance_metric]
        hip_cfg = dict(self._pg_config.get("hip", {}) or {})
        self._dtype = hip_cfg.get("dtype", "f32")
        self._capacity = int(hip_cfg.get("capacity", 1 << 16))
        self._shards = int(hip_cfg.get("shards", 1) or 1)
        self._index_factory = index_factory or _default_index_factory

    # -- plumbing ---------------------------------------------------------
    @property
    def embeddings(self) -> Optional[Any]:
        return self._embedding_function

    def _collection(self, dim: Optional[int] = None) -> Optional[_Collection]:
        key = (self._collection_name, self._distance_metric)
        with _collections_lock:
            col = _collections.get(key)
            if col is None and dim is not None:
                if self._shards > 1 and self._index_factory is _default_index_factory:
                    index = self._index_factory(dim, self._capacity, self._dtype, self._distance_metric, self._shards)
                else:
                    index = self._index_factory(dim, self._capacity, self._dtype, self._distance_metric)
                log.info("collection %r (%s): new index, %d-d %s, first reservation %d rows, %d shard(s)",
                         self._collection_name, self._distance_metric, dim, self._dtype, self._capacity, self._shards)
                col = _Collection(index, ChunkTable())
                _collections[key] = col
            return col

    @property
    def table(self) -> Optional[ChunkTable]:
        col = self._collection()
        return col.table if col else None

    # -- writes -----------------------------------------------------------
    def add_texts(
        self,
        texts: Iterable[str],
        metadatas: Optional[List[Dict[str, Any]]] = None,
        *,
        ids: Optional[List[str]] = None,
        **kwargs: Any,
    ) -> List[str]:
        """Embed and upsert (reference :105-186). Returns the chunk ids."""
        texts_list = list(texts)
        if not texts_list:
            return []
        if ids is None:
            ids = [str(uuid.uuid4()) for _ in texts_list]
        if metadatas is None:
            metadatas = [{} for _ in texts_list]
        for meta in metadatas:
            meta["collection"] = self._collection_name
        # `embeddings=` (build extension, rides in **kwargs): vectors already computed by a cross-file
        # batched embed call (archi_amd.ingest.BatchedIngestor); otherwise embed here like the reference (:143)
        embeddings = kwargs.get("embeddings")
        document_id = kwargs.get("document_id")
        if embeddings is not None:
            # one transaction like the reference's upsert (:168-182): nothing of a failed call stays behind
            self._upsert([(texts_list, metadatas, document_id, embeddings, ids)], plan=None)
            return ids
        # row-sharded index: this rank embeds only the chunks whose row will live on its shard (embed_for_rows); a failure
        # of one rank's share fails the call on every rank (agree_embedded), like the single embed call it stands for. The
        # row ids are read before the embedding and checked under the table lock by the upsert: when another writer got in
        # between (PlanMismatch, nothing stored) the chunks are embedded again for the new ids, a bounded number of times --
        # BatchedIngestor does the same around add_texts_batch.
        for attempt in range(3):
            err: Optional[BaseException] = None
            mine = rid0 = None
            try:
                embeddings, mine, rid0 = self.embed_for_rows(texts_list)
            except Exception as exc:                 # noqa: BLE001 -- re-raised by agree_embedded, on EVERY rank
                err = exc
            self.agree_embedded(err)
            plan = (rid0, mine) if mine is not None else None
            try:
                self._upsert([(texts_list, metadatas, document_id, embeddings, ids)], plan=plan)
                break
            except PlanMismatch:
                if attempt == 2:
                    raise
        return ids

    # -- data-parallel embedding behind a row-sharded index (replicate the weights, shard the chunk batch, no
    # collective). Every rank makes the same store calls (SPMD) and assigns the same row ids -- the table's SERIAL key --, and a
    # row's shard is id % world: so before anything is embedded every rank knows which of the new chunks will be ITS rows,
    # embeds only those (1 / world of the encoder work; the reference's loop embeds every chunk of every file in one
    # process, manager.py:362-373) and hands the index a block whose other rows are placeholders the sharded index drops.
    def shard_layout(self) -> Tuple[int, int]:
        """(world, rank) of the row-sharded index behind this collection, (1, 0) for a single index / no collection yet."""
        col = self._collection()
        if col is None:
            return 1, 0
        return int(getattr(col.index, "world", 1)), int(getattr(col.index, "rank", 0))

    def agree_embedded(self, error: Optional[BaseException]) -> None:
        """SPMD agreement after a data-parallel embedding step: raises on EVERY rank when the step failed on ANY rank (the
        rank's own exception where there is one), returns on all of them otherwise. The reference's loop marks a file failed
        when its embed call raises and carries on with the next (manager.py:374-389); with the chunks of a file spread over
        the ranks only the owner of a bad chunk sees the exception, and a rank that left the common call sequence alone would
        leave the others waiting in their next collective. One int32 all-reduce -- a verdict, never a vector (embedding
        itself stays collective-free). No-op on a single index."""
        world, _ = self.shard_layout()
        col = self._collection()
        if world == 1 and self._shards > 1 and col is None:
            # first call of a collection on a row-sharded store: no index (hence no shard_layout) yet, but the ranks are there --
            # an embed failure on one of them must still fail the call on all (round-4 advisor)
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                import torch
                flag = torch.tensor([1 if error is not None else 0], dtype=torch.int32)
                if dist.get_backend() == "nccl":
                    flag = flag.cuda()
                dist.all_reduce(flag)
                if error is not None:
                    raise error
                if int(flag.item()):
                    raise RuntimeError("embedding failed on another shard's share of the batch")
                return
        if world == 1:
            if error is not None:
                raise error
            return
        failed = bool(col.index.reduce_flags(np.array([error is not None]))[0])
        if error is not None:
            raise error
        if failed:
            raise RuntimeError("embedding failed on another shard's share of the batch")

    def next_row_id(self) -> Optional[int]:
        """The row id the next inserted chunk gets (document_chunks.id is SERIAL), None before the collection exists."""
        col = self._collection()
        return None if col is None else int(col.table.next_id)

    def embed_for_rows(self, texts: List[str], rid0: Optional[int] = None):
        """Embeddings for chunks that will become rows rid0, rid0 + 1, ... (rid0=None: the table's next id, right for a caller
        that inserts before anyone else does). Returns (vectors [n, D] float32, mine, rid0): on a single index mine is None and
        every row is real; on a row-sharded index only rows with mine[i] (= this rank's shard) are embedded, the others are
        zero placeholders. The first call of a collection (no index yet, its width unknown) embeds everything."""
        fn = self._embedding_function
        has_array = callable(getattr(type(fn), "embed_documents_array", None))       # class-level: mocks do not qualify
        embed = fn.embed_documents_array if has_array else fn.embed_documents
        world, rank = self.shard_layout()
        col = self._collection()
        if world == 1 or col is None:
            return embed(texts), None, rid0
        if rid0 is None:
            rid0 = int(col.table.next_id)
        n = len(texts)
        mine = ((rid0 + np.arange(n, dtype=np.int64)) % world) == rank
        vecs = np.zeros((n, int(col.index.dim)), dtype=np.float32)
        own = np.flatnonzero(mine)
        if len(own):
            v = np.asarray(embed([texts[i] for i in own.tolist()]), dtype=np.float32)
            if v.shape != (len(own), vecs.shape[1]):
                raise ValueError("embed_documents must return one vector per text")
            vecs[own] = v
        return vecs, mine, rid0

    def add_texts_batch(self, items: List[Tuple[List[str], List[Dict[str, Any]], Any, Any]],
                        plan: Optional[Tuple[int, np.ndarray]] = None) -> List[List[str]]:
        """Build extension for the ingestion harness: several `add_texts(texts, metadatas, document_id=..., embeddings=...)`
        calls -- one item per file -- with ONE index update for all of them. Row by row it does what add_texts does
        (uuid4 chunk ids, `collection` / `chunk_id` written into the caller's metadata dicts, ON CONFLICT replacement per
        (document_id, chunk_index)); per-file `ak_index_add` calls each synchronise with the GPU, which is busy embedding
        the next group at that moment (0.3 ms per file, two thirds of the ingestion time).
        plan = (rid0, mine) of embed_for_rows over the concatenated texts of all items: the vectors were embedded for rows
        rid0, rid0 + 1, ... with only mine[i] real; if the table hands out other ids (another writer got in between the
        embedding and this call) the batch is refused -- ValueError, nothing stored -- and the caller embeds again."""
        if not items:
            return []
        return self._upsert([(t, m, d, v, None) for t, m, d, v in items], plan=plan)

    def _upsert(self, items: List[Tuple[List[str], Optional[List[Dict[str, Any]]], Any, Any, Optional[List[str]]]],
                plan: Optional[Tuple[int, np.ndarray]] = None) -> List[List[str]]:
        """INSERT ... ON CONFLICT (document_id, chunk_index) DO UPDATE for every item (texts, metadatas, document_id,
        vectors, chunk ids or None) as ONE transaction: widths are checked before anything is touched, the new vectors go
        into the index first, the replaced rows leave only after that add succeeded, and any exception undoes the table
        rows and the (document_id, chunk_index) map. The reference runs its upsert inside one database transaction
        (postgres_vectorstore.py:168-182). plan: see add_texts_batch."""
        blocks_in = []
        for texts, metadatas, document_id, vectors, ids in items:
            texts = list(texts)
            vecs = np.asarray(vectors, dtype=np.float32)
            if vecs.ndim != 2 or vecs.shape[0] != len(texts):
                raise ValueError("embed_documents must return one vector per text")
            blocks_in.append((texts, metadatas, document_id, vecs, ids))
        dims = {b[3].shape[1] for b in blocks_in if len(b[0])}
        if len(dims) > 1:
            raise ValueError("add_texts_batch: every item needs a [n, D] embedding block of one width")
        if not dims:
            return [[] for _ in blocks_in]
        dim = dims.pop()
        col = self._collection(dim)
        have = getattr(col.index, "dim", dim)
        if have != dim:
            raise ValueError(f"collection {self._collection_name!r} holds {have}-d vectors, got {dim}-d")
        t = col.table
        out: List[List[str]] = []
        blocks, all_rows, stale, suspects = [], [], [], []
        added = False
        with t.lock:
          # checked before a row id is spent: the caller embeds again for the ids the table really hands out. A plan exists only
          # behind a row-sharded index, and there the verdict is AGREED over the ranks (one flag all-reduce): a rank that alone
          # saw another writer get in between would otherwise loop back into embed_for_rows / agree_embedded while the others
          # went on to reduce_flags and the index add -- mis-paired collectives (round-5 advisor finding)
          if plan is not None and bool(col.index.reduce_flags(np.array([int(plan[0]) != int(t.next_id)]))[0]):
              raise PlanMismatch(f"sharded upsert: the vectors were embedded for rows {int(plan[0])}.. but the table's next row id is "
                                 f"{int(t.next_id)} (here or on another rank): embed again")
          try:
            for texts, metadatas, document_id, vecs, ids in blocks_in:
                ids = _uuid4_many(len(texts)) if ids is None else list(ids)
                metadatas = metadatas if metadatas is not None else [{} for _ in texts]
                if len(metadatas) != len(texts) or len(ids) != len(texts):
                    raise ValueError("add_texts: texts, metadatas and ids must have one entry per text")
                for metadata, chunk_id in zip(metadatas, ids):
                    metadata["collection"] = self._collection_name
                    metadata["chunk_id"] = chunk_id
                if t.has_document(document_id):                       # ON CONFLICT (document_id, chunk_index) can only fire
                    for i in range(len(texts)):                       # for a document that already has rows
                        prev = t.find(document_id, i)
                        if prev is not None:
                            stale.append(prev)
                rid0 = t.append_block(document_id, texts, metadatas)  # next_id moves with it
                all_rows.extend(range(rid0, rid0 + len(texts)))
                blocks.append(vecs)
                out.append(ids)
            if all_rows:
                rows = np.concatenate(blocks)
                bad = _suspect_rows(rows)                                                       # one pass for the whole batch
                if plan is not None:
                    rid0, mine = plan
                    if len(mine) != len(all_rows) or all_rows[0] != rid0 or all_rows[-1] != rid0 + len(all_rows) - 1:
                        raise PlanMismatch(f"sharded upsert: the vectors were embedded for rows {rid0}.. but the table assigns "
                                           f"{all_rows[0]}..{all_rows[-1]}: embed again")
                    # only this rank's rows are real; every rank needs the same suspect list (hybrid_search asks all shards
                    # for their distances): one small all-reduce of the flags
                    bad = col.index.reduce_flags(bad & np.asarray(mine, bool))
                suspects = [all_rows[int(i)] for i in np.nonzero(bad)[0]]
                col.index.add(rows, ids=all_rows)
                added = True
            if stale:                               # only once the new rows are in: a failed batch leaves the old ones
                stale = list(dict.fromkeys(stale))
                col.index.remove(stale)
                for rid in stale:
                    t.kill(rid)
                    t.suspects.discard(rid)
            t.suspects.update(suspects)
            t.version += 1
            if t.dead_fraction() > 0.5 and t.positions > 4096:
                t.vacuum()
          except Exception:
            for rid in reversed(all_rows):          # nothing of a failed batch stays behind: the rows it replaced are still
                t.kill(rid)                         # live, so find() answers with them again
            if added:                               # the vectors went in before the failure: take them out again (best effort)
                try:
                    col.index.remove(all_rows)
                except Exception:
                    log.exception("upsert rollback: %d orphan vectors left in the index of %r", len(all_rows), self._collection_name)
            t.version += 1                          # text-index / WHERE caches keyed on the version may have seen the rows
            raise
        return out

    def add_documents(self, documents: List[Any], **kwargs: Any) -> List[str]:
        texts = [doc.page_content for doc in documents]
        metadatas = [doc.metadata for doc in documents]
        return self.add_texts(texts, metadatas=metadatas, **kwargs)

    def delete(self, ids: Optional[List[str]] = None, **kwargs: Any) -> Optional[bool]:
        """Reference :493-535: False when neither ids nor document_id is given, else True."""
        document_id = kwargs.get("document_id")
        if ids is None and document_id is None:
            return False
        col = self._collection()
        if col is None:
            return True
        t = col.table
        with t.lock:
            victims = t.rids_of_document(document_id) if document_id is not None else t.rids_of_chunk_ids(ids or [])
            self._delete_rows(col, victims)
        return True

    @staticmethod
    def _delete_rows(col: "_Collection", victims: List[int]) -> int:
        """DELETE the listed rows from the index and the table (caller holds the table lock)."""
        if not victims:
            return 0
        t = col.table
        col.index.remove(victims)
        for rid in victims:
            t.suspects.discard(rid)
            t.kill(rid)
        t.version += 1
        if t.dead_fraction() > 0.5 and t.positions > 4096:
            t.vacuum()
        return len(victims)

    # -- the sync step of the data manager (manager.py:177-252) ----------------------------------
    def resource_hashes(self) -> set:
        """SELECT DISTINCT metadata->>'resource_hash' FROM document_chunks WHERE (collection = this OR NULL) AND it IS NOT
        NULL -- `_collect_postgres_hashes` (manager.py:216-232)."""
        col = self._collection()
        if col is None:
            return set()
        with col.table.lock:
            return col.table.distinct_values("resource_hash")

    def delete_resource_hashes(self, hashes: Iterable[str]) -> int:
        """DELETE FROM document_chunks WHERE metadata->>'resource_hash' = %s AND (collection ...) for every hash --
        `_remove_from_postgres` (manager.py:234-252). Returns the number of chunks removed."""
        col = self._collection()
        if col is None:
            return 0
        t = col.table
        total = 0
        with t.lock:
            for h in hashes:
                total += self._delete_rows(col, t.rids_at(t.positions_matching({"resource_hash": h})).tolist())
        return total

    def sync(self, files_in_data: Dict[str, Any], add: Callable[[Dict[str, Any]], Any]) -> Dict[str, Any]:
        """`update_vectorstore` (manager.py:177-214): hashes in the store vs hashes in the catalog; stale hashes are removed,
        missing ones handed to `add` ({hash: path-or-payload}; the ingestion harness), which may raise without undoing the
        removals -- like the reference, which logs the error and carries on (:208-211). Returns what was done."""
        in_store = self.resource_hashes()
        in_data = set(files_in_data.keys())
        report = {"in_store": len(in_store), "in_data": len(in_data), "removed": [], "added": [], "error": None}
        if in_data == in_store:
            log.info("Vectorstore is up to date")
            return report
        stale = sorted(in_store - in_data)
        if stale:
            log.info("Removing %d stale documents", len(stale))
            self.delete_resource_hashes(stale)
            report["removed"] = stale
        missing = {h: files_in_data[h] for h in sorted(in_data - in_store)}
        if missing:
            log.info("Adding %d new documents", len(missing))
            try:
                add(missing)
                report["added"] = list(missing)
            except Exception as exc:                 # manager.py:208-211
                log.error("Files could not be added", exc_info=exc)
                report["error"] = str(exc)
        return report

    # -- reads ------------------------------------------------------------
    def similarity_search(self, query: str, k: int = 4, **kwargs: Any) -> List[Any]:
        return [doc for doc, _ in self.similarity_search_with_score(query, k=k, **kwargs)]

    def similarity_search_with_score(self, query: str, k: int = 4, **kwargs: Any) -> List[Tuple[Any, float]]:
        query_embedding = self._embedding_function.embed_query(query)
        return self.similarity_search_by_vector_with_score(query_embedding, k=k, **kwargs)

    def similarity_search_by_vector(self, embedding: List[float], k: int = 4, **kwargs: Any) -> List[Any]:
        return [doc for doc, _ in self.similarity_search_by_vector_with_score(embedding, k=k, **kwargs)]

    def similarity_search_by_vector_with_score(
        self, embedding: List[float], k: int = 4, **kwargs: Any
    ) -> List[Tuple[Any, float]]:
        """Reference :272-364: filter -> distance -> ORDER BY distance ASC LIMIT k -> score."""
        metadata_filter = kwargs.get("filter", {}) or {}
        include_deleted = kwargs.get("include_deleted", False)
        col = self._collection()
        if col is None or k <= 0:
            return []
        t = col.table
        # the value pgvector would see: python float -> text -> float4 (a4, :313)
        q = np.asarray([float(x) for x in embedding], dtype=np.float32)
        # the WHERE clause is resolved under the table lock, the scan runs WITHOUT it: request threads search concurrently (the
        # library coalesces concurrent single-query calls into one launch) and a writer is never kept waiting behind a GPU call
        for attempt in range(self.STALE_RETRIES + 1):
            ids, dist, cnt = self._search_snapshot(col, q, k, lambda: self._where(col, metadata_filter, include_deleted)[::2])
            results: List[Tuple[Any, float]] = []
            vanished = 0
            with t.lock:
                for j in range(int(cnt[0])):
                    p = t.pos(int(ids[0, j]))
                    if p < 0:                        # deleted between the scan and here
                        vanished += 1
                        continue
                    distance = float(dist[0, j])
                    score = 1.0 - distance if self._distance_metric == "cosine" else distance   # :361
                    results.append((self._document(t, p), score))
            if not vanished:
                break
            # a writer deleted (or replaced: re-ingestion is delete + add) rows of the answer before their text could be read: the
            # statement's snapshot would still have held them, and dropping them would return fewer than k -- search again, against
            # the later state (after the retries the shortened list stands: every row in it is live and in order)
        return results

    STALE_RETRIES = 2

    def _search_snapshot(self, col: _Collection, q: np.ndarray, k: int, build_mask: Callable[[], Tuple[Any, int]]):
        """One filtered top-k with the snapshot semantics of the reference's single SQL statement (WHERE, distance, ORDER BY and
        LIMIT see one state of the table: postgres_vectorstore.py:296-332). build_mask() -> (per-slot byte mask or None, layout
        epoch) runs under the table lock; the scan runs outside it and hands the epoch back to the library, which refuses the
        mask (StaleFilterError, nothing read) if a writer has added rows or reclaimed tombstones in between -- a stale mask
        would address other rows: soft-deleted or filtered-out chunks could come back. Then the mask is rebuilt; after
        STALE_RETRIES collisions the scan runs under the table lock, which every writer of the index takes."""
        t = col.table
        for _ in range(self.STALE_RETRIES):
            with t.lock:
                mask, epoch = build_mask()
            if mask is None:
                return col.index.search(q[None, :], k)
            try:
                return col.index.search(q[None, :], k, row_filter=mask, filter_epoch=epoch)
            except StaleFilterError:
                continue
        with t.lock:
            mask, epoch = build_mask()
            if mask is None:
                return col.index.search(q[None, :], k)
            return col.index.search(q[None, :], k, row_filter=mask, filter_epoch=epoch)

    def _where(self, col: _Collection, metadata_filter: Dict[str, Any], include_deleted: bool):
        """The WHERE clause (:296-310) as a per-slot byte mask for the scan, or None when every row passes; the passing row ids
        as a sorted array (None = all); and the index layout epoch the mask was built for (HipIndex.layout()). Caller holds
        the table lock -- every writer of the index holds it too, so slots, lookups and epoch are one state. The collection term (:296) is
        true for every row of this table by construction (one table per collection; rows loaded from a dump are filtered
        on load). Metadata terms come from the table's inverted maps, soft deletes from the per-document row lists: no
        pass over the rows."""
        t = col.table
        n_slots, epoch = col.index.layout()
        try:
            key = (t.version, t.doc_version, epoch, bool(include_deleted), json.dumps(metadata_filter, sort_keys=True, default=str))
        except (TypeError, ValueError):
            key = None
        if key is not None and key in t.where_cache:
            return t.where_cache[key]
        deleted_docs = [] if include_deleted else [d for d, c in t.documents.items() if c.get("is_deleted", False)]
        if not metadata_filter and not deleted_docs:
            result = (None, None, epoch)
        elif not metadata_filter:
            # only soft deletes: everything passes except the rows of the deleted documents
            gone = t.positions_of_documents(deleted_docs)
            if not len(gone):
                result = (None, None, epoch)
            else:
                row_filter = np.ones(n_slots, dtype=np.uint8)
                slots = col.index.lookup(t.rids_at(gone))
                row_filter[slots[slots >= 0]] = 0
                result = (row_filter, _AllBut(np.sort(t.rids_at(gone))), epoch)      # "every row except these": no pass over the rows
        else:
            pos = t.positions_matching(metadata_filter)
            if deleted_docs and len(pos):
                pos = np.setdiff1d(pos, t.positions_of_documents(deleted_docs))
            live = np.sort(t.rids_at(pos)) if len(pos) else np.zeros(0, np.int64)
            row_filter = np.zeros(n_slots, dtype=np.uint8)
            if len(live):
                slots = col.index.lookup(live)
                row_filter[slots[slots >= 0]] = 1
            result = (row_filter, live, epoch)
        if key is not None:
            for old in [k for k in t.where_cache if k[:3] != key[:3]]:
                del t.where_cache[old]
            if len(t.where_cache) >= 16:
                t.where_cache.pop(next(iter(t.where_cache)))
            t.where_cache[key] = result
        return result

    @staticmethod
    def _document(t: ChunkTable, p: int) -> Any:
        metadata = t.metadata_at(p) or {}
        doc_id = t.document_id_at(p)
        d = t.documents.get(doc_id) if doc_id is not None else None
        if d:                                                      # :347-354
            for col_name in ("resource_hash", "display_name", "source_type", "url"):
                if d.get(col_name):
                    metadata[col_name] = d[col_name]
        return Document(page_content=t.text_at(p), metadata=metadata)

    # -- N2: an existing deployment's table, loaded without re-embedding --------------------------
    def _own_predicate(self):
        if self._shards > 1:
            # row-sharded store: every rank reads the same stream (SPMD) but decodes only the vectors of the rows it will hold
            import torch.distributed as dist
            world, rank = self._shards, dist.get_rank()
            return lambda rid: rid % world == rank           # noqa: E731
        return None

    def _append_copy_block(self, blk: dict, known: str, stats: Optional[dict] = None) -> int:
        """One decoded block of six-column COPY tuples (pgbridge.iter_pgcopy_chunks) into table AND index, as one unit: the
        width is checked before the table is touched, a failed index add takes the block's rows out of the table again.
        known: what to do with a row id the collection already holds -- "raise" (a first load: the stream and the collection
        must be disjoint) or "update" (refresh: the row was rewritten in place by another process -- its columns are updated
        and its vector replaced under the same id). Returns the number of rows appended or updated."""
        keep = [i for i, md in enumerate(blk["metadata"])
                if (md or {}).get("collection") in (None, self._collection_name)]
        if not keep:
            return 0
        vecs = blk["vectors"][keep]
        col = self._collection(vecs.shape[1])
        have = getattr(col.index, "dim", vecs.shape[1])
        if have != vecs.shape[1]:           # checked before the table is touched: a block either enters whole or not at all
            raise ValueError(f"load_from_pgcopy: the stream holds {vecs.shape[1]}-d vectors, collection "
                             f"{self._collection_name!r} holds {have}-d ones")
        t = col.table
        with t.lock:
            kept = np.asarray(keep, np.int64)
            order = np.argsort(blk["ids"][kept], kind="stable")
            sel = kept[order]
            rids = blk["ids"][sel]
            vecs = vecs[order]
            own = blk["own"][sel] if blk.get("own") is not None else None
            if len(rids) > 1 and not (np.diff(rids) > 0).all():
                dup = int(rids[1:][np.diff(rids) == 0][0])
                raise ValueError(f"load_from_pgcopy: row id {dup} appears twice in the stream")
            there = t.pos_many(rids) >= 0
            if there.any() and known == "raise":
                raise ValueError(f"load_from_pgcopy: row id {int(rids[there][0])} is already in collection {self._collection_name!r}")
            pick = sel.tolist()
            new = ~there
            if there.any():
                # rows another process UPDATEd in place (the reference's ON CONFLICT (document_id, chunk_index) DO UPDATE keeps
                # the row id, postgres_vectorstore.py:168-180): same id, new text / metadata / vector. The index holds one live
                # row per id, so the old vector leaves before the new one enters -- a reader between the two sees neither,
                # where PostgreSQL's snapshot would show the old one; both steps run under the table lock.
                upd = np.flatnonzero(there)
                col.index.remove(rids[upd])
                try:
                    col.index.add(vecs[upd], ids=rids[upd].tolist())
                except Exception:
                    for rid in rids[upd].tolist():          # the old vectors are gone and the new ones did not enter: the rows
                        t.suspects.discard(int(rid))        # leave the table too (the next refresh lists them as missing)
                        t.kill(int(rid))
                    t.version += 1
                    raise
                for j in upd.tolist():
                    i = pick[j]
                    t.update_row(int(rids[j]), document_id=blk["document_ids"][i], chunk_index=int(blk["chunk_index"][i]),
                                 text=blk["text_bytes"][i].decode("utf-8", "surrogatepass"), metadata=blk["metadata"][i])
                if stats is not None:
                    stats["updated"] = stats.get("updated", 0) + len(upd)
            if new.any():
                nsel = sel[new]
                npick = nsel.tolist()
                nrids = rids[new]
                t.append_rows(nrids, [blk["document_ids"][i] for i in npick], blk["chunk_index"][nsel], [blk["text_bytes"][i] for i in npick],
                              [blk["metadata"][i] for i in npick], [blk["meta_json"][i] for i in npick])
                try:
                    col.index.add(vecs[new], ids=nrids.tolist())
                except Exception:
                    # the vectors did not go in (capacity, out of memory, a duplicate the table did not know): the block's rows
                    # leave the table again, as a failed upsert's do -- count() == len(table), a retry starts from a clean state
                    for rid in reversed(nrids.tolist()):
                        t.kill(int(rid))
                    t.version += 1
                    raise
                if stats is not None:
                    stats["added"] = stats.get("added", 0) + int(new.sum())
            bad = _suspect_rows(vecs)
            if own is not None:          # foreign rows are zero placeholders here: the flags of all shards, OR-ed
                bad = col.index.reduce_flags(bad & own)
            for r in rids[there & ~bad].tolist():
                t.suspects.discard(int(r))
            t.suspects.update(int(r) for r in rids[bad])
            t.version += 1
            return int(len(rids))

    def load_from_pgcopy(self, chunks_stream: Any, documents_stream: Any = None, batch: int = 65536,
                         versions_stream: Any = None) -> int:
        """Fill this collection -- table AND index -- from PostgreSQL binary COPY streams (archi_amd/pgbridge.py):
        `COPY (SELECT id, document_id, chunk_index, chunk_text, metadata, embedding FROM document_chunks ...)` and,
        optionally, `COPY (SELECT id, resource_hash, display_name, source_type, url, is_deleted FROM documents)`.
        Row ids stay `document_chunks.id`; rows of other collections (metadata->>'collection' set and different) and rows
        without an embedding are skipped. versions_stream: optional `COPY (SELECT id, xmin::text::bigint ...)` of the SAME
        snapshot -- the row versions a later refresh_from_pgcopy compares (without it the first refresh re-reads every row it
        cannot vouch for). Returns the number of chunks loaded."""
        from . import pgbridge
        if documents_stream is not None:
            docs = pgbridge.read_pgcopy_documents(documents_stream)
        else:
            docs = []
        total = 0
        for blk in pgbridge.iter_pgcopy_chunks(chunks_stream, batch, own=self._own_predicate()):
            total += self._append_copy_block(blk, known="raise")
        col = self._collection()
        if col is not None and docs:
            for d in docs:
                col.table.register_document(d.pop("id"), **d)
        if col is not None and versions_stream is not None:
            vid, vver = pgbridge.read_pgcopy_ids(versions_stream)
            if vver is not None:
                with col.table.lock:
                    col.table.set_versions(vid, vver)
        log.info("collection %r: %d chunks loaded from a COPY stream", self._collection_name, total)
        return total

    def max_row_id(self) -> int:
        """Largest `document_chunks.id` this collection has ever held (0 when empty): the `%s` of the tail refresh's
        `WHERE id > %s`."""
        col = self._collection()
        return 0 if col is None else col.table.max_rid()

    def refresh_from_pgcopy(self, ids_stream: Any, fetch_rows: Optional[Callable[[np.ndarray], Any]] = None,
                            documents_stream: Any = None, batch: int = 65536) -> Dict[str, int]:
        """RECONCILE this live collection with the table another process writes (SURVEY 8f N2 "load/refresh"; round-5 review,
        missing #1). The reference needs no such call: every chat request builds a fresh PostgresVectorStore over the one
        table (src/archi/archi.py:61-65 -> src/archi/utils/vectorstore_connector.py:60-81) while the data-manager process
        adds, replaces and deletes rows in it (src/data_manager/vectorstore/manager.py:177-214). Here the index lives in this
        process's HBM, so a chat process calls this to see what the ingestion process did.

          ids_stream        COPY (SELECT id [, xmin::text::bigint] FROM document_chunks
                                  WHERE embedding IS NOT NULL AND (metadata->>'collection' = %s OR metadata->>'collection' IS NULL))
                            TO STDOUT (FORMAT binary)         -- the whole collection, 12-24 bytes per row
          fetch_rows(ids)   -> the six-column stream of load_from_pgcopy for `WHERE id = ANY(%s)`: called once, with the ids the
                            collection lacks plus the ids whose version differs (or is unknown here). None is allowed when
                            nothing needs fetching; otherwise it is an error.
          documents_stream  the `documents` columns, as for load_from_pgcopy: soft deletes, renames and removals are applied.

        Under the table lock, in this order: rows whose id left the table are deleted (index + table, as delete() does);
        rewritten rows are replaced under their id; new rows are appended; the `documents` mirror is brought up to date. The
        layout epoch, the table version and the WHERE-mask cache move exactly as for the add / delete calls they are made of --
        a concurrent filtered search either completes on the old state or is told its mask is stale and rebuilds it
        (AK_ERR_STALE_FILTER), so it never returns a row that was deleted or soft-deleted before it started. Nothing moves
        when nothing changed (idempotent: no epoch change, no cache invalidation).
        Take the three streams in ONE repeatable-read transaction; if they are not, rows that appear or vanish between them
        are picked up by the next refresh (a requested row the rows stream does not hold is skipped, never invented).
        Returns {"removed", "added", "updated", "documents_changed", "fetched"}."""
        from . import pgbridge
        tid, tver = pgbridge.read_pgcopy_ids(ids_stream)
        order = np.argsort(tid, kind="stable")
        tid = tid[order]
        if tver is not None:
            tver = tver[order]
        if len(tid) > 1 and not (np.diff(tid) > 0).all():
            raise ValueError("refresh_from_pgcopy: a row id appears twice in the id stream")
        stats = {"removed": 0, "added": 0, "updated": 0, "documents_changed": 0, "fetched": 0}
        docs = pgbridge.read_pgcopy_documents(documents_stream) if documents_stream is not None else None
        col = self._collection()
        t = col.table if col is not None else None
        # 1. the diff and the deletes, under the table lock (every writer of this collection holds it; the chat process that
        #    refreshes has no other writer). The lock is NOT held while the missing rows travel from the database: searches
        #    keep running on the state "deleted rows gone, new rows not there yet", each block then enters under the lock.
        with (t.lock if t is not None else threading.RLock()):
            live = np.sort(t.live_rids()) if t is not None else np.zeros(0, np.int64)
            gone = np.setdiff1d(live, tid, assume_unique=True)
            if len(gone):
                stats["removed"] = self._delete_rows(col, [int(r) for r in gone])
            need = np.setdiff1d(tid, live, assume_unique=True)
            if tver is not None and t is not None and len(live):
                both = np.intersect1d(tid, live, assume_unique=True)
                if len(both):
                    want = tver[np.searchsorted(tid, both)]
                    have = t.versions_of(both)
                    changed = both[(have != want) | (have == 0)]
                    need = np.union1d(need, changed)
        # 2. new and rewritten rows
        got: List[np.ndarray] = []
        if len(need):
            if fetch_rows is None:
                raise ValueError(f"refresh_from_pgcopy: {len(need)} rows are new or rewritten and no fetch_rows callback was given")
            stream = fetch_rows(need.copy())
            for blk in pgbridge.iter_pgcopy_chunks(stream, batch, own=self._own_predicate()):
                if not np.isin(blk["ids"], need, assume_unique=False).all():
                    raise ValueError("refresh_from_pgcopy: fetch_rows returned a row that was not asked for")
                stats["fetched"] += self._append_copy_block(blk, known="update", stats=stats)
                got.append(blk["ids"])
            col = self._collection()
            t = col.table if col is not None else None
        # 3. versions (only of rows this collection now really holds in that version) and the `documents` mirror
        if t is not None:
            with t.lock:
                if tver is not None:
                    fetched = np.concatenate(got) if got else np.zeros(0, np.int64)
                    settled = np.isin(tid, np.setdiff1d(need, fetched), invert=True)    # asked for and not delivered: version stays unknown
                    t.set_versions(tid[settled], tver[settled])
                if docs is not None:
                    seen = set()
                    changed_docs = 0
                    for d in docs:
                        d = dict(d)
                        did = d.pop("id")
                        seen.add(did)
                        cur = t.documents.get(did)
                        if cur is None or any(cur.get(k) != v for k, v in d.items()):
                            t.register_document(did, **d)
                            changed_docs += 1
                    for did in [k for k in t.documents if k not in seen]:
                        del t.documents[did]              # the `documents` row is gone (its chunks went with it: ON DELETE CASCADE)
                        t.doc_version += 1
                        changed_docs += 1
                    stats["documents_changed"] = changed_docs
        log.info("collection %r refreshed from the table: %s", self._collection_name, stats)
        return stats

    def refresh_tail_from_pgcopy(self, rows_stream: Any, documents_stream: Any = None, batch: int = 65536) -> int:
        """The append-only fast path of refresh_from_pgcopy: the six-column stream of `... WHERE id > %s` with %s =
        max_row_id() -- what a deployment that only ever ADDS documents needs between two full reconciliations (SERIAL ids
        only grow: every row inserted since the last call is in that range; deletes and in-place rewrites are NOT seen).
        Rows already present are an error, as for load_from_pgcopy. Returns the number of rows appended."""
        from . import pgbridge
        total = 0
        for blk in pgbridge.iter_pgcopy_chunks(rows_stream, batch, own=self._own_predicate()):
            total += self._append_copy_block(blk, known="raise")
        col = self._collection()
        if col is not None and documents_stream is not None:
            t = col.table
            with t.lock:
                for d in pgbridge.read_pgcopy_documents(documents_stream):
                    d = dict(d)
                    did = d.pop("id")
                    cur = t.documents.get(did)
                    if cur is None or any(cur.get(k) != v for k, v in d.items()):
                        t.register_document(did, **d)
        return total

    def dump_to_pgcopy(self, chunks_stream: Any, documents_stream: Any = None, batch: int = 65536,
                       only_ids: Optional[Iterable[int]] = None) -> int:
        """The inverse of load_from_pgcopy: every live row of this collection -- id, document_id, chunk_index, chunk_text,
        metadata, the STORED vector -- as a binary COPY stream `COPY document_chunks (id, document_id, chunk_index,
        chunk_text, metadata, embedding) FROM STDIN (FORMAT binary)` accepts, and the mirrored `documents` columns. With it
        the collection survives a restart of the process (dump -> load) and can seed a fresh Postgres table. Document ids
        must be integers (they are `documents.id` in the reference). only_ids: restrict the chunk stream to these row ids
        (`WHERE id = ANY(%s)`; ids that are not live are left out) -- the stream a refresh_from_pgcopy of ANOTHER process asks
        for when this process is the writer. Returns the number of chunks written."""
        from . import pgbridge
        col = self._collection()
        if col is None:
            pgbridge.write_pgcopy_chunks(chunks_stream, [])
            if documents_stream is not None:
                pgbridge.write_pgcopy_documents(documents_stream, [])
            return 0
        t = col.table
        with t.lock:
            rids = t.live_rids()
            if only_ids is not None:
                want = np.asarray(sorted({int(r) for r in only_ids}), np.int64)
                rids = want[t.pos_many(want) >= 0] if len(want) else want
            slots = col.index.lookup(rids)
            if self._shards > 1:
                # row-sharded index: every rank holds the vectors of its own ids only (id % shards == rank) and dumps exactly
                # those rows -- each rank passes its OWN stream; the per-rank streams together are the collection, and
                # load_from_pgcopy of all of them (in any order, on any shard count) restores it
                mine = slots >= 0
                owned = getattr(col.index, "_mine", None)
                if owned is not None and (np.asarray(owned(rids)) & ~mine).any():
                    raise RuntimeError("dump_to_pgcopy: a table row of this shard has no vector in the index")
                rids, slots = rids[mine], slots[mine]
            elif (slots < 0).any():
                raise RuntimeError("dump_to_pgcopy: a table row has no vector in the index")

            def rows():
                for o in range(0, len(rids), batch):
                    vecs = col.index.fetch(slots[o:o + batch])
                    for j, rid in enumerate(rids[o:o + batch].tolist()):
                        p = t.pos(rid)
                        yield (rid, t.document_id_at(p), t.chunk_index_at(p), t.text_at(p), t.metadata_at(p), vecs[j])
            pgbridge.write_pgcopy_chunks(chunks_stream, rows())
            if documents_stream is not None:
                pgbridge.write_pgcopy_documents(documents_stream, [dict(d, id=k) for k, d in t.documents.items()])
        return int(len(rids))

    @classmethod
    def from_texts(
        cls: Type["ArchiHipVectorStore"],
        texts: List[str],
        embedding: Any,
        metadatas: Optional[List[Dict[str, Any]]] = None,
        **kwargs: Any,
    ) -> "ArchiHipVectorStore":
        pg_config = kwargs.pop("pg_config")          # KeyError when missing, like the reference (:557)
        collection_name = kwargs.pop("collection_name", "default")
        distance_metric = kwargs.pop("distance_metric", "cosine")
        index_factory = kwargs.pop("index_factory", None)
        store = cls(pg_config=pg_config, embedding_function=embedding, collection_name=collection_name,
                    distance_metric=distance_metric, index_factory=index_factory)
        store.add_texts(texts, metadatas=metadatas, **kwargs)
        return store

    def count(self) -> int:
        """Reference :570-585."""
        col = self._collection()
        return 0 if col is None else int(col.index.count())


class HostBm25:
    """Okapi BM25 over the chunk texts of a ChunkTable: host-side stand-in for the pg_textsearch index the
    reference builds when that extension exists (src/cli/templates/init.sql:294-300). pg_textsearch is
    third-party and absent from the image, so its tokenisation (text_config 'english': stemming, stop
    words) is NOT reproduced -- any object with `scores(query, table) -> {row id: score}` can replace this
    one. `sign=-1` reproduces the `<@>` operator's convention of returning negated scores.
    """

    _tok = re.compile(r"\w+")

    def __init__(self, k1: float = 1.2, b: float = 0.75, sign: float = 1.0):
        self.k1, self.b, self.sign = k1, b, sign
        self._reset(None)

    def _reset(self, table: Optional[ChunkTable]) -> None:
        self._table_key = None if table is None else (id(table), table.text_epoch)
        self._terms: Dict[str, int] = {}
        self._ppos: List[Any] = []          # per term: array('I') of table positions (append order = position order)
        self._ptf: List[Any] = []           # per term: array('I') of term frequencies, aligned
        self._dlen = np.zeros(1024, np.int32)
        self._upto = 0                      # table positions [0, _upto) are indexed

    def _sync(self, table: ChunkTable) -> None:
        """Index the positions appended since the last call. The table is append-only between vacuums: deletes only clear
        `alive` (applied at query time), so the postings never need rewriting; a text changed in place or a vacuum
        (`text_epoch`) starts over. The first version rebuilt a dict-of-dicts index of the WHOLE table on every version
        change -- every insert -- which is a pass over the corpus per hybrid query at ingestion time."""
        from array import array
        if self._table_key != (id(table), table.text_epoch):
            self._reset(table)
        n = table.positions
        if n > len(self._dlen):
            grown = np.zeros(max(n, 2 * len(self._dlen)), np.int32)
            grown[: self._upto] = self._dlen[: self._upto]
            self._dlen = grown
        terms, ppos, ptf, findall = self._terms, self._ppos, self._ptf, self._tok.findall
        alive = table._alive
        for p in range(self._upto, n):
            if not alive[p]:
                continue                     # dead on arrival (replaced within its own batch): never scored
            toks = findall(table.text_at(p).lower())
            self._dlen[p] = len(toks)
            counts: Dict[str, int] = {}
            for w in toks:
                counts[w] = counts.get(w, 0) + 1
            for w, c in counts.items():
                tid = terms.get(w)
                if tid is None:
                    tid = terms[w] = len(ppos)
                    ppos.append(array("I"))
                    ptf.append(array("I"))
                ppos[tid].append(p)
                ptf[tid].append(c)
        self._upto = n

    def scores_arrays(self, query: str, table: ChunkTable) -> Tuple[np.ndarray, np.ndarray]:
        """(table positions, scores) of the live rows matching at least one query term, positions ascending. Arithmetic and
        its order are the scalar formulation's (idf * tf * (k1 + 1) / (tf + k1 * (1 - b + b * len / avg)), terms added
        in query order), evaluated on whole posting arrays."""
        with table.lock:
            self._sync(table)
            npos = self._upto
            alive = table._alive[:npos]
            n = int(alive.sum())
            if n == 0:
                return np.zeros(0, np.int64), np.zeros(0, np.float64)
            avg = int(self._dlen[:npos][alive].sum()) / n
            acc = np.zeros(npos, np.float64)
            seen = np.zeros(npos, bool)
            for w in dict.fromkeys(self._tok.findall(query.lower())):
                tid = self._terms.get(w)
                if tid is None:
                    continue
                pos = np.frombuffer(self._ppos[tid], dtype=np.uint32).astype(np.int64)
                tf = np.frombuffer(self._ptf[tid], dtype=np.uint32).astype(np.float64)
                live = alive[pos]
                pos, tf = pos[live], tf[live]
                df = len(pos)
                if df == 0:
                    continue
                idf = math.log(1.0 + (n - df + 0.5) / (df + 0.5))
                if avg > 0:
                    norm = tf + self.k1 * ((1.0 - self.b) + self.b * self._dlen[pos] / avg)
                else:
                    norm = tf + self.k1 * (1.0 - self.b)
                acc[pos] = acc[pos] + idf * tf * (self.k1 + 1.0) / norm
                seen[pos] = True
            hit = np.flatnonzero(seen)
            return hit, self.sign * acc[hit]

    def scores(self, query: str, table: ChunkTable) -> Dict[int, float]:
        with table.lock:
            pos, sc = self.scores_arrays(query, table)
            return dict(zip(table.rids_at(pos).tolist(), sc.tolist()))


class ArchiHipHybridVectorStore(ArchiHipVectorStore):
    """ArchiHipVectorStore + `hybrid_search` (reference :366-491), for deployments that attach a BM25
    scorer: `pg_config["hip"]["bm25"]` or the `bm25=` keyword (an object with
    `scores(query, table) -> {row id: bm25 score}`; HostBm25 is the built-in one).

    The reference scores EVERY row: combined = (1.0 - distance) * w_s + COALESCE(bm25, 0) * w_b, ORDER BY
    combined DESC LIMIT k (:435-457). Rows without a BM25 match have combined = semantic * w_s, so among them
    the best k are the GPU scan's top-k (with the matches masked out, w_s >= 0); rows with a match need their
    exact distance whatever its rank (ak_index_distances). The union of the two legs holds the exact answer;
    its top-k by (combined desc, id asc) is returned. A NaN combined score (zero-norm or non-finite vectors) ranks FIRST,
    as Postgres orders float8 NaN above every number under DESC: the few rows that can produce one are tracked at insert
    time (ChunkTable.suspects) and scored through the same exact-distance call as the BM25 hits.
    """

    def __init__(self, *args: Any, bm25: Any = None, **kwargs: Any):
        super().__init__(*args, **kwargs)
        self._bm25 = bm25 if bm25 is not None else (self._pg_config.get("hip", {}) or {}).get("bm25")

    def hybrid_search(self, query: str, k: int = 4, *, semantic_weight: float = 0.7, bm25_weight: float = 0.3,
                      **kwargs: Any) -> List[Tuple[Any, float]]:
        query_embedding = self._embedding_function.embed_query(query)
        metadata_filter = kwargs.get("filter", {}) or {}
        include_deleted = kwargs.get("include_deleted", False)
        if self._bm25 is None:                                                      # :415-418
            raise RuntimeError("Hybrid search requires pg_textsearch BM25 index on document_chunks; none found.")
        if semantic_weight < 0:
            raise ValueError("semantic_weight must be >= 0 (the two-leg evaluation relies on it)")
        col = self._collection()
        results: List[Tuple[Any, float]] = []
        if col is not None and k > 0:
            t = col.table
            q = np.asarray([float(x) for x in query_embedding], dtype=np.float32)    # a4 round trip (:389)
            def allowed_mask(rids: np.ndarray) -> np.ndarray:
                if allowed is None:
                    return np.ones(len(rids), bool)
                if isinstance(allowed, _AllBut):
                    return ~np.isin(rids, allowed.denied)
                return np.isin(rids, allowed)

            with t.lock:
                row_filter, allowed, _ = self._where(col, metadata_filter, include_deleted)
                # BM25 leg as arrays (a frequent word matches a large share of the corpus: nothing below is a Python pass over
                # the hits); a plug-in scorer that only offers scores() -> {row id: score} goes through the same arrays
                if callable(getattr(self._bm25, "scores_arrays", None)):
                    hpos, hsc = self._bm25.scores_arrays(query, t)
                    hrid = t.rids_at(hpos) if len(hpos) else np.zeros(0, np.int64)
                else:
                    sd = self._bm25.scores(query, t)
                    hrid = np.fromiter(sd.keys(), np.int64, len(sd))
                    hsc = np.fromiter((float(v) for v in sd.values()), np.float64, len(sd))
                    live = t.pos_many(hrid) >= 0 if len(hrid) else np.zeros(0, bool)
                    hrid, hsc = hrid[live], hsc[live]
                    o = np.argsort(hrid, kind="stable")
                    hrid, hsc = hrid[o], hsc[o]
                keep = allowed_mask(hrid)
                hrid, hsc = hrid[keep], np.asarray(hsc, np.float64)[keep]
                # rows whose semantic score can be NaN (zero / non-finite vectors; every row when the QUERY is degenerate):
                # Postgres' ORDER BY combined DESC ranks NaN above every number, the top-k scan ranks it last -> their
                # exact distances are fetched like the BM25 hits'
                q_bad = bool(_suspect_rows(q[None, :])[0])
                pool = t.live_rids().astype(np.int64) if q_bad else np.fromiter(t.suspects, np.int64, len(t.suspects))
                pool = pool[allowed_mask(pool)]
                hit_ids = np.union1d(hrid, pool)                        # sorted, unique
                bm = np.zeros(len(hit_ids), np.float64)
                bm[np.searchsorted(hit_ids, hrid)] = hsc
            # the two GPU legs run without the table lock (see similarity_search_by_vector_with_score). The distances leg goes by
            # row id; the scan's mask (WHERE clause minus the hits) is built under the lock for one layout epoch and refused by
            # the library if the index has moved on by the time the scan starts (_search_snapshot)
            c_score = np.zeros(0, np.float64)
            c_id = np.zeros(0, np.int64)
            if len(hit_ids):
                hd, found = col.index.distances(q, hit_ids)
                c_score = ((1.0 - hd) * semantic_weight + bm * bm25_weight)[found]
                c_id = hit_ids[found]

            def scan_mask():
                rf, _, epoch = self._where(col, metadata_filter, include_deleted)
                if not len(hit_ids):
                    return rf, epoch
                mask = np.ones(col.index.layout()[0], dtype=np.uint8) if rf is None else rf.copy()
                slots = col.index.lookup(hit_ids)
                mask[slots[slots >= 0]] = 0
                return mask, epoch
            ids, dist, cnt = self._search_snapshot(col, q, k, scan_mask)
            m = int(cnt[0])
            c_score = np.concatenate([c_score, (1.0 - dist[0, :m].astype(np.float64)) * semantic_weight + 0 * bm25_weight])
            c_id = np.concatenate([c_id, ids[0, :m].astype(np.int64)])
            # ORDER BY combined_score DESC LIMIT k: float8 NaN sorts above every number; ties by id (a build decision).
            # Only the NaNs and the scores that reach the k-th largest need the full ordering.
            isnan = c_score != c_score
            nn = int(isnan.sum())
            sel = np.flatnonzero(isnan)
            rest = np.flatnonzero(~isnan)
            want = k - nn
            if want > 0 and len(rest):
                if len(rest) > want:
                    kth = np.partition(c_score[rest], len(rest) - want)[len(rest) - want]
                    rest = rest[c_score[rest] >= kth]
                sel = np.concatenate([sel, rest])
            order = sel[np.lexsort((c_id[sel], -np.where(isnan[sel], 0.0, c_score[sel]), ~isnan[sel]))][:k]
            with t.lock:
                for j in order.tolist():
                    pz = t.pos(int(c_id[j]))
                    if pz >= 0:
                        results.append((self._document(t, pz), float(c_score[j])))
        if not results:                                                              # :467-469
            return self.similarity_search_with_score(query, k=k, **kwargs)
        return results
