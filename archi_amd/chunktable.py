"""ChunkTable -- the columns of `document_chunks` other than `embedding`, held on the host, columnar.

The reference keeps `id, document_id, chunk_index, chunk_text, embedding, metadata` in Postgres
(/root/reference/src/cli/templates/init.sql:256-274) and runs its WHERE clauses, deletes and the `DISTINCT
metadata->>'resource_hash'` of the sync step (src/data_manager/vectorstore/manager.py:216-252) inside the database.
Here the vectors live in HBM (HipIndex) and everything else in this table, laid out for the corpus sizes the index
handles (10M+ chunks):

  ids / document number / chunk_index / chunk_id hash   numpy columns, one entry per row position (append-only)
  chunk_text, metadata (JSON text, like JSONB)          one growing byte buffer each + (start, length) columns
  document_id -> row positions                          inverted map: delete(document_id=...) and the UNIQUE
                                                        (document_id, chunk_index) lookup of the upsert touch one document
  metadata key -> value -> row positions                inverted maps for the keys a WHERE clause or the sync step asks
                                                        for (`resource_hash` always; others are built on first use by one
                                                        pass and kept current afterwards)

Row ids are the SERIAL primary key: assigned in ascending order, so id -> position is a binary search (a dict only if a
caller ever appends out of order). Nothing here is O(rows) Python per request; the only full passes are numpy.
"""
from __future__ import annotations

import json
import threading
from typing import Any, Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np


def meta_text(value: Any) -> Optional[str]:
    """`metadata->>'key'` of Postgres: text as it is, anything else as its JSON text, NULL stays NULL."""
    if value is None:
        return None
    return value if isinstance(value, str) else json.dumps(value)


class _Bytes:
    """Variable-length byte rows: one growing buffer, (start, length) per row; a replaced row's old bytes are garbage until
    the table is vacuumed."""

    def __init__(self) -> None:
        self.buf = bytearray()
        self.start = np.zeros(1024, np.int64)
        self.length = np.zeros(1024, np.int64)

    def _room(self, n: int) -> None:
        if n > len(self.start):
            cap = max(n, 2 * len(self.start))
            self.start = np.concatenate([self.start, np.zeros(cap - len(self.start), np.int64)])
            self.length = np.concatenate([self.length, np.zeros(cap - len(self.length), np.int64)])

    def put(self, pos: int, data: bytes) -> None:
        self._room(pos + 1)
        self.start[pos] = len(self.buf)
        self.length[pos] = len(data)
        self.buf += data

    def put_many(self, pos: int, items: List[bytes]) -> None:
        """items -> positions pos .. pos + len(items) - 1: one buffer extension, the offsets as one cumulative sum"""
        n = len(items)
        self._room(pos + n)
        lens = np.fromiter(map(len, items), np.int64, n)
        ends = np.cumsum(lens)
        self.start[pos: pos + n] = len(self.buf) + ends - lens
        self.length[pos: pos + n] = lens
        self.buf += b"".join(items)

    def get(self, pos: int) -> bytes:
        s = int(self.start[pos])
        return bytes(self.buf[s:s + int(self.length[pos])])


class _RowsView:
    """Mapping view {row id: {"document_id", "chunk_index", "text", "metadata"}} over the columns (what the first
    version stored as a dict of dicts). Row dicts are materialised per access: mutate through ChunkTable.update_row."""

    def __init__(self, table: "ChunkTable") -> None:
        self._t = table

    def __len__(self) -> int:
        return len(self._t)

    def __contains__(self, rid: Any) -> bool:
        return self._t.pos(rid) >= 0

    def __iter__(self) -> Iterator[int]:
        return iter(self._t.live_rids().tolist())

    def __getitem__(self, rid: int) -> Dict[str, Any]:
        r = self._t.row(rid)
        if r is None:
            raise KeyError(rid)
        return r

    def get(self, rid: int, default: Any = None) -> Any:
        r = self._t.row(rid)
        return default if r is None else r

    def keys(self) -> List[int]:
        return self._t.live_rids().tolist()

    def values(self) -> List[Dict[str, Any]]:
        return [self._t.row(r) for r in self.keys()]

    def items(self) -> List[Tuple[int, Dict[str, Any]]]:
        return [(r, self._t.row(r)) for r in self.keys()]

    def __setitem__(self, rid: int, row: Dict[str, Any]) -> None:
        if self._t.pos(rid) >= 0:
            self._t.update_row(rid, **{("text" if k == "text" else k): v for k, v in row.items()})
        else:
            self._t.append(rid, row.get("document_id"), row.get("chunk_index", 0), row.get("text", ""), row.get("metadata"))

    def pop(self, rid: int, default: Any = None) -> Any:
        r = self._t.row(rid)
        if r is None:
            return default
        self._t.kill(rid)
        return r

    def __eq__(self, other: Any) -> bool:
        return dict(self.items()) == (dict(other.items()) if isinstance(other, _RowsView) else other)


class ChunkTable:
    """Host-side rows of one collection: everything in `document_chunks` except the vector. Callers hold `lock` around
    compound operations (the store does)."""

    ALWAYS_INDEXED = ("resource_hash",)

    def __init__(self) -> None:
        self.lock = threading.RLock()
        self.next_id = 1                                   # SERIAL PRIMARY KEY
        self.documents: Dict[Any, Dict[str, Any]] = {}     # documents.id -> {resource_hash, display_name, source_type, url, is_deleted}
        self.version = 0                                   # bumped on every row change (text-index caches key on it)
        self.text_epoch = 0                                # bumped when a stored text changes in place or positions move (vacuum):
                                                           # an incremental text index starts over; appends and kills do not bump it
        self.doc_version = 0                               # bumped on every `documents` change (soft deletes)
        self.where_cache: Dict[Any, Any] = {}              # WHERE-clause masks of the current (version, doc_version)
        self.suspects: set = set()                         # row ids whose distance to a healthy query can be NaN
        self._n = 0                                        # row positions in use (live + dead)
        self._alive_n = 0
        cap = 1024
        self._ids = np.zeros(cap, np.int64)
        self._alive = np.zeros(cap, bool)
        self._doc = np.full(cap, -1, np.int32)             # document number (index into _dockeys), -1 = NULL
        self._cidx = np.zeros(cap, np.int32)
        self._chash = np.zeros(cap, np.int64)              # hash of metadata['chunk_id'] (0 = none): delete(ids=...) prefilter
        self._ver = np.zeros(cap, np.int64)                # row version of the table this mirror was refreshed from (xmin-style; 0 = unknown)
        self._text = _Bytes()
        self._meta = _Bytes()
        self._sorted = True                                # ids ascending with position
        self._idmap: Optional[Dict[int, int]] = None       # only when a caller appended out of order
        self._docno: Dict[Any, int] = {}
        self._dockeys: List[Any] = []
        self._docrows: List[List[int]] = []                # document number -> positions (dead ones pruned lazily)
        self._kidx: Dict[str, Dict[str, List[int]]] = {k: {} for k in self.ALWAYS_INDEXED}

    # ---- documents -------------------------------------------------------------------------------------------------
    def register_document(self, document_id: Any, **cols: Any) -> None:
        """Mirror of a `documents` row (catalog side; collectors own the real table)."""
        with self.lock:
            self.documents.setdefault(document_id, {}).update(cols)
            self.doc_version += 1

    # ---- columns ---------------------------------------------------------------------------------------------------
    def _room(self, n: int) -> None:
        if n <= len(self._ids):
            return
        cap = max(n, 2 * len(self._ids))

        def grow(a, fill):
            b = np.full(cap, fill, a.dtype)
            b[: len(a)] = a
            return b
        self._ids, self._alive, self._doc = grow(self._ids, 0), grow(self._alive, False), grow(self._doc, -1)
        self._cidx, self._chash, self._ver = grow(self._cidx, 0), grow(self._chash, 0), grow(self._ver, 0)

    def __len__(self) -> int:
        return self._alive_n

    @property
    def positions(self) -> int:
        return self._n

    def pos(self, rid: Any) -> int:
        """Position of a LIVE row id, -1 otherwise."""
        try:
            rid = int(rid)
        except (TypeError, ValueError):
            return -1
        if self._idmap is not None:
            p = self._idmap.get(rid, -1)
        else:
            p = int(np.searchsorted(self._ids[: self._n], rid))
            if p >= self._n or self._ids[p] != rid:
                return -1
        return p if p >= 0 and self._alive[p] else -1

    def pos_many(self, rids: Iterable[int]) -> np.ndarray:
        r = np.asarray(list(rids) if not isinstance(rids, np.ndarray) else rids, dtype=np.int64)
        if self._idmap is not None:
            p = np.array([self._idmap.get(int(x), -1) for x in r], dtype=np.int64)
        else:
            p = np.searchsorted(self._ids[: self._n], r).astype(np.int64)
            ok = p < self._n
            ok[ok] &= self._ids[p[ok]] == r[ok]
            p[~ok] = -1
        good = p >= 0
        good[good] &= self._alive[p[good]]
        p[~good] = -1
        return p

    def live_rids(self) -> np.ndarray:
        return self._ids[: self._n][self._alive[: self._n]]

    def _doc_number(self, document_id: Any) -> int:
        if document_id is None:
            return -1
        no = self._docno.get(document_id)
        if no is None:
            no = len(self._dockeys)
            self._docno[document_id] = no
            self._dockeys.append(document_id)
            self._docrows.append([])
        return no

    # ---- writes ----------------------------------------------------------------------------------------------------
    def append(self, rid: int, document_id: Any, chunk_index: int, text: str, metadata: Optional[Dict[str, Any]]) -> int:
        """INSERT one row (the caller decided about ON CONFLICT). metadata is stored as its JSON text, like JSONB."""
        p = self._n
        md = metadata if metadata is not None else {}
        enc_text, enc_meta = text.encode("utf-8", "surrogatepass"), json.dumps(md).encode("utf-8")     # may raise: nothing touched yet
        self._room(p + 1)
        rid = int(rid)
        if p and rid <= int(self._ids[p - 1]) and self._idmap is None:
            self._sorted = False
            self._idmap = {int(self._ids[i]): i for i in range(p)}     # rare: ids handed in out of order
        self._ids[p] = rid
        self._alive[p] = True
        no = self._doc_number(document_id)
        self._doc[p] = no
        if no >= 0:
            self._docrows[no].append(p)
        self._cidx[p] = int(chunk_index)
        cid = md.get("chunk_id") if isinstance(md, dict) else None
        self._chash[p] = hash(cid) if isinstance(cid, str) else 0
        self._text.put(p, enc_text)
        self._meta.put(p, enc_meta)
        if isinstance(md, dict):
            for key, idx in self._kidx.items():
                v = meta_text(md.get(key))
                if v is not None:
                    idx.setdefault(v, []).append(p)
        if self._idmap is not None:
            self._idmap[rid] = p
        self._n = p + 1
        self._alive_n += 1
        if rid >= self.next_id:
            self.next_id = rid + 1
        return p

    def has_document(self, document_id: Any) -> bool:
        """True if a row was ever appended under this document id (live or not): without one, ON CONFLICT cannot fire."""
        return document_id is not None and document_id in self._docno

    def append_block(self, document_id: Any, texts: List[str], metadatas: List[Dict[str, Any]]) -> int:
        """INSERT len(texts) rows of ONE document with chunk_index 0 .. n-1 and the next n row ids; returns the first id.
        What n append() calls would leave behind, with the columns written as slices and each byte column extended once
        (ingestion: 8.6 us per row in append() -- numpy scalar stores, two buffer extensions -- against ~3 here; the JSON
        text of the metadata is what is left)."""
        n = len(texts)
        rid0 = self.next_id
        if n == 0:
            return rid0
        p0 = self._n
        if self._idmap is not None or (p0 and rid0 <= int(self._ids[p0 - 1])):
            for i, (text, md) in enumerate(zip(texts, metadatas)):      # ids out of order somewhere: the general path
                self.append(self.next_id, document_id, i, text, md)
            return rid0
        # everything that can fail (a text that is not a str, metadata JSON cannot express) BEFORE the first column is touched:
        # a failed block leaves the table as it was
        mds = [md if md is not None else {} for md in metadatas]
        if len(mds) != n:
            raise ValueError("append_block: one metadata entry per text")
        enc_text = [t.encode("utf-8", "surrogatepass") for t in texts]
        dumps = json.dumps
        enc_meta = [dumps(md).encode("utf-8") for md in mds]
        chash = np.fromiter(
            (hash(c) if isinstance(c, str) else 0 for c in (md.get("chunk_id") if isinstance(md, dict) else None for md in mds)),
            self._chash.dtype, n)
        self._room(p0 + n)
        self._ids[p0: p0 + n] = np.arange(rid0, rid0 + n, dtype=self._ids.dtype)
        self._alive[p0: p0 + n] = True
        no = self._doc_number(document_id)
        self._doc[p0: p0 + n] = no
        if no >= 0:
            self._docrows[no].extend(range(p0, p0 + n))
        self._cidx[p0: p0 + n] = np.arange(n, dtype=self._cidx.dtype)
        self._chash[p0: p0 + n] = chash
        self._text.put_many(p0, enc_text)
        self._meta.put_many(p0, enc_meta)
        for key, idx in self._kidx.items():
            for i, md in enumerate(mds):
                if isinstance(md, dict):
                    v = meta_text(md.get(key))
                    if v is not None:
                        idx.setdefault(v, []).append(p0 + i)
        self._n = p0 + n
        self._alive_n += n
        self.next_id = rid0 + n
        return rid0

    def append_rows(self, rids: np.ndarray, document_ids: List[Any], chunk_index: np.ndarray, text_bytes: List[bytes],
                    metadatas: List[Optional[Dict[str, Any]]], meta_json: List[Optional[bytes]]) -> None:
        """INSERT rows with GIVEN ids (ascending, all above the table's last id), documents and chunk indices -- the bulk-load
        path (a COPY dump): columns as slices, the stored UTF-8 and JSON text taken as they come. Falls back to append() row by
        row when the ids do not extend the table in order."""
        n = len(rids)
        if n == 0:
            return
        rids = np.asarray(rids, np.int64)
        p0 = self._n
        in_order = self._idmap is None and (n == 1 or bool((np.diff(rids) > 0).all())) and (p0 == 0 or int(rids[0]) > int(self._ids[p0 - 1]))
        mds = [md if md is not None else {} for md in metadatas]
        if not in_order:
            for i in range(n):
                self.append(int(rids[i]), document_ids[i], int(chunk_index[i]), text_bytes[i].decode("utf-8", "surrogatepass"), mds[i])
            return
        enc_meta = [mj if mj is not None else json.dumps(md).encode("utf-8") for mj, md in zip(meta_json, mds)]
        chash = np.fromiter(
            (hash(c) if isinstance(c, str) else 0 for c in (md.get("chunk_id") if isinstance(md, dict) else None for md in mds)),
            self._chash.dtype, n)
        self._room(p0 + n)
        self._ids[p0: p0 + n] = rids
        self._alive[p0: p0 + n] = True
        docno = self._doc_number
        nos = np.fromiter((docno(d) for d in document_ids), self._doc.dtype, n)
        self._doc[p0: p0 + n] = nos
        docrows = self._docrows
        for i, no in enumerate(nos.tolist()):
            if no >= 0:
                docrows[no].append(p0 + i)
        self._cidx[p0: p0 + n] = np.asarray(chunk_index, self._cidx.dtype)
        self._chash[p0: p0 + n] = chash
        self._text.put_many(p0, text_bytes)
        self._meta.put_many(p0, enc_meta)
        for key, idx in self._kidx.items():
            for i, md in enumerate(mds):
                if isinstance(md, dict):
                    v = meta_text(md.get(key))
                    if v is not None:
                        idx.setdefault(v, []).append(p0 + i)
        self._n = p0 + n
        self._alive_n += n
        self.next_id = max(self.next_id, int(rids[-1]) + 1)

    def kill(self, rid: int) -> bool:
        p = self.pos(rid)
        if p < 0:
            return False
        self._alive[p] = False
        self._alive_n -= 1
        no = int(self._doc[p])
        if no >= 0:
            rows = self._docrows[no]
            try:
                rows.remove(p)
            except ValueError:
                pass
        if self._idmap is not None:
            self._idmap.pop(int(rid), None)
        return True

    def update_row(self, rid: int, **cols: Any) -> None:
        """UPDATE of single columns of a live row (tests, catalog back-fills): document_id, chunk_index, text, metadata."""
        p = self.pos(rid)
        if p < 0:
            raise KeyError(rid)
        if "document_id" in cols:
            old = int(self._doc[p])
            if old >= 0 and p in self._docrows[old]:
                self._docrows[old].remove(p)
            no = self._doc_number(cols["document_id"])
            self._doc[p] = no
            if no >= 0:
                self._docrows[no].append(p)
        if "chunk_index" in cols:
            self._cidx[p] = int(cols["chunk_index"])
        if "text" in cols:
            self._text.put(p, cols["text"].encode("utf-8", "surrogatepass"))
            self.text_epoch += 1
        if "metadata" in cols:
            md = cols["metadata"] if cols["metadata"] is not None else {}
            old_md = self.metadata_at(p)
            for key, idx in self._kidx.items():          # keep the inverted maps current
                ov, nv = meta_text(old_md.get(key)), meta_text(md.get(key))
                if ov != nv:
                    if ov is not None and p in idx.get(ov, ()):
                        idx[ov].remove(p)
                    if nv is not None:
                        idx.setdefault(nv, []).append(p)
            cid = md.get("chunk_id")
            self._chash[p] = hash(cid) if isinstance(cid, str) else 0
            self._meta.put(p, json.dumps(md).encode("utf-8"))
        self.version += 1

    def set_versions(self, rids: Iterable[int], versions: Iterable[int]) -> None:
        """Record the row versions (`xmin`-style, any int64 that changes when the row is rewritten) of live rows: what
        ArchiHipVectorStore.refresh_from_pgcopy compares to find rows another process UPDATEd in place. Rows that are not
        live are skipped."""
        r = np.asarray(list(rids) if not isinstance(rids, np.ndarray) else rids, np.int64)
        v = np.asarray(list(versions) if not isinstance(versions, np.ndarray) else versions, np.int64)
        p = self.pos_many(r)
        ok = p >= 0
        self._ver[p[ok]] = v[ok]

    def versions_of(self, rids: Iterable[int]) -> np.ndarray:
        """Recorded versions of the listed rows (0 = unknown / not live)."""
        p = self.pos_many(rids)
        out = np.zeros(len(p), np.int64)
        ok = p >= 0
        out[ok] = self._ver[p[ok]]
        return out

    def max_rid(self) -> int:
        """The largest row id ever stored (live or dead), 0 for an empty table: `WHERE id > %s` of the tail refresh."""
        return int(self.next_id) - 1

    # ---- reads -----------------------------------------------------------------------------------------------------
    def text_at(self, p: int) -> str:
        return self._text.get(p).decode("utf-8", "surrogatepass")

    def metadata_at(self, p: int) -> Dict[str, Any]:
        return json.loads(self._meta.get(p))

    def document_id_at(self, p: int) -> Any:
        no = int(self._doc[p])
        return None if no < 0 else self._dockeys[no]

    def chunk_index_at(self, p: int) -> int:
        return int(self._cidx[p])

    def row_at(self, p: int) -> Dict[str, Any]:
        return {"document_id": self.document_id_at(p), "chunk_index": int(self._cidx[p]), "text": self.text_at(p),
                "metadata": self.metadata_at(p)}

    def row(self, rid: int) -> Optional[Dict[str, Any]]:
        p = self.pos(rid)
        return None if p < 0 else self.row_at(p)

    @property
    def rows(self) -> _RowsView:
        return _RowsView(self)

    @property
    def by_doc_chunk(self) -> Dict[Tuple[Any, int], int]:
        """UNIQUE(document_id, chunk_index) as a dict (materialised: tests and tooling; the upsert uses find())."""
        out: Dict[Tuple[Any, int], int] = {}
        for no, rows in enumerate(self._docrows):
            for p in rows:
                if self._alive[p]:
                    out[(self._dockeys[no], int(self._cidx[p]))] = int(self._ids[p])
        return out

    def find(self, document_id: Any, chunk_index: int) -> Optional[int]:
        """Row id holding (document_id, chunk_index): the most recently inserted live one."""
        no = self._docno.get(document_id) if document_id is not None else None
        if no is None:
            return None
        for p in reversed(self._docrows[no]):
            if self._alive[p] and self._cidx[p] == chunk_index:
                return int(self._ids[p])
        return None

    def rids_of_document(self, document_id: Any) -> List[int]:
        no = self._docno.get(document_id) if document_id is not None else None
        if no is None:
            return []
        return [int(self._ids[p]) for p in self._docrows[no] if self._alive[p]]

    def positions_of_documents(self, document_ids: Iterable[Any]) -> np.ndarray:
        out: List[int] = []
        for d in document_ids:
            no = self._docno.get(d)
            if no is not None:
                out.extend(p for p in self._docrows[no] if self._alive[p])
        return np.asarray(out, dtype=np.int64)

    def rids_of_chunk_ids(self, chunk_ids: Iterable[str]) -> List[int]:
        """`metadata->>'chunk_id' = ANY(ids)`: hash column pre-filter (one numpy pass), then the stored value decides."""
        wanted = {c for c in chunk_ids if isinstance(c, str)}
        if not wanted or not self._n:
            return []
        hs = np.fromiter((hash(c) for c in wanted), dtype=np.int64, count=len(wanted))
        cand = np.flatnonzero(np.isin(self._chash[: self._n], hs) & self._alive[: self._n])
        return [int(self._ids[p]) for p in cand if self.metadata_at(int(p)).get("chunk_id") in wanted]

    def _index_key(self, key: str) -> Dict[str, List[int]]:
        idx = self._kidx.get(key)
        if idx is None:                                   # first use of this key: one pass, current from here on
            idx = {}
            for p in np.flatnonzero(self._alive[: self._n]).tolist():
                v = meta_text(self.metadata_at(p).get(key))
                if v is not None:
                    idx.setdefault(v, []).append(p)
            self._kidx[key] = idx
        return idx

    def positions_matching(self, metadata_filter: Dict[str, Any]) -> np.ndarray:
        """Live positions with `metadata->>key = str(value)` for every pair (postgres_vectorstore.py:300-302)."""
        cur: Optional[np.ndarray] = None
        for key, value in metadata_filter.items():
            plist = self._index_key(str(key)).get(str(value), [])
            arr = np.asarray(plist, dtype=np.int64)
            arr = arr[self._alive[arr]] if len(arr) else arr
            cur = arr if cur is None else np.intersect1d(cur, arr, assume_unique=False)
            if not len(cur):
                break
        if cur is None:
            cur = np.flatnonzero(self._alive[: self._n])
        return np.unique(cur)

    def distinct_values(self, key: str) -> set:
        """SELECT DISTINCT metadata->>key ... WHERE it IS NOT NULL (manager.py:221-229)."""
        idx = self._index_key(key)
        return {v for v, plist in idx.items() if any(self._alive[p] for p in plist)}

    def rids_at(self, positions: np.ndarray) -> np.ndarray:
        return self._ids[np.asarray(positions, dtype=np.int64)]

    # ---- maintenance -----------------------------------------------------------------------------------------------
    def dead_fraction(self) -> float:
        return 0.0 if not self._n else 1.0 - self._alive_n / self._n

    def vacuum(self) -> None:
        """Rewrite the columns without dead rows and replaced byte ranges (positions change; ids do not)."""
        live = np.flatnonzero(self._alive[: self._n])
        fresh = ChunkTable()
        for p in live.tolist():
            fresh.append(int(self._ids[p]), self.document_id_at(p), int(self._cidx[p]), self.text_at(p), self.metadata_at(p))
        for key in self._kidx:
            fresh._index_key(key)
        fresh._ver[: len(live)] = self._ver[live]
        for name in ("_n", "_alive_n", "_ids", "_alive", "_doc", "_cidx", "_chash", "_ver", "_text", "_meta", "_sorted", "_idmap",
                     "_docno", "_dockeys", "_docrows", "_kidx"):
            setattr(self, name, getattr(fresh, name))
        self.version += 1
        self.text_epoch += 1
