"""Persistence bridge to archi's `document_chunks` table.

The reference keeps embeddings in `document_chunks.embedding vector(D)`
(/root/reference/src/cli/templates/init.sql:256-274) and inserts them as text
(`%s::vector`, src/data_manager/vectorstore/manager.py:414-422). To bring an existing deployment's
vectors into the GPU index without re-embedding, a maintainer runs

    COPY (SELECT c.id, c.embedding FROM document_chunks c
          LEFT JOIN documents d ON c.document_id = d.id
          WHERE d.id IS NULL OR NOT d.is_deleted) TO STDOUT (FORMAT binary)

and feeds the stream to `read_pgcopy_vectors`, which parses PostgreSQL's binary COPY framing and
pgvector's `vector_send` wire format [upstream, not in /root/reference: format restated from the
PostgreSQL COPY documentation and pgvector's src/vector.c; no Postgres exists in this image, so the
parser is exercised against the writer below (round trip) -- parity with a live server is unpinned]:

    file   : "PGCOPY\\n\\377\\r\\n\\0" | int32 flags | int32 header-extension length | tuples | int16 -1
    tuple  : int16 field count | per field: int32 byte length (-1 = NULL) | data
    int4   : 4 bytes big-endian          int8: 8 bytes big-endian
    vector : int16 dim | int16 unused(0) | dim x float4 big-endian

Everything here is host-side byte shuffling with numpy; the vectors then go through
`HipIndex.add(rows, ids=...)` (ak_index_add) like any other ingest.
"""
from __future__ import annotations

import io
import struct
from typing import BinaryIO, Iterable, Iterator, Optional, Tuple

import numpy as np

SIGNATURE = b"PGCOPY\n\xff\r\n\x00"


def write_pgcopy_vectors(out: BinaryIO, ids: Iterable[int], vectors: np.ndarray, id_bytes: int = 4) -> None:
    """Produce what `COPY (SELECT id, embedding ...) TO STDOUT (FORMAT binary)` emits (tests / tooling)."""
    vectors = np.ascontiguousarray(vectors, dtype=np.float32)
    out.write(SIGNATURE + struct.pack(">ii", 0, 0))
    be = vectors.astype(">f4")
    for i, row_id in enumerate(ids):
        out.write(struct.pack(">h", 2))
        out.write(struct.pack(">i", id_bytes) + int(row_id).to_bytes(id_bytes, "big", signed=True))
        if vectors.shape[1] == 0 or np.isnan(vectors[i]).all():
            out.write(struct.pack(">i", -1))                       # NULL embedding
        else:
            payload = struct.pack(">hh", vectors.shape[1], 0) + be[i].tobytes()
            out.write(struct.pack(">i", len(payload)) + payload)
    out.write(struct.pack(">h", -1))


def _read_exact(f: BinaryIO, n: int) -> bytes:
    b = f.read(n)
    if len(b) != n:
        raise ValueError("truncated PGCOPY stream")
    return b


def iter_pgcopy_vectors(f: BinaryIO) -> Iterator[Tuple[int, Optional[np.ndarray]]]:
    if _read_exact(f, 11) != SIGNATURE:
        raise ValueError("not a PostgreSQL binary COPY stream")
    flags, ext = struct.unpack(">ii", _read_exact(f, 8))
    if flags & (1 << 16):
        raise ValueError("COPY stream carries OIDs; export without them")
    if ext:
        _read_exact(f, ext)
    while True:
        (nf,) = struct.unpack(">h", _read_exact(f, 2))
        if nf == -1:
            return
        if nf != 2:
            raise ValueError(f"expected 2 fields per tuple (id, embedding), got {nf}")
        (ln,) = struct.unpack(">i", _read_exact(f, 4))
        if ln not in (2, 4, 8):
            raise ValueError(f"id field must be int2/int4/int8, got {ln} bytes")
        row_id = int.from_bytes(_read_exact(f, ln), "big", signed=True)
        (ln,) = struct.unpack(">i", _read_exact(f, 4))
        if ln == -1:
            yield row_id, None
            continue
        body = _read_exact(f, ln)
        dim, unused = struct.unpack(">hh", body[:4])
        if unused != 0 or ln != 4 + 4 * dim:
            raise ValueError("malformed pgvector value")
        yield row_id, np.frombuffer(body, dtype=">f4", offset=4, count=dim).astype(np.float32)


def iter_pgcopy_blocks(f: BinaryIO, batch: int = 65536) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
    """(ids int64 [m], vectors float32 [m, D]) blocks of up to `batch` rows; NULL embeddings are skipped.
    Tuples of one table have one size (same id width, same D), so a run of them is a fixed-stride record array: runs
    are decoded with numpy in one pass (GB/s instead of the ~10 us per row of a Python loop -- 10M rows in seconds, not
    minutes); a NULL embedding or the trailer ends a run and is stepped over tuple by tuple."""
    if _read_exact(f, 11) != SIGNATURE:
        raise ValueError("not a PostgreSQL binary COPY stream")
    flags, ext = struct.unpack(">ii", _read_exact(f, 8))
    if flags & (1 << 16):
        raise ValueError("COPY stream carries OIDs; export without them")
    if ext:
        _read_exact(f, ext)
    buf = bytearray()
    eof = False

    def fill(n: int) -> bool:
        nonlocal eof
        while len(buf) < n and not eof:
            b = f.read(max(n - len(buf), 1 << 20))
            if not b:
                eof = True
            else:
                buf.extend(b)
        return len(buf) >= n

    rec = None          # (id_len, dim, numpy record dtype)
    while True:
        if not fill(2):
            raise ValueError("truncated PGCOPY stream")
        (nf,) = struct.unpack_from(">h", buf, 0)
        if nf == -1:
            return
        if nf != 2:
            raise ValueError(f"expected 2 fields per tuple (id, embedding), got {nf}")
        if not fill(6):
            raise ValueError("truncated PGCOPY stream")
        (idl,) = struct.unpack_from(">i", buf, 2)
        if idl not in (2, 4, 8):
            raise ValueError(f"id field must be int2/int4/int8, got {idl} bytes")
        if not fill(6 + idl + 4):
            raise ValueError("truncated PGCOPY stream")
        (ln,) = struct.unpack_from(">i", buf, 6 + idl)
        if ln == -1:                                   # NULL embedding: step over this tuple
            del buf[: 6 + idl + 4]
            continue
        if ln < 4 or (ln - 4) % 4:
            raise ValueError("malformed pgvector value")
        dim = (ln - 4) // 4
        if rec is None or rec[0] != idl or rec[1] != dim:
            if rec is not None and rec[1] != dim:
                raise ValueError(f"{dim}-d vector in a {rec[1]}-d column")
            rec = (idl, dim, np.dtype([("nf", ">i2"), ("l1", ">i4"), ("id", f">i{idl}"), ("l2", ">i4"), ("dim", ">i2"),
                                       ("unused", ">i2"), ("v", ">f4", (dim,))]))
        size = rec[2].itemsize
        if not fill(size):
            raise ValueError("truncated PGCOPY stream")
        fill(size * batch)                             # as many whole tuples as are there, up to one batch
        count = min(len(buf) // size, batch)
        arr = np.frombuffer(buf, dtype=rec[2], count=count)
        ok = (arr["nf"] == 2) & (arr["l1"] == idl) & (arr["l2"] == ln) & (arr["dim"] == dim) & (arr["unused"] == 0)
        good = count if ok.all() else int(np.argmin(ok))
        if good == 0:
            raise ValueError("malformed pgvector value")
        ids = arr["id"][:good].astype(np.int64)
        vecs = arr["v"][:good].astype(np.float32)
        del arr, ok                                    # release the buffer export before shrinking it
        del buf[: good * size]
        yield ids, vecs


def read_pgcopy_vectors(f: BinaryIO, dim: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """(ids int64 [n], vectors float32 [n, D]); rows with a NULL embedding are skipped."""
    ids, rows = [], []
    for bi, bv in iter_pgcopy_blocks(f):
        if dim is None:
            dim = bv.shape[1]
        if bv.shape[1] != dim:
            raise ValueError(f"row {int(bi[0])}: {bv.shape[1]}-d vector in a {dim}-d column")
        ids.append(bi)
        rows.append(bv)
    d = dim or 0
    if not rows:
        return np.zeros(0, np.int64), np.zeros((0, d), np.float32)
    return np.concatenate(ids), np.concatenate(rows)


def load_index_from_pgcopy(index, f: BinaryIO, batch: int = 65536) -> int:
    """Stream a COPY dump into a HipIndex (ids = document_chunks.id). Returns rows added."""
    total = 0
    for ids, rows in iter_pgcopy_blocks(f, batch):
        index.add(rows, ids=ids)
        total += len(ids)
    return total


def dump_index_to_pgcopy(index, slots: np.ndarray, ids: np.ndarray, out: BinaryIO, batch: int = 65536,
                         id_bytes: Optional[int] = None) -> None:
    """Inverse direction (e.g. seeding a fresh document_chunks table): stored rows as a COPY stream. id_bytes: 4 (the
    SERIAL id of init.sql:257) or 8 (BIGSERIAL); default = 4 unless an id needs more."""
    ids = np.asarray(ids, dtype=np.int64)
    if id_bytes is None:
        id_bytes = 8 if len(ids) and (int(ids.max()) > 2 ** 31 - 1 or int(ids.min()) < -2 ** 31) else 4
    buf = io.BytesIO()
    first = True
    for o in range(0, len(slots), batch):
        chunk = io.BytesIO()
        write_pgcopy_vectors(chunk, ids[o:o + batch], index.fetch(slots[o:o + batch]), id_bytes=id_bytes)
        b = chunk.getvalue()
        body = b[19:-2]                       # strip this chunk's header and trailer
        if first:
            buf.write(b[:19])
            first = False
        buf.write(body)
    if first:
        buf.write(SIGNATURE + struct.pack(">ii", 0, 0))
    buf.write(struct.pack(">h", -1))
    out.write(buf.getvalue())


# ---------------------------------------------------------------------------------------------------------------------
# Whole rows (round 3): an index loaded from (id, embedding) alone answers searches with ids nobody can turn into
# documents. The reference's table also holds chunk_text and metadata (init.sql:256-274) and its query joins `documents`
# (postgres_vectorstore.py:317-332), so the deployment path is
#     COPY (SELECT id, document_id, chunk_index, chunk_text, metadata, embedding FROM document_chunks) TO STDOUT (FORMAT binary)
#     COPY (SELECT id, resource_hash, display_name, source_type, url, is_deleted FROM documents)    TO STDOUT (FORMAT binary)
# -> ArchiHipVectorStore.load_from_pgcopy. Field formats [upstream PostgreSQL *_send functions, restated; pinned by the
# hand-written known answer in tests/test_host_logic_cpu.py]: int4 4 bytes big-endian; text raw UTF-8; jsonb one version
# byte (1) + the JSON text; bool one byte; vector as above; any field may be NULL (length -1).
# ---------------------------------------------------------------------------------------------------------------------
def _header(f: BinaryIO) -> None:
    if _read_exact(f, 11) != SIGNATURE:
        raise ValueError("not a PostgreSQL binary COPY stream")
    flags, ext = struct.unpack(">ii", _read_exact(f, 8))
    if flags & (1 << 16):
        raise ValueError("COPY stream carries OIDs; export without them")
    if ext:
        _read_exact(f, ext)


def _tuples(f: BinaryIO, nfields: int, chunk: int = 8 << 20) -> Iterator[list]:
    """Field payloads (memoryview | None) of every tuple of a binary COPY stream (variable-length rows: one pass in Python,
    ~3 us per row). The stream is read through a refillable buffer of `chunk` bytes -- a 10M x 768 column is 31 GB of
    embeddings alone, and the first version read the whole stream before it yielded a row."""
    _header(f)
    buf, o = b"", 0

    def need(k: int) -> bool:                 # make buf[o : o + k] available; False at the end of the stream
        nonlocal buf, o
        while len(buf) - o < k:
            more = f.read(max(chunk, k))
            if not more:
                return False
            buf, o = buf[o:] + more, 0        # views handed out earlier keep the old bytes object alive
        return True

    while True:
        if not need(2):
            raise ValueError("truncated PGCOPY stream")
        (nf,) = struct.unpack_from(">h", buf, o)
        o += 2
        if nf == -1:
            return
        if nf != nfields:
            raise ValueError(f"expected {nfields} fields per tuple, got {nf}")
        row = []
        for _ in range(nf):
            if not need(4):
                raise ValueError("truncated PGCOPY stream")
            (ln,) = struct.unpack_from(">i", buf, o)
            o += 4
            if ln == -1:
                row.append(None)
                continue
            if ln < 0 or not need(ln):
                raise ValueError("truncated PGCOPY stream")
            row.append(memoryview(buf)[o:o + ln])
            o += ln
        yield row


def _int(v) -> Optional[int]:
    return None if v is None else int.from_bytes(v, "big", signed=True)


def _jsonb(v):
    if v is None:
        return None
    b = bytes(v)
    if not b or b[0] != 1:
        raise ValueError("unsupported jsonb wire version")
    import json
    return json.loads(b[1:].decode("utf-8"))


def iter_pgcopy_chunks(f: BinaryIO, batch: int = 65536, own=None) -> Iterator[dict]:
    """Blocks {"ids" int64[m], "document_ids" list, "chunk_index" int64[m], "text_bytes" list[bytes] (UTF-8 as stored),
    "metadata" list[dict|None], "meta_json" list[bytes|None], "vectors" float32[m,D]} of (id, document_id, chunk_index, chunk_text, metadata, embedding) tuples; tuples with a NULL
    embedding are skipped (the reference's scan never returns them: `<=>` of NULL is NULL and sorts last / is filtered).
    own: optional predicate on the row id -- a rank of a row-sharded store decodes only the vectors of ITS rows (the others
    come out as zero rows, block["own"] says which are real): the vector field is nine tenths of a tuple's bytes."""
    cur: dict = {"ids": [], "document_ids": [], "chunk_index": [], "metadata": [], "vectors": [], "text_bytes": [], "meta_json": []}
    dim = None

    def flush():
        out = {"ids": np.asarray(cur["ids"], np.int64), "document_ids": list(cur["document_ids"]),
               "chunk_index": np.asarray(cur["chunk_index"], np.int64), "metadata": list(cur["metadata"]),
               # the stored UTF-8 / JSON text as it came (ChunkTable keeps exactly these bytes: no decode -> encode, no
               # loads -> dumps round trip on the bulk-load path)
               "text_bytes": list(cur["text_bytes"]), "meta_json": list(cur["meta_json"]),
               "vectors": None, "own": None}
        real = [i for i, v in enumerate(cur["vectors"]) if v is not None]
        if own is None:                       # (with a predicate the mask is ALWAYS produced: every rank must take the same path)
            out["vectors"] = (np.frombuffer(b"".join(cur["vectors"]), dtype=">f4").reshape(len(cur["vectors"]), dim).astype(np.float32)
                              if cur["vectors"] else np.zeros((0, dim or 0), np.float32))
        else:
            out["vectors"] = np.zeros((len(cur["vectors"]), dim), np.float32)
            if real:
                out["vectors"][real] = np.frombuffer(b"".join(cur["vectors"][i] for i in real), dtype=">f4").reshape(len(real), dim)
            out["own"] = np.zeros(len(cur["vectors"]), bool)
            out["own"][real] = True
        for v in cur.values():
            v.clear()
        return out

    for rid, doc, cidx, text, meta, emb in _tuples(f, 6):
        if emb is None:
            continue
        d, unused = struct.unpack_from(">hh", emb, 0)
        if unused != 0 or len(emb) != 4 + 4 * d:
            raise ValueError("malformed pgvector value")
        if dim is None:
            dim = d
        elif d != dim:
            raise ValueError(f"{d}-d vector in a {dim}-d column")
        cur["ids"].append(_int(rid))
        cur["document_ids"].append(_int(doc))
        cur["chunk_index"].append(_int(cidx) or 0)
        tb = b"" if text is None else bytes(text)
        cur["text_bytes"].append(tb)
        md = _jsonb(meta)
        cur["metadata"].append(md)
        cur["meta_json"].append(None if meta is None else bytes(meta[1:]))
        cur["vectors"].append(bytes(emb[4:]) if own is None or own(cur["ids"][-1]) else None)
        if len(cur["ids"]) >= batch:
            yield flush()
    if cur["ids"]:
        yield flush()


def read_pgcopy_documents(f: BinaryIO) -> list:
    """[{"id", "resource_hash", "display_name", "source_type", "url", "is_deleted"}] of the `documents` columns the
    retrieval query joins (postgres_vectorstore.py:323-326) and filters on (:305-308)."""
    out = []
    for rid, rhash, name, stype, url, deleted in _tuples(f, 6):
        s = lambda v: None if v is None else bytes(v).decode("utf-8")      # noqa: E731
        out.append({"id": _int(rid), "resource_hash": s(rhash), "display_name": s(name), "source_type": s(stype),
                    "url": s(url), "is_deleted": bool(deleted is not None and bytes(deleted) != b"\x00")})
    return out


def read_pgcopy_ids(f: BinaryIO) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """(ids int64 [n], versions int64 [n] | None) of a one- or two-column stream
        COPY (SELECT id [, xmin::text::bigint] FROM document_chunks WHERE ...) TO STDOUT (FORMAT binary)
    -- the cheap whole-collection listing ArchiHipVectorStore.refresh_from_pgcopy reconciles against (12 or 24 bytes per
    row; 10M rows = 240 MB). Integer fields of 2, 4 or 8 bytes (xid8 / bigint / int4); rows of one stream have one width,
    so the stream is decoded as a fixed-stride record array in one numpy pass. NULLs are not allowed."""
    _header(f)
    body = f.read()
    if len(body) < 2:
        raise ValueError("truncated PGCOPY stream")
    (nf,) = struct.unpack_from(">h", body, 0)
    if nf == -1:
        return np.zeros(0, np.int64), None
    if nf not in (1, 2):
        raise ValueError(f"expected 1 or 2 fields per tuple (id [, version]), got {nf}")
    if len(body) < 6:
        raise ValueError("truncated PGCOPY stream")
    (l1,) = struct.unpack_from(">i", body, 2)
    if l1 not in (2, 4, 8):
        raise ValueError(f"id field must be int2/int4/int8, got {l1} bytes")
    fields = [("nf", ">i2"), ("l1", ">i4"), ("id", f">i{l1}")]
    l2 = None
    if nf == 2:
        if len(body) < 6 + l1 + 4:
            raise ValueError("truncated PGCOPY stream")
        (l2,) = struct.unpack_from(">i", body, 6 + l1)
        if l2 not in (4, 8):
            raise ValueError(f"version field must be 4 or 8 bytes (xid / bigint), got {l2}")
        fields += [("l2", ">i4"), ("ver", f">i{l2}")]
    rec = np.dtype(fields)
    n, rest = divmod(len(body) - 2, rec.itemsize)
    if rest or struct.unpack_from(">h", body, n * rec.itemsize)[0] != -1:
        raise ValueError("malformed id stream (rows of several widths, a NULL, or a missing trailer)")
    arr = np.frombuffer(body, dtype=rec, count=n)
    ok = (arr["nf"] == nf) & (arr["l1"] == l1)
    if l2 is not None:
        ok &= arr["l2"] == l2
    if not ok.all():
        raise ValueError("malformed id stream (rows of several widths or a NULL)")
    return arr["id"].astype(np.int64), (arr["ver"].astype(np.int64) if l2 is not None else None)


def write_pgcopy_ids(out: BinaryIO, ids: Iterable[int], versions: Optional[Iterable[int]] = None, id_bytes: int = 4) -> None:
    """What the listing query above emits (tests / tooling); versions as bigint."""
    out.write(SIGNATURE + struct.pack(">ii", 0, 0))
    if versions is None:
        for rid in ids:
            out.write(struct.pack(">hi", 1, id_bytes) + int(rid).to_bytes(id_bytes, "big", signed=True))
    else:
        for rid, ver in zip(ids, versions):
            out.write(struct.pack(">hi", 2, id_bytes) + int(rid).to_bytes(id_bytes, "big", signed=True) + struct.pack(">iq", 8, int(ver)))
    out.write(struct.pack(">h", -1))


def _put(out: BinaryIO, payload: Optional[bytes]) -> None:
    out.write(struct.pack(">i", -1) if payload is None else struct.pack(">i", len(payload)) + payload)


def write_pgcopy_chunks(out: BinaryIO, rows: Iterable[tuple]) -> None:
    """rows: (id, document_id | None, chunk_index, chunk_text, metadata dict | None, embedding float32[D] | None) -> what the
    six-column COPY above emits (tests / tooling; also the way to seed a fresh document_chunks table from this backend)."""
    import json
    out.write(SIGNATURE + struct.pack(">ii", 0, 0))
    for rid, doc, cidx, text, meta, emb in rows:
        out.write(struct.pack(">h", 6))
        _put(out, int(rid).to_bytes(4, "big", signed=True))
        _put(out, None if doc is None else int(doc).to_bytes(4, "big", signed=True))
        _put(out, int(cidx).to_bytes(4, "big", signed=True))
        _put(out, None if text is None else text.encode("utf-8", "surrogatepass"))      # the table stores text with the same handler
        _put(out, None if meta is None else b"\x01" + json.dumps(meta).encode("utf-8", "surrogatepass"))
        if emb is None:
            _put(out, None)
        else:
            v = np.ascontiguousarray(emb, dtype=np.float32)
            _put(out, struct.pack(">hh", v.shape[0], 0) + v.astype(">f4").tobytes())
    out.write(struct.pack(">h", -1))


def write_pgcopy_documents(out: BinaryIO, docs: Iterable[dict]) -> None:
    out.write(SIGNATURE + struct.pack(">ii", 0, 0))
    for d in docs:
        out.write(struct.pack(">h", 6))
        _put(out, int(d["id"]).to_bytes(4, "big", signed=True))
        for k in ("resource_hash", "display_name", "source_type", "url"):
            _put(out, None if d.get(k) is None else str(d[k]).encode("utf-8"))
        _put(out, b"\x01" if d.get("is_deleted") else b"\x00")
    out.write(struct.pack(">h", -1))
