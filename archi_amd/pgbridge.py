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
