"""CPU suite: host-side pieces of the path that need no GPU."""
import numpy as np
import pytest

from archi_amd import config_plugin as cp


def test_embedding_text_round_trip_is_lossless():
    """a4: the reference ships vectors to pgvector as text, "[" + ",".join(str(x)) + "]" then
    ::vector (float4) (postgres_vectorstore.py:313-314). For float32-origin values that round trip
    is exact, so the store may take the list[float] straight to float32."""
    rng = np.random.default_rng(0)
    v32 = (rng.standard_normal(5000) * np.exp(rng.uniform(-20, 20, 5000))).astype(np.float32)
    as_python = [float(x) for x in v32]                      # what embed_query returns
    text = "[" + ",".join(str(x) for x in as_python) + "]"
    back = np.array([float(t) for t in text.strip("[]").split(",")], dtype=np.float64).astype(np.float32)
    assert np.array_equal(back, v32)
    assert str(float(np.float32(0.1))) == "0.10000000149011612"


def test_config_plugin_resolves_like_the_reference():
    m = {"ArchiHipEmbeddings": {"class": "ArchiHipEmbeddings", "kwargs": {"model_name": "BAAI/bge-base-en"}},
         "OpenAIEmbeddings": {"class": "OpenAIEmbeddings", "kwargs": {"model": "text-embedding-3-small"}},
         "Bare": None}
    r = cp.resolve_embedding_classes(m)
    from archi_amd.embeddings import ArchiHipEmbeddings
    assert r["ArchiHipEmbeddings"]["class"] is ArchiHipEmbeddings
    assert r["OpenAIEmbeddings"]["class"] == "OpenAIEmbeddings" and r["Bare"] == {}
    assert cp.resolve_embedding_classes({}) == {}
    assert cp.embedding_dimensions(r["ArchiHipEmbeddings"]) == 768
    assert cp.embedding_dimensions({"dimensions": 1024}) == 1024
    assert cp.map_distance_metric("ip") == "inner_product" and cp.map_distance_metric("l2") == "l2"
    with pytest.raises(ValueError, match="is not supported"):
        cp.map_distance_metric("manhattan")


def test_pgcopy_bridge_round_trip():
    """N2: PostgreSQL binary COPY framing + pgvector vector_send format, writer <-> parser."""
    import io
    from archi_amd import pgbridge as pb
    from tests.fake_index import OracleIndex
    rng = np.random.default_rng(4)
    vec = rng.standard_normal((300, 48)).astype(np.float32)
    vec[17] = np.nan                                     # a NULL embedding row
    ids = (np.arange(300) * 5 + 7)
    buf = io.BytesIO()
    pb.write_pgcopy_vectors(buf, ids, vec)
    raw = buf.getvalue()
    assert raw.startswith(b"PGCOPY\n\xff\r\n\x00") and raw.endswith(b"\xff\xff")
    # one tuple on the wire: int16 2 | int32 4 | id | int32 4+4*48 | int16 48 | int16 0 | 48 big-endian floats
    assert raw[19:21] == b"\x00\x02" and raw[21:25] == b"\x00\x00\x00\x04" and raw[29:33] == (4 + 4 * 48).to_bytes(4, "big")
    gi, gv = pb.read_pgcopy_vectors(io.BytesIO(raw))
    keep = np.arange(300) != 17
    assert np.array_equal(gi, ids[keep]) and np.array_equal(gv, vec[keep])
    ix = OracleIndex(48, 1000, dtype="f32")
    assert pb.load_index_from_pgcopy(ix, io.BytesIO(raw), batch=64) == 299 and ix.count() == 299
    with pytest.raises(ValueError):
        pb.read_pgcopy_vectors(io.BytesIO(b"not a copy stream....."))
    with pytest.raises(ValueError):
        pb.read_pgcopy_vectors(io.BytesIO(raw[:200]))


def test_pgcopy_block_reader_equals_tuple_reader():
    """The vectorised run decoder against the tuple-by-tuple parser: NULLs first / adjacent / last, int8 ids, a stream
    handed over in small pieces (pipe-like reads), and a size where the per-row loop would take minutes."""
    import io
    import time
    from archi_amd import pgbridge as pb

    class Dribble(io.RawIOBase):                       # read() returns at most 1000 bytes at a time
        def __init__(self, data):
            self.b = io.BytesIO(data)

        def read(self, n=-1):
            return self.b.read(min(n, 1000) if n and n > 0 else 1000)

    rng = np.random.default_rng(9)
    vec = rng.standard_normal((500, 24)).astype(np.float32)
    for i in (0, 1, 250, 251, 252, 499):
        vec[i] = np.nan
    ids = rng.permutation(10**12 + np.arange(500))
    buf = io.BytesIO()
    pb.write_pgcopy_vectors(buf, ids, vec, id_bytes=8)
    raw = buf.getvalue()
    slow = [(i, v) for i, v in pb.iter_pgcopy_vectors(io.BytesIO(raw)) if v is not None]
    for src in (io.BytesIO(raw), Dribble(raw)):
        blocks = list(pb.iter_pgcopy_blocks(src, batch=97))
        gi, gv = np.concatenate([b[0] for b in blocks]), np.concatenate([b[1] for b in blocks])
        assert all(len(b[0]) <= 97 for b in blocks)
        assert np.array_equal(gi, [i for i, _ in slow]) and np.array_equal(gv, np.stack([v for _, v in slow]))
    only_null = io.BytesIO()
    pb.write_pgcopy_vectors(only_null, [1, 2], np.full((2, 8), np.nan, np.float32))
    assert pb.read_pgcopy_vectors(io.BytesIO(only_null.getvalue()))[0].size == 0
    big = rng.standard_normal((50_000, 384)).astype(np.float32)
    out = io.BytesIO()
    be = big.astype(">f4")
    rec = np.zeros(50_000, dtype=[("nf", ">i2"), ("l1", ">i4"), ("id", ">i4"), ("l2", ">i4"), ("dim", ">i2"), ("u", ">i2"), ("v", ">f4", (384,))])
    rec["nf"], rec["l1"], rec["id"], rec["l2"], rec["dim"], rec["v"] = 2, 4, np.arange(50_000), 4 + 4 * 384, 384, be
    out.write(pb.SIGNATURE + b"\x00" * 8 + rec.tobytes() + b"\xff\xff")
    t0 = time.perf_counter()
    gi, gv = pb.read_pgcopy_vectors(io.BytesIO(out.getvalue()))
    assert time.perf_counter() - t0 < 5.0
    assert np.array_equal(gi, np.arange(50_000)) and np.array_equal(gv, big)
