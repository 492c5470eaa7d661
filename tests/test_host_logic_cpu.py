"""CPU suite: host-side pieces of the path that need no GPU."""
import os

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


def test_config_plugin_replays_the_reference_resolver_and_dimension_bookkeeping():
    """N4 against the reference itself: tests/golden/make_reference_fixtures.py RAN ConfigService._resolve_embedding_classes
    (src/utils/config_service.py:470-496) and TemplateManager._render_postgres_init (templates_manager.py:393-431, the
    `vector(D)` of init.sql:266) and recorded their output; the plug-in reproduces both."""
    import json
    import os
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_wrapper.json")))["config"]
    # the reference's table (two stand-in classes of the same names) merged with this backend's entry
    ref_table = {"HuggingFaceEmbeddings": type("HuggingFaceEmbeddings", (), {}), "OpenAIEmbeddings": type("OpenAIEmbeddings", (), {})}
    got = cp.resolve_embedding_classes(G["embedding_class_map"], mapping=ref_table)       # reference table only
    plain = {k: {kk: ({"__class__": vv.__name__} if isinstance(vv, type) else vv) for kk, vv in e.items()} for k, e in got.items()}
    assert plain == G["resolved"]
    assert cp.resolve_embedding_classes({}) == G["resolved_empty"] == {}
    merged = cp.resolve_embedding_classes(G["embedding_class_map"], mapping={**ref_table, **cp.embedding_mapping()})
    from archi_amd.embeddings import ArchiHipEmbeddings
    assert merged["ArchiHipEmbeddings"]["class"] is ArchiHipEmbeddings              # the one entry the merge adds
    for name, e in merged.items():
        if name != "ArchiHipEmbeddings":
            assert e == got[name]
    for case in G["init_sql_dimensions"]:
        dm = {"embedding_class_map": G["embedding_class_map"] if case["uses_class_map"] else {}}
        if case["embedding_name"] is not None:
            dm["embedding_name"] = case["embedding_name"]
        assert cp.init_sql_dimensions(dm) == case["dimensions"], case
    # the model-name table fills in what the reference needs spelled out for a 768-d model
    assert cp.embedding_dimensions({"kwargs": {"model_name": "BAAI/bge-base-en"}}) == 768


# A PostgreSQL binary COPY stream of `COPY (SELECT id, embedding FROM document_chunks) TO STDOUT (FORMAT binary)` written
# out BY HAND from the published formats, independently of archi_amd.pgbridge's writer: PostgreSQL docs, COPY, "Binary
# Format" (11-byte signature, int32 flags, int32 header-extension length; per tuple int16 field count, per field int32
# byte length or -1 for NULL, big-endian; trailer int16 -1) and pgvector's vector_send (int16 dim, int16 unused = 0, dim x
# float4 big-endian). Three tuples: (id int4 7, [1.0, -2.5, 0.15625]), (id 8, NULL embedding), (id 9, [0, FLT_MAX, -0.0]).
PGCOPY_KNOWN_ANSWER = bytes.fromhex(
    "5047434f50590aff0d0a00" "00000000" "00000000"
    "0002" "00000004" "00000007" "00000010" "0003" "0000" "3f800000" "c0200000" "3e200000"
    "0002" "00000004" "00000008" "ffffffff"
    "0002" "00000004" "00000009" "00000010" "0003" "0000" "00000000" "7f7fffff" "80000000"
    "ffff")


def test_pgcopy_reader_on_a_hand_written_stream():
    import io
    from archi_amd import pgbridge as pb
    tuples = list(pb.iter_pgcopy_vectors(io.BytesIO(PGCOPY_KNOWN_ANSWER)))
    assert [t[0] for t in tuples] == [7, 8, 9] and tuples[1][1] is None
    assert tuples[0][1].tolist() == [1.0, -2.5, 0.15625]
    assert tuples[2][1].view(np.uint32).tolist() == [0x00000000, 0x7f7fffff, 0x80000000]       # bit patterns survive
    ids, vec = pb.read_pgcopy_vectors(io.BytesIO(PGCOPY_KNOWN_ANSWER))
    assert ids.tolist() == [7, 9] and vec.shape == (2, 3) and vec[0].tolist() == [1.0, -2.5, 0.15625]
    out = io.BytesIO()                                  # and the writer emits exactly these bytes for the same table
    pb.write_pgcopy_vectors(out, [7, 8, 9], np.array([[1.0, -2.5, 0.15625], [np.nan] * 3, [0.0, 3.4028235e38, -0.0]], np.float32))
    assert out.getvalue() == PGCOPY_KNOWN_ANSWER


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


def test_bench_gpus_n_without_launcher_refuses_when_the_node_has_fewer_gpus():
    """bench.py --gpus N starts its own ranks; with fewer than N GPUs it must exit non-zero, never run fewer ranks silently."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"refusing" in p.stderr


def test_lazy_layernorm_identities():
    """The algebra csrc/gemm.hip's lazy LayerNorm rests on (hidden-768 path; the kernels are held to the oracle on the GPU): rows
    travel as r~ = gamma (.) r with r's per-token (mean, 1 / std) beside them, and
        LN(r) W^T + b  ==  rstd (r~ W^T - mu (W gamma)) + (b + W beta)        the GEMMs that read such rows (weights unchanged)
        LN(r)_k        ==  rstd (r~_k - mu gamma_k) + beta_k                   the residual (no division: gamma may be 0 or negative)
    with the statistics recovered from per-slice (sum, sum of squares) partials in float32."""
    rng = np.random.default_rng(4)
    T, H, N, eps = 37, 768, 96, 1e-12
    r = (rng.normal(0.3, 1.7, size=(T, H))).astype(np.float64)
    gamma = rng.uniform(-0.5, 2.5, size=H); gamma[:7] = 0.0
    beta = rng.normal(0, 0.5, size=H)
    W = rng.normal(0, 0.04, size=(N, H)); b = rng.normal(0, 0.02, size=N)
    mu = r.mean(1, keepdims=True); var = ((r - mu) ** 2).mean(1, keepdims=True); rstd = 1.0 / np.sqrt(var + eps)
    ln = (r - mu) * rstd * gamma + beta
    rt = r * gamma
    assert np.allclose(rstd * (rt @ W.T - mu * (W @ gamma)) + (b + W @ beta), ln @ W.T + b, rtol=0, atol=1e-10)
    assert np.allclose(rstd * (rt - mu * gamma) + beta, ln, rtol=0, atol=1e-12)
    # statistics from six 128-feature slice partials, float32, one pass (sum, sum of squares), as k_ln_finalize forms them
    r32 = r.astype(np.float32)
    parts = [(r32[:, s:s + 128].sum(1, dtype=np.float32), (r32[:, s:s + 128] ** 2).sum(1, dtype=np.float32)) for s in range(0, H, 128)]
    s0 = np.sum([p[0] for p in parts], axis=0, dtype=np.float32); s1 = np.sum([p[1] for p in parts], axis=0, dtype=np.float32)
    mu32 = s0 / np.float32(H); var32 = np.maximum(s1 / np.float32(H) - mu32 * mu32, np.float32(0))
    assert np.allclose(mu32, mu[:, 0], rtol=0, atol=1e-5) and np.allclose(1.0 / np.sqrt(var32 + eps), rstd[:, 0], rtol=2e-5)
