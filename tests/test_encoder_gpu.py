"""GPU parity tests of the HIP encoder (through the C ABI) against the fixtures produced by
transformers.BertModel and against the torch-fp32 oracle on other shapes.

Tolerance (stated by the build, north star only fixes 1e-5 for retrieval scores): the HIP encoder
computes its GEMMs on bf16 MFMA with fp32 accumulation and keeps activations between GEMMs in
bf16, so embeddings are compared by cosine similarity >= 1 - 2e-3 and max|diff| <= 2e-2 on unit
vectors against the fp32 reference of the same weights."""
import glob
import os

import numpy as np
import pytest

from oracle import encoder_oracle as eo

pytestmark = pytest.mark.gpu
FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "encoder_*.npz")))
COS_TOL, ABS_TOL = 2e-3, 2e-2


def _encoder(hip, shape, seed=7):
    from archi_amd.encoder import HipEncoder
    vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
    w = eo.synth_weights(shape, seed=seed)
    return HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0), w


def _check(got, want):
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert cos.min() >= 1 - COS_TOL, f"min cosine {cos.min()}"
    assert np.abs(got - want).max() <= ABS_TOL, f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p) for p in FIX])
def test_encoder_matches_hf_fixture(hip, path):
    f = np.load(path)
    enc, _ = _encoder(hip, str(f["shape"]), int(f["weight_seed"]))
    got = enc.forward(f["ids"], f["mask"], pooling=str(f["pooling"]), normalise=True).cpu().numpy()
    _check(got, f["expected"])
    enc.close()


@pytest.mark.parametrize("B,S", [(1, 32), (7, 96), (3, 500), (40, 64)])
def test_encoder_matches_oracle_other_shapes(hip, B, S):
    enc, w = _encoder(hip, "minilm-l6")
    ids, mask = eo.synth_tokens(B, S, seed=B * 1000 + S)
    for pooling in ("mean", "cls"):
        got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
        _check(got, eo.forward("minilm-l6", w, ids, mask, pooling=pooling))
    raw = enc.forward(ids, mask, pooling="mean", normalise=False).cpu().numpy()
    want = eo.forward("minilm-l6", w, ids, mask, pooling="mean", normalise=False)
    assert np.abs(raw - want).max() <= 5e-2 * max(1.0, np.abs(want).max())
    enc.close()


def test_embeddings_provider_surface(hip):
    from archi_amd.embeddings import ArchiHipEmbeddings
    emb = ArchiHipEmbeddings(model_name="sentence-transformers/all-MiniLM-L6-v2",
                             model_kwargs={"device": "cuda", "synthetic_seed": 0},
                             encode_kwargs={"normalize_embeddings": True})
    texts = ["First test document about physics", "Second test\ndocument about chemistry", "x"]
    out = emb.embed_documents(texts)
    # what the reference's own test asserts (tests/unit/test_ingestion_pipeline_isolation.py:141-142)
    assert len(out) == 3 and all(len(v) == 384 for v in out) and isinstance(out[0][0], float)
    q = emb.embed_query(texts[0])
    assert np.allclose(q, out[0], atol=1e-6) and abs(np.linalg.norm(q) - 1.0) < 1e-4
    assert emb.embed_documents([]) == []
    # batching is order-preserving whatever the length mix
    many = [("word " * (i % 37 + 1)).strip() for i in range(70)]
    a = np.array(emb.embed_documents(many)); b = np.array([emb.embed_query(t) for t in many])
    assert np.abs(a - b).max() < 5e-3
    with pytest.raises(FileNotFoundError):
        ArchiHipEmbeddings(model_name="sentence-transformers/all-MiniLM-L6-v2")


def test_l2_normalize_kernel(hip):
    import ctypes
    import torch
    from oracle import knn_oracle as ko
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1000, 384)).astype(np.float32); x[7] = 0
    t = torch.from_numpy(x).cuda()
    from archi_amd import _lib
    _lib.check(hip.ak_l2_normalize_dev(ctypes.c_void_p(t.data_ptr()), 1000, 384,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "l2")
    torch.cuda.synchronize()
    assert np.allclose(t.cpu().numpy(), ko.l2_normalize(x), atol=1e-6)
