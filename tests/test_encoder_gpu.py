"""GPU parity tests of the HIP encoder (through the C ABI) against the fixtures produced by
transformers.BertModel and against the torch-fp32 oracle on other shapes.

Tolerance (stated by the build, north star only fixes 1e-5 for retrieval scores): the HIP encoder
computes its GEMMs on bf16 MFMA with fp32 accumulation and keeps activations between GEMMs in
bf16, so embeddings are compared by cosine similarity >= 1 - 1e-4 and max|diff| <= 2e-3 on unit
vectors against the fp32 reference of the same weights (measured on these cases: min cosine 0.999997, max|diff| ~1e-3;
a regression ten times worse than that fails). These bounds hold for the seeds of this file at BOTH hidden sizes. The stated
bf16 tolerance of the 12-layer hidden-768 shape (bge-base) on ARBITRARY weight seeds is wider, 1 - cos <= 3e-4 and max|diff| <=
3e-3: scripts/gpu_soak_enc.py holds random seeds, batch sizes and masks to it in both residual modes
(profiles/r05_soak_encoder_*); precision="f32" is the mode held to 1e-5 everywhere."""
import glob
import os

import numpy as np
import pytest

from oracle import encoder_oracle as eo

pytestmark = pytest.mark.gpu
FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "encoder_*.npz")))
COS_TOL, ABS_TOL = 1e-4, 2e-3


def _encoder(hip, shape, seed=7, residual="bf16", precision="bf16"):
    from archi_amd.encoder import HipEncoder
    vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
    w = eo.synth_weights(shape, seed=seed)
    return HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0, residual=residual, precision=precision), w


def _check(got, want):
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert cos.min() >= 1 - COS_TOL, f"min cosine {cos.min()}"
    assert np.abs(got - want).max() <= ABS_TOL, f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("residual", ["bf16", "f32"])      # residual stream kept in bf16 (default) or fp32 between layers
@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p) for p in FIX])
def test_encoder_matches_hf_fixture(hip, path, residual):
    f = np.load(path)
    enc, _ = _encoder(hip, str(f["shape"]), int(f["weight_seed"]), residual)
    got = enc.forward(f["ids"], f["mask"], pooling=str(f["pooling"]), normalise=True).cpu().numpy()
    _check(got, f["expected"])
    enc.close()


# ---- fp32 parity mode (AkBertConfig.precision = 1): float32 weights and arithmetic throughout. This is the number that
# ties a1 to the reference's CPU embedder (sentence-transformers on torch fp32, manager.py:373): max|diff| <= 1e-5 on unit
# vectors against transformers.BertModel -- the bf16 MFMA path cannot be held to that.
F32_ABS_TOL, F32_COS_TOL = 1e-5, 1e-9


def _check_f32(got, want):
    cos = (got.astype(np.float64) * want).sum(1) / (np.linalg.norm(got.astype(np.float64), axis=1) * np.linalg.norm(want.astype(np.float64), axis=1))
    assert np.abs(got - want).max() <= F32_ABS_TOL, f"max abs diff {np.abs(got - want).max()}"
    assert cos.min() >= 1 - 1e-6, f"min cosine {cos.min()}"


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p) for p in FIX])
def test_fp32_parity_mode_matches_hf_fixture(hip, path, precision):
    """precision="f32": exact float32 on v_mfma_f32_32x32x2_f32; "bf16x3": the split-bf16 parity mode (round 6) -- float32
    weights, every GEMM as hi.hi + lo.hi + hi.lo on the bf16 matrix cores. Both are held to the SAME bar: 1e-5 on unit vectors
    against transformers.BertModel."""
    f = np.load(path)
    enc, _ = _encoder(hip, str(f["shape"]), int(f["weight_seed"]), precision=precision)
    got = enc.forward(f["ids"], f["mask"], pooling=str(f["pooling"]), normalise=True).cpu().numpy()
    _check_f32(got, f["expected"])
    enc.close()


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("shape,B,S", [("minilm-l6", 5, 96), ("minilm-l6", 2, 500), ("tiny", 9, 33), ("bge-base", 3, 160)])
def test_fp32_parity_mode_matches_oracle_other_shapes(hip, shape, B, S, precision):
    """ragged masks, both poolings, un-normalised output: against the torch-fp32 oracle"""
    enc, w = _encoder(hip, shape, precision=precision)
    ids, mask = eo.synth_tokens(B, S, seed=B * 100 + S, vocab=eo.SHAPES[shape][0])
    for pooling in ("mean", "cls"):
        got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
        _check_f32(got, eo.forward(shape, w, ids, mask, pooling=pooling))
    raw = enc.forward(ids, mask, pooling="mean", normalise=False).cpu().numpy()
    want = eo.forward(shape, w, ids, mask, pooling="mean", normalise=False)
    assert np.abs(raw - want).max() <= 1e-5 * max(1.0, np.linalg.norm(want, axis=1).max())
    enc.close()


@pytest.mark.parametrize("B,S", [(1, 32), (7, 96), (3, 500), (40, 64)])
def test_encoder_matches_oracle_other_shapes(hip, B, S):
    enc, w = _encoder(hip, "minilm-l6", residual="f32" if B == 7 else "bf16")
    ids, mask = eo.synth_tokens(B, S, seed=B * 1000 + S)
    for pooling in ("mean", "cls"):
        got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
        _check(got, eo.forward("minilm-l6", w, ids, mask, pooling=pooling))
    raw = enc.forward(ids, mask, pooling="mean", normalise=False).cpu().numpy()
    want = eo.forward("minilm-l6", w, ids, mask, pooling="mean", normalise=False)
    assert np.abs(raw - want).max() <= 5e-2 * max(1.0, np.abs(want).max())
    enc.close()


def test_split_bf16_mode_with_full_precision_weights(hip):
    """The synthetic weights of the other tests are bf16-exact (lo = 0 for every weight): here they carry full float32
    mantissas, so the weight split's lo half matters, and the batch is large enough for several 128-row GEMM tiles plus a
    ragged last one. Against the float32 oracle on the same weights: 1e-5."""
    rng = np.random.default_rng(5)
    shape = "minilm-l6"
    vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
    w = eo.synth_weights(shape, seed=9)
    w = {k: (v * (1.0 + 1e-3 * rng.standard_normal(v.shape))).astype(np.float32) if v.ndim == 2 else v for k, v in w.items()}
    from archi_amd.encoder import HipEncoder
    ids, mask = eo.synth_tokens(11, 96, seed=77, vocab=vocab)           # 1056 tokens: 8 full row tiles + 32 rows
    want = eo.forward(shape, w, ids, mask, pooling="mean")
    for precision in ("bf16x3", "f32"):
        enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0, precision=precision)
        got = enc.forward(ids, mask, pooling="mean", normalise=True).cpu().numpy()
        _check_f32(got, want)
        enc.close()


def test_split_bf16_mode_large_batch_on_the_gemm_tiles(hip):
    """Large batches of the split-bf16 mode run on gemm.hip's tiles (from 20 480 tokens at hidden 384, 16 384 at hidden 768; activations
    as bf16 [hi | lo] rows between the launches, rows padded to whole 256-token tiles): ragged batches just above the switch with a
    token count off the 256 grid, full-mantissa weights, both head sizes, against the float32 oracle at the parity bar on sampled rows
    (a row's embedding does not depend on its neighbours)."""
    rng = np.random.default_rng(6)
    from archi_amd.encoder import HipEncoder
    for shape, B, S in (("minilm-l6", 103, 200), ("bge-base", 35, 480)):          # 20 600 / 16 800 tokens
        vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
        w = eo.synth_weights(shape, seed=13)
        w = {k: (v * (1.0 + 1e-3 * rng.standard_normal(v.shape))).astype(np.float32) if v.ndim == 2 else v for k, v in w.items()}
        ids, mask = eo.synth_tokens(B, S, seed=B + S, vocab=vocab)
        pick = np.array([0, 1, B // 2, B - 2, B - 1])
        enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0, precision="bf16x3")
        for pooling in ("mean", "cls"):
            got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
            assert np.isfinite(got).all()
            _check_f32(got[pick], eo.forward(shape, w, ids[pick], mask[pick], pooling=pooling))
        small = enc.forward(ids[:2], mask[:2], pooling="cls", normalise=True).cpu().numpy()      # the same encoder, a k3_gemm-sized batch
        _check_f32(small, got[:2])
        enc.close()


@pytest.mark.parametrize("precision", ["bf16", "f32", "bf16x3"])
def test_forward_lens_equals_the_mask_entry_point_bit_for_bit(hip, precision):
    """ak_encoder_forward_lens (right-padded rows given by their lengths, the provider's tile layout [rows, S + 1] with the length
    in column S; the mask is laid out by the library, rows land at the caller's offset of one result buffer) against
    ak_encoder_forward on the explicit 0 / 1 mask: the same kernels on the same mask, so the same bits -- including a row of
    length S, a row of length 1, garbage ids past a row's length, and a length-0 row (embeds to zeros)."""
    import torch
    enc, _ = _encoder(hip, "tiny", precision=precision)
    vocab = eo.SHAPES["tiny"][0]
    rng = np.random.default_rng(3)
    B, S = 13, 64
    lens = rng.integers(1, S + 1, size=B).astype(np.int32)
    lens[0], lens[1], lens[2] = S, 1, 0
    stage = rng.integers(1, vocab, size=(B, S + 1)).astype(np.int32)      # garbage past the length on purpose
    stage[:, S] = lens
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
    want = enc.forward(stage[:, :S] * mask, mask, pooling="mean", normalise=True).cpu().numpy()
    out = torch.full((B + 4, enc.hidden), 7.0, dtype=torch.float32, device="cuda")
    enc.forward_lens(torch.from_numpy(stage).cuda(), B, S, out[2:2 + B], pooling="mean", normalise=True)
    got = out.cpu().numpy()
    assert np.array_equal(got[2:2 + B], want)
    assert np.all(got[:2] == 7.0) and np.all(got[2 + B:] == 7.0)          # nothing outside the tile's rows was touched
    assert np.all(got[2 + 2] == 0.0)                                      # the length-0 row
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


def test_randomised_shapes_and_masks_against_oracle(hip):
    """16 seeded random (architecture, batch, padded length, ragged masks, pooling) configurations against the
    torch-fp32 oracle: covers every attention workgroup width (4 / 8 / 16 waves), the fused and the stand-alone
    LayerNorm paths (hidden 384 vs 128 / 768), tail tiles of the 256-token GEMM tile and 1-token sequences."""
    rng = np.random.default_rng(77)
    encs = {}
    for case in range(16):
        shape = str(rng.choice(["tiny", "minilm-l6", "minilm-l6", "bge-base"])) if case % 5 else "bge-base"
        if shape == "bge-base" and case not in (0, 5):
            shape = "minilm-l6"                         # two bge-base cases are enough (110M parameters on the CPU side)
        if shape not in encs:
            encs[shape] = _encoder(hip, shape)
        enc, w = encs[shape]
        vocab = eo.SHAPES[shape][0]
        if shape == "tiny":
            S = int(rng.choice([32, 64]))                # max_position 64
        elif shape == "bge-base":
            S = int(rng.choice([64, 512]))
        else:
            S = int(rng.choice([32, 64, 96, 160, 256, 384, 512]))
        B = int(rng.integers(1, 10)) if shape != "bge-base" else 2
        ids = rng.integers(min(1000, vocab // 2), min(30000, vocab), size=(B, S)).astype(np.int32)
        lens = rng.integers(1, S + 1, size=B)
        lens[0] = S                                      # one full-length row, the rest ragged
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
        pooling = str(rng.choice(["mean", "cls"]))
        got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
        want = eo.forward(shape, w, ids, mask, pooling=pooling)
        cos = (got * want).sum(1)
        assert cos.min() >= 1 - COS_TOL, (case, shape, B, S, pooling, cos.min())
        # unit vectors of fewer dimensions have larger components: the absolute bound scales with 1/sqrt(dim) below 384
        assert np.abs(got - want).max() <= ABS_TOL * max(1.0, (384 / got.shape[1]) ** 0.5), (case, shape, B, S, pooling)
    for enc, _ in encs.values():
        enc.close()


def test_oracle_with_strong_layernorm_parameters(hip):
    """LayerNorm weights far from (1, 0): gamma in [0.3, 2.5] with a few negative entries, beta ~ N(0, 0.5). The synthetic and
    fixture weights keep gamma within 5 % of 1, which would hide an error in anything that folds the LayerNorm into its
    neighbours (csrc/gemm.hip, lazy LayerNorm: gamma-scaled weight copies, column sums, folded biases; forced for every batch
    by AK_ENC_LAZYLN=2 in tests/test_02_encoder_variants_gpu.py) or fuses it into an epilogue (hidden 384)."""
    from archi_amd.encoder import HipEncoder
    rng = np.random.default_rng(11)
    for shape, B, S, pooling in (("bge-base", 3, 512, "cls"), ("bge-base", 2, 96, "mean"), ("minilm-l6", 9, 256, "mean")):
        vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
        w = eo.synth_weights(shape, seed=21)
        for k in list(w):
            if k.endswith("_g"):
                g = rng.uniform(0.3, 2.5, size=H)
                g[rng.integers(0, H, size=5)] *= -1.0
                w[k] = g.astype(np.float32)
            elif k.endswith("ln1_b") or k.endswith("ln2_b") or k == "emb_ln_b":
                w[k] = rng.normal(0, 0.5, size=H).astype(np.float32)
        enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0)
        ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
        lens = rng.integers(1, S + 1, size=B)
        lens[0] = S
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
        got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
        _check(got, eo.forward(shape, w, ids, mask, pooling=pooling))
        enc.close()


def test_left_padded_and_holed_masks_against_oracle(hip):
    """Attention masks that are not right-padded: the first 32-key block with a real key is not block 0 (the attention kernel
    peels THAT block for its running maximum), whole blocks of padding between real keys, a single real key."""
    enc, w = _encoder(hip, "minilm-l6")
    rng = np.random.default_rng(5)
    for S in (256, 160, 512):
        B = 6
        ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
        mask = np.ones((B, S), np.int32)
        mask[1, : S - 40] = 0                            # left-padded: real keys only in the last blocks
        mask[2, :64] = 0
        mask[2, 96:128] = 0                              # first live block is 2, a dead block follows it
        mask[3, :] = 0
        mask[3, S - 1] = 1                               # one real key, in the last block
        mask[4, 1::2] = 0                                # every other key masked
        mask[5, 33:] = 0
        mask[5, :31] = 0                                 # two real keys straddling a block boundary
        for pooling in ("mean", "cls"):
            got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
            want = eo.forward("minilm-l6", w, ids, mask, pooling=pooling)
            cos = (got * want).sum(1)
            assert cos.min() >= 1 - COS_TOL, (S, pooling, cos)
            assert np.abs(got - want).max() <= ABS_TOL, (S, pooling)
    enc.close()


@pytest.mark.parametrize("pooling,normalize,dtype", [("mean", True, "float32"), ("cls", True, "float16"),
                                                     ("mean", False, "float32")])
def test_provider_from_checkpoint_directory(hip, tmp_path, pooling, normalize, dtype):
    """ArchiHipEmbeddings(model_name=<local sentence-transformers directory>) against transformers.BertModel (fp32, CPU)
    + transformers' BertTokenizer on the same directory: tokenisation, truncation at the directory's max_seq_length,
    pooling mode and the Normalize module all come from the checkpoint's own files, as under the reference's
    HuggingFaceEmbeddings."""
    pytest.importorskip("transformers")
    from transformers import BertTokenizer
    from archi_amd.embeddings import ArchiHipEmbeddings
    from tests.hf_checkpoint import TEXTS, hf_embed, write_checkpoint
    d = str(tmp_path / "ckpt")
    model = write_checkpoint(d, pooling=pooling, max_seq_length=32, normalize=normalize, dtype=dtype)
    emb = ArchiHipEmbeddings(model_name=d, model_kwargs={"device": "cuda:0"})
    assert (emb.pooling, emb.max_seq_length, emb.normalize, emb.dimensions) == (pooling, 32, normalize, 128)
    got = np.asarray(emb.embed_documents(TEXTS), np.float32)
    words = open(os.path.join(d, "vocab.txt")).read().split("\n")[:-1]
    tok = BertTokenizer(vocab={w: i for i, w in enumerate(words)}, do_lower_case=True)
    enc = tok(TEXTS, truncation=True, max_length=32, padding=True, return_tensors="np")
    want = hf_embed(model, enc["input_ids"], enc["attention_mask"], pooling, normalize)
    if not normalize:
        scale = np.linalg.norm(want, axis=1).max()          # ABS_TOL is stated on unit vectors: scale by the row norm
        assert np.abs(got - want).max() <= ABS_TOL * scale
        got, want = got / np.linalg.norm(got, axis=1, keepdims=True), want / np.linalg.norm(want, axis=1, keepdims=True)
    _check(got, want)
    # encode_kwargs can switch normalisation on for a checkpoint without a Normalize module, never off for one with it
    if not normalize:
        e2 = ArchiHipEmbeddings(model_name=d, model_kwargs={"device": "cuda:0"}, encode_kwargs={"normalize_embeddings": True})
        assert np.allclose(np.linalg.norm(np.asarray(e2.embed_documents(TEXTS[:2])), axis=1), 1.0, atol=1e-4)


def test_bge_base_tile_path_and_small_batch_path(hip):
    """Hidden 768: more than 640 tokens run the 128-token-tile kernels (GEMM + stand-alone LayerNorm), fewer run the
    output/K-parallel small-batch GEMMs (gemm_skinny.hip); both against the torch-fp32 oracle, and against each other."""
    enc, w = _encoder(hip, "bge-base")
    ids, mask = eo.synth_tokens(3, 512, seed=99)
    mask[1, 300:] = 0
    got = enc.forward(ids, mask, pooling="cls", normalise=True).cpu().numpy()            # 1536 tokens: tile path
    _check(got, eo.forward("bge-base", w, ids, mask, pooling="cls"))
    one = enc.forward(ids[1:2], mask[1:2], pooling="cls", normalise=True).cpu().numpy()    # 512 tokens: small-batch path
    _check(one, got[1:2])
    enc.close()


def test_provider_is_safe_under_concurrent_request_threads(hip, tmp_path):
    """Chat request threads call embed_query concurrently (src/interfaces/chat_app/app.py:1554) while an ingestion thread
    runs embed_documents: one provider, one pinned staging area, one encoder handle -- every call must return what it
    returns alone."""
    import threading
    pytest.importorskip("transformers")
    from archi_amd.embeddings import ArchiHipEmbeddings
    from tests.hf_checkpoint import TEXTS, write_checkpoint
    d = str(tmp_path / "ckpt")
    write_checkpoint(d, pooling="mean", max_seq_length=32, normalize=True)
    emb = ArchiHipEmbeddings(model_name=d, model_kwargs={"device": "cuda:0"})
    want_q = [np.asarray(emb.embed_query(t), np.float32) for t in TEXTS]
    docs = [TEXTS[i % len(TEXTS)] + " grid" * (i % 5) for i in range(300)]
    want_d = emb.embed_documents_array(docs)
    bad = []

    def asker(tid):
        for it in range(30):
            j = (tid + it) % len(TEXTS)
            if not np.array_equal(np.asarray(emb.embed_query(TEXTS[j]), np.float32), want_q[j]):
                bad.append(("query", tid, it))

    def ingester():
        for it in range(5):
            if not np.array_equal(emb.embed_documents_array(docs), want_d):
                bad.append(("docs", it))

    threads = [threading.Thread(target=asker, args=(t,)) for t in range(4)] + [threading.Thread(target=ingester)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad[:3]


def test_bulk_call_rows_come_back_in_input_order(hip, tmp_path):
    """A call of several thousand texts of mixed lengths (length-sorted into many tiles, gathered tile by tile on the host):
    every row equals the row of the same text embedded in a small call."""
    pytest.importorskip("transformers")
    from archi_amd.embeddings import ArchiHipEmbeddings
    from tests.hf_checkpoint import TEXTS, write_checkpoint
    d = str(tmp_path / "ckpt")
    write_checkpoint(d, pooling="mean", max_seq_length=32, normalize=True)
    emb = ArchiHipEmbeddings(model_name=d, model_kwargs={"device": "cuda:0"}, encode_kwargs={"batch_tokens": 2048})
    docs = [TEXTS[(7 * i) % len(TEXTS)] + " grid" * (i % 9) + " cell" * (i % 4) for i in range(5000)]
    got = emb.embed_documents_array(docs)
    pick = [0, 1, 17, 2500, 4998, 4999]
    small = emb.embed_documents_array([docs[i] for i in pick])
    cos = (got[pick] * small).sum(1)
    assert cos.min() >= 1 - 1e-5, cos                     # tile composition differs, arithmetic per row does not
    assert np.abs(got[pick] - small).max() <= 2e-3
    assert emb.embed_documents_array([]).shape == (0, got.shape[1])
    emb.encoder.close()


def test_bge_base_bf16_holds_its_stated_tolerance_on_random_weight_seeds(hip):
    """DESIGN.md section 9 STATES the bf16 tolerance of hidden 768 (bge-base, 12 layers of bf16 activations, bf16 residual
    stream) as 1 - cos <= 3e-4 and max |diff| <= 3e-3 on unit rows against the float32 oracle; the fixed-seed tests above assert
    tighter numbers on their seeds and the soak script that measured the tail (worst 2.3e-4 / 2.6e-3 over rounds 4-5) is not a
    test. This one holds the stated number: 20 random weight seeds, ragged / left-padded masks, both poolings, several padded
    lengths (round-5 review, hygiene)."""
    rng = np.random.default_rng(2026)
    shape = "bge-base"
    vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
    from archi_amd.encoder import HipEncoder
    worst_cos, worst_abs = 0.0, 0.0
    for case in range(20):
        seed = int(rng.integers(1, 100000))
        w = eo.synth_weights(shape, seed=seed)
        enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0)
        S = int(rng.choice([64, 96, 128, 256]))
        B = int(rng.integers(3, 9))
        ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
        lens = rng.integers(1, S + 1, size=B)
        lens[0] = S
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
        pooling = "cls" if case % 2 else "mean"
        if case % 5 == 3 and pooling == "mean":
            mask = mask[:, ::-1].copy()                     # left-padded
        got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
        want = eo.forward(shape, w, ids, mask, pooling=pooling)
        enc.close()
        assert np.isfinite(got).all()
        cos = (got.astype(np.float64) * want).sum(1)
        worst_cos = max(worst_cos, float(1 - cos.min()))
        worst_abs = max(worst_abs, float(np.abs(got - want).max()))
        assert 1 - cos.min() <= 3e-4 and np.abs(got - want).max() <= 3e-3, (case, seed, B, S, pooling, 1 - cos.min(), np.abs(got - want).max())
    print(f"bge-base bf16, 20 weight seeds: worst 1 - cos {worst_cos:.2e}, worst max|diff| {worst_abs:.2e} (stated 3e-4 / 3e-3)")


@pytest.mark.parametrize("residual", ["bf16", "f32"])
def test_single_launch_query_forward_is_bit_identical_to_the_multi_launch_path(hip, residual):
    """embed_query's forward pass (<= 64 token rows) as ONE launch confined to one XCD (csrc/query_forward.hip, round 6) against
    the 47 launches it replaces: the same rows BIT FOR BIT -- one and two sequences, 32 and 64 padded tokens, ragged lengths,
    both poolings, normalised or not, the mask entry point and the lengths entry point. AK_QUERY_FUSED=2 makes the library fail
    rather than fall back, so a pass means the single launch really ran (and none of its bounded waits gave up)."""
    import ctypes
    import torch
    from archi_amd import _lib
    enc, w = _encoder(hip, "minilm-l6", residual=residual)
    vocab = eo.SHAPES["minilm-l6"][0]
    rng = np.random.default_rng(17)
    try:
        for case, (B, S) in enumerate([(1, 32), (1, 64), (2, 32), (1, 32), (1, 64)]):
            ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
            lens = rng.integers(1, S + 1, size=B).astype(np.int32)
            if case == 0:
                lens[:] = S
            mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
            for pooling in ("mean", "cls"):
                for normalise in (True, False):
                    _lib.debug_set("AK_QUERY_FUSED", "0")
                    want = enc.forward(ids, mask, pooling=pooling, normalise=normalise).cpu().numpy()
                    _lib.debug_set("AK_QUERY_FUSED", "2")
                    for rep in range(3):                   # (repeated: the barrier slots advance with every launch)
                        got = enc.forward(ids, mask, pooling=pooling, normalise=normalise).cpu().numpy()
                        assert np.array_equal(got, want), (B, S, pooling, normalise, rep, np.abs(got - want).max())
            # the lengths entry point (what the provider's tiles use): ids rows with their length in column S
            tile = np.zeros((B, S + 1), np.int32)
            tile[:, :S] = ids
            tile[:, S] = lens
            t_dev = torch.from_numpy(tile).cuda()
            outs = []
            for mode in ("0", "2"):
                _lib.debug_set("AK_QUERY_FUSED", mode)
                out = torch.empty((B, enc.hidden), dtype=torch.float32, device="cuda")
                _lib.check(enc._lib.ak_encoder_forward_lens(enc._h, ctypes.c_void_p(t_dev.data_ptr()), S + 1,
                                                            ctypes.c_void_p(t_dev.data_ptr() + 4 * S), S + 1, B, S, 0, 1,
                                                            ctypes.c_void_p(out.data_ptr()), None), "ak_encoder_forward_lens")
                torch.cuda.synchronize()
                outs.append(out.cpu().numpy())
            assert np.array_equal(outs[0], outs[1])
            _check(outs[1], eo.forward("minilm-l6", w, ids, mask, pooling="mean"))
        # 70 launches later the slots have wrapped once: still identical
        _lib.debug_set("AK_QUERY_FUSED", "2")
        ids, mask = eo.synth_tokens(1, 32, seed=5, vocab=vocab)
        first = enc.forward(ids, mask).cpu().numpy()
        for _ in range(80):
            assert np.array_equal(enc.forward(ids, mask).cpu().numpy(), first)
    finally:
        _lib.debug_set("AK_QUERY_FUSED", None)
        enc.close()
