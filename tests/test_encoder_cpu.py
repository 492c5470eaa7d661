"""CPU suite: the encoder oracle (plain torch restatement) against the fixtures produced by
transformers.BertModel in the build container (tests/golden/make_encoder_fixtures.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import encoder_oracle as eo

FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "encoder_*.npz")))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p) for p in FIX])
def test_oracle_reproduces_hf_fixture(path):
    f = np.load(path)
    shape, pooling = str(f["shape"]), str(f["pooling"])
    if shape == "bge-base":
        pytest.skip("bge-base fixture is covered on the GPU box (12 layers x 110M params is slow on this CPU)")
    w = eo.synth_weights(shape, seed=int(f["weight_seed"]))
    got = eo.forward(shape, w, f["ids"], f["mask"], pooling=pooling)
    assert np.abs(got - f["expected"]).max() < 2e-6
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)


def test_padding_tokens_do_not_change_embeddings():
    w = eo.synth_weights("tiny", seed=7)
    ids, mask = eo.synth_tokens(3, 20, seed=1, vocab=1000)
    a = eo.forward("tiny", w, ids, mask)
    ids2 = np.concatenate([ids, np.zeros((3, 12), np.int32)], axis=1)
    mask2 = np.concatenate([mask, np.zeros((3, 12), np.int32)], axis=1)
    assert np.abs(a - eo.forward("tiny", w, ids2, mask2)).max() < 1e-6


def test_hash_tokenizer_and_batching_host_logic():
    from archi_amd.embeddings import CLS, SEP, HashWordPiece
    t = HashWordPiece(30522)
    ids = t.encode("Hello, world! hello", 16)
    assert ids[0] == CLS and ids[-1] == SEP and ids[1] == ids[5] and all(0 <= i < 30522 for i in ids)
    assert len(t.encode("a " * 1000, 256)) == 256


def test_vocab_wordpiece_batch_equals_single_and_truncates(tmp_path):
    """BERT WordPiece through the `tokenizers` wheel from a local vocab.txt (no network): batch encoding is the
    path embed_documents uses; it must equal one-by-one encoding, add [CLS]/[SEP] and truncate like the reference's
    tokenizer (max_seq_length, keeping the final [SEP])."""
    pytest.importorskip("tokenizers")
    from archi_amd.embeddings import CLS, SEP, VocabWordPiece
    words = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + \
            ["the", "muon", "detector", "cal", "##ib", "##ration", "run", "grid", ".", ","]
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(words) + "\n")
    tok = VocabWordPiece(str(vf))
    texts = ["The muon detector calibration run.", "grid, grid grid", "", "zzz unknown", "run " * 40]
    single = [tok.encode(t, 16) for t in texts]
    batch = tok.encode_batch(texts, 16)
    assert batch == single
    assert batch[0][0] == CLS and batch[0][-1] == SEP and batch[2] == [CLS, SEP]
    assert words.index("##ib") in batch[0] and words.index("##ration") in batch[0]       # WordPiece continuation pieces
    assert len(batch[4]) == 16 and batch[4][-1] == SEP                                    # truncated, [SEP] kept
    assert batch[3][1:-1] == [100, 100]                                                   # [UNK] for out-of-vocab words


def test_checkpoint_directory_loader(tmp_path):
    """A local sentence-transformers directory (what the reference's HuggingFaceEmbeddings points at): every tensor
    is picked up under the right name (oracle forward on the loaded weights == transformers.BertModel on the same
    checkpoint), f16 storage loads, and pooling / max_seq_length / Normalize come from the directory's own files."""
    pytest.importorskip("transformers")
    from archi_amd.encoder import load_hf_weights, read_sentence_transformers_config
    from tests.hf_checkpoint import hf_embed, write_checkpoint
    d = str(tmp_path / "ckpt")
    model = write_checkpoint(d, pooling="cls", max_seq_length=32, normalize=True)
    shape, w, eps = load_hf_weights(d)
    assert shape == (1000, 128, 2, 4, 256, 64) and eps == 1e-12
    assert read_sentence_transformers_config(d) == ("cls", 32, True)
    ids, mask = eo.synth_tokens(4, 24, seed=3, vocab=1000)
    wn = {k: v.numpy() for k, v in w.items()}
    for pooling in ("cls", "mean"):
        got = eo.forward("tiny", wn, ids, mask, pooling=pooling, eps=eps)
        assert np.abs(got - hf_embed(model, ids, mask, pooling)).max() < 2e-6
    d16 = str(tmp_path / "ckpt16")
    write_checkpoint(d16, pooling="mean", normalize=False, dtype="float16")
    _, w16, _ = load_hf_weights(d16)
    assert all(v.dtype.is_floating_point and v.element_size() == 4 for v in w16.values())
    assert np.abs(w16["l1.w2"].numpy() - wn["l1.w2"]).max() < 1e-3
    assert read_sentence_transformers_config(d16) == ("mean", 32, False)
    assert read_sentence_transformers_config(str(tmp_path)) == ("mean", None, False)     # plain HF directory


def test_checkpoint_loader_refuses_what_the_encoder_does_not_implement(tmp_path):
    import json
    from archi_amd.encoder import load_hf_weights
    for bad in ({"model_type": "roberta"}, {"hidden_act": "relu"}, {"position_embedding_type": "relative_key"}):
        d = tmp_path / next(iter(bad))
        d.mkdir()
        (d / "config.json").write_text(json.dumps({"num_hidden_layers": 1, **bad}))
        with pytest.raises(ValueError):
            load_hf_weights(str(d))
    d = tmp_path / "noweights"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"num_hidden_layers": 1}))
    with pytest.raises(FileNotFoundError):
        load_hf_weights(str(d))


def test_vocab_wordpiece_equals_transformers_bert_tokenizer(tmp_path):
    """Token ids must be the ones the reference's tokenizer (transformers BertTokenizer over the checkpoint's
    vocab.txt) produces, truncation at max_seq_length included."""
    pytest.importorskip("transformers")
    from transformers import BertTokenizer
    from archi_amd.embeddings import VocabWordPiece
    from tests.hf_checkpoint import TEXTS, WORDS
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(vocab) + "\n")
    ref = BertTokenizer(vocab={w: i for i, w in enumerate(vocab)}, do_lower_case=True)
    texts = TEXTS + ["", "Ünïcode café, the MUON!", "a" * 300, "the\tdetector\nrun"]
    for max_len in (8, 16, 64):
        want = ref(texts, truncation=True, max_length=max_len, add_special_tokens=True)["input_ids"]
        assert VocabWordPiece(str(vf)).encode_batch(texts, max_len) == want
