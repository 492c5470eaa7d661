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
