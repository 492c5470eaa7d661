"""CPU ORACLE for the chunk-embedding forward pass (TEST INFRASTRUCTURE ONLY).

Restates, in plain torch fp32 on the CPU, what the reference's embedding provider computes
at /root/reference/src/data_manager/vectorstore/manager.py:373 and
src/data_manager/vectorstore/postgres_vectorstore.py:143,245,390:

    HuggingFaceEmbeddings(model_name, model_kwargs, encode_kwargs).embed_documents(texts)
      -> sentence_transformers.SentenceTransformer.encode
         -> transformers BertModel forward -> Pooling (mean | cls) -> Normalize

The arithmetic lives in third-party packages that are ABSENT from /root/reference
(sentence-transformers==5.1.2, langchain-huggingface==1.0.0, torch==2.6.0, transformers
unpinned: requirements/requirements-base.txt:47,49,87). What is restated is their published
algorithm for BERT-family encoders:

  embeddings : LayerNorm(word[ids] + position[0..S-1] + token_type[0]), eps = 1e-12
  layer x L  : q,k,v = x Wq^T+bq, ...; heads of size H/heads;
               p = softmax(q k^T / sqrt(hd) + (1-mask)*finfo.min); ctx = p v
               x = LayerNorm(x + ctx Wo^T + bo)
               x = LayerNorm(x + gelu_erf(x W1^T + b1) W2^T + b2)
  pooling    : mean over tokens with attention_mask==1 (sum / clamp(count, 1e-9))
               [all-MiniLM-L6-v2], or the [CLS] token [bge-base-en]
  normalise  : x / max(||x||_2, 1e-12)   (encode_kwargs.normalize_embeddings,
               src/cli/templates/base-config.yaml:149-150)

Pinned against the real thing where it exists in the build container: tests/golden/
make_encoder_fixtures.py runs transformers.BertModel (same weights) and stores its outputs;
tests check this restatement against those fixtures. No pretrained weights or vocab exist
offline, so weights are seeded synthetic ones (SURVEY.md section 7 "hard parts").
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from . import knn_oracle as ko

SHAPES = {
    # name: (vocab, hidden, layers, heads, intermediate, max_pos, pooling)
    "minilm-l6": (30522, 384, 6, 12, 1536, 512, "mean"),     # sentence-transformers/all-MiniLM-L6-v2
    "bge-base": (30522, 768, 12, 12, 3072, 512, "cls"),      # BAAI/bge-base-en
    "tiny": (1000, 128, 2, 4, 256, 64, "mean"),
}


def weight_names(layers: int):
    names = ["word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b"]
    for l in range(layers):
        for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2",
                  "ln2_g", "ln2_b"):
            names.append(f"l{l}.{n}")
    return names


def synth_weights(shape: str, seed: int = 7) -> Dict[str, np.ndarray]:
    """Deterministic synthetic weights from the counter-based generator (same on every box).
    Matrices ~ N(0, 0.04^2)-like, rounded through bf16 (the dtype the HIP encoder keeps them in),
    so oracle and HIP path see identical parameter values."""
    vocab, H, L, heads, I, max_pos, _ = SHAPES[shape]
    w: Dict[str, np.ndarray] = {}
    stream = [100]

    def mat(rows, cols, scale):
        stream[0] += 1
        m = ko.gen_rows(seed, stream[0], 0, rows, cols + (cols & 1), False, "f32")[:, :cols]
        return ko.round_through(np.ascontiguousarray(m * (scale / 0.8164)), "bf16")   # raw sigma ~ 0.8164

    def vec(n, scale, offset=0.0):
        return (mat(1, n, scale)[0] + offset).astype(np.float32)

    w["word_emb"] = mat(vocab, H, 0.05)
    w["pos_emb"] = mat(max_pos, H, 0.05)
    w["type_emb"] = mat(2, H, 0.05)
    w["emb_ln_g"] = vec(H, 0.05, 1.0)
    w["emb_ln_b"] = vec(H, 0.05)
    for l in range(L):
        for n, (r, c) in (("wq", (H, H)), ("wk", (H, H)), ("wv", (H, H)), ("wo", (H, H)), ("w1", (I, H)),
                          ("w2", (H, I))):
            w[f"l{l}.{n}"] = mat(r, c, 0.04)
        for n, d in (("bq", H), ("bk", H), ("bv", H), ("bo", H), ("b1", I), ("b2", H)):
            w[f"l{l}.{n}"] = vec(d, 0.02)
        for n in ("ln1", "ln2"):
            w[f"l{l}.{n}_g"] = vec(H, 0.05, 1.0)
            w[f"l{l}.{n}_b"] = vec(H, 0.05)
    return w


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def forward(shape: str, w: Dict[str, np.ndarray], ids: np.ndarray, mask: np.ndarray, pooling: str = None,
            normalise: bool = True, eps: float = 1e-12) -> np.ndarray:
    """ids, mask: [B,S] int. Returns [B,H] float32 embeddings."""
    vocab, H, L, heads, I, max_pos, default_pool = SHAPES[shape]
    pooling = pooling or default_pool
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in w.items()}
    ids_t = torch.from_numpy(np.asarray(ids)).long()
    m = torch.from_numpy(np.asarray(mask)).float()
    B, S = ids_t.shape
    hd = H // heads
    x = t["word_emb"][ids_t] + t["pos_emb"][:S][None] + t["type_emb"][0][None, None]
    x = _ln(x, t["emb_ln_g"], t["emb_ln_b"], eps)
    bias = (1.0 - m)[:, None, None, :] * torch.finfo(torch.float32).min
    for l in range(L):
        p = f"l{l}."
        q = (x @ t[p + "wq"].T + t[p + "bq"]).view(B, S, heads, hd).transpose(1, 2)
        k = (x @ t[p + "wk"].T + t[p + "bk"]).view(B, S, heads, hd).transpose(1, 2)
        v = (x @ t[p + "wv"].T + t[p + "bv"]).view(B, S, heads, hd).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(hd) + bias
        ctx = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, S, H)
        x = _ln(x + ctx @ t[p + "wo"].T + t[p + "bo"], t[p + "ln1_g"], t[p + "ln1_b"], eps)
        h = x @ t[p + "w1"].T + t[p + "b1"]
        h = h * 0.5 * (1.0 + torch.erf(h / math.sqrt(2.0)))
        x = _ln(x + h @ t[p + "w2"].T + t[p + "b2"], t[p + "ln2_g"], t[p + "ln2_b"], eps)
    if pooling == "cls":
        out = x[:, 0]
    else:
        out = (x * m[:, :, None]).sum(1) / m.sum(1, keepdim=True).clamp(min=1e-9)
    if normalise:
        out = out / out.norm(dim=1, keepdim=True).clamp(min=1e-12)
    return out.numpy().astype(np.float32)


def synth_tokens(B: int, S: int, seed: int = 11, ragged: bool = True, vocab: int = 30522):
    """Token ids uniform in [1000, 30000) clipped to the vocab; ragged attention masks."""
    rng = np.random.default_rng(seed)
    hi = min(30000, vocab)
    lo = min(1000, hi // 2)
    ids = rng.integers(lo, hi, size=(B, S)).astype(np.int32)
    mask = np.ones((B, S), np.int32)
    if ragged:
        lens = rng.integers(max(1, S // 4), S + 1, size=B)
        lens[0] = S
        for b in range(B):
            mask[b, lens[b]:] = 0
            ids[b, lens[b]:] = 0
    return ids, mask
