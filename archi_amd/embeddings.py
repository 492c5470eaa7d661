"""ArchiHipEmbeddings -- drop-in for the embedding provider the reference builds from
`data_manager.embedding_class_map[name]["class"](**kwargs)`
(/root/reference/src/data_manager/vectorstore/manager.py:66-73,
 src/archi/utils/vectorstore_connector.py:28-35, src/utils/config_service.py:470-496):
LangChain's `Embeddings` duck type with the constructor keywords of HuggingFaceEmbeddings
(src/cli/templates/base-config.yaml:143-150).

    embed_documents(texts: List[str]) -> List[List[float]]
    embed_query(text: str) -> List[float]

Errors are RAISED (never partial results): the manager records a failed file per exception
(manager.py:374-389).

Host side (this file): text normalisation, WordPiece tokenisation, length-sorted batching.
Device side: archi_amd.encoder.HipEncoder (hand-written HIP). No CPU fallback.
"""
from __future__ import annotations

import os
import re
import zlib
from typing import Any, Dict, List, Optional

import numpy as np

from .encoder import (MODEL_SHAPES, HipEncoder, load_hf_weights, random_init_weights,
                      read_sentence_transformers_config)

CLS, SEP, PAD, UNK = 101, 102, 0, 100


class HashWordPiece:
    """Deterministic stand-in tokenizer for synthetic runs (no vocab.txt exists offline):
    lower-cased word/punctuation split, ids by CRC32 into [1000, vocab)."""

    def __init__(self, vocab: int = 30522):
        self.vocab = vocab
        self._re = re.compile(r"\w+|[^\w\s]")

    def encode(self, text: str, max_len: int) -> List[int]:
        ids = [CLS]
        for tok in self._re.findall(text.lower()):
            ids.append(1000 + zlib.crc32(tok.encode("utf-8")) % (self.vocab - 1000))
            if len(ids) >= max_len - 1:
                break
        ids.append(SEP)
        return ids

    def encode_batch(self, texts: List[str], max_len: int) -> List[List[int]]:
        return [self.encode(t, max_len) for t in texts]


class VocabWordPiece:
    """BERT WordPiece through the `tokenizers` wheel, from a local vocab.txt."""

    def __init__(self, vocab_file: str, lowercase: bool = True):
        from tokenizers import BertWordPieceTokenizer
        self._tok = BertWordPieceTokenizer(vocab_file, lowercase=lowercase)

    def encode(self, text: str, max_len: int) -> List[int]:
        ids = self._tok.encode(text).ids
        if len(ids) > max_len:
            ids = ids[: max_len - 1] + [SEP]
        return ids

    def encode_batch(self, texts: List[str], max_len: int) -> List[List[int]]:
        """One call into the Rust tokenizer for the whole list (it parallelises over its own thread pool), so
        ingestion-sized batches are not bound by a Python loop."""
        out = []
        for enc in self._tok.encode_batch(list(texts)):
            ids = enc.ids
            out.append(ids[: max_len - 1] + [SEP] if len(ids) > max_len else ids)
        return out


class ArchiHipEmbeddings:
    def __init__(self, model_name: str = "sentence-transformers/all-MiniLM-L6-v2",
                 model_kwargs: Optional[Dict[str, Any]] = None, encode_kwargs: Optional[Dict[str, Any]] = None,
                 **_ignored: Any):
        """model_name: a known architecture name or a local HF checkpoint directory.
        model_kwargs: {"device": "cuda[:i]"} ; {"synthetic_seed": int} builds seeded random-init weights of the
        named architecture (benchmarks/tests: the image has no checkpoints and no network); {"residual": "f32"} keeps
        the residual stream between layers in fp32 (default "bf16", see HipEncoder).
        encode_kwargs: {"normalize_embeddings": bool, "batch_tokens": int}."""
        self.model_name = model_name
        self.model_kwargs = dict(model_kwargs or {})
        self.encode_kwargs = dict(encode_kwargs or {})
        self.normalize = bool(self.encode_kwargs.get("normalize_embeddings", False))
        self.batch_tokens = int(self.encode_kwargs.get("batch_tokens", 65536))
        dev = str(self.model_kwargs.get("device", "cuda"))
        device = int(dev.split(":")[1]) if ":" in dev else None
        if os.path.isdir(model_name):
            shape, weights, eps = load_hf_weights(model_name)
            vocab, H, L, heads, I, max_pos = shape
            st_pool, st_len, st_norm = read_sentence_transformers_config(model_name)
            self.pooling = self.model_kwargs.get("pooling", st_pool)
            self.max_seq_length = min(int(self.model_kwargs.get("max_seq_length", st_len or max_pos)), max_pos, 512)
            self.normalize = self.normalize or st_norm     # a Normalize module in the checkpoint always applies
            vf = os.path.join(model_name, "vocab.txt")
            self.tokenizer = VocabWordPiece(vf) if os.path.exists(vf) else HashWordPiece(vocab)
        elif model_name in MODEL_SHAPES and "synthetic_seed" in self.model_kwargs:
            vocab, H, L, heads, I, max_pos, self.pooling, self.max_seq_length = MODEL_SHAPES[model_name]
            weights = random_init_weights(vocab, H, L, I, max_pos, seed=int(self.model_kwargs["synthetic_seed"]))
            eps = 1e-12
            self.tokenizer = HashWordPiece(vocab)
        else:
            raise FileNotFoundError(
                f"{model_name!r}: no local checkpoint directory (offline image). Pass a directory with config.json + "
                "model.safetensors (+ vocab.txt), or model_kwargs={'synthetic_seed': N} for seeded random weights")
        self.dimensions = H
        self.encoder = HipEncoder(vocab, H, L, heads, I, max_pos, weights, ln_eps=eps, device=device,
                                  residual=str(self.model_kwargs.get("residual", "bf16")))

    # -- LangChain Embeddings duck type -------------------------------------
    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        texts = [t.replace("\n", " ") for t in texts]       # langchain_huggingface does the same [upstream]
        if not texts:
            return []
        toks = self.tokenizer.encode_batch(texts, self.max_seq_length)
        out = self.embed_token_lists(toks)
        return [[float(x) for x in row] for row in out]     # float32 values widened to Python floats (a1)

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents([text])[0]

    # -- batching harness (the build's counterpart of manager.py:362-373: cross-file, length-sorted) --
    def embed_token_lists(self, toks: List[List[int]]) -> np.ndarray:
        order = sorted(range(len(toks)), key=lambda i: -len(toks[i]))
        out = np.empty((len(toks), self.dimensions), dtype=np.float32)
        i = 0
        while i < len(order):
            S = (len(toks[order[i]]) + 31) // 32 * 32
            nb = max(1, self.batch_tokens // S)
            chunk = order[i: i + nb]
            ids = np.zeros((len(chunk), S), np.int32)
            mask = np.zeros((len(chunk), S), np.int32)
            for r, j in enumerate(chunk):
                n = len(toks[j])
                ids[r, :n] = toks[j]
                mask[r, :n] = 1
            emb = self.encoder.forward(ids, mask, pooling=self.pooling, normalise=self.normalize)
            out[chunk] = emb.cpu().numpy()
            i += nb
        return out
