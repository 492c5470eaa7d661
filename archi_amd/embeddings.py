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

import ctypes
import os
import re
import threading
import zlib
from typing import Any, Dict, List, Optional

import numpy as np

from . import _lib
from ._lib import check
from .encoder import (MODEL_SHAPES, HipEncoder, load_hf_weights, random_init_weights,
                      read_sentence_transformers_config)

CLS, SEP, PAD, UNK = 101, 102, 0, 100


def _do_lower_case(model_dir: str) -> bool:
    """tokenizer_config.json's do_lower_case (BERT default: true), as the reference's AutoTokenizer reads it."""
    import json
    path = os.path.join(model_dir, "tokenizer_config.json")
    if os.path.exists(path):
        return bool(json.load(open(path)).get("do_lower_case", True))
    return True


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
        sep = self._tok.token_to_id("[SEP]")
        self._sep = SEP if sep is None else sep

    def encode(self, text: str, max_len: int) -> List[int]:
        ids = self._tok.encode(text).ids
        if len(ids) > max_len:
            ids = ids[: max_len - 1] + [self._sep]
        return ids

    def encode_batch(self, texts: List[str], max_len: int) -> List[List[int]]:
        """One call into the Rust tokenizer for the whole list (it parallelises over its own thread pool), so
        ingestion-sized batches are not bound by a Python loop."""
        out = []
        for enc in self._tok.encode_batch(list(texts)):
            ids = enc.ids
            out.append(ids[: max_len - 1] + [self._sep] if len(ids) > max_len else ids)
        return out


class NativeWordPiece:
    """BERT WordPiece from a local vocab.txt through libarchi_hip.so's multi-threaded host tokenizer
    (ak_wordpiece_encode, csrc/wordpiece.cpp). Texts that need Unicode tables (any non-ASCII byte) or hold a literal
    special token come back flagged and go through the `tokenizers` wheel (VocabWordPiece), so the ids are always the
    reference tokenizer's."""

    def __init__(self, vocab_file: str, lowercase: bool = True, threads: int = 0):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        check(self._lib.ak_wordpiece_create(vocab_file.encode(), int(lowercase), ctypes.byref(h)), "ak_wordpiece_create")
        self._h = h
        self._threads = int(os.environ.get("ARCHI_TOKENIZER_THREADS", threads))
        self._vocab_file, self._lowercase, self._full = vocab_file, lowercase, None

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ak_wordpiece_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fallback(self) -> "VocabWordPiece":
        if self._full is None:
            self._full = VocabWordPiece(self._vocab_file, lowercase=self._lowercase)
        return self._full

    def encode_batch_array(self, texts: List[str], max_len: int):
        """-> (ids [n, max_len] int32 zero padded, lens [n] int32)."""
        n = len(texts)
        enc = [t.encode("utf-8", "surrogatepass") for t in texts]
        offs = np.zeros(n + 1, np.int64)
        np.cumsum(np.fromiter((len(e) for e in enc), np.int64, n), out=offs[1:])
        blob = b"".join(enc)
        ids = np.empty((n, max_len), np.int32)
        lens = np.empty(n, np.int32)
        check(self._lib.ak_wordpiece_encode(self._h, blob, offs.ctypes.data, n, max_len, self._threads,
                                            ids.ctypes.data, lens.ctypes.data), "ak_wordpiece_encode")
        rest = np.flatnonzero(lens < 0)
        if rest.size:
            for i, row in zip(rest, self._fallback().encode_batch([texts[i] for i in rest], max_len)):
                ids[i, : len(row)] = row
                lens[i] = len(row)
        return ids, lens

    def encode_batch(self, texts: List[str], max_len: int) -> List[List[int]]:
        ids, lens = self.encode_batch_array(texts, max_len)
        return [ids[i, : lens[i]].tolist() for i in range(len(texts))]

    def encode(self, text: str, max_len: int) -> List[int]:
        return self.encode_batch([text], max_len)[0]


class ArchiHipEmbeddings:
    def __init__(self, model_name: str = "sentence-transformers/all-MiniLM-L6-v2",
                 model_kwargs: Optional[Dict[str, Any]] = None, encode_kwargs: Optional[Dict[str, Any]] = None,
                 **_ignored: Any):
        """model_name: a known architecture name or a local HF checkpoint directory.
        model_kwargs: {"device": "cuda[:i]"} ; {"synthetic_seed": int} builds seeded random-init weights of the
        named architecture (benchmarks/tests: the image has no checkpoints and no network); {"residual": "f32"} keeps
        the residual stream between layers in fp32 (default "bf16", see HipEncoder); {"precision": "f32"} selects the
        float32 parity mode (float32 weights and arithmetic, ~1e-6 from the reference's CPU embedder, ~1/9 of the bf16 rate),
        {"precision": "bf16x3"} the split-bf16 parity mode (float32 weights, GEMMs as three bf16 MFMA passes into one float32
        accumulator: the same top-k and scores within 1e-5 of the CPU path at ~3x the "f32" mode's rate).
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
            if not os.path.exists(vf):
                # real weights + hashed token ids = garbage embeddings with no error. The reference's embedder
                # (HuggingFaceEmbeddings -> SentenceTransformer -> AutoTokenizer [upstream]) raises when the checkpoint
                # has no tokenizer; HashWordPiece is only for seeded random-init models (synthetic_seed).
                raise FileNotFoundError(f"{model_name}: vocab.txt not found -- a checkpoint directory needs its WordPiece "
                                        "vocabulary (the hashing stand-in tokenizer is only used with synthetic_seed)")
            self.tokenizer = NativeWordPiece(vf, lowercase=_do_lower_case(model_name))
        elif model_name in MODEL_SHAPES and "synthetic_seed" in self.model_kwargs:
            vocab, H, L, heads, I, max_pos, self.pooling, self.max_seq_length = MODEL_SHAPES[model_name]
            weights = random_init_weights(vocab, H, L, I, max_pos, seed=int(self.model_kwargs["synthetic_seed"]))
            eps = 1e-12
            vf = self.model_kwargs.get("vocab_file")          # benchmarks: a synthetic vocab.txt for the random-init model
            self.tokenizer = NativeWordPiece(vf) if vf else HashWordPiece(vocab)
        else:
            raise FileNotFoundError(
                f"{model_name!r}: no local checkpoint directory (offline image). Pass a directory with config.json + "
                "model.safetensors (+ vocab.txt), or model_kwargs={'synthetic_seed': N} for seeded random weights")
        self.dimensions = H
        self._stage = self._stage_out = None
        self._stage_lock = threading.Lock()
        self.encoder = HipEncoder(vocab, H, L, heads, I, max_pos, weights, ln_eps=eps, device=device,
                                  residual=str(self.model_kwargs.get("residual", "bf16")),
                                  precision=str(self.model_kwargs.get("precision", "bf16")))

    # -- LangChain Embeddings duck type -------------------------------------
    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return self.embed_documents_array(texts).tolist()   # float32 values widened to Python floats (a1)

    def embed_documents_array(self, texts: List[str]) -> np.ndarray:
        """embed_documents without the List[List[float]] conversion (which costs more than the GPU work at ingestion
        sizes): the build's own callers (ArchiHipVectorStore.add_texts, BatchedIngestor) take the float32 rows."""
        texts = [t.replace("\n", " ") for t in texts]       # langchain_huggingface does the same [upstream]
        if not texts:
            return np.empty((0, self.dimensions), np.float32)
        if hasattr(self.tokenizer, "encode_batch_array"):
            return self.embed_token_arrays(*self.tokenizer.encode_batch_array(texts, self.max_seq_length))
        return self.embed_token_lists(self.tokenizer.encode_batch(texts, self.max_seq_length))

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents([text])[0]

    # -- batching harness (the build's counterpart of manager.py:362-373: cross-file, length-sorted) --
    def embed_token_lists(self, toks: List[List[int]]) -> np.ndarray:
        """Token lists -> embeddings (see embed_token_arrays)."""
        import itertools
        n = len(toks)
        lens = np.fromiter((len(t) for t in toks), np.int32, n)
        width = max(1, int(lens.max())) if n else 1
        ids = np.zeros((n, width), np.int32)
        if n:
            ids[np.arange(width)[None, :] < lens[:, None]] = np.fromiter(itertools.chain.from_iterable(toks), np.int32,
                                                                         int(lens.sum()))
        return self.embed_token_arrays(ids, lens)

    def embed_token_arrays(self, ids: np.ndarray, lens: np.ndarray) -> np.ndarray:
        """ids [n, W] int32 (row i holds lens[i] ids, zero padded) -> [n, D] float32.
        Length-sorted [B,S] tiles (S a multiple of 32, about `batch_tokens` tokens per tile). Per tile the host only
        gathers the tile's rows into a pinned buffer (one asynchronous copy); the mask is laid out on the device, every
        forward pass is enqueued without waiting for the previous one, and the embeddings come back in ONE
        device-to-host copy at the end -- the host prepares tile i+1 while the GPU runs tile i."""
        import torch
        n = len(lens)
        out = np.empty((n, self.dimensions), dtype=np.float32)
        if n == 0:
            return out
        lens = np.asarray(lens, np.int64)
        order = np.argsort(-lens, kind="stable")
        dev = getattr(self.encoder, "_dev", None)
        on_gpu = dev is not None and dev.type == "cuda"
        # tile plan first: (start, rows, S) and the staging offset of each tile ([rows, S + 1] int32, column S = length)
        plan, need, i = [], 0, 0
        while i < n:
            S = max(32, (int(lens[order[i]]) + 31) // 32 * 32)
            nb = min(max(1, self.batch_tokens // S), n - i)
            plan.append((i, nb, S, need))
            need += nb * (S + 1)
            i += nb
        with self._stage_lock:
            # ONE pinned staging area per provider, grow-only and reused across calls (a call ends with a full
            # synchronisation, so nothing of it is in flight when the next one starts): a pinned allocation per tile made
            # the caching host allocator fall back to hipHostMalloc whenever the previous tiles' copies were still in
            # flight -- identical calls took 39 to 95 ms
            if self._stage is None or self._stage.numel() < need:
                self._stage = torch.empty(max(need, 1 << 20), dtype=torch.int32, pin_memory=on_gpu)
            if self._stage_out is None or self._stage_out.numel() < n * self.dimensions:
                self._stage_out = torch.empty(max(n * self.dimensions, 1 << 20), dtype=torch.float32, pin_memory=on_gpu)
            host = self._stage.numpy()
            parts = []
            lens_path = on_gpu and hasattr(self.encoder, "forward_lens")
            dev_out = torch.empty((n, self.dimensions), dtype=torch.float32, device=dev) if lens_path else None
            for start, nb, S, off in plan:
                chunk = order[start: start + nb]
                view = host[off: off + nb * (S + 1)].reshape(nb, S + 1)
                w = min(S, ids.shape[1])
                # rows first, columns second: np.take on the column-sliced (strided) view with a strided `out` copied the WHOLE
                # id matrix once per tile -- quadratic in the call size (81k chunks: 2.5 s instead of 0.6). Measured with this
                # fixed and not kept: tokenising slice i + 1 on a host thread while the GPU embeds slice i (8k / 16k / 32k
                # texts per slice: 129 / 127 / 130 k chunks/s against 131 k for the one pass -- a sync and a D2H per slice
                # cost what the hidden tokeniser time, a tenth of the call, would have saved)
                view[:, :w] = ids[chunk, :w] if w == ids.shape[1] else ids[chunk][:, :w]
                if w < S:
                    view[:, w:S] = 0
                view[:, S] = lens[chunk]
                stage = self._stage[off: off + nb * (S + 1)].view(nb, S + 1)
                if lens_path:
                    # ONE asynchronous copy per tile, then the library: it lays the mask out from the lengths and writes the
                    # tile's rows at their place in the call's result buffer (no torch kernel between the H2D and the final D2H)
                    self.encoder.forward_lens(stage.to(dev, non_blocking=True), nb, S, dev_out[start: start + nb],
                                              pooling=self.pooling, normalise=self.normalize)
                    continue
                if on_gpu:
                    stage = stage.to(dev, non_blocking=True)
                valid = torch.arange(S, device=stage.device)[None, :] < stage[:, S:]
                tile = torch.where(valid, stage[:, :S], 0)       # whatever sits past a row's length is not a token
                parts.append(self.encoder.forward(tile, valid.int(), pooling=self.pooling, normalise=self.normalize))
            res = self._stage_out[: n * self.dimensions].view(n, self.dimensions)
            res.copy_(dev_out if lens_path else torch.cat(parts), non_blocking=False)      # one device-to-host copy, into pinned memory
            out[order] = res.numpy()
        return out
