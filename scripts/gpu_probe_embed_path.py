#!/usr/bin/env python3
"""The product embed path and nothing else (run under rocprofv3 --kernel-trace --stats): ragged token rows through
ArchiHipEmbeddings.embed_token_arrays -- length-sorted tiles, one H2D per tile, ak_encoder_forward_lens, one D2H. The kernel list
must hold this build's kernels and the runtime's copies only (round-4 review: torch arange / where / cat kernels sat on this path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.embeddings import ArchiHipEmbeddings
prov = ArchiHipEmbeddings("sentence-transformers/all-MiniLM-L6-v2", model_kwargs={"synthetic_seed": 0, "device": "cuda:0"},
                          encode_kwargs={"normalize_embeddings": True})
rng = np.random.default_rng(0)
lens = rng.integers(8, 257, size=8192).astype(np.int32)
ids = np.zeros((len(lens), 256), np.int32)
for i, n in enumerate(lens):
    ids[i, :n] = rng.integers(1000, 30000, size=n)
for _ in range(3):
    out = prov.embed_token_arrays(ids, lens)
print("rows", out.shape, "norms", float(np.linalg.norm(out, axis=1).min()), float(np.linalg.norm(out, axis=1).max()))
