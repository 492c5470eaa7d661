#!/usr/bin/env python3
"""Length-sorted ingestion rate against the tile size (encode_kwargs batch_tokens): 16 384 token lists, lengths uniform in [32, 256]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.embeddings import ArchiHipEmbeddings
rng = np.random.default_rng(0)
lens = rng.integers(32, 257, size=16384).astype(np.int32)
ids = np.zeros((len(lens), 256), np.int32)
for i, n in enumerate(lens):
    ids[i, :n] = rng.integers(1000, 30000, size=n)
for bt in (32768, 65536, 131072, 262144):
    prov = ArchiHipEmbeddings("sentence-transformers/all-MiniLM-L6-v2", model_kwargs={"synthetic_seed": 0, "device": "cuda:0"},
                              encode_kwargs={"normalize_embeddings": True, "batch_tokens": bt})
    prov.embed_token_arrays(ids[:2048], lens[:2048])
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prov.embed_token_arrays(ids, lens)
        best = min(best, time.perf_counter() - t0)
    print(f"batch_tokens {bt}: {len(lens) / best:.0f} chunks/s ({best * 1e3:.1f} ms)")
    del prov
