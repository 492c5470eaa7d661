#!/usr/bin/env python3
"""Latency of the read path's embedding step (a2): embed_query on one short text, and its parts."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from archi_amd.embeddings import ArchiHipEmbeddings
from tests.synth_text import make_vocab_file

td = tempfile.mkdtemp()
prov = ArchiHipEmbeddings("sentence-transformers/all-MiniLM-L6-v2",
                          model_kwargs={"synthetic_seed": 0, "device": "cuda:0", "vocab_file": make_vocab_file(os.path.join(td, "v.txt"))},
                          encode_kwargs={"normalize_embeddings": True})
text = "kalo miren stavor quzen phitor elan droxi bune sygra kalomi renstavor quzenphi"
for _ in range(20):
    prov.embed_query(text)
t0 = time.perf_counter()
for _ in range(200):
    prov.embed_query(text)
print(f"embed_query: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
ids, lens = prov.tokenizer.encode_batch_array([text], 256)
t0 = time.perf_counter()
for _ in range(200):
    prov.tokenizer.encode_batch_array([text], 256)
print(f"  tokenizer: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms  ({int(lens[0])} tokens)")
S = 32
tile = torch.zeros((1, S), dtype=torch.int32, device="cuda"); tile[0, :lens[0]] = torch.from_numpy(ids[0, :lens[0]]).cuda()
mask = (torch.arange(S, device="cuda")[None] < int(lens[0])).int()
for _ in range(20):
    prov.encoder.forward(tile, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    prov.encoder.forward(tile, mask)
torch.cuda.synchronize()
print(f"  encoder.forward [1,{S}] back to back: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(200):
    prov.encoder.forward(tile, mask).cpu()
print(f"  encoder.forward [1,{S}] + D2H each: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(200):
    prov.embed_token_arrays(ids, lens)
print(f"  embed_token_arrays: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
