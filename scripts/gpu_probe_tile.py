#!/usr/bin/env python3
"""One encoder tile shape, a few forward passes (run under rocprofv3 --kernel-trace --stats): AK_TILE=B,S"""
import os

import torch

from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights

B, S = (int(x) for x in os.environ.get("AK_TILE", "682,96").split(","))
vocab, H, L, heads, I, max_pos, pooling, _ = MODEL_SHAPES[os.environ.get("AK_MODEL", "sentence-transformers/all-MiniLM-L6-v2")]
enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=0)
ids = torch.randint(1000, 30000, (B, S), dtype=torch.int32, device="cuda")
mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
for _ in range(6):
    enc.forward(ids, mask)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(int(os.environ.get('AK_REPS', '200'))):
    enc.forward(ids, mask)
torch.cuda.synchronize()
print(f"tile [{B},{S}] skinny_max={os.environ.get('AK_ENC_SKINNY_MAX', 'default')}: {(time.perf_counter() - t0) / int(os.environ.get('AK_REPS', '200')) * 1e3:.3f} ms per forward")
