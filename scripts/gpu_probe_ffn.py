#!/usr/bin/env python3
"""MiniLM-shape forward passes [256,256]: time per pass with the fused feed-forward kernel and without; AK_FFN_DBG=1 prints its phase cycles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES["sentence-transformers/all-MiniLM-L6-v2"]
enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=0)
ids = torch.from_numpy(np.random.default_rng(0).integers(1000, 30000, size=(256, S)).astype(np.int32)).cuda()
mask = torch.ones((256, S), dtype=torch.int32, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3): enc.forward(ids, mask, pooling=pooling)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): enc.forward(ids, mask, pooling=pooling)
torch.cuda.synchronize()
print(f"forward {(time.perf_counter() - t0) / n * 1e3:.3f} ms  ({256 * n / (time.perf_counter() - t0):.0f} chunks/s)")
