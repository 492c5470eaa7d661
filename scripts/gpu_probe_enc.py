#!/usr/bin/env python3
"""Forward passes of one encoder shape: scripts/gpu_probe_enc.py <minilm|bge> [B] [n]; run under rocprofv3 --kernel-trace --stats
for the per-kernel table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
name = {"minilm": "sentence-transformers/all-MiniLM-L6-v2", "bge": "BAAI/bge-base-en-v1.5"}[sys.argv[1] if len(sys.argv) > 1 else "bge"]
vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
if os.environ.get("SEQ"):
    S = int(os.environ["SEQ"])
enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=0)
ids = torch.from_numpy(np.random.default_rng(0).integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
if os.environ.get("RAGGED"):   # padded batch: real lengths uniform in [32, S]
    lens = torch.from_numpy(np.random.default_rng(1).integers(32, S + 1, size=B)).cuda()
    mask = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).to(torch.int32)
for _ in range(3): enc.forward(ids, mask, pooling=pooling)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): enc.forward(ids, mask, pooling=pooling)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"{name} B={B} S={S}: forward {dt * 1e3:.3f} ms  ({B / dt:.0f} chunks/s)")
