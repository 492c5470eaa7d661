#!/usr/bin/env python3
"""Forward time against the batch size at the bench shapes: would the library gain by walking a large batch in token slabs whose
activations fit the 256 MB Infinity Cache?   python3 scripts/gpu_probe_slab.py [model] [batches...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
model = sys.argv[1] if len(sys.argv) > 1 else "BAAI/bge-base-en-v1.5"
batches = [int(x) for x in sys.argv[2:]] or [16, 32, 64, 128, 256]
vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[model]
enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=0)
rng = np.random.default_rng(0)
for B in batches:
    ids = torch.from_numpy(rng.integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
    mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
    for _ in range(3): enc.forward(ids, mask, pooling=pooling)
    n = max(5, 2048 // B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): enc.forward(ids, mask, pooling=pooling)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{model} B={B} S={S}: forward {dt * 1e3:.3f} ms  ({B / dt:.0f} chunks/s, {dt * 1e6 / (B * S / 1024):.2f} us per 1024 tokens)", flush=True)
