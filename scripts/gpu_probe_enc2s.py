#!/usr/bin/env python3
"""A/B: one [B,S] forward on one stream against NS sub-batches on NS streams (one encoder handle each, same weights):
scripts/gpu_probe_enc2s.py <minilm|bge> [B] [n] [NS]. Do the sub-batches' HBM-bound kernel heads / tails overlap the others' loops?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
name = {"minilm": "sentence-transformers/all-MiniLM-L6-v2", "bge": "BAAI/bge-base-en-v1.5"}[sys.argv[1] if len(sys.argv) > 1 else "minilm"]
vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
NS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
w = random_init_weights(vocab, H, L, I, max_pos, seed=0)
encs = [HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
ids = torch.from_numpy(np.random.default_rng(0).integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
sub = B // NS
parts = [(ids[i * sub:(i + 1) * sub].contiguous(), mask[i * sub:(i + 1) * sub].contiguous()) for i in range(NS)]

def one():
    encs[0].forward(ids, mask, pooling=pooling)

def split():
    for e, st, (pi, pm) in zip(encs, streams, parts):
        with torch.cuda.stream(st):
            e.forward(pi, pm, pooling=pooling)

for fn, label in ((one, "one stream"), (split, f"{NS} streams x {sub}"), (one, "one stream"), (split, f"{NS} streams x {sub}")):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{label:18s} B={B} S={S}: {dt * 1e3:.3f} ms  ({B / dt:.0f} chunks/s)")
