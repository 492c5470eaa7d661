#!/usr/bin/env python3
"""Where the time of the ragged-chunk path goes: host tiling vs per-tile GPU time (HIP events) vs the whole call."""
import time

import numpy as np
import torch

from archi_amd.embeddings import ArchiHipEmbeddings

S = 256
prov = ArchiHipEmbeddings("sentence-transformers/all-MiniLM-L6-v2", model_kwargs={"synthetic_seed": 0, "device": "cuda:0"},
                          encode_kwargs={"normalize_embeddings": True})
rng = np.random.default_rng(0)
toks = [rng.integers(1000, 30000, size=int(n)).tolist() for n in rng.integers(32, S + 1, size=4096)]
prov.embed_token_lists(toks[:512])
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prov.embed_token_lists(toks)
    print(f"whole call: {(time.perf_counter() - t0) * 1e3:.1f} ms")
lens = np.array([len(t) for t in toks])
order = np.argsort(-lens, kind="stable")
i = 0
tot = 0.0
while i < len(toks):
    Sx = max(32, (int(lens[order[i]]) + 31) // 32 * 32)
    nb = max(1, 65536 // Sx)
    n = len(order[i:i + nb])
    ids = torch.randint(1000, 30000, (n, Sx), dtype=torch.int32, device="cuda")
    mask = torch.ones((n, Sx), dtype=torch.int32, device="cuda")
    prov.encoder.forward(ids, mask)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        prov.encoder.forward(ids, mask)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        prov.encoder.forward(ids, mask)
    host_ms = (time.perf_counter() - t0) * 1e3 / 5
    torch.cuda.synchronize()
    ids_h, mask_h = ids.cpu().numpy(), mask.cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(5):
        prov.encoder.forward(ids_h, mask_h)
    hostnp_ms = (time.perf_counter() - t0) * 1e3 / 5
    torch.cuda.synchronize()
    tot += ms
    print(f"tile [{n},{Sx}]: gpu {ms:.3f} ms ({n * Sx / ms / 1e3:.0f}k tok/ms-ish), enqueue {host_ms:.3f} ms, enqueue from numpy {hostnp_ms:.3f} ms")
    i += nb
print(f"sum of tile GPU times {tot:.1f} ms -> {4096 / tot * 1e3:.0f} chunks/s if the host kept up")
