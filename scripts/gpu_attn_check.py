#!/usr/bin/env python3
"""Streamed attention kernel against the unstreamed one (AK_ATTN_STREAM=1 against 0) on ragged masks: run with `child <out.npz>` per
mode from the parent, which compares the embeddings case by case."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
CASES = [("minilm", 3, 32), ("minilm", 5, 64), ("minilm", 4, 96), ("minilm", 3, 128), ("minilm", 6, 160), ("minilm", 2, 224),
         ("minilm", 7, 256), ("minilm", 3, 384), ("minilm", 2, 512), ("bge", 2, 64), ("bge", 2, 512), ("bge", 3, 288), ("minilm", 40, 256)]

def child(out):
    import torch
    from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
    res = {}
    encs = {}
    rng = np.random.default_rng(5)
    for ci, (name, B, S) in enumerate(CASES):
        full = {"minilm": "sentence-transformers/all-MiniLM-L6-v2", "bge": "BAAI/bge-base-en-v1.5"}[name]
        vocab, H, L, heads, I, max_pos, pooling, _ = MODEL_SHAPES[full]
        if name not in encs:
            encs[name] = HipEncoder(vocab, H, 2, heads, I, 512, random_init_weights(vocab, H, 2, I, 512, seed=1), device=0)
        ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
        lens = rng.integers(1, S + 1, size=B); lens[0] = S
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
        res[f"c{ci}"] = encs[name].forward(ids, mask, pooling="mean", normalise=True).cpu().numpy()
    np.savez(out, **res)

if len(sys.argv) > 2 and sys.argv[1] == "child":
    child(sys.argv[2]); sys.exit(0)
env = dict(os.environ)
env["AK_ATTN_STREAM"] = "1"
subprocess.check_call([sys.executable, __file__, "child", "/tmp/attn_new.npz"], env=env)
env["AK_ATTN_STREAM"] = "0"
subprocess.check_call([sys.executable, __file__, "child", "/tmp/attn_old.npz"], env=env)
a, b = np.load("/tmp/attn_new.npz"), np.load("/tmp/attn_old.npz")
for ci, c in enumerate(CASES):
    d = np.abs(a[f"c{ci}"] - b[f"c{ci}"]).max(axis=1)
    print(c, "max|diff| per row:", np.array2string(d, precision=5, max_line_width=200))
