import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES["sentence-transformers/all-MiniLM-L6-v2"]
enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=3), device=0)
rng = np.random.default_rng(11)
out = {}
for B, S_ in ((256, 256), (160, 224), (300, 128)):
    ids = rng.integers(1000, 30000, size=(B, S_)).astype(np.int32)
    lens = rng.integers(1, S_ + 1, size=B); lens[0] = S_
    mask = (np.arange(S_)[None, :] < lens[:, None]).astype(np.int32)
    out[f"{B}x{S_}"] = enc.forward(ids, mask, pooling="mean").cpu().numpy()
np.savez(sys.argv[1], **out)
