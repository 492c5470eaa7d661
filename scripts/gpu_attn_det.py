import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
vocab, H, L, heads, I, max_pos, pooling, S0 = MODEL_SHAPES["sentence-transformers/all-MiniLM-L6-v2"]
enc = HipEncoder(vocab, H, 2, heads, I, 512, random_init_weights(vocab, H, 2, I, 512, seed=1), device=0)
rng = np.random.default_rng(0)
def run(B, S, seed):
    r = np.random.default_rng(seed)
    ids = r.integers(1000, 30000, size=(B, S)).astype(np.int32)
    lens = r.integers(1, S + 1, size=B); lens[0] = S
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
    return enc.forward(ids, mask).cpu().numpy()
for (B, S) in [(300, 32), (64, 32), (40, 64), (256, 256), (33, 96)]:
    a = run(B, S, 1)
    run(1, 32, 2); run(7, 64, 3)
    b = run(B, S, 1)
    c = run(B, S, 1)
    print(B, S, "repeat equal:", np.array_equal(a, b), np.array_equal(b, c), np.abs(a - b).max())
