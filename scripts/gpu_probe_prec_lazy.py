#!/usr/bin/env python3
"""bge-base shape, 48 x 512 tokens (the batch size from which the lazy LayerNorm is launched) against the torch-fp32 oracle on
sampled chunks: min cosine, mean 1 - cos, max |diff|. Run once with AK_ENC_LAZYLN unset and once with AK_ENC_LAZYLN=0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
from oracle import encoder_oracle as eo
vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES["BAAI/bge-base-en-v1.5"]
B = 48
rng = np.random.default_rng(0)
ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
mask = np.ones((B, S), np.int32)
pick = list(range(0, B, 6))
for label, w in (("random_init", random_init_weights(vocab, H, L, I, max_pos, seed=0)), ("synth", eo.synth_weights("bge-base", seed=31))):
    w = {k: np.asarray(v) for k, v in w.items()}
    enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0)
    for pool in ("cls", "mean"):
        got = enc.forward(ids, mask, pooling=pool, normalise=True).cpu().numpy()[pick]
        ref = eo.forward("bge-base", w, ids[pick], mask[pick], pooling=pool)
        cos = (got * ref).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(ref, axis=1))
        print(f"LAZYLN={os.environ.get('AK_ENC_LAZYLN', '1')} {label:12s} {pool:4s}: min cos 1-{1 - cos.min():.2e}  mean 1-cos {np.mean(1 - cos):.2e}  max|diff| {np.abs(got - ref).max():.2e}")
    enc.close()
