"""Child of tests/test_02_encoder_variants_gpu.py: embeddings of a fixed set of (shape, batch, length, ragged mask) cases under
whatever AK_* kernel-selection variables the parent set; one .npz out."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [("minilm", 3, 32), ("minilm", 5, 96), ("minilm", 6, 160), ("minilm", 7, 256), ("minilm", 3, 384), ("minilm", 2, 512),
         ("minilm", 40, 256), ("bge", 2, 64), ("bge", 3, 288), ("bge", 2, 512),
         ("bge", 12, 512)]        # 6144 tokens: 288 wide FFN-up tiles on 256 workgroups -- some carry two tiles through the loop


def main(out):
    from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
    res, encs = {}, {}
    rng = np.random.default_rng(5)
    for ci, (name, B, S) in enumerate(CASES):
        full = {"minilm": "sentence-transformers/all-MiniLM-L6-v2", "bge": "BAAI/bge-base-en-v1.5"}[name]
        vocab, H, L, heads, I, max_pos, pooling, _ = MODEL_SHAPES[full]
        if name not in encs:
            encs[name] = HipEncoder(vocab, H, 2, heads, I, 512, random_init_weights(vocab, H, 2, I, 512, seed=1), device=0)
        ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
        lens = rng.integers(1, S + 1, size=B)
        lens[0] = S
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
        res[f"c{ci}"] = encs[name].forward(ids, mask, pooling="mean", normalise=True).cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
