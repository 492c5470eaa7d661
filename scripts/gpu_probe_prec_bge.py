"""bge-base (hidden 768: GEMM + stand-alone LayerNorm) against the torch-fp32 oracle: residual modes, and the bf16 GEMM
output of the bf16-residual mode (AK_ENC_Y32=1 keeps it fp32)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from archi_amd.encoder import HipEncoder
from oracle import encoder_oracle as eo

vocab, H, L, heads, I, max_pos, _ = eo.SHAPES["bge-base"]
w = eo.synth_weights("bge-base", seed=7)
ids, mask = eo.synth_tokens(6, 512, seed=3)          # 3072 tokens: tile kernels
want = eo.forward("bge-base", w, ids, mask, pooling="cls")
for res in ("f32", "bf16"):
    enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0, residual=res)
    got = enc.forward(ids, mask, pooling="cls").cpu().numpy()
    enc.close()
    cos = (got * want).sum(1)
    print(f"residual={res} AK_ENC_Y32={os.environ.get('AK_ENC_Y32')}: min cos {cos.min():.8f}  mean 1-cos {(1 - cos).mean():.2e}  max abs {np.abs(got - want).max():.2e}")
