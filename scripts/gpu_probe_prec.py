import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import encoder_oracle as eo
from archi_amd.encoder import HipEncoder
vocab, H, L, heads, I, max_pos, _ = eo.SHAPES["minilm-l6"]
w = eo.synth_weights("minilm-l6", seed=7)
ids, mask = eo.synth_tokens(32, 256, seed=3)
want = eo.forward("minilm-l6", w, ids, mask)
for res in ("f32", "bf16"):
    enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0, residual=res)
    got = enc.forward(ids, mask).cpu().numpy(); enc.close()
    cos = (got*want).sum(1)
    print(res, "min cos", cos.min(), "mean 1-cos", (1-cos).mean(), "max abs", np.abs(got-want).max())
