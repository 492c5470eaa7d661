import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.encoder import HipEncoder, MODEL_SHAPES, random_init_weights
name = sys.argv[1] if len(sys.argv) > 1 else "sentence-transformers/all-MiniLM-L6-v2"
B, S = int(sys.argv[2]), int(sys.argv[3])
vocab, H, L, heads, I, max_pos, pooling, _ = MODEL_SHAPES[name]
enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=0)
ids = torch.from_numpy(np.random.default_rng(0).integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
for _ in range(8): out = enc.forward(ids, mask, pooling=pooling)
torch.cuda.synchronize()
