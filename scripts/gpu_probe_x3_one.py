"""A few split-bf16 (precision="bf16x3") MiniLM-shape forwards: the workload for scripts/gpu_pmc_kernel.sh k3_gemm / k3_attn."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
name = sys.argv[1] if len(sys.argv) > 1 else "sentence-transformers/all-MiniLM-L6-v2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[name]
enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=0, precision="bf16x3")
ids = torch.from_numpy(np.random.default_rng(5).integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
import time
for _ in range(3):
    enc.forward(ids, mask, pooling=pooling)
torch.cuda.synchronize()
if os.environ.get("X3_TIME"):
    n = 8
    t0 = time.perf_counter()
    for _ in range(n):
        enc.forward(ids, mask, pooling=pooling)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name} {B}x{S} bf16x3 [{os.environ.get('X3_TAG', '')}]: {ms:.2f} ms per forward = {B / ms * 1e3:.0f} chunks/s", flush=True)
enc.close()
