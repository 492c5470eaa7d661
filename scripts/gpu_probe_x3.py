"""Split-bf16 parity mode (precision="bf16x3") against the float32 mode and the default bf16 path: forward time of both encoder
shapes at the bench's batch sizes + max |delta| against the float32 mode on the same weights. python scripts/gpu_probe_x3.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
for name, B in (("sentence-transformers/all-MiniLM-L6-v2", 256), ("BAAI/bge-base-en", 128)):
    vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[name]
    w = random_init_weights(vocab, H, L, I, max_pos, seed=0)
    rng = np.random.default_rng(5)
    ids = torch.from_numpy(rng.integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
    mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
    outs = {}
    for prec in ("f32", "bf16x3", "bf16"):
        enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0, precision=prec)
        for _ in range(2):
            o = enc.forward(ids, mask, pooling=pooling)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            o = enc.forward(ids, mask, pooling=pooling)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        outs[prec] = o.cpu().numpy()
        print(f"{name} {B}x{S} {prec}: {ms:.2f} ms per forward = {B / ms * 1e3:.0f} chunks/s", flush=True)
        enc.close()
    for prec in ("bf16x3", "bf16"):
        d = np.abs(outs[prec] - outs["f32"]).max()
        sc = np.abs(outs[prec][:64] @ outs[prec][:64].T - outs["f32"][:64] @ outs["f32"][:64].T).max()
        print(f"  {prec} vs f32: max |delta component| {d:.2e}, max |delta score| {sc:.2e}")
