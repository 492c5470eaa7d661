"""GPU probe: chunk-embeds/s of the HIP encoder (synthetic weights/tokens)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.encoder import HipEncoder, MODEL_SHAPES, random_init_weights

def flops_per_token(H, I, L, S):
    return L * (2 * (4 * H * H + 2 * H * I) + 4 * S * H)

for name, B, S in (("sentence-transformers/all-MiniLM-L6-v2", 256, 256), ("BAAI/bge-base-en", 128, 512),
                   ("sentence-transformers/all-MiniLM-L6-v2", 1024, 256)):
    vocab, H, L, heads, I, max_pos, pooling, _ = MODEL_SHAPES[name]
    enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=0)
    rng = np.random.default_rng(0)
    ids = torch.from_numpy(rng.integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
    mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
    for _ in range(2): out = enc.forward(ids, mask, pooling=pooling)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): out = enc.forward(ids, mask, pooling=pooling)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = flops_per_token(H, I, L, S) * B * S
    print(f"{name} B={B} S={S}: {ms:.3f} ms/batch  {B/ms*1e3:.0f} chunks/s  {fl/ms/1e9:.1f} TFLOP/s", flush=True)
    enc.close()
