"""float32 parity mode of the encoder (csrc/encoder_f32.hip): time per forward at the bench's batch sizes against the 157.3 TFLOP/s
float32 matrix roof, and -- ARCHI_HIP_DBG=1 AK_F32_SCALAR=1 in a child process -- the scalar-fmaf kernels it replaced on a small batch
(same weights, same ids): max |difference| of the embeddings.   python3 scripts/gpu_probe_f32.py [minilm|bge|both]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

def small(out):
    from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
    res = {}
    for key, name, B in (("minilm", "sentence-transformers/all-MiniLM-L6-v2", 6), ("bge", "BAAI/bge-base-en", 3)):
        vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[name]
        enc = HipEncoder(vocab, H, 2, heads, I, max_pos, random_init_weights(vocab, H, 2, I, max_pos, seed=3), device=0, precision="f32")
        rng = np.random.default_rng(9)
        ids = rng.integers(1000, 30000, size=(B, 160)).astype(np.int32)
        lens = rng.integers(1, 161, size=B); lens[0] = 160
        mask = (np.arange(160)[None, :] < lens[:, None]).astype(np.int32)
        res[key] = enc.forward(ids, mask, pooling=pooling).cpu().numpy()
        enc.close()
    np.savez(out, **res)

if len(sys.argv) > 2 and sys.argv[1] == "--small":
    small(sys.argv[2]); sys.exit(0)

from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for key, name, B, steps in (("minilm", "sentence-transformers/all-MiniLM-L6-v2", 256, 5), ("bge", "BAAI/bge-base-en", 128, 3)):
    if which not in ("both", key): continue
    vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[name]
    enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=0, precision="f32")
    rng = np.random.default_rng(5)
    ids = torch.from_numpy(rng.integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
    mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
    for _ in range(2): enc.forward(ids, mask, pooling=pooling)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): enc.forward(ids, mask, pooling=pooling)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    fl = S * L * (2 * (4 * H * H + 2 * H * I) + 4 * S * H)
    tf = B * fl / (ms * 1e-3) / 1e12
    print(f"{key} f32 {B} x {S}: {ms:.2f} ms per forward, {B / ms * 1e3:.0f} chunks/s, {tf:.1f} TFLOP/s = {tf / 157.3:.3f} of the float32 matrix roof", flush=True)
    enc.close()
a = "/tmp/f32_mfma.npz"; b = "/tmp/f32_scalar.npz"
env = {k: v for k, v in os.environ.items() if not k.startswith("AK_")}
subprocess.check_call([sys.executable, os.path.abspath(__file__), "--small", a], env=env)
subprocess.check_call([sys.executable, os.path.abspath(__file__), "--small", b], env=dict(env, ARCHI_HIP_DBG="1", AK_F32_SCALAR="1"))
x, y = np.load(a), np.load(b)
for k in x.files:
    print(f"{k}: MFMA kernels vs scalar kernels, max |diff| {np.abs(x[k] - y[k]).max():.3e}, bit-equal rows {int((x[k] == y[k]).all(1).sum())}/{len(x[k])}")
