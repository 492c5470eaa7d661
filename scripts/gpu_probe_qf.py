"""embed_query's forward pass: the single launch on one XCD (csrc/query_forward.hip) against the 47-launch path -- host wall per
call of HipEncoder.forward on [1, 32] and [1, 64] token rows, and a soak: python scripts/gpu_probe_qf.py [soak_seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd import _lib
from archi_amd.encoder import HipEncoder
from oracle import encoder_oracle as eo
soak = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
shape = "minilm-l6"
vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
w = eo.synth_weights(shape, seed=7)
enc = HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0)
for S in (32, 64):
    ids, mask = eo.synth_tokens(1, S, seed=3, vocab=vocab)
    i_d, m_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    res = {}
    for mode in ("0", "2", "0", "2"):
        _lib.debug_set("AK_QUERY_FUSED", mode)
        for _ in range(20):
            out = enc.forward(i_d, m_d)
        torch.cuda.synchronize()
        ts = []
        for _ in range(300):
            t0 = time.perf_counter(); out = enc.forward(i_d, m_d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res.setdefault(mode, []).append((np.median(ts) * 1e3, np.percentile(ts, 99) * 1e3, out.cpu().numpy()))
    a, b = res["0"][-1], res["2"][-1]
    print(f"[1,{S}] forward: 47 launches p50 {a[0]:.3f} ms p99 {a[1]:.3f} | single launch p50 {b[0]:.3f} ms p99 {b[1]:.3f} | identical {np.array_equal(a[2], b[2])}", flush=True)
if soak > 0:
    rng = np.random.default_rng(1)
    _lib.debug_set("AK_QUERY_FUSED", "2")
    t_end, n, bad = time.time() + soak, 0, 0
    while time.time() < t_end:
        S = int(rng.choice([32, 64])); B = 1 if S == 64 or rng.random() < 0.7 else 2
        ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
        lens = rng.integers(1, S + 1, size=B)
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
        pooling = "mean" if rng.random() < 0.5 else "cls"
        got = enc.forward(ids, mask, pooling=pooling).cpu().numpy()
        if n % 50 == 0:
            _lib.debug_set("AK_QUERY_FUSED", "0")
            want = enc.forward(ids, mask, pooling=pooling).cpu().numpy()
            _lib.debug_set("AK_QUERY_FUSED", "2")
            bad += not np.array_equal(got, want)
        n += 1
    print(f"soak: {n} single-launch forwards in {soak:.0f} s, every 50th compared with the 47-launch path: {bad} differences, no hang, no give-up")
enc.close()
