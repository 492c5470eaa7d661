#!/usr/bin/env python3
"""Soak of the encoder against the torch-fp32 oracle on random batches: model shape, batch size from a few rows to the bench's
size, padded length, ragged / left-padded / holed masks, pooling. Large batches are checked on sampled rows (a row's embedding
does not depend on its neighbours). Kernel-selection switches are read once per process: run it again under
AK_ENC_LAZYLN=2 AK_ENC_SKINNY_MAX=0 (lazy LayerNorm + tile kernels at every size) and AK_FFN_NWV=8 AK_ENC_SKINNY_MAX=0.
  python3 scripts/gpu_soak_enc.py [seconds] [seed] [strong]
Without `strong` the weights are the oracle's synthetic ones (LayerNorm parameters within 5 % of (1, 0), like the suite's) and the
suite's bf16 tolerances apply: cosine >= 1 - 1e-4, max |diff| <= 2e-3 (x sqrt(384 / dim) below 384). With `strong` the LayerNorm
weights are drawn from [0.5, 1.8] / N(0, 0.3): they amplify the bf16 rounding of everything they multiply, the run reports the worst
case and holds it to twice those bounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.encoder import HipEncoder
from oracle import encoder_oracle as eo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
strong = len(sys.argv) > 3 and sys.argv[3] == "strong"
COS_TOL, ABS_TOL = (2e-4, 4e-3) if strong else (1e-4, 2e-3)
# hidden 768 (bge-base): 12 layers of bf16 activations have a tail beyond the suite's 1e-4 / 2e-3 on random weight seeds (round 4 soaks):
# the STATED bf16 tolerance of that shape is 3e-4 / 3e-3 (worst seen over the round-4 and round-5 soaks: 2.9e-4 / 2.6e-3); cases beyond the suite's are counted beside it
TOL = {"minilm-l6": (COS_TOL, ABS_TOL), "bge-base": (max(COS_TOL, 3e-4), max(ABS_TOL, 3e-3))}
beyond_suite = 0
worst_abs = {"minilm-l6": 0.0, "bge-base": 0.0}
t_end = time.time() + budget
encs = {}
cases = bad = 0
worst = {"minilm-l6": 0.0, "bge-base": 0.0}
while time.time() < t_end:
    shape = "minilm-l6" if rng.random() < 0.55 else "bge-base"
    vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
    if shape not in encs:
        w = eo.synth_weights(shape, seed=int(rng.integers(1, 1000)))
        for k in list(w) if strong else []:                 # LayerNorm parameters away from (1, 0)
            if k.endswith("_g"): w[k] = rng.uniform(0.5, 1.8, size=H).astype(np.float32)
            elif k.endswith("ln1_b") or k.endswith("ln2_b"): w[k] = rng.normal(0, 0.3, size=H).astype(np.float32)
        encs[shape] = (HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0), w)
    enc, w = encs[shape]
    S = int(rng.choice([32, 64, 96, 128, 160, 256, 384, 512]))
    big = rng.random() < 0.3
    max_tok = (70000 if shape == "minilm-l6" else 50000) if big else 6000
    B = int(rng.integers(1, max(2, max_tok // S)))
    ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
    kind = int(rng.integers(0, 4))
    lens = rng.integers(1, S + 1, size=B)
    lens[0] = S
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
    if kind == 1: mask = mask[:, ::-1].copy()               # left-padded
    if kind == 2 and S >= 96: mask[:, 32:64] = 0            # a dead 32-key block between live ones
    if kind == 3: mask[:] = 1
    mask[:, 0] = 1 if kind != 1 else mask[:, 0]
    mask[np.arange(B), np.argmax(mask, 1)] = 1
    pooling = "mean" if rng.random() < 0.5 else "cls"
    if pooling == "cls": mask[:, 0] = 1
    got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
    pick = np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, size=4)]))
    want = eo.forward(shape, w, ids[pick], mask[pick], pooling=pooling)
    cos = (got[pick] * want).sum(1)
    err = float(np.abs(got[pick] - want).max())
    ok = cos.min() >= 1 - TOL[shape][0] and err <= TOL[shape][1] and np.isfinite(got).all()
    cases += 1; bad += (not ok)
    beyond_suite += not (cos.min() >= 1 - COS_TOL and err <= ABS_TOL)
    worst[shape] = max(worst[shape], float(1 - cos.min()))
    worst_abs[shape] = max(worst_abs[shape], err)
    if not ok or cases % 10 == 0:
        print(f"{'ok ' if ok else 'BAD'} {shape} B={B} S={S} mask={kind} {pooling}: min cos 1-{1 - cos.min():.1e} max|diff| {err:.1e}", flush=True)
print(f"encoder soak ({'strong' if strong else 'suite'} LayerNorm weights, residual {os.environ.get('ARCHI_ENCODER_RESIDUAL', 'bf16')}; tolerance by shape {TOL}): "
      f"{cases} cases, {bad} outside ({beyond_suite} beyond the suite's {COS_TOL:g} / {ABS_TOL:g}); worst 1 - cos: {worst}; worst max|diff|: {worst_abs}")
sys.exit(1 if bad else 0)
