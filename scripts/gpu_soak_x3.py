#!/usr/bin/env python3
"""Soak of precision="bf16x3" against the torch-fp32 oracle at the parity bar (1e-5 per component of the unit rows): both encoder
shapes with full-mantissa weights (so the lo halves of the weight split matter), batches on either side of the switch (16 384 tokens at hidden 768, 20 480 at hidden 384)
between k3_gemm and gemm.hip's tiles, token counts that are not multiples of 256 (padded tiles), ragged / left-padded / holed masks,
both poolings, ONE encoder per shape for the whole run (the workspace is re-laid-out whenever the padded token count changes, so
its padding rows hold whatever an earlier batch left there). Large batches are checked on sampled rows.
  python3 scripts/gpu_soak_x3.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.encoder import HipEncoder
from oracle import encoder_oracle as eo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
TOL = 1e-5
encs = {}
cases = bad = tiles = 0
worst = {"minilm-l6": 0.0, "bge-base": 0.0}
t_end = time.time() + budget
while time.time() < t_end:
    shape = "minilm-l6" if rng.random() < 0.6 else "bge-base"
    vocab, H, L, heads, I, max_pos, _ = eo.SHAPES[shape]
    if shape not in encs:
        w = eo.synth_weights(shape, seed=int(rng.integers(1, 1000)))
        w = {k: (v * (1.0 + 1e-3 * rng.standard_normal(v.shape))).astype(np.float32) if v.ndim == 2 else v for k, v in w.items()}
        encs[shape] = (HipEncoder(vocab, H, L, heads, I, max_pos, w, device=0, precision="bf16x3"), w)
    enc, w = encs[shape]
    S = int(rng.choice([32, 40, 64, 96, 100, 128, 160, 256, 384, 512]))
    max_tok = int(rng.choice([3000, 9000, 40000 if shape == "minilm-l6" else 24000, 60000 if shape == "minilm-l6" else 34000]))
    B = int(rng.integers(1, max(2, max_tok // S)))
    ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
    kind = int(rng.integers(0, 4))
    lens = rng.integers(1, S + 1, size=B)
    lens[0] = S
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
    if kind == 1: mask = mask[:, ::-1].copy()               # left-padded
    if kind == 2 and S >= 96: mask[:, 32:64] = 0            # a dead 32-key block between live ones
    if kind == 3: mask[:] = 1
    mask[np.arange(B), np.argmax(mask, 1)] = 1
    pooling = "mean" if rng.random() < 0.5 else "cls"
    if pooling == "cls": mask[:, 0] = 1
    got = enc.forward(ids, mask, pooling=pooling, normalise=True).cpu().numpy()
    pick = np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, size=4)]))
    want = eo.forward(shape, w, ids[pick], mask[pick], pooling=pooling)
    err = float(np.abs(got[pick] - want).max())
    worst[shape] = max(worst[shape], err)
    cases += 1
    tiles += B * S >= (16384 if shape == "bge-base" else 20480)
    if not np.isfinite(got).all() or err > TOL:
        bad += 1
        print(f"MISMATCH {shape} B={B} S={S} kind={kind} {pooling}: max |diff| {err:.3e}, finite {bool(np.isfinite(got).all())}", flush=True)
print(f"bf16x3 soak: {cases} batches ({tiles} on the GEMM tiles), {bad} beyond {TOL:g}; worst max |diff| MiniLM {worst['minilm-l6']:.2e}, bge-base {worst['bge-base']:.2e}")
sys.exit(1 if bad else 0)
