#!/usr/bin/env python3
"""Ingestion-side cost of ak_index_add: many small adds (one per file) and one large add."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.index import HipIndex
d = 384
rng = np.random.default_rng(0)
rows = rng.standard_normal((25, d)).astype(np.float32)
ix = HipIndex(d, 2_000_000, dtype="f32", metric="cosine", device=0)
for _ in range(20):
    ix.add(rows, ids=None)
t0 = time.perf_counter(); n = 2000
for i in range(n):
    ix.add(rows, ids=np.arange(1_000_000 + 25 * i, 1_000_000 + 25 * (i + 1)))
dt = time.perf_counter() - t0
print(f"{n} adds of 25 rows: {dt / n * 1e3:.3f} ms per add -> {25 * n / dt:.0f} rows/s")
big = rng.standard_normal((500_000, d)).astype(np.float32)
t0 = time.perf_counter(); ix.add(big, ids=np.arange(5_000_000, 5_500_000)); dt = time.perf_counter() - t0
print(f"one add of 500000 rows: {dt * 1e3:.1f} ms -> {500000 / dt:.0f} rows/s ({big.nbytes / dt / 1e9:.1f} GB/s of float32 in)")
ix.close()
