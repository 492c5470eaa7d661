#!/usr/bin/env python3
"""Single-query and small-batch search latency on SMALL collections (what a real archi deployment holds): certified MFMA path
(mode auto) against the exact path (reference arithmetic for every row), host-buffer entry point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.index import HipIndex
from oracle import knn_oracle as ko
for d, dtype in ((384, "f32"), (768, "f32")):
    for n in (4096, 8192, 20000, 50000, 100000, 300000, 1000000):
        ix = HipIndex(d, n, dtype=dtype, metric="cosine", device=0)
        ix.generate(seed=1234, n=n)
        for nq in (1, 8):
            q = ko.gen_rows(4321, 1, 0, nq, d, True, "f32")
            out = []
            for mode in ("auto", "exact"):
                for _ in range(5): ix.search(q, 10, mode=mode)
                t0 = time.perf_counter()
                for _ in range(30): ix.search(q, 10, mode=mode)
                out.append((time.perf_counter() - t0) / 30 * 1e3)
            print(f"{n:8d} x {d} {dtype} Q={nq}: auto {out[0]:.3f} ms   exact {out[1]:.3f} ms")
        ix.close()
