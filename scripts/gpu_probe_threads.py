#!/usr/bin/env python3
"""Cost of the first search on a fresh host thread (thread-per-request servers) vs a warm thread."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.index import HipIndex
ix = HipIndex(384, 200000, dtype="f32", metric="cosine", device=0)
ix.generate(seed=1, n=200000)
q = ix.fetch(np.arange(1))
for _ in range(5):
    ix.search(q, 4)
t0 = time.perf_counter()
for _ in range(200):
    ix.search(q, 4)
print(f"warm thread: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per search")
lat = []
def one():
    t = time.perf_counter(); ix.search(q, 4); lat.append((time.perf_counter() - t) * 1e3)
for _ in range(100):
    th = threading.Thread(target=one); th.start(); th.join()
lat = np.array(lat)
print(f"fresh thread per search: median {np.median(lat):.3f} ms, p90 {np.percentile(lat, 90):.3f} ms, max {lat.max():.3f} ms")
