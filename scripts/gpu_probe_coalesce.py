"""Request coalescing on the read path: one query per ak_index_search call from T request threads against the serial rate
(1M x 384 f32). AK_COALESCE_STATS=1 prints requests per launch; AK_COALESCE=0 / AK_COALESCE_WINDOW_US tune it."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AK_COALESCE_STATS", "1")
import numpy as np
from archi_amd.index import HipIndex
n, dim, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 384, 10
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
for T in (1, 2, 4, 8, 16, 32, 64):
    ix = HipIndex(dim, n, dtype=dtype, metric="cosine"); ix.generate(seed=1234, n=n, normalise=True)
    tmp = HipIndex(dim, 4096, dtype="f32", metric="cosine"); tmp.generate(seed=4321, n=4096, stream=1)
    qs = tmp.fetch(np.arange(4096)); tmp.close()
    for i in range(8): ix.search(qs[i:i + 1], k)
    per = max(2048 // T, 16)
    def worker(t):
        for j in range(per):
            i = (t * per + j) % 4096
            ix.search(qs[i:i + 1], k)
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    print(f"{n}x{dim} {dtype} threads={T:3d}: {T * per / dt:9.0f} q/s  ({dt / (T * per) * 1e6:7.1f} us per request)", flush=True)
    ix.close()
