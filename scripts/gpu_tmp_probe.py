import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.index import HipIndex
for (n, d, dt, qs) in ((1_000_000, 384, "f32", (1, 16, 256)), (10_000_000, 768, "bf16", (1, 256, 1024))):
    ix = HipIndex(d, n, dtype=dt, metric="cosine", device=0); ix.generate(seed=1234, n=n, normalise=True)
    for nq in qs:
        tmp = HipIndex(d, nq, dtype="f32", metric="cosine", device=0); tmp.generate(seed=4321, n=nq, stream=1)
        q = tmp.fetch(np.arange(nq)); tmp.close()
        r = ix.search(q, 10, return_stats=True)
        st = r[3]
        print(f"{n}x{d} {dt} Q={nq}: survivors per query {st['reranked'] / nq:.1f}, appended per query {st['second_chance'] / nq:.1f}, certified {st['certified']}", flush=True)
    ix.close()
