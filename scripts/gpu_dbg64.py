import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.index import HipIndex
from oracle import knn_oracle as ko
for (n, d, nq) in [(5000, 64, 130), (5000, 128, 130), (5000, 64, 30), (20000, 64, 130)]:
    ix = HipIndex(d, n, dtype="bf16", metric="cosine", device=0); ix.generate(seed=1234, n=n)
    stored = ko.gen_rows(1234, 0, 0, n, d, True, "bf16")
    q = ko.gen_rows(4321, 1, 0, nq, d, True, "f32")
    gi, gd, gc, st = ix.search(q, 10, mode="fast_only", return_stats=True)
    oi, od, oc = ko.search(stored, q, 10, "cosine")
    bad = np.nonzero((gi != oi).any(1))[0]
    print(f"n={n} d={d} nq={nq} plan={ix.scan_plan(nq,10)} stats={st} bad_queries={len(bad)} first={bad[:10]}")
    for b in bad[:3]:
        print("   q", b, "got", gi[b], "want", oi[b]); print("      gd", gd[b][:4], "od", od[b][:4])
    ix.close()
