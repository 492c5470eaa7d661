"""First GPU probe: exact-path timing at scale."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.index import HipIndex
from oracle import knn_oracle as ko

for (n, d, dtype) in [(1_000_000, 384, "f32"), (2_000_000, 768, "bf16")]:
    t = time.time(); ix = HipIndex(d, n, dtype=dtype, metric="cosine", device=0)
    ix.generate(seed=1234, n=n); print(f"generate {n}x{d} {dtype}: {time.time()-t:.3f}s", flush=True)
    q = ko.gen_rows(4321, 1, 0, 16, d, True, "f32")
    for nq in (1, 8, 16):
        ix.search(q[:nq], 10, mode="exact")
        t = time.time(); ids, dist, _ = ix.search(q[:nq], 10, mode="exact"); dt = time.time() - t
        print(f"exact search nq={nq}: {dt*1e3:.1f} ms", flush=True)
    # oracle check on a slice-free basis: compare with oracle over first 200k rows restricted
    m = 200_000
    sub = ko.gen_rows(1234, 0, 0, m, d, True, dtype)
    flt = np.zeros(n, np.uint8); flt[:m] = 1
    gi, gd, _ = ix.search(q[:4], 10, mode="exact", row_filter=flt)
    oi, od, _ = ko.search(sub, q[:4], 10, "cosine")
    print("parity on 200k slice:", np.array_equal(gi, oi), np.array_equal(gd, od), flush=True)
    ix.close()
