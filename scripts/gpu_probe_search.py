#!/usr/bin/env python3
"""Device-resident searches of one shape (run under rocprofv3 --kernel-trace --stats): rows dim dtype Q"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from archi_amd.index import HipIndex
from archi_amd.sharded import HipLocalSearch

n, d, dtype, nq = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
ix = HipIndex(d, n, dtype=dtype, metric="cosine", device=0)
ix.generate(seed=1234, n=n)
tmp = HipIndex(d, nq, dtype=dtype, metric="cosine", device=0)
tmp.generate(seed=4321, n=nq, stream=1)
q = torch.from_numpy(tmp.fetch(np.arange(nq))).cuda()
tmp.close()
loc = HipLocalSearch(ix)
for _ in range(5):
    loc(q, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    loc(q, 10)
torch.cuda.synchronize()
print(f"{n}x{d} {dtype} Q={nq}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per search, plan {ix.scan_plan(nq, 10)}")
ix.close()
