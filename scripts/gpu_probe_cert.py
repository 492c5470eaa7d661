#!/usr/bin/env python3
"""Which (dtype, metric, k) combinations certify in the first MFMA pass, which need the wide second pass or the exact path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from archi_amd.index import HipIndex
from oracle import knn_oracle as ko

n, d, nq = 1_000_000, 256, 100
for dtype in ("f32", "bf16", "f16"):
    for metric in ("cosine", "l2", "inner_product"):
        ix = HipIndex(d, n, dtype=dtype, metric=metric, device=0)
        ix.generate(seed=5, n=n, normalise=True)
        q = ko.gen_rows(6, 1, 0, nq, d, True, "f32")
        for k in (10, 33, 64, 100):
            _, _, _, st = ix.search(q, k, mode="auto", return_stats=True)
            print(f"{dtype:5s} {metric:14s} k={k:3d}: {st}  plan kprime={ix.scan_plan(nq, k)['kprime']}")
        ix.close()
