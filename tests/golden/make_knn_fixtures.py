#!/usr/bin/env python3
"""Golden kNN / L2-normalise vectors: small seeded cases with their expected outputs,
computed by the pure-numpy restatement of pgvector's float32 arithmetic (oracle/knn_oracle.py: search_numpy,
sequential float32 accumulation) -- NOT by the C oracle, so the fixtures pin the C oracle and the HIP path alike.

Inputs are regenerated from the recorded seeds (`inputs()` below is imported by the tests); the files hold
the expected ids / float8 distances plus a checksum of the inputs, so they stay a few KB.

    python tests/golden/make_knn_fixtures.py        # writes tests/golden/knn_*.npz, l2norm_*.npz
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import knn_oracle as ko  # noqa: E402

CASES = {
    # name: (dtype, N, D, Q, k, seed)
    "knn_f32_N4096_D384_Q16_k10": ("f32", 4096, 384, 16, 10, 101),
    "knn_bf16_N8192_D768_Q16_k10": ("bf16", 8192, 768, 16, 10, 102),
    "knn_f16_N8192_D384_Q32_k10": ("f16", 8192, 384, 32, 10, 103),
    # adversarial: duplicated rows (exact ties), a zero row (NaN cosine distance), k > N, k = 1
    "knn_adv_ties_nan_N300_D64_Q8_k12": ("f32", 300, 64, 8, 12, 104),
    "knn_adv_k_gt_n_N7_D32_Q3_k10": ("bf16", 7, 32, 3, 10, 105),
    "knn_adv_k1_N1000_D128_Q5_k1": ("f16", 1000, 128, 5, 1, 106),
}
METRICS = ("cosine", "l2", "inner_product")


def inputs(name):
    """(stored rows as float32 values of the storage dtype, queries float32, ids int64)."""
    dtype, n, d, q, k, seed = CASES[name]
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    qs = rng.standard_normal((q, d)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    if "ties_nan" in name:
        rows[10] = rows[3]; rows[11] = rows[3]; rows[200] = rows[150]      # exact ties -> id order decides
        rows[42] = 0.0                                                     # zero vector: NaN cosine distance
        qs[1] = rows[3]                                                    # a query equal to a tied row
        qs[2] = 0.0                                                        # zero query: every cosine distance NaN
    ids = (rng.permutation(5 * n)[:n] + 1).astype(np.int64)
    stored = ko.round_through(rows.astype(np.float32), dtype)
    return stored, qs.astype(np.float32), ids


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def l2_inputs(seed=201, n=24, d=384):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n, d)) * np.exp(rng.uniform(-6, 6, (n, 1)))).astype(np.float32)
    x[5] = 0.0                      # zero row: x / max(||x||, 1e-12) stays zero
    x[6] *= np.float32(1e-20)       # norm below the epsilon
    return x


def main():
    for name, (dtype, n, d, q, k, seed) in CASES.items():
        stored, qs, ids = inputs(name)
        out = {"dtype": dtype, "inputs_sha256": checksum(stored, qs, ids)}
        for metric in METRICS:
            oi, od = ko.search_numpy(stored, qs, k, metric, ids=ids)
            out[f"ids_{metric}"] = oi
            out[f"dist_{metric}"] = od
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name)
    x = l2_inputs()
    # torch.nn.functional.normalize semantics in float32: x / max(||x||_2, eps), eps = 1e-12
    nrm = np.sqrt((x.astype(np.float64) ** 2).sum(1)).astype(np.float32)
    want = (x / np.maximum(nrm, np.float32(1e-12))[:, None]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "l2norm_N24_D384.npz"), inputs_sha256=checksum(x), expected=want)
    print("wrote l2norm_N24_D384")


if __name__ == "__main__":
    main()
