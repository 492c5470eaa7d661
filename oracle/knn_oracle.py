"""ctypes front-end of the CPU oracle (oracle/knn_oracle.c).

TEST INFRASTRUCTURE ONLY. Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from archi_amd/ (the product path fails
loudly when the HIP library is missing; it has no CPU fallback).

Also holds `search_numpy`, an independent pure-numpy restatement of the same
arithmetic (strictly sequential float32 sums) used to cross-check the C code on
small cases.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libknn_oracle.so")

METRICS = {"cosine": 0, "l2": 1, "inner_product": 2}
DTYPES = {"f32": 0, "bf16": 1, "f16": 2}

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "knn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        L = _lib
        c_i64, c_int, c_dbl = ctypes.c_int64, ctypes.c_int, ctypes.c_double
        P = ctypes.c_void_p
        L.ako_distance.restype = c_dbl
        L.ako_distance.argtypes = [c_int, c_int, P, P]
        L.ako_distance_f64.restype = c_dbl
        L.ako_distance_f64.argtypes = [c_int, c_int, P, P]
        L.ako_search.restype = c_int
        L.ako_search.argtypes = [c_int, c_i64, c_int, P, P, P, c_int, P, c_int, P, P, P]
        L.ako_merge.restype = c_int
        L.ako_merge.argtypes = [c_int, c_int, c_int, P, P, P, P]
        L.ako_l2_normalize.restype = None
        L.ako_l2_normalize.argtypes = [c_i64, c_int, P]
        L.ako_round_through.restype = None
        L.ako_round_through.argtypes = [c_int, c_i64, P, P]
        L.ako_gen_rows.restype = None
        L.ako_gen_rows.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, c_i64, c_int,
                                   c_int, c_int, P]
        L.ako_gen_int.restype = c_int
        L.ako_gen_int.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, c_int]
        L.ako_philox4x32_10.restype = None
        L.ako_philox4x32_10.argtypes = [P, P, P]
        L.ako_f32_to_bf16.restype = ctypes.c_uint16
        L.ako_f32_to_bf16.argtypes = [ctypes.c_float]
        L.ako_f32_to_f16.restype = ctypes.c_uint16
        L.ako_f32_to_f16.argtypes = [ctypes.c_float]
        L.ako_f16_to_f32.restype = ctypes.c_float
        L.ako_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.ako_bf16_to_f32.restype = ctypes.c_float
        L.ako_bf16_to_f32.argtypes = [ctypes.c_uint16]
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def round_through(x: np.ndarray, dtype: str) -> np.ndarray:
    """fp32 values as the index stores them (RNE through bf16 / f16)."""
    x = _f32(x)
    out = np.empty_like(x)
    lib().ako_round_through(DTYPES[dtype], x.size, _p(x), _p(out))
    return out


def distance(metric: str, a, b) -> float:
    a, b = _f32(a), _f32(b)
    return lib().ako_distance(METRICS[metric], a.size, _p(a), _p(b))


def distance_f64(metric: str, a, b) -> float:
    a, b = _f32(a), _f32(b)
    return lib().ako_distance_f64(METRICS[metric], a.size, _p(a), _p(b))


def search(corpus, queries, k: int, metric: str = "cosine", ids=None, alive=None
           ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Exact top-k. Returns (ids [Q,k] int64, distances [Q,k] float64, counts [Q])."""
    corpus, queries = _f32(corpus), _f32(queries)
    if queries.ndim == 1:
        queries = queries[None, :]
    n, d = corpus.shape if corpus.ndim == 2 else (0, queries.shape[1])
    q = queries.shape[0]
    ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
    alive_a = None if alive is None else np.ascontiguousarray(alive, dtype=np.uint8)
    out_ids = np.empty((q, k), dtype=np.int64)
    out_d = np.empty((q, k), dtype=np.float64)
    cnt = np.empty((q,), dtype=np.int32)
    rc = lib().ako_search(METRICS[metric], n, d, _p(corpus), _p(ids_a), _p(alive_a), q, _p(queries), k,
                          _p(out_ids), _p(out_d), _p(cnt))
    if rc != 0:
        raise RuntimeError(f"ako_search failed rc={rc}")
    return out_ids, out_d, cnt


def merge(part_ids: np.ndarray, part_dist: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Merge [G,Q,k] per-shard partials with (distance asc, NaN last, id asc)."""
    part_ids = np.ascontiguousarray(part_ids, dtype=np.int64)
    part_dist = np.ascontiguousarray(part_dist, dtype=np.float64)
    g, q, k = part_ids.shape
    out_ids = np.empty((q, k), dtype=np.int64)
    out_d = np.empty((q, k), dtype=np.float64)
    rc = lib().ako_merge(g, q, k, _p(part_ids), _p(part_dist), _p(out_ids), _p(out_d))
    if rc != 0:
        raise RuntimeError(f"ako_merge failed rc={rc}")
    return out_ids, out_d


def l2_normalize(x) -> np.ndarray:
    x = _f32(x).copy()
    lib().ako_l2_normalize(x.shape[0], x.shape[1], _p(x))
    return x


def gen_rows(seed: int, stream: int, row0: int, n: int, dim: int, normalise: bool = True,
             dtype: str = "f32") -> np.ndarray:
    out = np.empty((n, dim), dtype=np.float32)
    lib().ako_gen_rows(seed, stream, row0, n, dim, int(normalise), DTYPES[dtype], _p(out))
    return out


def philox(ctr, key) -> np.ndarray:
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    o = np.empty(4, dtype=np.uint32)
    lib().ako_philox4x32_10(_p(c), _p(k), _p(o))
    return o


# ---------------------------------------------------------------------------
# Independent pure-numpy restatement (small cases only): sequential float32.
# ---------------------------------------------------------------------------
def _seq_sum_f32(terms: np.ndarray) -> np.ndarray:
    """Strictly sequential float32 sum along the last axis."""
    acc = np.zeros(terms.shape[:-1], dtype=np.float32)
    for i in range(terms.shape[-1]):
        acc = (acc + terms[..., i]).astype(np.float32)
    return acc


def distances_numpy(corpus, query, metric: str) -> np.ndarray:
    c, q = _f32(corpus), _f32(query)
    if metric == "l2":
        diff = (c - q[None, :]).astype(np.float32)
        return np.sqrt(_seq_sum_f32((diff * diff).astype(np.float32)).astype(np.float64))
    dot = _seq_sum_f32((c * q[None, :]).astype(np.float32))
    if metric == "inner_product":
        return (-dot).astype(np.float64)
    na = _seq_sum_f32((c * c).astype(np.float32)).astype(np.float64)
    nb = np.float64(_seq_sum_f32((q * q).astype(np.float32)[None, :])[0])
    with np.errstate(invalid="ignore", divide="ignore"):
        sim = dot.astype(np.float64) / np.sqrt(na * nb)
    sim = np.where(sim > 1.0, 1.0, np.where(sim < -1.0, -1.0, sim))
    return 1.0 - sim


def search_numpy(corpus, queries, k: int, metric: str = "cosine", ids=None, alive=None):
    corpus, queries = _f32(corpus), _f32(queries)
    n = corpus.shape[0]
    ids_a = np.arange(n, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
    out_i = np.full((queries.shape[0], k), -1, dtype=np.int64)
    out_d = np.full((queries.shape[0], k), np.nan, dtype=np.float64)
    for qi, q in enumerate(queries):
        d = distances_numpy(corpus, q, metric)
        rows = np.arange(n) if alive is None else np.nonzero(np.asarray(alive))[0]
        # lexsort: last key is primary -> (isnan, distance, id)
        dd = d[rows]
        order = np.lexsort((ids_a[rows], np.where(np.isnan(dd), 0.0, dd), np.isnan(dd)))[:k]
        out_i[qi, : len(order)] = ids_a[rows][order]
        out_d[qi, : len(order)] = dd[order]
    return out_i, out_d
