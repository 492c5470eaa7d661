"""HipIndex -- Python handle of one HBM-resident corpus shard (ak_index_*).

Replaces the `document_chunks.embedding vector(D)` column and the pgvector
operator scan of the reference (src/cli/templates/init.sql:256-292,
src/data_manager/vectorstore/postgres_vectorstore.py:317-332).
"""
from __future__ import annotations

import ctypes
import logging
import threading
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import DTYPES, METRICS, SEARCH_MODES, HipBackendError, check


log = logging.getLogger("archi_amd.index")    # one line per index growth / compaction (the reference logs through
#                                               src/utils/logging.py:23-40; collection sizes: manager.py:174)


def _ptr(a: Optional[np.ndarray]):
    # the address as a plain int (argtypes say c_void_p): a.ctypes.data_as(...) costs 2.3 us a piece, six per search call --
    # time spent holding the interpreter lock on the request path
    return None if a is None else a.__array_interface__["data"][0]


class HipIndex:
    """One corpus shard on one GPU."""

    def __init__(self, dim: int, capacity: int, dtype: str = "bf16", metric: str = "cosine",
                 device: Optional[int] = None):
        if metric not in METRICS:
            raise ValueError(f"distance_metric must be one of {list(METRICS.keys())}")
        if dtype not in DTYPES:
            raise ValueError(f"dtype must be one of {list(DTYPES.keys())}")
        self._lib = _lib.init(device)
        self.dim, self.capacity, self.dtype, self.metric = int(dim), int(capacity), dtype, metric
        h = ctypes.c_void_p()
        check(self._lib.ak_index_create(self.capacity, self.dim, DTYPES[dtype], METRICS[metric],
                                        ctypes.byref(h)), "ak_index_create")
        self._h = h

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ak_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- writes -----------------------------------------------------------
    def add(self, rows, ids: Optional[Sequence[int]] = None, normalise: bool = False) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n,{self.dim}] float32")
        n = rows.shape[0]
        ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        if ids_a is not None and ids_a.shape != (n,):
            raise ValueError("ids must have one entry per row")
        before = (self.slots, self.allocated_rows)
        check(self._lib.ak_index_add(self._h, _ptr(rows), 0, n, _ptr(ids_a), int(normalise)), "ak_index_add")
        self._log_layout_change(before, n)

    def _log_layout_change(self, before, n_added: int) -> None:
        slots, cap = self.slots, self.allocated_rows
        if cap != before[1]:
            log.info("index %dx%d %s: buffers grew %d -> %d rows (%d slots in use)", cap, self.dim, self.dtype, before[1], cap, slots)
        elif slots != before[0] + n_added:
            log.info("index %dx%d %s: %d tombstones reclaimed by an add (%d slots in use)", cap, self.dim, self.dtype,
                     before[0] + n_added - slots, slots)

    def add_device(self, rows_ptr: int, n: int, ids: Optional[Sequence[int]] = None,
                   normalise: bool = False) -> None:
        """rows_ptr: device pointer to [n,dim] float32 (e.g. torch tensor .data_ptr())."""
        ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
        before = (self.slots, self.allocated_rows)
        check(self._lib.ak_index_add(self._h, ctypes.c_void_p(rows_ptr), 1, n, _ptr(ids_a), int(normalise)),
              "ak_index_add")
        self._log_layout_change(before, n)

    def generate(self, seed: int, n: int, stream: int = 0, row0: int = 0, normalise: bool = True,
                 id0: Optional[int] = None) -> None:
        """Append synthetic rows [row0,row0+n) (generator: oracle/knn_oracle.c ako_gen_rows)."""
        check(self._lib.ak_index_generate(self._h, seed, stream, row0, n, int(normalise),
                                          row0 if id0 is None else id0), "ak_index_generate")

    def remove(self, ids: Sequence[int]) -> int:
        ids_a = np.ascontiguousarray(ids, dtype=np.int64)
        removed = ctypes.c_int64(0)
        check(self._lib.ak_index_remove(self._h, _ptr(ids_a), ids_a.size, ctypes.byref(removed)), "ak_index_remove")
        return removed.value

    def compact(self) -> int:
        """Reclaim every tombstone now (slot numbers change). Returns the number of slots reclaimed."""
        out = ctypes.c_int64(0)
        check(self._lib.ak_index_compact(self._h, ctypes.byref(out)), "ak_index_compact")
        if out.value:
            log.info("index %dx%d %s: compaction reclaimed %d slots (%d in use)", self.allocated_rows, self.dim, self.dtype,
                     out.value, self.slots)
        return out.value

    # -- reads ------------------------------------------------------------
    @property
    def slots(self) -> int:
        """Row slots in use (live + tombstones) == the length of a row_filter. The library owns the number: adds may
        reclaim tombstones or grow the buffers."""
        out = ctypes.c_int64(0)
        check(self._lib.ak_index_slots(self._h, ctypes.byref(out), None, None), "ak_index_slots")
        return out.value

    def layout(self) -> Tuple[int, int]:
        """(row slots in use, layout epoch), read in ONE call: what a row_filter is built for and what it is handed to
        search() with. The epoch changes with every add and every reclaim of tombstones; a search given a mask of another
        layout raises StaleFilterError instead of applying it to the wrong rows."""
        n, ep = ctypes.c_int64(0), ctypes.c_uint64(0)
        check(self._lib.ak_index_slots(self._h, ctypes.byref(n), None, ctypes.byref(ep)), "ak_index_slots")
        return n.value, ep.value

    @property
    def allocated_rows(self) -> int:
        cap = ctypes.c_int64(0)
        check(self._lib.ak_index_slots(self._h, None, ctypes.byref(cap), None), "ak_index_slots")
        return cap.value

    def count(self) -> int:
        out = ctypes.c_int64(0)
        check(self._lib.ak_index_count(self._h, ctypes.byref(out)), "ak_index_count")
        return out.value

    def lookup(self, ids: Sequence[int]) -> np.ndarray:
        ids_a = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty(ids_a.shape, dtype=np.int64)
        check(self._lib.ak_index_lookup(self._h, _ptr(ids_a), ids_a.size, _ptr(out)), "ak_index_lookup")
        return out

    def distances(self, query: np.ndarray, ids: Sequence[int]):
        """Exact distances of one query to the listed ids -> (dist [n] float64, found [n] bool);
        NaN / False for absent or deleted ids."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if q.size != self.dim:
            raise ValueError(f"query has {q.size} values, index dimension is {self.dim}")
        ids_a = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty(ids_a.shape, dtype=np.float64)
        found = np.zeros(ids_a.shape, dtype=np.uint8)
        if ids_a.size:
            check(self._lib.ak_index_distances(self._h, _ptr(q), _ptr(ids_a), ids_a.size, _ptr(out), _ptr(found)),
                  "ak_index_distances")
        return out, found.astype(bool)

    def fetch(self, slots: Sequence[int]) -> np.ndarray:
        """Stored rows (exact stored values widened to float32) by row slot."""
        s = np.ascontiguousarray(slots, dtype=np.int64)
        out = np.empty((s.size, self.dim), dtype=np.float32)
        check(self._lib.ak_index_fetch(self._h, _ptr(s), s.size, _ptr(out)), "ak_index_fetch")
        return out

    def search(self, queries, k: int, mode: str = "auto", row_filter: Optional[np.ndarray] = None,
               return_stats: bool = False, filter_epoch: Optional[int] = None):
        """Top-k by ascending pgvector distance. Returns (ids [Q,k], distances [Q,k] f64, counts [Q]).
        row_filter: one byte per row slot of the layout `filter_epoch` (layout()); the library refuses a mask of another
        layout (StaleFilterError) without reading it. filter_epoch=None takes the current epoch -- for callers that know no
        writer runs beside them."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"queries must be [nq,{self.dim}] float32")
        nq = q.shape[0]
        out_ids = np.full((nq, k), -1, dtype=np.int64)
        out_d = np.full((nq, k), np.nan, dtype=np.float64)
        cnt = np.zeros((nq,), dtype=np.int32)
        stats = np.zeros(4, dtype=np.int64)
        flt, flen, fep = None, 0, 0
        if row_filter is not None:
            flt = np.ascontiguousarray(row_filter, dtype=np.uint8)
            if flt.ndim != 1:
                raise ValueError("row_filter must be one byte per row slot")
            flen = flt.shape[0]
            if filter_epoch is None:
                slots, fep = self.layout()
                if flen != slots:
                    raise ValueError(f"row_filter must have {slots} entries (one per row slot)")
            else:
                fep = int(filter_epoch)
        check(self._lib.ak_index_search(self._h, _ptr(q), nq, k, SEARCH_MODES[mode], _ptr(flt), flen, fep, _ptr(out_ids),
                                        _ptr(out_d), _ptr(cnt), _ptr(stats)), "ak_index_search")
        if return_stats:
            return out_ids, out_d, cnt, {"certified": int(stats[0]), "exact_reruns": int(stats[1]),
                                         "reranked": int(stats[2]), "second_chance": int(stats[3])}
        return out_ids, out_d, cnt

    def search_device(self, queries_ptr: int, nq: int, k: int, out_ids_ptr: int, out_dist_ptr: int,
                      out_cert_ptr: int, stream: int = 0, mode: str = "fast_only", row_filter_ptr: int = 0,
                      filter_len: int = 0, filter_epoch: Optional[int] = None) -> None:
        """Device-resident search; all pointers are device pointers. mode "fast_only": asynchronous, the certificate
        flags say which queries are proven exact; "auto": flags read back, open queries re-run on the device (every row
        exact on return); "exact": reference arithmetic for every row."""
        if row_filter_ptr and filter_epoch is None:          # a caller that knows no writer runs beside it
            filter_len, filter_epoch = self.layout()
        check(self._lib.ak_index_search_dev(self._h, ctypes.c_void_p(queries_ptr), nq, k, SEARCH_MODES[mode],
                                            ctypes.c_void_p(row_filter_ptr) if row_filter_ptr else None,
                                            int(filter_len), int(filter_epoch or 0),
                                            ctypes.c_void_p(out_ids_ptr), ctypes.c_void_p(out_dist_ptr),
                                            ctypes.c_void_p(out_cert_ptr) if out_cert_ptr else None,
                                            ctypes.c_void_p(stream)),
              "ak_index_search_dev")

    def scan_plan(self, nq: int, k: int) -> dict:
        out = np.zeros(8, dtype=np.int64)
        check(self._lib.ak_index_scan_plan(self._h, nq, k, _ptr(out)), "ak_index_scan_plan")
        names = ["fast", "cfg", "kprime", "nslices", "nqg", "ns_seed", "seed_rows", "qtile"]
        d = {n: int(v) for n, v in zip(names, out)}
        d["cfg_name"] = ["256x128", "256x64", "256x32", "128x128", "256x256 in-step", "256x256", "256x128 phased", "256x192 phased"][d["cfg"]] if d["fast"] else None
        return d

    def debug_read(self) -> np.ndarray:
        """[2 launches][8192 waves][8] phase cycle counters (needs AK_SCAN_DBG=1 during the search)."""
        out = np.zeros(2 * 65536, dtype=np.int64)
        check(self._lib.ak_index_debug_read(self._h, _ptr(out), out.size), "ak_index_debug_read")
        return out.reshape(2, 8192, 8)

    def profile(self, enable: bool) -> None:
        check(self._lib.ak_index_profile(self._h, int(enable)), "ak_index_profile")

    def profile_read(self, cap: int = 4096) -> np.ndarray:
        """Durations (ms) of the scan kernel launches since the last read (sync the stream first)."""
        out = np.empty(cap, dtype=np.float32)
        n = ctypes.c_int(0)
        check(self._lib.ak_index_profile_read(self._h, _ptr(out), cap, ctypes.byref(n)), "ak_index_profile_read")
        return out[: n.value].copy()


def merge_topk_device(g: int, nq: int, k: int, part_ids_ptr: int, part_dist_ptr: int, out_ids_ptr: int,
                      out_dist_ptr: int, stream: int = 0) -> None:
    lib = _lib.init()
    check(lib.ak_merge_topk_dev(g, nq, k, ctypes.c_void_p(part_ids_ptr), ctypes.c_void_p(part_dist_ptr),
                                ctypes.c_void_p(out_ids_ptr), ctypes.c_void_p(out_dist_ptr),
                                ctypes.c_void_p(stream)), "ak_merge_topk_dev")


def merge_shards_device(g: int, nq: int, k: int, payload_ptr: int, stride: int, out_ids_ptr: int, out_dist_ptr: int,
                        out_open_ptr: int, stream: int = 0) -> None:
    lib = _lib.init()
    check(lib.ak_merge_shards_dev(g, nq, k, ctypes.c_void_p(payload_ptr), stride, ctypes.c_void_p(out_ids_ptr),
                                  ctypes.c_void_p(out_dist_ptr), ctypes.c_void_p(out_open_ptr), ctypes.c_void_p(stream)),
          "ak_merge_shards_dev")


# ---------------------------------------------------------------------------
# process-level cache: the reference re-creates its store object on every chat
# request (src/archi/archi.py:61-65 -> vectorstore_connector.py:60-81), so the
# GPU-resident index must outlive store instances.
# ---------------------------------------------------------------------------
_cache: Dict[Tuple[str, str], HipIndex] = {}
_cache_lock = threading.Lock()


def get_or_create(collection: str, metric: str, dim: int, capacity: int, dtype: str) -> HipIndex:
    key = (collection, metric)
    with _cache_lock:
        ix = _cache.get(key)
        if ix is None:
            ix = HipIndex(dim, capacity, dtype=dtype, metric=metric)
            _cache[key] = ix
        elif ix.dim != dim:
            raise HipBackendError(f"collection {collection!r} holds {ix.dim}-d vectors, got {dim}-d")
        return ix


def lookup_cached(collection: str, metric: str) -> Optional[HipIndex]:
    with _cache_lock:
        return _cache.get((collection, metric))


def drop(collection: str, metric: str) -> None:
    with _cache_lock:
        ix = _cache.pop((collection, metric), None)
    if ix is not None:
        ix.close()
