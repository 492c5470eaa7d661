"""Oracle-backed stand-in for archi_amd.index.HipIndex (TESTS ONLY): lets the CPU suite
exercise the host logic of the vector store without a GPU. Same method surface."""
import numpy as np

from oracle import knn_oracle as ko


class OracleIndex:
    def __init__(self, dim, capacity, dtype="bf16", metric="cosine"):
        self.dim, self.capacity, self.dtype, self.metric = dim, capacity, dtype, metric
        self.rows = np.zeros((0, dim), np.float32)
        self.ids = np.zeros((0,), np.int64)
        self.alive = np.zeros((0,), np.uint8)
        self.slots = 0
        self.epoch = 1          # layout epoch, as HipIndex: every add (and every reclaim, which this stand-in never does) moves it

    def add(self, rows, ids=None, normalise=False):
        rows = np.ascontiguousarray(rows, np.float32)
        if normalise:
            rows = ko.l2_normalize(rows)
        rows = ko.round_through(rows, self.dtype)
        n = len(rows)
        if ids is None:
            base = int(self.ids.max()) + 1 if len(self.ids) else 0
            ids = np.arange(base, base + n)
        self.rows = np.concatenate([self.rows, rows])
        self.ids = np.concatenate([self.ids, np.asarray(ids, np.int64)])
        self.alive = np.concatenate([self.alive, np.ones(n, np.uint8)])
        self.slots += n
        self.epoch += 1

    def layout(self):
        return self.slots, self.epoch

    def remove(self, ids):
        m = np.isin(self.ids, np.asarray(ids, np.int64)) & (self.alive == 1)
        self.alive[m] = 0
        return int(m.sum())

    def count(self):
        return int(self.alive.sum())

    def lookup(self, ids):
        out = np.full(len(ids), -1, np.int64)
        for j, i in enumerate(ids):
            w = np.nonzero((self.ids == i) & (self.alive == 1))[0]
            if len(w):
                out[j] = w[-1]
        return out

    def search(self, queries, k, mode="auto", row_filter=None, return_stats=False, filter_epoch=None):
        if row_filter is not None and ((filter_epoch is not None and filter_epoch != self.epoch) or len(row_filter) != self.slots):
            from archi_amd import StaleFilterError
            raise StaleFilterError("stale row_filter")
        alive = self.alive if row_filter is None else (self.alive & np.asarray(row_filter, np.uint8))
        q = np.asarray(queries, np.float32)
        q = q[None] if q.ndim == 1 else q
        return ko.search(self.rows, q, k, self.metric, ids=self.ids, alive=alive)

    def distances(self, query, ids):
        slots = self.lookup(list(ids))
        q = np.asarray(query, np.float32).reshape(-1)
        out = np.full(len(slots), np.nan)
        for j, sl in enumerate(slots):
            if sl >= 0:
                out[j] = ko.distance(self.metric, self.rows[sl], q)
        return out, slots >= 0

    def fetch(self, slots):
        return self.rows[np.asarray(slots, np.int64)].copy()

    def close(self):
        pass
