"""CPU suite: the N>1 path (row sharding + all-gather + merge + re-run of uncertified queries) with world_size 2 and 3
over gloo. The per-shard scan and the merge are stood in by the oracle (the checker); what is under test is the host
logic of archi_amd/sharded.py: shard bounds, id offsets, payload packing, gather layout, and the certificate protocol
(a query ANY shard leaves uncertified is re-run on every shard and merged again). The same protocol with the real HIP
kernels runs in tests/test_00_sharded_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from archi_amd.sharded import ShardedSearcher, shard_bounds
from oracle import knn_oracle as ko

N, D, NQ, K = 1237, 48, 9, 10


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 1000, 10_000_001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    corpus = ko.gen_rows(7, 0, 0, N, D, True, "bf16")
    corpus[N // 2 + 3] = corpus[5]                      # a cross-shard exact tie
    ids = (np.arange(N, dtype=np.int64) * 3 + 11)
    queries = np.concatenate([corpus[5][None], ko.gen_rows(8, 1, 0, NQ - 1, D, True, "f32")])
    lo, hi = shard_bounds(N, world, rank)

    calls = []

    def local_search(q, k, mode="fast_only", row_filter=None):
        calls.append((mode, q.shape[0]))
        i, d, _ = ko.search(corpus[lo:hi], q.numpy(), k, "cosine", ids=ids[lo:hi])
        cert = np.ones(q.shape[0], np.int32)
        if mode == "fast_only":
            # a shard that cannot prove some of its answers returns GARBAGE rows for them and says so: query 2 on rank 0,
            # query 4 on the last rank (every rank must still end up with the exact answer)
            for qi, r in ((2, 0), (4, world - 1)):
                if rank == r and qi < q.shape[0]:
                    cert[qi] = 0
                    i[qi] = -1
                    d[qi] = np.nan
        return torch.from_numpy(i), torch.from_numpy(d), torch.from_numpy(cert)

    def merge(gathered, q, k):
        g = gathered.numpy()
        pi = g[:, :q * k].reshape(world, q, k)
        pd = g[:, q * k:2 * q * k].copy().view(np.float64).reshape(world, q, k)
        cert = np.ascontiguousarray(g[:, 2 * q * k:]).view(np.int32)[:, :q]
        i, d = ko.merge(np.ascontiguousarray(pi), np.ascontiguousarray(pd))
        open_q = (cert == 0).any(axis=0).astype(np.int32)
        return torch.from_numpy(i), torch.from_numpy(d), torch.from_numpy(np.concatenate([open_q, [open_q.sum()]]).astype(np.int32))

    s = ShardedSearcher(local_search, merge=merge)
    gi, gd = s.search(torch.from_numpy(queries), K)
    wi, wd, _ = ko.search(corpus, queries, K, "cosine", ids=ids)
    ok = np.array_equal(gi.numpy(), wi) and np.array_equal(gd.numpy(), wd)
    ok = ok and s.last_open == 2 and calls == [("fast_only", NQ), ("auto", 2)]
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "MISMATCH")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_gloo(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
