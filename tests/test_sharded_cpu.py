"""CPU suite: the N>1 path (row sharding + all-gather + merge) with world_size 2 over gloo.
The per-shard scan and the merge are stood in by the oracle (the checker); what is under test is
the host logic of archi_amd/sharded.py: shard bounds, id offsets, payload packing, gather layout."""
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

    def local_search(q, k):
        i, d, _ = ko.search(corpus[lo:hi], q.numpy(), k, "cosine", ids=ids[lo:hi])
        return torch.from_numpy(i), torch.from_numpy(d)

    def merge(pi, pd):
        i, d = ko.merge(pi.numpy(), pd.numpy())
        return torch.from_numpy(i), torch.from_numpy(d)

    s = ShardedSearcher(local_search, merge=merge)
    gi, gd = s.search(torch.from_numpy(queries), K)
    wi, wd, _ = ko.search(corpus, queries, K, "cosine", ids=ids)
    ok = np.array_equal(gi.numpy(), wi) and np.array_equal(gd.numpy(), wd)
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
