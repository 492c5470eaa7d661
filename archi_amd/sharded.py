"""Row-sharded retrieval across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI).
The corpus is partitioned row-wise into contiguous blocks; every rank scans its
shard for the whole query batch, the per-shard partial top-k ([Q,k] ids +
float8 distances, Q*k*16 bytes per rank -- latency bound) is exchanged with ONE
all-gather, and every rank merges the G partial lists with the same
(distance asc, NaN last, id asc) comparator, so the result is identical for any
shard count. No other collective touches the data path.

The reference has no distributed code at all (its scan runs inside one Postgres
backend: src/data_manager/vectorstore/postgres_vectorstore.py:317-332); this
module is new work specified by the north star, not a restatement.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one row."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# local_search(queries[Q,D] f32 on device, k) -> (ids [Q,k] int64, dist [Q,k] float64) on the same device
LocalSearch = Callable[[torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]]
# merge(part_ids [G,Q,k], part_dist [G,Q,k]) -> (ids [Q,k], dist [Q,k])
Merge = Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]


def hip_merge(part_ids: torch.Tensor, part_dist: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge kernel of libarchi_hip.so (ak_merge_topk_dev) on the current stream."""
    from .index import merge_topk_device
    g, q, k = part_ids.shape
    out_i = torch.empty((q, k), dtype=torch.int64, device=part_ids.device)
    out_d = torch.empty((q, k), dtype=torch.float64, device=part_ids.device)
    merge_topk_device(g, q, k, part_ids.data_ptr(), part_dist.data_ptr(), out_i.data_ptr(), out_d.data_ptr(),
                      torch.cuda.current_stream(part_ids.device).cuda_stream)
    return out_i, out_d


class HipLocalSearch:
    """local_search over a HipIndex through the device-resident C-ABI entry point."""

    def __init__(self, index) -> None:
        self.index = index
        self._bufs = {}

    def __call__(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        nq = queries.shape[0]
        key = (nq, k)
        if key not in self._bufs:
            dev = queries.device
            self._bufs[key] = (torch.empty((nq, k), dtype=torch.int64, device=dev),
                               torch.empty((nq, k), dtype=torch.float64, device=dev),
                               torch.empty((nq,), dtype=torch.int32, device=dev))
        oi, od, oc = self._bufs[key]
        self.index.search_device(queries.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(),
                                 torch.cuda.current_stream(queries.device).cuda_stream)
        self.last_cert = oc
        return oi, od


class ShardedSearcher:
    """Scan the local shard, all-gather the partial top-k, merge."""

    def __init__(self, local_search: LocalSearch, merge: Optional[Merge] = None,
                 group: Optional[dist.ProcessGroup] = None) -> None:
        self.local_search = local_search
        self.merge = merge or hip_merge
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def search(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        ids, dd = self.local_search(queries, k)
        if self.world == 1:
            return ids, dd
        q = ids.shape[0]
        # one collective: ids and distances travel as one int64 payload [2,Q,k] per rank
        payload = torch.stack([ids, dd.view(torch.int64)], dim=0).contiguous()
        # concatenation form ([G*2,Q,k]) is the one both RCCL and gloo accept
        if payload.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (bench.py --backend gloo: several ranks sharing one GPU, where RCCL refuses duplicate
            # devices): gloo has no CUDA all-gather, so the 2*Q*k*8-byte payload is staged through the host
            host = torch.empty((self.world * 2, q, k), dtype=torch.int64)
            dist.all_gather_into_tensor(host, payload.cpu(), group=self.group)
            flat = host.to(payload.device)
        else:
            flat = torch.empty((self.world * 2, q, k), dtype=torch.int64, device=payload.device)
            dist.all_gather_into_tensor(flat, payload, group=self.group)
        gathered = flat.view(self.world, 2, q, k)
        part_ids = gathered[:, 0].contiguous()
        part_dist = gathered[:, 1].contiguous().view(torch.float64)
        assert part_ids.shape == (self.world, q, k)
        return self.merge(part_ids, part_dist)
