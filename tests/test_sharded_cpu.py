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

    def local_search(q, k, mode="fast_only", row_filter=None, filter_epoch=None):
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


# ---- a rank-local failure is raised on EVERY rank, after the exchange (round-4 review: the others used to wait for ever) ----------
def _failure_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from archi_amd import StaleFilterError
    from archi_amd.sharded import ShardSearchError
    corpus = ko.gen_rows(7, 0, 0, N, D, True, "bf16")
    ids = np.arange(N, dtype=np.int64)
    queries = ko.gen_rows(8, 1, 0, NQ, D, True, "f32")
    lo, hi = shard_bounds(N, world, rank)
    fail = {"what": None, "rerun_open": False}

    def local_search(q, k, mode="fast_only", row_filter=None, filter_epoch=None):
        if rank == 1 and fail["what"] == "stale" and mode == "fast_only":
            raise StaleFilterError("row_filter built for layout epoch 3, the index is at 4")
        if rank == world - 1 and fail["what"] == "boom" and mode == "fast_only":
            raise RuntimeError("workspace could not grow")
        if rank == 0 and fail["what"] == "rerun" and mode == "auto":
            raise StaleFilterError("a writer moved the layout between the scan and the re-run")
        i, d, _ = ko.search(corpus[lo:hi], q.numpy(), k, "cosine", ids=ids[lo:hi])
        cert = np.ones(q.shape[0], np.int32)
        if mode == "fast_only" and fail["what"] == "rerun" and rank == 1:
            cert[3] = 0                                   # some shard leaves a query open: everybody re-runs it
        return torch.from_numpy(i), torch.from_numpy(d), torch.from_numpy(cert)

    def merge(gathered, q, k):
        g = gathered.numpy()
        pi = g[:, :q * k].reshape(world, q, k)
        pd = g[:, q * k:2 * q * k].copy().view(np.float64).reshape(world, q, k)
        cert = np.ascontiguousarray(g[:, 2 * q * k:-1]).view(np.int32)[:, :q]
        i, d = ko.merge(np.ascontiguousarray(pi), np.ascontiguousarray(pd))
        open_q = (cert == 0).any(axis=0).astype(np.int32)
        return torch.from_numpy(i), torch.from_numpy(d), torch.from_numpy(np.concatenate([open_q, [open_q.sum()]]).astype(np.int32))

    s = ShardedSearcher(local_search, merge=merge)
    wi, wd, _ = ko.search(corpus, queries, K, "cosine", ids=ids)
    log = []
    for what, exc_type in (("stale", StaleFilterError), ("boom", (ShardSearchError, RuntimeError)), ("rerun", StaleFilterError), (None, None)):
        fail["what"] = what
        try:
            gi, gd = s.search(torch.from_numpy(queries), K)
            log.append("ok" if what is None and np.array_equal(gi.numpy(), wi) and np.array_equal(gd.numpy(), wd) else f"no error for {what}")
        except Exception as exc:                          # noqa: BLE001
            log.append("raised" if exc_type is not None and isinstance(exc, exc_type) else f"wrong error {type(exc).__name__}: {exc}")
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(",".join(log))
    dist.destroy_process_group()


def test_a_failed_local_search_is_raised_on_every_rank_and_the_next_search_works(tmp_path):
    """One rank's local search fails (stale filter on the scan, an arbitrary error, a stale filter on the re-run of an open
    query): the failing rank still joins the all-gather with empty rows and its code in the payload's status word, every rank
    raises after the exchange (the same error class everywhere), and the next search on the same group succeeds."""
    world = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_failure_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "raised,raised,raised,ok", r


# ---- the store API over row shards: ShardedHipIndex behind ArchiHipVectorStore, world 2 over gloo ----------
def _store_worker(rank, world, port, out_dir):
    import json
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from archi_amd import vectorstore as vs
    from archi_amd.sharded import ShardedHipIndex
    from tests.fake_index import OracleIndex
    D2 = 32

    class Emb:
        def embed_documents(self, texts):
            return [[float(x) for x in r] for r in ko.gen_rows(77, 5, 0, len(texts), D2, True, "f32")]

        def embed_query(self, text):
            return [float(x) for x in ko.gen_rows(77, 6, len(text), 1, D2, True, "f32")[0]]

    def sharded_factory(dim, capacity, dtype, metric):
        local = OracleIndex(dim, capacity, dtype=dtype, metric=metric)

        def local_search(q, k, mode="fast_only", row_filter=None, filter_epoch=None):
            flt = None if row_filter is None else row_filter.numpy()
            i, d, _ = local.search(q.numpy(), k, row_filter=flt)
            return torch.from_numpy(i), torch.from_numpy(d), torch.ones(q.shape[0], dtype=torch.int32)

        def merge(gathered, q, k):
            g = gathered.numpy()
            pi = g[:, :q * k].reshape(world, q, k)
            pd = g[:, q * k:2 * q * k].copy().view(np.float64).reshape(world, q, k)
            i, d = ko.merge(np.ascontiguousarray(pi), np.ascontiguousarray(pd))
            return torch.from_numpy(i), torch.from_numpy(d), torch.zeros(q + 1, dtype=torch.int32)

        return ShardedHipIndex(dim, capacity, dtype=dtype, metric=metric, shards=world, local_index=local,
                               local_search=local_search, merge=merge)

    def single_factory(dim, capacity, dtype, metric):
        return OracleIndex(dim, capacity, dtype=dtype, metric=metric)

    def drive(store):
        """add, delete, re-add, soft delete, filtered search: what the data manager and the chat service do"""
        log = []
        for doc in range(1, 9):
            texts = [f"doc {doc} chunk {i}" for i in range(5 + doc)]
            vecs = ko.gen_rows(500 + doc, 5, 0, len(texts), D2, True, "f32")
            store.add_texts(texts, [{"source": "web" if doc % 2 else "git", "resource_hash": f"h{doc}"} for _ in texts],
                            document_id=doc, embeddings=vecs)
        log.append(store.count())
        store.delete(document_id=3)
        store.add_texts(["doc 4 chunk 0 v2"], [{"source": "git", "resource_hash": "h4"}], document_id=4,
                        embeddings=ko.gen_rows(900, 5, 0, 1, D2, True, "f32"))          # ON CONFLICT (4, 0)
        store.table.register_document(6, is_deleted=True, display_name="Doc six")
        log.append(store.count())
        log.append(sorted(store.resource_hashes()))
        store.delete_resource_hashes(["h8"])
        log.append(store.count())
        for kw in ({}, {"filter": {"source": "web"}}, {"filter": {"source": "git"}, "include_deleted": True}):
            for qtext in ("alpha", "beta beta", "g"):
                res = store.similarity_search_with_score(qtext, k=7, **kw)
                log.append([(d.page_content, s) for d, s in res])
        return log

    a = drive(vs.ArchiHipVectorStore({"hip": {"dtype": "f32"}}, Emb(), collection_name="sharded", index_factory=sharded_factory))
    b = drive(vs.ArchiHipVectorStore({"hip": {"dtype": "f32"}}, Emb(), collection_name="single", index_factory=single_factory))
    # the hybrid combine over shards: exact distances of the BM25 hits come from whichever shard holds the row (all-reduce)
    def hybrid(factory, name):
        st = vs.ArchiHipHybridVectorStore({"hip": {"dtype": "f32"}}, Emb(), collection_name=name, index_factory=factory,
                                          bm25=vs.HostBm25())
        st.add_texts([f"muon chamber {i} alignment" if i % 3 == 0 else f"grid job {i}" for i in range(40)],
                     [{"page": i % 2} for i in range(40)], document_id=1,
                     embeddings=np.concatenate([ko.gen_rows(321, 5, 0, 39, D2, True, "f32"), np.zeros((1, D2), np.float32)]))
        out = []
        for kw in ({}, {"filter": {"page": 1}}):
            res = st.hybrid_search("muon alignment", k=6, semantic_weight=0.6, bm25_weight=0.4, **kw)
            out.append([(d.page_content, None if s != s else s) for d, s in res])
        return out
    a.append(hybrid(sharded_factory, "hsharded"))
    b.append(hybrid(single_factory, "hsingle"))
    ok = a == b and len(a) == 14 and a[0] == sum(5 + d for d in range(1, 9)) and len(a[-1][0]) == 6
    # ---- data-parallel embedding behind the sharded store (VERDICT r3 #5 / SURVEY 8e): every rank runs the same ingestion, each
    # embeds only the chunks whose rows land on its shard -- 1 / world of them, give or take one per call --, the result equals
    # the single-index store, and a file whose embedding fails on ONE rank is marked failed on ALL of them
    from archi_amd.ingest import BatchedIngestor

    class CountingEmb:
        """A deterministic embedder: the vector is a function of the text alone (so any rank computes the same one)."""
        def __init__(self):
            self.seen = 0

        def _vec(self, text):
            import zlib
            return ko.gen_rows(zlib.crc32(text.encode()) % 100000, 5, 0, 1, D2, True, "f32")[0]

        def embed_documents(self, texts):
            if any("POISON" in t for t in texts):
                raise RuntimeError("bad chunk")
            self.seen += len(texts)
            return [[float(x) for x in self._vec(t)] for t in texts]

        def embed_query(self, text):
            return [float(x) for x in self._vec(text)]

    files = [(f"hash{i}", f"file{i}.txt", "\n\n".join(f"file {i} paragraph {j} " + "x" * 30 for j in range(3 + i % 5))) for i in range(40)]
    files[17] = ("hash17", "file17.txt", "fine paragraph " + "y" * 30 + "\n\nPOISON paragraph " + "z" * 30)      # one bad chunk
    def ingest(factory, name):
        emb = CountingEmb()
        st = vs.ArchiHipVectorStore({"hip": {"dtype": "f32"}}, emb, collection_name=name, index_factory=factory)
        st.add_texts(["seed row"], [{"source": "seed"}], document_id=0)          # creates the collection (the first call embeds everything)
        first = emb.seen
        status = {}
        ing = BatchedIngestor(st, collection=name, chunk_size=60, on_status=lambda h, s_, e: status.__setitem__(h, s_), group_chunks=25)
        done = ing.ingest(files, document_ids={f"hash{i}": i + 1 for i in range(40)})
        st.add_texts([f"late {i}" for i in range(7)], [{"source": "late"} for _ in range(7)], document_id=99)
        res = [[(d.page_content, sc) for d, sc in st.similarity_search_with_score(f"file {i} paragraph 1 " + "x" * 30, k=5)] for i in (3, 8, 21)]
        return emb.seen - first, st.count(), sorted(h for h, v in status.items() if v == "failed"), sorted(done), res
    seen_s, count_s, failed_s, done_s, res_s = ingest(sharded_factory, "dp_sharded")
    seen_1, count_1, failed_1, done_1, res_1 = ingest(single_factory, "dp_single")
    total = count_1 - 1
    calls = 40                        # embed calls that can each be off by one row in the split
    dp_ok = (count_s == count_1 and failed_s == failed_1 == ["hash17"] and done_s == done_1 and res_s == res_1
             and seen_1 >= total and abs(seen_s - total / world) <= calls and seen_s < 0.75 * seen_1)
    ok = ok and dp_ok
    # ---- refresh_from_pgcopy behind a row-sharded collection (round 6): every rank runs the same refresh with the same streams
    # (SPMD) and decodes only the vectors of its own rows; deletes, in-place rewrites, new rows and the documents mirror must leave
    # the sharded reader answering exactly like a single-index store loaded from scratch from the writer's final dump
    from tests.refresh_scenario import Table, answers, ingest as rs_ingest, unit as rs_unit, writer_moves
    rng = np.random.default_rng(12345)                     # the same seed on every rank: the same writer on every rank
    d3 = 32

    class NoEmb:
        def embed_documents(self, texts):
            raise AssertionError("vectors are handed in")

        def embed_query(self, text):
            raise AssertionError("by vector")
    w = vs.ArchiHipVectorStore({"hip": {"dtype": "f32"}}, NoEmb(), collection_name="shared2", distance_metric="l2", index_factory=single_factory)
    for doc in range(1, 31):
        rs_ingest(w, rng, doc, 12 + doc % 5, d3)
    table = Table(w)
    table.commit()
    r = vs.ArchiHipVectorStore({"hip": {"dtype": "f32", "shards": world}}, NoEmb(), collection_name="shared2", index_factory=sharded_factory)
    r.load_from_pgcopy(table.rows_stream(), table.documents_stream(), versions_stream=table.ids_stream())
    writer_moves(w, table, rng, d3, 31)
    queries = rs_unit(rng, 4, d3)
    stats = r.refresh_from_pgcopy(table.ids_stream(), lambda ids: table.rows_stream(np.asarray(ids).tolist()), table.documents_stream())
    got = answers(r, queries, hybrid=False)
    again = r.refresh_from_pgcopy(table.ids_stream(), None, table.documents_stream())
    final = (table.rows_stream(), table.documents_stream())
    vs._collections.pop(("shared2", "cosine"))             # the sharded reader aside: the reference takes its (name, metric) key
    ref = vs.ArchiHipVectorStore({"hip": {"dtype": "f32"}}, NoEmb(), collection_name="shared2", index_factory=single_factory)
    ref.load_from_pgcopy(*final)
    want = answers(ref, queries, hybrid=False)
    rf_ok = (got == want and stats["updated"] == 1 and stats["added"] == 5 * 40 + 35 and stats["removed"] > 0
             and again == {"removed": 0, "added": 0, "updated": 0, "documents_changed": 0, "fetched": 0})
    ok = ok and rf_ok
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "MISMATCH " + json.dumps([dp_ok, rf_ok, stats, again, seen_s, seen_1, count_s, count_1, failed_s, a, b], default=str)[:2000])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_store_api_over_row_shards_equals_single_index_store(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_store_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


# ---- tests/native/fake_rccl.cpp (TEST INFRASTRUCTURE): the shared-memory stand-in for RCCL that the GPU suite names to the
# library through AK_RCCL_PATH so that ak_index_search_sharded_dev runs at world 2 / 3 on a one-GPU box. Its segment / barrier /
# time-out logic is checked here in the host-only build (plain memcpy instead of HIP copies), several processes, no GPU.
def _fake_rccl_rank(path, uid, rank, world, absent, q):
    import ctypes

    class Uid(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    try:
        lib = ctypes.CDLL(path)
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
        lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        u = Uid()
        ctypes.memmove(ctypes.byref(u), uid, 128)
        comm = ctypes.c_void_p()
        rc = lib.ncclCommInitRank(ctypes.byref(comm), world, u, rank)
        if rc != 0:
            q.put((rank, "init", rc)); return
        out = []
        for rnd, n in enumerate((5, 40000, 1)):                     # int64 words per rank; sizes differ between collectives
            send = (np.arange(n, dtype=np.int64) + 1000003 * rank + 17 * rnd)
            recv = np.full(world * n, -7, np.int64)
            rc = lib.ncclAllGather(send.ctypes.data, recv.ctypes.data, n, 4, comm, None)
            want = np.concatenate([np.arange(n, dtype=np.int64) + 1000003 * r + 17 * rnd for r in range(world)])
            out.append((rc, bool(np.array_equal(recv, want))))
        rc_absent = None
        if absent is not None:                                      # one rank never enters the next collective: the others must
            if rank != absent:                                      # come back with an error inside the time-out, not hang
                send = np.zeros(8, np.int64); recv = np.zeros(8 * world, np.int64)
                rc_absent = lib.ncclAllGather(send.ctypes.data, recv.ctypes.data, 8, 4, comm, None)
        lib.ncclCommDestroy(comm)
        q.put((rank, out, rc_absent))
    except Exception as exc:                                        # noqa: BLE001
        q.put((rank, "exception", repr(exc)))


@pytest.mark.parametrize("world,absent", [(2, None), (3, 1)])
def test_fake_rccl_segment_barrier_and_timeout(world, absent):
    import ctypes
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "archi_amd", "csrc"), "fake-rccl-host"])
    path = os.path.join(root, "archi_amd", "csrc", "build", "libfake_rccl_host.so")
    lib = ctypes.CDLL(path)
    uid = ctypes.create_string_buffer(128)
    assert lib.ncclGetUniqueId(uid) == 0
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "2"
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        t0 = time.time()
        procs = [ctx.Process(target=_fake_rccl_rank, args=(path, uid.raw, r, world, absent, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=60) for _ in range(world)]
        for p in procs:
            p.join(timeout=30)
        assert time.time() - t0 < 50
    finally:
        del os.environ["FAKE_RCCL_TIMEOUT_S"]
    for rank, out, rc_absent in got:
        assert isinstance(out, list), (rank, out, rc_absent)
        assert all(rc == 0 and ok for rc, ok in out), (rank, out)
        if absent is not None and rank != absent:
            assert rc_absent == 2, (rank, rc_absent)                # ncclSystemError: the time-out, reported on every waiting rank
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("akfake_")]     # rank 0 unlinks the name once everybody is attached
