"""One rank of tests/test_00_sharded_gpu.py (TEST INFRASTRUCTURE): the row-sharded search of archi_amd/sharded.py with the
REAL per-shard HIP search (ak_index_search_dev through HipLocalSearch) and the REAL merge kernel (ak_merge_shards_dev),
several ranks sharing the one GPU of the test box. RCCL refuses two ranks on one device, so the all-gather of the
Q*(2k+1)*8-byte payload is staged through the host over gloo here -- the product default (archi_amd.sharded
._rccl_all_gather) is replaced through ShardedSearcher's `gather` hook; everything else is the product path.

Each scenario is checked on every rank against the CPU oracle over the WHOLE corpus (ids and float8 bits), and on rank
0 also against one unsharded HipIndex."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from archi_amd import _lib  # noqa: E402
from archi_amd.index import HipIndex  # noqa: E402
from archi_amd.sharded import HipLocalSearch, ShardedSearcher, shard_bounds  # noqa: E402
from oracle import knn_oracle as ko  # noqa: E402


def host_staged_gather(world):
    def gather(payload):
        host = torch.empty((world * payload.numel(),), dtype=payload.dtype)
        dist.all_gather_into_tensor(host, payload.cpu())
        return host.view(world, payload.numel()).to(payload.device)
    return gather


def unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def scenarios():
    """name -> (rows f32 [N,D], ids, queries, k, dtype, metric, row mask or None, expectation on re-runs)"""
    rng = np.random.default_rng(20260)
    out = {}
    # 1. plain corpus, a zero query in the batch (cosine: every distance NaN -> never certifiable -> exact re-run)
    rows = unit(rng, 30011, 128)
    q = unit(rng, 40, 128)
    q[7] = 0.0
    out["plain_bf16_zero_query"] = (rows, np.arange(30011, dtype=np.int64) * 7 + 3, q, 10, "bf16", "cosine", None, "open>=1")
    # 2. duplicate pile-up wider than k' = 64 (second scan with k' = 512 certifies it), spread over every shard
    rows = unit(rng, 24000, 64)
    dup = np.arange(200) * 117 + 5
    rows[dup] = rows[5]
    q = np.concatenate([rows[5][None], unit(rng, 12, 64)])
    out["pileup_200"] = (rows, None, q, 10, "bf16", "cosine", None, "open>=1")
    # 3. pile-up beyond the widest lists (700 > 512): only the exact path answers
    rows = unit(rng, 20000, 64)
    dup = np.arange(700) * 27 + 1
    rows[dup] = rows[1]
    q = np.concatenate([unit(rng, 5, 64), rows[1][None]])
    out["pileup_700_f16"] = (rows, None, q, 10, "f16", "cosine", None, "open>=1")
    # 4. mostly zero rows: NaN-distance rows are needed to fill k (ordered last, id ascending)
    rows = np.zeros((13000, 64), np.float32)
    live = rng.choice(13000, 6, replace=False)
    rows[live] = unit(rng, 6, 64)
    out["nan_rows_fill_k"] = (rows, np.arange(13000, dtype=np.int64)[::-1].copy(), unit(rng, 4, 64), 10, "bf16", "cosine", None, "open>=1")
    # 5. clustered f32 corpus: rows closer to each other than the bf16 shadow scan resolves
    base = unit(rng, 1, 384)
    rows = (base + 1e-4 * rng.standard_normal((16000, 384))).astype(np.float32)
    q = (base + 1e-4 * rng.standard_normal((9, 384))).astype(np.float32)
    out["clustered_f32"] = (rows, None, q, 10, "f32", "cosine", None, None)
    # 6. shards below the MFMA scan's 4096-row floor (a 5 000-row collection over 2 or 3 GPUs)
    rows = unit(rng, 5000, 96) * 2.5
    out["small_shards_l2"] = (rows, None, unit(rng, 17, 96), 10, "f32", "l2", None, None)
    out["small_shards_ip"] = (rows, None, unit(rng, 3, 96), 25, "bf16", "inner_product", None, None)
    # 7. fewer rows than ranks x k: some shards are EMPTY, k > N
    rows = unit(rng, 2, 64)
    out["two_rows"] = (rows, np.array([41, 40], np.int64), unit(rng, 3, 64), 10, "bf16", "cosine", None, None)
    # 8. WHERE-clause mask on the device path (a7), fast path and exact path
    rows = unit(rng, 26000, 128)
    mask = (rng.random(26000) < 0.3).astype(np.uint8)
    out["row_filter_fast"] = (rows, None, unit(rng, 33, 128), 10, "bf16", "cosine", mask, None)
    rows = unit(rng, 6000, 128)
    mask = (rng.random(6000) < 0.01).astype(np.uint8)
    out["row_filter_small"] = (rows, None, unit(rng, 5, 128), 10, "f16", "cosine", mask, None)
    # 9. a batch large enough for the MFMA-bound tile (Q > 128) with one uncertifiable member
    rows = unit(rng, 40000, 256)
    q = unit(rng, 300, 256)
    q[123] = 0.0
    out["q300_l2_free"] = (rows, None, q, 10, "bf16", "cosine", None, "open>=1")
    return out


def store_scenario(rank, world):
    """The drop-in STORE over row shards (pg_config["hip"]["shards"] -> ShardedHipIndex): add, delete, re-add (ON CONFLICT),
    soft delete, the sync step's hash queries and filtered searches through ArchiHipVectorStore with the real kernels on
    every shard, against the same calls on a single-index store (every rank builds that one locally: no collectives in it)."""
    from archi_amd import vectorstore as vs
    from archi_amd.sharded import ShardedHipIndex
    D = 128
    rng = np.random.default_rng(99)
    vec = {doc: unit(rng, 900 + 37 * doc, D) for doc in range(1, 13)}         # ~13k chunks: shards above the scan's row floor
    qs = unit(rng, 6, D)

    class Emb:
        def embed_documents(self, texts):
            raise AssertionError("vectors are handed in")

        def embed_query(self, text):
            return [float(x) for x in qs[len(text) % 6]]

    def sharded_factory(dim, capacity, dtype, metric):
        return ShardedHipIndex(dim, capacity, dtype=dtype, metric=metric, shards=world, gather=host_staged_gather(world))

    def drive(store):
        log = []
        for doc, v in vec.items():
            store.add_texts([f"doc {doc} chunk {i}" for i in range(len(v))],
                            [{"source": "web" if doc % 3 else "git", "resource_hash": f"h{doc}"} for _ in range(len(v))],
                            document_id=doc, embeddings=v)
        log.append(store.count())
        store.delete(document_id=5)
        store.add_texts(["doc 7 chunk 0 v2", "doc 7 chunk 1 v2"], [{"source": "git", "resource_hash": "h7"}] * 2, document_id=7,
                        embeddings=vec[7][::-1][:2].copy())                       # ON CONFLICT (7, 0), (7, 1)
        store.table.register_document(9, is_deleted=True)
        log.append(store.count())
        log.append(sorted(store.resource_hashes()))
        store.delete_resource_hashes(["h12", "h2"])
        log.append(store.count())
        for kw in ({}, {"filter": {"source": "git"}}, {"filter": {"source": "web"}, "include_deleted": True}):
            for qtext in ("a", "bb", "ccc", "dddd"):
                res = store.similarity_search_with_score(qtext, k=10, **kw)
                log.append([(d.page_content, s) for d, s in res])
        return log

    for dtype in ("f32", "bf16"):
        vs.reset_collections()
        a = drive(vs.ArchiHipVectorStore({"hip": {"dtype": dtype, "capacity": 4096}}, Emb(), collection_name="sharded",
                                         index_factory=sharded_factory))
        b = drive(vs.ArchiHipVectorStore({"hip": {"dtype": dtype, "capacity": 4096}}, Emb(), collection_name="single"))
        if a != b:
            bad = next(i for i, (x, y) in enumerate(zip(a, b)) if x != y)
            vs.reset_collections()
            return {"ok": False, "why": f"{dtype}: step {bad}: sharded {str(a[bad])[:300]} single {str(b[bad])[:300]}"}
    vs.reset_collections()
    return {"ok": True, "open": 0}


def dp_embedding_scenario(rank, world):
    """Data-parallel embedding behind the sharded store (replicate the weights, shard the chunk batch, no
    collective) with the REAL encoder: every rank ingests the same files through BatchedIngestor, its encoder must see only
    the chunks whose rows land on its shard (1 / world of them, +- one per embed call), and the collection must equal a
    single-index store fed by an encoder that embedded everything: same row count, every stored vector equal to bf16 noise
    (the two embed the chunks in different batches, hence through different tile kernels), same top-1 for chunk texts."""
    from archi_amd import vectorstore as vs
    from archi_amd.embeddings import ArchiHipEmbeddings
    from archi_amd.ingest import BatchedIngestor
    from archi_amd.sharded import ShardedHipIndex

    class Counting(ArchiHipEmbeddings):
        seen = 0

        def embed_documents_array(self, texts):
            self.seen += len(texts)
            return super().embed_documents_array(texts)

    def sharded_factory(dim, capacity, dtype, metric):
        return ShardedHipIndex(dim, capacity, dtype=dtype, metric=metric, shards=world, gather=host_staged_gather(world))

    words = ["muon", "trigger", "calorimeter", "grid", "job", "alignment", "tracker", "release", "notes", "luminosity", "beam", "pixel"]
    rng = np.random.default_rng(4)
    files = []
    for i in range(60):
        paras = [" ".join(rng.choice(words, size=int(rng.integers(8, 40)))) + f" file {i} part {j}" for j in range(int(rng.integers(4, 30)))]
        files.append((f"hash{i}", f"file{i}.txt", "\n\n".join(paras)))
    kw = {"model_kwargs": {"synthetic_seed": 3}, "encode_kwargs": {"normalize_embeddings": True}}

    def build(factory, name):
        emb = Counting("sentence-transformers/all-MiniLM-L6-v2", **kw)
        st = vs.ArchiHipVectorStore({"hip": {"dtype": "f32", "capacity": 1024}}, emb, collection_name=name,
                                    **({"index_factory": factory} if factory else {}))
        st.add_texts(["seed row"], [{"source": "seed"}], document_id=0)      # creates the collection (the first call embeds everything)
        emb.seen = 0
        BatchedIngestor(st, collection=name, chunk_size=200, group_chunks=300).ingest(files, document_ids={f"hash{i}": i + 1 for i in range(60)})
        return st, emb

    vs.reset_collections()
    sh, emb_s = build(sharded_factory, "dp_sharded")
    one, emb_1 = build(None, "dp_single")
    n = one.count() - 1
    res = {"ok": True, "chunks": int(n), "seen_sharded": int(emb_s.seen), "seen_single": int(emb_1.seen)}
    calls = 8                                    # embed calls (groups), each split can be off by one row
    if sh.count() != one.count():
        res.update(ok=False, why=f"row counts differ: {sh.count()} vs {one.count()}")
    elif emb_1.seen != n or abs(emb_s.seen - n / world) > calls:
        res.update(ok=False, why=f"this rank's encoder saw {emb_s.seen} of {n} chunks (world {world})")
    else:
        # this rank's stored vectors against the single store's, row by row
        t = one.table
        rids = t.live_rids()
        mine = rids[(rids % world) == rank]
        a = sh._collection().index.local.fetch(sh._collection().index.local.lookup(mine))
        b = one._collection().index.fetch(one._collection().index.lookup(mine))
        cos = (a * b).sum(1)
        if len(mine) == 0 or cos.min() < 1 - 1e-4 or np.abs(a - b).max() > 2e-3:
            res.update(ok=False, why=f"stored vectors differ: min cos {float(cos.min()) if len(mine) else None}")
        else:
            probe = [t.text_at(t.pos(int(r))) for r in rids[:: max(1, len(rids) // 12)][:12]]
            for text in probe:
                x = sh.similarity_search(text, k=1)[0].page_content
                y = one.similarity_search(text, k=1)[0].page_content
                if x != y or x != text:
                    res.update(ok=False, why=f"top-1 differs for {text[:40]!r}: {x[:40]!r} / {y[:40]!r}")
                    break
    vs.reset_collections()
    return res


def failure_scenario(rank, world):
    """One rank hands its local search a row_filter built for an OLDER layout epoch (a writer ran in between): the library refuses
    it (AK_ERR_STALE_FILTER) on that rank only -- the rank still joins the all-gather with empty rows and its code in the status
    word, EVERY rank raises StaleFilterError after the exchange, and the next search on the same searcher is exact again."""
    from archi_amd import StaleFilterError
    n, d, k = 9000, 64, 5
    rows = ko.gen_rows(31, 0, 0, n, d, True, "f32")
    queries = ko.gen_rows(32, 1, 0, 6, d, True, "f32")
    lo, hi = shard_bounds(n, world, rank)
    ix = HipIndex(d, hi - lo + 16, dtype="bf16", metric="cosine")
    ix.add(rows[lo:hi], ids=np.arange(lo, hi, dtype=np.int64))
    searcher = ShardedSearcher(HipLocalSearch(ix), gather=host_staged_gather(world))
    qd = torch.from_numpy(queries).cuda()
    slots, epoch = ix.layout()
    flt = torch.ones((slots,), dtype=torch.uint8, device="cuda")
    res = {"ok": True}
    try:
        searcher.search(qd, k, row_filter=flt, filter_epoch=epoch - 1 if rank == world - 1 else epoch)
        res.update(ok=False, why="no error although one rank's filter was stale")
    except StaleFilterError:
        pass
    except Exception as exc:                              # noqa: BLE001
        res.update(ok=False, why=f"wrong error {type(exc).__name__}: {exc}")
    gi, gd = searcher.search(qd, k, row_filter=flt, filter_epoch=epoch)
    torch.cuda.synchronize()
    wi, wd, _ = ko.search(ko.round_through(rows, "bf16"), queries, k, "cosine")
    if not (np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gd.cpu().numpy(), wd)):
        res.update(ok=False, why="the search after the failed one differs from the oracle")
    ix.close()
    return res


def _same_bits(a, b):       # float8 bits, every NaN equal to every NaN
    return bool(np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.where(np.isnan(a), 0, a).view(np.int64),
                                                                              np.where(np.isnan(b), 0, b).view(np.int64)))


def abi_failure_scenarios(rank, world, abi):
    """The failure contract of ak_index_search_sharded_dev (include/archi_knn.h), through the C path at world > 1 over the
    shared-memory stand-in for RCCL (tests/native/fake_rccl.cpp): (a) a stale filter on one rank, (b) an injected local scan
    failure, (c) an injected failure of one rank's FIRST merge -- with and without open queries, i.e. with and without a second
    collective for that rank to follow the others into --, each followed by a correct search on the same communicator;
    (d) the fatal class (the exchange buffers): the rank returns, its communicator is broken, the others come back by the
    transport's time-out, and every later call on every rank fails fast with AK_ERR_COMM_BROKEN."""
    from archi_amd import StaleFilterError
    from archi_amd._lib import HipBackendError, debug_set
    from archi_amd.sharded import AbiShardedSearcher
    n, d, k = 15000, 64, 5                                  # every shard above the MFMA scan's 4096-row floor at world 2 AND 3: below it a
    rows = ko.gen_rows(41, 0, 0, n, d, True, "f32")         # shard answers exactly and certifies even the zero query -- nothing would be open
    queries = ko.gen_rows(42, 1, 0, 6, d, True, "f32")
    qz = queries.copy(); qz[2] = 0.0                       # a zero query: never certifiable -> a second collective
    lo, hi = shard_bounds(n, world, rank)
    ix = HipIndex(d, hi - lo + 16, dtype="bf16", metric="cosine")
    ix.add(rows[lo:hi], ids=np.arange(lo, hi, dtype=np.int64))
    abi.index = ix
    stored = ko.round_through(rows, "bf16")
    res = {"ok": True, "steps": []}

    def fail(why):
        res["ok"] = False
        res.setdefault("why", why)

    def good(tag, q, searcher=None):
        gi, gd = (searcher or abi).search(torch.from_numpy(q).cuda(), k)
        torch.cuda.synchronize()
        wi, wd, _ = ko.search(stored, q, k, "cosine")
        if not (np.array_equal(gi.cpu().numpy(), wi) and _same_bits(gd.cpu().numpy(), wd)):
            fail(f"{tag}: the search after the failed one differs from the oracle")
        res["steps"].append(tag + ":ok")

    def expect(tag, q, exc_type, code=None, flt=None, epoch=None, only_rank=None, searcher=None):
        try:
            (searcher or abi).search(torch.from_numpy(q).cuda(), k, row_filter=flt, filter_epoch=epoch)
            if only_rank is None or rank == only_rank:
                fail(f"{tag}: no error on rank {rank}")
        except exc_type as exc:
            if only_rank is not None and rank != only_rank:
                fail(f"{tag}: rank {rank} raised although only rank {only_rank} failed and no collective followed: {exc}")
            if code is not None and f"rc={code}" not in str(exc):
                fail(f"{tag}: wrong code in {exc}")
        except Exception as exc:                              # noqa: BLE001
            fail(f"{tag}: wrong error {type(exc).__name__}: {exc}")
        res["steps"].append(tag)

    good("warm", queries)
    if abi.last_open != 0:
        fail(f"warm: {abi.last_open} open queries in a batch that was meant to certify everywhere")
    # (a) stale filter on the last rank
    slots, epoch = ix.layout()
    flt = torch.ones((slots,), dtype=torch.uint8, device="cuda")
    expect("stale", queries, StaleFilterError, flt=flt, epoch=epoch - 1 if rank == world - 1 else epoch)
    good("after_stale", qz)
    # (b) injected local scan failure on rank 0: -10 on every rank
    if rank == 0:
        debug_set("AK_SHARD_INJECT", "1")
    expect("scan_fail", queries, HipBackendError, code=-10)
    debug_set("AK_SHARD_INJECT", None)
    good("after_scan_fail", queries)
    # (c1) first merge fails on the last rank, one query open: the others enter a second all-gather, the failing rank follows them
    if rank == world - 1:
        debug_set("AK_SHARD_INJECT", "2")
    expect("merge_fail_open", qz, HipBackendError, code=-10)
    debug_set("AK_SHARD_INJECT", None)
    good("after_merge_fail_open", qz)
    # (c2) ... nothing open: the others have their result, only the failing rank returns the error, nobody waits
    if rank == world - 1:
        debug_set("AK_SHARD_INJECT", "2")
    expect("merge_fail_closed", queries, HipBackendError, code=-10, only_rank=world - 1)
    debug_set("AK_SHARD_INJECT", None)
    good("after_merge_fail_closed", queries)
    dist.barrier()
    # (d) the fatal class, on a communicator of its own with a short transport time-out
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "4"
    abi2 = AbiShardedSearcher(ix)
    good("second_comm", queries, searcher=abi2)
    if rank == 0:
        debug_set("AK_SHARD_INJECT", "3")
    import time
    t0 = time.time()
    expect("fatal", queries, HipBackendError, code=-10 if rank == 0 else -12, searcher=abi2)
    debug_set("AK_SHARD_INJECT", None)
    if time.time() - t0 > 30:
        fail("fatal: the other ranks did not come back within the transport's time-out")
    expect("broken", queries, HipBackendError, code=-13, searcher=abi2)
    abi2.close()
    dist.barrier()
    good("old_comm_still_fine", queries)
    ix.close()
    return res


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    _lib.init(0)
    report = {}
    abi = None
    if os.environ.get("AK_RCCL_PATH"):
        # the C path (ak_comm_create + ak_index_search_sharded_dev) beside the torch path, same scenarios, same process: the
        # communicator library is tests/native/fake_rccl.cpp (several ranks on ONE GPU); one communicator for all scenarios
        from archi_amd.sharded import AbiShardedSearcher
        abi = AbiShardedSearcher(None)
    for name, (rows, ids, queries, k, dtype, metric, mask, expect) in scenarios().items():
        n, d = rows.shape
        ids = np.arange(n, dtype=np.int64) if ids is None else ids
        lo, hi = shard_bounds(n, world, rank)
        ix = HipIndex(d, max(hi - lo, 1), dtype=dtype, metric=metric)
        if hi > lo:
            ix.add(rows[lo:hi], ids=ids[lo:hi])
        searcher = ShardedSearcher(HipLocalSearch(ix), gather=host_staged_gather(world))
        qd = torch.from_numpy(queries).cuda()
        flt = None if mask is None else torch.from_numpy(np.ascontiguousarray(mask[lo:hi])).cuda()
        gi, gd = searcher.search(qd, k, row_filter=flt)
        torch.cuda.synchronize()
        gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
        stored = ko.round_through(rows, dtype)
        wi, wd, _ = ko.search(stored, queries, k, metric, ids=ids, alive=mask)
        same = _same_bits
        ok = bool(np.array_equal(gi, wi) and same(gd, wd))
        res = {"ok": ok, "open": searcher.last_open}
        if abi is not None:
            abi.index = ix
            ai, ad = abi.search(qd, k, row_filter=flt)
            torch.cuda.synchronize()
            ai, ad = ai.cpu().numpy(), ad.cpu().numpy()
            res["abi_equal"] = bool(np.array_equal(ai, gi) and same(ad, gd) and np.array_equal(ai, wi))
            res["abi_open"] = abi.last_open
            if not res["abi_equal"] or abi.last_open != searcher.last_open:
                res["ok"] = False
                res["why"] = f"C path differs from the torch path / oracle (open {abi.last_open} vs {searcher.last_open})"
        if expect == "open>=1" and searcher.last_open < 1:
            res["ok"] = False
            res["why"] = "expected at least one query to need the exact re-run"
        if not ok:
            bad = np.argwhere((gi != wi) | ~((gd == wd) | (np.isnan(gd) & np.isnan(wd))))
            res["why"] = f"differs from the oracle at {bad[:4].tolist()}: got {gi[bad[0][0]].tolist()} want {wi[bad[0][0]].tolist()}"
        if rank == 0:
            full = HipIndex(d, n, dtype=dtype, metric=metric)
            full.add(rows, ids=ids)
            fi, fd, _ = full.search(queries, k, mode="auto", row_filter=mask)
            res["single_index_equal"] = bool(np.array_equal(fi, gi) and same(fd, gd))
            res["ok"] = res["ok"] and res["single_index_equal"]
            full.close()
        ix.close()
        report[name] = res
        dist.barrier()
    report["failure_agreement"] = failure_scenario(rank, world)
    dist.barrier()
    if abi is not None:
        report["abi_failures"] = abi_failure_scenarios(rank, world, abi)
        abi.close()
        dist.barrier()
    report["store_api"] = store_scenario(rank, world)
    dist.barrier()
    report["dp_embedding"] = dp_embedding_scenario(rank, world)
    dist.barrier()
    json.dump(report, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
