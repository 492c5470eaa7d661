"""GPU suite: the index as a long-running data manager uses it (SURVEY a10 / manager.py:192-211: update_vectorstore deletes
and re-adds changed files for ever) and the device-resident entry point's modes. Oracle = CPU restatement; ids and float8
distances compared with array_equal."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_index_grows_and_reclaims_tombstones(hip, dtype):
    """capacity is only the first reservation: adds beyond it grow the buffers, and a delete / re-add cycle that keeps the
    live row count constant never exhausts them (tombstones are reclaimed). Results stay the oracle's throughout."""
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(5)
    d = 128
    ix = HipIndex(d, 1000, dtype=dtype, metric="cosine", device=0)
    live = {}                                  # id -> stored row
    nxt = 0
    q = _unit(rng, 7, d)

    def add(n):
        nonlocal nxt
        rows = _unit(rng, n, d)
        ids = np.arange(nxt, nxt + n, dtype=np.int64)
        nxt += n
        ix.add(rows, ids=ids)
        for i, r in zip(ids, ko.round_through(rows, dtype)):
            live[int(i)] = r

    def check(mode):
        ids = np.array(sorted(live), dtype=np.int64)
        stored = np.stack([live[int(i)] for i in ids])
        gi, gd, gc = ix.search(q, 10, mode=mode)
        wi, wd, wc = ko.search(stored, q, 10, "cosine", ids=ids)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd) and np.array_equal(gc, wc)
        assert ix.count() == len(live)

    add(900)
    add(5000)                                  # beyond the 1000-row reservation: grows
    assert ix.allocated_rows >= 5900
    check("auto")
    cap_after_growth = None
    for cycle in range(12):                    # re-ingest 2000 rows per cycle: 24 000 slots' worth of appends
        victims = rng.choice(sorted(live), 2000, replace=False)
        assert ix.remove(victims) == 2000
        for v in victims:
            del live[int(v)]
        add(2000)
        if cycle == 3:
            cap_after_growth = ix.allocated_rows
    assert ix.allocated_rows == cap_after_growth, "a steady-state delete/re-add cycle must not keep growing the buffers"
    assert ix.slots <= ix.allocated_rows
    check("auto")
    check("exact")
    # row_filter follows the current slot numbering
    slots = ix.lookup(sorted(live))
    assert (slots >= 0).all() and len(set(slots.tolist())) == len(live)
    mask = np.zeros(ix.slots, np.uint8)
    keep = sorted(live)[::3]
    mask[ix.lookup(keep)] = 1
    ids = np.array(keep, dtype=np.int64)
    gi, gd, _ = ix.search(q, 10, row_filter=mask)
    wi, wd, _ = ko.search(np.stack([live[i] for i in keep]), q, 10, "cosine", ids=ids)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    # explicit VACUUM
    ix.remove(sorted(live)[:100])
    for i in sorted(live)[:100]:
        del live[i]
    before = ix.slots
    assert ix.compact() == before - len(live) and ix.slots == len(live)
    check("auto")
    # ids == None continues above the largest id ever stored, also after deletes and compaction
    ix.add(_unit(rng, 1, d))
    assert ix.lookup([nxt])[0] >= 0
    ix.close()


def test_duplicate_ids_are_refused(hip):
    from archi_amd._lib import HipBackendError
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(1)
    ix = HipIndex(64, 100, dtype="bf16", metric="cosine", device=0)
    ix.add(_unit(rng, 3, 64), ids=[5, 6, 7])
    with pytest.raises(HipBackendError, match="duplicate id"):
        ix.add(_unit(rng, 2, 64), ids=[9, 9])               # inside one batch
    with pytest.raises(HipBackendError, match="duplicate id"):
        ix.add(_unit(rng, 1, 64), ids=[6])                  # against a live row
    assert ix.count() == 3
    ix.remove([6])
    ix.add(_unit(rng, 1, 64), ids=[6])                      # a deleted id may come back
    assert ix.count() == 3 and ix.lookup([6])[0] >= 0
    ix.close()


def _dev_search(ix, q, k, mode, mask=None):
    tq = torch.from_numpy(q).cuda()
    nq = len(q)
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    oc = torch.zeros((nq,), dtype=torch.int32, device="cuda")
    flt = None if mask is None else torch.from_numpy(mask).cuda()
    ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(),
                     torch.cuda.current_stream().cuda_stream, mode=mode, row_filter_ptr=0 if flt is None else flt.data_ptr())
    torch.cuda.synchronize()
    return oi.cpu().numpy(), od.cpu().numpy(), oc.cpu().numpy()


def test_device_search_auto_reruns_what_the_scan_cannot_certify(hip):
    """ak_index_search_dev, AUTO: duplicate pile-ups (200: second scan; 700: exact path), a zero query, NaN rows -- every
    row the oracle's, every flag 1. FAST_ONLY on the same input reports the open queries instead of answering them."""
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(11)
    n, d, k = 30000, 64, 10
    rows = _unit(rng, n, d)
    rows[np.arange(200) * 97 + 3] = rows[3]
    rows[np.arange(700) * 31 + 11] = rows[11]
    rows[20000:20050] = 0.0
    q = np.concatenate([rows[3][None], rows[11][None], np.zeros((1, d), np.float32), _unit(rng, 30, d)])
    ix = HipIndex(d, n, dtype="bf16", metric="cosine", device=0)
    ix.add(rows)
    stored = ko.round_through(rows, "bf16")
    wi, wd, _ = ko.search(stored, q, k, "cosine")
    gi, gd, gc = _dev_search(ix, q, k, "fast_only")
    assert gc[0] == 0 and gc[1] == 0 and gc[2] == 0 and gc[3:].all()
    ok = gc.astype(bool)
    assert np.array_equal(gi[ok], wi[ok]) and np.array_equal(gd[ok], wd[ok])
    gi, gd, gc = _dev_search(ix, q, k, "auto")
    assert gc.all() and np.array_equal(gi, wi) and np.array_equal(gd, wd, equal_nan=True)
    gi, gd, gc = _dev_search(ix, q, k, "exact")
    assert gc.all() and np.array_equal(gi, wi) and np.array_equal(gd, wd, equal_nan=True)
    mask = (rng.random(n) < 0.2).astype(np.uint8)
    wi, wd, _ = ko.search(stored, q, k, "cosine", alive=mask)
    gi, gd, gc = _dev_search(ix, q, k, "auto", mask)
    assert gc.all() and np.array_equal(gi, wi) and np.array_equal(gd, wd, equal_nan=True)
    ix.close()


@pytest.mark.parametrize("n", [0, 1, 3000])
def test_device_search_takes_indexes_below_the_scan_floor(hip, n):
    """fewer than 4096 rows (or none): the device entry point runs the exact path in every mode instead of failing."""
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(n)
    d, k = 96, 10
    rows = _unit(rng, n, d)
    q = _unit(rng, 5, d)
    ix = HipIndex(d, max(n, 1), dtype="f32", metric="l2", device=0)
    if n:
        ix.add(rows)
    wi, wd, _ = ko.search(rows, q, k, "l2")
    for mode in ("fast_only", "auto", "exact"):
        gi, gd, gc = _dev_search(ix, q, k, mode)
        assert gc.all() and np.array_equal(gi, wi) and np.array_equal(gd, wd, equal_nan=True)
    ix.close()


def test_readers_and_device_searches_survive_growth_and_compaction(hip):
    """Flask request threads (host searches) and an asynchronous device-resident searcher keep running while the single
    ingestion writer grows the buffers, deletes, re-adds and compacts (src/bin/service_data_manager.py:38,62-73 serialises
    writers; readers are concurrent): no HIP error, every answer sorted and made of ids that were live at some point, and the
    final state equals the oracle. The base rows are never deleted and the queries are copies of base rows, so the top-1 of
    every search is known throughout."""
    import threading
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(77)
    d, n_base = 128, 6000
    base = _unit(rng, n_base, d)
    ix = HipIndex(d, 1024, dtype="bf16", metric="cosine", device=0)           # far too small: grows at once
    ix.add(base, ids=np.arange(n_base, dtype=np.int64))
    probe = rng.choice(n_base, 16, replace=False)
    q = ko.round_through(base[probe], "bf16")
    stop = threading.Event()
    errors = []

    def reader():
        try:
            while not stop.is_set():
                gi, gd, gc = ix.search(q, 5)
                assert (gc == 5).all() and (np.diff(gd, axis=1) >= 0).all()
                assert np.array_equal(gi[:, 0], probe), "a base row lost its top-1 place"
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    def device_reader():
        try:
            st = torch.cuda.Stream()
            tq = torch.from_numpy(q).cuda()
            oi = torch.empty((16, 5), dtype=torch.int64, device="cuda"); od = torch.empty((16, 5), dtype=torch.float64, device="cuda")
            oc = torch.empty((16,), dtype=torch.int32, device="cuda")
            while not stop.is_set():
                with torch.cuda.stream(st):
                    ix.search_device(tq.data_ptr(), 16, 5, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream, mode="fast_only")
                st.synchronize()
                ok = oc.cpu().numpy().astype(bool)
                assert np.array_equal(oi.cpu().numpy()[ok, 0], probe[ok])
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=reader) for _ in range(3)] + [threading.Thread(target=device_reader)]
    for t in threads:
        t.start()
    live = {}
    nxt = n_base
    try:
        for cycle in range(25):
            rows = _unit(rng, 1500, d) * 1.0
            ids = np.arange(nxt, nxt + 1500, dtype=np.int64); nxt += 1500
            ix.add(rows, ids=ids)                                            # grows / reclaims tombstones
            for i, r in zip(ids, ko.round_through(rows, "bf16")):
                live[int(i)] = r
            if len(live) > 3000:
                victims = rng.choice(sorted(live), 1500, replace=False)
                ix.remove(victims)
                for v in victims:
                    del live[int(v)]
            if cycle % 8 == 7:
                ix.compact()
    finally:
        stop.set()
        for t in threads:
            t.join()
    assert not errors, errors[:3]
    ids = np.array(list(range(n_base)) + sorted(live), dtype=np.int64)
    stored = np.concatenate([ko.round_through(base, "bf16"), np.stack([live[i] for i in sorted(live)])])
    gi, gd, _ = ix.search(q, 10)
    wi, wd, _ = ko.search(stored, q, 10, "cosine", ids=ids)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    ix.close()


def test_row_filter_is_bound_to_one_layout_of_the_index(hip):
    """ak_index_slots hands out (slots, layout epoch) in one call; ak_index_search / ak_index_search_dev take the pair with the
    mask and return AK_ERR_STALE_FILTER -- without reading the mask -- once an add or a reclaim of tombstones has moved the
    index on (include/archi_knn.h). A delete alone does not: the mask still describes the slots, the row is dead in the index."""
    from archi_amd import StaleFilterError
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(3)
    d, n, k = 64, 5000, 10
    rows = _unit(rng, n, d)
    q = _unit(rng, 4, d)
    ix = HipIndex(d, n + 2000, dtype="f32", metric="cosine", device=0)
    ix.add(rows, ids=np.arange(n, dtype=np.int64))
    slots, ep = ix.layout()
    assert slots == n
    mask = (rng.random(n) < 0.3).astype(np.uint8)
    wi, wd, _ = ko.search(rows, q, k, "cosine", alive=mask)
    gi, gd, _ = ix.search(q, k, row_filter=mask, filter_epoch=ep)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    # a delete keeps the epoch: the old mask still works and the dead row is gone from the answer
    victim = int(wi[0, 0])
    ix.remove([victim])
    assert ix.layout() == (n, ep)
    gi2, _, _ = ix.search(q, k, row_filter=mask, filter_epoch=ep)
    assert victim not in gi2[0] and np.array_equal(gi2[0, :k - 1], wi[0, 1:])
    # an add moves it: same mask, same length claim -> refused; a SHORTER buffer than the index has slots is never read
    ix.add(_unit(rng, 10, d), ids=np.arange(n, n + 10, dtype=np.int64))
    slots2, ep2 = ix.layout()
    assert slots2 == n + 10 and ep2 != ep
    with pytest.raises(StaleFilterError):
        ix.search(q, k, row_filter=mask, filter_epoch=ep)
    with pytest.raises(StaleFilterError):
        ix.search(q, k, row_filter=mask, filter_epoch=ep2)          # right epoch, wrong length
    dmask = torch.from_numpy(mask).cuda()
    oi = torch.empty((4, k), dtype=torch.int64, device="cuda"); od = torch.empty((4, k), dtype=torch.float64, device="cuda")
    with pytest.raises(StaleFilterError):
        ix.search_device(torch.from_numpy(q).cuda().data_ptr(), 4, k, oi.data_ptr(), od.data_ptr(), 0, 0, mode="auto",
                         row_filter_ptr=dmask.data_ptr(), filter_len=n, filter_epoch=ep)
    # compaction renumbers the slots: the epoch moves although the slot count may not
    ix.remove(list(range(n, n + 10)))
    before = ix.layout()
    ix.compact()
    after = ix.layout()
    assert after[0] == n - 1 and after[1] != before[1]
    mask3 = np.ones(after[0], np.uint8)
    gi3, _, _ = ix.search(q, k, row_filter=mask3, filter_epoch=after[1])
    stored_alive = np.ones(n, np.uint8); stored_alive[victim] = 0
    wi3, _, _ = ko.search(rows, q, k, "cosine", alive=stored_alive)
    assert np.array_equal(gi3, wi3)
    ix.close()


def test_sharded_search_through_the_c_abi_at_world_size_one(hip):
    """ak_comm_unique_id / ak_comm_create / ak_index_search_sharded_dev (csrc/shardcomm.hip): the row-sharded search with the
    all-gather issued by the library itself (RCCL), exercised with the one rank a one-GPU box allows. The whole sequence runs --
    local FAST_ONLY scan into the exchange layout, ncclAllGather, merge + flag reduction, and the re-run of the queries the scan
    could not certify (a duplicate pile-up, a zero query, a WHERE mask) -- and must equal ak_index_search and the oracle."""
    from archi_amd.index import HipIndex
    from archi_amd.sharded import AbiShardedSearcher
    rng = np.random.default_rng(21)
    n, d, k = 20000, 64, 10
    rows = _unit(rng, n, d)
    rows[np.arange(300) * 37 + 5] = rows[5]                    # pile-up wider than the candidate lists: needs the re-run
    q = np.concatenate([rows[5][None], np.zeros((1, d), np.float32), _unit(rng, 21, d)])
    ix = HipIndex(d, n, dtype="bf16", metric="cosine", device=0)
    ix.add(rows)
    s = AbiShardedSearcher(ix)
    stored = ko.round_through(rows, "bf16")
    tq = torch.from_numpy(q).cuda()
    for mask in (None, (rng.random(n) < 0.5).astype(np.uint8)):
        wi, wd, _ = ko.search(stored, q, k, "cosine", alive=mask)
        hi_, hd, _ = ix.search(q, k, row_filter=mask)
        flt = None if mask is None else torch.from_numpy(mask).cuda()
        gi, gd = s.search(tq, k, row_filter=flt)
        torch.cuda.synchronize()
        gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
        assert s.last_open >= 2                                 # the pile-up and the zero query went through the re-run
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd, equal_nan=True)
        assert np.array_equal(gi, hi_) and np.array_equal(gd, hd, equal_nan=True)
    # odd batch (the flag words of the payload are padded to a whole int64) and a batch with nothing to re-run
    gi, gd = s.search(tq[2:9], k)
    wi, wd, _ = ko.search(stored, q[2:9], k, "cosine")
    assert s.last_open == 0 and np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gd.cpu().numpy(), wd)
    # a stale filter (built for an older layout epoch) fails the local scan: the rank still runs the exchange -- its code travels
    # in the payload's status word -- and the call returns AK_ERR_STALE_FILTER after it; the next search on the communicator works
    from archi_amd import StaleFilterError
    slots, epoch = ix.layout()
    flt = torch.ones((slots,), dtype=torch.uint8, device="cuda")
    with pytest.raises(StaleFilterError):
        s.search(tq, k, row_filter=flt, filter_epoch=epoch - 1)
    gi, gd = s.search(tq[2:9], k, row_filter=flt, filter_epoch=epoch)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gd.cpu().numpy(), wd)
    s.close()
    ix.close()
