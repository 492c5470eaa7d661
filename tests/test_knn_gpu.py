"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs. Bar: ids bit-exact, float8 distances bit-exact (integer/byte
class parity -- stricter than the 1e-5 the north star allows for scores)."""
import os

import numpy as np
import pytest

from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu

METRICS = ["cosine", "l2", "inner_product"]
DTYPES = ["f32", "bf16", "f16"]


def _mk(hip, rows, dtype, metric, ids=None, cap=None):
    from archi_amd.index import HipIndex
    ix = HipIndex(rows.shape[1], cap or max(len(rows), 1), dtype=dtype, metric=metric, device=0)
    if len(rows):
        ix.add(rows, ids=ids)
    return ix


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def _check(ix, stored, queries, k, metric, mode, ids=None, alive=None, row_filter=None):
    gi, gd, gc = ix.search(queries, k, mode=mode, row_filter=row_filter)
    oi, od, oc = ko.search(stored, queries, k, metric, ids=ids, alive=alive)
    assert np.array_equal(gi, oi), f"ids differ: {np.argwhere(gi != oi)[:5]}"
    assert np.array_equal(gd, od, equal_nan=True), "distances differ"
    assert np.array_equal(gc, oc)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("metric", METRICS)
def test_exact_path_matches_oracle(hip, dtype, metric):
    rng = np.random.default_rng(42)
    n, d, nq, k = 5000, 384, 11, 10
    rows = _unit(rng, n, d) * (1.0 if metric == "cosine" else 3.0)
    q = _unit(rng, nq, d)
    ids = rng.permutation(10 * n)[:n].astype(np.int64)
    ix = _mk(hip, rows, dtype, metric, ids=ids)
    stored = ko.round_through(rows, dtype)
    assert np.array_equal(ix.fetch(np.arange(n)), stored)       # storage rounding parity (RNE)
    _check(ix, stored, q, k, metric, "exact", ids=ids)
    ix.close()


def test_exact_odd_dim_and_reference_test_vector(hip):
    # tests/unit/test_postgres_vectorstore.py:44-50: [0.1,0.2,0.3]*128 ; and a 3-d index
    v = np.array([[0.1, 0.2, 0.3] * 128], dtype=np.float32)
    rng = np.random.default_rng(1)
    rows = np.concatenate([v, rng.standard_normal((50, 384)).astype(np.float32)])
    ix = _mk(hip, rows, "f32", "cosine")
    _check(ix, rows, v, 4, "cosine", "exact")
    ix.close()
    rows3 = rng.standard_normal((9, 3)).astype(np.float32)
    ix = _mk(hip, rows3, "f32", "l2")
    _check(ix, rows3, rows3[:2], 4, "l2", "exact")
    ix.close()


def test_pgvector_published_regression_values_through_the_abi(hip):
    """pgvector's own regression expectations (zero vector -> NaN, clamping, float32 overflow -> Infinity / NaN, negative
    inner product) come out of ak_index_search exactly: a one-row f32 index per case, the query as the other operand."""
    from tests.test_oracle_cpu import PGVECTOR_CASES
    for metric, a, b, want in PGVECTOR_CASES:
        ix = _mk(hip, np.asarray([b], np.float32), "f32", metric, ids=[7])
        gi, gd, gc = ix.search(np.asarray([a], np.float32), 1)
        assert gc[0] == 1 and gi[0, 0] == 7
        assert (np.isnan(gd[0, 0]) and np.isnan(want)) or gd[0, 0] == want, (metric, a, b, gd[0, 0], want)
        ix.close()


@pytest.mark.parametrize("mode", ["exact", "auto"])
def test_ties_nan_k_gt_n_empty(hip, mode):
    rng = np.random.default_rng(5)
    c = _unit(rng, 64, 32)
    c[10] = c[3]; c[40] = c[3]
    c[20] = 0.0
    ids = np.arange(64, dtype=np.int64)[::-1].copy()
    ix = _mk(hip, c, "f32", "cosine", ids=ids)
    _check(ix, c, c[3][None], 64, "cosine", mode, ids=ids)      # duplicates + NaN row, k == n
    _check(ix, c, c[:5], 100, "cosine", mode, ids=ids)          # k > n
    _check(ix, c, np.zeros((1, 32), np.float32), 5, "cosine", mode, ids=ids)  # zero query: all NaN, id order
    _check(ix, c, c[:3], 1, "cosine", mode, ids=ids)            # k = 1
    ix.close()
    from archi_amd.index import HipIndex
    ix = HipIndex(32, 8, dtype="f32", metric="cosine", device=0)  # empty index
    gi, gd, gc = ix.search(c[:2], 3, mode=mode)
    assert (gi == -1).all() and np.isnan(gd).all() and (gc == 0).all()
    ix.close()


@pytest.mark.parametrize("mode", ["exact", "auto"])
def test_remove_filter_count(hip, mode):
    rng = np.random.default_rng(8)
    n, d = 3000, 64
    rows, q = _unit(rng, n, d), _unit(rng, 4, d)
    ids = (np.arange(n, dtype=np.int64) * 7 + 3)
    ix = _mk(hip, rows, "bf16", "cosine", ids=ids)
    stored = ko.round_through(rows, "bf16")
    assert ix.count() == n
    kill = ids[rng.permutation(n)[:500]]
    assert ix.remove(kill) == 500 and ix.remove(kill) == 0 and ix.count() == n - 500
    alive = np.ones(n, np.uint8); alive[(kill - 3) // 7] = 0
    _check(ix, stored, q, 10, "cosine", mode, ids=ids, alive=alive)
    flt = (rng.random(n) < 0.3).astype(np.uint8)                 # WHERE clause (a7)
    _check(ix, stored, q, 10, "cosine", mode, ids=ids, alive=alive & flt, row_filter=flt)
    assert (ix.lookup(kill[:5]) == -1).all() and (ix.lookup(ids[alive == 1][:5]) >= 0).all()
    ix.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_device_generator_matches_oracle(hip, dtype):
    from archi_amd.index import HipIndex
    n, d = 4096, 768
    for normalise in (True, False):
        ix = HipIndex(d, n, dtype=dtype, metric="cosine", device=0)
        ix.generate(seed=1234, n=n, stream=2, row0=10_000_000_000, normalise=normalise, id0=0)
        got = ix.fetch(np.arange(n))
        want = ko.gen_rows(1234, 2, 10_000_000_000, n, d, normalise, dtype)
        assert np.array_equal(got, want)
        ix.close()


def test_merge_kernel_matches_oracle(hip):
    import torch
    from archi_amd.index import merge_topk_device
    rng = np.random.default_rng(21)
    g, nq, k = 8, 37, 10
    pd = np.sort(rng.random((g, nq, k)), axis=2)
    pd[2, 5, 7:] = np.nan
    pi = rng.permutation(g * nq * k).reshape(g, nq, k).astype(np.int64)
    pi[2, 5, 8:] = -1
    pd[3, 0, :] = pd[4, 0, :]                                    # cross-shard exact ties
    ti, td = torch.from_numpy(pi).cuda(), torch.from_numpy(pd).cuda()
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    merge_topk_device(g, nq, k, ti.data_ptr(), td.data_ptr(), oi.data_ptr(), od.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    wi, wd = ko.merge(pi, pd)
    assert np.array_equal(oi.cpu().numpy(), wi) and np.array_equal(od.cpu().numpy(), wd, equal_nan=True)


# ---------------------------------------------------------------------------
# fast path: MFMA candidate scan + exact re-rank + certificate
# ---------------------------------------------------------------------------
def _gen_index(dtype, metric, n, d, seed=1234, normalise=True):
    from archi_amd.index import HipIndex
    ix = HipIndex(d, n, dtype=dtype, metric=metric, device=0)
    ix.generate(seed=seed, n=n, normalise=normalise)
    return ix, ko.gen_rows(seed, 0, 0, n, d, normalise, dtype)


@pytest.mark.parametrize("dtype,d,n,nq", [("bf16", 768, 8192, 16), ("f16", 384, 20000, 200), ("bf16", 64, 5000, 130),
                                          ("bf16", 64, 20000, 130)])   # D=64: one K-step per tile (barrier-race regression)
@pytest.mark.parametrize("metric", METRICS)
def test_fast_path_matches_oracle(hip, dtype, d, n, nq, metric):
    ix, stored = _gen_index(dtype, metric, n, d, normalise=(metric == "cosine"))
    q = ko.gen_rows(4321, 1, 0, nq, d, True, "f32")
    gi, gd, gc, st = ix.search(q, 10, mode="fast_only", return_stats=True)
    oi, od, oc = ko.search(stored, q, 10, metric)
    assert st["certified"] >= 0.9 * nq, st           # the MFMA path really ran and certified
    gi2, gd2, gc2, st2 = ix.search(q, 10, mode="auto", return_stats=True)
    assert np.array_equal(gi2, oi) and np.array_equal(gd2, od) and np.array_equal(gc2, oc)
    if st["certified"] == nq:
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    ix.close()


def test_fast_path_unnormalised_cosine_and_k_sweep(hip):
    ix, stored = _gen_index("f16", "cosine", 30000, 384, normalise=False)   # cfg5 shape: fused normalise
    q = ko.gen_rows(99, 3, 0, 40, 384, False, "f32")
    for k in (1, 4, 10, 33, 100):
        gi, gd, gc, st = ix.search(q, k, mode="auto", return_stats=True)
        oi, od, oc = ko.search(stored, q, k, "cosine")
        assert np.array_equal(gi, oi) and np.array_equal(gd, od), (k, st)
    ix.close()


def test_fast_path_filter_remove_duplicates(hip):
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(31)
    n, d = 9000, 128
    rows = _unit(rng, n, d)
    rows[100:180] = rows[7]                     # 80 exact duplicates of one row (> k' - k is not reached, > k is)
    rows[500] = 0.0
    ids = rng.permutation(10 * n)[:n].astype(np.int64)
    ix = HipIndex(d, n, dtype="bf16", metric="cosine", device=0)
    ix.add(rows, ids=ids)
    stored = ko.round_through(rows, "bf16")
    q = np.concatenate([rows[7][None], _unit(rng, 20, d)])
    _check(ix, stored, q, 10, "cosine", "auto", ids=ids)
    kill = ids[rng.permutation(n)[:2000]]
    ix.remove(kill)
    assert (ix.lookup(kill) == -1).all()
    alive = np.isin(ids, kill, invert=True).astype(np.uint8)
    flt = (rng.random(n) < 0.5).astype(np.uint8)
    _check(ix, stored, q, 10, "cosine", "auto", ids=ids, alive=alive)
    _check(ix, stored, q, 10, "cosine", "auto", ids=ids, alive=alive & flt, row_filter=flt)
    ix.close()


def test_fast_path_adversarial_sorted_corpus(hip):
    """Rows ordered by increasing similarity to the query: every tile beats the threshold."""
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(77)
    n, d = 16384, 64
    rows = _unit(rng, n, d)
    q = _unit(rng, 3, d)
    order = np.argsort(rows @ q[0])
    rows = rows[order]
    ix = HipIndex(d, n, dtype="bf16", metric="cosine", device=0)
    ix.add(rows)
    _check(ix, ko.round_through(rows, "bf16"), q, 10, "cosine", "auto")
    ix.close()


def test_device_resident_search(hip):
    import torch
    ix, stored = _gen_index("bf16", "cosine", 50000, 768)
    nq, k = 256, 10
    q = ko.gen_rows(4321, 1, 0, nq, 768, True, "bf16")     # bf16-representable queries
    tq = torch.from_numpy(q).cuda()
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
    ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(),
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    wi, wd, _ = ko.search(stored, q, k, "cosine")
    cert = oc.cpu().numpy().astype(bool)
    assert cert.mean() > 0.95
    assert np.array_equal(oi.cpu().numpy()[cert], wi[cert]) and np.array_equal(od.cpu().numpy()[cert], wd[cert])
    ix.close()


@pytest.fixture
def scan_cfg():
    """Force a scan tile for the duration of a test (ak_debug_set: the library reads its environment only once)."""
    from archi_amd import _lib
    yield lambda cfg: _lib.debug_set("AK_SCAN_CFG", cfg)
    _lib.debug_set("AK_SCAN_CFG", None)


@pytest.mark.parametrize("cfg", ["P", "Q", "R", "L", "M", "S"])
def test_every_scan_tile_config_matches_oracle(hip, cfg, scan_cfg):
    """The tile configuration is a speed choice only: results are identical. (The A/B reference tiles X and O live in
    libarchi_hip_dbg.so: test_reference_tiles_of_the_dbg_library_match_oracle.)"""
    scan_cfg(cfg)
    ix, stored = _gen_index("bf16", "cosine", 70001, 192)      # ragged: n % 256 != 0, several slices
    q = ko.gen_rows(4321, 1, 0, 70, 192, True, "f32")
    gi, gd, gc, st = ix.search(q, 10, mode="fast_only", return_stats=True)
    oi, od, oc = ko.search(stored, q, 10, "cosine")
    assert st["certified"] == 70, st
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    ix.close()


def test_reference_tiles_of_the_dbg_library_match_oracle():
    """The A/B reference tiles X (256 x 256, in-step K-loop of rounds 1-2) and O (the first 128 x 128 version) are instantiated in
    libarchi_hip_dbg.so only; a child process loads that library (one library per process) and holds both to the oracle."""
    import subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    if not os.path.exists(os.path.join(os.path.dirname(here), "archi_amd", "lib", "libarchi_hip_dbg.so")):
        pytest.skip("libarchi_hip_dbg.so not built (make -C archi_amd/csrc dbg)")
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.dirname(here)!r})\n"
        "from archi_amd import _lib\n"
        "from archi_amd.index import HipIndex\n"
        "from oracle import knn_oracle as ko\n"
        "assert _lib.is_dbg_library()\n"
        "ix = HipIndex(192, 70001, dtype='bf16', metric='cosine', device=0)\n"
        "ix.generate(seed=1234, n=70001, normalise=True)\n"
        "stored = ko.gen_rows(1234, 0, 0, 70001, 192, True, 'bf16')\n"
        "q = ko.gen_rows(4321, 1, 0, 70, 192, True, 'f32')\n"
        "oi, od, oc = ko.search(stored, q, 10, 'cosine')\n"
        "for cfg in ('X', 'O'):\n"
        "    _lib.debug_set('AK_SCAN_CFG', cfg)\n"
        "    gi, gd, gc, st = ix.search(q, 10, mode='fast_only', return_stats=True)\n"
        "    assert st['certified'] == 70, (cfg, st)\n"
        "    assert np.array_equal(gi, oi) and np.array_equal(gd, od), cfg\n"
        "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("AK_")}
    env["ARCHI_HIP_DBG"] = "1"
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and b"ok" in p.stdout, p.stderr.decode("utf-8", "replace")[-3000:]


@pytest.mark.parametrize("metric", METRICS)
def test_f32_corpus_fast_path_through_bf16_shadow(hip, metric):
    """f32 storage (the reference's vector(D) column type): candidates from the bf16 shadow scan,
    ids/distances from the exact re-rank on the f32 rows."""
    ix, stored = _gen_index("f32", metric, 60000, 384, normalise=(metric == "cosine"))
    q = ko.gen_rows(4321, 1, 0, 33, 384, True, "f32")
    gi, gd, gc, st = ix.search(q, 10, mode="auto", return_stats=True)
    oi, od, oc = ko.search(stored, q, 10, metric)
    assert st["certified"] >= 30, st
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    ix.close()


def test_concurrent_searches_from_threads(hip):
    """Flask request threads call similarity_search concurrently (src/interfaces/chat_app/app.py:1554):
    ak_index_search must be re-entrant (per-thread stream, shared lock), also while a writer adds rows."""
    import threading
    ix, stored = _gen_index("bf16", "cosine", 30000, 128)
    q = ko.gen_rows(4321, 1, 0, 64, 128, True, "f32")
    want_i, want_d, _ = ko.search(stored, q, 10, "cosine")
    errors = []

    def reader(tid):
        try:
            for it in range(15):
                lo = (tid * 7 + it * 3) % 48
                gi, gd, _ = ix.search(q[lo:lo + 16], 10, mode="auto" if it % 2 else "exact")
                if not (np.array_equal(gi, want_i[lo:lo + 16]) and np.array_equal(gd, want_d[lo:lo + 16])):
                    errors.append((tid, it))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=reader, args=(t,)) for t in range(6)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors[:3]
    ix.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("metric", METRICS)
def test_clustered_corpus_below_the_scan_resolution(hip, dtype, metric):
    """Rows packed closer together than the 16-bit scan can resolve (a tight cluster around one direction + a far
    background): the certificate must notice that its margin swallows the k-th gap and hand those queries to the wide
    second scan or the exact path -- whatever the route, ids and distances equal the oracle's."""
    rng = np.random.default_rng(31)
    d, n = 128, 30000
    base = _unit(rng, 1, d)[0]
    rows = _unit(rng, n, d)
    cluster = rng.choice(n, size=4000, replace=False)
    rows[cluster] = base[None, :] + 2e-4 * rng.standard_normal((4000, d)).astype(np.float32)
    rows = ko.round_through(rows.astype(np.float32), dtype)
    q = np.stack([base, base + 1e-4 * rng.standard_normal(d).astype(np.float32), _unit(rng, 1, d)[0]]).astype(np.float32)
    ix = _mk(hip, rows, dtype, metric)
    gi, gd, gc, st = ix.search(q, 10, mode="auto", return_stats=True)
    oi, od, oc = ko.search(rows, q, 10, metric)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od), st
    assert st["certified"] + st["exact_reruns"] == 3
    ix.close()


@pytest.mark.parametrize("metric", METRICS)
def test_shadow_rounding_aligned_with_the_query(hip, metric):
    """f32 corpus, worst case for the bf16 shadow: a cluster of rows that share ONE shadow (same bf16 values) while their
    float32 remainders are all aligned with the query (+) or against it (-), i.e. the dot-product error of each row sits
    near the Cauchy-Schwarz bound the certificate assumes (rho_c = max |a - shadow(a)| / |a|, measured at ingest) and the
    scan cannot tell the rows apart at all. Exact ids and distances must still come back (through the certificate when
    the re-ranked list covers the cluster, else through the wide second scan or the exact path)."""
    rng = np.random.default_rng(77)
    d, n, m = 64, 20000, 300
    q = _unit(rng, 1, d)[0]
    grid = ko.round_through(_unit(rng, 1, d), "bf16")[0]                 # one direction, on the bf16 grid
    grid = ko.round_through((0.8 * q + 0.6 * grid).astype(np.float32)[None], "bf16")[0]
    ulp = np.abs(grid) * 2.0 ** -8                                       # bf16 spacing is 2^-7 relative: stay inside half of it
    rows = _unit(rng, n, d)
    sign = np.where(rng.random(m) < 0.5, 1.0, -1.0).astype(np.float32)
    frac = rng.uniform(0.05, 0.45, size=(m, 1)).astype(np.float32)
    cluster = grid[None, :] + (sign[:, None] * frac) * ulp[None, :] * np.sign(q)[None, :]
    assert np.array_equal(ko.round_through(cluster.astype(np.float32), "bf16"), np.repeat(grid[None], m, 0))   # one shadow
    rows[:m] = cluster
    perm = rng.permutation(n)
    rows = np.ascontiguousarray(rows[perm], np.float32)
    ix = _mk(hip, rows, "f32", metric)
    qs = np.stack([q, grid / np.linalg.norm(grid)]).astype(np.float32)
    for k in (10, 40):
        gi, gd, gc, st = ix.search(qs, k, mode="auto", return_stats=True)
        oi, od, oc = ko.search(rows, qs, k, metric)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od), (k, st)
    ix.close()


def test_thread_per_request_lifecycle(hip):
    """A thread-per-request server creates and drops a thread for every search: each thread's stream and cached scratch
    (device block, workspace, pinned staging) are built on its first call and released when it ends."""
    import threading
    ix, stored = _gen_index("bf16", "cosine", 20000, 64)
    q = ko.gen_rows(4321, 1, 0, 8, 64, True, "f32")
    want_i, want_d, _ = ko.search(stored, q, 5, "cosine")
    bad = []

    def one(i):
        gi, gd, _ = ix.search(q[i % 8], 5)
        if not (np.array_equal(gi[0], want_i[i % 8]) and np.array_equal(gd[0], want_d[i % 8])):
            bad.append(i)

    for i in range(60):
        t = threading.Thread(target=one, args=(i,))
        t.start()
        t.join()
    assert not bad
    ix.close()


# ---- committed golden vectors (tests/golden/make_knn_fixtures.py) ---------------------------------
def _golden_cases():
    from tests.golden import make_knn_fixtures as mk
    return sorted(mk.CASES)


@pytest.mark.parametrize("name", _golden_cases())
def test_hip_reproduces_golden_knn(hip, name):
    import os
    from archi_amd.index import HipIndex
    from tests.golden import make_knn_fixtures as mk
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    stored, qs, ids = mk.inputs(name)
    dtype, n, d, q, k, seed = mk.CASES[name]
    for metric in mk.METRICS:
        ix = HipIndex(d, n, dtype=dtype, metric=metric, device=0)
        ix.add(stored, ids=ids)                      # already storage-rounded: the add is value-preserving
        for mode in ("exact", "auto"):
            gi, gd, gc = ix.search(qs, k, mode=mode)
            assert np.array_equal(gi, f[f"ids_{metric}"]), (name, metric, mode)
            assert np.array_equal(gd, f[f"dist_{metric}"], equal_nan=True), (name, metric, mode)
        ix.close()


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_seeded_plan_large_k(hip, metric):
    """A shard big enough for all three passes (pre-seeding, seeding pass, main pass) with k = 10 / 33 / 100:
    k' = 64 / 128 / 256 changes the seed-list width (k_seed_kth_lists<16|32|64>) and the final selection width
    (k_finalize<1|2|4>). 1024 queries are searched; a sample of them is checked against the oracle."""
    n, d, nq = 600_000, 64, 1024
    ix, stored = _gen_index("bf16", metric, n, d, normalise=(metric == "cosine"))
    plan = ix.scan_plan(nq, 10)
    assert plan["ns_seed"] > 0, plan                              # the seeded three-pass plan is what runs
    q = ko.gen_rows(4321, 1, 0, nq, d, True, "f32")
    sample = np.arange(0, nq, 43)
    for k in (10, 33, 100):
        gi, gd, gc, st = ix.search(q, k, mode="fast_only", return_stats=True)
        assert st["certified"] >= 0.97 * nq, (k, st)
        oi, od, oc = ko.search(stored, q[sample], k, metric)
        gi2, gd2, gc2 = ix.search(q, k, mode="auto")
        assert np.array_equal(gi2[sample], oi) and np.array_equal(gd2[sample], od), k
    ix.close()


@pytest.mark.parametrize("dtype,metric,n,d", [("bf16", "cosine", 600_000, 128), ("f16", "l2", 300_000, 256), ("f32", "inner_product", 200_000, 384)])
def test_192_query_tile_across_dtypes_k_filters_and_group_fill(hip, dtype, metric, n, d, scan_cfg):
    """The phased 256 x 192 tile (the plan's choice between the regimes; forced here on shards the plan would give another tile):
    one / two / three query groups, the last one partly filled, dense lists (k = 10) and the slot layout of the wide plans
    (k = 33, 100), a WHERE mask, the bf16 shadow of an f32 corpus; a sample of the queries against the oracle, and every query
    against the 128-query tile's answer."""
    ix, stored = _gen_index(dtype, metric, n, d, normalise=(metric == "cosine"))
    mask = (np.arange(n) % 5 != 0).astype(np.uint8)
    for nq in (150, 200, 390, 577):
        q = ko.gen_rows(9000 + nq, 1, 0, nq, d, True, "f32")
        sample = np.unique(np.concatenate([np.arange(0, nq, 37), np.arange(max(nq - 20, 0), nq)]))
        for k in ((10, 33, 100) if nq == 390 else (10,)):
            scan_cfg("R")
            assert ix.scan_plan(nq, k)["cfg_name"] == "256x192 phased"
            ri, rd, rc, st = ix.search(q, k, mode="auto", return_stats=True)
            assert st["certified"] >= 0.9 * nq, (nq, k, st)
            fi, fd, fc = ix.search(q, k, mode="auto", row_filter=mask)
            scan_cfg("L")
            li, ld, lc = ix.search(q, k, mode="auto")
            gi, gd, gc = ix.search(q, k, mode="auto", row_filter=mask)
            assert np.array_equal(ri, li) and np.array_equal(rd, ld) and np.array_equal(rc, lc), (nq, k)
            assert np.array_equal(fi, gi) and np.array_equal(fd, gd), (nq, k)
            oi, od, oc = ko.search(stored, q[sample], k, metric)
            assert np.array_equal(ri[sample], oi) and np.array_equal(rd[sample], od), (nq, k)
            oi, od, oc = ko.search(stored, q[sample[:4]], k, metric, alive=mask)
            assert np.array_equal(fi[sample[:4]], oi) and np.array_equal(fd[sample[:4]], od), (nq, k)
    ix.close()


def test_duplicate_pileup_is_certified_by_the_wide_second_scan(hip):
    """200 byte-identical chunks (boilerplate repeated across documents) tie at the top of one query: more equal
    scores than the k' = 64 candidate list holds, so the first scan cannot certify it. The second scan with the
    widest lists must (no exact-path rerun), and the answer is the oracle's: ties broken by id."""
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(5)
    n, d = 40000, 128
    rows = _unit(rng, n, d)
    dup = rng.choice(n, size=200, replace=False)
    rows[dup] = rows[dup[0]]
    ids = rng.permutation(10 * n)[:n].astype(np.int64)
    ix = HipIndex(d, n, dtype="bf16", metric="cosine", device=0)
    ix.add(rows, ids=ids)
    stored = ko.round_through(rows, "bf16")
    q = np.concatenate([rows[dup[0]][None], _unit(rng, 30, d)])
    gi, gd, gc, st = ix.search(q, 10, mode="auto", return_stats=True)
    oi, od, oc = ko.search(stored, q, 10, "cosine", ids=ids)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert gi[0].tolist() == sorted(ids[dup].tolist())[:10]
    assert st["second_chance"] >= 1 and st["exact_reruns"] == 0, st
    ix.close()


@pytest.mark.parametrize("dtype,metric", [("f32", "cosine"), ("bf16", "l2"), ("f16", "inner_product")])
def test_tail_selection_with_hundreds_of_survivors(hip, dtype, metric):
    """Collections without a seeding pass (up to ~1M chunks) hand k_tail 600-800 surviving keys per query: the best k' = 64 are
    found by a histogram cut + rank counting (csrc/exact.hip, round 5; a bitonic network before). Bit-exact against the oracle at
    several batch sizes; and with 700 byte-identical rows at the top of one query -- every key up to the cut bin is an equal
    score: the pile-up falls back to the network, the query goes to the wide second scan, the answer is still the oracle's."""
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(11)
    n, d = 300_000, 128
    rows = _unit(rng, n, d) * (1.0 if metric == "cosine" else 2.0)
    ids = rng.permutation(4 * n)[:n].astype(np.int64)
    ix = HipIndex(d, n, dtype=dtype, metric=metric, device=0)
    ix.add(rows, ids=ids)
    stored = ko.round_through(rows, dtype)
    assert ix.scan_plan(8, 10)["ns_seed"] == 0                 # no seeding pass at this size: every appended key survives the cut-off
    for nq in (1, 8, 70, 300):
        q = _unit(rng, nq, d)
        gi, gd, gc = ix.search(q, 10, mode="auto")
        oi, od, oc = ko.search(stored, q, 10, metric, ids=ids)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True) and np.array_equal(gc, oc), nq
    ix.close()
    if metric != "cosine":
        return
    dup = rng.choice(n, size=700, replace=False)
    rows[dup] = rows[dup[0]]
    ix = HipIndex(d, n, dtype=dtype, metric=metric, device=0)
    ix.add(rows, ids=ids)
    stored = ko.round_through(rows, dtype)
    q = np.concatenate([rows[dup[0]][None], _unit(rng, 5, d)])
    gi, gd, gc, st = ix.search(q, 10, mode="auto", return_stats=True)
    oi, od, oc = ko.search(stored, q, 10, metric, ids=ids)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert gi[0].tolist() == sorted(ids[dup].tolist())[:10]
    ix.close()


def test_error_behaviour_through_the_abi(hip):
    """Errors come back as negative codes + ak_last_error text and are raised, never swallowed."""
    from archi_amd import HipBackendError
    from archi_amd.index import HipIndex
    rng = np.random.default_rng(0)
    ix = HipIndex(32, 100, dtype="bf16", metric="cosine", device=0)
    ix.add(_unit(rng, 60, 32), ids=np.arange(60))
    ix.add(_unit(rng, 50, 32), ids=np.arange(100, 150))     # past the first reservation: the index grows (the table has no capacity)
    assert ix.count() == 110 and ix.allocated_rows >= 110
    ix.remove(np.arange(100, 150))
    with pytest.raises(HipBackendError, match="duplicate id"):
        ix.add(_unit(rng, 2, 32), ids=np.array([5, 200]))
    with pytest.raises(HipBackendError, match="ids must be >= 0"):
        ix.add(_unit(rng, 1, 32), ids=np.array([-3]))
    assert ix.count() == 60                                   # failed adds left nothing behind
    with pytest.raises(HipBackendError, match="k > 4096"):
        ix.search(_unit(rng, 1, 32), 5000)
    with pytest.raises(ValueError):
        ix.search(_unit(rng, 1, 31), 5)                       # wrong query dimension is caught on the host side
    with pytest.raises(ValueError):
        HipIndex(32, 10, dtype="int8", metric="cosine", device=0)
    with pytest.raises(ValueError):
        HipIndex(32, 10, dtype="bf16", metric="manhattan", device=0)
    ix.remove([5])
    ix.add(_unit(rng, 1, 32), ids=np.array([5]))              # a deleted id may be re-inserted
    assert ix.count() == 60
    ix.close()


def test_randomised_differential_against_oracle(hip):
    """120 seeded random configurations (dimension incl. odd and non-multiples of 64, size, batch, k, dtype, metric,
    ids, deletions, WHERE-mask, duplicated and zero rows, unnormalised data) through AUTO mode vs the oracle:
    ids, float8 distances and counts identical."""
    from archi_amd.index import HipIndex
    import os
    rng = np.random.default_rng(int(os.environ.get("AK_TEST_SEED", "20260101")))    # AK_TEST_SEED: soak runs with other seeds
    dims = [3, 17, 64, 100, 128, 192, 384, 768]
    ran_fast = 0
    for case in range(120):
        d = int(rng.choice(dims))
        n = int(rng.integers(1, 40000))
        nq = int(rng.integers(1, 200))
        while n * nq * d > 4e8:                       # keep the oracle side to a second or so per case
            n = max(1, n // 2); nq = max(1, nq // 2)
        k = int(rng.integers(1, 60))
        dtype = str(rng.choice(DTYPES)); metric = str(rng.choice(METRICS))
        scale = float(rng.choice([1.0, 1.0, 0.05, 7.0]))
        rows = (rng.standard_normal((n, d)) * scale).astype(np.float32)
        if rng.random() < 0.5:
            rows /= np.maximum(np.linalg.norm(rows, axis=1, keepdims=True), 1e-12)
        if n > 20 and rng.random() < 0.4:
            m = int(rng.integers(2, min(n // 2, 120)))
            rows[rng.choice(n, m, replace=False)] = rows[0]         # duplicates: ties by id
        if n > 5 and rng.random() < 0.3:
            rows[int(rng.integers(0, n))] = 0.0                     # zero row: NaN cosine distance
        ids = (rng.permutation(4 * n + 10)[:n]).astype(np.int64)
        q = rng.standard_normal((nq, d)).astype(np.float32)
        if rng.random() < 0.3:
            q[0] = rows[0]
        ix = HipIndex(d, n, dtype=dtype, metric=metric, device=0)
        ix.add(rows, ids=ids)
        stored = ko.round_through(rows, dtype)
        alive = np.ones(n, np.uint8)
        if n > 10 and rng.random() < 0.4:
            kill = ids[rng.choice(n, int(rng.integers(1, n // 3 + 1)), replace=False)]
            ix.remove(kill)
            alive = np.isin(ids, kill, invert=True).astype(np.uint8)
        flt = None
        if rng.random() < 0.3:
            flt = (rng.random(n) < 0.6).astype(np.uint8)
        gi, gd, gc, st = ix.search(q, k, mode="auto", row_filter=flt, return_stats=True)
        oi, od, oc = ko.search(stored, q, k, metric, ids=ids, alive=alive if flt is None else alive & flt)
        tag = (case, d, n, nq, k, dtype, metric, st)
        assert np.array_equal(gi, oi), tag
        assert np.array_equal(gd, od, equal_nan=True), tag
        assert np.array_equal(gc, oc), tag
        ran_fast += st["certified"] > 0
        ix.close()
    assert ran_fast >= 20          # the MFMA path took part (it needs dim % 64 == 0 and >= 4096 rows)
