"""CPU suite: pins the oracle (oracle/knn_oracle.c) against independent statements
of the same arithmetic and against published known-answer vectors."""
import numpy as np
import pytest

from oracle import knn_oracle as ko

METRICS = ["cosine", "l2", "inner_product"]


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def test_philox_known_answers():
    # Random123 kat_vectors (philox4x32-10): published known-answer tests
    assert [hex(v) for v in ko.philox([0, 0, 0, 0], [0, 0])] == \
        ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in ko.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in ko.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                      [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_storage_rounding_matches_numpy_and_torch():
    import torch
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(100000) * np.exp(rng.uniform(-30, 12, 100000))).astype(np.float32)
    x[:8] = [0.0, -0.0, 65504.0, 65520.0, 6.1e-5, 5.96e-8, 2.98e-8, -1e-10]
    with np.errstate(over="ignore"):
        assert np.array_equal(ko.round_through(x, "f16"), x.astype(np.float16).astype(np.float32))
    assert np.array_equal(ko.round_through(x, "bf16"),
                          torch.from_numpy(x).to(torch.bfloat16).float().numpy())


@pytest.mark.parametrize("metric", METRICS)
def test_c_oracle_equals_numpy_restatement(metric):
    rng = np.random.default_rng(11)
    c = rng.standard_normal((700, 96)).astype(np.float32)
    q = rng.standard_normal((5, 96)).astype(np.float32)
    i1, d1, cnt = ko.search(c, q, 10, metric)
    i2, d2 = ko.search_numpy(c, q, 10, metric)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2) and (cnt == 10).all()


def test_reference_test_vector_score_convention():
    # tests/unit/test_postgres_vectorstore.py:44-50 uses [0.1,0.2,0.3]*128 as both the
    # stored and the query embedding; cosine distance of a vector with itself is ~0
    v = np.array([0.1, 0.2, 0.3] * 128, dtype=np.float32)
    d = ko.distance("cosine", v, v)
    assert abs(d) < 1e-6
    assert ko.distance("l2", v, v) == 0.0
    assert ko.distance("inner_product", v, v) < 0


@pytest.mark.parametrize("metric", METRICS)
def test_f32_path_within_1e5_of_f64(metric):
    # north_star tolerance: cosine scores within 1e-5 (fp32)
    rng = np.random.default_rng(3)
    c, q = _unit(rng, 200, 768), _unit(rng, 1, 768)[0]
    for r in c:
        assert abs(ko.distance(metric, r, q) - ko.distance_f64(metric, r, q)) < 1e-5


def test_ties_break_by_id_and_nan_last():
    rng = np.random.default_rng(5)
    c = _unit(rng, 64, 32)
    c[10] = c[3]; c[40] = c[3]            # exact duplicates -> exact ties
    c[20] = 0.0                           # zero vector -> NaN cosine distance
    ids = np.arange(64, dtype=np.int64)[::-1].copy()   # ids descending in row order
    i, d, cnt = ko.search(c, c[3][None], 64, "cosine", ids=ids)
    assert list(i[0, :3]) == sorted([ids[3], ids[10], ids[40]])
    assert d[0, 0] == d[0, 1] == d[0, 2]
    assert i[0, -1] == ids[20] and np.isnan(d[0, -1]) and cnt[0] == 64
    assert not np.isnan(d[0, :-1]).any() and (np.diff(d[0, :-1]) >= 0).all()


def test_k_larger_than_n_and_filter_and_empty():
    rng = np.random.default_rng(9)
    c, q = _unit(rng, 7, 16), _unit(rng, 2, 16)
    i, d, cnt = ko.search(c, q, 10, "cosine")
    assert (cnt == 7).all() and (i[:, 7:] == -1).all() and np.isnan(d[:, 7:]).all()
    alive = np.array([1, 0, 1, 0, 1, 0, 1], dtype=np.uint8)
    i, d, cnt = ko.search(c, q, 3, "l2", alive=alive)
    assert set(i.ravel()) <= {0, 2, 4, 6}
    i, d, cnt = ko.search(np.zeros((0, 16), np.float32), q, 3, "cosine")
    assert (cnt == 0).all() and (i == -1).all()


def test_merge_equals_single_shard():
    rng = np.random.default_rng(13)
    c, q = _unit(rng, 999, 64), _unit(rng, 6, 64)
    ids = rng.permutation(5000)[:999].astype(np.int64)
    full_i, full_d, _ = ko.search(c, q, 10, "cosine", ids=ids)
    for g in (1, 2, 4, 8):
        bounds = np.linspace(0, 999, g + 1).astype(int)
        pi = np.stack([ko.search(c[a:b], q, 10, "cosine", ids=ids[a:b])[0] for a, b in zip(bounds[:-1], bounds[1:])])
        pd = np.stack([ko.search(c[a:b], q, 10, "cosine", ids=ids[a:b])[1] for a, b in zip(bounds[:-1], bounds[1:])])
        mi, md = ko.merge(pi, pd)
        assert np.array_equal(mi, full_i) and np.array_equal(md, full_d)


def test_generator_properties():
    r = ko.gen_rows(1234, 0, 0, 256, 768, True, "f32")
    assert np.allclose(np.linalg.norm(r.astype(np.float64), axis=1), 1.0, atol=1e-6)
    # counter-based: any row range reproduces
    assert np.array_equal(ko.gen_rows(1234, 0, 100, 10, 768, True, "f32"), r[100:110])
    # distinct streams / seeds differ
    assert not np.array_equal(ko.gen_rows(1234, 1, 0, 4, 768, True, "f32"), r[:4])
    raw = ko.gen_rows(1234, 0, 0, 2000, 384, False, "f16")
    assert abs(raw.mean()) < 0.01 and abs(raw.std() - 209.0 / 256) < 0.01
    b = ko.gen_rows(1234, 0, 0, 16, 768, True, "bf16")
    assert np.array_equal(b, ko.round_through(r[:16], "bf16"))


def test_l2_normalize_matches_torch():
    import torch
    rng = np.random.default_rng(17)
    x = rng.standard_normal((33, 384)).astype(np.float32)
    x[5] = 0.0
    ref = torch.nn.functional.normalize(torch.from_numpy(x), p=2, dim=1).numpy()
    assert np.allclose(ko.l2_normalize(x), ref, atol=1e-6)


# pgvector's own regression expectations for the dense vector type [upstream pgvector: test/sql/functions.sql and
# test/expected/functions.out, v0.5-0.8; the extension is not in this image, the cases are restated from the published
# repository]. They pin what the formula alone does not say: a zero vector gives NaN, similarity is clamped to [-1, 1],
# float32 overflow gives Infinity / NaN, "<#>" is the NEGATIVE inner product.
PGVECTOR_CASES = [
    ("l2", [0, 0], [3, 4], 5.0), ("l2", [0, 0], [0, 1], 1.0), ("l2", [3e38], [-3e38], float("inf")),
    ("inner_product", [1, 2], [3, 4], -11.0), ("inner_product", [3e38], [3e38], float("-inf")),
    ("cosine", [1, 2], [2, 4], 0.0), ("cosine", [1, 2], [0, 0], float("nan")), ("cosine", [1, 1], [1, 1], 0.0),
    ("cosine", [1, 0], [0, 2], 1.0), ("cosine", [1, 1], [-1, -1], 2.0), ("cosine", [1, 1], [1.1, 1.1], 0.0),
    ("cosine", [1, 1], [-1.1, -1.1], 2.0), ("cosine", [3e38], [3e38], float("nan")),
]


@pytest.mark.parametrize("metric,a,b,want", PGVECTOR_CASES)
def test_pgvector_published_regression_values(metric, a, b, want):
    got = ko.distance(metric, np.asarray(a, np.float32), np.asarray(b, np.float32))
    assert (np.isnan(got) and np.isnan(want)) or got == want, (metric, a, b, got, want)
