"""CPU suite: the C oracle against the committed golden kNN / L2-normalise vectors
(tests/golden/make_knn_fixtures.py: expected outputs from the independent numpy restatement)."""
import glob
import os

import numpy as np
import pytest

from oracle import knn_oracle as ko
from tests.golden import make_knn_fixtures as mk

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", sorted(mk.CASES))
def test_c_oracle_reproduces_golden_knn(name):
    f = np.load(os.path.join(GOLD, name + ".npz"))
    stored, qs, ids = mk.inputs(name)
    assert mk.checksum(stored, qs, ids) == str(f["inputs_sha256"])        # the seeded inputs are what was recorded
    k = mk.CASES[name][4]
    for metric in mk.METRICS:
        oi, od, cnt = ko.search(stored, qs, k, metric, ids=ids)
        assert np.array_equal(oi, f[f"ids_{metric}"]), (name, metric)
        assert np.array_equal(od, f[f"dist_{metric}"], equal_nan=True), (name, metric)
        assert np.array_equal(cnt, (f[f"ids_{metric}"] >= 0).sum(1))


def test_golden_adversarial_cases_contain_what_they_claim():
    f = np.load(os.path.join(GOLD, "knn_adv_ties_nan_N300_D64_Q8_k12.npz"))
    stored, qs, ids = mk.inputs("knn_adv_ties_nan_N300_D64_Q8_k12")
    d, i = f["dist_cosine"], f["ids_cosine"]
    # query 1 equals rows 3, 10, 11: three zero-ish distances, returned in id order
    tied = sorted(ids[[3, 10, 11]].tolist())
    assert i[1, :3].tolist() == tied and d[1, 0] == d[1, 1] == d[1, 2]
    assert np.isnan(d[2]).all()                                            # zero query: every cosine distance NaN
    assert i[2].tolist() == sorted(ids.tolist())[:12]                      # NaN ties break by id
    g = np.load(os.path.join(GOLD, "knn_adv_k_gt_n_N7_D32_Q3_k10.npz"))
    assert (g["ids_l2"][:, 7:] == -1).all() and np.isnan(g["dist_l2"][:, 7:]).all()


def test_l2_normalize_golden():
    f = np.load(os.path.join(GOLD, "l2norm_N24_D384.npz"))
    x = mk.l2_inputs()
    assert mk.checksum(x) == str(f["inputs_sha256"])
    got = ko.l2_normalize(x)
    # torch.nn.functional.normalize does not fix a summation order: the float32 norm may differ by a few ulp
    assert np.allclose(got, f["expected"], rtol=1e-6, atol=0)
    assert (got[5] == 0).all() and np.abs(np.linalg.norm(got[[0, 1, 2]], axis=1) - 1).max() < 1e-6
