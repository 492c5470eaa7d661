"""Soak: random LARGE configurations (seeded three-pass plans, many slices, padded query tiles) of the certified MFMA
path against the oracle on a sample of queries.  python3 scripts/gpu_soak.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archi_amd.index import HipIndex
from oracle import knn_oracle as ko

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = bad = 0
while time.time() < t_end:
    d = int(rng.choice([64, 128, 256, 384, 768]))
    n = int(rng.integers(100_000, 3_000_000))
    n = min(n, int(2.5e9 // (d * 2)))
    nq = int(rng.choice([1, 7, 32, 33, 64, 100, 128, 129, 256, 500, 1024]))
    k = int(rng.choice([1, 4, 10, 10, 10, 33, 100]))
    dtype = str(rng.choice(["bf16", "f16", "f32"])); metric = str(rng.choice(["cosine", "l2", "inner_product"]))
    norm = bool(rng.random() < 0.6) or metric != "cosine"
    if dtype == "f32":
        n = min(n, 1_000_000)
    ix = HipIndex(d, n, dtype=dtype, metric=metric, device=0)
    seed = int(rng.integers(1, 1 << 30))
    ix.generate(seed=seed, n=n, normalise=norm)
    q = ko.gen_rows(seed + 1, 1, 0, nq, d, True, "f32")
    gi, gd, gc, st = ix.search(q, k, mode="auto", return_stats=True)
    ns = max(1, min(nq, int(1.5e9 // (n * d))))
    sample = rng.choice(nq, size=ns, replace=False)
    stored = ko.gen_rows(seed, 0, 0, n, d, norm, dtype)
    oi, od, oc = ko.search(stored, q[sample], k, metric)
    ok = np.array_equal(gi[sample], oi) and np.array_equal(gd[sample], od, equal_nan=True)
    cases += 1; bad += (not ok)
    plan = ix.scan_plan(nq, k)
    print(f"{'ok ' if ok else 'BAD'} n={n} d={d} nq={nq} k={k} {dtype} {metric} norm={norm} cfg={plan.get('cfg_name')} "
          f"seed_slices={plan.get('ns_seed')} certified={st['certified']} second={st['second_chance']} exact={st['exact_reruns']}", flush=True)
    ix.close()
print(f"soak: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
