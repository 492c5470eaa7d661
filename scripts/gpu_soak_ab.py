"""Soak of the phased scan kernels (round 3): EVERY query of large batches through the default plan (phased 256x256 /
256x192 / 256x128 tiles), through the 256x192 tile forced (AK_SCAN_CFG=R) and through the in-step kernels of rounds 1-2 (AK_SCAN_CFG=X / L) -- two independent K-loop structures that
must return identical ids and float8 distance bits -- repeated with fresh query sets (a staging race would be timing
dependent).  python3 scripts/gpu_soak_ab.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("ARCHI_HIP_DBG", "1")     # the in-step reference tile X lives in libarchi_hip_dbg.so (make -C archi_amd/csrc dbg)
from archi_amd import _lib
from archi_amd.index import HipIndex

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
rng = np.random.default_rng(7)
shapes = [(10_000_000, 768, "bf16"), (3_000_000, 384, "f16"), (6_000_000, 256, "bf16"), (1_000_000, 384, "f32"), (2_000_000, 64, "bf16")]
cases = bad = 0
for n, d, dtype in shapes:
    if time.time() > t_end: break
    ix = HipIndex(d, n, dtype=dtype, metric="cosine", device=0)
    ix.generate(seed=int(rng.integers(1, 1 << 30)), n=n, normalise=True)
    per_shape_end = time.time() + budget / len(shapes)
    while time.time() < min(t_end, per_shape_end):
        nq = int(rng.choice([1024, 1024, 768, 512, 384, 300, 256, 200, 129]))
        k = int(rng.choice([10, 10, 4, 16]))
        tmp = HipIndex(d, nq, dtype="f32", metric="cosine", device=0)
        tmp.generate(seed=int(rng.integers(1, 1 << 30)), n=nq, stream=1)
        q = tmp.fetch(np.arange(nq)); tmp.close()
        _lib.debug_set("AK_SCAN_CFG", None)
        plan = ix.scan_plan(nq, k)["cfg_name"]
        a = ix.search(q, k, return_stats=True)
        # in-step reference: tile X exists only in the dbg library; with ARCHI_HIP_DBG= (empty) the PRODUCT library is soaked against its
        # own in-step 256 x 128 tile (round 5: the product's phased kernels carry the v_max3 filter and the carried lane constants)
        _lib.debug_set("AK_SCAN_CFG", "X" if (nq > 256 and _lib.is_dbg_library()) else "L")
        b = ix.search(q, k, return_stats=True)
        _lib.debug_set("AK_SCAN_CFG", "R")                   # the 256 x 192 phased tile, whatever the plan would pick
        c = ix.search(q, k, return_stats=True)
        _lib.debug_set("AK_SCAN_CFG", None)
        ok = (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True) and
              np.array_equal(c[0], b[0]) and np.array_equal(c[1], b[1], equal_nan=True))
        cases += 1; bad += (not ok)
        print(f"{'ok ' if ok else 'BAD'} {n}x{d} {dtype} nq={nq} k={k} plan={plan} certified {a[3]['certified']}/{b[3]['certified']} "
              f"exact reruns {a[3]['exact_reruns']}/{b[3]['exact_reruns']}", flush=True)
    ix.close()
print(f"soak A/B: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
