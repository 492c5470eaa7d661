"""GPU probe: scan tile configurations at scale (scan-kernel time via HIP events)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
runs = [a.split(":") for a in sys.argv[4:]]   # nq:cfg[:blocks[:ablate]]
# the product library unless a run needs what only libarchi_hip_dbg.so has (tiles X / O, the ablation switch); ARCHI_HIP_DBG=1 in
# the environment selects the dbg library for everything (its main-pass kernel keeps the round-4 filter: the A/B reference)
if any((len(r) > 1 and r[1][:1] in ("X", "O")) or len(r) > 3 for r in runs):
    os.environ.setdefault("ARCHI_HIP_DBG", "1")
from archi_amd import _lib
from archi_amd.index import HipIndex

n, d, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ix = HipIndex(d, n, dtype=dtype, metric="cosine", device=0)
ix.generate(seed=1234, n=n)
k = 10
for r in runs:
    nq, cfg = int(r[0]), r[1]
    _lib.debug_set("AK_SCAN_CFG", cfg)            # (the library reads its environment once: switches change through ak_debug_set)
    _lib.debug_set("AK_SCAN_BLOCKS", r[2] if len(r) > 2 and r[2] else None)
    if len(r) > 3:                                # (the product library does not know the ablation switch)
        _lib.debug_set("AK_SCAN_ABLATE", r[3])
    elif _lib.is_dbg_library():
        try: _lib.debug_set("AK_SCAN_ABLATE", None)
        except Exception: pass
    tmp = HipIndex(d, nq, dtype=dtype, metric="cosine", device=0); tmp.generate(seed=4321, n=nq, stream=1)
    q = tmp.fetch(np.arange(nq)); tmp.close()
    tq = torch.from_numpy(q).cuda()
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
    torch.cuda.synchronize()
    ix.profile(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    scan = float(ix.profile_read().mean()); ix.profile(False)
    flops = 2.0 * nq * n * d
    print(f"nq={nq:5d} cfg={cfg} blocks={r[2] if len(r)>2 and r[2] else '-':>4} abl={r[3] if len(r)>3 else '0'}: total {ms:7.3f} ms  scan {scan:7.3f} ms  "
          f"{nq/ms*1e3:9.0f} q/s  scan: {flops/scan/1e9:7.1f} TF  {n*d*2/scan/1e6:7.1f} GB/s  cert={int(oc.sum())}/{nq}", flush=True)
ix.close()
