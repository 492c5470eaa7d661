"""Latency of the host-buffer entry point (ak_index_search through HipIndex.search) vs the device-resident one."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.index import HipIndex
n, d = int(sys.argv[1]), int(sys.argv[2])
ix = HipIndex(d, n, dtype="bf16", metric="cosine", device=0); ix.generate(seed=1234, n=n)
for nq in (1, 16, 256):
    tmp = HipIndex(d, nq, dtype="bf16", metric="cosine", device=0); tmp.generate(seed=4321, n=nq, stream=1)
    q = tmp.fetch(np.arange(nq)); tmp.close()
    for _ in range(3): ix.search(q, 10)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps): ids, dist, cnt = ix.search(q, 10)
    host_ms = (time.perf_counter() - t0) / reps * 1e3
    tq = torch.from_numpy(q).cuda(); k = 10
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda"); od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
    oc = torch.empty((nq,), dtype=torch.int32, device="cuda"); st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
    torch.cuda.synchronize(); dev_ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"n={n} nq={nq:4d}: host-buffer call {host_ms:7.3f} ms   device-resident {dev_ms:7.3f} ms   overhead {host_ms-dev_ms:6.3f} ms", flush=True)
ix.close()
