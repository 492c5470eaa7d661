"""GPU probe: fast path timing at scale + parity against the exact HIP path."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.index import HipIndex
from oracle import knn_oracle as ko

cfgs = [(1_000_000, 384, "bf16"), (10_000_000, 768, "bf16")]
if len(sys.argv) > 1:
    cfgs = [(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])]
for (n, d, dtype) in cfgs:
    t = time.time(); ix = HipIndex(d, n, dtype=dtype, metric="cosine", device=0)
    ix.generate(seed=1234, n=n); print(f"generate {n}x{d} {dtype}: {time.time()-t:.3f}s", flush=True)
    k = 10
    for nq in (1, 64, 128, 1024):
        q = ko.gen_rows(4321, 1, 0, nq, d, True, dtype)
        tq = torch.from_numpy(q).cuda()
        oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
        oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = 2.0 * nq * n * d
        print(f"  fast nq={nq}: {ms:.3f} ms/batch  {nq/ms*1e3:.0f} q/s  {flops/ms/1e9:.1f} TFLOP/s  "
              f"{n*d*2/ms/1e6:.1f} GB/s(alg)  certified={int(oc.sum())}/{nq}", flush=True)
        if nq <= 64:
            ei, ed, _ = ix.search(q[:8], k, mode="exact")
            ok = np.array_equal(oi.cpu().numpy()[:8], ei) and np.array_equal(od.cpu().numpy()[:8], ed)
            print(f"  parity vs exact HIP path (first 8 queries): {ok}", flush=True)
    ix.close()
