"""Phase cycle counters of the scan kernel (AK_SCAN_DBG)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AK_SCAN_DBG"] = "1"
import numpy as np, torch
from archi_amd.index import HipIndex
n, d, dtype, nq = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
ix = HipIndex(d, n, dtype=dtype, metric="cosine", device=0); ix.generate(seed=1234, n=n)
tmp = HipIndex(d, nq, dtype=dtype, metric="cosine", device=0); tmp.generate(seed=4321, n=nq, stream=1)
q = tmp.fetch(np.arange(nq)); tmp.close()
tq = torch.from_numpy(q).cuda(); k = 10
oi = torch.empty((nq, k), dtype=torch.int64, device="cuda"); od = torch.empty((nq, k), dtype=torch.float64, device="cuda")
oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
for _ in range(3):
    ix.search_device(tq.data_ptr(), nq, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
dbg = ix.debug_read()
names = ["k-loop", "filter", "sync", "compact", "final", "n_slow", "n_comp", "tiles"]
for li, lname in enumerate(("seed", "main")):
    a = dbg[li]; a = a[a[:, 7] > 0]
    if not len(a): continue
    print(f"{lname}: waves={len(a)} tiles/wave={a[:,7].mean():.1f}")
    tot = a[:, :5].sum(1).mean()
    stage = (a[:, 2] & 0xffffffff).astype(np.float64); wait = (a[:, 2] >> 32).astype(np.float64)
    a = a.astype(np.float64); a[:, 2] = 0
    for j in range(5):
        print(f"   {names[j]:8s} mean {a[:, j].mean()/1e3:9.1f} kcyc  ({100*a[:, j].mean()/tot:5.1f}%)  max {a[:, j].max()/1e3:9.1f}")
    print(f"   k-loop split: staging issue {stage.mean()/1e3:9.1f} kcyc, wait+barrier {wait.mean()/1e3:9.1f} kcyc, "
          f"fragments+MFMA {(a[:,0]-stage-wait).mean()/1e3:9.1f} kcyc; per k-step: "
          f"{stage.mean()/(a[:,7].mean()*int(sys.argv[2])//64):.0f} / {wait.mean()/(a[:,7].mean()*int(sys.argv[2])//64):.0f} / "
          f"{(a[:,0]-stage-wait).mean()/(a[:,7].mean()*int(sys.argv[2])//64):.0f} cycles")
    print(f"   slow-path entries/wave {a[:,5].mean():.1f} (of {a[:,7].mean()*8:.0f} groups)  compactions/wave {a[:,6].mean():.1f}   total {tot/1e3:.0f} kcyc")
ix.close()
