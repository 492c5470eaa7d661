"""TEST INFRASTRUCTURE: runs bench.py's N > 1 code path (sharding, one all-gather per search, merge, certificate reduction,
in-run verification against the single index, replicated-corpus leg, max-over-ranks timing) with several ranks sharing the
test box's ONE GPU. RCCL refuses two ranks on one device, so the process group is gloo and the 168 KB payload of the
all-gather is staged through the host; CUDA tensors handed to all_reduce go through the host as well. Everything else is
bench.py as the driver launches it. Numbers from this are meaningless; what is checked is that the path runs and verifies."""
import os
import runpy
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LOCAL_RANK"] = "0"                       # every rank on cuda:0

_init = dist.init_process_group
dist.init_process_group = lambda backend=None, **kw: _init("gloo")
_all_reduce = dist.all_reduce


def all_reduce(t, op=dist.ReduceOp.SUM, **kw):
    if t.is_cuda:
        h = t.cpu()
        _all_reduce(h, op=op, **kw)
        t.copy_(h)
    else:
        _all_reduce(t, op=op, **kw)


dist.all_reduce = all_reduce

import archi_amd.sharded as sh  # noqa: E402


def host_gather(group, world):
    def gather(payload):
        host = torch.empty((world * payload.numel(),), dtype=payload.dtype)
        dist.all_gather_into_tensor(host, payload.cpu(), group=group)
        return host.view(world, payload.numel()).to(payload.device)
    return gather


sh._rccl_all_gather = host_gather

if "WORLD_SIZE" not in os.environ:
    # plain `bench.py --gpus N` (no launcher): bench.py starts its own ranks. On the one-GPU test box the ranks must be
    # this shim again (gloo, all on cuda:0), and the node must look as if it had N devices.
    import subprocess
    _BENCH, _SHIM = os.path.join(ROOT, "bench.py"), os.path.abspath(__file__)
    _popen = subprocess.Popen

    class ShimPopen(_popen):
        def __init__(self, argv, *a, **kw):
            super().__init__([_SHIM if os.path.abspath(str(x)) == _BENCH else x for x in argv], *a, **kw)

    subprocess.Popen = ShimPopen
    torch.cuda.device_count = lambda: 64
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
