"""GPU suite, early on purpose (fresh child processes; this process has not touched the GPU yet): bench.py's N > 1 path end
to end with 2 and 3 ranks on the one GPU of the test box (tests/bench_rehearsal.py swaps RCCL for gloo + host staging):
the JSON line must carry sharded_equals_single_index, the verified object and the replicated-corpus leg."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 3])
def test_bench_multirank_path(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "bench_rehearsal.py"), "--gpus", str(world),
                                       "--rows", "300000", "--queries", "300", "--steps", "3", "--warmup", "1",
                                       "--no-embed", "--no-cpu-baseline"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            raise
        outs.append((o.decode("utf-8", "replace"), e.decode("utf-8", "replace")))
    for rank, p in enumerate(procs):
        assert p.returncode == 0, f"rank {rank} failed:\n{outs[rank][1][-3000:]}"
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["steps"] == 3 and d["value"] > 0
    assert d["sharded_equals_single_index"] is True
    assert d["verified"]["equals_exact_hip_path"] and d["verified"]["equals_oracle_scan_of_slice"]
    assert d["config"]["rows_per_gpu"] in (300000 // world, 300000 // world + 1)
    assert "replicated_corpus" in d and "error" not in d["replicated_corpus"]
    assert d["roofline"]["launches_timed"] == 3
