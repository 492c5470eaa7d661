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


def test_plain_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (what a driver may type): bench.py must start the two ranks itself and
    print ONE JSON line with n_gpus == 2 (through the rehearsal shim: the test box has one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(HERE, "bench_rehearsal.py"), "--gpus", "2", "--rows", "200000",
                        "--queries", "300", "--steps", "2", "--warmup", "1", "--no-embed", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode("utf-8", "replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["sharded_equals_single_index"] is True


def test_plain_bench_refuses_more_ranks_than_gpus():
    """No silent one-rank run: asking for more GPUs than the node has exits non-zero before anything touches a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "64", "--steps", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"refusing" in p.stderr
