"""GPU suite, FIRST file on purpose: the row-sharded search (archi_amd/sharded.py) at world size 2 and 3 with the real
per-shard HIP search and merge kernels, against the CPU oracle and the single-index result, on adversarial corpora
(duplicate pile-ups, NaN rows, a zero query, a clustered corpus, shards below the MFMA scan's row floor, empty shards,
WHERE masks). The ranks are fresh child processes started before this process has touched the GPU (a process that has
initialised the GPU must never be replaced by another program on this pool; children are fine) -- hence the file name,
which sorts before every other GPU test, and no `hip` fixture here."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_real_kernels(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # the C path of the same exchange (ak_comm_create + ak_index_search_sharded_dev) runs beside the torch path in every rank,
    # over tests/native/fake_rccl.cpp: RCCL itself refuses several ranks on the one GPU of this box (round-5 review, missing #4)
    root = os.path.dirname(HERE)
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "archi_amd", "csrc"), "fake-rccl"])
    fake = os.path.join(root, "archi_amd", "csrc", "build", "libfake_rccl.so")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", AK_RCCL_PATH=fake, FAKE_RCCL_TIMEOUT_S="120")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_gpu_worker.py"),
                                       str(tmp_path / f"rank{rank}.json")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            raise
        outs.append(o.decode("utf-8", "replace"))
    for rank, p in enumerate(procs):
        assert p.returncode == 0, f"rank {rank} failed:\n{outs[rank][-3000:]}"
    for rank in range(world):
        rep = json.load(open(tmp_path / f"rank{rank}.json"))
        bad = {k: v for k, v in rep.items() if not v["ok"]}
        assert not bad, f"rank {rank}: {bad}"
        assert len(rep) >= 15 and rep["store_api"]["ok"] and rep["dp_embedding"]["ok"] and rep["failure_agreement"]["ok"], rep.get("dp_embedding")
        assert rep["abi_failures"]["ok"] and "broken" in rep["abi_failures"]["steps"], rep["abi_failures"]
        assert all(v["abi_equal"] for v in rep.values() if "abi_equal" in v)
        assert sum(1 for v in rep.values() if v.get("abi_equal")) >= 11, "the C path did not run"
