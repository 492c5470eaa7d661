#!/bin/bash
# Round-6 evidence visit (after the GPU suite is green): default bench + kernel trace + scan traffic (gpu_round.sh), FETCH / WRITE of the
# encoder kernels on the FINAL tree (round-5 review: the feature-block tile order had no after-measurement), per-kernel trace of the
# float32-grade modes, the query forward's parts.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
SKIP_TESTS=1 ROUND_OUT=gpurun_out/round bash scripts/gpu_round.sh > $O/round.log 2>&1
tail -3 $O/round.log
ENC_ARGS="BAAI/bge-base-en-v1.5 128 512" bash scripts/gpu_pmc_fetch_enc.sh > $O/pmc_fetch_bge.txt 2>&1
ENC_ARGS="sentence-transformers/all-MiniLM-L6-v2 256 256" bash scripts/gpu_pmc_fetch_enc.sh > $O/pmc_fetch_minilm.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -o x3 -- python3 scripts/gpu_probe_x3.py > $O/x3.log 2>&1
bash scripts/gpu_probe_qf_parts.sh > $O/qf_parts.log 2>&1
tail -4 $O/pmc_fetch_bge.txt; cat $O/x3.log | grep -v amdgpu; tail -6 $O/qf_parts.log
