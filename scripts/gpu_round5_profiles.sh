#!/bin/bash
# Round-5 evidence visit (after the GPU suite is green): default bench + kernel trace + scan traffic (gpu_round.sh), SQ counters and
# FETCH / WRITE of the attention kernels and the encoder GEMMs, SQ counters of the scan on the final tree, single-query kernel chain.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
SKIP_TESTS=1 ROUND_OUT=gpurun_out/round bash scripts/gpu_round.sh > $O/round.log 2>&1
tail -3 $O/round.log
bash scripts/gpu_pmc_kernel.sh k_attn_s python3 scripts/gpu_probe_enc1.py BAAI/bge-base-en-v1.5 128 512 > $O/pmc_attn_s_bge.txt 2>&1
bash scripts/gpu_pmc_kernel.sh k_attn_d python3 scripts/gpu_probe_enc1.py sentence-transformers/all-MiniLM-L6-v2 256 256 > $O/pmc_attn_d_minilm.txt 2>&1
ENC_ARGS="BAAI/bge-base-en-v1.5 128 512" bash scripts/gpu_pmc_fetch_enc.sh > $O/pmc_fetch_bge.txt 2>&1
ENC_ARGS="sentence-transformers/all-MiniLM-L6-v2 256 256" bash scripts/gpu_pmc_fetch_enc.sh > $O/pmc_fetch_minilm.txt 2>&1
PROBE_ARGS="10000000 768 bf16 1024:P" bash scripts/gpu_pmc_scan.sh r05_kscan > $O/pmc_scan.log 2>&1
bash scripts/gpu_trace_search.sh "1000000 384 f32 1" "1000000 384 f32 256" "10000000 768 bf16 1" > $O/trace_search.txt 2>&1
bash scripts/gpu_trace_enc.sh > $O/trace_enc.txt 2>&1
tail -5 $O/pmc_attn_s_bge.txt; tail -3 $O/pmc_fetch_bge.txt; cat $O/trace_search.txt | head -40
