#!/bin/bash
# k_tail phase ablation (timings only; results are wrong with ablation on): AK_TAIL_ABLATE bit0 sort, bit1 loads+chain, bit2 chain, bit3 final select
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for a in 0 1 2 4 8 15; do
  echo "== AK_TAIL_ABLATE=$a"
  AK_TAIL_ABLATE=$a python3 scripts/gpu_probe_search.py 1000000 384 f32 1
  AK_TAIL_ABLATE=$a python3 scripts/gpu_probe_search.py 1250000 768 bf16 1024
done
echo "== old tail"
AK_TAIL_OLD=1 python3 scripts/gpu_probe_search.py 1000000 384 f32 1
AK_TAIL_OLD=1 python3 scripts/gpu_probe_search.py 1250000 768 bf16 1024
