#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for q in 1 16 64; do
for d in 32 64 128 256; do
  echo "Q=$q AK_SEED_DIV=$d:"; AK_SEED_DIV=$d python3 scripts/gpu_probe_search.py 10000000 768 bf16 $q 2>&1 | grep -v amdgpu | sed "s/, plan.*//"
done
echo "Q=$q no seed pass:"; AK_SEED_RATIO=100000 python3 scripts/gpu_probe_search.py 10000000 768 bf16 $q 2>&1 | grep -v amdgpu | sed "s/, plan.*//"
done
