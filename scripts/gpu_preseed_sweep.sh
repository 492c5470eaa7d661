#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "default:"; python3 scripts/gpu_probe_search.py 1250000 768 bf16 1024 2>&1 | grep -v amdgpu | sed "s/, plan.*//"
for d in 16 32 64 128 256; do
  echo "no seed pass, AK_PRE_DIV=$d:"; AK_SEED_RATIO=100000 AK_PRE_DIV=$d python3 scripts/gpu_probe_search.py 1250000 768 bf16 1024 2>&1 | grep -v amdgpu | sed "s/, plan.*//"
done
for d in 8 16 64; do
  echo "seed pass AK_SEED_DIV=$d:"; AK_SEED_DIV=$d python3 scripts/gpu_probe_search.py 1250000 768 bf16 1024 2>&1 | grep -v amdgpu | sed "s/, plan.*//"
done
