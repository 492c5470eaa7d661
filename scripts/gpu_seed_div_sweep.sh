#!/bin/bash
# Search time against the seeding pass's share of the rows (AK_SEED_DIV: 1/div of the tiles) and the pre-seeding sample
# (AK_PRE_DIV), for the shapes of the headline config and its 8-GPU shard.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for shape in "10000000 768 bf16 1024" "1250000 768 bf16 1024" "10000000 768 bf16 256"; do
  for d in 16 32 48 64 96 128; do
    echo -n "AK_SEED_DIV=$d  "; AK_SEED_DIV=$d python3 scripts/gpu_probe_search.py $shape 2>&1 | grep -v amdgpu.ids | sed "s/, plan.*//"
  done
  for d in 256 512 1024; do
    echo -n "AK_PRE_DIV=$d  "; AK_PRE_DIV=$d python3 scripts/gpu_probe_search.py $shape 2>&1 | grep -v amdgpu.ids | sed "s/, plan.*//"
  done
done
