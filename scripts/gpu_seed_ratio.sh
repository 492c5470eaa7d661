#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in 32 64 128 100000; do
  echo "== AK_SEED_RATIO=$r"
  for shape in "1250000 768 bf16 1024" "2500000 768 bf16 1024" "5000000 768 bf16 1024" "10000000 768 bf16 1024" "10000000 768 bf16 1" "1562500 384 f16 1024"; do
    AK_SEED_RATIO=$r python3 scripts/gpu_probe_search.py $shape 2>&1 | grep -v amdgpu.ids | sed "s/, plan.*ns_seed/ ns_seed/; s/, .seed_rows.*//"
  done
done
