#!/bin/bash
# ridge-regime sweep: search time by tile configuration for Q between the HBM-bound and the MFMA-bound regimes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for shape in "1000000 384 f32" "10000000 768 bf16" "1250000 768 bf16"; do
  for q in 96 128 160 192 256 320 384 512; do
    for cfg in X L M; do
      AK_SCAN_CFG=$cfg python3 scripts/gpu_probe_search.py $shape $q 2>&1 | grep -v amdgpu.ids | sed "s/ per search.*nslices.: \([0-9]*\), .nqg.: \([0-9]*\).*/ cfg=$cfg ns=\1 nqg=\2/"
    done
  done
done
