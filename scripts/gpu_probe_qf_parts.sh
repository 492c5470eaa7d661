#!/bin/bash
# Where the single-launch query forward's time goes: the whole kernel, barriers alone (AK_QF_SKIP=1), phase bodies alone (2): dbg library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for sk in 0 1 2; do
  echo "AK_QF_SKIP=$sk"; ARCHI_HIP_DBG=1 AK_QF_SKIP=$sk timeout 120 python scripts/gpu_probe_qf.py 2>&1 | grep -v amdgpu.ids
done
