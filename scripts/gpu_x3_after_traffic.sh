cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
X3_OUT=r6r bash scripts/gpu_x3_quick.sh
timeout 300 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "fp32_parity" 2>&1 | tail -2
bash scripts/gpu_pmc_fetch_x3.sh 2>&1 | grep -E "==|k3_attn|k_gemm" | tee gpurun_out/r6r/pmc_fetch_x3.txt
