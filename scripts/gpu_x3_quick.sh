# split-bf16 mode: parity suite (both GEMM families forced) + forward time of both bench shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${X3_OUT:-r6q}; mkdir -p $O
timeout 900 python -m pytest tests/test_02_encoder_variants_gpu.py -m gpu -x -q -k split_bf16 2>&1 | tail -5 | tee $O/tests.log
timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_cfg1_gpu.py -m gpu -x -q -k "bf16x3 or split_bf16" 2>&1 | tail -5 | tee $O/tests_default.log
for m in "sentence-transformers/all-MiniLM-L6-v2 256" "BAAI/bge-base-en 128"; do
  for i in 1 2; do X3_TIME=1 python3 scripts/gpu_probe_x3_one.py $m 2>&1 | grep chunks | tee -a $O/time.txt; done
done
