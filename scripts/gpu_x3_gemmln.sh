# split-bf16 mode, hidden 384: out-projection / FFN-down with the LayerNorm fused (gemm_ln.hip X3) against MODE 5 + k3_add_ln
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r6q8; mkdir -p $O
timeout 900 python -m pytest tests/test_02_encoder_variants_gpu.py -m gpu -x -q -k split_bf16 2>&1 | tail -5 | tee $O/tests.log
timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "bf16x3 or split_bf16" 2>&1 | tail -5 | tee $O/tests_default.log
for f in 1 0 1 0; do X3_TIME=1 X3_TAG="GEMMLN=$f" AK_X3_GEMMLN=$f python3 scripts/gpu_probe_x3_one.py sentence-transformers/all-MiniLM-L6-v2 256 2>&1 | grep chunks | tee -a $O/time.txt; done
