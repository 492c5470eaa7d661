cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
AK_GEMM_BN=256 AK_ENC_SKINNY_MAX=0 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "hf_fixture or oracle or bge_base" 2>&1 | tail -3
for rep in 1 2 3; do python3 scripts/gpu_probe_enc.py bge 128 20 2>&1 | grep forward; done
rm -rf /tmp/pr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o x -- python3 scripts/gpu_probe_enc.py bge 128 10 > /tmp/pr.out 2>&1
f=$(find /tmp/pr -name x_kernel_stats.csv | head -1); grep -E "k_gemm|k_attn|k_layernorm" $f | awk -F'","' '{print $1, "calls", $2, "avg_ns", $4}'
