cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
AK_ENC_SKINNY_MAX=0 AK_ENC_LAZYLN=2 timeout 1200 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "hf_fixture or oracle or bge_base" 2>&1 | tail -15
for rep in 1 2; do
python3 scripts/gpu_probe_enc.py bge 128 20 2>&1 | grep forward
AK_ENC_LAZYLN=0 python3 scripts/gpu_probe_enc.py bge 128 20 2>&1 | grep forward
done
