cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python -m pytest tests/test_02_encoder_variants_gpu.py -x -q -m gpu -k "bit_identical or NWV" 2>&1 | tail -3
for rep in 1 2 3; do
  echo -n "role kernel  "; python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
  echo -n "pair kernel  "; AK_FFN_ROLE=0 python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
done
AK_FFN_DBG=1 python3 scripts/gpu_probe_enc.py minilm 256 1 2>&1 | grep k_ffn | tail -3
