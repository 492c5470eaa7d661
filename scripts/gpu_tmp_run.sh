cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for rep in 1 2 3; do
  echo -n "with next-input prefetch  "; python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
  echo -n "without                   "; AK_FFN_NOPFX=1 python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
done
AK_FFN_DBG=1 python3 scripts/gpu_probe_enc.py minilm 256 1 2>&1 | grep k_ffn | tail -3
