# split-bf16 mode, hidden 384: which output widths to pad onto the wide tile (AK_X3_PADN bits: 1 QKV, 2 out-projection, 4 FFN-down)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r6z2; mkdir -p $O
timeout 900 python -m pytest tests/test_02_encoder_variants_gpu.py -m gpu -x -q -k split_bf16 2>&1 | tail -5 | tee $O/tests.log
for pn in 0 1 4 5 7 0 5; do
  X3_TIME=1 X3_TAG="PADN=$pn" AK_X3_PADN=$pn python3 scripts/gpu_probe_x3_one.py sentence-transformers/all-MiniLM-L6-v2 256 2>&1 | grep chunks | tee -a $O/padn.txt
done
