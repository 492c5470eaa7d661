#!/bin/bash
# A/B of the hidden-384 layer kernels on one box: k_ffn384r (table GELU, default) / k_ffn384r with the polynomial GELU
# (AK_FFN_GELU=poly: bit-identical to the pair kernel) / k_ffn384p (AK_FFN_ROLE=0): bit-identity test, forward times in
# alternation, section cycle stamps (needs `make dbg`), per-kernel durations under rocprofv3.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/ffn_role; mkdir -p $O
python -m pytest tests/test_02_encoder_variants_gpu.py -x -q -k "bit_identical or wave_pair" 2>&1 | tail -1 | tee $O/ident.txt
for rep in 1 2 3; do
  echo -n "k_ffn384r table  "; python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
  echo -n "k_ffn384r poly   "; AK_FFN_GELU=poly python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
  echo -n "k_ffn384p        "; AK_FFN_ROLE=0 python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
done 2>&1 | tee $O/ab.txt
if [ -f archi_amd/lib/libarchi_hip_dbg.so ]; then
  AK_FFN_DBG=1 python3 scripts/gpu_probe_enc.py minilm 256 1 2>&1 | grep k_ffn | tail -2 | tee $O/dbg.txt
fi
rm -rf /tmp/pr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o x -- python3 scripts/gpu_probe_enc.py minilm 256 10 > /tmp/pr.out 2>&1
f=$(find /tmp/pr -name x_kernel_stats.csv | head -1)
grep -E "k_attn|k_qkv384<|k_ffn384" $f | awk -F'","' '{print $1, "calls", $2, "avg_ns", $4}' | tee $O/prof.txt
