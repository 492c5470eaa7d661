#!/bin/bash
# A/B of the role-split layer kernel k_ffn384r against k_ffn384p on one box: bit-identity test, forward times per knob set,
# section cycle stamps, per-kernel durations under rocprofv3.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/ffn_role; mkdir -p $O
python -m pytest tests/test_02_encoder_variants_gpu.py -x -q -k "wave_pair" 2>&1 | tail -3
for rep in 1 2 3; do
  for v in ${KNOBS:-11 10 1 0 411 410 401 400}; do
    echo -n "AK_FFN_R=$v  "; AK_FFN_R=$v python3 scripts/gpu_probe_enc.py minilm 256 60
  done
  echo -n "pair kernel  "; AK_FFN_ROLE=0 python3 scripts/gpu_probe_enc.py minilm 256 60
done 2>&1 | tee $O/ab.txt
AK_FFN_DBG=1 python3 scripts/gpu_probe_enc.py minilm 256 1 2>&1 | tail -4 | tee $O/dbg.txt
for v in ${PROF_KNOBS:-11 0}; do
  rm -rf /tmp/pr; AK_FFN_R=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o x -- python3 scripts/gpu_probe_enc.py minilm 256 10 > /tmp/pr.out 2>&1
  f=$(find /tmp/pr -name x_kernel_stats.csv | head -1); echo "== rocprof AK_FFN_R=$v"; grep -E "k_attn|k_qkv384<|k_ffn384" $f | awk -F'","' '{print $1, "calls", $2, "avg_ns", $4}'
done 2>&1 | tee $O/prof.txt
rm -rf /tmp/pr; AK_FFN_ROLE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o x -- python3 scripts/gpu_probe_enc.py minilm 256 10 > /tmp/pr.out 2>&1
f=$(find /tmp/pr -name x_kernel_stats.csv | head -1); echo "== rocprof pair"; grep -E "k_attn|k_qkv384<|k_ffn384" $f | awk -F'","' '{print $1, "calls", $2, "avg_ns", $4}' | tee -a $O/prof.txt
