#!/bin/bash
# A/B of the role-split layer kernel k_ffn384r against k_ffn384p on one box: bit-identity test, forward times per knob set,
# section cycle stamps, per-kernel durations under rocprofv3.   KNOBS="100 110 ..." DBGKNOBS=".." PROF_KNOBS=".."
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/ffn_role; mkdir -p $O
for v in ${KNOBS:-100 110 101 111 1101 1111 4011 11}; do
  echo -n "bit-identity AK_FFN_R=$v: "; AK_FFN_R=$v python -m pytest tests/test_02_encoder_variants_gpu.py -x -q -k "wave_pair" 2>&1 | tail -1
done 2>&1 | tee $O/ident.txt
for rep in 1 2 3; do
  for v in ${KNOBS:-100 110 101 111 4101 4111 4011 11}; do
    echo -n "AK_FFN_R=$v  "; AK_FFN_R=$v python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
  done
  echo -n "pair kernel  "; AK_FFN_ROLE=0 python3 scripts/gpu_probe_enc.py minilm 256 60 2>&1 | grep forward
done 2>&1 | tee $O/ab.txt
for v in ${DBGKNOBS:-100 111 4111}; do AK_FFN_R=$v AK_FFN_DBG=1 python3 scripts/gpu_probe_enc.py minilm 256 1 2>&1 | grep k_ffn | tail -2; done | tee $O/dbg.txt
for v in ${PROF_KNOBS:-111}; do
  rm -rf /tmp/pr; AK_FFN_R=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o x -- python3 scripts/gpu_probe_enc.py minilm 256 10 > /tmp/pr.out 2>&1
  f=$(find /tmp/pr -name x_kernel_stats.csv | head -1); echo "== rocprof AK_FFN_R=$v"; grep -E "k_attn|k_qkv384<|k_ffn384" $f | awk -F'","' '{print $1, "calls", $2, "avg_ns", $4}'
done 2>&1 | tee $O/prof.txt
