cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for b in 24 32 64 128; do
  rm -rf /tmp/abl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o x -- python3 scripts/gpu_probe_enc.py minilm $b 10 > /tmp/abl.out 2>&1
  tail -1 /tmp/abl.out | cut -c40-
  f=$(find /tmp/abl -name x_kernel_stats.csv | head -1)
  grep -E "k_embed|k_pool|k_attn|k_qkv384<|k_ffn384w8|skinny|k_layernorm" $f | awk -F'","' '{printf "   %-60s calls %s avg_us %.1f\n", substr($1,2,60), $2, $4/1000}'
done
