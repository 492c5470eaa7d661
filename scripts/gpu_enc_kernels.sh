cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for i in 1 2 3 4; do python3 scripts/gpu_probe_enc.py minilm 256 20; done
rm -rf /tmp/abl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o x -- python3 scripts/gpu_probe_enc.py minilm 256 6 > /tmp/abl.out 2>&1
f=$(find /tmp/abl -name x_kernel_stats.csv | head -1)
grep -E "k_embed|k_pool|k_attn|k_qkv384<|k_ffn384w8" $f | awk -F'","' '{print $1, "calls", $2, "avg_ns", $4}'
