cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for d in ${DBGS:-0 1 2 3}; do
  export AK_QKV_DBG=$d
  rm -rf /tmp/abl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o x -- python3 scripts/gpu_probe_enc.py minilm 256 6 > /tmp/abl.out 2>&1
  f=$(find /tmp/abl -name x_kernel_stats.csv | head -1)
  echo "dbg=$d $(tail -1 /tmp/abl.out | cut -c50-) :: $(grep 'k_qkv384<' $f | awk -F'","' '{print $1, "avg_ns", $4}')"
done
