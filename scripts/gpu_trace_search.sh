#!/bin/bash
# per-kernel times of device-resident searches: bash scripts/gpu_trace_search.sh "rows dim dtype Q" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for spec in "$@"; do
  rm -rf /tmp/pq
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o s -- python3 scripts/gpu_probe_search.py $spec > /tmp/pq.out 2> /tmp/pq.err
  tail -1 /tmp/pq.out
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pq/**/s_kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    if int(r['Calls']) >= 50:
        print(f"   {r['Name'][:96]:96s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
done
