#!/bin/bash
# per-kernel times of the encoder at given tile shapes: bash scripts/gpu_trace_tile.sh 682,96 512,96 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONPATH=.
for t in "$@"; do
  export AK_TILE=$t
  rm -rf /tmp/pt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python3 scripts/gpu_probe_tile.py > /dev/null 2> /tmp/pt.err
  echo "---- tile $t"
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pt/**/t_kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:9]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
done
