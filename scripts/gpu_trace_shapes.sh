#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of device-resident searches at several shapes: "rows dim dtype Q" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
i=0
for shape in "$@"; do
  i=$((i+1)); rm -rf /tmp/ps$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps$i -o s -- python3 scripts/gpu_probe_search.py $shape > /tmp/ps$i.out 2> /tmp/ps$i.err
  cat /tmp/ps$i.out
  python3 - /tmp/ps$i <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/s_kernel_stats.csv", recursive=True)
tot = 0.0
for r in list(csv.DictReader(open(f[0])))[:16]:
    print(f"  {r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
done
