#!/bin/bash
# per-kernel times of one search on a per-GPU shard of the 8-GPU configuration (1.25M x 768 bf16, Q=1024)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python3 bench.py --rows ${1:-1250000} --steps 20 --warmup 3 --no-embed --no-cpu-baseline > /tmp/ps.json 2> /tmp/ps.err
python3 - <<'PY'
import csv, glob, json
print(json.load(open("/tmp/ps.json"))["ms_per_step"])
f = glob.glob("/tmp/ps/**/s_kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:24]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
