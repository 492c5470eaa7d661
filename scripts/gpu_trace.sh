#!/bin/bash
# per-kernel timing (rocprofv3 --kernel-trace --stats) of a probe command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/trace; export TMPDIR=/tmp
NAME=${TRACE_NAME:-t}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace/$NAME -o $NAME -- python3 scripts/gpu_probe3.py $PROBE_ARGS > gpurun_out/trace/$NAME.log 2>&1
grep "nq=" gpurun_out/trace/$NAME.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/trace/$NAME/${NAME}_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} total_us={float(r['TotalDurationNs'])/1e3:10.1f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
