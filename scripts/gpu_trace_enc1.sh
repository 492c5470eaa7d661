#!/bin/bash
# per-kernel timing of one encoder shape; ENC_ARGS="model B S", TRACE_NAME names the output
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/trace; export TMPDIR=/tmp
NAME=${TRACE_NAME:-enc1}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace/$NAME -o $NAME -- python3 scripts/gpu_probe_enc1.py $ENC_ARGS > gpurun_out/trace/$NAME.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/trace/$NAME/${NAME}_kernel_stats.csv")))
for r in rows[:9]:
    print(f"$NAME {r['Name'][:50]:50s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
