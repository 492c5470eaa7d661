#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/trace; export TMPDIR=/tmp
python3 scripts/gpu_probe_enc.py 2>&1 | grep -v amdgpu.ids
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace/enc -o enc -- python3 scripts/gpu_probe_enc.py > gpurun_out/trace/enc.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/trace/enc/enc_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
