#!/bin/bash
# per-kernel durations (rocprofv3) of the encoder at mid-size batches: MODEL=minilm|bge, BATCHES="24 32 64"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for b in ${BATCHES:-24 32 64 128}; do
  rm -rf /tmp/abl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o x -- python3 scripts/gpu_probe_enc.py ${MODEL:-minilm} $b 10 > /tmp/abl.out 2>&1
  grep forward /tmp/abl.out
  f=$(find /tmp/abl -name x_kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "ak::" in n and not any(x in n for x in ("relayout", "generate")):
        print(f"   {n.split('ak::')[1][:56]:56s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs']) / 1e3:8.1f}")
PY
done
