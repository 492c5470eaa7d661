#!/bin/bash
# HBM bytes (FETCH_SIZE / WRITE_SIZE, KB as counted; separate passes) and L2 hits / misses per encoder kernel launch; ENC_ARGS="model B S"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 scripts/gpu_probe_enc1.py ${ENC_ARGS:-BAAI/bge-base-en-v1.5 128 512} > /tmp/pmc.out 2>&1
  f=$(find /tmp/pmc -name p_counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "ak::" in r["Kernel_Name"] and "relayout" not in r["Kernel_Name"]:
        a = agg[(r["Kernel_Name"].split("(")[0].replace("void ak::", "")[:30], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()): print(f"{k:32s} {c:16s} launches {n:4d} per_launch {v / n:16.0f}")
PY
done
