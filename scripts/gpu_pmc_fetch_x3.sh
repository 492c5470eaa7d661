#!/bin/bash
# HBM bytes (FETCH_SIZE / WRITE_SIZE, KB as counted; FETCH_SIZE x2 on gfx950 per the guide, WRITE_SIZE as counted; separate passes) per kernel launch of the split-bf16
# mode at the bench's batch sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for m in "sentence-transformers/all-MiniLM-L6-v2 256" "BAAI/bge-base-en 128"; do
echo "== $m"
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 scripts/gpu_probe_x3_one.py $m > /tmp/pmc.out 2>&1
  f=$(find /tmp/pmc -name p_counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "ak::" in r["Kernel_Name"] and "split_" not in r["Kernel_Name"]:
        a = agg[(r["Kernel_Name"].split("(")[0].replace("void ak::", "")[:34], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()): print(f"{k:36s} {c:12s} launches {n:4d} per_launch_KB_as_counted {v / n:14.0f}  MB {v / n * (2 if c == 'FETCH_SIZE' else 1) / 1024:10.1f}" + (" (x2: wide coalesced reads count half on gfx950)" if c == "FETCH_SIZE" else ""))
PY
done
done
