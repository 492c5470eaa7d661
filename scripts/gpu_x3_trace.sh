# per-kernel trace of the split-bf16 mode at the bench's batch sizes (MiniLM 256 x 256, bge-base 128 x 512)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${X3_OUT:-r6z}; mkdir -p $O
for m in "sentence-transformers/all-MiniLM-L6-v2 256" "BAAI/bge-base-en 128"; do
  tag=$(echo $m | cut -d/ -f2 | cut -d' ' -f1)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o x3 -- python3 scripts/gpu_probe_x3_one.py $m > $O/$tag.log 2>&1
  f=$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $m"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']:>6}%")
PY
  cp "$f" $O/kernel_stats_$tag.csv
done
