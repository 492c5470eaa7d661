#!/bin/bash
# One GPU visit: parity tests, default bench, rocprof kernel trace + stats of the same command,
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
# ROUND_OUT / BENCH_ARGS / SKIP_TESTS: the same visit for another workload (e.g. BENCH_ARGS="--queries 1 --no-embed")
O=${ROUND_OUT:-gpurun_out/round}
mkdir -p $O; export TMPDIR=/tmp
[ -z "$SKIP_TESTS" ] && python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py $BENCH_ARGS > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r01 -- python3 bench.py $BENCH_ARGS --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --no-verify > $O/bench_prof.json 2> $O/prof.err
cat $O/bench_prof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- python3 bench.py $BENCH_ARGS --steps 4 --warmup 1 --no-cpu-baseline --no-side-configs --no-verify --no-embed > $O/bench_fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- python3 bench.py $BENCH_ARGS --steps 4 --warmup 1 --no-cpu-baseline --no-side-configs --no-verify --no-embed > $O/bench_write.json 2> $O/write.err
# the encoder alone (all-MiniLM-L6 shape, 256 x 256 tokens): per-kernel table for profiles/<tag>_encoder_kernels.md
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc -o enc -- python3 scripts/gpu_probe_ffn.py 20 > $O/enc.out 2> $O/enc.err
cat $O/enc.out
python3 - <<'PY'
import csv, collections, json, os
O=os.environ.get("ROUND_OUT", "gpurun_out/round")
rows=list(csv.DictReader(open(f"{O}/prof/r01_kernel_stats.csv")))
print("---- kernel stats (rocprofv3 --kernel-trace --stats) ----")
for r in rows[:12]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
for name in ("fetch","write"):
    agg=collections.defaultdict(lambda: [0.0,0])
    for r in csv.DictReader(open(f"{O}/pmc_{name}/{name}_counter_collection.csv")):
        if "ak::k_scan" in r["Kernel_Name"]:
            key=(r["Counter_Name"], r["Grid_Size"])
            agg[key][0]+=float(r["Counter_Value"]); agg[key][1]+=1
    for (c,g),(v,n) in agg.items(): print(name, c, "grid", g, "launches", n, "per_launch", v/n)
PY
