#!/bin/bash
# PMC passes on the scan kernel (separate runs per counter set, as the guide prescribes).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
ARGS="${PROBE_ARGS:-10000000 768 bf16 1024:L}"
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc/$name -o $name -- python3 scripts/gpu_probe3.py $ARGS > gpurun_out/pmc/$name.log 2>&1; tail -2 gpurun_out/pmc/$name.log; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run tcc3 TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    print("==", f)
    for k, d in agg.items():
        if "k_scan" in k or "scan" in k:
            print(k, {c: v for c, v in d.items()})
PY
