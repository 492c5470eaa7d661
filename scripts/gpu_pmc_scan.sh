#!/bin/bash
# SQ / GRBM counters of the scan kernel for one probe configuration (separate rocprofv3 passes per counter set):
#   PROBE_ARGS="10000000 768 bf16 1024:P" bash scripts/gpu_pmc_scan.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; T=${1:-scan}; O=gpurun_out/pmc_$T; mkdir -p $O; export TMPDIR=/tmp
ARGS="${PROBE_ARGS:-10000000 768 bf16 1024:P}"
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o $name -- python3 scripts/gpu_probe3.py $ARGS > $O/$name.log 2>&1; tail -1 $O/$name.log; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("$O/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "k_scan" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 256 * 512:
            agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g, d in agg.items():
        for c, v in d.items():
            # the main pass is the longest launch of each search: take the per-launch maximum group (seed launches are ~1/30)
            v = sorted(v); big = [x for x in v if x > 0.5 * v[-1]]
            out[c] = {"main_pass_per_launch": sum(big) / len(big), "launches": len(big)}
m = out
if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
    act = m["GRBM_GUI_ACTIVE"]["main_pass_per_launch"] / 8
    out["derived"] = {"gpu_active_cycles_per_xcd": act, "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"]["main_pass_per_launch"] / (act * 1024)}
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out.get("derived")), {k: round(v["main_pass_per_launch"] / 1e6, 1) for k, v in out.items() if k != "derived"})
PY
