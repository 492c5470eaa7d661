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
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_scan" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        # the main pass is the longest launch of each search (pre-seeding and seeding launches are < 1/20 of it)
        v = sorted(v); big = [x for x in v if x >= 0.5 * v[-1]]
        out[c] = {"main_pass_per_launch": sum(big) / max(len(big), 1), "launches": len(big)}
if "GRBM_GUI_ACTIVE" in out and "SQ_VALU_MFMA_BUSY_CYCLES" in out:
    act = out["GRBM_GUI_ACTIVE"]["main_pass_per_launch"] / 8           # summed over the 8 XCDs
    out["derived"] = {"gpu_active_cycles_per_xcd": act,
                      "mfma_busy_frac": out["SQ_VALU_MFMA_BUSY_CYCLES"]["main_pass_per_launch"] / (act * 1024),
                      "note": "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); clock = active cycles per XCD / launch time"}
import re
ms = [float(m.group(1)) for f in glob.glob("$O/*.log") for m in re.finditer(r"scan\s+([0-9.]+) ms", open(f).read())]
if ms and "derived" in out:
    out["derived"]["scan_ms_under_profiler"] = sum(ms) / len(ms)
    out["derived"]["clock_GHz"] = out["derived"]["gpu_active_cycles_per_xcd"] / (sum(ms) / len(ms) * 1e-3) / 1e9
out["probe"] = "$ARGS"
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out.get("derived")))
PY
