#!/bin/bash
# PMC counters of one kernel (name substring $1) over a probe command ($2...): separate rocprofv3 passes per counter group.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
K=$1; shift
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc -o p -- "$@" > /tmp/pmc.out 2>&1
  f=$(find /tmp/pmc -name p_counter_collection.csv | head -1)
  python3 - "$f" "$K" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        a = agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()): print(f"{k:42s} {c:32s} launches {n:4d} per_launch {v / n:16.0f}")
PY
done
