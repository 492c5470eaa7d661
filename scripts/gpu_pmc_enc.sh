#!/bin/bash
# PMC on the encoder kernels (one counter set per run); ENC_ARGS="model B S"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmc_enc; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_enc/$name -o $name -- python3 scripts/gpu_probe_enc1.py $ENC_ARGS > gpurun_out/pmc_enc/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run grbm GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in sorted(glob.glob("gpurun_out/pmc_enc/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "ak::" not in r["Kernel_Name"]: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, d in agg.items():
    per = {c: v / cnt[k][c] for c, v in d.items()}
    wc = per.get("SQ_WAVE_CYCLES", 1)
    print(k, "| launches", max(cnt[k].values()))
    print("   wave-cycles split: wait_any %.0f%%  wait_inst %.0f%%  active %.0f%%  (wait_inst_lds %.0f%%)" % (
        100*per.get("SQ_WAIT_ANY",0)/wc, 100*per.get("SQ_WAIT_INST_ANY",0)/wc, 100*per.get("SQ_ACTIVE_INST_ANY",0)/wc, 100*per.get("SQ_WAIT_INST_LDS",0)/wc))
    g = per.get("GRBM_GUI_ACTIVE", 0)
    if g: print("   MFMA busy %.1f%% of GPU-active cycles; VALU/MFMA inst %.1f; LDS/MFMA %.2f; bank conflict cycles %.0f%% of LDS active" % (
        100*per.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(g*4*256/ (1 if True else 1)) if False else 100*per.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/max(per.get("SQ_BUSY_CYCLES",1),1),
        per.get("SQ_INSTS_VALU",0)/max(per.get("SQ_INSTS_MFMA",1),1), per.get("SQ_INSTS_LDS",0)/max(per.get("SQ_INSTS_MFMA",1),1),
        100*per.get("SQ_LDS_BANK_CONFLICT",0)/max(per.get("SQ_LDS_IDX_ACTIVE",1),1)))
PY
