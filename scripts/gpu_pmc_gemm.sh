#!/bin/bash
# MFMA-pipe occupancy of the wide GEMM tile on the bge-base forward: phased K-loop against the in-step loop (AK_GEMM_PHASED=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_gemm
for ph in 1 0; do
  export AK_GEMM_PHASED=$ph
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    rm -rf /tmp/pmc; rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 scripts/gpu_probe_enc1.py BAAI/bge-base-en-v1.5 128 512 > /tmp/pmc.out 2>&1
    f=$(find /tmp/pmc -name p_counter_collection.csv | head -1)
    python3 - "$f" "$ph" <<'PY'
import csv, sys, collections, json, os
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_gemm" in r["Kernel_Name"]:
        a = agg[(r["Kernel_Name"].split("(")[0].replace("void ak::", ""), r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = "gpurun_out/pmc_gemm/summary.json"
d = json.load(open(out)) if os.path.exists(out) else {}
for (k, c), (v, n) in sorted(agg.items()):
    d.setdefault("phased" if sys.argv[2] == "1" else "in_step", {}).setdefault(k, {})[c] = v / n
    print(f"phased={sys.argv[2]} {k:28s} {c:28s} launches {n:4d} per_launch {v / n:16.0f}")
json.dump(d, open(out, "w"), indent=1)
PY
  done
done
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_gemm/summary.json"))
for mode, ks in d.items():
    for k, c in ks.items():
        if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            act = c["GRBM_GUI_ACTIVE"] / 8
            print(f"{mode:8s} {k:28s} active cycles/XCD {act:10.0f}  MFMA pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (act * 1024):.3f}")
PY
