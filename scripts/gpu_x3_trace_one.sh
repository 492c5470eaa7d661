cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r6t2; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o x3 -- python3 scripts/gpu_probe_x3_one.py sentence-transformers/all-MiniLM-L6-v2 256 > $O/log.txt 2>&1
python3 - $O/prof <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/x3_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][-44:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:46s} n {len(v):4d} avg {sum(v)/len(v):8.1f} min {v2[0]:8.1f} med {v2[len(v)//2]:8.1f} max {v2[-1]:8.1f}")
# the gemm_ln calls alternate K = 1152 / 4608
g = [x for k, v in d.items() if "k_gemm_ln" in k for x in v]
print("gemm_ln even / odd calls (us):", sum(g[0::2]) / max(1, len(g[0::2])), sum(g[1::2]) / max(1, len(g[1::2])))
PY
