#!/bin/bash
# attention + QKV launch times (kernel trace) under environment switches: each argument is one "VAR=value" setting ("-" = none)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python3 scripts/gpu_probe_enc1.py sentence-transformers/all-MiniLM-L6-v2 256 256 > /dev/null 2>&1
for setting in "${@:--}"; do
  rm -rf /tmp/abl
  if [ "$setting" != "-" ]; then export "$setting"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o x -- python3 scripts/gpu_probe_enc1.py sentence-transformers/all-MiniLM-L6-v2 256 256 > /tmp/abl.out 2>&1
  SETTING="$setting" python3 - <<'PY'
import csv, glob, os
f = glob.glob("/tmp/abl/**/x_kernel_stats.csv", recursive=True)
out = []
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in ("k_attn", "k_ffn384", "k_qkv384<")):
        out.append("%s %.1f us" % (r["Name"].split("(")[0].replace("void ak::", "").replace("ak::", ""), float(r["AverageNs"]) / 1e3))
print("%-24s" % os.environ["SETTING"], " | ".join(sorted(out)))
PY
  if [ "$setting" != "-" ]; then unset "${setting%%=*}"; fi
done
