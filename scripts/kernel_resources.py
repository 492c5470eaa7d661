"""Compile one .hip file of archi_amd/csrc for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per
kernel: VGPRs, spills, scratch, SGPRs, occupancy (demangled names). python scripts/kernel_resources.py scan.hip [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-Wno-inline-asm", "-c", src, "-o", "/tmp/_kres.o", "-Rpass-analysis=kernel-resource-usage"]
p = subprocess.run(cmd, cwd=os.path.join(root, "archi_amd", "csrc"), stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
if p.returncode:
    print(p.stderr[-4000:]); sys.exit(1)
cur = None; rows = []
for ln in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
names = subprocess.run(["/usr/bin/c++filt"] + [r["name"] for r in rows], stdout=subprocess.PIPE, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    if flt and flt not in n: continue
    print(f"{n[:110]:110s} vgpr {r.get('VGPRs','?'):>3} agpr {r.get('AGPRs','?'):>3} vspill {r.get('VGPRs Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} "
          f"sgpr {r.get('TotalSGPRs','?'):>3} sspill {r.get('SGPRs Spill','?'):>3} occ {r.get('Occupancy [waves/SIMD]','?')}")
