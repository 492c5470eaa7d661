cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r6q2
X3_OUT=r6q2 bash scripts/gpu_x3_trace.sh > gpurun_out/r6q2/trace.txt 2>&1
bash scripts/gpu_pmc_kernel.sh k3_attn python3 scripts/gpu_probe_x3_one.py sentence-transformers/all-MiniLM-L6-v2 256 > gpurun_out/r6q2/pmc_attn_minilm.txt 2>&1
cat gpurun_out/r6q2/trace.txt gpurun_out/r6q2/pmc_attn_minilm.txt
