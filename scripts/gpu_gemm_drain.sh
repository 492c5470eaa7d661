# how much of the hidden-768 GEMM launches is exposed store drain: AK_GEMM_ABLATE=16 (dbg library, timing only) lets the first two
# K-tiles of every tile run without waiting for the previous tile's stores; 32 = no ablation, same (non-lazy) launch path
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for f in 32 16 32 16; do
  AK_GEMM_ABLATE=$f TRACE_NAME=drain_$f ENC_ARGS="BAAI/bge-base-en-v1.5 128 512" bash scripts/gpu_trace_enc1.sh 2>&1 | grep -E "k_gemm|k_attn|layernorm" | sed "s/^/ABLATE=$f /"
done
