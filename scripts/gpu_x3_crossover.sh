# split-bf16 mode: from how many tokens on gemm.hip's tiles beat k3_gemm (AK_X3_TILES=2 forces the tiles, 0 forces k3_gemm)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r6x3c; mkdir -p $O
for B in ${MINILM_B:-2 4 8 16 32 48 64 96 128}; do for t in 0 2; do
  X3_TIME=1 X3_TAG="TILES=$t" AK_X3_TILES=$t python3 scripts/gpu_probe_x3_one.py sentence-transformers/all-MiniLM-L6-v2 $B 2>&1 | grep chunks | tee -a $O/cross.txt
done; done
for B in ${BGE_B:-1 2 4 8 16 24 32 48 64}; do for t in 0 2; do
  X3_TIME=1 X3_TAG="TILES=$t" AK_X3_TILES=$t python3 scripts/gpu_probe_x3_one.py BAAI/bge-base-en $B 2>&1 | grep chunks | tee -a $O/cross.txt
done; done
