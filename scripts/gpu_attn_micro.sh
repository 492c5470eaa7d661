#!/bin/bash
# attention microbenchmark matrix (scripts/micro/attn_bench): baseline kernels against k_attn_x variants, then correctness on
# padded / holed / spiked inputs and odd sequence lengths. Output: gpurun_out/attn_micro.txt
mkdir -p gpurun_out
OUT=gpurun_out/attn_micro.txt
: > $OUT
B=scripts/micro/attn_bench
run() { env "$@" 2>&1 | tail -1 >> $OUT; }
for rep in 1 2; do
run AK_ATTN_STREAM=1 $B 64 12 128 512 0
run AK_ATTN_STREAM=3 AK_ATTN_PIPE=0 $B 64 12 128 512 0
run AK_ATTN_STREAM=3 AK_ATTN_PIPE=1 $B 64 12 128 512 0
run AK_ATTN_STREAM=3 AK_ATTN_PIPE=2 $B 64 12 128 512 0
run AK_ATTN_STREAM=2 $B 32 12 256 256 0
run AK_ATTN_STREAM=3 AK_ATTN_PIPE=0 $B 32 12 256 256 0
run AK_ATTN_STREAM=3 AK_ATTN_PIPE=1 $B 32 12 256 256 0
run AK_ATTN_STREAM=3 AK_ATTN_PIPE=2 $B 32 12 256 256 0
done
for p in 0 1 2; do
  for m in 1 2 3; do
    run AK_ATTN_STREAM=3 AK_ATTN_PIPE=$p $B 64 12 16 512 $m 3
    run AK_ATTN_STREAM=3 AK_ATTN_PIPE=$p $B 32 12 16 256 $m 3
  done
  for s in 32 96 160 320 448; do
    run AK_ATTN_STREAM=3 AK_ATTN_PIPE=$p $B 64 12 5 $s 2 3
    run AK_ATTN_STREAM=3 AK_ATTN_PIPE=$p $B 32 12 5 $s 2 3
  done
done
run AK_ATTN_STREAM=1 $B 64 12 16 512 3 3
run AK_ATTN_STREAM=1 $B 64 12 16 512 2 3
cat $OUT
