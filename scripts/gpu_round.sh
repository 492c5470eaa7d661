#!/bin/bash
# One GPU visit: parity tests, default bench, rocprof kernel trace of the same command.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
tail -2 gpurun_out/prof.err; cat gpurun_out/bench_prof.json
find gpurun_out/prof -name "*stats*" | head; 
