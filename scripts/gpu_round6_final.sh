#!/bin/bash
# Final round-6 visit: the whole GPU suite, smoke, the default bench line, and the per-kernel trace of the parity modes
# (scripts/gpu_probe_x3.py: f32 / bf16x3 / bf16 at the bench's batch sizes) on the final tree.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 > $O/suite.log; cat $O/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; head -c 600 $O/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -o x3 -- python3 scripts/gpu_probe_x3.py > $O/x3.log 2>&1
grep -v amdgpu $O/x3.log
