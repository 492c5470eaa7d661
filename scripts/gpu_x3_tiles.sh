# split-bf16 mode on gemm.hip's tiles: parity tests (both tile widths forced, k3_gemm forced) + forward times
mkdir -p gpurun_out/r6y; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_02_encoder_variants_gpu.py -m gpu -x -q -k split_bf16 2>&1 | tail -30 > gpurun_out/r6y/tests.log; cat gpurun_out/r6y/tests.log
timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "bf16x3 or split_bf16" 2>&1 | tail -15 | tee gpurun_out/r6y/tests_default.log
timeout 600 python scripts/gpu_probe_x3.py 2>&1 | tee gpurun_out/r6y/probe.txt
AK_X3_TILES=0 timeout 600 python scripts/gpu_probe_x3.py 2>&1 | grep bf16x3 | tee gpurun_out/r6y/probe_k3.txt
