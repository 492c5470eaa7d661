mkdir -p gpurun_out/r6x; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_02_encoder_variants_gpu.py -m gpu -x -q --durations=12 2>&1 | tail -25 > gpurun_out/r6x/variants.log; cat gpurun_out/r6x/variants.log
