#!/bin/bash
# round 2, call p: device tokenizer -- goldens through the drivers, T2 timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02p
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_kernels.py -m gpu -x -q -k "device_tokenizer" > gpurun_out/r02p/pytest.log 2>&1; tail -5 gpurun_out/r02p/pytest.log
timeout 900 python tools/t2_bench.py 10000000 100 > gpurun_out/r02p/t2.log 2>&1; cat gpurun_out/r02p/t2.log
