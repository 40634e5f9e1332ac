#!/bin/bash
# round 3, call o: the matrix-instruction micro-benchmark, the pair kernel's clock probe, the kernel tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03o; mkdir -p $O
timeout 120 ./gpurun_variants/mfma_rate > $O/mfma_rate.txt 2>&1
PG_PAIR_CLOCK=1 PG_PLACE_TRIALS=1 timeout 120 python bench.py --workload northstar --steps 3 --warmup 1 --no-cpu-baseline --no-tiers 2>&1 | grep -E "k_pairC_big<" | tail -3
PG_PAIR_CLOCK=1 timeout 120 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --no-tiers 2>&1 | grep -E "k_pairC_big<" | tail -2
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
