#!/bin/bash
# round 4, call i: `.pgeno` with raw cells from the file to the device (pg_stage_file / pg_unpack_staged): goldens, e2e, then tier T2 from packed input
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -8
timeout 900 python tools/t2_pgeno_bench.py 25000000 200 > $O/t2_pgeno.txt 2>&1; cat $O/t2_pgeno.txt
