#!/bin/bash
# round 3, call r: one rank's share of config 5 with its sub-batches one after the other (default now) and pipelined on two streams
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O
timeout 600 python tools/c5_share.py 3 > $O/c5_share_sequential.txt 2>&1; tail -3 $O/c5_share_sequential.txt
PG_OVERLAP=1 timeout 600 python tools/c5_share.py 3 > $O/c5_share_two_streams.txt 2>&1; tail -3 $O/c5_share_two_streams.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
