#!/bin/bash
# round 4, call w: the whole -m gpu suite with k_popdist_np in place, then the 300-seed sweep of random command lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04w; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | grep -v "were tested\|were written\|^$\|Done" | tail -12
PG_FUZZ_SEEDS=300 timeout 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $O/fuzz300.log; grep -n "^E  .*AssertionError\|^E    .*row\|passed\|failed" $O/fuzz300.log | cut -c1-420 | tail -40
