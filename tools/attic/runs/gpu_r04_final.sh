#!/bin/bash
# round 4, last call: the whole -m gpu suite, smoke, and sweeps of the three fuzz tests with the generator as committed
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04final; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | grep "passed\|failed\|real"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
PG_FUZZ_SEEDS=400 PG_FUZZ_LONG_SEEDS=200 PG_FUZZ_RANK_SEEDS=12 timeout 1300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $O/fuzz_all.log; grep -n "^E  .*Error\|^E    .*row\|passed\|failed" $O/fuzz_all.log | cut -c1-500 | tail -12
