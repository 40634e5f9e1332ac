#!/bin/bash
# round 4, call q: after ABBABABA's sitesUsed = nan for windows without a good site -- the whole -m gpu suite, smoke, the C3 line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04q; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --no-tiers ) > $O/bench_c3.json 2> $O/bench_c3.err; tail -3 $O/bench_c3.err; cut -c1-600 $O/bench_c3.json
