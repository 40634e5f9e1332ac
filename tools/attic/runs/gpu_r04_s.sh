#!/bin/bash
# round 4, call s: thetaPi as the reference's sequential sum (k_popfreq_ordered) -- the whole -m gpu suite incl. the random command lines, the popFreq bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | grep -v "were tested\|were written\|^$\|Done" | tail -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py --workload popfreq --steps 10 --warmup 2 --no-cpu-baseline --no-tiers ) > $O/bench_popfreq.json 2> $O/bench_popfreq.err; tail -3 $O/bench_popfreq.err; cut -c1-900 $O/bench_popfreq.json
