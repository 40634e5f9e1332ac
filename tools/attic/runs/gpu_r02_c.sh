#!/bin/bash
# GPU call C of round 2: -m gpu suite on the pipelined ingestion, default bench line with tiers, C2 line, T2 text bench.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 4200 $O/bench_default.json; tail -4 $O/bench_default.err
( time timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 ) > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 3000 $O/bench_c2.json; tail -3 $O/bench_c2.err
( time timeout 900 python tools/t2_bench.py 10000000 100 ) > $O/t2_10M_100.txt 2>&1
cat $O/t2_10M_100.txt | cut -c1-700
