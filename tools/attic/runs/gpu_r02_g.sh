#!/bin/bash
# GPU call G: smoke, suite, default bench line (with t1_packed), T2 bench (native .pgeno inflate).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 2500 $O/bench_default.json; tail -3 $O/bench_default.err
( time timeout 900 python tools/t2_bench.py 10000000 100 ) > $O/t2_10M_100.txt 2>&1
grep -E "^run 0|^packed run|end to end" $O/t2_10M_100.txt | cut -c1-420
