#!/bin/bash
# round 3, call e: C4 bench line + where a C4 step spends its time, the default bench line with the new CPU and T2 legs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O
timeout 300 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 1500 $O/bench_c4.json
timeout 300 python tools/c4_host_time.py 10 > $O/c4_host_time.json 2> $O/c4_host_time.err; tail -c 1500 $O/c4_host_time.json; tail -3 $O/c4_host_time.err
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -c 7000 $O/bench_default.json; cat $O/bench_default.time; tail -3 $O/bench_default.err
