#!/bin/bash
# GPU call K: rocprofv3 stats + PMC passes of the north-star shape (one batch per pass, no side measurements), default bench line.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k
mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
wl=northstar
B="python bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-tiers"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
tail -1 $O/bench_prof_$wl.log | cut -c1-100
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
rm -f $O/*/*agent_info.csv $O/prof_stats/*kernel_trace.csv
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
