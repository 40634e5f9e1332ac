#!/bin/bash
# GPU call D of round 2: suite, rocprofv3 stats + PMC passes of the final kernel selection (c2, northstar), T2 text bench, bench lines.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j
mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
for wl in c2 northstar; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  tail -1 $O/bench_prof_$wl.log | cut -c1-100
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
done
rm -f $O/*/*agent_info.csv $O/prof_stats/*kernel_trace.csv
( time timeout 900 python tools/t2_bench.py 10000000 100 ) > $O/t2_10M_100.txt 2>&1
cat $O/t2_10M_100.txt | cut -c1-600
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1800 $O/bench_default.json; tail -3 $O/bench_default.err
echo "== overlap experiment: north-star shape forced into >= 8 pipelined sub-batches"
PG_OVERLAP=1 timeout 300 python bench.py --workload northstar --steps 6 --warmup 2 --no-cpu-baseline --no-tiers 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('PG_OVERLAP=1', d['ms_per_step'], d.get('kernel_ms_per_step'))"
timeout 300 python bench.py --workload northstar --steps 6 --warmup 2 --no-cpu-baseline --no-tiers 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('one batch   ', d['ms_per_step'], d.get('kernel_ms_per_step'))"
