#!/bin/bash
# GPU call B of round 2: microbenchmark, the -m gpu suite, benches on correctly filled data, rocprofv3 stats + PMC passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02b
mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
tools/valu_rate > $O/valu_rate.txt 2>&1
head -14 $O/valu_rate.txt
( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1
tail -22 $O/pytest_gpu.log
show='
import sys, json
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln); print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d.get("kernel_ms_per_step"))
    elif "rror" in ln: print(ln.strip())
'
for wl in c2 northstar; do
  for v in PG_NONE=1 PG_PACK2=1 PG_NONE=1; do
    echo "== $wl $v"
    env $v timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tee -a $O/ab_$wl.log | python -c "$show"
  done
done
for wl in c2 northstar; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  tail -1 $O/bench_prof_$wl.log | cut -c1-120
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
done
rm -f $O/*/*agent_info.csv
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 2500 $O/bench_default.json; tail -4 $O/bench_default.err
( timeout 300 python bench.py --workload c4 --steps 5 ) > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 1200 $O/bench_c4.json
( timeout 300 python bench.py --workload c3 ) > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 1200 $O/bench_c3.json
du -sh $O
