#!/bin/bash
# round 6, call b: the placement lottery at 64 slices with the planes allocated again; the counter passes behind
# profiles/r06 (kernel stats + FETCH / WRITE / SQ per workload; PG_PLACE_TRIALS=1: every launch in a pass is a real step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06b; mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout 900 python tools/pack_placement2.py 4 64 > $O/pack_placement2.txt 2> $O/pack_placement2.err; cut -c1-600 $O/pack_placement2.txt; tail -3 $O/pack_placement2.err
export PG_PLACE_TRIALS=1
for wl in northstar c2 c3 c4; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  tail -1 $O/bench_prof_$wl.log | cut -c1-160
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
done
find $O -name "*kernel_trace.csv" -size +20M -delete
ls $O/prof_stats | head -30; du -sh $O
