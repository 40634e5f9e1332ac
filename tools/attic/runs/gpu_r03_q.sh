#!/bin/bash
# round 3, call q: profile passes of the workloads whose called-count kernel is now k_pairC_big (north-star shape, C2), one rank's
# share of config 5 on one GPU, the default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r03prof; mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma
export PG_PLACE_TRIALS=1
for wl in c2 northstar; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  tail -1 $O/bench_prof_$wl.log | cut -c1-160
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/pmc_mfma -o $wl --output-format csv -- $B > $O/pmc_mfma_$wl.log 2>&1
done
PG_PAIR_CLOCK=1 timeout 120 python bench.py --workload northstar --steps 3 --warmup 1 --no-cpu-baseline --no-tiers 2> $O/pair_clock_northstar.txt > /dev/null; grep k_pairC_big $O/pair_clock_northstar.txt | tail -3
unset PG_PLACE_TRIALS
timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 300 $O/bench_c2.json; echo
timeout 600 python tools/c5_share.py 3 > $O/c5_share.txt 2>&1; tail -5 $O/c5_share.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -c 600 $O/bench_default.json; cat $O/bench_default.time | tail -3
du -sh $O
