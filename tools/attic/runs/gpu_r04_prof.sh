#!/bin/bash
# round 4: the whole -m gpu suite, then the profile passes behind profiles/r04 (kernel stats + four counter passes per workload;
# PG_PLACE_TRIALS=1: no placement probes, so every launch of a kernel in a pass is a real step), then the un-profiled bench lines
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r04prof; mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -4
export PG_PLACE_TRIALS=1
for wl in northstar c2 c3 c4; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  tail -1 $O/bench_prof_$wl.log | cut -c1-160
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/pmc_mfma -o $wl --output-format csv -- $B > $O/pmc_mfma_$wl.log 2>&1
done
unset PG_PLACE_TRIALS
for wl in c2 c3 c4; do
  timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-tiers > $O/bench_$wl.json 2> $O/bench_$wl.err
  tail -c 300 $O/bench_$wl.json; echo
done
timeout 900 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; tail -c 400 $O/bench_northstar_default.json; echo
PG_COMM=file timeout 600 python bench.py --gpus 2 --strong --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/bench_strong_2ranks_1gpu.json 2> $O/bench_strong2.err
ls $O/prof_stats | head; du -sh $O
